import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import mixture_nd
m = mixture_nd.BimodeGauss_ND(8)
tid, tp, d = m._bpm_target_spec()
R, N = 8, 262144
kw = dict(burnin_gen=110, n_cr_gen=20, outlier_every=50)
np.random.seed(15)
x0 = m.rvs(N)
parked = np.arange(0, N, 4099)
x0[parked] = 25.0
def single():
    one = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=31, **kw)
    one.set_state(x0); one.begin_run()
    out = []
    for seg in (50, 50, 30):
        one.step(seg); out.append((one.stats()["n_outlier_resets"], float(one.get_state().sum())))
    one.close(); return out
def ranks():
    uid = b"BPMLOCAL" + bytes(120)
    rs = [HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=31, rank=r, world_size=R, nccl_uid=uid, **kw) for r in range(R)]
    for e in rs:
        e.set_state(x0); e.begin_run()
    arr = (C.c_void_p * R)(*[e._h for e in rs])
    out = []
    for seg in (50, 50, 30):
        L.check(rs[0].lib.bpm_local_group_step(arr, R, seg))
        out.append(([e.stats()["n_outlier_resets"] for e in rs], float(rs[0].get_state().sum())))
    for e in rs: e.close()
    return out
print("single", single()); print("single", single()); print("ranks", ranks()); print("ranks", ranks())
