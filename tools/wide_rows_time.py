import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from bipymc_amd import _lib as L
from bipymc_amd.engine import HipEngine
from bipymc_amd.utils import d100_gauss
for d, N, G in [(100, 8192, 500), (256, 8192, 300), (512, 8192, 200), (640, 8192, 200), (1000, 8192, 100), (1024, 8192, 100), (1500, 4096, 100), (2048, 4096, 100)]:
    tid, tp, dd = d100_gauss.Gauss_100D(dim=d)._bpm_target_spec()
    e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=1, burnin_gen=0, keep_history=True)
    rs = np.random.RandomState(0)
    X0 = np.sqrt(np.arange(d) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    e.set_state(X0)
    e.reserve_history(3 * G + 4)
    e.begin_run()
    e.step(G); e.synchronize()
    t0 = time.perf_counter(); e.step(G); e.synchronize(); dt = (time.perf_counter() - t0) / G
    B = 8 * d * 9 + 16
    st = e.stats()
    print("d=%5d N=%5d: %8.1f us per generation, %.2e chain-updates/s, algorithmic %.2f TB/s (%.2f of 8), acc %.3f, %s" % (
        d, N, dt * 1e6, N / dt, N * B / dt / 1e12, N * B / dt / 8e12, st["local_n_accepted"] / (st["local_n_accepted"] + st["local_n_rejected"]), e.launch_stats()["direct"] > 0), flush=True)
    e.close()
