#!/usr/bin/env python3
"""bench.py -- chain-updates/s of the DREAM hot path on the 100-D Gaussian (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W

One "step" = one generation = every chain updated once (two half-generation kernel launches,
plus, for N > 1, the RCCL all-gather of the state after each).  N = 1: DREAM, 100-D
equicorrelated Gaussian, n_chains = 8192, del_pairs = 3 (BASELINE configs[1]).  N > 1: weak
scaling, 8192 chains per GPU sharded by contiguous global-id blocks (configs[3] at N = 8), one
process per GPU launched by torch.distributed.run.

Timed region = steady state after burn-in (CR adaptation finished), history append included,
inputs resident in HBM; bracketed by barrier + synchronize on both sides, max over ranks.
Before burn-in the GPU is kept busy for a fixed wall time (--preheat seconds, default 0.5) with untimed steady-state
generations of the same workload on a scratch sampler that keeps no history: a fresh box idles at low clocks, and a
timed region of 20 generations is 0.25 ms long (tools/window_anatomy.py, profiles/r02_window_anatomy.txt).
The posterior gate reported with the number is evaluated over at least POSTERIOR_MIN_GENS post-burn-in generations
(the timed ones plus an untimed extension when --steps is short).
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHAINS_PER_GPU = 8192
DIM = 100
DEL_PAIRS = 3
BURNIN_GEN = 200
N_CR_GEN = 50
# algorithmic bytes per chain-update, f64, steady state, history kept:
# 8*d*(1 own read + 2P partner reads + 1 state write + 1 history append) + 16 (cached ln_like r/w)
BYTES_PER_UPDATE = 8 * DIM * (2 * DEL_PAIRS + 3) + 16          # 7216
HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: 8 TB/s spec
POSTERIOR_MIN_GENS = 1200                                      # post-burn-in generations the moment gate is evaluated over


def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle timed on this host on a bounded sample of the same workload (N=8192, d=100, steady
    state): the plain-C + OpenMP restatement (oracle/csrc/dream_ref.c, checked against the NumPy oracle in
    tests/test_oracle_c.py) on the box's CPU share, and the NumPy oracle (1 process) for reference."""
    from oracle import dream_ref_c as CR
    from oracle import sampler_ref as R
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(DIM) + 1.0))
    rs = np.random.RandomState(0)
    X = np.sqrt(np.arange(DIM) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((CHAINS_PER_GPU, 1))
                                         + np.sqrt(0.5) * rs.standard_normal((CHAINS_PER_GPU, DIM)))
    threads = max(1, min(16, os.cpu_count() or 1, CR.max_threads()))      # a one-GPU box's CPU share is 16 cores
    out = {}
    for label, nt in (("c1", 1), ("cN", threads)):
        Xc = X.copy()
        ll = R.ll_gauss_equicorr(Xc, params)
        CR.dream_run(Xc, ll, params, 42, 0, 0, 2, del_pairs=DEL_PAIRS, n_threads=nt)     # warm-up
        gens, el, t0 = 0, 0.0, time.perf_counter()
        while el < seconds_budget / 3 and gens < 2000:
            CR.dream_run(Xc, ll, params, 42, 2 + gens, 2 + gens, 10, del_pairs=DEL_PAIRS, n_threads=nt)
            gens += 10
            el = time.perf_counter() - t0
        out[label] = (CHAINS_PER_GPU * gens / el, gens, el)
    ora = R.OracleSampler(R.ALGO_DREAM, CHAINS_PER_GPU, DIM, R.TARGET_GAUSS_EQUICORR, params, 42,
                          del_pairs=DEL_PAIRS, burnin_gen=0, n_cr_gen=N_CR_GEN)
    ora.set_state(X)
    ora.run(1)
    gens, el, t0 = 0, 0.0, time.perf_counter()
    while el < seconds_budget / 3 and gens < 200:
        ora.run(1)
        gens += 1
        el = time.perf_counter() - t0
    return dict(value=out["cN"][0], unit="chain-updates/s", cores=threads, kind="port",
                sample="%d generations of DREAM d=100 n_chains=8192 in %.1f s with oracle/csrc/dream_ref.c on %d OpenMP threads "
                       "(1 thread: %.3g chain-updates/s; NumPy oracle, 1 process: %.3g chain-updates/s over %d generations); "
                       "host has %d cores" % (out["cN"][1], out["cN"][2], threads, out["c1"][0],
                                              CHAINS_PER_GPU * gens / el, gens, os.cpu_count() or 0))


def measured_copy_bandwidth(torch, dev):
    """Device-to-device copy of 1 GiB (read + write = 2 GiB of HBM traffic per pass), HIP-event timed: the
    achievable-bandwidth yardstick SURVEY section 8(d) asks for beside the 8 TB/s datasheet figure."""
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device="cuda:%d" % dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    return 2.0 * n * 8 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    global CHAINS_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-moments", action="store_true")
    ap.add_argument("--preheat", type=float, default=0.5, help="seconds of untimed steady-state generations before burn-in (0: none)")
    ap.add_argument("--chains-per-gpu", type=int, default=CHAINS_PER_GPU, help="experiments only; the default is the BASELINE workload")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))

    import torch                      # first: the HIP runtime is then shared with libbipymc_hip.so
    ndev = torch.cuda.device_count()
    if ndev > 0 and local_rank >= ndev:   # launcher exposed one device per process
        local_rank = 0
    dist = None
    # BPM_FORCE_DIST=1 takes the multi-process path with a single rank (process group + one-rank RCCL
    # communicator): the only way to rehearse it on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("BPM_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils.d100_gauss import Gauss_100D

    CHAINS_PER_GPU = args.chains_per_gpu
    n_chains = CHAINS_PER_GPU * world
    target = Gauss_100D(rho=0.5, dim=DIM)
    tid, tparams, _ = target._bpm_target_spec()
    uid = None
    if use_dist:
        box = [HipEngine.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    eng = HipEngine(algo=L.ALGO_DREAM, n_chains=n_chains, dim=DIM, target_id=tid, target_params=tparams, seed=42,
                    device=local_rank, rank=rank, world_size=world, nccl_uid=uid,
                    del_pairs=DEL_PAIRS, burnin_gen=BURNIN_GEN, n_cr_gen=N_CR_GEN, n_cr=3)
    # Synthetic start: exact draws of the target (x_i = sigma_i (sqrt(rho) g + sqrt(1-rho) e_i)), so the
    # timed region is the stationary regime and the moment gate below tests invariance.  (From the
    # reference's default start -- theta_0 + 1e-3 jitter -- or an independent over-dispersed one the
    # population needs ~1000 generations to find the correlated scale: tools/convergence_check.py.)
    rs = np.random.RandomState(1234)
    X0 = np.sqrt(np.arange(DIM) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((n_chains, 1))
                                          + np.sqrt(0.5) * rs.standard_normal((n_chains, DIM)))
    eng.set_state(X0)
    total_gens = BURNIN_GEN + max(args.warmup + args.steps + 32, POSTERIOR_MIN_GENS) + 64
    eng.reserve_history(1 + total_gens)

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    eng.begin_run()
    # ---- pre-heat: the same steady-state kernels on a scratch sampler without history, for a fixed wall time
    preheat = dict(seconds=0.0, generations=0)
    if args.preheat > 0:
        heat = HipEngine(algo=L.ALGO_DREAM, n_chains=CHAINS_PER_GPU, dim=DIM, target_id=tid, target_params=tparams, seed=7,
                         device=local_rank, del_pairs=DEL_PAIRS, burnin_gen=0, n_cr_gen=N_CR_GEN, n_cr=3, keep_history=False)
        heat.set_state(X0[:CHAINS_PER_GPU])
        heat.begin_run()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.preheat:
            heat.step(500)
            heat.synchronize()
            preheat["generations"] += 500
        preheat["seconds"] = time.perf_counter() - t0
        heat.close()
    # ---- burn-in with CR adaptation: timed separately, never part of `value`
    fence()
    t0 = time.perf_counter()
    eng.step(BURNIN_GEN)
    fence()
    burn_s = time.perf_counter() - t0
    # ---- warm-up: W generations, the last of them through the same timed entry point as the timed region (the first
    # event-bound dispatch of a process sets up profiling signals: 10-30 us of host time, once)
    if args.warmup > 1:
        eng.step(args.warmup - 1)
    if args.warmup > 0:
        eng.step_timed(1)          # (reads the events too: every host-side path of the timed call has run once)
    fence()
    # ---- timed region: exactly K generations.  Wall clock for `value`; for the kernel's per-launch duration two HIP events
    # bound to the first and the last update-kernel dispatch of the same K generations on the sampler's own stream
    # (bpm_step_timed: end of launch K/2 -> end of launch 2K, back-to-back launch periods of the last three quarters of the timed
    # region -- binding an event to a dispatch costs the host 10-30 us, which would stall the GPU at the head of the region; it
    # returns with the sampler's stream drained).
    t0 = time.perf_counter()
    eng.step_timed(args.steps, read=False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    ev_ms, ev_launches = eng.last_step_time()
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    value = n_chains * args.steps / el

    # ---- dominant kernel (phase_fused_kernel): 2 launches per generation, back to back on one stream; at
    # N = 1 nothing else runs in the region, so event time / launches is its average launch duration
    # (inter-launch gaps included; rocprofv3 --kernel-trace gives the gap-free figure, profiles/).
    n_launch = ev_launches
    k_avg_ms = ev_ms / max(n_launch, 1)
    units_per_launch = CHAINS_PER_GPU / 2.0                       # half the local chains per launch
    achieved = units_per_launch * BYTES_PER_UPDATE / (k_avg_ms * 1e-3) / 1e9
    pair_ms, pair_n = eng.step_profiled(32)                       # cross-check: an event pair around every launch
    fence()
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_cfg2.json")   # HBM bytes per launch from rocprofv3 --pmc (offline)
    if os.path.exists(tfile) and world == 1 and CHAINS_PER_GPU == 8192:
        traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")

    extra = {"evaluated": False}
    if not args.no_moments:
        # parity gate reported with the number: posterior moments of the post-burn-in rows vs the
        # analytic ones (mean 0, var_i = i+1), from the on-device reduction over this rank's rows.  A short timed
        # region (the driver's 20 generations) is extended by untimed generations: 52 correlated generations say nothing.
        post = args.warmup + args.steps + 32
        if post < POSTERIOR_MIN_GENS:
            eng.step(POSTERIOR_MIN_GENS - post)
            fence()
            post = POSTERIOR_MIN_GENS
        n_burn = (1 + BURNIN_GEN) * n_chains
        cnt, s1, s2, sh = eng.reduce_moments(n_burn)
        if dist is not None:
            pack = torch.tensor(np.concatenate([[cnt], s1, s2]), dtype=torch.float64, device="cuda")
            dist.all_reduce(pack)
            pack = pack.cpu().numpy()
            cnt, s1, s2 = pack[0], pack[1:1 + DIM], pack[1 + DIM:]
        mean = sh + s1 / cnt
        var = s2 / cnt - (s1 / cnt) ** 2
        sig2 = np.arange(DIM) + 1.0
        st = eng.stats()
        acc = st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"])
        vr = float(np.mean(var / sig2))
        mm = float(np.max(np.abs(mean) / np.sqrt(sig2)))
        extra = dict(evaluated=True, generations=int(post), rows=int(cnt),
                     # the gate: pooled variance ratio within 1 % of the analytic value, every mean within 0.05 sigma
                     # (tests/test_gpu_api.py::test_posterior_moments_at_baseline_sizes is the asserted form)
                     var_ratio_mean=vr, max_abs_mean_over_sigma=mm, gate_pass=bool(abs(vr - 1.0) < 0.01 and mm < 0.05),
                     var_ratio_min=float(np.min(var / sig2)), var_ratio_max=float(np.max(var / sig2)),
                     acceptance_fraction=acc, p_cr=[float(v) for v in st["p_cr"]])

    import hashlib
    state_sha = hashlib.sha256(np.ascontiguousarray(eng.get_state()).tobytes()).hexdigest()[:16]   # (A/B of launch paths: same bits)
    lstat = eng.launch_stats()
    if rank == 0:
        copy_gbs = measured_copy_bandwidth(torch, local_rank)
        out = {
            "metric": "chain-updates/sec", "value": value, "unit": "chain-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            # the same K generations by the HIP event pair on the sampler's stream (no host launch / wake-up latency)
            "value_event_timed": n_chains / (2.0 * k_avg_ms * 1e-3),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "DREAM, 100-D equicorrelated Gaussian (tests/test_100dgauss.py target), "
                                   "n_chains=%d (%d per GPU), del_pairs=3, n_cr=3, steady state after %d burn-in "
                                   "generations, history appended every generation; GPU pre-heated for %.2f s (%d untimed "
                                   "steady-state generations on a scratch sampler) before burn-in"
                                   % (n_chains, CHAINS_PER_GPU, BURNIN_GEN, preheat["seconds"], preheat["generations"]),
                       "n_chains": n_chains, "dim": DIM, "parallelism": "chains sharded x%d" % world,
                       "burnin_updates_per_s": n_chains * BURNIN_GEN / burn_s,
                       "exchange": eng.exchange_stats() if use_dist else None,
                       # how the update kernels were dispatched: packets written by the library into its own AQL queue
                       # (bipymc_amd/csrc/aql_queue.h) or launches on the HIP stream (burn-in, multi-GPU, BPM_DIRECT_QUEUE=0)
                       "update_dispatches": {"direct_aql_queue": lstat["direct"], "hip_stream": lstat["stream"],
                                             "state_in_hw_coherent_memory": lstat["coherent_state"], "packet_fence": lstat["fence"]},
                       "final_state_sha256_16": state_sha},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "bpm::phase_fused_kernel<1, 1, 64, 2, 3, %d> (DREAM, Gauss target, 64 lanes/chain, 3 pairs, steady-state instantiation%s)" % ((5, " of the sharded launch mode") if world > 1 else (1, "")),
                         "bytes_per_unit": BYTES_PER_UPDATE,
                         "units_per_launch": units_per_launch, "avg_launch_us": k_avg_ms * 1e3,
                         "launches_timed": n_launch, "avg_launch_us_event_pairs": pair_ms / pair_n * 1e3,
                         "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs},
            "posterior": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
