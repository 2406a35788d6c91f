#!/usr/bin/env python3
"""bench.py -- chain-updates/s of the DREAM hot path on the 100-D Gaussian (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W

One "step" = one generation = every chain updated once (two half-generation kernel launches,
plus, for N > 1, the exchange of the accepted updates after each).  N = 1: DREAM, 100-D
equicorrelated Gaussian, n_chains = 8192, del_pairs = 3 (BASELINE configs[1]).  N > 1: weak
scaling, 8192 chains per GPU sharded by contiguous global-id blocks (configs[3] at N = 8), one
process per GPU.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts
its N rank processes ITSELF, as children (never an exec; the parent makes no GPU call), relays
rank 0's JSON line and returns non-zero when a rank fails; under torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) it is one of the ranks.

Timed region = steady state after burn-in (CR adaptation finished), history append included,
inputs resident in HBM; bracketed by barrier + synchronize on both sides, max over ranks.
Before burn-in the GPU is kept busy for a fixed wall time (--preheat seconds, default 0.5) with untimed steady-state
generations of the same workload on a scratch sampler that keeps no history: a fresh box idles at low clocks, and a
timed region of 20 generations is 0.25 ms long (tools/window_anatomy.py, profiles/r02_window_anatomy.txt).
The posterior gate reported with the number is evaluated over at least POSTERIOR_MIN_GENS post-burn-in generations
(the timed ones plus an untimed extension when --steps is short).
At N = 1 the line also carries `configs`: the other BASELINE configurations that fit one GPU (cfg3, cfg5's per-GPU
share, cfg5 whole, cfg2 / cfg5 burn-in), each a few hundred generations timed OUTSIDE the headline's region.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
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
# ---- bench.py cannot die silent ---------------------------------------------------------------------------
# Every stage of a run has a wall-clock limit, and the whole run an absolute one (BENCH_DEADLINE_S, default 480 s: below the 600 s the driver gives
# the command).  When a limit is hit the rank's watchdog (StageWatch) writes ONE parseable JSON line -- the headline measured so far, or a line
# with "value": null -- naming the stage and the collective that was pending, and the process exits with EXIT_WATCHDOG: the line is kept, the
# return code is not 0.  The launcher of an N > 1 run (launch_ranks) relays that line, ends the other ranks by PID, and -- when the stall was in
# the push exchange's connection / validation / timed region and time is left -- starts a FRESH set of child processes once under the dense RCCL
# all-gather (never a re-exec of a process that touched the GPU).
EXIT_WATCHDOG = 125
DEADLINE_S = float(os.environ.get("BENCH_DEADLINE_S", "480"))
STAGE_LIMITS_S = dict(init=150.0, create=60.0, connect=75.0, validation=90.0, headline=120.0, posterior=90.0, alternatives=120.0, extras=150.0,
                      teardown=45.0)
METRIC_NAME = "chain-updates/sec"


WATCH = None          # the rank's StageWatch (main() sets it): helpers name the collective they are about to block in


def _pend(what):
    if WATCH is not None:
        WATCH.pending(what)


def null_line(n_gpus, steps=None, warmup=None, **why):
    """the line of a run that measured nothing: same keys as a measurement, value null, and what happened"""
    d = {"metric": METRIC_NAME, "value": None, "unit": "chain-updates/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup, "ms_per_step": None,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
         "config": {"workload": "DREAM, 100-D equicorrelated Gaussian, %d chains per GPU (BASELINE configs[1] / [3]): NOT MEASURED" % CHAINS_PER_GPU}}
    d.update(why)
    return d


class StageWatch(object):
    """One per rank process.  enter(stage) starts the stage's clock (its own limit, cut to the run's absolute deadline); pending(what) names the
    collective / library call the rank is about to block in; headline(line) hands over rank 0's measured line as soon as it exists.  fire() --
    from the timer thread, the main thread may be stuck inside a collective with the GIL released -- writes rank 0's ONE line (the headline so far
    with a "watchdog" entry, or a null line) and ends the process with EXIT_WATCHDOG.  No teardown is attempted: the process is hung."""

    def __init__(self, rank, world, write_line, steps=None, warmup=None, deadline_at=None, limits=None, exit_fn=os._exit, err=None, clock=time.time):
        import threading
        self.rank, self.world, self.write_line, self.steps, self.warmup = rank, world, write_line, steps, warmup
        self.deadline_at = deadline_at if deadline_at is not None else clock() + DEADLINE_S
        self.limits = dict(STAGE_LIMITS_S, **(limits or {}))
        self.exit_fn, self.err, self.clock = exit_fn, err or sys.stderr, clock
        self.stage, self.what, self.line, self.t_stage, self.limit, self.history = "start", None, None, clock(), None, []
        self._timer, self._lock, self.fired = None, threading.Lock(), False

    def enter(self, stage, limit_s=None):
        import threading
        with self._lock:
            now = self.clock()
            if self.stage != "start":
                self.history.append((self.stage, round(now - self.t_stage, 2)))
            if self._timer is not None:
                self._timer.cancel()
            self.stage, self.what, self.t_stage = stage, None, now
            limit = self.limits.get(stage, 120.0) if limit_s is None else limit_s
            self.limit = max(0.05, min(limit, self.deadline_at - now))
            self._timer = threading.Timer(self.limit, self.fire)
            self._timer.daemon = True
            self._timer.start()
        if os.environ.get("BENCH_TEST_STALL") == "%s:%d" % (stage, self.rank):      # rehearsal hook (tests/test_bench_launcher.py): this rank hangs here
            self.what = "BENCH_TEST_STALL"
            time.sleep(10 ** 6)

    def pending(self, what):
        self.what = what

    def headline(self, line):
        self.line = line

    def info(self):
        return dict(fired=True, rank=self.rank, stage=self.stage, pending=self.what, stage_limit_s=round(self.limit or 0.0, 1),
                    seconds_in_stage=round(self.clock() - self.t_stage, 1), stages_done=self.history,
                    note="a stage of bench.py ran into its wall-clock limit: the process was ended by its own watchdog (exit code %d)" % EXIT_WATCHDOG)

    def fire(self):
        with self._lock:
            if self.fired:
                return
            self.fired = True
        info = self.info()
        try:
            self.err.write("bench.py: rank %d: watchdog: stage %r exceeded %.0f s (pending: %s)\n" % (self.rank, self.stage, self.limit or 0.0, self.what))
            if self.rank == 0:
                line = dict(self.line) if self.line else null_line(self.world, self.steps, self.warmup)
                line["watchdog"] = info
                if not self.line:
                    line["error"] = "stage %r did not finish within %.0f s (pending: %s)" % (self.stage, self.limit or 0.0, self.what)
                self.write_line(json.dumps(line))
        finally:
            self.exit_fn(EXIT_WATCHDOG)

    def done(self):
        with self._lock:
            if self._timer is not None:
                self._timer.cancel()
                self._timer = None


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts the rank processes (reference: one MPI rank per chain block in lock-step,
# bipymc/demc.py:14-32,93-94,116-117 -- `mpirun -n N python script.py`; here `python bench.py --gpus N`)
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count(timeout=180.0):
    """hipGetDeviceCount, asked in a CHILD process (bpm_device_count of the product library): the launching parent itself
    never initialises HIP.  -1 when the probe cannot run (library not built)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from bipymc_amd import _lib\n"
            "print('BPM_NDEV', _lib.device_count())\n" % ROOT)
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except Exception:
        return -1
    for line in out.stdout.decode("utf-8", "replace").splitlines():
        if line.startswith("BPM_NDEV"):
            return int(line.split()[1])
    return -1


def _launch_once(n, cmd, out_lines, err, grace_s, deadline_at, rank_deadline_at):
    """one set of n fresh child processes -> (rc, last JSON line of rank 0 or None).  rc 124: ended at the wall-clock limit."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BPM_BENCH_CHILD="1", BENCH_DEADLINE_AT="%.3f" % rank_deadline_at)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL and the IPC-mapped exchange buffers need here
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else err, stderr=err))
    first_fail_t = None
    rc = 0
    # rank 0's stdout is read in a thread so that a full pipe never blocks it
    import threading
    got = []

    def reader():
        for raw in procs[0].stdout:
            got.append(raw.decode("utf-8", "replace"))
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0:
                err.write("bench.py: rank %d exited with code %d%s\n" % (r, c, " (its watchdog fired: a stage ran into its wall-clock limit)" if c == EXIT_WATCHDOG else ""))
                rc = rc or (c if 0 < c < 256 else 1)
                if first_fail_t is None:
                    first_fail_t = time.time()
        timed_out = live and time.time() > deadline_at
        if live and ((first_fail_t is not None and time.time() - first_fail_t > grace_s) or timed_out):
            for r in sorted(live):
                err.write("bench.py: ending rank %d (pid %d): %s\n" % (r, procs[r].pid, "the run exceeded its wall-clock limit" if timed_out else "another rank failed"))
                procs[r].kill()
            for r in sorted(live):
                procs[r].wait()
            live.clear()
            if timed_out:
                rc = rc or 124
        if live:
            time.sleep(0.05)
    th.join(timeout=5.0)
    line = None
    for text in got:
        t = text.strip()
        if t.startswith("{") and t.endswith("}"):
            line = t
        elif t:
            err.write(text if text.endswith("\n") else text + "\n")
    return rc, line


def launch_ranks(n, argv, worker_cmd=None, n_visible=None, grace_s=15.0, out=sys.stdout, err=sys.stderr, deadline_s=None, retry_min_s=150.0):
    """Start n rank processes of this script as children and relay rank 0's last JSON line to `out`, everything else to `err`.  EXACTLY ONE line
    reaches `out` whatever happens -- rank 0's, or a line with "value": null written here that says what went wrong (fewer GPUs than ranks, a rank
    that died, a run that hit the wall-clock limit).  Returns 0 only when every rank exited 0 and rank 0 printed a line with a value.
    Fewer visible GPUs than ranks is an error before anything starts (RCCL refuses two ranks on one device; no silent oversubscription).  When one
    rank fails the others are given grace_s seconds, then ended (their exact PIDs) -- a rank that died inside a collective would leave its peers
    waiting for ever.  The wall-clock limit deadline_s (BENCH_LAUNCH_DEADLINE_S, else BENCH_DEADLINE_S = 480 s: below the driver's own limit)
    runs from the moment this function is entered, the GPU-count probe included; the ranks get an absolute deadline 25 s earlier
    (BENCH_DEADLINE_AT), so their own watchdogs -- which can still write the line -- fire first: exit code 125 then, 124 when the parent had to end
    them.  A run whose ranks stalled in the push exchange's connection, validation or timed region is repeated ONCE, in fresh child processes,
    under the dense RCCL all-gather (north_star's exchange) when at least retry_min_s seconds are left; the line then lists the first attempt.
    worker_cmd / n_visible: test hooks (tests/test_bench_launcher.py)."""
    t_start = time.time()
    if deadline_s is None:
        deadline_s = float(os.environ.get("BENCH_LAUNCH_DEADLINE_S", str(DEADLINE_S)))
    deadline_at = t_start + deadline_s

    def steps_of(a):
        try:
            return int(a[a.index("--steps") + 1]), int(a[a.index("--warmup") + 1])
        except (ValueError, IndexError):
            return None, None

    def emit(line_dict):
        out.write(json.dumps(line_dict) + "\n")
        out.flush()
    if n_visible is None:
        n_visible = visible_gpu_count(timeout=min(180.0, max(5.0, deadline_s / 3.0)))
    steps, warmup = steps_of(list(argv))
    if n_visible < n:
        msg = ("--gpus %d but %s GPU(s) visible to this process: one process per GPU, no oversubscription (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES "
               "narrow the set)" % (n, "no" if n_visible <= 0 else str(n_visible)))
        err.write("bench.py: " + msg + "\n")
        emit(null_line(n, steps, warmup, error=msg))
        return 2
    attempts = []
    cur_argv = list(argv)
    while True:
        cmd = (list(worker_cmd) if worker_cmd else [sys.executable, os.path.abspath(__file__)]) + cur_argv
        rc, line = _launch_once(n, cmd, None, err, grace_s, deadline_at, deadline_at - min(25.0, 0.25 * deadline_s))
        parsed = None
        if line is not None:
            try:
                parsed = json.loads(line)
            except ValueError:
                parsed = None
        wd = (parsed or {}).get("watchdog") or {}
        has_value = parsed is not None and parsed.get("value") is not None
        left = deadline_at - time.time()
        can_retry = (not attempts and not has_value and wd.get("stage") in ("connect", "validation", "headline") and left >= retry_min_s
                     and "--exchange" not in cur_argv and "--share-gpu" not in cur_argv)
        if can_retry:
            attempts.append(dict(argv=cur_argv, rc=rc, watchdog=wd, seconds=round(time.time() - t_start, 1)))
            err.write("bench.py: the ranks stalled in stage %r of the default (push) exchange: starting %d FRESH rank processes under the dense RCCL "
                      "all-gather (%.0f s left)\n" % (wd.get("stage"), n, left))
            cur_argv = cur_argv + ["--exchange", "dense", "--no-exchange-alternatives"]
            continue
        break
    if parsed is None:
        why = ("rank 0 exited without printing its JSON line" if rc == 0 else
               ("the run exceeded its wall-clock limit of %.0f s and its ranks were ended" % deadline_s if rc == 124 else "a rank failed (exit code %d)" % rc))
        err.write("bench.py: %s\n" % why)
        parsed = null_line(n, steps, warmup, error=why)
        rc = rc or 1
    elif not has_value:
        rc = rc or 1
    if attempts:
        parsed["launcher"] = dict(attempts_before_this_one=attempts,
                                  note="the first set of rank processes stalled; this line comes from a fresh set under --exchange dense")
    emit(parsed)
    return rc


# ---------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=16.0):
    """The CPU oracle timed on this host on a bounded sample of the same workload (N=8192, d=100, steady state): the
    plain-C + OpenMP restatement (oracle/csrc/dream_ref.c, checked against the NumPy oracle in tests/test_oracle_c.py) on
    1 core, on a one-GPU box's CPU share (<= 16 cores: `value`) and on all cores this process may use, and the NumPy
    oracle (1 process).  BASELINE.md section 5."""
    from oracle import dream_ref_c as CR
    from oracle import sampler_ref as R
    params = R.gauss_equicorr_params(0.5, np.sqrt(np.arange(DIM) + 1.0))
    rs = np.random.RandomState(0)
    X = np.sqrt(np.arange(DIM) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((CHAINS_PER_GPU, 1))
                                         + np.sqrt(0.5) * rs.standard_normal((CHAINS_PER_GPU, DIM)))
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cores_all = max(1, min(usable, CR.max_threads()))
    threads = max(1, min(16, cores_all))                                  # a one-GPU box's CPU share is 16 cores
    legs = [("c1", 1), ("cN", threads)]
    if cores_all != threads:
        legs.append(("call", cores_all))
    out = {}
    per_leg = seconds_budget / (len(legs) + 1)
    for label, nt in legs:
        Xc = X.copy()
        ll = R.ll_gauss_equicorr(Xc, params)
        CR.dream_run(Xc, ll, params, 42, 0, 0, 2, del_pairs=DEL_PAIRS, n_threads=nt)     # warm-up
        gens, el, t0 = 0, 0.0, time.perf_counter()
        step = 2 if nt == 1 else 10
        while el < per_leg and gens < 4000:
            CR.dream_run(Xc, ll, params, 42, 2 + gens, 2 + gens, step, del_pairs=DEL_PAIRS, n_threads=nt)
            gens += step
            el = time.perf_counter() - t0
        out[label] = (CHAINS_PER_GPU * gens / el, gens, el)
    if "call" not in out:
        out["call"] = out["cN"]
    ora = R.OracleSampler(R.ALGO_DREAM, CHAINS_PER_GPU, DIM, R.TARGET_GAUSS_EQUICORR, params, 42,
                          del_pairs=DEL_PAIRS, burnin_gen=0, n_cr_gen=N_CR_GEN)
    ora.set_state(X)
    ora.run(1)
    gens, el, t0 = 0, 0.0, time.perf_counter()
    while el < per_leg and gens < 200:
        ora.run(1)
        gens += 1
        el = time.perf_counter() - t0
    return dict(value=out["cN"][0], unit="chain-updates/s", cores=threads, kind="port",
                value_1core=out["c1"][0], value_allcores=out["call"][0], cores_all=cores_all,
                value_numpy_1proc=CHAINS_PER_GPU * gens / el,
                cpu_model=_cpu_model(), cpu_count_host=os.cpu_count() or 0,
                sample="DREAM d=100 n_chains=8192 steady state with oracle/csrc/dream_ref.c (C + OpenMP over the chains of a "
                       "half generation): %d generations in %.1f s on %d threads (value), %d in %.1f s on 1 thread, %d in %.1f s "
                       "on %d threads (all cores this process may use); NumPy oracle %d generations in %.1f s"
                       % (out["cN"][1], out["cN"][2], threads, out["c1"][1], out["c1"][2], out["call"][1], out["call"][2],
                          cores_all, gens, el))


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


# ---------------------------------------------------------------------------------------------------------------------
# The other BASELINE configurations that fit one GPU, under the same clock.  bytes = SURVEY 8(d).
# ---------------------------------------------------------------------------------------------------------------------
OTHER_CONFIG_PREHEAT_S = 0.1      # untimed steady-state generations of a scratch sampler (no history) in front of every extra configuration


def _counter_traffic(tag, lib):
    """HBM-side bytes per update launch of a configuration from the committed counter passes (profiles/traffic_configs.json, written by tools/prof_cfgs.sh
    from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes) -- or None when
    the counters were taken on ANOTHER build of the library than the one being timed (the file carries the build id)."""
    try:
        from bipymc_amd import _lib as L
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_configs.json")))
        if t.get("build_id") != L.build_id(lib):
            return None
        return t[tag]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def other_configs(device, budget_s=6.0):
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils import banana_rv, d100_gauss, mixture_nd
    np.random.seed(20261004)
    gauss, banana, mix8 = d100_gauss.Gauss_100D(rho=0.5, dim=DIM), banana_rv.Banana_2D(), mixture_nd.BimodeGauss_ND(8)
    # (name, algo, target, N, generations, bytes per update, kernel, engine kwargs)
    specs = [
        ("cfg3 DE-MC banana d=2 N=65536 snooker 0.1", L.ALGO_DEMC, banana, 65536, 400, 97.6,
         "phase_fused_kernel<0,3,1,2,1,2> (one lane per chain)", dict(p_snooker=0.1)),
        ("cfg5 one GPU's share: DREAM mixture d=8 N=32768 steady", L.ALGO_DREAM, mix8, 32768, 400, 592.0,
         "phase_fused_kernel<1,2,4,2,3,2> (4 lanes per chain)", dict(burnin_gen=0)),
        ("cfg5 whole on one GPU: DREAM mixture d=8 N=262144 steady", L.ALGO_DREAM, mix8, 262144, 200, 592.0,
         "phase_fused_kernel<1,2,4,2,3,2> (4 lanes per chain)", dict(burnin_gen=0)),
        ("cfg2 burn-in: DREAM gauss d=100 N=8192, CR adaptation on", L.ALGO_DREAM, gauss, 8192, 300, 10416.0,
         "phase_fused_kernel<1,1,64,2,3,3> (level 1 of the CR reduction inside) + cr_final_kernel", dict(burnin_gen=10 ** 6, n_cr_gen=5)),
        ("cfg5 burn-in: DREAM mixture d=8 N=262144, CR adaptation + outlier check every 50", L.ALGO_DREAM, mix8, 262144, 100,
         848.0, "phase_fused_kernel<1,2,4,2,3,4> (level 1 of the CR reduction inside) + cr_mid_kernel + cr_final_kernel + outlier kernels",
         dict(burnin_gen=10 ** 6, n_cr_gen=5, outlier_every=50)),
    ]
    out = []
    t_begin = time.perf_counter()
    for name, algo, tgt, N, gens, bpu, kernel, kw in specs:
        if time.perf_counter() - t_begin > budget_s:
            out.append(dict(config=name, skipped="time budget of the extra configurations used up"))
            continue
        tid, tp, d = tgt._bpm_target_spec()
        x0 = tgt.rvs(N)
        if isinstance(x0, tuple):
            x0 = np.stack(x0, axis=1)
        e = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, device=device, **kw)
        try:
            e.set_state(x0)
            e.reserve_history(gens + 48)
            e.begin_run()
            # the GPU kept busy right up to the timed generations, as for the headline (--preheat): creating a sampler and reserving its history leaves
            # the GPU idle for tens of milliseconds, and the first 8 ms block after that ran 8 % slower than the following ones (cfg5: 41.8 against 38.5 us)
            heat_kw = dict(kw, burnin_gen=0, outlier_every=0) if algo == L.ALGO_DREAM else dict(kw)
            heat = HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=7, device=device, keep_history=False, **heat_kw)
            heat.set_state(x0)
            heat.begin_run()
            th = time.perf_counter()
            while time.perf_counter() - th < OTHER_CONFIG_PREHEAT_S:
                heat.step(64)
                heat.synchronize()
            heat.close()
            e.step(30)
            e.synchronize()
            t0 = time.perf_counter()
            ev_ms, ev_n = e.step_timed(gens)
            el = time.perf_counter() - t0
            st = e.stats()
            ls = e.launch_stats()
        finally:
            e.close()
        value = N * gens / el
        gen_us_dev = (ev_ms * 1e3 / ev_n * 2.0) if ev_n > 0 else None     # two update launches per generation; with CR adaptation
        ach = N * bpu / ((gen_us_dev or el / gens * 1e6) * 1e-6) / 1e9      # the reduction dispatches sit inside the period
        out.append(dict(config=name, value=value, unit="chain-updates/s", n_chains=N, dim=d, steps=gens,
                        ms_per_step=el / gens * 1e3, start="exact draws of the target", preheat_s=OTHER_CONFIG_PREHEAT_S,
                        acceptance_fraction=st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"]),
                        packet_fence=ls["fence"],
                        roofline=dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                      traffic=(_counter_traffic("cfg3", e.lib) if name.startswith("cfg3") else
                                               (_counter_traffic("cfg5", e.lib) if name.startswith("cfg5 whole") else None)),
                                      kernel=kernel, bytes_per_unit=bpu, units_per_launch=N / 2.0,
                                      avg_launch_us=(ev_ms * 1e3 / ev_n) if ev_n > 0 else None, launches_timed=ev_n,
                                      note="achieved = n_chains x bytes_per_unit / device-timed generation period (two update "
                                           "launches back to back, reduction dispatches of burn-in included)")))
    return out


def connect_exchange(eng, dist, rank, world, want):
    """-> dict(mode=..., why=...).  Every decision is taken on all ranks together (all_gather_object of what each rank saw)."""
    def everyone(v, what="connect_exchange: all_gather_object"):
        _pend(what)
        box = [None] * world
        dist.all_gather_object(box, v)
        return box
    if want in ("replay", "rows", "dense"):
        eng.set_exchange(mode=want)
        return dict(mode=want, why="--exchange")
    err = None
    try:
        blob = eng.push_export()
    except Exception as e:                                             # noqa: BLE001 -- reported collectively
        blob, err = None, "export: %s" % e
    blobs = everyone((blob, err), "connect_exchange: all_gather_object of the ranks' bpm_push_export blobs")
    ok = all(b[0] is not None for b in blobs)
    if ok:
        try:
            eng.push_connect([b[0] for b in blobs])
        except Exception as e:                                         # noqa: BLE001
            ok, err = False, "connect: %s" % e
    seen = everyone((ok, err), "connect_exchange: all_gather_object after bpm_push_connect")
    ok = all(o[0] for o in seen)
    if ok:
        _pend("connect_exchange: dist.barrier before bpm_push_selftest")
        dist.barrier()
        try:
            _pend("connect_exchange: bpm_push_selftest (cross-rank waits inside)")
            mine = bool(eng.push_selftest())
        except Exception as e:                                         # noqa: BLE001
            mine, err = False, "self-test: %s" % e
        seen = everyone((mine, err), "connect_exchange: all_gather_object after bpm_push_selftest")
        ok = all(o[0] for o in seen)
    if ok:
        eng.set_exchange(mode="push")
        return dict(mode="push", why="every rank mapped every peer and the self-test passed", ranks_connected=world)
    why = "; ".join("rank %d: %s" % (i, o[1]) for i, o in enumerate(seen) if o[1]) or "self-test failed on some rank"
    if want == "push":
        raise SystemExit("bench.py: --exchange push but the push exchange is not available: " + why)
    if want == "push-or-nothing":                                      # (a world without an RCCL communicator: the caller creates another one)
        return dict(mode=None, why=why)
    if eng.exchange_stats()["push_connected"]:
        eng.set_exchange(mode="replay")
    return dict(mode="replay", why="push exchange not available (%s)" % why)


def validate_exchange(eng, dist, X0, make_single, candidates, gens=1000, fatal=True, wall_s=45.0, chunk=250):
    """Which exchange the timed run uses is decided by a RUN, not by what connected: up to `gens` generations with CR adaptation (so the per-update
    statistics travel too) from the same start, once on a single-rank sampler holding the whole population on this rank's own GPU (what
    the reference computes on one MPI rank, demc.py:63-151) and then under each candidate in order -- push with agent-scope fences (cheapest),
    push with system-scope fences (what the HSA memory model asks for between agents), accept bytes + replay through RCCL, the dense all-gather
    (the reference's own exchange, demc.py:93-94).  The first candidate that leaves EVERY rank's replica bit-identical to the single-rank run
    wins; a candidate that raises (a cross-rank wait that ran into its limit) is recorded and skipped.  Bounded by wall time as well as by
    generations: the runs advance in chunks of `chunk` generations, the single-rank run keeps the state's hash after every chunk, and a candidate
    stops (all ranks together) after the chunk in which the slowest rank passed wall_s seconds -- it is then compared at that generation.
    -> dict(mode=..., validation=[...])."""
    import hashlib

    def sha(x):
        return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16]
    world = dist.get_world_size()
    sizes = [min(chunk, gens - g0) for g0 in range(0, gens, chunk)]
    _pend("validate_exchange: single-rank run (local)")
    single = make_single()
    single.set_state(X0)
    single.begin_run()
    want = []                                   # hash of the single-rank state after every chunk
    t0 = time.perf_counter()
    for n in sizes:
        single.step(n)
        single.synchronize()
        want.append(sha(single.get_state()))
        if time.perf_counter() - t0 > wall_s and len(want) < len(sizes):
            break
    single.close()
    sizes = sizes[:len(want)]
    log = []
    push_dead = False
    for cand in candidates:
        if push_dead and cand.startswith("push"):
            log.append(dict(exchange=cand, ok=False, why="skipped: a push wait timed out under an earlier candidate"))
            continue
        mine, err, done_chunks = None, None, 0
        t0 = time.perf_counter()
        try:
            eng.set_exchange(mode=cand)
            eng.set_state(X0)
            eng.set_adapt_state(t_abs=0)
            eng.begin_run()
            _pend("validate_exchange[%s]: dist.barrier before the run" % cand)
            dist.barrier()
        except Exception as e:                                         # noqa: BLE001 -- decided collectively below
            err = "%s: %s" % (type(e).__name__, e)
        for n in sizes:
            if err is None:
                try:
                    _pend("validate_exchange[%s]: bpm_step + bpm_synchronize (cross-rank hand-overs inside)" % cand)
                    eng.step(n)
                    eng.synchronize()
                except Exception as e:                                 # noqa: BLE001
                    err = "%s: %s" % (type(e).__name__, e)
            # every rank enters this collective after every chunk, whatever happened to it: stop together (an error anywhere, or the wall clock)
            _pend("validate_exchange[%s]: all_gather_object after chunk %d" % (cand, done_chunks))
            box = [None] * world
            dist.all_gather_object(box, (err is None, time.perf_counter() - t0))
            if not all(b[0] for b in box):
                break
            done_chunks += 1
            if max(b[1] for b in box) > wall_s:
                break
        if err is None and done_chunks > 0:
            try:
                mine = sha(eng.get_state())
            except Exception as e:                                     # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
        _pend("validate_exchange[%s]: all_gather_object of the state hashes" % cand)
        box = [None] * world
        dist.all_gather_object(box, (mine, err))
        target = want[done_chunks - 1] if done_chunks > 0 else None
        ok = target is not None and all(b[0] == target for b in box)
        entry = dict(exchange=cand, ok=ok, generations=int(sum(sizes[:done_chunks])), ranks_equal_to_single_rank_run=sum(1 for b in box if target is not None and b[0] == target))
        if done_chunks < len(sizes) and ok:
            entry["stopped_by_wall_clock_s"] = wall_s
        errs = ["rank %d: %s" % (i, b[1]) for i, b in enumerate(box) if b[1]]
        if errs:
            entry["errors"] = errs[:4]
            if cand.startswith("push"):
                push_dead = True
                try:                                                    # (clears the time-out mark so that bpm_synchronize works again)
                    eng.set_exchange(mode="dense")
                except Exception:                                      # noqa: BLE001
                    pass
        log.append(entry)
        if ok:
            eng.set_exchange(mode=cand)
            eng.set_adapt_state(t_abs=0)
            return dict(mode="push" if cand.startswith("push") else cand, fence_scope={"push-agent": "agent", "push": "system"}.get(cand),
                        candidate=cand, candidates_left=[c for c in candidates[candidates.index(cand) + 1:] if not (push_dead and c.startswith("push"))],
                        validation=log)
    if not fatal:
        return dict(mode=None, validation=log)
    raise SystemExit("bench.py: no exchange reproduced the single-rank run on every rank: " + json.dumps(log))


def _everyone(dist, world, v, what="all_gather_object"):
    _pend(what)
    box = [None] * world
    dist.all_gather_object(box, v)
    return box


def _collectively(dist, world, fn):
    """fn() on every rank; -> (True, [results]) only when NO rank raised (every rank learns the others' outcome before anyone goes on into
    the next collective), else (False, ["rank r: error", ...])"""
    try:
        res, err = fn(), None
    except BaseException as e:                                            # noqa: BLE001 -- reported, never fatal: SystemExit of a helper included
        res, err = None, "%s: %s" % (type(e).__name__, e)
    box = _everyone(dist, world, (res, err))
    errs = ["rank %d: %s" % (i, b[1]) for i, b in enumerate(box) if b[1]]
    return (not errs), (errs if errs else [b[0] for b in box])


def _timed_generations(eng, dist, world, X0, n_chains, burn, gens, coll_max):
    """burn untimed generations (CR adaptation), then `gens` generations under the host clock between barriers; max over ranks.
    -> dict(value, ms_per_step, replicas_identical)"""
    import hashlib
    eng.set_state(X0)
    eng.set_adapt_state(t_abs=0)
    eng.begin_run()
    _pend("_timed_generations: dist.barrier before burn-in")
    dist.barrier()
    _pend("_timed_generations: bpm_step(burn-in) + bpm_synchronize")
    eng.step(burn)
    eng.synchronize()
    _pend("_timed_generations: dist.barrier before the timed generations")
    dist.barrier()
    t0 = time.perf_counter()
    _pend("_timed_generations: bpm_step + bpm_synchronize (exchange inside)")
    eng.step(gens)
    eng.synchronize()
    _pend("_timed_generations: dist.barrier / all_reduce(MAX) behind the timed generations")
    dist.barrier()
    el = coll_max(time.perf_counter() - t0)
    sha = hashlib.sha256(np.ascontiguousarray(eng.get_state()).tobytes()).hexdigest()[:16]
    shas = _everyone(dist, world, sha)
    return dict(value=n_chains * gens / el, ms_per_step=el / gens * 1e3, steps=gens, replicas_identical=len(set(shas)) == 1)


def exchange_alternatives(eng, dist, world, X0, n_chains, make_single, make_rccl_engine, headline, coll_max, rccl_ranks_torch, share_gpu=False,
                          gens=200, burn=BURNIN_GEN):
    """OUTSIDE the headline's timed region: the same workload under the exchanges the headline did NOT use,
    each >= `gens` steady-state generations behind `burn` untimed burn-in generations.  A failure of an alternative is recorded, never fatal.
      dense       the exchange north_star names and the reference has -- MPI_Allgather twice per generation (bipymc/demc.py:93-94,116-117) as one
                  in-place ncclAllGather of a rank block -- on a SECOND sampler created with the library's own RCCL communicator
      push-agent  the push exchange with agent-scope packet fences (4 us less per hand-over on one GPU; between GPUs outside the HSA memory model,
                  so never the headline): only when the arena self-test passed under that form, validated against a single-rank run first
    -> list of dicts(mode, value, ms_per_step, rccl_ranks, replicas_identical | error | skipped)"""
    out = []
    # ---- dense, through the library's own communicator
    entry = dict(mode="dense", collective="ncclAllGather of a rank block, in place, twice per generation (demc.py:93-94,116-117)",
                 rccl_ranks=None, rccl_ranks_of_the_torch_process_group=rccl_ranks_torch)
    if share_gpu:
        entry["skipped"] = "the ranks of this rehearsal share one GPU: RCCL refuses two ranks on one device (torch.distributed runs over gloo, rccl_ranks null)"
    elif headline == "dense":
        entry["skipped"] = "the headline itself ran under the dense exchange"
    else:
        box = {}

        def create():
            box["eng"] = make_rccl_engine()
            box["eng"].set_exchange(mode="dense")
            return True
        ok, res = _collectively(dist, world, create)
        if ok:
            e2 = box["eng"]
            ok, res = _collectively(dist, world, lambda: _timed_generations(e2, dist, world, X0, n_chains, burn, gens, coll_max))
            if ok:
                entry.update(res[0])
                entry["rccl_ranks"] = world                       # the communicator bpm_create made: ncclCommInitRank(world) succeeded on every rank
                entry["exchange_stats"] = e2.exchange_stats()
        if not ok:
            entry["error"] = res[:4]
        if box.get("eng") is not None:
            try:
                dist.barrier()
                box["eng"].close()
            except Exception as e:                                    # noqa: BLE001
                entry.setdefault("error", []).append("close: %s" % e)
    out.append(entry)
    # ---- push with agent-scope fences
    if headline == "push":
        entry = dict(mode="push-agent", fence_scope="agent", rccl_ranks=None,
                     note="agent-scope packet fences around the update kernels; the pushed rows are system-scope write-through stores either way.  Between "
                          "GPUs this is outside the HSA memory model: reported beside the headline (system scope: what the sampler classes select), never as it")
        xs = eng.exchange_stats()
        if not xs.get("arena_probe_agent", False):
            entry["skipped"] = "the arena self-test failed under agent-scope fences (bpm_push_selftest)"
        else:
            def run():
                got = validate_exchange(eng, dist, X0, make_single, ["push-agent"], gens=300, fatal=False)
                if not got.get("mode"):
                    raise RuntimeError("300 generations under agent-scope fences did not reproduce the single-rank run: %s" % json.dumps(got["validation"]))
                eng.set_exchange(mode="push-agent")
                return _timed_generations(eng, dist, world, X0, n_chains, burn, gens, coll_max)
            ok, res = _collectively(dist, world, run)
            if ok:
                entry.update(res[0])
                entry["validated_against_single_rank_run"] = True
            else:
                entry["error"] = res[:4]
            try:
                eng.set_exchange(mode="dense" if not ok else "push")      # (clears a time-out mark, then back to what the headline used)
                eng.set_exchange(mode="push")
            except Exception:                                         # noqa: BLE001
                pass
        out.append(entry)
    return out


def host_callback_config(device, budget_s=4.5):
    """f1 under the driver's clock: cfg2's shape -- DREAM, 100-D Gaussian, 8192 chains -- with the likelihood
    as a CALLBACK of the caller, the path every user-written likelihood takes (samplers.py:36-43; examples/ex_para_fit.py:39-72 calls it row by row):
      (a) a vectorised NumPy ln_like on the host: every half generation the proposals come back in `chunks` pieces into pinned staging
          (bpm_propose_begin / _chunk), the DMA of piece k + 1 under the evaluation of piece k, the values go in piece by piece (bpm_commit_chunk / _end);
          beside it the one-piece form of rounds 1-4 (bpm_propose / bpm_commit through caller-owned buffers);
      (b) the same likelihood as a torch function ON THE DEVICE (bpm_propose_device / bpm_commit_device: nothing crosses PCIe).
    -> list of config entries"""
    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    N, d = CHAINS_PER_GPU, DIM
    sig = np.sqrt(np.arange(d) + 1.0)
    rho = 0.5
    c0 = -0.5 * (d * np.log(2 * np.pi) + 2 * np.sum(np.log(sig)) + (d - 1) * np.log(1 - rho) + np.log(1 + (d - 1) * rho))
    a, b = 1.0 / (1 - rho), rho / ((1 - rho) * (1 + (d - 1) * rho))
    isig = 1.0 / sig

    def ln_like(X):                                                  # (n, d) -> (n,): the equicorrelated Gaussian in O(n d)
        z = X * isig
        s1 = z.sum(axis=1)
        return c0 - 0.5 * (a * np.einsum("ij,ij->i", z, z) - b * s1 * s1)
    rs = np.random.RandomState(1234)
    X0 = sig * (np.sqrt(0.5) * rs.standard_normal((N, 1)) + np.sqrt(0.5) * rs.standard_normal((N, d)))
    half = N // 2
    out = []

    def run(name, one_generation, init, seconds, extra, max_calls=5000):
        e = HipEngine(algo=L.ALGO_DREAM, n_chains=N, dim=d, target_id=L.TARGET_HOST_CALLBACK, target_params=None, seed=42, device=device,
                      del_pairs=DEL_PAIRS, burnin_gen=0, n_cr_gen=N_CR_GEN, n_cr=3)
        try:
            e.set_state(X0)
            init(e)
            e.reserve_history(6000)
            e.begin_run()
            t_py = [0.0]
            for _ in range(3):                                        # warm-up
                one_generation(e, t_py)
            gens, t_py[0], t0 = 0, 0.0, time.perf_counter()
            while time.perf_counter() - t0 < seconds and gens < max_calls:
                one_generation(e, t_py)
                gens += 1
            e.synchronize()
            el = time.perf_counter() - t0
            st = e.stats()
        finally:
            e.close()
        ent = dict(config=name, value=N * gens / el, unit="chain-updates/s", n_chains=N, dim=d, steps=gens, ms_per_step=el / gens * 1e3,
                   start="exact draws of the target",
                   acceptance_fraction=st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"]),
                   share_of_time_in_the_callback=t_py[0] / el,
                   note="the host clock includes the callback; not a roofline configuration")
        ent.update(extra)
        out.append(ent)

    def gen_one_piece(e, t_py):
        for _h in range(2):
            props, _ids = e.propose()
            tp0 = time.perf_counter()
            ll = ln_like(props)
            t_py[0] += time.perf_counter() - tp0
            e.commit(ll)

    def gen_chunked(chunks):
        def g(e, t_py):
            for _h in range(2):
                for k, rows, _ids in e.propose_chunks(chunks):
                    tp0 = time.perf_counter()
                    ll = ln_like(rows)
                    t_py[0] += time.perf_counter() - tp0
                    e.commit_chunk(k, ll)
                e.commit_end()
        return g
    def gen_threaded(chunks, threads):
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=threads)

        def work(e, k):
            rows, _ids = e.propose_chunk(k)                           # (waits for the DMA of piece k only)
            t = time.perf_counter()
            ll = ln_like(rows)
            return ll, time.perf_counter() - t

        def g(e, t_py):
            for _h in range(2):
                e.propose_begin(chunks)
                futs = [pool.submit(work, e, k) for k in range(chunks)]
                for k, f in enumerate(futs):
                    ll, dt = f.result()
                    t_py[0] += dt / threads                           # (thread-seconds / threads: the callback's share of the wall clock)
                    e.commit_chunk(k, ll)
                e.commit_end()
        return g
    d2h = N * 4 + half * d * 8                                        # work-item ids + the proposals of the half's chains
    h2d = half * 8                                                    # their ln-likes
    pcie = dict(d2h_bytes_per_half_generation=d2h, h2d_bytes_per_half_generation=h2d)
    host_init = lambda e: e.set_loglike(ln_like(X0))                  # noqa: E731
    CH, TH = 4, 2
    try:
        run("cfg2 shape with a host-callback ln_like_fn (vectorised NumPy), read-back in %d overlapped pieces evaluated by %d host threads "
            "(DreamMpi(..., vectorized=True, callback_threads=%d)): DREAM gauss d=100 N=8192 steady" % (TH, TH, TH), gen_threaded(TH, TH), host_init, budget_s * 0.2,
            dict(pcie=dict(pcie, staging="pinned staging of the library, %d pieces per half generation" % TH), host_threads=TH))
    except Exception as ex:                                           # noqa: BLE001
        out.append(dict(config="cfg2 shape with a host-callback ln_like_fn, overlapped read-back, threaded evaluation", error=str(ex)))
    try:
        run("cfg2 shape with a host-callback ln_like_fn (vectorised NumPy), read-back in %d overlapped pieces: DREAM gauss d=100 N=8192 steady, "
            "bpm_propose_begin / _chunk + bpm_commit_chunk / _end, ONE host thread" % CH, gen_chunked(CH), host_init, budget_s * 0.25,
            dict(pcie=dict(pcie, staging="pinned staging of the library, %d pieces per half generation, the DMA of piece k + 1 under the evaluation of piece k" % CH)))
    except Exception as ex:                                           # noqa: BLE001
        out.append(dict(config="cfg2 shape with a host-callback ln_like_fn, overlapped read-back", error=str(ex)))
    try:
        run("cfg2 shape with a host-callback ln_like_fn (vectorised NumPy), ONE call per half generation through caller-owned buffers (what DreamMpi does by "
            "default; bpm_propose / bpm_commit, the read-back in 4 pieces under the library's own compaction copy)",
            gen_one_piece, host_init, budget_s * 0.3, dict(pcie=dict(pcie, staging="pinned staging, the DMA of piece k + 1 under the compaction copy of piece k")))
    except Exception as ex:                                           # noqa: BLE001
        out.append(dict(config="cfg2 shape with a host-callback ln_like_fn, one piece", error=str(ex)))
    try:
        import torch
        dev = "cuda:%d" % device
        isig_t = torch.tensor(isig, dtype=torch.float64, device=dev)

        def ln_like_dev(rows):
            X = torch.as_tensor(rows, device=dev)                     # the library's buffer, no copy
            z = X * isig_t
            s1 = z.sum(dim=1)
            return c0 - 0.5 * (a * (z * z).sum(dim=1) - b * s1 * s1)

        def gen_dev(e, t_py):
            for _h in range(2):
                rows = e.propose_device()
                tp0 = time.perf_counter()
                ll = ln_like_dev(rows)
                t_py[0] += time.perf_counter() - tp0                  # (launch time of the torch kernels: they run asynchronously)
                e.commit_device(ll)
        run("cfg2 shape with a DEVICE-RESIDENT callback (the same likelihood as a torch function on the GPU, vectorized=\"device\"): "
            "bpm_propose_device / bpm_commit_device, no PCIe", gen_dev, lambda e: e.set_loglike_device(ln_like_dev(e.state_device())), budget_s * 0.25,
            dict(pcie=dict(d2h_bytes_per_half_generation=N * 4, h2d_bytes_per_half_generation=0, staging="none: proposals and ln-likes stay in device memory")))
    except Exception as ex:                                           # noqa: BLE001
        out.append(dict(config="cfg2 shape with a device-resident (torch) callback", error=str(ex)))
    # (c) the same likelihood as a few lines of HIP source, compiled with hiprtc into ONE kernel between the library's proposal and commit kernels
    # (bpm_set_device_likelihood; bipymc_amd.HipLikelihood): bpm_step drives the sampler, no host code inside a generation
    try:
        src = ("__device__ double ln_like(const double* x, int d, const double* p) {\n"
               "    double s1 = 0.0, s2 = 0.0;\n"
               "    for (int j = 0; j < d; ++j) { const double z = x[j] * p[3 + j]; s1 += z; s2 += z * z; }\n"
               "    return p[0] - 0.5 * (p[1] * s2 - p[2] * s1 * s1);\n}\n")
        pblock = np.concatenate([[c0, a, b], isig])
        STEP = 50

        form = {}

        def gen_src(e, t_py):                                         # (`run` counts generations: STEP of them per call here)
            if not form:
                fused, why = e.device_likelihood_info()
                form.update(update_kernel_compiled_around_the_likelihood=bool(fused), why_not=why or None)
            e.step(STEP)
        # the per-coordinate form of the same likelihood (HipLikelihood(..., terms=2)): every lane of a chain adds the terms of its own coordinates
        src_terms = ("#define BPM_LN_LIKE_TERMS 2\n"
                     "__device__ void ln_like_terms(double xj, int j, int d, const double* p, double* acc) { const double z = xj * p[3 + j]; acc[0] += z; acc[1] += z * z; }\n"
                     "__device__ double ln_like_finish(const double* acc, int d, const double* p) { return p[0] - 0.5 * (p[1] * acc[1] - p[2] * acc[0] * acc[0]); }\n")
        for the_src, what in ((src, "plain form: ln_like(x, d, p) runs on one lane per chain"),
                              (src_terms, "per-coordinate form: ln_like_terms / ln_like_finish, every lane adds its own coordinates' terms")):
            form.clear()
            n0 = len(out)
            run("cfg2 shape with ln_like_fn given as HIP SOURCE (bipymc_amd.HipLikelihood, %s; the update kernel compiled at construction around the caller's "
                "function, bpm_step drives the sampler): DREAM gauss d=100 N=8192 steady" % what, gen_src,
                lambda e, the_src=the_src: e.set_device_likelihood(the_src, pblock), budget_s * 0.12,
                dict(pcie=dict(d2h_bytes_per_half_generation=0, h2d_bytes_per_half_generation=0, staging="none"), generations_per_step_call=STEP),
                max_calls=80)                                         # (3 + 80 calls of 50 generations: inside the 6000 reserved history rows)
            if len(out) > n0:                                         # `run` timed step calls of STEP generations each
                ent = out[-1]
                ent["steps"] *= STEP
                ent["value"] *= STEP
                ent["ms_per_step"] /= STEP
                ent["share_of_time_in_the_callback"] = 0.0
                ent["form"] = dict(form, note="true: ONE launch per half generation (the library's update kernel compiled at run time with the caller's function "
                                              "as its target); false: proposal / likelihood / commit kernels")
    except Exception as ex:                                           # noqa: BLE001
        out.append(dict(config="cfg2 shape with ln_like_fn given as HIP source", error=str(ex)))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The posterior gate (BASELINE.json north_star: "posterior moments within 1 % of reference"): EVERY coordinate's variance
# within 1 % of the analytic value and EVERY coordinate's mean within 0.01 sigma, each with a batch-means Monte-Carlo standard error beside it.
# A sampler with running_moments keeps two 2 d-double sums per generation instead of a history (10^5 generations of cfg2 would be 660 GB), so the
# gate costs a second of GPU time; bpm_reduce_moments answers for any burn-in of whole generations, batches are differences of two such answers.
# ---------------------------------------------------------------------------------------------------------------------
GATE_VAR_TOL = 0.01           # |var_j / var_j(analytic) - 1| for every coordinate j
GATE_MEAN_TOL = 0.01          # |mean_j - mean_j(analytic)| / sigma_j for every coordinate j
GATE_BATCHES = 20


def batch_moment_gate(eng, n_chains, g_lo, g_hi, true_mean, true_var, batches=GATE_BATCHES, reduce_all=None):
    """Per-coordinate posterior moments of the generations [g_lo, g_hi) of a single-rank sampler (rows g * n_chains + i of the interleaved super chain,
    demc.py:260-270; param_est's mean / std(ddof=0), demc.py:235-248) and their batch-means standard errors over `batches` consecutive blocks of
    generations.  reduce_all: sums the per-rank (count | s1 | s2) over the ranks of a world (None: one rank).  -> dict for the JSON line."""
    true_mean, true_var = np.asarray(true_mean, dtype=np.float64), np.asarray(true_var, dtype=np.float64)
    edges = np.unique(np.linspace(g_lo, g_hi, batches + 1).astype(np.int64))
    B = len(edges) - 1
    cum = []
    for g in edges:                                   # sums over rows >= g * n_chains
        cnt, s1, s2, sh = eng.reduce_moments(int(g) * n_chains)
        if reduce_all is not None:
            cnt, s1, s2 = reduce_all(cnt, s1, s2)
        cum.append((float(cnt), s1.copy(), s2.copy(), sh.copy()))
    sh = cum[0][3]
    n = np.array([cum[k][0] - cum[k + 1][0] for k in range(B)])
    S1 = np.array([cum[k][1] - cum[k + 1][1] for k in range(B)])
    S2 = np.array([cum[k][2] - cum[k + 1][2] for k in range(B)])
    n_tot = n.sum()
    mu_s = S1.sum(axis=0) / n_tot                                   # mean in shifted coordinates (x - sh)
    mean = sh + mu_s
    var = S2.sum(axis=0) / n_tot - mu_s ** 2                        # std(ddof=0)^2 of param_est
    mean_b = S1 / n[:, None]                                        # batch means (shifted)
    var_b = S2 / n[:, None] - 2.0 * mu_s * mean_b + mu_s ** 2       # batch second moments about the OVERALL mean
    sig = np.sqrt(true_var)
    mcse_mean = mean_b.std(axis=0, ddof=1) / np.sqrt(B) / sig if B > 1 else np.full_like(sig, np.nan)
    mcse_vr = (var_b / true_var).std(axis=0, ddof=1) / np.sqrt(B) if B > 1 else np.full_like(sig, np.nan)
    vr = var / true_var
    mz = (mean - true_mean) / sig
    jv, jm = int(np.argmax(np.abs(vr - 1.0))), int(np.argmax(np.abs(mz)))
    ok = bool(np.all(np.abs(vr - 1.0) <= GATE_VAR_TOL) and np.all(np.abs(mz) <= GATE_MEAN_TOL))
    return dict(generations=int(g_hi - g_lo), first_generation=int(g_lo), rows=int(n_tot), batches=int(B),
                var_ratio_mean=float(vr.mean()), var_ratio_min=float(vr.min()), var_ratio_max=float(vr.max()),
                max_abs_mean_over_sigma=float(np.max(np.abs(mz))),
                mcse_var_ratio_max=float(np.nanmax(mcse_vr)), mcse_mean_over_sigma_max=float(np.nanmax(mcse_mean)),
                worst_variance=dict(coordinate=jv, var_ratio=float(vr[jv]), mcse=float(mcse_vr[jv])),
                worst_mean=dict(coordinate=jm, mean_over_sigma=float(mz[jm]), mcse=float(mcse_mean[jm])),
                gate=dict(var_ratio_every_coordinate_within=GATE_VAR_TOL, abs_mean_every_coordinate_within_sigma=GATE_MEAN_TOL,
                          standard_errors="batch means over %d consecutive blocks of generations" % B),
                gate_pass=ok)


POSTERIOR_GATE_GENS = 40000      # post-burn-in generations of the per-coordinate gate at cfg2 (tools/posterior_gate_sweep.py, profiles/r05_posterior_gate_sweep.txt)
REFERENCE_START_TRANSIENT = 3000 # generations dropped from the reference's start (theta_0 = 0, varepsilon = 1e-6): the population needs ~2250 generations to
                                 # find the stationary scale (profiles/r04_convergence_from_reference_start.txt)


def _gate_engine(device, algo, tgt, N, **kw):
    from bipymc_amd.engine import HipEngine
    tid, tp, d = tgt._bpm_target_spec()
    return HipEngine(algo=algo, n_chains=N, dim=d, target_id=tid, target_params=tp, seed=42, device=device, keep_history=False, running_moments=True, **kw), d


def posterior_gate(device, gens=POSTERIOR_GATE_GENS, start="exact"):
    """The per-coordinate posterior gate of config 2 (DREAM, 100-D Gaussian, N = 8192; analytic moments mean 0, var_j = j + 1, d100_gauss.py:14-35) on a
    sampler that keeps per-generation population sums instead of a history.  start = "exact": exact draws of the target (the stationary regime from the
    first generation; BURNIN_GEN generations with CR adaptation are dropped); "reference": the reference's own start, theta_0 = 0 with varepsilon = 1e-6
    (chain.py:25-27, tests/test_100dgauss.py:105-110), REFERENCE_START_TRANSIENT generations dropped.  Untimed."""
    from bipymc_amd import _lib as L
    from bipymc_amd.utils.d100_gauss import Gauss_100D
    tgt = Gauss_100D(rho=0.5, dim=DIM)
    e, d = _gate_engine(device, L.ALGO_DREAM, tgt, CHAINS_PER_GPU, del_pairs=DEL_PAIRS, burnin_gen=BURNIN_GEN, n_cr_gen=N_CR_GEN, n_cr=3)
    t0 = time.perf_counter()
    try:
        if start == "exact":
            rs = np.random.RandomState(4321)
            e.set_state(np.sqrt(np.arange(DIM) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((CHAINS_PER_GPU, 1)) + np.sqrt(0.5) * rs.standard_normal((CHAINS_PER_GPU, DIM))))
            drop = BURNIN_GEN
        else:
            e.init_chains(np.zeros(DIM), 1e-6)
            drop = REFERENCE_START_TRANSIENT
        e.begin_run()
        e.step(drop + gens)
        e.synchronize()
        g = batch_moment_gate(e, CHAINS_PER_GPU, 1 + drop, 1 + drop + gens, np.zeros(DIM), np.arange(DIM) + 1.0)
        st = e.stats()
    finally:
        e.close()
    g.update(start=("exact draws of the target" if start == "exact" else
                    "the reference's: theta_0 = 0, varepsilon = 1e-6 (every chain within 1e-3 of the origin)"),
             generations_dropped=int(drop), seconds=round(time.perf_counter() - t0, 2),
             acceptance_fraction=st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"]), p_cr=[float(v) for v in st["p_cr"]],
             sampler="DREAM N=%d d=%d, keep_history=0, running_moments=1 (two 2 d-double population sums per generation)" % (CHAINS_PER_GPU, DIM))
    return g


def posterior_gates_other_configs(device):
    """The same gate for cfg3 (DE-MC, banana, N = 65536, snooker 0.1: E = (0, b (1 + a^2)), Var = (a^2, 1 / a^2 + 2 b^2), banana_rv.py:11-40) and for the
    8-D pairwise mixture of cfg5 (one GPU's share, N = 32768, and the whole population, N = 262144: per axis mean 1.5, var 0.8125 = 0.0625 within a mode +
    the spread of the two modes, dblgauss_rv.py:11-32 -- moves between the modes included), from exact draws of the targets.  Untimed."""
    from bipymc_amd import _lib as L
    from bipymc_amd.utils import banana_rv, mixture_nd
    out = []
    np.random.seed(20261005)
    a_, b_ = 1.15, 0.5
    specs = [("cfg3 DE-MC banana d=2 N=65536 snooker 0.1", L.ALGO_DEMC, banana_rv.Banana_2D(), 65536, 0, 30000,
              [0.0, b_ * (1 + a_ * a_)], [a_ * a_, 1.0 / (a_ * a_) + 2 * b_ * b_], dict(p_snooker=0.1)),
             ("cfg5 one GPU's share: DREAM mixture d=8 N=32768", L.ALGO_DREAM, mixture_nd.BimodeGauss_ND(8), 32768, 300, 30000,
              np.full(8, 1.5), np.full(8, 0.8125), dict(burnin_gen=300, n_cr_gen=50)),
             ("cfg5 whole: DREAM mixture d=8 N=262144", L.ALGO_DREAM, mixture_nd.BimodeGauss_ND(8), 262144, 300, 10000,
              np.full(8, 1.5), np.full(8, 0.8125), dict(burnin_gen=300, n_cr_gen=50))]
    for name, algo, tgt, N, drop, gens, tm, tv, kw in specs:
        try:
            e, d = _gate_engine(device, algo, tgt, N, **kw)
            t0 = time.perf_counter()
            try:
                # (the mixture's start is stratified: exactly a quarter of the chains in the first mode -- the overall variance 0.0625 + 4 w (1 - w)
                # moves by 0.6 % per standard deviation of a RANDOM occupancy at N = 32768, and the modes exchange chains only slowly)
                x0 = tgt.rvs(N, stratified=True) if algo == L.ALGO_DREAM else tgt.rvs(N)
                if isinstance(x0, tuple):
                    x0 = np.stack(x0, axis=1)
                e.set_state(x0)
                e.begin_run()
                e.step(drop + gens)
                e.synchronize()
                g = batch_moment_gate(e, N, 1 + drop, 1 + drop + gens, tm, tv)
                st = e.stats()
            finally:
                e.close()
            g.update(config=name, start="exact draws of the target" + (", mode occupancy stratified (exactly 1/4 : 3/4)" if algo == L.ALGO_DREAM else ""), generations_dropped=drop, seconds=round(time.perf_counter() - t0, 2),
                     acceptance_fraction=st["local_n_accepted"] / float(st["local_n_accepted"] + st["local_n_rejected"]))
        except Exception as ex:                                        # noqa: BLE001
            g = dict(config=name, error=str(ex))
        out.append(g)
    return out


def reference_at_the_headline_configuration():
    """What the GENUINE reference does at bench.py's own workload (tests/golden/e2e_anchor_cfg2_headline.json, recorded by oracle/gen_anchor_cfg2.py: DreamMpi,
    100-D Gaussian, 8192 chains from exact draws, n_cr_gen = 50, burnin_gen = 200): the numbers that belong beside posterior.p_cr / acceptance_fraction."""
    path = os.path.join(ROOT, "tests", "golden", "e2e_anchor_cfg2_headline.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return None
    runs = doc["runs"]
    return dict(config=doc["config"],
                acceptance_with_uniform_p_cr_generations_1_to_50=[float(np.mean([t["window_acceptance"] for t in r["trajectory"] if t["generation"] <= 50])) for r in runs],
                p_cr_after_burnin=[[float("%.3g" % v) for v in r["p_cr_final"]] for r in runs],
                acceptance_after_burnin=[r["acceptance_after_burnin"] for r in runs],
                var_ratio_pooled_after_burnin=[r["var_ratio_pooled_after_burnin"] for r in runs],
                note="the reference's CR adaptation collapses to a ONE-HOT p_cr within a generation of its start at this configuration (the zero-variance clamp of "
                     "dream.py:129 meets the per-update re-estimation of dream.py:134-140) and its acceptance is that of the surviving CR value; this build "
                     "re-estimates p_cr once per generation from all chains and stays mixed (posterior.p_cr).  At the reference's own p_cr -- uniform or one-hot -- "
                     "the device accepts the same fraction (tests/test_gpu_api.py::test_headline_configuration_at_the_references_own_p_cr)")


def reference_scenario_on_the_device(device):
    """The reference's OWN d = 100 scenario (tests/test_100dgauss.py:100-110: DreamMpi n_chains = 100, n_cr_gen = 50, burnin_gen = 2000, run_mcmc(500000),
    n_burn = 200000; DeMcMpi n_chains = 200) through the drop-in classes on the device, beside the family the genuine reference produced under
    three np.random seeds (tests/golden/e2e_anchor_gauss100_*.json, recorded by oracle/gen_anchor_families.py).  The asserted form is
    tests/test_gpu_api.py::test_reference_families_hold_the_device."""
    from bipymc_amd import DeMcMpi, DreamMpi
    from bipymc_amd.utils.d100_gauss import Gauss_100D
    out = []
    for name, mk in (("gauss100_dream", lambda g: DreamMpi(g.ln_like, np.zeros(100), n_chains=100, n_cr_gen=50, burnin_gen=2000, seed=42, device=device)),
                     ("gauss100_demc", lambda g: DeMcMpi(g.ln_like, np.zeros(100), n_chains=200, seed=42, device=device))):
        path = os.path.join(ROOT, "tests", "golden", "e2e_anchor_%s.json" % name)
        try:
            fam = json.load(open(path))["family"]
            g = Gauss_100D()
            s = mk(g)
            s.run_mcmc(500000)
            mean, std, _ = s.param_est(n_burn=200000)
            vr = std ** 2 / (np.arange(100) + 1.0)
            mine = dict(acceptance_fraction=float(s.acceptance_fraction), var_ratio_pooled=float(vr.mean()), var_ratio_min=float(vr.min()), var_ratio_max=float(vr.max()))
            if hasattr(s, "p_cr"):
                mine["p_cr"] = [float(v) for v in np.asarray(s.p_cr)]
            ref = {k: fam[k] for k in ("acceptance_fraction", "var_ratio_pooled", "var_ratio_min", "var_ratio_max", "p_cr") if k in fam}
            out.append(dict(scenario=name, config=json.load(open(path))["config"], device=mine, reference_family=ref))
        except Exception as ex:                                        # noqa: BLE001
            out.append(dict(scenario=name, error=str(ex)))
    return out


def main(argv=None):
    global CHAINS_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-moments", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--burnin-gens", type=int, default=BURNIN_GEN,
                    help="burn-in generations (CR adaptation) in front of the steady state (default %d).  0 for the rocprofv3 kernel trace: once the process has run the "
                         "burn-in kernels the profiler shows a mode of slow steady-state launches that un-profiled runs do not have "
                         "(profiles/r05_rocprof_torch_artefact.txt)" % BURNIN_GEN)
    ap.add_argument("--no-torch", action="store_true",
                    help="N = 1 only: keep PyTorch out of the process (synchronisation through the library alone, no measured copy rate, no torch callback entry). "
                         "For the rocprofv3 kernel trace: under the profiler a process on the torch wheel's HIP runtime shows a second mode of slow launches "
                         "that un-profiled runs do not have (profiles/r05_rocprof_torch_artefact.txt)")
    ap.add_argument("--no-exchange-alternatives", action="store_true", help="N > 1: skip the untimed runs under the exchanges the headline did not use")
    ap.add_argument("--preheat", type=float, default=0.5, help="seconds of untimed steady-state generations before burn-in (0: none)")
    ap.add_argument("--chains-per-gpu", type=int, default=CHAINS_PER_GPU, help="experiments only; the default is the BASELINE workload")
    ap.add_argument("--exchange", default=None, help="N > 1: push (default) | replay | rows | dense")
    ap.add_argument("--share-gpu", action="store_true", help="REHEARSAL ONLY (never a benchmark number): the N ranks share GPU 0 (push exchange "
                    "between processes, no RCCL: it refuses two ranks on one device; torch.distributed over gloo) -- runs every line of the N > 1 path on a one-GPU box")
    args = ap.parse_args(argv)
    globals()["BURNIN_GEN"] = max(0, int(args.burnin_gens))      # (every use below reads the module attribute)

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the parent of the N ranks.  Nothing of torch.cuda / HIP has been touched.
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv), n_visible=args.gpus if args.share_gpu else None)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d in the environment\n" % (args.gpus, world))
        return 2

    # stdout carries ONE JSON line and nothing else: libraries that print to file descriptor 1 (RCCL's version banner at communicator
    # creation) are sent to stderr for the whole run; the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def write_line(text):
        json_out.write(text + "\n")
        json_out.flush()
    # the rank's watchdog: every stage below has a wall-clock limit, the run an absolute deadline (the launcher's, minus a margin, for its children)
    global WATCH
    watch = WATCH = StageWatch(rank, world, write_line, steps=args.steps, warmup=args.warmup,
                               deadline_at=float(os.environ["BENCH_DEADLINE_AT"]) if os.environ.get("BENCH_DEADLINE_AT") else None)
    watch.enter("init")

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (before anything initialises HSA: dmabuf IPC, what hipIpcGetMemHandle and RCCL need on this pool)
    if args.no_torch:
        if world > 1 or os.environ.get("BPM_FORCE_DIST"):
            sys.stderr.write("bench.py: --no-torch is for N = 1 (the ranks of a world meet through torch.distributed)\n")
            return 2
        torch = None
        from bipymc_amd.demc import _visible_device_count
        ndev = _visible_device_count()
    else:
        import torch                  # first: the HIP runtime is then shared with libbipymc_hip.so
        ndev = torch.cuda.device_count()
    if ndev <= 0:
        sys.stderr.write("bench.py: no GPU visible (this benchmark has no CPU path)\n")
        return 2
    if args.share_gpu:
        local_rank = 0
    if world > 1 and local_rank >= ndev and ndev > 1:
        sys.stderr.write("bench.py: local rank %d but only %d visible GPU(s): one process per GPU\n" % (local_rank, ndev))
        return 2
    if local_rank >= ndev:            # launcher exposed one device per process
        local_rank = 0
    dist = None
    # BPM_FORCE_DIST=1 takes the multi-process path with a single rank (process group + one-rank RCCL
    # communicator): the only way to rehearse it on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("BPM_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        watch.pending("torch.distributed.init_process_group")
        if args.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if args.share_gpu else "cuda"

    from bipymc_amd import _lib as L
    from bipymc_amd.engine import HipEngine
    from bipymc_amd.utils.d100_gauss import Gauss_100D

    CHAINS_PER_GPU = args.chains_per_gpu
    n_chains = CHAINS_PER_GPU * world
    target = Gauss_100D(rho=0.5, dim=DIM)
    tid, tparams, _ = target._bpm_target_spec()

    def make_engine(uid):
        return HipEngine(algo=L.ALGO_DREAM, n_chains=n_chains, dim=DIM, target_id=tid, target_params=tparams, seed=42,
                         device=local_rank, rank=rank, world_size=world, nccl_uid=uid,
                         del_pairs=DEL_PAIRS, burnin_gen=BURNIN_GEN, n_cr_gen=N_CR_GEN, n_cr=3)

    def make_single():
        return HipEngine(algo=L.ALGO_DREAM, n_chains=n_chains, dim=DIM, target_id=tid, target_params=tparams, seed=42, device=local_rank,
                         del_pairs=DEL_PAIRS, burnin_gen=BURNIN_GEN, n_cr_gen=N_CR_GEN, n_cr=3, keep_history=False)

    def rccl_uid():
        box = [HipEngine.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    # Synthetic start: exact draws of the target (x_i = sigma_i (sqrt(rho) g + sqrt(1-rho) e_i)), so the
    # timed region is the stationary regime and the moment gate below tests invariance.  (From the
    # reference's default start -- theta_0 = 0 + 1e-3 jitter, SURVEY 8(d) -- or an independent over-dispersed one the
    # population needs ~1000 generations to find the correlated scale: tools/convergence_check.py.)
    rs = np.random.RandomState(1234)
    X0 = np.sqrt(np.arange(DIM) + 1.0) * (np.sqrt(0.5) * rs.standard_normal((n_chains, 1))
                                          + np.sqrt(0.5) * rs.standard_normal((n_chains, DIM)))
    total_gens = BURNIN_GEN + max(args.warmup + args.steps + 32, POSTERIOR_MIN_GENS) + 64
    # ---- N > 1: how the ranks exchange state where the reference calls comm.Allgather (demc.py:93-94,116-117).  First a world WITHOUT an RCCL
    # communicator of the library's own: the push exchange (owners store accepted rows straight into the peers' replicas through IPC-mapped
    # buffers; include/bipymc_hip.h) needs none, and nothing of RCCL can then stand between the run and its number.  It is taken only when every
    # rank mapped every peer, the connection self-test passed everywhere AND 1000 generations reproduce a single-rank run bit for bit
    # (validate_exchange).  Otherwise the sampler is created again with a communicator and the RCCL exchanges are tried the same way.
    # --exchange restricts the candidates (and then failing is fatal).  Ranks sharing one GPU (--share-gpu) have no RCCL at all.
    exchange_info = None
    eng = None
    rccl_comm_ranks = None                            # ranks of the RCCL communicator the LIBRARY created (ncclCommInitRank inside bpm_create), if any
    watch.enter("create")
    if not use_dist:
        eng = make_engine(None)
    elif world == 1:                                   # BPM_FORCE_DIST=1: the one-rank rehearsal of the RCCL path
        watch.pending("bpm_create with a one-rank RCCL communicator")
        eng = make_engine(rccl_uid())
        rccl_comm_ranks = 1
        watch.enter("connect")
        exchange_info = connect_exchange(eng, dist, rank, world, args.exchange)
    else:
        tried = []
        if args.exchange in (None, "push"):
            eng = make_engine(HipEngine.push_uid())
            watch.enter("connect")
            exchange_info = connect_exchange(eng, dist, rank, world, "push-or-nothing")
            if exchange_info["mode"] == "push":
                watch.enter("validation")
                eng.reserve_history(1 + total_gens)
                # the headline runs under what the sampler classes select (DeMcMpi._connect_exchange: set_exchange("push") = system-scope packet
                # fences, what the HSA memory model asks for between agents); the agent-scope form is timed beside it as an alternative
                got = validate_exchange(eng, dist, X0, make_single, ["push"], gens=min(1000, total_gens), fatal=False)
                tried += got["validation"]
                if got.get("mode"):
                    exchange_info.update(got)
                else:
                    exchange_info = dict(mode=None, why="no push candidate reproduced the single-rank run")
            if exchange_info["mode"] != "push":
                tried.append(dict(exchange="push", ok=False, why=exchange_info.get("why")))
                watch.pending("dist.barrier before closing the push world's samplers")
                dist.barrier()                         # (nobody unmaps while a peer may still be inside its last call)
                eng.close()
                eng = None
                if args.exchange == "push" or args.share_gpu:
                    raise SystemExit("bench.py: the push exchange is not available: " + json.dumps(tried))
        if eng is None:
            watch.enter("create")
            watch.pending("bpm_create: ncclCommInitRank over %d ranks (RCCL's first contact)" % world)
            eng = make_engine(rccl_uid())
            rccl_comm_ranks = world
            watch.enter("validation")
            eng.reserve_history(1 + total_gens)
            cands = [args.exchange] if args.exchange in ("replay", "rows", "dense") else ["replay", "dense"]
            eng.set_exchange(mode=cands[0])
            exchange_info = dict(why="--exchange" if args.exchange in cands else "push exchange not available or not validated")
            exchange_info.update(validate_exchange(eng, dist, X0, make_single, cands, gens=min(1000, total_gens)))
            exchange_info["validation"] = tried + exchange_info["validation"]

    def fence(what="fence"):
        watch.pending(what + ": bpm_synchronize")
        eng.synchronize()
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            watch.pending(what + ": dist.barrier")
            dist.barrier()

    # The measurement below runs once.  At N > 1 it runs again under the next (more conservative) exchange candidate if the ranks' replicas
    # are not bit-identical at its end: a run whose replicas diverged is not a measurement (validate_exchange has passed 1000 generations by then,
    # so this is the net under a rare ordering fault of the cheaper fences); with no candidate left the bench fails without a line.
    attempts = []
    while True:
        watch.enter("headline")
        eng.set_state(X0)
        eng.reserve_history(1 + total_gens)
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
        fence("before burn-in")
        t0 = time.perf_counter()
        eng.step(BURNIN_GEN)
        fence("behind burn-in")
        burn_s = time.perf_counter() - t0
        # ---- warm-up: W generations, the last of them through the same timed entry point as the timed region (the first
        # event-bound dispatch of a process sets up profiling signals: 10-30 us of host time, once)
        if args.warmup > 1:
            eng.step(args.warmup - 1)
        if args.warmup > 0:
            eng.step_timed(1)          # (reads the events too: every host-side path of the timed call has run once)
        fence("behind the warm-up")
        # ---- timed region: exactly K generations.  Wall clock for `value`; for the kernel's per-launch duration two time stamps
        # bound to the first and the last update-kernel dispatch of the same K generations (bpm_step_timed; it returns with the
        # sampler's queue / stream drained).
        watch.pending("the timed region: bpm_step_timed (%d generations) + barrier" % args.steps)
        t0 = time.perf_counter()
        eng.step_timed(args.steps, read=False)
        if torch is not None:
            torch.cuda.synchronize()
        else:
            eng.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        ev_ms, ev_launches = eng.last_step_time()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        value = n_chains * args.steps / el

        # ---- dominant kernel (phase_fused_kernel): 2 launches per generation, back to back on one queue; at
        # N = 1 nothing else runs in the region, so event time / launches is its average launch duration
        # (inter-launch gaps included; rocprofv3 --kernel-trace gives the gap-free figure, profiles/).
        n_launch = ev_launches
        k_avg_ms = ev_ms / max(n_launch, 1)
        units_per_launch = CHAINS_PER_GPU / 2.0                       # half the local chains per launch
        achieved = units_per_launch * BYTES_PER_UPDATE / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        # cross-check at N = 1, OUTSIDE the timed region and on a scratch sampler: an event pair around every launch.  bpm_step_profiled is part of the
        # test surface (include/bipymc_hip_test.h): it runs on the test variant of the library -- the same sources, same kernels (steady-state flavour,
        # launched on the HIP stream) -- never on the sampler whose generations are the number
        pair_ms, pair_n = 0.0, 1
        if world == 1 and not use_dist:
            try:
                pe = HipEngine(algo=L.ALGO_DREAM, n_chains=n_chains, dim=DIM, target_id=tid, target_params=tparams, seed=42, device=local_rank,
                               del_pairs=DEL_PAIRS, burnin_gen=0, n_cr_gen=N_CR_GEN, n_cr=3, lib=L.load_test())
                try:
                    pe.set_state(X0)
                    pe.begin_run()
                    pe.step(16)
                    pair_ms, pair_n = pe.step_profiled(32)
                finally:
                    pe.close()
            except Exception as e:                                     # noqa: BLE001 -- a cross-check, never fatal
                sys.stderr.write("bench.py: event-pair cross-check skipped: %s\n" % e)
        fence()
        # HBM bytes per launch from rocprofv3 --pmc (offline, separate passes: tools/profile_bench.sh writes the file) -- tied to the library that is
        # being timed: the file carries the build id of the library the counters were taken on, and another library's counters are not this one's
        traffic, traffic_note = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_cfg2.json")
        if os.path.exists(tfile) and world == 1 and CHAINS_PER_GPU == 8192:
            tj = json.load(open(tfile))
            have = L.build_id(eng.lib)
            if tj.get("build_id") == have:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_note = "profiles/traffic_cfg2.json, counters taken on this build (%s)" % have
            else:
                traffic_note = ("null: profiles/traffic_cfg2.json holds the counters of build %s, the library being timed is %s (re-collect with tools/profile_bench.sh)"
                                % (tj.get("build_id"), have))

        extra = {"evaluated": False}
        if not args.no_moments:
            watch.enter("posterior")
            # the TIMED sampler's own post-burn-in rows vs the analytic moments (mean 0, var_i = i+1), from the on-device reduction over this rank's
            # history rows: a pooled cross-check (1200 generations of 8192 chains resolve the pooled variance to 0.1 %, a single coordinate to ~0.6 %);
            # the per-coordinate 1 % gate needs 30-40 times as many generations and runs on a sampler without history (posterior_gate below).  A short
            # timed region (the driver's 20 generations) is extended by untimed generations: 52 correlated generations say nothing.
            post = args.warmup + args.steps + 32
            if post < POSTERIOR_MIN_GENS:
                eng.step(POSTERIOR_MIN_GENS - post)
                fence("posterior extension")
                post = POSTERIOR_MIN_GENS
            n_burn = (1 + BURNIN_GEN) * n_chains
            cnt, s1, s2, sh = eng.reduce_moments(n_burn)
            if dist is not None:
                pack = torch.tensor(np.concatenate([[cnt], s1, s2]), dtype=torch.float64, device=coll_dev)
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
            pooled_ok = bool(abs(vr - 1.0) < 0.01 and mm < 0.05)
            extra = dict(evaluated=True,
                         # the rows of the sampler that was timed: POOLED variance ratio within 1 %, every mean within 0.05 sigma
                         timed_sampler=dict(generations=int(post), rows=int(cnt), var_ratio_mean=vr, max_abs_mean_over_sigma=mm,
                                            var_ratio_min=float(np.min(var / sig2)), var_ratio_max=float(np.max(var / sig2)), pooled_gate_pass=pooled_ok,
                                            note="per-coordinate spread over %d generations is Monte-Carlo noise (standard error ~0.6 %% per coordinate); "
                                                 "the per-coordinate gate is `gate` below" % post),
                         acceptance_fraction=acc, p_cr=[float(v) for v in st["p_cr"]],
                         # overwritten by the per-coordinate gate at N = 1 (posterior_gate); at N > 1 the pooled form over the sharded sampler's rows stands
                         gate_pass=pooled_ok, gate_kind="pooled, over the timed sampler's rows")

        import hashlib
        state_sha = hashlib.sha256(np.ascontiguousarray(eng.get_state()).tobytes()).hexdigest()[:16]   # (A/B of launch paths: same bits)
        if os.environ.get("BENCH_TEST_FIRST_ATTEMPT_DIVERGES") and not attempts and rank == world - 1:
            state_sha = "0" * 16          # rehearsal hook: exercises the second measurement under the next exchange candidate
        lstat = eng.launch_stats()
        xstat = eng.exchange_stats() if use_dist else None
        # N > 1: what every rank saw -- its update / exchange kernel periods and the size of the world its exchange runs in
        per_rank = None
        if dist is not None:
            mine = dict(rank=rank, device=local_rank, avg_launch_us=k_avg_ms * 1e3, launches_timed=int(n_launch),
                        exchange=xstat, update_dispatches=dict(direct_aql_queue=lstat["direct"], hip_stream=lstat["stream"]),
                        final_state_sha256_16=state_sha)
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        if per_rank is None or len(set(p_["final_state_sha256_16"] for p_ in per_rank)) == 1:
            break
        attempts.append(dict(exchange=exchange_info.get("candidate"), replicas_identical=False))
        left = exchange_info.get("candidates_left") or []
        if not left:
            if rank == 0:
                sys.stderr.write("bench.py: the ranks' replicas differ at the end of the run and no further exchange candidate is left: " + json.dumps(attempts) + "\n")
            try:
                dist.barrier()
                eng.close()
                dist.destroy_process_group()
            except Exception:                                          # noqa: BLE001
                pass
            return 3
        if rank == 0:
            sys.stderr.write("bench.py: the ranks' replicas differ under exchange %r: measuring again under %r\n" % (exchange_info.get("candidate"), left[0]))
        eng.set_exchange(mode=left[0])
        eng.set_adapt_state(t_abs=0)
        exchange_info.update(mode="push" if left[0].startswith("push") else left[0], candidate=left[0], candidates_left=left[1:],
                             fence_scope={"push-agent": "agent", "push": "system"}.get(left[0]))
    out = None
    if rank == 0:
        copy_gbs = measured_copy_bandwidth(torch, local_rank) if torch is not None else None
        out = {
            "metric": "chain-updates/sec" if not args.share_gpu else "REHEARSAL (ranks share one GPU): not a benchmark number",
            "value": value, "unit": "chain-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            # the same K generations by the time stamps of the first / last update dispatch (no host launch / wake-up latency)
            "value_event_timed": (n_chains / (2.0 * k_avg_ms * 1e-3)) if k_avg_ms > 0 else None,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "DREAM, 100-D equicorrelated Gaussian (tests/test_100dgauss.py target), "
                                   "n_chains=%d (%d per GPU), del_pairs=3, n_cr=3; start = EXACT DRAWS OF THE TARGET (not the "
                                   "reference's theta_0=0 + 1e-3 jitter: from there the population needs ~1000 generations to reach "
                                   "the stationary scale), steady state after %d burn-in generations with CR adaptation, history "
                                   "appended every generation; GPU pre-heated for %.2f s (%d untimed steady-state generations on a "
                                   "scratch sampler) before burn-in"
                                   % (n_chains, CHAINS_PER_GPU, BURNIN_GEN, preheat["seconds"], preheat["generations"]),
                       "start": "exact draws of the target", "n_chains": n_chains, "dim": DIM,
                       "parallelism": "chains sharded x%d" % world,
                       "burnin_updates_per_s": n_chains * BURNIN_GEN / burn_s,
                       "exchange": dict(xstat or {}, **(exchange_info or {})) if use_dist else None,
                       # how the update kernels were dispatched: packets written by the library into its own AQL queue
                       # (bipymc_amd/csrc/aql_queue.h) or launches on the HIP stream (BPM_DIRECT_QUEUE=0)
                       "burnin_generations": BURNIN_GEN, "torch_in_the_process": "torch" in sys.modules,
                       "update_dispatches": {"direct_aql_queue": lstat["direct"], "hip_stream": lstat["stream"],
                                             "packet_fence": lstat["fence"]},
                       "final_state_sha256_16": state_sha},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "bpm::phase_fused_kernel<1, 1, 64, 2, 3, %d> (DREAM, Gauss target, 64 lanes/chain, 3 pairs, steady-state instantiation%s)" % ((5, " of the sharded launch mode") if world > 1 else (1, "")),
                         "bytes_per_unit": BYTES_PER_UPDATE,
                         "units_per_launch": units_per_launch, "avg_launch_us": k_avg_ms * 1e3,
                         "launches_timed": n_launch, "avg_launch_us_event_pairs": (pair_ms / pair_n * 1e3) if pair_ms > 0 else None,
                         "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None},
            "posterior": extra,
        }
        if per_rank is not None:
            out["ranks"] = per_rank
            out["config"]["replicas_identical"] = len(set(p["final_state_sha256_16"] for p in per_rank)) == 1
            out["config"]["exchange"]["attempts_discarded"] = attempts
            out["config"]["exchange"]["never_measured"] = ("agent-scope packet fences between GPUs (push-agent) have never run on a multi-GPU node; "
                                                           "the headline uses system scope")
    # From here on rank 0 HAS its line: whatever stalls below, the watchdog prints it (with a "watchdog" entry) and the process ends with
    # EXIT_WATCHDOG -- the line is kept, the return code is not 0 (the round-4 watchdog ended hung ranks with exit code 0).
    if rank == 0:
        watch.headline(out)
    # ---- N > 1: the exchanges the headline did not use, outside its timed region; a FAILURE of an alternative is recorded in the line, never fatal;
    # a HANG ends the run through the watchdog (stage "alternatives": the pending collective is named in the line)
    if dist is not None:
        out_x = out["config"]["exchange"] if rank == 0 else {}
        # what the N > 1 number is a number OF: the default exchange is NOT the RCCL all-gather north_star names (that one is
        # exchange_alternatives[0], mode "dense", measured beside the headline by default)
        mode = (exchange_info or {}).get("mode")
        out_x["rccl_ranks"] = rccl_comm_ranks                      # ranks of the RCCL communicator the library created for THIS sampler (null: none exists)
        out_x["torch_process_group_backend"] = dist.get_backend()
        if rank == 0:
            out["config"]["parallelism"] = ("chains sharded x%d, exchange = %s%s" % (
                world, mode, {"push": " (owners store accepted rows into the peers' replicas through IPC-mapped buffers, %s-scope packet fences; NOT the RCCL "
                                      "all-gather of north_star: that is exchange_alternatives[mode=dense])" % (exchange_info or {}).get("fence_scope"),
                              "dense": " (in-place ncclAllGather of a rank block twice per generation: north_star's exchange, demc.py:93-94,116-117)",
                              "replay": " (ncclAllGather of one accept byte per chain + recomputation of the accepted proposals)",
                              "rows": " (ncclAllGather of packed accepted rows)"}.get(mode, "")))
    if dist is not None and not args.no_exchange_alternatives:
        watch.enter("alternatives")

        def coll_max(x):
            _pend("all_reduce(MAX) of the ranks' elapsed times")
            t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        try:
            alts = exchange_alternatives(eng, dist, world, X0, n_chains, make_single, lambda: make_engine(rccl_uid()), (exchange_info or {}).get("mode"),
                                         coll_max, world if dist.get_backend() == "nccl" else None, share_gpu=args.share_gpu)
        except BaseException as e:                                     # noqa: BLE001 -- never fatal
            alts = [dict(error="%s: %s" % (type(e).__name__, e))]
        if rank == 0:
            out["exchange_alternatives"] = alts
    elif dist is not None and rank == 0:
        out["exchange_alternatives"] = [dict(skipped="--no-exchange-alternatives")]
    watch.enter("teardown")
    if dist is not None:
        watch.pending("dist.barrier before bpm_destroy")
        dist.barrier()                                                 # (the library orders the teardown itself; this keeps the ranks' exits together)
    watch.pending("bpm_destroy")
    eng.close()
    if rank == 0:
        watch.enter("extras")
        if world == 1 and not use_dist and not args.no_other_configs and not args.no_torch and CHAINS_PER_GPU == 8192:
            watch.pending("other_configs")
            out["configs"] = other_configs(local_rank)
            try:
                out["configs"] += host_callback_config(local_rank)
            except Exception as e:                                     # noqa: BLE001
                out["configs"].append(dict(config="cfg2 shape with a host-callback ln_like_fn", error=str(e)))
        if world == 1 and not use_dist and not args.no_moments and CHAINS_PER_GPU == 8192:
            # the per-coordinate gate (every variance within 1 %, every mean within 0.01 sigma, batch-means standard errors): cfg2 from exact draws
            # -- this defines posterior.gate_pass -- and from the reference's start; cfg3 and cfg5; the reference's own scenario beside its family
            out["posterior"]["reference_at_this_configuration"] = reference_at_the_headline_configuration()
            for key, fn in (("gate", lambda: posterior_gate(local_rank, start="exact")),
                            ("gate_from_reference_start", lambda: posterior_gate(local_rank, start="reference")),
                            ("gates_other_configs", lambda: posterior_gates_other_configs(local_rank)),
                            ("reference_scenarios", lambda: reference_scenario_on_the_device(local_rank))):
                watch.pending("posterior." + key)
                try:
                    out["posterior"][key] = fn()
                except Exception as e:                                 # noqa: BLE001
                    out["posterior"][key] = dict(error=str(e))
            g = out["posterior"]["gate"]
            if isinstance(g, dict) and "gate_pass" in g:
                out["posterior"]["gate_pass"] = bool(g["gate_pass"])
                out["posterior"]["gate_kind"] = ("per coordinate: every variance within %g, every |mean| within %g sigma over %d generations "
                                                 "(posterior.gate); batch-means standard errors beside each" % (GATE_VAR_TOL, GATE_MEAN_TOL, g.get("generations", 0)))
                for k in ("var_ratio_min", "var_ratio_max", "var_ratio_mean", "max_abs_mean_over_sigma", "mcse_var_ratio_max", "mcse_mean_over_sigma_max"):
                    out["posterior"][k] = g.get(k)
        if world == 1 and not args.no_cpu_baseline:
            watch.pending("cpu_baseline")
            out["cpu_baseline"] = cpu_baseline()
        out["stage_seconds"] = watch.history + [(watch.stage, round(time.time() - watch.t_stage, 2))]
        write_line(json.dumps(out))
    if dist is not None:
        watch.enter("teardown")
        watch.pending("final dist.barrier / destroy_process_group")
        dist.barrier()
        dist.destroy_process_group()
    watch.done()
    return 0


if __name__ == "__main__":
    sys.exit(main())
