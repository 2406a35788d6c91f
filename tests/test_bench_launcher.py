"""bench.py --gpus N without a launcher: the parent starts N rank processes as children, relays rank 0's JSON line and
fails loudly (CPU; stub workers stand in for the ranks).  The reference's counterpart is `mpirun -n N python script.py`
(bipymc/demc.py:14-32: every rank enters the same collectives in lock-step, demc.py:93-94,116-117)."""
import io
import json
import os
import sys
import time

import bench

STUB = r"""
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
if mode == "ok":
    if rank == 0:
        print("some chatter that is not the line")
        print(json.dumps(dict(metric="chain-updates/sec", value=1.0, n_gpus=world, port=os.environ["MASTER_PORT"], addr=os.environ["MASTER_ADDR"],
                              local_rank=os.environ["LOCAL_RANK"], ipc=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))))
    else:
        print("rank %d says hello" % rank)
    sys.exit(0)
if mode == "fail1":
    if rank == 1:
        sys.exit(3)
    time.sleep(600)          # a peer waiting in a collective for ever
if mode == "silent":
    sys.exit(0)
"""


def _run(mode, n, **kw):
    out, err = io.StringIO(), io.StringIO()

    class E(object):            # Popen needs a real file for the children's stderr: send it to ours
        def fileno(self):
            return sys.stderr.fileno()

        def write(self, s):
            err.write(s)
    rc = bench.launch_ranks(n, [], worker_cmd=[sys.executable, "-c", STUB, mode], out=out, err=E(), **kw)
    return rc, out.getvalue(), err.getvalue()


def test_launcher_relays_rank0_line_and_sets_rendezvous():
    rc, out, err = _run("ok", 3, n_visible=8)
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["addr"] == "127.0.0.1" and d["local_rank"] == "0" and int(d["port"]) > 0
    assert d["ipc"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert "chatter" in err                                  # rank 0's other output goes to stderr


def _one_null_line(out):
    """exactly ONE parseable line, with "value": null and the reason (bench.py --gpus N cannot die silent)"""
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["value"] is None and d["metric"] == "chain-updates/sec" and d["unit"] == "chain-updates/s" and d["higher_is_better"] is True
    return d


def test_launcher_fails_fast_when_fewer_gpus_than_ranks():
    t0 = time.time()
    rc, out, err = _run("ok", 2, n_visible=1)
    assert rc != 0 and time.time() - t0 < 10
    assert "--gpus 2" in err and "1 GPU(s) visible" in err
    d = _one_null_line(out)
    assert d["n_gpus"] == 2 and "1 GPU(s) visible" in d["error"]
    rc, out, err = _run("ok", 2, n_visible=0)
    assert rc != 0 and "no GPU(s) visible" in err and "no GPU(s) visible" in _one_null_line(out)["error"]


def test_launcher_propagates_a_rank_failure_and_ends_the_peers():
    t0 = time.time()
    rc, out, err = _run("fail1", 2, n_visible=2, grace_s=1.0)
    assert rc == 3 and "exit code 3" in _one_null_line(out)["error"]
    assert time.time() - t0 < 30                             # the sleeping peer was ended, not waited for
    assert "rank 1 exited with code 3" in err and "ending rank 0" in err


def test_launcher_needs_the_json_line():
    rc, out, err = _run("silent", 2, n_visible=2)
    assert rc != 0 and "without printing its JSON line" in err and "without printing" in _one_null_line(out)["error"]


def test_gpus_gt_1_without_world_size_takes_the_launcher(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake(n, argv, **kw):
        seen["n"], seen["argv"] = n, list(argv)
        return 7
    monkeypatch.setattr(bench, "launch_ranks", fake)
    assert bench.main(["--gpus", "4", "--steps", "5", "--warmup", "1"]) == 7
    assert seen == {"n": 4, "argv": ["--gpus", "4", "--steps", "5", "--warmup", "1"]}


def test_visible_gpu_count_probe_runs_in_a_child_and_reports_zero_here():
    # this container has no GPU: hipGetDeviceCount fails -> 0 (and -1 only if the library is not built)
    n = bench.visible_gpu_count()
    assert n >= 0, "the probe child could not load bipymc_amd/libbipymc_hip.so"
    if not os.path.exists("/dev/kfd"):
        assert n == 0


# ---- bench.py: which exchange an N > 1 run uses is decided by a run against a single-rank sampler (validate_exchange) ----------------
class _StubDist(object):
    def get_world_size(self):
        return 1

    def barrier(self):
        pass

    def all_gather_object(self, box, v):
        box[0] = v


class _StubEngine(object):
    """final state = a function of the exchange mode: `good` modes reproduce the single-rank state, `bad` ones do not, `boom` ones raise"""
    def __init__(self, good=(), boom=()):
        self.good, self.boom, self.mode, self.modes_set = set(good), set(boom), None, []

    def set_exchange(self, mode):
        self.mode = mode
        self.modes_set.append(mode)

    def set_state(self, X):
        self.X = X

    def set_adapt_state(self, **kw):
        pass

    def begin_run(self):
        pass

    def step(self, n):
        if self.mode in self.boom:
            raise RuntimeError("push exchange: rank 0 waited longer than the limit for rank 1")

    def synchronize(self):
        pass

    def get_state(self):
        import numpy as np
        return self.X + (0.0 if self.mode is None or self.mode in self.good else 1.0)

    def close(self):
        pass


def test_exchange_is_chosen_by_reproducing_the_single_rank_run():
    import numpy as np
    import pytest
    X0 = np.arange(12.0).reshape(3, 4)
    cands = ["push-agent", "push", "replay", "dense"]
    # the cheapest candidate that reproduces the single-rank run wins
    e = _StubEngine(good=cands)
    got = bench.validate_exchange(e, _StubDist(), X0, lambda: _StubEngine(), cands, gens=3)
    assert got["mode"] == "push" and got["fence_scope"] == "agent" and [v["ok"] for v in got["validation"]] == [True]
    # agent-scope fences give other bits: system scope
    e = _StubEngine(good=["push", "replay", "dense"])
    got = bench.validate_exchange(e, _StubDist(), X0, lambda: _StubEngine(), cands, gens=3)
    assert got["mode"] == "push" and got["fence_scope"] == "system" and [v["ok"] for v in got["validation"]] == [False, True]
    assert e.mode == "push"
    # a push wait that times out: the other push candidate is skipped, the time-out mark is cleared (set_exchange away from push), RCCL replay is taken
    e = _StubEngine(good=["replay", "dense"], boom=["push-agent"])
    got = bench.validate_exchange(e, _StubDist(), X0, lambda: _StubEngine(), cands, gens=3)
    assert got["mode"] == "replay" and got["fence_scope"] is None
    assert [v["exchange"] for v in got["validation"]] == cands[:3] and "errors" in got["validation"][0] and "skipped" in got["validation"][1]["why"]
    assert e.modes_set == ["push-agent", "dense", "replay", "replay"]
    # nothing reproduces it: no number is printed
    with pytest.raises(SystemExit) as ei:
        bench.validate_exchange(_StubEngine(good=[]), _StubDist(), X0, lambda: _StubEngine(), cands, gens=3)
    assert "no exchange reproduced" in str(ei.value)


def test_launcher_has_a_wall_clock_limit_when_every_rank_hangs():
    """all ranks hanging WITHOUT exiting (a collective nobody leaves) used to leave the parent polling for ever."""
    stub = "import time; time.sleep(600)"
    out, err = io.StringIO(), io.StringIO()

    class E(object):
        def fileno(self):
            return sys.stderr.fileno()

        def write(self, s):
            err.write(s)
    t0 = time.time()
    rc = bench.launch_ranks(2, [], worker_cmd=[sys.executable, "-c", stub], out=out, err=E(), n_visible=2, deadline_s=1.5)
    assert rc == 124 and time.time() - t0 < 30
    assert "wall-clock limit" in err.getvalue() and "wall-clock limit" in _one_null_line(out.getvalue())["error"]


# ---- a rank that STALLS: every stage of a rank process is under bench.StageWatch ------------------------------------------------------
# A stand-in for bench.main's rank side: the same StageWatch, the same stage sequence, the same hand-over of the headline line -- no GPU.
# BENCH_TEST_STALL=<stage>:<rank> makes that rank hang at the entry of that stage (the hook bench.py itself honours); the peers then wait in
# their "collective" (here: a sleep) until their own watchdogs fire.
RANK_STUB = r"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
argv = sys.argv[1:]
if "--exchange" in argv:                      # the launcher's retry under the dense exchange: a fresh process that does not stall
    os.environ.pop("BENCH_TEST_STALL", None)
lim = float(os.environ.get("STUB_STAGE_LIMIT_S", "1.0"))
w = bench.StageWatch(rank, world, lambda text: (sys.stdout.write(text + chr(10)), sys.stdout.flush()), steps=20, warmup=5,
                     deadline_at=float(os.environ["BENCH_DEADLINE_AT"]), limits={k: lim for k in bench.STAGE_LIMITS_S})
stall = os.environ.get("BENCH_TEST_STALL", "")
def collective(stage):                        # every rank meets here; with a stalled peer nobody leaves
    w.pending("all_gather_object in " + stage)
    if stall and stall.split(":")[0] == stage:
        time.sleep(10 ** 6)
line = None
for stage in ("init", "create", "connect", "validation", "headline", "alternatives", "teardown"):
    w.enter(stage)
    collective(stage)
    if stage == "headline" and rank == 0:
        line = dict(metric="chain-updates/sec", value=5.0e9, unit="chain-updates/s", n_gpus=world, exchange="dense" if "--exchange" in argv else "push", pid=os.getpid())
        w.headline(line)
w.done()
if rank == 0:
    print(json.dumps(line))
"""


def _run_stub_ranks(stall, n=2, **kw):
    out, err = io.StringIO(), io.StringIO()

    class E(object):
        def fileno(self):
            return sys.stderr.fileno()

        def write(self, s):
            err.write(s)
    env = dict(BENCH_TEST_STALL=stall, STUB_STAGE_LIMIT_S="1.0")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        t0 = time.time()
        rc = bench.launch_ranks(n, ["--gpus", str(n), "--steps", "20", "--warmup", "5"], worker_cmd=[sys.executable, "-c", RANK_STUB], out=out, err=E(),
                                n_visible=n, grace_s=3.0, deadline_s=kw.pop("deadline_s", 40.0), **kw)
        el = time.time() - t0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return rc, out.getvalue(), err.getvalue(), el


def test_a_rank_that_stalls_in_connect_validation_or_the_timed_region_ends_with_one_line_and_nonzero_rc():
    import pytest
    for stage in ("connect", "validation", "headline"):
        rc, out, err, el = _run_stub_ranks(stage + ":1", retry_min_s=1e9)         # (no time for a retry)
        assert rc == bench.EXIT_WATCHDOG, (stage, rc, err)
        assert el < 30.0                                                        # ended by the stage limits (1 s each), far below the launcher's deadline
        d = _one_null_line(out)
        assert d["watchdog"]["stage"] == stage and d["watchdog"]["fired"] is True and "all_gather_object in " + stage in d["watchdog"]["pending"]
        assert stage in d["error"] and d["n_gpus"] == 2 and d["steps"] == 20
        assert "watchdog fired" in err
    with pytest.raises(AssertionError):
        _one_null_line("")


def test_a_stall_behind_the_headline_keeps_the_line_but_not_the_return_code():
    """the round-4 watchdog of the alternative exchanges ended hung ranks with os._exit(0) -- every rank reported success.  Now the
    measured line is printed WITH a watchdog entry and the exit code is EXIT_WATCHDOG."""
    rc, out, err, el = _run_stub_ranks("alternatives:0", retry_min_s=1e9)
    assert rc == bench.EXIT_WATCHDOG and el < 30.0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 5.0e9 and d["watchdog"]["stage"] == "alternatives" and d["watchdog"]["pending"] == "BENCH_TEST_STALL"


def test_a_stall_in_the_push_exchange_is_retried_once_in_fresh_processes_under_the_dense_exchange():
    """the first set of ranks stalls in `connect`; the launcher starts a FRESH set of child processes with --exchange dense (never a re-exec of a
    process that touched the GPU) and relays THEIR line, which lists the first attempt; rc 0 (a valid measurement under north_star's exchange)"""
    rc, out, err, el = _run_stub_ranks("connect:1", retry_min_s=5.0)
    assert rc == 0, err
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 5.0e9 and d["exchange"] == "dense"
    a = d["launcher"]["attempts_before_this_one"]
    assert len(a) == 1 and a[0]["rc"] == bench.EXIT_WATCHDOG and a[0]["watchdog"]["stage"] == "connect"
    assert "FRESH rank processes" in err


def test_stage_watch_unit():
    """StageWatch without processes: the absolute deadline cuts a stage's limit; fire() writes rank 0's line once and calls the exit function"""
    written, exits = [], []
    clock = [100.0]
    w = bench.StageWatch(0, 4, written.append, steps=7, warmup=2, deadline_at=103.0, limits=dict(connect=50.0), exit_fn=exits.append, err=io.StringIO(),
                         clock=lambda: clock[0])
    w.enter("connect")
    assert abs(w.limit - 3.0) < 1e-9                       # 50 s asked, 3 s left to the deadline
    w.pending("all_gather_object of the blobs")
    clock[0] = 102.5
    w.fire()
    w.fire()                                               # (a second timer: nothing more)
    w.done()
    assert exits == [bench.EXIT_WATCHDOG] and len(written) == 1
    d = json.loads(written[0])
    assert d["value"] is None and d["n_gpus"] == 4 and d["steps"] == 7 and d["watchdog"]["stage"] == "connect" and d["watchdog"]["seconds_in_stage"] == 2.5
    # with a headline in hand the line is the headline + the watchdog entry; other ranks write nothing
    written, exits = [], []
    w = bench.StageWatch(0, 2, written.append, exit_fn=exits.append, err=io.StringIO())
    w.enter("headline", 60.0)
    w.headline(dict(metric="chain-updates/sec", value=1.0))
    w.enter("alternatives", 60.0)
    w.fire(); w.done()
    d = json.loads(written[0])
    assert d["value"] == 1.0 and d["watchdog"]["stage"] == "alternatives" and ["headline"] == [s_[0] for s_ in d["watchdog"]["stages_done"]]
    written, exits = [], []
    w = bench.StageWatch(1, 2, written.append, exit_fn=exits.append, err=io.StringIO())
    w.enter("connect", 60.0); w.fire(); w.done()
    assert written == [] and exits == [bench.EXIT_WATCHDOG]


class _AltEngine(_StubEngine):
    def __init__(self, probe_agent=True, **kw):
        super(_AltEngine, self).__init__(**kw)
        self.probe_agent, self.closed = probe_agent, False

    def exchange_stats(self):
        return dict(mode=self.mode, arena_probe_agent=self.probe_agent, arena_probe_system=True)

    def close(self):
        self.closed = True


def test_exchange_alternatives_are_timed_beside_the_headline_and_never_fatal():
    """the N > 1 line carries the exchange north_star names (the dense all-gather on a second sampler with the library's own
    RCCL communicator) and the agent-scope push beside the headline; an alternative that fails is recorded, the headline stands."""
    import numpy as np
    X0 = np.arange(12.0).reshape(3, 4)
    made = []

    def make_rccl():
        made.append(_AltEngine(good=["dense"]))
        return made[-1]
    eng = _AltEngine(good=["push", "push-agent"])
    alts = bench.exchange_alternatives(eng, _StubDist(), 1, X0, 3, lambda: _StubEngine(), make_rccl, "push", lambda x: x, 1, gens=5, burn=2)
    assert [a["mode"] for a in alts] == ["dense", "push-agent"]
    d, pa = alts
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["rccl_ranks"] == 1 and d["replicas_identical"] is True and made[0].closed and made[0].mode == "dense"
    assert pa["value"] > 0 and pa["validated_against_single_rank_run"] is True and pa["fence_scope"] == "agent" and "never as it" in pa["note"]
    assert eng.mode == "push"                                   # the sampler is handed back in the headline's mode
    # the second sampler cannot be created (RCCL's first contact fails): recorded; the agent form did not reproduce the single-rank run: recorded
    def boom():
        raise RuntimeError("ncclCommInitRank failed: unhandled system error")
    eng = _AltEngine(good=["push"])
    alts = bench.exchange_alternatives(eng, _StubDist(), 1, X0, 3, lambda: _StubEngine(), boom, "push", lambda x: x, None, gens=5, burn=2)
    assert "ncclCommInitRank failed" in alts[0]["error"][0] and alts[0]["rccl_ranks"] is None and "value" not in alts[0]
    assert "did not reproduce the single-rank run" in alts[1]["error"][0] and eng.mode == "push"
    # the agent-scope arena probe failed: the form is not even tried; ranks sharing one GPU: no RCCL, said so
    eng = _AltEngine(good=["push", "push-agent"], probe_agent=False)
    alts = bench.exchange_alternatives(eng, _StubDist(), 1, X0, 3, lambda: _StubEngine(), make_rccl, "push", lambda x: x, None, share_gpu=True, gens=5, burn=2)
    assert "RCCL refuses two ranks on one device" in alts[0]["skipped"] and alts[0]["rccl_ranks"] is None and alts[0]["rccl_ranks_of_the_torch_process_group"] is None
    assert "arena self-test failed under agent-scope" in alts[1]["skipped"]
    # a headline that already ran dense has nothing to add
    alts = bench.exchange_alternatives(_AltEngine(good=["dense"]), _StubDist(), 1, X0, 3, lambda: _StubEngine(), make_rccl, "dense", lambda x: x, 1, gens=5, burn=2)
    assert len(alts) == 1 and "headline itself" in alts[0]["skipped"]
