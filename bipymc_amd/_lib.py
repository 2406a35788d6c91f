"""ctypes binding of libbipymc_hip.so (include/bipymc_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible
the engine fails loudly.
"""
import ctypes as C
import os


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPM_LIB_PATH") or os.path.join(_HERE, "libbipymc_hip.so")   # override: experiment builds (tools/)
# the same sources built with -DBPM_TEST_HOOKS (include/bipymc_hip_test.h): bpm_debug_* / bpm_selftest_philox and the BPM_TEST_PATHS switches.
# Only tests and tools load it (load_test(), or BPM_LIB_PATH=<this> for a whole child process); the product library has none of it.
TEST_LIB_PATH = os.path.join(os.path.dirname(_HERE), "build_variants", "libbipymc_test.so")

ABI_VERSION = 2
ALGO_DEMC, ALGO_DREAM, ALGO_DEMC_SYNC = 0, 1, 2
TARGET_HOST_CALLBACK, TARGET_GAUSS_EQUICORR, TARGET_MIXTURE_PAIRS, TARGET_BANANA_2D = 0, 1, 2, 3
MAX_CR = 8
UID_BYTES = 128
PUSH_BLOB_BYTES = 256
TRACE_I32, TRACE_F64, MAX_PARTNERS = 32, 4, 23


class BpmConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("algo", C.c_int32), ("n_chains", C.c_int32), ("dim", C.c_int32),
        ("target_id", C.c_int32), ("n_target_params", C.c_int32), ("target_params", C.POINTER(C.c_double)),
        ("seed", C.c_uint64), ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("nccl_uid", C.c_char_p),
        ("gamma_scale", C.c_double), ("del_pairs", C.c_int32), ("burnin_gen", C.c_int32),
        ("n_cr_gen", C.c_int32), ("n_cr", C.c_int32),
        ("p_snooker", C.c_double), ("outlier_every", C.c_int32), ("keep_history", C.c_int32),
        ("running_moments", C.c_int32), ("_pad2", C.c_int32),
    ]


class BpmRunOpts(C.Structure):
    _fields_ = [("flip", C.c_double), ("shuffle", C.c_int32), ("_pad", C.c_int32), ("epsilon", C.c_double),
                ("u_epsilon", C.c_double), ("gamma", C.c_double)]


class BpmStats(C.Structure):
    _fields_ = [
        ("local_n_accepted", C.c_int64), ("local_n_rejected", C.c_int64), ("n_nan_alpha", C.c_int64),
        ("k_gen", C.c_int64), ("t_abs", C.c_int64), ("history_rows", C.c_int64), ("n_outlier_resets", C.c_int64),
        ("n_cr", C.c_int32), ("_pad", C.c_int32),
        ("p_cr", C.c_double * MAX_CR), ("delta_m", C.c_double * MAX_CR), ("n_cr_updates", C.c_double * MAX_CR),
    ]


_P = C.POINTER
_dp, _ip, _u8p, _u32p = _P(C.c_double), _P(C.c_int32), _P(C.c_uint8), _P(C.c_uint32)
_H = C.c_void_p

# name -> (restype, argtypes): every symbol include/bipymc_hip.h declares
SIGNATURES = {
    "bpm_last_error": (C.c_char_p, []),
    "bpm_abi_version": (C.c_int, []),
    "bpm_build_id": (C.c_char_p, []),
    "bpm_device_count": (C.c_int, [_ip]),
    "bpm_get_unique_id": (C.c_int, [C.c_char_p]),
    "bpm_create": (C.c_int, [_P(BpmConfig), _P(_H)]),
    "bpm_destroy": (C.c_int, [_H]),
    "bpm_init_chains": (C.c_int, [_H, _dp, _dp]),
    "bpm_set_state": (C.c_int, [_H, _dp]),
    "bpm_get_state": (C.c_int, [_H, _dp]),
    "bpm_set_loglike": (C.c_int, [_H, _dp]),
    "bpm_get_loglike": (C.c_int, [_H, _dp]),
    "bpm_begin_run": (C.c_int, [_H, _P(BpmRunOpts)]),
    "bpm_step": (C.c_int, [_H, C.c_int64]),
    "bpm_step_timed": (C.c_int, [_H, C.c_int64, _P(C.c_float), _P(C.c_int64)]),
    "bpm_get_step_time": (C.c_int, [_H, _P(C.c_float), _P(C.c_int64)]),
    "bpm_synchronize": (C.c_int, [_H]),
    "bpm_set_exchange": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bpm_get_exchange_stats": (C.c_int, [_H, _P(C.c_int64)]),
    "bpm_push_export": (C.c_int, [_H, C.c_void_p]),
    "bpm_push_connect": (C.c_int, [_H, C.c_void_p]),
    "bpm_push_selftest": (C.c_int, [_P(_H), C.c_int32, _P(C.c_int32)]),
    "bpm_get_launch_stats": (C.c_int, [_H, _P(C.c_int64)]),
    "bpm_set_launch_path": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bpm_set_history": (C.c_int, [_H, C.c_int64, _dp, _dp]),
    "bpm_reduce_moments": (C.c_int, [_H, C.c_int64, _dp, _dp, _dp, _P(C.c_int64)]),
    "bpm_propose": (C.c_int, [_H, _dp, _ip, _ip]),
    "bpm_commit": (C.c_int, [_H, _dp]),
    "bpm_propose_begin": (C.c_int, [_H, C.c_int32]),
    "bpm_propose_chunk": (C.c_int, [_H, C.c_int32, _P(_dp), _P(_ip), _ip, _ip]),
    "bpm_commit_chunk": (C.c_int, [_H, C.c_int32, _dp]),
    "bpm_commit_end": (C.c_int, [_H]),
    "bpm_propose_device": (C.c_int, [_H, _P(C.c_void_p), _P(C.c_void_p), _ip, _ip]),
    "bpm_commit_device": (C.c_int, [_H, C.c_void_p]),
    "bpm_state_device": (C.c_int, [_H, _P(C.c_void_p), _ip, _ip]),
    "bpm_set_loglike_device": (C.c_int, [_H, C.c_void_p]),
    "bpm_set_device_likelihood": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_double), C.c_int32]),
    "bpm_refresh_device_loglike": (C.c_int, [_H]),
    "bpm_check_device_likelihood": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64]),
    "bpm_get_device_likelihood_info": (C.c_int, [_H, C.POINTER(C.c_int32), C.c_char_p, C.c_int64]),
    "bpm_get_history": (C.c_int, [_H, C.c_int64, C.c_int64, _dp]),
    "bpm_get_loglike_history": (C.c_int, [_H, C.c_int64, C.c_int64, _dp]),
    "bpm_reserve_history": (C.c_int, [_H, C.c_int64]),
    "bpm_get_stats": (C.c_int, [_H, _P(BpmStats)]),
    "bpm_set_adapt_state": (C.c_int, [_H, _dp, _dp, _dp, C.c_int64]),
    "bpm_eval_loglike": (C.c_int, [_H, _dp, C.c_int32, _dp]),
}

# include/bipymc_hip_test.h: exported by the test variant only
TEST_SIGNATURES = {
    "bpm_local_group_step": (C.c_int, [_P(_H), C.c_int32, C.c_int64]),
    "bpm_step_profiled": (C.c_int, [_H, C.c_int64, _dp, _P(C.c_int64)]),
    "bpm_set_trace": (C.c_int, [_H, C.c_int32]),
    "bpm_get_trace": (C.c_int, [_H, _ip, _dp, _u8p]),
    "bpm_debug_destroy_plan": (C.c_int, [C.c_int32, C.c_int32]),
    "bpm_debug_fail_queue": (C.c_int, [_H, C.c_int32]),
    "bpm_debug_queue_pad": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int64)]),
    "bpm_debug_coherence_probe": (C.c_int, [C.c_int32, C.c_int32, _P(C.c_int64)]),
    "bpm_selftest_philox": (C.c_int, [C.c_int32, C.c_int32, C.c_uint64, _u32p, _u32p]),
    "bpm_debug_perm": (C.c_int, [_H, C.c_int64, C.c_int32, C.c_double, _ip, _ip, _ip]),
    "bpm_debug_outlier_select": (C.c_int, [_H, _dp, _dp]),
    "bpm_debug_time_kernels": (C.c_int, [_H, C.c_int32, _P(C.c_float), _P(C.c_float)]),
}

_lib = None
_test_lib = None


class BpmError(RuntimeError):
    pass


def load():
    """Load the C-ABI library and bind every declared entry point. No GPU is touched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "bipymc_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bipymc_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    _lib = _bind(LIB_PATH, hooks=None)
    return _lib


# The files a library's build id is the SHA-256 of, in this order (bipymc_amd/csrc/Makefile: ID_SRCS)
_ID_SRCS = ("csrc/sampler.hip", "csrc/kernels.h", "csrc/kernels_wide.h", "csrc/philox.h", "csrc/rocrand_check.h", "csrc/aql_queue.h", "csrc/user_likelihood.h",
            "../include/bipymc_hip.h", "../include/bipymc_hip_test.h", "csrc/Makefile")


def source_id():
    """The build id of the sources IN THE TREE: first 16 hex digits of the SHA-256 over _ID_SRCS' bytes -- what bpm_build_id() of a library built
    from them returns (the Makefile bakes it in).  None when a source file is missing (an installation without sources)."""
    import hashlib
    h = hashlib.sha256()
    for rel in _ID_SRCS:
        path = os.path.join(_HERE, rel)
        if not os.path.exists(path):
            return None
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_id(lib):
    v = lib.bpm_build_id()
    return v.decode("ascii", "replace") if v else ""


def _bind(path, hooks):
    """hooks: True = the test surface must be there, None = bind it when it is (BPM_LIB_PATH may name a test / experiment build)"""
    lib = C.CDLL(path)       # RTLD_LOCAL: the product and the test variant define the same symbols and may live in one process
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in TEST_SIGNATURES.items():
        if not hooks and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.bpm_abi_version() != ABI_VERSION:
        raise ImportError("bipymc_amd: %s ABI %d != binding ABI %d" % (os.path.basename(path), lib.bpm_abi_version(), ABI_VERSION))
    # stale-binary guard: *.so is built in-tree and travels to the GPU box as built; what runs must be what the tree says
    want, have = source_id(), build_id(lib)
    if want is not None and have != want and os.environ.get("BPM_ALLOW_STALE_LIB", "0") != "1":
        raise ImportError("bipymc_amd: %s was built from other sources (its build id %s, the tree's %s): rebuild with `make -C bipymc_amd/csrc` "
                          "(or __graft_entry__.build(force=True)); BPM_ALLOW_STALE_LIB=1 loads it anyway" % (path, have, want))
    return lib


def load_test():
    """The test variant (build_variants/libbipymc_test.so): every entry point of the product plus include/bipymc_hip_test.h.  Tests / tools only."""
    global _test_lib
    if _test_lib is None:
        if os.path.abspath(LIB_PATH) == os.path.abspath(TEST_LIB_PATH):
            _test_lib = load()
        else:
            if not os.path.exists(TEST_LIB_PATH):
                raise ImportError("bipymc_amd: %s not found (make -C bipymc_amd/csrc builds it beside the product library)" % TEST_LIB_PATH)
            _test_lib = _bind(TEST_LIB_PATH, hooks=True)
    return _test_lib


def want_dmabuf_ipc():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: what hipIpcGetMemHandle of the push exchange and RCCL need on hosts whose driver supports
    only that mode), as a default, for the MULTI-RANK paths only -- DeMcMpi with a communicator of more than one rank, bench.py's rank
    processes.  Not at import: it changes IPC behaviour for every HSA user of the process, single-GPU users included.  It takes effect only
    if nothing has initialised HSA yet (set it in the launcher's environment to be sure); where it did not, bpm_push_connect's error message
    names the variable."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def device_count():
    n = C.c_int32(0)
    check(load().bpm_device_count(C.byref(n)))
    return int(n.value)


def check(rc, lib=None):
    """lib: the library the failing call was made on (its bpm_last_error holds the message); default the product library"""
    if rc != 0:
        msg = (lib or load()).bpm_last_error()
        raise BpmError(msg.decode("utf-8", "replace") if msg else "libbipymc_hip error %d" % rc)
