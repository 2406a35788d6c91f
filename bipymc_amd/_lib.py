"""ctypes binding of libbipymc_hip.so (include/bipymc_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible
the engine fails loudly.
"""
import ctypes as C
import os

# dmabuf IPC (hipIpcGetMemHandle of the push exchange, RCCL): takes effect only if nothing has started HSA yet -- import this package (or
# set the variable) before torch in a multi-process run; harmless in a single process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPM_LIB_PATH") or os.path.join(_HERE, "libbipymc_hip.so")   # override: experiment builds (tools/)

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
    "bpm_device_count": (C.c_int, [_ip]),
    "bpm_get_unique_id": (C.c_int, [C.c_char_p]),
    "bpm_create": (C.c_int, [_P(BpmConfig), _P(_H)]),
    "bpm_destroy": (C.c_int, [_H]),
    "bpm_debug_destroy_plan": (C.c_int, [C.c_int32, C.c_int32]),
    "bpm_debug_fail_queue": (C.c_int, [_H, C.c_int32]),
    "bpm_debug_queue_pad": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int64)]),
    "bpm_init_chains": (C.c_int, [_H, _dp, _dp]),
    "bpm_set_state": (C.c_int, [_H, _dp]),
    "bpm_get_state": (C.c_int, [_H, _dp]),
    "bpm_set_loglike": (C.c_int, [_H, _dp]),
    "bpm_get_loglike": (C.c_int, [_H, _dp]),
    "bpm_begin_run": (C.c_int, [_H, _P(BpmRunOpts)]),
    "bpm_step": (C.c_int, [_H, C.c_int64]),
    "bpm_step_timed": (C.c_int, [_H, C.c_int64, _P(C.c_float), _P(C.c_int64)]),
    "bpm_get_step_time": (C.c_int, [_H, _P(C.c_float), _P(C.c_int64)]),
    "bpm_step_profiled": (C.c_int, [_H, C.c_int64, _dp, _P(C.c_int64)]),
    "bpm_synchronize": (C.c_int, [_H]),
    "bpm_local_group_step": (C.c_int, [_P(_H), C.c_int32, C.c_int64]),
    "bpm_set_exchange": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bpm_get_exchange_stats": (C.c_int, [_H, _P(C.c_int64)]),
    "bpm_push_export": (C.c_int, [_H, C.c_void_p]),
    "bpm_push_connect": (C.c_int, [_H, C.c_void_p]),
    "bpm_push_selftest": (C.c_int, [_P(_H), C.c_int32, _P(C.c_int32)]),
    "bpm_get_launch_stats": (C.c_int, [_H, _P(C.c_int64)]),
    "bpm_set_launch_path": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bpm_debug_coherence_probe": (C.c_int, [C.c_int32, C.c_int32, _P(C.c_int64)]),
    "bpm_set_history": (C.c_int, [_H, C.c_int64, _dp, _dp]),
    "bpm_reduce_moments": (C.c_int, [_H, C.c_int64, _dp, _dp, _dp, _P(C.c_int64)]),
    "bpm_propose": (C.c_int, [_H, _dp, _ip, _ip]),
    "bpm_commit": (C.c_int, [_H, _dp]),
    "bpm_get_history": (C.c_int, [_H, C.c_int64, C.c_int64, _dp]),
    "bpm_get_loglike_history": (C.c_int, [_H, C.c_int64, C.c_int64, _dp]),
    "bpm_reserve_history": (C.c_int, [_H, C.c_int64]),
    "bpm_get_stats": (C.c_int, [_H, _P(BpmStats)]),
    "bpm_set_adapt_state": (C.c_int, [_H, _dp, _dp, _dp, C.c_int64]),
    "bpm_eval_loglike": (C.c_int, [_H, _dp, C.c_int32, _dp]),
    "bpm_selftest_philox": (C.c_int, [C.c_int32, C.c_int32, C.c_uint64, _u32p, _u32p]),
    "bpm_debug_perm": (C.c_int, [_H, C.c_int64, C.c_int32, C.c_double, _ip, _ip, _ip]),
    "bpm_debug_outlier_select": (C.c_int, [_H, _dp, _dp]),
    "bpm_debug_time_kernels": (C.c_int, [_H, C.c_int32, _P(C.c_float), _P(C.c_float)]),
    "bpm_set_trace": (C.c_int, [_H, C.c_int32]),
    "bpm_get_trace": (C.c_int, [_H, _ip, _dp, _u8p]),
}

_lib = None


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
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.bpm_abi_version() != ABI_VERSION:
        raise ImportError("bipymc_amd: libbipymc_hip.so ABI %d != binding ABI %d" % (lib.bpm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def device_count():
    n = C.c_int32(0)
    check(load().bpm_device_count(C.byref(n)))
    return int(n.value)


def check(rc):
    if rc != 0:
        msg = load().bpm_last_error()
        raise BpmError(msg.decode("utf-8", "replace") if msg else "libbipymc_hip error %d" % rc)
