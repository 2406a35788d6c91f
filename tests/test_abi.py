"""The C-ABI library: builds, loads, and exports every symbol include/bipymc_hip.h declares.
No compute call is made (no GPU in the CPU test tier)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built():
    so = os.path.join(ROOT, "bipymc_amd", "libbipymc_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bipymc_amd", "csrc")])
    return so


@pytest.fixture(scope="module")
def built_test(built):
    so = os.path.join(ROOT, "build_variants", "libbipymc_test.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bipymc_amd", "csrc")])
    return so


def _declared_symbols(header="bipymc_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bpm_[a-z_0-9]+)\s*\(", text)))


def _exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    return set(re.findall(r" T (bpm_[a-z_0-9]+)", out))


def test_header_and_binding_agree():
    from bipymc_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert sorted(_lib.SIGNATURES) == declared
    assert sorted(_lib.TEST_SIGNATURES) == _declared_symbols("bipymc_hip_test.h")


def test_library_exports_every_declared_symbol(built):
    exported = _exported(built)
    missing = [s for s in _declared_symbols() if s not in exported]
    assert not missing, missing


def test_product_library_carries_no_test_surface(built, built_test):
    """bpm_debug_* / bpm_selftest_* and BPM_TEST_PATHS live in build_variants/libbipymc_test.so (-DBPM_TEST_HOOKS,
    include/bipymc_hip_test.h) only.  The product exports exactly what include/bipymc_hip.h declares and does not contain the variable's name."""
    exported = _exported(built)
    assert exported == set(_declared_symbols()), sorted(exported ^ set(_declared_symbols()))
    assert not [s for s in exported if "debug" in s or "selftest_philox" in s]
    blob = open(built, "rb").read()
    assert b"BPM_TEST_PATHS" not in blob and b"bpm_debug" not in blob
    # round 5: the lock-step driver of local rank groups, the per-launch event profile and the decision trace are test surface too
    assert not [s for s in exported if s in ("bpm_local_group_step", "bpm_step_profiled", "bpm_set_trace", "bpm_get_trace")]
    hooks = _declared_symbols("bipymc_hip_test.h")
    assert len(hooks) >= 12 and all("debug" in h or "selftest" in h or h in ("bpm_local_group_step", "bpm_step_profiled", "bpm_set_trace", "bpm_get_trace")
                                    for h in hooks)
    test_exported = _exported(built_test)
    assert test_exported == set(_declared_symbols()) | set(hooks)
    assert b"BPM_TEST_PATHS" in open(built_test, "rb").read()


def test_library_loads_and_reports_abi(built):
    from bipymc_amd import _lib
    lib = _lib.load()
    assert lib.bpm_abi_version() == _lib.ABI_VERSION
    assert lib.bpm_last_error() is not None


def test_binaries_are_built_from_the_sources_in_the_tree(built, built_test, tmp_path, monkeypatch):
    """*.so is git-ignored and travels to the GPU box as built -- what proves the tested binary is the tested source?  The
    Makefile bakes the SHA-256 of the sources (ID_SRCS) into every library; the binding recomputes it from the tree and refuses another."""
    import ctypes as C
    import hashlib
    from bipymc_amd import _lib
    want = _lib.source_id()
    assert want is not None and re.fullmatch(r"[0-9a-f]{16}", want)
    # the Makefile's list and the binding's list are the same files in the same order
    mk = open(os.path.join(ROOT, "bipymc_amd", "csrc", "Makefile")).read()
    ids = re.search(r"^ID_SRCS = (.*)$", mk, re.M).group(1).split()
    norm = lambda rel: os.path.normpath(os.path.join(ROOT, "bipymc_amd", "csrc", rel))
    assert [norm(x) for x in ids] == [os.path.normpath(os.path.join(ROOT, "bipymc_amd", x)) for x in _lib._ID_SRCS]
    h = hashlib.sha256()
    for x in ids:
        h.update(open(norm(x), "rb").read())
    assert h.hexdigest()[:16] == want
    for so in (built, built_test):
        lib = C.CDLL(so)
        lib.bpm_build_id.restype = C.c_char_p
        assert lib.bpm_build_id().decode() == want, (so, "stale binary: run make -C bipymc_amd/csrc")
    assert _lib.build_id(_lib.load()) == want and _lib.build_id(_lib.load_test()) == want
    # a library built from OTHER sources is refused, loudly
    monkeypatch.setattr(_lib, "source_id", lambda: "0123456789abcdef")
    with pytest.raises(ImportError, match="built from other sources"):
        _lib._bind(built, hooks=None)
    monkeypatch.setenv("BPM_ALLOW_STALE_LIB", "1")
    assert _lib._bind(built, hooks=None) is not None


def test_struct_layouts_match_header(built, tmp_path):
    """sizeof/offsetof of the three ABI structs as the C compiler sees them == ctypes."""
    import ctypes as C
    from bipymc_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bipymc_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(bpm_config_t), sizeof(bpm_run_opts_t),'
                   ' sizeof(bpm_stats_t), offsetof(bpm_config_t, seed), offsetof(bpm_config_t, gamma_scale),'
                   ' offsetof(bpm_config_t, keep_history), offsetof(bpm_stats_t, p_cr));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    exp = [C.sizeof(_lib.BpmConfig), C.sizeof(_lib.BpmRunOpts), C.sizeof(_lib.BpmStats), _lib.BpmConfig.seed.offset,
           _lib.BpmConfig.gamma_scale.offset, _lib.BpmConfig.keep_history.offset, _lib.BpmStats.p_cr.offset]
    assert got == exp


def test_no_gpu_means_loud_failure(built):
    """The product has no CPU path: creating a sampler without a device is an error, not a fallback."""
    import numpy as np
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    from bipymc_amd import _lib
    from bipymc_amd.engine import HipEngine
    with pytest.raises(_lib.BpmError):
        HipEngine(algo=0, n_chains=8, dim=2, target_id=3, target_params=np.zeros(9), seed=1)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under bipymc_amd/ may import, include, link or load it."""
    pkg = os.path.join(ROOT, "bipymc_amd")
    bad = re.compile(r"^\s*(import\s+oracle|from\s+oracle|from\s+\.\.?oracle|#\s*include\s+\"[^\"]*oracle)|dlopen\([^)]*oracle|CDLL\([^)]*oracle",
                     re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), os.path.join(dirpath, f)


def test_destroy_plan_never_frees_under_a_failed_queue(built_test):
    """bpm_destroy's decision about the sampler's device buffers (sampler.hip: bpm_debug_destroy_plan): freed unless the library's own
    queue failed AND could not be quiesced (kernels that are slow, not dead, would write freed memory).  The reference
    has no counterpart: its chains are NumPy arrays (chain.py:13-29)."""
    from bipymc_amd import _lib
    lib = _lib.load_test()                              # (a hook of the test variant: the product's bpm_destroy calls the same function)
    assert lib.bpm_debug_destroy_plan(0, 1) == 1        # healthy queue, drained: free
    assert lib.bpm_debug_destroy_plan(0, 0) == 1        # (no failure: nothing can still run)
    assert lib.bpm_debug_destroy_plan(1, 1) == 1        # failed, then inactivated: free
    assert lib.bpm_debug_destroy_plan(1, 0) == 0        # failed and not quiesced: LEAK


def test_build_variants_have_disjoint_kernel_symbols(built, built_test):
    """The library's own AQL queue finds its kernels BY NAME among all code objects the HSA loader holds (aql_queue.h: DirectQueue::kernel).  A process
    that holds the product and the test variant at once (the GPU tests, bench.py's event-pair cross-check) must never dispatch one library's packet
    with the other's kernel: the variants' argument blocks differ (the product has no trace fields) -- a GPU memory fault at address 0 when it
    happened (round 5, gpurun_out/r5a).  Every variant's device code lives in an inline namespace of its own (philox.h: BPM_VARIANT_NS)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import kernel_resources
    prod = set(k["name"] for k in kernel_resources(built))
    hooks = set(k["name"] for k in kernel_resources(built_test))
    assert len(prod) > 150 and len(hooks) >= len(prod)
    assert not (prod & hooks), sorted(prod & hooks)[:5]
    assert all("7product" in n for n in prod), [n for n in prod if "7product" not in n][:5]      # (Itanium mangling of bpm::product::)
    assert all("5hooks" in n for n in hooks), [n for n in hooks if "5hooks" not in n][:5]


def test_no_kernel_spills_or_touches_scratch_memory(built):
    """kernels are dispatched through the library's own AQL queue by what the code object says about them.  No kernel of the
    library may spill registers or execute a scratch instruction -- whatever its shape (round 3's 32-coordinates-per-lane update kernel spilled
    ~1 KB per lane and had to stay on the HIP stream; the looped wide-row kernel replaced it).  Read from the code object's metadata and its
    disassembly; no GPU needed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import LLVM, kernel_resources
    ks = kernel_resources(built)
    assert len(ks) > 150
    assert not [k["name"] for k in ks if k["spill"] != 0]
    wide = [k for k in ks if "phase_wide_kernel" in k["name"] or "eval_ll_wide" in k["name"] or "phase_wide_commit" in k["name"]]
    assert len(wide) >= 10 and all(k["priv"] == 0 and k["vgpr"] <= 128 for k in wide), wide
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, built, os.path.join(td, "x.so")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        asm = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co]).decode()
    assert asm.count("scratch_load") + asm.count("scratch_store") == 0


def test_hip_source_likelihood_compiles_without_a_gpu():
    """bpm_check_device_likelihood (include/bipymc_hip.h): hiprtc builds the caller's ln_like + the wrapper kernel for gfx950 with no device in the
    machine; a source that does not compile comes back as an error carrying the compiler's log (through HipLikelihood.check as ValueError)."""
    from bipymc_amd import HipLikelihood
    ok = HipLikelihood("__device__ double ln_like(const double* x, int d, const double* p) { double s = 0; for (int j = 0; j < d; ++j) s += x[j] * x[j] * p[0]; return -0.5 * s; }",
                       params=[2.0])
    assert ok.check() and ok.check("gfx950")
    with pytest.raises(ValueError, match="does not compile(.|\n)*expected"):
        HipLikelihood("__device__ double ln_like(const double* x, int d, const double* p) { return x[0] }").check()
    with pytest.raises(ValueError, match="ln_like"):                 # the wrapper calls a function the source does not define
        HipLikelihood("__device__ double other(const double* x) { return x[0]; }").check()
    with pytest.raises(TypeError):
        ok(np.zeros(3))                                              # no python_fn: device only
    # the per-coordinate form: ln_like is derived from ln_like_terms / ln_like_finish
    terms = HipLikelihood("__device__ void ln_like_terms(double xj, int j, int d, const double* p, double* acc) { acc[0] += xj * xj; }\n"
                          "__device__ double ln_like_finish(const double* acc, int d, const double* p) { return -0.5 * acc[0]; }", terms=1)
    assert terms.check() and terms.source.startswith("#define BPM_LN_LIKE_TERMS 1")
    with pytest.raises(ValueError):
        HipLikelihood("", terms=9)
