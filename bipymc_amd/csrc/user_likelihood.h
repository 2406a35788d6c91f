// A caller's ln_like_fn given as HIP SOURCE (samplers.py:36-43 takes any Python callable): compiled at run time with hiprtc into ONE
// small kernel -- a thread per proposal row calls the caller's `ln_like` -- that runs between the library's own proposal and commit kernels
// (phase_propose_kernel / phase_commit_kernel, the kernels of the host-callback path).  Nothing leaves the device and no host code runs inside a
// generation: bpm_step drives such a sampler like one with a shipped target.  The update kernels themselves are not recompiled.
//
// What the caller writes (HIP device code; double precision; no includes needed):
//     __device__ double ln_like(const double* x, int d, const double* p)      // x: one parameter vector, p: the caller's parameter block
// (the device math functions, INFINITY, NAN and M_PI are there; a prior outside its support returns -INFINITY like a Python ln_like_fn would)
// hiprtc is loaded on demand (libhiprtc.so): a process that never installs such a likelihood never needs it.
#pragma once
#include <dlfcn.h>

#include <cstdint>
#include <string>
#include <vector>

namespace bpm {
inline namespace BPM_VARIANT_NS {

struct Hiprtc {
    void* lib = nullptr;
    int (*CreateProgram)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*CompileProgram)(void*, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(void*, size_t*) = nullptr;
    int (*GetProgramLog)(void*, char*) = nullptr;
    int (*GetCodeSize)(void*, size_t*) = nullptr;
    int (*GetCode)(void*, char*) = nullptr;
    int (*DestroyProgram)(void**) = nullptr;
};

// -> "" or the reason hiprtc cannot be used
inline std::string load_hiprtc(Hiprtc& h) {
    if (h.lib) return "";
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) return std::string("cannot load hiprtc: ") + dlerror();
#define BPM_RTC_SYM(f)                                                              \
    h.f = reinterpret_cast<decltype(h.f)>(dlsym(lib, "hiprtc" #f));                 \
    if (!h.f) return "hiprtc symbol hiprtc" #f " missing";
    BPM_RTC_SYM(CreateProgram) BPM_RTC_SYM(CompileProgram) BPM_RTC_SYM(GetProgramLogSize) BPM_RTC_SYM(GetProgramLog)
    BPM_RTC_SYM(GetCodeSize) BPM_RTC_SYM(GetCode) BPM_RTC_SYM(DestroyProgram)
#undef BPM_RTC_SYM
    h.lib = lib;
    return "";
}

// The kernel around the caller's function: work item i of a half generation (row i of `rows`, stride ld) -> out[i]; an inactive work item (ids[i] < 0:
// a rank of a world launches one item per local chain and half of them sit in the other pool) is not evaluated.  ids == nullptr: every row.
// A workgroup of 64 threads takes `rpb` consecutive rows: it copies them -- contiguous in memory, so the loads coalesce -- into LDS (row stride ldp
// doubles, odd: the threads' column reads spread over the banks), then thread r calls the caller's function on row r in LDS.  A thread per row reading
// its row straight from memory (stride 800 B between the lanes at d = 100) took 15.6 us for 4096 rows; staged: see profiles/r05_hip_source_likelihood.txt.
// rpb == 0 (rows too wide for a useful tile): every thread reads its row where it lies.
constexpr int USER_EVAL_BLOCK = 64;
constexpr int USER_EVAL_LDS_BYTES = 60 * 1024;
inline const char* user_eval_wrapper() {
    return "\nextern \"C\" __global__ void __launch_bounds__(64) bpm_user_eval(const double* rows, const int* ids, int n, int ld, int d, const double* params,\n"
           "                                                              double* out, int rpb, int ldp) {\n"
           "    extern __shared__ double bpm_tile[];\n"
           "    if (rpb == 0) {\n"
           "        const int i = (int)(blockIdx.x * 64 + threadIdx.x);\n"
           "        if (i < n) out[i] = (ids == nullptr || ids[i] >= 0) ? (double)ln_like(rows + (unsigned long long)i * (unsigned long long)ld, d, params) : 0.0;\n"
           "        return;\n"
           "    }\n"
           "    const int r0 = (int)blockIdx.x * rpb;\n"
           "    const int nr = (n - r0) < rpb ? (n - r0) : rpb;\n"
           "    // the workgroup's nr rows lie back to back (ld even: 16-byte pairs): pair k of the region -> row k / (ld / 2), 8 pairs per thread in flight\n"
           "    typedef double bpm_d2 __attribute__((ext_vector_type(2)));\n"
           "    const bpm_d2* src = (const bpm_d2*)(rows + (unsigned long long)r0 * (unsigned long long)ld);\n"
           "    const int h = ld >> 1, total = nr * h;\n"
           "    for (int k0 = 0; k0 < total; k0 += 64 * 8) {\n"
           "        bpm_d2 v[8];\n"
           "#pragma unroll\n"
           "        for (int u = 0; u < 8; ++u) { const int k = k0 + u * 64 + (int)threadIdx.x; v[u] = src[k < total ? k : total - 1]; }\n"
           "#pragma unroll\n"
           "        for (int u = 0; u < 8; ++u) {\n"
           "            const int k = k0 + u * 64 + (int)threadIdx.x;\n"
           "            if (k < total) { const int r = k / h, j = 2 * (k - r * h); bpm_tile[r * ldp + j] = v[u].x; if (j + 1 < d) bpm_tile[r * ldp + j + 1] = v[u].y; }\n"
           "        }\n"
           "    }\n"
           "    __syncthreads();\n"
           "    const int t = (int)threadIdx.x;\n"
           "    if (t < nr) out[r0 + t] = (ids == nullptr || ids[r0 + t] >= 0) ? (double)ln_like(bpm_tile + t * ldp, d, params) : 0.0;\n"
           "}\n";
}
// rows per workgroup and their LDS stride for dimension d (0, *: no tile)
inline void user_eval_tile(uint32_t d, int& rpb, int& ldp) {
    ldp = (int)(d | 1u);
    const long rows = (long)USER_EVAL_LDS_BYTES / ((long)ldp * 8);
    // (16 rows per workgroup: 4096 rows are 256 workgroups, one per CU, each with 13 16-byte loads per thread in flight -- 64 rows per workgroup left 192
    // CUs idle and took 35 us with a row-by-row copy loop)
    rpb = rows >= 4 ? (int)(rows < 16 ? rows : 16) : 0;
}

// user source + wrapper -> code object for `arch` ("gfx950", or a device's gcnArchName).  -> "" and `code`, or the reason (compiler log included).
// f64 arithmetic unfused (-ffp-contract=off), like the library's own kernels: a formula written the same way in NumPy gives the same bits.
inline std::string compile_user_likelihood(Hiprtc& h, const std::string& user_src, const std::string& arch, std::vector<char>& code) {
    const std::string why = load_hiprtc(h);
    if (!why.empty()) return why;
    // (hiprtc declares the device math functions -- exp, log, sqrt, lgamma, erf ... -- but not <cmath>'s macros: a prior returns -INFINITY)
    static const char* prelude =
        "#ifndef INFINITY\n#define INFINITY (__builtin_huge_val())\n#endif\n"
        "#ifndef NAN\n#define NAN (__builtin_nan(\"\"))\n#endif\n"
        "#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n"
        "#line 1 \"ln_like.hip\"\n";
    const std::string src = prelude + user_src + user_eval_wrapper();
    void* prog = nullptr;
    if (h.CreateProgram(&prog, src.c_str(), "ln_like.hip", 0, nullptr, nullptr) != 0 || !prog) return "hiprtcCreateProgram failed";
    const std::string a = "--offload-arch=" + arch;
    const char* opts[] = {a.c_str(), "-O3", "-ffp-contract=off"};
    const int rc = h.CompileProgram(prog, 3, opts);
    std::string log;
    size_t n = 0;
    if (h.GetProgramLogSize(prog, &n) == 0 && n > 1) {
        log.resize(n);
        if (h.GetProgramLog(prog, &log[0]) != 0) log.clear();
        while (!log.empty() && (log.back() == '\0' || log.back() == '\n')) log.pop_back();
    }
    if (rc != 0) {
        h.DestroyProgram(&prog);
        return "the likelihood source does not compile (it must define `__device__ double ln_like(const double* x, int d, const double* p)`):\n" + log;
    }
    size_t sz = 0;
    if (h.GetCodeSize(prog, &sz) != 0 || sz == 0) { h.DestroyProgram(&prog); return "hiprtcGetCodeSize failed"; }
    code.resize(sz);
    const int rg = h.GetCode(prog, code.data());
    h.DestroyProgram(&prog);
    if (rg != 0) return "hiprtcGetCode failed";
    return "";
}

}  // namespace BPM_VARIANT_NS
}  // namespace bpm
