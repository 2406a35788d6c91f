// A caller's ln_like_fn given as HIP SOURCE (samplers.py:36-43 takes any Python callable): compiled at run time with hiprtc into ONE
// small kernel -- a thread per proposal row calls the caller's `ln_like` -- that runs between the library's own proposal and commit kernels
// (phase_propose_kernel / phase_commit_kernel, the kernels of the host-callback path).  Nothing leaves the device and no host code runs inside a
// generation: bpm_step drives such a sampler like one with a shipped target.  The update kernels themselves are not recompiled.
//
// What the caller writes (HIP device code; double precision; no includes needed):
//     __device__ double ln_like(const double* x, int d, const double* p)      // x: one parameter vector, p: the caller's parameter block
// (the device math functions, INFINITY, NAN and M_PI are there; a prior outside its support returns -INFINITY like a Python ln_like_fn would)
// The PER-COORDINATE form (optional; for likelihoods that are a function of a few sums over the coordinates): the source says
//     #define BPM_LN_LIKE_TERMS K                                                               // number of accumulators, <= 8
//     __device__ void ln_like_terms(double xj, int j, int d, const double* p, double* acc)       // adds coordinate j's contribution into acc[0 .. K)
//     __device__ double ln_like_finish(const double* acc, int d, const double* p)                // the value from the K sums
// and the library derives ln_like from them (user_ln_like_from_terms).  Inside the update kernel every lane of a chain then adds the terms of ITS
// coordinates and the sums meet in the kernel's own reduction tree -- the shape of the shipped targets -- where the plain form runs on one lane per chain.
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
    int (*AddNameExpression)(void*, const char*) = nullptr;
    int (*GetLoweredName)(void*, const char*, const char**) = nullptr;
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
    BPM_RTC_SYM(GetCodeSize) BPM_RTC_SYM(GetCode) BPM_RTC_SYM(DestroyProgram) BPM_RTC_SYM(AddNameExpression) BPM_RTC_SYM(GetLoweredName)
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

// ln_like for a source in the per-coordinate form (appended behind the caller's source in every program)
inline const char* user_ln_like_from_terms() {
    return "\n#ifdef BPM_LN_LIKE_TERMS\n"
           "__device__ double ln_like(const double* x, int d, const double* p) {\n"
           "    double acc[BPM_LN_LIKE_TERMS];\n"
           "    for (int k = 0; k < BPM_LN_LIKE_TERMS; ++k) acc[k] = 0.0;\n"
           "    for (int j = 0; j < d; ++j) ln_like_terms(x[j], j, d, p, acc);\n"
           "    return ln_like_finish(acc, d, p);\n"
           "}\n#endif\n";
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
    const std::string src = prelude + user_src + user_ln_like_from_terms() + user_eval_wrapper();
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

// ---- the caller's likelihood INSIDE the update kernel ---------------------------------------------------------------------------------------------
// The second, faster form: the library's own update kernel (kernels.h: phase_fused_kernel, the general instantiation) compiled at run time with the
// caller's function as its target -- one launch per half generation instead of three.  Target<TARGET_USER>::eval: the lanes of a chain put their
// coordinates of the row into LDS, the chain's first lane calls ln_like on it, the value goes back to the chain's lanes.  kernels.h / philox.h travel
// inside the library as string literals (embedded_src.h, written by the Makefile).  The kernel-argument block must be the library's own: the program
// is compiled with the library's BPM_TEST_HOOKS setting and exports sizeof(PhaseArgs) for the caller to compare.
// Two instantiations: the general one (HOT 0) and the steady-state one (`hot`: 1 with update records, 2 without -- what phase_args_hot(a, dream, with_plan,
// false) fixes is a compile-time constant) -- name_expr[0 / 1]; name_expr[2]: eval_ll_kernel with the same target; name_expr[3]: DREAM's burn-in instantiation (hot + 2).
// `ns`: the inline namespace the program's device code lives in -- unique per module of the process: the library's queue finds kernels by name.
inline std::string user_fused_program(const std::string& user_src, const std::string& ns, int algo, int lpc, int dpl, int np, uint32_t dim, bool test_hooks,
                                      int hot, std::string name_expr[4]) {
    std::string s;
    s += "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long uint64_t;\n"
         "typedef signed char int8_t; typedef short int16_t; typedef int int32_t; typedef long int64_t;\n"
         "#ifndef INFINITY\n#define INFINITY (__builtin_huge_val())\n#endif\n#ifndef NAN\n#define NAN (__builtin_nan(\"\"))\n#endif\n"
         "#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n"
         "#define BPM_VARIANT_NS " + ns + "\n";
    if (test_hooks) s += "#define BPM_TEST_HOOKS 1\n";
    s += "#define BPM_USER_LDP " + std::to_string((int)(dim | 1u)) + "\n";
    s += "#define BPM_USER_DIM " + std::to_string((int)dim) + "\n";      // (the caller's loops over d get a compile-time trip count: the sampler's dimension is fixed)
    s += "#include \"kernels.h\"\n#line 1 \"ln_like.hip\"\n" + user_src + "\n" + user_ln_like_from_terms();
    s += "namespace bpm { inline namespace BPM_VARIANT_NS {\n"
         "constexpr int TARGET_USER = 64;\n"
         "template <int LPC, int DPL>\n"
         "struct Target<TARGET_USER, LPC, DPL> {\n"
         "    struct Consts { const double* tp; };\n"
         "    static __device__ __forceinline__ Consts load(int, uint32_t, const double* tp) { Consts k; k.tp = tp; return k; }\n"
         "    static __device__ __forceinline__ double eval(const double* v, int q, uint32_t dim, const Consts& k) {\n"
         "#ifdef BPM_LN_LIKE_TERMS\n"
         "        // the per-coordinate form: every lane adds the terms of its own coordinates, the kernel's reduction tree adds the lanes\n"
         "        double acc[BPM_LN_LIKE_TERMS];\n"
         "#pragma unroll\n"
         "        for (int t = 0; t < BPM_LN_LIKE_TERMS; ++t) acc[t] = 0.0;\n"
         "#pragma unroll\n"
         "        for (int s = 0; s < DPL; ++s) {\n"
         "            const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (uint32_t)(s & 1);\n"
         "            if (j < dim) ::ln_like_terms(v[s], (int)j, BPM_USER_DIM, k.tp, acc);\n"
         "        }\n"
         "#pragma unroll\n"
         "        for (int t = 0; t < BPM_LN_LIKE_TERMS; ++t) acc[t] = gsum<LPC>(acc[t]);\n"
         "        return (double)::ln_like_finish(acc, BPM_USER_DIM, k.tp);\n"
         "#else\n"
         "        if (LPC == 1) return (double)::ln_like(v, BPM_USER_DIM, k.tp);      // (a lane is a chain: the row is the lane's registers)\n"
         "        __shared__ double rows[(block_for_hot(LPC, 3, DPL) / LPC) * BPM_USER_LDP];      // (the burn-in flavours' workgroups hold the most chains)\n"
         "        const int cw = (int)threadIdx.x / LPC;\n"
         "        double* row = rows + cw * BPM_USER_LDP;\n"
         "#pragma unroll\n"
         "        for (int s = 0; s < DPL; ++s) {\n"
         "            const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (uint32_t)(s & 1);\n"
         "            if (j < dim) row[j] = v[s];\n"
         "        }\n"
         "        __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\");\n"
         "        __builtin_amdgcn_wave_barrier();\n"
         "        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n"
         "        double r = 0.0;\n"
         "        if (q == 0) r = (double)::ln_like(row, BPM_USER_DIM, k.tp);\n"
         "        if (LPC == WAVE) r = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r)), __builtin_amdgcn_readfirstlane(__double2loint(r)));\n"
         "        else if (LPC > 1) r = __shfl(r, (((int)threadIdx.x & (WAVE - 1)) / LPC) * LPC);\n"
         "        __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\");\n"
         "        __builtin_amdgcn_wave_barrier();\n"
         "        return r;\n"
         "#endif\n"
         "    }\n"
         "};\n";
    // [0] general, [1] steady state (HOT 1 / 2), [3] DREAM's burn-in (HOT 3 / 4: sums level 1 of the CR reduction itself, folds the previous generation's sums)
    for (int k = 0; k < 3; ++k) {
        const int h = k == 0 ? 0 : (k == 1 ? hot : hot + 2);
        if (k == 2 && algo != 1 /* ALGO_DREAM */) { name_expr[3].clear(); continue; }
        const std::string inst = "phase_fused_kernel<" + std::to_string(algo) + ", TARGET_USER, " + std::to_string(lpc) + ", " + std::to_string(dpl) + ", " +
                                 std::to_string(np) + ", " + std::to_string(h) + ">";
        s += "template __global__ void " + inst + "(const PhaseArgs);\n";
        std::string& e = name_expr[k == 2 ? 3 : k];
        e = "bpm::" + inst;
        e.replace(e.find("TARGET_USER"), 11, "bpm::TARGET_USER");
    }
    // ... and the library's evaluation kernel with the same target: the ln-likes of given states by exactly the arithmetic the update kernel uses
    {
        const std::string inst = "eval_ll_kernel<TARGET_USER, " + std::to_string(lpc) + ", " + std::to_string(dpl) + ">";
        s += "template __global__ void " + inst + "(const double*, uint32_t, uint32_t, uint32_t, const double*, double*);\n";
        name_expr[2] = "bpm::" + inst;
        name_expr[2].replace(name_expr[2].find("TARGET_USER"), 11, "bpm::TARGET_USER");
    }
    s += "extern \"C\" __global__ void bpm_user_sizeof(unsigned int* out) { out[0] = (unsigned int)sizeof(PhaseArgs); out[1] = (unsigned int)block_for(" +
         std::to_string(lpc) + "); out[2] = (unsigned int)block_for_hot(" + std::to_string(lpc) + ", 3, " + std::to_string(dpl) + "); }\n"
         "}}\n";
    return s;
}
// -> "" with `code` and the kernel's lowered (mangled) name, or the reason
inline std::string compile_user_fused(Hiprtc& h, const std::string& user_src, const std::string& ns, const std::string& arch, const char* kernels_h,
                                      const char* philox_h, int algo, int lpc, int dpl, int np, uint32_t dim, bool test_hooks, int hot, std::vector<char>& code, std::string lowered[4]) {
    const std::string why = load_hiprtc(h);
    if (!why.empty()) return why;
    std::string expr[4];
    const std::string src = user_fused_program(user_src, ns, algo, lpc, dpl, np, dim, test_hooks, hot, expr);
    const char* hdr_src[] = {kernels_h, philox_h};
    const char* hdr_names[] = {"kernels.h", "philox.h"};
    void* prog = nullptr;
    if (h.CreateProgram(&prog, src.c_str(), "bpm_user_fused.hip", 2, hdr_src, hdr_names) != 0 || !prog) return "hiprtcCreateProgram failed";
    for (int k = 0; k < 4; ++k)
        if (!expr[k].empty() && h.AddNameExpression(prog, expr[k].c_str()) != 0) { h.DestroyProgram(&prog); return "hiprtcAddNameExpression failed"; }
    const std::string a = "--offload-arch=" + arch;
    const char* opts[] = {a.c_str(), "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function"};
    const int rc = h.CompileProgram(prog, 5, opts);
    std::string log;
    size_t n = 0;
    if (h.GetProgramLogSize(prog, &n) == 0 && n > 1) {
        log.resize(n);
        if (h.GetProgramLog(prog, &log[0]) != 0) log.clear();
        while (!log.empty() && (log.back() == '\0' || log.back() == '\n')) log.pop_back();
    }
    if (rc != 0) { h.DestroyProgram(&prog); return "the update kernel does not compile around this likelihood:\n" + log; }
    for (int k = 0; k < 4; ++k) {
        const char* low = nullptr;
        if (expr[k].empty()) { lowered[k].clear(); continue; }
        if (h.GetLoweredName(prog, expr[k].c_str(), &low) != 0 || !low) { h.DestroyProgram(&prog); return "hiprtcGetLoweredName failed for " + expr[k]; }
        lowered[k] = low;
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
