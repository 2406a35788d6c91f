// Host side of libbipymc_hip.so: the sampler object behind the C ABI of
// include/bipymc_hip.h.  It owns the device state (replicated chain-state matrix,
// log-like cache, history, Welford moments, CR statistics) and runs the generation
// loop of bipymc/demc.py:63-151 as back-to-back dispatches on a user-mode AQL queue of
// the library's own (aql_queue.h) or on one HIP stream.  For world_size > 1 the two
// MPI_Allgathers per generation (demc.py:93-94,116-117) are replaced by the PUSH
// exchange -- the owner of a chain stores an accepted row into every other rank's
// replica from inside the update kernel (IPC-mapped buffers), a one-wavefront kernel
// orders the ranks -- or, as the fallback, by in-place RCCL all-gathers on the stream:
// of one accept byte per chain followed by a replay of the accepted proposals, of
// packed accepted rows, or of whole rank blocks.
#include <dlfcn.h>
#include <sys/mman.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstddef>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bipymc_hip.h"
#include "kernels.h"
#include "kernels_wide.h"
#ifdef BPM_TEST_HOOKS
#include "rocrand_check.h"
#endif
#include "aql_queue.h"
#include "user_likelihood.h"
#include "embedded_src.h"

using namespace bpm;

// The ONE environment variable that changes which kernels / launch shapes the library uses: BPM_TEST_PATHS, a comma-separated list read
// once per process, for the tests that pin every alternative kernel path to the default one bit for bit
// (tests/test_gpu_api.py::test_alternative_kernel_paths_on_one_gpu) and for the timing tools:
//   mode1        one work item per LOCAL chain filtered by its position (what a rank of a world without sorted records launches)
//   noplan       header block and partner ids drawn in the update kernel (what > 16384 chains per GPU use) instead of plan records
//   noperm       the shuffle bijection walked in the kernel instead of looked up
//   planall      plan records whatever the number of chains
//   nohot        the general kernel instantiation instead of the specialised ones
//   groupqueues  every rank of a local group on an AQL queue of its own (the ranks' barrier kernels wait for each other across queues)
//   wt8          rows written through with two 8-byte agent-scope atomic stores per lane (round 2's form) instead of one 16-byte sc1 store
//   ctrlarena    the push exchange's control block inside the coarse-grained arena instead of a fine-grained allocation of its own
//   serial       the emulated ranks of a local group take turns on the GPU (tools/emulate_ranks.py)
//   hosttiming   host nanoseconds spent preparing generations and inside launch calls, printed by bpm_destroy
// Operational switches (documented in README.md): BPM_DIRECT_QUEUE=0, BPM_QUEUE_INFLIGHT, BPM_QUEUE_TIMEOUT_S, BPM_EXCHANGE, BPM_VERBOSE.
// The PRODUCT library does not read it: BPM_TEST_PATHS, the bpm_debug_* / bpm_selftest_* entry points and the BPM_FAKE_* timing builds exist
// only in build_variants/libbipymc_test.so (-DBPM_TEST_HOOKS, declared in include/bipymc_hip_test.h); the GPU tests that need them load
// that variant, the parity tests run against the product library.
#ifdef BPM_TEST_HOOKS
static bool test_path(const char* name) {
    static const std::string all = [] { const char* e = getenv("BPM_TEST_PATHS"); return std::string(",") + (e ? e : "") + ","; }();
    return all.find(std::string(",") + name + ",") != std::string::npos;
}
#else
static constexpr bool test_path(const char*) { return false; }
#endif

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return 1;
}
#define HIPCK(expr)                                                                                    \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                        std::to_string(__LINE__) + ")");                                               \
    } while (0)
#define CK(expr)                 \
    do {                         \
        int _r = (expr);         \
        if (_r != 0) return _r;  \
    } while (0)

// Wait for a stream by POLLING it (up to spin_ms, then a blocking wait).  hipStreamSynchronize parks the calling thread; on
// this class of host the core then drops to its base clock and the launch calls that follow take ~4.7 us instead of ~2.4 us
// for the first few hundred microseconds -- longer than the 6 us the GPU needs per update kernel, so a short bpm_step call
// issued right after a wait ran HOST-paced (profiles/r02_host_launch_pacing.txt).  A polling wait keeps the core clocked up.
// (50 ms of polling cover the short calls this is for -- a 1000-generation step is 11 ms; longer waits park the thread instead of burning a
// core per rank: ADVICE r02)
static int wait_stream(hipStream_t st, double spin_ms = 50.0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > spin_ms) break;
    }
    HIPCK(hipStreamSynchronize(st));
    return 0;
}

// ---- RCCL, loaded on demand (single-GPU use needs no communicator library) -------
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static Rccl g_rccl;
static int load_rccl() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return fail(std::string("cannot load RCCL: ") + dlerror());
#define SYM(f)                                                          \
    g_rccl.f = reinterpret_cast<decltype(g_rccl.f)>(dlsym(lib, "nccl" #f)); \
    if (!g_rccl.f) return fail("RCCL symbol nccl" #f " missing");
    SYM(GetUniqueId) SYM(CommInitRank) SYM(AllGather) SYM(AllReduce) SYM(CommDestroy) SYM(GetErrorString)
#undef SYM
    g_rccl.lib = lib;
    return 0;
}
#define NCCLCK(expr)                                                                              \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) return fail(std::string(#expr) + " failed: " + g_rccl.GetErrorString(_r)); \
    } while (0)

// ---- kernel dispatch ---------------------------------------------------------------
// bpm_step_timed: the NEXT update-kernel launch of this thread carries this event as its stop event (hipExtLaunchKernelGGL
// binds it to the dispatch itself: its time stamp is the kernel's end, and no marker packet enters the queue)
static thread_local hipEvent_t g_stop_event = nullptr;
static thread_local int64_t g_timed_launches = 0;
static inline hipEvent_t take_stop_event() {
    hipEvent_t e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}
typedef void (*PhaseLaunch)(const PhaseArgs&, hipStream_t);
typedef void (*EvalLaunch)(const double*, uint32_t, uint32_t, uint32_t, const double*, double*, hipStream_t);

// update launches of the generation being run that took a flavour which writes level 1 of the CR reduction ITSELF (kernels.h: CRP): what
// finish_generation holds against the host's prediction (gen_cr_inkernel) before it lets cr_final_kernel fold the partial sums
static thread_local int g_crp_launched = 0;
// ... and those that were handed a pending fold of the previous generation's sums AND took a flavour that folds (PhaseArgs::cr_fold_part; bpm_sampler::cr_pending)
static thread_local int g_crfold_launched = 0;
static thread_local bool g_call_last_gen = false;      // run_generations: the generation being run is the last one of the bpm_step call (its fold is dispatched)
static inline uint32_t grid_for(uint32_t n_items, int lpc) {
    const uint32_t cpw = (uint32_t)(block_for(lpc) / lpc);
    return (n_items + cpw - 1) / cpw;
}
#ifdef BPM_PRELOAD
// The kernel-argument block of phase_fused_kernel exactly as the device reads it (natural alignment = the kernarg layout): the
// launch goes through hipExtModuleLaunchKernel with this buffer -- no per-launch symbol lookup, no per-argument marshalling.
struct FusedKernarg {
    const uint32_t* pl_plan;
    uint32_t pl_upd_off, pl_n_items, pl_mode, _pad;
    PhaseArgs a;
};
static_assert(offsetof(FusedKernarg, a) == 24 && sizeof(FusedKernarg) == 24 + sizeof(PhaseArgs), "kernarg layout of phase_fused_kernel");
// Direct mode (aql_queue.h): while run_generations has the sampler in direct mode, g_dq names its queue and every kernel of the
// generation loop -- update kernels here, table builds in build_window -- is dispatched by a packet this library writes itself.
static thread_local bpm::DirectQueue* g_dq = nullptr;
static thread_local int g_dq_sig = -1;            // the next update dispatch carries this timing signal (bpm_step_timed)
static thread_local bool g_dq_error = false;      // a dispatch could not be made: run_generations reports it
// Fences of an update-kernel packet.  A HIP stream puts agent-scope acquire + release around every kernel; the release (write-back of
// every XCD's L2 at the end of the kernel) is 0.6 us of a 6 us launch period at cfg2.  On the library's own queue the update kernel instead
// sends what later kernels read -- accepted rows, ln-like, accept counters, during CR adaptation the Welford rows and the CR slots --
// through write-through stores (PhaseArgs::wt, kernels.h: store_row_wt16) and its packet carries the acquire only; history rows
// (non-temporal stores, read by nobody before the drain) are written back by the fenced empty kernel DirectQueue::drain puts behind
// such packets.  Round 2 did this in the steady state up to 4 MiB per half generation only (two 8-byte atomic stores per lane cost
// more than the release beyond that, and in burn-in); with one 16-byte store per lane it wins everywhere (profiles/r03_write_through_16B.txt).
// bpm_set_launch_path(h, 1, 3): acquire + release on every packet with plain stores.  The packet after a table build or after entering
// direct mode always acquires.
static thread_local int g_dq_update_fence = bpm::DirectQueue::FENCED;
static thread_local bool g_dq_call_last_gen = false;      // run_generations: this is the last generation of the bpm_step call ...
static thread_local bool g_dq_release_this = false;       // ... whose last update dispatch carries the release: the drain that usually
                                                           // follows the call then needs no fence kernel (2.5 us of a short window)
static thread_local bool g_wt_stores = false;             // this generation's update kernels store through (PhaseArgs::wt)
static thread_local bool g_dq_need_acquire = false;
static thread_local int64_t g_n_direct = 0, g_n_stream = 0;   // update-kernel dispatches of this thread by path (bpm_get_launch_stats)
// A local group whose ranks each have a queue of their own (BPM_TEST_PATHS=groupqueues) runs in direct mode rank by rank: g_dq is
// re-bound to the rank a piece of work belongs to (bind_rank_queue); the ranks' kernels then wait for each other ACROSS queues the way
// the ranks of a multi-GPU world do.
static thread_local bool g_group_direct = false;
template <class K>
static inline void launch_packed(K kernel, hipFunction_t& fn, const PhaseArgs& a, unsigned grid, unsigned block, hipStream_t s) {
    if (g_dq) {
        FusedKernarg ka;
        ka.pl_plan = a.rec_tab; ka.pl_upd_off = a.rec_off; ka.pl_n_items = a.n_items; ka.pl_mode = a.mode; ka._pad = 0u;
        ka.a = a;
        const bpm::DqKernel* k = g_dq->kernel(reinterpret_cast<const void*>(kernel));
        const int sig = g_dq_sig; g_dq_sig = -1;
        int fence = g_dq_update_fence;
        if (g_dq_need_acquire) { fence |= bpm::DirectQueue::ACQUIRE; g_dq_need_acquire = false; }
        if (g_dq_release_this) { fence |= bpm::DirectQueue::RELEASE; g_dq_release_this = false; }
        if (!k || g_dq->launch(*k, grid, 1, block, &ka, sizeof(ka), fence, sig) != 0) g_dq_error = true;
        ++g_timed_launches; ++g_n_direct;
        return;
    }
    ++g_n_stream;
    if (!fn) { if (hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(kernel)) != hipSuccess) { (void)hipGetLastError(); fn = nullptr; } }
    if (fn) {
        FusedKernarg ka;
        ka.pl_plan = a.rec_tab; ka.pl_upd_off = a.rec_off; ka.pl_n_items = a.n_items; ka.pl_mode = a.mode; ka._pad = 0u;
        ka.a = a;
        size_t sz = sizeof(ka);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        (void)hipExtModuleLaunchKernel(fn, grid * block, 1, 1, block, 1, 1, 0, s, nullptr, extra, nullptr, take_stop_event(), 0);
    } else {
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, nullptr, take_stop_event(), 0, a.rec_tab, a.rec_off, a.n_items, a.mode, a);
    }
    ++g_timed_launches;
}
template <int ALGO, int T, int NP, int LPC, int DPL, int HOT>
static void launch_hot(const PhaseArgs& a, hipStream_t s) {
    static hipFunction_t fn = nullptr;
    if (ALGO == ALGO_DREAM && hot_is_adapt(HOT) && crp_shape(LPC, DPL)) ++g_crp_launched;
    if (ALGO == ALGO_DREAM && hot_is_adapt(HOT) && crp_shape(LPC, DPL) && a.cr_fold_part != nullptr) ++g_crfold_launched;
    constexpr unsigned blk = (unsigned)block_for_hot(LPC, HOT, DPL), cpw = blk / (unsigned)LPC;      // (burn-in flavours of one wavefront per chain: 16 chains per workgroup)
    launch_packed(phase_fused_kernel<ALGO, T, LPC, DPL, NP, HOT>, fn, a, (a.n_items + cpw - 1u) / cpw, blk, s);
}
#endif
template <int ALGO, int T, int NP, int LPC, int DPL>
static void launch_fused(const PhaseArgs& a, hipStream_t s) {
#ifdef BPM_PRELOAD
    {   // frequent cases take a specialised instantiation (kernels.h: HOT): steady state, DREAM's burn-in, a rank of a world
        constexpr bool CAN_PLAN = (LPC == WAVE && DPL == 2);               // the shape that can read plan records
        const bool wp = CAN_PLAN && a.rec_tab != nullptr;
        constexpr bool DREAM_ = ALGO == ALGO_DREAM;
        static const bool no_hot = test_path("nohot");
        if (!no_hot && phase_args_hot(a, DREAM_, wp, false)) {
            if (wp) launch_hot<ALGO, T, NP, LPC, DPL, (CAN_PLAN ? 1 : 2)>(a, s); else launch_hot<ALGO, T, NP, LPC, DPL, 2>(a, s);
            return;
        }
        if (!no_hot && phase_args_hot_sharded(a, DREAM_, wp)) {
            if (wp) launch_hot<ALGO, T, NP, LPC, DPL, (CAN_PLAN ? 5 : 6)>(a, s); else launch_hot<ALGO, T, NP, LPC, DPL, 6>(a, s);
            return;
        }
        if (!no_hot && DREAM_ && phase_args_hot(a, true, wp, true)) {
            if (wp) launch_hot<ALGO, T, NP, LPC, DPL, (DREAM_ ? (CAN_PLAN ? 3 : 4) : 2)>(a, s); else launch_hot<ALGO, T, NP, LPC, DPL, (DREAM_ ? 4 : 2)>(a, s);
            return;
        }
    }
    {   // the general instantiation: same argument block, same two launch paths
        static hipFunction_t fn = nullptr;
        launch_packed(phase_fused_kernel<ALGO, T, LPC, DPL, NP>, fn, a, grid_for(a.n_items, LPC), (unsigned)block_for(LPC), s);
        return;
    }
#else
    hipExtLaunchKernelGGL((phase_fused_kernel<ALGO, T, LPC, DPL, NP>), dim3(grid_for(a.n_items, LPC)), dim3(block_for(LPC)), 0, s,
                          nullptr, take_stop_event(), 0, a);
#endif
    ++g_timed_launches;
}
template <int ALGO, int LPC, int DPL>
static void launch_propose(const PhaseArgs& a, hipStream_t s) {
    // the usual pair counts as compile-time constants (DREAM's default 3, DE-MC's 1): the partner ids stay in registers (kernels.h: phase_propose_kernel)
    constexpr int NPC = ALGO == ALGO_DREAM ? 3 : 1;
    if (a.mode != 2u && a.P == (uint32_t)NPC && !(ALGO != ALGO_DREAM && a.p_snooker > 0.0))
        hipLaunchKernelGGL((phase_propose_kernel<ALGO, LPC, DPL, NPC>), dim3(grid_for(a.n_items, LPC)), dim3(block_for(LPC)), 0, s, a);
    else
        hipLaunchKernelGGL((phase_propose_kernel<ALGO, LPC, DPL>), dim3(grid_for(a.n_items, LPC)), dim3(block_for(LPC)), 0, s, a);
}
template <int ALGO, int LPC, int DPL>
static void launch_commit(const PhaseArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((phase_commit_kernel<ALGO, LPC, DPL>), dim3(grid_for(a.n_items, LPC)), dim3(block_for(LPC)), 0, s, a);
}
template <int T, int LPC, int DPL>
static void launch_eval(const double* X, uint32_t n, uint32_t ld, uint32_t dim, const double* tp, double* out,
                        hipStream_t s) {
    hipLaunchKernelGGL((eval_ll_kernel<T, LPC, DPL>), dim3(grid_for(n, LPC)), dim3(block_for(LPC)), 0, s, X, n, ld, dim, tp, out);
}

struct Shape {
    int lpc, dpl, idx;
};
// lanes per chain / coordinates per lane for a row of ld doubles (ld even)
static bool pick_shape(uint32_t ld, Shape& sh) {
    const uint32_t np = ld / 2;
    if (np <= 1) sh = {1, 2, 0};
    else if (np <= 4) sh = {4, 2, 1};
    else if (np <= 16) sh = {16, 2, 2};
    else if (np <= 64) sh = {64, 2, 3};   // (two / four chains per wavefront, 32x4 and 16x8, measured no faster: DESIGN.md)
    else if (np <= 128) sh = {64, 4, 4};
    else if (np <= 256) sh = {64, 8, 5};
    // d > 512: one wavefront per chain LOOPING over the row (kernels_wide.h) -- no dimension limit, no scratch.  (Round 3 had 16 and 32 coordinates
    // per lane for d <= 1024 / 2048: the looped kernel is faster than the first -- 0.71 / 0.74 / 0.76 of the HBM roof at d = 640 / 1000 / 1024 against
    // 0.56 / 0.71 / 0.74 -- and replaces the second, which spilled: profiles/r04_wide_rows.txt.)
    else sh = {64, 0, 6};
    return true;
}
constexpr int SHAPE_WIDE = 6;
// [shape]: the register-resident kernels by (lanes per chain, coordinates per lane), then the looped wide-row kernel
#define SHAPE_TABLE(FN, WIDE, ...)                                                               \
    {FN<__VA_ARGS__ 1, 2>, FN<__VA_ARGS__ 4, 2>, FN<__VA_ARGS__ 16, 2>, FN<__VA_ARGS__ 64, 2>, \
     FN<__VA_ARGS__ 64, 4>, FN<__VA_ARGS__ 64, 8>, WIDE}
#define COMMA ,
// ---- the looped wide-row kernels (kernels_wide.h): same argument block and launch paths as phase_fused_kernel
template <int ALGO, int T, int NP>
static void launch_wide(const PhaseArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.n_items + (BPM_BLOCK_WAVE / WAVE) - 1) / (BPM_BLOCK_WAVE / WAVE));
#ifdef BPM_PRELOAD
    static hipFunction_t fn = nullptr;
    launch_packed(phase_wide_kernel<ALGO, T, NP, STAGE_FUSED>, fn, a, grid, (unsigned)BPM_BLOCK_WAVE, s);
#else
    hipExtLaunchKernelGGL((phase_wide_kernel<ALGO, T, NP, STAGE_FUSED>), dim3(grid), dim3(BPM_BLOCK_WAVE), 0, s, nullptr, take_stop_event(), 0, a);
    ++g_timed_launches;
#endif
}
template <int ALGO>
static void launch_wide_propose(const PhaseArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.n_items + (BPM_BLOCK_WAVE / WAVE) - 1) / (BPM_BLOCK_WAVE / WAVE));
#ifdef BPM_PRELOAD
    hipLaunchKernelGGL((phase_wide_kernel<ALGO, TARGET_HOST, 0, STAGE_PROPOSE>), dim3(grid), dim3(BPM_BLOCK_WAVE), 0, s, a.rec_tab, a.rec_off, a.n_items, a.mode, a);
#else
    hipLaunchKernelGGL((phase_wide_kernel<ALGO, TARGET_HOST, 0, STAGE_PROPOSE>), dim3(grid), dim3(BPM_BLOCK_WAVE), 0, s, a);
#endif
}
template <int ALGO>
static void launch_wide_commit(const PhaseArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.n_items + (BPM_BLOCK_WAVE / WAVE) - 1) / (BPM_BLOCK_WAVE / WAVE));
    hipLaunchKernelGGL((phase_wide_commit_kernel<ALGO>), dim3(grid), dim3(BPM_BLOCK_WAVE), 0, s, a);
}
template <int T>
static void launch_eval_wide(const double* X, uint32_t n, uint32_t ld, uint32_t dim, const double* tp, double* out, hipStream_t s) {
    hipLaunchKernelGGL((eval_ll_wide_kernel<T>), dim3((n + (BPM_BLOCK_WAVE / WAVE) - 1) / (BPM_BLOCK_WAVE / WAVE)), dim3(BPM_BLOCK_WAVE), 0, s, X, n, ld, dim, tp, out);
}
// update-kernel variants: [DE-MC (1 pair) | DREAM del_pairs = 3 (compile-time) | DREAM any del_pairs][shape]
static PhaseLaunch g_fused_gauss[3][7] = {SHAPE_TABLE(launch_fused, launch_wide<ALGO_DEMC COMMA TARGET_GAUSS COMMA 1>, ALGO_DEMC COMMA TARGET_GAUSS COMMA 1 COMMA),
                                          SHAPE_TABLE(launch_fused, launch_wide<ALGO_DREAM COMMA TARGET_GAUSS COMMA 3>, ALGO_DREAM COMMA TARGET_GAUSS COMMA 3 COMMA),
                                          SHAPE_TABLE(launch_fused, launch_wide<ALGO_DREAM COMMA TARGET_GAUSS COMMA 0>, ALGO_DREAM COMMA TARGET_GAUSS COMMA 0 COMMA)};
static PhaseLaunch g_fused_mixture[3][7] = {SHAPE_TABLE(launch_fused, launch_wide<ALGO_DEMC COMMA TARGET_MIXTURE COMMA 1>, ALGO_DEMC COMMA TARGET_MIXTURE COMMA 1 COMMA),
                                            SHAPE_TABLE(launch_fused, launch_wide<ALGO_DREAM COMMA TARGET_MIXTURE COMMA 3>, ALGO_DREAM COMMA TARGET_MIXTURE COMMA 3 COMMA),
                                            SHAPE_TABLE(launch_fused, launch_wide<ALGO_DREAM COMMA TARGET_MIXTURE COMMA 0>, ALGO_DREAM COMMA TARGET_MIXTURE COMMA 0 COMMA)};
static PhaseLaunch g_fused_banana[3] = {launch_fused<ALGO_DEMC, TARGET_BANANA, 1, 1, 2>, launch_fused<ALGO_DREAM, TARGET_BANANA, 3, 1, 2>,
                                        launch_fused<ALGO_DREAM, TARGET_BANANA, 0, 1, 2>};
static PhaseLaunch g_propose[2][7] = {SHAPE_TABLE(launch_propose, launch_wide_propose<ALGO_DEMC>, ALGO_DEMC COMMA),
                                      SHAPE_TABLE(launch_propose, launch_wide_propose<ALGO_DREAM>, ALGO_DREAM COMMA)};
template <int ALGO, int NP, int LPC, int DPL>
static void launch_replay(const PhaseArgs& a, hipStream_t s) {
    // (a workgroup per 64 positions that compacts the remote accepted chains in LDS and rebuilds only those -- 8192 wavefronts
    // launched instead of 32768 -- measured no faster: 11.5 vs 11.3 us at 8 ranks, 6.9 vs 5.2 at 2; profiles/r02_replay_variants.txt)
    hipLaunchKernelGGL((phase_replay_kernel<ALGO, LPC, DPL, NP>), dim3(grid_for(a.n_upd, LPC)), dim3(block_for(LPC)), 0, s, a);
}
template <int ALGO, int NP, int DPL>
static void launch_replay_sorted(const PhaseArgs& a, hipStream_t s) {
    uint32_t max_cnt = 0;                                     // the largest count among the OTHER ranks' updates of this half generation
    for (uint32_t r = 0; r < a.n_seg; ++r)
        if (r != a.seg_me) max_cnt = std::max(max_cnt, a.seg_off[r + 1] - a.seg_off[r]);
    if (max_cnt == 0) return;
    hipLaunchKernelGGL((phase_replay_sorted_kernel<ALGO, DPL, NP>), dim3((max_cnt + REPLAY_WG / WAVE - 1) / (REPLAY_WG / WAVE), a.n_seg), dim3(REPLAY_WG),
                       0, s, a);
}
// [DE-MC | DREAM del_pairs = 3 | DREAM any del_pairs][dims per lane 2 / 4 / 8]: owner-sorted records exist for one wavefront per chain only
#define RS_ROW(A, P) {launch_replay_sorted<A, P, 2>, launch_replay_sorted<A, P, 4>, launch_replay_sorted<A, P, 8>}
static PhaseLaunch g_replay_sorted[3][3] = {RS_ROW(ALGO_DEMC, 1), RS_ROW(ALGO_DREAM, 3), RS_ROW(ALGO_DREAM, 0)};
// [DE-MC | DREAM del_pairs = 3 | DREAM any del_pairs][shape]: the same pair-count variants as the update kernels (no target: no ln-like)
// (wide rows have no replay kernel: a world of wide-row samplers exchanges by push or by the dense all-gather -- bpm_set_exchange refuses the rest)
static PhaseLaunch g_replay[3][7] = {SHAPE_TABLE(launch_replay, nullptr, ALGO_DEMC COMMA 1 COMMA), SHAPE_TABLE(launch_replay, nullptr, ALGO_DREAM COMMA 3 COMMA),
                                     SHAPE_TABLE(launch_replay, nullptr, ALGO_DREAM COMMA 0 COMMA)};
static PhaseLaunch g_commit[2][7] = {SHAPE_TABLE(launch_commit, launch_wide_commit<ALGO_DEMC>, ALGO_DEMC COMMA),
                                     SHAPE_TABLE(launch_commit, launch_wide_commit<ALGO_DREAM>, ALGO_DREAM COMMA)};
static EvalLaunch g_eval_gauss[7] = SHAPE_TABLE(launch_eval, launch_eval_wide<TARGET_GAUSS>, TARGET_GAUSS COMMA);
static EvalLaunch g_eval_mixture[7] = SHAPE_TABLE(launch_eval, launch_eval_wide<TARGET_MIXTURE>, TARGET_MIXTURE COMMA);

// ---- the sampler ------------------------------------------------------------------------
struct bpm_sampler {
    bpm_config_t cfg{};
    std::vector<double> tparams_h;
    uint32_t N = 0, dim = 0, ld = 0, n_local = 0, lo = 0, world = 1, rank = 0;
    Shape shape{};
    Layout L{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // bpm_step_timed: ev0 rides on the first update launch of the call, ev1 on the last one
    bool timed_want_first = false;
    int64_t timed_skip = 0;           // update launches to let pass before ev0 is attached (see bpm_step_timed)
    int64_t timed_last_gen = -1;      // t_abs of the call's last generation (-1: not timing)
    int64_t timed_l0 = 0, timed_l1 = 0;   // launch counter values at those two launches
    double* G = nullptr;
    double* ll = nullptr;
    double* hist = nullptr;
    double* llhist = nullptr;
    int64_t hist_cap = 0;      // rows allocated
    // Position-ordered history append (single GPU, fewer than 64 lanes per chain): the update kernels write a generation's history row in
    // that generation's shuffle order -- consecutive work items, consecutive rows: 1.0 us per generation at cfg3, 5.5 us at cfg5 against
    // the scattered append by chain index (profiles/r03_small_d_hops_and_position_order.txt) -- and hist_tag[row] remembers how to read it:
    // -1 chain order, else (generation << 2) | (2: the row's ln-like went by chain all the same) | shuffle.  normalize_history puts rows back into chain
    // order, in place, before anything reads them by chain (bpm_get_history, the moment rebuild, a partial first generation of bpm_reduce_moments);
    // the outlier check reads them where they lie (outlier_row_keys).
    bool hist_by_pos = false;
    std::vector<int64_t> hist_tag;
    PermKey* okeys = nullptr;               // outlier check: per history row the shuffle key of its state row [0, cap) and of its ln-like [cap, 2 cap)
    size_t okeys_cap = 0;                   // (outlier_row_keys)
    std::vector<PermKey> okeys_host;
    bool okeys_x = false, okeys_ll = false; // any state row / any ln-like row of the history still lies in position order
    uint32_t* olist = nullptr;              // outlier check: [0] count, then the local chains that were reset (outlier_rebuild_kernel)
    double* hist_tmp = nullptr;    // one row (n_local * ld) + its ln-likes (n_local): staging of normalize_history
    int64_t hist_rows = 0;     // rows stored (0 when keep_history == 0 and nothing stored)
    int64_t rows_logical = 0;  // len(chain.chain) of the reference: 1 + generations since (re)initialisation
    // running population sums (cfg.running_moments): per history row g the 2 ld doubles [sum_i (x_ij - shift_j) | sum_i (x_ij - shift_j)^2]
    // over this rank's chains, shift = chain 0's state at the last (re)initialisation.  bpm_reduce_moments answers from them when
    // the history is not resident: param_est (demc.py:235-248) of 10^5 generations of config 2 without 660 GB of history.
    double* gen_sums = nullptr;
    int64_t gen_sums_cap = 0;      // rows allocated
    double* gs_shift = nullptr;    // [ld]
    double* gs_part = nullptr;     // per-block partial sums of one row
    uint32_t gs_nb = 0;
    double* w_mean = nullptr;
    double* w_m2 = nullptr;
    int64_t w_rows = 0;        // history rows folded into the Welford moments
    double* tparams = nullptr;
    double* cr_state = nullptr;       // the CURRENT totals block p_cr | delta_m | n_cr_updates: one of the two halves of cr_state_base (cr_state_alt the other)
    double* cr_state_alt = nullptr;
    double* cr_state_base = nullptr;
    // CR reduction (kernels.h): level-1 partial sums [2 MAX_CR][cr_n1] of a generation (chunks of cr_g1 positions), two buffers for the cr_mid_kernel passes
    double* cr_p1 = nullptr;          // two generations' worth: generation t writes half t & 1 (cr_p1_cur) -- the update kernel that folds generation t's sums
    double* cr_p1_cur = nullptr;      // (consumer-side fold, below) writes its own level 1 in the same launch
    // Consumer-side fold (round 5): where the update kernels sum level 1 themselves (one GPU, HOT 3 / 4) a generation's fold is NOT dispatched (cr_final_kernel:
    // one wavefront at the floor of a dependent launch, 4.5 us of cfg2's 20.6 us burn-in generation): it stays pending, and every workgroup of the NEXT
    // generation's first update launch folds the sums itself (PhaseArgs::cr_fold_part: a wavefront per array of partial sums, the same additions in the same order), workgroup 0 stores
    // the new totals into the other totals block, which the second launch and everything later read.  What cannot consume a pending fold -- the last generation
    // of a bpm_step call, the first steady-state generation, a launch that takes another flavour -- gets cr_final_kernel as before (cr_flush_pending).
    bool cr_pending = false;
    const double* cr_pend_src = nullptr;      // (the generation's level-1 sums, or what the cr_mid_kernel passes left of them: cr_pend_cnt <= CR_FINAL_MAX per array)
    uint32_t cr_pend_cnt = 0;
    bool gen_fold_planned = false;    // this generation's first update launch was given a pending fold: finish_generation checks that it took a flavour that folds
    double* cr_p2[2] = {nullptr, nullptr};
    uint32_t cr_g1 = 16, cr_n1 = 0;
    bool gen_cr_inkernel = false;     // this generation's update kernels write level 1 themselves (both launches are burn-in flavours: HOT 3 / 4)
    unsigned long long* counters = nullptr;  // device: [2] = NaN ratios
    uint32_t* acc_count = nullptr;           // device: accepted updates per local chain, this run
    int64_t gens_this_run_local = 0;
    double* prop_buf = nullptr;
    // host-callback path: pinned staging of what bpm_propose reads back and bpm_commit sends (work-item order)
    int32_t* h_ids = nullptr;
    double* h_props = nullptr;
    double* h_ll = nullptr;                  // [n_local] ln-likes by work item on their way in (bpm_commit_chunk)
    std::vector<hipEvent_t> chunk_ev;        // one event behind every chunk of the proposals' read-back (bpm_propose_begin)
    int32_t prop_chunks = 0, prop_given = 0; // chunks of the open half generation / chunks whose ln-likes have been handed in
    int prop_mode = 0;                       // how the open half generation was proposed: 1 host staging (chunks), 2 device-resident
    int64_t prop_active = 0;                 // its active work items
    bool prop_whole = false;                 // it was proposed through bpm_propose (caller-owned buffers): bpm_commit finishes it
    std::vector<uint8_t> prop_done;          // [prop_chunks] 1: the piece's ln-likes have been handed in
    // the caller's ln_like_fn as a kernel compiled from HIP source (user_likelihood.h; bpm_set_device_likelihood): bpm_step then drives a host-callback sampler
    hipModule_t user_mod = nullptr;
    hipFunction_t user_fn = nullptr;
    double* user_params = nullptr;
    // ... and the update kernel itself compiled around it (user_likelihood.h: compile_user_fused): one launch per half generation; nullptr: the three-kernel form
    hipModule_t user_fused_mod = nullptr;
    hipFunction_t user_fused_fn = nullptr;     // the general instantiation
    hipFunction_t user_fused_hot = nullptr;    // the steady-state one (HOT 1 / 2): launched when phase_args_hot(a, dream, with_plan, false) holds
    hipFunction_t user_fused_eval = nullptr;   // eval_ll_kernel with the same target: the current states' ln-likes by the update kernel's own arithmetic
    hipFunction_t user_fused_adapt = nullptr;  // DREAM's burn-in instantiation (HOT 3 / 4): level 1 of the CR reduction and the consumer-side fold inside the launch
    unsigned user_fused_block_adapt = 0;
    unsigned user_fused_block = 0;
    std::string user_fused_names[3];         // their lowered names ([2]: the burn-in instantiation): what the library's own queue dispatches them by (DirectQueue::kernel_by_name)
    bool user_fused_dq = false;              // ... and both were found among the loaded code objects
    std::string user_fused_why;              // why the fused form is not in use (bpm_get_device_likelihood_info)
    double* aux_buf = nullptr;
    int32_t* ids_buf = nullptr;
    int32_t* trace_i32 = nullptr;      // per-chain decision trace (bpm_set_trace: test variant only; always null in the product library)
    double* trace_f64 = nullptr;
    uint8_t* trace_mask = nullptr;
    double* scratch = nullptr;   // small device scratch (theta0, var, moments)
    // Per-generation tables that depend only on (seed, generation, N) -- shuffle orders, update records -- are built a WINDOW
    // of win_K generations at a time (window W = generations [W win_K, (W + 1) win_K)) into one of two buffers, one window
    // AHEAD of the update kernels: entering window W enqueues the build of W + 1.  A bpm_step call therefore never starts
    // with a table build (round 1 built the tables of a call at its head: 13 us inside a timed window of 20 generations),
    // and a rank of a world finds the launch sizes of a window on the host without a stall.
    struct TabBuf {
        uint32_t* perm = nullptr;       // [win_K * N] shuffle orders, position -> chain id
        uint32_t* inv = nullptr;        // [win_K * N] chain id -> position
        uint32_t* plan = nullptr;       // [win_K * N * PLAN_WORDS] update records (plan_kernel) or nullptr
        uint32_t* sidx = nullptr;       // world > 1: [win_K * N] slot of every position in the owner-sorted order of its generation (plan_slot_kernel);
                                        // `plan` then holds the records in THAT order (rank segment by rank segment inside each group)
        uint32_t* chunk_count = nullptr; // push exchange: per-chunk counts of this rank's positions (plan_slot_own_kernel)
        uint32_t* plan_count = nullptr; // device [win_K * 2 * world]: updates of every rank in every half generation
        uint32_t* count_h = nullptr;    // the same in pinned host memory, copied behind the build
        bool own_only = false;          // its records cover this rank's chains only (built under the push exchange)
        int64_t W = -1;                 // window held (or being built)
        int shuffle = -1;
        hipEvent_t built = nullptr;     // recorded on the build stream behind the window's last kernel / copy
    };
    TabBuf tb[2];
    int cur = -1;                       // buffer the update stream is using
    int win_K = 0;                      // generations per window (<= PERM_CHUNK)
    bool plan_on = false, sorted_on = false;   // sorted_on: world > 1 with owner-sorted records
    // the current window (aliases into tb[cur])
    uint32_t* perm_tab = nullptr;
    uint32_t* inv_tab = nullptr;
    uint32_t* plan_tab = nullptr;
    const uint32_t* plan_count_h = nullptr;
    int64_t tab_t0 = -1;
    int tab_K = 0;
    int tab_shuffle = -1;
    double* gamma_tab = nullptr;    // [dim + 1]
    double* x_next = nullptr;       // [n_local * ld] banked updates of a synchronous DE-MC generation
    unsigned long long* stamps = nullptr;   // diagnostic build (-DBPM_STAMPS) only
    size_t scratch_doubles = 0;
    ncclComm_t comm = nullptr;
    bool local_group = false;     // test mode: ranks are handles of ONE process, exchanged by device copies
    // push exchange (world > 1): the exchange buffer G, the outlier block om and the control block live in ONE
    // allocation (the arena) that the other ranks map (hipIpcOpenMemHandle); owners push accepted rows into every replica from the
    // update kernel, push_sync_kernel orders the half generations across ranks.  No collective, no replay kernel.
    void* arena = nullptr;
    size_t arena_bytes = 0, off_om = 0, off_ctrl = 0;
    PushCtrl* ctrl = nullptr;
    bool ctrl_fine = false;                 // the control block is a FINE-GRAINED allocation of its own (what a flag polled inside a kernel while
                                            // another agent writes it should live in; inside the coarse-grained arena otherwise)
    void* peer_ctrl_base[MAX_SEG] = {};     // the peers' fine-grained control blocks as this process addresses them
    bool peer_ctrl_opened[MAX_SEG] = {};
    bool push_connected = false, push_enabled = false, push_no_rccl = false;
    bool push_failed = false;      // a cross-rank wait ran into its limit: push cannot be re-enabled (bpm_set_exchange)
    bool push_agent_scope = false;          // update packets fence at agent scope instead of system scope (bpm_set_exchange(h, 3, 1))
    bool probe_sys_ok = false, probe_agent_ok = false, probe_direct = false;   // bpm_push_selftest: the arena probe under system- / agent-scope packet
                                                                               // fences, and whether it ran on the library's own queue
    void* peer_base[MAX_SEG] = {};          // every rank's arena as THIS process addresses it (own entry: arena)
    bool peer_opened[MAX_SEG] = {};         // mapped with hipIpcOpenMemHandle (to be closed)
    unsigned long long* tab_peerG = nullptr;    // device [MAX_PEERS]: G of the other ranks (PhaseArgs::peer_tab)
    unsigned long long* tab_all = nullptr;      // device [3][MAX_SEG]: G | ctrl | om of every rank by rank
    unsigned long long push_seq = 0;            // barrier sequence number (the same on every rank: lock-step call sequences)
    int64_t n_push_gens = 0;
    // sparse exchange (world > 1, outside CR adaptation): only accepted rows travel, in fixed-capacity packed blocks;
    // a chunk whose capacity was exceeded is rolled back to its checkpoint and replayed with the dense all-gather
    bool sparse_enabled = false;
    // replay exchange (the default for world > 1 outside CR adaptation): owners publish one accept byte per update, every
    // other rank recomputes the accepted proposals into its replica (phase_replay_kernel)
    bool replay_enabled = false;
    bool replay_active = false;   // the generation being prepared writes accept bytes
    bool push_active = false;     // the generation being prepared pushes its accepted rows into the other ranks' replicas
    uint8_t* accbits_all = nullptr;   // [N] accept bytes by global chain id; this rank's block at rank * n_local
    int64_t n_replay_gens = 0;
    bool sparse_active = false;   // the generation being prepared packs its accepted rows
    uint32_t xnsub = 1;           // sub-blocks per rank (a counter each; local chain li packs into sub-block li % xnsub)
    uint32_t xcap = 0, xcap_max = 0;   // capacity of a sub-block (rows per half generation, even) and its ceiling
    double* PK = nullptr;         // packed blocks: rank r at r * xnsub * xstride(), sub-block = [count | ids(cap) | rows(cap * ld)]
    uint32_t xstride() const { return 2u + xcap * (ld + 1u); }
    uint32_t* xstat = nullptr;    // [0] overflow flag, [1] largest count seen
    double* ckpt_G = nullptr;
    double* ckpt_ll = nullptr;
    uint32_t* ckpt_acc = nullptr;
    unsigned long long* ckpt_counters = nullptr;
    int64_t n_sparse_chunks = 0, n_sparse_replays = 0;
    // direct mode: the generation loop's kernels go through the library's own AQL queue (aql_queue.h) instead of the HIP stream.
    // dq_active: work may be in flight on that queue -- every entry point that uses the stream or reads device memory drains
    // it first (check_handle); run_generations waits for the stream before it enters direct mode.
    bpm::DirectQueue* dq = nullptr;
    bool dq_private = false;          // a queue of this sampler's own (rank of a local group under BPM_TEST_PATHS=groupqueues)
    bool dq_active = false;
    bool dq_enabled = true;           // bpm_set_launch_path
    bool shape_needs_scratch = false; // 1024 < d <= 2048 (16 coordinate pairs per lane): the update kernels spill to scratch memory, which the HIP
                                      // runtime provisions for ITS queues -- these samplers launch on the stream
    bool coherent = false;            // state buffers live in cached-coherent device memory (dev_alloc_state)
    int dq_fence = bpm::DirectQueue::FENCED;   // fences of the update-kernel packets (run_generations)
    bool timed_direct = false;        // the last bpm_step_timed was timed by the queue's dispatch time stamps
    // run state
    bpm_run_opts_t opts{};
    bool run_open = false;
    int64_t k_gen = 0, t_abs = 0;
    int phase = 0;               // host-callback: next half generation to propose (0/1)
    int64_t phase_a_updates = 0; // host-callback: local chains already updated in an open generation
    bool proposed = false;
    bool state_set = false;
    double* om = nullptr;        // outlier check: world x [omega (n_local) | ln_like (n_local)], all-gathered in place
    double* sel = nullptr;       // outlier check: [0..3] order statistics around Q1 / Q3, [4] first argmax of omega
    unsigned char* sel_state = nullptr;   // radix-select state between the passes (SelState)
    bool outlier_due = false;    // set by finish_generation, served by the group driver (all ranks take part)
    // per-generation cache (host-callback path keeps it between propose and commit)
    PhaseArgs cur_args[2];
    // several chains per wavefront + device target (kernels.h: lean_scalars): the update kernels neither read nor write the ln-like cache `ll`
    // (they re-evaluate it from the own row); ll_stale says that generations have run since it was last evaluated -- refresh_ll brings it up
    // to date for whoever reads it (bpm_get_loglike, the outlier check)
    bool lean = false, ll_stale = false;
#ifdef BPM_EXPERIMENT_XCD      // (round 5 experiment: a generation loop resident on one XCD, kernels.h: xcd_resident_kernel)
    PhaseArgs* xcd_args = nullptr;           // device: the argument blocks of one batch of half generations
    uint32_t* xcd_ctl = nullptr;             // device: ticket | barrier counter | workers | error | XCC ids seen
    int64_t xcd_gens = 0;                    // generations run this way
#endif
    bool gen_adapt_on = false;
    bool gen_cr_reduce = false;       // this generation's (delta, cr) slots are reduced (adaptation on AND the gate of dream.py:123 open)
};


static PhaseLaunch pick_fused(const bpm_sampler* s) {
    const int v = s->cfg.algo != BPM_ALGO_DREAM ? 0 : (s->cfg.del_pairs == 3 ? 1 : 2);
    switch (s->cfg.target_id) {
        case BPM_TARGET_GAUSS_EQUICORR: return g_fused_gauss[v][s->shape.idx];
        case BPM_TARGET_MIXTURE_PAIRS: return g_fused_mixture[v][s->shape.idx];
        case BPM_TARGET_BANANA_2D: return g_fused_banana[v];
        default: return nullptr;
    }
}

// leave direct mode: everything dispatched on the library's own queue has finished and is visible to the HIP stream and the host
static int leave_direct(bpm_sampler* s) {
    if (!s->dq_active) return 0;
    s->dq_active = false;
    if (s->dq && s->dq->drain() != 0) return fail("direct AQL queue: " + s->dq->why());
    return 0;
}
static int check_handle(bpm_handle_t h) {
    if (!h) return fail("null handle");
    return leave_direct(h);
}
// A piece of HIP-stream work in the middle of a direct-mode generation loop: queue drained before, stream drained after.
struct StreamSection {
    bpm_sampler* s;
    bool was;
    int rc;
    explicit StreamSection(bpm_sampler* s_);
    int end();
};
static int check_handle_keep_direct(bpm_handle_t h) {      // (bpm_step and friends: consecutive calls stay on the queue)
    if (!h) return fail("null handle");
    return 0;
}
StreamSection::StreamSection(bpm_sampler* s_) : s(s_), was(s_->dq_active && g_dq != nullptr), rc(0) {
    if (was) { s->dq->flush(); rc = s->dq->drain() != 0 ? fail("direct AQL queue: " + s->dq->why()) : 0; }
}
int StreamSection::end() {
    if (!was) return 0;
    CK(wait_stream(s->stream));
    g_dq_need_acquire = true;
    return 0;
}



static int set_device(bpm_sampler* s) {
    HIPCK(hipSetDevice(s->cfg.device));
    return 0;
}

template <class T>
static int dev_alloc(T** p, size_t n) {
    HIPCK(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T)));
    return 0;
}

// Buffers that one update kernel writes and the next one reads (state matrix, ln-like cache, accept counters, Welford moments): ordinary
// device memory, or -- coherent == true, ONLY in the experiment build -DBPM_EXPERIMENT_COHERENT -- the GPU's EXTENDED-SCOPE FINE-GRAINED pool
// (hipDeviceMallocUncached on this runtime): local HBM mapped with the cached-coherent memory type, i.e. the XCDs' L2s stay coherent on
// these lines by themselves (tools/micro/aql_direct.cpp: a dependent chain of kernels is correct on it with NO release fence between
// the dispatches, wrong on hipMalloc memory).  Gathers from it cost 1-5 % more than from ordinary device memory.
template <class T>
static int dev_alloc_state(T** p, size_t n, bool coherent) {
#ifdef BPM_EXPERIMENT_COHERENT
    if (coherent) {
        HIPCK(hipExtMallocWithFlags(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T), hipDeviceMallocUncached));
        return 0;
    }
#endif
    (void)coherent;
    return dev_alloc(p, n);
}

// Does memory from dev_alloc_state really stay coherent across the XCDs without a release fence?  What hipDeviceMallocUncached maps
// to is the runtime's choice (hsa_amd_pointer_info reports the same flags for it as for plain fine-grained memory, which is NOT
// coherent that way), so the property itself is tested once per device: 48 dependent dispatches with acquire-only packets hand
// every block of a 2 MB buffer from workgroup to workgroup; ordinary memory fails this in every element (tools/micro/aql_direct.cpp).
// -> number of wrong elements (0 = every launch saw its predecessor's writes), -1 = the probe could not run
#if defined(BPM_TEST_HOOKS) || defined(BPM_EXPERIMENT_COHERENT)
static long long coherence_probe(bpm::DirectQueue* dq, bool coherent_alloc) {
    constexpr uint32_t NB = 4096, L = 48;
    double* x = nullptr;
    long long wrong = -1;
    const bpm::DqKernel* k = dq->kernel(reinterpret_cast<const void*>(coherence_probe_kernel));
    if (!dq->set_fence_kernel(reinterpret_cast<const void*>(queue_fence_kernel))) k = nullptr;      // (drain() behind release-less packets)
    if (k && dev_alloc_state(&x, (size_t)NB * WAVE, coherent_alloc) == 0) {
        std::vector<double> h((size_t)NB * WAVE);
        bool run = hipMemset(x, 0, h.size() * sizeof(double)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        for (uint32_t i = 0; i < L && run; ++i) {
            struct { double* x; uint32_t nb, shift; } a{x, NB, 7u * i};
            run = dq->launch(*k, NB, 1, WAVE, &a, sizeof(a), i == 0 ? bpm::DirectQueue::FENCED : bpm::DirectQueue::ACQUIRE) == 0;
        }
        run = run && dq->drain() == 0 && hipMemcpy(h.data(), x, h.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        if (run) {
            wrong = 0;
            for (double v : h) wrong += v != (double)L;
        }
        (void)hipGetLastError();
    }
    if (x) (void)hipFree(x);
    return wrong;
}
#endif
#ifdef BPM_EXPERIMENT_COHERENT
static bool state_memory_is_coherent(bpm::DirectQueue* dq, int device) {
    static std::mutex mu;
    static std::map<int, bool> known;
    std::lock_guard<std::mutex> lk(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    const bool ok = coherence_probe(dq, true) == 0;
    known[device] = ok;
    return ok;
}
#endif

static int ensure_history(bpm_sampler* s, int64_t rows) {
    if (!s->cfg.keep_history) rows = std::min<int64_t>(rows, 1);
    if (rows <= s->hist_cap) return 0;
    const bool was_direct = s->dq_active;
    CK(leave_direct(s));              // (kernels in flight on the library's own queue still write the old buffers)
    int64_t cap = std::max<int64_t>(rows, s->hist_cap + s->hist_cap / 2);
    const size_t row_d = (size_t)s->n_local * s->ld;
    double* nh = nullptr;
    double* nl = nullptr;
    CK(dev_alloc(&nh, (size_t)cap * row_d));
    CK(dev_alloc(&nl, (size_t)cap * s->n_local));
    if (s->hist_rows > 0) {
        HIPCK(hipMemcpyAsync(nh, s->hist, (size_t)s->hist_rows * row_d * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        HIPCK(hipMemcpyAsync(nl, s->llhist, (size_t)s->hist_rows * s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    }
    HIPCK(hipStreamSynchronize(s->stream));
    if (s->hist) HIPCK(hipFree(s->hist));
    if (s->llhist) HIPCK(hipFree(s->llhist));
    s->hist = nh;
    s->llhist = nl;
    s->hist_cap = cap;
    s->hist_tag.resize((size_t)cap, -1);
    s->dq_active = was_direct && g_dq != nullptr;      // (inside a direct-mode generation: the stream is idle again, go on)
    return 0;
}

static int eval_local_ll(bpm_sampler* s) {
    // ln_like of the local chains' current state into the cache (device targets only)
    const double* Xl = s->G + (uint64_t)s->rank * s->L.blk;
    switch (s->cfg.target_id) {
        case BPM_TARGET_GAUSS_EQUICORR: g_eval_gauss[s->shape.idx](Xl, s->n_local, s->ld, s->dim, s->tparams, s->ll, s->stream); break;
        case BPM_TARGET_MIXTURE_PAIRS: g_eval_mixture[s->shape.idx](Xl, s->n_local, s->ld, s->dim, s->tparams, s->ll, s->stream); break;
        case BPM_TARGET_BANANA_2D: launch_eval<TARGET_BANANA, 1, 2>(Xl, s->n_local, s->ld, s->dim, s->tparams, s->ll, s->stream); break;
        default: return 0;  // host callback: caller supplies values via bpm_set_loglike
    }
    HIPCK(hipGetLastError());
    return 0;
}

// the ln-like cache of the local chains, current again (stream work: the caller has drained the library's own queue)
static int refresh_ll(bpm_sampler* s) {
    if (!s->ll_stale) return 0;
    CK(eval_local_ll(s));
    s->ll_stale = false;
    return 0;
}

// History rows [r0, r1) into chain order (see bpm_sampler::hist_by_pos).  On the sampler's stream: the caller has drained the library's
// own queue (check_handle / StreamSection).
static int normalize_history(bpm_sampler* s, int64_t r0, int64_t r1) {
    if (!s->hist_by_pos) return 0;
    r0 = std::max<int64_t>(r0, 0);
    r1 = std::min<int64_t>(r1, std::min<int64_t>(s->hist_rows, (int64_t)s->hist_tag.size()));
    const size_t row_d = (size_t)s->n_local * s->ld;
    const uint64_t n = (uint64_t)s->n_local * (s->ld / 2u);
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t tag = s->hist_tag[(size_t)r];
        if (tag < 0) continue;
        const PermKey key = make_perm_key(s->cfg.seed, (uint64_t)(tag >> 2), s->N, (tag & 1) != 0);
        const bool ll_by_chain = (tag & 2) != 0;
        double* row = s->hist + (uint64_t)r * row_d;
        double* llrow = s->llhist + (uint64_t)r * s->n_local;
        hipLaunchKernelGGL(hist_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, key, s->N, s->ld, (const double*)row,
                           ll_by_chain ? (const double*)nullptr : (const double*)llrow, s->hist_tmp, s->hist_tmp + row_d);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(row, s->hist_tmp, row_d * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        if (!ll_by_chain) HIPCK(hipMemcpyAsync(llrow, s->hist_tmp + row_d, (size_t)s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        s->hist_tag[(size_t)r] = -1;
    }
    return 0;
}

// ---- running population sums (cfg.running_moments) -------------------------------------------------------------------------
static int ensure_gen_sums(bpm_sampler* s, int64_t rows) {
    if (!s->cfg.running_moments || rows <= s->gen_sums_cap) return 0;
    const bool was_direct = s->dq_active;
    CK(leave_direct(s));
    const int64_t cap = std::max<int64_t>(std::max<int64_t>(rows, 1024), s->gen_sums_cap + s->gen_sums_cap / 2);
    double* n = nullptr;
    CK(dev_alloc(&n, (size_t)cap * 2 * s->ld));
    if (s->gen_sums && s->rows_logical > 0)
        HIPCK(hipMemcpyAsync(n, s->gen_sums, (size_t)std::min<int64_t>(s->rows_logical, s->gen_sums_cap) * 2 * s->ld * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    if (s->gen_sums) HIPCK(hipFree(s->gen_sums));
    s->gen_sums = n;
    s->gen_sums_cap = cap;
    s->dq_active = was_direct && g_dq != nullptr;
    return 0;
}
// the sums of history row `row` from this rank's block of the state matrix as it stands (two dispatches: per-block partial sums, a
// fixed-order fold -- the kernels of bpm_reduce_moments, hence the same bits for the same rows)
static int push_gen_sums(bpm_sampler* s, int64_t row, const double* src = nullptr) {
    if (!s->cfg.running_moments) return 0;
    const double* Xl = src ? src : s->G + (uint64_t)s->rank * s->L.blk;
    double* out = s->gen_sums + (size_t)row * 2 * s->ld;
    if (g_dq) {
        struct { const double* H; uint64_t lo, hi; uint32_t ld, _pad; const double* shift; double* part; } pa{Xl, 0, s->n_local, s->ld, 0u, s->gs_shift, s->gs_part};
        struct { const double* part; uint32_t nb, ld; double* out; } fa{s->gs_part, s->gs_nb, s->ld, out};
        const bpm::DqKernel* kp = g_dq->kernel(reinterpret_cast<const void*>(moments_partial_kernel));
        const bpm::DqKernel* kf = g_dq->kernel(reinterpret_cast<const void*>(moments_final_kernel));
        // (reads rows that update kernels wrote, possibly with agent-scope stores behind release-less packets: acquire; writes what only
        // bpm_reduce_moments reads, behind a drain: plain stores, release left to the drain's fence kernel)
        if (!kp || !kf || g_dq->launch(*kp, s->gs_nb, 1, MOM_THREADS, &pa, sizeof(pa), bpm::DirectQueue::FENCED) != 0 ||
            g_dq->launch(*kf, s->ld, 1, MOM_THREADS, &fa, sizeof(fa), bpm::DirectQueue::FENCED) != 0)
            return fail("direct AQL queue: running-moment kernels: " + g_dq->why());
        g_dq_need_acquire = true;
        return 0;
    }
    hipLaunchKernelGGL(moments_partial_kernel, dim3(s->gs_nb), dim3(MOM_THREADS), 0, s->stream, Xl, (uint64_t)0, (uint64_t)s->n_local, s->ld,
                       (const double*)s->gs_shift, s->gs_part);
    hipLaunchKernelGGL(moments_final_kernel, dim3(s->ld), dim3(MOM_THREADS), 0, s->stream, (const double*)s->gs_part, s->gs_nb, s->ld, out);
    HIPCK(hipGetLastError());
    return 0;
}

// after the state matrix was (re)initialised: history := [state], moments reset
static int reset_history(bpm_sampler* s) {
    CK(eval_local_ll(s));
    s->ll_stale = false;
    s->hist_rows = 0;
    s->rows_logical = 1;
    s->w_rows = 0;
    CK(ensure_history(s, 1));
    const size_t row_d = (size_t)s->n_local * s->ld;
    HIPCK(hipMemcpyAsync(s->hist, s->G + (uint64_t)s->rank * s->L.blk, row_d * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIPCK(hipMemcpyAsync(s->llhist, s->ll, (size_t)s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    if (!s->hist_tag.empty()) s->hist_tag[0] = -1;
    s->hist_rows = 1;
    // Welford over the single row: mean = row, m2 = 0
    // (one record per chain: [mean (ld) | m2 (ld)])
    HIPCK(hipMemsetAsync(s->w_mean, 0, 2 * row_d * sizeof(double), s->stream));
    HIPCK(hipMemcpy2DAsync(s->w_mean, (size_t)2 * s->ld * sizeof(double), s->hist, (size_t)s->ld * sizeof(double), (size_t)s->ld * sizeof(double), s->n_local,
                           hipMemcpyDeviceToDevice, s->stream));
    s->w_rows = 1;
    if (s->cfg.running_moments) {      // shift := chain 0's state; sums of row 0
        HIPCK(hipMemcpyAsync(s->gs_shift, s->G, s->ld * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        CK(ensure_gen_sums(s, 1));
        CK(push_gen_sums(s, 0));
    }
    s->state_set = true;
    s->phase = 0;
    s->proposed = false;
    return 0;
}

// BPM_TEST_PATHS=hosttiming (diagnostic): host nanoseconds spent preparing generations and inside the launch calls, printed by bpm_destroy
static bool g_host_timing = test_path("hosttiming");
static long long g_ns_prepare = 0, g_ns_launch = 0, g_n_launch = 0;
static std::vector<long long> g_launch_log;       // (start, end) of every launch call of the current bpm_step_timed
static inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The decision bpm_destroy takes about the sampler's device buffers, as a pure function: 1 = free them, 0 = leak them.  Buffers are freed unless
// the queue that may still run kernels on them failed AND could not be quiesced.  (CPU-tested through the test variant: bpm_debug_destroy_plan.)
static int destroy_plan(int queue_failed, int quiesced) { return (queue_failed != 0 && quiesced == 0) ? 0 : 1; }
extern "C" const char* bpm_last_error(void) { return g_err.c_str(); }
extern "C" int bpm_abi_version(void) { return BPM_ABI_VERSION; }
// Which sources is this binary?  The Makefile bakes in the first 16 hex digits of the SHA-256 over sampler.hip, the kernel headers, both C headers and
// the Makefile itself (ID_SRCS, in that order); bipymc_amd/_lib.py recomputes it from the tree and refuses a library built from other sources.
#ifndef BPM_BUILD_ID
#define BPM_BUILD_ID "unknown"
#endif
extern "C" const char* bpm_build_id(void) { return BPM_BUILD_ID; }
extern "C" int bpm_device_count(int32_t* out) {
    if (!out) return fail("bpm_device_count: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *out = n;
    return 0;
}

extern "C" int bpm_get_unique_id(char out[BPM_UID_BYTES]) {
    CK(load_rccl());
    ncclUniqueId id;
    NCCLCK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == BPM_UID_BYTES, "uid size");
    std::memcpy(out, &id, BPM_UID_BYTES);
    return 0;
}

extern "C" int bpm_destroy(bpm_handle_t s) {
    if (!s) return 0;
    if (g_host_timing && g_n_launch > 0) {
        fprintf(stderr, "[bpm host timing] %lld launches: %.2f us per launch call, %.2f us per generation in prepare_generation\n", g_n_launch,
                g_ns_launch * 1e-3 / g_n_launch, g_ns_prepare * 2e-3 / g_n_launch);
        g_ns_prepare = g_ns_launch = g_n_launch = 0;
    }
    (void)hipSetDevice(s->cfg.device);
    // What the library's own queue still holds must be finished -- or aborted -- before the buffers its kernels use are freed: the raw
    // HSA queue is invisible to hipStreamSynchronize and hipFree.  After a drain that ran into its time limit (a tool stalling the
    // queue: DESIGN.md section 5 "Dispatch") the kernels may be slow, not dead; DirectQueue::quiesce then inactivates the queue, and if
    // even that is refused the device buffers are deliberately LEAKED and the call reports it.
    const int drained = leave_direct(s);
    const bool queue_failed = s->dq != nullptr && (drained != 0 || s->dq->failed());
    const bool quiet = !queue_failed || s->dq->quiesce();
    const bool free_buffers = destroy_plan(queue_failed ? 1 : 0, quiet ? 1 : 0) == 1;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    // Push exchange between processes: the library orders the teardown itself (round 3 relied on the caller's barrier).  Announce "closing" in
    // every peer's control block -- a peer that still waits for this rank then reports it at once instead of sitting out its limit -- and wait,
    // bounded (BPM_PUSH_CLOSE_TIMEOUT_S, default 3), until the peers have announced the same: ranks that end a run together unmap each other's
    // arenas only after ALL of them have left their last kernels.  A peer that is late keeps what it mapped alive through its own mapping.
    // (Ranks of ONE process -- local test groups -- are destroyed one after the other by one host thread: nothing to order, and a peer's
    // control block may be gone already.)
    if (s->push_connected && s->world > 1 && !s->local_group && s->stream && s->tab_all && s->ctrl && free_buffers) {
        static const double cto = getenv("BPM_PUSH_CLOSE_TIMEOUT_S") ? atof(getenv("BPM_PUSH_CLOSE_TIMEOUT_S")) : 3.0;
        hipLaunchKernelGGL(push_close_kernel, dim3(1), dim3(WAVE), 0, s->stream, s->ctrl, (const unsigned long long*)(s->tab_all + MAX_SEG), s->world, s->rank,
                           cto > 0.0 ? 1u : 0u, (unsigned long long)(std::max(0.0, cto) * 1e8));
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s->stream);
    }
    if (s->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(s->comm);
    for (uint32_t p = 0; p < (uint32_t)MAX_SEG; ++p) {
        if (s->peer_opened[p] && s->peer_base[p]) (void)hipIpcCloseMemHandle(s->peer_base[p]);
        if (s->peer_ctrl_opened[p] && s->peer_ctrl_base[p]) (void)hipIpcCloseMemHandle(s->peer_ctrl_base[p]);
    }
    if (s->ctrl_fine && s->ctrl && free_buffers) (void)hipFree(s->ctrl);
    if (s->arena) { s->G = nullptr; s->om = nullptr; }       // (both live inside the arena)
    void* ptrs[] = {s->tb[0].chunk_count, s->tb[1].chunk_count, s->hist_tmp, s->gen_sums, s->gs_shift, s->gs_part, s->arena, s->tab_peerG, s->tab_all, s->om, s->sel, s->sel_state, s->okeys, s->olist, s->G, s->ll, s->hist, s->llhist, s->w_mean, s->tparams, s->cr_state_base, s->cr_p1, s->cr_p2[0], s->cr_p2[1], s->counters, s->acc_count,
                    s->prop_buf, s->aux_buf, s->ids_buf, s->tb[0].perm, s->tb[0].inv, s->tb[0].plan, s->tb[0].sidx, s->tb[0].plan_count,
                    s->tb[1].perm, s->tb[1].inv, s->tb[1].plan, s->tb[1].sidx, s->tb[1].plan_count, s->gamma_tab, s->x_next, s->accbits_all, s->PK, s->xstat, s->ckpt_G, s->ckpt_ll, s->ckpt_acc, s->ckpt_counters, s->trace_i32, s->trace_f64, s->trace_mask, s->scratch};
    if (free_buffers)
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    if (s->h_ids) (void)hipHostFree(s->h_ids);
    if (s->h_props) (void)hipHostFree(s->h_props);
    if (s->h_ll) (void)hipHostFree(s->h_ll);
#ifdef BPM_EXPERIMENT_XCD
    if (free_buffers) { if (s->xcd_args) (void)hipFree(s->xcd_args); if (s->xcd_ctl) (void)hipFree(s->xcd_ctl); }
#endif
    for (hipEvent_t e : s->chunk_ev) if (e) (void)hipEventDestroy(e);
    if (free_buffers) { if (s->user_mod) (void)hipModuleUnload(s->user_mod); if (s->user_fused_mod) { if (s->dq) for (const std::string& nm : s->user_fused_names) s->dq->forget_named(nm); (void)hipModuleUnload(s->user_fused_mod); } if (s->user_params) (void)hipFree(s->user_params); }
    for (auto& B : s->tb) {
        if (B.count_h) (void)hipHostFree(B.count_h);
        if (B.built) (void)hipEventDestroy(B.built);
    }
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->dq_private && free_buffers) bpm::DirectQueue::destroy_private(s->dq);
    delete s;
    if (!free_buffers)
        return fail("bpm_destroy: the library's AQL queue failed and could not be quiesced; the sampler's device buffers were leaked "
                    "rather than freed under kernels that may still run");
    return 0;
}

#ifdef BPM_TEST_HOOKS
// ---- test hooks (include/bipymc_hip_test.h; build_variants/libbipymc_test.so only) -----------------------------------------------------
extern "C" int bpm_debug_destroy_plan(int32_t queue_failed, int32_t quiesced) { return destroy_plan(queue_failed, quiesced); }

// Test hook: put the handle's queue into the state a timed-out drain leaves (refuse_quiesce != 0: and make quiesce() fail as if
// hsa_queue_inactivate had been refused).  The queue of this device is then unusable for the rest of the PROCESS -- run in a child process.
extern "C" int bpm_debug_fail_queue(bpm_handle_t s, int32_t refuse_quiesce) {
    if (!s) return fail("null handle");
    if (!s->dq) return fail("bpm_debug_fail_queue: this sampler has no queue of its own");
    s->dq->test_mark_failed(refuse_quiesce != 0);
    return 0;
}

// Test hook: no-op packets on the handle's queue until its next packet takes position `pos` (0 ... 254) of an epoch of 256 packets; *widx = the
// queue's write index afterwards.  (tests/test_gpu_api.py::test_drains_at_every_position_of_the_queues_epochs)
extern "C" int bpm_debug_queue_pad(bpm_handle_t s, int32_t pos, int64_t* widx) {
    if (!s) return fail("null handle");
    if (!s->dq) return fail("bpm_debug_queue_pad: this sampler has no queue of its own");
    if (pos < 0 || pos > 254) return fail("bpm_debug_queue_pad: pos must be 0 ... 254");
    const int64_t w = s->dq->test_pad_to((uint32_t)pos);
    if (w < 0) return fail("bpm_debug_queue_pad: " + s->dq->why());
    if (widx) *widx = w;
    return 0;
}
#endif   // BPM_TEST_HOOKS

extern "C" int bpm_create(const bpm_config_t* cfg, bpm_handle_t* out) {
    if (!cfg || !out) return fail("bpm_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != BPM_ABI_VERSION) return fail("bpm_create: ABI version mismatch");
    if (cfg->algo != BPM_ALGO_DEMC && cfg->algo != BPM_ALGO_DREAM && cfg->algo != BPM_ALGO_DEMC_SYNC) return fail("bpm_create: unknown algo");
    if (cfg->n_chains < 4) return fail("bpm_create: n_chains >= 4 required (samplers.py:249)");
    if (cfg->dim < 1) return fail("bpm_create: dim >= 1 required");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) return fail("bpm_create: bad rank/world_size");
    if (cfg->n_chains % cfg->world_size != 0)
        return fail("bpm_create: n_chains must be divisible by world_size (unequal blocks break Allgather, demc.py:39,93)");
    if (cfg->algo == BPM_ALGO_DREAM) {
        if (cfg->del_pairs < 1 || cfg->del_pairs > MAX_PAIRS) return fail("bpm_create: 1 <= del_pairs <= 10");
        if (cfg->n_cr < 1 || cfg->n_cr > BPM_MAX_CR) return fail("bpm_create: 1 <= n_cr <= 8");
    }
    int ndev = 0;
    HIPCK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("bpm_create: no HIP device (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("bpm_create: device ordinal out of range");

    bpm_sampler* s = new bpm_sampler();
    s->cfg = *cfg;
    s->cfg.target_params = nullptr;
    s->cfg.nccl_uid = nullptr;
    s->N = (uint32_t)cfg->n_chains;
    s->dim = (uint32_t)cfg->dim;
    s->ld = (s->dim + 1u) & ~1u;
    s->world = (uint32_t)cfg->world_size;
    s->rank = (uint32_t)cfg->rank;
    s->n_local = s->N / s->world;
    s->lo = s->rank * s->n_local;
    (void)pick_shape(s->ld, s->shape);
    // (the one limit on dim: a coordinate pair's Philox block is addressed by a 16-bit slot, philox.h: SLOT_BITS)
    if (s->dim / 2u + SLOT_DIM0 >= (1u << SLOT_BITS)) { delete s; return fail("bpm_create: dim must stay below 131056 (16-bit Philox slot per coordinate pair)"); }
    s->shape_needs_scratch = false;        // (round 3's 32-coordinates-per-lane shape spilled to scratch memory; the looped kernel replaced it)
    if ((uint64_t)s->N * (s->ld + 2) >= (1ull << 31)) { delete s; return fail("bpm_create: n_chains * (dim + 2) must stay below 2^31 (32-bit device offsets)"); }
    const int tid = cfg->target_id;
    const int np = cfg->n_target_params;
    bool ok = true;
    if (tid == BPM_TARGET_GAUSS_EQUICORR) ok = (np == 4 + (int)s->dim);
    else if (tid == BPM_TARGET_MIXTURE_PAIRS) ok = (np == 16 && s->dim % 2 == 0);
    else if (tid == BPM_TARGET_BANANA_2D) ok = (np == 9 && s->dim == 2);
    else if (tid != BPM_TARGET_HOST_CALLBACK) ok = false;
    if (!ok || (np > 0 && !cfg->target_params)) { delete s; return fail("bpm_create: target id / parameter block / dim mismatch"); }
    if (np > 0) s->tparams_h.assign(cfg->target_params, cfg->target_params + np);
    if (cfg->algo != BPM_ALGO_DREAM) { s->cfg.del_pairs = 1; s->cfg.n_cr = 1; }
    if (cfg->algo == BPM_ALGO_DEMC_SYNC) s->cfg.p_snooker = 0.0;

#define CKD(expr)                                   \
    do {                                            \
        int _r = (expr);                            \
        if (_r != 0) { bpm_destroy(s); return _r; } \
    } while (0)
#define HIPCKD(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            bpm_destroy(s);                                                                      \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(_e));               \
        }                                                                                        \
    } while (0)
    HIPCKD(hipSetDevice(cfg->device));
    HIPCKD(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // The per-generation tables are built on the update stream itself, ONE WINDOW AHEAD (on entering window W the build of W + 1 is
    // enqueued in front of W's first update kernel): 16 us per 64 generations at cfg2, and no bpm_step call ever starts with a table
    // build.  (A second stream for the build measured SLOWER on MI355X, 13.7 vs 12.3 us per generation at cfg2:
    // profiles/r02_table_build_modes.txt; removed in round 3.)
    // the library's own AQL queue for the steady state of a single-GPU sampler (aql_queue.h); without it (no large BAR, a runtime
    // without the loader extension, BPM_DIRECT_QUEUE=0) the same kernels are launched on the stream
    // (a host-callback sampler uses it once it has been given a likelihood as HIP source: the update kernel compiled around it, bpm_set_device_likelihood)
    {
        s->dq = bpm::DirectQueue::for_device(cfg->device);
        if (s->dq && s->dq->failed()) s->dq = nullptr;       // (a queue that failed earlier in this process is not adopted: HIP stream launches)
        if (s->dq && !s->dq->kernel(reinterpret_cast<const void*>(perm_table_kernel))) s->dq = nullptr;      // (HIP's copy of the code object not found)
    }
    // The state lives in ordinary device memory; the steady-state packets of the generation loop carry the acquire fence only and the
    // update kernel writes what its successor reads with agent-scope stores (g_dq_update_fence above).  Without the fence kernel (not
    // found among the loaded code objects): acquire + release on every packet, like a HIP stream; bpm_set_launch_path chooses per handle.
    // The round-2 experiment that kept the state in the GPU's hardware-coherent memory type with acquire-only packets is NOT part of
    // this library: histories came back with older contents in sequences of samplers created and destroyed in one process, cause not
    // found (profiles/r02_coherent_memory_hazard.txt).  It builds only as `make variant NAME=coherent DEFS=-DBPM_EXPERIMENT_COHERENT`.
    {
#ifdef BPM_EXPERIMENT_COHERENT
        const char* c = getenv("BPM_COHERENT_STATE");      // (=2: take the memory type on trust, without the probe)
        s->coherent = s->dq != nullptr && c && atoi(c) != 0 && (atoi(c) == 2 || state_memory_is_coherent(s->dq, cfg->device));
#endif
        const bool fence_kernel = s->dq != nullptr && s->dq->set_fence_kernel(reinterpret_cast<const void*>(queue_fence_kernel));
        s->dq_fence = fence_kernel ? bpm::DirectQueue::ACQUIRE : bpm::DirectQueue::FENCED;
#ifdef BPM_EXPERIMENT_COHERENT
        if (const char* f = getenv("BPM_DQ_FENCE")) if (s->coherent && !strcmp(f, "none")) s->dq_fence = 0;
#endif
    }
    HIPCKD(hipEventCreate(&s->ev0));
    HIPCKD(hipEventCreate(&s->ev1));
    for (auto& B : s->tb) HIPCKD(hipEventCreateWithFlags(&B.built, hipEventDisableTiming));
    s->L.blk = (uint64_t)s->n_local * (s->ld + 2);
    s->L.magic = (uint32_t)((1ull << 32) / s->n_local) + 1u;
    s->L.n_local = s->n_local; s->L.ld = s->ld; s->L.dim = s->dim; s->L.world = s->world;
    const size_t row_d = (size_t)s->n_local * s->ld;
    const bool want_om = cfg->algo == BPM_ALGO_DREAM && cfg->outlier_every > 0;
    if (s->world > 1 && s->world <= (uint32_t)MAX_SEG && cfg->algo != BPM_ALGO_DEMC_SYNC && cfg->target_id != BPM_TARGET_HOST_CALLBACK) {
        // the arena of the push exchange: [G | om | control block], one allocation = one IPC handle
        auto up = [](size_t b) { return (b + 255u) & ~(size_t)255u; };
        const size_t g_bytes = up((size_t)s->world * s->L.blk * sizeof(double));
        const size_t om_bytes = want_om ? up((size_t)s->world * 2 * s->n_local * sizeof(double)) : 0;
        s->off_om = g_bytes; s->off_ctrl = g_bytes + om_bytes;
        s->arena_bytes = s->off_ctrl + up(sizeof(PushCtrl));
        HIPCKD(hipMalloc(&s->arena, s->arena_bytes));
        HIPCKD(hipMemsetAsync(s->arena, 0, s->arena_bytes, s->stream));
        s->G = reinterpret_cast<double*>(s->arena);
        if (want_om) s->om = reinterpret_cast<double*>(reinterpret_cast<char*>(s->arena) + s->off_om);
        s->ctrl = reinterpret_cast<PushCtrl*>(reinterpret_cast<char*>(s->arena) + s->off_ctrl);
        {   // The flags are polled INSIDE a kernel while other agents write them: the memory model promises coherence across agents for
            // fine-grained memory only (coarse-grained: at kernel boundaries).  So the control block gets a fine-grained allocation of its own
            // when the runtime gives one that can be exported; else it stays in the arena (ranks sharing one GPU do not need more).
            void* fg = nullptr;
            hipIpcMemHandle_t probe;
            if (!test_path("ctrlarena") && hipExtMallocWithFlags(&fg, up(sizeof(PushCtrl)), hipDeviceMallocFinegrained) == hipSuccess && fg &&
                hipMemset(fg, 0, up(sizeof(PushCtrl))) == hipSuccess && hipIpcGetMemHandle(&probe, fg) == hipSuccess) {
                s->ctrl = reinterpret_cast<PushCtrl*>(fg);
                s->ctrl_fine = true;
            } else {
                (void)hipGetLastError();
                if (fg) (void)hipFree(fg);
            }
        }
        CKD(dev_alloc(&s->tab_peerG, (size_t)MAX_PEERS));
        CKD(dev_alloc(&s->tab_all, (size_t)3 * MAX_SEG));
        HIPCKD(hipMemsetAsync(s->tab_peerG, 0, MAX_PEERS * sizeof(unsigned long long), s->stream));
        HIPCKD(hipMemsetAsync(s->tab_all, 0, 3 * MAX_SEG * sizeof(unsigned long long), s->stream));
    } else {
        CKD(dev_alloc_state(&s->G, (size_t)s->world * s->L.blk, s->coherent));
        HIPCKD(hipMemsetAsync(s->G, 0, (size_t)s->world * s->L.blk * sizeof(double), s->stream));
    }
    s->L.G = s->G;
    CKD(dev_alloc_state(&s->ll, s->n_local, s->coherent));
    CKD(dev_alloc_state(&s->w_mean, 2 * row_d, s->coherent));      // one record per chain: [mean (ld) | m2 (ld)]
    s->w_m2 = s->w_mean + s->ld;
    CKD(dev_alloc(&s->tparams, (size_t)np + 2));       // (+2: the wide-row kernels read the Gaussian's 1/sigma as pairs)
    HIPCKD(hipMemsetAsync(s->tparams, 0, ((size_t)np + 2) * sizeof(double), s->stream));
    if (np > 0) HIPCKD(hipMemcpyAsync(s->tparams, s->tparams_h.data(), (size_t)np * sizeof(double), hipMemcpyHostToDevice, s->stream));
    CKD(dev_alloc_state(&s->cr_state_base, 6 * MAX_CR, s->coherent));
    s->cr_state = s->cr_state_base; s->cr_state_alt = s->cr_state_base + 3 * MAX_CR;
    if (cfg->algo == BPM_ALGO_DREAM) {       // the partial sums of the CR reduction (kernels.h)
        const uint32_t n_first = (s->N + 1u) / 2u;
        s->cr_g1 = (uint32_t)cr_g1(s->shape.lpc);
        s->cr_n1 = cr_chunks_of(n_first, s->cr_g1) + cr_chunks_of(s->N - n_first, s->cr_g1);
        const size_t n2 = ((size_t)s->cr_n1 + WAVE - 1) / WAVE;
        CKD(dev_alloc_state(&s->cr_p1, (size_t)4 * MAX_CR * s->cr_n1, s->coherent));
        HIPCKD(hipMemsetAsync(s->cr_p1, 0, (size_t)4 * MAX_CR * s->cr_n1 * sizeof(double), s->stream));
        s->cr_p1_cur = s->cr_p1;
        if (s->cr_n1 > CR_FINAL_MAX)
            for (auto& b : s->cr_p2) { CKD(dev_alloc_state(&b, (size_t)2 * MAX_CR * n2, s->coherent)); HIPCKD(hipMemsetAsync(b, 0, (size_t)2 * MAX_CR * n2 * sizeof(double), s->stream)); }
    }
    {
        double init[3 * MAX_CR] = {0};
        for (int m = 0; m < s->cfg.n_cr; ++m) init[m] = 1.0 / s->cfg.n_cr;   // dream.py:114
        HIPCKD(hipMemcpyAsync(s->cr_state, init, sizeof(init), hipMemcpyHostToDevice, s->stream));
        HIPCKD(hipMemcpyAsync(s->cr_state_alt, init, sizeof(init), hipMemcpyHostToDevice, s->stream));
        HIPCKD(hipStreamSynchronize(s->stream));
    }
    // position-ordered history append: one wavefront per chain writes whole 128-byte lines whatever the row's place, nothing to gain there
    // the lean form of the small-d update kernels (kernels.h: lean_scalars) where a launch is transaction bound: from 49152 chains per GPU
    // (cfg3's 65536 and cfg5's 262144 gain 0.6 / 2.6 us per generation, cfg5's per-GPU share of 32768 would lose 0.6)
    // (the banana's kernels have the lean form only: its ln_like is a dozen flops)
    s->lean = s->shape.idx < 3 && tid != BPM_TARGET_HOST_CALLBACK &&
              (tid == BPM_TARGET_BANANA_2D || ((s->n_local >= 49152u || test_path("lean")) && !test_path("nolean")));
    // (a host-callback sampler: used only while the update kernel compiled around its HIP-source likelihood drives it -- prepare_generation asks g_user_cur;
    // the proposal / commit kernels of the host transports append by chain)
    s->hist_by_pos = s->world == 1 && s->shape.idx < 3 && cfg->keep_history != 0 && cfg->algo != BPM_ALGO_DEMC_SYNC && !test_path("histchain");
    if (s->hist_by_pos) CKD(dev_alloc(&s->hist_tmp, (size_t)s->n_local * (s->ld + 1)));
    if (s->cfg.running_moments) {
        s->gs_nb = (uint32_t)std::max<uint32_t>(1u, std::min<uint32_t>(256u, (s->n_local + 63u) / 64u));
        CKD(dev_alloc(&s->gs_shift, (size_t)s->ld));
        CKD(dev_alloc(&s->gs_part, (size_t)s->gs_nb * 2 * s->ld));
    }
    CKD(dev_alloc(&s->counters, 8));      // [2] NaN ratios of this run; [4] outlier resets since creation
    CKD(dev_alloc_state(&s->acc_count, (size_t)s->n_local + ACC_SHARDS, s->coherent));      // per-chain counters | per-wavefront shards (kernels.h: lean_scalars)
    HIPCKD(hipMemsetAsync(s->acc_count, 0, ((size_t)s->n_local + ACC_SHARDS) * sizeof(uint32_t), s->stream));
    HIPCKD(hipMemsetAsync(s->counters, 0, 8 * sizeof(unsigned long long), s->stream));
    if (want_om) {
        if (!s->om) CKD(dev_alloc(&s->om, (size_t)s->world * 2 * s->n_local));
        CKD(dev_alloc(&s->sel, 8));
        // (behind the state: one (value, index) pair per workgroup of a pass for the first maximum)
        const size_t sel_bytes = sizeof(SelState) + (size_t)sel_blocks(s->N) * (sizeof(double) + sizeof(uint32_t));
        CKD(dev_alloc(&s->sel_state, sel_bytes));
        HIPCKD(hipMemsetAsync(s->sel_state, 0, sel_bytes, s->stream));
        CKD(dev_alloc(&s->olist, (size_t)s->n_local + 1));
    }
    // update records drawn ahead (plan_kernel) for the fused device kernels, while a launch is latency bound.  Measured
    // on cfg2's target (one wavefront per chain): 11.4 vs 11.8 us/generation at N=2048, 15.9 vs 16.4 at 8192, 24.9 vs
    // 25.3 at 16384; from 32768 chains per GPU the records' extra 64 B per update cost more than the shorter critical
    // path gains (42.7 vs 41.6, 77.9 vs 75.8 at 65536).
    static const bool no_plan = test_path("noplan") || test_path("noperm");
    const uint32_t plan_max_local = test_path("planall") ? 0xFFFFFFFFu : 16384u;
    s->win_K = PERM_CHUNK;
    s->plan_on = !no_plan && s->shape.idx == 3 && s->n_local <= plan_max_local &&
                 (cfg->algo == BPM_ALGO_DREAM ? cfg->del_pairs <= 5 : cfg->algo == BPM_ALGO_DEMC) &&
                 // (a single-rank host-callback sampler too: the update kernel compiled around a likelihood given as HIP source reads them, bpm_set_device_likelihood)
                 (tid != BPM_TARGET_HOST_CALLBACK || s->world == 1);
    if (s->plan_on) {
        const size_t per_gen = (size_t)s->N * PLAN_WORDS * sizeof(uint32_t);
        s->win_K = (int)std::max<size_t>(1, std::min<size_t>((size_t)s->win_K, ((size_t)512 << 20) / per_gen));
        s->sorted_on = s->world > 1 && s->world <= (uint32_t)MAX_SEG && tid != BPM_TARGET_HOST_CALLBACK;
    }
    for (auto& B : s->tb) {
        CKD(dev_alloc(&B.perm, (size_t)s->win_K * s->N));
        CKD(dev_alloc(&B.inv, (size_t)s->win_K * s->N));
        if (s->plan_on) CKD(dev_alloc(&B.plan, (size_t)s->win_K * s->N * PLAN_WORDS));
        if (s->sorted_on) {
            CKD(dev_alloc(&B.sidx, (size_t)s->win_K * s->N));
            HIPCKD(hipMemsetAsync(B.sidx, 0xFF, (size_t)s->win_K * s->N * sizeof(uint32_t), s->stream));      // "no slot": plan_kernel skips such positions
            CKD(dev_alloc(&B.plan_count, (size_t)s->win_K * 2 * s->world));
            HIPCKD(hipHostMalloc(reinterpret_cast<void**>(&B.count_h), (size_t)s->win_K * 2 * s->world * sizeof(uint32_t), hipHostMallocDefault));
        }
    }
    if (s->plan_on) s->plan_tab = s->tb[0].plan;       // (non-null from here on: "this sampler launches with records")
    CKD(dev_alloc(&s->gamma_tab, (size_t)s->dim + 1));
    if (cfg->algo == BPM_ALGO_DEMC_SYNC) CKD(dev_alloc(&s->x_next, row_d));
    {
        std::vector<double> gt(s->dim + 1, 0.0);      // dream.py:61, same operation order as the reference
        for (uint32_t dp = 1; dp <= s->dim; ++dp)
            gt[dp] = s->cfg.gamma_scale * 2.38 / std::sqrt(2. * (double)s->cfg.del_pairs * (double)dp);
        HIPCKD(hipMemcpyAsync(s->gamma_tab, gt.data(), gt.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
        HIPCKD(hipStreamSynchronize(s->stream));
    }
    s->scratch_doubles = 4 * (size_t)s->ld + 64;
    CKD(dev_alloc(&s->scratch, s->scratch_doubles));
    if (tid == BPM_TARGET_HOST_CALLBACK) {
        CKD(dev_alloc(&s->prop_buf, row_d));
        CKD(dev_alloc(&s->aux_buf, 2 * (size_t)s->n_local));
        CKD(dev_alloc(&s->ids_buf, s->n_local));
        HIPCKD(hipHostMalloc(reinterpret_cast<void**>(&s->h_ids), s->n_local * sizeof(int32_t), hipHostMallocDefault));
        HIPCKD(hipHostMalloc(reinterpret_cast<void**>(&s->h_props), row_d * sizeof(double), hipHostMallocDefault));
        HIPCKD(hipHostMalloc(reinterpret_cast<void**>(&s->h_ll), (size_t)s->n_local * sizeof(double), hipHostMallocDefault));
    }
    if (s->world > 1 && !cfg->nccl_uid) { bpm_destroy(s); return fail("bpm_create: nccl_uid required when world_size > 1"); }
    // Test mode: a uid starting with "BPMLOCAL" makes the ranks of a world handles of ONE process on one GPU;
    // bpm_local_group_step drives them in lock-step and does the all-gather with device copies.  It runs the
    // very kernels, layouts and host logic of a multi-GPU run (everything but RCCL) where only one GPU exists.
    if (cfg->nccl_uid && std::memcmp(cfg->nccl_uid, "BPMLOCAL", 8) == 0) {
#ifndef BPM_TEST_HOOKS
        bpm_destroy(s);
        return fail("bpm_create: local rank groups (a nccl_uid starting with BPMLOCAL, driven by bpm_local_group_step) exist only in the test variant of the library (include/bipymc_hip_test.h)");
#endif
        s->local_group = true;
        if (test_path("groupqueues") && s->dq) {
            bpm::DirectQueue* q = bpm::DirectQueue::create_private(cfg->device);
            if (q && q->kernel(reinterpret_cast<const void*>(perm_table_kernel)) && q->set_fence_kernel(reinterpret_cast<const void*>(queue_fence_kernel))) { s->dq = q; s->dq_private = true; }
            else if (q) bpm::DirectQueue::destroy_private(q);
        }
    } else if (cfg->nccl_uid && std::memcmp(cfg->nccl_uid, "BPMPUSH", 7) == 0) {
        // ranks in processes of their own WITHOUT an RCCL communicator: the push exchange is the only one (bpm_push_connect before the
        // first step).  What N processes sharing one GPU can use -- RCCL refuses two ranks on one device -- and what a caller without
        // RCCL on its nodes can use.
        if (!s->arena) { bpm_destroy(s); return fail("bpm_create: the push exchange needs a device target, 2..16 ranks and the pool-based samplers"); }
        s->push_no_rccl = true;
    } else
    if (cfg->nccl_uid) {      // world_size == 1 with a uid: a one-rank communicator (exercises the RCCL path on one GPU)
        CKD(load_rccl());
        ncclUniqueId id;
        std::memcpy(&id, cfg->nccl_uid, BPM_UID_BYTES);
        ncclResult_t r = g_rccl.CommInitRank(&s->comm, (int)s->world, id, (int)s->rank);
        if (r != ncclSuccess) { std::string m = std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(r); bpm_destroy(s); return fail(m); }
    }
    // (a one-rank communicator takes the same path, so a one-GPU box can time and test it through RCCL)
    if ((s->world > 1 || s->comm) && cfg->algo != BPM_ALGO_DEMC_SYNC && cfg->target_id != BPM_TARGET_HOST_CALLBACK) {
        // BPM_EXCHANGE = dense | rows | replay (default replay): the initial value of bpm_set_exchange
        const char* xe = getenv("BPM_EXCHANGE");
        const bool want_dense = xe && std::strcmp(xe, "dense") == 0;
        const bool want_rows = xe && std::strcmp(xe, "rows") == 0;
        s->replay_enabled = !want_dense && !want_rows && s->shape.idx != SHAPE_WIDE;      // (wide rows: dense, or push once connected)
        s->sparse_enabled = !want_dense && want_rows && s->shape.idx != SHAPE_WIDE;
        CKD(dev_alloc(&s->accbits_all, (size_t)s->N));
        HIPCKD(hipMemsetAsync(s->accbits_all, 0, (size_t)s->N, s->stream));
        s->xnsub = 1u;
        // 4 sub-blocks: ~175 acceptances per counter and half generation at cfg2 cost the same on one GPU as 16
        // sub-blocks (23.2 vs 23.0 us/generation through a one-rank communicator; ONE counter: 34 us), and the
        // capacity margin, hence the bytes on the wire, shrinks with the count per sub-block (1.5 x instead of 2.2 x)
        const uint32_t nsub_max = 4u;
        while (2u * s->xnsub <= nsub_max && 2u * s->xnsub * 8u <= s->n_local) s->xnsub *= 2u;      // power of two, >= 8 chains each
        s->xcap_max = ((s->n_local + s->xnsub - 1u) / s->xnsub + 1u) & ~1u;    // every chain of a sub-block accepted
        s->xcap = s->xcap_max;                               // first chunk: cannot overflow; then sized from the counts seen
        const size_t pk_doubles = (size_t)s->world * s->xnsub * s->xstride();
        CKD(dev_alloc(&s->PK, pk_doubles));
        HIPCKD(hipMemsetAsync(s->PK, 0, pk_doubles * sizeof(double), s->stream));
        CKD(dev_alloc(&s->xstat, 2));
        HIPCKD(hipMemsetAsync(s->xstat, 0, 2 * sizeof(uint32_t), s->stream));
        CKD(dev_alloc(&s->ckpt_G, (size_t)s->world * s->L.blk));
        CKD(dev_alloc(&s->ckpt_ll, s->n_local));
        CKD(dev_alloc(&s->ckpt_acc, (size_t)s->n_local + ACC_SHARDS));
        CKD(dev_alloc(&s->ckpt_counters, 4));
    }
    HIPCKD(hipStreamSynchronize(s->stream));
    *out = s;
    return 0;
}

extern "C" int bpm_init_chains(bpm_handle_t s, const double* theta_0, const double* varepsilon) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!theta_0 || !varepsilon) return fail("bpm_init_chains: null argument");
    int jitter = 1;
    for (uint32_t j = 0; j < s->dim; ++j) {
        if (varepsilon[j] < 0.0) return fail("bpm_init_chains: varepsilon must be >= 0 (chain.py:22)");
        if (!(varepsilon[j] > 0.0)) jitter = 0;   // util.py:12: all > 0 or no noise at all
    }
    double* d_theta = s->scratch;
    double* d_var = s->scratch + s->ld;
    HIPCK(hipMemcpyAsync(d_theta, theta_0, s->dim * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIPCK(hipMemcpyAsync(d_var, varepsilon, s->dim * sizeof(double), hipMemcpyHostToDevice, s->stream));
    const uint64_t n_elem = (uint64_t)s->n_local * s->ld;
    for (uint32_t r = 0; r < s->world; ++r) {   // every rank fills the whole replicated matrix: no exchange needed
        hipLaunchKernelGGL(init_jitter_kernel, dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0, s->stream, s->L,
                           r * s->n_local, (uint64_t)s->cfg.seed, d_theta, d_var, jitter);
    }
    HIPCK(hipGetLastError());
    CK(reset_history(s));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_set_state(bpm_handle_t s, const double* X) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!X) return fail("bpm_set_state: null argument");
    for (uint32_t r = 0; r < s->world; ++r) {
        HIPCK(hipMemcpy2DAsync(s->G + (uint64_t)r * s->L.blk, s->ld * sizeof(double),
                               X + (uint64_t)r * s->n_local * s->dim, s->dim * sizeof(double), s->dim * sizeof(double),
                               s->n_local, hipMemcpyHostToDevice, s->stream));
    }
    CK(reset_history(s));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_get_state(bpm_handle_t s, double* X) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!X) return fail("bpm_get_state: null argument");
    for (uint32_t r = 0; r < s->world; ++r) {
        HIPCK(hipMemcpy2DAsync(X + (uint64_t)r * s->n_local * s->dim, s->dim * sizeof(double),
                               s->G + (uint64_t)r * s->L.blk, s->ld * sizeof(double), s->dim * sizeof(double),
                               s->n_local, hipMemcpyDeviceToHost, s->stream));
    }
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_set_loglike(bpm_handle_t s, const double* ll_local) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!ll_local) return fail("bpm_set_loglike: null argument");
    // (the lean kernels never read the cache and refresh_ll would overwrite what a caller put there, a history row appended by position
    // would get its ln-likes by chain -- and nothing needs the call for a device target: the library evaluates the cache itself, reset_history)
    if (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK)
        return fail("bpm_set_loglike: host-callback targets only (a sampler with a device target evaluates ln_like of its chains itself)");
    HIPCK(hipMemcpyAsync(s->ll, ll_local, s->n_local * sizeof(double), hipMemcpyHostToDevice, s->stream));
    // the log-like history row of the current state: row 0 after (re)initialisation, the LAST row after a warm start
    // (bpm_set_history with a host-callback target could only leave NaN there; older rows stay NaN: the reference's
    // checkpoint does not store log-likes, chain.py:59-70)
    if (s->hist_rows >= 1 && s->hist_rows == s->rows_logical)
        HIPCK(hipMemcpyAsync(s->llhist + (size_t)(s->hist_rows - 1) * s->n_local, s->ll, s->n_local * sizeof(double),
                             hipMemcpyDeviceToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_get_loglike(bpm_handle_t s, double* ll_local) {
    CK(check_handle(s));
    CK(set_device(s));
    CK(refresh_ll(s));
    HIPCK(hipMemcpyAsync(ll_local, s->ll, s->n_local * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

static int ensure_perm_table(bpm_sampler* s, int64_t t, int64_t n_ahead);

extern "C" int bpm_begin_run(bpm_handle_t s, const bpm_run_opts_t* o) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->state_set) return fail("ERROR: chains not initilized");   // demc.py:65-66
    if (s->proposed) return fail("bpm_begin_run: a proposed half generation is awaiting bpm_commit");
    bpm_run_opts_t d;
    d.flip = 0.5; d.shuffle = 1; d._pad = 0; d.epsilon = -1.0; d.u_epsilon = -1.0; d.gamma = -1.0;
    if (o) d = *o;
    d.flip = std::min(1.0, std::max(0.0, d.flip));                                   // demc.py:73
    if (d.epsilon < 0.0) d.epsilon = (s->cfg.algo == BPM_ALGO_DREAM) ? 1e-12 : 1e-15;   // dream.py:40 / demc.py:161
    if (d.u_epsilon < 0.0) d.u_epsilon = 1e-2;                                        // dream.py:41
    if (!(d.gamma > 0.0)) d.gamma = 2.38 / std::sqrt(2.0 * (double)s->dim);            // demc.py:162
    s->opts = d;
    s->k_gen = 0;                                                                    // demc.py:78
    s->phase = 0;
    HIPCK(hipMemsetAsync(s->counters, 0, 4 * sizeof(unsigned long long), s->stream));
    HIPCK(hipMemsetAsync(s->acc_count, 0, ((size_t)s->n_local + ACC_SHARDS) * sizeof(uint32_t), s->stream));     // demc.py:67
    HIPCK(hipStreamSynchronize(s->stream));
    s->run_open = true;
    // the tables of the window this run starts in (and of the next one) are built from here on, beside whatever the caller does
    // before its first bpm_step
    if (s->cfg.algo != BPM_ALGO_DEMC_SYNC) CK(ensure_perm_table(s, s->t_abs, 1));
    return 0;
}

static int allgather_state(bpm_sampler* s) {
    if (!s->comm) return 0;
    NCCLCK(g_rccl.AllGather(s->G + (uint64_t)s->rank * s->L.blk, s->G, (size_t)s->L.blk, ncclDouble, s->comm, s->stream));
    return 0;
}

// Window W of the per-generation tables into buffer b, on the build stream.
static int build_window(bpm_sampler* s, int b, int64_t W, int shuffle) {
    bpm_sampler::TabBuf& B = s->tb[b];
    const int K = s->win_K;
    const int64_t t0 = W * K;
    hipStream_t bs = s->stream;
    PermKeys keys;
    for (int g = 0; g < K; ++g) keys.k[g] = make_perm_key(s->cfg.seed, (uint64_t)(t0 + g), s->N, shuffle != 0);
    for (int g = K; g < PERM_CHUNK; ++g) keys.k[g] = keys.k[0];
    const uint64_t n = (uint64_t)K * s->N;
    // the inverse tables (chain id -> position) serve launches with one work item per LOCAL chain: the ranks of a world (and the test path that takes
    // their kernel path on one GPU); a single-GPU sampler's buffer stays allocated (PhaseArgs::inv_tab is never null with tables) but is not filled
    static const bool force_mode1 = test_path("mode1");
    uint32_t* const inv_out = (s->world > 1 || force_mode1) ? B.inv : nullptr;
    if (g_dq && !B.sidx) {
        // direct mode, records by position: the same two kernels as packets on the library's queue, in order with
        // the update kernels around them
        struct { PermKeys keys; uint32_t n_gens, N; uint32_t* tab; uint32_t* inv; } pa{keys, (uint32_t)K, s->N, B.perm, inv_out};
        static_assert(offsetof(decltype(pa), tab) == sizeof(PermKeys) + 8, "kernarg layout of perm_table_kernel");
        const bpm::DqKernel* kp = g_dq->kernel(reinterpret_cast<const void*>(perm_table_kernel));
        if (!kp || g_dq->launch(*kp, (uint32_t)((n + 255) / 256), 1, 256, &pa, sizeof(pa), bpm::DirectQueue::FENCED) != 0) return fail("direct AQL queue: perm_table_kernel: " + g_dq->why());
        if (B.plan) {
            struct { PlanParams P; const uint32_t* tab; uint32_t* plan; const uint32_t* sidx; } qa{
                PlanParams{s->cfg.seed, (uint64_t)t0, (uint32_t)K, s->N, s->cfg.algo == BPM_ALGO_DREAM ? (uint32_t)s->cfg.del_pairs : 1u,
                           (s->cfg.algo == BPM_ALGO_DEMC && s->cfg.p_snooker > 0.0) ? 1u : 0u, 0u, 0u},
                B.perm, B.plan, B.sidx};
            const bpm::DqKernel* kq = g_dq->kernel(reinterpret_cast<const void*>(plan_kernel));
            if (!kq || g_dq->launch(*kq, (uint32_t)((n + 255) / 256), 1, 256, &qa, sizeof(qa), bpm::DirectQueue::FENCED) != 0) return fail("direct AQL queue: plan_kernel: " + g_dq->why());
        }
        B.W = W;
        B.shuffle = shuffle;
        g_dq_need_acquire = true;
        return 0;
    }
    // A rank of a world with owner-sorted records (B.sidx) builds on the HIP stream even while its generation loop runs on the library's
    // own queue (push exchange): the slot table comes from plan_slot_kernel and the launch sizes of the window must reach the HOST
    // (pinned copy + event), neither of which the queue path above does -- it once took a window with an unbuilt slot table (records
    // scattered to garbage slots: a memory fault after 64 generations, the first window built inside direct mode).  Queue drained
    // before, stream drained after: once per 64 generations.
    StreamSection sec(s);
    CK(sec.rc);
    hipLaunchKernelGGL(perm_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, bs, keys, (uint32_t)K, s->N, B.perm, inv_out);
    HIPCK(hipGetLastError());
    if (B.plan) {
        // (push exchange: nobody replays another rank's updates -- only this rank's own run of the owner-sorted records is built;
        // a window remembers it, and a change of the exchange mode rebuilds)
        const bool own_only = B.sidx != nullptr && s->push_enabled && s->push_connected;
        B.own_only = own_only;
        if (B.sidx) {              // world > 1: every position's slot in the owner-sorted order, every rank's counts (they size the launches)
            if (own_only) {        // this rank's positions only: the other ranks' counts stay zero, its own run starts at the half's offset
                const uint32_t n_chunks = ((s->N + 1u) / 2u + SLOT_CHUNK - 1u) / SLOT_CHUNK;
                if (!B.chunk_count) CK(dev_alloc(&B.chunk_count, (size_t)s->win_K * 2 * n_chunks));
                HIPCK(hipMemsetAsync(B.plan_count, 0, (size_t)K * 2 * s->world * sizeof(uint32_t), bs));
                for (uint32_t assign = 0; assign < 2; ++assign)
                    hipLaunchKernelGGL(plan_slot_own_kernel, dim3(n_chunks, 2, (unsigned)K), dim3(SLOT_CHUNK), 0, bs, B.perm, s->N, s->lo, s->n_local, s->world,
                                       s->rank, n_chunks, B.chunk_count, B.sidx, B.plan_count, assign);
            } else {
                hipLaunchKernelGGL(plan_slot_kernel, dim3(2, (unsigned)K), dim3(PLAN_LOCAL_THREADS), 0, bs, B.perm, s->N, s->n_local, s->world,
                                   B.sidx, B.plan_count);
            }
            HIPCK(hipGetLastError());
            HIPCK(hipMemcpyAsync(B.count_h, B.plan_count, (size_t)K * 2 * s->world * sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
        }
        PlanParams pp{s->cfg.seed, (uint64_t)t0, (uint32_t)K, s->N, s->cfg.algo == BPM_ALGO_DREAM ? (uint32_t)s->cfg.del_pairs : 1u,
                      (s->cfg.algo == BPM_ALGO_DEMC && s->cfg.p_snooker > 0.0) ? 1u : 0u, own_only ? s->lo : 0u, own_only ? s->n_local : 0u};
        hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, bs, pp, B.perm, B.plan, B.sidx);
        HIPCK(hipGetLastError());
    }
    HIPCK(hipEventRecord(B.built, bs));
    CK(sec.end());
    B.W = W;
    B.shuffle = shuffle;
    return 0;
}

// The tables of generation t are current on the update stream; on entering a window the next one is started on the build
// stream.  In a run that crosses windows the update kernels therefore find every window but the very first already built
// (bpm_begin_run starts that one), and a rank of a world finds its launch sizes on the host without stalling.
static int ensure_perm_table(bpm_sampler* s, int64_t t, int64_t /*n_ahead*/) {
    const int shuffle = s->opts.shuffle != 0 ? 1 : 0;
    if (s->cur >= 0 && s->tab_shuffle == shuffle && t >= s->tab_t0 && t < s->tab_t0 + s->tab_K) return 0;
    const int K = s->win_K;
    const int64_t W = t / K;
    const int b = (int)(W & 1);
    bpm_sampler::TabBuf& B = s->tb[b];
    const bool want_own = B.sidx != nullptr && s->push_enabled && s->push_connected;
    if (B.W != W || B.shuffle != shuffle || (B.sidx && B.own_only && !want_own)) CK(build_window(s, b, W, shuffle));
    if (B.sidx) HIPCK(hipEventSynchronize(B.built));              // the window's launch sizes (count_h)
    s->cur = b;
    s->perm_tab = B.perm; s->inv_tab = B.inv; s->plan_tab = B.plan; s->plan_count_h = B.count_h;
    s->tab_t0 = W * K;
    s->tab_K = K;
    s->tab_shuffle = shuffle;
    bpm_sampler::TabBuf& Bn = s->tb[b ^ 1];
    if (Bn.W != W + 1 || Bn.shuffle != shuffle || (Bn.sidx && Bn.own_only && !want_own)) CK(build_window(s, b ^ 1, W + 1, shuffle));
    return 0;
}

// Everything of one generation that is decided on the host: flip, shuffle key, group ranges
// (demc.py:81-86,95-100), gating flags (dream.py:92,123), history row.
// the host-callback sampler whose generation loop the update kernel compiled around its HIP-source likelihood is driving (run_generations; else nullptr)
static thread_local bpm_sampler* g_user_cur = nullptr;
static thread_local bool g_user_launch_failed = false;
// totals (cr_state) + `cnt` partial sums of one generation -> cr_state: cr_final_kernel on the queue the generation loop runs on
static int launch_cr_final(bpm_sampler* s, const double* src, uint32_t cnt) {
    const uint32_t n_cr = (uint32_t)s->cfg.n_cr;
    const int fence = s->dq_fence | bpm::DirectQueue::ACQUIRE;
    // (ROUNDS = partials per lane, the next power of two: a partial beyond cnt reads as +0.0, the sums do not depend on the choice)
    typedef void (*FinalK)(const double*, const double*, uint32_t, uint32_t, double*);
    const uint32_t rounds = (cnt + WAVE - 1) / WAVE;
    const FinalK kfn = rounds <= 1 ? cr_final_kernel<1> : (rounds <= 2 ? cr_final_kernel<2> : (rounds <= 4 ? cr_final_kernel<4> : cr_final_kernel<8>));
    if (g_dq) {
        struct { const double* tot; const double* part; uint32_t nb, n_cr; double* cr_state; } fa{s->cr_state, src, cnt, n_cr, s->cr_state};
        const bpm::DqKernel* kf = g_dq->kernel(reinterpret_cast<const void*>(kfn));
        if (!kf || g_dq->launch(*kf, 1, 1, WAVE, &fa, sizeof(fa), fence) != 0) return fail("direct AQL queue: cr_final_kernel: " + g_dq->why());
        g_dq_need_acquire = false;
    } else {
        hipLaunchKernelGGL(kfn, dim3(1), dim3(WAVE), 0, s->stream, (const double*)s->cr_state, src, cnt, n_cr, s->cr_state);
        HIPCK(hipGetLastError());
    }
    return 0;
}
// the fold nobody consumed (bpm_sampler::cr_pending)
static int cr_flush_pending(bpm_sampler* s) {
    if (!s->cr_pending) return 0;
    s->cr_pending = false;
    return launch_cr_final(s, s->cr_pend_src, s->cr_pend_cnt);
}

static int prepare_generation(bpm_sampler* s, int64_t n_ahead) {
    const bool dream = s->cfg.algo == BPM_ALGO_DREAM;
    const uint64_t t = (uint64_t)s->t_abs;
    const bool sync = s->cfg.algo == BPM_ALGO_DEMC_SYNC;
    if (!sync) CK(ensure_perm_table(s, s->t_abs, n_ahead));
    const bool flip = flip_draw(s->cfg.seed, t, s->opts.flip);
    const PermKey pk = make_perm_key(s->cfg.seed, t, s->N, s->opts.shuffle != 0);
    const uint32_t n_first = (s->N + 1) / 2, n_second = s->N - n_first;    // np.array_split: first gets ceil
    uint32_t a_off = 0, a_n = n_first, b_off = n_first, b_n = n_second;
    if (flip) { std::swap(a_off, b_off); std::swap(a_n, b_n); }
    const bool adapt_on = dream && (s->cfg.burnin_gen > s->k_gen);          // dream.py:92
    s->gen_adapt_on = adapt_on;
    // the CR statistics of a generation are reduced only when some update could contribute: while the chains' histories are not longer than
    // n_cr_gen (dream.py:123) every slot says "no update" and the reduction would leave p_cr, delta_m and n_cr_updates as they are -- two
    // dependent dispatches (6 us at cfg2) for nothing, in the first n_cr_gen generations of every fresh run
    s->gen_cr_reduce = adapt_on && s->rows_logical > s->cfg.n_cr_gen;
    if (adapt_on && s->w_rows != s->rows_logical) {
        if (!s->cfg.keep_history || s->hist_rows != s->rows_logical)
            return fail("CR adaptation needs the chain history (keep_history=1) to rebuild its moments");
        const uint64_t n_elem = (uint64_t)s->n_local * s->ld;
        StreamSection sec(s);
        CK(sec.rc);
        CK(normalize_history(s, 0, s->hist_rows));
        hipLaunchKernelGGL(welford_rebuild_kernel, dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0, s->stream,
                           s->hist, n_elem, n_elem, (uint32_t)s->hist_rows, s->ld, s->w_mean, s->w_m2);
        HIPCK(hipGetLastError());
        CK(sec.end());
        s->w_rows = s->rows_logical;
    }
    CK(ensure_gen_sums(s, s->rows_logical + 1));
    double* hist_row = nullptr;
    double* llhist_row = nullptr;
    if (s->cfg.keep_history) {
        CK(ensure_history(s, s->hist_rows + 1));
        hist_row = s->hist + (uint64_t)s->hist_rows * s->n_local * s->ld;
        llhist_row = s->llhist + (uint64_t)s->hist_rows * s->n_local;
        s->hist_tag[(size_t)s->hist_rows] = -1;      // (how the row about to be appended has to be read: decided in finish_generation)
    }
    for (int ph = 0; ph < 2; ++ph) {
        PhaseArgs& a = s->cur_args[ph];
        std::memset(&a, 0, sizeof(a));
        a.L = s->L;
        a.ll = s->ll;
        a.hist_row = hist_row;
        a.llhist_row = llhist_row;
        a.w_mean = s->w_mean;
        a.w_m2 = s->w_m2;
        a.tparams = s->tparams;
        a.cr_state = s->cr_state;
        a.counters = s->counters;
        a.acc_count = s->acc_count;
        a.prop_buf = s->prop_buf;
        a.aux_buf = s->aux_buf;
        a.ids_buf = s->ids_buf;
        trace_set(a, s->trace_i32, s->trace_f64, s->trace_mask);      // (test variant only: the product's argument block has no trace fields)
        a.pk = pk;
        a.perm_tab = s->perm_tab + (uint64_t)(s->t_abs - s->tab_t0) * s->N;
        a.inv_tab = s->inv_tab + (uint64_t)(s->t_abs - s->tab_t0) * s->N;
        a.plan = (s->plan_tab && !s->sorted_on) ? s->plan_tab + (uint64_t)(s->t_abs - s->tab_t0) * s->N * PLAN_WORDS : nullptr;   // records BY POSITION
        a.rec_tab = a.plan;
        { static const bool no_tab = test_path("noperm");
          if (no_tab) { a.perm_tab = nullptr; a.inv_tab = nullptr; } }
        a.gamma_tab = s->gamma_tab;
        a.stamps = s->stamps;
        for (int m = 0; m < s->cfg.n_cr && m < MAX_CR; ++m)
            a.thr[m] = (uint32_t)std::floor(((double)(m + 1) / (double)s->cfg.n_cr) * 65536.0);   // dream.py:113
        a.seed = s->cfg.seed;
        a.t = t;
        a.k = (uint32_t)s->k_gen;
        a.N = s->N;
        a.lo = s->lo;
        a.upd_off = ph == 0 ? a_off : b_off;
        a.n_upd = ph == 0 ? a_n : b_n;
        a.pool_off = ph == 0 ? b_off : a_off;
        a.M = ph == 0 ? b_n : a_n;
        // mode 1 (work item = local chain, filtered by its position) is what world_size > 1 uses;
        // BPM_TEST_PATHS=mode1 selects it on one GPU so the sharded kernel path can be tested there
        static const bool force_mode1 = test_path("mode1");
        const bool by_chain = s->world > 1 || force_mode1;
        a.mode = by_chain ? 1u : 0u;
        a.n_items = by_chain ? s->n_local : a.n_upd;
        a.rec_off = a.upd_off;
        if (s->sorted_on && s->plan_tab) {
            // a rank of a world with owner-sorted records: its updates of this half generation are one contiguous run of records --
            // item k -> its k-th update: the single-GPU launch shape (no idle items, no position lookup); the other ranks' runs are
            // what the replay kernel walks
            const uint32_t grp = a.upd_off == 0 ? 0u : 1u;
            const uint32_t* cnt = s->plan_count_h + ((uint64_t)(s->t_abs - s->tab_t0) * 2 + grp) * s->world;   // (pinned copy, complete:
                                                                                                          // ensure_perm_table waited for the build)
            uint32_t acc = a.upd_off;
            for (uint32_t r = 0; r < s->world; ++r) { a.seg_off[r] = acc; acc += cnt[r]; }
            for (uint32_t r = s->world; r <= (uint32_t)MAX_SEG; ++r) a.seg_off[r] = acc;
            // (the counts size this rank's launch: a window whose slot pass did not run -- round 3's first push runs, DESIGN.md section 6 -- must be
            // an error here, not a launch of garbage size)
            if (acc > a.upd_off + a.n_upd || cnt[s->rank] > a.n_upd)
                return fail("owner-sorted update records: the counts of generation " + std::to_string((long long)s->t_abs) + " exceed its half generation (window not built?)");
            a.n_seg = s->world; a.seg_me = s->rank; a.acc_by_item = 1u;
            a.rec_sorted = s->plan_tab + (uint64_t)(s->t_abs - s->tab_t0) * s->N * PLAN_WORDS;
            a.rec_tab = a.rec_sorted + (uint64_t)a.seg_off[s->rank] * PLAN_WORDS;
            a.rec_off = 0u;
            a.n_items = cnt[s->rank];
            a.mode = 0u;
        }
        // (while the outlier check is due every few generations -- DREAM burn-in with outlier_every > 0 -- the ln-like goes by chain: the check sums
        // every chain's ln-like history, and reads the state rows of the few chains it resets where they lie: outlier_row_keys)
        const bool outlier_phase = dream && s->cfg.outlier_every > 0 && s->k_gen < s->cfg.burnin_gen;
        a.hist_by_pos = (s->hist_by_pos && (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK || g_user_cur == s) && hist_row != nullptr && !by_chain && !sync &&
                         s->trace_i32 == nullptr) ? (outlier_phase ? 2u : 1u) : 0u;
        { static const bool wt8 = test_path("wt8"); a.wt = g_wt_stores ? (wt8 ? 1u : 2u) : 0u; }
        a.lean = s->lean ? 1u : 0u;
        a.algo = (uint32_t)s->cfg.algo;
        a.P = (uint32_t)s->cfg.del_pairs;
        a.n_cr = (uint32_t)s->cfg.n_cr;
        a.adapt_on = adapt_on ? 1u : 0u;
        a.cr_gate = (s->rows_logical > s->cfg.n_cr_gen) ? 1u : 0u;          // dream.py:123
        a.hist_len = (uint32_t)s->rows_logical;
        a.gamma_scale = s->cfg.gamma_scale;
        a.gamma_demc = s->opts.gamma;
        a.epsilon = s->opts.epsilon;
        a.u_epsilon = s->opts.u_epsilon;
        a.p_snooker = s->cfg.p_snooker;
        if (s->replay_active) a.accbits = s->accbits_all + (uint64_t)s->rank * s->n_local;
        if (s->push_active) { a.peer_tab = s->tab_peerG; a.n_peers = s->world - 1u; }
        if (s->sparse_active) {
            a.pack = s->PK + (uint64_t)s->rank * s->xnsub * s->xstride();
            a.pack_cap = s->xcap;
            a.pack_nsub = s->xnsub;
            a.pack_stride = s->xstride();
        }
        a.cr_part1 = nullptr; a.cr_chunk0 = 0u; a.cr_n1 = s->cr_n1;
        a.cr_fold_part = nullptr; a.cr_fold_tot = nullptr; a.cr_fold_out = nullptr; a.cr_fold_nb = 0u;
        if (sync) {      // samplers.py:261-308: one launch, every local chain against all other chains, updates banked
            a.algo = (uint32_t)BPM_ALGO_DEMC;
            a.mode = 2u;
            a.n_items = ph == 0 ? s->n_local : 0u;
            a.perm_tab = nullptr; a.inv_tab = nullptr; a.pk.on = 0u;
            a.upd_off = 0; a.n_upd = s->N; a.pool_off = 0; a.M = s->N - 1;
            a.x_next = s->x_next;
        }
    }
    // CR reduction, level 1 (kernels.h): written by the update kernels themselves when BOTH launches of the generation take a burn-in flavour
    // (HOT 3 / 4: single GPU, work item = position, no trace ...: the predicate launch_fused applies), else computed from the slots in finish_generation
    s->gen_cr_inkernel = false;
    g_crp_launched = 0;
    g_crfold_launched = 0;
    s->gen_fold_planned = false;
    if (s->cr_p1) s->cr_p1_cur = s->cr_p1 + (size_t)(s->t_abs & 1) * 2 * MAX_CR * s->cr_n1;
#ifdef BPM_PRELOAD      // (the specialised flavours exist only in the preload build: without it every launch ends in the general kernel, which writes the slots)
    // (a host-callback sampler: only while the update kernel compiled around its HIP-source likelihood drives it -- g_user_cur -- and has the burn-in flavour)
    if (s->gen_cr_reduce && dream && (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK || (g_user_cur == s && s->user_fused_adapt != nullptr && (s->cur_args[0].rec_tab != nullptr) == s->plan_on)) &&
        s->shape.idx != SHAPE_WIDE && crp_shape(s->shape.lpc, s->shape.dpl) &&
        !test_path("nohot") && !test_path("crslots")) {
        bool ok = true;
        for (int ph = 0; ph < 2; ++ph) {
            const PhaseArgs& a = s->cur_args[ph];
            ok = ok && (a.n_items == 0 || phase_args_hot(a, true, a.rec_tab != nullptr, true));
        }
        if (ok) {
            const uint32_t n_first = (s->N + 1u) / 2u;
            for (int ph = 0; ph < 2; ++ph) {
                PhaseArgs& a = s->cur_args[ph];
                a.cr_part1 = s->cr_p1_cur;
                a.cr_chunk0 = a.upd_off == 0u ? 0u : cr_chunks_of(n_first, s->cr_g1);
            }
            s->gen_cr_inkernel = true;
        }
    }
#endif
    // a pending fold of the previous generation's CR statistics (bpm_sampler::cr_pending): consumed by this generation's first update launch when that one
    // takes a flavour that can (a burn-in flavour that sums level 1 itself: HOT 3 / 4), else dispatched now, ahead of the update launches
    if (s->cr_pending) {
        PhaseArgs& a0 = s->cur_args[0];
        if (s->gen_cr_inkernel && a0.n_items > 0 && a0.cr_part1 != nullptr) {
            a0.cr_fold_part = s->cr_pend_src; a0.cr_fold_nb = s->cr_pend_cnt; a0.cr_fold_tot = s->cr_state; a0.cr_fold_out = s->cr_state_alt;
            std::swap(s->cr_state, s->cr_state_alt);
            s->cur_args[1].cr_state = s->cr_state;      // (the second launch reads what workgroup 0 of the first one stored; the first takes p_cr from its own fold)
            s->cr_pending = false;
            s->gen_fold_planned = true;
        } else {
            CK(cr_flush_pending(s));
        }
    }
    return 0;
}

static int finish_generation(bpm_sampler* s) {
    if (s->lean) s->ll_stale = true;
    if (s->gen_adapt_on && !s->gen_cr_reduce) s->w_rows += 1;       // (the update kernels advanced the Welford moments all the same)
    if (s->gen_cr_reduce) {
        // this generation's CR statistics -> totals, p_cr (kernels.h: "CR reduction"): level 1 came out of the update kernels themselves
        // (gen_cr_inkernel) or is computed from the slots here, cr_mid_kernel passes bring more than 1024 partial sums down, cr_final_kernel folds
        const uint32_t n_cr = (uint32_t)s->cfg.n_cr;
        const int fence = s->dq_fence | bpm::DirectQueue::ACQUIRE;      // (a few hundred bytes, all through agent-scope stores: no release on these packets)
        if (g_dq) g_dq_need_acquire = false;
        // The host PREDICTED (prepare_generation) that both launches take a flavour that writes level 1 itself; what was launched decides.  Every
        // flavour that does not write level 1 writes the chains' slots (kernels.h: finish_update<..., CRP_W>), so "none did" falls back to
        // cr_level1_kernel over the slots; a mix of the two would fold stale partial sums into p_cr -- an error, never a silent result.
        bool level1_from_slots = !s->gen_cr_inkernel;
        if (s->gen_fold_planned && g_crfold_launched != 1)
            return fail("CR reduction: the first update launch of generation " + std::to_string((long long)s->t_abs) + " was handed the previous generation's fold and took a flavour that "
                        "does not fold (launch_fused's flavour choice and prepare_generation's prediction disagree)");
        if (s->gen_cr_inkernel) {
            const int want = (s->cur_args[0].n_items > 0 ? 1 : 0) + (s->cur_args[1].n_items > 0 ? 1 : 0);
            if (g_crp_launched == 0 && want > 0) level1_from_slots = true;
            else if (g_crp_launched != want)
                return fail("CR reduction: " + std::to_string(g_crp_launched) + " of " + std::to_string(want) + " update launches of generation " + std::to_string((long long)s->t_abs) +
                            " wrote their level-1 partial sums (launch_fused's flavour choice and prepare_generation's prediction disagree)");
        }
        if (level1_from_slots) {
            const PhaseArgs& a0 = s->cur_args[0];
            typedef void (*L1K)(Layout, PermKey, const uint32_t*, uint32_t, uint32_t, uint32_t, double*);
            const L1K l1 = s->cr_g1 == 4u ? cr_level1_kernel<4> : (s->cr_g1 == 16u ? cr_level1_kernel<16> : cr_level1_kernel<64>);
            if (g_dq) {
                struct { Layout L; PermKey pk; const uint32_t* perm; uint32_t N, n_cr, n1, _pad; double* part1; } ka{s->L, a0.pk, a0.perm_tab, s->N, n_cr, s->cr_n1, 0u, s->cr_p1_cur};
                const bpm::DqKernel* k = g_dq->kernel(reinterpret_cast<const void*>(l1));
                if (!k || g_dq->launch(*k, (uint32_t)(((uint64_t)s->cr_n1 * s->cr_g1 + CR_L1_THREADS - 1) / CR_L1_THREADS), 1, CR_L1_THREADS, &ka, sizeof(ka), fence) != 0)
                    return fail("direct AQL queue: cr_level1_kernel: " + g_dq->why());
            } else {
                hipLaunchKernelGGL(l1, dim3((unsigned)(((uint64_t)s->cr_n1 * s->cr_g1 + CR_L1_THREADS - 1) / CR_L1_THREADS)), dim3(CR_L1_THREADS), 0, s->stream, s->L, a0.pk, a0.perm_tab, s->N,
                                   n_cr, s->cr_n1, s->cr_p1_cur);
                HIPCK(hipGetLastError());
            }
        }
        const double* src = s->cr_p1_cur;
        uint32_t cnt = s->cr_n1;
        int flip = 0;
        while (cnt > CR_FINAL_MAX) {
            const uint32_t nn = (cnt + WAVE - 1) / WAVE;
            double* dst = s->cr_p2[flip];
            if (g_dq) {
                struct { const double* p1; uint32_t n1, n_cr, n2, _pad; double* p2; } ka{src, cnt, n_cr, nn, 0u, dst};
                const bpm::DqKernel* k = g_dq->kernel(reinterpret_cast<const void*>(cr_mid_kernel));
                if (!k || g_dq->launch(*k, nn, 1, WAVE, &ka, sizeof(ka), fence) != 0) return fail("direct AQL queue: cr_mid_kernel: " + g_dq->why());
            } else {
                hipLaunchKernelGGL(cr_mid_kernel, dim3(nn), dim3(WAVE), 0, s->stream, src, cnt, n_cr, nn, dst);
                HIPCK(hipGetLastError());
            }
            src = dst; cnt = nn; flip ^= 1;
        }
        static const bool no_defer = test_path("crnofold");
        if (!level1_from_slots && cnt <= CR_FINAL_MAX && !g_call_last_gen && !no_defer) {
            // consumer-side fold: no dispatch -- the next generation's first update launch folds (prepare_generation), or cr_flush_pending
            s->cr_pending = true;
            s->cr_pend_src = src;
            s->cr_pend_cnt = cnt;
        } else {
            CK(launch_cr_final(s, src, cnt));
        }
        s->w_rows += 1;
    }
    if (s->cfg.running_moments) CK(push_gen_sums(s, s->rows_logical));      // (the row this generation appended: index rows_logical)
    if (s->cfg.keep_history) {
        if (s->cur_args[0].hist_by_pos || s->cur_args[1].hist_by_pos)      // appended by position: generation t_abs, ln-like by chain?, shuffle switch
            s->hist_tag[(size_t)s->hist_rows] = (s->t_abs << 2) | ((s->cur_args[0].hist_by_pos | s->cur_args[1].hist_by_pos) == 2u ? 2 : 0) |
                                                (s->opts.shuffle != 0 ? 1 : 0);
        s->hist_rows += 1;
    }
    s->rows_logical += 1;
    s->k_gen += 1;      // demc.py:134
    s->t_abs += 1;
    s->outlier_due = s->cfg.algo == BPM_ALGO_DREAM && s->cfg.outlier_every > 0 && s->k_gen < s->cfg.burnin_gen &&
                     s->k_gen % s->cfg.outlier_every == 0;
    return 0;
}

// ---- generation drivers ---------------------------------------------------------------------------------
// A Group is what advances in lock-step: the single handle of a process (exchange = RCCL on its stream) or, in
// the local-group test mode, the R handles that stand for the R ranks (exchange = device copies).  Both run the
// SAME code below, so what the one-GPU tests verify is what a multi-GPU run executes, transport aside.
struct Group {
    bpm_sampler** h;
    int R;
    bool rccl;
};
constexpr int64_t SPARSE_CHUNK = 64;

// BPM_TEST_PATHS=serial (tools/emulate_ranks.py): the emulated ranks of a local group take turns on the GPU, so that a kernel
// trace shows each rank's kernels as they would run on a GPU of their own
static bool local_serial(const Group& g) {
    static const bool on = test_path("serial");
    return on && g.R > 1;
}

static int group_sync(const Group& g) {
    for (int r = 0; r < g.R; ++r) HIPCK(hipStreamSynchronize(g.h[r]->stream));
    return 0;
}

// ---- push exchange ---------------------------------------------------------------------------------------------------------
// bound of every cross-rank wait inside push_sync_kernel, in ticks of the 100 MHz constant clock (BPM_PUSH_TIMEOUT_S, default 30 s:
// ranks of a world start their step calls within seconds of each other -- the host-side classes put a communicator barrier in front)
static unsigned long long push_timeout_ticks() {
    static const double v = getenv("BPM_PUSH_TIMEOUT_S") ? atof(getenv("BPM_PUSH_TIMEOUT_S")) : 30.0;
    return (unsigned long long)(std::max(0.001, v) * 1e8);
}
// one push_sync_kernel of rank s: on the library's own queue when the generation loop runs there, else on the sampler's stream
static int launch_push_sync(bpm_sampler* s, unsigned long long seq, bool notify, bool wait) {
    const unsigned long long* ctab = s->tab_all + MAX_SEG;
    if (g_dq) {
        struct { PushCtrl* mine; const unsigned long long* ctab; uint32_t world, me; unsigned long long seq; uint32_t nf, wf; unsigned long long to; } ka{
            s->ctrl, ctab, s->world, s->rank, seq, notify ? 1u : 0u, wait ? 1u : 0u, push_timeout_ticks()};
        const bpm::DqKernel* k = g_dq->kernel(reinterpret_cast<const void*>(push_sync_kernel));
        // NO fences on this packet (the barrier bit orders it behind the update kernel, whose own packet released): the kernel reads and
        // writes nothing but the flags, with system-scope atomics that pass the caches; the update kernel that follows acquires.
        // (With acquire + release at system scope here as well a hand-over measured 10.0 us instead of 3.8 with agent-scope fences:
        // profiles/r03_push_barrier_cost.txt.)
        if (!k || g_dq->launch(*k, 1, 1, WAVE, &ka, sizeof(ka), 0) != 0)
            return fail("direct AQL queue: push_sync_kernel: " + g_dq->why());
        g_dq_need_acquire = true;
        return 0;
    }
    hipLaunchKernelGGL(push_sync_kernel, dim3(1), dim3(WAVE), 0, s->stream, s->ctrl, ctab, s->world, s->rank, seq, notify ? 1u : 0u, wait ? 1u : 0u,
                       push_timeout_ticks());
    HIPCK(hipGetLastError());
    return 0;
}
// The cross-rank barrier of the push exchange: "every rank has finished -- and pushed -- everything it enqueued before this point".
// A process per rank: one kernel announces and waits.  A local group (R handles of one process on one GPU, possibly sharing hardware
// queues): all ranks announce, the host joins the streams, all ranks wait -- the waits then find their flags set, no kernel ever
// spins on a kernel queued behind it.
static inline void bind_rank_queue(bpm_sampler* s) { if (g_group_direct) g_dq = s->dq; }
static int push_barrier(const Group& g) {
    for (int r = 0; r < g.R; ++r) g.h[r]->push_seq += 1;
    if (g.R == 1) return launch_push_sync(g.h[0], g.h[0]->push_seq, true, true);
    if (g_group_direct) {          // every rank on a queue of its own: the kernels wait for each other across the queues
        for (int r = 0; r < g.R; ++r) { bind_rank_queue(g.h[r]); CK(launch_push_sync(g.h[r], g.h[r]->push_seq, true, true)); g.h[r]->dq->flush(); }
        return 0;
    }
    for (int r = 0; r < g.R; ++r) CK(launch_push_sync(g.h[r], g.h[r]->push_seq, true, false));
    CK(group_sync(g));
    for (int r = 0; r < g.R; ++r) CK(launch_push_sync(g.h[r], g.h[r]->push_seq, false, true));
    return 0;
}
// has a wait of this rank's sync kernels run into its limit? (read with the queue / stream drained)
static int push_check_error(bpm_sampler* s) {
    if (!s->push_connected) return 0;
    unsigned long long e = 0;
    HIPCK(hipMemcpy(&e, &s->ctrl->err, sizeof(e), hipMemcpyDeviceToHost));
    if (e != 0) s->push_failed = true;
    if (e >= PUSH_ERR_CLOSED)
        return fail("push exchange: rank " + std::to_string((long long)(e - PUSH_ERR_CLOSED)) + " closed its sampler (bpm_destroy) while rank " + std::to_string(s->rank) +
                    " was still exchanging with it: the ranks of a world must make the same sequence of calls; the replicas are no longer consistent");
    if (e != 0) return fail("push exchange: rank " + std::to_string(s->rank) + " waited longer than the limit for rank " + std::to_string((long long)e - 1) +
                            " (a rank that died, or ranks that entered bpm_step more than BPM_PUSH_TIMEOUT_S apart); the replicas are no longer consistent");
    return 0;
}

// The shuffle keys of the history rows for the outlier check's kernels (identity for what lies in chain order): state rows and ln-like rows have
// a key each (hist_tag).  On the sampler's stream, from a host vector that outlives the copy.
static int outlier_row_keys(bpm_sampler* s) {
    s->okeys_x = s->okeys_ll = false;
    if (!s->hist_by_pos) return 0;
    const size_t rows = (size_t)s->hist_rows;
    HIPCK(hipStreamSynchronize(s->stream));      // (the previous check's kernels and copy are done with okeys / okeys_host)
    if (rows > s->okeys_cap) {
        if (s->okeys) HIPCK(hipFree(s->okeys));
        s->okeys = nullptr;
        s->okeys_cap = std::max<size_t>(rows + rows / 2, 256);
        HIPCK(hipMalloc(&s->okeys, 2 * s->okeys_cap * sizeof(PermKey)));
    }
    PermKey id{};
    id.n = s->N; id.nbits = perm_nbits(s->N); id.on = 0u;
    s->okeys_host.assign(2 * s->okeys_cap, id);
    for (size_t r = 0; r < rows; ++r) {
        const int64_t tag = r < s->hist_tag.size() ? s->hist_tag[r] : -1;
        if (tag < 0) continue;
        const PermKey k = make_perm_key(s->cfg.seed, (uint64_t)(tag >> 2), s->N, (tag & 1) != 0);
        s->okeys_host[r] = k;
        s->okeys_x = true;
        if ((tag & 2) == 0) { s->okeys_host[s->okeys_cap + r] = k; s->okeys_ll = true; }
    }
    if (s->okeys_x) HIPCK(hipMemcpyAsync(s->okeys, s->okeys_host.data(), 2 * s->okeys_cap * sizeof(PermKey), hipMemcpyHostToDevice, s->stream));
    return 0;
}
// the eight passes of the radix select (+ first maximum) over s->om -> s->sel
static void launch_outlier_select(bpm_sampler* s, const SelRanks& R) {
    const uint32_t nblk = sel_blocks(s->N);
    for (int pass = 0; pass < 8; ++pass)
        hipLaunchKernelGGL(outlier_select_pass_kernel, dim3(nblk), dim3(SEL_THREADS), 0, s->stream, s->om, s->n_local, s->N, pass, R,
                           reinterpret_cast<SelState*>(s->sel_state), s->sel);
}

// DREAM outlier-chain reset (Vrugt et al. 2009; extension): chains whose mean ln_like over the last half of their history
// lies below Q1 - 2 IQR (quartiles over all N chains, np.percentile's interpolation) restart from the best chain's state.
// Everything stays on the device and on the samplers' streams: omega of the local chains, all-gather of the (omega | ln_like)
// blocks (RCCL, or device copies in a local group), radix select of the four order statistics + first argmax, reset.
static int group_outlier_check(const Group& g) {
    bpm_sampler* s0 = g.h[0];
    const uint32_t N = s0->N;
    for (int r = 0; r < g.R; ++r) {
        bpm_sampler* s = g.h[r];
        s->outlier_due = false;
        if (!s->cfg.keep_history || s->hist_rows != s->rows_logical) return fail("outlier detection needs the chain history");
        CK(outlier_row_keys(s));                             // (rows appended by position are read where they lie)
        CK(refresh_ll(s));                                   // (the (omega | ln-like) block carries the chains' current ln-like)
        const uint32_t rows = (uint32_t)s->hist_rows, r0 = rows / 2;
        hipLaunchKernelGGL(outlier_omega_kernel, dim3((s->n_local + 255) / 256), dim3(256), 0, s->stream, s->llhist, s->ll, s->n_local, r0, rows,
                           s->okeys_ll ? (const PermKey*)(s->okeys + s->okeys_cap) : (const PermKey*)nullptr, s->om + (size_t)s->rank * 2 * s->n_local);
        HIPCK(hipGetLastError());
    }
    if (s0->push_active) {
        // push exchange: every rank writes its (omega | ln-like) block into the others' om buffers, then the cross-rank barrier
        for (int r = 0; r < g.R; ++r) {
            bpm_sampler* s = g.h[r];
            const uint32_t n = 2u * s->n_local;
            hipLaunchKernelGGL(push_copy_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s->stream, s->om + (size_t)s->rank * n,
                               (const unsigned long long*)(s->tab_all + 2 * MAX_SEG), s->world, s->rank, (uint64_t)s->rank * n, n);
            HIPCK(hipGetLastError());
        }
        CK(push_barrier(g));
    } else if (g.rccl) {
        if (s0->world > 1 || s0->comm)
            NCCLCK(g_rccl.AllGather(s0->om + (size_t)s0->rank * 2 * s0->n_local, s0->om, (size_t)2 * s0->n_local, ncclDouble, s0->comm, s0->stream));
    } else if (g.R > 1) {
        CK(group_sync(g));
        const size_t blk = (size_t)2 * s0->n_local;
        for (int r = 0; r < g.R; ++r)
            for (int o = 0; o < g.R; ++o)
                if (o != r)
                    HIPCK(hipMemcpyAsync(g.h[o]->om + (size_t)r * blk, g.h[r]->om + (size_t)r * blk, blk * sizeof(double), hipMemcpyDeviceToDevice,
                                         g.h[r]->stream));
        CK(group_sync(g));
    }
    // np.percentile(omega, [25, 75]): position q (N - 1) / 100, the two order statistics around it and the fraction between them
    SelRanks R;
    double tq[2];
    for (int i = 0; i < 2; ++i) {
        const double pos = (i == 0 ? 25.0 : 75.0) / 100.0 * (double)(N - 1);
        const uint32_t lo = (uint32_t)std::floor(pos);
        R.k[2 * i] = lo;
        R.k[2 * i + 1] = std::min(lo + 1u, N - 1u);
        tq[i] = pos - (double)lo;
    }
    for (int r = 0; r < g.R; ++r) {
        bpm_sampler* s = g.h[r];
        launch_outlier_select(s, R);
        HIPCK(hipMemsetAsync(s->olist, 0, sizeof(uint32_t), s->stream));
        hipLaunchKernelGGL(outlier_reset_kernel, dim3((N + WAVE - 1) / WAVE), dim3(WAVE), 0, s->stream, s->L, N, s->lo, s->om, s->sel, tq[0], tq[1],
                           s->ll, s->llhist, (uint32_t)s->hist_rows, s->okeys_ll ? (const PermKey*)(s->okeys + s->okeys_cap) : (const PermKey*)nullptr,
                           s->olist, s->counters + 4);
        hipLaunchKernelGGL(outlier_rebuild_kernel, dim3(std::min<uint32_t>(s->n_local, 4096u)), dim3(WAVE), 0, s->stream, s->L, (const double*)s->sel,
                           (const uint32_t*)s->olist, s->hist, (uint32_t)s->hist_rows, s->okeys_x ? (const PermKey*)s->okeys : (const PermKey*)nullptr,
                           s->w_rows == s->rows_logical ? s->w_mean : (double*)nullptr, s->w_m2);
        HIPCK(hipGetLastError());
        if (local_serial(g)) HIPCK(hipStreamSynchronize(s->stream));
    }
    // (a reset rewrites rows of the generation just appended: its population sums follow)
    for (int r = 0; r < g.R; ++r) CK(push_gen_sums(g.h[r], g.h[r]->rows_logical - 1));
    // push exchange: nobody starts the next generation (whose accepted rows land in THIS replica) while a reset kernel still writes
    if (s0->push_active) CK(push_barrier(g));
    return 0;
}

// demc.py:93-94,116-117: every rank's block into every replica
static int exchange_dense(const Group& g) {
    if (g.rccl) return allgather_state(g.h[0]);
    if (g.R == 1) return 0;
    CK(group_sync(g));
    const bpm_sampler* s0 = g.h[0];
    for (int r = 0; r < g.R; ++r)
        for (int o = 0; o < g.R; ++o)
            if (o != r)
                HIPCK(hipMemcpyAsync(g.h[o]->G + (uint64_t)r * s0->L.blk, g.h[r]->G + (uint64_t)r * s0->L.blk,
                                     (size_t)s0->L.blk * sizeof(double), hipMemcpyDeviceToDevice, g.h[r]->stream));
    return group_sync(g);
}

// the same exchange with only the accepted rows: all-gather of the packed blocks, then scatter into the replicas
static int exchange_sparse(const Group& g) {
    bpm_sampler* s0 = g.h[0];
    const uint32_t cap = s0->xcap;
    const uint64_t S = (uint64_t)s0->xnsub * s0->xstride();
    if (g.rccl) {
        NCCLCK(g_rccl.AllGather(s0->PK + (uint64_t)s0->rank * S, s0->PK, (size_t)S, ncclDouble, s0->comm, s0->stream));
    } else {
        CK(group_sync(g));
        for (int r = 0; r < g.R; ++r)
            for (int o = 0; o < g.R; ++o)
                if (o != r)
                    HIPCK(hipMemcpyAsync(g.h[o]->PK + (uint64_t)r * S, g.h[r]->PK + (uint64_t)r * S, (size_t)S * sizeof(double),
                                         hipMemcpyDeviceToDevice, g.h[r]->stream));
        CK(group_sync(g));
    }
    for (int r = 0; r < g.R; ++r) {
        bpm_sampler* s = g.h[r];
        hipLaunchKernelGGL(exchange_scatter_kernel, dim3(cap * s->xnsub, s->world), dim3(WAVE), 0, s->stream, s->L, s->PK, s->xnsub,
                           s->xstride(), cap, s->rank, s->xstat);
        if (local_serial(g)) HIPCK(hipStreamSynchronize(s->stream));
    }
    HIPCK(hipGetLastError());
    return 0;
}

static void launch_replay_any(bpm_sampler* s, const PhaseArgs& a) {
    const int v = s->cfg.algo != BPM_ALGO_DREAM ? 0 : (s->cfg.del_pairs == 3 ? 1 : 2);
    if (a.rec_sorted && s->shape.idx >= 3 && s->shape.idx < SHAPE_WIDE) g_replay_sorted[v][s->shape.idx - 3](a, s->stream);
    else g_replay[v][s->shape.idx](a, s->stream);
}

// demc.py:93-94,116-117 with one byte per local chain on the wire: all-gather of the accept bytes, then every rank
// recomputes the accepted updates of the other ranks' chains of this half generation into its replica
static int exchange_replay(const Group& g, int ph) {
    bpm_sampler* s0 = g.h[0];
    if (g.rccl) {
        NCCLCK(g_rccl.AllGather(s0->accbits_all + (uint64_t)s0->rank * s0->n_local, s0->accbits_all, (size_t)s0->n_local, ncclUint8,
                                s0->comm, s0->stream));
    } else if (g.R > 1) {
        CK(group_sync(g));
        for (int r = 0; r < g.R; ++r)
            for (int o = 0; o < g.R; ++o)
                if (o != r)
                    HIPCK(hipMemcpyAsync(g.h[o]->accbits_all + (uint64_t)r * s0->n_local, g.h[r]->accbits_all + (uint64_t)r * s0->n_local,
                                         (size_t)s0->n_local, hipMemcpyDeviceToDevice, g.h[r]->stream));
        CK(group_sync(g));
    }
    for (int r = 0; r < g.R; ++r) {
        bpm_sampler* s = g.h[r];
        if (s->world == 1) continue;                        // (one-rank communicator: nobody else's chains)
        PhaseArgs a = s->cur_args[ph];
        if (a.n_upd == 0) continue;
        a.replay = 1u;
        a.accbits = nullptr;
        a.accbits_all = s->accbits_all;
        trace_set(a, nullptr, nullptr, nullptr);
        a.pack = nullptr; a.hist_row = nullptr; a.llhist_row = nullptr; a.adapt_on = 0u;
        launch_replay_any(s, a);
        if (local_serial(g)) HIPCK(hipStreamSynchronize(s->stream));
    }
    HIPCK(hipGetLastError());
    return 0;
}

// xmode: how the half generations' updates reach the other ranks: 0 dense all-gather, 1 accepted rows, 2 accept bytes + replay,
// 3 pushed by their owners from inside the update kernel (then only the cross-rank barrier follows a half generation)
static int group_generation(const Group& g, int64_t n_ahead, int xmode, PhaseLaunch fn) {
    const long long tp0 = g_host_timing ? now_ns() : 0;
    for (int r = 0; r < g.R; ++r) {
        g.h[r]->sparse_active = xmode == 1;
        g.h[r]->replay_active = xmode == 2;
        g.h[r]->push_active = xmode == 3;
        bind_rank_queue(g.h[r]);
        CK(prepare_generation(g.h[r], n_ahead));
    }
    if (g_host_timing) g_ns_prepare += now_ns() - tp0;
    if (g.h[0]->cfg.algo == BPM_ALGO_DEMC_SYNC) {
        for (int r = 0; r < g.R; ++r) {
            bpm_sampler* s = g.h[r];
            fn(s->cur_args[0], s->stream);
            HIPCK(hipMemcpyAsync(s->G + (uint64_t)s->rank * s->L.blk, s->x_next, (size_t)s->n_local * s->ld * sizeof(double),
                                 hipMemcpyDeviceToDevice, s->stream));      // apply the banked updates
        }
        HIPCK(hipGetLastError());
        CK(exchange_dense(g));
    } else {
        for (int ph = 0; ph < 2; ++ph) {
            for (int r = 0; r < g.R; ++r) {
                bpm_sampler* s = g.h[r];
                bind_rank_queue(s);
                if (s->cur_args[ph].n_items > 0) {
                    g_dq_release_this = g_dq && g_dq_call_last_gen && (ph == 1 || s->cur_args[1].n_items == 0);
                    if (g_dq && s->timed_last_gen >= 0) {
                        // direct mode: attaching a time stamp costs the host nothing -- the first and the last update dispatch of the call
                        if (s->timed_want_first) { s->timed_want_first = false; g_dq_sig = 0; s->timed_l0 = g_timed_launches; }
                        else if (s->timed_last_gen == s->t_abs && (ph == 1 || s->cur_args[1].n_items == 0)) { g_dq_sig = 1; s->timed_l1 = g_timed_launches; }
                    } else if (s->timed_want_first && s->timed_skip > 0) {
                        --s->timed_skip;
                    } else if (s->timed_want_first) {
                        s->timed_want_first = false; g_stop_event = s->ev0; s->timed_l0 = g_timed_launches;
                    } else if (s->timed_last_gen == s->t_abs && (ph == 1 || s->cur_args[1].n_items == 0)) {
                        g_stop_event = s->ev1; s->timed_l1 = g_timed_launches;
                    }
                    const long long tl0 = g_host_timing ? now_ns() : 0;
                    fn(s->cur_args[ph], s->stream);
                    if (g_host_timing) { const long long tl1 = now_ns(); g_ns_launch += tl1 - tl0; ++g_n_launch; if (s->timed_last_gen >= 0) { g_launch_log.push_back(tl0); g_launch_log.push_back(tl1); } }
                }
                if (local_serial(g)) HIPCK(hipStreamSynchronize(g.h[r]->stream));
            }
            HIPCK(hipGetLastError());
            CK(xmode == 3 ? push_barrier(g) : (xmode == 2 ? exchange_replay(g, ph) : (xmode == 1 ? exchange_sparse(g) : exchange_dense(g))));
        }
    }
    for (int r = 0; r < g.R; ++r) { bind_rank_queue(g.h[r]); CK(finish_generation(g.h[r])); }
    // push exchange during CR adaptation: the next generation's updates write their (delta, cr) slots into every replica; no rank may
    // get there while another rank's reduction kernels still read this generation's slots
    if (xmode == 3 && g.h[0]->gen_cr_reduce) CK(push_barrier(g));
    if (g.h[0]->outlier_due) {
        // everything of the check, its cross-rank barriers included, runs on the HIP streams: queues drained before, streams after
        std::vector<StreamSection> secs;
        for (int r = 0; r < (g_group_direct ? g.R : 1); ++r) { bind_rank_queue(g.h[r]); secs.emplace_back(g.h[r]); CK(secs.back().rc); }
        bpm::DirectQueue* const dq_saved = g_dq;
        const bool gd_saved = g_group_direct;
        g_dq = nullptr; g_group_direct = false;
        const int rc_out = group_outlier_check(g);
        g_dq = dq_saved; g_group_direct = gd_saved;
        CK(rc_out);
        for (auto& sec : secs) CK(sec.end());
    }
    return 0;
}

struct HostCkpt {
    int64_t k_gen, t_abs, hist_rows, rows_logical, w_rows;
};

#ifdef BPM_EXPERIMENT_XCD
// EXPERIMENT (never the product): up to `want_gens` steady-state generations of a single-GPU sampler as ONE launch of xcd_resident_kernel -- the
// host walks prepare_generation / finish_generation for every generation of the batch (a batch never crosses a table window), collects the two
// argument blocks of each, copies them to the device and launches on the HIP stream (acquire + release around the one kernel).
static int xcd_mode() { static const int m = getenv("BPM_XCD") ? atoi(getenv("BPM_XCD")) : 0; return m; }
static bool xcd_eligible(const Group& g) {
    bpm_sampler* s = g.h[0];
    if (xcd_mode() == 0 || g.R != 1 || s->world != 1 || s->comm || s->trace_i32 || s->cfg.running_moments || s->cfg.outlier_every > 0) return false;
    if (s->cfg.algo == BPM_ALGO_DREAM && s->cfg.burnin_gen > s->k_gen) return false;
    if (s->cfg.algo == BPM_ALGO_DEMC && s->cfg.target_id == BPM_TARGET_BANANA_2D && s->shape.idx == 0) return true;
    if (s->cfg.algo == BPM_ALGO_DREAM && s->cfg.target_id == BPM_TARGET_MIXTURE_PAIRS && s->shape.idx == 1 && s->cfg.del_pairs == 3 && s->cfg.n_cr == 3) return true;
    return false;
}
static int xcd_batch(bpm_sampler* s, int64_t want_gens, int64_t* did) {
    CK(leave_direct(s));
    constexpr int MAX_PH = 2 * PERM_CHUNK;
    if (!s->xcd_args) {
        CK(dev_alloc(&s->xcd_args, (size_t)MAX_PH));
        CK(dev_alloc(&s->xcd_ctl, 16));
        HIPCK(hipMemsetAsync(s->xcd_ctl, 0, 16 * sizeof(uint32_t), s->stream));
    }
    bpm::DirectQueue* const dq_saved = g_dq;
    g_dq = nullptr; g_wt_stores = false;
    std::vector<PhaseArgs> host;
    int64_t n = 0;
    while (n < want_gens) {
        CK(prepare_generation(s, want_gens - n));
        for (int ph = 0; ph < 2; ++ph) if (s->cur_args[ph].n_items > 0) host.push_back(s->cur_args[ph]);
        CK(finish_generation(s));
        ++n;
        if (s->t_abs >= s->tab_t0 + s->tab_K) break;          // the next generation belongs to the next window of the tables
    }
    g_dq = dq_saved;
    const uint32_t want = 32u, xcc = (uint32_t)(xcd_mode() >> 4) & 7u;
    const unsigned long long timeout = 50000000ull;          // 0.5 s of the 100 MHz clock
    HIPCK(hipMemcpyAsync(s->xcd_args, host.data(), host.size() * sizeof(PhaseArgs), hipMemcpyHostToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));                   // (host is a local vector)
    HIPCK(hipMemsetAsync(s->xcd_ctl, 0, 3 * sizeof(uint32_t), s->stream));
    if ((xcd_mode() & 15) == 2) {
        hipLaunchKernelGGL(xcd_barrier_only_kernel, dim3(256), dim3(1024), 0, s->stream, (uint32_t)host.size(), s->xcd_ctl, want, xcc, timeout);
    } else if (s->cfg.algo == BPM_ALGO_DEMC) {
        hipLaunchKernelGGL((xcd_resident_kernel<ALGO_DEMC, TARGET_BANANA, 1, 2, 1>), dim3(256), dim3(1024), 0, s->stream, (const PhaseArgs*)s->xcd_args, (uint32_t)host.size(),
                           s->xcd_ctl, want, xcc, timeout);
    } else {
        hipLaunchKernelGGL((xcd_resident_kernel<ALGO_DREAM, TARGET_MIXTURE, 4, 2, 3>), dim3(256), dim3(1024), 0, s->stream, (const PhaseArgs*)s->xcd_args, (uint32_t)host.size(),
                           s->xcd_ctl, want, xcc, timeout);
    }
    HIPCK(hipGetLastError());
    s->xcd_gens += n;
    *did = n;
    return 0;
}
static int xcd_check(bpm_sampler* s) {
    if (!s->xcd_ctl) return 0;
    uint32_t c[6];
    HIPCK(hipMemcpyAsync(c, s->xcd_ctl, sizeof(c), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    if (c[3] != 0 || (c[2] != 32u && (xcd_mode() & 15) != 2))
        return fail("XCD-resident experiment: the workers did not meet (registered " + std::to_string(c[0]) + " workgroups on the chosen XCD, workers " + std::to_string(c[2]) +
                    ", error " + std::to_string(c[3]) + ", XCC ids seen mask " + std::to_string(c[4]) + "): results of this call are invalid");
    return 0;
}
#endif

// Does this group's generation loop go through the library's own AQL queue(s)?  A single GPU's sampler, or -- with the push exchange, where
// nothing of the loop is a collective call -- a rank of a world; group_direct: the R ranks of a local group, each on a queue of its own.
// (run_generations and the arena self-test of bpm_push_selftest ask the same question: the probe must take the path the updates take.)
static bool group_goes_direct(const Group& g, bool push, bool& group_direct) {
    bpm_sampler* s0 = g.h[0];
    group_direct = push && g.R > 1 && !g_host_timing && !local_serial(g);
    for (int r = 0; r < g.R && group_direct; ++r)
        group_direct = g.h[r]->dq_private && g.h[r]->dq_enabled && !g.h[r]->dq->failed() && !g.h[r]->trace_i32 && !g.h[r]->stamps && !g.h[r]->shape_needs_scratch;
    return group_direct ||
           (s0->dq && s0->dq_enabled && g.R == 1 && ((!g.rccl && s0->world == 1) || push) && (!s0->local_group || s0->dq_private) &&
            s0->cfg.algo != BPM_ALGO_DEMC_SYNC && !s0->trace_i32 && !s0->stamps && !g_host_timing && !s0->dq->failed() && !s0->shape_needs_scratch);
}

static int run_generations_user(bpm_sampler* s, int64_t n_gens);      // (a host-callback sampler with a device likelihood: below, beside the host-callback core)
// the update kernel compiled at run time around a caller's likelihood (bpm_sampler::user_fused_fn): the general instantiation's launch, from a module
static void launch_user_fused(const PhaseArgs& a, hipStream_t st) {
    bpm_sampler* s = g_user_cur;
    if (!s || !s->user_fused_fn) { g_user_launch_failed = true; return; }      // (never: run_generations routes here only with the module in place -- and says so if not)
    PhaseArgs ka = a;
    ka.tparams = s->user_params;                   // the caller's parameter block is the target's
    size_t sz = sizeof(ka);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    static const bool no_hot = test_path("nohot");
    // burn-in: prepare_generation handed this launch the level-1 sums to write (cr_part1) only after phase_args_hot(a, true, plan, true) held for BOTH launches
    const bool adapt = a.cr_part1 != nullptr && s->user_fused_adapt != nullptr;
    if (adapt) { ++g_crp_launched; if (a.cr_fold_part != nullptr) ++g_crfold_launched; }
    const unsigned block = adapt ? s->user_fused_block_adapt : s->user_fused_block, cpw = block / (unsigned)s->shape.lpc, grid = (a.n_items + cpw - 1u) / cpw;
    // (the steady-state instantiation was compiled for update records exactly when this sampler builds them: HOT 1 / HOT 2)
    const bool hot = !adapt && !no_hot && s->user_fused_hot && (a.rec_tab != nullptr) == s->plan_on && phase_args_hot(a, s->cfg.algo == BPM_ALGO_DREAM, a.rec_tab != nullptr, false);
    if (g_dq) {      // the library's own queue (launch_packed's packet: the module's kernels take the argument block alone, no preloaded leading arguments)
        const bpm::DqKernel* k = g_dq->kernel_by_name(s->user_fused_names[adapt ? 2 : (hot ? 1 : 0)]);
        const int sig = g_dq_sig; g_dq_sig = -1;
        int fence = g_dq_update_fence;
        if (g_dq_need_acquire) { fence |= bpm::DirectQueue::ACQUIRE; g_dq_need_acquire = false; }
        if (g_dq_release_this) { fence |= bpm::DirectQueue::RELEASE; g_dq_release_this = false; }
        if (!k || g_dq->launch(*k, grid, 1, block, &ka, sizeof(ka), fence, sig) != 0) g_dq_error = true;
        ++g_timed_launches; ++g_n_direct;
        return;
    }
    (void)hipExtModuleLaunchKernel(adapt ? s->user_fused_adapt : (hot ? s->user_fused_hot : s->user_fused_fn), grid * block, 1, 1, block, 1, 1, 0, st, nullptr, extra, nullptr, take_stop_event(), 0);
    ++g_timed_launches; ++g_n_stream;
}
static int run_generations(const Group& g, int64_t n_gens) {
    bpm_sampler* s0 = g.h[0];
    // a caller's likelihood compiled from HIP source (bpm_set_device_likelihood): the update kernel compiled around it runs the ordinary generation loop
    // (one launch per half generation, on the HIP stream); without it, the proposal / likelihood / commit kernels of run_generations_user
    const bool user_fused = s0->cfg.target_id == BPM_TARGET_HOST_CALLBACK && s0->user_fused_fn != nullptr && g.R == 1;
    if (s0->cfg.target_id == BPM_TARGET_HOST_CALLBACK && s0->user_fn && g.R == 1 && !user_fused) return run_generations_user(s0, n_gens);
    if (user_fused && s0->world > 1 && !s0->comm) return fail("bpm_step: a host-callback sampler of a world needs an RCCL communicator (create it with a unique id)");
    g_user_cur = user_fused ? s0 : nullptr;
    struct UserCurReset { ~UserCurReset() { g_user_cur = nullptr; } } user_cur_reset;      // (prepare_generation asks g_user_cur who drives a host-callback sampler)
    PhaseLaunch fn = user_fused ? launch_user_fused : pick_fused(s0);
    if (!fn) return fail("bpm_step: host-callback target must be driven with bpm_propose / bpm_commit (or give it a device likelihood: bpm_set_device_likelihood)");
    const bool dream = s0->cfg.algo == BPM_ALGO_DREAM;
    int64_t done = 0;
    const bool push = s0->push_enabled && s0->push_connected;
    if (s0->world > 1 && !push && !g.rccl && g.R == 1) return fail("bpm_step: this rank has no exchange yet: call bpm_push_connect (and bpm_set_exchange(h, 3, 0)) on every rank first");
    bool push_entered = false;
    while (done < n_gens) {
        const bool adapting = dream && s0->cfg.burnin_gen > s0->k_gen;          // dream.py:92: CR statistics travel in the dense block
#ifdef BPM_EXPERIMENT_XCD
        if (xcd_eligible(g)) {
            int64_t did = 0;
            CK(xcd_batch(s0, n_gens - done, &did));
            done += did;
            if (done >= n_gens) CK(xcd_check(s0));
            continue;
        }
#endif
        if (push || !(s0->sparse_enabled && !adapting)) {
            // (HIP-graph replay of steady-state chunks was built and measured in round 1 -- 13.1 vs 12.5 us per generation at cfg2 -- and
            // removed in round 3: DESIGN.md section 5 item 6)
            const bool replay = s0->replay_enabled && !adapting;       // burn-in: delta / cr_idx of every chain travel in the dense block
            // Direct mode: a single GPU's generation loop (the two update kernels, during burn-in the CR reduction, once per window
            // the table build) is dispatched through the library's own AQL queue.  The outlier check and the moment rebuild (rare) run
            // on the HIP stream between two drains (StreamSection); the synchronous mode, tracing and everything with an exchange
            // stay on the stream altogether.
            // With the push exchange a rank of a world runs on its own queue too: nothing of its generation loop is a collective call.
            bool group_direct = false;
            // (a run-time module's kernels go through the library's queue when it found them by name, else on the stream)
            const bool direct = group_goes_direct(g, push, group_direct) && (!user_fused || s0->user_fused_dq);
            for (int r = 0; r < g.R; ++r) {
                bpm_sampler* s = g.h[r];
                if (direct && !s->dq_active) {
                    CK(wait_stream(s->stream));                         // what the stream still holds (burn-in, table builds) comes first
                    s->dq_active = true;
                    g_dq_need_acquire = true;
                } else if (!direct && s->dq_active) {
                    CK(leave_direct(s));
                }
            }
            g_group_direct = group_direct;
            g_dq = direct ? s0->dq : nullptr;
            g_dq_error = false;
            // (half a generation rewrites at most N/2 + 1 rows)
            // Release-less packets + write-through stores in every generation, burn-in included, at every size: with ONE 16-byte sc1 store
            // per lane (store_row_wt16) the write-through form beats the end-of-kernel release everywhere it was measured -- round 2's two
            // 8-byte atomic stores lost above 4 MiB per half generation and in burn-in, hence its size rule and the fenced burn-in
            // (profiles/r03_write_through_16B.txt: cfg2 burn-in 23.1 -> 21.9 us, cfg5 57.6 -> 55.9, cfg5 burn-in 90 -> 81, N = 65536 x d = 100 70.0 -> 68.7).
            // BPM_TEST_PATHS=wt8 selects the 8-byte form, bpm_set_launch_path(h, 1, 3) the fenced packets with plain stores.
            const bool plain_stores = false;
            g_dq_update_fence = plain_stores ? (int)bpm::DirectQueue::FENCED : s0->dq_fence;
            // push exchange: every update packet acquires and releases at SYSTEM scope (rows go to and come from other agents)
            // push exchange: system scope -- every update packet acquires and releases at SYSTEM scope, plain stores into the own replica (rows
            // go to and come from other agents); agent scope -- the single-GPU form: acquire only, write-through stores into the own replica
            if (push && !s0->push_agent_scope) g_dq_update_fence = bpm::DirectQueue::FENCED | bpm::DirectQueue::SYSTEM;
            g_wt_stores = !plain_stores && !s0->coherent && direct && !(g_dq_update_fence & bpm::DirectQueue::RELEASE);
            g_dq_call_last_gen = direct && done == n_gens - 1;
            g_call_last_gen = done == n_gens - 1;
            if (push && !push_entered) {
                // entry barrier of the call: a peer's first update kernel may push into THIS replica only after this rank has finished
                // whatever its host did to the replica before the call (bpm_set_state, a warm start, ...) -- and vice versa
                for (int r = 0; r < g.R; ++r) g.h[r]->push_active = true;
                CK(push_barrier(g));
                push_entered = true;
            }
            const int rc_gen = group_generation(g, n_gens - done, push ? 3 : (replay ? 2 : 0), fn);
            g_wt_stores = false; g_dq_call_last_gen = false; g_dq_release_this = false; g_call_last_gen = false;
            g_group_direct = false;
            if (direct) {
                g_dq = nullptr;
                for (int r = 0; r < (group_direct ? g.R : 1); ++r) {
                    bpm_sampler* s = g.h[r];
                    s->dq->flush();                                     // one doorbell per generation
                    if (g_dq_error || s->dq->failed()) return fail("direct AQL queue: " + (s->dq->why().empty() ? std::string("update kernel not found among the loaded code objects") : s->dq->why()));
                }
            }
            CK(rc_gen);
            if (g_user_launch_failed) { g_user_launch_failed = false; return fail("bpm_step: the update kernel compiled around the HIP-source likelihood was not launched (module gone)"); }
            if (push) for (int r = 0; r < g.R; ++r) g.h[r]->n_push_gens += 1;
            else if (replay) for (int r = 0; r < g.R; ++r) g.h[r]->n_replay_gens += 1;
            ++done;
            continue;
        }
        // ---- a chunk of generations with the sparse exchange, under a checkpoint
        const int64_t K = std::min<int64_t>(SPARSE_CHUNK, n_gens - done);
        std::vector<HostCkpt> hc((size_t)g.R);
        for (int r = 0; r < g.R; ++r) {
            bpm_sampler* s = g.h[r];
            hc[(size_t)r] = HostCkpt{s->k_gen, s->t_abs, s->hist_rows, s->rows_logical, s->w_rows};
            HIPCK(hipMemcpyAsync(s->ckpt_G, s->G, (size_t)s->world * s->L.blk * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
            HIPCK(hipMemcpyAsync(s->ckpt_ll, s->ll, (size_t)s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
            HIPCK(hipMemcpyAsync(s->ckpt_acc, s->acc_count, ((size_t)s->n_local + ACC_SHARDS) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
            HIPCK(hipMemcpyAsync(s->ckpt_counters, s->counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s->stream));
            HIPCK(hipMemsetAsync(s->xstat, 0, 2 * sizeof(uint32_t), s->stream));
            for (uint32_t b = 0; b < s->xnsub; ++b)      // own counters armed (the capacity, hence the layout, may have changed)
                HIPCK(hipMemsetAsync(s->PK + ((uint64_t)s->rank * s->xnsub + b) * s->xstride(), 0, sizeof(double), s->stream));
        }
        for (int64_t i = 0; i < K; ++i) CK(group_generation(g, n_gens - done - i, 1, fn));
        bool overflow = false;
        uint32_t maxc = 0;
        for (int r = 0; r < g.R; ++r) {
            bpm_sampler* s = g.h[r];
            uint32_t xs[2] = {0, 0};
            HIPCK(hipMemcpyAsync(xs, s->xstat, sizeof(xs), hipMemcpyDeviceToHost, s->stream));
            HIPCK(hipStreamSynchronize(s->stream));
            overflow = overflow || xs[0] != 0;
            maxc = std::max(maxc, xs[1]);
            s->n_sparse_chunks += 1;
        }
        if (overflow) {
            // some rank accepted more rows than a packed block holds: replicas diverged.  Roll every rank back to the
            // checkpoint and replay the chunk with the dense exchange; the draws are counter-addressed, so the replay
            // is exactly the run that would have happened without the sparse exchange.
            for (int r = 0; r < g.R; ++r) {
                bpm_sampler* s = g.h[r];
                HIPCK(hipMemcpyAsync(s->G, s->ckpt_G, (size_t)s->world * s->L.blk * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
                HIPCK(hipMemcpyAsync(s->ll, s->ckpt_ll, (size_t)s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
                HIPCK(hipMemcpyAsync(s->acc_count, s->ckpt_acc, ((size_t)s->n_local + ACC_SHARDS) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
                HIPCK(hipMemcpyAsync(s->counters, s->ckpt_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s->stream));
                const HostCkpt& k = hc[(size_t)r];
                s->k_gen = k.k_gen; s->t_abs = k.t_abs; s->hist_rows = k.hist_rows; s->rows_logical = k.rows_logical; s->w_rows = k.w_rows;
                s->n_sparse_replays += 1;
            }
            for (int64_t i = 0; i < K; ++i) CK(group_generation(g, n_gens - done - i, 0, fn));
        }
        // capacity for the next chunk from the largest sub-block count seen (identical on every rank: all see all
        // counters): a margin of ~3 sigma of a Poisson count on top of an observed maximum over >= 64 half generations
        const uint32_t want = overflow ? 2u * maxc + 8u : maxc + 3u * (uint32_t)std::ceil(std::sqrt((double)maxc)) + 4u;
        for (int r = 0; r < g.R; ++r) {
            bpm_sampler* s = g.h[r];
            s->xcap = std::min<uint32_t>(s->xcap_max, (want + 1u) & ~1u);
        }
        done += K;
    }
    for (int r = 0; r < g.R; ++r) { g.h[r]->sparse_active = false; g.h[r]->replay_active = false; g.h[r]->push_active = false; }
    return 0;
}

#ifdef BPM_TEST_HOOKS
// Lock-step driver of a local group (see bpm_create): handles[r] = rank r of R, all on one GPU.  Test variant only (include/bipymc_hip_test.h).
extern "C" int bpm_local_group_step(bpm_handle_t* handles, int32_t R, int64_t n_gens) {
    if (!handles || R < 1) return fail("bpm_local_group_step: bad argument");
    for (int r = 0; r < R; ++r) {
        bpm_sampler* s = handles[r];
        CK(check_handle(s));
        if (!s->local_group || (int)s->world != R || (int)s->rank != r) return fail("bpm_local_group_step: handles are not ranks 0..R-1 of one local group");
        if (!s->run_open) return fail("bpm_local_group_step: call bpm_begin_run on every rank first");
        if (s->cfg.keep_history) CK(ensure_history(s, s->hist_rows + n_gens));
    CK(ensure_gen_sums(s, s->rows_logical + n_gens));
    }
    CK(set_device(handles[0]));
    Group g{handles, R, false};
    CK(run_generations(g, n_gens));
    for (int r = 0; r < R; ++r) CK(leave_direct(handles[r]));      // (ranks with queues of their own: everything is enqueued on all of them by now)
    return group_sync(g);
}
#endif   // BPM_TEST_HOOKS

extern "C" int bpm_step(bpm_handle_t s, int64_t n_gens) {
    CK(check_handle_keep_direct(s));
    CK(set_device(s));
    // (a rank of a local group with a queue of its own and the push exchange is as independent as a process of its own: one host
    // thread per rank may drive it through bpm_step, the ranks then meet in the cross-rank barrier kernels like the ranks of a world)
    if (s->local_group && !(s->dq_private && s->push_connected && s->push_enabled))
        return fail("bpm_step: ranks of a local test group are driven by bpm_local_group_step (test variant)");
    if (!s->run_open) return fail("bpm_step: call bpm_begin_run first");
    if (n_gens < 0) return fail("bpm_step: n_gens < 0");
    if (s->cfg.keep_history) CK(ensure_history(s, s->hist_rows + n_gens));
    CK(ensure_gen_sums(s, s->rows_logical + n_gens));
    bpm_sampler* one[1] = {s};
    Group g{one, 1, s->comm != nullptr};
    return run_generations(g, n_gens);
}

// Exchange policy of a world_size > 1 sampler.  mode: 2 = accept bytes + replay (default), 1 = accepted rows in packed
// blocks, 0 = dense all-gather; cap > 0 sets the capacity of the next chunk of mode 1 (rows per sub-block per half
// generation; rounded up to even, clamped to [2, all chains of a sub-block]).  The same values on every rank of a world.
extern "C" int bpm_set_exchange(bpm_handle_t s, int32_t mode, int32_t cap) {
    CK(check_handle(s));
    if (mode < 0 || mode > 3) return fail("bpm_set_exchange: mode must be 0 (dense), 1 (rows), 2 (replay) or 3 (push)");
    if (mode == 3 && !s->push_connected) return fail("bpm_set_exchange: the push exchange needs bpm_push_connect first");
    if (mode == 3 && s->push_failed) return fail("bpm_set_exchange: a wait of the push exchange ran into its limit earlier (the ranks' barrier counters no longer agree): this sampler can only go on with an RCCL exchange");
    if (mode != 3 && s->push_no_rccl) return fail("bpm_set_exchange: this sampler was created without an RCCL communicator: the push exchange is its only one");
    if (!s->PK) return mode ? fail("bpm_set_exchange: this sampler only has the dense exchange (world_size 1, synchronous DE-MC or host callback)") : 0;
    if ((mode == 1 || mode == 2) && s->shape.idx == SHAPE_WIDE)
        return fail("bpm_set_exchange: rows wider than 512 coordinates exchange by push (3) or by the dense all-gather (0): the looped kernel has no replay / packed-rows form");
    if (s->push_enabled && mode != 3) s->cur = -1;      // (windows built with this rank's records only are rebuilt: ensure_perm_table)
    if (s->push_failed && mode != 3) {                  // leaving a push exchange that timed out: the next bpm_synchronize must not report it again
        CK(set_device(s));
        HIPCK(hipMemset(&s->ctrl->err, 0, sizeof(unsigned long long)));
    }
    s->push_enabled = mode == 3;
    if (mode == 3) s->push_agent_scope = (cap & 1) != 0;
    s->sparse_enabled = mode == 1;
    s->replay_enabled = mode == 2;
    if (mode == 1 && cap > 0) s->xcap = std::min<uint32_t>(s->xcap_max, ((uint32_t)cap + 1u) & ~1u);
    return 0;
}

// out[0] = mode (0 dense, 1 rows, 2 replay), out[1] = current capacity of mode 1, out[2] = chunks run with mode 1,
// out[3] = chunks of mode 1 replayed dense, out[4] = generations exchanged by replay
extern "C" int bpm_get_exchange_stats(bpm_handle_t s, int64_t* out) {
    CK(check_handle(s));
    if (!out) return fail("bpm_get_exchange_stats: null argument");
    out[0] = (s->push_enabled && s->push_connected) ? 3 : (s->replay_enabled ? 2 : (s->sparse_enabled ? 1 : 0));
    out[1] = (int64_t)s->xcap;
    out[2] = s->n_sparse_chunks;
    out[3] = s->n_sparse_replays;
    out[4] = s->n_replay_gens;
    out[5] = s->n_push_gens;
    // bits 0-1: connected (2: control block fine-grained); bit 2 / 3: the arena self-test passed with system- / agent-scope packet fences;
    // bit 4: it ran on the library's own queue
    out[6] = (s->push_connected ? (s->ctrl_fine ? 2 : 1) : 0) | (s->probe_sys_ok ? 4 : 0) | (s->probe_agent_ok ? 8 : 0) | (s->probe_direct ? 16 : 0);
    out[7] = (int64_t)s->push_seq | (s->push_agent_scope ? (1ll << 62) : 0);
    return 0;
}

// ---- push exchange: connection ---------------------------------------------------------------------------------------------
// What a rank publishes about its arena (bpm_push_export), BPM_PUSH_BLOB_BYTES in all.  The blobs of all ranks, in rank order, are what
// bpm_push_connect takes: the caller moves them with whatever it has (mpi4py allgather, torch.distributed, a file).
struct PushBlob {
    uint64_t magic;
    int64_t pid;
    int32_t rank, world, device, has_ipc;
    uint64_t arena_addr, arena_bytes, off_om, off_ctrl, n_chains, dim;
    hipIpcMemHandle_t handle;
    uint64_t ctrl_addr;            // != 0: the control block is an allocation of its own (fine-grained), exported by ctrl_handle
    hipIpcMemHandle_t ctrl_handle;
};
static_assert(sizeof(PushBlob) <= BPM_PUSH_BLOB_BYTES, "blob size");
static constexpr uint64_t PUSH_MAGIC = 0x4850555350504D42ull;      // "BPMPUSPH"

extern "C" int bpm_push_export(bpm_handle_t s, void* blob) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!blob) return fail("bpm_push_export: null argument");
    if (!s->arena) return fail("bpm_push_export: this sampler has no push exchange (world_size 1, more than 16 ranks, synchronous DE-MC or host callback)");
    PushBlob b;
    std::memset(&b, 0, sizeof(b));
    b.magic = PUSH_MAGIC; b.pid = (int64_t)getpid(); b.rank = (int32_t)s->rank; b.world = (int32_t)s->world; b.device = s->cfg.device;
    b.arena_addr = (uint64_t)(uintptr_t)s->arena; b.arena_bytes = s->arena_bytes; b.off_om = s->off_om; b.off_ctrl = s->off_ctrl;
    b.n_chains = s->N; b.dim = s->dim;
    if (hipIpcGetMemHandle(&b.handle, s->arena) == hipSuccess) b.has_ipc = 1;
    else { (void)hipGetLastError(); b.has_ipc = 0; }      // (ranks of ONE process need no handle; another process will be told)
    if (s->ctrl_fine) {
        b.ctrl_addr = (uint64_t)(uintptr_t)s->ctrl;
        if (hipIpcGetMemHandle(&b.ctrl_handle, s->ctrl) != hipSuccess) { (void)hipGetLastError(); b.has_ipc = 0; }
    }
    std::memset(blob, 0, BPM_PUSH_BLOB_BYTES);
    std::memcpy(blob, &b, sizeof(b));
    return 0;
}

extern "C" int bpm_push_connect(bpm_handle_t s, const void* blobs) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!blobs) return fail("bpm_push_connect: null argument");
    if (!s->arena) return fail("bpm_push_connect: this sampler has no push exchange");
    if (s->push_connected) return fail("bpm_push_connect: already connected");
    HIPCK(hipStreamSynchronize(s->stream));
    std::vector<unsigned long long> all(3 * MAX_SEG, 0ull), others(MAX_PEERS, 0ull);
    uint32_t n_others = 0;
    // (a failure half way leaves nothing mapped: a later attempt starts from scratch)
    struct Undo {
        bpm_sampler* s; bool armed = true;
        ~Undo() {
            if (!armed) return;
            for (uint32_t p = 0; p < (uint32_t)MAX_SEG; ++p) {
                if (s->peer_opened[p] && s->peer_base[p]) (void)hipIpcCloseMemHandle(s->peer_base[p]);
                if (s->peer_ctrl_opened[p] && s->peer_ctrl_base[p]) (void)hipIpcCloseMemHandle(s->peer_ctrl_base[p]);
                s->peer_opened[p] = s->peer_ctrl_opened[p] = false; s->peer_base[p] = s->peer_ctrl_base[p] = nullptr;
            }
        }
    } undo{s};
    for (uint32_t p = 0; p < s->world; ++p) {
        PushBlob b;
        std::memcpy(&b, static_cast<const char*>(blobs) + (size_t)p * BPM_PUSH_BLOB_BYTES, sizeof(b));
        if (b.magic != PUSH_MAGIC || b.rank != (int32_t)p || b.world != (int32_t)s->world)
            return fail("bpm_push_connect: blob " + std::to_string(p) + " is not the export of rank " + std::to_string(p) + " of this world");
        if (b.arena_bytes != s->arena_bytes || b.off_om != s->off_om || b.off_ctrl != s->off_ctrl || b.n_chains != s->N || b.dim != s->dim)
            return fail("bpm_push_connect: rank " + std::to_string(p) + " has another sampler shape");
        void* base = nullptr;
        if (p == s->rank) {
            if (b.arena_addr != (uint64_t)(uintptr_t)s->arena || b.pid != (int64_t)getpid()) return fail("bpm_push_connect: the blob of this rank is not its own export");
            base = s->arena;
        } else if (b.pid == (int64_t)getpid()) {
            base = reinterpret_cast<void*>((uintptr_t)b.arena_addr);          // a rank of the same process (local test group): the pointer itself
        } else {
            if (!b.has_ipc) return fail("bpm_push_connect: rank " + std::to_string(p) + " could not export an IPC handle of its buffer (hipIpcGetMemHandle failed there: on hosts "
                                        "whose driver supports dmabuf IPC only, HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment before anything initialises HSA)");
            const hipError_t e = hipIpcOpenMemHandle(&base, b.handle, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(std::string("bpm_push_connect: hipIpcOpenMemHandle of rank ") + std::to_string(p) + "'s buffer failed: " + hipGetErrorString(e) +
                            " (HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every rank? peer access between the devices?)");
            }
            s->peer_opened[p] = true;
            s->peer_base[p] = base;
            // what was mapped must be at least as large as the arena the blob describes: a handle of some other, smaller allocation would make
            // the first store beyond its end a GPU memory fault instead of an error here
            void* rb = nullptr; size_t rsz = 0;
            if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&rb), &rsz, base) == hipSuccess) {
                const size_t avail = rsz - (size_t)((char*)base - (char*)rb);
                if (avail < s->arena_bytes)
                    return fail("bpm_push_connect: the buffer mapped for rank " + std::to_string(p) + " holds " + std::to_string(avail) + " bytes, its arena needs " +
                                std::to_string(s->arena_bytes) + " (not the handle of that rank's exchange buffer)");
            } else (void)hipGetLastError();
        }
        s->peer_base[p] = base;
        const unsigned long long a = (unsigned long long)(uintptr_t)base;
        unsigned long long ctrl_a = a + s->off_ctrl;
        if (b.ctrl_addr != 0) {                              // the peer keeps its control block in a fine-grained allocation of its own
            void* cb = nullptr;
            if (p == s->rank) cb = s->ctrl;
            else if (b.pid == (int64_t)getpid()) cb = reinterpret_cast<void*>((uintptr_t)b.ctrl_addr);
            else {
                const hipError_t e = hipIpcOpenMemHandle(&cb, b.ctrl_handle, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) { (void)hipGetLastError(); return fail(std::string("bpm_push_connect: hipIpcOpenMemHandle of rank ") + std::to_string(p) + "'s control block failed: " + hipGetErrorString(e)); }
                s->peer_ctrl_opened[p] = true;
            }
            s->peer_ctrl_base[p] = cb;
            ctrl_a = (unsigned long long)(uintptr_t)cb;
        }
        all[p] = a; all[MAX_SEG + p] = ctrl_a; all[2 * MAX_SEG + p] = a + s->off_om;
        if (p != s->rank) others[n_others++] = a;
    }
    for (uint32_t i = n_others; i < (uint32_t)MAX_PEERS; ++i) others[i] = others[0];
    HIPCK(hipMemcpy(s->tab_all, all.data(), all.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(s->tab_peerG, others.data(), others.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    undo.armed = false;
    s->push_connected = true;
    s->push_enabled = true;               // connected: the default exchange from here on (bpm_set_exchange chooses another)
    return 0;
}

// One launch of push_arena_probe_kernel for rank s (kernels.h): on the library's own queue when the self-test runs there, else on the stream.
static int launch_arena_probe(bpm_sampler* s, int mode, int form, unsigned long long seed, double* save, int fence) {
    ProbeGeo geo{s->L.blk, s->n_local, s->ld, (s->om && s->off_om) ? 2u * s->n_local : 0u, s->world, s->rank, 2u * s->ld + 8u};
    unsigned long long* bad = &s->ctrl->arena_bad[form];
    if (g_dq) {
        struct { const unsigned long long* tab; ProbeGeo g; int mode; unsigned long long seed; double* save; unsigned long long* bad; } ka{s->tab_all, geo, mode, seed, save, bad};
        const bpm::DqKernel* k = g_dq->kernel(reinterpret_cast<const void*>(push_arena_probe_kernel));
        if (!k || g_dq->launch(*k, PROBE_REGIONS, s->world, WAVE, &ka, sizeof(ka), fence) != 0) return fail("direct AQL queue: push_arena_probe_kernel: " + g_dq->why());
        g_dq_need_acquire = true;
        return 0;
    }
    hipLaunchKernelGGL(push_arena_probe_kernel, dim3(PROBE_REGIONS, s->world), dim3(WAVE), 0, s->stream, (const unsigned long long*)s->tab_all, geo, mode, seed, save, bad);
    HIPCK(hipGetLastError());
    return 0;
}

// Collective over the ranks (all processes call it together; a local group passes its R handles), in two parts.
// (1) control blocks: every rank writes a pattern into every other rank's (fine-grained) control block, one cross-rank barrier, every rank checks.
// (2) arenas: every rank stores probe rows into ITS block of every peer's arena -- first and last row, the block's last slots, both ends of its
//     om block: the addresses its update kernels will push to -- FROM THE QUEUE AND UNDER THE PACKET FENCES THE UPDATE KERNELS USE (the library's
//     own AQL queue where the sampler runs on one), a cross-rank barrier, every rank verifies what arrived in its own arena and puts back what
//     was there.  Once with system-scope fences (decides *ok with part 1), once with the agent-scope form (reported by bpm_get_exchange_stats:
//     a caller skips "push with agent-scope fences" when its probe failed).  Round 3 probed the control block only, from the HIP stream: the
//     arena was first touched by real updates.
// *ok = 1: this rank (every rank of the group) received everything its peers sent.
extern "C" int bpm_push_selftest(bpm_handle_t* handles, int32_t R, int32_t* ok) {
    if (!handles || R < 1 || !ok) return fail("bpm_push_selftest: bad argument");
    *ok = 0;
    for (int r = 0; r < R; ++r) {
        bpm_sampler* s = handles[r];
        CK(check_handle(s));
        if (!s->push_connected) return fail("bpm_push_selftest: call bpm_push_connect first");
        if (R > 1 && (!s->local_group || (int)s->world != R || (int)s->rank != r)) return fail("bpm_push_selftest: handles are not ranks 0..R-1 of one local group");
    }
    CK(set_device(handles[0]));
    Group g{handles, R, false};
    const unsigned long long seed = 0xB1900000ull + handles[0]->push_seq;
    for (int r = 0; r < R; ++r) {
        bpm_sampler* s = handles[r];
        HIPCK(hipMemsetAsync(&s->ctrl->arena_bad[0], 0, 2 * sizeof(unsigned long long), s->stream));
        hipLaunchKernelGGL(push_probe_kernel, dim3(1), dim3(WAVE), 0, s->stream, (const unsigned long long*)(s->tab_all + MAX_SEG), s->world, s->rank, seed);
        HIPCK(hipGetLastError());
    }
    CK(push_barrier(g));
    CK(group_sync(g));
    bool ctrl_ok = true;
    for (int r = 0; r < R; ++r) {
        bpm_sampler* s = handles[r];
        PushCtrl c;
        HIPCK(hipMemcpy(&c, s->ctrl, sizeof(c), hipMemcpyDeviceToHost));
        if (c.err != 0) ctrl_ok = false;
        for (uint32_t p = 0; p < s->world; ++p)
            if (p != s->rank && c.probe[p] != seed + p) ctrl_ok = false;
    }
    // ---- part 2: the arenas, on the path the update kernels take
    std::vector<double*> save((size_t)R, nullptr);
    struct FreeSave { std::vector<double*>& v; ~FreeSave() { for (double* p : v) if (p) (void)hipFree(p); } } free_save{save};
    for (int r = 0; r < R; ++r) CK(dev_alloc(&save[(size_t)r], (size_t)handles[r]->world * (2 * handles[r]->ld + 8)));
    bool group_direct = false;
    const bool direct = group_goes_direct(g, true, group_direct);
    for (int r = 0; r < R; ++r) {
        bpm_sampler* s = handles[r];
        if (direct && !s->dq_active) { CK(wait_stream(s->stream)); s->dq_active = true; }
    }
    g_group_direct = group_direct;
    g_dq = direct ? handles[0]->dq : nullptr;
    g_dq_need_acquire = true;
    int rc = 0;
    for (int form = 0; form < 2 && rc == 0; ++form) {
        // form 0: acquire + release at system scope around the writer (what the HSA memory model asks for between agents); form 1: the single-GPU
        // form, acquire only at agent scope -- the probe rows themselves are system-scope write-through stores either way, as the pushed rows are
        const int f_read = form == 0 ? (bpm::DirectQueue::ACQUIRE | bpm::DirectQueue::SYSTEM) : bpm::DirectQueue::ACQUIRE;
        const int f_write = form == 0 ? (bpm::DirectQueue::FENCED | bpm::DirectQueue::SYSTEM) : bpm::DirectQueue::ACQUIRE;
        for (int mode = 0; mode < 3 && rc == 0; ++mode) {
            rc = push_barrier(g);
            for (int r = 0; r < R && rc == 0; ++r) {
                bind_rank_queue(handles[r]);
                rc = launch_arena_probe(handles[r], mode, form, seed + 16u * (unsigned)form, save[(size_t)r], mode == 1 ? f_write : f_read);
                if (g_dq) g_dq->flush();
            }
        }
        if (rc == 0) rc = push_barrier(g);      // nobody goes on (to the next form, to a generation) while a peer still verifies
    }
    g_group_direct = false;
    g_dq = nullptr;
    for (int r = 0; r < R; ++r) { const int rl = leave_direct(handles[r]); if (rc == 0) rc = rl; }
    CK(rc);
    CK(group_sync(g));
    bool sys_ok = true, agent_ok = true;
    for (int r = 0; r < R; ++r) {
        bpm_sampler* s = handles[r];
        PushCtrl c;
        HIPCK(hipMemcpy(&c, s->ctrl, sizeof(c), hipMemcpyDeviceToHost));
        if (c.err != 0) { sys_ok = false; agent_ok = false; }
        if (c.arena_bad[0] != 0) sys_ok = false;
        if (c.arena_bad[1] != 0) agent_ok = false;
    }
    for (int r = 0; r < R; ++r) { handles[r]->probe_sys_ok = sys_ok; handles[r]->probe_agent_ok = agent_ok; handles[r]->probe_direct = direct; }
    *ok = (ctrl_ok && sys_ok) ? 1 : 0;
    return 0;
}

extern "C" int bpm_get_launch_stats(bpm_handle_t s, int64_t* out) {
    CK(check_handle_keep_direct(s));
    if (!out) return fail("bpm_get_launch_stats: null argument");
    out[0] = s->dq != nullptr ? 1 : 0;
    out[1] = g_n_direct;
    out[2] = g_n_stream;
    out[3] = s->dq_active ? 1 : 0;
    out[4] = s->coherent ? 1 : 0;
    out[5] = s->dq_fence;
    return 0;
}

#ifdef BPM_TEST_HOOKS
// Test hook: the probe behind the choice of packet fences, on the memory type the state would use (coherent_alloc != 0) or on
// ordinary device memory.  *wrong = elements that missed an update (0 = coherent without a release fence), -1 = could not run.
extern "C" int bpm_debug_coherence_probe(int32_t device, int32_t coherent_alloc, int64_t* wrong) {
    if (!wrong) return fail("bpm_debug_coherence_probe: null argument");
    HIPCK(hipSetDevice(device));
    bpm::DirectQueue* dq = bpm::DirectQueue::for_device(device);
    *wrong = -1;
    if (!dq) return 0;
#ifndef BPM_EXPERIMENT_COHERENT
    if (coherent_alloc != 0) return fail("bpm_debug_coherence_probe: the hardware-coherent memory type is not compiled into this library (experiment build only)");
#endif
    *wrong = coherence_probe(dq, coherent_alloc != 0);
    return 0;
}
#endif   // BPM_TEST_HOOKS

extern "C" int bpm_set_launch_path(bpm_handle_t s, int32_t direct, int32_t fence) {
    CK(check_handle(s));
    if (fence != -1 && fence != 0 && fence != 1 && fence != 3) return fail("bpm_set_launch_path: fence must be -1 (keep), 0 (none), 1 (acquire) or 3 (acquire + release)");
    if (fence == 0 && !s->coherent) return fail("bpm_set_launch_path: packets without an acquire fence are not available in this library");
    if (fence == 1 && (!s->dq || !s->dq->has_fence_kernel())) return fail("bpm_set_launch_path: acquire-only packets need the library's queue and its fence kernel");
    s->dq_enabled = direct != 0;
    if (fence >= 0) s->dq_fence = fence;
    return 0;
}

extern "C" int bpm_get_step_time(bpm_handle_t s, float* elapsed_ms, int64_t* n_launches);
extern "C" int bpm_step_timed(bpm_handle_t s, int64_t n_gens, float* elapsed_ms, int64_t* n_launches) {
    CK(check_handle_keep_direct(s));
    CK(set_device(s));
    if (s->cfg.algo == BPM_ALGO_DEMC_SYNC) return fail("bpm_step_timed: not available for the synchronous DE-MC mode");
    if (s->cfg.keep_history) CK(ensure_history(s, s->hist_rows + n_gens));
    CK(ensure_gen_sums(s, s->rows_logical + n_gens));
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (n_launches) *n_launches = 0;
    s->timed_want_first = n_gens > 0;
    // An event-bound dispatch costs the host 10-30 us instead of 2.5.  At the head of a call that starts from an idle GPU this is time
    // the GPU waits; a quarter into the call the host is far enough ahead to absorb it.  So ev0 rides on launch number n/4 (the
    // interval then covers the last three quarters of the call's launches), ev1 on the last one.
    s->timed_skip = n_gens >= 8 ? (2 * n_gens) / 4 : 0;
    s->timed_last_gen = n_gens > 0 ? s->t_abs + n_gens - 1 : -1;
    s->timed_l0 = s->timed_l1 = -1;
    g_launch_log.clear();
    const long long tt0 = g_host_timing ? now_ns() : 0;
    const int rc = bpm_step(s, n_gens);
    s->timed_want_first = false;
    s->timed_last_gen = -1;
    g_stop_event = nullptr;
    g_dq_sig = -1;
    CK(rc);
    s->timed_direct = s->dq_active;
    if (s->dq_active) { if (s->dq->drain() != 0) return fail("direct AQL queue: " + s->dq->why()); }     // (the sampler stays in direct mode: the next call goes on)
    else CK(wait_stream(s->stream));
    if (g_host_timing && !g_launch_log.empty() && g_launch_log.size() <= 400) {
        const long long tt1 = now_ns();
        fprintf(stderr, "[bpm host timing] bpm_step_timed(%lld): entry -> drained %.1f us; launch calls (start offset us : duration us):", (long long)n_gens, (tt1 - tt0) * 1e-3);
        for (size_t i = 0; i + 1 < g_launch_log.size(); i += 2)
            fprintf(stderr, " %.1f:%.1f", (g_launch_log[i] - tt0) * 1e-3, (g_launch_log[i + 1] - g_launch_log[i]) * 1e-3);
        fprintf(stderr, "\n");
    }
    // (reading the two events costs tens of microseconds of host time: left to bpm_get_step_time, outside the caller's own clock)
    if (elapsed_ms || n_launches) return bpm_get_step_time(s, elapsed_ms, n_launches);
    return 0;
}

extern "C" int bpm_get_step_time(bpm_handle_t s, float* elapsed_ms, int64_t* n_launches) {
    CK(check_handle_keep_direct(s));
    CK(set_device(s));
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (n_launches) *n_launches = 0;
    if (s->timed_direct && s->timed_l0 >= 0 && s->timed_l1 > s->timed_l0) {
        double t0 = 0.0, t1 = 0.0;                    // end-of-kernel time stamps of the two dispatches, written by the packet processor
        if (!s->dq || !s->dq->dispatch_end_ns(&t0, &t1)) return fail("bpm_get_step_time: no dispatch time stamps");
        if (elapsed_ms) *elapsed_ms = (float)((t1 - t0) * 1e-6);
        if (n_launches) *n_launches = s->timed_l1 - s->timed_l0;
    } else if (s->timed_l0 >= 0 && s->timed_l1 > s->timed_l0) {
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        if (elapsed_ms) *elapsed_ms = ms;
        if (n_launches) *n_launches = s->timed_l1 - s->timed_l0;
    }
    return 0;
}


#ifdef BPM_TEST_HOOKS
// Same generations as bpm_step, with a HIP event pair on the sampler's stream around every
// update-kernel launch: returns the summed kernel time and the number of launches -- a cross-check of
// the per-launch duration bench.py prices the roofline with (rocprofv3 --kernel-trace gives the same
// figure offline).  Test variant only (include/bipymc_hip_test.h): bench.py runs it outside its timed region.
extern "C" int bpm_step_profiled(bpm_handle_t s, int64_t n_gens, double* kernel_ms_sum, int64_t* n_launches) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->run_open) return fail("bpm_step_profiled: call bpm_begin_run first");
    if (s->cfg.target_id == BPM_TARGET_HOST_CALLBACK) return fail("bpm_step_profiled: device targets only");
    if (s->cfg.algo == BPM_ALGO_DEMC_SYNC) return fail("bpm_step_profiled: not available for the synchronous DE-MC mode");
    if (n_gens <= 0 || n_gens > 4096) return fail("bpm_step_profiled: 1 <= n_gens <= 4096");
    // (this entry point exchanges with the dense RCCL all-gather; a world without a communicator has no exchange here)
    if (s->world > 1 && !s->comm) return fail("bpm_step_profiled: not available for a rank of a world without an RCCL communicator (push exchange only): use bpm_step_timed");
    if (s->cfg.keep_history) CK(ensure_history(s, s->hist_rows + n_gens));
    CK(ensure_gen_sums(s, s->rows_logical + n_gens));
    std::vector<hipEvent_t> ev((size_t)n_gens * 4);
    for (auto& e : ev) HIPCK(hipEventCreate(&e));
    PhaseLaunch fn = pick_fused(s);
    s->sparse_active = false;
    s->replay_active = false;
    for (int64_t g = 0; g < n_gens; ++g) {
        CK(prepare_generation(s, n_gens - g));
        for (int ph = 0; ph < 2; ++ph) {
            HIPCK(hipEventRecord(ev[(size_t)g * 4 + 2 * ph], s->stream));
            if (s->cur_args[ph].n_items > 0) fn(s->cur_args[ph], s->stream);
            HIPCK(hipEventRecord(ev[(size_t)g * 4 + 2 * ph + 1], s->stream));
            CK(allgather_state(s));
        }
        HIPCK(hipGetLastError());
        CK(finish_generation(s));
        if (s->outlier_due) { bpm_sampler* one[1] = {s}; CK(group_outlier_check(Group{one, 1, s->comm != nullptr})); }
    }
    HIPCK(hipStreamSynchronize(s->stream));
    double tot = 0.0;
    for (int64_t i = 0; i < 2 * n_gens; ++i) {
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, ev[(size_t)2 * i], ev[(size_t)2 * i + 1]));
        tot += ms;
    }
    for (auto& e : ev) HIPCK(hipEventDestroy(e));
    if (kernel_ms_sum) *kernel_ms_sum = tot;
    if (n_launches) *n_launches = 2 * n_gens;
    return 0;
}
#endif   // BPM_TEST_HOOKS

// Warm start (demc.py:46-51,217-233): install `rows` history rows of this rank's chains
// (hist_local: rows x n_local x dim) and the full current state X (N x dim, = last row of every chain).
extern "C" int bpm_set_history(bpm_handle_t s, int64_t rows, const double* hist_local, const double* X) {
    CK(check_handle(s));
    CK(set_device(s));
    if (rows < 1 || !hist_local || !X) return fail("bpm_set_history: bad argument");
    if (!s->cfg.keep_history && rows > 1) return fail("bpm_set_history: sampler keeps no history");
    CK(bpm_set_state(s, X));
    if (rows > 1) {
        CK(ensure_history(s, rows));
        HIPCK(hipMemsetAsync(s->hist, 0, (size_t)rows * s->n_local * s->ld * sizeof(double), s->stream));
        HIPCK(hipMemcpy2DAsync(s->hist, s->ld * sizeof(double), hist_local, s->dim * sizeof(double), s->dim * sizeof(double),
                               (size_t)rows * s->n_local, hipMemcpyHostToDevice, s->stream));
        // ln_like of the old rows is not stored in the reference's checkpoint: recompute (device targets) or NaN
        if (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK) {
            const uint32_t n = (uint32_t)((size_t)rows * s->n_local);
            switch (s->cfg.target_id) {
                case BPM_TARGET_GAUSS_EQUICORR: g_eval_gauss[s->shape.idx](s->hist, n, s->ld, s->dim, s->tparams, s->llhist, s->stream); break;
                case BPM_TARGET_MIXTURE_PAIRS: g_eval_mixture[s->shape.idx](s->hist, n, s->ld, s->dim, s->tparams, s->llhist, s->stream); break;
                default: launch_eval<TARGET_BANANA, 1, 2>(s->hist, n, s->ld, s->dim, s->tparams, s->llhist, s->stream); break;
            }
            HIPCK(hipGetLastError());
        } else {
            // host-callback target: the caller supplies the current ln-likes next (bpm_set_loglike fills the last row);
            // the older rows are unknown -- NaN, which the outlier check skips
            std::vector<double> nanv((size_t)rows * s->n_local, std::nan(""));
            HIPCK(hipMemcpyAsync(s->llhist, nanv.data(), nanv.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
            HIPCK(hipStreamSynchronize(s->stream));
        }
        s->hist_rows = rows;
        s->rows_logical = rows;
        for (int64_t r = 0; r < rows && r < (int64_t)s->hist_tag.size(); ++r) s->hist_tag[(size_t)r] = -1;
        s->w_rows = 0;      // moments are rebuilt from the rows when adaptation next needs them
        if (s->cfg.running_moments) {      // population sums of every installed row
            CK(ensure_gen_sums(s, rows));
            for (int64_t g = 0; g < rows; ++g) CK(push_gen_sums(s, g, s->hist + (uint64_t)g * s->n_local * s->ld));
        }
        HIPCK(hipStreamSynchronize(s->stream));
    }
    return 0;
}

// param_est (demc.py:235-248) without moving the history: raw moments of this rank's part of the
// super chain (row g*N + i = chain i at generation g) for rows >= n_burn:
// count, sum_j (x - shift_j), sum_j (x - shift_j)^2, shift_j (= chain 0's current state, identical
// on every rank).  Ranks combine the three and finish mean / std(ddof=0) on the host.
extern "C" int bpm_reduce_moments(bpm_handle_t s, int64_t n_burn, double* sum, double* sumsq, double* shift,
                                  int64_t* count) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!sum || !sumsq || !shift || !count) return fail("bpm_reduce_moments: null argument");
    if (n_burn < 0) n_burn = 0;
    if (s->cfg.running_moments && (!s->cfg.keep_history || s->hist_rows != s->rows_logical)) {
        // no resident history: answer from the per-generation population sums -- exact for a burn-in of whole generations
        if (n_burn % s->N != 0)
            return fail("bpm_reduce_moments: without a resident history n_burn must be a multiple of n_chains (whole generations)");
        const int64_t g0 = std::min<int64_t>(n_burn / s->N, s->rows_logical), g1 = s->rows_logical;
        std::vector<double> h(3 * (size_t)s->ld, 0.0);
        HIPCK(hipMemcpyAsync(h.data() + 2 * s->ld, s->gs_shift, s->ld * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        *count = (g1 - g0) * (int64_t)s->n_local;
        if (g1 > g0) {
            double* out = nullptr;
            CK(dev_alloc(&out, 2 * (size_t)s->ld));
            hipLaunchKernelGGL(moments_final_kernel, dim3(s->ld), dim3(MOM_THREADS), 0, s->stream, (const double*)(s->gen_sums + (size_t)g0 * 2 * s->ld),
                               (uint32_t)(g1 - g0), s->ld, out);
            HIPCK(hipGetLastError());
            HIPCK(hipMemcpyAsync(h.data(), out, 2 * s->ld * sizeof(double), hipMemcpyDeviceToHost, s->stream));
            HIPCK(hipStreamSynchronize(s->stream));
            HIPCK(hipFree(out));
        } else {
            HIPCK(hipStreamSynchronize(s->stream));
        }
        for (uint32_t j = 0; j < s->dim; ++j) { sum[j] = h[j]; sumsq[j] = h[s->ld + j]; shift[j] = h[2 * s->ld + j]; }
        return 0;
    }
    const int64_t g0 = n_burn / s->N;
    int64_t first = n_burn % s->N - (int64_t)s->lo;      // first local chain of generation g0 that counts
    first = std::max<int64_t>(0, std::min<int64_t>(first, s->n_local));
    if (first != 0 && g0 < s->hist_rows) CK(normalize_history(s, g0, g0 + 1));      // a partial first generation is counted by chain index
    const uint64_t m_lo = (uint64_t)std::min<int64_t>(g0, s->hist_rows) * s->n_local + (g0 < s->hist_rows ? (uint64_t)first : 0);
    const uint64_t m_hi = (uint64_t)s->hist_rows * s->n_local;
    std::vector<double> h(3 * (size_t)s->ld, 0.0);
    HIPCK(hipMemcpyAsync(h.data() + 2 * s->ld, s->G, s->ld * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    *count = (int64_t)(m_hi > m_lo ? m_hi - m_lo : 0);
    if (m_hi > m_lo) {
        // enough blocks for ~8 per CU (each thread keeps MOM_UNR 16-byte loads in flight), at least 64 rows each
        const uint32_t nb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(2048, (m_hi - m_lo + 63) / 64));
        double* part = nullptr;
        double* out = nullptr;
        CK(dev_alloc(&part, (size_t)nb * 2 * s->ld));
        CK(dev_alloc(&out, 2 * (size_t)s->ld));
        hipLaunchKernelGGL(moments_partial_kernel, dim3(nb), dim3(MOM_THREADS), 0, s->stream, s->hist, m_lo, m_hi, s->ld,
                           s->G, part);
        hipLaunchKernelGGL(moments_final_kernel, dim3(s->ld), dim3(MOM_THREADS), 0, s->stream, part, nb, s->ld, out);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(h.data(), out, 2 * s->ld * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIPCK(hipStreamSynchronize(s->stream));
        HIPCK(hipFree(part));
        HIPCK(hipFree(out));
    } else {
        HIPCK(hipStreamSynchronize(s->stream));
    }
    for (uint32_t j = 0; j < s->dim; ++j) { sum[j] = h[j]; sumsq[j] = h[s->ld + j]; shift[j] = h[2 * s->ld + j]; }
    return 0;
}

extern "C" int bpm_synchronize(bpm_handle_t s) {
    CK(check_handle(s));
    CK(set_device(s));
    CK(wait_stream(s->stream));
    return push_check_error(s);
}

// ---- host-callback ln_like_fn (samplers.py:36-43): one half generation = proposals out, ln-likes in ------------------------------------------------
// Round 5: ONE core in three transports.  The proposal kernel writes the half generation's proposals (work-item order) into prop_buf and the
// snooker corrections into aux_buf[0 .. n_local); the ln-likes arrive in aux_buf[n_local ..) -- from pinned host memory chunk by chunk while the
// caller evaluates (bpm_propose_begin / _chunk, bpm_commit_chunk / _end), through the caller-owned buffers of rounds 1-4 (bpm_propose / bpm_commit:
// the same calls with one chunk and a compaction copy), or from DEVICE memory the caller's own framework computed them in (bpm_propose_device /
// bpm_commit_device: no PCIe).  Then the commit kernel: Metropolis (samplers.py:328-336), append (chain.py:51-54), CR statistics.
static int propose_launch(bpm_sampler* s, const char* who, bool clear_ids = true) {
    if (!s->run_open) return fail(std::string(who) + ": call bpm_begin_run first");
    if (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK) return fail(std::string(who) + ": sampler has a device target; use bpm_step");
    if (s->proposed) return fail(std::string(who) + ": previous proposals not committed");
    if (s->phase == 0) CK(prepare_generation(s, 1));
    const PhaseArgs& a = s->cur_args[s->phase];
    // (only entries the proposal kernel does not write -- work items beyond n_items, which nobody reads: kept for the host transports as rounds 1-4 had it)
    if (clear_ids) HIPCK(hipMemsetAsync(s->ids_buf, 0xFF, s->n_local * sizeof(int32_t), s->stream));
    if (a.n_items > 0) g_propose[s->cfg.algo == BPM_ALGO_DREAM ? 1 : 0][s->shape.idx](a, s->stream);
    HIPCK(hipGetLastError());
    return 0;
}
// the commit kernel + the bookkeeping of the half generation (the ln-likes of every active work item are in aux_buf[n_local ..) on the stream by now)
static int commit_finish(bpm_sampler* s, int64_t n_active, bool sync = true) {
    const PhaseArgs& a = s->cur_args[s->phase];
    if (a.n_items > 0) g_commit[s->cfg.algo == BPM_ALGO_DREAM ? 1 : 0][s->shape.idx](a, s->stream);
    HIPCK(hipGetLastError());
    if (s->cfg.algo == BPM_ALGO_DEMC_SYNC) {       // one propose/commit per generation: apply the banked updates
        HIPCK(hipMemcpyAsync(s->G + (uint64_t)s->rank * s->L.blk, s->x_next, (size_t)s->n_local * s->ld * sizeof(double),
                             hipMemcpyDeviceToDevice, s->stream));
        CK(allgather_state(s));
        HIPCK(hipStreamSynchronize(s->stream));
        s->proposed = false; s->prop_mode = 0;
        return finish_generation(s);       // (synchronous DE-MC: never DREAM, no outlier check)
    }
    CK(allgather_state(s));
    if (sync) HIPCK(hipStreamSynchronize(s->stream));
    s->proposed = false; s->prop_mode = 0;
    if (s->phase == 0) {
        s->phase = 1;
        s->phase_a_updates = n_active;
    } else {
        s->phase = 0;
        CK(finish_generation(s));
        if (s->outlier_due) { bpm_sampler* one[1] = {s}; CK(group_outlier_check(Group{one, 1, s->comm != nullptr})); }
    }
    return 0;
}
static inline uint32_t chunk_lo(uint32_t n_items, int32_t n_chunks, int32_t k) { return (uint32_t)(((uint64_t)n_items * (uint64_t)k) / (uint64_t)n_chunks); }

extern "C" int bpm_propose_begin(bpm_handle_t s, int32_t n_chunks) {
    CK(check_handle(s));
    CK(set_device(s));
    if (n_chunks < 1 || n_chunks > 256) return fail("bpm_propose_begin: 1 <= n_chunks <= 256");
    CK(propose_launch(s, "bpm_propose_begin"));
    const PhaseArgs& a = s->cur_args[s->phase];
    while ((int32_t)s->chunk_ev.size() < n_chunks) {
        hipEvent_t e = nullptr;
        HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->chunk_ev.push_back(e);
    }
    // the ids first (32 KB), then the proposals chunk by chunk, an event behind each: the DMA of chunk k + 1 runs while the caller evaluates chunk k
    HIPCK(hipMemcpyAsync(s->h_ids, s->ids_buf, s->n_local * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    for (int32_t k = 0; k < n_chunks; ++k) {
        const uint32_t w0 = chunk_lo(a.n_items, n_chunks, k), w1 = chunk_lo(a.n_items, n_chunks, k + 1);
        if (w1 > w0)
            HIPCK(hipMemcpyAsync(s->h_props + (size_t)w0 * s->ld, s->prop_buf + (size_t)w0 * s->ld, (size_t)(w1 - w0) * s->ld * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIPCK(hipEventRecord(s->chunk_ev[(size_t)k], s->stream));
    }
    s->prop_chunks = n_chunks; s->prop_mode = 1; s->prop_given = 0; s->prop_active = 0; s->prop_whole = false;
    s->prop_done.assign((size_t)n_chunks, 0);
    s->proposed = true;
    return 0;
}

extern "C" int bpm_propose_chunk(bpm_handle_t s, int32_t k, const double** rows, const int32_t** ids, int32_t* n_rows, int32_t* ld) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->proposed || s->prop_mode != 1) return fail("bpm_propose_chunk: call bpm_propose_begin first");
    if (k < 0 || k >= s->prop_chunks || !rows || !ids || !n_rows || !ld) return fail("bpm_propose_chunk: bad argument");
    const PhaseArgs& a = s->cur_args[s->phase];
    HIPCK(hipEventSynchronize(s->chunk_ev[(size_t)k]));
    const uint32_t w0 = chunk_lo(a.n_items, s->prop_chunks, k), w1 = chunk_lo(a.n_items, s->prop_chunks, k + 1);
    *rows = s->h_props + (size_t)w0 * s->ld;
    *ids = s->h_ids + w0;
    *n_rows = (int32_t)(w1 - w0);
    *ld = (int32_t)s->ld;
    return 0;
}

extern "C" int bpm_commit_chunk(bpm_handle_t s, int32_t k, const double* ll) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->proposed || s->prop_mode != 1) return fail("bpm_commit_chunk: call bpm_propose_begin first");
    if (k < 0 || k >= s->prop_chunks) return fail("bpm_commit_chunk: bad chunk");
    if (s->prop_done[(size_t)k]) return fail("bpm_commit_chunk: the values of piece " + std::to_string(k) + " were handed in already");
    const PhaseArgs& a = s->cur_args[s->phase];
    const uint32_t w0 = chunk_lo(a.n_items, s->prop_chunks, k), w1 = chunk_lo(a.n_items, s->prop_chunks, k + 1);
    if (w1 > w0) {
        if (!ll) return fail("bpm_commit_chunk: null ll");
        // (into pinned staging first: the caller's array may be gone before the copy runs)
        std::memcpy(s->h_ll + w0, ll, (size_t)(w1 - w0) * sizeof(double));
        for (uint32_t w = w0; w < w1; ++w) s->prop_active += s->h_ids[w] >= 0 ? 1 : 0;
        HIPCK(hipMemcpyAsync(s->aux_buf + s->n_local + w0, s->h_ll + w0, (size_t)(w1 - w0) * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
    s->prop_done[(size_t)k] = 1;
    s->prop_given += 1;
    return 0;
}

extern "C" int bpm_commit_end(bpm_handle_t s) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->proposed || s->prop_mode != 1) return fail("bpm_commit_end: nothing proposed with bpm_propose_begin");
    s->prop_whole = false;
    if (s->prop_given != s->prop_chunks) return fail("bpm_commit_end: " + std::to_string(s->prop_given) + " of " + std::to_string(s->prop_chunks) + " chunks were given their ln-likes (bpm_commit_chunk)");
    return commit_finish(s, s->prop_active);
}

// ---- the caller's likelihood as a kernel compiled from HIP source (user_likelihood.h): proposal kernel -> the caller's kernel -> commit kernel, all on the
// sampler's stream, no host code inside a generation.  samplers.py:36-43 evaluates ln_like_fn(theta, **ln_kwargs) row by row on the host; here `params` takes
// the place of ln_kwargs.
static bpm::Hiprtc g_hiprtc;
static std::mutex g_hiprtc_mu;      // (the loader's state and hiprtc's own: one compilation at a time per process)
static int user_eval_launch(bpm_sampler* s, const double* rows, const int32_t* ids, uint32_t n, double* out) {
    if (n == 0) return 0;
    int n_i = (int)n, ld_i = (int)s->ld, d_i = (int)s->dim, rpb = 0, ldp = 0;
    bpm::user_eval_tile(s->dim, rpb, ldp);
    const double* params = s->user_params;
    void* args[] = {(void*)&rows, (void*)&ids, (void*)&n_i, (void*)&ld_i, (void*)&d_i, (void*)&params, (void*)&out, (void*)&rpb, (void*)&ldp};
    const unsigned block = (unsigned)bpm::USER_EVAL_BLOCK, per = rpb > 0 ? (unsigned)rpb : block, grid = (n + per - 1u) / per;
    const unsigned lds = rpb > 0 ? (unsigned)rpb * (unsigned)ldp * (unsigned)sizeof(double) : 0u;
    HIPCK(hipModuleLaunchKernel(s->user_fn, grid, 1, 1, block, 1, 1, lds, s->stream, args, nullptr));
    return 0;
}
// ln-like of the local chains' CURRENT states from the installed device likelihood (what bpm_set_loglike takes from the host)
static int user_refresh_ll(bpm_sampler* s) {
    if (s->user_fused_eval) {      // (the update kernel was compiled around the likelihood: the same Target::eval, the same bits)
        const double* X = s->G + (uint64_t)s->rank * s->L.blk;
        uint32_t n = s->n_local, ld = s->ld, dim = s->dim;
        const double* tp = s->user_params;
        double* out = s->ll;
        void* args[] = {(void*)&X, (void*)&n, (void*)&ld, (void*)&dim, (void*)&tp, (void*)&out};
        HIPCK(hipModuleLaunchKernel(s->user_fused_eval, grid_for(n, s->shape.lpc), 1, 1, s->user_fused_block, 1, 1, 0, s->stream, args, nullptr));
    } else {
        CK(user_eval_launch(s, s->G + (uint64_t)s->rank * s->L.blk, nullptr, s->n_local, s->ll));
    }
    if (s->hist_rows >= 1 && s->hist_rows == s->rows_logical)
        HIPCK(hipMemcpyAsync(s->llhist + (size_t)(s->hist_rows - 1) * s->n_local, s->ll, s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}
static int run_generations_user(bpm_sampler* s, int64_t n_gens) {
    // (a rank of a world exchanges through its RCCL communicator like the host transports do: allgather_state in commit_finish)
    if (s->world > 1 && !s->comm) return fail("bpm_step: a host-callback sampler of a world needs an RCCL communicator (create it with a unique id)");
    const int halves = s->cfg.algo == BPM_ALGO_DEMC_SYNC ? 1 : 2;
    for (int64_t g = 0; g < n_gens; ++g) {
        for (int h = 0; h < halves; ++h) {
            CK(propose_launch(s, "bpm_step", false));      // (the proposal kernel writes the id of EVERY work item below n_items, -1 for an idle one)
            const PhaseArgs& a = s->cur_args[s->phase];
            int64_t act = (int64_t)a.n_items;
            if (s->world > 1 && a.n_items > 0) {      // (a rank of a world: half of its work items sit in the other pool)
                HIPCK(hipMemcpyAsync(s->h_ids, s->ids_buf, s->n_local * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
                HIPCK(hipStreamSynchronize(s->stream));
                act = 0;
                for (uint32_t w = 0; w < a.n_items; ++w) act += s->h_ids[w] >= 0 ? 1 : 0;
            }
            CK(user_eval_launch(s, s->prop_buf, s->world > 1 ? s->ids_buf : nullptr, a.n_items, s->aux_buf + s->n_local));
            s->proposed = true; s->prop_mode = 3; s->prop_chunks = 0;
            CK(commit_finish(s, act, false));
        }
    }
    return 0;
}

extern "C" int bpm_check_device_likelihood(const char* hip_source, const char* arch, char* log, int64_t log_cap) {
    if (!hip_source) return fail("bpm_check_device_likelihood: null source");
    std::vector<char> code;
    std::string why;
    { std::lock_guard<std::mutex> lk(g_hiprtc_mu); why = bpm::compile_user_likelihood(g_hiprtc, hip_source, (arch && *arch) ? arch : "gfx950", code); }
    if (log && log_cap > 0) {
        const size_t n = std::min(why.size(), (size_t)log_cap - 1);
        std::memcpy(log, why.data(), n);
        log[n] = '\0';
    }
    return why.empty() ? 0 : fail("bpm_check_device_likelihood: " + why);
}

extern "C" int bpm_set_device_likelihood(bpm_handle_t s, const char* hip_source, const double* params, int32_t n_params) {
    CK(check_handle(s));
    CK(set_device(s));
    if (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK) return fail("bpm_set_device_likelihood: the sampler has a shipped device target (create it with BPM_TARGET_HOST_CALLBACK)");
    if (s->proposed) return fail("bpm_set_device_likelihood: a half generation is open (bpm_propose ... without its commit)");
    if (!hip_source || n_params < 0 || (n_params > 0 && !params)) return fail("bpm_set_device_likelihood: bad argument");
    hipDeviceProp_t prop;
    HIPCK(hipGetDeviceProperties(&prop, s->cfg.device));
    std::vector<char> code;
    std::string why;
    { std::lock_guard<std::mutex> lk(g_hiprtc_mu); why = bpm::compile_user_likelihood(g_hiprtc, hip_source, prop.gcnArchName, code); }
    if (!why.empty()) return fail("bpm_set_device_likelihood: " + why);
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    HIPCK(hipModuleLoadData(&mod, code.data()));
    if (hipModuleGetFunction(&fn, mod, "bpm_user_eval") != hipSuccess || !fn) {
        (void)hipGetLastError();
        (void)hipModuleUnload(mod);
        return fail("bpm_set_device_likelihood: the compiled module has no bpm_user_eval kernel");
    }
    HIPCK(hipStreamSynchronize(s->stream));                      // (a previous likelihood's launches are done)
    if (s->user_mod) (void)hipModuleUnload(s->user_mod);
    if (s->user_params) { (void)hipFree(s->user_params); s->user_params = nullptr; }
    s->user_mod = mod; s->user_fn = fn;
    CK(dev_alloc(&s->user_params, (size_t)std::max(n_params, 1)));
    HIPCK(hipMemsetAsync(s->user_params, 0, (size_t)std::max(n_params, 1) * sizeof(double), s->stream));
    if (n_params > 0) HIPCK(hipMemcpyAsync(s->user_params, params, (size_t)n_params * sizeof(double), hipMemcpyHostToDevice, s->stream));
    // the faster form: the update kernel itself compiled around the likelihood.  Whatever goes wrong here leaves the three-kernel form in place
    // (bpm_get_device_likelihood_info says which is in use and why).  BPM_USER_FUSED=0: not attempted (A/B, tests).
    if (s->user_fused_mod) {
        if (s->dq) for (const std::string& nm : s->user_fused_names) s->dq->forget_named(nm);
        (void)hipModuleUnload(s->user_fused_mod); s->user_fused_mod = nullptr;
    }
    s->user_fused_fn = nullptr; s->user_fused_hot = nullptr; s->user_fused_eval = nullptr; s->user_fused_adapt = nullptr; s->user_fused_dq = false; s->user_fused_why.clear();
    CK(user_refresh_ll(s));      // (by the kernel of its own; again below by the update kernel's arithmetic once that one is built)
    const bool want_fused = !(getenv("BPM_USER_FUSED") && atoi(getenv("BPM_USER_FUSED")) == 0);      // (read at every call: a test switches it)
    if (!want_fused) { s->user_fused_why = "BPM_USER_FUSED=0"; return 0; }
    if (s->shape.idx == SHAPE_WIDE) { s->user_fused_why = "rows wider than 512 coordinates run on the looped kernel, which has no run-time form"; return 0; }
    {
        const int algo = s->cfg.algo == BPM_ALGO_DREAM ? ALGO_DREAM : ALGO_DEMC;
        const int np = s->cfg.algo != BPM_ALGO_DREAM ? 1 : (s->cfg.del_pairs == 3 ? 3 : 0);
#ifdef BPM_TEST_HOOKS
        const bool hooks = true;
#else
        const bool hooks = false;
#endif
        std::vector<char> fcode;
        std::string lowered[4], fwhy, ns;
        { std::lock_guard<std::mutex> lk(g_hiprtc_mu);
          static std::atomic<int> module_no{0};
          ns = "v_user" + std::to_string(module_no.fetch_add(1));
          fwhy = bpm::compile_user_fused(g_hiprtc, hip_source, ns, prop.gcnArchName, bpm_src_kernels_h, bpm_src_philox_h, algo, s->shape.lpc, s->shape.dpl, np, s->dim, hooks,
                                         s->plan_on ? 1 : 2, fcode, lowered); }
        if (!fwhy.empty()) { s->user_fused_why = fwhy; return 0; }
        hipModule_t fm = nullptr;
        hipFunction_t ff = nullptr, fh = nullptr, fs = nullptr, fe = nullptr, fa = nullptr;
        if (hipModuleLoadData(&fm, fcode.data()) != hipSuccess) { (void)hipGetLastError(); s->user_fused_why = "hipModuleLoadData failed"; return 0; }
        if (hipModuleGetFunction(&ff, fm, lowered[0].c_str()) != hipSuccess || hipModuleGetFunction(&fh, fm, lowered[1].c_str()) != hipSuccess ||
            hipModuleGetFunction(&fe, fm, lowered[2].c_str()) != hipSuccess || hipModuleGetFunction(&fs, fm, "bpm_user_sizeof") != hipSuccess || !ff || !fh || !fe || !fs) {
            (void)hipGetLastError(); (void)hipModuleUnload(fm);
            s->user_fused_why = "the compiled module lacks " + lowered[0];
            return 0;
        }
        // the module's view of the argument block must be this library's
        unsigned int* d_out = nullptr;
        unsigned int h_out[3] = {0u, 0u, 0u};
        if (!lowered[3].empty() && (hipModuleGetFunction(&fa, fm, lowered[3].c_str()) != hipSuccess || !fa)) { (void)hipGetLastError(); fa = nullptr; }
        if (dev_alloc(&d_out, 3) != 0) { (void)hipModuleUnload(fm); s->user_fused_why = "out of device memory"; return 0; }
        void* sargs[] = {(void*)&d_out};
        const bool ran = hipModuleLaunchKernel(fs, 1, 1, 1, 1, 1, 1, 0, s->stream, sargs, nullptr) == hipSuccess &&
                         hipMemcpyAsync(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost, s->stream) == hipSuccess && hipStreamSynchronize(s->stream) == hipSuccess;
        (void)hipFree(d_out);
        if (!ran || h_out[0] != (unsigned)sizeof(PhaseArgs) || h_out[1] != (unsigned)block_for(s->shape.lpc)) {
            (void)hipGetLastError(); (void)hipModuleUnload(fm);
            s->user_fused_why = "the module's argument block differs from the library's (" + std::to_string(h_out[0]) + " against " + std::to_string(sizeof(PhaseArgs)) + " bytes)";
            return 0;
        }
        s->user_fused_mod = fm; s->user_fused_fn = ff; s->user_fused_hot = fh; s->user_fused_block = h_out[1];
        s->user_fused_names[0] = lowered[0]; s->user_fused_names[1] = lowered[1];
        s->user_fused_eval = fe;
        s->user_fused_adapt = fa; s->user_fused_block_adapt = h_out[2]; s->user_fused_names[2] = lowered[3];
        if (fa && h_out[2] != (unsigned)block_for_hot(s->shape.lpc, 3, s->shape.dpl)) s->user_fused_adapt = nullptr;
        static const bool dq_wanted = !(getenv("BPM_USER_FUSED") && atoi(getenv("BPM_USER_FUSED")) == 2);      // (2: fused, but launched on the stream -- A/B)
        s->user_fused_dq = dq_wanted && s->dq && s->dq->kernel_by_name(lowered[0]) != nullptr && s->dq->kernel_by_name(lowered[1]) != nullptr &&
                           (!s->user_fused_adapt || s->dq->kernel_by_name(lowered[3]) != nullptr);
        CK(user_refresh_ll(s));      // (again, now by the update kernel's own arithmetic: the per-coordinate form adds in the kernel's reduction order)
    }
    return 0;
}

// which form of the device likelihood runs: *fused = 1 the update kernel compiled around it (one launch per half generation), 0 the three-kernel form
// (then `why`, if given, says why: up to why_cap - 1 characters); error when no device likelihood is installed
extern "C" int bpm_get_device_likelihood_info(bpm_handle_t s, int32_t* fused, char* why, int64_t why_cap) {
    CK(check_handle(s));
    if (!s->user_fn) return fail("bpm_get_device_likelihood_info: no device likelihood installed (bpm_set_device_likelihood)");
    if (fused) *fused = s->user_fused_fn ? 1 : 0;
    if (why && why_cap > 0) {
        const size_t n = std::min(s->user_fused_why.size(), (size_t)why_cap - 1);
        std::memcpy(why, s->user_fused_why.data(), n);
        why[n] = '\0';
    }
    return 0;
}

extern "C" int bpm_refresh_device_loglike(bpm_handle_t s) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->user_fn) return fail("bpm_refresh_device_loglike: no device likelihood installed (bpm_set_device_likelihood)");
    return user_refresh_ll(s);
}

// the caller-owned-buffer form of rounds 1-4, compacted to the active work items.  Since round 5 the read-back runs in pieces INSIDE the call: the
// compaction copy of piece k (one host core reading freshly DMA'd memory: ~100 us for cfg2's 3.3 MB) runs under the DMA of piece k + 1 (~130 us); four
// pieces from 1 MB per half generation.
extern "C" int bpm_propose(bpm_handle_t s, double* out_prop, int32_t* out_ids, int32_t* n_out) {
    CK(check_handle(s));
    if (!out_prop || !out_ids || !n_out) return fail("bpm_propose: null argument");
    if (s->proposed) return fail("bpm_propose: previous proposals not committed");
    const size_t bytes = (size_t)(s->n_local / 2u + 1u) * s->ld * sizeof(double);
    static const int forced = getenv("BPM_PROPOSE_PIECES") ? atoi(getenv("BPM_PROPOSE_PIECES")) : 0;      // (A/B switch: tools/host_callback_ab.py)
    const int32_t pieces = forced > 0 ? std::min(forced, 256) : (bytes >= ((size_t)1 << 20) ? 4 : 1);
    // (cfg2's shape, 3.3 MB per half generation, two runs each: 1 piece 6.5 / 6.5e6 chain-updates/s, 2 pieces 6.9 / 5.7, 4 pieces 8.3 / 7.0, 8 pieces 5.7 / 5.5 --
    // every piece costs an event wait and a DMA submission: profiles/r05_host_callback.txt)
    CK(bpm_propose_begin(s, pieces));
    int32_t n = 0;
    for (int32_t k = 0; k < pieces; ++k) {
        const double* props = nullptr; const int32_t* ids = nullptr; int32_t nr = 0, ld = 0;
        CK(bpm_propose_chunk(s, k, &props, &ids, &nr, &ld));
        // compact the active work items (work-item order is kept; commit uses the same order)
        if (s->ld == s->dim && nr > 0) {                                  // (runs of active rows in one copy)
            int32_t w = 0;
            while (w < nr) {
                if (ids[w] < 0) { ++w; continue; }
                int32_t e = w;
                while (e < nr && ids[e] >= 0) ++e;
                std::memcpy(out_prop + (size_t)n * s->dim, props + (size_t)w * s->ld, (size_t)(e - w) * s->dim * sizeof(double));
                std::memcpy(out_ids + n, ids + w, (size_t)(e - w) * sizeof(int32_t));
                n += e - w; w = e;
            }
        } else {
            for (int32_t w = 0; w < nr; ++w) {
                if (ids[w] < 0) continue;
                std::memcpy(out_prop + (size_t)n * s->dim, props + (size_t)w * s->ld, s->dim * sizeof(double));
                out_ids[n++] = ids[w];
            }
        }
    }
    *n_out = n;
    s->prop_whole = true;
    return 0;
}

extern "C" int bpm_commit(bpm_handle_t s, const double* ll_prop) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->proposed) return fail("bpm_commit: nothing proposed");
    if (s->prop_mode != 1 || !s->prop_whole) return fail("bpm_commit: the open half generation was proposed with another entry point (bpm_propose_begin / bpm_propose_device)");
    const PhaseArgs& a = s->cur_args[s->phase];
    // scatter the values back to work-item order (the ids are still in the pinned buffer bpm_propose filled)
    std::vector<double> full(a.n_items, 0.0);
    size_t n = 0;
    for (uint32_t w = 0; w < a.n_items; ++w) {
        if (s->h_ids[w] < 0) continue;
        if (!ll_prop) return fail("bpm_commit: null ll_prop");
        full[w] = ll_prop[n++];
    }
    for (int32_t k = 0; k < s->prop_chunks; ++k) CK(bpm_commit_chunk(s, k, full.data() + chunk_lo(a.n_items, s->prop_chunks, k)));
    return bpm_commit_end(s);
}

// ---- the likelihood evaluated ON THE DEVICE by the caller's own framework (torch, cupy ...): nothing crosses PCIe --------------------------
extern "C" int bpm_propose_device(bpm_handle_t s, const double** rows_dev, const int32_t** ids_dev, int32_t* n_rows, int32_t* ld) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!rows_dev || !ids_dev || !n_rows || !ld) return fail("bpm_propose_device: null argument");
    CK(propose_launch(s, "bpm_propose_device"));
    const PhaseArgs& a = s->cur_args[s->phase];
    // (the active-row count for the accept bookkeeping: the ids come back, 32 KB; with one rank every work item is active)
    HIPCK(hipMemcpyAsync(s->h_ids, s->ids_buf, s->n_local * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));                  // the proposals are complete: the caller's framework reads them on ITS stream
    int64_t act = 0;
    for (uint32_t w = 0; w < a.n_items; ++w) act += s->h_ids[w] >= 0 ? 1 : 0;
    s->prop_active = act;
    *rows_dev = s->prop_buf; *ids_dev = s->ids_buf; *n_rows = (int32_t)a.n_items; *ld = (int32_t)s->ld;
    s->prop_mode = 2; s->prop_chunks = 0;
    s->proposed = true;
    return 0;
}

extern "C" int bpm_commit_device(bpm_handle_t s, const double* ll_dev) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->proposed || s->prop_mode != 2) return fail("bpm_commit_device: nothing proposed with bpm_propose_device");
    const PhaseArgs& a = s->cur_args[s->phase];
    if (a.n_items > 0) {
        if (!ll_dev) return fail("bpm_commit_device: null ll_dev");
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ll_dev) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != s->cfg.device) {
            (void)hipGetLastError();
            return fail("bpm_commit_device: ll_dev must be device memory of this sampler's GPU (float64, one value per proposal row, contiguous)");
        }
        // the caller's framework may have computed the values on any stream of this device: wait for the device, then copy on ours
        HIPCK(hipDeviceSynchronize());
        HIPCK(hipMemcpyAsync(s->aux_buf + s->n_local, ll_dev, (size_t)a.n_items * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    }
    return commit_finish(s, s->prop_active);
}

// the local chains' CURRENT states where they lie in device memory (n_local rows of *ld doubles), and their ln-likes from device memory: what a
// device-resident likelihood needs once per (re)initialisation (the host form: bpm_get_state + bpm_set_loglike)
extern "C" int bpm_state_device(bpm_handle_t s, const double** rows_dev, int32_t* n_rows, int32_t* ld) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!rows_dev || !n_rows || !ld) return fail("bpm_state_device: null argument");
    HIPCK(hipStreamSynchronize(s->stream));
    *rows_dev = s->G + (uint64_t)s->rank * s->L.blk; *n_rows = (int32_t)s->n_local; *ld = (int32_t)s->ld;
    return 0;
}
extern "C" int bpm_set_loglike_device(bpm_handle_t s, const double* ll_dev) {
    CK(check_handle(s));
    CK(set_device(s));
    if (s->cfg.target_id != BPM_TARGET_HOST_CALLBACK) return fail("bpm_set_loglike_device: host-callback targets only");
    hipPointerAttribute_t at;
    if (!ll_dev || hipPointerGetAttributes(&at, ll_dev) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != s->cfg.device) {
        (void)hipGetLastError();
        return fail("bpm_set_loglike_device: ll_dev must be device memory of this sampler's GPU (float64, n_local values)");
    }
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpyAsync(s->ll, ll_dev, s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    if (s->hist_rows >= 1 && s->hist_rows == s->rows_logical)
        HIPCK(hipMemcpyAsync(s->llhist + (size_t)(s->hist_rows - 1) * s->n_local, s->ll, s->n_local * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_reserve_history(bpm_handle_t s, int64_t total_rows) {
    CK(check_handle(s));
    CK(set_device(s));
    return ensure_history(s, total_rows);
}

// Large device-to-host transfers into the caller's pageable buffer.  HIP's own pageable path stages through pinned
// memory with ONE host thread copying out of it (~10 GB/s here, and the caller's fresh NumPy array page-faults on
// first touch on that same thread).  Here `n_workers` host threads, each with its own stream and two pinned staging
// buffers, take chunks from a shared counter: the DMA of a thread's next chunk overlaps its memcpy of the current one.
static int d2h_rows_parallel(bpm_sampler* s, double* out, const double* src, size_t rows, uint32_t dim, uint32_t ld) {
    const size_t row_bytes = (size_t)dim * sizeof(double);
    const size_t chunk_rows = std::max<size_t>(1, ((size_t)8 << 20) / row_bytes);
    const size_t n_chunks = (rows + chunk_rows - 1) / chunk_rows;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int n_workers = (int)std::min<size_t>(std::min<size_t>(8, std::max(1u, hw / 2)), n_chunks);
    {   // the caller's buffer is typically a fresh anonymous mapping (np.empty): ask for transparent huge pages before the
        // first touch -- 800 k page faults of 4 KiB otherwise cost as much as the copy itself
        const uintptr_t b0 = ((uintptr_t)out + 4095u) & ~(uintptr_t)4095u, b1 = ((uintptr_t)out + rows * row_bytes) & ~(uintptr_t)4095u;
        if (b1 > b0) (void)madvise((void*)b0, (size_t)(b1 - b0), MADV_HUGEPAGE);
    }
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    const int device = s->cfg.device;
    auto work = [&]() {
        hipStream_t st = nullptr;
        double* pin[2] = {nullptr, nullptr};
        auto ok = [&](hipError_t e) { if (e != hipSuccess) { err.store((int)e); return false; } return true; };
        if (ok(hipSetDevice(device)) && ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) &&
            ok(hipHostMalloc((void**)&pin[0], chunk_rows * row_bytes, hipHostMallocDefault)) &&
            ok(hipHostMalloc((void**)&pin[1], chunk_rows * row_bytes, hipHostMallocDefault))) {
            auto issue = [&](size_t ch, int b) {
                const size_t r0 = ch * chunk_rows, nr = std::min(chunk_rows, rows - r0);
                return ok(hipMemcpy2DAsync(pin[b], row_bytes, src + r0 * ld, (size_t)ld * sizeof(double), row_bytes, nr,
                                           hipMemcpyDeviceToHost, st));
            };
            size_t cur = next.fetch_add(1);
            int b = 0;
            bool live = cur < n_chunks && issue(cur, b);
            while (live && err.load() == 0) {
                if (!ok(hipStreamSynchronize(st))) break;                 // chunk `cur` is in pin[b]
                const size_t nxt = next.fetch_add(1);
                const bool more = nxt < n_chunks;
                if (more && !issue(nxt, b ^ 1)) break;                    // next DMA runs under this memcpy
                const size_t r0 = cur * chunk_rows, nr = std::min(chunk_rows, rows - r0);
                std::memcpy(out + r0 * dim, pin[b], nr * row_bytes);
                if (!more) break;
                cur = nxt;
                b ^= 1;
            }
            if (st) (void)hipStreamSynchronize(st);
        }
        if (pin[0]) (void)hipHostFree(pin[0]);
        if (pin[1]) (void)hipHostFree(pin[1]);
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_workers; ++w) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (err.load() != 0) return fail(std::string("bpm_get_history: ") + hipGetErrorString((hipError_t)err.load()));
    return 0;
}

extern "C" int bpm_get_history(bpm_handle_t s, int64_t g_lo, int64_t g_hi, double* out) {
    CK(check_handle(s));
    CK(set_device(s));
    if (g_lo < 0 || g_hi < g_lo || g_hi > s->hist_rows) return fail("bpm_get_history: generation range out of bounds");
    if (g_hi == g_lo) return 0;
    if (!out) return fail("bpm_get_history: null argument");
    CK(normalize_history(s, g_lo, g_hi));
    const size_t rows = (size_t)(g_hi - g_lo) * s->n_local;
    if (rows * s->dim * sizeof(double) >= ((size_t)128 << 20)) {
        HIPCK(hipStreamSynchronize(s->stream));
        return d2h_rows_parallel(s, out, s->hist + (uint64_t)g_lo * s->n_local * s->ld, rows, s->dim, s->ld);
    }
    HIPCK(hipMemcpy2DAsync(out, s->dim * sizeof(double), s->hist + (uint64_t)g_lo * s->n_local * s->ld,
                           s->ld * sizeof(double), s->dim * sizeof(double), rows, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_get_loglike_history(bpm_handle_t s, int64_t g_lo, int64_t g_hi, double* out) {
    CK(check_handle(s));
    CK(set_device(s));
    if (g_lo < 0 || g_hi < g_lo || g_hi > s->hist_rows) return fail("bpm_get_loglike_history: range out of bounds");
    if (g_hi == g_lo) return 0;
    CK(normalize_history(s, g_lo, g_hi));
    HIPCK(hipMemcpyAsync(out, s->llhist + (uint64_t)g_lo * s->n_local, (size_t)(g_hi - g_lo) * s->n_local * sizeof(double),
                         hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int bpm_get_stats(bpm_handle_t s, bpm_stats_t* out) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!out) return fail("bpm_get_stats: null argument");
    unsigned long long c[8];
    double cr[3 * MAX_CR];
    std::vector<uint32_t> acc((size_t)s->n_local + ACC_SHARDS);       // per-chain counters and per-wavefront shards: the total is what counts
    HIPCK(hipMemcpyAsync(c, s->counters, sizeof(c), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipMemcpyAsync(cr, s->cr_state, sizeof(cr), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipMemcpyAsync(acc.data(), s->acc_count, acc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    std::memset(out, 0, sizeof(*out));
    int64_t n_acc = 0;
    for (uint32_t v : acc) n_acc += v;
    // every local chain is updated exactly once per generation (it is in exactly one of the two pools)
    const int64_t n_upd = s->k_gen * (int64_t)s->n_local + (s->phase == 1 ? s->phase_a_updates : 0);
    out->local_n_accepted = n_acc;                       // demc.py:67,190
    out->local_n_rejected = 1 + n_upd - n_acc;           // demc.py:68,193: starts at 1
    out->n_nan_alpha = (int64_t)c[2];
    out->k_gen = s->k_gen;
    out->t_abs = s->t_abs;
    out->history_rows = s->hist_rows;
    out->n_outlier_resets = (int64_t)c[4];
    out->n_cr = s->cfg.n_cr;
    for (int m = 0; m < MAX_CR; ++m) {
        out->p_cr[m] = cr[m];
        out->delta_m[m] = cr[MAX_CR + m];
        out->n_cr_updates[m] = cr[2 * MAX_CR + m];
    }
    return 0;
}

extern "C" int bpm_set_adapt_state(bpm_handle_t s, const double* p_cr, const double* delta_m, const double* n_cr_updates,
                                   int64_t t_abs) {
    CK(check_handle(s));
    CK(set_device(s));
    double cr[3 * MAX_CR] = {0};
    for (int m = 0; m < s->cfg.n_cr; ++m) {
        cr[m] = p_cr ? p_cr[m] : 1.0 / s->cfg.n_cr;
        cr[MAX_CR + m] = delta_m ? delta_m[m] : 0.0;
        cr[2 * MAX_CR + m] = n_cr_updates ? n_cr_updates[m] : 0.0;
    }
    HIPCK(hipMemcpyAsync(s->cr_state, cr, sizeof(cr), hipMemcpyHostToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    if (t_abs >= 0) s->t_abs = t_abs;
    return 0;
}

extern "C" int bpm_eval_loglike(bpm_handle_t s, const double* X, int32_t n, double* out) {
    CK(check_handle(s));
    CK(set_device(s));
    if (n <= 0) return 0;
    if (!X || !out) return fail("bpm_eval_loglike: null argument");
    double* dX = nullptr;
    double* dO = nullptr;
    CK(dev_alloc(&dX, (size_t)n * s->ld));
    CK(dev_alloc(&dO, (size_t)n));
    HIPCK(hipMemsetAsync(dX, 0, (size_t)n * s->ld * sizeof(double), s->stream));
    HIPCK(hipMemcpy2DAsync(dX, s->ld * sizeof(double), X, s->dim * sizeof(double), s->dim * sizeof(double), n,
                           hipMemcpyHostToDevice, s->stream));
    switch (s->cfg.target_id) {
        case BPM_TARGET_GAUSS_EQUICORR: g_eval_gauss[s->shape.idx](dX, n, s->ld, s->dim, s->tparams, dO, s->stream); break;
        case BPM_TARGET_MIXTURE_PAIRS: g_eval_mixture[s->shape.idx](dX, n, s->ld, s->dim, s->tparams, dO, s->stream); break;
        case BPM_TARGET_BANANA_2D: launch_eval<TARGET_BANANA, 1, 2>(dX, n, s->ld, s->dim, s->tparams, dO, s->stream); break;
        default: (void)hipFree(dX); (void)hipFree(dO); return fail("bpm_eval_loglike: host-callback target has no device ln_like");
    }
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(out, dO, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    HIPCK(hipFree(dX));
    HIPCK(hipFree(dO));
    return 0;
}

#ifdef BPM_TEST_HOOKS
// ---- debug / parity hooks (include/bipymc_hip_test.h; build_variants/libbipymc_test.so only) ------------------------------------------
extern "C" int bpm_set_trace(bpm_handle_t s, int32_t on) {
    CK(check_handle(s));
    CK(set_device(s));
    HIPCK(hipStreamSynchronize(s->stream));
    if (on && !s->trace_i32) {
        CK(dev_alloc(&s->trace_i32, (size_t)s->n_local * TRACE_I32));
        CK(dev_alloc(&s->trace_f64, (size_t)s->n_local * TRACE_F64));
        CK(dev_alloc(&s->trace_mask, (size_t)s->n_local * s->dim));
        HIPCK(hipMemset(s->trace_i32, 0xFF, (size_t)s->n_local * TRACE_I32 * sizeof(int32_t)));
        HIPCK(hipMemset(s->trace_f64, 0, (size_t)s->n_local * TRACE_F64 * sizeof(double)));
        HIPCK(hipMemset(s->trace_mask, 0, (size_t)s->n_local * s->dim));
    } else if (!on && s->trace_i32) {
        HIPCK(hipFree(s->trace_i32)); HIPCK(hipFree(s->trace_f64)); HIPCK(hipFree(s->trace_mask));
        s->trace_i32 = nullptr; s->trace_f64 = nullptr; s->trace_mask = nullptr;
    }
    return 0;
}

// trace of the LAST generation: out_i32 [n_local*32] = (cr_idx, d_prime, jump, accepted, snooker, partners[23]..),
// out_f64 [n_local*4] = (alpha, ll_prop, delta, gamma), out_mask [n_local*dim]
extern "C" int bpm_get_trace(bpm_handle_t s, int32_t* out_i32, double* out_f64, uint8_t* out_mask) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->trace_i32) return fail("bpm_get_trace: tracing is off");
    HIPCK(hipStreamSynchronize(s->stream));
    if (out_i32) HIPCK(hipMemcpy(out_i32, s->trace_i32, (size_t)s->n_local * TRACE_I32 * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_f64) HIPCK(hipMemcpy(out_f64, s->trace_f64, (size_t)s->n_local * TRACE_F64 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_mask) HIPCK(hipMemcpy(out_mask, s->trace_mask, (size_t)s->n_local * s->dim, hipMemcpyDeviceToHost));
    return 0;
}

// the host-side per-generation decisions, for parity against oracle/philox_ref.py
extern "C" int bpm_debug_perm(bpm_handle_t s, int64_t t, int32_t shuffle, double flip_prob, int32_t* out_order,
                              int32_t* out_inverse, int32_t* out_flip) {
    CK(check_handle(s));
    const PermKey pk = make_perm_key(s->cfg.seed, (uint64_t)t, s->N, shuffle != 0);
    for (uint32_t x = 0; x < s->N; ++x) {
        if (out_order) out_order[x] = (int32_t)perm_fwd(x, pk);
        if (out_inverse) out_inverse[x] = (int32_t)perm_inv(x, pk);
    }
    if (out_flip) *out_flip = flip_draw(s->cfg.seed, (uint64_t)t, flip_prob) ? 1 : 0;
    return 0;
}

extern "C" int bpm_selftest_philox(int32_t device, int32_t n, uint64_t seed, uint32_t* out_mine, uint32_t* out_rocrand) {
    if (n <= 0 || !out_mine || !out_rocrand) return fail("bpm_selftest_philox: bad argument");
    HIPCK(hipSetDevice(device));
    uint32_t* d = nullptr;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 8 * sizeof(uint32_t)));
    launch_rocrand_check(d, n, seed);
    HIPCK(hipGetLastError());
    HIPCK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)n * 8);
    HIPCK(hipMemcpy(h.data(), d, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCK(hipFree(d));
    for (int i = 0; i < n; ++i) {
        std::memcpy(out_mine + 4 * (size_t)i, h.data() + 8 * (size_t)i, 16);
        std::memcpy(out_rocrand + 4 * (size_t)i, h.data() + 8 * (size_t)i + 4, 16);
    }
    return 0;
}

// Test hook: the quartile / best-chain selection of the outlier check on caller-supplied omega values (world_size 1, sampler created
// with outlier_every > 0): out = (sorted[k0], sorted[k0 + 1], sorted[k1], sorted[k1 + 1], first argmax, cut = Q1 - 2 IQR) with the
// order statistics np.percentile(omega, [25, 75]) interpolates between.  tests/test_gpu_parity.py compares with NumPy on ties,
// infinities, signed zeros and all-equal inputs.
extern "C" int bpm_debug_outlier_select(bpm_handle_t s, const double* omega, double out[6]) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!omega || !out) return fail("bpm_debug_outlier_select: null argument");
    if (!s->om || s->world != 1) return fail("bpm_debug_outlier_select: needs a single-rank sampler created with outlier_every > 0");
    const uint32_t N = s->N;
    HIPCK(hipMemcpyAsync(s->om, omega, (size_t)N * sizeof(double), hipMemcpyHostToDevice, s->stream));
    SelRanks R;
    double tq[2];
    for (int i = 0; i < 2; ++i) {
        const double pos = (i == 0 ? 25.0 : 75.0) / 100.0 * (double)(N - 1);
        const uint32_t lo = (uint32_t)std::floor(pos);
        R.k[2 * i] = lo;
        R.k[2 * i + 1] = std::min(lo + 1u, N - 1u);
        tq[i] = pos - (double)lo;
    }
    launch_outlier_select(s, R);
    HIPCK(hipGetLastError());
    double h[5];
    HIPCK(hipMemcpyAsync(h, s->sel, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < 5; ++i) out[i] = h[i];
    auto lerp = [](double a, double b, double t) { const double d = b - a; return t >= 0.5 ? b - d * (1.0 - t) : a + d * t; };
    const double q1 = lerp(h[0], h[1], tq[0]), q3 = lerp(h[2], h[3], tq[1]);
    out[5] = q1 - 2.0 * (q3 - q1);
    return 0;
}

// Diagnostic (tools/emulate_ranks.py): the update kernel and the replay kernel of this handle's LAST half generation launched
// `reps` times back to back and timed with an event pair -- DESTRUCTIVE (the replica is advanced again and again), for timing
// only.  Why: R ranks emulated on one GPU share ONE Infinity Cache, so kernel times read from a trace of the lock-step run are
// those of a GPU whose cache holds eight replicas; re-launching one rank's kernels alone shows them with that rank's data warm.
extern "C" int bpm_debug_time_kernels(bpm_handle_t s, int32_t reps, float* update_us, float* replay_us) {
    CK(check_handle(s));
    CK(set_device(s));
    if (reps < 1 || !update_us || !replay_us) return fail("bpm_debug_time_kernels: bad argument");
    PhaseLaunch fn = pick_fused(s);
    if (!fn) return fail("bpm_debug_time_kernels: device targets only");
    const PhaseArgs& a = s->cur_args[1];
    if (a.n_items == 0) return fail("bpm_debug_time_kernels: run at least one generation first");
    HIPCK(hipStreamSynchronize(s->stream));
    *update_us = 0.f; *replay_us = 0.f;
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && (!s->accbits_all || s->world == 1 || a.accbits == nullptr)) break;     // no replay exchange on this handle
        PhaseArgs r = a;
        if (which == 1) {
            r.replay = 1u; r.accbits = nullptr; r.accbits_all = s->accbits_all;
            trace_set(r, nullptr, nullptr, nullptr); r.pack = nullptr; r.hist_row = nullptr; r.llhist_row = nullptr;
            r.adapt_on = 0u;
        }
        for (int i = 0; i < 3; ++i) { if (which == 0) fn(r, s->stream); else launch_replay_any(s, r); }
        HIPCK(hipEventRecord(s->ev0, s->stream));
        for (int i = 0; i < reps; ++i) { if (which == 0) fn(r, s->stream); else launch_replay_any(s, r); }
        HIPCK(hipEventRecord(s->ev1, s->stream));
        HIPCK(hipEventSynchronize(s->ev1));
        HIPCK(hipGetLastError());
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        (which == 0 ? *update_us : *replay_us) = ms * 1e3f / (float)reps;
    }
    return 0;
}

#endif   // BPM_TEST_HOOKS

#ifdef BPM_STAMPS
// diagnostic build only: per-work-item s_memtime stamps of the LAST launched phase kernel
extern "C" int bpm_debug_stamps(bpm_handle_t s, unsigned long long* out, int64_t n_items) {
    CK(check_handle(s));
    CK(set_device(s));
    if (!s->stamps) {
        HIPCK(hipMalloc(reinterpret_cast<void**>(&s->stamps), (size_t)s->n_local * 8 * sizeof(unsigned long long)));
        HIPCK(hipMemset(s->stamps, 0, (size_t)s->n_local * 8 * sizeof(unsigned long long)));
        return 0;
    }
    HIPCK(hipStreamSynchronize(s->stream));
    if (out) HIPCK(hipMemcpy(out, s->stamps, (size_t)n_items * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif
