// HIP kernels of the per-generation hot path (gfx950 / MI355X).
//
// One *chain subgroup* of LPC lanes (1..64, power of two) updates one chain; a
// lane owns DPL/2 coordinate PAIRS (2*pi, 2*pi+1), pi = q + u*LPC, so every row
// access is one 16-byte load per lane, contiguous across the subgroup.  d = 100
// -> LPC = 64 (one wavefront per chain, 50 lanes hold data); d = 8 -> LPC = 4
// (16 chains per wavefront); d = 2 -> LPC = 1 (64 chains per wavefront).
// Workgroup = one wavefront (64 threads): no workgroup barrier is ever needed
// and the grid is n_items / (64/LPC) workgroups (>> 256 at the benchmark sizes).
//
// What one launch does = one half generation of demc.py:103-109 / 126-132: every
// chain of group `upd` is updated against the frozen complementary pool:
//   DREAM  dream.py:32-107  CR index, subspace mask, P distinct pairs, gamma (+jump
//          every 5th generation), uniform + normal jitter, CR statistic, Metropolis,
//          append to history, Welford moments of the chain's own history;
//   DE-MC  demc.py:153-196  one pair, gamma (+jump every 10th), normal jitter,
//          optional snooker update (extension), Metropolis, append.
// ln_like of the shipped analytic targets is evaluated in registers (TARGET).
// For an arbitrary host ln_like_fn the same code is split into a propose and a
// commit kernel (STAGE).
#pragma once
#ifndef __HIPCC_RTC__      // (under hiprtc -- user_likelihood.h compiles this header at run time around a caller's likelihood -- both come built in)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "philox.h"

// Diagnostic build only (-DBPM_STAMPS): s_memtime stamps along one wavefront's critical path, written to
// PhaseArgs::stamps (8 x u64 per work item).  Never compiled into the product library.
#ifdef BPM_STAMPS
#define BPM_STAMP(i)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long _t;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        bpm_stamp[i] = _t;                                                                    \
    } while (0)
#else
#define BPM_STAMP(i) do { } while (0)
#endif

// The last-workgroup ticket below (outlier_select_pass_kernel) publishes with relaxed agent-scope atomics,
// `s_waitcnt vmcnt(0)`, then a relaxed fetch_add on the ticket.  That is NOT release/acquire in the HIP / LLVM memory model; it is
// correct on gfx942 / gfx950 because (a) stores and atomics are counted in vmcnt there (gfx10+ counts stores in vscnt) and (b) agent-scope
// atomics are performed at the memory side, past the per-XCD L2s -- an agent-scope release would instead write back the whole L2
// (buffer_wbl2), which is the cost this form avoids.  Any other target must not compile this file silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "kernels.h: the ticket hand-overs rely on gfx942 / gfx950 memory behaviour (vmcnt counts stores; agent-scope atomics bypass the L2s): use __ATOMIC_RELEASE / __ATOMIC_ACQUIRE tickets on other targets"
#endif

namespace bpm {
inline namespace BPM_VARIANT_NS {      // (philox.h: one kernel-symbol namespace per build variant)

constexpr int ALGO_DEMC = 0, ALGO_DREAM = 1;
constexpr int TARGET_HOST = 0, TARGET_GAUSS = 1, TARGET_MIXTURE = 2, TARGET_BANANA = 3;
constexpr int STAGE_FUSED = 0, STAGE_PROPOSE = 1, STAGE_COMMIT = 2;
constexpr int WAVE = 64;
// Threads per workgroup of the update kernels, by lanes-per-chain: the wavefronts of a workgroup are independent
// (no workgroup barrier anywhere), so this is purely a dispatch-granularity choice.  Measured (profiles/): one
// wavefront per workgroup is best for one-wavefront-per-chain (d=100) and one-lane-per-chain (d=2) kernels, four
// wavefronts for the 4-lanes-per-chain kernel (d=8: 14.8 vs 17.0 us/generation at N=32768).
// (re-measured with the final kernel: 128- or 256-thread workgroups for one wavefront per chain stay within 1 % of 64)
// Workgroup size of the one-wavefront-per-chain and one-lane-per-chain shapes.  Two wavefronts per workgroup halve the number of
// workgroups the dispatcher has to place: 11.00 vs 11.35 us per generation at cfg2, 13.66 vs 13.94 at cfg3 once launches are no longer
// host-paced (round 2, direct queue; under HIP launches round 1 measured no difference).  256 is as good at cfg2 and worse for one lane
// per chain (15.5 us at cfg3: too few workgroups).  The kernels use no barrier and no cross-wavefront LDS in these shapes.
#ifndef BPM_BLOCK_WAVE
#define BPM_BLOCK_WAVE 128
#endif
constexpr int block_for(int lpc) { return (lpc == 4 || lpc == 16) ? 256 : BPM_BLOCK_WAVE; }
// ... and of the burn-in flavours (HOT 3 / 4) of the one-wavefront-per-chain shapes: 16 wavefronts = 16 chains per workgroup, whose CR statistics
// the workgroup sums itself (cr_level1: "CR reduction" below).  1024-thread workgroups cost the update nothing (cfg2 steady: 10.8 us either way).
constexpr bool hot_is_adapt(int hot) { return hot == 3 || hot == 4; }
// (one wavefront per chain: only with 2 coordinates per lane, d <= 128 -- a 1024-thread workgroup caps the kernel at 128 VGPRs, and the burn-in kernels
// of the wider register-resident shapes need more: those shapes take level 1 from the slots, like every other path)
constexpr bool crp_shape(int lpc, int dpl) { return lpc < WAVE || dpl == 2; }
// (several chains per wavefront keep their 256-thread workgroups in burn-in too: with 1024 threads -- 256 chains per level-1 chunk, no cr_mid_kernel pass
// at cfg5 -- the register budget allows one workgroup per CU: cfg5's share, 64 workgroups, 16.8 -> 28.2 us per generation, cfg5 70.0 -> 71.5)
constexpr int block_for_hot(int lpc, int hot, int dpl) { return (lpc == WAVE && dpl == 2 && hot_is_adapt(hot)) ? 1024 : block_for(lpc); }
// chains (positions) per level-1 partial sum of the CR statistics: a function of the kernel SHAPE alone (every rank, every launch path agrees)
constexpr int cr_g1(int lpc) { return lpc == WAVE ? 16 : WAVE / lpc; }
constexpr int MAX_CR = 8;
constexpr int TRACE_I32 = 32;   // ints per chain in the debug trace
constexpr int TRACE_F64 = 4;
constexpr int MAX_PARTNERS = 2 * MAX_PAIRS + 3;

// Exchange buffer: world blocks of [X_local (n_local x ld) | delta (n_local) | cr_idx (n_local)],
// rank r's block at G + r*blk.  An in-place all-gather of `blk` doubles per rank replaces the
// MPI_Allgather of demc.py:93-94,116-117; with world == 1 it is the plain (N x ld) state matrix.
struct Layout {
    double* G;
    uint64_t blk;      // doubles per rank block = n_local * (ld + 2)
    uint32_t n_local;
    uint32_t ld;       // row stride in doubles (dim rounded up to even: rows stay 16-byte aligned)
    uint32_t dim;
    uint32_t world;
    uint32_t magic;    // floor(2^32 / n_local) + 1 (c < 2^31)
};

// Addresses in 32-bit element offsets (the whole exchange buffer is < 2^31 doubles, checked in bpm_create).
// With blk = n_local (ld + 2): row(c) = c ld + r 2 n_local, delta(c) = c + r n_local (ld + 1) + n_local ld, r = c / n_local.
__device__ __forceinline__ uint32_t rank_of(const Layout& L, uint32_t c) {
    if (L.world == 1) return 0u;
    // c / n_local without a hardware divide: magic = floor(2^32 / n_local) + 1 over-estimates by at most one
    uint32_t r = __umulhi(c, L.magic);
    r -= (r * L.n_local > c) ? 1u : 0u;
    return r;
}
__device__ __forceinline__ double* row_ptr(const Layout& L, uint32_t c) {
    return L.G + (c * L.ld + rank_of(L, c) * (2u * L.n_local));
}
__device__ __forceinline__ double* delta_ptr(const Layout& L, uint32_t c) {
    return L.G + (c + rank_of(L, c) * (L.n_local * (L.ld + 1u)) + L.n_local * L.ld);
}
__device__ __forceinline__ double* cridx_ptr(const Layout& L, uint32_t c) {
    return L.G + (c + rank_of(L, c) * (L.n_local * (L.ld + 1u)) + L.n_local * (L.ld + 1u));
}

// Kernels with several chains per wavefront (LPC < 64: d <= 32) and a device target keep no per-chain scalars on their path (round 4):
//   * ln_like of the CURRENT state is re-evaluated from the own row, which the update has in registers anyway, instead of read from the
//     cache `ll` -- a scattered 8-byte load per update that cost a memory transaction of its own (and, on accept, a scattered 8-byte store).
//     The value is the one the cache would hold, bit for bit: the same Target::eval on the same row (the row was written by the update that
//     produced the cached value).  The reference re-evaluates too (samplers.py:330).  The host refreshes the cache with one eval_ll_kernel
//     where something else reads it (bpm_get_loglike, the outlier check): sampler.hip, ll_stale.
//   * accepted updates are counted per WAVEFRONT (ballot + popcount) into ACC_SHARDS counters behind the per-chain ones
//     (acc_count[n_local + shard]); the host sums everything (demc.py:143-150 needs the total only).
// Timing-only build without that traffic (profiles/r04_small_d.txt): cfg3 11.47 -> 10.47 us per generation, cfg5 47.9 -> 43.4; built for real:
// 10.9 and 45.3.  It is a RUN-TIME choice of the host (PhaseArgs::lean, the same bits either way): a launch that is latency bound rather
// than transaction bound -- cfg5's per-GPU share, 32768 chains: one wavefront per SIMD -- pays for the extra exp / log of the re-evaluation
// (9.3 -> 9.9 us per generation) and keeps the cached form; from 49152 chains per GPU the lean form runs.
constexpr int ACC_SHARDS = 1024;
template <int TARGET, int LPC>
constexpr bool lean_scalars() { return LPC < WAVE && TARGET != TARGET_HOST; }
// ... and where the target costs a dozen flops (the banana: no exp, no log) the lean form is the only one compiled: as a run-time switch it
// cost cfg3 0.2 us per generation (11.1 instead of 10.9)
template <int TARGET>
constexpr bool lean_always() { return TARGET == TARGET_BANANA; }

constexpr int PLAN_WORDS = 16;    // chain id | header block (4 words) | up to 10 partner ids | pad
constexpr int MAX_SEG = 16;       // ranks the owner-sorted record table serves (more: records by position, a wavefront per position replays)
constexpr int MAX_PEERS = MAX_SEG - 1;   // other ranks an owner pushes its accepted rows to (push exchange)

// Push exchange (world > 1): the OWNER of a chain writes an accepted row straight into every other rank's replica of
// the state matrix -- peer memory mapped into this process (hipIpcOpenMemHandle; over xGMI on a multi-GPU node) -- where the reference
// has MPI_Allgather (demc.py:93-94,116-117).  What orders the half generations across ranks is one small block per rank:
//   flag[p]   last barrier sequence number rank p has ANNOUNCED to this rank (written by p's push_sync_kernel, read by this rank's)
//   err       set by this rank's push_sync_kernel when a wait ran into its time limit: 1 + the rank it was waiting for
//   probe[p]  connection self-test pattern written by rank p
//   arena_bad[f]  connection self-test of the arena (bpm_push_selftest): words of the probe rows the peers stored into THIS rank's
//             arena that did not arrive, under packet fences of form f (0 system scope, 1 agent scope); written by this rank only
// A flag >= PUSH_CLOSING says "rank p is destroying its sampler" (bpm_destroy's hand-over): a rank that still waits for p learns it at once
// (err = PUSH_ERR_CLOSED + p) instead of running into the time limit.
constexpr unsigned long long PUSH_CLOSING = 1ull << 62;
constexpr unsigned long long PUSH_ERR_CLOSED = 0x100ull;
struct PushCtrl {
    unsigned long long flag[MAX_SEG];
    unsigned long long err;
    unsigned long long arena_bad[2];
    unsigned long long pad[5];
    unsigned long long probe[MAX_SEG];
};
struct PhaseArgs {
    Layout L;
    double* ll;            // [n_local] cached ln_like of the local chains (samplers.py:330 re-evaluates it)
    double* hist_row;      // [n_local * ld] history row being appended (chain.py:51-54) or nullptr
    double* llhist_row;    // [n_local] or nullptr
    double* w_mean;        // Welford moments of each chain's own history (dream.py:128), ONE record per chain: [mean (ld) | m2 (ld)], chain li at w_mean + li * 2 ld;
    double* w_m2;          // w_m2 = w_mean + ld (same stride).  (Two arrays until round 4: two scattered reads and writes per burn-in update where one does.)
    const double* tparams; // target parameter block
    const double* cr_state;  // p_cr[MAX_CR] | delta_m[MAX_CR] | n_cr_updates[MAX_CR]
    unsigned long long* counters;  // [2] = NaN Metropolis ratios (rare; the only atomic)
    uint32_t* acc_count;   // [n_local] accepted updates of each local chain in this run (one writer per chain)
    double* prop_buf;      // host-callback path: [n_local * ld] proposals by work item
    double* aux_buf;       // host-callback path: [log_corr (n_local) | ll_prop (n_local)] by work item (two dense arrays: the ln-likes arrive as ONE dense copy,
                           // host -> device or device -> device; the snooker correction never leaves the device)
    int32_t* ids_buf;      // host-callback path: [n_local] global id by work item (-1 = inactive)
#ifdef BPM_TEST_HOOKS      // the per-chain decision trace of the parity tests: test variant only (include/bipymc_hip_test.h: bpm_set_trace); the PRODUCT's
    int32_t* trace_i32;    // argument block has no such fields and its kernels no traced branch (trace_i32_of & co. below are compile-time null there)
    double* trace_f64;     // [n_local * TRACE_F64]   ([n_local * TRACE_I32] above)
    uint8_t* trace_mask;   // [n_local * dim]
#endif
    PermKey pk;
    const uint32_t* perm_tab;   // [N] shuffle order of this generation, position -> chain id (nullptr: evaluate the bijection)
    const uint32_t* inv_tab;    // [N] its inverse, chain id -> position
    const double* gamma_tab;    // [dim + 1] DREAM gamma_base by d' (dream.py:61), host-evaluated
    const uint32_t* plan;       // [N * PLAN_WORDS] by position in shuffle order: chain id, header block, partner ids of
                                // this generation, precomputed by plan_kernel (nullptr: drawn in the update kernel)
    const uint32_t* rec_tab;    // what the update kernel reads its records from: `plan` (item w -> record rec_off + w, rec_off =
    uint32_t rec_off;           // upd_off), or this rank's own run of the owner-sorted records of the half generation (rec_off = 0)
    uint32_t thr[MAX_CR];       // mask thresholds floor(CR_m * 2^16) (dream.py:53,113)
    unsigned long long* stamps;  // diagnostic build only
    double* pack;               // sparse exchange (world > 1): this rank's block [count u32 | pad | ids[cap] | rows[cap][ld]] or nullptr
    uint32_t pack_cap, pack_nsub, pack_stride;   // per sub-block: capacity (rows), count of sub-blocks, doubles per sub-block          // rows the block can take (even)
    double* x_next;             // mode 2 (synchronous DE-MC): new states go here, the state matrix stays frozen
    // replay exchange (world > 1): the owner of a chain publishes ONE BYTE per update -- accepted or not -- and every
    // other rank recomputes the accepted proposals itself (phase_replay_kernel): all inputs of a proposal (the replicated
    // state matrix, counter-addressed draws, update records) are already on every rank
    uint8_t* accbits;           // this rank's [n_local] accept bytes, written by the update kernel (or nullptr)
    const uint8_t* accbits_all; // replay kernel: the gathered [N] bytes, by global chain id
    uint32_t replay;            // 1 in phase_replay_kernel: the chain is remote -- nothing indexed by (chain - lo) may be touched
    // world > 1 with update records SORTED BY OWNER (plan_slot_kernel): a half generation's records are stored rank segment by rank
    // segment, positions in order inside a segment; rank r's k-th update of the half generation is record seg_off[r] + k of
    // `rec_sorted`, and its accept byte is byte k of rank r's block of the gathered bytes (acc_by_item) -- so a replay wavefront
    // finds the byte and the record of "rank r's k-th update" at addresses it can compute: ONE round trip decides whether it works
    const uint32_t* rec_sorted; // the half generation's sorted records (replay kernel), or nullptr
    uint32_t seg_off[MAX_SEG + 1];
    uint32_t n_seg, seg_me;     // number of ranks, this rank
    uint32_t acc_by_item;       // accept bytes indexed by the owner's item number (sorted records) instead of its local chain index
    // push exchange (world > 1): base addresses of the OTHER ranks' exchange buffers (their `L.G`), a device table of MAX_PEERS entries
    // (unused ones repeat the first), and how many are real.  An accepted row -- during CR adaptation also every update's (delta, cr)
    // slots -- is stored at the same offset in each of them with system-scope stores; n_peers == 0: no push
    const unsigned long long* peer_tab;
    uint32_t n_peers;
    uint32_t hist_by_pos;       // 1: the history row and its ln-like are appended at the update's POSITION in this generation's shuffle order
                                // (work items of a wavefront write consecutive rows) instead of at the chain's index; the host remembers the
                                // generation, and rows are put back into chain order when anything reads them (sampler.hip: normalize_history)
                                // 2: the row by position, its ln-like by chain (while DREAM's outlier check is due every few generations: it sums
                                // the ln-like history of every chain, see outlier_omega_kernel)
#ifndef BPM_LEAN_LAST
    uint32_t lean;              // 1 (several chains per wavefront, device target, many chains): ln_like of the current state is re-evaluated from the own
                                // row instead of read from `ll`, `ll` is not written, accepted updates are counted per wavefront (kernels.h: lean_scalars)
#endif
    uint32_t wt;                // 1: the dispatch packet of this launch carries NO release fence -- what a later kernel reads (accepted state
                                // rows, ln-like cache, accept counters) leaves through agent-scope (write-through) stores (store_row_wt)
    uint64_t seed;
    uint64_t t;            // absolute generation
    uint32_t k;            // generation within this run_mcmc call (demc.py:78)
    uint32_t N, lo;
    uint32_t upd_off, n_upd, pool_off, M;   // position ranges in shuffle order
    uint32_t mode;         // 0: work item = position (world == 1); 1: work item = local chain filtered by its group;
                           // 2: synchronous DE-MC (samplers.py:261-308): every local chain, pool = all OTHER chains
    uint32_t n_items;
    uint32_t algo;
    uint32_t P, n_cr;
    double* cr_part1;      // != nullptr: this launch sums its updates' CR statistics itself, chunk by chunk of cr_g1 positions (HOT 3 / 4 only)
    // consumer-side fold (round 5; HOT 3 / 4): != nullptr -- the PREVIOUS generation's level-1 partial sums have not been folded
    // into the totals yet: wavefront 0 of every workgroup of this launch folds them itself (cr_fold: cr_final_kernel's own code, same bits), the workgroup
    // takes p_cr from that, workgroup 0 stores the new totals into cr_fold_out (another block than cr_fold_tot: the other workgroups still read that one)
    const double* cr_fold_part;
    const double* cr_fold_tot;
    double* cr_fold_out;
    uint32_t cr_fold_nb;   // partial sums per array (= their stride): the generation's level-1 sums, or what cr_mid_kernel passes left of them; <= CR_FINAL_MAX
    uint32_t cr_chunk0, cr_n1;   // first chunk of this half generation, chunks per generation (the stride of the m-major partial arrays)
    uint32_t adapt_on;     // dream.py:92  burnin_gen > k
    uint32_t cr_gate;      // dream.py:123 history length > n_cr_gen
    uint32_t hist_len;     // rows of every chain's history before this generation
    double gamma_scale, gamma_demc, epsilon, u_epsilon, p_snooker;
#ifdef BPM_LEAN_LAST
    uint32_t lean, pad_lean;
#endif
};

// The trace pointers of an argument block: fields of the test variant only (-DBPM_TEST_HOOKS); in the product library these are compile-time null
// pointers, so every `if (trace_i32_of(a))` branch folds away and the argument block carries nothing of the test surface.
#ifdef BPM_TEST_HOOKS
__host__ __device__ __forceinline__ int32_t* trace_i32_of(const PhaseArgs& a) { return a.trace_i32; }
__host__ __device__ __forceinline__ double* trace_f64_of(const PhaseArgs& a) { return a.trace_f64; }
__host__ __device__ __forceinline__ uint8_t* trace_mask_of(const PhaseArgs& a) { return a.trace_mask; }
__host__ __device__ __forceinline__ void trace_set(PhaseArgs& a, int32_t* i, double* f, uint8_t* m) { a.trace_i32 = i; a.trace_f64 = f; a.trace_mask = m; }
#else
__host__ __device__ constexpr int32_t* trace_i32_of(const PhaseArgs&) { return nullptr; }
__host__ __device__ constexpr double* trace_f64_of(const PhaseArgs&) { return nullptr; }
__host__ __device__ constexpr uint8_t* trace_mask_of(const PhaseArgs&) { return nullptr; }
__host__ __device__ __forceinline__ void trace_set(PhaseArgs&, int32_t*, double*, uint8_t*) {}
#endif

// Sum over the LPC lanes of a chain subgroup, result in every lane of the subgroup.
// Cross-lane moves are DPP (no LDS traffic, ~8 cycles each instead of a ds_bpermute round trip):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror give the sum of each row of
// 16 lanes; a full wavefront then combines its four row sums through v_readlane (scalar operands).
// Must be called with all 64 lanes active (it is: subgroups never diverge around a reduction).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int LPC>
__device__ __forceinline__ double gsum(double v) {
    static_assert(LPC == 1 || LPC == 4 || LPC == 16 || LPC == 32 || LPC == 64, "subgroup sizes");
    if (LPC >= 4) {
        v += dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    }
    if (LPC >= 16) {
        v += dpp_f64<0x141>(v);    // row_half_mirror
        v += dpp_f64<0x140>(v);    // row_mirror
    }
    if (LPC == 32) v += __shfl_xor(v, 16);
    if (LPC == 64) v = (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
    return v;
}
template <int LPC>
__device__ __forceinline__ int gsum_i(int v) {
    if (LPC >= 4) {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    }
    if (LPC >= 16) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    }
    if (LPC == 32) v += __shfl_xor(v, 16);
    if (LPC == 64)
        v = (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
            (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
    return v;
}

// ---- CR reduction (round 4) --------------------------------------------------------------------------------------------------
// dream.py:119-140 once per generation from the (delta, cr) slots of ALL N chains.  Rounds 2-3 spent two dependent dispatches on it behind
// the two update kernels (cr_partial_kernel over the slots in chain order + cr_final_kernel: 2.9 + 3.0 us of cfg2's 22 us burn-in
// generation, each at the floor of a dependent launch).  Now the first level is done where the statistics are born:
//   level 1  sums over CHUNKS OF cr_g1 CONSECUTIVE POSITIONS of the generation's shuffle order, each half generation chunked by itself,
//            positions in order (for every CR value m: delta_m += delta, n_m += 1 of the chunk's updates that drew m).  On a single GPU the
//            burn-in flavours of the update kernel (HOT 3 / 4) write them: a workgroup of 16 wavefronts = 16 chains through LDS (one wavefront
//            per chain), or a wavefront by itself (several chains per wavefront).  Everywhere else -- a rank of a world (it updates only its own
//            chains; the slots of all chains are in its replica), the general kernel, the looped wide-row kernel, the host-callback path --
//            cr_level1_kernel computes THE SAME sums from the slots (one thread per chunk, positions through the shuffle table): same chunks,
//            same order, same bits, whatever the number of GPUs and the launch path (tested: p_cr of worlds == single rank, nohot == default).
//   level 2  (only beyond 512 chunks: N > 8192 at 16 positions per chunk) cr_mid_kernel: one wavefront per 64 consecutive chunks, a fixed DPP tree.
//   final    cr_final_kernel: one wavefront; lane l adds partials l, l + 64, ... in order (all loads in flight at once), a fixed DPP tree over the
//            lanes, then dream.py:132-140.  (Folding in the last workgroup of the generation's last launch instead -- an agent-scope ticket, no
//            dispatch at all -- was built and measured SLOWER: 21.1 vs 20.7 us per generation at cfg2, the sign round 3 found for its ticket form.)
// cfg2 burn-in: 22.0 -> 18.7 us per generation in the timing-only build that left cr_partial_kernel out (profiles/r04_burnin.txt).
constexpr uint32_t CR_FINAL_MAX = 512;        // partial sums cr_final_kernel folds by itself (8 per lane)
__host__ __device__ inline uint32_t cr_chunks_of(uint32_t n_half, uint32_t g1) { return (n_half + g1 - 1u) / g1; }
struct CrTotals { double p[MAX_CR], d[MAX_CR], n[MAX_CR]; };
// the end of a fold: T = old totals with this generation's sums already added where some update contributed (any) -> p_cr (dream.py:132-140)
__device__ __forceinline__ void cr_finalize(bool any, uint32_t n_cr, CrTotals& T) {
    if (!any) return;
    uint32_t nz = 0;
#pragma unroll
    for (int m = 0; m < MAX_CR; ++m) nz += (m < (int)n_cr && T.n[m] != 0.0) ? 1u : 0u;
    if (nz == n_cr) {
#pragma unroll
        for (int m = 0; m < MAX_CR; ++m) if (m < (int)n_cr) T.p[m] = T.d[m] / T.n[m];     // dream.py:134-137
    }
    double sum = 0.0;
#pragma unroll
    for (int m = 0; m < MAX_CR; ++m) if (m < (int)n_cr) sum += T.p[m];
#pragma unroll
    for (int m = 0; m < MAX_CR; ++m) if (m < (int)n_cr) T.p[m] = T.p[m] / sum;            // dream.py:140
}
// ONE array of partial sums (the delta or the count sums of one CR value) the way cr_fold adds it: lane l holds partials l, l + 64, ... (plain loads:
// see PLAIN above), adds them in order, the fixed DPP tree over the lanes.  A workgroup that folds with a wavefront per array (phase_fused_kernel) gets
// cr_fold's bits.
template <int ROUNDS>
__device__ __forceinline__ double cr_array_sum(const double* arr, uint32_t nb) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    double v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {      // (every lane loads, index clamped, the value selected afterwards: a predicated load is guarded by s_waitcnt vmcnt(0))
        const uint32_t b = (uint32_t)r * WAVE + lane;
        const double x = arr[b < nb ? b : nb - 1u];
        v[r] = b < nb ? x : 0.0;
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) s += v[r];
    return gsum<WAVE>(s);
}
// totals T = (p_cr | delta_m | n_cr_updates) + one generation's partial sums part[m * stride + b], b < nb <= 64 ROUNDS (delta) and
// part[(MAX_CR + m) * stride + b] (counts), by ONE wavefront: lane l holds partials l, l + 64, ... -- ALL loads of the kernel issued before the first
// add (a loop with a round trip per 64 partials cost 5 us at cfg2; a workgroup of 16 wavefronts with an LDS hand-over 4.7) -- adds them in order, the
// lanes' sums meet in the fixed DPP tree.  (A partial beyond nb reads as +0.0: x + 0.0 == x, the result is a function of nb alone.)  Nothing changes
// when no update contributed; p_cr is re-estimated once every CR value has been used, then normalised.  All 64 lanes take part.
// PLAIN: ordinary loads of the partial sums -- for a caller whose packet acquired (an update kernel: its L2 starts clean, the sums were stored through by
// an EARLIER launch) and whose 256 workgroups read the same 24 KB: the agent-scope form misses the L2 every time
template <int ROUNDS, bool PLAIN = false>
__device__ __forceinline__ void cr_fold(const double* tot, const double* part, uint32_t nb, uint32_t stride, uint32_t n_cr, CrTotals& T) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int m = 0; m < MAX_CR; ++m) { T.p[m] = tot[m]; T.d[m] = tot[MAX_CR + m]; T.n[m] = tot[2 * MAX_CR + m]; }
    bool any = false;
#pragma unroll
    for (int m0 = 0; m0 < MAX_CR; m0 += 4) {                 // four CR values at a time: 8 ROUNDS doubles in registers
        if (m0 < (int)n_cr) {                                // uniform
            double vd[4][ROUNDS], vn[4][ROUNDS];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) {
                    const uint32_t b = (uint32_t)r * WAVE + lane;
                    vd[mm][r] = 0.0; vn[mm][r] = 0.0;
                    if (m0 + mm < (int)n_cr && b < nb) {     // (agent-scope loads: written by kernels whose packets may carry no release fence)
                        if (PLAIN) {
                            vd[mm][r] = part[(uint64_t)(m0 + mm) * stride + b];
                            vn[mm][r] = part[(uint64_t)(MAX_CR + m0 + mm) * stride + b];
                        } else {
                            vd[mm][r] = __hip_atomic_load(&part[(uint64_t)(m0 + mm) * stride + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            vn[mm][r] = __hip_atomic_load(&part[(uint64_t)(MAX_CR + m0 + mm) * stride + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                if (m0 + mm < (int)n_cr) {                   // uniform
                    double sd = 0.0, sn = 0.0;
#pragma unroll
                    for (int r = 0; r < ROUNDS; ++r) { sd += vd[mm][r]; sn += vn[mm][r]; }
                    const double td = gsum<WAVE>(sd), tn = gsum<WAVE>(sn);
                    if (tn > 0.0) { any = true; T.n[m0 + mm] += tn; T.d[m0 + mm] += td; }
                }
            }
        }
    }
    cr_finalize(any, n_cr, T);
}
__device__ __forceinline__ void cr_write_totals(const CrTotals& T, uint32_t n_cr, double* out) {
#pragma unroll
    for (int m = 0; m < MAX_CR; ++m)
        if (m < (int)n_cr) {      // (agent-scope stores: the packet of a CR reduction kernel carries no release fence on the library's own queue)
            __hip_atomic_store(&out[m], T.p[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&out[MAX_CR + m], T.d[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&out[2 * MAX_CR + m], T.n[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}
// One chunk's sums from values in order: acc_d / acc_n of CR value m (the lane's or the thread's own m), element k contributes when idx == m.
// THE definition of a level-1 partial: the update kernels and cr_level1_kernel both add through this function, in position order.
__device__ __forceinline__ void cr_chunk_add(double& acc_d, double& acc_n, int m, int idx, double delta) {
    if (idx == m) { acc_d += delta; acc_n += 1.0; }
}

__device__ __forceinline__ double box_muller(uint32_t w1, uint32_t w2) {
    const double u1 = ((double)w1 + 1.0) * 2.3283064365386963e-10;
    const double u2 = u01_32(w2);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// ---------------------------------------------------------------------------------
// Targets: ln_like in registers.  `v[DPL]` holds the lane's coordinates, pair chunk u
// = coordinates (2*pi, 2*pi+1), pi = q + u*LPC.
// ---------------------------------------------------------------------------------
template <int TARGET, int LPC, int DPL>
struct Target;
// Every target splits into load() -- fetch its constants, called first thing in the kernel so the misses
// overlap with the draw arithmetic -- and eval() on registers only.

// utils/d100_gauss.py:14-35, equicorrelated Gaussian in O(d):
// ll = c0 - 0.5 (a S2 - b S1^2), z = y / sigma.  params [rho, c0, a, b, 1/sigma...]
template <int LPC, int DPL>
struct Target<TARGET_GAUSS, LPC, DPL> {
    struct Consts { double is[DPL]; double c0, a, b; };
    static __device__ __forceinline__ Consts load(int q, uint32_t dim, const double* tp) {
        Consts k;
        k.c0 = tp[1]; k.a = tp[2]; k.b = tp[3];
#pragma unroll
        for (int s = 0; s < DPL; ++s) {
            const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (s & 1);
            k.is[s] = j < dim ? tp[4 + j] : 0.0;
        }
        return k;
    }
    static __device__ __forceinline__ double eval(const double* v, int q, uint32_t dim, const Consts& k) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int s = 0; s < DPL; ++s) {
            const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (s & 1);
            if (j < dim) {
                const double z = v[s] * k.is[s];
                s1 += z;
                s2 += z * z;
            }
        }
        s1 = gsum<LPC>(s1);
        s2 = gsum<LPC>(s2);
        return k.c0 - 0.5 * (k.a * s2 - k.b * s1 * s1);
    }
};

// utils/dblgauss_rv.py:11-32 and its pairwise-block extension to even d:
// params [lw1, lw2, (mx, my, 1/sx, 1/sy, rho, 1/(1-rho^2), ln_norm) x 2]
template <int LPC, int DPL>
struct Target<TARGET_MIXTURE, LPC, DPL> {
    struct Consts { double p[16]; };
    // (the sixteen values are the same for every lane, yet they arrive through the vector memory path and sit in 32 VGPRs for the whole kernel: 98 VGPRs,
    // 4 wavefronts per SIMD at cfg5.  Round 4 read them through the constant address space instead -- scalar loads, 67 VGPRs, 7 wavefronts per SIMD --
    // and every small-d workload got SLOWER (cfg5 45.2 -> 46.7, its share 9.4 -> 10.5 us per generation): the kernel is not short of wavefronts, and
    // the SGPR file was full already.  DESIGN.md section 5.)
    static __device__ __forceinline__ Consts load(int, uint32_t, const double* tp) {
        Consts k;
#pragma unroll
        for (int i = 0; i < 16; ++i) k.p[i] = tp[i];
        return k;
    }
    static __device__ __forceinline__ double eval(const double* v, int q, uint32_t dim, const Consts& k) {
        double q0 = 0.0, q1 = 0.0;
#pragma unroll
        for (int u = 0; u < DPL / 2; ++u) {
            const uint32_t j = 2u * (uint32_t)(q + u * LPC);
            if (j + 1 < dim) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double* p = k.p + 2 + 7 * c;
                    const double a = (v[2 * u] - p[0]) * p[2];
                    const double b = (v[2 * u + 1] - p[1]) * p[3];
                    const double qq = (a * a - 2.0 * p[4] * a * b + b * b) * p[5];
                    if (c == 0) q0 += qq; else q1 += qq;
                }
            }
        }
        q0 = gsum<LPC>(q0);
        q1 = gsum<LPC>(q1);
        const double npairs = (double)(dim / 2);
        const double c0 = k.p[0] + npairs * k.p[8] - 0.5 * q0;
        const double c1 = k.p[1] + npairs * k.p[15] - 0.5 * q1;
        const double m = fmax(c0, c1);
        return m + log(exp(c0 - m) + exp(c1 - m));
    }
};

// utils/banana_rv.py:26-37; params [mu1, mu2, 1/s1, 1/s2, rho, 1/(1-rho^2), ln_norm, a, b]; d = 2, LPC = 1
template <int LPC, int DPL>
struct Target<TARGET_BANANA, LPC, DPL> {
    struct Consts { double p[9]; };
    static __device__ __forceinline__ Consts load(int, uint32_t, const double* tp) {
        Consts k;
#pragma unroll
        for (int i = 0; i < 9; ++i) k.p[i] = tp[i];
        return k;
    }
    static __device__ __forceinline__ double eval(const double* v, int, uint32_t, const Consts& k) {
        const double a = k.p[7], b = k.p[8];
        const double x1 = v[0] / a;
        const double x2 = (v[1] - b * (x1 * x1 + a * a)) * a;
        const double u = (x1 - k.p[0]) * k.p[2];
        const double w = (x2 - k.p[1]) * k.p[3];
        return k.p[6] - 0.5 * (u * u - 2.0 * k.p[4] * u * w + w * w) * k.p[5];
    }
};

// ---------------------------------------------------------------------------------
template <int DPL>
struct Work {
    double x[DPL];     // current state (lane's coordinates)
    double p[DPL];     // proposal
    double delta;      // CR statistic (dream.py:130)
    double log_corr;   // snooker Jacobian term
    double gamma;
    uint32_t acc_hi, acc_lo;   // words of the accept uniform
    uint32_t item;             // work-item number of this update (index of its accept byte when acc_by_item)
    uint32_t pos_own;          // its position in this generation's shuffle order (mode 0)
    double ll_cur;             // cached ln_like of the current state, fetched early
    uint32_t acc_prev;         // this chain's accept counter, fetched early
    double w_mean[DPL], w_m2[DPL];   // burn-in only: Welford moments of this chain's own history, fetched early
    int cr_idx, d_prime, jump, snk;
    uint32_t maskbits;
};

typedef double bpm_d2v __attribute__((ext_vector_type(2)));
template <int LPC, int DPL>
__device__ __forceinline__ void load_row(const double* row, int q, uint32_t ld, double* v) {
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        double2 t = make_double2(0.0, 0.0);
        if (2 * pi < ld) t = reinterpret_cast<const double2*>(row)[pi];
        v[2 * u] = t.x;
        v[2 * u + 1] = t.y;
    }
}
template <int LPC, int DPL>
__device__ __forceinline__ void store_row(double* row, int q, uint32_t ld, const double* v) {
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        if (2 * pi < ld) reinterpret_cast<double2*>(row)[pi] = make_double2(v[2 * u], v[2 * u + 1]);
    }
}

// Release-less launches (the library's own queue in the steady state, DESIGN.md section 5 "Packet fences"): an ordinary store leaves its
// line dirty in one XCD's L2 until a release fence writes it back; without that fence the next kernel, on another XCD, would read the old
// row from memory.  A relaxed agent-scope atomic store goes through to the memory side -- the memory model's own guarantee -- so the
// state is visible to whatever runs next without any write-back of the caches.  (Two 8-byte stores per lane instead of one 16-byte
// store: 3 % on the launch period when the packets do carry a release, which is why this path is chosen per launch.)
template <int LPC, int DPL>
__device__ __forceinline__ void store_row_wt(double* row, int q, uint32_t ld, const double* v) {
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        if (2 * pi < ld) {
            __hip_atomic_store(row + 2 * pi, v[2 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(row + 2 * pi + 1, v[2 * u + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The same through ONE 16-byte store per lane: `global_store_dwordx4 ... sc1` is what the compiler emits for an agent-scope store of
// up to 8 bytes; HIP has no 16-byte atomic store, hence the instruction by hand (gfx942 / gfx950 cache-policy bits: sc0 sc1 nt).
template <int LPC, int DPL>
__device__ __forceinline__ void store_row_wt16(double* row, int q, uint32_t ld, const double* v) {
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        if (2 * pi < ld) {
            bpm_d2v t = {v[2 * u], v[2 * u + 1]};
            bpm_d2v* p = reinterpret_cast<bpm_d2v*>(row) + pi;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
        }
    }
}

// Push exchange: a row into ANOTHER rank's replica (peer device memory over xGMI, or another process's buffer on the same GPU):
// relaxed system-scope stores -- written through to the memory that owns the line, visible to the peer's next kernel once this
// kernel's completion has been announced (push_sync_kernel) and that kernel's packet has acquired.
template <int LPC, int DPL>
__device__ __forceinline__ void store_row_sys(double* row, int q, uint32_t ld, const double* v) {
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        if (2 * pi < ld) {
            __hip_atomic_store(row + 2 * pi, v[2 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(row + 2 * pi + 1, v[2 * u + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Streaming (non-temporal) row store for what the update kernels write and do not read back in the same launch:
// history rows and the Welford moments.  An ordinary store leaves the line dirty in the XCD's L2 until the
// end-of-kernel release writes everything back -- with 3.3 MB of history per launch that flush sat on the critical
// path between the two half generations: 14.4 vs 15.7 us/generation at cfg2, 72.0 vs 76.2 at N=65536.  (Agent- or
// system-scope stores measured slower, 16.5; the accepted state rows gain nothing, they are few.)
template <int LPC, int DPL>
__device__ __forceinline__ void store_row_stream(double* row, int q, uint32_t ld, const double* v) {
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        if (2 * pi < ld) {
            bpm_d2v t = {v[2 * u], v[2 * u + 1]};
            __builtin_nontemporal_store(t, reinterpret_cast<bpm_d2v*>(row) + pi);
        }
    }
}

// Two standard normals from two words in float32 on the hardware transcendental units
// (v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32; the trig units take revolutions, so
// cos(2 pi u2) needs no multiply and no range reduction).  Used for the per-generation jitter
// e_n ~ N(0, eps^2), eps ~ 1e-12: float32 resolution is ~1e-19 absolute, below the state's ulp.
__device__ __forceinline__ void box_muller_pair_f32(uint32_t w1, uint32_t w2, double& n0, double& n1) {
    const float u1 = ((float)w1 + 1.0f) * 2.3283064365386963e-10f;
    const float u2 = (float)w2 * 2.3283064365386963e-10f;
    const float r = __builtin_amdgcn_sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(u1));
    n0 = (double)(r * __builtin_amdgcn_cosf(u2));
    n1 = (double)(r * __builtin_amdgcn_sinf(u2));
}

// Lane K of every quad (4 consecutive lanes) to all four lanes of that quad: one DPP move, no LDS.
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, false);
}

// Partner chains of chain c (pool positions -> global ids through the generation's bijection): in registers when
// the pair count is a compile-time constant and a chain is one lane or one wavefront (make_proposal: FAST / RL /
// planned paths), otherwise resolved by the lanes of the subgroup in parallel and handed over through LDS.
template <int LPC, int NPART_CT>
struct Partners {
    uint32_t r[NPART_CT > 0 ? NPART_CT : 1];
    const uint32_t* lds;
    __device__ __forceinline__ uint32_t get(int i) const { return NPART_CT > 0 ? r[i] : lds[i]; }
};

__device__ __forceinline__ uint32_t partner_pos(const PhaseArgs& a, uint32_t c, uint32_t idx, uint32_t n_pair_idx) {
    if (idx < n_pair_idx) {
        const uint32_t p = idx >> 1;
        const u32x4 wb = chain_block(a.seed, c, a.t, SLOT_PAIR0 + (p >> 1));
        uint32_t ia, ib;
        distinct_pair((p & 1) ? wb.z : wb.x, (p & 1) ? wb.w : wb.y, a.M, ia, ib);
        uint32_t pos = (idx & 1) ? ib : ia;
        if (a.mode == 2) pos += (pos >= c) ? 1u : 0u;      // np.delete(range(N), i) (samplers.py:274): skip the chain itself
        return pos;
    }
    const u32x4 ws = chain_block(a.seed, c, a.t, SLOT_SNK);
    uint32_t iz, i1, i2;
    distinct_three(ws.x, ws.y, ws.z, a.M, iz, i1, i2);
    const uint32_t s = idx - n_pair_idx;
    return s == 0 ? iz : (s == 1 ? i1 : i2);
}

__device__ __forceinline__ uint32_t pos_to_chain(const PhaseArgs& a, uint32_t pos) {
#if defined(BPM_TEST_HOOKS) && defined(BPM_FAKE_NO_PARTNER_HOP)      // timing experiment only (wrong results): what the position -> chain id lookup of a partner costs
    return pos;                     // (profiles/r03_small_d_hops_and_position_order.txt)
#else
    return a.perm_tab ? a.perm_tab[pos] : perm_fwd(pos, a.pk);
#endif
}
__device__ __forceinline__ uint32_t own_pos_to_chain(const PhaseArgs& a, uint32_t pos) {
#if defined(BPM_TEST_HOOKS) && defined(BPM_FAKE_NO_OWN_HOP)          // timing experiment only (wrong results): the work item's own position -> chain id lookup
    return pos;
#else
    return a.perm_tab ? a.perm_tab[pos] : perm_fwd(pos, a.pk);
#endif
}
__device__ __forceinline__ uint32_t chain_to_pos(const PhaseArgs& a, uint32_t c) {
    return a.inv_tab ? a.inv_tab[c] : perm_inv(c, a.pk);
}

// Build the proposal of chain c (dream.py:43-93 / demc.py:161-182).
// ALGO compile-time; NP = compile-time number of pairs (0: runtime a.P, DREAM only).
struct NoEarly { template <class W> __device__ __forceinline__ void operator()(W&) const {} };
// p_cr handed in by the caller (a workgroup that folded the CR totals itself, PhaseArgs::cr_fold_part) instead of read from a.cr_state -- by value: registers
struct PcrGiven { bool on = false; double p[MAX_CR]; };
// EARLY: work on the own row (wk.x) that the caller wants done while the partner rows are still on their way -- the re-evaluation of the
// current state's ln_like (lean_scalars): behind the proposal it sat on the critical path of the latency-bound launches (cfg5's share:
// 10.05 instead of 9.3 us per generation), here it runs in the shadow of the partner fetches.
template <int ALGO, int LPC, int DPL, int NP, int LOAD_LL = 1 /* 1 always, 0 never, 2 unless a.lean */, class EARLY = NoEarly>
__device__ __forceinline__ void make_proposal(const PhaseArgs& a, uint32_t c, bool active, int q, int cw,
                                              uint32_t* s_part, Work<DPL>& wk, const uint32_t* rec = nullptr,
                                              unsigned long long* bpm_stamp = nullptr, EARLY early = EARLY(), const PcrGiven pcr_ovr = PcrGiven()) {
    constexpr bool DREAM = ALGO == ALGO_DREAM;
    // FAST: partner ids straight into registers, every lane for itself, when a lane IS a chain (LPC == 1: no other
    // lane to share the work with, so the LDS hand-over loop would only re-evaluate the same Philox block once per
    // partner).  (Resolving them on the scalar unit for one wavefront per chain measured slower than the lane-parallel
    // RL path below: 25.9 vs 19.8 us/generation at cfg2.)
    constexpr bool FAST = (NP > 0) && (LPC == 1);
    constexpr bool SNK_DIRECT = (LPC == 1);
    const uint32_t dim = a.L.dim, ld = a.L.ld;
    const uint32_t P = NP > 0 ? (uint32_t)NP : a.P;
    // state-dependent loads first: they overlap with all the draw arithmetic below
    load_row<LPC, DPL>(row_ptr(a.L, c), q, ld, wk.x);      // (a non-temporal load of this read-once row measured no faster)
    wk.ll_cur = 0.0;
    wk.acc_prev = 0u;
    if (!a.replay) {
#if defined(BPM_TEST_HOOKS) && defined(BPM_FAKE_NO_LL)      // timing experiment only (wrong results): what the chain-indexed ln-like cache and accept counter cost
        if (LPC < WAVE) wk.ll_cur = wk.x[0] * 1e-300; else
#endif
        if (LOAD_LL == 1 || (LOAD_LL == 2 && !a.lean)) wk.ll_cur = a.ll[c - a.lo];   // (else: the caller re-evaluates it from wk.x -- lean_scalars)
        // the accept counter: one wavefront per chain reads it here (a scalar load, early) and stores + 1 on accept; with several chains per
        // wavefront the read is a scattered 4-byte load per update -- there an accepted update bumps it with a device-scope atomic add without
        // return instead (one writer per address and launch: nothing serialises): cfg3 12.7 -> 11.9 us per generation, cfg5 52.2 -> 48.2
        if (LPC == WAVE) wk.acc_prev = a.acc_count[c - a.lo];
    }
    if (DREAM && a.adapt_on) {                 // dream.py:128: requested here, used after the proposal and after the accept test
        // (as streaming loads -- the moments are read once per launch -- burn-in got slower: cfg5 68.6 -> 75.4 us per generation, cfg2 20.7 -> 21.3)
        load_row<LPC, DPL>(a.w_mean + (uint32_t)((c - a.lo) * 2u * ld), q, ld, wk.w_mean);
        load_row<LPC, DPL>(a.w_m2 + (uint32_t)((c - a.lo) * 2u * ld), q, ld, wk.w_m2);
    }
    double pcr[MAX_CR];                        // p_cr (uniform pointer: one scalar load of the whole block)
    if (DREAM) {
#pragma unroll
        for (int m = 0; m < MAX_CR; ++m) pcr[m] = pcr_ovr.on ? pcr_ovr.p[m] : a.cr_state[m];      // (pcr_ovr: the workgroup folded the totals itself, cr_fold_part)
    }
    // gamma table held across the wavefront's lanes: the lookup by d' is then a v_readlane, not a dependent load
    double gtab[DPL];
    if (DREAM && LPC == WAVE) {
#pragma unroll
        for (int u = 0; u < DPL; ++u) {
            // every lane loads (index clamped to the table's last entry), the lanes beyond dim are zeroed by a select on the
            // loaded value.  As a predicated load (`cond ? tab[i] : 0.0`) the zero of those lanes is a second write to the
            // load's destination register, which the compiler guards with s_waitcnt vmcnt(0): every wavefront then sat
            // out its own-row fetch before it could even request the partner rows.
            const uint32_t gi = (uint32_t)(q + u * WAVE);
            const double gl = a.gamma_tab[gi <= dim ? gi : dim];
            gtab[u] = gi <= dim ? gl : 0.0;
        }
    }
    // ... and with several chains per wavefront (at most 32 coordinates): lane l holds entry l, the lookup by d' is a lane read through the LDS crossbar
    // (ds_bpermute) instead of a dependent load behind the mask count -- the first touch of the table on a CU is an L2 round trip on the critical path
    double gt_lane = 0.0;
    if (DREAM && LPC * DPL < WAVE) {
        const uint32_t wl = (uint32_t)threadIdx.x & (uint32_t)(WAVE - 1);
        gt_lane = a.gamma_tab[wl <= dim ? wl : dim];
    }
    const bool snk_possible = !DREAM && a.p_snooker > 0.0 && a.M >= 3;
    const uint32_t npart = 2 * P + (snk_possible ? 3u : 0u);
    // RL: one wavefront per chain and a compile-time pair count: the partner ids are resolved by lanes in
    // parallel and handed over as scalars through v_readlane: no LDS, no barrier, scalar row addresses.
    constexpr bool RL = (LPC == WAVE) && (NP > 0) && !FAST;
    // MERGED (RL, one coordinate pair per lane, enough idle lanes): ONE Philox evaluation per lane serves
    // everything -- lanes [0, npairs) draw their coordinate pair's block, lanes [npairs, npairs + 2NP) the
    // pair-selection blocks, lane npairs + 2NP the chain's header block -- instead of three separate
    // evaluations (one of them a 110-instruction scalar one) per wavefront.
    const uint32_t npairs = (dim + 1u) >> 1;
    const bool merged = RL && DPL == 2 && (npairs + 2u * (uint32_t)NP + 1u <= (uint32_t)WAVE);
    u32x4 wpair[DPL / 2];
    u32x4 h0;
    // QUAD (DREAM, 4 lanes per chain, three pairs, d <= 8): the chain's seven Philox blocks -- four coordinate pairs,
    // header, two pair-selection blocks -- in TWO evaluations per lane instead of four (header once per lane and every
    // partner index on its own lane of the LDS hand-over loop), shared inside the quad by DPP quad_perm broadcasts
    constexpr bool QUAD = DREAM && LPC == 4 && DPL == 2 && NP == 3;
    constexpr bool REGS = FAST || RL || QUAD;             // partner ids live in registers
    Partners<LPC, REGS ? 2 * NP : 0> part;
    part.lds = s_part + cw * MAX_PARTNERS;
    // PLANNED (record of this update precomputed by plan_kernel): header words and partner ids are loads issued at
    // kernel entry (scalar ones when a wavefront is one chain), so the partner rows are requested BEFORE the lanes'
    // own Philox evaluation instead of after it (draw -> table lookup -> row fetch was a serial chain of ~3 k
    // cycles), and the header / pair / snooker blocks are not evaluated here at all.
    // (one wavefront per chain only.  With 4 lanes per chain the records measured no gain, cfg5/8 13.8 vs 13.9 us/generation;
    // with one lane per chain a loss -- cfg3 14.6 vs 12.3 with 64-byte records per lane, 13.8 vs 12.6 from a
    // structure-of-arrays table whose loads coalesce: at 65536 chains plan_kernel itself costs 1.3 us per generation)
    const bool planned = (LPC == WAVE) && rec != nullptr;
    // the coordinate-pair blocks follow the row requests when a wavefront is one chain with one pair per lane
    const bool dims_late = planned && RL && DPL == 2;
    if (planned) {
        h0.x = rec[1]; h0.y = rec[2]; h0.z = rec[3]; h0.w = rec[4];
        if (FAST || RL) {
#pragma unroll
            for (int i = 0; i < 2 * NP; ++i) part.r[i] = rec[5 + i];
        }
        if (!dims_late) {
#pragma unroll
            for (int u = 0; u < DPL / 2; ++u) wpair[u] = chain_block(a.seed, c, a.t, SLOT_DIM0 + (uint32_t)(q + u * LPC));
        }
    } else if (merged) {
        const uint32_t uq = (uint32_t)q, hdr_lane = npairs + 2u * (uint32_t)NP;
        const uint32_t pidx = uq - npairs;                       // partner index for lanes [npairs, hdr_lane)
        const uint32_t slot = uq < npairs ? SLOT_DIM0 + uq : (uq < hdr_lane ? SLOT_PAIR0 + (pidx >> 2) : SLOT_HDR0);
        const u32x4 wb = chain_block(a.seed, c, a.t, slot);
        wpair[0] = wb;
        h0.x = (uint32_t)__builtin_amdgcn_readlane((int)wb.x, hdr_lane);
        h0.y = (uint32_t)__builtin_amdgcn_readlane((int)wb.y, hdr_lane);
        h0.z = (uint32_t)__builtin_amdgcn_readlane((int)wb.z, hdr_lane);
        h0.w = (uint32_t)__builtin_amdgcn_readlane((int)wb.w, hdr_lane);
        BPM_STAMP(2);
        uint32_t mine = 0;
        if (uq >= npairs && uq < hdr_lane) {
            const uint32_t p = pidx >> 1;
            uint32_t ia, ib;
            distinct_pair((p & 1) ? wb.z : wb.x, (p & 1) ? wb.w : wb.y, a.M, ia, ib);
            uint32_t pos = (pidx & 1) ? ib : ia;
            if (a.mode == 2) pos += (pos >= c) ? 1u : 0u;      // synchronous DE-MC: pool = all other chains
            mine = pos_to_chain(a, a.pool_off + pos);
        }
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i) part.r[i] = (uint32_t)__builtin_amdgcn_readlane((int)mine, npairs + i);
    } else if (QUAD) {
        wpair[0] = chain_block(a.seed, c, a.t, SLOT_DIM0 + (uint32_t)q);
        // second evaluation: lane 0 (and 3) the header block, lanes 1 and 2 the two pair-selection blocks
        const u32x4 wb = chain_block(a.seed, c, a.t, q == 1 ? SLOT_PAIR0 : (q == 2 ? SLOT_PAIR0 + 1u : SLOT_HDR0));
        h0.x = quad_bcast<0>(wb.x); h0.y = quad_bcast<0>(wb.y); h0.z = quad_bcast<0>(wb.z); h0.w = quad_bcast<0>(wb.w);
        uint32_t i0 = 0, i1 = 0, i2 = 0, i3 = 0;
        if (q == 1 || q == 2) {                                    // lane 1: pairs 0 and 1, lane 2: pair 2 (partner_pos's layout)
            uint32_t ia, ib;
            distinct_pair(wb.x, wb.y, a.M, ia, ib);
            i0 = pos_to_chain(a, a.pool_off + ia); i1 = pos_to_chain(a, a.pool_off + ib);
            if (q == 1) {
                distinct_pair(wb.z, wb.w, a.M, ia, ib);
                i2 = pos_to_chain(a, a.pool_off + ia); i3 = pos_to_chain(a, a.pool_off + ib);
            }
        }
        part.r[0] = quad_bcast<1>(i0); part.r[1] = quad_bcast<1>(i1); part.r[2] = quad_bcast<1>(i2); part.r[3] = quad_bcast<1>(i3);
        part.r[4] = quad_bcast<2>(i0); part.r[5] = quad_bcast<2>(i1);
    } else {
        // one header block: (select16|gamma16, forced dim [DREAM] / snooker gamma [DE-MC], accept hi, accept lo)
        h0 = chain_block(a.seed, c, a.t, SLOT_HDR0);
#pragma unroll
        for (int u = 0; u < DPL / 2; ++u) wpair[u] = chain_block(a.seed, c, a.t, SLOT_DIM0 + (uint32_t)(q + u * LPC));
        if (FAST) {
#pragma unroll
            for (int i = 0; i < 2 * NP; ++i) part.r[i] = pos_to_chain(a, a.pool_off + partner_pos(a, c, (uint32_t)i, 2u * NP));
        }
        if (RL) {
            uint32_t mine = 0;
            if (q < 2 * NP) mine = pos_to_chain(a, a.pool_off + partner_pos(a, c, (uint32_t)q, 2u * NP));
#pragma unroll
            for (int i = 0; i < 2 * NP; ++i) part.r[i] = (uint32_t)__builtin_amdgcn_readlane((int)mine, i);
        }
    }
    BPM_STAMP(3);
    wk.acc_hi = h0.z; wk.acc_lo = h0.w;
    const double u_sel = (double)(h0.x >> 16) * 1.52587890625e-05;      // CR select (DREAM) / snooker select (DE-MC)
    const double u_gam = (double)(h0.x & 0xFFFFu) * 1.52587890625e-05;  // gamma = 1 jump select
    uint32_t snk_id[3] = {0u, 0u, 0u};
    if (SNK_DIRECT && snk_possible && planned) {
        snk_id[0] = rec[5 + 2 * P]; snk_id[1] = rec[6 + 2 * P]; snk_id[2] = rec[7 + 2 * P];
    } else if (SNK_DIRECT && snk_possible && (u_sel < a.p_snooker || trace_i32_of(a) != nullptr)) {
        // three distinct snooker partners, one Philox block -- only for the lanes whose update IS a snooker update (one in ten at the
        // BASELINE setting): the three table lookups per lane are scattered 4-byte loads (the trace records them for every chain)
        const u32x4 ws = chain_block(a.seed, c, a.t, SLOT_SNK);
        uint32_t iz, i1, i2;
        distinct_three(ws.x, ws.y, ws.z, a.M, iz, i1, i2);
        snk_id[0] = pos_to_chain(a, a.pool_off + iz);
        snk_id[1] = pos_to_chain(a, a.pool_off + i1);
        snk_id[2] = pos_to_chain(a, a.pool_off + i2);
    }
    if (!REGS || (snk_possible && !SNK_DIRECT)) {
#pragma unroll 1
        for (uint32_t idx = (uint32_t)q; idx < npart; idx += LPC)
            s_part[cw * MAX_PARTNERS + idx] = planned ? rec[5 + idx] : pos_to_chain(a, a.pool_off + partner_pos(a, c, idx, 2 * P));
        // a chain subgroup never spans wavefronts: LDS ops of one wavefront complete in order, so a wavefront-scope
        // fence (compiler ordering + lgkmcnt wait) is all the hand-over needs -- no workgroup barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // partner rows requested as soon as their ids exist: their L2/MALL latency overlaps with the mask,
    // jitter and gamma arithmetic below
    constexpr int NPX = NP > 0 ? NP : 1;
    double ra[NPX][DPL], rb[NPX][DPL];
    if (DREAM && NP > 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {             // all 2 NP row loads in flight together
            load_row<LPC, DPL>(row_ptr(a.L, part.get(2 * p)), q, ld, ra[p]);
            load_row<LPC, DPL>(row_ptr(a.L, part.get(2 * p + 1)), q, ld, rb[p]);
        }
    }
    BPM_STAMP(7);
    if (dims_late) wpair[0] = chain_block(a.seed, c, a.t, SLOT_DIM0 + (uint32_t)q);      // overlaps with the row fetches
    if (DREAM) early(wk);

    // ---- per-pair draws: one Philox block per coordinate pair
    double eps_n[DPL], eps_u[DPL];
    uint32_t maskbits = 0;
    wk.cr_idx = -1; wk.d_prime = (int)dim; wk.jump = 0; wk.snk = 0; wk.delta = 0.0; wk.log_corr = 0.0;
    uint32_t thr = 65536u;
    if (DREAM) {
        // cr ~ Categorical(CR, p_cr)  (dream.py:51)
        const double uc = u_sel;
        double cum = 0.0;
        int idx = (int)a.n_cr - 1;
        bool found = false;
#pragma unroll
        for (int m = 0; m < MAX_CR; ++m) {               // first m with uc < cumsum(p_cr)[m]
            if (m < (int)a.n_cr) {
                cum += pcr[m];
                if (!found && uc < cum) { idx = m; thr = a.thr[m]; found = true; }   // k 2^-16 <= CR_m  <=>  k <= thr[m]
            }
        }
        if (!found) {                                    // (no dynamic index into the argument block: it must stay scalarisable)
#pragma unroll
            for (int m = 0; m < MAX_CR; ++m) if (m == (int)a.n_cr - 1) thr = a.thr[m];
        }
        wk.cr_idx = idx;
    }
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < DPL / 2; ++u) {
        const uint32_t pi = (uint32_t)(q + u * LPC);
        const uint32_t j0 = 2u * pi;
        eps_n[2 * u] = 0.0; eps_n[2 * u + 1] = 0.0; eps_u[2 * u] = 0.0; eps_u[2 * u + 1] = 0.0;
        if (j0 < dim) {
            const u32x4 wj = wpair[u];
            const bool two = j0 + 1 < dim;
            if (a.epsilon > 0.0) {                                   // util.py:5-16
                double n0, n1;
                box_muller_pair_f32(wj.z, wj.w, n0, n1);
                eps_n[2 * u] = a.epsilon * n0;
                if (two) eps_n[2 * u + 1] = a.epsilon * n1;
            }
            if (DREAM) {
                if (a.u_epsilon > 0.0) {                             // util.py:18-28
                    eps_u[2 * u] = -a.u_epsilon + (2.0 * a.u_epsilon) * (((double)(wj.y >> 16) + 0.5) * 1.52587890625e-05);
                    if (two) eps_u[2 * u + 1] = -a.u_epsilon + (2.0 * a.u_epsilon) * (((double)(wj.y & 0xFFFFu) + 0.5) * 1.52587890625e-05);
                }
                if ((wj.x >> 16) <= thr) { maskbits |= 1u << (2 * u); ++cnt; }              // dream.py:52-53
                if (two && (wj.x & 0xFFFFu) <= thr) { maskbits |= 1u << (2 * u + 1); ++cnt; }
            }
        }
    }
    if (DREAM) {
        if (LPC == WAVE) {
            cnt = 0;
#pragma unroll
            for (int sb = 0; sb < DPL; ++sb) cnt += (int)__popcll(__ballot((maskbits >> sb) & 1u));
        } else {
            cnt = gsum_i<LPC>(cnt);
        }
        if (cnt == 0) {                                // dream.py:55-57: force one dimension
            const uint32_t f = mulhi32(h0.y, dim);
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (s & 1);
                if (j == f) maskbits |= 1u << s;
            }
            cnt = 1;
        }
        wk.d_prime = cnt;
    }
    wk.maskbits = maskbits;

    if (DREAM) {
        // gamma (dream.py:61,77-80)
        double gamma;                               // gamma_scale * 2.38 / sqrt(2 P d')
        if (LPC == WAVE && (uint32_t)cnt < (uint32_t)(WAVE * DPL)) {
            double gsel = gtab[0];
#pragma unroll
            for (int u = 1; u < DPL; ++u) gsel = ((cnt >> 6) == u) ? gtab[u] : gsel;
            gamma = readlane_f64(gsel, cnt & 63);
        } else if (LPC * DPL < WAVE) {
            gamma = __shfl(gt_lane, cnt);
        } else {
            gamma = a.gamma_tab[cnt];
        }
        if (a.k % 5 == 0 && !(u_gam < 0.2)) { gamma = 1.0; wk.jump = 1; }
        wk.gamma = gamma;
        // sum over pairs of (A_p - B_p)  (dream.py:65-68,85-86), p = 0 first
        double sum[DPL];
        if (NP > 0) {
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                sum[s] = ra[0][s] - rb[0][s];
#pragma unroll
                for (int p = 1; p < NP; ++p) sum[s] = sum[s] + (ra[p][s] - rb[p][s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < DPL; ++s) sum[s] = 0.0;
#pragma unroll 1
            for (uint32_t p = 0; p < P; ++p) {
                double ra[DPL], rb[DPL];
                load_row<LPC, DPL>(row_ptr(a.L, part.lds[2 * p]), q, ld, ra);
                load_row<LPC, DPL>(row_ptr(a.L, part.lds[2 * p + 1]), q, ld, rb);
#pragma unroll
                for (int s = 0; s < DPL; ++s) sum[s] = (p == 0) ? (ra[s] - rb[s]) : (sum[s] + (ra[s] - rb[s]));
            }
        }
        // proposal (dream.py:85-89)
#pragma unroll
        for (int s = 0; s < DPL; ++s) {
            const double jump = (1.0 + eps_u[s]) * gamma * sum[s] + eps_n[s];
            wk.p[s] = ((maskbits >> s) & 1u) ? (jump + wk.x[s]) : wk.x[s];
        }
        BPM_STAMP(4);
        // CR statistic (dream.py:92-93,119-130): BEFORE the accept test, from the proposed jump
        if (a.adapt_on && a.cr_gate) {
            const double* m2 = wk.w_m2;
            double dl = 0.0;
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (s & 1);
                if (j < dim) {
                    // (x - x')^2 / std^2 with std^2 = m2 / n (std == 0 -> 1e-12, dream.py:129): ONE division instead of a division, a
                    // square root and another division -- ~100 instructions per lane less in the burn-in kernel; the result differs
                    // from sqrt-then-square by an ulp or two, far inside the 1e-12 the moments themselves are good for
                    const double df = wk.x[s] - wk.p[s];
                    dl += m2[s] > 0.0 ? (df * df * (double)a.hist_len) / m2[s] : (df * df) / 1e-24;
                }
            }
            wk.delta = gsum<LPC>(dl);
        }
    } else {
        // DE-MC (demc.py:161-182)
        double gamma = a.gamma_demc;
        if (a.mode != 2 && a.k % 10 == 0 && !(u_gam < 0.1)) { gamma = 1.0; wk.jump = 1; }   // demc.py:174-177 (not in samplers.py DeMc)
        wk.gamma = gamma;
        double ra[DPL], rb[DPL];
        const bool ids_in_regs = REGS && (!snk_possible || SNK_DIRECT);
        const uint32_t ca = ids_in_regs ? part.get(0) : part.lds[0];
        const uint32_t cb = ids_in_regs ? part.get(1) : part.lds[1];
        // ALL row requests of the update leave together.  With one lane per chain the partner ids come from table lookups (vector loads, which return
        // in order) and the snooker lookups sit in a divergent branch: left to itself the compiler requested row A as soon as id A was in, then waited
        // for id B with a count that also covers row A (the conservative merge over the branch), and fetched the three snooker rows only after the pair
        // proposal was built -- three memory round trips in a row where one does.  The empty asm makes every id an input at ONE point.
        const bool do_snk = snk_possible && u_sel < a.p_snooker;
        if (LPC == 1) asm volatile("" ::"v"(ca), "v"(cb), "v"(snk_id[0]), "v"(snk_id[1]), "v"(snk_id[2]));
        load_row<LPC, DPL>(row_ptr(a.L, ca), q, ld, ra);
        load_row<LPC, DPL>(row_ptr(a.L, cb), q, ld, rb);
        double rz[DPL], r1[DPL], r2[DPL];
#pragma unroll
        for (int s = 0; s < DPL; ++s) { rz[s] = 0.0; r1[s] = 0.0; r2[s] = 0.0; }
        if (do_snk) {
            load_row<LPC, DPL>(row_ptr(a.L, SNK_DIRECT ? snk_id[0] : part.lds[2]), q, ld, rz);
            load_row<LPC, DPL>(row_ptr(a.L, SNK_DIRECT ? snk_id[1] : part.lds[3]), q, ld, r1);
            load_row<LPC, DPL>(row_ptr(a.L, SNK_DIRECT ? snk_id[2] : part.lds[4]), q, ld, r2);
        }
        early(wk);
#pragma unroll
        for (int s = 0; s < DPL; ++s) {
            double pv = gamma * (ra[s] - rb[s]);
            pv = pv + wk.x[s];
            pv = pv + eps_n[s];
            wk.p[s] = pv;
        }
        if (do_snk) {
            // snooker update (ter Braak & Vrugt 2008) -- extension, absent from the reference
            double n2 = 0.0, dot = 0.0;
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                const double df = wk.x[s] - rz[s];
                n2 += df * df;
                dot += (r1[s] - r2[s]) * df;
            }
            n2 = gsum<LPC>(n2);
            dot = gsum<LPC>(dot);
            if (n2 > 0.0) {
                const double gs = (1.2 + u01_32(h0.y)) * (dot / n2);
                double n2p = 0.0;
#pragma unroll
                for (int s = 0; s < DPL; ++s) {
                    const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (s & 1);
                    const double df = wk.x[s] - rz[s];
                    double pv = wk.x[s] + gs * df;
                    pv = pv + eps_n[s];
                    if (j >= dim) pv = wk.x[s];
                    wk.p[s] = pv;
                    const double dn = pv - rz[s];
                    n2p += dn * dn;
                }
                n2p = gsum<LPC>(n2p);
                wk.log_corr = 0.5 * (double)(dim - 1) * (log(n2p) - log(n2));
                wk.snk = 1;
            }
        }
    }
    if (trace_i32_of(a) && active) {
        if (q == 0) {
            int32_t* tr = trace_i32_of(a) + (uint64_t)(c - a.lo) * TRACE_I32;
            tr[0] = wk.cr_idx; tr[1] = wk.d_prime; tr[2] = wk.jump; tr[4] = wk.snk;
            if (REGS) {
#pragma unroll
                for (int i = 0; i < 2 * NP; ++i) tr[5 + i] = (int32_t)part.get(i);
            }
            const bool regs = REGS && (!snk_possible || SNK_DIRECT);
#pragma unroll 1
            for (uint32_t i = regs ? 2u * P : 0u; i < MAX_PARTNERS; ++i) {
                int32_t v = -1;
                const uint32_t ks = i - 2u * P;      // (a select chain, not snk_id[ks]: a dynamically indexed array would live in scratch memory)
                if (i < npart) v = regs ? (int32_t)(ks == 1u ? snk_id[1] : (ks == 2u ? snk_id[2] : snk_id[0])) : (int32_t)part.lds[i];
                tr[5 + i] = v;
            }
        }
        if (trace_mask_of(a)) {
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                const uint32_t j = 2u * (uint32_t)(q + (s >> 1) * LPC) + (s & 1);
                if (j < dim) trace_mask_of(a)[(uint64_t)(c - a.lo) * dim + j] = (uint8_t)((maskbits >> s) & 1u);
            }
        }
    }
}

// Metropolis test (samplers.py:328-336), append (chain.py:51-54), Welford moments, CR outputs.
// CRP_W: this kernel flavour sums level 1 of the CR reduction itself (phase_fused_kernel: CRP) -- the only one that may leave the chains' CR slots unwritten.
template <int ALGO, int LPC, int DPL, int LEAN = 0 /* 0 never, 1 when a.lean, 2 always */, bool CRP_W = false>
__device__ __forceinline__ void finish_update(const PhaseArgs& a, uint32_t c, bool active, int q,
                                              const Work<DPL>& wk, double ll_prop) {
    const uint32_t ld = a.L.ld;
    const uint32_t li = c - a.lo;
    const double ll_cur = wk.ll_cur;
    double alpha = exp((ll_prop + wk.log_corr) - ll_cur);
    const bool is_nan = alpha != alpha;
    alpha = fmin(1.0, alpha);
    alpha = fmax(0.0, alpha);                            // np.clip(np.min((1, .)), 0, 1); NaN stays NaN in NumPy
    const bool accepted = !is_nan && (u01_53(wk.acc_hi, wk.acc_lo) < alpha);
    const bool lean = LEAN == 2 || (LEAN == 1 && a.lean != 0u);
    if (lean) {      // one counter bump per wavefront (all lanes are still here): acc_count[n_local + shard]
        const unsigned long long m = __ballot(active && accepted && q == 0);
        if ((threadIdx.x & (WAVE - 1)) == 0 && m != 0ull)
            __hip_atomic_fetch_add(&a.acc_count[a.L.n_local + ((blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE)) & (ACC_SHARDS - 1))],
                                   (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!active) return;
    // accept bookkeeping without same-address atomics (8192 of them serialise at ~12 ns each):
    // every chain owns one counter, the host sums them (demc.py:143-150)
    if (q == 0) {
        if (accepted && !lean) {
#if defined(BPM_TEST_HOOKS) && defined(BPM_FAKE_NO_LL)
            if (LPC < WAVE) {} else
#endif
            if (LPC < WAVE) __hip_atomic_fetch_add(&a.acc_count[li], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (a.wt) __hip_atomic_store(&a.acc_count[li], wk.acc_prev + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else a.acc_count[li] = wk.acc_prev + 1u;
        }
        if (is_nan) atomicAdd(&a.counters[2], 1ull);
        if (a.accbits) a.accbits[a.acc_by_item ? wk.item : li] = accepted ? (uint8_t)1 : (uint8_t)0;
    }
    double nv[DPL];
#pragma unroll
    for (int s = 0; s < DPL; ++s) nv[s] = accepted ? wk.p[s] : wk.x[s];
    const double new_ll = accepted ? ll_prop : ll_cur;
    if (a.x_next) {
        // synchronous generation (samplers.py:300-308 delayed_accept): updates are banked, applied after the launch
        store_row<LPC, DPL>(a.x_next + (uint32_t)(li * ld), q, ld, nv);
        if (accepted && q == 0 && !lean) a.ll[li] = new_ll;
    } else if (accepted) {
        // (lean: the ln-like cache is not written -- nothing on this path reads it, the host refreshes it on demand)
        if (a.wt == 2u) {
            store_row_wt16<LPC, DPL>(row_ptr(a.L, c), q, ld, nv);
#if defined(BPM_TEST_HOOKS) && defined(BPM_FAKE_NO_LL)
            if (LPC < WAVE) {} else
#endif
            if (q == 0 && !lean) __hip_atomic_store(&a.ll[li], new_ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (a.wt) {
            store_row_wt<LPC, DPL>(row_ptr(a.L, c), q, ld, nv);
            if (q == 0 && !lean) __hip_atomic_store(&a.ll[li], new_ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            store_row<LPC, DPL>(row_ptr(a.L, c), q, ld, nv);
            if (q == 0 && !lean) a.ll[li] = new_ll;
        }
    }
    if (a.pack && accepted) {
        // sparse exchange: only rows that changed travel.  One slot per accepted chain (order irrelevant: the
        // receivers scatter by chain id); a chain beyond the capacity is counted but not packed -- the host sees
        // count > cap after the chunk and replays it with the dense all-gather (results identical by construction).
        // The block is split into pack_nsub sub-blocks with a counter each (a power of two; chain li uses li % nsub): atomics on one
        // address retire one at a time (~12 ns each), hundreds of acceptances on one counter would cost more than
        // the update itself.
        double* sub = a.pack + (uint32_t)((li & (a.pack_nsub - 1u)) * a.pack_stride);
        uint32_t slot = 0;
        if (q == 0) slot = atomicAdd(reinterpret_cast<uint32_t*>(sub), 1u);
        slot = __shfl(slot, (threadIdx.x & (WAVE - 1)) - q);           // lane 0 of the subgroup
        if (slot < a.pack_cap) {
            if (q == 0) sub[2 + slot] = (double)c;
            store_row<LPC, DPL>(sub + 2 + a.pack_cap + (uint32_t)(slot * ld), q, ld, nv);
        }
    }
    if (a.n_peers) {
        // push exchange: the owner writes what changed into every other rank's replica (same offsets: the replicas have one layout)
        const uint32_t row_off = (uint32_t)(row_ptr(a.L, c) - a.L.G);
        // CR statistics of EVERY update travel during burn-in (dream.py:92) -- once the gate of dream.py:123 is open: before that nobody reduces
        // them (sampler.hip: gen_cr_reduce), and the first generation behind the gate rewrites every chain's slots in every replica
        const bool slots = ALGO == ALGO_DREAM && a.adapt_on && a.cr_gate && q == 0;
        const uint32_t d_off = (uint32_t)(delta_ptr(a.L, c) - a.L.G), c_off = (uint32_t)(cridx_ptr(a.L, c) - a.L.G);
        const bool gated = a.adapt_on && a.cr_gate;
#pragma unroll 1
        for (uint32_t p = 0; p < a.n_peers; ++p) {
            double* pg = reinterpret_cast<double*>(a.peer_tab[p]);
            if (accepted) store_row_sys<LPC, DPL>(pg + row_off, q, ld, nv);
            if (slots) {
                __hip_atomic_store(pg + d_off, gated ? wk.delta : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(pg + c_off, gated ? (double)wk.cr_idx : -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    const uint32_t hi = a.hist_by_pos ? wk.pos_own : li;
    if (a.hist_row) store_row_stream<LPC, DPL>(a.hist_row + (uint32_t)(hi * ld), q, ld, nv);
    // (a streaming store like the row's: nobody reads the ln-like history inside the generation loop.  As a plain store the by-chain form cost cfg5's burn-in
    // 3.4 us per generation, this one 1.0; the by-position form gains too: cfg5 45.5 -> 44.5)
    // (one wavefront per chain -- a single 8-byte store per wavefront -- keeps the ordinary store: nothing to gain, and under a kernel trace the streaming
    // form showed up as +0.25 us of kernel duration at cfg2)
    if (a.llhist_row && q == 0) {
        double* dst_ll = &a.llhist_row[a.hist_by_pos == 1u ? wk.pos_own : li];
        if (LPC < WAVE) __builtin_nontemporal_store(new_ll, dst_ll); else *dst_ll = new_ll;
    }
    if (ALGO == ALGO_DREAM) {
        if (a.adapt_on) {
            // running moments of this chain's own history (replaces np.std(chain.chain), dream.py:128)
            double mean[DPL], m2[DPL];
#pragma unroll
            for (int s = 0; s < DPL; ++s) { mean[s] = wk.w_mean[s]; m2[s] = wk.w_m2[s]; }
            const double cntp = (double)(a.hist_len + 1);
#pragma unroll
            for (int s = 0; s < DPL; ++s) {
                const double d1 = nv[s] - mean[s];
                mean[s] = mean[s] + d1 / cntp;
                m2[s] = m2[s] + d1 * (nv[s] - mean[s]);
            }
            if (a.wt) {      // release-less packets: the next generation's kernels read these rows
                store_row_wt16<LPC, DPL>(a.w_mean + (uint32_t)(li * 2u * ld), q, ld, mean);
                store_row_wt16<LPC, DPL>(a.w_m2 + (uint32_t)(li * 2u * ld), q, ld, m2);
            } else {
                store_row_stream<LPC, DPL>(a.w_mean + (uint32_t)(li * 2u * ld), q, ld, mean);
                store_row_stream<LPC, DPL>(a.w_m2 + (uint32_t)(li * 2u * ld), q, ld, m2);
            }
        }
        // The chain's CR slots (delta | CR index) are what the reduction kernels read (cr_level1_kernel) -- only while CR adaptation runs, and not when the
        // update kernels sum level 1 themselves (cr_part1).  Every generation of an adaptation phase rewrites every chain's slots before its reduction
        // reads them, so nothing has to be written in between: until round 4 every DREAM update stored "no statistic" here -- two scattered 8-byte
        // write-through stores per update in the steady state (cfg5: 44.5 -> ... us per generation without them).
        if (q == 0 && a.adapt_on && !(CRP_W && a.cr_part1 != nullptr)) {      // (structural: a flavour that cannot write level 1 ALWAYS writes the slots)
            const bool gated = a.adapt_on && a.cr_gate;
            if (a.wt) {
                __hip_atomic_store(delta_ptr(a.L, c), gated ? wk.delta : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cridx_ptr(a.L, c), gated ? (double)wk.cr_idx : -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *delta_ptr(a.L, c) = gated ? wk.delta : 0.0;
                *cridx_ptr(a.L, c) = gated ? (double)wk.cr_idx : -1.0;
            }
        }
    }
    if (trace_i32_of(a) && q == 0) {
        trace_i32_of(a)[(uint64_t)li * TRACE_I32 + 3] = accepted ? 1 : 0;
        double* tf = trace_f64_of(a) + (uint64_t)li * TRACE_F64;
        tf[0] = alpha; tf[1] = ll_prop; tf[2] = wk.delta; tf[3] = wk.gamma;
    }
    // push exchange: a wavefront ends only when its stores into the peers' replicas have been acknowledged at system scope (gfx942 / gfx950 count
    // stores in vmcnt: the #error at the top of this file).  The packet of this kernel may carry no release fence (agent-scope mode), and the flag
    // that announces this half generation to the peers is stored by the NEXT packet of the queue: with this wait "kernel complete" implies "pushes
    // performed" by the ISA's own rules rather than by the order in which a fabric happens to deliver posted writes.
    // (The GCN / CDNA ISA manuals describe S_ENDPGM as implying S_WAITCNT 0, which would make this redundant -- and the same wait behind the
    // single-GPU write-through stores measured free at every size but one: profiles/r03_wave_end_store_wait.txt.  Written out where another GPU
    // depends on it.)
    if (a.n_peers) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The kernel argument block is read with scalar loads.  Left alone, the compiler loads each field right before its
// first use -- ~80 separate s_load + s_waitcnt pairs threaded through the kernel's branches.  Naming the hot fields as
// SGPR inputs of an empty asm statement in the entry block makes them live there: the loads coalesce into a few wide
// s_load_dwordx8/x16 behind one wait.  Measured: 13.1 vs 15.2 us/generation for the 4-lanes-per-chain kernel
// (cfg5/8), no change with one lane per chain, and a LOSS with one wavefront per chain (17.1 vs 15.8, cfg2: there
// the lazy loads hide behind the row fetches) -- so only kernels with several chains per wavefront do it.
template <int LPC>
__device__ __forceinline__ void pin_args(const PhaseArgs& a) {
    if (LPC != WAVE) {
        asm volatile("" ::"s"(a.L.G), "s"(a.L.ld), "s"(a.L.dim), "s"(a.ll), "s"(a.hist_row), "s"(a.llhist_row), "s"(a.tparams),
                     "s"(a.cr_state), "s"(a.acc_count), "s"(a.perm_tab), "s"(a.inv_tab), "s"(a.gamma_tab), "s"(a.plan), "s"(a.pack),
                     "s"(a.thr[0]), "s"(a.thr[1]), "s"(a.thr[2]), "s"(a.seed), "s"(a.t), "s"(a.k), "s"(a.N),
                     "s"(a.lo), "s"(a.upd_off), "s"(a.n_upd), "s"(a.pool_off), "s"(a.M), "s"(a.mode), "s"(a.n_items), "s"(a.n_cr));
        asm volatile("" ::"s"(a.adapt_on), "s"(a.cr_gate), "s"(a.hist_len), "s"(a.epsilon), "s"(a.u_epsilon), "s"(a.L.n_local),
                     "s"(a.L.world), "s"(a.L.magic), "s"(a.counters), "s"(a.pack_cap), "s"(a.pack_nsub), "s"(a.pack_stride));
    }
}

// Which chain does this subgroup update?  mode 0: work item = position in shuffle order (all items
// active); mode 1: work item = local chain, active iff its position falls in the phase's group.
__device__ __forceinline__ bool resolve_chain(const PhaseArgs& a, uint32_t w, uint32_t& c) {
    bool active = w < a.n_items;
    if (a.mode == 0) {
        c = own_pos_to_chain(a, a.upd_off + (active ? w : 0u));
    } else if (a.mode == 2) {
        c = a.lo + (active ? w : 0u);
    } else {
        c = a.lo + (active ? w : 0u);
        const uint32_t pos = chain_to_pos(a, c);
        active = active && (pos - a.upd_off) < a.n_upd;
    }
    return active;
}

// HOT: frequent cases as their own instantiations: 1 = single GPU, steady state, with plan records, 2 = the same without,
// 3 / 4 = the same during DREAM's CR adaptation (burn-in), 5 / 6 = a rank of a multi-GPU world in the steady state
// with the replay exchange (accept bytes; its compacted records when 5, none when 6); 0 = the general kernel.
__host__ inline bool phase_args_hot_sharded(const PhaseArgs& a, bool dream, bool with_plan) {
    return a.mode == 0 && (a.rec_tab != nullptr) == with_plan && trace_i32_of(a) == nullptr && a.pack == nullptr && a.x_next == nullptr &&
           a.adapt_on == 0 && a.epsilon > 0.0 && a.L.world > 1 &&
           a.perm_tab != nullptr && a.inv_tab != nullptr && a.stamps == nullptr && (a.accbits != nullptr || a.n_peers > 0) && a.replay == 0 &&
           (!dream || (a.u_epsilon > 0.0 && a.n_cr == 3));
}
__host__ inline bool phase_args_hot(const PhaseArgs& a, bool dream, bool with_plan, bool adapting) {
    return a.n_peers == 0 && a.mode == 0 && (a.rec_tab != nullptr) == with_plan && trace_i32_of(a) == nullptr && a.pack == nullptr && a.x_next == nullptr &&
           (a.adapt_on != 0) == adapting && a.epsilon > 0.0 && a.L.world == 1 &&
           a.perm_tab != nullptr && a.inv_tab != nullptr && a.lo == 0 && a.stamps == nullptr && a.accbits == nullptr && a.replay == 0 &&
           (!dream || (a.u_epsilon > 0.0 && a.n_cr == 3));
}
template <int ALGO, int TARGET, int LPC, int DPL, int NP, int HOT = 0>
__global__ __launch_bounds__(block_for_hot(LPC, HOT, DPL)) void phase_fused_kernel(
#ifdef BPM_PRELOAD
    // the four values that locate a wavefront's update record, as leading scalar arguments: with
    // -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs at wavefront launch (no kernarg load, one miss less
    // on the critical path); they repeat a.rec_tab, a.rec_off, a.n_items, a.mode
    const uint32_t* pl_plan, uint32_t pl_upd_off, uint32_t pl_n_items, uint32_t pl_mode,
#endif
    const PhaseArgs a_in) {
    // HOT: the host launches this instantiation only when phase_args_hot() / phase_args_hot_sharded() holds.  What the
    // predicate fixes is written into a private copy of the argument block (constants the compiler propagates; an
    // assumption on a POINTER loaded from the kernarg segment does not survive address-space inference), the positive /
    // non-null ones are assumed; every other path drops out -- cfg2's kernel: 637 instead of 2520 instructions, 33
    // instead of 144 branches, 51 instead of 82 VGPRs -- and the loads move across what were branch boundaries:
    // 12.5 vs 14.4 us/generation at cfg2, 11.0 vs 12.9 at cfg5/8.  (The copy stays in registers only as long as nothing
    // indexes the argument block dynamically: one `a.thr[n_cr - 1]` had put it into scratch, 30 us/generation at cfg5/8.)
    constexpr bool COPY = (HOT != 0);
    constexpr bool SHARD = (HOT == 5 || HOT == 6), ADAPT = (HOT == 3 || HOT == 4), NOPLAN = (HOT == 2 || HOT == 4 || HOT == 6);
    PhaseArgs a_hot;
    if (COPY) {
        a_hot = a_in;
        a_hot.mode = 0u; trace_set(a_hot, nullptr, nullptr, nullptr); a_hot.pack = nullptr;
        a_hot.replay = 0u; a_hot.x_next = nullptr; a_hot.adapt_on = ADAPT ? 1u : 0u; a_hot.stamps = nullptr;
        if (!ADAPT) a_hot.cr_fold_part = nullptr;
        if (!SHARD) { a_hot.accbits = nullptr; a_hot.lo = 0; a_hot.L.world = 1; a_hot.acc_by_item = 0u; a_hot.n_peers = 0u; a_hot.peer_tab = nullptr; }
        if (ALGO == ALGO_DREAM) a_hot.n_cr = 3;
        if (NOPLAN) { a_hot.plan = nullptr; a_hot.rec_tab = nullptr; }
    }
    const PhaseArgs& a = COPY ? a_hot : a_in;
    if (HOT) {
        __builtin_assume(a_in.mode == 0u); __builtin_assume(a_in.pack == nullptr);
#ifdef BPM_TEST_HOOKS
        __builtin_assume(a_in.trace_i32 == nullptr);
#endif
        __builtin_assume(a_in.x_next == nullptr); __builtin_assume(a_in.adapt_on == (ADAPT ? 1u : 0u)); __builtin_assume(a_in.stamps == nullptr);
        if (!SHARD) { __builtin_assume(a_in.lo == 0); __builtin_assume(a_in.L.world == 1); }
        __builtin_assume(a_in.epsilon > 0.0);
        __builtin_assume(a_in.perm_tab != nullptr); __builtin_assume(a_in.inv_tab != nullptr);
        if (ALGO == ALGO_DREAM) { __builtin_assume(a_in.u_epsilon > 0.0); __builtin_assume(a_in.n_cr == 3); }
    }
#ifdef BPM_PRELOAD
    if (HOT) { __builtin_assume(pl_mode == 0u); __builtin_assume((pl_plan != nullptr) == !NOPLAN); }
#else
    const uint32_t* pl_plan = a.rec_tab;
    const uint32_t pl_upd_off = a.rec_off, pl_n_items = a.n_items, pl_mode = a.mode;
#endif
    constexpr int BLK = block_for_hot(LPC, HOT, DPL);
    // CRP: this flavour sums its updates' CR statistics itself (level 1 of the CR reduction); with one wavefront per chain the 16 wavefronts of the
    // workgroup meet at a barrier for it, so none of them may leave early
    constexpr bool CRP = ALGO == ALGO_DREAM && hot_is_adapt(HOT) && crp_shape(LPC, DPL);
    __shared__ uint32_t s_part[(BLK / LPC) * MAX_PARTNERS];
    const int lane = threadIdx.x;
    const int cw = lane / LPC, q = lane % LPC;      // chain slot inside the workgroup, lane inside the chain subgroup
    const uint32_t w = blockIdx.x * (BLK / LPC) + cw;
#ifdef BPM_STAMPS
    unsigned long long bpm_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long bpm_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bpm_rt0)::"memory");
    BPM_STAMP(0);
#endif
    pin_args<LPC>(a);
    uint32_t c;
    bool active;
    const uint32_t* rec = nullptr;
    bool run = true;                     // (false: an idle wavefront of a CRP workgroup -- it only joins the barrier)
    double cr_d = 0.0;
    int cr_i = -1;
    if (LPC == WAVE && pl_plan && !(CRP && w >= pl_n_items)) {
        // the update's record (by position in shuffle order) carries the chain id: wavefront-uniform scalar loads
        active = w < pl_n_items;
        if (!active) return;
        uint32_t pos;
        if (pl_mode == 0) {
            pos = pl_upd_off + w;
        } else {
            pos = a.inv_tab[a.lo + w];
            if ((pos - pl_upd_off) >= a.n_upd) return;
        }
        pos = __builtin_amdgcn_readfirstlane(pos);
        rec = pl_plan + (uint32_t)(pos * PLAN_WORDS);
        c = rec[0];
    } else {
        active = resolve_chain(a, w, c);
    }
    if (LPC == WAVE && !active) {                // whole wavefront idle
        if (CRP) run = false; else return;
    }
    // ---- consumer-side fold of the PREVIOUS generation's CR statistics (PhaseArgs::cr_fold_part), cr_final_kernel's arithmetic spread over the
    // workgroup: each of the 2 n_cr arrays of partial sums (delta sums / counts of one CR value) is summed by ONE wavefront exactly as cr_fold does --
    // up to 8 loads per lane, in flight behind the wavefront's record load -- the totals meet in LDS; behind the barrier every wavefront finishes the
    // fold for itself (a few additions and divisions on uniform values).  Same sums in the same order as cr_final_kernel: same bits (tested against the
    // crnofold path).  One wavefront folding everything (48 loads per lane, six DPP trees) cost cfg2's launch 2.0 us, this form 1.3 us -- against the 4.5 us
    // dispatch of cr_final_kernel it replaces: cfg2's burn-in generation 20.9 -> 18.5 us (profiles/r05_consumer_side_fold.txt).
    // (The barrier HERE, ahead of every wavefront's loads: with it where the CR value is drawn -- behind the row requests, so that only the folding
    // wavefronts would feel the fold -- the compiler serialised the partner-row loads of EVERY launch of the flavour, 23 instead of 8.3 us per launch.)
    PcrGiven pcr_fold;
    if constexpr (CRP) {
        pcr_fold.on = a.cr_fold_part != nullptr;         // uniform over the launch
        if (pcr_fold.on) {
            __shared__ double s_tot[2 * MAX_CR];
            constexpr uint32_t NWV = (uint32_t)(BLK / WAVE);
            const uint32_t n_cr = a.n_cr, wv = (uint32_t)lane / (uint32_t)WAVE;
            for (uint32_t k = wv; k < 2u * n_cr; k += NWV) {      // uniform per wavefront
                const uint32_t arr = k < n_cr ? k : (uint32_t)MAX_CR + (k - n_cr);
                const double tsum = cr_array_sum<CR_FINAL_MAX / WAVE>(a.cr_fold_part + (uint64_t)arr * a.cr_fold_nb, a.cr_fold_nb);
                if ((lane & (WAVE - 1)) == 0) s_tot[arr] = tsum;
            }
            __syncthreads();
            CrTotals T;
#pragma unroll
            for (int m = 0; m < MAX_CR; ++m) { T.p[m] = a.cr_fold_tot[m]; T.d[m] = a.cr_fold_tot[MAX_CR + m]; T.n[m] = a.cr_fold_tot[2 * MAX_CR + m]; }
            bool any = false;
#pragma unroll
            for (int m = 0; m < MAX_CR; ++m) {
                if (m < (int)n_cr) {
                    const double td = s_tot[m], tn = s_tot[MAX_CR + m];
                    if (tn > 0.0) { any = true; T.n[m] += tn; T.d[m] += td; }
                }
            }
            cr_finalize(any, n_cr, T);
            if (blockIdx.x == 0 && lane == 0) cr_write_totals(T, n_cr, a.cr_fold_out);
#pragma unroll
            for (int m = 0; m < MAX_CR; ++m)
                pcr_fold.p[m] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(T.p[m])), __builtin_amdgcn_readfirstlane(__double2loint(T.p[m])));
        }
    }
  if (run) {
    if (LPC == WAVE) c = __builtin_amdgcn_readfirstlane(c);   // wavefront-uniform: header draws and Feistel walks go to the scalar unit
    const typename Target<TARGET, LPC, DPL>::Consts tc = Target<TARGET, LPC, DPL>::load(q, a.L.dim, a.tparams);
    Work<DPL> wk;
    wk.item = w;
    wk.pos_own = a.upd_off + (active ? w : 0u);
    constexpr bool LEAN = lean_scalars<TARGET, LPC>();
    // (LEAN: ln_like of the current state from the own row == the cached value, bit for bit -- see lean_scalars)
    const uint32_t dim_early = a.L.dim;      // (captured by value, a few registers: a by-reference capture put one instantiation's constants on the stack)
    constexpr bool LEAN_CT = LEAN && lean_always<TARGET>();
#ifdef BPM_STAMPS
    bpm_stamp[1] = 0; BPM_STAMP(1);
    unsigned long long* const stamp_arg = bpm_stamp;
#else
    unsigned long long* const stamp_arg = nullptr;
#endif
    if constexpr (LEAN) {
        const bool lean_early = LEAN_CT || a.lean != 0u;
        auto early = [tc, q, dim_early, lean_early](Work<DPL>& k) { if (lean_early) k.ll_cur = Target<TARGET, LPC, DPL>::eval(k.x, q, dim_early, tc); };
        make_proposal<ALGO, LPC, DPL, NP, (LEAN_CT ? 0 : 2)>(a, c, active, q, cw, s_part, wk, rec, stamp_arg, early, pcr_fold);
    } else if constexpr (CRP) {
        make_proposal<ALGO, LPC, DPL, NP>(a, c, active, q, cw, s_part, wk, rec, stamp_arg, NoEarly(), pcr_fold);
    } else {      // (one wavefront per chain: exactly the call of rounds 1-3)
        make_proposal<ALGO, LPC, DPL, NP>(a, c, active, q, cw, s_part, wk, rec, stamp_arg);
    }
    const double ll_prop = Target<TARGET, LPC, DPL>::eval(wk.p, q, a.L.dim, tc);
    BPM_STAMP(5);
    finish_update<ALGO, LPC, DPL, (LEAN_CT ? 2 : (LEAN ? 1 : 0)), CRP>(a, c, active, q, wk, ll_prop);
    if (CRP && active && a.adapt_on && a.cr_gate) { cr_d = wk.delta; cr_i = wk.cr_idx; }      // what finish_update wrote into the chain's slots
    BPM_STAMP(6);
#ifdef BPM_STAMPS
    if (bpm_stamp[2] == 0) bpm_stamp[2] = bpm_rt0;
    // slot 7: device-wide 100 MHz counter at the end of the wavefront (s_memtime counters are per CU, not comparable
    // across CUs) with the XCD id in bits 60..63; slot 2 (in-kernel header draw, unused with plan records) holds the
    // same counter at entry when the wavefront took the planned path
    {
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        bpm_stamp[7] = (rt & 0x0FFFFFFFFFFFFFFFull) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 60);
    }
    if (a.stamps && lane == 0) for (int i = 0; i < 8; ++i) a.stamps[(uint64_t)w * 8 + i] = bpm_stamp[i];
#endif
  }   // run
    // ---- level 1 of the CR reduction ("CR reduction" above): this launch's updates, chunk by chunk of cr_g1 positions, in position order
    if (CRP && a.cr_part1) {
        constexpr int G1 = cr_g1(LPC);
        if (LPC == WAVE) {                                   // 16 wavefronts = 16 chains = one chunk: through LDS
            __shared__ double s_crd[G1];
            __shared__ int s_cri[G1];
            if (q == 0) { s_crd[cw] = cr_d; s_cri[cw] = cr_i; }
            __syncthreads();
            if (cw == 0 && q < (int)a.n_cr) {                 // lane m of the first wavefront: CR value m
                double sd = 0.0, sn = 0.0;
#pragma unroll
                for (int k = 0; k < G1; ++k) cr_chunk_add(sd, sn, q, s_cri[k], s_crd[k]);
                const uint32_t ch = a.cr_chunk0 + blockIdx.x;
                __hip_atomic_store(&a.cr_part1[(uint64_t)q * a.cr_n1 + ch], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.cr_part1[(uint64_t)(MAX_CR + q) * a.cr_n1 + ch], sn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {                                             // a wavefront holds G1 chains = one chunk: lane m sums CR value m over them
            const int wl = lane & (WAVE - 1);
            double sd = 0.0, sn = 0.0;
#pragma unroll
            for (int k = 0; k < G1; ++k) {
                const double dk = readlane_f64(cr_d, k * LPC);
                const int ik = __builtin_amdgcn_readlane(cr_i, k * LPC);
                cr_chunk_add(sd, sn, wl, ik, dk);
            }
            const uint32_t ch = a.cr_chunk0 + (blockIdx.x * (BLK / WAVE) + (uint32_t)(lane / WAVE));
            if (wl < (int)a.n_cr && (ch - a.cr_chunk0) * (uint32_t)G1 < a.n_items) {
                __hip_atomic_store(&a.cr_part1[(uint64_t)wl * a.cr_n1 + ch], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.cr_part1[(uint64_t)(MAX_CR + wl) * a.cr_n1 + ch], sn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

#ifdef BPM_EXPERIMENT_XCD
// ---- EXPERIMENT (round 5; `make variant NAME=xcd DEFS=-DBPM_EXPERIMENT_XCD`; never the product) ---------------------------------
// A generation loop RESIDENT ON ONE XCD for populations whose state fits one XCD's 4 MB L2 (cfg3: 1 MB, cfg5's share: 2 MB).  The shipped path
// pays two whole-GPU dependent dispatches per generation (2 x ~2.5 us of launch floor) and gathers partner rows through the Infinity Cache
// (every packet's acquire invalidates the L2s).  Here ONE launch runs n_phases half generations: 256 workgroups of 1024 threads are launched, the
// `want` (32) that find themselves on XCC `xcc_want` become workers (one per CU of that XCD), everybody else exits; a worker updates the work items
// worker, worker + want, ... of every half generation with THE SAME device code as phase_fused_kernel (make_proposal / Target::eval /
// finish_update: bits identical), rows written with plain stores -- they land in THAT XCD's L2, the only L2 any reader of this launch uses --
// and between two half generations the workers meet at an XCD-local barrier: every wave waits for its stores (s_waitcnt vmcnt(0): at workgroup
// scope in threadgroup-split terms, i.e. "visible in the L2", that is all a release needs on gfx942 / gfx950), one lane per workgroup adds to a
// counter WITHOUT sc1 -- the atomic executes in this XCD's L2 -- and polls it, then every wave drops its CU's vector L1 (buffer_inv sc0).
// Every wait is bounded (100 MHz clock): a worker that runs into the limit sets ctl[3] and every worker leaves at the next check.
//   ctl[0] registration ticket, ctl[1] barrier counter, ctl[2] workers registered (diagnostic), ctl[3] error, ctl[4..5] XCC ids seen (diagnostic)
__device__ __forceinline__ unsigned long long xcd_now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ uint32_t xcd_l2_add(uint32_t* p, uint32_t v) {      // returns the old value; performed in the XCD's own L2 (no sc1: not agent scope)
    uint32_t old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
// -> false when a wait ran into its limit (or another worker reported one)
__device__ __forceinline__ bool xcd_barrier(uint32_t* ctl, uint32_t target, unsigned long long deadline) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's rows are in the L2
    __syncthreads();
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) {
        uint32_t ok = 1u;
        (void)xcd_l2_add(&ctl[1], 1u);
        while (xcd_l2_add(&ctl[1], 0u) < target) {
            if (xcd_l2_add(&ctl[3], 0u) != 0u) { ok = 0u; break; }
            if (xcd_now() > deadline) { (void)xcd_l2_add(&ctl[3], 1u); ok = 0u; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_ok = ok;
    }
    __syncthreads();
    const bool ok = s_ok != 0u;
#ifdef BPM_XCD_INV_SC1
    asm volatile("buffer_inv sc1" ::: "memory");                           // (agent-scope invalidate: the L1 and the L2's non-local lines)
#else
    asm volatile("buffer_inv sc0" ::: "memory");                           // the other CUs' rows are in the L2, not in this CU's L1
#endif
    return ok;
}
template <int ALGO, int TARGET, int LPC, int DPL, int NP>
__global__ __launch_bounds__(1024) void xcd_resident_kernel(const PhaseArgs* args_g, uint32_t n_phases, uint32_t* ctl, uint32_t want, uint32_t xcc_want,
                                                            unsigned long long timeout_ticks) {
    static_assert(LPC < WAVE, "several chains per wavefront: the shapes whose populations fit an L2");
    constexpr int BLK = 1024, CPW = BLK / LPC;
    __shared__ uint32_t s_part[1];            // (these shapes keep partner ids in registers: FAST / QUAD paths of make_proposal)
    __shared__ uint32_t s_worker;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
    if (threadIdx.x == 0) {
        s_worker = 0xFFFFFFFFu;
        if (xcc == xcc_want) s_worker = atomicAdd(&ctl[0], 1u);
        atomicOr(&ctl[4 + (xcc >> 5)], 1u << (xcc & 31u));
    }
    __syncthreads();
    const uint32_t worker = s_worker;
    if (worker >= want) return;
    const unsigned long long deadline = xcd_now() + timeout_ticks;
    const int lane = threadIdx.x;
    const int cw = lane / LPC, q = lane % LPC;
    // the argument blocks through the constant address space: uniform addresses -> scalar loads, the fields live in SGPRs like kernel arguments
    typedef const uint32_t __attribute__((address_space(4)))* CWords;
    auto load_args = [args_g](uint32_t i, PhaseArgs& a) {
        static_assert(sizeof(PhaseArgs) % 4 == 0, "argument block in words");
        const CWords src = (CWords)(args_g + i);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&a);
#pragma unroll
        for (uint32_t j = 0; j < sizeof(PhaseArgs) / 4u; ++j) dst[j] = src[j];
    };

    if (!xcd_barrier(ctl, want, deadline)) return;                         // registration: all `want` workers are here (or nobody goes on)
    if (threadIdx.x == 0 && worker == 0) atomicAdd(&ctl[2], want);
    constexpr bool LEAN = lean_scalars<TARGET, LPC>();
    constexpr bool LEAN_CT = LEAN && lean_always<TARGET>();
    for (uint32_t i = 0; i < n_phases; ++i) {
        PhaseArgs a;
        load_args(i, a);
        // what the host fixes for every batch of this experiment, as constants (the HOT copies of phase_fused_kernel): single GPU, steady state
        a.mode = 0u; trace_set(a, nullptr, nullptr, nullptr); a.pack = nullptr; a.replay = 0u; a.x_next = nullptr; a.adapt_on = 0u; a.stamps = nullptr;
        a.accbits = nullptr; a.lo = 0; a.L.world = 1; a.acc_by_item = 0u; a.n_peers = 0u; a.peer_tab = nullptr; a.plan = nullptr; a.rec_tab = nullptr;
        a.wt = 0u; a.cr_part1 = nullptr; a.accbits_all = nullptr; a.rec_sorted = nullptr;
        if (ALGO == ALGO_DREAM) a.n_cr = 3;
        for (uint32_t base = worker * (uint32_t)CPW; base < a.n_items; base += want * (uint32_t)CPW) {        // (uniform per workgroup)
            const uint32_t w = base + (uint32_t)cw;
            uint32_t c;
            const bool active = resolve_chain(a, w, c);
            // (the target's constants are fetched per sweep, like a launch of the shipped kernel does: held across the barrier code they cost the
            // 4-lanes-per-chain shape 66 spilled registers under the 128-VGPR cap of a 1024-thread workgroup)
            const typename Target<TARGET, LPC, DPL>::Consts tc = Target<TARGET, LPC, DPL>::load(q, a.L.dim, a.tparams);
            Work<DPL> wk;
            wk.item = w;
            wk.pos_own = a.upd_off + (active ? w : 0u);
            const uint32_t dim_early = a.L.dim;
            if constexpr (LEAN) {
                const bool lean_early = LEAN_CT || a.lean != 0u;
                auto early = [tc, q, dim_early, lean_early](Work<DPL>& k) { if (lean_early) k.ll_cur = Target<TARGET, LPC, DPL>::eval(k.x, q, dim_early, tc); };
                make_proposal<ALGO, LPC, DPL, NP, (LEAN_CT ? 0 : 2)>(a, c, active, q, cw % 1, s_part, wk, nullptr, nullptr, early);
            } else {
                make_proposal<ALGO, LPC, DPL, NP>(a, c, active, q, cw % 1, s_part, wk, nullptr, nullptr);
            }
            const double ll_prop = Target<TARGET, LPC, DPL>::eval(wk.p, q, a.L.dim, tc);
            finish_update<ALGO, LPC, DPL, (LEAN_CT ? 2 : (LEAN ? 1 : 0))>(a, c, active, q, wk, ll_prop);
        }
        if (!xcd_barrier(ctl, want * (i + 2u), deadline)) return;
    }
}
// the same launch shape with NO work: what the barriers alone cost
__global__ __launch_bounds__(1024) void xcd_barrier_only_kernel(uint32_t n_phases, uint32_t* ctl, uint32_t want, uint32_t xcc_want, unsigned long long timeout_ticks) {
    __shared__ uint32_t s_worker;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
    if (threadIdx.x == 0) { s_worker = 0xFFFFFFFFu; if (xcc == xcc_want) s_worker = atomicAdd(&ctl[0], 1u); }
    __syncthreads();
    if (s_worker >= want) return;
    const unsigned long long deadline = xcd_now() + timeout_ticks;
    for (uint32_t i = 0; i <= n_phases; ++i)
        if (!xcd_barrier(ctl, want * (i + 1u), deadline)) return;
}
#endif   // BPM_EXPERIMENT_XCD

// Replay exchange, receiving side: one work item per position of the half generation's update group; the item of a
// chain that lives on ANOTHER rank and whose owner accepted its update (accbits_all) rebuilds that proposal -- same
// records / draws / arithmetic as the owner's update kernel, hence the same bits -- and writes it into this rank's
// replica.  Partner rows come from the other group, which nobody writes in this half generation; each row of the update
// group is written by exactly one item.  No ln-like, no accept test, no history: those are the owner's.
template <int ALGO, int LPC, int DPL, int NP>
__global__ __launch_bounds__(block_for(LPC)) void phase_replay_kernel(const PhaseArgs a_in) {
    // (what the host fixed for every replay launch as compile-time constants, as in phase_fused_kernel's HOT copies)
    PhaseArgs a_rep;
    if (LPC == WAVE) {
        a_rep = a_in;
        a_rep.replay = 1u; a_rep.adapt_on = 0u; trace_set(a_rep, nullptr, nullptr, nullptr);
        a_rep.pack = nullptr; a_rep.x_next = nullptr; a_rep.stamps = nullptr; a_rep.mode = 1u; a_rep.n_peers = 0u;
    }
    const PhaseArgs& a = (LPC == WAVE) ? a_rep : a_in;
    __shared__ uint32_t s_part[(block_for(LPC) / LPC) * MAX_PARTNERS];
    const int lane = threadIdx.x;
    const int cw = lane / LPC, q = lane % LPC;
    const uint32_t w = blockIdx.x * (block_for(LPC) / LPC) + cw;
    bool active = w < a.n_upd;
    uint32_t pos = a.upd_off + (active ? w : 0u);
    const uint32_t* rec = nullptr;
    uint32_t c;
    if (LPC == WAVE && a.plan) {
        if (!active) return;
        pos = __builtin_amdgcn_readfirstlane(pos);
        rec = a.plan + (uint32_t)(pos * PLAN_WORDS);
        c = rec[0];
    } else {
        c = pos_to_chain(a, pos);
    }
    active = active && (c - a.lo) >= a.L.n_local && a.accbits_all[c] != 0;      // remote, and accepted by its owner
    if (LPC == WAVE && !active) return;
    if (LPC == WAVE) c = __builtin_amdgcn_readfirstlane(c);
    Work<DPL> wk;
    make_proposal<ALGO, LPC, DPL, NP>(a, c, active, q, cw, s_part, wk, rec);
    if (active) store_row<LPC, DPL>(row_ptr(a.L, c), q, a.L.ld, wk.p);
}

// Replay with owner-sorted records (one wavefront per chain): one work item per update of the OTHER ranks (the host knows every
// rank's count).  Item (r, k) = rank r's k-th update, its record at segment offset + k, its accept byte at byte k of rank r's block:
// both sit at addresses the wavefront computes from the argument block, both loads leave together: a wavefront
// whose update was rejected is gone after ONE round trip (round 1: record -> chain id -> accept byte, two), an accepted one has the
// record's partner ids when it learns so.  11.3 -> x us per half generation at 8 ranks (profiles/r02_replay_variants.txt).
#ifndef BPM_REPLAY_WG
#define BPM_REPLAY_WG 256
#endif
constexpr int REPLAY_WG = BPM_REPLAY_WG;     // 4 independent wavefronts per workgroup: the kernel is bound by the DISPATCH of mostly idle workgroups
                                   // (~0.35 ns each: 32768 one-wavefront workgroups = 11 us, what round 1's replay kernel measured)
template <int ALGO, int DPL, int NP>
__global__ __launch_bounds__(REPLAY_WG) void phase_replay_sorted_kernel(const PhaseArgs a_in) {
    constexpr int LPC = WAVE;
    PhaseArgs a_rep = a_in;
    a_rep.replay = 1u; a_rep.adapt_on = 0u; trace_set(a_rep, nullptr, nullptr, nullptr);
    a_rep.pack = nullptr; a_rep.x_next = nullptr; a_rep.stamps = nullptr; a_rep.mode = 1u; a_rep.accbits = nullptr;
    const PhaseArgs& a = a_rep;
    __shared__ uint32_t s_part[(REPLAY_WG / WAVE) * MAX_PARTNERS];
    const int lane = threadIdx.x % WAVE, wv = threadIdx.x / WAVE;
    // grid (largest count of another rank / 4, ranks): wavefront wv of block (b, r) = rank r's (4 b + wv)-th update.  The two segment offsets come out of the
    // argument block by r (a scalar load with a register offset); a search through all offsets cost every wavefront ~150 scalar
    // instructions, and most wavefronts do nothing else (15.1 us per half generation at 8 ranks, against 11.3 for round 1's kernel)
    const uint32_t r = blockIdx.y, k = blockIdx.x * (REPLAY_WG / WAVE) + (uint32_t)__builtin_amdgcn_readfirstlane(wv);
    const uint32_t lo = a_in.seg_off[r], hi = a_in.seg_off[r + 1];
    if (r == a_in.seg_me || k >= hi - lo) return;                  // own segment: rows are in place
    const uint32_t e = lo + k;
    const uint32_t* rec = a.rec_sorted + (uint32_t)(e * PLAN_WORDS);
    const uint32_t acc = a.accbits_all[(uint64_t)r * a.L.n_local + k];     // rank r's k-th update of this half generation
    uint32_t c = rec[0];
    if (acc == 0u) return;
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    Work<DPL> wk;
    make_proposal<ALGO, LPC, DPL, NP>(a, c, true, lane, wv, s_part, wk, rec);
    store_row<LPC, DPL>(row_ptr(a.L, c), lane, a.L.ld, wk.p);
}

// Host-callback ln_like_fn: proposals out ...  (NP: the pair count as a compile-time constant -- partner ids in registers, one Philox evaluation per lane
// with one wavefront per chain, make_proposal: RL / FAST / QUAD -- or 0: read from a.P.  Same draws, same bits.)
template <int ALGO, int LPC, int DPL, int NP = 0>
__global__ __launch_bounds__(block_for(LPC)) void phase_propose_kernel(const PhaseArgs a) {
    __shared__ uint32_t s_part[(block_for(LPC) / LPC) * MAX_PARTNERS];
    const int lane = threadIdx.x;
    const int cw = lane / LPC, q = lane % LPC;      // chain slot inside the workgroup, lane inside the chain subgroup
    const uint32_t w = blockIdx.x * (block_for(LPC) / LPC) + cw;
    uint32_t c;
    const bool active = resolve_chain(a, w, c);
    if (w < a.n_items && q == 0) a.ids_buf[w] = active ? (int32_t)c : -1;
    if (LPC == WAVE && !active) return;
    if (LPC == WAVE) c = __builtin_amdgcn_readfirstlane(c);
    Work<DPL> wk;
    make_proposal<ALGO, LPC, DPL, NP>(a, c, active, q, cw, s_part, wk);
    if (!active) return;
    store_row<LPC, DPL>(a.prop_buf + (uint64_t)w * a.L.ld, q, a.L.ld, wk.p);
    if (q == 0) {
        a.aux_buf[w] = wk.log_corr;
        // CR statistic travels through the exchange slots directly
        if (ALGO == ALGO_DREAM) {
            const bool gated = a.adapt_on && a.cr_gate;
            *delta_ptr(a.L, c) = gated ? wk.delta : 0.0;
            *cridx_ptr(a.L, c) = gated ? (double)wk.cr_idx : -1.0;
        }
    }
}

// ... and ln_like values back in (aux_buf[n_local + w]).
template <int ALGO, int LPC, int DPL>
__global__ __launch_bounds__(block_for(LPC)) void phase_commit_kernel(const PhaseArgs a) {
    const int lane = threadIdx.x;
    const int cw = lane / LPC, q = lane % LPC;      // chain slot inside the workgroup, lane inside the chain subgroup
    const uint32_t w = blockIdx.x * (block_for(LPC) / LPC) + cw;
    bool active = w < a.n_items;
    int32_t id = active ? a.ids_buf[w] : -1;
    active = active && id >= 0;
    if (LPC == WAVE && !active) return;
    const uint32_t c = active ? (uint32_t)id : a.lo;
    Work<DPL> wk;
    load_row<LPC, DPL>(row_ptr(a.L, c), q, a.L.ld, wk.x);
    load_row<LPC, DPL>(a.prop_buf + (uint64_t)(active ? w : 0u) * a.L.ld, q, a.L.ld, wk.p);
    wk.log_corr = active ? a.aux_buf[w] : 0.0;
    wk.ll_cur = a.ll[c - a.lo];
    wk.acc_prev = a.acc_count[c - a.lo];
    if (ALGO == ALGO_DREAM && a.adapt_on) {
        load_row<LPC, DPL>(a.w_mean + (uint32_t)((c - a.lo) * 2u * a.L.ld), q, a.L.ld, wk.w_mean);
        load_row<LPC, DPL>(a.w_m2 + (uint32_t)((c - a.lo) * 2u * a.L.ld), q, a.L.ld, wk.w_m2);
    }
    const double ll_prop = active ? a.aux_buf[a.L.n_local + w] : 0.0;
    const u32x4 h0 = chain_block(a.seed, c, a.t, SLOT_HDR0);
    wk.acc_hi = h0.z; wk.acc_lo = h0.w;
    wk.delta = 0.0; wk.gamma = 0.0; wk.cr_idx = -1; wk.d_prime = 0; wk.jump = 0; wk.snk = 0; wk.maskbits = 0; wk.item = w; wk.pos_own = 0u;
    // finish_update rewrites the CR slots: carry the values written by the propose kernel
    if (ALGO == ALGO_DREAM && active) {
        wk.delta = *delta_ptr(a.L, c);
        wk.cr_idx = (int)*cridx_ptr(a.L, c);
    }
    PhaseArgs b = a;
    if (ALGO == ALGO_DREAM) b.cr_gate = (wk.cr_idx >= 0) ? 1u : 0u;
    finish_update<ALGO, LPC, DPL>(b, c, active, q, wk, ll_prop);
}

// ---------------------------------------------------------------------------------
// Crossover-probability re-estimation (dream.py:125-140), once per generation: the kernels of "CR reduction" above.
// cr_state: p_cr[MAX_CR] | delta_m[MAX_CR] | n_cr_updates[MAX_CR]
// ---------------------------------------------------------------------------------
// Level 1 from the slots (where the update kernels did not write it themselves): a LANE per position -- the table lookup and the two slot loads of a
// chunk's G1 positions leave together (a thread per chunk walked them in dependent rounds: 10 us at cfg2) -- then every lane of the chunk adds the
// chunk's values in position order, read across lanes with DPP broadcasts (quad_perm / row_newbcast / v_readlane for G1 = 4 / 16 / 64: a ds_bpermute
// per value and position cost 1.7 us), and the chunk's first lane stores.  16, 4 or 1 chunks per wavefront.
constexpr int CR_L1_THREADS = 256;
template <int G1, int J>
__device__ __forceinline__ int cr_bcast_i(int v) {          // lane J of every group of G1 consecutive lanes, to all lanes of the group
    if (G1 == 4) return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xF, 0xF, false);              // quad_perm [J, J, J, J]
    if (G1 == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);            // row_newbcast:J
    return __builtin_amdgcn_readlane(v, J);
}
// lane k of the chunk sums CR value k (and k + 4 where a chunk has only four lanes): one compare and one conditional add per position and lane
// (every lane summing all MAX_CR values compiled to 500 selects and 240 f64 adds per lane: 3.7 us)
template <int G1, int J>
__device__ __forceinline__ void cr_l1_steps(int idx, double dl, int m0, double& sd0, double& sn0, double& sd1, double& sn1) {
    const int ij = cr_bcast_i<G1, J>(idx);
    const double dj = __hiloint2double(cr_bcast_i<G1, J>(__double2hiint(dl)), cr_bcast_i<G1, J>(__double2loint(dl)));
    cr_chunk_add(sd0, sn0, m0, ij, dj);
    if (G1 == 4) cr_chunk_add(sd1, sn1, m0 + 4, ij, dj);
    if constexpr (J + 1 < G1) cr_l1_steps<G1, J + 1>(idx, dl, m0, sd0, sn0, sd1, sn1);             // position order
}
template <int G1>
__global__ __launch_bounds__(CR_L1_THREADS) void cr_level1_kernel(Layout L, PermKey pk, const uint32_t* perm_tab, uint32_t N, uint32_t n_cr, uint32_t n1, double* part1) {
    const uint32_t t = blockIdx.x * CR_L1_THREADS + threadIdx.x;      // = chunk * G1 + k  (a chunk never straddles wavefronts: G1 divides 64)
    const uint32_t ch = t / G1, k = t % G1;
    const uint32_t n_first = (N + 1u) / 2u, c_first = cr_chunks_of(n_first, G1);
    const bool second = ch >= c_first;
    const uint32_t pos = second ? n_first + (ch - c_first) * G1 + k : ch * G1 + k;
    const bool valid = ch < n1 && pos < (second ? N : n_first);
    int idx = -1;
    double dl = 0.0;
    if (valid) {
        const uint32_t c = perm_tab ? perm_tab[pos] : perm_fwd(pos, pk);
        idx = (int)*cridx_ptr(L, c);
        dl = *delta_ptr(L, c);
    }
    static_assert(MAX_CR <= 8, "a chunk's lanes cover the CR values: lane k sums value k (and k + 4 when G1 == 4)");
    double sd0 = 0.0, sn0 = 0.0, sd1 = 0.0, sn1 = 0.0;
    cr_l1_steps<G1, 0>(idx, dl, (int)k, sd0, sn0, sd1, sn1);
    if (ch < n1) {
        if (k < n_cr) {
            __hip_atomic_store(&part1[(uint64_t)k * n1 + ch], sd0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part1[(uint64_t)(MAX_CR + k) * n1 + ch], sn0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (G1 == 4 && k + 4u < n_cr) {
            __hip_atomic_store(&part1[(uint64_t)(k + 4u) * n1 + ch], sd1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part1[(uint64_t)(MAX_CR + k + 4u) * n1 + ch], sn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// Level 2: wavefront g folds level-1 partials [64 g, 64 g + 64) of every array with the DPP tree -> part2[m * n2 + g]
__global__ __launch_bounds__(WAVE) void cr_mid_kernel(const double* part1, uint32_t n1, uint32_t n_cr, uint32_t n2, double* part2) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x, b = g * WAVE + lane;
#pragma unroll
    for (int m = 0; m < MAX_CR; ++m) {
        if (m < (int)n_cr) {                       // uniform
            double d = 0.0, n = 0.0;
            if (b < n1) {
                d = __hip_atomic_load(&part1[(uint64_t)m * n1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                n = __hip_atomic_load(&part1[(uint64_t)(MAX_CR + m) * n1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            d = gsum<WAVE>(d);
            n = gsum<WAVE>(n);
            if (lane == 0) {
                __hip_atomic_store(&part2[(uint64_t)m * n2 + g], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&part2[(uint64_t)(MAX_CR + m) * n2 + g], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
// (Level 2 and the final fold in ONE launch -- the last wavefront of cr_mid_kernel's grid to finish, by an agent-scope ticket, folding the level-2 sums -- was
// built, bit-identical, and measured SLOWER than the two launches: cfg5 burn-in 71.4 vs 69.5 us per generation, its share 16.8 vs 16.8.  The third in-launch
// hand-over of this kind with that sign: a ticket's round trip plus the last wavefront's dependent loads cost more than a small dependent dispatch.)
// totals `tot` + the partial sums of one generation -> cr_state (one wavefront holding ROUNDS partials per lane; tot may be cr_state itself)
template <int ROUNDS>
__global__ __launch_bounds__(WAVE) void cr_final_kernel(const double* tot, const double* part, uint32_t nb, uint32_t n_cr, double* cr_state) {
    CrTotals T;
    cr_fold<ROUNDS>(tot, part, nb, nb, n_cr, T);
    if (threadIdx.x == 0) cr_write_totals(T, n_cr, cr_state);
}

// Sparse exchange, receiving side: after the all-gather of the packed blocks PK[r] = [count | pad | ids[cap] | rows[cap][ld]]
// every rank copies the rows the OTHER ranks accepted into its replica of the state matrix.  Block (slot, r).
// xstat[0] |= 1 when some rank had more accepted rows than capacity, xstat[1] = max count seen (capacity control).
__global__ __launch_bounds__(WAVE) void exchange_scatter_kernel(Layout L, double* PK, uint32_t nsub, uint32_t stride, uint32_t cap,
                                                              uint32_t me, uint32_t* xstat) {
    const uint32_t sb = blockIdx.x / cap, slot = blockIdx.x % cap, r = blockIdx.y;
    double* blk = PK + ((uint64_t)r * nsub + sb) * stride;
    const uint32_t cnt = *reinterpret_cast<const uint32_t*>(blk);
    if (slot == 0 && threadIdx.x == 0) {
        if (cnt > cap) atomicOr(&xstat[0], 1u);
        atomicMax(&xstat[1], cnt);
    }
    if (r == me) {
        // own rows are already in place; re-arm the counter for the next half generation (nobody else reads it here)
        if (slot == 0 && threadIdx.x == 0) *reinterpret_cast<uint32_t*>(blk) = 0u;
        return;
    }
    if (slot >= cnt) return;
    const uint32_t id = (uint32_t)blk[2 + slot];
    const double* src = blk + 2 + cap + (uint64_t)slot * L.ld;
    double* dst = row_ptr(L, id);
    for (uint32_t j = threadIdx.x; j < L.ld; j += WAVE) dst[j] = src[j];
}

// The (empty) kernel the library's own queue ends a session with when packets without a release fence have been dispatched: its own
// packet carries acquire + release, so whatever still waits in an L2 (history rows of the non-temporal stores) is written back the way the
// end of any HIP kernel would (DirectQueue::drain).
__global__ __launch_bounds__(WAVE) void queue_fence_kernel(uint32_t* sink) {
    if (sink && threadIdx.x == 0xffffu) *sink = blockIdx.x;        // (never true: a kernel that is not optimised away)
}

// Push exchange, the cross-rank hand-over between two half generations (and at the head of every bpm_step call): ONE wavefront.
// notify: lane p != me stores `seq` into flag[me] of rank p's control block -- this dispatch sits behind the update kernel whose packet
// released at system scope, so every row that kernel pushed has been performed before the flag; wait: lane p polls flag[p] of THIS
// rank's block until rank p has announced `seq` too.  The next packet of the queue (barrier bit, system-scope acquire) then starts with
// every rank's pushes of the finished half generation in this replica -- the lock-step of the reference's Allgather
// (demc.py:93-94,116-117) without a collective.  Sequence numbers only grow; a peer is never more than one barrier ahead.
// Every wait is bounded by `timeout_ticks` of the 100 MHz constant clock: on expiry err = 1 + p and the lane leaves (the host reports it).
__global__ __launch_bounds__(WAVE) void push_sync_kernel(PushCtrl* mine, const unsigned long long* ctrl_tab, uint32_t world, uint32_t me,
                                                         unsigned long long seq, uint32_t do_notify, uint32_t do_wait,
                                                         unsigned long long timeout_ticks) {
    const uint32_t p = threadIdx.x;
    if (p >= world || p == me) return;
    if (do_notify) {
        PushCtrl* pc = reinterpret_cast<PushCtrl*>(ctrl_tab[p]);
        // relaxed: what this flag announces was performed before this kernel started (the update kernel's packet released at system scope);
        // a release HERE would be a second write-back of every L2
        __hip_atomic_store(&pc->flag[me], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (do_wait) {
        // (a wait of this rank has already run into its limit: the replicas are inconsistent and the host will be told; do not sit out the
        // limit again at every later barrier)
        if (__hip_atomic_load(&mine->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull) return;
        const unsigned long long t0 = wall_clock64();
        // relaxed polls (an acquire load is a load + an invalidate of the caches, in a loop, under the other ranks' running kernels: measured
        // 10 us per hand-over at 2 ranks, 118 us at 8 on one GPU); the acquire is the next packet's fence
        for (;;) {
            const unsigned long long v = __hip_atomic_load(&mine->flag[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v >= PUSH_CLOSING) {      // the peer is destroying its sampler (push_close_kernel): it will never announce `seq`
                __hip_atomic_store(&mine->err, PUSH_ERR_CLOSED + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            if (v >= seq) break;
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(&mine->err, 1ull + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

// Push exchange, teardown (bpm_destroy of a connected rank of a world of processes): lane p announces PUSH_CLOSING in rank p's control block --
// from then on p's barrier kernels do not wait for this rank, they report "closed" -- and, do_wait, polls this rank's own block until p has
// announced the same or `timeout_ticks` have passed (no error then: the caller frees anyway; what a peer still has mapped stays alive through
// its mapping).  Ranks that end a run together therefore unmap each other's buffers only after ALL of them have left their last kernels.
__global__ __launch_bounds__(WAVE) void push_close_kernel(PushCtrl* mine, const unsigned long long* ctrl_tab, uint32_t world, uint32_t me, uint32_t do_wait,
                                                          unsigned long long timeout_ticks) {
    const uint32_t p = threadIdx.x;
    if (p >= world || p == me) return;
    PushCtrl* pc = reinterpret_cast<PushCtrl*>(ctrl_tab[p]);
    __hip_atomic_store(&pc->flag[me], PUSH_CLOSING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!do_wait) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&mine->flag[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < PUSH_CLOSING) {
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// Connection self-test of the ARENA (bpm_push_selftest): what an owner's update kernel does to a peer -- stores into ITS OWN rank block of the
// peer's replica (rows, the (delta, cr) slots at the block's end) and into its block of the peer's outlier buffer -- rehearsed on the first
// and the last row of that block, the block's last two slots and the two ends of the om block, from the same queue and under the same packet
// fences as the real thing.  Block (region, rank b) of 64 lanes; three launches with a cross-rank barrier between them:
//   mode 0  save: what THIS rank's arena holds in the probe regions of every other rank's block (the self-test leaves the state as it found it)
//   mode 1  write: this rank's pattern into its own block's regions of every PEER's arena (system-scope stores, acknowledged before the wave ends)
//   mode 2  verify + restore: this rank's arena must hold rank b's pattern in block b's regions; mismatching words are counted in bad[0]
// A mapping that points somewhere else (a wrong handle) or stores that are not visible behind the hand-over therefore end as *ok = 0 and the
// RCCL exchanges, not as a divergence -- or a fault -- in the first generation.
struct ProbeGeo {
    uint64_t blk;          // doubles per rank block of G
    uint32_t n_local, ld;
    uint32_t om_n;         // doubles per rank block of the om buffer (2 n_local), 0: none
    uint32_t world, me;
    uint32_t save_stride;  // doubles per rank in the save buffer (2 ld + 8)
};
constexpr int PROBE_REGIONS = 5;
__device__ __forceinline__ void probe_region(const ProbeGeo& g, uint32_t b, int r, bool& in_om, uint64_t& off, uint32_t& len, uint32_t& save_off) {
    in_om = r >= 3;
    if (r == 0) { off = (uint64_t)b * g.blk; len = g.ld; save_off = 0; }
    else if (r == 1) { off = (uint64_t)b * g.blk + (uint64_t)(g.n_local - 1u) * g.ld; len = g.ld; save_off = g.ld; }
    else if (r == 2) { off = (uint64_t)b * g.blk + g.blk - 2u; len = 2; save_off = 2u * g.ld; }
    else if (r == 3) { off = (uint64_t)b * g.om_n; len = g.om_n ? 2u : 0u; save_off = 2u * g.ld + 2u; }
    else { off = (uint64_t)b * g.om_n + (g.om_n ? g.om_n - 2u : 0u); len = g.om_n ? 2u : 0u; save_off = 2u * g.ld + 4u; }
}
__device__ __forceinline__ double probe_pattern(unsigned long long seed, uint32_t from, int r, uint32_t j) {
    return (double)(((seed & 0xFFFFull) << 32) + ((unsigned long long)from << 24) + ((unsigned long long)r << 20) + j) + 0.5;
}
__global__ __launch_bounds__(WAVE) void push_arena_probe_kernel(const unsigned long long* tab_all, ProbeGeo g, int mode, unsigned long long seed, double* save,
                                                                unsigned long long* bad) {
    const int r = (int)blockIdx.x;
    const uint32_t b = blockIdx.y, lane = threadIdx.x;
    if (b == g.me || b >= g.world) return;
    bool in_om; uint64_t off; uint32_t len, so;
    // mode 1: block (r, p) writes MY block's region r into peer p's arena; modes 0 / 2: block (r, b) looks at block b's region r of MY arena
    probe_region(g, mode == 1 ? g.me : b, r, in_om, off, len, so);
    if (len == 0) return;
    const uint32_t where = mode == 1 ? b : g.me;
    double* base = reinterpret_cast<double*>(tab_all[(in_om ? 2u * MAX_SEG : 0u) + where]) + off;
    double* sv = save + (uint64_t)b * g.save_stride + so;
    uint32_t n_bad = 0;
    for (uint32_t j = lane; j < len; j += WAVE) {
        if (mode == 0) {
            // (agent-scope store / load of the saved words: the packets of the agent-scope form carry no release fence)
            __hip_atomic_store(sv + j, __hip_atomic_load(base + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (mode == 1) {
            __hip_atomic_store(base + j, probe_pattern(seed, g.me, r, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            const double v = __hip_atomic_load(base + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v != probe_pattern(seed, b, r, j)) ++n_bad;
            __hip_atomic_store(base + j, __hip_atomic_load(sv + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (mode == 2 && n_bad) __hip_atomic_fetch_add(bad, (unsigned long long)n_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (as in finish_update: stores into the peers acknowledged before the wavefront ends)
}

// Push exchange for the rare dense blocks (outlier check: the (omega | ln-like) block of this rank's chains): n doubles from `src` to
// the same offset `off` (in doubles, from the peer buffers' bases in tab) of every other rank.
__global__ void push_copy_kernel(const double* src, const unsigned long long* tab, uint32_t world, uint32_t me, uint64_t off, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = src[i];
    for (uint32_t p = 0; p < world; ++p)
        if (p != me) __hip_atomic_store(reinterpret_cast<double*>(tab[p]) + off + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (as in finish_update: pushes acknowledged before the wavefront ends)
}

// Connection self-test of the push exchange: every rank writes (seed + me) into probe[me] of every other rank's control block.
__global__ __launch_bounds__(WAVE) void push_probe_kernel(const unsigned long long* ctrl_tab, uint32_t world, uint32_t me, unsigned long long seed) {
    const uint32_t p = threadIdx.x;
    if (p >= world || p == me) return;
    PushCtrl* pc = reinterpret_cast<PushCtrl*>(ctrl_tab[p]);
    __hip_atomic_store(&pc->probe[me], seed + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Probe of the memory type the sampler's state is allocated from (sampler.hip: state_memory_is_coherent): launch number `shift` lets
// workgroup b own block (b + shift) mod nb of x, so every block is touched by a different workgroup -- a different XCD, a different
// L2 -- in every launch, and x[i] == number of launches afterwards only if each launch saw what the previous one wrote.
__global__ __launch_bounds__(WAVE) void coherence_probe_kernel(double* x, uint32_t nb, uint32_t shift) {
    const uint32_t b = (blockIdx.x + shift) % nb;
    x[b * WAVE + threadIdx.x] += 1.0;
}

// Shuffle orders of K consecutive generations in one launch: tab[g*N + k] = pi_g(k), inv[g*N + pi_g(k)] = k.
// The update kernels then look partners up (one 4-byte load) instead of walking the Feistel network
// (~100 instructions per id, and the kernels are instruction-issue bound).
constexpr int PERM_CHUNK = 64;
struct PermKeys {
    PermKey k[PERM_CHUNK];
};
__global__ void perm_table_kernel(const PermKeys keys, uint32_t n_gens, uint32_t N, uint32_t* tab, uint32_t* inv) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)n_gens * N) return;
    const uint32_t g = (uint32_t)(e / N), k = (uint32_t)(e % N);
    tab[e] = perm_fwd(k, keys.k[g]);
    // (walking the network backwards costs less than the scattered store inv[g N + c] = k did: 7.5 -> 4.9 us at cfg2; inv == nullptr: nobody reads the
    // inverse -- it serves launches with a work item per LOCAL chain, the ranks of a world -- and half of this kernel's time is saved: 70 -> 35 us per
    // window of 64 generations at cfg5)
    if (inv) inv[e] = perm_inv(k, keys.k[g]);
}

// A history row appended by POSITION (PhaseArgs::hist_by_pos) back into chain order: dst row c = src row k with c = pi_t(k), the shuffle of
// the generation that wrote it (recomputed from its key: the tables of that generation are long gone).  One thread per coordinate pair.
__global__ void hist_unpermute_kernel(const PermKey key, uint32_t N, uint32_t ld, const double* src, const double* llsrc, double* dst, double* lldst) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t np = ld / 2u;
    if (e >= (uint64_t)N * np) return;
    const uint32_t k = (uint32_t)(e / np), p = (uint32_t)(e % np);
    const uint32_t c = perm_fwd(k, key);
    reinterpret_cast<double2*>(dst + (uint64_t)c * ld)[p] = reinterpret_cast<const double2*>(src + (uint64_t)k * ld)[p];
    if (p == 0 && llsrc) lldst[c] = llsrc[k];      // (llsrc == nullptr: this row's ln-like was appended by chain)
}

// Records of K consecutive generations in one launch, one thread per (generation, position in shuffle order):
// everything of an update that depends only on (seed, generation, chain id) -- never on chain states -- so it can
// be drawn ahead of time: the chain id at that position, its header block and its partner chains (dream.py:62-66;
// the pool of a position in the first group is the second group and vice versa, whatever `flip` says about the
// order of the two half generations, demc.py:95-100).  Same Philox blocks, same arithmetic as the in-kernel path.
struct PlanParams {
    uint64_t seed, t0;
    uint32_t n_gens, N, np, snooker;     // np pairs (<= 5); snooker: also the three snooker partners (DE-MC, pools of >= 3)
    uint32_t own_lo, own_n;              // own_n > 0 (push exchange, owner-sorted records): only the records of chains [own_lo, own_lo + own_n)
                                         // are built -- nobody replays another rank's updates, so a rank needs its own run only
};
constexpr int PLAN_THREADS = 256;
__global__ __launch_bounds__(PLAN_THREADS) void plan_kernel(const PlanParams P, const uint32_t* tab, uint32_t* plan, const uint32_t* sidx) {
    // records by position leave through LDS: a thread's 64-byte record as four 16-byte pieces 64 bytes apart makes every store instruction of a
    // wavefront touch 64 lines; transposed, it writes 1 KB in a row
    __shared__ uint4 s_stage[PLAN_THREADS * (PLAN_WORDS / 4)];
    const uint64_t e = (uint64_t)blockIdx.x * PLAN_THREADS + threadIdx.x;
    const uint64_t total = (uint64_t)P.n_gens * P.N;
    const bool valid = e < total;
    if (!valid && sidx) return;
    const uint64_t ev = valid ? e : total - 1u;
    const uint32_t g = (uint32_t)(ev / P.N), pos = (uint32_t)(ev % P.N);
    const uint64_t t = P.t0 + g;
    const uint32_t* tg = tab + (uint64_t)g * P.N;
    const uint32_t n_first = (P.N + 1u) / 2u;
    const bool first = pos < n_first;
    const uint32_t pool_off = first ? n_first : 0u, M = first ? P.N - n_first : n_first;
    const uint32_t c = tg[pos];
    if (sidx && P.own_n && (c - P.own_lo) >= P.own_n) return;       // another rank's chain
    uint32_t r[PLAN_WORDS];
#pragma unroll
    for (int i = 0; i < PLAN_WORDS; ++i) r[i] = 0u;
    const u32x4 h = chain_block(P.seed, c, t, SLOT_HDR0);
    r[0] = c; r[1] = h.x; r[2] = h.y; r[3] = h.z; r[4] = h.w;
#pragma unroll
    for (int p = 0; p < (PLAN_WORDS - 5) / 2; ++p) {
        if ((uint32_t)p < P.np) {
            const u32x4 wb = chain_block(P.seed, c, t, SLOT_PAIR0 + ((uint32_t)p >> 1));
            uint32_t ia, ib;
            distinct_pair((p & 1) ? wb.z : wb.x, (p & 1) ? wb.w : wb.y, M, ia, ib);
            r[5 + 2 * p] = tg[pool_off + ia];
            r[6 + 2 * p] = tg[pool_off + ib];
        }
    }
    if (P.snooker && P.np == 1u && M >= 3u) {          // DE-MC: the same block and arithmetic as make_proposal's in-kernel path
        const u32x4 ws = chain_block(P.seed, c, t, SLOT_SNK);
        uint32_t iz, i1, i2;
        distinct_three(ws.x, ws.y, ws.z, M, iz, i1, i2);
        r[7] = tg[pool_off + iz];
        r[8] = tg[pool_off + i1];
        r[9] = tg[pool_off + i2];
    }
    // where the record goes: by position, or (world > 1) to its slot in the owner-sorted order of its generation
    if (sidx) {
        // (a slot beyond the generation means "never assigned" -- the table is initialised to 0xFFFFFFFF: a window built without its slot pass
        // must not scatter records through memory, which is what the faults of round 3's first push runs were: DESIGN.md section 6)
        const uint32_t slot = sidx[e];
        if (slot >= P.N) return;
        const uint64_t oe = (uint64_t)g * P.N + slot;
        uint4* out = reinterpret_cast<uint4*>(plan + oe * PLAN_WORDS);
#pragma unroll
        for (int i = 0; i < PLAN_WORDS / 4; ++i) out[i] = make_uint4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
        return;
    }
    constexpr int Q = PLAN_WORDS / 4;
#pragma unroll
    for (int i = 0; i < Q; ++i) s_stage[threadIdx.x * Q + i] = make_uint4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * PLAN_THREADS * Q;               // this workgroup's first 16-byte piece
    uint4* out = reinterpret_cast<uint4*>(plan) + base;
    const uint64_t pieces = total * Q;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const uint32_t j = (uint32_t)i * PLAN_THREADS + threadIdx.x;
        if (base + j < pieces) out[j] = s_stage[j];
    }
}

// Owner-sorted order of the update records (world > 1): block (group, generation) walks the group's positions in order and gives
// each one its slot in a STABLE PARTITION BY OWNER RANK (owner = chain id / n_local): sidx[g * N + pos] = group offset + (records of
// lower ranks) + (earlier positions of the same rank); count[(g * 2 + group) * n_rank + r] = how many updates rank r has in that half
// generation.  plan_kernel then writes every record straight to its slot.  Rank r's updates of a half generation are then a
// contiguous run of records (its own update kernel's work list: item k -> record k of the run, no idle items) and "rank r's k-th
// update" names the same chain on every rank (what the replay kernel and the accept bytes are indexed by).  Fixed-order scans: the
// same table on every rank and in every run.  The host reads the counts once per window, one window ahead.
constexpr int PLAN_LOCAL_THREADS = 1024;
__global__ __launch_bounds__(PLAN_LOCAL_THREADS) void plan_slot_kernel(const uint32_t* tab, uint32_t N, uint32_t n_local, uint32_t n_rank,
                                                                       uint32_t* sidx, uint32_t* count) {
    __shared__ uint32_t s_wave[PLAN_LOCAL_THREADS / WAVE][MAX_SEG];
    __shared__ uint32_t s_base[MAX_SEG], s_run[MAX_SEG];
    const uint32_t grp = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1), wv = tid / WAVE;
    const uint32_t n_first = (N + 1u) / 2u;
    const uint32_t off = grp ? n_first : 0u, n = grp ? N - n_first : n_first;
    const uint32_t* tg = tab + (uint64_t)g * N + off;
    const uint32_t magic = (uint32_t)((1ull << 32) / n_local) + 1u;
    for (int pass = 0; pass < 2; ++pass) {                         // pass 0 counts per rank, pass 1 assigns the slots
        if (tid < MAX_SEG) s_run[tid] = 0u;
        __syncthreads();
        for (uint32_t t0 = 0; t0 < n; t0 += PLAN_LOCAL_THREADS) {
            const uint32_t i = t0 + tid;
            const bool valid = i < n;
            uint32_t r = 0xFFFFFFFFu;
            if (valid) {
                const uint32_t c = tg[i];
                r = __umulhi(c, magic);
                r -= (r * n_local > c) ? 1u : 0u;
            }
            uint32_t before = 0;
            for (uint32_t q = 0; q < n_rank; ++q) {
                const unsigned long long b = __ballot(r == q);
                if (r == q) before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                if (lane == 0) s_wave[wv][q] = (uint32_t)__popcll(b);
            }
            __syncthreads();
            if (pass == 1 && valid) {
                uint32_t wave_off = 0;
                for (uint32_t k = 0; k < wv; ++k) wave_off += s_wave[k][r];
                sidx[(uint64_t)g * N + off + i] = off + s_base[r] + s_run[r] + wave_off + before;
            }
            __syncthreads();
            if (tid < n_rank) {
                uint32_t tot = 0;
                for (uint32_t k = 0; k < PLAN_LOCAL_THREADS / WAVE; ++k) tot += s_wave[k][tid];
                s_run[tid] += tot;
            }
            __syncthreads();
        }
        if (pass == 0) {
            if (tid == 0) {
                uint32_t acc = 0;
                for (uint32_t q = 0; q < n_rank; ++q) { s_base[q] = acc; acc += s_run[q]; }
            }
            if (tid < n_rank) count[((uint64_t)g * 2u + grp) * n_rank + tid] = s_run[tid];
            __syncthreads();
        }
    }
}

// The same for ONE rank's chains only (push exchange: a rank launches its own updates and nobody replays anybody's): the slot of a position
// owned by this rank = half offset + number of earlier positions of the half that this rank owns too.  Two launches over (chunk of
// SLOT_CHUNK positions, half, generation) -- thousands of workgroups where plan_slot_kernel has two per generation (82 us per window
// at 8 x 8192 chains): count per chunk, then every chunk adds the counts of the chunks before it.  Positions of other ranks get no slot
// (plan_kernel skips them); count[(g * 2 + half) * n_rank + me] = this rank's updates, the other ranks' entries stay zero.
constexpr int SLOT_CHUNK = 1024;
__global__ __launch_bounds__(SLOT_CHUNK) void plan_slot_own_kernel(const uint32_t* tab, uint32_t N, uint32_t own_lo, uint32_t own_n, uint32_t n_rank,
                                                                   uint32_t me, uint32_t n_chunks, uint32_t* chunk_count, uint32_t* sidx, uint32_t* count,
                                                                   uint32_t assign) {
    __shared__ uint32_t s_wave[SLOT_CHUNK / WAVE];
    __shared__ uint32_t s_before;
    const uint32_t chunk = blockIdx.x, grp = blockIdx.y, g = blockIdx.z, tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1), wv = tid / WAVE;
    const uint32_t n_first = (N + 1u) / 2u;
    const uint32_t off = grp ? n_first : 0u, n = grp ? N - n_first : n_first;
    const uint32_t i = chunk * SLOT_CHUNK + tid;
    const bool mine = i < n && (tab[(uint64_t)g * N + off + i] - own_lo) < own_n;
    const unsigned long long b = __ballot(mine);
    if (lane == 0) s_wave[wv] = (uint32_t)__popcll(b);
    uint32_t* cc = chunk_count + ((uint64_t)g * 2u + grp) * n_chunks;
    __syncthreads();
    if (!assign) {
        if (tid == 0) {
            uint32_t tot = 0;
            for (int k = 0; k < SLOT_CHUNK / WAVE; ++k) tot += s_wave[k];
            cc[chunk] = tot;
        }
        return;
    }
    if (tid == 0) {                                     // chunks before this one, in order (at most a few hundred values)
        uint32_t acc = 0;
        for (uint32_t k = 0; k < chunk; ++k) acc += cc[k];
        s_before = acc;
        if (chunk == n_chunks - 1u) {                   // the half's total: this rank's launch size
            uint32_t tot = acc;
            for (int k = 0; k < SLOT_CHUNK / WAVE; ++k) tot += s_wave[k];
            count[((uint64_t)g * 2u + grp) * n_rank + me] = tot;
        }
    }
    __syncthreads();
    if (mine) {
        uint32_t wave_off = 0;
        for (uint32_t k = 0; k < wv; ++k) wave_off += s_wave[k];
        sidx[(uint64_t)g * N + off + i] = off + s_before + wave_off + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    }
}

// ---------------------------------------------------------------------------------
// DREAM outlier-chain reset (Vrugt et al. 2009; NOT in the reference -- extension), entirely on the device:
//   outlier_omega_kernel   omega_i = mean ln_like of chain i over the last half of its history, this rank's chains
//   [all-gather of the (omega | ln_like) blocks when world > 1]
//   outlier_select_kernel  Q1 / Q3 order statistics of omega over all N chains (radix select), first maximum
//   outlier_reset_kernel   chains with omega < Q1 - 2 IQR restart from the best chain's state
// No host round trip: round 1 copied omega to the host, selected the quartiles there (4.6 ms per check at N = 262144)
// and synchronised the stream three times per check.
// OM layout: rank r's block at OM + r * 2 * n_local = [omega (n_local) | ln_like (n_local)].
// ---------------------------------------------------------------------------------
// ln-like rows appended by POSITION (PhaseArgs::hist_by_pos == 1) are read where they lie: keys[g] is the shuffle of the generation that wrote row g
// (identity for a row in chain order), chain i's entry sits at position pi_g^-1(i).  That gather is slow -- every XCD pulls nearly the whole row
// through its L2: 335 us per check at cfg5 with 90 rows in the window -- so while the check is due every few generations the update kernels append
// the ln-like by chain (hist_by_pos == 2: one scattered 8-byte store per update) and keys is nullptr: coalesced reads.
__global__ void outlier_omega_kernel(const double* llhist, const double* ll, uint32_t n_local, uint32_t r0, uint32_t rows, const PermKey* keys, double* om_block) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    double acc = 0.0;
    uint32_t cnt = 0;
    constexpr uint32_t UNR = 4;                     // (the scattered loads of 4 rows in flight; summed in row order all the same)
    for (uint32_t g0 = r0; g0 < rows; g0 += UNR) {  // rows whose ln_like is unknown (NaN: warm start with a host callback) do not count
        double v[UNR];
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u) {
            const uint32_t g = g0 + u < rows ? g0 + u : rows - 1u;
            const uint32_t p = keys ? perm_inv(i, keys[g]) : i;
            v[u] = llhist[(uint64_t)g * n_local + p];
        }
#pragma unroll
        for (uint32_t u = 0; u < UNR; ++u)
            if (g0 + u < rows && v[u] == v[u]) { acc += v[u]; ++cnt; }
    }
    om_block[i] = cnt ? acc / (double)cnt : acc / 0.0;
    om_block[n_local + i] = ll[i];
}

__device__ __forceinline__ double om_at(const double* OM, uint32_t n_local, uint32_t e) {      // omega of global chain e
    const uint32_t r = e / n_local;
    return OM[(uint64_t)r * 2u * n_local + (e - r * n_local)];
}
// order-preserving map double -> uint64 (ascending)
__device__ __forceinline__ unsigned long long f64_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
constexpr int SEL_THREADS = 1024;
constexpr int SEL_UNR = 4;              // elements per thread and pass
inline uint32_t sel_blocks(uint32_t N) { return (N + SEL_THREADS * SEL_UNR - 1) / (SEL_THREADS * SEL_UNR); }      // workgroups of one pass
struct SelRanks { uint32_t k[4]; };      // 0-based order statistics: floor / ceil positions of the 25th and 75th percentile
// Radix-select state in global memory: per target the key prefix fixed so far and the rank left inside it; the histogram of
// the pass in flight ([4][256], zero between passes) and the launch's ticket follow it.
struct SelState {
    unsigned long long prefix[4];
    uint32_t krem[4];
    uint32_t ticket, pad;
    uint32_t ghist[4 * 256];
};
// One PASS (8 bits of the 64-bit keys, most significant first) of the radix select of the four order statistics, as one launch
// of N / (1024 * SEL_UNR) workgroups: each histograms its slice in LDS (targets whose prefixes still coincide share one
// histogram) and adds the non-empty bins to the global histogram; the LAST workgroup to finish (ticket) picks every
// target's bin, extends its prefix, and clears histogram and ticket for the next pass.  After pass 7 sel[b] is the value.
// Pass 0 also finds the first maximum (np.argmax) -> sel[4]: per-workgroup (value, first index) pairs behind the state, folded by the last workgroup
// (one extra workgroup walking all N values took 65 us at N = 262144).
// (Round 2's first version ran all eight passes in ONE workgroup per target: 1.26 ms per check at N = 262144, five CUs busy.)
__global__ __launch_bounds__(SEL_THREADS) void outlier_select_pass_kernel(const double* OM, uint32_t n_local, uint32_t N, int pass, SelRanks R,
                                                                         SelState* st, double* sel) {
    __shared__ uint32_t hist[4][256];
    __shared__ uint32_t s_last;
    __shared__ double s_v[SEL_THREADS];
    __shared__ uint32_t s_i[SEL_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_hist_blocks = gridDim.x;
    // the first maximum (pass 0): every workgroup folds its slice into (value, first index), the last one folds those
    double* pmax_v = reinterpret_cast<double*>(st + 1);
    uint32_t* pmax_i = reinterpret_cast<uint32_t*>(pmax_v + gridDim.x);
    auto fold_first_max = [&](double best, uint32_t bi) {      // -> s_v[0], s_i[0]; ties: the lower index
        s_v[tid] = best; s_i[tid] = bi;
        __syncthreads();
        for (uint32_t o = SEL_THREADS / 2; o > 0; o >>= 1) {
            if (tid < o) {
                const double vb = s_v[tid + o];
                const uint32_t ib = s_i[tid + o];
                const double va = s_v[tid];
                const uint32_t ia = s_i[tid];
                const bool take = ib != 0xFFFFFFFFu && (ia == 0xFFFFFFFFu || vb > va || (vb == va && ib < ia));
                if (take) { s_v[tid] = vb; s_i[tid] = ib; }
            }
            __syncthreads();
        }
    };
    const int shift = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    unsigned long long prefix[4];
    int grp[4];                       // target b counts in histogram grp[b] (the first target with the same prefix)
#pragma unroll
    for (int b = 0; b < 4; ++b) prefix[b] = pass == 0 ? 0ull : st->prefix[b];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        grp[b] = b;
#pragma unroll
        for (int a = b - 1; a >= 0; --a) if (prefix[a] == prefix[b]) grp[b] = a;
    }
    for (uint32_t i = tid; i < 4u * 256u; i += SEL_THREADS) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (SEL_THREADS * SEL_UNR);
    unsigned long long keys[SEL_UNR];
    {
        double best = 0.0;
        uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < SEL_UNR; ++u) {
            const uint32_t e = base + (uint32_t)u * SEL_THREADS + tid;
            const double v = e < N ? om_at(OM, n_local, e) : 0.0;
            keys[u] = e < N ? f64_key(v) : 0ull;
            if (pass == 0 && e < N && (bi == 0xFFFFFFFFu || v > best)) { best = v; bi = e; }      // ascending e per thread: keeps the first maximum
        }
        if (pass == 0) {
            fold_first_max(best, bi);
            if (tid == 0) {
                __hip_atomic_store(&pmax_v[blockIdx.x], s_v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&pmax_i[blockIdx.x], s_i[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int u = 0; u < SEL_UNR; ++u) {
        const uint32_t e = base + (uint32_t)u * SEL_THREADS + tid;
        const unsigned long long key = keys[u];
        const uint32_t digit = (uint32_t)(key >> shift) & 0xFFu;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (grp[b] != b) continue;                                   // uniform: shares an earlier target's histogram
            bool todo = e < N && (key & himask) == prefix[b];
            // LDS atomics of one wavefront on ONE address retire one lane at a time, and the leading bytes of a population's
            // values are nearly all equal: up to four rounds of "first pending lane's digit, counted by a ballot" take the
            // dominant digits out with one atomic each; what is left is spread over many bins
#pragma unroll 1
            for (int round = 0; round < 4; ++round) {
                const unsigned long long pend = __ballot(todo);
                if (pend == 0ull) break;
                const int first = __ffsll((long long)pend) - 1;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, first);
                const unsigned long long same = __ballot(todo && digit == d0);
                if ((int)(tid & (WAVE - 1)) == first) atomicAdd(&hist[b][d0], (uint32_t)__popcll(same));
                if (digit == d0) todo = false;
            }
            if (todo) atomicAdd(&hist[b][digit], 1u);
        }
    }
    __syncthreads();
    {   // non-empty bins to the global histogram (thread = bin, four histograms)
        const uint32_t b = tid >> 8, bin = tid & 255u;
        const uint32_t c = hist[b][bin];
        if (c) __hip_atomic_fetch_add(&st->ghist[b * 256u + bin], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every wavefront waits for ITS atomics to be performed before the barrier: __syncthreads() alone is a workgroup-scope
    // fence and leaves them in flight (seen: the ticket passed them, the last workgroup read an incomplete histogram)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {       // (agent-scope atomics are performed at the memory side: no cache-wide fence needed)
        const uint32_t t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == n_hist_blocks - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup: every thread fetches its bin of all four global histograms; threads 0..3 then walk one target each
    {
        const uint32_t b = tid >> 8, bin = tid & 255u;
        hist[b][bin] = __hip_atomic_load(&st->ghist[b * 256u + bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->ghist[b * 256u + bin], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < 4u * WAVE) {
        // wavefront b walks target b's histogram: lane l owns bins 4l .. 4l+3, an inclusive scan over the lanes finds the lane,
        // then the bin, that holds rank k (a serial walk of 256 LDS reads by one thread cost 10 us per pass)
        const int b = (int)(tid / WAVE), lane = (int)(tid % WAVE);
        uint32_t k = pass == 0 ? R.k[b] : st->krem[b];
        int g = b;
        unsigned long long pf = 0ull;
#pragma unroll
        for (int a = 0; a < 4; ++a) if (a == b) { g = grp[a]; pf = prefix[a]; }
        const uint32_t* h = hist[g];
        const uint32_t c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
        uint32_t incl = c0 + c1 + c2 + c3;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const uint32_t excl = incl - (c0 + c1 + c2 + c3);
        // first lane whose inclusive count exceeds k (k is below the total, so there is one; the last lane otherwise)
        const unsigned long long over = __ballot(k < incl);
        const int owner = over ? __ffsll((long long)over) - 1 : WAVE - 1;
        if (lane == owner) {
            uint32_t kk = k - excl, bin = 4u * (uint32_t)lane;
            if (kk >= c0) { kk -= c0; ++bin; if (kk >= c1) { kk -= c1; ++bin; if (kk >= c2) { kk -= c2; ++bin; } } }
            const unsigned long long np = pf | ((unsigned long long)bin << shift);
            st->prefix[b] = np;
            st->krem[b] = kk;
            if (pass == 7) sel[b] = key_f64(np);
        }
    }
    if (pass == 0) {
        __syncthreads();
        double best = 0.0;
        uint32_t bi = 0xFFFFFFFFu;
        for (uint32_t k = tid; k < gridDim.x; k += SEL_THREADS) {      // (ascending slices per thread: a later slice wins only with a larger value)
            const double v = __hip_atomic_load(&pmax_v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t i = __hip_atomic_load(&pmax_i[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || v > best)) { best = v; bi = i; }
        }
        fold_first_max(best, bi);
        if (tid == 0) sel[4] = (double)s_i[0];
    }
    if (tid == 0) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// np.percentile's linear interpolation between the order statistics a <= b at fraction t (NumPy's _lerp)
__device__ __forceinline__ double np_lerp(double a, double b, double t) {
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}
// One wavefront tests 64 chains (lane = chain); for each outlier among them (rare) the whole wavefront copies the best chain's
// row into the replica -- every rank does, for all N chains -- and the chain's OWNER also fixes the ln_like cache and the ln-like of the last
// history row, and lists the chain for outlier_rebuild_kernel (list[0] = count, then local chain indices).
__global__ __launch_bounds__(WAVE) void outlier_reset_kernel(Layout L, uint32_t N, uint32_t lo, const double* OM, const double* sel, double t1, double t3,
                                                             double* ll, double* llhist, uint32_t rows, const PermKey* llkeys, uint32_t* list,
                                                             unsigned long long* n_resets) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c0 = blockIdx.x * WAVE + lane;
    const double q1 = np_lerp(sel[0], sel[1], t1), q3 = np_lerp(sel[2], sel[3], t3);
    const double cut = q1 - 2.0 * (q3 - q1);
    const uint32_t best = (uint32_t)sel[4];
    const bool out = c0 < N && om_at(OM, L.n_local, c0) < cut;
    unsigned long long m = __ballot(out);
    if (m == 0ull) return;
    const uint32_t rb = best / L.n_local;
    const double ll_best = OM[(uint64_t)rb * 2u * L.n_local + L.n_local + (best - rb * L.n_local)];
    const double* src = row_ptr(L, best);
    while (m) {
        const uint32_t b = (uint32_t)__ffsll((long long)m) - 1u;
        m &= m - 1ull;
        const uint32_t c = blockIdx.x * WAVE + b;
        double* dst = row_ptr(L, c);
        for (uint32_t j = lane; j < L.ld; j += WAVE) dst[j] = src[j];
        if (lane == 0) {
            const uint32_t li = c - lo;
            if (li < L.n_local) {
                ll[li] = ll_best;
                if (llhist) llhist[(uint64_t)(rows - 1) * L.n_local + (llkeys ? perm_inv(li, llkeys[rows - 1]) : li)] = ll_best;
                if (list) list[1u + atomicAdd(&list[0], 1u)] = li;
            }
            atomicAdd(n_resets, 1ull);
        }
    }
}
// The owner's history of a chain that was reset: its last row (= the current state) becomes the best chain's row, and the chain's Welford
// moments are rebuilt over its rows in history order -- the very operations the running update applied, so the chains that were not
// reset keep moments that equal a full rebuild bit for bit.  One wavefront per listed chain (a kernel of its own: inside the reset kernel the
// rebuilds of a 64-chain block's outliers ran one after the other, 0.2-0.7 ms per check at cfg5).  The loads of 64 / JW rows x JW columns
// are in flight together (lane = (row, column)); row g holds the chain at position pi_g^-1(li) when it was appended by position (keys, see
// outlier_omega_kernel); the updates then run in row order on values handed over by lane.
__global__ __launch_bounds__(WAVE) void outlier_rebuild_kernel(Layout L, const double* sel, const uint32_t* list, double* hist, uint32_t rows,
                                                               const PermKey* keys, double* w_mean, double* w_m2) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n = list[0];
    const uint32_t ld = L.ld;
    const uint32_t JW = ld >= (uint32_t)WAVE ? (uint32_t)WAVE : (ld <= 1u ? 1u : 1u << (32 - __clz((int)(ld - 1u))));
    const uint32_t RW = (uint32_t)WAVE / JW;
    const uint32_t jj = lane % JW, rsub = lane / JW;
    const double* src = row_ptr(L, (uint32_t)sel[4]);
    const uint64_t stride = (uint64_t)L.n_local * ld;
    constexpr uint32_t UNR = 4;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t li = list[1u + i];
        const uint32_t p_last = keys ? perm_inv(li, keys[rows - 1]) : li;
        for (uint32_t jb = 0; jb < ld; jb += JW) {
            const uint32_t j = jb + jj;
            const bool valid = j < ld;
            const double v = valid ? src[j] : 0.0;
            if (valid && rsub == 0u) hist[(uint64_t)(rows - 1) * stride + (uint64_t)p_last * ld + j] = v;
            if (!w_mean) continue;
            double mean = 0.0, m2 = 0.0;                        // this chain's moments over its rows [0, rows), last row = v
            for (uint32_t g0 = 0; g0 < rows; g0 += RW * UNR) {
                double x[UNR];
#pragma unroll
                for (uint32_t u = 0; u < UNR; ++u) {
                    const uint32_t g = g0 + u * RW + rsub;
                    x[u] = v;
                    if (valid && g + 1u < rows) {
                        const uint32_t pg = keys ? perm_inv(li, keys[g]) : li;
                        x[u] = hist[(uint64_t)g * stride + (uint64_t)pg * ld + j];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < UNR; ++u) {
#pragma unroll 1
                    for (uint32_t r = 0; r < RW; ++r) {
                        const uint32_t g = g0 + u * RW + r;
                        const double xx = __shfl(x[u], (int)(r * JW + jj));
                        if (g < rows) {
                            const double d1 = xx - mean;
                            mean = mean + d1 / (double)(g + 1u);
                            m2 = m2 + d1 * (xx - mean);
                        }
                    }
                }
            }
            if (valid && rsub == 0u) {
                w_mean[(uint64_t)li * 2u * ld + j] = mean;
                w_m2[(uint64_t)li * 2u * ld + j] = m2;
            }
        }
    }
}

// Welford moments of every local chain's history rows [0, rows) recomputed from the history
// buffer (only needed when adaptation resumes after generations run without it).
__global__ void welford_rebuild_kernel(const double* hist, uint64_t row_stride, uint64_t n_elem, uint32_t rows, uint32_t ld,
                                       double* w_mean, double* w_m2) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem) return;
    const uint64_t we = (e / ld) * 2u * ld + e % ld;      // (one record per chain: [mean | m2])
    double mean = 0.0, m2 = 0.0;
    for (uint32_t g = 0; g < rows; ++g) {
        const double v = hist[(uint64_t)g * row_stride + e];
        const double d1 = v - mean;
        mean = mean + d1 / (double)(g + 1);
        m2 = m2 + d1 * (v - mean);
    }
    w_mean[we] = mean;
    w_m2[we] = m2;
}

// chain.py:25-27: state0 = theta_0 + N(0, diag(varepsilon)) for the local block of the exchange buffer
__global__ void init_jitter_kernel(Layout L, uint32_t lo, uint64_t seed, const double* theta0, const double* var,
                                   int jitter) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)L.n_local * L.ld) return;
    const uint32_t i = (uint32_t)(e / L.ld), j = (uint32_t)(e % L.ld);
    double v = 0.0;
    if (j < L.dim) {
        v = theta0[j];
        if (jitter) {
            const u32x4 w = chain_block(seed, lo + i, T_INIT, SLOT_DIM0 + j);
            v = v + sqrt(var[j]) * box_muller(w.z, w.w);
        }
    }
    row_ptr(L, lo + i)[j] = v;
}

// ln_like of n points (row stride ld) with the device target
template <int TARGET, int LPC, int DPL>
__global__ __launch_bounds__(block_for(LPC)) void eval_ll_kernel(const double* X, uint32_t n, uint32_t ld, uint32_t dim,
                                                      const double* tparams, double* out) {
    const int lane = threadIdx.x;
    const int cw = lane / LPC, q = lane % LPC;      // chain slot inside the workgroup, lane inside the chain subgroup
    const uint32_t w = blockIdx.x * (block_for(LPC) / LPC) + cw;
    const bool active = w < n;
    double v[DPL];
    load_row<LPC, DPL>(X + (uint64_t)(active ? w : 0u) * ld, q, ld, v);
    const double ll = Target<TARGET, LPC, DPL>::eval(v, q, dim, Target<TARGET, LPC, DPL>::load(q, dim, tparams));
    if (active && q == 0) out[w] = ll;
}

// param_est (demc.py:235-248) on the device: per-dimension sum and sum of squares (about `shift`)
// over the flat history rows [m_lo, m_hi) of the (rows x ld) matrix H.  Deterministic two-stage
// reduction: per-block partials, then a fixed-order sum over blocks.
constexpr int MOM_THREADS = 256;
constexpr int MOM_UNR = 8;              // independent 16-byte loads in flight per thread
// A thread owns one coordinate PAIR (columns 2p, 2p + 1; ld is even, rows are 16-byte aligned) of every rpi-th row of its
// block's slice: one 16-byte load per row, a row's ld / 2 loads contiguous across the threads (round 1 had 8-byte loads, one
// column per thread, at most 1024 blocks: 3.2 TB/s on cfg2's 6.8 GB history).  Per-block partial sums, then
// moments_final_kernel adds the blocks in a fixed order: deterministic for a given grid.
__global__ __launch_bounds__(MOM_THREADS) void moments_partial_kernel(const double* H, uint64_t m_lo, uint64_t m_hi,
                                                                    uint32_t ld, const double* shift, double* part) {
    __shared__ double s_a[2 * MOM_THREADS], s_b[2 * MOM_THREADS];
    const uint32_t np = ld / 2u;                                  // pairs per row
    const uint32_t ppp = np <= MOM_THREADS ? np : MOM_THREADS;   // pairs per pass over the columns
    const uint32_t rpi = MOM_THREADS / ppp;                      // rows per iteration
    const uint64_t M = m_hi - m_lo;
    const uint64_t rpb = (M + gridDim.x - 1) / gridDim.x;
    const uint64_t b0 = m_lo + (uint64_t)blockIdx.x * rpb;
    const uint64_t b1 = b0 + rpb < m_hi ? b0 + rpb : m_hi;
    for (uint32_t p0 = 0; p0 < np; p0 += ppp) {
        const uint32_t p = p0 + threadIdx.x % ppp, r = threadIdx.x / ppp;
        double sa0 = 0.0, sa1 = 0.0, sb0 = 0.0, sb1 = 0.0;
        if (r < rpi && p < np) {
            const double shx = shift[2u * p], shy = shift[2u * p + 1u];      // (two scalars: as a double2 the pair lived in scratch memory across the loop)
            for (uint64_t m = b0 + r; m < b1; m += (uint64_t)rpi * MOM_UNR) {
                double2 v[MOM_UNR];
#pragma unroll
                for (int u = 0; u < MOM_UNR; ++u) {
                    const uint64_t mm = m + (uint64_t)u * rpi;
                    v[u] = make_double2(shx, shy);
                    if (mm < b1) v[u] = reinterpret_cast<const double2*>(H + mm * ld)[p];
                }
#pragma unroll
                for (int u = 0; u < MOM_UNR; ++u) {
                    const double d0 = v[u].x - shx, d1 = v[u].y - shy;
                    sa0 += d0; sb0 += d0 * d0;
                    sa1 += d1; sb1 += d1 * d1;
                }
            }
        }
        s_a[2 * threadIdx.x] = sa0; s_a[2 * threadIdx.x + 1] = sa1;
        s_b[2 * threadIdx.x] = sb0; s_b[2 * threadIdx.x + 1] = sb1;
        __syncthreads();
        if (r == 0 && p < np) {
            for (uint32_t rr = 1; rr < rpi; ++rr) {              // the block's row slots in order
                const uint32_t o = 2u * (rr * ppp + threadIdx.x);
                sa0 += s_a[o]; sa1 += s_a[o + 1]; sb0 += s_b[o]; sb1 += s_b[o + 1];
            }
            double* pa = part + ((uint64_t)blockIdx.x * 2 + 0) * ld + 2u * p;
            double* pb = part + ((uint64_t)blockIdx.x * 2 + 1) * ld + 2u * p;
            pa[0] = sa0; pa[1] = sa1; pb[0] = sb0; pb[1] = sb1;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(MOM_THREADS) void moments_final_kernel(const double* part, uint32_t nblocks, uint32_t ld, double* out) {
    // one block per column: strided partial sums, then a fixed-order tree
    __shared__ double s_a[MOM_THREADS], s_b[MOM_THREADS];
    const uint32_t j = blockIdx.x;
    double sa = 0.0, sb = 0.0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += MOM_THREADS) {
        sa += part[((uint64_t)b * 2 + 0) * ld + j];
        sb += part[((uint64_t)b * 2 + 1) * ld + j];
    }
    s_a[threadIdx.x] = sa; s_b[threadIdx.x] = sb;
    __syncthreads();
    for (int o = MOM_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_a[threadIdx.x] += s_a[threadIdx.x + o]; s_b[threadIdx.x] += s_b[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[j] = s_a[0]; out[ld + j] = s_b[0]; }
}

}  // inline namespace BPM_VARIANT_NS
}  // namespace bpm
