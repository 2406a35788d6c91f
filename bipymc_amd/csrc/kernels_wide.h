// Wide rows (d > 512): the update kernel as a LOOP over the row -- no dimension limit, no scratch memory (round 4).
//
// The register-resident kernels of kernels.h hold a chain's whole row in registers (DPL coordinates per lane): fast up to d = 512 (0.9 of the roof), slower at d <= 1024
// (16 coordinates per lane, 256 VGPRs + AGPRs), spilling ~1 KB per lane to scratch memory at 32 per lane (0.23-0.29 of the HBM roof,
// HIP-stream launches only) and impossible beyond 2048, where round 3's bpm_create gave up.  The reference has no limit
// (bipymc/dream.py:52-58,61,85-89 work on self.dim).  Here ONE WAVEFRONT walks its chain's row in chunks of 64 lanes x WIDE_CP coordinate
// pairs; everything that is a sum over the row is a running reduction, everything that is a draw is an ADDRESSED Philox block
// (chain id, generation, slot 8 + pair index) and can be drawn again in a later pass instead of being kept:
//
//   pass A  (DREAM, CR < 1)   the coordinate blocks' mask uniforms only: d' = number of coordinates in the crossover subspace
//                             (dream.py:52-57) -> gamma = gamma_scale 2.38 / sqrt(2 P d') (dream.py:61).  No row is touched.
//   pass S  (snooker update)  |x - z|^2 and (z1 - z2).(x - z) over the four rows (ter Braak & Vrugt 2008) -> the jump along x - z.
//   pass B                    own row + 2 P partner rows streamed chunk by chunk, the same blocks drawn again, proposal x', the target's
//                             sums (ln_like in O(d)), the CR statistic (dream.py:119-130); the history row of this generation is
//                             written SPECULATIVELY with the current state x -- five updates in six are rejected (cfg2: 17 % accepted),
//                             and x is in registers here anyway.
//   [Metropolis test, samplers.py:328-336: wavefront-uniform]
//   pass C  (accepted updates, and every update during CR adaptation / in the synchronous mode)
//                             the proposal is built once more (same blocks, same partner rows, same operations in the same order: the same
//                             bits) and stored: state row (write-through where the packet carries no release; pushed to the peers' replicas
//                             in a world), history row; Welford moments of the chain's own history during burn-in.
//
// Traffic per update in the steady state, in rows: reads 7 + 0.17 x 7, writes 1 + 0.17 x 2 = 9.5 against SURVEY 8(d)'s algorithmic
// 2 P + 3 = 9 (1.06 x); a proposal stashed in memory between the accept test and the stores would cost 10.2.  Philox: two blocks per
// coordinate pair (pass A + B; one when CR = 1) + 0.17.
//
// Same draws, same arithmetic per coordinate as make_proposal / finish_update (kernels.h) -- the oracle does not know which kernel ran; sums
// over the row are taken per lane in chunk order, then across the wavefront (another order than the register-resident shapes: the ln_like of
// a row differs from theirs in the last bits, as theirs differ from NumPy's; every shape is its own deterministic function).
#pragma once
#include "kernels.h"

namespace bpm {
inline namespace BPM_VARIANT_NS {      // (philox.h: one kernel-symbol namespace per build variant)

constexpr int WIDE_CP = 2;                          // coordinate pairs per lane and chunk: 7 rows x 2 loads of 16 bytes in flight per lane
constexpr int WIDE_CHUNK_PAIRS = WAVE * WIDE_CP;    // 128 pairs = 256 coordinates per chunk

// ---- the targets as running sums over coordinate pairs (same per-coordinate arithmetic as kernels.h: Target<>::eval) -----------------
template <int TARGET>
struct WideTarget {      // TARGET_HOST: nothing to evaluate here (propose stage)
    __device__ __forceinline__ void add(const double*, uint32_t, uint32_t, double, double) {}
    __device__ __forceinline__ double finish(const double*, uint32_t) { return 0.0; }
};
template <>
struct WideTarget<TARGET_GAUSS> {      // utils/d100_gauss.py:14-35; params [rho, c0, a, b, 1/sigma...] (allocation padded by two doubles)
    double s1 = 0.0, s2 = 0.0;
    __device__ __forceinline__ void add(const double* tp, uint32_t pi, uint32_t dim, double v0, double v1) {
        const uint32_t j0 = 2u * pi;
        if (j0 < dim) {
            const double2 is = *reinterpret_cast<const double2*>(tp + 4 + j0);
            const double z0 = v0 * is.x;
            s1 += z0; s2 += z0 * z0;
            if (j0 + 1u < dim) {
                const double z1 = v1 * is.y;
                s1 += z1; s2 += z1 * z1;
            }
        }
    }
    __device__ __forceinline__ double finish(const double* tp, uint32_t) {
        const double t1 = gsum<WAVE>(s1), t2 = gsum<WAVE>(s2);
        return tp[1] - 0.5 * (tp[2] * t2 - tp[3] * t1 * t1);
    }
};
template <>
struct WideTarget<TARGET_MIXTURE> {    // utils/dblgauss_rv.py:11-32, pairwise blocks; params as Target<TARGET_MIXTURE>
    double q0 = 0.0, q1 = 0.0;
    __device__ __forceinline__ void add(const double* tp, uint32_t pi, uint32_t dim, double v0, double v1) {
        if (2u * pi + 1u < dim) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double* p = tp + 2 + 7 * c;
                const double a = (v0 - p[0]) * p[2];
                const double b = (v1 - p[1]) * p[3];
                const double qq = (a * a - 2.0 * p[4] * a * b + b * b) * p[5];
                if (c == 0) q0 += qq; else q1 += qq;
            }
        }
    }
    __device__ __forceinline__ double finish(const double* tp, uint32_t dim) {
        const double t0 = gsum<WAVE>(q0), t1 = gsum<WAVE>(q1);
        const double npairs = (double)(dim / 2);
        const double c0 = tp[0] + npairs * tp[8] - 0.5 * t0;
        const double c1 = tp[1] + npairs * tp[15] - 0.5 * t1;
        const double m = fmax(c0, c1);
        return m + log(exp(c0 - m) + exp(c1 - m));
    }
};

__device__ __forceinline__ double2 wide_load2(const double* row, uint32_t pi, bool valid) {
    double2 t = make_double2(0.0, 0.0);
    if (valid) t = reinterpret_cast<const double2*>(row)[pi];
    return t;
}
__device__ __forceinline__ void wide_store2_stream(double* row, uint32_t pi, double v0, double v1) {
    bpm_d2v t = {v0, v1};
    __builtin_nontemporal_store(t, reinterpret_cast<bpm_d2v*>(row) + pi);
}
__device__ __forceinline__ void wide_store2_wt16(double* row, uint32_t pi, double v0, double v1) {
    bpm_d2v t = {v0, v1};
    bpm_d2v* p = reinterpret_cast<bpm_d2v*>(row) + pi;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}

// What pass B and pass C share: the proposal of one coordinate pair.  Uniform inputs in U, the pair's Philox block in wj.
struct WideUni {
    uint32_t dim, ld, P;
    uint32_t thr, forced;          // DREAM: mask threshold, forced coordinate (dim: none)
    double gamma;                  // DREAM: gamma(d') or 1 (jump); DE-MC: gamma or 1
    double epsilon, u_epsilon;
    int snk;                       // DE-MC: this update is a snooker update
    double gs;                     // snooker: gamma_s (z1 - z2).(x - z) / |x - z|^2
};
template <int ALGO, int NP>
__device__ __forceinline__ void wide_pair_proposal(const PhaseArgs& a, const WideUni& U, uint32_t c, uint32_t pi, bool valid, double2 x, uint32_t mine,
                                                   double& p0, double& p1, uint32_t& mbits) {
    constexpr bool DREAM = ALGO == ALGO_DREAM;
    const uint32_t j0 = 2u * pi;
    const bool two = j0 + 1u < U.dim;
    const u32x4 wj = chain_block(a.seed, c, a.t, SLOT_DIM0 + pi);
    double en0 = 0.0, en1 = 0.0, eu0 = 0.0, eu1 = 0.0;
    mbits = 0u;
    if (valid) {
        if (U.epsilon > 0.0) {                                           // util.py:5-16
            double n0, n1;
            box_muller_pair_f32(wj.z, wj.w, n0, n1);
            en0 = U.epsilon * n0;
            if (two) en1 = U.epsilon * n1;
        }
        if (DREAM) {
            if (U.u_epsilon > 0.0) {                                     // util.py:18-28
                eu0 = -U.u_epsilon + (2.0 * U.u_epsilon) * (((double)(wj.y >> 16) + 0.5) * 1.52587890625e-05);
                if (two) eu1 = -U.u_epsilon + (2.0 * U.u_epsilon) * (((double)(wj.y & 0xFFFFu) + 0.5) * 1.52587890625e-05);
            }
            if ((wj.x >> 16) <= U.thr || j0 == U.forced) mbits |= 1u;                    // dream.py:52-57
            if (two && ((wj.x & 0xFFFFu) <= U.thr || j0 + 1u == U.forced)) mbits |= 2u;
        }
    }
    if (DREAM) {
        double s0 = 0.0, s1 = 0.0;       // sum over pairs of (A_p - B_p), p = 0 first (dream.py:65-68,85-86)
        if (NP > 0) {
            double2 ra[NP > 0 ? NP : 1], rb[NP > 0 ? NP : 1];
#pragma unroll
            for (int p = 0; p < NP; ++p) {                              // all 2 NP row loads in flight together
                ra[p] = wide_load2(row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 2 * p)), pi, valid);
                rb[p] = wide_load2(row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 2 * p + 1)), pi, valid);
            }
            s0 = ra[0].x - rb[0].x; s1 = ra[0].y - rb[0].y;
#pragma unroll
            for (int p = 1; p < NP; ++p) { s0 = s0 + (ra[p].x - rb[p].x); s1 = s1 + (ra[p].y - rb[p].y); }
        } else {
#pragma unroll 1
            for (uint32_t p = 0; p < U.P; ++p) {
                const double2 ra = wide_load2(row_ptr(a.L, (uint32_t)__shfl((int)mine, (int)(2u * p))), pi, valid);
                const double2 rb = wide_load2(row_ptr(a.L, (uint32_t)__shfl((int)mine, (int)(2u * p + 1u))), pi, valid);
                s0 = (p == 0) ? (ra.x - rb.x) : (s0 + (ra.x - rb.x));
                s1 = (p == 0) ? (ra.y - rb.y) : (s1 + (ra.y - rb.y));
            }
        }
        const double jump0 = (1.0 + eu0) * U.gamma * s0 + en0;           // dream.py:85-89
        const double jump1 = (1.0 + eu1) * U.gamma * s1 + en1;
        p0 = (mbits & 1u) ? (jump0 + x.x) : x.x;
        p1 = (mbits & 2u) ? (jump1 + x.y) : x.y;
    } else if (!U.snk) {                                                  // demc.py:161-182
        const double2 ra = wide_load2(row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 0)), pi, valid);
        const double2 rb = wide_load2(row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 1)), pi, valid);
        double q0 = U.gamma * (ra.x - rb.x), q1 = U.gamma * (ra.y - rb.y);
        q0 = q0 + x.x; q1 = q1 + x.y;
        q0 = q0 + en0; q1 = q1 + en1;
        p0 = q0; p1 = q1;
    } else {                                                              // snooker (ter Braak & Vrugt 2008): along x - z
        const double2 rz = wide_load2(row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 2)), pi, valid);
        double q0 = x.x + U.gs * (x.x - rz.x), q1 = x.y + U.gs * (x.y - rz.y);
        q0 = q0 + en0; q1 = q1 + en1;
        p0 = (j0 < U.dim) ? q0 : x.x;
        p1 = two ? q1 : x.y;
    }
}

// STAGE_FUSED: the whole update with a device target.  STAGE_PROPOSE (TARGET_HOST): passes A / S / B with the proposal written to prop_buf
// (host-callback ln_like_fn, samplers.py:36-43); phase_wide_commit_kernel finishes.
template <int ALGO, int TARGET, int NP, int STAGE>
__global__ __launch_bounds__(BPM_BLOCK_WAVE) void phase_wide_kernel(
#ifdef BPM_PRELOAD
    const uint32_t* pl_plan, uint32_t pl_upd_off, uint32_t pl_n_items, uint32_t pl_mode,      // (the argument block of phase_fused_kernel: one launch path)
#endif
    const PhaseArgs a) {
    constexpr bool DREAM = ALGO == ALGO_DREAM;
    constexpr bool FUSED = STAGE == STAGE_FUSED;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE)));
    if (w >= a.n_items) return;
    uint32_t c;
    if (a.mode == 0) {
        c = own_pos_to_chain(a, a.upd_off + w);
    } else {
        c = a.lo + w;
        if (a.mode == 1) {
            const uint32_t pos = chain_to_pos(a, c);
            if ((pos - a.upd_off) >= a.n_upd) {                            // not in this half generation's group
                if (!FUSED && lane == 0) a.ids_buf[w] = -1;
                return;
            }
        }
    }
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    if (!FUSED && lane == 0) a.ids_buf[w] = (int32_t)c;
    const uint32_t li = c - a.lo;
    WideUni U;
    U.dim = a.L.dim; U.ld = a.L.ld; U.P = NP > 0 ? (uint32_t)NP : a.P;
    U.thr = 65536u; U.forced = U.dim; U.epsilon = a.epsilon; U.u_epsilon = a.u_epsilon; U.snk = 0; U.gs = 0.0;
    const uint32_t npairs = (U.dim + 1u) >> 1;
    const uint32_t n_chunks = (npairs + WIDE_CHUNK_PAIRS - 1u) / WIDE_CHUNK_PAIRS;
    const double* xrow = row_ptr(a.L, c);
    // ---- header block, partner chains (one lane per partner: dream.py:62-66 / demc.py:169; snooker: three more)
    const u32x4 h0 = chain_block(a.seed, c, a.t, SLOT_HDR0);
    const bool snk_possible = !DREAM && a.p_snooker > 0.0 && a.M >= 3;
    const uint32_t npart = 2u * U.P + (snk_possible ? 3u : 0u);
    uint32_t mine = 0;
    if (lane < npart) mine = pos_to_chain(a, a.pool_off + partner_pos(a, c, lane, 2u * U.P));
    double ll_cur = 0.0;
    uint32_t acc_prev = 0u;
    if (FUSED) { ll_cur = a.ll[li]; acc_prev = a.acc_count[li]; }
    const double u_sel = (double)(h0.x >> 16) * 1.52587890625e-05;
    const double u_gam = (double)(h0.x & 0xFFFFu) * 1.52587890625e-05;
    int cr_idx = -1, d_prime = (int)U.dim, jump = 0;
    // ---- DREAM: CR value, pass A, gamma
    if (DREAM) {
        double cum = 0.0;
        int idx = (int)a.n_cr - 1;
        bool found = false;
#pragma unroll
        for (int m = 0; m < MAX_CR; ++m) {               // cr ~ Categorical(CR, p_cr): first m with u < cumsum(p_cr)[m]  (dream.py:51)
            if (m < (int)a.n_cr) {
                cum += a.cr_state[m];
                if (!found && u_sel < cum) { idx = m; found = true; }
            }
        }
        cr_idx = idx;
#pragma unroll
        for (int m = 0; m < MAX_CR; ++m) if (m == idx) U.thr = a.thr[m];
        int cnt = (int)U.dim;
        if (U.thr < 65536u) {                             // (CR = 1: every 16-bit uniform passes, d' = d without a draw)
            cnt = 0;
#pragma unroll 1
            for (uint32_t k = 0; k < n_chunks; ++k) {
#pragma unroll
                for (int u = 0; u < WIDE_CP; ++u) {
                    const uint32_t pi = k * WIDE_CHUNK_PAIRS + (uint32_t)u * WAVE + lane;
                    const uint32_t j0 = 2u * pi;
                    bool b0 = false, b1 = false;
                    if (j0 < U.dim) {
                        const u32x4 wj = chain_block(a.seed, c, a.t, SLOT_DIM0 + pi);
                        b0 = (wj.x >> 16) <= U.thr;
                        b1 = (j0 + 1u < U.dim) && (wj.x & 0xFFFFu) <= U.thr;
                    }
                    cnt += (int)__popcll(__ballot(b0)) + (int)__popcll(__ballot(b1));
                }
            }
        }
        if (cnt == 0) { U.forced = mulhi32(h0.y, U.dim); cnt = 1; }      // dream.py:55-57
        d_prime = cnt;
        double gamma = a.gamma_tab[cnt];                   // gamma_scale 2.38 / sqrt(2 P d')  (dream.py:61)
        if (a.k % 5 == 0 && !(u_gam < 0.2)) { gamma = 1.0; jump = 1; }   // dream.py:77-80
        U.gamma = gamma;
    } else {
        double gamma = a.gamma_demc;
        if (a.mode != 2 && a.k % 10 == 0 && !(u_gam < 0.1)) { gamma = 1.0; jump = 1; }   // demc.py:174-177
        U.gamma = gamma;
    }
    // ---- snooker: pass S
    double sn_n2 = 0.0;
    if (snk_possible && u_sel < a.p_snooker) {
        const double* zrow = row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 2));
        const double* r1row = row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 3));
        const double* r2row = row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 4));
        double n2 = 0.0, dot = 0.0;
#pragma unroll 1
        for (uint32_t k = 0; k < n_chunks; ++k) {
#pragma unroll
            for (int u = 0; u < WIDE_CP; ++u) {
                const uint32_t pi = k * WIDE_CHUNK_PAIRS + (uint32_t)u * WAVE + lane;
                const bool valid = 2u * pi < U.dim;
                const double2 x = wide_load2(xrow, pi, valid), z = wide_load2(zrow, pi, valid), r1 = wide_load2(r1row, pi, valid), r2 = wide_load2(r2row, pi, valid);
                const double d0 = x.x - z.x, d1 = x.y - z.y;
                n2 += d0 * d0; dot += (r1.x - r2.x) * d0;
                n2 += d1 * d1; dot += (r1.y - r2.y) * d1;
            }
        }
        n2 = gsum<WAVE>(n2);
        dot = gsum<WAVE>(dot);
        if (n2 > 0.0) { U.snk = 1; U.gs = (1.2 + u01_32(h0.y)) * (dot / n2); sn_n2 = n2; }
    }
    // ---- pass B
    WideTarget<TARGET> T;
    double dl = 0.0, n2p = 0.0;
    const bool cr_stat = DREAM && a.adapt_on && a.cr_gate;
    const double* m2row = a.w_m2 + (uint32_t)(li * 2u * U.ld);
    double* hrow = (FUSED && a.hist_row) ? a.hist_row + (uint32_t)(li * U.ld) : nullptr;
    double* prow = FUSED ? nullptr : a.prop_buf + (uint64_t)w * U.ld;
    const double* zrow_b = U.snk ? row_ptr(a.L, (uint32_t)__builtin_amdgcn_readlane((int)mine, 2)) : xrow;
#pragma unroll 1
    for (uint32_t k = 0; k < n_chunks; ++k) {
#pragma unroll
        for (int u = 0; u < WIDE_CP; ++u) {
            const uint32_t pi = k * WIDE_CHUNK_PAIRS + (uint32_t)u * WAVE + lane;
            const bool valid = 2u * pi < U.dim;
            const double2 x = wide_load2(xrow, pi, valid);
            double2 m2 = make_double2(0.0, 0.0);
            if (cr_stat) m2 = wide_load2(m2row, pi, valid);
            double p0, p1;
            uint32_t mb;
            wide_pair_proposal<ALGO, NP>(a, U, c, pi, valid, x, mine, p0, p1, mb);
            if (FUSED) T.add(a.tparams, pi, U.dim, p0, p1);
            if (cr_stat && valid) {                              // dream.py:119-130, one division per coordinate as in make_proposal
                const double f0 = x.x - p0, f1 = x.y - p1;
                dl += m2.x > 0.0 ? (f0 * f0 * (double)a.hist_len) / m2.x : (f0 * f0) / 1e-24;
                if (2u * pi + 1u < U.dim) dl += m2.y > 0.0 ? (f1 * f1 * (double)a.hist_len) / m2.y : (f1 * f1) / 1e-24;
            }
            if (U.snk) {
                const double2 z = wide_load2(zrow_b, pi, valid);
                const double e0 = p0 - z.x, e1 = p1 - z.y;
                n2p += e0 * e0; n2p += e1 * e1;
            }
            if (valid) {
                if (hrow) wide_store2_stream(hrow, pi, x.x, x.y);           // speculative: the update will most likely be rejected
                if (prow) reinterpret_cast<double2*>(prow)[pi] = make_double2(p0, p1);
                if (trace_mask_of(a)) {
                    const uint32_t j0 = 2u * pi;
                    trace_mask_of(a)[(uint64_t)li * U.dim + j0] = (uint8_t)(mb & 1u);
                    if (j0 + 1u < U.dim) trace_mask_of(a)[(uint64_t)li * U.dim + j0 + 1u] = (uint8_t)((mb >> 1) & 1u);
                }
            }
        }
    }
    const double delta = cr_stat ? gsum<WAVE>(dl) : 0.0;
    double log_corr = 0.0;
    if (U.snk) {
        n2p = gsum<WAVE>(n2p);
        log_corr = 0.5 * (double)(U.dim - 1u) * (log(n2p) - log(sn_n2));
    }
    if (trace_i32_of(a)) {
        int32_t* tr = trace_i32_of(a) + (uint64_t)li * TRACE_I32;
        if (lane == 0) { tr[0] = cr_idx; tr[1] = d_prime; tr[2] = jump; tr[4] = U.snk; }
        if (lane < (uint32_t)MAX_PARTNERS) tr[5 + lane] = lane < npart ? (int32_t)mine : -1;      // (lane i resolved partner i)
    }
    if (!FUSED) {
        if (lane == 0) {
            a.aux_buf[w] = log_corr;
            if (DREAM) {
                *delta_ptr(a.L, c) = cr_stat ? delta : 0.0;
                *cridx_ptr(a.L, c) = cr_stat ? (double)cr_idx : -1.0;
            }
        }
        return;
    }
    // ---- Metropolis (samplers.py:328-336): everything here is wavefront-uniform
    const double ll_prop = T.finish(a.tparams, U.dim);
    double alpha = exp((ll_prop + log_corr) - ll_cur);
    const bool is_nan = alpha != alpha;
    alpha = fmin(1.0, alpha);
    alpha = fmax(0.0, alpha);
    const bool accepted = __builtin_amdgcn_readfirstlane((!is_nan && (u01_53(h0.z, h0.w) < alpha)) ? 1 : 0) != 0;
    const double new_ll = accepted ? ll_prop : ll_cur;
    if (lane == 0) {
        if (accepted) {
            if (a.wt) __hip_atomic_store(&a.acc_count[li], acc_prev + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else a.acc_count[li] = acc_prev + 1u;
            if (a.wt) __hip_atomic_store(&a.ll[li], new_ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else a.ll[li] = new_ll;
        }
        if (is_nan) atomicAdd(&a.counters[2], 1ull);
        if (a.llhist_row) a.llhist_row[li] = new_ll;
        if (DREAM && a.adapt_on) {      // (the CR slots are read by the reduction of an adapting generation only: kernels.h finish_update)
            if (a.wt) {
                __hip_atomic_store(delta_ptr(a.L, c), cr_stat ? delta : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cridx_ptr(a.L, c), cr_stat ? (double)cr_idx : -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *delta_ptr(a.L, c) = cr_stat ? delta : 0.0;
                *cridx_ptr(a.L, c) = cr_stat ? (double)cr_idx : -1.0;
            }
            if (a.n_peers && a.adapt_on && a.cr_gate) {              // push exchange: the slots of every update travel during CR adaptation
                const uint32_t d_off = (uint32_t)(delta_ptr(a.L, c) - a.L.G), c_off = (uint32_t)(cridx_ptr(a.L, c) - a.L.G);
#pragma unroll 1
                for (uint32_t p = 0; p < a.n_peers; ++p) {
                    double* pg = reinterpret_cast<double*>(a.peer_tab[p]);
                    __hip_atomic_store(pg + d_off, delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pg + c_off, (double)cr_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if (trace_i32_of(a)) {
            trace_i32_of(a)[(uint64_t)li * TRACE_I32 + 3] = accepted ? 1 : 0;
            double* tf = trace_f64_of(a) + (uint64_t)li * TRACE_F64;
            tf[0] = alpha; tf[1] = ll_prop; tf[2] = delta; tf[3] = U.gamma;
        }
    }
    // ---- pass C
    const bool welford = DREAM && a.adapt_on;
    if (accepted || welford || a.x_next) {
        double* srow = row_ptr(a.L, c);
        double* nrow = a.x_next ? a.x_next + (uint32_t)(li * U.ld) : nullptr;
        double* wm = a.w_mean + (uint32_t)(li * 2u * U.ld);
        double* w2 = a.w_m2 + (uint32_t)(li * 2u * U.ld);
        const uint32_t row_off = (uint32_t)(srow - a.L.G);
        const double cntp = (double)(a.hist_len + 1);
#pragma unroll 1
        for (uint32_t k = 0; k < n_chunks; ++k) {
#pragma unroll
            for (int u = 0; u < WIDE_CP; ++u) {
                const uint32_t pi = k * WIDE_CHUNK_PAIRS + (uint32_t)u * WAVE + lane;
                const bool valid = 2u * pi < U.dim;
                const double2 x = wide_load2(xrow, pi, valid);
                double n0 = x.x, n1 = x.y;
                if (accepted) {
                    uint32_t mb;
                    wide_pair_proposal<ALGO, NP>(a, U, c, pi, valid, x, mine, n0, n1, mb);
                }
                double2 mean = make_double2(0.0, 0.0), m2 = make_double2(0.0, 0.0);
                if (welford) { mean = wide_load2(wm, pi, valid); m2 = wide_load2(w2, pi, valid); }
                if (valid) {
                    if (nrow) {
                        reinterpret_cast<double2*>(nrow)[pi] = make_double2(n0, n1);     // synchronous generation: banked (samplers.py:300-308)
                    } else if (accepted) {
                        if (a.wt == 2u) wide_store2_wt16(srow, pi, n0, n1);
                        else if (a.wt) {
                            __hip_atomic_store(srow + 2u * pi, n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(srow + 2u * pi + 1u, n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else reinterpret_cast<double2*>(srow)[pi] = make_double2(n0, n1);
#pragma unroll 1
                        for (uint32_t p = 0; p < a.n_peers; ++p) {         // push exchange: the owner writes the row into every other replica
                            double* pg = reinterpret_cast<double*>(a.peer_tab[p]) + row_off;
                            __hip_atomic_store(pg + 2u * pi, n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(pg + 2u * pi + 1u, n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                    if (accepted && hrow) wide_store2_stream(hrow, pi, n0, n1);      // (over the speculative copy of x)
                    if (welford) {                                                    // dream.py:128 as running moments, as finish_update
                        const double d0 = n0 - mean.x, d1 = n1 - mean.y;
                        mean.x = mean.x + d0 / cntp; mean.y = mean.y + d1 / cntp;
                        m2.x = m2.x + d0 * (n0 - mean.x); m2.y = m2.y + d1 * (n1 - mean.y);
                        if (a.wt) { wide_store2_wt16(wm, pi, mean.x, mean.y); wide_store2_wt16(w2, pi, m2.x, m2.y); }
                        else { wide_store2_stream(wm, pi, mean.x, mean.y); wide_store2_stream(w2, pi, m2.x, m2.y); }
                    }
                }
            }
        }
    }
    if (a.n_peers) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // pushes acknowledged before the wavefront ends (finish_update)
}

// ... and the ln_like values of the host callback back in (aux_buf[n_local + w]): Metropolis, state, history, Welford moments.
template <int ALGO>
__global__ __launch_bounds__(BPM_BLOCK_WAVE) void phase_wide_commit_kernel(const PhaseArgs a) {
    constexpr bool DREAM = ALGO == ALGO_DREAM;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE)));
    if (w >= a.n_items) return;
    const int32_t id = a.ids_buf[w];
    if (id < 0) return;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(id), li = c - a.lo;
    const uint32_t dim = a.L.dim, ld = a.L.ld, npairs = (dim + 1u) >> 1;
    const uint32_t n_chunks = (npairs + WIDE_CHUNK_PAIRS - 1u) / WIDE_CHUNK_PAIRS;
    const double ll_cur = a.ll[li], ll_prop = a.aux_buf[a.L.n_local + w], log_corr = a.aux_buf[w];
    const uint32_t acc_prev = a.acc_count[li];
    const u32x4 h0 = chain_block(a.seed, c, a.t, SLOT_HDR0);
    double alpha = exp((ll_prop + log_corr) - ll_cur);
    const bool is_nan = alpha != alpha;
    alpha = fmin(1.0, alpha);
    alpha = fmax(0.0, alpha);
    const bool accepted = __builtin_amdgcn_readfirstlane((!is_nan && (u01_53(h0.z, h0.w) < alpha)) ? 1 : 0) != 0;
    const double new_ll = accepted ? ll_prop : ll_cur;
    if (lane == 0) {
        if (accepted) { a.acc_count[li] = acc_prev + 1u; a.ll[li] = new_ll; }
        if (is_nan) atomicAdd(&a.counters[2], 1ull);
        if (a.llhist_row) a.llhist_row[li] = new_ll;
        if (trace_i32_of(a)) {
            trace_i32_of(a)[(uint64_t)li * TRACE_I32 + 3] = accepted ? 1 : 0;
            double* tf = trace_f64_of(a) + (uint64_t)li * TRACE_F64;
            tf[0] = alpha; tf[1] = ll_prop; tf[2] = DREAM ? *delta_ptr(a.L, c) : 0.0; tf[3] = 0.0;
        }
    }
    const bool welford = DREAM && a.adapt_on;
    double* srow = row_ptr(a.L, c);
    const double* prow = a.prop_buf + (uint64_t)w * ld;
    double* hrow = a.hist_row ? a.hist_row + (uint32_t)(li * ld) : nullptr;
    double* nrow = a.x_next ? a.x_next + (uint32_t)(li * ld) : nullptr;
    double* wm = a.w_mean + (uint32_t)(li * 2u * ld);
    double* w2 = a.w_m2 + (uint32_t)(li * 2u * ld);
    const double cntp = (double)(a.hist_len + 1);
#pragma unroll 1
    for (uint32_t k = 0; k < n_chunks; ++k) {
#pragma unroll
        for (int u = 0; u < WIDE_CP; ++u) {
            const uint32_t pi = k * WIDE_CHUNK_PAIRS + (uint32_t)u * WAVE + lane;
            const bool valid = 2u * pi < dim;
            const double2 nv = accepted ? wide_load2(prow, pi, valid) : wide_load2(srow, pi, valid);
            double2 mean = make_double2(0.0, 0.0), m2 = make_double2(0.0, 0.0);
            if (welford) { mean = wide_load2(wm, pi, valid); m2 = wide_load2(w2, pi, valid); }
            if (!valid) continue;
            if (nrow) reinterpret_cast<double2*>(nrow)[pi] = nv;
            else if (accepted) reinterpret_cast<double2*>(srow)[pi] = nv;
            if (hrow) wide_store2_stream(hrow, pi, nv.x, nv.y);
            if (welford) {
                const double d0 = nv.x - mean.x, d1 = nv.y - mean.y;
                mean.x = mean.x + d0 / cntp; mean.y = mean.y + d1 / cntp;
                m2.x = m2.x + d0 * (nv.x - mean.x); m2.y = m2.y + d1 * (nv.y - mean.y);
                wide_store2_stream(wm, pi, mean.x, mean.y);
                wide_store2_stream(w2, pi, m2.x, m2.y);
            }
        }
    }
}

// ln_like of n points with the device target, one wavefront per row, the sums in the order the update kernel takes them
template <int TARGET>
__global__ __launch_bounds__(BPM_BLOCK_WAVE) void eval_ll_wide_kernel(const double* X, uint32_t n, uint32_t ld, uint32_t dim, const double* tparams, double* out) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE)));
    if (w >= n) return;
    const double* row = X + (uint64_t)w * ld;
    const uint32_t npairs = (dim + 1u) >> 1, n_chunks = (npairs + WIDE_CHUNK_PAIRS - 1u) / WIDE_CHUNK_PAIRS;
    WideTarget<TARGET> T;
#pragma unroll 1
    for (uint32_t k = 0; k < n_chunks; ++k) {
#pragma unroll
        for (int u = 0; u < WIDE_CP; ++u) {
            const uint32_t pi = k * WIDE_CHUNK_PAIRS + (uint32_t)u * WAVE + lane;
            const double2 x = wide_load2(row, pi, 2u * pi < dim);
            T.add(tparams, pi, dim, x.x, x.y);
        }
    }
    const double ll = T.finish(tparams, dim);
    if (lane == 0) out[w] = ll;
}

}  // inline namespace BPM_VARIANT_NS
}  // namespace bpm
