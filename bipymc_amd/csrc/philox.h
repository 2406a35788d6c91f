// Counter-based random-number layer of the MI355X DE-MC/DREAM sampler.
//
// The reference (wgurecky/bipymc) draws every random decision of a chain update
// from NumPy's global MT19937 stream (dream.py:51-84, demc.py:81-86,169-182,
// samplers.py:336).  Here every decision is a pure function of
// (seed, global chain id, absolute generation t, slot): a Philox4x32-10 block
// addressed with rocRAND's (seed, subsequence, offset) convention, so the
// streams are bit-identical to `rocrand_init(seed, chain_id, 4*blk, &s);
// rocrand4(&s)` of <rocrand/rocrand_kernel.h> (checked on device by
// bpm_selftest_philox) but evaluated block-wise in registers, without the
// rocRAND engine's read-ahead block or any generator state in memory.
//
// CPU statement of the same layout: oracle/philox_ref.py (tests only).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#if defined(__HIPCC__)
#define BPM_HD __host__ __device__ __forceinline__
#else
#define BPM_HD inline
#endif

// Every build variant of the library puts its device code into an inline namespace of its own (bpm::product, bpm::hooks, bpm::v_<name>): source code
// is unaffected, but the KERNEL SYMBOLS differ.  The library's own AQL queue locates its kernels BY NAME among the code objects the HSA loader holds
// (aql_queue.h: DirectQueue::kernel), and a process may hold the product and the test variant at once (tests, bench.py's cross-check): with equal names
// the second library dispatched the FIRST library's kernels -- harmless while both were the same code, a memory fault at address 0 as soon as the test
// variant's argument block carried trace fields the product's does not (round 5).
#ifndef BPM_VARIANT_NS
#ifdef BPM_TEST_HOOKS
#define BPM_VARIANT_NS hooks
#else
#define BPM_VARIANT_NS product
#endif
#endif
namespace bpm {
inline namespace BPM_VARIANT_NS {

// ---- draw layout (keep equal to oracle/philox_ref.py) ----------------------
constexpr int SLOT_BITS = 16;
constexpr uint32_t SLOT_HDR0 = 0;   // (select16|gamma16, forced dim [DREAM] / snooker gamma [DE-MC], accept hi, accept lo)
constexpr uint32_t SLOT_PAIR0 = 2;  // two pairs per block: (ia, ib, ia', ib')
constexpr uint32_t SLOT_SNK = 7;    // (iz, i1, i2, spare)
constexpr uint32_t SLOT_DIM0 = 8;   // per coordinate pair pi: (z16|z16, e16|e16, bm1, bm2); init jitter: per coordinate
constexpr int MAX_PAIRS = 10;
constexpr uint64_t SUBSEQ_GLOBAL = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t SLOT_G_FLIP = 0;
constexpr uint32_t SLOT_G_SHUF = 1;  // blocks 1,2: eight round keys (six used)
constexpr uint64_t T_INIT = (1ull << 47) - 1;
constexpr int FEISTEL_ROUNDS = 6;

struct u32x4 {
    uint32_t x, y, z, w;
};

BPM_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

// Philox4x32-10 (Salmon et al., SC'11; Random123 constants).
BPM_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;   // one 32x32->64 multiply each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// rocRAND addressing: key = seed words, counter.xy = block, counter.zw = subsequence.
BPM_HD u32x4 philox_block(uint64_t seed, uint64_t subseq, uint64_t blk) {
    return philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)subseq, (uint32_t)(subseq >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

BPM_HD u32x4 chain_block(uint64_t seed, uint64_t chain_id, uint64_t t, uint32_t slot) {
#if defined(BPM_TEST_HOOKS) && defined(BPM_FAKE_DRAWS) && defined(__HIP_DEVICE_COMPILE__)     // ablation builds only (tools/): what do the draws cost?
    uint32_t x = (uint32_t)chain_id * 0x9e3779b9u + (uint32_t)t * 0x85ebca6bu + slot * 0xc2b2ae35u + (uint32_t)seed;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
    return u32x4{x, x * 0x297a2d39u, x ^ 0x5bd1e995u, x + 0x7f4a7c15u};
#else
    return philox_block(seed, chain_id, (t << SLOT_BITS) | (uint64_t)slot);
#endif
}

BPM_HD u32x4 global_block(uint64_t seed, uint64_t t, uint32_t slot) {
    return chain_block(seed, SUBSEQ_GLOBAL, t, slot);
}

// ---- word -> variate ---------------------------------------------------------
BPM_HD double u01_32(uint32_t w) { return (double)w * 2.3283064365386963e-10; }  // w * 2^-32, exact

BPM_HD double u01_53(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * 1.1102230246251565e-16;  // 2^-53
}

// two distinct positions in [0, m)  (dream.py:66 / demc.py:169 `choice(replace=False, size=2)`)
BPM_HD void distinct_pair(uint32_t wa, uint32_t wb, uint32_t m, uint32_t& ia, uint32_t& ib) {
    ia = mulhi32(wa, m);
    ib = mulhi32(wb, m - 1);
    ib += (ib >= ia) ? 1u : 0u;
}

// three distinct positions in [0, m) for the snooker update
BPM_HD void distinct_three(uint32_t wz, uint32_t w1, uint32_t w2, uint32_t m, uint32_t& iz, uint32_t& i1,
                           uint32_t& i2) {
    iz = mulhi32(wz, m);
    i1 = mulhi32(w1, m - 1);
    i1 += (i1 >= iz) ? 1u : 0u;
    const uint32_t lo = iz < i1 ? iz : i1, hi = iz < i1 ? i1 : iz;
    i2 = mulhi32(w2, m - 2);
    i2 += (i2 >= lo) ? 1u : 0u;
    i2 += (i2 >= hi) ? 1u : 0u;
}

// ---- keyed bijection on [0, n): alternating Feistel + cycle walking --------------
BPM_HD uint32_t hash32(uint32_t x) {  // lowbias32
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

struct PermKey {
    uint32_t k[FEISTEL_ROUNDS];
    uint32_t n;       // domain size
    uint32_t nbits;   // bits of the enclosing power of two (>= 1)
    uint32_t on;      // 0: identity (shuffle=False, demc.py:85)
};

BPM_HD uint32_t perm_nbits(uint32_t n) {
    uint32_t b = 1;
    while (b < 32 && (1u << b) < n) ++b;
    return b;
}

BPM_HD uint32_t feistel_fwd(uint32_t x, const PermKey& pk) {
    uint32_t la = pk.nbits >> 1, lb = pk.nbits - la;
    uint32_t L = x >> lb, R = x & ((1u << lb) - 1u);
#pragma unroll
    for (int i = 0; i < FEISTEL_ROUNDS; ++i) {
        const uint32_t F = hash32(R ^ pk.k[i]) & ((1u << la) - 1u);
        const uint32_t nl = R;
        R = L ^ F; L = nl;
        const uint32_t tmp = la; la = lb; lb = tmp;
    }
    return (L << lb) | R;
}

BPM_HD uint32_t feistel_inv(uint32_t y, const PermKey& pk) {
    const uint32_t a = pk.nbits >> 1, b = pk.nbits - a;
    uint32_t la = (FEISTEL_ROUNDS % 2 == 0) ? a : b, lb = (FEISTEL_ROUNDS % 2 == 0) ? b : a;
    uint32_t L = y >> lb, R = y & ((1u << lb) - 1u);
#pragma unroll
    for (int i = FEISTEL_ROUNDS - 1; i >= 0; --i) {
        const uint32_t Rp = L;
        const uint32_t Lp = R ^ (hash32(Rp ^ pk.k[i]) & ((1u << lb) - 1u));
        L = Lp; R = Rp;
        const uint32_t tmp = la; la = lb; lb = tmp;
    }
    return (L << lb) | R;
}

// pi(x); x < n.  The walk stays on the cycle through x, so it returns below n.
BPM_HD uint32_t perm_fwd(uint32_t x, const PermKey& pk) {
    if (!pk.on) return x;
    x = feistel_fwd(x, pk);
#pragma unroll 1
    while (x >= pk.n) x = feistel_fwd(x, pk);
    return x;
}

BPM_HD uint32_t perm_inv(uint32_t y, const PermKey& pk) {
    if (!pk.on) return y;
    y = feistel_inv(y, pk);
#pragma unroll 1
    while (y >= pk.n) y = feistel_inv(y, pk);
    return y;
}

inline PermKey make_perm_key(uint64_t seed, uint64_t t, uint32_t n, bool shuffle) {
    PermKey pk;
    const u32x4 a = global_block(seed, t, SLOT_G_SHUF), b = global_block(seed, t, SLOT_G_SHUF + 1);
    pk.k[0] = a.x; pk.k[1] = a.y; pk.k[2] = a.z; pk.k[3] = a.w; pk.k[4] = b.x; pk.k[5] = b.y;
    pk.n = n;
    pk.nbits = perm_nbits(n);
    pk.on = shuffle ? 1u : 0u;
    return pk;
}

inline bool flip_draw(uint64_t seed, uint64_t t, double flip_prob) {
    return u01_32(global_block(seed, t, SLOT_G_FLIP).x) < flip_prob;
}

}  // inline namespace BPM_VARIANT_NS
}  // namespace bpm
