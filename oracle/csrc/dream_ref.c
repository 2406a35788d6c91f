/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the DREAM generation of wgurecky/bipymc on the equicorrelated Gaussian
 * (reference: bipymc/dream.py:32-107, bipymc/demc.py:63-151, bipymc/samplers.py:328-336,
 * bipymc/utils/d100_gauss.py:14-35), steady state (no CR adaptation), with the counter-based draw
 * layout of oracle/philox_ref.py.  Purpose: the CPU baseline bench.py times beside the GPU numbers
 * (OpenMP over the chains of a half generation, the only parallelism the algorithm has), and an
 * independent cross-check of oracle/sampler_ref.py (tests/test_oracle_c.py).
 *
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg load the library built from this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint32_t x, y, z, w; } u32x4;

static inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 o = {c0, c1, c2, c3};
    return o;
}
static inline u32x4 chain_block(uint64_t seed, uint64_t chain, uint64_t t, uint32_t slot) {
    const uint64_t blk = (t << 16) | slot;
    return philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)chain, (uint32_t)(chain >> 32), (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
static uint32_t feistel(uint32_t x, uint32_t nbits, const uint32_t* k) {
    uint32_t la = nbits >> 1, lb = nbits - la;
    uint32_t L = x >> lb, R = x & ((1u << lb) - 1u);
    for (int i = 0; i < 6; ++i) {
        const uint32_t F = hash32(R ^ k[i]) & ((1u << la) - 1u), nl = R;
        R = L ^ F; L = nl;
        const uint32_t tmp = la; la = lb; lb = tmp;
    }
    return (L << lb) | R;
}

/* One generation (demc.py:79-134).  X: (N, d) row-major state, ll: (N) cached ln_like, both updated
 * in place; params: [rho, c0, a, b, 1/sigma...]; p_cr: (n_cr).  Returns the number of accepted updates. */
long dream_generation(double* X, double* ll, int N, int d, const double* params, uint64_t seed, uint64_t t, uint32_t k,
                      int P, int n_cr, const double* p_cr, double gamma_scale, double flip_prob, double epsilon,
                      double u_epsilon, uint32_t* order /* scratch N */) {
    const u32x4 gf = chain_block(seed, 0xFFFFFFFFFFFFFFFFull, t, 0);
    const u32x4 ka = chain_block(seed, 0xFFFFFFFFFFFFFFFFull, t, 1), kb = chain_block(seed, 0xFFFFFFFFFFFFFFFFull, t, 2);
    const uint32_t keys[6] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y};
    const int flip = ((double)gf.x * 2.3283064365386963e-10) < flip_prob;
    uint32_t nbits = 1;
    while ((1u << nbits) < (uint32_t)N) ++nbits;
    for (int i = 0; i < N; ++i) {                       /* shuffle order (demc.py:84-86) */
        uint32_t x = feistel((uint32_t)i, nbits, keys);
        while (x >= (uint32_t)N) x = feistel(x, nbits, keys);
        order[i] = x;
    }
    const int n_first = (N + 1) / 2;
    long accepted = 0;
    for (int ph = 0; ph < 2; ++ph) {
        const int first_is_upd = (ph == 0) != (flip != 0);
        const int upd_off = first_is_upd ? 0 : n_first, n_upd = first_is_upd ? n_first : N - n_first;
        const int pool_off = first_is_upd ? n_first : 0, M = first_is_upd ? N - n_first : n_first;
        long acc_ph = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc_ph)
        for (int w = 0; w < n_upd; ++w) {
            const uint32_t c = order[upd_off + w];
            const u32x4 h0 = chain_block(seed, c, t, 0);
            /* CR index (dream.py:51) */
            const double uc = (double)(h0.x >> 16) * 1.52587890625e-05;
            double cum = 0.0;
            int idx = n_cr - 1, found = 0;
            for (int m = 0; m < n_cr; ++m) { cum += p_cr[m]; if (!found && uc < cum) { idx = m; found = 1; } }
            const uint32_t thr = (uint32_t)floor(((double)(idx + 1) / (double)n_cr) * 65536.0);
            double* x = X + (size_t)c * d;
            double prop[512], eu[512], en[512];
            unsigned char mask[512];
            int cnt = 0;
            for (int pi = 0; 2 * pi < d; ++pi) {
                const u32x4 wj = chain_block(seed, c, t, 8 + (uint32_t)pi);
                const float u1 = ((float)wj.z + 1.0f) * 2.3283064365386963e-10f, u2 = (float)wj.w * 2.3283064365386963e-10f;
                const float r = sqrtf(-2.0f * logf(u1)), ang = 6.2831855f * u2;
                const int j0 = 2 * pi, j1 = j0 + 1;
                en[j0] = epsilon * (double)(r * cosf(ang));
                eu[j0] = -u_epsilon + (2.0 * u_epsilon) * (((double)(wj.y >> 16) + 0.5) * 1.52587890625e-05);
                mask[j0] = (wj.x >> 16) <= thr;
                cnt += mask[j0];
                if (j1 < d) {
                    en[j1] = epsilon * (double)(r * sinf(ang));
                    eu[j1] = -u_epsilon + (2.0 * u_epsilon) * (((double)(wj.y & 0xFFFFu) + 0.5) * 1.52587890625e-05);
                    mask[j1] = (wj.x & 0xFFFFu) <= thr;
                    cnt += mask[j1];
                }
            }
            if (cnt == 0) { mask[mulhi32(h0.y, (uint32_t)d)] = 1; cnt = 1; }       /* dream.py:55-57 */
            double gamma = gamma_scale * 2.38 / sqrt(2. * (double)P * (double)cnt); /* dream.py:61 */
            if (k % 5 == 0 && !(((double)(h0.x & 0xFFFFu) * 1.52587890625e-05) < 0.2)) gamma = 1.0;
            const double* A[10];
            const double* B[10];
            for (int p = 0; p < P; ++p) {                                           /* dream.py:65-68 */
                const u32x4 wb = chain_block(seed, c, t, 2 + (uint32_t)(p >> 1));
                const uint32_t wa = (p & 1) ? wb.z : wb.x, wbb = (p & 1) ? wb.w : wb.y;
                uint32_t ia = mulhi32(wa, (uint32_t)M), ib = mulhi32(wbb, (uint32_t)M - 1);
                ib += ib >= ia;
                A[p] = X + (size_t)order[pool_off + ia] * d;
                B[p] = X + (size_t)order[pool_off + ib] * d;
            }
            double s1 = 0.0, s2 = 0.0;
            for (int j = 0; j < d; ++j) {                                           /* dream.py:85-89 */
                double sum = A[0][j] - B[0][j];
                for (int p = 1; p < P; ++p) sum = sum + (A[p][j] - B[p][j]);
                const double jump = (1.0 + eu[j]) * gamma * sum + en[j];
                prop[j] = mask[j] ? jump + x[j] : x[j];
                const double z = prop[j] * params[4 + j];                           /* d100_gauss.py, O(d) form */
                s1 += z;
                s2 += z * z;
            }
            const double ll_prop = params[1] - 0.5 * (params[2] * s2 - params[3] * s1 * s1);
            double alpha = exp(ll_prop - ll[c]);                                    /* samplers.py:328-332 */
            if (alpha > 1.0) alpha = 1.0;
            const double ua = ((double)(h0.z >> 5) * 67108864.0 + (double)(h0.w >> 6)) * 1.1102230246251565e-16;
            if (ua < alpha) {                                                       /* samplers.py:334-336 */
                memcpy(x, prop, (size_t)d * sizeof(double));
                ll[c] = ll_prop;
                ++acc_ph;
            }
        }
        accepted += acc_ph;
    }
    return accepted;
}

/* n_gens generations; optionally appends every generation to hist ((n_gens, N, d)), as chain.py:51-54 does. */
long dream_run(double* X, double* ll, int N, int d, const double* params, uint64_t seed, uint64_t t0, uint32_t k0, int n_gens,
               int P, int n_cr, const double* p_cr, double gamma_scale, double flip_prob, double epsilon, double u_epsilon,
               double* hist, int n_threads) {
    if (d > 512 || P > 10) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    uint32_t* order = (uint32_t*)malloc((size_t)N * sizeof(uint32_t));
    long acc = 0;
    for (int g = 0; g < n_gens; ++g) {
        acc += dream_generation(X, ll, N, d, params, seed, t0 + (uint64_t)g, k0 + (uint32_t)g, P, n_cr, p_cr, gamma_scale, flip_prob,
                                epsilon, u_epsilon, order);
        if (hist) memcpy(hist + (size_t)g * N * d, X, (size_t)N * d * sizeof(double));
    }
    free(order);
    return acc;
}

int dream_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
