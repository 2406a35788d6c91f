// Microbenchmark (diagnostic, never part of the product): can the kernel boundary between half generations be
// replaced by per-chain version flags inside ONE launch that covers many generations?
//
// Model of the sampler's dependency structure: N chains, every chain updated once per generation, in two groups
// (positions [0, N/2) then [N/2, N) of a per-generation permutation); an update reads its own row and 6 rows of the
// OTHER group -- as they are after the other group's update of the same generation (second group) or before it
// (first group) -- and writes a new row.  Rows are write-once (row (g, c) = chain c after g updates: the sampler's
// history buffer has exactly this shape), so the only hazards are read-after-write.
//
//   boundary : 2 launches per generation (what the sampler does today)
//   dataflow : one launch, workgroup id = (generation, group, position) in dependency order; a workgroup spins
//              (bounded) until version[chain] of the 7 chains it reads has reached the update it needs.  Forward
//              progress relies on workgroups being dispatched in id order (every dependency has a smaller id, so the
//              smallest unfinished workgroup is always resident and never blocked); a wave that spins longer than
//              SPIN_MAX raises the abort flag and everybody leaves.
// Row data and versions move with agent-scope (sc1) loads / stores: XCD L2s are not coherent with each other.
// The host replays the same recurrence and compares bit for bit (a stale read shows up as a mismatch).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr unsigned SPIN_MAX = 1u << 20;
constexpr int DIM = 100;

struct Params {
    double* hist;          // [(G + 1) * N * ld]
    unsigned* version;     // [N] completed updates of each chain
    unsigned* abort_flag;
    unsigned* max_spins;   // diagnostic
    uint32_t N, ld, g0, n_gens, half_first;
    uint32_t mul[64], add[64];      // per-generation permutation position -> chain: (mul * pos + add) mod N, N a power of two
};

__host__ __device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t chain_at(const Params& P, uint32_t g, uint32_t pos) { return (P.mul[g] * pos + P.add[g]) & (P.N - 1u); }
// partner i of the update at (g, pos): a position in the other group
__host__ __device__ inline uint32_t partner_pos(const Params& P, uint32_t g, uint32_t pos, int i) {
    const uint32_t half = P.N / 2u;
    const uint32_t r = mix(mix(g * 0x9e3779b9u + pos) + (uint32_t)i * 0x85ebca6bu) % half;
    return pos < half ? half + r : r;
}

#ifdef PLAIN_ACCESS      // boundary mode only: ordinary cached loads / stores, as the sampler's kernels use (dataflow results are then invalid)
__device__ __forceinline__ double ld_agent(const double* p) { return *p; }
__device__ __forceinline__ void st_agent(double* p, double v) { *p = v; }
#else
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

template <bool DATAFLOW>
__global__ __launch_bounds__(64) void update_kernel(const Params P, uint32_t g_fixed, uint32_t group_fixed) {
    const uint32_t half = P.N / 2u;
    uint32_t g, pos;
    if (DATAFLOW) {
        g = blockIdx.x / P.N;
        pos = blockIdx.x % P.N;
    } else {
        g = g_fixed;
        pos = group_fixed * half + blockIdx.x;
    }
    const uint32_t lane = threadIdx.x;
    const uint32_t c = chain_at(P, g, pos);
    const bool second = pos >= half;
    uint32_t pc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pc[i] = chain_at(P, g, partner_pos(P, g, pos, i));
    const uint32_t need_p = second ? g + 1u : g;          // partners: after their update of this generation, or before it
    if (DATAFLOW) {
        // lanes 0..6 poll one version word each
        const uint32_t who = lane == 0 ? c : pc[(lane - 1u) % 6u];
        const uint32_t need = lane == 0 ? g : need_p;
        unsigned spins = 0;
        bool ok = lane > 6;
        while (true) {
            if (!ok) ok = __hip_atomic_load(P.version + who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.g0 + need;
            if (__all(ok)) break;
            // (the abort flag is ONE address: polled by thousands of waves on every spin it would serialise them)
            if (++spins > SPIN_MAX || ((spins & 255u) == 0u && __hip_atomic_load(P.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(P.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0 && spins > 2000u) atomicMax(P.max_spins, spins);        // diagnostic, rare (same-address atomics serialise)
        asm volatile("" ::: "memory");
    }
    const size_t row = (size_t)P.N * P.ld;
    const double* own = P.hist + (size_t)g * row + (size_t)c * P.ld;
    double x[2] = {0.0, 0.0}, s[2] = {0.0, 0.0};
    const uint32_t j0 = 2u * lane;
    if (j0 < (uint32_t)DIM) {
        x[0] = ld_agent(own + j0); x[1] = ld_agent(own + j0 + 1);
#pragma unroll
        for (int i = 0; i < 6; i += 2) {
            const double* a = P.hist + (size_t)need_p * row + (size_t)pc[i] * P.ld;
            const double* b = P.hist + (size_t)need_p * row + (size_t)pc[i + 1] * P.ld;
            s[0] += ld_agent(a + j0) - ld_agent(b + j0);
            s[1] += ld_agent(a + j0 + 1) - ld_agent(b + j0 + 1);
        }
    }
    // a wave-wide reduction and a data-dependent decision, like the ln-like + accept step
    double q = s[0] * s[0] + s[1] * s[1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const bool accept = (mix((uint32_t)(__double_as_longlong(q) >> 20) + c + g) & 3u) != 0u;
    double* out = P.hist + (size_t)(g + 1u) * row + (size_t)c * P.ld;
    if (j0 < (uint32_t)DIM) {
        st_agent(out + j0, accept ? x[0] * 0.5 + 0.25 * s[0] : x[0]);
        st_agent(out + j0 + 1, accept ? x[1] * 0.5 + 0.25 * s[1] : x[1]);
    }
    if (DATAFLOW) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the row is performed at agent scope before the version says so
        if (lane == 0) __hip_atomic_store(P.version + c, P.g0 + g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 8192u;
    const uint32_t G = argc > 2 ? (uint32_t)atoi(argv[2]) : 32u;
    const uint32_t ld = argc > 3 ? (uint32_t)atoi(argv[3]) : 112u;      // 112 doubles = 7 x 128 B: rows do not share cache lines
    if (N & (N - 1u) || G > 64u || ld < (uint32_t)DIM) { printf("N must be a power of two, G <= 64, ld >= 100\n"); return 1; }
    const size_t row = (size_t)N * ld;
    Params P{};
    CK(hipMalloc(&P.hist, (G + 1) * row * sizeof(double)));
    CK(hipMalloc(&P.version, N * sizeof(unsigned)));
    CK(hipMalloc(&P.abort_flag, sizeof(unsigned)));
    CK(hipMalloc(&P.max_spins, sizeof(unsigned)));
    P.N = N; P.ld = ld; P.g0 = 0; P.n_gens = G;
    for (uint32_t g = 0; g < 64; ++g) { P.mul[g] = (mix(g + 1u) | 1u); P.add[g] = mix(g + 77u); }
    std::vector<double> h0(row, 0.0);
    for (uint32_t c = 0; c < N; ++c) for (int j = 0; j < DIM; ++j) h0[(size_t)c * ld + j] = (double)(mix(c * 131u + j) & 0xFFFFu) / 65536.0 - 0.5;
    // host reference of the recurrence
    std::vector<double> ref((G + 1) * row, 0.0);
    std::copy(h0.begin(), h0.end(), ref.begin());
    for (uint32_t g = 0; g < G; ++g) {
        for (uint32_t pos = 0; pos < N; ++pos) {
            const uint32_t c = chain_at(P, g, pos);
            const uint32_t need = pos >= N / 2 ? g + 1 : g;
            double q = 0.0;
            double s[DIM], x[DIM];
            for (int j = 0; j < DIM; ++j) { s[j] = 0.0; x[j] = ref[(size_t)g * row + (size_t)c * ld + j]; }
            for (int i = 0; i < 6; i += 2) {
                const uint32_t a = chain_at(P, g, partner_pos(P, g, pos, i)), b = chain_at(P, g, partner_pos(P, g, pos, i + 1));
                for (int j = 0; j < DIM; ++j) s[j] += ref[(size_t)need * row + (size_t)a * ld + j] - ref[(size_t)need * row + (size_t)b * ld + j];
            }
            // same reduction tree as the device: per-lane pair sums, then xor butterflies 32,16,...,1
            double lane_q[64];
            for (int l = 0; l < 64; ++l) lane_q[l] = 2 * l < DIM ? s[2 * l] * s[2 * l] + s[2 * l + 1] * s[2 * l + 1] : 0.0;
            for (int o = 32; o > 0; o >>= 1) { double t[64]; for (int l = 0; l < 64; ++l) t[l] = lane_q[l] + lane_q[l ^ o]; for (int l = 0; l < 64; ++l) lane_q[l] = t[l]; }
            q = lane_q[0];
            long long qb; memcpy(&qb, &q, 8);
            const bool accept = (mix((uint32_t)(qb >> 20) + c + g) & 3u) != 0u;
            for (int j = 0; j < DIM; ++j) ref[(size_t)(g + 1) * row + (size_t)c * ld + j] = accept ? x[j] * 0.5 + 0.25 * s[j] : x[j];
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> got((G + 1) * row);
#ifdef PLAIN_ACCESS
    const int n_modes = 1;
#else
    const int n_modes = 2;
#endif
    for (int mode = 0; mode < n_modes; ++mode) {
        float best = 1e30f;
        size_t bad = 0;
        unsigned h_abort = 0, h_spins = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(P.hist, 0, (G + 1) * row * sizeof(double)));
            CK(hipMemcpy(P.hist, h0.data(), row * sizeof(double), hipMemcpyHostToDevice));
            CK(hipMemset(P.version, 0, N * sizeof(unsigned)));
            CK(hipMemset(P.abort_flag, 0, sizeof(unsigned)));
            CK(hipMemset(P.max_spins, 0, sizeof(unsigned)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (mode == 0) {
                for (uint32_t g = 0; g < G; ++g)
                    for (uint32_t grp = 0; grp < 2; ++grp) hipLaunchKernelGGL(update_kernel<false>, dim3(N / 2), dim3(64), 0, 0, P, g, grp);
            } else {
                hipLaunchKernelGGL(update_kernel<true>, dim3(G * N), dim3(64), 0, 0, P, 0u, 0u);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&h_abort, P.abort_flag, sizeof(unsigned), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&h_spins, P.max_spins, sizeof(unsigned), hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), P.hist, (G + 1) * row * sizeof(double), hipMemcpyDeviceToHost));
            bad = 0;
            for (size_t i = 0; i < got.size(); ++i) bad += memcmp(&got[i], &ref[i], 8) != 0;
            if (h_abort || bad) break;
        }
        printf("%-9s N=%u G=%u ld=%u: %.2f us per generation, mismatching doubles %zu, abort %u, longest wait %u polls\n",
               mode == 0 ? "boundary" : "dataflow", N, G, ld, best * 1e3f / G, bad, h_abort, h_spins);
    }
    return 0;
}
