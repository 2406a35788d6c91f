// Microbenchmark (diagnostic): the memory pattern of cfg5's update kernel alone -- how fast can 131072 "updates" of a 262144 x 64-byte state matrix
// go when an update is nothing but its memory operations?  Per update (4 lanes, 16 bytes each, like the kernel's 4 lanes per chain):
//   hop 1  one 4-byte read of a position -> chain table (coalesced), 6 more 4-byte reads of partner ids through the same kind of table (random)
//   hop 2  the own 64-byte row and 6 partner rows (random rows of the matrix)
//   out    one 64-byte row appended in position order (streaming store); `acc_pct` percent of the updates also write their own row back (sc1)
// Reported: us per launch (dependent launches on one stream) and rows/s, so that the kernel's 23.7 us per launch at cfg5 can be set against the
// floor of its access pattern.  Variants: rows of 64 bytes at a 64-byte stride (cfg5), the same rows at a 128-byte stride (one row per L2 line:
// does the neighbour's half line cost anything?), and 7 sequential instead of random rows (the bandwidth bound without the randomness).
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/row_gather tools/micro/row_gather.hip && ./build_variants/row_gather
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pool, double* state, double* hist, uint32_t n_upd,
                                                uint32_t stride_d, uint32_t n_rows, int mode, uint32_t acc_thr) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t w = t >> 2, q = t & 3u;
    if (w >= n_upd) return;
    uint32_t c, p[6];
    if (mode == 2) {                       // sequential rows: no table hop, neighbours
        c = w;
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = (w + (uint32_t)(k + 1) * n_upd / 8u) % n_rows;
    } else {
        c = perm[w];
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = pool[(c * 6u + (uint32_t)k) % (6u * n_rows)];      // random ids, dependent on c like a partner lookup through the shuffle table
    }
    d2 own = *reinterpret_cast<const d2*>(state + (uint64_t)c * stride_d + 2u * q);
    d2 r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = *reinterpret_cast<const d2*>(state + (uint64_t)p[k] * stride_d + 2u * q);
    d2 s = own;
#pragma unroll
    for (int k = 0; k < 6; k += 2) s += 0.5 * (r[k] - r[k + 1]);
    __builtin_nontemporal_store(s, reinterpret_cast<d2*>(hist + (uint64_t)w * 8u + 2u * q));
    if ((c * 2654435761u >> 16) < acc_thr) *reinterpret_cast<d2*>(state + (uint64_t)c * stride_d + 2u * q) = s;
}
int main(int argc, char** argv) {
    // row_gather [only_variant [launches]]: one access pattern, few launches -- for a counter pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) on known byte counts
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const int n_timed = argc > 2 ? atoi(argv[2]) : 500;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const uint32_t N = 262144, n_upd = N / 2;
    std::vector<uint32_t> perm(N), pool(6 * (size_t)N);
    uint32_t x = 777u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    for (uint32_t i = 0; i < N; ++i) perm[i] = i;
    for (uint32_t i = N - 1; i > 0; --i) { const uint32_t j = rnd() % (i + 1); std::swap(perm[i], perm[j]); }
    for (auto& e : pool) e = rnd() % N;
    uint32_t *d_perm, *d_pool; double *d_state, *d_hist;
    hipMalloc(&d_perm, N * 4); hipMalloc(&d_pool, pool.size() * 4);
    hipMalloc(&d_state, (size_t)N * 16 * 8); hipMalloc(&d_hist, (size_t)64 * n_upd * 8 * 8);
    hipMemcpy(d_perm, perm.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(d_pool, pool.data(), pool.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_state, 0, (size_t)N * 16 * 8);
    printf("pattern                          accepted  us per launch   rows read per s   bytes of rows per s\n");
    struct V { const char* name; uint32_t stride; int mode; };
    const V vs[] = {{"random 64 B rows, stride 64 B ", 8, 0}, {"random 64 B rows, stride 128 B", 16, 0}, {"sequential rows, stride 64 B  ", 8, 2}};
    int vi = -1;
    for (const V& v : vs)
        for (uint32_t acc_pct : {0u, 6u, 100u}) {
            if (++vi, only >= 0 && vi != only) continue;
            const uint32_t thr = acc_pct * 65536u / 100u;
            int g = 0;
            auto run = [&](int n) {
                for (int i = 0; i < n; ++i, ++g)
                    hipLaunchKernelGGL(k_gather, dim3((n_upd * 4 + 255) / 256), dim3(256), 0, st, d_perm + (g & 1) * n_upd, d_pool, d_state, d_hist + (size_t)(g & 63) * n_upd * 8,
                                       n_upd, v.stride, N, v.mode, thr);
            };
            run(only >= 0 ? 5 : 50); hipStreamSynchronize(st);
            auto t0 = std::chrono::high_resolution_clock::now();
            run(n_timed); hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n_timed;
            printf("%s   %3u %%     %8.2f       %.3e        %.2f TB/s\n", v.name, acc_pct, us, 7.0 * n_upd / (us * 1e-6), 7.0 * n_upd * 64 / (us * 1e-6) / 1e12);
        }
    return 0;
}
