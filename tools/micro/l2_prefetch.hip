// Microbenchmark (diagnostic): do lines that launch N touches survive in the XCDs' L2 for launch N+1?
// Launch i walks `hops` dependent random lookups through slice i of a read-only table (256 KB per slice, 64 slices: a slice is
// long gone from a 4 MB L2 when it comes round again, so every lookup is a first touch served by the Infinity Cache) and writes
// a state array (like an update kernel).  With `pf` every wavefront of launch i also touches 32 lines of slice i+1 -- the 64
// wavefronts an XCD gets (workgroup id modulo 8, if dispatch is round-robin) cover the whole slice -- so that launch i+1 finds it
// in ITS L2.  If the per-launch time drops with pf, prefetching the next launch's tables from inside the current one works.
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/l2_prefetch tools/micro/l2_prefetch.hip && ./build_variants/l2_prefetch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
constexpr uint32_t SLICE = 65536;        // u32 entries per slice (256 KB)
constexpr int NSLICE = 64;
__global__ void k(const uint32_t* __restrict__ tab, uint4* state, int slice, int hops, int pf, int waves) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t* t = tab + (size_t)slice * SLICE;
    uint32_t idx = (i * 2654435761u) & (SLICE - 1);
    uint32_t sink = 0;
    if (pf) {        // 2048 lines of 128 B per slice; wavefront (blockIdx.x >> 3) of its XCD touches 32 of them (waves / 8 wavefronts per XCD)
        const uint32_t per_wave = 2048u / (uint32_t)(waves / 8);
        const uint32_t* nt = tab + (size_t)((slice + 1) % NSLICE) * SLICE;
        if (threadIdx.x < per_wave) sink = nt[((blockIdx.x >> 3) * per_wave + threadIdx.x) * 32u];
    }
    uint32_t v = idx;
    for (int h = 0; h < hops; ++h) { v = t[idx]; idx = v & (SLICE - 1); }
    uint4 s = state[i & 65535u];
    s.x += v + (sink & 1u);
    state[i & 65535u] = s;                // a written state array: every launch dirties lines, like the sampler's
}
int main() {
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<uint32_t> h((size_t)SLICE * NSLICE);
    uint32_t x = 777u;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = x >> 7; }
    uint32_t* tab; uint4* st;
    (void)hipMalloc(&tab, h.size() * 4); (void)hipMalloc(&st, 65536 * sizeof(uint4));
    (void)hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemset(st, 0, 65536 * sizeof(uint4));
    printf("waves hops prefetch  us per launch\n");
    for (int waves : {512, 4096})
        for (int hops : {1, 2, 3})
            for (int pf = 0; pf < 2; ++pf) {
                auto run = [&](int n, int i0) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, s, tab, st, (i0 + i) % NSLICE, hops, pf, waves); };
                run(128, 0); (void)hipStreamSynchronize(s);
                auto t0 = std::chrono::high_resolution_clock::now();
                run(2048, 0); (void)hipStreamSynchronize(s);
                printf("%5d %4d %8d  %.2f\n", waves, hops, pf, std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 2048.0);
            }
    return 0;
}
