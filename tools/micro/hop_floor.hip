// Microbenchmark (diagnostic): what a latency-bound update kernel costs as a function of the number of DEPENDENT memory hops on a
// wavefront's critical path -- the floor under the small-d kernels (BASELINE configs 3 and 5), whose launches have too few
// bytes to be bandwidth bound.  Kernel: every lane walks `hops` dependent random 16-byte gathers through a table the size of
// the state matrix (next index taken from the loaded value), then stores 16 bytes; `waves` wavefronts per launch; two dependent
// launches back to back on one stream, like the two half generations of a generation.
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/hop_floor tools/micro/hop_floor.hip && ./build_variants/hop_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_hops(uint4* tab, uint4* out, uint32_t mask, int hops, int dirty) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = (i * 2654435761u) & mask;
    uint4 v = make_uint4(idx, 0, 0, 0);
    for (int h = 0; h < hops; ++h) {
        v = tab[idx];
        idx = v.x & mask;          // the next address depends on the loaded value
    }
    out[i] = v;
    if (dirty) tab[(i * 40503u) & mask].y = v.y + 1u;      // the table is WRITTEN by every launch, like a state matrix (never the .x links)
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const uint32_t n_tab = 1u << 16;                      // 65536 rows of 16 bytes = 1 MB (cfg3's state matrix)
    std::vector<uint4> h(n_tab);
    uint32_t x = 12345u;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = make_uint4(x >> 8, x, x, x); }
    uint4 *tab, *out;
    hipMalloc(&tab, n_tab * sizeof(uint4)); hipMalloc(&out, (1u << 20) * sizeof(uint4));
    hipMemcpy(tab, h.data(), n_tab * sizeof(uint4), hipMemcpyHostToDevice);
    printf("table  waves  hops  us per launch (dependent launches on one stream, 2000 launches)\n");
    for (int dirty = 0; dirty < 2; ++dirty)
    for (int waves : {512, 2048, 8192}) {
        for (int hops : {0, 1, 2, 3, 4, 5, 10, 20}) {
            auto run = [&](int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_hops, dim3(waves), dim3(64), 0, s, tab, out, n_tab - 1, hops, dirty); };
            run(100); hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            run(2000); hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 2000.0;
            printf("%s  %5d  %4d  %.2f\n", dirty ? "written  " : "read-only", waves, hops, us);
        }
    }
    return 0;
}
