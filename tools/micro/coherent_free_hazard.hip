// Microbenchmark (diagnostic): does data written by kernels into an allocation of the GPU's hardware-coherent memory type
// (hipExtMallocWithFlags(..., hipDeviceMallocUncached) = the extended-scope fine-grained pool on this runtime) come back AFTER the
// allocation was freed and its memory handed to an ordinary hipMalloc?  The suspicion behind tools/coherent_memory_hazard.py: lines of
// that memory type that an L2 holds dirty are not written back by the agent-scope release at the end of a kernel (they are "coherent
// already") and land in memory whenever they are evicted -- on top of whatever owns the page by then.
//   hipcc -O2 --offload-arch=gfx950 -o build_variants/coherent_free_hazard tools/micro/coherent_free_hazard.hip && ./build_variants/coherent_free_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_fill(uint64_t* p, uint64_t n, uint64_t v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + i;
}
__global__ void k_rmw(uint64_t* p, uint64_t n) {                       // read-modify-write: pulls the lines into the L2s first
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] + 1;
}
__global__ void k_thrash(uint64_t* p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 3 + 1;
}
int main() {
    const uint64_t SZ = 8ull << 20, N = SZ / 8, BIG = 1ull << 30;
    uint64_t* big; (void)hipMalloc(&big, BIG); (void)hipMemset(big, 1, BIG);
    std::vector<uint64_t> h(N);
    for (int mode = 0; mode < 2; ++mode) {                                 // 0: first allocation ordinary (control), 1: first allocation of the coherent type
        long long total_bad = 0, total_old = 0; int same_va = 0;
        for (int t = 0; t < 40; ++t) {
            uint64_t *A = nullptr, *B = nullptr;
            if (mode) hipExtMallocWithFlags((void**)&A, SZ, hipDeviceMallocUncached); else hipMalloc((void**)&A, SZ);
            const uint64_t P1 = 0x1111000000000000ull + ((uint64_t)t << 32), P2 = 0x2222000000000000ull + ((uint64_t)t << 32);
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, A, N, P1);
            for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_rmw, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, A, N);
            hipDeviceSynchronize();
            hipFree(A);
            hipMalloc((void**)&B, SZ);
            same_va += (A == B);
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, B, N, P2);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_thrash, dim3((unsigned)((BIG / 8 + 255) / 256)), dim3(256), 0, 0, big, BIG / 8);    // evict whatever the L2s still hold
            hipDeviceSynchronize();
            hipMemcpy(h.data(), B, SZ, hipMemcpyDeviceToHost);
            long long bad = 0, old = 0;
            for (uint64_t i = 0; i < N; ++i) if (h[i] != P2 + i) { ++bad; old += (h[i] >> 48) == 0x1111; }
            total_bad += bad; total_old += old;
            hipFree(B);
        }
        printf("first allocation %-28s: 40 rounds of 8 MB, same address handed out again %d times; elements of the SECOND allocation wrong: %lld, of them holding the FIRST allocation's data: %lld\n",
               mode ? "hardware-coherent (Uncached)" : "ordinary (control)", same_va, total_bad, total_old);
    }
    return 0;
}
