// Microbenchmark (diagnostic): can the ~2.6 us floor of a DEPENDENT kernel launch be hidden by launching the next half generation on
// a second stream BEFORE the current one has finished, its wavefronts doing their state-independent prologue and then polling a flag
// that the command processor writes when the current kernel completes (hipStreamWriteValue32 behind it)?
// Model: kernel = prologue (one dependent table load + ALU), [wait for flag >= target], body (two dependent loads through a
// written state array + ALU + store).  "serial": one stream, plain dependent launches.  "overlap": streams A / B alternate, kernel
// g+1 is enqueued on the other stream right away and waits IN THE KERNEL for the flag of kernel g.  Every spin is bounded
// (a wavefront gives up after ~20 ms and raises an error flag), so nothing can hang.
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/overlap_flag tools/micro/overlap_flag.hip && ./build_variants/overlap_flag
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s failed: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
__global__ void k(const uint32_t* __restrict__ tab, uint4* state, const uint32_t* flag, uint32_t target, uint32_t* err, uint32_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = (i * 2654435761u) & mask;
    uint32_t v = tab[idx];                                   // prologue: state-independent (the update record)
    for (int r = 0; r < 40; ++r) v = v * 1664525u + 1013904223u;
    if (flag) {                                              // wait until the previous half generation is complete
        uint32_t seen = 0, spins = 0;
        do {
            seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (++spins > 200000u) { if (threadIdx.x == 0) atomicOr(err, 1u); break; }
        } while (seen < target);
    }
    // body: rows of the state the previous kernel wrote (system-scope loads: this kernel started before that one ended)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4* sv = reinterpret_cast<const u4*>(state);
    const u4 a0 = __builtin_nontemporal_load(&sv[(v >> 3) & mask]);
    const u4 a1 = __builtin_nontemporal_load(&sv[(a0.x + v) & mask]);
    uint4 s0 = make_uint4(a0.x, a0.y, a0.z, a0.w), s1 = make_uint4(a1.x, a1.y, a1.z, a1.w);
    s1.x += s0.y + v; s1.y ^= v;
    state[i & mask] = s1;
}
int main() {
    const uint32_t n = 1u << 16, mask = n - 1;
    std::vector<uint32_t> h(n);
    uint32_t x = 99u;
    for (auto& e : h) { x = x * 1664525u + 1013904223u; e = x >> 5; }
    uint32_t *tab, *err, *fl;
    uint4* st;
    CK(hipMalloc(&tab, n * 4)); CK(hipMalloc(&st, n * sizeof(uint4))); CK(hipMalloc(&err, 4));
    CK(hipMemcpy(tab, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(st, 1, n * sizeof(uint4))); CK(hipMemset(err, 0, 4));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (hipExtMallocWithFlags((void**)&fl, 64, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc(&fl, 64)); printf("flags in plain device memory\n"); }
    CK(hipMemset(fl, 0, 64));
    hipStream_t sA, sB;
    CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    const int G = 2000;
    for (int waves : {1024, 2048, 4096}) {
        // serial: dependent launches on one stream
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, sA, tab, st, (const uint32_t*)nullptr, 0u, err, mask);
        CK(hipStreamSynchronize(sA));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < G; ++i) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, sA, tab, st, (const uint32_t*)nullptr, 0u, err, mask);
        CK(hipStreamSynchronize(sA));
        const double serial = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / G;
        // overlap: kernel g on stream g % 2 waits in-kernel for flag[(g - 1) % 2] >= g (0: nothing to wait for), then the stream writes flag[g % 2] = g + 1
        CK(hipMemset(fl, 0, 64));
        CK(hipDeviceSynchronize());
        t0 = std::chrono::high_resolution_clock::now();
        hipError_t we = hipSuccess;
        for (int g = 0; g < G; ++g) {
            hipStream_t s = (g & 1) ? sB : sA;
            const uint32_t* wait = g == 0 ? nullptr : fl + 8 * ((g - 1) & 1);
            hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, s, tab, st, wait, (uint32_t)g, err, mask);
            we = hipStreamWriteValue32(s, fl + 8 * (g & 1), (uint32_t)(g + 1), 0);
            if (we != hipSuccess) break;
        }
        if (we != hipSuccess) { printf("hipStreamWriteValue32 failed: %s\n", hipGetErrorString(we)); return 1; }
        CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
        const double overlap = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / G;
        uint32_t e = 0;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("waves %5d: serial dependent launches %.2f us per kernel; overlapped (second stream + in-kernel flag wait) %.2f us per kernel; spin timeouts: %u\n",
               waves, serial, overlap, e);
        CK(hipMemset(err, 0, 4));
    }
    return 0;
}
