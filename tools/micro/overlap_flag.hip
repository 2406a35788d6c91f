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
// Variant "tickets": the producer kernel signals its own completion -- every wavefront, after its (write-through) stores have been
// acknowledged, takes a ticket on one of 64 counters; the last of a counter's group takes a ticket on the second level; the last of
// those stores the generation number into the flag the consumer polls.  Counters only ever grow (targets scale with the generation).
__global__ void k2(const uint32_t* __restrict__ tab, uint4* state, const uint32_t* flag, uint32_t wait_target, uint32_t own_j, uint32_t* err,
                   uint32_t mask, uint32_t* tickets, uint32_t* done, uint32_t waves) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = (i * 2654435761u) & mask;
    uint32_t v = tab[idx];
    for (int r = 0; r < 40; ++r) v = v * 1664525u + 1013904223u;
    if (flag) {
        uint32_t seen = 0, spins = 0;
        do {
            seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > 200000u) { if (threadIdx.x == 0) atomicOr(err, 1u); break; }
        } while (seen < wait_target);
    }
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4* sv = reinterpret_cast<u4*>(state);
    const u4 a0 = __builtin_nontemporal_load(&sv[(v >> 3) & mask]);
    u4 a1 = __builtin_nontemporal_load(&sv[(a0.x + v) & mask]);
    a1.x += a0.y + v; a1.y ^= v;
    __hip_atomic_store(&reinterpret_cast<uint32_t*>(state)[4 * (i & mask)], a1.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    __hip_atomic_store(&reinterpret_cast<uint32_t*>(state)[4 * (i & mask) + 1], a1.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        const uint32_t grp = blockIdx.x & 63u, per = waves / 64u;
        const uint32_t t1 = __hip_atomic_fetch_add(&tickets[grp * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t1 + 1u == (own_j + 1u) * per) {
            const uint32_t t2 = __hip_atomic_fetch_add(&tickets[64u * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 + 1u == (own_j + 1u) * 64u) __hip_atomic_store(done, own_j + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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
        // tickets: kernel g on stream g % 2 waits for done[(g - 1) % 2] >= g; it signals done[g % 2] = g + 1 itself
        uint32_t* tk; CK(hipMalloc(&tk, 2 * 65 * 16 * 4)); CK(hipMemset(tk, 0, 2 * 65 * 16 * 4)); CK(hipMemset(fl, 0, 64));
        CK(hipDeviceSynchronize());
        t0 = std::chrono::high_resolution_clock::now();
        for (int g = 0; g < G; ++g) {
            hipStream_t s = (g & 1) ? sB : sA;
            const uint32_t* wait = g == 0 ? nullptr : fl + 8 * ((g - 1) & 1);
            // per-stream generation counter: kernel g is the (g / 2)-th kernel of its stream
            hipLaunchKernelGGL(k2, dim3(waves), dim3(64), 0, s, tab, st, wait, (uint32_t)(g == 0 ? 0 : (g - 1) / 2 + 1), (uint32_t)(g / 2), err, mask,
                               tk + (g & 1) * 65 * 16, fl + 8 * (g & 1), (uint32_t)waves);
        }
        CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
        const double tick = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / G;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("waves %5d: overlapped with in-kernel completion tickets %.2f us per kernel; spin timeouts: %u\n", waves, tick, e);
        CK(hipMemset(err, 0, 4)); CK(hipFree(tk));
    }
    return 0;
}
