// Device side of tools/micro/aql_direct.cpp (built with --cuda-device-only --no-gpu-bundle-output into a plain code object
// that the host program loads through the HSA loader).  No blockDim/gridDim use: the kernels need no hidden arguments.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void k_empty(int* p) {
    if (p && threadIdx.x == 9999) *p = 1;
}

// Launch number `shift` lets workgroup b own the 64-element block (b + shift) mod nb: a block of x is touched by a different
// workgroup -- hence a different XCD and a different L2 -- in every launch, so x[i] == number of launches afterwards only if
// each launch saw what the previous one wrote (cross-XCD visibility at the kernel boundary).
extern "C" __global__ void k_chain(double* x, uint32_t nb, uint32_t shift) {
    const uint32_t b = (blockIdx.x + shift) % nb;
    const uint32_t i = b * 64u + threadIdx.x;
    x[i] += 1.0;
}

// The same with accesses the hardware keeps coherent across the XCDs by itself (agent-scope relaxed atomics: sc1 loads / stores
// that do not rest in the non-coherent L2), for packets without acquire / release fences.
extern "C" __global__ void k_chain_coherent(double* x, uint32_t nb, uint32_t shift) {
    const uint32_t b = (blockIdx.x + shift) % nb;
    const uint32_t i = b * 64u + threadIdx.x;
    const double v = __hip_atomic_load(&x[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&x[i], v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A gather kernel shaped like the update kernel's traffic: every lane reads `rows` random 16-byte pieces of a table that the
// previous launch wrote, and writes one.
extern "C" __global__ void k_gather(uint4* tab, uint32_t mask, uint32_t rows, uint32_t shift) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    uint32_t idx = (i * 2654435761u + shift * 40503u) & mask;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rows; ++r) {
        const uint4 v = tab[(idx + r * 7919u) & mask];
        acc += v.y;
    }
    tab[(i + shift * 64u) & mask].y = acc;
}

// One workgroup of 1024 threads with 10 KB of LDS that does nothing: the shape of cr_adapt_kernel (part 4 of aql_direct.cpp).
extern "C" __global__ __launch_bounds__(1024) void k_empty_1024(int* p) {
    __shared__ double pad[1280];
    if (p && threadIdx.x == 99999) { pad[threadIdx.x & 1023] = 1.0; *p = (int)pad[3]; }
}

// Part 5: does a line that launch i pulls into an XCD's L2 survive into launch i + 1?  Launch i: workgroup (= wavefront) w reads the
// 64-byte record w of slice i of a read-only table (first touch: every launch has a slice of its own), then one dependent 16-byte gather,
// and, with prefetch != 0, also touches record w of slice i + 1 (workgroup w of the next launch runs on the same XCD: w mod 8).
extern "C" __global__ void k_slice(const uint4* rec, const uint4* data, uint4* out, uint32_t slice, uint32_t n_per_slice, uint32_t mask, uint32_t prefetch) {
    const uint32_t w = blockIdx.x;
    const uint4 r = rec[(slice * n_per_slice + w) * 4u + (threadIdx.x & 3u)];          // 64-byte record, 4 x 16 B
    const uint4 v = data[(r.x + threadIdx.x) & mask];                                   // dependent gather
    if (prefetch) {
        const uint4 nx = rec[((slice + 1u) * n_per_slice + w) * 4u + (threadIdx.x & 3u)];
        if (nx.w == 0xdeadbeefu) out[0] = nx;                                           // (keeps the load alive)
    }
    if (threadIdx.x == 0) out[w] = v;
}
