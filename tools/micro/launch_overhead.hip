// Microbenchmark (diagnostic): cost of back-to-back dependent kernel launches on one stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char pad[640]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_empty_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = b.pad[3]; }
__global__ void k_touch(double* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] += 1.0; }
template <class F> double timeit(hipStream_t s, int iters, F f) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < iters; ++i) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / iters;
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4);
    double* x; hipMalloc(&x, 8192 * 100 * 8); hipMemset(x, 0, 8192 * 100 * 8);
    Big b{};
    printf("empty 1x64          : %.2f us/launch\n", timeit(s, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d); }));
    printf("empty 1x1024        : %.2f us/launch\n", timeit(s, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, s, d); }));
    printf("empty 4096x64       : %.2f us/launch\n", timeit(s, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, s, d); }));
    printf("empty 1024x256      : %.2f us/launch\n", timeit(s, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, d); }));
    printf("empty big-arg 4096x64: %.2f us/launch\n", timeit(s, 2000, [&] { hipLaunchKernelGGL(k_empty_big, dim3(4096), dim3(64), 0, s, b, d); }));
    printf("touch 6.5MB 3200x256: %.2f us/launch\n", timeit(s, 2000, [&] { hipLaunchKernelGGL(k_touch, dim3(3200), dim3(256), 0, s, x, 819200); }));
    hipStream_t s0 = 0;
    printf("empty 4096x64 null stream: %.2f us/launch\n", timeit(s0, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, s0, d); }));
    return 0;
}
