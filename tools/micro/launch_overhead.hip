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
    {   // the same dependent chain as ONE graph launch (stream capture of 2000 launches, replayed)
        hipGraph_t g; hipGraphExec_t ge;
        const int n = 2000;
        for (int variant = 0; variant < 3; ++variant) {
            hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int i = 0; i < n; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, s, d);
                else if (variant == 1) hipLaunchKernelGGL(k_touch, dim3(3200), dim3(256), 0, s, x, 819200);
                else hipLaunchKernelGGL(k_empty_big, dim3(4096), dim3(64), 0, s, b, d);
            }
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (5.0 * n);
            printf("graph of %d x %s: %.2f us/launch\n", n, variant == 0 ? "empty 4096x64" : (variant == 1 ? "touch 6.5MB 3200x256" : "empty big-arg (640 B) 4096x64"), us);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    hipStream_t s0 = 0;
    printf("empty 4096x64 null stream: %.2f us/launch\n", timeit(s0, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, s0, d); }));
    return 0;
}
