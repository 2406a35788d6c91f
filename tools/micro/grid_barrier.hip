// Microbenchmark (diagnostic, never part of the product): cost of a device-wide barrier inside ONE persistent
// kernel, as the alternative to a kernel boundary between the two half generations (DESIGN.md section 5, item 2:
// an empty dependent launch costs ~2.8 us).  Every workgroup is one wavefront; all of them must be co-resident,
// which the host checks with the occupancy API before launching.  Spins are BOUNDED: a wave that waits more than
// SPIN_MAX polls raises the abort flag and every wave leaves, so the grid always drains.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

constexpr unsigned SPIN_MAX = 1u << 22;

// mode 0: counter barrier only (relaxed arrive / relaxed poll at agent scope)
// mode 1: release on arrive + acquire on leave (agent scope: what makes other XCDs' plain stores visible)
// mode 2: mode 1 + every wave writes a row before the barrier and reads another wave's row after it (checks the
//         visibility the sampler would need: rows written in phase A read by phase B on any XCD)
template <int MODE>
__global__ __launch_bounds__(64) void barrier_kernel(unsigned* cnt, unsigned* abort_flag, double* rows, unsigned* bad, int n_barriers) {
    const unsigned G = gridDim.x, w = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k < n_barriers; ++k) {
        if (MODE == 2) rows[(size_t)w * 128 + lane] = (double)(k + 1) * 1000.0 + (double)w;      // "phase" work: my row
        if (lane == 0) {
            if (MODE == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(k + 1) * G;
            unsigned spins = 0;
            while (true) {
                const unsigned v = MODE == 0 ? __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= target) break;
                if (++spins > SPIN_MAX || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();          // one wave per workgroup: orders the other lanes behind lane 0's acquire
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (MODE == 2) {
            const unsigned other = (w * 2654435761u + (unsigned)k * 40503u) % G;              // some other wave's row
            const double v = rows[(size_t)other * 128 + lane];
            if (v != (double)(k + 1) * 1000.0 + (double)other) atomicAdd(bad, 1u);
            acc += v;
            // second barrier of the "generation": nobody overwrites a row before all have read it
            if (lane == 0) {
                __hip_atomic_fetch_add(cnt + 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(k + 1) * G;
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > SPIN_MAX || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        }
    }
    if (acc == -1.0) rows[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE> int run(const char* name, unsigned G, int n_barriers, unsigned* cnt, unsigned* abort_flag, double* rows, unsigned* bad) {
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, barrier_kernel<MODE>, 64, 0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned resident = (unsigned)per_cu * (unsigned)prop.multiProcessorCount;
    if (G > resident) { printf("%-34s G=%u: only %u workgroups can be resident, skipped\n", name, G, resident); return 0; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned h_abort = 0, h_bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(cnt, 0, 64 * sizeof(unsigned)));
        CK(hipMemset(abort_flag, 0, sizeof(unsigned)));
        CK(hipMemset(bad, 0, sizeof(unsigned)));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(barrier_kernel<MODE>, dim3(G), dim3(64), 0, 0, cnt, abort_flag, rows, bad, n_barriers);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h_abort, abort_flag, sizeof(unsigned), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&h_bad, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (h_abort) { printf("%-34s G=%u: ABORTED (a wave waited too long: grid not co-resident?)\n", name, G); return 0; }
        if (ms < best) best = ms;
    }
    printf("%-34s G=%5u (resident limit %u): %.3f us per %s, stale reads %u\n", name, G, resident, best * 1e3f / n_barriers,
           MODE == 2 ? "write+barrier+read+barrier" : "barrier", h_bad);
    return 0;
}

int main() {
    unsigned *cnt, *abort_flag, *bad;
    double* rows;
    CK(hipMalloc(&cnt, 64 * sizeof(unsigned)));
    CK(hipMalloc(&abort_flag, sizeof(unsigned)));
    CK(hipMalloc(&bad, sizeof(unsigned)));
    CK(hipMalloc(&rows, (size_t)8192 * 128 * sizeof(double)));
    CK(hipMemset(rows, 0, (size_t)8192 * 128 * sizeof(double)));
    const int K = 500;
    for (unsigned G : {256u, 1024u, 2048u, 4096u}) {
        if (run<0>("relaxed counter barrier", G, K, cnt, abort_flag, rows, bad)) return 1;
        if (run<1>("release/acquire (agent) barrier", G, K, cnt, abort_flag, rows, bad)) return 1;
        if (run<2>("rel/acq + row write/read check", G, K, cnt, abort_flag, rows, bad)) return 1;
    }
    return 0;
}
