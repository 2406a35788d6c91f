// Device side of tools/micro/handover.cpp (plain code object, no hidden arguments).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void k_work(int* p) {          // stands for a half generation's update kernel: 64 empty workgroups
    if (p && threadIdx.x == 9999) *p = 1;
}

// the push exchange's hand-over kernel for two ranks: announce `seq` in the peer's flag, wait for the peer's announcement in one's own
extern "C" __global__ void k_sync(unsigned long long* mine, unsigned long long* peer, unsigned long long seq, unsigned long long timeout_ticks) {
    if (threadIdx.x != 0) return;
    __hip_atomic_store(peer, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
