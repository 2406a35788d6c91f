// Device self-test: the inline Philox4x32-10 of philox.h against rocRAND's own device
// engine (rocrand_init + rocrand4 from <rocrand/rocrand_kernel.h>) on the same
// (seed, subsequence, offset) triples.  Used by bpm_selftest_philox / tests only.
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#include "philox.h"

namespace bpm {
inline namespace BPM_VARIANT_NS {      // (philox.h: one kernel-symbol namespace per build variant)

__global__ void rocrand_check_kernel(uint32_t* out, int n, uint64_t seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // a spread of subsequences / blocks incl. the layout's own corners
    const uint64_t subseq = (i % 3 == 0) ? SUBSEQ_GLOBAL : (uint64_t)i * 2654435761ull;
    const uint64_t blk = ((uint64_t)(i * 37) << SLOT_BITS) | (uint64_t)(i % 120);
    const u32x4 m = philox_block(seed, subseq, blk);
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, subseq, 4ull * blk, &st);
    const uint4 r = rocrand4(&st);
    uint32_t* o = out + 8 * (size_t)i;
    o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.w;
    o[4] = r.x; o[5] = r.y; o[6] = r.z; o[7] = r.w;
}

inline void launch_rocrand_check(uint32_t* out, int n, uint64_t seed) {
    hipLaunchKernelGGL(rocrand_check_kernel, dim3((n + 127) / 128), dim3(128), 0, 0, out, n, seed);
}

}  // inline namespace BPM_VARIANT_NS
}  // namespace bpm
