// MSM bucket accumulation: the dominant kernel (W*m mixed additions).
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

// one thread per bucket
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_accum_kernel(const uint32_t* __restrict__ points,
                                                                    const uint32_t* __restrict__ offsets,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    uint32_t nbuckets, uint32_t* buckets) {
    const uint32_t g = blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (g >= nbuckets) return;
    const uint32_t start = offsets[g], end = offsets[g + 1];
    G1Xyzz acc = G1Xyzz::inf();
    for (uint32_t pos = start; pos < end; ++pos) {
        const uint32_t pl = sorted[pos];
        const G1Affine p = ld_affine(points, pl & 0x7fffffffu);
        g1_madd(acc, p, (pl >> 31) != 0);
    }
    st_xyzz(buckets, g, acc);
}


void launch_msm_accum(const uint32_t* points, const uint32_t* offsets, const uint32_t* sorted, uint32_t nbuckets,
                      uint32_t* buckets, hipStream_t s) {
    hipLaunchKernelGGL(msm_accum_kernel, dim3((nbuckets + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS), dim3(MSM_ACC_THREADS), 0,
                       s, points, offsets, sorted, nbuckets, buckets);
}

}  // namespace ty
