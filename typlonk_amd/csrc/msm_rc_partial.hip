// MSM bucket reduction, first launch: plain row and column partial sums of the bucket grid (launch.hpp, msm_reduce.hip).
// The one throughput-bound kernel of the reduction -- two wavefronts per SIMD, 14 full additions per thread pair of roles --
// compiled with the single-accumulator multiplication (the rest of the reduction, dependent chains, is msm_reduce.hip).
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

// Two roles in one launch.  Row role: thread sums 2^llc consecutive buckets of one row.  Column role: thread sums the
// buckets of 2^lhc consecutive rows in one column (adjacent lanes = adjacent columns, so the loads coalesce) and stores
// hchunk-fastest so the fold reads runs.
__global__ __launch_bounds__(64) void msm_rc_partial_kernel(const uint32_t* __restrict__ buckets, RcShape sh,
                                                            uint32_t nrow, uint32_t nrow_pad, uint32_t ncol,
                                                            uint32_t* pb, uint32_t* pa) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t < nrow_pad) {
        if (t >= nrow) return;
        const uint64_t base = (uint64_t)t << sh.llc;
        G1Xyzz v = ld_xyzz(buckets, base);
        for (uint32_t l = 1; l < (1u << sh.llc); ++l) v = g1_add(v, ld_xyzz(buckets, base + l));
        st_xyzz(pb, t, v);
        return;
    }
    const uint32_t u = t - nrow_pad;
    if (u >= ncol) return;
    const uint32_t per_set = sh.c1 - sh.lhc;
    const uint32_t set = u >> per_set, rem = u & ((1u << per_set) - 1);
    const uint32_t lo = rem & ((1u << sh.cl) - 1), hchunk = rem >> sh.cl;
    const uint64_t first = ((uint64_t)set << sh.c1) + ((uint64_t)(hchunk << sh.lhc) << sh.cl) + lo;
    G1Xyzz v = ld_xyzz(buckets, first);
    for (uint32_t h = 1; h < (1u << sh.lhc); ++h) v = g1_add(v, ld_xyzz(buckets, first + ((uint64_t)h << sh.cl)));
    st_xyzz(pa, ((((uint64_t)set << sh.cl) + lo) << (sh.ch - sh.lhc)) + hchunk, v);
}

void launch_msm_rc_partial(const uint32_t* buckets, const RcShape& sh, uint32_t nrow, uint32_t nrow_pad, uint32_t ncol, uint32_t* pb,
                           uint32_t* pa, hipStream_t s) {
    hipLaunchKernelGGL(msm_rc_partial_kernel, dim3((nrow_pad + ncol + 63) / 64), dim3(64), 0, s, buckets, sh, nrow, nrow_pad, ncol, pb,
                       pa);
}

}  // namespace ty
