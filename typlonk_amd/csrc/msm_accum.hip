// MSM bucket accumulation: the dominant kernel (W*m mixed additions).
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

// one thread per bucket, buckets taken in the size-sorted order of launch_bucket_order.  The next
// entry's index and point are fetched (two dependent gathers) before the current mixed addition is
// issued, so their latency hides under ~2000 instructions of arithmetic.
struct PackedPoint {
    uint4 w[6];
};
__device__ __forceinline__ PackedPoint ld_packed(const uint32_t* points, uint32_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(points + (uint64_t)idx * PT_WORDS);
    PackedPoint p;
#pragma unroll
    for (int i = 0; i < 6; ++i) p.w[i] = q[i];
    return p;
}
__device__ __forceinline__ G1Affine unpack_point(const PackedPoint& p) {
    const uint32_t wx[12] = {p.w[0].x, p.w[0].y, p.w[0].z, p.w[0].w, p.w[1].x, p.w[1].y,
                             p.w[1].z, p.w[1].w, p.w[2].x, p.w[2].y, p.w[2].z, p.w[2].w};
    const uint32_t wy[12] = {p.w[3].x, p.w[3].y, p.w[3].z, p.w[3].w, p.w[4].x, p.w[4].y,
                             p.w[4].z, p.w[4].w, p.w[5].x, p.w[5].y, p.w[5].z, p.w[5].w};
    G1Affine r;
    r.x = fq30_unpack(wx);
    r.y = fq30_unpack(wy);
    return r;
}

__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_accum_kernel(const uint32_t* __restrict__ points,
                                                                    const uint32_t* __restrict__ offsets,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    const uint32_t* __restrict__ order,
                                                                    uint32_t nbuckets, uint32_t* buckets) {
    const uint32_t t = blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (t >= nbuckets) return;
    const uint32_t g = order[t];
    const uint32_t start = offsets[g], end = offsets[g + 1];
    G1Xyzz acc = G1Xyzz::inf();
    uint32_t pl_next = 0;
    PackedPoint pk_next;
    if (start < end) {
        pl_next = sorted[start];
        pk_next = ld_packed(points, pl_next & 0x7fffffffu);
    }
    for (uint32_t pos = start; pos < end; ++pos) {
        const uint32_t pl = pl_next;
        const PackedPoint pk = pk_next;
        if (pos + 1 < end) {
            pl_next = sorted[pos + 1];
            pk_next = ld_packed(points, pl_next & 0x7fffffffu);
        }
        g1_madd(acc, unpack_point(pk), (pl >> 31) != 0);
    }
    st_xyzz(buckets, g, acc);
}

void launch_msm_accum(const uint32_t* points, const uint32_t* offsets, const uint32_t* sorted, const uint32_t* order,
                      uint32_t nbuckets, uint32_t* buckets, hipStream_t s) {
    hipLaunchKernelGGL(msm_accum_kernel, dim3((nbuckets + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS), dim3(MSM_ACC_THREADS), 0,
                       s, points, offsets, sorted, order, nbuckets, buckets);
}

}  // namespace ty
