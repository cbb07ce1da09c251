// Pippenger bucket MSM over BLS12-381 G1 (replaces the per-term double-and-add of
// KzgScheme::evaluate_in_s, /root/reference/kzg/src/lib.rs:41-54).
//
//   1. msm_digits_kernel   one thread per scalar: Montgomery -> canonical (ark-ff into_repr, the
//                          conversion lib.rs:49 performs per term), signed c-bit window slicing,
//                          one key per (window, scalar) + bucket histogram.
//   2. scan kernels        exclusive prefix sum of the histogram -> bucket offsets.
//   3. msm_scatter_kernel  counting sort of (point index, sign) by bucket.
//   4. msm_accum_kernel    one thread per bucket: XYZZ accumulator in registers, mixed additions of
//                          the bucket's affine points gathered from the resident SRS.
//   5. msm_rc_* kernels    per bucket set sum_k w(k)*B_k by the row/column split (launch.hpp): plain row
//                          and column sums, then bit planes of the R + C weighted sums via wavefront
//                          __shfl_xor butterflies of whole points.  (msm_reduce_kernel / msm_fold_kernel:
//                          the first version, running sums + a small scalar multiplication per thread.)
// The W window sums go to the host, which applies the 2^(c*j) weights (Horner) and normalises
// to the canonical affine point.  Group addition is commutative and the result is canonical, so
// the non-deterministic order inside a bucket (atomics in step 3) cannot change the output.
#pragma once
#include "g1.hpp"
#include "launch.hpp"

namespace ty {

constexpr uint32_t MSM_SKIP = 0xffffffffu;
constexpr int MSM_THREADS = 256;
constexpr int MSM_ACC_THREADS = 128;

// HBM form of a field element: 12 packed 32-bit words (fq30.hpp); three 16-byte accesses.
__device__ __forceinline__ Fq30 ld_fq(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    return fq30_unpack(w);
}
__device__ __forceinline__ void st_fq(uint32_t* p, const Fq30& r) {
    uint32_t w[12];
    fq30_pack(r, w);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}
__device__ __forceinline__ G1Affine ld_affine(const uint32_t* pts, uint64_t idx) {
    const uint32_t* p = pts + idx * PT_WORDS;
    G1Affine r;
    r.x = ld_fq(p);
    r.y = ld_fq(p + 12);
    return r;
}
__device__ __forceinline__ G1Xyzz ld_xyzz(const uint32_t* b, uint64_t idx) {
    const uint32_t* p = b + idx * 48;
    G1Xyzz r;
    r.x = ld_fq(p);
    r.y = ld_fq(p + 12);
    r.zz = ld_fq(p + 24);
    r.zzz = ld_fq(p + 36);
    return r;
}
__device__ __forceinline__ void st_xyzz(uint32_t* b, uint64_t idx, const G1Xyzz& r) {
    uint32_t* p = b + idx * 48;
    st_fq(p, r.x);
    st_fq(p + 12, r.y);
    st_fq(p + 24, r.zz);
    st_fq(p + 36, r.zzz);
}

}  // namespace ty
