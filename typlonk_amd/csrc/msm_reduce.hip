// MSM bucket reduction: per-window sum_k k*B_k with running sums + wavefront __shfl_xor butterflies.
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

__device__ __forceinline__ G1Xyzz shfl_xor_point(const G1Xyzz& p, int mask) {
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        r.x.v[i] = __shfl_xor(p.x.v[i], mask);
        r.y.v[i] = __shfl_xor(p.y.v[i], mask);
        r.zz.v[i] = __shfl_xor(p.zz.v[i], mask);
        r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask);
    }
    return r;
}

// Thread t owns buckets [t*L, (t+1)*L) of the flat (window-major) bucket array, L = min(8, B).
// node value = sum_l w(s*L + l) * bucket[l]   with s = t mod (B/L) and bucket weight
// w(k) = (k >> v) + 1, v = 0 except in the top window (v = top_v, see msm_digits_kernel);
// lanes of the same window are then summed with a __shfl_xor butterfly over `group` lanes and
// lane 0 of each group stores one partial.  partials[t / group].
__global__ __launch_bounds__(64) void msm_reduce_kernel(const uint32_t* __restrict__ buckets, uint32_t B, uint32_t L,
                                                        uint32_t nodes_total, uint32_t group, uint32_t cbits,
                                                        uint32_t W, uint32_t top_v, uint32_t* partials) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    G1Xyzz v = G1Xyzz::inf();
    if (t < nodes_total) {
        const uint32_t npw = B / L;
        const uint32_t s = t % npw;
        const uint32_t wv = (t / npw + 1 == W) ? top_v : 0u;
        const uint32_t vmask = (1u << wv) - 1;
        const uint64_t base = (uint64_t)t * L;
        // u = sum_l (w(k0 + l) - w(k0)) * x_l : the running sum is added once per weight step
        G1Xyzz running = G1Xyzz::inf(), u = G1Xyzz::inf();
        for (uint32_t l = L - 1; l >= 1; --l) {
            running = g1_add(running, ld_xyzz(buckets, base + l));
            if (((s * L + l) & vmask) == 0) u = g1_add(u, running);
        }
        running = g1_add(running, ld_xyzz(buckets, base));
        // w(k0) * S with fixed 2-bit windows (table S, 2S, 3S): lanes hold different multipliers, so a
        // bitwise double-and-add executes its conditional add at every position anyway; base 4 halves them
        const uint32_t kmul = ((s * L) >> wv) + 1;
        const G1Xyzz s2 = g1_dbl(running);
        const G1Xyzz s3 = g1_add(s2, running);
        G1Xyzz acc = G1Xyzz::inf();
        for (int d = (int)(cbits + 1) / 2 - 1; d >= 0; --d) {
            acc = g1_dbl(g1_dbl(acc));
            const uint32_t dig = (kmul >> (2 * d)) & 3u;
            if (dig) acc = g1_add(acc, dig == 1 ? running : (dig == 2 ? s2 : s3));
        }
        v = g1_add(u, acc);
    }
    for (uint32_t mask = 1; mask < group; mask <<= 1) {
        const G1Xyzz o = shfl_xor_point(v, (int)mask);
        v = g1_add(v, o);
    }
    if (t < nodes_total && (threadIdx.x & (group - 1)) == 0) st_xyzz(partials, t / group, v);
}

// in: W * n_in points (window-major); sums groups of `group` (= min(64, n_in)) consecutive points.
__global__ __launch_bounds__(64) void msm_fold_kernel(const uint32_t* __restrict__ in, uint32_t total, uint32_t group,
                                                      uint32_t* out) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    G1Xyzz v = G1Xyzz::inf();
    if (t < total) v = ld_xyzz(in, t);
    for (uint32_t mask = 1; mask < group; mask <<= 1) {
        const G1Xyzz o = shfl_xor_point(v, (int)mask);
        v = g1_add(v, o);
    }
    if (t < total && (threadIdx.x & (group - 1)) == 0) st_xyzz(out, t / group, v);
}


void launch_msm_reduce(const uint32_t* buckets, uint32_t B, uint32_t L, uint32_t nodes_total, uint32_t group,
                       uint32_t cbits, uint32_t W, uint32_t top_v, uint32_t* partials, hipStream_t s) {
    hipLaunchKernelGGL(msm_reduce_kernel, dim3((nodes_total + 63) / 64), dim3(64), 0, s, buckets, B, L, nodes_total, group,
                       cbits, W, top_v, partials);
}
void launch_msm_fold(const uint32_t* in, uint32_t total, uint32_t group, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(msm_fold_kernel, dim3((total + 63) / 64), dim3(64), 0, s, in, total, group, out);
}

}  // namespace ty
