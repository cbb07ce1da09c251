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

// wavefronts per SIMD the compiler budgets registers for (512 / MSM_ACC_WAVES VGPRs per lane, spilling what does not fit)
#ifndef MSM_ACC_WAVES
#define MSM_ACC_WAVES 2
#endif
__global__ __launch_bounds__(MSM_ACC_THREADS) __attribute__((amdgpu_waves_per_eu(MSM_ACC_WAVES, MSM_ACC_WAVES))) void msm_accum_kernel(const uint32_t* __restrict__ points,
                                                                    const uint32_t* __restrict__ offsets,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    const uint32_t* __restrict__ order,
                                                                    uint32_t nbuckets, uint32_t cap, uint32_t init,
                                                                    uint32_t* buckets) {
    const uint32_t t = blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    if (t >= nbuckets) return;
    const uint32_t g = order[t];
    const uint32_t start = offsets[g];
    const uint32_t end = offsets[g + 1] - start > cap ? start : offsets[g + 1];  // a heavy bucket is msm_heavy_kernel's
    // init != 0: a later chunk of the same MSM (msm_host.hip, msm_enqueue) continues from the stored bucket
    if (init && start >= end) return;
    G1Xyzz acc = init ? ld_xyzz(buckets, g) : G1Xyzz::inf();
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

// The same with L = 2, 4, 8 or 16 lanes per bucket (KzgScheme::evaluate_in_s of a SHORT vector -- an 8-way shard of
// a 2^20-term commitment, kzg/src/lib.rs:41-54 -- with small windows: 2^14..2^16 buckets of 30..130 entries each).
// One thread per bucket would leave three quarters of the chip idle and every thread with a long dependent chain;
// here the L lanes of a group stride over the bucket's entries and a wavefront butterfly (butterfly_add: the two
// lanes of a pair share one XYZZ addition) folds the L partial sums.  Schedule, cap and `init` as above.
template <int L>
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_accum_ml_kernel(const uint32_t* __restrict__ points,
                                                                       const uint32_t* __restrict__ offsets,
                                                                       const uint32_t* __restrict__ sorted,
                                                                       const uint32_t* __restrict__ order,
                                                                       uint32_t nbuckets, uint32_t cap, uint32_t init,
                                                                       uint32_t split, uint32_t* buckets) {
    // Two size classes: the schedule is sorted by population, so slots [0, split) -- the larger buckets -- get L lanes and
    // the rest L / 2 (split = nbuckets: one class).  split * L is a multiple of the workgroup size: a workgroup, and
    // with it every wavefront, is in one class.  Half the buckets at half the lanes save a third of the butterfly work.
    const uint32_t t = blockIdx.x * MSM_ACC_THREADS + threadIdx.x;
    const uint32_t ta = split * L;
    const uint32_t lanes = t < ta ? (uint32_t)L : (uint32_t)(L > 1 ? L / 2 : 1);
    const uint32_t t2 = t < ta ? t : t - ta;
    const uint32_t slot = (t < ta ? 0u : split) + t2 / lanes, lane = t2 % lanes;
    // every lane of the wavefront takes part in the butterfly: lanes past the last bucket carry the identity
    uint32_t g = 0, start = 0, end = 0;
    if (slot < nbuckets) {
        g = order[slot];
        start = offsets[g];
        end = offsets[g + 1] - start > cap ? start : offsets[g + 1];   // a heavy bucket is msm_heavy_kernel's
    }
    const bool skip = slot >= nbuckets || (init && start >= end);   // nothing to add: the stored bucket stays
    G1Xyzz acc = (init && !skip && lane == 0) ? ld_xyzz(buckets, g) : G1Xyzz::inf();
    uint32_t pl_next = 0;
    PackedPoint pk_next;
    uint32_t pos = start + lane;
    if (pos < end) {
        pl_next = sorted[pos];
        pk_next = ld_packed(points, pl_next & 0x7fffffffu);
    }
    for (; pos < end; pos += lanes) {
        const uint32_t pl = pl_next;
        const PackedPoint pk = pk_next;
        if (pos + lanes < end) {
            pl_next = sorted[pos + lanes];
            pk_next = ld_packed(points, pl_next & 0x7fffffffu);
        }
        g1_madd(acc, unpack_point(pk), (pl >> 31) != 0);
    }
#pragma unroll 1
    for (int mask = 1; mask < (int)lanes; mask <<= 1) acc = butterfly_add(acc, mask);
    if (!skip && lane == 0) st_xyzz(buckets, g, acc);
}

// Heavy buckets (adversarial scalar sets, or a fixed-base window count whose top window is only a few bits wide): a bucket
// with more than `cap` entries is skipped by the accumulate kernel and was cut into tasks of msm_task_len(count) entries
// (msm_seg_sort_kernel / order kernels).  ONE WAVEFRONT per task, grid-strided (the task count lives on the device): the 64
// lanes stride over the task's entries and a butterfly folds them -- a task is 4 (16) additions and six butterfly steps
// deep instead of cap additions in a row.  The wavefront that finishes the LAST task of a bucket (a counter per heavy
// bucket) folds that bucket's partials in the same way and adds them to the bucket: every bucket is closed as soon as it
// can be, by whichever wavefront gets there, with no second launch and no grid-wide wait.  With no task at all (every
// ordinary MSM) the launch returns at once.  Round 2 ran a task on ONE thread (cap = 512: a 3.4-ms chain) and folded all
// buckets in the last workgroup: a 2^20-term MSM of 1024 distinct scalars took 349 ms, a 2^16-term one whose top window
// holds 2 bits 10 ms (profiles/r03_heavy_tasks_ab.txt).
// hist516: [512] heavy buckets, [513] tasks.
__global__ __launch_bounds__(MSM_ACC_THREADS) void msm_heavy_kernel(const uint32_t* __restrict__ points,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    const uint32_t* __restrict__ hist516, uint32_t* heavy,
                                                                    const uint32_t* __restrict__ tasks, uint32_t* partial,
                                                                    uint32_t* buckets) {
    const uint32_t ntasks = hist516[513];
    if (ntasks == 0) return;
    constexpr uint32_t WPG = MSM_ACC_THREADS / 64;   // wavefronts per workgroup
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t t = blockIdx.x * WPG + wave; t < ntasks; t += gridDim.x * WPG) {   // t is uniform in the wavefront
        G1Xyzz acc = G1Xyzz::inf();
        const uint32_t end = tasks[3 * t + 1], h = tasks[3 * t + 2];
        for (uint32_t pos = tasks[3 * t] + lane; pos < end; pos += 64) {
            const uint32_t pl = sorted[pos];
            g1_madd(acc, unpack_point(ld_packed(points, pl & 0x7fffffffu)), (pl >> 31) != 0);
        }
#pragma unroll 1
        for (int mask = 1; mask < 64; mask <<= 1) acc = butterfly_add(acc, mask);
        const uint32_t k = heavy[4 * h + 2];
        uint32_t last = 0;
        if (lane == 0) {
            st_xyzz(partial, t, acc);
            __threadfence();   // the partial is visible device-wide before the task counts as done
            last = atomicAdd(&heavy[4 * h + 3], 1u) == k - 1 ? 1u : 0u;
        }
        last = (uint32_t)__shfl((int)last, 0);
        if (!last) continue;
        __threadfence();       // ... and whoever closes the bucket sees every task's partial
        const uint32_t b = heavy[4 * h], t0 = heavy[4 * h + 1];
        acc = G1Xyzz::inf();
        for (uint32_t u = lane; u < k; u += 64) acc = g1_add(acc, ld_xyzz(partial, t0 + u));
#pragma unroll 1
        for (int mask = 1; mask < 64; mask <<= 1) acc = butterfly_add(acc, mask);
        if (lane == 0) st_xyzz(buckets, b, g1_add(ld_xyzz(buckets, b), acc));
    }
}

void launch_msm_accum(const uint32_t* points, const uint32_t* offsets, const uint32_t* sorted, const uint32_t* order,
                      uint32_t nbuckets, uint32_t cap, bool init, uint32_t lanes, uint32_t split, uint32_t* buckets,
                      hipStream_t s) {
    const uint32_t in = init ? 1u : 0u;
    if (lanes <= 1) {
        const dim3 grid((nbuckets + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS), block(MSM_ACC_THREADS);
        hipLaunchKernelGGL(msm_accum_kernel, grid, block, 0, s, points, offsets, sorted, order, nbuckets, cap, in, buckets);
        return;
    }
    // slots [0, split) with `lanes` lanes, the rest with lanes / 2; split * lanes must fill whole workgroups
    split = split >= nbuckets ? nbuckets : split / (MSM_ACC_THREADS / lanes) * (MSM_ACC_THREADS / lanes);
    const uint64_t threads = (uint64_t)split * lanes + (uint64_t)(nbuckets - split) * (lanes / 2);
    const dim3 grid((unsigned)((threads + MSM_ACC_THREADS - 1) / MSM_ACC_THREADS)), block(MSM_ACC_THREADS);
    switch (lanes) {
        case 2: hipLaunchKernelGGL(msm_accum_ml_kernel<2>, grid, block, 0, s, points, offsets, sorted, order, nbuckets, cap, in, split, buckets); break;
        case 4: hipLaunchKernelGGL(msm_accum_ml_kernel<4>, grid, block, 0, s, points, offsets, sorted, order, nbuckets, cap, in, split, buckets); break;
        case 8: hipLaunchKernelGGL(msm_accum_ml_kernel<8>, grid, block, 0, s, points, offsets, sorted, order, nbuckets, cap, in, split, buckets); break;
        default: hipLaunchKernelGGL(msm_accum_ml_kernel<16>, grid, block, 0, s, points, offsets, sorted, order, nbuckets, cap, in, split, buckets); break;
    }
}
void launch_msm_heavy(const uint32_t* points, const uint32_t* sorted, const uint32_t* hist516, uint32_t* heavy,
                      const uint32_t* tasks, uint32_t* partial, uint32_t* buckets, hipStream_t s) {
    hipLaunchKernelGGL(msm_heavy_kernel, dim3(MSM_HEAVY_GRID), dim3(MSM_ACC_THREADS), 0, s, points, sorted, hist516, heavy, tasks, partial,
                       buckets);
}

}  // namespace ty
