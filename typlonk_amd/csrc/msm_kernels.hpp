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
//   5. msm_reduce_kernel   per window sum_k k*B_k: 8-bucket running sums per thread, offset by a
//                          small scalar multiplication, then a wavefront __shfl_xor butterfly of
//                          whole points; msm_fold_kernel repeats the butterfly until one point per
//                          window is left.
// The W window sums go to the host, which applies the 2^(c*j) weights (Horner) and normalises
// to the canonical affine point.  Group addition is commutative and the result is canonical, so
// the non-deterministic order inside a bucket (atomics in step 3) cannot change the output.
#pragma once
#include "g1.hpp"

namespace ty {

constexpr uint32_t MSM_SKIP = 0xffffffffu;
constexpr int MSM_THREADS = 256;
constexpr int MSM_ACC_THREADS = 64;
constexpr int MSM_SEG = 8;  // buckets per reduce thread

__device__ __forceinline__ Fq ld_fq(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    Fq r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w;
    return r;
}
__device__ __forceinline__ void st_fq(uint32_t* p, const Fq& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
    q[2] = make_uint4(r.v[8], r.v[9], r.v[10], r.v[11]);
}
__device__ __forceinline__ G1Affine ld_affine(const uint32_t* pts, uint64_t idx) {
    const uint32_t* p = pts + idx * 24;
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

// Fold the C-ABI's separate infinity flags into the device encoding (0, 0).
__global__ void msm_mark_inf_kernel(uint32_t* pts, const uint8_t* inf, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !inf[i]) return;
    for (int w = 0; w < 24; ++w) pts[i * 24 + w] = 0;
}

// bits [o, o+c) of a 256-bit little-endian integer, c <= 24
__device__ __forceinline__ uint32_t msm_bits(const uint32_t (&v)[8], uint32_t o, uint32_t c) {
    const uint32_t w = o >> 5, sh = o & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lo = (w == (uint32_t)i) ? v[i] : lo;
        hi = (w + 1 == (uint32_t)i) ? v[i] : hi;
    }
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> sh;
    return (uint32_t)x & ((1u << c) - 1);
}

// keys[j*m + i] = bucket id (j*B + |d| - 1) | sign << 31, or MSM_SKIP for a zero digit
__global__ __launch_bounds__(MSM_THREADS) void msm_digits_kernel(const Fr* scalars, uint64_t m, uint32_t c,
                                                                 uint32_t W, uint32_t* keys, uint32_t* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= m) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + i);
    const uint4 a = sp[0], b = sp[1];
    Fr s;
    s.v[0] = a.x; s.v[1] = a.y; s.v[2] = a.z; s.v[3] = a.w;
    s.v[4] = b.x; s.v[5] = b.y; s.v[6] = b.z; s.v[7] = b.w;
    s = fe_from_mont(s);
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    for (uint32_t j = 0; j < W; ++j) {
        const uint32_t o = j * c;
        uint32_t d = (o < 256 ? msm_bits(s.v, o, c) : 0u) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > B) {
            d = (1u << c) - d;
            neg = 1;
            carry = 1;
        }
        uint32_t key = MSM_SKIP;
        if (d != 0) {
            const uint32_t bucket = j * B + d - 1;
            key = bucket | (neg << 31);
            atomicAdd(&counts[bucket], 1u);
        }
        keys[(uint64_t)j * m + i] = key;
    }
}

// ---- exclusive scan of `n` counters (three launches) -------------------------------------------
constexpr int SCAN_PER_BLOCK = 2048;  // 256 threads x 8

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint32_t* in, uint64_t n, uint32_t* block_sums) {
    __shared__ uint32_t red[256];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 8;
    uint32_t s = 0;
    for (int e = 0; e < 8; ++e)
        if (base + e < n) s += in[base + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}

// single block: exclusive scan of nblocks values in place (nblocks arbitrary, processed in chunks)
__global__ __launch_bounds__(256) void scan_top_kernel(uint32_t* block_sums, uint32_t nblocks) {
    __shared__ uint32_t buf[256];
    __shared__ uint32_t running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            uint32_t t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = buf[threadIdx.x];
        const uint32_t r = running;
        if (i < nblocks) block_sums[i] = r + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) running = r + incl;
        __syncthreads();
    }
}

// offsets[i] = exclusive prefix; cursor[i] = same (scatter positions); offsets[n] = total
__global__ __launch_bounds__(256) void scan_finish_kernel(const uint32_t* in, uint64_t n, const uint32_t* block_sums,
                                                          uint32_t* offsets, uint32_t* cursor) {
    __shared__ uint32_t buf[256];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 8;
    uint32_t v[8], s = 0;
    for (int e = 0; e < 8; ++e) {
        v[e] = base + e < n ? in[base + e] : 0;
        s += v[e];
    }
    buf[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = block_sums[blockIdx.x] + buf[threadIdx.x] - s;
    for (int e = 0; e < 8; ++e) {
        if (base + e < n) {
            offsets[base + e] = run;
            cursor[base + e] = run;
        }
        run += v[e];
        if (base + e + 1 == n) offsets[n] = run;
    }
}

__global__ __launch_bounds__(MSM_THREADS) void msm_scatter_kernel(const uint32_t* keys, uint64_t m, uint64_t total,
                                                                  uint32_t* cursor, uint32_t* sorted) {
    const uint64_t e = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (e >= total) return;
    const uint32_t key = keys[e];
    if (key == MSM_SKIP) return;
    const uint32_t pos = atomicAdd(&cursor[key & 0x7fffffffu], 1u);
    sorted[pos] = (uint32_t)(e % m) | (key & 0x80000000u);
}

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

__device__ __forceinline__ G1Xyzz shfl_xor_point(const G1Xyzz& p, int mask) {
    G1Xyzz r;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        r.x.v[i] = __shfl_xor(p.x.v[i], mask);
        r.y.v[i] = __shfl_xor(p.y.v[i], mask);
        r.zz.v[i] = __shfl_xor(p.zz.v[i], mask);
        r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask);
    }
    return r;
}

// Thread t owns buckets [t*L, (t+1)*L) of the flat (window-major) bucket array, L = min(8, B).
// node value = sum_l (s*L + l + 1) * bucket[l]   with s = t mod (B/L);
// lanes of the same window are then summed with a __shfl_xor butterfly over `group` lanes and
// lane 0 of each group stores one partial.  partials[t / group].
__global__ __launch_bounds__(64) void msm_reduce_kernel(const uint32_t* __restrict__ buckets, uint32_t B, uint32_t L,
                                                        uint32_t nodes_total, uint32_t group, uint32_t cbits,
                                                        uint32_t* partials) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    G1Xyzz v = G1Xyzz::inf();
    if (t < nodes_total) {
        const uint32_t npw = B / L;
        const uint32_t s = t % npw;
        const uint64_t base = (uint64_t)t * L;
        G1Xyzz running = G1Xyzz::inf(), u = G1Xyzz::inf();
        for (uint32_t l = L - 1; l >= 1; --l) {
            running = g1_add(running, ld_xyzz(buckets, base + l));
            u = g1_add(u, running);
        }
        running = g1_add(running, ld_xyzz(buckets, base));
        // (s*L + 1) * S by double-and-add over cbits bits
        const uint32_t kmul = s * L + 1;
        G1Xyzz acc = G1Xyzz::inf();
        for (int bit = (int)cbits - 1; bit >= 0; --bit) {
            acc = g1_dbl(acc);
            if ((kmul >> bit) & 1) acc = g1_add(acc, running);
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

}  // namespace ty
