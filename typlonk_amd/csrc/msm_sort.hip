// MSM staging kernels: infinity marking, scalar -> signed window digits + histogram, exclusive scan,
// counting-sort scatter.  See msm_common.hpp for the overall MSM structure.
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

// SRS upload: arkworks residues (R = 2^384, 12 words per coordinate) -> the internal packed form
// (R = 2^390, canonical); the C-ABI's separate infinity flags fold into the (0, 0) encoding.
__global__ void msm_convert_points_kernel(uint32_t* pts, const uint8_t* inf, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* p = pts + i * 24;
    if (inf && inf[i]) {
        for (int w = 0; w < 24; ++w) p[w] = 0;
        return;
    }
    uint32_t w[12];
    for (int k = 0; k < 2; ++k) {
        for (int j = 0; j < 12; ++j) w[j] = p[12 * k + j];
        st_fq(p + 12 * k, fq30_from_ark(w));
    }
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
//
// The top window holds only t = 255 - c*(W-1) scalar bits (plus the carry), i.e. 2^t distinct
// digits: taken as is, its buckets would each receive m/2^t entries -- up to m/2 in one bucket --
// while every other window's buckets receive m/2^(c-1).  Its entries are therefore spread over
// V = 2^top_v "virtual copies" of each digit's bucket (copy = i mod V), so that the top window fills
// the same 2^(c-1) buckets as evenly as the others; the reduction weighs bucket k of that window
// by (k >> top_v) + 1.
__global__ __launch_bounds__(MSM_THREADS) void msm_digits_kernel(const Fr* scalars, uint64_t m, uint32_t c,
                                                                 uint32_t W, uint32_t top_v, uint32_t* keys,
                                                                 uint32_t* counts) {
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
            const uint32_t bucket = (j + 1 == W) ? j * B + ((d - 1) << top_v) + ((uint32_t)i & ((1u << top_v) - 1))
                                                 : j * B + d - 1;
            key = bucket | (neg << 31);
            atomicAdd(&counts[bucket], 1u);
        }
        keys[(uint64_t)j * m + i] = key;
    }
}

// ---- exclusive scan of `n` counters (three launches) -------------------------------------------

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


void launch_convert_points(uint32_t* pts, const uint8_t* inf, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(msm_convert_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pts, inf, n);
}
void launch_msm_digits(const Fr* scalars, uint64_t m, uint32_t c, uint32_t W, uint32_t top_v, uint32_t* keys,
                       uint32_t* counts, hipStream_t s) {
    hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((m + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, s,
                       scalars, m, c, W, top_v, keys, counts);
}
void launch_scan(const uint32_t* counts, uint64_t n, uint32_t* block_sums, uint32_t* offsets, uint32_t* cursor,
                 hipStream_t s) {
    const uint32_t nblk = (uint32_t)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblk), dim3(256), 0, s, counts, n, block_sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, s, block_sums, nblk);
    hipLaunchKernelGGL(scan_finish_kernel, dim3(nblk), dim3(256), 0, s, counts, n, block_sums, offsets, cursor);
}
void launch_msm_scatter(const uint32_t* keys, uint64_t m, uint64_t total, uint32_t* cursor, uint32_t* sorted,
                        hipStream_t s) {
    hipLaunchKernelGGL(msm_scatter_kernel, dim3((unsigned)((total + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                       s, keys, m, total, cursor, sorted);
}

}  // namespace ty
