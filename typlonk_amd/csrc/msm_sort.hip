// MSM staging kernels: SRS conversion, and the bucket sort of the (window, term) pairs -- the two-level segmented
// counting sort every table-mode MSM takes (level 1 through the LDS, level 2 one workgroup per segment, bucket schedule
// included; rebuilt in round 6, profiles/r06_ab_sort.txt) and the atomic counting sort kept for shapes it cannot take.
// See msm_common.hpp for the overall MSM structure.
#include "launch.hpp"
#include <algorithm>
#include "msm_common.hpp"

namespace ty {

// SRS upload: arkworks residues (R = 2^384, 12 words per coordinate) -> the internal packed form
// (R = 2^390, canonical); the C-ABI's separate infinity flags fold into the (0, 0) encoding.
__global__ void msm_convert_points_kernel(uint32_t* pts, const uint8_t* inf, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* p = pts + i * PT_WORDS;
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
// (also clears `zero` [0, nzero): the bucket-schedule counters the segment sort adds into two launches later -- a
// hipMemsetAsync of 2 KB costs two 7-us fill kernels on the stream)
__global__ __launch_bounds__(256) void scan_top_kernel(uint32_t* block_sums, uint32_t nblocks, uint32_t* zero, uint32_t nzero) {
    __shared__ uint32_t buf[256];
    __shared__ uint32_t running;
    for (uint32_t i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;
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

// ---- segmented counting sort (no global atomics) ------------------------------------------------------
// Bucket id = (window j, bucket-in-window b).  Level 1 partitions the W*m entries into
// nseg = W << hb segments keyed by (j, b >> lb); level 2 sorts every segment by the low lb <= 8 bits
// inside one workgroup.  Level 1 is the classic radix-sort scheme: every workgroup histograms its
// chunk of MSM_CHUNK scalars in LDS and publishes the column blk_hist[seg * nblk + blk]; one
// exclusive scan over that matrix gives each (segment, workgroup) its private output range, so
// the scatter needs only LDS atomics (for the rank inside the workgroup).
// Level-1 entry: idx (23 bits) | sign << 23 | (b & (2^lb - 1)) << 24     (m <= 2^23)
// scalars per workgroup: 2048 (8 per thread) up to 2^20 terms, growing with m beyond that so the
// workgroup x segment matrix (nblk * nseg counters) stays bounded instead of growing like m^2
// threads per workgroup of the level-1 passes (512 and 1024 were measured: no gain)
inline uint32_t msm_seg1_threads() { return 256; }
// scalars per thread of the level-1 passes: 8 from 2^20 terms on; a short MSM (an index shard) gets fewer, so that its
// level-1 launches still have 512 workgroups -- at 2^17 terms 8 per thread is 64 workgroups and 34 + 39 us for the
// histogram and the scatter, 1 per thread 512 workgroups (profiles/r03_shard_timeline.txt)
inline uint32_t msm_seg1_per_thread(uint64_t m) {
    // (2^19-term chunks keep 8: their scatter, 4096 segments wide, wants long runs per workgroup and segment)
    return m <= (1u << 17) ? 1u : (m <= (1u << 18) ? 2u : 8u);
}
inline uint32_t msm_chunk_for(uint64_t m) {
    uint32_t chunk = msm_seg1_per_thread(m) * msm_seg1_threads();
    while (((uint64_t)chunk << 9) < m) chunk <<= 1;  // at most 512 workgroups
    return chunk;
}

struct MsmShape {
    uint32_t c, W, top_v, hb, lb, nseg, nblk, chunk;
    // level-1 entry layout: [i : ibits][j : jbits][sign : 1][low bucket bits : lb]
    uint32_t ibits, jbits;
    // fixed-base table mode (tlen != 0): base (j, i) lives at gather index j * tlen + i and all windows
    // share one bucket set (nsets = 1)
    uint32_t tlen, nsets;
    // centred scalars: k > (r - 1)/2 is replaced by r - k with every digit's sign flipped, so |k| < 2^254 and
    // c = 17 needs 15 windows instead of 16, c = 15 17 instead of 18 (launch.hpp, msm_windows)
    uint32_t centred;
    // > 0: the kernels of this sort raise their wavefronts' issue priority (s_setprio).  Set for a sort that runs BESIDE an
    // accumulation (an overlapped chunk, a queued MSM): its few, short wavefronts then get the issue slots they ask for
    // instead of the ones two accumulation wavefronts per SIMD leave over.
    uint32_t prio;
};
__device__ __forceinline__ void msm_sort_prio(const MsmShape& sh) {
    if (sh.prio) __builtin_amdgcn_s_setprio(2);
}

// Column of workgroup `blk` in the workgroup x segment matrix = its place inside every segment's output range.
// Workgroups are dealt to the eight XCDs round-robin (blk mod 8), each XCD with its own L2: with the columns ordered by
// XCD first, the runs that are NEIGHBOURS in memory come from workgroups that share an L2, so a cache line that two runs
// straddle is completed there instead of leaving two L2s as two partial writes.
__device__ __forceinline__ uint32_t msm_seg_col(const MsmShape& sh, uint32_t blk) {
    return (sh.nblk & 7u) ? blk : (blk & 7u) * (sh.nblk >> 3) + (blk >> 3);
}

__device__ __forceinline__ uint32_t msm_seg_of(const MsmShape& sh, uint32_t j, uint32_t b) {
    if (sh.tlen) return b >> sh.lb;
    return (j << sh.hb) | (b >> sh.lb);
}

// canonical scalar -> (bucket-in-window, sign) for every window, in window order
template <class F>
__device__ __forceinline__ void msm_for_each_digit(const uint32_t (&v)[8], const MsmShape& sh, uint32_t i, F&& emit) {
    const uint32_t B = 1u << (sh.c - 1);
    uint32_t carry = 0;
    for (uint32_t j = 0; j < sh.W; ++j) {
        const uint32_t o = j * sh.c;
        uint32_t d = (o < 256 ? msm_bits(v, o, sh.c) : 0u) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > B) {
            d = (1u << sh.c) - d;
            neg = 1;
            carry = 1;
        }
        if (d != 0) {
            const uint32_t b = (j + 1 == sh.W) ? ((d - 1) << sh.top_v) + (i & ((1u << sh.top_v) - 1)) : d - 1;
            emit(j, b, neg);
        }
    }
}

// The same with the window width a compile-time constant (the table windows 15 / 17 / 20): bit offsets, word indices and
// shifts fold into the instructions -- one funnel shift and a mask per digit where the run-time form selects two words
// out of eight (16 compares + selects) and shifts 64 bits: ~10 instead of ~50 instructions per digit, 13-18 digits per
// scalar, in BOTH level-1 passes.  C = 0: the run-time form.
template <uint32_t C, class F>
__device__ __forceinline__ void msm_for_each_digit_c(const uint32_t (&v)[8], const MsmShape& sh, uint32_t i, F&& emit) {
    if constexpr (C == 0) {
        msm_for_each_digit(v, sh, i, emit);
    } else {
        constexpr uint32_t WMAX = (256 + C - 1) / C;
        constexpr uint32_t B = 1u << (C - 1);
        uint32_t carry = 0;
#pragma unroll
        for (uint32_t j = 0; j < WMAX; ++j) {
            if (j < sh.W) {
                const uint32_t o = j * C, w = o >> 5, sft = o & 31;
                uint32_t x = v[w] >> sft;
                if (sft + C > 32 && w + 1 < 8) x |= v[w + 1] << (32 - sft);
                uint32_t d = (x & ((1u << C) - 1)) + carry;
                uint32_t neg = 0;
                carry = 0;
                if (d > B) {
                    d = (1u << C) - d;
                    neg = 1;
                    carry = 1;
                }
                if (d != 0) {
                    const uint32_t b = (j + 1 == sh.W) ? ((d - 1) << sh.top_v) + (i & ((1u << sh.top_v) - 1)) : d - 1;
                    emit(j, b, neg);
                }
            }
        }
    }
}

// canonical value of a scalar from its two raw 16-byte halves; centred != 0: min(k, r - k) instead, returns 1 when it is
// r - k (signs flip)
struct MsmRawScalar {
    uint4 a, b;
};
__device__ __forceinline__ MsmRawScalar msm_load_raw(const Fr* scalars, uint64_t i) {
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + i);
    MsmRawScalar r;
    r.a = sp[0];
    r.b = sp[1];
    return r;
}
__device__ __forceinline__ uint32_t msm_canon(const MsmRawScalar& raw, uint32_t centred, uint32_t (&out)[8]) {
    const uint4 a = raw.a, b = raw.b;
    Fr s;
    s.v[0] = a.x; s.v[1] = a.y; s.v[2] = a.z; s.v[3] = a.w;
    s.v[4] = b.x; s.v[5] = b.y; s.v[6] = b.z; s.v[7] = b.w;
    s = fe_from_mont(s);
    uint32_t flip = 0;
    if (centred) {
        const Fr n = fe_neg(s);  // r - k (0 for k = 0)
        bool lt = false;         // n < s, from the top limb down
#pragma unroll
        for (int k = 0; k < 8; ++k) lt = (n.v[k] != s.v[k]) ? (n.v[k] < s.v[k]) : lt;
        if (lt) {
            s = n;
            flip = 1;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = s.v[k];
    return flip;
}
// The level-1 passes walk `per` scalars per thread (base + tid + nt * e).  One wavefront per SIMD is all the LDS staging
// leaves room for, so nothing hides a load's latency but the thread's own work: the NEXT scalar is requested before the
// current one's digits are cut.  body(i, canonical words, flip).
template <class F>
__device__ __forceinline__ void msm_walk_scalars(const Fr* scalars, uint64_t m, uint64_t base, uint32_t per, uint32_t centred, F&& body) {
    const uint32_t nt = blockDim.x;
    uint64_t i = base + threadIdx.x;
    MsmRawScalar cur{};
    if (per && i < m) cur = msm_load_raw(scalars, i);
    for (uint32_t e = 0; e < per; ++e) {
        const uint64_t nxt = i + nt;
        MsmRawScalar ahead{};
        if (e + 1 < per && nxt < m) ahead = msm_load_raw(scalars, nxt);
        if (i < m) {
            uint32_t v[8];
            const uint32_t flip = msm_canon(cur, centred, v);
            body((uint32_t)i, v, flip);
        }
        cur = ahead;
        i = nxt;
    }
}

// blk_cnt != nullptr: also the workgroup's own row of counts, contiguous (the staged scatter scans it for its LDS layout)
template <uint32_t C>
__global__ __launch_bounds__(1024) void msm_seg_hist_kernel(const Fr* scalars, uint64_t m, MsmShape sh,
                                                            uint32_t* blk_hist, uint32_t* blk_cnt) {
    extern __shared__ uint32_t seg_h[];
    msm_sort_prio(sh);
    const uint32_t nt = blockDim.x;
    for (uint32_t s = threadIdx.x; s < sh.nseg; s += nt) seg_h[s] = 0;
    __syncthreads();
    msm_walk_scalars(scalars, m, (uint64_t)blockIdx.x * sh.chunk, sh.chunk / nt, sh.centred,
                     [&](uint32_t i, const uint32_t (&v)[8], uint32_t) {
                         msm_for_each_digit_c<C>(v, sh, i, [&](uint32_t j, uint32_t b, uint32_t) {
                             atomicAdd(&seg_h[msm_seg_of(sh, j, b)], 1u);
                         });
                     });
    __syncthreads();
    const uint32_t col = msm_seg_col(sh, blockIdx.x);
    for (uint32_t s = threadIdx.x; s < sh.nseg; s += nt) {
        const uint32_t n = seg_h[s];
        blk_hist[(uint64_t)s * sh.nblk + col] = n;
        if (blk_cnt) blk_cnt[(uint64_t)blockIdx.x * sh.nseg + s] = n;
    }
}

// One launch instead of the three of the exclusive scan over the whole workgroup x segment matrix (short MSMs: three
// 4.6-us launches between two 10-us kernels).  Workgroup s turns row s of the matrix into its exclusive prefix over the
// workgroups (blk_base[s * nblk + b] = entries of segment s that workgroups < b hold) and publishes the row total
// seg_tot[s]; the consumers add the segment's start, an exclusive scan of the <= 16K totals that each of their workgroups
// redoes in LDS (msm_seg_starts).  Workgroup 0 also clears the bucket-schedule counters (`zero`).
__global__ __launch_bounds__(256) void msm_seg_prefix_kernel(const uint32_t* __restrict__ blk_hist, uint32_t nblk,
                                                             uint32_t* blk_base, uint32_t* seg_tot, uint32_t* zero,
                                                             uint32_t nzero, uint32_t prio) {
    __shared__ uint32_t buf[256];
    __shared__ uint32_t running;
    if (prio) __builtin_amdgcn_s_setprio(2);
    {   // every workgroup clears its slice
        const uint32_t per = (nzero + gridDim.x - 1) / gridDim.x;
        for (uint32_t i = blockIdx.x * per + threadIdx.x; i < min((blockIdx.x + 1) * per, nzero); i += 256) zero[i] = 0;
    }
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const uint64_t row = (uint64_t)blockIdx.x * nblk;
    for (uint32_t base = 0; base < nblk; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblk ? blk_hist[row + i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            uint32_t t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = buf[threadIdx.x], r = running;
        if (i < nblk) blk_base[row + i] = r + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) running = r + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) seg_tot[blockIdx.x] = running;
}

// start[s] = sum of seg_tot[0..s) for every segment, in LDS (nseg <= 16K), by all threads of the workgroup
__device__ __forceinline__ void msm_seg_starts(const uint32_t* __restrict__ seg_tot, uint32_t nseg, uint32_t* start,
                                               uint32_t* scratch /* blockDim.x words */) {
    const uint32_t nt = blockDim.x, per = (nseg + nt - 1) / nt;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nseg);
    uint32_t sum = 0;
    for (uint32_t s = lo; s < hi; ++s) sum += seg_tot[s];
    scratch[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < nt; off <<= 1) {
        const uint32_t t = threadIdx.x >= off ? scratch[threadIdx.x - off] : 0;
        __syncthreads();
        scratch[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = scratch[threadIdx.x] - sum;
    for (uint32_t s = lo; s < hi; ++s) {
        start[s] = run;
        run += seg_tot[s];
    }
    __syncthreads();
}

// two exclusive scans of nseg words each in one pass (sa[] of a[], sb[] of b[]); scratch: 2 * blockDim.x words
__device__ __forceinline__ void msm_seg_starts2(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t nseg,
                                                uint32_t* sa, uint32_t* sb, uint32_t* scratch) {
    const uint32_t nt = blockDim.x, per = (nseg + nt - 1) / nt;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nseg);
    uint32_t suma = 0, sumb = 0;
    for (uint32_t s = lo; s < hi; ++s) {
        suma += a[s];
        sumb += b[s];
    }
    scratch[threadIdx.x] = suma;
    scratch[nt + threadIdx.x] = sumb;
    __syncthreads();
    for (uint32_t off = 1; off < nt; off <<= 1) {
        const uint32_t ta = threadIdx.x >= off ? scratch[threadIdx.x - off] : 0;
        const uint32_t tb = threadIdx.x >= off ? scratch[nt + threadIdx.x - off] : 0;
        __syncthreads();
        scratch[threadIdx.x] += ta;
        scratch[nt + threadIdx.x] += tb;
        __syncthreads();
    }
    uint32_t runa = scratch[threadIdx.x] - suma, runb = scratch[nt + threadIdx.x] - sumb;
    __syncthreads();   // (scratch may be the caller's output area)
    for (uint32_t s = lo; s < hi; ++s) {
        sa[s] = runa;
        sb[s] = runb;
        runa += a[s];
        runb += b[s];
    }
    __syncthreads();
}

// seg_tot == nullptr: blk_base holds absolute positions (the three-launch scan); else row prefixes + segment totals
// (the direct form: every entry goes straight to its place in global memory, 4 bytes at a time.  Kept for the shapes whose
// staging area does not fit the LDS; the staged form below is the one the table-mode MSMs take)
template <uint32_t C>
__global__ __launch_bounds__(1024) void msm_seg_scatter_kernel(const Fr* scalars, uint64_t m, MsmShape sh,
                                                               const uint32_t* blk_base, const uint32_t* seg_tot,
                                                               uint32_t* seg_start, uint32_t* entries) {
    extern __shared__ uint32_t seg_sm[];
    uint32_t* cur = seg_sm;             // running position of this workgroup inside each segment
    const uint32_t nt = blockDim.x;
    const uint32_t col = msm_seg_col(sh, blockIdx.x);
    if (seg_tot) {
        msm_seg_starts(seg_tot, sh.nseg, cur, seg_sm + sh.nseg);
        if (blockIdx.x == 0)   // level 2 reads every segment's range from here instead of summing the totals again
            for (uint32_t s = threadIdx.x; s < sh.nseg; s += nt) {
                seg_start[2 * s] = cur[s];
                seg_start[2 * s + 1] = cur[s] + seg_tot[s];
            }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < sh.nseg; s += nt) cur[s] += blk_base[(uint64_t)s * sh.nblk + col];
    } else {
        for (uint32_t s = threadIdx.x; s < sh.nseg; s += nt) cur[s] = blk_base[(uint64_t)s * sh.nblk + col];
    }
    __syncthreads();
    const uint32_t lmask = (1u << sh.lb) - 1;
    msm_walk_scalars(scalars, m, (uint64_t)blockIdx.x * sh.chunk, sh.chunk / nt, sh.centred,
                     [&](uint32_t i, const uint32_t (&v)[8], uint32_t flip) {
                         msm_for_each_digit_c<C>(v, sh, i, [&](uint32_t j, uint32_t b, uint32_t neg) {
                             const uint32_t pos = atomicAdd(&cur[msm_seg_of(sh, j, b)], 1u);
                             neg ^= flip;
                             entries[pos] = i | ((sh.jbits ? j : 0u) << sh.ibits) | (neg << (sh.ibits + sh.jbits)) |
                                            ((b & lmask) << (sh.ibits + sh.jbits + 1));
                         });
                     });
}

// The staged form (round 6).  The direct form above writes 4 bytes wherever an entry belongs: W entries per scalar into
// nseg different runs, each run of a workgroup ~13 entries long and filled over the whole lifetime of the kernel -- the
// lines leave the L2 partially written (profiles/r05_pmc_traffic.json: 168,821 KiB written for 27 MB of entries, 6.2 x).
// Here the workgroup first counting-sorts its entries by segment INSIDE the LDS -- its own row of the count matrix
// (blk_cnt, from the histogram pass) scanned gives every segment's place in the staging area, an LDS atomic the rank --
// and then copies every run out in one piece: a quarter wavefront (16 lanes) per run, so a store instruction covers four
// runs of <= 64 contiguous bytes, and a run's cache lines are written once.
// LDS: nseg running ranks | nseg local starts | nseg global starts | the staging area (chunk * W entries).
template <uint32_t C>
__global__ __launch_bounds__(512) void msm_seg_scatter_staged_kernel(const Fr* scalars, uint64_t m, MsmShape sh,
                                                                     const uint32_t* __restrict__ blk_base,
                                                                     const uint32_t* __restrict__ seg_tot,
                                                                     const uint32_t* __restrict__ blk_cnt, uint32_t* seg_start,
                                                                     uint32_t* entries) {
    extern __shared__ uint32_t seg_sm[];
    msm_sort_prio(sh);
    uint32_t* rank = seg_sm;
    uint32_t* loc = seg_sm + sh.nseg;
    uint32_t* gst = seg_sm + 2 * sh.nseg;
    uint32_t* stage = seg_sm + 3 * sh.nseg;
    const uint32_t nt = blockDim.x, tid = threadIdx.x;
    const uint32_t col = msm_seg_col(sh, blockIdx.x);
    // global start of every segment (exclusive scan of the segment totals) and local start of every segment in the staging
    // area (exclusive scan of this workgroup's own counts), both in ONE pass; the staging area doubles as the scans' scratch
    msm_seg_starts2(seg_tot, blk_cnt + (uint64_t)blockIdx.x * sh.nseg, sh.nseg, gst, loc, stage);
    if (blockIdx.x == 0)   // level 2 reads every segment's range from here instead of summing the totals again
        for (uint32_t s = tid; s < sh.nseg; s += nt) {
            seg_start[2 * s] = gst[s];
            seg_start[2 * s + 1] = gst[s] + seg_tot[s];
        }
    __syncthreads();
    for (uint32_t s = tid; s < sh.nseg; s += nt) {
        gst[s] += blk_base[(uint64_t)s * sh.nblk + col];
        rank[s] = 0;
    }
    __syncthreads();
    const uint32_t lmask = (1u << sh.lb) - 1;
    msm_walk_scalars(scalars, m, (uint64_t)blockIdx.x * sh.chunk, sh.chunk / nt, sh.centred,
                     [&](uint32_t i, const uint32_t (&v)[8], uint32_t flip) {
                         msm_for_each_digit_c<C>(v, sh, i, [&](uint32_t j, uint32_t b, uint32_t neg) {
                             const uint32_t sg = msm_seg_of(sh, j, b);
                             const uint32_t r = atomicAdd(&rank[sg], 1u);
                             neg ^= flip;
                             stage[loc[sg] + r] = i | ((sh.jbits ? j : 0u) << sh.ibits) | (neg << (sh.ibits + sh.jbits)) |
                                                  ((b & lmask) << (sh.ibits + sh.jbits + 1));
                         });
                     });
    __syncthreads();
    // copy-out: a quarter wavefront (16 lanes) per run, four runs in flight per quarter (their LDS look-ups are issued
    // together); rank[] now holds the run lengths
    const uint32_t q = tid >> 4, l = tid & 15u, nq = nt >> 4;
    for (uint32_t sg0 = 4 * q; sg0 < sh.nseg; sg0 += 4 * nq) {
        uint32_t len[4], src[4], dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t sg = min(sg0 + u, sh.nseg - 1);
            len[u] = sg0 + u < sh.nseg ? rank[sg] : 0u;
            src[u] = loc[sg];
            dst[u] = gst[sg];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (uint32_t k = l; k < len[u]; k += 16) entries[dst[u] + k] = stage[src[u] + k];
    }
}

// Level 2, one workgroup per segment, in two launches around the one global dependency of the bucket schedule (the size
// histogram over ALL buckets must be complete before any bucket can be placed in the size-sorted order):
//   msm_seg_count_kernel  count the segment's entries by their low bits -> counts[], offsets[] of its 2^lb buckets, the
//                         size histogram (hist514[0..255]) and the task lists of heavy buckets (what order_hist_kernel
//                         does for the atomic sort); publishes the segment's start for the second launch.
//   msm_seg_place_kernel  re-read the segment (L2), rank every entry inside its bucket with an LDS atomic and write the
//                         final `sorted` array -- AND place the segment's buckets into order[] (what order_fused_kernel did in
//                         a launch of its own, 27 us of latency per accumulation launch: every workgroup scans the complete
//                         256-bin histogram for itself and claims its run inside each bin from a global cursor).  The claim's
//                         round trip to the L2 hides under the placement loop.
// 256, 512 or 1024 threads: all of them walk the segment, the first 256 own the 2^lb <= 256 buckets.
// Schedule counters of the segmented sort.  Words [0, 516) keep the layout of the atomic sort's hist514 ([512] heavy buckets,
// [513] tasks -- msm_heavy_kernel reads those); the size histogram and the claim cursors of the SEGMENTED sort are kept in
// MSM_SCHED_REPLICAS copies behind them, replica r = segment mod R at word 1024 + 512 r (256 bins + 256 cursors, a KiB apart).
// Why: ~25 size bins are hot, adjacent words of ONE cache line, and every one of the 2048-4096 level-2 workgroups adds
// to each of them -- 51 K atomics on one line, which the L2 retires one per clock: ~24 us per launch, the whole run time of
// order_fused_kernel in rounds 1-5 and most of msm_seg_count / msm_seg_place.  Sixteen lines take them sixteen at a time.
constexpr uint32_t MSM_PLACE_STAGE = 5120;   // entries of a segment msm_seg_place_kernel assembles in the LDS (20 KiB; mean 3328)
constexpr uint32_t MSM_SCHED_REPLICAS = 16;
constexpr uint32_t MSM_SCHED_WORDS = 1024 + 512 * MSM_SCHED_REPLICAS;
uint32_t msm_sched_words() { return MSM_SCHED_WORDS; }

// inclusive scan over the values of threads 0..255 (four wavefronts): shuffles inside a wavefront, one LDS hand-over
// between them -- two barriers where the Hillis-Steele form over LDS takes sixteen.  Every thread of the workgroup calls it.
__device__ __forceinline__ uint32_t msm_scan256_incl(uint32_t v, uint32_t* wsum /* 4 words of LDS */) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(x, off);
        if ((int)lane >= off) x += t;
    }
    if (tid < 256 && lane == 63) wsum[wave] = x;
    __syncthreads();
    if (tid < 256)
        for (uint32_t w = 0; w < wave; ++w) x += wsum[w];
    __syncthreads();
    return x;
}

__global__ __launch_bounds__(1024) void msm_seg_count_kernel(const uint32_t* __restrict__ entries,
                                                             const uint32_t* __restrict__ blk_base,
                                                             const uint32_t* __restrict__ seg_tot, MsmShape sh,
                                                             uint32_t total_slot, uint32_t* counts, uint32_t* offsets,
                                                             uint32_t* seg_start, uint32_t cap, uint32_t* ohist, uint32_t* heavy,
                                                             uint32_t* tasks) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    msm_sort_prio(sh);
    const uint32_t s = blockIdx.x, nt = blockDim.x, tid = threadIdx.x;
    uint32_t start, end;
    if (seg_tot) {  // the fused row-prefix form: the level-1 scatter has published every segment's range
        start = seg_start[2 * s];
        end = seg_start[2 * s + 1];
    } else {
        start = blk_base[(uint64_t)s * sh.nblk];
        end = (s + 1 < sh.nseg) ? blk_base[(uint64_t)(s + 1) * sh.nblk] : blk_base[total_slot];
        if (tid == 0) {
            seg_start[2 * s] = start;
            seg_start[2 * s + 1] = end;
        }
    }
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t low_sh = sh.ibits + sh.jbits + 1;
    // (eight loads in flight per thread: a thread walks ~13 entries of its segment, and one load per trip of a loop whose
    // trip count the compiler does not know was one exposed L2 round trip each)
    for (uint32_t e0 = start + tid; e0 < end; e0 += 8 * nt) {
        uint32_t x[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) x[u] = e0 + u * nt < end ? entries[e0 + u * nt] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            if (e0 + u * nt < end) atomicAdd(&hist[x[u] >> low_sh], 1u);
    }
    __syncthreads();
    const uint32_t mine = tid < 256 ? hist[tid] : 0;
    const uint32_t excl = msm_scan256_incl(mine, wsum) - mine;
    if (tid < 256) hist[tid] = 0;
    const uint32_t nlow = 1u << sh.lb;
    if (tid < nlow) {
        const uint32_t bucket = s * nlow + tid;
        counts[bucket] = mine;
        offsets[bucket] = start + excl;
        if (s + 1 == sh.nseg && tid + 1 == nlow) offsets[bucket + 1] = end;
        if (mine > cap) {  // heavy bucket: the accumulate kernel leaves it alone, tasks of tl entries take all of it
            const uint32_t tl = msm_task_len(mine);
            const uint32_t k = (mine + tl - 1) / tl;
            const uint32_t hi = atomicAdd(&ohist[512], 1u);
            const uint32_t t0 = atomicAdd(&ohist[513], k);
            heavy[4 * hi] = bucket;
            heavy[4 * hi + 1] = t0;
            heavy[4 * hi + 2] = k;
            heavy[4 * hi + 3] = 0;   // tasks done (msm_heavy_kernel: the wavefront that finishes the last one folds the bucket)
            uint32_t b = start + excl;
            const uint32_t e = start + excl + mine;
            for (uint32_t t = 0; t < k; ++t) {
                tasks[3 * (t0 + t)] = b;
                tasks[3 * (t0 + t) + 1] = min(b + tl, e);
                tasks[3 * (t0 + t) + 2] = hi;
                b += tl;
            }
        }
    }
    __syncthreads();
    // schedule histogram: size bins of this segment's buckets, one global atomic per occupied bin
    if (tid < nlow) atomicAdd(&hist[min(mine, 255u)], 1u);
    __syncthreads();
    if (tid < 256) {
        const uint32_t nbin = hist[tid];
        if (nbin) atomicAdd(&ohist[1024 + 512 * (s & (MSM_SCHED_REPLICAS - 1)) + tid], nbin);
    }
}

__global__ __launch_bounds__(1024) void msm_seg_place_kernel(const uint32_t* __restrict__ entries,
                                                             const uint32_t* __restrict__ seg_start, MsmShape sh,
                                                             const uint32_t* __restrict__ counts,
                                                             const uint32_t* __restrict__ offsets, uint32_t* sorted,
                                                             uint32_t* sched, uint32_t* order) {
    __shared__ uint32_t cur[256];    // running rank inside each bucket
    __shared__ uint32_t pref[256];   // the bucket's start inside the segment
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t h[256];      // this segment's buckets per size bin
    __shared__ uint32_t blk[256];    // where this segment's run inside each bin starts in order[]
    // The segment's output, staged: an entry's place is data-dependent, so written straight to global memory every lane of
    // a store instruction touches a different cache line -- 6.8 M four-byte requests per launch, the kernel's run time.
    // The segment's slice of `sorted` is contiguous (~3300 entries), so it is assembled here and copied out in whole lines.
    // A longer segment (adversarial scalars) writes directly.
    __shared__ uint32_t seg_out[MSM_PLACE_STAGE];
    msm_sort_prio(sh);
    const uint32_t s = blockIdx.x, nt = blockDim.x, tid = threadIdx.x;
    const uint32_t start = seg_start[2 * s], end = seg_start[2 * s + 1];
    const uint32_t nlow = 1u << sh.lb;
    uint32_t key = 0, rank = 0, bucket = 0, v = 0;
    const uint32_t k = 255 - tid;  // thread t scans size key 255 - t (descending order)
    const uint32_t rep = s & (MSM_SCHED_REPLICAS - 1);
    uint32_t before = 0;   // buckets of this size in the replicas before this segment's
    if (tid < 256) {
        cur[tid] = 0;
        h[tid] = 0;
#pragma unroll
        for (uint32_t r = 0; r < MSM_SCHED_REPLICAS; ++r) {
            const uint32_t c = sched[1024 + 512 * r + k];
            v += c;
            before += r < rep ? c : 0u;
        }
        if (tid < nlow) {
            bucket = s * nlow + tid;
            pref[tid] = offsets[bucket] - start;
            key = min(counts[bucket], 255u);
        }
    }
    __syncthreads();
    if (tid < nlow) rank = atomicAdd(&h[key], 1u);
    const uint32_t larger = msm_scan256_incl(v, wsum) - v;   // buckets with a larger size key (the scan has two barriers: h[] is complete)
    // claim this segment's run inside every occupied bin (the answer is needed only after the placement loop)
    if (tid < 256) blk[k] = larger + before + (h[k] ? atomicAdd(&sched[1024 + 512 * rep + 256 + k], h[k]) : 0u);
    const uint32_t low_sh = sh.ibits + sh.jbits + 1;
    const bool staged = end - start <= MSM_PLACE_STAGE;
    for (uint32_t e0 = start + tid; e0 < end; e0 += 8 * nt) {   // (eight loads in flight, as in the count kernel)
        uint32_t xs[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) xs[u] = e0 + u * nt < end ? entries[e0 + u * nt] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            if (e0 + u * nt < end) {
                const uint32_t x = xs[u];
                const uint32_t b = x >> low_sh;
                const uint32_t r = atomicAdd(&cur[b], 1u);
                const uint32_t i = x & ((1u << sh.ibits) - 1);
                const uint32_t j = (x >> sh.ibits) & ((1u << sh.jbits) - 1);
                const uint32_t neg = (x >> (sh.ibits + sh.jbits)) & 1u;
                const uint32_t val = (j * sh.tlen + i) | (neg << 31);
                if (staged) seg_out[pref[b] + r] = val;
                else sorted[start + pref[b] + r] = val;
            }
        }
    }
    __syncthreads();
    if (staged)
        for (uint32_t e = tid; e < end - start; e += nt) sorted[start + e] = seg_out[e];
    if (tid < nlow) order[blk[key] + rank] = bucket;
}

// ---- bucket schedule: order[] = bucket ids sorted by population, largest first --------------------
// A wave's 64 lanes run their buckets in lock step, so its time is the LARGEST of its 64 bucket
// sizes; handing each wave buckets of (nearly) equal size removes that imbalance.  Counting sort on
// min(count, 255): LDS-privatised histogram per block, one global atomic per (block, bin).
//
// Heavy buckets.  A bucket with more than `cap` entries (adversarial inputs produce them -- equal scalars, tiny
// scalars, ... -- and a table window count whose top window is a few bits wide) would be summed by a single
// thread; instead the accumulate kernel skips it and ALL its entries are cut into tasks of msm_task_len(count)
// entries that msm_heavy_kernel spreads over the whole chip, one wavefront per task.  hist[512] = number of heavy buckets, hist[513] = number of tasks;
// heavy[4h..4h+3] = (bucket, first task, task count, tasks done); tasks[3t..3t+2] = (begin, end in `sorted`, h).
__global__ __launch_bounds__(256) void order_hist_kernel(const uint32_t* counts, const uint32_t* offsets, uint32_t n,
                                                         uint32_t cap, uint32_t* hist, uint32_t* heavy, uint32_t* tasks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g < n) {
        const uint32_t cnt = counts[g];
        atomicAdd(&h[min(cnt, 255u)], 1u);
        if (cnt > cap) {
            const uint32_t tl = msm_task_len(cnt);
            const uint32_t k = (cnt + tl - 1) / tl;
            const uint32_t hi = atomicAdd(&hist[512], 1u);
            const uint32_t t0 = atomicAdd(&hist[513], k);
            heavy[4 * hi] = g;
            heavy[4 * hi + 1] = t0;
            heavy[4 * hi + 2] = k;
            heavy[4 * hi + 3] = 0;   // tasks done (msm_heavy_kernel: the wavefront that finishes the last one folds the bucket)
            uint32_t b = offsets[g];
            const uint32_t e = offsets[g] + cnt;
            for (uint32_t t = 0; t < k; ++t) {
                tasks[3 * (t0 + t)] = b;
                tasks[3 * (t0 + t) + 1] = min(b + tl, e);
                tasks[3 * (t0 + t) + 2] = hi;
                b += tl;
            }
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// Bucket schedule in one launch: every workgroup scans the (complete) 256-bin size histogram
// for itself and claims its run inside each bin from a zero-initialised global cursor (hist514[256..511])
__global__ __launch_bounds__(256) void order_fused_kernel(const uint32_t* counts, uint32_t n, const uint32_t* hist,
                                                          uint32_t* gcur, uint32_t* order) {
    __shared__ uint32_t buf[256];
    __shared__ uint32_t h[256];
    __shared__ uint32_t blk[256];
    const uint32_t k = 255 - threadIdx.x;  // thread t scans size key 255 - t (descending order)
    const uint32_t v = hist[k];
    buf[threadIdx.x] = v;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    const uint32_t base_k = buf[threadIdx.x] - v;  // buckets with a larger size key
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = 0, rank = 0;
    if (g < n) {
        key = min(counts[g], 255u);
        rank = atomicAdd(&h[key], 1u);
    }
    __syncthreads();
    blk[k] = base_k + (h[k] ? atomicAdd(&gcur[k], h[k]) : 0u);
    __syncthreads();
    if (g < n) order[blk[key] + rank] = g;
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
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, s, block_sums, nblk, (uint32_t*)nullptr, 0u);
    hipLaunchKernelGGL(scan_finish_kernel, dim3(nblk), dim3(256), 0, s, counts, n, block_sums, offsets, cursor);
}
// plain exclusive scan of n counters into out[0..n] (out[n] = total); scratch: ceil(n/2048) u32
void launch_exclusive_scan(const uint32_t* in, uint64_t n, uint32_t* block_sums, uint32_t* out, uint32_t* out2,
                           uint32_t* zero, uint32_t nzero, hipStream_t s) {
    const uint32_t nblk = (uint32_t)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblk), dim3(256), 0, s, in, n, block_sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, s, block_sums, nblk, zero, nzero);
    hipLaunchKernelGGL(scan_finish_kernel, dim3(nblk), dim3(256), 0, s, in, n, block_sums, out, out2);
}

uint32_t msm_segsort_blocks(uint64_t m) { return (uint32_t)((m + msm_chunk_for(m) - 1) / msm_chunk_for(m)); }

// LDS of the staged scatter: three arrays of nseg words + the staging area; it is taken when that fits the 160 KiB of a CU
// (table mode c = 20: 2048 scalars x 13 windows = 104 KiB + 24-48 KiB) and the fused row-prefix form applies
static size_t msm_staged_lds(const MsmShape& sh) {
    return ((size_t)3 * sh.nseg + std::max<size_t>((size_t)sh.chunk * sh.W, 1024)) * sizeof(uint32_t);   // (>= the scans' scratch)
}
template <uint32_t C>
static bool msm_staged_raise_lds() {
    static int state[64] = {};   // per device ordinal: 0 = not tried, 1 = raised, -1 = refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[dev] == 0)
        state[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(msm_seg_scatter_staged_kernel<C>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : -1;
    if (state[dev] < 0) (void)hipGetLastError();
    return state[dev] > 0;
}

template <uint32_t C>
static void launch_msm_segsort_c(const Fr* scalars, uint64_t m, const MsmShape& sh, uint32_t* blk_hist, uint32_t* blk_base,
                                 uint32_t* scan_scratch, uint32_t* blk_cnt, uint32_t* seg_start, uint32_t* entries,
                                 uint32_t* counts, uint32_t* offsets, uint32_t* sorted, uint32_t cap, uint32_t* hist514,
                                 uint32_t* heavy, uint32_t* tasks, uint32_t* order, int staged_mode, uint32_t l1_threads,
                                 hipStream_t s) {
    const uint64_t nmat = (uint64_t)sh.nseg * sh.nblk;
    const uint32_t nt1 = msm_seg1_threads();
    // threads of the staged scatter and of the histogram beside it: 256 (one wavefront per SIMD) or 512 where the chunk divides
    const uint32_t nts = (l1_threads == 512 && sh.chunk % 512 == 0) ? 512u : 256u;
    // the fused row-prefix form scans the segment totals in LDS next to the scatter's cursors (nseg + nt1 words); wider
    // segment sets take the three-launch scan of the whole workgroup x segment matrix
    const bool fused = sh.nseg <= 8192;
    uint32_t* seg_tot = fused ? scan_scratch : nullptr;
    const size_t lds_staged = msm_staged_lds(sh);
    const bool staged = staged_mode != 0 && fused && blk_cnt && lds_staged <= 160 * 1024 &&
                        (lds_staged <= 64 * 1024 || msm_staged_raise_lds<C>());
    hipLaunchKernelGGL(msm_seg_hist_kernel<C>, dim3(sh.nblk), dim3(staged ? nts : nt1), sh.nseg * sizeof(uint32_t), s, scalars, m, sh,
                       blk_hist, staged ? blk_cnt : (uint32_t*)nullptr);
    if (fused)
        hipLaunchKernelGGL(msm_seg_prefix_kernel, dim3(sh.nseg), dim3(256), 0, s, blk_hist, sh.nblk, blk_base, seg_tot, hist514,
                           MSM_SCHED_WORDS, sh.prio);
    else
        launch_exclusive_scan(blk_hist, nmat, scan_scratch, blk_base, blk_hist /* second copy unused */, hist514, MSM_SCHED_WORDS, s);
    if (staged)
        hipLaunchKernelGGL(msm_seg_scatter_staged_kernel<C>, dim3(sh.nblk), dim3(nts), lds_staged, s, scalars, m, sh, blk_base, seg_tot,
                           blk_cnt, seg_start, entries);
    else
        hipLaunchKernelGGL(msm_seg_scatter_kernel<C>, dim3(sh.nblk), dim3(nt1), (fused ? sh.nseg + nt1 : sh.nseg) * sizeof(uint32_t),
                           s, scalars, m, sh, blk_base, seg_tot, seg_start, entries);
    // long segments (short MSMs with few of them) get more threads per segment
    const uint64_t seg_len = (uint64_t)sh.W * m / sh.nseg;
    const uint32_t nt2 = seg_len >= 4096 ? 1024u : (seg_len >= 1536 ? 512u : 256u);
    hipLaunchKernelGGL(msm_seg_count_kernel, dim3(sh.nseg), dim3(nt2), 0, s, entries, blk_base, seg_tot, sh, (uint32_t)nmat,
                       counts, offsets, seg_start, cap, hist514, heavy, tasks);
    hipLaunchKernelGGL(msm_seg_place_kernel, dim3(sh.nseg), dim3(nt2), 0, s, entries, seg_start, sh, counts, offsets, sorted,
                       hist514, order);
}

// The whole bucket sort of one chunk of terms, bucket schedule (order[]) included.  staged_mode: 0 = the direct level-1
// scatter everywhere (TYPLONK_MSM_SCATTER=direct, the A/B reference), else the LDS-staged one where it fits.
void launch_msm_segsort(const Fr* scalars, uint64_t m, uint32_t c, uint32_t W, uint32_t top_v, uint32_t hb,
                        uint32_t ibits, uint32_t tlen, uint32_t nsets, uint32_t* blk_hist, uint32_t* blk_base,
                        uint32_t* scan_scratch, uint32_t* blk_cnt, uint32_t* seg_start, uint32_t* entries, uint32_t* counts,
                        uint32_t* offsets, uint32_t* sorted, uint32_t cap, uint32_t* hist514, uint32_t* heavy, uint32_t* tasks,
                        uint32_t* order, bool centred, int staged_mode, uint32_t l1_threads, bool beside_accum, hipStream_t s) {
    MsmShape sh;
    sh.centred = centred ? 1u : 0u;
    sh.prio = beside_accum ? 1u : 0u;
    sh.c = c;
    sh.W = W;
    sh.top_v = top_v;
    sh.hb = hb;
    sh.lb = c - 1 - hb;
    sh.ibits = ibits;
    sh.jbits = tlen ? (W > 16 ? 5 : 4) : 0;
    sh.tlen = tlen;
    sh.nsets = nsets;
    sh.nseg = (tlen ? nsets : W) << hb;
    sh.chunk = msm_chunk_for(m);
    sh.nblk = (uint32_t)((m + sh.chunk - 1) / sh.chunk);
#define TY_SEGSORT(C) launch_msm_segsort_c<C>(scalars, m, sh, blk_hist, blk_base, scan_scratch, blk_cnt, seg_start, entries, counts, \
                                              offsets, sorted, cap, hist514, heavy, tasks, order, staged_mode, l1_threads, s)
    // the table windows get the constant-width digit extraction; every other width the run-time form
    if (c == 20) TY_SEGSORT(20);
    else if (c == 17) TY_SEGSORT(17);
    else if (c == 15) TY_SEGSORT(15);
    else TY_SEGSORT(0);
#undef TY_SEGSORT
}

// bucket schedule after the ATOMIC counting sort (the segmented sort builds it itself, msm_seg_place_kernel)
void launch_bucket_order(const uint32_t* counts, const uint32_t* offsets, uint32_t n, uint32_t cap, uint32_t* hist514,
                         uint32_t* order, uint32_t* heavy, uint32_t* tasks, hipStream_t s) {
    // hist514: histogram (256) + running bases (256) + heavy-bucket and task counters (2)
    const uint32_t nblk = (n + 255) / 256;
    (void)hipMemsetAsync(hist514, 0, 516 * sizeof(uint32_t), s);
    hipLaunchKernelGGL(order_hist_kernel, dim3(nblk), dim3(256), 0, s, counts, offsets, n, cap, hist514, heavy, tasks);
    hipLaunchKernelGGL(order_fused_kernel, dim3(nblk), dim3(256), 0, s, counts, n, hist514, hist514 + 256, order);
}
void launch_msm_scatter(const uint32_t* keys, uint64_t m, uint64_t total, uint32_t* cursor, uint32_t* sorted,
                        hipStream_t s) {
    hipLaunchKernelGGL(msm_scatter_kernel, dim3((unsigned)((total + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                       s, keys, m, total, cursor, sorted);
}

}  // namespace ty
