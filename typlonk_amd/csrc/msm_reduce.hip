// MSM bucket reduction: per bucket set sum_k w(k)*B_k by the row/column split (launch.hpp) + wavefront butterflies.
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

// ---- row/column reduction (see launch.hpp) -------------------------------------------------------
// 1. msm_rc_partial_kernel  two roles in one launch.  Row role: thread sums 2^llc consecutive buckets of
//    one row.  Column role: thread sums the buckets of 2^lhc consecutive rows in one column (adjacent
//    lanes = adjacent columns, so the loads coalesce) and stores hchunk-fastest so step 2 reads runs.
// 2. msm_fold_seq_kernel    each thread sums 2^lseq consecutive partials, then a __shfl_xor butterfly
//    over `lanes` lanes -> one row sum / column sum.
// 3. msm_rc_bits_kernel     one wavefront per (set, kind, weight bit, 64-item chunk): butterfly sum of the
//    items whose weight has that bit set.
// 4. msm_rc_final_kernel    one wavefront per (set, kind, bit): sum over the chunks.
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

struct FoldSeg {
    const uint32_t* in;
    uint32_t* out;
    uint32_t threads, lseq, lanes;
};
// (workgroups of four wavefronts, one per SIMD of a CU, each wavefront working alone: with one-wavefront workgroups the
// dispatcher stacks two wavefronts of a one-per-SIMD launch on one SIMD now and then and leaves another idle -- seen on
// msm_rc2_sums_kernel, 112 us instead of 76)
__global__ __launch_bounds__(256) void msm_fold_seq_kernel(FoldSeg a, FoldSeg b, uint32_t waves_a, uint32_t waves) {
    const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= waves) return;   // whole wavefronts only
    const bool first = wv < waves_a;
    const FoldSeg& g = first ? a : b;
    const uint32_t t = (wv - (first ? 0u : waves_a)) * 64 + (threadIdx.x & 63u);
    G1Xyzz v = G1Xyzz::inf();
    if (t < g.threads) {
        const uint64_t base = (uint64_t)t << g.lseq;
        v = ld_xyzz(g.in, base);
        for (uint32_t i = 1; i < (1u << g.lseq); ++i) v = g1_add(v, ld_xyzz(g.in, base + i));
    }
    v = butterfly_reduce(v, g.lanes);
    if (t < g.threads && (t & (g.lanes - 1)) == 0) st_xyzz(g.out, t / g.lanes, v);
}

// sums: row sums [set][hi] (nsets * R points) followed by column sums [set][lo]
__global__ __launch_bounds__(64) void msm_rc_bits_kernel(const uint32_t* __restrict__ sums, RcShape sh, uint32_t rw,
                                                         uint32_t cw, uint32_t* bitsum) {
    const uint32_t nbr_max = sh.ch + 1, nbc_max = sh.cl + 1;
    const uint32_t wps = nbr_max * rw + nbc_max * cw;
    const uint32_t set = blockIdx.x / wps;
    uint32_t r = blockIdx.x % wps, kind = 0, bit, chunk;
    if (r < nbr_max * rw) {
        bit = r / rw;
        chunk = r % rw;
    } else {
        r -= nbr_max * rw;
        kind = 1;
        bit = r / cw;
        chunk = r % cw;
    }
    uint32_t nbr, nbc, shift;
    rc_bits(sh, set, &nbr, &nbc, &shift);
    if (bit >= (kind ? nbc : nbr)) return;
    const uint32_t idx = chunk * 64 + threadIdx.x;
    const uint32_t n_items = kind ? (1u << sh.cl) : (1u << sh.ch);
    G1Xyzz v = G1Xyzz::inf();
    if (idx < n_items && ((rc_weight(sh, set, kind, idx) >> bit) & 1u)) {
        const uint64_t at = kind ? ((uint64_t)sh.nsets << sh.ch) + ((uint64_t)set << sh.cl) + idx : ((uint64_t)set << sh.ch) + idx;
        v = ld_xyzz(sums, at);
    }
    for (uint32_t mask = 1; mask < 64; mask <<= 1) v = butterfly_add(v, (int)mask);
    if (threadIdx.x == 0) st_xyzz(bitsum, (uint64_t)((set * 2 + kind) * RC_NB + bit) * 64 + chunk, v);
}

__global__ __launch_bounds__(64) void msm_rc_final_kernel(const uint32_t* __restrict__ bitsum, RcShape sh, uint32_t rw,
                                                          uint32_t cw, uint32_t* out) {
    const uint32_t bit = blockIdx.x % RC_NB, kind = (blockIdx.x / RC_NB) & 1u, set = blockIdx.x / (2 * RC_NB);
    uint32_t nbr, nbc, shift;
    rc_bits(sh, set, &nbr, &nbc, &shift);
    if (bit >= (kind ? nbc : nbr)) return;
    const uint32_t count = kind ? cw : rw;
    G1Xyzz v = G1Xyzz::inf();
    if (threadIdx.x < count) v = ld_xyzz(bitsum, (uint64_t)blockIdx.x * 64 + threadIdx.x);
    for (uint32_t mask = 1; mask < count; mask <<= 1) v = butterfly_add(v, (int)mask);
    if (threadIdx.x == 0) st_xyzz(out, blockIdx.x, v);
}

// ---- the same reduction in two launches (bucket sets of >= 2^12 buckets) ---------------------------------------------
// The four kernels above are a chain of ~24 dependent additions behind four launches; a short MSM (an 8-way shard)
// spends more time in that chain than in its bucket additions.  Here
//   1. msm_rc2_sums_kernel    one wavefront per run of 64 * 2^s buckets of a row (row role) or of a column (column
//                             role): every lane sums 2^s buckets, a 6-step butterfly (butterfly_add: 7 multiplication
//                             times per step) folds the 64 lanes -> P partials per row / column.  s is chosen so that
//                             the launch has ~1024 wavefronts, one per SIMD (2^16 buckets: s = 1, 14 + 42
//                             multiplication times per wavefront);
//   2. msm_rc2_planes_kernel  one workgroup per (set, kind, weight bit): the partials of the items whose weight has
//                             that bit set are butterfly-summed per wavefront, then across the wavefronts through LDS.
// Same bit planes out as msm_rc_final_kernel (the host finish does not change).
// (workgroups of four wavefronts: one per SIMD of a CU.  With one-wavefront workgroups the dispatcher was seen to stack
// two of the 1024 wavefronts on one SIMD and leave another idle, 112 us instead of 76.)
__global__ __launch_bounds__(256) void msm_rc2_sums_kernel(const uint32_t* __restrict__ buckets, RcShape sh, uint32_t sr,
                                                           uint32_t sc, uint32_t nwave_row, uint32_t nwave, uint32_t* prow,
                                                           uint32_t* pcol) {
    const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (wv >= nwave) return;   // whole wavefronts only
    G1Xyzz v;
    if (wv < nwave_row) {
        const uint64_t base = ((uint64_t)wv * 64 + lane) << sr;   // a run never leaves its row: 64 * 2^sr | C
        v = ld_xyzz(buckets, base);
        for (uint32_t i = 1; i < (1u << sr); ++i) v = g1_add(v, ld_xyzz(buckets, base + i));
    } else {
        const uint32_t u = wv - nwave_row;
        const uint32_t lpc = sh.ch - 6 - sc;                                      // log2 partials per column
        const uint32_t rchunk = u & ((1u << lpc) - 1), col = u >> lpc;            // col = set << cl | lo
        const uint32_t set = col >> sh.cl, lo = col & ((1u << sh.cl) - 1);
        const uint32_t row0 = ((rchunk * 64 + lane) << sc);
        const uint64_t first = ((uint64_t)set << sh.c1) + ((uint64_t)row0 << sh.cl) + lo;
        v = ld_xyzz(buckets, first);
        for (uint32_t i = 1; i < (1u << sc); ++i) v = g1_add(v, ld_xyzz(buckets, first + ((uint64_t)i << sh.cl)));
    }
    v = butterfly_reduce(v, 64);
    if (lane == 0) {
        if (wv < nwave_row) st_xyzz(prow, wv, v);
        else st_xyzz(pcol, wv - nwave_row, v);
    }
}

constexpr int RC2_THREADS = 256;  // 4 wavefronts, one per SIMD: two per SIMD double every butterfly step's latency (measured
                                  // 115 us with 512 threads on 512 partials)
__global__ __launch_bounds__(RC2_THREADS) void msm_rc2_planes_kernel(const uint32_t* __restrict__ prow,
                                                                     const uint32_t* __restrict__ pcol, RcShape sh,
                                                                     uint32_t lpr, uint32_t lpc, uint32_t* out) {
    __shared__ uint32_t xch[RC2_THREADS / 64][52];
    const uint32_t bit = blockIdx.x % RC_NB, kind = (blockIdx.x / RC_NB) & 1u, set = blockIdx.x / (2 * RC_NB);
    uint32_t nbr, nbc, shift;
    rc_bits(sh, set, &nbr, &nbc, &shift);
    if (bit >= (kind ? nbc : nbr)) return;
    const uint32_t litems = kind ? sh.cl : sh.ch, lp = kind ? lpc : lpr;
    const uint32_t* src = kind ? pcol : prow;
    const uint32_t np = 1u << (litems + lp);
    const uint64_t sbase = (uint64_t)set << (litems + lp);
    G1Xyzz v = G1Xyzz::inf();
    if (rc_set_v(sh, set) == 0) {
        // Only the items whose weight has the bit: rows weigh `item`, columns `item + 1` (launch.hpp), so the j-th selected
        // weight is j with a 1 inserted at position `bit` (columns: the single weight 2^cl for bit = cl).  Half the
        // partials, so that 2^16 buckets need one sweep of the workgroup instead of two (one dependent addition less).
        const bool top = kind && bit == sh.cl;
        const uint32_t nsel = (top ? 1u : (1u << (litems - 1))) << lp;
        for (uint32_t idx = threadIdx.x; idx < nsel; idx += blockDim.x) {
            const uint32_t j = idx >> lp, part = idx & ((1u << lp) - 1);
            const uint32_t w = top ? (1u << sh.cl) : (((j >> bit) << (bit + 1)) | (1u << bit) | (j & ((1u << bit) - 1)));
            const uint32_t item = w - kind;
            v = g1_add(v, ld_xyzz(src, sbase + ((uint64_t)item << lp) + part));
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < np; idx += blockDim.x) {
            const uint32_t item = idx >> lp;
            if ((rc_weight(sh, set, kind, item) >> bit) & 1u) v = g1_add(v, ld_xyzz(src, sbase + idx));
        }
    }
    v = butterfly_reduce(v, 64);
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            xch[wave][i] = v.x.v[i];
            xch[wave][13 + i] = v.y.v[i];
            xch[wave][26 + i] = v.zz.v[i];
            xch[wave][39 + i] = v.zzz.v[i];
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // the wavefronts' sums, each on a lane PAIR (lanes 2w, 2w + 1), so that the steps across them are four-lane steps too
    v = G1Xyzz::inf();
    if (threadIdx.x < 2 * nwaves) {
        const uint32_t w = threadIdx.x >> 1;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            v.x.v[i] = xch[w][i];
            v.y.v[i] = xch[w][13 + i];
            v.zz.v[i] = xch[w][26 + i];
            v.zzz.v[i] = xch[w][39 + i];
        }
    }
#pragma unroll 1
    for (int mask = 2; mask < (int)(2 * nwaves); mask <<= 1) v = butterfly_add4(v, mask);
    if (threadIdx.x == 0) st_xyzz(out, blockIdx.x, v);
}

bool msm_rc2_ok(const RcShape& sh) { return sh.cl >= 6 && sh.ch >= 6; }
// scratch: prow holds nsets << (c1 - 6 - sr) partials, pcol nsets << (c1 - 6 - sc); both <= nsets << (c1 - 6)
void launch_msm_rc2_reduce(const uint32_t* buckets, const RcShape& sh, uint32_t* prow, uint32_t* pcol, uint32_t* out,
                           uint32_t log_waves, hipStream_t s) {
    uint32_t lnb = sh.c1;
    for (uint32_t n = sh.nsets; n > 1; n >>= 1) ++lnb;        // ~log2 of all buckets
    const uint32_t want = lnb > 5 + log_waves ? lnb - 5 - log_waves : 0;   // ~2^log_waves wavefronts in the first launch (10: one per SIMD)
    const uint32_t sr = want < sh.cl - 6 ? want : sh.cl - 6, sc = want < sh.ch - 6 ? want : sh.ch - 6;
    const uint32_t nwave_row = sh.nsets << (sh.c1 - 6 - sr), nwave_col = sh.nsets << (sh.c1 - 6 - sc);
    hipLaunchKernelGGL(msm_rc2_sums_kernel, dim3((nwave_row + nwave_col + 3) / 4), dim3(256), 0, s, buckets, sh, sr, sc, nwave_row,
                       nwave_row + nwave_col, prow, pcol);
    const uint32_t lpr = sh.cl - 6 - sr, lpc = sh.ch - 6 - sc;
    // partials one (kind, bit) sums: half of the items' (the ones whose weight has the bit) unless a set has virtual copies
    uint32_t np = 1u << ((sh.ch + lpr) > (sh.cl + lpc) ? (sh.ch + lpr) : (sh.cl + lpc));
    if (sh.top_v == 0 && np > 64) np >>= 1;
    const uint32_t threads = np < 64 ? 64u : (np > (uint32_t)RC2_THREADS ? (uint32_t)RC2_THREADS : np);
    hipLaunchKernelGGL(msm_rc2_planes_kernel, dim3(sh.nsets * 2 * RC_NB), dim3(threads), 0, s, prow, pcol, sh, lpr, lpc, out);
}

// Plain (multi-set) MSMs: one wavefront per set applies the powers of two -- lane (kind, b) doubles its bit
// plane b (+ shift for rows) times, then a butterfly sums the 2 * RC_NB lanes -> one point per set, so the
// host's Horner over the windows stays c doublings + one addition per window.
__global__ __launch_bounds__(64) void msm_rc_combine_kernel(const uint32_t* __restrict__ planes, RcShape sh,
                                                            uint32_t* set_sums) {
    const uint32_t set = blockIdx.x, kind = threadIdx.x / RC_NB, b = threadIdx.x % RC_NB;
    uint32_t nbr, nbc, shift;
    rc_bits(sh, set, &nbr, &nbc, &shift);
    const bool valid = kind < 2 && b < (kind ? nbc : nbr);
    G1Xyzz v = G1Xyzz::inf();
    if (valid) v = ld_xyzz(planes, (uint64_t)(set * 2 + kind) * RC_NB + b);
    const uint32_t e = valid ? b + (kind ? 0u : shift) : 0u;
    const uint32_t emax = max(nbr ? nbr - 1 + shift : 0u, nbc ? nbc - 1 : 0u);
    for (uint32_t i = 0; i < emax; ++i)
        if (i < e) v = g1_dbl(v);
    for (uint32_t mask = 1; mask < 2 * RC_NB; mask <<= 1) v = butterfly_add(v, (int)mask);
    if (threadIdx.x == 0) st_xyzz(set_sums, set, v);
}

void launch_msm_rc_combine(const uint32_t* planes, const RcShape& sh, uint32_t* set_sums, hipStream_t s) {
    hipLaunchKernelGGL(msm_rc_combine_kernel, dim3(sh.nsets), dim3(64), 0, s, planes, sh, set_sums);
}

void launch_msm_rc_reduce(const uint32_t* buckets, const RcShape& sh, uint32_t* pb, uint32_t* pa, uint32_t* sums,
                          uint32_t* bitsum, uint32_t* out, hipStream_t s) {
    const uint32_t nrow = sh.nsets << (sh.c1 - sh.llc), ncol = sh.nsets << (sh.c1 - sh.lhc);
    const uint32_t nrow_pad = (nrow + 63) & ~63u;
    hipLaunchKernelGGL(msm_rc_partial_kernel, dim3((nrow_pad + ncol + 63) / 64), dim3(64), 0, s, buckets, sh, nrow, nrow_pad,
                       ncol, pb, pa);
    // row partials: 2^(cl - llc) per row; column partials: 2^(ch - lhc) per column
    auto seg = [](const uint32_t* in, uint32_t* out, uint32_t n_in, uint32_t lcnt) {
        FoldSeg g;
        g.in = in;
        g.out = out;
        g.lseq = lcnt < 1 ? lcnt : 1;
        g.lanes = 1u << (lcnt - g.lseq);
        g.threads = n_in >> g.lseq;
        return g;
    };
    uint32_t* rsum = sums;
    uint32_t* csum = sums + ((uint64_t)sh.nsets << sh.ch) * 48;
    const FoldSeg a = seg(pb, rsum, nrow, sh.cl - sh.llc), b = seg(pa, csum, ncol, sh.ch - sh.lhc);
    const uint32_t ba = (a.threads + 63) / 64, bb = (b.threads + 63) / 64;
    hipLaunchKernelGGL(msm_fold_seq_kernel, dim3((ba + bb + 3) / 4), dim3(256), 0, s, a, b, ba, ba + bb);
    if (sh.ch >= 6 && sh.cl >= 6) {
        // the bit planes of the R + C sums in ONE launch (msm_rc2_planes_kernel with one "partial" per row / column):
        // a workgroup per (set, kind, bit) butterfly-sums the selected sums per wavefront and across wavefronts
        uint32_t np = 1u << (sh.ch > sh.cl ? sh.ch : sh.cl);
        if (sh.top_v == 0 && np > 64) np >>= 1;
        const uint32_t threads = np < 64 ? 64u : (np > (uint32_t)RC2_THREADS ? (uint32_t)RC2_THREADS : np);
        hipLaunchKernelGGL(msm_rc2_planes_kernel, dim3(sh.nsets * 2 * RC_NB), dim3(threads), 0, s, rsum, csum, sh, 0u, 0u, out);
        return;
    }
    const uint32_t rw = ((1u << sh.ch) + 63) / 64, cw = ((1u << sh.cl) + 63) / 64;
    const uint32_t wps = (sh.ch + 1) * rw + (sh.cl + 1) * cw;
    hipLaunchKernelGGL(msm_rc_bits_kernel, dim3(sh.nsets * wps), dim3(64), 0, s, sums, sh, rw, cw, bitsum);
    hipLaunchKernelGGL(msm_rc_final_kernel, dim3(sh.nsets * 2 * RC_NB), dim3(64), 0, s, bitsum, sh, rw, cw, out);
}

}  // namespace ty
