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

__device__ __forceinline__ Fq30 shfl_xor_fq(const Fq30& a, int mask) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; ++i) r.v[i] = __shfl_xor(a.v[i], mask);
    return r;
}
__device__ __forceinline__ Fq30 fq_sel(bool c, const Fq30& a, const Fq30& b) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// v <- v + (value of lane ^ mask), on both lanes of every pair: the butterfly step of the reductions.
// The two lanes of a pair split the 12M + 2S of add-2008-s between them (7 multiplication times instead of 14):
// with a = the lower lane's point and b = the upper lane's,
//   step 1  every lane: own.x * other.zz (U1 on the lower lane, U2 on the upper), own.y * other.zzz (S1 / S2);
//           lower: a.zz * b.zz, upper: a.zzz * b.zzz                                  -> exchange
//   step 2  lower: PP = P^2, upper: R^2  (P = U2 - U1, R = S2 - S1)                   -> exchange
//   step 3  lower: PPP = P * PP, upper: Q = U1 * PP                                   -> exchange
//   step 4  lower: R * (Q - X3) and ZZ12 * PP, upper: S1 * PPP and ZZZ12 * PPP        -> exchange
// Bounds are those of g1_add (g1.hpp).  An identity operand selects the other point at the end.  Equal or opposite
// points are rare: if ANY pair of the wavefront meets them, the whole wavefront takes the one-lane g1_add instead.
__device__ __forceinline__ G1Xyzz butterfly_add(const G1Xyzz& v, int mask) {
    const G1Xyzz o = shfl_xor_point(v, mask);
    const bool lower = (threadIdx.x & (uint32_t)mask) == 0;
    // step 1
    const Fq30 u_own = fq30_mul(v.x, o.zz);                              // < 1.01
    const Fq30 s_own = fq30_mul(v.y, o.zzz);                             // < 1.01
    const Fq30 zz_own = fq30_mul(fq_sel(lower, v.zz, v.zzz), fq_sel(lower, o.zz, o.zzz));   // ZZ12 | ZZZ12  < 1.01
    const Fq30 u_oth = shfl_xor_fq(u_own, mask), s_oth = shfl_xor_fq(s_own, mask), zz_oth = shfl_xor_fq(zz_own, mask);
    const Fq30 u1 = fq_sel(lower, u_own, u_oth), u2 = fq_sel(lower, u_oth, u_own);
    const Fq30 s1 = fq_sel(lower, s_own, s_oth), s2 = fq_sel(lower, s_oth, s_own);
    const Fq30 zz12 = fq_sel(lower, zz_own, zz_oth), zzz12 = fq_sel(lower, zz_oth, zz_own);
    const Fq30 pd = fq30_sub_lazy<2>(u2, u1);                            // < 3.1
    const Fq30 rd = fq30_sub_lazy<2>(s2, s1);                            // < 3.1
    // step 2
    const Fq30 sq_own = fq30_sqr(fq_sel(lower, pd, rd));                 // PP | RR  < 1.02
    const Fq30 sq_oth = shfl_xor_fq(sq_own, mask);
    const Fq30 pp = fq_sel(lower, sq_own, sq_oth), rr = fq_sel(lower, sq_oth, sq_own);
    // an identity operand just selects the other point below (the formulas then run on zeros, harmlessly);
    // equal or opposite points need the doubling / identity branches of g1_add
    const bool v_inf = v.is_inf(), o_inf = o.is_inf();
    if (__any(!v_inf && !o_inf && fq30_is_zero_mod(pp))) return g1_add(v, o);
    // step 3
    const Fq30 m3_own = fq30_mul(fq_sel(lower, pd, u1), pp);             // PPP | Q  < 1.01
    const Fq30 m3_oth = shfl_xor_fq(m3_own, mask);
    const Fq30 ppp = fq_sel(lower, m3_own, m3_oth), q = fq_sel(lower, m3_oth, m3_own);
    G1Xyzz out;
    out.x = fq30_sub2_lazy<4>(rr, ppp, fq30_mulk_lazy<2>(q));           // < 5.1
    const Fq30 t = fq30_sub_lazy<6>(q, out.x);                           // < 7.1
    // step 4
    const Fq30 y_own = fq30_mul(fq_sel(lower, rd, s1), fq_sel(lower, t, ppp));       // R*T | S1*PPP
    const Fq30 z_own = fq30_mul(fq_sel(lower, zz12, zzz12), fq_sel(lower, pp, ppp));  // ZZ3 | ZZZ3  < 1.01
    const Fq30 y_oth = shfl_xor_fq(y_own, mask), z_oth = shfl_xor_fq(z_own, mask);
    out.y = fq30_sub_lazy<2>(fq_sel(lower, y_own, y_oth), fq_sel(lower, y_oth, y_own));   // R*T - S1*PPP  < 3.1
    out.zz = fq_sel(lower, z_own, z_oth);
    out.zzz = fq_sel(lower, z_oth, z_own);
    if (v_inf) return o;
    if (o_inf) return v;
    return out;
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
__global__ __launch_bounds__(64) void msm_fold_seq_kernel(FoldSeg a, FoldSeg b, uint32_t blocks_a) {
    const bool first = blockIdx.x < blocks_a;
    const FoldSeg& g = first ? a : b;
    const uint32_t t = (blockIdx.x - (first ? 0u : blocks_a)) * 64 + threadIdx.x;
    G1Xyzz v = G1Xyzz::inf();
    if (t < g.threads) {
        const uint64_t base = (uint64_t)t << g.lseq;
        v = ld_xyzz(g.in, base);
        for (uint32_t i = 1; i < (1u << g.lseq); ++i) v = g1_add(v, ld_xyzz(g.in, base + i));
    }
    for (uint32_t mask = 1; mask < g.lanes; mask <<= 1) v = butterfly_add(v, (int)mask);
    if (t < g.threads && (threadIdx.x & (g.lanes - 1)) == 0) st_xyzz(g.out, t / g.lanes, v);
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
    hipLaunchKernelGGL(msm_fold_seq_kernel, dim3(ba + bb), dim3(64), 0, s, a, b, ba);
    const uint32_t rw = ((1u << sh.ch) + 63) / 64, cw = ((1u << sh.cl) + 63) / 64;
    const uint32_t wps = (sh.ch + 1) * rw + (sh.cl + 1) * cw;
    hipLaunchKernelGGL(msm_rc_bits_kernel, dim3(sh.nsets * wps), dim3(64), 0, s, sums, sh, rw, cw, bitsum);
    hipLaunchKernelGGL(msm_rc_final_kernel, dim3(sh.nsets * 2 * RC_NB), dim3(64), 0, s, bitsum, sh, rw, cw, out);
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
