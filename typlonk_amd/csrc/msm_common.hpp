// Pippenger bucket MSM over BLS12-381 G1 (replaces the per-term double-and-add of
// KzgScheme::evaluate_in_s, /root/reference/kzg/src/lib.rs:41-54).
//
//   1. sort (msm_sort.hip)     the (window, term) pairs by bucket.  Table mode and every shape up to 2^23 terms: the
//                              SEGMENTED counting sort -- msm_seg_hist / msm_seg_prefix / msm_seg_scatter_staged (level 1:
//                              Montgomery -> canonical, ark-ff into_repr, the conversion lib.rs:49 performs per term; signed
//                              c-bit window slicing; entries partitioned into segments of <= 256 buckets through the LDS, no
//                              global atomics), msm_seg_count / msm_seg_place (level 2: one workgroup per segment; bucket
//                              counts, offsets, the final (point index, sign) list AND the bucket schedule order[]).
//                              Beyond that: msm_digits_kernel + scan + msm_scatter_kernel (global atomics) + order_* kernels.
//   2. msm_accum_kernel        one thread per bucket in population order: XYZZ accumulator in registers, mixed additions
//                              of the bucket's affine points gathered from the resident SRS (msm_accum.hip).
//   3. msm_rc_* kernels        per bucket set sum_k w(k)*B_k by the row/column split (launch.hpp): plain row and column
//                              sums, then bit planes of the R + C weighted sums via wavefront __shfl butterflies of whole
//                              points (msm_reduce.hip).
// The W window sums go to the host, which applies the 2^(c*j) weights (Horner) and normalises
// to the canonical affine point.  Group addition is commutative and the result is canonical, so
// the non-deterministic order inside a bucket (the atomics of the sort) cannot change the output.
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

// ---- wavefront exchange of whole points (bucket reduction, multi-lane bucket accumulation) ----------------------
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
// Bounds are those of g1_add (g1.hpp).  An identity operand selects the other point at the end; equal points are
// doubled and opposite points give the identity there too (wave-uniform branch, taken only when some pair needs it).
__device__ __forceinline__ G1Xyzz butterfly_add(const G1Xyzz& v, int mask) {
    const bool lower = (threadIdx.x & (uint32_t)mask) == 0;
    const bool v_inf = v.is_inf();
    const bool o_inf = __shfl_xor((int)v_inf, mask) != 0;
    // Only the partner's ZZ and ZZZ are fetched up front (its X and Y are needed only when this lane holds the identity,
    // below): 26 registers fewer alive through the four steps -- what keeps the multi-lane accumulation kernel at two
    // wavefronts per SIMD.
    Fq30 u1, s1, pd, rd, zz_own;
    {
        const Fq30 ozz = shfl_xor_fq(v.zz, mask), ozzz = shfl_xor_fq(v.zzz, mask);
        // step 1
        const Fq30 u_own = fq30_mul(v.x, ozz);                               // < 1.01
        const Fq30 s_own = fq30_mul(v.y, ozzz);                              // < 1.01
        zz_own = fq30_mul(fq_sel(lower, v.zz, v.zzz), fq_sel(lower, ozz, ozzz));   // ZZ12 (lower) | ZZZ12 (upper)  < 1.01
        const Fq30 u_oth = shfl_xor_fq(u_own, mask), s_oth = shfl_xor_fq(s_own, mask);
        u1 = fq_sel(lower, u_own, u_oth);
        s1 = fq_sel(lower, s_own, s_oth);
        pd = fq30_sub_lazy<2>(fq_sel(lower, u_oth, u_own), u1);              // U2 - U1  < 3.1
        rd = fq30_sub_lazy<2>(fq_sel(lower, s_oth, s_own), s1);              // S2 - S1  < 3.1
    }
    // step 2
    const Fq30 sq_own = fq30_sqr(fq_sel(lower, pd, rd));                     // PP | RR  < 1.02
    const Fq30 sq_oth = shfl_xor_fq(sq_own, mask);
    const Fq30 pp = fq_sel(lower, sq_own, sq_oth), rr = fq_sel(lower, sq_oth, sq_own);
    // equal or opposite points (P = 0): the formulas below do not apply to that pair; settled at the end
    const bool exc = !v_inf && !o_inf && fq30_is_zero_mod(pp);
    const bool same = fq30_is_zero_mod(rr);                                  // ... and R = 0: the same point
    // step 3
    const Fq30 m3_own = fq30_mul(fq_sel(lower, pd, u1), pp);                 // PPP | Q  < 1.01
    const Fq30 m3_oth = shfl_xor_fq(m3_own, mask);
    const Fq30 ppp = fq_sel(lower, m3_own, m3_oth), q = fq_sel(lower, m3_oth, m3_own);
    G1Xyzz out;
    out.x = fq30_sub2_lazy<4>(rr, ppp, fq30_mulk_lazy<2>(q));               // < 5.1
    const Fq30 t = fq30_sub_lazy<6>(q, out.x);                               // < 7.1
    // step 4: each lane multiplies its own ZZ12 / ZZZ12 (no exchange needed for those)
    const Fq30 y_own = fq30_mul(fq_sel(lower, rd, s1), fq_sel(lower, t, ppp));   // R*T | S1*PPP
    const Fq30 z_own = fq30_mul(zz_own, fq_sel(lower, pp, ppp));                  // ZZ3 | ZZZ3  < 1.01
    const Fq30 y_oth = shfl_xor_fq(y_own, mask), z_oth = shfl_xor_fq(z_own, mask);
    out.y = fq30_sub_lazy<2>(fq_sel(lower, y_own, y_oth), fq_sel(lower, y_oth, y_own));   // R*T - S1*PPP  < 3.1
    out.zz = fq_sel(lower, z_own, z_oth);
    out.zzz = fq_sel(lower, z_oth, z_own);
    // An identity operand selects the other point (the formulas above then ran on zeros, harmlessly); equal points
    // double, opposite points cancel -- both lanes of such a pair hold the same point, so each settles it alone.
    if (__any(v_inf || o_inf || exc)) {
        const G1Xyzz o = shfl_xor_point(v, mask);
        if (exc) out = same ? g1_dbl(v) : G1Xyzz::inf();
        else if (v_inf) out = o;
        else if (o_inf) out = v;
    }
    return out;
}

// ---- the same step with one addition split over FOUR lanes (mask >= 2) ------------------------------------------------------
// From the second step of a butterfly on, lanes l and l ^ 1 hold the SAME point (both ended the step before with the sum), so
// the two points of a step live on four lanes: A on {q0, q1}, B = the (l ^ mask) side on {q2, q3}.  add-2008-s is four
// multiplications deep (U -> PP -> PPP -> Y3), and with four lanes it is four ROUNDS of one multiplication per lane where
// the lane pair needs seven.  The chains of the bucket reduction (msm_fold_seq, msm_rc2_planes, msm_rc2_sums: one wavefront
// per SIMD or fewer) pay the latency of every multiplication in full, so a step goes from 7 multiplication times + 8 field
// exchanges to 4 + 11.
//   round 1   q0: U1 = X_A ZZ_B    q1: S1 = Y_A ZZZ_B    q2: U2 = X_B ZZ_A    q3: S2 = Y_B ZZZ_A       (1 exchange before)
//   round 2   q0: ZZ12             q1: ZZZ12             q2: PP = P^2         q3: RR = R^2             (P = U2 - U1 on the even,
//             R = S2 - S1 on the odd lanes: both sides can form them after one exchange of the round-1 products)
//   round 3   q0: Q = U1 PP        q1: --                q2: PPP = P PP       q3: --                   (1 exchange: PP <-> ZZ12)
//   round 4   q0: ZZ3 = ZZ12 PP    q1: R (Q - X3)        q2: ZZZ3 = ZZZ12 PPP q3: S1 PPP               (3 exchanges)
// and a gather of X3, Y3 = R(Q - X3) - S1 PPP, ZZ3, ZZZ3 onto all four lanes (5 exchanges).  Bounds: those of butterfly_add.
// Identity operands, equal and opposite points are settled at the end exactly as there (every lane of a side holds the
// whole point, so each lane settles its own copy).  Precondition: v is identical on lanes l and l ^ 1.
__device__ __forceinline__ Fq30 shfl_fq(const Fq30& a, int src_lane) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; ++i) r.v[i] = __shfl(a.v[i], src_lane);
    return r;
}
__device__ __forceinline__ G1Xyzz butterfly_add4(const G1Xyzz& v, int mask) {
    const int lane = (int)(threadIdx.x & 63u);
    const bool odd = (lane & 1) != 0, bside = (lane & mask) != 0;
    const int qbase = lane & ~(mask | 1);
    const int l0 = qbase, l1 = qbase | 1, l2 = qbase | mask, l3 = qbase | mask | 1;   // the lanes holding roles q0..q3
    const bool v_inf = v.is_inf();
    const bool o_inf = __shfl_xor((int)v_inf, mask) != 0;
    // round 1
    const Fq30 send = fq_sel(odd, v.zzz, v.zz);
    const Fq30 recv1 = shfl_xor_fq(send, mask);                               // the other point's ZZ (even lanes) / ZZZ (odd)
    const Fq30 m1 = fq30_mul(fq_sel(odd, v.y, v.x), recv1);                   // U1 | S1 | U2 | S2   < 1.01
    // round 2
    const Fq30 x1 = shfl_xor_fq(m1, mask);                                    // U2 | S2 | U1 | S1
    const Fq30 d = fq30_sub_lazy<2>(fq_sel(bside, m1, x1), fq_sel(bside, x1, m1));   // P (even lanes) | R (odd lanes)   < 3.1
    const Fq30 m2 = fq30_mul(fq_sel(bside, d, send), fq_sel(bside, d, recv1));       // ZZ12 | ZZZ12 | PP | RR   < 1.02
    // equal or opposite points (P = 0); and R = 0: the same point
    const bool z2 = fq30_is_zero_mod(m2);
    const bool exc = !v_inf && !o_inf && (__shfl((int)z2, l2) != 0);
    const bool same = __shfl((int)z2, l3) != 0;
    // round 3 (even lanes; the odd lanes' product is not used)
    const Fq30 x2 = shfl_xor_fq(m2, mask);                                    // PP | RR | ZZ12 | ZZZ12
    const Fq30 m3 = fq30_mul(fq_sel(bside, d, m1), fq_sel(bside, m2, x2));    // Q = U1 PP | -- | PPP = P PP | --   < 1.01
    // round 4
    const Fq30 a = shfl_xor_fq(m3, 1);                                        // -- | Q | -- | PPP
    const Fq30 b = shfl_xor_fq(m3, mask ^ 1);                                 // -- | PPP | -- | Q
    const Fq30 c = shfl_xor_fq(x2, 1);                                        // (q2 takes ZZZ12 from q3)
    // q1: X3 = RR - PPP - 2Q (RR = x2, PPP = b, Q = a), T = Q - X3
    const Fq30 x3 = fq30_sub2_lazy<4>(x2, b, fq30_mulk_lazy<2>(a));           // < 5.1 (on q1; elsewhere unused)
    const Fq30 t = fq30_sub_lazy<6>(a, x3);                                   // < 7.1
    Fq30 opa, opb;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        // q0: ZZ12 PP = m2 x2;  q1: R T = d t;  q2: ZZZ12 PPP = c m3;  q3: S1 PPP = x1 a
        opa.v[i] = bside ? (odd ? x1.v[i] : c.v[i]) : (odd ? d.v[i] : m2.v[i]);
        opb.v[i] = bside ? (odd ? a.v[i] : m3.v[i]) : (odd ? t.v[i] : x2.v[i]);
    }
    const Fq30 m4 = fq30_mul(opa, opb);                                       // ZZ3 | R T | ZZZ3 | S1 PPP   < 1.04
    // gather onto all four lanes
    G1Xyzz out;
    out.x = shfl_fq(x3, l1);
    out.y = fq30_sub_lazy<2>(shfl_fq(m4, l1), shfl_fq(m4, l3));              // R T - S1 PPP   < 3.1
    out.zz = shfl_fq(m4, l0);
    out.zzz = shfl_fq(m4, l2);
    if (__any(v_inf || o_inf || exc)) {
        const G1Xyzz o = shfl_xor_point(v, mask);
        if (exc) out = same ? g1_dbl(v) : G1Xyzz::inf();
        else if (v_inf) out = o;
        else if (o_inf) out = v;
    }
    return out;
}
// a whole butterfly over lanes [0, lanes) of a wavefront: the first step on lane pairs, the others on lane quadruples
__device__ __forceinline__ G1Xyzz butterfly_reduce(G1Xyzz v, uint32_t lanes) {
    if (lanes > 1) v = butterfly_add(v, 1);
#pragma unroll 1
    for (uint32_t mask = 2; mask < lanes; mask <<= 1) v = butterfly_add4(v, (int)mask);
    return v;
}

}  // namespace ty
