// Fr on nine unsaturated 30-bit limbs for the NTT butterflies (device only).
//
// HBM and the C ABI keep arkworks' form (8 x 32-bit words of x * 2^256 mod r).  Inside an NTT pass the words are
// re-cut into 9 x 30-bit limbs and every multiplication is a Montgomery multiplication with R' = 2^270:
//     fr30_mul(a, b) = a * b / 2^270 mod r.
// The data are never converted: with a twiddle stored as w * 2^270 mod r the product of a datum d = x * 2^256 is
// d * w, still in arkworks' domain -- only the twiddle tables live in the 2^270 domain (ntt_host.hip builds them with an
// extra factor 2^14).
//
// Why it pays (tools/ubench3, profiles/r02_ubench3_fr30.txt): a column of 9 products of 30-bit limbs plus 9 reduction
// terms stays below 2^64 (bounds below), so a whole column accumulates with v_mad_u64_u32 alone -- no carry
// instruction per partial product as the 8 x 32-bit product scanning needs -- and because r = 1 mod 2^30 the
// Montgomery digit is just the negated low limb (no multiplication by -r^-1).  Additions and subtractions are
// limb-wise without carry propagation, followed by one parallel carry step.
//
// Bounds (checked by tests/test_gpu_ntt.py through whole-transform equality with the oracle at every size):
//   * limbs 0..7 of every operand of fr30_mul are <= 2^30 + 3, limb 8 < 2^29; twiddles are exact (limbs < 2^30, < 2r).
//     Column sum <= 9 * (2^30 + 3) * 2^30 + (2^30 - 1) * sum_j r_j + carry, with sum_j r_j = 4.91 * 2^30:
//     < 13.92 * 2^60 < 2^64.
//   * values: a product is < r + a * b / 2^270 < 2r for a < 2^269, b < 2r.  Inside a pass of k <= 10 stages the
//     un-multiplied outputs grow: sums double (< 2^k * 2r), differences carry the bias 2^12 r; the largest value that
//     can reach the last stage is < 2.2 * 2^12 r < 2^268.2, so nothing exceeds 270 bits and every subtrahend is
//     < 2^12 r (the bias keeps each limb, and therefore the number, non-negative).
#pragma once
#include "ff.hpp"

namespace ty {

struct Fr30 {
    uint32_t v[9];
};
constexpr uint32_t FR30_MASK = 0x3fffffffu;
// the most butterfly stages one pass may run on this form: the value bounds below (and fr30_reduce_lazy's x_8 < 2^29) are
// derived for k <= 10; ntt_run (ntt_host.hip) sends a plan with a longer pass to the 8 x 32-bit kernel
constexpr uint32_t FR30_MAX_STAGES = 10;

TY_HD constexpr uint32_t fr30_r(int i) {
    constexpr uint32_t t[9] = {0x1u, 0x3ffffffcu, 0x3fe5bfefu, 0x2f6900bfu, 0x21d80553u, 0x27602026u, 0x17d48333u, 0x29d4ca67u, 0x73edu};
    return t[i];
}
// 2^270 mod r: multiplying by it is a plain reduction (v -> v mod r up to one subtraction)
TY_HD constexpr uint32_t fr30_one(int i) {
    constexpr uint32_t t[9] = {0x3fff72acu, 0x2354fu, 0x3de5d540u, 0x1c220139u, 0x2a2f2112u, 0x22c03acbu, 0x22018550u, 0x12b2a694u, 0x10dcu};
    return t[i];
}
// 2^12 r in "spread" form: the limbs n_i of 2^12 r with 2^31 lent downwards along the chain (s_0 = n_0 + 2^31,
// s_i = n_i + 2^31 - 2, s_8 = n_8 - 2; same number), so that x_i + s_i - y_i never goes negative for y_i <= 2^30 + 3
// and never wraps (max 0xffffc001 for x_i <= 2^30 + 3)
TY_HD constexpr uint32_t fr30_bias(int i) {
    constexpr uint32_t n[9] = {0x1000u, 0x3fffc000u, 0x1bfeffffu, 0x100bfff9u, 0x553bdau, 0x2026876u, 0x83339d8u, 0xca675f5u, 0x73eda75u};
    return i == 0 ? n[0] + 0x80000000u : (i < 8 ? n[i] + 0x7ffffffeu : n[8] - 2u);
}

#if defined(__HIPCC__)
// a * b / 2^270 mod r; exact limbs out (< 2^30), value < r + a * b / 2^270
__device__ __forceinline__ Fr30 fr30_mul(const Fr30& a, const Fr30& b) {
    uint32_t m[9];
    Fr30 o;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i < (k < 9 ? k : 9); ++i) acc += (uint64_t)m[i] * fr30_r(k - i);
        if (k < 17) {
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        }
        if (k < 9) {
            m[k] = (0u - (uint32_t)acc) & FR30_MASK;  // -r^-1 = -1 mod 2^30
            acc += m[k];                              // m_k * r_0, r_0 = 1: the low limb cancels
        } else {
            o.v[k - 9] = (uint32_t)acc & FR30_MASK;
        }
        acc >>= 30;
    }
    return o;
}
// 8 x 32 words (any value < 2^256) -> 9 exact limbs
__device__ __forceinline__ Fr30 fr30_unpack(const Fr& x) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t t = x.v[wi] >> sh;
        if (sh > 2 && wi + 1 < 8) t |= x.v[wi + 1] << (32 - sh);
        r.v[i] = t & FR30_MASK;
    }
    return r;
}
// 9 exact limbs of a value < 2^256 -> 8 words
__device__ __forceinline__ Fr fr30_pack(const Fr30& a) {
    Fr o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, li = bit / 30, off = bit % 30;
        uint32_t x = a.v[li] >> off;
        if (li + 1 < 9) x |= a.v[li + 1] << (30 - off);
        if (off > 28 && li + 2 < 9) x |= a.v[li + 2] << (60 - off);
        o.v[j] = x;
    }
    return o;
}
// one parallel carry step: limbs 0..7 <= 2^30 + 3 afterwards for inputs below 2^32, limb 8 takes what is left
__device__ __forceinline__ Fr30 fr30_norm(const Fr30& a) {
    Fr30 r;
    r.v[0] = a.v[0] & FR30_MASK;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.v[i] = (a.v[i] & FR30_MASK) + (a.v[i - 1] >> 30);
    r.v[8] = a.v[8] + (a.v[7] >> 30);
    return r;
}
// x + y, carry step included
__device__ __forceinline__ Fr30 fr30_add(const Fr30& a, const Fr30& b) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
    return fr30_norm(r);
}
// x - y + 2^12 r (y < 2^12 r), carry step included
__device__ __forceinline__ Fr30 fr30_sub(const Fr30& a, const Fr30& b) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + fr30_bias(i) - b.v[i];
    return fr30_norm(r);
}
// exact-limbed value < 2r -> the canonical residue in arkworks' words
__device__ __forceinline__ Fr fr30_to_canonical(const Fr30& a) {
    Fr o = fr30_pack(a);
    fe_reduce_once(o);
    return o;
}
// x - q r for a small q (< 2^15: fr30_reduce_lazy passes x_8 / 0x73ee <= 18089), limb-wise with a signed running carry;
// exact limbs out.  Needs x >= q r.
__device__ __forceinline__ Fr30 fr30_sub_qr(const Fr30& x, uint32_t q) {
    Fr30 o;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        acc += (int64_t)x.v[i] - (int64_t)((uint64_t)q * fr30_r(i));
        o.v[i] = (uint32_t)acc & FR30_MASK;
        acc >>= 30;   // arithmetic: the borrow travels as a negative carry
    }
    return o;
}
// Lazy value -- limbs 0..7 <= 2^30 + 3, limb 8 < 2^29, the contract of every fr30_mul operand -- -> exact limbs,
// value < 2 r, WITHOUT a multiplication: one quotient estimate from the top limb.  r = 0x73ed * 2^240 + (< 2^240), so
// q = floor(x_8 / (0x73ed + 1)) satisfies q r < x_8 2^240 <= x (the subtraction cannot go negative), and the true
// quotient Q = floor(x / r) < (x_8 + 1 + 2^-27) / 0x73ed exceeds it by less than x_8 / (0x73ed * 0x73ee) + 1 + 2^-14
// < 2^29 / 2^29.71 + 1.0001 < 1.62, i.e. by at most 1: x - q r < 2 r.  ~120 cycles of VALU issue against ~860 for
// fr30_mul by 2^270 mod r (ntt_pass30_kernel: the last pass of a forward transform has no factor to fold the reduction into).
__device__ __forceinline__ Fr30 fr30_reduce_lazy(const Fr30& x) {
    constexpr uint32_t R8P1 = 0x73edu + 1u;
    return fr30_sub_qr(x, x.v[8] / R8P1);
}
__device__ __forceinline__ Fr30 fr30_const_one() {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = fr30_one(i);
    return r;
}
#endif

}  // namespace ty
