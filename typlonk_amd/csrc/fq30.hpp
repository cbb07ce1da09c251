// BLS12-381 base field Fq on 13 unsaturated 30-bit limbs -- the arithmetic of the MSM hot path
// (replaces ark-ff 0.3.0 Fp384 as used by ark-ec's group law; reference call sites
// /root/reference/kzg/src/lib.rs:49-52).
//
// Why 30-bit limbs.  Measured on MI355X (profiles/r01_ubench_*.txt): v_mad_u64_u32
// (32x32+64 -> 64) issues at ~5.3 cycles per wave, an add-with-carry at ~4.4, and a saturated
// 12x32-bit CIOS multiplication needs two carry adds per mad, so the carries cost more than the
// multiplies (40 G mul/s).  With 30-bit limbs a column of 13 partial products is < 13 * 2^60 <
// 2^64: a whole column accumulates in one 64-bit register with NOTHING but mads, and carries are
// resolved once per column (73 G mul/s, 1.8x).
//
// Representation.  x = sum v[i] * 2^(30 i), every limb < 2^30 ("normalised").  Montgomery domain
// with R = 2^390.  Values are kept LAZILY reduced: R / p = 2^9.3 ~ 630, so a Montgomery product of
// inputs a < A*p, b < B*p is < (1 + A*B/630) * p -- inputs may be several p large and the output
// is still ~p.  Every function documents its bound contract in units of p; the group law in
// g1.hpp states the bound of every intermediate.  Zero tests are exact tests mod p on values
// known to be < 2p (value in {0, p}).
//
// Memory format.  In HBM a field element is packed into 12 x 32-bit words (384 bits; every stored
// value is < 8p < 2^384), so points stay 96 B and XYZZ buckets 192 B with 16-byte alignment;
// pack/unpack are a few dozen shifts next to a ~2200-cycle multiplication.
//
// The same header compiles for the host (final window combine, affine normalisation, folds).
#pragma once
#include <stdint.h>

#include "ff.hpp"

namespace ty {

struct Fq30 {
    uint32_t v[13];
};

constexpr uint32_t FQ30_MASK = 0x3fffffffu;
constexpr uint32_t FQ30_NINV = 0x3ffcfffdu;  // -p^-1 mod 2^30

// k*p for k = 1..8 as normalised 30-bit digits
TY_HD constexpr uint32_t fq30_kp(int k, int i) {
    constexpr uint32_t t[8][13] = {
        /* 1p */ {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x034a83dau, 0x112bf673u, 0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x001a0111u},
        /* 2p */ {0x3fff5556u, 0x0ff7ffffu, 0x2a7ffff7u, 0x15ffff58u, 0x21ec483du, 0x069507b5u, 0x2257ece6u, 0x25c279c2u, 0x19aec8eeu, 0x3db21a5du, 0x13496374u, 0x351cbff3u, 0x00340223u},
        /* 3p */ {0x3fff0001u, 0x37f3ffffu, 0x3fbffff2u, 0x00ffff04u, 0x12e26c5cu, 0x09df8b90u, 0x3383e359u, 0x38a3b6a3u, 0x06862d65u, 0x1c8b278cu, 0x3cee152fu, 0x2fab1fecu, 0x004e0335u},
        /* 4p */ {0x3ffeaaacu, 0x1fefffffu, 0x14ffffeeu, 0x2bfffeb1u, 0x03d8907au, 0x0d2a0f6bu, 0x04afd9ccu, 0x0b84f385u, 0x335d91ddu, 0x3b6434bau, 0x2692c6e9u, 0x2a397fe6u, 0x00680447u},
        /* 5p */ {0x3ffe5557u, 0x07ebffffu, 0x2a3fffeau, 0x16fffe5du, 0x34ceb499u, 0x10749345u, 0x15dbd03fu, 0x1e663066u, 0x2034f654u, 0x1a3d41e9u, 0x103778a4u, 0x24c7dfe0u, 0x00820559u},
        /* 6p */ {0x3ffe0002u, 0x2fe7ffffu, 0x3f7fffe5u, 0x01fffe09u, 0x25c4d8b8u, 0x13bf1720u, 0x2707c6b2u, 0x31476d47u, 0x0d0c5acbu, 0x39164f18u, 0x39dc2a5eu, 0x1f563fd9u, 0x009c066bu},
        /* 7p */ {0x3ffdaaadu, 0x17e3ffffu, 0x14bfffe1u, 0x2cfffdb6u, 0x16bafcd6u, 0x17099afbu, 0x3833bd25u, 0x0428aa28u, 0x39e3bf43u, 0x17ef5c46u, 0x2380dc19u, 0x19e49fd3u, 0x00b6077du},
        /* 8p */ {0x3ffd5558u, 0x3fdfffffu, 0x29ffffdcu, 0x17fffd62u, 0x07b120f5u, 0x1a541ed6u, 0x095fb398u, 0x1709e70au, 0x26bb23bau, 0x36c86975u, 0x0d258dd3u, 0x1472ffcdu, 0x00d0088fu}};
    return t[k - 1][i];
}
// R mod p = Montgomery one
TY_HD constexpr uint32_t fq30_one_limb(int i) {
    constexpr uint32_t t[13] = {0x00d1ff2eu, 0x19d80000u, 0x34800ac4u, 0x2e00cde6u, 0x02431c84u, 0x269f83a2u, 0x3dcf80ddu,
                                0x09b42da0u, 0x25eec26cu, 0x15d98f12u, 0x04b29f14u, 0x259fcfa0u, 0x00015de9u};
    return t[i];
}
// 2^396 mod p : montmul(y, C_IN) turns an arkworks residue y = x*2^384 into x*2^390
TY_HD constexpr uint32_t fq30_cin_limb(int i) {
    constexpr uint32_t t[13] = {0x3480cb7fu, 0x3e0c0000u, 0x2042b126u, 0x3f337aafu, 0x3de4b4d1u, 0x1e015cf1u, 0x005c540du,
                                0x3467b19au, 0x352a6da3u, 0x19d89d19u, 0x2fb9afe6u, 0x3848c817u, 0x0009772fu};
    return t[i];
}
// 2^384 mod p : montmul(x*2^390, C_OUT) = x*2^384, the arkworks residue
TY_HD constexpr uint32_t fq30_cout_limb(int i) {
    constexpr uint32_t t[13] = {0x0002fffdu, 0x18240000u, 0x00c00027u, 0x3d0002f1u, 0x0758baebu, 0x22615d4fu, 0x257455f4u,
                                0x1614dc14u, 0x2c6d77ceu, 0x2a5e895bu, 0x0935c071u, 0x30fea039u, 0x0015f65eu};
    return t[i];
}

// R^3 mod p: one Montgomery product with it turns (a R)^-1 into a^-1 R
TY_HD constexpr uint32_t fq30_r3_limb(int i) {
    constexpr uint32_t t[13] = {0x1347c98du, 0x2c2194ccu, 0x1b027ceau, 0x2b8cb5a5u, 0x1214bbdeu, 0x0b5efd04u, 0x39edd0bfu,
                                0x1b2b39a4u, 0x32b00b66u, 0x1d1b6db4u, 0x3b8f0b8fu, 0x12c853b7u, 0x0006d0afu};
    return t[i];
}

TY_HD Fq30 fq30_zero() {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; ++i) r.v[i] = 0;
    return r;
}
TY_HD Fq30 fq30_one() {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; ++i) r.v[i] = fq30_one_limb(i);
    return r;
}

// acc += x for a 32-bit x.  On the device this is one v_mad_u64_u32 (x * 1 + acc): the compiler would otherwise
// zero-extend x into a register pair and issue a 64-bit add (v_mov + v_lshl_add_u64), measurably slower
// (tools/ubench: 72.1 -> 74.6 G Fq-mul/s).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FQ30_PLAIN_ADD32)
#define FQ30_ACC_ADD32(acc, x) asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(x) : "vcc")
#else
#define FQ30_ACC_ADD32(acc, x) ((acc) += (x))
#endif

// ---- multiplication -----------------------------------------------------------------------------
// Montgomery reduction of a 26-digit product T (T[25] may exceed 30 bits): T * 2^-390 mod p,
// result < p + T / 2^390, normalised.
TY_HD Fq30 fq30_redc(const uint32_t (&T)[26]) {
    uint32_t m[13];
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        FQ30_ACC_ADD32(acc, T[k]);
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * fq30_kp(1, k - i);
        m[k] = ((uint32_t)acc * FQ30_NINV) & FQ30_MASK;
        acc += (uint64_t)m[k] * fq30_kp(1, 0);
        acc >>= 30;
    }
#pragma unroll
    for (int k = 13; k < 26; ++k) {
        FQ30_ACC_ADD32(acc, T[k]);
#pragma unroll
        for (int i = k - 12; i < 13; ++i) acc += (uint64_t)m[i] * fq30_kp(1, k - i);
        r.v[k - 13] = (uint32_t)acc & FQ30_MASK;
        acc >>= 30;
    }
    return r;
}

// a*b*2^-390 mod p.  Needs normalised limbs and a*b < 2^780; output < (1 + A*B/630) p for
// a < A p, b < B p.  338 + 26 mads, no carry instructions inside a column.
TY_HD Fq30 fq30_mul_split(const Fq30& a, const Fq30& b) {
    uint32_t T[26];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
#pragma unroll
        for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k : 12); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        T[k] = (uint32_t)acc & FQ30_MASK;
        acc >>= 30;
    }
    T[25] = (uint32_t)acc;
    return fq30_redc(T);
}

// a*a*2^-390 mod p: cross terms once with a doubled operand (91 mads in the product phase).
// Column bound: 6 * 2^61 + 2^60 < 2^64.
TY_HD Fq30 fq30_sqr_split(const Fq30& a) {
    uint32_t T[26];
    uint32_t d[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) d[i] = a.v[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        const int lo = (k > 12 ? k - 12 : 0);
#pragma unroll
        for (int i = lo; 2 * i < k; ++i) acc += (uint64_t)d[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
        T[k] = (uint32_t)acc & FQ30_MASK;
        acc >>= 30;
    }
    T[25] = (uint32_t)acc;
    return fq30_redc(T);
}


// ==== fused product + reduction ==========================================================================
//
// fq30_mul forms the 26-digit product first (25 column normalisations: shift, mask, carry add) and
// then reduces it (26 more, plus 26 "T[k] * 1 + acc" mads to bring the digits back in).  Here column k
// of the product and column k of the reduction share a 64-bit accumulator,
//     acc_k = carry + sum_i a_i b_(k-i) + sum_i m_i p_(k-i),
// so a multiplication needs 26 normalisations instead of 51 and no digit re-adds: 338 + 13 multiplier
// instructions instead of 338 + 13 + 26.
//
// Overflow.  The reduction part of a column is at most (2^30 - 1) * (sum of the limbs of p it meets)
// <= 5.76 * 2^60 and every product term is < 2^60, so columns 0..9 and 15..25 stay below 2^64 whatever
// the (normalised) inputs are.  Columns 10..14 can reach 1.24 * 2^64: there the reduction terms and the
// first ten product terms (which provably fit) are accumulated first and only the remaining 11 terms of a
// multiplication (8 of a squaring) capture the carry-out of the mad into a third word
// (v_mad_u64_u32 ..., vcc + v_addc_co_u32).  The schedule below was computed with exact bounds by
// tools/fq30_fused_bounds.py; tests/cpp and the GPU tests compare every variant with fq30_mul on
// random and extreme (all-ones digits) inputs.
//
// Same contract as fq30_mul / fq30_sqr: normalised limbs, a*b < 2^780, result < p + a*b / 2^390.


#if defined(__HIP_DEVICE_COMPILE__)
// (hi : acc) += a * b   with the carry out of the 64-bit accumulator counted in hi
#define FQ30_MAC_CC_VV(acc, hi, a, b) \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "v"(b) : "vcc")
#define FQ30_MAC_CC_VS(acc, hi, a, b) \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "s"(b) : "vcc")
#else
#define FQ30_MAC_CC_VV(acc, hi, a, b)                          \
    do {                                                        \
        const uint64_t _t = (uint64_t)(a) * (uint64_t)(b);      \
        (acc) += _t;                                            \
        (hi) += ((acc) < _t) ? 1u : 0u;                         \
    } while (0)
#define FQ30_MAC_CC_VS(acc, hi, a, b) FQ30_MAC_CC_VV(acc, hi, a, b)
#endif

// acc += a * b.  -DFQ30_ASM_CHAIN (experiment, tools/ubench2.hip) pins every mad of a column into ONE dependent
// chain that starts from the shifted carry, so the compiler cannot give the column a fresh accumulator and merge the
// carry with a separate 64-bit add.
#if defined(__HIP_DEVICE_COMPILE__) && defined(FQ30_ASM_CHAIN)
#define FQ30_MAD_VV(acc, a, b) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc")
#define FQ30_MAD_VS(acc, a, b) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc")
#else
#define FQ30_MAD_VV(acc, a, b) ((acc) += (uint64_t)(a) * (b))
#define FQ30_MAD_VS(acc, a, b) ((acc) += (uint64_t)(a) * (b))
#endif

// capture schedule (tools/fq30_fused_bounds.py): in columns 11..14 the product terms from index `FIRST` on,
// and the closing m_k * p_0 of columns 10..12
TY_HD constexpr bool fq30_fused_wide_col(int k) { return k >= 10 && k <= 14; }
TY_HD constexpr bool fq30_fused_cap_last(int k) { return k >= 10 && k <= 12; }

template <bool SQR>
TY_HD Fq30 fq30_mulsqr_fused(const Fq30& a, const Fq30& b) {
    constexpr int FIRST = SQR ? 5 : 10;  // index (in program order) of the first captured product term of a wide column
    uint32_t m[13];
    uint32_t d[13];
    if (SQR) {
#pragma unroll
        for (int i = 0; i < 13; ++i) d[i] = a.v[i] << 1;
    }
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 26; ++k) {
        uint32_t hi = 0;
        const bool wide = fq30_fused_wide_col(k);
        // reduction terms of this column: m_i p_(k-i), i < k (first half) / i >= k - 12 (second half)
#pragma unroll
        for (int i = (k > 12 ? k - 12 : 0); i < (k < 13 ? k : 13); ++i) FQ30_MAD_VS(acc, m[i], fq30_kp(1, k - i));
        // product terms
        if (k < 25) {
            int idx = 0;
            const int lo = (k > 12 ? k - 12 : 0), hi_i = (k < 12 ? k : 12);
            if (SQR) {
                if ((k & 1) == 0) {
                    FQ30_MAD_VV(acc, a.v[k / 2], a.v[k / 2]);
                    ++idx;
                }
#pragma unroll
                for (int i = lo; 2 * i < k; ++i) {
                    if (wide && k >= 11 && idx >= FIRST) {
                        FQ30_MAC_CC_VV(acc, hi, d[i], a.v[k - i]);
                    } else {
                        FQ30_MAD_VV(acc, d[i], a.v[k - i]);
                    }
                    ++idx;
                }
            } else {
#pragma unroll
                for (int i = lo; i <= hi_i; ++i) {
                    if (wide && k >= 11 && idx >= FIRST) {
                        FQ30_MAC_CC_VV(acc, hi, a.v[i], b.v[k - i]);
                    } else {
                        FQ30_MAD_VV(acc, a.v[i], b.v[k - i]);
                    }
                    ++idx;
                }
            }
        }
        if (k < 13) {
            m[k] = ((uint32_t)acc * FQ30_NINV) & FQ30_MASK;
            if (fq30_fused_cap_last(k)) {
                FQ30_MAC_CC_VS(acc, hi, m[k], fq30_kp(1, 0));
            } else {
                FQ30_MAD_VS(acc, m[k], fq30_kp(1, 0));
            }
        } else {
            r.v[k - 13] = (uint32_t)acc & FQ30_MASK;
        }
        acc >>= 30;
        if (wide) acc |= (uint64_t)hi << 34;
    }
    return r;
}

TY_HD Fq30 fq30_mul_fused(const Fq30& a, const Fq30& b) { return fq30_mulsqr_fused<false>(a, b); }
TY_HD Fq30 fq30_sqr_fused(const Fq30& a) { return fq30_mulsqr_fused<true>(a, a); }

// ---- un-reduced products: a*b + c*d with ONE reduction -------------------------------------------------
// 26 normalised digits of a*b (T[25] may exceed 30 bits)
TY_HD void fq30_mul_wide(const Fq30& a, const Fq30& b, uint32_t (&T)[26]) {
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
#pragma unroll
        for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k : 12); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        T[k] = (uint32_t)acc & FQ30_MASK;
        acc >>= 30;
    }
    T[25] = (uint32_t)acc;
}
// (a*b + c*d) * 2^-390 mod p; needs a*b + c*d < 2^780; result < p + (a*b + c*d) / 2^390.
// The digit sums are < 2^31 (T[25]: < 2^32 by the value bound), which fq30_redc takes as they are.
TY_HD Fq30 fq30_mul2_add(const Fq30& a, const Fq30& b, const Fq30& c, const Fq30& d) {
    uint32_t T[26], U[26];
    fq30_mul_wide(a, b, T);
    fq30_mul_wide(c, d, U);
#pragma unroll
    for (int k = 0; k < 26; ++k) T[k] += U[k];
    return fq30_redc(T);
}

// ---- the multiplication the library uses ---------------------------------------------------------------
// -DFQ30_SPLIT_MUL selects the two-phase form (product digits, then reduction) for A/B measurements
// (tools/ubench2.hip); results are identical digit for digit.
#if defined(FQ30_SPLIT_MUL)
TY_HD Fq30 fq30_mul(const Fq30& a, const Fq30& b) { return fq30_mul_split(a, b); }
TY_HD Fq30 fq30_sqr(const Fq30& a) { return fq30_sqr_split(a); }
#else
TY_HD Fq30 fq30_mul(const Fq30& a, const Fq30& b) { return fq30_mul_fused(a, b); }
TY_HD Fq30 fq30_sqr(const Fq30& a) { return fq30_sqr_fused(a); }
#endif

// ---- lazy additive operations (no modular reduction; callers track bounds) -----------------------
// a + b, normalised.  Value a + b must be < 2^390.
TY_HD Fq30 fq30_add_lazy(const Fq30& a, const Fq30& b) {
    Fq30 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const uint32_t s = a.v[i] + b.v[i] + c;
        r.v[i] = s & FQ30_MASK;
        c = s >> 30;
    }
    return r;
}
// k*a for k = 2 or 3, normalised.
template <int K>
TY_HD Fq30 fq30_mulk_lazy(const Fq30& a) {
    Fq30 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const uint32_t s = a.v[i] * (uint32_t)K + c;  // < 3 * 2^30 + 3 < 2^32
        r.v[i] = s & FQ30_MASK;
        c = s >> 30;
    }
    return r;
}
// a - b + K p, normalised.  Requires b <= K p (so the value is >= 0) and a + K p < 2^390.
template <int K>
TY_HD Fq30 fq30_sub_lazy(const Fq30& a, const Fq30& b) {
    Fq30 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int32_t s = (int32_t)(a.v[i] + fq30_kp(K, i)) + c - (int32_t)b.v[i];
        r.v[i] = (uint32_t)s & FQ30_MASK;
        c = s >> 30;
    }
    return r;
}
// a - b - c + K p, normalised.  Requires b + c <= K p.
template <int K>
TY_HD Fq30 fq30_sub2_lazy(const Fq30& a, const Fq30& b, const Fq30& cc) {
    Fq30 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int32_t s = (int32_t)(a.v[i] + fq30_kp(K, i)) + c - (int32_t)b.v[i] - (int32_t)cc.v[i];
        r.v[i] = (uint32_t)s & FQ30_MASK;
        c = s >> 30;
    }
    return r;
}
// K p - a  (a <= K p)
template <int K>
TY_HD Fq30 fq30_neg_lazy(const Fq30& a) {
    return fq30_sub_lazy<K>(fq30_zero(), a);
}

// ---- exact predicates / canonical form ------------------------------------------------------------
// a == 0 mod p for a normalised a < 2p
TY_HD bool fq30_is_zero_mod(const Fq30& a) {
    uint32_t z = 0, q = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        z |= a.v[i];
        q |= a.v[i] ^ fq30_kp(1, i);
    }
    return z == 0 || q == 0;
}
TY_HD bool fq30_is_zero_exact(const Fq30& a) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) z |= a.v[i];
    return z == 0;
}
// r = a - K p if that is >= 0 else a
template <int K>
TY_HD void fq30_cond_sub(Fq30& a) {
    uint32_t d[13];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int32_t s = (int32_t)a.v[i] + c - (int32_t)fq30_kp(K, i);
        d[i] = (uint32_t)s & FQ30_MASK;
        c = s >> 30;
    }
    if (c >= 0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) a.v[i] = d[i];
    }
}
// canonical representative in [0, p) of a normalised a < 8p
TY_HD Fq30 fq30_canon(const Fq30& a) {
    Fq30 r = a;
    fq30_cond_sub<4>(r);
    fq30_cond_sub<2>(r);
    fq30_cond_sub<1>(r);
    return r;
}

// ---- packed 12 x 32-bit memory form ------------------------------------------------------------------
TY_HD Fq30 fq30_unpack(const uint32_t (&w)[12]) {
    Fq30 r;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t x = w[wi] >> sh;
        if (sh > 2 && wi + 1 < 12) x |= w[wi + 1] << (32 - sh);
        r.v[i] = x & FQ30_MASK;
    }
    return r;
}
// value must be < 2^384
TY_HD void fq30_pack(const Fq30& a, uint32_t (&w)[12]) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int bit = 32 * j, li = bit / 30, off = bit % 30;
        uint32_t x = a.v[li] >> off;
        if (li + 1 < 13) x |= a.v[li + 1] << (30 - off);
        if (off > 28 && li + 2 < 13) x |= a.v[li + 2] << (60 - off);
        w[j] = x;
    }
}

// arkworks residue (12 words, x*2^384 mod p, canonical) -> internal x*2^390, canonical
TY_HD Fq30 fq30_from_ark(const uint32_t (&w)[12]) {
    Fq30 c;
#pragma unroll
    for (int i = 0; i < 13; ++i) c.v[i] = fq30_cin_limb(i);
    return fq30_canon(fq30_mul(fq30_unpack(w), c));
}
// internal (< 8p) -> arkworks residue words, canonical
TY_HD void fq30_to_ark(const Fq30& a, uint32_t (&w)[12]) {
    Fq30 c;
#pragma unroll
    for (int i = 0; i < 13; ++i) c.v[i] = fq30_cout_limb(i);
    fq30_pack(fq30_canon(fq30_mul(a, c)), w);
}

// (plain host functions: the device pass parses them -- host code in the .hip units names them -- and emits nothing)
// Host-side inversion by the binary extended Euclid algorithm on six 64-bit words: ~10 us instead of the ~50 us of the
// Fermat ladder below.  It is the last step of every MSM (canonical affine output, g1_to_affine), on the host's critical
// path.  Data-dependent branches are no concern here: the inverted value is a projective denominator of a public result.
struct Fq30U384 {
    uint64_t w[6];
};
inline bool fq30_u384_geq(const Fq30U384& a, const Fq30U384& b) {
    for (int i = 5; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
inline void fq30_u384_sub(Fq30U384& a, const Fq30U384& b) {  // a -= b (a >= b)
    unsigned __int128 br = 0;
    for (int i = 0; i < 6; ++i) {
        const unsigned __int128 t = (unsigned __int128)a.w[i] - b.w[i] - (uint64_t)br;
        a.w[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
inline uint64_t fq30_u384_add(Fq30U384& a, const Fq30U384& b) {  // a += b, returns the carry out
    unsigned __int128 c = 0;
    for (int i = 0; i < 6; ++i) {
        c += (unsigned __int128)a.w[i] + b.w[i];
        a.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
inline void fq30_u384_shr1(Fq30U384& a, uint64_t top) {
    for (int i = 0; i < 5; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 63);
    a.w[5] = (a.w[5] >> 1) | (top << 63);
}
inline bool fq30_u384_is_one(const Fq30U384& a) { return a.w[0] == 1 && !(a.w[1] | a.w[2] | a.w[3] | a.w[4] | a.w[5]); }
// x / 2 mod p for x < p
inline void fq30_u384_half_mod(Fq30U384& x, const Fq30U384& p) {
    if (x.w[0] & 1) {
        const uint64_t c = fq30_u384_add(x, p);  // x + p is even and < 2^382: no carry, but keep the general form
        fq30_u384_shr1(x, c);
    } else {
        fq30_u384_shr1(x, 0);
    }
}
// u^-1 mod p for a plain integer 0 < u < p: binary extended Euclid on 64-bit words
inline Fq30U384 fq30_u384_modinv(Fq30U384 u) {
    static const Fq30U384 P = {{0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull,
                                0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};
    Fq30U384 v = P, x1 = {{1, 0, 0, 0, 0, 0}}, x2 = {{0, 0, 0, 0, 0, 0}};
    // invariant: x1 * residue = u, x2 * residue = v (mod p)
    while (!fq30_u384_is_one(u) && !fq30_u384_is_one(v)) {
        while (!(u.w[0] & 1)) {
            fq30_u384_shr1(u, 0);
            fq30_u384_half_mod(x1, P);
        }
        while (!(v.w[0] & 1)) {
            fq30_u384_shr1(v, 0);
            fq30_u384_half_mod(x2, P);
        }
        if (fq30_u384_geq(u, v)) {
            fq30_u384_sub(u, v);
            if (!fq30_u384_geq(x1, x2)) fq30_u384_add(x1, P);
            fq30_u384_sub(x1, x2);
        } else {
            fq30_u384_sub(v, u);
            if (!fq30_u384_geq(x2, x1)) fq30_u384_add(x2, P);
            fq30_u384_sub(x2, x1);
        }
    }
    return fq30_u384_is_one(u) ? x1 : x2;
}
inline Fq30 fq30_inv_gcd(const Fq30& a) {
    const Fq30 c = fq30_canon(a);
    if (fq30_is_zero_exact(c)) return fq30_zero();
    // the residue itself, as a plain integer < p
    Fq30U384 u = {{0, 0, 0, 0, 0, 0}};
    for (int i = 0; i < 13; ++i) {
        const int bit = 30 * i, wi = bit >> 6, sh = bit & 63;
        u.w[wi] |= (uint64_t)c.v[i] << sh;
        if (sh > 34 && wi + 1 < 6) u.w[wi + 1] |= (uint64_t)c.v[i] >> (64 - sh);
    }
    const Fq30U384 inv = fq30_u384_modinv(u);  // residue^-1 as a plain integer < p
    Fq30 r;
    for (int i = 0; i < 13; ++i) {
        const int bit = 30 * i, wi = bit >> 6, sh = bit & 63;
        uint64_t x = inv.w[wi] >> sh;
        if (sh > 34 && wi + 1 < 6) x |= inv.w[wi + 1] << (64 - sh);
        r.v[i] = (uint32_t)x & FQ30_MASK;
    }
    // (a R)^-1 -> a^-1 R: times R^2, i.e. one Montgomery product with R^3 mod p
    Fq30 k;
    for (int i = 0; i < 13; ++i) k.v[i] = fq30_r3_limb(i);   // (a R)^-1 -> a^-1 R
    return fq30_mul(r, k);
}

// ---- SIMT inversion: Bernstein-Yang divsteps ("safegcd") on the 13 x 30-bit limbs ----------------------------------
// The device had only the Fermat ladder below (a^(p-2): 380 squarings + ~190 multiplications = 551 multiplication times
// per wavefront, profiles/r04_ubench4_pricing.txt).  This is the branch-free divstep recurrence
//     divstep(delta, f, g) = (1 - delta, g, (g - f)/2)               if delta > 0 and g odd
//                            (1 + delta, f, (g + (g mod 2) f)/2)     otherwise
// started at (1, p, a): Theorem 11.2 of Bernstein-Yang ("Fast constant-time gcd computation and modular inversion",
// TCHES 2019) bounds the number of divsteps that reach g = 0 by floor((49 d + 57)/17) = 1101 for d = 381 bits, i.e.
// FQ30_DIVSTEP_ROUNDS = 37 rounds of 30.  A round runs 30 divsteps on the low 30 bits of f and g with 32-bit full-rate
// instructions only, collecting the transition matrix t = (u v; q r), 2^30 (f', g') = t (f, g), |u| + |v| <= 2^30,
// |q| + |r| <= 2^30; then applies t to the full-width (f, g) (exact division by 2^30) and to the Bezout pair (d, e),
// which is kept mod p in (-2p, p) with the multiple of p added that makes the division by 2^30 exact.  Invariant:
// d * a = f and e * a = g (mod p).  At g = 0: f = +-gcd = +-1 and a^-1 = +-d; a = 0 leaves d = 0 (0 -> 0).
// Integers are 13 signed limbs: limbs 0..11 in [0, 2^30), limb 12 carries the sign.
// A lane never leaves the loop alone: the exit test is wave-uniform (all lanes at g = 0; once there, further rounds
// change neither f nor d), so the usual count is the slowest lane's 26-28 rounds, the bound 37 (-DFQ30_INV_FIXED_ROUNDS:
// always 37, for a data-independent instruction stream).
constexpr int FQ30_DIVSTEP_ROUNDS = 37;
constexpr uint32_t FQ30_PINV = 0x00030003u;   // p^-1 mod 2^30

// 30 divsteps on the low words; eta = -delta.  Only bit 0 of g is ever inspected and step i sees input bits <= i, so the
// two spare bits of the 32-bit registers may hold anything.  t = {u, v, q, r}.
//
// The step is linear in the matrix rows -- negate, add, double -- so a row travels as ONE register, u + v 2^16 as a plain
// integer (and q + r 2^16): six instructions per step for the matrix instead of twelve.  A packed row is exact while
// |u|, |v| < 2^15, so the 30 steps run as three groups of ten (entries <= 2^10), and the group matrices are multiplied
// together (4 multiplications + 4 multiply-adds on 32-bit words per product; |u| + |v| <= 2^30 at the end).
// Round 5: 780 -> 640 instructions for the 30 steps.
TY_HD int32_t fq30_divsteps30(int32_t eta, uint32_t f, uint32_t g, int32_t (&t)[4]) {
    int32_t mu = 1, mv = 0, mq = 0, mr = 1;               // the matrix of the groups done so far
#pragma unroll 1
    for (int grp = 0; grp < 3; ++grp) {
        uint32_t a = 1u, b = 1u << 16;                    // rows (u, v) = (1, 0) and (q, r) = (0, 1)
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            uint32_t c1 = (uint32_t)(eta >> 31);          // delta > 0
            const uint32_t c2 = 0u - (g & 1u);            // g odd
            // g <- g +- f (minus when delta > 0), likewise the second matrix row
            g += ((f ^ c1) - c1) & c2;
            b += ((a ^ c1) - c1) & c2;
            c1 &= c2;                                     // swap: delta > 0 and g odd
            eta = (int32_t)(((uint32_t)eta ^ c1) + ~c1);  // swap: -eta - 1 = ~eta; else eta - 1
            f += g & c1;                                  // swap: f <- old g
            a += b & c1;
            g >>= 1;
            a += a;                                       // the f-row is scaled instead of halving the g-row
        }
        // unpack: low half sign-extended, the rest is the high half exactly
        const int32_t u = (int32_t)(a << 16) >> 16, v = (int32_t)(a - (uint32_t)u) >> 16;
        const int32_t q = (int32_t)(b << 16) >> 16, r = (int32_t)(b - (uint32_t)q) >> 16;
        // (this group) x (the groups before it)
        const int32_t nu = u * mu + v * mq, nv = u * mv + v * mr, nq = q * mu + r * mq, nr = q * mv + r * mr;
        mu = nu;
        mv = nv;
        mq = nq;
        mr = nr;
    }
    t[0] = mu;
    t[1] = mv;
    t[2] = mq;
    t[3] = mr;
    return eta;
}
// acc += a * b on signed 32-bit factors: ONE v_mad_i64_i32 on the device.  The limbs below the top one are known to be
// non-negative (they were just masked), the compiler therefore zero-extends them, and a product of a sign-extended and a
// zero-extended factor has no single instruction: it emitted three multiplier instructions and two moves per product.
// Passing the limb through an empty asm hides the known bits and costs nothing (1155 -> 960 instructions per round).
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ30_OPAQUE(x) asm("" : "+v"(x))
#else
#define FQ30_OPAQUE(x) ((void)0)
#endif
#define FQ30_SMAD_VV(acc, a, b) do { int32_t _b = (b); FQ30_OPAQUE(_b); (acc) += (int64_t)(a) * (int64_t)_b; } while (0)
#define FQ30_SMAD_VS(acc, a, b) ((acc) += (int64_t)(a) * (int64_t)(b))
// the first product of a column takes the shifted carry as its addend; pinned, or the compiler starts the column from zero
// and merges the carry with a separate 64-bit addition (48 per round)
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ30_SMAD_FIRST(acc, a, b) do { uint64_t _co; asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(_co) : "v"(a), "v"(b)); } while (0)
#else
#define FQ30_SMAD_FIRST(acc, a, b) FQ30_SMAD_VV(acc, a, b)
#endif

// (f, g) <- t (f, g) / 2^30, exact
TY_HD void fq30_divsteps_update_fg(int32_t (&f)[13], int32_t (&g)[13], const int32_t (&t)[4]) {
    int64_t cf = 0, cg = 0;
    FQ30_SMAD_VV(cf, t[0], f[0]);
    FQ30_SMAD_VV(cf, t[1], g[0]);
    FQ30_SMAD_VV(cg, t[2], f[0]);
    FQ30_SMAD_VV(cg, t[3], g[0]);
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 13; ++i) {
        FQ30_SMAD_FIRST(cf, t[0], f[i]);
        FQ30_SMAD_FIRST(cg, t[2], f[i]);
        FQ30_SMAD_VV(cf, t[1], g[i]);
        FQ30_SMAD_VV(cg, t[3], g[i]);
        f[i - 1] = (int32_t)((uint32_t)cf & FQ30_MASK);
        g[i - 1] = (int32_t)((uint32_t)cg & FQ30_MASK);
        cf >>= 30;
        cg >>= 30;
    }
    f[12] = (int32_t)cf;
    g[12] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod p, both kept in (-2p, p): with d + [d<0] p and e + [e<0] p in (-p, p) and
// |u| + |v| <= 2^30 the combination is in (-2^30 p, 2^30 p); subtracting k p, 0 <= k < 2^30 chosen so that the low 30
// bits vanish, leaves (-2^31 p, 2^30 p) before the exact shift.
TY_HD void fq30_divsteps_update_de(int32_t (&d)[13], int32_t (&e)[13], const int32_t (&t)[4]) {
    const int32_t sd = d[12] >> 31, se = e[12] >> 31;
    int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
    int64_t cd = 0, ce = 0;
    FQ30_SMAD_VV(cd, t[0], d[0]);
    FQ30_SMAD_VV(cd, t[1], e[0]);
    FQ30_SMAD_VV(ce, t[2], d[0]);
    FQ30_SMAD_VV(ce, t[3], e[0]);
    md -= (int32_t)((FQ30_PINV * (uint32_t)cd + (uint32_t)md) & FQ30_MASK);
    me -= (int32_t)((FQ30_PINV * (uint32_t)ce + (uint32_t)me) & FQ30_MASK);
    FQ30_SMAD_VS(cd, md, (int32_t)fq30_kp(1, 0));
    FQ30_SMAD_VS(ce, me, (int32_t)fq30_kp(1, 0));
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 13; ++i) {
        FQ30_SMAD_FIRST(cd, t[0], d[i]);
        FQ30_SMAD_FIRST(ce, t[2], d[i]);
        FQ30_SMAD_VV(cd, t[1], e[i]);
        FQ30_SMAD_VV(ce, t[3], e[i]);
        FQ30_SMAD_VS(cd, md, (int32_t)fq30_kp(1, i));
        FQ30_SMAD_VS(ce, me, (int32_t)fq30_kp(1, i));
        d[i - 1] = (int32_t)((uint32_t)cd & FQ30_MASK);
        e[i - 1] = (int32_t)((uint32_t)ce & FQ30_MASK);
        cd >>= 30;
        ce >>= 30;
    }
    d[12] = (int32_t)cd;
    e[12] = (int32_t)ce;
}

// a^-1 by divsteps; input normalised and < 8p, output < 1.01 p (as fq30_inv_fermat).  0 -> 0.
// `rounds_out` (host tests): the number of rounds this call ran.
TY_HD Fq30 fq30_inv_divsteps(const Fq30& a, int* rounds_out = nullptr) {
    const Fq30 c = fq30_canon(a);
    int32_t f[13], g[13], d[13], e[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        f[i] = (int32_t)fq30_kp(1, i);
        g[i] = (int32_t)c.v[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;
    int rounds = 0;
#pragma unroll 1
    for (; rounds < FQ30_DIVSTEP_ROUNDS; ++rounds) {
#if !defined(FQ30_INV_FIXED_ROUNDS)
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 13; ++i) nz |= (uint32_t)g[i];
#if defined(__HIP_DEVICE_COMPILE__)
        if (!__any(nz != 0)) break;
#else
        if (nz == 0) break;
#endif
#endif
        int32_t t[4];
        eta = fq30_divsteps30(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        fq30_divsteps_update_de(d, e, t);
        fq30_divsteps_update_fg(f, g, t);
    }
    if (rounds_out) *rounds_out = rounds;
    // +-d + 2p is in (0, 4p): sign of f decides, no conditional correction needed before the Montgomery product
    const int32_t sf = f[12] >> 31;
    Fq30 x;
    int32_t cy = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int32_t s = ((d[i] ^ sf) - sf) + (int32_t)fq30_kp(2, i) + cy;
        x.v[i] = (uint32_t)s & FQ30_MASK;
        cy = s >> 30;
    }
    Fq30 k;
#pragma unroll
    for (int i = 0; i < 13; ++i) k.v[i] = fq30_r3_limb(i);
    return fq30_mul(x, k);
}

// Fermat ladder a^(p-2): the round-1..4 device form, kept as the reference the two other inversions are tested against
TY_HD Fq30 fq30_inv_fermat(const Fq30& a);

// a^-1; input < 8p, output < 1.01 p.  0 -> 0.  Host: binary Euclid on 64-bit words; device: divsteps.
TY_HD Fq30 fq30_inv(const Fq30& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return fq30_inv_gcd(a);
#elif defined(FQ30_INV_FERMAT)
    return fq30_inv_fermat(a);
#else
    return fq30_inv_divsteps(a);
#endif
}

// a^-1 (Fermat, a^(p-2)); input < 2p, output < 1.01 p.  0 -> 0.
TY_HD Fq30 fq30_inv_fermat(const Fq30& a) {
    // exponent p - 2 as 32-bit words (little endian)
    constexpr uint32_t e[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    Fq30 acc = fq30_one();
    bool started = false;
    for (int w = 11; w >= 0; --w) {
        for (int b = 31; b >= 0; --b) {
            if (started) acc = fq30_sqr(acc);
            if ((e[w] >> b) & 1) {
                acc = started ? fq30_mul(acc, a) : a;
                started = true;
            }
        }
    }
    return acc;
}

}  // namespace ty
