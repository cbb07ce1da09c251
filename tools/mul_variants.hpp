// Experimental Fq multiplication variants for tools/ubench.hip (timing only).
#pragma once
#include "../typlonk_amd/csrc/ff.hpp"
namespace ty {

// acc(96) += a*b : one mad + one carry-capture add
#define TY_MAC_VV(lo, hi, a, b) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc")
#define TY_MAC_VS(lo, hi, a, b) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "s"(b) : "vcc")

// V2: product scanning (FIPS), 12x32 saturated limbs, 96-bit column accumulator
template <class P>
__device__ __forceinline__ Fe<P> fe_mul_ps(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    uint32_t m[N], r[N];
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) TY_MAC_VV(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) TY_MAC_VS(lo, hi, m[i], P::mod(k - i));
        m[k] = (uint32_t)lo * P::INV;
        TY_MAC_VS(lo, hi, m[k], P::mod(0));
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            TY_MAC_VV(lo, hi, a.v[i], b.v[k - i]);
            TY_MAC_VS(lo, hi, m[i], P::mod(k - i));
        }
        r[k - N] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    r[N - 1] = (uint32_t)lo;
    Fe<P> o;
#pragma unroll
    for (int i = 0; i < N; ++i) o.v[i] = r[i];
    fe_reduce_once(o);
    return o;
}

// V3: unsaturated 13 x 30-bit limbs, R = 2^390; column sums of <= 13 products fit 64 bits, so the
// mads need no carry handling at all.  Separate product and reduction phases.
struct Fq30 {
    uint32_t v[13];
};
__device__ __forceinline__ constexpr uint32_t p30(int i) {
    constexpr uint32_t m[13] = {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x34a83dau, 0x112bf673u,
                                0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x1a0111u};
    return m[i];
}
constexpr uint32_t NINV30 = 0x3ffcfffdu;
constexpr uint32_t MASK30 = 0x3fffffffu;

__device__ __forceinline__ Fq30 fq30_mul(const Fq30& a, const Fq30& b) {
    uint32_t T[26];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
#pragma unroll
        for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k : 12); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        T[k] = (uint32_t)acc & MASK30;
        acc >>= 30;
    }
    T[25] = (uint32_t)acc;
    uint32_t m[13];
    Fq30 r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        acc += T[k];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p30(k - i);
        m[k] = ((uint32_t)acc * NINV30) & MASK30;
        acc += (uint64_t)m[k] * p30(0);
        acc >>= 30;
    }
#pragma unroll
    for (int k = 13; k < 26; ++k) {
        acc += T[k];
#pragma unroll
        for (int i = k - 12; i < 13; ++i) acc += (uint64_t)m[i] * p30(k - i);
        r.v[k - 13] = (uint32_t)acc & MASK30;
        acc >>= 30;
    }
    return r;
}

}  // namespace ty
