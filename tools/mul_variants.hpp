// Experimental Fq multiplication variants for tools/ubench.hip (timing only).
#pragma once
#include "../typlonk_amd/csrc/fq30.hpp"
namespace ty {

// acc(96) += a*b : one mad + one carry-capture add
#undef TY_MAC_VV
#undef TY_MAC_VS
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

// V4: 13x30 product phase with two interleaved accumulator chains (even / odd columns carry-free,
// merged afterwards) to expose ILP to a single wave
__device__ __forceinline__ Fq30 fq30_mul_ilp(const Fq30& a, const Fq30& b) {
    uint64_t col[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k : 12); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        col[k] = acc;
    }
    uint32_t T[26];
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        c += col[k];      // < 2^64: col < 13*2^60, carry < 2^35
        T[k] = (uint32_t)c & FQ30_MASK;
        c >>= 30;
    }
    T[25] = (uint32_t)c;
    return fq30_redc(T);
}

// V5: the reduction adds the product digit T[k] with a mad (T[k] * 1 + acc) instead of zero-extending it and a
// 64-bit add (v_mov + v_lshl_add_u64)
__device__ __forceinline__ Fq30 fq30_redc_v5(const uint32_t (&T)[26]) {
    uint32_t m[13];
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(T[k]) : "vcc");
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * fq30_kp(1, k - i);
        m[k] = ((uint32_t)acc * FQ30_NINV) & FQ30_MASK;
        acc += (uint64_t)m[k] * fq30_kp(1, 0);
        acc >>= 30;
    }
#pragma unroll
    for (int k = 13; k < 26; ++k) {
        asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(T[k]) : "vcc");
#pragma unroll
        for (int i = k - 12; i < 13; ++i) acc += (uint64_t)m[i] * fq30_kp(1, k - i);
        r.v[k - 13] = (uint32_t)acc & FQ30_MASK;
        acc >>= 30;
    }
    return r;
}
__device__ __forceinline__ Fq30 fq30_mul_v5(const Fq30& a, const Fq30& b) {
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
    return fq30_redc_v5(T);
}

// V6: V5's reduction after the independent-column product phase of V4
__device__ __forceinline__ Fq30 fq30_mul_v6(const Fq30& a, const Fq30& b) {
    uint64_t col[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k > 12 ? k - 12 : 0); i <= (k < 12 ? k : 12); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        col[k] = acc;
    }
    uint32_t T[26];
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        c += col[k];
        T[k] = (uint32_t)c & FQ30_MASK;
        c >>= 30;
    }
    T[25] = (uint32_t)c;
    return fq30_redc_v5(T);
}
// V7: V5 with the column carry folded into the first mad of the next column's m*p chain by keeping the chain in
// one asm-visible accumulator (no separate partial chain to merge)
__device__ __forceinline__ Fq30 fq30_redc_v7(const uint32_t (&T)[26]) {
    uint32_t m[13];
    Fq30 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(T[k]) : "vcc");
#pragma unroll
        for (int i = 0; i < k; ++i) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[i]), "s"(fq30_kp(1, k - i)) : "vcc");
        m[k] = ((uint32_t)acc * FQ30_NINV) & FQ30_MASK;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[k]), "s"(fq30_kp(1, 0)) : "vcc");
        acc >>= 30;
    }
#pragma unroll
    for (int k = 13; k < 26; ++k) {
        asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(T[k]) : "vcc");
#pragma unroll
        for (int i = k - 12; i < 13; ++i) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m[i]), "s"(fq30_kp(1, k - i)) : "vcc");
        r.v[k - 13] = (uint32_t)acc & FQ30_MASK;
        acc >>= 30;
    }
    return r;
}
__device__ __forceinline__ Fq30 fq30_mul_v7(const Fq30& a, const Fq30& b) {
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
    return fq30_redc_v7(T);
}

}  // namespace ty
