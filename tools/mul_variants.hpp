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

}  // namespace ty
