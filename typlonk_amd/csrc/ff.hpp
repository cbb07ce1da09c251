// Fp (381-bit) / Fr (255-bit) Montgomery arithmetic for BLS12-381 on 32-bit limbs.
//
// Replaces ark-ff 0.3.0 Fp384 / Fp256 (un-vendored; /root/reference/Cargo.lock:42-43) on the
// MSM + NTT hot path.  In-memory form is the arkworks one -- little-endian limbs of the Montgomery
// residue (R = 2^384 / 2^256) -- so a [u64; 6] / [u64; 4] from Rust reinterprets as 12 / 8 u32
// limbs with no conversion on a little-endian host.
//
// gfx950 has no 64x64 multiplier; the widest integer multiply is v_mad_u64_u32 (32x32+64 -> 64),
// so the natural limb is 32 bits.  Everything is written as fully unrolled straight-line code on
// `uint32_t v[N]` so that limbs live in VGPRs and modulus limbs fold into SGPR constants.
//
// The same header compiles for the host (g++ or hipcc host pass): the host side of the library
// (twiddle tables, final affine normalisation, partial-sum folds) uses exactly this code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TY_HD __host__ __device__ __forceinline__
#else
#define TY_HD inline __attribute__((always_inline))
#endif

namespace ty {

struct FqParams {
    static constexpr int N = 12;
    // p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    static TY_HD constexpr uint32_t mod(int i) {
        constexpr uint32_t m[N] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                   0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return m[i];
    }
    // R = 2^384 mod p  (Montgomery one)
    static TY_HD constexpr uint32_t one(int i) {
        constexpr uint32_t m[N] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                   0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return m[i];
    }
    // R^2 mod p
    static TY_HD constexpr uint32_t r2(int i) {
        constexpr uint32_t m[N] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                   0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xfffcfffdu;  // -p^-1 mod 2^32
};

struct FrParams {
    static constexpr int N = 8;
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static TY_HD constexpr uint32_t mod(int i) {
        constexpr uint32_t m[N] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return m[i];
    }
    static TY_HD constexpr uint32_t one(int i) {
        constexpr uint32_t m[N] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return m[i];
    }
    static TY_HD constexpr uint32_t r2(int i) {
        constexpr uint32_t m[N] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
};

template <class P>
struct Fe {
    static constexpr int N = P::N;
    uint32_t v[N];

    static TY_HD Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    static TY_HD Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::one(i);
        return r;
    }
    static TY_HD Fe modulus() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::mod(i);
        return r;
    }
    static TY_HD Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::r2(i);
        return r;
    }
    TY_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= v[i];
        return o == 0;
    }
    TY_HD bool operator==(const Fe& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    TY_HD bool operator!=(const Fe& b) const { return !(*this == b); }
};

#if defined(__HIP_DEVICE_COMPILE__)
// ---- 8-limb (Fr) carry chains in inline assembly -------------------------------------------------------------------
// The portable forms below (64-bit temporaries) compile to a v_mov + v_lshl_add_u64 pair per limb on gfx950: an
// addition + conditional subtraction costs ~60 instructions, and the three of a radix-2 butterfly (add, sub, the
// multiplication's final reduction) were ~30 % of its time.  Written as v_add_co / v_addc_co chains they are 24-25
// instructions each.  (N = 12 keeps the portable form: Fq only runs at set-up time.)
#define TY_L8(x) "v"((x)[0]), "v"((x)[1]), "v"((x)[2]), "v"((x)[3]), "v"((x)[4]), "v"((x)[5]), "v"((x)[6]), "v"((x)[7])
#define TY_O8(x) "=&v"((x)[0]), "=&v"((x)[1]), "=&v"((x)[2]), "=&v"((x)[3]), "=&v"((x)[4]), "=&v"((x)[5]), "=&v"((x)[6]), "=&v"((x)[7])
// modulus limbs: in VGPRs for the carry chains (VOP2 with a carry-in already reads VCC, and gfx9 allows ONE constant-bus
// operand per instruction), in SGPRs where no carry is involved
#define TY_M8V(P) "v"(P::mod(0)), "v"(P::mod(1)), "v"(P::mod(2)), "v"(P::mod(3)), "v"(P::mod(4)), "v"(P::mod(5)), "v"(P::mod(6)), "v"(P::mod(7))
#define TY_M8(P) "s"(P::mod(0)), "s"(P::mod(1)), "s"(P::mod(2)), "s"(P::mod(3)), "s"(P::mod(4)), "s"(P::mod(5)), "s"(P::mod(6)), "s"(P::mod(7))
// r = a + b   (no carry out of limb 7: both operands < 2^255)
__device__ __forceinline__ void ty_add8(uint32_t (&r)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %15, %23, vcc"
        : TY_O8(r)
        : TY_L8(a), TY_L8(b)
        : "vcc");
}
// d = x - p if x >= p else x   (x < 2p)
template <class P>
__device__ __forceinline__ void ty_csub8(uint32_t (&d)[8], const uint32_t (&x)[8]) {
    asm("v_subrev_co_u32 %0, vcc, %16, %8\n\t"
        "v_subbrev_co_u32 %1, vcc, %17, %9, vcc\n\t"
        "v_subbrev_co_u32 %2, vcc, %18, %10, vcc\n\t"
        "v_subbrev_co_u32 %3, vcc, %19, %11, vcc\n\t"
        "v_subbrev_co_u32 %4, vcc, %20, %12, vcc\n\t"
        "v_subbrev_co_u32 %5, vcc, %21, %13, vcc\n\t"
        "v_subbrev_co_u32 %6, vcc, %22, %14, vcc\n\t"
        "v_subbrev_co_u32 %7, vcc, %23, %15, vcc\n\t"
        "v_cndmask_b32 %0, %0, %8, vcc\n\t"   // borrow: x < p, keep x
        "v_cndmask_b32 %1, %1, %9, vcc\n\t"
        "v_cndmask_b32 %2, %2, %10, vcc\n\t"
        "v_cndmask_b32 %3, %3, %11, vcc\n\t"
        "v_cndmask_b32 %4, %4, %12, vcc\n\t"
        "v_cndmask_b32 %5, %5, %13, vcc\n\t"
        "v_cndmask_b32 %6, %6, %14, vcc\n\t"
        "v_cndmask_b32 %7, %7, %15, vcc"
        : TY_O8(d)
        : TY_L8(x), TY_M8V(P)
        : "vcc");
}
// r = a - b mod 2^256, m = all ones if the subtraction wrapped else 0
__device__ __forceinline__ void ty_sub8(uint32_t (&r)[8], uint32_t& m, const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_cndmask_b32 %8, 0, -1, vcc"
        : TY_O8(r), "=&v"(m)
        : TY_L8(a), TY_L8(b)
        : "vcc");
}
// o = r + (p & m)
template <class P>
__device__ __forceinline__ void ty_addmask8(uint32_t (&o)[8], const uint32_t (&r)[8], uint32_t m) {
    asm("v_and_b32 %0, %17, %16\n\t"
        "v_and_b32 %1, %18, %16\n\t"
        "v_and_b32 %2, %19, %16\n\t"
        "v_and_b32 %3, %20, %16\n\t"
        "v_and_b32 %4, %21, %16\n\t"
        "v_and_b32 %5, %22, %16\n\t"
        "v_and_b32 %6, %23, %16\n\t"
        "v_and_b32 %7, %24, %16\n\t"
        "v_add_co_u32 %0, vcc, %8, %0\n\t"
        "v_addc_co_u32 %1, vcc, %9, %1, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %10, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %11, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %12, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %13, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %14, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %15, %7, vcc"
        : TY_O8(o)
        : TY_L8(r), "v"(m), TY_M8(P)
        : "vcc");
}
#endif

// r = a - p if a >= p else a          (a < 2p)
template <class P>
TY_HD void fe_reduce_once(Fe<P>& a) {
    constexpr int N = P::N;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (N == 8) {
        uint32_t d[8];
        ty_csub8<P>(d, a.v);
#pragma unroll
        for (int i = 0; i < 8; ++i) a.v[i] = d[i];
        return;
    }
#endif
    uint32_t d[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)a.v[i] - P::mod(i) - borrow;
        d[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < N; ++i) a.v[i] = d[i];
    }
}

template <class P>
TY_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    Fe<P> r;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (N == 8) {
        uint32_t t[8];
        ty_add8(t, a.v, b.v);
        ty_csub8<P>(r.v, t);
        return r;
    }
#endif
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
        r.v[i] = (uint32_t)t;
        c = t >> 32;
    }
    fe_reduce_once(r);  // both moduli leave the top bit of the top limb free: no carry out of limb N-1
    return r;
}

template <class P>
TY_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    Fe<P> r;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (N == 8) {
        uint32_t t[8], m;
        ty_sub8(t, m, a.v, b.v);
        ty_addmask8<P>(r.v, t, m);
        return r;
    }
#endif
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    // add p back under mask when the subtraction wrapped
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)r.v[i] + (P::mod(i) & mask) + c;
        r.v[i] = (uint32_t)t;
        c = t >> 32;
    }
    return r;
}

template <class P>
TY_HD Fe<P> fe_neg(const Fe<P>& a) {
    if (a.is_zero()) return a;
    constexpr int N = P::N;
    Fe<P> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)P::mod(i) - a.v[i] - borrow;
        r.v[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return r;
}

template <class P>
TY_HD Fe<P> fe_dbl(const Fe<P>& a) {
    constexpr int N = P::N;
    Fe<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r.v[i] = (a.v[i] << 1) | c;
        c = a.v[i] >> 31;
    }
    fe_reduce_once(r);
    return r;
}

// Montgomery product a*b*R^-1 mod p.
//
// Device code: product scanning (FIPS) with a 96-bit column accumulator -- every partial product is
// ONE v_mad_u64_u32 into the low 64 bits plus ONE v_addc_co_u32 capturing the carry, instead of the
// two carry additions per multiply of an operand-scanning CIOS.  Measured on MI355X
// (profiles/r01_ubench_*.txt, Fq): 58 vs 40 G mul/s.  The accumulator chain is written in inline
// assembly because the compiler has no way to express "mad with carry-out".
// Host code (and any non-HIP compiler): portable CIOS without the extra carry limb, valid because the
// top bit of both moduli's top limb is clear.
#if defined(__HIP_DEVICE_COMPILE__)
#define TY_MAC_VV(lo, hi, a, b) \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc")
#define TY_MAC_VS(lo, hi, a, b) \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "s"(b) : "vcc")
#endif
template <class P>
TY_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
#if defined(__HIP_DEVICE_COMPILE__)
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
#else
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t bi = b.v[i];
        uint64_t x = (uint64_t)a.v[0] * bi + t[0];
        uint32_t A = (uint32_t)(x >> 32);
        const uint32_t m = (uint32_t)x * P::INV;
        uint64_t y = (uint64_t)m * P::mod(0) + (uint32_t)x;
        uint32_t C = (uint32_t)(y >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            x = (uint64_t)a.v[j] * bi + t[j] + A;
            A = (uint32_t)(x >> 32);
            y = (uint64_t)m * P::mod(j) + (uint32_t)x + C;
            C = (uint32_t)(y >> 32);
            t[j - 1] = (uint32_t)y;
        }
        t[N - 1] = A + C;
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    fe_reduce_once(r);
    return r;
#endif
}

template <class P>
TY_HD Fe<P> fe_sqr(const Fe<P>& a) {
    return fe_mul(a, a);
}

// Montgomery residue -> canonical integer (ark-ff `into_repr`): multiply by 1.
template <class P>
TY_HD Fe<P> fe_from_mont(const Fe<P>& a) {
    constexpr int N = P::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = t[0] * P::INV;
        uint64_t y = (uint64_t)m * P::mod(0) + t[0];
        uint32_t C = (uint32_t)(y >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            y = (uint64_t)m * P::mod(j) + t[j] + C;
            C = (uint32_t)(y >> 32);
            t[j - 1] = (uint32_t)y;
        }
        t[N - 1] = C;
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    fe_reduce_once(r);
    return r;
}

template <class P>
TY_HD Fe<P> fe_to_mont(const Fe<P>& a) {
    return fe_mul(a, Fe<P>::r2());
}

// a^e for a little-endian u32 exponent of `words` words (not constant time; host-side use and
// tiny device kernels only).
template <class P>
TY_HD Fe<P> fe_pow(const Fe<P>& a, const uint32_t* e, int words) {
    Fe<P> acc = Fe<P>::one();
    bool started = false;
    for (int w = words - 1; w >= 0; --w) {
        for (int b = 31; b >= 0; --b) {
            if (started) acc = fe_sqr(acc);
            if ((e[w] >> b) & 1) {
                acc = started ? fe_mul(acc, a) : a;
                started = true;
            }
        }
    }
    return acc;
}

// a^-1 = a^(p-2)  (0 -> 0)
template <class P>
TY_HD Fe<P> fe_inv(const Fe<P>& a) {
    constexpr int N = P::N;
    uint32_t e[N];
    uint64_t borrow = 2;
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)P::mod(i) - borrow;
        e[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    return fe_pow(a, e, N);
}

using Fq = Fe<FqParams>;
using Fr = Fe<FrParams>;

}  // namespace ty
