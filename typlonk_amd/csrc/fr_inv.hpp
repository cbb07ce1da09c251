// Fr inversion by Bernstein-Yang divsteps on nine signed 30-bit limbs -- the device-side form of the ONE field inversion a
// proof's grand product needs (permutation/src/proving.rs:18-24: Z_j = prod num / prod den; prover.hip keeps the prefix and
// suffix products and divides by the total once).  Rounds 1-5 fetched the total, inverted it on the host and relaunched: a
// full drain of the context's stream inside round 2.  The recurrence, the packed 30-step core (fq30_divsteps30) and the
// exact-division updates are those of fq30.hpp's fq30_inv_divsteps, re-cut for the 255-bit modulus r:
//   * Theorem 11.2 (Bernstein-Yang 2019): floor((49 * 255 + 57) / 17) = 738 divsteps reach g = 0 -> 25 rounds of 30;
//   * r = 1 mod 2^30, so r^-1 mod 2^30 = 1 and the multiple of r that makes the Bezout update divisible is read off the
//     low column directly.
// In and out: arkworks' form (8 x 32-bit words of x * 2^256 mod r, canonical).  0 -> 0.  Host and device (the host shim
// checks it against a^(r-2) and Python's pow on the CPU).
#pragma once
#include "ff.hpp"
#include "fq30.hpp"
#include "fr30.hpp"

namespace ty {

constexpr int FR_DIVSTEP_ROUNDS = 25;

// (f, g) <- t (f, g) / 2^30, exact; limbs 0..7 in [0, 2^30), limb 8 signed
TY_HD void fr_divsteps_update_fg(int32_t (&f)[9], int32_t (&g)[9], const int32_t (&t)[4]) {
    int64_t cf = 0, cg = 0;
    FQ30_SMAD_VV(cf, t[0], f[0]);
    FQ30_SMAD_VV(cf, t[1], g[0]);
    FQ30_SMAD_VV(cg, t[2], f[0]);
    FQ30_SMAD_VV(cg, t[3], g[0]);
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        FQ30_SMAD_VV(cf, t[0], f[i]);
        FQ30_SMAD_VV(cg, t[2], f[i]);
        FQ30_SMAD_VV(cf, t[1], g[i]);
        FQ30_SMAD_VV(cg, t[3], g[i]);
        f[i - 1] = (int32_t)((uint32_t)cf & FR30_MASK);
        g[i - 1] = (int32_t)((uint32_t)cg & FR30_MASK);
        cf >>= 30;
        cg >>= 30;
    }
    f[8] = (int32_t)cf;
    g[8] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod r, both kept in (-2r, r) (the argument of fq30_divsteps_update_de with p -> r)
TY_HD void fr_divsteps_update_de(int32_t (&d)[9], int32_t (&e)[9], const int32_t (&t)[4]) {
    const int32_t sd = d[8] >> 31, se = e[8] >> 31;
    int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
    int64_t cd = 0, ce = 0;
    FQ30_SMAD_VV(cd, t[0], d[0]);
    FQ30_SMAD_VV(cd, t[1], e[0]);
    FQ30_SMAD_VV(ce, t[2], d[0]);
    FQ30_SMAD_VV(ce, t[3], e[0]);
    md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & FR30_MASK);   // r^-1 = 1 mod 2^30
    me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & FR30_MASK);
    FQ30_SMAD_VS(cd, md, (int32_t)fr30_r(0));
    FQ30_SMAD_VS(ce, me, (int32_t)fr30_r(0));
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        FQ30_SMAD_VV(cd, t[0], d[i]);
        FQ30_SMAD_VV(ce, t[2], d[i]);
        FQ30_SMAD_VV(cd, t[1], e[i]);
        FQ30_SMAD_VV(ce, t[3], e[i]);
        FQ30_SMAD_VS(cd, md, (int32_t)fr30_r(i));
        FQ30_SMAD_VS(ce, me, (int32_t)fr30_r(i));
        d[i - 1] = (int32_t)((uint32_t)cd & FR30_MASK);
        e[i - 1] = (int32_t)((uint32_t)ce & FR30_MASK);
        cd >>= 30;
        ce >>= 30;
    }
    d[8] = (int32_t)cd;
    e[8] = (int32_t)ce;
}

// a^-1 (Montgomery in, Montgomery out, both canonical).  `rounds_out` (host tests): rounds this call ran.
TY_HD Fr fr_inv_divsteps(const Fr& a, int* rounds_out = nullptr) {
    int32_t f[9], g[9], d[9], e[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {   // the words of a as a plain integer < r, re-cut into 30-bit limbs
        const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t t = a.v[wi] >> sh;
        if (sh > 2 && wi + 1 < 8) t |= a.v[wi + 1] << (32 - sh);
        f[i] = (int32_t)fr30_r(i);
        g[i] = (int32_t)(t & FR30_MASK);
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;
    int rounds = 0;
#pragma unroll 1
    for (; rounds < FR_DIVSTEP_ROUNDS; ++rounds) {
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) nz |= (uint32_t)g[i];
#if defined(__HIP_DEVICE_COMPILE__)
        if (!__any(nz != 0)) break;
#else
        if (nz == 0) break;
#endif
        int32_t t[4];
        eta = fq30_divsteps30(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        fr_divsteps_update_de(d, e, t);
        fr_divsteps_update_fg(f, g, t);
    }
    if (rounds_out) *rounds_out = rounds;
    // x = +-d + 2r in (0, 4r), exact limbs; then below r by two trial subtractions (2r, r)
    const int32_t sf = f[8] >> 31;
    uint32_t x[9];
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t s = (int64_t)((d[i] ^ sf) - sf) + (int64_t)(2ull * fr30_r(i)) + cy;
        x[i] = i < 8 ? ((uint32_t)s & FR30_MASK) : (uint32_t)s;
        cy = s >> 30;
    }
#pragma unroll
    for (int k = 2; k >= 1; --k) {
        uint32_t y[9];
        int64_t br = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int64_t s = (int64_t)x[i] - (int64_t)((uint64_t)k * fr30_r(i)) + br;
            y[i] = i < 8 ? ((uint32_t)s & FR30_MASK) : (uint32_t)s;
            br = s >> 30;
        }
        const bool ge = (int32_t)y[8] >= 0;   // x >= k r
#pragma unroll
        for (int i = 0; i < 9; ++i) x[i] = ge ? y[i] : x[i];
    }
    Fr o;   // 9 exact limbs of a value < r -> 8 words
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, li = bit / 30, off = bit % 30;
        uint32_t w = x[li] >> off;
        if (li + 1 < 9) w |= x[li + 1] << (30 - off);
        if (off > 28 && li + 2 < 9) w |= x[li + 2] << (60 - off);
        o.v[j] = w;
    }
    // o = (a R)^-1 as a plain integer: times R^3 / R gives a^-1 R
    const Fr r2 = Fr::r2();
    return fe_mul(o, fe_mul(r2, r2));
}

}  // namespace ty
