// Host-side finish of an MSM on 6 x 64-bit words (host only; the device arithmetic is fq30.hpp / g1.hpp).
//
// What the host does per MSM is small and serial -- add the bit planes of the bucket reduction, one Horner pass over
// powers of two (~20 doublings + ~40 additions), one affine normalisation -- but it sits at the very end of every MSM
// and of every prover round.  The 13 x 30-bit limb code shared with the device spends ~75 ns per multiplication on a
// CPU (338 64-bit multiply-adds); a plain 6 x 64-bit Montgomery multiplication (36 + 36 with unsigned __int128) takes a
// third of that.  R = 2^384 is arkworks' own form, so the affine result needs no conversion on the way out.
//
// Device values arrive as 12 packed words holding x * 2^390 mod p, lazily reduced (< 8p < 2^384); one Montgomery
// multiplication by 2^378 turns them into x * 2^384, fully reduced.
#pragma once
#include <vector>
#include <stdint.h>
#include <string.h>

#include "fq30.hpp"  // Fq30U384 and the binary-Euclid inversion

namespace ty {
namespace h64 {

struct Fq {
    uint64_t v[6];
};
constexpr uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                           0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
constexpr uint64_t INV = 0x89f3fffcfffcfffdull;  // -p^-1 mod 2^64

inline bool is_zero(const Fq& a) { return !(a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5]); }
inline bool eq(const Fq& a, const Fq& b) { return memcmp(a.v, b.v, sizeof(a.v)) == 0; }
inline bool geq_p(const uint64_t (&a)[6]) {
    for (int i = 5; i >= 0; --i)
        if (a[i] != P[i]) return a[i] > P[i];
    return true;
}
inline void sub_p(uint64_t (&a)[6]) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 6; ++i) {
        const unsigned __int128 t = (unsigned __int128)a[i] - P[i] - (uint64_t)br;
        a[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
// a * b / 2^384 mod p, result < p.  a < 8p, b < p (or both < p): the value before the last step is < 2p.
inline Fq mul(const Fq& a, const Fq& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 6; ++j) {
            c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV;
        c = ((unsigned __int128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 6; ++j) {
            c += (unsigned __int128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    Fq r;
    uint64_t w[6] = {t[0], t[1], t[2], t[3], t[4], t[5]};
    if (t[6] || geq_p(w)) sub_p(w);
    memcpy(r.v, w, sizeof(w));
    return r;
}
inline Fq sqr(const Fq& a) { return mul(a, a); }
inline Fq add(const Fq& a, const Fq& b) {  // both < p
    uint64_t w[6];
    unsigned __int128 c = 0;
    for (int i = 0; i < 6; ++i) {
        c += (unsigned __int128)a.v[i] + b.v[i];
        w[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_p(w)) sub_p(w);  // a + b < 2p < 2^384: no carry out
    Fq r;
    memcpy(r.v, w, sizeof(w));
    return r;
}
inline Fq sub(const Fq& a, const Fq& b) {  // both < p
    uint64_t w[6];
    unsigned __int128 br = 0;
    for (int i = 0; i < 6; ++i) {
        const unsigned __int128 t = (unsigned __int128)a.v[i] - b.v[i] - (uint64_t)br;
        w[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 6; ++i) {
            c += (unsigned __int128)w[i] + P[i];
            w[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    Fq r;
    memcpy(r.v, w, sizeof(w));
    return r;
}
inline Fq dbl(const Fq& a) { return add(a, a); }
// a^-1 in the same (Montgomery, R = 2^384) form; 0 -> 0
inline Fq inv(const Fq& a) {
    if (is_zero(a)) return a;
    Fq30U384 u;
    memcpy(u.w, a.v, sizeof(u.w));
    const Fq30U384 i = fq30_u384_modinv(u);  // (x R)^-1 as a plain integer
    Fq r, r3 = {{0xed48ac6bd94ca1e0ull, 0x315f831e03a7adf8ull, 0x9a53352a615e29ddull, 0x34c04e5e921e1761ull,
                 0x2512d43565724728ull, 0x0aa6346091755d4dull}};  // R^3 mod p: (x R)^-1 * R^3 / R = x^-1 R
    memcpy(r.v, i.w, sizeof(r.v));
    return mul(r, r3);
}

// ---- G1 in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; identity: ZZ = 0) --------------------------------
struct Xyzz {
    Fq x, y, zz, zzz;
};
inline Xyzz inf() {
    Xyzz r;
    memset(&r, 0, sizeof(r));
    return r;
}
inline bool is_inf(const Xyzz& a) { return is_zero(a.zz); }
// 12 packed device words (x * 2^390, < 8p) -> x * 2^384, reduced
inline Fq from_device(const uint32_t* w) {
    Fq v;
    memcpy(v.v, w, 48);
    const Fq c378 = {{0, 0, 0, 0, 0, 0x0400000000000000ull}};  // 2^378 as a plain integer
    return mul(v, c378);
}
inline Xyzz xyzz_from_device(const uint32_t* p) {
    Xyzz r;
    r.x = from_device(p);
    r.y = from_device(p + 12);
    r.zz = from_device(p + 24);
    r.zzz = from_device(p + 36);
    return r;
}
// dbl-2008-s-1 (a = 0)
inline Xyzz xyzz_dbl(const Xyzz& a) {
    if (is_inf(a) || is_zero(a.y)) return inf();
    const Fq u = dbl(a.y), v = sqr(u), w = mul(u, v), s = mul(a.x, v);
    const Fq x2 = sqr(a.x), m = add(dbl(x2), x2);
    Xyzz r;
    r.x = sub(sqr(m), dbl(s));
    r.y = sub(mul(m, sub(s, r.x)), mul(w, a.y));
    r.zz = mul(v, a.zz);
    r.zzz = mul(w, a.zzz);
    return r;
}
// add-2008-s with the exceptional cases
inline Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) {
    if (is_inf(a)) return b;
    if (is_inf(b)) return a;
    const Fq u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz), s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
    const Fq p = sub(u2, u1), r = sub(s2, s1);
    if (is_zero(p)) return is_zero(r) ? xyzz_dbl(a) : inf();
    const Fq pp = sqr(p), ppp = mul(p, pp), q = mul(u1, pp);
    Xyzz o;
    o.x = sub(sub(sqr(r), ppp), dbl(q));
    o.y = sub(mul(r, sub(q, o.x)), mul(s1, ppp));
    o.zz = mul(mul(a.zz, b.zz), pp);
    o.zzz = mul(mul(a.zzz, b.zzz), ppp);
    return o;
}
// a + (bx, by): mixed addition (madd-2008-s), the affine operand not the identity; 8M + 2S instead of 12M + 2S
inline Xyzz xyzz_madd(const Xyzz& a, const Fq& bx, const Fq& by, const Fq& one) {
    if (is_inf(a)) {
        Xyzz r;
        r.x = bx;
        r.y = by;
        r.zz = one;
        r.zzz = one;
        return r;
    }
    const Fq u2 = mul(bx, a.zz), s2 = mul(by, a.zzz);
    const Fq p = sub(u2, a.x), r = sub(s2, a.y);
    if (is_zero(p)) {
        if (!is_zero(r)) return inf();
        Xyzz b;
        b.x = bx;
        b.y = by;
        b.zz = one;
        b.zzz = one;
        return xyzz_dbl(b);
    }
    const Fq pp = sqr(p), ppp = mul(p, pp), q = mul(a.x, pp);
    Xyzz o;
    o.x = sub(sub(sqr(r), ppp), dbl(q));
    o.y = sub(mul(r, sub(q, o.x)), mul(a.y, ppp));
    o.zz = mul(a.zz, pp);
    o.zzz = mul(a.zzz, ppp);
    return o;
}
// canonical affine forms of n points with ONE field inversion (Montgomery's trick over the denominators ZZ * ZZZ);
// ok[i] = 0 for the identity (its out_xy slot is left untouched)
inline void xyzz_to_affine_batch(const Xyzz* pts, size_t n, uint64_t* out_xy /* n * 12 */, char* ok) {
    std::vector<size_t> live;
    std::vector<Fq> den, pre;   // denominators of the live points and their running products
    for (size_t i = 0; i < n; ++i) {
        ok[i] = is_inf(pts[i]) ? 0 : 1;
        if (!ok[i]) continue;
        live.push_back(i);
        den.push_back(mul(pts[i].zz, pts[i].zzz));
        pre.push_back(pre.empty() ? den.back() : mul(pre.back(), den.back()));
    }
    if (live.empty()) return;
    Fq invr = inv(pre.back());
    for (size_t k = live.size(); k-- > 0;) {
        const Fq t = k ? mul(invr, pre[k - 1]) : invr;   // 1 / den[k]
        if (k) invr = mul(invr, den[k]);
        const Xyzz& p = pts[live[k]];
        const Fq x = mul(p.x, mul(t, p.zzz)), y = mul(p.y, mul(t, p.zz));
        memcpy(out_xy + 12 * live[k], x.v, 48);
        memcpy(out_xy + 12 * live[k] + 6, y.v, 48);
    }
}
// canonical affine point in arkworks' words (x || y, Montgomery R = 2^384); false for the identity
inline bool xyzz_to_affine(const Xyzz& a, uint64_t out_xy[12]) {
    if (is_inf(a)) return false;
    const Fq t = inv(mul(a.zz, a.zzz));
    const Fq x = mul(a.x, mul(t, a.zzz)), y = mul(a.y, mul(t, a.zz));
    memcpy(out_xy, x.v, 48);
    memcpy(out_xy + 6, y.v, 48);
    return true;
}

}  // namespace h64
}  // namespace ty
