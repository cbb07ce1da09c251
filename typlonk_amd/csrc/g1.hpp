// BLS12-381 G1 group law for the MSM hot path (replaces ark-ec 0.3.0 `add_assign_mixed`,
// `double_in_place`, `From<GroupProjective> for GroupAffine`; reference call sites
// /root/reference/kzg/src/lib.rs:49-52).
//
// E: y^2 = x^3 + 4 (a = 0).  Accumulators use extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is 8M+2S with no field inversion and bucket
// accumulation is made of nothing else.  The result of an MSM is a group element, and the library
// returns its unique canonical affine form, so the choice of coordinates cannot change a single
// output bit -- provided every exceptional case (identity operand, P+P, P+(-P)) is handled, which
// the functions below do explicitly.
//
// Field values are lazily reduced (fq30.hpp).  Invariants of a stored XYZZ point, in units of p:
//     X < 5.1    Y < 3.2    ZZ, ZZZ < 1.1      (all < 8p < 2^384, so they pack into 12 words)
// Affine points have canonical coordinates (< p).  Each line below carries the bound of its
// result; "m(A,B)" = 1 + A*B/630 is the bound of a Montgomery product of inputs < A p and < B p.
#pragma once
#include "fq30.hpp"

namespace ty {

// Affine point; identity is encoded (0, 0), which is not on the curve (0 != 0 + 4).  The C-ABI's
// separate `inf` flag byte is folded into this form at upload.
struct G1Affine {
    Fq30 x, y;  // canonical, Montgomery R = 2^390
    TY_HD bool is_inf() const { return fq30_is_zero_exact(x) && fq30_is_zero_exact(y); }
    static TY_HD G1Affine inf() {
        G1Affine r;
        r.x = fq30_zero();
        r.y = fq30_zero();
        return r;
    }
};

struct G1Xyzz {
    Fq30 x, y, zz, zzz;
    TY_HD bool is_inf() const { return fq30_is_zero_mod(zz); }
    static TY_HD G1Xyzz inf() {
        G1Xyzz r;
        r.x = fq30_one();
        r.y = fq30_one();
        r.zz = fq30_zero();
        r.zzz = fq30_zero();
        return r;
    }
    static TY_HD G1Xyzz from_affine(const G1Affine& p) {
        if (p.is_inf()) return inf();
        G1Xyzz r;
        r.x = p.x;
        r.y = p.y;
        r.zz = fq30_one();
        r.zzz = fq30_one();
        return r;
    }
};

// r*t - y*w with ONE Montgomery reduction (fq30_mul2_add): r*t + (4p - y)*w.  Needs y <= 4p.
// Result < 1 + (R*T + 4*W)/630 in units of p for r < R, t < T, w < W.
// -DG1_SPLIT_Y3 keeps the two separately reduced products (A/B measurements, tools/ubench2.hip).
TY_HD Fq30 g1_y3(const Fq30& r, const Fq30& t, const Fq30& y, const Fq30& w) {
#if defined(G1_SPLIT_Y3)
    return fq30_sub_lazy<2>(fq30_mul(r, t), fq30_mul(y, w));
#else
    return fq30_mul2_add(r, t, fq30_neg_lazy<4>(y), w);
#endif
}

// 2*(x, y) for an affine non-identity point, x, y < 1.1  (dbl-2008-s-1 with ZZ = ZZZ = 1)
TY_HD G1Xyzz g1_dbl_affine(const Fq30& x, const Fq30& y) {
    G1Xyzz r;
    const Fq30 u = fq30_mulk_lazy<2>(y);                               // < 2.2
    const Fq30 v = fq30_sqr(u);                                        // m(2.2,2.2) < 1.01
    if (fq30_is_zero_mod(v)) return G1Xyzz::inf();                     // y = 0: order-2 point (none on G1)
    const Fq30 w = fq30_mul(u, v);                                     // < 1.01
    const Fq30 s = fq30_mul(x, v);                                     // < 1.01
    const Fq30 m = fq30_mulk_lazy<3>(fq30_sqr(x));                     // 3 * 1.01 < 3.1
    r.x = fq30_sub_lazy<3>(fq30_sqr(m), fq30_mulk_lazy<2>(s));         // m(3.1,3.1) + 3 < 4.1   (2s < 2.1 <= 3)
    const Fq30 t = fq30_sub_lazy<5>(s, r.x);                           // 1.01 + 5 < 6.1         (X3 < 4.1 <= 5)
    r.y = g1_y3(m, t, y, w);                                           // < 3.1   (split: m(3.1,6.1) + 2; merged: 1 + (3.1*6.1 + 4*1.01)/630)
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P  (dbl-2008-s-1)
TY_HD G1Xyzz g1_dbl(const G1Xyzz& p) {
    if (p.is_inf()) return G1Xyzz::inf();
    G1Xyzz r;
    const Fq30 u = fq30_mulk_lazy<2>(p.y);                             // < 6.4
    const Fq30 v = fq30_sqr(u);                                        // m(6.4,6.4) < 1.07
    if (fq30_is_zero_mod(v)) return G1Xyzz::inf();
    const Fq30 w = fq30_mul(u, v);                                     // m(6.4,1.07) < 1.02
    const Fq30 s = fq30_mul(p.x, v);                                   // m(5.1,1.07) < 1.01
    const Fq30 m = fq30_mulk_lazy<3>(fq30_sqr(p.x));                   // 3 * m(5.1,5.1) < 3.2
    r.x = fq30_sub_lazy<3>(fq30_sqr(m), fq30_mulk_lazy<2>(s));         // m(3.2,3.2) + 3 < 4.1
    const Fq30 t = fq30_sub_lazy<5>(s, r.x);                           // < 6.1
    r.y = g1_y3(m, t, p.y, w);                                         // < 3.1   (Y < 3.2 <= 4)
    r.zz = fq30_mul(v, p.zz);                                          // < 1.01
    r.zzz = fq30_mul(w, p.zzz);                                        // < 1.01
    return r;
}

// acc += (qx, qy)   mixed addition, madd-2008-s; qx, qy < 1.1; the affine operand must not be the
// identity (callers test G1Affine::is_inf first).
TY_HD void g1_madd_xy(G1Xyzz& acc, const Fq30& qx, const Fq30& qy) {
    if (acc.is_inf()) {
        acc.x = qx;
        acc.y = qy;
        acc.zz = fq30_one();
        acc.zzz = fq30_one();
        return;
    }
    const Fq30 u2 = fq30_mul(qx, acc.zz);                              // < 1.01
    const Fq30 s2 = fq30_mul(qy, acc.zzz);                             // < 1.01
    const Fq30 p = fq30_sub_lazy<6>(u2, acc.x);                        // 1.01 + 6 < 7.1   (X1 < 5.1 <= 6)
    const Fq30 r = fq30_sub_lazy<4>(s2, acc.y);                        // 1.01 + 4 < 5.1   (Y1 < 3.2 <= 4)
    const Fq30 pp = fq30_sqr(p);                                       // m(7.1,7.1) < 1.09
    if (fq30_is_zero_mod(pp)) {                                        // P = 0  <=>  same x
        if (fq30_is_zero_mod(fq30_sqr(r))) {                           // and same y: doubling
            acc = g1_dbl_affine(qx, qy);
        } else {
            acc = G1Xyzz::inf();
        }
        return;
    }
    const Fq30 ppp = fq30_mul(p, pp);                                  // m(7.1,1.09) < 1.02
    const Fq30 q = fq30_mul(acc.x, pp);                                // m(5.1,1.09) < 1.01
    const Fq30 x3 = fq30_sub2_lazy<4>(fq30_sqr(r), ppp, fq30_mulk_lazy<2>(q));  // m(5.1,5.1) + 4 < 5.1  (ppp + 2q < 3.1 <= 4)
    const Fq30 t = fq30_sub_lazy<6>(q, x3);                            // 1.01 + 6 < 7.1
    acc.y = g1_y3(r, t, acc.y, ppp);                                   // < 3.1   (split: m(5.1,7.1) + 2; merged: 1 + (5.1*7.1 + 4*1.02)/630 < 1.07)
    acc.x = x3;
    acc.zz = fq30_mul(acc.zz, pp);                                     // < 1.01
    acc.zzz = fq30_mul(acc.zzz, ppp);                                  // < 1.01
}

// acc += (neg ? -q : q)
TY_HD void g1_madd(G1Xyzz& acc, const G1Affine& q, bool neg) {
    if (q.is_inf()) return;
    const Fq30 qy = neg ? fq30_neg_lazy<1>(q.y) : q.y;                 // p - y <= p
    g1_madd_xy(acc, q.x, qy);
}

// a + b, both XYZZ  (add-2008-s)
TY_HD G1Xyzz g1_add(const G1Xyzz& a, const G1Xyzz& b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const Fq30 u1 = fq30_mul(a.x, b.zz);                               // m(5.1,1.1) < 1.01
    const Fq30 u2 = fq30_mul(b.x, a.zz);                               // < 1.01
    const Fq30 s1 = fq30_mul(a.y, b.zzz);                              // < 1.01
    const Fq30 s2 = fq30_mul(b.y, a.zzz);                              // < 1.01
    const Fq30 p = fq30_sub_lazy<2>(u2, u1);                           // < 3.1
    const Fq30 r = fq30_sub_lazy<2>(s2, s1);                           // < 3.1
    const Fq30 pp = fq30_sqr(p);                                       // < 1.02
    if (fq30_is_zero_mod(pp)) {
        if (fq30_is_zero_mod(fq30_sqr(r))) return g1_dbl(a);
        return G1Xyzz::inf();
    }
    const Fq30 ppp = fq30_mul(p, pp);                                  // < 1.01
    const Fq30 q = fq30_mul(u1, pp);                                   // < 1.01
    G1Xyzz o;
    o.x = fq30_sub2_lazy<4>(fq30_sqr(r), ppp, fq30_mulk_lazy<2>(q));   // m(3.1,3.1) + 4 < 5.1
    const Fq30 t = fq30_sub_lazy<6>(q, o.x);                           // < 7.1
    o.y = g1_y3(r, t, s1, ppp);                                        // < 3.1   (s1 < 1.01 <= 4)
    o.zz = fq30_mul(fq30_mul(a.zz, b.zz), pp);                         // < 1.01
    o.zzz = fq30_mul(fq30_mul(a.zzz, b.zzz), ppp);                     // < 1.01
    return o;
}

// Canonical affine form (one field inversion).  Identity -> (0, 0).
TY_HD G1Affine g1_to_affine(const G1Xyzz& p) {
    if (p.is_inf()) return G1Affine::inf();
    const Fq30 t = fq30_inv(fq30_mul(p.zz, p.zzz));
    G1Affine r;
    r.x = fq30_canon(fq30_mul(p.x, fq30_mul(t, p.zzz)));  // X / ZZ
    r.y = fq30_canon(fq30_mul(p.y, fq30_mul(t, p.zz)));   // Y / ZZZ
    return r;
}

// ---- Jacobian doubling chains (fixed-base table set-up, srs_gen.hip) ---------------------------------------------------
// (X, Y, Z): x = X/Z^2, y = Y/Z^3.  A run of doublings costs 3S + 2M + one two-product reduction each (~6.0
// multiplication times) against the 8.0 of g1_dbl, and carries ONE denominator, so that a whole column of table entries
// can be normalised with one shared inversion.  Invariants as for XYZZ: X < 5.1, Y < 3.2, Z < 1.1 (units of p).
// Z = 0 (mod p) is the identity.
struct G1Jac {
    Fq30 x, y, z;
    TY_HD bool is_inf() const { return fq30_is_zero_mod(z); }
    static TY_HD G1Jac from_affine(const G1Affine& p) {   // p must not be the identity
        G1Jac r;
        r.x = p.x;
        r.y = p.y;
        r.z = fq30_one();
        return r;
    }
};
// 2*P.  With B = Y^2, V = 4B, S = X V = 4XY^2, M = 3X^2:  X3 = M^2 - 2S,  Y3 = M (S - X3) - 8B^2,  Z3 = 2YZ.
// (The identity doubles to itself: Z3 = 0; the other coordinates are then meaningless but bounded.)
TY_HD G1Jac g1_jac_dbl(const G1Jac& p) {
    G1Jac r;
    const Fq30 b = fq30_sqr(p.y);                                      // m(3.2,3.2) < 1.02
    const Fq30 v = fq30_mulk_lazy<2>(fq30_mulk_lazy<2>(b));            // < 4.1
    const Fq30 s = fq30_mul(p.x, v);                                   // m(5.1,4.1) < 1.04
    const Fq30 m = fq30_mulk_lazy<3>(fq30_sqr(p.x));                   // 3 * m(5.1,5.1) < 3.2
    r.x = fq30_sub_lazy<3>(fq30_sqr(m), fq30_mulk_lazy<2>(s));         // m(3.2,3.2) + 3 < 4.1   (2s < 2.1 <= 3)
    const Fq30 t = fq30_sub_lazy<5>(s, r.x);                           // 1.04 + 5 < 6.1         (X3 < 4.1 <= 5)
    r.y = g1_y3(m, t, fq30_mulk_lazy<2>(b), v);                        // m t - 2b v = m t - 8 b^2:  1 + (3.2*6.1 + 4*4.1)/630 < 1.06   (2b < 2.1 <= 4)
    r.z = fq30_mul(fq30_mulk_lazy<2>(p.y), p.z);                       // m(6.4,1.1) < 1.02
    return r;
}
// the same point in XYZZ form (ZZ = Z^2, ZZZ = Z^3)
TY_HD G1Xyzz g1_jac_to_xyzz(const G1Jac& p) {
    if (p.is_inf()) return G1Xyzz::inf();
    G1Xyzz r;
    r.x = p.x;
    r.y = p.y;
    r.zz = fq30_sqr(p.z);                                              // < 1.01
    r.zzz = fq30_mul(r.zz, p.z);                                       // < 1.01
    return r;
}
// canonical affine form given zinv = 1/Z
TY_HD G1Affine g1_jac_to_affine_with(const G1Jac& p, const Fq30& zinv) {
    const Fq30 zi2 = fq30_sqr(zinv);                                   // < 1.01
    G1Affine r;
    r.x = fq30_canon(fq30_mul(p.x, zi2));
    r.y = fq30_canon(fq30_mul(p.y, fq30_mul(zi2, zinv)));
    return r;
}

// k * P for a small unsigned k, double-and-add.
TY_HD G1Xyzz g1_mul_small(const G1Xyzz& p, uint32_t k) {
    G1Xyzz acc = G1Xyzz::inf();
    for (int b = 31; b >= 0; --b) {
        acc = g1_dbl(acc);
        if ((k >> b) & 1) acc = g1_add(acc, p);
    }
    return acc;
}

}  // namespace ty
