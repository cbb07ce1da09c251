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
#pragma once
#include "ff.hpp"

namespace ty {

// Affine base point as stored on the device: identity is encoded (0, 0), which is not on the
// curve (0 != 0 + 4).  The C-ABI's separate `inf` flag byte is folded into this form at upload.
struct G1Affine {
    Fq x, y;
    TY_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static TY_HD G1Affine inf() {
        G1Affine r;
        r.x = Fq::zero();
        r.y = Fq::zero();
        return r;
    }
};

struct G1Xyzz {
    Fq x, y, zz, zzz;
    TY_HD bool is_inf() const { return zz.is_zero(); }
    static TY_HD G1Xyzz inf() {
        G1Xyzz r;
        r.x = Fq::one();
        r.y = Fq::one();
        r.zz = Fq::zero();
        r.zzz = Fq::zero();
        return r;
    }
    static TY_HD G1Xyzz from_affine(const G1Affine& p) {
        if (p.is_inf()) return inf();
        G1Xyzz r;
        r.x = p.x;
        r.y = p.y;
        r.zz = Fq::one();
        r.zzz = Fq::one();
        return r;
    }
};

// 2*(x, y) for an affine non-identity point  (dbl-2008-s-1 with ZZ = ZZZ = 1)
TY_HD G1Xyzz g1_dbl_affine(const Fq& x, const Fq& y) {
    G1Xyzz r;
    if (y.is_zero()) return G1Xyzz::inf();  // order-2 point; none on G1, kept for totality
    Fq u = fe_dbl(y);
    Fq v = fe_sqr(u);
    Fq w = fe_mul(u, v);
    Fq s = fe_mul(x, v);
    Fq xx = fe_sqr(x);
    Fq m = fe_add(fe_dbl(xx), xx);
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P  (dbl-2008-s-1)
TY_HD G1Xyzz g1_dbl(const G1Xyzz& p) {
    if (p.is_inf() || p.y.is_zero()) return G1Xyzz::inf();
    G1Xyzz r;
    Fq u = fe_dbl(p.y);
    Fq v = fe_sqr(u);
    Fq w = fe_mul(u, v);
    Fq s = fe_mul(p.x, v);
    Fq xx = fe_sqr(p.x);
    Fq m = fe_add(fe_dbl(xx), xx);
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, p.y));
    r.zz = fe_mul(v, p.zz);
    r.zzz = fe_mul(w, p.zzz);
    return r;
}

// acc += (qx, qy)   mixed addition, madd-2008-s; the affine operand must not be the identity
// (callers test G1Affine::is_inf first).
TY_HD void g1_madd_xy(G1Xyzz& acc, const Fq& qx, const Fq& qy) {
    if (acc.is_inf()) {
        acc.x = qx;
        acc.y = qy;
        acc.zz = Fq::one();
        acc.zzz = Fq::one();
        return;
    }
    Fq u2 = fe_mul(qx, acc.zz);
    Fq s2 = fe_mul(qy, acc.zzz);
    Fq p = fe_sub(u2, acc.x);
    Fq r = fe_sub(s2, acc.y);
    if (p.is_zero()) {
        if (r.is_zero()) {
            acc = g1_dbl_affine(qx, qy);
        } else {
            acc = G1Xyzz::inf();
        }
        return;
    }
    Fq pp = fe_sqr(p);
    Fq ppp = fe_mul(p, pp);
    Fq q = fe_mul(acc.x, pp);
    Fq x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(q));
    Fq y3 = fe_sub(fe_mul(r, fe_sub(q, x3)), fe_mul(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul(acc.zz, pp);
    acc.zzz = fe_mul(acc.zzz, ppp);
}

// acc += (neg ? -q : q)
TY_HD void g1_madd(G1Xyzz& acc, const G1Affine& q, bool neg) {
    if (q.is_inf()) return;
    Fq qy = neg ? fe_neg(q.y) : q.y;
    g1_madd_xy(acc, q.x, qy);
}

// a + b, both XYZZ  (add-2008-s)
TY_HD G1Xyzz g1_add(const G1Xyzz& a, const G1Xyzz& b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    Fq u1 = fe_mul(a.x, b.zz);
    Fq u2 = fe_mul(b.x, a.zz);
    Fq s1 = fe_mul(a.y, b.zzz);
    Fq s2 = fe_mul(b.y, a.zzz);
    Fq p = fe_sub(u2, u1);
    Fq r = fe_sub(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) return g1_dbl(a);
        return G1Xyzz::inf();
    }
    Fq pp = fe_sqr(p);
    Fq ppp = fe_mul(p, pp);
    Fq q = fe_mul(u1, pp);
    G1Xyzz o;
    o.x = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(q));
    o.y = fe_sub(fe_mul(r, fe_sub(q, o.x)), fe_mul(s1, ppp));
    o.zz = fe_mul(fe_mul(a.zz, b.zz), pp);
    o.zzz = fe_mul(fe_mul(a.zzz, b.zzz), ppp);
    return o;
}

TY_HD G1Xyzz g1_neg(const G1Xyzz& a) {
    G1Xyzz r = a;
    r.y = fe_neg(a.y);
    return r;
}

// Canonical affine form (one field inversion).  Identity -> (0, 0) (device encoding).
TY_HD G1Affine g1_to_affine(const G1Xyzz& p) {
    if (p.is_inf()) return G1Affine::inf();
    Fq t = fe_inv(fe_mul(p.zz, p.zzz));
    G1Affine r;
    r.x = fe_mul(p.x, fe_mul(t, p.zzz));  // X / ZZ
    r.y = fe_mul(p.y, fe_mul(t, p.zz));   // Y / ZZZ
    return r;
}

// k * P for a small unsigned k (bucket-reduction segment offsets), double-and-add.
TY_HD G1Xyzz g1_mul_small(const G1Xyzz& p, uint32_t k) {
    G1Xyzz acc = G1Xyzz::inf();
    for (int b = 31; b >= 0; --b) {
        acc = g1_dbl(acc);
        if ((k >> b) & 1) acc = g1_add(acc, p);
    }
    return acc;
}

}  // namespace ty
