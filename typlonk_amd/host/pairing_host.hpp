// BLS12-381 pairing and the reference's KZG verifier, host side ("CPU pairing verifier").
//
//   kzg::KzgScheme::verify      /root/reference/kzg/src/lib.rs:66-81
//         pairing(W, [s]G2 - z G2) == pairing(C - y G1, G2)
//   Srs::g2 / g2s               /root/reference/kzg/src/srs.rs:26-34
//
// The pairing lives in ark-ec / ark-bls12-381 0.3.0 (un-vendored, /root/reference/Cargo.lock:17-18, 28-29); it is
// restated from the published definition: optimal ate pairing, Miller loop over |x| = 0xd201000000010000 (x < 0),
// M-type sextic twist E': y^2 = x^3 + 4(1 + u) over Fq2 = Fq[u]/(u^2 + 1), final exponentiation (p^12 - 1)/r, with
// Fq12 written as Fq[w]/(w^12 - 2 w^6 + 2) (w^6 = 1 + u).  This is CPU glue with no GPU relevance: it favours being
// obviously the definition over speed (affine line steps, one square-and-multiply final exponentiation per CHECK -- a
// KZG verification is e(W, A) * e(-B, G2) == 1, i.e. two Miller loops and one exponentiation, ~0.06 s with the 6 x 64-bit
// field arithmetic of csrc/g1_host64.hpp).  An accept /
// reject decision does not depend on the normalisation of e.  Pinned by what a pairing must satisfy (bilinearity,
// non-degeneracy, order r) and, in the tests only, against the Python big-integer statement the test infrastructure keeps
// (tests/cpp/test_pairing_host.cpp, tests/test_host_mirror.py).  Field elements are arkworks residues (Montgomery, R = 2^384): the C-ABI form.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>

#include "../csrc/ff.hpp"
#include "../csrc/g1_host64.hpp"

namespace typlonk {
namespace pairing {

using ty::Fq;
// Fq arithmetic on 6 x 64-bit words (csrc/g1_host64.hpp): the 12 x 32-bit words of ty::Fq are the same 48 bytes
inline ty::h64::Fq q64(const Fq& a) {
    ty::h64::Fq r;
    std::memcpy(r.v, a.v, 48);
    return r;
}
inline Fq q32(const ty::h64::Fq& a) {
    Fq r;
    std::memcpy(r.v, a.v, 48);
    return r;
}
inline Fq qmul(const Fq& a, const Fq& b) { return q32(ty::h64::mul(q64(a), q64(b))); }
inline Fq qadd(const Fq& a, const Fq& b) { return q32(ty::h64::add(q64(a), q64(b))); }
inline Fq qsub(const Fq& a, const Fq& b) { return q32(ty::h64::sub(q64(a), q64(b))); }
inline Fq qdbl(const Fq& a) { return qadd(a, a); }
inline Fq qneg(const Fq& a) { return qsub(Fq::zero(), a); }
inline Fq qinv(const Fq& a) { return q32(ty::h64::inv(q64(a))); }
inline Fq fq_from_u64(uint64_t x) {
    Fq c = Fq::zero();
    c.v[0] = (uint32_t)x;
    c.v[1] = (uint32_t)(x >> 32);
    return ty::fe_to_mont(c);
}
inline Fq fq_from_words_be(const uint32_t (&be)[12]) {  // canonical integer, most significant word first -> residue
    Fq c;
    for (int i = 0; i < 12; ++i) c.v[i] = be[11 - i];
    return ty::fe_to_mont(c);
}

// ---- Fq2 = Fq[u]/(u^2 + 1) ------------------------------------------------------------------------------------------
struct Fq2 {
    Fq a, b;  // a + b u
    bool operator==(const Fq2& o) const { return a == o.a && b == o.b; }
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
};
inline Fq2 f2_zero() { return {Fq::zero(), Fq::zero()}; }
inline Fq2 f2_add(const Fq2& x, const Fq2& y) { return {qadd(x.a, y.a), qadd(x.b, y.b)}; }
inline Fq2 f2_sub(const Fq2& x, const Fq2& y) { return {qsub(x.a, y.a), qsub(x.b, y.b)}; }
inline Fq2 f2_neg(const Fq2& x) { return {qneg(x.a), qneg(x.b)}; }
inline Fq2 f2_mul(const Fq2& x, const Fq2& y) {
    return {qsub(qmul(x.a, y.a), qmul(x.b, y.b)), qadd(qmul(x.a, y.b), qmul(x.b, y.a))};
}
inline Fq2 f2_scalar(const Fq2& x, const Fq& k) { return {qmul(x.a, k), qmul(x.b, k)}; }
inline Fq2 f2_inv(const Fq2& x) {
    const Fq d = qinv(qadd(qmul(x.a, x.a), qmul(x.b, x.b)));
    return {qmul(x.a, d), qneg(qmul(x.b, d))};
}

// ---- E'(Fq2): y^2 = x^3 + 4(1 + u), affine ------------------------------------------------------------------------------
struct G2Affine {
    Fq2 x, y;
    bool infinity = false;
    bool operator==(const G2Affine& o) const {
        if (infinity || o.infinity) return infinity == o.infinity;
        return x == o.x && y == o.y;
    }
};
inline G2Affine g2_identity() {
    G2Affine r{};
    r.x = f2_zero();
    r.y = f2_zero();
    r.infinity = true;
    return r;
}
// the standard generator (ark-bls12-381 g2::G2_GENERATOR_X / _Y)
inline G2Affine g2_generator() {
    static const uint32_t X0[12] = {0x024aa2b2u, 0xf08f0a91u, 0x26080527u, 0x2dc51051u, 0xc6e47ad4u, 0xfa403b02u,
                                    0xb4510b64u, 0x7ae3d177u, 0x0bac0326u, 0xa805bbefu, 0xd48056c8u, 0xc121bdb8u};
    static const uint32_t X1[12] = {0x13e02b60u, 0x52719f60u, 0x7dacd3a0u, 0x88274f65u, 0x596bd0d0u, 0x9920b61au,
                                    0xb5da61bbu, 0xdc7f5049u, 0x334cf112u, 0x13945d57u, 0xe5ac7d05u, 0x5d042b7eu};
    static const uint32_t Y0[12] = {0x0ce5d527u, 0x727d6e11u, 0x8cc9cdc6u, 0xda2e351au, 0xadfd9baau, 0x8cbdd3a7u,
                                    0x6d429a69u, 0x5160d12cu, 0x923ac9ccu, 0x3baca289u, 0xe1935486u, 0x08b82801u};
    static const uint32_t Y1[12] = {0x0606c4a0u, 0x2ea734ccu, 0x32acd2b0u, 0x2bc28b99u, 0xcb3e287eu, 0x85a763afu,
                                    0x267492abu, 0x572e99abu, 0x3f370d27u, 0x5cec1da1u, 0xaaa9075fu, 0xf05f79beu};
    G2Affine g;
    g.x = {fq_from_words_be(X0), fq_from_words_be(X1)};
    g.y = {fq_from_words_be(Y0), fq_from_words_be(Y1)};
    g.infinity = false;
    return g;
}
inline bool g2_is_on_curve(const G2Affine& q) {
    if (q.infinity) return true;
    const Fq four = fq_from_u64(4);
    const Fq2 b2{four, four};
    return f2_mul(q.y, q.y) == f2_add(f2_mul(f2_mul(q.x, q.x), q.x), b2);
}
inline G2Affine g2_neg(const G2Affine& q) {
    if (q.infinity) return q;
    G2Affine r = q;
    r.y = f2_neg(q.y);
    return r;
}
inline G2Affine g2_add(const G2Affine& a, const G2Affine& b) {
    if (a.infinity) return b;
    if (b.infinity) return a;
    Fq2 lam;
    if (a.x == b.x) {
        if (!(a.y == b.y) || a.y.is_zero()) return g2_identity();
        lam = f2_mul(f2_scalar(f2_mul(a.x, a.x), fq_from_u64(3)), f2_inv(f2_add(a.y, a.y)));
    } else {
        lam = f2_mul(f2_sub(b.y, a.y), f2_inv(f2_sub(b.x, a.x)));
    }
    G2Affine r;
    r.x = f2_sub(f2_sub(f2_mul(lam, lam), a.x), b.x);
    r.y = f2_sub(f2_mul(lam, f2_sub(a.x, r.x)), a.y);
    r.infinity = false;
    return r;
}
// k * Q for a canonical 256-bit scalar (8 little-endian 32-bit words), MSB-first double-and-add
inline G2Affine g2_mul_words(const G2Affine& q, const uint32_t (&k)[8]) {
    G2Affine acc = g2_identity();
    for (int w = 7; w >= 0; --w)
        for (int b = 31; b >= 0; --b) {
            acc = g2_add(acc, acc);
            if ((k[w] >> b) & 1u) acc = g2_add(acc, q);
        }
    return acc;
}
// scalar given as an Fr Montgomery residue (the C-ABI form)
inline G2Affine g2_mul(const G2Affine& q, const ty::Fr& k_mont) {
    const ty::Fr c = ty::fe_from_mont(k_mont);
    uint32_t w[8];
    for (int i = 0; i < 8; ++i) w[i] = c.v[i];
    return g2_mul_words(q, w);
}

// ---- Fq12 = Fq[w]/(w^12 - 2 w^6 + 2) ----------------------------------------------------------------------------------
struct Fq12 {
    Fq c[12];
    bool operator==(const Fq12& o) const {
        for (int i = 0; i < 12; ++i)
            if (!(c[i] == o.c[i])) return false;
        return true;
    }
};
inline Fq12 f12_one() {
    Fq12 r;
    for (auto& x : r.c) x = Fq::zero();
    r.c[0] = Fq::one();
    return r;
}
inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
    Fq t[23];
    for (auto& x : t) x = Fq::zero();
    for (int i = 0; i < 12; ++i) {
        if (a.c[i].is_zero()) continue;
        for (int j = 0; j < 12; ++j) {
            if (b.c[j].is_zero()) continue;
            t[i + j] = qadd(t[i + j], qmul(a.c[i], b.c[j]));
        }
    }
    for (int i = 22; i >= 12; --i) {  // w^12 = 2 w^6 - 2
        const Fq two_c = qdbl(t[i]);
        t[i - 6] = qadd(t[i - 6], two_c);
        t[i - 12] = qsub(t[i - 12], two_c);
    }
    Fq12 r;
    for (int i = 0; i < 12; ++i) r.c[i] = t[i];
    return r;
}
// w -> -w: the p^6 Frobenius; the inverse of a unitary element (anything after the final exponentiation)
inline Fq12 f12_conj(const Fq12& a) {
    Fq12 r = a;
    for (int i = 1; i < 12; i += 2) r.c[i] = qneg(a.c[i]);
    return r;
}

// w^3 * l(P) for the line of slope lam / w through the untwisted psi(T) = (xT / w^2, yT / w^3):
// (lam xT - yT) - lam xP w^2 + yP w^3, with a + b u = (a - b) + b w^6.  The factor w^3 lies in Fq4 and dies in the
// final exponentiation.
inline Fq12 line(const Fq2& lam, const Fq2& xt, const Fq2& yt, const Fq& px, const Fq& py) {
    const Fq2 c0 = f2_sub(f2_mul(lam, xt), yt);
    const Fq2 c2 = f2_scalar(lam, qneg(px));
    Fq12 o;
    for (auto& x : o.c) x = Fq::zero();
    o.c[0] = qsub(c0.a, c0.b);
    o.c[6] = c0.b;
    o.c[2] = qsub(c2.a, c2.b);
    o.c[8] = c2.b;
    o.c[3] = py;
    return o;
}

struct G1Aff {
    Fq x, y;
    bool infinity;
};

constexpr uint64_t ATE_LOOP = 0xd201000000010000ull;  // |x|; x = -ATE_LOOP

inline Fq12 miller_loop(const G1Aff& p, const G2Affine& q) {
    if (p.infinity || q.infinity) return f12_one();
    Fq12 f = f12_one();
    G2Affine t = q;
    const Fq three = fq_from_u64(3);
    for (int bit = 62; bit >= 0; --bit) {  // below the leading one of the 64-bit loop count
        Fq2 lam = f2_mul(f2_scalar(f2_mul(t.x, t.x), three), f2_inv(f2_add(t.y, t.y)));
        f = f12_mul(f12_mul(f, f), line(lam, t.x, t.y, p.x, p.y));
        t = g2_add(t, t);
        if ((ATE_LOOP >> bit) & 1ull) {
            lam = f2_mul(f2_sub(q.y, t.y), f2_inv(f2_sub(q.x, t.x)));
            f = f12_mul(f, line(lam, t.x, t.y, p.x, p.y));
            t = g2_add(t, q);
        }
    }
    return f;
}

// (p^12 - 1) / r as 135 little-endian 32-bit words (computed and checked by tests/cpp/gen_final_exp.py)
#include "final_exp_words.inc"

inline Fq12 final_exponentiation(const Fq12& f) {
    Fq12 acc = f12_one();
    bool started = false;
    for (int w = FINAL_EXP_WORDS - 1; w >= 0; --w)
        for (int b = 31; b >= 0; --b) {
            if (started) acc = f12_mul(acc, acc);
            if ((FINAL_EXP[w] >> b) & 1u) {
                acc = started ? f12_mul(acc, f) : f;
                started = true;
            }
        }
    return f12_conj(acc);  // x < 0
}

inline Fq12 pairing(const G1Aff& p, const G2Affine& q) { return final_exponentiation(miller_loop(p, q)); }

// prod_i e(P_i, Q_i) == 1 with one final exponentiation
inline bool pairing_product_is_one(const G1Aff* ps, const G2Affine* qs, int count) {
    Fq12 f = f12_one();
    for (int i = 0; i < count; ++i) f = f12_mul(f, miller_loop(ps[i], qs[i]));
    return final_exponentiation(f) == f12_one();
}

}  // namespace pairing
}  // namespace typlonk
