// Host-side test shim: exposes the product's shared host/device arithmetic headers
// (typlonk_amd/csrc/ff.hpp, fq30.hpp, g1.hpp) through a C ABI so pytest can check them against the
// Python big-int oracle without a GPU.  Built by __graft_entry__.build() with g++.
// All Fq / G1 arguments are in the C-ABI (arkworks) form: 12 u32 words per coordinate, R = 2^384.
#include "../../typlonk_amd/csrc/g1.hpp"
#include "../../typlonk_amd/csrc/g1_host64.hpp"
#include "../../typlonk_amd/csrc/fr_inv.hpp"
#include "../../typlonk_amd/csrc/transcript.hpp"
#include <string.h>
using namespace ty;

template <class F> static F ld(const uint32_t* p) { F f; memcpy(f.v, p, sizeof(f.v)); return f; }
template <class F> static void st(uint32_t* p, const F& f) { memcpy(p, f.v, sizeof(f.v)); }

static Fq30 ldq(const uint32_t* p) { uint32_t w[12]; memcpy(w, p, 48); return fq30_from_ark(w); }
static void stq(uint32_t* p, const Fq30& a) { uint32_t w[12]; fq30_to_ark(a, w); memcpy(p, w, 48); }

// scale a value up by adding k*p without changing it mod p (exercises the lazy-bound paths)
static Fq30 lift(const Fq30& a, int k) {
    Fq30 r = a;
    for (int j = 0; j < k; ++j) { Fq30 pp; for (int i = 0; i < 13; ++i) pp.v[i] = fq30_kp(1, i); r = fq30_add_lazy(r, pp); }
    return r;
}

extern "C" {
void shim_fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_mul(ld<Fr>(a), ld<Fr>(b))); }
void shim_fr_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_add(ld<Fr>(a), ld<Fr>(b))); }
void shim_fr_sub(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_sub(ld<Fr>(a), ld<Fr>(b))); }
void shim_fr_inv(const uint32_t* a, uint32_t* o) { st(o, fe_inv(ld<Fr>(a))); }
// the device's divsteps inversion of Fr (fr_inv.hpp), compiled for the host; returns the number of 30-step rounds it ran
int shim_fr_inv_divsteps(const uint32_t* a, uint32_t* o) {
    int rounds = 0;
    st(o, fr_inv_divsteps(ld<Fr>(a), &rounds));
    return rounds;
}
void shim_fr_from_mont(const uint32_t* a, uint32_t* o) { st(o, fe_from_mont(ld<Fr>(a))); }
void shim_fr_to_mont(const uint32_t* a, uint32_t* o) { st(o, fe_to_mont(ld<Fr>(a))); }

// Fq30: la / lb = number of extra multiples of p added to the operands before the operation
void shim_fq_roundtrip(const uint32_t* a, uint32_t* o) { stq(o, ldq(a)); }
void shim_fq_mul(const uint32_t* a, const uint32_t* b, int la, int lb, uint32_t* o) { stq(o, fq30_mul(lift(ldq(a), la), lift(ldq(b), lb))); }
void shim_fq_sqr(const uint32_t* a, int la, uint32_t* o) { stq(o, fq30_sqr(lift(ldq(a), la))); }
void shim_fq_add(const uint32_t* a, const uint32_t* b, int la, int lb, uint32_t* o) { stq(o, fq30_add_lazy(lift(ldq(a), la), lift(ldq(b), lb))); }
void shim_fq_sub(const uint32_t* a, const uint32_t* b, int la, int lb, uint32_t* o) {  // b + lb*p <= 6p
    stq(o, fq30_sub_lazy<6>(lift(ldq(a), la), lift(ldq(b), lb)));
}
void shim_fq_sub2(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* o) {
    stq(o, fq30_sub2_lazy<4>(lift(ldq(a), 1), lift(ldq(b), 1), lift(ldq(c), 1)));
}
void shim_fq_mul3(const uint32_t* a, uint32_t* o) { stq(o, fq30_mulk_lazy<3>(lift(ldq(a), 1))); }
void shim_fq_neg(const uint32_t* a, uint32_t* o) { stq(o, fq30_neg_lazy<1>(ldq(a))); }
void shim_fq_inv(const uint32_t* a, uint32_t* o) { stq(o, fq30_inv(ldq(a))); }
// host inversion (binary Euclid) against the Fermat ladder the device uses, la extra multiples of p on the input
int shim_fq_inv_agree(const uint32_t* a, int la) {
    const Fq30 x = lift(ldq(a), la);
    const Fq30 g = fq30_canon(fq30_inv(x)), f = fq30_canon(fq30_inv_fermat(fq30_canon(x)));
    for (int i = 0; i < 13; ++i)
        if (g.v[i] != f.v[i]) return 0;
    return 1;
}
// the divsteps inversion the device uses against the Fermat ladder AND the host's Euclid, digit for digit after
// canonicalisation; la extra multiples of p on the input (values up to 8p).  Returns the number of 30-divstep rounds the
// call ran (1..37), or -1 on a mismatch.
int shim_fq_inv_divsteps_agree(const uint32_t* a, int la) {
    const Fq30 x = lift(ldq(a), la);
    int rounds = 0;
    const Fq30 d = fq30_canon(fq30_inv_divsteps(x, &rounds));
    const Fq30 f = fq30_canon(fq30_inv_fermat(fq30_canon(x))), g = fq30_canon(fq30_inv_gcd(x));
    for (int i = 0; i < 13; ++i)
        if (d.v[i] != f.v[i] || d.v[i] != g.v[i]) return -1;
    return rounds;
}
// bulk form: n pseudo-random residues (xorshift seeded by `seed`, each lifted by (i mod 8) multiples of p when that keeps
// it below 8p), divsteps against Euclid (the Fermat ladder is ~50 us a call: sampled every `fermat_every`-th).
// Returns the number of mismatches; hist[r] counts the calls that ran r rounds (r <= 37).
int shim_fq_inv_divsteps_bulk(uint64_t seed, int n, int fermat_every, uint32_t* hist) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    int bad = 0;
    for (int it = 0; it < n; ++it) {
        Fq30 x;
        for (int i = 0; i < 13; ++i) x.v[i] = (uint32_t)next() & FQ30_MASK;
        x.v[12] &= 0x000fffffu;                       // < 2^380 < p
        if ((it & 15) == 3) for (int i = 1; i < 13; ++i) x.v[i] = 0;          // tiny values
        if ((it & 15) == 7) { Fq30 pm; for (int i = 0; i < 13; ++i) pm.v[i] = fq30_kp(1, i); x = fq30_sub_lazy<1>(pm, x); x.v[12] &= 0x001fffffu; x = fq30_canon(x); }
        const Fq30 xl = lift(x, it & 7 ? (it & 7) - 1 : 0);
        int rounds = 0;
        const Fq30 d = fq30_canon(fq30_inv_divsteps(xl, &rounds));
        const Fq30 g = fq30_canon(fq30_inv_gcd(x));
        bool ok = true;
        for (int i = 0; i < 13; ++i) ok = ok && d.v[i] == g.v[i];
        if (ok && fermat_every > 0 && it % fermat_every == 0) {
            const Fq30 f = fq30_canon(fq30_inv_fermat(fq30_canon(x)));
            for (int i = 0; i < 13; ++i) ok = ok && d.v[i] == f.v[i];
        }
        // and it IS the inverse: x * x^-1 = R (Montgomery one), unless x = 0
        if (ok && !fq30_is_zero_exact(fq30_canon(x))) {
            const Fq30 one = fq30_canon(fq30_mul(xl, d)), r1 = fq30_one();
            for (int i = 0; i < 13; ++i) ok = ok && one.v[i] == r1.v[i];
        }
        if (!ok) ++bad;
        if (rounds >= 0 && rounds <= 37) hist[rounds]++;
    }
    return bad;
}
int shim_fq_is_zero_mod(const uint32_t* a, int la) { return fq30_is_zero_mod(lift(ldq(a), la)) ? 1 : 0; }
// pack(unpack(w)) on raw words (any 384-bit pattern whose value is < 2^384)
void shim_fq_pack_unpack(const uint32_t* w, uint32_t* o) { uint32_t t[12]; memcpy(t, w, 48); uint32_t r[12]; fq30_pack(fq30_unpack(t), r); memcpy(o, r, 48); }

// points: 24 u32 (x||y, arkworks form); identity = all zero.
static G1Affine lda(const uint32_t* p) {
    bool z = true; for (int i = 0; i < 24; ++i) z = z && p[i] == 0;
    if (z) return G1Affine::inf();
    G1Affine a; a.x = ldq(p); a.y = ldq(p + 12); return a;
}
static void sta(uint32_t* p, const G1Affine& a) {
    if (a.is_inf()) { memset(p, 0, 96); return; }
    stq(p, a.x); stq(p + 12, a.y);
}
static void scramble(G1Xyzz& p, const uint32_t* z) {
    if (!z || p.is_inf()) return;
    Fq30 zv = ldq(z), zz = fq30_sqr(zv), zzz = fq30_mul(zz, zv);
    p.x = fq30_mul(p.x, zz); p.y = fq30_mul(p.y, zzz); p.zz = zz; p.zzz = zzz;
}

// out = a + (neg ? -b : b) through the mixed-add path (a lifted to XYZZ with a random Z when given)
void shim_g1_madd(const uint32_t* a, const uint32_t* b, int neg, const uint32_t* z, uint32_t* o) {
    G1Xyzz acc = G1Xyzz::from_affine(lda(a));
    scramble(acc, z);
    g1_madd(acc, lda(b), neg != 0);
    sta(o, g1_to_affine(acc));
}
void shim_g1_add(const uint32_t* a, const uint32_t* b, const uint32_t* z1, const uint32_t* z2, uint32_t* o) {
    G1Xyzz pa = G1Xyzz::from_affine(lda(a)), pb = G1Xyzz::from_affine(lda(b));
    scramble(pa, z1); scramble(pb, z2);
    sta(o, g1_to_affine(g1_add(pa, pb)));
}
void shim_g1_mul_small(const uint32_t* a, uint32_t k, uint32_t* o) {
    sta(o, g1_to_affine(g1_mul_small(G1Xyzz::from_affine(lda(a)), k)));
}
// long chain: acc = sum_i (+/-) pts[i] via repeated mixed adds, then n doublings -- checks that the
// lazy bounds hold over many dependent operations
void shim_g1_chain(const uint32_t* pts, int n, int ndbl, uint32_t* o) {
    G1Xyzz acc = G1Xyzz::inf();
    for (int i = 0; i < n; ++i) g1_madd(acc, lda(pts + 24 * i), (i % 3) == 1);
    for (int i = 0; i < ndbl; ++i) acc = g1_dbl(acc);
    G1Xyzz acc2 = g1_add(acc, acc);
    sta(o, g1_to_affine(acc2));
}
// the 64-bit host finish (g1_host64.hpp) on the device's packed form: every point goes through the 13 x 30-bit code
// into 48 packed words -- lifted to XYZZ with a Z derived from z, coordinates left lazily reduced, as the kernels store
// them -- and is then read, summed, doubled and normalised by the 6 x 64-bit code.  The sum runs as
//   ((p_0 + p_0) + (p_1 + (-p_1)) + p_2 + ... + p_{n-1}) * 2^ndbl
// so that the doubling and the cancellation branches of the addition are taken too.  Returns 0 for the identity.
int shim_h64_chain(const uint32_t* pts, int n, int ndbl, const uint32_t* z, uint32_t* o) {
    namespace H = ty::h64;
    auto dev = [&](const G1Affine& a, bool negate) {
        G1Xyzz q = G1Xyzz::from_affine(a);
        scramble(q, z);
        if (negate) q.y = fq30_neg_lazy<2>(q.y);
        q.x = fq30_add_lazy(q.x, fq30_mulk_lazy<2>(fq30_sub_lazy<2>(q.x, q.x)));  // + a multiple of p: still the same residue
        uint32_t w[48];
        uint32_t t[12];
        fq30_pack(q.x, t); memcpy(w, t, 48);
        fq30_pack(q.y, t); memcpy(w + 12, t, 48);
        fq30_pack(q.zz, t); memcpy(w + 24, t, 48);
        fq30_pack(q.zzz, t); memcpy(w + 36, t, 48);
        return H::xyzz_from_device(w);
    };
    H::Xyzz acc = H::inf();
    for (int i = 0; i < n; ++i) {
        const G1Affine a = lda(pts + 24 * i);
        if (i == 0) acc = H::xyzz_add(dev(a, false), dev(a, false));                       // doubling branch
        else if (i == 1) acc = H::xyzz_add(acc, H::xyzz_add(dev(a, false), dev(a, true)));  // cancellation -> identity
        else acc = H::xyzz_add(acc, dev(a, false));
    }
    for (int i = 0; i < ndbl; ++i) acc = H::xyzz_dbl(acc);
    uint64_t xy[12];
    if (!H::xyzz_to_affine(acc, xy)) return 0;
    memcpy(o, xy, 96);
    return 1;
}

// the native transcript's generator (csrc/transcript.hpp), for the published known-answer vectors of the crates
void shim_stdrng_words(const uint32_t key[8], uint64_t* out, int n) {   // StdRng::from_seed(key) -> n x next_u64
    ty::StdRng r = ty::StdRng::from_key(key);
    for (int i = 0; i < n; ++i) out[i] = r.next_u64();
}
void shim_seed_from_u64(uint64_t state, uint32_t key_out[8]) {          // rand_core SeedableRng::seed_from_u64 expansion
    ty::StdRng r(state);
    for (int i = 0; i < 8; ++i) key_out[i] = r.key[i];
}
}
