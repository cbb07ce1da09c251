// Host-side test shim: exposes the product's shared host/device arithmetic headers
// (typlonk_amd/csrc/ff.hpp, g1.hpp) through a C ABI so pytest can check them against the
// Python big-int oracle without a GPU.  Built by __graft_entry__.build() with g++.
#include "../../typlonk_amd/csrc/g1.hpp"
#include <string.h>
using namespace ty;

template <class F> static F ld(const uint32_t* p) { F f; memcpy(f.v, p, sizeof(f.v)); return f; }
template <class F> static void st(uint32_t* p, const F& f) { memcpy(p, f.v, sizeof(f.v)); }

extern "C" {
void shim_fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_mul(ld<Fr>(a), ld<Fr>(b))); }
void shim_fr_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_add(ld<Fr>(a), ld<Fr>(b))); }
void shim_fr_sub(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_sub(ld<Fr>(a), ld<Fr>(b))); }
void shim_fr_inv(const uint32_t* a, uint32_t* o) { st(o, fe_inv(ld<Fr>(a))); }
void shim_fr_from_mont(const uint32_t* a, uint32_t* o) { st(o, fe_from_mont(ld<Fr>(a))); }
void shim_fr_to_mont(const uint32_t* a, uint32_t* o) { st(o, fe_to_mont(ld<Fr>(a))); }
void shim_fq_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_mul(ld<Fq>(a), ld<Fq>(b))); }
void shim_fq_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_add(ld<Fq>(a), ld<Fq>(b))); }
void shim_fq_sub(const uint32_t* a, const uint32_t* b, uint32_t* o) { st(o, fe_sub(ld<Fq>(a), ld<Fq>(b))); }
void shim_fq_neg(const uint32_t* a, uint32_t* o) { st(o, fe_neg(ld<Fq>(a))); }
void shim_fq_dbl(const uint32_t* a, uint32_t* o) { st(o, fe_dbl(ld<Fq>(a))); }
void shim_fq_inv(const uint32_t* a, uint32_t* o) { st(o, fe_inv(ld<Fq>(a))); }
void shim_fq_from_mont(const uint32_t* a, uint32_t* o) { st(o, fe_from_mont(ld<Fq>(a))); }

// points: 24 u32 (x||y Montgomery); identity = all zero.  acc/out: affine, same encoding.
static G1Affine lda(const uint32_t* p) { G1Affine a; a.x = ld<Fq>(p); a.y = ld<Fq>(p + 12); return a; }
static void sta(uint32_t* p, const G1Affine& a) { st(p, a.x); st(p + 12, a.y); }

// out = a + (neg ? -b : b) through the mixed-add path (a lifted to XYZZ with a random-looking Z
// when `scramble` != 0 so the projective formulas are really exercised)
void shim_g1_madd(const uint32_t* a, const uint32_t* b, int neg, const uint32_t* z, uint32_t* o) {
    G1Xyzz acc = G1Xyzz::from_affine(lda(a));
    if (z && !acc.is_inf()) {
        Fq zz = fe_sqr(ld<Fq>(z)), zzz = fe_mul(zz, ld<Fq>(z));
        acc.x = fe_mul(acc.x, zz); acc.y = fe_mul(acc.y, zzz); acc.zz = zz; acc.zzz = zzz;
    }
    g1_madd(acc, lda(b), neg != 0);
    sta(o, g1_to_affine(acc));
}
void shim_g1_add(const uint32_t* a, const uint32_t* b, const uint32_t* z1, const uint32_t* z2, uint32_t* o) {
    G1Xyzz pa = G1Xyzz::from_affine(lda(a)), pb = G1Xyzz::from_affine(lda(b));
    if (z1 && !pa.is_inf()) { Fq zz = fe_sqr(ld<Fq>(z1)), zzz = fe_mul(zz, ld<Fq>(z1));
        pa.x = fe_mul(pa.x, zz); pa.y = fe_mul(pa.y, zzz); pa.zz = zz; pa.zzz = zzz; }
    if (z2 && !pb.is_inf()) { Fq zz = fe_sqr(ld<Fq>(z2)), zzz = fe_mul(zz, ld<Fq>(z2));
        pb.x = fe_mul(pb.x, zz); pb.y = fe_mul(pb.y, zzz); pb.zz = zz; pb.zzz = zzz; }
    sta(o, g1_to_affine(g1_add(pa, pb)));
}
void shim_g1_mul_small(const uint32_t* a, uint32_t k, uint32_t* o) {
    sta(o, g1_to_affine(g1_mul_small(G1Xyzz::from_affine(lda(a)), k)));
}
}
