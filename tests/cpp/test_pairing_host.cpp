// The host pairing (typlonk_amd/host/pairing_host.hpp) -- CPU only.  What a pairing must satisfy: the G2 generator is on
// the twist and has order r, e is bilinear and non-degenerate, e(P, Q)^r == 1; plus e(G1, G2) printed coefficient by
// coefficient so that tests/test_host_mirror.py can compare it with the independent Python statement (oracle/pairing.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/typlonk.h"
#include "../../typlonk_amd/host/pairing_host.hpp"

using namespace typlonk::pairing;

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);      \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

static const uint64_t GX[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                               0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
static const uint64_t GY[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                               0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};

// k * G1 generator by repeated addition through the library's host fold (small k)
static G1Aff g1_small_multiple(int k) {
    std::vector<uint64_t> xy(12 * k);
    for (int i = 0; i < k; ++i) {
        memcpy(&xy[12 * i], GX, 48);
        memcpy(&xy[12 * i + 6], GY, 48);
    }
    uint64_t out[12];
    uint8_t inf = 0;
    REQUIRE(typlonk_g1_sum_host(xy.data(), nullptr, k, out, &inf) == 0);
    G1Aff p;
    memcpy(p.x.v, out, 48);
    memcpy(p.y.v, out + 6, 48);
    p.infinity = inf != 0;
    return p;
}
static Fq12 f12_pow_small(const Fq12& a, unsigned e) {
    Fq12 acc = f12_one();
    for (int b = 31; b >= 0; --b) {
        acc = f12_mul(acc, acc);
        if ((e >> b) & 1u) acc = f12_mul(acc, a);
    }
    return acc;
}

int main() {
    const G2Affine q = g2_generator();
    REQUIRE(g2_is_on_curve(q));
    static const uint32_t R_WORDS[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    REQUIRE(g2_mul_words(q, R_WORDS).infinity);                        // order r
    uint32_t five[8] = {5, 0, 0, 0, 0, 0, 0, 0};
    const G2Affine q5 = g2_mul_words(q, five);
    REQUIRE(g2_is_on_curve(q5) && !q5.infinity);
    REQUIRE(q5 == g2_add(g2_add(g2_add(q, q), g2_add(q, q)), q));
    std::printf("g2 ok\n");

    const G1Aff p = g1_small_multiple(1), p3 = g1_small_multiple(3);
    const Fq12 e = pairing(p, q);
    REQUIRE(!(e == f12_one()));                                        // non-degenerate
    REQUIRE(pairing(p3, q5) == f12_pow_small(e, 15));                  // bilinear
    const uint32_t three[8] = {3, 0, 0, 0, 0, 0, 0, 0};
    REQUIRE(pairing(p3, q) == pairing(p, g2_mul_words(q, three)));
    // e^r == 1 through r = 2^255-ish: square-and-multiply over the words of r
    Fq12 er = f12_one();
    for (int w = 7; w >= 0; --w)
        for (int b = 31; b >= 0; --b) {
            er = f12_mul(er, er);
            if ((R_WORDS[w] >> b) & 1u) er = f12_mul(er, e);
        }
    REQUIRE(er == f12_one());
    // e(P, Q) * e(-P, Q) == 1 with one final exponentiation
    G1Aff pn = p;
    pn.y = ty::fe_neg(p.y);
    const G1Aff ps[2] = {p, pn};
    const G2Affine qs[2] = {q, q};
    REQUIRE(pairing_product_is_one(ps, qs, 2));
    const G1Aff ps2[2] = {p, p3};
    REQUIRE(!pairing_product_is_one(ps2, qs, 2));
    std::printf("pairing ok\n");
    // e(G1, G2) as canonical integers, w^0 .. w^11 (for the cross-check against oracle/pairing.py)
    for (int i = 0; i < 12; ++i) {
        const Fq c = ty::fe_from_mont(e.c[i]);
        std::printf("e%d=", i);
        for (int k = 11; k >= 0; --k) std::printf("%08x", c.v[k]);
        std::printf("\n");
    }
    std::printf("all ok\n");
    return 0;
}
