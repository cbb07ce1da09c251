// The reference's own unit tests for this path, restated against the C++ host mirror
// (typlonk_amd/host/typlonk_host.hpp) -- needs a GPU (everything below the mirror is the HIP library).
//   commit       /root/reference/kzg/src/lib.rs:95-109
//   scalar_mul   /root/reference/kzg/src/lib.rs:160-171
//   l0           /root/reference/plonk/src/utils.rs:150-177
//   interpolate -> commit chain, plonk/src/builder.rs:84-88
#include <cstdio>
#include <cstdlib>

#include "../../typlonk_amd/host/typlonk_host.hpp"

using namespace typlonk;
using kzg::KzgScheme;
using kzg::Poly;
using kzg::Srs;

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);      \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

static void commit(const Context& ctx) {
    Srs srs = Srs::from_secret(ctx, Fr(2), 10);
    REQUIRE(srs.len() == 13);
    KzgScheme scheme(srs);
    Poly poly = Poly::from_coefficients_slice({Fr(1), Fr(2), Fr(3)});
    auto commitment = scheme.commit(poly);
    Fr d(1);
    // commitment == G * p(2)
    auto g = srs.g1_ref()[0];
    REQUIRE(commitment.p == kzg::g1_mul(ctx, g, poly.evaluate(Fr(2))));
    REQUIRE(poly.evaluate(d) == Fr(6));
    auto opening = scheme.open(poly, d);
    REQUIRE(opening.eval() == Fr(6));
    // scheme.verify(): e(W, [s - z]G2) == e(C - yG, G2); with the secret known (s = 2) this is the
    // G1 statement (s - z) W == C - y G (the pairing itself is out of scope)
    auto lhs = kzg::g1_mul(ctx, opening.p, Fr(2) - d);
    auto rhs = kzg::g1_add(commitment.p, kzg::g1_neg(ctx, kzg::g1_mul(ctx, g, opening.y)));
    REQUIRE(lhs == rhs);
    // assert!(srs.len() > polynomial.degree()) -> exception
    bool threw = false;
    try {
        scheme.commit(Poly::from_coefficients_vec(std::vector<Fr>(14, Fr(1))));
    } catch (const std::runtime_error&) {
        threw = true;
    }
    REQUIRE(threw);
    // identity() == G
    REQUIRE(scheme.identity().p == g);
    // zero polynomial commits to the identity (0, 1, inf)
    auto z = scheme.commit(Poly::from_coefficients_vec({Fr(0), Fr(0)}));
    REQUIRE(z.p.infinity);
    std::puts("commit ok");
}

static void scalar_mul(const Context& ctx) {
    Srs srs = Srs::from_secret(ctx, Fr(0x1234567) * Fr(0x89abcdef), 5);
    KzgScheme scheme(srs);
    Poly poly = Poly::from_coefficients_slice({Fr(1), Fr(2), Fr(3)});
    auto commit1 = scheme.commit(poly);
    auto commit2 = scheme.commit(poly * Fr(9));
    REQUIRE(kzg::g1_mul(ctx, commit1.p, Fr(9)) == commit2.p);
    std::puts("scalar_mul ok");
}

// plonk/src/utils.rs:150-159: l0 = (X^n - 1) / (n (X - 1)) = (1/n) sum X^i
static Poly l0_poly(const poly::Radix2EvaluationDomain& domain) {
    std::vector<Fr> c(domain.size(), domain.size_inv);
    return Poly::from_coefficients_vec(c);
}
static void l0(const Context& ctx) {
    poly::Radix2EvaluationDomain domain(ctx, 1ull << 16);
    auto l0 = l0_poly(domain);
    auto evals = poly::evaluate_over_domain(l0, domain);
    Fr sum;
    for (auto& e : evals) sum += e;
    REQUIRE(sum == Fr::one());
    REQUIRE(evals[0] == Fr::one());
    REQUIRE(l0.evaluate(domain.element(0)) == Fr::one());
    REQUIRE(l0.evaluate(domain.element(5)).is_zero());
    std::puts("l0 ok");
}

static void interpolate_then_commit(const Context& ctx) {
    const uint64_t n = 1 << 10;
    poly::Radix2EvaluationDomain domain(ctx, n);
    Srs srs = Srs::from_secret(ctx, Fr(2), n);
    KzgScheme scheme(srs);
    std::vector<Fr> evals(n);
    Fr x(3);
    for (auto& e : evals) { e = x; x = x * x + Fr(7); }
    Poly p = poly::interpolate(evals, domain);
    // the interpolant reproduces the evaluations ...
    REQUIRE(p.evaluate(domain.element(0)) == evals[0]);
    REQUIRE(p.evaluate(domain.element(77)) == evals[77]);
    // ... and commit(p) == [p(s)]G
    auto g = srs.g1_ref()[0];
    REQUIRE(scheme.commit(p).p == kzg::g1_mul(ctx, g, p.evaluate(Fr(2))));
    // a selector-like column with trailing zero coefficients is trimmed (sets the MSM length)
    std::vector<Fr> c = {Fr(5), Fr(7), Fr(0), Fr(0)};
    poly::Radix2EvaluationDomain d4(ctx, 4);
    Poly q = poly::interpolate(d4.fft(c), d4);
    REQUIRE(q.coeffs.size() == 2 && q.degree() == 1);
    // a GROUP of columns through one batched transform (builder.rs:84-88) == the columns one by one
    {
        std::vector<std::vector<Fr>> cols(5, std::vector<Fr>(n));
        Fr y(11);
        for (auto& col : cols)
            for (auto& e : col) { e = y; y = y * y + Fr(3); }
        cols[4].assign(n, Fr(0));                       // an all-zero selector: the zero polynomial after the trim
        cols[3] = domain.fft({Fr(9), Fr(4)});           // degree 1: trailing zeros trimmed
        auto polys = poly::interpolate_batch(ctx, cols, domain);
        REQUIRE(polys.size() == 5 && polys[4].is_zero() && polys[3].coeffs.size() == 2 && polys[3].coeffs[1] == Fr(4));
        for (int k = 0; k < 5; ++k) REQUIRE(polys[k].coeffs == poly::interpolate(cols[k], domain).coeffs);
    }
    // domain construction beyond the two-adicity fails like GeneralEvaluationDomain::new(..).unwrap()
    bool threw = false;
    try { poly::Radix2EvaluationDomain too_big(ctx, (1ull << 32) + 1); } catch (const std::runtime_error&) { threw = true; }
    REQUIRE(threw);
    std::puts("interpolate_then_commit ok");
}

int main() {
    Context ctx(0);
    commit(ctx);
    scalar_mul(ctx);
    l0(ctx);
    interpolate_then_commit(ctx);
    std::puts("all ok");
    return 0;
}
