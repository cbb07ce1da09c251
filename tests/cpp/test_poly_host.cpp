// CPU-only checks of the host mirror's polynomial / field glue (no GPU, no library calls):
// trailing-zero trimming, degree(), Horner, division by X - z, Fr::from(negative).
#include <cstdio>
#include <cstdlib>

#include "../../typlonk_amd/host/typlonk_host.hpp"
using namespace typlonk;
using poly::DensePolynomial;

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

int main() {
    REQUIRE(Fr(-1) + Fr(1) == Fr(0));
    REQUIRE(Fr(6) * Fr(7) == Fr(42));
    REQUIRE(Fr(5).inverse() * Fr(5) == Fr::one());
    REQUIRE(Fr(3).pow(5) == Fr(243));
    auto p = DensePolynomial::from_coefficients_vec({Fr(1), Fr(2), Fr(3), Fr(0), Fr(0)});
    REQUIRE(p.coeffs.size() == 3 && p.degree() == 2);
    REQUIRE(p.evaluate(Fr(2)) == Fr(17) && p.evaluate(Fr(1)) == Fr(6));
    auto z = DensePolynomial::from_coefficients_vec({Fr(0), Fr(0)});
    REQUIRE(z.is_zero() && z.degree() == 0 && z.evaluate(Fr(9)).is_zero());
    Fr y;
    auto q = p.divide_by_linear(Fr(1), &y);      // (3X^2 + 2X + 1 - 6) / (X - 1) = 3X + 5
    REQUIRE(y == Fr(6) && q.coeffs.size() == 2 && q.coeffs[0] == Fr(5) && q.coeffs[1] == Fr(3));
    // q(x) (x - z) + y == p(x)
    Fr x(123456789), zz(987);
    auto q2 = p.divide_by_linear(zz, &y);
    REQUIRE(q2.evaluate(x) * (x - zz) + y == p.evaluate(x));
    auto c = DensePolynomial::from_coefficients_vec({Fr(7)});
    auto q3 = c.divide_by_linear(Fr(3), &y);     // constant: quotient 0, value 7
    REQUIRE(q3.is_zero() && y == Fr(7));
    REQUIRE((p * Fr(9)).evaluate(Fr(2)) == Fr(153));
    std::puts("all ok");
    return 0;
}
