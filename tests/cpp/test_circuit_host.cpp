// The reference's circuit tests (plonk/src/builder/test.rs:26-44) through the C++ mirror, end to end on the GPU:
// build() -> prove() -> verify() for the README circuit and the five-input adder, the wrong witness refused, and a
// 1000-gate chain written as a loop.
#include <cstdio>
#include <cstdlib>

#include "circuit_host.hpp"

using namespace typlonk;

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);      \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

struct Circuit1 : plonk::CircuitDescription<5, Circuit1> {
    template <class V>
    static void run(std::array<V, 5> in) {
        V x = (in[2] + in[3]) + in[4];
        V a = in[0] + in[1];
        a.assert_eq(x);
    }
};
struct Circuit2 : plonk::CircuitDescription<3, Circuit2> {
    template <class V>
    static void run(std::array<V, 3> in) {
        V a = in[0].clone() * in[0];
        V b = in[1].clone() * in[1];
        V c = in[2].clone() * in[2];
        V d = a + b;
        d.assert_eq(c);
    }
};
struct Chain1000 : plonk::CircuitDescription<2, Chain1000> {  // y == x^(2^500) * ... a loop of mixed gates
    template <class V>
    static void run(std::array<V, 2> in) {
        V x = in[0], acc = in[1];
        for (int i = 0; i < 500; ++i) {
            x = x.clone() * x;
            acc = acc + x.clone();
        }
    }
};

template <int LOG>
struct SquaringChainN : plonk::CircuitDescription<1, SquaringChainN<LOG>> {  // 2^LOG - 3 gates: BASELINE configs 3 and 5
    template <class V>
    static void run(std::array<V, 1> in) {
        V x = in[0];
        for (int i = 0; i < (1 << LOG) - 3; ++i) x = x.clone() * x;
    }
};

#include <chrono>
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// `test_circuit_host big [22]`: the 2^20-row (2^22-row) circuit through the same front end -- build, prove twice, verify
template <int LOG>
static int big(const Context& ctx) {
    double t0 = now_ms();
    auto circuit = SquaringChainN<LOG>::build(ctx);
    const double t_build = now_ms() - t0;
    REQUIRE(circuit.rows == (size_t)1 << LOG);
    t0 = now_ms();
    auto proof = circuit.prove({3}, {0});
    const double t_first = now_ms() - t0;
    t0 = now_ms();
    proof = circuit.prove({5}, {0});
    const double t_prove = now_ms() - t0;
    t0 = now_ms();
    REQUIRE(circuit.verify(proof));
    const double t_verify = now_ms() - t0;
    plonk::Proof t = proof;
    t.permutation.zw.y = t.permutation.zw.y + Fr::one();
    REQUIRE(!circuit.verify(t));
    std::printf("big: rows=%zu build_ms=%.0f first_prove_ms=%.0f prove_ms=%.0f verify_ms=%.0f\n", circuit.rows, t_build, t_first,
                t_prove, t_verify);
    std::printf("all ok\n");
    return 0;
}

int main(int argc, char** argv) {
    Context ctx(0);
    if (argc > 1 && std::string(argv[1]) == "big") return (argc > 2 && std::string(argv[2]) == "22") ? big<22>(ctx) : big<20>(ctx);
    {   // circuit2_test
        auto circuit = Circuit2::build(ctx);
        REQUIRE(circuit.rows == 8);
        auto proof = circuit.prove({3, 4, 5}, {0});
        REQUIRE(circuit.verify(proof));
        REQUIRE(proof.r.eval().is_zero() && proof.public_inputs.size() == 8);
        // a proof is bound to its circuit: another SRS does not accept it
        auto other = Circuit2::build(ctx);
        REQUIRE(!other.verify(proof));
        // circuit2_test_bad_inputs (#[should_panic]): 9 + 16 != 36
        bool threw = false;
        try {
            circuit.prove({3, 4, 6}, {0});
        } catch (const std::exception& e) {
            threw = true;
            std::printf("bad inputs: %s\n", e.what());
        }
        REQUIRE(threw);
        // blinding: two proofs of the same statement differ, both verify
        auto again = circuit.prove({3, 4, 5}, {0});
        REQUIRE(circuit.verify(again) && !(again.a_commit == proof.a_commit));
        std::printf("circuit2 ok\n");
    }
    {   // circuit1_test
        auto circuit = Circuit1::build(ctx);
        auto proof = circuit.prove({2, 7, 2, 3, 4}, {0});
        REQUIRE(circuit.verify(proof));
        bool threw = false;
        try {
            circuit.prove({2, 7, 2, 3, 5}, {0});
        } catch (const std::exception&) {
            threw = true;
        }
        REQUIRE(threw);
        std::printf("circuit1 ok\n");
    }
    {   // 1000 gates, s = 2 (the kzg tests' secret), chosen blinders: the same proof twice
        auto circuit = plonk::Circuit<2, Chain1000>::compile_with_secret(ctx, Fr(2));
        REQUIRE(circuit.rows == 1024);
        const Fr bl[3][3] = {{Fr(11), Fr(12), Fr(13)}, {Fr(14), Fr(15), Fr(16)}, {Fr(17), Fr(18), Fr(19)}};
        auto p1 = circuit.prove_with_blinders({3, 1}, {0}, bl);
        auto p2 = circuit.prove_with_blinders({3, 1}, {0}, bl);
        REQUIRE(circuit.verify(p1));
        REQUIRE(p1.a_commit == p2.a_commit && p1.r.p == p2.r.p && p1.evaluation_point == p2.evaluation_point);
        auto p3 = circuit.prove({5, 9}, {0});
        REQUIRE(circuit.verify(p3));
        plonk::Proof t = p3;
        t.c.y = t.c.y + Fr::one();
        REQUIRE(!circuit.verify(t));
        std::printf("chain ok\n");
    }
    std::printf("all ok\n");
    return 0;
}
