// The front end of the C++ mirror (tests/cpp/circuit_host.hpp, test harness) without a GPU: the recording run of the
// reference's two test circuits (plonk/src/builder/test.rs:3-24), the permutation builder, the witness of the computing
// run.  Prints the tables of the README circuit for tests/test_host_mirror.py to compare with the tables
// oracle/plonk_oracle.py lays out by hand.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <vector>

#include "circuit_host.hpp"

using namespace typlonk;
using plonk::Tag;

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
struct Dangling : plonk::CircuitDescription<2, Dangling> {  // an equality on inputs that never enter a gate
    template <class V>
    static void run(std::array<V, 2> in) {
        in[0].assert_eq(in[1]);
    }
};
template <int N>
struct Chain : plonk::CircuitDescription<1, Chain<N>> {  // x -> x^2 -> x^4 ... N squarings
    template <class V>
    static void run(std::array<V, 1> in) {
        V x = in[0];
        for (int i = 0; i < N; ++i) x = x.clone() * x;
    }
};

static void print_fr(const Fr& f);
// A circuit description as data (the same generator lives in the Python oracle of the front end,
// random_program): x <- x * 6364136223846793005 + 1442695040888963407, take x >> 33; per step kind = r % 8
// (0..2 add, 3..5 mul, 6..7 assert_eq), operands = two draws modulo the number of variables so far.
struct Op {
    int kind;  // 0 add, 1 mul, 2 eq
    size_t a, b;
};
static std::vector<Op> g_program;
static void make_program(uint64_t seed, size_t n_inputs, size_t n_ops) {
    g_program.clear();
    uint64_t x = seed;
    auto draw = [&]() {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        return x >> 33;
    };
    size_t nvars = n_inputs;
    for (size_t i = 0; i < n_ops; ++i) {
        const uint64_t kind = draw() % 8;
        const size_t a = draw() % nvars, b = draw() % nvars;
        if (kind <= 5) {
            g_program.push_back({kind <= 2 ? 0 : 1, a, b});
            ++nvars;
        } else {
            g_program.push_back({2, a, b});
        }
    }
}
struct Programmed : plonk::CircuitDescription<3, Programmed> {
    template <class V>
    static void run(std::array<V, 3> in) {
        std::vector<V> v(in.begin(), in.end());
        for (const Op& op : g_program) {
            if (op.kind == 0) v.push_back(v[op.a] + v[op.b]);
            else if (op.kind == 1) v.push_back(v[op.a] * v[op.b]);
            else v[op.a].assert_eq(v[op.b]);
        }
    }
};

// the cycles of a permutation as a set of sorted cell sets
static std::set<std::set<size_t>> cycles(const std::vector<size_t>& perm) {
    std::set<std::set<size_t>> out;
    std::vector<bool> seen(perm.size());
    for (size_t s = 0; s < perm.size(); ++s) {
        if (seen[s]) continue;
        std::set<size_t> cyc;
        for (size_t k = s; !seen[k]; k = perm[k]) {
            seen[k] = true;
            cyc.insert(k);
        }
        out.insert(cyc);
    }
    return out;
}
static void print_fr(const Fr& f) {
    const ty::Fr c = ty::fe_from_mont(f.v);
    for (int k = 7; k >= 0; --k) std::printf("%08x", c.v[k]);
}

// `test_circuit_tables_host random <seed> <ops>`: the tables and the witness (inputs 3, 4, 5) of a generated circuit
static int random_mode(uint64_t seed, size_t ops) {
    make_program(seed, 3, ops);
    plonk::CircuitTables t;
    try {
        t = plonk::compile_tables<3, Programmed>();
    } catch (const std::exception& e) {
        std::printf("dangling\n");
        return 0;
    }
    std::printf("rows=%zu\ngates=", t.rows);
    for (plonk::Gate g : t.gates) std::printf("%c", g == plonk::Gate::Mul ? 'M' : (g == plonk::Gate::Add ? 'A' : 'D'));
    std::printf("\nperm=");
    for (size_t k = 0; k < t.permutation.perm.size(); ++k) std::printf("%zu%s", t.permutation.perm[k], k + 1 < t.permutation.perm.size() ? "," : "\n");
    auto rec = std::make_shared<plonk::Advice>();
    std::array<plonk::ComputeVar, 3> in = {plonk::ComputeVar(Fr(3), rec), plonk::ComputeVar(Fr(4), rec), plonk::ComputeVar(Fr(5), rec)};
    Programmed::run<plonk::ComputeVar>(in);
    for (int c = 0; c < 3; ++c) {
        std::printf("w%d=", c);
        for (size_t j = 0; j < rec->col[c].size(); ++j) {
            print_fr(rec->col[c][j]);
            std::printf(j + 1 < rec->col[c].size() ? "," : "");
        }
        std::printf("\n");
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 4 && std::string(argv[1]) == "random") return random_mode(std::strtoull(argv[2], nullptr, 10), std::strtoull(argv[3], nullptr, 10));
    // ---- permutation builder on its own ----
    {
        auto pb = plonk::PermutationBuilder<3>::with_rows(4);
        REQUIRE(!pb.add_constrain(Tag{3, 0}, Tag{0, 0}));   // column outside
        REQUIRE(!pb.add_constrain(Tag{0, 4}, Tag{0, 0}));   // row outside
        REQUIRE(pb.add_constrain(Tag{0, 0}, Tag{1, 1}));
        REQUIRE(pb.add_constrain(Tag{1, 1}, Tag{2, 3}));
        REQUIRE(pb.add_constrain(Tag{2, 3}, Tag{0, 0}));    // closes nothing new
        REQUIRE(pb.add_constrain(Tag{0, 2}, Tag{0, 3}));
        const auto p = pb.build(4);
        REQUIRE(p.perm.size() == 12);
        const auto cyc = cycles(p.perm);
        REQUIRE(cyc.count({0, 5, 11}) == 1 && cyc.count({2, 3}) == 1 && cyc.size() == 2 + 7);
        const auto cp = p.compile();
        REQUIRE(cp.cosets[0] == Fr(2) && cp.cosets[1] == Fr(3) && cp.cosets[2] == Fr(4));
        const Fr w = poly::two_adic_root(2);
        REQUIRE(cp.cols[1][1].first == Fr(3) * w);
        // sigma is a bijection on the labels, and fixed cells map to themselves
        REQUIRE(cp.cols[1][0].second == cp.cols[1][0].first);
        // product over all cells of (id / sigma) is 1: the same multiset of labels on both sides
        Fr num = Fr::one(), den = Fr::one();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                num *= cp.cols[i][j].first;
                den *= cp.cols[i][j].second;
            }
        REQUIRE(num == den);
        bool threw = false;
        try {
            pb.add_constrains({{Tag{0, 0}, Tag{0, 9}}});
        } catch (const std::exception&) {
            threw = true;
        }
        REQUIRE(threw);
    }
    std::printf("permutation ok\n");

    // ---- the recording run ----
    {
        const plonk::CircuitTables t = plonk::compile_tables<3, Circuit2>();
        REQUIRE(t.rows == 8 && t.log_rows == 3);   // 4 gates + 3 -> 8
        using plonk::Gate;
        const Gate want[8] = {Gate::Mul, Gate::Mul, Gate::Mul, Gate::Add, Gate::Dummy, Gate::Dummy, Gate::Dummy, Gate::Dummy};
        for (int j = 0; j < 8; ++j) REQUIRE(t.gates[j] == want[j]);
        std::printf("circuit2 rows=%zu\n", t.rows);
        for (int k = 0; k < 5; ++k) {
            std::printf("q%d=", k);
            for (size_t j = 0; j < t.rows; ++j) {
                print_fr(t.selector_evals[k][j]);
                std::printf(j + 1 < t.rows ? "," : "\n");
            }
        }
        std::printf("perm=");
        for (size_t k = 0; k < t.permutation.perm.size(); ++k) std::printf("%zu%s", t.permutation.perm[k], k + 1 < t.permutation.perm.size() ? "," : "\n");
        for (int i = 0; i < 3; ++i) {
            std::printf("sigma%d=", i);
            for (size_t j = 0; j < t.rows; ++j) {
                print_fr(t.copy_constrains.cols[i][j].second);
                std::printf(j + 1 < t.rows ? "," : "\n");
            }
        }
    }
    {
        const plonk::CircuitTables t = plonk::compile_tables<5, Circuit1>();
        REQUIRE(t.rows == 8);                       // 3 gates + 3 -> 8
        // (c + d) -> row 0, (.. + e) -> row 1 with its left operand copied from (2, 0); a + b -> row 2; the two results tied
        const auto cyc = cycles(t.permutation.perm);
        REQUIRE(cyc.count({0 * 8 + 1, 2 * 8 + 0}) == 1);
        REQUIRE(cyc.count({2 * 8 + 1, 2 * 8 + 2}) == 1);
        REQUIRE(cyc.size() == 24 - 2);
    }
    {
        const plonk::CircuitTables t = plonk::compile_tables<1, Chain<1000>>();
        REQUIRE(t.rows == 1024);                    // 1000 + 3 -> 1024
        const auto cyc = cycles(t.permutation.perm);
        // row j: (0, j) ~ (1, j); its output (2, j) is both operands of row j + 1
        REQUIRE(cyc.count({0 * 1024 + 5, 1 * 1024 + 5, 2 * 1024 + 4}) == 1);
        const plonk::CircuitTables t2 = plonk::compile_tables<1, Chain<1022>>();
        REQUIRE(t2.rows == 2048);                   // 1022 + 3 = 1025 -> 2048
        const plonk::CircuitTables t3 = plonk::compile_tables<1, Chain<1021>>();
        REQUIRE(t3.rows == 1024);
    }
    {
        bool threw = false;
        try {
            plonk::compile_tables<2, Dangling>();
        } catch (const std::exception&) {
            threw = true;
        }
        REQUIRE(threw);
    }
    std::printf("tables ok\n");

    // ---- the computing run ----
    {
        auto rec = std::make_shared<plonk::Advice>();
        std::array<plonk::ComputeVar, 3> in = {plonk::ComputeVar(Fr(3), rec), plonk::ComputeVar(Fr(4), rec), plonk::ComputeVar(Fr(6), rec)};
        Circuit2::run<plonk::ComputeVar>(in);      // 9 + 16 != 36: assert_eq does not stop the run
        REQUIRE(rec->col[0].size() == 4);
        REQUIRE(rec->col[0][3] == Fr(9) && rec->col[1][3] == Fr(16) && rec->col[2][3] == Fr(25) && rec->col[2][2] == Fr(36));
    }
    {   // blinding values: distinct draws, every one a canonical residue (round trip through the integer form)
        const Fr x = plonk::random_fr(), y = plonk::random_fr();
        REQUIRE(x != y);
        REQUIRE(Fr(ty::fe_to_mont(ty::fe_from_mont(x.v))) == x);
    }
    std::printf("witness ok\n");
    std::printf("all ok\n");
    return 0;
}
