// C++ mirror of the reference's front end -- the caller side of the prover path: a circuit written once as a generic
// function over a variable type, run with a recording variable to lay out gates and copy constraints and with a
// computing variable to produce the witness.
//
//   plonk::CircuitDescription / Var    /root/reference/plonk/src/description.rs:4-16
//   plonk::BuildVar / ComputeVar       /root/reference/plonk/src/builder.rs:326-441 (tag allocation :339-369, witness
//                                      recording :381-396, ComputeVar::assert_eq is a no-op :435-440)
//   plonk::BuildContext                /root/reference/plonk/src/builder.rs:128-195 (ids, tags, deferred equalities, finish)
//   gate rows, fill()                  /root/reference/plonk/src/builder.rs:47-58, 314-324
//   plonk::PermutationBuilder<C>       /root/reference/permutation/src/lib.rs:26-93  (add_row, add_constrain, build)
//   plonk::Permutation<C>::compile     /root/reference/permutation/src/lib.rs:101-154 (ids k_i w^j, cosets 2, 3, 4)
//   plonk::Circuit<I, DESC>            CompiledCircuit<INPUTS, DESC>: CircuitBuilder::compile (builder.rs:60-110),
//                                      prove (proof.rs:26-57), verify (proof.rs:59-62)
//
// The reference's own tests (plonk/src/builder/test.rs) read the same here:
//
//   struct Circuit2 : plonk::CircuitDescription<3, Circuit2> {
//       template <class V> static void run(std::array<V, 3> in) {
//           auto a = in[0].clone() * in[0]; auto b = in[1].clone() * in[1]; auto c = in[2].clone() * in[2];
//           (a + b).assert_eq(c);
//       }
//   };
//   auto circuit = Circuit2::build(ctx);
//   auto proof = circuit.prove({3, 4, 5}, {0});
//   assert(circuit.verify(proof));
//
// Everything here is O(gates) host glue, as it is in the reference; the transforms and commitments behind build(),
// prove() and verify() go to the GPU through plonk::CompiledCircuit (typlonk_host.hpp).  Nothing here touches the oracle.
#pragma once
#include <array>
#include <memory>
#include <random>
#include <unordered_map>
#include <utility>

#include "../../typlonk_amd/csrc/transcript.hpp"
#include "../../typlonk_amd/host/typlonk_host.hpp"

namespace typlonk {
namespace plonk {

// Fr::rand(&mut rand::thread_rng()) (proof.rs:42-46, srs.rs:36-40): blinding rows and the SRS secret need a
// cryptographic generator -- the library's ChaCha12 (csrc/transcript.hpp, the same StdRng the transcript uses), keyed per
// thread with 256 bits from the operating system
inline Fr random_fr() {
    static thread_local ty::StdRng rng = [] {
        std::random_device os;
        uint32_t key[8];
        for (auto& k : key) k = os();
        return ty::StdRng::from_key(key);
    }();
    Fr f;
    ty::fr_rand(rng, f.limbs());
    return f;
}

// ---- permutation argument: cells, copy constraints, sigma ------------------------------------------------------------
struct Tag {
    size_t i, j;  // column, row
    bool operator==(const Tag& o) const { return i == o.i && j == o.j; }
};

template <size_t C>
struct CompiledPermutation {
    std::array<std::vector<std::pair<Fr, Fr>>, C> cols;  // per cell: (its own label k_i w^j, the label sigma sends it to)
    std::array<Fr, C> cosets;
    size_t rows = 0;
    std::vector<Fr> sigma_column(size_t i) const {
        std::vector<Fr> s(rows);
        for (size_t j = 0; j < rows; ++j) s[j] = cols[i][j].second;
        return s;
    }
};

template <size_t C>
struct Permutation {
    std::vector<size_t> perm;  // flat cell index j + i * rows -> next cell of its cycle

    // the first C field elements k = 1, 2, ... that are not n-th roots of unity (k = 1 always is, so 2, 3, 4 for C = 3)
    static std::array<Fr, C> cosets(size_t rows) {
        std::array<Fr, C> out;
        Fr k = Fr::one();
        for (auto& c : out) {
            while ((k.pow(rows) - Fr::one()).is_zero()) k += Fr::one();
            c = k;
            k += Fr::one();
        }
        return out;
    }
    CompiledPermutation<C> compile() const {
        if (perm.size() % C != 0 || perm.empty()) throw std::runtime_error("permutation length is not a multiple of the column count");
        CompiledPermutation<C> out;
        out.rows = perm.size() / C;
        out.cosets = cosets(out.rows);
        uint32_t log_rows = 0;
        while (((size_t)1 << log_rows) < out.rows) ++log_rows;
        std::vector<Fr> roots(out.rows);
        const Fr w = poly::two_adic_root(log_rows);
        Fr acc = Fr::one();
        for (auto& r : roots) {
            r = acc;
            acc *= w;
        }
        for (size_t i = 0; i < C; ++i) {
            out.cols[i].resize(out.rows);
            for (size_t j = 0; j < out.rows; ++j) {
                const size_t to = perm[j + i * out.rows];
                out.cols[i][j] = {out.cosets[i] * roots[j], out.cosets[to / out.rows] * roots[to % out.rows]};
            }
        }
        return out;
    }
};

template <size_t C>
class PermutationBuilder {
   public:
    void add_row() { ++rows_; }
    static PermutationBuilder with_rows(size_t rows) {
        PermutationBuilder b;
        b.rows_ = rows;
        return b;
    }
    size_t rows() const { return rows_; }
    // false where the reference returns Err(()): a tag outside the table.  (The reference lets column index C through,
    // `i <= &C`, lib.rs:44, and then indexes out of bounds in build(); here it is refused with the others.)
    bool add_constrain(const Tag& left, const Tag& right) {
        if (!inside(left) || !inside(right)) return false;
        constrains_.push_back({left, right});
        return true;
    }
    void add_constrains(const std::vector<std::pair<Tag, Tag>>& cs) {
        for (const auto& c : cs)
            if (!add_constrain(c.first, c.second)) throw std::runtime_error("copy constraint outside the table");  // unwrap()
    }
    // Merge the cycles of every constrained pair (the permutation starts as the identity: every cell its own cycle).
    // Two cells of different cycles are joined by exchanging their successors; the smaller cycle takes the other's
    // label so that relabelling costs O(n log n) overall.  The reference walks its constraints in HashMap order
    // (lib.rs:68), so the ORDER inside a cycle differs run to run there too; the partition is what is defined.
    Permutation<C> build(size_t size) {
        const size_t len = size * C;
        std::vector<size_t> next(len), label(len), members(len, 1);
        for (size_t k = 0; k < len; ++k) next[k] = label[k] = k;
        for (const auto& c : constrains_) {
            size_t keep = c.first.j + c.first.i * size, fold = c.second.j + c.second.i * size;
            if (keep >= len || fold >= len) throw std::runtime_error("copy constraint outside the padded table");
            if (label[keep] == label[fold]) continue;  // already in one cycle
            if (members[label[keep]] < members[label[fold]]) std::swap(keep, fold);
            const size_t into = label[keep];
            members[into] += members[label[fold]];
            for (size_t k = fold; label[k] != into; k = next[k]) label[k] = into;
            std::swap(next[keep], next[fold]);
        }
        constrains_.clear();
        return Permutation<C>{std::move(next)};
    }

   private:
    bool inside(const Tag& t) const { return t.i < C && t.j < rows_; }
    std::vector<std::pair<Tag, Tag>> constrains_;
    size_t rows_ = 0;
};

// ---- gates -------------------------------------------------------------------------------------------------------
enum class Gate { Mul, Add, Dummy };
// one row of (q_l, q_r, q_o, q_m, q_c): the gate equation is q_l a + q_r b + q_m a b + q_c - q_o c = 0
inline std::array<Fr, 5> gate_row(Gate g) {
    const Fr o = Fr::one(), z = Fr::zero();
    switch (g) {
        case Gate::Mul: return {z, z, o, o, z};
        case Gate::Add: return {o, o, o, z, z};
        default: return {z, z, z, z, z};
    }
}

// ---- the recording run ------------------------------------------------------------------------------------------------
// State shared by all BuildVars of one compile(): the gate list, the copy constraints, which variable sits in which
// cell.  A variable gets its cell (tag) the first time it enters a gate; every later use occupies a new cell tied to
// the first one by a copy constraint.  assert_eq on a variable that has no cell yet is kept until finish().
class BuildContext {
   public:
    size_t new_id() { return next_id_++; }
    size_t add_gate(Gate g) {
        gates_.push_back(g);
        permutation_.add_row();
        return gates_.size() - 1;
    }
    void place(size_t id, const Tag& t) { cell_[id] = t; }
    const Tag* cell_of(size_t id) const {
        auto it = cell_.find(id);
        return it == cell_.end() ? nullptr : &it->second;
    }
    void add_eq(size_t left, size_t right) {
        const Tag *a = cell_of(left), *b = cell_of(right);
        if (a && b) {
            if (!permutation_.add_constrain(*a, *b)) throw std::runtime_error("copy constraint outside the table");  // unwrap()
        } else {
            pending_.push_back({left, right});
        }
    }
    // flush the deferred equalities (one that still has no cell is an error -- the reference asserts), pad with dummy
    // gates to the first power of two >= gates + 3, counting from 2 (room for the three blinding rows)
    void finish(std::vector<Gate>* gates, PermutationBuilder<3>* permutation) {
        std::vector<std::pair<size_t, size_t>> waiting;
        waiting.swap(pending_);
        for (const auto& e : waiting) add_eq(e.first, e.second);
        if (!pending_.empty()) throw std::runtime_error("assert_eq on a variable that never enters a gate");
        size_t size = 2;
        while (size < gates_.size() + 3) size *= 2;
        gates_.resize(size, Gate::Dummy);
        *gates = std::move(gates_);
        *permutation = std::move(permutation_);
    }

   private:
    std::vector<Gate> gates_;
    PermutationBuilder<3> permutation_;
    size_t next_id_ = 0;
    std::vector<std::pair<size_t, size_t>> pending_;
    std::unordered_map<size_t, Tag> cell_;
};

class BuildVar {
   public:
    static BuildVar input(const std::shared_ptr<BuildContext>& cx) { return BuildVar(cx, cx->new_id()); }
    BuildVar clone() const { return *this; }
    void assert_eq(const BuildVar& other) const { cx_->add_eq(id_, other.id_); }
    friend BuildVar operator+(const BuildVar& l, const BuildVar& r) { return l.gate(r, Gate::Add); }
    friend BuildVar operator*(const BuildVar& l, const BuildVar& r) { return l.gate(r, Gate::Mul); }

   private:
    BuildVar(std::shared_ptr<BuildContext> cx, size_t id) : cx_(std::move(cx)), id_(id) {}
    BuildVar gate(const BuildVar& rhs, Gate g) const {
        const size_t j = cx_->add_gate(g);
        const size_t out = cx_->new_id();
        cx_->place(out, Tag{2, j});
        const size_t operand[2] = {id_, rhs.id_};
        for (size_t col = 0; col < 2; ++col) {  // left, then right: `x * x` therefore ties (0, j) to (1, j)
            if (cx_->cell_of(operand[col])) {
                const size_t copy = cx_->new_id();
                cx_->place(copy, Tag{col, j});
                cx_->add_eq(operand[col], copy);
            } else {
                cx_->place(operand[col], Tag{col, j});
            }
        }
        return BuildVar(cx_, out);
    }
    std::shared_ptr<BuildContext> cx_;
    size_t id_;
};

// ---- the computing run -------------------------------------------------------------------------------------------------
struct Advice {
    std::vector<Fr> col[3];  // a, b, c in gate order
};
class ComputeVar {
   public:
    ComputeVar(const Fr& value, std::shared_ptr<Advice> advice) : value_(value), advice_(std::move(advice)) {}
    ComputeVar clone() const { return *this; }
    const Fr& value() const { return value_; }
    // deliberately not a check: a wrong witness must reach the prover (which refuses it) and the verifier
    void assert_eq(const ComputeVar&) const {}
    friend ComputeVar operator+(const ComputeVar& l, const ComputeVar& r) { return l.record(r, l.value_ + r.value_); }
    friend ComputeVar operator*(const ComputeVar& l, const ComputeVar& r) { return l.record(r, l.value_ * r.value_); }

   private:
    ComputeVar record(const ComputeVar& rhs, const Fr& out) const {
        advice_->col[0].push_back(value_);
        advice_->col[1].push_back(rhs.value_);
        advice_->col[2].push_back(out);
        return ComputeVar(out, advice_);
    }
    Fr value_;
    std::shared_ptr<Advice> advice_;
};

namespace detail {
template <size_t... K>
std::array<BuildVar, sizeof...(K)> inputs_impl(const std::shared_ptr<BuildContext>& cx, std::index_sequence<K...>) {
    return {{((void)K, BuildVar::input(cx))...}};
}
template <size_t... K>
std::array<ComputeVar, sizeof...(K)> compute_inputs_impl(const std::array<Fr, sizeof...(K)>& v, const std::shared_ptr<Advice>& a,
                                                           std::index_sequence<K...>) {
    return {{ComputeVar(v[K], a)...}};
}
}  // namespace detail
template <size_t INPUTS>
std::array<BuildVar, INPUTS> make_inputs(const std::shared_ptr<BuildContext>& cx) {
    return detail::inputs_impl(cx, std::make_index_sequence<INPUTS>{});
}

// ---- what compile() produces before anything touches the GPU ----------------------------------------------------------
struct CircuitTables {
    size_t rows = 0;
    uint32_t log_rows = 0;
    std::vector<Gate> gates;
    std::vector<Fr> selector_evals[5];  // q_l, q_r, q_o, q_m, q_c over the domain
    Permutation<3> permutation;
    CompiledPermutation<3> copy_constrains;
};

template <size_t INPUTS, class DESC>
CircuitTables compile_tables() {
    auto cx = std::make_shared<BuildContext>();
    std::array<BuildVar, INPUTS> inputs = make_inputs<INPUTS>(cx);
    DESC::template run<BuildVar>(inputs);
    CircuitTables t;
    PermutationBuilder<3> pb;
    cx->finish(&t.gates, &pb);
    t.rows = t.gates.size();
    while (((size_t)1 << t.log_rows) < t.rows) ++t.log_rows;
    for (auto& col : t.selector_evals) col.reserve(t.rows);
    for (Gate g : t.gates) {
        const auto row = gate_row(g);
        for (int k = 0; k < 5; ++k) t.selector_evals[k].push_back(row[k]);
    }
    t.permutation = pb.build(t.rows);
    t.copy_constrains = t.permutation.compile();
    return t;
}

// ---- CompiledCircuit<INPUTS, DESC> -------------------------------------------------------------------------------------
template <size_t INPUTS, class DESC>
class Circuit {
   public:
    // CircuitBuilder::compile: run the description once with BuildVar, lay out the tables, draw an SRS for the padded
    // size (Srs::random(domain.size()), builder.rs:71), interpolate and commit the fixed polynomials
    static Circuit compile(const Context& ctx) { return compile_with_secret(ctx, random_fr()); }
    // the same over [s^i]G for a given s (tests; the reference's kzg tests use s = 2)
    static Circuit compile_with_secret(const Context& ctx, const Fr& s) {
        Circuit c;
        c.tables_ = compile_tables<INPUTS, DESC>();
        c.rows = c.tables_.rows;
        c.srs_.reset(new kzg::Srs(kzg::Srs::from_secret(ctx, s, c.rows)));
        if (c.rows >= ((size_t)1 << 14)) c.srs_->precompute();   // setup-time tables: every commitment of every proof is faster
        std::vector<Fr> sigma[3];
        Fr cosets[3];
        for (int i = 0; i < 3; ++i) {
            sigma[i] = c.tables_.copy_constrains.sigma_column(i);
            cosets[i] = c.tables_.copy_constrains.cosets[i];
        }
        c.compiled_.reset(new CompiledCircuit(*c.srs_, c.tables_.log_rows, c.tables_.selector_evals, sigma, cosets));
        return c;
    }
    // CompiledCircuit::prove: run the description with ComputeVar on the inputs, pad the three witness columns to
    // rows - 3, append three random blinding rows each, pad the public inputs with zeros to `rows`, prove
    Proof prove(const std::array<Fr, INPUTS>& inputs, const std::vector<Fr>& public_inputs) const {
        std::vector<Fr> advice[3];
        witness(inputs, advice);
        for (auto& col : advice)
            for (int k = 0; k < 3; ++k) col.push_back(random_fr());
        return prove_columns(advice, public_inputs);
    }
    // the same with chosen blinding rows (deterministic proofs for tests)
    Proof prove_with_blinders(const std::array<Fr, INPUTS>& inputs, const std::vector<Fr>& public_inputs,
                              const Fr (&blinders)[3][3]) const {
        std::vector<Fr> advice[3];
        witness(inputs, advice);
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) advice[i].push_back(blinders[i][k]);
        return prove_columns(advice, public_inputs);
    }
    bool verify(const Proof& proof) const { return compiled_->verify(proof, proof.public_inputs); }

    size_t rows = 0;
    const CircuitTables& tables() const { return tables_; }
    const CompiledCircuit& compiled() const { return *compiled_; }
    const kzg::Srs& srs() const { return *srs_; }

   private:
    Circuit() = default;
    void witness(const std::array<Fr, INPUTS>& inputs, std::vector<Fr> (&advice)[3]) const {
        auto rec = std::make_shared<Advice>();
        for (auto& col : rec->col) col.reserve(rows);
        DESC::template run<ComputeVar>(detail::compute_inputs_impl(inputs, rec, std::make_index_sequence<INPUTS>{}));
        for (int i = 0; i < 3; ++i) {
            advice[i] = std::move(rec->col[i]);
            if (advice[i].size() > rows - 3) throw std::runtime_error("the computing run made more gates than the recording run");
            advice[i].resize(rows - 3);
        }
    }
    Proof prove_columns(const std::vector<Fr> (&advice)[3], const std::vector<Fr>& public_inputs) const {
        if (public_inputs.size() > rows) throw std::runtime_error("more public inputs than rows");
        std::vector<Fr> pi(public_inputs);
        pi.resize(rows);
        bool any = false;
        for (const Fr& x : pi) any = any || !x.is_zero();
        Proof p = compiled_->prove(advice, any ? pi : std::vector<Fr>{});
        p.public_inputs = std::move(pi);
        return p;
    }
    CircuitTables tables_;
    std::unique_ptr<kzg::Srs> srs_;
    std::unique_ptr<CompiledCircuit> compiled_;
};

// CircuitDescription<INPUTS>: derive as `struct My : CircuitDescription<N, My>` and give it
// `template <class V> static void run(std::array<V, N> inputs)`; V offers +, *, clone() and assert_eq().
template <size_t INPUTS, class DESC>
struct CircuitDescription {
    static Circuit<INPUTS, DESC> build(const Context& ctx) { return Circuit<INPUTS, DESC>::compile(ctx); }
};

}  // namespace plonk
}  // namespace typlonk
