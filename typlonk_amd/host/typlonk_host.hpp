// C++ host-side mirror of the reference's interface for the MSM + NTT path, written above the C ABI
// (include/typlonk.h).  The reference is Rust; this image has no Rust toolchain, so the host side is
// C++ with the reference's names, argument meaning and error behaviour (panics -> exceptions):
//
//   kzg::Srs                    /root/reference/kzg/src/srs.rs:8-51      (from_secret, g1_ref)
//   kzg::KzgScheme              /root/reference/kzg/src/lib.rs:15-86     (commit, open, identity)
//   kzg::KzgCommitment/Opening  /root/reference/kzg/src/lib.rs:16-31, 110-158
//   poly::DensePolynomial       ark-poly 0.3.0 as used by the reference  (from_coefficients_vec trims
//                               trailing zeros; degree(); evaluate = Horner; division by X - z)
//   poly::Radix2EvaluationDomain / Evaluations::interpolate / evaluate_over_domain
//                               call sites /root/reference/plonk/src/proof.rs:50,106,115 ;
//                               plonk/src/builder.rs:70,85 ; plonk/src/utils.rs:150-159 (l0_poly)
//   kzg::KzgScheme::verify      /root/reference/kzg/src/lib.rs:66-81 (host pairing, pairing_host.hpp); Srs::g2 srs.rs:26-34
//   plonk::CompiledCircuit      /root/reference/plonk/src/lib.rs:19-35 (srs, domain, gate_constrains, copy_constrains)
//   plonk::CompiledCircuit::verify
//                               /root/reference/plonk/src/proof.rs:195-281, 441-503 (verify, verify_challenges,
//                               verify_openings, linearisation_commitment), plonk/src/utils.rs:96-109
//   plonk::CompiledCircuit::prove / plonk::Proof
//                               /root/reference/plonk/src/proof.rs:26-57, 65-95, 96-194 -> typlonk_prove
//
// Every group / transform operation goes to the GPU through the C ABI; the O(n) glue the reference
// does on the CPU (Horner, synthetic division) is done on the CPU here too, with the library's own
// Fr arithmetic (csrc/ff.hpp).  Nothing here touches the oracle.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/typlonk.h"
#include "../csrc/ff.hpp"
#include "pairing_host.hpp"

namespace typlonk {

inline void check(int rc, typlonk_ctx* ctx = nullptr) {
    if (rc < 0)
        throw std::runtime_error(std::string(typlonk_strerror(rc)) + (ctx ? std::string(": ") + typlonk_last_error(ctx) : ""));
}

// ---- Fr ------------------------------------------------------------------------------------------
// Same bytes as ark_bls12_381::Fr: 4 x u64 little-endian Montgomery limbs.
struct Fr {
    ty::Fr v;
    Fr() : v(ty::Fr::zero()) {}
    explicit Fr(const ty::Fr& x) : v(x) {}
    Fr(int64_t x) {  // Fr::from(i32/i64): negative values wrap to r - |x|
        ty::Fr c = ty::Fr::zero();
        const uint64_t a = x < 0 ? (uint64_t)(-x) : (uint64_t)x;
        c.v[0] = (uint32_t)a;
        c.v[1] = (uint32_t)(a >> 32);
        v = ty::fe_to_mont(c);
        if (x < 0) v = ty::fe_neg(v);
    }
    static Fr zero() { return Fr(); }
    static Fr one() { return Fr(ty::Fr::one()); }
    bool is_zero() const { return v.is_zero(); }
    Fr operator+(const Fr& o) const { return Fr(ty::fe_add(v, o.v)); }
    Fr operator-(const Fr& o) const { return Fr(ty::fe_sub(v, o.v)); }
    Fr operator*(const Fr& o) const { return Fr(ty::fe_mul(v, o.v)); }
    Fr operator-() const { return Fr(ty::fe_neg(v)); }
    Fr& operator+=(const Fr& o) { return *this = *this + o; }
    Fr& operator-=(const Fr& o) { return *this = *this - o; }
    Fr& operator*=(const Fr& o) { return *this = *this * o; }
    bool operator==(const Fr& o) const { return v == o.v; }
    bool operator!=(const Fr& o) const { return !(v == o.v); }
    Fr inverse() const { return Fr(ty::fe_inv(v)); }
    Fr pow(uint64_t e) const {
        uint32_t w[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
        return Fr(ty::fe_pow(v, w, 2));
    }
    const uint64_t* limbs() const { return reinterpret_cast<const uint64_t*>(v.v); }
    uint64_t* limbs() { return reinterpret_cast<uint64_t*>(v.v); }
};
static_assert(sizeof(Fr) == 32, "Fr must be 4 x u64");

// ---- context (one per process / GPU) --------------------------------------------------------------
class Context {
   public:
    explicit Context(int device = 0) { check(typlonk_init(&ctx_, device)); }
    ~Context() { typlonk_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    typlonk_ctx* raw() const { return ctx_; }

   private:
    typlonk_ctx* ctx_ = nullptr;
};

namespace poly {

// ark-poly 0.3.0 DensePolynomial<Fr>
struct DensePolynomial {
    std::vector<Fr> coeffs;

    static DensePolynomial from_coefficients_vec(std::vector<Fr> c) {
        while (!c.empty() && c.back().is_zero()) c.pop_back();  // ark-poly trims trailing zeros
        DensePolynomial p;
        p.coeffs = std::move(c);
        return p;
    }
    static DensePolynomial from_coefficients_slice(const std::vector<Fr>& c) { return from_coefficients_vec(c); }
    bool is_zero() const { return coeffs.empty(); }
    size_t degree() const { return coeffs.empty() ? 0 : coeffs.size() - 1; }
    Fr evaluate(const Fr& x) const {  // Horner
        Fr acc;
        for (size_t i = coeffs.size(); i-- > 0;) acc = acc * x + coeffs[i];
        return acc;
    }
    // (self - self(z)) / (X - z): the division `&polynomial / &root` of kzg/src/lib.rs:58-61.
    // Returns the quotient; *y receives self(z).
    DensePolynomial divide_by_linear(const Fr& z, Fr* y) const {
        std::vector<Fr> q(coeffs.size() > 0 ? coeffs.size() - 1 : 0);
        Fr carry;
        for (size_t i = coeffs.size(); i-- > 1;) {
            carry = carry * z + coeffs[i];
            q[i - 1] = carry;
        }
        if (!coeffs.empty()) carry = carry * z + coeffs[0];
        if (y) *y = carry;
        return from_coefficients_vec(std::move(q));
    }
    DensePolynomial operator*(const Fr& k) const {
        std::vector<Fr> c(coeffs);
        for (auto& x : c) x *= k;
        return from_coefficients_vec(std::move(c));
    }
};

// Fr::get_root_of_unity(2^log_size): TWO_ADIC_ROOT_OF_UNITY = 7^((r-1)/2^32), squared (32 - log_size) times
inline Fr two_adic_root(uint32_t log_size) {
    if (log_size > 32) check(TYPLONK_ERR_DOMAIN);
    ty::Fr c;
    const uint32_t root[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u, 0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
    for (int i = 0; i < 8; ++i) c.v[i] = root[i];
    Fr w(ty::fe_to_mont(c));
    for (uint32_t i = log_size; i < 32; ++i) w = w * w;
    return w;
}

// ark-poly 0.3.0 Radix2EvaluationDomain<Fr> (what GeneralEvaluationDomain::new returns for BLS12-381 Fr)
class Radix2EvaluationDomain {
   public:
    // GeneralEvaluationDomain::new(num_coeffs): size = next power of two; None (here: exception, the
    // reference unwrap()s, plonk/src/builder.rs:70) beyond the two-adicity 2^32.
    Radix2EvaluationDomain(const Context& ctx, uint64_t num_coeffs) : ctx_(&ctx) {
        log_size_ = 0;
        while ((1ull << log_size_) < num_coeffs) {
            if (++log_size_ > 32) check(TYPLONK_ERR_DOMAIN);
        }
        size_ = 1ull << log_size_;
        const Fr w = two_adic_root(log_size_);
        group_gen = w;
        group_gen_inv = w.inverse();
        size_inv = Fr((int64_t)size_).inverse();
    }
    uint64_t size() const { return size_; }
    uint32_t log_size_of_group() const { return log_size_; }
    Fr element(uint64_t i) const { return group_gen.pow(i); }
    Fr evaluate_vanishing_polynomial(const Fr& tau) const { return tau.pow(size_) - Fr::one(); }

    // natural order in / out; input shorter than size() is zero-padded (ark-poly fft semantics)
    std::vector<Fr> fft(const std::vector<Fr>& coeffs) const { return transform(coeffs, 0, nullptr); }
    std::vector<Fr> ifft(const std::vector<Fr>& evals) const { return transform(evals, 1, nullptr); }
    std::vector<Fr> coset_fft(const std::vector<Fr>& coeffs, const Fr& g) const { return transform(coeffs, 0, &g); }
    std::vector<Fr> coset_ifft(const std::vector<Fr>& evals, const Fr& g) const { return transform(evals, 1, &g); }

    Fr group_gen, group_gen_inv, size_inv;

   private:
    std::vector<Fr> transform(const std::vector<Fr>& in, int inverse, const Fr* coset) const {
        if (in.size() > size_) throw std::runtime_error("more coefficients than the domain size");
        std::vector<Fr> v(in);
        v.resize(size_);
        check(typlonk_ntt_fr(ctx_->raw(), v[0].limbs(), log_size_, inverse, coset ? coset->limbs() : nullptr), ctx_->raw());
        return v;
    }
    const Context* ctx_;
    uint64_t size_;
    uint32_t log_size_;
};

// Evaluations::from_vec_and_domain(v, D).interpolate()  /  DensePolynomial::evaluate_over_domain(D)
inline DensePolynomial interpolate(const std::vector<Fr>& evals, const Radix2EvaluationDomain& d) {
    return DensePolynomial::from_coefficients_vec(d.ifft(evals));
}
inline std::vector<Fr> evaluate_over_domain(const DensePolynomial& p, const Radix2EvaluationDomain& d) {
    return d.fft(p.coeffs);
}
// A GROUP of interpolations -- the reference maps `interpolate` over the three wire columns (plonk/src/proof.rs:50), the five
// selector columns (plonk/src/builder.rs:84-88), the three sigma columns (proof.rs:334-338): one upload, ONE
// typlonk_ntt_fr_batch_devptr (every pass one launch for all columns), one download; ark-poly's trim per column.
inline std::vector<DensePolynomial> interpolate_batch(const Context& ctx, const std::vector<std::vector<Fr>>& columns,
                                                      const Radix2EvaluationDomain& d) {
    const size_t count = columns.size(), n = d.size();
    std::vector<DensePolynomial> out;
    if (!count) return out;
    typlonk_ctx* c = ctx.raw();
    typlonk_buf* buf = nullptr;
    check(typlonk_buf_alloc(c, n * count, &buf), c);
    struct Free {
        typlonk_ctx* c;
        typlonk_buf* b;
        ~Free() { typlonk_buf_free(c, b); }
    } guard{c, buf};
    std::vector<void*> ptrs(count);
    for (size_t v = 0; v < count; ++v) {
        if (columns[v].size() != n) throw std::runtime_error("every column must hold n evaluations");
        check(typlonk_buf_upload(c, buf, v * n, columns[v][0].limbs(), n), c);
        ptrs[v] = (char*)typlonk_buf_devptr(buf) + 32 * n * v;
    }
    check(typlonk_ntt_fr_batch_devptr(c, ptrs.data(), count, d.log_size_of_group(), 1, nullptr), c);
    for (size_t v = 0; v < count; ++v) {
        std::vector<Fr> co(n);
        check(typlonk_buf_download(c, buf, v * n, co[0].limbs(), n), c);
        out.push_back(DensePolynomial::from_coefficients_vec(std::move(co)));
    }
    return out;
}

}  // namespace poly

namespace kzg {

using Poly = poly::DensePolynomial;

// ark_bls12_381::G1Affine at the C ABI: x || y Montgomery limbs + infinity flag
struct G1Point {
    uint64_t xy[12];
    bool infinity;
    bool operator==(const G1Point& o) const {
        if (infinity || o.infinity) return infinity == o.infinity;
        return std::memcmp(xy, o.xy, sizeof(xy)) == 0;
    }
    bool operator!=(const G1Point& o) const { return !(*this == o); }
};

// kzg/src/srs.rs: g1 = [G, sG, s^2 G, ...] of length gates + 3, resident on the device
class Srs {
   public:
    static Srs from_secret(const Context& ctx, const Fr& s, size_t gates) {
        Srs r(ctx);
        r.len_ = gates + 3;
        check(typlonk_srs_generate(ctx.raw(), s.limbs(), 0, r.len_, &r.id_), ctx.raw());
        // srs.rs:26-34: g2 = generator, g2s = [s] generator (host-side: the verifier's two G2 elements)
        r.g2_ = pairing::g2_generator();
        r.g2s_ = pairing::g2_mul(r.g2_, s.v);
        r.has_g2_ = true;
        return r;
    }
    // an SRS loaded from points brings its G2 pair along (needed by KzgScheme::verify only)
    void set_g2(const pairing::G2Affine& g2, const pairing::G2Affine& g2s) {
        g2_ = g2;
        g2s_ = g2s;
        has_g2_ = true;
    }
    const pairing::G2Affine& g2() const {
        if (!has_g2_) throw std::runtime_error("this SRS has no G2 elements");
        return g2_;
    }
    const pairing::G2Affine& g2s() const {
        if (!has_g2_) throw std::runtime_error("this SRS has no G2 elements");
        return g2s_;
    }
    static Srs from_points(const Context& ctx, const std::vector<G1Point>& pts) {
        Srs r(ctx);
        r.len_ = pts.size();
        std::vector<uint64_t> xy(pts.size() * 12);
        std::vector<uint8_t> inf(pts.size());
        for (size_t i = 0; i < pts.size(); ++i) {
            std::memcpy(&xy[i * 12], pts[i].xy, 96);
            inf[i] = pts[i].infinity;
        }
        check(typlonk_srs_load(ctx.raw(), xy.data(), inf.data(), pts.size(), &r.id_), ctx.raw());
        return r;
    }
    Srs(Srs&& o) noexcept : ctx_(o.ctx_), id_(o.id_), len_(o.len_), g2_(o.g2_), g2s_(o.g2s_), has_g2_(o.has_g2_) { o.id_ = 0; }
    Srs(const Srs&) = delete;
    ~Srs() {
        if (id_) typlonk_srs_free(ctx_->raw(), id_);
    }
    size_t len() const { return len_; }
    // optional, once per SRS: fixed-base tables in HBM (window chosen by the library from the length when 0); every later
    // commitment over this SRS is faster, results are unchanged -- nothing in the reference corresponds to it
    void precompute(uint32_t window_bits = 0) const { check(typlonk_srs_precompute(ctx_->raw(), id_, window_bits), ctx_->raw()); }
    std::vector<G1Point> g1_ref() const {  // downloads the points (the reference returns &Vec<G1Point>)
        std::vector<uint64_t> xy(len_ * 12);
        std::vector<uint8_t> inf(len_);
        check(typlonk_srs_download(ctx_->raw(), id_, 0, len_, xy.data(), inf.data()), ctx_->raw());
        std::vector<G1Point> out(len_);
        for (size_t i = 0; i < len_; ++i) {
            std::memcpy(out[i].xy, &xy[i * 12], 96);
            out[i].infinity = inf[i] != 0;
        }
        return out;
    }
    G1Point g1_generator() const {  // g1_ref()[0] without the other len - 1 points
        G1Point g;
        uint8_t inf = 0;
        check(typlonk_srs_download(ctx_->raw(), id_, 0, 1, g.xy, &inf), ctx_->raw());
        g.infinity = inf != 0;
        return g;
    }
    uint32_t id() const { return id_; }
    const Context& ctx() const { return *ctx_; }

   private:
    explicit Srs(const Context& c) : ctx_(&c) {}
    const Context* ctx_;
    uint32_t id_ = 0;
    size_t len_ = 0;
    pairing::G2Affine g2_{}, g2s_{};
    bool has_g2_ = false;
};

struct KzgCommitment {
    G1Point p;
    const G1Point& inner() const { return p; }
    bool operator==(const KzgCommitment& o) const { return p == o.p; }
};
struct KzgOpening {
    G1Point p;
    Fr y;
    Fr eval() const { return y; }
};

class KzgScheme {
   public:
    explicit KzgScheme(const Srs& srs) : srs_(srs) {}
    KzgCommitment commit(const Poly& polynomial) const { return KzgCommitment{evaluate_in_s(polynomial)}; }
    // kzg/src/lib.rs:55-64
    KzgOpening open(const Poly& polynomial, const Fr& z) const {
        if (polynomial.coeffs.empty()) throw std::runtime_error("at least 1");  // `.expect("at least 1")`
        Fr y;
        const Poly q = polynomial.divide_by_linear(z, &y);
        return KzgOpening{evaluate_in_s(q), y};
    }
    KzgCommitment identity() const { return commit(Poly::from_coefficients_vec({Fr(1)})); }
    // kzg/src/lib.rs:66-81: pairing(W, [s]G2 - z G2) == pairing(C - y G1, G2), as e(W, A) * e(-(C - y G1), G2) == 1
    bool verify(const KzgCommitment& commitment, const KzgOpening& opening, const Fr& z) const;
    const Srs& srs() const { return srs_; }

   private:
    // kzg/src/lib.rs:41-54: assert!(srs.len() > degree) then sum coeff_i * srs_i -> the MSM
    G1Point evaluate_in_s(const Poly& polynomial) const {
        if (!(srs_.len() > polynomial.degree())) throw std::runtime_error("assertion failed: srs.len() > polynomial.degree()");
        G1Point out;
        uint8_t inf = 0;
        const uint64_t* sc = polynomial.coeffs.empty() ? nullptr : polynomial.coeffs[0].limbs();
        check(typlonk_msm_g1(srs_.ctx().raw(), srs_.id(), sc, polynomial.coeffs.size(), out.xy, &inf), srs_.ctx().raw());
        out.infinity = inf != 0;
        return out;
    }
    const Srs& srs_;
};

// group helpers used by the tests (KzgCommitment: Add / Mul<Fr>, kzg/src/lib.rs:110-158)
inline G1Point g1_add(const G1Point& a, const G1Point& b) {
    uint64_t xy[24];
    uint8_t inf[2] = {(uint8_t)a.infinity, (uint8_t)b.infinity};
    std::memcpy(xy, a.xy, 96);
    std::memcpy(xy + 12, b.xy, 96);
    G1Point out;
    uint8_t oi = 0;
    check(typlonk_g1_sum_host(xy, inf, 2, out.xy, &oi));
    out.infinity = oi != 0;
    return out;
}
inline G1Point g1_mul(const Context& ctx, const G1Point& p, const Fr& k) {
    Srs one = Srs::from_points(ctx, {p});
    return KzgScheme(one).commit(Poly::from_coefficients_vec({k})).p;
}
inline G1Point g1_neg(const Context& ctx, const G1Point& p) { return g1_mul(ctx, p, -Fr::one()); }
// sum_i k_i P_i: one small MSM over an SRS made of the points (the verifier's linear combinations of commitments)
inline G1Point g1_lincomb(const Context& ctx, const std::vector<G1Point>& pts, const std::vector<Fr>& ks) {
    Srs bases = Srs::from_points(ctx, pts);
    std::vector<Fr> c = ks;
    G1Point out;
    uint8_t inf = 0;
    check(typlonk_msm_g1(ctx.raw(), bases.id(), c[0].limbs(), c.size(), out.xy, &inf), ctx.raw());
    out.infinity = inf != 0;
    return out;
}
inline pairing::G1Aff to_pairing(const G1Point& p, bool negate = false) {
    pairing::G1Aff a;
    std::memcpy(a.x.v, p.xy, 48);
    std::memcpy(a.y.v, p.xy + 6, 48);
    if (negate) a.y = ty::fe_neg(a.y);
    a.infinity = p.infinity;
    return a;
}
// C - y * G on the host (g1_host64.hpp), G = the FIXED generator G1Point::prime_subgroup_generator() the reference's
// verifier uses (kzg/src/lib.rs:76) -- not srs[0], which an Srs::from_points caller is free to choose --, with no device
// round trip: a verifier that checks six openings does twelve double-and-add ladders on the host instead of six SRS
// uploads + MSMs.
inline G1Point g1_sub_y_times_generator(const G1Point& c, const Fr& y) {
    namespace H = ty::h64;
    // arkworks' Montgomery limbs of the generator (the constants pinned in tests/test_oracle.py)
    static const uint64_t GEN[12] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull,
                                     0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull,
                                     0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
    static const uint64_t ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                                    0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};   // R mod p
    auto lift = [&](const uint64_t* xy) {
        H::Xyzz r;
        std::memcpy(r.x.v, xy, 48);
        std::memcpy(r.y.v, xy + 6, 48);
        std::memcpy(r.zz.v, ONE, 48);
        std::memcpy(r.zzz.v, ONE, 48);
        return r;
    };
    const ty::Fr k = ty::fe_from_mont(ty::fe_neg(y.v));   // canonical -y
    const H::Xyzz g = lift(GEN);
    H::Xyzz acc = H::inf();
    for (int w = 7; w >= 0; --w)
        for (int b = 31; b >= 0; --b) {
            acc = H::xyzz_dbl(acc);
            if ((k.v[w] >> b) & 1u) acc = H::xyzz_add(acc, g);
        }
    if (!c.infinity) acc = H::xyzz_add(acc, lift(c.xy));
    G1Point out;
    out.infinity = !H::xyzz_to_affine(acc, out.xy);
    if (out.infinity) {
        std::memset(out.xy, 0, 48);
        std::memcpy(out.xy + 6, ONE, 48);   // GroupAffine::zero() = (0, 1, infinity)
    }
    return out;
}
inline bool KzgScheme::verify(const KzgCommitment& commitment, const KzgOpening& opening, const Fr& z) const {
    const pairing::G2Affine a = pairing::g2_add(srs_.g2s(), pairing::g2_neg(pairing::g2_mul(srs_.g2(), z.v)));
    const G1Point b = g1_sub_y_times_generator(commitment.p, opening.y);  // C - y G1
    const pairing::G1Aff ps[2] = {to_pairing(opening.p), to_pairing(b, /*negate=*/true)};
    const pairing::G2Affine qs[2] = {a, srs_.g2()};
    return pairing::pairing_product_is_one(ps, qs, 2);
}

}  // namespace kzg

namespace plonk {

// plonk::proof::Proof (/root/reference/plonk/src/proof.rs:65-95)
struct PermutationProof {
    kzg::KzgCommitment commitment;
    kzg::KzgOpening z, zw;
};
struct Proof {
    kzg::KzgCommitment a_commit, b_commit, c_commit;
    kzg::KzgOpening a, b, c;          // openings at evaluation_point
    PermutationProof permutation;     // [Z], Z at zeta, Z at zeta * w
    Fr evaluation_point;              // zeta
    kzg::KzgCommitment t[3];          // quotient slices
    kzg::KzgOpening r;                // linearisation polynomial at zeta: r.eval() == 0 for a valid proof
    Fr beta, gamma, alpha;            // the challenges used (the reference's verifier recomputes them, :236-246)
    std::vector<Fr> public_inputs;    // padded column (filled by plonk::Circuit::prove; the verifier reads it, :205-210)
};

// The prover-relevant part of plonk::CompiledCircuit (/root/reference/plonk/src/lib.rs:19-35): the SRS, the domain,
// the five selector polynomials (gate_constrains, builder.rs:76-90) and the sigma columns with their cosets
// (copy_constrains, permutation/src/lib.rs:141-154).  The tables come from the front end
// (CircuitDescription::build; circuit_host.hpp) as EVALUATIONS over the domain, exactly as builder.rs:85 interpolates
// them; they are interpolated on the device once and cached for every later proof (typlonk_circuit_load).
class CompiledCircuit {
   public:
    CompiledCircuit(const kzg::Srs& srs, uint32_t log_n, const std::vector<Fr> (&selector_evals)[5],
                    const std::vector<Fr> (&sigma_evals)[3], const Fr (&cosets)[3])
        : srs_(srs), log_n_(log_n), n_((size_t)1 << log_n) {
        for (int i = 0; i < 3; ++i) cosets_[i] = cosets[i];
        typlonk_ctx* c = srs.ctx().raw();
        typlonk_buf* polys[8];
        for (int k = 0; k < 8; ++k) {
            const std::vector<Fr>& ev = k < 5 ? selector_evals[k] : sigma_evals[k - 5];
            if (ev.size() != n_) throw std::runtime_error("circuit table must hold n evaluations");
            polys[k] = upload(ev);
        }
        {   // interpolate(), builder.rs:84-88 and proof.rs:334-338: the eight columns as ONE batched inverse transform
            void* ptrs[8];
            for (int k = 0; k < 8; ++k) ptrs[k] = typlonk_buf_devptr(polys[k]);
            const int rc = typlonk_ntt_fr_batch_devptr(c, ptrs, 8, log_n, 1, nullptr);
            if (rc < 0) {
                for (typlonk_buf* b : polys) typlonk_buf_free(c, b);
                check(rc, c);
            }
        }
        // the commitments of the fixed polynomials: [q_l] .. [q_c] (builder.rs:86) and [sigma_0..2], which the
        // reference's verifier recomputes with three MSMs on EVERY verify() (permutation/src/lib.rs:178-194 via
        // proof.rs:459) -- here once per circuit, one batch of eight MSMs straight from the interpolated buffers
        {
            const void* ptrs[8];
            size_t lens[8];
            uint64_t xy[8][12];
            uint8_t inf[8];
            for (int k = 0; k < 8; ++k) {
                ptrs[k] = typlonk_buf_devptr(polys[k]);
                lens[k] = n_;
            }
            const int rc = typlonk_msm_g1_batch_devptr(c, srs.id(), ptrs, lens, 8, &xy[0][0], inf);
            if (rc < 0) {
                for (typlonk_buf* b : polys) typlonk_buf_free(c, b);
                check(rc, c);
            }
            for (int k = 0; k < 8; ++k) {
                kzg::G1Point g;
                std::memcpy(g.xy, xy[k], 96);
                g.infinity = inf[k] != 0;
                (k < 5 ? fixed_commitments[k] : sigma_commitments[k - 5]) = kzg::KzgCommitment{g};
            }
        }
        for (int i = 0; i < 3; ++i) {  // the verifier evaluates the sigma polynomials at zeta (sigma_evals)
            std::vector<Fr> co(n_);
            const int rc = typlonk_buf_download(c, polys[5 + i], 0, co[0].limbs(), n_);
            if (rc < 0) {
                for (typlonk_buf* b : polys) typlonk_buf_free(c, b);
                check(rc, c);
            }
            sigma_polys_[i] = poly::DensePolynomial::from_coefficients_vec(std::move(co));
        }
        const typlonk_buf* sel[5] = {polys[0], polys[1], polys[2], polys[3], polys[4]};
        const typlonk_buf* sig[3] = {polys[5], polys[6], polys[7]};
        const int rc = typlonk_circuit_load(c, sel, sig, log_n, &circuit_);
        for (typlonk_buf* b : polys) typlonk_buf_free(c, b);
        check(rc, c);
    }
    CompiledCircuit(const CompiledCircuit&) = delete;
    ~CompiledCircuit() {
        if (circuit_) typlonk_circuit_free(srs_.ctx().raw(), circuit_);
    }
    size_t rows() const { return n_; }
    kzg::KzgCommitment fixed_commitments[5];  // [q_l], [q_r], [q_o], [q_m], [q_c]  (GateConstrains::fixed_commitments)
    kzg::KzgCommitment sigma_commitments[3];  // what CompiledPermutation::sigma_commitments returns

    // CompiledCircuit::prove (proof.rs:26-57) from the point where the witness columns exist: `advice` = the three
    // columns padded to n rows with their blinding rows (:43-49), `public_inputs` = the padded public-input column or
    // empty for the all-zero one (:52-53).  A witness that does not satisfy the circuit throws (the reference panics).
    Proof prove(const std::vector<Fr> (&advice)[3], const std::vector<Fr>& public_inputs = {}) const {
        typlonk_ctx* c = srs_.ctx().raw();
        typlonk_proof raw;
        // the columns go over as they are, in host memory (typlonk_prove_host: each is uploaded right before its
        // interpolation and commitment are queued, beside the previous column's kernels) -- as the Rust layer's Backend::prove
        for (int i = 0; i < 3; ++i)
            if (advice[i].size() != n_) throw std::runtime_error("witness column must hold n values");
        if (!public_inputs.empty() && public_inputs.size() != n_) throw std::runtime_error("public-input column must hold n values");
        uint64_t ks[3][4];
        for (int i = 0; i < 3; ++i) std::memcpy(ks[i], cosets_[i].limbs(), 32);
        const uint64_t* wc[3] = {advice[0][0].limbs(), advice[1][0].limbs(), advice[2][0].limbs()};
        check(typlonk_prove_host(c, srs_.id(), circuit_, wc, public_inputs.empty() ? nullptr : public_inputs[0].limbs(), ks, &raw), c);
        Proof p;
        auto pt = [](const uint64_t xy[12], uint8_t inf) {
            kzg::G1Point g;
            std::memcpy(g.xy, xy, 96);
            g.infinity = inf != 0;
            return g;
        };
        auto fr = [](const uint64_t l[4]) {
            Fr f;
            std::memcpy(f.v.v, l, 32);
            return f;
        };
        p.a_commit = {pt(raw.commit_xy[0], raw.commit_inf[0])};
        p.b_commit = {pt(raw.commit_xy[1], raw.commit_inf[1])};
        p.c_commit = {pt(raw.commit_xy[2], raw.commit_inf[2])};
        const typlonk_proof_tail& t = raw.tail;
        p.a = {pt(t.w_xy[0], t.w_inf[0]), fr(t.evals[0])};
        p.b = {pt(t.w_xy[1], t.w_inf[1]), fr(t.evals[1])};
        p.c = {pt(t.w_xy[2], t.w_inf[2]), fr(t.evals[2])};
        p.permutation.commitment = {pt(raw.z_xy, raw.z_inf)};
        p.permutation.z = {pt(t.w_xy[3], t.w_inf[3]), fr(t.evals[3])};
        p.permutation.zw = {pt(t.w_xy[4], t.w_inf[4]), fr(t.evals[4])};
        p.evaluation_point = fr(raw.zeta);
        for (int i = 0; i < 3; ++i) p.t[i] = {pt(t.t_xy[i], t.t_inf[i])};
        p.r = {pt(t.w_xy[5], t.w_inf[5]), fr(t.evals[5])};
        p.beta = fr(raw.beta);
        p.gamma = fr(raw.gamma);
        p.alpha = fr(raw.alpha);
        return p;
    }

    // plonk::proof::verify (proof.rs:195-235): recompute the challenges from the commitments (verify_challenges,
    // :236-246), check the five openings (verify_openings, :247-272), rebuild the linearisation commitment (:441-503)
    // and check its opening, which must evaluate to zero.  `public_inputs` as for prove().
    //
    // Public inputs: the reference's prover ADDS PI(zeta) to r (proof.rs:401-402) and its verifier adds public_eval to the
    // constant it SUBTRACTS (:497-502), so with PI(zeta) != 0 the reference rejects its own honest proofs (its tests only
    // use vec![0]).  The default follows the reference; PublicInputSign::AsProver is the consistent verifier.
    enum class PublicInputSign { AsReference, AsProver };
    bool verify(const Proof& proof, const std::vector<Fr>& public_inputs = {},
                PublicInputSign pi_sign = PublicInputSign::AsReference) const {
        const Context& ctx = srs_.ctx();
        const kzg::KzgScheme scheme(srs_);
        // verify_challenges
        uint64_t xy[4][12];
        uint8_t inf[4];
        const kzg::G1Point* cm[4] = {&proof.a_commit.p, &proof.b_commit.p, &proof.c_commit.p, &proof.permutation.commitment.p};
        for (int i = 0; i < 4; ++i) {
            std::memcpy(xy[i], cm[i]->xy, 96);
            inf[i] = cm[i]->infinity;
        }
        uint64_t ch[8];
        Fr beta, gamma, alpha, point;
        check(typlonk_transcript_challenges(&xy[0][0], inf, 3, 2, ch));
        std::memcpy(beta.v.v, ch, 32);
        std::memcpy(gamma.v.v, ch + 4, 32);
        check(typlonk_transcript_challenges(&xy[0][0], inf, 4, 2, ch));
        std::memcpy(alpha.v.v, ch, 32);
        std::memcpy(point.v.v, ch + 4, 32);
        if (proof.evaluation_point != point) return false;  // :212-214
        const poly::Radix2EvaluationDomain domain(ctx, n_);
        Fr public_eval = Fr::zero();
        if (!public_inputs.empty()) public_eval = poly::interpolate(public_inputs, domain).evaluate(point);
        // verify_openings
        const Fr w = domain.element(1);
        if (!scheme.verify(proof.a_commit, proof.a, point) || !scheme.verify(proof.b_commit, proof.b, point) ||
            !scheme.verify(proof.c_commit, proof.c, point))
            return false;
        if (!scheme.verify(proof.permutation.commitment, proof.permutation.z, point) ||
            !scheme.verify(proof.permutation.commitment, proof.permutation.zw, point * w))
            return false;
        // linearisation_commitment
        const Fr a = proof.a.eval(), b = proof.b.eval(), c = proof.c.eval();
        const Fr zw_eval = proof.permutation.zw.eval();
        const Fr advice[3] = {a, b, c};
        Fr sigma_evals[3];
        for (int i = 0; i < 3; ++i) sigma_evals[i] = sigma_polys_[i].evaluate(point);  // permutation/src/lib.rs:165-176
        Fr l2 = Fr::one();
        for (int i = 0; i < 3; ++i) l2 *= advice[i] + beta * cosets_[i] * point + gamma;
        const Fr zn = point.pow(n_);
        const Fr vanish = zn - Fr::one();
        Fr l0 = Fr::one();  // L0(zeta) = (zeta^n - 1) / (n (zeta - 1)); 1 at zeta = 1  (utils.rs:150-159)
        if (point != Fr::one()) l0 = vanish * (Fr((int64_t)n_) * (point - Fr::one())).inverse();
        Fr l3 = Fr::one();
        for (int i = 0; i < 2; ++i) l3 *= advice[i] + beta * sigma_evals[i] + gamma;
        const Fr constant = alpha * (l3 * (c + gamma) * zw_eval) + l0 * alpha * alpha +
                            (pi_sign == PublicInputSign::AsReference ? public_eval : -public_eval);
        const std::vector<kzg::G1Point> bases = {fixed_commitments[0].p, fixed_commitments[1].p, fixed_commitments[2].p,
                                                 fixed_commitments[3].p, fixed_commitments[4].p, proof.permutation.commitment.p,
                                                 sigma_commitments[2].p, srs_.g1_generator(), proof.t[0].p, proof.t[1].p, proof.t[2].p};
        const std::vector<Fr> ks = {a, b, -c, a * b, Fr::one(), l2 * alpha + l0 * alpha * alpha,
                                    -(l3 * alpha * beta * zw_eval), -constant, -vanish, -(vanish * zn), -(vanish * zn * zn)};
        const kzg::KzgCommitment r{kzg::g1_lincomb(ctx, bases, ks)};
        return scheme.verify(r, proof.r, point) && proof.r.eval().is_zero();
    }

   private:
    typlonk_buf* upload(const std::vector<Fr>& v) const {
        typlonk_ctx* c = srs_.ctx().raw();
        typlonk_buf* b = nullptr;
        check(typlonk_buf_alloc(c, v.size(), &b), c);
        const int rc = typlonk_buf_upload(c, b, 0, v[0].limbs(), v.size());
        if (rc < 0) {
            typlonk_buf_free(c, b);
            check(rc, c);
        }
        return b;
    }
    const kzg::Srs& srs_;
    uint32_t log_n_;
    size_t n_;
    Fr cosets_[3];
    uint32_t circuit_ = 0;
    poly::DensePolynomial sigma_polys_[3];
};

}  // namespace plonk
}  // namespace typlonk
