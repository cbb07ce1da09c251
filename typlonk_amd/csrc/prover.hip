// libtyplonk_hip.so -- the prover's device-side flow: quotient, grand product, openings, the three rounds, typlonk_prove
// Part of the host driver of include/typlonk.h (see host.hpp for the shared state).  There is deliberately no CPU compute
// fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "host.hpp"
#include "transcript.hpp"

using namespace ty;
using namespace tyh;

// (entry points: C linkage comes from their declarations in include/typlonk.h)

namespace {
const uint64_t* quotient_coset_g(Fr* g_out) {
    // coset generator: Fr's multiplicative generator 7 (7^(4n) != 1, so X^n - 1 never vanishes on g*H_4n)
    static uint64_t limbs[4];
    const Fr g = fr_from_u64(7);
    memcpy(limbs, g.v, sizeof(limbs));
    if (g_out) *g_out = g;
    return limbs;
}

// zero-extend an n-coefficient vector (or the constant-coefficient polynomial `fill`) to 4n and
// evaluate it on the coset g*H_4n, in place in `e`
int quotient_extend(typlonk_ctx* ctx, Fr* e, const Fr* src, const Fr* fill, uint64_t n, uint32_t log4) {
    hipStream_t s = ctx->stream;
    // the coefficients are read in place, zero-padded to 4n by the first pass itself (no copy + 3n-element memset + reads
    // of the zeros: 160 MB of traffic and two launches per extension at n = 2^20)
    if (src) return ntt_run(ctx, e, log4, 0, quotient_coset_g(nullptr), /*sync=*/false, src, n);
    launch_fr_fill(e, n, *fill, s);
    HIPCHK(hipMemsetAsync(e + n, 0, 3 * n * sizeof(Fr), s));
    return ntt_run(ctx, e, log4, 0, quotient_coset_g(nullptr), /*sync=*/false);
}
// the same for `count` coefficient vectors at once (one launch per pass for the whole group)
int quotient_extend_batch(typlonk_ctx* ctx, Fr* const* e, const Fr* const* src, size_t count, uint64_t n, uint32_t log4) {
    return ntt_run_batch(ctx, e, count, log4, 0, quotient_coset_g(nullptr), /*sync=*/false, src, n);
}
}  // namespace

int typlonk_circuit_load(typlonk_ctx* ctx, const typlonk_buf* const selectors[5], const typlonk_buf* const sigma[3],
                         uint32_t log_n, uint32_t* circuit_id) {
    if (!ctx || !selectors || !sigma || !circuit_id) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (log_n < 1 || log_n > TYPLONK_MAX_PROVER_LOG_N) return fail(ctx, TYPLONK_ERR_DOMAIN, "quotient needs 1 <= log_n <= 22");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n, n4 = 4 * n;
    const typlonk_buf* in[8] = {selectors[0], selectors[1], selectors[2], selectors[3], selectors[4],
                                sigma[0], sigma[1], sigma[2]};
    for (const typlonk_buf* b : in)
        if (!b || b->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "circuit polynomial shorter than n");
    CircuitEntry e;
    e.log_n = log_n;
    DevGuard guard;
    HIPCHK(hipMalloc((void**)&e.ext, 9 * n4 * sizeof(Fr)));
    guard.add(e.ext);
    HIPCHK(hipMalloc((void**)&e.coef, 8 * n * sizeof(Fr)));
    guard.add(e.coef);
    HIPCHK(hipMalloc((void**)&e.sig_ev, 3 * n * sizeof(Fr)));
    guard.add(e.sig_ev);
    for (int k = 0; k < 8; ++k)
        HIPCHK(hipMemcpyAsync(e.coef + (uint64_t)k * n, in[k]->d, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(e.sig_ev, e.coef + 5 * n, 3 * n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    ProfilingOff prof_off(ctx);  // stage events are per call
    int rc = TYPLONK_OK;
    const Fr ninv = fe_inv(fr_from_u64(n));
    {
        // the eight coefficient vectors (builder.rs:84-88's five selectors, the three sigmas) as one batch, then L0
        Fr* dst[8];
        const Fr* src[8];
        for (int k = 0; k < 8; ++k) {
            dst[k] = e.ext + (uint64_t)k * n4;
            src[k] = in[k]->d;
        }
        rc = quotient_extend_batch(ctx, dst, src, 8, n, log_n + 2);
        if (!rc) rc = quotient_extend(ctx, e.ext + 8 * n4, nullptr, &ninv, n, log_n + 2);
        Fr* sig[3] = {e.sig_ev, e.sig_ev + n, e.sig_ev + 2 * n};
        if (!rc) rc = ntt_run_batch(ctx, sig, 3, log_n, 0, nullptr, /*sync=*/false);   // proof.rs:334-338
    }
    if (!rc) {
        hipError_t he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, hipGetErrorString(he));
    }
    if (rc) return rc;
    guard.dismiss();
    const uint32_t id = ctx->next_circuit++;
    ctx->circuits[id] = e;
    *circuit_id = id;
    return TYPLONK_OK;
}

int typlonk_circuit_free(typlonk_ctx* ctx, uint32_t circuit_id) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    auto it = ctx->circuits.find(circuit_id);
    if (it == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown circuit id");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(it->second.ext));
    HIPCHK(hipFree(it->second.coef));
    HIPCHK(hipFree(it->second.sig_ev));
    ctx->circuits.erase(it);
    return TYPLONK_OK;
}

namespace {
// typlonk_quotient_dev with bit k of `extended` set when ext[k] (k = 0..4: a, b, c, Z, PI) already holds that
// polynomial's coset evaluations -- the prover session extends them in rounds 1 and 2, beside the commitments
int quotient_run(typlonk_ctx* ctx, const typlonk_quotient_args* args, uint32_t log_n, typlonk_buf* t_out, uint32_t extended);
}  // namespace

int typlonk_quotient_dev(typlonk_ctx* ctx, const typlonk_quotient_args* args, uint32_t log_n, typlonk_buf* t_out) {
    if (!ctx || !args || !t_out) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    // a prover session keeps coset evaluations in the context's quotient workspace between its rounds
    if (ctx->prover_busy) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "a proof is in flight on this context");
    return quotient_run(ctx, args, log_n, t_out, 0);
}

namespace {
int quotient_run(typlonk_ctx* ctx, const typlonk_quotient_args* args, uint32_t log_n, typlonk_buf* t_out, uint32_t extended) {
    if (log_n < 1 || log_n > TYPLONK_MAX_PROVER_LOG_N) return fail(ctx, TYPLONK_ERR_DOMAIN, "quotient needs 1 <= log_n <= 22");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n, n4 = 4 * n;
    const uint32_t log4 = log_n + 2;
    // per-proof inputs first, then (without a cached circuit) the per-circuit ones
    const typlonk_buf* in[13] = {args->wires[0], args->wires[1], args->wires[2], args->z, args->public_inputs,
                                 args->selectors[0], args->selectors[1], args->selectors[2], args->selectors[3],
                                 args->selectors[4], args->sigma[0], args->sigma[1], args->sigma[2]};
    const Fr* cached = nullptr;
    if (args->circuit) {
        auto it = ctx->circuits.find(args->circuit);
        if (it == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown circuit id");
        if (it->second.log_n != log_n) return fail(ctx, TYPLONK_ERR_DOMAIN, "circuit was loaded for another domain size");
        cached = it->second.ext;
    }
    const int n_in = cached ? 5 : 13;
    const bool has_pi = args->public_inputs != nullptr;  // NULL = zero polynomial (public inputs [0])
    for (int k = 0; k < n_in; ++k) {
        if (k == 4 && !has_pi) continue;
        if (!in[k] || in[k]->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "quotient input shorter than n");
    }
    if (t_out->n < n4) return fail(ctx, TYPLONK_ERR_RANGE, "t_out must hold 4n elements");
    int rc = ensure(ctx, ctx->quot_ext, (size_t)(cached ? 5 : 14) * n4 * sizeof(Fr));
    if (rc) return rc;
    Fr* ext = (Fr*)ctx->quot_ext.p;
    hipStream_t s = ctx->stream;
    Fr g;
    const uint64_t* g_limbs = quotient_coset_g(&g);
    ProfilingOff prof_off(ctx);  // stage events are per call
    const Fr ninv = fe_inv(fr_from_u64(n));
    {
        // every input that is not extended yet, as batched coset transforms (one launch per pass and group, ntt_run_batch);
        // L0 (k = 13, a constant-coefficient polynomial) is built in place
        Fr* dst[13];
        const Fr* src[13];
        size_t cnt = 0;
        for (int k = 0; k < (cached ? 5 : 13); ++k) {
            if (k == 4 && !has_pi) continue;
            if (k < 5 && ((extended >> k) & 1u)) continue;
            dst[cnt] = ext + (uint64_t)k * n4;
            src[cnt++] = in[k]->d;
        }
        if (cnt) rc = quotient_extend_batch(ctx, dst, src, cnt, n, log4);
        if (!rc && !cached) rc = quotient_extend(ctx, ext + (uint64_t)13 * n4, nullptr, &ninv, n, log4);
    }
    if (rc) {
        return rc;
    }
    QuotientArgs qa{};
    for (int k = 0; k < 3; ++k) qa.wires[k] = ext + (uint64_t)k * n4;
    qa.z = ext + 3 * n4;
    qa.pi = has_pi ? ext + 4 * n4 : nullptr;
    const Fr* cbase = cached ? cached : ext + 5 * n4;
    for (int k = 0; k < 5; ++k) qa.sel[k] = cbase + (uint64_t)k * n4;
    for (int k = 0; k < 3; ++k) qa.sigma[k] = cbase + (uint64_t)(5 + k) * n4;
    qa.l0 = cbase + 8 * n4;
    qa.out = t_out->d;
    qa.n4 = n4;
    {
        Table lo, hi;
        const Fr w4 = fr_domain_root(log4);
        rc = get_pow2l(ctx, "tw:f:" + std::to_string(log4), w4, Fr::one(), log4, &lo, &hi, &qa.w_h);
        if (rc) {
            return rc;
        }
        qa.w_lo = lo.d;
        const uint64_t n_hi = 1ull << (log4 - qa.w_h);
        rc = ensure(ctx, ctx->quot_tab, n_hi * sizeof(Fr));
        if (rc) return rc;
        memcpy(qa.beta.v, args->beta, 32);
        launch_fr_scale(hi.d, n_hi, fe_mul(qa.beta, g), (Fr*)ctx->quot_tab.p, s);
        qa.bx_hi = (const Fr*)ctx->quot_tab.p;
        // X^n - 1 on the coset: g^n * iota^k - 1 with iota = w_{4n}^n (a primitive 4th root of unity)
        Fr gn = g;
        for (uint32_t i = 0; i < log_n; ++i) gn = fe_sqr(gn);
        Fr iota = w4;
        for (uint32_t i = 0; i < log_n; ++i) iota = fe_sqr(iota);
        Fr cur = gn;
        for (int k = 0; k < 4; ++k) {
            qa.zh_inv[k] = fe_inv(fe_sub(cur, Fr::one()));
            cur = fe_mul(cur, iota);
        }
    }
    memcpy(qa.alpha.v, args->alpha, 32);
    memcpy(qa.gamma.v, args->gamma, 32);
    qa.alpha2 = fe_sqr(qa.alpha);
    for (int k = 0; k < 3; ++k) memcpy(qa.k[k].v, args->cosets[k], 32);
    qa.k0_is_one = qa.k[0] == Fr::one();
    launch_quotient_pointwise(qa, s);
    HIPCHK(hipGetLastError());
    rc = ntt_run(ctx, t_out->d, log4, 1, g_limbs, /*sync=*/false);
    return rc;
}
}  // namespace

int typlonk_grand_product_dev(typlonk_ctx* ctx, const typlonk_buf* const wires[3], const typlonk_buf* const sigma[3],
                              const uint64_t beta[4], const uint64_t gamma[4], const uint64_t cosets[3][4],
                              uint32_t log_n, typlonk_buf* z_evals_out) {
    if (!ctx || !wires || !sigma || !beta || !gamma || !cosets || !z_evals_out)
        return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (log_n > TYPLONK_MAX_PROVER_LOG_N) return fail(ctx, TYPLONK_ERR_DOMAIN, "grand product needs log_n <= 22");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n = 1ull << log_n;
    for (int i = 0; i < 3; ++i)
        if (!wires[i] || !sigma[i] || wires[i]->n < n || sigma[i]->n < n)
            return fail(ctx, TYPLONK_ERR_RANGE, "grand product input shorter than n");
    if (z_evals_out->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "z_evals_out shorter than n");
    const uint64_t nblk = (n + 2047) / 2048;
    int rc = ensure(ctx, ctx->ops_tmp, (4 * n + nblk + 8) * sizeof(Fr));
    if (rc) return rc;
    Fr* num = (Fr*)ctx->ops_tmp.p;
    Fr* den = num + n;
    Fr* npre = den + n;
    Fr* dsuf = npre + n;
    Fr* blk = dsuf + n;
    hipStream_t s = ctx->stream;
    GrandProductArgs a{};
    for (int i = 0; i < 3; ++i) {
        a.wires[i] = wires[i]->d;
        a.sigma[i] = sigma[i]->d;
    }
    a.num = num;
    a.den = den;
    a.n = n;
    memcpy(a.beta.v, beta, 32);
    memcpy(a.gamma.v, gamma, 32);
    for (int i = 0; i < 3; ++i) {
        Fr k;
        memcpy(k.v, cosets[i], 32);
        a.kbeta[i] = fe_mul(k, a.beta);
    }
    {
        Table lo, hi;
        const uint32_t lg = std::max<uint32_t>(log_n, 1);  // a two-level table needs at least one bit
        rc = get_pow2l(ctx, "tw:f:" + std::to_string(lg), fr_domain_root(lg), Fr::one(), lg, &lo, &hi, &a.w_h);
        if (rc) return rc;
        a.w_lo = lo.d;
        a.w_hi = hi.d;
    }
    launch_gp_terms(a, s);
    launch_product_scan(num, n, 0, blk, npre, s);
    launch_product_scan(den, n, 1, blk, dsuf, s);
    HIPCHK(hipGetLastError());
    // S_0 = dsuf[0] = the product of all denominators: inverted ON THE DEVICE (fr_inv.hpp) into the slot behind the carries
    // -- no copy, no wait, no host inversion between the scans and the finish (rounds 1-5 drained the stream here)
    Fr* inv_total = blk + nblk;
    launch_fr_inv(dsuf, inv_total, s);
    launch_gp_finish(npre, dsuf, inv_total, n, z_evals_out->d, s);
    HIPCHK(hipGetLastError());
    return TYPLONK_OK;
}

int typlonk_open_dev(typlonk_ctx* ctx, const typlonk_buf* poly, size_t offset, size_t m, const uint64_t z[4],
                     typlonk_buf* q_out, uint64_t y_out[4]) {
    if (!ctx || !poly || !z || !y_out) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (m < 1) return fail(ctx, TYPLONK_ERR_LENGTH, "open needs at least 1 coefficient (kzg/src/lib.rs:58)");
    if (m > (1u << 22)) return fail(ctx, TYPLONK_ERR_LENGTH, "open supports up to 2^22 coefficients");
    if (offset > poly->n || m > poly->n - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside buffer");
    if (q_out && q_out->n < m - 1) return fail(ctx, TYPLONK_ERR_RANGE, "q_out shorter than m - 1");
    if (q_out && q_out->d == poly->d) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "q_out must not alias poly");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->ops_tmp, (2048 + 8) * sizeof(Fr));
    if (rc) return rc;
    Fr* blocks = (Fr*)ctx->ops_tmp.p;
    Fr* y_dev = blocks + 2048;
    Fr zz;
    memcpy(zz.v, z, 32);
    launch_open(poly->d + offset, m, zz, q_out ? q_out->d : nullptr, blocks, y_dev, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(y_out, y_dev, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TYPLONK_OK;
}

int typlonk_lincomb_dev(typlonk_ctx* ctx, const typlonk_buf* const* polys, const uint64_t (*scalars)[4], size_t terms,
                        const uint64_t* constant, size_t n, typlonk_buf* out) {
    if (!ctx || !out || (terms && (!polys || !scalars))) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (terms > 12) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "at most 12 terms");
    if (out->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "out shorter than n");
    HIPCHK(hipSetDevice(ctx->device));
    LincombArgs a{};
    for (size_t k = 0; k < terms; ++k) {
        if (!polys[k] || polys[k]->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "term shorter than n");
        if (polys[k]->d == out->d) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "out must not alias a term");
        a.poly[k] = polys[k]->d;
        memcpy(a.scalar[k].v, scalars[k], 32);
    }
    a.constant = Fr::zero();
    if (constant) memcpy(a.constant.v, constant, 32);
    a.out = out->d;
    a.n = n;
    a.terms = (uint32_t)terms;
    if (n) launch_lincomb(a, ctx->stream);
    HIPCHK(hipGetLastError());
    return TYPLONK_OK;
}

// ================================================================================================
// The prover's device-side flow: plonk::proof::prove (/root/reference/plonk/src/proof.rs:26-57, 96-194)
// as three rounds around the two Fiat-Shamir squeezes.  Every polynomial stays in HBM from the
// witness upload to the last commitment; the host only handles the few scalars of the linearisation.
struct typlonk_prover {
    typlonk_ctx* ctx = nullptr;
    uint32_t srs_id = 0, circuit = 0, log_n = 0;
    uint64_t n = 0;
    Fr* mem = nullptr;  // one allocation, carved below
    Fr *ev[3], *co[3], *pi, *z, *t, *q[6], *r;
    Fr beta, gamma, k[3];
    bool has_pi = true;
    int round = 0;
    uint32_t extended = 0;  // bit k: coset evaluations of a, b, c, Z, PI already sit in the quotient workspace
    // batched-opening flow (round3_evals / round4_batched)
    bool evals_only = false;
    Fr zeta;
};

namespace {
int prover_commit_batch(typlonk_prover* p, const Fr* const* polys, const size_t* m, size_t count, uint64_t* xy, uint8_t* inf) {
    std::vector<const void*> ptrs(count);
    for (size_t i = 0; i < count; ++i) ptrs[i] = polys[i];
    return msm_batch(p->ctx, p->srs_id, ptrs.data(), m, count, xy, inf);
}
// Coset evaluations of one per-proof quotient input (k = 0..4: a, b, c, Z, PI), queued on the context's stream as soon
// as its coefficients exist.  Rounds 1 and 2 commit on the other lanes at that time, so these transforms fill the
// latency-bound stretches of the MSMs (sort, bucket reduction) instead of sitting on round 3's critical path.
int prover_extend(typlonk_prover* p, int k, const Fr* coeffs) {
    typlonk_ctx* ctx = p->ctx;
    const uint64_t n4 = 4 * p->n;
    int rc = ensure(ctx, ctx->quot_ext, (size_t)5 * n4 * sizeof(Fr));
    if (rc) return rc;
    const Fr ninv = Fr::one();  // unused: src is never null here
    rc = quotient_extend(ctx, (Fr*)ctx->quot_ext.p + (uint64_t)k * n4, coeffs, &ninv, p->n, p->log_n + 2);
    if (!rc) p->extended |= 1u << k;
    return rc;
}
// the same for several inputs at once (ks[]: their slots): one launch per pass for the group
int prover_extend_batch(typlonk_prover* p, const int* ks, const Fr* const* coeffs, size_t count) {
    typlonk_ctx* ctx = p->ctx;
    const uint64_t n4 = 4 * p->n;
    int rc = ensure(ctx, ctx->quot_ext, (size_t)5 * n4 * sizeof(Fr));
    if (rc) return rc;
    Fr* dst[5];
    for (size_t i = 0; i < count; ++i) dst[i] = (Fr*)ctx->quot_ext.p + (uint64_t)ks[i] * n4;
    rc = quotient_extend_batch(ctx, dst, coeffs, count, p->n, p->log_n + 2);
    if (!rc)
        for (size_t i = 0; i < count; ++i) p->extended |= 1u << ks[i];
    return rc;
}
// ops_tmp layout of the prover's openings: [0, 8*2048) per-workgroup carries, then 16 result slots.
// The slots are only ever WRITTEN by kernels (p(z) of an opening) and read by the host, so they live in pinned host memory the
// kernels store into directly: a fetch is one stream synchronisation, no copy.  (Device slots + hipMemcpyAsync into a stack
// array -- pageable, so staged by the runtime -- left the GPU idle for ~170 us before the linearisation and ~120 us before
// round 3's commitments, profiles/r06_prove_timeline.txt 15.50-15.67 and 15.89-16.02 ms.)
constexpr size_t PROVER_EVAL_BLOCKS = 8 * 2048;
int prover_ops_tmp(typlonk_prover* p, Fr** blocks, Fr** slots) {
    typlonk_ctx* ctx = p->ctx;
    int rc = ensure(ctx, ctx->ops_tmp, (PROVER_EVAL_BLOCKS + 16) * sizeof(Fr));
    if (rc) return rc;
    *blocks = (Fr*)ctx->ops_tmp.p;
    *slots = *blocks + PROVER_EVAL_BLOCKS;
    if (ctx->prover_pinned_slots) {
        if (!ctx->eval_slots_host) HIPCHK(hipHostMalloc((void**)&ctx->eval_slots_host, 16 * sizeof(Fr)));
        *slots = ctx->eval_slots_host;
    }
    return TYPLONK_OK;
}
// open() without waiting: p(z) lands in result slot `slot`, the quotient (if q) in q; stream-ordered
int prover_open_async(typlonk_prover* p, const Fr* poly, uint64_t m, const Fr& z, Fr* q, int slot) {
    Fr *blocks, *slots;
    int rc = prover_ops_tmp(p, &blocks, &slots);
    if (rc) return rc;
    typlonk_ctx* ctx = p->ctx;
    launch_open(poly, m, z, q, blocks, slots + slot, ctx->stream);
    HIPCHK(hipGetLastError());
    return TYPLONK_OK;
}
// one synchronisation for `count` result slots
int prover_fetch(typlonk_prover* p, Fr* out, int count) {
    typlonk_ctx* ctx = p->ctx;
    Fr *blocks, *slots;
    int rc = prover_ops_tmp(p, &blocks, &slots);
    if (rc) return rc;
    if (ctx->prover_pinned_slots) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        memcpy((void*)out, (const void*)slots, (size_t)count * sizeof(Fr));
        return TYPLONK_OK;
    }
    HIPCHK(hipMemcpyAsync(out, slots, (size_t)count * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TYPLONK_OK;
}
int prover_open(typlonk_prover* p, const Fr* poly, uint64_t m, const Fr& z, Fr* q, Fr* y) {
    int rc = prover_open_async(p, poly, m, z, q, 0);
    if (rc) return rc;
    return prover_fetch(p, y, 1);
}
}  // namespace

namespace {
// One column of round 1's input: n evaluations on the device (a typlonk_buf) or still on the HOST (typlonk_prove_host: the
// reference's prove() holds its padded, blinded columns as Vec<Fr>, plonk/src/proof.rs:43-49).
struct ColumnSrc {
    const Fr* dev = nullptr;
    const uint64_t* host = nullptr;
    bool present() const { return dev || host; }
};
int prover_round1_impl(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const ColumnSrc (&wires)[3], const ColumnSrc& pi,
                       typlonk_prover** out, uint64_t commit_xy[3][12], uint8_t commit_inf[3]);
}  // namespace

int typlonk_prover_round1(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const wire_evals[3],
                          const typlonk_buf* pi_evals, typlonk_prover** out, uint64_t commit_xy[3][12],
                          uint8_t commit_inf[3]) {
    if (!ctx || !wire_evals || !out || !commit_xy || !commit_inf) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    auto ci = ctx->circuits.find(circuit_id);
    if (ci == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown circuit id");
    const uint64_t n = 1ull << ci->second.log_n;
    ColumnSrc w[3], pi;
    for (int i = 0; i < 3; ++i) {
        if (!wire_evals[i] || wire_evals[i]->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "wire column shorter than n");
        w[i].dev = wire_evals[i]->d;
    }
    if (pi_evals) {
        if (pi_evals->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "public-input column shorter than n");
        pi.dev = pi_evals->d;
    }
    return prover_round1_impl(ctx, srs_id, circuit_id, w, pi, out, commit_xy, commit_inf);
}

namespace {
int prover_round1_impl(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const ColumnSrc (&wires)[3], const ColumnSrc& pi_src,
                       typlonk_prover** out, uint64_t commit_xy[3][12], uint8_t commit_inf[3]) {
    HIPCHK(hipSetDevice(ctx->device));
    auto ci = ctx->circuits.find(circuit_id);
    if (ci == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown circuit id");
    const SrsEntry* srs = nullptr;
    const uint32_t log_n = ci->second.log_n;
    const uint64_t n = 1ull << log_n;
    int rc = msm_validate(ctx, srs_id, n, &srs);  // every committed polynomial has <= n coefficients
    if (rc) return rc;
    if (n > (1u << 22)) return fail(ctx, TYPLONK_ERR_LENGTH, "prover supports up to 2^22 rows");
    if (ctx->prover_busy) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "a proof is already in flight on this context");
    rc = ensure(ctx, ctx->prover_mem, (uint64_t)19 * n * sizeof(Fr));  // 3+3+1+1+4+6+1 vectors, kept across proofs
    if (rc) return rc;
    typlonk_prover* p = new typlonk_prover();
    p->ctx = ctx;
    p->srs_id = srs_id;
    p->circuit = circuit_id;
    p->log_n = log_n;
    p->n = n;
    p->mem = (Fr*)ctx->prover_mem.p;
    Fr* c = p->mem;
    for (int i = 0; i < 3; ++i) { p->ev[i] = c; c += n; }
    for (int i = 0; i < 3; ++i) { p->co[i] = c; c += n; }
    p->pi = c; c += n;
    p->z = c; c += n;
    p->t = c; c += 4 * n;
    for (int i = 0; i < 6; ++i) { p->q[i] = c; c += n; }
    p->r = c;
    hipStream_t s = ctx->stream;
    ProfilingOff prof_off(ctx);  // stage events are per call
    ProverRound in_round(ctx);
    // a, b, c = interpolate(columns) (proof.rs:50); the column values themselves are kept for round 2
    // (proof.rs:113-115 recomputes them with three forward FFTs).  Each commitment (round1, proof.rs:107-110) is
    // submitted to its own lane as soon as its polynomial exists, so the next interpolation and the coset transforms
    // of the quotient inputs run while it is being sorted and accumulated.
    auto d2d = [&](Fr* dst, const Fr* src) -> int {
        const hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(Fr), hipMemcpyDeviceToDevice, s);
        return e == hipSuccess ? TYPLONK_OK : fail(ctx, TYPLONK_ERR_HIP, hipGetErrorString(e));
    };
    // a column into its place on the device: device -> device, or host -> device on the context's stream -- issued column by
    // column, each right before that column's transform and commitment are queued, so column i + 1 crosses PCIe while column i
    // is being transformed, sorted and accumulated
    auto fetch = [&](Fr* dst, const ColumnSrc& c) -> int {
        if (c.dev) return d2d(dst, c.dev);
        const hipError_t e = hipMemcpyAsync(dst, c.host, n * sizeof(Fr), hipMemcpyHostToDevice, s);
        return e == hipSuccess ? TYPLONK_OK : fail(ctx, TYPLONK_ERR_HIP, hipGetErrorString(e));
    };
    MsmQueue q(ctx, srs, /*first_lane=*/1);
    p->has_pi = pi_src.present();  // absent: public inputs [0] -> the zero polynomial
    if (ctx->prover_ntt_batch == 1 || ctx->prover_ntt_batch == 2) {
        // mode 1: the three interpolations (and the public-input column's, proof.rs:105-106) as ONE batched transform
        // (ntt_run_batch), then the three commitments; mode 2: the first column alone -- its commitment starts at once --
        // and the others as one batch beside it.  Either way the coset extensions of the group are one batch.
        for (int i = 0; i < 3 && !rc; ++i) {
            if ((rc = fetch(p->ev[i], wires[i]))) break;
            rc = d2d(p->co[i], p->ev[i]);
        }
        if (!rc && p->has_pi) rc = fetch(p->pi, pi_src);
        Fr* grp[4] = {p->co[0], p->co[1], p->co[2], p->pi};
        const size_t cnt = p->has_pi ? 4 : 3;
        if (ctx->prover_ntt_batch == 2) {
            if (!rc) rc = ntt_run(ctx, p->co[0], log_n, 1, nullptr, false);
            if (!rc) rc = q.submit(p->co[0], n, commit_xy[0], commit_inf);
            if (!rc) rc = ntt_run_batch(ctx, grp + 1, cnt - 1, log_n, 1, nullptr, false);
            for (int i = 1; i < 3 && !rc; ++i) rc = q.submit(p->co[i], n, commit_xy[i], commit_inf + i);
        } else {
            if (!rc) rc = ntt_run_batch(ctx, grp, cnt, log_n, 1, nullptr, false);
            for (int i = 0; i < 3 && !rc; ++i) rc = q.submit(p->co[i], n, commit_xy[i], commit_inf + i);
        }
        const int slots[4] = {0, 1, 2, 4};
        if (!rc) rc = prover_extend_batch(p, slots, grp, cnt);
    } else {
        for (int i = 0; i < 3 && !rc; ++i) {
            if ((rc = fetch(p->ev[i], wires[i]))) break;
            if ((rc = d2d(p->co[i], p->ev[i]))) break;
            if ((rc = ntt_run(ctx, p->co[i], log_n, 1, nullptr, false))) break;
            rc = q.submit(p->co[i], n, commit_xy[i], commit_inf + i);
        }
        if (!rc && p->has_pi) {
            rc = fetch(p->pi, pi_src);
            if (!rc) rc = ntt_run(ctx, p->pi, log_n, 1, nullptr, false);  // proof.rs:105-106
        }
        // the coset transforms of the quotient's per-proof inputs run beside the commitments (measured: -1 % per proof;
        // submitting round 3's first opening MSMs before the quotient loses 1 %: profiles/r02_ab_prover_overlap.txt)
        if (ctx->prover_ntt_batch == 3) {   // mode 3: interpolations one by one (above), the extensions as one batch
            Fr* grp[4] = {p->co[0], p->co[1], p->co[2], p->pi};
            const int slots[4] = {0, 1, 2, 4};
            if (!rc) rc = prover_extend_batch(p, slots, grp, p->has_pi ? 4 : 3);
        } else {
            for (int i = 0; i < 3 && !rc; ++i) rc = prover_extend(p, i, p->co[i]);
            if (!rc && p->has_pi) rc = prover_extend(p, 4, p->pi);
        }
    }
    {
        const int r = q.wait_all();
        if (!rc) rc = r;
    }
    if (rc) {
        delete p;
        return rc;
    }
    p->round = 1;
    ctx->prover_busy = true;
    *out = p;
    return TYPLONK_OK;
}
}  // namespace

int typlonk_prover_round2(typlonk_prover* p, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t cosets[3][4],
                          uint64_t z_xy[12], uint8_t* z_inf) {
    if (!p || !beta || !gamma || !cosets || !z_xy || !z_inf) return TYPLONK_ERR_INVALID_ARG;
    typlonk_ctx* ctx = p->ctx;
    if (p->round != 1) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "round2 must follow round1");
    HIPCHK(hipSetDevice(ctx->device));
    auto cit = ctx->circuits.find(p->circuit);
    if (cit == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "circuit was freed during the proof");
    const CircuitEntry& ce = cit->second;
    const uint64_t n = p->n;
    memcpy(p->beta.v, beta, 32);
    memcpy(p->gamma.v, gamma, 32);
    for (int i = 0; i < 3; ++i) memcpy(p->k[i].v, cosets[i], 32);
    typlonk_buf wb[3] = {{p->ev[0], n}, {p->ev[1], n}, {p->ev[2], n}};
    typlonk_buf sb[3] = {{ce.sig_ev, n}, {ce.sig_ev + n, n}, {ce.sig_ev + 2 * n, n}};
    const typlonk_buf* wp[3] = {&wb[0], &wb[1], &wb[2]};
    const typlonk_buf* sp[3] = {&sb[0], &sb[1], &sb[2]};
    typlonk_buf zb{p->z, n};
    ProfilingOff prof_off(ctx);  // stage events are per call
    ProverRound in_round(ctx);
    int rc = typlonk_grand_product_dev(ctx, wp, sp, beta, gamma, cosets, p->log_n, &zb);  // proof.rs:119-120
    if (!rc) rc = ntt_run(ctx, p->z, p->log_n, 1, nullptr, false);                          // :127-128
    if (!rc) {
        const SrsEntry* srs = nullptr;
        rc = msm_validate(ctx, p->srs_id, n, &srs);
        if (!rc) {
            MsmQueue q(ctx, srs, /*first_lane=*/1);
            rc = q.submit(p->z, n, z_xy, z_inf, /*standalone=*/true);                       // :129
            if (!rc) rc = prover_extend(p, 3, p->z);  // Z's coset transform runs beside its commitment
            const int r = q.wait_all();
            if (!rc) rc = r;
        }
    }
    if (!rc) p->round = 2;
    return rc;
}

namespace {
// Round 3 in both shapes.  tail != NULL: the reference's six separate openings (proof.rs:147-175).
// evals != NULL: evaluations only -- the quotients (p - p(zeta)) / (X - zeta) are not formed here; after
// the caller has squeezed v from the evaluations, round4_batched opens a + v b + v^2 c + v^3 Z + v^4 r once.
int prover_round3_core(typlonk_prover* p, const uint64_t alpha[4], const uint64_t zeta[4], typlonk_proof_tail* out,
                       typlonk_proof_evals* evals_out) {
    const bool batched = evals_out != nullptr;
    typlonk_ctx* ctx = p->ctx;
    if (p->round != 2) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "round3 must follow round2");
    HIPCHK(hipSetDevice(ctx->device));
    auto cit = ctx->circuits.find(p->circuit);
    if (cit == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "circuit was freed during the proof");
    const CircuitEntry& ce = cit->second;
    const uint64_t n = p->n;
    const uint32_t log_n = p->log_n;
    Fr al, ze;
    memcpy(al.v, alpha, 32);
    memcpy(ze.v, zeta, 32);
    ProfilingOff prof_off(ctx);  // stage events are per call
    ProverRound in_round(ctx);
    const SrsEntry* srs = nullptr;
    int rc = msm_validate(ctx, p->srs_id, n, &srs);
    if (rc) return rc;
    // commitments of this round: 6 opening witnesses + 3 quotient slices (:181).  The queue outlives every early
    // return (its destructor-side wait below), because the MSMs write into xy / inf.
    uint64_t xy[9][12];
    uint8_t inf[9];
    MsmQueue q(ctx, srs, /*first_lane=*/1);
    struct WaitAll {
        MsmQueue& q;
        ~WaitAll() { (void)q.wait_all(); }
    } wait_guard{q};
    // ---- openings of a, b, c at zeta; Z at zeta and zeta*w (proof.rs:147-163) ----
    Fr ev[6];
    const Fr w = fr_domain_root(log_n);
    const Fr zw = fe_mul(ze, w);
    Fr s0, s1, pi_z = Fr::zero();
    const Fr one = Fr::one();
    Fr zn = ze, zh = one, l0z = one;   // zeta^n, Z_H(zeta), L0(zeta): filled under the kernels, before the wait
    {
        // ONE synchronisation for everything evaluated here.  Result slots: 0..3 = a, b, c, Z at zeta (with their
        // quotients unless batched), 4, 5 = sigma_0, sigma_1 and 6 = the public-input polynomial at zeta (for the
        // linearisation, proof.rs:376-439, :138), 8 = Z at zeta*w (always with its quotient)
        Fr host[9];
        {
            // all of them in three launches (launch_open_multi): quotients only where the proof shape opens separately
            Fr *blocks = nullptr, *slots = nullptr;
            rc = prover_ops_tmp(p, &blocks, &slots);
            const Fr* polys[8];
            Fr* quots[8];
            Fr* ys[8];
            uint8_t zsel[8];
            uint32_t cnt = 0;
            auto item = [&](const Fr* poly, Fr* quot, int slot, uint8_t at) {
                polys[cnt] = poly;
                quots[cnt] = quot;
                ys[cnt] = slots + slot;
                zsel[cnt++] = at;
            };
            for (int i = 0; i < 3; ++i) item(p->co[i], batched ? nullptr : p->q[i], i, 0);
            item(p->z, batched ? nullptr : p->q[3], 3, 0);
            item(ce.coef + 5 * n, nullptr, 4, 0);             // sigma_0
            item(ce.coef + 6 * n, nullptr, 5, 0);             // sigma_1
            if (p->has_pi) item(p->pi, nullptr, 6, 0);
            item(p->z, p->q[4], 8, 1);                        // Z at zeta * w, always with its quotient
            if (!rc) {
                launch_open_multi(polys, quots, ys, zsel, cnt, n, ze, zw, blocks, ctx->stream);
                const hipError_t he = hipGetLastError();
                if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, hipGetErrorString(he));
            }
        }
        // ---- quotient (proof.rs:139-145): queued behind the opening scans; a, b, c, Z (and PI) were transformed to the
        // coset domain in rounds 1 and 2, so what is left is the pointwise kernel and one inverse transform ----
        if (!rc) {
            typlonk_buf b[5] = {{p->co[0], n}, {p->co[1], n}, {p->co[2], n}, {p->z, n}, {p->pi, n}};
            typlonk_buf tb{p->t, 4 * n};
            typlonk_quotient_args qa{};
            for (int i = 0; i < 3; ++i) qa.wires[i] = &b[i];
            qa.z = &b[3];
            qa.public_inputs = p->has_pi ? &b[4] : nullptr;
            memcpy(qa.alpha, alpha, 32);
            memcpy(qa.beta, p->beta.v, 32);
            memcpy(qa.gamma, p->gamma.v, 32);
            for (int i = 0; i < 3; ++i) memcpy(qa.cosets[i], p->k[i].v, 32);
            qa.circuit = p->circuit;
            rc = quotient_run(ctx, &qa, log_n, &tb, p->extended);
        }
        // what the linearisation needs of zeta alone (one host inversion among it): while the kernels above run
        for (uint32_t i = 0; i < log_n; ++i) zn = fe_sqr(zn);
        zh = fe_sub(zn, one);  // evaluate_vanishing_polynomial(zeta)
        {
            // L0(zeta) = (zeta^n - 1) / (n (zeta - 1)); the polynomial (1/n) sum X^i evaluates to 1 at zeta = 1
            const Fr zm1 = fe_sub(ze, one);
            if (!zm1.is_zero()) l0z = fe_mul(zh, fe_inv(fe_mul(fr_from_u64(n), zm1)));
        }
        if (!rc) rc = prover_fetch(p, host, 9);
        for (int i = 0; i < 4; ++i) ev[i] = host[i];
        ev[4] = host[8];
        s0 = host[4];
        s1 = host[5];
        if (p->has_pi) pi_z = host[6];
    }
    if (!rc) {
        const Fr &a = ev[0], &b = ev[1], &c = ev[2], &zwe = ev[4];
        const Fr &beta = p->beta, &gamma = p->gamma;
        const Fr bz = fe_mul(beta, ze);
        Fr l2 = one;  // prod_i (w_i(zeta) + k_i beta zeta + gamma)
        for (int i = 0; i < 3; ++i) l2 = fe_mul(l2, fe_add(fe_add(ev[i], fe_mul(p->k[i], bz)), gamma));
        const Fr ab = fe_mul(fe_add(fe_add(a, fe_mul(beta, s0)), gamma), fe_add(fe_add(b, fe_mul(beta, s1)), gamma));
        const Fr abz = fe_mul(ab, zwe);          // copy_permutation_ab * Z(zeta w)
        const Fr al2 = fe_sqr(al);
        LincombArgs la{};
        int k = 0;
        auto term = [&](const Fr* poly, const Fr& sc) { la.poly[k] = poly; la.scalar[k] = sc; ++k; };
        term(ce.coef + 0 * n, a);                                   // q_l a
        term(ce.coef + 1 * n, b);                                   // q_r b
        term(ce.coef + 2 * n, fe_neg(c));                           // - q_o c
        term(ce.coef + 3 * n, fe_mul(a, b));                        // q_m a b
        term(ce.coef + 4 * n, one);                                 // q_c
        term(p->z, fe_add(fe_mul(al, l2), fe_mul(al2, l0z)));       // Z (alpha line2 + alpha^2 L0)
        term(ce.coef + 7 * n, fe_neg(fe_mul(al, fe_mul(beta, abz))));  // - alpha beta sigma_2 AB Z(zw)
        term(p->t, fe_neg(zh));                                     // - Z_H t_lo
        term(p->t + n, fe_neg(fe_mul(zh, zn)));                     // - Z_H zeta^n t_mid
        term(p->t + 2 * n, fe_neg(fe_mul(zh, fe_sqr(zn))));         // - Z_H zeta^2n t_hi
        la.terms = (uint32_t)k;
        // constant: PI(zeta) - alpha (gamma + c) AB Z(zw) - alpha^2 L0
        la.constant = fe_sub(fe_sub(pi_z, fe_mul(al, fe_mul(fe_add(gamma, c), abz))), fe_mul(al2, l0z));
        la.out = p->r;
        la.n = n;
        launch_lincomb(la, ctx->stream);
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, hipGetErrorString(he));
    }
    // r(zeta) and its witness polynomial (proof.rs:175).  The reference shape does not wait for the value here: it is an
    // OUTPUT (and the r(zeta) != 0 check), nothing of this round's commitments depends on it -- it is read from its pinned slot
    // once the commitments have been waited for, and the first sort starts without a drain of the context's stream in between.
    const bool late_r = !batched && ctx->prover_pinned_slots;
    if (!rc) rc = late_r ? prover_open_async(p, p->r, n, ze, p->q[5], 0)
                         : prover_open(p, p->r, n, ze, batched ? nullptr : p->q[5], &ev[5]);
    if (!rc && batched) {
        // evaluations only: every commitment of this shape is issued by round4_batched in ONE five-MSM batch
        for (int i = 0; i < 6; ++i) memcpy(evals_out->evals[i], ev[i].v, 32);
        p->zeta = ze;
        p->evals_only = true;
    }
    // ---- the nine remaining commitments in one batch: 6 opening witnesses + 3 quotient slices (:181) ----
    if (!rc && !batched) {
        const Fr* polys[9] = {p->q[0], p->q[1], p->q[2], p->q[3], p->q[4], p->q[5], p->t, p->t + n, p->t + 2 * n};
        const size_t m[9] = {n - 1, n - 1, n - 1, n - 1, n - 1, n - 1, n, n, n > 3 ? n - 3 : 0};
        q.set_first_lane(0);  // the context's stream has nothing left to do but commit
        // All nine polynomials exist once what is queued on the context's stream NOW has run: the lanes wait for this
        // mark, not for "everything on the context's stream at submit time" -- which, with the context's stream itself a
        // lane, included the whole MSM submitted to it just before.  (Rounds 1-4: commitments 4 and 5 started their sorts
        // only when commitment 3 had finished, and 7 and 8 after 6: two stretches of 1.2 ms with no accumulation in
        // flight, profiles/r04_prove_timeline.txt 23.1-24.4 and 30.6-31.9 ms.)
        if (ctx->prover_pipe) {
            if (!ctx->batch_fence) HIPCHK(hipEventCreateWithFlags(&ctx->batch_fence, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->batch_fence, ctx->stream));
            q.fence = ctx->batch_fence;
        }
        for (int k = 0; k < 9 && !rc; ++k) rc = q.submit(polys[k], m[k], xy[k], inf + k);
        {
            const int r = q.wait_all();
            if (!rc) rc = r;
        }
        if (!rc && late_r) rc = prover_fetch(p, &ev[5], 1);   // (every lane has been waited for: this returns at once)
        if (!rc) {
            memcpy(out->w_xy, xy, 6 * 96);
            memcpy(out->w_inf, inf, 6);
            memcpy(out->t_xy, xy[6], 3 * 96);
            memcpy(out->t_inf, inf + 6, 3);
            for (int i = 0; i < 6; ++i) memcpy(out->evals[i], ev[i].v, 32);
        }
    }
    if (!rc) {
        p->round = 3;
        // the verifier's check (proof.rs:234-235).  A witness that violates a gate makes the reference panic in
        // vanishes() (:321, :361); here the division by Z_H leaves a remainder the slices drop, and r(zeta) != 0
        // is how that shows.  Everything in `out` is filled; the caller learns the proof cannot verify.
        if (!ev[5].is_zero())
            return fail(ctx, TYPLONK_ERR_UNSATISFIED, "r(zeta) != 0: the witness does not satisfy the circuit (proof.rs:234-235)");
    }
    return rc;
}
}  // namespace

int typlonk_prover_round3(typlonk_prover* p, const uint64_t alpha[4], const uint64_t zeta[4], typlonk_proof_tail* out) {
    if (!p || !alpha || !zeta || !out) return TYPLONK_ERR_INVALID_ARG;
    return prover_round3_core(p, alpha, zeta, out, nullptr);
}

int typlonk_prover_round3_evals(typlonk_prover* p, const uint64_t alpha[4], const uint64_t zeta[4],
                                typlonk_proof_evals* out) {
    if (!p || !alpha || !zeta || !out) return TYPLONK_ERR_INVALID_ARG;
    return prover_round3_core(p, alpha, zeta, nullptr, out);
}

int typlonk_prover_round4_batched(typlonk_prover* p, const uint64_t v[4], typlonk_proof_batched* out) {
    if (!p || !v || !out) return TYPLONK_ERR_INVALID_ARG;
    typlonk_ctx* ctx = p->ctx;
    if (p->round != 3 || !p->evals_only) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "round4_batched must follow round3_evals");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n = p->n;
    ProfilingOff prof_off(ctx);  // stage events are per call
    ProverRound in_round(ctx);
    // F = a + v b + v^2 c + v^3 Z + v^4 r; division by (X - zeta) is linear, so its witness is
    // sum_i v^i W_i of the six-opening proof
    LincombArgs la{};
    const Fr* polys[5] = {p->co[0], p->co[1], p->co[2], p->z, p->r};
    Fr vv, pw = Fr::one();
    memcpy(vv.v, v, 32);
    for (int i = 0; i < 5; ++i) {
        la.poly[i] = polys[i];
        la.scalar[i] = pw;
        pw = fe_mul(pw, vv);
    }
    la.terms = 5;
    la.constant = Fr::zero();
    la.out = p->q[5];
    la.n = n;
    launch_lincomb(la, ctx->stream);
    int rc = TYPLONK_OK;
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, hipGetErrorString(he));
    if (!rc) rc = prover_open_async(p, p->q[5], n, p->zeta, p->q[0], 0);  // F(zeta) itself is not needed: stream-ordered
    if (!rc) {
        // one batch: [t_lo], [t_mid], [t_hi] (proof.rs:181), the witness of Z at zeta*w, the batched witness at zeta
        const Fr* ms[5] = {p->t, p->t + n, p->t + 2 * n, p->q[4], p->q[0]};
        const size_t m[5] = {n, n, n > 3 ? n - 3 : 0, n - 1, n - 1};
        uint64_t xy[5][12];
        uint8_t inf[5];
        rc = prover_commit_batch(p, ms, m, 5, &xy[0][0], inf);
        if (!rc) {
            memcpy(out->t_xy, xy, 3 * 96);
            memcpy(out->t_inf, inf, 3);
            memcpy(out->w_xy[1], xy[3], 96);
            out->w_inf[1] = inf[3];
            memcpy(out->w_xy[0], xy[4], 96);
            out->w_inf[0] = inf[4];
            p->round = 4;
        }
    }
    return rc;
}

int typlonk_transcript_challenges(const uint64_t* xy, const uint8_t* inf, size_t count, size_t n_challenges, uint64_t* out) {
    if ((!xy && count) || (!out && n_challenges)) return TYPLONK_ERR_INVALID_ARG;
    ChallengeGenerator g;
    for (size_t i = 0; i < count; ++i) g.digest(xy + 12 * i, inf ? inf[i] : 0);
    g.generate(n_challenges, out);
    return TYPLONK_OK;
}

namespace {
int prove_impl(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const* wire_bufs, const typlonk_buf* pi_buf,
               const uint64_t* const* wire_host, const uint64_t* pi_host, const uint64_t cosets[3][4], typlonk_proof* out);
}  // namespace

int typlonk_prove(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const wire_evals[3],
                  const typlonk_buf* pi_evals, const uint64_t cosets[3][4], typlonk_proof* out) {
    if (!ctx || !wire_evals || !cosets || !out) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    return prove_impl(ctx, srs_id, circuit_id, wire_evals, pi_evals, nullptr, nullptr, cosets, out);
}

int typlonk_prove_host(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const uint64_t* const wire_evals[3],
                       const uint64_t* pi_evals, const uint64_t cosets[3][4], typlonk_proof* out) {
    if (!ctx || !wire_evals || !cosets || !out) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < 3; ++i)
        if (!wire_evals[i]) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null wire column");
    return prove_impl(ctx, srs_id, circuit_id, nullptr, nullptr, wire_evals, pi_evals, cosets, out);
}

namespace {
int prove_impl(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const* wire_bufs, const typlonk_buf* pi_buf,
               const uint64_t* const* wire_host, const uint64_t* pi_host, const uint64_t cosets[3][4], typlonk_proof* out) {
    typlonk_prover* p = nullptr;
    // An SRS shard on a context with a communicator: every round's partial commitments are folded over the ranks (one
    // all-gather per round), so all ranks hash the same points and end with the same proof.  A rank whose round fails
    // (an OOM, say) still joins that round's collective with flagged records, so its peers return TYPLONK_ERR_COMM
    // instead of waiting for ever (comm_fold).
    const bool folds = comm_folds(ctx, srs_id);
    int rc;
    if (wire_host) {   // the columns are on the host: each is uploaded right before its transform and commitment are queued
        auto ci = ctx->circuits.find(circuit_id);
        if (ci == ctx->circuits.end()) {
            rc = fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown circuit id");
        } else {
            ColumnSrc w[3], pi;
            for (int i = 0; i < 3; ++i) w[i].host = wire_host[i];
            pi.host = pi_host;
            rc = prover_round1_impl(ctx, srs_id, circuit_id, w, pi, &p, out->commit_xy, out->commit_inf);
        }
    } else {
        rc = typlonk_prover_round1(ctx, srs_id, circuit_id, wire_bufs, pi_buf, &p, out->commit_xy, out->commit_inf);
    }
    if (folds) rc = comm_fold(ctx, &out->commit_xy[0][0], out->commit_inf, 3, rc);
    if (rc) {
        if (p) typlonk_prover_free(p);
        return rc;
    }
    // (beta, gamma) <- H([a], [b], [c])                                                   proof.rs:111
    ChallengeGenerator g;
    for (int i = 0; i < 3; ++i) g.digest(out->commit_xy[i], out->commit_inf[i]);
    uint64_t ch[8];
    g.generate(2, ch);
    memcpy(out->beta, ch, 32);
    memcpy(out->gamma, ch + 4, 32);
    rc = typlonk_prover_round2(p, out->beta, out->gamma, cosets, out->z_xy, &out->z_inf);
    if (folds) rc = comm_fold(ctx, out->z_xy, &out->z_inf, 1, rc);
    if (!rc) {
        // (alpha, zeta) <- H([a], [b], [c], [Z])                                          proof.rs:133-136
        g.digest(out->z_xy, out->z_inf);
        g.generate(2, ch);
        memcpy(out->alpha, ch, 32);
        memcpy(out->zeta, ch + 4, 32);
        rc = typlonk_prover_round3(p, out->alpha, out->zeta, &out->tail);
        if (folds) {   // (an unsatisfied witness, r(zeta) != 0, is the same on every rank: the points are still folded)
            const int round_rc = rc;
            uint64_t xy[9][12];
            uint8_t inf[9];
            memcpy(xy, out->tail.t_xy, 3 * 96);
            memcpy(xy + 3, out->tail.w_xy, 6 * 96);
            memcpy(inf, out->tail.t_inf, 3);
            memcpy(inf + 3, out->tail.w_inf, 6);
            const int r2 = comm_fold(ctx, &xy[0][0], inf, 9, round_rc == TYPLONK_ERR_UNSATISFIED ? TYPLONK_OK : round_rc);
            memcpy(out->tail.t_xy, xy, 3 * 96);
            memcpy(out->tail.w_xy, xy + 3, 6 * 96);
            memcpy(out->tail.t_inf, inf, 3);
            memcpy(out->tail.w_inf, inf + 3, 6);
            if (r2) rc = r2;
            else rc = round_rc;
        }
    }
    typlonk_prover_free(p);
    return rc;
}
}  // namespace

void typlonk_prover_free(typlonk_prover* p) {
    if (!p) return;
    (void)hipStreamSynchronize(p->ctx->stream);
    p->ctx->prover_busy = false;
    delete p;
}

