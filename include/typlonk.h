/* typlonk.h -- C ABI of libtyplonk_hip.so: MI355X (gfx950) backend for TyPLONK's MSM + NTT hot path.
 *
 * The reference (fabrizio-m/TyPLONK, Rust) has no FFI of its own; these entry points are what a
 * Rust `extern "C"` block would bind to replace the two seams named in SURVEY.md section 8(b):
 *
 *   MSM seam  kzg::KzgScheme::evaluate_in_s            /root/reference/kzg/src/lib.rs:41-54
 *             (reached from commit :37-40, open :55-64, identity :82-85)
 *   SRS       kzg::srs::Srs::g1 / g1_ref               /root/reference/kzg/src/srs.rs:8-12, 43-45
 *   NTT seam  Evaluations::interpolate (ark-poly ifft) /root/reference/plonk/src/proof.rs:50,106,125,128,337,415
 *             DensePolynomial::evaluate_over_domain    /root/reference/plonk/src/proof.rs:115
 *
 * Data formats (all little-endian, exactly arkworks 0.3.0's in-memory form, so a Rust caller passes
 * `fr.0.0` / `pt.x.0.0` with no conversion):
 *   Fr  : 4 x uint64 limbs, Montgomery residue, R = 2^256
 *   Fq  : 6 x uint64 limbs, Montgomery residue, R = 2^384
 *   G1  : 12 x uint64 = x limbs then y limbs, plus a separate flag byte (1 = point at infinity).
 *         The identity is returned as x = 0, y = 1 (Montgomery one), inf = 1 -- ark-ec's
 *         GroupAffine::zero().
 *
 * Conventions: every function returns TYPLONK_OK (0) or a negative error code; the caller owns all
 * host buffers and the library never keeps a host pointer after returning; calls block until the
 * result is on the host unless documented otherwise; one host thread per context.  Results are
 * deterministic and bit-exact (modular integer arithmetic only).
 */
#ifndef TYPLONK_H
#define TYPLONK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TYPLONK_OK 0
#define TYPLONK_ERR_INVALID_ARG (-1)   /* NULL pointer, bad handle                                        */
#define TYPLONK_ERR_LENGTH (-2)        /* m > srs length: the reference's assert!, kzg/src/lib.rs:43      */
#define TYPLONK_ERR_DOMAIN (-3)        /* log_n outside 0..32: the reference's unwrap(), builder.rs:70    */
#define TYPLONK_ERR_NO_DEVICE (-4)     /* no usable HIP device (the library has NO CPU fallback)          */
#define TYPLONK_ERR_HIP (-5)           /* a HIP runtime call failed; see typlonk_last_error               */
#define TYPLONK_ERR_OOM (-6)           /* device allocation failed                                        */
#define TYPLONK_ERR_RANGE (-7)         /* offset/length outside a device buffer                            */
#define TYPLONK_ERR_UNSATISFIED (-8)   /* the witness does not satisfy the circuit: r(zeta) != 0.  The reference
                                          panics (vanishes(), plonk/src/proof.rs:321, 361, 504-507) or produces a
                                          proof its own verifier rejects (:234-235)                          */

#define TYPLONK_ERR_COMM (-9)          /* an RCCL call failed or librccl could not be loaded; see typlonk_last_error */

typedef struct typlonk_ctx typlonk_ctx; /* one HIP device + stream + workspaces + cached NTT plans */
typedef struct typlonk_buf typlonk_buf; /* device-resident vector of Fr elements                   */

/* ---- context --------------------------------------------------------------------------------- */
int typlonk_init(typlonk_ctx** out, int device_ordinal);
void typlonk_destroy(typlonk_ctx* ctx);
const char* typlonk_strerror(int code);
const char* typlonk_last_error(const typlonk_ctx* ctx); /* detail text of the last failure on ctx  */
/* Run all subsequent work of `ctx` on an existing hipStream_t (e.g. the caller's torch stream);
 * NULL restores the context's own stream.
 *
 * STREAM ORDERING of device-resident inputs.  Every entry point that reads caller-owned device memory
 * (typlonk_msm_g1_devptr, typlonk_msm_g1_batch_devptr, typlonk_ntt_fr_devptr, typlonk_ntt_fr_batch_devptr, and the typlonk_buf forms when the
 * buffer was written through typlonk_buf_devptr) reads it in the order of the CONTEXT'S STREAM.  The context's own
 * stream is an ordinary (blocking) HIP stream, so it is also ordered after everything previously submitted to the
 * legacy default stream -- which is where PyTorch-ROCm runs unless told otherwise.  A producer on any OTHER stream
 * (a non-blocking stream, a torch side stream) must either be synchronised before the call or be made the
 * context's stream with typlonk_set_stream; otherwise the MSM / NTT may read half-written input. */
int typlonk_set_stream(typlonk_ctx* ctx, void* hip_stream);
int typlonk_sync(typlonk_ctx* ctx);

/* ---- SRS: the fixed MSM base vector ([s^i]G, affine), uploaded once per circuit ---------------- */
/* xy: len*12 limbs; inf: len flag bytes or NULL (= no identity points).  Returns a handle id. */
int typlonk_srs_load(typlonk_ctx* ctx, const uint64_t* xy, const uint8_t* inf, size_t len, uint32_t* srs_id);
/* Build g1[i] = [secret^(start+i)] G, i < len, on the device (Srs::from_secret, kzg/src/srs.rs:15-34;
 * `start` lets each GPU create only its shard).  secret: 4 limbs (Montgomery).  Setup-time only. */
int typlonk_srs_generate(typlonk_ctx* ctx, const uint64_t secret[4], uint64_t start, size_t len, uint32_t* srs_id);
/* Copy `count` points starting at `offset` back to the host (xy: count*12 limbs; inf: count or NULL). */
int typlonk_srs_download(typlonk_ctx* ctx, uint32_t srs_id, size_t offset, size_t count, uint64_t* xy, uint8_t* inf);
/* Optional fixed-base precomputation (the SRS is immutable per circuit, plonk/src/lib.rs:22): builds the
 * tables 2^(c*t) * g1[i] for every window t (T = ceil(256/c) copies of the SRS in HBM -- 15 for c = 17 and 17 for
 * c = 15, which slice centred scalars |k| < 2^254 --, c = window_bits in 14..20, or 0 = chosen by length: 15 below
 * 2^16 points, 17 below 2^19 -- an index shard --, else 20, whose top window still has 15 bits; with 0 an SRS shorter than
 * TYPLONK_TABLES_AUTO_MIN_LEN points gets NO tables and the call returns TYPLONK_OK: 2^16 buckets for a handful of
 * terms would be slower than the plain path).  Later MSMs of at least len/4
 * terms over this SRS then let all windows share ONE bucket set: no cross-window doublings on the
 * host, fewer windows, fewer bucket additions.  Results are unchanged bit for bit.  Setup-time cost: T*c Jacobian doublings + ONE shared inversion per point
 * (plus T * len * 48 bytes of scratch for the duration of the call; TYPLONK_ERR_OOM if either allocation is refused).
 * An MSM length the table-mode sort cannot handle (more than 2^22 terms with 20-bit windows) silently takes the
 * plain path over the same SRS: precomputation never turns a valid MSM into an error. */
#define TYPLONK_TABLES_AUTO_MIN_LEN 16384
int typlonk_srs_precompute(typlonk_ctx* ctx, uint32_t srs_id, uint32_t window_bits);
/* Multi-GPU: declare that this entry holds bases [first_index, first_index + len) of a total_len-point SRS
 * (one process per GPU, each with its own slice).  Every MSM / prover call on it then takes the FULL
 * coefficient vector (pointer to coefficient 0, global length m <= total_len, same length check as
 * kzg/src/lib.rs:43) and returns this rank's PARTIAL sum over its index range -- the identity if the range is
 * empty.  The partial points are folded by typlonk_comm_fold_g1 / the *_sharded_* entry points below (RCCL), or by
 * the caller's own exchange + typlonk_g1_sum_host. */
int typlonk_srs_set_shard(typlonk_ctx* ctx, uint32_t srs_id, size_t first_index, size_t total_len);
int typlonk_srs_free(typlonk_ctx* ctx, uint32_t srs_id);
int typlonk_srs_len(typlonk_ctx* ctx, uint32_t srs_id, size_t* len);

/* ---- multi-GPU exchange: one process per GPU, RCCL over xGMI ------------------------------------------------------
 * The reference's evaluate_in_s returns the FULL sum (kzg/src/lib.rs:41-54); with the base vector index-sharded over
 * the GPUs of a node (typlonk_srs_set_shard) the full sum is the fold of the ranks' partial sums.  RCCL has no
 * elliptic-curve reduction, so the library's "all-reduce" is one ncclAllGather of 104-byte records (12 limbs + flag)
 * on the context's stream followed by a fold in rank order 0..world-1 on every rank: the same bits everywhere.
 * librccl is loaded on first use (dlopen "librccl.so.1"): single-GPU callers never need it.
 *   typlonk_comm_unique_id   rank 0 creates the rendezvous id (ncclGetUniqueId) and hands the 128 bytes to the other
 *                            ranks by whatever channel the host program has (a file, a socket, MPI, a Rust channel).
 *   typlonk_comm_init        every rank: ncclCommInitRank on the context's device (collective: blocks until all
 *                            `world` ranks have called it).  world = 1 is allowed (the exchange is then a local copy).
 *   typlonk_comm_fold_g1     in place: count points in, on return each holds sum over ranks of that rank's point
 *                            (collective).  What a caller that drives the prover rounds itself uses between rounds.
 *   typlonk_msm_g1_sharded_devptr / _batch_devptr
 *                            typlonk_msm_g1_devptr / _batch_devptr on an SRS shard + the fold: every rank passes the
 *                            full coefficient vector(s) and gets the full sum(s) (collective).
 * typlonk_prove on a context with a communicator and an SRS shard folds the commitments of every round itself, so
 * all ranks hash identical points, squeeze identical challenges and return the identical proof.
 * Failure on one rank: a member of a communicator never leaves between "decided to fold" and the collective.  The
 * *_sharded_* entry points, typlonk_comm_fold_g1 and typlonk_prove join the collective of the step that failed with
 * flagged records -- whether the failure is the local MSM / prover round, a bad argument (a null output, m > len) or the
 * staging copy of the records itself (the send buffer is kept poisoned except between a successful copy and the
 * all-gather) -- so that rank returns its own error code and every other rank TYPLONK_ERR_COMM (typlonk_last_error
 * names the rank); nobody is left waiting and the communicator stays usable.  The exchange buffers are allocated by
 * typlonk_comm_init BEFORE it joins ncclCommInitRank (a rank that cannot allocate never becomes a member); a fold never
 * allocates (more than 32 points go through in pieces).  Exercised with 2 and 8 ranks: tests/test_gpu_dist.py.
 *   typlonk_comm_available   1 if librccl can be loaded in this process (TYPLONK_RCCL_LIB names it, default: the SONAME
 *                            librccl.so.1), else 0.  NOT collective: ranks agree on it BEFORE the collective
 *                            typlonk_comm_init, in which a rank that cannot load the library would leave the others waiting. */
#define TYPLONK_COMM_ID_BYTES 128
int typlonk_comm_available(void);
int typlonk_comm_unique_id(uint8_t id[TYPLONK_COMM_ID_BYTES]);
int typlonk_comm_init(typlonk_ctx* ctx, const uint8_t id[TYPLONK_COMM_ID_BYTES], int rank, int world);
int typlonk_comm_destroy(typlonk_ctx* ctx);
int typlonk_comm_info(const typlonk_ctx* ctx, int* rank, int* world); /* world = 0: no communicator */
int typlonk_comm_fold_g1(typlonk_ctx* ctx, uint64_t* xy /* count*12, in/out */, uint8_t* inf /* count, in/out */,
                         size_t count);
int typlonk_msm_g1_sharded_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* d_scalars, size_t m,
                                  uint64_t out_xy[12], uint8_t* out_inf);
int typlonk_msm_g1_sharded_batch_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m,
                                        size_t count, uint64_t* out_xy, uint8_t* out_inf);

/* ---- MSM: sum_{i<m} scalars[i] * srs[i]  (== evaluate_in_s with coeffs = scalars) --------------- */
/* scalars: m Fr elements (Montgomery).  0 <= m <= srs length, else TYPLONK_ERR_LENGTH.  m = 0 gives
 * the identity (the reference's empty `.sum()`). */
int typlonk_msm_g1(typlonk_ctx* ctx, uint32_t srs_id, const uint64_t* scalars, size_t m,
                   uint64_t out_xy[12], uint8_t* out_inf);
/* Same with the scalars already in HBM: a typlonk_buf range, or a raw device pointer. */
int typlonk_msm_g1_dev(typlonk_ctx* ctx, uint32_t srs_id, const typlonk_buf* scalars, size_t offset, size_t m,
                       uint64_t out_xy[12], uint8_t* out_inf);
int typlonk_msm_g1_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* d_scalars, size_t m,
                          uint64_t out_xy[12], uint8_t* out_inf);

/* `count` independent MSMs over the same SRS (prove() issues them in groups: the three wire
 * commitments plonk/src/proof.rs:107-110, the openings :147-175, the quotient slices :181).  Three
 * are kept in flight on separate workspaces/streams (TYPLONK_MSM_INFLIGHT=1..4), their accumulations one after the
 * other, so one MSM's sort and reduction tail run beside another's accumulation.  d_scalars[k]: device pointer to m[k] Fr elements; out_xy: count*12 limbs; out_inf: count. */
int typlonk_msm_g1_batch_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m,
                                size_t count, uint64_t* out_xy, uint8_t* out_inf);

/* ---- NTT over Fr: radix-2 domain of size 2^log_n with arkworks' generator -------------------------
 * omega = TWO_ADIC_ROOT_OF_UNITY^(2^(32-log_n)).  Natural order in, natural order out, in place.
 *   inverse = 0: data[k] <- sum_i data[i] * (g*omega^k)^i                (fft / coset_fft)
 *   inverse = 1: data[i] <- g^-i * n^-1 * sum_k data[k] * omega^(-ik)    (ifft / coset_ifft)
 * coset_shift: NULL (g = 1) or 4 limbs (Montgomery).  The caller zero-pads to 2^log_n.
 * Sizes: every log_n <= 32 is accepted (the reference's domain constructor fails above the two-adicity, builder.rs:70);
 * the parity suite compares whole vectors with the CPU restatement up to 2^27 (a 4-GiB vector); one-multiplication
 * twiddle tables are kept up to 2^24 points, larger transforms compose their twiddles from two-level tables.
 * Tables keyed by a caller-chosen coset shift are a cache (8 groups / 3 GiB): building a ninth evicts the least recently
 * used group after a hipDeviceSynchronize() -- a device-wide wait, so a caller that cycles through many shifts stalls
 * every stream of the device at each eviction (the prover uses one shift, which stays resident).
 * typlonk_ntt_fr blocks until the result is back on the host; the _dev / _devptr forms (and
 * typlonk_quotient_dev) are stream-ordered on the context's stream and return once enqueued -- any
 * later call on the same context, a typlonk_buf_download or typlonk_sync observes the result. */
int typlonk_ntt_fr(typlonk_ctx* ctx, uint64_t* data, uint32_t log_n, int inverse, const uint64_t* coset_shift);
int typlonk_ntt_fr_dev(typlonk_ctx* ctx, typlonk_buf* buf, size_t offset, uint32_t log_n, int inverse,
                       const uint64_t* coset_shift);
int typlonk_ntt_fr_devptr(typlonk_ctx* ctx, void* d_data, uint32_t log_n, int inverse,
                          const uint64_t* coset_shift);
/* `count` transforms of the same size, direction and coset in one call: d_data[v] is a device pointer to 2^log_n Fr
 * elements, transformed in place; the vectors must not overlap (TYPLONK_ERR_INVALID_ARG).  The reference always
 * transforms in groups -- `interpolate` of the three wire columns (plonk/src/proof.rs:50), their re-evaluation
 * (:113-115), the three sigma columns (:334-338, :412-418), the five selector columns (plonk/src/builder.rs:84-88) --
 * so every pass of the group is ONE launch carrying count x the tiles of a single vector over shared twiddle tables
 * (as typlonk_msm_g1_batch_devptr does for the group's commitments).  Bit-identical to `count` single calls;
 * count = 0 is a no-op; stream-ordered like the _devptr form. */
int typlonk_ntt_fr_batch_devptr(typlonk_ctx* ctx, void* const* d_data, size_t count, uint32_t log_n, int inverse,
                                const uint64_t* coset_shift);

/* ---- quotient polynomial (plonk::proof::quotient_polynomial, /root/reference/plonk/src/proof.rs:292-375)
 * All inputs are device-resident coefficient vectors of n = 2^log_n elements (zero padded), as
 * produced by typlonk_ntt_fr_dev(inverse): the wire polynomials a, b, c (proof.rs:50), the grand
 * product Z (:127), the five selector polynomials q_l q_r q_o q_m q_c (builder.rs:84-88), the three
 * sigma polynomials (proof.rs:334-338) and the public-input polynomial (:105).  Z(wX) is derived
 * internally; public_inputs may be NULL (= the zero polynomial, public inputs [0] as in the reference's
 * README and tests).  Scalars are 4-limb Montgomery Fr: challenges alpha, beta, gamma (proof.rs:111, 133) and
 * the identity-permutation cosets k_0..k_2 (permutation/src/lib.rs:141-154; 2, 3, 4 in the reference).
 * t_out must hold >= 4n elements; on return its first 3n hold the coefficients of t (degree <= 3n - 4;
 * the three commitments of SlicedPoly<3> are MSMs of [0,n), [n,2n), [2n,3n)), the rest is zero.
 * The schoolbook products of the reference are replaced by a 4n coset NTT: identical result
 * whenever the constraint numerator vanishes on the domain (every valid witness). */
/* The prover-side entry points (quotient, grand product, open, the rounds, typlonk_prove) take 1 <= log_n <= 22 -- the
 * sizes the parity suite covers (BASELINE config 5 is 2^22 rows); larger domains return TYPLONK_ERR_DOMAIN / _LENGTH. */
#define TYPLONK_MAX_PROVER_LOG_N 22
typedef struct typlonk_quotient_args {
    const typlonk_buf* wires[3];
    const typlonk_buf* z;
    const typlonk_buf* selectors[5];
    const typlonk_buf* sigma[3];
    const typlonk_buf* public_inputs;
    uint64_t alpha[4], beta[4], gamma[4];
    uint64_t cosets[3][4];
    uint32_t circuit; /* 0, or an id from typlonk_circuit_load: selectors/sigma above are then ignored */
} typlonk_quotient_args;
/* Transform the per-circuit constants of the quotient (five selector + three sigma polynomials, and
 * L0) to the 4n coset domain once; they are fixed per CompiledCircuit (plonk/src/lib.rs:19-35). */
int typlonk_circuit_load(typlonk_ctx* ctx, const typlonk_buf* const selectors[5], const typlonk_buf* const sigma[3],
                         uint32_t log_n, uint32_t* circuit_id);
int typlonk_circuit_free(typlonk_ctx* ctx, uint32_t circuit_id);
int typlonk_quotient_dev(typlonk_ctx* ctx, const typlonk_quotient_args* args, uint32_t log_n, typlonk_buf* t_out);

/* ---- grand product Z of the copy-constraint argument (permutation::CompiledPermutation::prove,
 * /root/reference/permutation/src/proving.rs:7-31, called at plonk/src/proof.rs:119-120).
 * wires[i] / sigma[i]: the i-th witness column and sigma column as EVALUATIONS over the size-2^log_n
 * domain (n elements each); cell (i, j) carries the identity tag cosets[i] * w^j.  z_evals_out receives
 * Z_0 = 1, Z_j = prod_{k<j} prod_i (w_ik + beta id_ik + gamma) / (w_ik + beta sigma_ik + gamma), j < n
 * (the reference's n+1 values without the last one).  Stream-ordered except for one 32-byte readback. */
int typlonk_grand_product_dev(typlonk_ctx* ctx, const typlonk_buf* const wires[3], const typlonk_buf* const sigma[3],
                              const uint64_t beta[4], const uint64_t gamma[4], const uint64_t cosets[3][4],
                              uint32_t log_n, typlonk_buf* z_evals_out);

/* ---- open(): the polynomial half of kzg::KzgScheme::open (/root/reference/kzg/src/lib.rs:55-61).
 * poly: m coefficients starting at `offset` of a device vector.  y_out <- p(z) (Horner, :57); when
 * q_out is not NULL it receives the m - 1 coefficients of (p - p(z)) / (X - z) (:58-61), ready for
 * typlonk_msm_g1_dev (:62).  q_out must not alias poly.  1 <= m <= 2^22.  Blocks for the 32-byte result. */
int typlonk_open_dev(typlonk_ctx* ctx, const typlonk_buf* poly, size_t offset, size_t m, const uint64_t z[4],
                     typlonk_buf* q_out, uint64_t y_out[4]);
/* out[i] = sum_k scalars[k] * polys[k][i] for i < n, plus `constant` (may be NULL) on coefficient 0: the
 * scalar-times-polynomial sums of linearisation_poly (/root/reference/plonk/src/proof.rs:376-439).
 * terms <= 12; every polys[k] holds >= n elements; out may alias none of them.  Stream-ordered. */
int typlonk_lincomb_dev(typlonk_ctx* ctx, const typlonk_buf* const* polys, const uint64_t (*scalars)[4], size_t terms,
                        const uint64_t* constant, size_t n, typlonk_buf* out);

/* ---- the prover's device-side flow: plonk::proof::prove (/root/reference/plonk/src/proof.rs:26-57, 96-194)
 * split at its two Fiat-Shamir squeezes (challenges.rs is CPU-side and out of scope, so the caller
 * supplies the challenges):
 *   round1  columns (EVALUATIONS, n each, blinding rows included -- proof.rs:43-49) and the public-input
 *           column (NULL = all zero) -> a, b, c, PI by iNTT (:50, :105) and the commitments [a], [b], [c] (:107-110)
 *   round2  beta, gamma (:111) -> grand product Z (:119), iNTT (:127), [Z] (:129)
 *   round3  alpha, zeta (:133-136) -> quotient (:139), openings of a, b, c, Z at zeta and Z at zeta*w (:147-163),
 *           linearisation polynomial r and its opening (:165-175), [t_lo], [t_mid], [t_hi] (:181)
 * The circuit id comes from typlonk_circuit_load; the SRS must hold > n points.  n <= 2^22. */
typedef struct typlonk_prover typlonk_prover;
typedef struct typlonk_proof_tail {
    uint64_t t_xy[3][12];   /* quotient slice commitments                                   */
    uint8_t t_inf[3];
    uint64_t w_xy[6][12];   /* opening witnesses: a, b, c at zeta; Z at zeta; Z at zeta*w; r at zeta */
    uint8_t w_inf[6];
    uint64_t evals[6][4];   /* a(zeta) b(zeta) c(zeta) Z(zeta) Z(zeta w) r(zeta)  (r(zeta) = 0 for a valid witness) */
} typlonk_proof_tail;
int typlonk_prover_round1(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const wire_evals[3],
                          const typlonk_buf* pi_evals, typlonk_prover** out, uint64_t commit_xy[3][12],
                          uint8_t commit_inf[3]);
int typlonk_prover_round2(typlonk_prover* p, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t cosets[3][4],
                          uint64_t z_xy[12], uint8_t* z_inf);
/* round3 / round3_evals return TYPLONK_ERR_UNSATISFIED when r(zeta) != 0 (the identity the verifier checks,
 * proof.rs:234-235): the witness does not satisfy the circuit and the quotient had a remainder the slices silently
 * drop.  `out` is completely filled in that case too (what the reference's verifier would be handed), but the
 * proof will not verify. */
int typlonk_prover_round3(typlonk_prover* p, const uint64_t alpha[4], const uint64_t zeta[4], typlonk_proof_tail* out);
/* Batched openings -- the reference's own to-do (/root/reference/README.md:4, "opening batching"); the proof
 * shape differs from proof.rs:178-192, so this is a separate pair of calls and round3 above stays the default.
 *   round3_evals   quotient, linearisation polynomial and the six evaluations -- no commitment yet
 *   round4_batched v (squeezed by the caller from those evaluations) -> one batch of five MSMs: [t_lo], [t_mid],
 *                  [t_hi]; w[0] = witness of a + v b + v^2 c + v^3 Z + v^4 r at zeta, which equals sum_i v^i W_i of
 *                  round3's witnesses (a, b, c, Z, r order); w[1] = witness of Z at zeta*w.
 * 9 MSMs per proof instead of 13.  (As in the reference, t is committed after zeta -- and here v -- are known:
 * plonk/src/proof.rs:133-136 vs :181; the quotient commitments are never hashed.) */
typedef struct typlonk_proof_evals {
    uint64_t evals[6][4];   /* same order as typlonk_proof_tail.evals */
} typlonk_proof_evals;
typedef struct typlonk_proof_batched {
    uint64_t t_xy[3][12];
    uint8_t t_inf[3];
    uint64_t w_xy[2][12];
    uint8_t w_inf[2];
} typlonk_proof_batched;
int typlonk_prover_round3_evals(typlonk_prover* p, const uint64_t alpha[4], const uint64_t zeta[4],
                                typlonk_proof_evals* out);
int typlonk_prover_round4_batched(typlonk_prover* p, const uint64_t v[4], typlonk_proof_batched* out);
void typlonk_prover_free(typlonk_prover* p);

/* ---- prove(): plonk::proof::prove in ONE call (/root/reference/plonk/src/proof.rs:26-57, 96-194), the Fiat-Shamir
 * transcript included.  The rounds above are driven with the challenges the reference's ChallengeGenerator would
 * squeeze (/root/reference/plonk/src/proof/challenges.rs:9-46; restated natively in csrc/transcript.hpp from the
 * published behaviour of ark-serialize, blake2, rand and ark-ff -- not verifiable against Rust in this image):
 * (beta, gamma) from [a], [b], [c] (proof.rs:111), (alpha, zeta) from [a], [b], [c], [Z] (:133-136).  `out` holds the
 * fields of the reference's Proof (proof.rs:65-95): the three wire commitments with their openings, the permutation
 * commitment with its two openings, evaluation_point = zeta, the three quotient-slice commitments and the opening of
 * r; the challenges are returned too (the reference's verifier recomputes them, :236-246).
 * wire_evals / pi_evals / circuit / SRS exactly as for typlonk_prover_round1; cosets as for round2.
 * Returns TYPLONK_ERR_UNSATISFIED (with `out` filled) when r(zeta) != 0. */
typedef struct typlonk_proof {
    uint64_t commit_xy[3][12];  /* [a], [b], [c] */
    uint8_t commit_inf[3];
    uint64_t z_xy[12];          /* [Z] */
    uint8_t z_inf;
    typlonk_proof_tail tail;    /* [t_lo], [t_mid], [t_hi]; witnesses a, b, c, Z at zeta, Z at zeta*w, r; the six evaluations */
    uint64_t beta[4], gamma[4], alpha[4], zeta[4];
} typlonk_proof;
int typlonk_prove(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const wire_evals[3],
                  const typlonk_buf* pi_evals, const uint64_t cosets[3][4], typlonk_proof* out);
/* The same with the columns still in HOST memory -- how the reference holds them when prove() starts (the padded, blinded
 * Vec<Fr> columns of plonk/src/proof.rs:43-49 and the padded public inputs :52-53): wire_evals[i] and pi_evals (NULL = the zero
 * polynomial) point at n = 2^log_n Fr elements each, 4 limbs per element.  Each column is copied to the device right before its
 * interpolation and commitment are queued, so column i + 1 crosses PCIe while column i is transformed, sorted and accumulated
 * (128 MiB of uploads at 2^20 that a caller of typlonk_prove pays before the first kernel starts).  Same proof, bit for bit. */
int typlonk_prove_host(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const uint64_t* const wire_evals[3],
                       const uint64_t* pi_evals, const uint64_t cosets[3][4], typlonk_proof* out);
/* The transcript alone (host-only, no GPU): digest `count` commitments (C-ABI form) in order and squeeze
 * n_challenges Fr elements (4 Montgomery limbs each) -- ChallengeGenerator::with_digest(..).generate_challenges::<N>(). */
int typlonk_transcript_challenges(const uint64_t* xy, const uint8_t* inf, size_t count, size_t n_challenges, uint64_t* out);

/* ---- device-resident Fr vectors (so an iNTT result feeds an MSM without crossing PCIe) ---------- */
int typlonk_buf_alloc(typlonk_ctx* ctx, size_t n_elems, typlonk_buf** out);
int typlonk_buf_free(typlonk_ctx* ctx, typlonk_buf* buf);
int typlonk_buf_upload(typlonk_ctx* ctx, typlonk_buf* buf, size_t offset, const uint64_t* src, size_t n_elems);
int typlonk_buf_download(typlonk_ctx* ctx, const typlonk_buf* buf, size_t offset, uint64_t* dst, size_t n_elems);
int typlonk_buf_zero(typlonk_ctx* ctx, typlonk_buf* buf, size_t offset, size_t n_elems);
size_t typlonk_buf_len(const typlonk_buf* buf);
void* typlonk_buf_devptr(const typlonk_buf* buf);

/* ---- host-only helpers (no GPU needed) --------------------------------------------------------- */
/* Fold `count` affine points in index order: the deterministic combine step after an all-gather of
 * per-GPU partial MSM results (RCCL has no elliptic-curve reduction). */
int typlonk_g1_sum_host(const uint64_t* xy, const uint8_t* inf, size_t count, uint64_t out_xy[12], uint8_t* out_inf);
/* The fold the library applies after its all-gather, exposed for hosts that run their own exchange (MPI, sockets):
 * `records` = what an all-gather of the ranks' send buffers yields, rank-major: world x count records of 13 uint64
 * (12 limbs x || y, then the infinity flag in bits 0..31; bits 32.. non-zero = "this rank failed", its error code).
 * out point i = sum over ranks r of record (r, i), folded in rank order.  Returns TYPLONK_ERR_COMM and the failing
 * rank in *failed_rank (may be NULL) when a record is flagged.  Host-only, no GPU. */
#define TYPLONK_COMM_RECORD_WORDS 13
int typlonk_g1_fold_records_host(const uint64_t* records, size_t world, size_t count, uint64_t* out_xy /* count*12 */,
                                 uint8_t* out_inf /* count */, int* failed_rank);

/* ---- measurement -------------------------------------------------------------------------------
 * With profiling on, every kernel stage of the next MSM / NTT call is bracketed by HIP events on
 * the context's stream.  typlonk_profile_get returns up to `cap` (name, milliseconds) pairs of the
 * last call and the number of stages it had.
 *   on = 0  off
 *   on = 1  every stage (sort, accumulate, reduce, NTT passes ...): ~0.1 ms of event traffic per MSM, and NTT calls wait
 *           for their result
 *   on = 2  the dominant kernel only (the bucket accumulation launches, "msm_accum"): what a timed loop can afford */
int typlonk_set_profiling(typlonk_ctx* ctx, int on);
int typlonk_profile_get(typlonk_ctx* ctx, const char** names, float* ms, int cap);
/* Pippenger shape chosen for an m-term MSM: window bits c, number of windows, and the number of
 * group operations (mixed adds + full adds + doublings) the kernels execute for it. */
int typlonk_msm_plan(typlonk_ctx* ctx, size_t m, uint32_t* window_bits, uint32_t* n_windows, uint64_t* group_ops);

/* Self-test of the device's field inversion (the per-point `into_affine` of kzg/src/lib.rs:50 and kzg/src/srs.rs:20 is
 * one Fq inversion): `count` pseudo-random and edge residues, lazily reduced up to 8p, inverted by the divsteps routine
 * the kernels use and by the Fermat ladder a^(p-2); *mismatches = results that differ (or fail x * x^-1 = 1),
 * *max_rounds = the largest number of 30-divstep rounds any call ran (proven bound: 37). */
int typlonk_selftest_fq_inv(typlonk_ctx* ctx, uint64_t seed, size_t count, uint64_t* mismatches, uint32_t* max_rounds);

const char* typlonk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TYPLONK_H */
