// Host-callable launchers of the HIP kernels; each is defined in the .hip unit that holds the kernel,
// so the units compile independently (and in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ff.hpp"
#include "ntt_args.hpp"

namespace ty {

constexpr int SCAN_PER_BLOCK = 2048;  // 256 threads x 8
// Resident SRS points are 96 B of payload (x || y packed) on a 128-B stride: every gather of the
// accumulate kernel then touches exactly one 128-B line instead of 1.75 on average (PMC FETCH_SIZE
// of msm_accum_kernel: 3.9 GB -> see profiles/).
constexpr int PT_WORDS = 32;
// a heavy bucket of `count` entries (more than `cap`, msm_sort.hip) is cut into tasks of this many entries, one wavefront
// each: 4 entries per lane before the six butterfly steps, 16 for the very large buckets (where the tasks are many and
// the butterfly is the cost: 2^20 equal scalars 20 -> 5.4 ms; with a few thousand heavy entries the short tasks win)
__host__ __device__ inline uint32_t msm_task_len(uint32_t count) { return count >= 65536u ? 1024u : 256u; }
constexpr uint32_t MSM_TASK_LEN_MIN = 256;
constexpr int MSM_HEAVY_GRID = 1024;   // workgroups of msm_heavy_kernel (four wavefronts each; grid-strided over the tasks)

// arguments of the quotient pointwise kernel (quotient.hip): every array holds 4n coset evaluations
struct QuotientArgs {
    const Fr* wires[3];
    const Fr* z;
    const Fr* sel[5];  // q_l q_r q_o q_m q_c
    const Fr* sigma[3];
    const Fr* pi;
    const Fr* l0;
    const Fr* w_lo;    // powers of w_{4n}: lo[e & mask] * hi[e >> w_h]
    const Fr* bx_hi;   // beta * g * w_hi[j]  (launch_fr_scale per call: beta is per proof)
    Fr* out;
    uint64_t n4;
    uint32_t w_h;
    uint32_t k0_is_one;  // cosets[0] = 1 (plonk/src/lib.rs: the first wire's coset is H itself): no multiplication
    Fr alpha, alpha2, beta, gamma;
    Fr k[3];
    Fr zh_inv[4];      // 1 / (g^n * i^k - 1), k = index mod 4
};
void launch_fr_fill(Fr* out, uint64_t n, const Fr& value, hipStream_t s);
void launch_fr_scale(const Fr* in, uint64_t n, const Fr& factor, Fr* out, hipStream_t s);
void launch_quotient_pointwise(const QuotientArgs& a, hipStream_t s);

// grand product (plonk_ops.hip)
struct GrandProductArgs {
    const Fr* wires[3];   // witness column evaluations over the domain (n each)
    const Fr* sigma[3];   // sigma evaluations (n each)
    const Fr* w_lo;       // powers of w_n (two-level table)
    const Fr* w_hi;
    Fr* num;
    Fr* den;
    uint64_t n;
    uint32_t w_h;
    Fr beta, gamma;
    Fr kbeta[3];          // k_i * beta
};
void launch_gp_terms(const GrandProductArgs& a, hipStream_t s);
void launch_product_scan(const Fr* in, uint64_t n, int reverse, Fr* block_scratch, Fr* out, hipStream_t s);
void launch_fr_inv(const Fr* in, Fr* out, hipStream_t s);   // out[0] = in[0]^-1 on the device (one wavefront)
void launch_gp_finish(const Fr* nprefix, const Fr* dsuffix, const Fr* inv_total /* device */, uint64_t n, Fr* z, hipStream_t s);

// open(): y = p(z) and q = (p - y)/(X - z) in one suffix scan; m <= 2^22 coefficients, q may be null
void launch_open(const Fr* c, uint64_t m, const Fr& z, Fr* q, Fr* blocks, Fr* y, hipStream_t s);
struct LincombArgs {
    const Fr* poly[12];
    Fr scalar[12];
    Fr constant;
    Fr* out;
    uint64_t n;
    uint32_t terms;
};
void launch_lincomb(const LincombArgs& a, hipStream_t s);
// count <= 8 openings (quotients[k] != null: y and (p - y)/(X - z)) or evaluations (null) of polynomials of m <= 2^22
// coefficients, each at z0 (zsel[k] = 0) or z1 (1), in three launches; blocks: 8 * ceil(m/2048) Fr; ys[k]: device scalar
void launch_open_multi(const Fr* const* polys, Fr* const* quotients, Fr* const* ys, const uint8_t* zsel, uint32_t count,
                       uint64_t m, const Fr& z0, const Fr& z1, Fr* blocks, hipStream_t s);

// can this device run the 4096-element tiles (144 KiB of dynamic LDS)?  false: the opt-in was refused -> 1024-element tiles
bool ntt_big_tiles_available();
void launch_ntt_pass(const NttPassArgs& a, unsigned blocks, unsigned threads, size_t lds_bytes, hipStream_t s);
// the same pass on 9 x 30-bit limbs: every table in `a` is a full table in the 2^270 domain, 36 B of LDS per element
void launch_ntt_pass30(const NttPassArgs& a, unsigned blocks, unsigned threads, size_t lds_bytes, hipStream_t s);
void launch_ntt_full_table(const Fr* lo, const Fr* hi, uint32_t h, uint64_t S, uint64_t n, Fr* out, hipStream_t s);

void launch_convert_points(uint32_t* pts, const uint8_t* inf, uint64_t n, hipStream_t s);
void launch_msm_digits(const Fr* scalars, uint64_t m, uint32_t c, uint32_t W, uint32_t top_v, uint32_t* keys,
                       uint32_t* counts, hipStream_t s);
void launch_scan(const uint32_t* counts, uint64_t n, uint32_t* block_sums, uint32_t* offsets, uint32_t* cursor,
                 hipStream_t s);
uint32_t msm_segsort_blocks(uint64_t m);  // workgroups of the level-1 passes for an m-term MSM
uint32_t msm_sched_words();               // words of the bucket-schedule counters (hist514 of the calls below)
// the whole segmented bucket sort of one chunk, bucket schedule (order[]) included; beside_accum: the sort runs beside an
// accumulation (raised wavefront priority); blk_cnt: nseg * nblk words (the staged
// level-1 scatter's per-workgroup count rows; NULL = direct scatter), seg_start: 2 * nseg words
void launch_msm_segsort(const Fr* scalars, uint64_t m, uint32_t c, uint32_t W, uint32_t top_v, uint32_t hb,
                        uint32_t ibits, uint32_t tlen, uint32_t nsets, uint32_t* blk_hist, uint32_t* blk_base,
                        uint32_t* scan_scratch, uint32_t* blk_cnt, uint32_t* seg_start, uint32_t* entries, uint32_t* counts,
                        uint32_t* offsets, uint32_t* sorted, uint32_t cap, uint32_t* hist514, uint32_t* heavy, uint32_t* tasks,
                        uint32_t* order, bool centred, int staged_mode, uint32_t l1_threads, bool beside_accum, hipStream_t s);
// windows of the signed c-bit digit decomposition.  Scalars are canonical (< r < 2^255): the top window holds
// t = bits - c (W0 - 1) bits, W0 = ceil(bits / c), and a digit <= 2^t cannot exceed 2^(c-1) (no carry out of it) unless
// t = c.  Centred scalars (|k| <= (r - 1)/2 < 2^254) have one bit less: c = 17 -> 15 windows instead of 16.
TY_HD uint32_t msm_windows(uint32_t c, bool centred) {
    const uint32_t bits = centred ? 254u : 255u;
    const uint32_t w0 = (bits + c - 1) / c;
    return w0 + ((bits - c * (w0 - 1)) == c ? 1u : 0u);
}
// zbuf: srs_tables_scratch_bytes(len, T) bytes of scratch (one denominator per table entry)
size_t srs_tables_scratch_bytes(uint64_t len, uint32_t T);
void launch_srs_tables(uint32_t* pts, uint32_t* zbuf, uint64_t len, uint32_t c, uint32_t T, hipStream_t s);
void launch_bucket_order(const uint32_t* counts, const uint32_t* offsets, uint32_t n, uint32_t cap, uint32_t* hist514,
                         uint32_t* order, uint32_t* heavy, uint32_t* tasks, hipStream_t s);
void launch_msm_heavy(const uint32_t* points, const uint32_t* sorted, const uint32_t* hist516, uint32_t* heavy,
                      const uint32_t* tasks, uint32_t* partial, uint32_t* buckets, hipStream_t s);
void launch_msm_scatter(const uint32_t* keys, uint64_t m, uint64_t total, uint32_t* cursor, uint32_t* sorted,
                        hipStream_t s);
// lanes = 1, 2, 4, 8, 16: lanes per bucket (msm_accum_kernel / msm_accum_ml_kernel<L>); the first `split` buckets of the
// schedule (the larger ones) get `lanes`, the others lanes / 2 (split >= nbuckets: all get `lanes`)
void launch_msm_accum(const uint32_t* points, const uint32_t* offsets, const uint32_t* sorted, const uint32_t* order,
                      uint32_t nbuckets, uint32_t cap, bool init, uint32_t lanes, uint32_t split, uint32_t* buckets,
                      hipStream_t s);

// Row/column bucket reduction (msm_reduce.hip).  A bucket set of B = 2^c1 buckets is read as a grid of
// R = 2^ch rows x C = 2^cl columns, k = hi * C + lo.  Every bucket weight splits as w(k) = wr(hi) + wc(lo):
//   v <= cl : wr = hi << (cl - v),            wc = (lo >> v) + 1
//   v >  cl : wr = (hi >> (v - cl)) + 1,      wc = 0
// (v = virtual-copy bits of the set: top_v for the last set of a plain MSM, else 0), so
// sum_k w(k) B_k = sum_hi wr(hi) Rsum_hi + sum_lo wc(lo) Csum_lo needs only PLAIN sums of buckets plus two
// weighted sums of R + C points.  Those are returned as bit planes -- out[(set*2 + kind)*RC_NB + b] =
// sum of the row (kind 0) / column (kind 1) sums whose weight has bit b set, rows WITHOUT their common
// factor 2^shift -- and the host finishes with one Horner pass over powers of two.
constexpr uint32_t RC_NB = 16;
struct RcShape {
    uint32_t nsets, c1, ch, cl;
    uint32_t lhc, llc;  // log2 rows per column partial / columns per row partial
    uint32_t top_v;
};
TY_HD uint32_t rc_set_v(const RcShape& sh, uint32_t set) { return set + 1 == sh.nsets ? sh.top_v : 0u; }
// bits of the row / column weights of a set and the rows' common shift
TY_HD void rc_bits(const RcShape& sh, uint32_t set, uint32_t* nbr, uint32_t* nbc, uint32_t* shift) {
    const uint32_t v = rc_set_v(sh, set);
    if (v <= sh.cl) {
        *nbr = sh.ch;
        *nbc = sh.cl - v + 1;
        *shift = sh.cl - v;
    } else {
        const uint32_t d = v - sh.cl;
        *nbr = (d <= sh.ch ? sh.ch - d : 0u) + 1;
        *nbc = 0;
        *shift = 0;
    }
}
TY_HD uint32_t rc_weight(const RcShape& sh, uint32_t set, uint32_t kind, uint32_t idx) {
    const uint32_t v = rc_set_v(sh, set);
    if (v <= sh.cl) return kind == 0 ? idx : (idx >> v) + 1;
    return kind == 0 ? (idx >> (v - sh.cl)) + 1 : 0u;
}
void launch_msm_rc_reduce(const uint32_t* buckets, const RcShape& sh, uint32_t* pb, uint32_t* pa, uint32_t* sums,
                          uint32_t* bitsum, uint32_t* out, hipStream_t s);
// the same bit planes in two launches (msm_reduce.hip); needs cl, ch >= 6; prow, pcol: nsets << (c1 - 6) points each
bool msm_rc2_ok(const RcShape& sh);
void launch_msm_rc2_reduce(const uint32_t* buckets, const RcShape& sh, uint32_t* prow, uint32_t* pcol, uint32_t* out,
                           uint32_t log_waves, hipStream_t s);
void launch_msm_rc_combine(const uint32_t* planes, const RcShape& sh, uint32_t* set_sums, hipStream_t s);
// comb: the fixed-base table of G (srs_comb_bytes() bytes, filled once by launch_srs_comb)
// divsteps inversion against the Fermat ladder on threads * per_thread residues; out: {mismatches, max rounds, calls}
void launch_fq_inv_selftest(uint64_t seed, uint32_t threads, uint32_t per_thread, uint32_t* out, hipStream_t st);
size_t srs_comb_bytes();
void launch_srs_comb(uint32_t* comb, hipStream_t st);
void launch_srs_generate(const Fr& s, uint64_t start, uint64_t n, const uint32_t* comb, uint32_t* pts, hipStream_t st);

}  // namespace ty
