// Shared state of the host driver of libtyplonk_hip.so (implementation of include/typlonk.h for gfx950).
//   ctx.hip       context, workspaces, profiling events, device vectors
//   ntt_host.hip  NTT planning (tables, pass decomposition) + typlonk_ntt_*
//   msm_host.hip  MSM staging (sort / accumulate / reduce launches, lanes of a batch, host finish) + SRS + typlonk_msm_*
//   comm.hip      RCCL exchange behind the C ABI
//   prover.hip    quotient, grand product, openings, the prover rounds, typlonk_prove
// There is deliberately no CPU compute fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#pragma once
#include "../../include/typlonk.h"
#include "g1.hpp"
#include "g1_host64.hpp"
#include "launch.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace tyh {
using namespace ty;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct SrsEntry {
    uint32_t* d_points = nullptr;  // len * PT_WORDS u32 (96 B payload on a 128-B stride), identity = (0,0)
    size_t len = 0;
    uint32_t table_c = 0, table_T = 0;  // fixed-base tables 2^(c t) P_i at index t*len + i (typlonk_srs_precompute)
    bool table_centred = false;         // table_T counts the windows of CENTRED scalars (launch.hpp, msm_windows)
    // typlonk_srs_set_shard: this entry holds bases [shard_first, shard_first + len) of a total_len-point SRS
    size_t shard_first = 0, total_len = 0;
    size_t total() const { return total_len ? total_len : len; }
    // part [off, off + ml) of an m-term MSM that falls into this entry
    void local_range(size_t m, size_t* off, size_t* ml) const {
        const size_t lo = std::min(shard_first, m), hi = std::min(shard_first + len, m);
        *off = lo;
        *ml = hi - lo;
    }
};

struct Table {
    Fr* d = nullptr;
    size_t n = 0;
    uint64_t last_use = 0;  // typlonk_ctx::table_tick at the last lookup (coset tables are evicted LRU)
};

// Per-circuit constants of the quotient: the 4n coset evaluations of q_l q_r q_o q_m q_c, sigma_0..2
// and L0 (9 vectors).  They are fixed per CompiledCircuit (plonk/src/lib.rs:19-35), so they are
// transformed once instead of on every proof.
struct CircuitEntry {
    Fr* ext = nullptr;     // 9 * 4n : coset evaluations of q_l q_r q_o q_m q_c sigma_0..2 L0
    Fr* coef = nullptr;    // 8 * n  : coefficient copies of the selectors and sigmas (linearisation, sigma(zeta))
    Fr* sig_ev = nullptr;  // 3 * n  : sigma evaluations over the domain (grand product)
    uint32_t log_n = 0;
};

struct ProfStage {
    const char* name;
    hipEvent_t a, b;
};

// Everything one in-flight MSM needs: grow-only device workspaces, the stream it was enqueued on and
// the pinned landing zone of its window sums.  A context owns one per lane, so that a batch of independent
// MSMs (prove() issues them in groups: 3 wire commitments, 5-6 openings, 3 quotient slices --
// plonk/src/proof.rs:107-110, 147-175, 181) can overlap one MSM's host-side finish (window combine, affine
// normalisation) and kernel tail with the next one's sort + accumulate.
constexpr size_t HOST_WIN_POINTS = 32 * 2 * RC_NB;  // up to 32 bucket sets x {rows, columns} x RC_NB bit planes

// Outputs of the bucket sort of one chunk of terms; a workspace owns two sets so that the sort of chunk k + 1 can run
// (on the workspace's side stream) while chunk k is being accumulated.
struct SortBufs {
    DevBuf keys, sorted, counts, offsets, cursor, blocksums, order, ohist, blk_hist, blk_base, blk_cnt, seg_start, heavy, tasks, hpart;
    std::vector<DevBuf*> all() {
        return {&keys, &sorted, &counts, &offsets, &cursor, &blocksums, &order, &ohist, &blk_hist, &blk_base, &blk_cnt, &seg_start,
                &heavy, &tasks, &hpart};
    }
};
constexpr int MSM_MAX_CHUNKS = 8;
constexpr size_t MSM_CHAIN_MIN_TERMS = (size_t)1 << 17;   // typlonk_ctx::msm_chain
constexpr size_t MSM_FOUR_LANES_BELOW = (size_t)1 << 17;  // typlonk_ctx::msm_inflight

struct MsmWs {
    SortBufs sb[2];
    DevBuf buckets, part_a, part_b, rc_sums, rc_bits, rc_out;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;     // sorts of the chunks after the first (created on first use)
    hipEvent_t ev_in = nullptr, ev_sorted[MSM_MAX_CHUNKS] = {}, ev_acc[MSM_MAX_CHUNKS] = {};
    uint32_t* host_wins = nullptr;  // pinned, HOST_WIN_POINTS x 48 words
    bool pending = false;
    uint32_t W = 0, c = 0;
    bool rc = false;                // row/column reduction: host_wins holds bit planes (launch.hpp)
    RcShape rcs{};
    uint64_t* out_xy = nullptr;
    uint8_t* out_inf = nullptr;
};

constexpr size_t COMM_REC = 13;  // 12 limbs + the infinity flag, one u64 each: 104 bytes per point and rank
struct Comm {
    void* comm = nullptr;        // ncclComm_t (comm.hip)
    int rank = 0, world = 0;
    uint64_t* d_send = nullptr;  // cap records
    uint64_t* d_recv = nullptr;  // world * cap records
    uint64_t* h_buf = nullptr;   // pinned: cap records out + world * cap records back
    size_t cap = 0;
};

}  // namespace tyh

struct typlonk_buf {
    ty::Fr* d = nullptr;
    size_t n = 0;
};

// Environment switches read by typlonk_init (the ones the parity tests parametrise; every combination gives the same bits):
//   TYPLONK_MSM_INFLIGHT  MSMs of a batch in flight at once (1..4; default: by SRS length, MsmQueue)
//   TYPLONK_MSM_CHAIN     0 | 1: the lanes of a batch run free / chain their accumulations (default: by term count)
//   TYPLONK_MSM_CHUNKS    chunks of a stand-alone MSM (0 = by length)
//   TYPLONK_MSM_LANES     lanes per bucket of the accumulation (1, 2, 4, 8, 16; 0 = by bucket load)
//   TYPLONK_MSM_SCATTER   staged | direct: level 1 of the bucket sort stages its runs in the LDS / writes entry by entry
//   TYPLONK_MSM_L1_THREADS 256 | 512: workgroup size of the sort's level-1 passes (default: by whether the sort runs beside an accumulation)
//   TYPLONK_MSM_SORT_PRIO 0 | 1: raised wavefront priority + 256-thread level 1 for the sorts of a stand-alone MSM's overlapped chunks
//   TYPLONK_MSM_REDUCE    rc2 | rc4: force the two- / four-launch row/column bucket reduction
//   TYPLONK_NTT_FR30      0 | 1 | 2: the 9 x 30-bit butterflies never / where they measure faster / always
//   TYPLONK_NTT_BIG       0 | 1 | 2: the two-pass 2^20 plan (4096-element tiles) never / where it measures faster / always
//   TYPLONK_PROVER_NTT_BATCH 0 | 1 | 2 | 3: round 1 transforms its columns one by one (each commitment submitted as soon as its
//                         polynomial exists) / as one batched transform per group (ntt_run_batch) / the first alone, the rest
//                         batched / interpolations one by one, coset extensions batched
//   TYPLONK_PROVER_PIPE   0 | 1: round 3's nine commitments queued as in rounds 1-4 / behind one fence (prover_round3_core)
// (TYPLONK_RCCL_LIB, read by comm.hip, names the RCCL library to load.)
struct typlonk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<uint32_t, tyh::SrsEntry> srs;
    uint32_t next_srs = 1;
    std::map<uint32_t, tyh::CircuitEntry> circuits;
    uint32_t next_circuit = 1;
    tyh::DevBuf srs_comb;          // fixed-base comb of G (typlonk_srs_generate, srs_gen.hip)
    bool srs_comb_ready = false;   // set once the comb's build kernel has run to completion
    // MSM
    tyh::DevBuf scal;
    static constexpr int MSM_LANES = 4;
    tyh::MsmWs ws[MSM_LANES];
    hipStream_t lane[MSM_LANES] = {};  // lanes 1.. of typlonk_msm_g1_batch* (lane 0 = stream)
    int msm_inflight = 0;           // MSMs of a batch in flight at once (1..MSM_LANES; 0 = by SRS length: 3, where the
                                    // accumulations are chained and a fourth lane only adds a sort competing for the same
                                    // slots (profiles/r03_msm_chain_ab.txt), 4 below 2^17 points, where they run free)
    hipEvent_t lane_evt[MSM_LANES] = {};  // "scalars ready" marks (MsmQueue::submit)
    hipEvent_t batch_fence = nullptr;   // typlonk_msm_g1_batch_devptr: everything queued before the call (MsmQueue::fence)
    // Queued MSMs (a batch, a prover round) run their accumulations ONE AFTER THE OTHER, whichever lanes they are on: the
    // kernel fills every SIMD by itself, so two of them side by side only take turns -- while the sort of the next MSM
    // and the reduction of the previous one, latency-bound kernels, do hide beside an accumulation.  Without the chain
    // the lanes move in lockstep (four sorts together, four accumulations together, four reductions together) and
    // nothing overlaps (profiles/r03_msm_batch_timeline_before.txt).
    hipEvent_t accum_chain = nullptr;
    bool accum_chain_live = false;
    // A short accumulation does NOT fill the chip: below 2^17 terms the lanes run free (profiles/r04_ab_chain_by_size.txt:
    // prove() at 2^16 5.75 -> 5.09 ms, 2^14 4.18 -> 3.41, 2^12 3.24 -> 3.05).  Free lanes also win inside a 2^17 / 2^18
    // proof (7.4 -> 7.2, 11.8 -> 11.4 ms) but lose in a pure batch of nine 2^17-term MSMs -- an 8-way shard's round,
    // 0.419 -> 0.464 ms per MSM -- so the switch sits below the shard size; from 2^19 on the chain is never worse and
    // 2^20 needs it.  -1 = by term count (MSM_CHAIN_MIN_TERMS), 0 / 1 = TYPLONK_MSM_CHAIN.
    int msm_chain = -1;
    int prover_ntt_batch = 0;      // TYPLONK_PROVER_NTT_BATCH: round 1's interpolations / coset extensions 0 = one by one (each commitment
                                   // submitted as soon as its polynomial exists), 1 = one batched transform per group, 2 = the first
                                   // column alone, the rest batched, 3 = interpolations one by one, extensions batched.  Measured
                                   // (profiles/r06_ab_prover_ntt_batch.txt): inside a proof the batched forms LOSE 0.3-0.5 ms at 2^20 --
                                   // they delay a commitment's start by the other columns' transforms, and the transforms were
                                   // already hidden beside the commitments' sorts; the batched entry point pays where nothing
                                   // runs beside it (typlonk_circuit_load, a caller's interpolate groups)
    bool prover_pipe = true;       // TYPLONK_PROVER_PIPE (A/B switch of the round-5 queueing fix, prover_round3_core)
    bool prover_pinned_slots = true;  // the prover's evaluation slots in pinned host memory (TYPLONK_PROVER_FETCH=0: device slots + copy)
    tyh::Fr* eval_slots_host = nullptr;  // 16 pinned, device-visible result slots (prover_ops_tmp)
    int msm_chunks = 0;            // chunks of a stand-alone MSM (0 = choose by length)
    int msm_first_pct = 0;         // share of the terms in the first chunk, per cent (0 = equal chunks)
    int msm_lanes = 0;             // lanes per bucket of the accumulation (0 = choose by bucket load)
    bool msm_scatter_staged = true;  // TYPLONK_MSM_SCATTER=direct: level 1 of the bucket sort writes every entry straight to global
                                   // memory (the rounds 1-5 form, the A/B reference) instead of staging runs in the LDS
    int msm_l1_threads = 0;        // TYPLONK_MSM_L1_THREADS = 256 | 512: threads per workgroup of the sort's level-1 passes (staged form);
                                   // 0 = 512 for a sort that has the chip to itself (histogram 18.5 -> 14 us, scatter 49 -> 32 us per
                                   // 2^19 terms), 256 for an overlapped chunk's (msm_enqueue)
    bool msm_sort_prio = true;     // TYPLONK_MSM_SORT_PRIO=0: the overlapped chunks' sorts get neither the raised wavefront priority
                                   // nor the 256-thread shape (the A/B reference)
    bool msm_rc4 = false;          // always the four-launch row/column reduction
    bool msm_rc2_force = false;    // the two-launch form for every bucket-set size
    int msm_rc2_logw = 10;         // log2 wavefronts of the two-launch form's first launch (TYPLONK_MSM_RC2_LOGW)
    // NTT
    tyh::DevBuf ntt_scratch, ntt_io, quot_ext, quot_tab, ops_tmp, prover_mem;
    bool prover_busy = false;  // one proof in flight per context (the arena above is shared)
    int prover_rounds_active = 0;  // > 0 while a typlonk_prover_round* call is running (ProverRound)
    std::map<std::string, tyh::Table> tables;
    uint64_t table_tick = 0;
    // tables keyed by a caller-chosen coset shift ("cs:" keys) are a cache, not a plan: at most this many distinct
    // (direction, size, shift) groups / bytes stay resident, the least recently used group is dropped first
    static constexpr size_t COSET_GROUPS_MAX = 8;
    static constexpr size_t COSET_BYTES_MAX = (size_t)3 << 30;
    int ntt_fr30 = 1;              // 0 = the 8 x 32-bit kernel everywhere, 1 = the default policy (9 x 30-bit butterflies, fr30.hpp, for
                                   // every inverse transform and for forward ones up to 2^NTT_FR30_FWD_MAX_LOG), 2 = 30-bit everywhere
    int ntt_big = 1;               // TYPLONK_NTT_BIG: the two-pass 2^20 plan on 4096-element tiles 0 = never, 1 = where it measures
                                   // faster (ntt_run_batch), 2 = for every 2^20 transform outside a prover round
    // profiling
    bool profiling = false;
    bool prof_light = false;       // typlonk_set_profiling(ctx, 2): only the bucket-accumulation launches are bracketed
    std::vector<tyh::ProfStage> prof;
    std::vector<hipEvent_t> event_pool;  // timing events of finished stages, reused by the next call
    std::vector<std::pair<const char*, float>> prof_result;
    tyh::Comm comm;                // typlonk_comm_init: RCCL communicator of this rank (world = 0: none)
};

namespace tyh {

// smallest heavy-bucket threshold (entries one accumulation thread may sum; msm_enqueue)
constexpr uint32_t MSM_CAP_MIN = 32;
// full (one-multiplication) twiddle / coset tables are built up to this many entries (512 MB); above, two-level tables
constexpr uint32_t NTT_FULL_MAX_LOG = 24;
// default policy (TYPLONK_NTT_FR30 = 1): forward transforms take the 9 x 30-bit kernel up to this size, inverse ones always
// (their closing factor is free); set from the same-box A/B of the two kernels (ntt_host.hip, ntt_run)
constexpr uint32_t NTT_FR30_FWD_MAX_LOG = 32;

int fail(typlonk_ctx* c, int code, const std::string& msg);

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return tyh::fail(ctx, _e == hipErrorOutOfMemory ? TYPLONK_ERR_OOM : TYPLONK_ERR_HIP,          \
                             std::string(#expr) + ": " + hipGetErrorString(_e));                          \
    } while (0)

int ensure(typlonk_ctx* ctx, DevBuf& b, size_t bytes);   // grow-only device workspace
void release(DevBuf& b);

// Frees the device allocations registered with it unless dismiss()ed: setup functions allocate several
// buffers and may fail half-way (HIPCHK returns early).
struct DevGuard {
    std::vector<void*> ptrs;
    void* add(void* p) {
        ptrs.push_back(p);
        return p;
    }
    void dismiss() { ptrs.clear(); }
    ~DevGuard() {
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
};
struct ProverRound {
    typlonk_ctx* ctx;
    explicit ProverRound(typlonk_ctx* c) : ctx(c) { ++c->prover_rounds_active; }
    ~ProverRound() { --ctx->prover_rounds_active; }
};
// Stage events are per call: composite calls switch them off for their inner calls and restore on every exit path.
struct ProfilingOff {
    typlonk_ctx* ctx;
    bool saved;
    explicit ProfilingOff(typlonk_ctx* c) : ctx(c), saved(c->profiling) { c->profiling = false; }
    ~ProfilingOff() { ctx->profiling = saved; }
};

// ---- profiling ------------------------------------------------------------------------------
struct StageTimer {
    typlonk_ctx* ctx;
    bool on;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    hipStream_t st;
    StageTimer(typlonk_ctx* c, const char* n, hipStream_t s = nullptr)
        : ctx(c), on(c->profiling && (!c->prof_light || strcmp(n, "msm_accum") == 0)), name(n), st(s ? s : c->stream) {
        if (on) {
            a = take();
            b = take();
            (void)hipEventRecord(a, st);
        }
    }
    // events are recycled through the context (creating two per stage and call costs more than recording them)
    hipEvent_t take() {
        hipEvent_t e = nullptr;
        if (!ctx->event_pool.empty()) {
            e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            (void)hipEventCreate(&e);
        }
        return e;
    }
    ~StageTimer() {
        if (on) {
            (void)hipEventRecord(b, st);
            ctx->prof.push_back({name, a, b});
        }
    }
};
void prof_begin(typlonk_ctx* ctx);
void prof_collect(typlonk_ctx* ctx);

// ---- ntt_host.hip ---------------------------------------------------------------------------------------------------
Fr fr_domain_root(uint32_t log_n);      // generator of the size-2^log_n domain (ark-poly Radix2EvaluationDomain::group_gen)
Fr fr_domain_root_inv(uint32_t log_n);
Fr fr_inv_pow2(uint32_t log_n);         // (2^log_n)^-1 (size_inv)
Fr fr_from_u64(uint64_t x);
// two-level power tables of `base` covering exponents < 2^log_len: lo[j] = base^j (j < 2^h), hi[j] = hi_scale * base^(j 2^h)
int get_pow2l(typlonk_ctx* ctx, const std::string& key, const Fr& base, const Fr& hi_scale, uint32_t log_len, Table* lo,
              Table* hi, uint32_t* h_out);
// short_in / n_valid: the transform of a ZERO-PADDED vector -- the first pass reads short_in[0, n_valid) and takes every
// element beyond as zero (no padded copy, no reads of zeros or of their coset factors); the result lands in d_data.
int ntt_run(typlonk_ctx* ctx, Fr* d_data, uint32_t log_n, int inverse, const uint64_t* coset_shift, bool sync = true,
            const Fr* short_in = nullptr, uint64_t n_valid = ~0ull);
// `count` transforms of one size / direction / coset, pass by pass in shared launches (d_data[v] in place; short_in: NULL or
// one zero-padded source per vector, all n_valid long)
int ntt_run_batch(typlonk_ctx* ctx, Fr* const* d_data, size_t count, uint32_t log_n, int inverse, const uint64_t* coset_shift,
                  bool sync = true, const Fr* const* short_in = nullptr, uint64_t n_valid = ~0ull);

// ---- msm_host.hip ---------------------------------------------------------------------------------------------------
void write_affine_out(const G1Affine& a, uint64_t out_xy[12], uint8_t* out_inf);   // internal affine -> the C-ABI's arkworks form
int msm_validate(typlonk_ctx* ctx, uint32_t srs_id, size_t m, const SrsEntry** srs);
// d_scalars points at coefficient 0 of the m-term vector (ptr_is_local: at the first coefficient of this
// entry's share instead); an SRS shard sums only its own index range
// h_scalars != NULL (with ptr_is_local): the local share still lives on the host and is copied chunk by chunk beside the kernels
int msm_run(typlonk_ctx* ctx, uint32_t srs_id, const Fr* d_scalars, size_t m, uint64_t out_xy[12], uint8_t* out_inf,
            bool ptr_is_local = false, const uint64_t* h_scalars = nullptr);
// count independent MSMs over the same SRS, up to MSM_LANES in flight (separate workspaces/streams)
int msm_batch(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m, size_t count, uint64_t* out_xy,
              uint8_t* out_inf);
int msm_finish(typlonk_ctx* ctx, MsmWs& ws);

// Asynchronous MSM submissions over one SRS (a prover round, or typlonk_msm_g1_batch_devptr).
//   submit()    puts an MSM on the next lane of [lane_lo, lanes): the lane first waits for everything queued on the
//               context's stream so far -- the kernels that produce the scalars -- and a lane that still holds an
//               unfinished MSM is finished first (the only way submit() blocks).  Work queued on the context's stream
//               AFTER the call runs concurrently with the MSM.  Lane 0 is the context's stream itself.
//   wait_all()  finishes every MSM in flight (host-side window combine + affine normalisation of each).
// out_xy / out_inf of an MSM must stay valid until it has been finished.
struct MsmQueue {
    typlonk_ctx* ctx;
    const SrsEntry* srs;
    int lanes, lane_lo, next;
    hipEvent_t fence = nullptr;  // set: the lanes wait for this mark instead of for everything on the context's stream
    MsmQueue(typlonk_ctx* c, const SrsEntry* s, int first_lane = 0);
    // keep the context's stream (lane 0) free for other work when there is another lane to use
    void set_first_lane(int l);
    int submit(const Fr* d_scalars, size_t m, uint64_t* out_xy, uint8_t* out_inf, bool standalone = false);
    int wait_all();
};

// ---- comm.hip -------------------------------------------------------------------------------------------------------
void comm_release(typlonk_ctx* ctx);
// every point <- sum over the ranks of that rank's point (all-gather + fold in rank order); local_rc: the status of the
// local work the points come from -- a failed rank still joins the collective, with flagged records
int comm_fold(typlonk_ctx* ctx, uint64_t* xy, uint8_t* inf, size_t count, int local_rc = TYPLONK_OK);
// does this MSM / prover call need the fold?  (an SRS shard on a context with a communicator)
bool comm_folds(typlonk_ctx* ctx, uint32_t srs_id);

}  // namespace tyh
