// libtyplonk_hip.so -- implementation of include/typlonk.h for gfx950.
// Host driver: context/workspace management, NTT planning + twiddle tables, MSM staging and the
// final (host) window combine.  There is deliberately no CPU compute fallback: without a HIP
// device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "../../include/typlonk.h"
#include "g1.hpp"
#include "g1_host64.hpp"
#include "launch.hpp"
#include "transcript.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen'ed on first use (typlonk_comm_*)

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace ty;

namespace {

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
// the pinned landing zone of its window sums.  A context owns two, so that a batch of independent
// MSMs (prove() issues them in groups: 3 wire commitments, 5-6 openings, 3 quotient slices --
// plonk/src/proof.rs:107-110, 147-175, 181) can overlap one MSM's host-side finish (window combine, affine
// normalisation) and kernel tail with the next one's sort + accumulate.
constexpr size_t HOST_WIN_POINTS = 32 * 2 * RC_NB;  // up to 32 bucket sets x {rows, columns} x RC_NB bit planes

// Outputs of the bucket sort of one chunk of terms; a workspace owns two sets so that the sort of chunk k + 1 can run
// (on the workspace's side stream) while chunk k is being accumulated.
struct SortBufs {
    DevBuf keys, sorted, counts, offsets, cursor, blocksums, order, ohist, blk_hist, blk_base, heavy, tasks, hpart;
    std::vector<DevBuf*> all() {
        return {&keys, &sorted, &counts, &offsets, &cursor, &blocksums, &order, &ohist, &blk_hist, &blk_base, &heavy, &tasks, &hpart};
    }
};
constexpr int MSM_MAX_CHUNKS = 8;

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

}  // namespace

struct typlonk_buf {
    Fr* d = nullptr;
    size_t n = 0;
};

namespace {
// RCCL entry points, resolved once per process.  The library is NOT linked: a single-GPU caller never loads it, and in
// a process that already holds a copy (PyTorch's) dlopen by SONAME returns that copy, which is bound to the same HIP
// runtime as this library there.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
RcclApi* rccl_api() {
    static RcclApi api;
    if (api.handle || !api.err.empty()) return &api;
    // TYPLONK_RCCL_LIB names the library to load (a deployment with its own RCCL build); otherwise the SONAME, which a
    // process that already holds a copy resolves to that copy
    const char* forced = getenv("TYPLONK_RCCL_LIB");
    std::string why = "not found";
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (forced && *forced) name = forced;
        api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
        if (const char* e = dlerror()) why = e;   // ONE call: dlerror() clears the message it returns
        if (forced && *forced) break;
    }
    if (!api.handle) {
        api.err = "cannot load librccl: " + why;
        return &api;
    }
    auto sym = [&](const char* n) -> void* {
        void* f = dlsym(api.handle, n);
        if (!f && api.err.empty()) api.err = std::string("librccl lacks ") + n;
        return f;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    return &api;
}
constexpr size_t COMM_REC = 13;  // 12 limbs + the infinity flag, one u64 each: 104 bytes per point and rank
struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    uint64_t* d_send = nullptr;  // cap records
    uint64_t* d_recv = nullptr;  // world * cap records
    uint64_t* h_buf = nullptr;   // pinned: cap records out + world * cap records back
    size_t cap = 0;
};
}  // namespace

struct typlonk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<uint32_t, SrsEntry> srs;
    uint32_t next_srs = 1;
    std::map<uint32_t, CircuitEntry> circuits;
    uint32_t next_circuit = 1;
    // MSM
    DevBuf scal;
    static constexpr int MSM_LANES = 4;
    MsmWs ws[MSM_LANES];
    hipStream_t lane[MSM_LANES] = {nullptr, nullptr, nullptr, nullptr};  // lanes 1.. of typlonk_msm_g1_batch* (lane 0 = stream)
    int msm_inflight = 3;           // MSMs of a batch in flight at once (TYPLONK_MSM_INFLIGHT, 1..MSM_LANES); with the accumulations
                                    // chained, a fourth lane only adds a sort competing for the same slots (profiles/r03_msm_chain_ab.txt)
    hipEvent_t lane_evt[MSM_LANES] = {nullptr, nullptr, nullptr, nullptr};  // "scalars ready" marks (MsmQueue::submit)
    hipEvent_t batch_fence = nullptr;   // typlonk_msm_g1_batch_devptr: everything queued before the call (MsmQueue::fence)
    // Queued MSMs (a batch, a prover round) run their accumulations ONE AFTER THE OTHER, whichever lanes they are on: the
    // kernel fills every SIMD by itself, so two of them side by side only take turns -- while the sort of the next MSM
    // and the reduction of the previous one, latency-bound kernels, do hide beside an accumulation.  Without the chain
    // the lanes move in lockstep (four sorts together, four accumulations together, four reductions together) and
    // nothing overlaps (profiles/r03_msm_batch_timeline_before.txt).  TYPLONK_MSM_CHAIN=0 switches it off.
    hipEvent_t accum_chain = nullptr;
    bool accum_chain_live = false;
    bool msm_chain = true;
    uint32_t msm_cap_min = 32;    // TYPLONK_MSM_CAP_MIN: smallest heavy-bucket threshold (round 2: 512)
    // NTT
    DevBuf ntt_scratch, ntt_io, quot_ext, quot_tab, ops_tmp, prover_mem;
    bool prover_busy = false;  // one proof in flight per context (the arena above is shared)
    std::map<std::string, Table> tables;
    uint64_t table_tick = 0;
    // tables keyed by a caller-chosen coset shift ("cs:" keys) are a cache, not a plan: at most this many distinct
    // (direction, size, shift) groups / bytes stay resident, the least recently used group is dropped first
    static constexpr size_t COSET_GROUPS_MAX = 8;
    static constexpr size_t COSET_BYTES_MAX = (size_t)3 << 30;
    // profiling
    bool profiling = false;
    std::vector<ProfStage> prof;
    std::vector<hipEvent_t> event_pool;  // timing events of finished stages, reused by the next call
    std::vector<std::pair<const char*, float>> prof_result;
    int msm_c_override = 0;
    // TYPLONK_PROVER_OVERLAP (A/B switch): bit 0 = coset transforms of a, b, c, PI in round 1, bit 1 = of Z in round 2,
    // bit 2 = first opening MSMs of round 3 submitted before the quotient
    int prover_overlap = 3;   // measured (profiles/r02_ab_prover_overlap.txt): bits 0-1 gain ~1 %, bit 2 loses ~1 %
    int msm_chunks = 0;            // TYPLONK_MSM_CHUNKS: chunks of a stand-alone MSM (0 = choose by length)
    bool msm_side_prio = false;    // TYPLONK_MSM_SIDE_PRIO=1: the side stream of the chunk sorts at the highest stream priority
                                   // (measured: no effect -- 2.61-2.65 ms either way, profiles/r03_side_prio.txt)
    bool msm_host_planes = true;   // TYPLONK_MSM_HOST_PLANES=0: bit planes go to device memory and are copied to the host
    bool msm_stagger = true;       // TYPLONK_MSM_STAGGER=0: the second chunk's sort runs beside the first one's (round-2 order)
    bool msm_lanes_split = true;   // TYPLONK_MSM_LANES_SPLIT=0: one lane count for every bucket
    int msm_lanes = 0;             // TYPLONK_MSM_LANES: lanes per bucket of the accumulation (0 = choose by bucket load)
    bool msm_rc4 = false;          // TYPLONK_MSM_REDUCE=rc4: always the four-launch row/column reduction (round-2 form)
    bool msm_rc2_force = false;    // TYPLONK_MSM_REDUCE=rc2: the two-launch form for every bucket-set size
    int prover_rounds_active = 0;  // > 0 while a typlonk_prover_round* call is running (ProverRound)
    bool ntt_big_tiles = true;     // TYPLONK_NTT_BIG=0: always 1024-element tiles (three passes at 2^20)
    bool ntt_full_tables = true;   // TYPLONK_NTT_FULL_TABLES=0: compose twiddles / coset powers from two-level tables
    int ntt_fr30 = 1;              // TYPLONK_NTT_FR30: 0 = off, 1 = 9 x 30-bit butterflies (fr30.hpp) up to 2^20 and for every inverse transform,
                                   // 2 = for every transform (the 4096-element tiles of 2^20 are then not used)
    uint32_t ntt_full_max_log = 24;
    int ntt_tile_log = 0;          // TYPLONK_NTT_TILE (measurement): tiles of the two-pass 2^20 transform, see ntt_run
    bool ntt_short_in = true;      // TYPLONK_NTT_SHORT_IN=0: the quotient's extensions copy + zero-fill a 4n buffer first
    bool ntt_direct = true;        // TYPLONK_NTT_DIRECT=0: the radix-4 groups all go through the LDS tile (staging copy in / out)
    bool ntt_radix4 = true;        // TYPLONK_NTT_RADIX=2: one LDS round trip per butterfly stage (the round-2 form)
    Comm comm;                     // typlonk_comm_init: RCCL communicator of this rank (world = 0: none)
    bool msm_legacy_sort = false;  // TYPLONK_MSM_SORT=atomic: per-entry global-atomic counting sort
    bool msm_tree_reduce = false;  // TYPLONK_MSM_REDUCE=running: running-sum + small-multiple reduction (first version)
};

namespace {

int fail(typlonk_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(ctx, _e == hipErrorOutOfMemory ? TYPLONK_ERR_OOM : TYPLONK_ERR_HIP,               \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                               \
    } while (0)

int ensure(typlonk_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes) return TYPLONK_OK;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return TYPLONK_OK;
}

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

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
    StageTimer(typlonk_ctx* c, const char* n, hipStream_t s = nullptr) : ctx(c), on(c->profiling), name(n), st(s ? s : c->stream) {
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

void prof_begin(typlonk_ctx* ctx) {
    for (auto& s : ctx->prof) {
        ctx->event_pool.push_back(s.a);
        ctx->event_pool.push_back(s.b);
    }
    ctx->prof.clear();
}

void prof_collect(typlonk_ctx* ctx) {
    if (!ctx->profiling) return;
    ctx->prof_result.clear();
    for (auto& s : ctx->prof) {
        float ms = 0.f;
        (void)hipEventSynchronize(s.b);
        (void)hipEventElapsedTime(&ms, s.a, s.b);
        ctx->prof_result.push_back({s.name, ms});
    }
    prof_begin(ctx);
}

// ---- host Fr helpers ---------------------------------------------------------------------------
Fr fr_root_of_unity_2_32() {
    // ark-bls12-381 FrParameters::TWO_ADIC_ROOT_OF_UNITY = 7^((r-1)/2^32), canonical value
    Fr c;
    const uint32_t limbs[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                               0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
    for (int i = 0; i < 8; ++i) c.v[i] = limbs[i];
    return fe_to_mont(c);
}

// generator of the size-2^log_n domain (ark-poly Radix2EvaluationDomain::group_gen)
Fr fr_domain_root(uint32_t log_n) {
    Fr w = fr_root_of_unity_2_32();
    for (uint32_t i = log_n; i < 32; ++i) w = fe_sqr(w);
    return w;
}
// its inverse (group_gen_inv) from the inverse of the 2^32-th root: squarings instead of a field inversion per call
Fr fr_domain_root_inv(uint32_t log_n) {
    Fr c;
    const uint32_t limbs[8] = {0x3cf19a78u, 0x0fb4d6e1u, 0xb566f833u, 0x6f67d4a2u, 0xa35d0168u, 0xed4f2f74u, 0x6e19c653u, 0x0538a6f6u};
    for (int i = 0; i < 8; ++i) c.v[i] = limbs[i];
    Fr w = fe_to_mont(c);
    for (uint32_t i = log_n; i < 32; ++i) w = fe_sqr(w);
    return w;
}
// (2^log_n)^-1 = ((r + 1) / 2)^log_n  (size_inv)
Fr fr_inv_pow2(uint32_t log_n) {
    Fr c;
    const uint32_t limbs[8] = {0x80000001u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};
    for (int i = 0; i < 8; ++i) c.v[i] = limbs[i];
    const Fr half = fe_to_mont(c);
    Fr x = Fr::one();
    for (uint32_t i = 0; i < log_n; ++i) x = fe_mul(x, half);
    return x;
}

Fr fr_from_u64(uint64_t x) {
    Fr c = Fr::zero();
    c.v[0] = (uint32_t)x;
    c.v[1] = (uint32_t)(x >> 32);
    return fe_to_mont(c);
}

int upload_table(typlonk_ctx* ctx, const std::string& key, const std::vector<Fr>& h, Table* out) {
    Table t;
    t.n = h.size();
    t.last_use = ++ctx->table_tick;
    HIPCHK(hipMalloc((void**)&t.d, h.size() * sizeof(Fr)));
    DevGuard g;
    g.add(t.d);
    HIPCHK(hipMemcpyAsync(t.d, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // h goes out of scope
    g.dismiss();
    ctx->tables[key] = t;
    *out = t;
    return TYPLONK_OK;
}

// powers table: out[j] = scale * base^j, j < n
int get_pow_table(typlonk_ctx* ctx, const std::string& key, const Fr& base, const Fr& scale, size_t n, Table* out) {
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        it->second.last_use = ++ctx->table_tick;
        *out = it->second;
        return TYPLONK_OK;
    }
    std::vector<Fr> h(n);
    Fr x = scale;
    for (size_t j = 0; j < n; ++j) {
        h[j] = x;
        x = fe_mul(x, base);
    }
    return upload_table(ctx, key, h, out);
}

// Full one-multiplication table built on the device from a two-level pair (launch_ntt_full_table); kept per
// context like every other table.  Sizes above TYPLONK_NTT_FULL_MAX_LOG (default 2^24 entries = 512 MB) are not
// built: *out stays empty and the kernel composes the factor from the two-level tables instead.
int get_full_table(typlonk_ctx* ctx, const std::string& key, const Table& lo, const Table& hi, uint32_t h, uint64_t S,
                   uint64_t n, Table* out) {
    *out = Table{};
    if (!ctx->ntt_full_tables || n > (1ull << ctx->ntt_full_max_log)) return TYPLONK_OK;
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        it->second.last_use = ++ctx->table_tick;
        *out = it->second;
        return TYPLONK_OK;
    }
    Table t;
    t.n = n;
    t.last_use = ++ctx->table_tick;
    hipError_t e = hipMalloc((void**)&t.d, n * sizeof(Fr));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return TYPLONK_OK;  // no room: fall back to the two-level tables
    }
    launch_ntt_full_table(lo.d, hi.d, h, S, n, t.d, ctx->stream);
    {
        DevGuard g;
        g.add(t.d);
        HIPCHK(hipGetLastError());
        g.dismiss();
    }
    ctx->tables[key] = t;
    *out = t;
    return TYPLONK_OK;
}

std::string fr_hex(const Fr& f) {
    char buf[80];
    snprintf(buf, sizeof(buf), "%08x%08x%08x%08x%08x%08x%08x%08x", f.v[7], f.v[6], f.v[5], f.v[4], f.v[3], f.v[2],
             f.v[1], f.v[0]);
    return buf;
}

// two-level power tables of `base` covering exponents < 2^log_len:
//   lo[j] = base^j (j < 2^h),  hi[j] = hi_scale * base^(j * 2^h) (j < 2^(log_len-h))
int get_pow2l(typlonk_ctx* ctx, const std::string& key, const Fr& base, const Fr& hi_scale, uint32_t log_len,
              Table* lo, Table* hi, uint32_t* h_out) {
    const uint32_t h = (log_len + 1) / 2;
    *h_out = h;
    int rc = get_pow_table(ctx, key + ":lo", base, Fr::one(), (size_t)1 << h, lo);
    if (rc) return rc;
    Fr step = base;
    for (uint32_t i = 0; i < h; ++i) step = fe_sqr(step);
    return get_pow_table(ctx, key + ":hi", step, hi_scale, (size_t)1 << (log_len - h), hi);
}

// Coset tables are keyed by the caller's shift ("cs:<dir>:<log_n>:<shift hex>..."): a caller that varies the shift
// would otherwise grow HBM without bound.  Before a new group is built, drop least-recently-used groups until the
// cache is inside its limits (the quotient's generator 7 is looked up on every proof and therefore stays).
int evict_coset_tables(typlonk_ctx* ctx, const std::string& incoming_group, size_t incoming_bytes) {
    if (ctx->tables.count(incoming_group + ":lo")) return TYPLONK_OK;  // resident (the per-proof case)
    for (;;) {
        std::map<std::string, std::pair<uint64_t, size_t>> groups;  // group -> (last use, bytes)
        size_t bytes = 0;
        for (const auto& kv : ctx->tables) {
            if (kv.first.compare(0, 3, "cs:") != 0) continue;
            size_t cut = kv.first.find(':', kv.first.find(':', kv.first.find(':', 3) + 1) + 1);  // after the shift hex
            const std::string grp = kv.first.substr(0, cut);
            auto& g = groups[grp];
            g.first = std::max(g.first, kv.second.last_use);
            g.second += kv.second.n * sizeof(Fr);
            bytes += kv.second.n * sizeof(Fr);
        }
        if (groups.count(incoming_group)) return TYPLONK_OK;  // already resident: nothing new is built
        if (groups.size() < typlonk_ctx::COSET_GROUPS_MAX && bytes + incoming_bytes <= typlonk_ctx::COSET_BYTES_MAX)
            return TYPLONK_OK;
        if (groups.empty()) return TYPLONK_OK;
        std::string victim;
        uint64_t oldest = ~0ull;
        for (const auto& g : groups)
            if (g.second.first < oldest) {
                oldest = g.second.first;
                victim = g.first;
            }
        // kernels still reading the victim's tables: they may sit on a stream the context has since been moved away
        // from (typlonk_set_stream) or come from an un-synchronised *_devptr call, so the rare eviction waits for the
        // whole device rather than for the current stream only
        HIPCHK(hipDeviceSynchronize());
        for (auto it = ctx->tables.begin(); it != ctx->tables.end();) {
            if (it->first.compare(0, victim.size(), victim) == 0 &&
                (it->first.size() == victim.size() || it->first[victim.size()] == ':')) {
                (void)hipFree(it->second.d);
                it = ctx->tables.erase(it);
            } else {
                ++it;
            }
        }
    }
}

// big = 4096-element tiles (1024 threads, 128 KiB of LDS): sub-transforms of 2^10 points with 4 adjacent columns, so a
// 2^20 transform needs two passes instead of three -- one load/store round and one inter-pass twiddle fewer.  Only 2^20:
// that is 256 tiles, one per CU; 2^17..2^19 would leave most of the chip idle (measured: 2^19 0.097 -> 0.132 ms) and
// 2^21.. do not fit (2^11-point sub-transforms x 4 columns = 256 KiB).
void split_log(uint32_t L, uint32_t ks[4], uint32_t* P, bool big = false) {
    uint32_t p = L <= 10 ? 1 : (L <= 16 ? 2 : (L <= 24 ? 3 : 4));
    if (big && L == 20) p = 2;
    *P = p;
    for (uint32_t i = 0; i < p; ++i) ks[i] = L / p + (i < L % p ? 1 : 0);
}

uint32_t ilog2_u64(uint64_t x) {
    uint32_t r = 0;
    while ((1ull << (r + 1)) <= x) ++r;
    return r;
}

// short_in / n_valid: the transform of a ZERO-PADDED vector -- the first pass reads short_in[0, n_valid) and takes every
// element beyond as zero (no padded copy, no reads of zeros or of their coset factors); the result lands in d_data.
int ntt_run(typlonk_ctx* ctx, Fr* d_data, uint32_t log_n, int inverse, const uint64_t* coset_shift, bool sync = true,
            const Fr* short_in = nullptr, uint64_t n_valid = ~0ull) {
    if (log_n > 32) return fail(ctx, TYPLONK_ERR_DOMAIN, "log_n > 32 (Fr two-adicity)");
    if (!d_data) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null data");
    prof_begin(ctx);
    if (log_n == 0) {
        prof_collect(ctx);
        return TYPLONK_OK;  // size-1 transform is the identity (g^0 = 1, n^-1 = 1)
    }
    const uint64_t N = 1ull << log_n;
    uint32_t ks[4], P;
    // (not inside a prover round: there the 144 KiB workgroups crowd out the LDS of the MSM lanes' sort kernels running
    // beside them -- prove() 38.1 -> 38.3 ms in the same-box A/B)
    // (forward only: an inverse 2^20 transform is 4 % faster in three passes of the 30-bit kernel, 0.148 against 0.153 ms)
    bool big = ctx->ntt_big_tiles && log_n == 20 && (!inverse || ctx->ntt_fr30 == 0) && ctx->prover_rounds_active == 0 && ntt_big_tiles_available();
    if (big && ctx->ntt_fr30 == 2 && ctx->ntt_full_tables) big = false;  // 36 B per element: 4096 of them do not fit
    // log2 of the tile capacity; TYPLONK_NTT_TILE=11 tries 2048-element tiles (two columns, two workgroups per CU) for the
    // two-pass 2^20 transform
    // TYPLONK_NTT_TILE: 12 = 4096-element tiles in both passes, 11 = 2048 in both, 0 (default) = 4096 in the first
    // (strided: four columns make 128-byte runs) and 2048 in the last (rows are contiguous, and two workgroups per CU
    // overlap each other's load / compute / store phases: 0.0775 -> 0.070 ms, profiles/r03_ntt_2_20_tiles.txt)
    const uint32_t cap_first = big ? (ctx->ntt_tile_log == 11 ? 11u : 12u) : 10u;
    const uint32_t cap_last = big ? (ctx->ntt_tile_log == 12 ? 12u : 11u) : 10u;
    split_log(log_n, ks, &P, big);
    const std::string dir = inverse ? "i" : "f";

    Fr* scratch = nullptr;
    if (P >= 2) {
        int rc = ensure(ctx, ctx->ntt_scratch, N * sizeof(Fr));
        if (rc) return rc;
        scratch = (Fr*)ctx->ntt_scratch.p;
    }

    // measured: the 9 x 30-bit kernel is 9-12 % faster up to 2^19; at 2^20 the two-pass 4096-element
    // tiles of the 8 x 32 kernel win, and from 2^21 on the 36-B LDS elements cost a workgroup per CU (3 instead of 4)
    // and a forward transform pays one extra reducing multiplication per element: mode 1 (default) stops at 2^19
    // Round 3 (radix-4 groups in both kernels, profiles/r03_ntt_fr30_modes.txt): an INVERSE transform is 4-9 % faster on the
    // 30-bit kernel at every size (its n^-1 / coset factor closes the last pass for free), a forward one from 2^21 on is
    // not; at 2^20 a forward transform takes the two-pass big tiles when they are allowed, the 30-bit kernel otherwise
    const bool want30 = ctx->ntt_fr30 != 0 && ctx->ntt_full_tables && !big &&
                        (ctx->ntt_fr30 == 2 || log_n <= 20 || inverse);

    // coset / scaling tables (the full tables of the 8 x 32 kernel are not built when the other kernel will run)
    Table pre_lo{}, pre_hi{}, post_lo{}, post_hi{}, scale{}, pre_full{}, post_full{};
    uint32_t pre_h = 0, post_h = 0;
    Fr n_inv = Fr::one();
    if (inverse) n_inv = fr_inv_pow2(log_n);
    if (coset_shift) {
        Fr g;
        memcpy(g.v, coset_shift, sizeof(g.v));
        if (!inverse) {
            const std::string key = "cs:f:" + std::to_string(log_n) + ":" + fr_hex(g);
            int rc = evict_coset_tables(ctx, key, (size_t)N * sizeof(Fr));
            if (rc) return rc;
            rc = get_pow2l(ctx, key, g, Fr::one(), log_n, &pre_lo, &pre_hi, &pre_h);
            if (rc) return rc;
            if (!want30 && (rc = get_full_table(ctx, key + ":full", pre_lo, pre_hi, pre_h, 0, N, &pre_full))) return rc;
        } else {
            const std::string key = "cs:i:" + std::to_string(log_n) + ":" + fr_hex(g);
            const bool resident = ctx->tables.count(key + ":lo") && ctx->tables.count(key + ":hi");
            const Fr gi = resident ? Fr::one() : fe_inv(g);  // only a table build needs the value
            int rc = evict_coset_tables(ctx, key, (size_t)N * sizeof(Fr));
            if (rc) return rc;
            rc = get_pow2l(ctx, key, gi, n_inv, log_n, &post_lo, &post_hi, &post_h);
            if (rc) return rc;
            if (!want30 && (rc = get_full_table(ctx, key + ":full", post_lo, post_hi, post_h, 0, N, &post_full))) return rc;
        }
    } else if (inverse) {
        int rc = get_pow_table(ctx, "ninv:" + std::to_string(log_n), Fr::one(), n_inv, 1, &scale);
        if (rc) return rc;
    }

    // The 9 x 30-bit kernel (fr30.hpp) multiplies with R' = 2^270: its tables carry an extra factor 2^14 and every one
    // of them must exist as a full table; if one cannot be built (size, memory) the transform runs on the 8 x 32 kernel.
    Table sub30[4]{}, tw30[4]{}, pre30{}, post30{}, scale30{};
    bool f30 = want30;
    if (f30) {
        const Fr c14 = fr_from_u64(1u << 14);
        int rc = TYPLONK_OK;
        if (coset_shift) {
            Fr g;
            memcpy(g.v, coset_shift, sizeof(g.v));
            const std::string key = std::string("cs:") + (inverse ? "i:" : "f:") + std::to_string(log_n) + ":" + fr_hex(g) + ":30";
            Table lo, hi;
            uint32_t h = 0;
            if (!inverse) {
                if ((rc = get_pow2l(ctx, key, g, c14, log_n, &lo, &hi, &h))) return rc;
                if ((rc = get_full_table(ctx, key + ":full", lo, hi, h, 0, N, &pre30))) return rc;
                f30 = pre30.d != nullptr;
            } else {
                const bool resident = ctx->tables.count(key + ":lo") && ctx->tables.count(key + ":hi");
                if ((rc = get_pow2l(ctx, key, resident ? Fr::one() : fe_inv(g), fe_mul(n_inv, c14), log_n, &lo, &hi, &h))) return rc;
                if ((rc = get_full_table(ctx, key + ":full", lo, hi, h, 0, N, &post30))) return rc;
                f30 = post30.d != nullptr;
            }
        } else if (inverse) {
            if ((rc = get_pow_table(ctx, "ninv30:" + std::to_string(log_n), Fr::one(), fe_mul(n_inv, c14), 1, &scale30))) return rc;
        }
        uint64_t rl = N;
        for (uint32_t p = 0; p < P && f30; ++p) {
            const uint32_t k = ks[p];
            const uint64_t M = 1ull << k;
            const Fr w = inverse ? fr_domain_root_inv(k) : fr_domain_root(k);
            if ((rc = get_pow_table(ctx, "sub30:" + dir + ":" + std::to_string(k), w, c14, (size_t)std::max<uint64_t>(M / 2, 1), &sub30[p])))
                return rc;
            if (p + 1 < P) {
                const uint32_t lrow = ilog2_u64(rl);
                const Fr wr = inverse ? fr_domain_root_inv(lrow) : fr_domain_root(lrow);
                Table lo, hi;
                uint32_t h = 0;
                if ((rc = get_pow2l(ctx, "tw30:" + dir + ":" + std::to_string(lrow), wr, c14, lrow, &lo, &hi, &h))) return rc;
                if ((rc = get_full_table(ctx, "tw30:" + dir + ":" + std::to_string(lrow) + ":full:" + std::to_string(k), lo, hi, h,
                                         rl / M, rl, &tw30[p])))
                    return rc;
                f30 = tw30[p].d != nullptr;
            }
            rl /= M;
        }
    }

    uint64_t row_len = N;  // length of the rows the current pass works inside
    uint64_t rows = 1;
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t k = ks[p];
        const uint64_t M = 1ull << k;
        const bool last = (p + 1 == P);
        NttPassArgs a{};
        a.radix4 = ctx->ntt_radix4 ? (ctx->ntt_direct ? 2u : 1u) : 0u;
        a.k = k;
        a.last = last ? 1 : 0;
        a.S = row_len / M;
        a.row_len = row_len;
        // sub-transform twiddles w_M^e
        {
            const Fr w = inverse ? fr_domain_root_inv(k) : fr_domain_root(k);
            Table t;
            int rc = get_pow_table(ctx, "sub:" + dir + ":" + std::to_string(k), w, Fr::one(), (size_t)std::max<uint64_t>(M / 2, 1), &t);
            if (rc) return rc;
            a.sub_tw = t.d;
        }
        uint32_t logT;
        if (!last) {
            const uint32_t lrow = ilog2_u64(row_len);
            const Fr w = inverse ? fr_domain_root_inv(lrow) : fr_domain_root(lrow);
            Table lo, hi;
            int rc = get_pow2l(ctx, "tw:" + dir + ":" + std::to_string(lrow), w, Fr::one(), lrow, &lo, &hi, &a.tw_h);
            if (rc) return rc;
            a.tw_lo = lo.d;
            a.tw_hi = hi.d;
            Table full;
            if (!f30) {
                rc = get_full_table(ctx, "tw:" + dir + ":" + std::to_string(lrow) + ":full:" + std::to_string(k), lo, hi, a.tw_h,
                                    a.S, row_len, &full);
                if (rc) return rc;
            }
            a.tw_full = full.d;
            logT = std::min<uint32_t>(cap_first - k, ilog2_u64(a.S));
        } else {
            const uint64_t N1 = 1ull << ks[0];
            a.N1 = (P == 1) ? 1 : N1;
            a.Q = (P <= 2) ? 1 : rows / N1;
            a.N2 = (P >= 3) ? (1ull << ks[1]) : 1;
            a.N3 = (P == 4) ? (1ull << ks[2]) : 1;
            a.out_stride = N / M;
            logT = (P == 1) ? 0 : std::min<uint32_t>(cap_last - k, ks[0]);
            a.post_lo = post_lo.d;
            a.post_hi = post_hi.d;
            a.post_h = post_h;
            a.post_full = post_full.d;
            a.scale = scale.d;
        }
        a.logT = logT;
        if (p == 0) {
            a.pre_lo = pre_lo.d;
            a.pre_hi = pre_hi.d;
            a.pre_h = pre_h;
            a.pre_full = pre_full.d;
        }
        a.n_valid = ~0ull;
        if (p == 0 && short_in) {
            a.n_valid = n_valid;
        }
        if (P == 1) {
            a.in = short_in ? short_in : d_data;
            a.out = d_data;
        } else if (p == 0) {
            a.in = short_in ? short_in : d_data;
            a.out = scratch;
        } else if (!last) {
            a.in = scratch;
            a.out = scratch;
        } else {
            a.in = scratch;
            a.out = d_data;
        }
        const uint64_t E = M << logT;
        const uint64_t blocks = N / E;
        if (f30) {
            a.sub_tw = sub30[p].d;
            a.tw_full = tw30[p].d;
            a.tw_lo = a.tw_hi = nullptr;
            a.pre_lo = a.pre_hi = a.post_lo = a.post_hi = nullptr;
            a.pre_full = p == 0 ? pre30.d : nullptr;
            a.post_full = last ? post30.d : nullptr;
            a.scale = last ? scale30.d : nullptr;
        }
        const size_t lds = (size_t)(E + std::max<uint64_t>(M / 2, 1)) * (f30 ? 36 : sizeof(Fr));
        const unsigned threads = big ? (unsigned)std::max<uint64_t>(E / 4, 64) : 256u;
        {
            static const char* names[4] = {"ntt_pass1", "ntt_pass2", "ntt_pass3", "ntt_pass4"};
            StageTimer st(ctx, names[p]);
            if (f30) launch_ntt_pass30(a, (unsigned)blocks, threads, lds, ctx->stream);
            else launch_ntt_pass(a, (unsigned)blocks, threads, lds, ctx->stream);
        }
        HIPCHK(hipGetLastError());
        rows *= M;
        row_len /= M;
    }
    if (sync || ctx->profiling) HIPCHK(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    return TYPLONK_OK;
}

// ---- MSM ------------------------------------------------------------------------------------
void msm_shape(typlonk_ctx* ctx, size_t m, uint32_t* c_out, uint32_t* w_out) {
    uint32_t lg = 0;  // ceil(log2 m)
    while (((size_t)1 << lg) < m) ++lg;
    // measured on MI355X (tools/sweep_c.py): the best window is c ~ ceil(log2 m) clamped to [8, 16];
    // the bucket reduction is a fixed ~50-operation dependent chain whatever c is, so small MSMs want
    // many small buckets (short accumulate chains) rather than few windows
    int c = (int)lg;
    if (c < 8) c = 8;
    if (c > 16) c = 16;
    if (ctx && ctx->msm_c_override) c = ctx->msm_c_override;
    *c_out = (uint32_t)c;
    *w_out = msm_windows((uint32_t)c, false);
}

// Shape of the two-level (segmented) counting sort for an m-term MSM with c-bit windows: hb high bucket bits pick the
// segment, the low lb <= 8 bits are sorted in LDS; a level-1 entry packs [i : ibits][j : 4 in table mode][sign][low : lb]
// into 32 bits.  ok = the segmented sort can handle it (otherwise: plain MSMs use the atomic sort, table mode is not
// available).
struct SegShape {
    uint32_t ibits = 0;
    int hb = 0;
    uint64_t nseg = 0, nblk = 0, nmat = 0;
    bool ok = false;
};
SegShape msm_seg_shape(size_t m, uint32_t c, uint32_t W, uint32_t nsets, bool tables) {
    SegShape sh;
    uint32_t lgm = 0;
    while (((uint64_t)1 << lgm) < m) ++lgm;
    sh.ibits = tables ? std::max<uint32_t>(lgm, 1) : 23;
    const int jbits = W > 16 ? 5 : 4;   // table mode: the window index travels in the level-1 entry
    const int lb_max = tables ? std::min<int>(8, 32 - (int)sh.ibits - jbits - 1) : 8;
    int hb = std::max<int>((int)c - 1 - lb_max, tables ? 0 : (int)lgm - 13);
    sh.hb = std::max(0, std::min<int>(hb, (int)c - 1));
    sh.nseg = (uint64_t)nsets << sh.hb;
    sh.nblk = msm_segsort_blocks(m);
    sh.nmat = sh.nseg * sh.nblk;
    sh.ok = m <= (1u << 23) && lb_max >= 1 && sh.nseg * 4 <= 64 * 1024 && sh.nmat < (1ull << 31) && (!tables || W <= 32) &&
            (uint64_t)W * m < (1ull << 31);
    return sh;
}
// can a full-length MSM over a len-point SRS run in table mode with c-bit windows?  (the longest MSM is the worst case)
bool msm_table_shape_ok(size_t len, uint32_t c, uint32_t T) { return msm_seg_shape(len, c, T, 1, true).ok; }

// internal affine -> the C-ABI's arkworks form
void write_affine_out(const G1Affine& a, uint64_t out_xy[12], uint8_t* out_inf) {
    uint32_t w[12];
    if (a.is_inf()) {
        // ark-ec GroupAffine::zero(): x = 0, y = 1 (Montgomery one, R = 2^384), infinity = true
        memset(out_xy, 0, 6 * sizeof(uint64_t));
        fq30_to_ark(fq30_one(), w);
        memcpy(out_xy + 6, w, sizeof(w));
        *out_inf = 1;
    } else {
        fq30_to_ark(a.x, w);
        memcpy(out_xy, w, sizeof(w));
        fq30_to_ark(a.y, w);
        memcpy(out_xy + 6, w, sizeof(w));
        *out_inf = 0;
    }
}

// Launch every kernel of one m-term MSM (m > 0, validated by the caller) on `stream` using workspace
// `ws`, ending with the asynchronous copy of the W window sums into ws.host_wins.
int msm_enqueue(typlonk_ctx* ctx, MsmWs& ws, hipStream_t stream, const SrsEntry& srs, const Fr* d_scalars, size_t m,
                uint64_t* out_xy, uint8_t* out_inf, bool standalone) {
    ws.stream = stream;
    if (!ws.host_wins) HIPCHK(hipHostMalloc((void**)&ws.host_wins, HOST_WIN_POINTS * 192));
    uint32_t c, W;
    msm_shape(ctx, m, &c, &W);
    // fixed-base tables: every window reads its own pre-shifted copy of the base, so all windows share
    // one bucket set (plus a separate set for a thin top window) and no cross-window doublings remain
    // (typlonk_srs_precompute refuses shapes the table-mode sort cannot handle; the check here keeps a plain MSM
    // possible should one slip through)
    const bool tables = srs.table_T != 0 && m >= srs.len / 4 && srs.len <= (1u << 23) && !ctx->msm_legacy_sort &&
                        msm_seg_shape(m, srs.table_c, srs.table_T, 1, true).ok;
    if (tables) {
        c = srs.table_c;
        W = srs.table_T;
    }
    const bool centred = tables && srs.table_centred;
    const uint32_t B = 1u << (c - 1);
    // top window: t scalar bits -> 2^t digits, spread over 2^top_v virtual bucket copies
    const uint32_t t_bits = (centred ? 254u : 255u) - c * (W - 1);
    const uint32_t top_v = (t_bits >= c - 1) ? 0u : (c - 1 - t_bits);
    // table mode: ONE bucket set for all windows -- the top window's digits d <= 2^t go to the shared
    // buckets d - 1 with their true weight (no virtual copies).  Balanced when t is large (c = 20: t = 15);
    // for a thin top window the heavy-bucket tasks keep it correct, just slower.
    const uint32_t nsets = tables ? 1u : W;
    const uint32_t digit_v = tables ? 0u : top_v;
    const uint64_t nb = (uint64_t)nsets * B;
    const uint64_t nb_used = nb;
    if ((uint64_t)W * m >= (1ull << 31)) return fail(ctx, TYPLONK_ERR_LENGTH, "MSM too large for 32-bit entry indices");
    const uint32_t L = std::min<uint32_t>(MSM_SEG, B);
    const uint32_t npw = B / L;
    const uint32_t nodes = nsets * npw;
    const uint32_t scan_blocks = (uint32_t)((nb + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);

    // Chunks of terms.  A stand-alone MSM (nothing else in flight to hide behind) is cut into chunks that all add into
    // the SAME buckets: while chunk k is accumulated on the MSM's stream, chunk k + 1 is sorted on the workspace's side
    // stream, so only the first chunk's sort (and the last one's reduction) stay exposed.  Later chunks start from the
    // stored buckets (192 B read + written per bucket and chunk -- noise next to the additions).  Bit-identical
    // results: group addition is commutative and the output is the canonical affine point.
    uint32_t nch = 1;
    if (standalone && !ctx->msm_legacy_sort) {
        // measured (tools/msm_chunks.py, profiles/r02_msm_chunks.jsonl): the overlapped sort is not free -- it competes
        // with the accumulation for issue slots -- and chunks of ~2^19 terms are the best grain: 2 chunks at 2^20
        // (2.76 -> 2.68 ms), 4 at 2^21 (5.09 -> 4.81), 8 at 2^22 (9.92 -> 8.99); below 2^20 one chunk wins
        nch = ctx->msm_chunks ? (uint32_t)ctx->msm_chunks
                              : (m >= (1u << 20) ? (uint32_t)std::min<size_t>(m >> 19, MSM_MAX_CHUNKS) : 1u);
        while (nch > 1 && m / nch < 4096) --nch;
    }
    const size_t step = (m + nch - 1) / nch;
    hipStream_t s = ws.stream;
    int rc;
    if (nch > 1) {
        if (!ws.side) {
            // the side stream carries the sorts of the chunks after the first: short, latency-bound kernels beside an
            // accumulation that fills every wavefront slot.  A high stream priority (TYPLONK_MSM_SIDE_PRIO=1) was tried to
            // get them dispatched as slots retire: no measurable effect, so the default stays a plain stream
            int lo = 0, hi = 0;
            if (ctx->msm_side_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
                HIPCHK(hipStreamCreateWithPriority(&ws.side, hipStreamNonBlocking, hi));
            else
                HIPCHK(hipStreamCreateWithFlags(&ws.side, hipStreamNonBlocking));
        }
        if (!ws.ev_in) HIPCHK(hipEventCreateWithFlags(&ws.ev_in, hipEventDisableTiming));
        for (uint32_t k = 0; k < nch; ++k) {
            if (!ws.ev_sorted[k]) HIPCHK(hipEventCreateWithFlags(&ws.ev_sorted[k], hipEventDisableTiming));
            if (!ws.ev_acc[k]) HIPCHK(hipEventCreateWithFlags(&ws.ev_acc[k], hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ws.ev_in, s));  // the scalars (and whatever produced them) are ordered on s
        HIPCHK(hipStreamWaitEvent(ws.side, ws.ev_in, 0));
    }
    if ((rc = ensure(ctx, ws.buckets, nb * 192))) return rc;
    if ((rc = ensure(ctx, ws.part_a, (size_t)nodes * 192))) return rc;
    if ((rc = ensure(ctx, ws.part_b, (size_t)nodes * 192))) return rc;
    uint32_t* buckets = (uint32_t*)ws.buckets.p;
    uint32_t* pa = (uint32_t*)ws.part_a.p;
    uint32_t* pb = (uint32_t*)ws.part_b.p;

    for (uint32_t k = 0; k < nch; ++k) {
        const size_t off = (size_t)k * step;
        if (off >= m) break;
        const size_t mk = std::min(step, m - off);
        const Fr* sc = d_scalars + off;
        const uint32_t* pts = srs.d_points + off * PT_WORDS;  // chunk-local term index i -> base off + i (table t: + t*len)
        SortBufs& sb = ws.sb[k & 1];
        hipStream_t ss = (nch > 1 && k > 0) ? ws.side : s;   // the first sort has nothing to overlap with
        if (nch > 1 && k >= 2 && ss != s) HIPCHK(hipStreamWaitEvent(ss, ws.ev_acc[k - 2], 0));  // sb[k & 1] is free again
        // the first chunk's sort is the exposed one: the second chunk's sort starts behind it (it then has the whole first
        // accumulation to hide under) instead of beside it, where it doubled its time (profiles/r03_msm_2_20_timeline.txt)
        if (nch > 1 && k == 1 && ctx->msm_stagger) HIPCHK(hipStreamWaitEvent(ss, ws.ev_sorted[0], 0));
        const uint64_t total = (uint64_t)W * mk;
        if ((rc = ensure(ctx, sb.keys, total * 4))) return rc;
        if ((rc = ensure(ctx, sb.sorted, total * 4))) return rc;
        if ((rc = ensure(ctx, sb.counts, nb * 4))) return rc;
        if ((rc = ensure(ctx, sb.offsets, (nb + 1) * 4))) return rc;
        if ((rc = ensure(ctx, sb.cursor, nb * 4))) return rc;
        if ((rc = ensure(ctx, sb.blocksums, (size_t)scan_blocks * 4))) return rc;
        if ((rc = ensure(ctx, sb.order, nb * 4))) return rc;
        if ((rc = ensure(ctx, sb.ohist, 516 * 4))) return rc;
        // heavy-bucket splitting: cap = entries one thread may sum; at most total/cap heavy buckets/tasks
        // 8 x the mean, at least 32 (round 2: 4 x the mean, at least 512).  The accumulate kernel's thread walks a bucket's
        // first cap entries one after the other -- 6.7 us each when it is the last one running -- so a few buckets of 500
        // were a 3.4-ms tail; and the factor is 8 because table mode is not uniform: the top window's 2^t digits land
        // on the first 2^t buckets of the shared set (c = 20: 2.2 x the mean there), which 4 x the mean would already
        // turn into heavy buckets now and then (measured: +0.3 ms per 2^20 MSM for the extra launch's work)
        const uint32_t cap = (uint32_t)std::max<uint64_t>(ctx->msm_cap_min, 8 * ((total + nb_used - 1) / nb_used));
        const uint64_t max_tasks = total / MSM_TASK_LEN_MIN + total / cap + 2;   // sum of ceil(count / task length) over buckets > cap
        if ((rc = ensure(ctx, sb.heavy, max_tasks * 16))) return rc;
        if ((rc = ensure(ctx, sb.tasks, max_tasks * 12))) return rc;
        if ((rc = ensure(ctx, sb.hpart, max_tasks * 192))) return rc;
        uint32_t* keys = (uint32_t*)sb.keys.p;
        uint32_t* sorted = (uint32_t*)sb.sorted.p;
        uint32_t* counts = (uint32_t*)sb.counts.p;
        uint32_t* offsets = (uint32_t*)sb.offsets.p;
        uint32_t* cursor = (uint32_t*)sb.cursor.p;
        uint32_t* blocksums = (uint32_t*)sb.blocksums.p;

        // segmented sort shape: hb high bucket bits pick the segment, lb <= 8 low bits are sorted in LDS;
        // the level-1 entry packs [i : ibits][j : 4 in table mode][sign][low : lb] into 32 bits
        const SegShape seg = msm_seg_shape(mk, c, W, nsets, tables);
        const bool segsort = !ctx->msm_legacy_sort && seg.ok;
        if (tables && !segsort) return fail(ctx, TYPLONK_ERR_LENGTH, "table-mode MSM shape not supported");  // unreachable
        if (segsort) {
            if ((rc = ensure(ctx, sb.blk_hist, seg.nmat * 4))) return rc;
            if ((rc = ensure(ctx, sb.blk_base, (seg.nmat + 1) * 4))) return rc;
            if ((rc = ensure(ctx, sb.blocksums, (size_t)((seg.nmat + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK + scan_blocks + seg.nseg) * 4))) return rc;
            blocksums = (uint32_t*)sb.blocksums.p;
            StageTimer st(ctx, ss == s ? "msm_sort" : "msm_sort_overlapped", ss);
            launch_msm_segsort(sc, (uint64_t)mk, c, W, digit_v, (uint32_t)seg.hb, seg.ibits, tables ? (uint32_t)srs.len : 0u,
                               tables ? nsets : 0u, (uint32_t*)sb.blk_hist.p, (uint32_t*)sb.blk_base.p, blocksums, keys,
                               counts, offsets, sorted, cap, (uint32_t*)sb.ohist.p, (uint32_t*)sb.heavy.p,
                               (uint32_t*)sb.tasks.p, centred, ss);
        } else {
            {
                StageTimer st(ctx, "msm_digits", ss);
                HIPCHK(hipMemsetAsync(counts, 0, nb * 4, ss));
                launch_msm_digits(sc, (uint64_t)mk, c, W, top_v, keys, counts, ss);
            }
            {
                StageTimer st(ctx, "msm_scan", ss);
                launch_scan(counts, nb, blocksums, offsets, cursor, ss);
            }
            {
                StageTimer st(ctx, "msm_scatter", ss);
                launch_msm_scatter(keys, (uint64_t)mk, total, cursor, sorted, ss);
            }
        }
        {
            StageTimer st(ctx, "msm_order", ss);
            launch_bucket_order(counts, offsets, (uint32_t)nb_used, cap, (uint32_t*)sb.ohist.p, (uint32_t*)sb.order.p,
                                (uint32_t*)sb.heavy.p, (uint32_t*)sb.tasks.p, /*hist_done=*/segsort, ss);
        }
        if (ss != s) {
            HIPCHK(hipEventRecord(ws.ev_sorted[k], ss));
            HIPCHK(hipStreamWaitEvent(s, ws.ev_sorted[k], 0));
        } else if (nch > 1 && k == 0) {
            HIPCHK(hipEventRecord(ws.ev_sorted[0], s));
        }
        {
            // lanes per bucket: a short MSM over a small bucket set has few, long buckets -- spread each over L lanes so
            // that the launch fills the chip twice over (>= 2^18 threads: two rounds of two wavefronts per SIMD balance the
            // size-sorted schedule; one round leaves the SIMDs with the largest buckets 30 % behind), while a lane keeps >= 4 terms
            uint32_t lanes = 1;
            if (ctx->msm_lanes) {
                lanes = (uint32_t)ctx->msm_lanes;
            } else {
                const uint64_t mean = total / nb_used;
                while (lanes < 16 && nb_used * lanes < (1u << 18)) lanes *= 2;
                while (lanes > 1 && mean / lanes < 4) lanes /= 2;
            }
            // two size classes (the larger half of the buckets: `lanes`, the smaller half: lanes / 2) when lanes were
            // chosen from the load; TYPLONK_MSM_LANES forces one class, TYPLONK_MSM_LANES_SPLIT=0 switches the split off
            const uint32_t split = (lanes >= 2 && !ctx->msm_lanes && ctx->msm_lanes_split) ? (uint32_t)(nb_used / 2) : (uint32_t)nb_used;
            const bool chain = !standalone && ctx->msm_chain;
            if (chain && ctx->accum_chain_live) HIPCHK(hipStreamWaitEvent(s, ctx->accum_chain, 0));
            StageTimer st(ctx, "msm_accum", s);
            launch_msm_accum(pts, offsets, sorted, (const uint32_t*)sb.order.p, (uint32_t)nb_used, cap, /*init=*/k > 0, lanes,
                             split, buckets, s);
            launch_msm_heavy(pts, sorted, (const uint32_t*)sb.ohist.p, (uint32_t*)sb.heavy.p,
                             (const uint32_t*)sb.tasks.p, (uint32_t*)sb.hpart.p, buckets, s);
            if (chain) {
                if (!ctx->accum_chain) HIPCHK(hipEventCreateWithFlags(&ctx->accum_chain, hipEventDisableTiming));
                HIPCHK(hipEventRecord(ctx->accum_chain, s));
                ctx->accum_chain_live = true;
            }
        }
        if (nch > 1 && k + 2 < nch) HIPCHK(hipEventRecord(ws.ev_acc[k], s));
    }
    // reduce one group of `nwin` equally sized windows of `Bw` buckets starting at bucket `first`; the
    // window sums land in ws.host_wins[slot ...]
    auto reduce_group = [&](uint64_t first, uint32_t Bw, uint32_t nwin, uint32_t v_last, uint32_t slot) -> int {
        const uint32_t Lw = std::min<uint32_t>(MSM_SEG, Bw);
        const uint32_t npw_w = Bw / Lw;
        const uint32_t nodes_w = nwin * npw_w;
        uint32_t* cur = pa;
        uint32_t* other = pb;
        const uint32_t group = std::min<uint32_t>(64, npw_w);
        launch_msm_reduce(buckets + first * 48, Bw, Lw, nodes_w, group, c, nwin, v_last, cur, s);
        uint32_t n_in = npw_w / group;
        while (n_in > 1) {
            const uint32_t g2 = std::min<uint32_t>(64, n_in);
            launch_msm_fold(cur, nwin * n_in, g2, other, s);
            std::swap(cur, other);
            n_in /= g2;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ws.host_wins + (size_t)slot * 48, cur, (size_t)nwin * 192, hipMemcpyDeviceToHost, s));
        return TYPLONK_OK;
    };
    ws.rc = !ctx->msm_tree_reduce && c >= 3 && nsets <= 32;
    if (ws.rc) {
        RcShape& sh = ws.rcs;
        sh.nsets = nsets;
        sh.c1 = c - 1;
        sh.cl = (c - 1 + 1) / 2;
        sh.ch = c - 1 - sh.cl;
        sh.lhc = std::min<uint32_t>(3, sh.ch);
        sh.llc = std::min<uint32_t>(3, sh.cl);
        sh.top_v = digit_v;
        const uint64_t nrow = (uint64_t)nsets << (sh.c1 - sh.llc), ncol = (uint64_t)nsets << (sh.c1 - sh.lhc);
        if ((rc = ensure(ctx, ws.part_a, ncol * 192))) return rc;
        if ((rc = ensure(ctx, ws.part_b, nrow * 192))) return rc;
        if ((rc = ensure(ctx, ws.rc_sums, (((uint64_t)nsets << sh.ch) + ((uint64_t)nsets << sh.cl)) * 192))) return rc;
        if ((rc = ensure(ctx, ws.rc_bits, (uint64_t)nsets * 2 * RC_NB * 64 * 192))) return rc;
        if ((rc = ensure(ctx, ws.rc_out, (uint64_t)nsets * 2 * RC_NB * 192))) return rc;
        // one shared bucket set (table mode): the last kernel of the reduction writes its <= 32 plane points straight into
        // the pinned host landing zone (device-visible) -- no copy kernel between it and the host's wait
        uint32_t* planes_out = (nsets == 1 && ctx->msm_host_planes) ? ws.host_wins : (uint32_t*)ws.rc_out.p;
        StageTimer st(ctx, "msm_reduce", s);
        // two launches for small bucket sets, where the reduction is a latency chain; big sets are work-bound and the
        // four-launch form wastes fewer lanes (2^19 buckets: 0.39 ms against 0.49, profiles/r03_shard_variants.jsonl)
        if (!ctx->msm_rc4 && msm_rc2_ok(sh) && (ctx->msm_rc2_force || nb <= (1u << 17)))
            launch_msm_rc2_reduce(buckets, sh, (uint32_t*)ws.part_b.p, (uint32_t*)ws.part_a.p, planes_out, s);
        else
            launch_msm_rc_reduce(buckets, sh, (uint32_t*)ws.part_b.p, (uint32_t*)ws.part_a.p, (uint32_t*)ws.rc_sums.p,
                                 (uint32_t*)ws.rc_bits.p, planes_out, s);
        if (nsets > 1) {
            // plain MSM: per-set powers of two on the device, the host keeps its Horner over the windows
            uint32_t* set_sums = (uint32_t*)ws.part_a.p;  // the column partials are consumed by now
            launch_msm_rc_combine((const uint32_t*)ws.rc_out.p, sh, set_sums, s);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(ws.host_wins, set_sums, (size_t)nsets * 192, hipMemcpyDeviceToHost, s));
            ws.rc = false;
        } else {
            HIPCHK(hipGetLastError());
            if (planes_out != ws.host_wins)
                HIPCHK(hipMemcpyAsync(ws.host_wins, ws.rc_out.p, (size_t)nsets * 2 * RC_NB * 192, hipMemcpyDeviceToHost, s));
        }
    } else {
        StageTimer st(ctx, "msm_reduce", s);
        if ((rc = reduce_group(0, B, nsets, digit_v, 0))) return rc;
    }
    ws.pending = true;
    ws.W = nsets;
    ws.c = tables ? 0 : c;  // table mode: the set sums are simply added
    ws.out_xy = out_xy;
    ws.out_inf = out_inf;
    return TYPLONK_OK;
}

// Wait for an enqueued MSM and finish it on the host: sum_j 2^(c*j) * window_j (Horner from the top
// window), then the canonical affine form.
int msm_finish(typlonk_ctx* ctx, MsmWs& ws) {
    if (!ws.pending) return TYPLONK_OK;
    ws.pending = false;
    HIPCHK(hipStreamSynchronize(ws.stream));
    // host arithmetic on 6 x 64-bit words (g1_host64.hpp): a third of the time of the 13 x 30-bit limb code here
    namespace H = h64;
    auto out = [&](const H::Xyzz& acc) {
        if (H::xyzz_to_affine(acc, ws.out_xy)) *ws.out_inf = 0;
        else write_affine_out(G1Affine::inf(), ws.out_xy, ws.out_inf);
    };
    if (ws.rc) {
        // bit planes -> points by power of two: set j (offset c*j; 0 in table mode), rows carry 2^shift
        const RcShape& sh = ws.rcs;
        std::vector<H::Xyzz> pe(ws.c * sh.nsets + 2 * RC_NB + sh.cl + 2, H::inf());
        int top = -1;
        for (uint32_t j = 0; j < sh.nsets; ++j) {
            uint32_t nbr, nbc, shift;
            rc_bits(sh, j, &nbr, &nbc, &shift);
            for (uint32_t kind = 0; kind < 2; ++kind)
                for (uint32_t b = 0; b < (kind ? nbc : nbr); ++b) {
                    const H::Xyzz pt = H::xyzz_from_device(ws.host_wins + (size_t)((j * 2 + kind) * RC_NB + b) * 48);
                    if (H::is_inf(pt)) continue;
                    const uint32_t e = ws.c * j + b + (kind ? 0u : shift);
                    pe[e] = H::xyzz_add(pe[e], pt);
                    top = std::max(top, (int)e);
                }
        }
        H::Xyzz acc = H::inf();
        for (int e = top; e >= 0; --e) {
            if (!H::is_inf(acc)) acc = H::xyzz_dbl(acc);
            if (!H::is_inf(pe[e])) acc = H::xyzz_add(acc, pe[e]);
        }
        out(acc);
        return TYPLONK_OK;
    }
    H::Xyzz acc = H::inf();
    for (int j = (int)ws.W - 1; j >= 0; --j) {
        if (!H::is_inf(acc))
            for (uint32_t d = 0; d < ws.c; ++d) acc = H::xyzz_dbl(acc);
        acc = H::xyzz_add(acc, H::xyzz_from_device(ws.host_wins + (size_t)j * 48));
    }
    out(acc);
    return TYPLONK_OK;
}

int msm_validate(typlonk_ctx* ctx, uint32_t srs_id, size_t m, const SrsEntry** srs) {
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    if (m > it->second.total()) return fail(ctx, TYPLONK_ERR_LENGTH, "MSM length exceeds SRS length (kzg/src/lib.rs:43)");
    *srs = &it->second;
    return TYPLONK_OK;
}

// d_scalars points at coefficient 0 of the m-term vector (ptr_is_local: at the first coefficient of this
// entry's share instead); an SRS shard sums only its own index range
int msm_run(typlonk_ctx* ctx, uint32_t srs_id, const Fr* d_scalars, size_t m, uint64_t out_xy[12],
            uint8_t* out_inf, bool ptr_is_local = false) {
    if (!out_xy || !out_inf) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null output");
    const SrsEntry* srs = nullptr;
    int rc = msm_validate(ctx, srs_id, m, &srs);
    if (rc) return rc;
    prof_begin(ctx);
    size_t off, ml;
    srs->local_range(m, &off, &ml);
    if (ml == 0) {
        write_affine_out(G1Affine::inf(), out_xy, out_inf);
        prof_collect(ctx);
        return TYPLONK_OK;
    }
    if (!d_scalars) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null scalars");
    if (!ptr_is_local) d_scalars += off;
    m = ml;
    if ((rc = msm_enqueue(ctx, ctx->ws[0], ctx->stream, *srs, d_scalars, m, out_xy, out_inf, /*standalone=*/true))) return rc;
    if ((rc = msm_finish(ctx, ctx->ws[0]))) return rc;
    prof_collect(ctx);
    return TYPLONK_OK;
}

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
    MsmQueue(typlonk_ctx* c, const SrsEntry* s, int first_lane = 0)
        : ctx(c), srs(s), lanes(std::max(1, std::min<int>(c->msm_inflight, typlonk_ctx::MSM_LANES))), lane_lo(0), next(0) {
        set_first_lane(first_lane);
    }
    // keep the context's stream (lane 0) free for other work when there is another lane to use
    void set_first_lane(int l) {
        lane_lo = (l < lanes) ? l : 0;
        if (next < lane_lo) next = lane_lo;
    }
    int submit(const Fr* d_scalars, size_t m, uint64_t* out_xy, uint8_t* out_inf, bool standalone = false) {
        size_t off, ml;
        srs->local_range(m, &off, &ml);
        if (ml == 0) {
            write_affine_out(G1Affine::inf(), out_xy, out_inf);
            return TYPLONK_OK;
        }
        if (next >= lanes || next < lane_lo) next = lane_lo;
        const int l = next++;
        MsmWs& ws = ctx->ws[l];
        int rc = msm_finish(ctx, ws);
        if (rc) return rc;
        hipStream_t st = ctx->stream;
        if (l) {
            if (!ctx->lane[l]) HIPCHK(hipStreamCreateWithFlags(&ctx->lane[l], hipStreamNonBlocking));
            if (!ctx->lane_evt[l]) HIPCHK(hipEventCreateWithFlags(&ctx->lane_evt[l], hipEventDisableTiming));
            if (fence) {
                HIPCHK(hipStreamWaitEvent(ctx->lane[l], fence, 0));
            } else {
                HIPCHK(hipEventRecord(ctx->lane_evt[l], ctx->stream));
                HIPCHK(hipStreamWaitEvent(ctx->lane[l], ctx->lane_evt[l], 0));
            }
            st = ctx->lane[l];
        }
        return msm_enqueue(ctx, ws, st, *srs, d_scalars + off, ml, out_xy, out_inf, standalone);
    }
    int wait_all() {
        int rc = TYPLONK_OK;
        for (int l = 0; l < typlonk_ctx::MSM_LANES; ++l) {
            const int r = msm_finish(ctx, ctx->ws[l]);
            if (!rc) rc = r;
        }
        return rc;
    }
};

// count independent MSMs over the same SRS, up to MSM_LANES in flight (separate workspaces/streams)
int msm_batch(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m, size_t count,
              uint64_t* out_xy, uint8_t* out_inf) {
    if (!out_xy || !out_inf || !m || (!d_scalars && count)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    const SrsEntry* srs = nullptr;
    for (size_t k = 0; k < count; ++k) {
        int rc = msm_validate(ctx, srs_id, m[k], &srs);
        if (rc) return rc;
        if (m[k] && !d_scalars[k]) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null scalars");
    }
    if (!count) return TYPLONK_OK;
    prof_begin(ctx);
    ProfilingOff prof_off(ctx);  // stage events are per call
    MsmQueue q(ctx, srs);
    // all scalars exist when the call is made: the lanes wait for what is on the context's stream NOW, not for the
    // MSMs of this batch that lane 0 (the context's stream itself) receives in the meantime
    if (!ctx->batch_fence) HIPCHK(hipEventCreateWithFlags(&ctx->batch_fence, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ctx->batch_fence, ctx->stream));
    q.fence = ctx->batch_fence;
    int rc = TYPLONK_OK;
    for (size_t k = 0; k < count && !rc; ++k)
        rc = q.submit((const Fr*)d_scalars[k], m[k], out_xy + 12 * k, out_inf + k, /*standalone=*/count == 1);
    const int r = q.wait_all();
    return rc ? rc : r;
}

// ---- RCCL exchange ------------------------------------------------------------------------------------------------
#define NCCLCHK(expr)                                                                                                 \
    do {                                                                                                              \
        ncclResult_t _r = (expr);                                                                                     \
        if (_r != ncclSuccess)                                                                                        \
            return fail(ctx, TYPLONK_ERR_COMM, std::string(#expr) + ": " + rccl_api()->GetErrorString(_r));           \
    } while (0)

void comm_release(typlonk_ctx* ctx) {
    Comm& c = ctx->comm;
    if (c.comm) (void)rccl_api()->CommDestroy(c.comm);
    if (c.d_send) (void)hipFree(c.d_send);
    if (c.d_recv) (void)hipFree(c.d_recv);
    if (c.h_buf) (void)hipHostFree(c.h_buf);
    c = Comm{};
}

// The exchange buffers are allocated ONCE, by typlonk_comm_init (COMM_CAP records: more than the 9 points of a prover
// round): a fold never allocates, so no rank can fail locally between "decided to fold" and the collective and leave
// its peers waiting.  Longer point lists go through in pieces of COMM_CAP records (comm_fold).
constexpr size_t COMM_CAP = 32;
int comm_reserve(typlonk_ctx* ctx) {
    Comm& c = ctx->comm;
    HIPCHK(hipMalloc((void**)&c.d_send, COMM_CAP * COMM_REC * 8));
    HIPCHK(hipMalloc((void**)&c.d_recv, (size_t)c.world * COMM_CAP * COMM_REC * 8));
    HIPCHK(hipHostMalloc((void**)&c.h_buf, (size_t)(c.world + 1) * COMM_CAP * COMM_REC * 8));
    c.cap = COMM_CAP;
    return TYPLONK_OK;
}

// every point <- sum over the ranks of that rank's point: all-gather of the records on the context's stream, fold in
// rank order on the host (fixed order and a canonical result: bit-identical on every rank)
//
// local_rc: the status of the local work the points come from.  A rank whose MSM or prover round failed must not leave
// its peers waiting inside the collective, so it still takes part -- with its records flagged (bits 32.. of the flag
// word) -- and EVERY rank then returns an error: the failing rank its own code, the others TYPLONK_ERR_COMM naming it.
int comm_fold(typlonk_ctx* ctx, uint64_t* xy, uint8_t* inf, size_t count, int local_rc = TYPLONK_OK) {
    Comm& c = ctx->comm;
    if (!c.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "no communicator on this context (typlonk_comm_init)");
    if (!count) return local_rc;
    if (count > c.cap) {   // pieces of COMM_CAP records, each its own collective: the same sequence on every rank
        int rc = TYPLONK_OK;
        for (size_t i = 0; i < count; i += c.cap) {
            const int r = comm_fold(ctx, xy + 12 * i, inf + i, std::min(c.cap, count - i), local_rc);
            if (!rc) rc = r;
        }
        return rc;
    }
    const std::string local_err = local_rc ? ctx->err : std::string();
    int rc = TYPLONK_OK;
    uint64_t* out = c.h_buf;
    uint64_t* back = c.h_buf + c.cap * COMM_REC;
    for (size_t i = 0; i < count; ++i) {
        if (local_rc) {
            memset(out + i * COMM_REC, 0, 96);
            out[i * COMM_REC + 12] = 1u | ((uint64_t)(uint32_t)(-local_rc) << 32);   // identity + the error code
        } else {
            memcpy(out + i * COMM_REC, xy + 12 * i, 96);
            out[i * COMM_REC + 12] = inf[i];
        }
    }
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemcpyAsync(c.d_send, out, count * COMM_REC * 8, hipMemcpyHostToDevice, s));
    NCCLCHK(rccl_api()->AllGather(c.d_send, c.d_recv, count * COMM_REC, ncclUint64, c.comm, s));
    HIPCHK(hipMemcpyAsync(back, c.d_recv, (size_t)c.world * count * COMM_REC * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (local_rc) return fail(ctx, local_rc, local_err);   // (its flagged records made every peer fail too)
    int failed = -1;
    rc = typlonk_g1_fold_records_host(back, (size_t)c.world, count, xy, inf, &failed);
    if (rc == TYPLONK_ERR_COMM) {
        const uint64_t flag = back[(size_t)failed * count * COMM_REC + 12];
        return fail(ctx, rc, "rank " + std::to_string(failed) + " failed before the exchange (its error code " +
                                 std::to_string(-(int)(flag >> 32)) + ")");
    }
    if (rc) return fail(ctx, rc, "fold of the gathered partial sums failed");
    return TYPLONK_OK;
}

// does this MSM / prover call need the fold?  (an SRS shard on a context with a communicator)
bool comm_folds(typlonk_ctx* ctx, uint32_t srs_id) {
    if (!ctx->comm.comm) return false;
    auto it = ctx->srs.find(srs_id);
    return it != ctx->srs.end() && it->second.total_len != 0;
}

}  // namespace

// ================================================================================================
extern "C" {

int typlonk_comm_unique_id(uint8_t id[TYPLONK_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == TYPLONK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return TYPLONK_ERR_INVALID_ARG;
    RcclApi* api = rccl_api();
    if (!api->err.empty()) return TYPLONK_ERR_COMM;
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return TYPLONK_ERR_COMM;
    memcpy(id, &u, sizeof(u));
    return TYPLONK_OK;
}

int typlonk_comm_init(typlonk_ctx* ctx, const uint8_t id[TYPLONK_COMM_ID_BYTES], int rank, int world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "bad communicator arguments");
    if (ctx->comm.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "this context already has a communicator");
    RcclApi* api = rccl_api();
    if (!api->err.empty()) return fail(ctx, TYPLONK_ERR_COMM, api->err);
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    NCCLCHK(api->CommInitRank(&comm, world, u, rank));
    ctx->comm.comm = comm;
    ctx->comm.rank = rank;
    ctx->comm.world = world;
    const int rc = comm_reserve(ctx);   // every rank allocates here, before any fold: a failure is reported by this call
    if (rc) {
        const std::string msg = ctx->err;
        comm_release(ctx);
        return fail(ctx, rc, msg);
    }
    return TYPLONK_OK;
}

int typlonk_comm_available(void) { return rccl_api()->err.empty() ? 1 : 0; }

int typlonk_comm_destroy(typlonk_ctx* ctx) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (ctx->comm.comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    comm_release(ctx);
    return TYPLONK_OK;
}

int typlonk_comm_info(const typlonk_ctx* ctx, int* rank, int* world) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (rank) *rank = ctx->comm.rank;
    if (world) *world = ctx->comm.comm ? ctx->comm.world : 0;
    return TYPLONK_OK;
}

int typlonk_comm_fold_g1(typlonk_ctx* ctx, uint64_t* xy, uint8_t* inf, size_t count) {
    if (!ctx || ((!xy || !inf) && count)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return comm_fold(ctx, xy, inf, count);
}

const char* typlonk_version(void) { return "typlonk-mi355x 0.1 (gfx950)"; }

const char* typlonk_strerror(int code) {
    switch (code) {
        case TYPLONK_OK: return "ok";
        case TYPLONK_ERR_INVALID_ARG: return "invalid argument";
        case TYPLONK_ERR_LENGTH: return "MSM length exceeds SRS length";
        case TYPLONK_ERR_DOMAIN: return "unsupported evaluation-domain size";
        case TYPLONK_ERR_NO_DEVICE: return "no HIP device available (no CPU fallback)";
        case TYPLONK_ERR_HIP: return "HIP runtime error";
        case TYPLONK_ERR_OOM: return "device out of memory";
        case TYPLONK_ERR_RANGE: return "range outside device buffer";
        case TYPLONK_ERR_UNSATISFIED: return "witness does not satisfy the circuit (r(zeta) != 0)";
        case TYPLONK_ERR_COMM: return "RCCL error";
        default: return "unknown error";
    }
}

const char* typlonk_last_error(const typlonk_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int typlonk_init(typlonk_ctx** out, int device_ordinal) {
    if (!out) return TYPLONK_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TYPLONK_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= count) return TYPLONK_ERR_INVALID_ARG;
    if (hipSetDevice(device_ordinal) != hipSuccess) return TYPLONK_ERR_HIP;
    typlonk_ctx* ctx = new typlonk_ctx();
    ctx->device = device_ordinal;
    // an ordinary (blocking) stream: ordered after work on the legacy default stream, where a caller that never
    // created a stream (PyTorch-ROCm by default) produced the device-resident inputs of the *_devptr calls
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamDefault) != hipSuccess) {
        delete ctx;
        return TYPLONK_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (const char* e = getenv("TYPLONK_MSM_C")) {
        int c = atoi(e);
        if (c >= 4 && c <= 20) ctx->msm_c_override = c;
    }
    if (const char* e = getenv("TYPLONK_MSM_SORT")) ctx->msm_legacy_sort = (strcmp(e, "atomic") == 0);
    if (const char* e = getenv("TYPLONK_MSM_REDUCE")) {
        ctx->msm_tree_reduce = (strcmp(e, "running") == 0);
        ctx->msm_rc4 = (strcmp(e, "rc4") == 0);
        ctx->msm_rc2_force = (strcmp(e, "rc2") == 0);
    }
    if (const char* e = getenv("TYPLONK_MSM_STAGGER")) ctx->msm_stagger = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_MSM_HOST_PLANES")) ctx->msm_host_planes = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_MSM_SIDE_PRIO")) ctx->msm_side_prio = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_MSM_CHAIN")) ctx->msm_chain = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_MSM_CAP_MIN")) ctx->msm_cap_min = (uint32_t)std::max(16, atoi(e));
    if (const char* e = getenv("TYPLONK_MSM_LANES_SPLIT")) ctx->msm_lanes_split = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_MSM_LANES")) {
        const int l = atoi(e);
        if (l == 1 || l == 2 || l == 4 || l == 8 || l == 16) ctx->msm_lanes = l;
    }
    if (const char* e = getenv("TYPLONK_MSM_INFLIGHT")) ctx->msm_inflight = atoi(e);
    if (const char* e = getenv("TYPLONK_PROVER_OVERLAP")) ctx->prover_overlap = atoi(e);
    if (const char* e = getenv("TYPLONK_MSM_CHUNKS")) ctx->msm_chunks = std::max(0, std::min(atoi(e), MSM_MAX_CHUNKS));
    if (const char* e = getenv("TYPLONK_NTT_BIG")) ctx->ntt_big_tiles = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_NTT_FULL_TABLES")) ctx->ntt_full_tables = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_NTT_FR30")) ctx->ntt_fr30 = std::max(0, std::min(atoi(e), 2));
    if (const char* e = getenv("TYPLONK_NTT_FULL_MAX_LOG")) ctx->ntt_full_max_log = (uint32_t)atoi(e);
    if (const char* e = getenv("TYPLONK_NTT_RADIX")) ctx->ntt_radix4 = atoi(e) != 2;
    if (const char* e = getenv("TYPLONK_NTT_DIRECT")) ctx->ntt_direct = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_NTT_TILE")) ctx->ntt_tile_log = atoi(e);
    if (const char* e = getenv("TYPLONK_NTT_SHORT_IN")) ctx->ntt_short_in = atoi(e) != 0;
    *out = ctx;
    return TYPLONK_OK;
}

void typlonk_destroy(typlonk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    prof_begin(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    ctx->event_pool.clear();
    for (auto& kv : ctx->srs) (void)hipFree(kv.second.d_points);
    for (auto& kv : ctx->circuits) {
        (void)hipFree(kv.second.ext);
        (void)hipFree(kv.second.coef);
        (void)hipFree(kv.second.sig_ev);
    }
    for (auto& kv : ctx->tables) (void)hipFree(kv.second.d);
    for (DevBuf* b : {&ctx->scal, &ctx->ntt_scratch, &ctx->ntt_io, &ctx->quot_ext, &ctx->quot_tab, &ctx->ops_tmp, &ctx->prover_mem}) release(*b);
    for (MsmWs& ws : ctx->ws) {
        for (SortBufs& sb : ws.sb)
            for (DevBuf* b : sb.all()) release(*b);
        for (DevBuf* b : {&ws.buckets, &ws.part_a, &ws.part_b, &ws.rc_sums, &ws.rc_bits, &ws.rc_out}) release(*b);
        if (ws.host_wins) (void)hipHostFree(ws.host_wins);
        if (ws.side) (void)hipStreamDestroy(ws.side);
        if (ws.ev_in) (void)hipEventDestroy(ws.ev_in);
        for (hipEvent_t e : ws.ev_sorted)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ws.ev_acc)
            if (e) (void)hipEventDestroy(e);
    }
    comm_release(ctx);
    for (hipStream_t l : ctx->lane)
        if (l) (void)hipStreamDestroy(l);
    for (hipEvent_t e : ctx->lane_evt)
        if (e) (void)hipEventDestroy(e);
    if (ctx->batch_fence) (void)hipEventDestroy(ctx->batch_fence);
    if (ctx->accum_chain) (void)hipEventDestroy(ctx->accum_chain);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int typlonk_set_stream(typlonk_ctx* ctx, void* hip_stream) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return TYPLONK_OK;
}

int typlonk_sync(typlonk_ctx* ctx) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TYPLONK_OK;
}

int typlonk_srs_load(typlonk_ctx* ctx, const uint64_t* xy, const uint8_t* inf, size_t len, uint32_t* srs_id) {
    if (!ctx || !srs_id || (!xy && len)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    SrsEntry e;
    e.len = len;
    DevGuard guard;
    HIPCHK(hipMalloc((void**)&e.d_points, std::max<size_t>(len, 1) * PT_WORDS * 4));
    guard.add(e.d_points);
    if (len) {
        HIPCHK(hipMemcpy2DAsync(e.d_points, PT_WORDS * 4, xy, 96, 96, len, hipMemcpyHostToDevice, ctx->stream));
        DevGuard flags;  // freed on every path out of this block
        uint8_t* d_inf = nullptr;
        if (inf) {
            HIPCHK(hipMalloc((void**)&d_inf, len));
            flags.add(d_inf);
            HIPCHK(hipMemcpyAsync(d_inf, inf, len, hipMemcpyHostToDevice, ctx->stream));
        }
        launch_convert_points(e.d_points, d_inf, (uint64_t)len, ctx->stream);  // arkworks -> internal form
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    guard.dismiss();
    const uint32_t id = ctx->next_srs++;
    ctx->srs[id] = e;
    *srs_id = id;
    return TYPLONK_OK;
}

int typlonk_srs_free(typlonk_ctx* ctx, uint32_t srs_id) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(it->second.d_points));
    ctx->srs.erase(it);
    return TYPLONK_OK;
}

int typlonk_srs_set_shard(typlonk_ctx* ctx, uint32_t srs_id, size_t first_index, size_t total_len) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    if (first_index > total_len || it->second.len > total_len - first_index)
        return fail(ctx, TYPLONK_ERR_RANGE, "shard does not fit into total_len");
    it->second.shard_first = first_index;
    it->second.total_len = total_len;
    return TYPLONK_OK;
}

int typlonk_srs_len(typlonk_ctx* ctx, uint32_t srs_id, size_t* len) {
    if (!ctx || !len) return TYPLONK_ERR_INVALID_ARG;
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    *len = it->second.len;
    return TYPLONK_OK;
}

int typlonk_srs_generate(typlonk_ctx* ctx, const uint64_t secret[4], uint64_t start, size_t len, uint32_t* srs_id) {
    if (!ctx || !secret || !srs_id) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    SrsEntry e;
    e.len = len;
    DevGuard guard;
    HIPCHK(hipMalloc((void**)&e.d_points, std::max<size_t>(len, 1) * PT_WORDS * 4));
    guard.add(e.d_points);
    if (len) {
        Fr s;
        memcpy(s.v, secret, sizeof(s.v));
        launch_srs_generate(s, start, (uint64_t)len, e.d_points, ctx->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    guard.dismiss();
    const uint32_t id = ctx->next_srs++;
    ctx->srs[id] = e;
    *srs_id = id;
    return TYPLONK_OK;
}

int typlonk_srs_precompute(typlonk_ctx* ctx, uint32_t srs_id, uint32_t window_bits) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    if (window_bits == 0) {
        // auto: 17 below 2^19 points, else 20 (measured best: DESIGN.md section 6) -- and nothing at all for an SRS shorter
        // than 2^14 points: 2^16 buckets (sort, reduction, heavy-bucket launch) for a handful of terms would be slower
        // than the plain path, whose window follows the length
        if (it->second.len < TYPLONK_TABLES_AUTO_MIN_LEN) return TYPLONK_OK;
        window_bits = it->second.len < (1u << 19) ? 17 : 20;
    }
    if (window_bits < 14 || window_bits > 20) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "window_bits must be 0 (auto) or 14..20");
    SrsEntry& e = it->second;
    if (e.table_T) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "tables already built for this SRS");
    if (e.len == 0 || e.len > (1u << 23)) return fail(ctx, TYPLONK_ERR_LENGTH, "tables need 1 <= len <= 2^23");
    HIPCHK(hipSetDevice(ctx->device));
    // centred scalars (|k| < 2^254) save a window -- and a table -- for c = 17 (15 instead of 16) and c = 15
    const bool centred = msm_windows(window_bits, true) < msm_windows(window_bits, false);
    const uint32_t T = msm_windows(window_bits, centred);
    // An MSM whose length has no table-mode sort shape (m > 2^22 with 20-bit windows: 23 index bits leave too few low
    // bucket bits for the LDS level of the sort) simply takes the plain path over table 0, which IS the SRS
    // (msm_enqueue) -- a set-up call that is supposed to be speed-only never turns a valid MSM into an error.  Only a
    // window for which not even the shortest table-mode MSM (len / 4 terms) could be sorted is refused.
    if (!msm_table_shape_ok(std::max<size_t>(e.len / 4, 1), window_bits, T))
        return fail(ctx, TYPLONK_ERR_LENGTH, "fixed-base tables with this window are not supported for an SRS of this length");
    uint32_t* big = nullptr;
    HIPCHK(hipMalloc((void**)&big, (size_t)T * e.len * PT_WORDS * 4));
    DevGuard guard;
    guard.add(big);
    HIPCHK(hipMemcpyAsync(big, e.d_points, e.len * PT_WORDS * 4, hipMemcpyDeviceToDevice, ctx->stream));
    launch_srs_tables(big, (uint64_t)e.len, window_bits, T, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    guard.dismiss();
    HIPCHK(hipFree(e.d_points));
    e.d_points = big;
    e.table_c = window_bits;
    e.table_T = T;
    e.table_centred = centred;
    return TYPLONK_OK;
}

int typlonk_srs_download(typlonk_ctx* ctx, uint32_t srs_id, size_t offset, size_t count, uint64_t* xy, uint8_t* inf) {
    if (!ctx || (!xy && count)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    if (offset > it->second.len || count > it->second.len - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside SRS");
    if (!count) return TYPLONK_OK;
    HIPCHK(hipMemcpy2DAsync(xy, 96, it->second.d_points + offset * PT_WORDS, PT_WORDS * 4, 96, count, hipMemcpyDeviceToHost,
                            ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < count; ++i) {  // internal packed form -> arkworks; (0,0) -> ark-ec (0, 1, inf)
        G1Affine a;
        uint32_t w[12];
        memcpy(w, xy + i * 12, 48);
        a.x = fq30_unpack(w);
        memcpy(w, xy + i * 12 + 6, 48);
        a.y = fq30_unpack(w);
        uint8_t f = 0;
        write_affine_out(a, xy + i * 12, &f);
        if (inf) inf[i] = f;
    }
    return TYPLONK_OK;
}

int typlonk_msm_g1_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* d_scalars, size_t m, uint64_t out_xy[12],
                          uint8_t* out_inf) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return msm_run(ctx, srs_id, (const Fr*)d_scalars, m, out_xy, out_inf);
}

int typlonk_msm_g1_batch_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m,
                                size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return msm_batch(ctx, srs_id, d_scalars, m, count, out_xy, out_inf);
}

// evaluate_in_s over the whole node: this rank's partial sum over its SRS shard, then the fold of all ranks' sums
int typlonk_msm_g1_sharded_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* d_scalars, size_t m, uint64_t out_xy[12],
                                  uint8_t* out_inf) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (!ctx->comm.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "no communicator on this context (typlonk_comm_init)");
    HIPCHK(hipSetDevice(ctx->device));
    if (!out_xy || !out_inf) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null output");
    const int rc = msm_run(ctx, srs_id, (const Fr*)d_scalars, m, out_xy, out_inf);
    return comm_fold(ctx, out_xy, out_inf, 1, rc);   // a failed rank still joins the collective, flagged
}

int typlonk_msm_g1_sharded_batch_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m,
                                        size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (!ctx->comm.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "no communicator on this context (typlonk_comm_init)");
    HIPCHK(hipSetDevice(ctx->device));
    if (!out_xy || !out_inf || !m) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    const int rc = msm_batch(ctx, srs_id, d_scalars, m, count, out_xy, out_inf);
    return comm_fold(ctx, out_xy, out_inf, count, rc);   // one collective for the whole group; failures travel with it
}

int typlonk_msm_g1_dev(typlonk_ctx* ctx, uint32_t srs_id, const typlonk_buf* scalars, size_t offset, size_t m,
                       uint64_t out_xy[12], uint8_t* out_inf) {
    if (!ctx || !scalars) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (offset > scalars->n || m > scalars->n - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside buffer");
    HIPCHK(hipSetDevice(ctx->device));
    return msm_run(ctx, srs_id, scalars->d + offset, m, out_xy, out_inf);
}

int typlonk_msm_g1(typlonk_ctx* ctx, uint32_t srs_id, const uint64_t* scalars, size_t m, uint64_t out_xy[12],
                   uint8_t* out_inf) {
    if (!ctx || (!scalars && m)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    // validate the length before touching the device so the error matches the reference's assert
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    if (m > it->second.total()) return fail(ctx, TYPLONK_ERR_LENGTH, "MSM length exceeds SRS length (kzg/src/lib.rs:43)");
    size_t off, ml;
    it->second.local_range(m, &off, &ml);
    if (ml) {  // only this entry's share of the coefficients crosses PCIe
        int rc = ensure(ctx, ctx->scal, ml * sizeof(Fr));
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(ctx->scal.p, scalars + 4 * off, ml * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    }
    return msm_run(ctx, srs_id, (const Fr*)ctx->scal.p, m, out_xy, out_inf, /*ptr_is_local=*/true);
}

int typlonk_ntt_fr_devptr(typlonk_ctx* ctx, void* d_data, uint32_t log_n, int inverse, const uint64_t* coset_shift) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return ntt_run(ctx, (Fr*)d_data, log_n, inverse, coset_shift, /*sync=*/false);
}

int typlonk_ntt_fr_dev(typlonk_ctx* ctx, typlonk_buf* buf, size_t offset, uint32_t log_n, int inverse,
                       const uint64_t* coset_shift) {
    if (!ctx || !buf) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (log_n > 32) return fail(ctx, TYPLONK_ERR_DOMAIN, "log_n > 32 (Fr two-adicity)");
    const uint64_t N = 1ull << log_n;
    if (offset > buf->n || N > buf->n - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside buffer");
    HIPCHK(hipSetDevice(ctx->device));
    return ntt_run(ctx, buf->d + offset, log_n, inverse, coset_shift, /*sync=*/false);
}

int typlonk_ntt_fr(typlonk_ctx* ctx, uint64_t* data, uint32_t log_n, int inverse, const uint64_t* coset_shift) {
    if (!ctx || !data) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (log_n > 32) return fail(ctx, TYPLONK_ERR_DOMAIN, "log_n > 32 (Fr two-adicity)");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = ((size_t)1 << log_n) * sizeof(Fr);
    int rc = ensure(ctx, ctx->ntt_io, bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->ntt_io.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = ntt_run(ctx, (Fr*)ctx->ntt_io.p, log_n, inverse, coset_shift);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(data, ctx->ntt_io.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TYPLONK_OK;
}

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
    if (src && ctx->ntt_short_in) return ntt_run(ctx, e, log4, 0, quotient_coset_g(nullptr), /*sync=*/false, src, n);
    if (src) {
        HIPCHK(hipMemcpyAsync(e, src, n * sizeof(Fr), hipMemcpyDeviceToDevice, s));
    } else {
        launch_fr_fill(e, n, *fill, s);
    }
    HIPCHK(hipMemsetAsync(e + n, 0, 3 * n * sizeof(Fr), s));
    return ntt_run(ctx, e, log4, 0, quotient_coset_g(nullptr), /*sync=*/false);
}
}  // namespace

int typlonk_circuit_load(typlonk_ctx* ctx, const typlonk_buf* const selectors[5], const typlonk_buf* const sigma[3],
                         uint32_t log_n, uint32_t* circuit_id) {
    if (!ctx || !selectors || !sigma || !circuit_id) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (log_n < 1 || log_n > 30) return fail(ctx, TYPLONK_ERR_DOMAIN, "quotient needs 1 <= log_n <= 30");
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
    for (int k = 0; k < 9 && !rc; ++k)
        rc = quotient_extend(ctx, e.ext + (uint64_t)k * n4, k < 8 ? in[k]->d : nullptr, &ninv, n, log_n + 2);
    for (int k = 0; k < 3 && !rc; ++k) rc = ntt_run(ctx, e.sig_ev + (uint64_t)k * n, log_n, 0, nullptr, /*sync=*/false);
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
    if (log_n < 1 || log_n > 30) return fail(ctx, TYPLONK_ERR_DOMAIN, "quotient needs 1 <= log_n <= 30");
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
    for (int k = 0; k < (cached ? 5 : 14) && !rc; ++k) {
        if (k == 4 && !has_pi) continue;
        if (k < 5 && ((extended >> k) & 1u)) continue;
        rc = quotient_extend(ctx, ext + (uint64_t)k * n4, k < 13 ? in[k]->d : nullptr, &ninv, n, log4);
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
    if (log_n > 30) return fail(ctx, TYPLONK_ERR_DOMAIN, "log_n > 30");
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
    Fr total;
    HIPCHK(hipMemcpyAsync(&total, dsuf, sizeof(Fr), hipMemcpyDeviceToHost, s));  // S_0 = prod of all denominators
    HIPCHK(hipStreamSynchronize(s));
    launch_gp_finish(npre, dsuf, fe_inv(total), n, z_evals_out->d, s);
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
    int early = 0;          // opening witnesses of round 3 whose MSM was submitted before the quotient
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
// ops_tmp layout of the prover's openings: [0, 8*2048) per-workgroup carries, then 16 result slots
constexpr size_t PROVER_EVAL_BLOCKS = 8 * 2048;
int prover_ops_tmp(typlonk_prover* p, Fr** blocks, Fr** slots) {
    typlonk_ctx* ctx = p->ctx;
    int rc = ensure(ctx, ctx->ops_tmp, (PROVER_EVAL_BLOCKS + 16) * sizeof(Fr));
    if (rc) return rc;
    *blocks = (Fr*)ctx->ops_tmp.p;
    *slots = *blocks + PROVER_EVAL_BLOCKS;
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

int typlonk_prover_round1(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const wire_evals[3],
                          const typlonk_buf* pi_evals, typlonk_prover** out, uint64_t commit_xy[3][12],
                          uint8_t commit_inf[3]) {
    if (!ctx || !wire_evals || !out || !commit_xy || !commit_inf) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    auto ci = ctx->circuits.find(circuit_id);
    if (ci == ctx->circuits.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown circuit id");
    const SrsEntry* srs = nullptr;
    const uint32_t log_n = ci->second.log_n;
    const uint64_t n = 1ull << log_n;
    int rc = msm_validate(ctx, srs_id, n, &srs);  // every committed polynomial has <= n coefficients
    if (rc) return rc;
    if (n > (1u << 22)) return fail(ctx, TYPLONK_ERR_LENGTH, "prover supports up to 2^22 rows");
    for (int i = 0; i < 3; ++i)
        if (!wire_evals[i] || wire_evals[i]->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "wire column shorter than n");
    if (pi_evals && pi_evals->n < n) return fail(ctx, TYPLONK_ERR_RANGE, "public-input column shorter than n");
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
    MsmQueue q(ctx, srs, /*first_lane=*/1);
    for (int i = 0; i < 3 && !rc; ++i) {
        if ((rc = d2d(p->ev[i], wire_evals[i]->d))) break;
        if ((rc = d2d(p->co[i], wire_evals[i]->d))) break;
        if ((rc = ntt_run(ctx, p->co[i], log_n, 1, nullptr, false))) break;
        rc = q.submit(p->co[i], n, commit_xy[i], commit_inf + i);
    }
    p->has_pi = pi_evals != nullptr;  // NULL: public inputs [0] -> the zero polynomial
    if (!rc && p->has_pi) {
        rc = d2d(p->pi, pi_evals->d);
        if (!rc) rc = ntt_run(ctx, p->pi, log_n, 1, nullptr, false);  // proof.rs:105-106
    }
    if (ctx->prover_overlap & 1) {
        for (int i = 0; i < 3 && !rc; ++i) rc = prover_extend(p, i, p->co[i]);
        if (!rc && p->has_pi) rc = prover_extend(p, 4, p->pi);
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
            if (!rc && (ctx->prover_overlap & 2)) rc = prover_extend(p, 3, p->z);  // Z's coset transform runs beside its commitment
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
        // the witnesses of a, b, c at zeta need nothing else: their MSMs start on the lanes beside the context's
        // stream while the quotient below (and the linearisation after it) is still being computed
        int early = 0;
        if (!batched && (ctx->prover_overlap & 4))
            for (; early < 3 && early < q.lanes - 1 && !rc; ++early) rc = q.submit(p->q[early], n - 1, xy[early], inf + early);
        p->early = early;
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
        if (!rc) rc = prover_fetch(p, host, 9);
        for (int i = 0; i < 4; ++i) ev[i] = host[i];
        ev[4] = host[8];
        s0 = host[4];
        s1 = host[5];
        if (p->has_pi) pi_z = host[6];
    }
    if (!rc) {
        const Fr one = Fr::one();
        Fr zn = ze;  // zeta^n
        for (uint32_t i = 0; i < log_n; ++i) zn = fe_sqr(zn);
        const Fr zh = fe_sub(zn, one);  // evaluate_vanishing_polynomial(zeta)
        // L0(zeta) = (zeta^n - 1) / (n (zeta - 1)); the polynomial (1/n) sum X^i evaluates to 1 at zeta = 1
        Fr l0z = one;
        const Fr zm1 = fe_sub(ze, one);
        if (!zm1.is_zero()) l0z = fe_mul(zh, fe_inv(fe_mul(fr_from_u64(n), zm1)));
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
    if (!rc) rc = prover_open(p, p->r, n, ze, batched ? nullptr : p->q[5], &ev[5]);          // proof.rs:175
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
        for (int k = p->early; k < 9 && !rc; ++k) rc = q.submit(polys[k], m[k], xy[k], inf + k);
        {
            const int r = q.wait_all();
            if (!rc) rc = r;
        }
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

int typlonk_prove(typlonk_ctx* ctx, uint32_t srs_id, uint32_t circuit_id, const typlonk_buf* const wire_evals[3],
                  const typlonk_buf* pi_evals, const uint64_t cosets[3][4], typlonk_proof* out) {
    if (!ctx || !wire_evals || !cosets || !out) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    typlonk_prover* p = nullptr;
    // An SRS shard on a context with a communicator: every round's partial commitments are folded over the ranks (one
    // all-gather per round), so all ranks hash the same points and end with the same proof.  A rank whose round fails
    // (an OOM, say) still joins that round's collective with flagged records, so its peers return TYPLONK_ERR_COMM
    // instead of waiting for ever (comm_fold).
    const bool folds = comm_folds(ctx, srs_id);
    int rc = typlonk_prover_round1(ctx, srs_id, circuit_id, wire_evals, pi_evals, &p, out->commit_xy, out->commit_inf);
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

void typlonk_prover_free(typlonk_prover* p) {
    if (!p) return;
    (void)hipStreamSynchronize(p->ctx->stream);
    p->ctx->prover_busy = false;
    delete p;
}

int typlonk_buf_alloc(typlonk_ctx* ctx, size_t n_elems, typlonk_buf** out) {
    if (!ctx || !out) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    typlonk_buf* b = new typlonk_buf();
    b->n = n_elems;
    hipError_t e = hipMalloc((void**)&b->d, std::max<size_t>(n_elems, 1) * sizeof(Fr));
    if (e != hipSuccess) {
        delete b;
        return fail(ctx, TYPLONK_ERR_OOM, hipGetErrorString(e));
    }
    *out = b;
    return TYPLONK_OK;
}

int typlonk_buf_free(typlonk_ctx* ctx, typlonk_buf* buf) {
    if (!ctx || !buf) return TYPLONK_ERR_INVALID_ARG;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(buf->d));
    delete buf;
    return TYPLONK_OK;
}

int typlonk_buf_upload(typlonk_ctx* ctx, typlonk_buf* buf, size_t offset, const uint64_t* src, size_t n_elems) {
    if (!ctx || !buf || (!src && n_elems)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (offset > buf->n || n_elems > buf->n - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside buffer");
    if (!n_elems) return TYPLONK_OK;
    HIPCHK(hipMemcpyAsync(buf->d + offset, src, n_elems * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TYPLONK_OK;
}

int typlonk_buf_download(typlonk_ctx* ctx, const typlonk_buf* buf, size_t offset, uint64_t* dst, size_t n_elems) {
    if (!ctx || !buf || (!dst && n_elems)) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (offset > buf->n || n_elems > buf->n - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside buffer");
    if (!n_elems) return TYPLONK_OK;
    HIPCHK(hipMemcpyAsync(dst, buf->d + offset, n_elems * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return TYPLONK_OK;
}

int typlonk_buf_zero(typlonk_ctx* ctx, typlonk_buf* buf, size_t offset, size_t n_elems) {
    if (!ctx || !buf) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (offset > buf->n || n_elems > buf->n - offset) return fail(ctx, TYPLONK_ERR_RANGE, "range outside buffer");
    if (!n_elems) return TYPLONK_OK;
    HIPCHK(hipMemsetAsync(buf->d + offset, 0, n_elems * sizeof(Fr), ctx->stream));
    return TYPLONK_OK;
}

size_t typlonk_buf_len(const typlonk_buf* buf) { return buf ? buf->n : 0; }
void* typlonk_buf_devptr(const typlonk_buf* buf) { return buf ? (void*)buf->d : nullptr; }

int typlonk_g1_sum_host(const uint64_t* xy, const uint8_t* inf, size_t count, uint64_t out_xy[12], uint8_t* out_inf) {
    if ((!xy && count) || !out_xy || !out_inf) return TYPLONK_ERR_INVALID_ARG;
    // arkworks' words ARE the 6 x 64-bit Montgomery form of g1_host64.hpp: no conversion in or out
    namespace H = h64;
    const H::Fq one = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull,
                        0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};  // 2^384 mod p
    H::Xyzz acc = H::inf();
    for (size_t i = 0; i < count; ++i) {
        if (inf && inf[i]) continue;
        H::Xyzz p;
        memcpy(p.x.v, xy + i * 12, 48);
        memcpy(p.y.v, xy + i * 12 + 6, 48);
        p.zz = one;
        p.zzz = one;
        acc = H::xyzz_add(acc, p);
    }
    if (H::xyzz_to_affine(acc, out_xy)) *out_inf = 0;
    else write_affine_out(G1Affine::inf(), out_xy, out_inf);
    return TYPLONK_OK;
}

int typlonk_g1_fold_records_host(const uint64_t* records, size_t world, size_t count, uint64_t* out_xy, uint8_t* out_inf,
                                 int* failed_rank) {
    static_assert(COMM_REC == TYPLONK_COMM_RECORD_WORDS, "record layout");
    if (!records || !world || ((!out_xy || !out_inf) && count)) return TYPLONK_ERR_INVALID_ARG;
    for (size_t r = 0; r < world; ++r)
        for (size_t i = 0; i < count; ++i)
            if (records[(r * count + i) * COMM_REC + 12] >> 32) {
                if (failed_rank) *failed_rank = (int)r;
                return TYPLONK_ERR_COMM;
            }
    std::vector<uint64_t> pxy(world * 12);
    std::vector<uint8_t> pinf(world);
    for (size_t i = 0; i < count; ++i) {
        for (size_t r = 0; r < world; ++r) {   // all-gather layout: rank-major, `count` records per rank
            const uint64_t* rec = records + (r * count + i) * COMM_REC;
            memcpy(&pxy[r * 12], rec, 96);
            pinf[r] = (uint8_t)(rec[12] & 1u);
        }
        const int rc = typlonk_g1_sum_host(pxy.data(), pinf.data(), world, out_xy + 12 * i, out_inf + i);
        if (rc) return rc;
    }
    return TYPLONK_OK;
}

int typlonk_set_profiling(typlonk_ctx* ctx, int on) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    ctx->profiling = on != 0;
    return TYPLONK_OK;
}

int typlonk_profile_get(typlonk_ctx* ctx, const char** names, float* ms, int cap) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    const int n = (int)ctx->prof_result.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = ctx->prof_result[i].first;
        if (ms) ms[i] = ctx->prof_result[i].second;
    }
    return n;
}

int typlonk_msm_plan(typlonk_ctx* ctx, size_t m, uint32_t* window_bits, uint32_t* n_windows, uint64_t* group_ops) {
    uint32_t c, W;
    msm_shape(ctx, m ? m : 1, &c, &W);
    if (window_bits) *window_bits = c;
    if (n_windows) *n_windows = W;
    // Pippenger operation count for this shape: one mixed add per (term, window), two adds per
    // bucket in the running-sum reduction, c doublings per window in the final combine.
    if (group_ops) *group_ops = (uint64_t)W * m + 2ull * W * (1ull << (c - 1)) + (uint64_t)c * (W - 1);
    return TYPLONK_OK;
}

}  // extern "C"
