// libtyplonk_hip.so -- NTT planning: twiddle / coset tables, pass decomposition, the typlonk_ntt_* entry points
// Part of the host driver of include/typlonk.h (see host.hpp for the shared state).  There is deliberately no CPU compute
// fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "host.hpp"
#include "fr30.hpp"

using namespace ty;
using namespace tyh;

namespace tyh {

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

// powers table of the 9 x 30-bit kernel: out[j] = scale * base^j re-cut into nine 30-bit limbs on a 12-word (48-byte)
// stride -- what NttArith30::ldtw (ntt_kernels.hip) reads from global memory
int get_pow_table30(typlonk_ctx* ctx, const std::string& key, const Fr& base, const Fr& scale, size_t n, Table* out) {
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        it->second.last_use = ++ctx->table_tick;
        *out = it->second;
        return TYPLONK_OK;
    }
    std::vector<Fr> h((12 * n + 7) / 8, Fr::zero());
    uint32_t* w = reinterpret_cast<uint32_t*>(h.data());
    Fr x = scale;
    for (size_t j = 0; j < n; ++j) {
        for (int i = 0; i < 9; ++i) {   // fr30_unpack (fr30.hpp) on the host
            const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
            uint32_t t = x.v[wi] >> sh;
            if (sh > 2 && wi + 1 < 8) t |= x.v[wi + 1] << (32 - sh);
            w[12 * j + i] = t & 0x3fffffffu;
        }
        x = fe_mul(x, base);
    }
    return upload_table(ctx, key, h, out);
}

// Full one-multiplication table built on the device from a two-level pair (launch_ntt_full_table); kept per
// context like every other table.  Sizes above 2^NTT_FULL_MAX_LOG entries (2^24 = 512 MB) are not
// built: *out stays empty and the kernel composes the factor from the two-level tables instead.
int get_full_table(typlonk_ctx* ctx, const std::string& key, const Table& lo, const Table& hi, uint32_t h, uint64_t S,
                   uint64_t n, Table* out) {
    *out = Table{};
    if (n > (1ull << NTT_FULL_MAX_LOG)) return TYPLONK_OK;
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
        // built once per context: complete before it enters the cache, so that a later lookup from another stream
        // (typlonk_set_stream, the prover's extension lane) needs no ordering against the build
        HIPCHK(hipStreamSynchronize(ctx->stream));
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

int ntt_run(typlonk_ctx* ctx, Fr* d_data, uint32_t log_n, int inverse, const uint64_t* coset_shift, bool sync, const Fr* short_in,
            uint64_t n_valid) {
    if (!d_data) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null data");
    return ntt_run_batch(ctx, &d_data, 1, log_n, inverse, coset_shift, sync, short_in ? &short_in : nullptr, n_valid);
}

// `count` transforms of the same size, direction and coset in ONE sequence of pass launches: pass p of every vector is
// one launch whose grid is count x the tiles of one vector (NttPassArgs::in / out / blocks_per_vec), the tables are shared.
// The reference always transforms in groups -- the three wire columns (plonk/src/proof.rs:50), their re-evaluation
// (:113-115), the three sigmas (:334-338, :412-418), the five selectors (plonk/src/builder.rs:84-88) -- and a 2^20
// transform alone is one round of tiles on the chip (every CU loads, computes and stores in step); a launch with 3 x
// the tiles lets rounds overlap.  Bit-identical to `count` single calls (the same kernels on the same tiles).
// More than NTT_BATCH_MAX vectors (or more than the scratch cap) go through in groups.
int ntt_run_batch(typlonk_ctx* ctx, Fr* const* d_data, size_t count, uint32_t log_n, int inverse, const uint64_t* coset_shift,
                  bool sync, const Fr* const* short_in, uint64_t n_valid) {
    if (log_n > 32) return fail(ctx, TYPLONK_ERR_DOMAIN, "log_n > 32 (Fr two-adicity)");
    if (!d_data && count) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null data");
    for (size_t v = 0; v < count; ++v)
        if (!d_data[v]) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null data");
    if (count == 0) return TYPLONK_OK;
    // vectors per launch: at most NTT_BATCH_MAX, and a multi-pass transform keeps one scratch vector per batch member
    // (<= 2^25 elements = 1 GiB of scratch in all: the 2^24-point coset extensions of a 2^22-row proof go two at a time)
    size_t group = std::min<size_t>(count, NTT_BATCH_MAX);
    while (group > 1 && (((uint64_t)group) << log_n) > (1ull << 25)) --group;
    if (count > group) {
        int rc = TYPLONK_OK;
        for (size_t g0 = 0; g0 < count && !rc; g0 += group)
            rc = ntt_run_batch(ctx, d_data + g0, std::min(group, count - g0), log_n, inverse, coset_shift, sync,
                               short_in ? short_in + g0 : nullptr, n_valid);
        return rc;
    }
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
    // (round 4: the 30-bit kernel reads its sub-transform twiddles from global memory, so 4096 of its 36-byte elements fit
    // the LDS -- 144 KiB -- and it can take the two-pass form too)
    const bool big = log_n == 20 && ctx->ntt_big != 0 && ctx->prover_rounds_active == 0 && ntt_big_tiles_available();
    // log2 of the tile capacity: 4096-element tiles in the first pass of the two-pass 2^20 transform (strided: four columns
    // make 128-byte runs) and 2048 in the last (rows are contiguous, and two workgroups per CU overlap each other's
    // load / compute / store phases: 0.0775 -> 0.070 ms, profiles/r03_ntt_2_20_tiles.txt)
    const uint32_t cap_first = big ? 12u : 10u;
    const uint32_t cap_last = big ? 11u : 10u;
    split_log(log_n, ks, &P, big);
    const std::string dir = inverse ? "i" : "f";

    Fr* scratch = nullptr;
    if (P >= 2) {
        int rc = ensure(ctx, ctx->ntt_scratch, count * N * sizeof(Fr));
        if (rc) return rc;
        scratch = (Fr*)ctx->ntt_scratch.p;
    }

    // measured (HISTORY.md section 5): the 9 x 30-bit kernel is 9-12 % faster up to 2^19; at 2^20 the two-pass 4096-element
    // tiles of the 8 x 32 kernel win, and from 2^21 on the 36-B LDS elements cost a workgroup per CU (3 instead of 4)
    // and a forward transform pays one extra reducing multiplication per element: mode 1 (default) stops at 2^19
    // Round 3 (radix-4 groups in both kernels, profiles/r03_ntt_fr30_modes.txt): an INVERSE transform is 4-9 % faster on the
    // 30-bit kernel at every size (its n^-1 / coset factor closes the last pass for free), a forward one from 2^21 on is
    // not; at 2^20 a forward transform takes the two-pass big tiles when they are allowed, the 30-bit kernel otherwise
    // Round 4 (profiles/r04_ntt_fr30_global_twiddles.txt): with its twiddles in global memory the 30-bit kernel runs four
    // 1024-element workgroups per CU and takes the two-pass 2^20 form: 2^20 0.141 -> 0.129 ms in both directions, 2^21
    // forward 0.268 -> 0.252; mode 1 (default) = the 30-bit kernel for every inverse transform and for forward ones up to
    // 2^NTT_FR30_FWD_MAX_LOG points (host.hpp)
    const bool want30 = ctx->ntt_fr30 == 2 || (ctx->ntt_fr30 == 1 && (inverse || log_n <= NTT_FR30_FWD_MAX_LOG));

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
    for (uint32_t p = 0; p < P; ++p) f30 = f30 && ks[p] <= FR30_MAX_STAGES;   // the bounds of fr30.hpp hold for k <= 10
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
            if ((rc = get_pow_table30(ctx, "sub30:" + dir + ":" + std::to_string(k), w, c14, (size_t)std::max<uint64_t>(M / 2, 1), &sub30[p])))
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
        for (size_t v = 0; v < count; ++v) {
            const Fr* src = short_in ? short_in[v] : d_data[v];
            Fr* scr = scratch ? scratch + v * N : nullptr;
            a.in[v] = (p == 0) ? src : scr;
            a.out[v] = last ? d_data[v] : scr;
        }
        const uint64_t E = M << logT;
        a.blocks_per_vec = (uint32_t)(N / E);
        const uint64_t blocks = (N / E) * count;
        if (f30) {
            a.sub_tw = nullptr;
            a.sub_tw30 = reinterpret_cast<const uint32_t*>(sub30[p].d);
            a.tw_full = tw30[p].d;
            a.tw_lo = a.tw_hi = nullptr;
            a.pre_lo = a.pre_hi = a.post_lo = a.post_hi = nullptr;
            a.pre_full = p == 0 ? pre30.d : nullptr;
            a.post_full = last ? post30.d : nullptr;
            a.scale = last ? scale30.d : nullptr;
        }
        // LDS: the tile, and (8 x 32 kernel) the sub-transform's twiddles behind it
        const size_t lds = f30 ? (size_t)E * 36 : (size_t)(E + std::max<uint64_t>(M / 2, 1)) * sizeof(Fr);
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
    if (sync || (ctx->profiling && !ctx->prof_light)) HIPCHK(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    return TYPLONK_OK;
}

}  // namespace tyh

// (entry points: C linkage comes from their declarations in include/typlonk.h)

int typlonk_ntt_fr_devptr(typlonk_ctx* ctx, void* d_data, uint32_t log_n, int inverse, const uint64_t* coset_shift) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return ntt_run(ctx, (Fr*)d_data, log_n, inverse, coset_shift, /*sync=*/false);
}

int typlonk_ntt_fr_batch_devptr(typlonk_ctx* ctx, void* const* d_data, size_t count, uint32_t log_n, int inverse,
                                const uint64_t* coset_shift) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (!d_data && count) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (log_n > 32) return fail(ctx, TYPLONK_ERR_DOMAIN, "log_n > 32 (Fr two-adicity)");
    // the vectors are transformed in place and concurrently: two of them must not be the same (or overlapping) storage
    const uint64_t bytes = sizeof(Fr) << log_n;
    for (size_t i = 0; i < count; ++i) {
        if (!d_data[i]) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null data");
        for (size_t j = 0; j < i; ++j) {
            const uintptr_t a = (uintptr_t)d_data[i], b = (uintptr_t)d_data[j];
            if (a < b + bytes && b < a + bytes) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "batched transforms overlap");
        }
    }
    HIPCHK(hipSetDevice(ctx->device));
    return ntt_run_batch(ctx, reinterpret_cast<Fr* const*>(d_data), count, log_n, inverse, coset_shift, /*sync=*/false);
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

