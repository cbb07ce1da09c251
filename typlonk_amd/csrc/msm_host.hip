// libtyplonk_hip.so -- MSM staging: sort / accumulate / reduce launches, the lanes of a batch, the host finish; SRS entry points
// Part of the host driver of include/typlonk.h (see host.hpp for the shared state).  There is deliberately no CPU compute
// fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "host.hpp"

using namespace ty;
using namespace tyh;

namespace tyh {

// ---- MSM ------------------------------------------------------------------------------------
void msm_shape(size_t m, uint32_t* c_out, uint32_t* w_out) {
    uint32_t lg = 0;  // ceil(log2 m)
    while (((size_t)1 << lg) < m) ++lg;
    // measured on MI355X in round 1 (a sweep over forced windows): the best window is c ~ ceil(log2 m) clamped to [8, 16];
    // the bucket reduction is a fixed ~50-operation dependent chain whatever c is, so small MSMs want
    // many small buckets (short accumulate chains) rather than few windows
    int c = (int)lg;
    if (c < 8) c = 8;
    if (c > 16) c = 16;
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

// stand-alone MSM over host scalars with the default chunking: a short first chunk (see msm_enqueue)
static bool overlap_host_first(const typlonk_ctx* ctx, bool standalone, const uint64_t* h_scalars, uint32_t nch, size_t m) {
    // (two default chunks only, i.e. 2^20 <= m < 3 * 2^19: at 2^22, eight chunks, the short first chunk costs 0.07 ms instead)
    return standalone && h_scalars && nch == 2 && !ctx->msm_chunks && m >= ((size_t)1 << 20);
}

// Launch every kernel of one m-term MSM (m > 0, validated by the caller) on `stream` using workspace
// `ws`, ending with the asynchronous copy of the W window sums into ws.host_wins.
// h_scalars != NULL: the scalars are still on the HOST (typlonk_msm_g1: the reference's commit() hands over a Vec<Fr>); every
// chunk's slice is copied to d_scalars on the stream that sorts that chunk, so the copy of chunk k + 1 crosses PCIe while chunk k
// is sorted and accumulated instead of the whole vector crossing before the first kernel starts.
int msm_enqueue(typlonk_ctx* ctx, MsmWs& ws, hipStream_t stream, const SrsEntry& srs, const Fr* d_scalars, size_t m,
                uint64_t* out_xy, uint8_t* out_inf, bool standalone, const uint64_t* h_scalars) {
    ws.stream = stream;
    if (!ws.host_wins) HIPCHK(hipHostMalloc((void**)&ws.host_wins, HOST_WIN_POINTS * 192));
    uint32_t c, W;
    msm_shape(m, &c, &W);
    // fixed-base tables: every window reads its own pre-shifted copy of the base, so all windows share
    // one bucket set (plus a separate set for a thin top window) and no cross-window doublings remain
    // (typlonk_srs_precompute refuses shapes the table-mode sort cannot handle; the check here keeps a plain MSM
    // possible should one slip through)
    const bool tables = srs.table_T != 0 && m >= srs.len / 4 && srs.len <= (1u << 23) && msm_seg_shape(m, srs.table_c, srs.table_T, 1, true).ok;
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
    const uint32_t scan_blocks = (uint32_t)((nb + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);

    // Chunks of terms.  A stand-alone MSM (nothing else in flight to hide behind) is cut into chunks that all add into
    // the SAME buckets: while chunk k is accumulated on the MSM's stream, chunk k + 1 is sorted on the workspace's side
    // stream, so only the first chunk's sort (and the last one's reduction) stay exposed.  Later chunks start from the
    // stored buckets (192 B read + written per bucket and chunk -- noise next to the additions).  Bit-identical
    // results: group addition is commutative and the output is the canonical affine point.
    uint32_t nch = 1;
    if (standalone) {
        // measured (tools/msm_chunks.py, profiles/r02_msm_chunks.jsonl): the overlapped sort is not free -- it competes
        // with the accumulation for issue slots -- and chunks of ~2^19 terms are the best grain: 2 chunks at 2^20
        // (2.76 -> 2.68 ms), 4 at 2^21 (5.09 -> 4.81), 8 at 2^22 (9.92 -> 8.99); below 2^20 one chunk wins
        // (scalars still on the host: two chunks from 2^19 terms on, so that half of the copy hides -- 1.68 -> 1.57 ms at 2^19;
        // neutral at 2^18, a loss at 2^17: profiles/r06_ab_host_scalar_path.txt, call W)
        nch = ctx->msm_chunks ? (uint32_t)ctx->msm_chunks
                              : (m >= (1u << 20) ? (uint32_t)std::min<size_t>(m >> 19, MSM_MAX_CHUNKS)
                                                 : (h_scalars && m >= (1u << 19) ? 2u : 1u));
        while (nch > 1 && m / nch < 4096) --nch;
    } else if (tables && m > ((size_t)1 << 20)) {
        // A queued MSM (a batch, a prover round) of more than 2^20 terms: chunks of <= 2^20 terms one after the other on the
        // MSM's own stream, all adding into the same buckets.  Not for overlap (the other lanes provide that) but for the
        // sort's shape: above 2^20 terms a level-1 entry has too few bits left for the low bucket bits, the segment count
        // passes 8192 and the sort falls back to the three-launch scan and the direct scatter (rounds 1-5: every
        // commitment of a 2^22-row proof).
        nch = (uint32_t)std::min<size_t>((m + ((size_t)1 << 20) - 1) >> 20, MSM_MAX_CHUNKS);
    }
    // chunk k = terms [cut(k), cut(k + 1)): equal shares, or a first chunk of its own size and equal shares of the rest:
    //  * scalars in HOST memory (typlonk_msm_g1): the first chunk's copy over PCIe is the exposed one, so it is 2^18 terms (8 MB)
    //    instead of 2^19 and there is one chunk more -- 2.73-2.75 -> 2.59-2.65 ms per 2^20-term commitment
    //    (profiles/r06_ab_host_scalar_path.txt, calls U and V); device-resident scalars keep equal chunks (an unequal first
    //    chunk loses there: profiles/r06_ab_first_chunk_and_rc2.txt);
    //  * TYPLONK_MSM_FIRST_PCT: that share of the terms (experiments).
    size_t first = 0;
    if (standalone && nch > 1 && ctx->msm_first_pct > 0 && ctx->msm_first_pct < 100) {
        first = std::max<size_t>(4096, (m / 100 * (size_t)ctx->msm_first_pct) & ~(size_t)63);
    } else if (overlap_host_first(ctx, standalone, h_scalars, nch, m)) {
        first = (size_t)1 << 18;
        nch = std::min<uint32_t>(nch + 1, MSM_MAX_CHUNKS);
    }
    const size_t step = first ? (m - first + nch - 2) / (nch - 1) : (m + nch - 1) / nch;
    auto cut = [&](uint32_t k) { return k == 0 ? (size_t)0 : std::min(m, first ? first + (size_t)(k - 1) * step : (size_t)k * step); };
    hipStream_t s = ws.stream;
    int rc;
    const bool overlap = nch > 1 && standalone;   // chunk k + 1 sorted on the side stream while chunk k accumulates
    if (overlap) {
        // the side stream carries the sorts of the chunks after the first: short, latency-bound kernels beside an
        // accumulation that fills every wavefront slot (a high stream priority for it was measured: no effect)
        if (!ws.side) HIPCHK(hipStreamCreateWithFlags(&ws.side, hipStreamNonBlocking));
        if (!ws.ev_in) HIPCHK(hipEventCreateWithFlags(&ws.ev_in, hipEventDisableTiming));
        for (uint32_t k = 0; k < nch; ++k) {
            if (!ws.ev_sorted[k]) HIPCHK(hipEventCreateWithFlags(&ws.ev_sorted[k], hipEventDisableTiming));
            if (!ws.ev_acc[k]) HIPCHK(hipEventCreateWithFlags(&ws.ev_acc[k], hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ws.ev_in, s));  // the scalars (and whatever produced them) are ordered on s
        HIPCHK(hipStreamWaitEvent(ws.side, ws.ev_in, 0));
    }
    if ((rc = ensure(ctx, ws.buckets, nb * 192))) return rc;
    uint32_t* buckets = (uint32_t*)ws.buckets.p;

    for (uint32_t k = 0; k < nch; ++k) {
        const size_t off = cut(k);
        if (off >= m) break;
        const size_t mk = cut(k + 1) - off;
        const Fr* sc = d_scalars + off;
        const uint32_t* pts = srs.d_points + off * PT_WORDS;  // chunk-local term index i -> base off + i (table t: + t*len)
        SortBufs& sb = ws.sb[k & 1];
        hipStream_t ss = (overlap && k > 0) ? ws.side : s;   // the first sort has nothing to overlap with
        if (overlap && k >= 2) HIPCHK(hipStreamWaitEvent(ss, ws.ev_acc[k - 2], 0));  // sb[k & 1] is free again
        // the first chunk's sort is the exposed one: the second chunk's sort starts behind it (it then has the whole first
        // accumulation to hide under) instead of beside it, where it doubled its time (profiles/r03_msm_2_20_timeline.txt)
        if (overlap && k == 1) HIPCHK(hipStreamWaitEvent(ss, ws.ev_sorted[0], 0));
        if (h_scalars)
            HIPCHK(hipMemcpyAsync(const_cast<Fr*>(sc), h_scalars + 4 * off, mk * sizeof(Fr), hipMemcpyHostToDevice, ss));
        const uint64_t total = (uint64_t)W * mk;
        if ((rc = ensure(ctx, sb.keys, total * 4))) return rc;
        if ((rc = ensure(ctx, sb.sorted, total * 4))) return rc;
        if ((rc = ensure(ctx, sb.counts, nb * 4))) return rc;
        if ((rc = ensure(ctx, sb.offsets, (nb + 1) * 4))) return rc;
        if ((rc = ensure(ctx, sb.cursor, nb * 4))) return rc;
        if ((rc = ensure(ctx, sb.blocksums, (size_t)scan_blocks * 4))) return rc;
        if ((rc = ensure(ctx, sb.order, nb * 4))) return rc;
        if ((rc = ensure(ctx, sb.ohist, (size_t)msm_sched_words() * 4))) return rc;
        // heavy-bucket splitting: cap = entries one thread may sum; at most total/cap heavy buckets/tasks
        // 8 x the mean, at least 32 (round 2: 4 x the mean, at least 512).  The accumulate kernel's thread walks a bucket's
        // first cap entries one after the other -- 6.7 us each when it is the last one running -- so a few buckets of 500
        // were a 3.4-ms tail; and the factor is 8 because table mode is not uniform: the top window's 2^t digits land
        // on the first 2^t buckets of the shared set (c = 20: 2.2 x the mean there), which 4 x the mean would already
        // turn into heavy buckets now and then (measured: +0.3 ms per 2^20 MSM for the extra launch's work)
        const uint32_t cap = (uint32_t)std::max<uint64_t>(MSM_CAP_MIN, 8 * ((total + nb_used - 1) / nb_used));
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
        const bool segsort = seg.ok;   // (else: shapes the segmented sort cannot take -- more than 2^23 terms -- use the atomic counting sort)
        if (tables && !segsort) return fail(ctx, TYPLONK_ERR_LENGTH, "table-mode MSM shape not supported");  // unreachable
        if (segsort) {
            if ((rc = ensure(ctx, sb.blk_hist, seg.nmat * 4))) return rc;
            if ((rc = ensure(ctx, sb.blk_base, (seg.nmat + 1) * 4))) return rc;
            if ((rc = ensure(ctx, sb.blk_cnt, seg.nmat * 4))) return rc;
            if ((rc = ensure(ctx, sb.seg_start, seg.nseg * 8))) return rc;
            if ((rc = ensure(ctx, sb.blocksums, (size_t)((seg.nmat + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK + scan_blocks + seg.nseg) * 4))) return rc;
            blocksums = (uint32_t*)sb.blocksums.p;
            // The sort of an OVERLAPPED chunk (side stream, beside the previous chunk's accumulation) takes 256-thread level-1
            // workgroups -- one 70-register wavefront per SIMD fits next to two 200-register accumulation wavefronts; two
            // do not, and the kernel then waits for the accumulation to drain -- and a raised wavefront priority: its five
            // kernels finish in 0.14 ms instead of trailing the whole accumulation (0.95 ms), and the next accumulation
            // starts 6 us after the previous one instead of 56 (profiles/r06_ab_sort_prio.txt).  The exposed first sort has
            // the chip to itself and takes 512.  NOT for the queued MSMs of a batch: there it is neutral to slightly
            // negative (the sorts steal from another MSM's accumulation what they gain).  TYPLONK_MSM_L1_THREADS forces one.
            const bool beside = ctx->msm_sort_prio && ss != s;
            const uint32_t l1 = ctx->msm_l1_threads ? (uint32_t)ctx->msm_l1_threads : (beside ? 256u : 512u);
            StageTimer st(ctx, ss == s ? "msm_sort" : "msm_sort_overlapped", ss);
            launch_msm_segsort(sc, (uint64_t)mk, c, W, digit_v, (uint32_t)seg.hb, seg.ibits, tables ? (uint32_t)srs.len : 0u,
                               tables ? nsets : 0u, (uint32_t*)sb.blk_hist.p, (uint32_t*)sb.blk_base.p, blocksums,
                               (uint32_t*)sb.blk_cnt.p, (uint32_t*)sb.seg_start.p, keys, counts, offsets, sorted, cap,
                               (uint32_t*)sb.ohist.p, (uint32_t*)sb.heavy.p, (uint32_t*)sb.tasks.p, (uint32_t*)sb.order.p,
                               centred, ctx->msm_scatter_staged ? 1 : 0, l1, beside, ss);
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
        // The segmented sort ends with the bucket schedule (order[], msm_seg_place_kernel); only the atomic sort of the
        // shapes it cannot take needs the separate schedule launches.
        if (!segsort) {
            StageTimer st(ctx, "msm_order", ss);
            launch_bucket_order(counts, offsets, (uint32_t)nb_used, cap, (uint32_t*)sb.ohist.p, (uint32_t*)sb.order.p,
                                (uint32_t*)sb.heavy.p, (uint32_t*)sb.tasks.p, ss);
        }
        if (ss != s) {   // an overlapped chunk: the accumulation on the MSM's stream waits for the side stream's sort
            HIPCHK(hipEventRecord(ws.ev_sorted[k], ss));
            HIPCHK(hipStreamWaitEvent(s, ws.ev_sorted[k], 0));
        } else if (overlap && k == 0) {
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
            // chosen from the load; TYPLONK_MSM_LANES forces one class
            const uint32_t split = (lanes >= 2 && !ctx->msm_lanes) ? (uint32_t)(nb_used / 2) : (uint32_t)nb_used;
            const bool chain = !standalone && (ctx->msm_chain < 0 ? m >= MSM_CHAIN_MIN_TERMS : ctx->msm_chain != 0);
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
        if (overlap && k + 2 < nch) HIPCHK(hipEventRecord(ws.ev_acc[k], s));
    }
    // row/column bucket reduction (launch.hpp): c is in 8..20 and a plain MSM has at most 32 windows, so it always applies
    ws.rc = true;
    {
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
        uint32_t* planes_out = nsets == 1 ? ws.host_wins : (uint32_t*)ws.rc_out.p;
        StageTimer st(ctx, "msm_reduce", s);
        // two launches for small bucket sets, where the reduction is a latency chain; big sets are work-bound and the
        // four-launch form wastes fewer lanes (2^19 buckets: 0.39 ms against 0.49, profiles/r03_shard_variants.jsonl)
        if (!ctx->msm_rc4 && msm_rc2_ok(sh) && (ctx->msm_rc2_force || nb <= (1u << 17)))
            launch_msm_rc2_reduce(buckets, sh, (uint32_t*)ws.part_b.p, (uint32_t*)ws.part_a.p, planes_out,
                                  (uint32_t)ctx->msm_rc2_logw, s);
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
int msm_run(typlonk_ctx* ctx, uint32_t srs_id, const Fr* d_scalars, size_t m, uint64_t out_xy[12], uint8_t* out_inf, bool ptr_is_local,
            const uint64_t* h_scalars) {
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
    if ((rc = msm_enqueue(ctx, ctx->ws[0], ctx->stream, *srs, d_scalars, m, out_xy, out_inf, /*standalone=*/true, h_scalars))) return rc;
    if ((rc = msm_finish(ctx, ctx->ws[0]))) return rc;
    prof_collect(ctx);
    return TYPLONK_OK;
}

// ---- MsmQueue (host.hpp) --------------------------------------------------------------------------------------------
MsmQueue::MsmQueue(typlonk_ctx* c, const SrsEntry* s, int first_lane)
    : ctx(c), srs(s),
      lanes(std::max(1, std::min<int>(c->msm_inflight ? c->msm_inflight : (s->len < MSM_FOUR_LANES_BELOW ? 4 : 3), typlonk_ctx::MSM_LANES))),
      lane_lo(0), next(0) {
    set_first_lane(first_lane);
}
void MsmQueue::set_first_lane(int l) {
    lane_lo = (l < lanes) ? l : 0;
    if (next < lane_lo) next = lane_lo;
}
int MsmQueue::submit(const Fr* d_scalars, size_t m, uint64_t* out_xy, uint8_t* out_inf, bool standalone) {
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
    return msm_enqueue(ctx, ws, st, *srs, d_scalars + off, ml, out_xy, out_inf, standalone, nullptr);
}
int MsmQueue::wait_all() {
    int rc = TYPLONK_OK;
    for (int l = 0; l < typlonk_ctx::MSM_LANES; ++l) {
        const int r = msm_finish(ctx, ctx->ws[l]);
        if (!rc) rc = r;
    }
    return rc;
}

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

}  // namespace tyh

// (entry points: C linkage comes from their declarations in include/typlonk.h)

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
        const bool build_comb = !ctx->srs_comb_ready;   // [j 2^(8w)] G, once per context
        if (build_comb) {
            const int rc = ensure(ctx, ctx->srs_comb, srs_comb_bytes());
            if (rc != TYPLONK_OK) return rc;
            launch_srs_comb((uint32_t*)ctx->srs_comb.p, ctx->stream);
            HIPCHK(hipGetLastError());
        }
        launch_srs_generate(s, start, (uint64_t)len, (const uint32_t*)ctx->srs_comb.p, e.d_points, ctx->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->srs_comb_ready = true;   // only now: a failed build is repeated by the next call, never read
    }
    guard.dismiss();
    const uint32_t id = ctx->next_srs++;
    ctx->srs[id] = e;
    *srs_id = id;
    return TYPLONK_OK;
}

int typlonk_selftest_fq_inv(typlonk_ctx* ctx, uint64_t seed, size_t count, uint64_t* mismatches, uint32_t* max_rounds) {
    if (!ctx || !mismatches) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    uint32_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 16));
    DevGuard guard;
    guard.add(d);
    HIPCHK(hipMemsetAsync(d, 0, 16, ctx->stream));
    const uint32_t threads = 1u << 16;
    const uint32_t per = (uint32_t)((count + threads - 1) / threads);
    if (per) launch_fq_inv_selftest(seed, threads, per, d, ctx->stream);
    HIPCHK(hipGetLastError());
    uint32_t h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *mismatches = h[0];
    if (max_rounds) *max_rounds = h[1];
    return TYPLONK_OK;
}

int typlonk_srs_precompute(typlonk_ctx* ctx, uint32_t srs_id, uint32_t window_bits) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    auto it = ctx->srs.find(srs_id);
    if (it == ctx->srs.end()) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "unknown srs id");
    if (window_bits == 0) {
        // auto: 15 below 2^16 points, 17 below 2^19, else 20 (measured best: HISTORY.md sections 4, 6;
        // profiles/r04_tables_small_sizes.txt) -- and nothing at all for an SRS shorter
        // than 2^14 points: 2^16 buckets (sort, reduction, heavy-bucket launch) for a handful of terms would be slower
        // than the plain path, whose window follows the length
        if (it->second.len < TYPLONK_TABLES_AUTO_MIN_LEN) return TYPLONK_OK;
        window_bits = it->second.len < (1u << 16) ? 15 : (it->second.len < (1u << 19) ? 17 : 20);
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
    // scratch of the shared normalisation: one denominator per table entry, freed when the call returns
    uint32_t* zbuf = nullptr;
    HIPCHK(hipMalloc((void**)&zbuf, srs_tables_scratch_bytes(e.len, T)));
    DevGuard zguard;
    zguard.add(zbuf);
    launch_srs_tables(big, zbuf, (uint64_t)e.len, window_bits, T, ctx->stream);
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
    if (ml) {  // only this entry's share of the coefficients crosses PCIe -- chunk by chunk, beside the kernels (msm_enqueue)
        int rc = ensure(ctx, ctx->scal, ml * sizeof(Fr));
        if (rc) return rc;
    }
    return msm_run(ctx, srs_id, (const Fr*)ctx->scal.p, m, out_xy, out_inf, /*ptr_is_local=*/true, ml ? scalars + 4 * off : nullptr);
}

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
    // every point: the ranks' partial sums added in rank order (mixed additions: the records are affine), then ONE
    // inversion for all `count` results -- 97 -> 40 us for nine points from eight ranks, the same canonical points
    namespace H = h64;
    const H::Fq one = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull,
                        0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};  // 2^384 mod p
    std::vector<H::Xyzz> sums(count, H::inf());
    for (size_t i = 0; i < count; ++i) {
        for (size_t r = 0; r < world; ++r) {   // all-gather layout: rank-major, `count` records per rank
            const uint64_t* rec = records + (r * count + i) * COMM_REC;
            if (rec[12] & 1u) continue;
            H::Fq x, y;
            memcpy(x.v, rec, 48);
            memcpy(y.v, rec + 6, 48);
            sums[i] = H::xyzz_madd(sums[i], x, y, one);
        }
    }
    std::vector<char> ok(count ? count : 1);
    H::xyzz_to_affine_batch(sums.data(), count, out_xy, ok.data());
    for (size_t i = 0; i < count; ++i) {
        if (ok[i]) out_inf[i] = 0;
        else write_affine_out(G1Affine::inf(), out_xy + 12 * i, out_inf + i);
    }
    return TYPLONK_OK;
}

int typlonk_msm_plan(typlonk_ctx* ctx, size_t m, uint32_t* window_bits, uint32_t* n_windows, uint64_t* group_ops) {
    (void)ctx;   // the plan of a plain MSM depends on the length only
    uint32_t c, W;
    msm_shape(m ? m : 1, &c, &W);
    if (window_bits) *window_bits = c;
    if (n_windows) *n_windows = W;
    // Pippenger operation count for this shape: one mixed add per (term, window), two adds per
    // bucket in the running-sum reduction, c doublings per window in the final combine.
    if (group_ops) *group_ops = (uint64_t)W * m + 2ull * W * (1ull << (c - 1)) + (uint64_t)c * (W - 1);
    return TYPLONK_OK;
}

