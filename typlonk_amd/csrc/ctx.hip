// libtyplonk_hip.so -- context, workspaces, profiling events, device vectors
// Part of the host driver of include/typlonk.h (see host.hpp for the shared state).  There is deliberately no CPU compute
// fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "host.hpp"

using namespace ty;
using namespace tyh;

namespace tyh {

int fail(typlonk_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

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

}  // namespace tyh

// (entry points: C linkage comes from their declarations in include/typlonk.h)

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
    if (const char* e = getenv("TYPLONK_MSM_REDUCE")) {
        ctx->msm_rc4 = (strcmp(e, "rc4") == 0);
        ctx->msm_rc2_force = (strcmp(e, "rc2") == 0);
    }
    if (const char* e = getenv("TYPLONK_MSM_RC2_LOGW")) ctx->msm_rc2_logw = std::max(8, std::min(atoi(e), 14));
    if (const char* e = getenv("TYPLONK_MSM_SCATTER")) ctx->msm_scatter_staged = strcmp(e, "direct") != 0;
    if (const char* e = getenv("TYPLONK_MSM_L1_THREADS")) ctx->msm_l1_threads = atoi(e) == 256 ? 256 : (atoi(e) == 512 ? 512 : 0);
    if (const char* e = getenv("TYPLONK_MSM_SORT_PRIO")) ctx->msm_sort_prio = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_MSM_CHAIN")) ctx->msm_chain = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("TYPLONK_MSM_LANES")) {
        const int l = atoi(e);
        if (l == 1 || l == 2 || l == 4 || l == 8 || l == 16) ctx->msm_lanes = l;
    }
    if (const char* e = getenv("TYPLONK_MSM_INFLIGHT")) ctx->msm_inflight = atoi(e);
    if (const char* e = getenv("TYPLONK_MSM_CHUNKS")) ctx->msm_chunks = std::max(0, std::min(atoi(e), MSM_MAX_CHUNKS));
    if (const char* e = getenv("TYPLONK_MSM_FIRST_PCT")) ctx->msm_first_pct = std::max(0, std::min(atoi(e), 99));
    if (const char* e = getenv("TYPLONK_PROVER_PIPE")) ctx->prover_pipe = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_PROVER_FETCH")) ctx->prover_pinned_slots = atoi(e) != 0;
    if (const char* e = getenv("TYPLONK_PROVER_NTT_BATCH")) ctx->prover_ntt_batch = std::max(0, std::min(atoi(e), 3));
    if (const char* e = getenv("TYPLONK_NTT_FR30")) ctx->ntt_fr30 = std::max(0, std::min(atoi(e), 2));
    if (const char* e = getenv("TYPLONK_NTT_BIG")) ctx->ntt_big = std::max(0, std::min(atoi(e), 2));
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
    if (ctx->eval_slots_host) (void)hipHostFree(ctx->eval_slots_host);
    for (DevBuf* b : {&ctx->srs_comb, &ctx->scal, &ctx->ntt_scratch, &ctx->ntt_io, &ctx->quot_ext, &ctx->quot_tab, &ctx->ops_tmp, &ctx->prover_mem}) release(*b);
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

int typlonk_set_profiling(typlonk_ctx* ctx, int on) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    ctx->profiling = on != 0;
    ctx->prof_light = on == 2;
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

