// libtyplonk_hip.so -- the RCCL exchange behind the C ABI (librccl is dlopen-ed on first use)
// Part of the host driver of include/typlonk.h (see host.hpp for the shared state).  There is deliberately no CPU compute
// fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "host.hpp"

using namespace ty;
using namespace tyh;

#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen'ed on first use (typlonk_comm_*)

#include <dlfcn.h>

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
}  // namespace

namespace tyh {

// ---- RCCL exchange ------------------------------------------------------------------------------------------------
#define NCCLCHK(expr)                                                                                                 \
    do {                                                                                                              \
        ncclResult_t _r = (expr);                                                                                     \
        if (_r != ncclSuccess)                                                                                        \
            return fail(ctx, TYPLONK_ERR_COMM, std::string(#expr) + ": " + rccl_api()->GetErrorString(_r));           \
    } while (0)

void comm_release(typlonk_ctx* ctx) {
    Comm& c = ctx->comm;
    if (c.comm) (void)rccl_api()->CommDestroy((ncclComm_t)c.comm);
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
int comm_fold(typlonk_ctx* ctx, uint64_t* xy, uint8_t* inf, size_t count, int local_rc) {
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
    NCCLCHK(rccl_api()->AllGather(c.d_send, c.d_recv, count * COMM_REC, ncclUint64, (ncclComm_t)c.comm, s));
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


}  // namespace tyh

// (entry points: C linkage comes from their declarations in include/typlonk.h)

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

