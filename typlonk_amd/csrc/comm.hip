// libtyplonk_hip.so -- the RCCL exchange behind the C ABI (librccl is dlopen-ed on first use)
// Part of the host driver of include/typlonk.h (see host.hpp for the shared state).  There is deliberately no CPU compute
// fallback: without a HIP device typlonk_init fails with TYPLONK_ERR_NO_DEVICE.
#include "host.hpp"

using namespace ty;
using namespace tyh;

#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen'ed on first use (typlonk_comm_*)

#include <dlfcn.h>

#include <atomic>
#include <mutex>

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
void rccl_resolve(RcclApi& api);
// first use from any thread resolves the entry points exactly once (a Rust caller keeps one Backend per thread, so two
// threads may reach their first typlonk_comm_* call together)
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { rccl_resolve(api); });
    return &api;
}
void rccl_resolve(RcclApi& api) {
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
        return;
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

// Fault injection for the one local failure between "decided to fold" and the collective that cannot be provoked from
// outside: a lost staging copy.  Compiled ONLY into the test build of the library (-DTYPLONK_TEST_HOOKS,
// typlonk_amd/build.py build_hip_test_hooks -> tests/cpp/hooks/libtyplonk_hip.so); the shipped library has no such switch.
//   TYPLONK_TEST_COMM_FAIL_STAGING=<k>: the k-th fold of this process (1-based) behaves as if its staging copy had failed
// (tests/test_gpu_dist.py: the peers must get TYPLONK_ERR_COMM, not a hang, and the next fold must work).
#ifdef TYPLONK_TEST_HOOKS
static bool comm_test_fail_staging() {
    static const int target = [] { const char* e = getenv("TYPLONK_TEST_COMM_FAIL_STAGING"); return e ? atoi(e) : 0; }();
    static std::atomic<int> calls{0};
    return target > 0 && ++calls == target;
}
#else
static constexpr bool comm_test_fail_staging() { return false; }
#endif

void comm_release(typlonk_ctx* ctx) {
    Comm& c = ctx->comm;
    if (c.comm) (void)rccl_api()->CommDestroy((ncclComm_t)c.comm);
    if (c.d_send) (void)hipFree(c.d_send);
    if (c.d_recv) (void)hipFree(c.d_recv);
    if (c.h_buf) (void)hipHostFree(c.h_buf);
    c = Comm{};
}

// The exchange buffers are allocated ONCE, by typlonk_comm_init BEFORE it joins ncclCommInitRank (COMM_CAP records: more
// than the 9 points of a prover round): a rank that cannot allocate never becomes a member, and a fold never allocates,
// so no member can fail locally between "decided to fold" and the collective and leave its peers waiting.  Longer point
// lists go through in pieces of COMM_CAP records (comm_fold).
// The send buffer is kept POISONED -- every flag word all ones, which the fold reads as "this rank failed" -- except
// between a successful staging copy and the all-gather that follows it: if the copy fails, the rank still joins the
// collective and what its peers receive says so.
constexpr size_t COMM_CAP = 32;
int comm_reserve(typlonk_ctx* ctx, int world) {
    Comm& c = ctx->comm;
    HIPCHK(hipMalloc((void**)&c.d_send, COMM_CAP * COMM_REC * 8));
    HIPCHK(hipMalloc((void**)&c.d_recv, (size_t)world * COMM_CAP * COMM_REC * 8));
    HIPCHK(hipHostMalloc((void**)&c.h_buf, (size_t)(world + 1) * COMM_CAP * COMM_REC * 8));
    HIPCHK(hipMemset(c.d_send, 0xff, COMM_CAP * COMM_REC * 8));
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
    // From here to the all-gather there is no early return: whatever goes wrong locally, the rank joins the collective.
    // A staging copy that fails leaves the poisoned send buffer in place (comm_reserve), i.e. flagged records.
    hipError_t stage = comm_test_fail_staging() ? hipErrorUnknown : hipMemcpyAsync(c.d_send, out, count * COMM_REC * 8, hipMemcpyHostToDevice, s);
    const ncclResult_t gathered = rccl_api()->AllGather(c.d_send, c.d_recv, count * COMM_REC, ncclUint64, (ncclComm_t)c.comm, s);
    (void)hipMemsetAsync(c.d_send, 0xff, c.cap * COMM_REC * 8, s);   // poisoned again for the next fold
    if (stage != hipSuccess) {
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        return fail(ctx, TYPLONK_ERR_HIP, std::string("staging the records for the exchange: ") + hipGetErrorString(stage) +
                                              " (the peers were told: this rank's records went out flagged)");
    }
    if (gathered != ncclSuccess) return fail(ctx, TYPLONK_ERR_COMM, std::string("ncclAllGather: ") + rccl_api()->GetErrorString(gathered));
    HIPCHK(hipMemcpyAsync(back, c.d_recv, (size_t)c.world * count * COMM_REC * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (local_rc) return fail(ctx, local_rc, local_err);   // (its flagged records made every peer fail too)
    int failed = -1;
    rc = typlonk_g1_fold_records_host(back, (size_t)c.world, count, xy, inf, &failed);
    if (rc == TYPLONK_ERR_COMM) {
        const uint64_t flag = back[(size_t)failed * count * COMM_REC + 12];
        if ((flag >> 32) == 0xffffffffull)
            return fail(ctx, rc, "rank " + std::to_string(failed) + " could not stage its records for the exchange (a HIP error on that rank)");
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
    // Buffers first: a rank that cannot allocate fails HERE, before it is a member -- as a rank that never called. Once
    // ncclCommInitRank has returned on every rank there is nothing left that can fail on one of them alone.
    const int rc = comm_reserve(ctx, world);
    if (rc) {
        const std::string msg = ctx->err;
        comm_release(ctx);
        return fail(ctx, rc, msg);
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api->CommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) {
        comm_release(ctx);
        return fail(ctx, TYPLONK_ERR_COMM, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
    }
    ctx->comm.comm = comm;
    ctx->comm.rank = rank;
    ctx->comm.world = world;
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
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (!ctx->comm.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "no communicator on this context (typlonk_comm_init)");
    // a member's bad argument / device error is a local failure that still joins the collective (flagged records)
    int rc = TYPLONK_OK;
    const hipError_t he = hipSetDevice(ctx->device);
    if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
    if (xy && inf) return comm_fold(ctx, xy, inf, count, rc);
    if (!count) return rc;
    if (!rc) rc = fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    std::vector<uint64_t> spare_xy(12 * count, 0);
    std::vector<uint8_t> spare_inf(count, 1);
    return comm_fold(ctx, spare_xy.data(), spare_inf.data(), count, rc);
}

int typlonk_msm_g1_sharded_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* d_scalars, size_t m, uint64_t out_xy[12],
                                  uint8_t* out_inf) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (!ctx->comm.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "no communicator on this context (typlonk_comm_init)");
    // A member of the communicator: from here every path ends in the collective.  A bad argument or a device error is a
    // LOCAL failure like any other -- it travels with flagged records instead of leaving the peers inside ncclAllGather.
    int rc = TYPLONK_OK;
    const hipError_t he = hipSetDevice(ctx->device);
    if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
    uint64_t spare_xy[12] = {0};
    uint8_t spare_inf = 1;
    if (!rc && (!out_xy || !out_inf)) rc = fail(ctx, TYPLONK_ERR_INVALID_ARG, "null output");
    if (!rc) rc = msm_run(ctx, srs_id, (const Fr*)d_scalars, m, out_xy, out_inf);
    const bool have_out = out_xy && out_inf;
    return comm_fold(ctx, have_out ? out_xy : spare_xy, have_out ? out_inf : &spare_inf, 1, rc);
}

int typlonk_msm_g1_sharded_batch_devptr(typlonk_ctx* ctx, uint32_t srs_id, const void* const* d_scalars, const size_t* m,
                                        size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    if (!ctx) return TYPLONK_ERR_INVALID_ARG;
    if (!ctx->comm.comm) return fail(ctx, TYPLONK_ERR_INVALID_ARG, "no communicator on this context (typlonk_comm_init)");
    // as above: every path of a member ends in the collective (`count` is what the ranks agree on)
    int rc = TYPLONK_OK;
    const hipError_t he = hipSetDevice(ctx->device);
    if (he != hipSuccess) rc = fail(ctx, TYPLONK_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
    if (!rc && (!out_xy || !out_inf || !m || !d_scalars)) rc = fail(ctx, TYPLONK_ERR_INVALID_ARG, "null argument");
    if (!rc) rc = msm_batch(ctx, srs_id, d_scalars, m, count, out_xy, out_inf);
    if (out_xy && out_inf) return comm_fold(ctx, out_xy, out_inf, count, rc);   // one collective for the whole group
    std::vector<uint64_t> spare_xy(12 * std::max<size_t>(count, 1), 0);
    std::vector<uint8_t> spare_inf(std::max<size_t>(count, 1), 1);
    return comm_fold(ctx, spare_xy.data(), spare_inf.data(), count, rc);
}

