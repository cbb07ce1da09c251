// TEST INFRASTRUCTURE -- a stand-in for librccl that lets N processes SHARING ONE GPU run the library's native exchange
// (typlonk_comm_*, typlonk_msm_g1_sharded_*, typlonk_prove on shards: typlonk_amd/csrc/comm.hip) with world > 1.
// RCCL itself refuses two ranks on one device, and the test box has one GPU; the product selects the library to load
// through TYPLONK_RCCL_LIB, which tests/test_gpu_dist.py points here.  Never loaded by the product on its own.
//   seam: KzgScheme::evaluate_in_s returns the FULL sum       /root/reference/kzg/src/lib.rs:41-54
//
// Exports the five entry points comm.hip resolves: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy,
// ncclGetErrorString -- with the prototypes of <rccl/rccl.h>.  The "fabric" is a POSIX shared-memory segment named in the
// unique id: a header with a sense-reversing barrier and one slot per rank.  ncclAllGather is stream-ordered like the
// real one as far as its caller can tell: it waits for the stream (the staging copy before it), copies the send buffer
// from the device into its slot, meets the other ranks, copies every slot into the receive buffer on the device, and
// meets them again before the slots are reused.  Every wait is bounded (FAKE_RCCL_TIMEOUT_S, default 120 s): a rank that
// never arrives turns into ncclSystemError on the others, not into a hung test.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t SLOT_BYTES = 64 * 1024;
constexpr int MAX_WORLD = 64;

struct Shm {
    std::atomic<uint32_t> arrived;   // barrier: ranks that reached it in this generation
    std::atomic<uint32_t> gen;       // barrier generation
    std::atomic<uint32_t> members;   // ranks that have joined (diagnostics)
    std::atomic<uint32_t> gathers;   // all-gathers completed (diagnostics; read by the tests through FAKE_RCCL_STATS)
    uint32_t world;
    uint32_t pad[11];
    unsigned char slots[MAX_WORLD][SLOT_BYTES];
};
struct FakeComm {
    Shm* shm = nullptr;
    int rank = 0, world = 0;
    char name[64] = {0};
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double timeout_s() {
    const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
    return e ? atof(e) : 120.0;
}
// all `world` ranks meet; false on timeout
bool barrier(FakeComm* c) {
    Shm* s = c->shm;
    const uint32_t g = s->gen.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->gen.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    const double t0 = now_s(), lim = timeout_s();
    while (s->gen.load(std::memory_order_acquire) == g) {
        if (now_s() - t0 > lim) return false;
        usleep(20);
    }
    return true;
}
size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned r = 0;
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) {
        (void)!fread(&r, sizeof(r), 1, f);
        fclose(f);
    }
    snprintf(id->internal, sizeof(id->internal), "/typlonk_fake_rccl_%d_%08x", (int)getpid(), r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_WORLD || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Shm)) != 0) {   // every rank sizes it the same; new pages are zero
        close(fd);
        return ncclSystemError;
    }
    void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    FakeComm* c = new FakeComm();
    c->shm = (Shm*)p;
    c->rank = rank;
    c->world = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->shm->world = (uint32_t)nranks;
    c->shm->members.fetch_add(1);
    if (!barrier(c)) {   // collective, like the real one: returns when everybody has joined
        munmap(p, sizeof(Shm));
        delete c;
        return ncclSystemError;
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    FakeComm* c = (FakeComm*)comm;
    const size_t bytes = sendcount * dtype_bytes(datatype);
    if (!c || !sendbuff || !recvbuff || bytes == 0 || bytes > SLOT_BYTES) return ncclInvalidArgument;
    // the caller's earlier work on the stream (the staging copy) first; its failure is the caller's to report -- this
    // rank still goes through both barriers, or the others would wait for it
    (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
    bool ok = hipMemcpy(c->shm->slots[c->rank], sendbuff, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    if (!barrier(c)) return ncclSystemError;
    std::vector<unsigned char> all(bytes * (size_t)c->world);
    for (int r = 0; r < c->world; ++r) memcpy(all.data() + (size_t)r * bytes, c->shm->slots[r], bytes);
    ok = ok && hipMemcpy(recvbuff, all.data(), all.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!barrier(c)) return ncclSystemError;   // nobody overwrites a slot somebody is still reading
    if (c->rank == 0) c->shm->gathers.fetch_add(1);
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = (FakeComm*)comm;
    if (!c) return ncclInvalidArgument;
    if (const char* path = getenv("FAKE_RCCL_STATS")) {   // "<gathers> <members>" for the test that asked
        if (c->rank == 0) {
            if (FILE* f = fopen(path, "w")) {
                fprintf(f, "%u %u\n", c->shm->gathers.load(), c->shm->members.load());
                fclose(f);
            }
        }
    }
    if (c->rank == 0) shm_unlink(c->name);   // the mappings of the other ranks stay valid until they unmap
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP copy failed";
        case ncclSystemError: return "fake rccl: shared-memory rendezvous failed or timed out";
        case ncclInvalidArgument: return "fake rccl: invalid argument";
        default: return "fake rccl: error";
    }
}

}  // extern "C"
