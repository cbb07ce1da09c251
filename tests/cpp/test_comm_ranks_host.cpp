// The native exchange with world > 1 from plain C++ processes -- no Python, no PyTorch (the path a Rust host takes,
// INTEGRATION.md section 5).  `test_comm_ranks_host <world> <scratch dir>` starts <world> fresh copies of itself BEFORE it
// touches the GPU; every copy is one rank on GPU 0: own context, own SRS shard, own communicator
// (TYPLONK_RCCL_LIB = tests/cpp/libfake_rccl.so carries the all-gather: RCCL refuses several ranks per device).
// Each rank checks, against a plain full-length SRS it also holds, that the sharded calls return the FULL sum bit for
// bit, that a batch longer than one exchange piece does, and the failure protocol: rank 1 passes m > len and gets
// TYPLONK_ERR_LENGTH, every other rank TYPLONK_ERR_COMM, and the next call works.
//   seam: KzgScheme::evaluate_in_s returns the FULL sum       /root/reference/kzg/src/lib.rs:41-54
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/typlonk.h"

#define REQUIRE(c)                                                                       \
    do {                                                                                 \
        if (!(c)) {                                                                      \
            std::printf("rank %d FAILED %s:%d: %s\n", g_rank, __FILE__, __LINE__, #c);   \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)
#define OK(call)                                                                                                          \
    do {                                                                                                                  \
        int _rc = (call);                                                                                                 \
        if (_rc) {                                                                                                        \
            std::printf("rank %d FAILED %s:%d: %s -> %d (%s)\n", g_rank, __FILE__, __LINE__, #call, _rc, typlonk_last_error(ctx)); \
            std::exit(1);                                                                                                 \
        }                                                                                                                 \
    } while (0)

static int g_rank = -1;

static int run_rank(int rank, int world, const std::string& dir) {
    g_rank = rank;
    typlonk_ctx* ctx = nullptr;
    REQUIRE(typlonk_init(&ctx, 0) == TYPLONK_OK);
    REQUIRE(typlonk_comm_available() == 1);
    // rendezvous id: rank 0 writes it, the others wait for the file
    uint8_t id[TYPLONK_COMM_ID_BYTES];
    const std::string path = dir + "/uid.bin", tmp = path + ".tmp";
    if (rank == 0) {
        REQUIRE(typlonk_comm_unique_id(id) == TYPLONK_OK);
        FILE* f = std::fopen(tmp.c_str(), "wb");
        REQUIRE(f && std::fwrite(id, 1, sizeof(id), f) == sizeof(id));
        std::fclose(f);
        REQUIRE(std::rename(tmp.c_str(), path.c_str()) == 0);
    } else {
        FILE* f = nullptr;
        for (int i = 0; i < 12000 && !(f = std::fopen(path.c_str(), "rb")); ++i) usleep(10000);
        REQUIRE(f && std::fread(id, 1, sizeof(id), f) == sizeof(id));
        std::fclose(f);
    }
    OK(typlonk_comm_init(ctx, id, rank, world));
    int r = -1, w = -1;
    OK(typlonk_comm_info(ctx, &r, &w));
    REQUIRE(r == rank && w == world);

    const size_t n = 1 << 12, total = n + 3;
    const uint64_t secret[4] = {0x0123456789abcdefull, 0x0fedcba987654321ull, 0x1122334455667788ull, 0x0102030405060708ull};
    const size_t lo = (size_t)rank * total / world, hi = (size_t)(rank + 1) * total / world;
    uint32_t plain = 0, shard = 0;
    OK(typlonk_srs_generate(ctx, secret, 0, total, &plain));
    OK(typlonk_srs_generate(ctx, secret, lo, hi - lo, &shard));
    OK(typlonk_srs_set_shard(ctx, shard, lo, total));
    std::vector<uint64_t> sc(4 * n);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < 4 * n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        sc[i] = (i % 4 == 3) ? (x >> 3) : x;
    }
    typlonk_buf* buf = nullptr;
    OK(typlonk_buf_alloc(ctx, n, &buf));
    OK(typlonk_buf_upload(ctx, buf, 0, sc.data(), n));
    const void* dp = typlonk_buf_devptr(buf);
    // single MSMs
    for (size_t m : {n, n - 1, (size_t)1, (size_t)0}) {
        uint64_t a[12], b[12];
        uint8_t ai = 0, bi = 0;
        OK(typlonk_msm_g1_devptr(ctx, plain, dp, m, a, &ai));
        OK(typlonk_msm_g1_sharded_devptr(ctx, shard, dp, m, b, &bi));
        REQUIRE(ai == bi && std::memcmp(a, b, sizeof(a)) == 0);
    }
    // a batch of 40 (> 32 records: goes through the exchange in two pieces)
    const size_t K = 40;
    std::vector<const void*> ptrs(K, dp);
    std::vector<size_t> ms(K);
    for (size_t i = 0; i < K; ++i) ms[i] = i % 5 == 4 ? 0 : n - 13 * i;
    std::vector<uint64_t> bx(12 * K), by(12 * K);
    std::vector<uint8_t> bxi(K), byi(K);
    OK(typlonk_msm_g1_batch_devptr(ctx, plain, ptrs.data(), ms.data(), K, bx.data(), bxi.data()));
    OK(typlonk_msm_g1_sharded_batch_devptr(ctx, shard, ptrs.data(), ms.data(), K, by.data(), byi.data()));
    REQUIRE(bx == by && bxi == byi);
    // failure protocol
    uint64_t a[12], b[12];
    uint8_t ai = 0, bi = 0;
    const int rc = typlonk_msm_g1_sharded_devptr(ctx, shard, dp, rank == 1 ? total + 1 : n, b, &bi);
    REQUIRE(rc == (rank == 1 ? TYPLONK_ERR_LENGTH : TYPLONK_ERR_COMM));
    if (rank != 1) REQUIRE(std::strstr(typlonk_last_error(ctx), "rank 1") != nullptr);
    // a null output on rank 0 is a local failure too: it must not strand the others
    const int rc2 = typlonk_msm_g1_sharded_devptr(ctx, shard, dp, n, rank == 0 ? nullptr : b, &bi);
    REQUIRE(rc2 == (rank == 0 ? TYPLONK_ERR_INVALID_ARG : TYPLONK_ERR_COMM));
    OK(typlonk_msm_g1_devptr(ctx, plain, dp, n, a, &ai));
    OK(typlonk_msm_g1_sharded_devptr(ctx, shard, dp, n, b, &bi));
    REQUIRE(ai == bi && std::memcmp(a, b, sizeof(a)) == 0);
    OK(typlonk_buf_free(ctx, buf));
    OK(typlonk_comm_destroy(ctx));
    typlonk_destroy(ctx);
    std::printf("rank %d of %d ok\n", rank, world);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 5 && std::strcmp(argv[1], "rank") == 0) return run_rank(std::atoi(argv[2]), std::atoi(argv[3]), argv[4]);
    if (argc != 3) {
        std::printf("usage: %s <world> <scratch dir>\n", argv[0]);
        return 2;
    }
    // the launcher: nothing here touches the GPU; the ranks are fresh processes (fork + exec of this binary)
    const int world = std::atoi(argv[1]);
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid == 0) {
            const std::string rs = std::to_string(r), ws = std::to_string(world);
            execl(argv[0], argv[0], "rank", rs.c_str(), ws.c_str(), argv[2], (char*)nullptr);
            _exit(127);
        }
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
    }
    std::printf(bad ? "%d rank(s) failed\n" : "all %d ranks ok\n", bad ? bad : world);
    return bad ? 1 : 0;
}
