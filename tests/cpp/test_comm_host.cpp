// The exchange behind the C ABI from a plain C++ process -- no Python, no PyTorch: the library dlopens the system's
// librccl itself (the path a Rust host takes).  One rank on the box's GPU: the fold over a
// one-rank communicator must return what the local calls return, through ncclAllGather and the host fold.
//   seam: KzgScheme::evaluate_in_s returns the FULL sum       /root/reference/kzg/src/lib.rs:41-54
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/typlonk.h"

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) {                                                         \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);      \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)
#define OK(call)                                                                                            \
    do {                                                                                                    \
        int _rc = (call);                                                                                   \
        if (_rc) {                                                                                          \
            std::printf("FAILED %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, _rc, typlonk_last_error(ctx)); \
            std::exit(1);                                                                                   \
        }                                                                                                   \
    } while (0)

int main() {
    typlonk_ctx* ctx = nullptr;
    REQUIRE(typlonk_init(&ctx, 0) == TYPLONK_OK);
    int rank = -1, world = -1;
    OK(typlonk_comm_info(ctx, &rank, &world));
    REQUIRE(world == 0);
    uint64_t xy[12] = {0};
    uint8_t inf = 1;
    REQUIRE(typlonk_comm_fold_g1(ctx, xy, &inf, 1) == TYPLONK_ERR_INVALID_ARG);   // no communicator yet
    REQUIRE(typlonk_comm_available() == 1);                                        // librccl loads here (not collective)
    uint8_t id[TYPLONK_COMM_ID_BYTES];
    REQUIRE(typlonk_comm_unique_id(id) == TYPLONK_OK);
    OK(typlonk_comm_init(ctx, id, 0, 1));
    REQUIRE(typlonk_comm_init(ctx, id, 0, 1) == TYPLONK_ERR_INVALID_ARG);          // one communicator per context
    OK(typlonk_comm_info(ctx, &rank, &world));
    REQUIRE(rank == 0 && world == 1);

    const size_t n = 1 << 12;
    const uint64_t secret[4] = {0x0123456789abcdefull, 0x0fedcba987654321ull, 0x1122334455667788ull, 0x0102030405060708ull};
    uint32_t plain = 0, shard = 0;
    OK(typlonk_srs_generate(ctx, secret, 0, n, &plain));
    OK(typlonk_srs_generate(ctx, secret, 1000, 2000, &shard));          // bases [1000, 3000) of the same SRS
    OK(typlonk_srs_set_shard(ctx, shard, 1000, n));
    std::vector<uint64_t> sc(4 * n);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < 4 * n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        sc[i] = (i % 4 == 3) ? (x >> 3) : x;                             // < 2^253: a valid Montgomery residue
    }
    typlonk_buf* buf = nullptr;
    OK(typlonk_buf_alloc(ctx, n, &buf));
    OK(typlonk_buf_upload(ctx, buf, 0, sc.data(), n));
    uint64_t a[12], b[12];
    uint8_t ai = 0, bi = 0;
    OK(typlonk_msm_g1_devptr(ctx, shard, typlonk_buf_devptr(buf), n, a, &ai));           // the rank's partial sum
    OK(typlonk_msm_g1_sharded_devptr(ctx, shard, typlonk_buf_devptr(buf), n, b, &bi));   // ... folded over one rank
    REQUIRE(ai == bi && std::memcmp(a, b, sizeof(a)) == 0);
    // the partial sum of [1000, 3000) = full sum over 3000 terms - sum over the first 1000, checked through the fold
    uint64_t p3[12], p1[12];
    uint8_t i3 = 0, i1 = 0;
    OK(typlonk_msm_g1_devptr(ctx, plain, typlonk_buf_devptr(buf), 3000, p3, &i3));
    OK(typlonk_msm_g1_devptr(ctx, plain, typlonk_buf_devptr(buf), 1000, p1, &i1));
    uint64_t two[24];
    uint8_t twoinf[2] = {bi, i1};
    std::memcpy(two, b, 96);
    std::memcpy(two + 12, p1, 96);
    uint64_t sum[12];
    uint8_t suminf = 0;
    REQUIRE(typlonk_g1_sum_host(two, twoinf, 2, sum, &suminf) == TYPLONK_OK);
    REQUIRE(suminf == i3 && std::memcmp(sum, p3, sizeof(sum)) == 0);
    // a batch of three, one of them empty, in one collective
    const void* ptrs[3] = {typlonk_buf_devptr(buf), typlonk_buf_devptr(buf), typlonk_buf_devptr(buf)};
    const size_t ms[3] = {n, 1500, 0};
    uint64_t bx[36], by[36];
    uint8_t bxi[3], byi[3];
    OK(typlonk_msm_g1_batch_devptr(ctx, shard, ptrs, ms, 3, bx, bxi));
    OK(typlonk_msm_g1_sharded_batch_devptr(ctx, shard, ptrs, ms, 3, by, byi));
    REQUIRE(std::memcmp(bx, by, sizeof(bx)) == 0 && std::memcmp(bxi, byi, 3) == 0 && byi[2] == 1);
    // a local failure travels with the collective instead of leaving the peers inside it
    REQUIRE(typlonk_msm_g1_sharded_devptr(ctx, shard, typlonk_buf_devptr(buf), n + 1, b, &bi) == TYPLONK_ERR_LENGTH);
    OK(typlonk_msm_g1_sharded_devptr(ctx, shard, typlonk_buf_devptr(buf), n, b, &bi));
    REQUIRE(ai == bi && std::memcmp(a, b, sizeof(a)) == 0);
    OK(typlonk_buf_free(ctx, buf));
    OK(typlonk_comm_destroy(ctx));
    OK(typlonk_comm_info(ctx, &rank, &world));
    REQUIRE(world == 0);
    typlonk_destroy(ctx);
    std::printf("all ok\n");
    return 0;
}
