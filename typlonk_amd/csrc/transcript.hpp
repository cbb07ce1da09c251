// Fiat-Shamir transcript of the reference prover, native host side.
//
// Restates plonk::proof::challenges::ChallengeGenerator (/root/reference/plonk/src/proof/challenges.rs:9-46):
//   digest(c)               ark-serialize `serialize_unchecked` bytes of the G1Affine commitment (:17-22)
//   generate_challenges<N>  Blake2b-512 over the bytes (:31-34), first 8 bytes little-endian -> u64 (:35-37) ->
//                           StdRng::seed_from_u64 (:38) -> N x Fr::rand (:40-45)
// and is what typlonk_prove uses between the prover rounds.  Everything below the Blake2b call lives in crates that
// are not in this container (ark-serialize / ark-ec / ark-ff 0.3.0, rand 0.8.4 = rand_chacha 0.3.1 ChaCha12,
// rand_core 0.6.3; pins in /root/reference/Cargo.lock:28-29, 42-43, 95-96, 435-436, 447-448, 457-458) and there is no
// Rust toolchain here: the code follows the published crate behaviour and is NOT verified against the reference.  It is
// pinned against the independent Python statement (tests/transcript_ref.py, tests/test_host.py), Blake2b against
// hashlib and RFC 7693's "abc" vector, the ChaCha block function against RFC 7539.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "ff.hpp"

namespace ty {

// ---- Blake2b-512, unkeyed (RFC 7693) -------------------------------------------------------------------------------
inline uint64_t tr_rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
inline void blake2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t0, uint64_t t1, bool last) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    static const uint8_t S[12][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
                                      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) {
        uint64_t w = 0;
        for (int b = 7; b >= 0; --b) w = (w << 8) | block[8 * i + b];
        m[i] = w;
    }
    for (int i = 0; i < 8; ++i) {
        v[i] = h[i];
        v[8 + i] = IV[i];
    }
    v[12] ^= t0;
    v[13] ^= t1;
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] = v[a] + v[b] + x; v[d] = tr_rotr64(v[d] ^ v[a], 32);
        v[c] = v[c] + v[d];     v[b] = tr_rotr64(v[b] ^ v[c], 24);
        v[a] = v[a] + v[b] + y; v[d] = tr_rotr64(v[d] ^ v[a], 16);
        v[c] = v[c] + v[d];     v[b] = tr_rotr64(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = S[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[8 + i];
}
inline void blake2b_512(const uint8_t* data, size_t len, uint8_t out[64]) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = IV[i];
    h[0] ^= 0x01010000ull ^ 64ull;  // digest length 64, no key, fanout 1, depth 1
    uint8_t block[128];
    size_t off = 0;
    while (len - off > 128) {
        blake2b_compress(h, data + off, (uint64_t)(off + 128), 0, false);
        off += 128;
    }
    memset(block, 0, sizeof(block));
    if (len > off) memcpy(block, data + off, len - off);
    blake2b_compress(h, block, (uint64_t)len, 0, true);
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
}

// ---- rand_core seed_from_u64 (PCG32) -> rand_chacha ChaCha12Rng = rand 0.8 StdRng ---------------------------------
inline uint32_t tr_rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
struct StdRng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int have = 0;  // words left in buf
    // keyed directly with 256 bits (rand's thread_rng is this generator seeded from the operating system)
    static StdRng from_key(const uint32_t k[8]) {
        StdRng r(0);
        for (int i = 0; i < 8; ++i) r.key[i] = k[i];
        return r;
    }
    explicit StdRng(uint64_t state) {
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            const uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    void block() {
        uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; ++i) init[4 + i] = key[i];
        init[12] = (uint32_t)counter;
        init[13] = (uint32_t)(counter >> 32);
        init[14] = 0;
        init[15] = 0;
        uint32_t x[16];
        for (int i = 0; i < 16; ++i) x[i] = init[i];
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = tr_rotl32(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = tr_rotl32(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = tr_rotl32(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = tr_rotl32(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 6; ++r) {  // 12 rounds
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) buf[i] = x[i] + init[i];
        ++counter;
        have = 16;
    }
    uint32_t next_u32() {
        if (!have) block();
        return buf[16 - have--];
    }
    uint64_t next_u64() {
        const uint64_t lo = next_u32();
        return lo | ((uint64_t)next_u32() << 32);
    }
};

// ark-ff Fp256::rand: four u64 limbs, top bit cleared, retry while >= r; the limbs ARE the Montgomery representation
inline void fr_rand(StdRng& rng, uint64_t out[4]) {
    static const uint64_t R[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    for (;;) {
        for (int i = 0; i < 4; ++i) out[i] = rng.next_u64();
        out[3] &= 0x7fffffffffffffffull;
        bool less = false;
        for (int i = 3; i >= 0; --i) {
            if (out[i] != R[i]) {
                less = out[i] < R[i];
                break;
            }
        }
        if (less) return;
    }
}

// serialize_unchecked of a G1Affine in the C-ABI form: x, y as 48 canonical little-endian bytes each, SWFlags in the two
// top bits of the last byte (infinity = 0x40); the identity is (0, 1) + the flag
inline void serialize_unchecked_g1(const uint64_t xy[12], uint8_t inf, uint8_t out[96]) {
    memset(out, 0, 96);
    if (inf) {
        out[48] = 1;
        out[95] |= 0x40;
        return;
    }
    for (int c = 0; c < 2; ++c) {
        Fq m;
        memcpy(m.v, xy + 6 * c, 48);
        const Fq canon = fe_from_mont(m);
        memcpy(out + 48 * c, canon.v, 48);  // little-endian host
    }
}

struct ChallengeGenerator {
    std::vector<uint8_t> data;
    void digest(const uint64_t xy[12], uint8_t inf) {
        uint8_t rec[96];
        serialize_unchecked_g1(xy, inf, rec);
        data.insert(data.end(), rec, rec + 96);
    }
    // n challenges, 4 Montgomery limbs each
    void generate(size_t n, uint64_t* out) const {
        uint8_t h[64];
        blake2b_512(data.data(), data.size(), h);
        uint64_t seed = 0;
        for (int b = 7; b >= 0; --b) seed = (seed << 8) | h[b];
        StdRng rng(seed);
        for (size_t i = 0; i < n; ++i) fr_rand(rng, out + 4 * i);
    }
};

}  // namespace ty
