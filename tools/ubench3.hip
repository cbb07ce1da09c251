// Round-2 experiment: would Fr on nine unsaturated 30-bit limbs (R' = 2^270) pay for the NTT?
// Measures, at the NTT kernel's occupancy, (a) the multiplication and (b) a whole radix-2 DIF butterfly
// (a' = x + y, b' = (x - y) * w) in the library's 8 x 32-bit form (ff.hpp, product scanning + assembly carry chains)
// and in a 9 x 30-bit form with a fused product + reduction (no carry instruction inside a column, m_k = -acc mod 2^30
// because r = 1 mod 2^30, lazy additions, one carry-normalisation before every multiplication).
// The 30-bit product is checked against the 32-bit one on random inputs first.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../typlonk_amd/csrc/ff.hpp"
using namespace ty;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

struct Fr30 { uint32_t v[9]; };
constexpr uint32_t M30 = 0x3fffffffu;
__host__ __device__ constexpr uint32_t r30(int i) {
    constexpr uint32_t t[9] = {0x1u, 0x3ffffffcu, 0x3fe5bfefu, 0x2f6900bfu, 0x21d80553u, 0x27602026u, 0x17d48333u, 0x29d4ca67u, 0x73edu};
    return t[i];
}
// a * b * 2^-270 mod r, inputs with (almost) normalised limbs; result < r + a*b / 2^270, normalised
__device__ __forceinline__ Fr30 fr30_mul(const Fr30& a, const Fr30& b) {
    uint32_t m[9];
    Fr30 o;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i < (k < 9 ? k : 9); ++i) acc += (uint64_t)m[i] * r30(k - i);
        if (k < 17) {
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
        }
        if (k < 9) {
            m[k] = (0u - (uint32_t)acc) & M30;  // -r^-1 = -1 mod 2^30
            acc += m[k];                       // m_k * r_0, r_0 = 1
        } else {
            o.v[k - 9] = (uint32_t)acc & M30;
        }
        acc >>= 30;
    }
    return o;
}
// 8 x 32 words (value < 2^256) <-> 9 x 30 limbs
__device__ __forceinline__ Fr30 fr30_unpack(const Fr& x) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t t = x.v[wi] >> sh;
        if (sh > 2 && wi + 1 < 8) t |= x.v[wi + 1] << (32 - sh);
        r.v[i] = t & M30;
    }
    return r;
}
__device__ __forceinline__ Fr fr30_pack(const Fr30& a) {
    Fr o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, li = bit / 30, off = bit % 30;
        uint32_t x = a.v[li] >> off;
        if (li + 1 < 9) x |= a.v[li + 1] << (30 - off);
        if (off > 28 && li + 2 < 9) x |= a.v[li + 2] << (60 - off);
        o.v[j] = x;
    }
    return o;
}
// lazy pieces of a butterfly
__device__ __forceinline__ Fr30 fr30_add_raw(const Fr30& a, const Fr30& b) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + 2^11 r limb-wise, every limb stays positive (bias in "spread" form: limbs ~2^31)
__host__ __device__ constexpr uint32_t bias30(int i) {
    // 2^11 * r, limbs n_i; spread: s_0 = n_0 + 2^31, s_i = n_i + 2^31 - 2 (0 < i < 8), s_8 = n_8 - 2
    constexpr uint32_t n[9] = {0x00000800u, 0x3fffe000u, 0x2dff7fffu, 0x0805fffcu, 0x002a9dedu, 0x0101343bu, 0x24199cecu, 0x26533afau, 0x039f6d3au};
    return i == 0 ? n[0] + 0x80000000u : (i < 8 ? n[i] + 0x7ffffffeu : n[8] - 2u);
}
__device__ __forceinline__ Fr30 fr30_sub_raw(const Fr30& a, const Fr30& b) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + bias30(i) - b.v[i];
    return r;
}
// one parallel carry step: limbs <= 2^30 + 3 afterwards (enough for the multiplier's column bound)
__device__ __forceinline__ Fr30 fr30_norm(const Fr30& a) {
    Fr30 r;
    r.v[0] = a.v[0] & M30;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.v[i] = (a.v[i] & M30) + (a.v[i - 1] >> 30);
    r.v[8] = a.v[8] + (a.v[7] >> 30);
    return r;
}

__global__ void check_kernel(uint32_t* bad) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int it = 0; it < 32; ++it) {
        Fr a, b;
        for (int i = 0; i < 8; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a.v[i] = (uint32_t)(s >> 11); s ^= s << 13; s ^= s >> 7; s ^= s << 17; b.v[i] = (uint32_t)(s >> 9); }
        a.v[7] &= 0x3fffffffu; b.v[7] &= 0x3fffffffu;      // < 2^254 < r
        // 8x32: a*b/2^256.  9x30 with the SAME words: a*b/2^270, so compare a*b*2^14/2^270... instead feed b * 2^14:
        // montmul30(a, b') with b' = b * 2^14 mod r computed by the 32-bit path (b * (2^14 * 2^256) / 2^256)
        Fr c14 = Fr::zero(); c14.v[0] = 1u << 14; c14 = fe_to_mont(c14);
        const Fr b14 = fe_mul(b, c14);
        const Fr want = fe_mul(a, b);
        Fr30 p = fr30_mul(fr30_unpack(a), fr30_unpack(b14));
        // canonical: p < 2r
        Fr got = fr30_pack(p);
        fe_reduce_once(got);
        uint32_t d = 0;
        for (int i = 0; i < 8; ++i) d |= got.v[i] ^ want.v[i];
        if (d) atomicAdd(bad, 1u);
    }
}

template <int V>
__global__ __launch_bounds__(256) void mul_kernel(uint32_t* out, int iters) {
    uint32_t s = 0;
    if (V == 0) {
        Fr a, b;
        for (int i = 0; i < 8; ++i) { a.v[i] = threadIdx.x * 77u + i * 13u + 1; b.v[i] = blockIdx.x * 31u + i * 7u + 3; }
        a.v[7] &= 0x3fffffffu; b.v[7] &= 0x3fffffffu;
        for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
        for (int i = 0; i < 8; ++i) s += a.v[i] ^ b.v[i];
    } else {
        Fr30 a, b;
        for (int i = 0; i < 9; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & M30; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & M30; }
        a.v[8] &= 0x3fffu; b.v[8] &= 0x3fffu;
        for (int i = 0; i < iters; ++i) { a = fr30_mul(a, b); b = fr30_mul(b, a); }
        for (int i = 0; i < 9; ++i) s += a.v[i] ^ b.v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// a chain of butterflies on register-resident values with a fixed twiddle: (x, y) <- (x + y, (x - y) * w)
template <int V>
__global__ __launch_bounds__(256) void bfly_kernel(uint32_t* out, int iters) {
    uint32_t s = 0;
    if (V == 0) {
        Fr x, y, w;
        for (int i = 0; i < 8; ++i) { x.v[i] = threadIdx.x * 77u + i * 13u + 1; y.v[i] = blockIdx.x * 31u + i * 7u + 3; w.v[i] = i * 0x01010101u + 5; }
        x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu; w.v[7] &= 0x3fffffffu;
        for (int i = 0; i < iters; ++i) {
            const Fr a = fe_add(x, y);
            const Fr b = fe_mul(fe_sub(x, y), w);
            x = b; y = a;
        }
        for (int i = 0; i < 8; ++i) s += x.v[i] ^ y.v[i];
    } else {
        Fr30 x, y, w;
        for (int i = 0; i < 9; ++i) { x.v[i] = (threadIdx.x * 77u + i * 13u + 1) & M30; y.v[i] = (blockIdx.x * 31u + i * 7u + 3) & M30; w.v[i] = (i * 0x01010101u + 5) & M30; }
        x.v[8] &= 0x3fffu; y.v[8] &= 0x3fffu; w.v[8] &= 0x3fffu;
        for (int i = 0; i < iters; ++i) {
            // a' is normalised every stage here (the NTT could do it every other stage); the value bound is reset by
            // feeding the product back as x (the real transform lets the a-chain grow to 2^k r inside a pass)
            // (timing only: the a-chain is not bounded here, the instruction stream is what the NTT would execute)
            const Fr30 a = fr30_norm(fr30_add_raw(x, y));
            const Fr30 b = fr30_mul(fr30_norm(fr30_sub_raw(x, y)), w);
            x = b;
            y = a;
        }
        for (int i = 0; i < 9; ++i) s += x.v[i] ^ y.v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static double time_ms(K&& launch, int reps = 3) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(a, 0); launch(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    uint32_t* out;
    CHK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    uint32_t* bad;
    CHK(hipMalloc(&bad, 4));
    CHK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check_kernel, dim3(256), dim3(256), 0, 0, bad);
    uint32_t h;
    CHK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    printf("== ubench3: Fr 8x32 (library) vs 9x30 fused; self-check mismatches: %u -> %s\n", h, h ? "FAIL" : "ok");
    if (h) return 2;
    const int threads = 256, it = 512;
    for (int occ = 8; occ >= 1; occ /= 2) {
        const int nb = prop.multiProcessorCount * occ;
        const double t0 = time_ms([&] { hipLaunchKernelGGL(mul_kernel<0>, dim3(nb), dim3(threads), 0, 0, out, it); });
        const double t1 = time_ms([&] { hipLaunchKernelGGL(mul_kernel<1>, dim3(nb), dim3(threads), 0, 0, out, it); });
        const double b0 = time_ms([&] { hipLaunchKernelGGL(bfly_kernel<0>, dim3(nb), dim3(threads), 0, 0, out, it); });
        const double b1 = time_ms([&] { hipLaunchKernelGGL(bfly_kernel<1>, dim3(nb), dim3(threads), 0, 0, out, it); });
        const double ops = (double)nb * threads * it * 1e-6;
        printf("@%d waves/SIMD: Fr mul 8x32 %.1f | 9x30 %.1f G/s;  butterfly 8x32 (add+sub+mul) %.1f G/s | 9x30 (add+sub+2 norm+mul) %.1f G/s\n",
               occ, 2 * ops / t0, 2 * ops / t1, ops / b0, ops / b1);
    }
    return 0;
}
