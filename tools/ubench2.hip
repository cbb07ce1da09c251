// Round-2 micro-benchmark: fused vs split Fq30 multiplication, merged-reduction Y3, mixed-add ceiling.
// Self-checking: every variant is compared digit for digit with the split form on random and extreme inputs
// before anything is timed.
// Build (three binaries, A/B on the same box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench2.hip -o tools/ubench2
//   hipcc ... -DFQ30_SPLIT_MUL -DG1_SPLIT_Y3 tools/ubench2.hip -o tools/ubench2_split
//   hipcc ... -DG1_SPLIT_Y3 tools/ubench2.hip -o tools/ubench2_fused_only
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../typlonk_amd/csrc/g1.hpp"
using namespace ty;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t rng(uint64_t& s) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (uint32_t)(s >> 16);
}
// kind 0: uniform digits; 1: all digits 2^30-1; 2: digits 2^30-1 except a few random; 3: value < 8p-ish (top digit small)
__device__ Fq30 gen(uint64_t& s, int kind) {
    Fq30 r;
    for (int i = 0; i < 13; ++i) {
        uint32_t x = rng(s) & FQ30_MASK;
        if (kind == 1) x = FQ30_MASK;
        if (kind == 2 && (rng(s) & 3)) x = FQ30_MASK;
        r.v[i] = x;
    }
    if (kind == 3) r.v[12] &= 0x00ffffffu;
    return r;
}
__device__ bool same(const Fq30& a, const Fq30& b) {
    uint32_t d = 0;
    for (int i = 0; i < 13; ++i) d |= a.v[i] ^ b.v[i];
    return d == 0;
}

__global__ void check_kernel(uint32_t* bad, int rounds) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int it = 0; it < rounds; ++it) {
        const int ka = it & 3, kb = (it >> 2) & 3;
        const Fq30 a = gen(s, ka), b = gen(s, kb);
        if (!same(fq30_mul_split(a, b), fq30_mul_fused(a, b))) atomicAdd(&bad[0], 1u);
        if (!same(fq30_sqr_split(a), fq30_sqr_fused(a))) atomicAdd(&bad[1], 1u);
        if (!same(fq30_sqr_fused(a), fq30_mul_fused(a, a))) atomicAdd(&bad[2], 1u);
        // merged reduction: operands as the group law uses them (values of a few p)
        Fq30 c = gen(s, 3), d = gen(s, 3), e = gen(s, 3), f = gen(s, 3);
        c.v[12] &= 0x003fffffu; d.v[12] &= 0x003fffffu; e.v[12] &= 0x003fffffu; f.v[12] &= 0x003fffffu;
        const Fq30 lhs = fq30_canon(fq30_mul2_add(c, d, e, f));
        const Fq30 rhs = fq30_canon(fq30_add_lazy(fq30_mul_split(c, d), fq30_mul_split(e, f)));
        if (!same(lhs, rhs)) atomicAdd(&bad[3], 1u);
    }
}

template <int V>
__global__ __launch_bounds__(256) void mul_kernel(uint32_t* out, int iters) {
    Fq30 a, b;
    for (int i = 0; i < 13; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK; }
    for (int i = 0; i < iters; ++i) {
        if (V == 0) { a = fq30_mul_split(a, b); b = fq30_mul_split(b, a); }
        if (V == 1) { a = fq30_mul_fused(a, b); b = fq30_mul_fused(b, a); }
        if (V == 2) { a = fq30_sqr_split(a); b = fq30_sqr_split(b); }
        if (V == 3) { a = fq30_sqr_fused(a); b = fq30_sqr_fused(b); }
        if (V == 4) { a = fq30_mul2_add(a, b, b, a); b = fq30_mul2_add(b, a, a, b); }
    }
    uint32_t s = 0;
    for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void madd_kernel(const uint32_t* pts, uint32_t npts, uint32_t* out, int iters) {
    const uint32_t t = blockIdx.x * THREADS + threadIdx.x;
    G1Xyzz acc = G1Xyzz::inf();
    uint32_t idx = (t * 2654435761u) % npts;
    for (int i = 0; i < iters; ++i) {
        G1Affine p;
        const uint4* q = reinterpret_cast<const uint4*>(pts + (uint64_t)idx * 24);
        uint4 w[6];
        for (int k = 0; k < 6; ++k) w[k] = q[k];
        for (int k = 0; k < 3; ++k) { p.x.v[4*k] = w[k].x; p.x.v[4*k+1] = w[k].y; p.x.v[4*k+2] = w[k].z; p.x.v[4*k+3] = w[k].w; }
        for (int k = 0; k < 3; ++k) { p.y.v[4*k] = w[3+k].x; p.y.v[4*k+1] = w[3+k].y; p.y.v[4*k+2] = w[3+k].z; p.y.v[4*k+3] = w[3+k].w; }
        p.x.v[12] = 0; p.y.v[12] = 0;
        for (int k = 0; k < 12; ++k) { p.x.v[k] &= FQ30_MASK; p.y.v[k] &= FQ30_MASK; }
        g1_madd(acc, p, (i & 1) != 0);
        idx = (idx * 1664525u + 1013904223u) % npts;
    }
    uint32_t s = 0;
    for (int i = 0; i < 12; ++i) s += acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
    out[t] = s;
}

// full (XYZZ + XYZZ) additions: the bucket reduction's operation
__global__ __launch_bounds__(128) void add_kernel(uint32_t* out, int iters) {
    G1Xyzz a, b;
    for (int i = 0; i < 13; ++i) {
        a.x.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; a.y.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK;
        a.zz.v[i] = (threadIdx.x * 5u + i) & FQ30_MASK; a.zzz.v[i] = (threadIdx.x * 3u + i * 11u) & FQ30_MASK;
    }
    a.x.v[12] &= 0xffff; a.y.v[12] &= 0xffff; a.zz.v[12] &= 0xffff; a.zzz.v[12] &= 0xffff;
    b = a; b.x.v[0] ^= 5;
    for (int i = 0; i < iters; ++i) { a = g1_add(a, b); b = g1_add(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 12; ++i) s += a.x.v[i] ^ a.y.v[i] ^ b.zz.v[i] ^ b.zzz.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static double time_ms(K&& launch, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const char* cfg =
#if defined(FQ30_SPLIT_MUL)
        "split mul"
#else
        "fused mul"
#endif
#if defined(G1_SPLIT_Y3)
        " + split Y3";
#else
        " + merged Y3";
#endif
    printf("== ubench2 [%s] device: %s CUs=%d ==\n", cfg, prop.name, prop.multiProcessorCount);
    uint32_t* out;
    CHK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    {
        uint32_t* bad;
        CHK(hipMalloc(&bad, 16));
        CHK(hipMemset(bad, 0, 16));
        hipLaunchKernelGGL(check_kernel, dim3(256), dim3(256), 0, 0, bad, 64);
        uint32_t h[4];
        CHK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
        printf("self-check (4.2M cases each): mul fused!=split %u | sqr fused!=split %u | sqr!=mul(a,a) %u | mul2_add %u  -> %s\n",
               h[0], h[1], h[2], h[3], (h[0] | h[1] | h[2] | h[3]) ? "FAIL" : "ok");
        if (h[0] | h[1] | h[2] | h[3]) return 2;
    }
    const int threads = 256, it = 256;
    for (int occ = 8; occ >= 1; occ /= 2) {
        const int nb = prop.multiProcessorCount * occ;
        double t[5];
        t[0] = time_ms([&] { hipLaunchKernelGGL(mul_kernel<0>, dim3(nb), dim3(threads), 0, 0, out, it); });
        t[1] = time_ms([&] { hipLaunchKernelGGL(mul_kernel<1>, dim3(nb), dim3(threads), 0, 0, out, it); });
        t[2] = time_ms([&] { hipLaunchKernelGGL(mul_kernel<2>, dim3(nb), dim3(threads), 0, 0, out, it); });
        t[3] = time_ms([&] { hipLaunchKernelGGL(mul_kernel<3>, dim3(nb), dim3(threads), 0, 0, out, it); });
        t[4] = time_ms([&] { hipLaunchKernelGGL(mul_kernel<4>, dim3(nb), dim3(threads), 0, 0, out, it); });
        const double ops = (double)nb * threads * it * 2 * 1e-6;
        printf("@%d waves/SIMD: mul split %.2f fused %.2f | sqr split %.2f fused %.2f | mul2_add %.2f  G op/s\n", occ,
               ops / t[0], ops / t[1], ops / t[2], ops / t[3], ops / t[4]);
    }
    {
        const uint32_t npts = 1 << 16;
        uint32_t* pts;
        CHK(hipMalloc(&pts, (size_t)npts * 96));
        uint32_t* h = (uint32_t*)malloc((size_t)npts * 96);
        uint64_t st = 88172645463325252ull;
        for (size_t i = 0; i < (size_t)npts * 24; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (uint32_t)st; }
        CHK(hipMemcpy(pts, h, (size_t)npts * 96, hipMemcpyHostToDevice));
        const int itm = 64;
        for (int wps = 1; wps <= 2; ++wps)
            for (int wpb = 1; wpb <= 4; wpb *= 2) {
                const int nb = prop.multiProcessorCount * 4 * wps / wpb;
                double t;
                if (wpb == 1) t = time_ms([&] { hipLaunchKernelGGL(madd_kernel<64>, dim3(nb), dim3(64), 0, 0, pts, npts, out, itm); });
                else if (wpb == 2) t = time_ms([&] { hipLaunchKernelGGL(madd_kernel<128>, dim3(nb), dim3(128), 0, 0, pts, npts, out, itm); });
                else t = time_ms([&] { hipLaunchKernelGGL(madd_kernel<256>, dim3(nb), dim3(256), 0, 0, pts, npts, out, itm); });
                printf("XYZZ mixed add, %d waves/SIMD, block=%3d: %8.3f ms  %6.3f G adds/s\n", wps, wpb * 64, t,
                       (double)nb * wpb * 64 * itm / t * 1e-6);
            }
        const int nb = prop.multiProcessorCount * 4;
        double t = time_ms([&] { hipLaunchKernelGGL(add_kernel, dim3(nb), dim3(128), 0, 0, out, 32); });
        printf("XYZZ full add, 2 waves/SIMD: %8.3f ms  %6.3f G adds/s\n", t, (double)nb * 128 * 32 * 2 / t * 1e-6);
    }
    return 0;
}
