// Integer-pipe micro-benchmarks for gfx950: decides the limb representation of the Fp/Fr library
// (the integer-multiply issue rate is not in the microarchitecture guide).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../typlonk_amd/csrc/g1.hpp"
#include "mul_variants.hpp"
using namespace ty;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int CH = 8;  // independent chains per lane

template <int OP>
__global__ __launch_bounds__(256) void op_kernel(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x * 3u + 1u, b = blockIdx.x * 7u + 5u;
    uint64_t acc[CH];
    uint32_t x[CH];
    double d[CH];
    for (int j = 0; j < CH; ++j) { acc[j] = a + j; x[j] = b + j; d[j] = 1.0 + j; }
    double da = 1.0000001, db = 0.5;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
            if (OP == 3) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[j]) : "v"(a) : "vcc");
            if (OP == 4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(da), "v"(db));
            if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[j]) : "v"(a));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[j]) : "v"(a));
            if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
            if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[j]) : "v"(a));
            if (OP == 9) asm volatile("v_lshrrev_b64 %0, 30, %0" : "+v"(acc[j]));
            if (OP == 10) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[j]) : "v"(acc[(j + 1) % CH]));
            if (OP == 11) asm volatile("v_alignbit_b32 %0, %1, %0, 30" : "+v"(x[j]) : "v"(a));
            if (OP == 12) asm volatile("v_and_b32 %0, 0x3fffffff, %0" : "+v"(x[j]));
            if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(a));
        }
    }
    uint32_t s = 0;
    for (int j = 0; j < CH; ++j) s += (uint32_t)acc[j] + (uint32_t)(acc[j] >> 32) + x[j] + (uint32_t)d[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
__global__ __launch_bounds__(256) void fmul_kernel(uint32_t* out, int iters) {
    F a, b;
    for (int i = 0; i < F::N; ++i) { a.v[i] = threadIdx.x * 77u + i * 13u + 1; b.v[i] = blockIdx.x * 31u + i * 7u + 3; }
    a.v[F::N - 1] &= 0x0fffffffu; b.v[F::N - 1] &= 0x0fffffffu;
    for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < F::N; ++i) s += a.v[i] ^ b.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
__global__ __launch_bounds__(256) void fq_variant_kernel(uint32_t* out, int iters) {
    uint32_t s = 0;
    if (V == 2) {
        Fq a, b;
        for (int i = 0; i < 12; ++i) { a.v[i] = threadIdx.x * 77u + i * 13u + 1; b.v[i] = blockIdx.x * 31u + i * 7u + 3; }
        a.v[11] &= 0x0fffffffu; b.v[11] &= 0x0fffffffu;
        for (int i = 0; i < iters; ++i) { a = fe_mul_ps(a, b); b = fe_mul_ps(b, a); }
        for (int i = 0; i < 12; ++i) s += a.v[i] ^ b.v[i];
    } else if (V == 6 || V == 7) {
        Fq30 a, b;
        for (int i = 0; i < 13; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK; }
        for (int i = 0; i < iters; ++i) {
            if (V == 6) { a = fq30_mul_v6(a, b); b = fq30_mul_v6(b, a); } else { a = fq30_mul_v7(a, b); b = fq30_mul_v7(b, a); }
        }
        for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i];
    } else if (V == 5) {
        Fq30 a, b;
        for (int i = 0; i < 13; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK; }
        for (int i = 0; i < iters; ++i) { a = fq30_mul_v5(a, b); b = fq30_mul_v5(b, a); }
        for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i];
    } else if (V == 4) {
        Fq30 a, b;
        for (int i = 0; i < 13; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK; }
        for (int i = 0; i < iters; ++i) { a = fq30_mul_ilp(a, b); b = fq30_mul_ilp(b, a); }
        for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i];
    } else {
        Fq30 a, b;
        for (int i = 0; i < 13; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK; }
        for (int i = 0; i < iters; ++i) { a = fq30_mul(a, b); b = fq30_mul(b, a); }
        for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i];
    }
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
        g1_madd(acc, p, (i & 1) != 0);
        idx = (idx * 1664525u + 1013904223u) % npts;
    }
    uint32_t s = 0;
    for (int i = 0; i < 12; ++i) s += acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
    out[t] = s;
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
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 4096;
    uint32_t* out;
    CHK(hipMalloc(&out, (size_t)blocks * threads * 4 * 4));
    const char* names[14] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co+addc (2 ops)", "v_fma_f64",
                            "v_mul_u32_u24", "v_mad_u32_u24", "v_add_u32", "v_mul_hi_u32_u24", "v_lshrrev_b64", "v_lshl_add_u64",
                            "v_alignbit_b32", "v_and_b32 (literal)", "v_mov_b32"};
    double ms[14];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(op_kernel<0>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[1] = time_ms([&] { hipLaunchKernelGGL(op_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[2] = time_ms([&] { hipLaunchKernelGGL(op_kernel<2>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[3] = time_ms([&] { hipLaunchKernelGGL(op_kernel<3>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[4] = time_ms([&] { hipLaunchKernelGGL(op_kernel<4>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[5] = time_ms([&] { hipLaunchKernelGGL(op_kernel<5>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[6] = time_ms([&] { hipLaunchKernelGGL(op_kernel<6>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[7] = time_ms([&] { hipLaunchKernelGGL(op_kernel<7>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[8] = time_ms([&] { hipLaunchKernelGGL(op_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[9] = time_ms([&] { hipLaunchKernelGGL(op_kernel<9>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[10] = time_ms([&] { hipLaunchKernelGGL(op_kernel<10>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[11] = time_ms([&] { hipLaunchKernelGGL(op_kernel<11>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[12] = time_ms([&] { hipLaunchKernelGGL(op_kernel<12>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    ms[13] = time_ms([&] { hipLaunchKernelGGL(op_kernel<13>, dim3(blocks), dim3(threads), 0, 0, out, iters); });
    for (int i = 0; i < 14; ++i) {
        const double ops = (double)blocks * threads * iters * CH;
        const double waves = ops / 64.0;
        // cycles per wave-instruction per SIMD at the nominal clock: SIMDs * clk * t / wave-instrs
        const double cyc = (double)prop.multiProcessorCount * 4 * (prop.clockRate * 1e3) * (ms[i] * 1e-3) / waves;
        printf("%-24s %8.3f ms  %8.2f Gop/s(lane)  ~%.2f cyc/wave-instr/SIMD\n", names[i], ms[i], ops / ms[i] * 1e-6, cyc);
    }
    {
        const int it = 256;
        double t = time_ms([&] { hipLaunchKernelGGL(fmul_kernel<Fq>, dim3(blocks), dim3(threads), 0, 0, out, it); });
        printf("Fq mul (12x32 CIOS)      %8.3f ms  %8.2f G mul/s\n", t, (double)blocks * threads * it * 2 / t * 1e-6);
        for (int occ = 8; occ >= 1; occ /= 2) {
            const int nb = prop.multiProcessorCount * occ;
            double t1 = time_ms([&] { hipLaunchKernelGGL(fmul_kernel<Fq>, dim3(nb), dim3(threads), 0, 0, out, it); });
            double t2 = time_ms([&] { hipLaunchKernelGGL(fq_variant_kernel<2>, dim3(nb), dim3(threads), 0, 0, out, it); });
            double t3 = time_ms([&] { hipLaunchKernelGGL(fq_variant_kernel<3>, dim3(nb), dim3(threads), 0, 0, out, it); });
            double t4 = time_ms([&] { hipLaunchKernelGGL(fq_variant_kernel<4>, dim3(nb), dim3(threads), 0, 0, out, it); });
            double t5 = time_ms([&] { hipLaunchKernelGGL(fq_variant_kernel<5>, dim3(nb), dim3(threads), 0, 0, out, it); });
            double t6 = time_ms([&] { hipLaunchKernelGGL(fq_variant_kernel<6>, dim3(nb), dim3(threads), 0, 0, out, it); });
            double t7 = time_ms([&] { hipLaunchKernelGGL(fq_variant_kernel<7>, dim3(nb), dim3(threads), 0, 0, out, it); });
            printf("Fq mul @%d blocks/CU: 13x30 with mad-add of T[k] %.2f | + indep columns %.2f | all-asm chain %.2f G mul/s\n", occ,
                   (double)nb * threads * it * 2 / t5 * 1e-6, (double)nb * threads * it * 2 / t6 * 1e-6,
                   (double)nb * threads * it * 2 / t7 * 1e-6);
            printf("Fq mul @%d blocks/CU: CIOS %.2f | product-scan asm %.2f | 13x30 %.2f | 13x30 indep-columns %.2f  G mul/s\n", occ,
                   (double)nb * threads * it * 2 / t1 * 1e-6, (double)nb * threads * it * 2 / t2 * 1e-6,
                   (double)nb * threads * it * 2 / t3 * 1e-6, (double)nb * threads * it * 2 / t4 * 1e-6);
        }
        t = time_ms([&] { hipLaunchKernelGGL(fmul_kernel<Fr>, dim3(blocks), dim3(threads), 0, 0, out, it); });
        printf("Fr mul (8x32 CIOS)       %8.3f ms  %8.2f G mul/s\n", t, (double)blocks * threads * it * 2 / t * 1e-6);
    }
    {
        // points: multiples of a fake generator are not needed for timing; use the Montgomery generator repeated
        const uint32_t npts = 1 << 16;
        uint32_t* pts;
        CHK(hipMalloc(&pts, (size_t)npts * 96));
        uint32_t* h = (uint32_t*)malloc((size_t)npts * 96);
        uint64_t st = 88172645463325252ull;
        for (size_t i = 0; i < (size_t)npts * 24; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (uint32_t)st; if (i % 12 == 11) h[i] &= 0x0fffffff; }
        CHK(hipMemcpy(pts, h, (size_t)npts * 96, hipMemcpyHostToDevice));
        const int it = 64;
        for (int wpb = 1; wpb <= 4; wpb *= 2) {
            const int nb = prop.multiProcessorCount * 4 * 2 / wpb;  // 2 waves per SIMD worth of work
            double t;
            if (wpb == 1) t = time_ms([&] { hipLaunchKernelGGL(madd_kernel<64>, dim3(nb), dim3(64), 0, 0, pts, npts, out, it); });
            else if (wpb == 2) t = time_ms([&] { hipLaunchKernelGGL(madd_kernel<128>, dim3(nb), dim3(128), 0, 0, pts, npts, out, it); });
            else t = time_ms([&] { hipLaunchKernelGGL(madd_kernel<256>, dim3(nb), dim3(256), 0, 0, pts, npts, out, it); });
            printf("XYZZ mixed add (block=%3d) %8.3f ms  %8.3f G adds/s  (%.1f us per add per wave)\n", wpb * 64, t,
                   (double)nb * wpb * 64 * it / t * 1e-6, t * 1e3 / it);
        }
    }
    return 0;
}
