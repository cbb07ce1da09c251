// Round-4 pricing micro-benchmark for gfx950: the issue cost of every VALU instruction the hot kernels use, measured so
// that the number does NOT depend on a wall-clock timing at a nominal frequency:
//   * cycles come from s_memtime (tick = shader cycle, /opt/skills/guides/MI355X_MICROARCH.md:446), read by every wavefront
//     around its own loop; the effective clock of the run is reported next to it from wall_clock64() (100 MHz);
//   * occupancy sweep: W = 1, 2, 3, 4, 6, 8 wavefronts per SIMD (256-thread workgroups = one wavefront per SIMD of a CU,
//     W workgroups per CU enforced through the LDS allocation, one wave of workgroups over the 256 CUs), so the figure
//     at which the issue port saturates is visible: cycles per wave-instruction per SIMD = elapsed / (W * instructions);
//   * 8 independent dependency chains per lane.
// Also: (a) the FP64-FMA alternative for the limb product (review item 3-i): the instruction mix of one 8 x 8 product on
// 52-bit limbs (2 v_fma_f64 + 2 64-bit adds per partial product) against the 13 x 13 product on 30-bit limbs (169
// v_mad_u64_u32 + 25 normalisations) -- pipe cost only, no arithmetic meaning;  (b) the cost of one SIMT field inversion
// in multiplication times (review item 3-ii, batched-affine accumulation).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench4.hip -o tools/ubench4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "../typlonk_amd/csrc/g1.hpp"
using namespace ty;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int CH = 8;      // independent chains per lane
constexpr int REP = 8;     // CH * REP instructions per loop iteration

struct Rec {
    uint64_t t0, t1, wall;
    uint32_t hw_id, xcc_id;   // HW_REG_HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; HW_REG_XCC_ID: xcc[3:0]
};
__device__ __forceinline__ void rec_store(Rec* out, uint64_t t0, uint64_t t1, uint64_t w0, uint64_t w1, uint32_t sink) {
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r.t0 = t0;
        r.t1 = t1 + (sink == 0x12345u ? 1 : 0);
        r.wall = w1 - w0;
        r.hw_id = hw;
        r.xcc_id = xcc;
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
    }
}

enum Op { MAD64 = 0, MUL_LO, MUL_HI, ADD, AND, MOV, LSHR64, LSHL_ADD64, ADDC, ADD3, SUB, ASHR, LSHL, CNDMASK, FMA64, ALIGNBIT, MAD24, XOR, OR3, N_OPS };
static const char* OP_NAMES[N_OPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32", "v_and_b32", "v_mov_b32", "v_lshrrev_b64",
                                      "v_lshl_add_u64", "v_addc_co_u32", "v_add3_u32", "v_sub_u32", "v_ashrrev_i32", "v_lshlrev_b32",
                                      "v_cndmask_b32", "v_fma_f64", "v_alignbit_b32", "v_mad_u32_u24", "v_xor_b32", "v_or3_b32"};

template <int OP>
__global__ __launch_bounds__(256) void op_kernel(Rec* out, int iters) {
    extern __shared__ char lds_hold[];   // sized by the host so that exactly W workgroups fit a CU
    uint32_t a = threadIdx.x * 3u + 1u, b = blockIdx.x * 7u + 5u;
    uint64_t acc[CH];
    uint32_t x[CH];
    double d[CH];
    for (int j = 0; j < CH; ++j) { acc[j] = a + j; x[j] = b + j; d[j] = 1.0 + j; }
    double da = 1.0000001, db = 0.5;
    __syncthreads();
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
                if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                if (OP == AND) asm volatile("v_and_b32 %0, 0x3fffffff, %0" : "+v"(x[j]));
                if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(a));
                if (OP == LSHR64) asm volatile("v_lshrrev_b64 %0, 30, %0" : "+v"(acc[j]));
                if (OP == LSHL_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[j]) : "v"(acc[(j + 1) % CH]));
                if (OP == ADDC) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[j]) : "v"(a) : "vcc");
                if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
                if (OP == SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                if (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x[j]));
                if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[j]));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(a) : "vcc");
                if (OP == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(da), "v"(db));
                if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %0, 30" : "+v"(x[j]) : "v"(a));
                if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[j]) : "v"(a));
                if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[j]) : "v"(a));
                if (OP == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    uint32_t s = 0;
    for (int j = 0; j < CH; ++j) s += (uint32_t)acc[j] + (uint32_t)(acc[j] >> 32) + x[j] + (uint32_t)d[j];
    rec_store(out, t0, t1, w0, w1, s);
}

// ---- instruction mixes (pipe cost only) -------------------------------------------------------------------------------
// MIX 0: one 13 x 13 product on 30-bit limbs as fq30 forms it: 169 mads in 25 columns, each closed by and + 64-bit shift
// MIX 1: one 8 x 8 product on 52-bit limbs through the FP64 pipe: per partial product hi = fma(a, b, C), lo = fma(a, b, -hi')
//        and two 64-bit integer additions of the bit patterns into the column sums (v_lshl_add_u64)
// MIX 2: MIX 1 with the two additions as v_add_co / v_addc pairs (4 x 32-bit)
template <int MIX>
__global__ __launch_bounds__(256) void mix_kernel(Rec* out, int iters) {
    extern __shared__ char lds_hold[];
    uint32_t a = threadIdx.x * 3u + 1u, b = blockIdx.x * 7u + 5u;
    uint64_t acc0 = a, acc1 = b, s0 = 1, s1 = 2;
    uint32_t dig = 0;
    double da = 1.0000001, db = 0.5, h = 1.0, l = 2.0;
    __syncthreads();
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MIX == 0) {
#pragma unroll
            for (int k = 0; k < 25; ++k) {
                const int terms = k < 13 ? k + 1 : 25 - k;
#pragma unroll
                for (int t = 0; t < terms; ++t) {
                    if (t & 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                    else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
                }
                asm volatile("v_and_b32 %0, 0x3fffffff, %1" : "=v"(dig) : "v"((uint32_t)acc0));
                asm volatile("v_lshrrev_b64 %0, 30, %0" : "+v"(acc0));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc0) : "v"(acc1));
            }
        } else {
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(h) : "v"(da), "v"(db));
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(l) : "v"(da), "v"(db));
                if (MIX == 1) {
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(s0) : "v"(acc0));
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(s1) : "v"(acc1));
                } else {
                    asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
                    asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(dig) : "v"(b) : "vcc");
                }
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    const uint32_t s = (uint32_t)acc0 + (uint32_t)acc1 + dig + (uint32_t)s0 + (uint32_t)s1 + (uint32_t)h + (uint32_t)l + a;
    rec_store(out, t0, t1, w0, w1, s);
}

// ---- field-level costs in cycles: one Fq multiplication, one mixed addition, one SIMT inversion (Fermat ladder) -------
// KIND 0: 64 dependent fq30_mul; 1: 64 dependent fq30_sqr; 2: 16 g1_madd on a register accumulator; 3: one fq30_inv_fermat
template <int KIND, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void field_kernel(Rec* out, int iters) {
    extern __shared__ char lds_hold[];
    Fq30 a, b;
    for (int i = 0; i < 13; ++i) { a.v[i] = (threadIdx.x * 77u + i * 13u + 1) & FQ30_MASK; b.v[i] = (blockIdx.x * 31u + i * 7u + 3) & FQ30_MASK; }
    a.v[12] &= 0xffff; b.v[12] &= 0xffff;
    G1Xyzz acc = G1Xyzz::inf();
    __syncthreads();
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { for (int k = 0; k < 32; ++k) { a = fq30_mul(a, b); b = fq30_mul(b, a); } }
        if (KIND == 1) { for (int k = 0; k < 64; ++k) a = fq30_sqr(a); }
        if (KIND == 2) {
            for (int k = 0; k < 16; ++k) {
                g1_madd_xy(acc, a, b);
                a.v[0] = (a.v[0] + 12345u) & FQ30_MASK;    // (not a curve point: the formulas run all the same)
            }
        }
        if (KIND == 3) a = fq30_inv_fermat(fq30_add_lazy(a, b));
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    uint32_t s = 0;
    for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i] ^ acc.x.v[i] ^ acc.zz.v[i];
    rec_store(out, t0, t1, w0, w1, s);
}


struct Result {
    double cyc_per_unit_per_simd, mhz, waves_per_simd;
};

// Per physical SIMD (xcc, se, sh, cu, simd from the hardware id registers): units executed by the wavefronts that ran on
// it / (last end - first start).  The figure reported is the median over the SIMDs that held the intended number of
// wavefronts at once -- it does not depend on how the dispatcher spread the workgroups.
template <class K>
static int run(K kern, int W, int iters, double units_per_wave, Result* res) {
    int cus = 0;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // LDS per workgroup such that at most W workgroups of 256 threads fit the 160 KiB of a CU
    const size_t lds = std::min<size_t>(160 * 1024 - 1024, (size_t)(160 * 1024 / W) - 1024);
    CHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = cus * W;
    Rec* d = nullptr;
    CHK(hipMalloc((void**)&d, sizeof(Rec) * blocks * 4));
    std::vector<Rec> h(blocks * 4);
    double best = 1e30, mhz = 0, wps = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), d, sizeof(Rec) * blocks * 4, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::vector<int>> by_simd;
        double csum = 0, wall = 0;
        for (int i = 0; i < blocks * 4; ++i) {
            const uint32_t key = ((h[i].xcc_id & 0xf) << 16) | (h[i].hw_id & 0xff30u);   // se, sh, cu, simd
            by_simd[key].push_back(i);
            csum += (double)(h[i].t1 - h[i].t0);
            wall += (double)h[i].wall;
        }
        std::vector<double> per, cnt;   // per SIMD: cycles per unit, wavefronts it held
        for (auto& kv : by_simd) {
            uint64_t a = ~0ull, b = 0;
            for (int i : kv.second) {
                a = std::min(a, h[i].t0);
                b = std::max(b, h[i].t1);
            }
            per.push_back((double)(b - a) / ((double)kv.second.size() * units_per_wave));
            cnt.push_back((double)kv.second.size());
        }
        if (per.empty()) continue;
        std::nth_element(per.begin(), per.begin() + per.size() / 2, per.end());
        const double v = per[per.size() / 2];
        if (rep > 0 && v < best) {
            best = v;
            mhz = csum / (wall / 100e6) / 1e6;   // shader cycles per second of the constant 100 MHz clock
            std::nth_element(cnt.begin(), cnt.begin() + cnt.size() / 2, cnt.end());
            wps = cnt[cnt.size() / 2];
        }
    }
    CHK(hipFree(d));
    res->cyc_per_unit_per_simd = best;
    res->mhz = mhz;
    res->waves_per_simd = wps;
    return 0;
}

template <int OP>
static int sweep_op(const int* Ws, int nW) {
    printf("%-16s", OP_NAMES[OP]);
    const int iters = 400;
    for (int k = 0; k < nW; ++k) {
        Result r;
        if (run(op_kernel<OP>, Ws[k], iters, (double)iters * CH * REP, &r)) return 1;
        printf("  %6.2f", r.cyc_per_unit_per_simd);
        if (k == nW - 1) printf("   (%4.0f MHz, %g waves/SIMD seen)", r.mhz, r.waves_per_simd);
    }
    printf("\n");
    fflush(stdout);
    return 0;
}

template <int OP>
struct SweepAll {
    static int go(const int* Ws, int nW) {
        if (sweep_op<OP>(Ws, nW)) return 1;
        return SweepAll<OP + 1>::go(Ws, nW);
    }
};
template <>
struct SweepAll<N_OPS> {
    static int go(const int*, int) { return 0; }
};

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, nominal %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int Ws[6] = {1, 2, 3, 4, 6, 8};
    printf("cycles (s_memtime) per wave-instruction per SIMD; W wavefronts per SIMD; %d chains per lane\n", CH);
    printf("%-16s  %6s  %6s  %6s  %6s  %6s  %6s\n", "instruction", "W=1", "W=2", "W=3", "W=4", "W=6", "W=8");
    if (SweepAll<0>::go(Ws, 6)) return 1;
    printf("\ninstruction mixes, cycles per wave per SIMD for ONE limb product (pipe cost only)\n");
    printf("%-44s  %6s  %6s  %6s\n", "mix", "W=1", "W=2", "W=4");
    const int Wm[3] = {1, 2, 4};
    const char* mix_names[3] = {"13x13 x 30-bit: 169 mad + 25 x (and, shr64, add64)", "8x8 x 52-bit: 64 x (2 fma_f64 + 2 lshl_add_u64)",
                                "8x8 x 52-bit: 64 x (2 fma_f64 + 2 x (add_co, addc))"};
    for (int mix = 0; mix < 3; ++mix) {
        printf("%-44s", mix_names[mix]);
        for (int k = 0; k < 3; ++k) {
            Result r;
            int rc = mix == 0 ? run(mix_kernel<0>, Wm[k], 200, 200.0, &r) : mix == 1 ? run(mix_kernel<1>, Wm[k], 200, 200.0, &r)
                                                                                        : run(mix_kernel<2>, Wm[k], 200, 200.0, &r);
            if (rc) return 1;
            printf("  %6.0f", r.cyc_per_unit_per_simd);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("\nfield level: cycles per wave per SIMD and unit; W wavefronts per SIMD, registers budgeted for W (spills: see -Rpass-analysis)\n");
    printf("%-44s  %7s  %7s  %7s  %7s\n", "unit", "W=1", "W=2", "W=3", "W=4");
    const char* names[4] = {"fq30_mul (fused product + reduction)", "fq30_sqr", "g1_madd (XYZZ += affine, 8M + 2S)", "fq30_inv_fermat (one SIMT inversion)"};
    const double units[4] = {64.0, 64.0, 16.0, 1.0};
    const int its[4] = {20, 20, 10, 2};
    double tab[4][4];
    for (int kind = 0; kind < 4; ++kind) {
        printf("%-44s", names[kind]);
        for (int w = 1; w <= 4; ++w) {
            Result r;
            int rc = 1;
#define FK(K, WV) if (kind == K && w == WV) rc = run(field_kernel<K, WV>, WV, its[K], units[K] * its[K], &r)
            FK(0, 1); FK(0, 2); FK(0, 3); FK(0, 4); FK(1, 1); FK(1, 2); FK(1, 3); FK(1, 4);
            FK(2, 1); FK(2, 2); FK(2, 3); FK(2, 4); FK(3, 1); FK(3, 2); FK(3, 3); FK(3, 4);
#undef FK
            if (rc) return 1;
            tab[kind][w - 1] = r.cyc_per_unit_per_simd;
            printf("  %7.0f", r.cyc_per_unit_per_simd);
            if (w == 4) printf("   (%4.0f MHz, %g waves/SIMD seen)", r.mhz, r.waves_per_simd);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("\none SIMT inversion = %.0f multiplication times (W=2); one mixed addition = %.2f multiplication times\n", tab[3][1] / tab[0][1], tab[2][1] / tab[0][1]);
    for (int w = 1; w <= 4; ++w)
        printf("chip-wide at W=%d: %.2f G mixed additions/s per GHz of shader clock (1024 SIMDs x 64 lanes / cycles)\n", w, 1024.0 * 64 / tab[2][w - 1]);
    return 0;
}
