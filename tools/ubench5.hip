// Round-5 micro-benchmark for gfx950: (a) the price of one SIMT field inversion by divsteps (fq30_inv_divsteps) against
// the Fermat ladder, in multiplication times, measured exactly as tools/ubench4.hip measures its field-level rows
// (s_memtime per physical SIMD, occupancy sweep); (b) the batched-affine bucket round -- K independent affine additions
// per lane that share ONE inversion (Montgomery's trick) -- against the XYZZ mixed addition the accumulation kernel is
// made of (msm_accum.hip), chip-wide, in additions per second.
//
// (b) in detail.  A lane owns K accumulators (affine, canonical-ish: < 1.1 p) and gets K incoming affine points per
// round.  Forward pass: d_j = x2_j - x1_j, running products kept in a scratch row; one inversion of the full product;
// backward pass: 1/d_j from the running products, lambda = (y2 - y1)/d, x3 = lambda^2 - x1 - x2,
// y3 = lambda (x1 - x3) - y1.  5M + 1S per addition + 1/K of an inversion, against 8M + 2S.  Exceptional pairs are
// routed exactly: an identity accumulator takes the point, an identity point leaves the accumulator, P + P doubles
// (lambda = 3 x^2 / 2y: the denominator of that slot is swapped to 2y before the product is formed), P + (-P) gives the
// identity.  State that does not fit on the chip goes through memory, laid out [slot][16-byte chunk][lane] (coalesced):
//   MODE_LDS     running products in the LDS (K <= 12 at one workgroup of 256 per CU), accumulators + points in HBM/L2
//   MODE_GLOBAL  running products in global scratch too (any K)
// The XYZZ leg reads the same points from the same layout and keeps its accumulator in registers, as msm_accum does.
// Every variant is checked against the XYZZ leg on the host (canonical affine results, all lanes, all slots) before it
// is timed, on an input set that contains every exceptional case.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench5.hip -o tools/ubench5
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

// ---- (a) field-level pricing, the harness of tools/ubench4.hip -----------------------------------------------------------
struct Rec {
    uint64_t t0, t1, wall;
    uint32_t hw_id, xcc_id;
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

// KIND 0: 64 dependent fq30_mul; 1: fq30_inv_fermat; 2: fq30_inv_divsteps (wave-uniform exit; build this file with
// -DFQ30_INV_FIXED_ROUNDS for the data-independent 37-round form)
template <int KIND, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void field_kernel(Rec* out, int iters, uint32_t* rounds_out) {
    extern __shared__ char lds_hold[];
    Fq30 a, b;
    uint64_t st = (uint64_t)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < 13; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        a.v[i] = (uint32_t)st & FQ30_MASK;
        b.v[i] = (uint32_t)(st >> 32) & FQ30_MASK;
    }
    a.v[12] &= 0xfffff; b.v[12] &= 0xfffff;
    int rsum = 0;
    __syncthreads();
    const uint64_t w0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { for (int k = 0; k < 32; ++k) { a = fq30_mul(a, b); b = fq30_mul(b, a); } }
        if (KIND == 1) a = fq30_inv_fermat(fq30_canon(fq30_add_lazy(a, b)));
        if (KIND == 2) { int r = 0; a = fq30_inv_divsteps(fq30_add_lazy(a, b), &r); rsum += r; }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    uint32_t s = 0;
    for (int i = 0; i < 13; ++i) s += a.v[i] ^ b.v[i];
    if (rounds_out && (threadIdx.x & 63) == 0) atomicAdd(rounds_out, (uint32_t)rsum);
    rec_store(out, t0, t1, w0, w1, s);
}

struct Result {
    double cyc_per_unit_per_simd, mhz, waves_per_simd;
};
template <class K>
static int run(K kern, int W, int iters, double units_per_wave, Result* res, uint32_t* d_rounds) {
    int cus = 0;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t lds = std::min<size_t>(160 * 1024 - 1024, (size_t)(160 * 1024 / W) - 1024);
    CHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = cus * W;
    Rec* d = nullptr;
    CHK(hipMalloc((void**)&d, sizeof(Rec) * blocks * 4));
    std::vector<Rec> h(blocks * 4);
    double best = 1e30, mhz = 0, wps = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters, rep == 3 ? d_rounds : nullptr);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), d, sizeof(Rec) * blocks * 4, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::vector<int>> by_simd;
        double csum = 0, wall = 0;
        for (int i = 0; i < blocks * 4; ++i) {
            const uint32_t key = ((h[i].xcc_id & 0xf) << 16) | (h[i].hw_id & 0xff30u);
            by_simd[key].push_back(i);
            csum += (double)(h[i].t1 - h[i].t0);
            wall += (double)h[i].wall;
        }
        std::vector<double> per, cnt;
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
            mhz = csum / (wall / 100e6) / 1e6;
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

// ---- (b) the affine round --------------------------------------------------------------------------------------------------
// field element e of slot j of lane t: chunk c (0..2) at base[((j * EPS + e) * 3 + c) * n + t], EPS elements per slot
__device__ __forceinline__ Fq30 ld_soa(const uint4* base, uint32_t row, uint32_t n, uint32_t t) {
    const uint4 a = base[((uint64_t)row * 3 + 0) * n + t], b = base[((uint64_t)row * 3 + 1) * n + t], c = base[((uint64_t)row * 3 + 2) * n + t];
    const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    return fq30_unpack(w);
}
__device__ __forceinline__ void st_soa(uint4* base, uint32_t row, uint32_t n, uint32_t t, const Fq30& v) {
    uint32_t w[12];
    fq30_pack(v, w);
    base[((uint64_t)row * 3 + 0) * n + t] = make_uint4(w[0], w[1], w[2], w[3]);
    base[((uint64_t)row * 3 + 1) * n + t] = make_uint4(w[4], w[5], w[6], w[7]);
    base[((uint64_t)row * 3 + 2) * n + t] = make_uint4(w[8], w[9], w[10], w[11]);
}
__device__ __forceinline__ bool is_id(const Fq30& x, const Fq30& y) { return fq30_is_zero_exact(x) && fq30_is_zero_exact(y); }

__device__ __forceinline__ bool is_2p(const Fq30& a) {
    uint32_t q = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) q |= a.v[i] ^ fq30_kp(2, i);
    return q == 0;
}
// The denominator and the kind of one slot: 0 = ordinary (d = x2 - x1), 1 = doubling (d = 2 y1), 2 = no arithmetic (an
// identity operand, or P + (-P)): d = 1.  Values < 1.1 p in, d < 3.2 p out.
__device__ __forceinline__ int slot_denominator(const Fq30& x1, const Fq30& y1, const Fq30& x2, const Fq30& y2, Fq30* d) {
    if (is_id(x1, y1) || is_id(x2, y2)) {
        *d = fq30_one();
        return 2;
    }
    // canonical coordinates: x2 - x1 + 2p is a multiple of p only as 2p itself
    const Fq30 dx = fq30_sub_lazy<2>(x2, x1);              // < 3.1
    if (!is_2p(dx)) {
        *d = dx;
        return 0;
    }
    if (is_2p(fq30_sub_lazy<2>(y2, y1))) {                  // the same point
        *d = fq30_mulk_lazy<2>(y1);                         // < 2.2
        return 1;
    }
    *d = fq30_one();                                        // opposite points
    return 2;
}

// raw (packed) field elements: loads are issued one slot ahead and unpacked when used, so that their latency hides
// under the ~10,000 cycles of arithmetic of the slot before (as msm_accum_kernel does with its points)
struct RawFq {
    uint4 w[3];
};
__device__ __forceinline__ RawFq ld_raw(const uint4* base, uint32_t row, uint32_t n, uint32_t t) {
    RawFq r;
#pragma unroll
    for (int c = 0; c < 3; ++c) r.w[c] = base[((uint64_t)row * 3 + c) * n + t];
    return r;
}
__device__ __forceinline__ Fq30 unraw(const RawFq& r) {
    const uint32_t w[12] = {r.w[0].x, r.w[0].y, r.w[0].z, r.w[0].w, r.w[1].x, r.w[1].y, r.w[1].z, r.w[1].w, r.w[2].x, r.w[2].y, r.w[2].z, r.w[2].w};
    return fq30_unpack(w);
}
// Stored accumulators: x canonical (the equality tests need it), y < 1.1 p.  Identity = (0, 0).  The ordinary case --
// neither x zero, x1 != x2 -- needs no y in the forward pass; everything else goes through slot_denominator.
__device__ __forceinline__ bool maybe_exceptional(const Fq30& x1, const Fq30& x2, const Fq30& dx) {
    return fq30_is_zero_exact(x1) || fq30_is_zero_exact(x2) || is_2p(dx);
}

template <int K, bool LDS_PREFIX, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void affine_round_kernel(uint4* acc, const uint4* pts, uint4* scratch,
                                                                                                            uint32_t n, int rounds) {
    extern __shared__ uint32_t lds_prefix[];   // [K][13][256]
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;   // (n is a multiple of 256 in every launch below; the inversion's exit test is wave-uniform)
    for (int r = 0; r < rounds; ++r) {
        Fq30 run = fq30_one();
        RawFq nx1 = ld_raw(acc, 0, n, t), nx2 = ld_raw(pts, 0, n, t);
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            const Fq30 x1 = unraw(nx1), x2 = unraw(nx2);
            if (j + 1 < K) {
                nx1 = ld_raw(acc, 2 * (j + 1), n, t);
                nx2 = ld_raw(pts, 2 * (j + 1), n, t);
            }
            Fq30 d = fq30_sub_lazy<2>(x2, x1);              // < 3.1
            if (maybe_exceptional(x1, x2, d)) {
                const Fq30 y1 = fq30_canon(ld_soa(acc, 2 * j + 1, n, t)), y2 = fq30_canon(ld_soa(pts, 2 * j + 1, n, t));
                (void)slot_denominator(x1, y1, x2, y2, &d);
            }
            if (LDS_PREFIX) {
#pragma unroll
                for (int l = 0; l < 13; ++l) lds_prefix[(j * 13 + l) * 256 + threadIdx.x] = run.v[l];
            } else {
                st_soa(scratch, j, n, t, run);
            }
            run = fq30_mul(run, d);                         // < 1.01
        }
        Fq30 inv = fq30_inv_divsteps(run);
        RawFq ry1, ry2, rb;
        nx1 = ld_raw(acc, 2 * (K - 1), n, t);
        ry1 = ld_raw(acc, 2 * (K - 1) + 1, n, t);
        nx2 = ld_raw(pts, 2 * (K - 1), n, t);
        ry2 = ld_raw(pts, 2 * (K - 1) + 1, n, t);
        if (!LDS_PREFIX) rb = ld_raw(scratch, K - 1, n, t);
#pragma unroll 1
        for (int j = K - 1; j >= 0; --j) {
            const Fq30 x1 = unraw(nx1), x2 = unraw(nx2);
            Fq30 y1 = unraw(ry1), y2 = unraw(ry2);
            Fq30 before;
            if (LDS_PREFIX) {
#pragma unroll
                for (int l = 0; l < 13; ++l) before.v[l] = lds_prefix[(j * 13 + l) * 256 + threadIdx.x];
            } else {
                before = unraw(rb);
            }
            if (j > 0) {
                nx1 = ld_raw(acc, 2 * (j - 1), n, t);
                ry1 = ld_raw(acc, 2 * (j - 1) + 1, n, t);
                nx2 = ld_raw(pts, 2 * (j - 1), n, t);
                ry2 = ld_raw(pts, 2 * (j - 1) + 1, n, t);
                if (!LDS_PREFIX) rb = ld_raw(scratch, j - 1, n, t);
            }
            Fq30 d = fq30_sub_lazy<2>(x2, x1);
            int kind = 0;
            if (maybe_exceptional(x1, x2, d)) {
                y1 = fq30_canon(y1);
                y2 = fq30_canon(y2);
                kind = slot_denominator(x1, y1, x2, y2, &d);
            }
            const Fq30 dinv = fq30_mul(inv, before);        // 1/d_j  < 1.01
            inv = fq30_mul(inv, d);
            if (kind == 2) {
                // identity operand: the other one; opposite points: the identity
                const bool a_id = is_id(x1, y1), p_id = is_id(x2, y2);
                const Fq30 rx = a_id ? x2 : (p_id ? x1 : fq30_zero());
                const Fq30 ry = a_id ? y2 : (p_id ? y1 : fq30_zero());
                st_soa(acc, 2 * j, n, t, rx);
                st_soa(acc, 2 * j + 1, n, t, ry);
                continue;
            }
            // numerator: y2 - y1, or 3 x1^2 when doubling
            const Fq30 num = kind == 1 ? fq30_mulk_lazy<3>(fq30_sqr(x1)) : fq30_sub_lazy<2>(y2, y1);   // < 3.2
            const Fq30 lam = fq30_mul(num, dinv);                                                      // < 1.01
            const Fq30 x3 = fq30_sub2_lazy<3>(fq30_sqr(lam), x1, x2);                                  // 1.01 + 3 < 4.1  (x1 + x2 < 2.2 <= 3)
            const Fq30 tt = fq30_sub_lazy<5>(x1, x3);                                                  // < 6.1
            const Fq30 y3 = g1_y3(lam, tt, y1, fq30_one());                                            // lam * tt - y1 * 1 with one reduction  < 1.02
            st_soa(acc, 2 * j, n, t, fq30_canon(x3));
            st_soa(acc, 2 * j + 1, n, t, y3);
        }
    }
}

// the XYZZ leg: accumulator j in registers for the whole run, `rounds` mixed additions of point j, then stored as XYZZ
template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void xyzz_kernel(const uint4* acc_in, const uint4* pts, uint4* out /* [K][4 el] */,
                                                                                                    uint32_t n, int K, int rounds) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    for (int j = 0; j < K; ++j) {
        G1Affine a;
        a.x = ld_soa(acc_in, 2 * j, n, t);
        a.y = ld_soa(acc_in, 2 * j + 1, n, t);
        G1Xyzz acc = G1Xyzz::from_affine(a);
        for (int r = 0; r < rounds; ++r) {
            G1Affine q;
            q.x = ld_soa(pts, 2 * j, n, t);
            q.y = ld_soa(pts, 2 * j + 1, n, t);
            g1_madd(acc, q, false);
        }
        st_soa(out, 4 * j, n, t, acc.x);
        st_soa(out, 4 * j + 1, n, t, acc.y);
        st_soa(out, 4 * j + 2, n, t, acc.zz);
        st_soa(out, 4 * j + 3, n, t, acc.zzz);
    }
}

// ---- host side of (b) ---------------------------------------------------------------------------------------------------
static Fq30 h_ld(const std::vector<uint4>& v, uint32_t row, uint32_t n, uint32_t t) {
    uint32_t w[12];
    for (int c = 0; c < 3; ++c) memcpy(w + 4 * c, &v[((uint64_t)row * 3 + c) * n + t], 16);
    return fq30_unpack(w);
}
static void h_st(std::vector<uint4>& v, uint32_t row, uint32_t n, uint32_t t, const Fq30& x) {
    uint32_t w[12];
    fq30_pack(x, w);
    for (int c = 0; c < 3; ++c) memcpy(&v[((uint64_t)row * 3 + c) * n + t], w + 4 * c, 16);
}
static G1Affine h_mul_g(uint64_t k) {   // k * G on the host
    G1Affine g;
    const uint32_t gx[13] = {0x14d1b01cu, 0x143790fdu, 0x34ffd633u, 0x1bc687f8u, 0x3e2228c0u, 0x04f86aa1u, 0x298df978u,
                             0x2e28c656u, 0x1b36e719u, 0x3ed397edu, 0x2f68adadu, 0x096840ceu, 0x00082ebcu};
    const uint32_t gy[13] = {0x39d1f18cu, 0x0d03d50cu, 0x10f63b65u, 0x3231b0b8u, 0x2e87afadu, 0x02eceb19u, 0x258480d0u,
                             0x31f25b61u, 0x08856e09u, 0x1fef8f3eu, 0x1a3501cbu, 0x1d6e0ad8u, 0x0016f1c9u};
    for (int i = 0; i < 13; ++i) { g.x.v[i] = gx[i]; g.y.v[i] = gy[i]; }
    G1Xyzz acc = G1Xyzz::inf();
    const G1Xyzz gp = G1Xyzz::from_affine(g);
    for (int b = 63; b >= 0; --b) {
        acc = g1_dbl(acc);
        if ((k >> b) & 1) acc = g1_add(acc, gp);
    }
    return g1_to_affine(acc);
}

struct Variant {
    const char* name;
    int K;
    bool lds;
    int waves;
};

template <int K, bool LDS, int WAVES>
static int launch_affine(uint4* acc, const uint4* pts, uint4* scratch, uint32_t n, int rounds) {
    const size_t lds = LDS ? (size_t)K * 13 * 256 * 4 : 0;
    if (lds > 64 * 1024) CHK(hipFuncSetAttribute((const void*)affine_round_kernel<K, LDS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((affine_round_kernel<K, LDS, WAVES>), dim3(n / 256), dim3(256), lds, 0, acc, pts, scratch, n, rounds);
    CHK(hipGetLastError());
    return 0;
}
static int launch_variant(int K, bool lds, int waves, uint4* acc, const uint4* pts, uint4* scratch, uint32_t n, int rounds) {
#define V(KK, LL, WW) if (K == KK && lds == LL && waves == WW) return launch_affine<KK, LL, WW>(acc, pts, scratch, n, rounds)
    V(4, true, 2); V(8, true, 1); V(8, true, 2); V(12, true, 1);
    V(8, false, 2); V(16, false, 2); V(32, false, 2);
    V(16, false, 1); V(32, false, 1); V(16, false, 3); V(32, false, 3);
#undef V
    printf("no such variant\n");
    return 1;
}
static int launch_xyzz(int waves, const uint4* acc, const uint4* pts, uint4* out, uint32_t n, int K, int rounds) {
    if (waves == 1) hipLaunchKernelGGL(xyzz_kernel<1>, dim3(n / 256), dim3(256), 0, 0, acc, pts, out, n, K, rounds);
    else if (waves == 2) hipLaunchKernelGGL(xyzz_kernel<2>, dim3(n / 256), dim3(256), 0, 0, acc, pts, out, n, K, rounds);
    else hipLaunchKernelGGL(xyzz_kernel<3>, dim3(n / 256), dim3(256), 0, 0, acc, pts, out, n, K, rounds);
    CHK(hipGetLastError());
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, nominal %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const bool only_b = argc > 1 && strcmp(argv[1], "affine") == 0;
    const bool only_a = argc > 1 && strcmp(argv[1], "inv") == 0;

    double mul_w[4] = {0, 0, 0, 0};
    if (!only_b) {
        printf("\n(a) field level: cycles per wave per SIMD and unit; W wavefronts per SIMD\n");
        printf("%-52s  %8s  %8s  %8s  %8s\n", "unit", "W=1", "W=2", "W=3", "W=4");
        const char* names[3] = {"fq30_mul (fused product + reduction)", "fq30_inv_fermat (a^(p-2))", "fq30_inv_divsteps (wave-uniform exit)"};
        const double units[3] = {64.0, 1.0, 1.0};
        const int its[3] = {20, 2, 8};
        double tab[3][4];
        uint32_t* d_rounds = nullptr;
        CHK(hipMalloc((void**)&d_rounds, 4));
        double avg_rounds = 0;
        for (int kind = 0; kind < 3; ++kind) {
            printf("%-52s", names[kind]);
            for (int w = 1; w <= 4; ++w) {
                Result r;
                int rc = 1;
                CHK(hipMemset(d_rounds, 0, 4));
#define FK(KD, WV) if (kind == KD && w == WV) rc = run(field_kernel<KD, WV>, WV, its[KD], units[KD] * its[KD], &r, d_rounds)
                FK(0, 1); FK(0, 2); FK(0, 3); FK(0, 4); FK(1, 1); FK(1, 2); FK(1, 3); FK(1, 4); FK(2, 1); FK(2, 2); FK(2, 3); FK(2, 4);
#undef FK
                if (rc) return 1;
                tab[kind][w - 1] = r.cyc_per_unit_per_simd;
                printf("  %8.0f", r.cyc_per_unit_per_simd);
                if (kind == 2 && w == 2) {
                    uint32_t hr = 0;
                    CHK(hipMemcpy(&hr, d_rounds, 4, hipMemcpyDeviceToHost));
                    avg_rounds = (double)hr / ((double)prop.multiProcessorCount * 2 * 4 * its[2]);
                }
                if (w == 4) printf("   (%4.0f MHz, %g waves/SIMD seen)", r.mhz, r.waves_per_simd);
            }
            printf("\n");
            fflush(stdout);
        }
        for (int w = 0; w < 4; ++w) mul_w[w] = tab[0][w];
        printf("\none SIMT inversion in multiplication times (same W):  W=1    W=2    W=3    W=4\n");
        printf("  Fermat ladder                                     %6.1f %6.1f %6.1f %6.1f\n", tab[1][0] / tab[0][0], tab[1][1] / tab[0][1], tab[1][2] / tab[0][2], tab[1][3] / tab[0][3]);
        printf("  divsteps                                          %6.1f %6.1f %6.1f %6.1f\n", tab[2][0] / tab[0][0], tab[2][1] / tab[0][1], tab[2][2] / tab[0][2], tab[2][3] / tab[0][3]);
        printf("  (divsteps: %.1f rounds of 30 per call on average over the wavefronts; %.0f cycles per round at W=2)\n", avg_rounds, tab[2][1] / std::max(avg_rounds, 1.0));
        CHK(hipFree(d_rounds));
    }

    if (only_a) return 0;
    // ---- (b) ----
    printf("\n(b) batched-affine bucket round against the XYZZ mixed addition (chip-wide, HIP events)\n");
    const uint32_t n = (uint32_t)prop.multiProcessorCount * 256 * 6;   // six workgroups of 256 per CU (1, 2 or 3 resident at a time)
    const int KMAX = 32;
    // points: slot j of lane t gets ((t * 131 + j * 17) % 1021 + 2) * G from a table of 1024 host-made multiples; the
    // accumulators start from other multiples, with the exceptional cases planted in lanes 0..7 of every workgroup
    std::vector<G1Affine> mult(1024);
    {
        G1Affine g = h_mul_g(1);
        G1Xyzz acc = G1Xyzz::from_affine(g);
        mult[0] = G1Affine::inf();
        mult[1] = g;
        for (int k = 2; k < 1024; ++k) {
            g1_madd(acc, g, false);
            mult[k] = g1_to_affine(acc);
        }
    }
    auto neg = [](const G1Affine& p) {
        G1Affine r = p;
        if (!p.is_inf()) r.y = fq30_canon(fq30_neg_lazy<1>(p.y));
        return r;
    };
    std::vector<uint4> h_acc((size_t)KMAX * 2 * 3 * n), h_pts((size_t)KMAX * 2 * 3 * n);
    for (uint32_t t = 0; t < n; ++t) {
        for (int j = 0; j < KMAX; ++j) {
            G1Affine p = mult[(t * 131u + j * 17u) % 1021u + 2];
            G1Affine a = mult[(t * 37u + j * 101u) % 1019u + 2];
            const uint32_t lane = t & 255u;
            if (lane == 0) a = G1Affine::inf();                     // identity accumulator
            if (lane == 1) p = G1Affine::inf();                     // identity point
            if (lane == 2) a = p;                                   // P + P in round 1 (then 2P + P ...)
            if (lane == 3) a = neg(p);                              // P + (-P) in round 1 (then inf + P ...)
            if (lane == 4) { a = G1Affine::inf(); p = G1Affine::inf(); }
            if (lane == 5) a = neg(mult[2 * ((t * 131u + j * 17u) % 400u + 2)]), p = mult[(t * 131u + j * 17u) % 400u + 2];   // -2P + P + P: hits inf in round 2
            h_st(h_acc, 2 * j, n, t, a.x);
            h_st(h_acc, 2 * j + 1, n, t, a.y);
            h_st(h_pts, 2 * j, n, t, p.x);
            h_st(h_pts, 2 * j + 1, n, t, p.y);
        }
    }
    uint4 *d_acc0, *d_acc, *d_pts, *d_scr, *d_out;
    const size_t bytes_state = (size_t)KMAX * 2 * 3 * n * 16;
    CHK(hipMalloc((void**)&d_acc0, bytes_state));
    CHK(hipMalloc((void**)&d_acc, bytes_state));
    CHK(hipMalloc((void**)&d_pts, bytes_state));
    CHK(hipMalloc((void**)&d_scr, (size_t)KMAX * 3 * n * 16));
    CHK(hipMalloc((void**)&d_out, (size_t)KMAX * 4 * 3 * n * 16));
    CHK(hipMemcpy(d_acc0, h_acc.data(), bytes_state, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_pts, h_pts.data(), bytes_state, hipMemcpyHostToDevice));

    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int ROUNDS = 6;   // >= 3: lane 5 reaches the identity in round 2 and leaves it in round 3

    // reference results: XYZZ leg, converted on the host
    printf("%u lanes (6 workgroups of 256 per CU); accumulators, points: %d slots x 96 B per lane each\n", n, KMAX);
    std::vector<uint4> h_out((size_t)KMAX * 4 * 3 * n);
    if (launch_xyzz(2, d_acc0, d_pts, d_out, n, KMAX, ROUNDS)) return 1;
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h_out.data(), d_out, h_out.size() * 16, hipMemcpyDeviceToHost));
    // host conversion of a sample of lanes (all exceptional lanes + a stride of ordinary ones)
    std::vector<uint32_t> sample;
    for (uint32_t t = 0; t < n; ++t)
        if ((t & 255u) < 8 || t % 97 == 0) sample.push_back(t);
    std::vector<G1Affine> want((size_t)sample.size() * KMAX);
    for (size_t si = 0; si < sample.size(); ++si)
        for (int j = 0; j < KMAX; ++j) {
            G1Xyzz p;
            p.x = h_ld(h_out, 4 * j, n, sample[si]);
            p.y = h_ld(h_out, 4 * j + 1, n, sample[si]);
            p.zz = h_ld(h_out, 4 * j + 2, n, sample[si]);
            p.zzz = h_ld(h_out, 4 * j + 3, n, sample[si]);
            want[si * KMAX + j] = g1_to_affine(p);
        }

    // XYZZ rate
    double xyzz_rate[4] = {0, 0, 0, 0};
    for (int w = 1; w <= 3; ++w) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (launch_xyzz(w, d_acc0, d_pts, d_out, n, 16, 16)) return 1;
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        xyzz_rate[w] = (double)n * 16 * 16 / (best * 1e-3) / 1e9;
        printf("XYZZ mixed addition, accumulator in registers, %d wavefront(s)/SIMD budget: %7.3f ms  -> %6.2f G additions/s\n", w, best, xyzz_rate[w]);
    }
    const double xyzz_best = std::max(xyzz_rate[1], std::max(xyzz_rate[2], xyzz_rate[3]));

    const Variant vars[] = {{"K=4  prefix in LDS, W=2", 4, true, 2},      {"K=8  prefix in LDS, W=1", 8, true, 1},    {"K=8  prefix in LDS, W=2", 8, true, 2},
                            {"K=12 prefix in LDS, W=1", 12, true, 1},      {"K=8  prefix in HBM, W=2", 8, false, 2},   {"K=16 prefix in HBM, W=2", 16, false, 2},
                            {"K=32 prefix in HBM, W=2", 32, false, 2},     {"K=16 prefix in HBM, W=1", 16, false, 1},
                            {"K=32 prefix in HBM, W=1", 32, false, 1},     {"K=16 prefix in HBM, W=3", 16, false, 3},  {"K=32 prefix in HBM, W=3", 32, false, 3}};
    printf("%-28s  %9s  %14s  %9s  %s\n", "affine round variant", "ms/launch", "G additions/s", "vs XYZZ", "check");
    for (const Variant& v : vars) {
        // correctness first
        CHK(hipMemcpy(d_acc, d_acc0, bytes_state, hipMemcpyDeviceToDevice));
        if (launch_variant(v.K, v.lds, v.waves, d_acc, d_pts, d_scr, n, ROUNDS)) return 1;
        CHK(hipDeviceSynchronize());
        std::vector<uint4> h_res((size_t)KMAX * 2 * 3 * n);
        CHK(hipMemcpy(h_res.data(), d_acc, bytes_state, hipMemcpyDeviceToHost));
        size_t bad = 0, checked = 0;
        for (size_t si = 0; si < sample.size(); ++si)
            for (int j = 0; j < v.K; ++j) {
                const Fq30 x = fq30_canon(h_ld(h_res, 2 * j, n, sample[si])), y = fq30_canon(h_ld(h_res, 2 * j + 1, n, sample[si]));
                const G1Affine& w = want[si * KMAX + j];
                bool ok = true;
                for (int i = 0; i < 13; ++i) ok = ok && x.v[i] == w.x.v[i] && y.v[i] == w.y.v[i];
                bad += ok ? 0 : 1;
                ++checked;
            }
        // timing: the accumulators keep evolving (ordinary additions from round 3 on)
        float best = 1e30f;
        const int TR = 4;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (launch_variant(v.K, v.lds, v.waves, d_acc, d_pts, d_scr, n, TR)) return 1;
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        const double rate = (double)n * v.K * TR / (best * 1e-3) / 1e9;
        printf("%-28s  %9.3f  %14.2f  %8.2fx  %s (%zu of %zu results differ)\n", v.name, best, rate, rate / xyzz_best, bad ? "FAIL" : "ok", bad, checked);
        fflush(stdout);
    }
    (void)mul_w;
    return 0;
}
