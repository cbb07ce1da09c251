// Device-side SRS generation: g1[i] = [s^(start+i)] G, canonical affine -- the vector
// Srs::from_secret builds (/root/reference/kzg/src/srs.rs:15-24, 30-34), one thread per power.
// Setup-time only (not on the prove() path); it exists so that benchmarks and large parity tests
// can create 2^20..2^22-point SRS shards directly in HBM.
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

struct SrsGenArgs {
    Fr s;
    Fq30 gx, gy;
    uint64_t start, n;
    uint32_t* pts;
};

__global__ __launch_bounds__(64) void srs_generate_kernel(SrsGenArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    // e = s^(start+i)
    const uint64_t ex = a.start + i;
    Fr e = Fr::one();
    for (int b = 63; b >= 0; --b) {
        e = fe_sqr(e);
        if ((ex >> b) & 1) e = fe_mul(e, a.s);
    }
    const Fr k = fe_from_mont(e);
    G1Xyzz acc = G1Xyzz::inf();
    for (int w = 7; w >= 0; --w) {
        for (int b = 31; b >= 0; --b) {
            acc = g1_dbl(acc);
            if ((k.v[w] >> b) & 1) g1_madd_xy(acc, a.gx, a.gy);
        }
    }
    const G1Affine r = g1_to_affine(acc);
    uint32_t* p = a.pts + i * PT_WORDS;
    st_fq(p, r.x);
    st_fq(p + 12, r.y);
}

// Fixed-base window tables: table t holds 2^(c*t) * P_i (affine, canonical) at point index t*len + i,
// table 0 being the SRS itself.  One thread per base walks the tables: c doublings, one inversion.
// Setup-time only; it trades HBM capacity (T x the SRS) for the whole cross-window recombination of
// every later MSM over this SRS.
__global__ __launch_bounds__(64) void srs_tables_kernel(uint32_t* pts, uint64_t len, uint32_t c, uint32_t T) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= len) return;
    G1Affine p = ld_affine(pts, i);
    for (uint32_t t = 1; t < T; ++t) {
        uint32_t* dst = pts + (t * len + i) * PT_WORDS;
        if (!p.is_inf()) {
            G1Xyzz acc = G1Xyzz::from_affine(p);
            for (uint32_t d = 0; d < c; ++d) acc = g1_dbl(acc);
            p = g1_to_affine(acc);
        }
        st_fq(dst, p.x);
        st_fq(dst + 12, p.y);
    }
}
void launch_srs_tables(uint32_t* pts, uint64_t len, uint32_t c, uint32_t T, hipStream_t s) {
    hipLaunchKernelGGL(srs_tables_kernel, dim3((unsigned)((len + 63) / 64)), dim3(64), 0, s, pts, len, c, T);
}

void launch_srs_generate(const Fr& s, uint64_t start, uint64_t n, uint32_t* pts, hipStream_t st) {
    SrsGenArgs a;
    a.s = s;
    // G1 generator in the internal form (x * 2^390 mod p as 30-bit digits)
    const uint32_t gx[13] = {0x14d1b01cu, 0x143790fdu, 0x34ffd633u, 0x1bc687f8u, 0x3e2228c0u, 0x04f86aa1u, 0x298df978u,
                             0x2e28c656u, 0x1b36e719u, 0x3ed397edu, 0x2f68adadu, 0x096840ceu, 0x00082ebcu};
    const uint32_t gy[13] = {0x39d1f18cu, 0x0d03d50cu, 0x10f63b65u, 0x3231b0b8u, 0x2e87afadu, 0x02eceb19u, 0x258480d0u,
                             0x31f25b61u, 0x08856e09u, 0x1fef8f3eu, 0x1a3501cbu, 0x1d6e0ad8u, 0x0016f1c9u};
    for (int i = 0; i < 13; ++i) {
        a.gx.v[i] = gx[i];
        a.gy.v[i] = gy[i];
    }
    a.start = start;
    a.n = n;
    a.pts = pts;
    hipLaunchKernelGGL(srs_generate_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
}

}  // namespace ty
