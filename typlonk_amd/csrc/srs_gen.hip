// Device-side SRS generation: g1[i] = [s^(start+i)] G, canonical affine -- the vector
// Srs::from_secret builds (/root/reference/kzg/src/srs.rs:15-24, 30-34), one thread per power.
// Setup-time only (not on the prove() path); it exists so that benchmarks and large parity tests
// can create 2^20..2^22-point SRS shards directly in HBM.
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

struct SrsGenArgs {
    Fr s;
    Fq gx, gy;
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
    uint32_t* p = a.pts + i * 24;
    st_fq(p, r.x);
    st_fq(p + 12, r.y);
}

void launch_srs_generate(const Fr& s, uint64_t start, uint64_t n, uint32_t* pts, hipStream_t st) {
    SrsGenArgs a;
    a.s = s;
    // G1 generator, Montgomery form (ark-bls12-381 G1_GENERATOR_X / _Y)
    const uint32_t gx[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                             0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    const uint32_t gy[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                             0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    for (int i = 0; i < 12; ++i) {
        a.gx.v[i] = gx[i];
        a.gy.v[i] = gy[i];
    }
    a.start = start;
    a.n = n;
    a.pts = pts;
    hipLaunchKernelGGL(srs_generate_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
}

}  // namespace ty
