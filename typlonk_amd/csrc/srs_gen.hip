// Device-side SRS generation: g1[i] = [s^(start+i)] G, canonical affine -- the vector
// Srs::from_secret builds (/root/reference/kzg/src/srs.rs:15-24, 30-34), one thread per power.
// Setup-time only (not on the prove() path); it exists so that benchmarks and large parity tests
// can create 2^20..2^22-point SRS shards directly in HBM.
#include "launch.hpp"
#include "msm_common.hpp"

namespace ty {

// ---- fixed-base comb for G ----------------------------------------------------------------------------------------------
// comb[w * 256 + j] = [j * 2^(8 w)] G, affine, w < 32, 1 <= j < 256 (entry 0 of a row is unused): a power of the secret
// then costs 32 mixed additions and one inversion instead of the 255 doublings + ~128 additions + 551-multiplication
// Fermat inversion of rounds 1-4 (69 ms per 2^20 points -> profiles/r05_*).  Built once per context (1 MiB).
constexpr uint32_t COMB_WINDOWS = 32, COMB_ROW = 256;

__device__ __forceinline__ void g1_generator(Fq30& gx, Fq30& gy) {
    // G1 generator in the internal form (x * 2^390 mod p as 30-bit digits)
    constexpr uint32_t x[13] = {0x14d1b01cu, 0x143790fdu, 0x34ffd633u, 0x1bc687f8u, 0x3e2228c0u, 0x04f86aa1u, 0x298df978u,
                                0x2e28c656u, 0x1b36e719u, 0x3ed397edu, 0x2f68adadu, 0x096840ceu, 0x00082ebcu};
    constexpr uint32_t y[13] = {0x39d1f18cu, 0x0d03d50cu, 0x10f63b65u, 0x3231b0b8u, 0x2e87afadu, 0x02eceb19u, 0x258480d0u,
                                0x31f25b61u, 0x08856e09u, 0x1fef8f3eu, 0x1a3501cbu, 0x1d6e0ad8u, 0x0016f1c9u};
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        gx.v[i] = x[i];
        gy.v[i] = y[i];
    }
}

__global__ __launch_bounds__(64) void srs_comb_kernel(uint32_t* comb) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;   // (w, j)
    const uint32_t w = t / COMB_ROW, j = t % COMB_ROW;
    if (w >= COMB_WINDOWS) return;
    G1Affine g;
    g1_generator(g.x, g.y);
    // B = 2^(8w) G, then j B by an 8-step double-and-add
    G1Jac b = G1Jac::from_affine(g);
    for (uint32_t d = 0; d < 8 * w; ++d) b = g1_jac_dbl(b);
    const G1Xyzz bx = g1_jac_to_xyzz(b);
    G1Xyzz acc = G1Xyzz::inf();
    for (int bit = 7; bit >= 0; --bit) {
        acc = g1_dbl(acc);
        if ((j >> bit) & 1) acc = g1_add(acc, bx);
    }
    const G1Affine r = g1_to_affine(acc);   // j = 0: the identity, (0, 0)
    uint32_t* p = comb + (uint64_t)t * PT_WORDS;
    st_fq(p, r.x);
    st_fq(p + 12, r.y);
}

struct SrsGenArgs {
    Fr s;
    uint64_t start, n;
    const uint32_t* comb;
    uint32_t* pts;
};

__global__ __launch_bounds__(64) void srs_generate_kernel(SrsGenArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    // (lanes past the end run along on the last power: the inversion's exit test is wave-uniform)
    const uint64_t ex = a.start + (i < a.n ? i : a.n - 1);
    // e = s^(start+i)
    Fr e = Fr::one();
    for (int b = 63; b >= 0; --b) {
        e = fe_sqr(e);
        if ((ex >> b) & 1) e = fe_mul(e, a.s);
    }
    const Fr k = fe_from_mont(e);
    G1Xyzz acc = G1Xyzz::inf();
    for (uint32_t w = 0; w < COMB_WINDOWS; ++w) {
        const uint32_t j = (k.v[w >> 2] >> (8 * (w & 3))) & 255u;
        if (j) {
            const G1Affine q = ld_affine(a.comb, (uint64_t)w * COMB_ROW + j);
            g1_madd_xy(acc, q.x, q.y);
        }
    }
    const G1Affine r = g1_to_affine(acc);
    if (i >= a.n) return;
    uint32_t* p = a.pts + i * PT_WORDS;
    st_fq(p, r.x);
    st_fq(p + 12, r.y);
}

// ---- fixed-base window tables -------------------------------------------------------------------------------------------
// Table t holds 2^(c*t) * P_i (affine, canonical) at point index t*len + i, table 0 being the SRS itself.  Setup-time
// only; it trades HBM capacity (T x the SRS) for the whole cross-window recombination of every later MSM over this SRS.
//
// One thread per base walks its column of T - 1 entries as ONE Jacobian doubling chain -- c doublings per entry, never
// normalised on the way -- parks (X_t, Y_t) in the entry's own slot and Z_t in a scratch vector, and keeps the running
// products Z_1 ... Z_(t-1) in the LDS; then one inversion (divsteps, fq30.hpp) of the full product and a walk back up
// the column (Montgomery's trick) turn every entry into its canonical affine form.  Per entry: c * 6.0 + 7
// multiplication times, and 1/(T-1) of an inversion, where rounds 1-4 paid c * 8.0 + 5 + a 551-multiplication Fermat
// ladder: 115 ms -> profiles/r05_* per 2^20-point SRS at c = 20.  Same points, bit for bit (canonical affine).
// LDS: (T - 1) x 13 words per thread, [t][limb][thread].
__global__ __launch_bounds__(64) void srs_tables_kernel(uint32_t* pts, uint32_t* zbuf, uint64_t len, uint32_t c, uint32_t T) {
    extern __shared__ uint32_t prefix[];
    const uint64_t i0 = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = i0 < len;
    const uint64_t i = live ? i0 : len - 1;            // spare lanes shadow the last base (wave-uniform inversion loop)
    const G1Affine p = ld_affine(pts, i);
    const bool inf = p.is_inf();
    G1Jac a;
    a.x = p.x;
    a.y = p.y;
    a.z = fq30_one();                                   // the identity walks along as a harmless non-point
    Fq30 run = fq30_one();
    uint32_t dead_from = inf ? 1u : T;                  // first table whose entry is the identity (T: none)
    for (uint32_t t = 1; t < T; ++t) {
        for (uint32_t d = 0; d < c; ++d) a = g1_jac_dbl(a);
        // a denominator that vanishes (an input that is not a point of odd order -- only typlonk_srs_load can bring one,
        // it does not validate) is taken out of the product; 2^k * identity = identity, so that entry AND every later one
        // of the column are the identity (the chain itself walks on from a harmless non-point)
        const bool zero = fq30_is_zero_mod(a.z);
        if (zero) {
            a.z = fq30_one();
            dead_from = min(dead_from, t);
        }
        if (live) {
            uint32_t* dst = pts + (t * len + i) * PT_WORDS;
            st_fq(dst, zero ? fq30_zero() : a.x);
            st_fq(dst + 12, zero ? fq30_zero() : a.y);
            st_fq(zbuf + ((uint64_t)(t - 1) * len + i) * 12, a.z);
        }
#pragma unroll
        for (int l = 0; l < 13; ++l) prefix[((t - 1) * 13 + l) * 64 + threadIdx.x] = run.v[l];
        run = fq30_mul(run, a.z);                       // < 1.01
    }
    Fq30 inv = fq30_inv(run);                           // 1 / (Z_1 ... Z_(T-1))
    if (!live) return;
    for (uint32_t t = T - 1; t >= 1; --t) {
        uint32_t* dst = pts + (t * len + i) * PT_WORDS;
        Fq30 before;
#pragma unroll
        for (int l = 0; l < 13; ++l) before.v[l] = prefix[((t - 1) * 13 + l) * 64 + threadIdx.x];
        const Fq30 z = ld_fq(zbuf + ((uint64_t)(t - 1) * len + i) * 12);
        const Fq30 zinv = fq30_mul(inv, before);        // 1 / Z_t
        inv = fq30_mul(inv, z);
        G1Jac e;
        e.x = ld_fq(dst);
        e.y = ld_fq(dst + 12);
        G1Affine r = g1_jac_to_affine_with(e, zinv);
        if (t >= dead_from) r = G1Affine::inf();
        st_fq(dst, r.x);
        st_fq(dst + 12, r.y);
    }
}
// scratch of the walk: one denominator (48 B) per entry of tables 1 .. T-1.  The LDS holds (T - 1) x 13 words per thread:
// 59.9 KiB for the largest table count the ABI accepts (c = 14: 19 tables), inside the 64 KiB a launch gets by default.
size_t srs_tables_scratch_bytes(uint64_t len, uint32_t T) { return T > 1 ? (size_t)(T - 1) * len * 48 : 0; }
static_assert((19 - 1) * 13 * 64 * 4 <= 64 * 1024, "the running products of the largest table count must fit the default LDS");
void launch_srs_tables(uint32_t* pts, uint32_t* zbuf, uint64_t len, uint32_t c, uint32_t T, hipStream_t s) {
    if (T <= 1 || len == 0) return;
    hipLaunchKernelGGL(srs_tables_kernel, dim3((unsigned)((len + 63) / 64)), dim3(64), (size_t)(T - 1) * 13 * 64 * 4, s, pts, zbuf, len, c, T);
}

// ---- self-test of the SIMT inversion (typlonk_selftest_fq_inv) --------------------------------------------------------------
// Every thread draws `per_thread` residues (xorshift; every 16th slot an edge value: 0, 1, 2, p - 1, p - 2, a one-limb
// value), lifts them by 0..7 multiples of p (the contract of fq30_inv: any normalised value < 8p), and compares
// fq30_inv_divsteps with the Fermat ladder a^(p-2) digit for digit after canonicalisation, plus x * x^-1 = 1.
// out[0] = mismatches, out[1] = largest number of 30-divstep rounds a call ran, out[2] = calls.
__global__ __launch_bounds__(64) void fq_inv_selftest_kernel(uint64_t seed, uint32_t per_thread, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    uint64_t st = (seed + tid) * 0x9E3779B97F4A7C15ull + 1;
    uint32_t bad = 0;
    int maxr = 0;
    for (uint32_t it = 0; it < per_thread; ++it) {
        Fq30 x;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            st ^= st << 13;
            st ^= st >> 7;
            st ^= st << 17;
            x.v[i] = (uint32_t)st & FQ30_MASK;
        }
        x.v[12] &= 0x000fffffu;   // < 2^380 < p
        const uint32_t cls = (it * 64u + threadIdx.x) & 255u;
        if (cls < 6) {
            const Fq30 keep = x;
            x = fq30_zero();
            if (cls == 1) x.v[0] = 1;
            if (cls == 2) x.v[0] = 2;
            if (cls == 3 || cls == 4) {
#pragma unroll
                for (int i = 0; i < 13; ++i) x.v[i] = fq30_kp(1, i);
                x.v[0] -= cls == 3 ? 1u : 2u;
            }
            if (cls == 5) x.v[0] = keep.v[0];
        }
        Fq30 xl = x;
        const uint32_t lift = (uint32_t)(st >> 40) & 7u;
        for (uint32_t l = 0; l < lift; ++l) {
            Fq30 pp;
#pragma unroll
            for (int i = 0; i < 13; ++i) pp.v[i] = fq30_kp(1, i);
            xl = fq30_add_lazy(xl, pp);
        }
        int rounds = 0;
        const Fq30 d = fq30_canon(fq30_inv_divsteps(xl, &rounds));
        const Fq30 f = fq30_canon(fq30_inv_fermat(x));
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 13; ++i) ok = ok && d.v[i] == f.v[i];
        if (!fq30_is_zero_exact(x)) {
            const Fq30 one = fq30_canon(fq30_mul(xl, d)), r1 = fq30_one();
#pragma unroll
            for (int i = 0; i < 13; ++i) ok = ok && one.v[i] == r1.v[i];
        } else {
            ok = ok && fq30_is_zero_exact(d);
        }
        bad += ok ? 0u : 1u;
        maxr = rounds > maxr ? rounds : maxr;
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicMax(&out[1], (uint32_t)maxr);
    atomicAdd(&out[2], per_thread);
}
void launch_fq_inv_selftest(uint64_t seed, uint32_t threads, uint32_t per_thread, uint32_t* out, hipStream_t st) {
    hipLaunchKernelGGL(fq_inv_selftest_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, seed, per_thread, out);
}

void launch_srs_comb(uint32_t* comb, hipStream_t st) {
    hipLaunchKernelGGL(srs_comb_kernel, dim3(COMB_WINDOWS * COMB_ROW / 64), dim3(64), 0, st, comb);
}
size_t srs_comb_bytes() { return (size_t)COMB_WINDOWS * COMB_ROW * PT_WORDS * 4; }

void launch_srs_generate(const Fr& s, uint64_t start, uint64_t n, const uint32_t* comb, uint32_t* pts, hipStream_t st) {
    if (n == 0) return;
    SrsGenArgs a;
    a.s = s;
    a.start = start;
    a.n = n;
    a.comb = comb;
    a.pts = pts;
    hipLaunchKernelGGL(srs_generate_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
}

}  // namespace ty
