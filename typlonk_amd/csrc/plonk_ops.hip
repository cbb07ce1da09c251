// O(n) prover steps that sit between the NTT / MSM kernels of prove(), kept on the device so that
// polynomials never cross PCIe inside a proof:
//
//   grand product   permutation::CompiledPermutation::prove   /root/reference/permutation/src/proving.rs:7-31
//   open            kzg::KzgScheme::open (Horner + division)   /root/reference/kzg/src/lib.rs:55-61
//   lincomb         the axpy combinations of linearisation_poly /root/reference/plonk/src/proof.rs:376-439
//
// All three are scans over Fr.  The grand product avoids the reference's one field division per cell:
//   Z_j = prod_{k<j} num_k / prod_{k<j} den_k = N_j * S_j * S_0^-1,
// N = exclusive prefix products of the numerators, S_j = prod_{k>=j} den_k (suffix products), so one
// inversion (of S_0, on the device: fr_inv_kernel) serves the whole column.  Field arithmetic is exact, so every Z_j is
// the same field element the reference computes.
#include "fr_inv.hpp"
#include "launch.hpp"

namespace ty {

__device__ __forceinline__ Fr p_ld(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void p_st(Fr* p, const Fr& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// num_j = prod_i (w_ij + beta k_i w^j + gamma),  den_j = prod_i (w_ij + beta sigma_ij + gamma)
__global__ __launch_bounds__(256) void gp_terms_kernel(GrandProductArgs a) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n) return;
    const Fr x = fe_mul(p_ld(a.w_lo + (j & ((1ull << a.w_h) - 1))), p_ld(a.w_hi + (j >> a.w_h)));
    Fr num = Fr::one(), den = Fr::one();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const Fr w = p_ld(a.wires[i] + j);
        const Fr wg = fe_add(w, a.gamma);
        num = fe_mul(num, fe_add(wg, fe_mul(a.kbeta[i], x)));
        den = fe_mul(den, fe_add(wg, fe_mul(a.beta, p_ld(a.sigma[i] + j))));
    }
    p_st(a.num + j, num);
    p_st(a.den + j, den);
}

// ---- product scan over Fr: three launches, 2048 elements per workgroup (8 per thread) ----------------
// reverse = 0: out[j] = prod_{k<j} in[k] (exclusive prefix);  reverse = 1: out[j] = prod_{k>=j} in[k]
constexpr int PSCAN_PER_BLOCK = 2048;

__device__ __forceinline__ uint64_t pscan_index(uint64_t pos, uint64_t n, int reverse) { return reverse ? n - 1 - pos : pos; }

__global__ __launch_bounds__(256) void pscan_block_kernel(const Fr* in, uint64_t n, int reverse, Fr* block_prod) {
    __shared__ Fr red[256];
    const uint64_t base = (uint64_t)blockIdx.x * PSCAN_PER_BLOCK + threadIdx.x * 8;
    Fr p = Fr::one();
    for (int e = 0; e < 8; ++e)
        if (base + e < n) p = fe_mul(p, p_ld(in + pscan_index(base + e, n, reverse)));
    red[threadIdx.x] = p;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fe_mul(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) p_st(block_prod + blockIdx.x, red[0]);
}
// single workgroup: exclusive scan of the block products in place (sequential over chunks of 256)
__global__ __launch_bounds__(256) void pscan_top_kernel(Fr* block_prod, uint32_t nblocks) {
    __shared__ Fr buf[256];
    __shared__ Fr running;
    if (threadIdx.x == 0) running = Fr::one();
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const Fr v = i < nblocks ? p_ld(block_prod + i) : Fr::one();
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            Fr t = Fr::one();
            if ((int)threadIdx.x >= off) t = buf[threadIdx.x - off];
            __syncthreads();
            buf[threadIdx.x] = fe_mul(buf[threadIdx.x], t);
            __syncthreads();
        }
        // exclusive value = running * (inclusive of the previous lane)
        Fr excl = running;
        if (threadIdx.x > 0) excl = fe_mul(running, buf[threadIdx.x - 1]);
        const Fr total = fe_mul(running, buf[255]);
        if (i < nblocks) p_st(block_prod + i, excl);
        __syncthreads();
        if (threadIdx.x == 0) running = total;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void pscan_finish_kernel(const Fr* in, uint64_t n, int reverse, const Fr* block_excl,
                                                           Fr* out) {
    __shared__ Fr buf[256];
    const uint64_t base = (uint64_t)blockIdx.x * PSCAN_PER_BLOCK + threadIdx.x * 8;
    Fr v[8];
    Fr p = Fr::one();
    for (int e = 0; e < 8; ++e) {
        v[e] = base + e < n ? p_ld(in + pscan_index(base + e, n, reverse)) : Fr::one();
        p = fe_mul(p, v[e]);
    }
    buf[threadIdx.x] = p;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        Fr t = Fr::one();
        if ((int)threadIdx.x >= off) t = buf[threadIdx.x - off];
        __syncthreads();
        buf[threadIdx.x] = fe_mul(buf[threadIdx.x], t);
        __syncthreads();
    }
    Fr run = p_ld(block_excl + blockIdx.x);
    if (threadIdx.x > 0) run = fe_mul(run, buf[threadIdx.x - 1]);
    for (int e = 0; e < 8; ++e) {
        if (base + e < n) {
            if (reverse) {
                run = fe_mul(run, v[e]);  // inclusive in scan order = product of in[k], k >= index
                p_st(out + pscan_index(base + e, n, 1), run);
            } else {
                p_st(out + base + e, run);  // exclusive prefix
                run = fe_mul(run, v[e]);
            }
        }
    }
}

// out[0] = in[0]^-1 (0 -> 0): one wavefront, every lane the same value -- the ONE inversion of a proof's grand product
// (fr_inv.hpp: divsteps, ~20 rounds of 30), on the device so that round 2 never drains the stream for it
__global__ __launch_bounds__(64) void fr_inv_kernel(const Fr* in, Fr* out) {
    const Fr x = fr_inv_divsteps(p_ld(in));
    if (threadIdx.x == 0) p_st(out, x);
}
// Z_j = N_j * S_j * inv_total
__global__ __launch_bounds__(256) void gp_finish_kernel(const Fr* nprefix, const Fr* dsuffix, const Fr* inv_total, uint64_t n, Fr* z) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    p_st(z + j, fe_mul(fe_mul(p_ld(nprefix + j), p_ld(dsuffix + j)), p_ld(inv_total)));
}

// ---- open(): H_j = c_j + z H_{j+1} (H_m = 0) for all j at once ------------------------------------------
// y = p(z) = H_0 and (p - y) / (X - z) has coefficients q_{j-1} = H_j: Horner evaluation and synthetic
// division are the same suffix recurrence.  A thread owns 8 consecutive coefficients, a workgroup
// 2048; inside the workgroup the per-thread values are combined by a Hillis-Steele suffix scan with
// ratio z^8 (multipliers z^(8*2^k) come precomputed as zpow[3+k]).  `seed` is the value of H at the END
// of the workgroup's range (0 in the first sweep, the scanned carry in the second).
__device__ __forceinline__ Fr horner_block(const Fr* c, uint64_t m, uint64_t base, const Fr& seed, const Fr* zpow,
                                           int pow0, Fr* lds, Fr (&loc)[8], Fr* carry_in) {
    // local Horner over [base + 8t, base + 8t + 8); the last thread starts from the seed
    const uint64_t s0 = base + (uint64_t)threadIdx.x * 8;
    Fr h = (threadIdx.x == 255) ? seed : Fr::zero();
    const Fr z = zpow[pow0];
    for (int e = 7; e >= 0; --e) {
        loc[e] = (s0 + e < m) ? p_ld(c + s0 + e) : Fr::zero();
        h = fe_add(loc[e], fe_mul(z, h));
    }
    lds[threadIdx.x] = h;
    __syncthreads();
    // G_t = sum_{t' >= t} h_t' (z^8)^(t' - t)
    for (int k = 0; k < 8; ++k) {
        const int off = 1 << k;
        Fr add = Fr::zero();
        if ((int)threadIdx.x + off < 256) add = fe_mul(zpow[pow0 + 3 + k], lds[threadIdx.x + off]);
        __syncthreads();
        lds[threadIdx.x] = fe_add(lds[threadIdx.x], add);
        __syncthreads();
    }
    // value of H just after this thread's range
    *carry_in = (threadIdx.x == 255) ? seed : lds[threadIdx.x + 1];
    return lds[0];
}

struct OpenArgs {
    const Fr* c;
    uint64_t m;
    Fr* q;          // m - 1 coefficients, may be null (evaluation only)
    Fr* blocks;     // per-workgroup values / carries
    Fr* y;          // device scalar: p(z)
    Fr zpow[32];    // z^(2^k)
};

// sweep 1: H at the start of every workgroup assuming a zero carry
__global__ __launch_bounds__(256) void open_block_kernel(OpenArgs a) {
    __shared__ Fr lds[256];
    Fr loc[8], ci;
    const Fr g0 = horner_block(a.c, a.m, (uint64_t)blockIdx.x * 2048, Fr::zero(), a.zpow, 0, lds, loc, &ci);
    if (threadIdx.x == 0) p_st(a.blocks + blockIdx.x, g0);
}
// single workgroup: carries between workgroups, C_b = A_b + z^2048 C_{b+1}; blocks[b] <- C_{b+1}
// (the value of H at the end of workgroup b); up to 2048 workgroups
__global__ __launch_bounds__(256) void open_top_kernel(OpenArgs a, uint32_t nblk) {
    __shared__ Fr lds[256];
    Fr loc[8], ci;
    horner_block(a.blocks, nblk, 0, Fr::zero(), a.zpow, 11, lds, loc, &ci);
    // recompute the local chain from the true carry-in and store, for every entry, H of the NEXT entry
    const Fr zb = a.zpow[11];
    Fr h = ci;
    const uint64_t s0 = (uint64_t)threadIdx.x * 8;
    for (int e = 7; e >= 0; --e) {
        if (s0 + e < nblk) p_st(a.blocks + s0 + e, h);
        h = fe_add(loc[e], fe_mul(zb, h));
    }
}
// sweep 2: seeded with the true carry; writes q and y
__global__ __launch_bounds__(256) void open_finish_kernel(OpenArgs a) {
    __shared__ Fr lds[256];
    Fr loc[8], ci;
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    const Fr seed = p_ld(a.blocks + blockIdx.x);
    horner_block(a.c, a.m, base, seed, a.zpow, 0, lds, loc, &ci);
    const Fr z = a.zpow[0];
    Fr h = ci;
    const uint64_t s0 = base + (uint64_t)threadIdx.x * 8;
    for (int e = 7; e >= 0; --e) {
        const uint64_t i = s0 + e;
        h = fe_add(loc[e], fe_mul(z, h));  // H_i
        if (i < a.m) {
            if (i == 0) p_st(a.y, h);
            else if (a.q) p_st(a.q + i - 1, h);
        }
    }
}

// ---- up to 8 openings / evaluations of polynomials of the same length at one of two points: three launches in all ----
// (round 3 of the prover: a, b, c, Z, sigma_0, sigma_1, PI at zeta and Z at zeta * w were seventeen launches of
// 25-55 us each on the context's stream before anything else of the round could start)
struct OpenMultiArgs {
    const Fr* c[8];
    Fr* q[8];        // m - 1 quotient coefficients, or null: evaluation only
    Fr* y[8];        // device scalars
    uint8_t zsel[8]; // which of the two points
    uint64_t m;
    Fr* blocks;      // 8 * nblk per-workgroup values / carries
    uint32_t nblk;
    Fr zpow[2][32];
};
__global__ __launch_bounds__(256) void open_multi_block_kernel(OpenMultiArgs a) {
    __shared__ Fr lds[256];
    Fr loc[8], ci;
    const uint32_t k = blockIdx.y;
    const Fr g0 = horner_block(a.c[k], a.m, (uint64_t)blockIdx.x * 2048, Fr::zero(), a.zpow[a.zsel[k]], 0, lds, loc, &ci);
    if (threadIdx.x == 0) p_st(a.blocks + (uint64_t)k * a.nblk + blockIdx.x, g0);
}
__global__ __launch_bounds__(256) void open_multi_top_kernel(OpenMultiArgs a) {
    __shared__ Fr lds[256];
    Fr loc[8], ci;
    const uint32_t k = blockIdx.x;
    const Fr* zp = a.zpow[a.zsel[k]];
    Fr* blocks = a.blocks + (uint64_t)k * a.nblk;
    const Fr y = horner_block(blocks, a.nblk, 0, Fr::zero(), zp, 11, lds, loc, &ci);
    if (!a.q[k]) {
        if (threadIdx.x == 0) p_st(a.y[k], y);  // p(z) = sum_b A_b (z^2048)^b: an evaluation is complete here
        return;
    }
    const Fr zb = zp[11];
    Fr h = ci;
    const uint64_t s0 = (uint64_t)threadIdx.x * 8;
    for (int e = 7; e >= 0; --e) {
        if (s0 + e < a.nblk) p_st(blocks + s0 + e, h);
        h = fe_add(loc[e], fe_mul(zb, h));
    }
}
__global__ __launch_bounds__(256) void open_multi_finish_kernel(OpenMultiArgs a) {
    __shared__ Fr lds[256];
    Fr loc[8], ci;
    const uint32_t k = blockIdx.y;
    if (!a.q[k]) return;  // whole workgroup: no barrier is skipped by part of it
    const Fr* zp = a.zpow[a.zsel[k]];
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    const Fr seed = p_ld(a.blocks + (uint64_t)k * a.nblk + blockIdx.x);
    horner_block(a.c[k], a.m, base, seed, zp, 0, lds, loc, &ci);
    const Fr z = zp[0];
    Fr h = ci;
    const uint64_t s0 = base + (uint64_t)threadIdx.x * 8;
    for (int e = 7; e >= 0; --e) {
        const uint64_t i = s0 + e;
        h = fe_add(loc[e], fe_mul(z, h));
        if (i < a.m) {
            if (i == 0) p_st(a.y[k], h);
            else p_st(a.q[k] + i - 1, h);
        }
    }
}
void launch_open_multi(const Fr* const* polys, Fr* const* quotients, Fr* const* ys, const uint8_t* zsel, uint32_t count,
                       uint64_t m, const Fr& z0, const Fr& z1, Fr* blocks, hipStream_t s) {
    OpenMultiArgs a;
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t j = k < count ? k : 0;
        a.c[k] = polys[j];
        a.q[k] = quotients[j];
        a.y[k] = ys[j];
        a.zsel[k] = zsel[j];
    }
    a.m = m;
    a.blocks = blocks;
    a.nblk = (uint32_t)((m + 2047) / 2048);
    a.zpow[0][0] = z0;
    a.zpow[1][0] = z1;
    for (int k = 1; k < 32; ++k) {
        a.zpow[0][k] = fe_sqr(a.zpow[0][k - 1]);
        a.zpow[1][k] = fe_sqr(a.zpow[1][k - 1]);
    }
    hipLaunchKernelGGL(open_multi_block_kernel, dim3(a.nblk, count), dim3(256), 0, s, a);
    hipLaunchKernelGGL(open_multi_top_kernel, dim3(count), dim3(256), 0, s, a);
    hipLaunchKernelGGL(open_multi_finish_kernel, dim3(a.nblk, count), dim3(256), 0, s, a);
}

// ---- out[i] = sum_k scalar_k * poly_k[i]  (+ constant on coefficient 0) ---------------------------------
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    Fr acc = (i == 0) ? a.constant : Fr::zero();
    for (uint32_t k = 0; k < a.terms; ++k) acc = fe_add(acc, fe_mul(a.scalar[k], p_ld(a.poly[k] + i)));
    p_st(a.out + i, acc);
}

void launch_lincomb(const LincombArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
}

void launch_open(const Fr* c, uint64_t m, const Fr& z, Fr* q, Fr* blocks, Fr* y, hipStream_t s) {
    OpenArgs a;
    a.c = c;
    a.m = m;
    a.q = q;
    a.blocks = blocks;
    a.y = y;
    a.zpow[0] = z;
    for (int k = 1; k < 32; ++k) a.zpow[k] = fe_sqr(a.zpow[k - 1]);
    const uint32_t nblk = (uint32_t)((m + 2047) / 2048);
    hipLaunchKernelGGL(open_block_kernel, dim3(nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(open_top_kernel, dim3(1), dim3(256), 0, s, a, nblk);
    hipLaunchKernelGGL(open_finish_kernel, dim3(nblk), dim3(256), 0, s, a);
}

void launch_gp_terms(const GrandProductArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(gp_terms_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
}
void launch_product_scan(const Fr* in, uint64_t n, int reverse, Fr* block_scratch, Fr* out, hipStream_t s) {
    const uint32_t nblk = (uint32_t)((n + PSCAN_PER_BLOCK - 1) / PSCAN_PER_BLOCK);
    hipLaunchKernelGGL(pscan_block_kernel, dim3(nblk), dim3(256), 0, s, in, n, reverse, block_scratch);
    hipLaunchKernelGGL(pscan_top_kernel, dim3(1), dim3(256), 0, s, block_scratch, nblk);
    hipLaunchKernelGGL(pscan_finish_kernel, dim3(nblk), dim3(256), 0, s, in, n, reverse, block_scratch, out);
}
void launch_fr_inv(const Fr* in, Fr* out, hipStream_t s) { hipLaunchKernelGGL(fr_inv_kernel, dim3(1), dim3(64), 0, s, in, out); }
void launch_gp_finish(const Fr* nprefix, const Fr* dsuffix, const Fr* inv_total, uint64_t n, Fr* z, hipStream_t s) {
    hipLaunchKernelGGL(gp_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nprefix, dsuffix, inv_total, n, z);
}

}  // namespace ty
