// Radix-2 NTT over Fr as a Bailey-style multi-pass transform (replaces ark-poly 0.3.0
// Radix2EvaluationDomain::{fft,ifft}; reference call sites /root/reference/plonk/src/proof.rs:50,
// 106, 115, 125, 128, 337, 415 and plonk/src/builder.rs:85).
//
// N = N_1 * N_2 * ... * N_P (P <= 4, N_p = 2^k_p <= 2^10).  With the input index written
// i = (i_1, i_2, ..., i_P) most-significant first and the output index k = (k_P, ..., k_2, k_1):
//   pass p < P : for every (k_1..k_{p-1}) row and every column r = (i_{p+1}..i_P), a size-N_p NTT
//                over i_p (stride S_p = N / (N_1..N_p)), times the inter-pass twiddle
//                w_{row_len}^(r * k_p), written back to the same positions (in place per tile);
//   pass P     : contiguous size-N_P NTTs; the store performs the digit reversal so the result
//                lands in natural order: out[k_1 + N_1 k_2 + ... + (N/N_P) k_P].
// Every global access moves >= 128-B contiguous chunks (a workgroup owns a tile of T adjacent
// columns / T rows whose outputs are adjacent), the butterflies run on an LDS-resident tile of
// 1024 elements (32 KiB) and the sub-transform's twiddles w_{N_p}^e are staged in LDS.
// Inter-pass twiddles and coset powers come from two-level tables (lo[e & mask] * hi[e >> h]),
// each <= 2^ceil(log/2) entries, so they stay in L2.
#include "fr30.hpp"
#include "launch.hpp"

namespace ty {

__device__ __forceinline__ Fr ntt_ld(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void ntt_st(Fr* p, const Fr& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ Fr ntt_pow2l(const Fr* lo, const Fr* hi, uint32_t h, uint64_t e) {
    return fe_mul(ntt_ld(lo + (e & ((1ull << h) - 1))), ntt_ld(hi + (e >> h)));
}

// ---- register-resident radix-4 stage groups -------------------------------------------------------------------------
// A radix-2 stage makes one LDS round trip (two loads, two stores, a __syncthreads): ten per 2^10-point
// sub-transform.  Here a thread takes the four elements of two consecutive DIF stages -- i0 + j * quarter, j = 0..3,
// quarter = half / 2 -- does both stages in registers and stores them back in place: five round trips for k = 10, the
// same multiplications in the same order on every element (so the results, and the lazy bounds of the 30-bit form,
// are those of the radix-2 stages).  An odd k starts with one radix-2 stage, so that the LAST group is always the
// one with half = 2, 1, whose twiddles are 1 except w^(M/4): one multiplication per four elements.
//   stage s     (x0, x2) pos = low, (x1, x3) pos = low + quarter;   twiddle w^(pos << s)
//   stage s + 1 (x0', x1') and (x2', x3'), both pos = low;          twiddle w^(low << (s + 1))
struct NttArith32 {
    using E = Fr;
    using P = Fr*;
    using TW = Fr*;   // sub-transform twiddles: staged in LDS behind the tile
    static __device__ __forceinline__ E ldtw(TW tw, uint32_t idx) { return ntt_ld(tw + idx); }
    static __device__ __forceinline__ E ld(P tile, uint32_t idx) { return ntt_ld(tile + idx); }
    static __device__ __forceinline__ void st(P tile, uint32_t idx, const E& x) { ntt_st(tile + idx, x); }
    static __device__ __forceinline__ E add(const E& a, const E& b) { return fe_add(a, b); }
    static __device__ __forceinline__ E sub(const E& a, const E& b) { return fe_sub(a, b); }
    static __device__ __forceinline__ E mul(const E& a, const E& b) { return fe_mul(a, b); }
};
struct NttArith30 {
    using E = Fr30;
    using P = uint32_t*;
    // sub-transform twiddles: read from GLOBAL memory, nine limbs on a 12-word (48-byte) stride (NttPassArgs::sub_tw30).
    // At 36 B per element the tile alone fills the LDS budget -- 4096 elements = 144 KiB, four 1024-element tiles = 144 KiB
    // per CU -- and the table (<= 24 KiB) lives in L1 / L2.
    using TW = const uint32_t*;
    static __device__ __forceinline__ E ldtw(TW tw, uint32_t idx) {
        const uint4* q = reinterpret_cast<const uint4*>(tw + 12 * (size_t)idx);
        const uint4 a = q[0], b = q[1];
        E r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        r.v[8] = tw[12 * (size_t)idx + 8];
        return r;
    }
    static __device__ __forceinline__ E ld(P tile, uint32_t idx) {
        E r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = tile[9 * idx + i];
        return r;
    }
    static __device__ __forceinline__ void st(P tile, uint32_t idx, const E& x) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tile[9 * idx + i] = x.v[i];
    }
    static __device__ __forceinline__ E add(const E& a, const E& b) { return fr30_add(a, b); }
    static __device__ __forceinline__ E sub(const E& a, const E& b) { return fr30_sub(a, b); }
    static __device__ __forceinline__ E mul(const E& a, const E& b) { return fr30_mul(a, b); }
};

// direct = true: the FIRST group takes its elements straight from global memory (gload(i, t): element (i, t) of the
// tile, pre-factor applied) and the LAST group hands its results straight to gstore(i, t, y) (y = the value at in-tile
// position i of the bit-reversed output; gstore applies the pass's closing factor and writes to global): the staging
// copy into LDS, the copy out of it and two barriers per pass disappear.  i_fast: in the first group consecutive lanes
// take consecutive i instead of consecutive t (the last pass reads rows of M contiguous elements).
template <class A, class LoadG, class StoreG>
__device__ __forceinline__ void ntt_stages_radix4(typename A::P tile, typename A::TW stw, uint32_t k, uint32_t logT, uint32_t E,
                                                  uint32_t tid, uint32_t nt, bool direct, bool i_fast, LoadG gload, StoreG gstore) {
    using El = typename A::E;
    const uint32_t T = 1u << logT;
    uint32_t s = 0;
    bool first = direct;
    if (k & 1u) {  // the odd stage first: half = 2^(k-1)
        const uint32_t lh = k - 1, half = 1u << lh;
        for (uint32_t qq = tid; qq < (E >> 1); qq += nt) {
            uint32_t t, pos;                                     // one butterfly per (pos, t): j < half, so pos = j
            if (first && i_fast) { pos = qq & (half - 1); t = qq >> lh; }
            else { t = qq & (T - 1); pos = qq >> logT; }
            const uint32_t ia = (pos << logT) + t, ib = ((pos + half) << logT) + t;
            const El x = first ? gload(pos, t) : A::ld(tile, ia), y = first ? gload(pos + half, t) : A::ld(tile, ib);
            A::st(tile, ia, A::add(x, y));
            const El d = A::sub(x, y);
            A::st(tile, ib, lh == 0 ? d : A::mul(d, A::ldtw(stw, pos)));
        }
        __syncthreads();
        s = 1;
        first = false;
    }
    for (; s + 1 < k; s += 2) {
        const uint32_t lh = k - 1 - s;          // log2(half) of stage s, >= 1
        const uint32_t quarter = 1u << (lh - 1);
        const bool lastg = lh == 1;             // stages with half = 2, 1: low = 0
        const bool out_direct = direct && lastg;
        for (uint32_t g = tid; g < (E >> 2); g += nt) {
            uint32_t t, rest;
            if (first && i_fast) { rest = g & ((1u << (k - 2)) - 1); t = g >> (k - 2); }
            else { t = g & (T - 1); rest = g >> logT; }
            const uint32_t low = rest & (quarter - 1), hi = rest >> (lh - 1);
            const uint32_t i0 = (hi << (lh + 1)) + low;
            const uint32_t a0 = (i0 << logT) + t, a1 = a0 + (quarter << logT), a2 = a1 + (quarter << logT),
                           a3 = a2 + (quarter << logT);
            const El x0 = first ? gload(i0, t) : A::ld(tile, a0), x1 = first ? gload(i0 + quarter, t) : A::ld(tile, a1),
                     x2 = first ? gload(i0 + 2 * quarter, t) : A::ld(tile, a2), x3 = first ? gload(i0 + 3 * quarter, t) : A::ld(tile, a3);
            const El s0 = A::add(x0, x2), s1 = A::add(x1, x3);
            El d0 = A::sub(x0, x2), d1 = A::sub(x1, x3);
            if (!lastg) {
                d0 = A::mul(d0, A::ldtw(stw, low << s));
                d1 = A::mul(d1, A::ldtw(stw, (low + quarter) << s));
            } else {
                d1 = A::mul(d1, A::ldtw(stw, 1u << s));   // w^(M/4); d0's twiddle is 1
            }
            El y1 = A::sub(s0, s1), y3 = A::sub(d0, d1);
            if (!lastg) {
                const El tw = A::ldtw(stw, low << (s + 1));
                y1 = A::mul(y1, tw);
                y3 = A::mul(y3, tw);
            }
            const El y0 = A::add(s0, s1), y2 = A::add(d0, d1);
            if (out_direct) {
                gstore(i0, t, y0);
                gstore(i0 + quarter, t, y1);
                gstore(i0 + 2 * quarter, t, y2);
                gstore(i0 + 3 * quarter, t, y3);
            } else {
                A::st(tile, a0, y0);
                A::st(tile, a1, y1);
                A::st(tile, a2, y2);
                A::st(tile, a3, y3);
            }
        }
        if (!out_direct) __syncthreads();
        first = false;
    }
}

// 256 threads on tiles of <= 1024 elements (32 KiB of LDS), or 1024 threads on tiles of <= 4096 elements (128 KiB:
// one workgroup per CU) when a 2^17..2^20 transform is done in two passes instead of three (ntt_host.hip, ntt_run)
__global__ __launch_bounds__(1024) void ntt_pass_kernel(NttPassArgs a) {
    const uint32_t NTT_THREADS = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];
    const uint32_t k = a.k, logT = a.logT;
    const uint32_t M = 1u << k, T = 1u << logT, E = M << logT;
    Fr* tile = reinterpret_cast<Fr*>(ntt_smem);
    Fr* stw = tile + E;
    const uint32_t tid = threadIdx.x;
    const uint32_t vec = blockIdx.x / a.blocks_per_vec;   // which vector of the batch (vector-major: tile b of every vector
    const uint64_t b = blockIdx.x % a.blocks_per_vec;     // lands on the same XCD, whose L2 then holds that tile's twiddles)
    const Fr* vin = a.in[vec];
    Fr* vout = a.out[vec];   // (middle passes run in place: vin == vout)

    for (uint32_t i = tid; i < (M >> 1); i += NTT_THREADS) ntt_st(stw + i, ntt_ld(a.sub_tw + i));

    uint64_t base = 0, c0 = 0, q = 0, k1base = 0, obase = 0;
    if (!a.last) {
        const uint64_t tiles_per_row = a.S >> logT;
        const uint64_t row = b / tiles_per_row;
        c0 = (b % tiles_per_row) << logT;
        base = row * a.row_len + c0;
    } else {
        q = b % a.Q;
        k1base = (b / a.Q) << logT;
        obase = k1base + a.N1 * ((q / a.N3) + a.N2 * (q % a.N3));
    }
    // element (i, t) of this workgroup's tile, pre-factor applied / its place in the output with the closing factor
    auto gload = [&](uint32_t i, uint32_t t) -> Fr {
        const uint64_t g = a.last ? ((k1base + t) * a.Q + q) * M + i : base + (uint64_t)i * a.S + t;
        if (g >= a.n_valid) return Fr::zero();
        Fr x = ntt_ld(vin + g);
        if (a.pre_full) x = fe_mul(x, ntt_ld(a.pre_full + g));
        else if (a.pre_lo) x = fe_mul(x, ntt_pow2l(a.pre_lo, a.pre_hi, a.pre_h, g));
        return x;
    };
    auto gstore = [&](uint32_t i, uint32_t t, Fr x) {   // the value at in-tile position i is output kk = bitrev_k(i)
        const uint32_t kk = (k ? (__brev(i) >> (32 - k)) : 0u);
        if (!a.last) {
            if (a.tw_full) x = fe_mul(x, ntt_ld(a.tw_full + ((uint64_t)kk * a.S + c0 + t)));
            else x = fe_mul(x, ntt_pow2l(a.tw_lo, a.tw_hi, a.tw_h, (c0 + t) * (uint64_t)kk));
            ntt_st(vout + (base + (uint64_t)kk * a.S + t), x);
        } else {
            const uint64_t o = obase + t + a.out_stride * kk;
            if (a.post_full) x = fe_mul(x, ntt_ld(a.post_full + o));
            else if (a.post_lo) x = fe_mul(x, ntt_pow2l(a.post_lo, a.post_hi, a.post_h, o));
            if (a.scale) x = fe_mul(x, ntt_ld(a.scale));
            ntt_st(vout + o, x);
        }
    };
    const bool direct = k >= 2;   // (k = 1: a single radix-2 stage through the LDS tile)
    if (!direct) {
        if (!a.last) {
            for (uint32_t idx = tid; idx < E; idx += NTT_THREADS) ntt_st(tile + idx, gload(idx >> logT, idx & (T - 1)));
        } else {
            for (uint32_t idx = tid; idx < E; idx += NTT_THREADS) {
                const uint32_t i = idx & (M - 1), t = idx >> k;
                ntt_st(tile + ((i << logT) + t), gload(i, t));
            }
        }
    }
    __syncthreads();   // (direct: the sub-transform's twiddles are in LDS)

    // k DIF stages as register-resident radix-4 groups, natural order in, bit-reversed order out (within the tile).
    // The twiddle of a butterfly is w^(pos << s) with pos < half: it is 1 for pos = 0, i.e. for EVERY butterfly of the
    // last stage (half = 1) and for every other one of the stage before (half = 2); those multiplications are skipped --
    // 0.75 of the k/2 multiplications per element of a pass (19 % of a 2^20 transform's).
    ntt_stages_radix4<NttArith32>(tile, stw, k, logT, E, tid, NTT_THREADS, direct, a.last != 0, gload, gstore);
    if (direct) return;
    for (uint32_t idx = tid; idx < E; idx += NTT_THREADS) {
        const uint32_t t = idx & (T - 1), kk = idx >> logT;
        const uint32_t src = (k ? (__brev(kk) >> (32 - k)) : 0u);
        gstore(src, t, ntt_ld(tile + ((src << logT) + t)));
    }
}

// ---- the same pass on 9 x 30-bit limbs (fr30.hpp) -------------------------------------------------------------------
// Same tiling, addressing and stage order as ntt_pass_kernel.  The LDS tile holds 9 words per element and nothing else
// (the sub-transform twiddles are read from global memory: sub_tw30); every table (sub_tw30, tw_full, pre_full, post_full,
// scale) is in the 2^270 domain and must be a full table (ntt_host.hip selects this kernel only then).  Additions are lazy (fr30.hpp states the bounds); a pass ends with a multiplication of every
// element -- the inter-pass twiddle, the coset / scaling factor of the last pass, or 2^270 mod r when the last pass has
// no factor -- which brings it below 2r; the last pass then subtracts r once more where needed, so that what reaches
// the caller is the canonical residue, bit for bit what ntt_pass_kernel writes.
__device__ __forceinline__ Fr30 lds_ld30(const uint32_t* p) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = p[i];
    return r;
}
__device__ __forceinline__ void lds_st30(uint32_t* p, const Fr30& r) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = r.v[i];
}
__global__ __launch_bounds__(1024) void ntt_pass30_kernel(NttPassArgs a) {
    const uint32_t NTT_THREADS = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];
    const uint32_t k = a.k, logT = a.logT;
    const uint32_t M = 1u << k, T = 1u << logT, E = M << logT;
    uint32_t* tile = reinterpret_cast<uint32_t*>(ntt_smem);
    const uint32_t* stw = a.sub_tw30;   // global memory (NttArith30::ldtw)
    const uint32_t tid = threadIdx.x;
    const uint32_t vec = blockIdx.x / a.blocks_per_vec;   // (as in ntt_pass_kernel)
    const uint64_t b = blockIdx.x % a.blocks_per_vec;
    const Fr* vin = a.in[vec];
    Fr* vout = a.out[vec];   // (middle passes run in place: vin == vout)

    uint64_t base = 0, c0 = 0, q = 0, k1base = 0, obase = 0;
    if (!a.last) {
        const uint64_t tiles_per_row = a.S >> logT;
        const uint64_t row = b / tiles_per_row;
        c0 = (b % tiles_per_row) << logT;
        base = row * a.row_len + c0;
    } else {
        q = b % a.Q;
        k1base = (b / a.Q) << logT;
        obase = k1base + a.N1 * ((q / a.N3) + a.N2 * (q % a.N3));
    }
    auto gload = [&](uint32_t i, uint32_t t) -> Fr30 {
        const uint64_t g = a.last ? ((k1base + t) * a.Q + q) * M + i : base + (uint64_t)i * a.S + t;
        if (g >= a.n_valid) return fr30_unpack(Fr::zero());
        Fr30 x = fr30_unpack(ntt_ld(vin + g));
        if (a.pre_full) x = fr30_mul(x, fr30_unpack(ntt_ld(a.pre_full + g)));
        return x;
    };
    auto gstore = [&](uint32_t i, uint32_t t, Fr30 x) {   // the value at in-tile position i is output kk = bitrev_k(i)
        const uint32_t kk = (k ? (__brev(i) >> (32 - k)) : 0u);
        if (!a.last) {
            x = fr30_mul(x, fr30_unpack(ntt_ld(a.tw_full + ((uint64_t)kk * a.S + c0 + t))));
            ntt_st(vout + (base + (uint64_t)kk * a.S + t), fr30_pack(x));  // < 2r: the next pass takes it as it is
        } else {
            const uint64_t o = obase + t + a.out_stride * kk;
            if (a.post_full) x = fr30_mul(x, fr30_unpack(ntt_ld(a.post_full + o)));
            else if (a.scale) x = fr30_mul(x, fr30_unpack(ntt_ld(a.scale)));
            else x = fr30_reduce_lazy(x);   // no factor to fold the reduction into: one quotient estimate instead of a multiplication
            ntt_st(vout + o, fr30_to_canonical(x));
        }
    };
    const bool direct = k >= 2;   // (k = 1: a single radix-2 stage through the LDS tile)
    if (!direct) {
        if (!a.last) {
            for (uint32_t idx = tid; idx < E; idx += NTT_THREADS) lds_st30(tile + 9 * idx, gload(idx >> logT, idx & (T - 1)));
        } else {
            for (uint32_t idx = tid; idx < E; idx += NTT_THREADS) {
                const uint32_t i = idx & (M - 1), t = idx >> k;
                lds_st30(tile + 9 * ((i << logT) + t), gload(i, t));
            }
        }
    }
    __syncthreads();

    ntt_stages_radix4<NttArith30>(tile, stw, k, logT, E, tid, NTT_THREADS, direct, a.last != 0, gload, gstore);
    if (direct) return;
    for (uint32_t idx = tid; idx < E; idx += NTT_THREADS) {
        const uint32_t t = idx & (T - 1), kk = idx >> logT;
        const uint32_t src = (k ? (__brev(kk) >> (32 - k)) : 0u);
        gstore(src, t, lds_ld30(tile + 9 * ((src << logT) + t)));
    }
}

// out[idx] = lo/hi power at exponent (idx % S) * (idx / S) (S != 0: inter-pass twiddles) or idx (S == 0)
__global__ __launch_bounds__(256) void ntt_full_table_kernel(const Fr* lo, const Fr* hi, uint32_t h, uint64_t S, uint64_t n,
                                                             Fr* out) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const uint64_t e = S ? (idx % S) * (idx / S) : idx;
    ntt_st(out + idx, ntt_pow2l(lo, hi, h, e));
}
void launch_ntt_full_table(const Fr* lo, const Fr* hi, uint32_t h, uint64_t S, uint64_t n, Fr* out, hipStream_t s) {
    hipLaunchKernelGGL(ntt_full_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lo, hi, h, S, n, out);
}

// More than the default 64 KiB of dynamic LDS needs an opt-in (gfx950 has 160 KiB per workgroup).  The attribute is
// per DEVICE: a process may hold contexts on several (typlonk_init(device_ordinal)), so the flag is kept per device
// ordinal, and a refusal is reported to the planner, which then stays with 1024-element tiles.
static bool ntt_raise_lds(const void* kernel, int which) {
    static int state[2][64] = {};   // 0 = not tried, 1 = raised, -1 = refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[which][dev] == 0)
        state[which][dev] = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : -1;
    if (state[which][dev] < 0) (void)hipGetLastError();
    return state[which][dev] > 0;
}
// (both kernels: which of the two runs a 2^20 transform is decided later, by the tables that could be built -- the plan
// must be launchable either way, so a refusal for one of them sends the planner back to 1024-element tiles for both)
bool ntt_big_tiles_available() {
    const bool a = ntt_raise_lds(reinterpret_cast<const void*>(ntt_pass_kernel), 0);
    const bool b = ntt_raise_lds(reinterpret_cast<const void*>(ntt_pass30_kernel), 1);
    return a && b;
}

void launch_ntt_pass(const NttPassArgs& a, unsigned blocks, unsigned threads, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 64 * 1024) (void)ntt_raise_lds(reinterpret_cast<const void*>(ntt_pass_kernel), 0);
    hipLaunchKernelGGL(ntt_pass_kernel, dim3(blocks), dim3(threads), lds_bytes, s, a);
}
void launch_ntt_pass30(const NttPassArgs& a, unsigned blocks, unsigned threads, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 64 * 1024) (void)ntt_raise_lds(reinterpret_cast<const void*>(ntt_pass30_kernel), 1);
    hipLaunchKernelGGL(ntt_pass30_kernel, dim3(blocks), dim3(threads), lds_bytes, s, a);
}

}  // namespace ty
