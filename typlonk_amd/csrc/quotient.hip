// Quotient polynomial t(X) of the PLONK prover on the 4n coset domain -- replaces the 12 schoolbook
// `naive_mul` products + divide_by_vanishing_poly of plonk::proof::quotient_polynomial
// (/root/reference/plonk/src/proof.rs:292-375) by pointwise arithmetic on coset evaluations:
//
//   t = [ q_l a + q_r b - q_o c + q_m a b + q_c + PI
//         + alpha ( (a + beta k0 X + gamma)(b + beta k1 X + gamma)(c + beta k2 X + gamma) Z
//                 - (a + beta s0 + gamma)(b + beta s1 + gamma)(c + beta s2 + gamma) Z(wX) )
//         + alpha^2 (Z - 1) L0 ] / (X^n - 1)
//
// evaluated at x_i = g w_{4n}^i, i < 4n.  deg(numerator) <= 4n - 4 < 4n, so the inverse coset NTT of the
// pointwise quotient is exactly t whenever the numerator vanishes on H (any valid witness; the
// reference discards the remainder otherwise).  On this domain Z(w x_i) is the evaluation at index
// i + 4 (w = w_{4n}^4) and X^n - 1 takes only four values g^n i^k - 1, whose inverses come as arguments.
#include "launch.hpp"

namespace ty {

__device__ __forceinline__ Fr q_ld(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void q_st(Fr* p, const Fr& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

__global__ __launch_bounds__(256) void fr_fill_kernel(Fr* out, uint64_t n, Fr value) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) q_st(out + i, value);
}

__global__ __launch_bounds__(256) void quotient_pointwise_kernel(QuotientArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n4) return;
    const Fr wa = q_ld(a.wires[0] + i), wb = q_ld(a.wires[1] + i), wc = q_ld(a.wires[2] + i);
    const Fr z = q_ld(a.z + i), zw = q_ld(a.z + ((i + 4) & (a.n4 - 1)));
    // gate constraint
    Fr line1 = fe_mul(q_ld(a.sel[0] + i), wa);
    line1 = fe_add(line1, fe_mul(q_ld(a.sel[1] + i), wb));
    line1 = fe_sub(line1, fe_mul(q_ld(a.sel[2] + i), wc));
    line1 = fe_add(line1, fe_mul(fe_mul(q_ld(a.sel[3] + i), wa), wb));
    line1 = fe_add(line1, q_ld(a.sel[4] + i));
    if (a.pi) line1 = fe_add(line1, q_ld(a.pi + i));  // no public-input polynomial = the zero polynomial
    // beta * x_i, x_i = g * w_{4n}^i: the two-level power table of w_{4n} with beta * g folded into its upper level
    const Fr bx = fe_mul(q_ld(a.w_lo + (i & ((1ull << a.w_h) - 1))), q_ld(a.bx_hi + (i >> a.w_h)));
    Fr l2 = fe_add(fe_add(wa, a.k0_is_one ? bx : fe_mul(a.k[0], bx)), a.gamma);
    l2 = fe_mul(l2, fe_add(fe_add(wb, fe_mul(a.k[1], bx)), a.gamma));
    l2 = fe_mul(l2, fe_add(fe_add(wc, fe_mul(a.k[2], bx)), a.gamma));
    l2 = fe_mul(l2, z);
    Fr l3 = fe_add(fe_add(wa, fe_mul(a.beta, q_ld(a.sigma[0] + i))), a.gamma);
    l3 = fe_mul(l3, fe_add(fe_add(wb, fe_mul(a.beta, q_ld(a.sigma[1] + i))), a.gamma));
    l3 = fe_mul(l3, fe_add(fe_add(wc, fe_mul(a.beta, q_ld(a.sigma[2] + i))), a.gamma));
    l3 = fe_mul(l3, zw);
    const Fr l4 = fe_mul(fe_sub(z, Fr::one()), q_ld(a.l0 + i));
    Fr t = fe_add(line1, fe_mul(a.alpha, fe_sub(l2, l3)));
    t = fe_add(t, fe_mul(a.alpha2, l4));
    t = fe_mul(t, a.zh_inv[i & 3]);
    q_st(a.out + i, t);
}

__global__ __launch_bounds__(256) void fr_scale_kernel(const Fr* in, uint64_t n, Fr factor, Fr* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) q_st(out + i, fe_mul(q_ld(in + i), factor));
}

void launch_fr_scale(const Fr* in, uint64_t n, const Fr& factor, Fr* out, hipStream_t s) {
    hipLaunchKernelGGL(fr_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, factor, out);
}
void launch_fr_fill(Fr* out, uint64_t n, const Fr& value, hipStream_t s) {
    hipLaunchKernelGGL(fr_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, value);
}
void launch_quotient_pointwise(const QuotientArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(quotient_pointwise_kernel, dim3((unsigned)((a.n4 + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace ty
