// Argument block of one NTT pass (see ntt_kernels.hip for the algorithm).
#pragma once
#include "ff.hpp"

namespace ty {

// vectors one launch of a pass can carry (typlonk_ntt_fr_batch_devptr): the grid is `count` x the tiles of one vector,
// tables shared
constexpr uint32_t NTT_BATCH_MAX = 8;

struct NttPassArgs {
    // vector v of the batch: workgroups [v * blocks_per_vec, (v + 1) * blocks_per_vec) read in[v] and write out[v]
    const Fr* in[NTT_BATCH_MAX];
    Fr* out[NTT_BATCH_MAX];
    uint32_t blocks_per_vec;
    uint32_t k;        // log2 of this pass's sub-transform size M
    uint32_t logT;     // log2 of the tile width T
    uint32_t last;     // 1 for the final (contiguous, digit-reversing) pass
    uint32_t tw_h;     // split of the inter-pass twiddle exponent
    uint64_t S;        // stride between consecutive i_p (elements); 1 on the last pass
    uint64_t row_len;  // M * S
    // last pass addressing: row rho = k_1 * Q + q ; out = k_1 + N1 * qrev(q) + out_stride * k_P
    uint64_t N1, Q, N2, N3, out_stride;
    uint64_t n_valid;  // input elements at index >= n_valid are taken as zero and not read (zero-padded transforms); ~0: all
    const Fr* sub_tw;  // w_M^e, e < M/2
    const uint32_t* sub_tw30;  // the 9 x 30-bit kernel: 2^14 w_M^e as nine limbs on a 12-word stride (global memory)
    const Fr* tw_lo;   // w_{row_len}^e,          e < 2^tw_h
    const Fr* tw_hi;   // w_{row_len}^(e * 2^tw_h)
    const Fr* pre_lo;  // coset powers g^i applied to the input of pass 1 (forward coset NTT)
    const Fr* pre_hi;
    const Fr* post_lo; // g^-k * n^-1 applied to the output of the last pass (inverse coset NTT)
    const Fr* post_hi;
    const Fr* scale;   // n^-1 applied to the output of the last pass (plain inverse NTT)
    uint32_t pre_h, post_h;
    // Full (one multiplication) tables, built once per size in HBM and laid out exactly like the data they
    // multiply, so their loads coalesce with the data's.  NULL = compose the factor from the two-level tables.
    const Fr* tw_full;    // inter-pass twiddles of this pass: [k_p * S + column], row_len entries
    const Fr* pre_full;   // pre_lo/hi product, N entries
    const Fr* post_full;  // post_lo/hi product (scale folded in), N entries
};

}  // namespace ty
