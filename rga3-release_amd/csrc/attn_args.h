// Argument block shared by the attention forward kernels (attn_fwd.hip, attn_causal32.hip).
#pragma once
#include "common.h"

namespace rga3 {

struct AttnArgs {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    unsigned short* o;
    float* lse;
    const int* cu_q;
    const int* cu_k;
    long q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh;
    int Hq, Hkv, D;
    long total_q;
    float scale_log2;  // softmax scale * log2(e)
    int causal;
    // split-KV (few query blocks x heads, long key range: SAM2 memory attention is 64 workgroups of one head over 28 736 keys):
    float* split_o;    // f32 [nsplit][total_q][Hq][D] normalised partial outputs
    float* split_lse;  // f32 [nsplit][Hq][total_q] partial log2-sum-exp (of the scaled scores)
    int nsplit;
    // block-diagonal visibility inside a segment (several tiny windows packed into one segment: Hiera's 4- and 16-token windows would
    // otherwise be one workgroup each): query i sees key j iff (i >> bq_shift) == (j >> bk_shift); -1 = off
    int bq_shift, bk_shift;
    // RoPE applied while loading (rotate-half pairing d <-> d +- D/2, tables [tokens, D] f32 indexed by the packed token): q always when rope_cos is set,
    // k too when rope_kcos is set (windowed ViT attention loads every key exactly once per head, so the stand-alone rope pass disappears altogether)
    const float* rope_cos;
    const float* rope_sin;
    const float* rope_kcos;
    const float* rope_ksin;
    int gx;            // workgroups per (segment, head): the grid is 1-D, gx * Hq * nseg, decoded XCD-aware in the kernel
};

// attn_causal32.hip: causal rows at D = 128 on 32-row waves (v_mfma_f32_32x32x16_bf16), balanced over the key range.  Same arguments / results as the general kernel.
int launch_causal32(const AttnArgs& a, int nseg, int max_q, hipStream_t st);

}  // namespace rga3
