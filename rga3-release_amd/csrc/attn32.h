// Interface of attn32.hip (32-query-row non-causal attention for head dims 33 .. 96) towards the dispatcher in attn_fwd.hip.
#pragma once
#include "common.h"

namespace rga3 {

struct Attn32Args {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    unsigned short* o;
    float* lse;          // optional [Hq, total_q], natural-log domain
    const int* cu_q;
    const int* cu_k;
    long q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh;
    long total_q;
    int Hq, Hkv, D;
    float scale_log2;
    int gx;              // query blocks of 256 rows per (segment, head); set by attn32_launch
};

bool attn32_applies(int D, int causal, int max_q, int max_k);
int attn32_launch(Attn32Args a, int nseg, int max_q, hipStream_t st);

}  // namespace rga3
