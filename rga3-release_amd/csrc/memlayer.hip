// Row-resident glue of a SAM2 memory-attention layer at inference (reference model/sam2.py:448-530 MemoryAttentionLayer._forward_sa / _forward_ca / forward:
// norm -> projection -> RoPE on one side of each attention, out-projection + residual on the other; RoPEAttention :1484-1548, apply_rotary_enc :1901-1923).
//
// Between the two attention kernels and the FFN products a layer is a chain of small row-wise steps on [4096, 256] rows -- merge of the split-KV partials,
// output projection (K = 256 or 64) + residual, LayerNorm, the next projection (N = 256 or 768), axial RoPE -- each a launch of 5 - 14 us at the launch floor
// (profiles/r03_stream_frame_timeline_fused_tail.txt: 9 of a layer's 14 launches).  A workgroup here owns 16 COMPLETE rows (256 workgroups for a 64 x 64 frame) and
// runs the whole chain on them; the rows never leave the CU between the steps:
//
//   [operand of product 1]  rows a [M, K1] bf16, or the cross-attention partials (csrc/memattn.hip: unnormalised sums + (max, sum) per key slice, merged in
//                           slice order exactly as memattn_combine_kernel does and rounded to bf16)
//   product 1 (optional)    x' = bf16(bf16(a W1^T + b1) + res)        -- written; the rounding points of the GEMM epilogue with a residual (gemm_bf16.hip)
//                           without it x' = res (the rows are only normalised)
//   LayerNorm               t = LN(x') gamma + beta, two passes over the register-resident row (mean, then centred squares) -- optionally written
//   product 2 (optional)    y = bf16(t W2^T + b2), N2 = 256 or 768; columns < rope_cols rotated in place of a separate pass: consecutive (even, odd) pairs,
//                           table row = token % nq, pair index = (column % 256) / 2; rounded to bf16 before and after the rotation as the two launches did
//
// Tokens sit on the MFMA's N side (B operand = the 16 rows, A operand = 16 weight rows), so lane (c, g) ends up with 4 consecutive output columns of token c:
// packed 8-byte stores, RoPE pairs inside a lane, LayerNorm sums = in-lane + two shuffles + one 8-wave exchange through LDS.  Each of the 8 waves owns an eighth
// of the output columns; weight fragments come straight from L2 as 16-byte loads (W1 + W2 <= 512 KB per workgroup; the matrices are shared by all 256 workgroups
// and stay L2-resident), the 16 input rows are staged once in LDS (row stride + 16 B: conflict-free ds_read_b128).  Latency-bound by design: the point is 1 launch
// instead of 3 - 5, not MFMA utilisation.
#include "common.h"

#include <stdlib.h>

namespace rga3 {

constexpr int ML_D = 256;                // model width
constexpr int ML_STR = ML_D * 2 + 16;    // LDS row stride (bytes)

struct MemLayerArgs {
    const unsigned short* a; long a_st;                 // operand rows of product 1 [M, K1] (null with partials)
    const float* part_o; const float* part_ml; int nsplit;   // attention partials (csrc/memattn.hip): [nsplit, M, K1] / [nsplit, M, 2]
    const unsigned short *w1, *b1;                      // [256, K1], [256]; w1 null: no product 1
    const unsigned short* res; long res_st;             // residual rows [M, 256] (the input rows when there is no product 1)
    unsigned short* x_out; long x_st;                   // x' rows (null: not written)
    const unsigned short *ln_w, *ln_b; float eps;
    unsigned short* t_out; long t_st;                   // LayerNorm output rows (null: not written)
    const unsigned short *w2, *b2; int N2;              // [N2, 256], [N2]; w2 null: no product 2
    unsigned short* y_out; long y_st;
    const float *cos, *sin; int rope_cols, nq;          // [nq, 128] f32 each
    int M;
};

// K1: width of product 1's operand (0: no product 1, 64: cross-attention partials or rows, 256: rows); NT2: 16-column tiles of product 2 per wave (0: none,
// 2: a 256-column group of product 2; blockIdx.y picks the group, so N2 = 768 is three workgroups per row block, each normalising its rows again -- holding all of
// a 768 x 256 weight matrix in one workgroup's registers does not fit); RT: 16-row tiles per workgroup.  8 waves: wave w owns columns [32 w, 32 w + 32) of x' / t
// and of its group of y, for all 16 RT rows.
// What was measured on the way (tools/memlayer_probe.py, device time inside a hipGraph):
//   * 4 waves x 16 rows walking product 2 tile by tile with one tile of prefetch: norm -> qkv -> RoPE 17.7 us against 15.0 us for the three launches it replaces --
//     a chain of L2 latencies; so EVERYTHING whose address does not depend on a previous stage is requested before the first wait (bias / residual / LayerNorm
//     vectors, all weight fragments of both products, the RoPE table entries);
//   * 8 waves x 16 rows, all requests up front: each product still cost ~4.5 us per 256 workgroups (norm only 3.6, norm -> q -> RoPE 7.8, norm -> qkv -> RoPE 18.6 us):
//     every workgroup pulls the whole weight matrix out of L2, 33 MB per product and launch through lines all CUs want at the same moment.  Hence RT row tiles per
//     workgroup: a weight fragment is loaded once and multiplied with RT token tiles.
constexpr int ML_NW = 8;
template <int K1, bool PARTIALS, int NT2, int RT>
__global__ __launch_bounds__(64 * ML_NW) void memlayer_rows_kernel(MemLayerArgs p) {
    constexpr int R = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* t1 = smem;                                   // operand of product 1  [R][ML_STR]
    char* t2 = smem + R * ML_STR;                      // LayerNorm output = operand of product 2
    float* red1 = (float*)(smem + 2 * R * ML_STR);     // [ML_NW][R]
    float* red2 = red1 + ML_NW * R;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * R;
    const int cg = blockIdx.y * ML_D;          // first column of this workgroup's group of product 2
    const bool first = blockIdx.y == 0;        // x' / t are written once
    auto lo_hi = [](const u32x2& v, int r) -> float { return __uint_as_float((r & 1) ? (v[r >> 1] & 0xffff0000u) : (v[r >> 1] << 16)); };

    // ---- requests that depend on nothing
    u32x2 bb1[2], gw[2], gb[2], rr[RT][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 32 * w + 16 * j + 4 * g;
        bb1[j] = (K1 > 0 && p.b1) ? *(const u32x2*)(p.b1 + col) : u32x2{0u, 0u};
        gw[j] = *(const u32x2*)(p.ln_w + col);
        gb[j] = p.ln_b ? *(const u32x2*)(p.ln_b + col) : u32x2{0u, 0u};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rr[rt][j] = *(const u32x2*)(p.res + (long)min(m0 + 16 * rt + c, p.M - 1) * p.res_st + col);
    }
    constexpr int KS1 = K1 / 32;
    bf16x8 afr[KS1 > 0 ? KS1 : 1][2];
    if constexpr (K1 > 0) {
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) afr[ks][j] = *(const bf16x8*)(p.w1 + (long)(32 * w + 16 * j + c) * K1 + ks * 32 + g * 8);
    }
    constexpr int NT2A = NT2 > 0 ? NT2 : 1;
    bf16x8 wf[NT2A][ML_D / 32];
    u32x2 bb2[NT2A];
    if constexpr (NT2 > 0) {
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int n0 = cg + w * (NT2 * 16) + 16 * j;
            const unsigned short* wrow = p.w2 + (long)(n0 + c) * ML_D + g * 8;
#pragma unroll
            for (int ks = 0; ks < ML_D / 32; ++ks) wf[j][ks] = *(const bf16x8*)(wrow + ks * 32);
            bb2[j] = p.b2 ? *(const u32x2*)(p.b2 + n0 + 4 * g) : u32x2{0u, 0u};
        }
    }

    // ---- operand of product 1 -> LDS
    if constexpr (K1 > 0) {
        if constexpr (PARTIALS) {
            constexpr int TPR = 512 / R, CPT = K1 / TPR;     // threads per row, columns per thread (K1 64, RT 1: 32 x 2; K1 256, RT 1: 32 x 8)
            const int r = tid / TPR, col = (tid % TPR) * CPT;
            const long row = min(m0 + r, p.M - 1);
            float mx = -INFINITY;
            for (int s = 0; s < p.nsplit; ++s) mx = fmaxf(mx, p.part_ml[((long)s * p.M + row) * 2]);
            float av[CPT], l = 0.f;
#pragma unroll
            for (int e = 0; e < CPT; ++e) av[e] = 0.f;
            for (int s = 0; s < p.nsplit; ++s) {
                const float2 ml = *(const float2*)(p.part_ml + ((long)s * p.M + row) * 2);
                const float wgt = (ml.x == -INFINITY) ? 0.f : exp2f(ml.x - mx);
                l += wgt * ml.y;
                const float* po = p.part_o + ((long)s * p.M + row) * K1 + col;
#pragma unroll
                for (int e = 0; e < CPT; e += 2) {
                    const float2 v = *(const float2*)(po + e);
                    av[e] += wgt * v.x;
                    av[e + 1] += wgt * v.y;
                }
            }
#pragma unroll
            for (int e = 0; e < CPT; e += 2) *(unsigned*)(t1 + r * ML_STR + (col + e) * 2) = l > 0.f ? pack_bf2(av[e] / l, av[e + 1] / l) : 0u;
        } else {
            constexpr int CH = K1 / 8;   // 16-byte chunks per row
#pragma unroll
            for (int i0 = 0; i0 < R * CH; i0 += 512) {
                const int i = i0 + tid;
                if ((R * CH) % 512 == 0 || i < R * CH) {
                    const int r = i / CH, ch = i - r * CH;
                    const long row = min(m0 + r, p.M - 1);
                    *(u32x4*)(t1 + r * ML_STR + ch * 16) = *(const u32x4*)(p.a + row * p.a_st + ch * 8);
                }
            }
        }
        __syncthreads();
    }

    // ---- product 1 + bias + residual: lane (c, g) ends with token 16 rt + c, columns 32 w + 16 j + 4 g .. + 3
    float xv[RT][2][4];
    if constexpr (K1 > 0) {
        f32x4 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 bfr = *(const bf16x8*)(t1 + (16 * rt + c) * ML_STR + ks * 64 + g * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][j], bfr, acc[rt][j], 0, 0, 0);
            }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = 32 * w + 16 * j + 4 * g, tok = m0 + 16 * rt + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lin = bf2f(f2bf(acc[rt][j][r] + lo_hi(bb1[j], r)));     // the linear output is rounded first (bf16 nn.Linear), then the sum
                    xv[rt][j][r] = bf2f(f2bf(lin + lo_hi(rr[rt][j], r)));
                }
                if (p.x_out && tok < p.M && first) {
                    u32x2 pk;
                    pk[0] = pack_bf2(xv[rt][j][0], xv[rt][j][1]);
                    pk[1] = pack_bf2(xv[rt][j][2], xv[rt][j][3]);
                    *(u32x2*)(p.x_out + (long)tok * p.x_st + col) = pk;
                }
            }
    } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[rt][j][r] = lo_hi(rr[rt][j], r);
    }

    // ---- LayerNorm over the 256 columns of each token: in-lane 8 values, the 4 lanes of a token (g), the 8 waves
    float mean[RT], rinv[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s1 += xv[rt][j][r];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (g == 0) red1[w * R + 16 * rt + c] = s1;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float m_ = 0.f;
#pragma unroll
        for (int i = 0; i < ML_NW; ++i) m_ += red1[i * R + 16 * rt + c];
        mean[rt] = m_ * (1.f / ML_D);
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = xv[rt][j][r] - mean[rt]; s2 += d * d; }
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) red2[w * R + 16 * rt + c] = s2;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < ML_NW; ++i) var += red2[i * R + 16 * rt + c];
        rinv[rt] = rsqrtf(var * (1.f / ML_D) + p.eps);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 32 * w + 16 * j + 4 * g, tok = m0 + 16 * rt + c;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (xv[rt][j][r] - mean[rt]) * rinv[rt] * lo_hi(gw[j], r) + lo_hi(gb[j], r);
            u32x2 pk;
            pk[0] = pack_bf2(y[0], y[1]);
            pk[1] = pack_bf2(y[2], y[3]);
            if constexpr (NT2 > 0) *(u32x2*)(t2 + (16 * rt + c) * ML_STR + col * 2) = pk;
            if (p.t_out && tok < p.M && first) *(u32x2*)(p.t_out + (long)tok * p.t_st + col) = pk;
        }
    }
    if constexpr (NT2 > 0) {
        __syncthreads();
        // ---- product 2 + bias (+ RoPE), one row tile at a time (the weight fragments stay in registers)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int tok = m0 + 16 * rt + c;
            const float* crow = p.cos + (long)(min(tok, p.M - 1) % p.nq) * (ML_D / 2);
            const float* srow = p.sin + (long)(min(tok, p.M - 1) % p.nq) * (ML_D / 2);
            float2 cs2[NT2], sn2[NT2];
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const int col = cg + w * (NT2 * 16) + 16 * j + 4 * g;
                cs2[j] = make_float2(1.f, 1.f);
                sn2[j] = make_float2(0.f, 0.f);
                if (col < p.rope_cols) {
                    const int pr = (col & (ML_D - 1)) >> 1;
                    cs2[j] = *(const float2*)(crow + pr);
                    sn2[j] = *(const float2*)(srow + pr);
                }
            }
            f32x4 acc[NT2];
#pragma unroll
            for (int j = 0; j < NT2; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < ML_D / 32; ++ks) {
                const bf16x8 tf = *(const bf16x8*)(t2 + (16 * rt + c) * ML_STR + ks * 64 + g * 16);
#pragma unroll
                for (int j = 0; j < NT2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], tf, acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NT2; ++j) {
                const int col = cg + w * (NT2 * 16) + 16 * j + 4 * g;
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = bf2f(f2bf(acc[j][r] + lo_hi(bb2[j], r)));
                if (col < p.rope_cols) {
                    const float2 cs = cs2[j], sn = sn2[j];
                    const float a0 = y[0] * cs.x - y[1] * sn.x, a1 = y[0] * sn.x + y[1] * cs.x;
                    const float a2 = y[2] * cs.y - y[3] * sn.y, a3 = y[2] * sn.y + y[3] * cs.y;
                    y[0] = a0; y[1] = a1; y[2] = a2; y[3] = a3;
                }
                if (tok < p.M) {
                    u32x2 pk;
                    pk[0] = pack_bf2(y[0], y[1]);
                    pk[1] = pack_bf2(y[2], y[3]);
                    *(u32x2*)(p.y_out + (long)tok * p.y_st + col) = pk;
                }
            }
        }
    }
}

template <int K1, bool PARTIALS, int NT2, int RT>
static int memlayer_launch(const MemLayerArgs& p, unsigned groups, hipStream_t st) {
    constexpr int R = 16 * RT, LDS = 2 * R * ML_STR + 2 * ML_NW * R * 4;
    auto kern = memlayer_rows_kernel<K1, PARTIALS, NT2, RT>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "memlayer_rows")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(p.M, R), groups), dim3(64 * ML_NW), LDS, st, p);
    RGA3_CHECK_LAUNCH("memlayer_rows_kernel");
    return 0;
}

template <int RT>
static int memlayer_dispatch(const MemLayerArgs& p, bool partials, int K1, hipStream_t st) {
    const unsigned groups = p.w2 ? (unsigned)(p.N2 / ML_D) : 1u;
    if (!p.w1) return p.w2 ? memlayer_launch<0, false, 2, RT>(p, groups, st) : memlayer_launch<0, false, 0, RT>(p, groups, st);
    if (partials) return p.w2 ? memlayer_launch<64, true, 2, RT>(p, groups, st) : memlayer_launch<64, true, 0, RT>(p, groups, st);
    if (K1 == 64) return p.w2 ? memlayer_launch<64, false, 2, RT>(p, groups, st) : memlayer_launch<64, false, 0, RT>(p, groups, st);
    return p.w2 ? memlayer_launch<256, false, 2, RT>(p, groups, st) : memlayer_launch<256, false, 0, RT>(p, groups, st);
}

}  // namespace rga3

using namespace rga3;

// One launch for the row-wise chain between the attention kernels of a memory-attention layer (model width 256).  Every stage but the LayerNorm is optional:
//   a / K1 (rows [M, K1], K1 = 64 or 256) or part_o / part_ml / nsplit (the workspace halves rga3_memattn_cross leaves when called with out = NULL), w1 [256, K1], b1:
//       x' = bf16(bf16(a w1^T + b1) + res), written to x_out when given;  w1 = NULL: x' = res;
//   t = LayerNorm(x'; ln_w, ln_b, eps), written to t_out when given;
//   w2 [N2, 256] (N2 = 256 or 768), b2: y = bf16(t w2^T + b2), columns < rope_cols rotated by the axial tables cos / sin [nq, 128] (row = token % nq), written to y_out.
// Row strides in elements, multiples of 4; weights contiguous and 16-byte aligned.
extern "C" int rga3_memlayer_rows(const void* a, int64_t a_stride, int K1, const float* part_o, const float* part_ml, int nsplit, const void* w1, const void* b1,
                                  const void* res, int64_t res_stride, void* x_out, int64_t x_stride, const void* ln_w, const void* ln_b, float eps, void* t_out,
                                  int64_t t_stride, const void* w2, const void* b2, int N2, void* y_out, int64_t y_stride, const float* cos, const float* sin,
                                  int rope_cols, int nq, int64_t M, void* stream) {
    RGA3_CHECK_ARG(res && ln_w && M > 0 && M < (1 << 30), "memlayer_rows: res / ln_w / M %ld", (long)M);
    RGA3_CHECK_ARG(res_stride >= ML_D && res_stride % 4 == 0, "memlayer_rows: residual stride");
    const bool partials = part_o != nullptr;
    if (w1) {
        RGA3_CHECK_ARG(partials ? (part_ml && nsplit >= 1 && nsplit <= 32 && K1 == 64 && !a) : (a && (K1 == 64 || K1 == 256) && a_stride >= K1 && a_stride % 8 == 0),
                       "memlayer_rows: operand of product 1 (K1 %d, nsplit %d)", K1, nsplit);
        RGA3_CHECK_ARG((((uintptr_t)w1 | (uintptr_t)a | (uintptr_t)part_o) & 15) == 0 && (((uintptr_t)b1 | (uintptr_t)part_ml) & 7) == 0, "memlayer_rows: alignment of product 1");
        RGA3_CHECK_ARG(!x_out || (x_stride >= ML_D && x_stride % 4 == 0), "memlayer_rows: x stride");
    } else {
        RGA3_CHECK_ARG(!a && !partials && !x_out, "memlayer_rows: operand / x_out without w1");
    }
    RGA3_CHECK_ARG(!t_out || (t_stride >= ML_D && t_stride % 4 == 0), "memlayer_rows: t stride");
    if (w2) {
        RGA3_CHECK_ARG(y_out && (N2 == 256 || N2 == 768) && y_stride >= N2 && y_stride % 4 == 0, "memlayer_rows: product 2 (N2 %d)", N2);
        RGA3_CHECK_ARG(rope_cols >= 0 && rope_cols <= N2 && rope_cols % 4 == 0 && (rope_cols == 0 || (cos && sin && nq > 0)), "memlayer_rows: rope (%d columns)", rope_cols);
        RGA3_CHECK_ARG((((uintptr_t)w2) & 15) == 0 && (((uintptr_t)b2 | (uintptr_t)cos | (uintptr_t)sin) & 7) == 0, "memlayer_rows: alignment of product 2");
    } else {
        RGA3_CHECK_ARG(t_out, "memlayer_rows: nothing to write");
    }
    RGA3_CHECK_ARG((((uintptr_t)res | (uintptr_t)x_out | (uintptr_t)t_out | (uintptr_t)y_out | (uintptr_t)ln_w | (uintptr_t)ln_b) & 7) == 0, "memlayer_rows: 8-byte alignment");
    MemLayerArgs p;
    p.a = (const unsigned short*)a; p.a_st = a_stride;
    p.part_o = part_o; p.part_ml = part_ml; p.nsplit = nsplit;
    p.w1 = (const unsigned short*)w1; p.b1 = (const unsigned short*)b1;
    p.res = (const unsigned short*)res; p.res_st = res_stride;
    p.x_out = (unsigned short*)x_out; p.x_st = x_stride;
    p.ln_w = (const unsigned short*)ln_w; p.ln_b = (const unsigned short*)ln_b; p.eps = eps;
    p.t_out = (unsigned short*)t_out; p.t_st = t_stride;
    p.w2 = (const unsigned short*)w2; p.b2 = (const unsigned short*)b2; p.N2 = w2 ? N2 : 0;
    p.y_out = (unsigned short*)y_out; p.y_st = y_stride;
    p.cos = cos; p.sin = sin; p.rope_cols = w2 ? rope_cols : 0; p.nq = nq > 0 ? nq : 1;
    p.M = (int)M;
    hipStream_t st = (hipStream_t)stream;
    // rows per workgroup: 16 where a product-1 weight matrix has to be pulled by every workgroup anyway (more workgroups = more CUs pulling in parallel), 64 for
    // the 768-wide product 2 alone (three column groups: 192 workgroups; tools/memlayer_probe.py: 15.6 vs 19.3 us)
    int rt = (!w1 && w2 && N2 > ML_D) ? 4 : 1;
#ifdef RGA3_AB   // measurement builds only (tools/memlayer_probe.py)
    { const char* e = getenv("RGA3_ML_RT"); if (e) rt = atoi(e); }
#endif
    if (rt == 1) return memlayer_dispatch<1>(p, partials, K1, st);
    if (rt == 2) return memlayer_dispatch<2>(p, partials, K1, st);
    return memlayer_dispatch<4>(p, partials, K1, st);
}
