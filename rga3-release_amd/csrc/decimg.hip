// Image side of a mask-decoder layer boundary at inference (reference model/sam2.py:1926-2100 TwoWayAttentionBlock.forward: cross_attn_image_to_token + norm4, then the
// next block's k / v projections for cross_attn_token_to_image -- or those of TwoWayTransformer.final_attn_token_to_image; Attention :1417-1481).
//
// For an image token everything here is row-local: its query is a projection of (keys + key_pe) [256 -> 128], it attends to the frame's 9 output / prompt tokens
// (8 heads x 16), the result goes through out_proj [128 -> 256] + residual + LayerNorm, and the next attention needs k = (keys' + key_pe) Wk^T and v = keys' Wv^T of the
// updated row.  As launches that is add_bcast, q_proj, the windowed attention kernel, out_proj, LayerNorm, add_bcast, k_proj, v_proj: eight launches over [4096, 256]
// rows, 8 - 9 us each for the products (every workgroup of a 4096 x 256 x 128 product pulls its weight tile out of L2).  Here a workgroup owns 16 rows and runs the
// chain; the token-side keys / values (9 x 128 per frame) come in already projected.
//
// Layout (8 waves): tokens on the MFMA's N side as in memlayer.hip; after the q projection wave w holds columns 16 w .. 16 w + 15 of its 16 rows -- exactly head w --
// so the 9-key attention is lane-local arithmetic plus two shuffles per key (a lane has 4 of the head's 16 dims).  All four weight matrices (4 x 64 KB) are requested
// before the first wait.  Rounding points are those of the launches replaced: bf16(keys + pe); bf16 q; softmax numerators rounded to bf16 for the value sum, the row
// sum kept in f32 (the attention kernel's arithmetic); bf16(bf16(o Wo^T + bo) + keys); LayerNorm on those rows; bf16(keys' + pe); bf16 k, v.
#include "common.h"

namespace rga3 {

constexpr int DI_R = 16, DI_D = 256, DI_I = 128, DI_STR = DI_D * 2 + 16, DI_OSTR = DI_I * 2 + 16, DI_MAXK = 16;

struct DecImgArgs {
    const unsigned short* keys; long keys_st;      // [M, 256] image rows (residual of the attention, input of the LayerNorm)
    const unsigned short* pe; long pe_st; int hw;  // [hw, 256] dense positional encoding, row r uses pe[r % hw]
    const unsigned short *kt, *vt;                 // token-side keys / values, already projected: [B * nk, 128]
    int nk;                                        // tokens per frame (<= 16)
    const unsigned short *wq, *bq;                 // cross_attn_image_to_token.q_proj [128, 256]
    const unsigned short *wo, *bo;                 // ... out_proj [256, 128]
    const unsigned short *ln_w, *ln_b; float eps;  // norm4
    const unsigned short *wk2, *bk2, *wv2, *bv2;   // the NEXT token-to-image attention's k_proj / v_proj [128, 256] (null: not wanted)
    unsigned short* keys_out; long ko_st;          // [M, 256]
    unsigned short *k2, *v2; long kv_st;           // [M, 128] each; v2_t != 0: v2 is written TRANSPOSED, [frames * 128, hw], and k2 head-major, [8][M][16] (what rga3_attn_fewq reads best)
    int v2_t;
    float scale_log2;
    int M;
};

__global__ __launch_bounds__(512) void decimg_rows_kernel(DecImgArgs p) {
    __shared__ __attribute__((aligned(16))) char ta[DI_R * DI_STR];     // bf16(keys + pe), later bf16(keys' + pe)
    __shared__ __attribute__((aligned(16))) char tb[DI_R * DI_STR];     // keys'
    __shared__ __attribute__((aligned(16))) char to[DI_R * DI_OSTR];    // attention output [16, 128]
    __shared__ float red1[8][DI_R], red2[8][DI_R];
    __shared__ __attribute__((aligned(16))) unsigned short tkt[2 * DI_MAXK * DI_I], tvt[2 * DI_MAXK * DI_I];   // token-side keys / values of <= 2 frames
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * DI_R;
    const int tok = min(m0 + c, p.M - 1);
    const bool tok_ok = m0 + c < p.M;
    auto lo_hi = [](const u32x2& v, int r) -> float { return __uint_as_float((r & 1) ? (v[r >> 1] & 0xffff0000u) : (v[r >> 1] << 16)); };

    // ---- requests that depend on nothing: weight fragments of all four products, small vectors, this lane's token-side keys / values
    bf16x8 fq[8], fo[2][4], fk[8], fv[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) fq[ks] = *(const bf16x8*)(p.wq + (long)(16 * w + c) * DI_D + ks * 32 + g * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fo[j][ks] = *(const bf16x8*)(p.wo + (long)(32 * w + 16 * j + c) * DI_I + ks * 32 + g * 8);
    const bool next = p.wk2 != nullptr;
    if (next) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            fk[ks] = *(const bf16x8*)(p.wk2 + (long)(16 * w + c) * DI_D + ks * 32 + g * 8);
            fv[ks] = *(const bf16x8*)(p.wv2 + (long)(16 * w + c) * DI_D + ks * 32 + g * 8);
        }
    }
    const int hcol = 16 * w + 4 * g;                    // this lane's 4 columns of the 128-wide head space (head w, dims 4 g .. 4 g + 3)
    const u32x2 bqv = p.bq ? *(const u32x2*)(p.bq + hcol) : u32x2{0u, 0u};
    u32x2 bov[2], gw[2], gb[2], rr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 32 * w + 16 * j + 4 * g;
        bov[j] = p.bo ? *(const u32x2*)(p.bo + col) : u32x2{0u, 0u};
        gw[j] = *(const u32x2*)(p.ln_w + col);
        gb[j] = p.ln_b ? *(const u32x2*)(p.ln_b + col) : u32x2{0u, 0u};
        rr[j] = *(const u32x2*)(p.keys + (long)tok * p.keys_st + col);
    }
    u32x2 bk2v = {0u, 0u}, bv2v = {0u, 0u};
    if (next) {
        if (p.bk2) bk2v = *(const u32x2*)(p.bk2 + hcol);
        if (p.bv2) bv2v = *(const u32x2*)(p.bv2 + hcol);
    }
    // the token-side keys / values of the frame(s) this row block touches -> LDS (a 16-row block lies in one frame unless hw is not a multiple of 16: then rows
    // read their own frame's copy below, staged per distinct frame: two at most)
    const int f0 = m0 / p.hw, f1 = min(m0 + DI_R - 1, p.M - 1) / p.hw;
    for (int i = tid; i < (f1 - f0 + 1) * p.nk * (DI_I / 4); i += 512) {
        const int fr = i / (p.nk * (DI_I / 4)), rem = i - fr * (p.nk * (DI_I / 4));
        const int j = rem / (DI_I / 4), q4 = rem - j * (DI_I / 4);
        if (fr < 2) {
            *(u32x2*)(tkt + ((fr * DI_MAXK + j) * DI_I + q4 * 4)) = *(const u32x2*)(p.kt + ((long)(f0 + fr) * p.nk + j) * DI_I + q4 * 4);
            *(u32x2*)(tvt + ((fr * DI_MAXK + j) * DI_I + q4 * 4)) = *(const u32x2*)(p.vt + ((long)(f0 + fr) * p.nk + j) * DI_I + q4 * 4);
        }
    }
    const int fsel = min(tok / p.hw - f0, 1);

    // ---- bf16(keys + pe) -> LDS (16 rows x 32 chunks = one chunk per thread)
    {
        const int r = tid >> 5, ch = tid & 31;
        const long row = min(m0 + r, p.M - 1);
        const u32x4 a = *(const u32x4*)(p.keys + row * p.keys_st + ch * 8);
        const u32x4 b = *(const u32x4*)(p.pe + (row % p.hw) * p.pe_st + ch * 8);
        u32x4 s;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            s[e] = pack_bf2(__uint_as_float(a[e] << 16) + __uint_as_float(b[e] << 16), __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(b[e] & 0xffff0000u));
        *(u32x4*)(ta + r * DI_STR + ch * 16) = s;
    }
    __syncthreads();

    // ---- q = bf16(kin Wq^T + bq): wave w = head w; lane (c, g): row c, dims 4 g .. 4 g + 3
    float q[4];
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[ks], *(const bf16x8*)(ta + c * DI_STR + ks * 64 + g * 16), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = bf2f(f2bf(acc[r] + lo_hi(bqv, r)));
    }
    // ---- attention over the frame's nk tokens: scores in the log2 domain, numerators rounded to bf16 for the value sum, f32 row sum
    {
        float s[DI_MAXK];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < DI_MAXK; ++j) {
            float d = 0.f;
            const u32x2 kv_ = *(const u32x2*)(tkt + (fsel * DI_MAXK + min(j, p.nk - 1)) * DI_I + hcol);
#pragma unroll
            for (int r = 0; r < 4; ++r) d += q[r] * lo_hi(kv_, r);
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
            s[j] = j < p.nk ? d * p.scale_log2 : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        float l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < DI_MAXK; ++j) {
            const float e = __builtin_amdgcn_exp2f(s[j] - mx);       // -inf -> 0 for the unused slots
            l += e;
            const float eb = bf2f(f2bf(e));
            const u32x2 vv_ = *(const u32x2*)(tvt + (fsel * DI_MAXK + min(j, p.nk - 1)) * DI_I + hcol);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += eb * lo_hi(vv_, r);
        }
        const float inv = 1.f / l;
        u32x2 pk;
        pk[0] = pack_bf2(o[0] * inv, o[1] * inv);
        pk[1] = pack_bf2(o[2] * inv, o[3] * inv);
        *(u32x2*)(to + c * DI_OSTR + hcol * 2) = pk;
    }
    __syncthreads();

    // ---- x' = bf16(bf16(o Wo^T + bo) + keys): wave w owns columns 32 w .. 32 w + 31
    float xv[2][4];
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 bfr = *(const bf16x8*)(to + c * DI_OSTR + ks * 64 + g * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo[j][ks], bfr, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[j][r] = bf2f(f2bf(bf2f(f2bf(acc[j][r] + lo_hi(bov[j], r))) + lo_hi(rr[j], r)));
    }
    // ---- LayerNorm over the 256 columns of each row
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1 += xv[j][r];
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (g == 0) red1[w][c] = s1;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mean += red1[i][c];
    mean *= 1.f / DI_D;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = xv[j][r] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (g == 0) red2[w][c] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) var += red2[i][c];
    const float rinv = rsqrtf(var * (1.f / DI_D) + p.eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 32 * w + 16 * j + 4 * g;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (xv[j][r] - mean) * rinv * lo_hi(gw[j], r) + lo_hi(gb[j], r);
        u32x2 pk;
        pk[0] = pack_bf2(y[0], y[1]);
        pk[1] = pack_bf2(y[2], y[3]);
        if (tok_ok) *(u32x2*)(p.keys_out + (long)tok * p.ko_st + col) = pk;
        if (next) {
            *(u32x2*)(tb + c * DI_STR + col * 2) = pk;
            // bf16(keys' + pe) for the k projection: same lane, same columns
            const u32x2 pv = *(const u32x2*)(p.pe + (long)(tok % p.hw) * p.pe_st + col);
            u32x2 sk;
            sk[0] = pack_bf2(__uint_as_float(pk[0] << 16) + __uint_as_float(pv[0] << 16), __uint_as_float(pk[0] & 0xffff0000u) + __uint_as_float(pv[0] & 0xffff0000u));
            sk[1] = pack_bf2(__uint_as_float(pk[1] << 16) + __uint_as_float(pv[1] << 16), __uint_as_float(pk[1] & 0xffff0000u) + __uint_as_float(pv[1] & 0xffff0000u));
            *(u32x2*)(ta + c * DI_STR + col * 2) = sk;
        }
    }
    if (!next) return;
    __syncthreads();
    // ---- k2 = bf16((keys' + pe) Wk2^T + bk2), v2 = bf16(keys' Wv2^T + bv2): wave w owns columns 16 w .. 16 w + 15 of each
    {
        f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks], *(const bf16x8*)(ta + c * DI_STR + ks * 64 + g * 16), ak, 0, 0, 0);
            // (both fragments have the same per-lane shape -- row c, k-chunk g -- so swapping them transposes the result: lane (c, g) then holds weight column c,
            //  tokens 4 g .. 4 g + 3)
            if (p.v2_t) av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tb + c * DI_STR + ks * 64 + g * 16), fv[ks], av, 0, 0, 0);
            else av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[ks], *(const bf16x8*)(tb + c * DI_STR + ks * 64 + g * 16), av, 0, 0, 0);
        }
        if (p.v2_t) {      // rows of v2^T: (frame, column 16 w + c), 4 consecutive tokens per lane; the block lies in one frame (hw % 16 == 0, checked by the entry point)
            const float bv = p.bv2 ? bf2f(p.bv2[16 * w + c]) : 0.f;
            const int t0 = m0 + 4 * g;
            if (t0 < p.M) {
                u32x2 pk;
                pk[0] = pack_bf2(av[0] + bv, av[1] + bv);
                pk[1] = pack_bf2(av[2] + bv, av[3] + bv);
                *(u32x2*)(p.v2 + ((long)(m0 / p.hw) * DI_I + 16 * w + c) * p.hw + (t0 % p.hw)) = pk;
            }
        }
        if (tok_ok) {
            u32x2 pk;
            pk[0] = pack_bf2(ak[0] + lo_hi(bk2v, 0), ak[1] + lo_hi(bk2v, 1));
            pk[1] = pack_bf2(ak[2] + lo_hi(bk2v, 2), ak[3] + lo_hi(bk2v, 3));
            if (p.v2_t) *(u32x2*)(p.k2 + ((long)w * p.M + tok) * 16 + 4 * g) = pk;      // head-major [8][M][16]: a head's keys are contiguous for rga3_attn_fewq
            else *(u32x2*)(p.k2 + (long)tok * p.kv_st + hcol) = pk;
            if (!p.v2_t) {
                pk[0] = pack_bf2(av[0] + lo_hi(bv2v, 0), av[1] + lo_hi(bv2v, 1));
                pk[1] = pack_bf2(av[2] + lo_hi(bv2v, 2), av[3] + lo_hi(bv2v, 3));
                *(u32x2*)(p.v2 + (long)tok * p.kv_st + hcol) = pk;
            }
        }
    }
}

// ---- Attention of a FEW queries over many keys (reference model/sam2.py:1417-1481 Attention.forward as cross_attn_token_to_image / final_attn_token_to_image: the
// frame's 9 tokens x 4096 image keys, 8 heads x 16).  The general kernel cuts the keys over workgroups and needs a merge launch (~10 + 7 us); here one workgroup per
// (frame, head) holds everything: 16 waves x 16-key tiles, S^T = K Q^T by 16x16x16 MFMAs straight from global memory (K rows as they are, V TRANSPOSED so that a lane
// reads 4 consecutive keys of one dim), all scores of a wave kept in registers, ONE maximum per query over all keys (two shuffles + an LDS exchange -- no online
// rescaling), numerators rounded to bf16 for the value product (the S^T accumulator layout IS the next MFMA's B operand), partial sums of the 16 waves added in
// wave order.  The value bias is added after the normalisation (softmax rows sum to one), so v^T can come from a product without a per-row bias.
constexpr int FQ_NW = 16, FQ_MAXT = 16;      // waves; key tiles per wave (16 x 16 x 16 = 4096 keys)
struct FewQArgs {
    const unsigned short* q; long q_st;      // [frames * nq, H * 16] projected queries
    const unsigned short* k; long k_st, k_hst; // projected keys: element (frame f, key j, head h, dim d) at ((f * nk + j) * k_st + h * k_hst + d)
    const unsigned short* vt;                // [frames * H * 16, nk] projected values, transposed
    const unsigned short* vbias;             // [H * 16] or null
    unsigned short* o; long o_st;            // [frames * nq, H * 16]
    int nq, nk, H;
    float scale_log2;
};

__global__ __launch_bounds__(64 * FQ_NW) void attn_fewq_kernel(FewQArgs p) {
    __shared__ float smax[FQ_NW][16];
    __shared__ float sl[FQ_NW][16];
    __shared__ f32x4 so[FQ_NW][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, f = blockIdx.y;
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    // B operand of S^T = K Q^T: lane (c, g) holds Q[q c][4 g .. 4 g + 3] of this head (queries past nq: zeros)
    s16x4 qf = {0, 0, 0, 0};
    if (c < p.nq) qf = *(const s16x4*)(p.q + ((long)f * p.nq + c) * p.q_st + h * 16 + 4 * g);
    const int ntile = (p.nk + 15) >> 4;
    const unsigned short* kb = p.k + (long)f * p.nk * p.k_st + (long)h * p.k_hst + 4 * g;
    const unsigned short* vb = p.vt + ((long)(f * p.H + h) * 16 + c) * p.nk + 4 * g;
    s16x4 kf[FQ_MAXT], vf[FQ_MAXT];
#pragma unroll
    for (int i = 0; i < FQ_MAXT; ++i) {
        const int t = w + i * FQ_NW;
        kf[i] = s16x4{0, 0, 0, 0};
        vf[i] = s16x4{0, 0, 0, 0};
        if (t < ntile) {
            kf[i] = *(const s16x4*)(kb + (long)min(16 * t + c, p.nk - 1) * p.k_st);
            if (16 * t + 4 * g + 3 < p.nk) {
                vf[i] = *(const s16x4*)(vb + 16 * t);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) vf[i][r] = (16 * t + 4 * g + r < p.nk) ? (short)vb[16 * t + r] : (short)0;
            }
        }
    }
    // ---- scores: lane (c = query, g) holds keys 16 t + 4 g + r of tile t
    f32x4 s[FQ_MAXT];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < FQ_MAXT; ++i) {
        const int t = w + i * FQ_NW;
        s[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf[i], qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = t < ntile && 16 * t + 4 * g + r < p.nk;
            s[i][r] = ok ? s[i][r] * p.scale_log2 : -INFINITY;
            mx = fmaxf(mx, s[i][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (g == 0) smax[w][c] = mx;
    __syncthreads();
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < FQ_NW; ++i) m = fmaxf(m, smax[i][c]);
    // ---- numerators, row sums, O^T += V^T P^T
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < FQ_MAXT; ++i) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[r] = __builtin_amdgcn_exp2f(s[i][r] - m);      // -inf -> 0
            l += e[r];
        }
        const unsigned p01 = pack_bf2(e[0], e[1]), p23 = pack_bf2(e[2], e[3]);
        s16x4 pf;
        pf[0] = (short)(p01 & 0xffffu); pf[1] = (short)(p01 >> 16); pf[2] = (short)(p23 & 0xffffu); pf[3] = (short)(p23 >> 16);
        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[i], pf, o, 0, 0, 0);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (g == 0) sl[w][c] = l;
    so[w][lane] = o;                                         // lane (c = query, g): dims 4 g .. 4 g + 3
    __syncthreads();
    if (w != 0) return;
    float lt = 0.f;
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FQ_NW; ++i) {
        lt += sl[i][c];
        ot += so[i][lane];
    }
    if (c >= p.nq) return;
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.vbias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = bf2f(p.vbias[h * 16 + 4 * g + r]);
    }
    u32x2 pk;
    pk[0] = pack_bf2(ot[0] * inv + bv[0], ot[1] * inv + bv[1]);
    pk[1] = pack_bf2(ot[2] * inv + bv[2], ot[3] * inv + bv[3]);
    *(u32x2*)(p.o + ((long)f * p.nq + c) * p.o_st + h * 16 + 4 * g) = pk;
}

}  // namespace rga3

using namespace rga3;

// keys' [M, 256] = LayerNorm(bf16(bf16(attn((keys + pe) Wq^T + bq; kt, vt) Wo^T + bo) + keys)) and, when wk2 / wv2 are given, k2 = (keys' + pe) Wk2^T + bk2 and
// v2 = keys' Wv2^T + bv2 [M, 128] -- the image side of a two-way block boundary in one launch.  Model width 256, internal width 128 = 8 heads x 16, nk <= 16 tokens per
// frame of hw image rows; kt / vt [B * nk, 128] contiguous; weights contiguous; row strides in elements (multiples of 8).
extern "C" int rga3_decimg_rows(const void* keys, int64_t keys_stride, const void* pe, int64_t pe_stride, int hw, const void* kt, const void* vt, int nk, const void* wq,
                                const void* bq, const void* wo, const void* bo, const void* ln_w, const void* ln_b, float eps, const void* wk2, const void* bk2,
                                const void* wv2, const void* bv2, void* keys_out, int64_t keys_out_stride, void* k2, void* v2, int64_t kv_stride, int v2_transposed,
                                float scale, int64_t M, void* stream) {
    RGA3_CHECK_ARG(keys && pe && kt && vt && wq && wo && ln_w && keys_out && M > 0 && M < (1LL << 31), "decimg_rows: null pointer / M");
    RGA3_CHECK_ARG(hw >= DI_R && M % hw == 0 && nk >= 1 && nk <= DI_MAXK, "decimg_rows: hw %d, nk %d (1..16)", hw, nk);
    RGA3_CHECK_ARG(keys_stride >= DI_D && keys_stride % 8 == 0 && pe_stride >= DI_D && pe_stride % 8 == 0 && keys_out_stride >= DI_D && keys_out_stride % 4 == 0, "decimg_rows: strides");
    RGA3_CHECK_ARG((wk2 == nullptr) == (wv2 == nullptr) && (!wk2 || (k2 && v2 && kv_stride >= DI_I && kv_stride % 4 == 0)), "decimg_rows: next projections");
    RGA3_CHECK_ARG(scale > 0.f, "decimg_rows: scale");
    RGA3_CHECK_ARG((((uintptr_t)keys | (uintptr_t)pe | (uintptr_t)wq | (uintptr_t)wo | (uintptr_t)wk2 | (uintptr_t)wv2) & 15) == 0 &&
                       (((uintptr_t)kt | (uintptr_t)vt | (uintptr_t)bq | (uintptr_t)bo | (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)bk2 | (uintptr_t)bv2 | (uintptr_t)keys_out |
                         (uintptr_t)k2 | (uintptr_t)v2) & 7) == 0,
                   "decimg_rows: alignment");
    DecImgArgs a;
    a.keys = (const unsigned short*)keys; a.keys_st = keys_stride; a.pe = (const unsigned short*)pe; a.pe_st = pe_stride; a.hw = hw;
    a.kt = (const unsigned short*)kt; a.vt = (const unsigned short*)vt; a.nk = nk;
    a.wq = (const unsigned short*)wq; a.bq = (const unsigned short*)bq; a.wo = (const unsigned short*)wo; a.bo = (const unsigned short*)bo;
    a.ln_w = (const unsigned short*)ln_w; a.ln_b = (const unsigned short*)ln_b; a.eps = eps;
    a.wk2 = (const unsigned short*)wk2; a.bk2 = (const unsigned short*)bk2; a.wv2 = (const unsigned short*)wv2; a.bv2 = (const unsigned short*)bv2;
    RGA3_CHECK_ARG(!v2_transposed || (wk2 && hw % 16 == 0), "decimg_rows: the transposed v2 needs hw %% 16 == 0");
    a.keys_out = (unsigned short*)keys_out; a.ko_st = keys_out_stride; a.k2 = (unsigned short*)k2; a.v2 = (unsigned short*)v2; a.kv_st = kv_stride; a.v2_t = v2_transposed;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.M = (int)M;
    hipLaunchKernelGGL(decimg_rows_kernel, dim3((unsigned)cdiv(M, DI_R)), dim3(512), 0, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("decimg_rows_kernel");
    return 0;
}

// out [frames * nq, H * 16] bf16 = softmax(scale q k^T) v (+ vbias) per frame and head of 16 dims, for nq <= 16 queries over nk <= 4096 keys per frame:
// q [frames * nq, H * 16] (row stride q_stride); keys: element (frame, key j, head h, dim d) at ((frame * nk + j) * k_stride + h * k_head_stride + d) -- row-major
// [frames * nk, H * 16]: (H * 16, 16); head-major [H][frames * nk][16]: (16, frames * nk * 16) -- a head's keys are then contiguous and a workgroup pulls 32 B per key
// instead of a 128-byte line per key; vt [frames * H * 16, nk] the values TRANSPOSED (contiguous), nk % 4 == 0.
extern "C" int rga3_attn_fewq(const void* q, int64_t q_stride, const void* k, int64_t k_stride, int64_t k_head_stride, const void* vt, const void* vbias, void* out,
                              int64_t out_stride, int frames, int nq, int nk, int H, float scale, void* stream) {
    RGA3_CHECK_ARG(q && k && vt && out && frames >= 1 && frames <= 65535 && H >= 1 && H <= 65535, "attn_fewq: null pointer / frames / heads");
    RGA3_CHECK_ARG(nq >= 1 && nq <= 16 && nk >= 1 && nk <= 16 * FQ_NW * FQ_MAXT && nk % 4 == 0, "attn_fewq: nq %d (1..16), nk %d (<= 4096, multiple of 4)", nq, nk);
    RGA3_CHECK_ARG(q_stride % 4 == 0 && k_stride % 4 == 0 && k_head_stride % 4 == 0 && out_stride % 4 == 0 && q_stride >= 16L * H && k_stride >= 16 && out_stride >= 16L * H,
                   "attn_fewq: strides");
    RGA3_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 7) == 0 && scale > 0.f, "attn_fewq: alignment / scale");
    FewQArgs a;
    a.q = (const unsigned short*)q; a.q_st = q_stride; a.k = (const unsigned short*)k; a.k_st = k_stride; a.k_hst = k_head_stride; a.vt = (const unsigned short*)vt; a.vbias = (const unsigned short*)vbias;
    a.o = (unsigned short*)out; a.o_st = out_stride; a.nq = nq; a.nk = nk; a.H = H; a.scale_log2 = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(attn_fewq_kernel, dim3((unsigned)H, (unsigned)frames), dim3(64 * FQ_NW), 0, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("attn_fewq_kernel");
    return 0;
}
