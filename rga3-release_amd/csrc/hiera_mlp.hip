// Fused MLP of a Hiera stage-1 block (reference model/sam2.py:1035-1117 MultiScaleBlock.forward: x = x + mlp(norm2(x)); MLP :2305-2329, dim 144 -> 576 -> 144, exact-erf
// GELU) for the frozen SAM2-L trunk:   y = x + W2 gelu(W1 LayerNorm(x) + b1) + b2   in ONE launch.
//
// Why: at stage 1 a 1024^2 frame is 65 536 tokens x 144 channels; as two GEMMs the 576-wide hidden activation of 16 frames (1.2 GB) is written and read back, and
// both products run HBM-bound at 2 - 3.8 TB/s (fc1 + GELU 0.73 ms, fc2 0.48 ms per block and 16 frames, profiles/r02_train_gemm_shapes.txt).  Fused, the hidden
// activation never leaves the CU: HBM traffic is x in + y out (0.6 GB), and the LayerNorm statistics pass disappears as well (a workgroup holds whole rows).
//
// Dataflow -- "tokens on the lanes": every product is computed TRANSPOSED with v_mfma_f32_32x32x16_bf16 so that the activations are always the B operand and the
// weights the A operand read from LDS:
//     H^T[hidden, tok] = W1'[hidden, ch] . X^T[ch, tok]        X^T fragments live in registers for the whole kernel (9 k-steps x 4 VGPRs per 32 tokens)
//     Y^T[ch, tok]    += W2[ch, hidden]  . G^T[hidden, tok]     G = gelu(LN-fold(H)) taken STRAIGHT from H's accumulator registers: a 32 x 32 result has its
//                                                               column (token) on the lane and its rows (hidden) in the 16 registers, so registers 8s .. 8s+7,
//                                                               rounded to bf16, are k-step s of the next B operand with the k order permuted (cdna_hip_programming.md
//                                                               3, "An accumulator tile as the next MFMA's operand"); the A operand W2 is read with that permutation
//                                                               (two ds_read_b64 per fragment).  No LDS round trip, no barrier between the two products.
// LayerNorm is folded as in rga3_gemm_ln_bf16: W1' = W1 diag(gamma) (bf16), h = rinv (acc - mean c_n) + d_n with c = row sums of W1', d = beta W1^T + b1; mean / rinv
// of a token are lane-local (each half-wave holds half of its row, one exchange).
// Workgroup = 8 waves x 32 tokens; the 576 hidden units stream through LDS in 9 chunks of 64 (W1' chunk 64 x 144, W2 chunk 144 x 64, their c / d), register-staged
// double buffer, one barrier per chunk; per chunk and wave 18 + 20 MFMAs against 38 KiB of LDS fragment reads (LDS array at ~50 %).  LDS images: W1' rows padded to
// 304 B (19 x 16 B, odd: conflict-free ds_read_b128), W2 rows to 136 B (34 dwords: the 32 rows of a half-wave's ds_read_b64 tile all 64 banks).  The output tile is
// transposed through LDS so that y leaves in whole 288-byte rows.
#include "common.h"

namespace rga3 {

constexpr int HM_C = 144, HM_H = 576, HM_TOK = 256, HM_HC = 64;
constexpr int HM_KS = HM_C / 16;                       // 9 k-steps of the first product
constexpr int HM_CB = 5;                               // channel blocks of 32 (144 -> 160: rows 144..159 are never stored)
constexpr int HM_W1STR = HM_C * 2 + 16;                // 304 B
constexpr int HM_W2STR = HM_HC * 2 + 8;                // 136 B
constexpr int HM_W1B = HM_HC * HM_W1STR;               // 19 456
constexpr int HM_W2B = HM_CB * 32 * HM_W2STR;          // 21 760
constexpr int HM_STAGE = HM_W1B + HM_W2B + HM_HC * 8;  // + c (f32) and d (f32) of the chunk = 41 728
constexpr int HM_OSTR = HM_C * 2 + 16;                 // output tile rows (304 B)
constexpr int HM_LDS = 2 * HM_STAGE;                   // 83 456 >= 256 * 304 = 77 824 (the output tile reuses it)

__device__ __forceinline__ float hm_gelu_erf(float x) {   // the GEMM epilogue's arithmetic (gemm_bf16.hip gelu_erf): Abramowitz-Stegun 7.1.26
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __expf(-z * z);
    return 0.5f * x + 0.5f * fabsf(x) * e;
}

struct HmArgs {
    const unsigned short* x;     // [M, 144] bf16
    const unsigned short* w1f;   // [576, 144] bf16 = W1 diag(gamma)
    const float* c1;             // [576] row sums of w1f
    const unsigned short* d1;    // [576] bf16 folded bias
    const unsigned short* w2;    // [144, 576] bf16
    const unsigned short* b2;    // [144] bf16
    unsigned short* y;           // [M, 144] bf16
    long M;
    float eps;
};

__global__ __launch_bounds__(512) void hiera_mlp144_kernel(HmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long tok0 = (long)blockIdx.x * HM_TOK;
    const long tok = tok0 + wave * 32 + r;
    const long tokc = tok < p.M ? tok : p.M - 1;

    // ---- this wave's 32 tokens as B operands: lane (r, h) holds x[tok][16 ks + 8 h .. + 8]
    bf16x8 xf[HM_KS];
    {
        const unsigned short* xr = p.x + tokc * HM_C + 8 * h;
#pragma unroll
        for (int ks = 0; ks < HM_KS; ++ks) xf[ks] = *(const bf16x8*)(xr + 16 * ks);
    }
    // ---- LayerNorm statistics of the token (two passes over the register-resident half row, halves exchanged)
    float mean, rinv;
    {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < HM_KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)xf[ks][e];
        s += __shfl_xor(s, 32, 64);
        mean = s * (1.0f / HM_C);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < HM_KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)xf[ks][e] - mean;
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        rinv = __builtin_amdgcn_rsqf(q * (1.0f / HM_C) + p.eps);
    }

    // ---- weight chunk staging: HBM / L2 -> registers -> LDS
    u32x4 w1r[3], w2r[3];
    float cr = 0.f;
    auto load_chunk = [&](int ch) {
        const int n0 = ch * HM_HC;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + j * 512;
            if (idx < HM_HC * 18) {            // W1' rows n0 .. n0 + 63, 18 chunks of 16 B each
                const int row = idx / 18, c16 = idx % 18;
                w1r[j] = *(const u32x4*)(p.w1f + (long)(n0 + row) * HM_C + c16 * 8);
            }
            if (idx < HM_C * 8) {              // W2 rows 0 .. 143, columns n0 .. n0 + 63: 8 chunks of 16 B each
                const int row = idx >> 3, c16 = idx & 7;
                w2r[j] = *(const u32x4*)(p.w2 + (long)row * HM_H + n0 + c16 * 8);
            }
        }
        if (tid < HM_HC) cr = p.c1[n0 + tid];
        else if (tid < 2 * HM_HC) cr = bf2f(p.d1[n0 + tid - HM_HC]);
    };
    auto store_chunk = [&](int buf) {
        char* base = smem + buf * HM_STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + j * 512;
            if (idx < HM_HC * 18) *(u32x4*)(base + (idx / 18) * HM_W1STR + (idx % 18) * 16) = w1r[j];
            if (idx < HM_C * 8) {
                // a 16-byte chunk of a 136-byte row is only 8-byte aligned: two 8-byte stores
                char* d = base + HM_W1B + (idx >> 3) * HM_W2STR + (idx & 7) * 16;
                *(u32x2*)d = u32x2{w2r[j][0], w2r[j][1]};
                *(u32x2*)(d + 8) = u32x2{w2r[j][2], w2r[j][3]};
            }
        }
        if (tid < 2 * HM_HC) *(float*)(base + HM_W1B + HM_W2B + tid * 4) = cr;
    };

    f32x16 y[HM_CB];
#pragma unroll
    for (int cb = 0; cb < HM_CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[cb][i] = 0.f;

    // rows 144 .. 159 of the W2 image feed output rows that are never stored: give them zeros once (both buffers) so no NaN pattern wanders through the MFMAs
    for (int i = tid; i < 2 * 16 * HM_W2STR / 8; i += 512) {
        const int buf = i / (16 * HM_W2STR / 8), o = i % (16 * HM_W2STR / 8);
        *(u32x2*)(smem + buf * HM_STAGE + HM_W1B + HM_C * HM_W2STR + o * 8) = u32x2{0u, 0u};
    }
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    constexpr int NCH = HM_H / HM_HC;
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < NCH) load_chunk(ch + 1);
        const char* w1s = smem + buf * HM_STAGE;
        const char* w2s = w1s + HM_W1B;
        const float* cs = (const float*)(w2s + HM_W2B);      // [64] c, then [64] d
        // ---- H^T chunk = W1' X^T for both 32-hidden blocks first (18 MFMAs back to back), then per block: LayerNorm fold + GELU + bf16 (vector ALU) followed by its
        //      share of Y^T += W2 chunk . G^T (10 MFMAs) -- block 1's GELU has no dependence on block 0's MFMAs, so the two pipes can overlap
        f32x16 acc[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[hb][i] = 0.f;
            const char* a0 = w1s + (hb * 32 + r) * HM_W1STR + h * 16;
#pragma unroll
            for (int ks = 0; ks < HM_KS; ++ks) acc[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(a0 + ks * 32), xf[ks], acc[hb], 0, 0, 0);
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            bf16x8 gb[2];
            float g[16];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 cc = *(const f32x4*)(cs + hb * 32 + 8 * i4 + 4 * h);
                const f32x4 dd = *(const f32x4*)(cs + HM_HC + hb * 32 + 8 * i4 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = rinv * (acc[hb][4 * i4 + e] - mean * cc[e]) + dd[e];
                    g[4 * i4 + e] = hm_gelu_erf(bf2f(f2bf(v)));      // bf16 rounding of the linear output before the activation, as the unfused pair does
                }
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                u32x4 pk;
                pk[0] = pack_bf2(g[8 * ss + 0], g[8 * ss + 1]);
                pk[1] = pack_bf2(g[8 * ss + 2], g[8 * ss + 3]);
                pk[2] = pack_bf2(g[8 * ss + 4], g[8 * ss + 5]);
                pk[3] = pack_bf2(g[8 * ss + 6], g[8 * ss + 7]);
                gb[ss] = __builtin_bit_cast(bf16x8, pk);
            }
            // Y^T += W2 chunk . G^T: fragment element j of half h is hidden 16 ss + 8 (j >> 2) + 4 h + (j & 3) of block hb
#pragma unroll
            for (int cb = 0; cb < HM_CB; ++cb) {
                const char* a0 = w2s + (cb * 32 + r) * HM_W2STR + 8 * h + hb * 64;
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const char* a = a0 + 32 * ss;
                    const u32x2 lo = *(const u32x2*)a, hi = *(const u32x2*)(a + 16);
                    const u32x4 af = {lo[0], lo[1], hi[0], hi[1]};
                    y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), gb[ss], y[cb], 0, 0, 0);
                }
            }
        }
        if (ch + 1 < NCH) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: + b2 + x, one bf16 rounding; the tile is transposed through LDS (now free) so that whole rows leave
    char* ot = smem;
#pragma unroll
    for (int cb = 0; cb < HM_CB; ++cb)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int c = cb * 32 + 8 * i4 + 4 * h;
            if (c < HM_C) {
                const u32x2 xb = *(const u32x2*)(p.x + tokc * HM_C + c);
                const u32x2 bb = *(const u32x2*)(p.b2 + c);
                const float v0 = y[cb][4 * i4 + 0] + __uint_as_float(bb[0] << 16) + __uint_as_float(xb[0] << 16);
                const float v1 = y[cb][4 * i4 + 1] + __uint_as_float(bb[0] & 0xffff0000u) + __uint_as_float(xb[0] & 0xffff0000u);
                const float v2 = y[cb][4 * i4 + 2] + __uint_as_float(bb[1] << 16) + __uint_as_float(xb[1] << 16);
                const float v3 = y[cb][4 * i4 + 3] + __uint_as_float(bb[1] & 0xffff0000u) + __uint_as_float(xb[1] & 0xffff0000u);
                *(u32x2*)(ot + (wave * 32 + r) * HM_OSTR + c * 2) = u32x2{pack_bf2(v0, v1), pack_bf2(v2, v3)};
            }
        }
    __syncthreads();
    for (int idx = tid; idx < HM_TOK * 18; idx += 512) {
        const int row = idx / 18, c16 = idx % 18;
        if (tok0 + row < p.M) *(u32x4*)(p.y + (tok0 + row) * HM_C + c16 * 8) = *(const u32x4*)(ot + row * HM_OSTR + c16 * 16);
    }
}

}  // namespace rga3

using namespace rga3;

// y [M, 144] = x + W2 gelu(LayerNorm(x; eps) folded into W1f / c1 / d1) + b2   (Hiera stage-1 MLP, dims 144 -> 576 -> 144), all bf16 except c1 (f32).
// w1f / c1 / d1 as rga3_gemm_ln_bf16 takes them: W1 diag(gamma) rounded to bf16, its row sums in f32, beta W1^T + b1 in bf16.  x, y contiguous, 16-byte aligned.
extern "C" int rga3_hiera_mlp144(const void* x, const void* w1f, const float* c1, const void* d1, const void* w2, const void* b2, void* y, int64_t M, float eps,
                                 void* stream) {
    RGA3_CHECK_ARG(x && w1f && c1 && d1 && w2 && b2 && y && M > 0, "hiera_mlp144: null pointer / M %ld", (long)M);
    RGA3_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1f | (uintptr_t)w2 | (uintptr_t)y | (uintptr_t)c1) & 15) == 0 && (((uintptr_t)b2 | (uintptr_t)d1) & 7) == 0, "hiera_mlp144: alignment");
    RGA3_CHECK_ARG(x != y, "hiera_mlp144: in place is not supported (the residual is re-read)");
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)hiera_mlp144_kernel, HM_LDS, lds_grant, "hiera_mlp144")) return rc;
    HmArgs a;
    a.x = (const unsigned short*)x; a.w1f = (const unsigned short*)w1f; a.c1 = c1; a.d1 = (const unsigned short*)d1;
    a.w2 = (const unsigned short*)w2; a.b2 = (const unsigned short*)b2; a.y = (unsigned short*)y; a.M = M; a.eps = eps;
    hipLaunchKernelGGL(hiera_mlp144_kernel, dim3((unsigned)cdiv(M, HM_TOK)), dim3(512), HM_LDS, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("hiera_mlp144_kernel");
    return 0;
}
