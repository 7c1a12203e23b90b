// Activation-stationary product for the short-K shapes of the frozen Hiera trunk (reference model/sam2.py:986-1117 MultiScaleBlock: attn.qkv, attn.proj, mlp fc1 at
// K = 576 over 65 536 tokens per 16 frames; 36 stage-3 blocks).
//
// The LDS-tiled kernels (gemm_bf16.hip) stage BOTH operands per output tile: a 128 x 192 tile at K = 576 pulls 368 KB into the CU for 28 MFLOP and pays a prologue
// and an epilogue for nine K-tiles -- these products run at 0.5 - 0.8 PF where the long-K ones reach 1.1 - 1.4 (profiles/r03_train_gemm_shapes.txt).  Here the
// ACTIVATIONS stay: a workgroup takes 256 token rows (8 waves x 32), every wave keeps its 32 rows x K as MFMA B-operand fragments in registers for the whole kernel
// (K = 576: 144 VGPRs), and the weight matrix streams past them, 32 output columns at a time, through a three-deep ring of LDS images (registers -> LDS staging, one
// barrier per block).  Per 32-column block a wave issues K / 16 back-to-back 32x32x16 MFMAs on ONE accumulator; there is no per-tile ramp, the only fixed cost is
// loading the 256 x K activation block once.  Per CU the stream is 32 x K x 2 B per 2 x (K / 16) x 32 MFMA cycles (33 GB/s at K = 576: well under what a CU takes
// from L2), the LDS array serves 8 waves x K / 16 ds_read_b128 per block = half of the matrix pipe's time.
// Tokens sit on the MFMA's N side, so a lane owns ONE token: LayerNorm-fold statistics (mean, 1 / sigma per row) are lane-local.  Results leave through a wave-private
// LDS tile (32 tokens x 64 columns) so that stores are whole 128-byte row segments; the residual is added on that path (16-byte coalesced reads), after the linear
// output has been rounded to bf16 -- gemm_epilogue's rounding points.
#include "common.h"

namespace rga3 {

enum { XS_ACT_NONE = 0, XS_ACT_GELU = 1, XS_ACT_RELU = 3 };   // the ids of gemm_bf16.hip's ACT_*

struct XsArgs {
    const unsigned short* A;      // [M, K] activations
    const unsigned short* W;      // [N, K] weights (nn.Linear layout)
    const unsigned short* bias;   // [N] or null
    const unsigned short* res;    // [M, N] or null
    unsigned short* C;            // [M, N]
    const float* rowstat;         // LayerNorm fold: [M][2] (mean, 1 / sqrt(var + eps)) of the rows of A, with colc [N] = row sums of the gamma-folded weight
    const float* colc;
    long lda, ldw, ldc, ldr;
    int M, N;
};

__device__ __forceinline__ float xs_gelu_erf(float x) {   // exact-erf GELU, erf by Abramowitz-Stegun 7.1.26: gemm_bf16.hip's gelu_erf, operation for operation
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

// KS = K / 16 k-steps (K = 576: 36)
template <int KS, int ACT, bool LNF>
__global__ __launch_bounds__(512) void gemm_xstat_kernel(XsArgs p) {
    constexpr int KB = KS * 32;               // bytes per weight row
    constexpr int STR = KB + 16;              // LDS row stride: an odd number of 16-byte slots -> the 16 lanes of a ds_read_b128 group hit 16 different slots
    constexpr int STAGE = 32 * STR;           // one 32-column weight block
    constexpr int CPR = KS * 2;               // 16-byte chunks per weight row
    constexpr int NLD = (32 * CPR + 511) / 512;   // chunks per thread and block
    constexpr int OSTR = 144;                 // output tile row stride (64 columns x 2 B + 16)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    char* otile = smem + 3 * STAGE + wave * (32 * OSTR);
    const long m0 = (long)blockIdx.x * 256 + wave * 32;
    const long tok = min(m0 + r, (long)p.M - 1);

    // ---- this wave's 32 tokens as B operands, for the whole kernel: lane (r, h) holds x[tok][16 ks + 8 h .. + 8]
    bf16x8 xf[KS];
    {
        const unsigned short* xrow = p.A + tok * p.lda + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(xrow + 16 * ks);
    }
    float ln_mean = 0.f, ln_rinv = 1.f;
    if constexpr (LNF) {
        const float2 st2 = *(const float2*)(p.rowstat + 2 * tok);
        ln_mean = st2.x;
        ln_rinv = st2.y;
    }

    // ---- weight blocks: HBM / L2 -> registers -> LDS
    u32x4 wreg[NLD];
    auto load_block = [&](int nb) {
        const unsigned short* wb = p.W + (long)nb * 32 * p.ldw;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int c = tid + j * 512;
            const int row = c / CPR, ch = c - row * CPR;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (c < 32 * CPR && nb * 32 + row < p.N) z = *(const u32x4*)(wb + (long)row * p.ldw + ch * 8);
            wreg[j] = z;
        }
    };
    auto store_block = [&](int buf) {
        char* dst = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int c = tid + j * 512;
            const int row = c / CPR, ch = c - row * CPR;
            if (c < 32 * CPR) *(u32x4*)(dst + row * STR + ch * 16) = wreg[j];
        }
    };
    const int nblk = (p.N + 31) >> 5;
    load_block(0);
    store_block(0);
    if (nblk > 1) {
        load_block(1);
        store_block(1);
    }
    // the activation fragments are due before the loop (otherwise the compiler counts their loads down inside it, behind each block's prefetch)
    asm volatile("" ::"v"(xf[0]), "v"(xf[KS / 2]), "v"(xf[KS - 1]));
    __syncthreads();

    for (int nb = 0; nb < nblk; ++nb) {
        if (nb + 2 < nblk) load_block(nb + 2);
        const char* wa = smem + (nb % 3) * STAGE + r * STR + h * 16;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wa + ks * 32), xf[ks], acc, 0, 0, 0);
        // ---- epilogue of this 32-column block: lane (r, h) holds token r, columns n0 + 8 i4 + 4 h + e in acc[4 i4 + e]
        const int n0 = nb * 32;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int col = n0 + 8 * i4 + 4 * h;
            u32x2 bb = {0u, 0u};
            if (p.bias && col < p.N) bb = *(const u32x2*)(p.bias + col);
            f32x4 cc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LNF) {
                if (col < p.N) cc = *(const f32x4*)(p.colc + col);
            }
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[4 * i4 + e];
                if constexpr (LNF) x = ln_rinv * (x - ln_mean * cc[e]);
                x += __uint_as_float((e & 1) ? (bb[e >> 1] & 0xffff0000u) : (bb[e >> 1] << 16));
                if constexpr (ACT == XS_ACT_GELU) x = xs_gelu_erf(bf2f(f2bf(x)));
                if constexpr (ACT == XS_ACT_RELU) x = fmaxf(x, 0.f);
                v[e] = x;
            }
            u32x2 pk;
            pk[0] = pack_bf2(v[0], v[1]);
            pk[1] = pack_bf2(v[2], v[3]);
            *(u32x2*)(otile + r * OSTR + ((nb & 1) * 32 + 8 * i4 + 4 * h) * 2) = pk;
        }
        // ---- every second block (or the last): 64 columns of 32 tokens leave as 128-byte row segments; residual added here
        if ((nb & 1) || nb + 1 == nblk) {
            const int c0 = (nb & ~1) * 32;                       // first column of the pair
            const int ncols = min(64, p.N - c0);                 // 32 or 64 (N % 32 == 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int trow = (lane >> 3) + 8 * j, ch = lane & 7;
                const long t = m0 + trow;
                if (t < p.M && ch * 8 < ncols) {
                    u32x4 val = *(const u32x4*)(otile + trow * OSTR + ch * 16);
                    if (p.res) {
                        const u32x4 rv = *(const u32x4*)(p.res + t * p.ldr + c0 + ch * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            val[e] = pack_bf2(__uint_as_float(val[e] << 16) + __uint_as_float(rv[e] << 16),
                                              __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rv[e] & 0xffff0000u));
                    }
                    *(u32x4*)(p.C + t * p.ldc + c0 + ch * 8) = val;
                }
            }
        }
        if (nb + 2 < nblk) store_block((nb + 2) % 3);
        __syncthreads();
    }
}

template <int KS, int ACT, bool LNF>
static int xstat_launch(const XsArgs& a, hipStream_t st) {
    constexpr int LDS = 3 * 32 * (KS * 32 + 16) + 8 * 32 * 144;
    auto kern = gemm_xstat_kernel<KS, ACT, LNF>;
    static bool attr_done = false;
    if (!attr_done && LDS > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail(-(int)e, "gemm_xstat: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(a.M, 256)), dim3(512), LDS, st, a);
    RGA3_CHECK_LAUNCH("gemm_xstat_kernel");
    return 0;
}

template <int KS>
static int xstat_dispatch(const XsArgs& a, int act, hipStream_t st) {
    if (a.rowstat) {
        if (act == XS_ACT_GELU) return xstat_launch<KS, XS_ACT_GELU, true>(a, st);
        if (act == XS_ACT_RELU) return xstat_launch<KS, XS_ACT_RELU, true>(a, st);
        return xstat_launch<KS, XS_ACT_NONE, true>(a, st);
    }
    if (act == XS_ACT_GELU) return xstat_launch<KS, XS_ACT_GELU, false>(a, st);
    if (act == XS_ACT_RELU) return xstat_launch<KS, XS_ACT_RELU, false>(a, st);
    return xstat_launch<KS, XS_ACT_NONE, false>(a, st);
}

}  // namespace rga3

using namespace rga3;

// C [M, N] bf16 = act(A W^T + bias) (+ residual), or with rowstat / colc given act(LayerNorm(A) W^T + b) in the folded form of rga3_gemm_ln_bf16, by the
// activation-stationary kernel: K = 576 (Hiera-L stage 3), N % 32 == 0, act none / gelu / relu.  Same epilogue arithmetic and rounding points as rga3_gemm_bf16.
extern "C" int rga3_gemm_xstat_bf16(const void* A, const void* W, const void* bias, const void* residual, const float* rowstat, const float* colc, void* C, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, int act, void* stream) {
    RGA3_CHECK_ARG(A && W && C && M > 0 && N > 0, "gemm_xstat: null pointer / empty shape");
    RGA3_CHECK_ARG(K == 576, "gemm_xstat: K = %ld (the activation-stationary kernel is built for K = 576)", (long)K);
    RGA3_CHECK_ARG(N % 32 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "gemm_xstat: N %% 32, strides %% 8");
    RGA3_CHECK_ARG(act == XS_ACT_NONE || act == XS_ACT_GELU || act == XS_ACT_RELU, "gemm_xstat: act %d", act);
    RGA3_CHECK_ARG((rowstat == nullptr) == (colc == nullptr), "gemm_xstat: rowstat and colc come together");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)residual | (uintptr_t)colc) & 15) == 0 && (((uintptr_t)bias | (uintptr_t)rowstat) & 7) == 0,
                   "gemm_xstat: pointer alignment");
    RGA3_CHECK_ARG(M < (1LL << 31) && N < (1 << 24), "gemm_xstat: shape");
    XsArgs a;
    a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.bias = (const unsigned short*)bias; a.res = (const unsigned short*)residual;
    a.C = (unsigned short*)C; a.rowstat = rowstat; a.colc = colc;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.M = (int)M; a.N = (int)N;
    return xstat_dispatch<36>(a, act, (hipStream_t)stream);
}
