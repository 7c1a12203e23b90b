// Token-side tail of the SAM2 mask decoder at inference (reference model/sam2.py:2129-2160 MaskDecoder.predict_masks: output_hypernetworks_mlps x 4, iou_prediction_head,
// pred_obj_score_head; :3396-3421 _forward_sam_heads: argmax over the multimask IoUs, token selection, obj_ptr_proj, object gating).
//
// Every one of these is a 3-layer MLP on ONE 256-wide row per frame.  As GEMV launches they were 21 launches + ~25 elementwise / index launches of the host framework
// per frame, each at the ~4.5 us launch floor: 0.2 ms of the 2.2 ms steady-state frame of the configs[3] stream (profiles/r03_stream_frame_timeline_memattn.txt).
// Two launches replace them:
//   mlp3_rows_kernel   grid (MLP, frame): the 3 layers of one MLP for one frame's token row; the row lives in LDS as bf16 and is the B operand of 16x16x32 MFMAs over 16
//                      weight rows at a time (see mlp_layer; the six weight sets together are 1.5 MB: L2-resident), bias + bf16 rounding + ReLU as the GEMM
//                      epilogue does, optional sigmoid on the last layer (IoU head);
//   sam_select_kernel  grid (frame): argmax over the 3 multimask IoUs (first maximum, on the bf16 values the IoU head wrote), the chosen mask token through obj_ptr_proj,
//                      hard gating against no_obj_ptr by the sign of the object score, and the plane index the bilinear / mask selection kernels take.
// Latency-bound byte work (nothing here belongs on MFMA).
#include "common.h"

namespace rga3 {

constexpr int MLP3_MAX = 8;      // MLPs per launch
constexpr int MLP3_MAXDIM = 512;

struct Mlp3 {
    const unsigned short* x;     // input rows: frame b reads x + b * x_bstride
    const unsigned short *w0, *b0, *w1, *b1, *w2, *b2;
    unsigned short* y;           // output rows: frame b writes y + b * y_bstride
    long x_bstride, y_bstride;
    int in, hid, out, final_act; // final_act: 0 none, 1 sigmoid (fp32 sigmoid of the bf16-rounded output, rounded to bf16 again: torch.sigmoid(x.float()).to(bf16))
};
struct Mlp3Batch { Mlp3 m[MLP3_MAX]; };

__device__ __forceinline__ float bf16_round(float v) { return bf2f(f2bf(v)); }

// yout[j] = act(bf16(sum_i W[j][i] xin[i] + bias[j])) for j < n_out; xin / yout bf16 rows in LDS (a layer's output IS bf16: nothing is lost between layers).
// The row is the B operand of 16x16x32 MFMAs (every column of the tile carries the same row; column 0 is kept), 16 weight rows the A operand, read straight from L2:
// a wave instruction touches 16 rows x 64 contiguous bytes and a wave keeps all 8 k-steps of a 16-row tile in flight.  History: a lane per output column walking
// its own weight row (64 cache lines per load instruction, one load in flight: ~5 us per 256 x 256 layer); rows read by 32 lanes and reduced by shuffles (five
// dependent ds_bpermute per row: 12 us per layer).
__device__ __forceinline__ void mlp_layer(const unsigned short* xin, int n_in, const unsigned short* W, const unsigned short* bias, int n_out, unsigned short* yout, bool relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nks = (n_in + 31) >> 5;                 // <= 16 (n_in <= 512)
    for (int n0 = wave * 16; n0 < n_out; n0 += nwave * 16) {
        const unsigned short* wrow = W + (long)min(n0 + c, n_out - 1) * n_in + 8 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // the 4 biases of this lane's outputs travel with the weight fragments (fetched after the products, one 2-byte load at a time, they were 16 dependent L2
        // round trips per layer -- most of the kernel)
        u32x2 bb = {0u, 0u};
        if (bias) {
            const int b0 = n0 + 4 * g;
            if (b0 + 3 < n_out && ((((uintptr_t)(bias + b0)) & 7) == 0)) {
                bb = *(const u32x2*)(bias + b0);
            } else {
                unsigned short t4[4] = {0, 0, 0, 0};
                for (int r = 0; r < 4; ++r) if (b0 + r < n_out) t4[r] = bias[b0 + r];
                bb = u32x2{(unsigned)t4[0] | ((unsigned)t4[1] << 16), (unsigned)t4[2] | ((unsigned)t4[3] << 16)};
            }
        }
        for (int ks0 = 0; ks0 < nks; ks0 += 8) {
            bf16x8 af[8], bf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = (ks0 + u) * 32 + 8 * g;
                u32x4 za = {0u, 0u, 0u, 0u}, zb = {0u, 0u, 0u, 0u};
                if (k < n_in) {          // n_in % 8 == 0: a chunk lies inside the row or outside
                    za = *(const u32x4*)(wrow + (ks0 + u) * 32);
                    zb = *(const u32x4*)(xin + k);
                }
                af[u] = __builtin_bit_cast(bf16x8, za);
                bf[u] = __builtin_bit_cast(bf16x8, zb);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u], bf[u], acc, 0, 0, 0);
        }
        if (c == 0) {                    // lane (0, g) holds outputs n0 + 4 g .. + 3
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = n0 + 4 * g + r;
                if (row < n_out) {
                    float v = bf16_round(acc[r] + __uint_as_float((r & 1) ? (bb[r >> 1] & 0xffff0000u) : (bb[r >> 1] << 16)));
                    if (relu) v = fmaxf(v, 0.f);
                    yout[row] = f2bf(v);
                }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void mlp3_rows_kernel(Mlp3Batch P) {
    __shared__ __attribute__((aligned(16))) unsigned short bufa[MLP3_MAXDIM], bufb[MLP3_MAXDIM];
    const Mlp3& m = P.m[blockIdx.x];
    const long b = blockIdx.y;
    const unsigned short* x = m.x + b * m.x_bstride;
    for (int i = threadIdx.x; i < m.in; i += blockDim.x) bufa[i] = x[i];
    __syncthreads();
    mlp_layer(bufa, m.in, m.w0, m.b0, m.hid, bufb, true);
    mlp_layer(bufb, m.hid, m.w1, m.b1, m.hid, bufa, true);
    mlp_layer(bufa, m.hid, m.w2, m.b2, m.out, bufb, false);
    unsigned short* y = m.y + b * m.y_bstride;
    for (int j = threadIdx.x; j < m.out; j += blockDim.x) {
        float v = bf2f(bufb[j]);
        if (m.final_act == 1) v = 1.f / (1.f + __expf(-v));
        y[j] = f2bf(v);
    }
}

struct SamSelectArgs {
    const unsigned short* iou;     // [B, 4] bf16 (sigmoid applied)
    const unsigned short* obj;     // [B, 1] bf16 object score logits
    const unsigned short* toks;    // mask tokens [B, 4, C], frame stride tok_bstride elements (a view of the decoder's output tokens)
    const unsigned short *w0, *b0, *w1, *b1, *w2, *b2;   // obj_ptr_proj
    const unsigned short* no_obj_ptr;                     // [C]
    long long* best;               // [2, B] int64: row 0 = argmax over IoU 1..3, row 1 = plane index b * 4 + 1 + best (for index ops that want int64)
    int* sel;                      // [B] int32: plane index b * 4 + 1 + best
    unsigned short* obj_ptr;       // [B, C]
    long tok_bstride;
    int C;
};

__global__ __launch_bounds__(1024) void sam_select_kernel(SamSelectArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short bufa[MLP3_MAXDIM], bufb[MLP3_MAXDIM];
    const long b = blockIdx.x;
    // first maximum of the three multimask IoUs (torch.argmax returns the first index on ties)
    int best = 0;
    float bv = bf2f(p.iou[b * 4 + 1]);
    for (int m = 1; m < 3; ++m) {
        const float v = bf2f(p.iou[b * 4 + 1 + m]);
        if (v > bv) { bv = v; best = m; }
    }
    if (threadIdx.x == 0) {
        p.best[b] = best;
        p.best[gridDim.x + b] = b * 4 + 1 + best;
        p.sel[b] = (int)(b * 4 + 1 + best);
    }
    const unsigned short* tok = p.toks + b * p.tok_bstride + (long)(1 + best) * p.C;      // multimask token 1 + best
    for (int i = threadIdx.x; i < p.C; i += blockDim.x) bufa[i] = tok[i];
    __syncthreads();
    mlp_layer(bufa, p.C, p.w0, p.b0, p.C, bufb, true);
    mlp_layer(bufb, p.C, p.w1, p.b1, p.C, bufa, true);
    mlp_layer(bufa, p.C, p.w2, p.b2, p.C, bufb, false);
    const bool is_obj = bf2f(p.obj[b]) > 0.f;
    for (int j = threadIdx.x; j < p.C; j += blockDim.x) p.obj_ptr[b * p.C + j] = is_obj ? bufb[j] : p.no_obj_ptr[j];
}

}  // namespace rga3

using namespace rga3;

// n (<= 8) three-layer MLPs (Linear + ReLU, Linear + ReLU, Linear [+ sigmoid]) on one row per frame, B frames, one launch.
// ptrs: n x 8 device pointers {x, w0, b0, w1, b1, w2, b2, y}; dims: n x 6 {x frame stride, y frame stride (elements), in, hidden, out, final_act}.  HOST arrays.
extern "C" int rga3_mlp3_rows(const void* const* ptrs, const int64_t* dims, int n, int64_t B, void* stream) {
    RGA3_CHECK_ARG(ptrs && dims && n >= 1 && n <= MLP3_MAX && B >= 1 && B <= 65535, "mlp3_rows: n %d (1..8), B %ld", n, (long)B);
    Mlp3Batch P;
    for (int i = 0; i < n; ++i) {
        Mlp3& m = P.m[i];
        const void* const* q = ptrs + 8 * i;
        const int64_t* d = dims + 6 * i;
        m.x = (const unsigned short*)q[0];
        m.w0 = (const unsigned short*)q[1]; m.b0 = (const unsigned short*)q[2];
        m.w1 = (const unsigned short*)q[3]; m.b1 = (const unsigned short*)q[4];
        m.w2 = (const unsigned short*)q[5]; m.b2 = (const unsigned short*)q[6];
        m.y = (unsigned short*)q[7];
        m.x_bstride = d[0]; m.y_bstride = d[1];
        m.in = (int)d[2]; m.hid = (int)d[3]; m.out = (int)d[4]; m.final_act = (int)d[5];
        RGA3_CHECK_ARG(m.x && m.w0 && m.w1 && m.w2 && m.y, "mlp3_rows: null pointer in MLP %d", i);
        RGA3_CHECK_ARG(m.in > 0 && m.hid > 0 && m.out > 0 && m.in % 8 == 0 && m.hid % 8 == 0 && m.in <= MLP3_MAXDIM && m.hid <= MLP3_MAXDIM && m.out <= MLP3_MAXDIM,
                       "mlp3_rows: dims %d -> %d -> %d (inputs multiples of 8, all <= 512)", m.in, m.hid, m.out);
        RGA3_CHECK_ARG((((uintptr_t)m.w0 | (uintptr_t)m.w1 | (uintptr_t)m.w2) & 15) == 0, "mlp3_rows: weights must be 16-byte aligned");
    }
    // 16 waves: a 256-wide layer is one 16-row tile per wave, so a layer costs one L2 round trip instead of four in a row
    hipLaunchKernelGGL(mlp3_rows_kernel, dim3((unsigned)n, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, P);
    RGA3_CHECK_LAUNCH("mlp3_rows_kernel");
    return 0;
}

extern "C" int rga3_sam_select_objptr(const void* iou, const void* obj, const void* toks, int64_t tok_bstride, int C, const void* w0, const void* b0, const void* w1, const void* b1,
                                      const void* w2, const void* b2, const void* no_obj_ptr, int64_t* best, int32_t* sel, void* obj_ptr, int64_t B, void* stream) {
    RGA3_CHECK_ARG(iou && obj && toks && w0 && w1 && w2 && no_obj_ptr && best && sel && obj_ptr, "sam_select_objptr: null pointer");
    RGA3_CHECK_ARG(B >= 1 && B <= 65535 && tok_bstride >= 4L * C && C > 0 && C % 8 == 0 && C <= MLP3_MAXDIM, "sam_select_objptr: B %ld C %d", (long)B, C);
    RGA3_CHECK_ARG((((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0, "sam_select_objptr: weights must be 16-byte aligned");
    SamSelectArgs a;
    a.iou = (const unsigned short*)iou; a.obj = (const unsigned short*)obj; a.toks = (const unsigned short*)toks;
    a.w0 = (const unsigned short*)w0; a.b0 = (const unsigned short*)b0; a.w1 = (const unsigned short*)w1; a.b1 = (const unsigned short*)b1;
    a.w2 = (const unsigned short*)w2; a.b2 = (const unsigned short*)b2; a.no_obj_ptr = (const unsigned short*)no_obj_ptr;
    a.best = (long long*)best; a.sel = sel; a.obj_ptr = (unsigned short*)obj_ptr; a.tok_bstride = tok_bstride; a.C = C;
    hipLaunchKernelGGL(sam_select_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("sam_select_kernel");
    return 0;
}
