// SAM-side input pipeline on the GPU: the reference resizes every frame on the CPU with Pillow
// (DirectResize.apply_image, reference utils/utils.py:246-256: Image.resize -> antialiased bicubic, uint8 fixed point),
// normalises in fp32 (preprocess, utils/utils.py:230-243) and casts to bf16 (evaluation/mevis_val_u/inference_mevis.py:178-180);
// that makes images_sam [T,3,1024,1024], the largest tensor handed to the hot path.
//
// Bit-exact restatement of Pillow's two-pass 8-bit resample (src/libImaging/Resample.c): coefficient tables are computed on
// the HOST in double precision with Pillow's operation order (rga3_pil_bicubic_coeffs), the passes run on the device in int32:
//   horizontal: tmp[t][y][xo][c] = clip8((sum_k src[t][y][xmin+k][c] * kk[xo][k] + 2^21) >> 22)
//   vertical  : u8 = clip8((sum_k tmp[t][ymin+k][xo][c] * kk[yo][k] + 2^21) >> 22);  out = bf16((u8 - mean[c]) / std[c])
// Both are HBM-bound byte kernels: the vertical pass writes 6 B per output pixel (bf16 CHW planes) and reads ksize x 3 B that
// neighbouring threads share through L2; nothing here belongs on MFMA.
#include "common.h"

namespace rga3 {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;  // arithmetic shift, as Pillow's lookup index
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct ResampleArgs {
    const unsigned char* src;
    unsigned char* dst_u8;
    unsigned short* dst_bf16;
    const int* bounds;
    const int* kk;
    int ksize;
    long T;
    int n_in_rows, in_w, out_h, out_w;  // horizontal: rows = n_in_rows, in_w -> out_w; vertical: n_in_rows -> out_h at width out_w
    float mean[3], stdv[3];
};

// thread = (t, y, xo): 3 channels of one output pixel of a row
__global__ __launch_bounds__(256) void resample_h_kernel(ResampleArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = p.T * p.n_in_rows * p.out_w;
    if (idx >= total) return;
    const int xo = (int)(idx % p.out_w);
    const long row = idx / p.out_w;  // t * rows + y
    const int xmin = p.bounds[2 * xo], n = p.bounds[2 * xo + 1];
    const int* k = p.kk + (long)xo * p.ksize;
    const unsigned char* s = p.src + (row * p.in_w + xmin) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        a0 += (int)s[3 * i] * w;
        a1 += (int)s[3 * i + 1] * w;
        a2 += (int)s[3 * i + 2] * w;
    }
    unsigned char* d = p.dst_u8 + idx * 3;
    d[0] = (unsigned char)clip8(a0);
    d[1] = (unsigned char)clip8(a1);
    d[2] = (unsigned char)clip8(a2);
}

// thread = (t, yo, 4 consecutive xo): 12 source bytes per tap (3 dwords), 8-byte bf16 stores per channel plane
__global__ __launch_bounds__(256) void resample_v_kernel(ResampleArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int w4 = p.out_w / 4;
    const long plane = (long)p.out_h * p.out_w;
    const long total = p.T * (long)p.out_h * w4;
    if (idx >= total) return;
    const int xo = (int)(idx % w4) * 4;
    const int yo = (int)((idx / w4) % p.out_h);
    const long t = idx / ((long)w4 * p.out_h);
    const int ymin = p.bounds[2 * yo], n = p.bounds[2 * yo + 1];
    const int* k = p.kk + (long)yo * p.ksize;
    const unsigned char* s = p.src + ((t * p.n_in_rows + ymin) * p.out_w + xo) * 3;
    const long rstride = (long)p.out_w * 3;
    int acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 1 << (PRECISION_BITS - 1);
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        const unsigned* r = (const unsigned*)(s + i * rstride);
        const unsigned d0 = r[0], d1 = r[1], d2 = r[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] += (int)((d0 >> (8 * e)) & 255u) * w;
            acc[4 + e] += (int)((d1 >> (8 * e)) & 255u) * w;
            acc[8 + e] += (int)((d2 >> (8 * e)) & 255u) * w;
        }
    }
    int v[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) v[e] = clip8(acc[e]);   // byte e = pixel e / 3, channel e % 3
    if (p.dst_u8) {
        unsigned* d = (unsigned*)(p.dst_u8 + ((t * p.out_h + yo) * (long)p.out_w + xo) * 3);
        d[0] = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
        d[1] = (unsigned)v[4] | ((unsigned)v[5] << 8) | ((unsigned)v[6] << 16) | ((unsigned)v[7] << 24);
        d[2] = (unsigned)v[8] | ((unsigned)v[9] << 8) | ((unsigned)v[10] << 16) | ((unsigned)v[11] << 24);
    }
    if (p.dst_bf16) {
        unsigned short* o = p.dst_bf16 + t * 3 * plane + (long)yo * p.out_w + xo;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // fp32 subtract, IEEE divide, round to nearest even: the reference's (x - mean) / std then .bfloat16()
            u32x2 pk;
            pk[0] = pack_bf2(((float)v[c] - p.mean[c]) / p.stdv[c], ((float)v[3 + c] - p.mean[c]) / p.stdv[c]);
            pk[1] = pack_bf2(((float)v[6 + c] - p.mean[c]) / p.stdv[c], ((float)v[9 + c] - p.mean[c]) / p.stdv[c]);
            *(u32x2*)(o + c * plane) = pk;
        }
    }
}


// Qwen-side tail: resized uint8 frames [T, h, w, 3] -> pixel_values_videos [gt*gh*gw, 3*tp*ps*ps], the HF video
// processor's rescale + normalise + patchify (installed transformers models/qwen2_vl/video_processing_qwen2_vl.py:236-274:
// view (gt, tp, C, gh/m, m, ps, gw/m, m, ps) -> permute (gt, gh/m, gw/m, m, m, C, tp, ps, ps)).  A normalised value depends only on
// (channel, byte): the host builds the 3x256 fp32 table in the reference's own operation order (rga3_qwen_norm_lut) and the kernel
// is a byte gather + table lookup + coalesced store; a frame count not divisible by tp repeats the last frame (:242-246).
struct PatchifyArgs {
    const unsigned char* src;
    void* dst;
    const float* lut;
    long T, rows;
    int h, w, gh, gw, ps, tp, m, out_f32;
};

// one wave per output row, lanes walk the row in element pairs (ps is even, so a pair never crosses a patch line)
__global__ __launch_bounds__(256) void qwen_patchify_kernel(PatchifyArgs p) {
    __shared__ float lut[768];
    for (int i = threadIdx.x; i < 768; i += 256) lut[i] = p.lut[i];
    __syncthreads();
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int lane = threadIdx.x & 63;
    const int m2 = p.m * p.m;
    const int mi = (int)(row % m2);
    const long blk = row / m2;
    const int bw = (int)(blk % (p.gw / p.m));
    const int bh = (int)((blk / (p.gw / p.m)) % (p.gh / p.m));
    const long gt = blk / ((long)(p.gw / p.m) * (p.gh / p.m));
    const int y0 = (bh * p.m + mi / p.m) * p.ps, x0 = (bw * p.m + mi % p.m) * p.ps;
    const int pp = p.ps * p.ps, per_c = p.tp * pp, width = 3 * per_c;
    for (int e = lane * 2; e < width; e += 128) {
        const int c = e / per_c, r = e % per_c;
        const int tpi = r / pp, py = (r % pp) / p.ps, px = r % p.ps;
        long f = gt * p.tp + tpi;
        if (f > p.T - 1) f = p.T - 1;
        const unsigned char* s = p.src + ((f * p.h + y0 + py) * (long)p.w + x0 + px) * 3 + c;
        const float v0 = lut[c * 256 + s[0]], v1 = lut[c * 256 + s[3]];
        if (p.out_f32) {
            float* o = (float*)p.dst + row * width + e;
            o[0] = v0;
            o[1] = v1;
        } else {
            *(uint32_t*)((unsigned short*)p.dst + row * width + e) = pack_bf2(v0, v1);
        }
    }
}

}  // namespace rga3

using namespace rga3;

// frames u8 [T, H, W, 3] -> resized u8 [T, out_h, out_w, 3] (dst_u8, optional) and / or normalised bf16 [T, 3, out_h, out_w]
// (dst_bf16, optional).  bh/kh (horizontal, out_w rows) and bv/kv (vertical, out_h rows) are DEVICE copies of the tables of
// rga3_pil_bicubic_coeffs; tmp is a device workspace of T*H*out_w*3 bytes (unused when W == out_w).  mean3 / std3: host floats.
extern "C" int rga3_sam_preprocess_u8(const void* frames, int64_t T, int H, int W, int out_h, int out_w, const int32_t* bh,
                                      const int32_t* kh, int ksize_h, const int32_t* bv, const int32_t* kv, int ksize_v, void* tmp,
                                      void* dst_u8, void* dst_bf16, const float* mean3, const float* std3, void* stream) {
    RGA3_CHECK_ARG(frames && T > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0, "preprocess: sizes");
    RGA3_CHECK_ARG(dst_u8 || dst_bf16, "preprocess: no output");
    RGA3_CHECK_ARG(!dst_bf16 || (mean3 && std3), "preprocess: mean/std");
    RGA3_CHECK_ARG(W == out_w || (bh && kh && tmp && ksize_h > 0), "preprocess: horizontal tables / workspace");
    RGA3_CHECK_ARG(H == out_h || (bv && kv && ksize_v > 0), "preprocess: vertical tables");
    RGA3_CHECK_ARG(H != out_h || W != out_w, "preprocess: same-size input is a copy + normalise: use the row kernels");
    hipStream_t st = (hipStream_t)stream;
    ResampleArgs a;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3 ? mean3[c] : 0.f; a.stdv[c] = std3 ? std3[c] : 1.f; }
    a.T = T;
    const unsigned char* mid = (const unsigned char*)frames;
    if (W != out_w) {
        const bool last = (H == out_h);
        RGA3_CHECK_ARG(!last || (dst_u8 && !dst_bf16), "preprocess: horizontal-only resize writes u8");
        a.src = (const unsigned char*)frames; a.dst_u8 = last ? (unsigned char*)dst_u8 : (unsigned char*)tmp; a.dst_bf16 = nullptr;
        a.bounds = bh; a.kk = kh; a.ksize = ksize_h; a.n_in_rows = H; a.in_w = W; a.out_h = H; a.out_w = out_w;
        const long total = T * (long)H * out_w;
        hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, a);
        RGA3_CHECK_LAUNCH("resample_h_kernel");
        mid = (const unsigned char*)tmp;
        if (last) return 0;
    }
    a.src = mid; a.dst_u8 = (unsigned char*)dst_u8; a.dst_bf16 = (unsigned short*)dst_bf16;
    a.bounds = bv; a.kk = kv; a.ksize = ksize_v; a.n_in_rows = H; a.in_w = out_w; a.out_h = out_h; a.out_w = out_w;
    RGA3_CHECK_ARG(out_w % 4 == 0, "preprocess: out_w %d must be a multiple of 4", out_w);
    const long total = T * (long)out_h * (out_w / 4);
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, a);
    RGA3_CHECK_LAUNCH("resample_v_kernel");
    return 0;
}

// frames u8 [T, h, w, 3] (device; h, w multiples of patch*merge) -> out [ceil(T/tpatch) * (h/patch) * (w/patch), 3*tpatch*patch*patch]
// bf16 (out_dtype 0) or fp32 (1); lut768 = DEVICE copy of rga3_qwen_norm_lut's table.
extern "C" int rga3_qwen_patchify_u8(const void* frames, int64_t T, int h, int w, const float* lut768, void* out, int out_dtype, int patch,
                                     int tpatch, int merge, void* stream) {
    RGA3_CHECK_ARG(frames && lut768 && out && T > 0, "patchify: null argument");
    RGA3_CHECK_ARG(patch > 0 && patch % 2 == 0 && tpatch > 0 && merge > 0, "patchify: patch %d (even), tpatch %d, merge %d", patch, tpatch, merge);
    RGA3_CHECK_ARG(h > 0 && w > 0 && h % (patch * merge) == 0 && w % (patch * merge) == 0, "patchify: %dx%d not a multiple of %d", h, w, patch * merge);
    RGA3_CHECK_ARG(out_dtype == 0 || out_dtype == 1, "patchify: out_dtype %d", out_dtype);
    PatchifyArgs a;
    a.src = (const unsigned char*)frames; a.dst = out; a.lut = lut768; a.T = T; a.h = h; a.w = w; a.gh = h / patch; a.gw = w / patch;
    a.ps = patch; a.tp = tpatch; a.m = merge; a.out_f32 = out_dtype;
    a.rows = cdiv(T, (int64_t)tpatch) * a.gh * a.gw;
    hipLaunchKernelGGL(qwen_patchify_kernel, dim3((unsigned)cdiv(a.rows, 4)), dim3(256), 0, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("qwen_patchify_kernel");
    return 0;
}
