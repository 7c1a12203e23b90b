// HBM-bound kernels specific to the SAM2 half of the path (reference model/sam2.py): patch im2col, windowed max-pool,
// FPN top-down add, broadcast adds, bilinear resampling (+ candidate select), small-channel strided conv, depthwise 7x7,
// axial complex RoPE, mask -> memory-encoder affine sigmoid, ConvTranspose pixel shuffle, BCE/dice partial sums.
// All feature maps are TOKEN-MAJOR ([pixels, C], i.e. NHWC): 1x1 convs / linears are plain GEMMs and every kernel
// below is coalesced along C.
#include "common.h"

namespace rga3 {

__device__ __forceinline__ void up8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pk8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- im2col for Conv2d(k=KS, stride=ST, pad=PD) on NCHW bf16 images -> rows [F*Ho*Wo, ldo], cols (c, kh, kw), zero tail.
//      Thread = one 16-byte chunk of an output row (8 consecutive columns): the (c, kh, kw) decode runs once per chunk and then steps, the store is one
//      16-byte write (the element-per-thread form wrote 2 bytes per lane with three divisions each: 585 us for 16 frames 1024^2 -> [1 M, 152]).
__global__ __launch_bounds__(256) void im2col_kernel(const unsigned short* __restrict__ img, unsigned short* __restrict__ out, int F, int C,
                                                     int H, int W, int KS, int ST, int PD, int Ho, int Wo, int ldo) {
    const int nch = ldo / 8;
    const long total = (long)F * Ho * Wo * nch;
    const int ncol = C * KS * KS;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long row = i / nch;
        const int x = (int)(row % Wo), y = (int)((row / Wo) % Ho);
        const long f = row / ((long)Wo * Ho);
        int col = ch * 8;
        int kw = col % KS, kh = (col / KS) % KS, c = col / (KS * KS);
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e, ++col) {
            unsigned short t = 0;
            if (col < ncol) {
                const int sy = y * ST - PD + kh, sx = x * ST - PD + kw;
                if (sy >= 0 && sy < H && sx >= 0 && sx < W) t = img[((f * C + c) * H + sy) * (long)W + sx];
            }
            v[e] = t;
            if (++kw == KS) { kw = 0; if (++kh == KS) { kh = 0; ++c; } }
        }
        u32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (unsigned)v[2 * e] | ((unsigned)v[2 * e + 1] << 16);
        *(u32x4*)(out + row * ldo + ch * 8) = pk;
    }
}

// ---- 2x2 max pool on window-major tokens: windows of w x w tokens -> (w/2) x (w/2), channels 8 at a time.
__global__ __launch_bounds__(256) void maxpool_win_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, long nwin, int w,
                                                          int C, long ldx, long ldy) {
    const int wo = w / 2, nch = C / 8;
    const long total = nwin * wo * wo * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long t = i / nch;
        const int c = (int)(t % wo), r = (int)((t / wo) % wo);
        const long win = t / ((long)wo * wo);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long tok = win * w * w + (2 * r + dy) * w + (2 * c + dx);
                float f[8];
                up8(*(const u32x4*)(x + tok * ldx + ch * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], f[e]);
            }
        *(u32x4*)(y + t * ldy + ch * 8) = pk8(m);
    }
}

// ---- out[f, y, x, :] = a[f, y, x, :] + b[f, y/2, x/2, :]   (FPN nearest x2 top-down, reference sam2.py:867-889)
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                             unsigned short* __restrict__ o, long F, int H, int W, int C) {
    const int nch = C / 8;
    const long total = F * H * W * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long t = i / nch;
        const int x = (int)(t % W), y = (int)((t / W) % H);
        const long f = t / ((long)W * H);
        const long tb = (f * (H / 2) + y / 2) * (W / 2) + x / 2;
        float fa[8], fb[8];
        up8(*(const u32x4*)(a + t * C + ch * 8), fa);
        up8(*(const u32x4*)(b + tb * C + ch * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += fb[e];
        *(u32x4*)(o + t * C + ch * 8) = pk8(fa);
    }
}

// ---- out[r, :] = a[r, :] + alpha * b[r % rows_b, :]
__global__ __launch_bounds__(256) void add_bcast_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                        unsigned short* __restrict__ o, long rows, long rows_b, int C, long lda, long ldb, long ldo,
                                                        float alpha) {
    const int nch = C / 8;
    const long total = rows * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long r = i / nch;
        float fa[8], fb[8];
        up8(*(const u32x4*)(a + r * lda + ch * 8), fa);
        up8(*(const u32x4*)(b + (r % rows_b) * ldb + ch * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += alpha * fb[e];
        *(u32x4*)(o + r * ldo + ch * 8) = pk8(fa);
    }
}

// ---- bilinear resize (align_corners=False, PyTorch area_pixel_compute_source_index), planes [N, Hi, Wi] -> [N, Ho, Wo] f32.
//      plane_idx (optional): output plane n reads input plane plane_idx[n] (select-then-upsample of the argmax-IoU mask).
template <bool IN_BF16>
__global__ __launch_bounds__(256) void bilinear_kernel(const void* __restrict__ in, float* __restrict__ out, const int* __restrict__ plane_idx,
                                                       long N, int Hi, int Wi, int Ho, int Wo) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const long total = N * Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const long n = i / ((long)Wo * Ho);
        const long pn = plane_idx ? plane_idx[n] : n;
        float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        auto ld = [&](int yy, int xx) -> float {
            const long o = (pn * Hi + yy) * (long)Wi + xx;
            if (IN_BF16) return bf2f(((const unsigned short*)in)[o]);
            return ((const float*)in)[o];
        };
        const float v = (1.f - ly) * ((1.f - lx) * ld(y0, x0) + lx * ld(y0, x1)) + ly * ((1.f - lx) * ld(y1, x0) + lx * ld(y1, x1));
        out[i] = v;
    }
}

// ---- Conv2d(k=3, s=2, p=1) on token-major maps: x [F, H, W, Cin] -> y [F, H/2, W/2, Cout]; w [Cout, Cin, 3, 3]; f32 accumulate.
//      IN_F32: x is an fp32 single-channel plane (the mask), transformed on load by sigmoid(x)*ascale+abias if ascale != 0.
template <bool IN_F32>
__global__ __launch_bounds__(256) void conv3x3s2_kernel(const void* __restrict__ x, const unsigned short* __restrict__ w, const unsigned short* __restrict__ bias,
                                                        unsigned short* __restrict__ y, long F, int H, int W, int Cin, int Cout, float ascale, float abias) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = F * Ho * Wo * Cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        const long t = i / Cout;
        const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho);
        const long f = t / ((long)Wo * Ho);
        float acc = bias ? bf2f(bias[co]) : 0.f;
        for (int kh = 0; kh < 3; ++kh) {
            const int sy = oy * 2 - 1 + kh;
            if (sy < 0 || sy >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int sx = ox * 2 - 1 + kw;
                if (sx < 0 || sx >= W) continue;
                const long pix = (f * H + sy) * (long)W + sx;
                for (int ci = 0; ci < Cin; ++ci) {
                    float xv;
                    if (IN_F32) {
                        xv = ((const float*)x)[pix * Cin + ci];
                        if (ascale != 0.f) xv = bf2f(f2bf(ascale / (1.f + __expf(-xv)) + abias));
                    } else {
                        xv = bf2f(((const unsigned short*)x)[pix * Cin + ci]);
                    }
                    acc += xv * bf2f(w[((co * Cin + ci) * 3 + kh) * 3 + kw]);
                }
            }
        }
        y[i] = f2bf(acc);
    }
}

// ---- The first two stages of the memory encoder's mask down-sampler in one launch each (reference model/sam2.py:611-643 MaskDownSampler: Conv2d(k=3, s=2, p=1) ->
//      LayerNorm2d -> GELU, 1 -> 4 channels at 1024 x 1024 and 4 -> 16 at 512 x 512).  A thread owns one OUTPUT PIXEL with all its channels, so the channel
//      LayerNorm and the GELU are in-thread arithmetic on the convolution's result; filter, bias, gamma and beta sit in LDS as f32 and are read as broadcasts.
//      The one-thread-per-output-element kernel above issued a 2-byte load per multiply and re-evaluated the input sigmoid once per output channel: 16 + 40 us for
//      the two stages, + 5.6 + 11.2 us for their LayerNorm launches (profiles/r03_stream_frame_timeline_fused_tail.txt).  Same taps in the same order, the same
//      bf16 rounding of the convolution's output before the statistics, LayerNorm sums associated as the stand-alone kernels associate them (4 channels: in sequence;
//      16: two runs of 8, then their sum): equal to conv3x3s2 + layernorm(act = gelu) except for the last bf16 bit of a few elements per million (the compiler
//      contracts the two forms differently).
template <int CIN, int COUT, bool IN_F32>
__global__ __launch_bounds__(256) void conv3x3s2_ln_gelu_kernel(const void* __restrict__ x, const unsigned short* __restrict__ w, const unsigned short* __restrict__ bias,
                                                                const unsigned short* __restrict__ ln_w, const unsigned short* __restrict__ ln_b, float eps,
                                                                unsigned short* __restrict__ y, long F, int H, int W, float ascale, float abias) {
    __shared__ float wf[COUT * CIN * 9], bf[COUT], gf[COUT], hf[COUT];
    for (int i = threadIdx.x; i < COUT * CIN * 9; i += 256) wf[i] = bf2f(w[i]);
    if (threadIdx.x < COUT) {
        bf[threadIdx.x] = bias ? bf2f(bias[threadIdx.x]) : 0.f;
        gf[threadIdx.x] = bf2f(ln_w[threadIdx.x]);
        hf[threadIdx.x] = ln_b ? bf2f(ln_b[threadIdx.x]) : 0.f;
    }
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2;
    const long total = F * Ho * Wo;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho);
        const long f = t / ((long)Wo * Ho);
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = bf[co];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int sy = oy * 2 - 1 + kh;
            if (sy < 0 || sy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int sx = ox * 2 - 1 + kw;
                if (sx < 0 || sx >= W) continue;
                const long pix = (f * H + sy) * (long)W + sx;
                float xv[CIN];
                if constexpr (IN_F32) {
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        float v = ((const float*)x)[pix * CIN + ci];
                        if (ascale != 0.f) v = bf2f(f2bf(ascale / (1.f + __expf(-v)) + abias));
                        xv[ci] = v;
                    }
                } else if constexpr (CIN == 4) {
                    const u32x2 pk = *(const u32x2*)((const unsigned short*)x + pix * 4);
                    xv[0] = __uint_as_float(pk[0] << 16); xv[1] = __uint_as_float(pk[0] & 0xffff0000u);
                    xv[2] = __uint_as_float(pk[1] << 16); xv[3] = __uint_as_float(pk[1] & 0xffff0000u);
                } else {
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) xv[ci] = bf2f(((const unsigned short*)x)[pix * CIN + ci]);
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc[co] += xv[ci] * wf[((co * CIN + ci) * 3 + kh) * 3 + kw];
            }
        }
        // LayerNorm over the channels of this pixel on the bf16-rounded convolution output, then the exact-erf GELU of the bf16-rounded normalised value
        float v[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) v[co] = bf2f(f2bf(acc[co]));
        float s1 = 0.f, s2 = 0.f;
        if constexpr (COUT == 16) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a += v[e]; b += v[8 + e]; }
            s1 = a + b;
        } else {
#pragma unroll
            for (int e = 0; e < COUT; ++e) s1 += v[e];
        }
        const float mean = s1 / (float)COUT;
        if constexpr (COUT == 16) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d0 = v[e] - mean, d1 = v[8 + e] - mean; a += d0 * d0; b += d1 * d1; }
            s2 = a + b;
        } else {
#pragma unroll
            for (int e = 0; e < COUT; ++e) { const float d = v[e] - mean; s2 += d * d; }
        }
        const float rinv = rsqrtf(s2 / (float)COUT + eps);
        unsigned short o[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float n = bf2f(f2bf((v[co] - mean) * rinv * gf[co] + hf[co]));
            o[co] = f2bf(0.5f * n * (1.0f + erff(n * 0.70710678118654752f)));
        }
        unsigned short* dst = y + t * COUT;
        if constexpr (COUT % 8 == 0) {
#pragma unroll
            for (int c8 = 0; c8 < COUT / 8; ++c8) {
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (unsigned)o[c8 * 8 + 2 * e] | ((unsigned)o[c8 * 8 + 2 * e + 1] << 16);
                *(u32x4*)(dst + c8 * 8) = pk;
            }
        } else {
            static_assert(COUT == 4, "4 or a multiple of 8 output channels");
            u32x2 pk = {(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
            *(u32x2*)dst = pk;
        }
    }
}

// ---- Several device-to-device copies in ONE launch (<= 24 segments of 16-byte chunks): what a video-session frame moves around its captured graph -- the frame's
//      tokens and two feature maps and up to 6 memories + 15 pointers into the graph's static inputs, 4 results out -- was 9 - 13 copy launches of ~5 us each between
//      two graph replays (profiles/r03_stream_frame_timeline_fused_tail.txt: __amd_rocclr_copyBuffer x 7 + the torch.cat / clone kernels).
constexpr int CM_MAX = 24;
struct CopyMany {
    const u32x4* src[CM_MAX];
    u32x4* dst[CM_MAX];
    long chunks[CM_MAX];
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyMany p) {
    const int sgm = blockIdx.y;
    const u32x4* __restrict__ s = p.src[sgm];
    u32x4* __restrict__ d = p.dst[sgm];
    const long n = p.chunks[sgm];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = s[i];
}

// ---- im2col for Conv2d(k=3, s=2, p=1) on token-major maps: x [F, H, W, C] -> cols [F*(H/2)*(W/2), 9*C], K index = (kh*3 + kw)*C + c,
//      zero padding.  Feeds the NT GEMM with the weight repacked [Cout, (kh, kw, ci)]: the direct kernel above issues one 2-byte load per
//      multiply and took 0.8 ms for the memory encoder's 64 -> 256 channel stage (0.3 GFLOP).
__global__ __launch_bounds__(256) void im2col3x3s2_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ cols, long F, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, nch = C / 8;
    const long total = F * Ho * Wo * 9 * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const int tap = (int)((i / nch) % 9);
        const long t = i / (9L * nch);
        const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho);
        const long f = t / ((long)Wo * Ho);
        const int sy = oy * 2 - 1 + tap / 3, sx = ox * 2 - 1 + tap % 3;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = *(const u32x4*)(x + ((f * H + sy) * (long)W + sx) * C + ch * 8);
        *(u32x4*)(cols + (t * 9 + tap) * C + ch * 8) = v;
    }
}

// ---- depthwise Conv2d(k=7, p=3) on token-major maps [F, H, W, C]; w [C, 1, 7, 7] (reference model/sam2.py:690-703 CXBlock.dwconv).
// A workgroup takes an 8 x 8 pixel tile x 64 channels: the 14 x 14 x 64 input halo (25 KB) and the 49 x 64 taps go to LDS once -- every global load of the workgroup in
// flight together -- and each of the 512 threads (pixel, 8 channels) walks its 49 taps out of LDS.  History of this kernel on one 64 x 64 x 256 map (4 MB in + out):
// 60 us with a scalar 2-byte global load per tap and channel; 23 us with the taps in LDS but the input read tap by tap from global memory (49 dependent-latency
// rounds at two waves per SIMD), of which ~10 us were a badly staged tap table; 18 us with row-batched loads.  Same taps in the same order, f32 sums: the same bits.
constexpr int DW_T = 8, DW_HALO = DW_T + 6, DW_CB = 64;
__global__ __launch_bounds__(512) void dwconv7_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                      const unsigned short* __restrict__ bias, unsigned short* __restrict__ y, long F, int H, int W, int C) {
    __shared__ __attribute__((aligned(16))) unsigned short xin[DW_HALO * DW_HALO * DW_CB];   // [14][14][64]
    __shared__ __attribute__((aligned(16))) unsigned short raw[DW_CB * 49 + 8];              // this channel block's taps as in memory: [64][49]
    __shared__ __attribute__((aligned(16))) unsigned short taps[49 * DW_CB];                 // [49][64]
    const int tid = threadIdx.x;
    const int tilesx = (W + DW_T - 1) / DW_T, tilesy = (H + DW_T - 1) / DW_T;
    const int tx = blockIdx.x % tilesx, ty = (blockIdx.x / tilesx) % tilesy;
    const long f = blockIdx.x / ((long)tilesx * tilesy);
    const int c0 = blockIdx.y * DW_CB;
    const int cb = min(DW_CB, C - c0);             // channels of this block (multiple of 8)
    const int x0 = tx * DW_T - 3, y0 = ty * DW_T - 3;
    // halo: 196 pixels x 8 chunks of 16 bytes; out-of-map pixels are zeros (adding 0 * w leaves a sum as it was: the same bits as skipping the tap)
    for (int i = tid; i < DW_HALO * DW_HALO * (DW_CB / 8); i += 512) {
        const int ch = i & 7, pix = i >> 3;
        const int hy = pix / DW_HALO, hx = pix - hy * DW_HALO;
        const int sy = y0 + hy, sx = x0 + hx;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W && ch * 8 < cb) v = *(const u32x4*)(x + ((f * H + sy) * (long)W + sx) * C + c0 + ch * 8);
        *(u32x4*)(xin + pix * DW_CB + ch * 8) = v;
    }
    // taps of channels c0 .. c0 + cb - 1: contiguous cb x 49 values, copied as they are, then transposed inside LDS
    {
        const unsigned short* wsrc = w + (long)c0 * 49;
        const int n = cb * 49;
        const int head = (int)(((16 - ((uintptr_t)wsrc & 15)) & 15) / 2);     // elements before the first 16-byte boundary
        for (int i = tid; i < min(head, n); i += 512) raw[i] = wsrc[i];
        const int nchunk = (n - min(head, n)) / 8;
        for (int i = tid; i < nchunk; i += 512) {
            const u32x4 v = *(const u32x4*)(wsrc + head + i * 8);
            unsigned short* d = raw + head + i * 8;     // raw + head is only 2-byte aligned in general: element stores
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[2 * e] = (unsigned short)(v[e] & 0xffffu); d[2 * e + 1] = (unsigned short)(v[e] >> 16); }
        }
        for (int i = min(head, n) + nchunk * 8 + tid; i < n; i += 512) raw[i] = wsrc[i];
    }
    __syncthreads();
    for (int i = tid; i < 49 * DW_CB; i += 512) {
        const int k = i / DW_CB, ch1 = i - k * DW_CB;
        taps[i] = ch1 < cb ? raw[ch1 * 49 + k] : (unsigned short)0;
    }
    __syncthreads();
    const int cg = tid & 7, pp = tid >> 3;            // 8 channels, pixel of the tile
    const int ly = pp >> 3, lx = pp & 7;
    const int py = ty * DW_T + ly, px = tx * DW_T + lx;
    if (py >= H || px >= W || cg * 8 >= cb) return;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bf2f(bias[c0 + cg * 8 + e]) : 0.f;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
            float fx[8], fw[8];
            up8(*(const u32x4*)(xin + ((ly + kh) * DW_HALO + lx + kw) * DW_CB + cg * 8), fx);
            up8(*(const u32x4*)(taps + (kh * 7 + kw) * DW_CB + cg * 8), fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += fx[e] * fw[e];
        }
    *(u32x4*)(y + ((f * H + py) * (long)W + px) * C + c0 + cg * 8) = pk8(acc);
}

// ---- axial complex RoPE in place (reference sam2.py:1901-1923): consecutive (even, odd) pairs, token t < n_rope uses
//      table row t % nq; cos/sin [nq, C/2] f32; x [T, C] bf16 with row stride ldx (single head).
__global__ __launch_bounds__(256) void rope_axial_kernel(unsigned short* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                         long n_rope, int nq, int C, long ldx) {
    const int nch = C / 8;
    const long total = n_rope * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long t = i / nch;
        const long tr = t % nq;
        float f[8];
        unsigned short* p = x + t * ldx + ch * 8;
        up8(*(const u32x4*)p, f);
        const float* c = cs + tr * (C / 2) + ch * 4;
        const float* s = sn + tr * (C / 2) + ch * 4;
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[2 * e] = f[2 * e] * c[e] - f[2 * e + 1] * s[e];
            o[2 * e + 1] = f[2 * e] * s[e] + f[2 * e + 1] * c[e];
        }
        *(u32x4*)p = pk8(o);
    }
}

// ---- ConvTranspose2d(k=2, s=2) second half: g [F*H*W, 4*Co] (GEMM output, column (dy*2+dx)*Co + co) -> token-major
//      [F, 2H, 2W, Co] with bias and an optional added map (high-res feature), reference sam2.py:2137-2140.
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const unsigned short* __restrict__ g, const unsigned short* __restrict__ bias,
                                                            const unsigned short* __restrict__ add, unsigned short* __restrict__ o, long F, int H, int W, int Co, int act) {
    const int nch = Co / 8;
    const long total = F * (2 * H) * (2 * W) * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long t = i / nch;
        const int X = (int)(t % (2 * W)), Y = (int)((t / (2 * W)) % (2 * H));
        const long f = t / ((long)4 * W * H);
        const long src = (f * H + Y / 2) * (long)W + X / 2;
        const int q = (Y & 1) * 2 + (X & 1);
        float fg[8], fb[8];
        up8(*(const u32x4*)(g + src * (4L * Co) + q * Co + ch * 8), fg);
        if (bias) {
            up8(*(const u32x4*)(bias + ch * 8), fb);
#pragma unroll
            for (int e = 0; e < 8; ++e) fg[e] = bf2f(f2bf(fg[e] + fb[e]));
        }
        if (add) {
            up8(*(const u32x4*)(add + t * Co + ch * 8), fb);
#pragma unroll
            for (int e = 0; e < 8; ++e) fg[e] += fb[e];
        }
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float tv = bf2f(f2bf(fg[e])); fg[e] = 0.5f * tv * (1.0f + erff(tv * 0.70710678118654752f)); }
        }
        *(u32x4*)(o + t * Co + ch * 8) = pk8(fg);
    }
}

// ---- per-mask partial sums for BCE-with-logits and dice (reference qwen_2_5_vl_sam2.py:17-60):
//      out[n] = { sum bce(x,t), sum sigmoid(x)*t, sum sigmoid(x), sum t }; one block per (mask, slice), atomics into out.
__global__ __launch_bounds__(256) void bce_dice_kernel(const float* __restrict__ x, const float* __restrict__ tg, float* __restrict__ out, long hw,
                                                       float* __restrict__ part = nullptr) {
    __shared__ float red[4][4];
    const long n = blockIdx.y;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float v = x[n * hw + i], t = tg[n * hw + i];
        s0 += fmaxf(v, 0.f) - v * t + log1pf(__expf(-fabsf(v)));  // stable BCE-with-logits
        const float p = 1.f / (1.f + __expf(-v));
        s1 += p * t;
        s2 += p;
        s3 += t;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = s0; red[wv][1] = s1; red[wv][2] = s2; red[wv][3] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (part) part[((long)n * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = v;   // summed in block order by bce_dice_finish_kernel: reproducible
        else atomicAdd(out + n * 4 + threadIdx.x, v);
    }
}

__global__ __launch_bounds__(64) void bce_dice_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nb) {
    const long n = blockIdx.x;
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += part[(n * nb + b) * 4 + threadIdx.x];
        out[n * 4 + threadIdx.x] = s;
    }
}

static inline unsigned grid1(long total, long cap = 256L * 32) {
    long b = cdiv(total, 256);
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace rga3

using namespace rga3;
typedef const unsigned short* cus;
typedef unsigned short* us;

extern "C" int rga3_im2col(const void* img, void* out, int64_t F, int C, int H, int W, int ks, int stride, int pad, int64_t ld_out,
                           void* stream) {
    RGA3_CHECK_ARG(img && out && F > 0 && C > 0 && H > 0 && W > 0 && ks > 0 && stride > 0, "im2col: bad args");
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    RGA3_CHECK_ARG(ld_out >= (int64_t)C * ks * ks && ld_out % 8 == 0 && (((uintptr_t)out) & 15) == 0, "im2col: ld_out too small / not a multiple of 8, or out not 16-byte aligned");
    hipLaunchKernelGGL(im2col_kernel, dim3(grid1(F * Ho * Wo * (ld_out / 8))), dim3(256), 0, (hipStream_t)stream, (cus)img, (us)out, (int)F, C, H, W, ks,
                       stride, pad, Ho, Wo, (int)ld_out);
    RGA3_CHECK_LAUNCH("im2col");
    return 0;
}

extern "C" int rga3_maxpool2x2_win(const void* x, void* y, int64_t nwin, int w, int C, int64_t ldx, int64_t ldy, void* stream) {
    RGA3_CHECK_ARG(x && y && nwin > 0 && w >= 2 && w % 2 == 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "maxpool2x2_win: bad args");
    hipLaunchKernelGGL(maxpool_win_kernel, dim3(grid1(nwin * (w / 2) * (w / 2) * (C / 8))), dim3(256), 0, (hipStream_t)stream, (cus)x, (us)y,
                       (long)nwin, w, C, (long)ldx, (long)ldy);
    RGA3_CHECK_LAUNCH("maxpool2x2_win");
    return 0;
}

extern "C" int rga3_upsample2x_add(const void* a, const void* b, void* out, int64_t F, int H, int W, int C, void* stream) {
    RGA3_CHECK_ARG(a && b && out && F > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "upsample2x_add: bad args");
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid1(F * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (cus)a, (cus)b, (us)out, (long)F, H, W, C);
    RGA3_CHECK_LAUNCH("upsample2x_add");
    return 0;
}

extern "C" int rga3_add_bcast(const void* a, const void* b, void* out, int64_t rows, int64_t rows_b, int C, int64_t lda, int64_t ldb,
                              int64_t ldo, float alpha, void* stream) {
    RGA3_CHECK_ARG(a && b && out && rows > 0 && rows_b > 0 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "add_bcast: bad args");
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid1(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream, (cus)a, (cus)b, (us)out, (long)rows,
                       (long)rows_b, C, (long)lda, (long)ldb, (long)ldo, alpha);
    RGA3_CHECK_LAUNCH("add_bcast");
    return 0;
}

extern "C" int rga3_bilinear(const void* in, int in_dtype, float* out, const int32_t* plane_idx, int64_t N, int Hi, int Wi, int Ho, int Wo,
                             void* stream) {
    RGA3_CHECK_ARG(in && out && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bilinear: bad args");
    RGA3_CHECK_ARG(in_dtype == RGA3_BF16 || in_dtype == RGA3_F32, "bilinear: dtype");
    dim3 g(grid1(N * Ho * Wo));
    if (in_dtype == RGA3_BF16)
        hipLaunchKernelGGL(bilinear_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, in, out, plane_idx, (long)N, Hi, Wi, Ho, Wo);
    else
        hipLaunchKernelGGL(bilinear_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, in, out, plane_idx, (long)N, Hi, Wi, Ho, Wo);
    RGA3_CHECK_LAUNCH("bilinear");
    return 0;
}

extern "C" int rga3_conv3x3s2(const void* x, int x_dtype, const void* w, const void* bias, void* y, int64_t F, int H, int W, int Cin, int Cout,
                              float sig_scale, float sig_bias, void* stream) {
    RGA3_CHECK_ARG(x && w && y && F > 0 && H % 2 == 0 && W % 2 == 0 && Cin > 0 && Cout > 0, "conv3x3s2: bad args");
    dim3 g(grid1(F * (H / 2) * (W / 2) * Cout));
    if (x_dtype == RGA3_F32)
        hipLaunchKernelGGL(conv3x3s2_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, x, (cus)w, (cus)bias, (us)y, (long)F, H, W, Cin, Cout, sig_scale, sig_bias);
    else
        hipLaunchKernelGGL(conv3x3s2_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, x, (cus)w, (cus)bias, (us)y, (long)F, H, W, Cin, Cout, 0.f, 0.f);
    RGA3_CHECK_LAUNCH("conv3x3s2");
    return 0;
}

// Conv2d(k=3, s=2, p=1) + LayerNorm over the output channels + exact GELU in one launch, for the two narrow stages of the mask down-sampler: (Cin, Cout) = (1, 4) with
// an f32 input plane (optionally sigmoid(x) * sig_scale + sig_bias on load) or (4, 16) with a bf16 token-major input.  y [F, H/2, W/2, Cout] bf16.
extern "C" int rga3_conv3x3s2_ln_gelu(const void* x, int x_dtype, const void* w, const void* bias, const void* ln_w, const void* ln_b, float eps, void* y, int64_t F, int H,
                                      int W, int Cin, int Cout, float sig_scale, float sig_bias, void* stream) {
    RGA3_CHECK_ARG(x && w && ln_w && y && F > 0 && H % 2 == 0 && W % 2 == 0, "conv3x3s2_ln_gelu: bad args");
    RGA3_CHECK_ARG((x_dtype == RGA3_F32 && Cin == 1 && Cout == 4) || (x_dtype == RGA3_BF16 && Cin == 4 && Cout == 16),
                   "conv3x3s2_ln_gelu: (Cin, Cout) = (%d, %d): (1, 4) from an f32 plane or (4, 16) from bf16", Cin, Cout);
    RGA3_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "conv3x3s2_ln_gelu: 16-byte alignment");
    dim3 g(grid1(F * (H / 2) * (W / 2)));
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1)
        hipLaunchKernelGGL((conv3x3s2_ln_gelu_kernel<1, 4, true>), g, dim3(256), 0, st, x, (cus)w, (cus)bias, (cus)ln_w, (cus)ln_b, eps, (us)y, (long)F, H, W, sig_scale, sig_bias);
    else
        hipLaunchKernelGGL((conv3x3s2_ln_gelu_kernel<4, 16, false>), g, dim3(256), 0, st, x, (cus)w, (cus)bias, (cus)ln_w, (cus)ln_b, eps, (us)y, (long)F, H, W, 0.f, 0.f);
    RGA3_CHECK_LAUNCH("conv3x3s2_ln_gelu");
    return 0;
}

// n (<= 24) device-to-device copies in one launch: dst[i] <- src[i], bytes[i] bytes each (multiples of 16, 16-byte aligned, no overlap).  HOST arrays.
extern "C" int rga3_copy_many(void* const* dst, const void* const* src, const int64_t* bytes, int n, void* stream) {
    RGA3_CHECK_ARG(dst && src && bytes && n >= 1 && n <= CM_MAX, "copy_many: n %d (1..%d)", n, CM_MAX);
    CopyMany p;
    long most = 0;
    for (int i = 0; i < n; ++i) {
        RGA3_CHECK_ARG(dst[i] && src[i] && bytes[i] > 0 && bytes[i] % 16 == 0 && ((((uintptr_t)dst[i]) | ((uintptr_t)src[i])) & 15) == 0,
                       "copy_many: segment %d: %ld bytes (multiple of 16, 16-byte aligned pointers)", i, (long)bytes[i]);
        p.src[i] = (const u32x4*)src[i];
        p.dst[i] = (u32x4*)dst[i];
        p.chunks[i] = bytes[i] / 16;
        if (p.chunks[i] > most) most = p.chunks[i];
    }
    unsigned gx = (unsigned)cdiv(most, 256 * 4);     // ~4 chunks per thread for the largest segment; smaller segments leave their extra workgroups idle
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(copy_many_kernel, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    RGA3_CHECK_LAUNCH("copy_many");
    return 0;
}

extern "C" int rga3_im2col3x3s2(const void* x, void* cols, int64_t F, int H, int W, int C, void* stream) {
    RGA3_CHECK_ARG(x && cols && F > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, "im2col3x3s2: bad args");
    hipLaunchKernelGGL(im2col3x3s2_kernel, dim3(grid1(F * (H / 2) * (W / 2) * 9L * (C / 8))), dim3(256), 0, (hipStream_t)stream, (cus)x, (us)cols, (long)F, H, W, C);
    RGA3_CHECK_LAUNCH("im2col3x3s2");
    return 0;
}

extern "C" int rga3_dwconv7x7(const void* x, const void* w, const void* bias, void* y, int64_t F, int H, int W, int C, void* stream) {
    RGA3_CHECK_ARG(x && w && y && F > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "dwconv7x7: bad args (C %d: a multiple of 8)", C);
    RGA3_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "dwconv7x7: 16-byte alignment of the maps");
    const long tiles = (long)cdiv(W, DW_T) * cdiv(H, DW_T) * F;
    RGA3_CHECK_ARG(tiles < (1L << 31) && cdiv(C, DW_CB) <= 65535, "dwconv7x7: grid");
    hipLaunchKernelGGL(dwconv7_kernel, dim3((unsigned)tiles, (unsigned)cdiv(C, DW_CB)), dim3(512), 0, (hipStream_t)stream, (cus)x, (cus)w, (cus)bias, (us)y, (long)F, H, W, C);
    RGA3_CHECK_LAUNCH("dwconv7x7");
    return 0;
}

extern "C" int rga3_rope_axial_inplace(void* x, const float* cos, const float* sin, int64_t n_rope, int nq, int C, int64_t ldx, void* stream) {
    RGA3_CHECK_ARG(x && cos && sin && nq > 0 && C % 8 == 0 && ldx % 8 == 0, "rope_axial: bad args");
    if (n_rope <= 0) return 0;
    hipLaunchKernelGGL(rope_axial_kernel, dim3(grid1(n_rope * (C / 8))), dim3(256), 0, (hipStream_t)stream, (us)x, cos, sin, (long)n_rope, nq, C, (long)ldx);
    RGA3_CHECK_LAUNCH("rope_axial");
    return 0;
}

extern "C" int rga3_pixel_shuffle2x(const void* g, const void* bias, const void* add, void* out, int64_t F, int H, int W, int Co, int act, void* stream) {
    RGA3_CHECK_ARG(g && out && F > 0 && Co % 8 == 0, "pixel_shuffle2x: bad args");
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid1(F * 4L * H * W * (Co / 8))), dim3(256), 0, (hipStream_t)stream, (cus)g, (cus)bias, (cus)add, (us)out,
                       (long)F, H, W, Co, act);
    RGA3_CHECK_LAUNCH("pixel_shuffle2x");
    return 0;
}

extern "C" int rga3_bce_dice_sums(const float* logits, const float* targets, float* out4, int64_t n_masks, int64_t hw, void* stream) {
    RGA3_CHECK_ARG(logits && targets && out4 && n_masks > 0 && hw > 0 && n_masks <= 65535, "bce_dice_sums: bad args");
    hipError_t e = hipMemsetAsync(out4, 0, sizeof(float) * 4 * n_masks, (hipStream_t)stream);
    if (e != hipSuccess) return fail(-(int)e, "bce_dice_sums: memset: %s", hipGetErrorString(e));
    unsigned gx = (unsigned)cdiv(hw, 256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(bce_dice_kernel, dim3(gx, (unsigned)n_masks), dim3(256), 0, (hipStream_t)stream, logits, targets, out4, (long)hw, (float*)nullptr);
    RGA3_CHECK_LAUNCH("bce_dice_sums");
    return 0;
}

static unsigned bce_dice_blocks(int64_t hw) {
    unsigned gx = (unsigned)cdiv(hw, 256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 128) gx = 128;
    return gx;
}

extern "C" int64_t rga3_bce_dice_sums_ws_floats(int64_t n_masks, int64_t hw) { return n_masks * (int64_t)bce_dice_blocks(hw) * 4; }

// The same four sums per mask, reproducible run to run: per-block partial sums in the caller's workspace (rga3_bce_dice_sums_ws_floats() f32 elements), added in
// block order; no memset, no atomics (the atomic form above adds its <= 512 block sums in arrival order: the mask losses and their gradients moved in the last bits
// from run to run).
extern "C" int rga3_bce_dice_sums_det(const float* logits, const float* targets, float* out4, float* ws, int64_t ws_floats, int64_t n_masks, int64_t hw, void* stream) {
    RGA3_CHECK_ARG(logits && targets && out4 && ws && n_masks > 0 && hw > 0 && n_masks <= 65535, "bce_dice_sums_det: bad args");
    const unsigned gx = bce_dice_blocks(hw);
    RGA3_CHECK_ARG(ws_floats >= n_masks * (int64_t)gx * 4, "bce_dice_sums_det: workspace of rga3_bce_dice_sums_ws_floats() f32 elements needed");
    hipLaunchKernelGGL(bce_dice_kernel, dim3(gx, (unsigned)n_masks), dim3(256), 0, (hipStream_t)stream, logits, targets, out4, (long)hw, ws);
    RGA3_CHECK_LAUNCH("bce_dice_sums_det");
    hipLaunchKernelGGL(bce_dice_finish_kernel, dim3((unsigned)n_masks), dim3(64), 0, (hipStream_t)stream, (const float*)ws, out4, (int)gx);
    RGA3_CHECK_LAUNCH("bce_dice_finish");
    return 0;
}
