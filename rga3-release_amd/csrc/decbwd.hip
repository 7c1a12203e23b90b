// Backward kernels of the SAM2 mask path (trainable mask decoder + text_hidden_fcs, reference train_joint.py:237-251):
// LayerNorm backward, column sums (bias grads), GELU fwd / activation backward, bilinear backward, pixel-shuffle backward,
// BCE + dice gradient.  Small tensors, HBM-bound; fp32 accumulation, fp32 atomics only where a reduction crosses workgroups.
#include "common.h"

namespace rga3 {

__device__ __forceinline__ void u8_(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 p8_(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}

// y = (x - mean) * r * w + b.  dx = r * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;  dw += dy * xhat, db += dy.
// One wave per row (dim <= 512*MAXC/8...), a workgroup walks ROWS_PER_WG rows and reduces dw/db in registers -> LDS -> atomics.
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                            const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, long rows, int dim, float eps, int rows_per_wg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nch = dim / 8;
    float aw[MAXC][8], ab[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ab[i][e] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_wg;
    for (long row = r0 + wv; row < min(rows, r0 + rows_per_wg); row += 4) {
        u32x4 xb[MAXC], db_[MAXC];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
                xb[i] = *(const u32x4*)(x + row * dim + ch * 8);
                db_[i] = *(const u32x4*)(dy + row * dim + ch * 8);
                float f[8];
                u8_(xb[i], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s1 += f[e];
            }
        }
        const float mean = wave_sum(s1) / (float)dim;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
                float f[8];
                u8_(xb[i], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { float d = f[e] - mean; s2 += d * d; }
            }
        }
        const float r = rsqrtf(wave_sum(s2) / (float)dim + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
                float fx[8], fd[8], fw[8];
                u8_(xb[i], fx);
                u8_(db_[i], fd);
                u8_(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (fx[e] - mean) * r;
                    const float g = fd[e] * fw[e];
                    sg += g;
                    sgx += g * xh;
                    aw[i][e] += fd[e] * xh;
                    ab[i][e] += fd[e];
                }
            }
        }
        sg = wave_sum(sg) / (float)dim;
        sgx = wave_sum(sgx) / (float)dim;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
                float fx[8], fd[8], fw[8], o[8];
                u8_(xb[i], fx);
                u8_(db_[i], fd);
                u8_(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (fx[e] - mean) * r;
                    o[e] = r * (fd[e] * fw[e] - sg - xh * sgx);
                }
                *(u32x4*)(dx + row * dim + ch * 8) = p8_(o);
            }
        }
    }
    if (dw) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(dw + ch * 8 + e, aw[i][e]);
                    atomicAdd(db + ch * 8 + e, ab[i][e]);
                }
            }
        }
    }
}

// Narrow rows (LayerNorm2d on 16..128 channels over 10^5..10^6 pixels: the mask decoder's upscaling path, reference model/sam2.py:1976-1986):
// G = dim / 8 lanes per row, 64 / G rows per wave pass; row statistics by xor-shuffles inside the lane group; dw / db accumulate per lane
// over the rows it sees, are folded across the groups of the wave by shuffles and leave by one atomic per (wave, column).  The
// wave-per-row form above keeps 4 of 64 lanes busy at 32 channels (3.2 ms for 2^20 rows).
template <int G>
__global__ __launch_bounds__(256) void layernorm_bwd_narrow_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                                   const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                                                                   float* __restrict__ dw, float* __restrict__ db, long rows, float eps, int rows_per_wg) {
    constexpr int DIM = G * 8, RPW = 64 / G;   // rows per wave pass
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = lane % G, sub = lane / G;
    float fw[8], aw[8], ab[8];
    u8_(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[e] = 0.f; ab[e] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = min(rows, r0 + rows_per_wg);
    for (long base = r0 + wv * RPW; base < r1; base += 4 * RPW) {
        const long row = base + sub;
        const bool ok = row < r1;
        float fx[8], fd[8];
        u32x4 vx = {0u, 0u, 0u, 0u}, vd = {0u, 0u, 0u, 0u};
        if (ok) {
            vx = *(const u32x4*)(x + row * DIM + ch * 8);
            vd = *(const u32x4*)(dy + row * DIM + ch * 8);
        }
        u8_(vx, fx);
        u8_(vd, fd);
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += fx[e];
#pragma unroll
        for (int o = 1; o < G; o <<= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)DIM;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = fx[e] - mean; s2 += d * d; }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) s2 += __shfl_xor(s2, o, 64);
        const float r = rsqrtf(s2 / (float)DIM + eps);
        float sg = 0.f, sgx = 0.f, xh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh[e] = (fx[e] - mean) * r;
            const float g = fd[e] * fw[e];
            sg += g;
            sgx += g * xh[e];
            if (ok) { aw[e] += fd[e] * xh[e]; ab[e] += fd[e]; }
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
        sg /= (float)DIM;
        sgx /= (float)DIM;
        if (ok) {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = r * (fd[e] * fw[e] - sg - xh[e] * sgx);
            *(u32x4*)(dx + row * DIM + ch * 8) = p8_(o8);
        }
    }
    if (dw) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) { aw[e] += __shfl_xor(aw[e], o, 64); ab[e] += __shfl_xor(ab[e], o, 64); }
        }
        if (sub == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(dw + ch * 8 + e, aw[e]);
                atomicAdd(db + ch * 8 + e, ab[e]);
            }
        }
    }
}

// out[c] += sum_r x[r, c]  (bf16 in, f32 atomics out); workgroup = 256 columns x rows_per_wg rows
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ x, float* __restrict__ out, long rows, long cols, long ld,
                                                     int rows_per_wg) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long r0 = (long)blockIdx.y * rows_per_wg;
    float s = 0.f;
    for (long r = r0; r < min(rows, r0 + rows_per_wg); ++r) s += bf2f(x[r * ld + c]);
    atomicAdd(out + c, s);
}

// kind 0: y = gelu(x) ; kind 1: dx = dy * gelu'(x) (x = pre-activation) ; kind 2: dx = dy * (y > 0) (y = relu output)
__global__ __launch_bounds__(256) void act_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ dy,
                                                  unsigned short* __restrict__ o, long n, int kind) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = bf2f(a[i]);
        float r;
        if (kind == 0) {
            r = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        } else if (kind == 1) {
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            r = bf2f(dy[i]) * (cdf + x * pdf);
        } else {
            r = x > 0.f ? bf2f(dy[i]) : 0.f;
        }
        o[i] = f2bf(r);
    }
}

// backward of bilinear_kernel: din[plane_idx[n] or n] += taps * dout   (f32 atomics; din pre-zeroed by the caller)
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, const int* __restrict__ plane_idx,
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
        const float g = dout[i];
        float* base = din + pn * Hi * (long)Wi;
        atomicAdd(base + y0 * (long)Wi + x0, g * (1.f - ly) * (1.f - lx));
        atomicAdd(base + y0 * (long)Wi + x1, g * (1.f - ly) * lx);
        atomicAdd(base + y1 * (long)Wi + x0, g * ly * (1.f - lx));
        atomicAdd(base + y1 * (long)Wi + x1, g * ly * lx);
    }
}

// inverse of pixel_shuffle_kernel: dg[(f,y,x), q*Co + co] = dout[(f, 2y+dy, 2x+dx), co]
__global__ __launch_bounds__(256) void pixel_shuffle_bwd_kernel(const unsigned short* __restrict__ dout, unsigned short* __restrict__ dg, long F, int H, int W,
                                                                int Co) {
    const int nch = Co / 8;
    const long total = F * (2 * H) * (2 * W) * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long t = i / nch;
        const int X = (int)(t % (2 * W)), Y = (int)((t / (2 * W)) % (2 * H));
        const long f = t / ((long)4 * W * H);
        const long src = (f * H + Y / 2) * (long)W + X / 2;
        const int q = (Y & 1) * 2 + (X & 1);
        *(u32x4*)(dg + src * (4L * Co) + q * Co + ch * 8) = *(const u32x4*)(dout + t * Co + ch * 8);
    }
}

// dlogits of  cb * sum_n mean_hw BCE(x, t) + cd * sum_n dice_n  given the per-mask forward sums {bce, sum p t, sum p, sum t}
__global__ __launch_bounds__(256) void bce_dice_grad_kernel(const float* __restrict__ x, const float* __restrict__ tg, const float* __restrict__ sums,
                                                            float* __restrict__ dx, long n_masks, long hw, float cb, float cd,
                                                            const float* __restrict__ cbp, const float* __restrict__ cdp) {
    if (cbp) cb *= cbp[0];   // upstream gradients of the two loss sums, read on the device (no device -> host sync in backward)
    if (cdp) cd *= cdp[0];
    const long total = n_masks * hw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / hw;
        const float v = x[i], t = tg[i];
        const float p = 1.f / (1.f + __expf(-v));
        const float num = 2.f * sums[n * 4 + 1] / 1000.f + 1e-6f;
        const float den = sums[n * 4 + 2] / 1000.f + sums[n * 4 + 3] / 1000.f + 1e-6f;
        // d dice / d p_i = -( (2 t_i / 1000) * den - num / 1000 ) / den^2
        const float ddice = -((2.f * t / 1000.f) * den - num / 1000.f) / (den * den);
        dx[i] = cb * (p - t) / (float)hw + cd * ddice * p * (1.f - p);
    }
}

static inline unsigned gd(long total, long cap = 256L * 32) {
    long b = cdiv(total, 256);
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace rga3

using namespace rga3;
typedef const unsigned short* cus;
typedef unsigned short* us;

extern "C" int rga3_layernorm_bwd(const void* x, const void* weight, const void* dy, void* dx, float* dweight, float* dbias, int64_t rows, int64_t dim,
                                  float eps, void* stream) {
    RGA3_CHECK_ARG(x && weight && dy && dx && rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 2048, "layernorm_bwd: bad args (dim <= 2048)");
    RGA3_CHECK_ARG((dweight == nullptr) == (dbias == nullptr), "layernorm_bwd: dweight and dbias go together");
    hipStream_t st = (hipStream_t)stream;
    if (dim == 16 || dim == 32 || dim == 64 || dim == 128) {   // narrow rows: several rows per wave
        const int rpn = 1024;
        dim3 gn((unsigned)cdiv(rows, rpn));
#define RGA3_LNB(G) hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<G>, gn, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (us)dx, dweight, dbias, (long)rows, eps, rpn)
        if (dim == 16) RGA3_LNB(2);
        else if (dim == 32) RGA3_LNB(4);
        else if (dim == 64) RGA3_LNB(8);
        else RGA3_LNB(16);
#undef RGA3_LNB
        RGA3_CHECK_LAUNCH("layernorm_bwd_narrow");
        return 0;
    }
    const int rpw = 64;
    dim3 grid((unsigned)cdiv(rows, rpw));
    if (dim <= 512) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (us)dx, dweight, dbias, (long)rows, (int)dim, eps, rpw);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (us)dx, dweight, dbias, (long)rows, (int)dim, eps, rpw);
    RGA3_CHECK_LAUNCH("layernorm_bwd");
    return 0;
}

extern "C" int rga3_colsum_accum(const void* x, float* out, int64_t rows, int64_t cols, int64_t ld, void* stream) {
    RGA3_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum_accum: bad args");
    const int rpw = 256;
    RGA3_CHECK_ARG(cdiv(rows, rpw) <= 65535, "colsum_accum: too many rows");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(cols, 256), (unsigned)cdiv(rows, rpw)), dim3(256), 0, (hipStream_t)stream, (cus)x, out, (long)rows,
                       (long)cols, (long)ld, rpw);
    RGA3_CHECK_LAUNCH("colsum_accum");
    return 0;
}

extern "C" int rga3_act(const void* a, const void* dy, void* out, int64_t n, int kind, void* stream) {
    RGA3_CHECK_ARG(a && out && n > 0 && kind >= 0 && kind <= 2 && (kind == 0 || dy), "act: bad args");
    hipLaunchKernelGGL(act_kernel, dim3(gd(n)), dim3(256), 0, (hipStream_t)stream, (cus)a, (cus)dy, (us)out, (long)n, kind);
    RGA3_CHECK_LAUNCH("act");
    return 0;
}

extern "C" int rga3_bilinear_bwd(const float* dout, float* din, const int32_t* plane_idx, int64_t N, int Hi, int Wi, int Ho, int Wo, void* stream) {
    RGA3_CHECK_ARG(dout && din && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bilinear_bwd: bad args");
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(gd(N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, dout, din, plane_idx, (long)N, Hi, Wi, Ho, Wo);
    RGA3_CHECK_LAUNCH("bilinear_bwd");
    return 0;
}

extern "C" int rga3_pixel_shuffle2x_bwd(const void* dout, void* dg, int64_t F, int H, int W, int Co, void* stream) {
    RGA3_CHECK_ARG(dout && dg && F > 0 && Co % 8 == 0, "pixel_shuffle2x_bwd: bad args");
    hipLaunchKernelGGL(pixel_shuffle_bwd_kernel, dim3(gd(F * 4L * H * W * (Co / 8))), dim3(256), 0, (hipStream_t)stream, (cus)dout, (us)dg, (long)F, H, W, Co);
    RGA3_CHECK_LAUNCH("pixel_shuffle2x_bwd");
    return 0;
}

extern "C" int rga3_bce_dice_grad(const float* logits, const float* targets, const float* sums4, float* dlogits, int64_t n_masks, int64_t hw,
                                  float coef_bce, float coef_dice, void* stream) {
    RGA3_CHECK_ARG(logits && targets && sums4 && dlogits && n_masks > 0 && hw > 0, "bce_dice_grad: bad args");
    hipLaunchKernelGGL(bce_dice_grad_kernel, dim3(gd(n_masks * hw)), dim3(256), 0, (hipStream_t)stream, logits, targets, sums4, dlogits, (long)n_masks,
                       (long)hw, coef_bce, coef_dice, (const float*)nullptr, (const float*)nullptr);
    RGA3_CHECK_LAUNCH("bce_dice_grad");
    return 0;
}

extern "C" int rga3_bce_dice_grad_dev(const float* logits, const float* targets, const float* sums4, float* dlogits, int64_t n_masks, int64_t hw,
                                      const float* coef_bce_dev, const float* coef_dice_dev, void* stream) {
    RGA3_CHECK_ARG(logits && targets && sums4 && dlogits && coef_bce_dev && coef_dice_dev && n_masks > 0 && hw > 0, "bce_dice_grad_dev: bad args");
    hipLaunchKernelGGL(bce_dice_grad_kernel, dim3(gd(n_masks * hw)), dim3(256), 0, (hipStream_t)stream, logits, targets, sums4, dlogits, (long)n_masks,
                       (long)hw, 1.f, 1.f, coef_bce_dev, coef_dice_dev);
    RGA3_CHECK_LAUNCH("bce_dice_grad_dev");
    return 0;
}
