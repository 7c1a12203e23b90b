// HBM-bound row kernels: RMSNorm / LayerNorm, rotary embedding, row gather/scatter, padding, elementwise, CE.
// All are 16-byte-per-lane vectorised, fp32 math, bf16 storage (cdna_hip_programming.md Guideline 13).
#include "common.h"

namespace rga3 {

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}

// ------------------------------------------------------------------------------------------------ RMSNorm
// one wave per row; the row stays in registers between the statistics pass and the scale pass.
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ add,
                                                      const unsigned short* __restrict__ w, unsigned short* __restrict__ y,
                                                      unsigned short* __restrict__ res_out, long rows, int dim, long ldx, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = dim / 8;
    u32x4 buf[MAXC], abuf[MAXC];
    float ss = 0.f;
    // all loads first (16 B per lane each, MAXC of them in flight), then chunk-by-chunk arithmetic behind scheduling barriers: the
    // unpacked floats of one chunk die before the next chunk is touched (free scheduling kept all of them live: 235 VGPRs at MAXC = 8)
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            buf[i] = *(const u32x4*)(x + row * ldx + ch * 8);
            if (add) abuf[i] = *(const u32x4*)(add + row * ldx + ch * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            u32x4 v = buf[i];
            float f[8];
            unpack8(v, f);
            if (add) {
                float fa[8];
                unpack8(abuf[i], fa);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bf2f(f2bf(f[e] + fa[e]));  // bf16 residual stream
                v = pack8(f);
                if (res_out) *(u32x4*)(res_out + row * ldx + ch * 8) = v;
                buf[i] = v;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    ss = wave_sum(ss);
    const float rinv = rsqrtf(ss / (float)dim + eps);
    // keep only the PACKED row between the two passes: left alone the compiler carries the 8 unpacked floats of every chunk across
    // the reduction (235 VGPRs for MAXC = 8, two waves per SIMD)
#pragma unroll
    for (int i = 0; i < MAXC; ++i) asm volatile("" : "+v"(buf[i]));
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            float f[8], fw[8];
            unpack8(buf[i], f);
            unpack8(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fw[e] * bf2f(f2bf(f[e] * rinv));  // HF: weight * normed.to(bf16)
            *(u32x4*)(y + row * ldx + ch * 8) = pack8(f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Wide rows (decoder hidden 3584 = 448 chunks): one 256-thread block per row, <= MAXC chunks per thread, cross-wave sum through LDS.
// The wave-per-row form needs 7 chunks (+ 7 of the residual) and as many 64-bit addresses per lane there: 220 VGPRs, 2 waves / SIMD.
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_block_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ add,
                                                            const unsigned short* __restrict__ w, unsigned short* __restrict__ y,
                                                            unsigned short* __restrict__ res_out, long rows, int dim, long ldx, float eps) {
    __shared__ float part[4];
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const int nch = dim / 8;
    u32x4 buf[MAXC], wv[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = tid + i * 256;
        if (ch < nch) {
            u32x4 v = *(const u32x4*)(x + row * ldx + ch * 8);
            wv[i] = *(const u32x4*)(w + ch * 8);
            float f[8];
            unpack8(v, f);
            if (add) {
                u32x4 a = *(const u32x4*)(add + row * ldx + ch * 8);
                float fa[8];
                unpack8(a, fa);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bf2f(f2bf(f[e] + fa[e]));  // bf16 residual stream
                v = pack8(f);
                if (res_out) *(u32x4*)(res_out + row * ldx + ch * 8) = v;
            }
            buf[i] = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    ss = part[0] + part[1] + part[2] + part[3];
    const float rinv = rsqrtf(ss / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = tid + i * 256;
        if (ch < nch) {
            float f[8], fw[8];
            unpack8(buf[i], f);
            unpack8(wv[i], fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fw[e] * bf2f(f2bf(f[e] * rinv));  // HF: weight * normed.to(bf16)
            *(u32x4*)(y + row * ldx + ch * 8) = pack8(f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                        const unsigned short* __restrict__ b, unsigned short* __restrict__ y,
                                                        long rows, int dim, long ldx, long ldy, float eps, int act, float* __restrict__ stats = nullptr) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = dim / 8;
    u32x4 buf[MAXC];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            buf[i] = *(const u32x4*)(x + row * ldx + ch * 8);
            float f[8];
            unpack8(buf[i], f);
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
            unpack8(buf[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = f[e] - mean; s2 += d * d; }
        }
    }
    const float rinv = rsqrtf(wave_sum(s2) / (float)dim + eps);
    if (stats) {   // statistics only (the normalisation itself is folded into the consuming product: rga3_gemm_ln_bf16)
        if (lane == 0) *(float2*)(stats + 2 * row) = make_float2(mean, rinv);
        return;
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            float f[8], fw[8], fb[8];
            unpack8(buf[i], f);
            unpack8(*(const u32x4*)(w + ch * 8), fw);
            if (b) unpack8(*(const u32x4*)(b + ch * 8), fb);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[e] = (f[e] - mean) * rinv * fw[e] + (b ? fb[e] : 0.f);
                if (act == 1) { float t = bf2f(f2bf(f[e])); f[e] = 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); }
            }
            *(u32x4*)(y + row * ldy + ch * 8) = pack8(f);
        }
    }
}

// Narrow rows (dim <= 512): LPR lanes per row, 64 / LPR rows per wave, at most two 16-byte chunks per lane -- Hiera-L's stage-1 / stage-2
// rows (144 / 288 channels: 18 / 36 chunks) keep 28 % / 56 % of a wave busy in the one-row-per-wave kernel above.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                             const unsigned short* __restrict__ b, unsigned short* __restrict__ y,
                                                             long rows, int dim, long ldx, long ldy, float eps, int act, float* __restrict__ stats = nullptr) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    const int nch = dim / 8;
    u32x4 buf[2];
    float f[2][8];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = sub + i * LPR;
        buf[i] = u32x4{0u, 0u, 0u, 0u};
        if (live && ch < nch) buf[i] = *(const u32x4*)(x + row * ldx + ch * 8);
        unpack8(buf[i], f[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += f[i][e];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 / (float)dim;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = sub + i * LPR;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; s2 += d * d; }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rinv = rsqrtf(s2 / (float)dim + eps);
    if (stats) {
        if (live && sub == 0) *(float2*)(stats + 2 * row) = make_float2(mean, rinv);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = sub + i * LPR;
        if (live && ch < nch) {
            float fw[8], fb[8], o8[8];
            unpack8(*(const u32x4*)(w + ch * 8), fw);
            if (b) unpack8(*(const u32x4*)(b + ch * 8), fb);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (f[i][e] - mean) * rinv * fw[e] + (b ? fb[e] : 0.f);
                if (act == 1) { float t = bf2f(f2bf(v)); v = 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); }
                o8[e] = v;
            }
            *(u32x4*)(y + row * ldy + ch * 8) = pack8(o8);
        }
    }
}

// Row statistics only (rga3_layernorm_stats), for the widths the one-row-per-wave kernel serves badly: at 576 channels (Hiera-L stage 3: 72 sixteen-byte chunks) a
// wave issued ONE full load and one eighth-full load per row and then ran two 6-step wave reductions -- 37.7 MB per 8 frames in 23.8 us = 1.6 TB/s, latency bound,
// ~160 launches per training step (profiles/r04_train_step_timeline.txt).  Here LPR lanes share a row and every lane has its NCHL chunks in flight before the first
// use: 8 lanes x 9 chunks (576 channels, 8 rows per wave), 16 x 9 (1152 channels, 4 rows per wave); the reductions are 3 / 4 shuffle steps.  Same two-pass
// arithmetic (mean, then centred squares) on the register-resident row.
template <int LPR, int NCHL>
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const unsigned short* __restrict__ x, float* __restrict__ stats, long rows, int dim, long ldx, float eps) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const long rowc = row < rows ? row : rows - 1;
    const int nch = dim / 8;
    u32x4 buf[NCHL];
#pragma unroll
    for (int i = 0; i < NCHL; ++i) {
        const int ch = sub + i * LPR;
        buf[i] = (ch < nch) ? *(const u32x4*)(x + rowc * ldx + ch * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NCHL; ++i) {
        float f[8];
        unpack8(buf[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += f[e];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 / (float)dim;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCHL; ++i) {
        if (sub + i * LPR < nch) {
            float f[8];
            unpack8(buf[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; s2 += d * d; }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    if (row < rows && sub == 0) *(float2*)(stats + 2 * row) = make_float2(mean, rsqrtf(s2 / (float)dim + eps));
}

// Narrow rows (dim < 8 or not a multiple of 8, <= 16): one thread per row, scalar bf16 loads.  The 4-channel LayerNorm2d + GELU of the
// memory encoder's first mask-downsampler stage (reference model/sam2.py:611-643) runs 4 x 512 x 512 rows per frame through this.
__global__ __launch_bounds__(256) void layernorm_tiny_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                             const unsigned short* __restrict__ b, unsigned short* __restrict__ y, long rows, int dim, long ldx,
                                                             long ldy, float eps, int act) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float f[16];
    float s1 = 0.f;
    for (int e = 0; e < dim; ++e) { f[e] = bf2f(x[row * ldx + e]); s1 += f[e]; }
    const float mean = s1 / (float)dim;
    float s2 = 0.f;
    for (int e = 0; e < dim; ++e) { const float d = f[e] - mean; s2 += d * d; }
    const float rinv = rsqrtf(s2 / (float)dim + eps);
    for (int e = 0; e < dim; ++e) {
        float v = (f[e] - mean) * rinv * bf2f(w[e]) + (b ? bf2f(b[e]) : 0.f);
        if (act == 1) { const float t = bf2f(f2bf(v)); v = 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); }
        y[row * ldy + e] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------ rotary
// thread = (token, group of HG heads, 8-wide chunk of the first half): the cos / sin values (4 x 8 floats, the bulk of this kernel's
// load instructions when fetched per head) are read once and reused for the HG heads; each head rotates the chunk and its partner
// in the second half.
template <int HG>
__global__ __launch_bounds__(256) void rope_kernel(unsigned short* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ sn,
                                                   long T, int h0, int nh, int D, long st, long sh) {
    const int half = D / 2, cph = half / 8;
    const int ngrp = nh / HG;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = T * ngrp * cph;
    if (idx >= total) return;
    const int ch = (int)(idx % cph);
    const int hg = (int)((idx / cph) % ngrp);
    const long t = idx / ((long)cph * ngrp);
    float c1[8], s1[8], c2[8], s2[8];
    {
        const float* cp = cs + t * D + ch * 8;
        const float* sp = sn + t * D + ch * 8;
        *(f32x4*)&c1[0] = *(const f32x4*)cp;        *(f32x4*)&c1[4] = *(const f32x4*)(cp + 4);
        *(f32x4*)&s1[0] = *(const f32x4*)sp;        *(f32x4*)&s1[4] = *(const f32x4*)(sp + 4);
        *(f32x4*)&c2[0] = *(const f32x4*)(cp + half); *(f32x4*)&c2[4] = *(const f32x4*)(cp + half + 4);
        *(f32x4*)&s2[0] = *(const f32x4*)(sp + half); *(f32x4*)&s2[4] = *(const f32x4*)(sp + half + 4);
    }
    unsigned short* p1 = x + t * st + (long)(h0 + hg * HG) * sh + ch * 8;
    u32x4 va[HG], vb[HG];
#pragma unroll
    for (int h = 0; h < HG; ++h) {
        va[h] = *(const u32x4*)(p1 + h * sh);
        vb[h] = *(const u32x4*)(p1 + h * sh + half);
    }
#pragma unroll
    for (int h = 0; h < HG; ++h) {
        float a[8], b[8], o1[8], o2[8];
        unpack8(va[h], a);
        unpack8(vb[h], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o1[e] = a[e] * c1[e] - b[e] * s1[e];   // x*cos + rotate_half(x)*sin, first half: -x2
            o2[e] = b[e] * c2[e] + a[e] * s2[e];   // second half: +x1
        }
        *(u32x4*)(p1 + h * sh) = pack8(o1);
        *(u32x4*)(p1 + h * sh + half) = pack8(o2);
    }
}

// ------------------------------------------------------------------------------------------------ gather / scatter rows
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(const unsigned short* __restrict__ src, const long* __restrict__ idx,
                                                        unsigned short* __restrict__ dst, long n_idx, long rpi, int dim, long ld_src,
                                                        long ld_dst) {
    const int nch = dim / 8;
    const long total = n_idx * rpi * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long r = i / nch;          // linear row in the "dense" side
        const long u = r / rpi, w = r % rpi;
        const long other = idx[u] * rpi + w;  // row on the indexed side
        const long srow = SCATTER ? r : other, drow = SCATTER ? other : r;
        *(u32x4*)(dst + drow * ld_dst + ch * 8) = *(const u32x4*)(src + srow * ld_src + ch * 8);
    }
}

__global__ __launch_bounds__(256) void pad_cols_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, long rows,
                                                       int cols, long ld_src, long ld_dst) {
    const int nch = (int)(ld_dst / 8);
    const long total = rows * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long r = i / nch;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ch * 8 + 8 <= cols && (ld_src & 7) == 0) {
            v = *(const u32x4*)(src + r * ld_src + ch * 8);
        } else if (ch * 8 < cols) {
            unsigned short tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int e = 0; e < 8 && ch * 8 + e < cols; ++e) tmp[e] = src[r * ld_src + ch * 8 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (unsigned)tmp[2 * e] | ((unsigned)tmp[2 * e + 1] << 16);
        }
        *(u32x4*)(dst + r * ld_dst + ch * 8) = v;
    }
}

template <int OP>  // 0: silu(a)*b   1: a+b
__global__ __launch_bounds__(256) void ew2_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                  unsigned short* __restrict__ o, long n) {
    const long nv = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float fa[8], fb[8];
        unpack8(*(const u32x4*)(a + i * 8), fa);
        unpack8(*(const u32x4*)(b + i * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (OP == 0) fa[e] = bf2f(f2bf(fa[e] / (1.f + __expf(-fa[e])))) * fb[e];
            else fa[e] = fa[e] + fb[e];
        }
        *(u32x4*)(o + i * 8) = pack8(fa);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (long i = nv * 8; i < n; ++i) {
            float x = bf2f(a[i]), y = bf2f(b[i]);
            o[i] = f2bf(OP == 0 ? bf2f(f2bf(x / (1.f + __expf(-x)))) * y : x + y);
        }
    }
}

// ------------------------------------------------------------------------------------------------ cross entropy
// one 256-thread block per row: online max / sum-exp in fp32, then optional gradient pass.
template <bool F32>
__global__ __launch_bounds__(256) void ce_rows_kernel(const void* __restrict__ logits, const long* __restrict__ labels, float* __restrict__ row_loss,
                                                      unsigned short* __restrict__ dlogits, long V, long ld, float gscale) {
    __shared__ float red_m[4], red_s[4];
    const long row = blockIdx.x;
    const long lab = labels[row];
    const int tid = threadIdx.x;
    if (lab < 0) {  // ignore_index
        if (tid == 0) row_loss[row] = 0.f;
        if (dlogits)
            for (long i = tid; i < V; i += 256) dlogits[row * ld + i] = 0;
        return;
    }
    auto ld1 = [&](long i) -> float {
        if (F32) return ((const float*)logits)[row * ld + i];
        return bf2f(((const unsigned short*)logits)[row * ld + i]);
    };
    // three passes over the L2-resident row (max, sum of exp, gradient), 16-byte loads when the row allows: the one-pass online form carried a dependent
    // exp chain per 2-byte load -- 305 us for the 6 labelled rows x 152 064 logits of a training step
    constexpr int EPV = F32 ? 4 : 8;
    const bool vec = (ld % EPV == 0) && ((((uintptr_t)logits) & 15) == 0);
    const long nv = vec ? V / EPV : 0;
    auto ldv = [&](long j, float* f) {
        if constexpr (F32) {
            const f32x4 v = *(const f32x4*)((const float*)logits + row * ld + j * 4);
            f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
        } else {
            unpack8(*(const u32x4*)((const unsigned short*)logits + row * ld + j * 8), f);
        }
    };
    float m = -INFINITY;
    for (long j = tid; j < nv; j += 256) {
        float f[EPV];
        ldv(j, f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) m = fmaxf(m, f[e]);
    }
    for (long i = nv * EPV + tid; i < V; i += 256) m = fmaxf(m, ld1(i));
    m = wave_max(m);
    if ((tid & 63) == 0) red_m[tid >> 6] = m;
    __syncthreads();
    const float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float s = 0.f;
    for (long j = tid; j < nv; j += 256) {
        float f[EPV];
        ldv(j, f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) s += __expf(f[e] - M);
    }
    for (long i = nv * EPV + tid; i < V; i += 256) s += __expf(ld1(i) - M);
    s = wave_sum(s);
    if ((tid & 63) == 0) red_s[tid >> 6] = s;
    __syncthreads();
    const float S = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    const float lse = M + logf(S);
    if (tid == 0) row_loss[row] = lse - ld1(lab);
    if (dlogits) {
        const bool vout = vec && (ld % 8 == 0) && ((((uintptr_t)dlogits) & 15) == 0) && !F32;
        const long nvo = vout ? nv : 0;
        for (long j = tid; j < nvo; j += 256) {
            float f[8];
            unpack8(*(const u32x4*)((const unsigned short*)logits + row * ld + j * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (__expf(f[e] - lse) - ((j * 8 + e) == lab ? 1.f : 0.f)) * gscale;
            *(u32x4*)(dlogits + row * ld + j * 8) = pack8(f);
        }
        for (long i = nvo * 8 + tid; i < V; i += 256) {
            float pr = __expf(ld1(i) - lse);
            dlogits[row * ld + i] = f2bf((pr - (i == lab ? 1.f : 0.f)) * gscale);
        }
    }
}

static inline unsigned grid_for(long total, int per_block = 256, long cap = 256L * 8 * 4) {
    long b = cdiv(total, per_block);
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace rga3

using namespace rga3;

extern "C" int rga3_rmsnorm_fwd(const void* x, const void* add, const void* weight, void* y, void* res_out, int64_t rows,
                                int64_t dim, int64_t ldx, float eps, void* stream) {
    RGA3_CHECK_ARG(x && weight && y, "rmsnorm: null pointer");
    RGA3_CHECK_ARG(rows > 0 && dim > 0 && dim % 8 == 0 && ldx % 8 == 0 && dim <= 8192, "rmsnorm: rows=%ld dim=%ld ldx=%ld", (long)rows, (long)dim, (long)ldx);
    RGA3_CHECK_ARG(!res_out || add, "rmsnorm: res_out requires add");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(rows, 4));
    const unsigned short *xp = (const unsigned short*)x, *ap = (const unsigned short*)add, *wp = (const unsigned short*)weight;
    unsigned short *yp = (unsigned short*)y, *rp = (unsigned short*)res_out;
    if (dim >= 2048 && dim <= 256 * 8 * 4) {  // wide rows: a block per row
        const int nb = (int)cdiv(dim / 8, 256);
        dim3 g2((unsigned)rows);
        if (nb <= 2) hipLaunchKernelGGL(rmsnorm_block_kernel<2>, g2, dim3(256), 0, st, xp, ap, wp, yp, rp, (long)rows, (int)dim, (long)ldx, eps);
        else hipLaunchKernelGGL(rmsnorm_block_kernel<4>, g2, dim3(256), 0, st, xp, ap, wp, yp, rp, (long)rows, (int)dim, (long)ldx, eps);
        RGA3_CHECK_LAUNCH("rmsnorm_block_kernel");
        return 0;
    }
    const int nslot = (int)cdiv(dim / 8, 64);  // 16-byte chunks per lane
#define RGA3_RMS(N) hipLaunchKernelGGL(rmsnorm_kernel<N>, grid, dim3(256), 0, st, xp, ap, wp, yp, rp, (long)rows, (int)dim, (long)ldx, eps)
    switch (nslot) {
        case 1: RGA3_RMS(1); break;
        case 2: RGA3_RMS(2); break;
        case 3: RGA3_RMS(3); break;
        case 4: RGA3_RMS(4); break;
        case 5: RGA3_RMS(5); break;
        case 6: RGA3_RMS(6); break;
        case 7: RGA3_RMS(7); break;
        case 8: RGA3_RMS(8); break;
        default: RGA3_RMS(16); break;
    }
#undef RGA3_RMS
    RGA3_CHECK_LAUNCH("rmsnorm_kernel");
    return 0;
}

extern "C" int rga3_layernorm_fwd(const void* x, const void* weight, const void* bias, void* y, int64_t rows, int64_t dim,
                                  int64_t ldx, int64_t ldy, float eps, int act, void* stream) {
    RGA3_CHECK_ARG(x && weight && y, "layernorm: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (rows > 0 && dim > 0 && dim <= 16 && (dim % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0)) {
        hipLaunchKernelGGL(layernorm_tiny_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0, st, (const unsigned short*)x, (const unsigned short*)weight,
                           (const unsigned short*)bias, (unsigned short*)y, (long)rows, (int)dim, (long)ldx, (long)ldy, eps, act);
        RGA3_CHECK_LAUNCH("layernorm_tiny_kernel");
        return 0;
    }
    RGA3_CHECK_ARG(rows > 0 && dim > 0 && dim % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && dim <= 8192, "layernorm: rows=%ld dim=%ld", (long)rows, (long)dim);
    dim3 grid((unsigned)cdiv(rows, 4));
    const unsigned short *xp = (const unsigned short*)x, *wp = (const unsigned short*)weight, *bp = (const unsigned short*)bias;
    unsigned short* yp = (unsigned short*)y;
    if (dim <= 16 * 8 * 2) {          // <= 32 chunks: 16 lanes per row, 4 rows per wave
        hipLaunchKernelGGL(layernorm_rows_kernel<16>, dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, st, xp, wp, bp, yp, (long)rows, (int)dim, (long)ldx, (long)ldy, eps, act, (float*)nullptr);
    } else if (dim <= 32 * 8 * 2) {   // <= 64 chunks: 32 lanes per row, 2 rows per wave
        hipLaunchKernelGGL(layernorm_rows_kernel<32>, dim3((unsigned)cdiv(rows, 8)), dim3(256), 0, st, xp, wp, bp, yp, (long)rows, (int)dim, (long)ldx, (long)ldy, eps, act, (float*)nullptr);
    } else if (dim <= 64 * 8 * 4) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, xp, wp, bp, yp, (long)rows, (int)dim, (long)ldx, (long)ldy, eps, act, (float*)nullptr);
    else if (dim <= 64 * 8 * 8) hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, st, xp, wp, bp, yp, (long)rows, (int)dim, (long)ldx, (long)ldy, eps, act, (float*)nullptr);
    else hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, st, xp, wp, bp, yp, (long)rows, (int)dim, (long)ldx, (long)ldy, eps, act, (float*)nullptr);
    RGA3_CHECK_LAUNCH("layernorm_kernel");
    return 0;
}

// stats [rows][2] f32 = (mean, 1 / sqrt(var + eps)) of every row of x (biased variance, two passes over the register-resident row: LayerNorm's own statistics)
extern "C" int rga3_layernorm_stats(const void* x, float* stats, int64_t rows, int64_t dim, int64_t ldx, float eps, void* stream) {
    RGA3_CHECK_ARG(x && stats && rows > 0 && dim > 0 && dim % 8 == 0 && ldx % 8 == 0 && dim <= 8192, "layernorm_stats: rows=%ld dim=%ld", (long)rows, (long)dim);
    RGA3_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)stats) & 7) == 0, "layernorm_stats: pointer alignment");
    hipStream_t st = (hipStream_t)stream;
    const unsigned short* xp = (const unsigned short*)x;
    const unsigned short* nul = nullptr;
    dim3 grid((unsigned)cdiv(rows, 4));
    if (dim <= 16 * 8 * 2) hipLaunchKernelGGL(layernorm_rows_kernel<16>, dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, st, xp, nul, nul, (unsigned short*)nullptr, (long)rows, (int)dim, (long)ldx, 0L, eps, 0, stats);
    else if (dim <= 32 * 8 * 2) hipLaunchKernelGGL(layernorm_rows_kernel<32>, dim3((unsigned)cdiv(rows, 8)), dim3(256), 0, st, xp, nul, nul, (unsigned short*)nullptr, (long)rows, (int)dim, (long)ldx, 0L, eps, 0, stats);
    else if (dim <= 8 * 9 * 8) hipLaunchKernelGGL((layernorm_stats_kernel<8, 9>), dim3((unsigned)cdiv(rows, 32)), dim3(256), 0, st, xp, stats, (long)rows, (int)dim, (long)ldx, eps);
    else if (dim <= 16 * 9 * 8) hipLaunchKernelGGL((layernorm_stats_kernel<16, 9>), dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, st, xp, stats, (long)rows, (int)dim, (long)ldx, eps);
    else if (dim <= 64 * 8 * 4) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, xp, nul, nul, (unsigned short*)nullptr, (long)rows, (int)dim, (long)ldx, 0L, eps, 0, stats);
    else if (dim <= 64 * 8 * 8) hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, st, xp, nul, nul, (unsigned short*)nullptr, (long)rows, (int)dim, (long)ldx, 0L, eps, 0, stats);
    else hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, st, xp, nul, nul, (unsigned short*)nullptr, (long)rows, (int)dim, (long)ldx, 0L, eps, 0, stats);
    RGA3_CHECK_LAUNCH("layernorm_stats");
    return 0;
}

extern "C" int rga3_rope_inplace(void* x, const float* cos, const float* sin, int64_t T, int h0, int nh, int D, int64_t st,
                                 int64_t sh, void* stream) {
    RGA3_CHECK_ARG(x && cos && sin, "rope: null pointer");
    RGA3_CHECK_ARG(T > 0 && nh > 0 && D > 0 && D % 16 == 0 && st % 8 == 0 && sh % 8 == 0, "rope: T=%ld nh=%d D=%d", (long)T, nh, D);
    RGA3_CHECK_ARG((((uintptr_t)cos | (uintptr_t)sin) & 15) == 0, "rope: cos/sin tables must be 16-byte aligned");
    if (nh % 4 == 0) {
        const long total = (long)T * (nh / 4) * (D / 16);
        hipLaunchKernelGGL(rope_kernel<4>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned short*)x, cos, sin, (long)T,
                           h0, nh, D, (long)st, (long)sh);
    } else {
        const long total = (long)T * nh * (D / 16);
        hipLaunchKernelGGL(rope_kernel<1>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned short*)x, cos, sin, (long)T,
                           h0, nh, D, (long)st, (long)sh);
    }
    RGA3_CHECK_LAUNCH("rope_kernel");
    return 0;
}

extern "C" int rga3_gather_rows(const void* table, const int64_t* idx, void* out, int64_t n_idx, int64_t rows_per_idx,
                                int64_t dim, int64_t ld_table, int64_t ld_out, void* stream) {
    RGA3_CHECK_ARG(table && idx && out, "gather_rows: null pointer");
    RGA3_CHECK_ARG(n_idx > 0 && rows_per_idx > 0 && dim > 0 && dim % 8 == 0 && ld_table % 8 == 0 && ld_out % 8 == 0, "gather_rows: bad shape");
    const long total = n_idx * rows_per_idx * (dim / 8);
    hipLaunchKernelGGL(move_rows_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)table,
                       (const long*)idx, (unsigned short*)out, (long)n_idx, (long)rows_per_idx, (int)dim, (long)ld_table, (long)ld_out);
    RGA3_CHECK_LAUNCH("gather_rows");
    return 0;
}

extern "C" int rga3_scatter_rows(const void* src, const int64_t* idx, void* out, int64_t n_idx, int64_t rows_per_idx, int64_t dim,
                                 int64_t ld_src, int64_t ld_out, void* stream) {
    RGA3_CHECK_ARG(src && idx && out, "scatter_rows: null pointer");
    RGA3_CHECK_ARG(n_idx > 0 && rows_per_idx > 0 && dim > 0 && dim % 8 == 0 && ld_src % 8 == 0 && ld_out % 8 == 0, "scatter_rows: bad shape");
    const long total = n_idx * rows_per_idx * (dim / 8);
    hipLaunchKernelGGL(move_rows_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src,
                       (const long*)idx, (unsigned short*)out, (long)n_idx, (long)rows_per_idx, (int)dim, (long)ld_src, (long)ld_out);
    RGA3_CHECK_LAUNCH("scatter_rows");
    return 0;
}

extern "C" int rga3_pad_cols(const void* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, void* stream) {
    RGA3_CHECK_ARG(src && dst, "pad_cols: null pointer");
    RGA3_CHECK_ARG(rows > 0 && cols > 0 && ld_dst >= cols && ld_dst % 8 == 0, "pad_cols: bad shape");
    hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for(rows * (ld_dst / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src,
                       (unsigned short*)dst, (long)rows, (int)cols, (long)ld_src, (long)ld_dst);
    RGA3_CHECK_LAUNCH("pad_cols");
    return 0;
}

extern "C" int rga3_silu_mul(const void* a, const void* b, void* out, int64_t n, void* stream) {
    RGA3_CHECK_ARG(a && b && out && n > 0, "silu_mul: bad args");
    hipLaunchKernelGGL(ew2_kernel<0>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                       (const unsigned short*)b, (unsigned short*)out, (long)n);
    RGA3_CHECK_LAUNCH("silu_mul");
    return 0;
}

extern "C" int rga3_add(const void* a, const void* b, void* out, int64_t n, void* stream) {
    RGA3_CHECK_ARG(a && b && out && n > 0, "add: bad args");
    hipLaunchKernelGGL(ew2_kernel<1>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a,
                       (const unsigned short*)b, (unsigned short*)out, (long)n);
    RGA3_CHECK_LAUNCH("add");
    return 0;
}

extern "C" int rga3_cross_entropy_rows(const void* logits, int logits_dtype, const int64_t* labels, float* row_loss,
                                       void* dlogits, int64_t rows, int64_t V, int64_t ld, float grad_scale, void* stream) {
    RGA3_CHECK_ARG(logits && labels && row_loss, "cross_entropy: null pointer");
    RGA3_CHECK_ARG(rows > 0 && V > 0 && ld >= V, "cross_entropy: bad shape");
    RGA3_CHECK_ARG(logits_dtype == RGA3_BF16 || logits_dtype == RGA3_F32, "cross_entropy: dtype");
    hipStream_t st = (hipStream_t)stream;
    if (logits_dtype == RGA3_F32)
        hipLaunchKernelGGL(ce_rows_kernel<true>, dim3((unsigned)rows), dim3(256), 0, st, logits, (const long*)labels, row_loss,
                           (unsigned short*)dlogits, (long)V, (long)ld, grad_scale);
    else
        hipLaunchKernelGGL(ce_rows_kernel<false>, dim3((unsigned)rows), dim3(256), 0, st, logits, (const long*)labels, row_loss,
                           (unsigned short*)dlogits, (long)V, (long)ld, grad_scale);
    RGA3_CHECK_LAUNCH("ce_rows_kernel");
    return 0;
}
