// Backward / optimiser kernels of the training step (HBM-bound, 16 B per lane, fp32 math):
// RMSNorm backward, SwiGLU backward on the interleaved gate/up layout, bf16 transpose (for dW = dY^T X through the NT
// GEMM), CSR row-segment sums (embedding gradient), fused AdamW.
#include "common.h"

namespace rga3 {

__device__ __forceinline__ void un8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pk8_(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}

// dx = r * g - x * r^3 * mean(g * x) (+ add), g = dy * w, r = rsqrt(mean(x^2) + eps); one wave per row.
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                          const unsigned short* __restrict__ dy, const unsigned short* __restrict__ add,
                                                          unsigned short* __restrict__ dx, long rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = dim / 8;
    u32x4 xb[MAXC], gb[MAXC];
    float ss = 0.f, sg = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            float fx[8], fd[8], fw[8];
            xb[i] = *(const u32x4*)(x + row * dim + ch * 8);
            un8(xb[i], fx);
            un8(*(const u32x4*)(dy + row * dim + ch * 8), fd);
            un8(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                fd[e] *= fw[e];
                ss += fx[e] * fx[e];
                sg += fd[e] * fx[e];
            }
            gb[i] = pk8_(fd);
        }
    }
    ss = wave_sum(ss);
    sg = wave_sum(sg);
    const float r = rsqrtf(ss / (float)dim + eps);
    const float coef = r * r * r * sg / (float)dim;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
        if (ch < nch) {
            float fx[8], fg[8], fa[8];
            un8(xb[i], fx);
            un8(gb[i], fg);
            if (add) un8(*(const u32x4*)(add + row * dim + ch * 8), fa);
#pragma unroll
            for (int e = 0; e < 8; ++e) fg[e] = r * fg[e] - fx[e] * coef + (add ? fa[e] : 0.f);
            *(u32x4*)(dx + row * dim + ch * 8) = pk8_(fg);
        }
    }
}

// Wide rows (decoder hidden size 3584: 448 chunks): one WORKGROUP per row, NC chunks per thread, fp32 kept in registers between the two passes.  The
// wave-per-row form above holds 7 chunks x (x, g) per lane and leaves 2 112 rows as 2 112 waves (8 per CU) with a three-phase latency chain each:
// 34 us per call against ~12 us of HBM time for x, dy, add, dx (55 calls per training step).
template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_block_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                                const unsigned short* __restrict__ dy, const unsigned short* __restrict__ add,
                                                                unsigned short* __restrict__ dx, int dim, float eps) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long row = blockIdx.x;
    const int nch = dim / 8;
    float fx[NC][8], fg[NC][8];
    float ss = 0.f, sg = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
            float fw[8];
            un8(*(const u32x4*)(x + row * dim + ch * 8), fx[i]);
            un8(*(const u32x4*)(dy + row * dim + ch * 8), fg[i]);
            un8(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                fg[i][e] *= fw[e];
                ss += fx[i][e] * fx[i][e];
                sg += fg[i][e] * fx[i][e];
            }
        }
    }
    ss = wave_sum(ss);
    sg = wave_sum(sg);
    if (lane == 0) { red[0][wv] = ss; red[1][wv] = sg; }
    __syncthreads();
    ss = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sg = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float r = rsqrtf(ss / (float)dim + eps);
    const float coef = r * r * r * sg / (float)dim;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
            float fa[8], o[8];
            if (add) un8(*(const u32x4*)(add + row * dim + ch * 8), fa);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r * fg[i][e] - fx[i][e] * coef + (add ? fa[e] : 0.f);
            *(u32x4*)(dx + row * dim + ch * 8) = pk8_(o);
        }
    }
}

// gu [T, 2I] pre-activations in 16-column blocks (gate | up), da [T, I]  ->  dgu [T, 2I] (same interleave)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const unsigned short* __restrict__ gu, const unsigned short* __restrict__ da,
                                                         unsigned short* __restrict__ dgu, long T, long I) {
    const long nch = I / 8;  // 8 outputs per thread (half of a 16-block)
    const long total = T * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long ch = i % nch, t = i / nch;
        const long blk = ch / 2, half = ch % 2;
        const long goff = t * 2 * I + blk * 32 + half * 8;
        float g[8], u[8], d[8], dg[8], du[8];
        un8(*(const u32x4*)(gu + goff), g);
        un8(*(const u32x4*)(gu + goff + 16), u);
        un8(*(const u32x4*)(da + t * I + ch * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) swiglu_bwd_elem(g[e], u[e], d[e], dg[e], du[e]);
        *(u32x4*)(dgu + goff) = pk8_(dg);
        *(u32x4*)(dgu + goff + 16) = pk8_(du);
    }
}

// a [T, I] = silu(gate) * up from gu [T, 2I] pre-activations in 16-column blocks (gate | up): the un-fused form of the GEMM's SwiGLU
// epilogue (same roundings: gate / up are bf16 already, silu rounded to bf16 before the product), used after the fp8 GEMM
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const unsigned short* __restrict__ gu, unsigned short* __restrict__ a, long T, long I) {
    const long nch = I / 8;
    const long total = T * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long ch = i % nch, t = i / nch;
        const long blk = ch / 2, half = ch % 2;
        const long goff = t * 2 * I + blk * 32 + half * 8;
        float g[8], u[8], o[8];
        un8(*(const u32x4*)(gu + goff), g);
        un8(*(const u32x4*)(gu + goff + 16), u);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = swiglu_fwd_elem(g[e], u[e]);
        *(u32x4*)(a + t * I + ch * 8) = pk8_(o);
    }
}

// out[c, r] = in[r, c] for 16-bit elements, 64x64 tiles through LDS (padded rows)
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, long R, long C,
                                                          long ldi, long ldo) {
    __shared__ unsigned short tile[64][66];
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const long r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[r * ldi + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const long c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[c * ldo + r] = tile[tx][i];
    }
}

// the same for up to 64 matrices in ONE launch (blockIdx.z = matrix; tiles beyond a matrix's extent exit): the mask path's backward needs W^T of ~46 small
// weight matrices per step -- as 46 launches at the ~5 us launch floor they were 0.4 ms of the step
constexpr int TR_MANY = 64;
struct TransposeMany {
    const unsigned short* in[TR_MANY];
    unsigned short* out[TR_MANY];
    int R[TR_MANY], C[TR_MANY];
};
__global__ __launch_bounds__(256) void transpose16_many_kernel(TransposeMany p) {
    __shared__ unsigned short tile[64][66];
    const int m = blockIdx.z;
    const long R = p.R[m], C = p.C[m];
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
    if (r0 >= R || c0 >= C) return;
    const unsigned short* in = p.in[m];
    unsigned short* out = p.out[m];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const long r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[r * C + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const long c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[c * R + r] = tile[tx][i];
    }
}

// out[u, :] = sum_{j in [off[u], off[u+1])} x[rows[j], :]   (fp32 accumulate, bf16 out); one block per output row
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const unsigned short* __restrict__ x, const long* __restrict__ rows,
                                                               const long* __restrict__ off, unsigned short* __restrict__ out, int dim, long ldx) {
    const long u = blockIdx.x;
    const long a = off[u], b = off[u + 1];
    for (int ch = threadIdx.x; ch < dim / 8; ch += 256) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long j = a; j < b; ++j) {
            float f[8];
            un8(*(const u32x4*)(x + rows[j] * ldx + ch * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
        *(u32x4*)(out + u * dim + ch * 8) = pk8_(acc);
    }
}

// AdamW on bf16 parameters with fp32 moments and an fp32 master copy (decoupled weight decay, bias-corrected)
__global__ __launch_bounds__(256) void adamw_kernel(unsigned short* __restrict__ p, float* __restrict__ master, const unsigned short* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gr = bf2f(g[i]) * gscale;
        float w = master[i];
        const float mi = b1 * m[i] + (1.f - b1) * gr;
        const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        w = w * (1.f - lr * wd) - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        master[i] = w;
        p[i] = f2bf(w);
    }
}

// sum of squares of a bf16 tensor into *out (fp32 atomic) — global grad-norm for clipping
__global__ __launch_bounds__(256) void sumsq_kernel(const unsigned short* __restrict__ g, float* __restrict__ out, long n) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = bf2f(g[i]);
        s += x * x;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// ---- deterministic global gradient norm: per-block partial sums (16 B per lane) into a caller buffer, then ONE block adds them in a fixed
//      tree order.  (sumsq_kernel's float atomics add in arrival order: data-parallel replicas would clip by slightly different factors and drift.)
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const unsigned short* __restrict__ g, float* __restrict__ partials, long n) {
    __shared__ float red[4];
    float s = 0.f;
    const long n8 = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float f[8];
        un8(*(const u32x4*)(g + i * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e] * f[e];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {   // ragged tail
        const float x = bf2f(g[n8 * 8 + threadIdx.x]);
        s += x * x;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partials, int np, float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + red[0];
}

// ---- AdamW, 8 elements per lane (16-byte bf16 / 2 x 16-byte f32 accesses); the clip factor min(1, max_norm / (sqrt(*sumsq) + 1e-6)) is derived
//      on the device from the norm sumsq_* left there (torch.nn.utils.clip_grad_norm_ / DeepSpeed gradient_clipping semantics, train_joint.py:324)
__global__ __launch_bounds__(256) void adamw8_kernel(unsigned short* __restrict__ p, float* __restrict__ master, const unsigned short* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2, const float* __restrict__ sumsq, float max_norm,
                                                     const unsigned char* __restrict__ row_active, long row_len8) {
    float gscale = 1.f;
    if (sumsq) gscale = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
    const long n8 = n / 8;
    const float decay = 1.f - lr * wd, ibc1 = 1.f / bc1, ibc2 = 1.f / bc2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        // rows whose gradient AND both moments are exactly zero (never touched) are left alone: with weight decay 0 their update is exactly zero
        if (row_active && !row_active[i / row_len8]) continue;
        float gr[8], w[8], mi[8], vi[8];
        un8(*(const u32x4*)(g + i * 8), gr);
        *(f32x4*)(w) = *(const f32x4*)(master + i * 8);     *(f32x4*)(w + 4) = *(const f32x4*)(master + i * 8 + 4);
        *(f32x4*)(mi) = *(const f32x4*)(m + i * 8);         *(f32x4*)(mi + 4) = *(const f32x4*)(m + i * 8 + 4);
        *(f32x4*)(vi) = *(const f32x4*)(v + i * 8);         *(f32x4*)(vi + 4) = *(const f32x4*)(v + i * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = gr[e] * gscale;
            mi[e] = b1 * mi[e] + (1.f - b1) * x;
            vi[e] = b2 * vi[e] + (1.f - b2) * x * x;
            w[e] = w[e] * decay - lr * (mi[e] * ibc1) / (sqrtf(vi[e] * ibc2) + eps);
        }
        *(f32x4*)(master + i * 8) = *(f32x4*)(w);           *(f32x4*)(master + i * 8 + 4) = *(f32x4*)(w + 4);
        *(f32x4*)(m + i * 8) = *(f32x4*)(mi);               *(f32x4*)(m + i * 8 + 4) = *(f32x4*)(mi + 4);
        *(f32x4*)(v + i * 8) = *(f32x4*)(vi);               *(f32x4*)(v + i * 8 + 4) = *(f32x4*)(vi + 4);
        *(u32x4*)(p + i * 8) = pk8_(w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {   // ragged tail, scalar
        const long i = n8 * 8 + threadIdx.x;
        const float x = bf2f(g[i]) * gscale;
        const float mm = b1 * m[i] + (1.f - b1) * x, vv = b2 * v[i] + (1.f - b2) * x * x;
        const float w = master[i] * decay - lr * (mm * ibc1) / (sqrtf(vv * ibc2) + eps);
        m[i] = mm; v[i] = vv; master[i] = w; p[i] = f2bf(w);
    }
}

// dst[idx[i]] += scale * src[i] on bf16 rows (one rounding per call); idx unique within the call (embedding-row gradient exchange)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(unsigned short* __restrict__ dst, const long* __restrict__ idx, const unsigned short* __restrict__ src,
                                                               long n, int dim, long ld_dst, long ld_src, float scale) {
    const int nch = dim / 8;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n * nch; t += (long)gridDim.x * 256) {
        const long r = t / nch;
        const int ch = (int)(t % nch);
        float a[8], b[8];
        unsigned short* d = dst + idx[r] * ld_dst + ch * 8;
        un8(*(const u32x4*)d, a);
        un8(*(const u32x4*)(src + r * ld_src + ch * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += scale * b[e];
        *(u32x4*)d = pk8_(a);
    }
}

static inline unsigned g1(long total, long cap = 256L * 32) {
    long b = cdiv(total, 256);
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

// ------------------------------------------------------------------------------------------------ dropout (LoRA branch)
// PEFT applies nn.Dropout(lora_dropout) to the INPUT of lora_A only (peft LoraLayer: result + lora_B(lora_A(dropout(x))) * scaling;
// reference train_joint.py:193-232 with lora_dropout = 0.05).  The mask is a pure function of (seed, element index) - a counter
// hash, not a stream - so the per-layer recompute in backward regenerates it bit for bit from the seed alone.
// y = (accumulate ? y : 0) + keep(x) / (1 - p); element e of 8-element group g keeps iff bits16(hash(seed, 4g + e/2), e & 1) >= p * 65536.
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned long long ctr) {
    unsigned x = (unsigned)ctr ^ (unsigned)seed, y = (unsigned)(ctr >> 32) ^ (unsigned)(seed >> 32);
    x *= 0xcc9e2d51u; x = (x << 15) | (x >> 17); x *= 0x1b873593u;
    y ^= x; y = (y << 13) | (y >> 19); y = y * 5u + 0xe6546b64u;
    y ^= y >> 16; y *= 0x85ebca6bu; y ^= y >> 13; y *= 0xc2b2ae35u; y ^= y >> 16;
    return y;
}

__global__ __launch_bounds__(256) void dropout_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, long n8, unsigned thr,
                                                      float inv_keep, unsigned long long seed, int accumulate) {
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long)gridDim.x * 256) {
        const u32x4 v = *(const u32x4*)(x + g * 8);
        u32x4 o = {0u, 0u, 0u, 0u};
        if (accumulate) o = *(const u32x4*)(y + g * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned h = dropout_hash(seed, (unsigned long long)g * 4 + j);
            const float lo = ((h & 0xffffu) >= thr) ? __uint_as_float(v[j] << 16) * inv_keep : 0.f;
            const float hi = ((h >> 16) >= thr) ? __uint_as_float(v[j] & 0xffff0000u) * inv_keep : 0.f;
            const float alo = accumulate ? __uint_as_float(o[j] << 16) : 0.f, ahi = accumulate ? __uint_as_float(o[j] & 0xffff0000u) : 0.f;
            o[j] = pack_bf2(alo + lo, ahi + hi);
        }
        *(u32x4*)(y + g * 8) = o;
    }
}

// The two LoRA branches of a decoder layer (q_proj, v_proj: reference train_joint.py:199-212) drop the SAME input with two masks, and their gradients meet in the
// same dh1: one launch for the pair.  sum == 0: ya = drop_a(xa), yb = drop_b(xb) (xa == xb in the forward: read once).  sum == 1: ya += drop_a(xa) + drop_b(xb)
// with ONE rounding (yb unused).  Masks exactly as dropout_kernel's.
__global__ __launch_bounds__(256) void dropout_pair_kernel(const unsigned short* __restrict__ xa, const unsigned short* __restrict__ xb, unsigned short* ya,
                                                           unsigned short* yb, long n8, unsigned thr_a, float ik_a, unsigned long long seed_a, unsigned thr_b,
                                                           float ik_b, unsigned long long seed_b, int sum) {
    const bool same = xa == xb;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long)gridDim.x * 256) {
        const u32x4 va = *(const u32x4*)(xa + g * 8);
        const u32x4 vb = same ? va : *(const u32x4*)(xb + g * 8);
        u32x4 o = {0u, 0u, 0u, 0u};
        if (sum) o = *(const u32x4*)(ya + g * 8);
        u32x4 oa, ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned ha = dropout_hash(seed_a, (unsigned long long)g * 4 + j), hb = dropout_hash(seed_b, (unsigned long long)g * 4 + j);
            const float alo = ((ha & 0xffffu) >= thr_a) ? __uint_as_float(va[j] << 16) * ik_a : 0.f;
            const float ahi = ((ha >> 16) >= thr_a) ? __uint_as_float(va[j] & 0xffff0000u) * ik_a : 0.f;
            const float blo = ((hb & 0xffffu) >= thr_b) ? __uint_as_float(vb[j] << 16) * ik_b : 0.f;
            const float bhi = ((hb >> 16) >= thr_b) ? __uint_as_float(vb[j] & 0xffff0000u) * ik_b : 0.f;
            if (sum) {
                oa[j] = pack_bf2(__uint_as_float(o[j] << 16) + alo + blo, __uint_as_float(o[j] & 0xffff0000u) + ahi + bhi);
            } else {
                oa[j] = pack_bf2(alo, ahi);
                ob[j] = pack_bf2(blo, bhi);
            }
        }
        *(u32x4*)(ya + g * 8) = oa;
        if (!sum) *(u32x4*)(yb + g * 8) = ob;
    }
}

}  // namespace rga3

using namespace rga3;
typedef const unsigned short* cus;
typedef unsigned short* us;

extern "C" int rga3_rmsnorm_bwd(const void* x, const void* weight, const void* dy, const void* add, void* dx, int64_t rows, int64_t dim,
                                float eps, void* stream) {
    RGA3_CHECK_ARG(x && weight && dy && dx && rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, "rmsnorm_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dim > 1024) {   // one workgroup per row
        dim3 gb((unsigned)rows);
        if (dim <= 2048) hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<1>, gb, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (cus)add, (us)dx, (int)dim, eps);
        else if (dim <= 4096) hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<2>, gb, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (cus)add, (us)dx, (int)dim, eps);
        else hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<4>, gb, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (cus)add, (us)dx, (int)dim, eps);
        RGA3_CHECK_LAUNCH("rmsnorm_bwd_block");
        return 0;
    }
    dim3 grid((unsigned)cdiv(rows, 4));
    if (dim <= 2048) hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, grid, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (cus)add, (us)dx, (long)rows, (int)dim, eps);
    else if (dim <= 4096) hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, grid, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (cus)add, (us)dx, (long)rows, (int)dim, eps);
    else hipLaunchKernelGGL(rmsnorm_bwd_kernel<16>, grid, dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (cus)add, (us)dx, (long)rows, (int)dim, eps);
    RGA3_CHECK_LAUNCH("rmsnorm_bwd");
    return 0;
}

extern "C" int rga3_dropout_bf16(const void* x, void* y, int64_t n, float p, int64_t seed, int accumulate, void* stream) {
    RGA3_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "dropout: n=%ld must be a positive multiple of 8", (long)n);
    RGA3_CHECK_ARG(p >= 0.f && p < 1.f, "dropout: p=%f", p);
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    const float inv_keep = 1.0f / (1.0f - (float)thr / 65536.0f);   // exactly the keep probability the 16-bit threshold realises
    long blocks = cdiv(n / 8, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, (long)(n / 8),
                       thr, inv_keep, (unsigned long long)seed, accumulate);
    RGA3_CHECK_LAUNCH("dropout_kernel");
    return 0;
}

extern "C" int rga3_dropout_pair_bf16(const void* xa, const void* xb, void* ya, void* yb, int64_t n, float pa, int64_t seed_a, float pb, int64_t seed_b, int sum,
                                      void* stream) {
    RGA3_CHECK_ARG(xa && xb && ya && (sum || yb) && n > 0 && n % 8 == 0, "dropout_pair: null pointer / n=%ld must be a positive multiple of 8", (long)n);
    RGA3_CHECK_ARG(pa >= 0.f && pa < 1.f && pb >= 0.f && pb < 1.f, "dropout_pair: p=%f / %f", pa, pb);
    RGA3_CHECK_ARG((((uintptr_t)xa | (uintptr_t)xb | (uintptr_t)ya | (uintptr_t)yb) & 15) == 0, "dropout_pair: pointers must be 16-byte aligned");
    const unsigned ta = (unsigned)(pa * 65536.0f + 0.5f), tb = (unsigned)(pb * 65536.0f + 0.5f);
    long blocks = cdiv(n / 8, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(dropout_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (cus)xa, (cus)xb, (us)ya, (us)yb, (long)(n / 8), ta,
                       1.0f / (1.0f - (float)ta / 65536.0f), (unsigned long long)seed_a, tb, 1.0f / (1.0f - (float)tb / 65536.0f), (unsigned long long)seed_b, sum);
    RGA3_CHECK_LAUNCH("dropout_pair_kernel");
    return 0;
}

extern "C" int rga3_swiglu_bwd(const void* gu, const void* da, void* dgu, int64_t T, int64_t I, void* stream) {
    RGA3_CHECK_ARG(gu && da && dgu && T > 0 && I > 0 && I % 16 == 0, "swiglu_bwd: bad args");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(g1(T * (I / 8))), dim3(256), 0, (hipStream_t)stream, (cus)gu, (cus)da, (us)dgu, (long)T, (long)I);
    RGA3_CHECK_LAUNCH("swiglu_bwd");
    return 0;
}

extern "C" int rga3_swiglu_fwd(const void* gu, void* a, int64_t T, int64_t I, void* stream) {
    RGA3_CHECK_ARG(gu && a && T > 0 && I > 0 && I % 16 == 0, "swiglu_fwd: bad args");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(g1(T * (I / 8))), dim3(256), 0, (hipStream_t)stream, (cus)gu, (us)a, (long)T, (long)I);
    RGA3_CHECK_LAUNCH("swiglu_fwd");
    return 0;
}

extern "C" int rga3_transpose16(const void* in, void* out, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, void* stream) {
    RGA3_CHECK_ARG(in && out && R > 0 && C > 0 && ld_in >= C && ld_out >= R, "transpose16: bad args");
    RGA3_CHECK_ARG(cdiv(R, 64) <= 65535, "transpose16: too many rows");
    hipLaunchKernelGGL(transpose16_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, (cus)in, (us)out, (long)R,
                       (long)C, (long)ld_in, (long)ld_out);
    RGA3_CHECK_LAUNCH("transpose16");
    return 0;
}

// n (<= 64) contiguous 16-bit matrices transposed in one launch: in[i] [R_i, C_i] -> out[i] [C_i, R_i].  HOST arrays: ptrs = n x {in, out}, dims = n x {R, C}.
extern "C" int rga3_transpose16_many(const void* const* ptrs, const int64_t* dims, int n, void* stream) {
    RGA3_CHECK_ARG(ptrs && dims && n >= 1 && n <= TR_MANY, "transpose16_many: n %d (1..64)", n);
    TransposeMany p;
    long mr = 0, mc = 0;
    for (int i = 0; i < n; ++i) {
        p.in[i] = (const unsigned short*)ptrs[2 * i];
        p.out[i] = (unsigned short*)ptrs[2 * i + 1];
        RGA3_CHECK_ARG(p.in[i] && p.out[i] && dims[2 * i] > 0 && dims[2 * i + 1] > 0 && dims[2 * i] < (1 << 30) && dims[2 * i + 1] < (1 << 30), "transpose16_many: matrix %d", i);
        p.R[i] = (int)dims[2 * i];
        p.C[i] = (int)dims[2 * i + 1];
        if (p.R[i] > mr) mr = p.R[i];
        if (p.C[i] > mc) mc = p.C[i];
    }
    RGA3_CHECK_ARG(cdiv(mr, 64) <= 65535, "transpose16_many: too many rows");
    hipLaunchKernelGGL(transpose16_many_kernel, dim3((unsigned)cdiv(mc, 64), (unsigned)cdiv(mr, 64), (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    RGA3_CHECK_LAUNCH("transpose16_many");
    return 0;
}

extern "C" int rga3_segment_sum_rows(const void* x, const int64_t* rows, const int64_t* offsets, void* out, int64_t n_out, int64_t dim,
                                     int64_t ldx, void* stream) {
    RGA3_CHECK_ARG(x && rows && offsets && out && n_out > 0 && dim % 8 == 0 && ldx % 8 == 0, "segment_sum_rows: bad args");
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3((unsigned)n_out), dim3(256), 0, (hipStream_t)stream, (cus)x, (const long*)rows, (const long*)offsets,
                       (us)out, (int)dim, (long)ldx);
    RGA3_CHECK_LAUNCH("segment_sum_rows");
    return 0;
}

extern "C" int rga3_adamw_step(void* param, float* master, const void* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, void* stream) {
    RGA3_CHECK_ARG(param && master && grad && m && v && n > 0 && step >= 1, "adamw_step: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, (us)param, master, (cus)grad, m, v, (long)n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, grad_scale);
    RGA3_CHECK_LAUNCH("adamw_step");
    return 0;
}

extern "C" int rga3_sumsq_accum(const void* g, float* out, int64_t n, void* stream) {
    RGA3_CHECK_ARG(g && out && n > 0, "sumsq_accum: bad args");
    hipLaunchKernelGGL(sumsq_kernel, dim3(g1(n, 2048)), dim3(256), 0, (hipStream_t)stream, (cus)g, out, (long)n);
    RGA3_CHECK_LAUNCH("sumsq_accum");
    return 0;
}

extern "C" int rga3_sumsq_det(const void* g, int64_t n, float* partials, int64_t partials_cap, float* out, int accumulate, void* stream) {
    RGA3_CHECK_ARG(g && partials && out && n > 0 && partials_cap >= 1, "sumsq_det: bad args");
    RGA3_CHECK_ARG(((uintptr_t)g & 15) == 0, "sumsq_det: gradient pointer must be 16-byte aligned");
    long nb = cdiv(n, 8 * 256 * 4);
    if (nb < 1) nb = 1;
    if (nb > partials_cap) nb = partials_cap;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (cus)g, partials, (long)n);
    RGA3_CHECK_LAUNCH("sumsq_partials");
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, (int)nb, out, accumulate);
    RGA3_CHECK_LAUNCH("sumsq_finish");
    return 0;
}

extern "C" int rga3_adamw_step_clip(void* param, float* master, const void* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, int step, const float* sumsq, float max_norm, void* stream) {
    RGA3_CHECK_ARG(param && master && grad && m && v && n > 0 && step >= 1, "adamw_step_clip: bad args");
    RGA3_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)master | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw_step_clip: pointers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw8_kernel, dim3(g1(cdiv(n, 8), 256L * 16)), dim3(256), 0, (hipStream_t)stream, (us)param, master, (cus)grad, m, v, (long)n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2, sumsq, max_norm, (const unsigned char*)nullptr, 1L);
    RGA3_CHECK_LAUNCH("adamw_step_clip");
    return 0;
}

// The same update for a [rows, row_len] table (embed_tokens) of which only the rows with row_active[r] != 0 are touched.  A row that never received a
// gradient has g = m = v = 0, and with weight_decay == 0 AdamW leaves it bit-for-bit unchanged (m, v stay 0, the step is lr * 0 / (0 + eps)) -- skipping it
// is exact, and one training sample touches <= S of the 152 064 rows (the dense update streams 30 B per element: 16 GB for the table).
extern "C" int rga3_adamw_step_clip_rows(void* param, float* master, const void* grad, float* m, float* v, int64_t rows, int64_t row_len, const uint8_t* row_active,
                                         float lr, float beta1, float beta2, float eps, int step, const float* sumsq, float max_norm, void* stream) {
    RGA3_CHECK_ARG(param && master && grad && m && v && row_active && rows > 0 && row_len > 0 && row_len % 8 == 0 && step >= 1, "adamw_step_clip_rows: bad args (row_len % 8)");
    RGA3_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)master | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw_step_clip_rows: pointers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    const int64_t n = rows * row_len;
    hipLaunchKernelGGL(adamw8_kernel, dim3(g1(cdiv(n, 8), 256L * 16)), dim3(256), 0, (hipStream_t)stream, (us)param, master, (cus)grad, m, v, (long)n, lr, beta1, beta2,
                       eps, 0.f, bc1, bc2, sumsq, max_norm, row_active, (long)(row_len / 8));
    RGA3_CHECK_LAUNCH("adamw_step_clip_rows");
    return 0;
}

extern "C" int rga3_scatter_add_rows(void* dst, const int64_t* idx, const void* src, int64_t n, int64_t dim, int64_t ld_dst, int64_t ld_src, float scale,
                                     void* stream) {
    RGA3_CHECK_ARG(dst && idx && src && n > 0 && dim > 0 && dim % 8 == 0 && ld_dst % 8 == 0 && ld_src % 8 == 0, "scatter_add_rows: bad args");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(g1(n * (dim / 8))), dim3(256), 0, (hipStream_t)stream, (us)dst, (const long*)idx, (cus)src, (long)n, (int)dim,
                       (long)ld_dst, (long)ld_src, scale);
    RGA3_CHECK_LAUNCH("scatter_add_rows");
    return 0;
}
