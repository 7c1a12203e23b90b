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

// Rows of 16..512 channels (LayerNorm2d of the upscaling path over 10^5..10^6 pixels, the two-way transformer's LayerNorms over 65 536 tokens x 256;
// reference model/sam2.py:1976-1986, 1364-1376): G = dim / 8 lanes per row, 64 / G rows per wave pass, row statistics by xor-shuffles inside the lane
// group.  dw / db accumulate per lane over the rows it sees, are folded across the lane groups of the wave by shuffles, across the four waves through
// LDS, and leave as ONE partial row per workgroup: part[wg][0..DIM) = dw, [DIM..2 DIM) = db.  colsum_finish_kernel adds the partial rows in index order,
// so the parameter gradients are bit-reproducible.  (Round 1 sent them out as f32 atomics: 2 M atomics on 512 addresses took 180 of the 206 us of a
// 65 536 x 256 call -- same-address atomics serialise at ~50 ns each.)
template <int G>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                                 const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                                                                 float* __restrict__ part, long rows, float eps) {
    constexpr int DIM = G * 8, RPW = 64 / G;   // rows per wave pass
    __shared__ float red[4][2 * DIM];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = lane % G, sub = lane / G;
    float fw[8], aw[8], ab[8];
    u8_(*(const u32x4*)(w + ch * 8), fw);
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[e] = 0.f; ab[e] = 0.f; }
    for (long base = ((long)blockIdx.x * 4 + wv) * RPW; base < rows; base += (long)gridDim.x * 4 * RPW) {
        const long row = base + sub;
        const bool ok = row < rows;
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
    if (part) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) { aw[e] += __shfl_xor(aw[e], o, 64); ab[e] += __shfl_xor(ab[e], o, 64); }
        }
        if (sub == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[wv][ch * 8 + e] = aw[e];
                red[wv][DIM + ch * 8 + e] = ab[e];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * DIM; i += 256) part[(long)blockIdx.x * (2 * DIM) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

// Column sums in two deterministic stages.  Stage 1: a workgroup = CG 16-byte column chunks x RS row lanes walks rows_per_wg rows (four independent
// loads in flight per thread), folds its row lanes through LDS and writes one partial row part[blockIdx.y][cols].  Stage 2 (colsum_finish_kernel) adds
// the partial rows in index order.  x [rows, cols] bf16, cols % 8 == 0, 16-byte aligned rows.
// With `counters` (one word per blockIdx.x, zeroed once by the caller, left zero) the workgroup that finishes LAST for a column block also runs stage 2 for those
// columns -- same order of additions as colsum_finish_kernel, so the result is bit-identical to the two-launch form -- and no second launch is needed.
__global__ __launch_bounds__(256) void colsum_partials_kernel(const unsigned short* __restrict__ x, float* part, long rows, int cols, long ld,
                                                              int CG, int RS, long rows_per_wg, unsigned* counters, float* out) {
    __shared__ float red[256 * 8];
    const int t = threadIdx.x;
    const int cg = t % CG, rs = t / CG;
    const int ch = (int)blockIdx.x * CG + cg;
    const bool act = rs < RS && ch * 8 < cols;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const long r0 = (long)blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    if (act) {
        const unsigned short* px = x + ch * 8;
        long r = r0 + rs;
        for (; r + 3L * RS < r1; r += 4L * RS) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *(const u32x4*)(px + (r + (long)j * RS) * ld);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f[8];
                u8_(v[j], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
        }
        for (; r < r1; r += RS) {
            float f[8];
            u8_(*(const u32x4*)(px + r * ld), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
    }
    if (rs < RS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(rs * CG + cg) * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int i = t; i < CG * 8; i += 256) {
        float s = 0.f;
        for (int q = 0; q < RS; ++q) s += red[q * CG * 8 + i];
        const long col = (long)blockIdx.x * CG * 8 + i;
        if (col < cols) part[(long)blockIdx.y * cols + col] = s;
    }
    if (!counters) return;
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {      // cdna_hip_programming.md Guideline 16: drain, agent-scope release, ticket; the last arrival acquires for the workgroup
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (prev == gridDim.y - 1) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    const int n = (int)gridDim.y;
    const int c = t & 31, cl = t >> 5;
    float* fin = red;                       // [8][32]
    for (int cb = 0; cb < CG * 8; cb += 32) {
        const long col = (long)blockIdx.x * CG * 8 + cb + c;
        float s = 0.f;
        if (col < cols)
            for (int i = cl; i < n; i += 8) s += part[(long)i * cols + col];
        __syncthreads();
        fin[cl * 32 + c] = s;
        __syncthreads();
        if (cl == 0 && col < cols)
            out[col] = ((fin[c] + fin[32 + c]) + (fin[64 + c] + fin[96 + c])) + ((fin[128 + c] + fin[160 + c]) + (fin[192 + c] + fin[224 + c]));
    }
    if (t == 0) __hip_atomic_store(counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out[c] = sum_i part[i][c], i ascending inside each of 8 interleaved lanes, lanes folded in a fixed order.  Columns < split go to out0, the rest to out1
// (LayerNorm: dw | db in one partial row).
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int n, int cols, float* __restrict__ out0, float* __restrict__ out1,
                                                            int split) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int col = (int)blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < cols)
        for (int i = cl; i < n; i += 8) s += part[(long)i * cols + col];
    red[cl][c] = s;
    __syncthreads();
    if (cl == 0 && col < cols) {
        float v = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
        if (col < split) out0[col] = v;
        else out1[col - split] = v;
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

// Backward of bilinear_kernel as a GATHER: thread = one input pixel of output plane n's source plane, which collects the taps of every output pixel
// whose bilinear footprint contains it.  The candidate output rows / columns come from inverting the source-index formula with a margin of one; each
// candidate re-evaluates the forward's float expressions, so the weights are exactly the forward's.  Without plane_idx every element of din is written
// (no pre-zeroing, bit-reproducible); with plane_idx the sum is added to the pre-zeroed plane plane_idx[n] with ONE atomic per pixel (distinct indices --
// the selected-candidate upsampling -- stay deterministic; repeated indices are still summed correctly).  (Round 1 scattered four atomics per OUTPUT
// pixel: 67 M same-plane atomics, 711 us - 1.3 ms per call on 16 x 256^2 <-> 1024^2.)
__global__ __launch_bounds__(256) void bilinear_bwd_gather_kernel(const float* __restrict__ dout, float* __restrict__ din, const int* __restrict__ plane_idx, long N,
                                                                  int Hi, int Wi, int Ho, int Wo) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const long total = N * Hi * Wi;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ix = (int)(i % Wi), iy = (int)((i / Wi) % Hi);
        const long n = i / ((long)Wi * Hi);
        // output rows y with y0(y) == iy or y1(y) == iy satisfy  iy - 1 <= sy (y + .5) - .5 < iy + 1  (clamped ends widen it: handled by the margin + exact test)
        const int ylo = max(0, (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1), yhi = min(Ho - 1, (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1);
        const int xlo = max(0, (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1), xhi = min(Wo - 1, (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1);
        const float* plane = dout + n * Ho * (long)Wo;
        float acc = 0.f;
        for (int y = ylo; y <= yhi; ++y) {
            const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)fy, y1 = min(y0 + 1, Hi - 1);
            const float ly = fy - (float)y0;
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            float rowacc = 0.f;
            for (int x = xlo; x <= xhi; ++x) {
                const float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
                const int x0 = (int)fx, x1 = min(x0 + 1, Wi - 1);
                const float lx = fx - (float)x0;
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                rowacc += wx * plane[(long)y * Wo + x];
            }
            acc += wy * rowacc;
        }
        if (plane_idx) atomicAdd(din + ((long)plane_idx[n] * Hi + iy) * Wi + ix, acc);
        else din[i] = acc;
    }
}

// ---- mask logits of all frames in one launch (reference model/sam2.py:2142-2149: masks = hyper_in[B,4,32] @ upscaled[B,32,HW]) and their backward.
//      hyper [B, NM, C] bf16, up [B * P, C] bf16 (token-major upscaled features), masks [B, NM, P] f32.  C = 8 CH <= 32, NM = 4.  CH lanes share a pixel,
//      each owns one 16-byte channel chunk (a wave reads 1 KiB of up contiguously), NM dot products in fp32 folded by xor-shuffles inside the lane group.
//      HBM-bound: 67 MB in + 17 MB out for 16 frames of 256^2.  (Per frame this was a 4-row skinny product with K = 32 -- 4 of 64 lanes busy, 79 us x 16.)
template <int CH, int NM>
__global__ __launch_bounds__(256) void mask_product_kernel(const unsigned short* __restrict__ hyper, const unsigned short* __restrict__ up, float* __restrict__ masks,
                                                           long P) {
    constexpr int C = CH * 8, PPB = 256 / CH;   // pixels per workgroup pass
    const long b = blockIdx.y;
    const int ch = threadIdx.x % CH, sub = threadIdx.x / CH;
    float hy[NM][8];
#pragma unroll
    for (int m = 0; m < NM; ++m) u8_(*(const u32x4*)(hyper + (b * NM + m) * C + ch * 8), hy[m]);
    for (long p = (long)blockIdx.x * PPB + sub; p < P; p += (long)gridDim.x * PPB) {
        float f[8], acc[NM];
        u8_(*(const u32x4*)(up + (b * P + p) * C + ch * 8), f);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(f[e], hy[m][e], a);
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) a += __shfl_xor(a, o, 64);
            acc[m] = a;
        }
        // lane j of the group writes mask j (CH < NM: lane 0 writes the rest)
#pragma unroll
        for (int m = 0; m < NM; ++m)
            if (ch == (m < CH ? m : 0)) masks[(b * NM + m) * P + p] = acc[m];
    }
}

// backward: dup[b,p,c] = sum_m dmasks[b,m,p] hyper[b,m,c] (bf16 out) and, in the same pass over up, the per-workgroup partial sums of
// dhyper[b,m,c] = sum_p dmasks[b,m,p] up[b,p,c]: part[b][blockIdx.x][NM * C] f32, added in index order by mask_product_finish_kernel.
template <int CH, int NM>
__global__ __launch_bounds__(256) void mask_product_bwd_kernel(const float* __restrict__ dmasks, const unsigned short* __restrict__ hyper,
                                                               const unsigned short* __restrict__ up, unsigned short* __restrict__ dup, float* __restrict__ part,
                                                               long P) {
    constexpr int C = CH * 8, PPB = 256 / CH;
    __shared__ float red[4][NM * C];
    const long b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = threadIdx.x % CH, sub = threadIdx.x / CH;
    float hy[NM][8], dh[NM][8];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        u8_(*(const u32x4*)(hyper + (b * NM + m) * C + ch * 8), hy[m]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dh[m][e] = 0.f;
    }
    for (long p = (long)blockIdx.x * PPB + sub; p < P; p += (long)gridDim.x * PPB) {
        float g[NM], f[8], o[8];
#pragma unroll
        for (int m = 0; m < NM; ++m) g[m] = dmasks[(b * NM + m) * P + p];
        const long off = (b * P + p) * C + ch * 8;
        u8_(*(const u32x4*)(up + off), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                a = fmaf(g[m], hy[m][e], a);
                dh[m][e] = fmaf(g[m], f[e], dh[m][e]);
            }
            o[e] = a;
        }
        *(u32x4*)(dup + off) = p8_(o);
    }
    // fold the pixel groups of the wave (lanes with equal ch), then the four waves
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = dh[m][e];
#pragma unroll
            for (int o = CH; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < CH) red[wv][m * C + ch * 8 + e] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < NM * C; i += 256)
        part[(b * gridDim.x + blockIdx.x) * (NM * C) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

__global__ __launch_bounds__(256) void mask_product_finish_kernel(const float* __restrict__ part, unsigned short* __restrict__ dhyper, int n, int len) {
    const long b = blockIdx.x;
    for (int i = threadIdx.x; i < len; i += 256) {
        float s = 0.f;
        for (int q = 0; q < n; ++q) s += part[(b * n + q) * len + i];
        dhyper[b * len + i] = f2bf(s);
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

static int ln_bwd_groups(int64_t dim) {
    return (dim == 16 || dim == 32 || dim == 64 || dim == 128 || dim == 256 || dim == 512) ? (int)(dim / 8) : 0;
}
static int ln_bwd_wgs(int64_t rows, int G) {
    const long rpw = 4L * (64 / G);                 // rows per workgroup pass
    long n = cdiv(rows, rpw * 2);                   // at least two passes per workgroup
    if (n > 512) n = 512;
    if (n < 1) n = 1;
    return (int)n;
}

// f32 scratch elements rga3_layernorm_bwd wants for this shape (0: none, the generic-width kernel accumulates with atomics)
extern "C" int64_t rga3_layernorm_bwd_ws_floats(int64_t rows, int64_t dim) {
    const int G = ln_bwd_groups(dim);
    return G ? (int64_t)ln_bwd_wgs(rows, G) * 2 * dim : 0;
}

extern "C" int rga3_layernorm_bwd(const void* x, const void* weight, const void* dy, void* dx, float* dweight, float* dbias, int64_t rows, int64_t dim,
                                  float eps, float* ws, int64_t ws_floats, void* stream) {
    RGA3_CHECK_ARG(x && weight && dy && dx && rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 2048, "layernorm_bwd: bad args (dim <= 2048)");
    RGA3_CHECK_ARG((dweight == nullptr) == (dbias == nullptr), "layernorm_bwd: dweight and dbias go together");
    hipStream_t st = (hipStream_t)stream;
    const int G = ln_bwd_groups(dim);
    if (G) {   // lane-group rows, parameter gradients through per-workgroup partial rows (deterministic; dweight / dbias are WRITTEN)
        const int nwg = ln_bwd_wgs(rows, G);
        RGA3_CHECK_ARG(!dweight || (ws && ws_floats >= (int64_t)nwg * 2 * dim), "layernorm_bwd: workspace of rga3_layernorm_bwd_ws_floats() f32 elements needed");
        float* part = dweight ? ws : nullptr;
#define RGA3_LNB(GG) hipLaunchKernelGGL(layernorm_bwd_rows_kernel<GG>, dim3(nwg), dim3(256), 0, st, (cus)x, (cus)weight, (cus)dy, (us)dx, part, (long)rows, eps)
        switch (G) {
            case 2: RGA3_LNB(2); break;
            case 4: RGA3_LNB(4); break;
            case 8: RGA3_LNB(8); break;
            case 16: RGA3_LNB(16); break;
            case 32: RGA3_LNB(32); break;
            default: RGA3_LNB(64); break;
        }
#undef RGA3_LNB
        RGA3_CHECK_LAUNCH("layernorm_bwd_rows");
        if (dweight) {
            hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)cdiv(2 * dim, 32)), dim3(256), 0, st, (const float*)part, nwg, (int)(2 * dim), dweight, dbias, (int)dim);
            RGA3_CHECK_LAUNCH("layernorm_bwd_finish");
        }
        return 0;
    }
    // other widths: wave per row, f32 atomics into dweight / dbias (the caller zeroes them)
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

static void colsum_plan(int64_t rows, int64_t cols, int& CG, int& RS, int& nchunk, long& rows_per_wg) {
    const int G = (int)(cols / 8);
    CG = G < 256 ? G : 256;
    RS = 256 / CG;
    long n = cdiv(rows, (long)RS * 8);
    if (n > 256) n = 256;
    if (n < 1) n = 1;
    rows_per_wg = cdiv(rows, n);
    nchunk = (int)cdiv(rows, rows_per_wg);
}

extern "C" int64_t rga3_colsum_ws_floats(int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0 || cols % 8) return 0;
    int CG, RS, nchunk;
    long rpw;
    colsum_plan(rows, cols, CG, RS, nchunk, rpw);
    return (int64_t)nchunk * cols;
}

// out[c] = sum_r x[r, c], written (not accumulated), deterministic; cols % 8 == 0, ld % 8 == 0, x 16-byte aligned; ws: rga3_colsum_ws_floats() f32 elements
extern "C" int rga3_colsum(const void* x, float* out, int64_t rows, int64_t cols, int64_t ld, float* ws, int64_t ws_floats, void* counters, void* stream) {
    RGA3_CHECK_ARG(x && out && ws && rows > 0 && cols > 0 && ld >= cols, "colsum: bad args");
    RGA3_CHECK_ARG(cols % 8 == 0 && ld % 8 == 0 && (((uintptr_t)x) & 15) == 0, "colsum: cols / ld multiples of 8 and 16-byte aligned rows (use rga3_colsum_accum otherwise)");
    int CG, RS, nchunk;
    long rpw;
    colsum_plan(rows, cols, CG, RS, nchunk, rpw);
    RGA3_CHECK_ARG(ws_floats >= (int64_t)nchunk * cols, "colsum: workspace of rga3_colsum_ws_floats() f32 elements needed");
    hipStream_t st = (hipStream_t)stream;
    if (counters && cdiv(cols / 8, CG) <= 128) {     // both stages in one launch (the last workgroup of a column block finishes it; same order of additions)
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)cdiv(cols / 8, CG), (unsigned)nchunk), dim3(256), 0, st, (cus)x, ws, (long)rows, (int)cols, (long)ld, CG,
                           RS, rpw, (unsigned*)counters, out);
        RGA3_CHECK_LAUNCH("colsum_partials<fused finish>");
        return 0;
    }
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)cdiv(cols / 8, CG), (unsigned)nchunk), dim3(256), 0, st, (cus)x, ws, (long)rows, (int)cols, (long)ld, CG, RS,
                       rpw, (unsigned*)nullptr, (float*)nullptr);
    RGA3_CHECK_LAUNCH("colsum_partials");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)cdiv(cols, 32)), dim3(256), 0, st, (const float*)ws, nchunk, (int)cols, out, out, (int)cols);
    RGA3_CHECK_LAUNCH("colsum_finish");
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
    hipLaunchKernelGGL(bilinear_bwd_gather_kernel, dim3(gd(N * (long)Hi * Wi)), dim3(256), 0, (hipStream_t)stream, dout, din, plane_idx, (long)N, Hi, Wi, Ho, Wo);
    RGA3_CHECK_LAUNCH("bilinear_bwd");
    return 0;
}

static int mask_product_blocks(int64_t P) {
    long n = cdiv(P, 1024);   // >= 16 passes per workgroup at C = 32: the per-lane partial sums of the backward are folded once per workgroup
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return (int)n;
}

extern "C" int rga3_mask_product(const void* hyper, const void* up, float* masks, int64_t B, int64_t NM, int64_t P, int64_t C, void* stream) {
    RGA3_CHECK_ARG(hyper && up && masks && B > 0 && B <= 65535 && P > 0, "mask_product: bad args");
    RGA3_CHECK_ARG(NM == 4 && (C == 32 || C == 16 || C == 8), "mask_product: 4 mask tokens x 8 / 16 / 32 channels (got %ld x %ld)", (long)NM, (long)C);
    RGA3_CHECK_ARG((((uintptr_t)up) & 15) == 0, "mask_product: up must be 16-byte aligned");
    dim3 grid((unsigned)cdiv(P, 256), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (C == 32) hipLaunchKernelGGL((mask_product_kernel<4, 4>), grid, dim3(256), 0, st, (cus)hyper, (cus)up, masks, (long)P);
    else if (C == 16) hipLaunchKernelGGL((mask_product_kernel<2, 4>), grid, dim3(256), 0, st, (cus)hyper, (cus)up, masks, (long)P);
    else hipLaunchKernelGGL((mask_product_kernel<1, 4>), grid, dim3(256), 0, st, (cus)hyper, (cus)up, masks, (long)P);
    RGA3_CHECK_LAUNCH("mask_product");
    return 0;
}

extern "C" int64_t rga3_mask_product_bwd_ws_floats(int64_t B, int64_t NM, int64_t P, int64_t C) {
    return B * mask_product_blocks(P) * NM * C;
}

extern "C" int rga3_mask_product_bwd(const float* dmasks, const void* hyper, const void* up, void* dup, void* dhyper, int64_t B, int64_t NM, int64_t P, int64_t C,
                                     float* ws, int64_t ws_floats, void* stream) {
    RGA3_CHECK_ARG(dmasks && hyper && up && dup && dhyper && ws && B > 0 && B <= 65535 && P > 0, "mask_product_bwd: bad args");
    RGA3_CHECK_ARG(NM == 4 && (C == 32 || C == 16 || C == 8), "mask_product_bwd: 4 mask tokens x 8 / 16 / 32 channels (got %ld x %ld)", (long)NM, (long)C);
    RGA3_CHECK_ARG(((((uintptr_t)up) | ((uintptr_t)dup)) & 15) == 0, "mask_product_bwd: up / dup must be 16-byte aligned");
    const int nb = mask_product_blocks(P);
    RGA3_CHECK_ARG(ws_floats >= B * nb * NM * C, "mask_product_bwd: workspace of rga3_mask_product_bwd_ws_floats() f32 elements needed");
    dim3 grid((unsigned)nb, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (C == 32) hipLaunchKernelGGL((mask_product_bwd_kernel<4, 4>), grid, dim3(256), 0, st, dmasks, (cus)hyper, (cus)up, (us)dup, ws, (long)P);
    else if (C == 16) hipLaunchKernelGGL((mask_product_bwd_kernel<2, 4>), grid, dim3(256), 0, st, dmasks, (cus)hyper, (cus)up, (us)dup, ws, (long)P);
    else hipLaunchKernelGGL((mask_product_bwd_kernel<1, 4>), grid, dim3(256), 0, st, dmasks, (cus)hyper, (cus)up, (us)dup, ws, (long)P);
    RGA3_CHECK_LAUNCH("mask_product_bwd");
    hipLaunchKernelGGL(mask_product_finish_kernel, dim3((unsigned)B), dim3(256), 0, st, (const float*)ws, (us)dhyper, nb, (int)(NM * C));
    RGA3_CHECK_LAUNCH("mask_product_finish");
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
