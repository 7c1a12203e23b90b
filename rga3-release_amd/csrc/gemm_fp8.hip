// FP8 (OCP e4m3) "NT" GEMM for the frozen-weight contractions of the LoRA fine-tune step (BASELINE.json configs[4]: "fp8 MFMA
// GEMMs"):  C[M,N] = (Aq[M,K] . Wq[N,K]^T) * sa[m] * sw[n] (+ bias) (+ residual),  bf16 out.
// Aq / Wq are row-quantised by rga3_quant_fp8_rows (scale = row amax / 448, RNE): activations per token, weights per output feature
// (and a second copy of W^T per input feature for the backward dX = dY . W).
//
// Same ping-pong main loop as gemm_nt_pp_kernel (gemm_bf16.hip) with bytes in place of bf16 pairs: a K-tile is 128 fp8 per row =
// the same 128-byte LDS rows, half-tiles, source-side XOR swizzle and counted-vmcnt schedule; a phase issues 8
// v_mfma_scale_f32_16x16x128_f8f6f4 (32 cycles each, 2x the bf16 MFMA rate; block scales fixed at 2^0) instead of 16 bf16 MFMAs, so
// every byte that moves through LDS carries twice the FLOPs.  Both operands read their 32 K-bytes per lane the same way
// (bytes [32 (lane>>4), +32) of row lane&15), so whatever order the instruction assigns to those bytes inside K is the same for A and B.
#include "common.h"

#include <type_traits>

namespace rga3 {

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Fp8GemmArgs {
    const unsigned char* A;
    const unsigned char* W;
    const float* sa;  // [M] row scales of A
    const float* sw;  // [N] row scales of W
    const unsigned short* bias;  // [N] bf16 or null
    const unsigned short* res;   // [M, N] bf16 or null
    unsigned short* C;
    int M, N, K;
    long lda, ldw, ldc, ldr;
    int ntm, ntn, group_m;
};

// ---------------------------------------------------------------------------------------------- row quantiser
// x / scale, correctly rounded, from r = RN(1 / scale) computed once per row: q0 = x r is within an ulp, e = x - q0 scale is exact in an FMA, q0 + e r rounds to
// RN(x / scale) (Markstein's division step; |x / scale| <= 448 here, nothing over- or underflows) -- 3 operations per element instead of the ~10 of the IEEE
// expansion, and the same e4m3 codes (tests/test_kernels_gpu.py compares every code with torch.float8_e4m3fn of the true quotient).
__device__ __forceinline__ float div_by_scale(float x, float scale, float r) {
    const float q0 = x * r;
    const float e = __builtin_fmaf(-q0, scale, x);
    return __builtin_copysignf(__builtin_fmaf(e, r, q0), x);   // (x = -0 would come back as +0 from the sum: the e4m3 code keeps the sign of zero)
}
// one 256-thread block per row: amax, scale = amax / 448 (1 for an all-zero row), q = e4m3(x / scale)
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const unsigned short* __restrict__ x, unsigned char* __restrict__ q, float* __restrict__ scales,
                                                             long rows, int K, long ldx, long ldq) {
    __shared__ float part[4];
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const unsigned short* xr = x + row * ldx;
    const int nch = K / 8;
    constexpr int KEEP = 4;   // a thread's first chunks stay in registers between the amax pass and the quantising pass (K <= 8192: the row is read once)
    u32x4 kept[KEEP];
    float amax = 0.f;
    auto amax8 = [&](const u32x4& v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = fmaxf(amax, fabsf(__uint_as_float(v[e] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(v[e] & 0xffff0000u)));
        }
    };
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ch = tid + i * 256;
        kept[i] = u32x4{0u, 0u, 0u, 0u};
        if (ch < nch) {
            kept[i] = *(const u32x4*)(xr + ch * 8);
            amax8(kept[i]);
        }
    }
    for (int ch = tid + KEEP * 256; ch < nch; ch += 256) amax8(*(const u32x4*)(xr + ch * 8));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((tid & 63) == 0) part[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    const float rs = 1.0f / scale;
    if (tid == 0) scales[row] = scale;
    unsigned char* qr = q + row * ldq;
    auto put = [&](int ch, const u32x4& v) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = div_by_scale(__uint_as_float(v[e] << 16), scale, rs);
            f[2 * e + 1] = div_by_scale(__uint_as_float(v[e] & 0xffff0000u), scale, rs);
        }
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        u32x2 o;
        o[0] = (unsigned)lo;
        o[1] = (unsigned)hi;
        *(u32x2*)(qr + ch * 8) = o;
    };
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ch = tid + i * 256;
        if (ch < nch) put(ch, kept[i]);
    }
    for (int ch = tid + KEEP * 256; ch < nch; ch += 256) put(ch, *(const u32x4*)(xr + ch * 8));
}

// ---------------------------------------------------------------------------------------------- GEMM
__global__ __launch_bounds__(512) void gemm_fp8_pp_kernel(Fp8GemmArgs p) {
    constexpr int BM = 256, BN = 256, BK = 128, ROWB = 128;
    constexpr int HALF = 128 * ROWB;
    constexpr int BUF = 4 * HALF;
    constexpr int MT = 8, NTL = 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const unsigned nwg = (unsigned)(p.ntm * p.ntn);
    const unsigned t = xcd_remap(blockIdx.x, nwg);
    const unsigned GROUP_M = (unsigned)p.group_m;
    const unsigned per_group = GROUP_M * p.ntn;
    const unsigned group = t / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = min((unsigned)p.ntm - first_m, GROUP_M);
    const unsigned tm = first_m + (t % per_group) % gsz;
    const unsigned tn = (t % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int sch = (lane & 7) ^ ((((wid & 1) << 2) + (lane >> 4)) & 7);
    unsigned soff[4][2];  // byte offsets
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid + 8 * i) * 8 + (lane >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (r >> 6) * 128 + h * 64 + (r & 63);
            const int bcol = (r >> 5) * 64 + h * 32 + (r & 31);
            soff[h][i] = (unsigned)((long)min(m0 + arow, p.M - 1) * p.lda + sch * 16);
            soff[2 + h][i] = (unsigned)((long)min(n0 + bcol, p.N - 1) * p.ldw + sch * 16);
        }
    }
    const int nk = p.K / BK;
    auto stage = [&](auto KIND, int kt, int buf) {
        constexpr int kind = decltype(KIND)::value;
        const unsigned char* base = ((kind < 2) ? p.A : p.W) + (long)kt * BK;
        char* dst = smem + buf * BUF + kind * HALF + wid * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + soff[kind][i]), (lds_void*)(dst + i * 8192), 16, 0, 0);
    };
    using K_A0 = std::integral_constant<int, 0>;
    using K_A1 = std::integral_constant<int, 1>;
    using K_B0 = std::integral_constant<int, 2>;
    using K_B1 = std::integral_constant<int, 3>;
    auto wait_halftiles = [&](int n) {
        if (n >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // lane (c = lane&15, g = lane>>4) reads bytes [32 g, 32 g + 32) of row c: 16-byte chunks 2g and 2g+1, swizzled like the staging
    int foff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) foff[h] = (lane & 15) * ROWB + ((((lane >> 4) * 2 + h) ^ ((lane >> 1) & 7)) << 4);
    const int a_rd = (wr * 64) * ROWB;
    const int b_rd = 2 * HALF + (wc * 32) * ROWB;

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x8 af[4], b0[2], b1[2];
    auto read_frag = [&](const char* base) {
        const u32x4 lo = *(const u32x4*)(base + foff[0]), hi = *(const u32x4*)(base + foff[1]);
        i32x8 r;
        r[0] = (int)lo[0]; r[1] = (int)lo[1]; r[2] = (int)lo[2]; r[3] = (int)lo[3];
        r[4] = (int)hi[0]; r[5] = (int)hi[1]; r[6] = (int)hi[2]; r[7] = (int)hi[3];
        return r;
    };
    auto read_a = [&](const char* cur, int h) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = read_frag(cur + a_rd + h * HALF + mi * 2048);
    };
    auto read_b = [&](i32x8 (&bf)[2], const char* cur, int h) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[ni] = read_frag(cur + b_rd + h * HALF + ni * 2048);
    };
    int one = 0x7F7F7F7F;  // E8M0 block scales 2^0 (a VGPR operand of the instruction)
    asm volatile("" : "+v"(one));
    auto mma_quadrant = [&](auto HA, auto HB, const i32x8 (&bf)[2]) {
        constexpr int ha = decltype(HA)::value, hb = decltype(HB)::value;
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                // inline asm pins the MFMAs between the two barriers (hipcc moves the builtin form across them and the two wave groups
                // lose their alternation) and accumulates in place; operands come from ds_read (waited above), results are first read
                // by compiler code after the loop's closing barriers
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                             : "+v"(acc[ha * 4 + mi][hb * 2 + ni])
                             : "v"(bf[ni]), "v"(af[mi]), "v"(one));
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    stage(K_A0{}, 0, 0);
    stage(K_B0{}, 0, 0);
    stage(K_B1{}, 0, 0);
    stage(K_A1{}, 0, 0);
    if (nk > 1) {
        stage(K_A0{}, 1, 1);
        stage(K_B0{}, 1, 1);
    }
    wait_halftiles(nk > 1 ? 4 : 2);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    const int last = 4 * nk - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* cur = smem + buf * BUF;
        const int g = 4 * kt;
        const bool steady = kt + 2 < nk;
        read_a(cur, 0);
        read_b(b0, cur, 0);
        if (kt + 1 < nk) stage(K_B1{}, kt + 1, buf ^ 1);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 6, last) - (g + 2));
        mma_quadrant(H0{}, H0{}, b0);
        read_b(b1, cur, 1);
        if (kt + 1 < nk) stage(K_A1{}, kt + 1, buf ^ 1);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 7, last) - (g + 3));
        mma_quadrant(H0{}, H1{}, b1);
        read_a(cur, 1);
        if (kt + 2 < nk) stage(K_A0{}, kt + 2, buf);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 8, last) - (g + 4));
        mma_quadrant(H1{}, H1{}, b1);
        if (kt + 2 < nk) stage(K_B0{}, kt + 2, buf);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(max(min(g + 9, last) - (g + 5), 0));
        mma_quadrant(H1{}, H0{}, b0);
    }
    asm volatile("s_nop 15" ::: "memory");  // asm MFMA results -> first compiler-generated reader
    if (wr == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue: x = acc * sa[row] * sw[col] (+ bias) -> bf16 (+ residual), 16-byte stores after one permlane16_swap per dword
    const int gq = lane >> 4, c = lane & 15;
    const int ncol0 = n0 + wc * 64;
    f32x4 swv[NTL];
    u32x2 bpk[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
        const int col = ncol0 + j * 16 + 4 * gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) swv[j][r] = p.sw[min(col + r, p.N - 1)];
        unsigned b0_ = 0, b1_ = 0;
        if (p.bias) {
            const unsigned a0 = p.bias[min(col, p.N - 1)], a1 = p.bias[min(col + 1, p.N - 1)], a2 = p.bias[min(col + 2, p.N - 1)], a3 = p.bias[min(col + 3, p.N - 1)];
            b0_ = a0 | (a1 << 16);
            b1_ = a2 | (a3 << 16);
        }
        bpk[j][0] = b0_;
        bpk[j][1] = b1_;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = m0 + wr * 128 + i * 16 + c;
        const float sar = p.sa[min(row, p.M - 1)];
        u32x2 pk[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] * sar * swv[j][r];
                if (p.bias) x += __uint_as_float((r & 1) ? (bpk[j][r >> 1] & 0xffff0000u) : (bpk[j][r >> 1] << 16));
                v[r] = x;
            }
            pk[j][0] = pack_bf2(v[0], v[1]);
            pk[j][1] = pack_bf2(v[2], v[3]);
        }
        const bool row_ok = row < p.M;
#pragma unroll
        for (int j = 0; j < NTL; j += 2) {
            const auto r0 = __builtin_amdgcn_permlane16_swap(pk[j][0], pk[j + 1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pk[j][1], pk[j + 1][1], false, false);
            u32x4 val = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            const int col = ncol0 + ((gq & 1) ? (j + 1) * 16 + 4 * (gq - 1) : j * 16 + 4 * gq);
            if (row_ok && col < p.N) {
                unsigned short* dst = p.C + (long)row * p.ldc + col;
                const bool vec = (col + 8 <= p.N) && ((p.ldc & 7) == 0);
                if (p.res) {
                    const unsigned short* rs = p.res + (long)row * p.ldr + col;
                    u32x4 rv;
                    if (vec && ((p.ldr & 7) == 0) && ((((size_t)p.res) & 15) == 0)) {
                        rv = *(const u32x4*)rs;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned lo = (col + 2 * e < p.N) ? rs[2 * e] : 0u, hi = (col + 2 * e + 1 < p.N) ? rs[2 * e + 1] : 0u;
                            rv[e] = lo | (hi << 16);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        val[e] = pack_bf2(__uint_as_float(val[e] << 16) + __uint_as_float(rv[e] << 16),
                                          __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rv[e] & 0xffff0000u));
                }
                if (vec) {
                    *(u32x4*)dst = val;
                } else {
                    for (int e = 0; e < 8 && col + e < p.N; ++e) dst[e] = (unsigned short)(val[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
    }
}

// ---- the row quantiser fused into the producers of the two widest activations of a decoder layer (config 5): the SwiGLU output a = silu(g) * u
//      [T, I] (consumed only by the down projection) and its backward dgu [T, 2I] (consumed only by the dX contraction through gate/up).  Unfused, each
//      is written as bf16 and read back by quant_fp8_rows_kernel (3 x 2 bytes per element of traffic); fused, only the e4m3 bytes + one scale per row
//      leave the kernel.  One workgroup per row, two passes over the (L2-resident) inputs: amax of the bf16-ROUNDED values, then the same values quantised
//      -- bit-identical to swiglu_{fwd,bwd}_kernel followed by quant_fp8_rows_kernel.
__device__ __forceinline__ void unpack8f(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ float bfround(float x) { return __uint_as_float(((unsigned)f2bf(x)) << 16); }
__device__ __forceinline__ u32x2 quant8(const float* f, float scale) {
    const float r = 1.0f / scale;
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(div_by_scale(f[0], scale, r), div_by_scale(f[1], scale, r), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(div_by_scale(f[2], scale, r), div_by_scale(f[3], scale, r), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(div_by_scale(f[4], scale, r), div_by_scale(f[5], scale, r), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(div_by_scale(f[6], scale, r), div_by_scale(f[7], scale, r), hi, true);
    u32x2 o;
    o[0] = (unsigned)lo;
    o[1] = (unsigned)hi;
    return o;
}
__device__ __forceinline__ float block_amax(float amax, float* part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax;
    __syncthreads();
    return fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

__global__ __launch_bounds__(256) void swiglu_fwd_quant_kernel(const unsigned short* __restrict__ gu, unsigned char* __restrict__ q, float* __restrict__ scales,
                                                               long I) {
    __shared__ float part[4];
    const long t = blockIdx.x;
    const int nch = (int)(I / 8);
    const unsigned short* gr = gu + t * 2 * I;
    auto values = [&](int ch, float* o) {   // swiglu_fwd_kernel's arithmetic, result rounded to bf16 as its store does
        float g[8], u[8];
        const long goff = (long)(ch / 2) * 32 + (ch % 2) * 8;
        unpack8f(*(const u32x4*)(gr + goff), g);
        unpack8f(*(const u32x4*)(gr + goff + 16), u);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bfround(swiglu_fwd_elem(g[e], u[e]));
    };
    // The row is walked ONCE: a thread's first KEEP chunks stay in registers as packed bf16 (the amax pass rounds them anyway) and are quantised from there --
    // the second evaluation of silu (exp + reciprocal per element) was 105 of this kernel's 125 us at I = 18 944.  Chunks beyond KEEP (I > 20 480) are recomputed.
    constexpr int KEEP = 10;
    u32x4 kept[KEEP];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ch = threadIdx.x + i * 256;
        kept[i] = u32x4{0u, 0u, 0u, 0u};
        if (ch < nch) {
            float o[8];
            values(ch, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
#pragma unroll
            for (int e = 0; e < 4; ++e) kept[i][e] = pack_bf2(o[2 * e], o[2 * e + 1]);   // exact: the values are bf16 already
        }
    }
    for (int ch = threadIdx.x + KEEP * 256; ch < nch; ch += 256) {
        float o[8];
        values(ch, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(o[e]));
    }
    amax = block_amax(amax, part);
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scales[t] = scale;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
            float o[8];
            unpack8f(kept[i], o);
            *(u32x2*)(q + t * I + ch * 8) = quant8(o, scale);
        }
    }
    for (int ch = threadIdx.x + KEEP * 256; ch < nch; ch += 256) {
        float o[8];
        values(ch, o);
        *(u32x2*)(q + t * I + ch * 8) = quant8(o, scale);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_quant_kernel(const unsigned short* __restrict__ gu, const unsigned short* __restrict__ da,
                                                               unsigned char* __restrict__ q, float* __restrict__ scales, long I) {
    __shared__ float part[4];
    const long t = blockIdx.x;
    const int nch = (int)(I / 8);
    const unsigned short* gr = gu + t * 2 * I;
    const unsigned short* dr = da + t * I;
    auto values = [&](int ch, float* dg, float* du) {   // swiglu_bwd_kernel's arithmetic, rounded to bf16 as its stores do
        float g[8], u[8], d[8];
        const long goff = (long)(ch / 2) * 32 + (ch % 2) * 8;
        unpack8f(*(const u32x4*)(gr + goff), g);
        unpack8f(*(const u32x4*)(gr + goff + 16), u);
        unpack8f(*(const u32x4*)(dr + ch * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            swiglu_bwd_elem(g[e], u[e], d[e], dg[e], du[e]);
            dg[e] = bfround(dg[e]);
            du[e] = bfround(du[e]);
        }
    };
    constexpr int KEEP = 10;   // (as in the forward twin: one evaluation per element, kept as packed bf16)
    u32x4 kg[KEEP], ku[KEEP];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ch = threadIdx.x + i * 256;
        kg[i] = u32x4{0u, 0u, 0u, 0u};
        ku[i] = u32x4{0u, 0u, 0u, 0u};
        if (ch < nch) {
            float dg[8], du[8];
            values(ch, dg, du);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fmaxf(fabsf(dg[e]), fabsf(du[e])));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kg[i][e] = pack_bf2(dg[2 * e], dg[2 * e + 1]);
                ku[i][e] = pack_bf2(du[2 * e], du[2 * e + 1]);
            }
        }
    }
    for (int ch = threadIdx.x + KEEP * 256; ch < nch; ch += 256) {
        float dg[8], du[8];
        values(ch, dg, du);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fmaxf(fabsf(dg[e]), fabsf(du[e])));
    }
    amax = block_amax(amax, part);
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scales[t] = scale;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int ch = threadIdx.x + i * 256;
        if (ch < nch) {
            float dg[8], du[8];
            unpack8f(kg[i], dg);
            unpack8f(ku[i], du);
            const long goff = (long)(ch / 2) * 32 + (ch % 2) * 8;
            *(u32x2*)(q + t * 2 * I + goff) = quant8(dg, scale);
            *(u32x2*)(q + t * 2 * I + goff + 16) = quant8(du, scale);
        }
    }
    for (int ch = threadIdx.x + KEEP * 256; ch < nch; ch += 256) {
        float dg[8], du[8];
        values(ch, dg, du);
        const long goff = (long)(ch / 2) * 32 + (ch % 2) * 8;
        *(u32x2*)(q + t * 2 * I + goff) = quant8(dg, scale);
        *(u32x2*)(q + t * 2 * I + goff + 16) = quant8(du, scale);
    }
}

}  // namespace rga3

using namespace rga3;

extern "C" int rga3_quant_fp8_rows(const void* x, void* q, float* scales, int64_t rows, int64_t K, int64_t ldx, int64_t ldq, void* stream) {
    RGA3_CHECK_ARG(x && q && scales && rows > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0, "quant_fp8_rows: rows=%ld K=%ld", (long)rows, (long)K);
    RGA3_CHECK_ARG(rows <= 0x7fffffff, "quant_fp8_rows: too many rows");
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned char*)q, scales,
                       (long)rows, (int)K, (long)ldx, (long)ldq);
    RGA3_CHECK_LAUNCH("quant_fp8_rows_kernel");
    return 0;
}

extern "C" int rga3_gemm_fp8(const void* Aq, const void* Wq, const float* sa, const float* sw, const void* bias, const void* residual, void* C,
                             int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, void* stream) {
    RGA3_CHECK_ARG(Aq && Wq && sa && sw && C, "gemm_fp8: null pointer");
    RGA3_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8: M=%ld N=%ld K=%ld (K must be a multiple of 128)", (long)M, (long)N, (long)K);
    RGA3_CHECK_ARG(lda % 16 == 0 && ldw % 16 == 0, "gemm_fp8: lda/ldw must be multiples of 16 bytes");
    RGA3_CHECK_ARG((((uintptr_t)Aq | (uintptr_t)Wq | (uintptr_t)C) & 15) == 0, "gemm_fp8: pointers must be 16-byte aligned");
    RGA3_CHECK_ARG(M * lda < (1LL << 32) && N * ldw < (1LL << 32), "gemm_fp8: operands must be < 2^32 bytes (32-bit staging offsets)");
    Fp8GemmArgs a;
    a.A = (const unsigned char*)Aq; a.W = (const unsigned char*)Wq; a.sa = sa; a.sw = sw;
    a.bias = (const unsigned short*)bias; a.res = (const unsigned short*)residual; a.C = (unsigned short*)C;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.ntm = (int)cdiv(M, 256);
    a.ntn = (int)cdiv(N, 256);
    a.group_m = a.ntm <= 16 ? a.ntm : 4;
    constexpr int LDS = 2 * 4 * 128 * 128;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)gemm_fp8_pp_kernel, LDS, lds_grant, "gemm_fp8")) return rc;
    hipLaunchKernelGGL(gemm_fp8_pp_kernel, dim3((unsigned)(a.ntm * a.ntn)), dim3(512), LDS, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("gemm_fp8_pp_kernel");
    return 0;
}

extern "C" int rga3_swiglu_fwd_quant_fp8(const void* gu, void* q, float* scales, int64_t T, int64_t I, void* stream) {
    RGA3_CHECK_ARG(gu && q && scales && T > 0 && T <= 0x7fffffff && I > 0 && I % 16 == 0, "swiglu_fwd_quant_fp8: T=%ld I=%ld", (long)T, (long)I);
    hipLaunchKernelGGL(swiglu_fwd_quant_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)gu, (unsigned char*)q, scales, (long)I);
    RGA3_CHECK_LAUNCH("swiglu_fwd_quant_kernel");
    return 0;
}

extern "C" int rga3_swiglu_bwd_quant_fp8(const void* gu, const void* da, void* q, float* scales, int64_t T, int64_t I, void* stream) {
    RGA3_CHECK_ARG(gu && da && q && scales && T > 0 && T <= 0x7fffffff && I > 0 && I % 16 == 0, "swiglu_bwd_quant_fp8: T=%ld I=%ld", (long)T, (long)I);
    hipLaunchKernelGGL(swiglu_bwd_quant_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)gu, (const unsigned short*)da,
                       (unsigned char*)q, scales, (long)I);
    RGA3_CHECK_LAUNCH("swiglu_bwd_quant_kernel");
    return 0;
}
