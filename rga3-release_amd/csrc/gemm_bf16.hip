// bf16 "NT" GEMM on CDNA4 MFMA:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Replaces every nn.Linear / 1x1-conv / patch-embed contraction the reference reaches through
// cuBLAS (SURVEY.md 2.2 rows K1,K3,K4,K5,K8,K9,K10,K12,K14,K16,K18; e.g. HF
// modeling_qwen2_5_vl.py:84-96,211-291,602-757 and reference model/sam2.py:986-1117).
//
// Design (MI355X-first, not a port of any CUDA tiling):
//  * nn.Linear weights are [N,K] row-major, activations [M,K] row-major: BOTH operands are K-contiguous,
//    which is exactly the per-lane fragment order of v_mfma_f32_16x16x32_bf16 (8 consecutive k per lane).
//  * Tiles go HBM -> LDS with 16-byte global_load_lds (no VGPR round trip).  The LDS image is lane-linear
//    (hardware rule), so the bank-conflict XOR swizzle is applied to the per-lane SOURCE address and
//    again on the ds_read_b128 side (cdna_hip_programming.md 5.4 rule 21).  Swizzle used:
//    phys_chunk = chunk ^ ((row>>1)&7) on 128-byte rows: every 16-lane ds_read_b128 group lands on 16
//    distinct 16-byte slots of the 256-byte bank row (conflict-free).
//  * MFMA operands are swapped (W fragment as "A", activation fragment as "B") so a lane ends up with 4
//    CONSECUTIVE output columns of one row: epilogue math stays lane-local and the bf16 pack is 8 B/lane.
//  * The accumulator tile is transposed through a small per-wave LDS buffer so global stores are
//    16 B/lane, 128 B contiguous per row; the residual add is fused there with coalesced loads.
//  * 1-D grid with a bijective XCD remap + grouped tile order so neighbouring tiles share an L2.
#include "common.h"
#include "act_table.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace rga3 {


enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SWIGLU = 2, ACT_RELU = 3 };

struct GemmArgs {
    const unsigned short* A;
    const unsigned short* W;
    void* C;
    const unsigned short* bias;  // [N] (for SWIGLU: [N], interleaved like W) or null
    const unsigned short* res;   // [M, Nout] bf16 or null
    const unsigned short* colscale;  // [Nout] bf16 or null: out = res + colscale[n] * act(acc + bias)  (ConvNeXt layer scale)
    int M, N, K;
    long lda, ldw, ldc, ldr;
    int ntm, ntn;
    int group_m;  // tile rows per group of the tile order (see launchers)
    void* ws;     // caller workspace for the stream-K / split-K tilings (may be null)
    long ws_bytes;
    const float* rowstat;  // LNF epilogues: [M][2] = (mean, 1/sqrt(var + eps)) of the rows of A (rga3_layernorm_stats)
    const float* colc;     // LNF epilogues: [N] column sums of the gamma-folded weight
    int ksl;      // gemm_nt_kernel only: K-tiles per grid.y slice (0: no split); slice y accumulates K-tiles [y ksl, (y+1) ksl) into f32 slab y of C
    // RMSNorm folded into producer and consumer (rga3_gemm_rms_bf16): row sums of squares as 64-bit FIXED-POINT integers (2^20 per unit), so the cross-tile
    // sum is a no-return integer atomic add -- associative, hence bitwise reproducible, unlike a float atomic.
    const unsigned long long* rs_in = nullptr;   // [M]: sum_k A[r][k]^2 of the consumer's rows (A un-normalised, W with the norm weight folded in); null = off
    float rs_in_scale = 0.f;                     // 2^-20 / (width of the normalised rows)
    float rs_eps = 0.f;
    unsigned long long* rs_out = nullptr;        // [M]: the output rows' sums of squares (of the bf16 values written) are ADDED here; null = off
    // LayerNorm statistics out of the producer's epilogue (rga3_gemm_lnsum_bf16 / rga3_gemm_lnq_bf16; Hiera's MultiScaleBlock, reference sam2.py:1085-1117).  Kernels
    // instantiated with LNM = 2 write, per output row and TILE COLUMN, (sum v, sum v^2) of the bf16 values that tile wrote: ln_parts [M][ntn][2] f32, plain stores
    // (the waves of a workgroup combine through LDS in a fixed order: no atomics, nothing to zero, bitwise reproducible).  The LayerNorm-folded consumer (LNM = 1) adds
    // the ln_ns partials of a row in f64 and derives (mean, 1 / sqrt(var + eps)) itself -- the stand-alone statistics pass over the rows (rga3_layernorm_stats)
    // disappears.  (First form of this round: 64-bit fixed-point atomics per wave and row as rs_out does -- 1.6 M atomics per Hiera product cost more than the pass
    // they replaced: DESIGN.md.)
    float* ln_parts = nullptr;                   // producer: [M][ntn][2]
    const float* ln_in = nullptr;                // consumer: [M][ln_ns][2]; 1 / width and eps travel in rs_in_scale / rs_eps
    int ln_ns = 0;
    // Concatenated operands (rga3_gemm_cat_bf16, single-phase kernels only): LoRA's low-rank products folded into the frozen ones.
    //   K side:  C = [A | A2] . [W | W2]^T   -- K-tiles [0, K / 64) come from (A, W), the next K2 / 64 from (A2, W2)          (y = x W^T + t (sB)^T in one product)
    //   N side:  tile columns at or beyond N belong to a SECOND weight / output pair: Cn [M, N2] = A . Wn^T (no bias, residual or K side there)   ([dx | dt] = dy [W | sB])
    const unsigned short* A2 = nullptr;
    const unsigned short* W2 = nullptr;
    long lda2 = 0, ldw2 = 0;
    int K2 = 0;
    unsigned short* pre = nullptr;   // SwiGLU epilogue (rga3_gemm_swiglu_pre_bf16): also store the bf16 gate | up pre-activations [M, N] (interleaved 16-column blocks, row stride ldpre)
    long ldpre = 0;
    const unsigned short* Wn = nullptr;
    void* Cn = nullptr;
    long ldwn = 0, ldcn = 0;
    int N2 = 0;
};

constexpr float kRowSumFix = 1048576.0f;   // 2^20

// 16 bytes of zeros in device memory: source for staging chunks that lie beyond K in the last K-tile
__device__ const u32x4 g_zero16 = {0u, 0u, 0u, 0u};

// exact-erf GELU, erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding of the result): one v_rcp, one
// v_exp and seven FMAs instead of libm's erff (~40 VALU instructions, 128 of them per lane in a 256x256 epilogue: +16 us per tile)
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __expf(-z * z);   // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * e;               // x * (1 + sign(x) erf) / 2
}
// x * sigmoid(x) with the hardware reciprocal (1 ulp; the result is rounded to bf16 right after): the IEEE division here was a ten-instruction sequence
// (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup) on each of the 64 outputs a lane owns in a 256 x 256 SwiGLU epilogue -- the matrix pipe idles
// while it runs
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int ACT> constexpr bool act_uses_table() { return ACT == ACT_GELU || ACT == ACT_SWIGLU; }

// (s, s) in a REAL register pair.  Left to itself the compiler broadcasts a scalar into packed-f32 operations through op_sel (one half of a pair feeding both lanes of
// the operation); the epilogue form  v_pk_mul_f32 t, c, v[n:n+1] op_sel_hi:[1,0]  +  v_pk_fma_f32 d, acc, v[m:m+1], t op_sel:[0,1,0]  returned d.lo WITHOUT the product term in
// lanes 48-63 of a few percent of the waves when two workgroups shared a CU (MI355X, ROCm 7.2; timing-dependent, never with one workgroup per CU; the element-wise and
// the plain-pair forms of the same arithmetic are exact; tools/probes/dbg_ln_gelu.py reproduces it with -DEPI_V3).  Every packed operation of the epilogue therefore
// takes plain pairs.
__device__ __forceinline__ f32x2 splat2(float s) {
    f32x2 v = {s, s};
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ f32x2 widen_bf2(unsigned pk) { return f32x2{__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)}; }
// the workgroup's NWAVES waves copy table `which` (0 GELU, 1 SiLU) into LDS at dst by LDS-DMA; the caller waits (vmcnt) and synchronises before the epilogue reads it
template <int NWAVES>
__device__ __forceinline__ void stage_act_table(char* dst, int which, int wid, int lane) {
    const char* src = (const char*)g_act_tab[which] + lane * 16;
#pragma unroll
    for (int pc = 0; pc < kActTabBytes / 1024; pc += NWAVES) {
        const int q = pc + wid;   // wave-uniform
        if (q < kActTabBytes / 1024) __builtin_amdgcn_global_load_lds((gbl_void*)(src + q * 1024), (lds_void*)(dst + q * 1024), 16, 0, 0);
    }
}

// ---- epilogue shared by all main loops.  Lane (g = lane>>4, c = lane&15) holds, for m-tile i / n-tile j of the wave's
//      WTM x WTN block, row c and columns 4g..4g+3 of the 16x16 sub-block (swapped-operand MFMA).
template <int NTL, int ACT, bool OUT_F32>
constexpr int epi_wave_bytes() {  // LDS bytes one wave parks per-row statistics in during gemm_epilogue (<= 128 rows x (mean, 1/std)); outputs go to memory straight from registers
    return 1024;
}

// LNF (LayerNorm folded into the product, rga3_gemm_ln_bf16): A holds the UN-normalised rows x, W the weight with gamma folded in (W' = W . diag(gamma)), and
//   LN(x) W^T + b  =  rinv_r (x W'^T - mean_r c_n) + d_n,   c_n = sum_k W'_nk,   d_n = sum_k beta_k W_nk + b_n  (handed over as the bias),
// so the normalised activations are never written or re-read: a row-statistics pass (one read of x) replaces the LayerNorm pass (read + write).
// ---- residual rows through LDS.  The epilogue's residual loads sat inside the m-tile loop: every m-tile paid an HBM / Infinity-Cache round trip in line (8 per
//      128-row wave tile: 31 000 of a 133 000-cycle Hiera fc2 item, tools/probes/sk_items.py).  The wave's whole residual block is instead fetched by LDS-DMA in one
//      burst before the loop (no registers, one round trip) and read back with ds_read_b128.  Image: MT*16 rows of OW/8 + 1 sixteen-byte slots (one pad slot per row:
//      the 16 lanes that read one chunk column of 16 consecutive rows then hit 16 different bank groups); slot s of the lane-linear DMA image = (row s / CPRP, chunk
//      s % CPRP), the source address is per lane.  Each wave reads only what it staged itself: its own vmcnt wait is all the synchronisation needed.
// ping-pong kernel: residual staging for the plain bf16 epilogue (8 waves x 128 x 64 blocks = 147 KiB + 8 KiB parking: more than the two 64-KiB buffers, so the launcher
// asks for it; one workgroup per CU either way)
template <int ACT, bool OUT_F32>
constexpr bool pp_res_lds() { return ACT == ACT_NONE && !OUT_F32; }
template <int MT, int OW>
constexpr int res_stage_bytes() { return ((MT * 16 * (OW / 8 + 1) * 16 + 1023) / 1024) * 1024; }

template <int MT, int OW>
__device__ __forceinline__ void stage_residual(const GemmArgs& p, char* dst, int row0, int col0, int lane) {
    constexpr int CPR = OW / 8, CPRP = CPR + 1, SLOTS = MT * 16 * CPRP, NI = (SLOTS + 63) / 64;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int sl = min(k * 64 + lane, SLOTS - 1);
        const int row = sl / CPRP, ch = min(sl - row * CPRP, CPR - 1);
        const unsigned short* src = p.res + (long)min(row0 + row, p.M - 1) * p.ldr + col0 + ch * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(dst + k * 1024), 16, 0, 0);
    }
}
// wave-uniform: the wave's OW output columns lie inside the matrix and the residual rows can be fetched in aligned 16-byte pieces
template <int OW, bool OUT_F32>
__device__ __forceinline__ bool residual_stageable(const GemmArgs& p, int col0, int Nout) {
    return !OUT_F32 && p.res && col0 + OW <= Nout && (p.ldr & 7) == 0 && (p.ldc & 7) == 0 && ((((size_t)p.res) & 15) == 0);
}

// PARTS: part1 / part2 are read (stream-K owner slices); tab: this kernel's activation table in LDS (act_tab; GELU / SwiGLU epilogues only).
template <int MT, int NTL, int WTM, int WTN, int ACT, bool OUT_F32, int LNM = 0, bool PARTS = false>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MT][NTL], const GemmArgs& p, char* est, const char* tab, int lane, int m0, int n0,
                                              int wm, int wn, const f32x4* part1 = nullptr, const f32x4* part2 = nullptr, const char* resl = nullptr,
                                              float poison = 0.f, int nwn = 1) {
    // LNM: 0 plain, 1 = LNF, the LayerNorm-folded CONSUMER (rowstat / ln_in, colc), 2 = LayerNorm-sum PRODUCER (rs_out is an [M][2] pair array: rga3_gemm_lnsum_bf16).
    // Compile-time, so the kernels of every other product keep the epilogue they had.
    constexpr bool LNF = LNM == 1, LNP = LNM == 2;
    static_assert(!(LNM != 0 && ACT == ACT_SWIGLU), "the LayerNorm-folded epilogues have no SwiGLU form");
    static_assert(!(LNP && (OUT_F32 || PARTS)), "the LayerNorm-sum producer writes bf16 from the single-pass kernels");
    // part1 / part2 (stream-K owner slices only): this lane's view of up to two f32 partial-sum slabs in accumulator order
    // (quad (i, j) at [(i * NTL + j) * 64]).  They are added to the accumulators as those are READ, one m-tile ahead of use,
    // so the accumulator registers are never modified after the main loop (a post-loop "acc += slab" makes the register
    // allocator keep two copies of the 128 accumulators and spill).
    constexpr int PD = 1;  // m-tiles of slab reads in flight ahead of use (2 would hide more latency but spills in the persistent kernel)
    f32x4 pn[PD][NTL];
    auto load_part = [&](int i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            f32x4 v = part1[(i * NTL + j) * 64];
            if (part2) v += part2[(i * NTL + j) * 64];
            // poison = +inf when the owner gave up waiting for a contributor (0 otherwise): the tile leaves as non-finite values (every epilogue form keeps +inf
            // non-finite, ReLU included) instead of a finite sum that lacks a slice -- a wrong product must not look like a right one
            v += f32x4{poison, poison, poison, poison};
            pn[i % PD][j] = v;
        }
    };
    if constexpr (PARTS) {
        load_part(0);
        if (PD > 1 && MT > 1) load_part(1);
    }
    // bias for this lane's 4 columns of every n-tile, loaded ONCE (packed bf16 x4 per n-tile); per-element
    // 2-byte loads inside the m-tile loop cost +9 us per 256x256 tile
    auto load4 = [&](const unsigned short* v, int col, int n) -> u32x2 {
        u32x2 r;
        if (col + 3 < n && (((size_t)(v + col)) & 7) == 0) {
            r = *(const u32x2*)(v + col);
        } else {
            const unsigned a0 = v[min(col, n - 1)], a1 = v[min(col + 1, n - 1)], a2 = v[min(col + 2, n - 1)], a3 = v[min(col + 3, n - 1)];
            r[0] = a0 | (a1 << 16);
            r[1] = a2 | (a3 << 16);
        }
        return r;
    };
    auto pick = [](const u32x2& pk, int r) -> float { return __uint_as_float((r & 1) ? (pk[r >> 1] & 0xffff0000u) : (pk[r >> 1] << 16)); };
    u32x2 bpk[NTL];
    f32x4 cc[LNF ? NTL : 1];
    {
        const int g4 = (lane >> 4) * 4;
        if (p.bias) {
            if (n0 + wn * WTN + NTL * 16 <= p.N && (((size_t)p.bias) & 7) == 0) {   // wave-uniform: no per-load branch, the NTL loads fly together
#pragma unroll
                for (int j = 0; j < NTL; ++j) bpk[j] = *(const u32x2*)(p.bias + n0 + wn * WTN + j * 16 + g4);
            } else {
#pragma unroll
                for (int j = 0; j < NTL; ++j) bpk[j] = load4(p.bias, n0 + wn * WTN + j * 16 + g4, p.N);
            }
        }
        if constexpr (LNF) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int col = n0 + wn * WTN + j * 16 + g4;
                cc[j] = (col + 3 < p.N) ? *(const f32x4*)(p.colc + col) : f32x4{0.f, 0.f, 0.f, 0.f};   // N % 4 == 0 (checked by the entry point)
            }
        }
    }

    // ---- epilogue.  Lane (g = lane>>4, c = lane&15) holds, for m-tile i / n-tile j, row c and
    //      columns 4g..4g+3 of the 16x16 block.
    constexpr int OUT_NT = (ACT == ACT_SWIGLU) ? NTL / 2 : NTL;  // output n-tiles per wave
    constexpr int OW = OUT_NT * 16;                               // output columns per wave
    constexpr int ESZ = OUT_F32 ? 4 : 2;
    constexpr int EDAT = OW * ESZ;  // payload bytes per staged row
    constexpr int EROW = EDAT + 16; // +16 B pad: rows no longer alias on the 128-B ds_write bank period (was 16-way conflicts)
    const int g = lane >> 4, c = lane & 15;
    const int ncol0 = (ACT == ACT_SWIGLU) ? (n0 / 2 + wn * OW) : (n0 + wn * OW);  // first output column of this wave
    const int Nout = (ACT == ACT_SWIGLU) ? p.N / 2 : p.N;

    // RMSNorm of the A rows folded in: 1 / sqrt(mean(x^2) + eps) from the producer's fixed-point row sums.  The wave's MT x 16 rows are fetched by lane
    // (row lane, row lane + 64), all loads in flight together, and parked in the wave's epilogue staging bytes (unused since the stores go straight from
    // registers): inside the m-tile loop every load's L2 round trip was in line (+12 us on the LLM gate-up product), and 8 long-lived registers per lane
    // pushed the stream-K epilogue into scratch.
    float* const rs_lds = (float*)est;
    const unsigned tb = act_uses_table<ACT>() ? act_tab_base(tab) : 0u;
    if constexpr (LNF) {   // the same for the LayerNorm-folded products: (mean, 1 / std) of the wave's rows fetched up front, not one L2 round trip per m-tile in line
        if (p.ln_in) {     // from the producer's per-tile-column partial sums: mean and variance in f64 (E[x^2] - mean^2 cancels; a handful of f64 operations per ROW)
            const double sc = (double)p.rs_in_scale;
#pragma unroll
            for (int q = 0; q < (MT * 16 + 63) / 64; ++q) {
                const int rl = lane + 64 * q;
                if (rl < MT * 16) {
                    const float2* pp = (const float2*)p.ln_in + (long)min(m0 + wm * WTM + rl, p.M - 1) * p.ln_ns;
                    double sx = 0.0, sxx = 0.0;
                    for (int k = 0; k < p.ln_ns; ++k) {
                        const float2 v = pp[k];
                        sx += (double)v.x;
                        sxx += (double)v.y;
                    }
                    const double mean = sx * sc;
                    const double var = fmax(sxx * sc - mean * mean, 0.0);
                    *(float2*)(rs_lds + 2 * rl) = make_float2((float)mean, __builtin_amdgcn_rsqf((float)var + p.rs_eps));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < (MT * 16 + 63) / 64; ++q) {
                const int rl = lane + 64 * q;
                if (rl < MT * 16) *(float2*)(rs_lds + 2 * rl) = *(const float2*)(p.rowstat + 2L * min(m0 + wm * WTM + rl, p.M - 1));
            }
        }
    }
    if (p.rs_in) {
#pragma unroll
        for (int q = 0; q < (MT * 16 + 63) / 64; ++q) {
            const int rl = lane + 64 * q;
            if (rl < MT * 16) {
                const uint2 raw = *(const uint2*)(p.rs_in + min(m0 + wm * WTM + rl, p.M - 1));
                // u64 -> f32 by halves (the generic conversion is a ~15-instruction emulation)
                rs_lds[rl] = __builtin_amdgcn_rsqf(__builtin_fmaf((float)raw.y, 4294967296.0f, (float)raw.x) * p.rs_in_scale + p.rs_eps);
            }
        }
    }
    if (resl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's staged residual rows have landed (plain LDS reads do not wait for LDS-DMA by themselves)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // ---- values of this m-tile, whole quads at a time (f32x4 arithmetic lowers to v_pk_mul / v_pk_fma / v_pk_add: two elements per VALU slot).  Optional steps
        //      sit behind ONE wave-uniform branch per m-tile, not per element: per-element forms compiled to both-sides-plus-select (bias) or a scalar branch in
        //      every element (column scale), and the unconditional "+ partial sum" of the stream-K owner was an add of 0.0 in every other tile (not foldable: -0.0).
        //      The epilogue is VALU time the matrix pipe idles through (r04 PMC: 70 % VALU-busy in the K = 576 GELU product).
        f32x4 x[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            x[j] = acc[i][j];
            if constexpr (PARTS) x[j] += pn[i % PD][j];
        }
        if constexpr (PARTS) {
            if (i + PD < MT) load_part(i + PD);
        }
        if constexpr (LNF) {
            const float2 st2 = *(const float2*)(rs_lds + 2 * (i * 16 + (lane & 15)));
            const float ln_rinv = st2.y, nmr = -st2.x * st2.y;   // rinv (x - mean c) = rinv x + (-mean rinv) c
            const f32x2 ri2 = splat2(ln_rinv), nm2 = splat2(nmr);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const f32x2 lo = x[j].xy * ri2 + cc[j].xy * nm2, hi = x[j].zw * ri2 + cc[j].zw * nm2;
                x[j] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        } else if (p.rs_in) {
            const float rs_rinv = rs_lds[i * 16 + (lane & 15)];   // same wave wrote it: program order + the compiler's lgkmcnt suffice
            const f32x2 rs2 = splat2(rs_rinv);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const f32x2 lo = x[j].xy * rs2, hi = x[j].zw * rs2;
                x[j] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                x[j] += f32x4{__uint_as_float(bpk[j][0] << 16), __uint_as_float(bpk[j][0] & 0xffff0000u), __uint_as_float(bpk[j][1] << 16),
                              __uint_as_float(bpk[j][1] & 0xffff0000u)};
        }
        f32x4 v[OUT_NT];
#pragma unroll
        for (int jo = 0; jo < OUT_NT; ++jo) {
            if constexpr (ACT == ACT_SWIGLU) {
                // packed weight layout: n-tile 2*jo = gate columns, 2*jo+1 = up columns of the same 16 outputs.  The reference rounds the gate / up linear outputs
                // to bf16 before the activation (bf16 nn.Linear), and silu(gate) before the product
                const f32x4 &gq = x[2 * jo], &uq = x[2 * jo + 1];
                const unsigned pg0 = pack_bf2(gq[0], gq[1]), pg1 = pack_bf2(gq[2], gq[3]), pu0 = pack_bf2(uq[0], uq[1]), pu1 = pack_bf2(uq[2], uq[3]);
                const f32x2 s01 = act_tab2(pg0, tb), s23 = act_tab2(pg1, tb);
                const f32x2 v01 = widen_bf2(pack_bf2(s01[0], s01[1])) * widen_bf2(pu0);
                const f32x2 v23 = widen_bf2(pack_bf2(s23[0], s23[1])) * widen_bf2(pu1);
                v[jo] = f32x4{v01[0], v01[1], v23[0], v23[1]};
                if (p.pre) {   // training: the backward needs the rounded pre-activations (autograd of silu(gate) * up); same 16-byte row pieces as the output stores
                    const auto q0 = __builtin_amdgcn_permlane16_swap(pg0, pu0, false, false);
                    const auto q1 = __builtin_amdgcn_permlane16_swap(pg1, pu1, false, false);
                    const u32x4 pv = {(unsigned)q0[0], (unsigned)q1[0], (unsigned)q0[1], (unsigned)q1[1]};
                    const int prow = m0 + wm * WTM + i * 16 + (lane & 15);
                    const int pcol = n0 + wn * WTN + ((g & 1) ? (2 * jo + 1) * 16 + 4 * (g - 1) : 2 * jo * 16 + 4 * g);
                    if (prow < p.M && pcol < p.N) {
                        unsigned short* dp = p.pre + (long)prow * p.ldpre + pcol;
                        if (pcol + 8 <= p.N && (p.ldpre & 7) == 0) {
                            *(u32x4*)dp = pv;
                        } else {
                            for (int e = 0; e < 8 && pcol + e < p.N; ++e) dp[e] = (unsigned short)(pv[e >> 1] >> (16 * (e & 1)));
                        }
                    }
                }
            } else if constexpr (ACT == ACT_GELU) {
                const f32x2 y01 = act_tab2(pack_bf2(x[jo][0], x[jo][1]), tb), y23 = act_tab2(pack_bf2(x[jo][2], x[jo][3]), tb);
                v[jo] = f32x4{y01[0], y01[1], y23[0], y23[1]};
            } else if constexpr (ACT == ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jo][r] = fmaxf(x[jo][r], 0.f);
            } else {
                v[jo] = x[jo];
            }
        }
        if (p.colscale) {   // rare (ConvNeXt layer scale)
#pragma unroll
            for (int jo = 0; jo < OUT_NT; ++jo)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float cs = bf2f(p.colscale[min(ncol0 + jo * 16 + 4 * g + r, Nout - 1)]);
                    v[jo][r] = ((ACT == ACT_SWIGLU) ? v[jo][r] : bf2f(f2bf(v[jo][r]))) * cs;
                }
        }
        u32x2 pk[OUT_NT];
        f32x4 vf[OUT_F32 ? OUT_NT : 1];
#pragma unroll
        for (int jo = 0; jo < OUT_NT; ++jo) {
            if constexpr (OUT_F32) {
                vf[jo] = v[jo];
            } else {
                pk[jo][0] = pack_bf2(v[jo][0], v[jo][1]);
                pk[jo][1] = pack_bf2(v[jo][2], v[jo][3]);
            }
        }
        // ---- stores straight from registers (no LDS round trip).  f32: the lane's 4 columns are 16 B already.  bf16: one
        //      v_permlane16_swap per dword hands the odd 16-lane rows' quads of n-tile jo to the even rows and the even rows'
        //      quads of n-tile jo+1 to the odd rows, so every lane owns 8 consecutive columns (16 B) of one row:
        //      a wave-store covers 16 rows x 64 contiguous bytes.
        const int row = m0 + wm * WTM + i * 16 + c;
        const bool row_ok = row < p.M;
        if constexpr (OUT_F32) {
#pragma unroll
            for (int jo = 0; jo < OUT_NT; ++jo) {
                const int col = ncol0 + jo * 16 + 4 * g;
                if (row_ok && col < Nout) {
                    float* dst = (float*)p.C + (long)row * p.ldc + col;
                    if (col + 4 <= Nout && ((p.ldc & 3) == 0)) {
                        *(f32x4*)dst = vf[jo];
                    } else {
                        for (int e = 0; e < 4 && col + e < Nout; ++e) dst[e] = vf[jo][e];
                    }
                }
            }
        } else {
            float rs_ss = 0.f, rs_s = 0.f;
#pragma unroll
            for (int jo = 0; jo + 1 < OUT_NT; jo += 2) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[jo][0], pk[jo + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[jo][1], pk[jo + 1][1], false, false);
                u32x4 val = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                const int col = ncol0 + ((g & 1) ? (jo + 1) * 16 + 4 * (g - 1) : jo * 16 + 4 * g);
                if (row_ok && col < Nout) {
                    unsigned short* dst = (unsigned short*)p.C + (long)row * p.ldc + col;
                    const bool vec = (col + 8 <= Nout) && ((p.ldc & 7) == 0);
                    if (p.res) {
                        const unsigned short* rs = p.res + (long)row * p.ldr + col;
                        u32x4 rv;
                        if (resl) {   // staged by this wave (stage_residual): row i*16 + c, chunk (col - ncol0) / 8
                            rv = *(const u32x4*)(resl + (((i * 16 + c) * (OW / 8 + 1) + ((col - ncol0) >> 3)) << 4));
                        } else if (vec && ((p.ldr & 7) == 0) && ((((size_t)p.res) & 15) == 0)) {
                            rv = *(const u32x4*)rs;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned lo = (col + 2 * e < Nout) ? rs[2 * e] : 0u, hi = (col + 2 * e + 1 < Nout) ? rs[2 * e + 1] : 0u;
                                rv[e] = lo | (hi << 16);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)  // the linear output is rounded to bf16 first (bf16 nn.Linear), then the sum is
                            val[e] = pack_bf2(__uint_as_float(val[e] << 16) + __uint_as_float(rv[e] << 16),
                                              __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rv[e] & 0xffff0000u));
                    }
                    if (vec) {
                        *(u32x4*)dst = val;
                    } else {
                        for (int e = 0; e < 8 && col + e < Nout; ++e) dst[e] = (unsigned short)(val[e >> 1] >> (16 * (e & 1)));
                    }
                    if (LNP || p.rs_out) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v1 = (col + e < Nout) ? __uint_as_float((e & 1) ? (val[e >> 1] & 0xffff0000u) : (val[e >> 1] << 16)) : 0.f;
                            rs_ss = __builtin_fmaf(v1, v1, rs_ss);
                            if constexpr (LNP) rs_s += v1;
                        }
                    }
                }
            }
            if constexpr (OUT_NT % 2 == 1) {  // leftover n-tile: 8 bytes per lane
            const int col = ncol0 + (OUT_NT - 1) * 16 + 4 * g;
            if (row_ok && col < Nout) {
                u32x2 val = pk[OUT_NT - 1];
                unsigned short* dst = (unsigned short*)p.C + (long)row * p.ldc + col;
                if (p.res && resl) {   // staged residual: 8 bytes of chunk (col - ncol0) / 8
                    const u32x2 rv = *(const u32x2*)(resl + (((i * 16 + c) * (OW / 8 + 1) + ((col - ncol0) >> 3)) << 4) + (((col - ncol0) & 4) << 1));
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        val[e] = pack_bf2(__uint_as_float(val[e] << 16) + __uint_as_float(rv[e] << 16), __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rv[e] & 0xffff0000u));
                } else if (p.res) {
                    const unsigned short* rs = p.res + (long)row * p.ldr + col;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float lo = (col + 2 * e < Nout) ? bf2f(rs[2 * e]) : 0.f, hi = (col + 2 * e + 1 < Nout) ? bf2f(rs[2 * e + 1]) : 0.f;
                        val[e] = pack_bf2(__uint_as_float(val[e] << 16) + lo, __uint_as_float(val[e] & 0xffff0000u) + hi);
                    }
                }
                if (col + 4 <= Nout && ((p.ldc & 3) == 0)) {
                    *(u32x2*)dst = val;
                } else {
                    for (int e = 0; e < 4 && col + e < Nout; ++e) dst[e] = (unsigned short)(val[e >> 1] >> (16 * (e & 1)));
                }
                if (LNP || p.rs_out) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v1 = (col + e < Nout) ? __uint_as_float((e & 1) ? (val[e >> 1] & 0xffff0000u) : (val[e >> 1] << 16)) : 0.f;
                        rs_ss = __builtin_fmaf(v1, v1, rs_ss);
                        if constexpr (LNP) rs_s += v1;
                    }
                }
            }
            }
            if constexpr (LNP) {
                // the four lanes (g = 0..3) of a row hold different columns: one (sum, sum of squares) per row and wave, parked in the wave's epilogue bytes
                rs_ss += __shfl_xor(rs_ss, 16, 64);
                rs_ss += __shfl_xor(rs_ss, 32, 64);
                rs_s += __shfl_xor(rs_s, 16, 64);
                rs_s += __shfl_xor(rs_s, 32, 64);
                if (g == 0) *(float2*)(rs_lds + 2 * (i * 16 + c)) = make_float2(rs_s, rs_ss);
            } else if (p.rs_out) {
                // the four lanes (g = 0..3) of a row hold different columns: one sum per row and wave, added as a fixed-point integer (order-free)
                rs_ss += __shfl_xor(rs_ss, 16, 64);
                rs_ss += __shfl_xor(rs_ss, 32, 64);
                if (g == 0 && row_ok) {
                    // f32 -> 2^20 fixed point as (hi, lo) words: the value has 24 significant bits, so both steps are exact (the generic f32 -> u64
                    // conversion is a long emulation sequence)
                    const float xs = rs_ss * kRowSumFix + 0.5f;
                    const unsigned hi = (unsigned)(xs * 2.3283064365386963e-10f);
                    const unsigned lo = (unsigned)__builtin_fmaf(-(float)hi, 4294967296.0f, xs);
                    atomicAdd(p.rs_out + row, ((unsigned long long)hi << 32) | lo);
                }
            }
        }
    }
    if constexpr (LNP) {
        // every wave of the workgroup has parked its rows' partial sums (wave id = wm * nwn + wn: the waves of one row block sit 1024 bytes apart); the wn = 0 wave
        // of each row block adds them in wave order and writes ONE pair per row and tile column.  (Single-pass kernels: every wave runs this epilogue exactly once.)
        __syncthreads();
        if (wn == 0) {
            const int tcol = n0 / (nwn * WTN);
#pragma unroll
            for (int q = 0; q < (MT * 16 + 63) / 64; ++q) {
                const int rl = lane + 64 * q;
                const int row = m0 + wm * WTM + rl;
                if (rl < MT * 16 && row < p.M) {
                    float s1 = 0.f, s2 = 0.f;
                    for (int w = 0; w < nwn; ++w) {
                        const float2 v = *(const float2*)(est + w * 1024 + 8 * rl);
                        s1 += v.x;
                        s2 += v.y;
                    }
                    *(float2*)(p.ln_parts + 2 * ((long)row * p.ntn + tcol)) = make_float2(s1, s2);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int ACT, bool OUT_F32, int PIPE, int LNF = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int BK = 64;
    constexpr int ROWB = BK * 2;  // 128 bytes per LDS row
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MT = WTM / 16, NTL = WTN / 16;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int APW = BM / 8 / NW;  // 1-KiB pieces of the A tile per wave
    constexpr int BPW = BN / 8 / NW;
    static_assert(APW >= 1 && BPW >= 1, "tile too small for wave count");
    static_assert(ACT != ACT_SWIGLU || (NTL % 2 == 0), "swiglu needs gate/up tile pairs");

    // GELU / SwiGLU epilogues read their activation table from LDS: it is copied into a DEAD stage while the last K-tile is multiplied (all stages are live
    // before that), and the epilogue's per-wave staging bytes then live in the stage consumed last, so the two never overlap
    constexpr bool TAB = act_uses_table<ACT>();
    static_assert(!TAB || (NW * epi_wave_bytes<NTL, ACT, OUT_F32>() <= STAGE && kActTabBytes <= STAGE), "activation table / epilogue staging do not fit a stage");
    int tab_stage = 0, est_stage = 0;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    // ---- logical tile for this workgroup: XCD-contiguous chunks, grouped (GROUP_M rows of tiles) order
    const unsigned nwg = (unsigned)(p.ntm * p.ntn);
    const unsigned t = xcd_remap(blockIdx.x, nwg);
    const unsigned GROUP_M = (unsigned)p.group_m;
    const unsigned per_group = GROUP_M * p.ntn;
    const unsigned group = t / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = min((unsigned)p.ntm - first_m, GROUP_M);
    const unsigned tm = first_m + (t % per_group) % gsz;
    const unsigned tn = (t % per_group) / gsz;
    const int m0 = tm * BM;
    int n0 = tn * BN;
    if (p.Cn && n0 >= p.N) {   // N-side concatenation: this tile column belongs to the second weight / output pair (workgroup-uniform)
        n0 -= p.N;
        p.W = p.Wn; p.ldw = p.ldwn; p.C = p.Cn; p.ldc = p.ldcn; p.N = p.N2;
        p.bias = nullptr; p.res = nullptr; p.colscale = nullptr; p.A2 = nullptr; p.rs_in = nullptr; p.rs_out = nullptr;
    }

    // ---- per-lane global source offsets (32-bit, in elements) for this wave's staging pieces (source-side swizzle)
    unsigned aoff[APW], boff[BPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        int r = (wid + i * NW) * 8 + (lane >> 3);
        int gr = min(m0 + r, p.M - 1);
        int ch = (lane & 7) ^ ((r >> 1) & 7);
        aoff[i] = (unsigned)((long)gr * p.lda + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        int r = (wid + i * NW) * 8 + (lane >> 3);
        int gr = min(n0 + r, p.N - 1);
        int ch = (lane & 7) ^ ((r >> 1) & 7);
        boff[i] = (unsigned)((long)gr * p.ldw + ch * 8);
    }

    // K-side concatenation: offsets into the second operand pair (K and K2 multiples of 64 there: no partial tile)
    unsigned aoff2[APW], boff2[BPW];
    const int nk1 = (p.K + BK - 1) / BK;
    if (p.A2) {
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int r = (wid + i * NW) * 8 + (lane >> 3);
            aoff2[i] = (unsigned)((long)min(m0 + r, p.M - 1) * p.lda2 + ((lane & 7) ^ ((r >> 1) & 7)) * 8);
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int r = (wid + i * NW) * 8 + (lane >> 3);
            boff2[i] = (unsigned)((long)min(n0 + r, p.N - 1) * p.ldw2 + ((lane & 7) ^ ((r >> 1) & 7)) * 8);
        }
    }
    const bool ktail = (p.K % BK) != 0;  // K is a multiple of 8: the last tile may be partial
    auto stage_tile = [&](int s, int kt, bool last) {
        char* sa = smem + s * STAGE;
        char* sb = sa + BM * ROWB;
        if (p.A2 && kt >= nk1) {   // (workgroup-uniform) K-tiles of the second operand pair
            const long k2 = (long)(kt - nk1) * BK;
#pragma unroll
            for (int i = 0; i < APW; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void*)(p.A2 + aoff2[i] + k2), (lds_void*)(sa + (wid + i * NW) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < BPW; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void*)(p.W2 + boff2[i] + k2), (lds_void*)(sb + (wid + i * NW) * 1024), 16, 0, 0);
            return;
        }
        const long koff = (long)kt * BK;
        if (last && ktail) {
            // chunks at or beyond K read 16 zero bytes instead (per-lane source address; LDS image unchanged)
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int r = (wid + i * NW) * 8 + (lane >> 3);
                const int ch = (lane & 7) ^ ((r >> 1) & 7);
                const unsigned short* src = (kt * BK + ch * 8 < p.K) ? p.A + aoff[i] + koff : (const unsigned short*)&g_zero16;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sa + (wid + i * NW) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const int r = (wid + i * NW) * 8 + (lane >> 3);
                const int ch = (lane & 7) ^ ((r >> 1) & 7);
                const unsigned short* src = (kt * BK + ch * 8 < p.K) ? p.W + boff[i] + koff : (const unsigned short*)&g_zero16;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sb + (wid + i * NW) * 1024), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(p.A + aoff[i] + koff), (lds_void*)(sa + (wid + i * NW) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(p.W + boff[i] + koff), (lds_void*)(sb + (wid + i * NW) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets (read-side swizzle; identical involution as the source side)
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = (lane & 15) * ROWB + (((kk * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4);

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = (p.K + BK - 1) / BK + (p.A2 ? p.K2 / BK : 0);
    static_assert(PIPE == 0 || PIPE == 3, "PIPE: 0 = two LDS stages (one K-tile of prefetch), 3 = three stages (two K-tiles in flight)");
    // K split (tile 14, skinny products: M x 128 over K = 3584 is 66 tiles of 64 x 64): slice blockIdx.y takes ksl K-tiles and writes an f32 slab
    const int kt0 = p.ksl ? (int)blockIdx.y * p.ksl : 0;
    const int nk = p.ksl ? min(nk_all, kt0 + p.ksl) : nk_all;
    if (p.ksl) p.C = (void*)((float*)p.C + (long)blockIdx.y * p.M * p.ldc);
    if constexpr (PIPE == 3) {
        // Three LDS stages, TWO K-tiles of LDS-DMA in flight (cdna_hip_programming.md 5, "Pipelining across barriers": counted vmcnt, never 0 in the steady
        // state, raw s_barrier).  With one tile of prefetch the loop waits out a whole memory round trip whenever a tile's load takes longer than one tile of
        // MFMAs -- every K-tile when the weights stream from HBM (cold, as inside the model: tools/gemm_cold.py measured -10..20 % against warm operands).
        // Order per K-tile: own loads of tile kt landed (all but the youngest tile's LPT loads) -> barrier (everyone's landed, and everyone has finished
        // reading stage (kt - 1) % 3) -> refill that stage with tile kt + 2 -> consume stage kt % 3.
        constexpr int LPT = APW + BPW;   // LDS-DMA instructions per lane and K-tile
        stage_tile(0, kt0, kt0 + 1 == nk_all);
        if (kt0 + 1 < nk) stage_tile(1, kt0 + 1, kt0 + 2 == nk_all);
        int s0 = 0;
        for (int kt = kt0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int s2 = s0 >= 1 ? s0 - 1 : 2;   // (s0 + 2) % 3
            if (kt + 2 < nk) stage_tile(s2, kt + 2, kt + 3 == nk_all);
            if constexpr (TAB) {
                if (kt + 1 == nk) {
                    stage_act_table<NW>(smem + s2 * STAGE, ACT == ACT_SWIGLU, wid, lane);
                    tab_stage = s2;
                    est_stage = s0;
                }
            }
            const char* As = smem + s0 * STAGE + (wm * WTM) * ROWB;
            const char* Bs = smem + s0 * STAGE + BM * ROWB + (wn * WTN) * ROWB;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[MT], wf[NTL];
#pragma unroll
                for (int j = 0; j < NTL; ++j) wf[j] = *(const bf16x8*)(Bs + j * 16 * ROWB + foff[kk]);
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(As + i * 16 * ROWB + foff[kk]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            }
            s0 = s0 == 2 ? 0 : s0 + 1;
        }
    } else {
    stage_tile(kt0 & 1, kt0, kt0 + 1 == nk_all);
    for (int kt = kt0; kt < nk; ++kt) {
        // tile kt has landed (own loads: vmcnt(0); everyone's: barrier) and everyone is done reading
        // the other stage, so it can be refilled while this one is consumed.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_tile((kt + 1) & 1, kt + 1, kt + 2 == nk_all);
        if constexpr (TAB) {
            if (kt + 1 == nk) {
                stage_act_table<NW>(smem + ((kt + 1) & 1) * STAGE, ACT == ACT_SWIGLU, wid, lane);
                tab_stage = (kt + 1) & 1;
                est_stage = kt & 1;
            }
        }
        const char* As = smem + (kt & 1) * STAGE + (wm * WTM) * ROWB;
        const char* Bs = smem + (kt & 1) * STAGE + BM * ROWB + (wn * WTN) * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[MT], wf[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) wf[j] = *(const bf16x8*)(Bs + j * 16 * ROWB + foff[kk]);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(As + i * 16 * ROWB + foff[kk]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    }
    if constexpr (TAB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of the table have landed ...
    __syncthreads();  // ... everyone's; all waves done with the last stage: LDS is free for the epilogue staging
    // residual rows of the wave's block: one LDS-DMA burst behind the per-wave parking bytes (epilogues without an activation table; the block must fit the tile's LDS)
    constexpr int OW = (ACT == ACT_SWIGLU ? NTL / 2 : NTL) * 16;
    constexpr int RSB = res_stage_bytes<MT, OW>();
    constexpr bool RES_LDS = !TAB && !OUT_F32 && (NW * 1024 + NW * RSB <= (PIPE == 3 ? 3 : 2) * STAGE);
    const char* resl = nullptr;
    if constexpr (RES_LDS) {
        const int col0 = (ACT == ACT_SWIGLU) ? (n0 / 2 + wn * OW) : (n0 + wn * OW);
        if (residual_stageable<OW, OUT_F32>(p, col0, (ACT == ACT_SWIGLU) ? p.N / 2 : p.N)) {
            char* dst = smem + NW * 1024 + wid * RSB;
            stage_residual<MT, OW>(p, dst, m0 + wm * WTM, col0, lane);
            resl = dst;
        }
    }
    gemm_epilogue<MT, NTL, WTM, WTN, ACT, OUT_F32, LNF>(acc, p, smem + est_stage * STAGE + wid * epi_wave_bytes<NTL, ACT, OUT_F32>(), smem + tab_stage * STAGE, lane, m0,
                                                        n0, wm, wn, nullptr, nullptr, resl, 0.f, WN);
}

// =====================================================================================================================
// 256x256 "ping-pong" main loop (tile id 20): 8 waves = two groups of four (one wave of each group per SIMD) that run
// the same phase sequence ONE BARRIER APART, so while one group issues its MFMA cluster the other reads fragments and
// issues the next LDS-DMA.  Structure after cdna_hip_programming.md "256^2 8-phase template" (counted vmcnt, never 0 in
// the steady state; raw s_barrier; s_setprio around the MFMA clusters), laid out for this kernel's operand order:
//
//  * a K-tile (64 deep) is four 16-KiB half-tiles  A0 A1 B0 B1  (128 LDS rows x 128 B).  A_h holds, for each of the two
//    wave rows wr, tile rows wr*128 + h*64 + [0,64); B_h holds for each wave column wc tile columns wc*64 + h*32 + [0,32):
//    every wave still owns one CONTIGUOUS 128x64 block of C (the epilogue above is unchanged), and quadrant (ha, hb) of
//    that block needs exactly half-tiles A_ha and B_hb.
//  * per K-tile four phases, one quadrant (16 MFMAs) each:  Q00 [reads A0,B0]  Q01 [reads B1]  Q11 [reads A1]  Q10 [-].
//  * half-tiles are staged in the order S = A0 B0 B1 A1 of tile 0, tile 1, ...; phase g (counted from 0) issues S[g+6]
//    into a region whose last read was >= 2 phases earlier, then waits until all but the 4 youngest half-tiles (8 loads
//    per lane) have landed => S[<= g+2] are complete, and they are first read in phase g+1 (two barriers later).
// LDS: 2 buffers x 4 half-tiles x 16 KiB = 128 KiB, one workgroup per CU, 2 waves per SIMD.
// N192 (tile 23): the same loop on a 4 x 2 wave grid -- wave blocks of 64 x 96, tile 256 x 192 -- for widths that are multiples of 192 but not of 256 (Hiera stage 3:
// N = 576 is 3 x 192 against 2.25 x 256: a quarter of the 256-wide tiles' MFMAs multiply padding).  A half-tile: 4 wave rows x 32 rows; B half-tile: 2 wave columns x 48
// columns = 96 LDS rows of its 128-row region (waves 4-7 repeat the second staging piece of waves 0-3, as the 192-ROW form of the persistent kernel does, so that every
// wave still issues two loads per half-tile); quadrant = 2 x 3 x 2 = 12 MFMAs.
template <int ACT, bool OUT_F32, int LNF = 0, bool N192 = false>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = N192 ? 192 : 256, BK = 64, ROWB = 128;
    constexpr int HALF = 128 * ROWB;  // 16 KiB
    constexpr int BUF = 4 * HALF;     // A0 | A1 | B0 | B1
    constexpr int WC = N192 ? 2 : 4;                 // wave columns (wave rows = 8 / WC)
    constexpr int MI = N192 ? 2 : 4, NI = N192 ? 3 : 2;   // 16-row m-tiles / 16-column n-tiles of a wave per half-tile
    constexpr int AH = 16 * MI, BH = 16 * NI;        // rows / columns of a wave inside one half-tile
    constexpr int MT = 2 * MI, NTL = 2 * NI;
    constexpr int WTM = 2 * AH, WTN = 2 * BH;        // wave block

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WC, wc = wid % WC;  // waves w and w+4 share a SIMD: one of each group (wid >> 2) per SIMD
    const int grp = wid >> 2;

    // activation table (GELU / SwiGLU epilogues): 10 KiB behind the two buffers, copied first (the oldest loads of every wave: every counted wait below covers them)
    if constexpr (act_uses_table<ACT>()) stage_act_table<8>(smem + 2 * BUF, ACT == ACT_SWIGLU, wid, lane);

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

    // ---- staging: every half-tile is 2 LDS-DMA pieces (8 rows x 128 B) per wave; source-side XOR swizzle as above.
    //      kind 0..3 = A0 A1 B0 B1 (LDS order); soff[kind][piece] = element offset of this lane's 16-byte chunk at k = 0
    const int sch = (lane & 7) ^ ((((wid & 1) << 2) + (lane >> 4)) & 7);  // source chunk (same for both pieces)
    unsigned soff[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid + 8 * i) * 8 + (lane >> 3);  // LDS row of the half-tile, 0..127
        // 96-row B half-tiles (N192): piece 12 + w of waves 4-7 does not exist; they repeat piece 8 + (w & 3) (same destination: see stage)
        const int rb = (!N192 || i == 0 || wid < 4) ? r : ((8 + (wid & 3)) * 8 + (lane >> 3));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (r / AH) * (2 * AH) + h * AH + (r % AH);
            const int bcol = (rb / BH) * (2 * BH) + h * BH + (rb % BH);
            soff[h][i] = (unsigned)((long)min(m0 + arow, p.M - 1) * p.lda + sch * 8);
            soff[2 + h][i] = (unsigned)((long)min(n0 + bcol, p.N - 1) * p.ldw + sch * 8);
        }
    }
    const int nk = (p.K + BK - 1) / BK;
    const bool ktail = (p.K % BK) != 0;
    auto stage = [&](auto KIND, int kt, int buf) {
        constexpr int kind = decltype(KIND)::value;
        const unsigned short* base = ((kind < 2) ? p.A : p.W) + (long)kt * BK;  // scalar part first: saddr + 32-bit voffset form
        char* dst = smem + buf * BUF + kind * HALF + wid * 1024;
        const int second = (N192 && kind >= 2 && wid >= 4) ? (8 + (wid & 3) - wid) * 1024 : 8192;   // second piece: 8 KiB on (N192 B half-tiles: see soff)
        if (ktail && kt == nk - 1) {
            const bool ok = kt * BK + sch * 8 < p.K;  // chunks at or beyond K come from 16 zero bytes
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned short* src = ok ? base + soff[kind][i] : (const unsigned short*)&g_zero16;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(dst + i * second), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + soff[kind][i]), (lds_void*)(dst + i * second), 16, 0, 0);
        }
    };
    using K_A0 = std::integral_constant<int, 0>;
    using K_A1 = std::integral_constant<int, 1>;
    using K_B0 = std::integral_constant<int, 2>;
    using K_B1 = std::integral_constant<int, 3>;
    // wait until at most n half-tiles (2 loads each) issued by this lane are still in flight
    auto wait_halftiles = [&](int n) {
        if (n >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = (lane & 15) * ROWB + (((kk * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4);
    const int a_rd = (wr * AH) * ROWB;                 // + h*HALF + mi*2048 + foff[kk]
    const int b_rd = 2 * HALF + (wc * BH) * ROWB;      // + h*HALF + ni*2048 + foff[kk]

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 af[MI][2], b0[NI][2], b1[NI][2];
    auto read_a = [&](const char* cur, int h) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *(const bf16x8*)(cur + a_rd + h * HALF + mi * 2048 + foff[kk]);
    };
    auto read_b = [&](bf16x8 (&bf)[NI][2], const char* cur, int h) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bf[ni][kk] = *(const bf16x8*)(cur + b_rd + h * HALF + ni * 2048 + foff[kk]);
    };
    auto mma_quadrant = [&](auto HA, auto HB, const bf16x8 (&bf)[NI][2]) {
        constexpr int ha = decltype(HA)::value, hb = decltype(HB)::value;
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[ha * MI + mi][hb * NI + ni] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni][kk], af[mi][kk], acc[ha * MI + mi][hb * NI + ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    // ---- prologue: S[0..5] = tile 0 complete + A0, B0 of tile 1
    stage(K_A0{}, 0, 0);
    stage(K_B0{}, 0, 0);
    stage(K_B1{}, 0, 0);
    stage(K_A1{}, 0, 0);
    if (nk > 1) {
        stage(K_A0{}, 1, 1);
        stage(K_B0{}, 1, 1);
    }
    wait_halftiles(nk > 1 ? 4 : 2);  // S[0], S[1] landed (own loads) ...
    __builtin_amdgcn_s_barrier();    // ... and everyone's
    if (grp == 1) __builtin_amdgcn_s_barrier();  // second group runs one barrier behind the first

    const int last = 4 * nk - 1;  // index of the last half-tile
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* cur = smem + buf * BUF;
        const int g = 4 * kt;
        const bool steady = kt + 2 < nk;
        // phase 1: Q00
        read_a(cur, 0);
        read_b(b0, cur, 0);
        if (kt + 1 < nk) stage(K_B1{}, kt + 1, buf ^ 1);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 6, last) - (g + 2));
        mma_quadrant(H0{}, H0{}, b0);
        // phase 2: Q01
        read_b(b1, cur, 1);
        if (kt + 1 < nk) stage(K_A1{}, kt + 1, buf ^ 1);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 7, last) - (g + 3));
        mma_quadrant(H0{}, H1{}, b1);
        // phase 3: Q11
        read_a(cur, 1);
        if (kt + 2 < nk) stage(K_A0{}, kt + 2, buf);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 8, last) - (g + 4));
        mma_quadrant(H1{}, H1{}, b1);
        // phase 4: Q10
        if (kt + 2 < nk) stage(K_B0{}, kt + 2, buf);
        if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(max(min(g + 9, last) - (g + 5), 0));
        mma_quadrant(H1{}, H0{}, b0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align the groups
    __syncthreads();  // LDS is free for the epilogue staging
    constexpr int OW = (ACT == ACT_SWIGLU ? NTL / 2 : NTL) * 16;
    constexpr int RSB = res_stage_bytes<MT, OW>();
    const char* resl = nullptr;
    if constexpr (pp_res_lds<ACT, OUT_F32>()) {   // the launcher grants 8 KiB + 8 x RSB of LDS for these instantiations
        const int col0 = (ACT == ACT_SWIGLU) ? (n0 / 2 + wc * OW) : (n0 + wc * OW);
        if (residual_stageable<OW, OUT_F32>(p, col0, (ACT == ACT_SWIGLU) ? p.N / 2 : p.N)) {
            char* dst = smem + 8 * 1024 + wid * RSB;
            stage_residual<MT, OW>(p, dst, m0 + wr * WTM, col0, lane);
            resl = dst;
        }
    }
    gemm_epilogue<MT, NTL, WTM, WTN, ACT, OUT_F32, LNF>(acc, p, smem + wid * epi_wave_bytes<NTL, ACT, OUT_F32>(), smem + 2 * BUF, lane, m0, n0, wr, wc, nullptr, nullptr, resl, 0.f, WC);
}

// In-place accumulate (C-in register == C-out register).  The builtin lets the register allocator pick a different
// destination, and inside the persistent item loop it does - 32 extra live registers and spills in the MFMA phases.
// Hazards: operands come from ds_read (waited by lgkmcnt) and from MFMAs >= 7 instructions earlier; the first VALU
// read of an accumulator after the loop is behind several barriers.
__device__ __forceinline__ void mfma_inplace(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// c = a . b (C operand = the inline constant 0): the first k-step of an accumulation defines the accumulator
__device__ __forceinline__ void mfma_zero(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}

// =====================================================================================================================
// Persistent ping-pong GEMM with a stream-K tail (tile ids 21 / 22).  One workgroup per CU walks a list of work items:
//   [its non-owner slice of a split tile]  [its data-parallel tiles  w, w+P, w+2P, ...]  [its owner slice of a split tile]
// * The first K-tile of the NEXT item is prefetched (LDS-DMA) before the epilogue of the current one, so the store tail
//   and the load prologue overlap (short-K shapes: ViT K = 1280 is 20 K-tiles per tile).
// * Stream-K tail (id 22): the T mod P tiles of the last, partly filled round are cut into equal runs of K-iterations over
//   P_sk workgroups instead of leaving P - (T mod P) CUs idle.  A run touches at most two tiles: the END of tile a (this
//   workgroup then owns a: it adds the partial sums of the lower-numbered workgroups that computed a's earlier
//   K-iterations and runs the epilogue) and the START of tile a+1 (partial sums go to this workgroup's f32 slab).
//   Non-owner slices never wait and are computed first, owner slices last, so an owner only ever waits for work that needs
//   nothing from anybody (no cycles; every spin is bounded all the same and reports into sk.tmo).
//   Hand-off = cdna_hip_programming.md Guideline 16, write-through form (R1): sc1 slab stores, every wave vmcnt(0), barrier, one lane's
//   relaxed agent flag store (no release fence); owner: one lane polls relaxed, agent-scope acquire fence,
//   wait, barrier, then plain vector loads.  Flags are zero at allocation and reset by their single consumer.
//   The split is a pure function of (M, N, K, P): results are reproducible run to run.
// dynamic LDS of the persistent kernel: two K-tile buffers (+ the activation table), or the residual staging image when that is larger; one more 16-byte slot
// behind it carries the owner's give-up flag from lane 0 to the workgroup
template <int ACT, bool OUT_F32, int MH>
constexpr int sk_lds_bytes() {
    constexpr int LDS_MAIN = 2 * 4 * 128 * 128 + (act_uses_table<ACT>() ? kActTabBytes : 0);
    constexpr int LDS_RES = pp_res_lds<ACT, OUT_F32>() ? 8 * 1024 + 8 * res_stage_bytes<2 * MH, 64>() : 0;
    return LDS_MAIN > LDS_RES ? LDS_MAIN : LDS_RES;
}

constexpr int kSkMaxP = 320;   // workgroups (= CUs) a run table has room for
struct SkArgs {
    float* slabs;     // [P][512 lanes][32] f32x4 in register order (256 KiB per workgroup)
    unsigned* flags;  // [P] 1 = slab written
    unsigned* tmo;    // tmo[0]: bounded-spin give-up counter; tmo[1]: fault injection (tests), 0 in real runs
    int P;            // workgroups launched
    int P_sk;         // workgroups that take part in the stream-K tail (host bookkeeping: the kernel reads the run table)
    int t_dp;         // tiles [0, t_dp) are data-parallel (t_dp % P == 0 or sk_tiles == 0)
    int sk_tiles;     // tiles [t_dp, t_dp + sk_tiles) are split
    int all_partial;  // split-K mode (tile 25): every slice only writes its slab; gemm_slab_reduce_kernel sums them afterwards
    int plain_slabs;  // measurement builds only (RGA3_AB, env RGA3_SK_PLAIN=1): the round-3 hand-off -- plain slab stores + agent-scope release fence
    // ragged = 1 (tiles 26 / 27): the last tile row holds <= 64 rows (M = 2112 = 8 x 256 + 64); its ntn tiles run the quarter-work loop.  Tile numbering then:
    // [0, ntn) the ragged tiles by column, [ntn, T) the full tiles in the grouped order over ntm - 1 tile rows.  rg_wgs = G > 0: G workgroups, evenly spaced in the
    // workgroup numbering (hence over the XCDs), take ONLY ragged tiles -- the one of rank j the columns j, j + G, ... -- and the other P - G share the full tiles
    // [ntn, t_dp) strided by P - G.  G is sized so that both kinds finish together (sk_plan_ragged); a ragged workgroup then walks the tile columns at the pace the
    // full-tile workgroups do, i.e. it reads weight columns that are in flight through the Infinity Cache anyway.  rg_wgs = 0: strided tiles w, w + P, ... < t_dp.
    int ragged, rg_wgs;
    unsigned long long* dbg = nullptr;   // measurement builds only (RGA3_AB, env RGA3_SK_DBG=1): [P][8 items][8] s_memtime stamps of wave 0 (tools/probes/sk_items.py)
    // run table of the stream-K tail: workgroup w takes K-iterations [start[w], start[w + 1]) of the line  tile t_dp (nk iterations), tile t_dp + 1, ...
    // (host-made: equal runs, or runs sized by cost when ragged tiles are cheaper; sk_plan_*).  Empty runs are allowed.
    unsigned start[kSkMaxP + 1];
};

// MH = 16-row m-tiles per A half-tile and wave row: 4 -> 256-row tiles, 3 -> 192-row tiles (M = 2112 = 11 x 192: no padded tile row;
// the A half-tiles then hold 96 rows in their 128-row LDS regions, and waves 4-7 repeat the second staging piece of waves 0-3 so that
// every wave still issues two loads per half-tile and the counted vmcnt schedule is the same).
// RG: the instantiation that also holds the ragged-tile loop (tiles 26 / 27).  It is a separate program: the persistent kernel sits at the register limit, and a second
// loop in the same program moved the allocation of the first (a staging offset spilled INTO the full loop: a scratch reload + vmcnt(0) per K-tile, half the rate).
template <int ACT, bool OUT_F32, int MH = 4, bool RG = false>
__global__ __launch_bounds__(512) void gemm_nt_sk_kernel(GemmArgs p, SkArgs sk) {
    constexpr int BM = 64 * MH, BN = 256, BK = 64, ROWB = 128;
    constexpr int HALF = 128 * ROWB;
    constexpr int BUF = 4 * HALF;
    constexpr int MT = 2 * MH, NTL = 4;
    constexpr int EPW = epi_wave_bytes<NTL, ACT, OUT_F32>();
    // epilogue staging lives in the A1 / B1 regions of buffer 0 (waves 0-3 / 4-7) so that the next item's first six
    // half-tiles (buffer 1 complete, A0 + B0 of buffer 0) can be in flight during the epilogue; the f32 epilogue needs
    // more than 16 KiB per four waves, so that variant prefetches after its epilogue instead.
    constexpr bool PREFETCH = (4 * EPW <= HALF);
#ifndef FIXB_N
#define FIXB_N 8
#endif
    constexpr int FIXB = FIXB_N;  // slab quads in flight per batch while adding partial sums

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int w = (int)xcd_remap(blockIdx.x, (unsigned)sk.P);  // logical workgroup id: consecutive ids share an XCD
    const int nk = (p.K + BK - 1) / BK;
    // K % 64 == 0 here (the launcher sends ragged K to the one-tile-per-workgroup kernel, which has the zero-source tail)
    char* est = PREFETCH ? smem + (wr ? 3 * HALF : HALF) + wc * EPW : smem + wid * EPW;

    // ---- work list.  A run [x0, x1) of the tail line decomposes into: the END of the tile it starts inside (owner slice: this workgroup adds the partial sums of the
    //      lower-numbered workgroups that computed the tile's earlier K-iterations), whole tiles, and the START of the tile it ends inside (non-owner slice).  A run that
    //      neither starts nor ends at a tile boundary of its single tile is a non-owner (middle) slice.
    int na_tile = -1, na_kb = 0, na_ke = 0;  // non-owner slice (first)
    int ow_tile = -1, ow_kb = 0;             // owner slice [ow_kb, nk) (last)
    int tw0 = 0, n_tw = 0;                   // whole tiles inside the run (after the data-parallel ones)
    if (sk.sk_tiles > 0) {
        const unsigned x0 = sk.start[w], x1 = sk.start[w + 1];
        if (x0 < x1) {
            const int ta = (int)(x0 / (unsigned)nk), tz = (int)((x1 - 1) / (unsigned)nk);
            const unsigned sa = (unsigned)ta * nk, ez = (unsigned)(tz + 1) * nk;
            int first_whole = ta, end_whole = tz + 1;
            if (x0 > sa) {   // the run starts inside tile ta
                first_whole = ta + 1;
                if (x1 >= sa + nk) { ow_tile = sk.t_dp + ta; ow_kb = (int)(x0 - sa); }
                else { na_tile = sk.t_dp + ta; na_kb = (int)(x0 - sa); na_ke = (int)(x1 - sa); }
            }
            if (x1 < ez && (tz > ta || x0 == sa)) {   // the run ends inside tile tz, whose first iteration it holds
                end_whole = tz;
                na_tile = sk.t_dp + tz; na_kb = 0; na_ke = (int)(x1 - (unsigned)tz * nk);
            }
            tw0 = sk.t_dp + first_whole;
            n_tw = max(end_whole - first_whole, 0);
        }
    }
    int n_dp, dp0 = w, dp_step = sk.P;   // data-parallel tiles dp0, dp0 + dp_step, ... (n_dp of them)
    if (sk.rg_wgs > 0) {
        const int G = sk.rg_wgs;
        const int below = (w * G) / sk.P;   // ragged workgroups with a lower number
        if (((w + 1) * G) / sk.P > below) {   // this workgroup takes ragged tiles only
            dp0 = below; dp_step = G;
            n_dp = (below < p.ntn) ? (p.ntn - below + G - 1) / G : 0;
        } else {
            const int rk = w - below, Pf = sk.P - G, nf = sk.t_dp - p.ntn;   // its rank among the full-tile workgroups; full tiles of the data-parallel part
            dp0 = p.ntn + rk; dp_step = Pf;
            n_dp = (rk < nf) ? (nf - rk + Pf - 1) / Pf : 0;
        }
    } else {
        n_dp = (w < sk.t_dp) ? (sk.t_dp - w + sk.P - 1) / sk.P : 0;
    }
    const int n_items = (na_tile >= 0) + n_dp + n_tw + (ow_tile >= 0);
    if (n_items == 0) return;
    // item i -> (tile, kb, ke, kind): kind 0 = whole tile, 1 = non-owner slice, 2 = owner slice
    auto item = [&](int i, int& tile, int& kb, int& ke, int& kind) {
        if (na_tile >= 0) {
            if (i == 0) { tile = na_tile; kb = na_kb; ke = na_ke; kind = 1; return; }
            --i;
        }
        if (i < n_dp) { tile = dp0 + i * dp_step; kb = 0; ke = nk; kind = 0; return; }
        i -= n_dp;
        if (i < n_tw) { tile = tw0 + i; kb = 0; ke = nk; kind = 0; return; }
        tile = ow_tile; kb = ow_kb; ke = nk; kind = sk.all_partial ? 1 : 2;
    };

    unsigned soff[4][2];
    unsigned soffA3[3];    // MH == 3: the three 64-row A units of a K-tile (one piece per wave each)
    int nm0 = 0, nn0 = 0;  // tile origin of the item whose offsets are in soff
    // Ragged tiles (MH = 4, sk.ragged): a tile of the last tile row that holds <= 64 rows (M = 2112 = 8 x 256 + 64: one tile row in nine multiplied 75 % padding).
    // Its rows are exactly LDS rows 0..63 of half-tile A0 under the normal staging map, so the item stages only the FIRST piece of A0 and no A1, and wave row wr reads LDS
    // rows wr * 32 + [0, 32) (two m-tiles: 16 MFMAs per K-tile and wave instead of 64); its own three-stage loop is at the item loop.
    constexpr bool RGOK = RG && (MH == 4);
    bool rg_n = false;     // ... and whether that tile is ragged
    auto setup_tile = [&](int tile, int lane) {
        const int sch = (lane & 7) ^ ((((wid & 1) << 2) + (lane >> 4)) & 7);
        const unsigned GROUP_M = (unsigned)p.group_m;
        bool ragged_tile = false;
        if constexpr (RGOK) ragged_tile = sk.ragged && tile < p.ntn;
        if (ragged_tile) {
            nm0 = (p.ntm - 1) * BM;
            nn0 = tile * BN;
        } else {
            const unsigned ntm_f = (unsigned)p.ntm - ((RGOK && sk.ragged) ? 1u : 0u);   // tile rows in the grouped order
            const unsigned t = (unsigned)tile - ((RGOK && sk.ragged) ? (unsigned)p.ntn : 0u);
            const unsigned per_group = GROUP_M * p.ntn;
            const unsigned group = t / per_group;
            const unsigned first_m = group * GROUP_M;
            const unsigned gsz = min(ntm_f - first_m, GROUP_M);
            nm0 = (int)(first_m + (t % per_group) % gsz) * BM;
            nn0 = (int)((t % per_group) / gsz) * BN;
        }
        if constexpr (RGOK) rg_n = ragged_tile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (wid + 8 * i) * 8 + (lane >> 3);
            // A: LDS row ra of a half-tile holds the wave row's (ra / (16 MH)) rows h * 16 MH + ra % (16 MH); with MH = 3 the second piece
            // of waves 4-7 (rows 96..127 do not exist) repeats the one of waves 0-3
            const int ra = (MH == 4) ? r : ((i == 0 ? wid : 8 + (wid & 3)) * 8 + (lane >> 3));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int arow = (ra / (16 * MH)) * (32 * MH) + h * (16 * MH) + (ra % (16 * MH));
                const int bcol = (r >> 5) * 64 + h * 32 + (r & 31);
                soff[h][i] = (unsigned)((long)min(nm0 + arow, p.M - 1) * p.lda + sch * 8);
                soff[2 + h][i] = (unsigned)((long)min(nn0 + bcol, p.N - 1) * p.ldw + sch * 8);
            }
        }
        if constexpr (MH == 3) {
            // A unit u (phase u of a K-tile) = m-tiles 2u, 2u + 1 of both wave rows: LDS row lr = 8 wid + lane / 8 of the unit is tile row (lr / 32) * 96 + 32 u + lr % 32
            const int lr = wid * 8 + (lane >> 3);
#pragma unroll
            for (int u = 0; u < 3; ++u) soffA3[u] = (unsigned)((long)min(nm0 + (lr >> 5) * 96 + 32 * u + (lr & 31), p.M - 1) * p.lda + sch * 8);
        }
    };
    auto stage_a3 = [&](int u, int kt, int buf) {   // MH == 3: one 1-KiB piece per wave
        const unsigned short* base = p.A + (long)kt * BK;
        unsigned o = soffA3[u];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((gbl_void*)(base + o), (lds_void*)(smem + buf * BUF + u * 8192 + wid * 1024), 16, 0, 0);
    };
    auto stage = [&](auto KIND, int kt, int buf) {
        constexpr int kind = decltype(KIND)::value;
        const unsigned short* base = ((kind < 2) ? p.A : p.W) + (long)kt * BK;  // scalar part first: saddr + 32-bit voffset form
        char* dst = smem + buf * BUF + kind * HALF + wid * 1024;
        // second piece: 8 KiB further on; the 96-row A half-tiles (MH = 3) put it at piece 8 + (wid & 3) (waves 4-7 rewrite what waves 0-3 write)
        const int second = (MH == 3 && kind < 2) ? (8 + (wid & 3) - wid) * 1024 : 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned o = soff[kind][i];
            asm volatile("" : "+v"(o));  // keep the offsets 32-bit in registers (the compiler otherwise holds 8 zero-extended pairs)
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + o), (lds_void*)(dst + i * second), 16, 0, 0);
        }
    };
    auto stage_a_rg = [&](int kt, int buf) {   // ragged tile: the 64 live rows of A0, one 1-KiB piece per wave
        const unsigned short* base = p.A + (long)kt * BK;
        unsigned o = soff[0][0];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((gbl_void*)(base + o), (lds_void*)(smem + buf * BUF + wid * 1024), 16, 0, 0);
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
    // the first K-tile of an item always goes to buffer 1, the second to buffer 0, ...
    auto prologue_issue = [&](int kb, int ke) {
        if constexpr (RGOK) {
            if (rg_n) {   // the same order without A1 and with one A0 piece: 5 (3) loads per wave
                stage_a_rg(kb, 1);
                stage(K_B0{}, kb, 1);
                stage(K_B1{}, kb, 1);
                if (ke - kb > 1) {
                    stage_a_rg(kb + 1, 0);
                    stage(K_B0{}, kb + 1, 0);
                }
                return;
            }
        }
        if constexpr (MH == 3) {
            // the steady-state issue order of the three-phase loop: B0 A_0 | B1 A_1 | A_2 of the first K-tile (buffer 1), B0 A_0 of the second (buffer 0) -- the counted
            // waits of the loop assume exactly this sequence; B1 / A_2 of buffer 0 (where the epilogue parks its row statistics) follow inside the loop
            stage(K_B0{}, kb, 1);
            stage_a3(0, kb, 1);
            stage(K_B1{}, kb, 1);
            stage_a3(1, kb, 1);
            stage_a3(2, kb, 1);
            if (ke - kb > 1) {
                stage(K_B0{}, kb + 1, 0);
                stage_a3(0, kb + 1, 0);
            }
            return;
        }
        stage(K_A0{}, kb, 1);
        stage(K_B0{}, kb, 1);
        stage(K_B1{}, kb, 1);
        stage(K_A1{}, kb, 1);
        if (ke - kb > 1) {
            stage(K_A0{}, kb + 1, 0);
            stage(K_B0{}, kb + 1, 0);
        }
    };

    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = (lane & 15) * ROWB + (((kk * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4);
    const int a_rd = (wr * 16 * MH) * ROWB;
    const int b_rd = 2 * HALF + (wc * 32) * ROWB;

    f32x4 acc[MT][NTL];
    bf16x8 af[MH][2], b0[2][2], b1[2][2];
    auto read_a = [&](const char* cur, int h) {
#pragma unroll
        for (int mi = 0; mi < MH; ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *(const bf16x8*)(cur + a_rd + h * HALF + mi * 2048 + foff[kk]);
    };
    auto read_b = [&](bf16x8 (&bf)[2][2], const char* cur, int h) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bf[ni][kk] = *(const bf16x8*)(cur + b_rd + h * HALF + ni * 2048 + foff[kk]);
    };
    // (Skipping the MFMA clusters of a ragged tile row's dead 64-row halves was tried: at M = 2112 the uniform branches around the
    //  clusters cost more in every tile than the skipped work saves in one tile row of nine.)
    auto mma_quadrant = [&](auto HA, auto HB, const bf16x8 (&bf)[2][2]) {
        constexpr int ha = decltype(HA)::value, hb = decltype(HB)::value;
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < MH; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    mfma_inplace(acc[ha * MH + mi][hb * 2 + ni], bf[ni][kk], af[mi][kk]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    // ---- MH == 3 (192-row tiles): THREE phases of 16 MFMAs per K-tile instead of four of 12.  The fixed cost of a phase (two barriers, the fragment reads, the DMA
    //      issue: ~170 cycles against 512 of matrix work for the two waves of a SIMD) is what made the four-phase form of these tiles lose the 8 % of padding they save
    //      at M = 2112 = 11 x 192 (520 vs 523 us on the gate | up product).  Phase u multiplies m-tiles 2u, 2u + 1 of the wave's 96 rows by ALL four n-tiles: the B
    //      fragments of a K-tile are read once (phase 0) and kept in registers (32), A unit u (64 rows: 2 m-tiles x 2 wave rows) is read at phase u.
    auto read_a3 = [&](const char* cur, int u) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *(const bf16x8*)(cur + u * 8192 + wr * 4096 + mi * 2048 + foff[kk]);
    };
    auto mma_phase3 = [&](auto U) {
        constexpr int u = decltype(U)::value;
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) mfma_inplace(acc[(MH == 3 ? 2 * u : 0) + mi][ni], b0[ni][kk], af[mi][kk]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) mfma_inplace(acc[(MH == 3 ? 2 * u : 0) + mi][2 + ni], b1[ni][kk], af[mi][kk]);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;

    // activation table (GELU / SwiGLU epilogues): 10 KiB behind the two buffers, once per workgroup = once per CU and launch
    if constexpr (act_uses_table<ACT>()) stage_act_table<8>(smem + 2 * BUF, ACT == ACT_SWIGLU, wid, lane);

    int tile, kb, ke, kind;
    item(0, tile, kb, ke, kind);
    setup_tile(tile, lane);
    prologue_issue(kb, ke);

    for (int it = 0; it < n_items; ++it) {
        const int m0 = nm0, n0 = nn0;
        const bool rg = rg_n;   // this item's tile is ragged (wave-uniform; false unless MH = 4 and sk.ragged)
        // ---- all six (or four) prologue half-tiles have been issued; older stores of the previous epilogue count in
        //      vmcnt too, so simply drain: the loads have been in flight for a whole epilogue
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef RGA3_AB
        if (sk.dbg && tid == 0 && it < 8) sk.dbg[(size_t)w * 64 + it * 8 + 0] = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef RGA3_AB
        if (sk.dbg && tid == 0 && it < 8) sk.dbg[(size_t)w * 64 + it * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
        if (wr == 1) __builtin_amdgcn_s_barrier();
        const int nkt = ke - kb;
        const int last = 4 * nkt - 1;
        int q = 0;
        if constexpr (MH == 3) {
            // Issue order (pieces per wave: B halves 2, A units 1):  ... | q.ph0: B1(q+1) A_1(q+1) | q.ph1: A_2(q+1) | q.ph2: B0(q+2) A_0(q+2) | ...
            // Every unit is issued >= 2 phases after the last read of its region and >= 3 phases before its first read; a counted wait sits before the FIRST barrier
            // of a phase and retires what the NEXT phase reads (one barrier more than the lockstep rule: the wave rows run a barrier apart).
            //   end of ph0 needs A_1(q):  issued after it  A_2(q) 1 + [B0 A_0 B1 A_1](q+1) 6   -> vmcnt(7), vmcnt(1) on the last K-tile
            //   end of ph1 needs A_2(q):  issued after it  [B0 A_0 B1 A_1 A_2](q+1) 7          -> vmcnt(7), vmcnt(0) on the last K-tile
            //   end of ph2 needs B0 A_0 B1 (q+1): after them  A_1(q+1) A_2(q+1) 2 + [B0 A_0](q+2) 3 -> vmcnt(5), vmcnt(2) when q + 2 does not exist
            do {
                const int kt = kb + q;
                const int buf = (q + 1) & 1;
                const char* cur = smem + buf * BUF;
                const bool e1 = q + 1 < nkt, e2 = q + 2 < nkt;
                read_a3(cur, 0);
                read_b(b0, cur, 0);
                read_b(b1, cur, 1);
                if (e1) { stage(K_B1{}, kt + 1, buf ^ 1); stage_a3(1, kt + 1, buf ^ 1); }
                if (e1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                mma_phase3(U0{});
                read_a3(cur, 1);
                if (e1) stage_a3(2, kt + 1, buf ^ 1);
                if (e1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                mma_phase3(U1{});
                read_a3(cur, 2);
                if (e2) { stage(K_B0{}, kt + 2, buf); stage_a3(0, kt + 2, buf); }
                if (e2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else if (e1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                mma_phase3(U2{});
            } while (++q < nkt);
        } else if (RGOK && rg) {
            // ---- ragged tile: THREE stages of 40 KiB (A' = the 64 live rows, one piece per wave; B0, B1 = two pieces each) in the two buffers' 128 KiB, K-tile q in stage q % 3:
            //        stage 0 = buffer 1's A0 (lower half) / B0 / B1,   stage 1 = buffer 0's,   stage 2 = upper half of buffer 1's A0 / buffer 1's A1 / buffer 0's A1.
            //      The four-phase skeleton of the full loop on a quarter of the MFMAs ran a K-tile in 0.78 of the full loop's time: every load had ONE K-tile to land and the
            //      loop sat at the memory latency.  Here a K-tile is issued two K-tiles ahead, and a K-tile is ONE cluster of 16 MFMAs between two barriers:
            //        barrier | read the 12 fragments of q | issue K-tile q + 2 (5 loads) | 16 MFMAs | vmcnt: K-tile q + 1 has landed | barrier
            //      With the wave rows a barrier apart (as in the full loop): a row's first barrier pairs with the other row's second barrier of the K-tile before, so the
            //      counted wait in front of the SECOND barrier is what makes q + 1 visible to the other row, and a stage is restaged only after both rows have read it
            //      (stage (q + 2) % 3 was read during q - 1; the issuing row has passed a barrier that the other row reaches after its reads of q - 1).
            //      vmcnt: issued after K-tile q + 1: the 5 loads of q + 2 -> vmcnt(5); none left to issue -> vmcnt(0).  The prologue brought K-tile kb complete and A', B0 of
            //      kb + 1 (B1 of stage 1 and all of stage 2 overlap the previous item's epilogue staging): q = 0 issues B1(kb + 1) first.
            auto rg_off = [&](int st, int which) {   // byte offset of stage st's A' (0) / B0 (1) / B1 (2) region
                return which == 0 ? (st == 0 ? BUF : st == 1 ? 0 : BUF + 8192)
                     : which == 1 ? (st == 0 ? BUF + 2 * HALF : st == 1 ? 2 * HALF : BUF + HALF)
                                  : (st == 0 ? BUF + 3 * HALF : st == 1 ? 3 * HALF : HALF);
            };
            auto stage_rg_b = [&](int h, int kt, int off) {
                const unsigned short* base = p.W + (long)kt * BK;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned o = soff[2 + h][i];
                    asm volatile("" : "+v"(o));
                    __builtin_amdgcn_global_load_lds((gbl_void*)(base + o), (lds_void*)(smem + off + wid * 1024 + i * 8192), 16, 0, 0);
                }
            };
            auto stage_rg_a = [&](int kt, int off) {
                const unsigned short* base = p.A + (long)kt * BK;
                unsigned o = soff[0][0];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + o), (lds_void*)(smem + off + wid * 1024), 16, 0, 0);
            };
            int st = 0;
            do {
                const int kt = kb + q;
                const int oa = rg_off(st, 0) + wr * (32 * ROWB), ob0 = rg_off(st, 1) + (wc * 32) * ROWB, ob1 = rg_off(st, 2) + (wc * 32) * ROWB;
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *(const bf16x8*)(smem + oa + mi * 2048 + foff[kk]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        b0[ni][kk] = *(const bf16x8*)(smem + ob0 + ni * 2048 + foff[kk]);
                        b1[ni][kk] = *(const bf16x8*)(smem + ob1 + ni * 2048 + foff[kk]);
                    }
                if (q == 0 && nkt > 1) stage_rg_b(1, kt + 1, rg_off(1, 2));
                const bool e2 = q + 2 < nkt;
                if (e2) {
                    const int s2 = st == 0 ? 2 : st - 1;   // (st + 2) % 3
                    stage_rg_a(kt + 2, rg_off(s2, 0));
                    stage_rg_b(0, kt + 2, rg_off(s2, 1));
                    stage_rg_b(1, kt + 2, rg_off(s2, 2));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_setprio(1);
                // The first k-step of an item takes the constant 0 as its C operand: the accumulators are DEFINED by those MFMAs.  With "acc = 0; acc += ..." the compiler
                // peeled the first K-tile and materialised each zero right before its first MFMA -- in a register that an MFMA issued just before still read as an operand
                // (no hazard is visible behind an asm MFMA): wrong sums in the ragged rows.
                if (q == 0) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) mfma_zero(acc[mi][ni], b0[ni][0], af[mi][0]);
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) mfma_zero(acc[mi][2 + ni], b1[ni][0], af[mi][0]);
                    }
                } else {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) mfma_inplace(acc[mi][ni], b0[ni][0], af[mi][0]);
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) mfma_inplace(acc[mi][2 + ni], b1[ni][0], af[mi][0]);
                    }
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) mfma_inplace(acc[mi][ni], b0[ni][1], af[mi][1]);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) mfma_inplace(acc[mi][2 + ni], b1[ni][1], af[mi][1]);
                }
                __builtin_amdgcn_s_setprio(0);
                if (e2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                st = st == 2 ? 0 : st + 1;
            } while (++q < nkt);
        } else
        do {  // nkt >= 1 always; the do-while form keeps one accumulator live range (no zero-trip merge after the loop)
            const int kt = kb + q;
            const int buf = (q + 1) & 1;
            const char* cur = smem + buf * BUF;
            const int g = 4 * q;
            const bool steady = q + 2 < nkt;
            read_a(cur, 0);
            read_b(b0, cur, 0);
            if (q + 1 < nkt) stage(K_B1{}, kt + 1, buf ^ 1);
            if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 6, last) - (g + 2));
            mma_quadrant(H0{}, H0{}, b0);
            read_b(b1, cur, 1);
            if (q + 1 < nkt) stage(K_A1{}, kt + 1, buf ^ 1);
            if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 7, last) - (g + 3));
            mma_quadrant(H0{}, H1{}, b1);
            read_a(cur, 1);
            if (q + 2 < nkt) stage(K_A0{}, kt + 2, buf);
            if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(min(g + 8, last) - (g + 4));
            mma_quadrant(H1{}, H1{}, b1);
            if (q + 2 < nkt) stage(K_B0{}, kt + 2, buf);
            if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else wait_halftiles(max(min(g + 9, last) - (g + 5), 0));
            mma_quadrant(H1{}, H0{}, b0);
        } while (++q < nkt);
        asm volatile("s_nop 15" ::: "memory");  // asm MFMA results -> first compiler-generated reader (hipcc pads nothing for asm)
        if (wr == 0) __builtin_amdgcn_s_barrier();
        __syncthreads();  // every wave is past its last fragment read: both buffers are free
#ifdef RGA3_AB
        if (sk.dbg && tid == 0 && it < 8) sk.dbg[(size_t)w * 64 + it * 8 + 2] = __builtin_amdgcn_s_memtime();
#endif

        const int cur_kind = kind;
        const bool has_next = it + 1 < n_items;
        // everything below is per-item work: keep its lane-dependent address math out of the main loop's live ranges
        // (the compiler otherwise hoists it above the item loop and spills inside the MFMA phases)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if (has_next) {
            item(it + 1, tile, kb, ke, kind);
            setup_tile(tile, lane_e);
            if constexpr (PREFETCH) prologue_issue(kb, ke);
        }

#ifdef RGA3_AB
        if (sk.dbg && tid == 0 && it < 8) sk.dbg[(size_t)w * 64 + it * 8 + 3] = __builtin_amdgcn_s_memtime();
#endif
        if (cur_kind == 1) {
            // ---- non-owner slice: partial sums -> this workgroup's slab, then publish
            f32x4* slab = (f32x4*)sk.slabs + (size_t)w * (512 * 32) + (size_t)wid * (32 * 64) + lane_e;
            // Slab stores are WRITE-THROUGH (sc1: cdna_hip_programming.md Guideline 16 R1, MI355X_MICROARCH.md price list "publish-large"): every storing wave drains
            // its own stores, the workgroup meets, one lane raises the flag -- no agent-scope release fence.  The plain-store form paid a buffer_wbl2 (write back
            // the XCD L2's dirty lines) behind 256 KiB of freshly dirtied slab per workgroup: 8.2 vs 3.0 us per publish in the guide's measurement.  The owner
            // still acquires (agent scope) before its plain loads.  In split-K mode (tile 25) the kernel boundary is the hand-off: plain stores there.
            if (RGOK && rg) {   // ragged tile: the eight live quads (m-tiles 0, 1), at the positions the owner's two-m-tile epilogue reads
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slab), "v"(acc[i][j]) : "memory");
                        slab += 64;
                        asm volatile("" : "+v"(slab));
                    }
            } else if (sk.all_partial || sk.plain_slabs) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        *slab = acc[i][j];
                        slab += 64;
                        asm volatile("" : "+v"(slab));  // one running address, not 32 precomputed ones
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slab), "v"(acc[i][j]) : "memory");
                        slab += 64;
                        asm volatile("" : "+v"(slab));
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !sk.all_partial) {   // split-K mode hands over at the kernel boundary instead
#ifdef RGA3_AB
                if (sk.plain_slabs) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#endif
                __hip_atomic_store(sk.flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            // the LAST item of a workgroup has no prologue in flight: both buffers are free, and the wave's residual rows go through LDS in one burst (before an
            // owner slice starts waiting for its contributors)
            const char* resl = nullptr;
            char* est_i = est;
            if constexpr (pp_res_lds<ACT, OUT_F32>()) {
                if (!has_next && !(RGOK && rg) && residual_stageable<64, OUT_F32>(p, n0 + wc * 64, p.N)) {
                    char* dst = smem + 8 * 1024 + wid * res_stage_bytes<MT, 64>();
                    stage_residual<MT, 64>(p, dst, m0 + wr * 32 * MH, n0 + wc * 64, lane_e);
                    resl = dst;
                    est_i = smem + wid * 1024;
                }
            }
            const f32x4 *part1 = nullptr, *part2 = nullptr;
            float poison = 0.f;
            int fl1 = -1, fl2 = -1;   // contributors whose flags the owner re-arms
            if (cur_kind == 2) {
                // ---- owner slice: add the partial sums of the (at most two: P_sk <= 2 * sk_tiles) lower-numbered
                //      workgroups that computed this tile's earlier K-iterations
                const unsigned sa = (unsigned)(ow_tile - sk.t_dp) * (unsigned)nk;   // first iteration of this tile on the tail line
                // contributors = the nearest lower-numbered workgroups with a run (the host's plan leaves at most two slices before the owner's)
                int c1 = w - 1;
                while (c1 > 0 && sk.start[c1] == sk.start[c1 + 1]) --c1;
                int c2 = -1;
                if (sk.start[c1] > sa) {
                    c2 = c1 - 1;
                    while (c2 > 0 && sk.start[c2] == sk.start[c2 + 1]) --c2;
                }
                const bool two = c2 >= 0;
                volatile unsigned* gave_up = (volatile unsigned*)(smem + sk_lds_bytes<ACT, OUT_F32, MH>());   // 16 bytes behind everything else in LDS
                if (tid == 0) {
                    // sk.tmo[1]: fault injection for tests (0 in every real run: the caller zeroes the flag page): 0xffffffff = behave as if a contributor never
                    // publishes; any other non-zero value = spin limit
                    const unsigned inject = __hip_atomic_load(sk.tmo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned limit = (inject != 0u && inject != 0xffffffffu) ? inject : (1u << 22);
                    unsigned gave = 0;
                    for (int c = 0; c < (two ? 2 : 1); ++c) {
                        unsigned spins = 0;
                        while (inject == 0xffffffffu || __hip_atomic_load(sk.flags + (c ? c2 : c1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (inject == 0xffffffffu || ++spins > limit) {
                                __hip_atomic_fetch_add(sk.tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                gave = 1;
                                break;
                            }
                        }
                    }
                    *gave_up = gave;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                // a give-up used to fall through and sum whatever was in the slab: a finite, wrong tile.  Now the tile is poisoned (+inf into every sum)
                poison = __builtin_amdgcn_readfirstlane((int)*gave_up) ? __builtin_inff() : 0.f;   // wave-uniform: lives in a scalar register
                part1 = (const f32x4*)sk.slabs + (size_t)c1 * (512 * 32) + (size_t)wid * (32 * 64) + lane_e;
                if (two) part2 = (const f32x4*)sk.slabs + (size_t)c2 * (512 * 32) + (size_t)wid * (32 * 64) + lane_e;
                fl1 = c1; fl2 = c2;
            }
            if (RGOK && rg) {   // two m-tiles per wave: rows m0 + wr * 32 + [0, 32)
                f32x4 (&acc2)[2][NTL] = *reinterpret_cast<f32x4 (*)[2][NTL]>(&acc[0][0]);
                if (cur_kind == 2) gemm_epilogue<2, NTL, 32, 64, ACT, OUT_F32, false, true>(acc2, p, est_i, smem + 2 * BUF, lane_e, m0, n0, wr, wc, part1, part2, nullptr, poison);
                else gemm_epilogue<2, NTL, 32, 64, ACT, OUT_F32, false, false>(acc2, p, est_i, smem + 2 * BUF, lane_e, m0, n0, wr, wc, nullptr, nullptr, nullptr);
            } else if (cur_kind == 2) gemm_epilogue<MT, NTL, 32 * MH, 64, ACT, OUT_F32, false, true>(acc, p, est_i, smem + 2 * BUF, lane_e, m0, n0, wr, wc, part1, part2, resl, poison);
            else gemm_epilogue<MT, NTL, 32 * MH, 64, ACT, OUT_F32, false, false>(acc, p, est_i, smem + 2 * BUF, lane_e, m0, n0, wr, wc, nullptr, nullptr, resl);
            if (cur_kind == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // every wave has consumed its slab values ...
                if (tid == 0) {   // ... re-arm the flags for the next launch
                    __hip_atomic_store(sk.flags + fl1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (fl2 >= 0) __hip_atomic_store(sk.flags + fl2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
#ifdef RGA3_AB
        if (sk.dbg && tid == 0 && it < 8) sk.dbg[(size_t)w * 64 + it * 8 + 4] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();  // epilogue staging reads are done before phase 0 / 1 of the next item restage A1 / B1 of buffer 0
#ifdef RGA3_AB
        if (sk.dbg && tid == 0 && it < 8) sk.dbg[(size_t)w * 64 + it * 8 + 5] = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (!PREFETCH) {
            if (has_next) prologue_issue(kb, ke);
        }
    }
}

// Split-K for few-tile / huge-K products (tile 25; weight gradients dW = dY^T X over 10^6 tokens are ONE 256x256 tile with K = 16 384
// K-tiles: 1.9 ms on one CU): every tile is cut into S equal K-slices, one workgroup each (gemm_nt_sk_kernel with all_partial), and this
// kernel adds the S slabs of a tile in slice order and writes C (bias optional, bf16 or f32).  Thread = one accumulator quad.
struct SlabReduceArgs {
    const float* slabs;
    void* C;
    const unsigned short* bias;
    int M, N, ntm, ntn, group_m, S, out_f32;
    long ldc;
};

__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(SlabReduceArgs p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // ((tile * 8 + wave) * 32 + quad) * 64 + lane
    const long total = (long)p.ntm * p.ntn * 8 * 32 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), quad = (int)((idx >> 6) & 31), wave = (int)((idx >> 11) & 7);
    const unsigned t = (unsigned)(idx >> 14);
    const unsigned GROUP_M = (unsigned)p.group_m;
    const unsigned per_group = GROUP_M * p.ntn;
    const unsigned group = t / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = min((unsigned)p.ntm - first_m, GROUP_M);
    const int m0 = (int)(first_m + (t % per_group) % gsz) * 256, n0 = (int)((t % per_group) / gsz) * 256;
    const float* src = p.slabs + (((size_t)t * p.S) * (512 * 32) + (size_t)wave * (32 * 64) + (size_t)quad * 64 + lane) * 4;
    f32x4 acc = *(const f32x4*)src;
    for (int s = 1; s < p.S; ++s) acc += *(const f32x4*)(src + (size_t)s * (512 * 32) * 4);
    const int i = quad >> 2, j = quad & 3, g = lane >> 4, c = lane & 15;
    const int row = m0 + (wave >> 2) * 128 + i * 16 + c;
    const int col = n0 + (wave & 3) * 64 + j * 16 + 4 * g;
    if (row >= p.M) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (col + r >= p.N) break;
        float v = acc[r];
        if (p.bias) v += bf2f(p.bias[col + r]);
        if (p.out_f32) ((float*)p.C)[(long)row * p.ldc + col + r] = v;
        else ((unsigned short*)p.C)[(long)row * p.ldc + col + r] = f2bf(v);
    }
}

// =====================================================================================================================
// 256x256 tile on FOUR waves (tile id 28; one tile per workgroup, K % 64 == 0): wave blocks of 128 x 128, accumulators in the 256 AGPRs, one wave per SIMD.
// Why: the 8-wave loops read 24 KiB of fragments per wave and K-tile -- 192 KiB per CU against 64 KiB staged: the LDS port is as busy as the matrix pipe (1536 + 512
// of 2048 cycles) and the main loop runs at 75 % of the MFMA rate.  128 x 128 wave blocks read 32 KiB per wave = 128 KiB per CU (-33 %).  A lone wave per SIMD overlaps
// nothing by itself, so the stream is hand-ordered (round 5 lesson 1): every fragment read and every LDS-DMA piece sits between two MFMAs, all as asm statements in
// source order; fragments of quadrant g + 1 are read while quadrant g multiplies.
//  * LDS image, swizzle, half-tiles A0 A1 B0 B1 as in the ping-pong kernels; a half-tile = 16 one-KiB pieces, four per wave.
//  * per K-tile four quadrants of 32 MFMAs:  Q0 = A0 B0,  Q1 = A0 B1,  Q2 = A1 B1,  Q3 = A1 B0, and TWO barriers (the four waves run in lockstep): X in front of Q0,
//    Y in front of Q2; the schedule is at the K-tile lambda below.
//  * B0 fragments live in two register sets (B0 of K-tile q is still multiplied in Q3 while B0 of q + 1 is read): the loop is unrolled by two K-tiles.
__device__ __forceinline__ void mfma_agpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void lds_frag(bf16x8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;
    constexpr int HALF = 128 * ROWB, BUF = 4 * HALF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const unsigned nwg = (unsigned)(p.ntm * p.ntn);
    const unsigned t = xcd_remap(blockIdx.x, nwg);
    const unsigned GROUP_M = (unsigned)p.group_m;
    const unsigned per_group = GROUP_M * p.ntn;
    const unsigned group = t / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = min((unsigned)p.ntm - first_m, GROUP_M);
    const int m0 = (int)(first_m + (t % per_group) % gsz) * BM;
    const int n0 = (int)((t % per_group) / gsz) * BN;
    const int nk = p.K / BK;

    // staging: piece pc = wid + 4 i of a half-tile = LDS rows 8 pc .. 8 pc + 7
    const int sch = (lane & 7) ^ ((((wid & 1) << 2) + (lane >> 4)) & 7);
    unsigned soff[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wid + 4 * i) * 8 + (lane >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (r >> 6) * 128 + h * 64 + (r & 63);
            const int bcol = (r >> 5) * 64 + h * 32 + (r & 31);
            soff[h][i] = (unsigned)((long)min(m0 + arow, p.M - 1) * p.lda + sch * 8);
            soff[2 + h][i] = (unsigned)((long)min(n0 + bcol, p.N - 1) * p.ldw + sch * 8);
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;   // LDS byte address of the buffers
    auto stage_piece = [&](auto KIND, int i, int kt, int buf) {
        constexpr int kind = decltype(KIND)::value;
        const unsigned short* base = ((kind < 2) ? p.A : p.W) + (long)kt * BK;
        unsigned o = soff[kind][i];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((gbl_void*)(base + o), (lds_void*)(smem + buf * BUF + kind * HALF + (wid + 4 * i) * 1024), 16, 0, 0);
    };
    using K_A0 = std::integral_constant<int, 0>;
    using K_A1 = std::integral_constant<int, 1>;
    using K_B0 = std::integral_constant<int, 2>;
    using K_B1 = std::integral_constant<int, 3>;
    auto stage_all = [&](auto KIND, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_piece(KIND, i, kt, buf);
    };

    // fragment addresses: one base register per (operand side, kk, buffer); half-tile and m- / n-tile offsets ride in the instruction's offset field
    unsigned fa[2][2], fb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned f = lds0 + (unsigned)(b * BUF) + (unsigned)((lane & 15) * ROWB + (((kk * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4));
            fa[b][kk] = f + (unsigned)((wr * 64) * ROWB);
            fb[b][kk] = f + (unsigned)(2 * HALF + (wc * 64) * ROWB);
        }
    auto rd = [&](bf16x8& d, unsigned base, int off) {   // off folds to a constant after unrolling ("i": checked by the backend)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(base), "i"(off) : "memory");
    };

    f32x4 acc[8][8];
    bf16x8 a0[4][2], a1[4][2], b1[4][2], b0x[4][2], b0y[4][2];

    if constexpr (act_uses_table<ACT>()) stage_act_table<4>(smem + 2 * BUF, ACT == ACT_SWIGLU, wid, lane);

    // ---- prologue, in the steady-state issue order  [A0 B0](0) [B1 A1](0) [A0 B0](1) [B1 A1](1)
    stage_all(K_A0{}, 0, 0);
    stage_all(K_B0{}, 0, 0);
    stage_all(K_B1{}, 0, 0);
    stage_all(K_A1{}, 0, 0);
    if (nk > 1) {
        stage_all(K_A0{}, 1, 1);
        stage_all(K_B0{}, 1, 1);
        stage_all(K_B1{}, 1, 1);
        stage_all(K_A1{}, 1, 1);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // [A0 B0](0) have landed
    } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            rd(a0[mi][kk], fa[0][kk], mi * 2048);
            rd(b0x[mi][kk], fb[0][kk], mi * 2048);
        }

    // quadrant (ha, hb): acc[ha * 4 + mi][(ni >> 1) * 4 + hb * 2 + (ni & 1)] += B frag (as the MFMA's A operand) x A frag; 32 MFMAs, `between(s)` runs after MFMA s.
    // FIRST (K-tile 0): the first k-step takes the constant 0 as C operand -- the accumulators are defined by MFMAs and never exist outside the AGPRs inside the loop
    auto quadrant = [&](auto FIRST, auto HA, auto HB, const bf16x8 (&af)[4][2], const bf16x8 (&bf)[4][2], auto&& between) {
        constexpr int ha = decltype(HA)::value, hb = decltype(HB)::value;
        constexpr bool first = decltype(FIRST)::value;
        int s = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f32x4& c = acc[ha * 4 + mi][(ni >> 1) * 4 + hb * 2 + (ni & 1)];
                    if (first && kk == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(bf[ni][kk]), "v"(af[mi][kk]));
                    else mfma_agpr(c, bf[ni][kk], af[mi][kk]);
                    between(s);
                    ++s;
                }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using TRUE_T = std::true_type;
    using FALSE_T = std::false_type;

    // One K-tile in buffer PARITY: TWO barriers.  X (start): [B1 A1](q) have landed, every wave has retired its reads of [A0 B0](q) -- their regions take [A0 B0](q+2)
    // during Q0 / Q1.  Y (middle): [A0 B0](q+1) have landed, the reads of [B1 A1](q) are retired -- their regions take [B1 A1](q+2) during Q2 / Q3.  In front of either
    // barrier exactly 16 loads have been issued after the youngest half-tile it needs: vmcnt(16) (vmcnt(0) once nothing is left to issue).
    // B0C = this K-tile's B0 fragments, B0N = the set the next K-tile's are read into.  STEADY: K-tiles q + 1 and q + 2 exist (no conditions inside).
    auto ktile = [&](auto FIRST, auto PARITY, auto STEADY, int q, bf16x8 (&B0C)[4][2], bf16x8 (&B0N)[4][2]) {
        constexpr int cur = decltype(PARITY)::value, nxt = cur ^ 1;
        constexpr bool steady = decltype(STEADY)::value;
        const bool e1 = steady || q + 1 < nk, e2 = steady || q + 2 < nk;
        const unsigned short* const gA = p.A + (long)(q + 2) * BK;
        const unsigned short* const gW = p.W + (long)(q + 2) * BK;
        auto piece = [&](const unsigned short* g, int kind, int i) {
            unsigned o = soff[kind][i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + o), (lds_void*)(smem + cur * BUF + kind * HALF + (wid + 4 * i) * 1024), 16, 0, 0);
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (e2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- Q0 = A0 B0: read B1(q); issue A0(q+2)
        quadrant(FIRST, H0{}, H0{}, a0, B0C, [&](int s) {
            if (s < 8) rd(b1[s >> 1][s & 1], fb[cur][s & 1], HALF + (s >> 1) * 2048);
            else if (s >= 12 && s < 28 && ((s - 12) & 3) == 0 && e2) piece(gA, 0, (s - 12) >> 2);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- Q1 = A0 B1: read A1(q); issue B0(q+2)
        quadrant(FIRST, H0{}, H1{}, a0, b1, [&](int s) {
            if (s < 8) rd(a1[s >> 1][s & 1], fa[cur][s & 1], HALF + (s >> 1) * 2048);
            else if (s >= 12 && s < 28 && ((s - 12) & 3) == 0 && e2) piece(gW, 2, (s - 12) >> 2);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (e2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- Q2 = A1 B1: issue B1(q+2); read A0(q+1)
        quadrant(FIRST, H1{}, H1{}, a1, b1, [&](int s) {
            if (s < 8) { if (e1) rd(a0[s >> 1][s & 1], fa[nxt][s & 1], (s >> 1) * 2048); }
            else if (s >= 12 && s < 28 && ((s - 12) & 3) == 0 && e2) piece(gW, 3, (s - 12) >> 2);
        });
        // ---- Q3 = A1 B0: issue A1(q+2); read B0(q+1)
        quadrant(FIRST, H1{}, H0{}, a1, B0C, [&](int s) {
            if (s < 8) { if (e1) rd(B0N[s >> 1][s & 1], fb[nxt][s & 1], (s >> 1) * 2048); }
            else if (s >= 12 && s < 28 && ((s - 12) & 3) == 0 && e2) piece(gA, 1, (s - 12) >> 2);
        });
    };
    // K-tile 0, then pairs (odd buffer, even buffer); the last two K-tiles run the conditional form
    if (nk >= 3) ktile(TRUE_T{}, H0{}, TRUE_T{}, 0, b0x, b0y); else ktile(TRUE_T{}, H0{}, FALSE_T{}, 0, b0x, b0y);
    int q = 1;
    for (; q + 3 < nk; q += 2) {
        ktile(FALSE_T{}, H1{}, TRUE_T{}, q, b0y, b0x);
        ktile(FALSE_T{}, H0{}, TRUE_T{}, q + 1, b0x, b0y);
    }
    for (; q < nk; q += 2) {
        ktile(FALSE_T{}, H1{}, FALSE_T{}, q, b0y, b0x);
        if (q + 1 < nk) ktile(FALSE_T{}, H0{}, FALSE_T{}, q + 1, b0x, b0y);
    }
    asm volatile("s_nop 15" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_epilogue<8, 8, 128, 128, ACT, OUT_F32>(acc, p, smem + wid * 1024, smem + 2 * BUF, lane, m0, n0, wr, wc);
}

// Tile order: consecutive logical tiles walk down GROUP_M tile rows, then step to the next tile column.  An XCD holds 32
// consecutive tiles, i.e. GROUP_M x (32 / GROUP_M) of them share its L2.  GROUP_M = 4 keeps that block squarish (least
// L2 fill traffic) but walks the whole weight matrix once per group of 4 tile rows: with few tile rows (M = 2112 is 9) and
// a weight matrix larger than the 256 MiB Infinity Cache that is 3 passes over HBM; one group spanning all rows streams the
// weights once and re-reads the small activation block from the Infinity Cache instead.
static int pick_group_m(int ntm, int tile_m) {
#ifdef RGA3_AB   // measurement builds only (tools/): the product library has one behaviour
    static const int forced = [] { const char* e = getenv("RGA3_GEMM_GROUPM"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
#endif
    (void)tile_m;
    return (ntm <= 16) ? ntm : 4;
}

struct SkWorkspace { float* slabs; unsigned* flags; int P; };

// The stream-K / split-K tilings need f32 slabs (256 KiB per CU) and one flag word per CU.  The CALLER owns that memory (SURVEY.md 8(b):
// the library allocates nothing): rga3_gemm_bf16 takes it as (workspace, workspace_bytes); layout = [4 KiB of flags, zeroed once by the
// caller and kept between calls on the same stream][slabs].  Without a large enough workspace tiles 22 / 25 run as 21 / 20.
constexpr size_t kSkFlagBytes = 4096;

static int g_cu_count[16] = {0};

static int cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (g_cu_count[dev] == 0) {   // read-only per-device cache, filled on first use
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
        g_cu_count[dev] = cus;
    }
    return g_cu_count[dev];
}

// 0: workspace usable (out filled); 1: not usable (caller falls back to a tiling without workspace)
static int sk_workspace(void* ws, int64_t ws_bytes, SkWorkspace& out) {
    const int cus = cu_count();
    if (!ws || cus <= 0 || cus * 4 + 16 > (int)kSkFlagBytes) return 1;
    const size_t need = kSkFlagBytes + (size_t)cus * 512 * 32 * 16;
    if ((size_t)ws_bytes < need || (((uintptr_t)ws) & 15)) return 1;
    out.P = cus;
    out.flags = (unsigned*)ws;
    out.slabs = (float*)((char*)ws + kSkFlagBytes);
    return 0;
}

static inline int sk_plain_slabs() {
#ifdef RGA3_AB
    const char* e = getenv("RGA3_SK_PLAIN");
    return e && atoi(e) != 0;
#else
    return 0;
#endif
}

template <int ACT, bool OUT_F32, int LNF = 0, bool N192 = false>
static int launch_pp(const GemmArgs& a0, hipStream_t st);

// ---- run tables of the stream-K tail (SkArgs::start)
// equal runs over the first P_sk workgroups (rounds 2 - 4: the kernel computed w * tot / P_sk itself)
static void sk_plan_uniform(SkArgs& sk, int nk) {
    const long tot = (long)sk.sk_tiles * nk;
    for (int w = 0; w <= kSkMaxP; ++w) sk.start[w] = (unsigned)((sk.P_sk > 0 && w < sk.P_sk) ? (long)w * tot / sk.P_sk : tot);
}

// Cost of a ragged tile in sixteenths of a full tile's time (tools/probes/ragged_cost.py, ragged_items.py: 0.58 next to the full tiles of its column, 0.62 - 0.8 when it
// streams its weight columns from HBM alone; its loop without the loads 0.35)
static int sk_ragged_cost16() {
#ifdef RGA3_AB
    static const int forced = [] { const char* e = getenv("RGA3_SK_RAGGED_COST"); return e ? atoi(e) : 0; }();
    if (forced > 0 && forced <= 16) return forced;
#endif
    return 10;
}

// Plan for a product whose last tile row is ragged (<= 64 rows: its ntn tiles run the quarter-work loop at 0.6 of a full tile's time).
//  * At least one round of tiles: G workgroups take ragged tiles only, P - G the full tiles (SkArgs::rg_wgs); G minimises max(ceil(ntn / G) c, rounds of the full
//    tiles over P - G workgroups).  M = 2112 gate | up: 1184 full + 148 ragged tiles: 19 workgroups x 8 ragged tiles (5.0 rounds at c = 0.625), 237 x 5 full
//    tiles -- five rounds where tile 22 runs five and a half and padded data-parallel rounds six.  (Two ragged tiles as ONE
//    unit of the strided rounds -- the first form of these tiles -- left every pair 0.24 of a round over: 464 us against 486 for tile 22.)
//    split: the full tiles beyond whole rounds of P - G go to the stream-K tail as tile 22's equal runs over the full-tile workgroups (ragged ones get an empty run).
//  * Less than one round of tiles: no plan (tiles 22 / 21 as they are).  Tried: everything in the tail, ragged tiles first, equal runs in unit time -- the critical path
//    stays half a full tile (down projection 247 against 239 us).  Also tried and dropped: tail runs sized by water-filling the workgroups' data-parallel cost to one
//    level (every slice at its own K offset: the workgroups of an XCD stopped sharing operand tiles in L2 and the tail ran 1.75 x slower per iteration), and the natural
//    tile order with the tail's half-tile slices handed to the workgroups that met ragged tiles (gate | up 494 against 478 us for tile 22; profiles/r05_ragged_experiments.log).
static bool sk_plan_ragged(SkArgs& sk, const GemmArgs& a, int nk, int P, bool split) {
    constexpr int MIN_SEG = 8;
    const int ntn = a.ntn, T = a.ntm * a.ntn, T_full = T - ntn;
    sk.ragged = 1;
    sk.rg_wgs = 0;
    sk.all_partial = 0;
    if (T < P) return false;   // less than one round of tiles: tile 22's half-tile slices are the critical path either way (a whole ragged tile costs more than half a full one)
    // G = the number of ragged workgroups with the shortest critical path (sixteenths of a full tile's time): ragged side ceil(ntn / G) c, full-tile side whole rounds
    // of P - G plus the tail (split) or the partly filled last round
    const int c16 = sk_ragged_cost16();
    int G = 0;
    long best = -1;
    for (int g = 1; g < P / 2 && g <= ntn; ++g) {
        const int pf = P - g;
        const long rag = cdiv(ntn, g) * c16;
        long full;
        if (!split) full = 16L * cdiv(T_full, pf);
        else {
            const int r = T_full % pf;
            long psk = 2L * r;
            if (psk > pf) psk = pf;
            full = 16L * (T_full / pf) + (r ? cdiv(16L * r, psk) : 0);
        }
        const long ms = rag > full ? rag : full;
        if (best < 0 || ms < best) { best = ms; G = g; }
    }
    if (G < 1) return false;
    const int Pf = P - G;
    sk.P = P;
    sk.rg_wgs = G;
    const int rem = split ? T_full % Pf : 0;
    sk.t_dp = T - rem;
    sk.sk_tiles = rem;
    if (rem == 0) {
        sk.P_sk = 0;
        sk_plan_uniform(sk, nk);
        return true;
    }
    long cap = (long)rem * nk / MIN_SEG, want = 2L * rem;
    if (want > cap) want = cap;
    if (want > Pf) want = Pf;
    if (want < rem) want = rem;
    sk.P_sk = (int)want;
    // equal runs over the first P_sk full-tile workgroups
    const long tot = (long)rem * nk;
    int rk = 0;
    sk.start[0] = 0;
    for (int w = 0; w < P; ++w) {
        const bool rgw = ((long)(w + 1) * G) / P > ((long)w * G) / P;
        if (!rgw && rk < sk.P_sk) ++rk;
        sk.start[w + 1] = (unsigned)(rk < sk.P_sk ? (long)rk * tot / sk.P_sk : tot);
    }
    for (int w = P + 1; w <= kSkMaxP; ++w) sk.start[w] = (unsigned)tot;
    return true;
}

template <int ACT, bool OUT_F32, int MH = 4>
static int launch_sk(const GemmArgs& a0, bool split, hipStream_t st, bool ragged = false) {
    void* ws_ptr = a0.ws; const int64_t ws_bytes = a0.ws_bytes;
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, 64 * MH);
    a.ntn = (int)cdiv(a.N, 256);
    a.group_m = pick_group_m(a.ntm, 64 * MH);
    SkWorkspace ws;
    if (a.K % 64 != 0) return launch_pp<ACT, OUT_F32>(a0, st);  // ragged K: zero-source tail lives in the one-tile kernel
    ws.P = cu_count();
    ws.slabs = nullptr;
    ws.flags = nullptr;
    if (ws.P <= 0 || ws.P > kSkMaxP) return launch_pp<ACT, OUT_F32>(a0, st);
    if (split && sk_workspace(ws_ptr, ws_bytes, ws)) split = false;   // no (or too small a) caller workspace: persistent without the stream-K tail
    const int T = a.ntm * a.ntn;
    const int nk = (int)cdiv(a.K, 64);
    SkArgs sk;
    sk.slabs = ws.slabs;
    sk.flags = ws.flags;
    sk.tmo = ws.flags ? ws.flags + ws.P : nullptr;
    const int rem = T % ws.P;
    sk.ragged = 0;
    sk.rg_wgs = 0;
    // ragged last tile row (MH = 4, more than one tile row): workgroups of their own for the ragged tiles
    const int last_rows = a.M - (a.ntm - 1) * 256;
    bool planned = false;
    if (MH == 4 && ragged && a.ntm >= 2 && last_rows <= 64) {
        a.group_m = pick_group_m(a.ntm - 1, 256);   // the grouped order covers the full tile rows only
        sk.plain_slabs = sk_plain_slabs();
        planned = sk_plan_ragged(sk, a, nk, ws.P, split);
        if (!planned) { sk.ragged = 0; sk.rg_wgs = 0; a.group_m = pick_group_m(a.ntm, 256); }
    }
    if (planned) {
    } else if (!split || rem == 0) {
        sk.P = T < ws.P ? T : ws.P;
        sk.t_dp = T;
        sk.sk_tiles = 0;
        sk.P_sk = 0;
        sk.all_partial = 0;
        sk.plain_slabs = sk_plain_slabs();
        sk_plan_uniform(sk, nk);
    } else {
        // slices per split tile <= 3 (owner + two contributors: the kernel's accumulator init reads at most two slabs)
        // <=> run length >= nk / 2 <=> P_sk <= 2 * sk_tiles; and no slice shorter than MIN_SEG K-iterations
        constexpr int MIN_SEG = 8;
        sk.P = ws.P;
        sk.t_dp = T - rem;
        sk.sk_tiles = rem;
        long cap = (long)rem * nk / MIN_SEG;
        long want = 2L * rem;
        if (want > cap) want = cap;
        if (want > ws.P) want = ws.P;
        if (want < rem) want = rem;
        sk.P_sk = (int)want;
        sk.all_partial = 0;
        sk.plain_slabs = sk_plain_slabs();
        sk_plan_uniform(sk, nk);
    }
#ifdef RGA3_AB
    {   // stamps go to the start of the slab area: tile 21 / 31 only (no stream-K tail writes slabs there)
        static const bool dbg = [] { const char* e = getenv("RGA3_SK_DBG"); return e && atoi(e) != 0; }();
        SkWorkspace w2;
        sk.dbg = (dbg && !split && !sk_workspace(ws_ptr, ws_bytes, w2)) ? (unsigned long long*)w2.slabs : nullptr;
    }
#endif
    constexpr int LDS = sk_lds_bytes<ACT, OUT_F32, MH>() + 16;
    static_assert(LDS <= 160 * 1024, "persistent kernel: LDS");
    if constexpr (MH == 4) {
        if (sk.ragged) {
            auto kern = gemm_nt_sk_kernel<ACT, OUT_F32, 4, true>;
            static LdsGrant lds_grant_rg;
            if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant_rg, "gemm")) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)sk.P), dim3(512), LDS, st, a, sk);
            RGA3_CHECK_LAUNCH("gemm_nt_sk_kernel<ragged>");
            return 0;
        }
    }
    auto kern = gemm_nt_sk_kernel<ACT, OUT_F32, MH>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "gemm")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)sk.P), dim3(512), LDS, st, a, sk);
    RGA3_CHECK_LAUNCH("gemm_nt_sk_kernel");
    return 0;
}

// tile 25: falls back to the persistent kernel when the product does not have few tiles and a long K
template <int ACT, bool OUT_F32>
static int launch_splitk(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, 256);
    a.ntn = (int)cdiv(a.N, 256);
    a.group_m = pick_group_m(a.ntm, 256);
    const int T = a.ntm * a.ntn;
    const int nk = (int)cdiv(a.K, 64);
    SkWorkspace ws;
    if (ACT != ACT_NONE || a.res || a.colscale || a.K % 64 != 0) return launch_sk<ACT, OUT_F32>(a0, false, st);
    if (sk_workspace(a0.ws, a0.ws_bytes, ws)) return launch_pp<ACT, OUT_F32>(a0, st);
    int S = ws.P / T;
    if (S > nk / 4) S = nk / 4;          // at least 4 K-tiles per slice
    if (S < 2) return launch_sk<ACT, OUT_F32>(a0, false, st);
    SkArgs sk;
    sk.slabs = ws.slabs; sk.flags = ws.flags; sk.tmo = ws.flags + ws.P;
    sk.P = T * S; sk.P_sk = T * S; sk.t_dp = 0; sk.sk_tiles = T; sk.all_partial = 1; sk.plain_slabs = 0; sk.ragged = 0; sk.rg_wgs = 0;
    if (sk.P > kSkMaxP) return launch_sk<ACT, OUT_F32>(a0, false, st);
    sk_plan_uniform(sk, nk);
    constexpr int LDS = 2 * 4 * 128 * 128;
    auto kern = gemm_nt_sk_kernel<ACT_NONE, false>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "gemm")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)sk.P), dim3(512), LDS, st, a, sk);
    RGA3_CHECK_LAUNCH("gemm_nt_sk_kernel<split-K>");
    SlabReduceArgs r;
    r.slabs = ws.slabs; r.C = a.C; r.bias = a.bias; r.M = a.M; r.N = a.N; r.ntm = a.ntm; r.ntn = a.ntn; r.group_m = a.group_m; r.S = S;
    r.out_f32 = OUT_F32 ? 1 : 0; r.ldc = a.ldc;
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)cdiv((long)T * 8 * 32 * 64, 256)), dim3(256), 0, st, r);
    RGA3_CHECK_LAUNCH("gemm_slab_reduce_kernel");
    return 0;
}

template <int ACT, bool OUT_F32, int LNF, bool N192>
static int launch_pp(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, 256);
    a.ntn = (int)cdiv(a.N, N192 ? 192 : 256);
    a.group_m = pick_group_m(a.ntm, 256);
    constexpr int LDS_MAIN = 2 * 4 * 128 * 128 + (act_uses_table<ACT>() ? kActTabBytes : 0);
    constexpr int LDS_RES = pp_res_lds<ACT, OUT_F32>() ? 8 * 1024 + 8 * (N192 ? res_stage_bytes<4, 96>() : res_stage_bytes<8, 64>()) : 0;
    constexpr int LDS = LDS_MAIN > LDS_RES ? LDS_MAIN : LDS_RES;
    static_assert(LDS <= 160 * 1024, "ping-pong kernel: LDS");
    auto kern = gemm_nt_pp_kernel<ACT, OUT_F32, LNF, N192>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "gemm")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.ntm * a.ntn)), dim3(512), LDS, st, a);
    RGA3_CHECK_LAUNCH("gemm_nt_pp_kernel");
    return 0;
}

template <int ACT, bool OUT_F32>
static int launch_w4(const GemmArgs& a0, hipStream_t st) {
    if (a0.K % 64 != 0) return launch_pp<ACT, OUT_F32>(a0, st);
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, 256);
    a.ntn = (int)cdiv(a.N, 256);
    a.group_m = pick_group_m(a.ntm, 256);
    constexpr int LDS = 2 * 4 * 128 * 128 + (act_uses_table<ACT>() ? kActTabBytes : 0);
    auto kern = gemm_nt_w4_kernel<ACT, OUT_F32>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "gemm")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.ntm * a.ntn)), dim3(256), LDS, st, a);
    RGA3_CHECK_LAUNCH("gemm_nt_w4_kernel");
    return 0;
}

template <int BM, int BN, int WM, int WN, int ACT, bool OUT_F32, int PIPE, int LNF = 0>
static int launch_cfg(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, BM);
    a.ntn = (int)cdiv(a.N, BN) + (a.Cn ? (int)cdiv(a.N2, BN) : 0);   // N-side concatenation: the second pair's tile columns follow (N % BN == 0, checked by the entry point)
    a.group_m = pick_group_m(a.ntm, BM);
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int LDS = (PIPE == 3 ? 3 : 2) * STAGE;
    static_assert(LDS <= 160 * 1024, "tile does not fit the CU's LDS");
    auto kern = gemm_nt_kernel<BM, BN, WM, WN, ACT, OUT_F32, PIPE, LNF>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "gemm")) return rc;
    dim3 grid((unsigned)(a.ntm * a.ntn));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), LDS, st, a);
    RGA3_CHECK_LAUNCH("gemm_nt_kernel");
    return 0;
}

__global__ __launch_bounds__(256) void tn_slab_sum_kernel(const float* __restrict__ slabs, void* __restrict__ out, long n, long ldc, int ncols, int Z, int out_f32);

// tile 14: 64 x 64 tiles with K cut into S grid.y slices (f32 slabs [S][M][N] in the caller's workspace, summed in slice order by tn_slab_sum_kernel:
// deterministic).  For skinny products -- LoRA's x A^T and dY B (M = 2112 tokens, N = r = 128, K = 3584: 66 tiles, one 56-K-tile loop each, 40 us on
// 66 of 256 CUs) -- the split fills the chip.  Plain products only (no bias / activation / residual); anything else runs as tile 13.
template <bool OUT_F32>
static int launch_split64(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, 64);
    a.ntn = (int)cdiv(a.N, 64);
    a.group_m = pick_group_m(a.ntm, 64);
    const long tiles = (long)a.ntm * a.ntn;
    const int nk = (int)cdiv(a.K, 64);
    SkWorkspace ws;
    long S = 1024 / tiles;                 // ~4 workgroups of 256 threads per CU
    if (S > nk / 2) S = nk / 2;            // at least 2 K-tiles per slice
    if (S > 32) S = 32;
    if (a.bias || a.res || a.colscale || a.N % 8 != 0 || S < 2 || sk_workspace(a0.ws, a0.ws_bytes, ws)) return launch_cfg<64, 64, 2, 2, ACT_NONE, OUT_F32, 0>(a0, st);
    const long fit = (long)((size_t)ws.P * 512 * 32 * 16 / ((size_t)a.M * a.N * 4));
    if (S > fit) S = fit;
    if (S < 2) return launch_cfg<64, 64, 2, 2, ACT_NONE, OUT_F32, 0>(a0, st);
    a.ksl = (int)cdiv(nk, S);
    S = cdiv(nk, a.ksl);
    a.C = ws.slabs;
    a.ldc = a.N;
    constexpr int LDS = 2 * (64 + 64) * 128;
    auto kern = gemm_nt_kernel<64, 64, 2, 2, ACT_NONE, true, 0>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "gemm")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)S), dim3(256), LDS, st, a);
    RGA3_CHECK_LAUNCH("gemm_nt_kernel<split 64>");
    const long n = (long)a0.M * a0.N;
    hipLaunchKernelGGL(tn_slab_sum_kernel, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, (const float*)ws.slabs, a0.C, n, (long)a0.ldc, (int)a0.N, (int)S, OUT_F32 ? 1 : 0);
    RGA3_CHECK_LAUNCH("tn_slab_sum_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// TN product for weight gradients: C[M, N] = A^T . B with A [K, M] and B [K, N] row-major, K = tokens (dW = dY^T X of nn.Linear /
// LoRA, reference autograd of train_joint.py:534).  The NT kernels above would need both operands transposed first (one extra
// pass over the activations per product, ~600 launches per RGA3 step); here the [k][m] / [k][n] tiles are staged as they lie
// and every MFMA fragment is fetched with two ds_read_b64_tr_b16 (the transposing LDS read: lane (g, c) gets, for its row c of the
// 16-wide tile, the k values 4g..4g+3 and 16+4g..16+4g+3 of the 32-deep step) -- both operands through the SAME recipe, so the
// k order inside a step is permuted identically on both sides and the contraction is unchanged.  128x128 tile, 4 waves (2x2, 64x64
// each), BK = 32, register-staged double buffer; rows padded by 32 B (conflict-free transposed reads, as the attention V image).
struct TnArgs {
    const unsigned short* A;  // [K, M]
    const unsigned short* B;  // [K, N]
    long lda, ldb;
    int K;
    int kt_per_z;   // K-tiles (of 32) per grid.z slice; slices > 0 write f32 partial sums to slab z of p.C (see rga3_gemm_tn_bf16)
    long slab;      // elements per slab (0: no K split)
    // fused slab sum (round 3): with `counters` the workgroup that arrives LAST at an output tile adds that tile's Z slabs in slice order and writes the result --
    // which workgroup does it depends on timing, the order of the additions does not (deterministic); no second launch.  One counter word per output tile, zeroed once
    // by the caller, left zero again by the kernel.
    unsigned* counters;
    void* out;      // final output (bf16 or f32, row stride ldo)
    long ldo;
    int out_f32;
};

template <bool OUT_F32>
__device__ __forceinline__ void gemm_tn_body(GemmArgs p, TnArgs t, const unsigned bx, const unsigned by, const unsigned bz, const unsigned gx, const unsigned gz) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int STRIDE = BM * 2 + 32;          // bytes per staged k-row (both operands: BM == BN)
    constexpr int TILE = BK * STRIDE;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [buf][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int g = lane >> 4, c = lane & 15;
    const int m0 = by * BM, n0 = bx * BN;

    // staging: 32 rows x 16 chunks (16 B) per operand tile = 512 chunks, two per thread
    int srow[2], scol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        srow[i] = idx >> 4;
        scol[i] = (idx & 15) * 8;
    }
    u32x4 ra[2], rb[2];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = kt * BK + srow[i];
            u32x4 za = {0u, 0u, 0u, 0u}, zb = {0u, 0u, 0u, 0u};
            if (k < t.K) {
                // columns beyond M / N only feed output rows / columns that are never stored: clamp the address instead of branching
                za = *(const u32x4*)(t.A + (long)k * t.lda + min(m0 + scol[i], p.M - 8));
                zb = *(const u32x4*)(t.B + (long)k * t.ldb + min(n0 + scol[i], p.N - 8));
            }
            ra[i] = za;
            rb[i] = zb;
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        char* sa = smem + buf * 2 * TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4*)(sa + srow[i] * STRIDE + scol[i] * 2) = ra[i];
            *(u32x4*)(sa + TILE + srow[i] * STRIDE + scol[i] * 2) = rb[i];
        }
    };
    auto frag = [&](const char* base, int col16) __attribute__((always_inline)) -> bf16x8 {
        const char* a0 = base + (4 * g + (c >> 2)) * STRIDE + (col16 * 16 + 4 * (c & 3)) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 16 * STRIDE));
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = (t.K + BK - 1) / BK;
    const int kt0 = (int)bz * t.kt_per_z;            // this workgroup's K range (the whole of K without a split)
    const int nk = min(nk_all, kt0 + t.kt_per_z);
    if (t.slab) p.C = (void*)((float*)p.C + (long)bz * t.slab);
    load_tile(kt0);
    for (int kt = kt0; kt < nk; ++kt) {
        store_tile(kt & 1);
        __syncthreads();   // tile kt visible; everyone is past the reads of tile kt-1 (the other buffer is free for kt+1)
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* sa = smem + (kt & 1) * 2 * TILE;
        const char* sb = sa + TILE;
        bf16x8 af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag(sa, wm * 4 + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = frag(sb, wn * 4 + j);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    const float* ws_base = (const float*)p.C - (t.slab ? (long)bz * t.slab : 0);
    gemm_epilogue<4, 4, 64, 64, ACT_NONE, OUT_F32>(acc, p, smem, nullptr, lane, m0, n0, wm, wn);
    if constexpr (OUT_F32) {
        if (t.counters) {
            // hand-off (cdna_hip_programming.md Guideline 16): every wave drains its slab stores, one lane releases at agent scope and takes a ticket; the last
            // arrival acquires, then the whole workgroup reads the other slices' slabs
            __shared__ unsigned s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned tile = by * gx + bx;
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned prev = __hip_atomic_fetch_add(t.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (prev == gz - 1) ? 1u : 0u;
                if (s_last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            if (s_last) {
                const int Z = (int)gz;
                for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
                    const int r = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                    const int row = m0 + r, col = n0 + c4;
                    if (row < p.M && col < p.N) {          // N % 8 == 0: a quad never straddles the edge
                        const float* src = ws_base + (long)row * p.N + col;
                        f32x4 sum = *(const f32x4*)src;
                        for (int z = 1; z < Z; ++z) sum += *(const f32x4*)(src + (long)z * t.slab);
                        if (t.out_f32) {
                            *(f32x4*)((float*)t.out + (long)row * t.ldo + col) = sum;
                        } else {
                            u32x2 pk;
                            pk[0] = pack_bf2(sum[0], sum[1]);
                            pk[1] = pack_bf2(sum[2], sum[3]);
                            *(u32x2*)((unsigned short*)t.out + (long)row * t.ldo + col) = pk;
                        }
                    }
                }
                if (tid == 0) __hip_atomic_store(t.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch on this stream
            }
        }
    }
}

// out[i] = sum_z slabs[z][i] in fixed order (deterministic), to bf16 or f32; 4 elements per thread
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs p, TnArgs t) {
    gemm_tn_body<OUT_F32>(p, t, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
}

// Up to 4 independent TN products in ONE launch (rga3_gemm_tn_many: the four LoRA weight-gradient products of a decoder layer, dB_q / dA_q / dB_v / dA_v -- each a
// handful of 128 x 128 tiles over K = 2112 tokens -- were four launches of 1 - 28 tiles plus four slab sums).  Every product is K-split into f32 slabs exactly as the
// single form would split it; workgroup b belongs to the product whose [first, first_next) range holds it.
struct TnMany {
    GemmArgs a[4];
    TnArgs t[4];
    unsigned first[5];   // first workgroup of every product; first[n] = grid size
    int n;
};
__global__ __launch_bounds__(256) void gemm_tn_many_kernel(TnMany P) {
    const unsigned b = blockIdx.x;
    int q = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) q += (i < P.n && b >= P.first[i]) ? 1 : 0;
    const unsigned local = b - P.first[q];
    const unsigned gx = (unsigned)P.a[q].ntn, gy = (unsigned)P.a[q].ntm;
    gemm_tn_body<true>(P.a[q], P.t[q], local % gx, (local / gx) % gy, local / (gx * gy), gx, 0u);
}
struct SlabSumMany {
    const float* slabs[4];
    void* out[4];
    long n[4], ldc[4];
    int ncols[4], Z[4], out_f32[4];
    unsigned first[5];   // first 256-thread block of every product
    int np;
};
__global__ __launch_bounds__(256) void tn_slab_sum_many_kernel(SlabSumMany S) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) q += (i < S.np && blockIdx.x >= S.first[i]) ? 1 : 0;
    const long i4 = ((long)(blockIdx.x - S.first[q]) * 256 + threadIdx.x) * 4;
    const long n = S.n[q];
    if (i4 >= n) return;
    const float* slabs = S.slabs[q];
    f32x4 acc = *(const f32x4*)(slabs + i4);
    for (int z = 1; z < S.Z[q]; ++z) acc += *(const f32x4*)(slabs + (long)z * n + i4);
    const long r = i4 / S.ncols[q], c = i4 % S.ncols[q];     // ncols % 8 == 0: a quad never crosses a row
    if (S.out_f32[q]) {
        *(f32x4*)((float*)S.out[q] + r * S.ldc[q] + c) = acc;
    } else {
        u32x2 pk;
        pk[0] = pack_bf2(acc[0], acc[1]);
        pk[1] = pack_bf2(acc[2], acc[3]);
        *(u32x2*)((unsigned short*)S.out[q] + r * S.ldc[q] + c) = pk;
    }
}

__global__ __launch_bounds__(256) void tn_slab_sum_kernel(const float* __restrict__ slabs, void* __restrict__ out, long n, long ldc, int ncols, int Z, int out_f32) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    f32x4 acc = *(const f32x4*)(slabs + i4);
    for (int z = 1; z < Z; ++z) acc += *(const f32x4*)(slabs + (long)z * n + i4);
    const long r = i4 / ncols, c = i4 % ncols;     // ncols % 8 == 0: a quad never crosses a row
    if (out_f32) {
        *(f32x4*)((float*)out + r * ldc + c) = acc;
    } else {
        u32x2 pk;
        pk[0] = pack_bf2(acc[0], acc[1]);
        pk[1] = pack_bf2(acc[2], acc[3]);
        *(u32x2*)((unsigned short*)out + r * ldc + c) = pk;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Skinny product for the decode step of generate() (reference app.py:308-317): M <= 4 activation rows against a whole weight matrix.
// Every weight byte is used once, so this is an HBM stream, not an MFMA problem (Qwen2.5-7B: 15.2 GB of weights per token; the tiled
// kernels above would pad M to 128-256 rows).  One wave owns CW output columns at a time: lane l reads the 16-byte chunks l, l+64, ...
// of each of its weight rows (coalesced 1 KiB per row and step, CW independent rows in flight) and of the activation rows (L1 / L2
// resident), accumulates in fp32 and wave-reduces.  Epilogue semantics equal gemm_epilogue's (bias, bf16 rounding before the
// activation, SwiGLU on the 16/16 interleaved gate/up row blocks, residual), so a decode step rounds exactly like a prefill row.
struct GemvArgs {
    const unsigned short* A;   // [M, K]
    const unsigned short* W;   // [N, K]
    void* C;
    const unsigned short* bias;
    const unsigned short* res;
    int M, N, K;               // N = weight rows (SwiGLU: 2 x outputs)
    long lda, ldw, ldc, ldr;
};

__device__ __forceinline__ float dot8(const u32x4& a, const u32x4& b, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        acc = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), acc);
        acc = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), acc);
    }
    return acc;
}

template <int MR, int ACT, bool OUT_F32, bool NT>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs p) {
    constexpr int CW = 4;                                   // output columns per wave and pass
    constexpr int RW = (ACT == ACT_SWIGLU) ? 2 * CW : CW;   // weight rows per wave and pass
    const int lane = threadIdx.x & 63;
    const int wave = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Nout = (ACT == ACT_SWIGLU) ? p.N / 2 : p.N;
    const int o0 = wave * CW;
    if (o0 >= Nout) return;
    // weight rows of this wave's outputs (SwiGLU packing: 32-row blocks = 16 gate rows, then the 16 up rows of the same outputs)
    const unsigned short* wrow[RW];
#pragma unroll
    for (int cidx = 0; cidx < CW; ++cidx) {
        const int o = min(o0 + cidx, Nout - 1);
        if constexpr (ACT == ACT_SWIGLU) {
            wrow[2 * cidx] = p.W + (long)((o >> 4) * 32 + (o & 15)) * p.ldw;
            wrow[2 * cidx + 1] = p.W + (long)((o >> 4) * 32 + 16 + (o & 15)) * p.ldw;
        } else {
            wrow[cidx] = p.W + (long)o * p.ldw;
        }
    }
    float acc[MR][RW];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[m][r] = 0.f;
    const int nch = p.K >> 3;
    for (int ch = lane; ch < nch; ch += 64) {
        u32x4 w[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if constexpr (NT) w[r] = __builtin_nontemporal_load((const u32x4*)(wrow[r] + ch * 8));   // read-once weight stream kept out of L2 / Infinity Cache
            else w[r] = *(const u32x4*)(wrow[r] + ch * 8);
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const u32x4 x = *(const u32x4*)(p.A + (long)min(m, p.M - 1) * p.lda + ch * 8);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[m][r] = dot8(w[r], x, acc[m][r]);
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[m][r] = wave_sum(acc[m][r]);
    if (lane >= CW * MR) return;
    const int cidx = lane % CW, m = lane / CW;
    const int o = o0 + cidx;
    if (o >= Nout || m >= p.M) return;
    float v = 0.f;
    // pick this lane's (m, column) out of the fully unrolled accumulator file
#pragma unroll
    for (int mm = 0; mm < MR; ++mm)
#pragma unroll
        for (int cc = 0; cc < CW; ++cc)
            if (mm == m && cc == cidx) {
                if constexpr (ACT == ACT_SWIGLU) {
                    float gt = acc[mm][2 * cc], up = acc[mm][2 * cc + 1];
                    if (p.bias) {
                        gt += bf2f(p.bias[(o >> 4) * 32 + (o & 15)]);
                        up += bf2f(p.bias[(o >> 4) * 32 + 16 + (o & 15)]);
                    }
                    gt = bf2f(f2bf(gt));
                    up = bf2f(f2bf(up));
                    v = bf2f(f2bf(silu_f(gt))) * up;
                } else {
                    float x = acc[mm][cc];
                    if (p.bias) x += bf2f(p.bias[o]);
                    if constexpr (ACT == ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
                    if constexpr (ACT == ACT_RELU) x = fmaxf(x, 0.f);
                    v = x;
                }
            }
    if constexpr (OUT_F32) {
        ((float*)p.C)[(long)m * p.ldc + o] = v;
    } else {
        if (p.res) v = bf2f(f2bf(v)) + bf2f(p.res[(long)m * p.ldr + o]);
        ((unsigned short*)p.C)[(long)m * p.ldc + o] = f2bf(v);
    }
}

template <int ACT, bool OUT_F32>
static int launch_gemv(const GemmArgs& a, hipStream_t st) {
    GemvArgs g;
    g.A = a.A; g.W = a.W; g.C = a.C; g.bias = a.bias; g.res = a.res;
    g.M = a.M; g.N = a.N; g.K = a.K; g.lda = a.lda; g.ldw = a.ldw; g.ldc = a.ldc; g.ldr = a.ldr;
    const int nout = (ACT == ACT_SWIGLU) ? a.N / 2 : a.N;
    const unsigned grid = (unsigned)cdiv(nout, 16);
#ifdef RGA3_AB   // measurement builds only (tools/): the product library has one behaviour
    static const bool nt = [] { const char* e = getenv("RGA3_GEMV_NT"); return e ? atoi(e) != 0 : true; }();   // A/B switch (default: nontemporal weight loads)
#else
    constexpr bool nt = true;
#endif
    if (nt) {
        if (a.M == 1) hipLaunchKernelGGL((gemv_kernel<1, ACT, OUT_F32, true>), dim3(grid), dim3(256), 0, st, g);
        else if (a.M == 2) hipLaunchKernelGGL((gemv_kernel<2, ACT, OUT_F32, true>), dim3(grid), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemv_kernel<4, ACT, OUT_F32, true>), dim3(grid), dim3(256), 0, st, g);
    } else {
        if (a.M == 1) hipLaunchKernelGGL((gemv_kernel<1, ACT, OUT_F32, false>), dim3(grid), dim3(256), 0, st, g);
        else if (a.M == 2) hipLaunchKernelGGL((gemv_kernel<2, ACT, OUT_F32, false>), dim3(grid), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemv_kernel<4, ACT, OUT_F32, false>), dim3(grid), dim3(256), 0, st, g);
    }
    RGA3_CHECK_LAUNCH("gemv_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Token-row products, 5 <= M <= 16 (tile 41): the token side of SAM2's two-way transformer (reference model/sam2.py:1926-2100: 9 output / prompt tokens per
// frame through q / k / v / out projections and a 256 -> 2048 -> 256 MLP, 22 products per frame).  The tiled kernels pad those rows to a 128-row tile and walk K
// inside ONE or two workgroups: 6 - 9 us per product, 24 us at K = 2048, all latency (profiles/r03_stream_frame_timeline_fused_tail.txt).  Here a workgroup owns 16
// output columns and its 8 waves split K between them (k-step ks goes to wave ks % 8): the M rows are the MFMA's B operand (tokens on the N side), the 16 weight rows
// the A operand, both read straight from L2 with every load of a wave in flight at once; the 8 partial tiles meet in LDS and wave 0 adds them in wave order (fixed
// order: reproducible) and runs gemm_epilogue's arithmetic (bias, bf16 rounding before GELU, ReLU, the linear output rounded before the residual is added).
constexpr int R16_NW = 8, R16_U = 8;
template <int ACT>
__global__ __launch_bounds__(64 * R16_NW) void gemm_rows16_kernel(GemmArgs p) {
    __shared__ f32x4 red[R16_NW][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const unsigned short* arow = p.W + (long)min(n0 + c, p.N - 1) * p.ldw + g * 8;
    const unsigned short* brow = p.A + (long)min(c, p.M - 1) * p.lda + g * 8;
    const int nks = (p.K + 31) >> 5;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks0 = w; ks0 < nks; ks0 += R16_NW * R16_U) {
        bf16x8 af[R16_U], bf[R16_U];
#pragma unroll
        for (int u = 0; u < R16_U; ++u) {
            const int k = (ks0 + u * R16_NW) * 32 + g * 8;
            u32x4 za = {0u, 0u, 0u, 0u}, zb = {0u, 0u, 0u, 0u};
            if (k < p.K) {          // K % 8 == 0: a 16-byte chunk lies inside K or outside, never across
                za = *(const u32x4*)(arow + (ks0 + u * R16_NW) * 32);
                zb = *(const u32x4*)(brow + (ks0 + u * R16_NW) * 32);
            }
            af[u] = __builtin_bit_cast(bf16x8, za);
            bf[u] = __builtin_bit_cast(bf16x8, zb);
        }
#pragma unroll
        for (int u = 0; u < R16_U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u], bf[u], acc, 0, 0, 0);
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w != 0) return;
    f32x4 v = red[0][lane];
#pragma unroll
    for (int i = 1; i < R16_NW; ++i) v += red[i][lane];
    // lane (c, g): row c, columns n0 + 4 g .. + 3
    const int col = n0 + 4 * g;
    if (c >= p.M || col >= p.N) return;
    unsigned short* dst = (unsigned short*)p.C + (long)c * p.ldc + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (col + r >= p.N) break;
        float x = v[r];
        if (p.bias) x += bf2f(p.bias[col + r]);
        if constexpr (ACT == ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
        if constexpr (ACT == ACT_RELU) x = fmaxf(x, 0.f);
        if (p.res) x = bf2f(f2bf(x)) + bf2f(p.res[(long)c * p.ldr + col + r]);
        dst[r] = f2bf(x);
    }
}

// Several token-row products in ONE launch, each optionally on the SUM of two row operands (rga3_gemm_rows16_many): the token side of a two-way block issues its
// q / k / v projections of (tokens + positional tokens) and of the tokens as 2 - 3 separate products behind an elementwise add each -- here grid.y picks the product
// and the add happens while the rows are loaded (rounded to bf16 as the add kernel rounds).
constexpr int R16_MAXSETS = 4;
struct Rows16Set {
    const unsigned short *A, *A2, *W, *bias, *res;
    unsigned short* C;
    int M, N, K, act;
    long lda, lda2, ldw, ldc, ldr;
};
struct Rows16Many { Rows16Set s[R16_MAXSETS]; };

__global__ __launch_bounds__(64 * R16_NW) void gemm_rows16_many_kernel(Rows16Many P) {
    __shared__ f32x4 red[R16_NW][64];
    const Rows16Set& p = P.s[blockIdx.y];
    const int n0 = blockIdx.x * 16;
    if (n0 >= p.N) return;                       // (uniform per workgroup: no barrier is skipped by part of it)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const unsigned short* arow = p.W + (long)min(n0 + c, p.N - 1) * p.ldw + g * 8;
    const unsigned short* brow = p.A + (long)min(c, p.M - 1) * p.lda + g * 8;
    const unsigned short* brow2 = p.A2 ? p.A2 + (long)min(c, p.M - 1) * p.lda2 + g * 8 : nullptr;
    const int nks = (p.K + 31) >> 5;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks0 = w; ks0 < nks; ks0 += R16_NW * R16_U) {
        bf16x8 af[R16_U], bf[R16_U];
#pragma unroll
        for (int u = 0; u < R16_U; ++u) {
            const int k = (ks0 + u * R16_NW) * 32 + g * 8;
            u32x4 za = {0u, 0u, 0u, 0u}, zb = {0u, 0u, 0u, 0u};
            if (k < p.K) {
                za = *(const u32x4*)(arow + (ks0 + u * R16_NW) * 32);
                zb = *(const u32x4*)(brow + (ks0 + u * R16_NW) * 32);
                if (brow2) {
                    const u32x4 z2 = *(const u32x4*)(brow2 + (ks0 + u * R16_NW) * 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        zb[e] = pack_bf2(__uint_as_float(zb[e] << 16) + __uint_as_float(z2[e] << 16), __uint_as_float(zb[e] & 0xffff0000u) + __uint_as_float(z2[e] & 0xffff0000u));
                }
            }
            af[u] = __builtin_bit_cast(bf16x8, za);
            bf[u] = __builtin_bit_cast(bf16x8, zb);
        }
#pragma unroll
        for (int u = 0; u < R16_U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u], bf[u], acc, 0, 0, 0);
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w != 0) return;
    f32x4 v = red[0][lane];
#pragma unroll
    for (int i = 1; i < R16_NW; ++i) v += red[i][lane];
    const int col = n0 + 4 * g;
    if (c >= p.M || col >= p.N) return;
    unsigned short* dst = p.C + (long)c * p.ldc + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (col + r >= p.N) break;
        float x = v[r];
        if (p.bias) x += bf2f(p.bias[col + r]);
        if (p.act == ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
        if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
        if (p.res) x = bf2f(f2bf(x)) + bf2f(p.res[(long)c * p.ldr + col + r]);
        dst[r] = f2bf(x);
    }
}

template <int ACT, bool OUT_F32>
static int launch_rows16(const GemmArgs& a, hipStream_t st) {
    if constexpr (OUT_F32 || ACT == ACT_SWIGLU) {
        return fail(-1, "gemm: tile 41 (token rows) writes bf16 and has no SwiGLU form");
    } else {
        hipLaunchKernelGGL(gemm_rows16_kernel<ACT>, dim3((unsigned)cdiv(a.N, 16)), dim3(64 * R16_NW), 0, st, a);
        RGA3_CHECK_LAUNCH("gemm_rows16_kernel");
        return 0;
    }
}

// tile choice: fill the 256 CUs.  score = useful fraction of the last wave of tiles x a per-config prior.
static int pick_tile(int M, int N, int K, bool plain, int forced) {
    if (forced >= 0) return forced;
    // a handful of output tiles over a very long K (per-frame mask products, weight gradients): split K over the CUs
    if (plain && K >= 8192 && cdiv(M, 256) * cdiv(N, 256) <= 4 && K % 64 == 0) return 25;
    struct Cfg { int bm, bn, slots; double prior; };
    // slots = resident workgroups per CU (LDS-limited)
    static const Cfg cfgs[3] = {{256, 256, 1, 1.00}, {256, 128, 1, 0.90}, {128, 128, 2, 0.78}};
    int best = 0;
    double bs = -1;
    for (int i = 0; i < 3; ++i) {
        double tiles = (double)cdiv(M, cfgs[i].bm) * (double)cdiv(N, cfgs[i].bn);
        double cap = 256.0 * cfgs[i].slots;
        double waves = (double)cdiv((int64_t)tiles, (int64_t)cap);
        double fill = tiles / (waves * cap);
        double useful = ((double)M * N) / (tiles * cfgs[i].bm * cfgs[i].bn);
        double s = fill * useful * cfgs[i].prior;
        if (s > bs) { bs = s; best = i; }
    }
    static const int ids[3] = {20, 11, 12};  // ping-pong 256x256, single-phase 256x128 / 128x128
    return ids[best];
}

template <int ACT, bool OUT_F32>
static int launch_act(const GemmArgs& a, int tile, hipStream_t st) {
    switch (tile) {
        case 3: return launch_cfg<128, 256, 2, 4, ACT, OUT_F32, 0>(a, st);
        case 4:  // 128x320: N = 1280 (ViT proj / fc2) at M = 8192 is exactly 256 tiles; 5 n-tiles per wave, so no SwiGLU pairs
            if constexpr (ACT != ACT_SWIGLU) return launch_cfg<128, 320, 2, 4, ACT, OUT_F32, 0>(a, st);
            else return launch_cfg<128, 256, 2, 4, ACT, OUT_F32, 0>(a, st);
        case 5:   // 128x192: N = 576 = 3 x 192 (Hiera stage-3 proj / fc2 outputs: 2.25 tiles of 256 otherwise); 3 n-tiles per wave, so no SwiGLU pairs
            if constexpr (ACT != ACT_SWIGLU) return launch_cfg<128, 192, 2, 4, ACT, OUT_F32, 0>(a, st);
            else return launch_cfg<128, 256, 2, 4, ACT, OUT_F32, 0>(a, st);
        case 6: return launch_cfg<128, 256, 2, 4, ACT, OUT_F32, 3>(a, st);   // 128x256, three LDS stages (144 KiB): two K-tiles in flight (cold weights)
        case 7:   // 128x192, three stages (120 KiB, one workgroup per CU)
            if constexpr (ACT != ACT_SWIGLU) return launch_cfg<128, 192, 2, 4, ACT, OUT_F32, 3>(a, st);
            else return launch_cfg<128, 256, 2, 4, ACT, OUT_F32, 3>(a, st);
        case 8: return launch_cfg<128, 128, 2, 2, ACT, OUT_F32, 3>(a, st);   // 128x128 on 4 waves, three stages (96 KiB)
        case 10: return launch_cfg<256, 256, 2, 4, ACT, OUT_F32, 0>(a, st);
        case 13: return launch_cfg<64, 64, 2, 2, ACT, OUT_F32, 0>(a, st);   // small products (SAM2 per-frame 4096 x 256 x 256: 256 tiles instead of 64)
        case 14:   // 64 x 64 with a K split (skinny plain products)
            if constexpr (ACT == ACT_NONE) return launch_split64<OUT_F32>(a, st);
            else return launch_cfg<64, 64, 2, 2, ACT, OUT_F32, 0>(a, st);
        case 20: return launch_pp<ACT, OUT_F32>(a, st);
        case 23:   // ping-pong on 256 x 192 tiles (4 x 2 wave grid, 6 n-tiles per wave: SwiGLU pairs exist, but its packed widths are multiples of 256 anyway)
            if constexpr (ACT == ACT_NONE) return launch_pp<ACT, OUT_F32, false, true>(a, st);
            else return launch_pp<ACT, OUT_F32>(a, st);
        case 21: return launch_sk<ACT, OUT_F32>(a, false, st);
        case 22: return launch_sk<ACT, OUT_F32>(a, true, st);
        case 26: return launch_sk<ACT, OUT_F32>(a, true, st, true);    // 22 / 21 whose ragged last tile row (<= 64 rows: M = 2112 = 8 x 256 + 64) runs the quarter-work loop
        case 27: return launch_sk<ACT, OUT_F32>(a, false, st, true);   //   on workgroups of its own
        case 31: return launch_sk<ACT, OUT_F32, 3>(a, false, st);   // 192 x 256 tiles, persistent
        case 32: return launch_sk<ACT, OUT_F32, 3>(a, true, st);    // ... + stream-K tail
        case 25: return launch_splitk<ACT, OUT_F32>(a, st);
        case 28: return launch_w4<ACT, OUT_F32>(a, st);   // 256 x 256 on four waves (128 x 128 wave blocks, AGPR accumulators), one tile per workgroup
        case 40: return launch_gemv<ACT, OUT_F32>(a, st);   // M <= 4: weight stream (decode step)
        case 41: return launch_rows16<ACT, OUT_F32>(a, st);  // 5 <= M <= 16 token rows
        default: return launch_cfg<128, 128, 2, 2, ACT, OUT_F32, 0>(a, st);
    }
}

}  // namespace rga3

using namespace rga3;

// n (<= 4) token-row products (M <= 16 rows each) in one launch: C_i [M_i, N_i] bf16 = act_i((A_i (+ A2_i)) W_i^T + bias_i) (+ residual_i).  ptrs: n x 6 pointers
// {A, A2 or NULL, W, bias or NULL, residual or NULL, C}; dims: n x 9 {M, N, K, act, lda, lda2, ldw, ldc, ldr} (elements; K and the A / W strides multiples of 8).
// A + A2 is rounded to bf16 before the product (what a separate add launch would have written).  HOST arrays.
extern "C" int rga3_gemm_rows16_many(const void* const* ptrs, const int64_t* dims, int n, void* stream) {
    RGA3_CHECK_ARG(ptrs && dims && n >= 1 && n <= R16_MAXSETS, "gemm_rows16_many: n %d (1..%d)", n, R16_MAXSETS);
    Rows16Many P;
    int64_t maxn = 0;
    for (int i = 0; i < n; ++i) {
        Rows16Set& s = P.s[i];
        const void* const* q = ptrs + 6 * i;
        const int64_t* d = dims + 9 * i;
        s.A = (const unsigned short*)q[0]; s.A2 = (const unsigned short*)q[1]; s.W = (const unsigned short*)q[2]; s.bias = (const unsigned short*)q[3];
        s.res = (const unsigned short*)q[4]; s.C = (unsigned short*)q[5];
        s.M = (int)d[0]; s.N = (int)d[1]; s.K = (int)d[2]; s.act = (int)d[3]; s.lda = d[4]; s.lda2 = d[5]; s.ldw = d[6]; s.ldc = d[7]; s.ldr = d[8];
        RGA3_CHECK_ARG(s.A && s.W && s.C && s.M >= 1 && s.M <= 16 && s.N >= 1 && s.K >= 8 && s.K % 8 == 0, "gemm_rows16_many: set %d: M %d (1..16), N %d, K %d", i, s.M, s.N, s.K);
        RGA3_CHECK_ARG(s.lda % 8 == 0 && s.ldw % 8 == 0 && (!s.A2 || s.lda2 % 8 == 0), "gemm_rows16_many: set %d: strides must be multiples of 8", i);
        RGA3_CHECK_ARG((((uintptr_t)s.A | (uintptr_t)s.A2 | (uintptr_t)s.W) & 15) == 0, "gemm_rows16_many: set %d: 16-byte alignment", i);
        RGA3_CHECK_ARG(s.act == ACT_NONE || s.act == ACT_GELU || s.act == ACT_RELU, "gemm_rows16_many: set %d: act %d", i, s.act);
        RGA3_CHECK_ARG(s.ldc >= s.N && (!s.res || s.ldr >= s.N), "gemm_rows16_many: set %d: ldc %ld / ldr %ld shorter than N %d (a residual row stride below N reads out of bounds)",
                       i, (long)s.ldc, (long)s.ldr, s.N);
        if (s.N > maxn) maxn = s.N;
    }
    hipLaunchKernelGGL(gemm_rows16_many_kernel, dim3((unsigned)cdiv(maxn, 16), (unsigned)n), dim3(64 * R16_NW), 0, (hipStream_t)stream, P);
    RGA3_CHECK_LAUNCH("gemm_rows16_many_kernel");
    return 0;
}

// Bytes of workspace rga3_gemm_bf16 wants on the current device for tiles 22 / 25 (0 on error).
extern "C" int64_t rga3_gemm_workspace_bytes(void) {
    const int cus = cu_count();
    return cus > 0 ? (int64_t)(kSkFlagBytes + (size_t)cus * 512 * 32 * 16) : 0;
}

// Diagnostic: number of bounded spins that gave up in stream-K launches that used this workspace (expected 0).  Synchronises the device.
extern "C" int rga3_gemm_stream_k_timeouts(const void* workspace) {
    SkWorkspace ws;
    if (sk_workspace((void*)workspace, (int64_t)1 << 40, ws)) return -1;
    unsigned v = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, ws.flags + ws.P, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}

// Host only (no device needed): the work plan tiles 26 (split = 1) / 27 (split = 0) launch for an [M, N, K] product on `cus` compute units.  plan[8] = {first tile of the
// stream-K tail, tiles of the tail, 1, workgroups that take ragged tiles only, 0, workgroups launched, workgroups with a run, tile rows per group}; start[0 .. cus]:
// workgroup w takes K-iterations [start[w], start[w + 1]) of the tail line.  Returns 0; 1 when the product has no ragged last tile row or no plan applies (the tiles
// then run as 22 / 21); < 0 on bad arguments.  Lets CPU tests check what the kernel's owner / contributor hand-off relies on.
extern "C" int rga3_gemm_ragged_plan(int64_t M, int64_t N, int64_t K, int cus, int split, int* plan, unsigned* start) {
    RGA3_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 64 == 0 && cus > 0 && cus <= kSkMaxP && plan && start, "gemm_ragged_plan: M, N > 0, K a multiple of 64, 0 < cus <= %d", kSkMaxP);
    GemmArgs a{};
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.ntm = (int)cdiv(M, 256);
    a.ntn = (int)cdiv(N, 256);
    for (int i = 0; i < 8; ++i) plan[i] = 0;
    if (a.ntm < 2 || M - (int64_t)(a.ntm - 1) * 256 > 64) return 1;
    a.group_m = pick_group_m(a.ntm - 1, 256);
    SkArgs sk;
    if (!sk_plan_ragged(sk, a, (int)(K / 64), cus, split != 0)) return 1;
    plan[0] = sk.t_dp; plan[1] = sk.sk_tiles; plan[2] = sk.ragged; plan[3] = sk.rg_wgs; plan[4] = 0; plan[5] = sk.P; plan[6] = sk.P_sk; plan[7] = a.group_m;
    for (int w = 0; w <= cus; ++w) start[w] = sk.start[w];
    return 0;
}

// Byte offset, inside a workspace of this device, of the 32-bit give-up counter rga3_gemm_stream_k_timeouts reads -- so a training loop can fetch it with its
// own asynchronous copy instead of a device synchronisation (rga3.hip.ops.GemmHealthWatch).  < 0 on error.
extern "C" int64_t rga3_gemm_timeout_counter_offset(void) {
    const int cus = cu_count();
    if (cus <= 0 || cus * 4 + 16 > (int)kSkFlagBytes) return -1;
    return (int64_t)cus * 4;
}

static int gemm_bf16_impl(const void* A, const void* W, const void* bias, const void* residual, const void* colscale, void* C,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr,
                          int act, int out_dtype, int tile, void* workspace, int64_t workspace_bytes, const unsigned long long* rs_in, int64_t rs_width, float rs_eps,
                          unsigned long long* rs_out, void* stream, void* pre = nullptr, int64_t ldpre = 0);

extern "C" int rga3_gemm_bf16(const void* A, const void* W, const void* bias, const void* residual, const void* colscale, void* C,
                              int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr,
                              int act, int out_dtype, int tile, void* workspace, int64_t workspace_bytes, void* stream) {
    return gemm_bf16_impl(A, W, bias, residual, colscale, C, M, N, K, lda, ldw, ldc, ldr, act, out_dtype, tile, workspace, workspace_bytes, nullptr, 0, 0.f, nullptr, stream);
}

// rga3_gemm_bf16 with an RMSNorm folded around it (HF Qwen2RMSNorm, modeling_qwen2_5_vl.py:470-486 / Qwen2_5_VLRMSNorm; the norm -> projection pairs of
// :602-757 and :211-321).  Consumer side (row_sumsq_in): A holds the UN-normalised rows x, W the weight with the norm weight folded in (W' = W diag(gamma)), and
//   RMSNorm(x) W^T = rinv_r (x W'^T),  rinv_r = 1 / sqrt(row_sumsq_in[r] / 2^20 / norm_width + eps)
// is applied to the accumulators before bias / activation.  Producer side (row_sumsq_out): the sums of squares of the bf16 output rows are ADDED to
// row_sumsq_out[r] as fixed-point integers (the caller zeroes it; integer atomics: any arrival order gives the same bits).  bf16 output only; tiles whose
// epilogue is the shared one (not 14 / 25 / 40 / 41).
extern "C" int rga3_gemm_rms_bf16(const void* A, const void* W, const void* bias, const void* residual, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                  int64_t ldw, int64_t ldc, int64_t ldr, int act, int tile, void* workspace, int64_t workspace_bytes, const uint64_t* row_sumsq_in,
                                  int64_t norm_width, float eps, uint64_t* row_sumsq_out, void* stream) {
    RGA3_CHECK_ARG(row_sumsq_in || row_sumsq_out, "gemm_rms: neither row_sumsq_in nor row_sumsq_out given (use rga3_gemm_bf16)");
    RGA3_CHECK_ARG(!row_sumsq_in || (norm_width > 0 && eps >= 0.f), "gemm_rms: norm_width %ld eps %g", (long)norm_width, (double)eps);
    RGA3_CHECK_ARG((((uintptr_t)row_sumsq_in | (uintptr_t)row_sumsq_out) & 7) == 0, "gemm_rms: row sums must be 8-byte aligned");
    RGA3_CHECK_ARG(tile != 14 && tile != 25 && tile != 40 && tile != 41, "gemm_rms: tile %d has its own epilogue", tile);
    RGA3_CHECK_ARG(M > 16, "gemm_rms: M %ld (token-row products take the unfused route)", (long)M);
    return gemm_bf16_impl(A, W, bias, residual, nullptr, C, M, N, K, lda, ldw, ldc, ldr, act, RGA3_BF16, tile, workspace, workspace_bytes,
                          (const unsigned long long*)row_sumsq_in, norm_width, eps, (unsigned long long*)row_sumsq_out, stream);
}

// rga3_gemm_bf16 whose epilogue also leaves the LayerNorm statistics of the rows it writes (Hiera MultiScaleBlock: x = shortcut + proj(attn) feeds norm2, x = x + mlp(..)
// feeds the next block's norm1; reference sam2.py:1085-1117): row_parts [M][slices][2] f32 with slices = rga3_gemm_lnsum_slices(N, tile) -- for row r and tile column
// t, (sum, sum of squares) of the bf16 values this launch wrote to columns of that tile.  Plain stores, every element written once: nothing to zero, bitwise
// reproducible.  bf16 output, no activation (proj / fc2 have none), tiles 3 / 5 / 12 / 13 / 20 / 23 (kernels of their own: LNM = 2).  Consumer: rga3_gemm_lnq_bf16.
static int lnsum_tile_width(int tile) { return (tile == 3 || tile == 20) ? 256 : (tile == 5 || tile == 23) ? 192 : (tile == 13) ? 64 : 128; }
extern "C" int64_t rga3_gemm_lnsum_slices(int64_t N, int tile) {
    if (N <= 0 || !(tile == -1 || tile == 3 || tile == 5 || tile == 12 || tile == 13 || tile == 20 || tile == 23)) return -1;
    return cdiv(N, lnsum_tile_width(tile));
}
extern "C" int rga3_gemm_lnsum_bf16(const void* A, const void* W, const void* bias, const void* residual, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                    int64_t ldw, int64_t ldc, int64_t ldr, int tile, float* row_parts, void* stream) {
    RGA3_CHECK_ARG(A && W && C && row_parts && (((uintptr_t)row_parts) & 7) == 0, "gemm_lnsum: null pointer / row_parts must be 8-byte aligned");
    RGA3_CHECK_ARG(M > 16 && N > 0 && K > 0 && K % 8 == 0, "gemm_lnsum: bad shape M=%ld N=%ld K=%ld (M > 16, K %% 8)", (long)M, (long)N, (long)K);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "gemm_lnsum: lda/ldw must be multiples of 8 elements (16-byte rows)");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0, "gemm_lnsum: pointers must be 16-byte aligned");
    RGA3_CHECK_ARG(tile == -1 || tile == 3 || tile == 5 || tile == 12 || tile == 13 || tile == 20 || tile == 23, "gemm_lnsum: tile %d (3 / 5 / 12 / 13 / 20 / 23)", tile);
    RGA3_CHECK_ARG(M * lda < (1LL << 32) && N * ldw < (1LL << 32), "gemm_lnsum: operands must be < 2^32 elements (32-bit staging offsets)");
    GemmArgs a;
    a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.C = C;
    a.bias = (const unsigned short*)bias; a.res = (const unsigned short*)residual; a.colscale = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.ws = nullptr; a.ws_bytes = 0; a.ksl = 0; a.rowstat = nullptr; a.colc = nullptr;
    a.ln_parts = row_parts;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 3: return launch_cfg<128, 256, 2, 4, ACT_NONE, false, 0, 2>(a, st);
        case 5: return launch_cfg<128, 192, 2, 4, ACT_NONE, false, 0, 2>(a, st);
        case 13: return launch_cfg<64, 64, 2, 2, ACT_NONE, false, 0, 2>(a, st);
        case 20: return launch_pp<ACT_NONE, false, 2, false>(a, st);
        case 23: return launch_pp<ACT_NONE, false, 2, true>(a, st);
        default: return launch_cfg<128, 128, 2, 2, ACT_NONE, false, 0, 2>(a, st);
    }
}

// SwiGLU product that ALSO stores its bf16 pre-activations (training forward of the decoder MLP: the backward of silu(gate) * up needs gate and up; HF Qwen2MLP,
// autograd under reference train_joint.py:534): C [M, N / 2] = silu(gate) * up as rga3_gemm_bf16 with RGA3_ACT_SWIGLU, pre [M, N] = the interleaved gate | up values
// rounded to bf16 (what a plain rga3_gemm_bf16 on the same packed weight writes) -- the stand-alone SwiGLU launch (read 2 x, write 1 x the widest activation) disappears.
extern "C" int rga3_gemm_swiglu_pre_bf16(const void* A, const void* W, const void* bias, void* C, void* pre, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                         int64_t ldc, int64_t ldpre, int tile, void* workspace, int64_t workspace_bytes, void* stream) {
    RGA3_CHECK_ARG(pre, "gemm_swiglu_pre: null pre-activation output");
    return gemm_bf16_impl(A, W, bias, nullptr, nullptr, C, M, N, K, lda, ldw, ldc, 0, ACT_SWIGLU, RGA3_BF16, tile, workspace, workspace_bytes, nullptr, 0, 0.f, nullptr, stream, pre,
                          ldpre);
}

// Concatenated operands on the single-phase kernels (LoRA's low-rank products folded into the frozen products of a decoder layer; PEFT LoRA layer under reference
// train_joint.py:193-232):
//   K side (A2, W2, K2 != 0):  C [M, N] = [A | A2] . [W | W2]^T (+ bias): y = x W^T + t B^T as ONE product over K + K2 (K, K2 multiples of 64)
//   N side (Wn, Cn, N2 != 0):  additionally Cn [M, N2] = A . Wn^T from the same launch (tile columns behind N; N must be a multiple of the tile width): [dx | dt] = dy [W | sB]
// bf16 in / out, no activation, optional bias on the first pair.  tile: -1 / 12 (128 x 128), 3 (128 x 256), 6 (128 x 256, three stages), 13 (64 x 64); K side only: also 4, 5.
extern "C" int rga3_gemm_cat_bf16(const void* A, const void* W, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
                                  const void* A2, const void* W2, int64_t K2, int64_t lda2, int64_t ldw2, const void* Wn, void* Cn, int64_t N2, int64_t ldwn,
                                  int64_t ldcn, int tile, void* stream) {
    RGA3_CHECK_ARG(A && W && C && M > 16 && N > 0 && K > 0, "gemm_cat: bad args");
    RGA3_CHECK_ARG((A2 != nullptr) == (W2 != nullptr) && (A2 != nullptr) == (K2 > 0), "gemm_cat: the K side takes A2, W2 and K2 together");
    RGA3_CHECK_ARG((Wn != nullptr) == (Cn != nullptr) && (Wn != nullptr) == (N2 > 0), "gemm_cat: the N side takes Wn, Cn and N2 together");
    RGA3_CHECK_ARG(A2 || Wn, "gemm_cat: nothing concatenated (use rga3_gemm_bf16)");
    RGA3_CHECK_ARG(K % 64 == 0 && K2 % 64 == 0, "gemm_cat: K = %ld and K2 = %ld must be multiples of 64", (long)K, (long)K2);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda2 % 8 == 0 && ldw2 % 8 == 0 && ldwn % 8 == 0 && N % 8 == 0 && N2 % 8 == 0, "gemm_cat: strides / widths must be multiples of 8 elements");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)A2 | (uintptr_t)W2 | (uintptr_t)Wn | (uintptr_t)Cn) & 15) == 0, "gemm_cat: pointer alignment");
    RGA3_CHECK_ARG(M * lda < (1LL << 32) && N * ldw < (1LL << 32) && M * (lda2 ? lda2 : 1) < (1LL << 32) && N * (ldw2 ? ldw2 : 1) < (1LL << 32) && N2 * (ldwn ? ldwn : 1) < (1LL << 32),
                   "gemm_cat: operands must be < 2^32 elements");
    const int bn = (tile == 3 || tile == 6) ? 256 : (tile == 4) ? 320 : (tile == 5) ? 192 : (tile == 13) ? 64 : 128;
    RGA3_CHECK_ARG(tile == -1 || tile == 3 || tile == 6 || tile == 12 || tile == 13 || (!Wn && (tile == 4 || tile == 5)), "gemm_cat: tile %d", tile);
    RGA3_CHECK_ARG(!Wn || N % bn == 0, "gemm_cat: with an N side, N = %ld must be a multiple of the tile width %d", (long)N, bn);
    GemmArgs a;
    a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.C = C;
    a.bias = (const unsigned short*)bias; a.res = nullptr; a.colscale = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = 0;
    a.ws = nullptr; a.ws_bytes = 0; a.ksl = 0; a.rowstat = nullptr; a.colc = nullptr;
    a.A2 = (const unsigned short*)A2; a.W2 = (const unsigned short*)W2; a.K2 = (int)K2; a.lda2 = lda2; a.ldw2 = ldw2;
    a.Wn = (const unsigned short*)Wn; a.Cn = Cn; a.N2 = (int)N2; a.ldwn = ldwn; a.ldcn = ldcn;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 3: return launch_cfg<128, 256, 2, 4, ACT_NONE, false, 0>(a, st);
        case 4: return launch_cfg<128, 320, 2, 4, ACT_NONE, false, 0>(a, st);
        case 5: return launch_cfg<128, 192, 2, 4, ACT_NONE, false, 0>(a, st);
        case 6: return launch_cfg<128, 256, 2, 4, ACT_NONE, false, 3>(a, st);
        case 13: return launch_cfg<64, 64, 2, 2, ACT_NONE, false, 0>(a, st);
        default: return launch_cfg<128, 128, 2, 2, ACT_NONE, false, 0>(a, st);
    }
}

static int gemm_bf16_impl(const void* A, const void* W, const void* bias, const void* residual, const void* colscale, void* C,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr,
                          int act, int out_dtype, int tile, void* workspace, int64_t workspace_bytes, const unsigned long long* rs_in, int64_t rs_width, float rs_eps,
                          unsigned long long* rs_out, void* stream, void* pre, int64_t ldpre) {
    RGA3_CHECK_ARG(A && W && C, "gemm: null pointer");
    RGA3_CHECK_ARG(!pre || (act == ACT_SWIGLU && M > 4 && tile != 40 && tile != 41 && (((uintptr_t)pre) & 15) == 0 && ldpre >= N), "gemm: the pre-activation output goes with the SwiGLU tile epilogues (M > 4)");
    RGA3_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    RGA3_CHECK_ARG(K % 8 == 0, "gemm: K=%ld must be a multiple of 8 (16-byte staging chunks)", (long)K);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements (16-byte rows)");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0, "gemm: pointers must be 16-byte aligned");
    RGA3_CHECK_ARG(out_dtype == RGA3_BF16 || out_dtype == RGA3_F32, "gemm: out_dtype %d", out_dtype);
    RGA3_CHECK_ARG(act >= 0 && act <= 3, "gemm: act %d", act);
    RGA3_CHECK_ARG(!(out_dtype == RGA3_F32 && (act != ACT_NONE || residual || colscale)), "gemm: f32 output supports bias only");
    RGA3_CHECK_ARG(act != ACT_SWIGLU || N % 32 == 0, "gemm: swiglu needs N %% 32 == 0");
    RGA3_CHECK_ARG(tile == -1 || (tile >= 3 && tile <= 8) || (tile >= 10 && tile <= 14) || (tile >= 20 && tile <= 23) || tile == 25 || (tile >= 26 && tile <= 28) || tile == 31 || tile == 32 || tile == 40 || tile == 41, "gemm: tile %d", tile);
    RGA3_CHECK_ARG(M * lda < (1LL << 32) && N * ldw < (1LL << 32), "gemm: operands must be < 2^32 elements (32-bit staging offsets)");
    GemmArgs a;
    a.A = (const unsigned short*)A;
    a.W = (const unsigned short*)W;
    a.C = C;
    a.bias = (const unsigned short*)bias;
    a.res = (const unsigned short*)residual;
    a.colscale = (const unsigned short*)colscale;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.ws = workspace; a.ws_bytes = workspace_bytes; a.ksl = 0; a.rowstat = nullptr; a.colc = nullptr;
    a.rs_in = rs_in; a.rs_in_scale = rs_in ? 1.0f / (kRowSumFix * (float)rs_width) : 0.f; a.rs_eps = rs_eps; a.rs_out = rs_out;
    a.pre = (unsigned short*)pre; a.ldpre = ldpre;
    hipStream_t st = (hipStream_t)stream;
    RGA3_CHECK_ARG(tile != 40 || (M <= 4 && !colscale), "gemm: the skinny kernel (tile 40) takes M <= 4 rows and no column scale");
    const bool rows16_ok = M <= 16 && !colscale && out_dtype == RGA3_BF16 && act != ACT_SWIGLU;
    RGA3_CHECK_ARG(tile != 41 || rows16_ok, "gemm: the token-row kernel (tile 41) takes M <= 16 rows, bf16 output, no column scale, no SwiGLU");
    int tl = (tile == -1 && M <= 4 && !colscale) ? 40 : (tile == -1 && rows16_ok) ? 41 : pick_tile((int)M, (int)N, (int)K, act == ACT_NONE && !residual && !colscale && !rs_in && !rs_out, tile);   // row sums live in the shared epilogue: never the split-K tile
    if (out_dtype == RGA3_F32) return launch_act<ACT_NONE, true>(a, tl, st);
    switch (act) {
        case ACT_NONE: return launch_act<ACT_NONE, false>(a, tl, st);
        case ACT_GELU: return launch_act<ACT_GELU, false>(a, tl, st);
        case ACT_SWIGLU: return launch_act<ACT_SWIGLU, false>(a, tl, st);
        default: return launch_act<ACT_RELU, false>(a, tl, st);
    }
}

template <int ACT>
static int launch_ln(const GemmArgs& a, int tile, hipStream_t st) {
    switch (tile) {
        case 3: return launch_cfg<128, 256, 2, 4, ACT, false, 0, true>(a, st);
        case 5: return launch_cfg<128, 192, 2, 4, ACT, false, 0, true>(a, st);
        case 6: return launch_cfg<128, 256, 2, 4, ACT, false, 3, true>(a, st);
        case 7: return launch_cfg<128, 192, 2, 4, ACT, false, 3, true>(a, st);
        case 13: return launch_cfg<64, 64, 2, 2, ACT, false, 0, true>(a, st);
        case 20: return launch_pp<ACT, false, true>(a, st);
        default: return launch_cfg<128, 128, 2, 2, ACT, false, 0, true>(a, st);
    }
}

// C[M, N] (bf16) = act(LayerNorm(A) . W^T + b) with the LayerNorm folded into the product (see gemm_epilogue, LNF): A [M, K] un-normalised rows, Wf [N, K] the
// weight with gamma folded in, colc [N] f32 column sums of Wf, bias [N] bf16 = beta . W^T + b, rowstat [M][2] f32 from rga3_layernorm_stats.
// act: none / gelu / relu.  tile: -1 (128 x 128) or 3 / 5 / 12 / 13 / 20.
extern "C" int rga3_gemm_ln_bf16(const void* A, const void* Wf, const void* bias, const float* colc, const float* rowstat, void* C, int64_t M, int64_t N, int64_t K,
                                 int64_t lda, int64_t ldw, int64_t ldc, int act, int tile, void* stream) {
    RGA3_CHECK_ARG(A && Wf && C && colc && rowstat, "gemm_ln: null pointer");
    RGA3_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 4 == 0, "gemm_ln: bad shape M=%ld N=%ld K=%ld (K %% 8, N %% 4)", (long)M, (long)N, (long)K);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "gemm_ln: lda/ldw must be multiples of 8 elements");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)Wf | (uintptr_t)C | (uintptr_t)colc) & 15) == 0 && (((uintptr_t)rowstat) & 7) == 0, "gemm_ln: pointer alignment");
    RGA3_CHECK_ARG(act == ACT_NONE || act == ACT_GELU || act == ACT_RELU, "gemm_ln: act %d", act);
    RGA3_CHECK_ARG(tile == -1 || tile == 3 || (tile >= 5 && tile <= 7) || tile == 12 || tile == 13 || tile == 20, "gemm_ln: tile %d", tile);
    RGA3_CHECK_ARG(M * lda < (1LL << 32) && N * ldw < (1LL << 32), "gemm_ln: operands must be < 2^32 elements");
    GemmArgs a;
    a.A = (const unsigned short*)A; a.W = (const unsigned short*)Wf; a.C = C;
    a.bias = (const unsigned short*)bias; a.res = nullptr; a.colscale = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = 0;
    a.ws = nullptr; a.ws_bytes = 0; a.ksl = 0; a.rowstat = rowstat; a.colc = colc;
    hipStream_t st = (hipStream_t)stream;
    if (act == ACT_GELU) return launch_ln<ACT_GELU>(a, tile, st);
    if (act == ACT_RELU) return launch_ln<ACT_RELU>(a, tile, st);
    return launch_ln<ACT_NONE>(a, tile, st);
}

// rga3_gemm_ln_bf16 reading the statistics as the producer's partial sums (rga3_gemm_lnsum_bf16) instead of (mean, 1 / std): row_parts [M][slices][2] f32, norm_width
// = the width the sums were taken over (= K here), eps the LayerNorm's.  mean = S1 / w, var = S2 / w - mean^2 (f64 inside the kernel), biased variance as nn.LayerNorm.
extern "C" int rga3_gemm_lnq_bf16(const void* A, const void* Wf, const void* bias, const float* colc, const float* row_parts, int64_t slices, int64_t norm_width,
                                  float eps, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int act, int tile, void* stream) {
    RGA3_CHECK_ARG(A && Wf && C && colc && row_parts, "gemm_lnq: null pointer");
    RGA3_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 4 == 0 && norm_width > 0 && eps >= 0.f && slices >= 1 && slices <= 4096,
                   "gemm_lnq: bad shape M=%ld N=%ld K=%ld slices=%ld (K %% 8, N %% 4)", (long)M, (long)N, (long)K, (long)slices);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "gemm_lnq: lda/ldw must be multiples of 8 elements");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)Wf | (uintptr_t)C | (uintptr_t)colc) & 15) == 0 && (((uintptr_t)row_parts) & 7) == 0, "gemm_lnq: pointer alignment");
    RGA3_CHECK_ARG(act == ACT_NONE || act == ACT_GELU || act == ACT_RELU, "gemm_lnq: act %d", act);
    RGA3_CHECK_ARG(tile == -1 || tile == 3 || (tile >= 5 && tile <= 7) || tile == 12 || tile == 13 || tile == 20, "gemm_lnq: tile %d", tile);
    RGA3_CHECK_ARG(M * lda < (1LL << 32) && N * ldw < (1LL << 32), "gemm_lnq: operands must be < 2^32 elements");
    GemmArgs a;
    a.A = (const unsigned short*)A; a.W = (const unsigned short*)Wf; a.C = C;
    a.bias = (const unsigned short*)bias; a.res = nullptr; a.colscale = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = 0;
    a.ws = nullptr; a.ws_bytes = 0; a.ksl = 0; a.rowstat = nullptr; a.colc = colc;
    a.ln_in = row_parts; a.ln_ns = (int)slices; a.rs_in_scale = 1.0f / (float)norm_width; a.rs_eps = eps;
    hipStream_t st = (hipStream_t)stream;
    if (act == ACT_GELU) return launch_ln<ACT_GELU>(a, tile, st);
    if (act == ACT_RELU) return launch_ln<ACT_RELU>(a, tile, st);
    return launch_ln<ACT_NONE>(a, tile, st);
}

// C[M, N] (bf16 or f32) = A^T . B (+ bias[n]) with A [K, M], B [K, N] bf16 row-major: the weight-gradient product dW = dY^T X without
// transposing either operand first.  M, N multiples of 8; lda / ldb / ldc in elements.
extern "C" int rga3_gemm_tn_bf16(const void* A, const void* B, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                 int64_t ldb, int64_t ldc, int out_dtype, void* workspace, int64_t workspace_bytes, void* counters, void* stream) {
    RGA3_CHECK_ARG(A && B && C, "gemm_tn: null pointer");
    RGA3_CHECK_ARG(M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0, "gemm_tn: bad shape M=%ld N=%ld K=%ld (M, N multiples of 8)", (long)M, (long)N, (long)K);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm_tn: lda/ldb must be multiples of 8 elements, ldc of 4");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)workspace) & 15) == 0, "gemm_tn: pointers must be 16-byte aligned");
    RGA3_CHECK_ARG(out_dtype == RGA3_BF16 || out_dtype == RGA3_F32, "gemm_tn: out_dtype %d", out_dtype);
    RGA3_CHECK_ARG(cdiv(M, 128) <= 65535, "gemm_tn: M too large");
    GemmArgs a;
    a.A = nullptr; a.W = nullptr; a.C = C;
    a.bias = (const unsigned short*)bias; a.res = nullptr; a.colscale = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = 0; a.ldw = 0; a.ldc = ldc; a.ldr = 0;
    a.ntm = (int)cdiv(M, 128); a.ntn = (int)cdiv(N, 128); a.group_m = 1;
    a.ws = nullptr; a.ws_bytes = 0; a.ksl = 0; a.rowstat = nullptr; a.colc = nullptr;
    TnArgs t;
    t.A = (const unsigned short*)A; t.B = (const unsigned short*)B; t.lda = lda; t.ldb = ldb; t.K = (int)K;
    t.counters = nullptr; t.out = nullptr; t.ldo = 0; t.out_f32 = 0;
    const int nk = (int)cdiv(K, 32);
    // few output tiles over many tokens (LoRA dW: 1 x 28 tiles, K = 2112 / 4160; mask-path dW: 1-4 tiles, K = 65 536 ..): cut K into Z slices, one workgroup each, f32 partial slabs in
    // the caller's workspace, summed in fixed order by a second launch (deterministic; no atomics)
    int Z = 1;
    const long tiles = (long)a.ntm * a.ntn;
    if (workspace && !bias && tiles < 128 && nk >= 32) {
        Z = (int)(256 / tiles);
        if (Z > 64) Z = 64;   // (16 until round 2: per-pixel weight gradients of the mask path, 1-4 tiles over 65 536 - 262 144 rows, filled 64 CUs)
        if (Z > nk / 8) Z = nk / 8;
        const long fit = workspace_bytes / (M * N * 4);
        if (Z > fit) Z = (int)fit;
        if (Z < 2) Z = 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (Z == 1) {
        t.kt_per_z = nk; t.slab = 0;
        dim3 grid((unsigned)a.ntn, (unsigned)a.ntm);
        if (out_dtype == RGA3_F32) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a, t);
        else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, st, a, t);
        RGA3_CHECK_LAUNCH("gemm_tn_kernel");
        return 0;
    }
    t.kt_per_z = (int)cdiv(nk, Z);
    Z = (int)cdiv(nk, t.kt_per_z);
    t.slab = M * N;
    a.C = workspace; a.ldc = N;
    dim3 grid((unsigned)a.ntn, (unsigned)a.ntm, (unsigned)Z);
    if (counters && tiles <= 128) {   // the slab sum happens inside the launch (last workgroup per tile, fixed order)
        RGA3_CHECK_ARG((((uintptr_t)counters) & 3) == 0, "gemm_tn: counters must be 4-byte aligned");
        t.counters = (unsigned*)counters; t.out = C; t.ldo = ldc; t.out_f32 = out_dtype == RGA3_F32 ? 1 : 0;
        hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a, t);
        RGA3_CHECK_LAUNCH("gemm_tn_kernel<fused sum>");
        return 0;
    }
    hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a, t);
    RGA3_CHECK_LAUNCH("gemm_tn_kernel");
    hipLaunchKernelGGL(tn_slab_sum_kernel, dim3((unsigned)cdiv(M * N / 4, 256)), dim3(256), 0, st, (const float*)workspace, C, (long)(M * N), (long)ldc, (int)N, Z,
                       out_dtype == RGA3_F32 ? 1 : 0);
    RGA3_CHECK_LAUNCH("tn_slab_sum_kernel");
    return 0;
}

// n (<= 4) products C_i [M_i, N_i] = A_i^T B_i (A_i [K_i, M_i], B_i [K_i, N_i] bf16 row-major) in one launch + one slab-sum launch.  ptrs: n x 3 {A, B, C}; dims: n x 7
// {M, N, K, lda, ldb, ldc, out_dtype} (HOST arrays).  Every product is cut into the K slices rga3_gemm_tn_bf16 would cut it into (same slab sums, same bits); the
// workspace must hold all slabs (sum of Z_i M_i N_i floats), else -> error (the caller then issues the products one by one).
extern "C" int rga3_gemm_tn_many(const void* const* ptrs, const int64_t* dims, int n, void* workspace, int64_t workspace_bytes, void* stream) {
    RGA3_CHECK_ARG(ptrs && dims && n >= 1 && n <= 4 && workspace && (((uintptr_t)workspace) & 15) == 0, "gemm_tn_many: bad args (1 <= n <= 4, workspace required)");
    TnMany P;
    SlabSumMany S;
    P.n = n; S.np = n;
    unsigned wg = 0, sb = 0;
    int64_t used = 0;
    for (int i = 0; i < n; ++i) {
        const void *A = ptrs[3 * i], *B = ptrs[3 * i + 1];
        void* C = (void*)ptrs[3 * i + 2];
        const int64_t M = dims[7 * i], N = dims[7 * i + 1], K = dims[7 * i + 2], lda = dims[7 * i + 3], ldb = dims[7 * i + 4], ldc = dims[7 * i + 5];
        const int out_dtype = (int)dims[7 * i + 6];
        RGA3_CHECK_ARG(A && B && C, "gemm_tn_many: null pointer (product %d)", i);
        RGA3_CHECK_ARG(M >= 8 && N >= 8 && K > 0 && M % 8 == 0 && N % 8 == 0, "gemm_tn_many: bad shape M=%ld N=%ld K=%ld (product %d)", (long)M, (long)N, (long)K, i);
        RGA3_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm_tn_many: lda/ldb must be multiples of 8 elements, ldc of 4 (product %d)", i);
        RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "gemm_tn_many: pointers must be 16-byte aligned (product %d)", i);
        RGA3_CHECK_ARG(out_dtype == RGA3_BF16 || out_dtype == RGA3_F32, "gemm_tn_many: out_dtype %d", out_dtype);
        GemmArgs& a = P.a[i];
        a = GemmArgs();
        a.A = nullptr; a.W = nullptr; a.bias = nullptr; a.res = nullptr; a.colscale = nullptr;
        a.M = (int)M; a.N = (int)N; a.K = (int)K;
        a.lda = 0; a.ldw = 0; a.ldr = 0;
        a.ntm = (int)cdiv(M, 128); a.ntn = (int)cdiv(N, 128); a.group_m = 1;
        a.ws = nullptr; a.ws_bytes = 0; a.ksl = 0; a.rowstat = nullptr; a.colc = nullptr;
        TnArgs& t = P.t[i];
        t.A = (const unsigned short*)A; t.B = (const unsigned short*)B; t.lda = lda; t.ldb = ldb; t.K = (int)K;
        t.counters = nullptr; t.out = nullptr; t.ldo = 0; t.out_f32 = 0;
        const int nk = (int)cdiv(K, 32);
        const long tiles = (long)a.ntm * a.ntn;
        int Z = 1;
        if (tiles < 128 && nk >= 32) {     // the single form's rule
            Z = (int)(256 / tiles);
            if (Z > 64) Z = 64;
            if (Z > nk / 8) Z = nk / 8;
            if (Z < 2) Z = 1;
        }
        t.kt_per_z = (int)cdiv(nk, Z);
        Z = (int)cdiv(nk, t.kt_per_z);
        t.slab = M * N;
        RGA3_CHECK_ARG(used + (int64_t)Z * M * N * 4 <= workspace_bytes, "gemm_tn_many: workspace too small (%ld bytes needed so far)", (long)(used + (int64_t)Z * M * N * 4));
        a.C = (char*)workspace + used; a.ldc = N;
        S.slabs[i] = (const float*)a.C; S.out[i] = C; S.n[i] = M * N; S.ldc[i] = ldc; S.ncols[i] = (int)N; S.Z[i] = Z; S.out_f32[i] = out_dtype == RGA3_F32 ? 1 : 0;
        used += (int64_t)Z * M * N * 4;
        P.first[i] = wg; wg += (unsigned)(tiles * Z);
        S.first[i] = sb; sb += (unsigned)cdiv(M * N / 4, 256);
    }
    for (int i = n; i <= 4; ++i) { P.first[i] = wg; S.first[i] = sb; }
    for (int i = n; i < 4; ++i) { P.a[i] = P.a[0]; P.t[i] = P.t[0]; S.slabs[i] = S.slabs[0]; S.out[i] = S.out[0]; S.n[i] = 0; S.ldc[i] = 0; S.ncols[i] = 8; S.Z[i] = 1; S.out_f32[i] = 0; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_many_kernel, dim3(wg), dim3(256), 0, st, P);
    RGA3_CHECK_LAUNCH("gemm_tn_many_kernel");
    hipLaunchKernelGGL(tn_slab_sum_many_kernel, dim3(sb), dim3(256), 0, st, S);
    RGA3_CHECK_LAUNCH("tn_slab_sum_many_kernel");
    return 0;
}
