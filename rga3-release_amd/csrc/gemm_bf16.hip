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
};

// 16 bytes of zeros in device memory: source for staging chunks that lie beyond K in the last K-tile
__device__ const u32x4 g_zero16 = {0u, 0u, 0u, 0u};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

template <int BM, int BN, int WM, int WN, int ACT, bool OUT_F32, int PIPE>
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

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    // ---- logical tile for this workgroup: XCD-contiguous chunks, grouped (GROUP_M rows of tiles) order
    const unsigned nwg = (unsigned)(p.ntm * p.ntn);
    const unsigned t = xcd_remap(blockIdx.x, nwg);
    constexpr unsigned GROUP_M = 4;
    const unsigned per_group = GROUP_M * p.ntn;
    const unsigned group = t / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = min((unsigned)p.ntm - first_m, GROUP_M);
    const unsigned tm = first_m + (t % per_group) % gsz;
    const unsigned tn = (t % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

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

    const bool ktail = (p.K % BK) != 0;  // K is a multiple of 8: the last tile may be partial
    auto stage_tile = [&](int s, int kt, bool last) {
        char* sa = smem + s * STAGE;
        char* sb = sa + BM * ROWB;
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

    const int nk = (p.K + BK - 1) / BK;
    if constexpr (PIPE == 0) {
    stage_tile(0, 0, nk == 1);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed (own loads: vmcnt(0); everyone's: barrier) and everyone is done reading
        // the other stage, so it can be refilled while this one is consumed.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_tile((kt + 1) & 1, kt + 1, kt + 2 == nk);
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
    } else {
    // Two-phase register pipeline: the fragments of the NEXT 32-deep k-step are read from LDS while the MFMAs of the
    // current one issue (two fragment sets ping-pong), so no ds_read latency is exposed after the single barrier per
    // K-tile; the LDS-DMA of tile kt+2 is issued right after that barrier and has a whole tile of MFMAs to land.
    bf16x8 af0[MT], wf0[NTL], af1[MT], wf1[NTL];
    auto load_set = [&](bf16x8* af, bf16x8* wf, int stage, int kk) {
        const char* As = smem + stage * STAGE + (wm * WTM) * ROWB;
        const char* Bs = smem + stage * STAGE + BM * ROWB + (wn * WTN) * ROWB;
#pragma unroll
        for (int j = 0; j < NTL; ++j) wf[j] = *(const bf16x8*)(Bs + j * 16 * ROWB + foff[kk]);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(As + i * 16 * ROWB + foff[kk]);
    };
    auto mma_set = [&](const bf16x8* af, const bf16x8* wf) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&]() {
        // one LDS fragment read between consecutive MFMAs while reads remain, then the rest of the MFMAs
#pragma unroll
        for (int g2 = 0; g2 < MT + NTL; ++g2) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NTL - (MT + NTL), 0);
    };
    stage_tile(0, 0, nk == 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nk > 1) stage_tile(1, 1, nk == 2);
    load_set(af0, wf0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        // phase A: prefetch (kt, kk=1), multiply (kt, kk=0)
        load_set(af1, wf1, st, 1);
        mma_set(af0, wf0);
        interleave();
        // my reads of stage st are complete, my DMA of tile kt+1 has landed; after the barrier that holds for everyone
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // phase B: refill stage st with tile kt+2, prefetch (kt+1, kk=0), multiply (kt, kk=1)
        if (kt + 2 < nk) stage_tile(st, kt + 2, kt + 3 == nk);
        if (kt + 1 < nk) load_set(af0, wf0, st ^ 1, 0);
        mma_set(af1, wf1);
        interleave();
    }
    }
    __syncthreads();  // all waves done with the last stage: LDS is free for the epilogue staging

    // ---- epilogue.  Lane (g = lane>>4, c = lane&15) holds, for m-tile i / n-tile j, row c and
    //      columns 4g..4g+3 of the 16x16 block.
    constexpr int OUT_NT = (ACT == ACT_SWIGLU) ? NTL / 2 : NTL;  // output n-tiles per wave
    constexpr int OW = OUT_NT * 16;                               // output columns per wave
    constexpr int ESZ = OUT_F32 ? 4 : 2;
    constexpr int EDAT = OW * ESZ;  // payload bytes per staged row
    constexpr int EROW = EDAT + 16; // +16 B pad: rows no longer alias on the 128-B ds_write bank period (was 16-way conflicts)
    char* est = smem + wid * (16 * EROW);
    const int g = lane >> 4, c = lane & 15;
    const int ncol0 = (ACT == ACT_SWIGLU) ? (n0 / 2 + wn * OW) : (n0 + wn * OW);  // first output column of this wave
    const int Nout = (ACT == ACT_SWIGLU) ? p.N / 2 : p.N;

#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int jo = 0; jo < OUT_NT; ++jo) {
            float v[4];
            if constexpr (ACT == ACT_SWIGLU) {
                // packed weight layout: n-tile 2*jo = gate columns, 2*jo+1 = up columns of the same 16 outputs
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gt = acc[i][2 * jo][r], up = acc[i][2 * jo + 1][r];
                    if (p.bias) {
                        int nb = n0 + wn * WTN + (2 * jo) * 16 + 4 * g + r;
                        gt += bf2f(p.bias[min(nb, p.N - 1)]);
                        up += bf2f(p.bias[min(nb + 16, p.N - 1)]);
                    }
                    // reference rounds gate/up linear outputs to bf16 before the activation (bf16 nn.Linear)
                    gt = bf2f(f2bf(gt));
                    up = bf2f(f2bf(up));
                    v[r] = bf2f(f2bf(silu_f(gt))) * up;
                    if (p.colscale) v[r] *= bf2f(p.colscale[min(ncol0 + jo * 16 + 4 * g + r, Nout - 1)]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][jo][r];
                    if (p.bias) {
                        int nb = n0 + wn * WTN + jo * 16 + 4 * g + r;
                        x += bf2f(p.bias[min(nb, p.N - 1)]);
                    }
                    if constexpr (ACT == ACT_GELU) x = gelu_erf(bf2f(f2bf(x)));
                    if constexpr (ACT == ACT_RELU) x = fmaxf(x, 0.f);
                    if (p.colscale) x = bf2f(f2bf(x)) * bf2f(p.colscale[min(ncol0 + jo * 16 + 4 * g + r, Nout - 1)]);
                    v[r] = x;
                }
            }
            if constexpr (OUT_F32) {
                *(f32x4*)(est + c * EROW + (jo * 16 + 4 * g) * 4) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                u32x2 pk;
                pk[0] = pack_bf2(v[0], v[1]);
                pk[1] = pack_bf2(v[2], v[3]);
                *(u32x2*)(est + c * EROW + (jo * 16 + 4 * g) * 2) = pk;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // read back row-major: 16 rows x EROW bytes, 16 B per lane
        constexpr int CPR = EDAT / 16;       // 16-byte chunks per row
        constexpr int TOTAL = 16 * CPR;      // chunks in the staged block
        constexpr int EPC = 16 / ESZ;        // elements per chunk
#pragma unroll
        for (int q = lane; q < TOTAL; q += 64) {
            int rr = q / CPR, cc = q % CPR;
            int gm = m0 + wm * WTM + i * 16 + rr;
            int gn = ncol0 + cc * EPC;
            u32x4 val = *(const u32x4*)(est + rr * EROW + cc * 16);
            if (gm < p.M && gn < Nout) {
                if constexpr (OUT_F32) {
                    float* dst = (float*)p.C + (long)gm * p.ldc + gn;
                    if (gn + 4 <= Nout && ((p.ldc & 3) == 0)) {
                        *(u32x4*)dst = val;
                    } else {
                        for (int e = 0; e < 4 && gn + e < Nout; ++e) dst[e] = __uint_as_float(val[e]);
                    }
                } else {
                    unsigned short* dst = (unsigned short*)p.C + (long)gm * p.ldc + gn;
                    const bool vec = (gn + 8 <= Nout) && ((p.ldc & 7) == 0);
                    if (p.res) {
                        const unsigned short* rs = p.res + (long)gm * p.ldr + gn;
                        unsigned short o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            unsigned int w = val[e >> 1];
                            unsigned short h = (e & 1) ? (unsigned short)(w >> 16) : (unsigned short)(w & 0xffff);
                            float rv = (gn + e < Nout) ? bf2f(rs[e]) : 0.f;
                            o[e] = f2bf(bf2f(h) + rv);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = (unsigned)o[2 * e] | ((unsigned)o[2 * e + 1] << 16);
                    }
                    if (vec) {
                        *(u32x4*)dst = val;
                    } else {
                        for (int e = 0; e < 8 && gn + e < Nout; ++e) {
                            unsigned int w = val[e >> 1];
                            dst[e] = (e & 1) ? (unsigned short)(w >> 16) : (unsigned short)(w & 0xffff);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int BM, int BN, int WM, int WN, int ACT, bool OUT_F32, int PIPE>
static int launch_cfg(const GemmArgs& a0, hipStream_t st) {
    GemmArgs a = a0;
    a.ntm = (int)cdiv(a.M, BM);
    a.ntn = (int)cdiv(a.N, BN);
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int LDS = 2 * STAGE;
    auto kern = gemm_nt_kernel<BM, BN, WM, WN, ACT, OUT_F32, PIPE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail(-(int)e, "gemm: hipFuncSetAttribute(%d): %s", LDS, hipGetErrorString(e));
        attr_done = true;
    }
    dim3 grid((unsigned)(a.ntm * a.ntn));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), LDS, st, a);
    RGA3_CHECK_LAUNCH("gemm_nt_kernel");
    return 0;
}

// tile choice: fill the 256 CUs.  score = useful fraction of the last wave of tiles x a per-config prior.
static int pick_tile(int M, int N, int forced) {
    if (forced >= 0) return forced;
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
    return best;
}

template <int ACT, bool OUT_F32>
static int launch_act(const GemmArgs& a, int tile, hipStream_t st) {
    switch (tile) {
        case 0: return launch_cfg<256, 256, 2, 4, ACT, OUT_F32, 1>(a, st);
        case 1: return launch_cfg<256, 128, 2, 4, ACT, OUT_F32, 1>(a, st);
        case 2: return launch_cfg<128, 128, 2, 2, ACT, OUT_F32, 1>(a, st);
        case 3: return launch_cfg<128, 256, 2, 4, ACT, OUT_F32, 0>(a, st);
        case 10: return launch_cfg<256, 256, 2, 4, ACT, OUT_F32, 0>(a, st);
        case 11: return launch_cfg<256, 128, 2, 4, ACT, OUT_F32, 0>(a, st);
        default: return launch_cfg<128, 128, 2, 2, ACT, OUT_F32, 0>(a, st);
    }
}

}  // namespace rga3

using namespace rga3;

extern "C" int rga3_gemm_bf16(const void* A, const void* W, const void* bias, const void* residual, const void* colscale, void* C,
                              int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr,
                              int act, int out_dtype, int tile, void* stream) {
    RGA3_CHECK_ARG(A && W && C, "gemm: null pointer");
    RGA3_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    RGA3_CHECK_ARG(K % 8 == 0, "gemm: K=%ld must be a multiple of 8 (16-byte staging chunks)", (long)K);
    RGA3_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements (16-byte rows)");
    RGA3_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0, "gemm: pointers must be 16-byte aligned");
    RGA3_CHECK_ARG(out_dtype == RGA3_BF16 || out_dtype == RGA3_F32, "gemm: out_dtype %d", out_dtype);
    RGA3_CHECK_ARG(act >= 0 && act <= 3, "gemm: act %d", act);
    RGA3_CHECK_ARG(!(out_dtype == RGA3_F32 && (act != ACT_NONE || residual || colscale)), "gemm: f32 output supports bias only");
    RGA3_CHECK_ARG(act != ACT_SWIGLU || N % 32 == 0, "gemm: swiglu needs N %% 32 == 0");
    RGA3_CHECK_ARG(tile >= -1 && tile <= 12, "gemm: tile %d", tile);
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
    hipStream_t st = (hipStream_t)stream;
    int tl = pick_tile((int)M, (int)N, tile);
    if (out_dtype == RGA3_F32) return launch_act<ACT_NONE, true>(a, tl, st);
    switch (act) {
        case ACT_NONE: return launch_act<ACT_NONE, false>(a, tl, st);
        case ACT_GELU: return launch_act<ACT_GELU, false>(a, tl, st);
        case ACT_SWIGLU: return launch_act<ACT_SWIGLU, false>(a, tl, st);
        default: return launch_act<ACT_RELU, false>(a, tl, st);
    }
}
