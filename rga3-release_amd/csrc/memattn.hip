// SAM2 memory cross-attention for gfx950 (reference model/sam2.py:448-530 MemoryAttentionLayer._forward_ca, :1484-1548 RoPEAttention): one 256-wide head,
// 4096 queries (the current frame's tokens) against the memory bank -- up to 7 x 4096 spatial memory tokens + 64 object-pointer tokens = 28 736 keys whose
// SOURCE is 64 wide (maskmem features, mem_dim 64); the layer projects them 64 -> 256 for keys (+ axial RoPE) and 64 -> 256 for values.
//
// Two things make this kernel different from the general attention kernel (attn_fwd.hip), which ran this call at 0.41 PF:
//
//  * VALUES STAY IN MEMORY SPACE.  V = M Wv^T + 1 bv^T is linear in the 64-wide memory rows M, and softmax rows sum to one, so
//        softmax(S) V = (softmax(S) M) Wv^T + bv .
//    The kernel accumulates  PM = softmax(S) M  (64 wide instead of 256): the P.V product has a quarter of the flops, its accumulator a quarter of the registers,
//    the V tile a quarter of the LDS bytes, and the 28 736 x 256 value projection (a GEMM + 14.7 MB written and re-read per layer and frame) is never formed.  The
//    caller applies the 64 -> 256 map once per QUERY row, folded with the output projection:  out = PM (Wo Wv)^T + (Wo bv + bo)  (rga3/model/sam2.py).
//
//  * 32 QUERY ROWS PER WAVE on v_mfma_f32_32x32x16_bf16.  With S^T = K Q^T the key tile is the A operand (one ds_read_b128 = 1 KiB per MFMA) and the wave's 32
//    queries sit in registers as the B operand; the 16-row form read the same 1 KiB per 16-cycle MFMA and was LDS-bound (4096 LDS cycles against 2048 MFMA cycles
//    per tile and CU), this one needs 4 LDS cycles per 32-cycle MFMA and SIMD: half the LDS array.  The 32 x 32 f32 result has the query on the lane and the keys
//    in the 16 registers, so (i) the online-softmax state is lane-local (one cross-half exchange per tile for the row maximum), and (ii) registers 8s .. 8s+7,
//    rounded to bf16, ARE the B operand of k-step s of PM^T = M^T P^T with the k order permuted (cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's
//    operand"); the A operand M^T comes from the row-major memory tile through two ds_read_b64_tr_b16 per MFMA that follow the same permutation.
//
// Workgroup = 8 waves x 32 queries = 256 query rows x one slice of the keys (split-KV: 16 query blocks x 16 key slices = 256 workgroups, one per CU; consecutive
// workgroup ids -- one XCD -- share a key slice, so a slice's K / M rows are fetched into one L2).  64-key tiles travel HBM -> registers -> LDS, double-buffered, one
// barrier per tile.  LDS images: K rows of 512 B padded to 528 B (33 x 16 B: the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots), M rows of 128 B
// padded to 192 B (the 4 rows x 64 B a 32-lane half of a transposed read touches fall into disjoint bank ranges).  Partial results (unnormalised PM, running
// maximum, row sum) go to a caller-owned f32 workspace; memattn_combine_kernel merges the slices in slice order and writes bf16 [Nq, 64].
// No atomics, fixed summation order: bitwise reproducible.
#include "common.h"

#include <math.h>

namespace rga3 {

constexpr int MA_KT = 64;                                      // keys per tile
constexpr int MA_D = 256;                                      // q / k width
constexpr int MA_DM = 64;                                      // memory width (value source of the cross-attention)
constexpr int MA_KSTR = MA_D * 2 + 16;                         // 528 B per K row in LDS
constexpr int MA_MAX_SPLIT = 32;
// value rows in LDS: DM * 2 + 64 B (192 B for the 64-wide memory rows, 576 B for 256-wide values): the 4 rows x 64 B a 32-lane half of a transposed read touches
// start 16 banks apart
template <int DM> struct MaGeom {
    static constexpr int MSTR = DM * 2 + 64;
    static constexpr int TILE_BYTES = MA_KT * (MA_KSTR + MSTR);   // 46 080 (DM 64) / 70 656 (DM 256)
    static constexpr int LDS = 2 * TILE_BYTES;                    // 92 160 / 141 312
};

struct MemAttnArgs {
    const unsigned short* q;   // [Nq, 256] bf16 (projected, rotated)
    const unsigned short* k;   // [Nk, 256] bf16 (projected, rotated)
    const unsigned short* m;   // [Nk, DM] bf16 value rows (DM 64: the un-projected memory rows; DM 256: the self-attention's values)
    unsigned short* out;       // [Nq, DM] bf16 = softmax(q k^T scale) m
    float* part_o;             // [nsplit, Nq, DM] f32 unnormalised partial sums
    float* part_ml;            // [nsplit, Nq, 2] f32 (running maximum in the log2 domain, row sum)
    long q_st, k_st, m_st, out_st;
    int Nq, Nk, nsplit;
    float scale_log2;
};

// DM: value width, NW: waves per workgroup.  The product uses DM 64 (values kept in memory space) with 8 waves x 32 queries (216 VGPRs, two waves per SIMD).
// Round 3 also ran the layer's SELF-attention through this loop (DM 256: 8 output blocks = 128 accumulator registers, so 4 waves, one per SIMD, on the 512-register
// budget): correct, 43.8 us against 47.6 us for the general attention kernel on 4096 x 4096 (tools/memlayer_probe.py), but merging its 256-wide f32 partials costs
// what it saves -- the 32-frame stream ran at 619.9 vs 619.6 and 604.9 vs 610.0 frames/s with and without it -- so that instantiation is not shipped.
template <int DM, int NW>
__global__ __launch_bounds__(64 * NW) void memattn_cross_kernel(MemAttnArgs p) {
    constexpr int NT = 64 * NW, MA_QB = 32 * NW, NB = DM / 32, MA_MSTR = MaGeom<DM>::MSTR, MA_TILE_BYTES = MaGeom<DM>::TILE_BYTES;
    constexpr int KJ = MA_KT * (MA_D / 8) / NT, MJ = MA_KT * (DM / 8) / NT, MCH = DM / 8;   // 16-byte chunks per thread and tile: keys, values; chunks per value row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nqb = (p.Nq + MA_QB - 1) / MA_QB;
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = (int)(lid % (unsigned)nqb), split = (int)(lid / (unsigned)nqb);
    const int q0 = qb * MA_QB + wave * 32;
    const int ntiles = (p.Nk + MA_KT - 1) / MA_KT;
    const int per = (ntiles + p.nsplit - 1) / p.nsplit;
    const int t0 = split * per, t1 = min(ntiles, t0 + per);

    // this wave's 32 query rows as the B operand of S^T = K Q^T: lane (r, h) holds Q[q0 + r][16 ks + 8 h .. + 8] for k-step ks
    bf16x8 qf[16];
    {
        const int qi = min(q0 + r, p.Nq - 1);
        const unsigned short* qrow = p.q + (long)qi * p.q_st + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
    }

    u32x4 kreg[KJ], mreg[MJ];
    auto load_tile = [&](int t) {
        const int k0 = t * MA_KT;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int idx = tid + j * NT, row = idx >> 5, ch = idx & 31;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (k0 + row < p.Nk) z = *(const u32x4*)(p.k + (long)(k0 + row) * p.k_st + ch * 8);
            kreg[j] = z;
        }
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int idx = tid + j * NT, row = idx / MCH, ch = idx % MCH;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (k0 + row < p.Nk) z = *(const u32x4*)(p.m + (long)(k0 + row) * p.m_st + ch * 8);
            mreg[j] = z;
        }
    };
    auto store_tile = [&](int buf) {
        char* Kb = smem + buf * MA_TILE_BYTES;
        char* Mb = Kb + MA_KT * MA_KSTR;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int idx = tid + j * NT, row = idx >> 5, ch = idx & 31;
            *(u32x4*)(Kb + row * MA_KSTR + ch * 16) = kreg[j];
        }
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int idx = tid + j * NT, row = idx / MCH, ch = idx % MCH;
            *(u32x4*)(Mb + row * MA_MSTR + ch * 16) = mreg[j];
        }
    };

    f32x16 o[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = p.scale_log2;
    // transposed-read lane roles (cdna_hip_programming.md T10): 16-lane group g, lane 4 q_ + p_ of the group supplies row q_, columns 4 p_ .. 4 p_ + 3 of its block
    const int g = lane >> 4, q_ = (lane >> 2) & 3, p_ = lane & 3;
    const int tr_off = (4 * (g >> 1) + q_) * MA_MSTR + (16 * (g & 1) + 4 * p_) * 2;

    if (t0 < t1) {
        load_tile(t0);
        store_tile(0);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        if (t + 1 < t1) load_tile(t + 1);
        const char* Kb = smem + buf * MA_TILE_BYTES;
        const char* Mb = Kb + MA_KT * MA_KSTR;
        // ---- S^T = K Q^T: two blocks of 32 keys x 32 queries, 16 k-steps each
        // (the two key blocks advance together: two independent accumulator chains, so a wave that is alone on its SIMD -- DM 256 -- does not wait out the
        // latency of every product; the same for the output blocks below)
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 a = *(const bf16x8*)(Kb + (kb * 32 + r) * MA_KSTR + h * 16 + ks * 32);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kb], 0, 0, 0);
            }
        if ((t + 1) * MA_KT > p.Nk) {   // ragged last tile: keys past the end see nothing (wave-uniform branch)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = t * MA_KT + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= p.Nk) s[kb][i] = -INFINITY;
                }
        }
        // ---- online softmax: lane (r, h) owns query q0 + r and 32 of the tile's 64 keys; the row maximum is shared with the other half-wave
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], c, -m_use));
                s[kb][i] = e;
                ps += e;
            }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[b][i] *= alpha;
        }
        // ---- PM^T += M^T P^T: registers 8 ss .. 8 ss + 7 of a block are k-step ss of the B operand (k order permuted: element j of half h = key 16 ss + 8 (j >> 2) + 4 h + (j & 3))
        bf16x8 pb[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                u32x4 pk;
                pk[0] = pack_bf2(s[kb][8 * ss + 0], s[kb][8 * ss + 1]);
                pk[1] = pack_bf2(s[kb][8 * ss + 2], s[kb][8 * ss + 3]);
                pk[2] = pack_bf2(s[kb][8 * ss + 4], s[kb][8 * ss + 5]);
                pk[3] = pack_bf2(s[kb][8 * ss + 6], s[kb][8 * ss + 7]);
                pb[kb][ss] = __builtin_bit_cast(bf16x8, pk);
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const char* a0 = Mb + (kb * 32 + 16 * ss) * MA_MSTR + b * 64 + tr_off;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 8 * MA_MSTR));
                    bf16x8 mf;
                    mf[0] = lo[0]; mf[1] = lo[1]; mf[2] = lo[2]; mf[3] = lo[3];
                    mf[4] = hi[0]; mf[5] = hi[1]; mf[6] = hi[2]; mf[7] = hi[3];
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mf, pb[kb][ss], o[b], 0, 0, 0);
                }
        if (t + 1 < t1) store_tile(buf ^ 1);
        __syncthreads();
    }
    // ---- partial result of this key slice: lane (r, h) holds PM[q0 + r][b*32 + 8 i4 + 4 h + (0..3)] in o[b][4 i4 .. 4 i4 + 3]
    l_run += __shfl_xor(l_run, 32, 64);
    const int qi = q0 + r;
    if (qi < p.Nq) {
        float* po = p.part_o + ((long)split * p.Nq + qi) * DM;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                f32x4 v = {o[b][4 * i4], o[b][4 * i4 + 1], o[b][4 * i4 + 2], o[b][4 * i4 + 3]};
                *(f32x4*)(po + b * 32 + 8 * i4 + 4 * h) = v;
            }
        if (h == 0) {
            float* pm = p.part_ml + ((long)split * p.Nq + qi) * 2;
            pm[0] = m_run;
            pm[1] = l_run;
        }
    }
}

// out[q] = sum_s w_s PM_s[q] / sum_s w_s l_s[q],  w_s = 2^(m_s - max_s m_s): one wave per query row (lane = DM / 64 consecutive columns), slices added in slice order
template <int DM>
__global__ __launch_bounds__(256) void memattn_combine_kernel(MemAttnArgs p) {
    constexpr int CPL = DM / 64;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p.Nq) return;
    float m = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) m = fmaxf(m, p.part_ml[((long)s * p.Nq + row) * 2]);
    float acc[CPL], l = 0.f;
#pragma unroll
    for (int e = 0; e < CPL; ++e) acc[e] = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
        const float ms = p.part_ml[((long)s * p.Nq + row) * 2];
        const float w = (ms == -INFINITY) ? 0.f : exp2f(ms - m);
        l += w * p.part_ml[((long)s * p.Nq + row) * 2 + 1];
#pragma unroll
        for (int e = 0; e < CPL; ++e) acc[e] += w * p.part_o[((long)s * p.Nq + row) * DM + lane * CPL + e];
    }
#pragma unroll
    for (int e = 0; e < CPL; ++e) p.out[row * p.out_st + lane * CPL + e] = f2bf(l > 0.f ? acc[e] / l : 0.f);
}

}  // namespace rga3

using namespace rga3;

template <int DM, int NW>
static int memattn_launch(const void* q, const void* k, const void* m, void* out, int64_t Nq, int64_t Nk, int64_t q_stride, int64_t k_stride, int64_t m_stride,
                          int64_t out_stride, float scale, int nsplit, float* ws, void* stream, const char* name) {
    RGA3_CHECK_ARG(q && k && m && ws, "%s: null pointer", name);
    RGA3_CHECK_ARG(Nq > 0 && Nk > 0 && Nq < (1 << 24) && Nk < (1 << 24), "%s: Nq %ld Nk %ld", name, (long)Nq, (long)Nk);
    RGA3_CHECK_ARG(nsplit >= 1 && nsplit <= MA_MAX_SPLIT, "%s: nsplit %d", name, nsplit);
    RGA3_CHECK_ARG(q_stride % 8 == 0 && k_stride % 8 == 0 && m_stride % 8 == 0 && q_stride >= MA_D && k_stride >= MA_D && m_stride >= DM && (!out || out_stride >= DM),
                   "%s: strides", name);
    RGA3_CHECK_ARG(scale > 0.f, "%s: scale must be positive", name);
    RGA3_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)m | (uintptr_t)ws) & 15) == 0, "%s: 16-byte alignment", name);
    auto kern = memattn_cross_kernel<DM, NW>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, MaGeom<DM>::LDS, lds_grant, name)) return rc;
    MemAttnArgs a;
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.m = (const unsigned short*)m; a.out = (unsigned short*)out;
    a.part_o = ws; a.part_ml = ws + (int64_t)nsplit * Nq * DM;
    a.q_st = q_stride; a.k_st = k_stride; a.m_st = m_stride; a.out_st = out_stride;
    a.Nq = (int)Nq; a.Nk = (int)Nk; a.nsplit = nsplit;
    a.scale_log2 = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nqb = (unsigned)cdiv(Nq, 32 * NW);
    hipLaunchKernelGGL(kern, dim3(nqb * (unsigned)nsplit), dim3(64 * NW), MaGeom<DM>::LDS, st, a);
    RGA3_CHECK_LAUNCH(name);
    if (!out) return 0;
    hipLaunchKernelGGL(memattn_combine_kernel<DM>, dim3((unsigned)cdiv(Nq, 4)), dim3(256), 0, st, a);
    RGA3_CHECK_LAUNCH(name);
    return 0;
}

// floats of caller workspace for nsplit key slices over Nq query rows (partial sums + (max, sum) pairs)
extern "C" int64_t rga3_memattn_cross_ws_floats(int64_t Nq, int nsplit) {
    if (Nq <= 0 || nsplit <= 0 || nsplit > MA_MAX_SPLIT) return -1;
    return (int64_t)nsplit * Nq * (MA_DM + 2);
}

// out [Nq, 64] bf16 = softmax(scale * q k^T) m  with q [Nq, 256], k [Nk, 256], m [Nk, 64] bf16 (row strides in elements, 16-byte aligned rows).
// nsplit key slices (1 .. 32; the caller sizes it so that ceil(Nq / 256) * nsplit covers the CUs), ws = rga3_memattn_cross_ws_floats(Nq, nsplit) floats.
// out = NULL: only the partial results are left in ws ([nsplit, Nq, 64] unnormalised sums, then [nsplit, Nq, 2] (maximum in the log2 domain, row sum)); the consumer
// merges them (rga3_memlayer_rows does, in the same slice order).
extern "C" int rga3_memattn_cross(const void* q, const void* k, const void* m, void* out, int64_t Nq, int64_t Nk, int64_t q_stride, int64_t k_stride,
                                  int64_t m_stride, int64_t out_stride, float scale, int nsplit, float* ws, void* stream) {
    return memattn_launch<MA_DM, 8>(q, k, m, out, Nq, Nk, q_stride, k_stride, m_stride, out_stride, scale, nsplit, ws, stream, "memattn_cross");
}
