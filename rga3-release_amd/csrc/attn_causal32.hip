// Causal decoder attention at D = 128 on 32-row waves (reference: HF modeling_qwen2_5_vl.py:602-700, flash-attn varlen causal with GQA; call site
// reference model/qwen_2_5_vl_sam2.py:182-200).  Round 4: the decoder's S = 2112 rows ran at 0.48 PF on the general kernel (attn_fwd.hip: 8 waves x 16 rows,
// v_mfma_f32_16x16x32_bf16) -- every wave read the whole K and V tile out of LDS for 16 query rows, and the LDS array was as busy as the matrix pipe.
//
//  * 32 QUERY ROWS PER WAVE on v_mfma_f32_32x32x16_bf16 (the structure of memattn.hip): S^T = K Q^T with the key tile as A operand (one ds_read_b128 per
//    32-cycle MFMA), the wave's queries as B operand in registers; the 32 x 32 result has the query on the lane, so the online-softmax state is lane-local
//    (one cross-half exchange per tile), and registers 8s .. 8s+7 of a score block, rounded to bf16, ARE k-step s of the B operand of O^T += V^T P^T
//    (cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's operand"); V^T comes from the row-major V tile through two ds_read_b64_tr_b16.
//    Half the LDS bytes per flop of the 16-row form.
//  * BALANCED OVER THE KEY RANGE.  S = 2112 is 66 32-row blocks per head, 1848 in all: 1.8 per SIMD -- one 8-wave workgroup per CU, and every wave has to carry
//    the same number of key tiles.  A workgroup takes the pair (H, L) = (n-1-i, i) of 128-row query blocks (kH + kL key tiles of 64; constant over i).  Waves 0-3
//    (group A) walk H's tiles [0, c), c = ceil((kH + kL) / 2); waves 4-7 (group B) walk L's kL tiles and then H's tiles [c, kH) for the same H rows; the two
//    partial (O, m, l) of H meet in LDS at the end.  Every wave runs c iterations, two waves per SIMD busy throughout (the plain pairing of the general kernel ran
//    H then L on all waves: same balance, but 16 rows per wave).
//  * K / V tiles go HBM -> LDS by 16-byte LDS-DMA (no staging registers: O 64 + S 32 + Q 32 + P 16 registers are committed), one tile ahead, one barrier per
//    iteration.  While group B is still on L both groups read the SAME tile (stream X); afterwards group B's tiles come through a second stream (Y).
//    LDS image: 256-byte rows, 16-byte chunk ch of row at ch ^ (((row & 3) << 2) | ((row >> 2) & 3)) -- conflict-free for the ds_read_b128 key reads (16 rows of a
//    lane group in 16 different slots) AND for the transposed value reads (the 4 rows x 64 B of a half-wave in 4 different 64-byte groups); the DMA writes
//    LDS linearly, so the permutation is applied to the per-lane SOURCE address (cdna_hip_programming.md 5.4 rule 21, T10 image (b)).
// Results: bf16 O, natural-log LSE, as rga3_attn_varlen_fwd's general kernel (same arguments); no atomics, fixed order: bitwise reproducible.
#include "attn_args.h"

namespace rga3 {

namespace {
constexpr int C32_KT = 64;                     // keys per tile
constexpr int C32_QB = 128;                    // query rows per block (4 waves x 32)
constexpr int C32_ROWB = 256;                  // bytes per K / V row in LDS (D = 128 bf16)
constexpr int C32_HALF = C32_KT * C32_ROWB;    // K image; the V image follows
constexpr int C32_TILE = 2 * C32_HALF;         // 32 KiB
constexpr int C32_LDS = 4 * C32_TILE;          // X0 X1 Y0 Y1

__device__ __forceinline__ int c32_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
}  // namespace

__global__ __launch_bounds__(512) void attn_causal32_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, w4 = wid & 3;
    const int r = lane & 31, h = lane >> 5;

    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);   // head-fastest: the 7 query heads of a GQA group share K / V in one XCD's L2
    const int hq = (int)(lid % (unsigned)p.Hq);
    const int bx = (int)((lid / (unsigned)p.Hq) % (unsigned)p.gx);
    const int seg = (int)(lid / ((unsigned)p.Hq * (unsigned)p.gx));
    const int hk = hq / (p.Hq / p.Hkv);
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks = p.cu_k[seg], Lk = p.cu_k[seg + 1] - ks;
    const int shift = Lk - Lq;   // key j visible to query i iff j <= i + shift
    const int nqb = (Lq + C32_QB - 1) / C32_QB;
    const int qbH = nqb - 1 - bx;
    if (bx > qbH) return;                       // workgroup-uniform: no barrier below is missed
    const int qbL = (bx == qbH) ? -1 : bx;      // the middle block of an odd count has no partner: its key range is simply cut in two
    auto ktiles = [&](int qb) {
        const int e = min(Lk, qb * C32_QB + C32_QB + shift);
        return e > 0 ? (e + C32_KT - 1) / C32_KT : 0;
    };
    const int kH = ktiles(qbH), kL = qbL >= 0 ? min(ktiles(qbL), kH) : 0;
    const int c = (kH + kL + 1) >> 1;           // iterations of every wave; kL <= c <= kH

    const unsigned short* kbase = p.k + (long)ks * p.k_st + (long)hk * p.k_sh;
    const unsigned short* vbase = p.v + (long)ks * p.v_st + (long)hk * p.v_sh;

    // ---- LDS-DMA of one K / V tile: wave w issues pieces 2w, 2w+1 (1 KiB = 4 rows each) of both images; lane -> (row 4 blk + lane / 16, physical chunk lane % 16)
    auto issue_tile = [&](int tile, char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = 2 * wid + j;
            const int row = 4 * blk + (lane >> 4);
            const int lc = (lane & 15) ^ (((lane >> 4) << 2) | (blk & 3));   // = physical chunk ^ c32_swz(row)
            const int key = min(tile * C32_KT + row, Lk - 1);                  // rows past the end repeat the last key: masked below
            __builtin_amdgcn_global_load_lds((gbl_void*)(kbase + (long)key * p.k_st + lc * 8), (lds_void*)(buf + blk * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(vbase + (long)key * p.v_st + lc * 8), (lds_void*)(buf + C32_HALF + blk * 1024), 16, 0, 0);
        }
    };

    // ---- per-lane LDS offsets
    // key fragment (A operand of S^T = K Q^T): lane (r, h) reads K[32 kb + r][16 ks + 8 h .. + 8] = chunk 2 ks + h of row 32 kb + r
    const int kfr = c32_swz(r);
    int koff[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) koff[kk] = r * C32_ROWB + (((2 * kk + h) ^ kfr) << 4);
    // value fragment (A operand of O^T += V^T P^T) by transposed reads (T10): 16-lane group g: d half g & 1, key half g >> 1 (= h); lane 4 q_ + p_ of the group
    // supplies row q_, columns 4 p_ .. 4 p_ + 3 of its 4 x 16 block; element j of the fragment = key 16 ss + 8 (j >> 2) + 4 h + (j & 3)
    int tro[2][4];
    {
        const int g1 = (lane >> 4) & 1, q_ = (lane >> 2) & 3, p_ = lane & 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int x = (q_ << 2) | ((2 * jj + h) & 3);          // c32_swz(32 kb + 16 ss + 8 jj + 4 h + q_)
                const int lc = 4 * b + 2 * g1 + (p_ >> 1);
                tro[jj][b] = (8 * jj + 4 * h + q_) * C32_ROWB + ((lc ^ x) << 4) + 8 * (p_ & 1);
            }
    }

    // ---- state of the wave's current 32 query rows
    bf16x8 qf[8];
    f32x16 o[4];
    float m_run, l_run;
    int qw0;
    const float cs = p.scale_log2;
    auto start_rows = [&](int qb) {
        qw0 = qb * C32_QB + w4 * 32;
        const int qi = min(qw0 + r, Lq - 1);
        const unsigned short* qrow = p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh + 8 * h;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qrow + 16 * kk);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
        m_run = -INFINITY;
        l_run = 0.f;
    };
    // bf16 O rows + LSE of the wave's rows (o unnormalised, l_full = the row sum over both half-waves)
    auto store_rows = [&](float l_full) {
        const float inv = (l_full > 0.f) ? 1.f / l_full : 0.f;
        const int qi = qw0 + r;
        if (qi < Lq) {
            unsigned short* orow = p.o + (long)(qs + qi) * p.o_st + (long)hq * p.o_sh;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    u32x2 pk;
                    pk[0] = pack_bf2(o[b][4 * i4] * inv, o[b][4 * i4 + 1] * inv);
                    pk[1] = pack_bf2(o[b][4 * i4 + 2] * inv, o[b][4 * i4 + 3] * inv);
                    *(u32x2*)(orow + 32 * b + 8 * i4 + 4 * h) = pk;
                }
            if (p.lse && h == 0) p.lse[(long)hq * p.total_q + qs + qi] = (l_full > 0.f) ? (m_run * 0.6931471805599453f + logf(l_full)) : -INFINITY;
        }
    };

    auto process_tile = [&](int tile, const char* buf) {
        const int k0 = tile * C32_KT;
        const char* Kb = buf;
        const char* Vb = buf + C32_HALF;
        // ---- S^T = K Q^T: two blocks of 32 keys x 32 queries, 8 k-steps of 16 each (two independent accumulator chains)
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 a = *(const bf16x8*)(Kb + kb * 32 * C32_ROWB + koff[kk]);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s[kb], 0, 0, 0);
            }
        // ---- mask (wave-uniform test: only the diagonal tiles and a ragged last tile pay for it)
        if ((k0 + C32_KT > Lk) || (k0 + C32_KT - 1 > qw0 + shift)) {
            const int lim = min(Lk - 1, qw0 + r + shift);   // last visible key of this lane's query
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key > lim) s[kb][i] = -INFINITY;
                }
        }
        // ---- online softmax: lane (r, h) owns query qw0 + r and 32 of the tile's 64 keys; the row maximum is shared with the other half-wave
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * cs);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], cs, -m_use));   // raw v_exp_f32: arguments <= 0
                s[kb][i] = e;
                ps += e;
            }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {   // the accumulators stay untouched on the common path
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[b][i] *= alpha;
        }
        // ---- O^T += V^T P^T: registers 8 ss .. 8 ss + 7 of score block kb are k-step ss of the B operand
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
                for (int b = 0; b < 4; ++b) {
                    const char* a0 = Vb + (kb * 32 + 16 * ss) * C32_ROWB;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + tro[0][b]));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + tro[1][b]));
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kb][ss], o[b], 0, 0, 0);
                }
    };

    char* const X0 = smem;
    char* const Y0 = smem + 2 * C32_TILE;

    // ---- prologue: tiles of iteration 0, then the wave's first rows
    if (c > 0) {
        issue_tile(0, X0);
        if (kL == 0 && c < kH) issue_tile(c, Y0);
    }
    bool onH = (grp == 0) || (qbL < 0);
    start_rows(onH ? qbH : qbL);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < c; ++it) {
        const int cb = it & 1, nb = cb ^ 1;
        if (it + 1 < c) {
            issue_tile(it + 1, X0 + nb * C32_TILE);
            const int ty = c + it + 1 - kL;
            if (it + 1 >= kL && ty < kH) issue_tile(ty, Y0 + nb * C32_TILE);
        }
        if (grp == 1 && !onH && it == kL) {   // group B: L is done, continue on H's rows with the upper key range
            l_run += __shfl_xor(l_run, 32, 64);
            store_rows(l_run);
            start_rows(qbH);
            onH = true;
        }
        int tile;
        const char* buf;
        if (grp == 0 || it < kL) { tile = it; buf = X0 + cb * C32_TILE; }
        else { tile = c + it - kL; buf = Y0 + cb * C32_TILE; }
        // a tile entirely above the wave's last row, or rows past the segment: nothing to do (wave-uniform)
        if (tile < (onH ? kH : kL) && qw0 < Lq && tile * C32_KT <= qw0 + 31 + shift) process_tile(tile, buf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next iteration's tiles have landed (own pieces) ...
        __builtin_amdgcn_s_barrier();                       // ... everyone's, and everyone is done reading this iteration's
    }

    // ---- the two partial results of H's rows meet in LDS (the tile buffers are free: the loop's last barrier is behind every read)
    l_run += __shfl_xor(l_run, 32, 64);
    if (grp == 1 && !onH) {   // group B never reached H (kH - c == 0): finish L, contribute nothing
        store_rows(l_run);
        start_rows(qbH);
    }
    float* const mo = (float*)smem;                       // [4 waves][16 quads][64 lanes] f32x4
    float* const mml = (float*)(smem + 4 * 16 * 64 * 16); // [4 waves][64 lanes] (m, l)
    if (grp == 1) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                *(f32x4*)(mo + (((w4 * 16 + b * 4 + i4) * 64 + lane) << 2)) = f32x4{o[b][4 * i4], o[b][4 * i4 + 1], o[b][4 * i4 + 2], o[b][4 * i4 + 3]};
        *(float2*)(mml + ((w4 * 64 + lane) << 1)) = float2{m_run, l_run};
    }
    __syncthreads();
    if (grp == 0) {
        const float2 ml = *(const float2*)(mml + ((w4 * 64 + lane) << 1));
        const float m = fmaxf(m_run, ml.x);
        const float mu = (m == -INFINITY) ? 0.f : m;
        const float wa = __builtin_amdgcn_exp2f(m_run - mu), wb = __builtin_amdgcn_exp2f(ml.x - mu);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 ob = *(const f32x4*)(mo + (((w4 * 16 + b * 4 + i4) * 64 + lane) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) o[b][4 * i4 + e] = o[b][4 * i4 + e] * wa + ob[e] * wb;
            }
        m_run = m;
        store_rows(l_run * wa + ml.y * wb);
    }
}

int launch_causal32(const AttnArgs& a, int nseg, int max_q, hipStream_t st) {
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)attn_causal32_kernel, C32_LDS, lds_grant, "attn_causal32")) return rc;
    AttnArgs b = a;
    const unsigned nqb = (unsigned)cdiv(max_q, C32_QB);
    b.gx = (int)((nqb + 1) / 2);
    hipLaunchKernelGGL(attn_causal32_kernel, dim3((unsigned)b.gx * (unsigned)a.Hq * (unsigned)nseg), dim3(512), C32_LDS, st, b);
    RGA3_CHECK_LAUNCH("attn_causal32_kernel");
    return 0;
}

}  // namespace rga3
