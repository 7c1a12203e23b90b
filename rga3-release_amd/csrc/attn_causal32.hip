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
//  * K / V tiles go HBM -> LDS by 16-byte LDS-DMA (no staging registers: O 64 + S 32 + Q 32 + P 16 registers are committed), V one tile ahead and K two, one
//    barrier per iteration.  While group B is still on L both groups read the SAME tiles (stream X); afterwards group B's tiles come through a second stream (Y).
//  * THE TWO GROUPS RUN THE LOOP BODY IN ROTATED ORDER.  The two waves of a SIMD (one of each group) leave every barrier together; running the same program
//    they sit in the matrix pipe together and in the vector ALU together.  Group A runs [softmax(j), P V(j), K Q^T(j+1)] between two barriers, group B
//    [P V(j), K Q^T(j+1), softmax(j+1)]: one wave's exponentials meet the other's products.  No extra register set (A carries the 32 score registers across
//    the barrier, B the 16 packed exponentials), no extra LDS, no extra barrier.
//    LDS image: 256-byte rows, 16-byte chunk ch of row at ch ^ (((row & 3) << 2) | ((row >> 2) & 3)) -- conflict-free for the ds_read_b128 key reads (16 rows of a
//    lane group in 16 different slots) AND for the transposed value reads (the 4 rows x 64 B of a half-wave in 4 different 64-byte groups); the DMA writes
//    LDS linearly, so the permutation is applied to the per-lane SOURCE address (cdna_hip_programming.md 5.4 rule 21, T10 image (b)).
// Results: bf16 O, natural-log LSE, as rga3_attn_varlen_fwd's general kernel (same arguments); no atomics, fixed order: bitwise reproducible.
#include "attn_args.h"
#include <cstdlib>
#include <type_traits>

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
    const int c = (kH + kL + 1) >> 1;           // slots (iterations) of every wave; kL <= c <= kH

    const unsigned short* kbase = p.k + (long)ks * p.k_st + (long)hk * p.k_sh;
    const unsigned short* vbase = p.v + (long)ks * p.v_st + (long)hk * p.v_sh;

    // LDS: per stream a two-slot K ring and a two-slot V ring of 16 KiB images; slot j of either ring holds what SLOT j of the walk needs
    char* const KX = smem;
    char* const VX = smem + 2 * C32_HALF;
    char* const KY = smem + 4 * C32_HALF;
    char* const VY = smem + 6 * C32_HALF;

    // ---- LDS-DMA of one 64 x 128 image: wave w issues pieces 2w, 2w+1 (1 KiB = 4 rows each); lane -> (row 4 blk + lane / 16, physical chunk lane % 16).
    //      Address = wave-uniform tile base + a 32-bit per-lane offset fixed for the kernel (the saddr + voffset form: one add per piece; 64-bit per-lane
    //      row arithmetic per piece cost ~9 vector instructions, two of them quarter-rate multiplies, 8 - 16 times per iteration and wave).
    unsigned dko[2], dvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int blk = 2 * wid + j;
        const int row = 4 * blk + (lane >> 4);
        const int lc = (lane & 15) ^ (((lane >> 4) << 2) | (blk & 3));   // = physical chunk ^ c32_swz(row)
        dko[j] = (unsigned)(row * (int)p.k_st + lc * 8);                   // k_st, v_st < 2^24 (checked by the entry point)
        dvo[j] = (unsigned)(row * (int)p.v_st + lc * 8);
    }
    auto issue_img = [&](const unsigned short* base, long st, const unsigned (&d)[2], int tile, char* img) {
        if (tile * C32_KT + C32_KT <= Lk) {
            const unsigned short* tb = base + (long)tile * C32_KT * st;   // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned o_ = d[j];
                asm volatile("" : "+v"(o_));   // keep the offset a 32-bit register (not a zero-extended pair)
                __builtin_amdgcn_global_load_lds((gbl_void*)(tb + o_), (lds_void*)(img + (2 * wid + j) * 1024), 16, 0, 0);
            }
        } else {   // ragged last tile: rows past the end repeat the last key (masked in the softmax)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int blk = 2 * wid + j;
                const int row = 4 * blk + (lane >> 4);
                const int lc = (lane & 15) ^ (((lane >> 4) << 2) | (blk & 3));
                const int key = min(tile * C32_KT + row, Lk - 1);
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + (long)key * st + lc * 8), (lds_void*)(img + blk * 1024), 16, 0, 0);
            }
        }
    };
    // what slot j needs: stream X carries H's tiles [0, c) (group A all of them, group B its L tiles j < kL: the same key tiles); stream Y carries H's tiles
    // [c, kH) for group B's slots j >= kL
    auto issue_K = [&](int j) {
        if (j >= c) return;
        issue_img(kbase, p.k_st, dko, j, KX + (j & 1) * C32_HALF);
        const int ty = c + j - kL;
        if (j >= kL && ty < kH) issue_img(kbase, p.k_st, dko, ty, KY + (j & 1) * C32_HALF);
    };
    auto issue_V = [&](int j) {
        if (j >= c) return;
        issue_img(vbase, p.v_st, dvo, j, VX + (j & 1) * C32_HALF);
        const int ty = c + j - kL;
        if (j >= kL && ty < kH) issue_img(vbase, p.v_st, dvo, ty, VY + (j & 1) * C32_HALF);
    };
    // the refill of iteration `it` (V of slot it + 1, K of slot it + 2, both streams) cut into four steps of <= 2 pieces per wave, so that it can be spread over
    // the value product instead of queueing 8 - 16 pieces per wave -- 64 - 128 per CU -- on the address unit at the top of every iteration
    auto issue_step = [&](int it, int st) {
        const int jv = it + 1, jk = it + 2;
        if (st == 0) { if (jv < c) issue_img(vbase, p.v_st, dvo, jv, VX + (jv & 1) * C32_HALF); }
        else if (st == 1) { if (jk < c) issue_img(kbase, p.k_st, dko, jk, KX + (jk & 1) * C32_HALF); }
        else if (st == 2) { const int ty = c + jv - kL; if (jv < c && jv >= kL && ty < kH) issue_img(vbase, p.v_st, dvo, ty, VY + (jv & 1) * C32_HALF); }
        else { const int ty = c + jk - kL; if (jk < c && jk >= kL && ty < kH) issue_img(kbase, p.k_st, dko, ty, KY + (jk & 1) * C32_HALF); }
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

    // Everything below is instantiated once per group (G = 0: A, G = 1: B) and entered through ONE wave-uniform branch, so the register allocator sees two
    // disjoint programs: A carries the score registers across the barrier, B the packed exponentials -- in a single program both sets were live at the loop
    // head and the kernel spilled.
    auto run = [&](auto GRP) {
    constexpr int G = decltype(GRP)::value;
    // ---- state of the wave's current 32 query rows
    bf16x8 qf[8];
    f32x16 o[4];
    f32x16 s[2];          // scores of the slot in flight (group A keeps them across the barrier)
    bf16x8 pb[2][2];      // their exponentials as B operand (group B keeps these across the barrier)
    float m_run, l_run;
    int qw0;
    const float cs = p.scale_log2;
    auto start_rows = [&](int qb) {
        qw0 = qb * C32_QB + w4 * 32;
        const int qi = min(qw0 + r, Lq - 1);
        const unsigned short* qrow = p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh + 8 * h;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8*)(qrow + 16 * kk);
        if (p.rope_cos) {
            // RoPE on the UN-rotated queries as they are loaded (rga3_attn_varlen_fwd_rope; HF apply_multimodal_rotary_pos_emb, modeling_qwen2_5_vl.py:557-599): the
            // rotate-half partner of column d is d + 64, i.e. fragment kk + 4 of the same lane, so the rotation is lane-local.  Tables [token][128] f32; f32
            // arithmetic and one bf16 rounding, as the stand-alone pass (rope_kernel).
            const float* cs = p.rope_cos + (long)(qs + qi) * 128 + 8 * h;
            const float* sn = p.rope_sin + (long)(qs + qi) * 128 + 8 * h;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u32x4 zl = __builtin_bit_cast(u32x4, qf[kk]), zh = __builtin_bit_cast(u32x4, qf[kk + 4]);
                float cl[8], sl[8], ch[8], sh[8];
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const f32x4 a0 = *(const f32x4*)(cs + 16 * kk + 4 * q4), a1 = *(const f32x4*)(sn + 16 * kk + 4 * q4);
                    const f32x4 a2 = *(const f32x4*)(cs + 64 + 16 * kk + 4 * q4), a3 = *(const f32x4*)(sn + 64 + 16 * kk + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cl[4 * q4 + e] = a0[e]; sl[4 * q4 + e] = a1[e]; ch[4 * q4 + e] = a2[e]; sh[4 * q4 + e] = a3[e]; }
                }
                u32x4 ol, oh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = __uint_as_float(zl[e] << 16), x1 = __uint_as_float(zl[e] & 0xffff0000u);
                    const float y0 = __uint_as_float(zh[e] << 16), y1 = __uint_as_float(zh[e] & 0xffff0000u);
                    ol[e] = pack_bf2(x0 * cl[2 * e] - y0 * sl[2 * e], x1 * cl[2 * e + 1] - y1 * sl[2 * e + 1]);
                    oh[e] = pack_bf2(y0 * ch[2 * e] + x0 * sh[2 * e], y1 * ch[2 * e + 1] + x1 * sh[2 * e + 1]);
                }
                qf[kk] = __builtin_bit_cast(bf16x8, ol);
                qf[kk + 4] = __builtin_bit_cast(bf16x8, oh);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
        m_run = -INFINITY;
        l_run = 0.f;
        // consume the query fragments HERE: otherwise these ordinary loads are still pending (for the compiler's counter model) where the conditional first score
        // product is skipped, the wait for them lands at the join inside the tile loop, and -- vmcnt being in order -- it drains every LDS-DMA in flight there
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));
    };
    // bf16 O rows + LSE of the wave's rows (o unnormalised, l_full = the row sum over both half-waves).  Lane (r, h) holds, per 8-column group k = 4 b + i4, columns
    // 8 k + 4 h .. + 3 of row r; one v_permlane32_swap per dword pairs groups (k, k + 1) so that lanes r and r + 32 own 16 + 16 contiguous bytes (T21).
    auto store_rows = [&](float l_full) {
        const float inv = (l_full > 0.f) ? 1.f / l_full : 0.f;
        const int qi = qw0 + r;
        unsigned short* orow = p.o + (long)(qs + min(qi, Lq - 1)) * p.o_st + (long)hq * p.o_sh + 8 * h;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; i4 += 2) {
                unsigned ax = pack_bf2(o[b][4 * i4] * inv, o[b][4 * i4 + 1] * inv), ay = pack_bf2(o[b][4 * i4 + 2] * inv, o[b][4 * i4 + 3] * inv);
                unsigned bx_ = pack_bf2(o[b][4 * i4 + 4] * inv, o[b][4 * i4 + 5] * inv), by = pack_bf2(o[b][4 * i4 + 6] * inv, o[b][4 * i4 + 7] * inv);
                const auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx_, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (qi < Lq) *(u32x4*)(orow + 32 * b + 8 * i4) = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
            }
        if (qi < Lq && p.lse && h == 0) p.lse[(long)hq * p.total_q + qs + qi] = (l_full > 0.f) ? (m_run * 0.6931471805599453f + logf(l_full)) : -INFINITY;
    };

    // slot j of this wave: key tile, images, and whether the wave has anything to do there (wave-uniform)
    bool onH = (G == 0) || (qbL < 0);
    auto slot_tile = [&](int j) { return (G == 0 || j < kL) ? j : c + j - kL; };
    auto slot_live = [&](int j, bool on_h) {
        const int t = (G == 0 || j < kL) ? j : c + j - kL;
        return j < c && t < (on_h ? kH : kL) && qw0 < Lq && t * C32_KT <= qw0 + 31 + shift;
    };

    // ---- S^T = K Q^T of a slot: two blocks of 32 keys x 32 queries, 8 k-steps of 16 each; fragments read two k-steps ahead of their products
    auto scores = [&](int j) {
        const char* Kb = ((G == 0 || j < kL) ? KX : KY) + (j & 1) * C32_HALF;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
        bf16x8 kf[3][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) kf[kk][kb] = *(const bf16x8*)(Kb + kb * 32 * C32_ROWB + koff[kk]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk + 2 < 8) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) kf[(kk + 2) % 3][kb] = *(const bf16x8*)(Kb + kb * 32 * C32_ROWB + koff[kk + 2]);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk % 3][kb], qf[kk], s[kb], 0, 0, 0);
        }
        // pin the order above: left alone the scheduler sinks every fragment read to just before its product (fewest live registers) and each k-step then waits
        // out a whole LDS round trip for 64 cycles of matrix work
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    };
    // ---- mask + online softmax of the slot whose scores are in s; leaves the exponentials in pb
    auto softmax = [&](int j) {
        const int k0 = slot_tile(j) * C32_KT;
        if ((k0 + C32_KT > Lk) || (k0 + C32_KT - 1 > qw0 + shift)) {   // wave-uniform: only the diagonal tiles and a ragged last tile pay for the mask
            const int lim = min(Lk - 1, qw0 + r + shift);   // last visible key of this lane's query
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key > lim) s[kb][i] = -INFINITY;
                }
        }
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
        // registers 8 ss .. 8 ss + 7 of score block kb are k-step ss of the B operand of O^T += V^T P^T
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
    };
    // ---- O^T += V^T P^T of a slot
    auto values = [&](int j) {
        const char* Vb = ((G == 0 || j < kL) ? VX : VY) + (j & 1) * C32_HALF;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
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
            }
    };
    // group B: L is done (its last value product has been issued): write it out, continue on H's rows with the upper key range
    auto switch_to_H = [&]() {
        l_run += __shfl_xor(l_run, 32, 64);
        store_rows(l_run);
        start_rows(qbH);
        onH = true;
    };

    // ---- prologue: K of slots 0 and 1, V of slot 0; the wave's first rows; scores of slot 0 (group B: its exponentials too)
    issue_K(0);
    issue_V(0);
    issue_K(1);
    start_rows(onH ? qbH : qbL);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool live = slot_live(0, onH);                       // the slot whose scores / exponentials are in flight has work for this wave
    if (live) {
        scores(0);
        if constexpr (G == 1) softmax(0);
    }
    __builtin_amdgcn_s_barrier();   // every wave has read K(slot 0): iteration 0 refills that ring slot with K(slot 2)

    // The two waves of a SIMD (w and w + 4: one of each group) reach every barrier together and would run the same phases in lockstep -- both in the matrix pipe,
    // then both in the vector ALU.  So the groups run the same three steps in ROTATED order: between two barriers group A does [softmax(j), values(j),
    // scores(j+1)] and group B [values(j), scores(j+1), softmax(j+1)]: A's exponentials meet B's products and the other way round.  Both need exactly V(j) and
    // K(j+1) inside iteration j, so the rings and the single barrier per iteration are the same for both.
    for (int it = 0; it < c; ++it) {
#ifdef RGA3_AB
        const bool refill = !(p.bk_shift & 1);
        const int dbg = p.bk_shift;
#else
        constexpr bool refill = true;
        constexpr int dbg = 0;
#endif
        // The refill (V of slot it + 1, K of slot it + 2) is issued AFTER this iteration's value product: the transposed LDS reads are intrinsics without a
        // memory operand, and the compiler drains every LDS-DMA in flight (s_waitcnt vmcnt(0)) in front of each of them -- issued at the top of the iteration
        // the refill was waited for on the spot, a full memory round trip per iteration in line (the ablation showed the DMA ADDING 12 us to a 60 us kernel).
        // The ds_read_b128 of the score product do not trigger that wait.
        if constexpr (G == 0) {
            if (live) {
                if (!(dbg & 2)) softmax(it);
                if (!(dbg & 4)) values(it);
            }
            if (refill) {
#pragma unroll
                for (int st = 0; st < 4; ++st) issue_step(it, st);
            }
            live = slot_live(it + 1, true);
            if (live && !(dbg & 8)) scores(it + 1);
        } else {
            if (live && !(dbg & 4)) values(it);
            if (refill) {
#pragma unroll
                for (int st = 0; st < 4; ++st) issue_step(it, st);
            }
            if (!onH && it + 1 >= kL) switch_to_H();   // after L's last slot (also when the loop ends there: it + 1 == c == kL)
            live = slot_live(it + 1, onH);
            if (live) {
                if (!(dbg & 8)) scores(it + 1);
                if (!(dbg & 2)) softmax(it + 1);
            }
        }
#ifdef RGA3_AB
        if (dbg & 16) continue;    // no wait, no barrier (results are garbage: timing only)
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the images issued at the top have landed (own pieces) ...
        __builtin_amdgcn_s_barrier();                       // ... everyone's, and everyone is done reading this iteration's
    }

    // ---- the two partial results of H's rows meet in LDS (the rings are free: the loop's last barrier is behind every read)
    l_run += __shfl_xor(l_run, 32, 64);
    if (G == 1 && !onH) {   // c == 0 (no visible key at all): group B still holds L's empty state
        store_rows(l_run);
        start_rows(qbH);
    }
    float* const mo = (float*)smem;                       // [4 waves][16 quads][64 lanes] f32x4
    float* const mml = (float*)(smem + 4 * 16 * 64 * 16); // [4 waves][64 lanes] (m, l)
    if constexpr (G == 1) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                *(f32x4*)(mo + (((w4 * 16 + b * 4 + i4) * 64 + lane) << 2)) = f32x4{o[b][4 * i4], o[b][4 * i4 + 1], o[b][4 * i4 + 2], o[b][4 * i4 + 3]};
        *(float2*)(mml + ((w4 * 64 + lane) << 1)) = float2{m_run, l_run};
    }
    __syncthreads();
    if constexpr (G == 0) {
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
    };   // run
    if (grp == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

int launch_causal32(const AttnArgs& a, int nseg, int max_q, hipStream_t st) {
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)attn_causal32_kernel, C32_LDS, lds_grant, "attn_causal32")) return rc;
    AttnArgs b = a;
#ifdef RGA3_AB   // measurement builds only (tools/): ablation mask 1 = no tile DMA in the loop, 2 = no softmax, 4 = no P V, 8 = no K Q^T, 16 = no wait / barrier
    { const char* e = getenv("RGA3_C32_DBG"); b.bk_shift = e ? atoi(e) : 0; }
#endif
    const unsigned nqb = (unsigned)cdiv(max_q, C32_QB);
    b.gx = (int)((nqb + 1) / 2);
    hipLaunchKernelGGL(attn_causal32_kernel, dim3((unsigned)b.gx * (unsigned)a.Hq * (unsigned)nseg), dim3(512), C32_LDS, st, b);
    RGA3_CHECK_LAUNCH("attn_causal32_kernel");
    return 0;
}

}  // namespace rga3
