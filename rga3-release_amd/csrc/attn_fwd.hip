// Variable-length fused attention forward for CDNA4 (wave64, v_mfma_f32_16x16x32_bf16).
//
// One kernel family covers every softmax(QK^T)V call site on the RGA3 hot path (SURVEY.md 2.2 K3, K8, K12,
// K16, K18): ViT window/full attention (HF modeling_qwen2_5_vl.py:211-291), causal GQA decoder attention
// (:602-700), Hiera windowed attention (reference model/sam2.py:1021), two-way decoder attention (:1476)
// and memory attention (:1543).
//
// Layout of the computation (chosen for the MFMA register maps, not translated from a CUDA kernel):
//  * S^T = K . Q^T  — K fragment is the MFMA "A" operand (rows = keys, straight ds_read_b128 from a row-major
//    LDS image), Q fragment the "B" operand (registers, loaded once).  The 16x16 result puts the QUERY on the
//    lane (lane&15) and 4 keys in the registers, so the online-softmax state (m, l, alpha) is one scalar per lane.
//  * O^T = V^T . P^T — the P registers are, as they stand, the "B" operand of the next MFMA (k order
//    permuted: element j of lane group g is key 16*(j>>2) + 4g + (j&3) of the 32-key step); the V^T "A"
//    operand with the same k permutation is exactly what two ds_read_b64_tr_b16 transposed reads of the
//    row-major V image deliver.  Result again has the query on the lane: alpha scaling is lane-local.
//  * K/V tiles are register-staged (global_load 16 B -> ds_write_b128) one tile ahead, into rows padded by
//    32 B so that both the b128 K reads and the transposed V reads are bank-conflict-free.
#include "attn_args.h"
#include <type_traits>

namespace rga3 {


// x[0..7] (bf16x8 as u32x4) of row `row_ptr` at column d, rotated: x cos + rotate_half(x) sin in f32, one bf16 rounding (rope_kernel's arithmetic)
__device__ __forceinline__ u32x4 rope_chunk(u32x4 z, const unsigned short* row_ptr, int d, int D, const float* cs, const float* sn) {
    const int half = D >> 1;
    const bool first = d < half;
    const u32x4 zp = *(const u32x4*)(row_ptr + (first ? d + half : d - half));
    const f32x4 c0 = *(const f32x4*)(cs + d), c1 = *(const f32x4*)(cs + d + 4), s0 = *(const f32x4*)(sn + d), s1 = *(const f32x4*)(sn + d + 4);
    const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    u32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = __uint_as_float(z[e] << 16), x1 = __uint_as_float(z[e] & 0xffff0000u);
        const float p0 = __uint_as_float(zp[e] << 16), p1 = __uint_as_float(zp[e] & 0xffff0000u);
        const float a0 = first ? x0 * cc[2 * e] - p0 * ss[2 * e] : x0 * cc[2 * e] + p0 * ss[2 * e];
        const float a1 = first ? x1 * cc[2 * e + 1] - p1 * ss[2 * e + 1] : x1 * cc[2 * e + 1] + p1 * ss[2 * e + 1];
        out[e] = pack_bf2(a0, a1);
    }
    return out;
}

constexpr int KV_TILE = 64;

template <int DP, int QT, int NWAVE, bool USE_TR, bool PAIR, bool SPLIT = false, bool ROPE = false, int KT = KV_TILE, int DTO = DP / 16>
__global__ __launch_bounds__(64 * NWAVE) void attn_fwd_kernel(AttnArgs p) {
    static_assert(!(PAIR && SPLIT), "pairing and KV splitting are alternatives");
    constexpr int NT = 64 * NWAVE;
    constexpr int BLOCK_M = NWAVE * QT * 16;
    constexpr int CH = DP / 8;            // 16-byte chunks per (padded) row
    constexpr int STRIDE = DP * 2 + 32;   // LDS row stride in bytes (see header comment)
    constexpr int DS = DP / 32;           // 32-wide d steps of the QK^T contraction
    constexpr int DT = DTO;               // 16-wide d tiles of the output (D = 72 / 80 need 5 of the 6 that DP = 96 spans: Hiera's global blocks, the ViT's heads)
    constexpr int LOADS = (KT * CH + NT - 1) / NT;   // the last pass may cover only part of the threads (D = 96 with 8 waves: 768 chunks over 512 threads)
    constexpr bool EVEN = (KT * CH) % NT == 0;
    constexpr int RING = (LOADS <= 2) ? 3 : (LOADS <= 3) ? 2 : 1;  // K/V register slots in flight (8 * LOADS registers each)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + KT * STRIDE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;

    // 1-D grid, remapped so that CONSECUTIVE logical ids share an XCD (workgroups go round-robin over the 8 XCDs), decoded head-fastest: all heads of a
    // (query block, segment) run on one XCD.  A head's slice of a packed qkv row is 144-256 B, so with the plain (x, head, segment) order -- head h of
    // every window on XCD h % 8 -- every XCD's L2 fetched every cache line of q, k and v for its one head (up to 8x the fabric reads), and the 7 query
    // heads of a GQA group each pulled their own copy of the group's K / V.
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int hq = (int)(lid % (unsigned)p.Hq);
    const int bx = (int)((lid / (unsigned)p.Hq) % (unsigned)p.gx);
    const int seg = (int)(lid / ((unsigned)p.Hq * (unsigned)p.gx));
    const int hk = hq / (p.Hq / p.Hkv);
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks = p.cu_k[seg], Lk = p.cu_k[seg + 1] - ks;
    const int shift = Lk - Lq;  // causal: key j visible to query i iff j <= i + shift
    // Causal launches pair q-block i with q-block n-1-i in one workgroup: every workgroup then walks the same number of
    // key tiles (a plain grid leaves the chip to the few longest rows at the end: 34 tiles vs 2 at S = 2112).
    const int nqb = (Lq + BLOCK_M - 1) / BLOCK_M;
    const int sp = SPLIT ? bx % p.nsplit : 0;   // this workgroup's slice of the key range
    int qb_first = SPLIT ? bx / p.nsplit : bx, qb_second = -1;
    if constexpr (PAIR) {
        qb_first = nqb - 1 - bx;           // the long one first
        qb_second = bx;
        if (qb_second > qb_first) return;
        if (qb_second == qb_first) qb_second = -1;
    } else if (qb_first >= nqb) {
        return;
    }
    for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
    const int qbi = pass == 0 ? qb_first : qb_second;
    if (qbi < 0) break;
    if (pass == 1) __syncthreads();  // the previous block's last tile is still being read from LDS
    const int qb0 = qbi * BLOCK_M;

    // ---- Q fragments (B operand): lane (c, g) holds Q[q = c][d = 32*ds + 8g .. +7]
    bf16x8 qf[QT][DS];
    const int qw0 = qb0 + wid * (QT * 16);
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = qw0 + t * 16 + c;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            const int d = ds * 32 + g * 8;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (qi < Lq && d < p.D) {
                const unsigned short* qrow = p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh;
                z = *(const u32x4*)(qrow + d);
                if constexpr (ROPE) z = rope_chunk(z, qrow, d, p.D, p.rope_cos + (long)(qs + qi) * p.D, p.rope_sin + (long)(qs + qi) * p.D);
            }
            qf[t][ds] = __builtin_bit_cast(bf16x8, z);
        }
    }

    f32x4 oacc[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int d = 0; d < DT; ++d) oacc[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { m_run[t] = -INFINITY; l_run[t] = 0.f; }

    // ---- KV range for this q block
    int kv_end = Lk;
    if (p.causal) kv_end = min(Lk, qb0 + BLOCK_M + shift);  // last visible key + 1 (for the block's last row)
    if (kv_end < 0) kv_end = 0;
    int kt_begin = 0, ntiles = (kv_end + KT - 1) / KT;
    if constexpr (SPLIT) {
        const int per = (ntiles + p.nsplit - 1) / p.nsplit;
        kt_begin = min(sp * per, ntiles);
        ntiles = min(kt_begin + per, ntiles);   // tiles [kt_begin, ntiles)
    }

    // K/V tiles travel HBM -> registers -> LDS.  A ring of RING register slots keeps RING tiles in flight: with one slot the
    // loop ran at one global-load latency per 64-key tile (~3 us for a 34-tile causal row at S = 2112, 7 % MFMA use).
    u32x4 kreg[RING][LOADS], vreg[RING][LOADS];
    // per-lane element offsets of this thread's chunks in tile 0 (32-bit: K/V of one call are < 2^31 elements from the
    // segment/head base); the tile index only adds kt * KT * stride
    const unsigned short* kbase = p.k + (long)ks * p.k_st + (long)hk * p.k_sh;
    const unsigned short* vbase = p.v + (long)ks * p.v_st + (long)hk * p.v_sh;
    int koff0[LOADS], voff0[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / CH, ch = idx % CH;
        koff0[i] = (int)(r * p.k_st + ch * 8);
        voff0[i] = (int)(r * p.v_st + ch * 8);
    }
    auto load_tile = [&](auto SLOT, int kt) {
        constexpr int slot = decltype(SLOT)::value;
        const unsigned short* kt_k = kbase + (long)kt * KT * p.k_st;
        const unsigned short* kt_v = vbase + (long)kt * KT * p.v_st;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / CH, ch = idx % CH;
            const int key = kt * KT + r;
            u32x4 zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
            if ((EVEN || idx < KT * CH) && key < Lk && ch * 8 < p.D) {
                zk = *(const u32x4*)(kt_k + koff0[i]);
                zv = *(const u32x4*)(kt_v + voff0[i]);
                if constexpr (ROPE) if (p.rope_kcos) zk = rope_chunk(zk, kt_k + koff0[i] - ch * 8, ch * 8, p.D, p.rope_kcos + (long)(ks + key) * p.D, p.rope_ksin + (long)(ks + key) * p.D);
            }
            kreg[slot][i] = zk;
            vreg[slot][i] = zv;
        }
    };
    auto store_tile = [&](auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / CH, ch = idx % CH;
            if (EVEN || idx < KT * CH) {
                *(u32x4*)(Ks + r * STRIDE + ch * 16) = kreg[slot][i];
                *(u32x4*)(Vs + r * STRIDE + ch * 16) = vreg[slot][i];
            }
        }
    };

    auto tile_body = [&](auto SLOT, int kt) {
        __syncthreads();  // everyone finished reading the previous tile
        store_tile(SLOT);
        __syncthreads();
        if (kt + RING < ntiles) load_tile(SLOT, kt + RING);  // refill the slot just emptied: RING tiles stay in flight

        // ---- S^T tiles: s[t][j] = keys 16j..16j+15 (lane holds keys 16j + 4g + r) x query c of q-tile t
        f32x4 s[QT][KT / 16];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int j = 0; j < KT / 16; ++j) s[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KT / 16; ++j) {
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                bf16x8 kf = *(const bf16x8*)(Ks + (j * 16 + c) * STRIDE + ds * 64 + g * 16);
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ds], s[t][j], 0, 0, 0);
            }
        }

        // ---- scale, mask, online softmax (per lane: one query per q-tile).  Interior tiles take the mask-free body.
        const int k0 = kt * KT;
        const bool need_mask = (k0 + KT > Lk) || (p.causal && (k0 + KT - 1 > qb0 + shift)) || (p.bq_shift >= 0);
        bf16x8 pf[QT][KT / 32];
        auto softmax_tile = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int qi = qw0 + t * 16 + c;
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < KT / 16; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = s[t][j][r] * p.scale_log2;  // scores are scaled first, as the reference does (folding the scale into the exp FMA saves
                                                                // 16 multiplies per tile and measurably perturbs bf16 gradient parity)
                        if constexpr (MASKED) {
                            const int key = k0 + j * 16 + 4 * g + r;
                            bool ok = (key < Lk) && (!p.causal || key <= qi + shift);
                            if (p.bq_shift >= 0) ok = ok && ((qi >> p.bq_shift) == (key >> p.bk_shift));
                            x = ok ? x : -INFINITY;
                        }
                        s[t][j][r] = x;
                        mx = fmaxf(mx, x);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[t], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_use);  // m_run = -inf -> 0
                m_run[t] = m_new;
                float ps = 0.f;
#pragma unroll
                for (int j = 0; j < KT / 16; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e = __builtin_amdgcn_exp2f(s[t][j][r] - m_use);   // raw v_exp_f32: arguments <= 0, flush of tiny values is harmless
                        s[t][j][r] = e;
                        ps += e;
                    }
                l_run[t] = l_run[t] * alpha + ps;
                // rescale O only when some lane's running max moved (alpha == 1 exactly otherwise): the accumulators stay in
                // the MFMA accumulator file on the common path
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) oacc[t][d] *= alpha;
                }
                // P^T B-operand fragments: 32-key step ss uses s[t][2ss] (elements 0..3) and s[t][2ss+1] (4..7)
#pragma unroll
                for (int ss = 0; ss < KT / 32; ++ss) {
                    u32x4 pk;
                    pk[0] = pack_bf2(s[t][2 * ss][0], s[t][2 * ss][1]);
                    pk[1] = pack_bf2(s[t][2 * ss][2], s[t][2 * ss][3]);
                    pk[2] = pack_bf2(s[t][2 * ss + 1][0], s[t][2 * ss + 1][1]);
                    pk[3] = pack_bf2(s[t][2 * ss + 1][2], s[t][2 * ss + 1][3]);
                    pf[t][ss] = __builtin_bit_cast(bf16x8, pk);
                }
            }
        };
        if (need_mask) softmax_tile(std::true_type{});
        else softmax_tile(std::false_type{});

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
            for (int ss = 0; ss < KT / 32; ++ss) {
                bf16x8 vf;
                if constexpr (USE_TR) {
                    // lane i of group g supplies row (key) i>>2, columns 4*(i&3)..+3 of the 4x16 block
                    const char* a0 = Vs + (ss * 32 + 4 * g + (c >> 2)) * STRIDE + (d * 16 + 4 * (c & 3)) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(a0 + 16 * STRIDE));
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int key = ss * 32 + 16 * (e >> 2) + 4 * g + (e & 3);
                        unsigned short u = *(const unsigned short*)(Vs + key * STRIDE + (d * 16 + c) * 2);
                        vf[e] = __builtin_bit_cast(__bf16, u);
                    }
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) oacc[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][ss], oacc[t][d], 0, 0, 0);
            }
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (kt_begin < ntiles) load_tile(S0{}, kt_begin);
    if constexpr (RING > 1) { if (kt_begin + 1 < ntiles) load_tile(S1{}, kt_begin + 1); }
    if constexpr (RING > 2) { if (kt_begin + 2 < ntiles) load_tile(std::integral_constant<int, 2>{}, kt_begin + 2); }
    for (int kt = kt_begin; kt < ntiles; kt += RING) {
        tile_body(S0{}, kt);
        if constexpr (RING > 1) { if (kt + 1 < ntiles) tile_body(S1{}, kt + 1); }
        if constexpr (RING > 2) { if (kt + 2 < ntiles) tile_body(std::integral_constant<int, 2>{}, kt + 2); }
    }

    // ---- finalize: lane holds query c, output dims 16d + 4g + r
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = (l > 0.f) ? 1.f / l : 0.f;
        const int qi = qw0 + t * 16 + c;
        if constexpr (SPLIT) {
            if (qi < Lq) {
                float* orow = p.split_o + (((long)sp * p.total_q + qs + qi) * p.Hq + hq) * p.D;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int dd = d * 16 + 4 * g;
                    if (dd < p.D) *(f32x4*)(orow + dd) = oacc[t][d] * inv;
                }
                if (g == 0) p.split_lse[((long)sp * p.Hq + hq) * p.total_q + qs + qi] = (l > 0.f) ? m_run[t] + log2f(l) : -INFINITY;
            }
            continue;
        }
        if (qi < Lq) {
            unsigned short* orow = p.o + (long)(qs + qi) * p.o_st + (long)hq * p.o_sh;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int dd = d * 16 + 4 * g;
                if (dd < p.D) {
                    u32x2 pk;
                    pk[0] = pack_bf2(oacc[t][d][0] * inv, oacc[t][d][1] * inv);
                    pk[1] = pack_bf2(oacc[t][d][2] * inv, oacc[t][d][3] * inv);
                    *(u32x2*)(orow + dd) = pk;
                }
            }
            if (p.lse && g == 0) {
                // natural-log LSE of the scaled scores
                p.lse[(long)hq * p.total_q + qs + qi] = (l > 0.f) ? (m_run[t] * 0.6931471805599453f + logf(l)) : -INFINITY;
            }
        }
    }
    }  // pass
}

// ---- Window attention with the WHOLE key / value range of a segment resident in LDS (non-causal, <= 256 keys, D <= DP <= 96): Hiera's 64- / 256-token
// windows (reference model/sam2.py:986-1033 inside window_partition), 16-token windows packed block-diagonally, the ViT's 64-token windows.  The
// pipelined kernel above walks 64-key tiles with two barriers each and a register ring -- for four tiles that is all prologue: 18 us per 256 x 256
// window and head, one 8-wave workgroup per CU (256 VGPRs).  Here K and V of the segment are staged ONCE (every load of the workgroup in flight at the
// same time, one barrier), then each wave walks the key tiles on its own 16 query rows with no further synchronisation, so the MFMA, softmax and
// LDS phases of the 16 waves of the co-resident workgroups interleave freely.  16 query rows per wave, 8 waves per workgroup (4 for windows of <= 64
// queries), <= 128 VGPRs.
// LDS rows are D padded to an ODD number of 16-byte chunks (72 -> 144 B, 80 -> 176 B): the 16 rows a quarter-wave reads then start in 16 different
// 4-bank groups, and a 256-key window of 72-d heads takes 2 x 36 KiB, so two workgroups share a CU.  Fragment reads of the padded d range (D..DP) run
// into the next row: finite numbers that meet Q's zero padding (scores) or land in output columns >= D that are never stored.
template <int DP, int NW, int QT, bool ROPE = false, int DTO = DP / 16>   // DTO: 16-column tiles of the OUTPUT (D = 72 needs 5 of the 6 that DP = 96 spans)
__global__ __launch_bounds__(64 * NW, (QT == 1 ? 16 : (NW == 8 ? 32 : 8)) / NW) void attn_win_kernel(AttnArgs p) {
    constexpr int NT = 64 * NW, BLOCK_M = 16 * QT * NW, DS = DP / 32, DT = DTO;
    constexpr int LDMAX = (256 * (DP / 8 + 1) + 511) / 512;   // 16-byte chunks per thread, operand and staging pass (8 waves: the largest segment in one pass)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int hq = (int)(lid % (unsigned)p.Hq);
    const int bx = (int)((lid / (unsigned)p.Hq) % (unsigned)p.gx);
    const int seg = (int)(lid / ((unsigned)p.Hq * (unsigned)p.gx));
    const int hk = hq / (p.Hq / p.Hkv);
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks = p.cu_k[seg], Lk = min(p.cu_k[seg + 1] - ks, 256);
    const int qb0 = bx * BLOCK_M;
    if (qb0 >= Lq) return;

    const int chd = (p.D + 7) >> 3;                 // chunks of a row that hold data
    const int chr = (((p.D * 2 + 15) >> 4) | 1);    // chunks per LDS row (odd)
    const int RS = chr * 16;
    const int nrows = (Lk + 63) & ~63;
    char* Ks = smem;
    char* Vs = smem + nrows * RS;

    // ---- Q fragments first (B operand): lane (c, g) holds Q[q = c][d = 32 ds + 8 g .. + 7] of each of the wave's QT 16-row tiles
    const int qw0 = qb0 + wid * (16 * QT);
    bf16x8 qf[QT][DS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = qw0 + t * 16 + c;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            const int d = ds * 32 + g * 8;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (qi < Lq && d < p.D) {
                const unsigned short* qrow = p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh;
                z = *(const u32x4*)(qrow + d);
                if constexpr (ROPE) z = rope_chunk(z, qrow, d, p.D, p.rope_cos + (long)(qs + qi) * p.D, p.rope_sin + (long)(qs + qi) * p.D);
            }
            qf[t][ds] = __builtin_bit_cast(bf16x8, z);
        }
    }
    // ---- stage all of K and V: every load of a pass is in flight before its first store
    {
        const unsigned short* kbase = p.k + (long)ks * p.k_st + (long)hk * p.k_sh;
        const unsigned short* vbase = p.v + (long)ks * p.v_st + (long)hk * p.v_sh;
        const int total = nrows * chr;
        for (int b0 = 0; b0 < total; b0 += NT * LDMAX) {   // one pass for 8 waves x 256 keys; 4-wave workgroups with a 256-key range take two
            u32x4 kreg[LDMAX], vreg[LDMAX];
#pragma unroll
            for (int i = 0; i < LDMAX; ++i) {
                const int idx = b0 + tid + i * NT;
                const int r = idx / chr, ch = idx - r * chr;
                u32x4 zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
                if (idx < total && r < Lk && ch < chd) {
                    zk = *(const u32x4*)(kbase + (long)r * p.k_st + ch * 8);
                    zv = *(const u32x4*)(vbase + (long)r * p.v_st + ch * 8);
                    if constexpr (ROPE)   // every key is staged exactly once per head: rotating it here replaces the stand-alone rope pass
                        if (p.rope_kcos) zk = rope_chunk(zk, kbase + (long)r * p.k_st, ch * 8, p.D, p.rope_kcos + (long)(ks + r) * p.D, p.rope_ksin + (long)(ks + r) * p.D);
                }
                kreg[i] = zk;
                vreg[i] = zv;
            }
#pragma unroll
            for (int i = 0; i < LDMAX; ++i) {
                const int idx = b0 + tid + i * NT;
                if (idx < total) {
                    *(u32x4*)(Ks + idx * 16) = kreg[i];     // rows are chr chunks wide: chunk idx lies at byte idx * 16
                    *(u32x4*)(Vs + idx * 16) = vreg[i];
                }
            }
        }
        if (tid < 16) *(u32x4*)(Vs + nrows * RS + tid * 16) = u32x4{0u, 0u, 0u, 0u};   // slack the last rows' padded-d reads run into
    }
    __syncthreads();
    if (qw0 >= Lq) return;   // no barrier below

    f32x4 oacc[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int d = 0; d < DT; ++d) oacc[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
    }
    const int ntiles = nrows >> 6;
    for (int kt = 0; kt < ntiles; ++kt) {
        const char* Kt = Ks + kt * 64 * RS;
        const char* Vt = Vs + kt * 64 * RS;
        f32x4 s[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                const bf16x8 kf = *(const bf16x8*)(Kt + (j * 16 + c) * RS + ds * 64 + g * 16);
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ds], s[t][j], 0, 0, 0);
            }
        const int k0 = kt * 64;
        const bool need_mask = (k0 + 64 > Lk) || (p.bq_shift >= 0);
        bf16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int qi = qw0 + t * 16 + c;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = s[t][j][r] * p.scale_log2;
                    if (need_mask) {
                        const int key = k0 + j * 16 + 4 * g + r;
                        bool ok = key < Lk;
                        if (p.bq_shift >= 0) ok = ok && ((qi >> p.bq_shift) == (key >> p.bk_shift));
                        x = ok ? x : -INFINITY;
                    }
                    s[t][j][r] = x;
                    mx = fmaxf(mx, x);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[t], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_use);
            m_run[t] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[t][j][r] - m_use);
                    s[t][j][r] = e;
                    ps += e;
                }
            l_run[t] = l_run[t] * alpha + ps;
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[t][d] *= alpha;
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                u32x4 pk;
                pk[0] = pack_bf2(s[t][2 * ss][0], s[t][2 * ss][1]);
                pk[1] = pack_bf2(s[t][2 * ss][2], s[t][2 * ss][3]);
                pk[2] = pack_bf2(s[t][2 * ss + 1][0], s[t][2 * ss + 1][1]);
                pk[3] = pack_bf2(s[t][2 * ss + 1][2], s[t][2 * ss + 1][3]);
                pf[t][ss] = __builtin_bit_cast(bf16x8, pk);
            }
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const char* a0 = Vt + (ss * 32 + 4 * g + (c >> 2)) * RS + (d * 16 + 4 * (c & 3)) * 2;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 16 * RS));
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
                for (int t = 0; t < QT; ++t) oacc[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][ss], oacc[t][d], 0, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = (l > 0.f) ? 1.f / l : 0.f;
        const int qi = qw0 + t * 16 + c;
        if (qi < Lq) {
            unsigned short* orow = p.o + (long)(qs + qi) * p.o_st + (long)hq * p.o_sh;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int dd = d * 16 + 4 * g;
                if (dd < p.D) {
                    u32x2 pk;
                    pk[0] = pack_bf2(oacc[t][d][0] * inv, oacc[t][d][1] * inv);
                    pk[1] = pack_bf2(oacc[t][d][2] * inv, oacc[t][d][3] * inv);
                    *(u32x2*)(orow + dd) = pk;
                }
            }
            if (p.lse && g == 0) p.lse[(long)hq * p.total_q + qs + qi] = (l > 0.f) ? (m_run[t] * 0.6931471805599453f + logf(l)) : -INFINITY;
        }
    }
}

// out = sum_i w_i o_i / sum_i w_i with w_i = 2^(lse2_i - max): one wave per (query row, head), 4 floats of D per lane per pass
__global__ __launch_bounds__(256) void attn_split_combine_kernel(AttnArgs p) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (token, head)
    const int lane = threadIdx.x & 63;
    if (row >= p.total_q * p.Hq) return;
    const long tq = row / p.Hq;
    const int hq = (int)(row % p.Hq);
    float m = -INFINITY;
    for (int i = 0; i < p.nsplit; ++i) m = fmaxf(m, p.split_lse[((long)i * p.Hq + hq) * p.total_q + tq]);
    float wsum = 0.f, w[8];
    for (int i = 0; i < p.nsplit; ++i) {
        const float l2 = p.split_lse[((long)i * p.Hq + hq) * p.total_q + tq];
        w[i] = (m == -INFINITY) ? 0.f : exp2f(l2 - m);
        wsum += w[i];
    }
    const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
    for (int dd = lane * 4; dd < p.D; dd += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < p.nsplit; ++i) acc += *(const f32x4*)(p.split_o + (((long)i * p.total_q + tq) * p.Hq + hq) * p.D + dd) * w[i];
        u32x2 pk;
        pk[0] = pack_bf2(acc[0] * inv, acc[1] * inv);
        pk[1] = pack_bf2(acc[2] * inv, acc[3] * inv);
        *(u32x2*)(p.o + tq * p.o_st + (long)hq * p.o_sh + dd) = pk;
    }
    if (p.lse && lane == 0) p.lse[(long)hq * p.total_q + tq] = wsum > 0.f ? (m + log2f(wsum)) * 0.6931471805599453f : -INFINITY;
}

static bool g_attn_full_tiles = false;  // A/B switch (impl bit 16): keep all DP / 16 output tiles where D <= 80 would need five of six

template <int DP, int QT, int NWAVE, bool USE_TR, bool PAIR, int KT = KV_TILE, int DTO = DP / 16>
static int launch_attn_p(const AttnArgs& a, int nseg, unsigned gx, hipStream_t st) {
    if constexpr (DP == 96 && DTO == 6 && !PAIR) {
        if (a.D <= 80 && !g_attn_full_tiles) return launch_attn_p<DP, QT, NWAVE, USE_TR, PAIR, KT, 5>(a, nseg, gx, st);
    }
    constexpr int LDS = 2 * KT * (DP * 2 + 32);
    auto kern = attn_fwd_kernel<DP, QT, NWAVE, USE_TR, PAIR, false, false, KT, DTO>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "attn")) return rc;
    AttnArgs b = a;
    b.gx = (int)gx;
    hipLaunchKernelGGL(kern, dim3(gx * (unsigned)a.Hq * (unsigned)nseg), dim3(64 * NWAVE), LDS, st, b);
    RGA3_CHECK_LAUNCH("attn_fwd_kernel");
    return 0;
}

template <int DP, int QT, int NWAVE, bool USE_TR>
static int launch_attn(const AttnArgs& a, int nseg, int max_q, hipStream_t st) {
    constexpr int BLOCK_M = NWAVE * QT * 16;
    const unsigned nqb = (unsigned)cdiv(max_q, BLOCK_M);
    // the paired-q-block form only for the long causal rows of the decoder (D = 64 / 128, 8 waves): elsewhere it would only
    // cost registers
    if constexpr (NWAVE == 8) {
        if (a.causal && nqb >= 4) {
            // the decoder's causal rows walk 128-key tiles: half the barriers, softmax rescales and tile bookkeeping per key (same 205 VGPRs: one register slot
            // of four loads instead of three of two) -- 66.3 -> 61.5 us at S = 2112, 219.5 -> 203.6 us at S = 4160 on one box
            if constexpr (DP == 128 && QT == 1) return launch_attn_p<DP, QT, NWAVE, USE_TR, true, 128>(a, nseg, (nqb + 1) / 2, st);
            else return launch_attn_p<DP, QT, NWAVE, USE_TR, true>(a, nseg, (nqb + 1) / 2, st);
        }
    }
    if (a.nsplit > 1) {
        constexpr int LDS = 2 * KV_TILE * (DP * 2 + 32);
        auto kern = attn_fwd_kernel<DP, QT, NWAVE, USE_TR, false, true>;
        static LdsGrant lds_grant;
        if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "attn")) return rc;
        AttnArgs b = a;
        b.gx = (int)(nqb * (unsigned)a.nsplit);
        hipLaunchKernelGGL(kern, dim3(nqb * (unsigned)a.nsplit * (unsigned)a.Hq * (unsigned)nseg), dim3(64 * NWAVE), LDS, st, b);
        RGA3_CHECK_LAUNCH("attn_fwd_kernel<split>");
        hipLaunchKernelGGL(attn_split_combine_kernel, dim3((unsigned)cdiv(a.total_q * a.Hq, 4)), dim3(256), 0, st, a);
        RGA3_CHECK_LAUNCH("attn_split_combine_kernel");
        return 0;
    }
    // 128-key tiles also for the non-paired long rows at D = 128 (3 280 -> 3 159 us at B16 H64 L2048); at D = 64 they lose (1 633 -> 2 000 us: the 8 loads of a
    // tile leave one register slot instead of three)
    if constexpr (DP == 128 && QT == 1 && NWAVE == 8) return launch_attn_p<DP, QT, NWAVE, USE_TR, false, 128>(a, nseg, nqb, st);
    else return launch_attn_p<DP, QT, NWAVE, USE_TR, false>(a, nseg, nqb, st);
}

// RoPE-while-loading variant: windows of <= 64 queries (one query block of 4 waves x 16 rows per segment and head)
template <int DP>
static int launch_rope_win(const AttnArgs& a, int nseg, hipStream_t st) {
    constexpr int LDS = 2 * KV_TILE * (DP * 2 + 32);
    auto kern = attn_fwd_kernel<DP, 1, 4, true, false, false, true>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, LDS, lds_grant, "attn")) return rc;
    AttnArgs b = a;
    b.gx = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)a.Hq * (unsigned)nseg), dim3(256), LDS, st, b);
    RGA3_CHECK_LAUNCH("attn_fwd_kernel<rope>");
    return 0;
}

template <int DP, int NW, int QT, bool ROPE = false, int DTO = DP / 16>
static int launch_win(const AttnArgs& a, int nseg, int max_q, int max_k, hipStream_t st) {
    const int chr = (((a.D * 2 + 15) >> 4) | 1);
    const int nrows = (max_k + 63) & ~63;
    const int lds = 2 * nrows * chr * 16 + 256;
    auto kern = attn_win_kernel<DP, NW, QT, ROPE, DTO>;
    static LdsGrant lds_grant;   // per device, grows monotonically; a racing second setter only repeats the call
    if (int rc = grant_dyn_lds((const void*)kern, lds, lds_grant, "attn")) return rc;
    AttnArgs b = a;
    b.gx = (int)cdiv(max_q, 16 * QT * NW);
    hipLaunchKernelGGL(kern, dim3((unsigned)b.gx * (unsigned)a.Hq * (unsigned)nseg), dim3(64 * NW), lds, st, b);
    RGA3_CHECK_LAUNCH("attn_win_kernel");
    return 0;
}

static int g_attn_variant = 0;  // 0: auto, 1: force 4 waves x QT=2, 2: force 8 waves x QT=1 (benchmark switch, read-only after init)

template <int DP, bool USE_TR>
static int launch_dp(const AttnArgs& a, int nseg, int max_q, hipStream_t st) {
    if constexpr (DP == 128 || DP == 64) {
        // long sequences: 8 waves x 16 query rows keeps the register footprint near 110 VGPRs (4 waves/SIMD) instead of
        // one 300-register wave per SIMD
        // (measured, round 2: 64-row paired query blocks on 4 waves -- twice the workgroups, two per CU at S = 2112 -- are slower: 73.8 vs 65.9 us)
        if (max_q > 64 && g_attn_variant != 1) return launch_attn<DP, 1, 8, USE_TR>(a, nseg, max_q, st);
    }
    if constexpr (DP == 96) {
        // (8 waves x 16 rows for the 4096-token global blocks: 1 428 vs 1 150 us -- the 32-row waves halve the K / V fragment reads per query row)
        // a whole 256-token window (Hiera stage 3) per workgroup: K / V staged once instead of once per 128-row half
        if (!a.causal && max_q >= 256 && max_q % 256 == 0 && g_attn_variant != 1) return launch_attn<DP, 2, 8, USE_TR>(a, nseg, max_q, st);
    }
    if constexpr (DP >= 256) {
        return launch_attn<DP, 1, 4, USE_TR>(a, nseg, max_q, st);
    } else {
        if (max_q <= 64) return launch_attn<DP, 1, 4, USE_TR>(a, nseg, max_q, st);
        return launch_attn<DP, 2, 4, USE_TR>(a, nseg, max_q, st);
    }
}

template <bool USE_TR>
static int launch_any(const AttnArgs& a, int nseg, int max_q, hipStream_t st) {
    const int D = a.D;
    if (D <= 32) return launch_dp<32, USE_TR>(a, nseg, max_q, st);
    if (D <= 64) return launch_dp<64, USE_TR>(a, nseg, max_q, st);
    if (D <= 96) return launch_dp<96, USE_TR>(a, nseg, max_q, st);
    if (D <= 128) return launch_dp<128, USE_TR>(a, nseg, max_q, st);
    return launch_dp<256, USE_TR>(a, nseg, max_q, st);
}

}  // namespace rga3

using namespace rga3;

extern "C" int rga3_attn_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                    const int32_t* cu_q, const int32_t* cu_k, int nseg, int max_q, int64_t total_q,
                                    int Hq, int Hkv, int D, int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh,
                                    int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh, float scale, int causal,
                                    int impl, float* split_ws, int64_t split_ws_elems, int max_k, int block_q, int block_k, void* stream) {
    RGA3_CHECK_ARG(q && k && v && o && cu_q && cu_k, "attn: null pointer");
    RGA3_CHECK_ARG(nseg > 0 && max_q > 0 && total_q > 0, "attn: nseg=%d max_q=%d total_q=%ld", nseg, max_q, (long)total_q);
    RGA3_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attn: Hq=%d Hkv=%d", Hq, Hkv);
    RGA3_CHECK_ARG(D >= 8 && D <= 256 && D % 8 == 0, "attn: head dim %d unsupported (need multiple of 8, <= 256)", D);
    RGA3_CHECK_ARG(q_st % 8 == 0 && q_sh % 8 == 0 && k_st % 8 == 0 && k_sh % 8 == 0 && v_st % 8 == 0 && v_sh % 8 == 0,
                   "attn: q/k/v strides must be multiples of 8 elements");
    RGA3_CHECK_ARG(o_st % 4 == 0 && o_sh % 4 == 0, "attn: o strides must be multiples of 4 elements");
    RGA3_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && (((uintptr_t)o) & 7) == 0,
                   "attn: pointer alignment");
    RGA3_CHECK_ARG(nseg <= 65535 && Hq <= 65535, "attn: grid dims too large");
    RGA3_CHECK_ARG(k_st < (1 << 24) && v_st < (1 << 24), "attn: k/v row stride too large for 32-bit tile offsets");
    RGA3_CHECK_ARG(impl >= 0 && impl <= 31, "attn: impl %d", impl);
    g_attn_full_tiles = (impl & 16) != 0;
    const bool win_q16 = (impl & 8) != 0;       // A/B switch: 256-query windows on the 16-rows-per-wave form (two workgroups per window and head)
    g_attn_variant = (impl & 2) ? 1 : 0;
    const bool no_causal32 = (impl & 4) != 0;   // A/B and parity switch: keep the long causal rows on the general kernel
    impl &= 1;
    AttnArgs a;
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.o = (unsigned short*)o; a.lse = lse; a.cu_q = cu_q; a.cu_k = cu_k;
    a.q_st = q_st; a.q_sh = q_sh; a.k_st = k_st; a.k_sh = k_sh; a.v_st = v_st; a.v_sh = v_sh; a.o_st = o_st; a.o_sh = o_sh;
    a.Hq = Hq; a.Hkv = Hkv; a.D = D;
    a.total_q = total_q;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.causal = causal;
    a.rope_cos = a.rope_sin = a.rope_kcos = a.rope_ksin = nullptr;
    // split the key range when the grid would leave most CUs idle: <= 8 slices, >= 8 key tiles each, workspace permitting
    RGA3_CHECK_ARG((block_q == 0) == (block_k == 0) && block_q >= 0 && (block_q & (block_q - 1)) == 0 && (block_k & (block_k - 1)) == 0,
                   "attn: block_q / block_k must both be 0 or powers of two (got %d, %d)", block_q, block_k);
    RGA3_CHECK_ARG(block_q == 0 || !causal, "attn: block-diagonal packing is for non-causal windows");
    a.bq_shift = a.bk_shift = -1;
    if (block_q > 0) { a.bq_shift = __builtin_ctz((unsigned)block_q); a.bk_shift = __builtin_ctz((unsigned)block_k); }
    a.split_o = nullptr; a.split_lse = nullptr; a.nsplit = 1; a.gx = 1;
    if (split_ws && !causal && block_q == 0 && D % 4 == 0 && max_k >= 1024) {
        const long wgs = (long)cdiv(max_q, 64) * Hq * nseg;
        int ns = (int)(256 / (wgs > 0 ? wgs : 1));
        if (ns > 8) ns = 8;
        if (ns > max_k / (8 * KV_TILE)) ns = max_k / (8 * KV_TILE);
        const long need = (long)ns * total_q * Hq * (D + 1);
        if (ns >= 2 && need <= split_ws_elems) {
            a.nsplit = ns;
            a.split_o = split_ws;
            a.split_lse = split_ws + (long)ns * total_q * Hq * D;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // long causal rows at D = 128 (the decoder's prefill / training rows): 32-row waves balanced over the key range (attn_causal32.hip)
    // (its output rows leave by 16-byte stores: an 8-byte-aligned `o` view, legal for the general kernel, stays there)
    const bool o16 = (((uintptr_t)o) & 15) == 0 && o_st % 8 == 0 && o_sh % 8 == 0;
    if (impl == 0 && g_attn_variant == 0 && !no_causal32 && o16 && causal && D == 128 && max_q >= 256 && block_q == 0) return launch_causal32(a, nseg, max_q, st);
    // whole-segment-in-LDS window kernel: non-causal, key range known and <= 256, D <= 96, 16-byte rows (impl bit 1 keeps the pipelined kernel: A/B)
    if (impl == 0 && g_attn_variant == 0 && !causal && a.nsplit == 1 && max_k > 0 && max_k <= 256 && D <= 96 && D % 8 == 0 &&
        (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && q_st % 8 == 0 && q_sh % 8 == 0 && k_st % 8 == 0 && k_sh % 8 == 0 && v_st % 8 == 0 && v_sh % 8 == 0) {
        // (4 waves x 32 rows per 256-token window -- half the fragment reads per row, one wave fewer per SIMD -- measured slower: 129 vs 112 us)
        // (64 < D <= 80 -- Hiera's 72, the ViT's 80 -- spans 5 of the 6 sixteen-column output tiles of DP = 96: a sixth of the P V work and of the V fragment reads)
        if (max_q <= 64) return (D <= 64) ? launch_win<64, 4, 1>(a, nseg, max_q, max_k, st) : (D <= 80) ? launch_win<96, 4, 1, false, 5>(a, nseg, max_q, max_k, st) : launch_win<96, 4, 1>(a, nseg, max_q, max_k, st);
        // 256-query windows at D > 64 (Hiera stage 3: 16 x 16 tokens, 8 heads x 72): ONE 8-wave workgroup per window and head, 32 query rows per wave -- K / V of the
        // window are fetched and staged once instead of twice, and every K / V fragment read from LDS feeds two MFMAs (the 16-row form is bound by the LDS array:
        // 96 KiB of fragment reads per wave against 1 536 MFMA cycles, four waves per SIMD): 118 -> 93 us per stage-3 block of 16 frames.  (A PERSISTENT form --
        // one workgroup per CU walking (window, head) items with the next item's K | V images arriving by LDS-DMA into a second LDS buffer while this one is
        // multiplied, all loads and the counted wait in one inline-asm statement -- was built, bit-checked and measured at 122 us: with one workgroup per CU an
        // item's 72 KiB take ~13 us to arrive, two independent workgroups per CU keep twice the bytes in flight; profiles/r06_hiera_attn_probe.log)
        if (D > 64 && max_q > 128 && !win_q16) return (D <= 80) ? launch_win<96, 8, 2, false, 5>(a, nseg, max_q, max_k, st) : launch_win<96, 8, 2>(a, nseg, max_q, max_k, st);
        return (D <= 64) ? launch_win<64, 8, 1>(a, nseg, max_q, max_k, st) : (D <= 80) ? launch_win<96, 8, 1, false, 5>(a, nseg, max_q, max_k, st) : launch_win<96, 8, 1>(a, nseg, max_q, max_k, st);
    }
    if (impl == 0) return launch_any<true>(a, nseg, max_q, st);
    return launch_any<false>(a, nseg, max_q, st);
}

extern "C" int rga3_attn_varlen_fwd_rope(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* cu_q, const int32_t* cu_k, int nseg,
                                        int max_q, int64_t total_q, int Hq, int Hkv, int D, int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st,
                                        int64_t v_sh, int64_t o_st, int64_t o_sh, float scale, int causal, const float* cos_q, const float* sin_q,
                                        const float* cos_k, const float* sin_k, void* stream) {
    RGA3_CHECK_ARG(q && k && v && o && cu_q && cu_k && cos_q && sin_q && ((cos_k == nullptr) == (sin_k == nullptr)), "attn_varlen_fwd_rope: null pointer");
    // long causal rows at D = 128 (decoder prefill): the paired-block kernel rotates its query fragments on load (keys arrive rotated: cos_k must be null)
    const bool c32 = causal && D == 128 && max_q >= 256 && !cos_k;
    RGA3_CHECK_ARG(nseg > 0 && nseg <= 65535 && max_q > 0 && (max_q <= 64 || c32) && total_q > 0,
                   "attn_varlen_fwd_rope: nseg=%d max_q=%d (windows of <= 64 queries, or causal D = 128 rows of >= 256 queries with rotated keys)", nseg, max_q);
    RGA3_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq <= 65535 && D >= 16 && D <= 128 && D % 16 == 0, "attn_varlen_fwd_rope: Hq=%d Hkv=%d D=%d", Hq, Hkv, D);
    RGA3_CHECK_ARG(q_st % 8 == 0 && q_sh % 8 == 0 && k_st % 8 == 0 && k_sh % 8 == 0 && v_st % 8 == 0 && v_sh % 8 == 0 && o_st % 4 == 0 && o_sh % 4 == 0,
                   "attn_varlen_fwd_rope: strides");
    RGA3_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)cos_q | (uintptr_t)sin_q | (uintptr_t)cos_k | (uintptr_t)sin_k) & 15) == 0 &&
                       (((uintptr_t)o) & 7) == 0, "attn_varlen_fwd_rope: pointer alignment");
    RGA3_CHECK_ARG(k_st < (1 << 24) && v_st < (1 << 24), "attn_varlen_fwd_rope: k/v row stride too large");
    RGA3_CHECK_ARG(!c32 || ((((uintptr_t)o) & 15) == 0 && o_st % 8 == 0 && o_sh % 8 == 0), "attn_varlen_fwd_rope: the causal D = 128 rows write 16-byte pieces: o must be 16-byte aligned, strides multiples of 8");
    AttnArgs a;
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.o = (unsigned short*)o; a.lse = lse; a.cu_q = cu_q; a.cu_k = cu_k;
    a.q_st = q_st; a.q_sh = q_sh; a.k_st = k_st; a.k_sh = k_sh; a.v_st = v_st; a.v_sh = v_sh; a.o_st = o_st; a.o_sh = o_sh;
    a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.total_q = total_q;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.causal = causal;
    a.rope_cos = cos_q; a.rope_sin = sin_q; a.rope_kcos = cos_k; a.rope_ksin = sin_k;
    a.bq_shift = a.bk_shift = -1;
    a.split_o = nullptr; a.split_lse = nullptr; a.nsplit = 1; a.gx = 1;
    hipStream_t st = (hipStream_t)stream;
    if (c32) return launch_causal32(a, nseg, max_q, st);
#ifdef RGA3_AB   // measurement builds only (tools/): the product library has one behaviour
    static const bool old_rope = [] { const char* e = getenv("RGA3_ATTN_ROPE_OLD"); return e && atoi(e) != 0; }();   // A/B switch: the pipelined rope kernel
#else
    constexpr bool old_rope = false;
#endif
    if (!old_rope && !causal && cu_q == cu_k && D <= 96 && D % 8 == 0)   // self-attention windows: key range = query range <= 64
        return (D <= 64) ? launch_win<64, 4, 1, true>(a, nseg, max_q, max_q, st) : (D <= 80) ? launch_win<96, 4, 1, true, 5>(a, nseg, max_q, max_q, st) : launch_win<96, 4, 1, true>(a, nseg, max_q, max_q, st);
    if (D <= 32) return launch_rope_win<32>(a, nseg, st);
    if (D <= 64) return launch_rope_win<64>(a, nseg, st);
    if (D <= 96) return launch_rope_win<96>(a, nseg, st);
    return launch_rope_win<128>(a, nseg, st);
}
