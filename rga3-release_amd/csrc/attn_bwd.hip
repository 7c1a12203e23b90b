// Variable-length fused attention BACKWARD for CDNA4 — companion of attn_fwd.hip (same MFMA layouts, same LDS images).
//
// Flash-style recomputation from Q, K, V, dO and the forward's LSE (never materialises the score matrix).  Two kernels,
// no atomics, bitwise reproducible:
//   * dq kernel   — one workgroup per (q block, q head): recomputes S^T = K.Q^T and dP^T = V.dO^T per KV tile with the
//                   QUERY on the lane (so lse[q], delta[q] are lane scalars), forms dS^T and accumulates
//                   dQ^T += K^T . dS^T  (K^T through ds_read_b64_tr_b16 of the row-major K image).
//   * dkv kernel  — one workgroup per (key block, kv head), looping over the q heads of the GQA group and over Q tiles:
//                   roles swapped (K/V fragments live in registers, Q / dO tiles are staged in LDS), KEY on the lane;
//                   dV^T += dO^T . P  and  dK^T += Q^T . dS  with transposed LDS reads of the dO / Q images.
// delta[q] = rowsum(dO * O) is computed by the dQ kernel for its rows (and left in memory for the dK / dV kernel).
// Call sites replaced: the autograd of flash_attn_varlen_func / SDPA under train_joint.py:534 (model.backward).
#include "common.h"

#include <mutex>
#include <set>

namespace rga3 {

struct AttnBwdArgs {
    const unsigned short *q, *k, *v, *o, *dout;
    unsigned short *dq, *dk, *dv;
    const float* lse;   // [Hq, total_q] natural log
    float* delta;       // [Hq, total_q]
    const int *cu_q, *cu_k;
    long q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh, do_st, do_sh, dq_st, dq_sh, dk_st, dk_sh, dv_st, dv_sh;
    int Hq, Hkv, D;
    long total_q;
    float scale, scale_log2;
    int causal;
    float* dkv_ws;   // GQA split: f32 partial dK | dV, each [group][total_k][Hkv][D]; null = loop the group inside one workgroup
    long total_k;
    int gx, gy;      // logical grid (blocks, heads) of the dq / dkv launches: the hardware grid is 1-D, decoded XCD-aware (xcd_decode3)
};

constexpr int BT = 64;  // tile of the streamed (LDS-staged) dimension

// ---------------------------------------------------------------------------------------------- dQ
template <int DP, int QT, int NWAVE, bool PAIR>
__global__ __launch_bounds__(64 * NWAVE) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    constexpr int NT = 64 * NWAVE;
    constexpr int BLOCK_M = NWAVE * QT * 16;
    constexpr int CH = DP / 8;
    constexpr int STRIDE = DP * 2 + 32;
    constexpr int DS = DP / 32, DT = DP / 16;
    constexpr int LOADS = (BT * CH) / NT;
    static_assert((BT * CH) % NT == 0, "tile chunks must divide evenly over threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + BT * STRIDE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const Wg3 wg = xcd_decode3(p.gx, p.gy);   // heads of one (query block, segment) share an XCD: K / V of a GQA group are fetched once per XCD
    const int seg = wg.z, hq = wg.y;
    const int hk = hq / (p.Hq / p.Hkv);
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks = p.cu_k[seg], Lk = p.cu_k[seg + 1] - ks;
    const int shift = Lk - Lq;
    // causal launches pair query block i with n-1-i (as the forward does): every workgroup walks the same number of key tiles
    const int nqb = (Lq + BLOCK_M - 1) / BLOCK_M;
    int qb_first = wg.x, qb_second = -1;
    if constexpr (PAIR) {
        qb_first = nqb - 1 - wg.x;
        qb_second = wg.x;
        if (qb_second > qb_first) return;
        if (qb_second == qb_first) qb_second = -1;
    } else if (qb_first >= nqb) {
        return;
    }
    for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
    const int qbi = pass == 0 ? qb_first : qb_second;
    if (qbi < 0) break;
    if (pass == 1) __syncthreads();
    const int qb0 = qbi * BLOCK_M;
    const int qw0 = qb0 + wid * (QT * 16);

    // Q and dO fragments ("B" operands): lane (c, g) holds row q = c, d = 32*ds + 8g .. +7
    bf16x8 qf[QT][DS], dof[QT][DS];
    float lse2[QT], dl[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = qw0 + t * 16 + c;
        const bool ok = qi < Lq;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            const int d = ds * 32 + g * 8;
            u32x4 z = {0u, 0u, 0u, 0u}, y = {0u, 0u, 0u, 0u};
            if (ok && d < p.D) {
                z = *(const u32x4*)(p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh + d);
                y = *(const u32x4*)(p.dout + (long)(qs + qi) * p.do_st + (long)hq * p.do_sh + d);
            }
            qf[t][ds] = __builtin_bit_cast(bf16x8, z);
            dof[t][ds] = __builtin_bit_cast(bf16x8, y);
        }
        lse2[t] = ok ? p.lse[(long)hq * p.total_q + qs + qi] * 1.4426950408889634f : 0.f;
    }
    // delta[q] = rowsum(dO * O) of this workgroup's query rows, computed HERE (it was a pre-pass launch of its own: 22 us of launch and latency per layer for 15 MB):
    // the dO fragments are in registers, the matching O chunks are loaded once, the four lanes (g) of a row add their quarters; the dK / dV kernel, launched after
    // this one, reads the values this kernel leaves in p.delta.  Every (row, head) belongs to exactly one workgroup and pass.
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = qw0 + t * 16 + c;
        const bool ok = qi < Lq;
        float dp = 0.f;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            const int d = ds * 32 + g * 8;
            if (ok && d < p.D) {
                const u32x4 ov = *(const u32x4*)(p.o + (long)(qs + qi) * p.o_st + (long)hq * p.o_sh + d);
                const u32x4 dv = __builtin_bit_cast(u32x4, dof[t][ds]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dp += __uint_as_float(ov[i] << 16) * __uint_as_float(dv[i] << 16);
                    dp += __uint_as_float(ov[i] & 0xffff0000u) * __uint_as_float(dv[i] & 0xffff0000u);
                }
            }
        }
        dp += __shfl_xor(dp, 16, 64);
        dp += __shfl_xor(dp, 32, 64);
        dl[t] = ok ? dp : 0.f;
        if (ok && g == 0) p.delta[(long)hq * p.total_q + qs + qi] = dp;
    }

    f32x4 dqacc[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int d = 0; d < DT; ++d) dqacc[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kv_end = Lk;
    if (p.causal) kv_end = min(Lk, qb0 + BLOCK_M + shift);
    if (kv_end < 0) kv_end = 0;
    const int ntiles = (kv_end + BT - 1) / BT;

    u32x4 kreg[LOADS], vreg[LOADS];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / CH, ch = idx % CH;
            const int key = kt * BT + r;
            u32x4 zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
            if (key < Lk && ch * 8 < p.D) {
                zk = *(const u32x4*)(p.k + (long)(ks + key) * p.k_st + (long)hk * p.k_sh + ch * 8);
                zv = *(const u32x4*)(p.v + (long)(ks + key) * p.v_st + (long)hk * p.v_sh + ch * 8);
            }
            kreg[i] = zk;
            vreg[i] = zv;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / CH, ch = idx % CH;
            *(u32x4*)(Ks + r * STRIDE + ch * 16) = kreg[i];
            *(u32x4*)(Vs + r * STRIDE + ch * 16) = vreg[i];
        }
    };

    if (ntiles > 0) load_tile(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < ntiles) load_tile(kt + 1);
        const int k0 = kt * BT;
        const bool need_mask = (k0 + BT > Lk) || (p.causal && (k0 + BT - 1 > qb0 + shift));
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int qi = qw0 + t * 16 + c;
            // S^T and dP^T for the 64 keys of this tile x query c
            f32x4 s[4], dp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    bf16x8 kf = *(const bf16x8*)(Ks + (j * 16 + c) * STRIDE + ds * 64 + g * 16);
                    bf16x8 vf = *(const bf16x8*)(Vs + (j * 16 + c) * STRIDE + ds * 64 + g * 16);
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ds], s[j], 0, 0, 0);
                    dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[t][ds], dp[j], 0, 0, 0);
                }
            // dS^T = P^T * (dP^T - delta) * scale, packed as the "B" operand of the dQ product
            bf16x8 dsf[2];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                float e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = 2 * ss + (u >> 2), r = u & 3;
                    float pr = __builtin_amdgcn_exp2f(s[j][r] * p.scale_log2 - lse2[t]);  // raw v_exp_f32: p <= 1, tiny values may flush
                    if (need_mask) {
                        const int key = k0 + j * 16 + 4 * g + r;
                        const bool ok = (key < Lk) && (!p.causal || key <= qi + shift);
                        pr = ok ? pr : 0.f;
                    }
                    if (qi >= Lq) pr = 0.f;
                    e[u] = pr * (dp[j][r] - dl[t]) * p.scale;
                }
                u32x4 pk;
                pk[0] = pack_bf2(e[0], e[1]); pk[1] = pack_bf2(e[2], e[3]); pk[2] = pack_bf2(e[4], e[5]); pk[3] = pack_bf2(e[6], e[7]);
                dsf[ss] = __builtin_bit_cast(bf16x8, pk);
            }
            // dQ^T += K^T . dS^T
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const char* a0 = Ks + (ss * 32 + 4 * g + (c >> 2)) * STRIDE + (d * 16 + 4 * (c & 3)) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 16 * STRIDE));
                    bf16x8 kt8;
                    kt8[0] = lo[0]; kt8[1] = lo[1]; kt8[2] = lo[2]; kt8[3] = lo[3];
                    kt8[4] = hi[0]; kt8[5] = hi[1]; kt8[6] = hi[2]; kt8[7] = hi[3];
                    dqacc[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt8, dsf[ss], dqacc[t][d], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qi = qw0 + t * 16 + c;
        if (qi < Lq) {
            unsigned short* row = p.dq + (long)(qs + qi) * p.dq_st + (long)hq * p.dq_sh;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int dd = d * 16 + 4 * g;
                if (dd < p.D) {
                    u32x2 pk;
                    pk[0] = pack_bf2(dqacc[t][d][0], dqacc[t][d][1]);
                    pk[1] = pack_bf2(dqacc[t][d][2], dqacc[t][d][3]);
                    *(u32x2*)(row + dd) = pk;
                }
            }
        }
    }
    }  // pass
}

// ---------------------------------------------------------------------------------------------- dK, dV
// SPLIT (GQA with a workspace): grid.y runs over the QUERY heads; each workgroup handles one head of the group and writes f32
// partial sums, attn_dkv_reduce_kernel adds the group in a fixed order (still no atomics, still bitwise reproducible).  Without
// it one workgroup loops all heads of the group: 33 x 4 = 132 workgroups for the decoder at S = 2112, 0.97 ms per layer.
template <int DP, int NWAVE, bool SPLIT, bool PAIR>
__global__ __launch_bounds__(64 * NWAVE) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
    constexpr int NT = 64 * NWAVE;
    constexpr int BLOCK_N = NWAVE * 16;  // keys per workgroup (one 16-key tile per wave)
    constexpr int CH = DP / 8;
    constexpr int STRIDE = DP * 2 + 32;
    constexpr int DS = DP / 32, DT = DP / 16;
    constexpr int LOADS = (BT * CH) / NT;
    static_assert((BT * CH) % NT == 0, "tile chunks must divide evenly over threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;                       // [BT][STRIDE] row-major Q tile
    char* Os = smem + BT * STRIDE;         // [BT][STRIDE] row-major dO tile
    float* Ls = (float*)(smem + 2 * BT * STRIDE);  // lse2[BT], delta[BT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const Wg3 wg = xcd_decode3(p.gx, p.gy);
    const int seg = wg.z;
    const int group = p.Hq / p.Hkv;
    const int hk = SPLIT ? wg.y / group : wg.y;
    const int hh0 = SPLIT ? wg.y % group : 0, hh1 = SPLIT ? hh0 + 1 : group;
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks = p.cu_k[seg], Lk = p.cu_k[seg + 1] - ks;
    const int shift = Lk - Lq;
    const int nkb = (Lk + BLOCK_N - 1) / BLOCK_N;
    int kb_first = wg.x, kb_second = -1;
    if constexpr (PAIR) {   // causal: key block j sees query tiles j..n-1; pair j with n-1-j
        kb_first = wg.x;                 // the long one (early keys) first
        kb_second = nkb - 1 - wg.x;
        if (kb_first > kb_second) return;
        if (kb_second == kb_first) kb_second = -1;
    } else if (kb_first >= nkb) {
        return;
    }
    for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
    const int kbi = pass == 0 ? kb_first : kb_second;
    if (kbi < 0) break;
    if (pass == 1) __syncthreads();
    const int kb0 = kbi * BLOCK_N;
    const int key = kb0 + wid * 16 + c;  // this lane's key

    // K and V fragments ("B" operands): lane holds key row `key`, d = 32*ds + 8g .. +7
    bf16x8 kf[DS], vf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
        const int d = ds * 32 + g * 8;
        u32x4 z = {0u, 0u, 0u, 0u}, y = {0u, 0u, 0u, 0u};
        if (key < Lk && d < p.D) {
            z = *(const u32x4*)(p.k + (long)(ks + key) * p.k_st + (long)hk * p.k_sh + d);
            y = *(const u32x4*)(p.v + (long)(ks + key) * p.v_st + (long)hk * p.v_sh + d);
        }
        kf[ds] = __builtin_bit_cast(bf16x8, z);
        vf[ds] = __builtin_bit_cast(bf16x8, y);
    }
    f32x4 dkacc[DT], dvacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { dkacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // first query that can see any key of this block (causal): q >= key - shift
    int q_begin = 0;
    if (p.causal) q_begin = max(0, kb0 - shift);
    const int t0 = q_begin / BT;
    const int ntq = (Lq + BT - 1) / BT;

    // Q / dO tiles (and their lse / delta) travel HBM -> registers -> LDS one tile AHEAD of their use, as the dQ kernel stages K / V: the loads of tile i + 1 are issued
    // right after tile i is stored, so a whole tile of matrix work hides their latency (round 5; before, every tile was loaded and stored inside one barrier pair:
    // 17 - 33 exposed memory round trips per workgroup)
    u32x4 qreg[LOADS], oreg[LOADS];
    float lreg = 0.f, dreg = 0.f;
    auto load_tile = [&](int hq_, int qt_) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / CH, ch = idx % CH;
            const int qi = qt_ * BT + r;
            u32x4 zq = {0u, 0u, 0u, 0u}, zo = {0u, 0u, 0u, 0u};
            if (qi < Lq && ch * 8 < p.D) {
                zq = *(const u32x4*)(p.q + (long)(qs + qi) * p.q_st + (long)hq_ * p.q_sh + ch * 8);
                zo = *(const u32x4*)(p.dout + (long)(qs + qi) * p.do_st + (long)hq_ * p.do_sh + ch * 8);
            }
            qreg[i] = zq;
            oreg[i] = zo;
        }
        if (tid < BT) {
            const int qi = qt_ * BT + tid;
            lreg = (qi < Lq) ? p.lse[(long)hq_ * p.total_q + qs + qi] * 1.4426950408889634f : 0.f;
            dreg = (qi < Lq) ? p.delta[(long)hq_ * p.total_q + qs + qi] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / CH, ch = idx % CH;
            *(u32x4*)(Qs + r * STRIDE + ch * 16) = qreg[i];
            *(u32x4*)(Os + r * STRIDE + ch * 16) = oreg[i];
        }
        if (tid < BT) {
            Ls[tid] = lreg;
            Ls[BT + tid] = dreg;
        }
    };
    if (t0 < ntq) load_tile(hk * group + hh0, t0);
    for (int hh = hh0; hh < hh1; ++hh) {
        const int hq = hk * group + hh;
        for (int qt = t0; qt < ntq; ++qt) {
            __syncthreads();  // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (qt + 1 < ntq) load_tile(hq, qt + 1);
            else if (hh + 1 < hh1) load_tile(hq + 1, t0);
            const int q0 = qt * BT;
            // S[q][key], dP[q][key]: "A" = Q / dO rows from LDS, "B" = K / V registers; lane = key, regs = 4 queries (4g + r) per 16-q tile j
            f32x4 s[4], dp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    bf16x8 qa = *(const bf16x8*)(Qs + (j * 16 + c) * STRIDE + ds * 64 + g * 16);
                    bf16x8 oa = *(const bf16x8*)(Os + (j * 16 + c) * STRIDE + ds * 64 + g * 16);
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ds], s[j], 0, 0, 0);
                    dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf[ds], dp[j], 0, 0, 0);
                }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                float e[8], f[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = 2 * ss + (u >> 2), r = u & 3;
                    const int ql = j * 16 + 4 * g + r;   // query index inside the tile
                    const int qi = q0 + ql;
                    float pr = __builtin_amdgcn_exp2f(s[j][r] * p.scale_log2 - Ls[ql]);
                    const bool ok = (qi < Lq) && (key < Lk) && (!p.causal || key <= qi + shift);
                    pr = ok ? pr : 0.f;
                    e[u] = pr;
                    f[u] = pr * (dp[j][r] - Ls[BT + ql]) * p.scale;
                }
                u32x4 a, b;
                a[0] = pack_bf2(e[0], e[1]); a[1] = pack_bf2(e[2], e[3]); a[2] = pack_bf2(e[4], e[5]); a[3] = pack_bf2(e[6], e[7]);
                b[0] = pack_bf2(f[0], f[1]); b[1] = pack_bf2(f[2], f[3]); b[2] = pack_bf2(f[4], f[5]); b[3] = pack_bf2(f[6], f[7]);
                pf[ss] = __builtin_bit_cast(bf16x8, a);
                dsf[ss] = __builtin_bit_cast(bf16x8, b);
            }
            // dV^T += dO^T . P ; dK^T += Q^T . dS   (k = queries, permuted exactly like the forward's P.V step)
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int roff = (ss * 32 + 4 * g + (c >> 2)) * STRIDE + (d * 16 + 4 * (c & 3)) * 2;
                    bf16x4 ol = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Os + roff));
                    bf16x4 oh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Os + roff + 16 * STRIDE));
                    bf16x4 ql_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Qs + roff));
                    bf16x4 qh_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(Qs + roff + 16 * STRIDE));
                    bf16x8 ot, qt8;
                    ot[0] = ol[0]; ot[1] = ol[1]; ot[2] = ol[2]; ot[3] = ol[3]; ot[4] = oh[0]; ot[5] = oh[1]; ot[6] = oh[2]; ot[7] = oh[3];
                    qt8[0] = ql_[0]; qt8[1] = ql_[1]; qt8[2] = ql_[2]; qt8[3] = ql_[3]; qt8[4] = qh_[0]; qt8[5] = qh_[1]; qt8[6] = qh_[2]; qt8[7] = qh_[3];
                    dvacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pf[ss], dvacc[d], 0, 0, 0);
                    dkacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt8, dsf[ss], dkacc[d], 0, 0, 0);
                }
        }
    }
    if constexpr (SPLIT) {
        if (key < Lk) {
            const long half = (long)group * p.total_k * p.Hkv * p.D;
            float* wk = p.dkv_ws + (((long)hh0 * p.total_k + ks + key) * p.Hkv + hk) * p.D;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const int dd = d * 16 + 4 * g;
                if (dd < p.D) {
                    *(f32x4*)(wk + dd) = dkacc[d];
                    *(f32x4*)(wk + half + dd) = dvacc[d];
                }
            }
        }
        continue;
    }
    if (key < Lk) {
        unsigned short* rk = p.dk + (long)(ks + key) * p.dk_st + (long)hk * p.dk_sh;
        unsigned short* rv = p.dv + (long)(ks + key) * p.dv_st + (long)hk * p.dv_sh;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int dd = d * 16 + 4 * g;
            if (dd < p.D) {
                u32x2 a, b;
                a[0] = pack_bf2(dkacc[d][0], dkacc[d][1]); a[1] = pack_bf2(dkacc[d][2], dkacc[d][3]);
                b[0] = pack_bf2(dvacc[d][0], dvacc[d][1]); b[1] = pack_bf2(dvacc[d][2], dvacc[d][3]);
                *(u32x2*)(rk + dd) = a;
                *(u32x2*)(rv + dd) = b;
            }
        }
    }
    }  // pass
}

// dK / dV = sum over the query heads of a GQA group of the f32 partials, in head order (thread = 4 consecutive d of one (key, kv head))
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnBwdArgs p) {
    const int group = p.Hq / p.Hkv;
    const int d4 = p.D / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = p.total_k * p.Hkv * d4;
    if (idx >= total) return;
    const int dd = (int)(idx % d4) * 4;
    const long row = idx / d4;            // key * Hkv + hk
    const int hk = (int)(row % p.Hkv);
    const long key = row / p.Hkv;
    const long plane = p.total_k * p.Hkv * (long)p.D, half = (long)group * plane;
    const float* w = p.dkv_ws + row * p.D + dd;
    f32x4 sk = *(const f32x4*)w, sv = *(const f32x4*)(w + half);
    for (int h = 1; h < group; ++h) {
        sk += *(const f32x4*)(w + h * plane);
        sv += *(const f32x4*)(w + half + h * plane);
    }
    u32x2 a, b;
    a[0] = pack_bf2(sk[0], sk[1]); a[1] = pack_bf2(sk[2], sk[3]);
    b[0] = pack_bf2(sv[0], sv[1]); b[1] = pack_bf2(sv[2], sv[3]);
    *(u32x2*)(p.dk + key * p.dk_st + (long)hk * p.dk_sh + dd) = a;
    *(u32x2*)(p.dv + key * p.dv_st + (long)hk * p.dv_sh + dd) = b;
}

template <int DP>
static int launch_bwd(const AttnBwdArgs& a0, int nseg, int max_q, int max_k, hipStream_t st) {
    AttnBwdArgs a = a0;
    a.gx = 1; a.gy = 1;
    // 8 waves per workgroup at D = 128 (the decoder): a K/V (resp. Q/dO) tile staged once serves twice the rows
    constexpr int NW = (DP == 128) ? 8 : 4;
#ifdef RGA3_BWD_DQ_QT2   // measurement variant: the dQ kernel on 4 waves x 32 query rows (every K / V fragment read serves two 16-row tiles: half the LDS bytes per flop)
    constexpr int QT = 2, NWQ = 4;
#else
    constexpr int QT = (DP >= 128) ? 1 : 2, NWQ = NW;
#endif
    constexpr int BLOCK_M = NWQ * QT * 16;
    constexpr int LDS_DQ = 2 * BT * (DP * 2 + 32);
    constexpr int LDS_DKV = 2 * BT * (DP * 2 + 32) + 2 * BT * 4;
    auto prep = [](const void* k, int lds) -> int {   // once per kernel instantiation
        if (lds <= 48 * 1024) return 0;
        static std::mutex mu;
        static std::set<std::pair<const void*, int>> done;   // (kernel, device): the attribute is per device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (done.count({k, dev})) return 0;
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(-(int)e, "attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        done.insert({k, dev});
        return 0;
    };
    // (delta = rowsum(dO * O) is computed inside the dQ kernel, which runs first and leaves it in a.delta for the dK / dV kernel)
    const unsigned nqb = (unsigned)cdiv(max_q, BLOCK_M), nkb = (unsigned)cdiv(max_k, 16 * NW);
    const bool pair = a.causal && nqb >= 4 && nkb >= 4;   // balanced causal rows (see the kernels)
    if (pair) {
        auto kq = attn_bwd_dq_kernel<DP, QT, NWQ, true>;
        if (int rc = prep((const void*)kq, LDS_DQ)) return rc;
        a.gx = (int)((nqb + 1) / 2); a.gy = a.Hq;
        hipLaunchKernelGGL(kq, dim3((unsigned)a.gx * (unsigned)a.gy * (unsigned)nseg), dim3(64 * NWQ), LDS_DQ, st, a);
    } else {
        auto kq = attn_bwd_dq_kernel<DP, QT, NWQ, false>;
        if (int rc = prep((const void*)kq, LDS_DQ)) return rc;
        a.gx = (int)nqb; a.gy = a.Hq;
        hipLaunchKernelGGL(kq, dim3((unsigned)a.gx * (unsigned)a.gy * (unsigned)nseg), dim3(64 * NWQ), LDS_DQ, st, a);
    }
    RGA3_CHECK_LAUNCH("attn_bwd_dq_kernel");
    const unsigned gx = pair ? (nkb + 1) / 2 : nkb;
    if (a.dkv_ws && a.Hq > a.Hkv) {
        if (pair) {
            auto ks_ = attn_bwd_dkv_kernel<DP, NW, true, true>;
            if (int rc = prep((const void*)ks_, LDS_DKV)) return rc;
            a.gx = (int)gx; a.gy = a.Hq;
            hipLaunchKernelGGL(ks_, dim3((unsigned)a.gx * (unsigned)a.gy * (unsigned)nseg), dim3(64 * NW), LDS_DKV, st, a);
        } else {
            auto ks_ = attn_bwd_dkv_kernel<DP, NW, true, false>;
            if (int rc = prep((const void*)ks_, LDS_DKV)) return rc;
            a.gx = (int)gx; a.gy = a.Hq;
            hipLaunchKernelGGL(ks_, dim3((unsigned)a.gx * (unsigned)a.gy * (unsigned)nseg), dim3(64 * NW), LDS_DKV, st, a);
        }
        RGA3_CHECK_LAUNCH("attn_bwd_dkv_kernel<split>");
        const long rows = a.total_k * a.Hkv;
        hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdiv(rows * (a.D / 4), 256)), dim3(256), 0, st, a);
        RGA3_CHECK_LAUNCH("attn_dkv_reduce_kernel");
        return 0;
    }
    if (pair) {
        auto kk = attn_bwd_dkv_kernel<DP, NW, false, true>;
        if (int rc = prep((const void*)kk, LDS_DKV)) return rc;
        a.gx = (int)gx; a.gy = a.Hkv;
        hipLaunchKernelGGL(kk, dim3((unsigned)a.gx * (unsigned)a.gy * (unsigned)nseg), dim3(64 * NW), LDS_DKV, st, a);
    } else {
        auto kk = attn_bwd_dkv_kernel<DP, NW, false, false>;
        if (int rc = prep((const void*)kk, LDS_DKV)) return rc;
        a.gx = (int)gx; a.gy = a.Hkv;
        hipLaunchKernelGGL(kk, dim3((unsigned)a.gx * (unsigned)a.gy * (unsigned)nseg), dim3(64 * NW), LDS_DKV, st, a);
    }
    RGA3_CHECK_LAUNCH("attn_bwd_dkv_kernel");
    return 0;
}

}  // namespace rga3

using namespace rga3;

extern "C" int rga3_attn_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                    void* dq, void* dk, void* dv, float* delta_ws, const int32_t* cu_q, const int32_t* cu_k, int nseg,
                                    int max_q, int max_k, int64_t total_q, int Hq, int Hkv, int D, const int64_t* strides16, float scale,
                                    int causal, float* dkv_ws, int64_t total_k, void* stream) {
    RGA3_CHECK_ARG(q && k && v && o && dout && lse && dq && dk && dv && delta_ws && cu_q && cu_k && strides16, "attn_bwd: null pointer");
    RGA3_CHECK_ARG(nseg > 0 && max_q > 0 && max_k > 0 && total_q > 0, "attn_bwd: sizes");
    RGA3_CHECK_ARG(Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attn_bwd: Hq=%d Hkv=%d", Hq, Hkv);
    RGA3_CHECK_ARG(D >= 8 && D <= 128 && D % 8 == 0, "attn_bwd: head dim %d unsupported (multiple of 8, <= 128)", D);
    for (int i = 0; i < 16; ++i) RGA3_CHECK_ARG(strides16[i] % 4 == 0, "attn_bwd: stride %d must be a multiple of 4 elements", i);
    for (int i = 0; i < 10; ++i) RGA3_CHECK_ARG(strides16[i] % 8 == 0, "attn_bwd: input stride %d must be a multiple of 8 elements", i);
    AttnBwdArgs a;
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v; a.o = (const unsigned short*)o;
    a.dout = (const unsigned short*)dout; a.dq = (unsigned short*)dq; a.dk = (unsigned short*)dk; a.dv = (unsigned short*)dv;
    a.lse = lse; a.delta = delta_ws; a.cu_q = cu_q; a.cu_k = cu_k;
    a.q_st = strides16[0]; a.q_sh = strides16[1]; a.k_st = strides16[2]; a.k_sh = strides16[3]; a.v_st = strides16[4]; a.v_sh = strides16[5];
    a.o_st = strides16[6]; a.o_sh = strides16[7]; a.do_st = strides16[8]; a.do_sh = strides16[9]; a.dq_st = strides16[10]; a.dq_sh = strides16[11];
    a.dk_st = strides16[12]; a.dk_sh = strides16[13]; a.dv_st = strides16[14]; a.dv_sh = strides16[15];
    a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.total_q = total_q; a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f; a.causal = causal;
    a.dkv_ws = dkv_ws; a.total_k = total_k;
    RGA3_CHECK_ARG(!dkv_ws || (total_k > 0 && D % 4 == 0), "attn_bwd: dkv_ws needs total_k");
    hipStream_t st = (hipStream_t)stream;
    if (D <= 32) return launch_bwd<32>(a, nseg, max_q, max_k, st);
    if (D <= 64) return launch_bwd<64>(a, nseg, max_q, max_k, st);
    if (D <= 96) return launch_bwd<96>(a, nseg, max_q, max_k, st);
    return launch_bwd<128>(a, nseg, max_q, max_k, st);
}
