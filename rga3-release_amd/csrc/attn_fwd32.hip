// Attention forward, second generation: one wave per SIMD, 32 query rows per wave, v_mfma_f32_32x32x16_bf16.
//
// Why (round-2 measurement of attn_fwd_kernel, the 8-wave x 16-row form): every wave re-reads the whole K / V tile from LDS for only 16 query
// rows (one ds_read_b128 per MFMA: the LDS array, not the matrix pipe, paces the tile) and two barriers per tile keep the eight waves in lockstep, so
// MFMA, softmax VALU and LDS time add up (28 % / 40 % / 37 % of the cycles).  Here a wave owns 32 query rows, so every K / V fragment read from LDS
// feeds a 32x32 MFMA (half the LDS bytes per flop), K tiles are staged two tiles ahead and V tiles one tile ahead in separate two-slot rings so ONE
// barrier per tile suffices, and the loop is software-pipelined: the S^T = K Q^T MFMAs of tile t+1 are issued between the softmax VALU of tile t
// (independent work in one instruction stream: the MFMA runs while the vector ALU exponentiates), then O^T += V^T P^T of tile t.
//
// Register maps (cdna_hip_programming.md, 32x32x16 bf16): A lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j]; B holds B[k = 8h + j][col r];
// C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4h.
//  * S^T = K . Q^T: A = K rows straight from the row-major LDS image (ds_read_b128), B = Q in registers; the QUERY is the lane's column, so the
//    online-softmax state is one scalar per lane (lanes l and l + 32 share a query: one shuffle for the row max / sum).
//  * O^T += V^T . P^T: the S registers are, as they stand, the B operand (key order within a 16-key step permuted: slot (h, j) is key
//    8 (j >> 2) + 4h + (j & 3)); the V^T A operand with the same permutation is what two ds_read_b64_tr_b16 of the row-major V image deliver.
//  * optional RoPE on Q at load time (cos / sin tables [T, D] f32, rotate-half pairing d <-> d +- D/2): the stand-alone rope pass then only touches K.
#include "common.h"
#include <type_traits>

namespace rga3 {

struct Attn32Args {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    unsigned short* o;
    float* lse;
    const int* cu_q;
    const int* cu_k;
    long q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh;
    int Hq, Hkv, D;
    long total_q;
    float scale_log2;
    int causal;
    const float* rope_cos;   // [total_q, D] f32 or null: rotate Q while loading it
    const float* rope_sin;
};

typedef __attribute__((ext_vector_type(16))) float f32x16v;

template <int DP, int NWAVE, bool PAIR>
__global__ __launch_bounds__(64 * NWAVE) void attn_fwd32_kernel(Attn32Args p) {
    constexpr int NT = 64 * NWAVE;
    constexpr int BLOCK_M = NWAVE * 32;
    constexpr int KV = 64;                 // keys per tile
    constexpr int CH = DP / 8;             // 16-byte chunks per row
    constexpr int STRIDE = DP * 2 + 32;    // LDS row stride (bytes)
    constexpr int TILE_B = KV * STRIDE;
    constexpr int KS = DP / 16;            // 16-wide d steps of K Q^T
    constexpr int DB = DP / 32;            // 32-row d blocks of the output
    constexpr int LOADS = (KV * CH + NT - 1) / NT;
    constexpr bool EVEN = (KV * CH) % NT == 0;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kb = smem;                  // two K tiles
    char* Vb = smem + 2 * TILE_B;     // two V tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int seg = blockIdx.z, hq = blockIdx.y;
    const int hk = hq / (p.Hq / p.Hkv);
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks0 = p.cu_k[seg], Lk = p.cu_k[seg + 1] - ks0;
    const int shift = Lk - Lq;
    const int nqb = (Lq + BLOCK_M - 1) / BLOCK_M;
    int qb_first = (int)blockIdx.x, qb_second = -1;
    if constexpr (PAIR) {
        qb_first = nqb - 1 - (int)blockIdx.x;   // the long one first
        qb_second = (int)blockIdx.x;
        if (qb_second > qb_first) return;
        if (qb_second == qb_first) qb_second = -1;
    } else if (qb_first >= nqb) {
        return;
    }

    const unsigned short* kbase = p.k + (long)ks0 * p.k_st + (long)hk * p.k_sh;
    const unsigned short* vbase = p.v + (long)ks0 * p.v_st + (long)hk * p.v_sh;
    int koff0[LOADS], voff0[LOADS], soff[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int idx = tid + i * NT;
        const int rr = idx / CH, ch = idx % CH;
        koff0[i] = (int)(rr * p.k_st + ch * 8);
        voff0[i] = (int)(rr * p.v_st + ch * 8);
        soff[i] = rr * STRIDE + ch * 16;
    }

    for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
        const int qbi = pass == 0 ? qb_first : qb_second;
        if (qbi < 0) break;
        if (pass == 1) __syncthreads();
        const int qb0 = qbi * BLOCK_M;
        const int qi = qb0 + wid * 32 + r;   // this lane's query (shared with lane ^ 32)

        // ---- Q fragments (B operand): lane (r, h) holds Q[qi][16 ks + 8h .. +7]
        bf16x8 qf[KS];
        {
            const unsigned short* qrow = p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh;
            const int half = p.D >> 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d = ks * 16 + h * 8;
                u32x4 z = {0u, 0u, 0u, 0u};
                if (qi < Lq && d < p.D) {
                    z = *(const u32x4*)(qrow + d);
                    if (p.rope_cos) {
                        const bool first = d < half;
                        const int dp = first ? d + half : d - half;
                        const u32x4 zp = *(const u32x4*)(qrow + dp);
                        const float* cs = p.rope_cos + (long)(qs + qi) * p.D + d;
                        const float* sn = p.rope_sin + (long)(qs + qi) * p.D + d;
                        const f32x4 c0 = *(const f32x4*)cs, c1 = *(const f32x4*)(cs + 4), s0 = *(const f32x4*)sn, s1 = *(const f32x4*)(sn + 4);
                        float x[8], xp[8], cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x[2 * e] = __uint_as_float(z[e] << 16); x[2 * e + 1] = __uint_as_float(z[e] & 0xffff0000u);
                            xp[2 * e] = __uint_as_float(zp[e] << 16); xp[2 * e + 1] = __uint_as_float(zp[e] & 0xffff0000u);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {   // rope_kernel's arithmetic: x cos + rotate_half(x) sin in f32 (first half: - partner, second half: + partner), one bf16 rounding
                            const float a0 = first ? x[2 * e] * cc[2 * e] - xp[2 * e] * ss[2 * e] : x[2 * e] * cc[2 * e] + xp[2 * e] * ss[2 * e];
                            const float a1 = first ? x[2 * e + 1] * cc[2 * e + 1] - xp[2 * e + 1] * ss[2 * e + 1] : x[2 * e + 1] * cc[2 * e + 1] + xp[2 * e + 1] * ss[2 * e + 1];
                            z[e] = pack_bf2(a0, a1);
                        }
                    }
                }
                qf[ks] = __builtin_bit_cast(bf16x8, z);
            }
        }

        f32x16v oacc[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        int kv_end = Lk;
        if (p.causal) kv_end = min(Lk, qb0 + BLOCK_M + shift);
        if (kv_end < 0) kv_end = 0;
        const int ntiles = (kv_end + KV - 1) / KV;

        u32x4 kreg[LOADS], vreg[LOADS];
        auto load_k = [&](int kt) {
            const unsigned short* src = kbase + (long)kt * KV * p.k_st;
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int idx = tid + i * NT;
                const int key = kt * KV + idx / CH, ch = idx % CH;
                u32x4 z = {0u, 0u, 0u, 0u};
                if ((EVEN || idx < KV * CH) && kt < ntiles && key < Lk && ch * 8 < p.D) z = *(const u32x4*)(src + koff0[i]);
                kreg[i] = z;
            }
        };
        auto load_v = [&](int kt) {
            const unsigned short* src = vbase + (long)kt * KV * p.v_st;
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int idx = tid + i * NT;
                const int key = kt * KV + idx / CH, ch = idx % CH;
                u32x4 z = {0u, 0u, 0u, 0u};
                if ((EVEN || idx < KV * CH) && kt < ntiles && key < Lk && ch * 8 < p.D) z = *(const u32x4*)(src + voff0[i]);
                vreg[i] = z;
            }
        };
        auto store_k = [&](int slot) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i)
                if (EVEN || tid + i * NT < KV * CH) *(u32x4*)(Kb + slot * TILE_B + soff[i]) = kreg[i];
        };
        auto store_v = [&](int slot) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i)
                if (EVEN || tid + i * NT < KV * CH) *(u32x4*)(Vb + slot * TILE_B + soff[i]) = vreg[i];
        };
        // S^T tile of K tile in `slot`: two 32-key blocks x 32 queries
        auto qk = [&](int slot, f32x16v* s) {
            const char* Ks = Kb + slot * TILE_B;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(Ks + (kb * 32 + r) * STRIDE + ks * 32 + h * 16);
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
                }
            }
        };

        // ---- prologue: K0 -> slot 0, S(0); then K1 -> slot 1, V0 -> slot 0
        load_k(0);
        store_k(0);
        load_k(1);
        load_v(0);
        __syncthreads();
        f32x16v s_cur[2], s_nxt[2];
        qk(0, s_cur);
        store_k(1);
        store_v(0);
        load_k(2);
        load_v(1);
        __syncthreads();

        // One tile: stage K(kt+2) / V(kt+1) (their slots were last read one iteration ago, behind the barrier that ended it), issue the S^T MFMAs of
        // tile kt+1 into `sn`, run the online softmax of tile kt on `sc`, accumulate O^T, barrier.
        auto tile = [&](int kt, f32x16v* sc, f32x16v* sn) {
            store_k(kt & 1);
            store_v((kt + 1) & 1);
            load_k(kt + 3);
            load_v(kt + 2);
            qk((kt + 1) & 1, sn);   // independent of this tile's softmax: the MFMAs run under the exponentials

            // ---- online softmax of this tile: lane = one query, 32 of its 64 scores (the other 32 in lane ^ 32)
            const int k0 = kt * KV;
            const bool need_mask = (k0 + KV > Lk) || (p.causal && (k0 + KV - 1 > qb0 + wid * 32 + shift));
            float mx = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const bool ok = (key < Lk) && (!p.causal || key <= qi + shift);
                        const float x = ok ? sc[kb][e] * p.scale_log2 : -INFINITY;
                        sc[kb][e] = x;
                        mx = fmaxf(mx, x);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float x = sc[kb][e] * p.scale_log2;
                        sc[kb][e] = x;
                        mx = fmaxf(mx, x);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float ex = __builtin_amdgcn_exp2f(sc[kb][e] - m_use);
                    sc[kb][e] = ex;
                    ps += ex;
                }
            l_run = l_run * alpha + ps;
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
            }
            // ---- O^T += V^T . P^T : 16-key steps t = 0..3, P fragment = S registers 8 (t & 1) .. +7 of key block t >> 1
            const char* Vs = Vb + (kt & 1) * TILE_B;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u32x4 pk;
                const int e0 = 8 * (t & 1);
                pk[0] = pack_bf2(sc[t >> 1][e0 + 0], sc[t >> 1][e0 + 1]);
                pk[1] = pack_bf2(sc[t >> 1][e0 + 2], sc[t >> 1][e0 + 3]);
                pk[2] = pack_bf2(sc[t >> 1][e0 + 4], sc[t >> 1][e0 + 5]);
                pk[3] = pack_bf2(sc[t >> 1][e0 + 6], sc[t >> 1][e0 + 7]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    // 16-lane group G = lane >> 4 reads the 4-key x 16-d block (keys 16t + 4h .., d = 32 d + 16 (G & 1) ..)
                    const char* a0 = Vs + (16 * t + 4 * h + ((lane & 15) >> 2)) * STRIDE + (d * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 8 * STRIDE));
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                }
            }
            __syncthreads();
        };
        for (int kt = 0; kt < ntiles; kt += 2) {   // two tiles per trip: the score registers swap roles instead of being copied
            tile(kt, s_cur, s_nxt);
            if (kt + 1 < ntiles) tile(kt + 1, s_nxt, s_cur);
        }

        // ---- finalize: lane holds query qi, output dims 32 d + 8 (e >> 2) + 4h + (e & 3)
        float l = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = (l > 0.f) ? 1.f / l : 0.f;
        if (qi < Lq) {
            unsigned short* orow = p.o + (long)(qs + qi) * p.o_st + (long)hq * p.o_sh;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int dd = d * 32 + 8 * qd + 4 * h;
                    if (dd < p.D) {
                        u32x2 pk;
                        pk[0] = pack_bf2(oacc[d][4 * qd] * inv, oacc[d][4 * qd + 1] * inv);
                        pk[1] = pack_bf2(oacc[d][4 * qd + 2] * inv, oacc[d][4 * qd + 3] * inv);
                        *(u32x2*)(orow + dd) = pk;
                    }
                }
            if (p.lse && h == 0) p.lse[(long)hq * p.total_q + qs + qi] = (l > 0.f) ? (m_run * 0.6931471805599453f + logf(l)) : -INFINITY;
        }
    }
}

template <int DP, int NWAVE, bool PAIR>
static int launch32(const Attn32Args& a, int nseg, unsigned gx, hipStream_t st) {
    constexpr int LDS = 4 * 64 * (DP * 2 + 32);
    auto kern = attn_fwd32_kernel<DP, NWAVE, PAIR>;
    static bool attr_done = false;
    if (!attr_done && LDS > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail(-(int)e, "attn32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)a.Hq, (unsigned)nseg), dim3(64 * NWAVE), LDS, st, a);
    RGA3_CHECK_LAUNCH("attn_fwd32_kernel");
    return 0;
}

// Entry used by rga3_attn_varlen_fwd (attn_fwd.hip).  Returns 1 when this form does not apply (caller falls back to the first-generation kernel).
int attn_fwd32_try(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* cu_q, const int32_t* cu_k, int nseg, int max_q,
                   int64_t total_q, int Hq, int Hkv, int D, int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh, int64_t o_st,
                   int64_t o_sh, float scale, int causal, const float* rope_cos, const float* rope_sin, void* stream) {
    if (D > 128 || D % 8 != 0 || max_q < 128) return 1;
    if (rope_cos && (D % 16 != 0)) return 1;
    Attn32Args a;
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v; a.o = (unsigned short*)o; a.lse = lse;
    a.cu_q = cu_q; a.cu_k = cu_k;
    a.q_st = q_st; a.q_sh = q_sh; a.k_st = k_st; a.k_sh = k_sh; a.v_st = v_st; a.v_sh = v_sh; a.o_st = o_st; a.o_sh = o_sh;
    a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.total_q = total_q; a.scale_log2 = scale * 1.4426950408889634f; a.causal = causal;
    a.rope_cos = rope_cos; a.rope_sin = rope_sin;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nqb = (unsigned)cdiv(max_q, 128);
    const bool pair = causal && nqb >= 4;
    const unsigned gx = pair ? (nqb + 1) / 2 : nqb;
    if (D <= 64) return pair ? launch32<64, 4, true>(a, nseg, gx, st) : launch32<64, 4, false>(a, nseg, gx, st);
    if (D <= 96) return pair ? launch32<96, 4, true>(a, nseg, gx, st) : launch32<96, 4, false>(a, nseg, gx, st);
    return pair ? launch32<128, 4, true>(a, nseg, gx, st) : launch32<128, 4, false>(a, nseg, gx, st);
}

}  // namespace rga3
