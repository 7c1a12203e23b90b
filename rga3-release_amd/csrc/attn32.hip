// Non-causal attention over long key ranges for head dims 33 .. 96, 32 query rows per wave on v_mfma_f32_32x32x16_bf16 -- the structure of memattn.hip (which took
// the SAM2 memory cross-attention from 0.41 to 0.94 PF) applied to the ordinary softmax(Q K^T) V calls that were still on the 16-row form:
//   * Hiera-L global attention (reference model/sam2.py:986-1033 MultiScaleAttention with window 0: 4096 tokens per frame, 8 heads x 72),
//   * the Qwen2.5-VL ViT's full-attention blocks (HF modeling_qwen2_5_vl.py:211-291: 1024-token segments, 16 heads x 80).
// S^T = K Q^T puts the query on the lane (online-softmax state lane-local, one cross-half exchange per 64-key tile) and the key tile in the A operand (one ds_read_b128
// = 1 KiB per 32-cycle MFMA: half the LDS array; the 16 x 16 x 32 form needs 1 KiB per 16 cycles).  Registers 8s .. 8s+7 of a 32 x 32 score block, rounded to bf16,
// are k-step s of the B operand of O^T = V^T P^T with a permuted k order that the two ds_read_b64_tr_b16 of the row-major V tile follow (cdna_hip_programming.md 3,
// "An accumulator tile as the next MFMA's operand").
// Head dims that are not a multiple of 16 are zero-padded IN LDS / registers only (72 -> 80 for Q K^T, -> 96 rows of O^T that are never stored).
// Workgroup = 8 waves x 32 queries = 256 query rows of one (segment, head); 64-key tiles HBM -> registers -> LDS, double-buffered, one barrier per tile; the output
// tile is transposed through LDS so that whole head rows (D x 2 bytes) leave per store.  GQA: kv head = q head / (Hq / Hkv).  Deterministic.
#include "attn32.h"

#include <math.h>

namespace rga3 {

constexpr int A32_KT = 64;
constexpr int A32_QB = 256;

template <int KS, int DB>   // KS = ceil(D / 16) k-steps of Q K^T; DB = ceil(D / 32) row blocks of O^T
__global__ __launch_bounds__(512) void attn32_kernel(Attn32Args p) {
    constexpr int KSTR = KS * 32 + 16;                        // K image row: padded so that 16 rows of a ds_read_b128 group hit 16 different 16-byte slots
    constexpr int VSTR = (DB * 64 % 256 == 64 || DB * 64 % 256 == 192) ? DB * 64 : DB * 64 + 64;   // V image row: 4 consecutive rows x 64 B disjoint mod 256 B
    constexpr int TILE = A32_KT * (KSTR + VSTR);
    constexpr int OSTR = DB * 64 + 16;                        // output tile row
    static_assert((KSTR / 16) % 2 == 1, "K row stride must be an odd number of 16-byte chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const Wg3 wg = xcd_decode3(p.gx, p.Hq);                   // (query block, head, segment), heads fastest: one XCD sees all heads of a block
    const int qb = wg.x, hq = wg.y, seg = wg.z;
    const int hk = hq / (p.Hq / p.Hkv);
    const int qs = p.cu_q[seg], Lq = p.cu_q[seg + 1] - qs;
    const int ks0 = p.cu_k[seg], Lk = p.cu_k[seg + 1] - ks0;
    if (qb * A32_QB >= Lq) return;
    const int q0 = qb * A32_QB + wave * 32;
    const int D = p.D;
    const int nchunk = D / 8;                                 // 16-byte chunks of a head row

    bf16x8 qf[KS];
    {
        const int qi = min(q0 + r, Lq - 1);
        const unsigned short* qrow = p.q + (long)(qs + qi) * p.q_st + (long)hq * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = 16 * ks + 8 * h;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (d < D) z = *(const u32x4*)(qrow + d);
            qf[ks] = __builtin_bit_cast(bf16x8, z);
        }
    }
    // zero the padding of both LDS images once (columns D .. of the K rows feed the products, rows D .. of O^T are never stored but must not carry NaN patterns)
    for (int i = tid; i < 2 * TILE / 16; i += 512) *(u32x4*)(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    const unsigned short* kbase = p.k + (long)ks0 * p.k_st + (long)hk * p.k_sh;
    const unsigned short* vbase = p.v + (long)ks0 * p.v_st + (long)hk * p.v_sh;
    constexpr int NLD = (A32_KT * 12 + 511) / 512;            // <= 12 chunks per row (D <= 96)
    u32x4 kreg[NLD], vreg[NLD];
    const int per_tile = A32_KT * nchunk;
    auto load_tile = [&](int t) {
        const int k0 = t * A32_KT;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + j * 512;
            u32x4 zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
            if (idx < per_tile) {
                const int row = idx / nchunk, ch = idx % nchunk;
                if (k0 + row < Lk) {
                    zk = *(const u32x4*)(kbase + (long)(k0 + row) * p.k_st + ch * 8);
                    zv = *(const u32x4*)(vbase + (long)(k0 + row) * p.v_st + ch * 8);
                }
            }
            kreg[j] = zk;
            vreg[j] = zv;
        }
    };
    auto store_tile = [&](int buf) {
        char* Kb = smem + buf * TILE;
        char* Vb = Kb + A32_KT * KSTR;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + j * 512;
            if (idx < per_tile) {
                const int row = idx / nchunk, ch = idx % nchunk;
                *(u32x4*)(Kb + row * KSTR + ch * 16) = kreg[j];
                *(u32x4*)(Vb + row * VSTR + ch * 16) = vreg[j];
            }
        }
    };

    f32x16 o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = p.scale_log2;
    const int g = lane >> 4, q_ = (lane >> 2) & 3, p_ = lane & 3;
    const int tr_off = (4 * (g >> 1) + q_) * VSTR + (16 * (g & 1) + 4 * p_) * 2;
    const int ntiles = (Lk + A32_KT - 1) / A32_KT;

    if (ntiles > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) load_tile(t + 1);
        const char* Kb = smem + buf * TILE;
        const char* Vb = Kb + A32_KT * KSTR;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
            const char* ka = Kb + (kb * 32 + r) * KSTR + h * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(ka + ks * 32), qf[ks], s[kb], 0, 0, 0);
        }
        if ((t + 1) * A32_KT > Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t * A32_KT + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= Lk) s[kb][i] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
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
            for (int b = 0; b < DB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[b][i] *= alpha;
        }
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
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const char* a0 = Vb + (kb * 32 + 16 * ss) * VSTR + b * 64 + tr_off;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 8 * VSTR));
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kb][ss], o[b], 0, 0, 0);
                }
        if (t + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    // ---- normalise, transpose through LDS, store whole head rows
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    char* ot = smem;
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int d = b * 32 + 8 * i4 + 4 * h;
            *(u32x2*)(ot + (wave * 32 + r) * OSTR + d * 2) =
                u32x2{pack_bf2(o[b][4 * i4] * inv, o[b][4 * i4 + 1] * inv), pack_bf2(o[b][4 * i4 + 2] * inv, o[b][4 * i4 + 3] * inv)};
        }
    const int qi = q0 + r;
    if (p.lse && h == 0 && qi < Lq) p.lse[(long)hq * p.total_q + qs + qi] = l_run > 0.f ? (m_run * 0.6931471805599453f + logf(l_run)) : -INFINITY;
    __syncthreads();
    const int nst = D / 4;                                    // 8-byte pieces per head row (o strides are multiples of 4 elements)
    for (int idx = tid; idx < A32_QB * nst; idx += 512) {
        const int row = idx / nst, pc = idx % nst;
        const int qrow = qb * A32_QB + row;
        if (qrow < Lq) *(u32x2*)(p.o + (long)(qs + qrow) * p.o_st + (long)hq * p.o_sh + pc * 4) = *(const u32x2*)(ot + row * OSTR + pc * 8);
    }
}

template <int KS, int DB>
static int launch32(const Attn32Args& a, int nseg, hipStream_t st) {
    constexpr int KSTR = KS * 32 + 16;
    constexpr int VSTR = (DB * 64 % 256 == 64 || DB * 64 % 256 == 192) ? DB * 64 : DB * 64 + 64;
    constexpr int LDS_KV = 2 * A32_KT * (KSTR + VSTR), LDS_O = A32_QB * (DB * 64 + 16);
    constexpr int LDS = LDS_KV > LDS_O ? LDS_KV : LDS_O;      // the output tile reuses the K / V buffers
    auto kern = attn32_kernel<KS, DB>;
    static bool attr_done = false;
    if (!attr_done && LDS > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail(-(int)e, "attn32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.gx * (unsigned)a.Hq * (unsigned)nseg), dim3(512), LDS, st, a);
    RGA3_CHECK_LAUNCH("attn32_kernel");
    return 0;
}

bool attn32_applies(int D, int causal, int max_q, int max_k) {
    return !causal && D > 32 && D <= 96 && D % 8 == 0 && max_q >= 256 && max_k >= 512;
}

int attn32_launch(Attn32Args a, int nseg, int max_q, hipStream_t st) {
    a.gx = (max_q + A32_QB - 1) / A32_QB;
    const int KS = (a.D + 15) / 16, DB = (a.D + 31) / 32;
    if (DB == 2) return KS <= 3 ? launch32<3, 2>(a, nseg, st) : launch32<4, 2>(a, nseg, st);
    return KS <= 5 ? launch32<5, 3>(a, nseg, st) : launch32<6, 3>(a, nseg, st);
}

}  // namespace rga3
