// Fused MLP of a Hiera block (reference model/sam2.py:1035-1117 MultiScaleBlock.forward: x = x + mlp(norm2(x)); MLP :2305-2329, dim C -> 4 C -> C, exact-erf GELU) for
// the frozen SAM2-L trunk:   y = x + W2 gelu(W1 LayerNorm(x) + b1) + b2   in ONE launch, for stage 1 (C = 144, 65 536 tokens per 1024^2 frame) and stage 2 (C = 288,
// 16 384 tokens per frame, blocks 2 - 7).
//
// Why: as two GEMMs the 4 C-wide hidden activation (1.2 GB at stage 1, 604 MB per stage-2 block and 16 frames) is written and read back and both products run
// HBM-bound at 2 - 3.8 TB/s (stage 1) or at a third of the matrix peak on K = 288 tiles (stage 2).  Fused, the hidden activation never leaves the CU: HBM traffic is
// x in + y out, and the LayerNorm statistics pass disappears as well (a wave holds whole rows).
//
// Dataflow -- "tokens on the lanes": every product is computed TRANSPOSED with v_mfma_f32_32x32x16_bf16 so that the activations are always the B operand and the
// weights the A operand read from LDS:
//     H^T[hidden, tok] = W1'[hidden, ch] . X^T[ch, tok]        X^T fragments live in registers for the whole kernel (C / 16 k-steps x 4 VGPRs per 32 tokens)
//     Y^T[ch, tok]    += W2[ch, hidden]  . G^T[hidden, tok]     G = gelu(LN-fold(H)) taken STRAIGHT from H's accumulator registers: a 32 x 32 result has its
//                                                               column (token) on the lane and its rows (hidden) in the 16 registers, so registers 8s .. 8s+7,
//                                                               rounded to bf16, are k-step s of the next B operand with the k order permuted (cdna_hip_programming.md
//                                                               3, "An accumulator tile as the next MFMA's operand"); the A operand W2 is read with that permutation
//                                                               (two ds_read_b64 per fragment).  No LDS round trip, no barrier between the two products.
// LayerNorm is folded as in rga3_gemm_ln_bf16: W1' = W1 diag(gamma) (bf16), h = rinv (acc - mean c_n) + d_n with c = row sums of W1', d = beta W1^T + b1; mean / rinv
// of a token are lane-local (each half-wave holds half of its row, one exchange).
// GELU is the tile epilogues' table form (act_table.h: the linear output has just been rounded to bf16, so gelu(t) = relu(t) - |t| Phi(-|t|) with Phi read from a
// 10-KiB LDS table: five full-rate vector instructions and one ds_read_b32 per element; the closed form -- v_rcp, v_exp and ten more -- made the kernel vector-ALU
// bound: 32 values x ~80 cycles per chunk and wave against 1 216 cycles of MFMA).
//
// C = 144: workgroup = 8 waves x 32 tokens (two waves per SIMD cover each other's vector phases); the 576 hidden units stream through LDS in 9 chunks of 64.
// C = 288: X^T (72) + Y^T (144) + H (2 x 16) registers per lane do not fit 256: 4 waves x 32 tokens, ONE wave per SIMD on the 512-register budget, hidden chunks of 32;
//          the first product of chunk n + 1 is issued before the GELU of chunk n (software pipeline over two accumulator sets), so that the lone wave's vector work
//          sits between independent MFMAs.
// Weight chunks (W1' chunk HC x C, W2 chunk C x HC, their c / d) are register-staged into a double buffer, one barrier per chunk.  LDS images: W1' rows padded to
// 2 C + 16 B (an odd number of 16-byte slots: conflict-free ds_read_b128), W2 rows to 2 HC + 8 B (34 / 18 dwords: the 32 rows of a half-wave's ds_read_b64 tile
// all 64 banks).  The output tile is transposed through LDS so that y leaves in whole rows.
#include "common.h"
#include "act_table.h"

namespace rga3 {

template <int C_, int NW_, int HC_, int NSTG_>
struct HmCfg {
    static constexpr int C = C_, NW = NW_, HC = HC_, NSTG = NSTG_;   // NSTG: LDS stages of weight chunks (2: double buffer; 3: the pipelined form's ring)
    static constexpr int H = 4 * C, NT = 64 * NW, TOK = 32 * NW;
    static constexpr int KS = C / 16;                          // k-steps of the first product
    static constexpr int CB = (C + 31) / 32;                   // channel blocks of 32 (144 -> 160: rows 144..159 are never stored)
    static constexpr int NHB = HC / 32;                        // hidden blocks of 32 per chunk
    static constexpr int W1STR = C * 2 + 16, W2STR = HC * 2 + 8;
    static constexpr int W1B = HC * W1STR, W2B = CB * 32 * W2STR;
    static constexpr int STAGE = W1B + W2B + HC * 8;           // + c (f32) and d (f32) of the chunk
    static constexpr int OSTR = C * 2 + 16;                    // output tile rows
    static constexpr int TAB = NSTG * STAGE;                   // activation table behind the stages
    static constexpr int LDS = NSTG * STAGE + kActTabBytes;
    static constexpr int W1P = HC * (C / 8), W2P = C * (HC / 8);   // 16-byte pieces of a chunk
    static constexpr int NJ1 = (W1P + NT - 1) / NT, NJ2 = (W2P + NT - 1) / NT;
    static constexpr int NCH = H / HC;
    static_assert(C % 16 == 0 && HC % 32 == 0 && H % HC == 0 && STAGE % 16 == 0 && NSTG * STAGE >= TOK * OSTR && 2 * HC <= NT && LDS <= 160 * 1024, "hiera_mlp: configuration");
};

struct HmArgs {
    const unsigned short* x;     // [M, C] bf16
    const unsigned short* w1f;   // [4C, C] bf16 = W1 diag(gamma)
    const float* c1;             // [4C] row sums of w1f
    const unsigned short* d1;    // [4C] bf16 folded bias
    const unsigned short* w2;    // [C, 4C] bf16
    const unsigned short* b2;    // [C] bf16
    unsigned short* y;           // [M, C] bf16
    long M;
    float eps;
};

template <class K>
__global__ __launch_bounds__(K::NT) void hiera_mlp_kernel(HmArgs p) {
    constexpr int C = K::C, HC = K::HC, KS = K::KS, CB = K::CB, NHB = K::NHB, NT = K::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long tok0 = (long)blockIdx.x * K::TOK;
    const long tok = tok0 + wave * 32 + r;
    const long tokc = tok < p.M ? tok : p.M - 1;

    // ---- this wave's 32 tokens as B operands: lane (r, h) holds x[tok][16 ks + 8 h .. + 8]
    bf16x8 xf[KS];
    {
        const unsigned short* xr = p.x + tokc * C + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(xr + 16 * ks);
    }
    // ---- LayerNorm statistics of the token (two passes over the register-resident half row, halves exchanged)
    float rinv, nmr;     // h = rinv acc + (d - mean rinv c)
    {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)xf[ks][e];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)xf[ks][e] - mean;
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        rinv = __builtin_amdgcn_rsqf(q * (1.0f / C) + p.eps);
        nmr = -mean * rinv;
    }

    // ---- weight chunk staging: HBM / L2 -> registers -> LDS
    u32x4 w1r[K::NJ1], w2r[K::NJ2];
    float cr = 0.f;
    auto load_chunk = [&](int ch) {
        const int n0 = ch * HC;
#pragma unroll
        for (int j = 0; j < K::NJ1; ++j) {
            const int idx = tid + j * NT;
            if (idx < K::W1P) {                // W1' rows n0 .. n0 + HC - 1, C / 8 chunks of 16 B each
                const int row = idx / (C / 8), c16 = idx % (C / 8);
                w1r[j] = *(const u32x4*)(p.w1f + (long)(n0 + row) * C + c16 * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < K::NJ2; ++j) {
            const int idx = tid + j * NT;
            if (idx < K::W2P) {                // W2 rows 0 .. C - 1, columns n0 .. n0 + HC - 1: HC / 8 chunks of 16 B each
                const int row = idx / (HC / 8), c16 = idx % (HC / 8);
                w2r[j] = *(const u32x4*)(p.w2 + (long)row * K::H + n0 + c16 * 8);
            }
        }
        if (tid < HC) cr = p.c1[n0 + tid];
        else if (tid < 2 * HC) cr = bf2f(p.d1[n0 + tid - HC]);
    };
    auto store_chunk = [&](int buf) {
        char* base = smem + buf * K::STAGE;
#pragma unroll
        for (int j = 0; j < K::NJ1; ++j) {
            const int idx = tid + j * NT;
            if (idx < K::W1P) *(u32x4*)(base + (idx / (C / 8)) * K::W1STR + (idx % (C / 8)) * 16) = w1r[j];
        }
#pragma unroll
        for (int j = 0; j < K::NJ2; ++j) {
            const int idx = tid + j * NT;
            if (idx < K::W2P) {
                // a 16-byte chunk of a (2 HC + 8)-byte row is only 8-byte aligned: two 8-byte stores
                char* d = base + K::W1B + (idx / (HC / 8)) * K::W2STR + (idx % (HC / 8)) * 16;
                *(u32x2*)d = u32x2{w2r[j][0], w2r[j][1]};
                *(u32x2*)(d + 8) = u32x2{w2r[j][2], w2r[j][3]};
            }
        }
        if (tid < 2 * HC) *(float*)(base + K::W1B + K::W2B + tid * 4) = cr;
    };

    f32x16 y[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[cb][i] = 0.f;

    if constexpr (CB * 32 > C) {
        // rows C .. 32 CB - 1 of the W2 image feed output rows that are never stored: give them zeros once (both buffers) so no NaN pattern wanders through the MFMAs
        constexpr int PADQ = (CB * 32 - C) * K::W2STR / 8;
        for (int i = tid; i < K::NSTG * PADQ; i += NT) {
            const int buf = i / PADQ, o = i % PADQ;
            *(u32x2*)(smem + buf * K::STAGE + K::W1B + C * K::W2STR + o * 8) = u32x2{0u, 0u};
        }
    }
    // the GELU table (10 KiB) behind the two stages
    for (int i = tid; i < kActTabN; i += NT) *(unsigned*)(smem + K::TAB + i * 4) = g_act_tab[0][i];
    const unsigned tb = act_tab_base(smem + K::TAB);

    // H^T block hb of the chunk staged in `st`: W1' chunk . X^T
    auto first_product = [&](f32x16& acc, const char* st, int hb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const char* a0 = st + (hb * 32 + r) * K::W1STR + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(a0 + ks * 32), xf[ks], acc, 0, 0, 0);
    };
    // LayerNorm fold + GELU + bf16 on block hb's accumulators (vector ALU), then its share of Y^T += W2 chunk . G^T
    auto second_product = [&](const f32x16& acc, const char* st, int hb) {
        const char* w2s = st + K::W1B;
        const float* cs = (const float*)(w2s + K::W2B);      // [HC] c, then [HC] d
        bf16x8 gb[2];
        unsigned pk[8];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 cc = *(const f32x4*)(cs + hb * 32 + 8 * i4 + 4 * h);
            const f32x4 dd = *(const f32x4*)(cs + HC + hb * 32 + 8 * i4 + 4 * h);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rinv, acc[4 * i4 + e], __builtin_fmaf(nmr, cc[e], dd[e]));
            // bf16 rounding of the linear output before the activation, as the unfused pair does; gelu from the table on the packed pair
            const f32x2 g01 = act_tab2(pack_bf2(v[0], v[1]), tb), g23 = act_tab2(pack_bf2(v[2], v[3]), tb);
            pk[2 * i4] = pack_bf2(g01[0], g01[1]);
            pk[2 * i4 + 1] = pack_bf2(g23[0], g23[1]);
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) gb[ss] = __builtin_bit_cast(bf16x8, u32x4{pk[4 * ss], pk[4 * ss + 1], pk[4 * ss + 2], pk[4 * ss + 3]});
        // fragment element j of half h is hidden 16 ss + 8 (j >> 2) + 4 h + (j & 3) of block hb
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const char* a0 = w2s + (cb * 32 + r) * K::W2STR + 8 * h + hb * 64;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const char* a = a0 + 32 * ss;
                const u32x2 lo = *(const u32x2*)a, hi = *(const u32x2*)(a + 16);
                const u32x4 af = {lo[0], lo[1], hi[0], hi[1]};
                y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), gb[ss], y[cb], 0, 0, 0);
            }
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < K::NCH; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < K::NCH) load_chunk(ch + 1);
        const char* st = smem + buf * K::STAGE;
        // ---- all H^T blocks of the chunk first (MFMAs back to back), then per block the vector work followed by its Y^T MFMAs -- block 1's GELU has no dependence
        //      on block 0's MFMAs, so the two pipes can overlap
        f32x16 acc[NHB];
#pragma unroll
        for (int hb = 0; hb < NHB; ++hb) first_product(acc[hb], st, hb);
#pragma unroll
        for (int hb = 0; hb < NHB; ++hb) second_product(acc[hb], st, hb);
        if (ch + 1 < K::NCH) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: + b2 + x, one bf16 rounding; the tile is transposed through LDS (now free) so that whole rows leave
    char* ot = smem;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int c = cb * 32 + 8 * i4 + 4 * h;
            if (c < C) {
                const u32x2 xb = *(const u32x2*)(p.x + tokc * C + c);
                const u32x2 bb = *(const u32x2*)(p.b2 + c);
                const float v0 = y[cb][4 * i4 + 0] + __uint_as_float(bb[0] << 16) + __uint_as_float(xb[0] << 16);
                const float v1 = y[cb][4 * i4 + 1] + __uint_as_float(bb[0] & 0xffff0000u) + __uint_as_float(xb[0] & 0xffff0000u);
                const float v2 = y[cb][4 * i4 + 2] + __uint_as_float(bb[1] << 16) + __uint_as_float(xb[1] << 16);
                const float v3 = y[cb][4 * i4 + 3] + __uint_as_float(bb[1] & 0xffff0000u) + __uint_as_float(xb[1] & 0xffff0000u);
                *(u32x2*)(ot + (wave * 32 + r) * K::OSTR + c * 2) = u32x2{pack_bf2(v0, v1), pack_bf2(v2, v3)};
            }
        }
    __syncthreads();
    for (int idx = tid; idx < K::TOK * (C / 8); idx += NT) {
        const int row = idx / (C / 8), c16 = idx % (C / 8);
        if (tok0 + row < p.M) *(u32x4*)(p.y + (tok0 + row) * C + c16 * 8) = *(const u32x4*)(ot + row * K::OSTR + c16 * 16);
    }
}


// ================================================================================================ stage 2: C = 288, one wave per SIMD
// X^T (72) + Y^T (144) + two H accumulator sets (32) registers per lane do not fit 256: 4 waves x 32 tokens on the 512-register budget, hidden chunks of 32.
// A lone in-order wave overlaps matrix and vector work only where the two are interleaved in its instruction stream (an MFMA holds the issue port for 8 of its 32
// cycles), so the loop is hand-scheduled: a HALF-TRIP of chunk n is
//     the first six W1' fragments of chunk n + 1 + the fold constants of chunk n read up front (one LDS latency, not one per MFMA), the rest six MFMAs ahead
//  -> { 1 MFMA of H(n + 1) ; 1 W2 fragment of chunk n ; one step of GELU(n) } x 18      (sched_barrier walls pin the order)
//  -> { 1 MFMA of Y += W2(n) G(n) ; one LDS-DMA piece of a later chunk } x 18
// Weight chunks come from PACKED images in global memory (rga3_hiera_mlp288_pack: per chunk the exact LDS bytes -- W1' rows padded to 592 B, W2 rows to 72 B, then c
// and d as f32 -- so a chunk is 19 + 21 one-KiB LDS-DMA pieces, lane-linear, no staging registers, no ds_write, no address arithmetic).  W1' and W2 of the same chunk
// are read one half-trip apart, so they live in SEPARATE three-slot rings: W1'(c) [slot c % 3] is read in half-trip c - 1 and refilled from half-trip c - 3 on;
// W2(c) is read in half-trip c and refilled from c - 2 on.  Half-trip n issues W1'(n + 3) and W2(n + 2): every piece has a whole half-trip to land before the counted
// wait in front of the barrier that precedes its first read (register-staged, the same data was waited for INSIDE the half-trip that fetched it: 0.33 -> 0.22 ms
// per block and 8 frames with the staging ablated).
namespace hm288 {
constexpr int C = 288, H = 1152, HC = 32, NCH = H / HC, KS = C / 16, CB = C / 32, NW = 4, NT = 256, TOK = 128;
constexpr int W1STR = C * 2 + 16, W2STR = HC * 2 + 8;
constexpr int W1B = HC * W1STR;                     // 18 944
constexpr int W2B = C * W2STR;                      // 20 736, then c[32] d[32] f32
constexpr int W1P = 19, W2P = 21;                   // one-KiB pieces per chunk image
constexpr int W1S = W1P * 1024, W2S = W2P * 1024;   // ring slots = image sizes
constexpr int RW1 = 0, RW2 = 3 * W1S, TAB = 3 * W1S + 3 * W2S;
constexpr int LDS = TAB + kActTabBytes;             // 133 120
constexpr int OSTR = C * 2 + 16;
static_assert(W1B <= W1S && W2B + HC * 8 <= W2S && TOK * OSTR <= TAB && LDS <= 160 * 1024 && NCH % 2 == 0 && 2 * CB == KS, "hiera_mlp288: configuration");
}  // namespace hm288

struct Hm288Args {
    const unsigned short* x;   // [M, 288] bf16
    const char* pack;          // [36][W1S + W2S] bytes (rga3_hiera_mlp288_pack)
    const unsigned short* b2;  // [288] bf16
    unsigned short* y;         // [M, 288] bf16
    long M;
    float eps;
};

__global__ __launch_bounds__(hm288::NT) void hiera_mlp288_kernel(Hm288Args p) {
    using namespace hm288;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long tok0 = (long)blockIdx.x * TOK;
    const long tok = tok0 + wave * 32 + r;
    const long tokc = tok < p.M ? tok : p.M - 1;

    // ---- LDS-DMA of chunk images: piece pc of an image = 1 KiB, lane-linear on both sides; wave w takes pieces w, w + 4, ...
    const char* const psrc = p.pack + lane * 16;
    auto dma_w1_piece = [&](int ch, int j) {      // j-th piece of this wave
        const int pc = wave + 4 * j;
        if (pc < W1P) __builtin_amdgcn_global_load_lds((gbl_void*)(psrc + (long)ch * (W1S + W2S) + pc * 1024), (lds_void*)(smem + RW1 + (ch % 3) * W1S + pc * 1024), 16, 0, 0);
    };
    auto dma_w2_piece = [&](int ch, int j) {
        const int pc = wave + 4 * j;
        if (pc < W2P) __builtin_amdgcn_global_load_lds((gbl_void*)(psrc + (long)ch * (W1S + W2S) + W1S + pc * 1024), (lds_void*)(smem + RW2 + (ch % 3) * W2S + pc * 1024), 16, 0, 0);
    };
    constexpr int J1 = (W1P + 3) / 4, J2 = (W2P + 3) / 4;   // 5, 6 issue slots per wave
    // prologue: W1'(0..2), W2(0..1)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < J1; ++j) dma_w1_piece(c, j);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < J2; ++j) dma_w2_piece(c, j);

    // ---- this wave's 32 tokens as B operands: lane (r, h) holds x[tok][16 ks + 8 h .. + 8]
    bf16x8 xf[KS];
    {
        const unsigned short* xr = p.x + tokc * C + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(xr + 16 * ks);
    }
    float rinv, nmr;     // h = rinv acc + (d - mean rinv c)
    {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)xf[ks][e];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)xf[ks][e] - mean;
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        rinv = __builtin_amdgcn_rsqf(q * (1.0f / C) + p.eps);
        nmr = -mean * rinv;
    }
    // the ordinary loads above are consumed (the statistics read every fragment): no pending VGPR load is left for the compiler to drain the LDS-DMAs behind
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
    // the GELU table (10 KiB) behind the rings
    for (int i = tid; i < kActTabN; i += NT) *(unsigned*)(smem + TAB + i * 4) = g_act_tab[0][i];
    const unsigned tb = act_tab_base(smem + TAB);

    f32x16 y[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[cb][i] = 0.f;

    auto w1slot = [&](int c) { return smem + RW1 + (c % 3) * W1S; };
    auto w2slot = [&](int c) { return smem + RW2 + (c % 3) * W2S; };
    auto first_product = [&](f32x16& acc, const char* w1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const char* a0 = w1 + r * W1STR + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(a0 + ks * 32), xf[ks], acc, 0, 0, 0);
    };
    // half-trip of chunk n: accN = H(n + 1) from w1n, Y += W2(n) gelu(fold(accC)) from w2c; DMA of W1'(n + 3) and W2(n + 2) in the second product's gaps
    auto half = [&](f32x16& accN, const f32x16& accC, const char* w1n, const char* w2c, int n) {
        bf16x8 a1[KS];
        f32x4 cc[4], dd[4];
#ifndef HM288_PRE
#define HM288_PRE 6
#endif
        constexpr int PRE = HM288_PRE;     // W1' fragments read ahead of their MFMA: all 18 up front is 104 KiB per CU = 400 cycles of the LDS array with nothing to overlap
        const char* const a0 = w1n + r * W1STR + h * 16;
        {
#pragma unroll
            for (int ks = 0; ks < PRE; ++ks) a1[ks] = *(const bf16x8*)(a0 + ks * 32);
            const float* cs = (const float*)(w2c + W2B);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                cc[i4] = *(const f32x4*)(cs + 8 * i4 + 4 * h);
                dd[i4] = *(const f32x4*)(cs + HC + 8 * i4 + 4 * h);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) accN[i] = 0.f;
        u32x4 a2[CB][2];
        unsigned pkv[8], gk[8];
        float q[16];
        const char* w2b = w2c + r * W2STR + 8 * h;
        // GELU of value pair pp (accumulator registers 2 pp, 2 pp + 1) in two steps: A(pp) at MFMA 2 pp: LayerNorm fold, bf16 rounding, table addresses, the two
        // ds_read_b32; B(pp) three MFMAs later: relu - |t| T(|t|), bf16 pair
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            accN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[k], xf[k], accN, 0, 0, 0);
            if (k + PRE < KS) a1[k + PRE] = *(const bf16x8*)(a0 + (k + PRE) * 32);
            {
                const char* b0 = w2b + (k >> 1) * 32 * W2STR + 32 * (k & 1);
                const u32x2 lo = *(const u32x2*)b0, hi = *(const u32x2*)(b0 + 16);
                a2[k >> 1][k & 1] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            if ((k & 1) == 0 && (k >> 1) < 8) {
                const int pp = k >> 1, i4 = pp >> 1, e = 2 * (pp & 1);
                const float v0 = __builtin_fmaf(rinv, accC[2 * pp], __builtin_fmaf(nmr, cc[i4][e], dd[i4][e]));
                const float v1 = __builtin_fmaf(rinv, accC[2 * pp + 1], __builtin_fmaf(nmr, cc[i4][e + 1], dd[i4][e + 1]));
                unsigned pk = pack_bf2(v0, v1);
                asm("" : "+v"(pk));
                constexpr unsigned LO = kActTabLoBits, HI = kActTabLoBits + kActTabN - 1;
                const unsigned m0 = pk & 0x7fffu, m1 = __builtin_amdgcn_ubfe(pk, 16, 15);
                const unsigned ad0 = (min(max(m0, LO), HI) << 2) + tb, ad1 = (min(max(m1, LO), HI) << 2) + tb;
                q[2 * pp] = *(lds_cfloat*)(size_t)ad0;
                q[2 * pp + 1] = *(lds_cfloat*)(size_t)ad1;
                pkv[pp] = pk;
            }
            if (k >= 3 && (k & 1) == 1 && ((k - 3) >> 1) < 8) {
                const int pp = (k - 3) >> 1;
                const unsigned t0 = pkv[pp] << 16, t1 = pkv[pp] & 0xffff0000u;
                const int r0 = max((int)t0, 0), r1 = max((int)t1, 0);
                float y0, y1;
                asm("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(y0) : "v"(t0), "v"(q[2 * pp]), "v"(r0));
                asm("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(y1) : "v"(t1), "v"(q[2 * pp + 1]), "v"(r1));
                gk[pp] = pack_bf2(y0, y1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        bf16x8 gb[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) gb[ss] = __builtin_bit_cast(bf16x8, u32x4{gk[4 * ss], gk[4 * ss + 1], gk[4 * ss + 2], gk[4 * ss + 3]});
        const bool f1 = n + 3 < NCH, f2 = n + 2 < NCH;     // wave-uniform
#pragma unroll
        for (int k = 0; k < 2 * CB; ++k) {
            y[k >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2[k >> 1][k & 1]), gb[k & 1], y[k >> 1], 0, 0, 0);
            if (k < J1) { if (f1) dma_w1_piece(n + 3, k); }
            else if (k < J1 + J2) { if (f2) dma_w2_piece(n + 2, k - J1); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // what the NEXT half-trip reads (W1'(n + 2), W2(n + 1)) was issued one half-trip ago or earlier: everything but this half-trip's own pieces must have landed.
        // Pieces per wave and half-trip: 5 + 6 (wave 0), 5 + 5 (waves 1, 2), 4 + 5 (wave 3); in the last three half-trips fewer are issued: drain.
        if (f1) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else if (wave == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 accA, accB;
    first_product(accA, w1slot(0));
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch += 2) {
        half(accB, accA, w1slot(ch + 1), w2slot(ch), ch);
        half(accA, accB, w1slot(ch + 2), w2slot(ch + 1), ch + 1);     // (the last trip's H(NCH) reads a stale slot: computed, never used)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- epilogue: + b2 + x, one bf16 rounding; the tile is transposed through LDS (the rings are free) so that whole rows leave
    char* ot = smem;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int c = cb * 32 + 8 * i4 + 4 * h;
            const u32x2 xb = *(const u32x2*)(p.x + tokc * C + c);
            const u32x2 bb = *(const u32x2*)(p.b2 + c);
            const float v0 = y[cb][4 * i4 + 0] + __uint_as_float(bb[0] << 16) + __uint_as_float(xb[0] << 16);
            const float v1 = y[cb][4 * i4 + 1] + __uint_as_float(bb[0] & 0xffff0000u) + __uint_as_float(xb[0] & 0xffff0000u);
            const float v2 = y[cb][4 * i4 + 2] + __uint_as_float(bb[1] << 16) + __uint_as_float(xb[1] << 16);
            const float v3 = y[cb][4 * i4 + 3] + __uint_as_float(bb[1] & 0xffff0000u) + __uint_as_float(xb[1] & 0xffff0000u);
            *(u32x2*)(ot + (wave * 32 + r) * OSTR + c * 2) = u32x2{pack_bf2(v0, v1), pack_bf2(v2, v3)};
        }
    __syncthreads();
    for (int idx = tid; idx < TOK * (C / 8); idx += NT) {
        const int row = idx / (C / 8), c16 = idx % (C / 8);
        if (tok0 + row < p.M) *(u32x4*)(p.y + (tok0 + row) * C + c16 * 8) = *(const u32x4*)(ot + row * OSTR + c16 * 16);
    }
}

template <class K>
static int launch_hiera_mlp(const HmArgs& a, hipStream_t st, const char* name) {
    auto kern = hiera_mlp_kernel<K>;
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)kern, K::LDS, lds_grant, name)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(a.M, K::TOK)), dim3(K::NT), K::LDS, st, a);
    RGA3_CHECK_LAUNCH(name);
    return 0;
}

}  // namespace rga3

using namespace rga3;

// y [M, 144] = x + W2 gelu(LayerNorm(x; eps) folded into W1f / c1 / d1) + b2   (Hiera-L stage-1 MLP, 144 -> 576 -> 144), all bf16 except c1 (f32).  w1f / c1 / d1 as
// rga3_gemm_ln_bf16 takes them: W1 diag(gamma) rounded to bf16, its row sums in f32, beta W1^T + b1 in bf16.  x, y contiguous, 16-byte aligned.
extern "C" int rga3_hiera_mlp144(const void* x, const void* w1f, const float* c1, const void* d1, const void* w2, const void* b2, void* y, int64_t M, float eps,
                                 void* stream) {
    RGA3_CHECK_ARG(x && w1f && c1 && d1 && w2 && b2 && y && M > 0, "hiera_mlp144: null pointer / M %ld", (long)M);
    RGA3_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1f | (uintptr_t)w2 | (uintptr_t)y | (uintptr_t)c1) & 15) == 0 && (((uintptr_t)b2 | (uintptr_t)d1) & 7) == 0, "hiera_mlp144: alignment");
    RGA3_CHECK_ARG(x != y, "hiera_mlp144: in place is not supported (the residual is re-read)");
    HmArgs a;
    a.x = (const unsigned short*)x; a.w1f = (const unsigned short*)w1f; a.c1 = c1; a.d1 = (const unsigned short*)d1;
    a.w2 = (const unsigned short*)w2; a.b2 = (const unsigned short*)b2; a.y = (unsigned short*)y; a.M = M; a.eps = eps;
    return launch_hiera_mlp<HmCfg<144, 8, 64, 2>>(a, (hipStream_t)stream, "hiera_mlp_kernel<144>");
}

// Stage 2 (288 -> 1152 -> 288).  The weights are handed over PACKED, once per block (they are frozen): rga3_hiera_mlp288_pack_bytes() bytes, 36 chunk images of
// 19 + 21 KiB; image of hidden chunk ch (32 hidden units):
//   [0, 18 944)            W1' rows 32 ch .. 32 ch + 31: 288 bf16 each, row stride 592 B (16 B of padding)
//   [19 456, 19 456 + 20 736)   W2 rows 0 .. 287, columns 32 ch .. 32 ch + 31: 32 bf16 each, row stride 72 B (8 B of padding)
//   then 32 f32 c (row sums of W1'), 32 f32 d (folded bias)         (unused bytes: anything)
// rga3_hiera_mlp288_pack builds it on the device from the operands of the C = 144 form.
extern "C" int64_t rga3_hiera_mlp288_pack_bytes(void) { return (int64_t)hm288::NCH * (hm288::W1S + hm288::W2S); }

__global__ void hiera_mlp288_pack_kernel(const unsigned short* w1f, const float* c1, const unsigned short* d1, const unsigned short* w2, char* pack) {
    using namespace hm288;
    const int ch = blockIdx.x;
    char* img = pack + (long)ch * (W1S + W2S);
    for (int i = threadIdx.x; i < (W1S + W2S) / 4; i += blockDim.x) ((unsigned*)img)[i] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < HC * C; i += blockDim.x) {
        const int row = i / C, col = i % C;
        *(unsigned short*)(img + row * W1STR + col * 2) = w1f[(long)(ch * HC + row) * C + col];
    }
    for (int i = threadIdx.x; i < C * HC; i += blockDim.x) {
        const int row = i / HC, col = i % HC;
        *(unsigned short*)(img + W1S + row * W2STR + col * 2) = w2[(long)row * H + ch * HC + col];
    }
    if (threadIdx.x < HC) {
        ((float*)(img + W1S + W2B))[threadIdx.x] = c1[ch * HC + threadIdx.x];
        ((float*)(img + W1S + W2B))[HC + threadIdx.x] = bf2f(d1[ch * HC + threadIdx.x]);
    }
}

extern "C" int rga3_hiera_mlp288_pack(const void* w1f, const float* c1, const void* d1, const void* w2, void* pack, void* stream) {
    RGA3_CHECK_ARG(w1f && c1 && d1 && w2 && pack && (((uintptr_t)pack) & 15) == 0, "hiera_mlp288_pack: null pointer / alignment");
    hipLaunchKernelGGL(hiera_mlp288_pack_kernel, dim3(hm288::NCH), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)w1f, c1, (const unsigned short*)d1,
                       (const unsigned short*)w2, (char*)pack);
    RGA3_CHECK_LAUNCH("hiera_mlp288_pack_kernel");
    return 0;
}

// y [M, 288] = x + W2 gelu(LayerNorm(x; eps) W1^T + b1) + b2 from the packed weights; x, y contiguous bf16, 16-byte aligned, x != y.
extern "C" int rga3_hiera_mlp288(const void* x, const void* pack, const void* b2, void* y, int64_t M, float eps, void* stream) {
    RGA3_CHECK_ARG(x && pack && b2 && y && M > 0, "hiera_mlp288: null pointer / M %ld", (long)M);
    RGA3_CHECK_ARG((((uintptr_t)x | (uintptr_t)pack | (uintptr_t)y) & 15) == 0 && (((uintptr_t)b2) & 7) == 0, "hiera_mlp288: alignment");
    RGA3_CHECK_ARG(x != y, "hiera_mlp288: in place is not supported (the residual is re-read)");
    static LdsGrant lds_grant;
    if (int rc = grant_dyn_lds((const void*)hiera_mlp288_kernel, hm288::LDS, lds_grant, "hiera_mlp288")) return rc;
    Hm288Args a;
    a.x = (const unsigned short*)x; a.pack = (const char*)pack; a.b2 = (const unsigned short*)b2; a.y = (unsigned short*)y; a.M = M; a.eps = eps;
    hipLaunchKernelGGL(hiera_mlp288_kernel, dim3((unsigned)cdiv(M, hm288::TOK)), dim3(hm288::NT), hm288::LDS, (hipStream_t)stream, a);
    RGA3_CHECK_LAUNCH("hiera_mlp288_kernel");
    return 0;
}
