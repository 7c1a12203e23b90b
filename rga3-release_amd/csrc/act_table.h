// Table-driven activations shared by the GEMM tile epilogues (gemm_bf16.hip) and the fused Hiera MLP kernels (hiera_mlp.hip).
#pragma once
#include "common.h"

namespace rga3 {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- activations of the tile epilogues: table-driven.  The value t fed to GELU / SiLU has just been rounded to bf16 (the reference's bf16 nn.Linear output), so the
//      activation is a function of 15 magnitude bits:   act(t) = relu(t) - |t| T(|t|),   T_gelu(a) = Phi(-a),  T_silu(a) = sigmoid(-a)   (tools/gen_act_tables.py:
//      f32 T for every bf16 magnitude in [2^-14, 2^6), 10 KiB; outside the range the clamped entry is exact to < 3e-5 relative / 1e-26 absolute).  Five VALU
//      slots and one ds_read_b32 per element instead of a quarter-rate v_rcp + v_exp and ten more (A-S erf) or 2 + 3 (sigmoid): the K = 576 GELU product of Hiera
//      stage 3 was 70 % VALU-busy (profiles/r04_k576_pmc_before.json).  Exact to f32 rounding, including the negative tail's relative accuracy; NaN stays NaN; an
//      infinite input (an activation that has already overflowed bf16) gives NaN where the closed form gives +inf / -0 (|inf| x the table's final 0).
#include "act_tables.inc"
constexpr int kActTabBytes = kActTabN * 4;
static_assert(kActTabBytes % 1024 == 0, "the table is staged in 1-KiB LDS-DMA pieces");

typedef const __attribute__((address_space(3))) float lds_cfloat;
// LDS byte address of the table minus the bytes of the magnitudes below its first entry (wave-uniform; computed once per epilogue)
__device__ __forceinline__ unsigned act_tab_base(const char* tab) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)tab - kActTabLoBits * 4u; }
// two activations from a PACKED bf16 pair (what v_cvt_pk_bf16_f32 just produced).  Per element: magnitude (v_and / v_bfe), clamp to the table (v_med3_u32), byte address
// (v_lshl_add_u32), ds_read_b32, relu as a signed-integer max (no canonicalising v_max_f32 pair), one v_fma_f32 with |t| and the negation as source modifiers (asm:
// left to itself the compiler pairs two elements into v_pk_fma_f32, which has no |x| modifier, and pays two extra v_and).
__device__ __forceinline__ f32x2 act_tab2(unsigned pk, unsigned tb) {
    float y0, y1;
    constexpr unsigned LO = kActTabLoBits, HI = kActTabLoBits + kActTabN - 1;
    asm("" : "+v"(pk));   // opaque: the compiler otherwise re-derives the low half by a second, single v_cvt_pk_bf16_f32
    const unsigned m0 = pk & 0x7fffu, m1 = __builtin_amdgcn_ubfe(pk, 16, 15);
    const unsigned a0 = (min(max(m0, LO), HI) << 2) + tb, a1 = (min(max(m1, LO), HI) << 2) + tb;
    const float q0 = *(lds_cfloat*)(size_t)a0, q1 = *(lds_cfloat*)(size_t)a1;
    const unsigned t0 = pk << 16, t1 = pk & 0xffff0000u;
    const int r0 = max((int)t0, 0), r1 = max((int)t1, 0);
    asm("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(y0) : "v"(t0), "v"(q0), "v"(r0));
    asm("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(y1) : "v"(t1), "v"(q1), "v"(r1));
    return f32x2{y0, y1};
}

}  // namespace rga3
