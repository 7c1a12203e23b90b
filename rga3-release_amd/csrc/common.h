// Shared device/host helpers for the rga3 HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>

#include "errors.h"

namespace rga3 {

#define RGA3_CHECK_LAUNCH(name)                                                                  \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess) return ::rga3::fail(-(int)e__, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that launches on a second GPU must set it there too
// (ADVICE r3).  One LdsGrant per kernel instantiation (a function-local static at the launch site) remembers the bytes granted on each device.
struct LdsGrant { int bytes[16] = {0}; };
inline int grant_dyn_lds(const void* kern, int bytes, LdsGrant& g, const char* name) {
    if (bytes <= 48 * 1024) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const bool tracked = dev < 16;
    if (tracked && g.bytes[dev] >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(-(int)e, "%s: hipFuncSetAttribute(%d B dynamic LDS): %s", name, bytes, hipGetErrorString(e));
    if (tracked) g.bytes[dev] = bytes;
    return 0;
}

// ---- vector types
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned int)u) << 16); }
// plain cast keeps NaN a NaN and lowers to v_cvt_pk_bf16_f32 (MI355X_MICROARCH.md, correctness boundaries)
__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// SwiGLU element arithmetic shared by the plain kernels (trainops.hip) and the ones with the e4m3 quantiser fused in (gemm_fp8.hip): one definition,
// contraction off, so both produce the same bits.
__device__ __forceinline__ float swiglu_fwd_elem(float g, float u) {
#pragma clang fp contract(off)
    const float silu = g / (1.f + __expf(-g));
    return __uint_as_float(((unsigned)f2bf(silu)) << 16) * u;
}
__device__ __forceinline__ void swiglu_bwd_elem(float g, float u, float d, float& dg, float& du) {
#pragma clang fp contract(off)
    const float sg = 1.f / (1.f + __expf(-g));
    const float silu = g * sg;
    du = d * silu;
    dg = d * u * sg * (1.f + g * (1.f - sg));
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// number of XCDs on MI355X (MI355X_MICROARCH.md chip table); used only for speed (L2 affinity), never correctness
constexpr int kNumXCD = 8;

// Bijective XCD-aware block remap (cdna_hip_programming.md 5, "XCD swizzle must be bijective"):
// blocks that share an XCD (same id % 8) get a contiguous chunk of the logical tile range.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    unsigned q = nwg / kNumXCD, r = nwg % kNumXCD, x = bid % kNumXCD;
    unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + bid / kNumXCD;
}

// Logical (x, y, z) of this workgroup for a launch flattened to a 1-D grid of gx * gy * gz workgroups: the id is XCD-remapped first, then decoded
// y-fastest, so consecutive logical ids -- which share an XCD -- differ in y (attention: the heads of one query / key block and segment).
struct Wg3 { int x, y, z; };
__device__ __forceinline__ Wg3 xcd_decode3(int gx, int gy) {
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    Wg3 w;
    w.y = (int)(lid % (unsigned)gy);
    w.x = (int)((lid / (unsigned)gy) % (unsigned)gx);
    w.z = (int)(lid / ((unsigned)gy * (unsigned)gx));
    return w;
}

}  // namespace rga3
