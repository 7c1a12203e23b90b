// Which waves of a 512-thread workgroup share a SIMD on this part?  Two waves of one workgroup run a bare MFMA chain, the other six exit at once; if the pair shares a
// SIMD the chain takes twice as long.  Also prints the SIMD_ID field of HW_REG_HW_ID per wave.   hipcc --offload-arch=gfx950 -O2 -o simd_pairing simd_pairing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void k(int wa, int wb, int iters, float* out, int* hw) {
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) hw[wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    if (wave != wa && wave != wb) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out; int* hw;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&hw, 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wb = 0; wb < 8; ++wb) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, 0, wb, 4000, out, hw);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("waves 0 and %d: %.3f ms\n", wb, ms);
        }
    }
    int h[8]; hipMemcpy(h, hw, 32, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) printf("wave %d HW_ID 0x%08x simd[5:4]=%d wave_id[3:0]=%d\n", w, h[w], (h[w] >> 4) & 3, h[w] & 15);
    return 0;
}
