/* AddressSanitizer / UBSan harness for the HOST-ONLY C entry points of librga3_hip: built and run on the CPU
   only, by sanitize/Makefile, with gcc and csrc/host_*.cpp (plain C++, no device code).  This directory is listed in .gpurunignore: it never travels to
   the GPU box. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
int rga3_pil_bicubic_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int64_t kk_capacity, int* ksize_out);
int rga3_qwen_norm_lut(const float* mean3, const float* std3, int fused, float* lut768);
int main(void) {
    static const int sizes[][2] = {{480, 1024}, {854, 1024}, {1024, 1024}, {2160, 1024}, {7, 1024}, {1, 5}, {4000, 28}, {28, 4000}, {333, 777}, {1024, 1}};
    for (unsigned i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
        int ks = 0;
        if (rga3_pil_bicubic_coeffs(sizes[i][0], sizes[i][1], NULL, NULL, 0, &ks) != 0 || ks <= 0) { printf("query failed %d %d\n", sizes[i][0], sizes[i][1]); return 1; }
        int32_t* b = malloc(sizeof(int32_t) * 2 * sizes[i][1]);
        int32_t* k = malloc(sizeof(int32_t) * (size_t)ks * sizes[i][1]);
        if (rga3_pil_bicubic_coeffs(sizes[i][0], sizes[i][1], b, k, (int64_t)ks * sizes[i][1], &ks) != 0) {
            if (ks > 512) { free(b); free(k); continue; }   /* a filter wider than the entry point's tap buffer must be refused, not overrun */
            printf("fill failed\n"); return 1;
        }
        /* exact-capacity buffers: any write past the end trips ASAN; a too-small capacity must be refused, not overrun */
        if (rga3_pil_bicubic_coeffs(sizes[i][0], sizes[i][1], b, k, (int64_t)ks * sizes[i][1] - 1, &ks) == 0) { printf("short capacity accepted\n"); return 1; }
        for (int o = 0; o < sizes[i][1]; ++o)
            if (b[2 * o] < 0 || b[2 * o] + b[2 * o + 1] > sizes[i][0]) { printf("bounds out of range\n"); return 1; }
        free(b); free(k);
    }
    float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, std[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    float* lut = malloc(sizeof(float) * 768);
    for (int fused = 0; fused < 2; ++fused)
        if (rga3_qwen_norm_lut(mean, std, fused, lut) != 0) { printf("lut failed\n"); return 1; }
    free(lut);
    printf("ok\n");
    return 0;
}
