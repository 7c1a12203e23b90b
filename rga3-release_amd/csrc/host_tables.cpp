// Host-only entry points of librga3_hip.so: the coefficient / normalisation tables the GPU input pipeline (preproc.hip) consumes.  Plain C++ with no
// device code and no HIP headers, so the same file builds under the CPU AddressSanitizer / UBSan (sanitize/Makefile) -- GPU sanitizers are not
// available on this pool.  Pillow restatement: src/libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc (reference utils/utils.py:246-256
// calls Image.resize -> antialiased bicubic).
#include "errors.h"

#include <cmath>

namespace rga3 {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // must equal preproc.hip's

static inline double bicubic_filter(double x) {
#pragma clang fp contract(off)  // Pillow's C is compiled without FMA contraction: keep the double rounding identical
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace rga3

using namespace rga3;

// Host-only: Pillow's precompute_coeffs + normalize_coeffs_8bpc for the full box.  bounds [out*2] = (first index, taps), kk
// [out*ksize].  With bounds == NULL only *ksize_out is written (size query).  Returns 0 or a negative code.
extern "C" int rga3_pil_bicubic_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int64_t kk_capacity, int* ksize_out) {
#pragma clang fp contract(off)
    RGA3_CHECK_ARG(in_size > 0 && out_size > 0 && ksize_out, "coeffs: sizes %d -> %d", in_size, out_size);
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!bounds) return 0;
    RGA3_CHECK_ARG(kk && kk_capacity >= (int64_t)out_size * ksize, "coeffs: kk capacity %ld < %ld", (long)kk_capacity, (long)out_size * ksize);
    const double ss = 1.0 / filterscale;
    double wbuf[512];
    RGA3_CHECK_ARG(ksize <= 512, "coeffs: filter too wide (%d taps)", ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            wbuf[x] = w;
            ww += w;
        }
        int32_t* k = kk + (int64_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = (ww != 0.0) ? wbuf[x] / ww : wbuf[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return 0;
}

// Host-only: the 3 x 256 table of normalised values, lut[c*256 + b], in the reference's fp32 operation order.
//   fused == 0: transformers 4.49 (the reference's pin, requirements.txt:26) image_transforms: rescale = float32(float64(b) * (1/255)),
//               then (x - mean) / std in float32;
//   fused == 1: the installed 5.x fast path (image_processing_backends.py:298-337): (float32(b) - mean*255) / (std*255) with the
//               products formed in float32.
extern "C" int rga3_qwen_norm_lut(const float* mean3, const float* std3, int fused, float* lut768) {
#pragma clang fp contract(off)
    RGA3_CHECK_ARG(mean3 && std3 && lut768, "norm_lut: null argument");
    for (int c = 0; c < 3; ++c)
        for (int b = 0; b < 256; ++b) {
            float v;
            if (fused) {
                const float m = mean3[c] * 255.0f, s = std3[c] * 255.0f;
                v = ((float)b - m) / s;
            } else {
                const float x = (float)((double)b * (1.0 / 255.0));
                v = (x - mean3[c]) / std3[c];
            }
            lut768[c * 256 + b] = v;
        }
    return 0;
}
