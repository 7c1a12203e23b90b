/*
 * rga3_hip.h — C ABI of librga3_hip.so, the MI355X (gfx950) kernel library behind the RGA3 / UniGR hot path.
 *
 * The reference (qirui-chen/RGA3-release) is pure Python and reaches all arithmetic through third-party
 * CUDA libraries (flash-attn, cuBLAS, cuDNN via torch).  Each entry point below replaces one such call
 * site; the reference file:line it stands in for is cited on every declaration ("HF" = the transformers
 * Qwen2.5-VL modeling file the reference subclasses, model/qwen_2_5_vl_sam2.py:9-12,104).
 *
 * Conventions:
 *  - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (PyTorch caching
 *    allocator), including workspaces.  The library allocates nothing and keeps no mutable global state.
 *  - every call enqueues on `stream` (a hipStream_t passed as void*) and returns without synchronising.
 *  - return 0 on success, a negative code otherwise (-hipError_t for runtime errors, RGA3_EINVAL for
 *    argument errors); the message is retrievable with rga3_last_error() (thread-local).
 *  - bf16 tensors are passed as raw 16-bit words; "ld*" / strides are in ELEMENTS.
 */
#ifndef RGA3_HIP_H
#define RGA3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGA3_EINVAL (-22)

/* dtype codes */
#define RGA3_BF16 0
#define RGA3_F32 1

/* GEMM epilogue activation codes */
#define RGA3_ACT_NONE 0
#define RGA3_ACT_GELU 1   /* exact erf GELU (nn.GELU default) */
#define RGA3_ACT_SWIGLU 2 /* W rows interleaved [16 gate | 16 up]; out[:, j] = silu(gate_j) * up_j, N_out = N/2 */
#define RGA3_ACT_RELU 3

int rga3_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
int rga3_last_error(char* buf, size_t n);

/* C[M,N] = residual + colscale * act(A[M,K] . W[N,K]^T + bias)  (bias / residual / colscale optional).
 * bf16 in, f32 accumulate, bf16 or f32 out.
 * Replaces nn.Linear / 1x1 conv / Conv3d-as-GEMM call sites: HF modeling_qwen2_5_vl.py:84-96 (MLP),
 * :99-122 (patch embed), :137-150 (merger), :211-291 (ViT qkv/proj), :602-757 (decoder projections),
 * :1383 (lm_head); reference model/qwen_2_5_vl_sam2.py:131-137 (text_hidden_fcs), model/sam2.py:986-1117
 * (Hiera qkv/proj/mlp), :857-889 (FPN 1x1), :1417-1481 (decoder attention projections).
 * colscale [N_out] is the ConvNeXt layer scale of reference model/sam2.py:690-703.
 * K, lda, ldw must be multiples of 8 (16-byte rows); tile = -1 lets the library choose (M <= 4 rows: the weight-stream kernel, 40; 5 - 16 rows: the token-row
 * kernel, 41).
 * Explicit tilings (all give the same bits except 22 / 32 / 25 / 14 / 40 / 41, whose K splits / lane-strided sums change the f32 summation order,
 * reproducibly):
 *   20 = 256x256 ping-pong, one tile per workgroup;  21 = the same, persistent (one workgroup per CU);
 *   22 = persistent + stream-K tail (needs the caller workspace below);  27 / 26 = 21 / 22 whose last tile row, when it holds <= 64 rows (M = 2112 = 8 x 256 + 64),
 *   runs a quarter-work loop on workgroups of its own (as 21 / 22 for other M);  31 / 32 = 21 / 22 with 192x256 tiles (M = 2112 = 11 x 192);
 *   28 = 256x256 on FOUR waves (128x128 wave blocks, accumulators in the AGPRs, hand-ordered MFMA / fragment-read / LDS-DMA stream), one tile per workgroup,
 *        K a multiple of 64 (runs as 20 otherwise): bit-identical to 20, faster on long K (8192^3: 1 465 against 1 369 TFLOP/s), slower below K ~ 2000;
 *   25 = split-K for few output tiles over a very long K (same workspace; falls back to 21 when it does not apply);
 *   11, 12 / 3 / 4 / 5 / 13 = single-phase 128x128 / 128x256 / 128x320 / 128x192 / 64x64;  10 = single-phase 256x256 (first generation, A/B);
 *   14 = 64x64 with K cut into up to 32 slices (skinny plain products such as LoRA's x A^T: N = 128 over K = 3584; same workspace; runs as 13
 *        when there is a bias / activation / residual);
 *   40 = skinny weight-stream kernel for M <= 4 (the decode step of generate(), reference app.py:308-317): no column scale;
 *   41 = token rows, M <= 16 (the 9 decoder tokens of reference model/sam2.py:1926-2100 TwoWayTransformer: a workgroup per 16 output columns, K split over its
 *        8 waves): bf16 output, no column scale, no SwiGLU. */
int rga3_gemm_bf16(const void* A, const void* W, const void* bias, const void* residual, const void* colscale, void* C, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, int act, int out_dtype, int tile,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* rga3_gemm_bf16 with an RMSNorm folded around it: replaces the norm launch + the write and re-read of the normalised rows between a residual-producing
 * projection and the projection that consumes the normalised rows (HF modeling_qwen2_5_vl.py: Qwen2RMSNorm :470-486 between o_proj / down_proj and
 * q|k|v / gate|up, :602-757; Qwen2_5_VLVisionBlock norm1 / norm2, :293-321).
 *  consumer (row_sumsq_in != NULL): A = the un-normalised rows, W = weight with the norm weight folded in (W diag(gamma)); the accumulators are scaled by
 *    1 / sqrt(row_sumsq_in[r] / 2^20 / norm_width + eps) before bias / activation (act as rga3_gemm_bf16, incl. SwiGLU);
 *  producer (row_sumsq_out != NULL): the sums of squares of the bf16 rows written to C are ADDED to row_sumsq_out[r] as 2^20 fixed-point integers by no-return
 *    integer atomics (the caller zeroes the array before the launch; any arrival order gives the same bits).
 * bf16 output, M > 16, tiles with the shared epilogue (not 14 / 25 / 40 / 41). */
int rga3_gemm_rms_bf16(const void* A, const void* W, const void* bias, const void* residual, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int64_t ldw, int64_t ldc, int64_t ldr, int act, int tile, void* workspace, int64_t workspace_bytes, const uint64_t* row_sumsq_in,
                       int64_t norm_width, float eps, uint64_t* row_sumsq_out, void* stream);
/* RGA3_ACT_SWIGLU product that also stores its pre-activations (round 4; training forward of the decoder MLP -- HF Qwen2MLP under autograd, reference
 * train_joint.py:534): C [M, N / 2] = silu(gate) * up, pre [M, N] (row stride ldpre) = the interleaved gate | up linear outputs rounded to bf16, which the backward
 * of the activation reads (rga3_swiglu_bwd).  Replaces rga3_gemm_bf16 (plain) + rga3_swiglu_fwd. */
int rga3_gemm_swiglu_pre_bf16(const void* A, const void* W, const void* bias, void* C, void* pre, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                              int64_t ldc, int64_t ldpre, int tile, void* workspace, int64_t workspace_bytes, void* stream);
/* Concatenated operands (round 4): LoRA's low-rank products folded into the frozen products of a decoder layer (PEFT LoRA layer, reference train_joint.py:193-232,
 * run_torchrun.sh:30-31; y = W x + s B A dropout(x)).
 *   K side (A2 [M, K2], W2 [N, K2]):  C [M, N] = [A | A2] . [W | W2]^T (+ bias)      forward:  qkv = h W^T + [t_q | t_v] [B_q 0; 0 0; 0 B_v]^T as ONE product
 *   N side (Wn [N2, K], Cn [M, N2]):  also Cn = A . Wn^T from the same launch        backward: [dh | dt_q | dt_v] = dqkv [W | sB_q | sB_v]
 * bf16, no activation; K, K2 multiples of 64; with an N side N must be a multiple of the tile width.  tile: -1 / 12 (128 x 128), 3 / 6 (128 x 256), 13 (64 x 64); K side
 * only: also 4 (128 x 320), 5 (128 x 192).  Single-phase kernels (no workspace). */
int rga3_gemm_cat_bf16(const void* A, const void* W, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc,
                       const void* A2, const void* W2, int64_t K2, int64_t lda2, int64_t ldw2, const void* Wn, void* Cn, int64_t N2, int64_t ldwn, int64_t ldcn,
                       int tile, void* stream);

/* bytes of caller workspace tiles 22 / 25 want on the current device (4 KiB of flags + 256 KiB per CU; 0 on error).  The workspace is
 * zeroed ONCE by the caller and then kept for the calls of one stream (the kernels leave the flag words zero again); without it (NULL / too
 * small) those tilings run as 21 / 20. */
int64_t rga3_gemm_workspace_bytes(void);
/* C[M,N] (bf16 / f32) = A^T . B (+ bias[n]); A [K,M], B [K,N] bf16 row-major (K = tokens): the weight-gradient product dW = dY^T X of
 * nn.Linear / LoRA (autograd under reference train_joint.py:534) without transposing the activations first.  M, N multiples of 8.
 * workspace (optional, caller-owned, 16-byte aligned): with few output tiles over many tokens (LoRA dW: 128 x 3584 over 2112-4160 tokens) K is
 * cut into <= 64 slices whose f32 partial sums (Z * M * N * 4 bytes) are added in fixed slice order -- by a second launch, or, with `counters` (>= 128 32-bit
 * words, zeroed ONCE by the caller and kept for the calls of one stream: the kernel leaves them zero), inside the same launch by the workgroup that reaches an output
 * tile last (which one is timing, the order of the additions is not: deterministic). */
int rga3_gemm_tn_bf16(const void* A, const void* B, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                      int64_t ldc, int out_dtype, void* workspace, int64_t workspace_bytes, void* counters, void* stream);
/* n (<= 4) such products in ONE launch (+ one slab-sum launch): the four LoRA weight gradients of a decoder layer (dB_q, dA_q, dB_v, dA_v: autograd of PEFT's
 * lora_A / lora_B under reference train_joint.py:193-232, 534) were four launches of 1 - 28 tiles and four slab sums.  ptrs: n x 3 {A, B, C}; dims: n x 7
 * {M, N, K, lda, ldb, ldc, out_dtype}; HOST arrays.  Every product is K-split exactly as rga3_gemm_tn_bf16 with a full-size workspace splits it (same bits); the
 * caller's workspace must hold sum_i Z_i M_i N_i floats, Z_i = min(64, 256 / tiles_i, K_i / 256) (>= 1), or the call fails. */
int rga3_gemm_tn_many(const void* const* ptrs, const int64_t* dims, int n, void* workspace, int64_t workspace_bytes, void* stream);


/* Variable-length fused attention forward (online softmax, fp32 statistics), bf16 in/out.
 *   q: [total_q, Hq, D], k/v: [total_k, Hkv, D] addressed through (token, head) element strides so the
 *   kernel reads straight out of a fused QKV projection; segments given by cu_q / cu_k (int32, nseg+1).
 *   causal != 0: key j visible to query i iff j <= i + (Lk - Lq) (bottom-right aligned).
 *   lse (optional, f32 [Hq, total_q]) receives log-sum-exp of the scaled scores for the backward pass.
 * Replaces flash_attn_varlen_func / F.scaled_dot_product_attention at HF modeling_qwen2_5_vl.py:211-291
 * (ViT windows), :602-700 (causal GQA), reference model/sam2.py:1021 (Hiera), :1476 (two-way decoder),
 * :1543 (memory attention).  D in {16..256}, multiple of 8. impl: 0 = transposed LDS read, 1 = scalar-read
 * variant (debug cross-check). 
 * split_ws (optional, f32, split_ws_elems floats): non-causal calls whose grid would leave most CUs idle (query blocks x heads x
 * segments < 128) over a long key range (max_k >= 1024; max_k is only read for this decision) cut the keys into <= 8 slices, one
 * workgroup each, and merge them by their log-sum-exp in a second pass; needs nsplit * total_q * Hq * (D + 1) floats.
 * block_q / block_k (0 = off, else powers of two, non-causal): block-diagonal visibility inside a segment -- query i sees key j iff
 * i / block_q == j / block_k -- so that several tiny windows (Hiera's 4- and 16-token windows, model/sam2.py:986-1033) are packed into one
 * 64-row segment instead of one workgroup per window. */
int rga3_attn_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* cu_q,
                         const int32_t* cu_k, int nseg, int max_q, int64_t total_q, int Hq, int Hkv, int D, int64_t q_st, int64_t q_sh,
                         int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                         float scale, int causal, int impl, float* split_ws, int64_t split_ws_elems, int max_k, int block_q, int block_k, void* stream);
/* Windowed attention (segments of <= 64 queries: the ViT's 64-token windows) with RoPE applied WHILE q and k ARE LOADED: cos / sin [tokens, D] f32
 * tables indexed by the packed token, rotate-half pairing d <-> d +- D/2 (HF apply_rotary_pos_emb_vision modeling_qwen2_5_vl.py:160-171), D % 16 == 0,
 * D <= 128.  Every key of a window is loaded exactly once per head, so the stand-alone rga3_rope_inplace pass over q and k disappears (measured on
 * MI355X, 128 windows x 16 heads x 80: rope 21 us + attention 25.6 us -> 42 us).  cos_k / sin_k NULL: k is already rotated.  Longer segments are
 * rejected: there every query block would rotate the keys again, and rotating Q in the long-row kernel cost more than the pass it removed
 * (decoder S = 2112: 72 -> 98 us), so those call sites keep rga3_rope_inplace + rga3_attn_varlen_fwd. */
int rga3_attn_varlen_fwd_rope(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* cu_q, const int32_t* cu_k, int nseg, int max_q,
                              int64_t total_q, int Hq, int Hkv, int D, int64_t q_st, int64_t q_sh, int64_t k_st, int64_t k_sh, int64_t v_st, int64_t v_sh,
                              int64_t o_st, int64_t o_sh, float scale, int causal, const float* cos_q, const float* sin_q, const float* cos_k,
                              const float* sin_k, void* stream);

/* y = weight * (x * rsqrt(mean(x^2) + eps)) rounded to bf16 before the weight multiply, exactly as
 * HF Qwen2_5_VLRMSNorm (modeling_qwen2_5_vl.py:65-79).  x,y: [rows, dim] bf16; optional fused residual:
 * if `res_out` != NULL the kernel first writes res_out = x + add (bf16) and normalises that sum. */
int rga3_rmsnorm_fwd(const void* x, const void* add, const void* weight, void* y, void* res_out, int64_t rows,
                     int64_t dim, int64_t ldx, float eps, void* stream);

/* LayerNorm over the last dim with affine, fp32 statistics (nn.LayerNorm; reference model/sam2.py:1050-1051,
 * :474-476; on token-major maps this is also LayerNorm2d, :2334-2346).  x,y [rows, dim] bf16.
 * act: 0 none, 1 exact GELU applied to the normalised output (MaskDownSampler / output_upscaling, :611-643, :1976-1986). */
int rga3_layernorm_fwd(const void* x, const void* weight, const void* bias, void* y, int64_t rows, int64_t dim,
                       int64_t ldx, int64_t ldy, float eps, int act, void* stream);
/* LayerNorm folded into the consuming product (Hiera norm1 -> qkv, norm2 -> fc1, reference model/sam2.py:1035-1117): rga3_layernorm_stats writes
 * stats [rows][2] f32 = (mean, 1/sqrt(var + eps)) -- one read of x instead of LayerNorm's read + write -- and rga3_gemm_ln_bf16 computes
 * C = act(LayerNorm(A) W^T + b) as rinv_r (A Wf^T - mean_r c_n) + bias_n with Wf = W diag(gamma) (bf16), colc_n = sum_k Wf_nk (f32), bias = beta W^T + b (bf16).
 * act none / gelu / relu; tile -1 or 3 / 5 / 12 / 13 / 20 (see rga3_gemm_bf16); K % 8 == 0, N % 4 == 0. */
int rga3_layernorm_stats(const void* x, float* stats, int64_t rows, int64_t dim, int64_t ldx, float eps, void* stream);
int rga3_gemm_ln_bf16(const void* A, const void* Wf, const void* bias, const float* colc, const float* rowstat, void* C, int64_t M, int64_t N, int64_t K,
                      int64_t lda, int64_t ldw, int64_t ldc, int act, int tile, void* stream);
/* The LayerNorm statistics out of the PRODUCER's epilogue (round 6): the residual-writing products of a Hiera MultiScaleBlock -- x = shortcut + proj(attn), which
 * norm2 reads, and x = x + mlp(norm2(x)), which the next block's norm1 reads (reference model/sam2.py:1085-1117) -- leave, per output row r and TILE COLUMN t,
 *   row_parts[r][t] = (sum, sum of squares) of the bf16 values written to row r inside tile column t          (f32 pairs, [M][slices][2])
 * with slices = rga3_gemm_lnsum_slices(N, tile).  Plain stores, each element written exactly once (the waves of a workgroup combine through LDS in wave order): no
 * atomics, nothing to zero, bitwise reproducible.  rga3_gemm_lnsum_bf16 = rga3_gemm_bf16 (bf16 output, M > 16, no activation, tiles -1 / 3 / 5 / 12 / 13 / 20 / 23:
 * single-pass kernels instantiated for this epilogue, so no other product's kernel changes) with that output; rga3_gemm_lnq_bf16 = rga3_gemm_ln_bf16 reading such
 * partial sums (norm_width = the row width they cover, eps the LayerNorm's; the partials of a row are added, and mean and the biased variance S2 / w - mean^2
 * formed, in f64 inside the kernel).  Together they replace the stand-alone rga3_layernorm_stats pass over the rows. */
int64_t rga3_gemm_lnsum_slices(int64_t N, int tile);
int rga3_gemm_lnsum_bf16(const void* A, const void* W, const void* bias, const void* residual, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                         int64_t ldw, int64_t ldc, int64_t ldr, int tile, float* row_parts, void* stream);
int rga3_gemm_lnq_bf16(const void* A, const void* Wf, const void* bias, const float* colc, const float* row_parts, int64_t slices, int64_t norm_width, float eps,
                       void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int act, int tile, void* stream);

/* In-place rotary embedding on q and k heads living inside one [T, nheads_total, D] buffer (fused QKV output):
 * x = x*cos + rotate_half(x)*sin in fp32, cos/sin: [T, D] f32 tables (HF apply_rotary_pos_emb_vision
 * modeling_qwen2_5_vl.py:160-171 and apply_multimodal_rotary_pos_emb :557-599 after section interleave).
 * Heads [h0, h0+nh) are rotated; st/sh are the token/head element strides. */
int rga3_rope_inplace(void* x, const float* cos, const float* sin, int64_t T, int h0, int nh, int D, int64_t st,
                      int64_t sh, void* stream);

/* out[i, :] = table[idx[i], :] (embedding lookup / row gather; HF modeling_qwen2_5_vl.py:1206, window reorder
 * :434-438). 16-bit rows; `rows_per_idx` consecutive rows move together (merge-unit granularity). */
int rga3_gather_rows(const void* table, const int64_t* idx, void* out, int64_t n_idx, int64_t rows_per_idx,
                     int64_t dim, int64_t ld_table, int64_t ld_out, void* stream);
/* out[idx[i], :] = src[i, :] (masked_scatter of video embeds, HF :1217-1223; inverse window permutation :464-466) */
int rga3_scatter_rows(const void* src, const int64_t* idx, void* out, int64_t n_idx, int64_t rows_per_idx, int64_t dim,
                      int64_t ld_src, int64_t ld_out, void* stream);

/* copy [rows, cols] bf16 into a [rows, ld_out] buffer, zero-filling columns cols..ld_out-1 (K padding for GEMM) */
int rga3_pad_cols(const void* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, void* stream);

/* elementwise helpers on bf16: out = silu(a) * b ; out = a + b */
int rga3_silu_mul(const void* a, const void* b, void* out, int64_t n, void* stream);
int rga3_add(const void* a, const void* b, void* out, int64_t n, void* stream);

/* Shifted-label cross entropy over bf16/f32 logits rows, fp32 math, ignore_index = -100
 * (HF ForCausalLMLoss, modeling_qwen2_5_vl.py:1383-1393).  logits [rows, V] (ld in elements), labels [rows]
 * already shifted by the caller.  Writes per-row loss (0 for ignored rows) and, if dlogits != NULL,
 * (softmax - onehot) * grad_scale in bf16. */
int rga3_cross_entropy_rows(const void* logits, int logits_dtype, const int64_t* labels, float* row_loss,
                            void* dlogits, int64_t rows, int64_t V, int64_t ld, float grad_scale, void* stream);

/* ---- SAM2-side kernels (all feature maps token-major [pixels, C]) ------------------------------------------- */

/* im2col for Conv2d(ks, stride, pad) on NCHW bf16 images -> [F*Ho*Wo, ld_out] rows, columns (c, kh, kw), zero tail;
 * feeds the patch-embed GEMM (reference model/sam2.py:940-970: Conv2d 3->144 k7 s4 p3). */
int rga3_im2col(const void* img, void* out, int64_t F, int C, int H, int W, int ks, int stride, int pad, int64_t ld_out,
                void* stream);
/* MaxPool2d(2,2) on window-major tokens (windows of w x w -> w/2 x w/2): Hiera q-pooling, model/sam2.py:972-983,1015-1018 */
int rga3_maxpool2x2_win(const void* x, void* y, int64_t nwin, int w, int C, int64_t ldx, int64_t ldy, void* stream);
/* out[f,y,x,:] = a[f,y,x,:] + b[f,y/2,x/2,:]: FPN nearest x2 top-down path, model/sam2.py:867-889 */
int rga3_upsample2x_add(const void* a, const void* b, void* out, int64_t F, int H, int W, int C, void* stream);
/* out[r,:] = a[r,:] + alpha * b[r % rows_b,:] (positional-encoding adds: model/sam2.py:571-572, 1390-1409, 355) */
int rga3_add_bcast(const void* a, const void* b, void* out, int64_t rows, int64_t rows_b, int C, int64_t lda, int64_t ldb,
                   int64_t ldo, float alpha, void* stream);
/* F.interpolate(mode="bilinear", align_corners=False) on planes [N,Hi,Wi] -> f32 [N,Ho,Wo]; plane_idx (optional, int32 [N])
 * selects the source plane per output (argmax-IoU candidate): model/sam2.py:3388-3402, qwen_2_5_vl_sam2.py:248,272,387 */
int rga3_bilinear(const void* in, int in_dtype, float* out, const int32_t* plane_idx, int64_t N, int Hi, int Wi, int Ho, int Wo,
                  void* stream);
/* Conv2d(k3,s2,p1) token-major; x_dtype F32 = single fp32 plane transformed by sigmoid(x)*sig_scale+sig_bias on load
 * (model/sam2.py:3017-3022 + MaskDownSampler :611-643) */
int rga3_conv3x3s2(const void* x, int x_dtype, const void* w, const void* bias, void* y, int64_t F, int H, int W, int Cin, int Cout,
                   float sig_scale, float sig_bias, void* stream);
/* The image side of a two-way-transformer block boundary of the mask decoder in ONE launch (csrc/decimg.hip), for reference model/sam2.py:1926-2100
 * (TwoWayAttentionBlock.forward: cross_attn_image_to_token + norm4, then the next cross_attn_token_to_image's k / v projections, Attention :1417-1481):
 *   keys' [M, 256] = LayerNorm(bf16(bf16(attn((keys + pe) Wq^T + bq; kt, vt) Wo^T + bo) + keys)),   k2 = (keys' + pe) Wk2^T + bk2,  v2 = keys' Wv2^T + bv2  [M, 128]
 * with attn = 8 heads x 16 over the nk (<= 16) tokens of the row's frame (frame = row / hw, hw >= 16), kt / vt [frames * nk, 128] the token-side keys / values already
 * projected, pe [hw, 256] broadcast over frames.  wk2 = wv2 = NULL: only keys'.  v2_transposed != 0 (hw % 16 == 0): v2 is written as [frames * 128, hw] and k2 head-major
 * [8][M][16] for rga3_attn_fewq.  Strides in elements. */
int rga3_decimg_rows(const void* keys, int64_t keys_stride, const void* pe, int64_t pe_stride, int hw, const void* kt, const void* vt, int nk, const void* wq, const void* bq,
                     const void* wo, const void* bo, const void* ln_w, const void* ln_b, float eps, const void* wk2, const void* bk2, const void* wv2, const void* bv2,
                     void* keys_out, int64_t keys_out_stride, void* k2, void* v2, int64_t kv_stride, int v2_transposed, float scale, int64_t M, void* stream);
/* Attention of a few queries over many keys, one workgroup per (frame, head) and no merge launch (csrc/decimg.hip), for reference model/sam2.py:1417-1481
 * (Attention.forward as cross_attn_token_to_image / final_attn_token_to_image: 9 tokens x 4096 image keys, 8 heads x 16):
 * out [frames * nq, H * 16] bf16 = softmax(scale q k^T) v (+ vbias), nq <= 16, nk <= 4096 (multiple of 4); q row-major [*, H * 16]; key element (frame, j, h, d) at
 * ((frame * nk + j) * k_stride + h * k_head_stride + d): row-major (H * 16, 16) or head-major [H][frames * nk][16] (16, frames * nk * 16), strides in elements;
 * vt [frames * H * 16, nk] the projected values TRANSPOSED (rga3_decimg_rows(v2_transposed = 1) writes that layout; so does a product with the operands swapped),
 * vbias [H * 16] added after the normalisation (softmax rows sum to one). */
int rga3_attn_fewq(const void* q, int64_t q_stride, const void* k, int64_t k_stride, int64_t k_head_stride, const void* vt, const void* vbias, void* out, int64_t out_stride,
                   int frames, int nq, int nk, int H, float scale, void* stream);
/* n (<= 24) device-to-device copies in one launch (dst[i] <- src[i], bytes[i] bytes: multiples of 16, 16-byte aligned pointers, no overlap; the three arrays are HOST
 * arrays).  The copies a video-session frame makes around its captured graph (reference model/sam2.py:2829-2989 builds the bank with torch.cat / index ops per frame). */
int rga3_copy_many(void* const* dst, const void* const* src, const int64_t* bytes, int n, void* stream);
/* The same convolution with the LayerNorm2d over its output channels and the exact GELU that follow it in reference model/sam2.py:611-643 (MaskDownSampler stages)
 * in ONE launch, for the two narrow stages: (Cin, Cout) = (1, 4) from an f32 plane (with the sigmoid affine on load) or (4, 16) from bf16.  The arithmetic and
 * rounding points of rga3_conv3x3s2 followed by rga3_layernorm_fwd(act = 1). */
int rga3_conv3x3s2_ln_gelu(const void* x, int x_dtype, const void* w, const void* bias, const void* ln_w, const void* ln_b, float eps, void* y, int64_t F, int H, int W,
                           int Cin, int Cout, float sig_scale, float sig_bias, void* stream);
/* cols [F*(H/2)*(W/2), 9*C] (K index (kh*3+kw)*C + c, zero padded) of x [F,H,W,C] for Conv2d(k3,s2,p1) through rga3_gemm_bf16 with the weight
 * repacked [Cout, (kh,kw,ci)]: the wide stages of the memory encoder's mask downsampler (reference model/sam2.py:611-643) */
int rga3_im2col3x3s2(const void* x, void* cols, int64_t F, int H, int W, int C, void* stream);
/* depthwise Conv2d(k7,p3) token-major (CXBlock.dwconv, model/sam2.py:669-675) */
int rga3_dwconv7x7(const void* x, const void* w, const void* bias, void* y, int64_t F, int H, int W, int C, void* stream);
/* axial complex RoPE in place on [T, C] single-head tokens; rows t < n_rope use table row t % nq (model/sam2.py:1901-1923) */
int rga3_rope_axial_inplace(void* x, const float* cos, const float* sin, int64_t n_rope, int nq, int C, int64_t ldx, void* stream);
/* ConvTranspose2d(k2,s2) = GEMM to [pixels, 4*Co] + this shuffle (+bias, + optional high-res feature add, then
 * act: 0 none / 1 GELU), model/sam2.py:2137-2140 */
int rga3_pixel_shuffle2x(const void* g, const void* bias, const void* add, void* out, int64_t F, int H, int W, int Co, int act, void* stream);
/* per-mask {sum BCE-with-logits, sum sigmoid*t, sum sigmoid, sum t} (model/qwen_2_5_vl_sam2.py:17-60); out4 f32 [n_masks,4] */
int rga3_bce_dice_sums(const float* logits, const float* targets, float* out4, int64_t n_masks, int64_t hw, void* stream);
/* the same sums, reproducible run to run (per-block partial sums in `ws`, rga3_bce_dice_sums_ws_floats() f32 elements, added in block order; no atomics) */
int64_t rga3_bce_dice_sums_ws_floats(int64_t n_masks, int64_t hw);
int rga3_bce_dice_sums_det(const float* logits, const float* targets, float* out4, float* ws, int64_t ws_floats, int64_t n_masks, int64_t hw, void* stream);

/* ---- training step (backward + optimiser) ---------------------------------------------------------------- */

/* Flash-style attention backward (recompute from q,k,v,dO and the forward's lse): dq, dk, dv in bf16, no atomics.
 * strides16 = HOST array {q_st,q_sh,k_st,k_sh,v_st,v_sh,o_st,o_sh,do_st,do_sh,dq_st,dq_sh,dk_st,dk_sh,dv_st,dv_sh} (elements);
 * delta_ws: f32 workspace [Hq*total_q].  dkv_ws (optional, GQA only): f32 workspace [2 * Hq * total_k * D]; with it the dK/dV pass
 * runs one workgroup per (key block, QUERY head) and a fixed-order reduce adds the heads of a group (7x the parallelism for the
 * decoder's 28:4 heads); NULL keeps the single-pass form.  total_k = rows of k/v (= cu_k[nseg]).
 * Replaces the autograd of flash-attn / SDPA under train_joint.py:534. D <= 128. */
int rga3_attn_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                         void* dk, void* dv, float* delta_ws, const int32_t* cu_q, const int32_t* cu_k, int nseg, int max_q,
                         int max_k, int64_t total_q, int Hq, int Hkv, int D, const int64_t* strides16, float scale, int causal,
                         float* dkv_ws, int64_t total_k, void* stream);
/* Fused MLP of a Hiera stage-1 block, frozen trunk (csrc/hiera_mlp.hip): y [M, 144] = x + W2 gelu(LayerNorm(x) W1^T + b1) + b2 in one launch -- reference
 * model/sam2.py:1113-1117 (x = x + drop_path(mlp(norm2(x)))), MLP :2305-2329, dims 144 -> 576 -> 144, nn.GELU (erf).  w1f / c1 / d1 are the LayerNorm-folded operands
 * rga3_gemm_ln_bf16 takes (W1 diag(gamma) in bf16, its row sums in f32, beta W1^T + b1 in bf16); the hidden activation and the row statistics never reach HBM.
 * x, y contiguous bf16, x != y. */
int rga3_hiera_mlp144(const void* x, const void* w1f, const float* c1, const void* d1, const void* w2, const void* b2, void* y, int64_t M, float eps, void* stream);
/* The same block for stage 2 of the frozen Hiera-L trunk (blocks 2 - 7, C = 288, 16 384 tokens per 1024^2 frame; reference model/sam2.py:1035-1117 with dim_out = 288,
 * MLP 288 -> 1152 -> 288): y [M, 288] = x + W2 gelu(LayerNorm(x) W1^T + b1) + b2.  A workgroup is four waves, one per SIMD on the 512-register budget; the weights are
 * streamed by LDS-DMA from PACKED chunk images (rga3_hiera_mlp288_pack_bytes() bytes, built once per frozen block by rga3_hiera_mlp288_pack from the operands of the
 * C = 144 form; layout in csrc/hiera_mlp.hip), W1' and W2 through separate three-slot LDS rings, the first product of hidden chunk n + 1 interleaved with chunk n's
 * GELU.  x, y contiguous bf16, x != y. */
int64_t rga3_hiera_mlp288_pack_bytes(void);
int rga3_hiera_mlp288_pack(const void* w1f, const float* c1, const void* d1, const void* w2, void* pack, void* stream);
int rga3_hiera_mlp288(const void* x, const void* pack, const void* b2, void* y, int64_t M, float eps, void* stream);
/* Token-side tail of the SAM2 mask decoder at inference (csrc/dechead.hip).
 * rga3_mlp3_rows: n (<= 8) three-layer MLPs (Linear+ReLU, Linear+ReLU, Linear [+ sigmoid]) on ONE row per frame, B frames, one launch -- replaces the per-layer calls of
 * reference model/sam2.py:2142-2155 (output_hypernetworks_mlps, iou_prediction_head, pred_obj_score_head; MLP.forward :2319-2329).  HOST arrays:
 * ptrs = n x 8 device pointers {x, w0, b0, w1, b1, w2, b2, y} (bf16; weights [out, in] row-major, 16-byte aligned), dims = n x 6 {x frame stride, y frame stride
 * (elements), in, hidden, out, final_act (0 none, 1 sigmoid)}; in / hidden multiples of 8, all <= 512.
 * rga3_sam_select_objptr: reference model/sam2.py:3396-3421 -- best = argmax of IoU 1..3 (first maximum), sel = b * 4 + 1 + best, the chosen multimask token through
 * obj_ptr_proj, replaced by no_obj_ptr where the object score is <= 0.  iou [B, 4], obj [B, 1], toks = mask tokens [B, 4, C] with frame stride tok_bstride;
 * best = int64 [2, B] (row 0 the argmax, row 1 the plane index again as int64), sel = int32 [B]. */
int rga3_mlp3_rows(const void* const* ptrs, const int64_t* dims, int n, int64_t B, void* stream);
int rga3_sam_select_objptr(const void* iou, const void* obj, const void* toks, int64_t tok_bstride, int C, const void* w0, const void* b0, const void* w1, const void* b1,
                           const void* w2, const void* b2, const void* no_obj_ptr, int64_t* best, int32_t* sel, void* obj_ptr, int64_t B, void* stream);
/* SAM2 memory cross-attention with the values kept in memory space (csrc/memattn.hip): out [Nq, 64] bf16 = softmax(scale * q k^T) m for ONE 256-wide head,
 * q [Nq, 256], k [Nk, 256] (projected + rotated), m [Nk, 64] the un-projected memory rows; the caller applies the value projection to the 64-wide result
 * (softmax rows sum to one: softmax(S) (m Wv^T + bv) = (softmax(S) m) Wv^T + bv).  Replaces, for reference model/sam2.py:1519-1548 (RoPEAttention.forward of
 * cross_attn_image, kv_in_dim 64), v_proj over the whole bank + F.scaled_dot_product_attention.  Strides in elements; nsplit key slices (1..32);
 * ws = rga3_memattn_cross_ws_floats(Nq, nsplit) floats of caller workspace (< 0 = bad arguments).  Deterministic (no atomics).
 * out = NULL leaves only the per-slice partial results in ws ([nsplit, Nq, 64] f32 unnormalised sums, then [nsplit, Nq, 2] f32 (maximum in the log2 domain,
 * row sum)) for rga3_memlayer_rows to merge. */
int64_t rga3_memattn_cross_ws_floats(int64_t Nq, int nsplit);
int rga3_memattn_cross(const void* q, const void* k, const void* m, void* out, int64_t Nq, int64_t Nk, int64_t q_stride, int64_t k_stride, int64_t m_stride,
                       int64_t out_stride, float scale, int nsplit, float* ws, void* stream);
/* The row-wise chain between the attention kernels of a memory-attention layer in ONE launch (csrc/memlayer.hip; model width 256), for reference
 * model/sam2.py:448-530 (MemoryAttentionLayer._forward_sa / _forward_ca / forward: out_proj + residual, norm, the next q / qkv projection) and :1901-1923
 * (apply_rotary_enc).  Every stage but the LayerNorm is optional:
 *   product 1: operand rows a [M, K1] (K1 = 64 or 256) or the partial results of rga3_memattn_cross(out = NULL) (part_o, part_ml, nsplit; K1 = 64),
 *              x' = bf16(bf16(a w1^T + b1) + res) with w1 [256, K1]; written to x_out if given.  w1 = NULL: x' = res.
 *   t = LayerNorm(x'; ln_w, ln_b, eps); written to t_out if given.
 *   product 2: y = bf16(t w2^T + b2), w2 [N2, 256], N2 = 256 or 768; columns < rope_cols get the axial rotation (cos / sin [nq, 128] f32, row = token % nq,
 *              pair index = (column % 256) / 2); written to y_out.
 * Row strides in elements (multiples of 4; a: of 8). */
int rga3_memlayer_rows(const void* a, int64_t a_stride, int K1, const float* part_o, const float* part_ml, int nsplit, const void* w1, const void* b1,
                       const void* res, int64_t res_stride, void* x_out, int64_t x_stride, const void* ln_w, const void* ln_b, float eps, void* t_out,
                       int64_t t_stride, const void* w2, const void* b2, int N2, void* y_out, int64_t y_stride, const float* cos, const float* sin,
                       int rope_cols, int nq, int64_t M, void* stream);
/* n (<= 4) token-row products (M <= 16 rows each, as tile 41) in ONE launch, each optionally on the SUM of two row operands: the token side of reference
 * model/sam2.py:1926-2100 (TwoWayAttentionBlock.forward: q = queries + query_pe, then q_proj(q) / k_proj(q) / v_proj(queries) of the same attention are independent
 * nn.Linear calls).  ptrs: n x 6 {A, A2 or NULL, W, bias or NULL, residual or NULL, C}; dims: n x 9 {M, N, K, act (0 none / 1 gelu / 3 relu), lda, lda2, ldw, ldc, ldr}.
 * A + A2 is rounded to bf16 before the product.  HOST arrays. */
int rga3_gemm_rows16_many(const void* const* ptrs, const int64_t* dims, int n, void* stream);
/* diagnostic for tile 22 (stream-K): how many bounded waits on a partial-sum slab gave up in launches that used this workspace (expected 0;
 * < 0 = error); synchronises the device.  A give-up never yields a finite wrong product: the owner adds +inf to every sum of the tile it could not complete, so
 * that tile of C (and everything computed from it) is non-finite, and this counter says why. */
int rga3_gemm_stream_k_timeouts(const void* workspace);
/* byte offset of that 32-bit counter inside a workspace of the current device (< 0 = error): lets the caller fetch it with its own asynchronous copy, so a
 * training loop can watch it every few steps without a device synchronisation (no reference counterpart: the reference's GEMMs are vendor BLAS calls) */
int64_t rga3_gemm_timeout_counter_offset(void);
/* HOST ONLY (no device call): the work plan tiles 26 (split = 1) / 27 (split = 0) launch for an [M, N, K] product on `cus` compute units.  plan[8] = {first tile of the
 * stream-K tail, tiles of the tail, 1, workgroups that take ragged tiles only, 0, workgroups launched, workgroups with a run, tile rows per group}; start[cus + 1]:
 * workgroup w takes K-iterations [start[w], start[w + 1]) of the tail line (tile-major, K / 64 iterations per tile).  Returns 0; 1 when the last tile row is not
 * ragged (more than 64 rows) or no plan applies (the tiles then run as 22 / 21); < 0 = bad arguments.  No reference counterpart (the reference's GEMMs are vendor
 * BLAS calls); exists so that what the kernel's hand-off relies on is testable without a GPU. */
int rga3_gemm_ragged_plan(int64_t M, int64_t N, int64_t K, int cus, int split, int* plan, unsigned* start);
/* (the 32-bit word BEHIND that counter is a fault-injection switch for tests: 0 in every real run -- the caller zeroes the flag page; 0xffffffff makes every stream-K
 * owner behave as if its contributors never publish; any other value is the spin limit) */
/* dx = d rmsnorm(x; weight)/dx . dy (+ add): backward of HF Qwen2_5_VLRMSNorm (modeling_qwen2_5_vl.py:74-79) w.r.t. x */
int rga3_rmsnorm_bwd(const void* x, const void* weight, const void* dy, const void* add, void* dx, int64_t rows, int64_t dim, float eps,
                     void* stream);
/* y = (accumulate ? y : 0) + dropout(x): n bf16 elements (multiple of 8), keep probability 1 - round(p * 65536) / 65536, scale 1 / keep;
 * the mask is a counter hash of (seed, element index), so a recompute with the same seed reproduces it.  nn.Dropout(lora_dropout) on the
 * LoRA branch input (PEFT LoraLayer; reference train_joint.py:193-232) and its backward (accumulate = 1 into the input gradient). */
int rga3_dropout_bf16(const void* x, void* y, int64_t n, float p, int64_t seed, int accumulate, void* stream);
/* The pair form for a decoder layer's two LoRA branches (q_proj and v_proj drop the SAME normed input with two masks, and their input gradients meet in one
 * buffer; PEFT LoraLayer under reference train_joint.py:193-232): sum == 0: ya = dropout_a(xa), yb = dropout_b(xb) (xa may equal xb: read once);
 * sum == 1: ya += dropout_a(xa) + dropout_b(xb) with one bf16 rounding (yb unused).  Masks bit-identical to rga3_dropout_bf16's for the same (seed, index). */
int rga3_dropout_pair_bf16(const void* xa, const void* xb, void* ya, void* yb, int64_t n, float pa, int64_t seed_a, float pb, int64_t seed_b, int sum, void* stream);
/* a [T, I] = silu(gate) * up from the interleaved [T, 2I] pre-activations of RGA3_ACT_SWIGLU's weight packing (un-fused form, used after rga3_gemm_fp8) */
int rga3_swiglu_fwd(const void* gu, void* a, int64_t T, int64_t I, void* stream);
/* backward of silu(gate)*up on the interleaved [T, 2I] pre-activation layout of RGA3_ACT_SWIGLU: dgu from da [T, I] */
int rga3_swiglu_bwd(const void* gu, const void* da, void* dgu, int64_t T, int64_t I, void* stream);
/* out[c, r] = in[r, c] (16-bit): operand layout for dW = dY^T X through the NT GEMM */
int rga3_transpose16(const void* in, void* out, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, void* stream);
/* the same for n (<= 64) CONTIGUOUS 16-bit matrices in one launch (the W^T operands of the mask path's dX products: reference autograd through nn.Linear,
 * model/sam2.py:1417-1481, 2305-2329): HOST arrays ptrs = n x {in, out}, dims = n x {R, C}; out[i] is [C_i, R_i]. */
int rga3_transpose16_many(const void* const* ptrs, const int64_t* dims, int n, void* stream);
/* out[u,:] = sum_{j in [offsets[u], offsets[u+1])} x[rows[j],:]: embedding-table gradient rows (HF embed_tokens, trainable per
 * train_joint.py:242-243) without atomics */
int rga3_segment_sum_rows(const void* x, const int64_t* rows, const int64_t* offsets, void* out, int64_t n_out, int64_t dim, int64_t ldx,
                          void* stream);
/* fused AdamW (decoupled decay, bias correction) on bf16 params with fp32 master/moments; grad is scaled by grad_scale first
 * (loss scaling / clipping factor). Optimiser of train_joint.py:300-324 (DeepSpeed AdamW lr 4e-5, betas (0.9,0.95), wd 0). */
int rga3_adamw_step(void* param, float* master, const void* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, void* stream);
/* *out += sum(g^2) (fp32 atomic): global gradient norm for clipping (train_joint.py:300 gradient_clipping 1.0) */
int rga3_sumsq_accum(const void* g, float* out, int64_t n, void* stream);
/* out[0] = (accumulate ? out[0] : 0) + sum(g^2), DETERMINISTIC: per-block partial sums into the caller's `partials` (>= 1 floats, <= 2048 used), then one
 * block adds them in a fixed order -- data-parallel replicas must derive bit-identical clipping factors (same call site as rga3_sumsq_accum). */
int rga3_sumsq_det(const void* g, int64_t n, float* partials, int64_t partials_cap, float* out, int accumulate, void* stream);
/* rga3_adamw_step with 16-byte accesses and the clipping factor min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) derived ON THE DEVICE from the norm
 * rga3_sumsq_det left there (sumsq == NULL: no clipping): the optimizer step of train_joint.py:300-324, 534-535 without a device -> host sync.
 * All pointers 16-byte aligned. */
int rga3_adamw_step_clip(void* param, float* master, const void* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, const float* sumsq, float max_norm, void* stream);
/* the same update (weight decay 0) for a [rows, row_len] table, restricted to the rows with row_active[r] != 0: a row that never received a gradient has
   g = m = v = 0 and stays bit-for-bit unchanged under AdamW without weight decay, so skipping it is exact (embed_tokens: <= S of 152 064 rows per sample) */
int rga3_adamw_step_clip_rows(void* param, float* master, const void* grad, float* m, float* v, int64_t rows, int64_t row_len, const uint8_t* row_active,
                              float lr, float beta1, float beta2, float eps, int step, const float* sumsq, float max_norm, void* stream);
/* dst[idx[i], :] += scale * src[i, :] on bf16 rows, idx UNIQUE within the call: applies one rank's (row ids, rows) contribution to the
 * embed_tokens gradient (the sparse replacement of the 1.09 GB dense bucket of the reference's ZeRO-2 exchange, train_joint.py:325-334). */
int rga3_scatter_add_rows(void* dst, const int64_t* idx, const void* src, int64_t n, int64_t dim, int64_t ld_dst, int64_t ld_src, float scale,
                          void* stream);

/* ---- mask-path backward (trainable sam_mask_decoder + text_hidden_fcs, train_joint.py:237-251) ---------------- */

/* LayerNorm backward: dx (bf16) and, if non-NULL, the f32 parameter gradients; model/sam2.py:1364-1376,2334-2346.
   dim in {16,32,64,128,256,512}: dweight / dbias are WRITTEN, summed deterministically through `ws` (rga3_layernorm_bwd_ws_floats() f32 elements);
   other widths (multiples of 8 up to 2048): dweight / dbias += by f32 atomics into buffers the caller zeroes, ws unused. */
int64_t rga3_layernorm_bwd_ws_floats(int64_t rows, int64_t dim);
int rga3_layernorm_bwd(const void* x, const void* weight, const void* dy, void* dx, float* dweight, float* dbias, int64_t rows, int64_t dim,
                       float eps, float* ws, int64_t ws_floats, void* stream);
/* out[c] += sum_r x[r,c] (bias gradients; any width, f32 atomics) */
int rga3_colsum_accum(const void* x, float* out, int64_t rows, int64_t cols, int64_t ld, void* stream);
/* out[c] = sum_r x[r,c], written, deterministic two-stage sum; cols and ld multiples of 8, 16-byte aligned x; ws: rga3_colsum_ws_floats() f32 elements;
 * counters (optional: >= 128 32-bit words zeroed once by the caller and kept for the calls of one stream, left zero): both stages run in ONE launch -- the workgroup
 * that finishes a column block last adds its partial rows, in the same order as the second launch would (bit-identical) */
int64_t rga3_colsum_ws_floats(int64_t rows, int64_t cols);
int rga3_colsum(const void* x, float* out, int64_t rows, int64_t cols, int64_t ld, float* ws, int64_t ws_floats, void* counters, void* stream);
/* kind 0: out = gelu(a); kind 1: out = dy * gelu'(a) (a = pre-activation); kind 2: out = dy * (a > 0) (a = relu output) */
int rga3_act(const void* a, const void* dy, void* out, int64_t n, int kind, void* stream);
/* backward of rga3_bilinear (gather form): plane_idx == NULL -> every element of din is written; with plane_idx -> one f32 atomic add per input pixel into
   the pre-zeroed plane plane_idx[n] (deterministic for distinct indices) */
int rga3_bilinear_bwd(const float* dout, float* din, const int32_t* plane_idx, int64_t N, int Hi, int Wi, int Ho, int Wo, void* stream);
/* mask logits of all frames at once, model/sam2.py:2142-2149: masks[b,m,p] = sum_c hyper[b,m,c] up[b*P+p,c]; hyper [B,4,C] bf16, up [B*P,C] bf16 (C = 8/16/32),
   masks [B,4,P] f32.  Backward: dup [B*P,C] bf16, dhyper [B,4,C] bf16 from f32 dmasks; ws: rga3_mask_product_bwd_ws_floats() f32 elements */
int rga3_mask_product(const void* hyper, const void* up, float* masks, int64_t B, int64_t NM, int64_t P, int64_t C, void* stream);
int64_t rga3_mask_product_bwd_ws_floats(int64_t B, int64_t NM, int64_t P, int64_t C);
int rga3_mask_product_bwd(const float* dmasks, const void* hyper, const void* up, void* dup, void* dhyper, int64_t B, int64_t NM, int64_t P, int64_t C,
                          float* ws, int64_t ws_floats, void* stream);
/* backward of rga3_pixel_shuffle2x w.r.t. the GEMM output */
int rga3_pixel_shuffle2x_bwd(const void* dout, void* dg, int64_t F, int H, int W, int Co, void* stream);
/* dlogits = coef_bce * d(sum_n mean BCE) + coef_dice * d(sum_n dice_n) (model/qwen_2_5_vl_sam2.py:17-60) from the forward sums */
int rga3_bce_dice_grad(const float* logits, const float* targets, const float* sums4, float* dlogits, int64_t n_masks, int64_t hw,
                       float coef_bce, float coef_dice, void* stream);
/* the same with the two upstream gradients read from DEVICE memory (f32 scalars): autograd hands them over as device tensors, and reading them on the
 * host would be a device -> host sync in the middle of backward */
int rga3_bce_dice_grad_dev(const float* logits, const float* targets, const float* sums4, float* dlogits, int64_t n_masks, int64_t hw,
                           const float* coef_bce_dev, const float* coef_dice_dev, void* stream);

/* ---- SAM-side input pipeline: Pillow-exact antialiased bicubic resize + normalise + bf16, replacing
 * DirectResize.apply_image + preprocess + .bfloat16() (reference utils/utils.py:230-256, inference_mevis.py:175-180).
 * rga3_pil_bicubic_coeffs is HOST-ONLY: Pillow's precompute_coeffs / normalize_coeffs_8bpc tables (bounds [out*2], kk [out*ksize]);
 * bounds == NULL only queries *ksize_out. */
int rga3_pil_bicubic_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int64_t kk_capacity, int* ksize_out);
/* frames u8 [T,H,W,3] -> u8 [T,out_h,out_w,3] (dst_u8, optional) and/or bf16 [T,3,out_h,out_w] = (u8 - mean)/std (dst_bf16, optional).
 * bh/kh, bv/kv: device copies of the tables above; tmp: device workspace T*H*out_w*3 bytes; mean3/std3: host floats. */
int rga3_sam_preprocess_u8(const void* frames, int64_t T, int H, int W, int out_h, int out_w, const int32_t* bh, const int32_t* kh,
                           int ksize_h, const int32_t* bv, const int32_t* kv, int ksize_v, void* tmp, void* dst_u8, void* dst_bf16,
                           const float* mean3, const float* std3, void* stream);

/* ---- Qwen-side input pipeline: replaces qwen_vl_utils.process_vision_info (list-of-frames branch: PIL bicubic to the
 * smart_resize size; reference call sites evaluation/mevis_val_u/inference_mevis.py:196-216, utils/dataset.py:41-87) + the HF video processor
 * (rescale, CLIP-normalise, patchify; installed transformers models/qwen2_vl/video_processing_qwen2_vl.py:236-336).  The resize is
 * rga3_sam_preprocess_u8 with dst_u8; the tail is a byte gather through a 3x256 table.
 * rga3_qwen_norm_lut is HOST-ONLY: lut768[c*256+b] = normalised value of byte b in channel c; fused = 0: transformers 4.49 order
 * (float32(b/255.) then (x-mean)/std), fused = 1: 5.x fast path ((b - 255*mean) / (255*std)). */
int rga3_qwen_norm_lut(const float* mean3, const float* std3, int fused, float* lut768);
/* frames u8 [T,h,w,3] -> out [ceil(T/tpatch)*(h/patch)*(w/patch), 3*tpatch*patch*patch] bf16 (out_dtype 0) / fp32 (1); rows ordered
 * (t, h/(patch*merge), w/(patch*merge), merge, merge), columns (C, tpatch, patch, patch); the last frame repeats if tpatch does not divide T;
 * lut768: device copy of the table above. */
int rga3_qwen_patchify_u8(const void* frames, int64_t T, int h, int w, const float* lut768, void* out, int out_dtype, int patch, int tpatch,
                          int merge, void* stream);

/* ---- FP8 (OCP e4m3) path for the frozen-weight GEMMs of the LoRA fine-tune step (BASELINE.json configs[4]).
 * q[r, :] = e4m3(x[r, :] / scales[r]), scales[r] = max|x[r, :]| / 448 (1 for a zero row); x bf16 [rows, K], K % 8 == 0; ldx / ldq in elements / bytes */
int rga3_quant_fp8_rows(const void* x, void* q, float* scales, int64_t rows, int64_t K, int64_t ldx, int64_t ldq, void* stream);
/* C[M,N] (bf16) = (Aq[M,K] . Wq[N,K]^T) * sa[m] * sw[n] (+ bias[n]) (+ residual[m,n]) with v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 MFMA
 * rate); K % 128 == 0; lda / ldw in bytes.  Replaces the frozen nn.Linear contractions (HF modeling_qwen2_5_vl.py:602-757) when fp8 is enabled. */
int rga3_gemm_fp8(const void* Aq, const void* Wq, const float* sa, const float* sw, const void* bias, const void* residual, void* C, int64_t M,
                  int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, void* stream);
/* The row quantiser fused into the producers of the two widest activations of a decoder layer (BASELINE configs[4]): q / scales as rga3_quant_fp8_rows would
 * give them for rga3_swiglu_fwd(gu) [T, I] and for rga3_swiglu_bwd(gu, da) [T, 2I] (interleaved like gu), bit for bit, without the bf16 tensors ever
 * reaching memory.  The MLP of HF Qwen2MLP (modeling_qwen2_5_vl.py:541-553) under the fp8 switch. */
int rga3_swiglu_fwd_quant_fp8(const void* gu, void* q, float* scales, int64_t T, int64_t I, void* stream);
int rga3_swiglu_bwd_quant_fp8(const void* gu, const void* da, void* q, float* scales, int64_t T, int64_t I, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RGA3_HIP_H */
