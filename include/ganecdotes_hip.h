/* ganecdotes_hip.h — C ABI of libganecdotes_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the per-pixel SwAV clustering hot path of avm-debatr/ganecdotes
 * (`--method=hfc_with_swav`).  The reference's existing native boundary is two JIT-built
 * pybind11 torch extensions — `upfirdn2d(Tensor input, Tensor kernel, int up_x, up_y, down_x,
 * down_y, pad_x0, pad_x1, pad_y0, pad_y1)` (lib/gan/optim/upfirdn2d.cpp:18-39) and
 * `fused_bias_act(Tensor input, Tensor bias, Tensor refer, int act, int grad, float alpha,
 * float scale)` (lib/gan/optim/fused_bias_act.cpp:18-36) — plus chains of torch ops in
 * hfc_with_swav/swav_clustering.py and models/stylegan2/model.py.  This header replaces both
 * with plain-pointer entry points; INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all functions):
 *   - every pointer is a DEVICE pointer (HIP, one process per GPU) unless said otherwise;
 *     the CALLER allocates every output and workspace (the reference's callee-allocates
 *     convention, upfirdn2d_kernel.cu:252-253, is dropped so nothing allocates in the hot loop);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - return 0 (GX_OK) or a negative GX_ERR_* code; the Python wrapper raises RuntimeError,
 *     mirroring TORCH_CHECK -> RuntimeError of the reference boundary (upfirdn2d.cpp:9-15);
 *   - thread-safe for distinct streams; the only process-wide state is the contraction-engine mode
 *     (gx_set_gemm_mode / env GX_GEMM_MODE), read at launch time;
 *   - feature maps are NHWC ([b][y][x][c]) fp32; matrices row-major fp32; indices int32.
 */
#ifndef GANECDOTES_HIP_H
#define GANECDOTES_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GX_OK 0
#define GX_ERR_BAD_ARG (-1)
#define GX_ERR_ALIGN (-2)
#define GX_ERR_WORKSPACE (-3)
#define GX_ERR_LAUNCH (-4)

/* ---- one-shot fine-tune head and its scoring (SURVEY.md par. 8f rank 1) -------------------------
 * OneShotSegmentor (hfc_with_swav/swav_clustering.py:697-758) trained by OneShotPipeline.run_trainer
 * (src/one_shot_pipeline.py:533-581) and scored by run_tests (:595-886) + lib/metrics/segmentation.py.
 * Activations are zero-padded flat NHWC [B][H+2P][W+2P][C] (P = largest dilation): a dilated 3x3 tap is
 * a row shift, so conv forward / dW / dX are nine gx_gemm_f32 calls each on row-offset views; these
 * entry points are the glue: padding, bias-activation-mask and its backward, nn.CrossEntropyLoss + its
 * gradient (softmax_ce workspace: 1024 floats), torch.optim.Adam, argmax, confusion matrix [C][C]. */
int gx_head_pad(const float* in, int B, int H, int W, int C, int P, float* out, void* stream);
int gx_head_act_mask(float* y, int B, int H, int W, int C, int P, int act, void* stream);
/* a (dilated) 3x3 conv with few output channels (9 C <= 128) as one GEMM over the un-shifted padded rows with the nine taps
 * stacked along N (y9 [rows][ld9], tap t in columns t C .. t C + C - 1) and this tap sum:
 * y[r][c] = interior(r) ? act(bias[c] + sum_t y9[r + shift_t][t C + c]) : 0  (the conv + gx_head_act_mask in one pass) */
int gx_head_tap_sum(const float* y9, long ld9, int B, int H, int W, int C, int P, int dil, const float* bias, int act, float* y,
                    void* stream);
int gx_head_act_mask_bwd(float* dy, const float* y, int B, int H, int W, int C, int P, int act, void* stream);
int gx_head_softmax_ce(const float* logits, const long long* labels, int B, int H, int W, int C, int P, float* dlogits,
                       float* loss_out, float* workspace, size_t workspace_bytes, void* stream);
int gx_head_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 int step, void* stream);
int gx_head_argmax(const float* logits, int B, int H, int W, int C, int P, long long* pred, void* stream);
int gx_confusion(const long long* gt, const long long* pred, long n, int C, unsigned long long* conf, void* stream);

const char* gx_version(void);

/* ---- dense fp32-MFMA GEMM -------------------------------------------------------------------
 * C = alpha * row_scale[m] * opA opB + col_bias[n] + beta * C.   transA=0: A is [M][K] (lda),
 * 1: [K][M].  transB=0: B is [K][N] (ldb), 1: [N][K].  split_k>1 needs a workspace of
 * gx_gemm_f32_workspace_bytes() and sums the K-slices in a fixed order.
 * Replaces: F.linear / autograd matmuls of swav_clustering.py:171-175, :459. */
/* Contraction engine behind gx_gemm_f32 / gx_gemm_f32_cat (and the modulated conv):
 *   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32; bitwise an fmaf chain, 157 TFLOP/s peak)
 *   1  "bf16x6": every fp32 operand element split into three bf16 (exactly), six bf16 MFMA products
 *      per fp32 product, fp32 accumulate — fp32-level error at 1/6 of the 2.5 PFLOP/s bf16 peak.
 * gx_set_gemm_mode returns the previous mode; env GX_GEMM_MODE=0|1 sets the initial one. */
int gx_set_gemm_mode(int mode);
int gx_get_gemm_mode(void);
/* lab / tests: 0 keeps the 128-row tile of the f16x3 weight-plane GEMM (default 1: 256-row tiles on the large launches) */
int gx_set_gemm_tall(int on);
/* lab: n > 0 overrides how many N tiles of the weight-plane GEMM share a sweep over the M tiles (0: chosen from the planes' size) */
int gx_set_gemm_ngroup(int n);
/* Plane format of the x-planes conv path under engine 1 (read at launch time by the producers of planes — gx_split_conv_weights,
 * gx_modsplit, the conv / blur passes that write the next conv's planes — and by the convs that read them; planes made under
 * one format must be consumed under the same): 0 = three bf16 planes, six products per fp32 product ("bf16x6"); 1 = two fp16
 * planes of the element scaled by a power of two, three products ("f16x3": fp32-class error measured against fp64, half the
 * matrix-core work; needs |x * style| < 32752 and |w| < 4094, an element out of range comes out NaN).  Buffer sizes
 * (gx_xplanes_bytes, gx_conv_weight_planes_bytes) are the three-plane ones under either format.  Returns the previous
 * format; env GX_CONV_PLANES=0|1 sets the initial one (default 1). */
int gx_set_conv_planes(int fmt);
int gx_get_conv_planes(void);
/* lab / tests: 0 keeps the 128 x 128 tile of the f16x3 convs on every layer (default 1: 256 x 128 on the large ones) */
int gx_set_conv_tall(int on);
size_t gx_gemm_f32_workspace_bytes(int M, int N, int split_k);
int gx_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                int split_k, float* workspace, size_t workspace_bytes, void* stream);

/* C = alpha * [A1 | A2] opB + col_bias + beta*C with A1 [M][K1] and A2 [M][K2] side by side along K
 * (K1 % 32 == 0): two same-resolution taps against adjacent column blocks of the projection. */
int gx_gemm_f32_cat(int transB, int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                    const float* B, long ldb, float* C, long ldc, float alpha, float beta, const float* col_bias,
                    void* stream);
/* One level of the pyramid projection in one launch: C = [A1 | A2] opB + beta*C + upsample2(coarse), where C's M rows
 * are the pixels (b, y, x) of fine_res x fine_res images (a power of two) and coarse [M/4][N] is the finished table
 * of the level below, row (b, y/2, x/2).  K2 = 0: a single A.  (F.interpolate(nearest) + concat + F.linear of
 * swav_clustering.py:108-130, :171, factorised; replaces a separate gx_upsample2_add pass over C.) */
int gx_gemm_f32_upadd(int transB, int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                      const float* B, long ldb, float* C, long ldc, float beta, const float* coarse, int fine_res,
                      void* stream);

/* C = alpha * A^T [B1 | B2] + beta * C with A [K][M], B1 [K][N1], B2 [K][N2] (N1 a multiple of 128): the projection-weight
 * gradient of one pyramid level against both of its taps in one pass over dZ (autograd of swav_clustering.py:171). */
int gx_gemm_f32_tn_cat(int M, int N1, int N2, int K, const float* A, long lda, const float* B1, long ldb1, const float* B2,
                       long ldb2, float* C, long ldc, float alpha, float beta, int split_k, float* workspace,
                       size_t workspace_bytes, void* stream);

/* The NT contraction with the weight operand pre-split into bf16x6 planes once per step (the projection / prototype
 * weights change once per optimizer step): C = alpha * row_scale .* ([A1 | A2] B^T) + col_bias + beta * C
 * + upsample2(coarse).  b_planes = gx_split_gemm_weights(B [N][K1 + K2], ldb) (gx_gemm_weight_planes_bytes(N, K1 + K2)
 * bytes); K2 = 0, coarse / row_scale / col_bias = NULL switch the optional parts off.  bf16x6 engine only (GX_ERR_BAD_ARG
 * otherwise); K1, K2 multiples of 16, 16-B aligned rows.  Bitwise the result of gx_gemm_f32 / _cat / _upadd on that
 * engine.  Same reference ops as gx_gemm_f32 (swav_clustering.py:171-175). */
size_t gx_gemm_weight_planes_bytes(int N, int K);
int gx_split_gemm_weights(const float* B, int N, int K, long ldb, unsigned short* planes, void* stream);
/* ... as the two fp16 planes of the f16x3 form (same buffer size): B times the power of two that brings max |B| (found on the
 * device) to [2^13, 2^14); max |B| is left in the buffer's 16-byte trailer for gx_gemm_f32_nt_planes(..., a_range > 0). */
int gx_split_gemm_weights_h2(const float* B, int N, int K, long ldb, unsigned short* planes, void* stream);
int gx_gemm_f32_nt_planes(int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                          const unsigned short* b_planes, float* C, long ldc, float alpha, float beta,
                          const float* row_scale, const float* col_bias, const float* coarse, int fine_res,
                          float* colmax_part, float* usum_part, float sk_eps, float a_range, const float* a_amax_dev,
                          void* stream);
/* a_range: 0 = b_planes from gx_split_gemm_weights (three bf16 planes, six products per fp32 product); > 0 = an upper bound of
 * |A1|, |A2| and b_planes from gx_split_gemm_weights_h2 (two fp16 planes, three products — csrc/gx_mfma_x6.h "f16x3": fp32-class
 * error, half the matrix-core work).  The A elements are scaled by the power of two that brings a_range to [2^13, 2^14): an
 * element above 4 x a_range overflows fp16 and comes out NaN; every element carries an absolute error of a_range x 2^-38
 * (the fp16 subnormal step), so the bound should sit within ~2^14 of A's RMS for fp32-class results.  a_amax_dev (optional, with
 * a_range > 0): a device float holding max |A| (or a bound of it) as A's producer left it — the bound is then *a_amax_dev x
 * a_range, read by the kernel (gx_absmax, gx_fineq_bwd's tt_amax, gx_gemm_f32_nt_planes_finish's g_amax). */
/* gx_absmax: *out = max |x| over an [rows][cols] matrix of row stride ld (sqrt_of != 0: max sqrt |x| — a column of squared row
 * norms gives a bound of the rows' elements); 0 for an all-zero matrix */
int gx_absmax(const float* x, long rows, int cols, long ld, int sqrt_of, float* out, void* stream);
/* colmax_part (optional, beta == 0): gx_gemm_colmax_part_bytes(M, N) bytes = [4 ceil(M / 128)][N] partial column maxima of C, one row per 32 rows
 * of C (-inf where a slice is empty); their maximum over rows is gx_sinkhorn_colmax's result (set it with
 * gx_sinkhorn_cmax_set) without a pass of its own over the score matrix (swav_clustering.py:509-544) */
size_t gx_gemm_colmax_part_bytes(int M, int N);
/* ... with the long dimension's data rows counted on the device (the drawn pixels of the many-prototype tables come first,
 * DESIGN.md par. 4.15): gx_gemm_f32_nt_planes_mlive skips the 128-row tiles at or past *m_live (their rows of C keep what they
 * held), gx_gemm_f32_klive (TN only) contracts over the first *k_live rows only, shared among its split_k splits */
int gx_gemm_f32_nt_planes_mlive(int M, int N, int K1, const float* A1, long lda1, const unsigned short* b_planes, float* C,
                                long ldc, float alpha, float beta, const float* col_bias, const int* m_live, float a_range,
                                void* stream);
int gx_gemm_f32_klive(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                      float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                      int split_k, float* workspace, size_t workspace_bytes, const int* k_live, void* stream);
/* gx_gemm_f32 / gx_gemm_f32_klive (k_live may be NULL) with the caller's bounds on |A| and |B|: where the wave-specialised
 * kernel takes the launch (long k loops: the gradient GEMMs over pixels) it runs the f16x3 form (operands scaled by the powers
 * of two that bring the bounds to [2^13, 2^14), two fp16 planes each, three MFMA products per fp32 product — half the
 * matrix-core work of bf16x6 at fp32-class error); elsewhere the bounds are ignored.  An element above 4 x its bound comes out
 * NaN; every element carries an absolute error of bound x 2^-38: a bound should sit within ~2^14 of its operand's RMS. */
int gx_gemm_f32_ranged(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                       float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                       int split_k, float* workspace, size_t workspace_bytes, const int* k_live, float a_range, float b_range,
                       const float* a_amax_dev, const float* b_amax_dev, void* stream);
/* a_amax_dev (optional; b_amax_dev / b_range likewise): A's bound lives on the device — max |A| as left behind by A's producer (g_amax of
 * gx_gemm_f32_nt_planes_finish) — and a_range is then the host factor it is multiplied by (4 per 2x2 sum-pooling since).
 * gx_gemm_f32_tn_cat_ranged: gx_gemm_f32_tn_cat the same way (|B1|, |B2| <= b_range). */
int gx_gemm_f32_tn_cat_ranged(int M, int N1, int N2, int K, const float* A, long lda, const float* B1, long ldb1, const float* B2,
                              long ldb2, float* C, long ldc, float alpha, float beta, int split_k, float* workspace,
                              size_t workspace_bytes, const float* a_amax_dev, float a_mul, float b_range, void* stream);
/* gx_gemm_f32_nt_planes_finish: G = A1 B^T with the low-rank path's dense finish in the epilogue (autograd of
 * swav_clustering.py:171-175 through the pyramid, K <= 16): G[q] += coeffs[q][:K] Wc + coeffs[q][K] Z[q], gWc = beta gWc +
 * coeffs[:, :K]^T Z, Gpool (may be NULL) = the 2x2 sums of G; rows = pixels of res x res images (DESIGN.md par. 4.14) */
int gx_gemm_finish_ok(int N, int K1, int res, int K);
size_t gx_gemm_finish_workspace_bytes(long M, int N, int K);
int gx_gemm_f32_nt_planes_finish(long M, int N, int K1, const float* A1, long lda1, const unsigned short* b_planes, float* G,
                                 int res, const float* coeffs, const float* Z, const float* Wc, int K, float* Gpool,
                                 float* gWc, float beta, float* workspace, size_t workspace_bytes, float* g_amax, float a_range,
                                 const float* a_amax_dev, void* stream);
/* g_amax (optional; one float the caller has zeroed): max |G| over the launch, for gx_gemm_f32_ranged's a_amax_dev;
 * a_range / a_amax_dev: as in gx_gemm_f32_nt_planes (a_range > 0: b_planes from gx_split_gemm_weights_h2, the f16x3 form) */
/* usum_part (optional, with colmax_part; same size): per 32-row slice and column the sum of exp((C - that slice's maximum) /
 * sk_eps) — the FIRST Sinkhorn pass (swav_clustering.py:519-520: Q = exp(scores / eps), its row sums) left behind by the GEMM
 * that writes the scores.  gx_sinkhorn_first_from_parts merges the slices of `groups` equal row groups by log-sum-exp:
 * cmax [groups][N] = the column maxima, u [groups][N] = sum over the group's rows of exp((C - cmax) / eps). */
size_t gx_sinkhorn_first_from_parts_workspace_bytes(int M, int N);
int gx_sinkhorn_first_from_parts(const float* colmax_part, const float* usum_part, int M, int N, int groups, float eps,
                                 float* cmax, float* u, float* workspace, size_t workspace_bytes, void* stream);
/* the first pass of a Sinkhorn problem whose producer left (cmax_local, u_local) behind: u (workspace + K) = u_local *
 * exp((cmax_local - cmax) / eps) with cmax = the maxima now in the workspace (after a cross-rank MAX all-reduce they may
 * exceed the local ones) — replaces gx_sinkhorn_pass(first = 1) */
int gx_sinkhorn_first_preset(const float* u_local, const float* cmax_local, int K, float eps, float* workspace, void* stream);

/* ---- the finest pyramid level of the low-rank (K <= 16) path without its projected table (csrc/gx_fineq.hip) ----
 * Reference: swav_clustering.py:120-128 (nearest upsample + concat), :171-175 (projection, F.normalize, prototypes) and
 * autograd of them.  A finest-level pixel q with taps t(q) (CF = K1 + K2 channels) and parent p has z(q) = Wf t(q) + Zc[p];
 * the loss needs only <Wc_k, z> = (Wc Wf)_k . t + Ac[p][k] and |z|^2 = t^T (Wf^T Wf) t + t . Y2[p] + nc[p]
 * (Ac = Zc Wc^T, Y2 = 2 Zc Wf, nc = |Zc|^2: gx_proto_raw_tables and one gx_gemm_f32_nt_planes on the level below, both
 * writing into ONE table of gx_fineq_ytable_ld() floats per parent pixel).
 *   gx_fineq_ok            shapes this path takes: CF = 256 as one tap or as two taps of 128, K <= 16
 *   gx_fineq_pack_weights  Bq [CF + 16][CF] = [tri(Mg) ; Wcf ; 0] from Mg = Wf^T Wf [CF][CF] and Wcf = Wc Wf [K][CF]
 *                          (tri: 2 x strict lower + diagonal — t^T tri(M) t = t^T M t, k-tile kt only meets the column
 *                          blocks >= kt / 2 and a block is final two k-tiles after its first)
 *   gx_fineq_fwd           rn_table [rows], sc_table [rows][K] (= gx_upsample2_add_scores' tables) from the taps, the planes
 *                          gx_split_gemm_weights(Bq) and the three coarse tables; rows = B * fine_res^2 pixels in (b, y, x)
 *                          order, fine_res a power of two >= 32
 *   gx_fineq_bwd           from the coefficient grid C [rows][gx_lowrank_stride(K)] (gx_proto_bwd_lowrank*), ONE pass over the
 *                          taps: S [256][256] = sum_q C[q][K] t(q) t(q)^T (the weighted Gram matrix; only the blocks on /
 *                          above the diagonal are contracted, the rest mirrored), P [K][256] = C[:, :K]^T t, Tt [rows/4][256]
 *                          = sum over the 2x2 children of C[q][K] t(q), Cpool [rows/4][stride] = the children's coefficient
 *                          sums.  workspace: gx_fineq_bwd_workspace_bytes(); fine_res a power of two >= 16 */
int gx_fineq_ok(int K1, int K2, int K);
int gx_fineq_pack_weights(const float* Mg, const float* Wcf, int CF, int K, float* Bq, void* stream);
int gx_fineq_ytable_ld(void);   /* floats per parent pixel of the coarse table: [0, 256) Y2 | [256, 272) Ac | 272 nc | pad */
int gx_fineq_fwd(int rows, int fine_res, int K1, const float* t1, long ld1, int K2, const float* t2, long ld2,
                 const unsigned short* bq_planes, const float* ytable, const float* bias, int K, float* rn_table,
                 float* sc_table, float a_range, float* t_amax, void* stream);
/* a_range: 0 = bq_planes from gx_split_gemm_weights (bf16x6); > 0 = a bound of |t1|, |t2| and bq_planes from
 * gx_split_gemm_weights_h2 (the f16x3 form, as in gx_gemm_f32_nt_planes); t_amax (optional device float): max |t1|, |t2| as the
 * pass met them — the tight bound gx_fineq_bwd's f16x3 form takes */
/* gx_lowrank_coeff_amax: out2[0] = max |C[q][k]| (k < K), out2[1] = max |C[q][K]| over the coefficient grid C [rows][gx_lowrank_stride(K)] */
int gx_lowrank_coeff_amax(const float* C, long rows, int K, float* out2, void* stream);
size_t gx_fineq_bwd_workspace_bytes(void);
int gx_fineq_bwd(const float* C, int rows, int fine_res, int K1, const float* t1, long ld1, int K2, const float* t2, long ld2,
                 int K, float* S, float* P, float* Tt, float* Cpool, float* workspace, size_t workspace_bytes, float* tt_amax,
                 const float* t_amax, const float* c_amax, void* stream);
/* tt_amax (optional device float): max |Tt| over the launch — the a_amax_dev / b_amax_dev of the f16x3 GEMMs that contract Tt.
 * t_amax + c_amax (both or neither; device floats from gx_fineq_fwd and gx_lowrank_coeff_amax): the pass runs its f16x3 form
 * (images scaled by the powers of two of max |t|, max |C[:, K]| max |t| and max |C[:, :K]|; DESIGN.md par. 4.21) */
/* dots[row * ld_dots + k] = (Z Wc^T)[row][k], sumsq[row * ld_sumsq] = |Z[row]|^2 of a finished table Z [rows][D] (no
 * normalisation, no bias); the strides let both land inside the coarse table of gx_fineq_fwd */
int gx_proto_raw_tables(const float* Z, long rows, int D, const float* Wc, int K, float* sumsq, long ld_sumsq, float* dots,
                        long ld_dots, void* stream);

/* ---- generator forward (models/stylegan2/model.py) ------------------------------------------
 * gx_pixel_norm      PixelNorm                                   model.py:105-110
 * gx_bias_act        EqualLinear epilogue / fused_leaky_relu     model.py:32-43, :242-252;
 *                    fused_bias_act_kernel.cu:18-85 (act=lrelu 0.2, scale sqrt2, grad=0)
 * gx_lerp_rows       truncation  m + psi (w - m)                 model.py:594-602
 * gx_demod           demodulation factors                        model.py:333-335
 * gx_modconv_fwd     ModulatedConv2d (k=3 pad 1, or k=1) + NoiseInjection + FusedLeakyReLU
 *                                                                model.py:362-366, :377-382, :426-432
 * gx_modconv_up_fwd  upsample variant: conv_transpose2d stride 2 + Blur(pad (1,1))
 *                                                                model.py:341-352, :166-182;
 *                    upfirdn2d_kernel.cu:114-215 mode 1 (up1 down1 4x4)
 * gx_torgb_fwd       ToRGB incl. Upsample of the skip            model.py:445-454, :124-142;
 *                    upfirdn2d_kernel.cu mode 3 (up2 down1 4x4, pad (2,1))
 * gx_nchw_to_nhwc    layout change at the API boundary */
int gx_pixel_norm(const float* x, long rows, int D, float* y, void* stream);
int gx_bias_act(float* y, long rows, int N, float alpha, const float* bias, float bscale, int act, void* stream);
/* gx_equal_linear_skinny: EqualLinear (+ fused_leaky_relu) on a few rows in one launch — the mapping network's layers on a
 * batch of latents (model.py:242-252, :420-435); wt = the layer's weight transposed, (K, N) */
int gx_equal_linear_skinny(const float* x, int rows, int K, const float* wt, int N, float alpha, const float* bias,
                           float bscale, int act, float* out, void* stream);
int gx_lerp_rows(const float* w, const float* base, long n, int D, float psi, float* out, void* stream);
/* the (B, n_latent, S) latents of one view: truncation (model.py:594-602) + the perturbation mix of
 * image_augmentor.py:42-53 / swav_clustering.py:599-640 in one pass */
int gx_view_latents(const float* w, const float* mean, float psi, const float* nw, const float* mix, int B, int n_latent,
                    int S, float* lat, void* stream);
/* One generator forward's per-layer work in single launches (the reference runs one EqualLinear per
 * ModulatedConv2d, model.py:239-243): gx_style_pick gathers every layer's style from ONE affine GEMM
 * of all latents x all layers; gx_demod_batch is gx_demod for all styled convs;
 * table layouts: see csrc/gx_conv.hip. */
int gx_style_pick(const float* S_all, int B, int n_latent, int total, const int* table, int nlayer, float* out,
                  void* stream);
int gx_demod_batch(const float* s_packed, const float* wsq_all, int B, const int* table, const float* scale, int nconv,
                   int total_rows, float* demod, void* stream);
/* gx_style_norm (after gx_demod_batch, same table): per (styled conv, sample) s /= c and demod *= c in place, c the power of
 * two that brings the sample's largest |s| into [2, 4).  Exact — demod * conv(x * s) is unchanged (ModulatedConv2d,
 * model.py:239-262, is linear in the style) — and the f16x3 planes' fixed range (DESIGN.md par. 4.21) then bounds the
 * activations alone, whatever magnitudes a trained checkpoint's style affines produce. */
int gx_style_norm(float* s_packed, float* demod, int B, const int* table, int nconv, void* stream);
int gx_demod(const float* s, const float* wsq, int B, int Cin, int Cout, float scale, float* demod, void* stream);
size_t gx_modconv_slab_bytes(int B, int outH, int outW, int Cout);   /* optional split-K scratch (small layers) */
int gx_modconv_fwd(const float* x, int B, int H, int W, int Cin, const float* wt, int ksize, int Cout, const float* s,
                   const float* demod, const float* noise, long noise_bstride, const float* noise_w,
                   const float* bias, int act, float* out, float* slab, size_t slab_bytes, void* stream);
size_t gx_modconv_up_scratch_bytes(int B, int H, int W, int Cout);
int gx_modconv_up_fwd(const float* x, int B, int H, int W, int Cin, const float* wt, int Cout, const float* s,
                      const float* demod, const float* kf, const float* noise, long noise_bstride,
                      const float* noise_w, const float* bias, int act, float* T, float* out, float* slab,
                      size_t slab_bytes, void* stream);
int gx_modconv_up_convt(const float* x, int B, int H, int W, int Cin, const float* wt, int Cout, const float* s,
                        float* T, float* slab, size_t slab_bytes, void* stream);
/* bf16x6 engine: conv weights pre-split once per (frozen) model into three bf16 planes, tiled in the order
 * the kernel copies them into LDS (gx_split_conv_weights; gx_conv_weight_planes_bytes bytes); the _p
 * forms take them (NULL = split the fp32 weight tile in every workgroup). */
size_t gx_conv_weight_planes_bytes(int taps, int Cout, int Cin);
int gx_split_conv_weights(const float* wt, int taps, int Cout, int Cin, unsigned short* planes, void* stream);
int gx_modconv_fwd_p(const float* x, int B, int H, int W, int Cin, const float* wt, const unsigned short* wt_planes,
                     int ksize, int Cout, const float* s, const float* demod, const float* noise, long noise_bstride,
                     const float* noise_w, const float* bias, int act, float* out, float* slab, size_t slab_bytes,
                     void* stream);
int gx_modconv_up_convt_p(const float* x, int B, int H, int W, int Cin, const float* wt, const unsigned short* wt_planes,
                          int Cout, const float* s, float* T, float* slab, size_t slab_bytes, void* stream);   /* stage 1: transposed conv -> T */
/* bf16x6 engine, the ACTIVATION side: the modulated input x * s split ONCE into three bf16 planes
 * (chunk-major [ceil(Cin/16)][3][B][H+2][W+2][16], zero border: the caller zeroes the gx_xplanes_bytes() buffer once,
 * kernels write the interior only) instead of once per tap and Cout tile inside the conv; the _xp convs stream both
 * operands HBM -> LDS by LDS-DMA.  Needs wt_planes and >= 256 output tiles (no split-K).  Same arithmetic as the _p forms. */
size_t gx_xplanes_bytes(int B, int H, int W, int Cin);
int gx_modsplit(const float* x, int B, int H, int W, int Cin, const float* s, unsigned short* planes, void* stream);
int gx_modconv_fwd_xp(const unsigned short* x_planes, int B, int H, int W, int Cin, const float* wt,
                      const unsigned short* wt_planes, int ksize, int Cout, const float* demod, const float* noise,
                      long noise_bstride, const float* noise_w, const float* bias, int act, float* out,
                      const float* s_next /* [B][Cout] or NULL */, unsigned short* planes_out /* or NULL */, void* stream);
/* s_next / planes_out: the producer also writes the x planes of out * s_next, the NEXT conv's modulated input
 * (gx_xplanes_bytes(B, out H, out W, Cout) bytes, border zeroed once by the caller): the activation is split once,
 * where it is made.  gx_blur_act_fwd_xp = gx_blur_act_fwd with the same two arguments. */
int gx_blur_act_fwd_xp(const float* T, int B, int TH, int TW, int C, const float* kf, const float* demod,
                       const float* noise, long noise_bstride, const float* noise_w, const float* bias, int act,
                       float* out, const float* s_next, unsigned short* planes_out, void* stream);
int gx_modconv_up_convt_xp(const unsigned short* x_planes, int B, int H, int W, int Cin, const float* wt,
                           const unsigned short* wt_planes, int Cout, float* T, void* stream);
/* the large blur launches take their T rows by LDS-DMA several rows ahead (1) or through registers (0: the round-2 form; A/B and
 * tests); returns the previous setting */
int gx_set_blur_dma(int on);
int gx_blur_act_fwd(const float* T, int B, int TH, int TW, int C, const float* kf, const float* demod,
                    const float* noise, long noise_bstride, const float* noise_w, const float* bias, int act,
                    float* out, void* stream);                         /* stage 2: blur + epilogue */
int gx_torgb_fwd(const float* x, int B, int H, int W, int Cin, const float* w3, const float* s, float scale,
                 const float* bias3, const float* prev_img, const float* kf, float* img, void* stream);
int gx_nchw_to_nhwc(const float* in, int B, int C, int HW, float* out, void* stream);

/* ---- the reference's two native ops, with THEIR argument lists --------------------------------------
 * gx_upfirdn2d       = upfirdn2d(input [major][in_h][in_w][minor], kernel [kh][kw], up_x, up_y, down_x, down_y,
 *                      pad_x0, pad_x1, pad_y0, pad_y1) -> out [major][out_h][out_w][minor]
 *                      (lib/gan/optim/upfirdn2d.cpp:18-35; arithmetic of upfirdn2d_kernel.cu:53-215 for any
 *                      up / down / pad, negative pads = crop: zero-insert upsample, pad, correlate with the
 *                      FLIPPED kernel, decimate).  out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y
 *                      (same for w): gx_upfirdn2d_out_shape.  The caller allocates `out` (the reference
 *                      allocates it, upfirdn2d_kernel.cu:252-253).
 * gx_fused_bias_act  = fused_bias_act(input, bias, refer, act, grad, alpha, scale) -> out
 *                      (lib/gan/optim/fused_bias_act.cpp:18-31, kernel :18-85): x += bias[(i / step_b) % size_b]
 *                      when bias != NULL; act 1 linear / 3 leaky-ReLU(alpha); grad 0 forward, 1 gradient
 *                      (gated by the sign of `refer`), 2 second gradient (= 0); out = y * scale.
 *                      step_b = product of input dims after dim 1 (the reference derives it from the tensor).
 * The hot path uses the FUSED forms above (gx_blur_act_fwd, gx_torgb_fwd, gx_modconv_*: no standalone
 * pass over a feature map); these two exist so that the reference's own Python wrappers
 * (lib/gan/optim/upfirdn2d.py:86-143, fused_act.py:110-168) can bind to this library unchanged. */
int gx_upfirdn2d_out_shape(int in_h, int in_w, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1, int* out_h, int* out_w /* host */);
int gx_upfirdn2d(const float* input, int major, int in_h, int in_w, int minor, const float* kernel, int kernel_h,
                 int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                 int pad_y1, float* out, void* stream);
int gx_fused_bias_act(const float* input, long n, const float* bias, int size_b, long step_b, const float* refer,
                      int act, int grad, float alpha, float scale, float* out, void* stream);

/* ---- multi-scale upsample + concat + rotate/flip + pixel sampling ---------------------------
 * gx_rotflip_map   source-pixel map of RandomRotation(10)+RandomHorizontalFlip
 *                                                  swav_clustering.py:98-102, :358-359 (torchvision, restated)
 * gx_gather_taps   X[row] = first hlen channels of concat_l nearest-upsampled tap_l at the
 *                  sampled (and rotated/flipped) pixel; row = b*rows_per_img + i, pixel =
 *                  picks[row], optional maps[b][H*W] (-1 = zero fill)
 *                                                  swav_clustering.py:108-130, :160-167
 * gx_row_sumsq / gx_norm_map   per-pixel L2 norm map of the stack   swav_clustering.py:361-362 */
/* the N pixels of a patch: a uniform random N-subset of the HW pixels per row (what randperm(HW)[:N]
 * selects, swav_clustering.py:388-390), as the indices of the N smallest of HW uniform keys — radix select +
 * order-preserving compaction, no sort.  keys [rows][HW] fp32 in [0, 1); picks [rows][N] ascending (pixels tying the N-th smallest key last). */
int gx_random_subset_picks(const float* keys, int rows, int HW, int N, int* picks, void* stream);
int gx_rotflip_map(float angle_deg, int flip, int H, int W, int* map, void* stream);
/* the maps of a whole view batch in one launch (angles / flips are host arrays of n entries) */
int gx_rotflip_map_batch(const float* angles_deg, const int* flips, int n, int H, int W, int* maps, void* stream);
/* interp: the `mode` of the reference's F.interpolate over the taps (swav_args['hf_interp'], swav_clustering.py:123-124):
 * 0 = 'nearest' (every shipped config), 1 = 'bilinear' (align_corners=False, ATen's arithmetic) */
int gx_gather_taps(int ntaps, const float* const* tap_ptrs /*host array of device ptrs*/, const int* tap_channels /*host*/,
                   const int* tap_res /*host*/, int B, int H, int W, int hlen, const int* picks, int rows_per_img,
                   const int* maps, int interp, float* X, long ldx, void* stream);
/* img [B][H][W] = L2 norm of every (rotated / flipped: maps, or NULL) pixel's descriptor with the taps bilinearly upsampled
 * (torch.norm(stack, dim=1), swav_clustering.py:361-362, under hf_interp='bilinear'; 'nearest' uses gx_row_sumsq + gx_norm_map) */
int gx_stack_norm_map_bilinear(int ntaps, const float* const* tap_ptrs, const int* tap_channels, const int* tap_res, int hlen,
                               const int* maps, int B, int H, int W, float* img, void* stream);
int gx_row_sumsq(const float* x, long rows, int cused, long ld, float* out, void* stream);
int gx_norm_map(int ntaps, const float* const* ss_ptrs /*host array*/, const int* tap_res /*host*/, const int* maps,
                int B, int H, int W, float* img, void* stream);

/* ---- pyramid projection: S1+S2+S3 factorised through the nearest-upsampling ---------
 * The projection (swav_clustering.py:171) commutes with F.interpolate(nearest) (:123): project each
 * tap at native resolution, gx_upsample2_add the levels into one full-resolution table Z
 * ([B][H*W][D]); a sampled pixel's projected descriptor is one row of Z:
 *   gx_gather_rows_normalize  zhat[row] = normalize(Z[b][map_b[picks[row]]]) (0 where filled), rnorm
 *   gx_normalize_bwd_scatter  G[b][map_b[picks[row]]] += normalize-backward(dzhat[row])   (atomics)
 *   gx_sumpool2               adjoint of gx_upsample2_add (G_{j-1} = 2x2 sums of G_j) */
int gx_upsample2_add(const float* coarse, float* fine, int B, int rc, int D, void* stream);
int gx_sumpool2(const float* fine, float* coarse, int B, int rc, int D, void* stream);
int gx_gather_rows_normalize(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps,
                             float* zhat, float* rnorm, void* stream);
int gx_normalize_bwd_scatter(const float* dzhat, const float* zhat, const float* rnorm, int B, int HW, int D,
                             const int* picks, int rows_per_img, const int* maps, float* G, float* amax, void* stream);
/* amax (optional; one float the caller has zeroed, shared by the calls that fill one G): the largest single contribution — times
 * the number of rows that can meet in one pixel it bounds |G| (the a_amax_dev of the f16x3 level GEMMs of the pyramid backward) */

/* ---- small-prototype fast path (K <= 16): the prototype layer as fused HBM-bound row kernels ----
 * gx_proto_fwd_small  = gx_gather_rows_normalize + scores = zhat Wc^T + b   (swav_clustering.py:171-175)
 * gx_proto_bwd_small  = dz_hat = dS Wc, gWc (+)= dS^T zhat, gb (+)= colsum(dS), normalise-backward,
 *                       scatter into G                                      (autograd of :171-175) */
size_t gx_proto_small_workspace_bytes(int K, int D);
int gx_proto_fwd_small(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps,
                       const float* Wc, const float* bias, int K, float* zhat, float* rnorm, float* scores,
                       void* stream);
int gx_proto_bwd_small(const float* dS, const float* zhat, const float* rnorm, int B, int HW, int D, const int* picks,
                       int rows_per_img, const int* maps, const float* Wc, int K, float* G, float* gWc, float* gb,
                       float beta, float* workspace, size_t workspace_bytes, void* stream);

/* ---- low-rank prototype backward (K <= 16): per-row gradient = K+1 coefficients per pixel ----------
 * dz(row) = sum_k (rn dS_k) Wc_k - (rn^2 <zhat, dS Wc>) Z[q]  lies in span{Wc, Z[q]} (DESIGN.md par. 4.3):
 * gx_proto_fwd_lowrank     scores, rnorm from Z rows (zhat never written); picks == NULL (rows_per_img
 *                          == HW, maps == NULL): the score table of EVERY pixel, computed once per view
 * gx_gather_scores         one patch's sampled rows of that table (the 5 patches of a step re-sample
 *                          the same 65 536 pixels: swav_clustering.py:160-171 gathers 2 KB rows 5 times)
 * gx_proto_bwd_lowrank     C[q][0..K] += coefficients (36-B atomics per row instead of 2 KB), gb
 * gx_proto_lowrank_finish  once per view: G = C[:, :K] Wc + C[:, K] .* Z ; gWc (+)= C[:, :K]^T Z
 *                          (+ the first 2x2 sum-pooling of G for the pyramid backward)
 * C is [B*HW][gx_lowrank_stride(K)] floats, zeroed by the caller each step. */
int gx_lowrank_stride(int K);
int gx_proto_fwd_lowrank(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps,
                         const float* Wc, const float* bias, int K, float* rnorm, float* scores, void* stream);
int gx_gather_scores(const float* table, const float* rn_table, const float* bias, int B, int HW, int K,
                     const int* picks, int rows_per_img, const int* maps, float* scores, float* rnorm, void* stream);
/* the whole step in single launches (both views = one 2B-image batch, all patches; rows ordered
 * (patch, view, image, sample)): gather, coefficient scatter, and — gx_swapped_ce_batch below — the loss.
 * shift > 0: the tables / the coefficient grid live `shift` pyramid levels below the picks' resolution (HW a square of
 * a power of two): with no tap above that level every pixel of a 2^shift x 2^shift block has its ancestor's descriptor
 * (swav_clustering.py:120-128: nearest upsampling), so the tables are [2B * HW / 4^shift] rows */
int gx_gather_scores_batch(const float* table, const float* rn_table, const float* bias, int B, int HW, int K,
                           const int* picks, int npatch, int N, const int* maps, int shift, float* scores, float* rnorm,
                           void* stream);
/* Cfix (optional, [rows of C][gx_lowrank_stride(K)] 64-bit integers, zeroed by the caller once per step): the DETERMINISTIC form —
 * the coefficients are accumulated as fixed-point integers (2^-44 units; integer atomics commute, float atomics do not) and C
 * is overwritten with the float value of everything accumulated in Cfix so far.  NULL: float atomics into C. */
int gx_proto_bwd_lowrank_batch(const float* dS, const float* scores, const float* rnorm, const float* bias, int B, int HW,
                               int K, const int* picks, int npatch, int N, const int* maps, int shift, float* C, long long* Cfix,
                               float* gb, float beta, float* workspace, size_t workspace_bytes, void* stream);
int gx_proto_bwd_lowrank(const float* dS, const float* scores, const float* rnorm, const float* bias, int B, int HW,
                         int K, const int* picks, int rows_per_img, const int* maps, float* C, float* gb, float beta,
                         float* workspace, size_t workspace_bytes, void* stream);
int gx_proto_lowrank_finish(const float* C, const float* Z, long npix, int rc, int D, const float* Wc, int K, float* G,
                            int accumulate, float* Gpool, float* gWc, float beta, float* workspace, size_t workspace_bytes,
                            void* stream);   /* Gpool != NULL: also the 2x2 sum-pooled G (B*rc*rc, D); accumulate: G += */
/* finest pyramid level + score table in one pass: fine += upsample2(coarse); rn_table, sc_table as
 * gx_proto_fwd_lowrank(picks = NULL) on the result.  coarse == NULL: fine is already finished
 * (gx_gemm_f32_upadd) and only the tables are written */
int gx_upsample2_add_scores(const float* coarse, float* fine, int B, int rc, int D, const float* Wc, const float* bias,
                            int K, float* rn_table, float* sc_table, void* stream);

/* ---- projection / prototype helpers ---------------------------------------------------------
 * gx_row_l2_normalize   F.normalize(p=2, dim=1) in place, rnorm out     swav_clustering.py:174
 * gx_l2_normalize_bwd   its backward (autograd of :174)                  swav_clustering.py:459
 * gx_row_argmax         out_preds.max(1)[1]                              swav_clustering.py:691
 * gx_colsum             out = alpha*colsum(x) + beta*out (prototype-bias gradient) */
int gx_row_l2_normalize(float* z, long rows, int D, long ld, float* rnorm, void* stream);
int gx_l2_normalize_bwd(float* g, const float* zhat, const float* rnorm, long rows, int D, long ld, void* stream);
int gx_row_argmax(const float* z, long rows, int D, long ld, long long* labels, void* stream);
size_t gx_colsum_workspace_bytes(int K);
int gx_colsum(const float* x, long rows, int K, long ld, float alpha, float beta, float* out, float* workspace,
              size_t workspace_bytes, void* stream);

/* ---- Sinkhorn-Knopp (log-domain scaling-vector form)   swav_clustering.py:509-544 ------------
 * The returned code is q[n][k] = softmax_k(S_nk/eps + la_k); gx_sinkhorn() leaves la[K]
 * (la = log of the reference's accumulated row scaling).  Codes via gx_sinkhorn_codes or fused
 * into gx_swapped_ce.  Sharded over ranks (SURVEY.md par. 8e): colmax -> all_reduce(MAX) of the
 * K column maxima (cmax_get / cmax_set) -> {pass -> all_reduce(SUM) of gx_sinkhorn_u(ws,K)[K]
 * -> update} x niters, `first` = 1 on the first iteration.  r (K) / c (rows) are the 'image'
 * marginals (NULL = uniform 1/K, c_uniform = 1/rows_global). */
size_t gx_sinkhorn_workspace_bytes(int K);
float* gx_sinkhorn_u(float* workspace, int K);
int gx_sinkhorn_colmax(const float* S, long rows, int K, long ld, float* workspace, size_t workspace_bytes,
                       void* stream);
int gx_sinkhorn_cmax_get(const float* workspace, int K, float* out, void* stream);
int gx_sinkhorn_cmax_set(float* workspace, int K, const float* in, void* stream);
int gx_sinkhorn_colmax_rows(const float* S, const int* rowmap, long rows, int K, long ld, float* workspace,
                            size_t workspace_bytes, void* stream);
int gx_sinkhorn_pass_rows(const float* S, const int* rowmap, long rows, int K, long ld, float eps, int first, const float* la,
                          const float* c, float c_uniform, float* lb_out, float* workspace, size_t workspace_bytes,
                          void* stream);
size_t gx_sinkhorn_first_rows_workspace_bytes(int K);
int gx_sinkhorn_first_rows(const float* S, const int* rowmap, long rows, int K, long ld, float eps, float* cmax, float* u,
                           float* workspace, size_t workspace_bytes, void* stream);
int gx_sinkhorn_rows(const float* S, const int* rowmap, long rows, int K, long ld, float eps, int niters, const float* r,
                     const float* c, float c_uniform, float* la, float* workspace, size_t workspace_bytes, void* stream);
int gx_sinkhorn_pass(const float* S, long rows, int K, long ld, float eps, int first, const float* la,
                     const float* c, float c_uniform, float* lb_out, float* workspace, size_t workspace_bytes,
                     void* stream);
int gx_sinkhorn_update(const float* u, const float* r, int K, int first, const float* workspace, float eps,
                       float* la, void* stream);
int gx_sinkhorn(const float* S, long rows, int K, long ld, float eps, int niters, const float* r, const float* c,
                float* la, float* workspace, size_t workspace_bytes, void* stream);
int gx_sinkhorn_small(const float* S, long rows, int K, long ld, float eps, int niters, const float* r,
                      const float* c, float* la, float* workspace, size_t workspace_bytes, void* stream);
                      /* same result for K <= 64 on one rank: one launch per iteration */
/* `nprob` independent, equally shaped problems in one launch per iteration: the 2 views x
 * num_patches score matrices of one optimizer step (the reference runs them one after another,
 * swav_clustering.py:389-406).  Problem i: S + i*sprob, r + i*rstride, c + i*cstride -> la + i*K.
 * Workspace: nprob * gx_sinkhorn_workspace_bytes(K). */
int gx_sinkhorn_small_batch(const float* S, int nprob, long sprob, long rows, int K, long ld, float eps, int niters,
                            const float* r, long rstride, const float* c, long cstride, float* la,
                            float* workspace, size_t workspace_bytes, void* stream);
/* The same batch with the scores row-sharded over ranks (sinkhorn_scope='global' on N GPUs): the
 * host all-reduces cmax_out (MAX) once and u_out (SUM) after every pass — 2 launches + 1 collective
 * of nprob*K floats per iteration for the whole batch.  pass: it == 0 takes the reduced maxima,
 * it >= 1 the reduced sums of pass it-1; c_uniform = 1 / (rows over all ranks). */
int gx_sinkhorn_small_batch_colmax(const float* S, int nprob, long sprob, long rows, int K, long ld, float* workspace,
                                   size_t workspace_bytes, float* cmax_out, void* stream);
int gx_sinkhorn_small_batch_pass(const float* S, int nprob, long sprob, long rows, int K, long ld, float eps, int it,
                                 const float* r, long rstride, const float* c, long cstride, float c_uniform,
                                 const float* cmax_in, const float* u_prev, float* workspace, size_t workspace_bytes,
                                 float* u_out, void* stream);
int gx_sinkhorn_small_batch_finish(int nprob, int K, float eps, int niters, const float* r, long rstride,
                                   const float* u_last, const float* workspace, float* la, void* stream);
int gx_sinkhorn_codes(const float* S, long rows, int K, long ld, float eps, const float* la, float* q, long ldq,
                      void* stream);
/* scratch: `scratch_words` 32-bit words, >= 2 (the data range as two ordered ints); from 16 384 bins on the normalising sum
 * is taken over all CUs and needs 256 more words of block partials: with fewer than 258 words such a call returns
 * GX_ERR_WORKSPACE (it never writes past what the caller said it owns). */
int gx_hist_marginal(const float* x, long n, int bins, float* hist, unsigned* scratch, int scratch_words, void* stream);
/* the same in three stages for the sharded form: range (all_reduce MIN/MAX of lohi) -> counts
 * (all_reduce SUM of hist) -> finish */
int gx_hist_range(const float* x, long n, unsigned* scratch, float* lohi, void* stream);
int gx_hist_counts(const float* x, long n, const float* lohi, int bins, float* hist, unsigned* scratch, void* stream);
int gx_hist_finish(float* hist, int bins, void* stream);

/* nprob equally shaped many-prototype problems (64 < K <= 8192) in one launch per pass (blockIdx.y = problem): the
 * per-image Sinkhorn scope at the reference's own K = 5000 (swav_clustering.py:509-544 once per image, view and patch).
 * S_all + i * prob_stride = problem i, [rows][K]; r_all / c_all NULL = uniform marginals; cmax_all NULL or the column
 * maxima [nprob][K] (from gx_gemm_f32_nt_planes' colmax_part: no pass for them); la_all [nprob][K]. */
size_t gx_sinkhorn_bigk_batch_workspace_bytes(int nprob, long rows, int K);
/* (u_first_all != NULL, with cmax_all: the first pass's column sums [nprob][K] relative to cmax_all come from the producer too) */
int gx_sinkhorn_bigk_batch(const float* S_all, int nprob, long prob_stride, long rows, int K, long ld, float eps, int niters,
                           const float* r_all, long r_stride, const float* c_all, long c_stride, const float* cmax_all,
                           const float* u_first_all, float* la_all, float* workspace, size_t workspace_bytes, void* stream);
/* nprob problems that are row selections of ONE score table (the patches of a view, global scope;
 * swav_clustering.py:509-544 once per view and patch): rowmap_all [nprob][rows] int32, each sorted ascending; one launch per
 * pass, the problems' blocks dispatched side by side so that a table row several patches drew is fetched from HBM once.
 * r_all [nprob][K] / c_all [nprob][rows] (in the order of the sorted rows) = the 'image' source pdf's marginals, or NULL = uniform with
 * c_uniform the row marginal; cmax_all / u_first_all [nprob][K] as above (gx_sinkhorn_first_rows), or NULL. */
size_t gx_sinkhorn_rows_batch_workspace_bytes(int nprob, long rows, int K);
int gx_sinkhorn_rows_batch(const float* S, const int* rowmap_all, int nprob, long rows, int K, long ld, float eps, int niters,
                           const float* r_all, const float* c_all, float c_uniform, const float* cmax_all,
                           const float* u_first_all, float* la_all, float* workspace, size_t workspace_bytes, void* stream);

/* All Sinkhorn problems of a many-prototype step ON the two per-pixel score tables (one per view; global scope:
 * swav_clustering.py:509-544 once per view and patch), every table row read ONCE per pass for all the patches that drew it
 * (csrc/gx_sktab.hip).  S0 / S1 [npix][ld]; inv [npatch][npix] int32: the row of table row q in patch p's problem, -1 = not
 * drawn; live (device int32 scalar or NULL): the drawn pixels are table rows [0, *live).  ngroups = 1: one problem per patch
 * and view (the global scope); ngroups = B with seg [B + 1] device int32: one per image too (the reference's batch-1
 * semantics), image b's live rows being table rows [seg[b], seg[b + 1]) and inv numbering a patch's samples image by image.
 * Problem index = (p * 2 + view) * ngroups + image in every per-problem array ([2 npatch ngroups][K] below).  c_all [2 npatch][rows] (row marginals in the order inv numbers a problem's rows) or NULL =
 * c_uniform; r_all [2 npatch][K] or NULL = uniform.
 *   _first : cmax = column maxima, u = sum_n exp((S_nk - cmax_k) / eps) over each problem's rows, one sweep
 *   _sweep : u = this rank's column sums of one iteration under la.  stats = NULL: every row's maximum is taken exactly, and
 *            the rows' log2-sum-exp are left in the workspace; stats = what the _update that made `la` wrote, after a sweep on
 *            the SAME workspace: the problems it vouches for are exponentiated relative to (previous log2-sum-exp + largest
 *            step of la), a bound known before the row is touched — no maxima pass
 *   _update: la = (first ? -cmax / eps : la) + log r - log u; stats (optional) [2 npatch][2] = {largest step in log2 units,
 *            int 1 if the steps lie within 64 log2 units of each other}
 *   _rebase: u *= exp((cm_local - cm_global) / eps) (sharded first pass, after the MAX all-reduce of cmax)
 *   gx_sinkhorn_table: the whole single-rank solve (niters iterations), cmax / u = scratch [2 npatch][K], stats [2 npatch][2]. */
/* shapes the table solver takes: npatch <= 5, K % 4 == 0, K > 64, and the patches' scaling vectors fit one CU's LDS — 8 columns per
 * thread (K <= 6144, npatch * 8 * 64 * ceil(K / 512) <= 32768 floats) or the wide form, 16 columns per thread (K <= 8192, npatch * K <= 40000) */
int gx_sinkhorn_table_ok(int npatch, int K);
size_t gx_sinkhorn_table_workspace_bytes(int npatch, long npix, int K, int ngroups);
int gx_sinkhorn_table_first(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch,
                            const int* live, const int* seg, int ngroups, float eps, float* cmax, float* u, float* workspace,
                            size_t workspace_bytes, void* stream);
int gx_sinkhorn_table_sweep(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch,
                            const int* live, const int* seg, int ngroups, float eps, const float* la, const float* stats,
                            const float* c_all, long rows, float c_uniform, float* u, float* workspace, size_t workspace_bytes,
                            void* stream);
int gx_sinkhorn_table_update(const float* u, const float* r_all, int nprob, int K, int first, const float* cmax, float eps,
                             float* la, float* stats, void* stream);
int gx_sinkhorn_table_rebase(float* u, const float* cm_local, const float* cm_global, long n, float eps, void* stream);
int gx_sinkhorn_table(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch, const int* live,
                      const int* seg, int ngroups, float eps, int niters, const float* r_all, const float* c_all, long rows,
                      float c_uniform, float* la, float* cmax, float* u, float* stats, float* workspace, size_t workspace_bytes,
                      void* stream);

/* ---- swapped-prediction cross-entropy forward + d(scores)   swav_clustering.py:547-570 -------
 * q_s/q_t come from the Sinkhorn log-scaling vectors la_s / la_t (never materialised).
 * loss_out[0] = loss_scale * sum_n loss_n + loss_beta * loss_out[0];
 * dS_t = -grad_scale/(2T) (q_s - softmax(S_t/T) sum_k q_s), dS_s likewise (NULL,NULL = forward only). */
size_t gx_swapped_ce_workspace_bytes(long rows, int K);
int gx_swapped_ce(const float* Ss, const float* St, long rows, int K, long ld, float eps, float temperature,
                  const float* la_s, const float* la_t, float grad_scale, float* dSs, float* dSt, long ldd,
                  float loss_scale, float loss_beta, float* loss_out, float* workspace, size_t workspace_bytes,
                  void* stream);
/* every (patch, row group) pair of a step in one launch; S_all / dS_all [npatch][2][rows_per_patch][K],
 * la_all in SinkhornBatch order; workspace >= npatch * ngroups * 512 floats */
/* many prototypes (64 < K <= 6144, 16-B aligned rows), ONE pass over the two score matrices: the same loss and dS,
 * plus gb = gb_beta * gb + column sums of (dSs + dSt) (the prototype-bias gradient, autograd of swav_clustering.py:175);
 * GX_ERR_BAD_ARG for shapes it does not take (use gx_swapped_ce + gx_colsum then) */
size_t gx_swapped_ce_fused_workspace_bytes(int K);
int gx_swapped_ce_fused(const float* Ss, const float* St, long rows, int K, long ld, float eps, float temperature,
                        const float* la_s, const float* la_t, float grad_scale, float* dSs, float* dSt, long ldd,
                        float loss_scale, float loss_beta, float* loss_out, float* gb, float gb_beta, float* workspace,
                        size_t workspace_bytes, void* stream);
int gx_swapped_ce_pixels(const float* S_all, int P, long rows, int K, long ld, const int* inv, long npix, int ngroups,
                         float eps, float temperature, const float* la_all, float grad_scale, float* dSs, float* dSt, long ldd,
                         int kpad, float loss_scale, float loss_beta, float* loss_out, float* gb, float gb_beta,
                         float* workspace, size_t workspace_bytes, void* stream);
int gx_swapped_ce_pixels_tab(const float* Ss_tab, const float* St_tab, int P, int K, long ld, const int* inv, long npix,
                             int ngroups, float eps, float temperature, const float* la_all, float grad_scale, float* dSs,
                             float* dSt, long ldd, int kpad, float loss_scale, float loss_beta, float* loss_out, float* gb,
                             float gb_beta, const int* live, const int* row_group, float* workspace, size_t workspace_bytes,
                             void* stream);
int gx_swapped_ce_batch(const float* S_all, int npatch, int ngroups, long rows_per_patch, long rows, int K, float eps,
                        float temperature, const float* la_all, float grad_scale, float* dS_all, float loss_scale,
                        float loss_beta, float* loss_out, float* workspace, size_t workspace_bytes, void* stream);

/* ---- SimCLR baseline head (baseline/hfc_with_simclr/simclr_clustering.py) -----------------------
 * gx_col_stats      BatchNorm1d batch statistics (train mode)                      :150
 * gx_bn_lrelu_fwd   BatchNorm1d + LeakyReLU(0.01)                                   :150-151
 * gx_bn_lrelu_bwd   their backward over the n = 2*batch samples                    (autograd, :270)
 * gx_simclr_ntxent  the inline loss over the first n feature rows + d loss/d out   :224-264
 * gx_row_div        z /= max(|x|, 1e-12): F.normalize(hfeat, dim=1) folded after the projection :191, :387 */
size_t gx_col_stats_workspace_bytes(int D);
int gx_col_stats(const float* x, long rows, int D, float* mean, float* var, float* workspace, size_t workspace_bytes,
                 void* stream);
int gx_bn_lrelu_fwd(const float* x, long rows, int D, const float* mean, const float* var, const float* gamma,
                    const float* beta, float eps, float slope, float* a, float* xhat, void* stream);
int gx_bn_lrelu_bwd(const float* da, const float* xhat, int n, int D, const float* var, const float* gamma,
                    const float* beta, float eps, float slope, float* dx, float* dgamma, float* dbeta, void* stream);
int gx_simclr_ntxent(const float* out, int n, int D, float temperature, float* loss, float* dout, void* stream);
int gx_row_div(float* z, long rows, int D, const float* nrm, void* stream);

/* ---- the non-linear projection heads (hfc_with_swav/swav_clustering.py:250-269; no shipped config uses them) ----
 * '1-layer' = Linear -> LeakyReLU(0.01); '2-layer' = Linear -> BatchNorm1d -> LeakyReLU(0.01) -> Linear -> BatchNorm1d -> Tanh,
 * BatchNorm in train mode throughout (the reference never calls .eval()).  Row passes of the generic path:
 * gx_gather_rows / gx_scatter_rows_add   the sampled rows of the projected table and the adjoint     :160-169, autograd
 * gx_leaky_relu / gx_leaky_relu_bwd      nn.LeakyReLU(inplace=True) and its backward from the output  :255, :262
 * gx_col_moment                          BatchNorm1d batch mean / biased variance (two calls)         :261, :266
 * gx_bn_act_fwd                          BatchNorm1d + activation (0 LeakyReLU(slope), 1 tanh)        :261-267
 * gx_bn_act_bwd_sums / _apply            their backward: column sums (all-reduced across ranks), then dx */
int gx_gather_rows(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps, float* out,
                   void* stream);
int gx_scatter_rows_add(const float* g, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps, float* G,
                        void* stream);
int gx_leaky_relu(float* x, long total, float slope, void* stream);
int gx_leaky_relu_bwd(float* dy, const float* y, long total, float slope, void* stream);
size_t gx_col_moment_workspace_bytes(int D);
int gx_col_moment(const float* x, long rows, int D, const float* center, int square, float scale, float* out,
                  float* workspace, size_t workspace_bytes, void* stream);
int gx_bn_act_fwd(const float* x, long rows, int D, const float* mean, const float* var, const float* gamma,
                  const float* beta, float eps, int act, float slope, float* a, float* xhat, void* stream);
int gx_bn_act_bwd_sums(const float* da, const float* xhat, long rows, int D, const float* gamma, const float* beta, int act,
                       float slope, float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes, void* stream);
int gx_bn_act_bwd_apply(const float* da, const float* xhat, long rows, int D, const float* var, const float* gamma,
                        const float* beta, float eps, int act, float slope, const float* dgamma, const float* dbeta,
                        float n_total, float* dx, void* stream);

/* ---- optimizer: apex LARC(clip=False) around SGD(momentum)   swav_clustering.py:286-292 ------ */
size_t gx_sumsq_workspace_bytes(void);
int gx_sumsq(const float* x, long n, float* out, float* workspace, size_t workspace_bytes, void* stream);
int gx_larc_sgd(float* p, const float* g, float* buf, long n, const float* norms2 /*{sum p^2, sum g^2}*/, float lr,
                float momentum, float trust, int first_step, int use_larc, void* stream);

/* ---- hfc_kmeans baseline (SURVEY.md par. 8f rank 4) ---------------------------------------------
 * FlatKMeansHFC (baseline/hfc_kmeans/hfc_kmeans_clustering.py:127-209) = one sklearn.cluster.KMeans per
 * generator layer on the (pixels, channels) panel of that layer; HFCPreprocessor
 * (baseline/hfc_kmeans/segmentor.py:66-232) builds the panels and turns labels into one-hot maps.
 * The per-iteration contractions (scores X.C^T, per-cluster sums onehot^T.X) are gx_gemm_f32 calls; these
 * entry points are the rest of sklearn's dense float32 path, all with fixed-order reductions:
 *   gx_kmeans_center_panel   X = [tap1 | tap2] - mean (KMeans.fit: X -= X.mean(0)), xsq = row_norms(X)^2;
 *                            taps are NHWC levels, i.e. already (pixels, channels) row-major
 *   gx_kmeans_assign         E step of _update_chunk_dense: label = argmin_k(csq[k] - 2 S[n][k]), first
 *                            minimum wins; optional one-hot rows (for the sums GEMM), label histogram
 *                            counts[K] and the number of rows whose label differs from labels_prev
 *   gx_kmeans_accumulate     M step sums, `centers_new[label] += X[i]`: one pass over X with per-block LDS
 *                            accumulators (K <= 256: gx_kmeans_accumulate_supported), chunk partials added in
 *                            a fixed order — no atomics; larger K: onehot^T.X through gx_gemm_f32
 *   gx_kmeans_update         _average_centers + _center_shift: centers_new = sums / counts (may alias centers_old),
 *                            shift2[k] = |new - old|^2, csq[k] = |new|^2
 *   gx_kmeans_row_dist       |X[n] - centers[labels[n]]|^2 by direct differences (+ their float64 sum):
 *                            _inertia_dense and the distances of _relocate_empty_clusters_dense
 *   gx_kmeans_pp_*           greedy k-means++ (_kmeans_plusplus): sample = searchsorted(cumsum_f64(closest),
 *                            targets) clipped; with G = X.X[cand]^T (rows x T) the squared distances are
 *                            max(xsq[n] + xsq[cand[t]] - 2 G[n][t], 0); pots[t] = sum_n min(closest, D_t)
 *                            (closest NULL: sum D_t); commit: closest = min(closest, D_best) (first: D_best)
 *   gx_kmeans_label_maps     FlatKMeansHFC.predict's one-hot maps of all layers, nearest-resized to
 *                            out_size, NHWC [B][out][out][sum K], values hi / lo (1 / -1 = the `*2-1`
 *                            of predict_hfc_vectors, segmentor.py:230) */
int gx_kmeans_center_panel(const float* tap1, int C1, const float* tap2, int C2, long rows, const float* mean, float* X,
                           float* xsq, void* stream);
int gx_kmeans_assign(const float* S, long rows, int K, long ld, const float* csq, const int* labels_prev, int* labels,
                     float* onehot, long ld_onehot, int* counts, int* nchanged, void* stream);
int gx_kmeans_accumulate_supported(int K);
size_t gx_kmeans_accumulate_workspace_bytes(long rows, int K, int C);
int gx_kmeans_accumulate(const float* X, long rows, int C, long ldx, const int* labels, int K, float* sums,
                         void* workspace, size_t workspace_bytes, void* stream);
int gx_kmeans_update(const float* sums, const int* counts, int K, int C, const float* centers_old, float* centers_new,
                     float* csq, float* shift2, void* stream);
int gx_kmeans_row_dist(const float* X, long rows, int C, long ldx, const float* centers, int K, const int* labels,
                       float* dist, double* total, void* stream);
size_t gx_kmeans_pp_workspace_bytes(int T);
int gx_kmeans_pp_pots(const float* G, long rows, int T, long ld, const float* xsq, const int* cand, const float* closest,
                      double* pots, void* workspace, size_t workspace_bytes, void* stream);
int gx_kmeans_pp_commit(const float* G, long rows, int T, long ld, const float* xsq, const int* cand, int best, int first,
                        float* closest, void* stream);
int gx_kmeans_pp_sample(const float* closest, long rows, const double* targets /*host, T values*/, int T, int* cand,
                        void* workspace, size_t workspace_bytes, void* stream);
int gx_kmeans_label_maps(int nlayers, const int* const* labels /*host array of device ptrs*/, const int* res /*host*/,
                         const int* K /*host*/, int B, int out_size, float hi, float lo, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
