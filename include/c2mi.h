/*
 * c2mi.h -- C-ABI of the MI355X (gfx950) Celligner2 training-step library (libc2mi.so).
 *
 * The reference (broadinstitute/celligner2) has no FFI: its hot path is Python calling
 * stock torch ops.  Each entry point below replaces the torch-op sequence of the cited
 * reference lines (paths relative to the reference root) with hand-written HIP kernels.
 * The host mirror (celligner2_amd/) binds these with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - matrices are row-major, leading dimensions (ld*) are in ELEMENTS;
 *   - `dtype` arguments: C2_F32 or C2_BF16 (storage type of GEMM operands / activations);
 *     accumulation, losses, parameter gradients and optimizer state are always fp32;
 *   - `stream` is a hipStream_t passed as void*; calls enqueue work and return, they never
 *     synchronise the device and never allocate device memory (workspaces are caller-owned);
 *   - return value: 0 = ok, non-zero = error; c2_last_error() returns a thread-local message;
 *   - "slabs": a split reduction writes `nslab` partial fp32 matrices, slab s at
 *     base + s*slab_stride elements; consumers take (nslab, slab_stride) and sum them.
 */
#ifndef C2MI_H
#define C2MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2_F32 0
#define C2_BF16 1
/* GEMM entry points only: fp32 operands in memory, each split in-kernel into hi+lo bf16 and multiplied as three
 * bf16 MFMA products (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-16 relative product error at 3/16 of the
 * fp32-MFMA cost (gfx950 has no TF32).  Used for the narrow hidden-layer GEMMs of bf16 mode. */
#define C2_F32X3 2
/* GEMM entry points only: the same bf16x3 product with the split done ONCE by the producer: each operand pointer
 * addresses two bf16 planes [2][rows][ld] (hi, then lo = x - hi; c2_split_bf16), rows = the operand's own row
 * count.  Staged by LDS-DMA like C2_BF16.  Needs 16-byte aligned planes, ld % 8 == 0 and a reduction length that
 * is a multiple of 8 and >= 64 (c2_gemm_tile returns 0 otherwise). */
#define C2_BF16X3P 3
/* IEEE half: operands of the GEMM entry points that take a dtype (staged and multiplied like C2_BF16, on
 * v_mfma_f32_*_f16: 11 significant bits instead of 8 at the same matrix rate; no (XC, RC) form), and an output type of
 * c2_prep_input / c2_gather_rows / c2_cast_f16.  bf16 mode reads the encoder's first-layer input (log(1 + x) <= 89, in
 * range) and weight this way: at 19 000 genes the 2^-9 rounding of a bf16 encoder input carried the gradients of the
 * latent heads past the 1e-2 bound. */
#define C2_F16 4

#define C2_MMD_MAX_PARTS 64 /* max distinct condition rows per batch for the MMD partition */

const char* c2_last_error(void);
int c2_version(void);

/* x_log = log(1 + x) (celligner2/model/celligner2.py:269-274) or a plain copy/convert, minus center[col] when center is
 * not NULL (see c2_col_mean).  out dtype f32 or bf16.  rows x cols, ldx/ldo in elements. */
int c2_prep_input(const float* x, int64_t rows, int64_t cols, int64_t ldx, void* out, int out_dtype,
                  int64_t ldo, int apply_log, const float* center, void* stream);

/* Centred encoder input of bf16 mode.  The encoder's first Linear (modules.py:77-107, 281-290) computes
 * h = t(x) W^T + b with t(x) = log(1 + x) or x (celligner2.py:269-274).  For any vector c [G]:
 *   h  = (t(x) - c) W^T + (b + W c)                 c2_prep_input / c2_gather_rows (center) + c2_gemv_bias
 *   dW = dh^T (t(x) - c) + (sum_b dh_b) (x) c       c2_linear_wgrad* + c2_rank1_add
 * which is the same layer, but the operand rounded to bf16 is the centred part: expression values share a large
 * per-gene level, and its 2^-9 rounding is what carried bf16 mode's gradient error past 1e-2 at 19 000 genes.
 *   c2_col_sum:   sums[col] = sum over rows of t(x)[row][col] in double (ws: c2_col_sum_ws(rows, cols) doubles); summed
 *                 in double the result does not depend on the order, so ranks that add their sums (an all-reduce) centre on
 *                 the same bits as one process on the whole batch
 *   c2_col_mean_from_sums: out[col] = (float)(sums[col] / total_rows)
 *   c2_gemv_bias: out[n] = bias[n] + sum_k W[n][k] c[k]          (fp32; bias may be NULL)
 *   c2_rank1_add: G[n][k] += r[n] * c[k] */
size_t c2_col_sum_ws(int64_t rows, int64_t cols);
int c2_col_sum(const float* x, int64_t ldx, int64_t rows, int64_t cols, int apply_log, double* sums, double* ws, void* stream);
int c2_col_mean_from_sums(const double* sums, int64_t cols, int64_t total_rows, float* out, void* stream);
int c2_gemv_bias(const float* W, int64_t ldw, int64_t N, int64_t K, const float* c, const float* bias, float* out,
                 void* stream);
int c2_rank1_add(float* G, int64_t ldg, int64_t N, int64_t K, const float* r, const float* c, void* stream);

/* nn.Linear forward without the bias/cond/norm tail (modules.py:100,126,296,314-315,443,469,481,617):
 *   Y[s][m][n] = sum_{k in split s} A[m][k] * W[n][k]      (+ bias[n] when bias != NULL; needs nslab == 1)
 * A [M,K] and W [N,K] have dtype `dtype`; Y is fp32. */
int c2_linear_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, int dtype, int64_t M, int64_t N,
                  int64_t K, float* Y, int64_t ldy, int nslab, int64_t slab_stride, const float* bias,
                  void* stream);

/* c2_linear_fwd with the output rounded to IEEE halves (Y16: [M][ldy] halves, unsplit, bias optional): the decoder's
 * logits in bf16 mode (modules.py:478-482 feeds them to Softmax; celligner2/model/losses.py:27-65 reads the result).  A
 * half keeps 2^-12 of a logit, below the 2^-9 at which the same mode rounds dlogits. */
int c2_linear_fwd_f16out(const void* A, int64_t lda, const void* W, int64_t ldw, int dtype, int64_t M, int64_t N,
                         int64_t K, void* Y16, int64_t ldy, const float* bias, void* stream);

/* autograd of the same Linear w.r.t. its input:
 *   dA[s][m][k] = alpha * sum_{n in split s} dY[m][n] * W[n][k]      alpha = *alpha_dev or 1 */
int c2_linear_dgrad(const void* dY, int64_t lddy, const void* W, int64_t ldw, int dtype, int64_t M, int64_t N,
                    int64_t K, float* dA, int64_t ldda, int nslab, int64_t slab_stride,
                    const float* alpha_dev, void* stream);

/* autograd of the same Linear w.r.t. its weight:
 *   dW[s][n][k] = alpha * sum_{m in split s} dY[m][n] * A[m][k]   (+ rank1_row[n] * rank1_col[k] in split 0)
 * rank1_row [N] / rank1_col [K] (both or neither; also on the _rows, _rows_at and _bt forms): a rank-one term added in the
 * epilogue -- the centred first layer's (sum_b dh_b) (x) c (see c2_col_sum), without a pass of its own over dW. */
int c2_linear_wgrad(const void* dY, int64_t lddy, const void* A, int64_t lda, int dtype, int64_t M, int64_t N,
                    int64_t K, float* dW, int64_t lddw, int nslab, int64_t slab_stride,
                    const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream);

/* Row-indexed forms of the gene-width forward and weight gradient (no counterpart in the reference, which gathers
 * every batch on the host: trainers/celligner2/_utils.py:188-225): the batch is rows row_index[0..M) of a RESIDENT
 * bf16 matrix A_full [a_rows, K] and is read where it lives by the LDS-DMA kernel (the per-lane source address carries
 * the row), so no gathered copy of the encoder input is made.
 *   c2_linear_fwd_rows:    Y[m][n]  = sum_k A_full[row_index[m]][k] W[n][k] (+ bias)
 *   c2_linear_wgrad_rows:  dW[n][k] = alpha * sum_m dY[m][n] A_full[row_index[m]][k]
 * bf16 operands only, shapes the 256x256 kernel takes (c2_gemm_tile(C2_BF16, ...) == 256), a_rows*lda < 2^31.
 * row_index: int32, device; for the weight gradient it must be readable up to M rounded up to a multiple of 64
 * (pad with any valid row).  Other shapes: error -- gather the rows (c2_gather_rows) and use the plain entry points. */
int c2_linear_fwd_rows(const void* A_full, int64_t lda, int64_t a_rows, const int32_t* row_index, const void* W,
                       int64_t ldw, int dtype /* C2_BF16 or C2_F16 */, int64_t M, int64_t N, int64_t K, float* Y,
                       int64_t ldy, int nslab, int64_t slab_stride, const float* bias, void* stream);
int c2_linear_wgrad_rows(const void* dY, int64_t lddy, const void* A_full, int64_t lda, int64_t a_rows,
                         const int32_t* row_index, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw, int nslab,
                         int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream);

/* The gene-width weight gradient (nn.Linear backward, modules.py:281-290, 467-482) with ONE operand handed over
 * TRANSPOSED, so that its reduction index (the cells of the batch) is contiguous and the 256x256 kernel reads it row-wise
 * and multiplies on v_mfma_f32_16x16x32_bf16 -- the layout of the forward / input-gradient launches, measured faster than
 * the weight gradient whose two operands both come through the transposing LDS read.  The transposed operand is the narrow
 * one ([B, H]: 8 MB at 4096 x 1024), so making the copy is cheap; the gene-width operand stays as it is.
 *   c2_linear_wgrad_bt:       dW[n][k] = alpha * sum_m dY[m][n] At[k][m]                  At: [K, M] bf16
 *   c2_linear_wgrad_rows_at:  dW[n][k] = alpha * sum_m dYt[n][m] A_full[row_index[m]][k]  dYt: [N, M] bf16
 * bf16, shapes the 256x256 kernel takes only (c2_gemm_tile(C2_BF16, N, K, M) == 256, 16-byte aligned rows); otherwise an
 * error -- use c2_linear_wgrad / c2_linear_wgrad_rows.  Results equal those entry points' up to fp32 summation order
 * inside the MFMA (same products, same K-tile order). */
int c2_linear_wgrad_bt(const void* dY, int64_t lddy, const void* At, int64_t ldat, int64_t M, int64_t N, int64_t K,
                       float* dW, int64_t lddw, int nslab, int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream);
int c2_linear_wgrad_rows_at(const void* dYt, int64_t lddyt, const void* A_full, int64_t lda, int64_t a_rows,
                            const int32_t* row_index, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw, int nslab,
                            int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col, void* stream);

/* Block-tile edge (128 or 256) the three GEMM entry points above use for operand dtype `dtype` and an
 * [M,N] output reduced over R: callers size `nslab` from it so that a launch fills the 256 CUs.
 * 0 = this dtype cannot take the shape (C2_BF16X3P only).
 * (No counterpart in the reference: torch picks its own BLAS blocking.) */
int c2_gemm_tile(int dtype, int64_t M, int64_t N, int64_t R);
/* 1 when an [M,N] = sum over R problem with fp32 operands (C2_F32 / C2_F32X3) runs on the one-thread-per-output kernel (at
 * most 2^21 multiply-adds): a caller that would split such operands into bf16 planes (C2_BF16X3P) should not. */
int c2_gemm_is_tiny(int64_t M, int64_t N, int64_t R);

/* The three Linear GEMMs on C2_BF16X3P operands whose two planes are given by TWO POINTERS each (hi, lo) instead of one
 * pointer and the convention "lo follows hi after rows x ld elements": operands whose planes live in padded storage -- a
 * width that is not a multiple of 8 (the reference's hidden width 820, latent widths 26 / 10) gets a row pitch rounded up
 * to 8 and zero pad columns, a weight that is also reduced over its rows (the input gradient) zero pad ROWS, and the
 * reduction then runs over the padded length, adding exact zeros.  Same kernels and arguments as c2_linear_fwd /
 * c2_linear_dgrad / c2_linear_wgrad with dtype C2_BF16X3P otherwise (no counterpart in the reference). */
int c2_linear_fwd_planes(const void* A_hi, const void* A_lo, int64_t lda, const void* W_hi, const void* W_lo, int64_t ldw,
                         int64_t M, int64_t N, int64_t K, float* Y, int64_t ldy, int nslab, int64_t slab_stride,
                         const float* bias, void* stream);
int c2_linear_dgrad_planes(const void* dY_hi, const void* dY_lo, int64_t lddy, const void* W_hi, const void* W_lo,
                           int64_t ldw, int64_t M, int64_t N, int64_t K, float* dA, int64_t ldda, int nslab,
                           int64_t slab_stride, const float* alpha_dev, void* stream);
int c2_linear_wgrad_planes(const void* dY_hi, const void* dY_lo, int64_t lddy, const void* A_hi, const void* A_lo,
                           int64_t lda, int64_t M, int64_t N, int64_t K, float* dW, int64_t lddw, int nslab,
                           int64_t slab_stride, const float* alpha_dev, const float* rank1_row, const float* rank1_col,
                           void* stream);

/* Input gradient and weight gradient of ONE tiny fp32 layer in one launch (no counterpart in the reference, whose autograd
 * launches the two GEMMs of nn.Linear's backward, modules.py:77-107, separately):  dA[M,K] = alpha * dY[M,N] W[N,K] and
 * dW[N,K] = alpha * dY^T A[M,K], alpha = *alpha_dev (1 when NULL).  Both problems must pass c2_gemm_is_tiny; same
 * arithmetic as c2_linear_dgrad / c2_linear_wgrad on such problems, bit for bit. */
int c2_linear_bwd_tiny(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* A, int64_t lda, int64_t M,
                       int64_t N, int64_t K, float* dA, int64_t ldda, float* dW, int64_t lddw, const float* alpha_dev,
                       void* stream);

/* fp32 [rows, cols] -> two bf16 planes dst[0] = hi = bf16(x), dst[1] = lo = bf16(x - hi); dst is [2][rows][ldd].
 * Producer side of C2_BF16X3P (no counterpart in the reference, which multiplies in fp32). */
int c2_split_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst, int64_t ldd, void* stream);
/* ... with the two planes at addresses of their own (padded storage, see c2_linear_fwd_planes): [rows][ldd] each. */
int c2_split_bf16_planes(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst_hi, void* dst_lo, int64_t ldd,
                         void* stream);

/* out[i] = sum_s slabs[s][i] (fp32), n elements: closes a split wgrad. */
int c2_sum_slabs(const float* slabs, int nslab, int64_t slab_stride, float* out, int64_t n, void* stream);

/* Several split reductions closed by one launch: out[i] = sum_s slabs[s * slab_stride + i] for i < n, per job.  The
 * narrow layers' weight gradients of a backward pass (each a split-K GEMM into slabs) are summed together at its end
 * instead of by one small launch each.  jobs_host: HOST array, copied by value into the launch. */
typedef struct {
    const float* slabs;
    float* out;
    int64_t slab_stride;
    int64_t n;
    int32_t nslab;
} c2_slab_sum;
int c2_sum_slabs_multi(const c2_slab_sum* jobs_host, int n_jobs, void* stream);
/* The same for a [rows, cols] block of a wider matrix: out[r*ldo + c] = sum_s slabs[s*slab_stride + r*lds + c]. */
int c2_sum_slabs_2d(const float* slabs, int nslab, int64_t slab_stride, int64_t rows, int64_t cols, int64_t lds,
                    float* out, int64_t ldo, void* stream);

/* Tail of one layer block (modules.py:77-107,124-130 cond add; :298-312 norm/act/dropout):
 *   h = sum_s Y[s] (+ bias) (+ sum_j cond[b][j] * Wc[:,j]);  LayerNorm(no affine, eps 1e-5, biased var)
 *   when use_ln; ReLU; dropout (keep-mask u8, scale 1/(1-p)) when mask != NULL.
 * Writes act (dtype act_dtype) and, for the backward, yhat (fp32, normalised pre-activation) and
 * rstd (fp32 [B]).  cond is the int64 multi-hot matrix [B,C] (row stride C).
 * act_planes: NULL, or (fp32 act only) [2][B][H] bf16: the activation again as hi / lo planes, the operand form
 * of C2_BF16X3P GEMMs (saves a c2_split_bf16 pass); needs H % 4 == 0, H <= 1024 and 16-byte aligned rows.
 * ld_planes: row pitch of the planes in elements (0: H), the low plane B x ld_planes elements behind the high one --
 * planes in zero-padded storage for widths that are not multiples of 8 (c2_linear_fwd_planes); % 4 == 0. */
int c2_epilogue_fwd(const float* Y, int64_t ldy, int nslab, int64_t slab_stride, const float* bias,
                    const int64_t* cond, int C, const float* Wc, int64_t ldwc, const uint8_t* mask,
                    float p_drop, int use_ln, int64_t B, int64_t H, void* act, int act_dtype, int64_t ldact,
                    float* yhat, float* rstd, void* act_planes, int64_t ld_planes, void* stream);

/* Backward of the same tail:  dact = sum_s dA[s] (+ extra_scale * extra); through dropout, ReLU, LayerNorm.
 * Writes dh (dtype dh_dtype), the gradient w.r.t. the Linear output; dh_planes as act_planes above. */
int c2_epilogue_bwd(const float* dA, int64_t ldda, int nslab, int64_t slab_stride, const float* extra,
                    int64_t ldextra, const float* extra_scale_dev, const uint8_t* mask, float p_drop,
                    int use_ln, const float* yhat, const float* rstd, int64_t B, int64_t H, void* dh,
                    int dh_dtype, int64_t lddh, void* dh_planes, int64_t ld_planes, void* stream);

/* Bias / condition-weight gradients of a block (autograd of modules.py:100-106):
 *   db[h] = alpha * sum_b dh[b][h];   dWc[h][j] = alpha * sum_b dh[b][h] * cond[b][j]
 * Either output may be NULL.  ws: fp32 workspace of c2_bias_cond_grad_ws(B,H,C) elements. */
size_t c2_bias_cond_grad_ws(int64_t B, int64_t H, int C);
int c2_bias_cond_grad(const void* dh, int dh_dtype, int64_t lddh, const int64_t* cond, int C, int64_t B,
                      int64_t H, float* db, float* dWc, int64_t lddwc, const float* alpha_dev, float* ws,
                      void* stream);

/* Reparameterisation + KL (othermodels/base/_base.py:464-480; celligner2.py:311-322):
 *   v = exp(lv)+1e-4; z = mu + sqrt(v)*eps;  kl (default) = (1/Bn) sum 0.5(v+mu^2-1-log v),
 *   use_own_kl: -0.5 sum(1+lv-mu^2-v).  inv_B = 1/Bn with Bn the GLOBAL batch rows.
 * z32 (fp32) and zt (dtype zt_dtype, may be NULL) are written; the KL value is ADDED to *kl_out
 * (zero it first) through a deterministic two-stage reduction using ws (>= c2_reduce_ws() floats). */
size_t c2_reduce_ws(void);
int c2_reparam_kl_fwd(const float* mu, const float* lv, const float* eps, int64_t B, int64_t L, int64_t ld,
                      float inv_B, int use_own_kl, float* z32, void* zt, int zt_dtype, int64_t ldzt,
                      float* kl_out, float* ws, void* stream);

/* Backward:  dz = sum_s dZ[s] (+ extra_scale*extra);  dmu = dz + g_kl*dKL/dmu;
 *   dlv = dz*eps*exp(lv)/(2 sqrt(v)) + g_kl*dKL/dlv.   g_kl = *g_kl_dev.  Outputs dtype out_dtype. */
int c2_reparam_kl_bwd(const float* dZ, int64_t lddz, int nslab, int64_t slab_stride, const float* extra,
                      int64_t ldextra, const float* extra_scale_dev, const float* mu, const float* lv,
                      const float* eps, int64_t B, int64_t L, int64_t ld, float inv_B, int use_own_kl,
                      const float* g_kl_dev, void* dmu, void* dlv, int out_dtype, int64_t ldout, void* stream);

/* NB reconstruction loss, forward AND backward in one pass over [B,G]
 * (celligner2.py:298-309, losses.py:27-65, _utils.py:8-24 incl. the one-hot quirk):
 *   p = softmax(logits); mu = sizefactor*p; theta_b = exp(q_b . theta^T) with q_b the scatter-quirk
 *   vector of the multi-hot row; ll as losses.py:57-65; recon = -(1/Bn) sum goodloc*ll.
 * Adds recon to *loss_out; writes dlogits (dtype dl_dtype, unit upstream gradient) and dtheta [G,C].
 * ws: c2_recon_nb_ws(B,G,C) floats.  status_dev: int32 word, set to 1 if a cond value is outside [0,C), to 2 if a
 * dispersion exp(theta) is below 1e-17 or not a number (outside the loss kernel's arithmetic).
 * lgx1_rowsum: NULL, or [B] with sum_g goodloc*lgamma(x+1) of each row (c2_nb_row_const): that term of
 * losses.py:60-64 depends on the data only, so a resident dataset evaluates it once instead of every step.
 * row_index: NULL, or [B] int64: x and goodloc are then the RESIDENT matrices and row b of the batch is their row
 * row_index[b] (logits, sizefactor, cond, lgx1_rowsum, dlogits stay batch-ordered): the loss reads the counts
 * where they live instead of from a gathered copy. */
size_t c2_recon_nb_ws(int64_t B, int64_t G, int C);
int c2_recon_nb_fwd_bwd(const float* logits, int64_t ldl, const float* x, int64_t ldx, const uint8_t* goodloc,
                        int64_t ldg, const float* sizefactor, const float* theta, const int64_t* cond, int C,
                        int64_t B, int64_t G, float inv_B, float* loss_out, void* dlogits, int dl_dtype,
                        int64_t lddl, float* dtheta, float* ws, int32_t* status_dev, const float* lgx1_rowsum,
                        const int64_t* row_index, void* stream);

/* c2_recon_nb_fwd_bwd, and in the same pass dbias[g] = sum_b dlogits[b][g] (fp32, unit upstream gradient): the bias
 * gradient of the Linear that produced the logits (modules.py:471-477, decoder.recon_decoder; autograd's column sum
 * of the output gradient), taken from the values before they are rounded to dl_dtype.  The register-resident form of
 * the loss kernel accumulates it beside the row walk; other shapes run a column-sum pass over dlogits. */
/* l_dtype: C2_F32, or C2_F16 -- logits as IEEE halves [B][ldl] (c2_linear_fwd_f16out), taken by the register-resident
 * forms of the kernel only: bf16 dlogits, 16-byte aligned rows and 18 432 < G <= 20 480 genes (the benchmark widths);
 * c2_recon_nb_logits_f16_ok(B, G) says whether a shape qualifies, any other use of C2_F16 is an error. */
int c2_recon_nb_logits_f16_ok(int64_t B, int64_t G);
int c2_recon_nb_fwd_bwd_dbias(const void* logits, int l_dtype, int64_t ldl, const float* x, int64_t ldx, const uint8_t* goodloc,
                              int64_t ldg, const float* sizefactor, const float* theta, const int64_t* cond, int C,
                              int64_t B, int64_t G, float inv_B, float* loss_out, void* dlogits, int dl_dtype,
                              int64_t lddl, float* dtheta, float* ws, int32_t* status_dev, const float* lgx1_rowsum,
                              const int64_t* row_index, float* dbias, void* stream);

/* ZINB reconstruction loss, forward and backward (celligner2.py:286-296, losses.py:68-124): as the NB entry point
 * with a second logit matrix `pi` (decoder.dropout_decoder, modules.py:471-477) and its gradient dpi:
 *   x < 1e-8: ll = softplus(-pi + theta (log(theta+eps) - log(theta+mu+eps))) - softplus(-pi)
 *   x > 1e-8: ll = NB log-likelihood - pi - softplus(-pi).
 * dlogits and dpi share d_dtype.  ws: c2_recon_nb_ws(B,G,C) floats. */
int c2_recon_zinb_fwd_bwd(const float* logits, int64_t ldl, const float* pi, int64_t ldpi, const float* x, int64_t ldx,
                          const uint8_t* goodloc, int64_t ldg, const float* sizefactor, const float* theta,
                          const int64_t* cond, int C, int64_t B, int64_t G, float inv_B, float* loss_out, void* dlogits,
                          void* dpi, int d_dtype, int64_t lddl, int64_t lddpi, float* dtheta, float* ws,
                          int32_t* status_dev, const int64_t* row_index, void* stream);

/* c2_recon_zinb_fwd_bwd with the data-only term handed in: lgx1_rowsum[b] = sum_g goodloc * lgamma(x + 1) of batch row b
 * (c2_nb_row_const; a zero count adds lgamma(1) = 0, so the NB path's per-row constant is also the ZINB one) -- the kernel
 * then evaluates no lgamma(x + 1).  Serves a device-resident dataset; no counterpart in the reference. */
int c2_recon_zinb_fwd_bwd_rc(const float* logits, int64_t ldl, const float* pi, int64_t ldpi, const float* x, int64_t ldx,
                             const uint8_t* goodloc, int64_t ldg, const float* sizefactor, const float* theta,
                             const int64_t* cond, int C, int64_t B, int64_t G, float inv_B, float* loss_out, void* dlogits,
                             void* dpi, int d_dtype, int64_t lddl, int64_t lddpi, float* dtheta, float* ws,
                             int32_t* status_dev, const int64_t* row_index, const float* lgx1_rowsum, void* stream);

/* out[b] = sum_g goodloc[b][g] * lgamma(x[b][g] + 1): the data-only term of the NB log-likelihood
 * (losses.py:60-64, `- torch.lgamma(x + 1)`), for c2_recon_nb_fwd_bwd's lgx1_rowsum. */
int c2_nb_row_const(const float* x, int64_t ldx, const uint8_t* goodloc, int64_t ldg, int64_t B, int64_t G,
                    float* out, void* stream);

/* MSE reconstruction (intent of celligner2.py:283-285; losses.py:9-24):  r = relu(logits);
 *   recon = (1/Bn) sum goodloc*(r-x)^2;  dlogits = (2/Bn)*goodloc*(r-x)*[logits>0]. */
int c2_recon_mse_fwd_bwd(const float* logits, int64_t ldl, const float* x, int64_t ldx, const uint8_t* goodloc,
                         int64_t ldg, int64_t B, int64_t G, float inv_B, float* loss_out, void* dlogits,
                         int dl_dtype, int64_t lddl, float* ws, void* stream);

/* classifier_hb_loss (losses.py:318-333), forward and backward; MUTATES classes in place (-1 -> 0)
 * exactly as the reference does.  du: unit-upstream gradient, dtype du_dtype. */
int c2_classifier_loss_fwd_bwd(const float* u, int64_t ldu, int64_t* classes, int64_t B, int P, float betaclass,
                               float* loss_out, void* du, int du_dtype, int64_t lddu, float* ws, void* stream);

/* partition() of _utils.py:27-36 + the pair loops of losses.py:232-248, as per-row partition ids and a
 * pair-weight table:  mmd = beta * sum_{a,b rows} Wt[part[a]][part[b]] * K(f_a, f_b).
 * part_id [B] int32 (order of torch.unique(dim=0); main-dataset rows merged into partition 0);
 * wt [C2_MMD_MAX_PARTS^2] fp32.  main_col = -1 for none; boundary = -1 for none.
 * status_dev set non-zero where the reference would raise (too many/few partitions, empty main).
 * ws: int64 workspace of c2_partition_ws(B) BYTES. */
size_t c2_partition_ws(int64_t B);
int c2_partition(const int64_t* cond, int64_t B, int C, int n_conditions, int main_col, int boundary,
                 int32_t* part_id, float* wt, void* ws, int32_t* status_dev, void* stream);

/* Multi-scale RBF MMD (losses.py:127-208) over rows [row0,row0+Bl) of feat against ALL Bg rows, forward and
 * backward:   loss += beta * sum_{i local, j} wt*K;   dfeat[i] (local rows, fp32, [Bl,D]) for unit upstream.
 * feat dtype f_dtype [Bg, ldf].  ws: c2_mmd_ws(Bl,Bg,D) floats. */
size_t c2_mmd_ws(int64_t Bl, int64_t Bg, int64_t D);
int c2_mmd_fwd_bwd(const void* feat, int f_dtype, int64_t ldf, const int32_t* part_id, const float* wt,
                   int64_t row0, int64_t Bl, int64_t Bg, int64_t D, float beta, float* loss_out, float* dfeat,
                   int64_t lddf, float* ws, void* stream);

/* The entry point that allocates nothing and takes no workspace: c2_mmd_fwd_bwd with its scratch (row norms, loss
 * partials, gradient slabs) in a file-scope device array of C2_MMD_STATIC_FLOATS floats (4 MiB, held by every GPU the
 * library runs on until the process ends).  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate or carry a workspace -- code under HIP-graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same arguments, results bit-identical to c2_mmd_fwd_bwd.  Sizes whose c2_mmd_ws(Bl,Bg,D) exceeds
 * C2_MMD_STATIC_FLOATS are rejected with an error (e.g. 512 x 512 rows x 64 features or 1024 x 1024 x 16 fit, 4096 x
 * 4096 x 64 does not).  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * host threads) race; calls on one stream are ordered and safe.  Calls no hipMalloc / hipFree and launches on `stream`.
 * c2_defer_loss_sums does not apply here: the loss partials live in the shared array, so their sum is always launched by
 * this call (loss_out is complete in stream order when it returns). */
#define C2_MMD_STATIC_FLOATS (1u << 20)
size_t c2_mmd_static_floats(void);
int c2_mmd_fwd_bwd_static(const void* feat, int f_dtype, int64_t ldf, const int32_t* part_id, const float* wt,
                          int64_t row0, int64_t Bl, int64_t Bg, int64_t D, float beta, float* loss_out, float* dfeat,
                          int64_t lddf, void* stream);

/* torch.optim.Adam step as trainer.py:262-266,316-323 configures it (L2 weight decay added to the gradient,
 * optional clip_grad_value_), over a table of tensors.  desc_host: array of n c2_adam_tensor (HOST memory,
 * copied by value into the launch).  step_t >= 1.  shadow (bf16 copy of the updated weights, may be NULL; with
 * shadow_lo the two bf16 planes) is refreshed in the same pass for the bf16-operand GEMMs.  Hyper-parameters are doubles so that 1-beta and the bias
 * corrections are formed exactly as torch forms them (in double, then rounded to fp32). */
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    void* shadow_bf16; /* NULL or [n] bf16 */
    int64_t n;
    void* shadow_lo;   /* NULL, or [n] bf16: the low plane lo = bf16(p - hi) next to shadow_bf16 = hi = bf16(p), the
                          C2_BF16X3P operand form of the fp32 weights of the narrow layers */
    int32_t shadow_f16; /* non-zero: shadow_bf16 addresses [n] IEEE halves (C2_F16) instead of bf16; no shadow_lo */
    int32_t shadow_cols; /* 0: the shadows are dense [n]; else the parameter is a matrix with this many columns */
    int64_t shadow_ld;   /* and its shadows have rows shadow_ld (% 4 == 0) elements apart (copies in zero-padded storage) */
} c2_adam_tensor;
int c2_adam_step(const c2_adam_tensor* desc_host, int n_tensors, double lr, double beta1, double beta2, double eps,
                 double weight_decay, double clip_value, int64_t step_t, void* stream);

/* The same step with its state ON THE DEVICE, for a training step replayed as a HIP graph (a graph bakes every by-value
 * argument in; torch's capturable optimizers keep `step` on the device for the same reason).  No counterpart in the
 * reference.  c2_adam_prepare: *state_dev.step += 1, then the kernel constants (bias corrections formed in double, as
 * c2_adam_step forms them on the host) into hyper_dev (c2_adam_hyper_bytes() bytes of device memory);
 * c2_adam_step_dev: the update of c2_adam_step with the constants read from hyper_dev.  lr may be rewritten between
 * steps by the host (learning-rate decay): a plain device write to state_dev->lr. */
typedef struct {
    int64_t step; /* steps taken so far (0 before the first) */
    double lr, beta1, beta2, eps, weight_decay, clip_value;
} c2_adam_state;
size_t c2_adam_hyper_bytes(void);
int c2_adam_prepare(c2_adam_state* state_dev, void* hyper_dev, void* stream);
int c2_adam_step_dev(const c2_adam_tensor* desc_host, int n_tensors, const void* hyper_dev, void* stream);

/* dst[c][r] = src[r][c], bf16 (or any 2-byte type): the narrow operand of c2_linear_wgrad_bt / c2_linear_wgrad_rows_at. */
int c2_transpose_bf16(const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int64_t ld_dst, void* stream);

/* fp32 -> bf16 copy (initial weight shadows), n elements; c2_cast_f16: fp32 -> IEEE half. */
int c2_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int c2_cast_f16(const float* src, void* dst, int64_t n, void* stream);

/* Device-resident dataset -> batch (replaces AnnotatedDataset.__getitem__ + custom_collate + .to(device),
 * dataset/celligner2/anndata.py:89-105, trainers/celligner2/_utils.py:188-225, trainer.py:274-275):
 * rows idx[0..n) of x/goodloc/sizefactor/cond/classes are gathered into dense batch tensors; xin
 * (= log(1+x) or x, dtype xin_dtype) is produced in the same pass (may be NULL).  x_out and goodloc_out may be
 * NULL when the consumer reads the resident rows itself (c2_recon_nb_fwd_bwd's row_index).  rowconst (may be NULL):
 * a second per-row fp32 scalar gathered like sizefactor (the dataset's c2_nb_row_const vector).  idx32_out (may be
 * NULL): idx as int32 [idx32_len >= n], entries past n filled with the last index -- the row index of
 * c2_linear_fwd_rows / c2_linear_wgrad_rows; with x_out, goodloc_out and xin_out all NULL nothing per gene is moved.
 * center (may be NULL): subtracted from xin per gene (see c2_col_mean).  xin2_out (may be NULL): a second copy of xin,
 * always bf16 [n, ldxin2] -- when xin is C2_F16 for the first layer's forward, its weight gradient (whose other operand,
 * dh, is bf16) reads this one. */
int c2_gather_rows(const float* X, const uint8_t* goodloc, const float* sizefactor, const int64_t* cond,
                   const int64_t* classes, int64_t G, int C, int P, const int64_t* idx, int64_t n, float* x_out,
                   uint8_t* goodloc_out, float* sf_out, int64_t* cond_out, int64_t* classes_out, void* xin_out,
                   int xin_dtype, int64_t ldxin, int apply_log, const float* rowconst, float* rowconst_out,
                   int32_t* idx32_out, int64_t idx32_len, const float* center, void* xin2_out, int64_t ldxin2,
                   void* stream);
/* c2_gather_rows with row pitches (elements) for the resident matrices and the gathered copies: X [rows, ldX], goodloc
 * [rows, ldg], x_out [n, ldx_out], goodloc_out [n, ldg_out].  A dataset whose gene count is not a multiple of 4 keeps its rows
 * G4 = G rounded up to 4 elements apart, pad columns zero (counts 0, mask 0): rows then start on 16 / 4 bytes and the
 * likelihood kernels walk them in 16-byte chunks.  When every row involved is at least G4 long the gather walks G4 columns
 * too and copies the sources' pad columns with the rest (so zero pads stay zero pads); center keeps G entries. */
int c2_gather_rows_ld(const float* X, int64_t ldX, const uint8_t* goodloc, int64_t ldg, const float* sizefactor,
                      const int64_t* cond, const int64_t* classes, int64_t G, int C, int P, const int64_t* idx, int64_t n,
                      float* x_out, int64_t ldx_out, uint8_t* goodloc_out, int64_t ldg_out, float* sf_out,
                      int64_t* cond_out, int64_t* classes_out, void* xin_out, int xin_dtype, int64_t ldxin, int apply_log,
                      const float* rowconst, float* rowconst_out, int32_t* idx32_out, int64_t idx32_len,
                      const float* center, void* xin2_out, int64_t ldxin2, void* stream);

/* ---- BatchNorm1d form of the layer tail (use_bn=True: modules.py:298-302, 405-413, 445-449, 557-566, 596-600;
 * nn.BatchNorm1d(affine=True), eps 1e-5, momentum 0.1).  The statistics run over the batch, so the tail is cut at
 * the points where a multi-GPU run exchanges them (SyncBN). ---- */

/* floats of workspace for c2_bn_stats / c2_bn_bwd_reduce. */
size_t c2_bn_ws(int64_t B, int64_t H);

/* h_out[B,H] = sum_s Y[s] + bias + sum_j cond[b][j]*Wc[:,j] (as c2_epilogue_fwd) and, per column, the mean and the
 * sum of squared deviations M2 of the B local rows: part_out[0][H] = mean, part_out[1][H] = M2. */
int c2_bn_stats(const float* Y, int64_t ldy, int nslab, int64_t slab_stride, const float* bias, const int64_t* cond,
                int C, const float* Wc, int64_t ldwc, int64_t B, int64_t H, float* h_out, float* part_out, float* ws,
                void* stream);

/* F.batch_norm + ReLU + dropout on h (overwritten by yhat = (h-mean)*rstd, kept for the backward).
 * nparts > 0: batch statistics = Chan combination of `nparts` partials [nparts][2][H] (one per rank; partial i covers
 *   rows [i*rows_per_part, min((i+1)*rows_per_part, total_rows))); update_running: running_mean / running_var
 *   (unbiased variance) move by `momentum` and *num_batches_tracked += 1 (torch.nn.BatchNorm1d in train mode).
 * nparts == 0: the running statistics normalise (eval mode).
 * stats_out[0][H] = mean used, stats_out[1][H] = 1/sqrt(var+eps).  act = dropout(relu(yhat*gamma+beta)). */
int c2_bn_fwd(float* h, const float* parts, int nparts, int64_t rows_per_part, int64_t total_rows, const float* gamma,
              const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
              int update_running, float eps, const uint8_t* mask, float p_drop, int64_t B, int64_t H, void* act,
              int act_dtype, int64_t ldact, float* stats_out, void* act_planes, void* stream);

/* Backward, first half: dy[B,H] = (sum_s dA[s] + extra_scale*extra) through dropout and ReLU (fp32), and the column
 * sums over the local rows sums_out[0][H] = sum dy, sums_out[1][H] = sum dy*yhat (= dbeta, dgamma, also written to
 * those pointers when not NULL). */
int c2_bn_bwd_reduce(const float* dA, int64_t ldda, int nslab, int64_t slab_stride, const float* extra,
                     int64_t ldextra, const float* extra_scale_dev, const uint8_t* mask, float p_drop,
                     const float* yhat, const float* gamma, const float* beta, int64_t B, int64_t H, float* dy_out,
                     float* sums_out, float* dgamma, float* dbeta, float* ws, void* stream);

/* Backward, second half: dh = gamma*rstd*(dy - sums[0]/n - yhat*sums[1]/n), n = total_rows (the sums all-reduced
 * over ranks first); batch_stats == 0 (eval-mode statistics): dh = gamma*rstd*dy. */
int c2_bn_bwd(const float* dy, const float* yhat, const float* sums, int64_t total_rows, int batch_stats,
              const float* gamma, const float* stats, int64_t B, int64_t H, void* dh, int dh_dtype, int64_t lddh,
              void* dh_planes, void* stream);

/* Dropout keep-masks (nn.Dropout of modules.py:309-312,426-431,458-461,579-582,607-610) for all layers of a step in
 * one launch: out[i] = 1 with probability 1-p (Philox-4x32-10 keyed by seed, counter offset + i/16).  The stream of
 * random numbers differs from torch's generator by construction; parity tests inject recorded masks instead. */
int c2_dropout_masks(uint8_t* out, int64_t n, float p_drop, uint64_t seed, uint64_t offset, void* stream);

/* The step's loss vector in one launch (semisupervised.py:63-66 `loss = recon + alpha*kl + mmd + ce` and
 * :69 `unweighted_loss`): out6 = (recon, kl, mmd, ce, w0*recon + w1*kl + w2*mmd + w3*ce, recon + kl + mmd). */
int c2_combine_losses(const float* losses4, float w0, float w1, float w2, float w3, float* out6, void* stream);

/* c2_defer_loss_sums(1): from now on the loss entry points called by THIS thread (c2_recon_*_fwd_bwd, c2_reparam_kl_fwd,
 * c2_classifier_loss_fwd_bwd, c2_mmd_fwd_bwd) leave their partial sums in their workspaces instead of launching the
 * one-block kernel that adds them to *loss_out, and the next c2_combine_losses / c2_combine_losses_dev of this thread sums
 * them in its own launch: the four slots of losses4 are then taken as zero on entry and receive the sums addressed to
 * them (no zero-fill needed), a sum addressed elsewhere is added to its target; sums are formed exactly as the separate
 * launches form them (bit-identical).  At most 8 sums wait; further ones are launched at once.  c2_defer_loss_sums(0)
 * ends the mode (sums already waiting stay until the next combine).  Returns the previous setting.  No counterpart in
 * the reference (semisupervised.py:63-69 sums with torch ops): it removes five ~4.5 us graph nodes from a replayed step. */
int c2_defer_loss_sums(int on);
/* Forget the sums that wait on this thread: for a caller whose forward failed between a loss entry point and the combine
 * call (their workspaces may be gone by the next combine). */
int c2_drop_pending_sums(void);

/* Its backward seed: out4[i] = w_i * g (g = *g_dev, or 1 when g_dev is NULL) -- the upstream gradients of the four
 * losses when the weighted total receives gradient g. */
int c2_scale4(const float* g_dev, float w0, float w1, float w2, float w3, float* out4, void* stream);

/* c2_combine_losses / c2_scale4 with the four weights in device memory (w_dev[4]): the KL weight changes per epoch
 * (trainer.py:231-253) and a replayed HIP graph cannot take it by value.  accum6 (may be NULL): += out6 in double, the
 * running sums of an epoch's loss logs (trainer.py:326-328 takes their float64 means) without a host read per step. */
int c2_combine_losses_dev(const float* losses4, const float* w_dev, float* out6, double* accum6, void* stream);
int c2_scale4_dev(const float* g_dev, const float* w_dev, float* out4, void* stream);
/* c2_scale4_dev and the first two uses of its out4[0] (the upstream gradient of the reconstruction term) in one launch:
 * y1[0..n1) = out4[0] * x1, y2[0..n2) = out4[0] * x2 -- the output layer's bias gradient and the dispersion gradient from
 * the unit-upstream sums c2_recon_nb_fwd_bwd_dbias leaves (autograd of losses.py:27-65 scales them the same way).  Either
 * vector may be absent (n = 0). */
int c2_scale4_dev_scale2(const float* g_dev, const float* w_dev, float* out4, const float* x1, float* y1, int64_t n1,
                         const float* x2, float* y2, int64_t n2, void* stream);

/* c2_dropout_masks with the Philox counter offset in device memory: masks from *offset_dev + lead, then
 * *offset_dev += advance (a second, one-thread launch), so that every replay of a captured step draws fresh masks.
 * lead: this rank's stretch of a step's counter range in a data-parallel run (rank x blocks per rank; advance is then
 * world x blocks per rank, and every rank's counter holds the same value). */
int c2_dropout_masks_dev(uint8_t* out, int64_t n, float p_drop, uint64_t seed, uint64_t* offset_dev, uint64_t lead,
                         uint64_t advance, void* stream);

/* All the noise of one training step in one launch, for steps replayed as a HIP graph (no counterpart in the reference, whose
 * noise comes from torch's generator: nn.Dropout in modules.py:304-306, 448-450 and celligner2.py:248 `eps = torch.randn_like`):
 * the dropout keep-masks exactly as c2_dropout_masks_dev draws them (n_mask bytes from counter value state[0] + lead), the
 * standard normals of the reparameterisation (eps[0 .. n_eps) = elements eps_first .. of the GLOBAL batch's [rows, latent]
 * matrix: element e is Box-Muller word pair e % 4 of the Philox block at counter state[0] + e / 4 under a second key, so
 * a rank's rows equal the rows one process draws), and the counter's advance, done by the block that is last to
 * read the counter (state[0] += advance; state[1], the count of blocks that have read it, returns to 0).  state: two 64-bit words in device memory,
 * state[1] zero before the first call.  Either part may be empty (n_mask 0 / n_eps 0). */
int c2_noise_dev(uint8_t* masks, int64_t n_mask, float p_drop, float* eps, int64_t n_eps, uint64_t eps_first, uint64_t seed,
                 uint64_t* state, uint64_t lead, uint64_t advance, void* stream);

/* Launch an instantiated HIP graph (hipGraphExec_t) on a stream: hipGraphLaunch.  For hosts that captured a step through
 * another runtime's wrapper and replay it themselves (no counterpart in the reference). */
int c2_graph_launch(void* graph_exec, void* stream);

/* Output activation of the decoder on the inference path -- Decoder.forward (celligner2/model/modules.py:467-482) as
 * Celligner2.reconstructLatent returns it (celligner2/model/celligner2.py:411-420): mode 0 = nn.Softmax(dim=-1) over the
 * genes (NB / ZINB mean head), mode 1 = nn.ReLU (MSE head).  logits [B,G] fp32 (row stride ldl) -> out [B,G] fp32. */
int c2_decoder_output(const float* logits, int64_t ldl, int64_t B, int64_t G, int mode, float* out, int64_t ldo,
                      void* stream);

/* y[i] = a*x[i] (a = *a_dev), fp32. */
int c2_scale(const float* x, const float* a_dev, float* y, int64_t n, void* stream);

/* y[i] = a*x[i] + y[i] (a = *a_dev), fp32; used to merge loss-branch gradients. */
int c2_axpy(const float* x, const float* a_dev, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* C2MI_H */
