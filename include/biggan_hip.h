/* biggan_hip.h - C ABI of libbiggan_hip.so: the MI355X (gfx950) kernels behind the BigGAN
 * training-step hot path of david-jk/BigGAN-Tensorflow.
 *
 * The reference has no FFI: its operator layer (ops.py, DiffAugment_tf.py) lowers to stock
 * TensorFlow ops.  Each entry point below names the reference call site whose arithmetic it
 * replaces (file:line relative to the reference repository).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensor storage); the library
 *     never allocates, frees or retains memory and never synchronises the device;
 *   - every call is asynchronous on the caller-supplied HIP stream (void* = hipStream_t) and is
 *     safe to capture into a hipGraph;
 *   - tensors are NHWC fp32, conv kernels HWIO [k,k,Cin,Cout], transposed-conv kernels
 *     [k,k,Cout,Cin] (the reference's variable layouts, ops.py:88,127);
 *   - return value 0 = success; non-zero = error, message in bg_last_error() (thread-local);
 *   - scratch memory is passed in by the caller, sized by the matching *_workspace_bytes().
 */
#ifndef BIGGAN_HIP_H
#define BIGGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_ABI_VERSION 3

enum { BG_OK = 0, BG_ERR_ARG = 1, BG_ERR_LAUNCH = 2, BG_ERR_UNSUPPORTED = 3 };
enum { BG_PAD_REFLECT = 0, BG_PAD_ZERO = 1 };
/* element types of activation tensors (the "_t" entry points and the dtype fields of BgConvDesc) */
enum { BG_F32 = 0, BG_BF16 = 1 };
/* arithmetic of a GEMM-shaped launch, chosen PER CALL in its descriptor (no process-wide state):
 *   BG_COMPUTE_F32  : v_mfma_f32_32x32x2_f32, exact fp32 FMA chain (the reference's precision, ops.py:14)
 *   BG_COMPUTE_BF16 : bf16 MFMA with fp32 accumulation; fp32 operands are rounded to bf16 (RNE) while staged */
enum { BG_COMPUTE_F32 = 0, BG_COMPUTE_BF16 = 1 };

int         bg_abi_version(void);
const char* bg_last_error(void);
/* "gfx950" - the only code object in the library */
const char* bg_target_arch(void);

/* Host-side helper of the input pipeline (utils.py:12-38 decode_png): reverse the PNG row filters of an 8-bit
 * image on the CPU.  raw: h rows of 1 filter byte + stride bytes (the inflated IDAT stream); out: h * stride. */
int bg_png_unfilter(const unsigned char* raw, int h, int stride, int bpp, unsigned char* out);

/* --------------------------------------------------------------------------------------------
 * Convolution geometry shared by conv / transposed conv (ops.py:49-139).
 *   conv   : x[N,H,W,Cin]  -> y[N,Ho,Wo,Cout],  Ho = (H + pad_lo + pad_hi - k)/stride + 1
 *   deconv : x[N,H,W,Cin]  -> y[N,Ho,Wo,Cout],  Ho = stride*H ('SAME'), pad_lo = TF's low crop
 * ------------------------------------------------------------------------------------------ */
typedef struct BgConvDesc {
    int32_t N, H, W, Cin;      /* input  */
    int32_t Ho, Wo, Cout;      /* output */
    int32_t k, stride;
    int32_t pad_lo;            /* low padding (conv) / low crop (deconv)            */
    int32_t pad_mode;          /* BG_PAD_REFLECT (tf.pad REFLECT + VALID, ops.py:82) or BG_PAD_ZERO */
    int32_t compute;           /* BG_COMPUTE_F32 / BG_COMPUTE_BF16 (fp32 tensors)                  */
    int32_t x_dtype;           /* element type of the layer's INPUT-side tensor  (x, dx): BG_F32 / BG_BF16 */
    int32_t y_dtype;           /* element type of the layer's OUTPUT-side tensor (y, dy)                    */
    int32_t w_packed;          /* 1: w is a bf16 K-contiguous packed copy (see below), 0: the fp32 variable */
} BgConvDesc;

/* The bf16-RESIDENT data path (BASELINE configs 3-5).  When the tensor a launch GATHERS from (x for the forward
 * and the weight gradient, dy for the input gradient) is BG_BF16, the launch runs on the global_load_lds kernels
 * of csrc/igemm16.hip: tiles go HBM -> LDS without passing through registers, so the weights must already be bf16
 * with the reduction dimension contiguous (w_packed = 1):
 *     forward of conv  : [k*k][Cout][Cin]      input gradient of conv  : [k*k][Cin][Cout]
 *     forward of deconv: [k*k][Cout][Cin]      input gradient of deconv: [k*k][Cin][Cout]
 * i.e. for each op one of the two copies bg_spectral_norm_batch_fwd writes next to w / sigma (BgSnItem::pack_p =
 * the variable's own order, pack_t = the two inner axes swapped).  Channel counts must be multiples of 8.  The
 * other tensor of the call may be bf16 or fp32 (its dtype field); weight gradients are always fp32.  With
 * BG_PAD_REFLECT the input gradient is computed on the padded grid and folded back (scratch from ws). */

/* Every MFMA GEMM entry point takes caller-provided scratch (ws, ws_bytes) sized by the matching
 * *_workspace_bytes(): it holds split-K partial slabs for layers whose output is too small to fill
 * the chip (4x4 / 8x8 feature maps, weight gradients).  ws may be NULL: the kernel then runs
 * un-split (same result up to fp32 summation order). */

/* tf.nn.conv2d (+ reflect tf.pad, + bias_add)                       ops.py:82,94-98
 *   y = alpha * conv(x, w) [+ bias] [+ y if accumulate]; alpha_dev (nullable) is a device scalar.
 *   x, y: fp32, or the dtypes named by d->x_dtype / d->y_dtype; w: fp32 [k,k,Cin,Cout] or the packed bf16 copy. */
size_t bg_conv2d_fwd_workspace_bytes(const BgConvDesc*);
int bg_conv2d_fwd  (const BgConvDesc*, const void* x, const void* w, const float* bias,
                    const float* alpha_dev, void* y, int accumulate,
                    void* ws, size_t ws_bytes, void* stream);
/* gradient of the above w.r.t. x (reflect padding folded back)      autodiff of ops.py:82,94
 *   dy has d->y_dtype, dx has d->x_dtype */
size_t bg_conv2d_dgrad_workspace_bytes(const BgConvDesc*);
int bg_conv2d_dgrad(const BgConvDesc*, const void* dy, const void* w, const float* alpha_dev,
                    void* dx, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* gradient w.r.t. w: dw[k,k,Cin,Cout], always fp32 */
size_t bg_conv2d_wgrad_workspace_bytes(const BgConvDesc*);
int bg_conv2d_wgrad(const BgConvDesc*, const void* x, const void* dy, float* dw,
                    void* ws, size_t ws_bytes, void* stream);

/* tf.nn.conv2d_transpose(SAME) (+ bias_add)                          ops.py:127-132 */
size_t bg_deconv2d_fwd_workspace_bytes(const BgConvDesc*);
int bg_deconv2d_fwd  (const BgConvDesc*, const void* x, const void* w, const float* bias,
                      const float* alpha_dev, void* y, int accumulate,
                      void* ws, size_t ws_bytes, void* stream);
/* Transposed-conv forward with the batch-norm statistics of its OUTPUT fused into the epilogue (ops.py:630
 * tf.nn.moments of the tensor the following condition_batch_norm / batch_norm normalises): sums[0..Cout) = sum of the
 * stored (bf16-rounded) outputs per channel over (N, Ho, Wo), sums[Cout..2 Cout) = sum of their squares, fp64, zeroed by
 * the call - what bg_bn_stats_t would compute by reading y again.  Available for the bf16-resident launches that take the
 * halo-tile form (bf16 y): bg_deconv2d_fwd_stats_workspace_bytes returns the bytes of per-block partial sums the call
 * needs (stats_ws), 0 when the launch has no fused statistics (callers then use bg_bn_stats_t).  Deterministic: one
 * partial row per block, summed in fp64. */
size_t bg_deconv2d_fwd_stats_workspace_bytes(const BgConvDesc*);
int bg_deconv2d_fwd_stats(const BgConvDesc*, const void* x, const void* w, const float* bias, const float* alpha_dev,
                          void* y, int accumulate, double* sums, void* stats_ws, size_t stats_ws_bytes, void* ws,
                          size_t ws_bytes, void* stream);
size_t bg_deconv2d_dgrad_workspace_bytes(const BgConvDesc*);
int bg_deconv2d_dgrad(const BgConvDesc*, const void* dy, const void* w, const float* alpha_dev,
                      void* dx, int accumulate, void* ws, size_t ws_bytes, void* stream);
size_t bg_deconv2d_wgrad_workspace_bytes(const BgConvDesc*);
int bg_deconv2d_wgrad(const BgConvDesc*, const void* x, const void* dy, float* dw,
                      void* ws, size_t ws_bytes, void* stream);

/* Direct (vector-ALU, HBM-bound) path for k x k stride-1 convolutions with <= 3 output channels:
 * the generator's RGB head G_logit (BigGAN.py:570 -> ops.py:49-113).  Same arithmetic as
 * bg_conv2d_*; bg_rgbconv_supported tells whether a geometry qualifies. */
int bg_rgbconv_supported(const BgConvDesc*);
int bg_rgbconv_fwd  (const BgConvDesc*, const float* x, const float* w, const float* bias, float* y,
                     int accumulate, void* stream);
int bg_rgbconv_dgrad(const BgConvDesc*, const float* dy, const float* w, float* dx, int accumulate, void* stream);
size_t bg_rgbconv_wgrad_workspace_bytes(const BgConvDesc*);
int bg_rgbconv_wgrad(const BgConvDesc*, const float* x, const float* dy, float* dw,
                     void* ws, size_t ws_bytes, void* stream);

/* --------------------------------------------------------------------------------------------
 * Plain (batched) matrix products: tf.matmul call sites ops.py:163-165 (dense), 481,485
 * (attention), utils.py:198,222 (Gram matrix of the regulariser).
 *   C[b] = alpha * op(A[b]) * op(B[b]) (+ bias[n]) (+ C[b] if accumulate), row-major, fp32.
 *   transA: A stored [K,M] (lda = M-stride), transB: B stored [N,K].
 * ------------------------------------------------------------------------------------------ */
typedef struct BgGemmDesc {
    int32_t M, N, K;
    int32_t transA, transB;
    int32_t lda, ldb, ldc;
    int32_t batch;
    int64_t strideA, strideB, strideC;   /* elements between batch items */
    int32_t compute;                     /* BG_COMPUTE_F32 / BG_COMPUTE_BF16 */
    int32_t reserved;
} BgGemmDesc;
size_t bg_gemm_workspace_bytes(const BgGemmDesc*);
int bg_gemm(const BgGemmDesc*, const float* A, const float* B, const float* bias,
            const float* alpha_dev, float* C, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* Grouped small dense projections on shared batch rows: the beta / gamma fully_connected(z -> C) pairs of the two
 * conditional batch norms of a generator block (ops.py:623-624 -> ops.py:163-165) as ONE launch per direction.
 *   fwd  : y_i[b, :] = x_i[b, :K_i] w_i + bias_i             (x_i rows ldx_i floats apart, w_i [K_i, N_i], y_i [B, N_i])
 *   wgrad: dw_i (+)= x_i^T dy_i, db_i (+)= column sums of dy_i  (y = dy_i; acc_w / acc_b select add vs overwrite; db
 *          may be NULL).  The items array is HOST memory (copied into the launch), all other pointers device fp32.
 * No atomics: every output element is produced by one thread in a fixed batch order. */
#define BG_DENSE_GROUP_MAX 8
typedef struct BgDenseItem {
    const float* x;
    int64_t ldx;
    const float* w;        /* fwd */
    const float* bias;     /* fwd, nullable */
    float* y;              /* fwd: output; wgrad: dy (read) */
    float* dw;             /* wgrad */
    float* db;             /* wgrad, nullable */
    int32_t K, N;
    int32_t acc_w, acc_b;
} BgDenseItem;
int bg_dense_group_fwd(const BgDenseItem* items_host, int n_items, int B, void* stream);
int bg_dense_group_wgrad(const BgDenseItem* items_host, int n_items, int B, void* stream);

/* --------------------------------------------------------------------------------------------
 * Fused attention of self_attention_2 (ops.py:481-485): o = softmax(q k^T) v, no 1/sqrt(d) scale.
 *   q [B,N,d], k [B,Nk,d], v [B,Nk,dv], o [B,N,dv], lse [B,N] (= max + log sum exp of each query's
 *   logits, kept for backward).  The [N,Nk] logits/probabilities are never written to memory; backward
 *   recomputes them (delta_ws: B*N floats of scratch).  fp32 MFMA, deterministic (no atomics).
 *   Supported when N % 128 == 0, Nk % 128 == 0, d % 4 == 0, dv % 4 == 0, d <= 32, dv <= 128
 *   (bg_attention2_supported); other shapes use bg_gemm + bg_softmax_* (materialised form).
 * ------------------------------------------------------------------------------------------ */
int bg_attention2_supported(int N, int Nk, int d, int dv);
int bg_attention2_fwd(const float* q, const float* k, const float* v, float* o, float* lse,
                      int B, int N, int Nk, int d, int dv, void* stream);
int bg_attention2_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                      const float* lse, float* dq, float* dk, float* dv_out, float* delta_ws,
                      int B, int N, int Nk, int d, int dv, void* stream);

/* --------------------------------------------------------------------------------------------
 * Fused attention on the bf16 MFMA for the bf16-resident path (csrc/attention16.hip): same function as
 * bg_attention2_* (ops.py:481-485, o = softmax(q k^T) v, no scale) with bf16 tensors that may be COLUMN SLICES of wider
 * tensors (the fused f|g|h projection of self_attention_2 and its max-pooled copy): every tensor has a row stride ld*
 * and a batch stride s* in elements.  d <= 64, dv <= 256 (all BASELINE topologies; BigGAN.py:292-293 puts d = 48 / 64,
 * dv = 192 / 256 in the 256^2 / 512^2 generators), d and dv multiples of 4, N and Nk multiples of 128, pointers 8-byte
 * aligned.  lse [B,N] fp32 (contiguous) is written by fwd and read by bwd; delta_ws: B*N floats of scratch.
 * Probabilities are rounded to bf16 before the P V product (fp32 accumulation); deterministic, no atomics.
 * bwd: dq may be NULL, or dk and dv_out may both be NULL (the two gradient kernels are independent); delta_ws is
 * (re)computed by a call that produces dk / dv and reused as it stands by a dq-only call.
 * ------------------------------------------------------------------------------------------ */
typedef struct BgAttn16Desc {
    int32_t B, N, Nk, d, dv;
    int32_t reserved;
    int64_t ldq, sq, ldk, sk, ldv, sv, ldo, so;          /* q, k, v, o                                  */
    int64_t ldg, sg, lddq, sdq, lddk, sdk, lddv, sdv;    /* backward: dout (layout of o), dq, dk, dv    */
} BgAttn16Desc;
int bg_attention16_supported(int N, int Nk, int d, int dv);
int bg_attention16_fwd(const BgAttn16Desc*, const void* q, const void* k, const void* v, void* o, float* lse,
                       void* stream);
int bg_attention16_bwd(const BgAttn16Desc*, const void* q, const void* k, const void* v, const void* o,
                       const void* dout, const float* lse, void* dq, void* dk, void* dv_out, float* delta_ws,
                       void* stream);

/* --------------------------------------------------------------------------------------------
 * Spectral norm, one power iteration (ops.py:718-747).  W is [rows, cols] (= reshape(w,[-1,last])).
 *   v = l2n(u W^T), u_out = l2n(v W), sigma = |v W|, w_norm = W / sigma.
 *   l2n(t) = t * rsqrt(max(sum t^2, 1e-12)).  scratch: bg_spectral_norm_workspace_bytes (fp64 accumulators).
 * bwd: dW = (G - <G, w_norm> v^T u_out) / sigma  (u_out, v stop-gradient, ops.py:738-739).
 * ------------------------------------------------------------------------------------------ */
size_t bg_spectral_norm_workspace_bytes(int rows, int cols);
int bg_spectral_norm_fwd(const float* w, const float* u_in, float* u_out, float* v_out,
                         float* sigma_out, float* w_norm, int rows, int cols,
                         void* ws, size_t ws_bytes, void* stream);
int bg_spectral_norm_bwd(const float* g_wnorm, const float* w_norm, const float* u_hat,
                         const float* v_hat, const float* sigma, float* dw, int rows, int cols,
                         void* ws, size_t ws_bytes, void* stream);

/* Multi-tensor form: every spectrally-normalised weight of one network in 4 (fwd) / 2 (bwd) launches.
 * The BgSnItem table lives in device memory (pointers are stable: flat arenas); ws_offset is the byte
 * offset (16-byte aligned) of the item's scratch of bg_spectral_norm_workspace_bytes(rows, cols) bytes
 * inside ws; fwd zeroes ws[0, ws_bytes).  u is updated in place.  bwd: bit i of enable_mask /
 * accumulate_mask (host arrays of ceil(n/64) words; NULL = all / none) selects whether item i is
 * processed and whether dw is added to instead of overwritten; ws >= 8 * n_items bytes. */
typedef struct BgSnItem {
    const float* w;        /* [rows, cols] */
    float* u;              /* [cols], in/out */
    float* v;              /* [rows], out */
    float* sigma;          /* [1], out */
    float* w_norm;         /* [rows, cols], out */
    const float* g_wnorm;  /* bwd in: dL/d(w_norm) */
    float* dw;             /* bwd out: dL/dw */
    int64_t ws_offset;
    int32_t rows, cols;
    /* bf16-resident path (optional, NULL otherwise): the weight seen as [taps][rows / taps][cols];
     * pack_p[tap][r][c] = pack_t[tap][c][r] = bf16(w[tap][r][c] / sigma).  rows / taps and cols must be even. */
    void* pack_p;
    void* pack_t;
    int32_t taps;
    int32_t pack_p_ld;     /* elements between rows of pack_p (0 = cols): kernels that share one GEMM (the f|g|h 1x1
                            * projections of self_attention_2) write column slices of a common [rows][sum cols] copy;
                            * their pack_t copies are row blocks of a common [sum cols][rows] matrix (plain offsets) */
} BgSnItem;
int bg_spectral_norm_batch_fwd(const BgSnItem* items_dev, int n_items, void* ws, size_t ws_bytes, void* stream);
/* The same in two halves, for data parallelism (last row: the batch-independent power iteration is
 * SHARDED by weight instead of repeated on every rank):
 *   BG_SN_POWER      the two GEMV passes + finalisation of the items given (a rank's OWN weights: a sub-table): writes
 *                    u (in place), sigma and v_hat of those items - nothing else
 *   BG_SN_NORMALIZE  w / sigma (+ packed bf16 copies) of the items given (ALL weights), sigma read from each item's
 *                    sigma slot - filled for the other ranks' weights by an all-gather of sigma | u | v_hat in between
 *   BG_SN_ALL        both, = bg_spectral_norm_batch_fwd */
#define BG_SN_ALL 0
#define BG_SN_POWER 1
#define BG_SN_NORMALIZE 2
int bg_spectral_norm_batch_phase(const BgSnItem* items_dev, int n_items, void* ws, size_t ws_bytes, int phase,
                                 void* stream);
int bg_spectral_norm_batch_bwd(const BgSnItem* items_dev, int n_items, const uint64_t* enable_mask,
                               const uint64_t* accumulate_mask, void* ws, size_t ws_bytes, void* stream);

/* Forward-mode tangent of TRAINING-mode batch norm and its backward: the gradient penalty (BigGAN.py:717-742) through a
 * discriminator with --bn_in_d (ops.py:546-561).  x, xd, s are [rows, C] fp32.
 *   bg_chan_dots3            sums[0:C] = sum p, [C:2C] = sum p q, [2C:3C] = sum p r (r may be NULL) in fp64 (the caller
 *                            all-reduces them under data parallelism)
 *   bg_bn_tangent_fwd_coefs  from sums(p = xd, q = x): coefs[3C] with yd = coefs[0] xd + coefs[1] x + coefs[2] and m12[2C] =
 *                            (mean(xd) | mean(xd xhat)), kept for the backward
 *   bg_bn_tangent_bwd_coefs  from sums(p = s, q = x, r = xd), s = dL/dyd: coefs_dxd[3C] (d_xd = . s + . x + .), coefs_dx[4C]
 *                            (d_x = . s + . x + . xd + .) and dgamma[C]
 *   bg_chan_lincomb3         out = cp[c] p + cq[c] q (+ cr[c] r) + c0[c] */
int bg_chan_dots3(const float* p, const float* q, const float* r, double* sums, int64_t rows, int C, void* stream);
int bg_bn_tangent_fwd_coefs(const double* sums, double count, const float* mean, const float* rstd, const float* gamma,
                            float* coefs, float* m12, int C, void* stream);
int bg_bn_tangent_bwd_coefs(const double* sums, double count, const float* mean, const float* rstd, const float* gamma,
                            const float* m12, float* coefs_dxd, float* coefs_dx, float* dgamma, int C, void* stream);
int bg_chan_lincomb3(const float* p, const float* cp, const float* q, const float* cq, const float* r, const float* cr,
                     const float* c0, float* out, int64_t rows, int C, void* stream);

/* Gram matrix out[c1][c2] = sum_r a[r][c1] * a[r][c2] (fp32, [cols][cols]) of a bf16 row-major matrix a[rows][ld]:
 * the ortho-cosine regulariser's W^T W (utils.py:198) from the packed bf16 copy of w / sigma (BgSnItem.pack_p); the caller
 * rescales by sigma^2.  cols % 8 == ld % 8 == 0; ws from bg_gram16_workspace_bytes (split-K slabs). */
size_t bg_gram16_workspace_bytes(int rows, int cols);
int bg_gram16(const void* a, int rows, int cols, int ld, float* out, void* ws, size_t ws_bytes, void* stream);

/* --------------------------------------------------------------------------------------------
 * Batch statistics + (conditional) batch-norm + PReLU (ops.py:532-537, 580-585, 611-643).
 *   x [N,HW,C].  stats: sums[0:C] = sum x, sums[C:2C] = sum x^2 over N*HW (fp64 accumulators, so the
 *   result does not depend on the arrival order of the blocks' atomics; caller zeroes or
 *   all-reduces them for cross-replica BN), then bg_bn_finalize turns sums into mean / rstd
 *   (biased variance, eps) and updates the moving statistics.
 *   apply: y = act((x - mean) * rstd * gamma + beta), gamma/beta per sample [N,C] (per_sample=1,
 *   condition_batch_norm) or per channel [C] (tf.layers.batch_normalization);
 *   act = PReLU with per-channel alpha when alpha != NULL (ops.py:535-537), identity otherwise.
 * ------------------------------------------------------------------------------------------ */
int bg_bn_stats(const float* x, double* sums, int64_t rows, int C, void* stream);
int bg_bn_finalize(const double* sums, double count, float eps, float momentum, int unbiased_moving_var,
                   float* mean, float* rstd, float* moving_mean, float* moving_var, int C, void* stream);
/* inference (ops.py:640-643): mean = pop_mean, rstd = 1/sqrt(pop_var + eps) for the apply kernel below */
int bg_bn_population(const float* pop_mean, const float* pop_var, float eps, float* mean, float* rstd, int C,
                     void* stream);
int bg_bn_apply_act_fwd(const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, int per_sample,
                        const float* alpha, float* y, int N, int HW, int C, void* stream);
/* backward, pass 1: per-(sample,channel) reductions
 *   part[0][n][c] = sum_hw g, part[1][n][c] = sum_hw g * xhat, part[2][n][c] = sum_hw dy * min(pre,0)
 *   where pre = xhat*gamma+beta, g = dy * act'(pre).            part is [3,N,C] fp32 */
int bg_bn_apply_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int per_sample,
                               const float* alpha, float* part, int N, int HW, int C, void* stream);
/* backward, pass 2: dx = rstd * (g*gamma - m1 - xhat * m2), m1 = mean(g*gamma), m2 = mean(g*gamma*xhat)
 *   given as per-channel device vectors cm[0:C] = m1, cm[C:2C] = m2 (caller reduces part over N,
 *   and across ranks for cross-replica BN). */
int bg_bn_apply_act_bwd_dx(const float* x, const float* dy, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int per_sample,
                           const float* alpha, const float* cm, float* dx,
                           int N, int HW, int C, void* stream);
/* reduce part[3,N,C] -> dgamma/dbeta (per sample: copies; per channel: sums over N), dalpha[C],
 *   cm[2C] = (sum_n gamma_n * part0_n, sum_n gamma_n * part1_n) / count */
int bg_bn_bwd_finalize(const float* part, const float* gamma, int per_sample, double count,
                       float* dgamma, float* dbeta, float* dalpha, float* cm, int N, int C, void* stream);

/* Batch renormalisation (ops.py:600-609 tf.layers.batch_normalization(renorm=True); ops.py:645-715
 * condition_batch_renorm).  The normalisation itself is the batch-norm above with the affine pair replaced by
 *   gamma_eff = r * gamma ,  beta_eff = beta + d * gamma        (r, d per channel, constants for the gradient)
 * bg_renorm_coeffs turns the batch sums of bg_bn_stats into the clipped corrections, measured against running
 * statistics that are read BEFORE any update of this run:
 *   sigma = sqrt(var + eps) ; sigma_ref = sqrt(ref_scale + eps)            (scale_is_var = 1, ops.py:688-689)
 *                             sigma_ref = max(ref_scale, sqrt(eps))        (scale_is_var = 0, Keras renorm_stddev)
 *   sigma_w = w * sigma_ref + (1-w) * sigma ;  mean_w = w * ref_mean + (1-w) * mean     (ops.py:690-691;
 *             w = *weight, or 1 when weight == NULL: --bn_renorm_shared / Keras)
 *   r = clip(sigma / sigma_w, rmin, rmax) ;  d = clip((mean - mean_w) / sigma_w, -dmax, dmax)   (ops.py:692-693)
 * update = 1 also moves the running statistics (ops.py:701-703):
 *   ref_mean  <- ref_mean  * decay + mean * (1-decay)
 *   ref_scale <- ref_scale * decay + (scale_is_var ? var : sigma) * (1-decay)
 *   *weight   <- *weight * fadein_decay + (1 - fadein_decay)
 * bg_renorm_affine_fwd/bwd: the [rows, C] (conditional, rows = batch) or [1, C] affine pair and the gradient
 *   dgamma = r * dgamma_eff + d * dbeta_eff   (dbeta = dbeta_eff). */
int bg_renorm_coeffs(const double* sums, double count, float* ref_mean, float* ref_scale, int scale_is_var,
                     float* weight, float eps, float rmin, float rmax, float dmax, float decay, float fadein_decay,
                     int update, float* r, float* d, int C, void* stream);
int bg_renorm_affine_fwd(const float* gamma, const float* beta, const float* r, const float* d,
                         float* gamma_eff, float* beta_eff, int64_t rows, int C, void* stream);
int bg_renorm_affine_bwd(const float* dgamma_eff, const float* dbeta_eff, const float* r, const float* d,
                         float* dgamma, int64_t rows, int C, void* stream);

/* stand-alone PReLU (ops.py:532-537): y = x>0 ? x : alpha_c*x ; bwd returns dx and per-channel
 * partial dalpha accumulated with atomics into dalpha[C] (caller zeroes). */
int bg_prelu_fwd(const float* x, const float* alpha, float* y, int64_t rows, int C, void* stream);
int bg_prelu_bwd(const float* x, const float* dy, const float* alpha, float* dx, float* dalpha,
                 int64_t rows, int C, void* stream);

/* tf.layers.max_pooling2d(2,2,'SAME') on even H,W (ops.py:508-510); bwd routes to the first max */
int bg_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int bg_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* 2x2 stride-2 box kernels (x is [N,H,W,C] for both): bg_box2_down: y[N,H/2,W/2,C] = scale * sum of each 2x2
 * block (avg_pooling forward with scale 1/4, ops.py:512-514; up_sample backward with scale 1);
 * bg_box2_up: y[N,2H,2W,C] = scale * x replicated 2x2 (up_sample = resize_nearest_neighbor x2 forward with
 * scale 1, ops.py:516-519; avg_pooling backward with scale 1/4). */
int bg_box2_down(const float* x, float* y, int N, int H, int W, int C, float scale, void* stream);
int bg_box2_up(const float* x, float* y, int N, int H, int W, int C, float scale, void* stream);

/* tf.nn.softmax over the last axis (ops.py:483): rows x cols, in place allowed; bwd: ds = p*(dp - sum(dp*p)) */
int bg_softmax_fwd(const float* s, float* p, int64_t rows, int cols, void* stream);
int bg_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int cols, void* stream);

/* tf.reduce_sum(x,[1,2]) (ops.py:503-506) and its gradient (broadcast) */
int bg_sum_pool_fwd(const float* x, float* y, int N, int HW, int C, void* stream);
int bg_sum_pool_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream);

/* y = a*x + b*y elementwise family used by residual adds, gamma*o + x (ops.py:490), tanh (ops.py:539) */
int bg_axpby(const float* x, float a, float* y, float b, int64_t n, void* stream);
int bg_add(const float* a, const float* b, float* y, int64_t n, void* stream);            /* y = a + b */
int bg_scale_add(const float* o, const float* gamma_dev, const float* x, float* y, int64_t n, void* stream);
int bg_dot(const float* a, const float* b, float* out_accum, int64_t n, void* stream);   /* out += <a,b> */
int bg_scale_dev(const float* x, const float* s_dev, float* y, int64_t n, void* stream);  /* y = s*x, s a device scalar */
int bg_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int bg_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
int bg_bias_grad(const float* dy, float* db, int64_t rows, int C, void* stream);         /* db[c] = sum_rows dy */

/* --------------------------------------------------------------------------------------------
 * bf16-resident data path (BASELINE configs 3-5): "_t" forms of the bandwidth-bound kernels above.  Activation
 * tensors are fp32 or bf16 in HBM (dtype arguments: BG_F32 / BG_BF16), arithmetic is fp32 in registers, parameters,
 * statistics and reductions stay fp32 / fp64.  x_dtype names the op's INPUT-side tensors (x, dx), y_dtype its
 * OUTPUT-side tensors (y, dy).  C % 4 == 0 (8-byte bf16 accesses).  Same formulas and reference call sites as the
 * fp32 entry points of the same name.
 * ------------------------------------------------------------------------------------------ */
int bg_cast(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, void* stream);
/* Widen or narrow the middle dimension of an [outer][C][inner] view from Cs to Cd entries, converting between fp32 and
 * bf16: the 3-channel image layers (ops.py:49 with Cin = 3, BigGAN.py:570 with Cout = 3) run on the bf16-resident GEMMs
 * with the thin side widened to 8 channels.  mode BG_PAD_ZERO_FILL (0): dst[c] = c < Cs ? src[c] : 0;
 * BG_PAD_SPLIT (1, Cd >= 2 Cs): dst[c] = bf16(src[c]), dst[Cs + c] = src[c] - bf16(src[c]) - the idle padding channels
 * carry the rounding residual, so the MFMA sees the thin operand to ~16 mantissa bits; BG_PAD_DUP (2): dst[c] =
 * dst[Cs + c] = src[c] (the partner operand of a split one); BG_PAD_FOLD (3, Cs >= 2 Cd): dst[c] = src[c] + src[Cd + c]. */
enum { BG_PAD_ZERO_FILL = 0, BG_PAD_SPLIT = 1, BG_PAD_DUP = 2, BG_PAD_FOLD = 3 };
int bg_pad_channels(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t outer, int Cs, int Cd,
                    int64_t inner, int mode, void* stream);
/* the two packed bf16 copies of a conv kernel that is NOT spectrally normalised (see BgSnItem::pack_p / pack_t) */
int bg_weight_pack(const float* w, int taps, int rows_per_tap, int cols, void* pack_p, void* pack_t, void* stream);
int bg_bn_stats_t(const void* x, int x_dtype, double* sums, int64_t rows, int C, void* stream);
int bg_bn_apply_act_fwd_t(const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int per_sample, const float* alpha, void* y, int y_dtype,
                          int N, int HW, int C, void* stream);
int bg_bn_apply_act_bwd_reduce_t(const void* x, int x_dtype, const void* dy, int y_dtype, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int per_sample,
                                 const float* alpha, float* part, int N, int HW, int C, void* stream);
int bg_bn_apply_act_bwd_dx_t(const void* x, int x_dtype, const void* dy, int y_dtype, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, int per_sample,
                             const float* alpha, const float* cm, void* dx, const void* dx_add, int N, int HW, int C,
                             void* stream);      /* dx_add (x_dtype, nullable, may alias dx): dx = dx_add + gradient - the
                                                    sum of the two branch gradients of a forked tensor, fused */
int bg_prelu_fwd_t(const void* x, int x_dtype, const float* alpha, void* y, int y_dtype, int64_t rows, int C,
                   void* stream);
/* dx (x_dtype, nullable) and / or dalpha (fp32 [C], accumulated: caller zeroes; nullable); dx_add as above */
int bg_prelu_bwd_t(const void* x, int x_dtype, const void* dy, int y_dtype, const float* alpha, void* dx,
                   float* dalpha, const void* dx_add, int64_t rows, int C, void* stream);
int bg_bias_grad_t(const void* dy, int dtype, float* db, int64_t rows, int C, void* stream);
int bg_maxpool2_fwd_t(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream);
int bg_maxpool2_bwd_t(const void* x, const void* dy, void* dx, int dtype, int N, int H, int W, int C, void* stream);
int bg_sum_pool_fwd_t(const void* x, int x_dtype, float* y, int N, int HW, int C, void* stream);
int bg_sum_pool_bwd_t(const float* dy, void* dx, int x_dtype, int N, int HW, int C, void* stream);
/* y = s * a + sb * b with s = *sa_dev (device scalar) when sa_dev != NULL, else sa; b may be NULL (y = s * a):
 * residual adds, gamma * o + x (ops.py:490), scaling by a device scalar.  n % 4 == 0. */
int bg_lincomb_t(const void* a, const float* sa_dev, float sa, const void* b, float sb, void* y, int dtype,
                 int64_t n, void* stream);
int bg_dot_t(const void* a, const void* b, int dtype, float* out_accum, int64_t n, void* stream);  /* out += <a,b> */

/* --------------------------------------------------------------------------------------------
 * DiffAugment 'color,translation,cutout' (DiffAugment_tf.py:8-73), x [N,S,S,C] fp32.
 *   policy bit 0 = color, 1 = translation, 2 = cutout.  Draws are explicit device arrays:
 *   u_b,u_s,u_c float[N] in [0,1); t_x,t_y int32[N] in [-shift,shift]; o_x,o_y int32[N].
 *   mean_ws: N doubles scratch (per-sample sums for rand_contrast).  Integer index math is
 *   bit-exact with DiffAugment_tf.py:40-66.
 * ------------------------------------------------------------------------------------------ */
int bg_diffaugment_fwd(const float* x, float* y, const float* u_b, const float* u_s, const float* u_c,
                       const int32_t* t_x, const int32_t* t_y, const int32_t* o_x, const int32_t* o_y,
                       int N, int S, int C, int policy, double* mean_ws, void* stream);
int bg_diffaugment_bwd(const float* dy, float* dx, const float* u_s, const float* u_c,
                       const int32_t* t_x, const int32_t* t_y, const int32_t* o_x, const int32_t* o_y,
                       int N, int S, int C, int policy, double* mean_ws, void* stream);

/* --------------------------------------------------------------------------------------------
 * Hinge losses with flood (ops.py:788-797, 832-840, 847-848).
 *   sums: device float[2] accumulators (caller zeroes; all-reduced across ranks for DP):
 *     D: sums[0] += sum relu(1-real), sums[1] += sum relu(1+fake);   G: sums[0] += sum fake
 *   grad kernels read the (global) sums, n_global = global batch, and write the flooded loss
 *   (loss_out[0]) and d(loss)/d(logit).
 * ------------------------------------------------------------------------------------------ */
int bg_hinge_d_sums(const float* real, const float* fake, float* sums, int n, void* stream);
int bg_hinge_d_grad(const float* real, const float* fake, const float* sums, double n_global, float flood,
                    float* d_real, float* d_fake, float* loss_out, int n, void* stream);
int bg_hinge_g_sums(const float* fake, float* sums, int n, void* stream);
int bg_hinge_g_grad(const float* sums, double n_global, float flood, float* d_fake, float* loss_out,
                    int n, void* stream);

/* General GAN losses (ops.py:753-840): kind 0 hinge, 1 lsgan, 2 gan (also 'dragan'), 3 ra-lsgan, 4 ra-gan (also
 * 'ra-dragan'), 5 ra-hinge, 6 wgan (wgan-gp / wgan-lp); generator = 0 for discriminator_loss, 1 for generator_loss.  Three steps so that data-parallel
 * ranks can exchange the sums in between:  bg_gan_loss_means -> sums = {sum real, sum fake};
 * bg_gan_loss_terms (given the GLOBAL sums / counts) -> tsums = {sum phi_r, sum phi_f, sum phi_r', sum phi_f'};
 * bg_gan_loss_grad (given the GLOBAL tsums) -> loss (flooded), d_real, d_fake (incl. the coupling through the
 * batch means of the relativistic kinds).  real may be NULL with nr = 0 (non-relativistic generator losses). */
int bg_gan_loss_means(const float* real, const float* fake, float* sums, int nr, int nf, void* stream);
int bg_gan_loss_terms(int kind, int generator, const float* real, const float* fake, const float* sums,
                      double n_real_global, double n_fake_global, float* tsums, int nr, int nf, void* stream);
int bg_gan_loss_grad(int kind, int generator, const float* real, const float* fake, const float* sums,
                     const float* tsums, double n_real_global, double n_fake_global, float flood, float* d_real,
                     float* d_fake, float* loss_out, int nr, int nf, void* stream);

/* --------------------------------------------------------------------------------------------
 * Gradient penalty (BigGAN.py:717-742) and the second-order pieces it needs.
 *   GP = ld * mean_n phi(||g_n||),  g = d sum(D(aug(x^)))/d x^,  phi = (n-1)^2 (wgan-gp, dragan) or max(0,n-1)^2 (wgan-lp).
 *   Its parameter gradient is taken as the gradient of the directional derivative of D along the constant
 *   direction v_n = ld * phi'(||g_n||) / count * g_n / ||g_n||  (d GP/d theta = d <g(theta), v>/d theta), i.e. by
 *   differentiating a forward-mode (tangent) pass of the discriminator; the entries below are the tangent maps that
 *   are not already linear kernels of this library, and their derivatives.
 * bg_gp_interpolate: x^ = real + alpha_n (other - real)           (sums == NULL; other = fake: BigGAN.py:726-727)
 *                    x^ = real + alpha_n * 0.5 * std(real) * other (sums = {sum real, sum real^2} in fp64 from
 *                         bg_bn_stats with C = 1, count = numel: BigGAN.py:719-724, other = eps ~ U[0,1))
 * bg_gp_penalty: g [N, per] -> *loss = ld * sum_n phi(||g_n||) / count_global (this rank's share of the mean),
 *                v = the direction above; ws: 2*N doubles of scratch; lp = 1 selects wgan-lp.
 * bg_maxpool2_gather: tangent of the 2x2 max pool, y = t at the first maximum of x's window (its transpose is
 *                bg_maxpool2_bwd).
 * bg_prelu_tangent_dalpha: the PReLU tangent is ydot = xdot * prelu'(x) (computed by bg_prelu_bwd with dy := xdot);
 *                its derivative w.r.t. alpha is dalpha[c] = sum_rows dy * xdot * [x < 0]  (1/2 at x == 0).
 * bg_softmax_tangent_bwd: the softmax tangent is pdot = p * (sdot - sum p sdot) (computed by bg_softmax_bwd with
 *                dp := sdot); given g = dL/dpdot:  dsdot = p * (g - u), dp = g * (sdot - t) - u * sdot,
 *                t = sum p sdot, u = sum g p (row sums).
 * ------------------------------------------------------------------------------------------ */
int bg_gp_interpolate(const float* real, const float* other, const float* alpha, const double* sums, double count,
                      float* out, int N, int64_t per, void* stream);
int bg_gp_penalty(const float* g, int N, int64_t per, double count_global, float ld, int lp, double* ws, float* loss,
                  float* v, void* stream);
int bg_maxpool2_gather(const float* x, const float* t, float* y, int N, int H, int W, int C, void* stream);
int bg_prelu_tangent_dalpha(const float* x, const float* xdot, const float* dy, float* dalpha, int64_t rows, int C,
                            void* stream);
int bg_softmax_tangent_bwd(const float* p, const float* sdot, const float* g, float* dp, float* dsdot, int64_t rows,
                           int cols, void* stream);

/* Class-label loss of the conditional model (utils.py:366-369, BigGAN.py:853,894), 'logistic' type:
 *   loss = scale * sum_{b,j} sigmoid_cross_entropy_with_logits(truth, logits)[b,j] * weights[j]
 *   dlogits = scale * weights[j] * (sigmoid(logits) - truth);  scale = loss_weight / (global_batch * n).
 *   weights may be NULL (all ones).  logits/truth/dlogits are [B, n]. */
int bg_sigmoid_ce(const float* logits, const float* truth, const float* weights, float scale, float* loss_out,
                  float* dlogits, int B, int n, void* stream);

/* --------------------------------------------------------------------------------------------
 * Orthogonal-cosine regulariser (utils.py:180-235) from the Gram matrix A = W^T W [c,c]
 * (computed with bg_gemm), using R[i,j] = (sum_k Ahat[i,k] - Ahat[i,j]) / sqrt(c-1):
 *   fwd: loss_accum[0] += scale/2 * sum R^2 ; bwd: dA (so that dW = W (dA + dA^T), via bg_gemm).
 * ------------------------------------------------------------------------------------------ */
int bg_ortho_cosine_fwd_bwd(const float* A, float scale, float* loss_accum, float* dA, int c, void* stream);
/* type 'ortho' (utils.py:199-200): loss_accum += scale/2 * sum (A - I)^2, dA = scale * (A - I). */
/* S = A + A^T (c x c, out of place): d(loss)/dW = W dA + W dA^T = W S, one GEMM instead of two (utils.py:197-203) */
int bg_symmetrize(const float* A, float* S, int c, void* stream);
int bg_ortho_identity_fwd_bwd(const float* A, float scale, float* loss_accum, float* dA, int c, void* stream);
/* Low-rank form of the same function for wide kernels W[rows, c] with rows < c (first/dense2 is
 * [184, 16*16*ch]): with G = W W^T, s = W 1, P = G W (bg_gemm) the c x c Gram matrix is never formed.
 *   cols:   alpha_beta[0:c] = alpha, [c:2c] = beta, Wb = W diag(beta), loss_accum[0] += loss
 *   finish: dW = 2*dW_in + s alpha^T + (W alpha) 1^T + 2 P diag(beta), dW_in = (Wb W^T) W */
/* out[r] = sum_c W[r,c] * v[c] (v == NULL: row sums) - the s = W 1 and W alpha vectors of the low-rank form */
int bg_gemv_rows(const float* w, const float* v, float* out, int rows, int cols, void* stream);
int bg_ortho_lowrank_cols(const float* W, const float* P, const float* s, float scale, float* alpha_beta,
                          float* Wb, float* loss_accum, int rows, int c, void* stream);
int bg_ortho_lowrank_finish(float* dW, const float* P, const float* s, const float* Walpha,
                            const float* alpha_beta, int rows, int c, void* stream);

/* --------------------------------------------------------------------------------------------
 * TF AdamOptimizer (+ MovingAverageOptimizer shadow) over a flat parameter arena (BigGAN.py:923-927):
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + eps);
 *   lr_t = lr sqrt(1-b2^t)/(1-b1^t) computed by the caller; ema (nullable) = d*ema + (1-d)*p.
 *   grad_scale multiplies g first (1/world_size for summed all-reduce).
 * ------------------------------------------------------------------------------------------ */
int bg_adam_tf_ema_step(float* p, const float* g, float* m, float* v, float* ema,
                        float lr_t, float b1, float b2, float eps, float ema_decay, float grad_scale,
                        int64_t n, void* stream);
/* Same update with the bias-corrected step size read from device memory (lr_t_dev[0]): the launch carries no
 * per-step host scalar, so a captured HIP graph of the training step can be replayed. */
int bg_adam_tf_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, const float* lr_t_dev,
                            float b1, float b2, float eps, float ema_decay, float grad_scale, int64_t n,
                            void* stream);

/* --------------------------------------------------------------------------------------------
 * Optional per-kernel timing for bench.py's roofline leg: when enabled, every MFMA GEMM launch
 * (conv / deconv / gemm families) is bracketed by hipEvents on its stream.
 * bg_prof_collect synchronises the events and returns totals since the last reset.
 * ------------------------------------------------------------------------------------------ */
void bg_prof_enable(int on);
void bg_prof_reset(void);
int  bg_prof_collect(double* total_ms, double* total_flops, int64_t* launches);
/* write one CSV line (tag,flops,ms) per recorded launch; does not clear the records */
int  bg_prof_dump(const char* path);

#ifdef __cplusplus
}
#endif
#endif /* BIGGAN_HIP_H */
