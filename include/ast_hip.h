/* libast_hip.so -- C ABI of the MI355X (gfx950) hot path.
 *
 * The reference (francescobrigante/Audio-Style-Transfer) is pure Python: it has
 * no FFI of its own.  The boundary this library replaces is the set of
 * torch/ATen operator calls made by the reference's nn.Module.forward / loss
 * functions; each entry point cites the reference call site it stands in for.
 * The host side (audio-style-transfer_amd/ast_amd) binds these with ctypes and
 * keeps the reference's Python call surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named h_*; the library borrows it
 *     for the duration of the enqueue, allocates nothing, never synchronises and
 *     enqueues on `stream` (a hipStream_t passed as void*), so calls are
 *     hipGraph-capturable;
 *   - return 0 on success, negative on error; ast_last_error() returns the
 *     message (thread-local);
 *   - activations are NHWC with the channel count padded to a multiple of 8;
 *     `dtype` selects their storage / MFMA operand type: AST_F32 (exact f32
 *     MFMA) or AST_BF16 (bf16 MFMA, f32 accumulate); parameters, statistics and
 *     losses are always f32.
 */
#ifndef AST_HIP_H
#define AST_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AST_DTYPE_F32 0
#define AST_DTYPE_BF16 1
#define AST_MAX_TAPS 9

int ast_version(void);
const char* ast_last_error(void);

/* Geometry of one gathered (implicit im2col) GEMM.  Rows of the GEMM are the
 * pixels (n, hm, wm) of a logical Hm x Wm grid; tap t reads source pixel
 * (hm*sh + oh + dh[t], wm*sw + ow + dw[t]) (zero outside the tensor) and weight
 * slice wtap[t]; the result goes to destination pixel (hm*dsh + doh, wm*dsw + dow).
 * This one description covers Conv2d forward, its data gradient (one launch per
 * output-parity class for stride 2), ConvTranspose2d forward/backward and
 * nn.Linear (H = W = 1, one tap). */
typedef struct ast_gather_t {
  int32_t N, Hs, Ws, Cs;          /* source NHWC, Cs multiple of 8 */
  int32_t Hm, Wm;                 /* logical pixel grid */
  int32_t sh, sw, oh, ow;         /* source base coordinate */
  int32_t Hd, Wd, Cd;             /* destination NHWC, Cd multiple of 8 */
  int32_t dsh, dsw, doh, dow;     /* destination coordinate */
  int32_t ntaps, wtaps;           /* taps used / taps per weight row */
  int32_t tap[AST_MAX_TAPS];      /* (dh+64) | (dw+64)<<8 | wtap<<16 */
} ast_gather_t;

/* dst[pix][co] (+)= sum_{t,c} src[gather(pix,t)][c] * wgt[co][wtap[t]][c] + bias[co]
 * Replaces: nn.Conv2d fwd/bwd-data (style_encoder.py:50-67, new_decoder.py:29-61),
 * nn.ConvTranspose2d fwd/bwd-data (new_decoder.py:72-96), nn.Linear fwd/bwd-data.
 * flags: bit0 accumulate into dst, bit1 ReLU, bit2 the split-K workspace is already zero (the finish pass
 * always hands it back zeroed, so a persistent workspace never needs a memset), bit3 `ws` is a zeroed
 * [64][Cd][2] f32 table and every tile adds (sum, sum of squares) of the values it stores into slot
 * (tile index mod 64): nn.BatchNorm2d's batch statistics without a second pass over the output
 * (style_encoder.py:54-58, new_decoder.py:30-61).  Only for plans that do not split K (ast_igemm_plan)
 * and plain stores (bits 0 and 1 clear); reduce with ast_norm_finalize(N = 64, count = pixels).
 * bit6 (with bit3): the table is [N][Cd][2] and every tile adds into the row of ITS IMAGE (tiles are laid out so that none
 * straddles two images): nn.InstanceNorm2d's per-image statistics (style_encoder.py:69) without a pass over the output;
 * gathered and patch kernels only (ast_igemm_plan: kch != 0), reduce with ast_norm_finalize(instance = 1). */
int ast_igemm(const void* src, const void* wgt, const float* bias, void* dst,
              const ast_gather_t* g, int dtype, int flags, float* ws, long ws_floats, void* stream);
/* ast_igemm as the data-gradient GEMM that PRODUCES dy of a BatchNorm2d(+ReLU) layer (flags bit 4; bit 5: no ReLU):
 * bn_x is that layer's input (same geometry and dtype as dst), bn_scale/bn_shift its forward coefficients; every tile adds
 * (sum dz, sum dz*x), dz = dy*[fma(x, scale, shift) > 0], of the values it stores into the zeroed [64][Cd][3] table `ws`,
 * so ast_norm_bwd_sums need not run (reduce with ast_norm_bwd_finalize(N = 64, count = pixels)).  Plans without split-K. */
int ast_igemm_bn(const void* src, const void* wgt, const float* bias, void* dst, const ast_gather_t* g, int dtype, int flags,
                 float* ws, long ws_floats, const void* bn_x, const float* bn_scale, const float* bn_shift, void* stream);
/* Narrow layers (Cd <= 16, ntaps*Cs <= 12 sixteen-byte chunks) run an LDS-free kernel: each lane's gather load is its
 * MFMA fragment, the weights stay in registers (ast_igemm_plan reports it as kch = 0).  Same results, same flags. */
/* f32 workspace (floats) ast_igemm needs for this geometry: >0 when the launch is split over K
 * (under-filled grids of the deep, small-M layers), 0 otherwise, <0 on a bad geometry. */
long ast_igemm_ws_floats(const ast_gather_t* g, int dtype);
/* the tile plan ast_igemm will use: out5 = {BM, BN, 16-byte chunks per row per barrier, grid-level split-K slices, in-workgroup K groups} */
int ast_igemm_plan(const ast_gather_t* g, int dtype, int* out5);

/* dw[cd][wtap[t]][c] += sum_pix dy[pix][cd] * src[gather(pix,t)][c]   (f32 atomics)
 * dy is the plain operand over the logical grid (N,Hm,Wm,Cd).
 * Replaces: weight gradients of Conv2d / ConvTranspose2d / Linear. */
int ast_wgrad(const void* dy, const void* src, float* dw, const ast_gather_t* g,
              int dtype, void* stream);
/* The same with the gradient spread over `nrep` copies of dw (copy r at dw + r * Cd*wtaps*Cs; the workgroups of pixel slice z
 * use copy z % nrep): same-address f32 atomics serialise, and the reader (ast_weight_grads_flush_t with
 * ast_weight_desc_t.dwp_replicas) sums the copies.  All copies must be zeroed by the caller. */
int ast_wgrad_rep(const void* dy, const void* src, float* dw, const ast_gather_t* g, int dtype, int nrep, void* stream);
/* Slab form of the same: the launch cuts the pixels into at most `nslabs` slices, and every slice STORES (plain 256-byte-row
 * stores, no atomics) its partial gradient into its own copy of dW -- slabs[z * Cd*wtaps*Cs ...] for slice z; *slices_out = the
 * number of copies written (every one of them completely: nothing needs to be zeroed first).  ast_slab_sum adds the copies up:
 * copy 0 <- sum of the first slabs[i] copies of record i, for up to AST_MAX_SLAB_RECS weights in ONE launch (bases / floats_per_copy /
 * slabs are HOST arrays).  NOT an accumulate: a weight used twice in one backward pass must use ast_wgrad / ast_wgrad_rep. */
#define AST_MAX_SLAB_RECS 48
int ast_wgrad_slab(const void* dy, const void* src, float* slabs, const ast_gather_t* g, int dtype, int nslabs, int* slices_out,
                   void* stream);
int ast_slab_sum(float* const* bases, const int64_t* floats_per_copy, const int* slabs, int nrec, void* stream);

/* Token-sized nn.Linear (M <= 64 rows, f32): y = act(x W^T + b), W row pitch ldw (W may be a row slice
 * of in_proj_weight or the transposed pack for the data gradient).  One wave per output column. */
int ast_skinny_gemm(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldw,
                    int ldy, int relu, void* stream);
/* ast_skinny_gemm with the fused-FFN epilogues (TransformerEncoderLayer/DecoderLayer._ff_block,
 * linear2(dropout(relu(linear1(x))))):  drop_mask != NULL draws a dropout mask (p, seed, d_offset as ast_dropout_fwd),
 * applies it and stores the COMBINED relu+dropout mask (0 or 1/(1-p)); mul_mask != NULL multiplies the result by a
 * stored mask (the backward's dh = (dy W2) * mask). */
int ast_skinny_gemm_ex(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldw, int ldy,
                       int relu, const float* mul_mask, float* drop_mask, float p, uint64_t seed, const int64_t* d_offset,
                       void* stream);
/* The two 2*287*513 x 256 linears of SimpleDecoder_TransformerOnly.py:16-17 on <= 64 token rows (f32, weights streamed once):
 * ast_bigk_gemm: y[M][N] = x[M][K] w[N][K]^T + bias, K huge and even (stft_to_embedding, :56-60); y is overwritten.
 * ast_bign_dgrad: dx[M][K] = dy[M][N] w[N][K], N huge, K <= 256 (data gradient of embedding_to_stft, :62-66); dx is overwritten.
 * (embedding_to_stft forward is ast_skinny_gemm with a huge N; both weight gradients are ast_linear_wgrad.) */
int ast_bigk_gemm(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldy, void* stream);
int ast_bign_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, void* stream);
/* dW[n][k] += sum_m dy[m][n] x[m][k]; db[n] += sum_m dy[m][n]  -- straight into the parameter gradients */
int ast_linear_wgrad(const float* dy, const float* x, float* dW, float* db, int M, int N, int K, int lddy, int ldw,
                     void* stream);

/* Batched ast_linear_wgrad: table = DEVICE array of `count` records
 * {const float* dy, x; float* dW, db; int32 M, N, K, lddy, ldw, pad[3]} (64 bytes each); max_tiles >= max over
 * records of ceil(K/64)*ceil(N/64).  One launch for every linear layer of a model, after backward. */
int ast_linear_wgrad_batched(const void* table, int count, int max_tiles, void* stream);
/* The same with the records read from HOST memory at call time and passed to the kernel by value (<= 56 per launch):
 * no device table and no host-to-device copy, so the call is a plain kernel node under hipGraph capture. */
int ast_linear_wgrad_batched_host(const void* host_table, int count, int max_tiles, void* stream);

/* ---- layout conversion at the module boundary ------------------------------ */
/* x (N,C,H,W) f32, element (n,c,h,w) at n*sn + c*sc + h*sh + w  ->  NHWC dtype, Cp>=C zero padded.
 * Replaces the implicit NCHW contract of x.view(B*S,C,T,F) (style_encoder.py:213). */
int ast_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int64_t sn, int64_t sc,
                     int64_t sh, int Cp, int dtype, void* stream);
int ast_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int Cp, int dtype, void* stream);
/* dtype casts of flat buffers */
int ast_cast(const void* x, int dtype_in, void* y, int dtype_out, int64_t n, void* stream);

/* ---- spectral norm + weight packing (torch spectral_norm.py:92-114) --------- */
typedef struct ast_weight_desc_t {
  const float* w;       /* weight_orig; element (co,ci,tap) at co*s_co + ci*s_ci + tap */
  float* u;             /* weight_u [Co] or NULL (no spectral norm) */
  float* v;             /* weight_v [Ci*KK] */
  float* sigma;         /* [1] out */
  float* scratch;       /* [ast_sn_scratch_floats(Co, Ci*KK)] = Co + Ci*KK*(1 + 32): s = W v, t = W^T u, row-chunk partials of t */
  void* wf;             /* packed [Cop][KK][Cip]  (rows = out channel) or NULL */
  void* wb;             /* packed [Cip][KK][Cop]  (rows = in channel) or NULL */
  int32_t Co, Ci, KK, s_co, s_ci, Cop, Cip;
  int32_t power_iter;   /* 1 in training, 0 in eval */
  float* dwp;           /* packed f32 gradient staging (zeroed by ast_weights_prepare_t in training) or NULL */
  float* grad;          /* gradient of `w` (same layout as w), accumulated by ast_weight_grads_flush_t */
  float* inner;         /* [1] scratch: <dWp, W/sigma> */
  int32_t dwp_from_wb;  /* 0: dwp is [Cop][KK][Cip]; 1: [Cip][KK][Cop] */
  int32_t dwp_replicas; /* dwp holds this many copies (0 = 1), Cop*KK*Cip floats apart, filled by ast_wgrad_rep; flush sums them */
} ast_weight_desc_t;
/* descs: DEVICE array of n descriptors, dtypes: DEVICE int[n] (packed dtype per weight), tiles: DEVICE array of
 * {int32 weight index, co0, ci0, pad} covering every 32x32 channel tile of every weight (padded extents).
 * prepare: power iteration (if requested), sigma = u^T W v, W/sigma written in both packed layouts and the
 * gradient staging zeroed -- five launches for all n weights; LDS-tiled so every global access is a contiguous run.
 * u, v and sigma are BIT-REPRODUCIBLE functions of (w, u): W^T u is summed over 32 row chunks through per-chunk slabs in a
 * fixed order (no float atomics), so data-parallel replicas -- which never exchange these buffers -- stay identical.
 * flush (once after backward): grad += (dWp - <dWp,W/sigma> u v^T)/sigma for every weight -- two launches. */
/* floats of ast_weight_desc_t.scratch for a (Co x ncols) weight */
long ast_sn_scratch_floats(int Co, int ncols);
int ast_weights_prepare_t(const ast_weight_desc_t* descs, const int* dtypes, int n, int max_co, int max_cols,
                          const void* tiles, int ntiles, void* stream);
int ast_weight_grads_flush_t(const ast_weight_desc_t* descs, const void* tiles, int ntiles, void* stream);
/* g_orig += (dWp - <dWp,W/sigma> u v^T)/sigma, dWp packed [Cop][KK][Cip] (from_wb=0)
 * or [Cip][KK][Cop] (from_wb=1).  u NULL => plain unpack-accumulate. */
int ast_weight_grad_unpack(const float* dwp, int from_wb, const float* w, const float* u, const float* v,
                           const float* sigma, float* g_orig, int Co, int Ci, int KK, int s_co, int s_ci,
                           int Cop, int Cip, float* scratch, void* stream);

/* ---- normalisation (nn.BatchNorm2d / nn.InstanceNorm2d / nn.LayerNorm) ------ */
/* sums[n][c][k]: k=0 sum x, k=1 sum x^2 over the H*W pixels of image n.  The buffer is zeroed inside unless
 * assume_zeroed (a persistent scratch that ast_norm_finalize(zero_sums=1) handed back clean). */
int ast_chan_stats(const void* x, float* sums, int N, int HW, int C, int dtype, int assume_zeroed, void* stream);
/* From sums -> per-channel (batch) or per-(n,c) (instance) scale/shift; updates
 * running stats when running_mean != NULL (momentum 0.1, unbiased var). eval_mode uses running stats.
 * count > 0 (batch form): the N rows of `sums` are partial-sum slots (the [64][C][2] table filled by ast_igemm's
 * flags bit 3) over `count` pixels in total; count == 0: N images of HW pixels. */
int ast_norm_finalize(float* sums, int zero_sums, int64_t* num_batches_tracked /* +1 when not NULL */,
                      int N, int HW, int C, int Creal, int instance,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      int eval_mode, float eps, float* mean, float* rstd, float* scale, float* shift,
                      long count, void* stream);
/* y = act(x*scale[c] + shift[c] + r*scale2[n,c] + shift2[n,c]) ; r may be NULL.
 * flags: bit0 ReLU, bit1 scale/shift are per (n,c) instead of per c. */
int ast_affine_act(const void* x, const float* scale, const float* shift, const void* r,
                   const float* scale2, const float* shift2, void* y, int N, int HW, int C,
                   int flags, int dtype, void* stream);
/* Training-mode BatchNorm2d(+ReLU) [+ InstanceNorm2d of a second input: the ResBlock tail, style_encoder.py:76-83] in ONE
 * launch, statistics finalize included: every workgroup reduces the statistics table tab1 ([rows1][C][2] = {sum x, sum x^2} over
 * count1 pixels: the slot table of ast_igemm flags bit 3, or the [N][C][2] image sums of ast_chan_stats) into the per-channel
 * scale / shift itself, then applies y = act(x*scale + shift [+ r*scale2[n] + shift2[n]]); tab2 = [N][C][2] sums of r.  Workgroup
 * (0,0) writes out1 = [4][C] (mean, rstd, scale, shift) [out2 = [4][N*C]], updates the running statistics (momentum 0.1, unbiased
 * variance) and num_batches_tracked.  The tables are only READ: the caller clears them (one memset per step for all of them).
 * Replaces ast_norm_finalize + ast_affine_act (38 dependent single-wave launches per train step). */
int ast_bn_apply_fwd(const void* x, const void* r, void* y, const float* tab1, int rows1, long count1, const float* tab2,
                     const float* gamma1, const float* beta1, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float eps1, const float* gamma2, const float* beta2, float eps2, float* out1, float* out2, int N, int HW, int C,
                     int Creal, int flags /* bit0 ReLU */, int dtype, void* stream);
/* Backward twin: tab3 = [rows][C][3] = {sum dz, sum dz*x, sum dz*r} (slot table of ast_igemm_bn over `count` pixels, or the
 * [N][C][3] image sums of ast_norm_bwd_sums); dx = k1[c]*(dz, x, 1), dr = k2[n][c]*(dz, r, 1) with the coefficients formed in
 * the kernel; dgamma / dbeta are ADDED by workgroup (0,0).  dz = dy * [fma(x, scale1, shift1) (+ fma(r, scale2[n], shift2[n])) > 0]
 * when flags bit0 (ReLU).  dr != NULL needs rows == N.  Replaces ast_norm_bwd_finalize + ast_norm_bwd_apply. */
int ast_bn_apply_bwd(const void* dy, const void* x, const void* r, void* dx, void* dr, const float* tab3, int rows, long count,
                     const float* gamma1, const float* mean1, const float* rstd1, float* dgamma1, float* dbeta1,
                     const float* gamma2, const float* mean2, const float* rstd2, float* dgamma2, float* dbeta2,
                     const float* scale1, const float* shift1, const float* scale2, const float* shift2, int N, int HW, int C,
                     int Creal, int flags, int dtype, void* stream);
/* backward of the above followed by the norm backward:
 * dz = dy * (y>0 if relu); sums3[n][c] = {sum dz, sum dz*x, sum dz*r} */
int ast_norm_bwd_sums(const void* dy, const void* y, const void* x, const void* r, float* sums3,
                      int N, int HW, int C, int relu, int dtype, int assume_zeroed, void* stream);
/* coefficients k[c][3] (batch branch) and j[n][c][3] (instance branch) and dgamma/dbeta accumulation */
int ast_norm_bwd_finalize(float* sums3, int zero_sums, int N, int HW, int C, int Creal,
                          const float* gamma1, const float* mean1, const float* rstd1,
                          float* dgamma1, float* dbeta1, float* k1,
                          const float* gamma2, const float* mean2, const float* rstd2,
                          float* dgamma2, float* dbeta2, float* k2, void* stream);
/* count > 0: the N rows of sums3 are partial-sum slots (the [64][C][3] table filled by ast_igemm_bn) over `count` pixels */
int ast_norm_bwd_finalize_n(float* sums3, int zero_sums, int N, int HW, int C, int Creal, const float* gamma1,
                            const float* mean1, const float* rstd1, float* dgamma1, float* dbeta1, float* k1,
                            const float* gamma2, const float* mean2, const float* rstd2, float* dgamma2,
                            float* dbeta2, float* k2, long count, void* stream);
/* dx = k1[c][0]*dz + k1[c][1]*x + k1[c][2];  dr = k2[n][c][0]*dz + k2[n][c][1]*r + k2[n][c][2] */
int ast_norm_bwd_apply(const void* dy, const void* y, const void* x, const void* r,
                       const float* k1, const float* k2, void* dx, void* dr,
                       int N, int HW, int C, int relu, int dtype, void* stream);
/* The same two passes with the ReLU mask RECOMPUTED from the pre-activation fma(x, scale1[c], shift1[c])
 * [+ fma(r, scale2[n][c], shift2[n][c])] -- bit-identical to what ast_affine_act formed in the forward pass -- so the
 * activation output y is not read (a third of each pass's HBM traffic); y may be NULL when scale1 is given. */
int ast_norm_bwd_sums_pre(const void* dy, const void* y, const void* x, const void* r, float* sums3, int N, int HW,
                          int C, int relu, int dtype, int assume_zeroed, const float* scale1, const float* shift1,
                          const float* scale2, const float* shift2, void* stream);
int ast_norm_bwd_apply_pre(const void* dy, const void* y, const void* x, const void* r, const float* k1,
                           const float* k2, void* dx, void* dr, int N, int HW, int C, int relu, int dtype,
                           const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                           void* stream);

int ast_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int rows, int D, float eps, int dtype, void* stream);
int ast_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int D,
                      int dtype, void* stream);

/* Fused residual + dropout + LayerNorm on f32 token rows (one launch for `norm(x + dropout(sub))`,
 * torch/nn/modules/transformer.py's post-norm and pre-norm blocks as style_encoder.py:181-187 and
 * new_decoder.py:111-118 instantiate them):
 *   s = x + dropout_p(sub)   (x NULL: s = dropout(sub); p == 0: mask untouched)
 *   y = LayerNorm(s)         (gamma NULL: s only)
 * mask holds 0 or 1/(1-p); seed/d_offset as ast_dropout_fwd.  Backward:
 *   ds = ds_ext + LN'(dy; s);  dx = ds (dx may be NULL);  dsub = ds * mask (mask NULL: ds). */
int ast_add_drop_ln_fwd(const float* x, const float* sub, float* mask, float* s_out, const float* gamma, const float* beta,
                        float* y, float* mean, float* rstd, int rows, int D, float eps, float p, uint64_t seed,
                        const int64_t* d_offset, void* stream);
int ast_add_drop_ln_bwd(const float* dy, const float* ds_ext, const float* s, const float* gamma, const float* mean,
                        const float* rstd, const float* mask, float* dx, float* dsub, float* dgamma, float* dbeta, int rows,
                        int D, void* stream);

/* Y[R_out][D] = A[R_out][R_in] X[R_in][D], A a small dense coefficient matrix (R_in <= 8192): class prototypes = per-class
 * means of style embeddings (style_encoder.py:243-253), per-row prototype gather, means over the section axis
 * (losses.py:88,142); the backward pass is the same call with A transposed. */
int ast_rowmix(const float* A, const float* X, float* Y, int R_out, int R_in, int D, void* stream);

/* ---- pooling / resampling ---------------------------------------------------- */
/* nn.AdaptiveAvgPool2d with bins [floor(i*In/Out), ceil((i+1)*In/Out)) (style_encoder.py:113-114, new_decoder.py:51) */
int ast_adaptive_pool_fwd(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype, void* stream);
int ast_adaptive_pool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int dtype, void* stream);
/* nn.Upsample(bilinear, align_corners=False) NHWC(Cp) -> NCHW f32 (new_decoder.py:99) */
int ast_bilinear_fwd(const void* x, float* y, int N, int C, int Cp, int H, int W, int Ho, int Wo, int dtype, void* stream);
int ast_bilinear_bwd(const float* dy, void* dx, int N, int C, int Cp, int H, int W, int Ho, int Wo, int dtype, void* stream);

/* ---- small-sequence attention core (nn.MultiheadAttention inner product part) - */
/* q:(B,Lq,ldq) k,v:(B,Lk,ldk) rows of f32 (already projected); heads of width dh; causal optional;
 * p_out (B,H,Lq,Lk) saved probabilities (after dropout mask scaling); drop_mask optional (B,H,Lq,Lk) of 0/1/(1-p) */
int ast_attn_fwd(const float* q, const float* k, const float* v, float* o, float* probs,
                 int B, int H, int Lq, int Lk, int dh, int ldq, int ldk, int ldo, int causal,
                 const float* drop_mask, void* stream);
int ast_attn_bwd(const float* dout, const float* q, const float* k, const float* v, const float* probs,
                 float* dq, float* dk, float* dv, int B, int H, int Lq, int Lk, int dh, int ldq,
                 int ldk, int ldo, const float* drop_mask, void* stream);
/* The same with the attention-probability dropout DRAWN in the kernel (p, seed, d_offset as ast_dropout_fwd; element
 * index = ((b*H + h)*Lq + i)*Lk + j): forward and backward draw identical values, so no mask tensor is stored and no
 * mask kernel runs (nn.MultiheadAttention(dropout=0.1), style_encoder.py:181-187, new_decoder.py:111-118). */
int ast_attn_fwd_p(const float* q, const float* k, const float* v, float* o, float* probs, int B, int H, int Lq, int Lk,
                   int dh, int ldq, int ldk, int ldo, int causal, const float* drop_mask, float p, uint64_t seed,
                   const int64_t* d_offset, void* stream);
int ast_attn_bwd_p(const float* dout, const float* q, const float* k, const float* v, const float* probs, float* dq, float* dk,
                   float* dv, int B, int H, int Lq, int Lk, int dh, int ldq, int ldk, int ldo, const float* drop_mask,
                   float p, uint64_t seed, const int64_t* d_offset, void* stream);

/* ---- elementwise ------------------------------------------------------------- */
/* out[c] += sum_r x[r][c], c < Creal  (bias gradients of Conv2d / Linear) */
int ast_colsum_acc(const void* x, int64_t rows, int C, int Creal, float* out, int dtype, void* stream);
int ast_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);
int ast_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);
/* fused counter-based dropout: mask[i] in {0, 1/(1-p)} is generated, stored (for backward) and applied: y = x*mask */
int ast_dropout_fwd(const float* x, float* y, float* mask, int64_t n, float p, uint64_t seed, const int64_t* d_offset, void* stream);
/* counter-based dropout: mask[i] in {0, 1/(1-p)} (f32), y = x*mask */
int ast_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, const int64_t* d_offset, void* stream);
int ast_mul(const void* a, const float* mask, void* y, int64_t n, int dtype, void* stream);

/* ---- losses -------------------------------------------------------------------- */
/* compute_comprehensive_loss (new_decoder.py:348-420), one pass: out (B,S,2,T,F) f32 contiguous,
 * tgt same logical shape with row stride tgt_ld (a [..., :513] view of x).  sums[5] raw sums
 * (mse, mag, phase, temporal, spectral); grad = d total / d out (may be NULL). */
int ast_recon_loss(const float* out, const float* tgt, int64_t tgt_ld, int B, int S, int T, int Fq,
                   float w_mse, float w_mag, float w_phase, float w_temporal, float w_spectral,
                   float* sums, float* grad, void* stream);
/* ast_recon_loss with the weighted total and the reported means formed on the device: coef5 / inv5 are HOST arrays (the five weights
 * c_mse .. c_spectral and the five 1/count factors of new_decoder.py:402-418), ws is AST_RECON_SLOTS x 5 floats of scratch (zeroed by
 * the call), res11 receives [5 raw sums][total = sum_k coef5[k] * sums[k]][5 means = inv5[k] * sums[k]].  grad as in ast_recon_loss. */
#define AST_RECON_SLOTS 64
int ast_recon_loss_total(const float* out, const float* tgt, int64_t tgt_ld, int B, int S, int T, int Fq, const float* coef5,
                         const float* inv5, float* ws, float* res11, float* grad, void* stream);
/* losses.py on (B,256) embeddings; each writes loss[0] and optional gradients */
int ast_infonce(const float* emb, const int32_t* labels, int B, int D, float temperature,
                float* loss, float* demb, float* ws /* 2*B*B + B floats */, void* stream);
int ast_margin(const float* cls, int C, int D, float margin, float* loss, float* dcls, void* stream);
int ast_hsic(const float* s, const float* c, int B, int D, float* loss, float* ds, float* dc,
             float* ws /* 6*B*B + 2*B + 8 floats */, void* stream);
/* cross-covariance variant of disentanglement_loss (losses.py:146-150); ws: 2*D + 2*B*B floats */
int ast_crosscov(const float* s, const float* c, int B, int D, float* loss, float* ds, float* dc, float* ws, void* stream);
/* mean cross entropy over rows of logits (R,C); dlogits optional */
int ast_cross_entropy(const float* logits, const int32_t* target, int R, int C, float* loss, float* dlogits, void* stream);
/* mean entropy with log(p+1e-8) (losses.py:118-120) */
int ast_softmax_entropy(const float* logits, int R, int C, float* loss, float* dlogits, void* stream);

/* y = (accumulate ? y : 0) + hscale * (dscale ? *dscale : 1) * x   (chain rule for scalar losses) */
int ast_scale(const float* x, const float* dscale, float hscale, float* y, int64_t n, int accumulate, void* stream);

/* ---- optimiser ------------------------------------------------------------------ */
int ast_counter_incr(int64_t* c, void* stream);
int ast_sumsq(const float* x, int64_t n, float* out /* accumulates */, void* stream);
/* Adam with bias correction; grad scaled by min(1, max_norm/(sqrt(*gnorm_sq)+1e-6)) when gnorm_sq != NULL */
int ast_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
             float eps, float wd, const int64_t* d_step, const float* gnorm_sq, float max_norm, void* stream);

/* Step scalars on the device, so that ONE captured hipGraph follows a learning-rate schedule, a ramped adversarial weight or
 * a changed clip norm (the reconstructed train2 step: `scheduler.step()` and lambda_adv(t); README.md:144-150).
 * ast_adam_dev: as ast_adam, with [lr, max_norm] read from device memory at run time (max_norm <= 0: no clipping).
 * ast_set_values: dst[i] = host_vals[i], i < n <= AST_MAX_STEP_SCALARS -- the values travel as kernel arguments (stream-ordered,
 * no pinned staging buffer to race with).
 * ast_weighted_sum: out[0] = sum_i weights[widx[i]] * terms[i][0] (widx[i] < 0: weight 1), terms added in order -- the total loss
 * `w_rec * rec + w_nce * nce + ...` of the train step in one launch; _bwd: grads[i] = g[0] * weights[widx[i]].
 * `terms` / `widx` are HOST arrays of n <= AST_MAX_STEP_SCALARS entries (device pointers / indices into `weights`). */
#define AST_MAX_STEP_SCALARS 16
int ast_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* d_hyper /* [lr, max_norm] */, float b1, float b2,
                 float eps, float wd, const int64_t* d_step, const float* gnorm_sq, void* stream);
int ast_set_values(float* dst, const float* host_vals, int n, void* stream);
int ast_weighted_sum(const float* const* terms, const int* widx, int n, const float* weights, float* out, void* stream);
int ast_weighted_sum_bwd(const float* g, const int* widx, int n, const float* weights, float* grads, void* stream);

/* ---- STFT front-end (utilityFunctions.py:12-37 + dataloader.py:9-13 + utilityFunctions.py:240-263) */
/* wave (Bc, nsamp) f32 -> x (Bc, S, 2, 287, F_total) f32: frames of a 1024-point Hann STFT (hop 256,
 * reflect padded), z-scored with mean/std (2,513), cut into S sections of 287 frames (step 191), written
 * to bins [0,513) of the F_total-wide rows. */
/* inverse_STFT (utilityFunctions.py:62-82; torch.istft defaults): spec (Bc,2,T,513) f32 -> wave (Bc, 256*(T-1));
 * frames_ws: Bc*T*1024 floats of scratch. */
int ast_istft(const float* spec, int Bc, int T, float* frames_ws, float* wave, void* stream);
/* sections2spectrogram (utilityFunctions.py:265-283): sections (Bc,S,2,wind,F_in) -> out (Bc,2,out_T,F_out), the
 * count-normalised overlap-average at step `hop`, bins [0,F_out) only, truncated to out_T <= hop*(S-1)+wind frames. */
/* Dataset statistics, one clip at a time (Preprocessing_Dataset/compute_unified_stats.py:34-44): per (channel, bin) mean over
 * the T frames and unbiased variance of a contiguous (C,T,F) f32 spectrogram, ADDED into mean_acc / var_acc (C*F floats). */
int ast_bin_stats_acc(const float* x, float* mean_acc, float* var_acc, int C, int T, int F, void* stream);
/* normalize (dataloader.py:9-13): out = (x - mean[c][f]) / (std[c][f] + eps) over a contiguous (C,T,F) f32 spectrogram */
int ast_zscore(const float* x, const float* mean, const float* std_, float* out, int C, int T, int F, float eps, void* stream);
int ast_sections_overlap_avg(const float* sections, float* out, int Bc, int S, int wind, int hop, int F_in, int F_out,
                             int out_T, void* stream);
int ast_stft_sections(const float* wave, int Bc, int nsamp, const float* mean, const float* std_,
                      float* x, int S, int win, int step, int F_total, void* stream);

/* get_CQT (utilityFunctions.py:39-60 -> librosa.cqt), all octaves in one launch.  For octave o (host arrays of n_oct
 * entries): ys[o] = device pointer to the o-times-halved signals (B rows of ns[o] floats), hops[o] its hop, los[o] its first
 * bin, nfs[o] its bin count, row0s[o] its first kernel row.  out[b][0|1][t][bin_off+los[o]+k] = Re|Im( scale[los[o]+k] *
 * sum_i y_o[b][t*hop_o - nfft/2 + i] * (w_re + i w_im)[row0s[o]+k][i] ), y = 0 outside the signal (pad_mode="constant").
 * w_* (rows, nfft), scale (n_bins), out (B, 2, T, ld) f32.  The kernels w fold librosa's rectangular-window STFT and its
 * sparsified wavelet FFT basis (built on the host, ast_amd/cqt.py). */
int ast_cqt_octaves(const float* const* ys, const int* ns, const int* hops, const int* los, const int* nfs, const int* row0s,
                    int n_oct, int B, const float* w_re, const float* w_im, const float* scale, int nfft, float* out, int T,
                    int ld, int bin_off, void* stream);
/* normalize + get_overlap_windows of the CQT planes (dataloader.py:9-18, utilityFunctions.py:240-263) into the bins behind
 * the STFT's: x[b][s][c][w][bin0+k] = (cqt[b][c][s*step+w][k] - mean[c][k]) / (std[c][k] + 1e-8), 0 past frame T-1.
 * cqt (Bc,2,T,nb), x (Bc,S,2,win,F_total) f32.  The CQT twin of ast_stft_sections. */
int ast_cqt_sections(const float* cqt, int Bc, int T, int nb, const float* mean, const float* std_, float* x, int S, int win,
                     int step, int F_total, int bin0, void* stream);
/* Polyphase FIR resampler: y[b][i*nnew + p] = gain * sum_k kern[p][k] * x[b][i*orig + k - width], x = 0 outside [0,n);
 * kern (nnew, klen), y (B, m).  Replaces torchaudio.functional.resample in load_audio (utilityFunctions.py:116-117;
 * kern = its sinc_interp_hann bank) and the halving resample between CQT octaves (orig=2, nnew=1). */
int ast_resample_poly(const float* x, int B, int n, const float* kern, int orig, int nnew, int klen, int width, float* y, int m,
                      float gain, void* stream);

/* ---- token programs: transformer layers in one launch (csrc/tokprog.hip) ------------------------------------------------
 * Replaces, for the <= 64 token rows of the transformer stacks (style_encoder.py:181-191, content_encoder.py:24-26,
 * new_decoder.py:49-51,111-119), the per-operator launches above (ast_skinny_gemm*, ast_attn_fwd_p / ast_attn_bwd_p,
 * ast_add_drop_ln_fwd / _bwd) by ONE launch that walks a list of ops with a grid barrier between them.  All tensors are
 * f32 row-major; an op reads what earlier ops of the same launch wrote.  ast_tok_max_ops() ops per launch at most.
 *   AST_TOK_GEMM      y[M][N] = epi(x[M][K] w[N][K]^T): i = {M, N, K, ldx, ldw, ldy}; in = {x, w, bias?, mul_mask?, addend?};
 *                     out = {y, drop_mask?}; flags & 1: ReLU.  epi: +bias, ReLU, dropout (p, seed; the combined ReLU &
 *                     dropout mask goes to drop_mask), * mul_mask, + addend.  M <= 64, N % 16 == 0, K % 64 == 0.
 *   AST_TOK_ATTN_FWD  softmax(q k^T / sqrt(dh)) v per (batch, head), Lq, Lk <= 8: i = {B, H, Lq, Lk, dh, ldq, ldk, ldo}; in = {q, k, v};
 *                     out = {o, probs (B,H,Lq,Lk)}; flags & 4: causal; p, seed: dropout on the probabilities (redrawn by _BWD).
 *   AST_TOK_ATTN_BWD  same i; in = {dout, q, k, v, probs}; out = {dq, dk, dv}.
 *   AST_TOK_ADLN_FWD  s = x + dropout(sub), y = LayerNorm(s): i = {rows, 256}; in = {x?, sub, gamma?, beta?};
 *                     out = {mask?, s?, y?, mean?, rstd?}; p, seed, eps.
 *   AST_TOK_ADLN_BWD  ds = ds_ext + LayerNorm_bwd(dy; s), dx = ds, dsub = ds * mask: i = {rows, 256};
 *                     in = {dy?, ds_ext?, s, gamma, mean, rstd, mask?}; out = {dx?, dsub?, dgamma?, dbeta?} (+= for the last two).
 * flags & AST_TOK_NO_BARRIER: the next op does not depend on this one (no grid barrier between them).
 * G (<= 32) workgroups do the work (those of an 8 G grid that land on XCD xcd); sync = 32 zeroed uint32 (128-byte aligned) owned by this
 * call chain (launches that may run concurrently need their own); *status becomes 1 if a barrier wait timed out;
 * d_offset = the device step counter of the dropout draws (as ast_dropout_fwd). */
enum { AST_TOK_GEMM = 1, AST_TOK_ATTN_FWD = 2, AST_TOK_ATTN_BWD = 3, AST_TOK_ADLN_FWD = 4, AST_TOK_ADLN_BWD = 5 };
enum { AST_TOK_RELU = 1, AST_TOK_NO_BARRIER = 2, AST_TOK_CAUSAL = 4 };
typedef struct {
  int32_t type, flags;
  int32_t i[8];
  float p, eps;
  uint64_t seed;
  const float* in[7];
  float* out[5];
} ast_tok_op_t;
int ast_tok_max_ops(void);
int ast_tok_program(const ast_tok_op_t* ops, int nops, int G, int xcd, void* sync, int* status, const int64_t* d_offset,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif
