/*
 * vlg_hip.h - C ABI of libvlg_hip.so, the MI355X (gfx950) kernels behind the
 * per-clip layout-generation training step.
 *
 * The reference (gongaa/video-layout-generation) has no native code and no FFI:
 * every device op is a stock torch.nn call made from src/trainer.py.  Each entry
 * point below therefore cites the torch call site in the reference it replaces
 * (file:line under /root/reference), or says SELF-ORACLE where BASELINE.json
 * names an op the reference does not contain.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless
 *     the name ends in _host; the caller (PyTorch) owns every buffer
 *   - `stream` is a hipStream_t passed as void*; kernels are stream-ordered on it,
 *     never allocate, never synchronise, keep no global state
 *   - return value: 0 on success, otherwise a hipError_t (launch/config error) or
 *     VLG_ERR_* below; the Python side raises RuntimeError on non-zero
 *   - token-major activations use the INTERNAL row order  m = (b*N + n)*T + t
 *     (all frames of one slot contiguous: one T x d tile per (clip, slot));
 *     inputs/targets keep the public (B,T,N) order and are re-indexed in-kernel
 */
#ifndef VLG_HIP_H
#define VLG_HIP_H

#include <stdint.h>

/* Every entry point that writes per-block partial sums ("slabs") takes slab_capacity = the number of floats available at
 * `slabs`; it is checked against slab count x slab_stride on the host BEFORE the launch (VLG_ERR_SHAPE), so a short buffer
 * can never become an out-of-bounds device write. */
typedef uint16_t vlg_bf16;    /* bfloat16 bit pattern (activation storage of the bf16 mode) */

#ifdef __cplusplus
extern "C" {
#endif

#define VLG_ERR_SHAPE   1001   /* unsupported or inconsistent shape argument */
#define VLG_ERR_ALIGN   1002   /* pointer or leading dimension not 16-byte aligned */

/* library / build identification */
int         vlg_abi_version(void);
const char* vlg_build_arch(void);          /* "gfx950" */

/* DIAGNOSTIC BUILD ONLY (`make -C csrc diag` -> libvlg_hip_diag.so, compiled with -DVLG_DIAG; the product library does
 * NOT export these and keeps no mutable process-wide state - its only configuration is environment variables read once).
 * Development tools load that build through VLG_HIP_LIB (tools/diag, tools/ab). */
#ifdef VLG_DIAG
/* when set to a device buffer of 2*blocks uint64, every block of the next GEMM launches records {shader-clock ticks,
 * 100 MHz ticks} of its main loop; NULL switches it off (tools/diag/gemm_shader_clock.py) */
void vlg_debug_set_clock_probe(unsigned long long* buf);
void vlg_debug_set_conv_probe(unsigned long long* buf);
/* force the contraction depth per LDS tile of the 128x128 fp32 GEMM kernels (16 | 32; 0 = the library's own choice per
 * epilogue; the VLG_GEMM_BK environment variable sets the initial value) (tools/ab/gemm_ab.py) */
void vlg_debug_set_gemm_bk(int bk);
/* consecutive N tiles per block of the chained fp32 GEMM path (0 = never chain, -1 = the library's choice; bit 16: chained
 * launches as ping-pong pairs of four-wave groups - measured slower, see csrc/gemm.hip) */
void vlg_debug_set_gemm_run(int run);
#endif

/* ------------------------------------------------------------------ embedding
 * Object-slot embedding.  SELF-ORACLE; lookup semantics = nn.Embedding row gather
 * (reference src/models/simple.py:23,41).
 *   x[m,:] = cls_emb[slot_class[b,t,n]] + box_w @ slot_box[b,t,n] + box_b + time_emb[t]
 * bwd writes per-block partial slabs laid out [cls_emb | box_w | box_b | time_emb]
 * (the first tensors of the flat gradient buffer); vlg_reduce_slabs sums them. */
int vlg_embed_fwd(const int64_t* slot_class, const float* slot_box,
                  const float* cls_emb, const float* box_w, const float* box_b,
                  const float* time_emb, float* x,
                  int B, int T, int N, int d, int vocab, void* stream);
int vlg_embed_bwd_slabs(void);             /* upper bound of the number of slabs any vlg_embed_bwd launch writes */
int vlg_embed_bwd_slabs_for(int B, int T, int N, int d, int vocab);   /* slabs THIS shape's launch writes (<= the bound): reduce exactly these */
int vlg_embed_bwd(const float* dx, const int64_t* slot_class, const float* slot_box,
                  float* slabs, int64_t slab_stride, int64_t slab_capacity,
                  int B, int T, int N, int d, int vocab, void* stream);

/* ------------------------------------------------------------------ layer-norm
 * SELF-ORACLE (F.layer_norm arithmetic, biased variance, eps inside the sqrt).
 * bwd: dx_out = (dres ? dres : 0) + LN'(dy); partial [dgamma | dbeta] slabs. */
int vlg_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                      float* y, float* mean, float* rstd,
                      int64_t rows, int d, float eps, void* stream);
/* same, y written as bf16 (the normalised activation only feeds a projection in the bf16 mode) */
int vlg_layernorm_fwd_bf16(const float* x, const float* gamma, const float* beta,
                           vlg_bf16* y, float* mean, float* rstd,
                           int64_t rows, int d, float eps, void* stream);
int vlg_layernorm_bwd_slabs(int64_t rows);
int vlg_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                      const float* gamma, const float* dres, float* dx_out,
                      float* slabs, int64_t slab_stride, int64_t slab_capacity,
                      int64_t rows, int d, void* stream);
/* same, dy read as bf16 (it comes out of a projection's data gradient); x, dres, dx stay fp32 */
int vlg_layernorm_bwd_bf16(const vlg_bf16* dy, const float* x, const float* mean, const float* rstd,
                           const float* gamma, const float* dres, float* dx_out,
                           float* slabs, int64_t slab_stride, int64_t slab_capacity,
                           int64_t rows, int d, void* stream);

/* ------------------------------------------------------------------------ GEMM
 * fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMMs for the dense QKV/FFN/head projections.
 * SELF-ORACLE (F.linear arithmetic); the reference's dense contraction is conv only.
 *
 * vlg_linear_fwd : C[M,N] = A[M,K] . W[N,K]^T + bias      (epilogue selects extras)
 * vlg_linear_dgrad: C[M,K] = A[M,N] . W[N,K]              (epilogue selects extras)
 * vlg_linear_wgrad: slabs[s][N*K + N] = partial ( dY[M,N]^T . X[M,K] | colsum dY )
 */
#define VLG_EPI_NONE   0
#define VLG_EPI_BIAS   1   /* + bias[col]                                            */
#define VLG_EPI_GELU   2   /* aux_out = pre-activation, C = gelu(pre)   (needs BIAS) */
#define VLG_EPI_RESID  4   /* C = acc (+bias) + aux_in[row,col]                      */
#define VLG_EPI_DGELU  8   /* C = acc * gelu'(aux_in[row,col])                       */
#define VLG_EPI_BF16   16  /* v_mfma_f32_32x32x16_bf16 with fp32 accumulate (BASELINE.json configs[2]); fp32 operands are rounded
                              to bf16 on their way to LDS, bf16 ones (storage bits below) are copied; default is native
                              fp32 MFMA */
#define VLG_EPI_SPLIT3 256  /* fp32 tensors, fp32-grade result on the bf16 matrix cores: each operand is split exactly into
                              three bf16 terms and a product block is six v_mfma_f32_32x32x16_bf16 (a1b1 + a1b2 + a2b1 +
                              a1b3 + a2b2 + a3b1, fp32 accumulate; the dropped terms are <= 3 * 2^-24 |a||b|)          */
#define VLG_EPI_ACT_GELU 512 /* native fp32 path: the activation operand (A of vlg_linear_fwd with BIAS | RESID, X of
                              vlg_linear_wgrad) holds PRE-activations and passes through GELU on its way to LDS - the
                              FFN hidden activation gelu(u) is then never written to HBM: the first projection stores
                              u only (plain BIAS epilogue), the second projection and its weight gradient recompute    */
#define VLG_EPI_GELU_GRAD 1024 /* native fp32 and bf16 paths, with BIAS | GELU: aux_out = gelu'(pre) instead of the pre-activation itself.
                              The epilogue has exp(-pre^2/2) and the tail polynomial in registers anyway, and the backward
                              pass needs the pre-activation for nothing but this derivative: the data gradient of the
                              second projection then takes VLG_EPI_MUL instead of VLG_EPI_DGELU (one multiply per
                              element instead of ~20 vector instructions - which the fp32 MFMA kernels pay for in matrix
                              time, see csrc/common.h)                                                                  */
#define VLG_EPI_MUL    2048 /* native fp32 and bf16 paths, vlg_linear_dgrad: C = acc * aux_in[row,col]                            */
/* bf16 ACTIVATION STORAGE (with VLG_EPI_BF16 only): the activation operands named below are vlg_bf16 arrays in
 * HBM instead of float - half the bytes of a mode that is HBM-bound.  Biases, gradient slabs, master weights and
 * the residual stream stay fp32; leading dimensions count elements.                                             */
#define VLG_EPI_A_BF16   32   /* first operand (A of fwd, dY of dgrad / wgrad) is bf16                          */
#define VLG_EPI_B_BF16   64   /* second operand is bf16: X of wgrad, W of fwd / dgrad (the shadow vlg_adam_step_bf16 keeps) */
#define VLG_EPI_OUT_BF16 128  /* C and the epilogue's auxiliary operands (aux_in, aux_out) are bf16             */
int vlg_linear_fwd(const void* A, int lda, const void* W, int ldw, const float* bias,
                   void* C, int ldc, const void* aux_in, void* aux_out,
                   int64_t M, int N, int K, int epilogue, void* stream);
int vlg_linear_dgrad(const void* dY, int ldy, const void* W, int ldw,
                     void* dX, int ldx, const void* aux_in,
                     int64_t M, int N, int K, int epilogue, void* stream);
int vlg_linear_wgrad_slabs(int64_t M, int N, int K);                 /* slab count of a flags = 0 launch */
int vlg_linear_wgrad_slabs_for(int64_t M, int N, int K, int flags);  /* slab count of a launch with these flags */
int vlg_linear_wgrad(const void* dY, int ldy, const void* X, int ldx,
                     float* slabs, int64_t slab_stride, int64_t slab_capacity,
                     int64_t M, int N, int K, int flags /* VLG_EPI_BF16 | storage bits */, void* stream);
/* vlg_linear_wgrad and vlg_linear_dgrad of ONE projection (same dY) as one call - and, for native fp32 tensors where it pays
 * (few tokens: neither product fills the chip alone), as ONE launch whose blocks are dealt both problems at once; results
 * are bit for bit those of the two calls.  epilogue: VLG_EPI_NONE | VLG_EPI_MUL (+ VLG_EPI_BF16 / VLG_EPI_SPLIT3 with fp32
 * storage: two launches).  Replaces the two autograd nodes of one nn.Linear backward. */
int vlg_linear_dgrad_wgrad(const void* dY, int ldy, const void* W, int ldw, void* dX, int ldx, const void* aux_in,
                           const void* X, int ldxx, float* slabs, int64_t slab_stride, int64_t slab_capacity,
                           int64_t M, int N, int K, int epilogue,
                           const int64_t* rider_table /* NULL, or a vlg_reduce_slabs_table table over OTHER buffers (a finished
                              gradient bucket's partial sums): reduced by extra blocks of the same launch where the products are
                              fused, by its own launch otherwise */, int rider_rows, void* stream);


/* ------------------------------------------------------------------- attention
 * Temporal encoder core: causal softmax attention along T for each (clip, slot,
 * head); one wavefront owns one T x 64 tile staged in LDS.  SELF-ORACLE.
 * qkv is [rows, 3d] = [q | k | v]; head h owns columns h*64..h*64+63 of each. */
int vlg_attention_fwd(const float* qkv, float* o, int64_t n_seq, int T, int d, void* stream);
int vlg_attention_bwd(const float* qkv, const float* dout, float* dqkv,
                      int64_t n_seq, int T, int d, void* stream);
/* same with every tensor stored as bf16 (arithmetic stays fp32) */
int vlg_attention_fwd_bf16(const vlg_bf16* qkv, vlg_bf16* o, int64_t n_seq, int T, int d, void* stream);
int vlg_attention_bwd_bf16(const vlg_bf16* qkv, const vlg_bf16* dout, vlg_bf16* dqkv,
                           int64_t n_seq, int T, int d, void* stream);

/* ---------------------------------------------------------------------- losses
 * Fused softmax-cross-entropy + smooth-L1 + IoU, forward and backward in one pass.
 *   CE    : nn.CrossEntropyLoss(reduction='mean')   reference src/trainer.py:124,250
 *   weights 40 / 20 / 10                            reference src/trainer.py:248-250
 *   smooth-L1, IoU : SELF-ORACLE
 * out/dout are [rows, ld] with columns [0,C) = logits, [C,C+4) = raw box.
 * loss_out[4] = {total, smooth_l1, iou, ce}.  scratch holds >= vlg_layout_loss_scratch()
 * floats, 16-byte aligned, ZERO-INITIALISED ONCE by the caller: scratch[1] is the integer ticket by which the last
 * block of the (single) launch folds the per-block partials into loss_out; that block leaves it zero again. */
int vlg_layout_loss_scratch(void);
int vlg_layout_loss(const float* out, int ld, const int64_t* tgt_class, const float* tgt_box,
                    const float* valid, float* dout, float* loss_out, float* scratch,
                    int B, int T, int N, int n_classes, float beta, float iou_eps,
                    float w_reg, float w_iou, float w_ce, void* stream);

/* ------------------------------------------------------------------ per-clip attention (option attention = "clip")
 * Block-causal softmax attention over ALL T*N tokens of a clip, per (clip, head): token (t, n) attends to every slot of
 * frames <= t; slots with valid == 0 (valid may be NULL: none) are never keys, except for themselves.  SELF-ORACLE
 * (oracle/layout_spec.py:clip_attention) - the reference has no attention; this is the "per-clip tile" reading of
 * BASELINE.json's temporal encoder next to the per-slot default (vlg_attention_fwd / _bwd).  fp32, head dim 64, T*N a multiple
 * of 32.  qkv [B*N*T, 3d] / out, dout [B*N*T, d] in the internal row order; valid (B,T,N) floats in the public order;
 * lse and delta: B * (d/64) * T*N floats each (log-sum-exp in the log2 domain, written by fwd; <dO, O>, scratch of bwd).
 * bwd = two launches (query owner: dQ; key owner: dK, dV): every gradient element has one owner, no atomics. */
int vlg_attention_clip_fwd(const float* qkv, const float* valid, float* out, float* lse,
                           int64_t B, int T, int N, int d, void* stream);
int vlg_attention_clip_bwd(const float* qkv, const float* valid, const float* out, const float* dout, const float* lse,
                           float* delta, float* dqkv, int64_t B, int T, int N, int d, void* stream);

/* ------------------------------------------------------------------ reductions */
int vlg_reduce_slabs(const float* slabs, int64_t slab_stride, int n_slabs,
                     float* dst, int64_t len, void* stream);

/* Adam whose step counter lives on the device, for a step captured in a hipGraph (a replayed launch cannot take new
 * by-value arguments): state = {float step_size, float sqrt_bc2, int step, int pad}, 16-byte aligned; the call advances
 * the counter (advance != 0), recomputes the two bias-correction factors in double as vlg_adam_step does on the host,
 * then updates.  shadow may be NULL. */
int vlg_adam_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, vlg_bf16* shadow,
                        int64_t n, float* state, int advance, float lr, float beta1, float beta2, float eps,
                        float grad_scale, void* stream);

/* Many reductions in one launch, driven by a DEVICE table (static graphs: the reference GridNet's 61 convolutions).
 * vlg_reduce_slabs_table: row i = {slabs pointer, slab stride, slab count, destination pointer, length} (int64 each);
 * every row is reduced exactly as vlg_reduce_slabs would.  vlg_sum_partials_table: row i = {partials pointer, count,
 * destination pointer}: dst[0] = sum (vlg_sum_partials with accumulate = 0). */
int vlg_reduce_slabs_table(const int64_t* table, int n_rows, int blocks_per_row, void* stream);
int vlg_sum_partials_table(const int64_t* table, int n_rows, void* stream);

/* ------------------------------------------------------------------- optimiser
 * torch.optim.Adam(lr, betas=(beta1, 0.999)) on one flat fp32 buffer
 * (reference src/trainer.py:83,258; src/main.py:139-141).  `step` is 1-based.
 * grad_scale multiplies the gradient first (1/world_size turns an all-reduce SUM
 * into DDP's mean, reference src/trainer.py:113). */
int vlg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t n, int step, float lr, float beta1, float beta2, float eps,
                  float grad_scale, void* stream);
/* same, and shadow[i] = bf16(param[i]) after the update: the weight operand of the bf16-MFMA projections */
int vlg_adam_step_bf16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, vlg_bf16* shadow,
                       int64_t n, int step, float lr, float beta1, float beta2, float eps,
                       float grad_scale, void* stream);

/* ---------------------------------------------------- reference-real image ops
 * Pixel-space ops of the reference step that exist verbatim in the reference.
 *   vlg_ce_nchw        nn.CrossEntropyLoss('mean') on (b,C,H,W) logits / (b,H,W) int64
 *                      targets; writes loss[0] and dlogits = grad_scale * dCE
 *                                                     reference src/trainer.py:124,250
 *   vlg_l1_mean        nn.L1Loss()                    reference src/trainer.py:130,248
 *   vlg_gradient_loss  GradientLoss.forward           reference src/loss.py:20-25
 *   vlg_ssim_loss      SsimLoss.SSIM/forward          reference src/loss.py:68-91
 *   vlg_prep_input     normalise + concat + flip      reference src/trainer.py:193-206
 * Each loss writes value to loss[0] and, when grad != NULL, d(loss)/d(first arg) *
 * grad_scale.  scratch >= vlg_image_loss_scratch() floats. */
int vlg_image_loss_scratch(void);
int vlg_ce_nchw(const float* logits, const int64_t* target, float* dlogits, float* loss,
                float* scratch, int b, int C, int64_t hw, float grad_scale, void* stream);
int vlg_l1_mean(const float* a, const float* b, float* da, float* loss, float* scratch,
                int64_t n, float grad_scale, void* stream);
int vlg_gradient_loss(const float* a, const float* b, float* da, float* loss, float* scratch,
                      int planes, int H, int W, float grad_scale, void* stream);
int vlg_ssim_loss(const float* x, const float* y, float* dx, float* loss, float* scratch,
                  int b, int C, int H, int W, float grad_scale, void* stream);
/* dst = (src - shift[c]) * scale[c] on (b,C,H,W), C <= 4; shift/scale are HOST arrays of C floats.
 * img = (img - mean_arr) / std_arr, reference src/trainer.py:120-121,212 (and its transpose in backward). */
int vlg_affine_nchw(const float* src, float* dst, int b, int C, int64_t hw, const float* shift_host,
                    const float* scale_host, void* stream);
int vlg_prep_input(const float* e1, const float* seg1, const float* frame1,
                   const float* frame2, const float* seg2, const float* e2,
                   const float* frame3, const int64_t* seg3,
                   float* x10, float* frame3_out, int64_t* seg3_out,
                   int b, int H, int W, int flip, void* stream);

/* Autoregressive rollout, reference src/trainer.py:453-476 (8 steps from two frames + two segmentation maps).
 *   vlg_argmax_nchw    out[b,1,hw] = float(argmax over C of logits[b,C,hw]), first maximum wins     trainer.py:467
 *   vlg_rollout_input  x10 = cat[e_a, seg_a, img_a, img_b, seg_b, e_b] (frames already normalised): trainer.py:461 with
 *                      the two edge channels the 10-channel net was trained with (trainer.py:197) */
int vlg_argmax_nchw(const float* logits, float* out, int b, int C, int64_t hw, void* stream);
int vlg_rollout_input(const float* e_a, const float* seg_a, const float* img_a, const float* img_b,
                      const float* seg_b, const float* e_b, float* x10, int b, int64_t hw, void* stream);

/* ------------------------------------------------- GridNet convolution path (reference-real)
 * The reference's trainable model (CoordGridNet / GridNet, reference src/models/gridnet.py:7-114) is built
 * from three blocks (reference src/models/modules.py:5-58): PReLU -> conv3x3 -> PReLU -> conv3x3, the first
 * conv stride 2 in down blocks, a bilinear x2 (align_corners=True) in front of up blocks.  These entry
 * points replace nn.Conv2d(k=3, padding=1[, stride=2]) + nn.PReLU() + nn.Upsample at modules.py:12-17,
 * 34-39, 49-55 and AddCoords at modules.py:65-96, forward and backward.
 *
 * Tensors are "padded NHWC": rows p = (n*(H+2) + y')*(W+2) + x' of Cp floats (Cp = channels rounded up to
 * 32, extras zero), zero one-pixel halo, zero guard rows before/after (csrc/conv.hip header).  Weights are
 * [cout_p][9][cin_p] (tap = ky*3+kx), bias [cout_p].  rowmask[p] = 1 on interior pixels, 0 on the halo.
 * rowtab (stride 2 forward / weight gradient): output row -> input row of the window centre.
 * tap_tables (stride 2 data gradient): [9][tab_stride] input row -> output row through that tap, or a
 * guard row.  prelu_slope: device pointer to the single shared slope (NULL = no activation); the
 * activation is applied to channels < act_ch only, so appended AddCoords channels stay linear. */
#define VLG_CEPI_BIAS   1   /* (fwd always adds bias when bias != NULL)                          */
#define VLG_CEPI_RESID  2   /* out += resid[row, col]   (sum with the other grid branch)         */
#define VLG_CEPI_PRELU  4   /* out = prelu(out)         (activation applied by the producer)     */
#define VLG_CEPI_DPRELU 8   /* dgrad: din = acc * prelu'(x_in); slope-gradient partials -> da    */
#define VLG_CEPI_ACCUM  16  /* C += result              (tensor consumed by several blocks)      */
#define VLG_CEPI_CIN4   32  /* fwd: only input channels 0..3 carry data (the image layers of the frozen trunks, 3 channels in a
                               32-channel padded tensor): contract over (tap, 4 channels) - same result, 1/6 of the products.
                               Stride 1, no activation on load, no residual / PReLU epilogue (else VLG_ERR_SHAPE). */
int vlg_conv3x3_fwd(const float* in, const float* w, const float* bias, float* out, const float* resid,
                    const float* rowmask, const float* prelu_slope, const int* rowtab, int64_t rows_out,
                    int cin_p, int cout, int cout_p, int wp_in, int act_ch, int epilogue,
                    float* workspace /* NULL, or >= vlg_conv3x3_fwd_workspace() floats: enables split-K (all tiles at the coarse
                                        levels; elsewhere the tiles beyond the last full round of 256, see csrc/conv.hip) */,
                    int64_t workspace_capacity /* floats available at workspace (checked; too small for the tail plan = no tail
                                                  split, too small for the coarse-level split = VLG_ERR_SHAPE) */, void* stream);
/* K ranges the forward uses for EVERY tile when given a workspace (1 = no such split): coarse levels of the 256-512 channel trunks */
int vlg_conv3x3_fwd_splits(int64_t rows_out, int cin_p, int cout, int cout_p);
/* floats of workspace the forward can use for this shape (0 = none): splits * rows_out * cout_p, or the tail plan's partial tiles */
int64_t vlg_conv3x3_fwd_workspace(int64_t rows_out, int cin_p, int cout, int cout_p);
int vlg_conv3x3_dgrad_slabs(int64_t rows_in, int cin_p);   /* length of the slope-gradient partial vector */
int vlg_conv3x3_dgrad(const float* dout, const float* w, float* din, const float* x_in,
                      const float* rowmask_in, const float* prelu_slope, float* da_slab,
                      const int* tap_tables, int64_t tab_stride, int64_t rows_in, int cin_p, int cout_p,
                      int wp, int act_ch, int epilogue,
                      float* workspace /* NULL, or >= vlg_conv3x3_dgrad_workspace() floats (split-K as in the forward; used only
                                          when da_slab and tap_tables are NULL, i.e. by the frozen trunks) */,
                      int64_t workspace_capacity /* floats available at workspace (checked) */,
                      int da_capacity /* floats available at da_slab, >= vlg_conv3x3_dgrad_slabs() (checked) */, void* stream);
int vlg_conv3x3_dgrad_splits(int64_t rows_in, int cin_p, int cout_p);
int64_t vlg_conv3x3_dgrad_workspace(int64_t rows_in, int cin_p, int cout_p);
int vlg_conv3x3_wgrad_slabs(int64_t rows, int cin_p, int cout_p);
int vlg_conv3x3_wgrad(const float* dout, const float* in, float* slabs, int64_t slab_stride, int64_t slab_capacity,
                      const int* rowtab, const float* prelu_slope, int64_t rows, int cin_p, int cout_p,
                      int wp_in, int act_ch, void* stream);
/* (b,C,H,W) <-> padded NHWC; to_padded can append AddCoords' two channels (modules.py:65-96) at c0, c0+1 */
int vlg_nchw_to_padded(const float* src, float* dst, int b, int C, int H, int W, int cp, int coord_c0, void* stream);
int vlg_padded_to_nchw(const float* src, float* dst, int b, int C, int H, int W, int cp, void* stream);
int vlg_fill_coords(float* dst, int b, int H, int W, int cp, int coord_c0, void* stream);
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True), modules.py:50, on padded NHWC */
int vlg_upsample2x_fwd(const float* in, float* out, int b, int h, int w, int cp, void* stream);
int vlg_upsample2x_bwd(const float* dout, float* din, int b, int h, int w, int cp, int accumulate, void* stream);
/* Frozen HED edge detector, forward only (reference src/models/hned.py:9-105; used at trainer.py:190-192,216).
 * 3x3 convs reuse vlg_conv3x3_fwd with prelu_slope pointing at 0.0 (ReLU, applied by the consumer on load). */
int vlg_maxpool2x2(const float* in, float* out, int b, int h, int w, int cp, void* stream);            /* hned.py:20,28,38,48 */
int vlg_score1x1_relu(const float* in, const float* w, const float* bias, float* out, int b, int H, int W,
                      int C, int cp, void* stream);                                                    /* hned.py:60-64,90-94 */
int vlg_hed_head(const float* s0, const float* s1, const float* s2, const float* s3, const float* s4,
                 const float* combine_w, const float* combine_b, float* out /* (6,b,H,W) */, int b, int H, int W,
                 void* stream);                                                                        /* hned.py:96-105 */
/* VggLoss (reference src/loss.py:29-49): frozen VGG19 features[:27] on both images, L1 mean of the relu4_4 features.
 * The trunk reuses vlg_conv3x3_fwd / _dgrad (ReLU = slope 0) and vlg_maxpool2x2; these two close the loop. */
int vlg_maxpool2x2_bwd(const float* in, const float* dout, float* din, int b, int h, int w, int cp, void* stream);
int vlg_l1_relu_padded(const float* a, const float* b, float* da, float* loss, float* scratch /* >= 4096 */,
                       int64_t rows, int cp, int64_t count, float grad_scale, void* stream);
/* dst = src (+ dst): gradient hand-over along the residual sums of gridnet.py:51-56 */
int vlg_add_rows(float* dst, const float* src, int64_t n, int accumulate, void* stream);
/* sums a vector of per-block partials into dst[0] (PReLU slope gradients); accumulate != 0 adds to dst */
int vlg_sum_partials(const float* partials, int n, float* dst, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLG_HIP_H */
