/*
 * dfine_hip.h -- C ABI of libdfine_hip.so, the gfx950 (MI355X) kernels behind the
 * D-FINE training hot path.
 *
 * Conventions (every entry point):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the
 *     caller (PyTorch's caching allocator in practice) unless the argument is documented
 *     "host"; the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - returns 0 on success, a hipError_t value (>0) if a launch failed, or a negative
 *     DFINE_E_* code when the arguments are outside what the kernels were built for
 *     (nothing is launched in that case);
 *   - tensors are contiguous, row-major in the index order written in the comment.
 *
 * The reference (uc-vision/D-FINE-seg) is pure Python; its operator seams are Python
 * callables. Each entry point cites the reference callable (file:line under
 * /root/reference) whose ATen op sequence it replaces.
 */
#ifndef DFINE_HIP_H_
#define DFINE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFINE_ABI_VERSION 24

/* element types of `value`/`out` tensors */
#define DFINE_DT_F32 0
#define DFINE_DT_BF16 1

/* argument errors (negative so they never collide with hipError_t) */
#define DFINE_E_BADARG (-1)   /* null pointer / non-positive size            */
#define DFINE_E_UNSUPPORTED (-2) /* shape or dtype the kernels are not built for */

#define DFINE_MAX_LEVELS 4
#define DFINE_MAX_POINTS 32 /* sum of points over levels, per head */

int dfine_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * Multi-scale deformable attention core (decoder cross-attention gather).
 * Replaces deformable_attention_core_func_v2, method="default"
 *   (src/d_fine/arch/utils.py:191-264; bound at src/d_fine/arch/dfine_decoder.py:90-92,
 *    called at :174-176), i.e. per level F.grid_sample(bilinear, zeros,
 *    align_corners=False) + weighted sum over the points.
 *
 * Layout difference to the reference, on purpose: `value` is the decoder memory itself,
 * channel-last [B, S, nH, c] with S = sum_l H_l*W_l (level l starts at row sum_{k<l} H_k*W_k,
 * row-major (y, x) inside a level). The reference permutes it to (B*nH, c, H_l, W_l) per
 * level (dfine_decoder.py:416-426); here one bilinear corner of one head is one contiguous
 * c*sizeof(elem) segment.
 *
 *   value      [B, S, nH, c]            f32 or bf16 (value_dtype)
 *   loc        [B, Lq, nH, P, 2] f32    sampling locations (x, y) in [0,1] image coords,
 *                                       P = sum_l num_points[l], level-major along P
 *   attn       [B, Lq, nH, P]    f32    attention weights (already soft-maxed)
 *   out        [B, Lq, nH*c]            same dtype as value; fp32 accumulation inside
 *   level_hw   host int32 [n_levels][2] (H_l, W_l)
 *   num_points host int32 [n_levels]
 * Supported: c in {16, 32}; n_levels <= DFINE_MAX_LEVELS; P <= DFINE_MAX_POINTS.
  *   grad_value_dtype: DFINE_DT_F32 (float atomics into an f32 map) or, for bf16 values, DFINE_DT_BF16: the gradient map is
 *   bf16 and the scatter uses packed bf16 atomics (two channels per add) -- half the atomic bytes, which are the floor of
 *   this kernel, and the accumulation precision of the stock bf16 autocast path (grid_sample's backward).
 * ------------------------------------------------------------------------------------- */
int dfine_msda_fwd(const void* value, int value_dtype, const float* loc, const float* attn,
                   void* out, int B, int S, int Lq, int nH, int c, int n_levels,
                   const int32_t* level_hw, const int32_t* num_points, void* stream);

/* Backward of the above (autograd of the same reference lines).
 *   grad_out   [B, Lq, nH*c]            same dtype as value
 *   grad_value [B, S, nH, c]     f32    ACCUMULATED with float atomics: the caller zeroes
 *                                       it (or keeps accumulating several layers into it:
 *                                       all decoder layers share one `value`,
 *                                       dfine_decoder.py:449)
 *   grad_loc   [B, Lq, nH, P, 2] f32    overwritten
 *   grad_attn  [B, Lq, nH, P]    f32    overwritten
 */
int dfine_msda_bwd(const void* value, int value_dtype, const float* loc, const float* attn,
                   const void* grad_out, void* grad_value, int grad_value_dtype, float* grad_loc, float* grad_attn,
                   int B, int S, int Lq, int nH, int c, int n_levels, const int32_t* level_hw,
                   const int32_t* num_points, void* stream);

/* ---------------------------------------------------------------------------------------
 * The elementwise glue in front of the gather (reference src/d_fine/arch/dfine_decoder.py:148-166 MSDeformableAttention.forward):
 *     loc[tok][h][p]  = ref.xy + offsets[tok][h][p] * (point_scale[p] * 0.5) * ref.wh       (4-d reference boxes, cxcywh)
 *     weights[tok][h] = softmax over the P points of the head's attention logits
 * and its backward (d offsets, d logits from d loc, d weights), one launch each instead of ~7 + ~10 elementwise launches per
 * decoder layer. offsets [n_tokens, nH, P, 2] and logits [n_tokens, nH, P] bf16 or f32 (the projections' outputs), ref_boxes
 * f32 [n_tokens, 4], point_scale f32 [P]; loc f32 [n_tokens, nH, P, 2], weights f32 [n_tokens, nH, P]. No gradient for the
 * reference boxes (the decoder detaches them, dfine_decoder.py:476).
 * ------------------------------------------------------------------------------------- */
int dfine_msda_prepare_fwd(const void* offsets, const void* logits, int dtype, const float* ref_boxes, const float* point_scale,
                           float* loc, float* weights, long long n_tokens, int nH, int P, void* stream);
int dfine_msda_prepare_bwd(const float* grad_loc, const float* grad_weights, const float* weights, const float* ref_boxes,
                           const float* point_scale, int dtype, void* d_offsets, void* d_logits, long long n_tokens, int nH,
                           int P, void* stream);

/* ---------------------------------------------------------------------------------------
 * Hungarian matching without leaving the device. Replaces HungarianMatcher.forward
 * (src/d_fine/matcher.py:52-130): the cost matrix (:76-112), the device->host copy (:112) and
 * the per-image scipy.optimize.linear_sum_assignment calls (:116), for N = heads*B problems in
 * one launch each. Problem p uses image p % B.
 *
 *   logits      [N, Q, C]  f32     boxes     [N, Q, 4] f32 cxcywh
 *   tgt_labels  [B, G]     int64   tgt_boxes [B, G, 4] f32   num_gt [B] int32 (rows >= num_gt ignored)
 *   cost        [N, G, Q]  f32     workspace/output: cost[p][g][q], gt-major so one LSAP row is
 *                                  contiguous over the queries; rows g >= num_gt are not written
 * ------------------------------------------------------------------------------------- */
int dfine_match_cost(const float* logits, const float* boxes, const int64_t* tgt_labels,
                     const float* tgt_boxes, const int32_t* num_gt, float* cost, int N, int B,
                     int Q, int C, int G, float alpha, float gamma, float w_class, float w_bbox,
                     float w_giou, void* stream);

/* Linear sum assignment of every problem's [num_gt x Q] cost block (double precision inside,
 * same optimum AND same tie-breaking as scipy's rectangular_lsap). Needs num_gt <= Q.
 *   query_of_gt [N, G] int32   query assigned to ground truth g (-1 for g >= num_gt)
 *   gt_of_query [N, Q] int32   ground truth assigned to query q (-1 = unmatched)
 */
int dfine_lsap(const float* cost, const int32_t* num_gt, int32_t* query_of_gt,
               int32_t* gt_of_query, int N, int B, int G, int Q, void* stream);

/* Union of the H heads' matchings (DFINECriterion._get_go_indices,
 * src/d_fine/dfine_criterion.py:371-392), dense: go[b][q] = ground truth most often matched to
 * query q over the heads (ties -> smallest index), -1 if never matched.
 *   gt_of_query [H, B, Q] int32   ->   go [B, Q] int32
 */
int dfine_go_union(const int32_t* gt_of_query, int32_t* go, int H, int B, int Q, void* stream);

/* ---------------------------------------------------------------------------------------
 * Depthwise k x k convolution, channel-last. Replaces nn.Conv2d(C, C, k, stride, padding=(k-1)//2,
 * groups=C, bias=False) forward/backward as used by LightConvBNAct.conv2 and HG_Stage.downsample
 * (src/d_fine/arch/hgnetv2.py:83-112, 295-304) and SCDown.cv2 (hybrid_encoder.py:96-103).
 *
 *   x  [B, H, W, C]    y / dy [B, OH, OW, C]   bf16 or f32 (dtype), OH = (H + 2*pad - k)/stride + 1
 *   w  [C, k, k] f32   (nn.Conv2d weight [C,1,k,k], fp32 master copy)
 *   dw [C, k, k] f32   ACCUMULATED with float atomics: the caller zeroes it
 * Supported: k in {3,5}, stride in {1,2}, C % 8 == 0 (bf16) / C % 4 == 0 (f32).
  *   stats (forward, stride 1 only; or NULL): f64 [8][2][C], ACCUMULATED -- per channel the sum and the sum of squares of the
 *   rounded outputs in the 8-replica layout dfine_bn_act_fwd reads (as dfine_igemm_bf16 does for dense convs), so the
 *   BatchNorm behind a depthwise conv needs no statistics pass.
 * ------------------------------------------------------------------------------------- */
int dfine_dwconv_fwd(const void* x, const float* w, void* y, int dtype, int B, int H, int W, int C,
                     int k, int stride, double* stats, void* stream);
int dfine_dwconv_bwd_data(const void* dy, const float* w, void* dx, int dtype, int B, int H, int W,
                          int C, int k, int stride, void* stream);
/* workspace: f32 scratch of at least dfine_dwconv_bwd_weight_workspace(...) floats (per-block partial
 * sums, summed by a second kernel); the query returns -1 for unsupported shapes. */
long long dfine_dwconv_bwd_weight_workspace(int B, int H, int W, int C, int k, int stride);
int dfine_dwconv_bwd_weight(const void* x, const void* dy, float* dw, float* workspace,
                            long long workspace_floats, int dtype, int B, int H, int W, int C, int k,
                            int stride, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused BatchNorm2d (batch statistics) [+ addend] + activation [+ learnable affine], channel-last:
 *     z = (x - mean_c) * rstd_c * gamma_c + beta_c (+ addend);   y = lab_scale * act(z) + lab_bias
 * Replaces the norm -> act -> lab tail of ConvBNAct (src/d_fine/arch/hgnetv2.py:66-80),
 * ConvNormLayer(_fuse).forward (hybrid_encoder.py:39-45, 92-93) and the sum + activation of
 * VGGBlock.forward (hybrid_encoder.py:115-121; `addend` = the other branch), forward and backward.
 *
 *   x, addend, y, dy, dx, dadd   [n_pix, C] (= [B,H,W,C]) bf16 or f32 (dtype); addend/dadd/lab_* may be NULL
 *   gamma, beta, running_mean, running_var, save_mean, save_rstd   f32 [C]
 *   dgamma, dbeta f32 [C]: ACCUMULATED (+=; zeroed scratch or the parameters' fp32 .grad)
 *   lab_scale, lab_bias  f32 [1] each on the device (both or neither);  dlab_scale, dlab_bias f32 [1]
 *       each, ACCUMULATED (zeroed scratch or the parameters' fp32 .grad)
 *   sums f64 [8][2*C], red f64 [8][4*C]: workspaces (8 replicas against atomic contention), ZEROED by
 *       the caller before the call
 *   act: 0 identity, 1 ReLU, 2 SiLU.  training=1: batch statistics, running stats updated
 *   (momentum, unbiased variance); training=0 (forward only): running statistics.
 *   training = 1 | (R << 8), R > 1: SyncBatchNorm over R ranks (reference src/dl/train.py:159-160). The caller has replaced
 *       `sums` (forward, between phase 1 and phase 2) / `red` (backward, likewise) by their AVERAGE over the ranks, every rank
 *       holding n_pix pixels: mean, variance and the dx terms then come out as the global ones with no other change, the
 *       parameter gradients as the mean of the ranks' local ones (what DDP's averaging makes of torch's local sums), and the
 *       running variance is unbiased with the global count R * n_pix.
 *   training=0 in the backward: dx = gamma * rstd * dz with save_mean / save_rstd = the running statistics
 *       (FrozenBatchNorm2d, src/d_fine/arch/common.py:29-67, and eval-mode BN); phase 1 is then only needed for
 *       dgamma / dbeta / d lab.
 *   phase: each direction is two kernels (forward: statistics, apply; backward: reductions, dx).
 *       0 = launch both, 1 = only the first, 2 = only the second (so a caller can time them apart).
 * Any C (the widest channel vector <= 16 bytes that divides C is used).
 * Pitches (elements; 0 = C): `ldy` lets the forward write its output as a CHANNEL SLICE of a wider map (the concat buffer of
 *   an HG block / ELAN block: torch.cat disappears); in the backward the output gradient is read with pitch `ld_dy` and an
 *   optional second part `dy2` (pitch `ld_dy2`) is added on load -- a layer whose output feeds the next layer AND a concat
 *   receives (next layer's input gradient) + (a channel slice of the concat's gradient) without an add kernel or a copy.
 *   Pitches and bases must keep the channel vectors aligned (multiples of the vector width), else DFINE_E_UNSUPPORTED.
 * `post` (NULL or [n_pix, C] of the same dtype, pitch `ld_post`): added AFTER activation and LAB, y = lab(act(z)) + post --
 *   the residual of HG_Block.forward (hgnetv2.py:267-268, `x = x + identity`) without an add kernel. It does not enter the
 *   backward kernels: its gradient is the output gradient itself.
 * ------------------------------------------------------------------------------------- */
int dfine_bn_act_fwd(const void* x, const void* addend, void* y, int dtype, double* sums,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float* save_mean, float* save_rstd, const float* lab_scale, const float* lab_bias,
                     int n_pix, int C, int act, int training, float eps, float momentum, int phase,
                     int ldy, const void* post, int ld_post, void* stream);
int dfine_bn_act_bwd(const void* x, const void* addend, const void* dy, int ld_dy, const void* dy2, int ld_dy2,
                     int dtype, const float* save_mean, const float* save_rstd, const float* gamma,
                     const float* beta, const float* lab_scale, double* red, void* dx, void* dadd,
                     float* dgamma, float* dbeta, float* dlab_scale, float* dlab_bias, int n_pix, int C,
                     int act, int training, float eps, int phase, void* stream);

/* ---------------------------------------------------------------------------------------
 * Two training BatchNorms summed under one activation: y = act(bn_a(xa) + bn_b(xb)) -- RepVgg's training form
 * silu(norm(conv3x3(x)) + norm(conv1x1(x))) (reference src/d_fine/arch/hybrid_encoder.py:106-121), one launch forward and
 * two backward instead of the 4 + 4 of two dfine_bn_act_* calls chained through `addend` / `dadd` (the second branch's
 * output and the shared dz never travel through memory).
 *   xa, xb, y, dy, dy2, dxa, dxb: [n_pix, C] of dtype (bf16 / f32); y has pixel pitch ldy, dy / dy2 pitches ld_dy / ld_dy2
 *   (dy2 optional: a second part of the output gradient, summed on load). sums_a / sums_b: the fp64 statistics replicas
 *   [8][2][C] the producing convolutions' epilogues filled (dfine_igemm_bf16 `stats`); save_a / save_b f32 [2][C] receive
 *   (mean, rstd); running statistics are updated like nn.BatchNorm2d. No LAB, no addend.
 *   Backward: red = zeroed fp64 workspace [8][3][C]; dgamma / dbeta of both norms are ACCUMULATED (f32 [C] each).
 * ------------------------------------------------------------------------------------- */
int dfine_bn_pair_act_fwd(const void* xa, const void* xb, void* y, int dtype, const double* sums_a, const double* sums_b,
                          const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                          float* running_mean_a, float* running_var_a, float* running_mean_b, float* running_var_b,
                          float* save_a, float* save_b, int n_pix, int C, int act, int training, float eps, float momentum,
                          int ldy, void* stream);
int dfine_bn_pair_act_bwd(const void* xa, const void* xb, const void* dy, int ld_dy, const void* dy2, int ld_dy2, int dtype,
                          const float* save_a, const float* save_b, const float* gamma_a, const float* beta_a,
                          const float* gamma_b, const float* beta_b, double* red, void* dxa, void* dxb, float* dgamma_a,
                          float* dbeta_a, float* dgamma_b, float* dbeta_b, int n_pix, int C, int act, int training, float eps,
                          void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer tail over FLAT fp32 buffers of n elements (all trainable parameters, grouped).
 * Replaces clip_grad_norm_ + AdamW.step + zero_grad + ModelEMA.update of the reference's
 * optimizer_step (src/dl/train.py:488-511, 61-72; groups from src/d_fine/dfine.py:87-124).
 *
 * dfine_grad_sumsq:      out[0] += sum(grad^2)   (out f64, zeroed by the caller; grad 16-byte aligned)
 * dfine_adamw_ema_step:  g = grad * min(1, max_norm / (sqrt(sumsq) + 1e-6));  AdamW (decoupled weight
 *     decay, PyTorch semantics);  ema = decay*ema + (1-decay)*param;  bf16_shadow = bf16(param);
 *     grad = 0.   ema / bf16_shadow may be NULL.
 *   hyper f32 (device): [0] max_norm (<=0 off) [1] beta1 [2] beta2 [3] eps [4] 1-beta1^t [5] 1-beta2^t
 *                       [6] ema decay (<0 off)  [8+2g] lr of group g  [9+2g] weight decay of group g
 *   group_begin host int64 [n_groups]: first flat index of each group (group_begin[0] == 0), <= 8 groups
 * ------------------------------------------------------------------------------------- */
int dfine_grad_sumsq(const float* grad, long long n, double* out, void* stream);
int dfine_adamw_ema_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                         void* bf16_shadow, const float* hyper, const double* grad_sumsq,
                         const long long* group_begin, int n_groups, long long n, void* stream);

/* ---------------------------------------------------------------------------------------
 * All box/class/distribution losses of ONE prediction head and their gradients, in one launch.
 * Replaces DFINECriterion.loss_labels_vfl + loss_boxes + loss_local (FGL, DDF) for one head
 * (src/d_fine/dfine_criterion.py:92-237, with arch/utils.py:267-354 bbox2distance/translate_gt and
 * :12-51 IoU/GIoU) AND their autograd backward, on dense matchings (assign[b][q] = gt index or -1).
 *
 *   logits [B,Q,C] f32|bf16 (logits_dtype)   boxes [B,Q,4] f32 cxcywh
 *   corners [B,Q,4*n_bins] f32 or NULL (no FGL/DDF)   ref_points [B,Q,4] f32
 *   teacher_corners [B,Q,4*n_bins] f32 / teacher_logits [B,Q,C] (logits dtype), or both NULL (no DDF)
 *   own_assign [B,Q] int32: matching used by the varifocal loss; go_assign [B,Q] int32: by boxes/FGL/DDF
 *   tgt_labels [B,G] int64, tgt_boxes [B,G,4] f32, Wn [n_bins] f32 = W(n)
 *   scalars f32 [9] (device): 1/num_boxes (vfl), 1/num_boxes (boxes, FGL), DDF coefficient of a matched
 *       edge, of an unmatched edge, then the 5 loss weights (vfl, bbox, giou, fgl, ddf)
 *   terms f32 [64][8]: 64 replicas (workgroup b adds into replica b % 64: thousands of workgroups adding into the same
 *       five floats serialise) of the weighted loss_vfl, loss_bbox, loss_giou, loss_fgl, loss_ddf in columns 0..4 --
 *       ACCUMULATED (caller zeroes, and sums the replicas)
 *   d_logits [B,Q,C] (logits dtype), d_boxes_l1 / d_boxes_giou [B,Q,4], d_corners_fgl / d_corners_ddf
 *       [B,Q,4*n_bins] f32: gradient of the corresponding weighted term (overwritten)
  *   batch_strides (HOST pointer to 6 values, or NULL): elements between consecutive images of logits, boxes, corners,
 *   ref_points, teacher_corners, teacher_logits (0 = dense). The rows of one image must be dense; with a stride a head can
 *   be given the [:, :Nd] / [:, Nd:] part of a [B, Nd + Q, .] decoder output in place. Outputs are always dense.
 * ------------------------------------------------------------------------------------- */
int dfine_criterion_head(const void* logits, int logits_dtype, const float* boxes, const float* corners,
                         const float* ref_points, const float* teacher_corners, const void* teacher_logits,
                         const int32_t* own_assign, const int32_t* go_assign, const int64_t* tgt_labels,
                         const float* tgt_boxes, const float* Wn, const float* scalars, float* terms,
                         void* d_logits, float* d_boxes_l1, float* d_boxes_giou, float* d_corners_fgl,
                         float* d_corners_ddf, int B, int Q, int C, int G, int n_bins, float alpha, float gamma,
                         float reg_scale, float temperature, const long long* batch_strides, void* stream);

/* Every head of one criterion call in ONE launch: the arguments of dfine_criterion_head per head (a HOST array of n_heads
 * structs), all heads with logits of one dtype; `terms` as there (f32 [64][8], zeroed by the caller) and additionally
 * out f32 [5] = the replicas' sums (written by a second tiny launch). has_strides = 0: dense inputs. */
typedef struct dfine_criterion_head_args {
  const void* logits;
  const float* boxes;
  const float* corners;
  const float* ref_points;
  const float* teacher_corners;
  const void* teacher_logits;
  const int32_t* own_assign;
  const int32_t* go_assign;
  const int64_t* tgt_labels;
  const float* tgt_boxes;
  const float* Wn;
  const float* scalars;
  float* terms;
  void* d_logits;
  float* d_boxes_l1;
  float* d_boxes_giou;
  float* d_corners_fgl;
  float* d_corners_ddf;
  float* out;
  int B, Q, C, G, n_bins, has_strides;
  float alpha, gamma, reg_scale, temperature;
  long long batch_strides[6];
} dfine_criterion_head_args;
int dfine_criterion_heads(const dfine_criterion_head_args* heads /* host */, int n_heads, int logits_dtype, void* stream);

/* Backward of dfine_criterion_head's five terms (autograd of the weighted sums in DFINECriterion.forward,
 * src/d_fine/dfine_criterion.py:560-580): given the five terms' upstream scalar gradients g_* (device f32 [1] each; NULL = 0),
 *     out_logits = d_logits * g_vfl   out_boxes = d_boxes_l1 * g_l1 + d_boxes_giou * g_giou
 *     out_corners = d_corners_fgl * g_fgl + d_corners_ddf * g_ddf      (n_corners = 0: no corners; d_corners_ddf may be NULL)
 * in one launch (n_* = element counts; out_logits in the logits' dtype, g_vfl rounded to it first like the PyTorch product). */
int dfine_criterion_head_scale(const void* d_logits, int logits_dtype, const float* d_boxes_l1, const float* d_boxes_giou,
                               const float* d_corners_fgl, const float* d_corners_ddf, const float* g_vfl, const float* g_l1,
                               const float* g_giou, const float* g_fgl, const float* g_ddf, void* out_logits, float* out_boxes,
                               float* out_corners, long long n_logits, long long n_boxes, long long n_corners, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight/bias gradient of y = x @ W^T + b (autograd of F.linear / nn.Linear, used by every
 * projection and MLP of the decoder and the AIFI layer: src/d_fine/arch/dfine_decoder.py:33-46,
 * 181-271; hybrid_encoder.py:243-290), split over the token axis, bf16 MFMA, fp32 accumulation:
 *     dW[n][k] += sum_m dY[m][n] * X[m][k]          db[n] += sum_m dY[m][n]   (db may be NULL)
 *   dY [M, N] bf16 rows of pitch ldy (>= N), X [M, K] bf16 rows of pitch ldx (>= K); dW f32 rows of length K
 *   with leading dimension ldw (>= K), db f32 [N]. Both are ACCUMULATED with float atomics: point them at the
 *   fp32 gradient buffer itself (or at zeroed scratch). Any N, K, pitch and alignment (16-byte vector loads
 *   when everything is a multiple of 8 elements, narrower loads otherwise).
 * ------------------------------------------------------------------------------------- */
int dfine_linear_wgrad(const void* dY, const void* X, float* dW, float* db, int M, int N, int K, int ldw,
                       int ldy, int ldx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight gradient of a dense (groups = 1, no bias) 2-d convolution on channels_last maps (autograd
 * of the nn.Conv2d inside ConvBNAct, src/d_fine/arch/hgnetv2.py:35-80, and ConvNormLayer,
 * hybrid_encoder.py:21-93), same split-M MFMA kernel with gathered X rows:
 *     dW[n][ky][kx][c] += sum_{b,oy,ox} dY[b][oy][ox][n] * X[b][oy*stride-pad+ky][ox*stride-pad+kx][c]
 *   dY [B, OH, OW, CO] bf16 pixels of pitch ldy (>= CO), X [B, H, W, CI] bf16 pixels of pitch ldx (>= CI) (NHWC
 *   memory, possibly channel slices of wider maps); dW f32 [CO][KH][KW][CI] -- the memory order of a
 *   channels_last weight -- ACCUMULATED with float atomics (point it at the fp32 gradient or at zeroed scratch).
 *   `pad` is the top/left padding; the bottom/right padding is implied by OH/OW (taps beyond the map read zeros;
 *   the stem's F.pad(x, (0,1,0,1)) + 2x2 conv is pad = 0 with OH = H), the last output's first tap must start
 *   inside the map. Any channel counts (CI % 8 != 0, e.g. the 3-channel image: element-wise gather).
 *   With a swapped reading it is also the weight gradient of ConvTranspose2d(k = stride, no bias): pass the module's
 *   INPUT as dY and its output gradient as X (dW [CI][KH][KW][CO] = its channels_last weight's memory order).
 * ------------------------------------------------------------------------------------- */
int dfine_conv_wgrad(const void* dY, const void* X, float* dW, int B, int H, int W, int CI, int OH, int OW,
                     int CO, int KH, int KW, int stride, int pad, int ldy, int ldx, void* stream);

/* ---------------------------------------------------------------------------------------
 * The same weight gradients, DEFERRED and GROUPED: a backward pass queues one `dfine_wgrad_problem` per Linear / Conv2d
 * (the operands stay alive -- 288 GB of HBM make that cheap) and launches them together when the pass, or one stage of
 * it, is done. One launch per tile shape (<= 9) covers every problem: the grid is full whatever the size of a single problem
 * (the 20x20 / 40x40 maps and the 15 744-token decoder GEMMs do not fill 256 CUs on their own), each problem's row axis
 * is split into few long chunks (few float atomics), and ~190 launches with their tails become a handful.
 *   problem: `conv` = 0 -> dfine_linear_wgrad's arguments (M, N, K, ldw, ldy, ldx, db optional);
 *            `conv` = 1 -> dfine_conv_wgrad's (B, H, W, CI, OH, OW, CO, KH, KW, stride, pad, ldy, ldx; db must be NULL).
 *   16-byte friendly problems (N, K resp. CI, ldy, ldx multiples of 8; dY, X 16-byte aligned) run on the wide-load tile
 *   variants, other Linear problems (the 4- / 132-wide heads, K = 20) on a narrow-load variant; gathered convolutions that
 *   are not 16-byte friendly (the 3-channel stem) cannot be grouped -- dfine_wgrad_group_eligible() says which, those go
 *   through the single-problem entry points.
 *   dfine_wgrad_group_plan (host only, no GPU call): writes the launch table for `n` problems into `table` (host memory,
 *   `capacity` bytes) and returns the bytes used (table == NULL: the bytes needed); < 0 on error. The caller copies
 *   those bytes to device memory (before the launch EXECUTES: under hipGraph capture that may be after the capture) and
 *   passes both copies to dfine_wgrad_group_launch, which reads the launch list from the host copy.
 * ------------------------------------------------------------------------------------- */
typedef struct dfine_wgrad_problem {
  const void* dY;
  const void* X;
  float* dW;
  float* db;
  int conv;
  int M, N, K, ldw;                                      /* conv = 0 */
  int B, H, W, CI, OH, OW, CO, KH, KW, stride, pad;      /* conv = 1 */
  int ldy, ldx;
} dfine_wgrad_problem;
int dfine_wgrad_group_eligible(const dfine_wgrad_problem* problem);
long long dfine_wgrad_group_plan(const dfine_wgrad_problem* problems, int n, void* table, long long capacity);
int dfine_wgrad_group_launch(const void* table_host, const void* table_dev, void* stream);

/* ---------------------------------------------------------------------------------------
 * FDR head of one decoder layer (reference src/d_fine/arch/dfine_decoder.py:276-296 Integral,
 * :298-313 LQE, :487-509 where the decoder applies them; arch/utils.py:119-142 distance2bbox):
 *     p        = softmax(corners[row][side][:])                         side in {l, t, r, b}
 *     d[side]  = sum_j p_j * W[j]                                        F.linear(prob, project)
 *     boxes    = distance2bbox(ref_boxes[row], d, |reg_scale|)           cxcywh
 *     stat     = (top-k values of p in descending order, their mean)     prob.topk(k) ++ mean
 *   corners f32 [rows, 4, n_bins], ref_boxes f32 [rows, 4] (cxcywh), W f32 [n_bins]
 *   boxes f32 [rows, 4], stat f32 [rows, 4, k+1], top_idx u8 [rows, 4, k] (bins of the top-k, kept
 *   for the backward). k must be 4 (LQE(4, ...), :598); 4 <= n_bins <= 60 (a workgroup holds 256 rows of n_bins logits in LDS so that global accesses are coalesced).
 * Backward: g_boxes f32 [rows, 4] and g_stat f32 [rows, 4, k+1] (either may be NULL = zero)
 *   -> g_corners f32 [rows, 4, n_bins] (written, not accumulated). ref_boxes and W get no gradient
 *   (the decoder detaches the reference boxes; up / reg_scale are frozen Parameters, :592-593).
 * ------------------------------------------------------------------------------------- */
int dfine_fdr_head_fwd(const float* corners, const float* ref_boxes, const float* W, float* boxes, float* stat,
                       unsigned char* top_idx, long long rows, int n_bins, int k, float reg_scale, void* stream);
int dfine_fdr_head_bwd(const float* corners, const float* ref_boxes, const float* W, const unsigned char* top_idx,
                       const float* g_boxes, const float* g_stat, float* g_corners, long long rows, int n_bins,
                       int k, float reg_scale, void* stream);

/* ---------------------------------------------------------------------------------------
 * Stem pooling of HGNetv2 (reference src/d_fine/arch/hgnetv2.py:213-224):
 *     y = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), kernel_size=2, stride=1, ceil_mode=True)
 * on a channels_last map, without the padded copy: the map is read as zero-extended by one row and
 * one column at the bottom/right, so y keeps x's extent.
 *   x, y, gy, gx [B, H, W, C] bf16 or f32 (dtype); C % 8 == 0 (bf16) / C % 4 == 0 (f32).
 * Backward sends each output gradient to the first maximum of its window in scan order (ATen's
 * rule); gradients that fall on the zero extension are dropped. gx is written, not accumulated.
 * `ldy` / `ld_gy` (elements, 0 = C): pixel pitch of the forward's output / the backward's output gradient, so the pooled map
 * can be one channel slice of the stem's concat buffer (hgnetv2.py:222 torch.cat([x1, x2], dim=1) disappears).
 * ------------------------------------------------------------------------------------- */
int dfine_stem_pool_fwd(const void* x, int dtype, void* y, int B, int H, int W, int C, int ldy, void* stream);
int dfine_stem_pool_bwd(const void* x, const void* gy, int ld_gy, int dtype, void* gx, int B, int H, int W, int C,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * Nearest-neighbour x2 upsampling of the encoder's FPN (reference src/d_fine/arch/hybrid_encoder.py:377
 * F.interpolate(feat_high, scale_factor=2., mode="nearest")) on a channels_last map, forward and backward:
 *     y[b, 2h+i, 2w+j, :] = x[b, h, w, :]            gx[b, h, w, :] = sum_{i,j in {0,1}} gy[b, 2h+i, 2w+j, :]
 *   x, gx [B, H, W, C]; y, gy [B, 2H, 2W, C] with pixel pitch ldy / ld_gy (elements, 0 = C): the upsampled map is one half of the
 *   FPN block's concat buffer (hybrid_encoder.py:378 torch.cat([upsample_feat, feat_low], dim=1) disappears). C a multiple of
 *   the 16-byte vector; pitched operand 16-byte aligned.
 * ------------------------------------------------------------------------------------- */
int dfine_upsample2x_fwd(const void* x, int dtype, void* y, int B, int H, int W, int C, int ldy, void* stream);
int dfine_upsample2x_bwd(const void* gy, int ld_gy, int dtype, void* gx, int B, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * out = base + bilinear_resize(x) (align_corners = false, F.interpolate's source-index rule), channel-last maps: the top-down
 * sum of MaskPixelDecoder.forward (src/d_fine/arch/dfine_decoder.py:352-360, `x = x + F.interpolate(lat, size=..., mode="bilinear")`)
 * in one launch, and the resize's backward as a gather (every input pixel sums its weights over the output pixels that read it:
 * no atomics, bitwise reproducible). x [B, h, w, C], base (may be NULL) / y / gy [B, H, W, C], gx [B, h, w, C], bf16 or f32, dense;
 * H >= h, W >= w, ratios <= 8; C a multiple of the 16-byte channel vector. The base's gradient is gy itself.
 * ------------------------------------------------------------------------------------- */
int dfine_bilinear_add_fwd(const void* x, const void* base, int dtype, void* y, int B, int h, int w, int H, int W, int C,
                           void* stream);
int dfine_bilinear_bwd(const void* gy, int dtype, void* gx, int B, int h, int w, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Mask loss of the matched queries (reference src/d_fine/dfine_criterion.py:272-357: _focal_loss_mask,
 * _dice_loss, loss_masks). Per (image b, target slot m) row of mask logits z against the ground-truth
 * mask t of slot m (ground truth m % G: M = groups * G, the denoising groups share the targets):
 *     p = sigmoid(z), p_t = p t + (1-p)(1-t), a_t = alpha t + (1-alpha)(1-t), bce = softplus(z) - z t
 *     stats[b][m] = ( sum a_t (1-p_t)^2 bce ,  sum p t ,  sum p )            over the HW pixels
 *   logits [B, M, HW] f32 or bf16 (dtype), tgt f32 [B, G, HW] in [0,1], alpha f32 [B, G]
 *   (0.5 + 0.25 clamp(1 - 2 fg, -1, 1), fg = mean of t, :285-291), stats f32 [B, M, 3] (ACCUMULATED: zero it).
 * dfine_mask_loss_grad: g_stats f32 [B, M, 3] = d(loss)/d(stats)  ->  d_logits [B, M, HW] (logits' dtype,
 *   written). The means, the dice ratio and the average over matched pairs are the caller's.
 *   Needs HW % 8 == 0 (bf16) / HW % 4 == 0 (f32); grad: B*M <= 65535.
 * ------------------------------------------------------------------------------------- */
int dfine_mask_loss_stats(const void* logits, int dtype, const float* tgt, const float* alpha, float* stats, int B,
                          int M, int G, int HW, void* stream);
int dfine_mask_loss_grad(const void* logits, int dtype, const float* tgt, const float* alpha, const float* g_stats,
                         void* d_logits, int B, int M, int G, int HW, void* stream);

/* ---------------------------------------------------------------------------------------
 * The mask head's contraction FUSED with that loss: logits[b][n][px] = sum_c rows[b][n][c] * feat[b][px][c]
 * (reference src/d_fine/arch/dfine_decoder.py:937-940 `einsum('bqc,bchw->bqhw')`, consumed by
 * src/d_fine/dfine_criterion.py:272-357) as an MFMA GEMM (csrc/igemm2.hip: pixels x rows tiles, K = C) whose epilogue
 * reduces the row statistics above -- the [B, N, HW] logits exist only in accumulators, in either direction.
 *   feat bf16 [B*HW, C] (the channels-last mask feature map), rows bf16 [B, N, C] (the matched queries' mask embeddings,
 *   every head's rows concatenated; row n belongs to ground truth n % G), tgt f32 [B, G, HW], alpha f32 [B, G].
 * dfine_mask_rows_stats: stats f32 [B, N, 3] ACCUMULATED (zero it) -- the numbers dfine_mask_loss_stats gives on the
 *   bf16-rounded logits.
 * dfine_mask_rows_dz: recomputes the logits and writes d(loss)/d(logit) = f(logit, t, alpha, g_stats[b][n]) as bf16
 *   dz [B*HW, ldz >= N] (pixel-major: the operand both backward GEMMs read); columns >= N are not written.
 * dfine_mask_rows_dfeat: d_feat [B*HW, C] bf16 = dz[b] [HW, Npad] @ rows[b] [Npad, C], given rows_t = rows[b]^T as
 *   bf16 [B, C, ld_rt] (zero beyond N) and dz's columns N..Npad zeroed by the caller. (d_rows = dz[b]^T feat[b] is a
 *   weight-gradient-shaped GEMM: dfine_linear_wgrad / the grouped launch.)
 * Needs HW % 128 == 0, C % 8 == 0, Npad % 8 == 0, ldz % 8 == 0, 16-byte aligned bases; DFINE_E_UNSUPPORTED otherwise
 * (the caller then materialises the logits: dfine_igemm_bf16 / a library GEMM + dfine_mask_loss_*).
 * ------------------------------------------------------------------------------------- */
int dfine_mask_rows_stats(const void* feat, const void* rows, const float* tgt, const float* alpha, float* stats, int B,
                          int HW, int N, int G, int C, void* stream);
int dfine_mask_rows_dz(const void* feat, const void* rows, const float* tgt, const float* alpha, const float* g_stats,
                       void* dz, int ldz, int B, int HW, int N, int G, int C, void* stream);
int dfine_mask_rows_dfeat(const void* dz, int ldz, const void* rows_t, int ld_rt, void* d_feat, int B, int HW, int Npad,
                          int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Residual add + LayerNorm over the last dimension (post-norm blocks: reference
 * src/d_fine/arch/dfine_decoder.py:233-271 `norm1(tgt + attn)`, `norm3(clamp(tgt + ffn))`, Gate;
 * hybrid_encoder.py:274-290), in the activations' dtype, statistics in fp32:
 *     s = clamp(x + res, -clamp, clamp)   (res may be NULL; clamp <= 0: none)
 *     y = (s - mean(s)) * rsqrt(var(s) + eps) * gamma + beta
 *   x, res, y, s_out, dy, s, ds [M, D] bf16 or f32 (dtype); gamma, beta, dgamma, dbeta f32 [D];
 *   mean, rstd f32 [M]. s_out (may be NULL) receives s rounded to the dtype -- pass it (or x when
 *   res is NULL and clamp <= 0) to the backward as `s`; the statistics are taken on the rounded s.
 * Backward: ds = d(loss)/d(s) (= gradient of both x and res; 0 where the clamp saturated);
 *   dgamma / dbeta are ACCUMULATED with float atomics.
 * D in {128, 256, 384, 512} (512: bf16 only).
 * ------------------------------------------------------------------------------------- */
int dfine_add_layernorm_fwd(const void* x, const void* res, int dtype, const float* gamma, const float* beta, void* y,
                            void* s_out, float* mean, float* rstd, long long M, int D, float eps, float clamp,
                            void* stream);
int dfine_add_layernorm_bwd(const void* dy, const void* s, int dtype, const float* gamma, const float* mean,
                            const float* rstd, void* ds, float* dgamma, float* dbeta, long long M, int D, float clamp,
                            void* stream);

/* ---------------------------------------------------------------------------------------
 * Elementwise middle of the decoder layer's gateway (reference src/d_fine/arch/dfine_decoder.py:181-196 Gate:
 * gates = sigmoid(Linear(cat(a, b))); out = norm(gate1 * a + gate2 * b)), forward and backward, one launch each:
 *     mix[m][d] = sigmoid(pre[m][d]) * a[m][d] + sigmoid(pre[m][D + d]) * b[m][d]
 *   pre [M, 2D] (the Linear's output), a, b, mix, g (= d mix), da, db [M, D], dpre [M, 2D]; bf16 or f32, dense.
 *   Factors and products are rounded to the tensor dtype like the separate ops' outputs.
 * ------------------------------------------------------------------------------------- */
int dfine_gate_mix_fwd(const void* pre, const void* a, const void* b, int dtype, void* out, long long M, int D, void* stream);
int dfine_gate_mix_bwd(const void* pre, const void* a, const void* b, const void* g, int dtype, void* dpre, void* da, void* db,
                       long long M, int D, void* stream);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / Linear, forward and input gradient, bf16 MFMA with fp32 accumulation:
 *     C[m][n] = act( sum_k A(m, k) * Bw[n][k] + bias[n] )
 * Replaces the ATen/MIOpen `convolution` + `convolution_backward(input)` behind every dense nn.Conv2d of
 * the path (ConvBNAct, reference src/d_fine/arch/hgnetv2.py:35-80; ConvNormLayer_fuse / VGGBlock / CSPLayer /
 * RepNCSPELAN4 / SCDown.cv1, hybrid_encoder.py:21-45,106-121,182-239; MaskPixelDecoder incl. its
 * ConvTranspose2d, dfine_decoder.py:316-376) and the `addmm`/`mm` behind every nn.Linear forward and input
 * gradient (MLP dfine_decoder.py:33-46, MSDeformableAttention projections :119-178, decoder layer
 * :233-271, AIFI hybrid_encoder.py:243-290).
 *
 * gather == 0 ("plain"): A(m, k) = A[m*lda + k]; output row m at C + m*ldc.
 * gather == 1: rows are the pixels of a [images, PH, PW] grid, m = (b*PH + py)*PW + px; K = ntaps*CI and
 *     A(m, tap*CI + c) = A[((b*IH + py*isy + tap_dy[tap])*IW + px*isx + tap_dx[tap])*lda + c]   (0 outside the map)
 *     Bw column of (tap, c) = tap_w[tap]*CI + c                                               (row pitch ldb)
 *     output row m at C + ((b*OHf + py*osy + ooy)*OWf + px*osx + oox)*ldc
 *   conv forward (stride s, padding p): taps (ky, kx) -> dy = ky - p, dx = kx - p, tap_w = ky*KW + kx, isy = isx = s,
 *   Bw = the channels_last weight [CO][KH][KW][CI]; input gradient of a stride-1 conv: dy = p - ky, same tap_w, Bw =
 *   [CI][KH][KW][CO] (dfine_transpose_weights); a strided conv's input gradient is one launch per parity class of input
 *   pixels (osy/ooy select the class, only the taps of that class are listed); transposed 2x2 stride-2 conv: four
 *   1-tap launches with osy = osx = 2 and (ooy, oox) = the tap.
 *   A, Bw, C bf16; bias f32 [N] or NULL; act 0 = none, 1 = ReLU; act | 0x400: C += result (bf16 sum of the stored C and the
 *   rounded result -- what autograd's accumulation of two input gradients computes, without its add kernel; no `stats`).
 *   stats (or NULL): f64 [8][2][N], ACCUMULATED: per output channel the sum and the sum of squares of the bf16-rounded
 *   outputs, spread over 8 replicas -- the layout dfine_bn_act_fwd reads, so a BatchNorm after the conv skips its
 *   statistics pass. Any alignment is accepted (16-byte vector path when CI (gather) / K (plain), lda, ldb are
 *   multiples of 8 and the bases are 16-byte aligned; 8-byte and element-wise paths otherwise).
 *   Two kernel bodies serve it, bit-identical in their results: plain rows on the 16-byte path run on an LDS-DMA ring
 *   (csrc/igemm2.hip: `buffer_load ... lds`, counted s_waitcnt, hardware range check instead of zero-fill branches), gathers
 *   and the narrow paths on the register-staged body (csrc/igemm.hip). DFINE_IGEMM_V=1|2 forces one body (measurements).
 * ------------------------------------------------------------------------------------- */
#define DFINE_IGEMM_MAX_TAPS 16
typedef struct dfine_igemm_geom {
  int M, N, K;
  int lda, ldb, ldc;
  int gather;
  int PH, PW, IH, IW, CI, ntaps, isy, isx;
  int OHf, OWf, osy, osx, ooy, oox;
  int act;
  signed char tap_dy[DFINE_IGEMM_MAX_TAPS], tap_dx[DFINE_IGEMM_MAX_TAPS], tap_w[DFINE_IGEMM_MAX_TAPS];
} dfine_igemm_geom;
int dfine_igemm_bf16(const void* A, const void* Bw, void* C, const float* bias, double* stats,
                     const dfine_igemm_geom* geom /* host */, void* stream);

/* The same GEMM as an INPUT-GRADIENT launch whose output is d(loss)/d(y) for y = lab(act(bn(x))), the previous layer's fused
 * norm (reference hgnetv2.py:35-80, hybrid_encoder.py:21-45): the epilogue also reads the norm's input x (and the optional
 * second gradient part dy2 of a y that was written into a concat slice) and adds that norm's backward sums
 *     red[replica][0..3][n] += ( sum dz, sum dz*xhat, sum dy*act(z), sum dy ),   dz = (dy + dy2) * lab_scale * act'(z)
 * (the pass dfine_bn_act_bwd phase 1 makes) -- dfine_bn_act_bwd then runs with phase = 2 only. With geom->act & 0x400
 * (C += result) the sums are taken over the finished sum. x bf16 [rows of the output map, ldx], dy2 bf16 [.., ld_dy2] or NULL,
 * mean / rstd / gamma / beta f32 [N], lab_scale f32 [1] or NULL (then entries 2, 3 are untouched), red = zeroed fp64 [8][4][N].
 * Needs the 16-byte paths (as dfine_igemm_bf16) and N, ldx, ld_dy2 multiples of 8; DFINE_E_UNSUPPORTED otherwise (the
 * caller then makes the plain launch + the reduction pass). */
typedef struct dfine_igemm_bnred {
  const void* x;
  const void* dy2;
  const float *mean, *rstd, *gamma, *beta, *lab_scale;
  double* red;
  int ldx, ld_dy2, act;
} dfine_igemm_bnred;
int dfine_igemm_bf16_bnred(const void* A, const void* Bw, void* C, const dfine_igemm_geom* geom /* host */,
                           const dfine_igemm_bnred* bnred /* host */, void* stream);

/* The same GEMM as the INPUT-GRADIENT launch of a Linear whose input x [M, N] (bf16, row pitch ld_mask) is the output of a ReLU
 * fused into the previous Linear (MLP, reference src/d_fine/arch/dfine_decoder.py:33-46; decoder FFN :233-271): C = (A Bw^T)
 * where x > 0, 0 elsewhere -- relu's backward (aten threshold_backward) in the epilogue instead of a pass over [M, N].
 * Plain rows on the 16-byte paths only (N, ld_mask multiples of 8, aligned bases, no `C +=`); DFINE_E_UNSUPPORTED otherwise. */
int dfine_igemm_bf16_relumask(const void* A, const void* Bw, void* C, const dfine_igemm_geom* geom /* host */, const void* mask,
                              int ld_mask, void* stream);

/* ---------------------------------------------------------------------------------------
 * Transposed bf16 copies of weights for the input-gradient GEMMs: for each of n weights
 *     dst[c][t][k] = src[k][t][c]      (src [K][T][C] -> dst [C][T][K]; a Linear weight has T = 1)
 * in ONE launch (the optimizer step refreshes every weight's transposed shadow after the update;
 * reference: what autograd's mm(grad, weight) / convolution_backward read implicitly).
 *   table: int64 [n][5] device = (src element offset, dst element offset, K, T, C); blocks: int32 [n_blocks][2]
 *   device = (weight index, tile index) for every 32x32 tile (tile index runs over T * ceil(K/32) * ceil(C/32)).
 * ------------------------------------------------------------------------------------- */
int dfine_transpose_weights(const void* src /* bf16 */, void* dst /* bf16 */, const long long* table,
                            const int* blocks, int n_blocks, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight gradient of nn.Embedding (the denoising class embedding, reference dfine_decoder.py:434 via
 * arch/utils.py:357-467): dW[idx[m]][:] += g[m][:] for idx[m] != padding_idx, as fp32 atomics straight into
 * the (flat) fp32 gradient. Replaces ATen's embedding_dense_backward (a sort + unique-by-key + segment pipeline
 * of ~10 launches whose scratch is sized by data read back from the device).
 *   idx int64 [M], g bf16 or f32 (dtype) [M, D], dW f32 [rows, D] ACCUMULATED; indices outside [0, rows) are skipped.
 * ------------------------------------------------------------------------------------- */
int dfine_embedding_wgrad(const long long* idx, const void* g, int dtype, float* dW, long long M, int D, int rows,
                          int padding_idx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Multi-head self-attention  O = softmax(scale * Q K^T + mask) V  per (image, head), forward and backward.
 * Replaces nn.MultiheadAttention's attention core at both sites of the path: decoder self-attention with the
 * denoising mask (reference src/d_fine/arch/dfine_decoder.py:233-240, need_weights slow path materialising
 * [B, 8, Lq, Lq]) and the AIFI layer (hybrid_encoder.py:243-290).
 *   q, k, v: bf16 row-major [B*L, >= H*dh] with row pitches ldq / ldk / ldv (q and k are typically the two halves of
 *   ONE [B*L, 2D] projection output: k = q + D, ldq = ldk = 2D); head h = columns [h*dh, (h+1)*dh). 16-byte
 *   aligned bases, pitches % 8 == 0, dh in {8..64} and a multiple of 8.
 *   blocked: u8 [L, L] (row = query, nonzero = may NOT attend), shared by all images and heads, or NULL.
 *   o: bf16 [B*L, ...] pitch ldo (written); lse: f32 [B, H, L] = log2-domain log-sum-exp of the scaled scores
 *   (-inf for a row with every key blocked, whose output is 0).
 * Backward: dout bf16 pitch lddo -> dq, dk, dv bf16 with their own pitches (dq / dk may be the halves of one
 *   [B*L, 2D] buffer), all written; dsum f32 [B, H, L] is scratch (rowsum(dO * O)). No atomics: results are
 *   bitwise reproducible.
  *   bits_q / bits_k (optional, both or neither in the backward; used only together with `blocked`): the SAME mask packed
 *   into 64-bit words, [L][ceil(L/64)] each -- bits_q[q][t] bit j = blocked[q][64t + j], bits_k[k][t] bit j =
 *   blocked[64t + j][k] (bits beyond L are ignored). With them a lane reads one word per 64-row tile instead of 16 single
 *   bytes; the byte path doubles the time of a masked launch (tools/time_attention.py).
 * ------------------------------------------------------------------------------------- */
int dfine_attention_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                        const unsigned char* blocked, const unsigned long long* bits_q, void* o, int ldo, float* lse,
                        int B, int H, int L, int dh, float scale, void* stream);
int dfine_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                        const unsigned char* blocked, const unsigned long long* bits_q, const unsigned long long* bits_k,
                        const void* o, int ldo, const float* lse, const void* dout, int lddo, void* dq, int lddq,
                        void* dk, int lddk, void* dv, int lddv, float* dsum, int B, int H, int L, int dh, float scale,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * Evaluation post-processing (reference src/dl/train.py:227-319 Trainer.preds_postprocess with
 * use_focal_loss=True; helpers src/dl/utils.py:160-183, 623-672, 715-789).
 * dfine_postprocess_topk: per image the K best of the Q*C sigmoid scores, descending (ties: lower flat index
 *   first), with label = idx % C, query = idx / C and that query's box mapped to the ORIGINAL image: cxcywh in [0,1]
 *   -> absolute xyxy on the network input (floor / ceil, clamped to [1, size-1]: norm_xywh_to_abs_xyxy) -> original
 *   size (scale_boxes, or scale_boxes_ratio_kept with the letterbox padding removed and clipping when keep_ratio).
 *   logits [B, Q, C] f32 or bf16 (dtype), boxes f32 [B, Q, 4], orig_sizes int64 [B, 2] (h, w) ->
 *   scores f32 [B, K], labels int64 [B, K], qidx int64 [B, K], out_boxes f32 [B, K, 4]. K <= 512.
 * dfine_postprocess_masks (one image): probs [Q, Hm, Wm] f32 or bf16 = that image's sigmoid mask maps, qidx int64
 *   [n_inst] the kept queries, boxes f32 [n_inst, 4] their xyxy boxes in original-image pixels -> out u8
 *   [n_inst, H0, W0]: process_masks' two bilinear resizes in fp16 (mask grid -> network input, letterbox crop when
 *   keep_ratio, -> original size), clamp, >= thresh, and cleanup_masks' "zero outside the box".
 * ------------------------------------------------------------------------------------- */
int dfine_postprocess_topk(const void* logits, int dtype, const float* boxes, const long long* orig_sizes, float* scores,
                           long long* labels, long long* qidx, float* out_boxes, int B, int Q, int C, int K, int proc_h,
                           int proc_w, int keep_ratio, void* stream);
int dfine_postprocess_masks(const void* probs, int dtype, const long long* qidx, const float* boxes, unsigned char* out,
                            int n_inst, int Hm, int Wm, int proc_h, int proc_w, int H0, int W0, int keep_ratio, float thresh,
                            void* stream);

/* ---------------------------------------------------------------------------------------
 * Mask-loss targets from a packed batch of ground-truth masks (input staging, reference
 * src/d_fine/dfine_criterion.py:239-270 _prepare_target_masks: per image m.float() -> F.interpolate(size=(Hm, Wm),
 * mode="nearest") -> clamp(0, 1)): masks u8 [n_maps, H, W] (n_maps = B*G padded slots, zero maps for padding)
 * -> out f32 [n_maps, Hm, Wm] in {0, 1}; source pixel = min(floor(dst * float(in)/float(out)), in - 1), bit-exact
 * with PyTorch's nearest rule. (Hv, Wv) (0, 0 = H, W): the size the batch was resized to by the multiscale augmentation
 * BEFORE the criterion (dataset.py:663-668 resizes the masks with mode="nearest" too): the two nearest maps are composed.
 * ------------------------------------------------------------------------------------- */
int dfine_mask_targets_nearest(const unsigned char* masks, float* out, long long n_maps, int H, int W, int Hv, int Wv, int Hm,
                               int Wm, void* stream);

/* ---------------------------------------------------------------------------------------
 * Image staging (the step before the path): what the reference's data loader does on the host per image --
 * A.Normalize(mean, std) + ToTensorV2 (src/dl/dataset.py:128-131), stacking, and the collate function's multiscale
 * F.interpolate(images, size=(OH, OW), mode="bilinear", align_corners=False) (dataset.py:645-670) -- as ONE kernel over the
 * packed uint8 batch:  src u8 [B, H, W, 3] (decoded pixels, HWC)  ->  dst f32 or bf16 [B, OH, OW, 3] (NHWC = the
 * channels_last memory of a [B, 3, OH, OW] tensor, which is what the stem's implicit GEMM reads).
 *   v = (u8 - mean_c*255) * (1 / (std_c*255)) in float32 (albumentations' arithmetic), then PyTorch's bilinear rule when
 *   (OH, OW) != (H, W).  mean, std: HOST pointers to 3 floats each.
 * ------------------------------------------------------------------------------------- */
int dfine_stage_images(const unsigned char* src, void* dst, int dtype, int B, int H, int W, int OH, int OW,
                       const float* mean, const float* std, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row maximum, out[m] = max_c x[m][c] (query selection: reference src/d_fine/arch/dfine_decoder.py:905
 * enc_outputs_logits.max(-1).values, whose top-k picks the decoder queries): x [M, C] bf16 or f32 dense, out f32 [M].
 * Not differentiable (the reference takes indices from it only).
 * ------------------------------------------------------------------------------------- */
int dfine_row_max(const void* x, int dtype, float* out, long long M, int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * Deformable-attention backward split so that the value gradient needs no global atomics:
 * dfine_msda_bwd_points = dfine_msda_bwd without its grad_value scatter (grad_loc, grad_attn only);
 * dfine_msda_bwd_value: d(loss)/d(value) of n_calls deformable-attention calls that sampled the SAME value
 *   tensor (every decoder layer does, dfine_decoder.py:449) in one launch: loc[i] f32 [B, Lq, nH, P, 2], attn[i] f32
 *   [B, Lq, nH, P], grad_out[i] [B, Lq, nH*c] in value's dtype -- host arrays of n_calls device pointers (<= 8) --
 *   -> grad_value f32 [B, S, nH, c], WRITTEN (every element exactly once: no zero fill, nothing accumulated).
 *   Workgroups own slabs of one level's pixels for one (image, head) in LDS and add with LDS atomics.
 *   DFINE_E_UNSUPPORTED when a single pixel row of a level exceeds the 96 KB slab (use dfine_msda_bwd then).
 * ------------------------------------------------------------------------------------- */
int dfine_msda_bwd_points(const void* value, int value_dtype, const float* loc, const float* attn, const void* grad_out,
                          float* grad_loc, float* grad_attn, int B, int S, int Lq, int nH, int c, int n_levels,
                          const int32_t* level_hw, const int32_t* num_points, void* stream);
int dfine_msda_bwd_value(int value_dtype, const void* const* loc /* host */, const void* const* attn /* host */,
                         const void* const* grad_out /* host */, int n_calls, float* grad_value, int B, int S, int Lq, int nH,
                         int c, int n_levels, const int32_t* level_hw, const int32_t* num_points, void* stream);

/* ---------------------------------------------------------------------------------------
 * dfine_msda_bwd_value_sorted: the same d(loss)/d(value) of n_calls (<= 8) deformable-attention calls on one value tensor
 * (reference src/d_fine/arch/utils.py:191-264, backward of the grid_sample gather; all decoder layers sample the same
 * memory, dfine_decoder.py:449) WITHOUT float atomics: one workgroup per (image, head[, token range]) counting-sorts that
 * (image, head)'s (call, query, point, corner) hits by token in LDS, writes 8-byte records into its own segment of `entries`
 * and then sums every token's list in fp32 registers from the grad_out rows. grad_value [B, S, nH, c], f32 or bf16 (grad_dtype:
 * the sum rounded once), is WRITTEN, every element exactly once (no zero fill). `entries`: 8-byte records, entries_cap >= dfine_msda_sorted_records(n_calls, B, S, Lq, nH, P)
 * (P = points per head over all levels), contents irrelevant on entry. The order inside a list follows the LDS atomics' arrival
 * order, so the fp32 sums are reproducible to rounding, like the scatter's.
 * ------------------------------------------------------------------------------------- */
long long dfine_msda_sorted_records(int n_calls, int B, int S, int Lq, int nH, int P);
int dfine_msda_bwd_value_sorted(int value_dtype, const void* const* loc /* host */, const void* const* attn /* host */,
                                const void* const* grad_out /* host */, int n_calls, void* grad_value, int grad_dtype, void* entries,
                                long long entries_cap, int B, int S, int Lq, int nH, int c, int n_levels, const int32_t* level_hw,
                                const int32_t* num_points, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFINE_HIP_H_ */
