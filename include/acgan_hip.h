/*
 * acgan_hip.h - C ABI of libacgan_hip.so: the MI355X (gfx950) kernels behind the
 * conv generator / discriminator forward+backward hot path of
 * yidingjiang/action_conditioned_GANs.
 *
 * The reference has no native boundary of its own: all hot-path arithmetic is TensorFlow-1.0
 * ops invoked from Python.  Each entry point below therefore cites the
 * reference call site(s) whose TF op it replaces (paths into /root/reference).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no framework types.
 *   - Device pointers are BORROWED from the caller (contiguous, NHWC, 16-byte aligned); the
 *     library never allocates: scratch comes in through (workspace, workspace_bytes), sized
 *     by the matching *_workspace_bytes() query.
 *   - Asynchronous on `stream` (a hipStream_t passed as void*); no internal synchronisation,
 *     safe to capture into a hipGraph.
 *   - Return 0 (ACG_OK) on success, an ACG_ERR_* code otherwise; the message is available
 *     from acg_last_error() (thread-local).  No global mutable state besides that.
 *   - `dtype` is the STORAGE type of the activation-class tensors of a call (x, y, dy, dx ...): ACG_F32, or
 *     ACG_BF16 = bfloat16 in memory (BASELINE configs 3 and 5).  bf16 activations are stored at the channel pitch
 *     round8(C) with zero pad channels, so that every 16-byte unit is 8 channels of one pixel.  The conv entry
 *     points then contract on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, float32 accumulation) and take
 *     their filter operand from the bf16 copies made by acg_weights_prepare_bf16; weight gradients, BatchNorm
 *     statistics, losses and optimizer state stay float32.  Where a call has two activation tensors of different
 *     type, `dtype` carries both: ACG_DTYPE2(first, second), documented per entry (plain ACG_F32 / ACG_BF16 mean
 *     "both").
 */
#ifndef ACGAN_HIP_H
#define ACGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature, a struct layout or the meaning of an argument changes; the Python binding
 * (_lib.Library) refuses a library whose acg_version() differs from the value it was written against.
 * 1: round 1.  2: round 2 (acg_bn_act_*, acg_bias_act_*, acg_dna_*, acg_copy_list, the flags of acg_*_bwd_pair changed
 * without a bump - any "version 1" build may be either).  3: round 3.  4: the `layout` argument of the slab hand-off
 * entries (acg_*_slabs, acg_bn_act_*_slabs), acg_bn_slabs_layout.  5: round 4 - struct acg_conv_desc is 17 int32 fields
 * (dgrad_c, adj_dgrad_c were appended in round 3 under version 4: a "version 4" build may have either layout).  6: the
 * workspace of acg_bn_act_fwd / acg_bn_act_bwd is state (zero before first use, private to its call site); new entries
 * acg_conv2d_tile, acg_opt_step_prepare_bf16.  7: struct acg_reduce_list carries `step_inc` (the deferred reduction launch also
 * advances an optimizer's device step counter).  8: round 5 - `flags` argument of acg_bn_act_fwd / _bwd / _fwd_slabs / _bwd_slabs and
 * acg_bn_slabs_layout (ACG_BN_NO_GRID_EXCHANGE); acg_bn_workspace_bytes covers the exchange area of the one-launch kernels for any
 * channel count; new entry acg_bn_exchange_selftest. */
#define ACG_ABI_VERSION 8

typedef void* acg_stream_t; /* hipStream_t */

enum { ACG_OK = 0, ACG_ERR_INVALID_ARG = 1, ACG_ERR_WORKSPACE = 2, ACG_ERR_LAUNCH = 3, ACG_ERR_UNSUPPORTED = 4 };
enum { ACG_F32 = 0, ACG_BF16 = 1 };
#define ACG_DTYPE2(first, second) ((first) | ((second) << 4) | 0x100)   /* two storage types in one dtype argument */
enum { ACG_ACT_NONE = 0, ACG_ACT_RELU = 1, ACG_ACT_LRELU = 2, ACG_ACT_TANH = 3 };
enum { ACG_CONV_FWD = 0, ACG_CONV_DGRAD = 1, ACG_CONV_WGRAD = 2 };

int32_t acg_version(void);          /* ACG_ABI_VERSION of the loaded library */
const char* acg_build_info(void);   /* e.g. "hip gfx950" */
const char* acg_last_error(void);   /* message of the calling thread's last failing call */

/* ------------------------------------------------------------------------------------------
 * Convolution.  One descriptor serves a conv and its adjoint:
 *   x [batch,in_h,in_w,in_c]  *  w [kh,kw,in_c,out_c] (HWIO)  ->  y [batch,out_h,out_w,out_c]
 *   y[b,p,q,o] = sum_{i,j,c} x[b, p*stride_h - pad_top + i, q*stride_w - pad_left + j, c] * w[i,j,c,o]
 * with zero padding; pad_top/pad_left are TF's pad_before (SAME: total//2).
 * ---------------------------------------------------------------------------------------- */
typedef struct acg_conv_desc {
  int32_t batch;
  int32_t in_h, in_w, in_c;
  int32_t out_h, out_w, out_c;
  int32_t kh, kw;
  int32_t stride_h, stride_w;
  int32_t pad_top, pad_left;
  int32_t in_pitch; /* channel pitch (floats) of x / dx in memory; 0 = dense (= in_c).  A pitch that is a
                       multiple of 4 lets 3- or 6-channel inputs (g/conv1, d/conv1) take the 16-byte gather path;
                       the pad channels of x must hold finite values (zeros), those of dx are left untouched. */
  int32_t out_pitch; /* same for y / dy (0 = dense = out_c): the 138- and 266-channel action-concatenated maps feeding
                        d/conv3 and g/tconv1 are stored at a pitch of 140 / 268.  Pad channels of dy must be zero. */
  int32_t dgrad_c;   /* acg_conv2d_dgrad, acg_conv2d_dgrad_slabs, acg_conv2d_bwd_pair: only the first dgrad_c channels of dx
                        are computed and written (0 = all in_c).  The last channels of those action-concatenated maps are tiled
                        inputs (train.py:48-50): nothing reads their gradient, and 138 columns cost a third 64-column tile. */
  int32_t adj_dgrad_c; /* the same on the ADJOINT descriptor of a transposed layer: acg_deconv2d_dgrad, _dgrad_slabs and
                          acg_deconv2d_bwd_pair compute the first adj_dgrad_c of the out_c channels of dx (0 = all).  Leave 0
                          on descriptors handed to acg_conv2d_fwd. */
} acg_conv_desc;

/* Fill a descriptor from slim-style arguments; same_padding != 0 -> TF 'SAME', else 'VALID'. */
int32_t acg_conv_desc_init(acg_conv_desc* d, int32_t batch, int32_t in_h, int32_t in_w, int32_t in_c,
                           int32_t kh, int32_t kw, int32_t out_c, int32_t stride, int32_t same_padding);

size_t acg_conv2d_workspace_bytes(const acg_conv_desc* d, int32_t which /* ACG_CONV_* */, int32_t dtype);

/* bf16 operand copies of the float32 master filters (dtype = ACG_BF16 conv entries).  A filter [kh,kw,A,B] - HWIO of
 * a conv layer, TF's [kh,kw,Cout,Cin] of a conv2d_transpose layer - gets two copies, zero padded to multiples of 8:
 *   rm [kh*kw][A][round8(B)]   the operand of acg_conv2d_dgrad and acg_deconv2d_fwd
 *   tr [kh*kw][B][round8(A)]   the operand of acg_conv2d_fwd and acg_deconv2d_dgrad
 * (each is k-fast for its contraction: one 16-byte load = 8 consecutive reduction indices).  All filters of a network
 * in ONE launch, behind the optimizer update of train.py:100-102. */
#define ACG_PREP_MAX 32
typedef struct acg_prep_list {
  const void* src[ACG_PREP_MAX]; /* float32 [taps][a][b] */
  void* rm[ACG_PREP_MAX];
  void* tr[ACG_PREP_MAX];
  int32_t taps[ACG_PREP_MAX], a[ACG_PREP_MAX], b[ACG_PREP_MAX];
} acg_prep_list;
int32_t acg_weights_prepare_bf16(const acg_prep_list* list, int32_t count, acg_stream_t stream);
/* The optimizer update of a whole scope (train.py:100-102; the formulas of acg_adam_step / acg_rmsprop_step, element by
 * element: bit-identical parameters and slots) AND the refresh of the scope's bf16 filter copies in ONE launch: every
 * `list->src[i]` is a filter inside the flat buffer `param` (16-byte aligned offset, no overlap); its elements are updated by
 * the blocks that write its two copies, everything else in [0, n) (beta, biases) by blocks of its own.  kind 0 = Adam
 * (slot1 = m, slot2 = v, step_dev = the device step counter, already incremented), 1 = RMSProp (slot1 = ms; slot2, step_dev
 * unused). */
typedef struct acg_opt_args {
  int32_t kind;
  float lr, beta1_or_decay, beta2, eps, grad_scale;
  int32_t use_clip;
  float clip_lo, clip_hi;
} acg_opt_args;
int32_t acg_opt_step_prepare_bf16(float* param, const float* grad, float* slot1, float* slot2, const int32_t* step_dev, int64_t n,
                                  const acg_opt_args* args, const acg_prep_list* list, int32_t count, acg_stream_t stream);

#ifdef ACG_TUNING
/* Tuning builds only (libacgan_hip_tuning.so, `make tuning`; absent from libacgan_hip.so, which has no process-wide
 * mutable state): force the tile configuration (fp32: 2 = 128x32, 3 = 64x64; bf16: 1 = 128x128, 3 = 64x64) and/or the
 * split-K factor chosen by the planner; -1 restores the heuristic.  Affects acg_conv2d_workspace_bytes too. */
int32_t acg_debug_conv_plan(int32_t cfg, int32_t splits);
#endif

/* slim.conv2d's tf.nn.conv2d: models.py:12-15,34-37,42-51,82-88.
 * dtype ACG_DTYPE2(ACG_BF16, ACG_F32) (acg_conv2d_fwd / acg_deconv2d_fwd): bf16 operands, y stored as float32 at the
 * bf16 tensor's channel pitch round8(out_c) - a head layer whose BatchNorm has no activation (d/conv6, models.py:87-88):
 * its BatchNorm backward is a difference of nearly equal terms, and bf16 rounding of its 1-channel input moved the
 * whole discriminator gradient by 4-11 % (round 2's test_epilogue_statistics tolerance). */
int32_t acg_conv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* d, int32_t dtype,
                       void* workspace, size_t workspace_bytes, acg_stream_t stream);
/* its gradient w.r.t. x (what tf.gradients emits for train.py:100-102).  A stride-2 layer with even input extents and at
 * most 8 input channels (d/conv1's 6-channel frame pair; the transposed forward of a 3-channel output) runs as ONE stride-1
 * contraction over the union window of its four stride-parity classes, N = 4 * in_c columns, with a small derived filter
 * built in the workspace in front of it (acg_conv2d_workspace_bytes accounts for it, so the workspace is needed even when
 * the contraction is not split). */
int32_t acg_conv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* d, int32_t dtype,
                         void* workspace, size_t workspace_bytes, acg_stream_t stream);
/* its gradient w.r.t. w:  dw = accumulate * dw + grad  (dw is always float32). */
int32_t acg_conv2d_wgrad(const void* x, const void* dy, float* dw, float accumulate, const acg_conv_desc* d,
                         int32_t dtype, void* workspace, size_t workspace_bytes, acg_stream_t stream);

/* slim.conv2d_transpose's tf.nn.conv2d_transpose: models.py:17-21,39-40,53-59.
 * The descriptor describes the ADJOINT conv: in_* is the deconv OUTPUT, out_* the deconv INPUT,
 * and w [kh,kw,in_c,out_c] is exactly TF's deconv filter layout [kh,kw,Cout,Cin].
 *   deconv2d_fwd   == conv2d_dgrad (x plays dy),  deconv2d_dgrad == conv2d_fwd,
 *   deconv2d_wgrad == conv2d_wgrad with the roles of x and dy exchanged. */
int32_t acg_deconv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype,
                         void* workspace, size_t workspace_bytes, acg_stream_t stream);
int32_t acg_deconv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* adj, int32_t dtype,
                           void* workspace, size_t workspace_bytes, acg_stream_t stream);
int32_t acg_deconv2d_wgrad(const void* x, const void* dy, float* dw, float accumulate, const acg_conv_desc* adj,
                           int32_t dtype, void* workspace, size_t workspace_bytes, acg_stream_t stream);

/* Deferred reduction of split weight gradients.  Nothing reads a weight gradient before the optimizer update of
 * train.py:100-102 (or, data parallel, the all-reduce of its bucket), so the per-layer slab reductions that
 * acg_conv2d_wgrad would launch one by one can run as ONE launch per step:
 *   acg_conv2d_splits          the split-K factor the planner picks (which = ACG_CONV_FWD/DGRAD/WGRAD); 0 on a bad desc
 *   acg_(de)conv2d_wgrad_slabs the contraction only: `splits` partial slabs of kh*kw*in_c*out_c floats are left in the
 *                              workspace (acg_conv2d_workspace_bytes); an error when the shape is not split
 *   acg_splitk_reduce_many     out[i] = accumulate[i] * out[i] + sum_z slabs[i][z], z in order: bit-identical to the
 *                              per-layer reduction.  Outputs must be distinct.  `step_inc` (may be NULL): the launch also
 *                              does *step_inc += 1 - the step counter of the optimizer it runs in front of (acg_step_inc
 *                              without a launch of its own: the update kernel behind it reads the incremented value). */
#define ACG_REDUCE_MAX 32
typedef struct acg_reduce_list {
  const void* slabs[ACG_REDUCE_MAX];
  void* out[ACG_REDUCE_MAX];
  int64_t numel[ACG_REDUCE_MAX];
  int32_t splits[ACG_REDUCE_MAX];
  float accumulate[ACG_REDUCE_MAX];
  int32_t* step_inc;
} acg_reduce_list;
int32_t acg_conv2d_splits(const acg_conv_desc* d, int32_t which, int32_t dtype);
/* The output tile (GEMM rows x columns) the planner runs contraction `which` of this layer on as a launch of its own, for
 * tests and tools: returns the number of tiles (all stride classes of an input gradient), 0 on a bad descriptor. */
int32_t acg_conv2d_tile(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t* tile_rows, int32_t* tile_cols);
int32_t acg_conv2d_wgrad_slabs(const void* x, const void* dy, const acg_conv_desc* d, int32_t dtype, void* workspace,
                               size_t workspace_bytes, acg_stream_t stream);
int32_t acg_deconv2d_wgrad_slabs(const void* x, const void* dy, const acg_conv_desc* adj, int32_t dtype, void* workspace,
                                 size_t workspace_bytes, acg_stream_t stream);
int32_t acg_splitk_reduce_many(const acg_reduce_list* list, int32_t count, acg_stream_t stream);

/* BatchNorm statistics out of the producing convolution (models.py:10-15,31-44,80-87: every conv / conv2d_transpose but
 * three feeds slim.batch_norm, whose first pass re-reads the whole activation for its per-channel mean and variance).
 * acg_(de)conv2d_fwd_stats are acg_(de)conv2d_fwd that also leave, per row tile ("block") of the output, the per-channel
 * SUM of the values as stored and their sum of squared deviations from the block's own mean (M2):
 *   partials[((g * nblk + b) * 2 + {0: sum, 1: M2}) * out_channels + c],  g < groups, b < nblk
 * (never a plain sum of squares: E[x^2] - E[x]^2 in float32 loses the variance once |mean| >> std).  Blocks are merged
 * with the parallel-variance formula, which needs their row counts: nblk = acg_conv2d_stats_layout(desc, which, dtype,
 * groups, &block_rows, &run_rows) - block b of a group covers min(block_rows, run_rows - (b % ceil(run_rows / block_rows))
 * * block_rows) rows (a "run" = the rows of one group, or of one stride class of a transposed layer).
 * which = ACG_CONV_FWD for a conv layer, ACG_CONV_DGRAD on the adjoint descriptor for a transposed layer; nblk = 0: this
 * shape cannot - it is split over K, a tile would straddle two groups, or the stride classes of a transposed layer differ
 * in size: use the plain entry and acg_bn_act_fwd.  acg_conv2d_stats_blocks returns nblk alone.
 * acg_bn_act_fwd_partials (below) consumes them: BatchNorm + activation in ONE launch. */
int32_t acg_conv2d_stats_blocks(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t groups);
int32_t acg_conv2d_stats_layout(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t groups, int32_t* block_rows,
                                int32_t* run_rows);
int32_t acg_conv2d_fwd_stats(const void* x, const void* w, void* y, const acg_conv_desc* d, int32_t dtype, void* workspace,
                             size_t workspace_bytes, float* partials, int32_t groups, acg_stream_t stream);
int32_t acg_deconv2d_fwd_stats(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype, void* workspace,
                               size_t workspace_bytes, float* partials, int32_t groups, acg_stream_t stream);

/* Bias + activation in the epilogue of a transposed layer's forward: y = act(conv2d_transpose(x, w) + bias), y float32 at
 * the pitch adj->in_pitch (0 = dense) whatever dtype the operands have - slim.conv2d_transpose(..., normalizer_fn=None,
 * activation_fn=tf.tanh) of models.py:20-21, the plain generator's frame, without the separate bias pass over the frame.
 * acg_deconv2d_fwd_bias_act_ok: 1 when the planner runs this shape unsplit on its 128x32 tile (at most 32 output channels,
 * 16-byte gathers), else use acg_deconv2d_fwd + acg_bias_act_fwd.  Backward is acg_bias_act_bwd on y as before. */
int32_t acg_deconv2d_fwd_bias_act_ok(const acg_conv_desc* adj, int32_t dtype);
int32_t acg_deconv2d_fwd_bias_act(const void* x, const void* w, const float* bias, float* y, const acg_conv_desc* adj, int32_t act,
                                  float leak, int32_t dtype, acg_stream_t stream);

/* Split-K hand-off to the consuming BatchNorm.  A small layer is split over K to fill the chip and would need a
 * launch of its own to sum the partial slabs; its output (forward) or input gradient (backward) is read next by the
 * layer's BatchNorm kernel (models.py:10-15: every conv but three is followed by batch_norm), which can sum the slabs as it
 * loads them: acg_(de)conv2d_fwd_slabs / _dgrad_slabs run the contraction only - `splits` float32 slabs stay in the
 * workspace (an error when acg_conv2d_splits == 1) - and acg_bn_act_fwd_slabs / acg_bn_act_bwd_slabs (below) take them.
 * Same values as the separate reduction, one launch less per layer and pass.
 * `layout` of a slab (rows = all pixels of the tensor, pitch = its channel pitch; a slab is rows * pitch floats either way):
 *   ACG_SLABS_ROWS   [rows][pitch], like the tensor
 *   ACG_SLABS_QUADS  [ceil(channels / 4)][rows][4]: the layout the one-launch BatchNorm kernels (four channels per block,
 *                    every row) read as consecutive 16-byte rows - ask acg_bn_slabs_layout which one the consumer takes. */
#define ACG_SLABS_ROWS 0
#define ACG_SLABS_QUADS 1
/* Bit (1 << layout) set for every layout this contraction (ACG_CONV_FWD / ACG_CONV_DGRAD; the transposed entries on their
 * adjoint descriptor) can leave its slabs in; 0 when it is not split.  (The merged input gradient of a stride-2 layer with
 * at most 8 input channels - see acg_conv2d_dgrad - leaves rows only.) */
int32_t acg_conv2d_slab_layouts(const acg_conv_desc* d, int32_t which, int32_t dtype);
int32_t acg_conv2d_fwd_slabs(const void* x, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* workspace,
                             size_t workspace_bytes, acg_stream_t stream);
int32_t acg_conv2d_dgrad_slabs(const void* dy, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* workspace,
                               size_t workspace_bytes, acg_stream_t stream);
int32_t acg_deconv2d_fwd_slabs(const void* x, const void* w, const acg_conv_desc* adj, int32_t dtype, int32_t layout, void* workspace,
                               size_t workspace_bytes, acg_stream_t stream);
int32_t acg_deconv2d_dgrad_slabs(const void* dy, const void* w, const acg_conv_desc* adj, int32_t dtype, int32_t layout, void* workspace,
                                 size_t workspace_bytes, acg_stream_t stream);

/* A layer's input gradient and weight gradient in ONE launch (both consume dy, neither reads the other's result):
 * the same results as acg_(de)conv2d_dgrad followed by acg_(de)conv2d_wgrad - or, with wgrad_slabs_only != 0, by
 * acg_(de)conv2d_wgrad_slabs (dw may then be NULL) - bit for bit; the blocks of the two contractions share the CUs
 * instead of running one grid after the other.  Workspaces as for the separate entries (acg_conv2d_workspace_bytes
 * with ACG_CONV_DGRAD / ACG_CONV_WGRAD; for the transposed layer ACG_CONV_FWD / ACG_CONV_WGRAD on the adjoint).
 * wgrad_slabs_only is a bit set: 1 = leave the weight-gradient slabs (dw may be NULL), 2 = leave the input-gradient
 * slabs for acg_bn_act_bwd_slabs (dx may be NULL; an error when that contraction is not split), 4 = those input-gradient
 * slabs in the ACG_SLABS_QUADS layout. */
int32_t acg_conv2d_bwd_pair(const void* dy, const void* w, const void* x, void* dx, float* dw, float dw_accumulate,
                            const acg_conv_desc* d, int32_t dtype, void* ws_dgrad, size_t ws_dgrad_bytes, void* ws_wgrad,
                            size_t ws_wgrad_bytes, int32_t wgrad_slabs_only, acg_stream_t stream);
int32_t acg_deconv2d_bwd_pair(const void* dy, const void* w, const void* x, void* dx, float* dw, float dw_accumulate,
                              const acg_conv_desc* adj, int32_t dtype, void* ws_dgrad, size_t ws_dgrad_bytes, void* ws_wgrad,
                              size_t ws_wgrad_bytes, int32_t wgrad_slabs_only, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Synchronised BatchNorm for data parallel runs (SURVEY 8(e) caveat 1; optional, no reference counterpart: the
 * reference is single-device).  The statistics are those of the GLOBAL batch: each direction is two calls with
 * one collective between them, issued by the caller.
 *   fwd:  acg_bn_moments            moments[g][0][c] = mean, moments[g][1][c] = biased variance of THIS rank's rows
 *         (caller: all-gather, combine into global moments - equal row counts per rank)
 *         acg_bn_act_fwd_moments    y = act((x - mean) * rsqrt(var + eps) + beta) with the GIVEN moments; saves mean, rstd
 *   bwd:  acg_bn_bwd_sums           sums[g][0][c] = sum dp, sums[g][1][c] = sum dp * xhat over this rank's rows
 *         (caller: keeps a copy as local_sums, all-reduces sums)
 *         acg_bn_act_bwd_sums       dx with the global sums and total_rows (global rows per group);
 *                                   dbeta = dbeta_acc * dbeta + sum_g local_sums[g][0][c]  (this rank's share; the
 *                                   gradient all-reduce averages it like every other parameter gradient)
 * Storage types and pitches as in acg_bn_act_fwd / acg_bn_act_bwd (round 3: bf16 networks - BASELINE config 3 - can run
 * the N-rank == 1-rank validation mode): acg_bn_moments takes the plain storage type of x; the other three the dtype of
 * the plain entry they stand in for, ACG_DTYPE2(ACG_F32, ACG_BF16) for the float32 head of a bf16 network included.
 * ---------------------------------------------------------------------------------------- */
int32_t acg_bn_moments(const void* x, float* moments, int64_t rows, int32_t channels, int32_t x_pitch, int32_t groups, int32_t dtype,
                       void* workspace, size_t workspace_bytes, acg_stream_t stream);
int32_t acg_bn_act_fwd_moments(const void* x, const float* beta, const float* moments, void* y, float* save_mean,
                               float* save_rstd, int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups,
                               float eps, int32_t act, float leak, int32_t dtype, acg_stream_t stream);
int32_t acg_bn_bwd_sums(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd,
                        float* sums, int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act,
                        float leak, int32_t dtype, void* workspace, size_t workspace_bytes, acg_stream_t stream);
int32_t acg_bn_act_bwd_sums(const void* x, const void* dy, const float* beta, const float* save_mean,
                            const float* save_rstd, const float* sums, const float* local_sums, int64_t total_rows,
                            void* dx, float* dbeta, float dbeta_acc, int64_t rows, int32_t channels, int32_t x_pitch,
                            int32_t y_pitch, int32_t groups, int32_t act, float leak, int32_t dtype, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * slim.batch_norm (training mode, scale=False, center=True) fused with the layer activation:
 * implicit via argscope at models.py:10-11,31-32,80-81; lrelu is ops.py:22-26.
 * x is viewed as [rows, channels]; `groups` splits the rows into equal contiguous chunks that
 * are normalised independently (groups=2 lets D(fake) and D(real), train.py:63-70, share one
 * launch while keeping separate batch statistics).  save_mean/save_rstd: [groups*channels].
 *   y = act((x - mean) * rsqrt(var + eps) + beta),  var biased.
 * bwd:  dbeta = dbeta_accumulate * dbeta + sum(dpre),  dx through mean and variance.
 * dtype: storage of x / dx, or ACG_DTYPE2(x, y) with y / dy the second type (ACG_DTYPE2(ACG_BF16, ACG_F32): a
 * loss-facing layer of a bf16 network keeps float32 logits).  acg_bn_act_bwd only: ACG_DTYPE2(ACG_F32, ACG_BF16) = x and
 * dy float32, dx bf16 - the head layer whose conv output is kept in float32 (acg_conv2d_fwd) hands its gradient back
 * into the bf16 network.  Statistics, beta and dbeta are float32.
 * x / dx rows are x_pitch elements apart, y / dy rows y_pitch (0 = dense = channels): the one-channel bf16 output of
 * d/conv6 sits at a pitch of 8; pad channels are neither read nor written.
 * WORKSPACE of acg_bn_act_fwd / acg_bn_act_bwd (version 6): for tensors whose grid is resident on the chip these run as ONE
 * launch whose blocks exchange their partial sums through the workspace (epoch-tagged 8-byte words; bn.hip).  The epoch counter
 * lives in the workspace and survives from call to call - also across the replays of a captured graph - so the workspace must be
 * ZERO before the first call that uses it, must not be written by anyone else between calls, and must belong to ONE call site
 * (one layer, one direction): hand every BatchNorm op its own, as the other entries' workspaces may be shared scratch but this
 * one is state.  The first 16 bytes are that state on every path (the two-launch kernels keep their partial sums behind them).
 * Word 2 (uint32) is set to 1 if a block ever gave up waiting for its peers (it then finishes with what it has:
 * a wrong result and this flag, never a hung GPU); it stays 0 in correct operation, and a caller should look at it at a point
 * where it synchronises anyway (the Python host: Runtime.check_exchange_flags - train() at every log interval and before every
 * checkpoint, bench.py before it prints its result, Session.close; the tests after every call).
 * `flags`: 0, or ACG_BN_NO_GRID_EXCHANGE - never take the one-launch grid kernels (register-resident or two-launch kernels
 * instead; acg_bn_slabs_layout answers accordingly).  Those kernels need every block of their grid resident at once (at most one
 * 1024-thread block per CU).  Kernels that finish on their own may run beside them (a convolution or an RCCL collective on a
 * second stream only delay the last blocks); ANOTHER grid-exchange kernel must not - two partially resident grids starve each
 * other until both time out - so a caller whose BatchNorm launches can overlap each other (two streams) passes this flag.
 * acg_bn_exchange_selftest: the in-launch exchange on its own - `blocks` (<= 512) blocks of `threads` (256 | 1024) publish b + 1
 * and gather the grid's sum into out[b] (float[blocks]; n (n + 1) / 2 everywhere, -1 if the 64 slots disagree).  `withhold` >= 0:
 * that block publishes nothing, so every block runs into `spin_limit` polls, sets word 2 of the workspace and finishes with a
 * short sum - the failure path, for tests (workspace: 16 + 512 * blocks bytes, zero before first use like any other).
 * ---------------------------------------------------------------------------------------- */
#define ACG_BN_NO_GRID_EXCHANGE 1
size_t acg_bn_workspace_bytes(int64_t rows, int32_t channels, int32_t groups);
int32_t acg_bn_exchange_selftest(void* workspace, size_t workspace_bytes, float* out, int32_t blocks, int32_t threads,
                                 int32_t withhold, uint32_t spin_limit, acg_stream_t stream);
/* The same with the split-K hand-off described at acg_conv2d_fwd_slabs: forward reads x as the sum of `splits` float32
 * slabs (each rows * x_pitch floats in `layout`, summed in slab order and rounded to x's storage type - what the separate
 * reduction would have stored) and WRITES x, which backward re-reads; backward reads dy as the sum of `splits` slabs
 * (each rows * y_pitch floats) and stores it nowhere.
 * acg_bn_slabs_layout: the slab layout this BatchNorm wants from its producer.  Round 4: ACG_SLABS_ROWS wherever the one-launch
 * GRID kernels run (channels and pitches multiples of 4, a plain ACG_F32 / ACG_BF16 dtype, the grid resident on the chip: every
 * layer of the reference's models up to 131072 x 128) - they read whole rows of every slab; ACG_SLABS_QUADS where only the
 * register-resident kernels with four channels per block apply (rows per group <= 2048); -1 when this BatchNorm cannot take
 * slabs at all (run the plain reduction + acg_bn_act_fwd / _bwd).  acg_bn_bwd_slabs_ok: the backward half of the resident test. */
int32_t acg_bn_slabs_layout(int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t dtype,
                            int32_t backward, int32_t flags);
int32_t acg_bn_act_fwd_slabs(const float* slabs, int32_t splits, void* x, const float* beta, void* y, float* save_mean,
                             float* save_rstd, int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups,
                             float eps, int32_t act, float leak, int32_t dtype, int32_t layout, int32_t flags, void* workspace,
                             size_t workspace_bytes, acg_stream_t stream);
int32_t acg_bn_bwd_slabs_ok(int64_t rows, int32_t groups);
int32_t acg_bn_act_bwd_slabs(const void* x, const float* dy_slabs, int32_t splits, const float* beta, const float* save_mean,
                             const float* save_rstd, void* dx, float* dbeta, float dbeta_accumulate, int64_t rows,
                             int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act, float leak,
                             int32_t dtype, int32_t layout, int32_t flags, void* workspace, size_t workspace_bytes, acg_stream_t stream);
/* BatchNorm + activation from the per-block (sum, M2) partials of acg_(de)conv2d_fwd_stats (`nblk` blocks per group, with
 * the block_rows / run_rows of acg_conv2d_stats_layout): mean = sum of sums / rows_per_group, variance = (sum of M2 +
 * sum_b n_b * (mean_b - mean)^2) / rows_per_group (merged about the first block's mean), then the apply pass of acg_bn_act_fwd.  More
 * than 512 blocks per group are merged by a small launch of their own first.  No workspace. */
int32_t acg_bn_act_fwd_partials(const void* x, const float* beta, const float* partials, int32_t nblk, int32_t block_rows,
                                int32_t run_rows, void* y, float* save_mean, float* save_rstd, int64_t rows, int32_t channels,
                                int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps, int32_t act, float leak, int32_t dtype,
                                acg_stream_t stream);
int32_t acg_bn_act_fwd(const void* x, const float* beta, void* y, float* save_mean, float* save_rstd,
                       int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps,
                       int32_t act, float leak, int32_t dtype, int32_t flags, void* workspace, size_t workspace_bytes,
                       acg_stream_t stream);
int32_t acg_bn_act_bwd(const void* x, const void* dy, const float* beta, const float* save_mean,
                       const float* save_rstd, void* dx, float* dbeta, float dbeta_accumulate,
                       int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act,
                       float leak, int32_t dtype, int32_t flags, void* workspace, size_t workspace_bytes, acg_stream_t stream);

/* Layers built with normalizer_fn=None: y = act(x + bias)   (models.py:20-21,44-51,54-59).
 * bwd takes the forward OUTPUT y; dx may be NULL when act == ACG_ACT_NONE and x, y share type and pitch (dx == dy).
 * bias == NULL (fwd) / dbias == NULL (bwd) give the plain activation (ops.py:22-26 lrelu called on its own).
 * x / dx rows are x_pitch elements apart, y / dy rows y_pitch (0 = dense = channels).  dtype = storage of x / dx, or
 * ACG_DTYPE2(x, y): a bf16 conv output (pitch round8) becoming the float32 dense frame or state the losses read. */
size_t acg_bias_workspace_bytes(int64_t rows, int32_t channels);
int32_t acg_bias_act_fwd(const void* x, const float* bias, void* y, int64_t rows, int32_t channels, int32_t x_pitch,
                         int32_t y_pitch, int32_t act, float leak, int32_t dtype, acg_stream_t stream);
int32_t acg_bias_act_bwd(const void* y, const void* dy, void* dx, float* dbias, float dbias_accumulate,
                         int64_t rows, int32_t channels, int32_t x_pitch, int32_t y_pitch, int32_t act, float leak,
                         int32_t dtype, void* workspace, size_t workspace_bytes, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dynamic Neural Advection tail, models.py:60-72 (softmax + extract_image_patches + mul + sum):
 *   out[b,y,x,c] = sum_{i,j} softmax(logits[b,y,x,:])[i*k+j] * image[b, y-p+i, x-p+j, c],
 *   p = (k-1)/2, zero outside the image.   logits [B,H,W,k*k], image/out [B,H,W,C], C <= 4.
 * bwd produces dlogits only (the image is a network input, train.py:53-54).
 * `bias` (float32 [k*k], may be NULL) folds in the bias of the layer that produces the logits - g/tconv4 has neither
 * BatchNorm nor activation (models.py:54-59) - i.e. the kernels take softmax(logits + bias) and backward also yields
 *   dbias = dbias_accumulate * dbias + sum over pixels of dlogits      (dbias may be NULL; workspace needed otherwise).
 * dtype: storage of logits / dlogits; ACG_BF16 rows are round8(k*k) elements apart (pad taps are not written).
 * image, out and dout are float32.
 * out2 / dout2 (may be NULL): the frame's second home.  train.py:63-66 feeds D concat(current frame, generated frame):
 * forward ALSO writes the frame into channels [out2_offset, out2_offset + c) of out2 - a tensor of out2_dtype (ACG_F32 /
 * ACG_BF16) whose pixels are out2_pitch elements apart, i.e. the discriminator's input - and backward adds the gradient
 * that comes back through those channels of dout2 (same addressing) to dout: no concat, slice or add launch.  When out2 is
 * exactly that concatenation at a pitch of 8 (c == 3, out2_offset == 3, out2_pitch == 8) forward writes the WHOLE pixel -
 * the image channels and two zero pad channels included - as one vector store; any other layout: the frame channels only.
 * ---------------------------------------------------------------------------------------- */
size_t acg_dna_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t ksize);
int32_t acg_dna_fwd(const void* logits, const float* bias, const void* image, void* out, void* out2, int32_t out2_pitch,
                    int32_t out2_offset, int32_t out2_dtype, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ksize,
                    int32_t dtype, acg_stream_t stream);
int32_t acg_dna_bwd(const void* logits, const float* bias, const void* image, const void* dout, const void* dout2,
                    int32_t dout2_pitch, int32_t dout2_offset, int32_t dout2_dtype, void* dlogits, float* dbias,
                    float dbias_accumulate, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ksize, int32_t dtype,
                    void* workspace, size_t workspace_bytes, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CDNA transformation (reference ops.py:52-98, the unused sibling of the DNA tail).
 *   params [B, k*k*M]: the raw kernel parameters of `masks` = M kernels per sample (the output of the reference's
 *   `cdna_params` fully-connected layer), read as [B,k,k,1,M].
 *   n[b,u,v,m] = (relu(p - relu_shift) + relu_shift) / sum_{u,v}(...)                      (ops.py:79-81)
 *   depthwise SAME correlation of image[b] (C colours) with n[b,:,:,m]; the [B,H,W,C*M] result (channel q = c*M+m,
 *   tf.nn.depthwise_conv2d) is split into M pieces of C channels along the channel axis (ops.py:94-96), so piece j,
 *   channel i is q = j*C + i, i.e. colour q/M under mask q%M - reproduced as written.
 *   out [M,B,H,W,C] (the list of M images); kern_norm [B,k*k*M] receives n (input of bwd; may be NULL in fwd).
 * bwd: dparams [B,k*k*M] (through normalisation and relu), dimage [B,H,W,C] or NULL.  k in {3,5,7}, C<=4, M<=32.
 * ---------------------------------------------------------------------------------------- */
size_t acg_cdna_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t c, int32_t masks, int32_t ksize);
int32_t acg_cdna_fwd(const void* params, const void* image, void* out, float* kern_norm, int32_t batch, int32_t h,
                     int32_t w, int32_t c, int32_t masks, int32_t ksize, float relu_shift, int32_t dtype,
                     acg_stream_t stream);
int32_t acg_cdna_bwd(const void* params, const float* kern_norm, const void* image, const void* dout, void* dparams,
                     void* dimage, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t masks, int32_t ksize,
                     float relu_shift, int32_t dtype, void* workspace, size_t workspace_bytes, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Channel plumbing: tf.tile + tf.concat at train.py:48-50,64,68 and models.py:16,38,84.
 * ---------------------------------------------------------------------------------------- */
/* y[b,s,0:c] = x[b,s,:],  y[b,s,c:c+a] = actions[b,:]   for s in [0,hw); y rows are y_pitch elements apart
 * (0 = dense = c+a), pad channels are not written.  x == NULL: the features are in y already (their producer - the
 * layer's BatchNorm - wrote them at this pitch), only the tiled actions are added. */
int32_t acg_concat_actions_fwd(const void* x, const float* actions, void* y, int32_t batch, int32_t hw,
                               int32_t c, int32_t a, int32_t y_pitch, int32_t dtype, acg_stream_t stream);
/* y[r,0:ca] = a[r,:], y[r,ca:ca+cb] = b[r,:]; y rows are y_pitch elements apart (0 = dense = ca+cb); pad channels
 * are not written.  cb may be 0 (b ignored): a plain re-pitching copy.  a may be NULL (with cb > 0): channels 0:ca of y
 * were written by someone else - the host's feed copy puts the fed frame there (train.py:64,68: both discriminator inputs
 * start with the fed current frame) - and only b's channels are written.  dtype: storage of a, b and y, or
 * ACG_DTYPE2(a and b, y) - ACG_DTYPE2(ACG_F32, ACG_BF16) builds the bf16 discriminator input from float32 frames. */
int32_t acg_concat_channels_fwd(const void* a, const void* b, void* y, int64_t rows, int32_t ca, int32_t cb,
                                int32_t y_pitch, int32_t dtype, acg_stream_t stream);
/* dst[r,:] = accumulate * dst[r,:] + src[r, c_off : c_off + c_dst]; dtype: storage of both, or ACG_DTYPE2(src, dst) */
int32_t acg_slice_channels(const void* src, void* dst, float accumulate, int64_t rows, int32_t c_src,
                           int32_t c_off, int32_t c_dst, int32_t dtype, acg_stream_t stream);
/* Stream-ordering edge "everything enqueued on `from` so far happens before what is enqueued on `to` from now on",
 * both streams on THIS device (no reference counterpart: TF's executor ordered its ops itself).  A default HIP event
 * carries a system-scope fence - an L2 write-back and invalidate on all 8 XCDs, ~20 us per edge on MI355X
 * (profiles/r1/w_stream_edges.txt); these use hipEventDisableTiming | hipEventDisableSystemFence: device memory
 * stays coherent between the device's own queues, which is all a dgrad-chain -> weight-gradient edge needs.
 * Not for host-visible or peer-visible data.  An edge object can be reused once per enqueue, indefinitely. */
typedef void* acg_edge_t;
int32_t acg_stream_edge_create(acg_edge_t* edge);
int32_t acg_stream_edge_destroy(acg_edge_t edge);
int32_t acg_stream_edge(acg_edge_t edge, acg_stream_t from, acg_stream_t to);

/* Up to ACG_COPY_MAX row-block copies in ONE launch: the feed_dict of a sess.run (train.py:115-154) lands in the
 * placeholders with a single kernel.  Segment i copies rows[i] x cols[i] floats from src[i] (dense rows) to dst[i] whose
 * rows are dst_pitch[i] floats apart (0 = dense; pad channels are not written). */
#define ACG_COPY_MAX 8
typedef struct acg_copy_list {
  const void* src[ACG_COPY_MAX];
  void* dst[ACG_COPY_MAX];
  int64_t rows[ACG_COPY_MAX];
  int32_t cols[ACG_COPY_MAX];
  int32_t dst_pitch[ACG_COPY_MAX];
  int32_t dst_dtype[ACG_COPY_MAX]; /* ACG_F32, or ACG_BF16: the float32 source is rounded into a bf16 placeholder */
  /* destination row r reads source row (r / src_div) % src_mod (0 = 1 / no wrap: a plain copy).  tf.tile of the action
   * vector over a feature map (train.py:48-50: [B,10] -> [B,h,w,10]) is src_div = h*w; the same vector shared by the
   * fake and the real half of a joined batch is src_mod = B.  rows[i] counts DESTINATION rows. */
  int32_t src_div[ACG_COPY_MAX];
  int32_t src_mod[ACG_COPY_MAX];
} acg_copy_list;
int32_t acg_copy_many(const acg_copy_list* list, int32_t count, int32_t dtype, acg_stream_t stream);
/* y = a + b (gradient fan-in where one tensor feeds two consumers, models.py:40-53) */
int32_t acg_add(const void* a, const void* b, void* y, int64_t n, int32_t dtype, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Losses.  Every loss writes its scalar value(s) to a float32 device slot and, when the
 * gradient pointer is non-NULL, d(scale * loss)/d(input) in the same pass.
 * ---------------------------------------------------------------------------------------- */
/* out[0] = sum|gen-gt|  (tf.norm ord=1, train.py:73);  out[1] = GDL(gen,gt) (ops.py:100-120);
 * dgen = w_l1 * d out[0] + w_gdl * d out[1]   (tf.abs gradient: sign, 0 at 0).  out2 == NULL (with dgen): the gradient
 * alone, ONE launch and no reductions - the training step of train.py:122-129 fetches the frame, not the loss value.
 * At most 4 channels. */
size_t acg_frame_loss_workspace_bytes(int64_t n);
int32_t acg_frame_loss(const void* gen, const void* gt, float* out2, void* dgen, int32_t batch, int32_t h,
                       int32_t w, int32_t c, float w_l1, float w_gdl, int32_t dtype, void* workspace,
                       size_t workspace_bytes, acg_stream_t stream);
/* out[0] = ||pred-gt||_2 (tf.norm ord=2, train.py:77); dpred = scale*(pred-gt)/norm (0 if norm==0). n <= 65536 */
int32_t acg_l2norm_loss(const float* pred, const float* gt, float* out, float* dpred, int64_t n, float scale,
                        acg_stream_t stream);
/* The same loss over a GLOBAL batch in a data-parallel run (SURVEY 8(e) caveat 3: the square root does not decompose over
 * ranks): acg_sumsq_diff gives this rank's sum (pred-gt)^2, the caller all-reduces it, acg_l2norm_loss_global takes
 * the norm from that global sum: out[0] = sqrt(global_sumsq[0]); dpred = scale*(pred-gt)/out[0]. */
int32_t acg_sumsq_diff(const float* pred, const float* gt, float* out, int64_t n, acg_stream_t stream);
int32_t acg_l2norm_loss_global(const float* pred, const float* gt, const float* global_sumsq, float* out, float* dpred,
                               int64_t n, float scale, acg_stream_t stream);
/* out[0] = mean(max(x,0) - x*label + log1p(exp(-|x|)))  (tf.losses.sigmoid_cross_entropy, ops.py:30-31,39-42);
 * dlogits = scale*(sigmoid(x)-label)/n.  n <= 65536 */
int32_t acg_sigmoid_ce_loss(const float* logits, float label, float* out, float* dlogits, int64_t n, float scale,
                            acg_stream_t stream);
/* out[0] = mean(x) (tf.reduce_mean, ops.py:32-33,44-45); dx = scale/n.  n <= 65536 */
int32_t acg_mean_loss(const float* x, float* out, float* dx, int64_t n, float scale, acg_stream_t stream);
/* out[0] = 10*log10(1/mean((a-b)^2))  (build_psnr, ops.py:19-20) */
int32_t acg_psnr(const void* a, const void* b, float* out, int64_t n, int32_t dtype, void* workspace,
                 size_t workspace_bytes, acg_stream_t stream);
/* out[0] = sum_i w_i * in_i[0] over the non-NULL inputs (loss sums of train.py:73-83,85) */
int32_t acg_scalar_combine(float* out, const float* in0, float w0, const float* in1, float w1, const float* in2,
                           float w2, const float* in3, float w3, acg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizers over FLAT float32 buffers (all variables of one scope are views into one buffer),
 * TensorFlow-1.0 formulas (train.py:91-102).  grad is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce).  use_clip != 0 fuses the D weight clip of
 * train.py:89,140-143 after the update (update -> clip order).
 * ---------------------------------------------------------------------------------------- */
/* tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), p -= lr_t*m/(sqrt(v)+eps); t read from *step_dev */
int32_t acg_adam_step(float* param, const float* grad, float* m, float* v, const int32_t* step_dev, int64_t n,
                      float lr, float beta1, float beta2, float eps, float grad_scale, int32_t use_clip,
                      float clip_lo, float clip_hi, acg_stream_t stream);
/* tf.train.RMSPropOptimizer (momentum 0): ms = decay*ms+(1-decay)*g*g; p -= lr*g/sqrt(ms+eps) */
int32_t acg_rmsprop_step(float* param, const float* grad, float* ms, int64_t n, float lr, float decay, float eps,
                         float grad_scale, int32_t use_clip, float clip_lo, float clip_hi, acg_stream_t stream);
/* tf.clip_by_value assign (train.py:89) */
int32_t acg_clip(float* param, int64_t n, float lo, float hi, acg_stream_t stream);
/* *step_dev += 1 (device-resident step counter so a captured graph replays correctly) */
int32_t acg_step_inc(int32_t* step_dev, acg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ACGAN_HIP_H */
