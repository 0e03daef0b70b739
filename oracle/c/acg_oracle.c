/*
 * acg_oracle.c - brute-force CPU restatement of the reference hot-path arithmetic, exporting the
 * same C ABI as libacgan_hip.so (include/acgan_hip.h) on HOST pointers.
 *
 * TEST INFRASTRUCTURE ONLY - PARITY UNPINNED (see oracle/__init__.py): the reference's arithmetic is
 * tensorflow==1.0.0 (requirements.txt:1), absent here; this file restates the published TF-1.0 / slim
 * op definitions as direct index loops with double accumulation, independently
 * of the torch composition in oracle/tf_ops.py.  The two must agree (tests/test_oracle.py).
 * It is loaded only by tests (as the checker, and as a stand-in device library for CPU-only
 * host-logic tests); the product never links or loads it.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/acgan_hip.h"

static __thread char g_err[256];
static int fail(int code, const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
#define REQUIRE_F32(dt) do { if ((dt) != ACG_F32) return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: float32 only"); } while (0)

int32_t acg_version(void) { return ACG_ABI_VERSION; }
const char* acg_build_info(void) { return "cpu-oracle"; }
const char* acg_last_error(void) { return g_err; }

static double sgn(double v) { return (v > 0) - (v < 0); }

/* ---- geometry: SURVEY A.1 */
int32_t acg_conv_desc_init(acg_conv_desc* d, int32_t batch, int32_t in_h, int32_t in_w, int32_t in_c,
                           int32_t kh, int32_t kw, int32_t out_c, int32_t stride, int32_t same) {
  if (!d || batch <= 0 || in_h <= 0 || in_w <= 0 || in_c <= 0 || kh <= 0 || kw <= 0 || out_c <= 0 || stride <= 0)
    return fail(ACG_ERR_INVALID_ARG, "conv_desc_init: non-positive dimension");
  d->batch = batch; d->in_h = in_h; d->in_w = in_w; d->in_c = in_c; d->out_c = out_c;
  d->kh = kh; d->kw = kw; d->stride_h = d->stride_w = stride; d->in_pitch = 0; d->out_pitch = 0;
  if (same) {
    d->out_h = (in_h + stride - 1) / stride; d->out_w = (in_w + stride - 1) / stride;
    int th = (d->out_h - 1) * stride + kh - in_h; if (th < 0) th = 0;
    int tw = (d->out_w - 1) * stride + kw - in_w; if (tw < 0) tw = 0;
    d->pad_top = th / 2; d->pad_left = tw / 2;
  } else {
    if (in_h < kh || in_w < kw) return fail(ACG_ERR_INVALID_ARG, "conv_desc_init: VALID kernel larger than input");
    d->out_h = (in_h - kh) / stride + 1; d->out_w = (in_w - kw) / stride + 1;
    d->pad_top = d->pad_left = 0;
  }
  return ACG_OK;
}

/* The oracle splits only the weight-gradient contraction, over the two halves of the batch (the product's planner splits
 * K = batch * out_h * out_w into up to 128 chunks): enough to state what the deferred-reduction entries mean. */
int32_t acg_conv2d_splits(const acg_conv_desc* d, int32_t which, int32_t dtype) {
  (void)dtype;
  if (!d || which < 0 || which > 2) return 0;
  return which == ACG_CONV_WGRAD && d->batch >= 2 ? 2 : 1;
}
/* the restatement has no tiles: one "tile" = the whole output */
int32_t acg_conv2d_tile(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t* tile_rows, int32_t* tile_cols) {
  (void)dtype;
  if (!d || which < 0 || which > 2) return 0;
  if (tile_rows) *tile_rows = which == ACG_CONV_WGRAD ? d->kh * d->kw * d->in_c : (which == ACG_CONV_DGRAD ? d->batch * d->in_h * d->in_w : d->batch * d->out_h * d->out_w);
  if (tile_cols) *tile_cols = which == ACG_CONV_DGRAD ? d->in_c : d->out_c;
  return 1;
}
size_t acg_conv2d_workspace_bytes(const acg_conv_desc* d, int32_t which, int32_t dtype) {
  const int32_t sp = acg_conv2d_splits(d, which, dtype);
  return sp > 1 ? (size_t)sp * d->kh * d->kw * d->in_c * d->out_c * sizeof(float) : 0;
}

#define XPITCH(d) ((d)->in_pitch > 0 ? (d)->in_pitch : (d)->in_c)
#define XI(d, b, y, x, c) ((((size_t)(b) * (d)->in_h + (y)) * (d)->in_w + (x)) * XPITCH(d) + (c))
#define YPITCH(d) ((d)->out_pitch > 0 ? (d)->out_pitch : (d)->out_c)
#define YI(d, b, p, q, o) ((((size_t)(b) * (d)->out_h + (p)) * (d)->out_w + (q)) * YPITCH(d) + (o))
#define WI(d, i, j, c, o) ((((size_t)(i) * (d)->kw + (j)) * (d)->in_c + (c)) * (d)->out_c + (o))

/* tf.nn.conv2d (models.py:12-15,34-37,42-51,82-88); SURVEY A.1 */
int32_t acg_conv2d_fwd(const void* xv, const void* wv, void* yv, const acg_conv_desc* d, int32_t dtype,
                       void* ws, size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  const float* x = xv; const float* w = wv; float* y = yv;
  for (int b = 0; b < d->batch; b++) for (int p = 0; p < d->out_h; p++) for (int q = 0; q < d->out_w; q++)
    for (int o = 0; o < d->out_c; o++) {
      double acc = 0;
      for (int i = 0; i < d->kh; i++) { int yy = p * d->stride_h - d->pad_top + i; if (yy < 0 || yy >= d->in_h) continue;
        for (int j = 0; j < d->kw; j++) { int xx = q * d->stride_w - d->pad_left + j; if (xx < 0 || xx >= d->in_w) continue;
          for (int c = 0; c < d->in_c; c++) acc += (double)x[XI(d, b, yy, xx, c)] * w[WI(d, i, j, c, o)]; } }
      y[YI(d, b, p, q, o)] = (float)acc;
    }
  return ACG_OK;
}

/* conv2d_backprop_input == tf.nn.conv2d_transpose (models.py:17-21,39-40,53-59); SURVEY A.2 (scatter form) */
int32_t acg_conv2d_dgrad(const void* dyv, const void* wv, void* dxv, const acg_conv_desc* d, int32_t dtype,
                         void* ws, size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  const float* dy = dyv; const float* w = wv; float* dx = dxv;
  size_t n = (size_t)d->batch * d->in_h * d->in_w * XPITCH(d);
  double* acc = calloc(n, sizeof(double));
  if (!acc) return fail(ACG_ERR_WORKSPACE, "oracle: out of memory");
  for (int b = 0; b < d->batch; b++) for (int p = 0; p < d->out_h; p++) for (int q = 0; q < d->out_w; q++)
    for (int i = 0; i < d->kh; i++) { int yy = p * d->stride_h - d->pad_top + i; if (yy < 0 || yy >= d->in_h) continue;
      for (int j = 0; j < d->kw; j++) { int xx = q * d->stride_w - d->pad_left + j; if (xx < 0 || xx >= d->in_w) continue;
        for (int c = 0; c < d->in_c; c++) { double a = 0;
          for (int o = 0; o < d->out_c; o++) a += (double)dy[YI(d, b, p, q, o)] * w[WI(d, i, j, c, o)];
          acc[XI(d, b, yy, xx, c)] += a; } } }
  for (size_t k = 0; k < n; k++) if ((int)(k % XPITCH(d)) < d->in_c) dx[k] = (float)acc[k];   /* pad channels untouched */
  free(acc);
  return ACG_OK;
}

int32_t acg_conv2d_wgrad(const void* xv, const void* dyv, float* dw, float accumulate, const acg_conv_desc* d,
                         int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  const float* x = xv; const float* dy = dyv;
  for (int i = 0; i < d->kh; i++) for (int j = 0; j < d->kw; j++) for (int c = 0; c < d->in_c; c++)
    for (int o = 0; o < d->out_c; o++) {
      double acc = 0;
      for (int b = 0; b < d->batch; b++) for (int p = 0; p < d->out_h; p++) { int yy = p * d->stride_h - d->pad_top + i; if (yy < 0 || yy >= d->in_h) continue;
        for (int q = 0; q < d->out_w; q++) { int xx = q * d->stride_w - d->pad_left + j; if (xx < 0 || xx >= d->in_w) continue;
          acc += (double)x[XI(d, b, yy, xx, c)] * dy[YI(d, b, p, q, o)]; } }
      size_t k = WI(d, i, j, c, o);
      dw[k] = (float)((accumulate != 0.f ? (double)accumulate * dw[k] : 0.0) + acc);
    }
  return ACG_OK;
}

/* include/acgan_hip.h "Deferred reduction of split weight gradients": slab z = the contribution of batch half z */
int32_t acg_conv2d_wgrad_slabs(const void* xv, const void* dyv, const acg_conv_desc* d, int32_t dtype, void* ws, size_t wsb,
                               acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  const int sp = acg_conv2d_splits(d, ACG_CONV_WGRAD, dtype);
  if (sp < 2) return fail(ACG_ERR_INVALID_ARG, "conv2d_wgrad_slabs: this shape is not split");
  const size_t numel = (size_t)d->kh * d->kw * d->in_c * d->out_c;
  if (!ws || wsb < sp * numel * sizeof(float)) return fail(ACG_ERR_WORKSPACE, "conv2d_wgrad_slabs: workspace too small");
  const float* x = xv; const float* dy = dyv; float* slabs = ws;
  for (int z = 0; z < sp; z++) {
    const int b0 = z * d->batch / sp, b1 = (z + 1) * d->batch / sp;
    for (int i = 0; i < d->kh; i++) for (int j = 0; j < d->kw; j++) for (int c = 0; c < d->in_c; c++)
      for (int o = 0; o < d->out_c; o++) {
        double acc = 0;
        for (int b = b0; b < b1; b++) for (int p = 0; p < d->out_h; p++) { int yy = p * d->stride_h - d->pad_top + i; if (yy < 0 || yy >= d->in_h) continue;
          for (int q = 0; q < d->out_w; q++) { int xx = q * d->stride_w - d->pad_left + j; if (xx < 0 || xx >= d->in_w) continue;
            acc += (double)x[XI(d, b, yy, xx, c)] * dy[YI(d, b, p, q, o)]; } }
        slabs[z * numel + WI(d, i, j, c, o)] = (float)acc;
      }
  }
  return ACG_OK;
}
int32_t acg_deconv2d_wgrad_slabs(const void* x, const void* dy, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb,
                                 acg_stream_t s) {
  return acg_conv2d_wgrad_slabs(dy, x, adj, dtype, ws, wsb, s); }
/* paired entries: the oracle simply runs the two contractions one after the other */
int32_t acg_conv2d_bwd_pair(const void* dy, const void* w, const void* x, void* dx, float* dw, float acc, const acg_conv_desc* d,
                            int32_t dtype, void* wsd, size_t wsbd, void* wsw, size_t wsbw, int32_t slabs_only, acg_stream_t s) {
  int rc = acg_conv2d_dgrad(dy, w, dx, d, dtype, wsd, wsbd, s);
  if (rc) return rc;
  return slabs_only ? acg_conv2d_wgrad_slabs(x, dy, d, dtype, wsw, wsbw, s) : acg_conv2d_wgrad(x, dy, dw, acc, d, dtype, wsw, wsbw, s);
}
int32_t acg_deconv2d_bwd_pair(const void* dy, const void* w, const void* x, void* dx, float* dw, float acc, const acg_conv_desc* adj,
                              int32_t dtype, void* wsd, size_t wsbd, void* wsw, size_t wsbw, int32_t slabs_only, acg_stream_t s) {
  int rc = acg_deconv2d_dgrad(dy, w, dx, adj, dtype, wsd, wsbd, s);
  if (rc) return rc;
  return slabs_only ? acg_deconv2d_wgrad_slabs(x, dy, adj, dtype, wsw, wsbw, s) : acg_deconv2d_wgrad(x, dy, dw, acc, adj, dtype, wsw, wsbw, s);
}

/* include/acgan_hip.h "BatchNorm statistics out of the producing convolution": the restatement runs the layer, then sums
 * its output rows in two blocks per group (one when the row count is odd), so callers see a multi-block partial layout. */
static int32_t stats_rows(const acg_conv_desc* d, int32_t which, int64_t* rows, int* C, int* pitch) {
  if (which == ACG_CONV_FWD) { *rows = (int64_t)d->batch * d->out_h * d->out_w; *C = d->out_c; *pitch = d->out_pitch > 0 ? d->out_pitch : d->out_c; return 1; }
  if (which == ACG_CONV_DGRAD) { *rows = (int64_t)d->batch * d->in_h * d->in_w; *C = d->in_c; *pitch = d->in_pitch > 0 ? d->in_pitch : d->in_c; return 1; }
  return 0;
}
int32_t acg_conv2d_stats_blocks(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t groups) {
  int64_t rows; int C, pitch;
  if (!d || dtype != ACG_F32 || groups < 1 || !stats_rows(d, which, &rows, &C, &pitch)) return 0;
  if (which == ACG_CONV_DGRAD && groups != 1) return 0;
  if (rows % groups) return 0;
  return (rows / groups) % 2 == 0 ? 2 : 1;
}
int32_t acg_conv2d_stats_layout(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t groups, int32_t* block_rows, int32_t* run_rows) {
  const int nblk = acg_conv2d_stats_blocks(d, which, dtype, groups);
  if (nblk > 0) {
    int64_t rows; int C, pitch;
    stats_rows(d, which, &rows, &C, &pitch);
    if (block_rows) *block_rows = (int32_t)(rows / groups / nblk);
    if (run_rows) *run_rows = (int32_t)(rows / groups);
  }
  return nblk;
}
/* per block: the sum and M2 = the sum of squared deviations from the block's own mean (include/acgan_hip.h) */
static void stats_of(const float* y, const acg_conv_desc* d, int32_t which, int32_t groups, float* part) {
  int64_t rows; int C, pitch;
  stats_rows(d, which, &rows, &C, &pitch);
  const int nblk = acg_conv2d_stats_blocks(d, which, ACG_F32, groups);
  const int64_t R = rows / groups, per = R / nblk;
  for (int g = 0; g < groups; g++) for (int b = 0; b < nblk; b++) for (int c = 0; c < C; c++) {
    double s1 = 0, m2 = 0;
    for (int64_t r = g * R + b * per; r < g * R + (b + 1) * per; r++) s1 += y[r * pitch + c];
    const double mb = s1 / (double)per;
    for (int64_t r = g * R + b * per; r < g * R + (b + 1) * per; r++) { double dv = y[r * pitch + c] - mb; m2 += dv * dv; }
    part[((size_t)(g * nblk + b) * 2) * C + c] = (float)s1;
    part[((size_t)(g * nblk + b) * 2 + 1) * C + c] = (float)m2;
  }
}
int32_t acg_conv2d_fwd_stats(const void* x, const void* w, void* y, const acg_conv_desc* d, int32_t dtype, void* ws, size_t wsb,
                             float* partials, int32_t groups, acg_stream_t s) {
  if (!partials || acg_conv2d_stats_blocks(d, ACG_CONV_FWD, dtype, groups) < 1) return fail(ACG_ERR_UNSUPPORTED, "conv2d_fwd_stats: no partials for this shape");
  int rc = acg_conv2d_fwd(x, w, y, d, dtype, ws, wsb, s);
  if (rc) return rc;
  stats_of(y, d, ACG_CONV_FWD, groups, partials);
  return ACG_OK;
}
int32_t acg_deconv2d_fwd_stats(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb,
                               float* partials, int32_t groups, acg_stream_t s) {
  if (!partials || acg_conv2d_stats_blocks(adj, ACG_CONV_DGRAD, dtype, groups) < 1) return fail(ACG_ERR_UNSUPPORTED, "deconv2d_fwd_stats: no partials for this shape");
  int rc = acg_deconv2d_fwd(x, w, y, adj, dtype, ws, wsb, s);
  if (rc) return rc;
  stats_of(y, adj, ACG_CONV_DGRAD, groups, partials);
  return ACG_OK;
}

static double act_f(int act, double u, double leak);
/* bias + activation in the deconv epilogue: the restatement never fuses (ok == 0); the entry itself is the plain composition */
int32_t acg_deconv2d_fwd_bias_act_ok(const acg_conv_desc* adj, int32_t dtype) { (void)adj; (void)dtype; return 0; }
int32_t acg_deconv2d_fwd_bias_act(const void* x, const void* w, const float* bias, float* y, const acg_conv_desc* adj, int32_t act,
                                  float leak, int32_t dtype, acg_stream_t s) {
  REQUIRE_F32(dtype);
  if (!bias || !y || !adj) return fail(ACG_ERR_INVALID_ARG, "deconv2d_fwd_bias_act: null pointer");
  int rc = acg_deconv2d_fwd(x, w, y, adj, dtype, NULL, 0, s);
  if (rc) return rc;
  const int C = adj->in_c, P = adj->in_pitch > 0 ? adj->in_pitch : C;
  const int64_t rows = (int64_t)adj->batch * adj->in_h * adj->in_w;
  for (int64_t r = 0; r < rows; r++) for (int c = 0; c < C; c++) y[r * P + c] = (float)act_f(act, (double)y[r * P + c] + bias[c], leak);
  return ACG_OK;
}

/* split-K hand-off entries: the restatement never splits forward / input-gradient contractions (acg_conv2d_splits) */
int32_t acg_conv2d_fwd_slabs(const void* x, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  (void)x; (void)w; (void)d; (void)dtype; (void)layout; (void)ws; (void)wsb; (void)s; return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: contraction is not split"); }
int32_t acg_conv2d_dgrad_slabs(const void* x, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  (void)x; (void)w; (void)d; (void)dtype; (void)layout; (void)ws; (void)wsb; (void)s; return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: contraction is not split"); }
int32_t acg_deconv2d_fwd_slabs(const void* x, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  (void)x; (void)w; (void)d; (void)dtype; (void)layout; (void)ws; (void)wsb; (void)s; return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: contraction is not split"); }
int32_t acg_deconv2d_dgrad_slabs(const void* x, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  (void)x; (void)w; (void)d; (void)dtype; (void)layout; (void)ws; (void)wsb; (void)s; return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: contraction is not split"); }
int32_t acg_bn_bwd_slabs_ok(int64_t rows, int32_t groups) { (void)rows; (void)groups; return 0; }
int32_t acg_conv2d_slab_layouts(const acg_conv_desc* d, int32_t which, int32_t dtype) { (void)d; (void)which; (void)dtype; return 0; }
int32_t acg_bn_slabs_layout(int64_t rows, int32_t C, int32_t xp, int32_t yp, int32_t groups, int32_t dtype, int32_t backward, int32_t flags) {
  (void)rows; (void)C; (void)xp; (void)yp; (void)groups; (void)dtype; (void)backward; (void)flags; return -1; }
int32_t acg_bn_exchange_selftest(void* ws, size_t wsb, float* out, int32_t blocks, int32_t threads, int32_t withhold, uint32_t spin_limit, acg_stream_t s) {
  (void)ws; (void)wsb; (void)out; (void)blocks; (void)threads; (void)withhold; (void)spin_limit; (void)s;
  return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: no in-launch exchange (a device mechanism)"); }
int32_t acg_bn_act_fwd_slabs(const float* slabs, int32_t splits, void* x, const float* beta, void* y, float* save_mean, float* save_rstd,
                             int64_t rows, int32_t C, int32_t xp, int32_t yp, int32_t groups, float eps, int32_t act, float leak, int32_t dtype,
                             int32_t layout, int32_t flags, void* ws, size_t wsb, acg_stream_t s) {
  (void)flags; (void)layout; (void)slabs; (void)splits; (void)x; (void)beta; (void)y; (void)save_mean; (void)save_rstd; (void)rows; (void)C; (void)xp; (void)yp; (void)groups;
  (void)eps; (void)act; (void)leak; (void)dtype; (void)ws; (void)wsb; (void)s; return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: no split-K hand-off"); }
int32_t acg_bn_act_bwd_slabs(const void* x, const float* dy_slabs, int32_t splits, const float* beta, const float* save_mean,
                             const float* save_rstd, void* dx, float* dbeta, float acc, int64_t rows, int32_t C, int32_t xp, int32_t yp,
                             int32_t groups, int32_t act, float leak, int32_t dtype, int32_t layout, int32_t flags, void* ws, size_t wsb, acg_stream_t s) {
  (void)flags; (void)layout; (void)x; (void)dy_slabs; (void)splits; (void)beta; (void)save_mean; (void)save_rstd; (void)dx; (void)dbeta; (void)acc; (void)rows; (void)C; (void)xp;
  (void)yp; (void)groups; (void)act; (void)leak; (void)dtype; (void)ws; (void)wsb; (void)s; return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: no split-K hand-off"); }
int32_t acg_weights_prepare_bf16(const acg_prep_list* l, int32_t count, acg_stream_t s) {
  (void)l; (void)count; (void)s;
  return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: float32 only");
}
int32_t acg_opt_step_prepare_bf16(float* param, const float* grad, float* slot1, float* slot2, const int32_t* step_dev, int64_t n,
                                  const acg_opt_args* args, const acg_prep_list* list, int32_t count, acg_stream_t s) {
  (void)param; (void)grad; (void)slot1; (void)slot2; (void)step_dev; (void)n; (void)args; (void)list; (void)count; (void)s;
  return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: float32 only");
}
int32_t acg_splitk_reduce_many(const acg_reduce_list* l, int32_t count, acg_stream_t s) {
  (void)s;
  if (!l || count < 1 || count > ACG_REDUCE_MAX) return fail(ACG_ERR_INVALID_ARG, "splitk_reduce_many: 1..32 entries");
  for (int e = 0; e < count; e++) {
    if (!l->slabs[e] || !l->out[e] || l->numel[e] <= 0 || l->splits[e] < 1) return fail(ACG_ERR_INVALID_ARG, "splitk_reduce_many: bad entry");
    for (int f = 0; f < e; f++) if (l->out[f] == l->out[e]) return fail(ACG_ERR_INVALID_ARG, "splitk_reduce_many: entries share an output");
  }
  for (int e = 0; e < count; e++) {
    const float* slabs = l->slabs[e]; float* out = l->out[e];
    for (int64_t i = 0; i < l->numel[e]; i++) {
      double acc = l->accumulate[e] != 0.f ? (double)l->accumulate[e] * out[i] : 0.0;
      for (int z = 0; z < l->splits[e]; z++) acc += slabs[(size_t)z * l->numel[e] + i];
      out[i] = (float)acc;
    }
  }
  if (l->step_inc) *l->step_inc += 1;
  return ACG_OK;
}

int32_t acg_deconv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  return acg_conv2d_dgrad(x, w, y, adj, dtype, ws, wsb, s); }
int32_t acg_deconv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  return acg_conv2d_fwd(dy, w, dx, adj, dtype, ws, wsb, s); }
int32_t acg_deconv2d_wgrad(const void* x, const void* dy, float* dw, float acc, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  return acg_conv2d_wgrad(dy, x, dw, acc, adj, dtype, ws, wsb, s); }

/* ---- activations: value and derivative w.r.t. the pre-activation u */
static double act_f(int act, double u, double leak) {
  switch (act) {
    case ACG_ACT_RELU: return u > 0 ? u : 0;
    case ACG_ACT_LRELU: return 0.5 * (1 + leak) * u + 0.5 * (1 - leak) * fabs(u);   /* ops.py:22-26 */
    case ACG_ACT_TANH: return tanh(u);
    default: return u;
  }
}
static double act_df(int act, double u, double leak) {
  switch (act) {
    case ACG_ACT_RELU: return u > 0 ? 1 : 0;
    case ACG_ACT_LRELU: return 0.5 * (1 + leak) + 0.5 * (1 - leak) * sgn(u);
    case ACG_ACT_TANH: { double t = tanh(u); return 1 - t * t; }
    default: return 1;
  }
}

/* slim.batch_norm training mode, SURVEY A.4 */
size_t acg_bn_workspace_bytes(int64_t rows, int32_t channels, int32_t groups) { (void)rows; (void)channels; (void)groups; return 0; }
int32_t acg_bn_act_fwd(const void* xv, const float* beta, void* yv, float* save_mean, float* save_rstd,
                       int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps, int32_t act,
                       float leak, int32_t dtype, int32_t flags, void* ws, size_t wsb, acg_stream_t s) {
  (void)flags; (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups) return fail(ACG_ERR_INVALID_ARG, "bn: rows not divisible by groups");
  const float* x = xv; float* y = yv; int64_t R = rows / groups;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  for (int g = 0; g < groups; g++) for (int c = 0; c < C; c++) {
    const float* xg = x + (size_t)g * R * XP; float* yg = y + (size_t)g * R * YP;
    double m = 0, v = 0;
    for (int64_t r = 0; r < R; r++) m += xg[r * XP + c];
    m /= (double)R;
    for (int64_t r = 0; r < R; r++) { double t = xg[r * XP + c] - m; v += t * t; }
    v /= (double)R;
    double rstd = 1.0 / sqrt(v + (double)eps);
    save_mean[g * C + c] = (float)m; save_rstd[g * C + c] = (float)rstd;
    for (int64_t r = 0; r < R; r++) yg[r * YP + c] = (float)act_f(act, (xg[r * XP + c] - m) * rstd + beta[c], leak);
  }
  return ACG_OK;
}

int32_t acg_bn_act_fwd_partials(const void* xv, const float* beta, const float* partials, int32_t nblk, int32_t block_rows, int32_t run_rows,
                                void* yv, float* save_mean, float* save_rstd, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch,
                                int32_t groups, float eps, int32_t act, float leak, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups || nblk < 1 || block_rows < 1 || run_rows < 1) return fail(ACG_ERR_INVALID_ARG, "bn partials: rows not divisible by groups / nblk < 1");
  const float* x = xv; float* y = yv; int64_t R = rows / groups;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  const int bpr = (run_rows + block_rows - 1) / block_rows;
  if (nblk % bpr || (int64_t)(nblk / bpr) * run_rows != R) return fail(ACG_ERR_INVALID_ARG, "bn partials: blocks do not cover the group");
  for (int g = 0; g < groups; g++) for (int c = 0; c < C; c++) {
    const float* xg = x + (size_t)g * R * XP; float* yg = y + (size_t)g * R * YP;
    double s1 = 0;
    for (int b = 0; b < nblk; b++) s1 += partials[((size_t)(g * nblk + b) * 2) * C + c];
    const double m = s1 / (double)R;
    double m2 = 0;     /* merge of the blocks: sum of M2_b + n_b (mean_b - mean)^2 */
    for (int b = 0; b < nblk; b++) {
      int nb = run_rows - (b % bpr) * block_rows; if (nb > block_rows) nb = block_rows;
      const double mb = partials[((size_t)(g * nblk + b) * 2) * C + c] / (double)nb;
      m2 += partials[((size_t)(g * nblk + b) * 2 + 1) * C + c] + (double)nb * (mb - m) * (mb - m);
    }
    double v = m2 / (double)R;
    if (v < 0) v = 0;
    double rstd = 1.0 / sqrt(v + (double)eps);
    save_mean[g * C + c] = (float)m; save_rstd[g * C + c] = (float)rstd;
    for (int64_t r = 0; r < R; r++) yg[r * YP + c] = (float)act_f(act, (xg[r * XP + c] - m) * rstd + beta[c], leak);
  }
  return ACG_OK;
}

int32_t acg_bn_act_bwd(const void* xv, const void* dyv, const float* beta, const float* save_mean,
                       const float* save_rstd, void* dxv, float* dbeta, float dbeta_acc,
                       int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act, float leak,
                       int32_t dtype, int32_t flags, void* ws, size_t wsb, acg_stream_t s) {
  (void)flags; (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups) return fail(ACG_ERR_INVALID_ARG, "bn: rows not divisible by groups");
  const float* x = xv; const float* dy = dyv; float* dx = dxv; int64_t R = rows / groups;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  for (int c = 0; c < C; c++) {
    double db_total = 0;
    for (int g = 0; g < groups; g++) {
      const float* xg = x + (size_t)g * R * XP; const float* dyg = dy + (size_t)g * R * YP; float* dxg = dx + (size_t)g * R * XP;
      double m = save_mean[g * C + c], rstd = save_rstd[g * C + c], s1 = 0, s2 = 0;
      for (int64_t r = 0; r < R; r++) { double xh = (xg[r * XP + c] - m) * rstd;
        double dp = dyg[r * YP + c] * act_df(act, xh + beta[c], leak); s1 += dp; s2 += dp * xh; }
      db_total += s1;
      for (int64_t r = 0; r < R; r++) { double xh = (xg[r * XP + c] - m) * rstd;
        double dp = dyg[r * YP + c] * act_df(act, xh + beta[c], leak);
        dxg[r * XP + c] = (float)(rstd * (dp - s1 / (double)R - xh * s2 / (double)R)); }
    }
    dbeta[c] = (float)((dbeta_acc != 0.f ? (double)dbeta_acc * dbeta[c] : 0.0) + db_total);
  }
  return ACG_OK;
}

/* ---- synchronised BatchNorm (SURVEY 8(e) caveat 1): statistics of the global batch, collectives by the caller */
int32_t acg_bn_moments(const void* xv, float* moments, int64_t rows, int32_t C, int32_t x_pitch, int32_t groups, int32_t dtype, void* ws,
                       size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups) return fail(ACG_ERR_INVALID_ARG, "bn: rows not divisible by groups");
  const float* x = xv; int64_t R = rows / groups; const int XP = x_pitch > 0 ? x_pitch : C;
  for (int g = 0; g < groups; g++) for (int c = 0; c < C; c++) {
    const float* xg = x + (size_t)g * R * XP; double m = 0, v = 0;
    for (int64_t r = 0; r < R; r++) m += xg[r * XP + c];
    m /= (double)R;
    for (int64_t r = 0; r < R; r++) { double t = xg[r * XP + c] - m; v += t * t; }
    moments[(g * 2 + 0) * C + c] = (float)m; moments[(g * 2 + 1) * C + c] = (float)(v / (double)R);
  }
  return ACG_OK;
}
int32_t acg_bn_act_fwd_moments(const void* xv, const float* beta, const float* moments, void* yv, float* save_mean,
                               float* save_rstd, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps,
                               int32_t act, float leak, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups) return fail(ACG_ERR_INVALID_ARG, "bn: rows not divisible by groups");
  const float* x = xv; float* y = yv; int64_t R = rows / groups;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  for (int g = 0; g < groups; g++) for (int c = 0; c < C; c++) {
    double m = moments[(g * 2 + 0) * C + c], rstd = 1.0 / sqrt((double)moments[(g * 2 + 1) * C + c] + (double)eps);
    save_mean[g * C + c] = (float)m; save_rstd[g * C + c] = (float)rstd;
    for (int64_t r = 0; r < R; r++) { size_t row = (size_t)g * R + r; y[row * YP + c] = (float)act_f(act, (x[row * XP + c] - m) * rstd + beta[c], leak); }
  }
  return ACG_OK;
}
int32_t acg_bn_bwd_sums(const void* xv, const void* dyv, const float* beta, const float* save_mean, const float* save_rstd,
                        float* sums, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act, float leak,
                        int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups) return fail(ACG_ERR_INVALID_ARG, "bn: rows not divisible by groups");
  const float* x = xv; const float* dy = dyv; int64_t R = rows / groups;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  for (int g = 0; g < groups; g++) for (int c = 0; c < C; c++) {
    double m = save_mean[g * C + c], rstd = save_rstd[g * C + c], s1 = 0, s2 = 0;
    for (int64_t r = 0; r < R; r++) { size_t row = (size_t)g * R + r; double xh = (x[row * XP + c] - m) * rstd;
      double dp = dy[row * YP + c] * act_df(act, xh + beta[c], leak); s1 += dp; s2 += dp * xh; }
    sums[(g * 2 + 0) * C + c] = (float)s1; sums[(g * 2 + 1) * C + c] = (float)s2;
  }
  return ACG_OK;
}
int32_t acg_bn_act_bwd_sums(const void* xv, const void* dyv, const float* beta, const float* save_mean, const float* save_rstd,
                            const float* sums, const float* local_sums, int64_t total_rows, void* dxv, float* dbeta,
                            float dbeta_acc, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act,
                            float leak, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (groups <= 0 || rows % groups) return fail(ACG_ERR_INVALID_ARG, "bn: rows not divisible by groups");
  const float* x = xv; const float* dy = dyv; float* dx = dxv; int64_t R = rows / groups;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  if (total_rows < R) return fail(ACG_ERR_INVALID_ARG, "bn_act_bwd_sums: total_rows smaller than this rank's rows");
  for (int c = 0; c < C; c++) {
    double db = 0;
    for (int g = 0; g < groups; g++) {
      double m = save_mean[g * C + c], rstd = save_rstd[g * C + c];
      double m1 = sums[(g * 2 + 0) * C + c] / (double)total_rows, m2 = sums[(g * 2 + 1) * C + c] / (double)total_rows;
      for (int64_t r = 0; r < R; r++) { size_t row = (size_t)g * R + r; double xh = (x[row * XP + c] - m) * rstd;
        double dp = dy[row * YP + c] * act_df(act, xh + beta[c], leak);
        dx[row * XP + c] = (float)(rstd * (dp - m1 - xh * m2)); }
      db += local_sums[(g * 2 + 0) * C + c];
    }
    dbeta[c] = (float)((dbeta_acc != 0.f ? (double)dbeta_acc * dbeta[c] : 0.0) + db);
  }
  return ACG_OK;
}

size_t acg_bias_workspace_bytes(int64_t rows, int32_t channels) { (void)rows; (void)channels; return 0; }
int32_t acg_bias_act_fwd(const void* xv, const float* bias, void* yv, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch,
                         int32_t act, float leak, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  const float* x = xv; float* y = yv;
  const int xp = x_pitch > 0 ? x_pitch : C, yp = y_pitch > 0 ? y_pitch : C;
  for (int64_t r = 0; r < rows; r++) for (int c = 0; c < C; c++)
    y[r * yp + c] = (float)act_f(act, (double)x[r * xp + c] + (bias ? bias[c] : 0.f), leak);
  return ACG_OK;
}
int32_t acg_bias_act_bwd(const void* yv, const void* dyv, void* dxv, float* dbias, float dbias_acc, int64_t rows,
                         int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t act, float leak, int32_t dtype, void* ws,
                         size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  const float* y = yv; const float* dy = dyv; float* dx = dxv;
  const int xp = x_pitch > 0 ? x_pitch : C, yp = y_pitch > 0 ? y_pitch : C;
  if (!dx && act != ACG_ACT_NONE) return fail(ACG_ERR_INVALID_ARG, "bias_act_bwd: dx NULL requires ACG_ACT_NONE");
  for (int c = 0; c < C; c++) {
    double sum = 0;
    for (int64_t r = 0; r < rows; r++) {
      double yy = y[r * yp + c], d;
      switch (act) {                                 /* derivative expressed through the OUTPUT y */
        case ACG_ACT_RELU: d = yy > 0 ? 1 : 0; break;
        case ACG_ACT_LRELU: d = 0.5 * (1 + leak) + 0.5 * (1 - leak) * sgn(yy); break;
        case ACG_ACT_TANH: d = 1 - yy * yy; break;
        default: d = 1;
      }
      double g = dy[r * yp + c] * d; sum += g;
      if (dx) dx[r * xp + c] = (float)g;
    }
    if (dbias) dbias[c] = (float)((dbias_acc != 0.f ? (double)dbias_acc * dbias[c] : 0.0) + sum);
  }
  return ACG_OK;
}

/* ---- CDNA transformation: ops.py:52-98 (normalised per-sample kernels, depthwise SAME correlation, split into M
 * pieces of C channels of the c-major depthwise output) */
static int cdna_check(int C, int M, int k) { return C >= 1 && C <= 4 && M >= 1 && M <= 32 && (k == 3 || k == 5 || k == 7); }
size_t acg_cdna_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t M, int32_t k) {
  if (B <= 0 || H <= 0 || W <= 0 || !cdna_check(C, M, k)) return 0;
  return (size_t)B * ((H + 15) / 16) * ((W + 15) / 16) * k * k * M * sizeof(float);
}
static void cdna_norm(const float* p, int kk, int M, double shift, double* n, double* S) {
  for (int m = 0; m < M; m++) {
    double s = 0;
    for (int t = 0; t < kk; t++) { double v = p[t * M + m] - shift; v = (v > 0 ? v : 0) + shift; n[t * M + m] = v; s += v; }
    S[m] = s;
    for (int t = 0; t < kk; t++) n[t * M + m] /= s;
  }
}
int32_t acg_cdna_fwd(const void* pv, const void* iv, void* ov, float* kern_norm, int32_t B, int32_t H, int32_t W,
                     int32_t C, int32_t M, int32_t k, float shift, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (B <= 0 || H <= 0 || W <= 0 || !cdna_check(C, M, k)) return fail(ACG_ERR_INVALID_ARG, "cdna: need c<=4, masks<=32, ksize in {3,5,7}");
  const float* par = pv; const float* img = iv; float* out = ov; int kk = k * k, pad = (k - 1) / 2;
  size_t plane = (size_t)B * H * W * C;
  double n[49 * 32], S[32];
  for (int b = 0; b < B; b++) {
    cdna_norm(par + (size_t)b * kk * M, kk, M, shift, n, S);
    if (kern_norm) for (int i = 0; i < kk * M; i++) kern_norm[(size_t)b * kk * M + i] = (float)n[i];
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) for (int c = 0; c < C; c++) for (int m = 0; m < M; m++) {
      double acc = 0;
      for (int u = 0; u < k; u++) for (int v = 0; v < k; v++) {
        int yy = y + u - pad, xx = x + v - pad; if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        acc += n[(u * k + v) * M + m] * img[(((size_t)b * H + yy) * W + xx) * C + c];
      }
      int q = c * M + m;
      out[(size_t)(q / C) * plane + (((size_t)b * H + y) * W + x) * C + q % C] = (float)acc;
    }
  }
  return ACG_OK;
}
int32_t acg_cdna_bwd(const void* pv, const float* kern_norm, const void* iv, const void* dov, void* dpv, void* div,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t M, int32_t k, float shift, int32_t dtype,
                     void* ws, size_t wsb, acg_stream_t s) {
  (void)s; (void)ws; (void)wsb; (void)kern_norm; REQUIRE_F32(dtype);
  if (B <= 0 || H <= 0 || W <= 0 || !cdna_check(C, M, k)) return fail(ACG_ERR_INVALID_ARG, "cdna: need c<=4, masks<=32, ksize in {3,5,7}");
  const float* par = pv; const float* img = iv; const float* dout = dov; float* dpar = dpv; float* dimg = div;
  int kk = k * k, pad = (k - 1) / 2; size_t plane = (size_t)B * H * W * C;
  double n[49 * 32], S[32], dn[49 * 32];
  for (int b = 0; b < B; b++) {
    cdna_norm(par + (size_t)b * kk * M, kk, M, shift, n, S);
    for (int i = 0; i < kk * M; i++) dn[i] = 0;
    if (dimg) for (size_t i = 0; i < (size_t)H * W * C; i++) dimg[(size_t)b * H * W * C + i] = 0.f;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) for (int c = 0; c < C; c++) for (int m = 0; m < M; m++) {
      int q = c * M + m;
      double g = dout[(size_t)(q / C) * plane + (((size_t)b * H + y) * W + x) * C + q % C];
      for (int u = 0; u < k; u++) for (int v = 0; v < k; v++) {
        int yy = y + u - pad, xx = x + v - pad; if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        size_t ii = (((size_t)b * H + yy) * W + xx) * C + c;
        dn[(u * k + v) * M + m] += g * img[ii];
        if (dimg) dimg[ii] += (float)(g * n[(u * k + v) * M + m]);
      }
    }
    for (int m = 0; m < M; m++) {
      double dot = 0;
      for (int t = 0; t < kk; t++) dot += dn[t * M + m] * n[t * M + m];
      for (int t = 0; t < kk; t++)
        dpar[(size_t)b * kk * M + t * M + m] = par[(size_t)b * kk * M + t * M + m] - shift > 0 ? (float)((dn[t * M + m] - dot) / S[m]) : 0.f;
    }
  }
  return ACG_OK;
}

/* ---- DNA tail: models.py:60-72, SURVEY A.7 */
static int dna_check(int c, int k) { return c >= 1 && c <= 4 && k >= 1 && k <= 15; }
size_t acg_dna_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t k) { (void)B; (void)H; (void)W; (void)k; return 0; }
int32_t acg_dna_fwd(const void* lv, const float* bias, const void* iv, void* ov, void* o2v, int32_t o2_pitch, int32_t o2_off, int32_t o2_dtype,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (!dna_check(C, k)) return fail(ACG_ERR_INVALID_ARG, "dna: need 1<=c<=4, 1<=ksize<=15");
  if (o2v && (o2_dtype != ACG_F32 || o2_off < 0 || o2_pitch < o2_off + C)) return fail(ACG_ERR_INVALID_ARG, "dna: second output");
  float* out2 = o2v;
  const float* lg = lv; const float* img = iv; float* out = ov; int kk = k * k, p = (k - 1) / 2;
  double l[225];
  for (int b = 0; b < B; b++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
    const float* l0 = lg + (((size_t)b * H + y) * W + x) * kk;
    for (int t = 0; t < kk; t++) l[t] = (double)l0[t] + (bias ? (double)bias[t] : 0.0);   /* softmax(logits + bias) */
    double mx = l[0], den = 0, acc[4] = {0, 0, 0, 0};
    for (int t = 1; t < kk; t++) if (l[t] > mx) mx = l[t];
    for (int t = 0; t < kk; t++) den += exp(l[t] - mx);
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) {
      int yy = y - p + i, xx = x - p + j; if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      double m = exp(l[i * k + j] - mx) / den;
      for (int c = 0; c < C; c++) acc[c] += m * img[(((size_t)b * H + yy) * W + xx) * C + c];
    }
    const size_t pix = ((size_t)b * H + y) * W + x;
    for (int c = 0; c < C; c++) { out[pix * C + c] = (float)acc[c];
      if (out2) out2[pix * o2_pitch + o2_off + c] = (float)acc[c]; }
    if (out2 && C == 3 && o2_off == 3 && o2_pitch == 8) {      /* concat(image, frame) at a pitch of 8: the whole pixel */
      for (int c = 0; c < 3; c++) out2[pix * 8 + c] = img[pix * 3 + c];
      out2[pix * 8 + 6] = out2[pix * 8 + 7] = 0.f; }
  }
  return ACG_OK;
}
int32_t acg_dna_bwd(const void* lv, const float* bias, const void* iv, const void* dov, const void* d2v, int32_t d2_pitch, int32_t d2_off,
                    int32_t d2_dtype, void* dlv, float* dbias, float dbias_acc,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  (void)s; (void)ws; (void)wsb; REQUIRE_F32(dtype);
  if (!dna_check(C, k)) return fail(ACG_ERR_INVALID_ARG, "dna: need 1<=c<=4, 1<=ksize<=15");
  if (d2v && (d2_dtype != ACG_F32 || d2_off < 0 || d2_pitch < d2_off + C)) return fail(ACG_ERR_INVALID_ARG, "dna: second gradient");
  const float* dout2 = d2v;
  const float* lg = lv; const float* img = iv; const float* dout = dov; float* dl = dlv; int kk = k * k, p = (k - 1) / 2;
  double m[225], g[225], l[225], bsum[225];
  for (int t = 0; t < kk; t++) bsum[t] = 0;
  for (int b = 0; b < B; b++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
    size_t pix = ((size_t)b * H + y) * W + x;
    const float* l0 = lg + pix * kk; double dO[4] = {0, 0, 0, 0};
    for (int c = 0; c < C; c++) dO[c] = (double)dout[pix * C + c] + (dout2 ? (double)dout2[pix * d2_pitch + d2_off + c] : 0.0);
    for (int t = 0; t < kk; t++) l[t] = (double)l0[t] + (bias ? (double)bias[t] : 0.0);
    double mx = l[0], den = 0, dot = 0;
    for (int t = 1; t < kk; t++) if (l[t] > mx) mx = l[t];
    for (int t = 0; t < kk; t++) { m[t] = exp(l[t] - mx); den += m[t]; }
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) {
      int t = i * k + j, yy = y - p + i, xx = x - p + j; m[t] /= den; g[t] = 0;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        for (int c = 0; c < C; c++) g[t] += dO[c] * img[(((size_t)b * H + yy) * W + xx) * C + c];
      dot += m[t] * g[t];
    }
    for (int t = 0; t < kk; t++) { const double v = m[t] * (g[t] - dot); dl[pix * kk + t] = (float)v; bsum[t] += v; }
  }
  if (dbias) for (int t = 0; t < kk; t++) dbias[t] = (float)((dbias_acc != 0.f ? (double)dbias_acc * dbias[t] : 0.0) + bsum[t]);
  return ACG_OK;
}

/* ---- channel plumbing: train.py:48-50,64,68; models.py:16,38,84 */
int32_t acg_concat_actions_fwd(const void* xv, const float* actions, void* yv, int32_t B, int32_t hw, int32_t c,
                               int32_t a, int32_t y_pitch, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  const float* x = xv; float* y = yv;
  const size_t py = y_pitch > 0 ? (size_t)y_pitch : (size_t)(c + a);
  if (py < (size_t)(c + a)) return fail(ACG_ERR_INVALID_ARG, "concat_actions: pitch smaller than the row");
  for (int b = 0; b < B; b++) for (int p = 0; p < hw; p++) {
    size_t r = (size_t)b * hw + p;
    if (x) memcpy(y + r * py, x + r * c, sizeof(float) * c);     /* x NULL: the producer wrote the features in place */
    memcpy(y + r * py + c, actions + (size_t)b * a, sizeof(float) * a);
  }
  return ACG_OK;
}
int32_t acg_concat_channels_fwd(const void* av, const void* bv, void* yv, int64_t rows, int32_t ca, int32_t cb,
                                int32_t y_pitch, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  const float* a = av; const float* b = bv; float* y = yv;
  const int64_t py = y_pitch > 0 ? y_pitch : ca + cb;
  if (py < ca + cb) return fail(ACG_ERR_INVALID_ARG, "concat_channels: pitch smaller than the row");
  if (!a && cb <= 0) return fail(ACG_ERR_INVALID_ARG, "concat_channels: null pointer");
  for (int64_t r = 0; r < rows; r++) { if (a) memcpy(y + r * py, a + r * ca, sizeof(float) * ca);
                                       if (cb > 0) memcpy(y + r * py + ca, b + r * cb, sizeof(float) * cb); }
  return ACG_OK;
}
int32_t acg_slice_channels(const void* sv, void* dv, float acc, int64_t rows, int32_t c_src, int32_t c_off,
                           int32_t c_dst, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (c_off < 0 || c_off + c_dst > c_src) return fail(ACG_ERR_INVALID_ARG, "slice_channels: range outside source");
  const float* src = sv; float* dst = dv;
  for (int64_t r = 0; r < rows; r++) for (int c = 0; c < c_dst; c++)
    dst[r * c_dst + c] = (acc != 0.f ? acc * dst[r * c_dst + c] : 0.f) + src[r * c_src + c_off + c];
  return ACG_OK;
}
/* stream-ordering edges: the CPU oracle is synchronous, an edge is a no-op */
int32_t acg_stream_edge_create(acg_edge_t* edge) { static int token; if (!edge) return fail(ACG_ERR_INVALID_ARG, "stream_edge_create: null output"); *edge = &token; return ACG_OK; }
int32_t acg_stream_edge_destroy(acg_edge_t edge) { return edge ? ACG_OK : fail(ACG_ERR_INVALID_ARG, "stream_edge_destroy: null edge"); }
int32_t acg_stream_edge(acg_edge_t edge, acg_stream_t from, acg_stream_t to) { (void)from; (void)to; return edge ? ACG_OK : fail(ACG_ERR_INVALID_ARG, "stream_edge: null edge"); }

int32_t acg_copy_many(const acg_copy_list* l, int32_t count, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  if (!l || count < 1 || count > ACG_COPY_MAX) return fail(ACG_ERR_INVALID_ARG, "copy_many: 1..8 segments");
  for (int i = 0; i < count; i++) {
    if (!l->src[i] || !l->dst[i] || l->rows[i] <= 0 || l->cols[i] <= 0) return fail(ACG_ERR_INVALID_ARG, "copy_many: bad segment");
    int pitch = l->dst_pitch[i] > 0 ? l->dst_pitch[i] : l->cols[i];
    if (pitch < l->cols[i]) return fail(ACG_ERR_INVALID_ARG, "copy_many: pitch smaller than the row");
    if (l->dst_dtype[i] != ACG_F32) return fail(ACG_ERR_UNSUPPORTED, "cpu oracle: float32 only");
    const float* src = l->src[i]; float* dst = l->dst[i];
    const int64_t dv = l->src_div[i] > 0 ? l->src_div[i] : 1, md = l->src_mod[i];
    for (int64_t r = 0; r < l->rows[i]; r++) { int64_t sr = r / dv; if (md > 0) sr %= md;
      for (int c = 0; c < l->cols[i]; c++) dst[r * pitch + c] = src[sr * l->cols[i] + c]; }
  }
  return ACG_OK;
}
int32_t acg_add(const void* av, const void* bv, void* yv, int64_t n, int32_t dtype, acg_stream_t s) {
  (void)s; REQUIRE_F32(dtype);
  const float* a = av; const float* b = bv; float* y = yv;
  for (int64_t i = 0; i < n; i++) y[i] = a[i] + b[i];
  return ACG_OK;
}

/* ---- losses: SURVEY A.5 */
size_t acg_frame_loss_workspace_bytes(int64_t n) { (void)n; return 0; }
int32_t acg_frame_loss(const void* gv, const void* tv, float* out2, void* dgv, int32_t B, int32_t H, int32_t W,
                       int32_t C, float w_l1, float w_gdl, int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  const float* gen = gv; const float* gt = tv; float* dgen = dgv;
  double l1 = 0, gdl = 0;
#define AT(t, b, y, x, c) (((y) < H && (x) < W) ? (double)(t)[((((size_t)(b)) * H + (y)) * W + (x)) * C + (c)] : 0.0)
  for (int b = 0; b < B; b++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) for (int c = 0; c < C; c++) {
    double e = AT(gen, b, y, x, c) - AT(gt, b, y, x, c);
    l1 += fabs(e);
    /* ops.py:100-120: dx = in[x+1]-in[x], dy = in[y]-in[y+1], zero beyond the edge */
    double gdx = AT(gen, b, y, x + 1, c) - AT(gen, b, y, x, c), tdx = AT(gt, b, y, x + 1, c) - AT(gt, b, y, x, c);
    double gdy = AT(gen, b, y, x, c) - AT(gen, b, y + 1, x, c), tdy = AT(gt, b, y, x, c) - AT(gt, b, y + 1, x, c);
    gdl += fabs(fabs(tdx) - fabs(gdx)) + fabs(fabs(tdy) - fabs(gdy));
    if (dgen) {
      /* d/dgen[y,x]: own dx term (-), left neighbour's dx term (+), own dy term (+), upper neighbour's dy term (-) */
      double d = -(-sgn(fabs(tdx) - fabs(gdx)) * sgn(gdx)) + (-sgn(fabs(tdy) - fabs(gdy)) * sgn(gdy));
      if (x > 0) { double g2 = AT(gen, b, y, x, c) - AT(gen, b, y, x - 1, c), t2 = AT(gt, b, y, x, c) - AT(gt, b, y, x - 1, c);
        d += -sgn(fabs(t2) - fabs(g2)) * sgn(g2); }
      if (y > 0) { double g2 = AT(gen, b, y - 1, x, c) - AT(gen, b, y, x, c), t2 = AT(gt, b, y - 1, x, c) - AT(gt, b, y, x, c);
        d -= -sgn(fabs(t2) - fabs(g2)) * sgn(g2); }
      dgen[(((size_t)b * H + y) * W + x) * C + c] = (float)(w_l1 * sgn(e) + w_gdl * d);
    }
  }
#undef AT
  if (out2) { out2[0] = (float)l1; out2[1] = (float)gdl; }      /* out2 == NULL: the gradient alone */
  return ACG_OK;
}
int32_t acg_l2norm_loss(const float* p, const float* g, float* out, float* dp, int64_t n, float scale, acg_stream_t s) {
  (void)s; double ss = 0;
  for (int64_t i = 0; i < n; i++) { double e = (double)p[i] - g[i]; ss += e * e; }
  double nrm = sqrt(ss); out[0] = (float)nrm;
  if (dp) for (int64_t i = 0; i < n; i++) dp[i] = nrm > 0 ? (float)(scale * ((double)p[i] - g[i]) / nrm) : 0.f;
  return ACG_OK;
}
int32_t acg_sumsq_diff(const float* p, const float* g, float* out, int64_t n, acg_stream_t s) {
  (void)s; double ss = 0;
  for (int64_t i = 0; i < n; i++) { double e = (double)p[i] - g[i]; ss += e * e; }
  out[0] = (float)ss;
  return ACG_OK;
}
int32_t acg_l2norm_loss_global(const float* p, const float* g, const float* gss, float* out, float* dp, int64_t n, float scale, acg_stream_t s) {
  (void)s; double nrm = sqrt((double)gss[0]); out[0] = (float)nrm;
  if (dp) for (int64_t i = 0; i < n; i++) dp[i] = nrm > 0 ? (float)(scale * ((double)p[i] - g[i]) / nrm) : 0.f;
  return ACG_OK;
}
int32_t acg_sigmoid_ce_loss(const float* x, float label, float* out, float* dx, int64_t n, float scale, acg_stream_t s) {
  (void)s; double sum = 0;
  for (int64_t i = 0; i < n; i++) { double v = x[i];
    sum += (v > 0 ? v : 0) - v * label + log1p(exp(-fabs(v)));
    if (dx) dx[i] = (float)(scale * (1.0 / (1.0 + exp(-v)) - label) / (double)n); }
  out[0] = (float)(sum / (double)n);
  return ACG_OK;
}
int32_t acg_mean_loss(const float* x, float* out, float* dx, int64_t n, float scale, acg_stream_t s) {
  (void)s; double sum = 0;
  for (int64_t i = 0; i < n; i++) { sum += x[i]; if (dx) dx[i] = (float)(scale / (double)n); }
  out[0] = (float)(sum / (double)n);
  return ACG_OK;
}
int32_t acg_psnr(const void* av, const void* bv, float* out, int64_t n, int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  (void)ws; (void)wsb; (void)s; REQUIRE_F32(dtype);
  const float* a = av; const float* b = bv; double ss = 0;
  for (int64_t i = 0; i < n; i++) { double e = (double)a[i] - b[i]; ss += e * e; }
  out[0] = (float)(10.0 * log(1.0 / (ss / (double)n)) / log(10.0));
  return ACG_OK;
}
int32_t acg_scalar_combine(float* out, const float* i0, float w0, const float* i1, float w1, const float* i2, float w2,
                           const float* i3, float w3, acg_stream_t s) {
  (void)s; double v = 0;
  if (i0) v += (double)w0 * i0[0];
  if (i1) v += (double)w1 * i1[0];
  if (i2) v += (double)w2 * i2[0];
  if (i3) v += (double)w3 * i3[0];
  out[0] = (float)v;
  return ACG_OK;
}

/* ---- optimizers: SURVEY A.6 (TensorFlow formulas) */
int32_t acg_adam_step(float* p, const float* g, float* m, float* v, const int32_t* step_dev, int64_t n, float lr,
                      float b1, float b2, float eps, float gs, int32_t use_clip, float lo, float hi, acg_stream_t s) {
  (void)s; int t = *step_dev;
  if (t < 1) return fail(ACG_ERR_INVALID_ARG, "adam: step counter must be >= 1 (call acg_step_inc first)");
  double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t));
  for (int64_t i = 0; i < n; i++) {
    double gi = (double)g[i] * gs;
    double mi = (double)b1 * m[i] + (1.0 - (double)b1) * gi;
    double vi = (double)b2 * v[i] + (1.0 - (double)b2) * gi * gi;
    double pi = p[i] - lr_t * mi / (sqrt(vi) + (double)eps);
    if (use_clip) pi = pi < lo ? lo : (pi > hi ? hi : pi);
    m[i] = (float)mi; v[i] = (float)vi; p[i] = (float)pi;
  }
  return ACG_OK;
}
int32_t acg_rmsprop_step(float* p, const float* g, float* ms, int64_t n, float lr, float decay, float eps, float gs,
                         int32_t use_clip, float lo, float hi, acg_stream_t s) {
  (void)s;
  for (int64_t i = 0; i < n; i++) {
    double gi = (double)g[i] * gs;
    double msi = (double)decay * ms[i] + (1.0 - (double)decay) * gi * gi;
    double pi = p[i] - (double)lr * gi / sqrt(msi + (double)eps);
    if (use_clip) pi = pi < lo ? lo : (pi > hi ? hi : pi);
    ms[i] = (float)msi; p[i] = (float)pi;
  }
  return ACG_OK;
}
int32_t acg_clip(float* p, int64_t n, float lo, float hi, acg_stream_t s) {
  (void)s; for (int64_t i = 0; i < n; i++) p[i] = p[i] < lo ? lo : (p[i] > hi ? hi : p[i]);
  return ACG_OK;
}
int32_t acg_step_inc(int32_t* step_dev, acg_stream_t s) { (void)s; *step_dev += 1; return ACG_OK; }
