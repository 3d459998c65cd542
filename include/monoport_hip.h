/* monoport_hip.h -- C-ABI of the MI355X-native MonoPort reconstruction hot path.
 *
 * Drop-in boundary: plain pointers and sizes, no torch types, no C++
 * exceptions.  Every entry point names the reference interface it replaces (file:line under
 * the MonoPort tree).  The reference is pure Python; its "FFI" for this path is the set of
 * torch ops listed in SURVEY.md section 2a, so a maintainer binds these symbols with ctypes (see
 * INTEGRATION.md) exactly as monoport_amd/_lib.py does.
 *
 * Conventions
 *   - all tensor arguments are DEVICE pointers on the context's GPU unless marked (host);
 *   - fp32 everywhere ("f32" compute on v_mfma_f32_32x32x2_f32); int64 for vertex indices;
 *   - work is enqueued on `stream` (a hipStream_t; NULL = the default stream) and the call
 *     returns without synchronising, except where noted;
 *   - the caller owns every input/output buffer; the context owns only its scratch arena and
 *     the packed MLP weights;
 *   - a context may be used from any host thread; calls on ONE context are serialised by an
 *     internal mutex (RTL/dataloader.py:1026-1053 runs every pipeline stage on its own thread:
 *     give each stage its own context);
 *   - return value: MP_OK or a negative MP_ERR_*; mp_last_error(ctx) gives the message of the
 *     last failing call MADE BY THE CALLING HOST THREAD (per-thread storage, like errno), valid
 *     until that thread's next failing call;
 *   - calls that SYNCHRONISE: mp_mlp_set_precision (drains the stream of the last mp_mlp_load),
 *     mp_mlp_destroy, mp_stream_release and mp_destroy (hipDeviceSynchronize), mp_profile_end
 *     (waits for the recorded events).  Everything else only enqueues.
 *   - scratch: one arena per (context, stream), grown by adding blocks -- a pointer handed to a
 *     kernel stays valid until mp_stream_release / mp_destroy, so work captured in a hipGraph
 *     keeps working after later, larger calls on the same stream.
 */
#ifndef MONOPORT_HIP_H
#define MONOPORT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mp_ctx mp_ctx;
typedef void *mp_stream; /* hipStream_t */

enum {
  MP_OK = 0,
  MP_ERR_ARG = -1,         /* bad pointer / size / enum */
  MP_ERR_HIP = -2,         /* a HIP runtime call failed */
  MP_ERR_UNSUPPORTED = -3, /* shape outside what the kernels are built for */
  MP_ERR_STATE = -4,       /* e.g. querying an MLP whose layers are not all loaded */
  MP_ERR_NOMEM = -5
};

/* SurfaceClassifier last_op (heads/SurfaceClassifier.py:68-69, :77, :85) */
enum { MP_ACT_NONE = 0, MP_ACT_SIGMOID = 1, MP_ACT_TANH = 2 };

/* arithmetic of the MLP GEMMs (mp_mlp_set_precision) */
enum {
  MP_PREC_F32 = 0,   /* v_mfma_f32_32x32x2_f32: exact f32 products, the default */
  MP_PREC_F16X3 = 1, /* f32 emulated on v_mfma_f32_32x32x16_f16: every operand split into two
                        halves (hi + lo, 22 significant bits), three MFMAs per product term
                        (hi*hi + hi*lo + lo*hi), f32 accumulation; netG heads (C = 256) only */
  MP_PREC_F16W = 2,  /* fp16 WEIGHTS (rounded once, per-layer power-of-two scale), activations
                        still split: hi*hi + hi*lo, two MFMAs (BASELINE configs[4]) */
  MP_PREC_F16 = 3    /* fp16 weights and fp16 activations, f32 accumulation: one MFMA */
};

/* forward_vertices direction (RTL/recon.py:39-49) */
enum { MP_DIR_FRONT = 0, MP_DIR_BACK = 1, MP_DIR_LEFT = 2, MP_DIR_RIGHT = 3 };

/* ---- context ---------------------------------------------------------------------------- */
int mp_version(void);
/* Creates a context bound to HIP device `device`.  Fails (MP_ERR_HIP) when no GPU is visible:
 * there is no CPU fallback. */
int mp_create(int device, mp_ctx **out);
void mp_destroy(mp_ctx *ctx);
const char *mp_last_error(mp_ctx *ctx); /* calling thread's last error; ctx may be NULL */
/* Frees the scratch arena the context keeps for `stream` (call when the stream is destroyed, e.g.
 * by the owner of a pipeline slot; arenas are keyed by the stream handle).  Synchronises the
 * device.  Unknown streams are fine (MP_OK). */
int mp_stream_release(mp_ctx *ctx, mp_stream stream);
/* A HIP stream whose kernels run on `n_cus` of the device's compute units only (round 6).  The reference runs
 * every pipeline stage on its own host thread (RTL/dataloader.py:1026-1053), so netG.filter of frame k+1 and
 * reconEngine of frame k are in flight together (RTL/main.py:366-395) -- but a persistent query launch owns every
 * CU's LDS for milliseconds and a batch-1 encoder cannot fill 256 CUs: on ordinary streams the two stages take
 * turns.  Giving each stage a disjoint share of the CUs lets them run side by side.  Bit i of the mask =
 * "CU i", i in [first_cu, first_cu + n_cus); the driver deals consecutive bits out round-robin over the 8 XCDs
 * and their shader engines, so a range that is a multiple of 32 takes the same share of every XCD.  The
 * persistent query kernels size their grids from the stream's share (launches on other streams: the whole
 * device).  The stream is created on the context's device, non-blocking; destroy it with mp_stream_destroy
 * (synchronises the stream, frees its arena; an allocator that pools memory per stream -- torch's -- must drop the
 * stream's cached blocks first, `torch.cuda.empty_cache()`, or it will touch a dead stream later).  MP_ERR_ARG unless 0 <= first_cu, n_cus >= 8 and first_cu +
 * n_cus <= the device's CU count. */
int mp_stream_create_cu_mask(mp_ctx *ctx, int first_cu, int n_cus, mp_stream *out);
int mp_stream_destroy(mp_ctx *ctx, mp_stream stream);
/* CUs a launch on `stream` is sized for: the share of a stream made by mp_stream_create_cu_mask on this
 * context, else the device's count. */
int mp_stream_cu_count(mp_ctx *ctx, mp_stream stream);
/* Device memory the context owns right now (host array of 4): [0] scratch arenas incl. outgrown blocks still
 * kept for captured graphs, [1] packed MLP weights (all precisions), [2] number of arenas, [3] number of
 * skip tables registered (the tables themselves are the caller's).  Soak tests assert these stay flat. */
int mp_memory_stats(mp_ctx *ctx, int64_t *out4);
/* Frames one fused-query / mp_recon_batch / mp_query_counted_batch call accepts (kMaxFrames). */
int mp_max_frames(void);

/* ---- SurfaceClassifier weights ------------------------------------------------------------ */
/* Replaces SurfaceClassifier.__init__ (heads/SurfaceClassifier.py:7-37) for the skip-concat
 * MLPs the reference instantiates: channels = {C+1,1024,512,256,128,Cout} with C in {256,512},
 * Cout in {1,3} (PIFuNetGMLP :74-79, PIFuNetCMLP :82-87).  Other shapes: MP_ERR_UNSUPPORTED. */
int mp_mlp_create(mp_ctx *ctx, int n_layers, const int *channels /*host, n_layers+1*/,
                  int last_op, int *mlp_out);
/* Loads filters.{layer}.{weight,bias} (state-dict layout, weight [out,in(,1)] row-major with the
 * K order [hidden | feature | z] of SurfaceClassifier.py:55) and re-packs it into MFMA fragment
 * order.  W and b are DEVICE pointers.  Replaces load_state_dict / load_legacy_pifu
 * (MonoPortNet.py:153-160).  Loading a layer FORGETS every skip table made with this head
 * (mp_skip_table below: a table holds products of the old weights); make them again afterwards. */
int mp_mlp_load(mp_ctx *ctx, int mlp, int layer, const float *W, const float *b, int out_ch,
                int in_ch, mp_stream stream);
int mp_mlp_destroy(mp_ctx *ctx, int mlp); /* synchronises the device before freeing */
/* Selects the arithmetic used by mp_query / mp_recon / mp_query_counted for this MLP.  Call after
 * every layer is loaded.  The f16 variants re-pack the weights ON THE STREAM OF THE LAST
 * mp_mlp_load (so the re-pack is ordered behind the loads), read max |W| of each layer back to
 * the host to choose the per-layer power-of-two operand scale, and drain that stream before
 * returning: SYNCHRONOUS, call it at set-up time. */
int mp_mlp_set_precision(mp_ctx *ctx, int mlp, int precision);

/* ---- feature-map layout ------------------------------------------------------------------ */
/* Copies src [Csrc,H,W] (the NCHW map MonoPortNet.filter emits, MonoPortNet.py:31-46) into
 * channels [c_offset, c_offset+Csrc) of the channels-last map dst [H,W,Cdst] the query kernels
 * read.  netC's cat([feat_prior, feat]) (MonoPortNet.py:44) is two calls into one dst. */
int mp_feat_pack_hwc(mp_ctx *ctx, const float *src_chw, int c_src, int h, int w, float *dst_hwc,
                     int c_dst, int c_offset, mp_stream stream);

/* Optional accelerator of the query of a netG head (C = 256; exact f32 and f16x3 kernels): the SKIP TABLE of a feature
 * map.  SurfaceClassifier (heads/SurfaceClassifier.py:39-71) multiplies weights with the SAMPLED
 * feature in every layer -- layer 0's 1024 x 256 block and the skip connections of layers 1-4 (:55)
 * -- and the sampled feature is a bilinear blend of four texels (geometry.py:4-16).  A linear map
 * commutes with that blend, so  table[y][x][r] = sum_c W[r][c] feat[y][x][c]  for those 1921 weight
 * rows (no bias, no z column; row layout: layers 0-3 at 0 / 1024 / 1536 / 1792, layer 4 at 1920,
 * padded to MP_SKIP_TABLE_ROWS) is computed ONCE per feature map here -- 16 GFLOP for a 128 x 128
 * map -- and the query of a point blends four rows of it instead of multiplying 492 k weights: 42 %
 * of the per-point MFMA work gone.  The result differs from the plain path by f32 rounding only
 * (1-3e-7 on the field, far inside the 1e-4 bar; tests/test_query_gpu.py::test_skip_table_*).
 * The call also REGISTERS the table for feat_hwc in the context: every later fused query launch
 * (mp_query*, mp_recon*) whose feature maps ALL have a registered table made with the same head
 * and map size uses it; others run the plain path.  The caller owns `table`
 * ([H, W, MP_SKIP_TABLE_ROWS] f32, 16-byte aligned -- 128-byte aligned keeps every 32-row block of a texel in
 * one cache line, which is what the row length of 61 x 32 floats is for; H * W a multiple of 64), must call mp_skip_table
 * again after it rewrites the feature map (stream-ordered with the queries, like any producer), and
 * mp_skip_table_release(ctx, feat_hwc, table) before freeing either buffer (table NULL: whatever is
 * registered for feat_hwc; otherwise only if it still is that table; feat_hwc NULL: everything).
 * mp_mlp_destroy and mp_mlp_load (new weights) drop the tables made with that head. */
#define MP_SKIP_TABLE_ROWS 1952
int mp_skip_table(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w, float *table,
                mp_stream stream);
/* The same for n_maps maps stored back to back ([n_maps, H, W, C] -> table [n_maps, H, W, MP_SKIP_TABLE_ROWS]):
 * one launch, one registration per map. */
int mp_skip_table_batch(mp_ctx *ctx, int mlp, int n_maps, const float *feat_hwc, int c, int h, int w,
                      float *table, mp_stream stream);
int mp_skip_table_release(mp_ctx *ctx, const float *feat_hwc, const float *table);

/* ---- per-point ops ------------------------------------------------------------------------- */
/* index(feat, uv) (geometry.py:4-16): bilinear grid_sample, align_corners=True, zero padding.
 * feat_hwc [H,W,C]; uv [2,N] in [-1,1]; out [C,N]. */
int mp_index(mp_ctx *ctx, const float *feat_hwc, int c, int h, int w, const float *uv, int64_t n,
             float *out, mp_stream stream);
/* orthogonal(points, calib) (geometry.py:19-34, transforms=None): out = R p + t.
 * points/out [3,N]; calib = row-major [>=3,4] matrix (rows 0-2 used, row stride 4). */
int mp_orthogonal(mp_ctx *ctx, const float *points, int64_t n, const float *calib, float *out,
                  mp_stream stream);
/* MonoPortNet.query in eval mode, one feature stage (MonoPortNet.py:48-91):
 * project -> in-image mask -> z*z_scale -> bilinear sample -> skip-concat MLP -> mask.
 * points: element (c, i) at points[i*stride_n + c*stride_c] (so both the [3,N] layout netG.query
 * takes and the [N,3] one query_func receives, RTL/main.py:169-183, bind without a copy).
 * out [Cout,N]; out-of-image points are exactly 0.0f. */
int mp_query(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w,
             const float *points, int64_t n, int64_t stride_n, int64_t stride_c,
             const float *calib, float z_scale, float *out, mp_stream stream);

/* SurfaceClassifier.forward on explicit features (heads/SurfaceClassifier.py:39-71; the shape of
 * the reference's own micro-benchmark, :95-116): feature [C+1,N] (sampled features + z_feat as
 * the last row, MonoPortNet.py:82-83) -> out [Cout,N] with the last_op applied. */
int mp_mlp_forward(mp_ctx *ctx, int mlp, const float *feature, int64_t n, float *out,
                   mp_stream stream);

/* Same, with the point count read from device memory at run time (netC.query over the vertices
 * forward_vertices found, RTL/main.py:239-242, without a host round trip).  points [3,capacity],
 * out [Cout,capacity]; only the first *count columns are read / written. */
int mp_query_counted(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w,
                     const float *points, int64_t capacity, const int32_t *count,
                     const float *calib, float z_scale, float *out, mp_stream stream);

/* mp_query_counted over n_frames (1..32) independent frames in ONE launch (the per-vertex colour
 * queries of all frames of a pipeline slot: ~14 k points each cannot fill 256 CUs alone).
 * feat_hwc / points / count / calib / out are HOST arrays of n_frames device pointers, each as in
 * mp_query_counted (one `capacity` for all); results are identical to n_frames separate calls. */
int mp_query_counted_batch(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c,
                           int h, int w, const float *const *points, int64_t capacity,
                           const int32_t *const *count, const float *const *calib, float z_scale,
                           float *const *out, mp_stream stream);

/* ---- coarse-to-fine reconstruction --------------------------------------------------------- */
/* Replaces implicit_seg.functional.Seg3dLossless.forward(faster=True) driving query_func
 * (RTL/main.py:185-195, :392-394; un-vendored dependency, requirements.txt:15).
 * resolutions (host) must satisfy r[i+1] = 2 r[i] - 1, r <= 1023.  volume [R,R,R] (z,y,x) f32
 * with R = resolutions[n_levels-1].  status (device, int32[1+n_levels]): status[0] = 1 if the
 * coarsest level has any value > balance (0 -> the reference returns None and `volume` is
 * unspecified), status[1+l] = points queried at level l.  Fully asynchronous. */
int mp_recon(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w, const float *calib,
             float z_scale, const float *b_min /*host[3]*/, const float *b_max /*host[3]*/,
             const int *resolutions /*host*/, int n_levels, float balance, float *volume,
             int32_t *status, mp_stream stream);

/* mp_recon over n_frames (1..32) independent frames sharing the MLP, box and resolutions: every
 * octree level evaluates the selected nodes of ALL frames in one fused-query launch, so the coarse
 * levels (5-25 k nodes per frame) fill the 256 CUs together.  feat_hwc / calib / volume / status
 * are HOST arrays of n_frames device pointers, each as in mp_recon; results are identical to
 * n_frames mp_recon calls. */
int mp_recon_batch(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c, int h,
                   int w, const float *const *calib, float z_scale, const float *b_min,
                   const float *b_max, const int *resolutions, int n_levels, float balance,
                   float *const *volume, int32_t *const *status, mp_stream stream);

/* mp_recon_batch with the selection rule of the LAST level as an argument.  The un-vendored upstream
 * engine (implicit_seg, RTL/main.py:188-195 constructs it with faster=True) is recalled to treat its last
 * level differently from the others; nothing under the reference pins it, so the
 * rule is the caller's choice:
 *   MP_FINAL_DILATE3      boundary nodes (0 < upsampled mask < 1) dilated by 3^3, as at levels >= 3: the
 *                         lossless schedule -- thresholded volume == thresholded dense evaluation on
 *                         ordinary bodies.  mp_recon / mp_recon_batch use it.
 *   MP_FINAL_UPSTREAM     only nodes whose upsampled mask is EXACTLY 0.5 (`is_boundary = valid == 0.5`,
 *                         no dilation): ~4x fewer points at the last level; ~0.4 % of the inside voxels
 *                         differ from dense evaluation (tests/test_recon_gpu.py::test_final_level_*).
 *   MP_FINAL_INTERPOLATE  nothing is evaluated at the last level ("last step no examine"): the volume is
 *                         the trilinear upsample of the level before; status[n_levels] = 0. */
enum { MP_FINAL_DILATE3 = 0, MP_FINAL_UPSTREAM = 1, MP_FINAL_INTERPOLATE = 2 };
int mp_recon_batch_ex(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c, int h,
                      int w, const float *const *calib, float z_scale, const float *b_min,
                      const float *b_max, const int *resolutions, int n_levels, float balance,
                      int final_level, float *const *volume, int32_t *const *status, mp_stream stream);

/* mp_recon_batch_ex with an EARLY hand-over (round 6).  The reference's engine returns None for a frame whose
 * coarsest level is empty (RTL/recon.py:32-33) and our drop-in class must know whether the caller's query_func is
 * the plain netG.query the fused kernels implement -- both facts are known after the coarsest level, ~0.1 ms into
 * a ~3.5 ms call.  With `early`, right after that level the call
 *   - compares frame f's coarsest-level values with expect_level0[f] (device, res[0]^3 floats in (z,y,x) order:
 *     what query_func returned for those nodes; NULL entry or NULL array = no comparison),
 *   - writes flags_dev[2f] = status[f][0] and flags_dev[2f+1] = 1 if any value differs (bitwise float !=),
 *   - copies the 2*n_frames flags to flags_host (PINNED host memory) and records `event` (a hipEvent_t, may be
 *     NULL) on `stream`, then enqueues the remaining levels.
 * A stage thread waits for `event`, reads two integers and hands the volume on while the GPU is still refining
 * it; the per-level counts in `status` are complete when the stream is.  Results are those of mp_recon_batch_ex. */
typedef struct mp_recon_early {
  const float *const *expect_level0;
  int32_t *flags_dev;  /* device int32[2 * n_frames] */
  int32_t *flags_host; /* pinned host int32[2 * n_frames] */
  void *event;
} mp_recon_early;
int mp_recon_batch_early(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c, int h,
                         int w, const float *const *calib, float z_scale, const float *b_min,
                         const float *b_max, const int *resolutions, int n_levels, float balance,
                         int final_level, float *const *volume, int32_t *const *status,
                         const mp_recon_early *early, mp_stream stream);

/* The same engine one level at a time, for an arbitrary Python ``query_func`` (the general
 * Seg3dLossless contract, RTL/main.py:169-195): the caller evaluates the selected nodes itself.
 *   mp_octree_select: level 0 (prev == NULL) selects every node; otherwise upsamples prev [rp^3]
 *     into cur [r^3] (r = 2 rp - 1), flags boundary nodes, dilates by the level's box, drops nodes
 *     evaluated earlier.  ev_prev / ev_cur / bnd are u64 bitsets of r*r*ceil(r/64) words (rp for
 *     ev_prev); packed (u32 [r^3]) receives x | y<<10 | z<<20, count (device int32) their number.
 *   mp_lattice_points: packed nodes -> world points [capacity,3] (row i = x,y,z of node i).
 *   mp_scatter_nodes: volume[z,y,x] = values[i] for the first *count nodes. */
int mp_octree_select(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                     const uint64_t *ev_prev, uint64_t *ev_cur, uint64_t *bnd, int level,
                     float balance, uint32_t *packed, int32_t *count, mp_stream stream);
/* mp_octree_select with an explicit dilation box (3, 7 or 9) instead of the faster-mode schedule
 * 9 / 7 / 3 by level: the upstream engine's faster=False mode dilates by 3^3 at every level.
 * box 1: the MP_FINAL_UPSTREAM rule (nodes whose upsampled mask is exactly 0.5, undilated);
 * box 0: the MP_FINAL_INTERPOLATE rule (upsample only, *count = 0). */
int mp_octree_select_box(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                         const uint64_t *ev_prev, uint64_t *ev_cur, uint64_t *bnd, int box,
                         float balance, uint32_t *packed, int32_t *count, mp_stream stream);
/* Conflict re-examination of the upstream engine's faster=False mode.  For each of the first
 * *count nodes of `packed` (just evaluated: values[i]; volume [r^3] still holds the value
 * INTERPOLATED from the coarser level at that node): if (interp - balance) * (value - balance) < 0
 * every node of its 3x3x3 neighbourhood that is not yet in the evaluated bitset `ev` is claimed
 * (bit set atomically) and appended to out_packed (capacity r^3; order unspecified); *out_count
 * = their number.  Call mp_scatter_nodes for `packed` afterwards, then evaluate out_packed and
 * repeat until *out_count is 0. */
int mp_octree_conflicts(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, int64_t capacity,
                        int r, const float *values, const float *volume, float balance,
                        uint64_t *ev, uint32_t *out_packed, int32_t *out_count, mp_stream stream);
int mp_lattice_points(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, int64_t capacity,
                      int stride, int res_final, const float *b_min /*host[3]*/,
                      const float *b_max /*host[3]*/, float *points, mp_stream stream);
int mp_scatter_nodes(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, int64_t capacity,
                     int r, const float *values, float *volume, mp_stream stream);

/* ---- visible-surface extraction ------------------------------------------------------------ */
/* forward_vertices (RTL/recon.py:27-89).  volume [R,R,R]; outputs sized for R*R rows:
 * X, Y int64 [R*R]; Z f32 [R*R]; norm f32 [R*R,3]; count (device int32[1]) = rows written,
 * in the reference's row order (x-major, RTL/recon.py:62). */
int mp_forward_vertices(mp_ctx *ctx, const float *volume, int r, int direction, int64_t *x,
                        int64_t *y, float *z, float *norm, int32_t *count, mp_stream stream);

/* mp_forward_vertices over n_frames (1..32) volumes of one size in ONE set of launches (the vertex extraction of a
 * single 257^3 volume is three launches of 65-2300 workgroups: launch-bound).  volume / x / y / z / norm / count are
 * HOST arrays of n_frames device pointers, each as in mp_forward_vertices; results are identical to n_frames calls. */
int mp_forward_vertices_batch(mp_ctx *ctx, int n_frames, const float *const *volume, int r, int direction,
                              int64_t *const *x, int64_t *const *y, float *const *z, float *const *norm,
                              int32_t *const *count, mp_stream stream);

/* ---- colorization (RTL/main.py:212-249) ---------------------------------------------------- */
/* verts = (X, Y, res - Z) mapped through the voxel->world matrix `mat` (host, row-major 4x4,
 * RTL/main.py:204-210, :231-237) -> points [3,N] for netC.query.  `count` (device int32) gives N
 * (<= capacity); rows beyond it are left untouched. */
int mp_vertex_points(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *z,
                     const int32_t *count, int64_t capacity, int res, const float *mat,
                     float *points, mp_stream stream);
/* image [res,res,3] = 1.0 then image[X[i],Y[i],:] = clamp(values[:,i]*scale + bias, lo, hi);
 * values is [3,capacity] when channel_major != 0 (netC preds, main.py:244-248: scale=bias=0.5)
 * or [capacity,3] otherwise (normals, main.py:220-225: scale=bias=0.5, clamp 0..1). */
int mp_paint(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *values,
             int channel_major, const int32_t *count, int64_t capacity, int res, float scale,
             float bias, float lo, float hi, float *image, mp_stream stream);

/* mp_paint over n_frames (1..32) renders of one size in two launches; x / y / values / count / image are HOST arrays
 * of n_frames device pointers, each as in mp_paint (one `capacity` for all). */
int mp_paint_batch(mp_ctx *ctx, int n_frames, const int64_t *const *x, const int64_t *const *y,
                   const float *const *values, int channel_major, const int32_t *const *count, int64_t capacity,
                   int res, float scale, float bias, float lo, float hi, float *const *image, mp_stream stream);

/* visulization (sic, RTL/main.py:252-281) for one render: out[i,j,:] = 255 * image[rot90, nearest
 * resized res -> size]; image [res,res,3] f32 in [0,1], out [size,size,3] f32, mask [size,size]
 * uint8 = 0 where all three channels are exactly 255 (the white background), else 1. */
int mp_visualize(mp_ctx *ctx, const float *image, int res, int size, float *out, uint8_t *mask,
                 mp_stream stream);

/* Background removal + normalisation in front of the encoders (RTL/main.py:352-364): segm is the
 * segmentation engine's [4,H,W] output (RGB in [-1,1], then the soft mask);
 *   input_g[c] = (((segm[c] * 0.5 + 0.5) - mean[c]) / std[c]) * segm[3]     (netG input)
 *   input_c[c] = segm[c] * segm[3]                                          (netC input, may be NULL)
 * in exactly this operation order (bit-identical to the reference's chain of torch ops).
 * mean / std: host float[3] (cfg.netG.mean / .std, RTL/main.py:288-289). */
int mp_prepare_inputs(mp_ctx *ctx, const float *segm, int64_t hw, const float *mean,
                      const float *std, float *input_g, float *input_c, mp_stream stream);

/* ---- triangle mesh (north star; no counterpart in the reference) ----------------- */
/* Marching cubes of volume [R,R,R] at `level` (inside = value > level) with the face-consistent
 * case table of tools/gen_mc_tables.py.  One welded vertex per crossing lattice edge, in edge-id
 * order ((z*R+y)*R+x)*3 + axis, positioned at the linear crossing and mapped to world space like
 * the octree lattice; triangles in cell order, wound counter-clockwise seen from the outside.
 * verts f32 [max_verts,3], faces int32 [max_faces,3]; counts (device int32[2]) = vertices and
 * faces NEEDED (compare with the capacities to detect truncation). */
int mp_marching_cubes(mp_ctx *ctx, const float *volume, int r, float level,
                      const float *b_min /*host[3]*/, const float *b_max /*host[3]*/, float *verts,
                      int64_t max_verts, int32_t *faces, int64_t max_faces, int32_t *counts,
                      mp_stream stream);

/* ---- encoder helpers (stand-alone GroupNorm / upsample / concat kernels -- the
 * convolutions are the mp_conv* entry points below, nothing of the inference path is left on MIOpen) ---------- */
/* y = [relu](GroupNorm(groups, C)(x)): x, y [N,C,HW] f32 (contiguous NCHW), gamma/beta [C];
 * biased variance, eps inside the sqrt -- torch.nn.GroupNorm as used by
 * backbones/HGFilters.py:23-27 and ResBlkFilters.py:19.  Needs HW % 4 == 0. */
int mp_group_norm(mp_ctx *ctx, const float *x, int n, int c, int64_t hw, int groups,
                  const float *gamma, const float *beta, float eps, int relu, float *y,
                  mp_stream stream);
/* y = [add +] bicubic_x2(x), align_corners=True, A = -0.75 (F.interpolate at HGFilters.py:108 and
 * the skip add at :111).  x [C,H,W] (fold a batch into C); add (may be NULL), y [C,2H,2W]. */
int mp_upsample_bicubic2x(mp_ctx *ctx, const float *x, int c, int h, int w, const float *add,
                          float *y, mp_stream stream);

/* y = cat((a, b, c), channel axis) + shortcut: the tail of the encoders' pyramid block
 * (backbones/HGFilters.py:57-60: torch.cat((out1, out2, out3), 1) followed by `out3 += residual`)
 * in one pass.  a [N,Ca,HW], b [N,Cb,HW], c [N,Cc,HW], shortcut and y [N,Ca+Cb+Cc,HW]; HW % 4 == 0. */
int mp_concat3_add(mp_ctx *ctx, const float *a, int ca, const float *b, int cb, const float *c,
                   int cc, const float *shortcut, int n, int64_t hw, float *y, mp_stream stream);

/* ---- encoder convolutions with fused GroupNorm (hand-written, f32 MFMA) ------------------------ */
/* The pyramid block of the encoders is conv3x3(relu(GroupNorm(x))) three times
 * (backbones/HGFilters.py:40-62).  mp_conv3x3_gn computes
 *     y = conv3x3(v, W),  v = relu?(x * scale[n,c] + shift[n,c])  (v = x when ss == NULL),
 * stride 1, zero padding 1 (applied to v), no bias -- nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False) on
 * the normalised tensor -- as an implicit GEMM on v_mfma_f32_32x32x2_f32, and optionally emits the
 * partial sums the NEXT GroupNorm(32, Cout) needs.  x [N,Cin,H,W], y [N,Cout,H,W] contiguous NCHW;
 * ss [N,Cin,2] = (scale, shift) from mp_gn_finalize; packed = W re-ordered by mp_conv3x3_pack
 * (Cout*Cin*9 floats).  stats: NULL or double [N,32,S,2] (per image, group and tile) with S = mp_conv3x3_stat_slices(Cout,N,H,W,f16) (f16 = 0 here, 1 for mp_conv3x3_gn16).
 * Needs Cin % 16 == 0, Cout % 32 == 0, H and W powers of two (W >= 32); else MP_ERR_UNSUPPORTED.
 * mp_conv3x3_tune(nr): measurement hook -- force nr (1, 2, 4) 32-pixel column blocks per wave
 * instead of the launch-size heuristic (0 restores it); | 0x100 forces the large-tile kernel, | 0x200
 * the split-K kernel, | mrw << 12 the row blocks per wave of mp_conv1x1; process-wide, not for
 * production use. */
int mp_conv3x3_pack(mp_ctx *ctx, const float *w /*[Cout,Cin,3,3]*/, int cout, int cin, float *packed,
                    mp_stream stream);
int mp_conv3x3_supported(int cin, int cout, int h, int w); /* 1 if the shape is built, else 0 */
int mp_conv3x3_stat_slices(int cout, int n, int h, int w, int f16);
void mp_conv3x3_tune(int nr);
/* mp_query_tune(small_tiles): measurement hook for the fused f32 query of the netG heads on
 * sampled features -- launches of fewer than small_tiles 64-point tiles run on the 32-point-tile
 * kernel (query_small.hip), the others on the 64-point kernel (query.hip).  0: never, 1: always,
 * negative: the default (2048).  The two kernels return identical bits. */
void mp_query_tune(int small_tiles);
int mp_conv3x3_gn(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss, int relu,
                  int reflect, const float *packed, int cout, float *y, double *stats, mp_stream stream);
/* reflect != 0: nn.ReflectionPad2d(1) + an unpadded 3x3 convolution (the residual blocks of the
 * netC encoder, backbones/ResBlkFilters.py:28-84) instead of zero padding.
 * mp_scale_shift_add: y = res + (t * scale[n,c] + shift[n,c]) -- x + GroupNorm(conv(.)) at the end of
 * such a block (no ReLU), ss from mp_gn_finalize; t, res, y [N,C,HW], HW % 4 == 0. */
int mp_scale_shift_add(mp_ctx *ctx, const float *t, const float *ss, const float *res, int n, int c,
                       int64_t hw, float *y, mp_stream stream);
/* The same convolution on split-f16 operands ("f16x3": every operand hi + lo, three
 * v_mfma_f32_32x32x16_f16 per product term, f32 accumulation; f32-class accuracy, the arithmetic of
 * MP_PREC_F16X3).  mp_conv3x3_pack16 writes the pre-split weights (Cout*Cin*9*4 bytes) and
 * max|W| (device float[1], from which pack and convolution derive the same power-of-two operand
 * scale); fully asynchronous. */
int mp_conv3x3_pack16(mp_ctx *ctx, const float *w, int cout, int cin, void *packed16, float *wmax,
                      mp_stream stream);
int mp_conv3x3_gn16(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss, int relu,
                    int reflect, const void *packed16, const float *wmax, int cout, float *y, double *stats,
                    mp_stream stream);
/* The 1x1 convolutions of the hourglass tail (backbones/HGFilters.py:184-204: conv_last, l, bl,
 * al; nn.Conv2d(C, 256, 1) with bias) and the 1x1 projection of a pyramid block whose channel
 * count changes (HGFilters.py:47-52: GroupNorm, ReLU, Conv2d(Cin, Cout, 1, bias=False)) as one
 * fused GEMM each:
 *     y = W [relu?(x1 * scale + shift) ; x2] + bias (+ res)
 * cout = 256, or 128 (then no stats / y_hwc); bias may be NULL; x1 [N,C1,HW] with optional fused GroupNorm (ss1 [N,C1,2]) + ReLU; x2 [N,C2,HW] an
 * optional second K segment (bl(y) + al(out) is ONE call with W = [W_bl | W_al], bias = b_bl +
 * b_al, res = x); outputs: y [N,256,HW] and / or y_hwc [N,HW,256] (the channels-last map the
 * query kernels read, written in 1 KB bursts) -- at least one; stats: NULL or the partial sums of
 * GroupNorm(32, 256) over the output, double [N,32,mp_conv1x1_stat_slices(HW),2].  mp_conv1x1_pack re-orders
 * W1 [cout,C1] (and W2 [cout,C2]) into fragment order: cout*(C1+C2) floats, or the same number of
 * (hi, lo) f16 pairs when f16 != 0 (then wmax, device float[1], receives max|W|).  C1, C2 and HW
 * must be multiples of 64. */
int mp_conv1x1_pack(mp_ctx *ctx, const float *w1, int c1, const float *w2, int c2, int cout, int f16,
                    void *packed, float *wmax, mp_stream stream);
int mp_conv1x1(mp_ctx *ctx, const float *x1, const float *ss1, int relu1, const float *x2, int n, int c1,
               int c2, int cout, int64_t hw, const void *packed, int f16, const float *wmax,
               const float *bias, const float *res, float *y, float *y_hwc, double *stats, mp_stream stream);
/* GroupNorm statistics as two steps.  mp_gn_stats: partial (sum, sum of squares) of x [N,C,HW] per
 * (image, group, slice) -> double [N*groups, mp_gn_stat_slices(), 2] (one read pass; for tensors
 * that do not come out of mp_conv3x3_gn).  mp_gn_finalize: partial sums (either source) -> ss [N,C,2] =
 * (gamma rstd, beta - mean gamma rstd), biased variance over `count` elements per group, eps inside
 * the square root (torch.nn.GroupNorm). */
int mp_gn_stat_slices(void);
int mp_gn_stats(mp_ctx *ctx, const float *x, int n, int c, int64_t hw, int groups, double *partial,
                mp_stream stream);
int mp_gn_finalize(mp_ctx *ctx, const double *partial, int n, int c, int groups, int slices,
                   int64_t count, const float *gamma, const float *beta, float eps, float *ss,
                   mp_stream stream);

/* ---- GroupNorm handed from producer to consumer (round 3) --------------------------------------
 * Every GroupNorm(32, C) of the encoders (backbones/HGFilters.py:23-27, ResBlkFilters.py:19) needs
 * the statistics of a whole (image, group) before the first normalised value exists.  The kernels
 * below ADD the per-group sums of the tensor they WRITE into an accumulator (fire-and-forget 64-bit
 * integer atomics on a fixed-point representation: order-independent, hence deterministic), and the
 * kernel that READS the tensor turns the accumulator into (mean, rstd) in its prologue and applies
 * scale = rstd * gamma[c], shift = beta[c] - mean * scale while it stages its input.  A normalised
 * tensor never exists in memory and a GroupNorm costs neither a launch nor a wait (csrc/gn_tail.h
 * records the variants that were measured and dropped).
 *   accumulator  int64 [R,N,32,4] per normalised tensor, R = mp_gn_acc_replicas() copies that the
 *                producers' workgroups spread over (same-address device atomics serialise) and the
 *                consumer adds up; a group's 4 words = (sum hi, sum lo, sumsq hi, sumsq lo) with
 *                value = hi * 2^-16 + lo * 2^-64; it must be ZERO before the producing launch
 *                (hipMemsetAsync one arena per encoder pass) and is complete when that launch has
 *                finished; several launches may fill disjoint groups of one accumulator (the three
 *                convolutions of a pyramid block);
 *   mp_gn_out    producer side: acc and / or the legacy partial-sum buffer (partial_doubles >=
 *                N * 32 * slices * 2 with the launch's mp_*_stat_slices; finalise with
 *                mp_gn_finalize); both NULL = no statistics;
 *   mp_gn_in     consumer side: acc + the GroupNorm's gamma / beta / eps, or the legacy precomputed
 *                ss [N,C,2] from mp_gn_finalize, or all NULL = plain input. */
int mp_gn_acc_replicas(void);
typedef struct mp_gn_out {
  int64_t *acc;
  double *partial;
  int64_t partial_doubles;
} mp_gn_out;
typedef struct mp_gn_in {
  const int64_t *acc;
  const float *gamma;
  const float *beta;
  float eps;
  const float *ss;
} mp_gn_in;

/* mp_conv3x3_gn / mp_conv3x3_gn16 with the pyramid block's tail and the GroupNorm hand-over fused in
 * (backbones/HGFilters.py:40-62).  gn: GroupNorm(32, Cin) of the input (+ ReLU with `relu`).  wmax ==
 * NULL: packed by mp_conv3x3_pack (exact f32); else by mp_conv3x3_pack16.  y may be NULL when y2 is
 * given.  y2 / res [N, y2_channels, H, W]:
 *     y2[n, y2_offset + c] = conv[n, c] + res[n, y2_offset + c]
 * -- torch.cat((out1, out2, out3), 1) + residual (HGFilters.py:57-60) written by the three
 * convolutions themselves.  fin: GroupNorm(32, Cout) over y (the next convolution of the block);
 * fin2: this launch's groups of GroupNorm(32, y2_channels) over y2 (the next block; groups must not
 * straddle launches: y2_offset % (y2_channels / 32) == 0).  Launches too small to fill the chip run
 * a split-K variant (the four waves of a workgroup share one 32-channel x 32/64-pixel tile). */
typedef struct mp_conv3x3_args {
  const float *x;
  int n, cin, h, w;
  mp_gn_in gn;
  int relu, reflect;
  const void *packed;
  const float *wmax;
  int cout;
  float *y;
  float *y2;
  const float *res;
  int y2_channels, y2_offset;
  mp_gn_out fin, fin2;
  /* round 6: NULL, or the Winograd-domain weights of the same convolution (mp_conv3x3_pack_wino, 16 * Cout * Cin
   * floats).  With them, launches that mp_conv3x3_wino_supported serves (zero padding, exact f32, no legacy
   * statistics buffers) run as Winograd F(2x2, 3x3) on the same f32 MFMAs -- 4 / 9 of the multiplies of the direct
   * form; `packed` must still be given (every other launch uses it). */
  const float *packed_wino;
} mp_conv3x3_args;
int mp_conv3x3_ex(mp_ctx *ctx, const mp_conv3x3_args *args, mp_stream stream);
/* U = G g G^T of nn.Conv2d(Cin, Cout, 3, 1, 1) weights (backbones/HGFilters.py:15-19) in the fragment order of
 * csrc/conv_wino.hip, computed in double and rounded once; 16 * Cout * Cin floats, 16-byte aligned. */
int mp_conv3x3_pack_wino(mp_ctx *ctx, const float *w /*[Cout,Cin,3,3]*/, int cout, int cin, float *packed_wino,
                         mp_stream stream);
int mp_conv3x3_wino_supported(int cin, int cout, int h, int w); /* 1 if the Winograd kernel is built for the shape */

/* mp_conv1x1 with the GroupNorm hand-over: gn1 = GroupNorm(32, C1) of x1 (+ ReLU with relu1); fin =
 * GroupNorm(32, 256) over the output (res included), i.e. bn_end after conv_last and the first
 * GroupNorm of the next stack after x + bl(.) + al(.) (HGFilters.py:184-204).  Same tensors and
 * restrictions as mp_conv1x1. */
typedef struct mp_conv1x1_args {
  const float *x1;
  mp_gn_in gn1;
  int relu1;
  const float *x2;
  int n, c1, c2, cout;
  int64_t hw;
  const void *packed;
  int f16;
  const float *wmax;
  const float *bias;
  const float *res;
  float *y;
  float *y_hwc;
  mp_gn_out fin;
} mp_conv1x1_args;
int mp_conv1x1_ex(mp_ctx *ctx, const mp_conv1x1_args *args, mp_stream stream);
int mp_conv1x1_stat_slices(int64_t hw);

/* The encoders' other convolutions on the same MFMA scheme (csrc/convim2col.hip), with an explicit
 * im2col tile in LDS: ks = 7, 3 -> 64 channels, stride 2 + zero padding 3 + bias (the hourglass stem,
 * HGFilters.py:125, :168) or stride 1 + nn.ReflectionPad2d(3) (netC's stem, ResBlkFilters.py:111-113);
 * ks = 3, stride 2, zero padding 1, Cin % 16 == 0, Cout % 128 == 0 (netC's two down-sampling
 * convolutions, ResBlkFilters.py:115-121).  y [N, Cout, H/stride, W/stride], W/stride % 64 == 0.
 * gn / relu: GroupNorm (+ReLU) of the INPUT applied while gathering, as in mp_conv3x3_ex; bias may be
 * NULL; fin: GroupNorm(32, Cout) over y.  packed: mp_convk_packed_floats floats from mp_convk_pack. */
typedef struct mp_convk_args {
  const float *x;
  int n, cin, h, w;
  mp_gn_in gn;
  int relu, reflect;
  const float *packed;
  const float *bias;
  int cout, ks, stride;
  float *y;
  mp_gn_out fin;
} mp_convk_args;
int mp_convk_supported(int cin, int cout, int ks, int stride, int h, int w);
int64_t mp_convk_packed_floats(int cin, int cout, int ks);
int mp_convk_stat_slices(int ks, int stride, int h, int w);
int mp_convk_pack(mp_ctx *ctx, const float *w /*[Cout,Cin,ks,ks]*/, int cout, int cin, int ks, float *packed,
                  mp_stream stream);
int mp_convk(mp_ctx *ctx, const mp_convk_args *args, mp_stream stream);

/* The tensors between the convolutions, written together with their statistics (fin may be NULL;
 * slices of a legacy partial buffer = mp_gn_stat_slices(); C % 32 == 0):
 *   mp_avgpool2_gn            y [N,C,H/2,W/2] = avg_pool2d(x, 2, stride 2)  (HGFilters.py:93, :171), W % 8 == 0
 *   mp_upsample_bicubic2x_gn  y [N,C,2H,2W] = add + bicubic_x2(x)           (HGFilters.py:108-111), add may be NULL
 *   mp_gn_apply               y = [res +] relu?(GroupNorm(x)), gn as above  (the stem's GroupNorm + ReLU, :168;
 *                             x + GroupNorm(conv(.)) at the end of a residual block, ResBlkFilters.py:75-84) */
int mp_avgpool2_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, float *y, const mp_gn_out *fin,
                   mp_stream stream);
int mp_upsample_bicubic2x_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, const float *add, float *y,
                             const mp_gn_out *fin, mp_stream stream);
int mp_gn_apply(mp_ctx *ctx, const float *x, const mp_gn_in *gn, int relu, int n, int c, int64_t hw,
                const float *res, float *y, const mp_gn_out *fin, mp_stream stream);

/* ---- recorded launch sequences ------------------------------------------------------------------
 * The reference calls netG.filter once per frame from a stage thread (RTL/main.py:366-370): here that is ~137
 * launches of the entry points above.  A PLAN is that sequence recorded once -- the same argument structs, in
 * order, each with the stream slot it ran on -- and replayed by ONE call: no host interpreter between the
 * launches.  The recorder owns every buffer (inputs, intermediates, outputs: static for the life of the plan,
 * copy in and out around mp_plan_run); the plan stores pointers only.  Slot 0 is the stream passed to
 * mp_plan_run; slots 1..n_side_streams are streams the plan creates (the hourglass's skip branches run on side
 * streams at batch <= 2); MP_PLAN_WAIT makes one slot wait for the work enqueued so far on another.  Replays of
 * one plan must be ordered on one stream (its buffers are static).  mp_plan_run returns the first failing
 * command's status (mp_last_error); the commands after it are not enqueued, and the plan's side streams are joined
 * into `stream` before the call returns, so nothing of the partial replay runs unordered against the caller. */
typedef struct mp_plan mp_plan;
enum {
  MP_PLAN_CONVK = 1,      /* args: mp_convk_args            -> mp_convk */
  MP_PLAN_GN_APPLY = 2,   /* args: mp_plan_gn_apply_args    -> mp_gn_apply */
  MP_PLAN_CONV3X3 = 3,    /* args: mp_conv3x3_args          -> mp_conv3x3_ex */
  MP_PLAN_CONV1X1 = 4,    /* args: mp_conv1x1_args          -> mp_conv1x1_ex */
  MP_PLAN_AVGPOOL2 = 5,   /* args: mp_plan_pool_args        -> mp_avgpool2_gn */
  MP_PLAN_UPSAMPLE2X = 6, /* args: mp_plan_upsample_args    -> mp_upsample_bicubic2x_gn */
  MP_PLAN_MEMSET = 7,     /* args: mp_plan_memset_args      -> hipMemsetAsync (the GroupNorm accumulator arena) */
  MP_PLAN_WAIT = 8        /* args: mp_plan_wait_args        -> event record on one slot, wait on another */
};
typedef struct mp_plan_gn_apply_args {
  const float *x;
  mp_gn_in gn;
  int relu, n, c;
  int64_t hw;
  const float *res;
  float *y;
  mp_gn_out fin;
} mp_plan_gn_apply_args;
typedef struct mp_plan_pool_args {
  const float *x;
  int n, c, h, w;
  float *y;
  mp_gn_out fin;
} mp_plan_pool_args;
typedef struct mp_plan_upsample_args {
  const float *x;
  int n, c, h, w;
  const float *add;
  float *y;
  mp_gn_out fin;
} mp_plan_upsample_args;
typedef struct mp_plan_memset_args {
  void *ptr;
  int64_t bytes;
  int value;
} mp_plan_memset_args;
typedef struct mp_plan_wait_args {
  int waiter_slot, signaller_slot;
} mp_plan_wait_args;
int mp_plan_create(mp_ctx *ctx, int n_side_streams, mp_plan **out);
int mp_plan_add(mp_plan *plan, int kind, const void *args, int64_t bytes, int stream_slot);
int mp_plan_size(mp_plan *plan);
int mp_plan_run(mp_plan *plan, mp_stream stream);
void mp_plan_destroy(mp_plan *plan); /* drains and destroys the plan's side streams */

/* ---- measurement ------------------------------------------------------------------------------ */
/* Brackets every fused-query kernel launch made through this context with a pair of HIP events
 * recorded on the launch stream (bench.py's roofline leg).  mp_profile_end waits for the last
 * event and writes the elapsed milliseconds of up to `capacity` launches, in launch order, to
 * ms_out (host); *n_out = launches recorded. */
int mp_profile_begin(mp_ctx *ctx, int max_records);
int mp_profile_end(mp_ctx *ctx, float *ms_out /*host*/, int capacity, int *n_out /*host*/);

/* What the f32 matrix pipe of this device sustains right now and at which shader clock: a register-only loop of
 * v_mfma_f32_32x32x2_f32 on random mantissas, two 4-wave workgroups per CU, about ms_target (1..2000) milliseconds,
 * timed with HIP events; every workgroup reads the shader clock counter and the 100 MHz reference around its loop.
 * out4 (host): [0] TFLOP/s, [1] average shader clock in MHz during the launch, [2] the launch's ms, [3] workgroups.
 * Synchronises the stream and holds the context's mutex while it runs (every other call on this context waits up
 * to ms_target): a set-up / measurement call, not one for a running pipeline.  On a stream of
 * mp_stream_create_cu_mask it measures that stream's share of the device.  bench.py reports it as `roofline.sustained` next to the nominal peak (boxes of one pool
 * hold clocks a few per cent apart under a matrix load).  No counterpart in the reference. */
int mp_mfma_clock_probe(mp_ctx *ctx, float ms_target, double *out4 /*host*/, mp_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOPORT_HIP_H */
