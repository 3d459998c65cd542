// Internal declarations shared by the HIP translation units of libmonoport_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/monoport_hip.h"

namespace mp {

constexpr int kTilePts = 64;     // query points per workgroup tile
constexpr int kQueryThreads = 256;
constexpr int kHidden[4] = {1024, 512, 256, 128};  // SurfaceClassifier.py:76 / :84
// rows of a feature map's skip table (mp_skip_table): the feature segments of layers 0-3 back to back
// (1024 + 512 + 256 + 128), then the last layer's (<= 3 outputs, padded to one 16-byte slot)
constexpr int kTableL[5] = {0, 1024, 1536, 1792, 1920};
// 1923 used floats, padded to 61 x 32: a texel's row is then 61 cache lines of 128 bytes and every 32-row block a
// query lane group reads (128 bytes) is ONE line -- with the minimal padding (1924) a texel's row started 16 bytes
// further every texel and 7 of 8 blocks straddled two lines: twice the lines per gathered block (round 4)
constexpr int kTableRows = 1952;

// Device-side view of one packed SurfaceClassifier (see pack.hip for the fragment order).
// One base pointer + float offsets keeps the kernel's SGPR footprint small.
struct MlpPack {
  const float *base;
  int ah[4];    // hidden-segment A fragments of MFMA layers 0..3 (ah[0] unused)
  int ax[4];    // feature-segment A fragments
  int az[4];    // z-column A fragments [n_out/32][64]
  int bias[5];  // biases of layers 0..4
  int w4;       // last layer, row-major [Cout][pad4(128 + C + 1)]
  int n_floats; // size of the whole packed buffer (buffer-resource bound of the weight loads)
};

// Split-precision copy of the same weights for the 3-term f16 MFMA kernel (query16.hip): every
// weight w is stored as hi = f16(w * S), lo = f16(w * S - hi) with a per-layer power-of-two scale
// S; fragments are [rb][g][hi|lo][lane] of 8 halves (16 bytes), K walked in groups of 16.
struct MlpPack16 {
  const void *base;   // half data, offsets below in units of 16 bytes
  int ah[4], ax[4];   // hidden / feature segments
  int az[4];          // z column: [rb][hi|lo][lane], only element 0 of lanes 0-31 non-zero
  float scale[4];     // S of layers 0..3
  int n16;            // size of the buffer in 16-byte units (buffer-resource bound)
};

// Where a query launch takes its points from and where it puts the results.
struct PointSrc {
  // explicit mode (packed == nullptr): world coords, element (c, i) at pts[i*sn + c*sc]
  const float *pts;
  long long sn, sc;
  // lattice mode: packed level-local node coords x | y<<10 | z<<20, scaled by `stride` into the
  // final-resolution index space and mapped to world like Seg3dLossless.batch_eval
  const uint32_t *packed;
  int stride;       // final-res index step of this level
  int level_res;    // nodes per axis at this level (scatter index = (z*res + y)*res + x)
  float res_final;  // R as float
  float half_step;  // (1/R)/2
  float bmin[3], blen[3];
  // count: *n_dev if n_dev != nullptr else n
  const int32_t *n_dev;
  long long n;
  long long out_stride;  // explicit mode: out[o*out_stride + i]
};

// One launch of the fused query kernels serves up to kMaxFrames independent frames (their own
// feature map, calibration, points and output): the tiles of all frames form one index space, so
// the small coarse levels of several frames fill the machine together.
constexpr int kMaxFrames = 32;  // a power of two <= 64 (query_table.hip: one lane per frame)
static_assert((kMaxFrames & (kMaxFrames - 1)) == 0 && kMaxFrames <= 64, "kMaxFrames");
struct QueryItem {
  const float *feat;   // channels-last feature map [H,W,C]
  const float *calib;  // [3,4] rows of the 4x4
  float *out;
  PointSrc src;
  const float *l0;     // optional skip table of `feat` [H,W,kTableRows] (mp_skip_table; filled in by the launcher)
};
// host-side description of a launch (the launchers turn it into a QuerySetDev)
struct QuerySet {
  int n;
  QueryItem it[kMaxFrames];
};

// What the kernels receive (kernel arguments are limited to 4 KB: 32 full QueryItems would be 4.1 KB).
// The frames of one launch share the point layout (explicit strides, or the lattice of one octree level);
// per frame only the pointers and the count differ.
struct QueryItemDev {
  const float *feat, *calib;
  float *out;
  const float *l0;
  const void *pts;        // PointSrc::packed when `lattice`, else PointSrc::pts
  const int32_t *n_dev;
  long long n;
};
struct QuerySetDev {
  int n;
  int lattice;
  long long sn, sc, out_stride;
  int stride, level_res;
  float res_final, half_step, bmin[3], blen[3];
  QueryItemDev it[kMaxFrames];
#ifdef __HIPCC__
  __device__ __forceinline__ long long count(int f) const {
    const int32_t *p = it[f].n_dev;
    return p ? (long long)*p : it[f].n;
  }
  __device__ __forceinline__ QueryItem item(int f) const {
    QueryItem q;
    const QueryItemDev &d = it[f];
    q.feat = d.feat;
    q.calib = d.calib;
    q.out = d.out;
    q.l0 = d.l0;
    q.src.pts = lattice ? nullptr : static_cast<const float *>(d.pts);
    q.src.packed = lattice ? static_cast<const uint32_t *>(d.pts) : nullptr;
    q.src.sn = sn;
    q.src.sc = sc;
    q.src.out_stride = out_stride;
    q.src.stride = stride;
    q.src.level_res = level_res;
    q.src.res_final = res_final;
    q.src.half_step = half_step;
    for (int i = 0; i < 3; ++i) {
      q.src.bmin[i] = bmin[i];
      q.src.blen[i] = blen[i];
    }
    q.src.n_dev = d.n_dev;
    q.src.n = d.n;
    return q;
  }
#endif
};
static_assert(sizeof(QuerySetDev) + 256 <= 4096, "kernel arguments: the set + two MLP descriptors must stay below 4 KB");

struct Mlp {
  bool used = false;
  int c = 0, cout = 0, act = 0;
  bool loaded[5] = {false, false, false, false, false};
  hipStream_t load_stream = nullptr;  // stream of the most recent mp_mlp_load
  float *buf = nullptr;  // one allocation holding every packed segment
  size_t off_ah[4], off_ax[4], off_az[4], off_bias[5], off_w4;
  size_t total = 0;
  MlpPack pack() const;
  // f16x3 copy (C == 256 heads only)
  int precision = 0;        // MP_PREC_F32 / MP_PREC_F16X3
  void *buf16 = nullptr;
  float *raw = nullptr;     // un-packed [out,in] copies of layers 0..3 (source for re-packing)
  size_t off_raw[4];
  size_t off16_ah[4], off16_ax[4], off16_az[4], total16 = 0;
  float scale16[4] = {1.f, 1.f, 1.f, 1.f};
  MlpPack16 pack16() const;
};

// mcubes.hip
size_t mc_scratch_bytes(int r);
int launch_marching_cubes(mp_ctx *ctx, void *scratch, const float *vol, int r, float level,
                          const float *bmin, const float *bmax, float *verts, long long max_v,
                          int32_t *faces, long long max_f, int32_t *counts, hipStream_t st);

}  // namespace mp

struct mp_ctx {
  int device = 0;
  int n_cu = 0;
  std::mutex mu;
  std::vector<mp::Mlp> mlps;
  // scratch arenas, one per stream the context has been used on (grown on demand): calls that
  // are in flight on different streams never share scratch; calls on one stream are ordered.
  struct Arena {
    void *ptr = nullptr;  // current block
    size_t bytes = 0;
    std::vector<void *> retired;  // outgrown blocks, kept until mp_stream_release / mp_destroy
    size_t retired_bytes = 0;
  };
  std::unordered_map<void *, Arena> arenas;
  // streams made by mp_stream_create_cu_mask -> compute units in their mask (guarded by mu)
  std::unordered_map<void *, int> stream_cus;
  // kernels whose dynamic-LDS limit has been raised on this context's device (guarded by mu)
  std::unordered_set<const void *> lds_attr_done;
  // skip tables registered for feature maps (mp_skip_table): feat pointer -> table + the packed
  // head it was computed with
  struct SkipTable {
    const float *table;
    const float *mlp_buf;
    int h, w;
  };
  std::unordered_map<const float *, SkipTable> skip_tables;
  // optional event bracketing of query launches (mp_profile_begin / mp_profile_end)
  std::vector<hipEvent_t> prof_events;  // start/stop pairs
  int prof_used = 0;                    // pairs recorded
};

namespace mp {

int fail(mp_ctx *ctx, int code, const char *fmt, ...);
// compute units a launch on `st` may use: the share of a CU-masked stream, else the whole device
inline int cus_of(const mp_ctx *ctx, hipStream_t st) {
  if (!ctx->stream_cus.empty()) {
    auto it = ctx->stream_cus.find((void *)st);
    if (it != ctx->stream_cus.end()) return it->second;
  }
  return ctx->n_cu;
}
int ensure_scratch(mp_ctx *ctx, hipStream_t st, size_t bytes, void **out);

#define MP_HIP(ctx, expr)                                                            \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess)                                                            \
      return mp::fail(ctx, MP_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                           \
  } while (0)

// clock_probe.hip
int launch_mfma_clock_probe(mp_ctx *ctx, float ms_target, double *out, hipStream_t st);

// query.hip
int launch_query(mp_ctx *ctx, const Mlp &m, const float *feat_hwc, int h, int w,
                 const float *calib, float z_scale, const PointSrc &src, float *out,
                 long long max_points, hipStream_t st);
int launch_query_set(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                     long long max_points, bool device_counts, hipStream_t st);
int launch_index(mp_ctx *ctx, const float *feat_hwc, int c, int h, int w, const float *uv,
                 long long n, float *out, hipStream_t st);
int launch_orthogonal(mp_ctx *ctx, const float *pts, long long n, const float *calib, float *out,
                      hipStream_t st);
// pack.hip
int launch_pack_hwc(mp_ctx *ctx, const float *src, int c_src, int h, int w, float *dst, int c_dst,
                    int c_off, hipStream_t st);
int launch_pack_layer(mp_ctx *ctx, Mlp &m, int layer, const float *w, const float *b,
                      hipStream_t st);
int launch_pack_layer16(mp_ctx *ctx, Mlp &m, int layer, const float *w, hipStream_t st);
int launch_copy(mp_ctx *ctx, const float *src, float *dst, long long n, hipStream_t st);
int launch_absmax(mp_ctx *ctx, const float *src, long long n, unsigned int *out_bits, hipStream_t st);
int launch_absmax_accumulate(mp_ctx *ctx, const float *src, long long n, unsigned int *out_bits,
                             hipStream_t st);
bool find_skip_tables(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, QuerySet &tset);
// host set -> kernel argument; MP_ERR_ARG if the frames do not share point layout / lattice
int compact_query_set(mp_ctx *ctx, const QuerySet &set, QuerySetDev &dset);
// query_small.hip: the netG f32 query on 32-point tiles, for launches of fewer than
// kSmallGateTiles 64-point tiles
constexpr int kSmallGateTiles = 2048;
int launch_query32(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                   long long max_points, bool device_counts, int gate_tiles64, hipStream_t st);
// query_table.hip: the skip table of a feature map and the query kernel that blends its rows
int launch_skip_table(mp_ctx *ctx, const Mlp &m, const float *feat_hwc, int h, int w, float *table,
                    hipStream_t st);
int launch_query_tab(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                     long long max_points, bool device_counts, hipStream_t st);
void query_small_set_gate(int gate);  // 0 never, 1 always, n > 1 gate in 64-point tiles, < 0 default
int query_small_gate();
// query16.hip
int launch_query16(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                   long long max_points, bool device_counts, hipStream_t st);
int launch_concat3_add(mp_ctx *ctx, const float *a, int ca, const float *b, int cb, const float *c,
                       int cc, const float *sc, int n, long long hw, float *y, hipStream_t st);
int launch_prepare_inputs(mp_ctx *ctx, const float *segm, long long hw, const float *mean,
                          const float *std, float *g, float *c, hipStream_t st);
// octree.hip
size_t recon_scratch_bytes(const int *res, int n_levels);
int launch_recon(mp_ctx *ctx, void *scratch, const Mlp &m, int n_frames,
                 const float *const *feat_hwc, int h, int w, const float *const *calib,
                 float z_scale, const float *bmin, const float *bmax, const int *res, int n_levels,
                 float balance, int final_level, float *const *volume, int32_t *const *status,
                 const mp_recon_early *early, hipStream_t st);
int launch_octree_select(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                         const unsigned long long *ev_prev, unsigned long long *ev_cur,
                         unsigned long long *bnd, int box, float balance, uint32_t *packed,
                         int32_t *count, hipStream_t st);
int octree_box_of_level(int level);
int launch_octree_conflicts(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, long long cap,
                            int r, const float *values, const float *vol, float balance,
                            unsigned long long *ev, uint32_t *out, int32_t *out_count,
                            hipStream_t st);
int launch_lattice_points(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, long long cap,
                          int stride, int res_final, const float *bmin, const float *bmax,
                          float *pts, hipStream_t st);
int launch_scatter_nodes(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, long long cap,
                         int r, const float *values, float *vol, hipStream_t st);
// vertices.hip
int launch_forward_vertices(mp_ctx *ctx, void *scratch, const float *vol, int r, int dir, int64_t *x, int64_t *y,
                            float *z, float *norm, int32_t *count, hipStream_t st);
size_t forward_vertices_scratch_bytes(int r);
int launch_forward_vertices_batch(mp_ctx *ctx, void *scratch, int n_frames, const float *const *vol, int r, int dir,
                                  int64_t *const *x, int64_t *const *y, float *const *z, float *const *norm,
                                  int32_t *const *count, hipStream_t st);
int launch_paint_batch(mp_ctx *ctx, int n_frames, const int64_t *const *x, const int64_t *const *y,
                       const float *const *vals, int ch_major, const int32_t *const *count, long long cap, int res,
                       float scale, float bias, float lo, float hi, float *const *image, hipStream_t st);
int launch_vertex_points(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *z,
                         const int32_t *count, long long cap, int res, const float *mat16,
                         float *pts, hipStream_t st);
int launch_paint(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *vals, int ch_major,
                 const int32_t *count, long long cap, int res, float scale, float bias, float lo,
                 float hi, float *image, hipStream_t st);

int launch_visualize(mp_ctx *ctx, const float *image, int res, int size, float *out, uint8_t *mask,
                     hipStream_t st);
// mcubes.hip
size_t mc_scratch_bytes(int r);
int launch_marching_cubes(mp_ctx *ctx, void *scratch, const float *vol, int r, float level,
                          const float *bmin, const float *bmax, float *verts, long long max_v,
                          int32_t *faces, long long max_f, int32_t *counts, hipStream_t st);

// conv3x3.hip
int launch_conv3x3_pack(mp_ctx *ctx, const float *w, int cout, int cin, float *wp, hipStream_t st);
int conv3x3_stat_slices(int cout, int n, int h, int w, bool f16);
void conv3x3_set_nr(int nr);
bool conv3x3_supported(int cin, int cout, int h, int w);
int launch_conv3x3_pack16(mp_ctx *ctx, const float *w, int cout, int cin, void *wp, float *wmax,
                          hipStream_t st);
int launch_conv3x3_gn(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss,
                      int relu, int reflect, const float *wp, const float *wmax16, int cout, float *y,
                      double *stats, hipStream_t st);
int launch_scale_shift_add(mp_ctx *ctx, const float *t, const float *ss, const float *res, long long planes,
                           long long hw, float *y, hipStream_t st);
int launch_conv1x1_pack(mp_ctx *ctx, const float *w1, int c1, const float *w2, int c2, int cout, int f16,
                        void *wp, float *wmax, hipStream_t st);
int launch_conv1x1_raw(mp_ctx *ctx, const float *x1, const float *ss1, int relu1, const float *x2, int n,
                       int c1, int c2, int cout, long long hw, const void *wp, int f16, const float *wmax,
                       const float *bias, const float *res, float *y, float *y_hwc, double *stats,
                       hipStream_t st);
int launch_gn_finalize(mp_ctx *ctx, const double *partial, int n, int c, int groups, int slices,
                       double count, const float *gamma, const float *beta, float eps, float *ss,
                       hipStream_t st);
// encoder_ops.hip
int gn_stat_slices();
int launch_gn_stats(mp_ctx *ctx, const float *x, int n, int c, long long hw, int groups,
                    double *partial, hipStream_t st);
size_t gn_scratch_bytes(int groups);
int launch_group_norm(mp_ctx *ctx, void *scratch, const float *x, int n, int c, long long hw,
                      int groups, const float *gamma, const float *beta, float eps, int relu,
                      float *y, hipStream_t st);
int launch_upsample_bicubic2x(mp_ctx *ctx, const float *x, int c, int h, int w, const float *add,
                              float *y, hipStream_t st);

}  // namespace mp
