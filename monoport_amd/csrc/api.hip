// C-ABI entry points of libmonoport_hip.so (see include/monoport_hip.h for the contract).
#include <cstdarg>
#include <cmath>
#include <cstring>

#include "mp_internal.h"
#include "encoder_kernels.h"

namespace mp {

// The last error message is kept PER HOST THREAD (like errno): the stage threads of a pipeline
// share one context, and a message must not be overwritten -- or its storage reallocated --
// between a failing call and the caller's mp_last_error on the same thread.
static thread_local std::string g_last_error;

int fail(mp_ctx *ctx, int code, const char *fmt, ...) {
  (void)ctx;
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// Scratch grows by ADDING a block: the previous block stays allocated until mp_stream_release /
// mp_destroy, because a hipGraph captured on this stream (the encoder graph of a pipeline slot
// bakes mp_group_norm's scratch pointer) or work still queued on it may reference it.  Sizes are
// rounded up geometrically so a stream retires at most a handful of blocks.
int ensure_scratch(mp_ctx *ctx, hipStream_t st, size_t bytes, void **out) {
  mp_ctx::Arena &a = ctx->arenas[(void *)st];
  if (bytes > a.bytes) {
    size_t want = a.bytes + a.bytes / 2;
    if (want < bytes) want = bytes;
    want = (want + 0xFFFF) & ~size_t(0xFFFF);
    void *p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();  // the failed attempt must not surface as the next launch's error
      if (want == bytes || hipMalloc(&p, want = bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, MP_ERR_NOMEM, "scratch arena: hipMalloc(%zu) failed", want);
      }
    }
    if (a.ptr) {
      a.retired.push_back(a.ptr);
      a.retired_bytes += a.bytes;
    }
    a.ptr = p;
    a.bytes = want;
  }
  *out = a.ptr;
  return MP_OK;
}

MlpPack Mlp::pack() const {
  MlpPack p;
  p.base = buf;
  for (int l = 0; l < 4; ++l) {
    p.ah[l] = (int)off_ah[l];
    p.ax[l] = (int)off_ax[l];
    p.az[l] = (int)off_az[l];
  }
  for (int l = 0; l < 5; ++l) p.bias[l] = (int)off_bias[l];
  p.w4 = (int)off_w4;
  p.n_floats = (int)total;
  return p;
}

MlpPack16 Mlp::pack16() const {
  MlpPack16 p;
  p.base = buf16;
  for (int l = 0; l < 4; ++l) {
    p.ah[l] = (int)off16_ah[l];
    p.ax[l] = (int)off16_ax[l];
    p.az[l] = (int)off16_az[l];
    p.scale[l] = scale16[l];
  }
  p.n16 = (int)total16;
  return p;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

static Mlp *get_mlp(mp_ctx *ctx, int id) {
  if (id < 0 || id >= (int)ctx->mlps.size() || !ctx->mlps[id].used) return nullptr;
  return &ctx->mlps[id];
}

static int check_ready(mp_ctx *ctx, const Mlp *m, int c) {
  if (!m) return fail(ctx, MP_ERR_ARG, "unknown mlp id");
  for (int l = 0; l < 5; ++l)
    if (!m->loaded[l]) return fail(ctx, MP_ERR_STATE, "mlp layer %d has not been loaded", l);
  if (m->c != c)
    return fail(ctx, MP_ERR_ARG, "feature map has C=%d but the mlp was built for C=%d", c, m->c);
  return MP_OK;
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_version(void) { return 100; }

int mp_create(int device, mp_ctx **out) {
  if (!out) return fail(nullptr, MP_ERR_ARG, "mp_create: out is NULL");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(nullptr, MP_ERR_HIP, "mp_create: no HIP device visible (there is no CPU fallback)");
  if (device < 0 || device >= count)
    return fail(nullptr, MP_ERR_ARG, "mp_create: device %d out of range (%d visible)", device, count);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return fail(nullptr, MP_ERR_HIP, "mp_create: hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, MP_ERR_UNSUPPORTED, "mp_create: kernels are built for gfx950 only, device is %s",
                prop.gcnArchName);
  mp_ctx *ctx = new mp_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  *out = ctx;
  return MP_OK;
}

void mp_destroy(mp_ctx *ctx) {
  if (!ctx) return;
  {
    DeviceGuard g(ctx->device);
    for (auto &m : ctx->mlps) {
      if (m.buf) (void)hipFree(m.buf);
      if (m.buf16) (void)hipFree(m.buf16);
      if (m.raw) (void)hipFree(m.raw);
    }
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    for (auto &kv : ctx->arenas) {
      if (kv.second.ptr) (void)hipFree(kv.second.ptr);
      for (void *p : kv.second.retired) (void)hipFree(p);
    }
  }
  delete ctx;
}

const char *mp_last_error(mp_ctx *ctx) {
  (void)ctx;
  return g_last_error.c_str();
}

int mp_stream_release(mp_ctx *ctx, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->arenas.find((void *)stream);
  if (it == ctx->arenas.end()) return MP_OK;
  DeviceGuard g(ctx->device);
  // the stream may already be destroyed (that is when this is called): drain the device instead
  MP_HIP(ctx, hipDeviceSynchronize());
  // free every block and forget the arena even if one hipFree fails (mp_destroy must not free twice)
  hipError_t first = hipSuccess;
  if (it->second.ptr) first = hipFree(it->second.ptr);
  for (void *p : it->second.retired) {
    const hipError_t e = hipFree(p);
    if (first == hipSuccess) first = e;
  }
  ctx->arenas.erase(it);
  MP_HIP(ctx, first);
  return MP_OK;
}

int mp_stream_create_cu_mask(mp_ctx *ctx, int first_cu, int n_cus, mp_stream *out) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!out || first_cu < 0 || n_cus < 8 || first_cu + n_cus > ctx->n_cu)
    return fail(ctx, MP_ERR_ARG, "mp_stream_create_cu_mask: CUs [%d, %d) of %d", first_cu, first_cu + n_cus, ctx->n_cu);
  DeviceGuard g(ctx->device);
  std::vector<uint32_t> mask((ctx->n_cu + 31) / 32, 0u);
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  MP_HIP(ctx, hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  ctx->stream_cus[(void *)st] = n_cus;
  *out = (mp_stream)st;
  return MP_OK;
}

int mp_stream_destroy(mp_ctx *ctx, mp_stream stream) {
  if (!ctx || !stream) return MP_ERR_ARG;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->stream_cus.find((void *)stream);
    if (it == ctx->stream_cus.end())
      return fail(ctx, MP_ERR_ARG, "mp_stream_destroy: not a stream of mp_stream_create_cu_mask on this context");
    ctx->stream_cus.erase(it);
  }
  const int rc = mp_stream_release(ctx, stream);  // drains the device, frees the stream's arena
  DeviceGuard g(ctx->device);
  MP_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
  return rc;
}

int mp_stream_cu_count(mp_ctx *ctx, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return cus_of(ctx, (hipStream_t)stream);
}

int mp_memory_stats(mp_ctx *ctx, int64_t *out4) {
  if (!ctx || !out4) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  int64_t arena = 0, weights = 0;
  for (const auto &kv : ctx->arenas) arena += (int64_t)(kv.second.bytes + kv.second.retired_bytes);
  for (const Mlp &m : ctx->mlps) {
    if (!m.used) continue;
    if (m.buf) weights += (int64_t)m.total * 4;
    if (m.raw) weights += (int64_t)(m.off_raw[3] + (size_t)kHidden[3] * (kHidden[2] + m.c + 1)) * 4;
    if (m.buf16) weights += (int64_t)m.total16 * 16;
  }
  out4[0] = arena;
  out4[1] = weights;
  out4[2] = (int64_t)ctx->arenas.size();
  out4[3] = (int64_t)ctx->skip_tables.size();
  return MP_OK;
}

int mp_max_frames(void) { return kMaxFrames; }

int mp_mlp_create(mp_ctx *ctx, int n_layers, const int *channels, int last_op, int *mlp_out) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!channels || !mlp_out) return fail(ctx, MP_ERR_ARG, "mp_mlp_create: NULL argument");
  if (last_op < MP_ACT_NONE || last_op > MP_ACT_TANH)
    return fail(ctx, MP_ERR_ARG, "mp_mlp_create: bad last_op %d", last_op);
  if (n_layers != 5)
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_mlp_create: only the 5-layer PIFu heads are built (got %d)",
                n_layers);
  const int c = channels[0] - 1, cout = channels[5];
  bool ok = (c == 256 || c == 512) && (cout == 1 || cout == 3);
  for (int l = 0; l < 4; ++l) ok = ok && channels[l + 1] == kHidden[l];
  if (!ok)
    return fail(ctx, MP_ERR_UNSUPPORTED,
                "mp_mlp_create: channels must be {C+1,1024,512,256,128,Cout}, C in {256,512}, Cout in {1,3}");
  DeviceGuard g(ctx->device);
  Mlp m;
  m.used = true;
  m.c = c;
  m.cout = cout;
  m.act = last_op;
  size_t off = 0;
  auto take = [&](size_t n) {
    size_t o = off;
    off += (n + 63) & ~size_t(63);
    return o;
  };
  for (int l = 0; l < 4; ++l) {
    const size_t n_out = kHidden[l], k_h = l == 0 ? 0 : kHidden[l - 1];
    m.off_ah[l] = take(n_out * k_h);
    m.off_ax[l] = take(n_out * c);
    m.off_az[l] = take(n_out * 2);
    m.off_bias[l] = take(n_out);
  }
  m.off_w4 = take((size_t)cout * ((kHidden[3] + c + 1 + 3) & ~3));
  m.off_bias[4] = take(cout);
  m.total = off;
  if (hipMalloc(reinterpret_cast<void **>(&m.buf), off * sizeof(float)) != hipSuccess)
    return fail(ctx, MP_ERR_NOMEM, "mp_mlp_create: hipMalloc of %zu floats failed", off);
  // raw copies (source for re-packing into other operand formats)
  size_t roff = 0;
  for (int l = 0; l < 4; ++l) {
    m.off_raw[l] = roff;
    roff += (size_t)kHidden[l] * ((l == 0 ? 0 : kHidden[l - 1]) + c + 1);
  }
  if (hipMalloc(reinterpret_cast<void **>(&m.raw), roff * sizeof(float)) != hipSuccess) {
    (void)hipFree(m.buf);
    return fail(ctx, MP_ERR_NOMEM, "mp_mlp_create: hipMalloc of %zu floats failed", roff);
  }
  int id = -1;
  for (size_t i = 0; i < ctx->mlps.size(); ++i)
    if (!ctx->mlps[i].used) id = (int)i;
  if (id < 0) {
    ctx->mlps.push_back(m);
    id = (int)ctx->mlps.size() - 1;
  } else {
    ctx->mlps[id] = m;
  }
  *mlp_out = id;
  return MP_OK;
}

int mp_mlp_load(mp_ctx *ctx, int mlp, int layer, const float *W, const float *b, int out_ch,
                int in_ch, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  Mlp *m = get_mlp(ctx, mlp);
  if (!m) return fail(ctx, MP_ERR_ARG, "mp_mlp_load: unknown mlp id %d", mlp);
  if (!W || !b || layer < 0 || layer > 4) return fail(ctx, MP_ERR_ARG, "mp_mlp_load: bad argument");
  const int want_out = layer < 4 ? kHidden[layer] : m->cout;
  const int want_in = (layer == 0 ? 0 : kHidden[layer - 1]) + m->c + 1;
  if (out_ch != want_out || in_ch != want_in)
    return fail(ctx, MP_ERR_ARG, "mp_mlp_load: layer %d expects weight [%d,%d], got [%d,%d]", layer,
                want_out, want_in, out_ch, in_ch);
  DeviceGuard g(ctx->device);
  // skip tables hold products of THESE weights (layer 0 + the skip segments): a table made before
  // the reload would blend the old head into the new one's hidden layers -- forget them
  for (auto it = ctx->skip_tables.begin(); it != ctx->skip_tables.end();)
    it = it->second.mlp_buf == m->buf ? ctx->skip_tables.erase(it) : std::next(it);
  int rc = launch_pack_layer(ctx, *m, layer, W, b, (hipStream_t)stream);
  if (rc == MP_OK && layer < 4)
    rc = launch_copy(ctx, W, m->raw + m->off_raw[layer], (long long)out_ch * in_ch,
                     (hipStream_t)stream);
  if (rc == MP_OK) {
    m->loaded[layer] = true;
    m->load_stream = (hipStream_t)stream;  // mp_mlp_set_precision orders itself behind it
  }
  if (rc == MP_OK && m->precision != MP_PREC_F32) {
    // weights changed under an f16-packed MLP: drop back to f32 until precision is selected again
    m->precision = MP_PREC_F32;
  }
  return rc;
}

int mp_mlp_set_precision(mp_ctx *ctx, int mlp, int precision) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  Mlp *m = get_mlp(ctx, mlp);
  if (!m) return fail(ctx, MP_ERR_ARG, "mp_mlp_set_precision: unknown mlp id %d", mlp);
  if (precision < MP_PREC_F32 || precision > MP_PREC_F16)
    return fail(ctx, MP_ERR_ARG, "mp_mlp_set_precision: bad precision %d", precision);
  if (precision == MP_PREC_F32) {
    m->precision = MP_PREC_F32;
    return MP_OK;
  }
  int rc = check_ready(ctx, m, m->c);
  if (rc != MP_OK) return rc;
  if (m->c != 256)
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_mlp_set_precision: the f16 kernels are built for C = 256 heads only");
  DeviceGuard g(ctx->device);
  if (!m->buf16) {
    size_t off = 0;  // units of 16 bytes (8 halves)
    for (int l = 0; l < 4; ++l) {
      const size_t n_rb = kHidden[l] / 32, k_h = l == 0 ? 0 : kHidden[l - 1];
      m->off16_ah[l] = off;
      off += n_rb * (k_h / 16) * 128;
      m->off16_ax[l] = off;
      off += n_rb * (m->c / 16) * 128;
      m->off16_az[l] = off;
      off += n_rb * 128;
    }
    if (hipMalloc(&m->buf16, off * 16) != hipSuccess)
      return fail(ctx, MP_ERR_NOMEM, "mp_mlp_set_precision: hipMalloc(%zu) failed", off * 16);
    m->total16 = off;
  }
  // The raw weight copies were written by mp_mlp_load on the caller's stream; torch's side
  // streams do not synchronise with the NULL stream, so the re-pack runs on that same stream
  // (ordered behind the loads) and is drained before returning.
  const hipStream_t st = m->load_stream;
  unsigned int *d_bits = nullptr;
  MP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d_bits), sizeof(unsigned int)));
  for (int l = 0; l < 4 && rc == MP_OK; ++l) {
    const long long n = (long long)kHidden[l] * ((l == 0 ? 0 : kHidden[l - 1]) + m->c + 1);
    rc = launch_absmax(ctx, m->raw + m->off_raw[l], n, d_bits, st);
    unsigned int bits = 0;
    if (rc == MP_OK &&
        (hipMemcpyAsync(&bits, d_bits, sizeof(bits), hipMemcpyDeviceToHost, st) != hipSuccess ||
         hipStreamSynchronize(st) != hipSuccess))
      rc = fail(ctx, MP_ERR_HIP, "mp_mlp_set_precision: reading max|W| back failed");
    float wmax;
    memcpy(&wmax, &bits, sizeof(wmax));
    // largest power of two S with max|w| * S <= 2^14 (f16 tops out at 65504), clamped
    int e = 0;
    if (wmax > 0.0f && wmax < 3.0e38f) {
      (void)frexpf(wmax, &e);  // wmax = f * 2^e, f in [0.5, 1)
      e = 14 - e;
    }
    if (e > 14) e = 14;
    if (e < -14) e = -14;
    m->scale16[l] = ldexpf(1.0f, e);
    if (rc == MP_OK) rc = launch_pack_layer16(ctx, *m, l, m->raw + m->off_raw[l], st);
  }
  // drained: the packed f16 weights are complete before any stream can launch a query on them
  if (hipStreamSynchronize(st) != hipSuccess && rc == MP_OK)
    rc = fail(ctx, MP_ERR_HIP, "mp_mlp_set_precision: hipStreamSynchronize failed");
  (void)hipFree(d_bits);
  if (rc == MP_OK) m->precision = precision;
  return rc;
}

int mp_mlp_destroy(mp_ctx *ctx, int mlp) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  Mlp *m = get_mlp(ctx, mlp);
  if (!m) return fail(ctx, MP_ERR_ARG, "mp_mlp_destroy: unknown mlp id %d", mlp);
  DeviceGuard g(ctx->device);
  MP_HIP(ctx, hipDeviceSynchronize());
  // skip tables made with this head die with it (a later head may get the same buffer address)
  for (auto it = ctx->skip_tables.begin(); it != ctx->skip_tables.end();)
    it = it->second.mlp_buf == m->buf ? ctx->skip_tables.erase(it) : std::next(it);
  if (m->buf) MP_HIP(ctx, hipFree(m->buf));
  if (m->buf16) MP_HIP(ctx, hipFree(m->buf16));
  if (m->raw) MP_HIP(ctx, hipFree(m->raw));
  *m = Mlp();
  return MP_OK;
}

int mp_feat_pack_hwc(mp_ctx *ctx, const float *src_chw, int c_src, int h, int w, float *dst_hwc,
                     int c_dst, int c_offset, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!src_chw || !dst_hwc || c_src <= 0 || h <= 0 || w <= 0 || c_offset < 0 ||
      c_offset + c_src > c_dst)
    return fail(ctx, MP_ERR_ARG, "mp_feat_pack_hwc: bad argument");
  DeviceGuard g(ctx->device);
  return launch_pack_hwc(ctx, src_chw, c_src, h, w, dst_hwc, c_dst, c_offset, (hipStream_t)stream);
}

static_assert(MP_SKIP_TABLE_ROWS == mp::kTableRows, "header and kernels disagree on the table row length");

int mp_skip_table(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w, float *table,
                mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  int rc = check_ready(ctx, m, c);
  if (rc != MP_OK) return rc;
  if (!feat_hwc || !table || h <= 0 || w <= 0) return fail(ctx, MP_ERR_ARG, "mp_skip_table: bad argument");
  if (!aligned16(feat_hwc) || !aligned16(table))
    return fail(ctx, MP_ERR_ARG, "mp_skip_table: feat_hwc and table must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  rc = launch_skip_table(ctx, *m, feat_hwc, h, w, table, (hipStream_t)stream);
  if (rc != MP_OK) return rc;
  ctx->skip_tables[feat_hwc] = mp_ctx::SkipTable{table, m->buf, h, w};
  return MP_OK;
}

int mp_skip_table_batch(mp_ctx *ctx, int mlp, int n_maps, const float *feat_hwc, int c, int h, int w,
                      float *table, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  int rc = check_ready(ctx, m, c);
  if (rc != MP_OK) return rc;
  if (!feat_hwc || !table || h <= 0 || w <= 0 || n_maps <= 0 || (long long)n_maps * h > (1 << 20))
    return fail(ctx, MP_ERR_ARG, "mp_skip_table_batch: bad argument");
  if (!aligned16(feat_hwc) || !aligned16(table))
    return fail(ctx, MP_ERR_ARG, "mp_skip_table_batch: feat_hwc and table must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  // the maps are contiguous: one launch over n_maps * H rows of texels
  rc = launch_skip_table(ctx, *m, feat_hwc, n_maps * h, w, table, (hipStream_t)stream);
  if (rc != MP_OK) return rc;
  for (int i = 0; i < n_maps; ++i)
    ctx->skip_tables[feat_hwc + (size_t)i * h * w * c] =
        mp_ctx::SkipTable{table + (size_t)i * h * w * kTableRows, m->buf, h, w};
  return MP_OK;
}

int mp_skip_table_release(mp_ctx *ctx, const float *feat_hwc, const float *table) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!feat_hwc) {
    ctx->skip_tables.clear();
    return MP_OK;
  }
  auto it = ctx->skip_tables.find(feat_hwc);
  if (it != ctx->skip_tables.end() && (!table || it->second.table == table)) ctx->skip_tables.erase(it);
  return MP_OK;
}

int mp_index(mp_ctx *ctx, const float *feat_hwc, int c, int h, int w, const float *uv, int64_t n,
             float *out, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!feat_hwc || (n > 0 && (!uv || !out)) || n < 0 || h <= 0 || w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_index: bad argument");
  if (!aligned16(feat_hwc)) return fail(ctx, MP_ERR_ARG, "mp_index: feat_hwc must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_index(ctx, feat_hwc, c, h, w, uv, n, out, (hipStream_t)stream);
}

int mp_orthogonal(mp_ctx *ctx, const float *points, int64_t n, const float *calib, float *out,
                  mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n < 0 || !calib || (n > 0 && (!points || !out)))
    return fail(ctx, MP_ERR_ARG, "mp_orthogonal: bad argument");
  DeviceGuard g(ctx->device);
  return launch_orthogonal(ctx, points, n, calib, out, (hipStream_t)stream);
}

int mp_query(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w, const float *points,
             int64_t n, int64_t stride_n, int64_t stride_c, const float *calib, float z_scale,
             float *out, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  int rc = check_ready(ctx, m, c);
  if (rc != MP_OK) return rc;
  if (!feat_hwc || !calib || n < 0 || h <= 0 || w <= 0 || (n > 0 && (!points || !out)))
    return fail(ctx, MP_ERR_ARG, "mp_query: bad argument");
  if (!aligned16(feat_hwc)) return fail(ctx, MP_ERR_ARG, "mp_query: feat_hwc must be 16-byte aligned");
  if (n == 0) return MP_OK;
  PointSrc src;
  std::memset(&src, 0, sizeof(src));
  src.pts = points;
  src.sn = stride_n;
  src.sc = stride_c;
  src.n = n;
  src.out_stride = n;
  DeviceGuard g(ctx->device);
  return launch_query(ctx, *m, feat_hwc, h, w, calib, z_scale, src, out, n, (hipStream_t)stream);
}

int mp_mlp_forward(mp_ctx *ctx, int mlp, const float *feature, int64_t n, float *out,
                   mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  if (!m) return fail(ctx, MP_ERR_ARG, "mp_mlp_forward: unknown mlp id %d", mlp);
  int rc = check_ready(ctx, m, m->c);
  if (rc != MP_OK) return rc;
  if (n < 0 || (n > 0 && (!feature || !out))) return fail(ctx, MP_ERR_ARG, "mp_mlp_forward: bad argument");
  if (n == 0) return MP_OK;
  PointSrc src;
  std::memset(&src, 0, sizeof(src));
  src.pts = feature;
  src.sn = 1;
  src.sc = n;
  src.n = n;
  src.out_stride = n;
  DeviceGuard g(ctx->device);
  return launch_query(ctx, *m, /*feat_hwc=*/nullptr, 0, 0, /*calib=*/nullptr, 0.0f, src, out, n,
                      (hipStream_t)stream);
}

int mp_query_counted(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w,
                     const float *points, int64_t capacity, const int32_t *count,
                     const float *calib, float z_scale, float *out, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  int rc = check_ready(ctx, m, c);
  if (rc != MP_OK) return rc;
  if (!feat_hwc || !calib || !count || capacity < 0 || h <= 0 || w <= 0 ||
      (capacity > 0 && (!points || !out)))
    return fail(ctx, MP_ERR_ARG, "mp_query_counted: bad argument");
  if (!aligned16(feat_hwc))
    return fail(ctx, MP_ERR_ARG, "mp_query_counted: feat_hwc must be 16-byte aligned");
  if (capacity == 0) return MP_OK;
  PointSrc src;
  std::memset(&src, 0, sizeof(src));
  src.pts = points;
  src.sn = 1;
  src.sc = capacity;
  src.n_dev = count;
  src.out_stride = capacity;
  DeviceGuard g(ctx->device);
  return launch_query(ctx, *m, feat_hwc, h, w, calib, z_scale, src, out, capacity,
                      (hipStream_t)stream);
}

int mp_query_counted_batch(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c,
                           int h, int w, const float *const *points, int64_t capacity,
                           const int32_t *const *count, const float *const *calib, float z_scale,
                           float *const *out, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  int rc = check_ready(ctx, m, c);
  if (rc != MP_OK) return rc;
  if (n_frames < 1 || n_frames > kMaxFrames)
    return fail(ctx, MP_ERR_ARG, "mp_query_counted_batch: 1..%d frames per call, got %d", kMaxFrames,
                n_frames);
  if (!feat_hwc || !points || !count || !calib || !out || capacity < 0 || h <= 0 || w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_query_counted_batch: bad argument");
  if (capacity == 0) return MP_OK;
  QuerySet set;
  std::memset(&set, 0, sizeof(set));
  set.n = n_frames;
  for (int f = 0; f < n_frames; ++f) {
    if (!feat_hwc[f] || !points[f] || !count[f] || !calib[f] || !out[f])
      return fail(ctx, MP_ERR_ARG, "mp_query_counted_batch: null buffer for frame %d", f);
    if (!aligned16(feat_hwc[f]))
      return fail(ctx, MP_ERR_ARG, "mp_query_counted_batch: feat_hwc must be 16-byte aligned");
    QueryItem &q = set.it[f];
    q.feat = feat_hwc[f];
    q.calib = calib[f];
    q.out = out[f];
    q.src.pts = points[f];
    q.src.sn = 1;
    q.src.sc = capacity;
    q.src.n_dev = count[f];
    q.src.out_stride = capacity;
  }
  DeviceGuard g(ctx->device);
  return launch_query_set(ctx, *m, set, h, w, z_scale, capacity * n_frames, true, (hipStream_t)stream);
}

static int check_resolutions(mp_ctx *ctx, const char *who, const int *resolutions, int n_levels) {
  for (int l = 0; l < n_levels; ++l) {
    if (resolutions[l] < 2 || resolutions[l] > 1023)
      return fail(ctx, MP_ERR_UNSUPPORTED, "%s: resolution %d outside [2,1023]", who, resolutions[l]);
    if (l > 0 && resolutions[l] != 2 * resolutions[l - 1] - 1)
      return fail(ctx, MP_ERR_UNSUPPORTED, "%s: resolutions must follow r -> 2r-1 (got %d after %d)",
                  who, resolutions[l], resolutions[l - 1]);
  }
  return MP_OK;
}

int mp_recon_batch(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c, int h,
                   int w, const float *const *calib, float z_scale, const float *b_min,
                   const float *b_max, const int *resolutions, int n_levels, float balance,
                   float *const *volume, int32_t *const *status, mp_stream stream) {
  return mp_recon_batch_ex(ctx, mlp, n_frames, feat_hwc, c, h, w, calib, z_scale, b_min, b_max, resolutions,
                           n_levels, balance, MP_FINAL_DILATE3, volume, status, stream);
}

int mp_recon_batch_ex(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c, int h,
                      int w, const float *const *calib, float z_scale, const float *b_min,
                      const float *b_max, const int *resolutions, int n_levels, float balance,
                      int final_level, float *const *volume, int32_t *const *status, mp_stream stream) {
  return mp_recon_batch_early(ctx, mlp, n_frames, feat_hwc, c, h, w, calib, z_scale, b_min, b_max, resolutions,
                              n_levels, balance, final_level, volume, status, nullptr, stream);
}

int mp_recon_batch_early(mp_ctx *ctx, int mlp, int n_frames, const float *const *feat_hwc, int c, int h,
                         int w, const float *const *calib, float z_scale, const float *b_min,
                         const float *b_max, const int *resolutions, int n_levels, float balance,
                         int final_level, float *const *volume, int32_t *const *status,
                         const mp_recon_early *early, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const Mlp *m = get_mlp(ctx, mlp);
  int rc = check_ready(ctx, m, c);
  if (rc != MP_OK) return rc;
  if (n_frames < 1 || n_frames > kMaxFrames)
    return fail(ctx, MP_ERR_ARG, "mp_recon_batch: 1..%d frames per call, got %d", kMaxFrames, n_frames);
  if (!feat_hwc || !calib || !b_min || !b_max || !resolutions || !volume || !status ||
      n_levels < 1 || n_levels > 8 || h <= 0 || w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_recon: bad argument");
  for (int f = 0; f < n_frames; ++f) {
    if (!feat_hwc[f] || !calib[f] || !volume[f] || !status[f])
      return fail(ctx, MP_ERR_ARG, "mp_recon: null buffer for frame %d", f);
    if (!aligned16(feat_hwc[f]))
      return fail(ctx, MP_ERR_ARG, "mp_recon: feat_hwc must be 16-byte aligned");
  }
  if (m->cout != 1) return fail(ctx, MP_ERR_ARG, "mp_recon: needs a 1-channel (occupancy) mlp");
  if (final_level != MP_FINAL_DILATE3 && final_level != MP_FINAL_UPSTREAM && final_level != MP_FINAL_INTERPOLATE)
    return fail(ctx, MP_ERR_ARG, "mp_recon: final_level must be MP_FINAL_DILATE3 / _UPSTREAM / _INTERPOLATE, got %d",
                final_level);
  rc = check_resolutions(ctx, "mp_recon", resolutions, n_levels);
  if (rc != MP_OK) return rc;
  if (early && (!early->flags_dev || !early->flags_host))
    return fail(ctx, MP_ERR_ARG, "mp_recon_batch_early: flags_dev and flags_host are required");
  DeviceGuard g(ctx->device);
  void *scratch = nullptr;
  rc = ensure_scratch(ctx, (hipStream_t)stream, n_frames * recon_scratch_bytes(resolutions, n_levels),
                      &scratch);
  if (rc != MP_OK) return rc;
  return launch_recon(ctx, scratch, *m, n_frames, feat_hwc, h, w, calib, z_scale, b_min, b_max,
                      resolutions, n_levels, balance, final_level, volume, status, early, (hipStream_t)stream);
}

int mp_recon(mp_ctx *ctx, int mlp, const float *feat_hwc, int c, int h, int w, const float *calib,
             float z_scale, const float *b_min, const float *b_max, const int *resolutions,
             int n_levels, float balance, float *volume, int32_t *status, mp_stream stream) {
  return mp_recon_batch(ctx, mlp, 1, &feat_hwc, c, h, w, &calib, z_scale, b_min, b_max, resolutions,
                        n_levels, balance, &volume, &status, stream);
}

static int octree_select_impl(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                              const uint64_t *ev_prev, uint64_t *ev_cur, uint64_t *bnd, int level,
                              int box, float balance, uint32_t *packed, int32_t *count,
                              mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ev_cur || !packed || !count || r < 2 || r > 1023 || level < 0)
    return fail(ctx, MP_ERR_ARG, "mp_octree_select: bad argument");
  if (prev && (!cur || !ev_prev || !bnd || r != 2 * rp - 1 || level < 1))
    return fail(ctx, MP_ERR_ARG, "mp_octree_select: refinement needs cur, ev_prev, bnd and r == 2 rp - 1");
  if (box != 0 && box != 1 && box != 3 && box != 7 && box != 9)
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_octree_select_box: dilation box must be 0, 1, 3, 7 or 9, got %d", box);
  DeviceGuard g(ctx->device);
  return launch_octree_select(ctx, prev, rp, cur, r,
                              reinterpret_cast<const unsigned long long *>(ev_prev),
                              reinterpret_cast<unsigned long long *>(ev_cur),
                              reinterpret_cast<unsigned long long *>(bnd), box, balance, packed,
                              count, (hipStream_t)stream);
}

int mp_octree_select(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                     const uint64_t *ev_prev, uint64_t *ev_cur, uint64_t *bnd, int level,
                     float balance, uint32_t *packed, int32_t *count, mp_stream stream) {
  return octree_select_impl(ctx, prev, rp, cur, r, ev_prev, ev_cur, bnd, level,
                            octree_box_of_level(level), balance, packed, count, stream);
}

int mp_octree_select_box(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                         const uint64_t *ev_prev, uint64_t *ev_cur, uint64_t *bnd, int box,
                         float balance, uint32_t *packed, int32_t *count, mp_stream stream) {
  return octree_select_impl(ctx, prev, rp, cur, r, ev_prev, ev_cur, bnd, prev ? 1 : 0, box, balance,
                            packed, count, stream);
}

int mp_octree_conflicts(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, int64_t capacity,
                        int r, const float *values, const float *volume, float balance,
                        uint64_t *ev, uint32_t *out_packed, int32_t *out_count, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!packed || !count || capacity < 0 || r < 2 || r > 1023 || !volume || !ev || !out_packed ||
      !out_count || (capacity > 0 && !values))
    return fail(ctx, MP_ERR_ARG, "mp_octree_conflicts: bad argument");
  DeviceGuard g(ctx->device);
  return launch_octree_conflicts(ctx, packed, count, capacity, r, values, volume, balance,
                                 reinterpret_cast<unsigned long long *>(ev), out_packed, out_count,
                                 (hipStream_t)stream);
}

int mp_lattice_points(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, int64_t capacity,
                      int stride, int res_final, const float *b_min, const float *b_max,
                      float *points, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!packed || !count || capacity < 0 || stride < 1 || res_final < 2 || !b_min || !b_max ||
      (capacity > 0 && !points))
    return fail(ctx, MP_ERR_ARG, "mp_lattice_points: bad argument");
  DeviceGuard g(ctx->device);
  return launch_lattice_points(ctx, packed, count, capacity, stride, res_final, b_min, b_max,
                               points, (hipStream_t)stream);
}

int mp_scatter_nodes(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, int64_t capacity,
                     int r, const float *values, float *volume, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!packed || !count || capacity < 0 || r < 2 || !volume || (capacity > 0 && !values))
    return fail(ctx, MP_ERR_ARG, "mp_scatter_nodes: bad argument");
  DeviceGuard g(ctx->device);
  return launch_scatter_nodes(ctx, packed, count, capacity, r, values, volume, (hipStream_t)stream);
}

int mp_forward_vertices(mp_ctx *ctx, const float *volume, int r, int direction, int64_t *x,
                        int64_t *y, float *z, float *norm, int32_t *count, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!volume || !x || !y || !z || !norm || !count || r < 1 || r > 4096 ||
      direction < MP_DIR_FRONT || direction > MP_DIR_RIGHT)
    return fail(ctx, MP_ERR_ARG, "mp_forward_vertices: bad argument");
  DeviceGuard g(ctx->device);
  void *scratch = nullptr;
  int rc = ensure_scratch(ctx, (hipStream_t)stream, forward_vertices_scratch_bytes(r) + 4096, &scratch);
  if (rc != MP_OK) return rc;
  return launch_forward_vertices(ctx, scratch, volume, r, direction, x, y, z, norm, count,
                                 (hipStream_t)stream);
}

int mp_forward_vertices_batch(mp_ctx *ctx, int n_frames, const float *const *volume, int r, int direction,
                              int64_t *const *x, int64_t *const *y, float *const *z, float *const *norm,
                              int32_t *const *count, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n_frames < 1 || n_frames > kMaxFrames)
    return fail(ctx, MP_ERR_ARG, "mp_forward_vertices_batch: 1..%d frames per call, got %d", kMaxFrames, n_frames);
  if (!volume || !x || !y || !z || !norm || !count || r < 1 || r > 4096 || direction < MP_DIR_FRONT ||
      direction > MP_DIR_RIGHT)
    return fail(ctx, MP_ERR_ARG, "mp_forward_vertices_batch: bad argument");
  for (int f = 0; f < n_frames; ++f)
    if (!volume[f] || !x[f] || !y[f] || !z[f] || !norm[f] || !count[f])
      return fail(ctx, MP_ERR_ARG, "mp_forward_vertices_batch: null buffer for frame %d", f);
  DeviceGuard g(ctx->device);
  void *scratch = nullptr;
  int rc = ensure_scratch(ctx, (hipStream_t)stream, (size_t)n_frames * forward_vertices_scratch_bytes(r) + 4096, &scratch);
  if (rc != MP_OK) return rc;
  return launch_forward_vertices_batch(ctx, scratch, n_frames, volume, r, direction, x, y, z, norm, count,
                                       (hipStream_t)stream);
}

int mp_vertex_points(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *z,
                     const int32_t *count, int64_t capacity, int res, const float *mat,
                     float *points, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !y || !z || !count || !mat || !points || capacity < 0 || res < 1)
    return fail(ctx, MP_ERR_ARG, "mp_vertex_points: bad argument");
  DeviceGuard g(ctx->device);
  return launch_vertex_points(ctx, x, y, z, count, capacity, res, mat, points, (hipStream_t)stream);
}

int mp_paint(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *values,
             int channel_major, const int32_t *count, int64_t capacity, int res, float scale,
             float bias, float lo, float hi, float *image, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !y || !values || !count || !image || capacity < 0 || res < 1)
    return fail(ctx, MP_ERR_ARG, "mp_paint: bad argument");
  DeviceGuard g(ctx->device);
  return launch_paint(ctx, x, y, values, channel_major, count, capacity, res, scale, bias, lo, hi,
                      image, (hipStream_t)stream);
}

int mp_paint_batch(mp_ctx *ctx, int n_frames, const int64_t *const *x, const int64_t *const *y,
                   const float *const *values, int channel_major, const int32_t *const *count, int64_t capacity,
                   int res, float scale, float bias, float lo, float hi, float *const *image, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n_frames < 1 || n_frames > kMaxFrames)
    return fail(ctx, MP_ERR_ARG, "mp_paint_batch: 1..%d frames per call, got %d", kMaxFrames, n_frames);
  if (!x || !y || !values || !count || !image || capacity < 0 || res < 1)
    return fail(ctx, MP_ERR_ARG, "mp_paint_batch: bad argument");
  for (int f = 0; f < n_frames; ++f)
    if (!x[f] || !y[f] || !values[f] || !count[f] || !image[f])
      return fail(ctx, MP_ERR_ARG, "mp_paint_batch: null buffer for frame %d", f);
  DeviceGuard g(ctx->device);
  return launch_paint_batch(ctx, n_frames, x, y, values, channel_major, count, capacity, res, scale, bias, lo, hi,
                            image, (hipStream_t)stream);
}

int mp_visualize(mp_ctx *ctx, const float *image, int res, int size, float *out, uint8_t *mask,
                 mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!image || !out || !mask || res < 1 || size < 1 || size > 16384)
    return fail(ctx, MP_ERR_ARG, "mp_visualize: bad argument");
  DeviceGuard g(ctx->device);
  return launch_visualize(ctx, image, res, size, out, mask, (hipStream_t)stream);
}

int mp_prepare_inputs(mp_ctx *ctx, const float *segm, int64_t hw, const float *mean,
                      const float *std, float *input_g, float *input_c, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!segm || !mean || !std || !input_g || hw < 0)
    return fail(ctx, MP_ERR_ARG, "mp_prepare_inputs: bad argument");
  if (hw == 0) return MP_OK;
  DeviceGuard g(ctx->device);
  return launch_prepare_inputs(ctx, segm, hw, mean, std, input_g, input_c, (hipStream_t)stream);
}

int mp_marching_cubes(mp_ctx *ctx, const float *volume, int r, float level, const float *b_min,
                      const float *b_max, float *verts, int64_t max_verts, int32_t *faces,
                      int64_t max_faces, int32_t *counts, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!volume || !b_min || !b_max || !counts || r < 2 || r > 1023 || max_verts < 0 ||
      max_faces < 0 || (max_verts > 0 && !verts) || (max_faces > 0 && !faces))
    return fail(ctx, MP_ERR_ARG, "mp_marching_cubes: bad argument");
  DeviceGuard g(ctx->device);
  void *scratch = nullptr;
  int rc = ensure_scratch(ctx, (hipStream_t)stream, mc_scratch_bytes(r), &scratch);
  if (rc != MP_OK) return rc;
  return launch_marching_cubes(ctx, scratch, volume, r, level, b_min, b_max, verts, max_verts,
                               faces, max_faces, counts, (hipStream_t)stream);
}

int mp_group_norm(mp_ctx *ctx, const float *x, int n, int c, int64_t hw, int groups,
                  const float *gamma, const float *beta, float eps, int relu, float *y,
                  mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !y || !gamma || !beta || n <= 0 || n > 4096 || c <= 0 || hw <= 0 || groups <= 0 ||
      groups > 4096)
    return fail(ctx, MP_ERR_ARG, "mp_group_norm: bad argument");
  if (c % groups || hw % 4 || !aligned16(x) || !aligned16(y))
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_group_norm: needs C %% groups == 0, HW %% 4 == 0, 16-byte aligned x/y");
  DeviceGuard g(ctx->device);
  void *scratch = nullptr;
  int rc = ensure_scratch(ctx, (hipStream_t)stream, gn_scratch_bytes(n * groups), &scratch);
  if (rc != MP_OK) return rc;
  return launch_group_norm(ctx, scratch, x, n, c, hw, groups, gamma, beta, eps, relu, y,
                           (hipStream_t)stream);
}

int mp_upsample_bicubic2x(mp_ctx *ctx, const float *x, int c, int h, int w, const float *add,
                          float *y, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !y || c <= 0 || h < 2 || w < 2)
    return fail(ctx, MP_ERR_ARG, "mp_upsample_bicubic2x: bad argument");
  DeviceGuard g(ctx->device);
  return launch_upsample_bicubic2x(ctx, x, c, h, w, add, y, (hipStream_t)stream);
}

int mp_concat3_add(mp_ctx *ctx, const float *a, int ca, const float *b, int cb, const float *c,
                   int cc, const float *shortcut, int n, int64_t hw, float *y, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!a || !b || !c || !shortcut || !y || ca <= 0 || cb <= 0 || cc <= 0 || n <= 0 || hw <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_concat3_add: bad argument");
  if (hw % 4 || !aligned16(a) || !aligned16(b) || !aligned16(c) || !aligned16(shortcut) || !aligned16(y))
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_concat3_add: needs HW %% 4 == 0 and 16-byte aligned buffers");
  DeviceGuard g(ctx->device);
  return launch_concat3_add(ctx, a, ca, b, cb, c, cc, shortcut, n, hw, y, (hipStream_t)stream);
}

int mp_conv3x3_pack(mp_ctx *ctx, const float *w, int cout, int cin, float *packed, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!w || !packed || cout <= 0 || cin <= 0) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_pack: bad argument");
  if (cin % 16 || cout % 32)
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_conv3x3_pack: needs Cin %% 16 == 0 and Cout %% 32 == 0");
  DeviceGuard g(ctx->device);
  return launch_conv3x3_pack(ctx, w, cout, cin, packed, (hipStream_t)stream);
}

int mp_conv3x3_pack_wino(mp_ctx *ctx, const float *w, int cout, int cin, float *packed_wino, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!w || !packed_wino || cout <= 0 || cin <= 0) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_pack_wino: bad argument");
  if (cin % 16 || cout % 32)
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_conv3x3_pack_wino: needs Cin %% 16 == 0 and Cout %% 32 == 0");
  if (!aligned16(packed_wino)) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_pack_wino: the output must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_conv3x3_wino_pack(ctx, w, cout, cin, packed_wino, (hipStream_t)stream);
}

int mp_conv3x3_wino_supported(int cin, int cout, int h, int w) { return conv3x3_wino_supported(cin, cout, h, w) ? 1 : 0; }

int mp_conv3x3_pack16(mp_ctx *ctx, const float *w, int cout, int cin, void *packed16, float *wmax,
                      mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!w || !packed16 || !wmax || cout <= 0 || cin <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_conv3x3_pack16: bad argument");
  if (cin % 16 || cout % 32)
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_conv3x3_pack16: needs Cin %% 16 == 0 and Cout %% 32 == 0");
  DeviceGuard g(ctx->device);
  return launch_conv3x3_pack16(ctx, w, cout, cin, packed16, wmax, (hipStream_t)stream);
}

int mp_conv3x3_gn16(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss, int relu,
                    int reflect, const void *packed16, const float *wmax, int cout, float *y, double *stats,
                    mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !packed16 || !wmax || !y || n <= 0 || n > 65535 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_conv3x3_gn16: bad argument");
  if (!aligned16(packed16)) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_gn16: packed weights must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_conv3x3_gn(ctx, x, n, cin, h, w, ss, relu, reflect, static_cast<const float *>(packed16),
                           wmax, cout, y, stats, (hipStream_t)stream);
}

int mp_conv1x1_pack(mp_ctx *ctx, const float *w1, int c1, const float *w2, int c2, int cout, int f16,
                    void *packed, float *wmax, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!w1 || !packed || c1 <= 0 || c2 < 0 || (c2 > 0 && !w2) || (f16 && !wmax))
    return fail(ctx, MP_ERR_ARG, "mp_conv1x1_pack: bad argument");
  if (c1 % 64 || c2 % 64 || (cout != 128 && cout != 256))
    return fail(ctx, MP_ERR_UNSUPPORTED,
                "mp_conv1x1_pack: input channel counts must be multiples of 64, output channels 128 or 256");
  DeviceGuard g(ctx->device);
  return launch_conv1x1_pack(ctx, w1, c1, w2, c2, cout, f16, packed, wmax, (hipStream_t)stream);
}

int mp_conv1x1(mp_ctx *ctx, const float *x1, const float *ss1, int relu1, const float *x2, int n, int c1,
               int c2, int cout, int64_t hw, const void *packed, int f16, const float *wmax, const float *bias,
               const float *res, float *y, float *y_hwc, double *stats, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x1 || !packed || n <= 0 || hw <= 0 || (!y && !y_hwc) || (f16 && !wmax))
    return fail(ctx, MP_ERR_ARG, "mp_conv1x1: bad argument");
  if (!aligned16(packed) || (y_hwc && !aligned16(y_hwc)))
    return fail(ctx, MP_ERR_ARG, "mp_conv1x1: packed weights / y_hwc must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_conv1x1_raw(ctx, x1, ss1, relu1, x2, n, c1, c2, cout, hw, packed, f16, wmax, bias, res, y,
                            y_hwc, stats, (hipStream_t)stream);
}

int mp_scale_shift_add(mp_ctx *ctx, const float *t, const float *ss, const float *res, int n, int c,
                       int64_t hw, float *y, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!t || !ss || !res || !y || n <= 0 || c <= 0 || hw <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_scale_shift_add: bad argument");
  if (hw % 4 || !aligned16(t) || !aligned16(res) || !aligned16(y))
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_scale_shift_add: needs HW %% 4 == 0 and 16-byte aligned buffers");
  DeviceGuard g(ctx->device);
  return launch_scale_shift_add(ctx, t, ss, res, (long long)n * c, hw, y, (hipStream_t)stream);
}

int mp_conv3x3_supported(int cin, int cout, int h, int w) { return conv3x3_supported(cin, cout, h, w) ? 1 : 0; }

int mp_gn_stat_slices(void) { return gn_stat_slices(); }

int mp_conv3x3_stat_slices(int cout, int n, int h, int w, int f16) {
  if (cout < 32 || cout % 32 || n < 1 || h < 1 || w < 32) return 0;
  return conv3x3_stat_slices(cout, n, h, w, f16 != 0);
}

void mp_query_tune(int small_tiles) { mp::query_small_set_gate(small_tiles); }

void mp_conv3x3_tune(int nr) {
  conv3x3_set_nr(nr & 0xfff);
  conv1x1_set_mrw((nr >> 12) & 3);
}

int mp_conv3x3_gn(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss, int relu,
                  int reflect, const float *packed, int cout, float *y, double *stats, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !packed || !y || n <= 0 || n > 65535 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_conv3x3_gn: bad argument");
  if (!aligned16(packed)) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_gn: packed weights must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_conv3x3_gn(ctx, x, n, cin, h, w, ss, relu, reflect, packed, nullptr, cout, y, stats,
                           (hipStream_t)stream);
}

int mp_gn_stats(mp_ctx *ctx, const float *x, int n, int c, int64_t hw, int groups, double *partial,
                mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!x || !partial || n <= 0 || c <= 0 || hw <= 0 || groups <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_gn_stats: bad argument");
  if (c % groups || hw % 4 || !aligned16(x))
    return fail(ctx, MP_ERR_UNSUPPORTED, "mp_gn_stats: needs C %% groups == 0, HW %% 4 == 0, 16-byte aligned x");
  DeviceGuard g(ctx->device);
  return launch_gn_stats(ctx, x, n, c, hw, groups, partial, (hipStream_t)stream);
}

int mp_gn_finalize(mp_ctx *ctx, const double *partial, int n, int c, int groups, int slices,
                   int64_t count, const float *gamma, const float *beta, float eps, float *ss,
                   mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!partial || !gamma || !beta || !ss || n <= 0 || c <= 0 || groups <= 0 || slices <= 0 ||
      count <= 0 || c % groups || c / groups > 64)
    return fail(ctx, MP_ERR_ARG, "mp_gn_finalize: bad argument");
  DeviceGuard g(ctx->device);
  return launch_gn_finalize(ctx, partial, n, c, groups, slices, (double)count, gamma, beta, eps, ss,
                            (hipStream_t)stream);
}

// mp_gn_out / mp_gn_in (C-ABI) -> GnOut / GnIn (kernel arguments); the launchers fill c / S / count
static GnOut to_out(const mp_gn_out *f) {
  GnOut g = gn_out_none();
  if (!f) return g;
  g.acc = reinterpret_cast<long long *>(f->acc);
  g.partial = f->partial;
  return g;
}
static long long out_cap(const mp_gn_out *f) { return f && f->partial ? (long long)f->partial_doubles : -1; }
static GnIn to_in(const mp_gn_in *f) {
  GnIn g = gn_in_none();
  if (!f) return g;
  g.acc = reinterpret_cast<const long long *>(f->acc);
  g.gamma = f->gamma;
  g.beta = f->beta;
  g.eps = f->eps;
  g.ss = f->acc ? nullptr : f->ss;
  return g;
}

int mp_conv3x3_ex(mp_ctx *ctx, const mp_conv3x3_args *q, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!q || !q->x || !q->packed || (!q->y && !q->y2) || q->n <= 0 || q->n > 65535 || q->cin <= 0 || q->cout <= 0 ||
      q->h <= 0 || q->w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_conv3x3_ex: bad argument");
  if (!aligned16(q->packed)) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_ex: packed weights must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  ConvArgs a;
  a.x = q->x;
  a.gn = to_in(&q->gn);
  a.wp = static_cast<const float *>(q->packed);
  a.y = q->y;
  a.y2 = q->y2;
  a.res = q->res;
  a.y2_c = q->y2 ? q->y2_channels : 32;
  a.y2_off = q->y2 ? q->y2_offset : 0;
  a.fin = to_out(&q->fin);
  a.fin2 = to_out(&q->fin2);
  a.n_img = q->n;
  a.cin = q->cin;
  a.cout = q->cout;
  a.h = q->h;
  a.w = q->w;
  a.relu = q->relu;
  a.reflect = q->reflect;
  if (q->packed_wino) {
    if (!aligned16(q->packed_wino)) return fail(ctx, MP_ERR_ARG, "mp_conv3x3_ex: packed_wino must be 16-byte aligned");
    a.wpw = q->packed_wino;
  }
  const long long cap[2] = {out_cap(&q->fin), out_cap(&q->fin2)};
  return launch_conv3x3(ctx, a, q->wmax, cap, (hipStream_t)stream);
}

int mp_conv1x1_ex(mp_ctx *ctx, const mp_conv1x1_args *q, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!q || !q->x1 || !q->packed || q->n <= 0 || q->hw <= 0 || (!q->y && !q->y_hwc) || (q->f16 && !q->wmax))
    return fail(ctx, MP_ERR_ARG, "mp_conv1x1_ex: bad argument");
  if (!aligned16(q->packed) || (q->y_hwc && !aligned16(q->y_hwc)))
    return fail(ctx, MP_ERR_ARG, "mp_conv1x1_ex: packed weights / y_hwc must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  Conv1Args a;
  a.x1 = q->x1;
  a.gn1 = to_in(&q->gn1);
  a.x2 = q->x2;
  a.wp = static_cast<const float *>(q->packed);
  a.bias = q->bias;
  a.res = q->res;
  a.y = q->y;
  a.y_hwc = q->y_hwc;
  a.fin = to_out(&q->fin);
  a.n_img = q->n;
  a.c1 = q->c1;
  a.c2 = q->c2;
  a.hw = (int)q->hw;
  a.relu1 = q->relu1;
  a.cout = q->cout;
  return launch_conv1x1(ctx, a, q->f16, q->wmax, out_cap(&q->fin), (hipStream_t)stream);
}

int mp_gn_acc_replicas(void) { return kGnReplicas; }

int mp_conv1x1_stat_slices(int64_t hw) { return hw > 0 ? conv1x1_stat_slices(hw) : 0; }

int mp_convk_supported(int cin, int cout, int ks, int stride, int h, int w) {
  return convk_supported(cin, cout, ks, stride, h, w) ? 1 : 0;
}

int64_t mp_convk_packed_floats(int cin, int cout, int ks) {
  if (cin <= 0 || cout <= 0 || (ks != 3 && ks != 7) || cin % (ks == 7 ? 3 : 16)) return 0;
  return convk_packed_floats(cin, cout, ks);
}

int mp_convk_stat_slices(int ks, int stride, int h, int w) {
  if ((ks != 3 && ks != 7) || stride < 1 || stride > 2 || h <= 0 || w <= 0) return 0;
  return convk_stat_slices(ks, stride, h, w);
}

int mp_convk_pack(mp_ctx *ctx, const float *w, int cout, int cin, int ks, float *packed, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!w || !packed || mp_convk_packed_floats(cin, cout, ks) <= 0 || cout % 32)
    return fail(ctx, MP_ERR_ARG, "mp_convk_pack: bad argument");
  DeviceGuard g(ctx->device);
  return launch_convk_pack(ctx, w, cout, cin, ks, packed, (hipStream_t)stream);
}

int mp_convk(mp_ctx *ctx, const mp_convk_args *q, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!q || !q->x || !q->packed || !q->y || q->n <= 0 || q->n > 65535 || q->cin <= 0 || q->cout <= 0 ||
      q->h <= 0 || q->w <= 0)
    return fail(ctx, MP_ERR_ARG, "mp_convk: bad argument");
  if (!aligned16(q->packed)) return fail(ctx, MP_ERR_ARG, "mp_convk: packed weights must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  ConvKArgs a;
  a.x = q->x;
  a.gn = to_in(&q->gn);
  a.wp = q->packed;
  a.bias = q->bias;
  a.y = q->y;
  a.fin = to_out(&q->fin);
  a.n_img = q->n;
  a.cin = q->cin;
  a.cout = q->cout;
  a.h = q->h;
  a.w = q->w;
  a.ho = a.wo = 0;
  a.ks = q->ks;
  a.stride = q->stride;
  a.pad = q->ks / 2;
  a.reflect = q->reflect;
  a.relu = q->relu;
  a.wp_floats = 0;
  return launch_convk(ctx, a, out_cap(&q->fin), (hipStream_t)stream);
}

static int check_ew(mp_ctx *ctx, const char *who, const void *x, const void *y, int n, int c, long long hw_out) {
  if (!x || !y || n <= 0 || n > 4096 || c <= 0 || hw_out <= 0) return fail(ctx, MP_ERR_ARG, "%s: bad argument", who);
  if (c % 32 || hw_out % 4 || !aligned16(x) || !aligned16(y))
    return fail(ctx, MP_ERR_UNSUPPORTED, "%s: needs C %% 32 == 0, output H*W %% 4 == 0, 16-byte aligned buffers", who);
  return MP_OK;
}

int mp_avgpool2_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, float *y, const mp_gn_out *fin,
                   mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (h < 2 || w < 8 || h % 2 || w % 8) return fail(ctx, MP_ERR_UNSUPPORTED, "mp_avgpool2_gn: needs H %% 2 == 0, W %% 8 == 0");
  int rc = check_ew(ctx, "mp_avgpool2_gn", x, y, n, c, (long long)(h / 2) * (w / 2));
  if (rc != MP_OK) return rc;
  DeviceGuard g(ctx->device);
  return launch_avgpool2_gn(ctx, x, n, c, h, w, y, to_out(fin), out_cap(fin), (hipStream_t)stream);
}

int mp_upsample_bicubic2x_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, const float *add, float *y,
                             const mp_gn_out *fin, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (h < 2 || w < 2 || w % 2) return fail(ctx, MP_ERR_UNSUPPORTED, "mp_upsample_bicubic2x_gn: needs H, W >= 2, W %% 2 == 0");
  int rc = check_ew(ctx, "mp_upsample_bicubic2x_gn", x, y, n, c, 4LL * h * w);
  if (rc != MP_OK) return rc;
  if (add && !aligned16(add)) return fail(ctx, MP_ERR_UNSUPPORTED, "mp_upsample_bicubic2x_gn: add must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_upsample_add_gn(ctx, x, n, c, h, w, add, y, to_out(fin), out_cap(fin), (hipStream_t)stream);
}

int mp_gn_apply(mp_ctx *ctx, const float *x, const mp_gn_in *gn, int relu, int n, int c, int64_t hw,
                const float *res, float *y, const mp_gn_out *fin, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!gn) return fail(ctx, MP_ERR_ARG, "mp_gn_apply: bad argument");
  int rc = check_ew(ctx, "mp_gn_apply", x, y, n, c, hw);
  if (rc != MP_OK) return rc;
  if (res && !aligned16(res)) return fail(ctx, MP_ERR_UNSUPPORTED, "mp_gn_apply: res must be 16-byte aligned");
  DeviceGuard g(ctx->device);
  return launch_gn_apply_gn(ctx, x, to_in(gn), relu, n, c, hw, res, y, to_out(fin), out_cap(fin), (hipStream_t)stream);
}

int mp_mfma_clock_probe(mp_ctx *ctx, float ms_target, double *out4, mp_stream stream) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!out4 || !(ms_target >= 1.0f) || ms_target > 2000.0f)
    return fail(ctx, MP_ERR_ARG, "mp_mfma_clock_probe: out4 must be given, 1 <= ms_target <= 2000");
  DeviceGuard g(ctx->device);
  return launch_mfma_clock_probe(ctx, ms_target, out4, (hipStream_t)stream);
}

int mp_profile_begin(mp_ctx *ctx, int max_records) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (max_records < 1 || max_records > (1 << 20))
    return fail(ctx, MP_ERR_ARG, "mp_profile_begin: bad max_records");
  DeviceGuard g(ctx->device);
  for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
  ctx->prof_events.clear();
  ctx->prof_used = 0;
  ctx->prof_events.resize(2 * (size_t)max_records);
  for (auto &e : ctx->prof_events) MP_HIP(ctx, hipEventCreate(&e));
  return MP_OK;
}

int mp_profile_end(mp_ctx *ctx, float *ms_out, int capacity, int *n_out) {
  if (!ctx) return MP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ms_out || !n_out || capacity < 0) return fail(ctx, MP_ERR_ARG, "mp_profile_end: bad argument");
  DeviceGuard g(ctx->device);
  const int n = ctx->prof_used;
  int rc = MP_OK;
  for (int i = 0; i < n && rc == MP_OK; ++i) {
    if (hipEventSynchronize(ctx->prof_events[2 * i + 1]) != hipSuccess) {
      rc = fail(ctx, MP_ERR_HIP, "mp_profile_end: hipEventSynchronize failed");
      break;
    }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]) != hipSuccess)
      rc = fail(ctx, MP_ERR_HIP, "mp_profile_end: hipEventElapsedTime failed");
    if (i < capacity) ms_out[i] = ms;
  }
  *n_out = n;
  for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
  ctx->prof_events.clear();
  ctx->prof_used = 0;
  return rc;
}

}  // extern "C"
