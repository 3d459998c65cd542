// Fused per-point PIFu query for gfx950 (MI355X):
//   project -> in-image mask -> bilinear feature gather -> z concat -> skip-connected MLP -> mask
// i.e. MonoPortNet.query in eval mode (monoport/lib/modeling/MonoPortNet.py:48-91) with
// orthogonal (geometry.py:19-34), index (geometry.py:4-16), DepthNormalizer (normalizers/
// DepthNormalizer.py:32) and SurfaceClassifier.forward (heads/SurfaceClassifier.py:39-71) in ONE
// kernel.  Nothing per-point ever goes to HBM except 12 B of coordinates in and 4*Cout B out.
//
// Decomposition (one workgroup = 4 waves = one 64-point tile, 2 workgroups per CU for netG):
//   * gather: each wave samples 16 points; a tap is one coalesced 16 B/lane read of a
//     channels-last feature row; the blended feature vector lands in LDS, point-major, XOR
//     swizzled in 16-byte slots so that both the gather's ds_write_b128 and the MFMA B-operand
//     ds_read_b128 are bank-conflict free (MI355X_MICROARCH.md, LDS table).
//   * MLP on v_mfma_f32_32x32x2_f32 (exact f32, 157 TF/s peak): M = output channels,
//     N = points, K = [hidden | feature | z].  Weights stream straight from L2 in pre-packed
//     fragment order (pack.hip) -- each wave owns distinct output rows, so there is no reuse to
//     stage through LDS; activations are the shared operand and live in LDS / registers.
//   * layer 0 (1024 x 257) is never materialised: it is produced in 64-row chunks that go
//     through a 16 KB LDS buffer straight into layer 1's K loop, whose 512 x 64 accumulator tile
//     is spread over the 4 waves' registers (128 VGPRs each).  Layers 2 and 3 consume their
//     inputs the same way, chunk by chunk from the owning wave's registers.  The skip-concat
//     (SurfaceClassifier.py:55) is just a second K segment read from the feature tile.
//   * last layer (Cout x 385) and the activation run on the VALU.
#include <cstring>

#include "mp_internal.h"
#include "query_common.h"

#include "query_mfma.h"

#pragma clang fp contract(off)

namespace mp {

// ---- the fused kernel ----------------------------------------------------------------------------
// DIRECT = true: SurfaceClassifier.forward on explicit features (SurfaceClassifier.py:39-71): the
// "points" are columns of a [C+1, N] tensor (src.pts, row stride src.sc) that already holds the
// sampled features and z_feat; no projection, no sampling, no mask.
template <int C, int COUT, int WPS, bool DIRECT>
__global__ __launch_bounds__(kQueryThreads, WPS) void pifu_query_kernel(
    MlpPack mlp, int fh, int fw, float z_scale, int act, QuerySetDev set, int gate_tiles) {
  constexpr int ROWB = C * 4;
  constexpr int NGX = C / 8;  // K groups of the feature segment
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *xs = smem;                    // [64 points][C] f32, swizzled 16-byte slots
  unsigned char *hb = smem + kTilePts * ROWB;  // [64 points][64 rows] f32, swizzled

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  // The tiles of all frames of the set form one index space: frame f owns the next
  // ceil(n_f / tile) global tiles.  The owner of a global tile is looked up from the (device-side)
  // counts at the top of every iteration -- eight scalar loads -- instead of keeping a prefix
  // table alive in SGPRs across the whole MLP.
  const int swz = h ^ (j & 15);  // this lane's 16-byte-slot swizzle (see gemm_seg)
  const WStream ws = make_wstream(mlp.base, mlp.n_floats, lane);

  for (long long gtile = blockIdx.x;; gtile += gridDim.x) {
    int fi = -1;
    long long tile0 = 0;
    {
      long long acc = 0;
      // groups of 8 frames: the 8 count loads of a group are in flight together, and the dynamic group offset
      // keeps the compiler from hoisting all kMaxFrames kernel-argument loads into SGPRs (spills)
      for (int f0 = 0; f0 < set.n; f0 += 8)
#pragma unroll
      for (int fk = 0; fk < 8; ++fk) {
        const int f = f0 + fk;
        if (f < set.n) {
          const long long nf = set.count(f);
          const long long t = (nf + kTilePts - 1) / kTilePts;
          if (fi < 0 && gtile < acc + t) {
            fi = f;
            tile0 = acc;
          }
          acc += t;
        }
      }
      // launches of fewer than gate_tiles tiles belong to the 32-point kernel (query_small.hip),
      // which was launched next to this one because the counts live on the device
      if (acc < gate_tiles) break;
    }
    if (fi < 0) break;  // past the last tile of the last frame
    const QueryItem item = set.item(fi);
    const float *__restrict__ feat = item.feat;
    const float *__restrict__ calib = item.calib;
    float *__restrict__ out = item.out;
    const PointSrc &src = item.src;
    const long long n_pts = src.n_dev ? (long long)*src.n_dev : src.n;
    const long long n0 = (gtile - tile0) * kTilePts;

    // ---------------- gather: 16 points per wave ----------------
    float zb[2];  // z_feat B operands of this wave's two column blocks
    if constexpr (DIRECT) {
      // lane = point; each wave moves C/16 four-channel slots: 4 coalesced 256-byte row loads,
      // one conflict-free ds_write_b128
      const long long n = n0 + lane;
      const bool live = n < n_pts;
      for (int s0 = wv; s0 < C / 4; s0 += 4) {
        f32x4 r = {0.0f, 0.0f, 0.0f, 0.0f};
        if (live) {
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = src.pts[(long long)(4 * s0 + k) * src.sc + n];
        }
        *reinterpret_cast<f32x4 *>(xs + lane * ROWB + ((s0 ^ (lane & 15)) << 4)) = r;
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const long long m = n0 + 32 * cb + j;
        zb[cb] = (h == 0 && m < n_pts) ? src.pts[(long long)C * src.sc + m] : 0.0f;
      }
    } else {
    // ---------------- gather: 16 points per wave ----------------
    {
    float cal[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) cal[i] = calib[i];
    // 4 points per batch: 16 (C=256) / 32 (C=512) independent 16-byte loads in flight per lane.
    // Dead points (past the end / out of image) read a clamped in-bounds tap with weight 0, so
    // the loads need no branch and the compiler can issue the whole batch back to back.
    constexpr int GB = 4;
#pragma unroll 1
    for (int i0 = 0; i0 < 16; i0 += GB) {
      Taps t[GB];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const long long n = n0 + 16 * wv + i0 + u;
        const bool live_n = n < n_pts;
        float px = 0, py = 0, pz = 0, x, y, z;
        uint32_t code;
        if (live_n) load_point(src, n, px, py, pz, code);
        project(cal, px, py, pz, x, y, z);
        t[u] = make_taps(x, y, fh, fw, C, live_n && in_image(x, y));
      }
      f32x4 v[GB][C / 256][4];
#pragma unroll
      for (int u = 0; u < GB; ++u)
#pragma unroll
        for (int part = 0; part < C / 256; ++part)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[u][part][k] =
                *reinterpret_cast<const f32x4 *>(feat + t[u].o[k] + 4 * (lane + 64 * part));
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int p = 16 * wv + i0 + u;
#pragma unroll
        for (int part = 0; part < C / 256; ++part) {
          const int slot = lane + 64 * part;
          const f32x4 r = blend(v[u][part][0], v[u][part][1], v[u][part][2], v[u][part][3], t[u]);
          *reinterpret_cast<f32x4 *>(xs + p * ROWB + ((slot ^ (p & 15)) << 4)) = r;
        }
      }
    }

    // z_feat for the z k-step: lanes 0-31 carry it, lanes 32-63 supply 0
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const long long n = n0 + 32 * cb + j;
      float px = 0, py = 0, pz = 0, x, y, z;
      uint32_t code;
      if (n < n_pts) load_point(src, n, px, py, pz, code);
      project(cal, px, py, pz, x, y, z);
      zb[cb] = (h == 0 && n < n_pts) ? __fmul_rn(z, z_scale) : 0.0f;
    }
    }
    }
    __syncthreads();

    const unsigned char *xrow = xs + j * ROWB;       // this lane's point row, column block 0
    const unsigned char *hrow = hb + j * kHbRowBytes;

    // ---------------- layers 0 + 1, fused over 64-row chunks of layer 0 ----------------
    f32x16 acc1[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      init_from_bias(acc1[m][0], ws, mlp.bias[1] + 32 * (4 * wv + m));
      acc1[m][1] = acc1[m][0];
    }
    {
      const int rb0 = wv >> 1, cb0 = wv & 1;  // this wave's tile inside a layer-0 chunk
      const int a0 = mlp.ax[0] / 4;           // 16-byte units (segments are 256-byte aligned)
      const int a1 = mlp.ah[1] / 4 + (4 * wv) * (kHidden[0] / 8) * 64;
      const float zz[1] = {zb[cb0]};
      f32x4 ring0[kPrefetch0 + 1][1];
      f32x16 acc0[1][1];
      float az0[1];
      seg_prefetch<1, kPrefetch0, (kAHot & 1) != 0>(ring0, ws, a0 + rb0 * NGX * 64, 0, NGX);
      init_from_bias(acc0[0][0], ws, mlp.bias[0] + 32 * rb0);
      az0[0] = wload32(ws, mlp.az[0] + rb0 * 64);
#pragma unroll 1
      for (int ck = 0; ck < kHidden[0] / 64; ++ck) {
        // layer-0 rows [64 ck + 32 rb0, +32) x points [32 cb0, +32)
        const int rb = 2 * ck + rb0;
        seg_main<1, 1, kPrefetch0, ROWB, (kAHot & 1) != 0>(acc0, ring0, ws, a0 + rb * NGX * 64, 0, NGX,
                                xrow + cb0 * 32 * ROWB, swz);
        // layer-1 weights of this chunk start streaming before the chunk is even stored
        f32x4 ring1[kPrefetch1 + 1][4];
        seg_prefetch<4, kPrefetch1, (kAHot & 2) != 0>(ring1, ws, a1 + ck * 8 * 64, (kHidden[0] / 8) * 64, 8);
        gemm_z<1, 1>(acc0, az0, zz);
        lrelu(acc0[0][0]);
        store_hidden(hb, acc0[0][0], rb0, cb0, j, h);
        // next chunk's layer-0 operands
        const int rbn = min(rb + 2, kHidden[0] / 32 - 2 + rb0);
        seg_prefetch<1, kPrefetch0, (kAHot & 1) != 0>(ring0, ws, a0 + rbn * NGX * 64, 0, NGX);
        init_from_bias(acc0[0][0], ws, mlp.bias[0] + 32 * rbn);
        az0[0] = wload32(ws, mlp.az[0] + rbn * 64);
        MP_CHUNK_SYNC();
        // layer-1 rows [128 wv, +128) += W1[:, 64 ck .. +64) * chunk
        seg_main<4, 2, kPrefetch1, kHbRowBytes, (kAHot & 2) != 0>(acc1, ring1, ws, a1 + ck * 8 * 64, (kHidden[0] / 8) * 64, 8,
                                       hrow, swz);
        MP_CHUNK_SYNC();
      }
      // skip segment of layer 1: W1[:, 1024 .. 1024 + C] * x, then the z column
      const int a1x = mlp.ax[1] / 4 + (4 * wv) * NGX * 64;
      f32x4 ring1[kPrefetch1 + 1][4];
      float az1[4];
      seg_prefetch<4, kPrefetch1>(ring1, ws, a1x, NGX * 64, NGX);
#pragma unroll
      for (int m = 0; m < 4; ++m) az1[m] = wload32(ws, mlp.az[1] + (4 * wv + m) * 64);
      seg_main<4, 2, kPrefetch1, ROWB>(acc1, ring1, ws, a1x, NGX * 64, NGX, xrow, swz);
      gemm_z<4, 2>(acc1, az1, zb);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) lrelu(acc1[m][n]);
    }

    // ---------------- layer 2: rows [64 wv, +64), K = 512 hidden (8 chunks) + skip ----------------
    f32x16 acc2[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      init_from_bias(acc2[m][0], ws, mlp.bias[2] + 32 * (2 * wv + m));
      acc2[m][1] = acc2[m][0];
    }
    {
      const int a2 = mlp.ah[2] / 4 + (2 * wv) * (kHidden[1] / 8) * 64;
      f32x4 ring2[2][2];
      seg_prefetch<2, 1>(ring2, ws, a2, (kHidden[1] / 8) * 64, 8);
#pragma unroll
      for (int ck = 0; ck < 8; ++ck) {
        if (wv == (ck >> 1)) {  // owner of hidden rows [64 ck, +64): row blocks 2(ck&1), +1
#pragma unroll
          for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int n = 0; n < 2; ++n) store_hidden(hb, acc1[2 * (ck & 1) + mm][n], mm, n, j, h);
        }
        MP_CHUNK_SYNC();
        seg_main<2, 2, 1, kHbRowBytes>(acc2, ring2, ws, a2 + ck * 8 * 64, (kHidden[1] / 8) * 64, 8,
                                       hrow, swz);
        if (ck < 7) seg_prefetch<2, 1>(ring2, ws, a2 + (ck + 1) * 8 * 64, (kHidden[1] / 8) * 64, 8);
        MP_CHUNK_SYNC();
      }
      const int a2x = mlp.ax[2] / 4 + (2 * wv) * NGX * 64;
      float az2[2];
      seg_prefetch<2, 1>(ring2, ws, a2x, NGX * 64, NGX);
#pragma unroll
      for (int m = 0; m < 2; ++m) az2[m] = wload32(ws, mlp.az[2] + (2 * wv + m) * 64);
      seg_main<2, 2, 1, ROWB>(acc2, ring2, ws, a2x, NGX * 64, NGX, xrow, swz);
      gemm_z<2, 2>(acc2, az2, zb);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) lrelu(acc2[m][n]);
    }

    // ---------------- layer 3: rows [32 wv, +32), K = 256 hidden (4 chunks) + skip ----------------
    f32x16 acc3[1][2];
    init_from_bias(acc3[0][0], ws, mlp.bias[3] + 32 * wv);
    acc3[0][1] = acc3[0][0];
    {
      const int a3 = mlp.ah[3] / 4 + wv * (kHidden[2] / 8) * 64;
      f32x4 ring3[4][1];
      seg_prefetch<1, 3>(ring3, ws, a3, 0, 8);
#pragma unroll
      for (int ck = 0; ck < 4; ++ck) {
        if (wv == ck) {
#pragma unroll
          for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int n = 0; n < 2; ++n) store_hidden(hb, acc2[mm][n], mm, n, j, h);
        }
        MP_CHUNK_SYNC();
        seg_main<1, 2, 3, kHbRowBytes>(acc3, ring3, ws, a3 + ck * 8 * 64, 0, 8, hrow, swz);
        if (ck < 3) seg_prefetch<1, 3>(ring3, ws, a3 + (ck + 1) * 8 * 64, 0, 8);
        MP_CHUNK_SYNC();
      }
      const int a3x = mlp.ax[3] / 4 + wv * NGX * 64;
      float az3[1];
      seg_prefetch<1, 3>(ring3, ws, a3x, 0, NGX);
      az3[0] = wload32(ws, mlp.az[3] + wv * 64);
      seg_main<1, 2, 3, ROWB>(acc3, ring3, ws, a3x, 0, NGX, xrow, swz);
      gemm_z<1, 2>(acc3, az3, zb);
#pragma unroll
      for (int n = 0; n < 2; ++n) lrelu(acc3[0][n]);
    }

    // ---------------- layer 4 (Cout x (128 + C + 1)) on the VALU ----------------
    // red[part][o][p]: parts 0-3 = hidden rows of wave `part`, parts 4-7 = feature quarter
    float *red = reinterpret_cast<float *>(hb);
    constexpr int K4 = (kHidden[3] + C + 1 + 3) & ~3;  // padded row stride (pack.hip)
    {
      // hidden part: this lane holds rows 32 wv + 8q + 4h + i of points 32 cb + j
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float *w4 = (mlp.base + mlp.w4) + o * K4 + 32 * wv + 4 * h;
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = *reinterpret_cast<const f32x4 *>(w4 + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s0 = fmaf(wq[i], acc3[0][0][4 * q + i], s0);
            s1 = fmaf(wq[i], acc3[0][1][4 * q + i], s1);
          }
        }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (h == 0) {
          red[(wv * COUT + o) * kTilePts + j] = s0;
          red[(wv * COUT + o) * kTilePts + 32 + j] = s1;
        }
      }
      // feature part: lane = point, wave = quarter of the C channels
      const int p = lane;
      float sx[COUT];
#pragma unroll
      for (int o = 0; o < COUT; ++o) sx[o] = 0.0f;
      constexpr int SLOTS = C / 16;  // 16-byte slots per quarter
#pragma unroll 4
      for (int s = 0; s < SLOTS; ++s) {
        const int slot = wv * SLOTS + s;
        const f32x4 xv =
            *reinterpret_cast<const f32x4 *>(xs + p * ROWB + ((slot ^ (p & 15)) << 4));
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          const f32x4 wq =
              *reinterpret_cast<const f32x4 *>((mlp.base + mlp.w4) + o * K4 + kHidden[3] + 4 * slot);
#pragma unroll
          for (int i = 0; i < 4; ++i) sx[o] = fmaf(wq[i], xv[i], sx[o]);
        }
      }
#pragma unroll
      for (int o = 0; o < COUT; ++o) red[((4 + wv) * COUT + o) * kTilePts + p] = sx[o];
    }
    __syncthreads();
    if (tid < COUT * kTilePts) {
      const int o = tid / kTilePts, p = tid % kTilePts;
      const long long n = n0 + p;
      if (n < n_pts) {
        float v = (mlp.base + mlp.bias[4])[o];
#pragma unroll
        for (int part = 0; part < 8; ++part) v += red[(part * COUT + o) * kTilePts + p];
        const float wz = (mlp.base + mlp.w4)[o * K4 + kHidden[3] + C];
        if constexpr (DIRECT) {
          v = fmaf(wz, src.pts[(long long)C * src.sc + n], v);
          out[o * src.out_stride + n] = activate(v, act);
        } else {
          float cal[12];
#pragma unroll
          for (int i = 0; i < 12; ++i) cal[i] = calib[i];
          float px, py, pz, x, y, z;
          uint32_t code;
          load_point(src, n, px, py, pz, code);
          project(cal, px, py, pz, x, y, z);
          v = fmaf(wz, __fmul_rn(z, z_scale), v);
          v = in_image(x, y) ? activate(v, act) : 0.0f;  // MonoPortNet.py:89
          if (src.packed) {
            const int ix = code & 1023u, iy = (code >> 10) & 1023u, iz = code >> 20;
            out[((long long)iz * src.level_res + iy) * src.level_res + ix] = v;
          } else {
            out[o * src.out_stride + n] = v;
          }
        }
      }
    }
    __syncthreads();  // red / xs are rewritten by the next tile
  }
}

// ---- stand-alone index() and orthogonal() -------------------------------------------------------
// geometry.py:4-16.  One wave per point and 256-channel slice; out is [C, N] like the reference.
__global__ __launch_bounds__(256) void index_kernel(const float *__restrict__ feat, int c, int fh,
                                                    int fw, const float *__restrict__ uv,
                                                    long long n, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
  const int slots = c / 4;
  for (long long i = wave; i < n; i += n_waves) {
    const float x = uv[i], y = uv[n + i];
    const Taps t = make_taps(x, y, fh, fw, c, true);
    for (int slot = lane; slot < slots; slot += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(feat + t.o[0] + 4 * slot);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(feat + t.o[1] + 4 * slot);
      const f32x4 cc = *reinterpret_cast<const f32x4 *>(feat + t.o[2] + 4 * slot);
      const f32x4 d = *reinterpret_cast<const f32x4 *>(feat + t.o[3] + 4 * slot);
      const f32x4 r = blend(a, b, cc, d, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) out[(long long)(4 * slot + k) * n + i] = r[k];
    }
  }
}

__global__ void orthogonal_kernel(const float *__restrict__ pts, long long n,
                                  const float *__restrict__ calib, float *__restrict__ out) {
  float cal[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) cal[i] = calib[i];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float x, y, z;
    project(cal, pts[i], pts[n + i], pts[2 * n + i], x, y, z);
    out[i] = x;
    out[n + i] = y;
    out[2 * n + i] = z;
  }
}

// ---- host side -----------------------------------------------------------------------------------
template <int C, int COUT, int WPS, bool DIRECT>
static int launch_query_t(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w,
                          float z_scale, long long max_points, bool device_counts,
                          hipStream_t st) {
  constexpr int lds = kTilePts * C * 4 + kHbBytes;
  auto kern = pifu_query_kernel<C, COUT, WPS, DIRECT>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {  // once per kernel and context (= device)
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ctx->lds_attr_done.insert(kern_id);
  }
  if (max_points <= 0) return MP_OK;
  // every frame of the set rounds its own tail tile up
  const long long tiles = (max_points + kTilePts - 1) / kTilePts + (set.n - 1);
  const long long resident = (long long)cus_of(ctx, st) * WPS;
  // device-side counts: launch the resident grid and let it stride; host-side counts: one
  // workgroup per tile up to a few waves of the machine
  long long grid = device_counts ? (tiles < resident ? tiles : resident)
                                 : (tiles < 8 * resident ? tiles : 8 * resident);
  // Launches with few 64-point tiles (long tail on 256 CUs) go to the 32-point kernel
  // (query_small.hip, same bits).  With host-side counts the choice is made here; with device-side
  // counts both kernels are launched and each looks at the counts (the excluded one leaves at its
  // first instruction).
  int small = 0;  // 0 = this kernel only, 1 = the 32-point kernel only, 2 = both, gated
  int gate = 0;
  bool table = false;
  QuerySet tset;  // the set with the layer-0 tables filled in (mp_skip_table), if every frame has one
  if constexpr (C == 256 && !DIRECT) {
    table = find_skip_tables(ctx, m, set, h, w, tset);
    gate = query_small_gate();
    if (gate == 1) {
      small = 1;
    } else if (gate > 1) {
      if (!device_counts) {
        long long t64 = 0;
        for (int f = 0; f < set.n; ++f) t64 += (set.it[f].src.n + kTilePts - 1) / kTilePts;
        small = t64 < gate ? 1 : 0;
      } else {
        small = tiles < gate ? 1 : 2;
      }
    }
  }
  const bool prof = 2 * (ctx->prof_used + 1) <= (int)ctx->prof_events.size();
  if (prof) MP_HIP(ctx, hipEventRecord(ctx->prof_events[2 * ctx->prof_used], st));
  if (table) {  // query_table.hip: 32-point tiles at every launch size
    small = 1;
    const int rc = launch_query_tab(ctx, m, tset, h, w, z_scale, max_points, device_counts, st);
    if (rc != MP_OK) return rc;
  } else if (small) {
    const int rc = launch_query32(ctx, m, set, h, w, z_scale, max_points, device_counts,
                                  small == 2 ? gate : 0, st);
    if (rc != MP_OK) return rc;
  }
  if (small != 1) {
    QuerySetDev dset;
    const int rc = compact_query_set(ctx, set, dset);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kQueryThreads), lds, st, m.pack(), h, w,
                       z_scale, m.act, dset, small == 2 ? gate : 0);
  }
  if (prof) {
    MP_HIP(ctx, hipEventRecord(ctx->prof_events[2 * ctx->prof_used + 1], st));
    ++ctx->prof_used;
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// tset = set with the skip table of every frame's feature map filled in; false unless EVERY map has a table
// registered that was made with this head and map size (mp_skip_table)
bool find_skip_tables(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, QuerySet &tset) {
  if (ctx->skip_tables.empty()) return false;
  tset = set;
  for (int f = 0; f < set.n; ++f) {
    auto it = ctx->skip_tables.find(set.it[f].feat);
    if (it == ctx->skip_tables.end() || it->second.mlp_buf != m.buf || it->second.h != h || it->second.w != w) return false;
    tset.it[f].l0 = it->second.table;
  }
  return true;
}

int compact_query_set(mp_ctx *ctx, const QuerySet &set, QuerySetDev &d) {
  std::memset(&d, 0, sizeof(d));
  const PointSrc &s0 = set.it[0].src;
  d.n = set.n;
  d.lattice = s0.packed != nullptr;
  d.sn = s0.sn;
  d.sc = s0.sc;
  d.out_stride = s0.out_stride;
  d.stride = s0.stride;
  d.level_res = s0.level_res;
  d.res_final = s0.res_final;
  d.half_step = s0.half_step;
  for (int i = 0; i < 3; ++i) {
    d.bmin[i] = s0.bmin[i];
    d.blen[i] = s0.blen[i];
  }
  for (int f = 0; f < set.n; ++f) {
    const QueryItem &q = set.it[f];
    const PointSrc &s = q.src;
    bool same = (s.packed != nullptr) == (s0.packed != nullptr) && s.sn == s0.sn && s.sc == s0.sc &&
                s.out_stride == s0.out_stride && s.stride == s0.stride && s.level_res == s0.level_res &&
                s.res_final == s0.res_final && s.half_step == s0.half_step;
    for (int i = 0; i < 3; ++i) same = same && s.bmin[i] == s0.bmin[i] && s.blen[i] == s0.blen[i];
    if (!same)
      return fail(ctx, MP_ERR_ARG, "query: the frames of one launch must share the point layout / octree level (frame %d differs)", f);
    QueryItemDev &o = d.it[f];
    o.feat = q.feat;
    o.calib = q.calib;
    o.out = q.out;
    o.l0 = q.l0;
    o.pts = s.packed ? static_cast<const void *>(s.packed) : static_cast<const void *>(s.pts);
    o.n_dev = s.n_dev;
    o.n = s.n;
  }
  return MP_OK;
}

int launch_query_set(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                     long long max_points, bool device_counts, hipStream_t st) {
  if (set.n < 1 || set.n > kMaxFrames)
    return fail(ctx, MP_ERR_ARG, "query: 1..%d frames per launch, got %d", kMaxFrames, set.n);
  const bool direct = set.it[0].feat == nullptr;
  if (m.precision != MP_PREC_F32 && !direct && m.c == 256)
    return launch_query16(ctx, m, set, h, w, z_scale, max_points, device_counts, st);
#define MP_QCASE(CC, CO, WP)                                                                     \
  if (m.c == CC && m.cout == CO) {                                                               \
    if (direct)                                                                                  \
      return launch_query_t<CC, CO, WP, true>(ctx, m, set, h, w, z_scale, max_points,            \
                                              device_counts, st);                                \
    return launch_query_t<CC, CO, WP, false>(ctx, m, set, h, w, z_scale, max_points,             \
                                             device_counts, st);                                 \
  }
  MP_QCASE(256, 1, 2)
  MP_QCASE(256, 3, 2)
  MP_QCASE(512, 1, 1)
  MP_QCASE(512, 3, 1)
#undef MP_QCASE
  return fail(ctx, MP_ERR_UNSUPPORTED, "query kernels are built for C in {256,512}, Cout in {1,3}; got C=%d Cout=%d",
              m.c, m.cout);
}

int launch_query(mp_ctx *ctx, const Mlp &m, const float *feat, int h, int w, const float *calib,
                 float z_scale, const PointSrc &src, float *out, long long max_points,
                 hipStream_t st) {
  QuerySet set;
  std::memset(&set, 0, sizeof(set));
  set.n = 1;
  set.it[0].feat = feat;
  set.it[0].calib = calib;
  set.it[0].out = out;
  set.it[0].src = src;
  return launch_query_set(ctx, m, set, h, w, z_scale, max_points, src.n_dev != nullptr, st);
}

int launch_index(mp_ctx *ctx, const float *feat, int c, int h, int w, const float *uv, long long n,
                 float *out, hipStream_t st) {
  if (c % 4) return fail(ctx, MP_ERR_UNSUPPORTED, "index: C must be a multiple of 4, got %d", c);
  if (n <= 0) return MP_OK;
  long long blocks = (n + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(index_kernel, dim3((unsigned)blocks), dim3(256), 0, st, feat, c, h, w, uv, n,
                     out);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_orthogonal(mp_ctx *ctx, const float *pts, long long n, const float *calib, float *out,
                      hipStream_t st) {
  if (n <= 0) return MP_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(orthogonal_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pts, n, calib,
                     out);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
