// The fused f32 query of the netG heads (C = 256, sampled features) on 32-point tiles, for the
// launches that a 64-point tiling leaves with a long tail.
//
// Same algorithm, same arithmetic in the same order per point (K order, FMA chains, bias first)
// as pifu_query_kernel (query.hip, 64-point tiles) -- the two return identical bits
// (tests/test_query_gpu.py::test_small_tile_kernel_is_bit_identical) -- but a workgroup takes 32
// points: half the MFMA work per tile, twice the tiles, 48 KB of LDS and 64 accumulator registers
// for layer 1, so three workgroups fit a CU.  Measured on MI355X (profiles/r03x_small_tile_probe.txt,
// profiles/r03w_query_gate_sweep.txt):
//   * a CU finishes a tile in 137 us instead of 280 us, so a launch takes 0.137 ms x
//     ceil(points / 8192) instead of 0.28 ms x ceil(points / 16384): the coarse octree levels of a
//     single frame (77 / 142 / 392 64-point tiles on 256 CUs) finish in 0.17 / 0.30 / 0.57 ms
//     instead of 0.31 / 0.31 / 0.57 ms, a 38 k-point level in 0.71 instead of 0.86 ms; one frame's
//     mp_recon at 257^3 6.0 -> 5.7 ms, four frames' 21.7 -> 21.3 ms;
//   * alone on the chip it is never slower up to 262 144 points (4.38 against 4.55 ms), but it
//     reads twice the weight bytes per point from L2 (4.66 MB per tile, 34 GB/s per CU): on the
//     16-frame launches of the bench (3.4 M points) it is level to 2 % slower alone, and with the
//     encoders of the two other pipeline slots running next to it the whole bench lost 3 %
//     (135.5 against 139.2 recon/s).
// The launcher therefore sends a launch here when it has fewer than kSmallGateTiles (2048)
// 64-point tiles (mp_query_tune moves the gate).  The point counts of octree levels >= 1 live on
// the DEVICE: for those both kernels are launched and each reads the counts -- the one the gate
// excludes leaves at its first instruction (2-3 us per level).
//
// Decomposition: layer 0 in 128-row chunks (one 32-row block per wave, the one column block),
// each chunk = 16 k-groups of layer 1, whose 512 x 32 accumulator tile is spread over the 4 waves
// (4 row blocks each); layers 2-4 as in query.hip with one column block.
#include <cstring>

#include "mp_internal.h"
#include "query_common.h"
#include "query_mfma.h"

#pragma clang fp contract(off)

namespace mp {

constexpr int kSmallPts = 32;
constexpr int kT32Wps = 3;  // workgroups per CU the register allocator is held to
constexpr int kSmallHbRow = 128 * 4;  // bytes per point of a 128-row hidden chunk

template <int COUT>
__global__ __launch_bounds__(kQueryThreads, kT32Wps) void pifu_query_t32_kernel(
    MlpPack mlp, int fh, int fw, float z_scale, int act, QuerySetDev set, int gate_tiles64) {
  constexpr int C = 256;
  constexpr int P = kSmallPts;
  constexpr int ROWB = C * 4;
  constexpr int NGX = C / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *xs = smem;             // [32 points][C] f32, swizzled 16-byte slots
  unsigned char *hb = smem + P * ROWB;  // hidden chunk: [32][128 rows] (layer 0 -> 1) or [32][64 rows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int swz = h ^ (j & 15);
  const WStream ws = make_wstream(mlp.base, mlp.n_floats, lane);

  for (long long gtile = blockIdx.x;; gtile += gridDim.x) {
    int fi = -1;
    long long tile0 = 0;
    {
      long long acc = 0, acc64 = 0;
      // groups of 8 frames: the 8 count loads of a group are in flight together, and the dynamic group offset
      // keeps the compiler from hoisting all kMaxFrames kernel-argument loads into SGPRs (spills)
      for (int f0 = 0; f0 < set.n; f0 += 8)
#pragma unroll
      for (int fk = 0; fk < 8; ++fk) {
        const int f = f0 + fk;
        if (f < set.n) {
          const long long nf = set.count(f);
          const long long t = (nf + P - 1) / P;
          if (fi < 0 && gtile < acc + t) {
            fi = f;
            tile0 = acc;
          }
          acc += t;
          acc64 += (nf + kTilePts - 1) / kTilePts;
        }
      }
      if (gate_tiles64 > 0 && acc64 >= gate_tiles64) break;  // the 64-point kernel serves this launch
    }
    if (fi < 0) break;
    const QueryItem item = set.item(fi);
    const float *__restrict__ feat = item.feat;
    const float *__restrict__ calib = item.calib;
    float *__restrict__ out = item.out;
    const PointSrc &src = item.src;
    const long long n_pts = src.n_dev ? (long long)*src.n_dev : src.n;
    const long long n0 = (gtile - tile0) * P;

    // ---------------- gather: 8 points per wave ----------------
    float zb[1];
    {
      float cal[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) cal[i] = calib[i];
      constexpr int GB = 4, PW = P / 4;
#pragma unroll 1
      for (int i0 = 0; i0 < PW; i0 += GB) {
        Taps t[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const long long n = n0 + PW * wv + i0 + u;
          const bool live_n = n < n_pts;
          float px = 0, py = 0, pz = 0, x, y, z;
          uint32_t code;
          if (live_n) load_point(src, n, px, py, pz, code);
          project(cal, px, py, pz, x, y, z);
          t[u] = make_taps(x, y, fh, fw, C, live_n && in_image(x, y));
        }
        f32x4 v[GB][4];
#pragma unroll
        for (int u = 0; u < GB; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[u][k] = *reinterpret_cast<const f32x4 *>(feat + t[u].o[k] + 4 * lane);
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int p = PW * wv + i0 + u;
          const f32x4 r = blend(v[u][0], v[u][1], v[u][2], v[u][3], t[u]);
          *reinterpret_cast<f32x4 *>(xs + p * ROWB + ((lane ^ (p & 15)) << 4)) = r;
        }
      }
      {
        const long long n = n0 + j;
        float px = 0, py = 0, pz = 0, x, y, z;
        uint32_t code;
        if (n < n_pts) load_point(src, n, px, py, pz, code);
        project(cal, px, py, pz, x, y, z);
        zb[0] = (h == 0 && n < n_pts) ? __fmul_rn(z, z_scale) : 0.0f;
      }
    }
    __syncthreads();

    const unsigned char *xrow = xs + j * ROWB;
    const unsigned char *hrow = hb + j * kHbRowBytes;   // 64-row chunks (layers 2, 3)
    const unsigned char *hrow1 = hb + j * kSmallHbRow;  // 128-row chunks (layer 0 -> 1)

    // ---------------- layers 0 + 1, fused over 128-row chunks of layer 0 ----------------
    f32x16 acc1[4][1];
#pragma unroll
    for (int m = 0; m < 4; ++m) init_from_bias(acc1[m][0], ws, mlp.bias[1] + 32 * (4 * wv + m));
    {
      const int a0 = mlp.ax[0] / 4;
      const int rs1 = (kHidden[0] / 8) * 64;
      const int a1 = mlp.ah[1] / 4 + (4 * wv) * rs1;
      f32x4 ring0[kPrefetch0 + 1][1];
      f32x16 acc0[1][1];
      float az0[1];
      seg_prefetch<1, kPrefetch0>(ring0, ws, a0 + wv * NGX * 64, 0, NGX);
      init_from_bias(acc0[0][0], ws, mlp.bias[0] + 32 * wv);
      az0[0] = wload32(ws, mlp.az[0] + wv * 64);
#pragma unroll 1
      for (int ck = 0; ck < kHidden[0] / 128; ++ck) {
        const int rb = 4 * ck + wv;  // layer-0 rows [32 rb, +32) x the 32 points
        seg_main<1, 1, kPrefetch0, ROWB>(acc0, ring0, ws, a0 + rb * NGX * 64, 0, NGX, xrow, swz);
        f32x4 ring1[kPrefetch1 + 1][4];
        seg_prefetch<4, kPrefetch1>(ring1, ws, a1 + ck * 16 * 64, rs1, 16);
        gemm_z<1, 1>(acc0, az0, zb);
        lrelu(acc0[0][0]);
        store_hidden<kSmallHbRow>(hb, acc0[0][0], wv, 0, j, h);
        const int rbn = min(rb + 4, kHidden[0] / 32 - 4 + wv);
        seg_prefetch<1, kPrefetch0>(ring0, ws, a0 + rbn * NGX * 64, 0, NGX);
        init_from_bias(acc0[0][0], ws, mlp.bias[0] + 32 * rbn);
        az0[0] = wload32(ws, mlp.az[0] + rbn * 64);
        __syncthreads();
        // layer-1 rows [128 wv, +128) += W1[:, 128 ck .. +128) * chunk
        seg_main<4, 1, kPrefetch1, kSmallHbRow>(acc1, ring1, ws, a1 + ck * 16 * 64, rs1, 16, hrow1, swz);
        __syncthreads();
      }
      const int a1x = mlp.ax[1] / 4 + (4 * wv) * NGX * 64;
      f32x4 ring1[kPrefetch1 + 1][4];
      float az1[4];
      seg_prefetch<4, kPrefetch1>(ring1, ws, a1x, NGX * 64, NGX);
#pragma unroll
      for (int m = 0; m < 4; ++m) az1[m] = wload32(ws, mlp.az[1] + (4 * wv + m) * 64);
      seg_main<4, 1, kPrefetch1, ROWB>(acc1, ring1, ws, a1x, NGX * 64, NGX, xrow, swz);
      gemm_z<4, 1>(acc1, az1, zb);
#pragma unroll
      for (int m = 0; m < 4; ++m) lrelu(acc1[m][0]);
    }

    // ---------------- layer 2: rows [64 wv, +64), K = 512 hidden (8 chunks of 64) + skip ----------------
    f32x16 acc2[2][1];
#pragma unroll
    for (int m = 0; m < 2; ++m) init_from_bias(acc2[m][0], ws, mlp.bias[2] + 32 * (2 * wv + m));
    {
      const int rs2 = (kHidden[1] / 8) * 64;
      const int a2 = mlp.ah[2] / 4 + (2 * wv) * rs2;
      f32x4 ring2[2][2];
      seg_prefetch<2, 1>(ring2, ws, a2, rs2, 8);
#pragma unroll
      for (int ck = 0; ck < 8; ++ck) {
        if (wv == (ck >> 1)) {  // owner of hidden rows [64 ck, +64): row blocks 2 (ck & 1), + 1
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) store_hidden(hb, acc1[2 * (ck & 1) + mm][0], mm, 0, j, h);
        }
        __syncthreads();
        seg_main<2, 1, 1, kHbRowBytes>(acc2, ring2, ws, a2 + ck * 8 * 64, rs2, 8, hrow, swz);
        if (ck < 7) seg_prefetch<2, 1>(ring2, ws, a2 + (ck + 1) * 8 * 64, rs2, 8);
        __syncthreads();
      }
      const int a2x = mlp.ax[2] / 4 + (2 * wv) * NGX * 64;
      float az2[2];
      seg_prefetch<2, 1>(ring2, ws, a2x, NGX * 64, NGX);
#pragma unroll
      for (int m = 0; m < 2; ++m) az2[m] = wload32(ws, mlp.az[2] + (2 * wv + m) * 64);
      seg_main<2, 1, 1, ROWB>(acc2, ring2, ws, a2x, NGX * 64, NGX, xrow, swz);
      gemm_z<2, 1>(acc2, az2, zb);
#pragma unroll
      for (int m = 0; m < 2; ++m) lrelu(acc2[m][0]);
    }

    // ---------------- layer 3: rows [32 wv, +32), K = 256 hidden (4 chunks) + skip ----------------
    f32x16 acc3[1][1];
    init_from_bias(acc3[0][0], ws, mlp.bias[3] + 32 * wv);
    {
      const int a3 = mlp.ah[3] / 4 + wv * (kHidden[2] / 8) * 64;
      f32x4 ring3[4][1];
      seg_prefetch<1, 3>(ring3, ws, a3, 0, 8);
#pragma unroll
      for (int ck = 0; ck < 4; ++ck) {
        if (wv == ck) {
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) store_hidden(hb, acc2[mm][0], mm, 0, j, h);
        }
        __syncthreads();
        seg_main<1, 1, 3, kHbRowBytes>(acc3, ring3, ws, a3 + ck * 8 * 64, 0, 8, hrow, swz);
        if (ck < 3) seg_prefetch<1, 3>(ring3, ws, a3 + (ck + 1) * 8 * 64, 0, 8);
        __syncthreads();
      }
      const int a3x = mlp.ax[3] / 4 + wv * NGX * 64;
      float az3[1];
      seg_prefetch<1, 3>(ring3, ws, a3x, 0, NGX);
      az3[0] = wload32(ws, mlp.az[3] + wv * 64);
      seg_main<1, 1, 3, ROWB>(acc3, ring3, ws, a3x, 0, NGX, xrow, swz);
      gemm_z<1, 1>(acc3, az3, zb);
      lrelu(acc3[0][0]);
    }

    // ---------------- layer 4 (Cout x (128 + C + 1)) on the VALU ----------------
    // red[part][o][p]: parts 0-3 = hidden rows of wave `part`, parts 4-7 = feature quarter
    float *red = reinterpret_cast<float *>(hb);
    constexpr int K4 = (kHidden[3] + C + 1 + 3) & ~3;
    {
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float *w4 = (mlp.base + mlp.w4) + o * K4 + 32 * wv + 4 * h;
        float s0 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = *reinterpret_cast<const f32x4 *>(w4 + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) s0 = fmaf(wq[i], acc3[0][0][4 * q + i], s0);
        }
        s0 += __shfl_xor(s0, 32);
        if (h == 0) red[(wv * COUT + o) * P + j] = s0;
      }
      // feature part: lane = point (lanes 0-31), wave = quarter of the C channels
      if (lane < P) {
        const int p = lane;
        float sx[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) sx[o] = 0.0f;
        constexpr int SLOTS = C / 16;
#pragma unroll 4
        for (int s = 0; s < SLOTS; ++s) {
          const int slot = wv * SLOTS + s;
          const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + p * ROWB + ((slot ^ (p & 15)) << 4));
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const f32x4 wq =
                *reinterpret_cast<const f32x4 *>((mlp.base + mlp.w4) + o * K4 + kHidden[3] + 4 * slot);
#pragma unroll
            for (int i = 0; i < 4; ++i) sx[o] = fmaf(wq[i], xv[i], sx[o]);
          }
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o) red[((4 + wv) * COUT + o) * P + p] = sx[o];
      }
    }
    __syncthreads();
    if (tid < COUT * P) {
      const int o = tid / P, p = tid % P;
      const long long n = n0 + p;
      if (n < n_pts) {
        float v = (mlp.base + mlp.bias[4])[o];
#pragma unroll
        for (int part = 0; part < 8; ++part) v += red[(part * COUT + o) * P + p];
        const float wz = (mlp.base + mlp.w4)[o * K4 + kHidden[3] + C];
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
    __syncthreads();  // red / xs are rewritten by the next tile
  }
}

// launches of fewer than this many 64-point tiles run on 32-point tiles; 0 = never, 1 = always
static int g_small_gate = kSmallGateTiles;
void query_small_set_gate(int gate) { g_small_gate = gate < 0 ? kSmallGateTiles : gate; }
int query_small_gate() { return g_small_gate; }

template <int COUT>
int launch_query32_t(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                     long long max_points, bool device_counts, int gate_tiles64, hipStream_t st) {
  constexpr int lds = kSmallPts * 256 * 4 + kSmallPts * kSmallHbRow;
  auto kern = pifu_query_t32_kernel<COUT>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ctx->lds_attr_done.insert(kern_id);
  }
  if (max_points <= 0) return MP_OK;
  const long long tiles = (max_points + kSmallPts - 1) / kSmallPts + (set.n - 1);
  const long long resident = (long long)cus_of(ctx, st) * kT32Wps;
  // device-side counts: launch the resident grid and let it stride; host-side counts: one
  // workgroup per tile up to a few waves of the machine
  long long grid = device_counts ? (tiles < resident ? tiles : resident)
                                 : (tiles < 8 * resident ? tiles : 8 * resident);
  // gated: this kernel only works on launches of < gate_tiles64 64-point tiles
  if (gate_tiles64 > 0 && grid > 2LL * gate_tiles64 + set.n) grid = 2LL * gate_tiles64 + set.n;
  QuerySetDev dset;
  {
    const int rc_set = compact_query_set(ctx, set, dset);
    if (rc_set != MP_OK) return rc_set;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kQueryThreads), lds, st, m.pack(), h, w, z_scale,
                     m.act, dset, gate_tiles64);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_query32(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                   long long max_points, bool device_counts, int gate_tiles64, hipStream_t st) {
  if (m.cout == 1)
    return launch_query32_t<1>(ctx, m, set, h, w, z_scale, max_points, device_counts, gate_tiles64, st);
  if (m.cout == 3)
    return launch_query32_t<3>(ctx, m, set, h, w, z_scale, max_points, device_counts, gate_tiles64, st);
  return fail(ctx, MP_ERR_UNSUPPORTED, "query32: Cout in {1,3}");
}

}  // namespace mp
