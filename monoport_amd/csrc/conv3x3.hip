// 3x3 convolutions of the image encoders on f32 MFMA, with the GroupNorm around them fused in
// (backbones/HGFilters.py:40-62 ConvBlock, :87-111 HourGlass).
//
// The reference's pyramid block runs  conv(relu(GroupNorm(x)))  three times.  Under MIOpen that is,
// per convolution, a statistics pass, a normalise+ReLU pass (read + write of the whole tensor) and
// an fp32 Winograd kernel (100-115 TFLOP/s direct-equivalent on the 128^2 maps, 20-60 on the small ones).  Here:
//   * ONE kernel per convolution: implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32, FMA chain
//     per output like the fused MLP kernel), M = output channels, N = pixels, K = (tap, cin);
//   * GroupNorm + ReLU are applied to the INPUT while it is staged into LDS:
//     v = max(x * scale[n,c] + shift[n,c], 0), zero outside the image (the convolution pads the
//     normalised tensor, HGFilters.py:15-19 padding=1);
//   * the epilogue writes NCHW and hands the statistics the NEXT GroupNorm needs on (round 3,
//     csrc/gn_tail.h): per-channel sums (f32 over <= 128 values, then double) meet in LDS, are folded
//     to per-group sums in a fixed order and ADDED to the consumer's accumulator with integer
//     atomics -- no statistics pass, no finalize launch, deterministic; it can also write the pyramid
//     block's tail  y2 = cat(out1, out2, out3) + residual  (HGFilters.py:57-60) for its channels;
//   * a consumer turns its input's accumulator into (mean, rstd) in its prologue (gn_load_stats);
//   * weights stream from L2 in MFMA fragment order through a buffer resource (conv3x3_pack_kernel),
//     the activation tile is staged pixel-major, XOR-swizzled, double-buffered over 16-channel chunks.
//
// Decomposition: workgroup = 4 waves = (32 RBW) output channels x (32 NR 4/RBW) pixels; wave =
// one 32-row block x NR column blocks of 32 consecutive pixels of one image row.
//   Cout = 128: RBW = 4 (64 / 32-pixel tiles);  Cout = 64: RBW = 2 (128 / 64 pixels);
//   Cout = 32:  RBW = 1 (256 / 128 pixels);  NR = 2, or 1 when the launch would be too small.
// Algorithmic work: 2 * 9 * Cin * Cout FLOP per output pixel; roofline = f32 MFMA (157.3 TFLOP/s).
#include "mp_internal.h"
#include "query_common.h"
#include "gn_tail.h"

namespace mp {

constexpr int kCK = 16;           // input channels per LDS chunk
constexpr int kPixBytes = kCK * 4;  // 64 bytes per staged pixel
constexpr int kMaxCin = 512;  // input channels of a 3x3 launch (the per-workgroup (scale, shift) table)
constexpr int kConvWps = 2;  // waves per SIMD the register allocator is held to
constexpr int kTileW = 32;        // tiles are 32 pixels wide and PX / 32 rows tall
// staged pixels (tile + 1 halo) of a PX-pixel tile, and the 64-lane passes a wave needs for them
constexpr int halo_pixels(int px) { return (px / kTileW + 2) * (kTileW + 2); }
constexpr int stage_iters(int px) { return (halo_pixels(px) + 63) / 64; }

// W [Cout][Cin][3][3] -> fragment order: group kg = (chunk * 9 + tap) * 2 + g holds, for lane
// (r = lane & 31, hh = lane >> 5), the 4 weights W[32 rb + r][16 chunk + 8 g + 4 hh + i][tap].
__global__ void conv3x3_pack_kernel(const float *__restrict__ w, int cout, int cin,
                                    float *__restrict__ wp) {
  const long long total = (long long)cout * cin * 9;
  const int kgt = (cin / kCK) * 18;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t & 3), lane = (int)((t >> 2) & 63);
    const long long q = t >> 8;
    const int kg = (int)(q % kgt), rb = (int)(q / kgt);
    const int g = kg & 1, tap = (kg >> 1) % 9, chunk = (kg >> 1) / 9;
    const int co = 32 * rb + (lane & 31);
    const int ci = kCK * chunk + 8 * g + 4 * (lane >> 5) + i;
    wp[t] = w[((long long)co * cin + ci) * 9 + tap];
  }
}


// ---- epilogue shared by the 3x3 kernels ------------------------------------------------------
// A wave holds rows r(t) = (t & 3) + 8 (t >> 2) + 4 h, t in [t0, t0 + TN), of the 32-row block
// `rbi` of its workgroup (NCH output channels, CW pixel-column waves), for NR column blocks of 32
// consecutive pixels of one tile row.  Writes the raw output (y), the pyramid-block tail
// y2 = conv + res (optional) and publishes the GroupNorm statistics of either (gn_tail.h): the
// per-channel sums of all waves meet in LDS, threads 0..ng-1 fold them into the workgroup's
// per-group partials in a fixed order.  smem: the kernel's dynamic LDS, idle by now (>= 8.5 KB).
template <int NCH, int CW, int NR, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &p, const float (&v)[NR][TN], int t0, int img,
                                              int tile, int tiles, int y0, int x0, int rbi, int cwi, int lane,
                                              unsigned char *smem) {
  const int j = lane & 31, h = lane >> 5;
  const int hw = p.h * p.w;
  const int ch0 = NCH * blockIdx.y + 32 * rbi;
  const bool cat = p.y2 != nullptr;
  const bool st1 = gn_wanted(p.fin), st2 = cat && gn_wanted(p.fin2);
  const int tid = threadIdx.x;
  // Addresses (round 4): one buffer resource per tensor and image; the lane part of value (n, tt) -- 4 h channels
  // down, the pixel of column block n -- in NR registers, the channel of register tt in the scalar offset.  No
  // per-element 64-bit address arithmetic on the VALU (recomputing it per use was the price of not keeping 16 NR
  // offsets alive) and a smaller register footprint.
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
      p.y ? p.y + (long long)img * p.cout * hw : const_cast<float *>(p.x), 0, p.y ? p.cout * hw * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y2 = __builtin_amdgcn_make_buffer_rsrc(
      cat ? p.y2 + (long long)img * p.y2_c * hw : const_cast<float *>(p.x), 0, cat ? p.y2_c * hw * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(cat ? p.res + (long long)img * p.y2_c * hw : p.x), 0, cat ? p.y2_c * hw * 4 : 0, 0x00020000);
  int vo[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / p.tw, tx = (32 * cb) - ty * p.tw;
    vo[n] = (4 * h * hw + (y0 + ty) * p.w + x0 + tx + j) * 4;
  }
  auto so1 = [&](int tt) {  // scalar
    const int t = t0 + tt;
    return (ch0 + (t & 3) + 8 * (t >> 2)) * hw * 4;
  };
  auto so2 = [&](int tt) { return so1(tt) + p.y2_off * hw * 4; };
  // ---- statistics: the per-channel sums of all waves meet in LDS, wave 0 folds them into the
  // workgroup's per-group sums (fixed order) and hands them on (gn_tail.h: fire-and-forget) ----
  double *cs1 = reinterpret_cast<double *>(smem);         // [CW][NCH][2] per-channel sums of y
  double *cs2 = reinterpret_cast<double *>(smem + 2048);  // ... of y2
  auto to_lds = [&](double *cs, float (&a1)[TN], float (&a2)[TN]) {
    // sum over the 32 lanes that share h (the pixels); lane kHalfSumLane of each half then holds the row sums
#pragma unroll
    for (int tt = 0; tt < TN; ++tt) {
      a1[tt] = half_wave_sum(a1[tt]);
      a2[tt] = half_wave_sum(a2[tt]);
    }
    if (j == kHalfSumLane) {
#pragma unroll
      for (int tt = 0; tt < TN; ++tt) {
        const int t = t0 + tt;
        const int idx = cwi * NCH + 32 * rbi + (t & 3) + 8 * (t >> 2) + 4 * h;
        cs[2 * idx] = (double)a1[tt];
        cs[2 * idx + 1] = (double)a2[tt];
      }
    }
  };
  float u[NR][TN];  // the block tail: conv + res
  if (cat) {
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int tt = 0; tt < TN; ++tt)
        u[n][tt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, vo[n], so2(tt), 0));
  }
  if (st1) {
    float s1[TN], s2[TN];
#pragma unroll
    for (int tt = 0; tt < TN; ++tt) s1[tt] = s2[tt] = 0.0f;
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int tt = 0; tt < TN; ++tt) {
        s1[tt] += v[n][tt];
        s2[tt] = fmaf(v[n][tt], v[n][tt], s2[tt]);
      }
    to_lds(cs1, s1, s2);
  }
  if (cat) {
    float q1[TN], q2[TN];
#pragma unroll
    for (int tt = 0; tt < TN; ++tt) q1[tt] = q2[tt] = 0.0f;
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int tt = 0; tt < TN; ++tt) {
        u[n][tt] += v[n][tt];
        q1[tt] += u[n][tt];
        q2[tt] = fmaf(u[n][tt], u[n][tt], q2[tt]);
      }
    if (st2) to_lds(cs2, q1, q2);
  }
  if (st1 || st2) {
    __syncthreads();
    if (tid < 64) {
      auto fold = [&](const GnOut &f, const double *cs, int c_off) {
        const int cpg = f.c / 32;  // channels per group of the normalised tensor
        const int ng = NCH / cpg;  // groups this workgroup covers
        double a = 0.0, b = 0.0;
        if (tid < ng)
          for (int cw = 0; cw < CW; ++cw)
            for (int ch = 0; ch < cpg; ++ch) {
              const int idx = cw * NCH + tid * cpg + ch;
              a += cs[2 * idx];
              b += cs[2 * idx + 1];
            }
        gn_emit(f, img, (c_off + NCH * (int)blockIdx.y) / cpg, ng, tile, a, b);
      };
      if (st1) fold(p.fin, cs1, 0);
      if (st2) fold(p.fin2, cs2, p.y2_off);
    }
  }
  // ---- bulk stores ----
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int tt = 0; tt < TN; ++tt) {
      const float a = v[n][tt], b = cat ? u[n][tt] : 0.0f;
      if (p.y) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(a), rs_y, vo[n], so1(tt), 0);
      if (cat) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(b), rs_y2, vo[n], so2(tt), 0);
    }
}

template <int RBW, int NR>
__global__ __launch_bounds__(256, kConvWps) void conv3x3_gn_kernel(ConvArgs p) {
  constexpr int CW = 4 / RBW;
  constexpr int kStageIters = stage_iters(32 * NR * CW);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int rbi = wv % RBW, cwi = wv / RBW;

  const int TW = p.tw, TH = p.th, PW = TW + 2;
  const int NPH = (TH + 2) * PW;          // staged pixels (tile + halo)
  const int buf_bytes = NPH * kPixBytes;
  const int tiles_x = p.w / TW, tiles = tiles_x * (p.h / TH);
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
  const int rb = blockIdx.y * RBW + rbi;  // 32-row block of output channels
  const int hw = p.h * p.w;
  const int n_chunks = p.cin / kCK;
  const int kgt = n_chunks * 18;          // K groups (8 deep) in total

  const WStream ws = make_wstream(p.wp, p.wp_floats, lane);

  // ---- staging plan: wave wv stages channels 4 wv .. 4 wv + 3 of every chunk, lane = pixel ----
  int goff[kStageIters];  // offset inside a channel plane, -1 = outside the image / the halo
#pragma unroll
  for (int it = 0; it < kStageIters; ++it) {
    const int lp = lane + 64 * it;
    const int r = lp / PW, c = lp - r * PW;
    int gy = y0 - 1 + r, gx = x0 - 1 + c;
    bool ok = lp < NPH;
    if (p.reflect) {  // -1 -> 1, H -> H - 2 (padding 1 never reaches further)
      gy = gy < 0 ? -gy : (gy >= p.h ? 2 * p.h - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= p.w ? 2 * p.w - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    }
    goff[it] = ok ? gy * p.w + gx : -1;
  }
  const float *xin = p.x + (long long)img * p.cin * hw;
  __shared__ float gn_stats[64];  // (mean, rstd) of the input's 32 groups (csrc/gn_tail.h)
  __shared__ float ss_in[2 * kMaxCin];  // (scale, shift) of every input channel, once per workgroup (filled below)

  f32x4 stg[kStageIters];
  int ch_staged = 0;  // first channel of the chunk in stg (wave-uniform)
  auto stage_load = [&](int chunk) {
    const float *pl = xin + (long long)(chunk * kCK + 4 * wv) * hw;
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int o = goff[it] < 0 ? 0 : goff[it];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        stg[it][k] = pl[(long long)k * hw + o];
    }
    ch_staged = chunk * kCK + 4 * wv;
  };
  auto stage_store = [&](unsigned char *buf) {
    float sc[4], sh[4];  // wave-uniform
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sc[k] = ss_in[2 * (ch_staged + k)];
      sh[k] = ss_in[2 * (ch_staged + k) + 1];
    }
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int lp = lane + 64 * it;
      if (lp < NPH) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = fmaf(stg[it][k], sc[k], sh[k]);
          if (p.relu) t = fmaxf(t, 0.0f);
          v[k] = goff[it] < 0 ? 0.0f : t;
        }
        *reinterpret_cast<f32x4 *>(buf + lp * kPixBytes + ((wv ^ ((lp >> 2) & 3)) << 4)) = v;
      }
    }
  };

  // ---- this wave's output pixels: column block n = 32 consecutive pixels of one tile row ----
  // byte offset (inside a stage buffer) of this lane's B operand for each of the 9 taps and column
  // blocks, K group 0 (slot h ^ swizzle); group 1 is the same address XOR 32 (slot ^ 2).  Computed
  // once: the inner loop spends 2 VALU instructions per operand read instead of ~8.
  int boff[NR][9];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / TW, tx = (32 * cb) - ty * TW;
    const int lpc = (ty + 1) * PW + tx + j + 1;  // linear halo-pixel index of this lane's pixel
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int lp = lpc + (tap / 3 - 1) * PW + (tap % 3 - 1);
      boff[n][tap] = lp * kPixBytes + ((h ^ ((lp >> 2) & 3)) << 4);
    }
  }

  f32x16 acc[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[n][t] = 0.0f;

  // A ring: 6 fragments = the 3 taps x 2 groups of one kernel row; slot k is refilled with the
  // same slot of the NEXT row-step right after its MFMAs were issued (prefetch distance 6 groups)
  const int a_base = rb * kgt * 64;
  f32x4 ring[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ring[k] = wload128(ws, a_base + min(k, kgt - 1) * 64);

  stage_load(0);
  GnAffine affine;
  gn_affine_load(p.gn, p.cin, affine);
  gn_load_stats(p.gn, img, gn_stats);
  __syncthreads();
  // (scale, shift) per input channel from the statistics + gamma / beta: by one thread per channel, once -- the
  // staging of every chunk took them again per wave (LDS reads, two loads from L2 and four VALU instructions per
  // channel, with the loads' round trip in front of the chunk's barrier)
  gn_table_fill(p.gn, img, p.cin, gn_stats, affine, ss_in);
  __syncthreads();
  stage_store(smem);
  __syncthreads();

  int kg0 = 0;  // first K group of the current row-step
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char *buf = smem + (chunk & 1) * buf_bytes;
    const bool more = chunk + 1 < n_chunks;
    // the three kernel rows are unrolled so that every s_waitcnt vmcnt is counted for its own
    // position: the next chunk's activations are requested at the END of row 0 -- the weight
    // fragments row 1 waits for are older, and by row 2 they have had a whole row-step to land
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      // B operand of the first group of this row-step
      f32x4 bcur[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) bcur[n] = *reinterpret_cast<const f32x4 *>(buf + boff[n][3 * ky]);
#pragma unroll
      for (int s = 0; s < 6; ++s) {  // s = 2 * kx + g
        // next group's B operand (the first group of the next row-step is read at its top)
        f32x4 bnxt[NR];
        if (s < 5) {
          const int sn = s + 1;
#pragma unroll
          for (int n = 0; n < NR; ++n)
            bnxt[n] = *reinterpret_cast<const f32x4 *>(buf + (boff[n][3 * ky + (sn >> 1)] ^ (32 * (sn & 1))));
        }
        const f32x4 a = ring[s];
        ring[s] = wload128(ws, a_base + min(kg0 + 6 + s, kgt - 1) * 64);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetches above the MFMAs (see query.hip)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < NR; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bcur[n][i], acc[n], 0, 0, 0);
        if (s < 5) {
#pragma unroll
          for (int n = 0; n < NR; ++n) bcur[n] = bnxt[n];
        }
      }
      kg0 += 6;
      if (ky == 0 && more) stage_load(chunk + 1);
    }
    if (more) stage_store(smem + ((chunk + 1) & 1) * buf_bytes);
    __syncthreads();
  }

  // ---- epilogue: stores + GroupNorm statistics (conv_epilogue) ----
  float v[NR][16];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) v[n][t] = acc[n][t];
  conv_epilogue<32 * RBW, CW, NR, 16>(p, v, 0, img, tile, tiles, y0, x0, rbi, cwi, lane, smem);
}

// ---- small-launch variant: the four waves of a workgroup split K ---------------------------------
// The kernel above gives a wave a whole K = 9 Cin loop, so a launch takes at least one such loop
// (31 us for 256 -> 128 channels) however small the map is, and a 64^2 / 32^2 map or a single frame
// yields fewer workgroups than the chip has slots (profiles/r02x_encoder_kernel_stats_b1.txt: the
// 32^2 convolutions took as long as the 128^2 ones).  Here a workgroup is ONE 32-channel row block
// x 32 NR pixels and wave ks takes the 16-channel chunks ks, ks + 4, ... of the input -- staged by
// itself into its own double-buffered LDS tile, so there is no workgroup barrier inside the K loop
// -- and the four partial accumulators meet in LDS at the end, added in wave order (deterministic).
// 4 x (Cout / 32) / RBW times as many workgroups, each a quarter as long; the price is that every
// wave stages all 16 channels of its chunks (the halo is staged once per row block instead of once
// per workgroup), which the launcher only pays when the large-tile launch would not fill the chip.
template <int NR>
__global__ __launch_bounds__(256, 2) void conv3x3_gn_sk_kernel(ConvArgs p) {
  constexpr int kStageIters = stage_iters(32 * NR);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // = K split index
  const int j = lane & 31, h = lane >> 5;

  const int TW = p.tw, TH = p.th, PW = TW + 2;
  const int NPH = (TH + 2) * PW;
  const int buf_bytes = NPH * kPixBytes;
  const int tiles_x = p.w / TW, tiles = tiles_x * (p.h / TH);
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
  const int rb = blockIdx.y;
  const int hw = p.h * p.w;
  const int n_chunks = p.cin / kCK;
  const int kgt = n_chunks * 18;
  unsigned char *mybuf = smem + wv * 2 * buf_bytes;

  const WStream ws = make_wstream(p.wp, p.wp_floats, lane);

  int goff[kStageIters];
#pragma unroll
  for (int it = 0; it < kStageIters; ++it) {
    const int lp = lane + 64 * it;
    const int r = lp / PW, c = lp - r * PW;
    int gy = y0 - 1 + r, gx = x0 - 1 + c;
    bool ok = lp < NPH;
    if (p.reflect) {
      gy = gy < 0 ? -gy : (gy >= p.h ? 2 * p.h - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= p.w ? 2 * p.w - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    }
    goff[it] = ok ? gy * p.w + gx : -1;
  }
  const float *xin = p.x + (long long)img * p.cin * hw;
  __shared__ float gn_stats[64];  // (mean, rstd) of the input's 32 groups (csrc/gn_tail.h)
  __shared__ float ss_in[2 * kMaxCin];  // (scale, shift) of every input channel, once per workgroup (filled below)

  // lane = pixel, all 16 channels of the chunk
  f32x4 stg[kStageIters][4];
  int ch_staged = 0;
  // activations through a buffer resource: lane part = the pixel's offset inside a plane (one VGPR
  // per pass), plane = wave-uniform scalar offset -- 64-bit flat addresses made hipcc hoist 16 plane
  // pointers per pass out of the loop and spill (the weight stream's story, query_common.h)
  const __amdgpu_buffer_rsrc_t xs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xin), 0, p.cin * hw * 4, 0x00020000);
  auto stage_load = [&](int chunk) {
    const int plane0 = chunk * kCK * hw * 4;  // bytes
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int o = (goff[it] < 0 ? 0 : goff[it]) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          stg[it][q][k] = __builtin_bit_cast(
              float, __builtin_amdgcn_raw_buffer_load_b32(xs, o, plane0 + (4 * q + k) * hw * 4, 0));
    }
    ch_staged = chunk * kCK;
  };
  auto stage_store = [&](unsigned char *buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sc[4], sh[4];  // wave-uniform
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sc[k] = ss_in[2 * (ch_staged + 4 * q + k)];
        sh[k] = ss_in[2 * (ch_staged + 4 * q + k) + 1];
      }
#pragma unroll
      for (int it = 0; it < kStageIters; ++it) {
        const int lp = lane + 64 * it;
        if (lp < NPH) {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = fmaf(stg[it][q][k], sc[k], sh[k]);
            if (p.relu) t = fmaxf(t, 0.0f);
            v[k] = goff[it] < 0 ? 0.0f : t;
          }
          *reinterpret_cast<f32x4 *>(buf + lp * kPixBytes + ((q ^ ((lp >> 2) & 3)) << 4)) = v;
        }
      }
    }
  };

  int boff[NR][9];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int ty = (32 * n) / TW, tx = (32 * n) - ty * TW;
    const int lpc = (ty + 1) * PW + tx + j + 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int lp = lpc + (tap / 3 - 1) * PW + (tap % 3 - 1);
      boff[n][tap] = lp * kPixBytes + ((h ^ ((lp >> 2) & 3)) << 4);
    }
  }

  f32x16 acc[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[n][t] = 0.0f;

  if (wv < n_chunks) stage_load(wv);  // in flight while the input's GroupNorm statistics are read
  GnAffine affine;
  gn_affine_load(p.gn, p.cin, affine);
  gn_load_stats(p.gn, img, gn_stats);
  __syncthreads();
  // (scale, shift) per input channel from the statistics + gamma / beta: by one thread per channel, once -- the
  // staging of every chunk took them again per wave (LDS reads, two loads from L2 and four VALU instructions per
  // channel, with the loads' round trip in front of the chunk's barrier)
  gn_table_fill(p.gn, img, p.cin, gn_stats, affine, ss_in);
  __syncthreads();
  if (wv < n_chunks) {
    const int a_base = rb * kgt * 64;
    f32x4 ring[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) ring[k] = wload128(ws, a_base + (wv * 18 + k) * 64);
    stage_store(mybuf);
    int it_n = 0;
    for (int chunk = wv; chunk < n_chunks; chunk += 4, ++it_n) {
      const unsigned char *buf = mybuf + (it_n & 1) * buf_bytes;
      const bool more = chunk + 4 < n_chunks;
      __builtin_amdgcn_wave_barrier();  // this wave's LDS tile was written by its own lanes
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // first K group of the NEXT row-step (the next chunk of this wave after the last row)
        const int kg_next = ky < 2 ? chunk * 18 + 6 * (ky + 1) : (chunk + 4) * 18;
        f32x4 bcur[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) bcur[n] = *reinterpret_cast<const f32x4 *>(buf + boff[n][3 * ky]);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          f32x4 bnxt[NR];
          if (s < 5) {
            const int sn = s + 1;
#pragma unroll
            for (int n = 0; n < NR; ++n)
              bnxt[n] = *reinterpret_cast<const f32x4 *>(buf + (boff[n][3 * ky + (sn >> 1)] ^ (32 * (sn & 1))));
          }
          const f32x4 a = ring[s];
          ring[s] = wload128(ws, a_base + min(kg_next + s, kgt - 1) * 64);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < NR; ++n)
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bcur[n][i], acc[n], 0, 0, 0);
          if (s < 5) {
#pragma unroll
            for (int n = 0; n < NR; ++n) bcur[n] = bnxt[n];
          }
        }
        if (ky == 0 && more) stage_load(chunk + 4);
      }
      if (more) stage_store(mybuf + ((it_n + 1) & 1) * buf_bytes);
    }
  }

  // ---- the four partial accumulators meet in LDS; wave wv finishes rows t = 4 wv .. 4 wv + 3 ----
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);  // [4][NR][16][64]
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) red[((wv * NR + n) * 16 + t) * 64 + lane] = acc[n][t];
  __syncthreads();
  float v[NR][4];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = 4 * wv + tt;
      float sum = red[((0 * NR + n) * 16 + t) * 64 + lane];
#pragma unroll
      for (int k = 1; k < 4; ++k) sum += red[((k * NR + n) * 16 + t) * 64 + lane];
      v[n][tt] = sum;
    }
  __syncthreads();
  conv_epilogue<32, 1, NR, 4>(p, v, 4 * wv, img, tile, tiles, y0, x0, 0, 0, lane, smem);
}

// ---- split-f16 ("f16x3") variant ----------------------------------------------------------------
// Same decomposition, operands carried as two f16 halves (v = hi + lo, 22 significant bits) and
// every product expanded as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with f32 accumulation
// -- the arithmetic of pifu_query16_kernel (query16.hip), f32-class accuracy at 5.3x fewer matrix
// cycles (3 MFMAs of 32 cycles per 16-deep k-step instead of 8 of 64).  One k16 step = one tap of
// one 16-channel chunk.  Weights are pre-split and pre-scaled by a power of two S derived from
// max|W| (so that the lo halves stay out of the f16 subnormals); the accumulators are multiplied by
// 1/S (exact) in the epilogue.  The staged pixel keeps its 64 bytes: [hi ch 0-7 | hi ch 8-15 |
// lo ch 0-7 | lo ch 8-15].
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float conv16_scale(float wmax) {
  // largest power of two S with max|w| * S <= 2^14 (f16 tops out at 65504), clamped to 2^+-14
  int e = 0;
  if (wmax > 0.0f && wmax < 3.0e38f) {
    (void)frexpf(wmax, &e);
    e = 14 - e;
  }
  e = e > 14 ? 14 : (e < -14 ? -14 : e);
  return ldexpf(1.0f, e);
}

// W [Cout][Cin][3][3] -> [rb][ks = chunk * 9 + tap][hi | lo][lane] of h8: lane (r, hh) holds
// W[32 rb + r][16 chunk + 8 hh + e][tap] * S, e = 0..7.
__global__ void conv3x3_pack16_kernel(const float *__restrict__ w, int cout, int cin,
                                      const float *__restrict__ wmax, _Float16 *__restrict__ wp) {
  const float S = conv16_scale(*wmax);
  const long long total = (long long)cout * cin * 9;  // (hi, lo) pairs
  const int kst = (cin / kCK) * 9;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(t & 7), lane = (int)((t >> 3) & 63);
    const long long q = t >> 9;
    const int ks = (int)(q % kst), rb = (int)(q / kst);
    const int tap = ks % 9, chunk = ks / 9;
    const int co = 32 * rb + (lane & 31);
    const int ci = kCK * chunk + 8 * (lane >> 5) + e;
    const float v = w[((long long)co * cin + ci) * 9 + tap] * S;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const long long base = ((q * 2) * 64 + lane) * 8 + e;
    wp[base] = hi;
    wp[base + 64 * 8] = lo;
  }
}

__device__ __forceinline__ h8 hload16(const WStream &w, int idx16) {
  return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.lane16, idx16 * 16, 0));
}

template <int RBW, int NR>
__global__ __launch_bounds__(256, kConvWps) void conv3x3_gn16_kernel(ConvArgs p, const float *__restrict__ wmax) {
  constexpr int CW = 4 / RBW;
  constexpr int kStageIters = stage_iters(32 * NR * CW);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int rbi = wv % RBW, cwi = wv / RBW;

  const int TW = p.tw, TH = p.th, PW = TW + 2;
  const int NPH = (TH + 2) * PW;
  const int buf_bytes = NPH * kPixBytes;
  const int tiles_x = p.w / TW, tiles = tiles_x * (p.h / TH);
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int y0 = (tile / tiles_x) * TH, x0 = (tile % tiles_x) * TW;
  const int rb = blockIdx.y * RBW + rbi;
  const int hw = p.h * p.w;
  const int n_chunks = p.cin / kCK;
  const int kst = n_chunks * 9;  // k16 steps in total
  const float inv_scale = 1.0f / conv16_scale(*wmax);

  const WStream ws = make_wstream(p.wp, p.wp_floats, lane);

  int goff[kStageIters];
#pragma unroll
  for (int it = 0; it < kStageIters; ++it) {
    const int lp = lane + 64 * it;
    const int r = lp / PW, c = lp - r * PW;
    int gy = y0 - 1 + r, gx = x0 - 1 + c;
    bool ok = lp < NPH;
    if (p.reflect) {  // -1 -> 1, H -> H - 2 (padding 1 never reaches further)
      gy = gy < 0 ? -gy : (gy >= p.h ? 2 * p.h - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= p.w ? 2 * p.w - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    }
    goff[it] = ok ? gy * p.w + gx : -1;
  }
  const float *xin = p.x + (long long)img * p.cin * hw;
  __shared__ float gn_stats[64];  // (mean, rstd) of the input's 32 groups (csrc/gn_tail.h)
  __shared__ float ss_in[2 * kMaxCin];  // (scale, shift) of every input channel, once per workgroup (filled below)

  f32x4 stg[kStageIters];
  int ch_staged = 0;  // first channel of the chunk in stg (wave-uniform)
  auto stage_load = [&](int chunk) {
    const float *pl = xin + (long long)(chunk * kCK + 4 * wv) * hw;
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int o = goff[it] < 0 ? 0 : goff[it];
#pragma unroll
      for (int k = 0; k < 4; ++k) stg[it][k] = pl[(long long)k * hw + o];
    }
    ch_staged = chunk * kCK + 4 * wv;
  };
  // channels 4 wv .. 4 wv + 3 of the chunk = 8 bytes at offset 8 (wv & 1) of hi slot (wv >> 1) and
  // of lo slot 2 + (wv >> 1)
  auto stage_store = [&](unsigned char *buf) {
    float sc[4], sh[4];  // wave-uniform
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sc[k] = ss_in[2 * (ch_staged + k)];
      sh[k] = ss_in[2 * (ch_staged + k) + 1];
    }
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int lp = lane + 64 * it;
      if (lp < NPH) {
        h4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = fmaf(stg[it][k], sc[k], sh[k]);
          if (p.relu) t = fmaxf(t, 0.0f);
          t = goff[it] < 0 ? 0.0f : t;
          hi[k] = (_Float16)t;
          lo[k] = (_Float16)(t - (float)hi[k]);
        }
        unsigned char *px = buf + lp * kPixBytes + 8 * (wv & 1);
        const int sw = (lp >> 2) & 3;
        *reinterpret_cast<h4 *>(px + (((wv >> 1) ^ sw) << 4)) = hi;
        *reinterpret_cast<h4 *>(px + (((2 + (wv >> 1)) ^ sw) << 4)) = lo;
      }
    }
  };

  // byte offsets of this lane's hi B operand per tap and column block; the lo operand is the same
  // address XOR 32 (slot ^ 2)
  int boff[NR][9];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / TW, tx = (32 * cb) - ty * TW;
    const int lpc = (ty + 1) * PW + tx + j + 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int lp = lpc + (tap / 3 - 1) * PW + (tap % 3 - 1);
      boff[n][tap] = lp * kPixBytes + ((h ^ ((lp >> 2) & 3)) << 4);
    }
  }

  f32x16 acc[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[n][t] = 0.0f;

  // A ring: (hi, lo) fragments of the 3 taps of one kernel row, refilled one row-step ahead
  const int a_base = rb * kst * 128;  // 16-byte units: [ks][hi | lo][64 lanes]
  h8 ring[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ring[k] = hload16(ws, a_base + min(k >> 1, kst - 1) * 128 + (k & 1) * 64);

  stage_load(0);
  GnAffine affine;
  gn_affine_load(p.gn, p.cin, affine);
  gn_load_stats(p.gn, img, gn_stats);
  __syncthreads();
  // (scale, shift) per input channel from the statistics + gamma / beta: by one thread per channel, once -- the
  // staging of every chunk took them again per wave (LDS reads, two loads from L2 and four VALU instructions per
  // channel, with the loads' round trip in front of the chunk's barrier)
  gn_table_fill(p.gn, img, p.cin, gn_stats, affine, ss_in);
  __syncthreads();
  stage_store(smem);
  __syncthreads();

  int ks0 = 0;  // first k16 step of the current row-step
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char *buf = smem + (chunk & 1) * buf_bytes;
    const bool more = chunk + 1 < n_chunks;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      h8 bh[NR], bl[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        bh[n] = *reinterpret_cast<const h8 *>(buf + boff[n][3 * ky]);
        bl[n] = *reinterpret_cast<const h8 *>(buf + (boff[n][3 * ky] ^ 32));
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        h8 nh[NR], nl[NR];
        if (kx < 2) {
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            nh[n] = *reinterpret_cast<const h8 *>(buf + boff[n][3 * ky + kx + 1]);
            nl[n] = *reinterpret_cast<const h8 *>(buf + (boff[n][3 * ky + kx + 1] ^ 32));
          }
        }
        const h8 ah = ring[2 * kx], al = ring[2 * kx + 1];
        const int nxt = a_base + min(ks0 + 3 + kx, kst - 1) * 128;
        ring[2 * kx] = hload16(ws, nxt);
        ring[2 * kx + 1] = hload16(ws, nxt + 64);
        __builtin_amdgcn_sched_barrier(0);
        // term-major order: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int n = 0; n < NR; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[n], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NR; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[n], acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NR; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[n], acc[n], 0, 0, 0);
        if (kx < 2) {
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            bh[n] = nh[n];
            bl[n] = nl[n];
          }
        }
      }
      ks0 += 3;
      if (ky == 0 && more) stage_load(chunk + 1);
    }
    if (more) stage_store(smem + ((chunk + 1) & 1) * buf_bytes);
    __syncthreads();
  }

  float v[NR][16];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) v[n][t] = acc[n][t] * inv_scale;
  conv_epilogue<32 * RBW, CW, NR, 16>(p, v, 0, img, tile, tiles, y0, x0, rbi, cwi, lane, smem);
}

// (scale, shift) of GroupNorm(groups, C) from partial sums: ss[n][c] = (gamma[c] rstd,
// beta[c] - mean gamma[c] rstd).  One wave per (image, group); partial [(n*groups + g)*S + s][2].
__global__ __launch_bounds__(64) void gn_finalize_kernel(const double *__restrict__ partial, int groups,
                                                         int slices, double count, int cpg,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps,
                                                         float *__restrict__ ss) {
  const int g = blockIdx.x % groups, n = blockIdx.x / groups;
  const double *pp = partial + (long long)blockIdx.x * slices * 2;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < slices; i += 64) {
    a += pp[2 * i];
    b += pp[2 * i + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  const double mean_d = a / count;
  const double var_d = fmax(b / count - mean_d * mean_d, 0.0);
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var_d + (double)eps));
  if ((int)threadIdx.x < cpg) {
    const int c = g * cpg + threadIdx.x;
    const float sc = rstd * gamma[c];
    float *o = ss + ((long long)n * groups * cpg + c) * 2;
    o[0] = sc;
    o[1] = beta[c] - mean * sc;
  }
}

// y = res + (t * scale[n,c] + shift[n,c]): the tail of a residual block whose last GroupNorm has no
// ReLU (ResBlkFilters.py:75-84: out = x + conv_block(x)); hw % 4 == 0.
__global__ __launch_bounds__(256) void scale_shift_add_kernel(const float *__restrict__ t,
                                                              const float *__restrict__ ss,
                                                              const float *__restrict__ res,
                                                              long long hw4, long long total4,
                                                              float *__restrict__ y) {
  const f32x4 *t4 = reinterpret_cast<const f32x4 *>(t), *r4 = reinterpret_cast<const f32x4 *>(res);
  f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long plane = i / hw4;  // image * C + channel
    const float sc = ss[2 * plane], sh = ss[2 * plane + 1];
    const f32x4 a = t4[i], r = r4[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = r[k] + (a[k] * sc + sh);
    y4[i] = o;
  }
}

int launch_scale_shift_add(mp_ctx *ctx, const float *t, const float *ss, const float *res, long long planes,
                           long long hw, float *y, hipStream_t st) {
  const long long total4 = planes * (hw / 4);
  long long blocks = (total4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(scale_shift_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t, ss, res, hw / 4,
                     total4, y);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_conv3x3_pack(mp_ctx *ctx, const float *w, int cout, int cin, float *wp, hipStream_t st) {
  const long long total = (long long)cout * cin * 9;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cout, cin, wp);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// Tile shape of the kernel instantiation that serves a launch.  Large launches: RBW waves split the
// output channels (32 each), the other 4 / RBW split the pixels, NR = 32-pixel column blocks per
// wave (smaller NR = more, smaller workgroups: better balance over 256 CUs x 2 slots, more weight
// re-streaming).  Launches that would leave workgroup slots empty take the split-K kernel (sk):
// one 32-channel row block x 32 NR pixels per workgroup, K split over its four waves.
static int g_conv_nr = 0;  // 0 = heuristic; tools/conv_probe.py overrides it for A/B runs
static int g_conv_sk = -1; // -1 = heuristic; 0 / 1 force the large-tile / the split-K kernel

static int g_conv_wino = 1;  // 0: never take the Winograd kernel (mp_conv3x3_tune(0x400 or a forced variant); A/B runs)
static long long g_conv_wino_min_wgs = 128;  // launches with fewer Winograd workgroups stay on the direct kernels (one 88 us workgroup per CU: tools/wino_check.py)

void conv3x3_set_nr(int nr) {
  conv3x3_wino_set_variant((nr & 0x800) ? 64 : (nr & 0x1000) ? 128 : 0);  // | 0x800 / | 0x1000: the 64- / 128-channel Winograd kernel
  nr &= ~0x1800;
  g_conv_nr = nr & 0xff;
  g_conv_wino = (nr >> 8) == 0 ? 1 : 0;  // any forced variant (| 0x100, | 0x200) or 0x400 (direct kernels, heuristic) turns Winograd off
  nr &= ~0x400;
  g_conv_sk = (nr >> 8) == 0 ? -1 : (nr >> 8) - 1;  // mp_conv3x3_tune(nr | 0x100: large tiles, | 0x200: split-K)
}

struct ConvPlan {
  int rbw, nr, tw, th, tiles;  // tiles per image
  bool sk;
};

static ConvPlan conv_plan(int cout, int n, int h, int w, bool f16) {
  ConvPlan c;
  c.rbw = cout % 128 == 0 ? 4 : cout % 64 == 0 ? 2 : 1;
  const int cw = 4 / c.rbw;
  const long long pix = (long long)n * h * w;
  // NR (32-pixel column blocks per wave).  Every workgroup of these launches is resident at once, so
  // a launch lasts as long as the busiest CU: ceil(workgroups / 256) x NR units of MFMA work.  NR = 2
  // halves the workgroups and the weight bytes streamed per FLOP and gives a wave two independent
  // accumulator chains; it wins every tie (measured, profiles/r03d_conv_bench.txt: 256 -> 128 at
  // 128^2, batch 1: 118 us as 256 workgroups of NR = 2, 130 us as 512 of NR = 1) and loses when the
  // halved launch quantises badly (128 -> 64 at 64^2 x 10: 320 workgroups of NR = 2 = 2 x 2 units,
  // 640 of NR = 1 = 3).  NR = 4 needs all 256 VGPRs (spills) and measured slower at every shape, so
  // it is built for the split-f16 kernels only.
  const long long wg1 = pix / (32 * cw) * (cout / (32 * c.rbw));  // workgroups at NR = 1
  const long long wg2 = wg1 / 2;
  c.nr = ((wg2 + 255) / 256) * 2 <= (wg1 + 255) / 256 ? 2 : 1;
  if (g_conv_nr > 0) c.nr = g_conv_nr;
  if (c.nr > 2 && (c.rbw == 1 || !f16)) c.nr = 2;
  // split-K when even the smallest large-tile launch does not give every slot a workgroup
  c.sk = !f16 && (g_conv_sk >= 0 ? g_conv_sk == 1 : wg1 < 512);
  if (c.sk) {
    c.rbw = 1;
    c.nr = g_conv_nr > 0 ? (g_conv_nr > 2 ? 2 : g_conv_nr) : (pix / 64 * (cout / 32) >= 512 ? 2 : 1);
    if (h < c.nr) c.nr = 1;
    c.tw = kTileW;
    c.th = c.nr;
  } else {
    // 32-pixel-wide tiles, PX / 32 rows tall: the staged halo is (TH + 2) x 34 pixels (1.3-2.1x the
    // tile; one-row tiles would stage 3x) and a wave stages it in 2-6 passes of 64 pixels
    c.tw = kTileW;
    c.th = 32 * c.nr * cw / c.tw;
  }
  c.tiles = (h / c.th) * (w / c.tw);
  return c;
}

// slots per (image, group) of the statistics a launch publishes = its tiles per image
int conv3x3_stat_slices(int cout, int n, int h, int w, bool f16) { return conv_plan(cout, n, h, w, f16).tiles; }

bool conv3x3_supported(int cin, int cout, int h, int w) {
  return !(cin % kCK || cout % 32 || cin < kCK || cout < 32 || w < 32 || (w & (w - 1)) || h < 8 ||
           (h & (h - 1)));
}

int launch_absmax(mp_ctx *ctx, const float *src, long long n, unsigned int *out_bits, hipStream_t st);

int launch_conv3x3_pack16(mp_ctx *ctx, const float *w, int cout, int cin, void *wp, float *wmax,
                          hipStream_t st) {
  const long long total = (long long)cout * cin * 9;
  int rc = launch_absmax(ctx, w, total, reinterpret_cast<unsigned int *>(wmax), st);
  if (rc != MP_OK) return rc;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_pack16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cout, cin, wmax,
                     static_cast<_Float16 *>(wp));
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

static int raise_lds_limit(mp_ctx *ctx, const void *kern_id, int bytes) {
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->lds_attr_done.insert(kern_id);
  }
  return MP_OK;
}

template <int RBW, int NR>
static int launch_conv16_t(mp_ctx *ctx, const ConvArgs &a, const float *wmax, int tiles, hipStream_t st) {
  const int lds = 2 * (a.th + 2) * (a.tw + 2) * kPixBytes;
  auto kern = conv3x3_gn16_kernel<RBW, NR>;
  int rc = raise_lds_limit(ctx, reinterpret_cast<const void *>(kern), 2 * halo_pixels(32 * NR * (4 / RBW)) * kPixBytes);
  if (rc != MP_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / (32 * RBW))), dim3(256),
                     lds, st, a, wmax);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

template <int RBW, int NR>
static int launch_conv_t(mp_ctx *ctx, const ConvArgs &a, int tiles, hipStream_t st) {
  const int lds = 2 * (a.th + 2) * (a.tw + 2) * kPixBytes;
  auto kern = conv3x3_gn_kernel<RBW, NR>;
  int rc = raise_lds_limit(ctx, reinterpret_cast<const void *>(kern), 2 * halo_pixels(32 * NR * (4 / RBW)) * kPixBytes);
  if (rc != MP_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / (32 * RBW))), dim3(256),
                     lds, st, a);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

template <int NR>
static int launch_conv_sk_t(mp_ctx *ctx, const ConvArgs &a, int tiles, hipStream_t st) {
  const int lds = 8 * halo_pixels(32 * NR) * kPixBytes;  // 4 waves x 2 buffers
  auto kern = conv3x3_gn_sk_kernel<NR>;
  int rc = raise_lds_limit(ctx, reinterpret_cast<const void *>(kern), lds);
  if (rc != MP_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / 32)), dim3(256), lds, st, a);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// Fills in what the launcher derives (tile shape, slots, group sizes) and checks the statistics
// request against the launch: `a` arrives with x / gn / wp / y / y2 / res / fin / fin2 as the caller
// gave them (fin.c, fin.S, gn.c, gn.count unset).  partial_cap[k]: doubles the caller allocated for
// fin / fin2 (-1 = unchecked legacy entry points).
int launch_conv3x3(mp_ctx *ctx, ConvArgs a, const float *wmax16, const long long partial_cap[2], hipStream_t st) {
  const int n = a.n_img, cin = a.cin, cout = a.cout, h = a.h, w = a.w;
  if (!conv3x3_supported(cin, cout, h, w))
    return fail(ctx, MP_ERR_UNSUPPORTED,
                "conv3x3: needs Cin %% 16 == 0, Cout %% 32 == 0, H and W powers of two (W >= 32, H >= 8); got %d -> %d at %dx%d",
                cin, cout, h, w);
  if (cin > kMaxCin)
    return fail(ctx, MP_ERR_UNSUPPORTED, "conv3x3: at most %d input channels (got %d)", kMaxCin, cin);
  a.wp_floats = cout * cin * 9;
  ConvPlan c = conv_plan(cout, n, h, w, wmax16 != nullptr);
  // Winograd F(2x2, 3x3) (conv_wino.hip) when the caller packed the weights for it: exact-f32 products,
  // statistics by hand-over only (the legacy partial buffers are sized from conv_plan's tiles)
  const bool wino = a.wpw && !wmax16 && !a.fin.partial && !a.fin2.partial && g_conv_wino &&
                    conv3x3_wino_supported(cin, cout, h, w) &&
                    (conv3x3_wino_forced() || conv3x3_wino_workgroups(a) >= g_conv_wino_min_wgs);
  if (wino) {
    a.wpw_floats = 16 * cout * cin;
    c.tiles = conv3x3_wino_tiles(h, w);
    c.th = 8;
    c.tw = 16;
  }
  a.tw = c.tw;
  a.th = c.th;
  if (a.th > h)
    return fail(ctx, MP_ERR_UNSUPPORTED, "conv3x3: %dx%d map too small for a %dx%d tile", h, w, a.th, a.tw);
  if (a.y2) {
    if (!a.res || a.y2_c % 32 || a.y2_off < 0 || a.y2_off + cout > a.y2_c || a.y2_off % (a.y2_c / 32) ||
        32 % (a.y2_c / 32))
      return fail(ctx, MP_ERR_ARG, "conv3x3: bad fused-tail request (%d channels at offset %d of %d)", cout,
                  a.y2_off, a.y2_c);
  } else if (!a.y) {
    return fail(ctx, MP_ERR_ARG, "conv3x3: no output buffer");
  }
  for (int k = 0; k < 2; ++k) {
    GnOut &f = k ? a.fin2 : a.fin;
    if (!gn_wanted(f)) continue;
    if (k && !a.y2) return fail(ctx, MP_ERR_ARG, "conv3x3: statistics of y2 requested without y2");
    f.c = k ? a.y2_c : cout;
    f.S = c.tiles;
    f.n = n;
    if (f.partial && partial_cap[k] >= 0 && partial_cap[k] < (long long)n * 32 * f.S * 2)
      return fail(ctx, MP_ERR_ARG, "conv3x3: statistics buffer holds %lld doubles, the launch writes %lld",
                  partial_cap[k], (long long)n * 32 * f.S * 2);
  }
  if (gn_active(a.gn)) {
    if (cin % 32) return fail(ctx, MP_ERR_ARG, "conv3x3: a GroupNorm(32, Cin) input needs Cin %% 32 == 0");
    if (a.gn.acc && (!a.gn.gamma || !a.gn.beta))
      return fail(ctx, MP_ERR_ARG, "conv3x3: GroupNorm hand-over without gamma / beta");
    a.gn.c = cin;
    a.gn.n = n;
    a.gn.count = (double)(cin / 32) * h * w;
  }
  if (wino) return launch_conv3x3_wino(ctx, a, st);
  if (c.sk) return c.nr == 2 ? launch_conv_sk_t<2>(ctx, a, c.tiles, st) : launch_conv_sk_t<1>(ctx, a, c.tiles, st);
  const int rbw = c.rbw, nr = c.nr, tiles = c.tiles;
#define MP_CONV_CASE(R, N)                                                                   \
  if (rbw == R && nr == N)                                                                   \
    return wmax16 ? launch_conv16_t<R, N>(ctx, a, wmax16, tiles, st) : launch_conv_t<R, N>(ctx, a, tiles, st);
  if (wmax16 && nr == 4) {
    if (rbw == 4) return launch_conv16_t<4, 4>(ctx, a, wmax16, tiles, st);
    if (rbw == 2) return launch_conv16_t<2, 4>(ctx, a, wmax16, tiles, st);
  }
  MP_CONV_CASE(4, 2)
  MP_CONV_CASE(4, 1)
  MP_CONV_CASE(2, 2)
  MP_CONV_CASE(2, 1)
  MP_CONV_CASE(1, 2)
  MP_CONV_CASE(1, 1)
#undef MP_CONV_CASE
  return fail(ctx, MP_ERR_UNSUPPORTED, "conv3x3: no instantiation for rbw %d nr %d", rbw, nr);
}

// legacy form: plain output + optional partial sums (finalised by mp_gn_finalize)
int launch_conv3x3_gn(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss,
                      int relu, int reflect, const float *wp, const float *wmax16, int cout, float *y,
                      double *stats, hipStream_t st) {
  ConvArgs a;
  a.x = x;
  a.gn = gn_in_none();
  a.gn.ss = ss;
  a.wp = wp;
  a.y = y;
  a.y2 = nullptr;
  a.res = nullptr;
  a.y2_c = 32;
  a.y2_off = 0;
  a.fin = gn_out_none();
  a.fin2 = gn_out_none();
  a.fin.partial = stats;
  a.n_img = n;
  a.cin = cin;
  a.cout = cout;
  a.h = h;
  a.w = w;
  a.relu = relu;
  a.reflect = reflect;
  const long long cap[2] = {-1, -1};
  return launch_conv3x3(ctx, a, wmax16, cap, st);
}

int launch_gn_finalize(mp_ctx *ctx, const double *partial, int n, int c, int groups, int slices,
                       double count, const float *gamma, const float *beta, float eps, float *ss,
                       hipStream_t st) {
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)(n * groups)), dim3(64), 0, st, partial, groups,
                     slices, count, c / groups, gamma, beta, eps, ss);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- 1x1 convolutions of the hourglass tail (HGFilters.py:184-204) ------------------------------
//   conv_last (+ bias) -> bn_end -> ReLU -> l (+ bias) = the stack's feature map
//   x_next = x + bl(relu(bn_end(.))) + al(l(.))                      (stacks 0..2)
// as GEMMs on the same MFMA scheme: M = 256 output channels (each wave two 32-row blocks), N = 64
// consecutive pixels of the flattened image (a 1x1 convolution has no halo), K = one or TWO input
// tensors (the second K segment is what turns bl(.) + al(.) into one GEMM, like the skip-concat of
// the MLP).  GroupNorm + ReLU of segment 1 are applied while staging; the epilogue adds the bias
// and an optional residual, writes NCHW and / or the channels-last [H*W, C] map the query kernels
// read (through LDS, so that every pixel row leaves as one 1 KB burst), and can emit the
// GroupNorm statistics of its output (bn_end).  F16 = split-f16 ("f16x3") operands as in
// conv3x3_gn16_kernel.
constexpr int kC1K = 64;          // channels per staged chunk
constexpr int kC1Px = 64;         // pixels per workgroup
constexpr int kC1Row = kC1K * 4;  // bytes per staged pixel (f32, or 32 hi + 32 lo halves... 128 + 128)

// W = [Cout][K] row-major with K = C1 + C2 (segment 1 first) -> fragment order
__global__ void conv1x1_pack_kernel(const float *__restrict__ w1, int c1, const float *__restrict__ w2,
                                    int c2, int cout, float *__restrict__ wp) {
  const int k = c1 + c2, kgt = k / 8;
  const long long total = (long long)cout * k;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t & 3), lane = (int)((t >> 2) & 63);
    const long long q = t >> 8;
    const int kg = (int)(q % kgt), rb = (int)(q / kgt);
    const int co = 32 * rb + (lane & 31);
    const int ci = 8 * kg + 4 * (lane >> 5) + i;
    wp[t] = ci < c1 ? w1[(long long)co * c1 + ci] : w2[(long long)co * c2 + (ci - c1)];
  }
}

__global__ void conv1x1_pack16_kernel(const float *__restrict__ w1, int c1, const float *__restrict__ w2,
                                      int c2, int cout, const float *__restrict__ wmax,
                                      _Float16 *__restrict__ wp) {
  const float S = conv16_scale(*wmax);
  const int k = c1 + c2, kst = k / 16;
  const long long total = (long long)cout * k;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(t & 7), lane = (int)((t >> 3) & 63);
    const long long q = t >> 9;
    const int ks = (int)(q % kst), rb = (int)(q / kst);
    const int co = 32 * rb + (lane & 31);
    const int ci = 16 * ks + 8 * (lane >> 5) + e;
    const float v = (ci < c1 ? w1[(long long)co * c1 + ci] : w2[(long long)co * c2 + (ci - c1)]) * S;
    const _Float16 hi = (_Float16)v;
    const long long base = ((q * 2) * 64 + lane) * 8 + e;
    wp[base] = hi;
    wp[base + 64 * 8] = (_Float16)(v - (float)hi);
  }
}

template <bool F16, int MRW>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(Conv1Args p, const float *__restrict__ wmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x 16 KB stage (64 KB with y_hwc)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int tiles = p.hw / kC1Px;
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int px0 = tile * kC1Px;
  const int ktot = p.c1 + p.c2;
  const int n_chunks = ktot / kC1K;
  const WStream ws = make_wstream(p.wp, p.wp_floats, lane);
  constexpr int buf_bytes = kC1Px * kC1Row;  // 16 KB

  // staging: lane = pixel, wave wv = channels 16 wv .. 16 wv + 15 of the chunk
  f32x4 stg[4];
  int c0_staged = -1;  // first channel (of segment 1) in stg, -1: segment 2 (plain)
  __shared__ float gn_stats[64];  // (mean, rstd) of x1's 32 groups (csrc/gn_tail.h)
  // (scale, shift) of every channel of segment 1, once per workgroup: the staging of a chunk took them per
  // wave and channel from the statistics + gamma / beta -- ~100 VALU instructions and 32 loads per chunk and wave
  __shared__ __attribute__((aligned(16))) float ss1[2 * 512];
  auto stage_load = [&](int chunk) {
    const int c0 = chunk * kC1K + 16 * wv;  // first channel in the concatenated K
    const bool seg2 = c0 >= p.c1;
    const float *pl = seg2 ? p.x2 + ((long long)img * p.c2 + (c0 - p.c1)) * p.hw
                           : p.x1 + ((long long)img * p.c1 + c0) * p.hw;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        stg[q][k] = pl[(long long)(4 * q + k) * p.hw + px0 + lane];
    c0_staged = seg2 ? -1 : c0;
  };
  auto stage_store = [&](int chunk, unsigned char *buf) {
    const bool act = chunk * kC1K < p.c1 && p.relu1;
    unsigned char *row = buf + lane * kC1Row;
    const int sw = lane & 15;
    float sc[16], sh[16];  // wave-uniform
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 t4 = {1.0f, 0.0f, 1.0f, 0.0f};
      if (c0_staged >= 0) t4 = *reinterpret_cast<const f32x4 *>(ss1 + 2 * (c0_staged + 2 * q));
      sc[2 * q] = t4[0];
      sh[2 * q] = t4[1];
      sc[2 * q + 1] = t4[2];
      sh[2 * q + 1] = t4[3];
    }
    if constexpr (!F16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = fmaf(stg[q][k], sc[4 * q + k], sh[4 * q + k]);
          v[k] = act ? fmaxf(t, 0.0f) : t;
        }
        *reinterpret_cast<f32x4 *>(row + (((4 * wv + q) ^ sw) << 4)) = v;
      }
    } else {
      // 16 channels -> hi slots 2 wv, 2 wv + 1 and lo slots 8 + 2 wv, 8 + 2 wv + 1 (8 halves each)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = 8 * half + e;
          float t = fmaf(stg[c >> 2][c & 3], sc[c], sh[c]);
          t = act ? fmaxf(t, 0.0f) : t;
          hi[e] = (_Float16)t;
          lo[e] = (_Float16)(t - (float)hi[e]);
        }
        *reinterpret_cast<h8 *>(row + (((2 * wv + half) ^ sw) << 4)) = hi;
        *reinterpret_cast<h8 *>(row + (((8 + 2 * wv + half) ^ sw) << 4)) = lo;
      }
    }
  };

  f32x16 acc[MRW][2];
#pragma unroll
  for (int m = 0; m < MRW; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[m][n][t] = 0.0f;
  if constexpr (!F16) {  // the bias starts the sums (the split-f16 sums are scaled: their bias is added at the end)
    if (p.bias) {
#pragma unroll
      for (int m = 0; m < MRW; ++m)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float b = p.bias[32 * ((int)blockIdx.y * (4 * MRW) + MRW * wv + m) + (t & 3) + 8 * (t >> 2) + 4 * h];
          acc[m][0][t] = b;
          acc[m][1][t] = b;
        }
    }
  }

  // K steps per chunk: 8 groups of 8 (f32) / 4 steps of 16 (f16); fragments of row block rb, step s:
  //   f32: ((rb * kgt + s) * 64 + lane) float4;  f16: ((rb * kst + s) * 2 + part) * 64 + lane h8
  constexpr int SPC = F16 ? kC1K / 16 : kC1K / 8;   // steps per chunk
  constexpr int FPS = F16 ? 2 : 1;                  // 16-byte fragments per step and row block
  const int steps = n_chunks * SPC;
  const int rb_stride = steps * FPS * 64;
  const int rb0 = (int)blockIdx.y * (4 * MRW) + MRW * wv;  // first 32-row block of this wave
  const int a_base = rb0 * rb_stride;
  auto a_load = [&](int m, int s, int part) {
    return wload128(ws, a_base + m * rb_stride + (min(s, steps - 1) * FPS + part) * 64);
  };
  // A fragments kAhead steps ahead of their MFMAs: one step (16 MFMAs = 0.4 us at MRW = 2) does not
  // cover an L2 hit under load, and unlike the 3x3 kernel there are no taps to spread the stream over
  constexpr int kAhead = 3;
  f32x4 ring[4][MRW][FPS];  // [step & 3][m][part]
#pragma unroll
  for (int q = 0; q < kAhead; ++q)
#pragma unroll
    for (int m = 0; m < MRW; ++m)
#pragma unroll
      for (int part = 0; part < FPS; ++part) ring[q][m][part] = a_load(m, q, part);

  stage_load(0);
  GnAffine affine;
  gn_affine_load(p.gn1, p.c1, affine);
  gn_load_stats(p.gn1, img, gn_stats);
  __syncthreads();
  gn_table_fill(p.gn1, img, p.c1, gn_stats, affine, ss1);
  __syncthreads();
  stage_store(0, smem);
  __syncthreads();

  const int swj = j & 15;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char *buf = smem + (chunk & 1) * buf_bytes;
    const bool more = chunk + 1 < n_chunks;
    if (more) stage_load(chunk + 1);
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
      const int gs = chunk * SPC + s;
      // the A fragments of step gs + kAhead (SPC is a multiple of 4: gs & 3 == s & 3)
#pragma unroll
      for (int m = 0; m < MRW; ++m)
#pragma unroll
        for (int part = 0; part < FPS; ++part) ring[(s + kAhead) & 3][m][part] = a_load(m, gs + kAhead, part);
      if constexpr (!F16) {
        f32x4 b[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
          b[n] = *reinterpret_cast<const f32x4 *>(buf + (32 * n + j) * kC1Row + (((2 * s + h) ^ swj) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int m = 0; m < MRW; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[s & 3][m][0][i], b[n][i], acc[m][n], 0, 0, 0);
      } else {
        h8 bh[2], bl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const unsigned char *row = buf + (32 * n + j) * kC1Row;
          bh[n] = *reinterpret_cast<const h8 *>(row + (((2 * s + h) ^ swj) << 4));
          bl[n] = *reinterpret_cast<const h8 *>(row + (((8 + 2 * s + h) ^ swj) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MRW; ++m) {
          const h8 ah = __builtin_bit_cast(h8, ring[s & 3][m][0]);
          const h8 al = __builtin_bit_cast(h8, ring[s & 3][m][FPS - 1]);
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[n], acc[m][n], 0, 0, 0);
        }
      }
    }
    if (more) stage_store(chunk + 1, smem + ((chunk + 1) & 1) * buf_bytes);
    __syncthreads();
  }

  // ---- epilogue: values (+ bias, + residual) and their per-channel sums first ----
  const float inv_scale = F16 ? 1.0f / conv16_scale(*wmax) : 1.0f;
  constexpr int NCH = 128 * MRW;  // output channels of this workgroup
  double *cs = reinterpret_cast<double *>(smem);  // [NCH][2] per-channel (sum, sum of squares)
  // output / residual addresses: one buffer resource per image, the lane part (4 h channels down, pixel px0 + j)
  // in ONE register, the channel of accumulator register t in the scalar offset, the second column block in the
  // instruction's immediate -- no per-element 64-bit address arithmetic on the VALU
  const long long img_off = (long long)img * p.cout * p.hw;
  const int img_bytes = p.cout * p.hw * 4;
  const __amdgpu_buffer_rsrc_t rs_res =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res + img_off : p.x1), 0, p.res ? img_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y =
      __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y + img_off : const_cast<float *>(p.x1), 0, p.y ? img_bytes : 0, 0x00020000);
  const int vo = (4 * h * p.hw + px0 + j) * 4;
  auto so_of = [&](int m, int t) { return (32 * (rb0 + m) + (t & 3) + 8 * (t >> 2)) * p.hw * 4; };  // scalar
#pragma unroll
  for (int m = 0; m < MRW; ++m) {
    float s1[16], s2[16];  // one row block at a time: 64 live sums next to 128 accumulators spilled
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float b = 0.0f;
      if constexpr (F16) b = p.bias ? p.bias[32 * (rb0 + m) + (t & 3) + 8 * (t >> 2) + 4 * h] : 0.0f;
      s1[t] = s2[t] = 0.0f;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float v = F16 ? acc[m][n][t] * inv_scale + b : acc[m][n][t];
        if (p.res)
          v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, vo + 128 * n, so_of(m, t), 0));
        acc[m][n][t] = v;
        s1[t] += v;
        s2[t] = fmaf(v, v, s2[t]);
      }
    }
    if (gn_wanted(p.fin)) {
      // GroupNorm(32, Cout) statistics of the output (bn_end after conv_last; the first GroupNorm of
      // the next stack after x + bl(.) + al(.)), handed on before the bulk stores (gn_tail.h)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        s1[t] = half_wave_sum(s1[t]);
        s2[t] = half_wave_sum(s2[t]);
      }
      if (j == kHalfSumLane) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int lc = 32 * (MRW * wv + m) + (t & 3) + 8 * (t >> 2) + 4 * h;
          cs[2 * lc] = (double)s1[t];
          cs[2 * lc + 1] = (double)s2[t];
        }
      }
    }
  }
  if (gn_wanted(p.fin)) {
    __syncthreads();
    if (tid < 64) {
      const int cpg = p.cout / 32, ng = NCH / cpg;
      double a = 0.0, b = 0.0;
      if (tid < ng)
        for (int ch = 0; ch < cpg; ++ch) {
          a += cs[2 * (tid * cpg + ch)];
          b += cs[2 * (tid * cpg + ch) + 1];
        }
      gn_emit(p.fin, img, (NCH * (int)blockIdx.y) / cpg, ng, tile, a, b);
    }
  }
  if (p.y) {
#pragma unroll
    for (int m = 0; m < MRW; ++m)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
        {
          const float v = acc[m][n][t];  // (a named float: __builtin_bit_cast on the vector element stores element 0 for every t)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs_y, vo + 128 * n, so_of(m, t), 0);
        }
      }
  }
  if (p.y_hwc) {
    // [64 px][NCH ch] f32 through LDS (16-byte slots swizzled with the pixel), then every pixel row
    // leaves as one burst of NCH * 4 bytes: wave wv writes pixels wv, wv + 4, ...
    if (gn_wanted(p.fin)) __syncthreads();
    constexpr int kRow = NCH * 4, kSlots = NCH / 4;
    unsigned char *tr = smem;
#pragma unroll
    for (int m = 0; m < MRW; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int px = 32 * n + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = (32 * (MRW * wv + m) + 8 * q + 4 * h) >> 2;  // 4 consecutive channels
          const f32x4 v = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
          *reinterpret_cast<f32x4 *>(tr + px * kRow + ((slot ^ (px & (kSlots - 1))) << 4)) = v;
        }
      }
    __syncthreads();
    float *dst = p.y_hwc + ((long long)img * p.hw + px0) * p.cout + NCH * (int)blockIdx.y;
    if (lane < kSlots)
      for (int px = wv; px < kC1Px; px += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tr + px * kRow + ((lane ^ (px & (kSlots - 1))) << 4));
        *reinterpret_cast<f32x4 *>(dst + (long long)px * p.cout + 4 * lane) = v;
      }
  }
}

int launch_conv1x1_pack(mp_ctx *ctx, const float *w1, int c1, const float *w2, int c2, int cout, int f16,
                        void *wp, float *wmax, hipStream_t st) {
  const long long total = (long long)cout * (c1 + c2);
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (!f16) {
    hipLaunchKernelGGL(conv1x1_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w1, c1, w2, c2, cout,
                       static_cast<float *>(wp));
  } else {
    // max|W| over both segments: two absmax passes into the same word (atomicMax keeps the larger)
    MP_HIP(ctx, hipMemsetAsync(wmax, 0, sizeof(float), st));
    int rc = launch_absmax_accumulate(ctx, w1, (long long)cout * c1, reinterpret_cast<unsigned int *>(wmax), st);
    if (rc == MP_OK && c2 > 0)
      rc = launch_absmax_accumulate(ctx, w2, (long long)cout * c2, reinterpret_cast<unsigned int *>(wmax), st);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(conv1x1_pack16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w1, c1, w2, c2, cout,
                       wmax, static_cast<_Float16 *>(wp));
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

static int g_conv1_mrw = 0;  // 0 = heuristic; measurement hook (mp_conv3x3_tune(nr | mrw << 12))

void conv1x1_set_mrw(int mrw) { g_conv1_mrw = mrw; }

// slots per (image, group) of the statistics a 1x1 launch publishes
int conv1x1_stat_slices(long long hw) { return (int)(hw / kC1Px); }

// `a` arrives with the tensors, cout, gn1 and fin.{acc, partial} as the caller gave them;
// partial_cap: doubles allocated for fin.partial (-1 = unchecked legacy entry point)
int launch_conv1x1(mp_ctx *ctx, Conv1Args a, int f16, const float *wmax, long long partial_cap, hipStream_t st) {
  if (a.c1 <= 0 || a.c1 % kC1K || a.c2 < 0 || a.c2 % kC1K || a.hw % kC1Px || (a.c2 > 0) != (a.x2 != nullptr) ||
      (a.cout != 256 && a.cout != 128))
    return fail(ctx, MP_ERR_UNSUPPORTED,
                "conv1x1: needs C1, C2 multiples of 64, H*W a multiple of 64 and 128 or 256 output channels "
                "(got %d + %d -> %d, %d)", a.c1, a.c2, a.cout, a.hw);
  if (a.c1 > 512)
    return fail(ctx, MP_ERR_UNSUPPORTED, "conv1x1: at most 512 channels in segment 1 (got %d)", a.c1);
  if (a.cout != 256 && (gn_wanted(a.fin) || a.y_hwc))
    return fail(ctx, MP_ERR_UNSUPPORTED, "conv1x1: statistics / channels-last output are built for 256 channels");
  a.wp_floats = a.cout * (a.c1 + a.c2);
  const int tiles = a.hw / kC1Px;
  if (gn_wanted(a.fin)) {
    a.fin.c = a.cout;
    a.fin.S = tiles;
    a.fin.n = a.n_img;
    if (a.fin.partial && partial_cap >= 0 && partial_cap < (long long)a.n_img * 32 * tiles * 2)
      return fail(ctx, MP_ERR_ARG, "conv1x1: statistics buffer holds %lld doubles, the launch writes %lld",
                  partial_cap, (long long)a.n_img * 32 * tiles * 2);
  }
  if (gn_active(a.gn1)) {
    if (a.gn1.acc && (!a.gn1.gamma || !a.gn1.beta))
      return fail(ctx, MP_ERR_ARG, "conv1x1: GroupNorm hand-over without gamma / beta");
    a.gn1.c = a.c1;
    a.gn1.n = a.n_img;
    a.gn1.count = (double)(a.c1 / 32) * a.hw;
  }
  // one 32-row block per wave, the row halves of 256 output channels as separate workgroups (blockIdx.y): since
  // the epilogue addresses through buffer resources (round 4) that form needs 121 registers -- four workgroups
  // per CU -- and beats two blocks per wave (166 registers, three workgroups) on every shape of the encoder
  // (batch 16: conv_last 100 vs 97, l 106 vs 106, bl|al 103 vs 86 TFLOP/s); mp_conv3x3_tune can still force 2
  int mrw = 1;
  if (g_conv1_mrw > 0 && a.cout == 256) mrw = g_conv1_mrw;
  const int lds = a.y_hwc ? kC1Px * 512 * mrw : 2 * kC1Px * kC1Row;
  void (*kern)(Conv1Args, const float *) =
      mrw == 2 ? (f16 ? conv1x1_kernel<true, 2> : conv1x1_kernel<false, 2>)
               : (f16 ? conv1x1_kernel<true, 1> : conv1x1_kernel<false, 1>);
  int rc = raise_lds_limit(ctx, reinterpret_cast<const void *>(kern), kC1Px * 1024);
  if (rc != MP_OK) return rc;
  const dim3 grid((unsigned)(tiles * a.n_img), (unsigned)(a.cout / (128 * mrw)));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, wmax);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// legacy form: optional partial sums (finalised by mp_gn_finalize)
int launch_conv1x1_raw(mp_ctx *ctx, const float *x1, const float *ss1, int relu1, const float *x2, int n,
                       int c1, int c2, int cout, long long hw, const void *wp, int f16, const float *wmax,
                       const float *bias, const float *res, float *y, float *y_hwc, double *stats,
                       hipStream_t st) {
  Conv1Args a;
  a.x1 = x1;
  a.gn1 = gn_in_none();
  a.gn1.ss = ss1;
  a.x2 = x2;
  a.wp = static_cast<const float *>(wp);
  a.bias = bias;
  a.res = res;
  a.y = y;
  a.y_hwc = y_hwc;
  a.fin = gn_out_none();
  a.fin.partial = stats;
  a.n_img = n;
  a.c1 = c1;
  a.c2 = c2;
  a.hw = (int)hw;
  a.relu1 = relu1;
  a.cout = cout;
  return launch_conv1x1(ctx, a, f16, wmax, -1, st);
}

}  // namespace mp
