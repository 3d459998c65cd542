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
//   * the epilogue writes NCHW and, on request, the per-channel sum / sum of squares of the tile
//     (f32 over <= 128 values, then double): the NEXT GroupNorm's statistics without another pass;
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

namespace mp {

constexpr int kCK = 16;           // input channels per LDS chunk
constexpr int kPixBytes = kCK * 4;  // 64 bytes per staged pixel
#ifndef MP_CONV_WPS
#define MP_CONV_WPS 2  // waves per SIMD the register allocator is held to (tools/ablate.py A/B)
#endif
constexpr int kTileW = 32;        // tiles are 32 pixels wide and PX / 32 rows tall
// staged pixels (tile + 1 halo) of a PX-pixel tile, and the 64-lane passes a wave needs for them
constexpr int halo_pixels(int px) { return (px / kTileW + 2) * (kTileW + 2); }
constexpr int stage_iters(int px) { return (halo_pixels(px) + 63) / 64; }

struct ConvArgs {
  const float *x;    // [N, Cin, H, W]
  const float *ss;   // [N, Cin, 2] (scale, shift) of the fused GroupNorm, or nullptr: plain input
  const float *wp;   // packed weights [Cout/32][Cin/16 * 18][64][4]
  float *y;          // [N, Cout, H, W]
  double *stats;     // nullptr or [N, 32, S, 2] partial (sum, sumsq), S = slots * (Cout/32)
  int n_img, cin, cout, h, w;
  int tw, th;        // tile width / height in pixels (th * tw = 32 * NR * CW)
  int relu;          // apply ReLU to the (normalised) input
  int wp_floats;     // size of wp
};

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

template <int RBW, int NR>
__global__ __launch_bounds__(256, MP_CONV_WPS) void conv3x3_gn_kernel(ConvArgs p) {
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
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = lp < NPH && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    goff[it] = ok ? gy * p.w + gx : -1;
  }
  const float *xin = p.x + (long long)img * p.cin * hw;
  const float *ssn = p.ss ? p.ss + (long long)img * p.cin * 2 : nullptr;

  f32x4 stg[kStageIters];
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};  // wave-uniform (SGPRs)
  auto stage_load = [&](int chunk) {
    const float *pl = xin + (long long)(chunk * kCK + 4 * wv) * hw;
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int o = goff[it] < 0 ? 0 : goff[it];
#pragma unroll
      for (int k = 0; k < 4; ++k) stg[it][k] = pl[(long long)k * hw + o];
    }
    if (ssn) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sc[k] = ssn[2 * (chunk * kCK + 4 * wv + k)];
        sh[k] = ssn[2 * (chunk * kCK + 4 * wv + k) + 1];
      }
    }
  };
  auto stage_store = [&](unsigned char *buf) {
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
  int lpc[NR];  // linear halo-pixel index of this lane's pixel, per column block
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / TW, tx = (32 * cb) - ty * TW;
    lpc[n] = (ty + 1) * PW + tx + j + 1;
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
      const int row_off = (ky - 1) * PW;
      // B operand of the first group of this row-step
      f32x4 bcur[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int lp = lpc[n] + row_off - 1;
        bcur[n] = *reinterpret_cast<const f32x4 *>(buf + lp * kPixBytes + ((h ^ ((lp >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) {  // s = 2 * kx + g
        // next group's B operand (the first group of the next row-step is read at its top)
        f32x4 bnxt[NR];
        if (s < 5) {
          const int sn = s + 1;
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            const int lp = lpc[n] + row_off + (sn >> 1) - 1;
            bnxt[n] = *reinterpret_cast<const f32x4 *>(
                buf + lp * kPixBytes + (((2 * (sn & 1) + h) ^ ((lp >> 2) & 3)) << 4));
          }
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

  // ---- epilogue: NCHW store + per-channel statistics of this wave's 32 x (32 NR) tile ----
  float *yb = p.y + ((long long)img * p.cout + 32 * rb) * hw;
  float s1[16], s2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) s1[t] = s2[t] = 0.0f;
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / TW, tx = (32 * cb) - ty * TW;
    float *row = yb + (long long)(y0 + ty) * p.w + x0 + tx + j;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int r = (t & 3) + 8 * (t >> 2) + 4 * h;
      const float v = acc[n][t];
      row[(long long)r * hw] = v;
      s1[t] += v;
      s2[t] = fmaf(v, v, s2[t]);
    }
  }
  if (p.stats) {
    // sum over the 32 lanes that share h (the pixels); lanes j == 0 then hold the row sums
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        s1[t] += __shfl_xor(s1[t], o);
        s2[t] += __shfl_xor(s2[t], o);
      }
    }
    if (j == 0) {
      const int cpg = p.cout / 32;                 // channels per GroupNorm(32, Cout) group
      const int slots = tiles * CW;
      const int S = slots * cpg;
      const int slot = tile * CW + cwi;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int co = 32 * rb + (t & 3) + 8 * (t >> 2) + 4 * h;
        const int grp = co / cpg, s = slot * cpg + (co - grp * cpg);
        double *dst = p.stats + (((long long)img * 32 + grp) * S + s) * 2;
        dst[0] = (double)s1[t];
        dst[1] = (double)s2[t];
      }
    }
  }
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
__global__ __launch_bounds__(256, MP_CONV_WPS) void conv3x3_gn16_kernel(ConvArgs p, const float *__restrict__ wmax) {
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
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = lp < NPH && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    goff[it] = ok ? gy * p.w + gx : -1;
  }
  const float *xin = p.x + (long long)img * p.cin * hw;
  const float *ssn = p.ss ? p.ss + (long long)img * p.cin * 2 : nullptr;

  f32x4 stg[kStageIters];
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  auto stage_load = [&](int chunk) {
    const float *pl = xin + (long long)(chunk * kCK + 4 * wv) * hw;
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
      const int o = goff[it] < 0 ? 0 : goff[it];
#pragma unroll
      for (int k = 0; k < 4; ++k) stg[it][k] = pl[(long long)k * hw + o];
    }
    if (ssn) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sc[k] = ssn[2 * (chunk * kCK + 4 * wv + k)];
        sh[k] = ssn[2 * (chunk * kCK + 4 * wv + k) + 1];
      }
    }
  };
  // channels 4 wv .. 4 wv + 3 of the chunk = 8 bytes at offset 8 (wv & 1) of hi slot (wv >> 1) and
  // of lo slot 2 + (wv >> 1)
  auto stage_store = [&](unsigned char *buf) {
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

  int lpc[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / TW, tx = (32 * cb) - ty * TW;
    lpc[n] = (ty + 1) * PW + tx + j + 1;
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
  stage_store(smem);
  __syncthreads();

  int ks0 = 0;  // first k16 step of the current row-step
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char *buf = smem + (chunk & 1) * buf_bytes;
    const bool more = chunk + 1 < n_chunks;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int row_off = (ky - 1) * PW;
      h8 bh[NR], bl[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int lp = lpc[n] + row_off - 1;
        const int sw = (lp >> 2) & 3;
        bh[n] = *reinterpret_cast<const h8 *>(buf + lp * kPixBytes + ((h ^ sw) << 4));
        bl[n] = *reinterpret_cast<const h8 *>(buf + lp * kPixBytes + (((2 + h) ^ sw) << 4));
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        h8 nh[NR], nl[NR];
        if (kx < 2) {
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            const int lp = lpc[n] + row_off + kx;  // tap kx + 1: dx = kx
            const int sw = (lp >> 2) & 3;
            nh[n] = *reinterpret_cast<const h8 *>(buf + lp * kPixBytes + ((h ^ sw) << 4));
            nl[n] = *reinterpret_cast<const h8 *>(buf + lp * kPixBytes + (((2 + h) ^ sw) << 4));
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

  float *yb = p.y + ((long long)img * p.cout + 32 * rb) * hw;
  float s1[16], s2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) s1[t] = s2[t] = 0.0f;
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int cb = cwi * NR + n;
    const int ty = (32 * cb) / TW, tx = (32 * cb) - ty * TW;
    float *row = yb + (long long)(y0 + ty) * p.w + x0 + tx + j;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int r = (t & 3) + 8 * (t >> 2) + 4 * h;
      const float v = acc[n][t] * inv_scale;
      row[(long long)r * hw] = v;
      s1[t] += v;
      s2[t] = fmaf(v, v, s2[t]);
    }
  }
  if (p.stats) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        s1[t] += __shfl_xor(s1[t], o);
        s2[t] += __shfl_xor(s2[t], o);
      }
    }
    if (j == 0) {
      const int cpg = p.cout / 32;
      const int slots = tiles * CW;
      const int S = slots * cpg;
      const int slot = tile * CW + cwi;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int co = 32 * rb + (t & 3) + 8 * (t >> 2) + 4 * h;
        const int grp = co / cpg, s = slot * cpg + (co - grp * cpg);
        double *dst = p.stats + (((long long)img * 32 + grp) * S + s) * 2;
        dst[0] = (double)s1[t];
        dst[1] = (double)s2[t];
      }
    }
  }
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

int launch_conv3x3_pack(mp_ctx *ctx, const float *w, int cout, int cin, float *wp, hipStream_t st) {
  const long long total = (long long)cout * cin * 9;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cout, cin, wp);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// Tile shape of the kernel instantiation that serves `cout`: RBW waves split the output channels
// (32 each), the other 4 / RBW split the pixels; NR = 32-pixel column blocks per wave.  Smaller NR
// = more, smaller workgroups (better balance over 256 CUs x 2 slots, more weight re-streaming).
static int g_conv_nr = 0;  // 0 = heuristic; tools/conv_probe.py overrides it for A/B runs

void conv3x3_set_nr(int nr) { g_conv_nr = nr; }

static void conv_shape(int cout, int n, int h, int w, int &rbw, int &nr, int &tw, int &th, int &slots) {
  rbw = cout % 128 == 0 ? 4 : cout % 64 == 0 ? 2 : 1;
  const int cw = 4 / rbw;
  // NR = 2 unless that leaves fewer than 2 workgroups per slot (512 slots); NR = 4 needs all 256
  // VGPRs (spills) and measured slower at every shape, so it is not built
  nr = 2;
  if (g_conv_nr > 0)
    nr = g_conv_nr > 2 ? 2 : g_conv_nr;
  else if ((long long)n * h * w / (32 * nr * cw) * (cout / (32 * rbw)) < 1024)
    nr = 1;
  const int px = 32 * nr * cw;
  // 32-pixel-wide tiles, PX / 32 rows tall: the staged halo is (TH + 2) x 34 pixels (1.3-2.1x the
  // tile; one-row tiles would stage 3x) and a wave stages it in 2-6 passes of 64 pixels
  tw = kTileW;
  th = px / tw;
  slots = (h / th) * (w / tw) * cw;
}

int conv3x3_stat_slices(int cout, int n, int h, int w) {
  int rbw, nr, tw, th, slots;
  conv_shape(cout, n, h, w, rbw, nr, tw, th, slots);
  return slots * (cout / 32);
}

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

template <int RBW, int NR>
static int launch_conv16_t(mp_ctx *ctx, const ConvArgs &a, const float *wmax, int tiles, hipStream_t st) {
  const int lds = 2 * (a.th + 2) * (a.tw + 2) * kPixBytes;
  auto kern = conv3x3_gn16_kernel<RBW, NR>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * halo_pixels(32 * NR * (4 / RBW)) * kPixBytes));
    ctx->lds_attr_done.insert(kern_id);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / (32 * RBW))), dim3(256),
                     lds, st, a, wmax);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

template <int RBW, int NR>
static int launch_conv_t(mp_ctx *ctx, const ConvArgs &a, int tiles, hipStream_t st) {
  const int lds = 2 * (a.th + 2) * (a.tw + 2) * kPixBytes;
  auto kern = conv3x3_gn_kernel<RBW, NR>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * halo_pixels(32 * NR * (4 / RBW)) * kPixBytes));
    ctx->lds_attr_done.insert(kern_id);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / (32 * RBW))), dim3(256),
                     lds, st, a);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_conv3x3_gn(mp_ctx *ctx, const float *x, int n, int cin, int h, int w, const float *ss,
                      int relu, const float *wp, const float *wmax16, int cout, float *y, double *stats,
                      hipStream_t st) {
  if (!conv3x3_supported(cin, cout, h, w))
    return fail(ctx, MP_ERR_UNSUPPORTED,
                "conv3x3: needs Cin %% 16 == 0, Cout %% 32 == 0, H and W powers of two (W >= 32, H >= 8); got %d -> %d at %dx%d",
                cin, cout, h, w);
  ConvArgs a;
  a.x = x;
  a.ss = ss;
  a.wp = wp;
  a.y = y;
  a.stats = stats;
  a.n_img = n;
  a.cin = cin;
  a.cout = cout;
  a.h = h;
  a.w = w;
  a.relu = relu;
  a.wp_floats = cout * cin * 9;
  int rbw, nr, slots;
  conv_shape(cout, n, h, w, rbw, nr, a.tw, a.th, slots);
  if (a.th > h)
    return fail(ctx, MP_ERR_UNSUPPORTED, "conv3x3: %dx%d map too small for a %dx%d tile", h, w, a.th, a.tw);
  const int tiles = (h / a.th) * (w / a.tw);
#define MP_CONV_CASE(R, N)                                                                   \
  if (rbw == R && nr == N)                                                                   \
    return wmax16 ? launch_conv16_t<R, N>(ctx, a, wmax16, tiles, st) : launch_conv_t<R, N>(ctx, a, tiles, st);
  MP_CONV_CASE(4, 2)
  MP_CONV_CASE(4, 1)
  MP_CONV_CASE(2, 2)
  MP_CONV_CASE(2, 1)
  MP_CONV_CASE(1, 2)
  MP_CONV_CASE(1, 1)
#undef MP_CONV_CASE
  return fail(ctx, MP_ERR_UNSUPPORTED, "conv3x3: no instantiation for rbw %d nr %d", rbw, nr);
}

int launch_gn_finalize(mp_ctx *ctx, const double *partial, int n, int c, int groups, int slices,
                       double count, const float *gamma, const float *beta, float eps, float *ss,
                       hipStream_t st) {
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)(n * groups)), dim3(64), 0, st, partial, groups,
                     slices, count, c / groups, gamma, beta, eps, ss);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
