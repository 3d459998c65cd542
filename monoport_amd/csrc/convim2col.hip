// The encoders' remaining convolutions on f32 MFMA:
//   * the 7x7 stride-2 stem of the hourglass encoder (backbones/HGFilters.py:125, :168:
//     nn.Conv2d(3, 64, 7, 2, 3) + bias), 1.2 GFLOP per frame;
//   * netC's stem (backbones/ResBlkFilters.py:111-121): ReflectionPad2d(3) + Conv2d(3, 64, 7) and the
//     two Conv2d(C, 2C, 3, stride 2, padding 1), each followed by GroupNorm + ReLU, 24 GFLOP.
// They were the last layers on MIOpen (whose find mode also added ~39 ms of naive_conv trials to the
// first frame of a cold process, profiles/r02x_encoder_kernel_stats_b1.txt).
//
// One kernel, implicit GEMM with an explicit im2col tile in LDS: M = output channels, N = 64
// consecutive pixels of one output row, K walked in chunks of CC input channels x KS x KS taps
// (padded to a multiple of 8): the workgroup gathers [64 pixels][K chunk] into LDS -- strided /
// mirrored / zero-padded reads, with the producer's GroupNorm + ReLU applied on the way like
// conv3x3.hip does -- and the waves run v_mfma_f32_32x32x2_f32 over it with the weights streamed in
// fragment order.  The epilogue adds the bias, writes NCHW and hands the GroupNorm statistics of
// the output on (gn_tail.h).  Algorithmic work 2 KS^2 Cin Cout FLOP per output pixel; these layers are
// 12 % of netC's and 0.6 % of netG's encoder FLOPs, so the kernel is built for simplicity: the
// gather is not overlapped with the MFMAs inside a workgroup (a second workgroup per CU fills in).
// Round 4: the tile is K-major and the gather coalesced (one k per wave-load, lane = pixel).
#include "mp_internal.h"
#include "query_common.h"
#include "gn_tail.h"

namespace mp {

constexpr int ceil8(int v) { return (v + 7) / 8 * 8; }

// W [Cout][Cin][KS][KS] -> [rb][chunk][g][lane][4]: lane (r, hh) holds, for k = 8 g + 4 hh + i inside
// the chunk, W[32 rb + r][chunk CC + k / KS^2][tap k % KS^2] (0 for the padding k's).
__global__ void convk_pack_kernel(const float *__restrict__ w, int cout, int cin, int ks, int cc,
                                  float *__restrict__ wp) {
  const int taps = ks * ks, kc = ceil8(cc * taps), n_chunks = cin / cc;
  const long long total = (long long)cout * n_chunks * kc;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t & 3), lane = (int)((t >> 2) & 63);
    const long long q = t >> 8;  // (rb * n_chunks + chunk) * (kc / 8) + g
    const int g = (int)(q % (kc / 8));
    const int chunk = (int)((q / (kc / 8)) % n_chunks), rb = (int)(q / ((long long)(kc / 8) * n_chunks));
    const int k = 8 * g + 4 * (lane >> 5) + i;
    const int co = 32 * rb + (lane & 31);
    float v = 0.0f;
    if (k < cc * taps) v = w[((long long)co * cin + chunk * cc + k / taps) * taps + k % taps];
    wp[t] = v;
  }
}

constexpr int kConvkWg = 2;  // workgroups per CU the register allocator is held to (measured: 4 fits 128 registers only with spills)
template <int RBW, int NR, int KS, int CC>
__global__ __launch_bounds__(256, kConvkWg) void convk_kernel(ConvKArgs p) {
  constexpr int kRing = 6;  // weight fragments in flight
  constexpr int CW = 4 / RBW;
  constexpr int PX = 32 * NR * CW;         // output pixels per workgroup (one row segment)
  constexpr int TAPS = KS * KS;
  constexpr int KC = ceil8(CC * TAPS);     // K per chunk
  constexpr int NG = KC / 8;               // MFMA groups (8 deep) per chunk
  static_assert(PX == 64, "the gather gives every lane of a wave one of the tile's 64 pixels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // im2col tile, K-MAJOR [KC][64 pixels] (round 4): a wave gathers ONE k = (channel, tap) for the 64 pixels per load
  // -- consecutive (stride-`stride`) addresses of one input row, where the pixel-major tile of round 3 had
  // every lane of a load in a different channel plane (64 cache lines per load; 57 TFLOP/s) -- and writes 64
  // consecutive floats.  The MFMA B operand (4 consecutive k of one pixel) becomes four ds_read_b32 of
  // consecutive lanes instead of one ds_read_b128: conflict-free either way.
  float *bt = reinterpret_cast<float *>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int rbi = wv % RBW, cwi = wv / RBW;
  const int segs = p.wo / PX, tiles = segs * p.ho;
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int oy = tile / segs, ox0 = (tile % segs) * PX;
  const int rb = blockIdx.y * RBW + rbi;
  const int n_chunks = p.cin / CC;
  const long long hw_in = (long long)p.h * p.w;
  const float *xin = p.x + (long long)img * p.cin * hw_in;
  const bool norm = gn_active(p.gn);
  __shared__ float gn_stats[64];  // (mean, rstd) of the input's 32 groups (csrc/gn_tail.h)
  __shared__ float ss_tab[2 * 512];  // (scale, shift) of every input channel, once per workgroup (gn_tail.h)
  const WStream ws = make_wstream(p.wp, p.wp_floats, lane);
  GnAffine affine;
  gn_affine_load(p.gn, p.cin, affine);
  gn_load_stats(p.gn, img, gn_stats);
  __syncthreads();
  if (norm) gn_table_fill(p.gn, img, p.cin, gn_stats, affine, ss_tab);  // visible behind the first chunk's barrier

  f32x16 acc[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[n][t] = 0.0f;

  // The gather of chunk c + 1 is IN FLIGHT under the MFMAs of chunk c: its raw values wait in registers
  // (up to (CC * KS / 4 + 1) * KS per thread) and are normalised and written to the tile at the top of the next iteration.
  // addresses: one buffer resource over the image's input planes; the lane part is the COLUMN of a tap (KS
  // values, mirrored / or pushed out of range so that the load returns 0), the row and the channel plane go
  // into the scalar offset -- KS lane registers instead of one 64-bit address per gathered value
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xin), 0, (int)(p.cin * hw_in * 4), 0x00020000);
  constexpr int kOob = 0x7ffffff0;  // beyond num_records: the raw buffer load returns 0
  int colo[KS];
#pragma unroll
  for (int kx = 0; kx < KS; ++kx) {
    int ix = (ox0 + lane) * p.stride - p.pad + kx;
    bool okx = true;
    if (p.reflect)
      ix = ix < 0 ? -ix : (ix >= p.w ? 2 * p.w - 2 - ix : ix);
    else
      okx = ix >= 0 && ix < p.w;
    colo[kx] = okx ? ix * 4 : kOob;
  }
  // wave wv gathers the tile rows of the (channel, ky) pairs wv, wv + 4, ...: KS loads per pair that differ in
  // the lane offset colo[kx] only (kx a compile-time index: no per-load address registers to keep alive)
  constexpr int NP = (CC * KS + 3) / 4;
  auto pair_of = [&](int ip, int &c, int &iy, bool &ok) {  // all scalar
    const int pr = wv + 4 * ip;
    c = pr / KS;
    const int ky = pr - c * KS;
    iy = oy * p.stride - p.pad + ky;
    ok = pr < CC * KS;
    if (p.reflect)
      iy = iy < 0 ? -iy : (iy >= p.h ? 2 * p.h - 2 - iy : iy);
    else
      ok = ok && iy >= 0 && iy < p.h;
  };
  float raw[NP * KS];
  auto gather_load = [&](int chunk) {
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
      int c, iy;
      bool ok;
      pair_of(ip, c, iy, ok);
      const int soff = ok ? (int)(((long long)(chunk * CC + c) * hw_in + (long long)iy * p.w) * 4) : 0;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx)
        raw[ip * KS + kx] =
            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? colo[kx] : kOob, soff, 0));
    }
  };
  // the K padding rows of the tile (CC * TAPS .. KC) are zero for every chunk
  for (int e = tid; e < (KC - CC * TAPS) * PX; e += 256) bt[CC * TAPS * PX + e] = 0.0f;
  gather_load(0);
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int a_base = (rb * n_chunks + chunk) * NG * 64;
    f32x4 ring[kRing];
#pragma unroll
    for (int k = 0; k < kRing; ++k) ring[k] = wload128(ws, a_base + (k < NG ? k : NG - 1) * 64);
    // (scale, shift) of the chunk's input channels, once per chunk instead of once per gathered element
    __syncthreads();  // the previous chunk's tile has been consumed
    // ---- the gathered values of this chunk -> the tile: wave wv holds k = wv, wv + 4, ...; lane = pixel ----
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
      int c, iy;
      bool ok;
      pair_of(ip, c, iy, ok);
      if (wv + 4 * ip < CC * KS) {
        const float sc = norm ? ss_tab[2 * (chunk * CC + c)] : 1.0f, sh = norm ? ss_tab[2 * (chunk * CC + c) + 1] : 0.0f;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          float v = raw[ip * KS + kx];
          // (a value gathered from outside the image is 0 and stays 0: padding comes after GroupNorm + ReLU)
          if (norm && ok && colo[kx] != kOob) {
            v = fmaf(v, sc, sh);
            if (p.relu) v = fmaxf(v, 0.0f);
          }
          bt[((wv + 4 * ip) * KS + kx) * PX + lane] = v;
        }
      }
    }
    __syncthreads();
    if (chunk + 1 < n_chunks) gather_load(chunk + 1);  // lands under the MFMAs below
    // ---- MFMAs over the chunk ----
    auto b_read = [&](int n, int g) {
      const float *col = bt + (8 * g + 4 * h) * PX + 32 * (cwi * NR + n) + j;
      const f32x4 b = {col[0], col[PX], col[2 * PX], col[3 * PX]};
      return b;
    };
    f32x4 bcur[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) bcur[n] = b_read(n, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f32x4 bnxt[NR];
      if (g + 1 < NG) {
#pragma unroll
        for (int n = 0; n < NR; ++n) bnxt[n] = b_read(n, g + 1);
      }
      const f32x4 a = ring[g % kRing];
      if (g + kRing < NG) ring[g % kRing] = wload128(ws, a_base + (g + kRing) * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < NR; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bcur[n][i], acc[n], 0, 0, 0);
      if (g + 1 < NG) {
#pragma unroll
        for (int n = 0; n < NR; ++n) bcur[n] = bnxt[n];
      }
    }
  }
  __syncthreads();  // the tile is dead: LDS is reused for the statistics

  // ---- epilogue: + bias, NCHW store, GroupNorm statistics of the output ----
  constexpr int NCH = 32 * RBW;
  const long long hwo = (long long)p.ho * p.wo;
  float s1[16], s2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int co = 32 * rb + (t & 3) + 8 * (t >> 2) + 4 * h;
    const float b = p.bias ? p.bias[co] : 0.0f;
    s1[t] = s2[t] = 0.0f;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const float v = acc[n][t] + b;
      p.y[((long long)img * p.cout + co) * hwo + (long long)oy * p.wo + ox0 + 32 * (cwi * NR + n) + j] = v;
      s1[t] += v;
      s2[t] = fmaf(v, v, s2[t]);
    }
  }
  if (gn_wanted(p.fin)) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      s1[t] = half_wave_sum(s1[t]);
      s2[t] = half_wave_sum(s2[t]);
    }
    double *cs = reinterpret_cast<double *>(smem);  // [CW][NCH][2]
    if (j == kHalfSumLane) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int idx = cwi * NCH + 32 * rbi + (t & 3) + 8 * (t >> 2) + 4 * h;
        cs[2 * idx] = (double)s1[t];
        cs[2 * idx + 1] = (double)s2[t];
      }
    }
    __syncthreads();
    const int cpg = p.cout / 32, ng = NCH / cpg;
    double a = 0.0, b = 0.0;
    if (tid < ng)
      for (int cw = 0; cw < CW; ++cw)
        for (int ch = 0; ch < cpg; ++ch) {
          const int idx = cw * NCH + tid * cpg + ch;
          a += cs[2 * idx];
          b += cs[2 * idx + 1];
        }
    gn_emit(p.fin, img, (NCH * (int)blockIdx.y) / cpg, ng, tile, a, b);
  }
}

bool convk_supported(int cin, int cout, int ks, int stride, int h, int w) {
  if (ks == 7 && cin == 3 && cout == 64 && (stride == 1 || stride == 2))
    return h % stride == 0 && w % stride == 0 && (w / stride) % 64 == 0;
  if (ks == 3 && stride == 2 && cin % 16 == 0 && cin >= 16 && cout % 128 == 0)
    return h % 2 == 0 && w % 2 == 0 && (w / 2) % 64 == 0;
  return false;
}

long long convk_packed_floats(int cin, int cout, int ks) {
  const int cc = ks == 7 ? 3 : 16;
  return (long long)cout * (cin / cc) * ceil8(cc * ks * ks);
}

int convk_stat_slices(int ks, int stride, int h, int w) { return (h / stride) * ((w / stride) / 64); }

int launch_convk_pack(mp_ctx *ctx, const float *w, int cout, int cin, int ks, float *wp, hipStream_t st) {
  const int cc = ks == 7 ? 3 : 16;
  const long long total = convk_packed_floats(cin, cout, ks);
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(convk_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cout, cin, ks, cc, wp);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

template <int RBW, int NR, int KS, int CC>
static int launch_convk_t(mp_ctx *ctx, const ConvKArgs &a, hipStream_t st) {
  constexpr int PX = 32 * NR * (4 / RBW);
  constexpr int lds_tile = PX * ceil8(CC * KS * KS) * 4;
  constexpr int lds = lds_tile > 4096 ? lds_tile : 4096;
  auto kern = convk_kernel<RBW, NR, KS, CC>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ctx->lds_attr_done.insert(kern_id);
  }
  const int tiles = a.ho * (a.wo / PX);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / (32 * RBW))), dim3(256), lds, st, a);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// `a` arrives with tensors, shapes, ks / stride / pad / reflect / relu, gn and fin as the caller gave them
int launch_convk(mp_ctx *ctx, ConvKArgs a, long long partial_cap, hipStream_t st) {
  if (!convk_supported(a.cin, a.cout, a.ks, a.stride, a.h, a.w))
    return fail(ctx, MP_ERR_UNSUPPORTED,
                "convk: built for 7x7 (3 -> 64, stride 1 / 2) and 3x3 stride 2 (Cin %% 16 == 0, Cout %% 128 == 0), "
                "output width a multiple of 64; got %dx%d stride %d, %d -> %d at %dx%d", a.ks, a.ks, a.stride,
                a.cin, a.cout, a.h, a.w);
  if (a.pad != a.ks / 2) return fail(ctx, MP_ERR_UNSUPPORTED, "convk: padding must be ks / 2");
  a.ho = a.h / a.stride;
  a.wo = a.w / a.stride;
  a.wp_floats = (int)convk_packed_floats(a.cin, a.cout, a.ks);
  if (gn_wanted(a.fin)) {
    a.fin.c = a.cout;
    a.fin.S = a.ho * (a.wo / 64);
    a.fin.n = a.n_img;
    if (a.fin.partial && partial_cap >= 0 && partial_cap < (long long)a.n_img * 32 * a.fin.S * 2)
      return fail(ctx, MP_ERR_ARG, "convk: statistics buffer holds %lld doubles, the launch writes %lld",
                  partial_cap, (long long)a.n_img * 32 * a.fin.S * 2);
  }
  if (gn_active(a.gn)) {
    if (a.cin % 32) return fail(ctx, MP_ERR_ARG, "convk: a GroupNorm(32, Cin) input needs Cin %% 32 == 0");
    if (a.cin > 512) return fail(ctx, MP_ERR_UNSUPPORTED, "convk: a GroupNorm input of at most 512 channels (got %d)", a.cin);
    if (a.gn.acc && (!a.gn.gamma || !a.gn.beta))
      return fail(ctx, MP_ERR_ARG, "convk: GroupNorm hand-over without gamma / beta");
    a.gn.c = a.cin;
    a.gn.n = a.n_img;
    a.gn.count = (double)(a.cin / 32) * a.h * a.w;
  }
  if (a.ks == 7) return launch_convk_t<2, 1, 7, 3>(ctx, a, st);
  return launch_convk_t<4, 2, 3, 16>(ctx, a, st);
}

}  // namespace mp
