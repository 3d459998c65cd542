// 3x3 convolutions of the hourglass encoder as Winograd F(2x2, 3x3) on f32 MFMA (round 6).
//
// The pyramid block (backbones/HGFilters.py:40-62) is conv3x3(relu(GroupNorm(x))) three times, and the 3x3
// convolutions are 176 of the 197 GFLOP of a netG.filter pass (RTL/main.py:366-370).  csrc/conv3x3.hip runs them as a
// direct implicit GEMM (K = 9 Cin) at 0.85 of the f32 MFMA roof for its best shape -- the roof, not the schedule, is
// what is left.  F(2x2, 3x3) (Lavin & Gray) computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per
// (Cin, Cout) pair instead of 36:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A
// so the convolution becomes 16 independent GEMMs  M[i][j] = U[i][j] (Cout x Cin) . V[i][j] (Cin x tiles)  -- 2.25x
// fewer MFMAs -- between an input transform (adds only) and an output transform (adds only).  The transformed weights
// U = G g G^T are computed once, in double, by conv3x3_wino_pack_kernel.  Everything around the GEMMs is what
// conv3x3.hip does: GroupNorm + ReLU applied while the input is staged (zero padding of the NORMALISED tensor, or the
// reflection padding of the ResNet encoder), and conv3x3.hip's epilogue (raw output, pyramid-block tail, the next
// GroupNorms' statistics by integer atomics).
//
// Two kernels.  conv3x3_wino_kernel<2> (below: one workgroup = 8 waves = 512 threads, one per CU) and, further down,
// conv3x3_wino64_kernel (64 output channels, 8-channel chunks, two workgroups per CU); the launcher picks by size.
// Decomposition of the first:
//   * workgroup = 8 x 4 Winograd tiles (16 x 8 output pixels, 18 x 10 input patch) x 128 output channels;
//   * K loop over 16-channel chunks of the input, one barrier per chunk, three stages in flight:
//       chunk k + 2/3: global -> registers (lane = patch pixel, wave = 2 channel planes), GroupNorm + ReLU,
//                      -> raw[2] in LDS, pixel-major (80-byte rows: 64 used, the pad keeps 128-bit accesses off each
//                      other's banks);
//       chunk k + 1:   input transform raw -> V[2][i][j][tile][16 ch]: thread = (tile, 4 channels, column j),
//                      8 ds_read_b128 + 14 packed multiply-adds / adds + 4 ds_write_b128;
//       chunk k:       wave (j, half) multiplies frequencies (0..3, j) for its two 32-channel row blocks: A = U
//                      fragments streamed from L2 in fragment order, B = V rows from LDS, 64 MFMAs per chunk;
//   * output transform: rows in registers (A^T M), columns through LDS (. A); wave (j, half) then owns output row
//     r = j & 1 of every tile for row block j >> 1 of its half and runs conv3x3.hip's epilogue on it.
// Executed FLOPs: 2 * 16 * Cin * Cout per 2 x 2 output pixels = 4 / 9 of the direct form.  Rounding: the products are
// exact-f32 MFMA FMA chains as before; the transforms add 2-3 roundings per side (F(2x2, 3x3) has the mildest
// constants of the family: 0, +-1, +-1/2) -- measured against the fp64 convolution in tests/test_conv_wino_gpu.py
// (closer to it than the direct kernel: 256 instead of 2304 additions per output).
#include "mp_internal.h"
#include "query_common.h"
#include "gn_tail.h"

#pragma clang fp contract(off)

namespace mp {

constexpr int kWnThreads = 512;
constexpr int kWnTX = 8, kWnTY = 4;              // Winograd tiles of a workgroup
constexpr int kWnPW = 2 * kWnTX + 2;             // 18 patch columns
constexpr int kWnPH = 2 * kWnTY + 2;             // 10 patch rows
constexpr int kWnPix = kWnPW * kWnPH;            // 180 staged pixels
constexpr int kWnPasses = (kWnPix + 63) / 64;    // 3 passes of 64 lanes
constexpr int kWnRow = 80;                       // bytes per LDS row of 16 channels (64 used): with the pad, 16 lanes of a
                                                 // 128-bit access fall into 16 different 16-byte bank groups -- no swizzle,
                                                 // so every address of a thread is ONE register + an immediate offset
constexpr int kWnRawBytes = kWnPix * kWnRow;     // one raw buffer: [pixel][16 ch] f32
constexpr int kWnVBytes = 16 * 32 * kWnRow;      // one V buffer: [i][j][tile][16 ch] f32
constexpr int kWnRaw = 0;
constexpr int kWnV = 2 * kWnRawBytes;
constexpr int kWnXchBytes = 8 * 2 * 2 * 4 * 64 * 16;  // output-transform exchange: [wave][r][m][q][lane] f32x4 (128 KB)
constexpr int kWnStat = kWnXchBytes;             // WinoTail's statistics scratch (8 KB)
constexpr int kWnLds = kWnStat + 8192;
constexpr int kWnAhead = 3;                    // A fragments are requested this many steps (of 16 MRB MFMAs) ahead
static_assert(kWnV + 2 * kWnVBytes <= kWnXchBytes, "the exchange region covers the K loop's buffers");
static_assert(kWnLds <= 160 * 1024, "LDS");

// W [Cout][Cin][3][3] -> U = G g G^T in MFMA fragment order: fragment ((((rb * 4 + j) * chunks + chunk) * 4 + i) * 2 + g)
// holds, for lane (r = lane & 31, hh = lane >> 5), U[i][j][32 rb + r][16 chunk + 8 g + 4 hh + 0..3].  Wave (j, half) of
// the kernel streams the fragments of (rb, j) front to back.
__global__ void conv3x3_wino_pack_kernel(const float *__restrict__ w, int cout, int cin, float *__restrict__ up) {
  const long long total = 16LL * cout * cin;
  const int n_chunks = cin / 16;
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int ii = (int)(t & 3), lane = (int)((t >> 2) & 63);
    long long q = t >> 8;
    const int g = (int)(q & 1);
    q >>= 1;
    const int i = (int)(q & 3);
    q >>= 2;
    const int chunk = (int)(q % n_chunks);
    q /= n_chunks;
    const int j = (int)(q & 3), rb = (int)(q >> 2);
    const int co = 32 * rb + (lane & 31);
    const int ci = 16 * chunk + 8 * g + 4 * (lane >> 5) + ii;
    const float *gk = w + ((long long)co * cin + ci) * 9;
    double s = 0.0;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) s += G[i][a] * (double)gk[3 * a + b] * G[j][b];
    up[t] = (float)s;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// a - b on four floats as two v_pk_add_f32 with negated second operands: hipcc has no packed subtract and emits four
// v_sub_f32 (it also folds fma(b, -1, a) back into them), and in the K loops every VALU instruction of a wave costs the
// other waves of its SIMD matrix-pipe time (side build without the input transform: -8.5 %)
__device__ __forceinline__ f32x4 pk_sub4(const f32x4 &a, const f32x4 &b) {
  f32x2 lo, hi;
  const f32x2 alo = {a[0], a[1]}, ahi = {a[2], a[3]}, blo = {b[0], b[1]}, bhi = {b[2], b[3]};
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// The epilogue of conv3x3.hip (conv_epilogue: raw output, pyramid-block tail y2 = conv + res, the next GroupNorms'
// statistics through csrc/gn_tail.h) for this kernel's register layout: a wave holds, for the 32-row block `rbi` of the
// workgroup's NCH channels, output row r_o of every Winograd tile -- lane (j = tile, h), pr[t] = the two neighbouring
// pixels (2 tx, 2 tx + 1) of channel row (t & 3) + 8 (t >> 2) + 4 h, t = t0 + 0 .. TN - 1.  The pair is 8-byte aligned:
// 64-bit accesses.  Two calls: WinoTail::load_res right after the K loop (the residual's round trip to memory runs under
// the output transform), WinoTail::finish with the outputs.  smem: 2 x [2][NCH][2] doubles of scratch.
template <int NCH, int TN>
struct WinoTail {
  const ConvArgs &p;
  int t0, img, tile, rbi, r_o, j, h, hw, ch0, vo;
  bool cat;
  __amdgpu_buffer_rsrc_t rs_y, rs_y2, rs_res;
  f32x2 u[TN];  // the block tail: conv + res

  __device__ __forceinline__ WinoTail(const ConvArgs &p_, int t0_, int img_, int tile_, int y0, int x0, int rbi_, int r_o_, int lane)
      : p(p_), t0(t0_), img(img_), tile(tile_), rbi(rbi_), r_o(r_o_) {
    j = lane & 31;
    h = lane >> 5;
    hw = p.h * p.w;
    ch0 = NCH * blockIdx.y + 32 * rbi;
    cat = p.y2 != nullptr;
    rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y + (long long)img * p.cout * hw : const_cast<float *>(p.x), 0,
                                             p.y ? p.cout * hw * 4 : 0, 0x00020000);
    rs_y2 = __builtin_amdgcn_make_buffer_rsrc(cat ? p.y2 + (long long)img * p.y2_c * hw : const_cast<float *>(p.x), 0,
                                              cat ? p.y2_c * hw * 4 : 0, 0x00020000);
    rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cat ? p.res + (long long)img * p.y2_c * hw : p.x), 0,
                                               cat ? p.y2_c * hw * 4 : 0, 0x00020000);
    vo = (4 * h * hw + (y0 + 2 * (j >> 3) + r_o) * p.w + x0 + 2 * (j & 7)) * 4;
  }
  __device__ __forceinline__ int so1(int tt) const {  // scalar
    const int t = t0 + tt;
    return (ch0 + (t & 3) + 8 * (t >> 2)) * hw * 4;
  }
  __device__ __forceinline__ int so2(int tt) const { return so1(tt) + p.y2_off * hw * 4; }

  __device__ __forceinline__ void load_res() {
    if (cat) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
        u[t] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, vo, so2(t), 0));
    }
  }

  // per-channel (sum, sum of squares) of a wave's pixels -> LDS, the channels of one lane that share a GroupNorm group
  // added up first: rows (t & 3) of a register quad are 4 consecutive channels, so with >= 4 channels per group one
  // cross-lane sum serves four registers (2 with 2 channels per group): 4 (2) x fewer DPP chains
  __device__ __forceinline__ void to_lds(double *cs, const float (&a1)[TN], const float (&a2)[TN], int cpg) const {
    if (cpg >= 4) {
#pragma unroll
      for (int q = 0; q < TN / 4; ++q) {
        const float s1 = half_wave_sum((a1[4 * q] + a1[4 * q + 1]) + (a1[4 * q + 2] + a1[4 * q + 3]));
        const float s2 = half_wave_sum((a2[4 * q] + a2[4 * q + 1]) + (a2[4 * q + 2] + a2[4 * q + 3]));
        if (j == kHalfSumLane) {
          const int idx = r_o * NCH + 32 * rbi + 8 * ((t0 >> 2) + q) + 4 * h;
          cs[2 * idx] = (double)s1;
          cs[2 * idx + 1] = (double)s2;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < TN / 2; ++q) {
        const float s1 = half_wave_sum(a1[2 * q] + a1[2 * q + 1]);
        const float s2 = half_wave_sum(a2[2 * q] + a2[2 * q + 1]);
        if (j == kHalfSumLane) {
          const int tr = t0 + 2 * q;
          const int idx = r_o * NCH + 32 * rbi + (tr & 3) + 8 * (tr >> 2) + 4 * h;
          cs[2 * idx] = (double)s1;
          cs[2 * idx + 1] = (double)s2;
        }
      }
    }
  }

  __device__ __forceinline__ void finish(const f32x2 (&pr)[TN], unsigned char *smem) {
    const bool st1 = gn_wanted(p.fin), st2 = cat && gn_wanted(p.fin2);
    const int tid = threadIdx.x;
    double *cs1 = reinterpret_cast<double *>(smem);             // [2 rows][NCH][2] sums of y (first channel of a quad / pair)
    double *cs2 = reinterpret_cast<double *>(smem + NCH * 32);  // ... of y2
    if (st1) {
      float s1[TN], s2[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        s1[t] = pr[t].x + pr[t].y;
        s2[t] = fmaf(pr[t].y, pr[t].y, pr[t].x * pr[t].x);
      }
      to_lds(cs1, s1, s2, p.fin.c / 32);
    }
    if (cat) {
      float q1[TN], q2[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        u[t] = u[t] + pr[t];
        q1[t] = u[t].x + u[t].y;
        q2[t] = fmaf(u[t].y, u[t].y, u[t].x * u[t].x);
      }
      if (st2) to_lds(cs2, q1, q2, p.fin2.c / 32);
    }
    if (st1 || st2) {
      __syncthreads();
      if (tid < 64) {
        auto fold = [&](const GnOut &f, const double *cs, int c_off) {
          const int cpg = f.c / 32;  // channels per group of the normalised tensor
          const int ng = NCH / cpg;  // groups this workgroup covers
          const int pre = cpg >= 4 ? 4 : 2;
          double a = 0.0, b = 0.0;
          if (tid < ng)
            for (int r = 0; r < 2; ++r)
              for (int ch = 0; ch < cpg; ch += pre) {
                const int idx = r * NCH + tid * cpg + ch;
                a += cs[2 * idx];
                b += cs[2 * idx + 1];
              }
          gn_emit(f, img, (c_off + NCH * (int)blockIdx.y) / cpg, ng, tile, a, b);
        };
        if (st1) fold(p.fin, cs1, 0);
        if (st2) fold(p.fin2, cs2, p.y2_off);
      }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      if (p.y) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pr[t]), rs_y, vo, so1(t), 0);
      if (cat) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, u[t]), rs_y2, vo, so2(t), 0);
    }
  }
};

template <int MRB>
__global__ __launch_bounds__(kWnThreads, 2) void conv3x3_wino_kernel(ConvArgs p) {
  static_assert(MRB == 2, "128 output channels per workgroup");
  constexpr int NCH = 64 * MRB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 31, h = lane >> 5;
  const int jf = wv & 3, rbh = wv >> 2;

  const int tiles_x = p.w / (2 * kWnTX), tiles = tiles_x * (p.h / (2 * kWnTY));
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int y0 = (tile / tiles_x) * (2 * kWnTY), x0 = (tile % tiles_x) * (2 * kWnTX);
  const int hw = p.h * p.w;
  const int n_chunks = p.cin / 16;

  const WStream ws = make_wstream(p.wpw, p.wpw_floats, lane);
  __shared__ float gn_stats[64];
  __shared__ float ss_in[2 * 512];

  // ---- staging plan: wave wv stages channel planes 2 wv, 2 wv + 1 of every chunk; lane = patch pixel ----
  int goff[kWnPasses];  // byte offset inside a channel plane, -1 = outside the image (zero padding) / no pixel
#pragma unroll
  for (int it = 0; it < kWnPasses; ++it) {
    const int lp = lane + 64 * it;
    const int r = lp / kWnPW, c = lp - r * kWnPW;
    int gy = y0 - 1 + r, gx = x0 - 1 + c;
    bool ok = lp < kWnPix;
    if (p.reflect) {  // nn.ReflectionPad2d(1) in front of the convolution (ResBlkFilters.py:28-84): -1 -> 1, H -> H - 2
      gy = gy < 0 ? -gy : (gy >= p.h ? 2 * p.h - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= p.w ? 2 * p.w - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    }
    goff[it] = ok ? (gy * p.w + gx) * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x + (long long)img * p.cin * hw), 0, p.cin * hw * 4, 0x00020000);
  float stg[kWnPasses][2];
  int ch_staged = 0;
  auto stage_load = [&](int chunk) {
    const int ch = min(chunk, n_chunks - 1) * 16 + 2 * wv;
#pragma unroll
    for (int it = 0; it < kWnPasses; ++it) {
      const int o = goff[it] < 0 ? 0 : goff[it];
#pragma unroll
      for (int k = 0; k < 2; ++k)
        stg[it][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, o, (ch + k) * hw * 4, 0));
    }
    ch_staged = ch;
  };
  auto stage_store = [&](int buf) {
    float sc[2], sh[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      sc[k] = ss_in[2 * (ch_staged + k)];
      sh[k] = ss_in[2 * (ch_staged + k) + 1];
    }
    unsigned char *raw = smem + kWnRaw + buf * kWnRawBytes;
#pragma unroll
    for (int it = 0; it < kWnPasses; ++it) {
      const int lp = lane + 64 * it;
      if (lp < kWnPix) {
        f32x2 x2 = {stg[it][0], stg[it][1]};
        f32x2 v = __builtin_elementwise_fma(x2, (f32x2){sc[0], sc[1]}, (f32x2){sh[0], sh[1]});
        if (p.relu) v = __builtin_elementwise_max(v, (f32x2)(0.0f));
        if (goff[it] < 0) v = (f32x2)(0.0f);
        *reinterpret_cast<f32x2 *>(raw + lp * kWnRow + wv * 8) = v;
      }
    }
  };

  // ---- input transform plan: thread = (tile (txx, tyy), channel quad chq, column jt); a 16-lane group = 8 txx x 2 chq ----
  const int txx = lane & 7, tyy = lane >> 4;
  const int jt = wv >> 1, chq = ((wv & 1) << 1) | ((lane >> 3) & 1);
  // t[r] = d[r][ca] + sg * d[r][cb]   (column jt of B):  j = 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
  const int ca = jt == 0 ? 0 : jt == 2 ? 2 : 1;
  const int cb = jt == 0 ? 2 : jt == 1 ? 2 : jt == 2 ? 1 : 3;
  const float sg = jt == 1 ? 1.0f : -1.0f;
  const int off_a = ((2 * tyy) * kWnPW + 2 * txx + ca) * kWnRow + chq * 16;  // row r: + r * kWnPW * kWnRow
  const int off_b = ((2 * tyy) * kWnPW + 2 * txx + cb) * kWnRow + chq * 16;
  const int voff_w = (jt * 32 + tyy * 8 + txx) * kWnRow + chq * 16;  // + i * 4 * 32 * kWnRow
  auto transform = [&](int buf) {
    const unsigned char *raw = smem + kWnRaw + buf * kWnRawBytes;
    unsigned char *v = smem + kWnV + buf * kWnVBytes + voff_w;
    f32x4 t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(raw + off_a + r * kWnPW * kWnRow);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(raw + off_b + r * kWnPW * kWnRow);
      t[r] = __builtin_elementwise_fma(b, (f32x4)sg, a);  // a +- b, exact; v_pk_fma_f32
    }
    // rows of B^T:  i = 0: t0 - t2;  1: t1 + t2;  2: t2 - t1;  3: t1 - t3
    *reinterpret_cast<f32x4 *>(v + 0 * 4 * 32 * kWnRow) = pk_sub4(t[0], t[2]);
    *reinterpret_cast<f32x4 *>(v + 1 * 4 * 32 * kWnRow) = t[1] + t[2];
    *reinterpret_cast<f32x4 *>(v + 2 * 4 * 32 * kWnRow) = pk_sub4(t[2], t[1]);
    *reinterpret_cast<f32x4 *>(v + 3 * 4 * 32 * kWnRow) = pk_sub4(t[1], t[3]);
  };

  // ---- GEMM plan: wave (jf, rbh): frequencies (i, jf), row blocks rbh * MRB + m ----
  const int boff = (jf * 32 + jl) * kWnRow + h * 16;  // + i * 4 * 32 * kWnRow + g * 32
  int a_base[MRB];
#pragma unroll
  for (int m = 0; m < MRB; ++m) {
    const int rb = (int)blockIdx.y * (2 * MRB) + rbh * MRB + m;
    a_base[m] = ((rb * 4 + jf) * n_chunks) * 8 * 64;
  }
  f32x16 acc[4][MRB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int m = 0; m < MRB; ++m)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][m][t] = 0.0f;

  // A ring: the 2 MRB fragments of step s = 4 chunk + i in slot s & 3, loaded kWnAhead steps ahead.  Three steps: a
  // wave's loads return in order, so the wait for a fragment also waits for every staging load issued before it -- an
  // HBM round trip; at three steps (6 k cycles of the SIMD's MFMA work) those have landed
  f32x4 ring[4][MRB][2];
  auto a_load = [&](int slot, int step) {
#pragma unroll
    for (int m = 0; m < MRB; ++m)
#pragma unroll
      for (int g = 0; g < 2; ++g) ring[slot][m][g] = wload128(ws, a_base[m] + step * 128 + g * 64);
  };
  const int n_steps = 4 * n_chunks;

  // ---- prologue ----
  stage_load(0);
  GnAffine affine;
  gn_affine_load(p.gn, p.cin, affine);
  gn_load_stats(p.gn, img, gn_stats);
#pragma unroll
  for (int d = 0; d < kWnAhead; ++d) a_load(d, min(d, n_steps - 1));
  __syncthreads();
  gn_table_fill(p.gn, img, p.cin, gn_stats, affine, ss_in);
  __syncthreads();
  stage_store(0);
  stage_load(1);
  __syncthreads();
  transform(0);
  stage_store(1);
  stage_load(2);
  __syncthreads();

#pragma unroll 1
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char *vb = smem + kWnV + (chunk & 1) * kWnVBytes + boff;
    const bool more1 = chunk + 1 < n_chunks;
    // B of step i + 1 is requested before step i's MFMAs (the two waves of a SIMD run in lock step behind the chunk
    // barrier: an LDS round trip at the top of every step was a bubble for both); staging is unconditional and clamped
    // so that the vector-memory waits are counted exactly
    f32x4 b[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) b[0][g] = *reinterpret_cast<const f32x4 *>(vb + g * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int step = 4 * chunk + i;
      if (i < 3) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
          b[(i + 1) & 1][g] = *reinterpret_cast<const f32x4 *>(vb + (i + 1) * 4 * 32 * kWnRow + g * 32);
      }
      // the two waves of a SIMD (rbh = 0 / 1) transform the next chunk at opposite ends of the iteration: one of them
      // is always in its MFMAs
      if (i == 0 && more1 && rbh == 0) transform((chunk + 1) & 1);
      a_load((i + kWnAhead) & 3, min(step + kWnAhead, n_steps - 1));
      if (i == 1) stage_store(chunk & 1);
      if (i == 1) stage_load(chunk + 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int m = 0; m < MRB; ++m)
            acc[i][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[i][m][g][ii], b[i & 1][g][ii], acc[i][m], 0, 0, 0);
      if (i == 3 && more1 && rbh == 1) transform((chunk + 1) & 1);
    }
    __syncthreads();
  }

  // ---- output transform, rows: S[r] = (A^T M)[r] over i:  r = 0: M0 + M1 + M2;  r = 1: M1 - M2 - M3 ----
  f32x4 *xch = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
  for (int m = 0; m < MRB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 s0, s1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = 4 * q + e;
        s0[e] = (acc[0][m][t] + acc[1][m][t]) + acc[2][m][t];
        s1[e] = (acc[1][m][t] - acc[2][m][t]) - acc[3][m][t];
      }
      xch[(((wv * 2 + 0) * MRB + m) * 4 + q) * 64 + lane] = s0;
      xch[(((wv * 2 + 1) * MRB + m) * 4 + q) * 64 + lane] = s1;
    }
  // ---- columns: Y[r][s] = (S A)[s] over j:  s = 0: S0 + S1 + S2;  s = 1: S1 - S2 - S3.  This wave: output row
  // r_o = jf & 1 of every tile, row block m_o = jf >> 1 of its half ----
  const int r_o = jf & 1, m_o = jf >> 1;
  WinoTail<NCH, 16> tail(p, 0, img, tile, y0, x0, rbh * MRB + m_o, r_o, lane);
  tail.load_res();
  __syncthreads();
  f32x2 pr[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = xch[((((4 * rbh + j) * 2 + r_o) * MRB + m_o) * 4 + q) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pr[4 * q + e].x = (s[0][e] + s[1][e]) + s[2][e];
      pr[4 * q + e].y = (s[1][e] - s[2][e]) - s[3][e];
    }
  }
  tail.finish(pr, smem + kWnStat);
}


// ---- 64 output channels per workgroup, two workgroups per CU --------------------------------------------------
// The kernel above keeps a CU to itself (128 accumulator registers per wave, 139 KB of LDS): its prologue (first
// chunks, GroupNorm table: ~10 k cycles) and epilogue (~15-25 k) run beside nothing, a fixed cost per tile block that a
// K loop of 16 chunks (162 k cycles) carries and one of 4-8 chunks (Cin = 64 / 128) does not.  Here a workgroup takes
// 64 output channels of the same 8 x 4 tiles -- wave (j, half) = frequencies (0..3, j) of ONE row block, 64
// accumulator registers -- and the input in 8-channel chunks, so that two workgroups fit a CU (<= 128 registers, 70 KB):
// one's prologue / epilogue / barriers under the other's MFMAs.  LDS rows are 48 bytes (32 used): 16 lanes of a
// 128-bit access then fall into 16 different 16-byte bank groups.  Serves Cout = 64 (the second and third
// convolution of every pyramid block) and, with two workgroups per tile block, Cout = 128.
constexpr int kW8Row = 48;                        // bytes per LDS row of 8 channels (padded)
constexpr int kW8RawBytes = kWnPix * kW8Row;      // 8640
constexpr int kW8VBytes = 16 * 32 * kW8Row;       // 24576: [i][j][tile][8 ch]
constexpr int kW8Raw = 0;
constexpr int kW8V = 2 * kW8RawBytes;
constexpr int kW8XchBytes = 8 * 2 * 4 * 64 * 16;  // [wave][r][q][lane] f32x4 (64 KB)
constexpr int kW8Stat = kW8V + 2 * kW8VBytes > kW8XchBytes ? kW8V + 2 * kW8VBytes : kW8XchBytes;
constexpr int kW8Lds = kW8Stat + 4096;
static_assert(2 * (kW8Lds + 4608) <= 160 * 1024, "two workgroups per CU");

__global__ __launch_bounds__(kWnThreads, 4) void conv3x3_wino64_kernel(ConvArgs p) {
  constexpr int NCH = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = lane & 31, h = lane >> 5;
  const int jf = wv & 3, rbh = wv >> 2;

  const int tiles_x = p.w / (2 * kWnTX), tiles = tiles_x * (p.h / (2 * kWnTY));
  const int tile = blockIdx.x % tiles, img = blockIdx.x / tiles;
  const int y0 = (tile / tiles_x) * (2 * kWnTY), x0 = (tile % tiles_x) * (2 * kWnTX);
  const int hw = p.h * p.w;
  const int n_chunks = p.cin / 8;

  const WStream ws = make_wstream(p.wpw, p.wpw_floats, lane);
  __shared__ float gn_stats[64];
  __shared__ float ss_in[2 * 512];

  // ---- staging: wave wv stages channel plane 8 chunk + wv; lane = patch pixel ----
  int goff[kWnPasses];
#pragma unroll
  for (int it = 0; it < kWnPasses; ++it) {
    const int lp = lane + 64 * it;
    const int r = lp / kWnPW, c = lp - r * kWnPW;
    int gy = y0 - 1 + r, gx = x0 - 1 + c;
    bool ok = lp < kWnPix;
    if (p.reflect) {  // nn.ReflectionPad2d(1) in front of the convolution (ResBlkFilters.py:28-84): -1 -> 1, H -> H - 2
      gy = gy < 0 ? -gy : (gy >= p.h ? 2 * p.h - 2 - gy : gy);
      gx = gx < 0 ? -gx : (gx >= p.w ? 2 * p.w - 2 - gx : gx);
    } else {
      ok = ok && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    }
    goff[it] = ok ? (gy * p.w + gx) * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.x + (long long)img * p.cin * hw), 0, p.cin * hw * 4, 0x00020000);
  float stg[kWnPasses];
  int ch_staged = 0;
  auto stage_load = [&](int chunk) {
    const int ch = min(chunk, n_chunks - 1) * 8 + wv;
#pragma unroll
    for (int it = 0; it < kWnPasses; ++it)
      stg[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, goff[it] < 0 ? 0 : goff[it], ch * hw * 4, 0));
    ch_staged = ch;
  };
  auto stage_store = [&](int buf) {
    const float sc = ss_in[2 * ch_staged], sh = ss_in[2 * ch_staged + 1];
    unsigned char *raw = smem + kW8Raw + buf * kW8RawBytes + wv * 4;
#pragma unroll
    for (int it = 0; it < kWnPasses; ++it) {
      const int lp = lane + 64 * it;
      if (lp < kWnPix) {
        float t = fmaf(stg[it], sc, sh);
        if (p.relu) t = fmaxf(t, 0.0f);
        *reinterpret_cast<float *>(raw + lp * kW8Row) = goff[it] < 0 ? 0.0f : t;
      }
    }
  };

  // ---- input transform: thread = (tile (txx, tyy), channel quad chq) x wave = (column jt, row pair ih) ----
  const int txx = lane & 7, chq = (lane >> 3) & 1, tyy = lane >> 4;
  const int jt = wv >> 1, ih = wv & 1;
  const int ca = jt == 0 ? 0 : jt == 2 ? 2 : 1;
  const int cb = jt == 0 ? 2 : jt == 1 ? 2 : jt == 2 ? 1 : 3;
  const float sg = jt == 1 ? 1.0f : -1.0f;
  // rows ih, ih + 1, ih + 2 of the patch:  ih = 0 -> V0 = t0 - t2, V1 = t1 + t2;  ih = 1 -> V2 = t2 - t1, V3 = t1 - t3
  const int off_a = ((2 * tyy + ih) * kWnPW + 2 * txx + ca) * kW8Row + chq * 16;
  const int off_b = ((2 * tyy + ih) * kWnPW + 2 * txx + cb) * kW8Row + chq * 16;
  const int voff_w = (((2 * ih) * 4 + jt) * 32 + tyy * 8 + txx) * kW8Row + chq * 16;  // second row: + 4 * 32 * kW8Row
  auto transform = [&](int buf) {
    const unsigned char *raw = smem + kW8Raw + buf * kW8RawBytes;
    unsigned char *v = smem + kW8V + buf * kW8VBytes + voff_w;
    f32x4 t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(raw + off_a + r * kWnPW * kW8Row);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(raw + off_b + r * kWnPW * kW8Row);
      t[r] = __builtin_elementwise_fma(b, (f32x4)sg, a);  // a +- b, exact
    }
    f32x4 lo, hi;
    if (ih == 0) {
      lo = pk_sub4(t[0], t[2]);
      hi = t[1] + t[2];
    } else {
      lo = pk_sub4(t[1], t[0]);
      hi = pk_sub4(t[0], t[2]);
    }
    *reinterpret_cast<f32x4 *>(v) = lo;
    *reinterpret_cast<f32x4 *>(v + 4 * 32 * kW8Row) = hi;
  };

  // ---- GEMM: wave (jf, rbh): frequencies (i, jf) of row block 2 blockIdx.y + rbh ----
  const int boff = (jf * 32 + jl) * kW8Row + h * 16;  // + i * 4 * 32 * kW8Row
  const int a_base = ((((int)blockIdx.y * 2 + rbh) * 4 + jf) * (p.cin / 16)) * 8 * 64;
  // fragment of (chunk c of 8 channels, frequency row i): a_base + ((c >> 1) * 8 + i * 2 + (c & 1)) * 64
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[i][t] = 0.0f;
  f32x4 ring[4];  // the chunk's four fragments; pair (0, 1) of the next chunk is requested while pair (2, 3) runs
  auto a_load = [&](int i, int c) {
    const int cc = min(c, n_chunks - 1);
    ring[i] = wload128(ws, a_base + ((cc >> 1) * 8 + i * 2 + (cc & 1)) * 64);
  };

  // ---- prologue ----
  stage_load(0);
  GnAffine affine;
  gn_affine_load(p.gn, p.cin, affine);
  gn_load_stats(p.gn, img, gn_stats);
  a_load(0, 0);
  a_load(1, 0);
  __syncthreads();
  gn_table_fill(p.gn, img, p.cin, gn_stats, affine, ss_in);
  __syncthreads();
  stage_store(0);
  stage_load(1);
  __syncthreads();
  transform(0);
  stage_store(1);
  stage_load(2);
  __syncthreads();

#pragma unroll 1
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char *vb = smem + kW8V + (chunk & 1) * kW8VBytes + boff;
    const bool more1 = chunk + 1 < n_chunks;
    // Vector-memory loads return in order, so a wait for a weight fragment also waits for every load issued before it.
    // Per chunk: [top] fragments 2, 3 | pair 0's MFMAs | fragments 0, 1 of the next chunk, then the staging loads of
    // chunk + 3 (unconditional and clamped: the waits are counted exactly) | pair 1's MFMAs.  The wait for fragments
    // 0, 1 leaves the staging loads in flight; they have a whole chunk until fragments 2, 3 of the next chunk are needed.
    f32x4 b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f32x4 *>(vb + i * 4 * 32 * kW8Row);
    a_load(2, chunk);
    a_load(3, chunk);
    if (more1 && rbh == 0) transform((chunk + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[i][ii], b[i][ii], acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f32x4 *>(vb + (i + 2) * 4 * 32 * kW8Row);
    a_load(0, chunk + 1);
    a_load(1, chunk + 1);
    stage_store(chunk & 1);  // chunk + 2 (past the end: the clamped last chunk again, into a buffer nobody reads)
    stage_load(chunk + 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int i = 2; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[i][ii], b[i - 2][ii], acc[i], 0, 0, 0);
    if (more1 && rbh == 1) transform((chunk + 1) & 1);
    __syncthreads();
  }

  // ---- output transform, rows ----
  f32x4 *xch = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 s0, s1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = 4 * q + e;
      s0[e] = (acc[0][t] + acc[1][t]) + acc[2][t];
      s1[e] = (acc[1][t] - acc[2][t]) - acc[3][t];
    }
    xch[((wv * 2 + 0) * 4 + q) * 64 + lane] = s0;
    xch[((wv * 2 + 1) * 4 + q) * 64 + lane] = s1;
  }
  // ---- columns: this wave = output row r_o = jf & 1 of every tile, registers 8 qh .. 8 qh + 7 (qh = jf >> 1) ----
  const int r_o = jf & 1, qh = jf >> 1;
  WinoTail<NCH, 8> tail(p, 8 * qh, img, tile, y0, x0, rbh, r_o, lane);
  tail.load_res();
  __syncthreads();
  f32x2 pr[8];
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = xch[(((4 * rbh + j) * 2 + r_o) * 4 + 2 * qh + qq) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pr[4 * qq + e].x = (s[0][e] + s[1][e]) + s[2][e];
      pr[4 * qq + e].y = (s[1][e] - s[2][e]) - s[3][e];
    }
  }
  tail.finish(pr, smem + kW8Stat);
}

int launch_conv3x3_wino_pack(mp_ctx *ctx, const float *w, int cout, int cin, float *up, hipStream_t st) {
  const long long total = 16LL * cout * cin;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv3x3_wino_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cout, cin, up);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

bool conv3x3_wino_supported(int cin, int cout, int h, int w) {
  return cin % 16 == 0 && cin >= 16 && cin <= 512 && cout % 64 == 0 && h % (2 * kWnTY) == 0 && w % (2 * kWnTX) == 0;
}

int conv3x3_wino_tiles(int h, int w) { return (h / (2 * kWnTY)) * (w / (2 * kWnTX)); }

// workgroups of the launch: 128 output channels each (one per CU) when Cout allows and variant != 64, else 64 (two per CU)
static int g_wino_variant = 0;  // 0 = heuristic, 64 / 128 forced (measurement hook: mp_conv3x3_tune(0x800 / 0x1000))
void conv3x3_wino_set_variant(int v) { g_wino_variant = v; }
bool conv3x3_wino_forced() { return g_wino_variant != 0; }
static bool wino_use64(const ConvArgs &a) {
  if (a.cout % 128) return true;
  if (g_wino_variant) return g_wino_variant == 64;
  // 128-channel workgroups (a CU each) once they come in many rounds; below that the 64-channel kernel's two
  // workgroups per CU win (profiles/r06s_wino_check.txt: 256 -> 128 at 64^2 x 20 = 640 workgroups: 233 vs 265 us; at
  // 128^2 x 1 = 128: 56 vs 85 us; at 128^2 x 20 = 2560: 904 vs 871 us)
  return (long long)conv3x3_wino_tiles(a.h, a.w) * a.n_img * (a.cout / 128) < 2048;
}
long long conv3x3_wino_workgroups(const ConvArgs &a) {
  return (long long)conv3x3_wino_tiles(a.h, a.w) * a.n_img * (a.cout / (wino_use64(a) ? 64 : 128));
}

int launch_conv3x3_wino(mp_ctx *ctx, const ConvArgs &a, hipStream_t st) {
  const int tiles = conv3x3_wino_tiles(a.h, a.w);
  if (wino_use64(a)) {
    auto kern = conv3x3_wino64_kernel;
    const void *kern_id = reinterpret_cast<const void *>(kern);
    if (!ctx->lds_attr_done.count(kern_id)) {
      MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, kW8Lds));
      ctx->lds_attr_done.insert(kern_id);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / 64)), dim3(kWnThreads), kW8Lds, st, a);
  } else {
    auto kern = conv3x3_wino_kernel<2>;
    const void *kern_id = reinterpret_cast<const void *>(kern);
    if (!ctx->lds_attr_done.count(kern_id)) {
      MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, kWnLds));
      ctx->lds_attr_done.insert(kern_id);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.n_img), (unsigned)(a.cout / 128)), dim3(kWnThreads), kWnLds, st, a);
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
