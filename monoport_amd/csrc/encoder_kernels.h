// Argument blocks and launchers of the encoder kernels (csrc/conv3x3.hip, convim2col.hip,
// encoder_ops.hip) shared with the C-ABI layer (api.hip).
#pragma once
#include "mp_internal.h"

namespace mp {

struct GnSet {
  const float *gamma, *beta;  // [C] affine parameters of the consuming GroupNorm
  float *ss;                  // out [N][C][2] = (gamma rstd, beta - mean gamma rstd)
  float eps;
};

struct GnFin {
  double *partial;  // [N][32][S][2] partial (sum, sum of squares); nullptr = no statistics
  int *counter;     // arrival counters, zero before and after every launch (n_sets > 0)
  GnSet set[2];
  int n_sets;       // 0: partial sums only (mp_gn_finalize turns them into ss); 1-2: consumers
  int c;            // channels of the normalised tensor (groups of c / 32 adjacent channels)
  int S;            // slots per (image, group)
  double count;     // elements per group = (c / 32) * H * W
};

inline GnFin gn_fin_none() {
  GnFin f;
  f.partial = nullptr;
  f.counter = nullptr;
  f.n_sets = 0;
  f.c = 32;
  f.S = 0;
  f.count = 1.0;
  for (int k = 0; k < 2; ++k) f.set[k] = GnSet{nullptr, nullptr, nullptr, 1e-5f};
  return f;
}

struct ConvArgs {
  const float *x;    // [N, Cin, H, W]
  const float *ss;   // [N, Cin, 2] (scale, shift) of the fused GroupNorm, or nullptr: plain input
  const float *wp;   // packed weights [Cout/32][Cin/16 * 18][64][4]
  float *y;          // [N, Cout, H, W], or nullptr when only y2 is wanted
  // pyramid-block tail fused into the epilogue (HGFilters.py:57-60: cat((out1, out2, out3), 1) + residual):
  // y2[n, y2_off + c] = conv[n, c] + res[n, y2_off + c], both [N, y2_c, H, W]
  float *y2;
  const float *res;
  int y2_c, y2_off;
  GnFin fin;         // GroupNorm(32, Cout) statistics of the raw output y (S = tiles per image)
  GnFin fin2;        // GroupNorm(32, y2_c) statistics of this launch's channels of y2
  int n_img, cin, cout, h, w;
  int tw, th;        // tile width / height in pixels (th * tw = 32 * NR * CW)
  int relu;          // apply ReLU to the (normalised) input
  int reflect;       // 0: zero padding (HGFilters.py ConvBlock); 1: nn.ReflectionPad2d(1) in front of
                     // the convolution (ResBlkFilters.py:28-84): halo pixels mirror the interior
  int wp_floats;     // size of wp
};

struct Conv1Args {
  const float *x1, *ss1;  // [N,C1,HW], [N,C1,2] or nullptr
  const float *x2;        // [N,C2,HW] or nullptr (plain second K segment)
  const float *wp;        // packed [8 row blocks][K/8 groups][64][4] f32, or [8][K/16][hi|lo][64] h8
  const float *bias;      // [256]
  const float *res;       // [N,256,HW] or nullptr
  float *y;               // [N,256,HW] or nullptr
  float *y_hwc;           // [N,HW,256] or nullptr
  GnFin fin;              // GroupNorm(32, Cout) statistics of the output (S = HW / 64), Cout = 256 only
  int n_img, c1, c2, hw, relu1, wp_floats;
  int cout;  // 256 (each wave two 32-row blocks) or 128 (one): the 1x1 projection of a pyramid block
};


// convim2col.hip: 7x7 (3 -> 64, stride 1 / 2) and 3x3 stride-2 convolutions
struct ConvKArgs {
  const float *x;     // [N, Cin, H, W]
  const float *ss;    // [N, Cin, 2] fused GroupNorm of the input, or nullptr
  const float *wp;    // packed by convk_pack_kernel
  const float *bias;  // [Cout] or nullptr
  float *y;           // [N, Cout, H / stride, W / stride]
  GnFin fin;          // GroupNorm(32, Cout) statistics of y (S = output rows * output width / 64)
  int n_img, cin, cout, h, w, ho, wo;
  int ks, stride, pad, reflect, relu;
  int wp_floats;
};

// conv3x3.hip
int launch_conv3x3(mp_ctx *ctx, ConvArgs a, const float *wmax16, const long long partial_cap[2], hipStream_t st);
int launch_conv1x1(mp_ctx *ctx, Conv1Args a, int f16, const float *wmax, long long partial_cap, hipStream_t st);
int conv1x1_stat_slices(long long hw);
void conv1x1_set_mrw(int mrw);
// convim2col.hip
bool convk_supported(int cin, int cout, int ks, int stride, int h, int w);
long long convk_packed_floats(int cin, int cout, int ks);
int convk_stat_slices(int ks, int stride, int h, int w);
int launch_convk_pack(mp_ctx *ctx, const float *w, int cout, int cin, int ks, float *wp, hipStream_t st);
int launch_convk(mp_ctx *ctx, ConvKArgs a, long long partial_cap, hipStream_t st);
// encoder_ops.hip: elementwise producers that publish the GroupNorm statistics of their output
int launch_avgpool2_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, float *y, GnFin fin,
                       long long partial_cap, hipStream_t st);
int launch_upsample_add_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, const float *add, float *y,
                           GnFin fin, long long partial_cap, hipStream_t st);
int launch_gn_apply_gn(mp_ctx *ctx, const float *x, const float *ss, int relu, int n, int c, long long hw,
                       float *y, GnFin fin, long long partial_cap, hipStream_t st);

}  // namespace mp
