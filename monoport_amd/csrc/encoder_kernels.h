// Argument blocks and launchers of the encoder kernels (csrc/conv3x3.hip, convim2col.hip,
// encoder_ops.hip) shared with the C-ABI layer (api.hip).
#pragma once
#include "mp_internal.h"

namespace mp {

// Producer side of a GroupNorm hand-over (csrc/gn_tail.h): where the statistics of the tensor a
// kernel writes go.  Both pointers may be set; both nullptr = no statistics.
constexpr int kGnReplicas = 16;  // copies of an accumulator the producers' workgroups spread over (gn_tail.h)

struct GnOut {
  long long *acc;   // [kGnReplicas][N][32][4] fixed-point accumulator, ZERO before the launch
  double *partial;  // legacy: [N][32][S][2] plain partial sums, finalised by mp_gn_finalize
  int c;            // channels of the normalised tensor (groups of c / 32 adjacent channels)
  int S;            // slots per (image, group) of `partial`
  int n;            // images (stride between replicas)
};

__host__ __device__ inline GnOut gn_out_none() { return GnOut{nullptr, nullptr, 32, 0, 1}; }
__host__ __device__ inline bool gn_wanted(const GnOut &f) { return f.acc || f.partial; }

// Consumer side: the GroupNorm applied to a kernel's INPUT while it is staged.
struct GnIn {
  const long long *acc;        // accumulator its producer filled, or nullptr
  const float *gamma, *beta;   // [C] affine parameters (with acc)
  const float *ss;             // legacy: precomputed [N][C][2] (scale, shift), or nullptr
  float eps;
  int c;                       // channels of the normalised tensor
  int n;                       // images (stride between replicas)
  double count;                // elements per group = (c / 32) * H * W
};

__host__ __device__ inline GnIn gn_in_none() { return GnIn{nullptr, nullptr, nullptr, nullptr, 1e-5f, 32, 1, 1.0}; }
__host__ __device__ inline bool gn_active(const GnIn &g) { return g.acc || g.ss; }

struct ConvArgs {
  const float *x;    // [N, Cin, H, W]
  GnIn gn;           // GroupNorm(32, Cin) of the input, fused into the staging (gn_in_none(): plain input)
  const float *wp;   // packed weights [Cout/32][Cin/16 * 18][64][4]
  float *y;          // [N, Cout, H, W], or nullptr when only y2 is wanted
  // pyramid-block tail fused into the epilogue (HGFilters.py:57-60: cat((out1, out2, out3), 1) + residual):
  // y2[n, y2_off + c] = conv[n, c] + res[n, y2_off + c], both [N, y2_c, H, W]
  float *y2;
  const float *res;
  int y2_c, y2_off;
  GnOut fin;         // GroupNorm(32, Cout) statistics of the raw output y (S = tiles per image)
  GnOut fin2;        // GroupNorm(32, y2_c) statistics of this launch's channels of y2
  int n_img, cin, cout, h, w;
  int tw, th;        // tile width / height in pixels (th * tw = 32 * NR * CW)
  int relu;          // apply ReLU to the (normalised) input
  int reflect;       // 0: zero padding (HGFilters.py ConvBlock); 1: nn.ReflectionPad2d(1) in front of
                     // the convolution (ResBlkFilters.py:28-84): halo pixels mirror the interior
  int wp_floats;     // size of wp
  // conv_wino.hip: the Winograd-domain weights U = G g G^T (mp_conv3x3_pack_wino, 16 Cout Cin floats), or nullptr:
  // launch_conv3x3 takes the F(2x2, 3x3) kernel for the shapes it serves when they are given
  const float *wpw = nullptr;
  int wpw_floats = 0;
};

struct Conv1Args {
  const float *x1;        // [N,C1,HW]
  GnIn gn1;               // GroupNorm(32, C1) of x1, fused into the staging
  const float *x2;        // [N,C2,HW] or nullptr (plain second K segment)
  const float *wp;        // packed [8 row blocks][K/8 groups][64][4] f32, or [8][K/16][hi|lo][64] h8
  const float *bias;      // [256]
  const float *res;       // [N,256,HW] or nullptr
  float *y;               // [N,256,HW] or nullptr
  float *y_hwc;           // [N,HW,256] or nullptr
  GnOut fin;              // GroupNorm(32, Cout) statistics of the output (S = HW / 64), Cout = 256 only
  int n_img, c1, c2, hw, relu1, wp_floats;
  int cout;  // 256 (each wave two 32-row blocks) or 128 (one): the 1x1 projection of a pyramid block
};


// conv_wino.hip: Winograd F(2x2, 3x3) for the 128-channel-block shapes
bool conv3x3_wino_supported(int cin, int cout, int h, int w);
int conv3x3_wino_tiles(int h, int w);
long long conv3x3_wino_workgroups(const ConvArgs &a);
void conv3x3_wino_set_variant(int v);
bool conv3x3_wino_forced();  // a variant is forced (mp_conv3x3_tune): the size threshold does not apply
int launch_conv3x3_wino_pack(mp_ctx *ctx, const float *w, int cout, int cin, float *up, hipStream_t st);
int launch_conv3x3_wino(mp_ctx *ctx, const ConvArgs &a, hipStream_t st);
// convim2col.hip: 7x7 (3 -> 64, stride 1 / 2) and 3x3 stride-2 convolutions
struct ConvKArgs {
  const float *x;     // [N, Cin, H, W]
  GnIn gn;            // GroupNorm(32, Cin) of the input, applied while gathering
  const float *wp;    // packed by convk_pack_kernel
  const float *bias;  // [Cout] or nullptr
  float *y;           // [N, Cout, H / stride, W / stride]
  GnOut fin;          // GroupNorm(32, Cout) statistics of y (S = output rows * output width / 64)
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
int launch_avgpool2_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, float *y, GnOut fin,
                       long long partial_cap, hipStream_t st);
int launch_upsample_add_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, const float *add, float *y,
                           GnOut fin, long long partial_cap, hipStream_t st);
int launch_gn_apply_gn(mp_ctx *ctx, const float *x, GnIn gn, int relu, int n, int c, long long hw,
                       const float *res, float *y, GnOut fin, long long partial_cap, hipStream_t st);

}  // namespace mp
