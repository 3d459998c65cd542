// Memory-bound helper kernels for the image encoders: everything between
// the convolutions.  Round 1 (the encoders still on MIOpen convolutions) replaced what PyTorch does
// badly at batch 1 on a 256-CU part; round 3 added the producers that hand their statistics on
// (ew_gn_kernel below), with which the encoders no longer call MIOpen or torch ops at all:
//   * GroupNorm(32, C) (+ReLU): torch launches RowwiseMoments on 32 workgroups (one per group,
//     12 % of the CUs), a parameter kernel, an affine kernel and a ReLU kernel -- 4.6 ms / frame.
//     Here: a split reduction over (group, slice) workgroups + one fused normalise/affine/ReLU pass.
//   * bicubic x2 upsample (align_corners=True) + the hourglass skip add (HGFilters.py:108-111):
//     one pass, 16 taps from L1/L2, instead of upsample + add.
// Both are HBM-bound: bytes moved = 3x (GroupNorm: two reads + one write) resp. ~2.5x
// (upsample-add) the tensor size; measured against the ~6.3 TB/s achievable HBM rate.
#include "mp_internal.h"
#include "gn_tail.h"

namespace mp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGnSlices = 16;   // slices per group for the split reduction
constexpr int kGnThreads = 256;

// partial[(g * kGnSlices + s) * 2 + {0,1}] = sum, sum of squares (double) of slice s of group g
__global__ __launch_bounds__(kGnThreads) void gn_partial_kernel(const float *__restrict__ x,
                                                                long long group_elems,
                                                                double *__restrict__ partial) {
  const int g = blockIdx.x / kGnSlices, s = blockIdx.x % kGnSlices;
  const long long per = (group_elems / 4 + kGnSlices - 1) / kGnSlices;  // float4 per slice
  const long long v0 = s * per, v1 = min(v0 + per, group_elems / 4);
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(x + g * group_elems);
  float s1 = 0.f, s2 = 0.f;  // per-thread f32 partials over <= a few hundred values
  double d1 = 0.0, d2 = 0.0;
  int k = 0;
  for (long long i = v0 + threadIdx.x; i < v1; i += kGnThreads) {
    const f32x4 v = xp[i];
    s1 += (v[0] + v[1]) + (v[2] + v[3]);
    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    if (++k == 64) {  // flush to double regularly: keeps the f32 running sums short
      d1 += s1;
      d2 += s2;
      s1 = s2 = 0.f;
      k = 0;
    }
  }
  d1 += s1;
  d2 += s2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_down(d1, o);
    d2 += __shfl_down(d2, o);
  }
  __shared__ double w1[kGnThreads / 64], w2[kGnThreads / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    w1[wv] = d1;
    w2[wv] = d2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int i = 0; i < kGnThreads / 64; ++i) {
      a += w1[i];
      b += w2[i];
    }
    partial[2 * blockIdx.x] = a;
    partial[2 * blockIdx.x + 1] = b;
  }
}

// y = relu?((x - mean) * rstd * gamma[c] + beta[c]); one workgroup per (group, slice)
__global__ __launch_bounds__(kGnThreads) void gn_apply_kernel(
    const float *__restrict__ x, long long group_elems, int ch_per_group, long long hw,
    int groups_per_image, const double *__restrict__ partial, const float *__restrict__ gamma,
    const float *__restrict__ beta, float eps, int relu, float *__restrict__ y) {
  const int g = blockIdx.x / kGnSlices, s = blockIdx.x % kGnSlices;  // g runs over images x groups
  double a = 0, b = 0;
#pragma unroll
  for (int i = 0; i < kGnSlices; ++i) {
    a += partial[2 * (g * kGnSlices + i)];
    b += partial[2 * (g * kGnSlices + i) + 1];
  }
  const double mean_d = a / (double)group_elems;
  const double var_d = fmax(b / (double)group_elems - mean_d * mean_d, 0.0);
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var_d + (double)eps));
  const long long per = (group_elems / 4 + kGnSlices - 1) / kGnSlices;
  const long long v0 = s * per, v1 = min(v0 + per, group_elems / 4);
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(x + g * group_elems);
  f32x4 *yp = reinterpret_cast<f32x4 *>(y + g * group_elems);
  for (long long i = v0 + threadIdx.x; i < v1; i += kGnThreads) {
    // hw % 4 == 0: one channel per float4; gamma / beta repeat per image
    const int c = (g % groups_per_image) * ch_per_group + (int)((4 * i) / hw);
    const float sc = rstd * gamma[c];
    const float sh = beta[c] - mean * sc;
    f32x4 v = xp[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o = v[k] * sc + sh;
      v[k] = (relu && o < 0.f) ? 0.f : o;
    }
    yp[i] = v;
  }
}

int gn_stat_slices() { return kGnSlices; }

int launch_gn_stats(mp_ctx *ctx, const float *x, int n, int c, long long hw, int groups,
                    double *partial, hipStream_t st) {
  const long long ge = (long long)(c / groups) * hw;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(n * groups * kGnSlices), dim3(kGnThreads), 0, st, x, ge,
                     partial);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

size_t gn_scratch_bytes(int groups) { return (size_t)groups * kGnSlices * 2 * sizeof(double); }

int launch_group_norm(mp_ctx *ctx, void *scratch, const float *x, int n, int c, long long hw,
                      int groups, const float *gamma, const float *beta, float eps, int relu,
                      float *y, hipStream_t st) {
  const int cpg = c / groups;
  const long long ge = (long long)cpg * hw;
  double *partial = static_cast<double *>(scratch);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(n * groups * kGnSlices), dim3(kGnThreads), 0, st, x, ge,
                     partial);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(n * groups * kGnSlices), dim3(kGnThreads), 0, st, x, ge,
                     cpg, hw, groups, partial, gamma, beta, eps, relu, y);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- bicubic x2, align_corners=True, optional fused add ------------------------------------------
// torch.nn.functional.interpolate(mode="bicubic", align_corners=True): source coordinate
// = dst * (in-1)/(out-1); cubic convolution with A = -0.75; taps clamped to the border.
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// one output pixel (oy, ox) of plane `src` [h, w]
__device__ __forceinline__ float bicubic2x_at(const float *__restrict__ src, int h, int w, float sy, float sx,
                                              int oy, int ox) {
  const float ry = sy * oy, rx = sx * ox;
  const float fy = floorf(ry), fx = floorf(rx);
  const int iy = (int)fy, ix = (int)fx;
  float wy[4], wx[4];
  cubic_coeffs(ry - fy, wy);
  cubic_coeffs(rx - fx, wx);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int yy = min(max(iy - 1 + j, 0), h - 1);
    float row = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int xx = min(max(ix - 1 + i, 0), w - 1);
      row += src[yy * w + xx] * wx[i];
    }
    acc += row * wy[j];
  }
  return acc;
}

__global__ __launch_bounds__(256) void upsample_bicubic2x_kernel(const float *__restrict__ x, int c,
                                                                 int h, int w,
                                                                 const float *__restrict__ add,
                                                                 float *__restrict__ y) {
  const int ho = 2 * h, wo = 2 * w;
  const long long total = (long long)c * ho * wo;
  const float sy = (float)(h - 1) / (float)(ho - 1), sx = (float)(w - 1) / (float)(wo - 1);
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % wo), oy = (int)((t / wo) % ho);
    const long long ch = t / ((long long)wo * ho);
    const float acc = bicubic2x_at(x + ch * (long long)h * w, h, w, sy, sx, oy, ox);
    y[t] = add ? add[t] + acc : acc;
  }
}

int launch_upsample_bicubic2x(mp_ctx *ctx, const float *x, int c, int h, int w, const float *add,
                              float *y, hipStream_t st) {
  const long long total = (long long)c * 4 * h * w;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(upsample_bicubic2x_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, c, h, w,
                     add, y);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- elementwise producers that take the GroupNorm statistics of what they write ----------------
// The tensors between the convolutions (2x2 average pool, HGFilters.py:93 / :171; bicubic x2 + skip
// add, :108-111; the stem's GroupNorm + ReLU, :168) feed a GroupNorm(32, C) next.  One pass writes
// the tensor AND adds its statistics to the consumer's accumulator (gn_tail.h): workgroup = (image,
// group, slice) -- a group is C / 32 adjacent planes, contiguous in NCHW -- with one set of integer
// atomics per workgroup: no statistics pass, no finalize launch.
// Op::run(gi, i) computes and stores the 4 consecutive outputs at float4 index i of (image, group)
// gi (hw % 4 == 0: they share a plane) and returns them.
template <class Op>
__global__ __launch_bounds__(kGnThreads) void ew_gn_kernel(Op op, long long group_elems, GnOut fin) {
  __shared__ double w1[kGnThreads / 64], w2[kGnThreads / 64];
  constexpr int V = Op::kVec;  // outputs per thread and step: 4 (one float4) or 1
  const int gi = blockIdx.x / kGnSlices, s = blockIdx.x % kGnSlices;
  const long long per = (group_elems / V + kGnSlices - 1) / kGnSlices;  // steps per slice
  const long long v0 = s * per, v1 = min(v0 + per, group_elems / V);
  op.begin(gi);
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
  int k = 0;
  for (long long i = v0 + threadIdx.x; i < v1; i += kGnThreads) {
    if constexpr (V == 4) {
      const f32x4 v = op.run(gi, i, group_elems);
      s1 += (v[0] + v[1]) + (v[2] + v[3]);
      s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    } else {
      const float v = op.run(gi, i, group_elems);
      s1 += v;
      s2 += v * v;
    }
    if (++k == 64) {
      d1 += s1;
      d2 += s2;
      s1 = s2 = 0.f;
      k = 0;
    }
  }
  if (!gn_wanted(fin)) return;
  d1 += s1;
  d2 += s2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_down(d1, o);
    d2 += __shfl_down(d2, o);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    w1[wv] = d1;
    w2[wv] = d2;
  }
  __syncthreads();
  double a = 0, b = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kGnThreads / 64; ++i) {
      a += w1[i];
      b += w2[i];
    }
  gn_emit(fin, gi / 32, gi % 32, 1, s, a, b);
}

struct AvgPool2Op {  // y [N*C, H/2, W/2] = avg_pool2d(x [N*C, H, W], 2, stride 2)
  static constexpr int kVec = 4;
  const float *x;
  float *y;
  int ho, wo;  // output size; wo % 4 == 0
  __device__ __forceinline__ void begin(int) {}
  __device__ __forceinline__ f32x4 run(int gi, long long i, long long group_elems) const {
    const long long e = gi * group_elems + 4 * i;  // first output element
    const long long plane = e / ((long long)ho * wo);
    const int rem = (int)(e - plane * ho * wo);
    const int oy = rem / wo, ox = rem - oy * wo;
    const float *r0 = x + (plane * 2 * ho + 2 * oy) * (2LL * wo) + 2 * ox;
    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(r0), a1 = *reinterpret_cast<const f32x4 *>(r0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(r0 + 2 * wo), b1 = *reinterpret_cast<const f32x4 *>(r0 + 2 * wo + 4);
    f32x4 v;  // torch's order: the window row by row, then the division
    v[0] = (((a0[0] + a0[1]) + b0[0]) + b0[1]) * 0.25f;
    v[1] = (((a0[2] + a0[3]) + b0[2]) + b0[3]) * 0.25f;
    v[2] = (((a1[0] + a1[1]) + b1[0]) + b1[1]) * 0.25f;
    v[3] = (((a1[2] + a1[3]) + b1[2]) + b1[3]) * 0.25f;
    *reinterpret_cast<f32x4 *>(y + e) = v;
    return v;
  }
};

struct UpsampleAddOp {  // y [N*C, 2H, 2W] = add + bicubic_x2(x [N*C, H, W])
  // one output per thread: 16 taps each, and four pixels per thread (64 loads in flight per thread,
  // 126 VGPRs) measured 1.5x slower than upsample_bicubic2x_kernel's one
  static constexpr int kVec = 1;
  const float *x, *add;
  float *y;
  int h, w;  // input size
  float sy, sx;
  __device__ __forceinline__ void begin(int) {}
  __device__ __forceinline__ float run(int gi, long long i, long long group_elems) const {
    const int ho = 2 * h, wo = 2 * w;
    const long long e = gi * group_elems + i;
    const long long plane = e / ((long long)ho * wo);
    const int rem = (int)(e - plane * ho * wo);
    const int oy = rem / wo, ox = rem - oy * wo;
    const float up = bicubic2x_at(x + plane * (long long)h * w, h, w, sy, sx, oy, ox);
    const float v = add ? add[e] + up : up;
    y[e] = v;
    return v;
  }
};

struct GnApplyOp {  // y = [res +] relu?(GroupNorm(x)), x / y / res [N*C, HW]; the GroupNorm as GnIn
  static constexpr int kVec = 4;
  const float *x, *res;
  float *y;
  long long hw;
  int relu;
  GnIn gn;
  float mean, rstd;  // of this workgroup's group (hand-over mode)
  __device__ __forceinline__ void begin(int gi) {
    mean = 0.0f;
    rstd = 1.0f;
    if (gn.acc) {  // one group per workgroup: lanes 0..R-1 fetch one replica each, wave 0 adds them up
      __shared__ float mr[2];
      if (threadIdx.x < 64) {
        unsigned long long w[4] = {0, 0, 0, 0};
        if (threadIdx.x < kGnReplicas) {
          const unsigned long long *a = reinterpret_cast<const unsigned long long *>(gn.acc) +
                                        ((long long)threadIdx.x * gn.n * 32 + gi) * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = a[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int o = 1; o < kGnReplicas; o <<= 1) w[k] += __shfl_xor(w[k], o);
        if (threadIdx.x == 0) gn_mean_rstd(gn, w, mr[0], mr[1]);
      }
      __syncthreads();
      mean = mr[0];
      rstd = mr[1];
    }
  }
  __device__ __forceinline__ f32x4 run(int gi, long long i, long long group_elems) const {
    const long long e = gi * group_elems + 4 * i;
    const long long plane = e / hw;
    float sc, sh;
    if (gn.acc) {
      const int c = (int)(plane % gn.c);
      sc = rstd * gn.gamma[c];
      sh = gn.beta[c] - mean * sc;
    } else {
      sc = gn.ss[2 * plane];
      sh = gn.ss[2 * plane + 1];
    }
    f32x4 v = *reinterpret_cast<const f32x4 *>(x + e);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float o = v[q] * sc + sh;
      v[q] = (relu && o < 0.f) ? 0.f : o;
    }
    if (res) {
      const f32x4 r = *reinterpret_cast<const f32x4 *>(res + e);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = r[q] + v[q];
    }
    *reinterpret_cast<f32x4 *>(y + e) = v;
    return v;
  }
};

static int check_fin(mp_ctx *ctx, GnOut &f, int n, int c, long long partial_cap, const char *who) {
  if (!gn_wanted(f)) return MP_OK;
  f.c = c;
  f.S = kGnSlices;
  f.n = n;
  if (f.partial && partial_cap >= 0 && partial_cap < (long long)n * 32 * kGnSlices * 2)
    return fail(ctx, MP_ERR_ARG, "%s: statistics buffer holds %lld doubles, the launch writes %lld", who,
                partial_cap, (long long)n * 32 * kGnSlices * 2);
  return MP_OK;
}

int launch_avgpool2_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, float *y, GnOut fin,
                       long long partial_cap, hipStream_t st) {
  const long long hw_out = (long long)(h / 2) * (w / 2);
  int rc = check_fin(ctx, fin, n, c, partial_cap, "avgpool2_gn");
  if (rc != MP_OK) return rc;
  AvgPool2Op op{x, y, h / 2, w / 2};
  hipLaunchKernelGGL(ew_gn_kernel<AvgPool2Op>, dim3(n * 32 * kGnSlices), dim3(kGnThreads), 0, st, op,
                     (long long)(c / 32) * hw_out, fin);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// Round 4: the same pass with the HORIZONTAL sums shared.  upsample_bicubic2d's value is
// sum_j wy[j] * (sum_i wx[i] * src[yy_j][xx_i]); the inner sum depends on the source row and the output column
// only, and a source row feeds up to eight output rows.  A workgroup -- still (image, group, slice), so the
// statistics slots are the same -- walks its rows in bands of 16: it first writes the row sums of the <= 12
// source rows a band touches to LDS (4 loads + 4 FMAs each, by the thread that owns the column and holds wx),
// then every output is 4 LDS reads + 4 FMAs.  Same operations in the same order as bicubic2x_at (bit-identical
// values), ~35 instructions per output instead of ~100 (16 loads, 16 FMAs, two weight sets and a 64-bit
// division for the plane index): the one-output-per-thread form ran at ~1 TB/s.
constexpr int kUpBand = 16;              // output rows per band
constexpr int kUpRows = kUpBand / 2 + 4; // source rows a band can touch
__global__ __launch_bounds__(kGnThreads) void upsample_add_gn_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ add,
                                                                     float *__restrict__ y, int h, int w, int cpg,
                                                                     float sy, float sx, GnOut fin) {
  __shared__ float rs[kUpRows * kGnThreads];  // [source row of the band][output column]
  __shared__ double w1[kGnThreads / 64], w2[kGnThreads / 64];
  const int ho = 2 * h, wo = 2 * w;
  const int gi = blockIdx.x / kGnSlices, s = blockIdx.x % kGnSlices;  // (image, group), slice
  const int rps = cpg * ho / kGnSlices;                               // output rows per slice (a multiple of kUpBand)
  const int rp_n = kGnThreads / wo;                                   // rows handled side by side
  const int ox = threadIdx.x % wo, rp = threadIdx.x / wo;
  // this thread's column: taps and weights, once
  const float rx = sx * ox, fx = floorf(rx);
  const int ix = (int)fx;
  float wx[4];
  cubic_coeffs(rx - fx, wx);
  int xx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xx[i] = min(max(ix - 1 + i, 0), w - 1);

  float s1 = 0.f, s2 = 0.f;
  for (int r0 = s * rps; r0 < (s + 1) * rps; r0 += kUpBand) {  // row index inside the group's cpg * ho rows
    const int pl = r0 / ho, oy0 = r0 - pl * ho;                // bands do not cross planes: ho % kUpBand == 0
    const long long plane = (long long)gi * cpg + pl;
    const float *src = x + plane * (long long)h * w;
    const int lo = (int)floorf(sy * oy0) - 1;                  // first source row (unclamped) of the band
    __syncthreads();                                           // the previous band's sums have been read
    for (int r = rp; r < kUpRows; r += rp_n) {
      const int yy = min(max(lo + r, 0), h - 1);
      float row = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) row += src[yy * w + xx[i]] * wx[i];
      rs[r * wo + ox] = row;
    }
    __syncthreads();
    for (int k = rp; k < kUpBand; k += rp_n) {
      const int oy = oy0 + k;
      const float ry = sy * oy, fy = floorf(ry);
      const int iy = (int)fy;
      float wy[4];
      cubic_coeffs(ry - fy, wy);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += rs[(iy - 1 + j - lo) * wo + ox] * wy[j];
      const long long e = (plane * ho + oy) * wo + ox;
      const float v = add ? add[e] + acc : acc;
      y[e] = v;
      s1 += v;
      s2 += v * v;
    }
  }
  if (!gn_wanted(fin)) return;
  double d1 = s1, d2 = s2;  // <= a few hundred values per thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_down(d1, o);
    d2 += __shfl_down(d2, o);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    w1[wv] = d1;
    w2[wv] = d2;
  }
  __syncthreads();
  double a = 0, b = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kGnThreads / 64; ++i) {
      a += w1[i];
      b += w2[i];
    }
  gn_emit(fin, gi / 32, gi % 32, 1, s, a, b);
}

int launch_upsample_add_gn(mp_ctx *ctx, const float *x, int n, int c, int h, int w, const float *add, float *y,
                           GnOut fin, long long partial_cap, hipStream_t st) {
  const long long hw_out = 4LL * h * w;
  int rc = check_fin(ctx, fin, n, c, partial_cap, "upsample_add_gn");
  if (rc != MP_OK) return rc;
  const int cpg = c / 32, ho = 2 * h, wo = 2 * w;
  static const bool old_form = getenv("MONOPORT_UPSAMPLE") && getenv("MONOPORT_UPSAMPLE")[0] == 'o';  // A/B: "old"
  if (!old_form && c % 32 == 0 && wo <= kGnThreads && kGnThreads % wo == 0 && ho % kUpBand == 0 &&
      (cpg * ho) % (kGnSlices * kUpBand) == 0) {
    hipLaunchKernelGGL(upsample_add_gn_kernel, dim3(n * 32 * kGnSlices), dim3(kGnThreads), 0, st, x, add, y, h, w,
                       cpg, (float)(h - 1) / (float)(2 * h - 1), (float)(w - 1) / (float)(2 * w - 1), fin);
    MP_HIP(ctx, hipGetLastError());
    return MP_OK;
  }
  UpsampleAddOp op{x, add, y, h, w, (float)(h - 1) / (float)(2 * h - 1), (float)(w - 1) / (float)(2 * w - 1)};
  hipLaunchKernelGGL(ew_gn_kernel<UpsampleAddOp>, dim3(n * 32 * kGnSlices), dim3(kGnThreads), 0, st, op,
                     (long long)(c / 32) * hw_out, fin);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_gn_apply_gn(mp_ctx *ctx, const float *x, GnIn gn, int relu, int n, int c, long long hw,
                       const float *res, float *y, GnOut fin, long long partial_cap, hipStream_t st) {
  int rc = check_fin(ctx, fin, n, c, partial_cap, "gn_apply");
  if (rc != MP_OK) return rc;
  if (!gn_active(gn) || (gn.acc && (!gn.gamma || !gn.beta)))
    return fail(ctx, MP_ERR_ARG, "gn_apply: needs the GroupNorm of the input (accumulator + gamma / beta, or ss)");
  gn.c = c;
  gn.n = n;
  gn.count = (double)(c / 32) * hw;
  GnApplyOp op{x, res, y, hw, relu, gn, 0.0f, 1.0f};
  hipLaunchKernelGGL(ew_gn_kernel<GnApplyOp>, dim3(n * 32 * kGnSlices), dim3(kGnThreads), 0, st, op,
                     (long long)(c / 32) * hw, fin);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- concat + residual add (HGFilters.py:57-60) --------------------------------------------------
// torch runs cat (read 1x, write 1x) and add (read 2x, write 1x); here 2x read + 1x write.
__global__ __launch_bounds__(256) void concat3_add_kernel(const float *__restrict__ a, int ca,
                                                          const float *__restrict__ b, int cb,
                                                          const float *__restrict__ c, int cc,
                                                          const float *__restrict__ sc, long long hw4,
                                                          long long total4, float *__restrict__ y) {
  const int ct = ca + cb + cc;
  const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a), *b4 = reinterpret_cast<const f32x4 *>(b),
              *c4 = reinterpret_cast<const f32x4 *>(c), *s4 = reinterpret_cast<const f32x4 *>(sc);
  f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long plane = i / hw4, off = i - plane * hw4;  // plane = image * ct + channel
    const long long img = plane / ct;
    const int ch = (int)(plane - img * ct);
    f32x4 v;
    if (ch < ca) v = a4[(img * ca + ch) * hw4 + off];
    else if (ch < ca + cb) v = b4[(img * cb + (ch - ca)) * hw4 + off];
    else v = c4[(img * cc + (ch - ca - cb)) * hw4 + off];
    const f32x4 r = s4[i];
    v[0] += r[0];
    v[1] += r[1];
    v[2] += r[2];
    v[3] += r[3];
    y4[i] = v;
  }
}

int launch_concat3_add(mp_ctx *ctx, const float *a, int ca, const float *b, int cb, const float *c,
                       int cc, const float *sc, int n, long long hw, float *y, hipStream_t st) {
  const long long total4 = (long long)n * (ca + cb + cc) * (hw / 4);
  long long blocks = (total4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(concat3_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, ca, b, cb, c,
                     cc, sc, hw / 4, total4, y);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- input pre-step (RTL/main.py:352-364) ------------------------------------------------------
// One pass over the segmentation output instead of seven elementwise torch kernels; the operation
// order of the reference expression is kept (file is built with -ffp-contract=off).
struct Norm3 {
  float mean[3], std[3];
};

__global__ __launch_bounds__(256) void prepare_inputs_kernel(const float *__restrict__ segm,
                                                             long long hw, Norm3 nrm,
                                                             float *__restrict__ g,
                                                             float *__restrict__ c) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < hw;
       i += (long long)gridDim.x * blockDim.x) {
    const float m = segm[3 * hw + i];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float v = segm[ch * hw + i];
      const float a = v * 0.5f + 0.5f;                       // segm[:, 0:3] * 0.5 + 0.5
      g[ch * hw + i] = ((a - nrm.mean[ch]) / nrm.std[ch]) * m;
      if (c) c[ch * hw + i] = v * m;
    }
  }
}

int launch_prepare_inputs(mp_ctx *ctx, const float *segm, long long hw, const float *mean,
                          const float *std, float *g, float *c, hipStream_t st) {
  Norm3 nrm;
  for (int i = 0; i < 3; ++i) {
    nrm.mean[i] = mean[i];
    nrm.std[i] = std[i];
  }
  long long blocks = (hw + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prepare_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, segm, hw, nrm,
                     g, c);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
