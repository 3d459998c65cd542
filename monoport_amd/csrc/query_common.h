// Device helpers shared by the fused query kernels (query.hip: f32 MFMA, query16.hip: f16x3).
#pragma once
#include "mp_internal.h"

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- weight stream -----------------------------------------------------------------------------
// Every weight / bias read of the MLP is  base + (wave-uniform offset) + (lane part): issued as
// buffer loads through ONE 128-bit resource descriptor in SGPRs, the lane part in one shared
// VGPR and the uniform part in the instruction's scalar offset (SALU adds).  With 64-bit flat
// addresses hipcc hoisted ~70 lane-dependent pointers out of the tile loop and spilled them
// (132 VGPRs, 161 scratch reloads and 6 scratch stores per tile -- the WRITE_SIZE of round 1).
struct WStream {
  __amdgpu_buffer_rsrc_t rs;
  int lane16;  // lane * 16: this lane's 16-byte slot of a 64-lane fragment
  int lane4;   // lane * 4
  int h16;     // (lane >> 5) * 16
};

__device__ __forceinline__ WStream make_wstream(const float *base, int n_floats, int lane) {
  WStream w;
  // 0x00020000: raw buffer, 32-bit data format (cdna_hip_programming.md T8); base and size come
  // from kernel arguments, so the descriptor is provably wave-uniform (no waterfall loops, T20)
  w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, n_floats * 4, 0x00020000);
  w.lane16 = lane * 16;
  w.lane4 = lane * 4;
  w.h16 = (lane >> 5) * 16;
  return w;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// fragment `idx` (units of 64 x 16 bytes would be idx * 64; here idx is in 16-byte units)
__device__ __forceinline__ f32x4 wload128(const WStream &w, int idx16) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.lane16, idx16 * 16, 0));
}
// one float per lane at float index f0 + lane
__device__ __forceinline__ float wload32(const WStream &w, int f0) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w.rs, w.lane4, f0 * 4, 0));
}
// 4 floats at float index f0 + 4 h (bias pieces of a C-layout tile)
__device__ __forceinline__ f32x4 wload_bias4(const WStream &w, int f0) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.h16, f0 * 4, 0));
}

constexpr int kHbRowBytes = 64 * 4;                    // one point's 64-row hidden chunk
constexpr int kHbBytes = kTilePts * kHbRowBytes;       // 16 KB

// ---- point sources ---------------------------------------------------------------------------
__device__ __forceinline__ float lattice_coord(int idx, const PointSrc &s, int axis) {
  // Seg3dLossless.batch_eval (align_corners=False): ((c / R) + (1/R)/2) * (b_max-b_min) + b_min,
  // each step rounded to f32 -- identical op sequence to oracle/pifu_oracle.py:lattice_points.
  const float c = (float)(idx * s.stride);
  const float u = __fadd_rn(__fdiv_rn(c, s.res_final), s.half_step);
  return __fadd_rn(__fmul_rn(u, s.blen[axis]), s.bmin[axis]);
}

__device__ __forceinline__ void load_point(const PointSrc &s, long long n, float &px, float &py,
                                           float &pz, uint32_t &code) {
  if (s.packed) {
    code = s.packed[n];
    px = lattice_coord(code & 1023u, s, 0);
    py = lattice_coord((code >> 10) & 1023u, s, 1);
    pz = lattice_coord(code >> 20, s, 2);
  } else {
    code = 0;
    px = s.pts[n * s.sn];
    py = s.pts[n * s.sn + s.sc];
    pz = s.pts[n * s.sn + 2 * s.sc];
  }
}

// geometry.py:27-29: trans + rot @ p through torch.baddbmm.  On the reference's CPU path that is an
// MKL sgemm (any N >= 64): the K = 3 dot product is an FMA chain started by a plain product, the
// translation is added last -- t + fma(r2, pz, fma(r1, py, r0 * px)).  Restated exactly (checked
// bit for bit against torch.baddbmm in tests/test_oracle_golden.py::test_orthogonal_*), because the
// in-image mask and the bilinear coordinates hang on the last bit of x and y.
__device__ __forceinline__ float project_row(const float *__restrict__ r, float px, float py,
                                             float pz) {
  return __fadd_rn(r[3], fmaf(r[2], pz, fmaf(r[1], py, __fmul_rn(r[0], px))));
}

__device__ __forceinline__ void project(const float *__restrict__ cal, float px, float py,
                                        float pz, float &x, float &y, float &z) {
  x = project_row(cal, px, py, pz);
  y = project_row(cal + 4, px, py, pz);
  z = project_row(cal + 8, px, py, pz);
}

__device__ __forceinline__ bool in_image(float x, float y) {  // MonoPortNet.py:74
  return x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f;
}

// grid_sample(align_corners=True, padding zeros): 4 tap offsets (in floats) + weights.
struct Taps {
  long long o[4];
  float w[4];
};

__device__ __forceinline__ Taps make_taps(float x, float y, int h, int w, int c, bool live) {
  Taps t;
  const float ix = __fmul_rn(__fmul_rn(__fadd_rn(x, 1.0f), 0.5f), (float)(w - 1));
  const float iy = __fmul_rn(__fmul_rn(__fadd_rn(y, 1.0f), 0.5f), (float)(h - 1));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
  const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
  const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)w);
  const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)h);
  const int x1 = x0 + 1, y1 = y0 + 1;
  const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w;
  const bool vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
  const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
  t.o[0] = ((long long)cy0 * w + cx0) * c;
  t.o[1] = ((long long)cy0 * w + cx1) * c;
  t.o[2] = ((long long)cy1 * w + cx0) * c;
  t.o[3] = ((long long)cy1 * w + cx1) * c;
  t.w[0] = (live && vx0 && vy0) ? __fmul_rn(wx0, wy0) : 0.0f;
  t.w[1] = (live && vx1 && vy0) ? __fmul_rn(wx1, wy0) : 0.0f;
  t.w[2] = (live && vx0 && vy1) ? __fmul_rn(wx0, wy1) : 0.0f;
  t.w[3] = (live && vx1 && vy1) ? __fmul_rn(wx1, wy1) : 0.0f;
  return t;
}

// torch's CPU grid_sample (vectorised bilinear kernel, built with FMA contraction) evaluates
// fma(se, w_se, fma(sw, w_sw, fma(ne, w_ne, nw * w_nw))): restated exactly -- the sampled features
// are bit-identical to the reference's (tests: index golden, array_equal).
__device__ __forceinline__ f32x4 blend(const f32x4 &a, const f32x4 &b, const f32x4 &c,
                                       const f32x4 &d, const Taps &t) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = fmaf(d[i], t.w[3], fmaf(c[i], t.w[2], fmaf(b[i], t.w[1], __fmul_rn(a[i], t.w[0]))));
  return r;
}

__device__ __forceinline__ float activate(float v, int act) {
  if (act == MP_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  if (act == MP_ACT_TANH) return tanhf(v);
  return v;
}

}  // namespace mp
