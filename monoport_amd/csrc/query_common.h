// Device helpers shared by the fused query kernels (query.hip: f32 MFMA, query16.hip: f16x3).
#pragma once
#include "mp_internal.h"

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// geometry.py:27-29: trans + rot @ p, unfused like the CPU oracle.
__device__ __forceinline__ void project(const float *__restrict__ cal, float px, float py,
                                        float pz, float &x, float &y, float &z) {
  x = __fadd_rn(cal[3], __fadd_rn(__fadd_rn(__fmul_rn(cal[0], px), __fmul_rn(cal[1], py)),
                                  __fmul_rn(cal[2], pz)));
  y = __fadd_rn(cal[7], __fadd_rn(__fadd_rn(__fmul_rn(cal[4], px), __fmul_rn(cal[5], py)),
                                  __fmul_rn(cal[6], pz)));
  z = __fadd_rn(cal[11], __fadd_rn(__fadd_rn(__fmul_rn(cal[8], px), __fmul_rn(cal[9], py)),
                                   __fmul_rn(cal[10], pz)));
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

__device__ __forceinline__ f32x4 blend(const f32x4 &a, const f32x4 &b, const f32x4 &c,
                                       const f32x4 &d, const Taps &t) {
  f32x4 r = a * t.w[0];
  r += b * t.w[1];
  r += c * t.w[2];
  r += d * t.w[3];
  return r;
}

__device__ __forceinline__ float activate(float v, int act) {
  if (act == MP_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  if (act == MP_ACT_TANH) return tanhf(v);
  return v;
}

}  // namespace mp
