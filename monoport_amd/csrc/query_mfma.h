// MFMA building blocks of the fused f32 query kernels (query.hip: 64-point tiles, query_small.hip:
// 32-point tiles for launches that cannot fill the chip): fragment-stream segments, the z column,
// bias initialisation, leaky ReLU and the point-major hidden-chunk store.
#pragma once
#include "mp_internal.h"
#include "query_common.h"

// Reference parity is op-order parity: keep every a*b+c exactly as written (the HIP headers
// define __fmul_rn & co. as plain operators, which hipcc would otherwise contract into FMAs).
// Fused multiply-adds are requested explicitly (fmaf / MFMA) where they are wanted.
#pragma clang fp contract(off)

// Timing experiments only (tools/ablate.py builds side libraries with these; never in the product):
//   MP32_NOBAR   drop the chunk-loop barriers (wrong results)   -> cost of the barriers
//   MP32_AHOT    every A fragment read hits one cached line     -> cost of weight streaming
//   MP32_GATHER_ONLY  stop after the gather                     -> the sampling stage on its own
#define MP_CHUNK_SYNC() __syncthreads()
constexpr int kPrefetch1 = 1;  // A-fragment prefetch distance (k-groups) of the MR = 4 / MR = 2 segments
constexpr int kPrefetch0 = 3;  // same for layer 0's MR = 1 segment
constexpr int kAHot = 0;
#define MP_AG(g) (HOT ? 0 : (g))

namespace mp {

// ---- MFMA building blocks ----------------------------------------------------------------------
template <int MR, int NR>
__device__ __forceinline__ void mma_group(f32x16 (&acc)[MR][NR], const f32x4 (&a)[MR],
                                          const f32x4 (&b)[NR]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int n = 0; n < NR; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][i], b[n][i], acc[m][n], 0, 0, 0);
}

// acc[MR][NR] += A[rows of this wave][K segment] * B[K segment][points], in two calls:
//   seg_prefetch : issue the first PF groups of A into the ring -- placed EARLY by the caller
//                  (before the previous segment's epilogue / barrier) so L2 latency is hidden;
//   seg_main     : the K loop.  Every iteration issues the A fragment PF groups ahead and the B
//                  operand one group ahead, then the 4*MR*NR MFMAs of the current group.
//   a: fragment stream of row block 0 at group 0 as a wave-uniform index into the weight buffer
//      (16-byte units; the lane's slot is added by the buffer load); row block m is
//      a + m * rb_stride; group g is + g * 64.
//   b: LDS byte address of this lane's point row for column block 0; column block n is
//      + n * 32 * ROWB; group g lives in 16-byte slot (2g + h) ^ (p & 15) = (2g) ^ swz.
// The loop is deliberately NOT unrolled beyond the ring size: hipcc clusters every load of a
// big unrolled block at its top and spills the accumulators.
template <int MR, int PF, bool HOT = (kAHot & 4) != 0>
__device__ __forceinline__ void seg_prefetch(f32x4 (&ring)[PF + 1][MR], const WStream &ws, int a,
                                             int rb_stride, int n_groups) {
#pragma unroll
  for (int d = 0; d < PF; ++d)
#pragma unroll
    for (int m = 0; m < MR; ++m)
      ring[d][m] = wload128(ws, a + m * rb_stride + MP_AG(min(d, n_groups - 1)) * 64);
}

template <int MR, int NR, int PF, int ROWB, bool HOT = (kAHot & 4) != 0>
__device__ __forceinline__ void seg_main(f32x16 (&acc)[MR][NR], f32x4 (&ring)[PF + 1][MR],
                                         const WStream &ws, int a, int rb_stride, int n_groups,
                                         const unsigned char *b, int swz) {
  constexpr int RS = PF + 1;
  f32x4 bcur[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n)
    bcur[n] = *reinterpret_cast<const f32x4 *>(b + n * 32 * ROWB + (swz << 4));
#pragma unroll 1
  for (int g0 = 0; g0 < n_groups; g0 += RS) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int g = g0 + r;
      const int gp = min(g + PF, n_groups - 1);
#pragma unroll
      for (int m = 0; m < MR; ++m)
        ring[(r + PF) % RS][m] = wload128(ws, a + m * rb_stride + MP_AG(gp) * 64);
      const int boff = ((2 * min(g + 1, n_groups - 1)) ^ swz) << 4;
      f32x4 bnxt[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n)
        bnxt[n] = *reinterpret_cast<const f32x4 *>(b + n * 32 * ROWB + boff);
      // keep the prefetches ABOVE this group's MFMAs: left alone, hipcc sinks them to the end of
      // the group (to recycle registers) and every group then starts with a full L2 round trip
      __builtin_amdgcn_sched_barrier(0);
      mma_group<MR, NR>(acc, ring[r % RS], bcur);
#pragma unroll
      for (int n = 0; n < NR; ++n) bcur[n] = bnxt[n];
    }
  }
}

// seg_main for ONE row block x ONE column block (layer 3 of the wave-specialised table kernel: 32 rows per consumer
// wave) with the k-steps alternating between TWO accumulators: a wave whose MFMAs chain through a single accumulator
// runs at 2/3 of the matrix rate when another wave shares its SIMD (profiles/r03y_mfma_peak_probe.txt); two
// independent chains restore it.  The caller adds the two accumulators at the end (addition order is free).
template <int PF, int ROWB>
__device__ __forceinline__ void seg_main_split(f32x16 &acc_e, f32x16 &acc_o, f32x4 (&ring)[PF + 1][1], const WStream &ws,
                                               int a, int n_groups, const unsigned char *b, int swz) {
  constexpr int RS = PF + 1;
  f32x4 bcur = *reinterpret_cast<const f32x4 *>(b + (swz << 4));
#pragma unroll 1
  for (int g0 = 0; g0 < n_groups; g0 += RS) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int g = g0 + r;
      const int gp = min(g + PF, n_groups - 1);
      ring[(r + PF) % RS][0] = wload128(ws, a + gp * 64);
      const int boff = ((2 * min(g + 1, n_groups - 1)) ^ swz) << 4;
      const f32x4 bnxt = *reinterpret_cast<const f32x4 *>(b + boff);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 af = ring[r % RS][0];
      acc_e = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bcur[0], acc_e, 0, 0, 0);
      acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bcur[1], acc_o, 0, 0, 0);
      acc_e = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bcur[2], acc_e, 0, 0, 0);
      acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bcur[3], acc_o, 0, 0, 0);
      bcur = bnxt;
    }
  }
}

// The z column: one k-step whose B operand is z_feat in lanes 0-31 and 0 in lanes 32-63.
template <int MR, int NR>
__device__ __forceinline__ void gemm_z(f32x16 (&acc)[MR][NR], const float (&az)[MR],
                                       const float (&zb)[NR]) {
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(az[m], zb[n], acc[m][n], 0, 0, 0);
}

// Accumulators start from the bias: register t of lane (j, h) of a C-layout tile holds row
// (t & 3) + 8 (t >> 2) + 4 h of the 32-row block (cdna_hip_programming.md section 3), so the 16
// registers are four 16-byte pieces of the bias vector.
__device__ __forceinline__ void init_from_bias(f32x16 &v, const WStream &ws, int bias32) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bq = wload_bias4(ws, bias32 + 8 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[4 * q + i] = bq[i];
  }
}

__device__ __forceinline__ void lrelu(f32x16 &v) {
#pragma unroll
  for (int t = 0; t < 16; ++t)
    v[t] = fmaxf(v[t], v[t] * 0.01f);  // = v > 0 ? v : 0.01 v (F.leaky_relu, SurfaceClassifier.py:58), bit for bit
}

// Store a C-layout 32x32 tile into the hidden-chunk buffer, point-major: rows 8q+4h..+3 of a
// point are 4 consecutive floats = one 16-byte slot.
template <int ROWBYTES = kHbRowBytes>
__device__ __forceinline__ void store_hidden(unsigned char *hb, const f32x16 &v, int rb_local,
                                             int cb, int j, int h) {
  const int p = 32 * cb + j;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int slot = 8 * rb_local + 2 * q + h;
    f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    *reinterpret_cast<f32x4 *>(hb + p * ROWBYTES + ((slot ^ (p & 15)) << 4)) = o;
  }
}

}  // namespace mp
