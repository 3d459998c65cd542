// GroupNorm statistics handed from the kernel that PRODUCES a tensor to the kernel that READS it.
//
// Every GroupNorm(32, C) of the encoders (backbones/HGFilters.py:23-27, ResBlkFilters.py:19) needs
// the mean / variance of a whole (image, group) before the first normalised value can be used, so
// each one is a grid-wide dependency.  Round 2 split it into "partial sums in the producer's
// epilogue" + a one-wave-per-group gn_finalize launch (111 launches per frame, 0.6 ms at batch 1).
//
// What was measured on the way here (profiles/r03b..e):
//   * a textbook "last block" reduction inside the producer (partials, release fence, arrival
//     counter, last workgroup finalises): the agent-scope release fence is buffer_wbl2 sc1 on
//     gfx950 -- a write-back of the XCD's whole L2, per workgroup, in kernels whose job is to dirty
//     the L2 with their output: elementwise producers 4-8x, convolutions 15-40 % slower;
//   * the same with agent-scope atomic stores / loads instead of fences (no cache-wide flush):
//     correct and deterministic, but every workgroup waits for its partial stores to complete and
//     for the counter round trip before it can retire: +20-30 us per convolution launch at batch
//     10, +8-20 us at batch 1 (tools/conv_epi_probe.py) -- more than the finalize launch it saves.
//
// So nothing waits.  Producers ADD their per-group sums into a small accumulator with fire-and-
// forget integer atomics; consumers turn the accumulator into (mean, rstd) in their prologue:
//   * acc [R][N][32][4] int64 per normalised tensor: (sum hi, sum lo, sumsq hi, sumsq lo), a 112-bit
//     fixed-point pair per sum -- value = hi * 2^-16 + lo * 2^-64.  Integer addition is
//     associative, so the result does not depend on the order the workgroups arrive in:
//     deterministic, unlike floating-point atomics.  R = kGnReplicas copies: device-scope atomics
//     on ONE address serialise at ~80 ns each, and a batch-1 launch sends 250-1000 of them per
//     group (+20 us on a 20 us convolution, profiles/r03f); workgroup w adds to copy w % R, the
//     consumer adds the copies up (integers again: exact);
//   * a producer workgroup folds its per-channel sums to per-group doubles in LDS (fixed order) and
//     issues 4 atomics per group it covers, without using their return values or waiting for them;
//   * a consumer workgroup reads the R x 32 x 4 words of its image (L2 hits after the first workgroup),
//     computes mean / rstd in double with gn_finalize_kernel's formulas, keeps them in LDS and
//     derives scale = rstd * gamma[c], shift = beta[c] - mean * scale per staged chunk;
//   * accumulators must be ZERO before the producing launch: the callers clear one arena per
//     encoder pass (one memset for ~130 GroupNorms).
#pragma once
#include <type_traits>

#include "encoder_kernels.h"

namespace mp {

// Sum of v over the 32 lanes that share lane >> 5 (the 32 pixels of a C-layout tile row): the total is valid in
// the lanes with (lane & 31) >= 16 -- callers let lane kHalfSumLane of each half write it.  Five DPP additions
// (xor 1, xor 2, row_half_mirror, row_mirror, row_bcast15), no LDS round trip: __shfl_xor compiles to
// ds_bpermute_b32 + s_waitcnt, and the statistics epilogue of a 3x3 convolution tile had 320 of those
// (13 k cycles next to a 37 k-cycle K loop for 64 -> 64 channels).
constexpr int kHalfSumLane = 16;
__device__ __forceinline__ float half_wave_sum(float v) {
  auto dpp = [](float x, auto ctrl_tag) {
    constexpr int ctrl = decltype(ctrl_tag)::value;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror: the other quad pair of each 8
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror: all 16 lanes of a row hold the row sum
  // row_bcast15 into rows 1 and 3: lanes 16-31 / 48-63 add the sum of lanes 0-15 / 32-47
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  return v;
}

// x -> (hi, lo) with x = hi * 2^-16 + lo * 2^-64, lo in [0, 2^48); exact for |x| < 2^36, rounded to a
// multiple of 2^-16 above that (a double has no finer bits there); needs |x| < 2^46.
__device__ __forceinline__ void gn_fixed(double x, long long &hi, unsigned long long &lo) {
  const double xs = x * 65536.0;
  const double fl = floor(xs);
  hi = (long long)fl;
  lo = (unsigned long long)((xs - fl) * 281474976710656.0);  // 2^48
}

__device__ __forceinline__ double gn_unfixed(long long hi, unsigned long long lo) {
  return (double)hi * (1.0 / 65536.0) + (double)lo * (1.0 / 18446744073709551616.0);  // 2^-16, 2^-64
}

// Producer: thread-level.  (a, b) = this workgroup's (sum, sum of squares) of one (image, group);
// dst = that group's 4 words in the replica this workgroup uses.
__device__ __forceinline__ void gn_acc_add(long long *dst_ll, double a, double b) {
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(dst_ll);
  long long hi;
  unsigned long long lo;
  gn_fixed(a, hi, lo);
  (void)__hip_atomic_fetch_add(dst, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)__hip_atomic_fetch_add(dst + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  gn_fixed(b, hi, lo);
  (void)__hip_atomic_fetch_add(dst + 2, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)__hip_atomic_fetch_add(dst + 3, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Producer: what a workgroup does with the (sum, sum of squares) of group g0 + tid it holds in
// threads tid < ng: accumulate for the hand-over (replica slot % R) and / or leave it in the legacy
// partial buffer (slot `slot` of [N][32][S][2], finalised by mp_gn_finalize).
__device__ __forceinline__ void gn_emit(const GnOut &f, int img, int g0, int ng, int slot, double a, double b) {
  const int tid = threadIdx.x;
  if (tid < ng) {
    if (f.acc)
      gn_acc_add(f.acc + (((long long)(slot % kGnReplicas) * f.n + img) * 32 + g0 + tid) * 4, a, b);
    if (f.partial) {
      double *dst = f.partial + (((long long)img * 32 + g0 + tid) * f.S + slot) * 2;
      dst[0] = a;
      dst[1] = b;
    }
  }
}

// Consumer: mean / rstd of one group from its 4 accumulator words (gn_finalize_kernel's formulas:
// biased variance, eps inside the square root).
__device__ __forceinline__ void gn_mean_rstd(const GnIn &g, const unsigned long long (&w)[4], float &mean,
                                             float &rstd) {
  const double s = gn_unfixed((long long)w[0], w[1]);
  const double q = gn_unfixed((long long)w[2], w[3]);
  const double mean_d = s / g.count;
  const double var_d = fmax(q / g.count - mean_d * mean_d, 0.0);
  mean = (float)mean_d;
  rstd = (float)(1.0 / sqrt(var_d + (double)g.eps));
}

// Consumer, called by all 256 threads of a workgroup: the accumulator of image img -> (mean, rstd)
// of its 32 groups in LDS.  Thread (group = tid / 8, part = tid % 8) adds replicas part, part + 8;
// the 8 lanes of a group combine with integer adds.  The caller puts a workgroup barrier between
// this and the first gn_scale_shift.
__device__ __forceinline__ void gn_load_stats(const GnIn &g, int img, float *lds_stats /*[32][2]*/) {
  if (!g.acc) return;
  const int tid = threadIdx.x;
  const int grp = (tid >> 3) & 31, part = tid & 7;
  unsigned long long w[4] = {0, 0, 0, 0};
  for (int r = part; r < kGnReplicas; r += 8) {
    const unsigned long long *a =
        reinterpret_cast<const unsigned long long *>(g.acc) + (((long long)r * g.n + img) * 32 + grp) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] += a[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) w[k] += __shfl_xor(w[k], o);
  if (part == 0 && tid < 256) {
    float mean, rstd;
    gn_mean_rstd(g, w, mean, rstd);
    lds_stats[2 * grp] = mean;
    lds_stats[2 * grp + 1] = rstd;
  }
}

// Consumer: (scale, shift) of channel c of image img -- from the LDS statistics (hand-over), a
// precomputed ss [N][C][2] (legacy) or the identity.
__device__ __forceinline__ void gn_scale_shift(const GnIn &g, int img, int c, const float *lds_stats, float &sc,
                                               float &sh) {
  if (g.acc) {
    const int grp = c / (g.c / 32);
    const float mean = lds_stats[2 * grp], rstd = lds_stats[2 * grp + 1];
    sc = rstd * g.gamma[c];
    sh = g.beta[c] - mean * sc;
  } else if (g.ss) {
    sc = g.ss[2 * ((long long)img * g.c + c)];
    sh = g.ss[2 * ((long long)img * g.c + c) + 1];
  } else {
    sc = 1.0f;
    sh = 0.0f;
  }
}

// Consumer, all 256 threads: the (scale, shift) table of the normalised tensor's channels in LDS, one thread
// per channel (two for up to 512 channels).  gamma / beta are requested BEFORE the statistics barrier
// (gn_affine_load, next to the first activation loads) and combined with the statistics after it
// (gn_table_fill): a workgroup's prologue holds one round trip to L2 instead of two -- at batch 1 the small
// convolutions are 12-30 us launches.
struct GnAffine {
  float g[2], b[2];
};
__device__ __forceinline__ void gn_affine_load(const GnIn &gn, int channels, GnAffine &a) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + 256 * i;
    a.g[i] = a.b[i] = 0.0f;
    if (gn.acc && c < channels) {
      a.g[i] = gn.gamma[c];
      a.b[i] = gn.beta[c];
    }
  }
}
__device__ __forceinline__ void gn_table_fill(const GnIn &gn, int img, int channels, const float *lds_stats,
                                              const GnAffine &a, float *ss /*[channels][2]*/) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < channels) {
      float sc, sh;
      if (gn.acc) {
        const int grp = c / (gn.c / 32);
        const float mean = lds_stats[2 * grp], rstd = lds_stats[2 * grp + 1];
        sc = rstd * a.g[i];
        sh = a.b[i] - mean * sc;
      } else {
        gn_scale_shift(gn, img, c, lds_stats, sc, sh);
      }
      ss[2 * c] = sc;
      ss[2 * c + 1] = sh;
    }
  }
}

}  // namespace mp
