// GroupNorm statistics taken by the kernel that PRODUCES a tensor, finalised inside that kernel.
//
// Every GroupNorm(32, C) of the encoders (backbones/HGFilters.py:23-27, ResBlkFilters.py:19) needs
// the mean / variance of a whole (image, group) before the first normalised value can be used, so
// each one is a grid-wide dependency.  Round 2 split it into "partial sums in the producer's
// epilogue" + a one-wave-per-group gn_finalize launch (111 launches per frame, 0.6 ms at batch 1).
// Here the producer's workgroups publish their partial sums, bump an arrival counter, and the LAST
// workgroup to arrive reduces the partials in a fixed order (deterministic: no floating-point
// atomics) and writes (scale, shift) for the one or two GroupNorm modules that will read the
// tensor.  The consumer's staging loop applies them (csrc/conv3x3.hip), so a normalised tensor
// never exists in memory and a GroupNorm costs no launch.
//
// Memory model: partials are plain global stores, followed by an agent-scope release fence and an
// agent-scope atomic on the counter; the last arriver issues an acquire fence before it reads the
// other workgroups' partials (the pattern of a "last block" reduction).  Counters are left at zero,
// so one zero-initialised buffer serves every launch on a stream.
#pragma once
#include "encoder_kernels.h"

namespace mp {

constexpr int kGnTailLdsBytes = 256 * 16 + 16;  // what gn_publish needs for a 256-thread workgroup

// Called by ALL NT threads of a workgroup with uniform arguments.  Threads tid < ng hold (a, b) =
// this workgroup's (sum, sum of squares) of group g0 + tid of image img; they go to slot `slot`.
// `cidx` / `expected`: the arrival counter shared by the workgroups that cover the same groups of
// the same image, and how many of them there are.  lds: >= kGnTailLdsBytes, free for use.
template <int NT>
__device__ __forceinline__ void gn_publish(const GnFin &f, int img, int g0, int ng, int slot, int cidx,
                                           int expected, double a, double b, unsigned char *lds_raw) {
  const int tid = threadIdx.x;
  if (tid < ng) {
    double *dst = f.partial + (((long long)img * 32 + g0 + tid) * f.S + slot) * 2;
    dst[0] = a;
    dst[1] = b;
  }
  if (f.n_sets == 0) return;
  double *lds = reinterpret_cast<double *>(lds_raw);
  int *flag = reinterpret_cast<int *>(lds_raw + NT * 16);
  if (tid < 64) {  // the partials were written by wave 0 (ng <= 64)
    __threadfence();
    if (tid == 0) {
      const int prev = atomicAdd(f.counter + cidx, 1);
      const int last = prev == expected - 1;
      if (last) f.counter[cidx] = 0;  // no other workgroup of this launch touches it any more
      *flag = last;
    }
  }
  __syncthreads();
  if (!*flag) return;
  __threadfence();
  const int tpg = NT / ng;  // threads per group (ng is a power of two <= 64)
  const int gl = tid / tpg, r = tid - gl * tpg;
  const double *src = f.partial + ((long long)img * 32 + g0 + gl) * f.S * 2;
  double sa = 0.0, sb = 0.0;
  for (int s = r; s < f.S; s += tpg) {
    sa += src[2 * s];
    sb += src[2 * s + 1];
  }
  lds[2 * tid] = sa;
  lds[2 * tid + 1] = sb;
  __syncthreads();
  if (r == 0) {
    sa = sb = 0.0;
    const int kmax = tpg < f.S ? tpg : f.S;  // threads r >= S had no slot to add
    for (int k = 0; k < kmax; ++k) {
      sa += lds[2 * (tid + k)];
      sb += lds[2 * (tid + k) + 1];
    }
    const double mean_d = sa / f.count;
    const double var_d = fmax(sb / f.count - mean_d * mean_d, 0.0);
    const float mean = (float)mean_d;
    const int cpg = f.c / 32;
    for (int k = 0; k < f.n_sets; ++k) {
      const GnSet &st = f.set[k];
      const float rstd = (float)(1.0 / sqrt(var_d + (double)st.eps));
      for (int ch = 0; ch < cpg; ++ch) {
        const int c = (g0 + gl) * cpg + ch;
        const float sc = rstd * st.gamma[c];
        float *o = st.ss + ((long long)img * f.c + c) * 2;
        o[0] = sc;
        o[1] = st.beta[c] - mean * sc;
      }
    }
  }
}

}  // namespace mp
