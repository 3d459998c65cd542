// Fused PIFu query, split-precision variant ("f16x3"): the same computation as query.hip
// (MonoPortNet.query in eval mode, monoport/lib/modeling/MonoPortNet.py:48-91) with every GEMM
// operand carried as TWO halves, v = hi + lo (hi = f16(v), lo = f16(v - hi): 22 significant bits)
// and every product expanded as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with f32
// accumulation.  The dropped lo*lo term is 2^-22 relative, so the result is f32-class (measured
// against the fp64 oracle in tests/test_query_gpu.py) while the matrix pipe runs 16x faster per
// instruction: 3 MFMAs of 32 cycles replace 8 MFMAs of 64 cycles per 16-deep k-step (5.3x).
//
// Everything that was "free" next to f32 MFMA now matters, so the decomposition changes:
//   * one workgroup = 4 waves = a 96-point tile, one wave per SIMD with the whole 512-entry register
//     file: layer 1's 128 rows x 96 points per wave are 192 accumulator registers, a layer-0 chunk
//     48 more -- 240, which fit the 256 AGPRs, so no accumulator tile ever moves between the two
//     register files (the 128-point tile of round 1 needed 256 + 32 and the allocator shuffled
//     tiles and spilled 100 registers; kQ16Nb=4 still builds it);
//   * LDS: xs[96][hi 512 B | lo 512 B] = 96 KB + one 128-row hidden chunk [96][hi 256 B | lo 256 B]
//     = 48 KB (144 KB, one workgroup per CU);
//   * layer 0 is produced in 128-row chunks -- one row block x all three column blocks per wave, so
//     every weight fragment of layer 0 is loaded by exactly one wave and feeds 9 MFMAs (the
//     128-point tile's 64-row chunks had two waves load each) -- split into halves on the way to
//     LDS and consumed by layer 1 (8 k16 groups per chunk); layers 2 and 3 stream their inputs
//     from the owning waves' registers in 64-row chunks made of 16 rows from each wave;
//   * the prefetches of a k16 group (weight fragments, LDS reads of the next group's activations)
//     are placed one behind each MFMA (sched_group_barrier) instead of in a clump in front of them.
// Weights are pre-split and pre-scaled by a per-layer power of two S (pack.hip) so that lo stays
// out of the f16 subnormals; accumulators start at bias * S and are multiplied by 1/S (exact)
// before the leaky ReLU.  Activations larger than 65504 would saturate -- PIFu activations are O(1-100).
//
// Status (round 2): 1 M points in 6.44-6.60 ms = 160 M points/s (2.7x the f32 kernel) = 375-385
// TFLOP/s-equivalent = 0.45-0.46 of the three-MFMA-per-product roof (2.5 PFLOP/s / 3); 0.44 over a
// reconstruction's launches (128-point tile: 0.42).
// What bounds it (side builds, tools/ablate.py -DMP16_ABLATE=..., profiles/r02w_f16x3_ablation.txt):
// removing the layer-0 conversion buys 2 %, the barriers 1.6 %, the LDS reads 4 % -- and removing
// the WEIGHT LOADS 27 % (6.60 -> 4.81 ms; layers 0-1 alone 4.42 -> 3.03 ms against a 2.69 ms MFMA
// floor).  The loads' cost is ADDITIVE, ~76 cycles of a SIMD's matrix-pipe time per 1-KB fragment
// (50 M fragments per 1 M points), and nothing tried moves it:
//   * deeper prefetch rings (layer 0: 7 instead of 3 groups, layer 1: 3 instead of 1): slower -- it
//     is not latency;
//   * eight waves per workgroup, two per SIMD, rows split so that every wave streams half as much
//     and no fragment is loaded twice (built, tests green, removed): 6.9 ms, layers 0-1 4.36 ms --
//     it is neither the per-wave streaming cap (tools/probes/l2_stream_probe.hip: ~8 KB in flight
//     = 7.7 B/clk per wave, which one wave per SIMD does sit at) nor something a partner wave hides;
//   * group-major fragment order (contiguous blocks for the waves of a workgroup): no change -- not
//     L2 channel camping; sc0 / sc1 / nt cache policies on the weight loads: equal or slower.
// With that cost fixed, cycles per point = 1729 (MFMA) + 76 x fragments per point per SIMD (942 at
// 96 points per tile, 707 at 128): 0.45-0.47 of the roof here; 0.6 would need a tile of >= 240 points
// = 240 KB of split features in LDS.  The f32 kernel streams the same bytes per point but spends 4x
// the matrix-pipe time per fragment, which is why it sits at 0.91.  Also measured and dropped: a
// software-pipelined layer-0/1 loop over 32-row chunks (6.97 vs 6.65 ms), kQ16Cs=2 (column split,
// duplicated loads: 9.3 ms) and the skip-connection MFMAs of layer 1 issued under the layer-0
// conversion (correct, the interleave comes out as written, neutral: 6.60-6.62 vs 6.57-6.59 ms).
#include <cstdlib>
#include <type_traits>

#include "mp_internal.h"
#include "query_common.h"

#pragma clang fp contract(off)

namespace mp {


typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int kThreads16 = 256;  // 4 waves = one per SIMD, each with the full 512-register file
constexpr int kXRow = 1024;      // bytes per point in xs: 32 hi slots | 32 lo slots (16 B each)
constexpr int kHRow = 256;       // bytes per point in the hidden chunk: 8 hi slots | 8 lo slots
constexpr int kQ16Nb = 3;  // tile shape the launcher instantiates (see pifu_query16_kernel)
// column split: 2 = eight waves, two per SIMD (see the kernel).  Correct (the f16 tests pass with
// it) but MEASURED SLOWER for f16x3 -- 1 M points 9.3 ms vs 7.0 ms: 128 accumulator registers + the
// A ring + double-buffered hi/lo B fragments do not fit 256 registers (103 spilled, scratch traffic
// inside the chunk loop) and both waves of a row group load every weight fragment.  Plain f16 gains
// 5 % (3.63 vs 3.82 ms).  Kept for tools/ablate.py; the product uses 1.
constexpr int kQ16Cs = 1;

struct AFrag {
  h8 hi, lo;
};

__device__ __forceinline__ void split4(const f32x4 &v, h4 &hi, h4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)v[i];
    lo[i] = (_Float16)(v[i] - (float)hi[i]);
  }
}

// a: wave-uniform index (16-byte units) of row block 0, group 0, part hi in the f16 weight buffer
// (the lane's slot is added by the buffer load, see WStream in query_common.h); lo is +64,
// group g +128 g, row block m + m * rb_stride
// TERMS selects the arithmetic: 3 = hi*hi + hi*lo + lo*hi (f32-class, "f16x3"); 2 = weights
// rounded to f16, activations still split (hi*hi + hi*lo, "f16w"); 1 = plain f16 operands ("f16").
__device__ __forceinline__ h8 hload(const WStream &w, int idx16) {
  return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.lane16, idx16 * 16, 0));
}

template <int MR, int PF, int TERMS>
__device__ __forceinline__ void seg_prefetch16(AFrag (&ring)[PF + 1][MR], const WStream &ws, int a,
                                               int rb_stride, int n_groups) {
#pragma unroll
  for (int d = 0; d < PF; ++d)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int p = a + m * rb_stride + min(d, n_groups - 1) * 128;
      ring[d][m].hi = hload(ws, p);
      if (TERMS == 3) ring[d][m].lo = hload(ws, p + 64);
    }
}

// acc += A * B over n_groups k16-steps.  b: LDS address of this lane's point row for column block
// 0 (+ n * 32 * ROWB for block n); the hi slot of group g is ((2g + hh) ^ (p & 15)) << 4 = (2g ^
// swz) << 4 with swz = hh ^ (p & 15), the lo slot sits LO bytes further.
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// hook(g) is called behind the MFMAs of every RS-th k16 group g (g = RS - 1, 2 RS - 1, ...): work that is
// independent of this segment (the table blends of query16's skip-table variant) placed INSIDE the K loop,
// so that its loads are in flight and its VALU work issues between runs of MFMAs
template <int MR, int NR, int PF, int ROWB, int LO_SLOT, int TERMS, typename Hook = NoHook>
__device__ __forceinline__ void seg_main16(f32x16 (&acc)[MR][NR], AFrag (&ring)[PF + 1][MR],
                                           const WStream &ws, int a, int rb_stride, int n_groups,
                                           const unsigned char *b, int swz, Hook hook = Hook()) {
  constexpr int RS = PF + 1;
  // B operands are double-buffered too: the ds_reads of group g+1 are issued before the MFMAs of
  // group g (a lone wave per SIMD has nobody to hide the ~130-cycle LDS latency behind)
  h8 bh[NR], bl[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    bh[n] = *reinterpret_cast<const h8 *>(b + n * 32 * ROWB + (swz << 4));
    if (TERMS >= 2) bl[n] = *reinterpret_cast<const h8 *>(b + n * 32 * ROWB + ((LO_SLOT ^ swz) << 4));
  }
#pragma unroll 1
  for (int g0 = 0; g0 < n_groups; g0 += RS) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int g = g0 + r;
      const int gp = min(g + PF, n_groups - 1);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const int p = a + m * rb_stride + gp * 128;
        ring[(r + PF) % RS][m].hi = hload(ws, p);
        if (TERMS == 3) ring[(r + PF) % RS][m].lo = hload(ws, p + 64);
      }
      const int gn = min(g + 1, n_groups - 1);
      const int boff = ((2 * gn) ^ swz) << 4;
      const int boff_lo = ((LO_SLOT + 2 * gn) ^ swz) << 4;  // lo slots start LO_SLOT slots later
      h8 nh[NR], nl[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        nh[n] = *reinterpret_cast<const h8 *>(b + n * 32 * ROWB + boff);
        if (TERMS >= 2) nl[n] = *reinterpret_cast<const h8 *>(b + n * 32 * ROWB + boff_lo);
      }
      // term-major order: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[r % RS][m].hi, bh[n], acc[m][n], 0, 0, 0);
      if (TERMS >= 2) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[r % RS][m].hi, bl[n], acc[m][n], 0, 0, 0);
      }
      if (TERMS == 3) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[r % RS][m].lo, bh[n], acc[m][n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        bh[n] = nh[n];
        if (TERMS >= 2) bl[n] = nl[n];
      }
      // One wave per SIMD: whatever the wave issues between two runs of MFMAs is time the matrix
      // pipe idles (nobody else feeds it).  So the prefetches of this group -- NV weight fragments,
      // ND LDS reads -- go INTO the run, one behind each MFMA (32 cycles of shadow each).
      {
        constexpr int NV = MR * (TERMS == 3 ? 2 : 1), ND = NR * (TERMS >= 2 ? 2 : 1), NM = MR * NR * TERMS;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (NM > NV + ND) __builtin_amdgcn_sched_group_barrier(0x008, NM - NV - ND, 0);
      }
      if (r == RS - 1) hook(g);
    }
  }
}

// z column: one k16-step whose only non-zero element is k = 0 of lanes 0-31.  The B operands
// are kept as (hi, lo) scalar pairs and widened here (6 live VGPRs instead of 24).
struct ZPair {
  _Float16 hi, lo;
};

__device__ __forceinline__ h8 widen(_Float16 v) {
  h8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (_Float16)0.0f;
  r[0] = v;
  return r;
}

template <int MR, int NR, int TERMS>
__device__ __forceinline__ void gemm_z16(f32x16 (&acc)[MR][NR], const WStream &ws, int az,
                                         const ZPair (&z)[NR]) {
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const h8 ah = hload(ws, az + m * 128);
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const h8 zh = widen(z[n].hi);
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, zh, acc[m][n], 0, 0, 0);
      if (TERMS >= 2)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, widen(z[n].lo), acc[m][n], 0, 0, 0);
      if (TERMS == 3)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hload(ws, az + m * 128 + 64), zh, acc[m][n], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void init_from_bias16(f32x16 &v, const WStream &w32, int bias32,
                                                 float scale) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bq = wload_bias4(w32, bias32 + 8 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[4 * q + i] = bq[i] * scale;
  }
}

// y = lrelu(acc / S)
__device__ __forceinline__ void finish16(f32x16 &v, float inv_scale) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float y = v[t] * inv_scale;
    v[t] = fmaxf(y, y * 0.01f);  // = y > 0 ? y : 0.01 y (SurfaceClassifier.py:58), one op less
  }
}

// C-layout tile (rows 32 rb_local + 8q + 4hh + i of point p = 32 cb + j) -> hidden chunk halves:
// hi slot 4 rb_local + q, lo slot 8 + that, 8 bytes at offset 8 hh inside the slot.
__device__ __forceinline__ void store_hidden16(unsigned char *hb, const f32x16 &v, int rb_local,
                                               int cb, int j, int hh) {
  const int p = 32 * cb + j;
  unsigned char *row = hb + p * kHRow + 8 * hh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 f = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    h4 hi, lo;
    split4(f, hi, lo);
    const int slot = 4 * rb_local + q;
    *reinterpret_cast<h4 *>(row + ((slot ^ (p & 15)) << 4)) = hi;
    *reinterpret_cast<h4 *>(row + (((8 + slot) ^ (p & 15)) << 4)) = lo;
  }
}

// registers 4q .. 4q+3 of a C-layout tile (rows 32 rb_local + 8q + 4hh + i of point p = 32 cb + j):
// y = lrelu(acc / S), split, store into the 128-row chunk buffer of the 96-point tile: 512 bytes
// per point = 16 hi slots | 16 lo slots, row block rb_local (= the wave) owns slots 4 rb_local .. +3
constexpr int kHRow128 = 512;
__device__ __forceinline__ void convert_store_q128(unsigned char *hb, const f32x16 &v, int q, int rb_local,
                                                   int cb, int j, int hh, float inv_scale) {
  const int p = 32 * cb + j;
  f32x4 f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float y = v[4 * q + i] * inv_scale;
    f[i] = fmaxf(y, y * 0.01f);  // SurfaceClassifier.py:58
  }
  h4 hi, lo;
  split4(f, hi, lo);
  unsigned char *row = hb + p * kHRow128 + 8 * hh;
  const int slot = 4 * rb_local + q;
  *reinterpret_cast<h4 *>(row + ((slot ^ (p & 15)) << 4)) = hi;
  *reinterpret_cast<h4 *>(row + (((16 + slot) ^ (p & 15)) << 4)) = lo;
}

// Layers 2 and 3 read their K in chunks of 64 = 16 rows from EACH wave (pack.hip permutes the
// weights to match), so all four waves convert and write a quarter of every chunk in parallel:
// rows 16 half .. +15 of a C-layout tile are registers 8 half .. 8 half + 7; wave `grp` fills K
// group `grp` (hi slots 2 grp, 2 grp + 1).
__device__ __forceinline__ void store_hidden16_part(unsigned char *hb, const f32x16 &v, int half,
                                                    int grp, int cb, int j, int hh) {
  const int p = 32 * cb + j;
  unsigned char *row = hb + p * kHRow + 8 * hh;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t0 = 8 * half + 4 * q;
    const f32x4 f = {v[t0], v[t0 + 1], v[t0 + 2], v[t0 + 3]};
    h4 hi, lo;
    split4(f, hi, lo);
    const int slot = 2 * grp + q;
    *reinterpret_cast<h4 *>(row + ((slot ^ (p & 15)) << 4)) = hi;
    *reinterpret_cast<h4 *>(row + (((8 + slot) ^ (p & 15)) << 4)) = lo;
  }
}

// NB = 32-point column blocks per tile: 3 = the 96-point tile described above (the product);
// 4 = round 1's 128-point tile (160 KB of LDS, 64-row layer-0 chunks computed by two waves per row
// block: 0.42 of the roof over a reconstruction vs 0.44; build it with -DMP16_SGB=0, the
// interleaved prefetches spill next to its 288 accumulator registers: 9.7 ms); 2 = a 64-point tile with half the LDS,
// two workgroups per CU (the partner hides barriers and epilogues, at twice the weight bytes per
// point).  Measured (1 M points, -DMP16_NB=2): f16x3 11.3 ms, plain f16 4.4 vs 4.0 -- weight
// streaming wins.
//
// CS = column split: 1 = four waves, each with all NB column blocks of its rows (one wave per SIMD,
// 512 registers); 2 = EIGHT waves -- wave (rg, cg) owns row group rg (as before) but only the column
// blocks [cg NB/2, +NB/2) -- so every SIMD holds two waves of 256 registers: while one converts a
// chunk (VALU) or waits at a barrier / for the gather, the other keeps the matrix pipe busy.  Both
// waves of a row group stream the same weight fragments (the second one hits the CU's L1); LDS
// traffic, MFMA count and the chunk-buffer layout are unchanged.
template <int COUT, int TERMS, int NB, int CS>
__global__ __launch_bounds__(kThreads16 * CS, NB >= 3 ? CS : 2) void pifu_query16_kernel(
    MlpPack mlp32, MlpPack16 mlp, int fh, int fw, float z_scale, int act, QuerySetDev set) {
  constexpr int C = 256;
  constexpr int NGX = C / 16;        // k16 groups of the feature segment
  constexpr int P = 32 * NB;         // points per tile
  constexpr int THREADS = kThreads16 * CS;
  constexpr int NBW = NB / CS;       // column blocks per wave in layers 1-3
  constexpr int NR0 = NB == 3 ? 3 : NB / 2 / CS;  // column blocks per wave in a layer-0 chunk
  constexpr int PW = P / (4 * CS);   // points gathered per wave
  static_assert(CS == 1 || (CS == 2 && NB == 4), "column split is built for the 128-point tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *xs = smem;
  unsigned char *hb = smem + P * kXRow;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = wave & 3;           // 0..3 = row group of layers 1-3
  const int cbase = (wave >> 2) * NBW;  // first column block of this wave in layers 1-3
  const int j = lane & 31, hh = lane >> 5;
  const int swz = hh ^ (j & 15);
  // layer-0 chunk (2 row blocks x NB column blocks): row block / first column block of this wave
  const int rb0 = CS == 1 ? wv >> 1 : wave & 1;
  const int c0 = CS == 1 ? NR0 * (wv & 1) : wave >> 1;

  // The tiles of all frames of the set form one index space: frame f owns the next
  // ceil(n_f / tile) global tiles.  The owner of a global tile is looked up from the (device-side)
  // counts at the top of every iteration -- eight scalar loads -- instead of keeping a prefix
  // table alive in SGPRs across the whole MLP.
  const float *wbase = mlp32.base;  // last layer (VALU) only
  const WStream w32 = make_wstream(mlp32.base, mlp32.n_floats, lane);        // biases
  const WStream ws = make_wstream(static_cast<const float *>(mlp.base), mlp.n16 * 4, lane);  // f16 fragments

  for (long long gtile = blockIdx.x;; gtile += gridDim.x) {
    int fi = -1;
    long long tile0 = 0;
    {
      long long acc = 0;
      // groups of 8 frames: the 8 count loads of a group are in flight together, and the dynamic group offset
      // keeps the compiler from hoisting all kMaxFrames kernel-argument loads into SGPRs (spills)
      for (int f0 = 0; f0 < set.n; f0 += 8)
#pragma unroll
      for (int fk = 0; fk < 8; ++fk) {
        const int f = f0 + fk;
        if (f < set.n) {
          const long long nf = set.count(f);
          const long long t = (nf + P - 1) / P;
          if (fi < 0 && gtile < acc + t) {
            fi = f;
            tile0 = acc;
          }
          acc += t;
        }
      }
    }
    if (fi < 0) break;  // past the last tile of the last frame
    const QueryItem item = set.item(fi);
    const float *__restrict__ feat = item.feat;
    const float *__restrict__ calib = item.calib;
    float *__restrict__ out = item.out;
    const PointSrc &src = item.src;
    const long long n_pts = src.n_dev ? (long long)*src.n_dev : src.n;
    const long long n0 = (gtile - tile0) * P;

    // ---------------- gather: 32 points per wave, features split into halves ----------------
    ZPair zc[NB];  // z_feat of the column blocks
    {
      float cal[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) cal[i] = calib[i];
      constexpr int GB = 8;  // 32 independent 16-byte loads in flight per lane
#pragma unroll 1
      for (int i0 = 0; i0 < PW; i0 += GB) {
        Taps t[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const long long n = n0 + PW * wave + i0 + u;
          const bool live_n = n < n_pts;
          float px = 0, py = 0, pz = 0, x, y, z;
          uint32_t code;
          if (live_n) load_point(src, n, px, py, pz, code);
          project(cal, px, py, pz, x, y, z);
          t[u] = make_taps(x, y, fh, fw, C, live_n && in_image(x, y));
        }
        f32x4 v[GB][4];
#pragma unroll
        for (int u = 0; u < GB; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            v[u][k] = *reinterpret_cast<const f32x4 *>(feat + t[u].o[k] + 4 * lane);
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int p = PW * wave + i0 + u;
          const f32x4 r = blend(v[u][0], v[u][1], v[u][2], v[u][3], t[u]);
          h4 hi, lo;
          split4(r, hi, lo);
          unsigned char *row = xs + p * kXRow + 8 * (lane & 1);
          const int slot = lane >> 1;
          *reinterpret_cast<h4 *>(row + ((slot ^ (p & 15)) << 4)) = hi;
          *reinterpret_cast<h4 *>(row + (((32 + slot) ^ (p & 15)) << 4)) = lo;
        }
      }
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const long long n = n0 + 32 * cb + j;
        float px = 0, py = 0, pz = 0, x, y, z;
        uint32_t code;
        if (n < n_pts) load_point(src, n, px, py, pz, code);
        project(cal, px, py, pz, x, y, z);
        const float zf = (hh == 0 && n < n_pts) ? __fmul_rn(z, z_scale) : 0.0f;
        zc[cb].hi = (_Float16)zf;
        zc[cb].lo = (_Float16)(zf - (float)zc[cb].hi);
      }
    }
    __syncthreads();

    const unsigned char *xrow = xs + (32 * cbase + j) * kXRow;  // this wave's first column block
    const unsigned char *hrow = hb + (32 * cbase + j) * kHRow;
    const unsigned char *xrow0 = xs + j * kXRow;                // column block 0 (layer-0 chunks)
    ZPair zw[NBW];  // z_feat of this wave's column blocks
#pragma unroll
    for (int n = 0; n < NBW; ++n) zw[n] = zc[cbase + n];

    // ---------------- layers 0 + 1, fused over 64-row chunks of layer 0 ----------------
    f32x16 acc1[4][NBW];  // layer-1 rows [128 wv, +128) x this wave's points (256 / CS registers)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      init_from_bias16(acc1[m][0], w32, mlp32.bias[1] + 32 * (4 * wv + m), mlp.scale[1]);
#pragma unroll
      for (int n = 1; n < NBW; ++n) acc1[m][n] = acc1[m][0];
    }
    {
      const int a0 = mlp.ax[0];                 // [rb][g][part][lane]
      const int rs1 = (kHidden[0] / 16) * 128;  // row-block stride of layer 1's hidden segment
      const int a1 = mlp.ah[1] + (4 * wv) * rs1;
      const float inv0 = 1.0f / mlp.scale[0];
      AFrag ring0[4][1];
      f32x16 acc0[1][NR0];
      const int a1x = mlp.ax[1] + (4 * wv) * NGX * 128;
      if constexpr (NB == 3) {
        // 96-point tile: a chunk is 128 rows of layer 0 = one row block x all three column blocks
        // per wave -- every layer-0 weight fragment is loaded by exactly one wave and feeds 9 MFMAs
        // -- and 8 k16 groups of layer 1; 240 accumulator registers (192 + 48) fit the AGPR file.
        static_assert(CS == 1, "the 96-point tile has no column split");
        const unsigned char *hrow1 = hb + j * kHRow128;
        seg_prefetch16<1, 3, TERMS>(ring0, ws, a0 + wv * NGX * 128, 0, NGX);
        init_from_bias16(acc0[0][0], w32, mlp32.bias[0] + 32 * wv, mlp.scale[0]);
#pragma unroll
        for (int n = 1; n < NB; ++n) acc0[0][n] = acc0[0][0];
#pragma unroll 1
        for (int ck = 0; ck < kHidden[0] / 128; ++ck) {
          const int rb = 4 * ck + wv;
          seg_main16<1, NB, 3, kXRow, 32, TERMS>(acc0, ring0, ws, a0 + rb * NGX * 128, 0, NGX, xrow0, swz);
          AFrag ring1[2][4];
          seg_prefetch16<4, 1, TERMS>(ring1, ws, a1 + ck * 8 * 128, rs1, 8);
          gemm_z16<1, NB, TERMS>(acc0, ws, mlp.az[0] + rb * 128, zc);
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) convert_store_q128(hb, acc0[0][n], q, wv, n, j, hh, inv0);
          const int rbn = min(rb + 4, kHidden[0] / 32 - 4 + wv);
          seg_prefetch16<1, 3, TERMS>(ring0, ws, a0 + rbn * NGX * 128, 0, NGX);
          init_from_bias16(acc0[0][0], w32, mlp32.bias[0] + 32 * rbn, mlp.scale[0]);
#pragma unroll
          for (int n = 1; n < NB; ++n) acc0[0][n] = acc0[0][0];
          __syncthreads();
          seg_main16<4, NB, 1, kHRow128, 16, TERMS>(acc1, ring1, ws, a1 + ck * 8 * 128, rs1, 8, hrow1, swz);
          __syncthreads();
        }
      } else {
      ZPair z0[NR0];
#pragma unroll
      for (int n = 0; n < NR0; ++n) z0[n] = zc[c0 + n];
      seg_prefetch16<1, 3, TERMS>(ring0, ws, a0 + rb0 * NGX * 128, 0, NGX);
      init_from_bias16(acc0[0][0], w32, mlp32.bias[0] + 32 * rb0, mlp.scale[0]);
#pragma unroll
      for (int n = 1; n < NR0; ++n) acc0[0][n] = acc0[0][0];
#pragma unroll 1
      for (int ck = 0; ck < kHidden[0] / 64; ++ck) {
        // layer-0 rows [64 ck + 32 rb0, +32) x column blocks [NR0 cp0, +NR0)
        const int rb = 2 * ck + rb0;
        seg_main16<1, NR0, 3, kXRow, 32, TERMS>(acc0, ring0, ws, a0 + rb * NGX * 128, 0, NGX,
                                         xrow0 + c0 * 32 * kXRow, swz);
        AFrag ring1[2][4];
        seg_prefetch16<4, 1, TERMS>(ring1, ws, a1 + ck * 4 * 128, rs1, 4);
        gemm_z16<1, NR0, TERMS>(acc0, ws, mlp.az[0] + rb * 128, z0);
#pragma unroll
        for (int n = 0; n < NR0; ++n) {
          finish16(acc0[0][n], inv0);
          store_hidden16(hb, acc0[0][n], rb0, c0 + n, j, hh);
        }
        // next chunk's layer-0 operands stream in underneath the layer-1 MFMAs
        const int rbn = min(rb + 2, kHidden[0] / 32 - 2 + rb0);
        seg_prefetch16<1, 3, TERMS>(ring0, ws, a0 + rbn * NGX * 128, 0, NGX);
        init_from_bias16(acc0[0][0], w32, mlp32.bias[0] + 32 * rbn, mlp.scale[0]);
#pragma unroll
        for (int n = 1; n < NR0; ++n) acc0[0][n] = acc0[0][0];
        __syncthreads();
        seg_main16<4, NBW, 1, kHRow, 8, TERMS>(acc1, ring1, ws, a1 + ck * 4 * 128, rs1, 4, hrow, swz);
        __syncthreads();
      }
      }
      // skip segment + z column of layer 1
      AFrag ring1[2][4];
      seg_prefetch16<4, 1, TERMS>(ring1, ws, a1x, NGX * 128, NGX);
      seg_main16<4, NBW, 1, kXRow, 32, TERMS>(acc1, ring1, ws, a1x, NGX * 128, NGX, xrow, swz);
      gemm_z16<4, NBW, TERMS>(acc1, ws, mlp.az[1] + (4 * wv) * 128, zw);
      const float inv1 = 1.0f / mlp.scale[1];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n) finish16(acc1[m][n], inv1);
    }

    // ---------------- layer 2: rows [64 wv, +64) x 128 points ----------------
    f32x16 acc2[2][NBW];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      init_from_bias16(acc2[m][0], w32, mlp32.bias[2] + 32 * (2 * wv + m), mlp.scale[2]);
#pragma unroll
      for (int n = 1; n < NBW; ++n) acc2[m][n] = acc2[m][0];
    }
    {
      const int rs2 = (kHidden[1] / 16) * 128;
      const int a2 = mlp.ah[2] + (2 * wv) * rs2;
      AFrag ring2[2][2];
      seg_prefetch16<2, 1, TERMS>(ring2, ws, a2, rs2, 4);
#pragma unroll
      for (int ck = 0; ck < 8; ++ck) {
#pragma unroll
        for (int n = 0; n < NBW; ++n) store_hidden16_part(hb, acc1[ck >> 1][n], ck & 1, wv, cbase + n, j, hh);
        __syncthreads();
        seg_main16<2, NBW, 1, kHRow, 8, TERMS>(acc2, ring2, ws, a2 + ck * 4 * 128, rs2, 4, hrow, swz);
        if (ck < 7) seg_prefetch16<2, 1, TERMS>(ring2, ws, a2 + (ck + 1) * 4 * 128, rs2, 4);
        __syncthreads();
      }
      const int a2x = mlp.ax[2] + (2 * wv) * NGX * 128;
      seg_prefetch16<2, 1, TERMS>(ring2, ws, a2x, NGX * 128, NGX);
      seg_main16<2, NBW, 1, kXRow, 32, TERMS>(acc2, ring2, ws, a2x, NGX * 128, NGX, xrow, swz);
      gemm_z16<2, NBW, TERMS>(acc2, ws, mlp.az[2] + (2 * wv) * 128, zw);
      const float inv2 = 1.0f / mlp.scale[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n) finish16(acc2[m][n], inv2);
    }

    // ---------------- layer 3: rows [32 wv, +32) x 128 points ----------------
    f32x16 acc3[1][NBW];
    init_from_bias16(acc3[0][0], w32, mlp32.bias[3] + 32 * wv, mlp.scale[3]);
#pragma unroll
    for (int n = 1; n < NBW; ++n) acc3[0][n] = acc3[0][0];
    {
      const int a3 = mlp.ah[3] + wv * (kHidden[2] / 16) * 128;
      AFrag ring3[4][1];
      seg_prefetch16<1, 3, TERMS>(ring3, ws, a3, 0, 4);
#pragma unroll
      for (int ck = 0; ck < 4; ++ck) {
#pragma unroll
        for (int n = 0; n < NBW; ++n) store_hidden16_part(hb, acc2[ck >> 1][n], ck & 1, wv, cbase + n, j, hh);
        __syncthreads();
        seg_main16<1, NBW, 3, kHRow, 8, TERMS>(acc3, ring3, ws, a3 + ck * 4 * 128, 0, 4, hrow, swz);
        if (ck < 3) seg_prefetch16<1, 3, TERMS>(ring3, ws, a3 + (ck + 1) * 4 * 128, 0, 4);
        __syncthreads();
      }
      const int a3x = mlp.ax[3] + wv * NGX * 128;
      seg_prefetch16<1, 3, TERMS>(ring3, ws, a3x, 0, NGX);
      seg_main16<1, NBW, 3, kXRow, 32, TERMS>(acc3, ring3, ws, a3x, 0, NGX, xrow, swz);
      gemm_z16<1, NBW, TERMS>(acc3, ws, mlp.az[3] + wv * 128, zw);
      const float inv3 = 1.0f / mlp.scale[3];
#pragma unroll
      for (int n = 0; n < NBW; ++n) finish16(acc3[0][n], inv3);
    }

    // ---------------- layer 4 (Cout x (128 + C + 1)) on the VALU, f32 ----------------
    // red[part][o][p]: parts 0-3 = hidden rows of wave `part`, parts 4.. = slices of the features
    constexpr int FP = THREADS / P;     // feature slices (threads per point)
    constexpr int SL = 32 / FP;         // 8-channel slots per slice
    float *red = reinterpret_cast<float *>(hb);
    constexpr int K4 = (kHidden[3] + C + 1 + 3) & ~3;  // padded row stride (pack.hip)
    {
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float *w4 = wbase + mlp32.w4 + o * K4 + 32 * wv + 4 * hh;
        float sv[NBW];
#pragma unroll
        for (int n = 0; n < NBW; ++n) sv[n] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wq = *reinterpret_cast<const f32x4 *>(w4 + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < NBW; ++n) sv[n] = fmaf(wq[i], acc3[0][n][4 * q + i], sv[n]);
        }
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          sv[n] += __shfl_xor(sv[n], 32);
          if (hh == 0) red[(wv * COUT + o) * P + 32 * (cbase + n) + j] = sv[n];
        }
      }
      // feature part: thread = (point, slice of the channels); x = hi + lo
      const int p = tid % P, hf = tid / P;  // threads past FP * P (96-point tile: 192 .. 255) sit out
      float sx[COUT];
#pragma unroll
      for (int o = 0; o < COUT; ++o) sx[o] = 0.0f;
#pragma unroll 2
      for (int s = 0; s < (hf < FP ? SL : 0); ++s) {
        const int slot = SL * hf + s;  // 8 channels per slot
        const h8 xh = *reinterpret_cast<const h8 *>(xs + p * kXRow + ((slot ^ (p & 15)) << 4));
        const h8 xl = *reinterpret_cast<const h8 *>(xs + p * kXRow + (((32 + slot) ^ (p & 15)) << 4));
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          const float *w4 = wbase + mlp32.w4 + o * K4 + kHidden[3] + 8 * slot;
#pragma unroll
          for (int e = 0; e < 8; ++e) sx[o] = fmaf(w4[e], (float)xh[e] + (float)xl[e], sx[o]);
        }
      }
#pragma unroll
      for (int o = 0; o < COUT; ++o)
        if (hf < FP) red[((4 + hf) * COUT + o) * P + p] = sx[o];
    }
    __syncthreads();
    for (int idx = tid; idx < COUT * P; idx += THREADS) {
      const int o = idx / P, p = idx % P;
      const long long n = n0 + p;
      if (n < n_pts) {
        float v = (wbase + mlp32.bias[4])[o];
#pragma unroll
        for (int part = 0; part < 4 + FP; ++part) v += red[(part * COUT + o) * P + p];
        float cal[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) cal[i] = calib[i];
        float px, py, pz, x, y, z;
        uint32_t code;
        load_point(src, n, px, py, pz, code);
        project(cal, px, py, pz, x, y, z);
        v = fmaf((wbase + mlp32.w4)[o * K4 + kHidden[3] + C], __fmul_rn(z, z_scale), v);
        v = in_image(x, y) ? activate(v, act) : 0.0f;  // MonoPortNet.py:89
        if (src.packed) {
          const int ix = code & 1023u, iy = (code >> 10) & 1023u, iz = code >> 20;
          out[((long long)iz * src.level_res + iy) * src.level_res + ix] = v;
        } else {
          out[o * src.out_stride + n] = v;
        }
      }
    }
    __syncthreads();  // red / xs are rewritten by the next tile
  }
}

// ---- round 4: the split-precision query through the SKIP TABLE of the feature map ---------------------
// query_table.hip's observation applied here: the products of weights with the sampled feature (layer 0
// and the skip segments of layers 1-4, 42 % of a point's multiply-adds and of this kernel's weight
// stream, which is what bounds it) are taken once per texel by skip_table_kernel -- in exact f32, from
// the f32 weights, whatever the precision of the hidden GEMMs -- and a point blends four table rows.
// The 96 KB of split features, their gather, layer 0's MFMAs and every skip-segment MFMA are gone; what
// is left on the matrix pipe is 1024 -> 512 -> 256 -> 128 on split-f16 operands and the z columns.
//   * layer 0: lrelu(b0 + z w0z + blend(T0 rows)) in f32 on the VALU, split into halves, written to the
//     128-row hidden chunk of the NEXT K step (the chunk buffer is double-buffered now: one barrier per
//     chunk); the table loads of a tile-blend (32 rows x 32 points: 16 loads per lane) are issued one
//     hook ahead INSIDE layer 1's K loop (seg_main16's hook) and blended two k16 groups later;
//   * skip rows of layers 1-3: blended when the layer's accumulators are initialised, S_l (bias + blend),
//     the three column blocks of a row block in flight together (exposed latency, ~1 us per row block:
//     inside the K loops the tile index of an accumulator register would have to be known at run time);
//   * layer 4's feature row is blended in the final reduction.
// Same tile (96 points), same four waves with 512 registers each, one workgroup per CU.
typedef f32x4 TabRows16[4][4];  // [q][tap]
typedef float f32x2_16 __attribute__((ext_vector_type(2)));
constexpr int kQ16TabHb = 96 * kHRow128;                       // one 128-row chunk of the 96-point tile
constexpr int kQ16TabLds = 2 * kQ16TabHb + kHidden[0] * 8;      // two chunks + layer 0's [bias | z weight] table

template <int COUT, int TERMS>
__global__ __launch_bounds__(kThreads16, 1) void pifu_query16_tab_kernel(MlpPack mlp32, MlpPack16 mlp, int fh, int fw,
                                                                        float z_scale, int act, QuerySetDev set) {
  constexpr int NB = 3, P = 96;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *hb = smem;  // hb[2]
  float *bz0 = reinterpret_cast<float *>(smem + 2 * kQ16TabHb);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int swz = hh ^ (j & 15);
  const float *wbase = mlp32.base;
  const WStream w32 = make_wstream(mlp32.base, mlp32.n_floats, lane);
  const WStream ws = make_wstream(static_cast<const float *>(mlp.base), mlp.n16 * 4, lane);
  for (int i = tid; i < kHidden[0]; i += kThreads16) {
    bz0[(i >> 2) * 8 + (i & 3)] = (wbase + mlp32.bias[0])[i];
    bz0[(i >> 2) * 8 + 4 + (i & 3)] = (wbase + mlp32.az[0])[(i >> 5) * 64 + (i & 31)];
  }
  __syncthreads();

  for (long long gtile = blockIdx.x;; gtile += gridDim.x) {
    int fi = -1;
    long long tile0 = 0;
    {
      long long acc = 0;
      // groups of 8 frames: the 8 count loads of a group are in flight together, and the dynamic group offset
      // keeps the compiler from hoisting all kMaxFrames kernel-argument loads into SGPRs (spills)
      for (int f0 = 0; f0 < set.n; f0 += 8)
#pragma unroll
      for (int fk = 0; fk < 8; ++fk) {
        const int f = f0 + fk;
        if (f < set.n) {
          const long long nf = set.count(f);
          const long long t = (nf + P - 1) / P;
          if (fi < 0 && gtile < acc + t) {
            fi = f;
            tile0 = acc;
          }
          acc += t;
        }
      }
    }
    if (fi < 0) break;
    const QueryItem item = set.item(fi);
    const float *__restrict__ calib = item.calib;
    float *__restrict__ out = item.out;
    const PointSrc &src = item.src;
    const long long n_pts = src.n_dev ? (long long)*src.n_dev : src.n;
    const long long n0 = (gtile - tile0) * P;
    const __amdgpu_buffer_rsrc_t prs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(item.l0), 0, fh * fw * kTableRows * 4, 0x00020000);

    // ---------------- this lane's three points (one per column block) ----------------
    int to[NB][4];
    f32x2_16 tw2[NB][4];
    float zf[NB];
    ZPair zc[NB];
    {
      float cal[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) cal[i] = calib[i];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const long long pn = n0 + 32 * n + j;
        float px = 0, py = 0, pz = 0, x, y, z;
        uint32_t code;
        if (pn < n_pts) load_point(src, pn, px, py, pz, code);
        project(cal, px, py, pz, x, y, z);
        zf[n] = pn < n_pts ? __fmul_rn(z, z_scale) : 0.0f;
        const float zm = hh == 0 ? zf[n] : 0.0f;  // B operand of the z column: k = 0 of lanes 0-31
        zc[n].hi = (_Float16)zm;
        zc[n].lo = (_Float16)(zm - (float)zc[n].hi);
        const Taps t = make_taps(x, y, fh, fw, kTableRows, pn < n_pts && in_image(x, y));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          to[n][k] = (int)t.o[k] * 4 + 16 * hh;
          tw2[n][k] = (f32x2_16)(t.w[k]);
        }
      }
    }
    auto rows_issue = [&](TabRows16 &tp, int n, int row0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          tp[q][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, to[n][k], (row0 + 8 * q) * 4, 0));
    };
    auto blend4 = [&](const f32x4 (&t)[4], int n, f32x4 v) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 w = {tw2[n][k][0], tw2[n][k][1], tw2[n][k][0], tw2[n][k][1]};
        v = __builtin_elementwise_fma(t[k], w, v);
      }
      return v;
    };
    // layer-0 rows of block rb for column block n: lrelu(b0 + z w0z + blend), split, -> chunk buffer hbuf
    auto l0_finish = [&](const TabRows16 &tp, int n, int rb, unsigned char *hbuf) {
      f32x16 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 *bz = reinterpret_cast<const f32x4 *>(bz0 + (8 * rb + 2 * q + hh) * 8);
        const f32x4 x = blend4(tp[q], n, __builtin_elementwise_fma(bz[1], (f32x4)(zf[n]), bz[0]));
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * q + i] = x[i];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) convert_store_q128(hbuf, v, q, wv, n, j, hh, 1.0f);
    };
    // acc += S * blend(rows) : the skip rows of a layer, into an accumulator tile that holds S * (...)
    auto skip_finish = [&](f32x16 &acc, const TabRows16 &tp, int n, float S) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 x = blend4(tp[q], n, z4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q + i] = fmaf(x[i], S, acc[4 * q + i]);
      }
    };

    // accumulator tiles of a layer start as S (bias + blended skip rows): the table loads of the NB tiles of
    // one row block in flight together (three sets of 64 registers; this is straight-line code once per
    // tile, before the K loops need their rings -- the rows inside the K loops would want the tile index
    // of an accumulator register at run time)
    auto init_skip = [&](f32x16 (&acc)[NB], int l, int rb, float S) {
      init_from_bias16(acc[0], w32, mlp32.bias[l] + 32 * rb, S);
#pragma unroll
      for (int n = 1; n < NB; ++n) acc[n] = acc[0];
      TabRows16 tp[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) rows_issue(tp[n], n, kTableL[l] + 32 * rb);
#pragma unroll
      for (int n = 0; n < NB; ++n) skip_finish(acc[n], tp[n], n, S);
      __builtin_amdgcn_sched_barrier(0);  // one row block at a time
    };

    // ---------------- layers 0 + 1, fused over 128-row chunks of layer 0 ----------------
    f32x16 acc1[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m) init_skip(acc1[m], 1, 4 * wv + m, mlp.scale[1]);
    {
      const int rs1 = (kHidden[0] / 16) * 128;
      const int a1 = mlp.ah[1] + (4 * wv) * rs1;
      TabRows16 tpa;
      // chunk 0 -> hb[0] (its loads are exposed once per tile)
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        rows_issue(tpa, n, kTableL[0] + 32 * wv);
        l0_finish(tpa, n, wv, hb);
      }
      __syncthreads();
      AFrag ring1[2][4];
      // one chunk: layer 1 += W1[:, chunk ck] * hb[ck & 1]; under it the three tile-blends of chunk ck + 1
      // -> hb[(ck + 1) & 1]: loads issued one hook (two k16 groups = 2300 cycles of MFMAs) ahead.
      // (Also built: the chunk as one straight-line block with a sched_group_barrier pipeline that puts 4
      // non-MFMA instructions behind every MFMA -- slower, 4.60 vs 4.37 ms per 885 k points.)
#pragma unroll 1
      for (int ck = 0; ck < kHidden[0] / 128; ++ck) {
        const bool more = ck < kHidden[0] / 128 - 1;
        const int rbn = 4 * (ck + 1) + wv;
        unsigned char *nxt = hb + ((ck + 1) & 1) * kQ16TabHb;
        seg_prefetch16<4, 1, TERMS>(ring1, ws, a1 + ck * 8 * 128, rs1, 8);
        if (more) rows_issue(tpa, 0, kTableL[0] + 32 * rbn);
        auto hook = [&](int g) {
          if (!more) return;
          if (g == 1) {
            l0_finish(tpa, 0, rbn, nxt);
            rows_issue(tpa, 1, kTableL[0] + 32 * rbn);
          } else if (g == 3) {
            l0_finish(tpa, 1, rbn, nxt);
            rows_issue(tpa, 2, kTableL[0] + 32 * rbn);
          } else if (g == 5) {
            l0_finish(tpa, 2, rbn, nxt);
          }
        };
        seg_main16<4, NB, 1, kHRow128, 16, TERMS>(acc1, ring1, ws, a1 + ck * 8 * 128, rs1, 8,
                                                  hb + (ck & 1) * kQ16TabHb + j * kHRow128, swz, hook);
        __syncthreads();
      }
      gemm_z16<4, NB, TERMS>(acc1, ws, mlp.az[1] + (4 * wv) * 128, zc);
      const float inv1 = 1.0f / mlp.scale[1];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) finish16(acc1[m][n], inv1);
    }

    // ---------------- layer 2: rows [64 wv, +64); K = 512 hidden in 8 chunks of 64 (16 rows from each wave) ----------------
    f32x16 acc2[2][NB];
#pragma unroll
    for (int m = 0; m < 2; ++m) init_skip(acc2[m], 2, 2 * wv + m, mlp.scale[2]);
    {
      const int rs2 = (kHidden[1] / 16) * 128;
      const int a2 = mlp.ah[2] + (2 * wv) * rs2;
      AFrag ring2[2][2];
      seg_prefetch16<2, 1, TERMS>(ring2, ws, a2, rs2, 4);
#pragma unroll
      for (int n = 0; n < NB; ++n) store_hidden16_part(hb, acc1[0][n], 0, wv, n, j, hh);
      __syncthreads();
#pragma unroll
      for (int ck = 0; ck < 8; ++ck) {
        if (ck < 7) {  // the next chunk -> the other buffer (its readers passed the barrier)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            store_hidden16_part(hb + ((ck + 1) & 1) * kQ16TabHb, acc1[(ck + 1) >> 1][n], (ck + 1) & 1, wv, n, j, hh);
        }
        seg_main16<2, NB, 1, kHRow, 8, TERMS>(acc2, ring2, ws, a2 + ck * 4 * 128, rs2, 4,
                                              hb + (ck & 1) * kQ16TabHb + j * kHRow, swz);
        if (ck < 7) seg_prefetch16<2, 1, TERMS>(ring2, ws, a2 + (ck + 1) * 4 * 128, rs2, 4);
        __syncthreads();
      }
      gemm_z16<2, NB, TERMS>(acc2, ws, mlp.az[2] + (2 * wv) * 128, zc);
      const float inv2 = 1.0f / mlp.scale[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) finish16(acc2[m][n], inv2);
    }

    // ---------------- layer 3: rows [32 wv, +32); K = 256 hidden in 4 chunks ----------------
    f32x16 acc3[1][NB];
    init_skip(acc3[0], 3, wv, mlp.scale[3]);
    {
      const int a3 = mlp.ah[3] + wv * (kHidden[2] / 16) * 128;
      AFrag ring3[4][1];
      seg_prefetch16<1, 3, TERMS>(ring3, ws, a3, 0, 4);
#pragma unroll
      for (int n = 0; n < NB; ++n) store_hidden16_part(hb, acc2[0][n], 0, wv, n, j, hh);
      __syncthreads();
#pragma unroll
      for (int ck = 0; ck < 4; ++ck) {
        if (ck < 3) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            store_hidden16_part(hb + ((ck + 1) & 1) * kQ16TabHb, acc2[(ck + 1) >> 1][n], (ck + 1) & 1, wv, n, j, hh);
        }
        seg_main16<1, NB, 3, kHRow, 8, TERMS>(acc3, ring3, ws, a3 + ck * 4 * 128, 0, 4,
                                              hb + (ck & 1) * kQ16TabHb + j * kHRow, swz);
        if (ck < 3) seg_prefetch16<1, 3, TERMS>(ring3, ws, a3 + (ck + 1) * 4 * 128, 0, 4);
        __syncthreads();
      }
      gemm_z16<1, NB, TERMS>(acc3, ws, mlp.az[3] + wv * 128, zc);
      const float inv3 = 1.0f / mlp.scale[3];
#pragma unroll
      for (int n = 0; n < NB; ++n) finish16(acc3[0][n], inv3);
    }

    // ---------------- layer 4 on the VALU: hidden part per wave, table row + z in the reduction ----------------
    float *red = reinterpret_cast<float *>(hb);  // red[wave][o][p]; both chunk buffers are idle (last barrier above)
    constexpr int K4 = (kHidden[3] + 256 + 1 + 3) & ~3;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const float *w4 = wbase + mlp32.w4 + o * K4 + 32 * wv + 4 * hh;
      float sv[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) sv[n] = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wq = *reinterpret_cast<const f32x4 *>(w4 + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < NB; ++n) sv[n] = fmaf(wq[i], acc3[0][n][4 * q + i], sv[n]);
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        sv[n] += __shfl_xor(sv[n], 32);
        if (hh == 0) red[(wv * COUT + o) * P + 32 * n + j] = sv[n];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < COUT * P; idx += kThreads16) {
      const int o = idx / P, p = idx % P;
      const long long n = n0 + p;
      if (n < n_pts) {
        float v = (wbase + mlp32.bias[4])[o];
#pragma unroll
        for (int part = 0; part < 4; ++part) v += red[(part * COUT + o) * P + p];
        float cal[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) cal[i] = calib[i];
        float px, py, pz, x, y, z;
        uint32_t code;
        load_point(src, n, px, py, pz, code);
        project(cal, px, py, pz, x, y, z);
        const bool inside = in_image(x, y);
        const Taps t = make_taps(x, y, fh, fw, kTableRows, inside);
        const float *row = item.l0 + kTableL[4] + o;
        v += fmaf(row[t.o[3]], t.w[3], fmaf(row[t.o[2]], t.w[2], fmaf(row[t.o[1]], t.w[1], __fmul_rn(row[t.o[0]], t.w[0]))));
        v = fmaf((wbase + mlp32.w4)[o * K4 + kHidden[3] + 256], __fmul_rn(z, z_scale), v);
        v = inside ? activate(v, act) : 0.0f;  // MonoPortNet.py:89
        if (src.packed) {
          const int ix = code & 1023u, iy = (code >> 10) & 1023u, iz = code >> 20;
          out[((long long)iz * src.level_res + iy) * src.level_res + ix] = v;
        } else {
          out[o * src.out_stride + n] = v;
        }
      }
    }
    __syncthreads();  // red / hb are rewritten by the next tile
  }
}

template <int COUT, int TERMS>
static int launch_query16_tab_t(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                                long long max_points, bool device_counts, hipStream_t st) {
  constexpr int P = 96;
  auto kern = pifu_query16_tab_kernel<COUT, TERMS>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, kQ16TabLds));
    ctx->lds_attr_done.insert(kern_id);
  }
  if (max_points <= 0) return MP_OK;
  const long long tiles = (max_points + P - 1) / P + (set.n - 1);
  const long long resident = (long long)cus_of(ctx, st);
  const long long grid = device_counts ? (tiles < resident ? tiles : resident)
                                       : (tiles < 8 * resident ? tiles : 8 * resident);
  QuerySetDev dset;
  {
    const int rc_set = compact_query_set(ctx, set, dset);
    if (rc_set != MP_OK) return rc_set;
  }
  const bool prof = 2 * (ctx->prof_used + 1) <= (int)ctx->prof_events.size();
  if (prof) MP_HIP(ctx, hipEventRecord(ctx->prof_events[2 * ctx->prof_used], st));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kThreads16), kQ16TabLds, st, m.pack(), m.pack16(), h, w, z_scale,
                     m.act, dset);
  if (prof) {
    MP_HIP(ctx, hipEventRecord(ctx->prof_events[2 * ctx->prof_used + 1], st));
    ++ctx->prof_used;
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

template <int COUT, int TERMS, int NB, int CS>
static int launch_query16_t(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w,
                            float z_scale, long long max_points, bool device_counts,
                            hipStream_t st) {
  constexpr int P = 32 * NB;
  constexpr int lds = P * (kXRow + (NB == 3 ? kHRow128 : kHRow));
  auto kern = pifu_query16_kernel<COUT, TERMS, NB, CS>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {  // once per kernel and context (= device)
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ctx->lds_attr_done.insert(kern_id);
  }
  if (max_points <= 0) return MP_OK;
  const long long tiles = (max_points + P - 1) / P + (set.n - 1);
  const long long resident = (long long)cus_of(ctx, st) * (NB >= 3 ? 1 : 2);  // 160 / 144 / 80 KB of LDS each
  const long long grid = device_counts ? (tiles < resident ? tiles : resident)
                                       : (tiles < 8 * resident ? tiles : 8 * resident);
  QuerySetDev dset;
  {
    const int rc_set = compact_query_set(ctx, set, dset);
    if (rc_set != MP_OK) return rc_set;
  }
  const bool prof = 2 * (ctx->prof_used + 1) <= (int)ctx->prof_events.size();
  if (prof) MP_HIP(ctx, hipEventRecord(ctx->prof_events[2 * ctx->prof_used], st));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kThreads16 * CS), lds, st, m.pack(), m.pack16(),
                     h, w, z_scale, m.act, dset);
  if (prof) {
    MP_HIP(ctx, hipEventRecord(ctx->prof_events[2 * ctx->prof_used + 1], st));
    ++ctx->prof_used;
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_query16(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                   long long max_points, bool device_counts, hipStream_t st) {
  if (m.c != 256) return fail(ctx, MP_ERR_UNSUPPORTED, "f16x3 query kernel is built for C = 256");
  // Through the skip tables when every map has one -- for f16x3 only by default: measured on 885 k lattice points /
  // a 16-frame reconstruction, f16x3 5.14 -> 4.37 ms / 1.99 -> 1.93 ms per frame; f16w 3.63 -> 3.68 / 1.53 -> 1.69
  // and plain f16 2.69 -> 3.28 / 1.25 -> 1.54 LOSE (fewer MFMAs and fragments to save, the same blends to pay:
  // a wave that is alone on its SIMD pays for every VALU and VMEM instruction with matrix-pipe time).
  // MONOPORT_TAB16=all routes every precision (tests, tools/tab16_probe.py), =off none.
  const char *t16 = getenv("MONOPORT_TAB16");
  const bool want_tab = t16 && t16[0] == 'a' ? true : t16 && t16[0] == 'o' ? false : m.precision == MP_PREC_F16X3;
  QuerySet tset;
  if (want_tab && kQ16Nb == 3 && kQ16Cs == 1 && find_skip_tables(ctx, m, set, h, w, tset)) {
    if ((long long)h * w * kTableRows * 4 >= (1LL << 31))
      return fail(ctx, MP_ERR_UNSUPPORTED, "table query: %dx%d map is too large for 32-bit table offsets", h, w);
#define MP_Q16TCASE(CO, PREC, TERMS) \
  if (m.cout == CO && m.precision == PREC)   \
    return launch_query16_tab_t<CO, TERMS>(ctx, m, tset, h, w, z_scale, max_points, device_counts, st);
    MP_Q16TCASE(1, MP_PREC_F16X3, 3)
    MP_Q16TCASE(3, MP_PREC_F16X3, 3)
    MP_Q16TCASE(1, MP_PREC_F16W, 2)
    MP_Q16TCASE(3, MP_PREC_F16W, 2)
    MP_Q16TCASE(1, MP_PREC_F16, 1)
    MP_Q16TCASE(3, MP_PREC_F16, 1)
#undef MP_Q16TCASE
  }
#define MP_Q16CASE(CO, PREC, TERMS)                                                         \
  if (m.cout == CO && m.precision == PREC)                                                 \
    return launch_query16_t<CO, TERMS, kQ16Nb, kQ16Cs>(ctx, m, set, h, w, z_scale, max_points, device_counts, st);
  MP_Q16CASE(1, MP_PREC_F16X3, 3)
  MP_Q16CASE(3, MP_PREC_F16X3, 3)
  MP_Q16CASE(1, MP_PREC_F16W, 2)
  MP_Q16CASE(3, MP_PREC_F16W, 2)
  MP_Q16CASE(1, MP_PREC_F16, 1)
  MP_Q16CASE(3, MP_PREC_F16, 1)
#undef MP_Q16CASE
  return fail(ctx, MP_ERR_UNSUPPORTED, "f16 query kernels: Cout must be 1 or 3");
}

}  // namespace mp
