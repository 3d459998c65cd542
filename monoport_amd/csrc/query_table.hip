// The fused f32 query of a netG head through the SKIP TABLE of its feature map (round 3).
//
// SurfaceClassifier (heads/SurfaceClassifier.py:39-71) multiplies weights with the SAMPLED FEATURE
// in every layer: layer 0's 1024 x 256 block and the skip connections of layers 1-4 (:55,
// 512 + 256 + 128 + Cout rows) -- 42 % of a point's 2,363,906 FLOP.  The sampled feature is a
// bilinear blend of four texels (geometry.py:4-16), and a linear map commutes with that blend:
//     W . sum_k w_k F[texel_k]  =  sum_k w_k (W . F[texel_k]).
// So skip_table_kernel takes those 1921 x 256 products ONCE per texel and frame
// (table[y][x][r], 16 GFLOP and 126 MB for a 128 x 128 map) and pifu_query_tab_kernel gives a point
// four rows of the table to blend -- with grid_sample's own FMA chain and zero padding -- instead of
// 492 k multiply-adds on the MFMAs.  What is left per point is the hidden-to-hidden work
// (1024 -> 512 -> 256 -> 128 -> Cout: 1,377,026 FLOP) and the z column.  The feature tile and its
// gather disappear with the products (LDS: 20 KB instead of 48 KB); the price is 31 KB of gathered
// table rows per point, L2-served because the lattice points of a tile share texels.
//
// The field differs from the plain kernels' (query.hip / query_small.hip) by f32 rounding only --
// the same products summed in another order: measured 1-3e-7 on the goldens, the same distance
// to the fp64 oracle as the plain path and as the reference itself (tests/test_query_gpu.py::
// test_skip_table_*); the bar is 1e-4.  The plain kernels stay the default of the C-ABI: a launch
// comes here only if every one of its feature maps has a registered table (mp_skip_table).
//
// The query kernel that ships is pifu_query_tabws_kernel (round 4: the waves of a workgroup specialised into
// MFMA-only consumers and table-blending producers, below).  Round 3's kernel, in which every wave did
// everything (0.72 of the roof: two workgroups sharing a SIMD's matrix pipe drift into phase), and the
// timing-experiment builds of both are in the history (last present at commit 78cb450; DESIGN_HISTORY.md 4.1c / 4.1d
// record their measurements).
#include <cstdlib>
#include <cstring>

#include "mp_internal.h"
#include "query_common.h"
#include "query_mfma.h"

#pragma clang fp contract(off)

namespace mp {

constexpr int kTabPts = 32;
constexpr int kTabQ = 4;             // row groups of a 32-row block fetched per round of table loads (16 loads = 64 registers per lane)
constexpr int kTabHbRow = 128 * 4;  // bytes per point of a 128-row hidden chunk

typedef f32x4 TabRows[kTabQ][4];  // [q][tap]: rows 8 (q0 + q) + 4 h .. + 3 of a 32-row block, four texels

// ---- the query with the waves of a workgroup SPECIALISED ------------------------------------------------
// In round 3's kernel every wave alternated between MFMA phases (the K loops) and phases that
// keep the matrix pipe idle for that wave: table-row loads, blends, bias / z / leaky ReLU, LDS stores.
// The second workgroup of the CU is supposed to fill them, but two workgroups that share a SIMD's
// matrix pipe drift into phase (both slow down while both want the pipe, both leave it together):
// 90 us per tile against 73 us of MFMA work, 0.72 of the roof.  Here a workgroup is
//   * 4 CONSUMER waves (one per SIMD): nothing but the K loops of layers 1-3 -- operands from LDS and
//     the weight stream -- the leaky ReLU of their own accumulators and the layer-4 partial sums.
//     They never see a point, a texel or the table;
//   * 4 PRODUCER waves (one per SIMD, VMEM + VALU only): everything that depends on the point.  A lane
//     owns one point and, per job, the 16 rows of a 32-row block that an accumulator register of the
//     consumers stands for (the MFMA C layout), and produces
//       - layer 0's output chunk by chunk:  lrelu(b0 + blend(T0 rows) + z w0z)  -> H0[2] (point-major,
//         the B operand of layer 1's K loop), one chunk AHEAD of the consumers;
//       - "pieces" = b_l + blend(T_l rows) + z w_lz for the row blocks of layers 1-3 -> PB, which the
//         consumers ADD to their accumulators at a fixed point of the K loop (addition order is free);
//       - the final reduction: bias + the consumers' partial sums + layer 4's blended row + z, the
//         activation, the in-image mask and the store / scatter.
// All eight waves meet at 14 barriers per tile in lock step -- S0-S7: layer 0 -> 1, one 128-row chunk
// each (256 MFMAs per consumer wave); T0-T3: layer 2 over 128-row K pairs (128 MFMAs); U0-U1: layer 3
// (64) -- and data written between two barriers is read after the second.  The K pairs of layers 2 / 3
// borrow a layer-0 chunk buffer, which is idle in those intervals.  The producers' work is placed so
// that it never has to finish inside a short interval: whole jobs in the long S intervals (6.8 us of
// MFMA work each), split jobs (loads issued in one interval, blended and written in a later one) in T / U.  Registers: both roles stay under 128, so a CU holds
// two workgroups = 16 waves, four per SIMD.
constexpr int kWsThreads = 512;
constexpr int kWsX = 0;                                 // X[3]: [32 points][128 rows] f32, swizzled (16 KB each):
                                                        //   X[0], X[1] layer-0 chunks; X[2], X[1] the 128-row K pairs of layers 2 / 3
constexpr int kWsXBytes = kTabPts * kTabHbRow;
constexpr int kWsPB = 3 * kWsXBytes;                    // PB: [consumer wave][q][lane] f32x4 (16 KB)
constexpr int kWsRed = kWsPB + 4 * 4 * 64 * 16;         // red[wave][o][p]
constexpr int kWsBZ0 = kWsRed + 4 * 3 * kTabPts * 4;    // layer 0: [row / 4][bias x 4 | z weight x 4] (8 KB)
constexpr int kWsB13 = kWsBZ0 + kHidden[0] * 8;         // biases of layers 1-3 (896 floats)
constexpr int kWsZv = kWsB13 + (kHidden[1] + kHidden[2] + kHidden[3]) * 4;  // zvec[2][32]: z * z_scale of the tile's points, by tile parity
constexpr int kWsTend = kWsZv + 2 * kTabPts * 4;        // tile_end[f]: tiles of frames 0..f (prefix sums), 16 ints
// What the producers need to know about a frame, staged once per workgroup.  Per tile they would otherwise chain
// scalar loads from the kernel arguments (dynamic frame index; a miss in the scalar cache goes to the kernarg
// buffer), the device-side point counter and the calibration: 25 k cycles per tile for the point load alone
// (tools/tab_ws_stamp_probe.py).
struct WsFrame {  // per frame
  float cal[12];
  const void *pts;  // packed node codes (WsShared::lattice) or explicit coordinates
  float *out;
  const float *l0;
  int npts, pad;
};
struct WsShared {  // the point layout / lattice of the launch, shared by its frames (QuerySetDev)
  long long sn, sc, out_stride;
  int stride, level_res;
  float res_final, half_step, bmin[3], blen[3];
  int lattice, pad;
};
static_assert(sizeof(WsFrame) == 80 && sizeof(WsShared) == 72, "two workgroups per CU: 2 x kWsLds <= 160 KB");
constexpr int kWsFrames = kWsTend + kMaxFrames * 4;
constexpr int kWsShared = kWsFrames + kMaxFrames * (int)sizeof(WsFrame);
constexpr int kWsLds = kWsShared + (int)sizeof(WsShared);
static_assert(2 * kWsLds <= 160 * 1024, "two workgroups per CU");

// timing experiments (tools/ablate.py; wrong results): the producers do no work / no barriers;
// kWsConsumerPrio: s_setprio of the consumer waves
#define WS_SYNC() __syncthreads()
#define WS_MARK(i)
constexpr int kWsTabAux = 0;  // cache policy of the table-row loads (2 = nt: stream past the L2-resident weights)
constexpr int kWsFinishAt = 8;  // the interval (8 = T0) in which the producers store the previous tile's outputs
constexpr int kWsSetupAt = 7;  // ... and in which they project the next tile's points
constexpr int kWsP4First = 1;  // S7: the rows of piece 4 requested before (1) or after (0) the next tile's points are set up
constexpr int kWsR4Late = 1;  // layer 4's skip row of the next tile's points: in T3 (0: with the projection)
constexpr int kWsSetupPrio = 3;  // s_setprio of the producers while they set up the next tile's points / store the outputs
constexpr int kWsProducerPrio = 0;  // s_setprio of the producer waves
constexpr int kWsConsumerPrio = 1;  // measured: 0.819 -> 0.830 of the roof on 885 k points (3 = the same)

struct TileLoc {
  int fi;  // frame of the tile, -1: past the end
  long long n0;
};
// tile -> (frame, first point) from the prefix sums of the frames' tile counts in LDS (filled once per
// workgroup: the counts live on the device but do not change during the launch)
__device__ __forceinline__ TileLoc locate_tile(const int *tend, long long gtile, int lane) {
  TileLoc loc;
  const int mine = tend[lane & (kMaxFrames - 1)];
  const unsigned long long below = __ballot(lane < kMaxFrames && gtile >= (long long)mine);
  const int fi = __builtin_amdgcn_readfirstlane(__popcll(below));
  if (fi >= kMaxFrames) {
    loc.fi = -1;
    loc.n0 = 0;
    return loc;
  }
  const int start = fi ? __builtin_amdgcn_readfirstlane(tend[fi - 1]) : 0;
  loc.fi = fi;
  loc.n0 = (gtile - start) * kTabPts;
  return loc;
}

// what a producer lane knows about its point
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct WsPoint {
  int to[4];     // byte offsets of the four table rows (+ this lane's half of a row group)
  f32x2 tw[2];   // grid_sample weights (0 outside the map / dead point) as two register PAIRS: v_pk_fma_f32 broadcasts
                 // either half of a pair (op_sel), a lone register would need a v_mov per use
  float zf;      // z * z_scale (0 for a dead point)
  float r4;      // layer 4's blended skip row of output 2 * (producer wave) + (lane >> 5)
  uint32_t code; // packed lattice coordinates (scatter address of the output)
  int ok;        // bit 0: the point exists, bit 1: it projects into the image
};

template <int COUT>
__global__ __launch_bounds__(kWsThreads, 4) void pifu_query_tabws_kernel(MlpPack mlp, int fh, int fw, float z_scale,
                                                                         int act, QuerySetDev set) {
  constexpr int P = kTabPts;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int swz = h ^ (j & 15);
  const WStream ws = make_wstream(mlp.base, mlp.n_floats, lane);
  float *bz0 = reinterpret_cast<float *>(smem + kWsBZ0);
  float *b13 = reinterpret_cast<float *>(smem + kWsB13);
  int *tend = reinterpret_cast<int *>(smem + kWsTend);
  for (int i = tid; i < kHidden[0]; i += kWsThreads) {
    bz0[(i >> 2) * 8 + (i & 3)] = (mlp.base + mlp.bias[0])[i];
    bz0[(i >> 2) * 8 + 4 + (i & 3)] = (mlp.base + mlp.az[0])[(i >> 5) * 64 + (i & 31)];  // az: [row block][64 lanes], rows in lanes 0-31
  }
  for (int i = tid; i < kHidden[1] + kHidden[2] + kHidden[3]; i += kWsThreads) {
    const int l = i < kHidden[1] ? 1 : i < kHidden[1] + kHidden[2] ? 2 : 3;
    const int r = i - (l == 1 ? 0 : l == 2 ? kHidden[1] : kHidden[1] + kHidden[2]);
    b13[i] = (mlp.base + mlp.bias[l])[r];
  }
  if (tid < kMaxFrames) {
    // tiles of frames 0..tid: every lane reads ITS frame's device-side counter (one round trip for the
    // workgroup), the prefix sum runs over the lanes -- a loop over the frames per lane was 2 x 16 dependent
    // loads in front of the first tile (tens of microseconds of a 1 ms level-0 launch)
    int mine = 0;
    if (tid < set.n) mine = (int)set.count(tid);
    int acc = (mine + P - 1) / P;
#pragma unroll
    for (int o = 1; o < kMaxFrames; o <<= 1) {
      const int up = __shfl_up(acc, o);
      if (tid >= o) acc += up;
    }
    tend[tid] = acc;
    if (tid < set.n) {
      const QueryItemDev &it = set.it[tid];
      WsFrame &fr = reinterpret_cast<WsFrame *>(smem + kWsFrames)[tid];
#pragma unroll
      for (int i = 0; i < 12; ++i) fr.cal[i] = it.calib[i];
      fr.pts = it.pts;
      fr.out = it.out;
      fr.l0 = it.l0;
      if (tid == 0) {
        WsShared &sh = *reinterpret_cast<WsShared *>(smem + kWsShared);
        sh.sn = set.sn;
        sh.sc = set.sc;
        sh.out_stride = set.out_stride;
        sh.stride = set.stride;
        sh.level_res = set.level_res;
        sh.res_final = set.res_final;
        sh.half_step = set.half_step;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          sh.bmin[i] = set.bmin[i];
          sh.blen[i] = set.blen[i];
        }
        sh.lattice = set.lattice;
      }
      fr.npts = mine;
    }
  }
  __syncthreads();
  const long long n_tiles = __builtin_amdgcn_readfirstlane(tend[kMaxFrames - 1]);
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own
  // L2.  Tiles that are neighbours in the list share texels (table rows) and frames, so XCD x takes the
  // CONTIGUOUS eighth [x T / 8, (x + 1) T / 8) of the tile list and its workgroups walk through it side
  // by side: a table row fetched into an L2 is reused there instead of being fetched by all eight.
  const int xcd = blockIdx.x & 7, n_xcd = gridDim.x < 8 ? (int)gridDim.x : 8;  // XCDs this launch reaches
  const long long tile_step = ((int)gridDim.x - xcd + 7) >> 3;                   // its workgroups on this XCD
  const long long tile_first = n_tiles * xcd / n_xcd + (blockIdx.x >> 3), tile_end = n_tiles * (xcd + 1) / n_xcd;

  if (wv < 4) {
    // =============================== consumers: the K loops ===============================
    if (kWsConsumerPrio) __builtin_amdgcn_s_setprio(kWsConsumerPrio);
    const int rs1 = (kHidden[0] / 8) * 64, rs2 = (kHidden[1] / 8) * 64;
    const int a1 = mlp.ah[1] / 4 + (4 * wv) * rs1;
    const int a2 = mlp.ah[2] / 4 + (2 * wv) * rs2;
    const int a3 = mlp.ah[3] / 4 + wv * (kHidden[2] / 8) * 64;
    unsigned char *x = smem + kWsX;
    const unsigned char *xrow = x + j * kTabHbRow;
    // a piece of this wave ([wave][q][lane] f32x4 in any 16 KB region: 0-2 = X[0..2], 3 = PB)
    const f32x4 *piece0 = reinterpret_cast<const f32x4 *>(smem + kWsX) + (wv * 4) * 64 + lane;
    const float *zvec = reinterpret_cast<const float *>(smem + kWsZv);
    float zb[1];
    // acc += piece of this wave (the producers' bias + blended skip rows of one row block), then the
    // z column as one MFMA k-step (B operand: z of the points in lanes 0-31, 0 in lanes 32-63)
    auto add_piece = [&](f32x16 (&acc)[1][1], int az_rb, int region) {
      float az[1];
      az[0] = wload32(ws, az_rb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 pc = piece0[region * (kWsXBytes / 16) + q * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][0][4 * q + i] = acc[0][0][4 * q + i] + pc[i];
      }
      gemm_z<1, 1>(acc, az, zb);
    };
    // accumulator tile m of this wave -> rows [32 m, +32) of a 128-row K buffer, point-major: store_hidden's
    // addresses from ONE register -- slot 8 m + 2 q + h, swizzled = (j * 512 | swz << 4) ^ ((8 m + 2 q) << 4).
    // (16 precomputed offsets would sit in registers for the whole kernel: the compiler hoists them out of
    // the tile loop and spills four of them -- reloads with s_waitcnt vmcnt(0) in front of the weight
    // prefetch; the asm below pins the computation inside the loop)
    int st0 = j * kTabHbRow | (swz << 4);
    auto store_k = [&](int region, const f32x16 &v, int m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        *reinterpret_cast<f32x4 *>(x + region * kWsXBytes + (st0 ^ ((8 * m + 2 * q) << 4))) = o;
      }
    };
    WS_SYNC();  // the producers' first chunk
    int par = 0;
    for (long long gtile = tile_first;; gtile += tile_step, par ^= 1) {
      if (gtile >= tile_end) break;
      asm volatile("" : "+v"(st0));
      zb[0] = h == 0 ? zvec[par * P + j] : 0.0f;
      // ---------------- S0-S7: layer 1 += W1[:, chunk k] * (layer-0 chunk k in X[k & 1]); piece k / 2 added on odd k ----------------
      f32x16 acc1[4][1];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc1[m][0][t] = 0.0f;
      {
        f32x4 ring1[kPrefetch1 + 1][4];
        seg_prefetch<4, kPrefetch1>(ring1, ws, a1, rs1, 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          seg_main<4, 1, kPrefetch1, kTabHbRow>(acc1, ring1, ws, a1 + k * 16 * 64, rs1, 16, xrow + (k & 1) * kWsXBytes, swz);
          if (k < 7) seg_prefetch<4, kPrefetch1>(ring1, ws, a1 + (k + 1) * 16 * 64, rs1, 16);
          if (k & 1) add_piece(*reinterpret_cast<f32x16(*)[1][1]>(&acc1[k >> 1]), mlp.az[1] + (4 * wv + (k >> 1)) * 64, 3);
          if (k == 7) {
#pragma unroll
            for (int m = 0; m < 4; ++m) lrelu(acc1[m][0]);
            // layer 2 reads K = hidden 1 in four 128-row pairs = the accumulators of waves 0..3: pairs 0, 1 ->
            // X[2], X[0] here (last read in U0 of the previous tile / S6), pairs 2, 3 -> X[1], PB at the top of T0
            if (wv < 2) {
#pragma unroll
              for (int m = 0; m < 4; ++m) store_k(wv == 0 ? 2 : 0, acc1[m][0], m);
            }
          }
          WS_SYNC();
        }
      }
      // ---------------- T0-T3: layer 2, rows [64 wv, +64); K pair p = hidden-1 rows [128 p, +128) in X[2], X[0], X[1], PB ----------------
      f32x16 acc2[2][1];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc2[m][0][t] = 0.0f;
      if (wv >= 2) {  // behind the barrier of S7: X[1] (chunk 7) and PB (piece 3) have been read
#pragma unroll
        for (int m = 0; m < 4; ++m) store_k(wv == 2 ? 1 : 3, acc1[m][0], m);
      }
      {
        f32x4 ring2[2][2];
        seg_prefetch<2, 1>(ring2, ws, a2, rs2, 16);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const int buf = pr == 0 ? 2 : pr == 1 ? 0 : pr == 2 ? 1 : 3;
          seg_main<2, 1, 1, kTabHbRow>(acc2, ring2, ws, a2 + pr * 16 * 64, rs2, 16, xrow + buf * kWsXBytes, swz);
          if (pr < 3) seg_prefetch<2, 1>(ring2, ws, a2 + (pr + 1) * 16 * 64, rs2, 16);
          if (pr == 2) add_piece(*reinterpret_cast<f32x16(*)[1][1]>(&acc2[0]), mlp.az[2] + (2 * wv) * 64, 2);  // piece 4: written to X[2] in T1
          if (pr == 3) {
            add_piece(*reinterpret_cast<f32x16(*)[1][1]>(&acc2[1]), mlp.az[2] + (2 * wv + 1) * 64, 0);  // piece 5: written to X[0] in T2
#pragma unroll
            for (int m = 0; m < 2; ++m) lrelu(acc2[m][0]);
            if (wv < 2) {  // layer 3's first K pair = hidden-2 rows [0, 128) = waves 0 and 1 -> X[2] (piece 4 was read in T2)
#pragma unroll
              for (int m = 0; m < 2; ++m) store_k(2, acc2[m][0], 2 * wv + m);
            }
          }
          WS_SYNC();
        }
      }
      // ---------------- U0-U1: layer 3, rows [32 wv, +32), K = 256 hidden in 2 pairs ----------------
      f32x16 acc3[1][1], acc3o;  // layer 3 on two accumulator chains (even / odd k-steps), added before piece 6
#pragma unroll
      for (int t = 0; t < 16; ++t) acc3[0][0][t] = acc3o[t] = 0.0f;
      {
        f32x4 ring3[4][1];
        seg_prefetch<1, 3>(ring3, ws, a3, 0, 16);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          if (pr == 0 && wv >= 2) {  // the second pair = waves 2 and 3 -> X[1] (last read in T2)
#pragma unroll
            for (int m = 0; m < 2; ++m) store_k(1, acc2[m][0], 2 * (wv - 2) + m);
          }
          seg_main_split<3, kTabHbRow>(acc3[0][0], acc3o, ring3, ws, a3 + pr * 16 * 64, 16, xrow + (pr ? 1 : 2) * kWsXBytes, swz);
          if (pr == 0) seg_prefetch<1, 3>(ring3, ws, a3 + 16 * 64, 0, 16);
          if (pr == 1) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc3[0][0][t] += acc3o[t];
            add_piece(acc3, mlp.az[3] + wv * 64, 3);  // piece 6: written to PB in U0
            lrelu(acc3[0][0]);
            // layer 4 on the VALU: this wave's 32 hidden rows; the producers finish the sum
            float *red = reinterpret_cast<float *>(smem + kWsRed);
            constexpr int K4 = (kHidden[3] + 256 + 1 + 3) & ~3;
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
              const float *w4 = (mlp.base + mlp.w4) + o * K4 + 32 * wv + 4 * h;
              float s0 = 0.0f;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 wq = *reinterpret_cast<const f32x4 *>(w4 + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) s0 = fmaf(wq[i], acc3[0][0][4 * q + i], s0);
              }
              s0 += __shfl_xor(s0, 32);
              if (h == 0) red[(wv * COUT + o) * P + j] = s0;
            }
          }
          WS_SYNC();
        }
      }
    }
    WS_SYNC();  // the producers' last final pass reads `red` behind this one
  } else {
    // =============================== producers: everything per point ===============================
    const int pw = wv - 4;  // partner of consumer wave pw
    if (kWsProducerPrio) __builtin_amdgcn_s_setprio(kWsProducerPrio);
    unsigned char *h0 = smem + kWsX;
    int st1 = (j * kTabHbRow | (swz << 4)) ^ (pw << 7);  // see the consumers' store_k
    f32x4 *piece0 = reinterpret_cast<f32x4 *>(smem + kWsX) + (pw * 4) * 64 + lane;  // + region * 16 KB (3 = PB)

    // (the frame index is wave-uniform, but not provably so for the compiler: without the readfirstlanes every
    // table load is wrapped in a waterfall loop over the descriptor)
    const WsFrame *frames = reinterpret_cast<const WsFrame *>(smem + kWsFrames);
    const WsShared &sh = *reinterpret_cast<const WsShared *>(smem + kWsShared);
    auto table_rsrc = [&](int fi) {
      const unsigned long long a = reinterpret_cast<unsigned long long>(frames[__builtin_amdgcn_readfirstlane(fi)].l0);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>((unsigned long long)hi << 32 | lo), 0,
                                               fh * fw * kTableRows * 4, 0x00020000);
    };
    float *zvec = reinterpret_cast<float *>(smem + kWsZv);
    // The point of this lane in tile (fi, n0), in two steps so that its round trips hide behind other work:
    // point_load (the coordinates: one dependent load -- the point count comes from LDS) an interval before
    // point_setup (projection, texels, weights, and layer 4's skip row for finish_tile).
    struct RawPoint {
      float px, py, pz;
      uint32_t code;
      int live;
    };
    auto point_load = [&](int fi, long long n0, RawPoint &rp) {  // load_point (query_common.h) on the staged frame
      const WsFrame &fr = frames[fi];
      const long long n = n0 + j;
      rp.px = rp.py = rp.pz = 0.0f;
      rp.code = 0;
      rp.live = n < fr.npts;
      if (rp.live) {
        if (sh.lattice) {
          rp.code = static_cast<const uint32_t *>(fr.pts)[n];
        } else {
          const float *pts = static_cast<const float *>(fr.pts);
          rp.px = pts[n * sh.sn];
          rp.py = pts[n * sh.sn + sh.sc];
          rp.pz = pts[n * sh.sn + 2 * sh.sc];
        }
      }
    };
    auto point_setup = [&](int fi, const RawPoint &rp, WsPoint &pt, int zpar) {
      const WsFrame &fr = frames[fi];
      float cal[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) cal[i] = fr.cal[i];
      float px = rp.px, py = rp.py, pz = rp.pz;
      if (sh.lattice && rp.live) {  // lattice_coord (query_common.h), the same operation sequence
        const int idx[3] = {(int)(rp.code & 1023u), (int)((rp.code >> 10) & 1023u), (int)(rp.code >> 20)};
        float c3[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float c = (float)(idx[a] * sh.stride);
          const float u = __fadd_rn(__fdiv_rn(c, sh.res_final), sh.half_step);
          c3[a] = __fadd_rn(__fmul_rn(u, sh.blen[a]), sh.bmin[a]);
        }
        px = c3[0];
        py = c3[1];
        pz = c3[2];
      }
      float x, y, z;
      project(cal, px, py, pz, x, y, z);
      pt.zf = rp.live ? __fmul_rn(z, z_scale) : 0.0f;
      if (pw == 0 && h == 0) zvec[zpar * P + j] = pt.zf;  // the consumers' B operand of the z column
      const bool inside = in_image(x, y);
      const Taps t = make_taps(x, y, fh, fw, kTableRows, rp.live && inside);
      pt.code = rp.code;
      pt.ok = (rp.live ? 1 : 0) | (inside ? 2 : 0);
      pt.r4 = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pt.to[k] = (int)t.o[k] * 4 + 16 * h;
        pt.tw[k >> 1][k & 1] = t.w[k];
      }
    };
    constexpr int K4 = (kHidden[3] + 256 + 1 + 3) & ~3;
    // layer 4's blended skip row of this lane's output (what finish_tile adds): its own round trip, taken in an
    // interval in which the producers have nothing else to do
    auto point_r4 = [&](int fi, WsPoint &pt) {
      if (2 * pw + h < COUT && (pt.ok & 1)) {
        const float *row = frames[fi].l0 + kTableL[4] + 2 * pw + h;
        const int o0 = (pt.to[0] - 16 * h) >> 2, o1 = (pt.to[1] - 16 * h) >> 2, o2 = (pt.to[2] - 16 * h) >> 2,
                  o3 = (pt.to[3] - 16 * h) >> 2;
        const float bias4 = (mlp.base + mlp.bias[4])[2 * pw + h];
        const float wz4 = (mlp.base + mlp.w4)[(2 * pw + h) * K4 + kHidden[3] + 256];
        const float r = fmaf(row[o3], pt.tw[1][1],
                             fmaf(row[o2], pt.tw[1][0], fmaf(row[o1], pt.tw[0][1], __fmul_rn(row[o0], pt.tw[0][0]))));
        pt.r4 = r + fmaf(wz4, pt.zf, bias4);  // + the z column and the bias
      }
    };
    auto setup_point = [&](int fi, long long n0, WsPoint &pt, int zpar) {
      RawPoint rp;
      point_load(fi, n0, rp);
      point_setup(fi, rp, pt, zpar);
      point_r4(fi, pt);
    };
    // a job = one 32-row block of the table for this lane's point: 16 loads, then the blend
    auto job_issue = [&](TabRows &tp, const __amdgpu_buffer_rsrc_t &prs, const WsPoint &pt, int row0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          tp[q][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, pt.to[k], (row0 + 8 * q) * 4,
                                                                                     kWsTabAux));
        }
    };
    // grid_sample's chain on four rows at once, started from `v0` (the bias, or bias + z column): whole
    // f32x4 fused multiply-adds -- v_pk_fma_f32, half the VALU issue slots of scalar code, and every
    // VALU instruction of a producer costs the consumer on its SIMD matrix-pipe time
    auto blend4 = [&](const f32x4 (&t)[4], const WsPoint &pt, f32x4 v0) {
      f32x4 v = v0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // weights kept as register PAIRS: v_pk_fma_f32 takes them without a v_mov per use
        const f32x4 w = (f32x4)(pt.tw[k >> 1][k & 1]);
        v = __builtin_elementwise_fma(t[k], w, v);
      }
      return v;
    };
    // layer-0 chunk ck of this lane's point -> X[buf]: this wave's row block 4 ck + pw;
    // lrelu(b0 + z w0z + blend) with bias and z weights from LDS
    auto chunk_finish = [&](const TabRows &tp, const WsPoint &pt, int ck, int buf) {
      const int rb = 4 * ck + pw;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 *bz = reinterpret_cast<const f32x4 *>(bz0 + (8 * rb + 2 * q + h) * 8);
        f32x4 v = __builtin_elementwise_fma(bz[1], (f32x4)(pt.zf), bz[0]);
        v = blend4(tp[q], pt, v);
        v = __builtin_elementwise_max(v, v * 0.01f);  // leaky ReLU (SurfaceClassifier.py:58)
        *reinterpret_cast<f32x4 *>(h0 + buf * kWsXBytes + (st1 ^ ((2 * q) << 4))) = v;  // slot 8 pw + 2 q + h, swizzled
      }
    };
    // piece = bias + blend of row block rb of layer l (b13 offset boff) -> PB; the z column is the consumers'
    auto piece_finish = [&](const TabRows &tp, const WsPoint &pt, int boff, int region) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(b13 + boff + 8 * q + 4 * h);
        piece0[region * (kWsXBytes / 16) + q * 64] = blend4(tp[q], pt, b);
      }
    };
    // the output of tile (fi, n0), whose points are `pt`: bias + the consumers' partial sums + layer 4's blended
    // row + z column.  Lane (j, h) of producer wave pw stores output 2 pw + h of point j; everything that
    // needs the point was loaded by setup_point a tile earlier -- no global load stands between the
    // partial sums and the store (with them, this was the longest producer interval of the tile: the
    // consumers waited 7 % of their time at its barrier, tools/tab_ws_stamp_probe.py)
    const int my_o = 2 * pw + h;
    auto finish_tile = [&](const WsPoint &pt, int fi, long long n0) {
      if (my_o < COUT && (pt.ok & 1)) {
        const WsFrame &fr = frames[fi];
        const float *red = reinterpret_cast<const float *>(smem + kWsRed);
        float v = red[my_o * P + j];
#pragma unroll
        for (int part = 1; part < 4; ++part) v += red[(part * COUT + my_o) * P + j];
        v += pt.r4;
        v = (pt.ok & 2) ? activate(v, act) : 0.0f;  // MonoPortNet.py:89
        if (sh.lattice) {
          const int ix = pt.code & 1023u, iy = (pt.code >> 10) & 1023u, iz = pt.code >> 20;
          fr.out[((long long)iz * sh.level_res + iy) * sh.level_res + ix] = v;
        } else {
          fr.out[my_o * sh.out_stride + n0 + j] = v;
        }
      }
    };

    WsPoint cur = {}, nxt, prev = {};
    const TileLoc no_tile = {-1, 0};
    TileLoc loc = tile_first < tile_end ? locate_tile(tend, tile_first, lane) : no_tile;
    TileLoc loc_ahead = tile_first + tile_step < tile_end ? locate_tile(tend, tile_first + tile_step, lane) : no_tile;
    __amdgpu_buffer_rsrc_t prs_cur = table_rsrc(loc.fi >= 0 ? loc.fi : 0), prs_nxt;
    TabRows tp;  // the rows of the job in flight; it may cross a barrier (and the end of a tile)
    if (loc.fi >= 0) {
      setup_point(loc.fi, loc.n0, cur, 0);
      job_issue(tp, prs_cur, cur, kTableL[0] + 32 * pw);
      chunk_finish(tp, cur, 0, 0);
    }
    prs_nxt = prs_cur;
    nxt = cur;
    WS_SYNC();  // the first chunk
    int prev_fi = -1, par = 0;
    long long prev_n0 = 0;
    for (long long gtile = tile_first;; gtile += tile_step, par ^= 1) {
      if (loc.fi < 0) break;
      const TileLoc loc_n = loc_ahead;  // located in the previous tile's T0: S0 is the producers' longest interval
      // ---------------- S0-S7 ----------------
      // Even k: piece k / 2 (row block 4 pw + k / 2 of layer 1) -> PB, read in S(k + 1) -- its loads were issued at
      // the end of the interval before, so the interval holds ONE load round trip (chunk k + 1), like the odd ones.
      // S7, where the producers have next to nothing to do: the previous tile's outputs, the next tile's points.
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k == kWsFinishAt && k < 7 && prev_fi >= 0) finish_tile(prev, prev_fi, prev_n0);
        if (k == kWsSetupAt && k < 7 && loc_n.fi >= 0) {
          setup_point(loc_n.fi, loc_n.n0, nxt, par ^ 1);
          prs_nxt = table_rsrc(loc_n.fi);
        }
        if (k == 0) asm volatile("" : "+v"(st1));
        if (!(k & 1)) {
          job_issue(tp, prs_cur, cur, kTableL[1] + 32 * (4 * pw + (k >> 1)));
          piece_finish(tp, cur, 32 * (4 * pw + (k >> 1)), 3);
        }
        if (k < 7) {  // chunk k + 1 -> X[(k + 1) & 1] (read in S(k + 1); last read in S(k - 1) / U1)
          job_issue(tp, prs_cur, cur, kTableL[0] + 32 * (4 * (k + 1) + pw));
          chunk_finish(tp, cur, k + 1, (k + 1) & 1);
        } else {
          // S7, where the producers have the least to do: the next tile's points (needed in U0) and the
          // previous tile's outputs.  Nothing is in flight at its top, so the outputs (which wait for a few
          // spilled values: s_waitcnt vmcnt(0)) go between the point load and the table rows of piece 4.
          WS_MARK(0);
          // the next tile's points are a chain of dependent steps (count, load, divide, project, texels) on the
          // critical path of the interval: they run at raised priority -- at the consumers' priority or below, a
          // producer instruction waits ~100 cycles for an issue slot between the MFMAs
          if (kWsSetupPrio) __builtin_amdgcn_s_setprio(kWsSetupPrio);
          RawPoint raw_n = {};
          if (kWsSetupAt == 7 && loc_n.fi >= 0) point_load(loc_n.fi, loc_n.n0, raw_n);
          WS_MARK(1);
          if (kWsFinishAt == 7 && prev_fi >= 0) finish_tile(prev, prev_fi, prev_n0);
          if (kWsP4First) job_issue(tp, prs_cur, cur, kTableL[2] + 32 * (2 * pw));  // piece 4 (layer 2, row block 2 pw) in flight
          WS_MARK(2);
          if (kWsSetupAt == 7 && loc_n.fi >= 0) {
            point_setup(loc_n.fi, raw_n, nxt, par ^ 1);
            if (!kWsR4Late) point_r4(loc_n.fi, nxt);
            prs_nxt = table_rsrc(loc_n.fi);
          }
          if (kWsSetupPrio) __builtin_amdgcn_s_setprio(kWsProducerPrio);
          WS_MARK(3);
          if (!kWsP4First) job_issue(tp, prs_cur, cur, kTableL[2] + 32 * (2 * pw));  // with all 64 row registers free until here
        }
        WS_SYNC();
      }
      // ---------------- T0-T3, U0-U1: split jobs -- loads in one interval, blend + write in a later one ----------------
      if (kWsFinishAt == 8 && prev_fi >= 0) {  // T0: nothing else to do
        if (kWsSetupPrio) __builtin_amdgcn_s_setprio(kWsSetupPrio);
        finish_tile(prev, prev_fi, prev_n0);
        if (kWsSetupPrio) __builtin_amdgcn_s_setprio(kWsProducerPrio);
      }
      loc_ahead = gtile + 2 * tile_step < tile_end ? locate_tile(tend, gtile + 2 * tile_step, lane) : no_tile;
      WS_SYNC();  // T0: every region holds a K pair of layer 2 or is being filled with one
      piece_finish(tp, cur, kHidden[1] + 32 * (2 * pw), 2);         // T1: piece 4 -> X[2] (pair 0 was read in T0; read in T2)
      job_issue(tp, prs_cur, cur, kTableL[2] + 32 * (2 * pw + 1));  //     piece 5 in flight
      WS_SYNC();
      piece_finish(tp, cur, kHidden[1] + 32 * (2 * pw + 1), 0);  // T2: piece 5 -> X[0] (pair 1 was read in T1; read in T3)
      job_issue(tp, prs_cur, cur, kTableL[3] + 32 * pw);         //     piece 6 in flight
      WS_SYNC();
      if (kWsSetupAt == 7 && kWsR4Late && loc_n.fi >= 0) point_r4(loc_n.fi, nxt);  // T3: nothing else to do
      WS_SYNC();  // T3: the consumers read pair 3 in PB and piece 5
      piece_finish(tp, cur, kHidden[1] + kHidden[2] + 32 * pw, 3);            // U0: piece 6 -> PB (read in U1)
      if (loc_n.fi >= 0) job_issue(tp, prs_nxt, nxt, kTableL[0] + 32 * pw);  //     the next tile's chunk 0 in flight
      WS_SYNC();
      if (loc_n.fi >= 0) {
        chunk_finish(tp, nxt, 0, 0);  // U1: -> X[0] (piece 5 was read in T3)
      }
      WS_SYNC();
      prev_fi = loc.fi;
      prev_n0 = loc.n0;
      prev = cur;
      loc = loc_n;
      cur = nxt;
      prs_cur = prs_nxt;
    }
    WS_SYNC();
    if (prev_fi >= 0) finish_tile(prev, prev_fi, prev_n0);
  }
}

template <int COUT>
static int launch_query_tabws_t(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                                long long max_points, bool device_counts, hipStream_t st) {
  if (max_points <= 0) return MP_OK;
  auto kern = pifu_query_tabws_kernel<COUT>;
  const void *kern_id = reinterpret_cast<const void *>(kern);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, kWsLds));
    ctx->lds_attr_done.insert(kern_id);
  }
  const long long tiles = (max_points + kTabPts - 1) / kTabPts + (set.n - 1);
  // MONOPORT_QUERY_WGS_PER_CU (measurement switch, results do not depend on it): 2 = the kernel takes both
  // workgroup slots of every CU (all of its LDS); 1 = half of them, so that a launch of ANOTHER stream -- a second
  // slot's query, or its encoder's convolutions -- can be resident next to it
  static const int wgs_per_cu = [] {
    const char *e = getenv("MONOPORT_QUERY_WGS_PER_CU");
    const int v = e ? atoi(e) : 2;
    return v == 1 ? 1 : 2;
  }();
  const long long resident = (long long)cus_of(ctx, st) * wgs_per_cu;
  // persistent: a workgroup's producers run one chunk ahead of its consumers ACROSS tiles, so a
  // workgroup should see several tiles; never more workgroups than are resident at once
  long long grid = tiles < resident ? tiles : resident;
  // A launch for ONE frame comes from a per-frame caller (the reference's recon stage, RTL/main.py:389-395) whose
  // neighbour stage is running netG.filter of the next frame on another stream: it gets 8 x the resident workgroups,
  // each with 1 / 8 of the tiles, so that the launch gives its workgroup slots back a few at a time and the encoder's
  // chain of small convolutions gets onto the chip between them instead of waiting for the whole level.  Measured on
  // the per-frame stage pipeline (tools/per_frame_overlap_probe.py, profiles/r06r_per_frame_grid_mult.txt): 130-131 ->
  // 136-137 recon/s at 8 x; the query launches alone and the batched headline do not notice (x 1 .. x 16 within noise).
  // MONOPORT_QUERY_GRID_MULT overrides the factor for every launch (measurement switch; results do not depend on it).
  static const int grid_mult_env = [] {
    const char *e = getenv("MONOPORT_QUERY_GRID_MULT");
    const int v = e ? atoi(e) : 0;
    return v < 0 ? 0 : v > 64 ? 64 : v;
  }();
  const int grid_mult = grid_mult_env ? grid_mult_env : set.n == 1 ? 8 : 1;
  if (grid_mult > 1) grid = tiles < resident * grid_mult ? tiles : resident * grid_mult;
  QuerySetDev dset;
  {
    const int rc_set = compact_query_set(ctx, set, dset);
    if (rc_set != MP_OK) return rc_set;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kWsThreads), kWsLds, st, m.pack(), h, w, z_scale, m.act, dset);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_query_tab(mp_ctx *ctx, const Mlp &m, const QuerySet &set, int h, int w, float z_scale,
                     long long max_points, bool device_counts, hipStream_t st) {
  if ((long long)h * w * kTableRows * 4 >= (1LL << 31))
    return fail(ctx, MP_ERR_UNSUPPORTED, "table query: %dx%d map is too large for 32-bit table offsets", h, w);
  if (m.cout == 1) return launch_query_tabws_t<1>(ctx, m, set, h, w, z_scale, max_points, device_counts, st);
  if (m.cout == 3) return launch_query_tabws_t<3>(ctx, m, set, h, w, z_scale, max_points, device_counts, st);
  return fail(ctx, MP_ERR_UNSUPPORTED, "table query: Cout in {1,3}");
}

// ---- the skip table -------------------------------------------------------------------------------
// table[texel][kTableL[l] + r] = sum_c W_l[r][hidden_l + c] F[texel][c]  for the feature segment of
// every layer l = 0..4 (no bias, no z column), layers 0-3 on the MFMAs in the K order of the plain
// kernels (the packed fragment streams mlp.ax[l]), layer 4 on the VALU.  One workgroup = 64 texels
// staged into LDS exactly like a tile of sampled points; the 60 row blocks of layers 0-3 go to the
// waves in pairs, each against both column blocks.  16 GFLOP and 126 MB per 128^2 map.
template <int COUT>
__global__ __launch_bounds__(kQueryThreads, 2) void skip_table_kernel(MlpPack mlp, const float *__restrict__ feat,
                                                                    long long texels, float *__restrict__ table) {
  constexpr int C = 256, ROWB = C * 4, NGX = C / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *xs = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int swz = h ^ (j & 15);
  const WStream ws = make_wstream(mlp.base, mlp.n_floats, lane);
  const long long n_tiles = texels / 64;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t0 = tile * 64;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int p = 16 * wv + i;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(feat + (t0 + p) * C + 4 * lane);
      *reinterpret_cast<f32x4 *>(xs + p * ROWB + ((lane ^ (p & 15)) << 4)) = v;
    }
    __syncthreads();
    const unsigned char *xrow = xs + j * ROWB;
#pragma unroll 1
    for (int pair = wv; pair < 30; pair += 4) {
      const int rb = 2 * pair;  // row block among the 32 + 16 + 8 + 4 of layers 0-3 (pairs never straddle)
      const int l = rb < 32 ? 0 : rb < 48 ? 1 : rb < 56 ? 2 : 3;
      const int rl = rb - (l == 0 ? 0 : l == 1 ? 32 : l == 2 ? 48 : 56);
      f32x16 acc[2][2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[m][n][t] = 0.0f;
      const int a = mlp.ax[l] / 4 + rl * NGX * 64;
      f32x4 ring[2][2];
      seg_prefetch<2, 1>(ring, ws, a, NGX * 64, NGX);
      seg_main<2, 2, 1, ROWB>(acc, ring, ws, a, NGX * 64, NGX, xrow, swz);
      const int row0 = kTableL[l] + 32 * rl;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 o = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
            *reinterpret_cast<f32x4 *>(table + (t0 + 32 * n + j) * kTableRows + row0 + 32 * m + 8 * q + 4 * h) = o;
          }
    }
    // layer 4's feature segment: thread = (output o, texel p)
    constexpr int K4 = (kHidden[3] + C + 1 + 3) & ~3;
    if (tid < 64 * COUT) {
      const int o = tid / 64, p = tid % 64;
      float s = 0.0f;
#pragma unroll 4
      for (int slot = 0; slot < C / 4; ++slot) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + p * ROWB + ((slot ^ (p & 15)) << 4));
        const f32x4 wq = *reinterpret_cast<const f32x4 *>((mlp.base + mlp.w4) + o * K4 + kHidden[3] + 4 * slot);
#pragma unroll
        for (int i = 0; i < 4; ++i) s = fmaf(wq[i], xv[i], s);
      }
      table[(t0 + p) * kTableRows + kTableL[4] + o] = s;
    }
    __syncthreads();  // xs is restaged by the next tile
  }
}

int launch_skip_table(mp_ctx *ctx, const Mlp &m, const float *feat_hwc, int h, int w, float *table,
                    hipStream_t st) {
  const long long texels = (long long)h * w;
  if (m.c != 256 || texels % 64 || (m.cout != 1 && m.cout != 3))
    return fail(ctx, MP_ERR_UNSUPPORTED, "skip table: C = 256 heads with Cout in {1,3} and H * W a multiple of 64; got C=%d Cout=%d %dx%d",
                m.c, m.cout, h, w);
  constexpr int lds = 64 * 256 * 4;
  const void *kern_id = m.cout == 1 ? reinterpret_cast<const void *>(skip_table_kernel<1>)
                                    : reinterpret_cast<const void *>(skip_table_kernel<3>);
  if (!ctx->lds_attr_done.count(kern_id)) {
    MP_HIP(ctx, hipFuncSetAttribute(kern_id, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ctx->lds_attr_done.insert(kern_id);
  }
  const long long tiles = texels / 64, resident = (long long)cus_of(ctx, st) * 2;
  const dim3 grid((unsigned)(tiles < resident ? tiles : resident));
  if (m.cout == 1)
    hipLaunchKernelGGL(skip_table_kernel<1>, grid, dim3(kQueryThreads), lds, st, m.pack(), feat_hwc, texels, table);
  else
    hipLaunchKernelGGL(skip_table_kernel<3>, grid, dim3(kQueryThreads), lds, st, m.pack(), feat_hwc, texels, table);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
