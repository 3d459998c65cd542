// Coarse-to-fine occupancy reconstruction on the GPU: the replacement for
// implicit_seg.functional.Seg3dLossless(faster=True) (un-vendored dependency of the reference:
// requirements.txt:15; constructed at RTL/main.py:185-195, called at :392-394).
//
// Per level r -> 2r-1 (the scheme recalled in SURVEY.md section 5.7 and restated on the CPU in
// oracle/pifu_oracle.py:seg3d_lossless, which this file must match bit for bit):
//   1. upsample_classify: trilinear (align_corners=True) upsample of the previous level's
//      occupancy into this level's volume, and a 1-bit "boundary" flag per node = the upsampled
//      binary mask (occ > balance) is strictly between 0 and 1.  Flags are produced with a wave
//      ballot: one 64-bit word per 64 consecutive x.
//   2. select_compact: binary dilation of the flags by the level's box (9^3 / 7^3 / 3^3) as
//      shift-OR on the 64-bit words, minus the nodes evaluated at earlier levels, popcount +
//      wave prefix sum + one atomic per wave -> a packed (x | y<<10 | z<<20) point list.
//   3. the fused query kernel (query.hip) reads the list and its device-side count, and scatters
//      exact occupancies straight into the level volume.
// No host synchronisation anywhere: counts stay on the device (`status`).
#include <cstdlib>
#include <cstring>

#include "mp_internal.h"

// Reference parity is op-order parity: keep every a*b+c exactly as written (the HIP headers
// define __fmul_rn & co. as plain operators, which hipcc would otherwise contract into FMAs).
// Fused multiply-adds are requested explicitly (fmaf / MFMA) where they are wanted.
#pragma clang fp contract(off)

namespace mp {

typedef unsigned long long u64;

struct LevelBufs {
  float *occ;   // [r][r][r]
  u64 *bnd;     // boundary bits [r][r][w64]
  u64 *ev;      // evaluated bits [r][r][w64]
};

static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
static inline int words64(int r) { return (r + 63) / 64; }

int octree_box_of_level(int level) { return level == 1 ? 9 : level == 2 ? 7 : 3; }

size_t recon_scratch_bytes(const int *res, int n_levels) {
  size_t total = 0;
  for (int l = 0; l < n_levels; ++l) {
    const size_t r = res[l];
    if (l < n_levels - 1) total += align256(r * r * r * sizeof(float));
    total += 2 * align256(r * r * words64(res[l]) * sizeof(u64));
  }
  const size_t rl = res[n_levels - 1];
  total += align256(rl * rl * rl * sizeof(uint32_t));  // packed point list (worst case: every node)
  return total + 4096;
}

// The housekeeping kernels of a level serve ALL frames of a batch in one launch (blockIdx.z = frame; round 5: per
// frame they were 40-odd launches of 3-25 us between two query launches, each too small to fill the chip): the
// per-frame pointers travel by value.
struct FrameBufs {
  const float *prev[kMaxFrames];   // previous level's volume
  float *cur[kMaxFrames];          // this level's volume
  u64 *bnd[kMaxFrames];            // boundary flags of this level
  const u64 *ev_prev[kMaxFrames];  // evaluated bits of the previous level
  u64 *ev[kMaxFrames];             // evaluated bits of this level
  uint32_t *packed[kMaxFrames];    // point list
  int32_t *count[kMaxFrames];      // its length (device side)
  int32_t *flag[kMaxFrames];       // level 0: "anything above the threshold" (status[0])
};
static_assert(sizeof(FrameBufs) <= 2048 + 64, "kernel argument");
constexpr int kHouseChunk = kMaxFrames;  // frames per housekeeping launch (see launch_recon)

// ---- level 0 ---------------------------------------------------------------------------------
__global__ void iota_nodes_kernel(int r, FrameBufs fb, int w64, int y_major) {
  uint32_t *__restrict__ packed = fb.packed[blockIdx.z];
  u64 *__restrict__ ev = fb.ev[blockIdx.z];
  int32_t *__restrict__ count = fb.count[blockIdx.z];
  const int total = r * r * r;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) *count = total;
  if (t < total) {
    const int x = t % r, a = (t / r) % r, b = t / (r * r);
    const int y = y_major ? b : a, z = y_major ? a : b;  // list order (see select_compact_kernel)
    packed[t] = (uint32_t)x | ((uint32_t)y << 10) | ((uint32_t)z << 20);
  }
  if (t < r * r * w64) {  // every node of level 0 is evaluated
    const int w = t % w64;
    const int nbits = min(64, r - 64 * w);
    ev[t] = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
  }
}

__global__ void any_above_kernel(FrameBufs fb, int n, float balance) {
  const float *__restrict__ occ = fb.cur[blockIdx.z];
  int32_t *__restrict__ flag = fb.flag[blockIdx.z];
  int hit = 0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
    hit |= occ[t] > balance;
  if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// mp_recon_batch_early: per frame [2f] = the frame's non-empty flag, [2f + 1] = 1 if any of its level-0 values
// differs from what the caller's query_func returned for the same nodes (NaN != NaN counts as a difference, as in
// the tensor comparison this replaces).  flags must be zeroed; one block column per frame.
struct EarlyBufs {
  const float *occ[kMaxFrames];
  const float *expect[kMaxFrames];
  const int32_t *flag[kMaxFrames];
};
__global__ void early_flags_kernel(EarlyBufs eb, int n, int32_t *__restrict__ flags) {
  const int f = blockIdx.z;
  const float *__restrict__ occ = eb.occ[f];
  const float *__restrict__ expect = eb.expect[f];
  int hit = 0;
  if (expect)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) hit |= occ[t] != expect[t];
  if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(flags + 2 * f + 1, 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[2 * f] = *eb.flag[f];
}

// ---- upsample + boundary flags ---------------------------------------------------------------
// One lane per PARENT node (z0, y0, x0): it loads the 2x2x2 parent cell once and emits the up to
// eight fine nodes (2 z0 + oz, 2 y0 + oy, 2 x0 + ox) that interpolate inside it -- 8 loads per 8
// outputs instead of 8 per output (the kernel was VALU-bound on address arithmetic).  Interpolation
// order z, then y, then x with weights 0.5/0.5 -- the exact sequence of oracle upsample2x (axis 0
// first), so values agree bit for bit.  A wave covers 64 parents = 128 fine x positions = two
// boundary words per fine row.  grid = (ceil(rp * ceil(rp/64) / 4), rp): blockIdx.y is z0.
__device__ __forceinline__ u64 spread32(u64 x) {  // bit i -> bit 2i
  x &= 0xffffffffull;
  x = (x | (x << 16)) & 0x0000ffff0000ffffull;
  x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
  x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

// `rule` (MP_FINAL_*): which nodes get a flag -- 0: 0 < upsampled mask < 1 (every level of the lossless schedule);
// 1: upsampled mask == 0.5 exactly, i.e. half of the corners with non-zero weight are inside (the last level of
// the "upstream" schedule); 2: none (the last level of the "interpolate" schedule).
__device__ __forceinline__ bool boundary_flag(int in, int all, int rule) {
  return rule == 0 ? (in > 0 && in < all) : rule == 1 ? (2 * in == all) : false;
}

__global__ __launch_bounds__(256) void upsample_classify_kernel(FrameBufs fb, int rp, int r, float balance,
                                                                int w64, int rule) {
  const float *__restrict__ prev = fb.prev[blockIdx.z];
  float *__restrict__ cur = fb.cur[blockIdx.z];
  u64 *__restrict__ bnd = fb.bnd[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int wpx = (rp + 63) >> 6;  // waves per parent row
  const unsigned plane_item = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (plane_item >= (unsigned)(rp * wpx)) return;
  const int wx = plane_item % (unsigned)wpx;
  const int y0 = plane_item / (unsigned)wpx;
  const int z0 = blockIdx.y;
  const int x0 = 64 * wx + lane;
  const bool valid = x0 < rp;

  float pv[2][2][2];  // [dz][dy][dx]
  bool pin[2][2][2];
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int zz = min(z0 + dz, rp - 1), yy = min(y0 + dy, rp - 1), xx = min(x0 + dx, rp - 1);
        const float v = valid ? prev[((long long)zz * rp + yy) * rp + xx] : 0.0f;
        pv[dz][dy][dx] = v;
        pin[dz][dy][dx] = v > balance;
      }

#pragma unroll
  for (int oz = 0; oz < 2; ++oz) {
    const int z = 2 * z0 + oz;
    if (z >= r) continue;  // uniform: only the last parent plane has no odd child
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      const int y = 2 * y0 + oy;
      if (y >= r) continue;
      float vx[2];
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float vy[2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
          vy[dy] = oz ? 0.5f * pv[0][dy][dx] + 0.5f * pv[1][dy][dx] : pv[0][dy][dx];
        vx[dx] = oy ? 0.5f * vy[0] + 0.5f * vy[1] : vy[0];
      }
      // corners with non-zero trilinear weight: dz <= oz, dy <= oy, dx <= ox
      int in0 = 0, in1 = 0;  // inside-corner counts for ox = 0 / the extra ones of ox = 1
#pragma unroll
      for (int dz = 0; dz <= oz; ++dz)
#pragma unroll
        for (int dy = 0; dy <= oy; ++dy) {
          in0 += pin[dz][dy][0];
          in1 += pin[dz][dy][1];
        }
      const int all0 = (1 + oz) * (1 + oy);
      const float v_even = vx[0];
      const float v_odd = 0.5f * vx[0] + 0.5f * vx[1];
      const bool has_odd = valid && 2 * x0 + 1 < r;
      float *row = cur + ((long long)z * r + y) * r;
      if (valid) row[2 * x0] = v_even;
      if (has_odd) row[2 * x0 + 1] = v_odd;
      const bool f_even = valid && boundary_flag(in0, all0, rule);
      const bool f_odd = has_odd && boundary_flag(in0 + in1, 2 * all0, rule);
      const u64 be = __ballot(f_even), bo = __ballot(f_odd);
      if (lane == 0) {
        u64 *words = bnd + ((long long)z * r + y) * w64 + 2 * wx;
        words[0] = spread32(be) | (spread32(bo) << 1);
        if (2 * wx + 1 < w64) words[1] = spread32(be >> 32) | (spread32(bo >> 32) << 1);
      }
    }
  }
}

// ---- dilate, drop evaluated nodes, compact ----------------------------------------------------
// D = half width of the dilation box (4 / 3 / 1 for the 9^3 / 7^3 / 3^3 boxes).  The (2D+1) rows of
// one dz are loaded branch-free (out-of-range rows read row 0 and are masked) so the 3 (2D+1)
// word loads are in flight together: on the small grids of levels 1-2 the kernel is a handful of
// waves and purely latency-bound.
// The ORDER of the point list is free (results are scattered by node code) and decides what the query
// kernel's table / feature gathers find in cache: the list is filled in the order of the work items
// (one atomic per wave, waves start in index order), and an item is (w, y in slab, z, slab) with SLABS
// OF kYSlab ROWS OF y OUTERMOST.  For the reference's turntable cameras (rotation about the y axis,
// RTL/main.py) world y is image y, so the points of a slab sample kYSlab / 2 + 1 texel rows whatever
// their x and z -- a few MB of table rows that stay in the XCD's L2 -- and consecutive tiles of 32 points
// stay compact in image space.  z-major order (MONOPORT_OCTREE_ORDER=z, the round-3 order) revisits a
// texel once per z plane, 25 MB of rows apart.  Inside a slab the items walk y fastest, so a wave still
// reads whole rows of flag words next to each other (y alone outermost made them 10 KB apart:
// select_compact 43 -> 125 us at 257^3).
constexpr int kYSlab = 8;
static bool octree_y_major() {
  const char *e = getenv("MONOPORT_OCTREE_ORDER");
  return !(e && e[0] == 'z');
}

template <int D, bool YMAJOR>
__global__ __launch_bounds__(256) void select_compact_kernel(FrameBufs fb, int rp, int w64p, int r, int w64) {
  const u64 *__restrict__ bnd = fb.bnd[blockIdx.z];
  const u64 *__restrict__ ev_prev = fb.ev_prev[blockIdx.z];
  u64 *__restrict__ ev = fb.ev[blockIdx.z];
  uint32_t *__restrict__ packed = fb.packed[blockIdx.z];
  int32_t *__restrict__ count = fb.count[blockIdx.z];
  // items: z-major r * r * w64; slab order ceil(r / kYSlab) * kYSlab * r * w64 (rows past r are empty)
  const unsigned n_items = (unsigned)((YMAJOR ? (r + kYSlab - 1) / kYSlab * kYSlab : r) * r * w64);  // <= 1024 * 1023 * 16
  const unsigned item = blockIdx.x * blockDim.x + threadIdx.x;
  u64 sel = 0;
  int w = 0, y = 0, z = 0;
  bool live = item < n_items;
  if (live) {
    w = item % (unsigned)w64;
    if (YMAJOR) {
      const unsigned t = item / (unsigned)w64;
      z = (t / kYSlab) % (unsigned)r;
      y = (t % kYSlab) + kYSlab * (t / (unsigned)(kYSlab * r));
      live = y < r;
    } else {
      y = (item / (unsigned)w64) % (unsigned)r;
      z = item / (unsigned)(w64 * r);
    }
  }
  if (live) {
    u64 acc = 0;
    for (int dz = -D; dz <= D; ++dz) {
      const int zz = z + dz;
      if (zz < 0 || zz >= r) continue;
      u64 c[2 * D + 1], lo[2 * D + 1], hi[2 * D + 1];
#pragma unroll
      for (int k = 0; k < 2 * D + 1; ++k) {
        const int yy = y + k - D;
        const bool ok = yy >= 0 && yy < r;
        const u64 *row = bnd + ((long long)zz * r + (ok ? yy : 0)) * w64;
        const u64 m = ok ? ~0ull : 0ull;
        c[k] = row[w] & m;
        lo[k] = w > 0 ? row[w - 1] & m : 0ull;
        hi[k] = w < w64 - 1 ? row[w + 1] & m : 0ull;
      }
      u64 cc = 0, ll = 0, hh = 0;  // OR over the rows first: the x dilation is linear in OR
#pragma unroll
      for (int k = 0; k < 2 * D + 1; ++k) {
        cc |= c[k];
        ll |= lo[k];
        hh |= hi[k];
      }
      u64 hd = cc;
#pragma unroll
      for (int s = 1; s <= D; ++s) hd |= (cc << s) | (ll >> (64 - s)) | (cc >> s) | (hh << (64 - s));
      acc |= hd;
    }
    // nodes already evaluated: even (z, y, x) that were evaluated one level up
    u64 done = 0;
    if (!(z & 1) && !(y & 1)) {
      const u64 pw = ev_prev[((long long)(z >> 1) * rp + (y >> 1)) * w64p + (w >> 1)];
      done = spread32(pw >> (32 * (w & 1)));
    }
    const int nbits = min(64, r - 64 * w);
    const u64 in_range = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
    sel = acc & ~done & in_range;
    ev[((long long)z * r + y) * w64 + w] = done | sel;
  }
  // wave-level exclusive prefix of popcounts, one atomic per wave
  const int lane = threadIdx.x & 63;
  const int cnt = __popcll(sel);
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63);
  int base = 0;
  if (lane == 63 && total > 0) base = atomicAdd(count, total);
  base = __shfl(base, 63);
  int pos = base + incl - cnt;
  const uint32_t yz = ((uint32_t)y << 10) | ((uint32_t)z << 20);
  while (sel) {
    const int b = __ffsll((long long)sel) - 1;
    sel &= sel - 1;
    packed[pos++] = (uint32_t)(64 * w + b) | yz;
  }
}

// 9^3, 7^3, 3^3 boxes at levels 1, 2, 3+ (the upstream engine's "faster" schedule)

static void launch_select(int box, hipStream_t st, const FrameBufs &fb, int n_frames, int rp, int w64p, int r, int w64) {
  const bool ym = octree_y_major();
  const long long items = (long long)(ym ? (r + kYSlab - 1) / kYSlab * kYSlab : r) * r * w64;
  const unsigned blocks = (unsigned)((items + 255) / 256);
#define MP_SELECT(D)                                                                                           \
  if (ym)                                                                                                      \
    hipLaunchKernelGGL((select_compact_kernel<D, true>), dim3(blocks, 1, n_frames), dim3(256), 0, st, fb, rp, \
                       w64p, r, w64);                                                                          \
  else                                                                                                         \
    hipLaunchKernelGGL((select_compact_kernel<D, false>), dim3(blocks, 1, n_frames), dim3(256), 0, st, fb, rp, \
                       w64p, r, w64)
  if (box == 9) {
    MP_SELECT(4);
  } else if (box == 7) {
    MP_SELECT(3);
  } else if (box == 3) {
    MP_SELECT(1);
  } else {  // box <= 1: the flagged nodes themselves, no dilation
    MP_SELECT(0);
  }
#undef MP_SELECT
}

// ---- conflict re-examination (the upstream engine's faster=False mode) ------------------------
// A node just evaluated is in CONFLICT when its exact value and the value interpolated from the
// coarser level lie on different sides of the threshold: the surface passes where the coarse
// level did not expect it.  Every not-yet-evaluated node of its 3x3x3 neighbourhood is then queued
// (claimed with an atomic test-and-set on the evaluated bitset) for the next round.
// vol still holds the INTERPOLATED values of the nodes in `packed` (scatter comes afterwards).
__global__ void conflict_expand_kernel(const uint32_t *__restrict__ packed,
                                       const int32_t *__restrict__ count, long long cap, int r,
                                       int w64, const float *__restrict__ values,
                                       const float *__restrict__ vol, float balance,
                                       u64 *__restrict__ ev, uint32_t *__restrict__ out,
                                       int32_t *__restrict__ out_count) {
  const long long n = min((long long)*count, cap);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const uint32_t c = packed[i];
    const int x = c & 1023u, y = (c >> 10) & 1023u, z = c >> 20;
    const float interp = vol[((long long)z * r + y) * r + x];
    if (!((interp - balance) * (values[i] - balance) < 0.0f)) continue;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int xx = x + dx, yy = y + dy, zz = z + dz;
          if (xx < 0 || yy < 0 || zz < 0 || xx >= r || yy >= r || zz >= r) continue;
          u64 *word = ev + ((long long)zz * r + yy) * w64 + (xx >> 6);
          const u64 bit = 1ull << (xx & 63);
          if (atomicOr(word, bit) & bit) continue;  // already evaluated / claimed
          out[atomicAdd(out_count, 1)] = (uint32_t)xx | ((uint32_t)yy << 10) | ((uint32_t)zz << 20);
        }
  }
}

int launch_octree_conflicts(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, long long cap,
                            int r, const float *values, const float *vol, float balance, u64 *ev,
                            uint32_t *out, int32_t *out_count, hipStream_t st) {
  MP_HIP(ctx, hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
  if (cap == 0) return MP_OK;
  long long blocks = (cap + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(conflict_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, count,
                     cap, r, words64(r), values, vol, balance, ev, out, out_count);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- level-at-a-time entry points (generic query_func) -------------------------------------------
__global__ void lattice_points_kernel(const uint32_t *__restrict__ packed,
                                      const int32_t *__restrict__ count, long long cap, int stride,
                                      float res_final, float half_step, float b0, float b1, float b2,
                                      float l0, float l1, float l2, float *__restrict__ pts) {
  const long long n = min((long long)*count, cap);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const uint32_t c = packed[i];
    const float cx = (float)((int)(c & 1023u) * stride), cy = (float)((int)((c >> 10) & 1023u) * stride),
                cz = (float)((int)(c >> 20) * stride);
    pts[3 * i + 0] = (cx / res_final + half_step) * l0 + b0;
    pts[3 * i + 1] = (cy / res_final + half_step) * l1 + b1;
    pts[3 * i + 2] = (cz / res_final + half_step) * l2 + b2;
  }
}

__global__ void scatter_nodes_kernel(const uint32_t *__restrict__ packed,
                                     const int32_t *__restrict__ count, long long cap, int r,
                                     const float *__restrict__ values, float *__restrict__ vol) {
  const long long n = min((long long)*count, cap);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const uint32_t c = packed[i];
    vol[((long long)(c >> 20) * r + ((c >> 10) & 1023u)) * r + (c & 1023u)] = values[i];
  }
}

int launch_octree_select(mp_ctx *ctx, const float *prev, int rp, float *cur, int r,
                         const u64 *ev_prev, u64 *ev_cur, u64 *bnd, int box, float balance,
                         uint32_t *packed, int32_t *count, hipStream_t st) {
  const int w64 = words64(r);
  FrameBufs fb;
  std::memset(&fb, 0, sizeof(fb));
  fb.prev[0] = prev;
  fb.cur[0] = cur;
  fb.bnd[0] = bnd;
  fb.ev_prev[0] = ev_prev;
  fb.ev[0] = ev_cur;
  fb.packed[0] = packed;
  fb.count[0] = count;
  if (!prev) {
    const int total = r * r * r;
    hipLaunchKernelGGL(iota_nodes_kernel, dim3((total + 255) / 256), dim3(256), 0, st, r, fb, w64, (int)octree_y_major());
  } else {
    MP_HIP(ctx, hipMemsetAsync(count, 0, sizeof(int32_t), st));
    // box 1: the undilated "upsampled mask == 0.5" rule; box 0: upsample only (nothing selected)
    hipLaunchKernelGGL(upsample_classify_kernel,
                       dim3((unsigned)((rp * ((rp + 63) / 64) + 3) / 4), (unsigned)rp), dim3(256), 0,
                       st, fb, rp, r, balance, w64,
                       box == 1 ? MP_FINAL_UPSTREAM : box == 0 ? MP_FINAL_INTERPOLATE : MP_FINAL_DILATE3);
    launch_select(box, st, fb, 1, rp, words64(rp), r, w64);
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_lattice_points(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, long long cap,
                          int stride, int res_final, const float *bmin, const float *bmax,
                          float *pts, hipStream_t st) {
  if (cap == 0) return MP_OK;
  long long blocks = (cap + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const float rf = (float)res_final;
  hipLaunchKernelGGL(lattice_points_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, count,
                     cap, stride, rf, (1.0f / rf) / 2.0f, bmin[0], bmin[1], bmin[2],
                     bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2], pts);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_scatter_nodes(mp_ctx *ctx, const uint32_t *packed, const int32_t *count, long long cap,
                         int r, const float *values, float *vol, hipStream_t st) {
  if (cap == 0) return MP_OK;
  long long blocks = (cap + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scatter_nodes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, count,
                     cap, r, values, vol);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---- driver ------------------------------------------------------------------------------------
int launch_recon(mp_ctx *ctx, void *scratch, const Mlp &m, int n_frames,
                 const float *const *feat_hwc, int h, int w, const float *const *calib,
                 float z_scale, const float *bmin, const float *bmax, const int *res, int n_levels,
                 float balance, int final_level, float *const *volume, int32_t *const *status,
                 const mp_recon_early *early, hipStream_t st) {
  // carve the scratch arena: one private set of level buffers per frame
  const size_t per_frame = recon_scratch_bytes(res, n_levels);
  LevelBufs lv[kMaxFrames][8];
  uint32_t *packed[kMaxFrames];
  for (int f = 0; f < n_frames; ++f) {
    unsigned char *p = static_cast<unsigned char *>(scratch) + f * per_frame;
    for (int l = 0; l < n_levels; ++l) {
      const size_t r = res[l];
      if (l < n_levels - 1) {
        lv[f][l].occ = reinterpret_cast<float *>(p);
        p += align256(r * r * r * sizeof(float));
      } else {
        lv[f][l].occ = volume[f];
      }
      const size_t wb = align256(r * r * words64(res[l]) * sizeof(u64));
      lv[f][l].bnd = reinterpret_cast<u64 *>(p);
      p += wb;
      lv[f][l].ev = reinterpret_cast<u64 *>(p);
      p += wb;
    }
    packed[f] = reinterpret_cast<uint32_t *>(p);
  }

  const int rf = res[n_levels - 1];
  QuerySet set;
  std::memset(&set, 0, sizeof(set));
  set.n = n_frames;
  for (int f = 0; f < n_frames; ++f) {
    QueryItem &q = set.it[f];
    q.feat = feat_hwc[f];
    q.calib = calib[f];
    q.src.packed = packed[f];
    q.src.res_final = (float)rf;
    q.src.half_step = (1.0f / (float)rf) / 2.0f;
    for (int i = 0; i < 3; ++i) {
      q.src.bmin[i] = bmin[i];
      q.src.blen[i] = bmax[i] - bmin[i];
    }
    MP_HIP(ctx, hipMemsetAsync(status[f], 0, sizeof(int32_t) * (1 + n_levels), st));
  }

  FrameBufs fb;
  std::memset(&fb, 0, sizeof(fb));
  // frames per housekeeping launch (MONOPORT_OCTREE_CHUNK, measurement switch; results do not depend on it)
  static const int chunk_env = [] {
    const char *e = getenv("MONOPORT_OCTREE_CHUNK");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : kHouseChunk;
  }();
  const int chunk = chunk_env < n_frames ? chunk_env : n_frames;
  auto sub = [&](int f0, int nf) {  // the descriptors of frames f0 .. f0 + nf - 1 as frames 0 .. nf - 1
    FrameBufs c;
    std::memset(&c, 0, sizeof(c));
    for (int f = 0; f < nf; ++f) {
      c.prev[f] = fb.prev[f0 + f];
      c.cur[f] = fb.cur[f0 + f];
      c.bnd[f] = fb.bnd[f0 + f];
      c.ev_prev[f] = fb.ev_prev[f0 + f];
      c.ev[f] = fb.ev[f0 + f];
      c.packed[f] = fb.packed[f0 + f];
      c.count[f] = fb.count[f0 + f];
      c.flag[f] = fb.flag[f0 + f];
    }
    return c;
  };
  // level 0: every node of every frame, one query launch for the whole set
  {
    const int r = res[0], total = r * r * r;
    for (int f = 0; f < n_frames; ++f) {
      fb.cur[f] = lv[f][0].occ;
      fb.ev[f] = lv[f][0].ev;
      fb.packed[f] = packed[f];
      fb.count[f] = status[f] + 1;
      fb.flag[f] = status[f];
      QueryItem &q = set.it[f];
      q.out = lv[f][0].occ;
      q.src.stride = (rf - 1) / (r - 1);
      q.src.level_res = r;
      q.src.n_dev = nullptr;
      q.src.n = total;
    }
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
      const int nf = min(chunk, n_frames - f0);
      hipLaunchKernelGGL(iota_nodes_kernel, dim3((total + 255) / 256, 1, nf), dim3(256), 0, st, r, sub(f0, nf), words64(r),
                         (int)octree_y_major());
    }
    int rc = launch_query_set(ctx, m, set, h, w, z_scale, (long long)total * n_frames, false, st);
    if (rc != MP_OK) return rc;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
      const int nf = min(chunk, n_frames - f0);
      hipLaunchKernelGGL(any_above_kernel, dim3(min((total + 255) / 256, 256), 1, nf), dim3(256), 0, st, sub(f0, nf), total,
                         balance);
    }
    if (early) {  // what the caller needs to decide "None or a volume" and "fused or not": ready after ~0.1 ms of GPU time
      EarlyBufs eb;
      std::memset(&eb, 0, sizeof(eb));
      for (int f = 0; f < n_frames; ++f) {
        eb.occ[f] = lv[f][0].occ;
        eb.expect[f] = early->expect_level0 ? early->expect_level0[f] : nullptr;
        eb.flag[f] = status[f];
      }
      MP_HIP(ctx, hipMemsetAsync(early->flags_dev, 0, sizeof(int32_t) * 2 * n_frames, st));
      hipLaunchKernelGGL(early_flags_kernel, dim3(min((total + 255) / 256, 32), 1, n_frames), dim3(256), 0, st, eb, total,
                         early->flags_dev);
      MP_HIP(ctx, hipMemcpyAsync(early->flags_host, early->flags_dev, sizeof(int32_t) * 2 * n_frames, hipMemcpyDeviceToHost, st));
      if (early->event) MP_HIP(ctx, hipEventRecord((hipEvent_t)early->event, st));
    }
  }
  for (int l = 1; l < n_levels; ++l) {
    const int r = res[l], rp = res[l - 1], w64 = words64(r);
    // the last level's selection rule (mp_recon_batch_ex): the lossless schedule dilates the boundary by 3^3 there
    // too; "upstream" evaluates only the nodes whose upsampled mask is exactly 0.5; "interpolate" none
    const int rule = l == n_levels - 1 ? final_level : MP_FINAL_DILATE3;
    for (int f = 0; f < n_frames; ++f) {
      fb.prev[f] = lv[f][l - 1].occ;
      fb.cur[f] = lv[f][l].occ;
      fb.bnd[f] = lv[f][l].bnd;
      fb.ev_prev[f] = lv[f][l - 1].ev;
      fb.ev[f] = lv[f][l].ev;
      fb.count[f] = status[f] + 1 + l;
      QueryItem &q = set.it[f];
      q.out = lv[f][l].occ;
      q.src.stride = (rf - 1) / (r - 1);
      q.src.level_res = r;
      q.src.n_dev = status[f] + 1 + l;
      q.src.n = 0;
    }
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
      const int nf = min(chunk, n_frames - f0);
      const FrameBufs c = sub(f0, nf);
      hipLaunchKernelGGL(upsample_classify_kernel,
                         dim3((unsigned)((rp * ((rp + 63) / 64) + 3) / 4), (unsigned)rp, (unsigned)nf), dim3(256), 0, st,
                         c, rp, r, balance, w64, rule);
      if (rule != MP_FINAL_INTERPOLATE)
        launch_select(rule == MP_FINAL_UPSTREAM ? 1 : octree_box_of_level(l), st, c, nf, rp, words64(rp), r, w64);
    }
    if (rule == MP_FINAL_INTERPOLATE) continue;  // status[1 + l] stays 0, no query
    int rc = launch_query_set(ctx, m, set, h, w, z_scale, (long long)r * r * r * n_frames, true, st);
    if (rc != MP_OK) return rc;
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
