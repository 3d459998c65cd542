// Visible-surface extraction and colour painting on the GPU.
//   forward_vertices  RTL/recon.py:27-89   (first hit along the view axis, sub-voxel depth, normal)
//   colorization      RTL/main.py:212-249  (vertex -> world points for netC.query; canvas scatter)
// The reference materialises ten full-volume temporaries (flip, permute, ramp, max, nonzero ...);
// here one pass walks each (x, y) column until its first occupied voxel and a single workgroup
// turns the R*R hit map into the reference's x-major row order with a block scan.
#include <cstring>

#include "mp_internal.h"

// Reference parity is op-order parity: keep every a*b+c exactly as written (the HIP headers
// define __fmul_rn & co. as plain operators, which hipcc would otherwise contract into FMAs).
// Fused multiply-adds are requested explicitly (fmaf / MFMA) where they are wanted.
#pragma clang fp contract(off)

namespace mp {

// s[x, y, z'] of RTL/recon.py:51-53 expressed on the original [z, y, x] volume, per direction
// (front :39-40, left :41-42, back :43-45, right :46-49).
__device__ __forceinline__ float view_sample(const float *__restrict__ v, int r, int dir, int x,
                                             int y, int zp) {
  long long zi, xi;
  if (dir == MP_DIR_FRONT) {
    zi = r - 1 - zp;
    xi = x;
  } else if (dir == MP_DIR_BACK) {
    zi = zp;
    xi = x;
  } else if (dir == MP_DIR_LEFT) {
    zi = x;
    xi = r - 1 - zp;
  } else {
    zi = r - 1 - x;
    xi = r - 1 - zp;
  }
  return v[(zi * r + y) * r + xi];
}

// hit[x * r + y] = first z' with s > 0.5 (recon.py:56-60), or kNoHit.  Each column is cut into
// segments of kSeg voxels scanned by different threads (one thread per column leaves the chip
// mostly idle: 66 k threads walking 257 dependent loads each); the segments meet in an atomicMin.
constexpr int kNoHit = 0x7f7f7f7f;  // what hipMemsetAsync(0x7f) leaves behind
constexpr int kSeg = 32;

// The kernels below serve up to kMaxFrames volumes per launch (blockIdx.y = frame: mp_forward_vertices_batch; a
// single 257^3 volume gives first_hit 2.3 k workgroups and the other two 65 each -- launch-bound).
struct VertFrames {
  const float *vol[kMaxFrames];
  int64_t *x[kMaxFrames], *y[kMaxFrames];
  float *z[kMaxFrames], *n[kMaxFrames];
  int32_t *count[kMaxFrames];
};

__global__ __launch_bounds__(256) void first_hit_kernel(VertFrames fr, int r, int dir, int n_seg,
                                                        int32_t *__restrict__ hit_all, long long hit_stride) {
  const float *__restrict__ v = fr.vol[blockIdx.y];
  int32_t *__restrict__ hit = hit_all + blockIdx.y * hit_stride;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long cols = (long long)r * r;
  if (t >= cols * n_seg) return;
  const int seg = (int)(t / cols);
  const int col = (int)(t % cols);
  // lanes run along the memory-contiguous axis for front/back, along y for left/right
  int x, y;
  if (dir == MP_DIR_FRONT || dir == MP_DIR_BACK) {
    x = col % r;
    y = col / r;
  } else {
    y = col % r;
    x = col / r;
  }
  // all kSeg samples of the segment are requested before the first is looked at: a loop that leaves at the first hit
  // has ONE load in flight per lane (32 dependent round trips per segment: 60 us per 257^3 volume, 0.85 TB/s);
  // the volume is read once either way (the segments behind a hit scan their part regardless)
  float vals[kSeg];
#pragma unroll
  for (int k = 0; k < kSeg; ++k) {
    const int zp = seg * kSeg + k;
    vals[k] = zp < r ? view_sample(v, r, dir, x, y, zp) : 0.0f;
  }
  int first = -1;
#pragma unroll
  for (int k = kSeg - 1; k >= 0; --k)
    if (vals[k] > 0.5f) first = seg * kSeg + k;
  if (first >= 0) atomicMin(&hit[x * r + y], first);
}

// Row order = x-major order of keep.nonzero() (recon.py:62): per-block hit counts, then every
// block sums the counts of the blocks before it and emits its rows (recon.py:63-87).
constexpr int kEmitBlock = 1024;

__global__ __launch_bounds__(kEmitBlock) void count_hits_kernel(int32_t *__restrict__ hit_all,
                                                                int total, long long hit_stride) {
  const int32_t *__restrict__ hit = hit_all + blockIdx.y * hit_stride;
  int32_t *__restrict__ blk = hit_all + blockIdx.y * hit_stride + total;
  __shared__ int wave_tot[kEmitBlock / 64];
  const int idx = blockIdx.x * kEmitBlock + threadIdx.x;
  const int flag = idx < total && hit[idx] != kNoHit;
  const unsigned long long m = __ballot(flag);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int k = 0; k < kEmitBlock / 64; ++k) s += wave_tot[k];
    blk[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kEmitBlock) void emit_vertices_kernel(VertFrames fr, int r, int dir,
                                                                   const int32_t *__restrict__ hit_all,
                                                                   long long hit_stride) {
  const float *__restrict__ v = fr.vol[blockIdx.y];
  const int32_t *__restrict__ hit = hit_all + blockIdx.y * hit_stride;
  const int32_t *__restrict__ blk = hit + r * r;
  int64_t *__restrict__ xo = fr.x[blockIdx.y];
  int64_t *__restrict__ yo = fr.y[blockIdx.y];
  float *__restrict__ zo = fr.z[blockIdx.y];
  float *__restrict__ no = fr.n[blockIdx.y];
  int32_t *__restrict__ count = fr.count[blockIdx.y];
  __shared__ int wave_tot[kEmitBlock / 64];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int total = r * r;
  const int idx = blockIdx.x * kEmitBlock + tid;
  const int z1 = idx < total ? hit[idx] : kNoHit;
  const int flag = z1 != kNoHit;
  const unsigned long long m = __ballot(flag);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wv] = __popcll(m);
  if (tid == 0) {
    int s = 0;
    for (int k = 0; k < (int)blockIdx.x; ++k) s += blk[k];
    base_s = s;
    if (blockIdx.x == gridDim.x - 1) *count = s + blk[blockIdx.x];
  }
  __syncthreads();
  if (!flag) return;
  int off = base_s;
  for (int k = 0; k < wv; ++k) off += wave_tot[k];
  const int row = off + before;
  const int x = idx / r, y = idx % r;
  const int z2 = min(max(z1 - 2, 0), r), y2 = min(max(y - 2, 0), r), x2 = min(max(x - 2, 0), r);
  const float v1 = view_sample(v, r, dir, x, y, z1);
  const float v2 = view_sample(v, r, dir, x, y, z2);
  const float v3 = view_sample(v, r, dir, x, y2, z1);
  const float v4 = view_sample(v, r, dir, x2, y, z1);
  // recon.py:77: p2z * (0.5 - v1) / (v2 - v1) + p1z * (v2 - 0.5) / (v2 - v1), left to right
  const float den = __fsub_rn(v2, v1);
  const float ta = __fdiv_rn(__fmul_rn((float)z2, __fsub_rn(0.5f, v1)), den);
  const float tb = __fdiv_rn(__fmul_rn((float)z1, __fsub_rn(v2, 0.5f)), den);
  float zz = __fadd_rn(ta, tb);
  zz = zz < 0.0f ? 0.0f : (zz > (float)r ? (float)r : zz);  // clamp keeps NaN (hit at z'=0)
  const float nx = __fsub_rn(v4, v1), ny = __fsub_rn(v3, v1), nz = den;
  const float len =
      __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
  xo[row] = x;
  yo[row] = y;
  zo[row] = zz;
  no[3 * row + 0] = __fdiv_rn(nx, len);
  no[3 * row + 1] = __fdiv_rn(ny, len);
  no[3 * row + 2] = __fdiv_rn(nz, len);
}

size_t forward_vertices_scratch_bytes(int r) {  // per frame: hit [r * r] + per-block counts
  const size_t total = (size_t)r * r;
  return ((total + (total + kEmitBlock - 1) / kEmitBlock + 2) * sizeof(int32_t) + 255) & ~size_t(255);
}

int launch_forward_vertices_batch(mp_ctx *ctx, void *scratch, int n_frames, const float *const *vol, int r, int dir,
                                  int64_t *const *x, int64_t *const *y, float *const *z, float *const *norm,
                                  int32_t *const *count, hipStream_t st) {
  VertFrames fr;
  std::memset(&fr, 0, sizeof(fr));
  for (int f = 0; f < n_frames; ++f) {
    fr.vol[f] = vol[f];
    fr.x[f] = x[f];
    fr.y[f] = y[f];
    fr.z[f] = z[f];
    fr.n[f] = norm[f];
    fr.count[f] = count[f];
  }
  int32_t *hit = static_cast<int32_t *>(scratch);
  const long long stride = (long long)(forward_vertices_scratch_bytes(r) / sizeof(int32_t));
  const int total = r * r;
  const int n_blk = (total + kEmitBlock - 1) / kEmitBlock;
  const int n_seg = (r + kSeg - 1) / kSeg;
  MP_HIP(ctx, hipMemsetAsync(hit, 0x7f, sizeof(int32_t) * (size_t)stride * n_frames, st));
  const long long threads = (long long)total * n_seg;
  hipLaunchKernelGGL(first_hit_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, fr,
                     r, dir, n_seg, hit, stride);
  hipLaunchKernelGGL(count_hits_kernel, dim3(n_blk, n_frames), dim3(kEmitBlock), 0, st, hit, total, stride);
  hipLaunchKernelGGL(emit_vertices_kernel, dim3(n_blk, n_frames), dim3(kEmitBlock), 0, st, fr, r, dir, hit, stride);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_forward_vertices(mp_ctx *ctx, void *scratch, const float *vol, int r, int dir, int64_t *x,
                            int64_t *y, float *z, float *norm, int32_t *count, hipStream_t st) {
  return launch_forward_vertices_batch(ctx, scratch, 1, &vol, r, dir, &x, &y, &z, &norm, &count, st);
}

// verts = (X, Y, res - Z) (main.py:231-233) through orthogonal(., mat_color) (main.py:237).
struct Mat34 {
  float m[12];
};

__global__ void vertex_points_kernel(const int64_t *__restrict__ x, const int64_t *__restrict__ y,
                                     const float *__restrict__ z, const int32_t *__restrict__ count,
                                     long long cap, int res, Mat34 mat, float *__restrict__ pts) {
  const long long n = min((long long)*count, cap);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float vx = (float)x[i], vy = (float)y[i], vz = __fsub_rn((float)res, z[i]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      // orthogonal() = torch.baddbmm: t + fma(r2, z, fma(r1, y, r0 * x)) (query_common.h: project)
      pts[k * cap + i] = __fadd_rn(
          mat.m[4 * k + 3],
          fmaf(mat.m[4 * k + 2], vz, fmaf(mat.m[4 * k + 1], vy, __fmul_rn(mat.m[4 * k], vx))));
  }
}

int launch_vertex_points(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *z,
                         const int32_t *count, long long cap, int res, const float *mat16,
                         float *pts, hipStream_t st) {
  if (cap == 0) return MP_OK;
  Mat34 m;
  for (int i = 0; i < 12; ++i) m.m[i] = mat16[i];
  long long blocks = (cap + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(vertex_points_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, z, count,
                     cap, res, m, pts);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

struct PaintFrames {
  const int64_t *x[kMaxFrames], *y[kMaxFrames];
  const float *vals[kMaxFrames];
  const int32_t *count[kMaxFrames];
  float *image[kMaxFrames];
};

__global__ void fill_kernel(PaintFrames fr, long long n, float v) {
  float *__restrict__ p = fr.image[blockIdx.y];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void paint_kernel(PaintFrames fr, int ch_major, long long cap, int res, float scale,
                             float bias, float lo, float hi) {
  const int64_t *__restrict__ x = fr.x[blockIdx.y];
  const int64_t *__restrict__ y = fr.y[blockIdx.y];
  const float *__restrict__ vals = fr.vals[blockIdx.y];
  float *__restrict__ image = fr.image[blockIdx.y];
  const long long n = min((long long)*fr.count[blockIdx.y], cap);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long px = x[i], py = y[i];
    if (px < 0 || px >= res || py < 0 || py >= res) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = ch_major ? vals[c * cap + i] : vals[3 * i + c];
      float o = __fadd_rn(__fmul_rn(v, scale), bias);
      o = o < lo ? lo : (o > hi ? hi : o);
      image[(px * res + py) * 3 + c] = o;
    }
  }
}

int launch_paint_batch(mp_ctx *ctx, int n_frames, const int64_t *const *x, const int64_t *const *y,
                       const float *const *vals, int ch_major, const int32_t *const *count, long long cap, int res,
                       float scale, float bias, float lo, float hi, float *const *image, hipStream_t st) {
  PaintFrames fr;
  std::memset(&fr, 0, sizeof(fr));
  for (int f = 0; f < n_frames; ++f) {
    fr.x[f] = x[f];
    fr.y[f] = y[f];
    fr.vals[f] = vals[f];
    fr.count[f] = count[f];
    fr.image[f] = image[f];
  }
  const long long n_img = (long long)res * res * 3;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n_img + 255) / 256 > 1024 ? 1024 : (n_img + 255) / 256), n_frames),
                     dim3(256), 0, st, fr, n_img, 1.0f);  // canvas of ones, main.py:201-203
  if (cap > 0) {
    long long blocks = (cap + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(paint_kernel, dim3((unsigned)blocks, n_frames), dim3(256), 0, st, fr, ch_major, cap, res, scale,
                       bias, lo, hi);
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_paint(mp_ctx *ctx, const int64_t *x, const int64_t *y, const float *vals, int ch_major,
                 const int32_t *count, long long cap, int res, float scale, float bias, float lo,
                 float hi, float *image, hipStream_t st) {
  return launch_paint_batch(ctx, 1, &x, &y, &vals, ch_major, &count, cap, res, scale, bias, lo, hi, &image, st);
}

// RTL/main.py:259-281: *255, torch.rot90(k=1, dims [0,1]) (out[i][j] = in[j][res-1-i]), nearest
// resize to `size` (src = floor(dst * res / size), the legacy 'nearest' rule) and the white-
// background mask, in one pass; the caller does the single D2H copy.
__global__ void visualize_kernel(const float *__restrict__ image, int res, int size,
                                 float *__restrict__ out, uint8_t *__restrict__ mask) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= size * size) return;
  const int i = t / size, j = t % size;
  const float scale = (float)res / (float)size;
  const int si = min((int)floorf((float)i * scale), res - 1);
  const int sj = min((int)floorf((float)j * scale), res - 1);
  const float *src = image + ((long long)sj * res + (res - 1 - si)) * 3;
  bool white = true;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = src[c] * 255.0f;
    out[3 * (long long)t + c] = v;
    white = white && v == 255.0f;
  }
  mask[t] = white ? 0 : 1;
}

int launch_visualize(mp_ctx *ctx, const float *image, int res, int size, float *out, uint8_t *mask,
                     hipStream_t st) {
  hipLaunchKernelGGL(visualize_kernel, dim3((size * size + 255) / 256), dim3(256), 0, st, image, res,
                     size, out, mask);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
