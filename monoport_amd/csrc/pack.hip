// Layout kernels: SurfaceClassifier weights -> MFMA fragment order, NCHW feature map -> channels
// last.  Both run once per model / once per frame; neither is on the per-point critical path.
#include "mp_internal.h"

namespace mp {

// ---------------------------------------------------------------------------------------------
// Weight packing.
//
// v_mfma_f32_32x32x2_f32 takes ONE f32 of A per lane: lane l supplies A[row = l & 31][k = l >> 5].
// The query kernel walks K in groups of 8: lane (j, h) reads 4 consecutive B values
// k = 8g + 4h + {0..3} with one ds_read_b128 and uses register i as the B operand of k-step
// (g, i).  The matching A fragment for that k-step is W[32 rb + j][8g + 4h + i], so we store
//     A[rb][g][lane][i] = W[32 rb + (lane & 31)][seg0 + 8 g + 4 (lane >> 5) + i]
// and a wave fetches the 4 k-steps of a group with a single coalesced 16-byte load per lane.
// The z column (K = C + 1 is odd) gets its own one-step fragment: lanes 0-31 carry the weight,
// lanes 32-63 carry 0 (the matching B operand is z_feat in lanes 0-31).
// ---------------------------------------------------------------------------------------------
__global__ void pack_segment_kernel(const float *__restrict__ w, int ld, int col0, int n_groups,
                                    int n_rb, float *__restrict__ dst) {
  const long long total = (long long)n_rb * n_groups * 256;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int i = t & 3;
    const int lane = (t >> 2) & 63;
    const long long q = t >> 8;
    const int g = q % n_groups;
    const int rb = q / n_groups;
    const int row = 32 * rb + (lane & 31);
    const int col = col0 + 8 * g + 4 * (lane >> 5) + i;
    dst[t] = w[(long long)row * ld + col];
  }
}

__global__ void pack_zcol_kernel(const float *__restrict__ w, int ld, int col, int n_rb,
                                 float *__restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_rb * 64) return;
  const int lane = t & 63, rb = t >> 6;
  dst[t] = lane < 32 ? w[(long long)(32 * rb + lane) * ld + col] : 0.0f;
}

__global__ void copy_kernel(const float *__restrict__ src, float *__restrict__ dst, long long n) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
       t += (long long)gridDim.x * blockDim.x)
    dst[t] = src[t];
}

__global__ void copy_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, int rows,
                                 int cols, int dst_stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cols) return;
  dst[(t / cols) * dst_stride + (t % cols)] = src[t];
}

int launch_pack_layer(mp_ctx *ctx, Mlp &m, int layer, const float *w, const float *b,
                      hipStream_t st) {
  const int c = m.c;
  if (layer < 4) {
    const int n_out = kHidden[layer];
    const int k_h = layer == 0 ? 0 : kHidden[layer - 1];
    const int ld = k_h + c + 1;
    const int n_rb = n_out / 32;
    if (k_h > 0) {
      const long long tot = (long long)n_rb * (k_h / 8) * 256;
      const int blocks = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
      hipLaunchKernelGGL(pack_segment_kernel, dim3(blocks), dim3(256), 0, st, w, ld, 0, k_h / 8,
                         n_rb, m.buf + m.off_ah[layer]);
    }
    {
      const long long tot = (long long)n_rb * (c / 8) * 256;
      const int blocks = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
      hipLaunchKernelGGL(pack_segment_kernel, dim3(blocks), dim3(256), 0, st, w, ld, k_h, c / 8,
                         n_rb, m.buf + m.off_ax[layer]);
    }
    hipLaunchKernelGGL(pack_zcol_kernel, dim3((n_rb * 64 + 255) / 256), dim3(256), 0, st, w, ld,
                       k_h + c, n_rb, m.buf + m.off_az[layer]);
    hipLaunchKernelGGL(copy_kernel, dim3((n_out + 255) / 256), dim3(256), 0, st, b,
                       m.buf + m.off_bias[layer], (long long)n_out);
  } else {
    // last layer stays row-major; rows padded to a multiple of 4 floats for 16-byte loads
    const int k4 = kHidden[3] + c + 1, k4s = (k4 + 3) & ~3;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((m.cout * k4 + 255) / 256), dim3(256), 0, st, w,
                       m.buf + m.off_w4, m.cout, k4, k4s);
    hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(256), 0, st, b, m.buf + m.off_bias[4],
                       (long long)m.cout);
  }
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// f16x3 packing (query16.hip): A16[rb][g][part][lane][e], part 0 = hi, 1 = lo,
//   value = W[32 rb + (lane & 31)][col0 + 16 g + 8 (lane >> 5) + e] * S  split into two halves.
// ---------------------------------------------------------------------------------------------
// perm_t > 0 (hidden segments of layers 2 and 3): K is re-ordered so that every 64-deep chunk
// takes 16 rows from each of the 4 waves of query16.hip (wave g owns tiles perm_t*g .. +perm_t-1
// of the previous layer): k' = 64 c + 16 g + e  <->  source row 32 (perm_t g + (c >> 1)) + 16 (c & 1) + e.
__global__ void pack_segment16_kernel(const float *__restrict__ w, int ld, int col0, int n_groups,
                                      int n_rb, float scale, int perm_t,
                                      _Float16 *__restrict__ dst) {
  const long long total = (long long)n_rb * n_groups * 512;  // (lane, e) pairs
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int e = t & 7;
    const int lane = (t >> 3) & 63;
    const long long q = t >> 9;
    const int g = q % n_groups;
    const int rb = q / n_groups;
    int k = 16 * g + 8 * (lane >> 5) + e;
    if (perm_t > 0) {
      const int c = k >> 6, gw = (k >> 4) & 3, ee = k & 15;
      k = 32 * (perm_t * gw + (c >> 1)) + 16 * (c & 1) + ee;
    }
    const float v = w[(long long)(32 * rb + (lane & 31)) * ld + col0 + k] * scale;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    _Float16 *blk = dst + q * 1024;  // 2 parts x 64 lanes x 8 halves
    blk[lane * 8 + e] = hi;
    blk[512 + lane * 8 + e] = lo;
  }
}

__global__ void pack_zcol16_kernel(const float *__restrict__ w, int ld, int col, int n_rb,
                                   float scale, _Float16 *__restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_rb * 512) return;
  const int e = t & 7, lane = (t >> 3) & 63, rb = t >> 9;
  float v = 0.0f;
  if (lane < 32 && e == 0) v = w[(long long)(32 * rb + lane) * ld + col] * scale;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  dst[(long long)rb * 1024 + lane * 8 + e] = hi;
  dst[(long long)rb * 1024 + 512 + lane * 8 + e] = lo;
}

__global__ void absmax_kernel(const float *__restrict__ src, long long n, unsigned int *out) {
  float m = 0.0f;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
       t += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(src[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order as uints
}

int launch_absmax(mp_ctx *ctx, const float *src, long long n, unsigned int *out_bits,
                  hipStream_t st) {
  MP_HIP(ctx, hipMemsetAsync(out_bits, 0, sizeof(unsigned int), st));
  hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, src, n, out_bits);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// like launch_absmax without zeroing the word first (maximum over several tensors)
int launch_absmax_accumulate(mp_ctx *ctx, const float *src, long long n, unsigned int *out_bits,
                             hipStream_t st) {
  hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, src, n, out_bits);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_copy(mp_ctx *ctx, const float *src, float *dst, long long n, hipStream_t st) {
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)),
                     dim3(256), 0, st, src, dst, n);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

int launch_pack_layer16(mp_ctx *ctx, Mlp &m, int layer, const float *w, hipStream_t st) {
  const int c = m.c;
  const int n_out = kHidden[layer];
  const int k_h = layer == 0 ? 0 : kHidden[layer - 1];
  const int ld = k_h + c + 1;
  const int n_rb = n_out / 32;
  const float s = m.scale16[layer];
  _Float16 *base = static_cast<_Float16 *>(m.buf16);
  if (k_h > 0)
    hipLaunchKernelGGL(pack_segment16_kernel, dim3(2048), dim3(256), 0, st, w, ld, 0, k_h / 16,
                       n_rb, s, layer == 2 ? 4 : (layer == 3 ? 2 : 0), base + m.off16_ah[layer] * 8);
  hipLaunchKernelGGL(pack_segment16_kernel, dim3(2048), dim3(256), 0, st, w, ld, k_h, c / 16, n_rb,
                     s, 0, base + m.off16_ax[layer] * 8);
  hipLaunchKernelGGL(pack_zcol16_kernel, dim3((n_rb * 512 + 255) / 256), dim3(256), 0, st, w, ld,
                     k_h + c, n_rb, s, base + m.off16_az[layer] * 8);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW -> NHWC through a padded 32x32 LDS tile: coalesced 128-byte rows on both sides.
// With channels last, one bilinear tap of the query kernel is C contiguous floats (1 KB for
// netG), i.e. one fully coalesced 16 B/lane wave load instead of C loads 64 KB apart.
// ---------------------------------------------------------------------------------------------
__global__ void chw_to_hwc_kernel(const float *__restrict__ src, int c_src, int hw,
                                  float *__restrict__ dst, int c_dst, int c_off) {
  __shared__ float tile[32][33];
  const int p0 = blockIdx.x * 32;  // pixel tile
  const int c0 = blockIdx.y * 32;  // channel tile
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int ch = c0 + ty + r, px = p0 + tx;
    tile[ty + r][tx] = (ch < c_src && px < hw) ? src[(long long)ch * hw + px] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int px = p0 + ty + r, ch = c0 + tx;
    if (ch < c_src && px < hw) dst[(long long)px * c_dst + c_off + ch] = tile[tx][ty + r];
  }
}

int launch_pack_hwc(mp_ctx *ctx, const float *src, int c_src, int h, int w, float *dst, int c_dst,
                    int c_off, hipStream_t st) {
  const int hw = h * w;
  dim3 grid((hw + 31) / 32, (c_src + 31) / 32);
  hipLaunchKernelGGL(chw_to_hwc_kernel, grid, dim3(256), 0, st, src, c_src, hw, dst, c_dst, c_off);
  MP_HIP(ctx, hipGetLastError());
  return MP_OK;
}

}  // namespace mp
