// How fast can every CU stream the SAME few MB out of L2?  (the weight stream of the fused query
// kernels: 4.7 MB re-read by every workgroup for every tile of points)
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream_probe tools/probes/l2_stream_probe.hip && ./l2_stream_probe
// Each wave reads 1 KB (16 B per lane) per load, UNROLL loads in flight, waves of a workgroup walk
// different quarters of the buffer (like the row blocks of the MLP), all workgroups walk the same
// addresses; `stagger` shifts each workgroup's starting point.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4 *__restrict__ buf, long long n16,
                                                     int rounds, int stagger, float *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long per_wave = n16 / 4;  // 16-byte units
  const long long chunks = per_wave / 64;
  f32x4 acc = {0, 0, 0, 0};
  const long long start = stagger ? ((long long)blockIdx.x * 977) % chunks : 0;
  for (int r = 0; r < rounds; ++r) {
    for (long long c0 = 0; c0 < chunks; c0 += UNROLL) {
      f32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        long long c = c0 + u + start;
        if (c >= chunks) c -= chunks;
        v[u] = buf[wave * per_wave + c * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// the same stream through LDS-DMA (global_load_lds_dwordx4: no VGPR destination)
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_lds_kernel(const f32x4 *__restrict__ buf, long long n16,
                                                         int rounds, float *sink) {
  __shared__ f32x4 ring[4][UNROLL][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long per_wave = n16 / 4;
  const long long chunks = per_wave / 64;
  for (int r = 0; r < rounds; ++r) {
    for (long long c0 = 0; c0 < chunks; c0 += UNROLL) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds(buf + wave * per_wave + (c0 + u) * 64 + lane, &ring[wave][u][0], 16, 0, 0);
#endif
      }
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0);
#endif
  __syncthreads();
  const f32x4 v = ring[wave][0][lane];
  if (v[0] == 12345.678f) sink[0] = v[1];
}

template <int UNROLL>
static void run_lds(const f32x4 *buf, long long bytes, int wgs, float *sink) {
  const int rounds = 20;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(stream_lds_kernel<UNROLL>, dim3(wgs), dim3(256), 0, 0, buf, bytes / 16, 2, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(stream_lds_kernel<UNROLL>, dim3(wgs), dim3(256), 0, 0, buf, bytes / 16, rounds, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)bytes * rounds * wgs;
  printf("buffer %5.1f MB  %4d workgroups  LDS-DMA, ring of %2d KB per wave : %7.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz)\n",
         bytes / 1e6, wgs, UNROLL, total / ms / 1e9, total / ms / 1e3 / 256 / 2.1e3);
}

template <int UNROLL>
static void run(const f32x4 *buf, long long bytes, int wgs, int stagger, float *sink) {
  const int rounds = 20;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(stream_kernel<UNROLL>, dim3(wgs), dim3(256), 0, 0, buf, bytes / 16, 2, stagger, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(stream_kernel<UNROLL>, dim3(wgs), dim3(256), 0, 0, buf, bytes / 16, rounds, stagger, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)bytes * rounds * wgs;
  printf("buffer %5.1f MB  %4d workgroups  %2d loads in flight per wave  stagger %d : %7.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz)\n",
         bytes / 1e6, wgs, UNROLL, stagger, total / ms / 1e9, total / ms / 1e3 / 256 / 2.1e3);
}

int main() {
  const long long max_bytes = 64ll << 20;
  f32x4 *buf;
  float *sink;
  hipMalloc(&buf, max_bytes);
  hipMalloc(&sink, 4);
  hipMemset(buf, 0, max_bytes);
  for (int wgs : {256, 512}) {
    run_lds<4>(buf, 5ll << 20, wgs, sink);
    run_lds<8>(buf, 5ll << 20, wgs, sink);
    run_lds<16>(buf, 5ll << 20, wgs, sink);
  }
  for (long long mb : {4ll, 5ll}) {
    const long long bytes = (mb << 20) / 1024 * 1024;
    for (int wgs : {256, 512}) {
      for (int stagger : {0, 1}) {
        run<4>(buf, bytes, wgs, stagger, sink);
        run<8>(buf, bytes, wgs, stagger, sink);
        run<16>(buf, bytes, wgs, stagger, sink);
      }
    }
  }
  return 0;
}
