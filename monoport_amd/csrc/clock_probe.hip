// What the f32 matrix pipe of THIS box sustains right now, and at which shader clock (mp_mfma_clock_probe,
// bench.py's `roofline.sustained`).  The roofline's `peak` is the nominal 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz
// = 157.3 TFLOP/s; boxes of one pool differ by a few per cent in the clock they hold under a matrix load, and the
// same query launch then lands at 0.78 or at 0.82 of that number.  A register-only loop -- every wave keeps four
// independent 32x32 accumulators and issues v_mfma_f32_32x32x2_f32 back to back on operands with random mantissas
// (the switching activity of real data), no memory, no LDS -- timed with HIP events; every workgroup reads the
// shader clock counter (s_memtime) and the constant 100 MHz reference (s_memrealtime) around its loop, so the
// average core clock DURING the launch comes with the rate.  Measurement only: nothing on the product path calls it.
#include <vector>

#include "mp_internal.h"
#include "query_common.h"

namespace mp {

constexpr int kProbeAcc = 4;

__global__ __launch_bounds__(256) void mfma_clock_probe_kernel(int iters, unsigned long long *clocks, float *sink) {
  f32x16 acc[kProbeAcc];
#pragma unroll
  for (int m = 0; m < kProbeAcc; ++m)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[m][t] = 0.0f;
  float a[8], b[8];
  uint32_t s = 0x9E3779B9u * (threadIdx.x + 1u);
#pragma unroll
  for (int r = 0; r < 8; ++r) {  // values in [1, 2) / [-2, -1): random mantissas, products that neither grow nor vanish
    s = s * 1664525u + 1013904223u;
    a[r] = __uint_as_float(0x3F800000u | (s >> 9));
    s = s * 1664525u + 1013904223u;
    b[r] = __uint_as_float(((r & 1) ? 0xBF800000u : 0x3F800000u) | (s >> 9)) * 1e-3f;
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int m = 0; m < kProbeAcc; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[(r + m) & 7], acc[m], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    clocks[2 * blockIdx.x] = c1 - c0;
    clocks[2 * blockIdx.x + 1] = w1 - w0;
  }
  float t = 0.0f;
#pragma unroll
  for (int m = 0; m < kProbeAcc; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[m][e];
  if (t == 123.456f) sink[threadIdx.x] = t;  // keeps the loop alive
}

// out[0] = TFLOP/s of the timed launch, out[1] = shader clock in MHz (cycles per 100 MHz reference tick, averaged
// over the workgroups), out[2] = its duration in ms, out[3] = workgroups.  Synchronises `st`.
int launch_mfma_clock_probe(mp_ctx *ctx, float ms_target, double *out, hipStream_t st) {
  const int grid = 2 * cus_of(ctx, st);  // two 4-wave workgroups per CU = two waves per SIMD, as the query kernels run
  unsigned long long *clocks = nullptr;
  float *sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = MP_OK;
  auto done = [&](int code) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (clocks) (void)hipFree(clocks);
    if (sink) (void)hipFree(sink);
    return code;
  };
  if (hipMalloc(&clocks, (size_t)grid * 16) != hipSuccess || hipMalloc(&sink, 1024) != hipSuccess)
    return done(fail(ctx, MP_ERR_NOMEM, "mp_mfma_clock_probe: hipMalloc failed"));
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    return done(fail(ctx, MP_ERR_HIP, "mp_mfma_clock_probe: hipEventCreate failed"));
  int iters = 600;  // ~1 ms: wakes the clocks up (rep 0, not used), calibrates the timed launch (rep 1)
  float ms = 0.0f;
  for (int rep = 0; rep < 3 && rc == MP_OK; ++rep) {
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_clock_probe_kernel, dim3((unsigned)grid), dim3(256), 0, st, iters, clocks, sink);
    (void)hipEventRecord(e1, st);
    if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
      rc = fail(ctx, MP_ERR_HIP, "mp_mfma_clock_probe: launch failed");
    if (rep == 1 && rc == MP_OK) {
      const double scale = ms > 0.0f ? ms_target / ms : 1.0;
      const double want = iters * scale;
      iters = want < 1000.0 ? 1000 : want > 4.0e6 ? 4000000 : (int)want;
    }
  }
  if (rc != MP_OK) return done(rc);
  std::vector<unsigned long long> h(2 * (size_t)grid);
  if (hipMemcpy(h.data(), clocks, (size_t)grid * 16, hipMemcpyDeviceToHost) != hipSuccess)
    return done(fail(ctx, MP_ERR_HIP, "mp_mfma_clock_probe: read-back failed"));
  double cyc = 0.0, ref = 0.0;
  for (int g = 0; g < grid; ++g) {
    cyc += (double)h[2 * g];
    ref += (double)h[2 * g + 1];
  }
  const double flop = (double)grid * 4 * iters * 8 * kProbeAcc * (2.0 * 32 * 32 * 2);
  out[0] = flop / (ms * 1e-3) / 1e12;
  out[1] = ref > 0.0 ? cyc / ref * 100.0 : 0.0;
  out[2] = ms;
  out[3] = grid;
  return done(MP_OK);
}

}  // namespace mp
