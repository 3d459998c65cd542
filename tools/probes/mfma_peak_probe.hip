// What does v_mfma_f32_32x32x2_f32 sustain on this box?  A register-only loop: every wave keeps
// NACC independent 32x32 accumulators and issues MFMAs back to back, no memory, no LDS.  The
// nominal peak (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz = 157.3 TFLOP/s) assumes the boost
// clock.  Two operand sets: CONSTANT (a = 1, b = 1e-7: few bits toggle) and RANDOM (eight
// random-mantissa operand pairs per lane, rotated every MFMA: the switching activity of real data).
// Every workgroup also reads the shader clock (s_memtime) and the 100 MHz reference (s_memrealtime)
// around its loop, so the average core clock during the launch is printed next to the rate.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/mfma_peak_probe tools/probes/mfma_peak_probe.hip
//   tools/probes/bin/mfma_peak_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float *out, const float *operands, int iters,
                                                 unsigned long long *clocks) {
  f32x16 acc[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[m][t] = 0.0f;
  float a[8], b[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    a[r] = RANDOM ? operands[(2 * r) * 256 + threadIdx.x] : 1.0f + threadIdx.x * 1e-9f;
    b[r] = RANDOM ? operands[(2 * r + 1) * 256 + threadIdx.x] : 1e-7f;
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int m = 0; m < NACC; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[(r + m) & 7], acc[m], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    clocks[2 * blockIdx.x] = c1 - c0;
    clocks[2 * blockIdx.x + 1] = w1 - w0;
  }
  float s = 0.0f;
#pragma unroll
  for (int m = 0; m < NACC; ++m)
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[m][t];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NACC, bool RANDOM>
static void run(int wg_per_cu, int n_cu, float ms_target, const float *operands) {
  float *out;
  unsigned long long *clocks;
  const int grid = n_cu * wg_per_cu;
  (void)hipMalloc(&out, 1024);
  (void)hipMalloc(&clocks, grid * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, RANDOM>), dim3(grid), dim3(256), 0, 0, out, operands, iters, clocks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), clocks, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, ref = 0;
    for (int g = 0; g < grid; ++g) {
      cyc += (double)h[2 * g];
      ref += (double)h[2 * g + 1];
    }
    const double flop = (double)grid * 4 * iters * 8 * NACC * (2.0 * 32 * 32 * 2);
    if (rep)
      printf("  %-8s NACC %d  %d WG/CU  %8.2f ms  %7.1f TFLOP/s   shader clock %.0f MHz (cycles / 100 MHz reference ticks)\n",
             RANDOM ? "random" : "constant", NACC, wg_per_cu, ms, flop / (ms * 1e-3) / 1e12, cyc / ref * 100.0);
    if (rep == 0 && ms < ms_target) iters = (int)(iters * ms_target / ms);
  }
  (void)hipFree(out);
  (void)hipFree(clocks);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s  %d CUs  clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  std::vector<float> h(16 * 256);
  srand(1);
  for (auto &v : h) v = (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0);
  float *operands;
  (void)hipMalloc(&operands, h.size() * 4);
  (void)hipMemcpy(operands, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (float target : {5.0f, 300.0f}) {
    printf("launches of ~%.0f ms\n", target);
    run<4, false>(2, p.multiProcessorCount, target, operands);
    run<4, true>(2, p.multiProcessorCount, target, operands);
    run<4, true>(1, p.multiProcessorCount, target, operands);
  }
  // dependent chains: NACC accumulators per wave = every MFMA waits for the one NACC issues back
  printf("dependent chains (launches of ~50 ms)\n");
  run<1, false>(2, p.multiProcessorCount, 50.0f, operands);
  for (int wg = 1; wg <= 4; ++wg) run<1, true>(wg, p.multiProcessorCount, 50.0f, operands);
  for (int wg = 1; wg <= 2; ++wg) run<2, true>(wg, p.multiProcessorCount, 50.0f, operands);
  for (int wg = 1; wg <= 2; ++wg) run<3, true>(wg, p.multiProcessorCount, 50.0f, operands);
  return 0;
}
