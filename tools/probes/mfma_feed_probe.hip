// What does FEEDING the matrix pipe cost?  mfma_peak_probe.hip shows v_mfma_f32_32x32x2_f32 sustains
// 154.5 TFLOP/s from registers; the fused query kernels reach 137-143 and the convolutions less.
// This probe builds the real kernels' inner loop up from that register-only loop: groups of 8 MFMAs
// on 4 accumulators whose A operands arrive, LPG 16-byte-per-lane loads per group (1 KB per wave and
// load, issued PF groups ahead as in seg_main), by one of four routes:
//   stream : buffer_load_dwordx4 from a 4.7 MB buffer every workgroup walks (the weight stream: L2)
//   hot    : the same instruction, always the same 1 KB (vector-L1 hit)
//   lds    : ds_read_b128 of data that sits in LDS
//   dma    : global_load_lds_dwordx4 into an LDS ring (no VGPR destination) + ds_read_b128 PF groups later
// and optionally the B operands through 2 ds_read_b128 per group (the activation tile).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/mfma_feed_probe tools/probes/mfma_feed_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { STREAM = 1, HOT = 2, LDSRES = 3, DMA = 4 };
constexpr int kMaxL = 4, kRing = 4;  // loads per group, ring slots (PF <= 3)

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int MODE, int LPG, int PF, bool BLDS>
__global__ __launch_bounds__(256) void feed_kernel(const float *w, int w_bytes, int groups, float *out, int stagger) {
  __shared__ f32x4 ring_lds[4][kRing][kMaxL][64];  // 64 KB: dma ring / resident A data
  __shared__ f32x4 b_lds[64][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int L = LPG > 0 ? LPG : 1, RS = PF + 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, w_bytes, 0x00020000);
  const int per_wave = w_bytes / 4;                  // bytes of this wave's quarter
  const int frags = per_wave / 1024;                 // 1 KB fragments in it
  const int wave_off = wave * per_wave;
  for (int i = threadIdx.x; i < 4 * kRing * kMaxL * 64; i += 256)
    (&ring_lds[0][0][0][0])[i] = f32x4{0.001f * i, 1.0f, -0.5f, 0.25f};
  for (int i = threadIdx.x; i < 64 * 8; i += 256) (&b_lds[0][0])[i] = f32x4{0.5f, -0.25f, 0.125f, 1.0f + 0.001f * i};
  __syncthreads();

  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[m][t] = 0.0f;
  f32x4 ring[RS][L];
  f32x4 bcur[2] = {f32x4{0.3f, -0.7f, 0.2f, 0.9f}, f32x4{-0.4f, 0.6f, 0.8f, -0.1f}};
#pragma unroll
  for (int d = 0; d < RS; ++d)
#pragma unroll
    for (int l = 0; l < L; ++l) ring[d][l] = f32x4{0.11f * (d + 1), -0.3f, 0.7f, 0.05f * (l + 1)};

  // next fragment of this wave's quarter (wave-uniform, wraps); stagger: every workgroup starts somewhere else
  int f_next = stagger ? (int)((blockIdx.x * 977u) % (unsigned)frags) : 0;
  auto issue = [&](int g, int slot) {  // the loads of group g
#pragma unroll
    for (int l = 0; l < LPG; ++l) {
      const int f = f_next;
      f_next = f_next + 1 == frags ? 0 : f_next + 1;
      if (MODE == STREAM) ring[slot][l] = bload(rs, lane * 16, wave_off + f * 1024);
      if (MODE == HOT) ring[slot][l] = bload(rs, lane * 16, wave_off + (l & 1) * 1024);
      if (MODE == LDSRES) ring[slot][l] = ring_lds[wave][g & (kRing - 1)][l][lane];
#if defined(__HIP_DEVICE_COMPILE__)
      if (MODE == DMA)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const f32x4 *>(w) + (wave_off + f * 1024) / 16 + lane,
                                         &ring_lds[wave][g & (kRing - 1)][l][0], 16, 0, 0);
#endif
    }
  };
  if (MODE != DMA) {
#pragma unroll
    for (int d = 0; d < PF; ++d) issue(d, d);
  } else {
    for (int d = 0; d < PF; ++d) issue(d, 0);
  }
#pragma unroll 1
  for (int g0 = 0; g0 < groups; g0 += RS) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int g = g0 + r;
      asm volatile("" ::: "memory");  // LDS contents are "new" every group: no hoisting of the ds_reads
      issue(g + PF, (r + PF) % RS);
      f32x4 bnxt[2] = {bcur[0], bcur[1]};
      if (BLDS) {
        bnxt[0] = b_lds[lane][(2 * g) & 7];
        bnxt[1] = b_lds[lane][(2 * g + 1) & 7];
      }
      if (MODE == DMA && LPG > 0) {
        // Software pipeline of the LDS route: the transfers of group g + 1 (everything but the
        // (PF - 1) LPG youngest) have landed -> read them into the ring registers now, one group
        // ahead of their MFMAs; the reads issued one iteration ago are waited for just before use.
        // (Inline asm: left to itself hipcc puts s_waitcnt vmcnt(0) in front of every ds_read that
        // may alias an LDS-DMA destination, which would serialise the ring.)
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int n = (PF - 1) * LPG;
        __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14));
#pragma unroll
        for (int l = 0; l < LPG; ++l) {
          const unsigned addr = (unsigned)(size_t)&ring_lds[wave][(g + 1) & (kRing - 1)][l][lane];
          asm volatile("ds_read_b128 %0, %1" : "=v"(ring[(r + 1) % RS][l]) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LPG));
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int q = 2 * i + n;
          acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[r % RS][q % L][i], bcur[n][i], acc[q & 3], 0, 0, 0);
        }
      bcur[0] = bnxt[0];
      bcur[1] = bnxt[1];
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[m][t];
  if (s == 123.456f) out[threadIdx.x] = s;
}

static float *g_w, *g_out;
static int g_wbytes = 4864 * 1024, g_cus, g_stagger = 0;

template <int MODE, int LPG, int PF, bool BLDS>
static void run(const char *name, int wpc) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = g_cus * wpc;
  int groups = 40000 / (PF + 1) * (PF + 1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((feed_kernel<MODE, LPG, PF, BLDS>), dim3(grid), dim3(256), 0, 0, g_w, g_wbytes, groups, g_out, g_stagger);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double mfmas = (double)grid * 4 * groups * 8;
  const double tf = mfmas * 4096 / (ms * 1e-3) / 1e12;
  // cycles of a SIMD's matrix pipe per group beyond the 8 x 64 of the MFMAs, per wave sharing it
  const double cyc_per_group = ms * 1e-3 * 2.38e9 / groups / wpc;
  printf("  %-7s LPG %d PF %d B-from-LDS %d  %d WG/CU  buffer %.2f MB stagger %d  %7.2f ms  %6.1f TFLOP/s  %6.1f cycles per group and wave (512 = MFMAs only)\n",
         name, LPG, PF, (int)BLDS, wpc, g_wbytes / 1048576.0, g_stagger, ms, tf, cyc_per_group);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  g_cus = p.multiProcessorCount;
  std::vector<float> h((16 << 20) / 4);
  srand(2);
  for (auto &v : h) v = (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0);
  (void)hipMalloc(&g_w, 16 << 20);
  (void)hipMalloc(&g_out, 4096);
  (void)hipMemcpy(g_w, h.data(), 16 << 20, hipMemcpyHostToDevice);
  // the weight stream against the L2 (4 MB per XCD): in-phase and staggered workgroups, three set sizes
  printf("weight-set size and workgroup phase (2 workgroups per CU)\n");
  for (int mb4 : {8, 14, 19, 24, 40}) {  // quarter MB
    g_wbytes = mb4 * 256 * 1024;
    for (int st = 0; st < 2; ++st) {
      g_stagger = st;
      run<STREAM, 1, 1, false>("stream", 2);
      run<STREAM, 1, 3, false>("stream", 2);
      run<STREAM, 4, 1, false>("stream", 2);
      run<STREAM, 4, 3, false>("stream", 2);
    }
  }
  g_wbytes = 4864 * 1024;
  g_stagger = 0;
  for (int wpc = 1; wpc <= 3; ++wpc) {
    printf("%d workgroup(s) per CU\n", wpc);
    run<STREAM, 0, 1, false>("none", wpc);
    run<STREAM, 0, 1, true>("none", wpc);
    run<STREAM, 1, 1, false>("stream", wpc);
    run<STREAM, 2, 1, false>("stream", wpc);
    run<STREAM, 4, 1, false>("stream", wpc);
    run<STREAM, 2, 3, false>("stream", wpc);
    run<STREAM, 2, 1, true>("stream", wpc);
    run<HOT, 1, 1, false>("hot", wpc);
    run<HOT, 2, 1, false>("hot", wpc);
    run<HOT, 4, 1, false>("hot", wpc);
    run<LDSRES, 1, 1, false>("lds", wpc);
    run<LDSRES, 2, 1, false>("lds", wpc);
    run<LDSRES, 4, 1, false>("lds", wpc);
    run<DMA, 1, 2, false>("dma", wpc);
    run<DMA, 2, 2, false>("dma", wpc);
    run<DMA, 4, 2, false>("dma", wpc);
    run<DMA, 2, 3, false>("dma", wpc);
  }
  return 0;
}
