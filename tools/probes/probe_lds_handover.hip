// Probe (GPU box): what the per-slice LDS hand-over of conv_box costs.  Per slice a workgroup writes 64 KB of staged data
// (16 ds_write_b128 per thread) and runs 27 tap steps of (NA x NB) MFMAs on fragments read from LDS.
//   MODE 0: barrier - write all - barrier - taps            (conv_box today: single buffer)
//   MODE 1: double buffer: the writes of the NEXT slice are interleaved with the taps (one per tap), one barrier per slice
// No global loads: isolates the LDS write/read/barrier interaction.  Compare with probe_lds_mfma_ratio (bare loop).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_handover.hip -o tools/probes/probe_lds_handover
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BUF = 64 * 1024;

template <int NA, int NB, int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? 2 : 1) void loop(int slices, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  u32x4 st = {0x3c003c00u + tid, 0x3c003c00u, 0x38003800u, 0x3c003800u};
  auto taps = [&](const char* buf, char* wbuf, bool wr) {
#pragma unroll 3
    for (int t = 0; t < 27; ++t) {
      f16x8 fa[NA], fb[NB];
      const int off = wave * 2048 + t * 1536;
#pragma unroll
      for (int a = 0; a < NA; ++a) fa[a] = *reinterpret_cast<const f16x8*>(buf + ((off + a * 1024 + lane * 16) & (BUF - 1)));
#pragma unroll
      for (int b = 0; b < NB; ++b)
        fb[b] = *reinterpret_cast<const f16x8*>(buf + ((off + 16384 + b * 1024 + lane * 16) & (BUF - 1)));
      if (wr && t < 16) *reinterpret_cast<u32x4*>(wbuf + (t * 256 + tid) * 16) = st;
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  };
  if (MODE == 0) {
    for (int s = 0; s < slices; ++s) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(lds + (i * 256 + tid) * 16) = st;
      __syncthreads();
      taps(lds, nullptr, false);
      st[0] += 1;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(lds + (i * 256 + tid) * 16) = st;
    for (int s = 0; s < slices; ++s) {
      __syncthreads();
      taps(lds + (s & 1) * BUF, lds + ((s + 1) & 1) * BUF, true);
      st[0] += 1;
    }
  }
  float v = 0.f;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) v += acc[a][b][0] + acc[a][b][9];
  if (v == 1234.5f) sink[0] = v;
}

template <int NA, int NB, int MODE>
static void run(float* sink) {
  auto k = loop<NA, NB, MODE>;
  const int ldsb = MODE == 0 ? BUF : 2 * BUF;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  const int wgs = MODE == 0 ? 512 : 256, slices = 200;
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), ldsb, 0, 4, sink);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), ldsb, 0, slices, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double flop = (double)wgs * 4 * slices * 27 * NA * NB * 2.0 * 32 * 32 * 16;
  printf("NA=%d NB=%d mode %d (%s, %d WG/CU): %.0f TFLOP/s (%.3f ms)\n", NA, NB, MODE,
         MODE == 0 ? "single buffer, 2 barriers" : "double buffer, 1 barrier", MODE == 0 ? 2 : 1, flop / ms / 1e9, ms);
}

int main() {
  float* sink;
  hipMalloc(&sink, 4);
  run<1, 4, 0>(sink); run<1, 4, 1>(sink); run<2, 2, 0>(sink); run<2, 2, 1>(sink); run<1, 1, 0>(sink); run<1, 1, 1>(sink);
  printf("status %d\n", (int)hipGetLastError());
  return 0;
}
