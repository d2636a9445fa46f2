// Probe (GPU box): MFMA rate of a tap-loop-like inner loop as a function of LDS fragment reads per MFMA.
// A wave reads NA "weight" fragments and NB "voxel" fragments (16 bytes per lane each, conflict-free ds_read_b128) per
// step and issues NA x NB v_mfma_f32_32x32x16_f16: (1,1) = the 2x4x8 small tile (2 KB per MFMA), (1,4) = the 8x8x8 x 32-cout
// tile (1.25 KB), (2,2) = 4x8x8 x 64-cout (1 KB), (2,4), (4,4) = larger register tiles.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_mfma_ratio.hip -o tools/probes/probe_lds_mfma_ratio
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NA, int NB>
__global__ __launch_bounds__(256, 2) void loop(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += 256) ((float*)lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  int off = (tid >> 6) * 4096;
  for (int it = 0; it < iters; ++it) {
    f16x8 fa[NA], fb[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) fa[a] = *reinterpret_cast<const f16x8*>(lds + ((off + a * 1024 + lane * 16) & 65535));
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = *reinterpret_cast<const f16x8*>(lds + ((off + 8192 + b * 1024 + lane * 16) & 65535));
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    off += 2048;
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][7];
  if (s == 1234.5f) sink[0] = s;
}

template <int NA, int NB>
static void run(float* sink) {
  auto k = loop<NA, NB>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int wgs = 512, iters = 4000;
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 65536, 0, 100, sink);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 65536, 0, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double flop = (double)wgs * 4 * iters * NA * NB * 2.0 * 32 * 32 * 16;
  printf("NA=%d NB=%d: %.2f KB LDS reads per MFMA -> %.0f TFLOP/s (%.3f ms)\n", NA, NB, (NA + NB) * 1.0 / (NA * NB),
         flop / ms / 1e9, ms);
}

int main() {
  float* sink;
  hipMalloc(&sink, 4);
  run<1, 1>(sink); run<1, 2>(sink); run<1, 4>(sink); run<2, 2>(sink); run<2, 4>(sink); run<4, 4>(sink);
  printf("status %d\n", (int)hipGetLastError());
  return 0;
}
