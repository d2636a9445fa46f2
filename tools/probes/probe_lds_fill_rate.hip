// Probe (GPU box): L2 -> LDS fill rate per CU with (A) LDS-direct 16-byte loads (global_load_lds_dwordx4) and (B) the
// register-staged path conv_box uses today (global_load_dwordx4 -> VGPR -> ds_write_b128), source resident in L2.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_fill_rate.hip -o tools/probes/probe_lds_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 64 * 1024;   // bytes per fill (one "slice": box + weights)
constexpr int PER_THREAD = CHUNK / (256 * 16);

template <bool DIRECT>
__global__ __launch_bounds__(256, 2) void fill(const char* __restrict__ src, size_t src_bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc = 0.f;
  size_t off = ((size_t)blockIdx.x * 7919 * 4096) % (src_bytes - CHUNK);
  for (int it = 0; it < iters; ++it) {
    const char* g = src + off;
    if (DIRECT) {
#pragma unroll
      for (int i = 0; i < PER_THREAD; ++i) {
        const int piece = (i * 4 + wave) * 64;  // wave-uniform LDS base; lane l -> base + 16 l
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + (size_t)(piece + lane) * 16),
                                         (void __attribute__((address_space(3)))*)(lds + piece * 16), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0);
    } else {
      u32x4 r[PER_THREAD];
#pragma unroll
      for (int i = 0; i < PER_THREAD; ++i) r[i] = *reinterpret_cast<const u32x4*>(g + (size_t)(i * 256 + tid) * 16);
#pragma unroll
      for (int i = 0; i < PER_THREAD; ++i) *reinterpret_cast<u32x4*>(lds + (i * 256 + tid) * 16) = r[i];
    }
    __syncthreads();
    acc += ((const float*)lds)[(tid * 37 + it) & 16383];  // consume something so nothing is dropped
    __syncthreads();
    off = (off + 131072) % (src_bytes - CHUNK);
  }
  if (acc == 1234.5f) sink[0] = acc;
}

template <bool DIRECT>
static float run(const char* src, size_t bytes, int wgs, int iters, float* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(fill<DIRECT>, dim3(wgs), dim3(256), CHUNK, 0, src, bytes, 8, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(fill<DIRECT>, dim3(wgs), dim3(256), CHUNK, 0, src, bytes, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const size_t bytes = 2u << 20;  // 2 MB source: L2 resident on every XCD
  char* src; float* sink;
  hipMalloc(&src, bytes); hipMalloc(&sink, 4);
  hipMemset(src, 1, bytes);
  hipFuncSetAttribute(reinterpret_cast<const void*>(fill<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK);
  hipFuncSetAttribute(reinterpret_cast<const void*>(fill<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CHUNK);
  const int iters = 400;
  for (int wgs : {256, 512}) {
    const float md = run<true>(src, bytes, wgs, iters, sink), mr = run<false>(src, bytes, wgs, iters, sink);
    const double total = (double)wgs * iters * CHUNK;
    printf("wgs %d (%.0f per CU): LDS-direct %.3f ms = %.1f GB/s per CU (%.2f TB/s chip) | register-staged %.3f ms = %.1f GB/s per CU\n",
           wgs, wgs / 256.0, md, total / md / 1e6 / 256, total / md / 1e9, mr, total / mr / 1e6 / 256);
  }
  printf("status %d\n", (int)hipGetLastError());
  return 0;
}
