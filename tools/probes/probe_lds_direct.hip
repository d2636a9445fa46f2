// Probe (GPU box): semantics of gfx950's LDS-direct 16-byte global loads (global_load_lds_dwordx4) for the next conv_box
// step: which LDS address does lane l write?  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_lds_direct.hip -o
// tools/probes/probe_lds_direct ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* __restrict__ g, const int* __restrict__ perm, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) ((float*)lds)[i] = -1.f;
  __syncthreads();
  // lane l fetches global piece perm[l] (16 bytes); LDS pointer is wave-uniform: base + 256 bytes
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + perm[lane] * 4),
                                   (void __attribute__((address_space(3)))*)(lds + 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = ((float*)lds)[i];
}

int main() {
  std::vector<float> h(64 * 4);
  std::vector<int> perm(64);
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  for (int l = 0; l < 64; ++l) perm[l] = (l * 7 + 3) % 64;
  float *g, *out;
  int* p;
  hipMalloc(&g, 1024); hipMalloc(&out, 2048); hipMalloc(&p, 256);
  hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(p, perm.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, p, out);
  std::vector<float> o(512);
  hipError_t e = hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
  printf("status %d\n", (int)e);
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const float want = (float)(perm[l] * 4 + j);          // hypothesis: lane l -> LDS base + 16 l
      if (o[64 + l * 4 + j] != want) ok = 0;
    }
  printf("lane l writes LDS[base + 16*l .. +15] with its own global piece: %s\n", ok ? "YES" : "NO");
  printf("first words before base: %g %g | at base: %g %g %g %g | lane1: %g %g\n", o[62], o[63], o[64], o[65], o[66], o[67],
         o[68], o[69]);
  return 0;
}
