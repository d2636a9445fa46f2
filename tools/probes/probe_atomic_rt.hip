// probe: what does "last workgroup finalises" cost in the tail of a STREAMING kernel (MI355X)?
// Each workgroup stores 64 KB (like a conv output tile), adds 128 fixed-point words to shared accumulators (8 replicas),
// then either just exits (A), or waits for ALL its memory operations + barrier + ticket (B: the first design), or lets only
// wave 0 add / wait / draw the ticket BEFORE it issues its own stores (C).
// build: hipcc --offload-arch=gfx950 -O3 -o probe_atomic_rt probe_atomic_rt.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(u64* acc, unsigned* counter, f32x4* out, int nwg) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ unsigned s_last;
  f32x4* o = out + (size_t)blockIdx.x * 4096;
  const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
  u64* a = acc + (size_t)(blockIdx.x & 7) * 1024;
  if (MODE == 2) {  // C: wave 0 alone runs the protocol, before its stores
    if (wave == 0) {
      atomicAdd(a + 4 * lane, 3ull); atomicAdd(a + 4 * lane + 1, 5ull);
      atomicAdd(a + 4 * (lane + 64), 3ull); atomicAdd(a + 4 * (lane + 64) + 1, 5ull);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned last = 0;
      if (lane == 0) { unsigned t = atomicAdd(counter, 1u); last = t == (unsigned)nwg - 1; if (last) atomicExch(counter, 0u); }
      last = __shfl(last, 0, 64);
      if (last) for (int i = lane; i < 1024 * 8; i += 64) __hip_atomic_store(acc + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < 4096; i += 256) o[i] = v;
    return;
  }
  for (int i = tid; i < 4096; i += 256) o[i] = v;
  if (tid < 128) { atomicAdd(a + 4 * tid, 3ull); atomicAdd(a + 4 * tid + 1, 5ull); }
  if (MODE == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) { unsigned t = atomicAdd(counter, 1u); s_last = t == (unsigned)nwg - 1; if (s_last) atomicExch(counter, 0u); }
  __syncthreads();
  if (s_last) for (int i = tid; i < 1024 * 8; i += 256) __hip_atomic_store(acc + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
float run(u64* acc, unsigned* c, f32x4* out, int nwg, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, acc, c, out, nwg);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, acc, c, out, nwg);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}
int main() {
  u64* acc; unsigned* c; f32x4* out;
  (void)hipMalloc(&acc, 1 << 20); (void)hipMalloc(&c, 256); (void)hipMalloc(&out, (size_t)8192 * 65536);
  (void)hipMemset(acc, 0, 1 << 20); (void)hipMemset(c, 0, 256);
  for (int nwg : {256, 2048, 8192}) {
    printf("nwg %5d (64 KB stores each): A fire-and-forget %.1f | B all threads wait + barrier + ticket %.1f | C wave 0 runs the protocol before its stores %.1f  (us per launch)\n",
           nwg, run<0>(acc, c, out, nwg, 50), run<1>(acc, c, out, nwg, 50), run<2>(acc, c, out, nwg, 50));
    (void)hipMemset(acc, 0, 1 << 20); (void)hipMemset(c, 0, 256);
  }
  return 0;
}
