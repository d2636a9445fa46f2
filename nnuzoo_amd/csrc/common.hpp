// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the nnUZoo hot path.
// Everything here is wave64 / MFMA specific; there is no portability layer by design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define NNZ_OK 0
#define NNZ_EINVAL (-22)

// hipGetLastError() reports the last error of ANY runtime call of the thread, including benign failures inside other
// libraries (torch / MIOpen probing calls) that happened before we were entered: clear that state in front of every
// launch, so that NNZ_LAUNCH_CHECK attributes only our own launch's failure to us.
#define NNZ_LAUNCH(...)              \
  do {                               \
    (void)hipGetLastError();         \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define NNZ_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

namespace nnz {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// 32x32x16 f16 MFMA, fp32 accumulate.  A: lane l holds A[row l&31][k 8(l>>5)+j]; B: B[k 8(l>>5)+j][col l&31];
// D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5)   (cdna_hip_programming.md §3).
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a 4-row x 16-col block of 16-bit elements,
// lane 4q+p supplies the address of row q / cols 4p..4p+3, lane i receives column i (rows 0..3).
__device__ __forceinline__ i16x4 lds_read_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ unsigned int wave_sum_u32(unsigned int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (unsigned int)__shfl_xor((int)v, o, 64);
  return v;
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// InstanceNorm(affine) + LeakyReLU of 8 fp16 channels (one 16-byte piece) with torch-autocast rounding points: the
// normalisation y = x * scale + shift is one fp32 FMA rounded ONCE to fp16 (what nn.InstanceNorm3d hands on under autocast),
// LeakyReLU then acts on the fp16 value, max(y, y * slope) in packed fp16 (slope in [0, 1]).  16 VALU operations per piece:
// v_fma_mixlo/hi_f16 take the fp16 element and the fp32 {scale, shift} pair directly (no conversions), then v_pk_mul_f16 +
// v_pk_max_f16 on channel pairs.  EVERY place that applies the norm - the apply pass of norm_act.hip and the consumers that
// normalise raw conv outputs while staging them (conv_fprop / conv_wgrad / conv_stem_head / conv_transpose) - goes through
// this function, so they all produce the same bits.  (The first consumer-side version converted to fp32, used a select
// for LeakyReLU and converted back: 36 operations per piece in the staging path cost more than the apply pass it removed.)
typedef _Float16 nnz_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned norm_lrelu_pair(unsigned x, float s0, float t0, float s1, float t1, nnz_h2 slope2) {
  unsigned y;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(y) : "v"(x), "v"(s0), "v"(t0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(y) : "v"(x), "v"(s1), "v"(t1));
  const nnz_h2 h = __builtin_bit_cast(nnz_h2, y);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(h, h * slope2));
}
// sc / sh: the piece's 8 {scale, shift} values; slope2 = {slope, slope} as fp16
template <class V4>
__device__ __forceinline__ V4 norm_lrelu8(V4 r, const float (&sc)[8], const float (&sh)[8], nnz_h2 slope2) {
  V4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = norm_lrelu_pair(r[q], sc[2 * q], sh[2 * q], sc[2 * q + 1], sh[2 * q + 1], slope2);
  return o;
}
// the same with the pairs as they lie in a [c]{scale, shift} fp32 table: 16 consecutive floats at `tp` (16-byte aligned)
template <class V4>
__device__ __forceinline__ V4 norm_lrelu8_tab(V4 r, const float* tp, nnz_h2 slope2) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  V4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 t = *reinterpret_cast<const f4*>(tp + 4 * q);
    o[q] = norm_lrelu_pair(r[q], t[0], t[1], t[2], t[3], slope2);
  }
  return o;
}
__device__ __forceinline__ nnz_h2 slope_pair(float slope) { return nnz_h2{(_Float16)slope, (_Float16)slope}; }

// XCD-aware bijective remap of a linear workgroup id: blocks that are neighbours in the remapped
// order share an XCD (and its L2).  cdna_hip_programming.md §5.5 T1 (bijective variant).
__device__ __forceinline__ unsigned xcd_remap(unsigned orig, unsigned nwg) {
  const unsigned nx = 8;
  unsigned q = nwg / nx, r = nwg % nx, xcd = orig % nx;
  unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / nx;
}

// Pin a wave-uniform kernel argument in SGPRs.  LLVM treats kernarg loads as free to rematerialise and re-issues
// s_load + s_waitcnt lgkmcnt(0) in front of every use inside hot loops (seen in conv_wgrad's prefetch: 14 scalar round
// trips per tile); a v_readfirstlane result cannot be rematerialised from the kernarg segment.
__device__ __forceinline__ int pin_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* pin_uniform(T* ptr) {
  const unsigned long long u = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

// ---- deterministic cross-workgroup sums ------------------------------------------------------------------------------
// fp32 atomicAdd makes a sum depend on the order the workgroups happen to finish in: two runs of the same step differ in
// the last bits, and LeakyReLU / softmax amplify that into visibly different gradients.  Integer addition commutes, so a
// sum kept as a FIXED-POINT integer is bit-identical however the adds interleave.  A partial (a double a workgroup has
// formed in a fixed order) is split into a coarse word in units of 2^-10 and a fine word in units of 2^-58: range
// |total| < 2^53, resolution 3.5e-18 absolute, up to 65 536 partials per accumulator (and replica) before the fine word could wrap -
// i.e. double precision for every magnitude an fp16 tensor's moments or a loss-scaled gradient sum can take.  Non-finite
// partials are counted in a third word and make the total NaN (GradScaler's inf check must still see them).
struct FxAcc {
  long long w[4];  // coarse (2^-10), fine (2^-58), non-finite partials, unused (32-byte records)
};
// Every logical accumulator exists in FX_REP replicas (replica r of record i of a bank of n records: acc[r * n + i]); a
// workgroup adds to replica (its index mod FX_REP); since the words are integers the replicas can be summed in any order
// without changing a bit.  Measured (tools/probes/probe_atomic_rt.hip): atomics to ONE address serialise at ~43 ns each
// (64-bit integer; fp32 adds: 69 ns), which a launch whose workgroups arrive spread over its run time absorbs; what costs
// is the FINALISING workgroup's uncached reads and resets - with 8 replicas x 3 words x (load + store) per record the lone
// last workgroup spent ~20 us per launch (+1 ms per training step) on thousands of 8-byte sc1 transactions.  Two replicas
// keep the per-address queue short enough and the tail at a few microseconds.
constexpr int FX_REP = 2;
__device__ __forceinline__ void fx_add(FxAcc* bank, long i, long n, unsigned rep, double v) {
  unsigned long long* w = reinterpret_cast<unsigned long long*>(bank[(long)(rep & (FX_REP - 1)) * n + i].w);
  if (!(fabs(v) < 0x1p52)) {  // inf, NaN or beyond the range: poison the accumulator
    atomicAdd(w + 2, 1ull);
    return;
  }
  const double h = rint(v * 1024.0);
  const double rem = v - h * (1.0 / 1024.0);  // exact: |rem| <= 2^-11
  atomicAdd(w + 0, (unsigned long long)(long long)h);
  atomicAdd(w + 1, (unsigned long long)(long long)rint(rem * 0x1p58));
}
// read AND reset all replicas of NR consecutive records i .. i + NR - 1 (the workgroup that finalises leaves the bank ready
// for the next launch: no zero-fill kernels).  Agent-scope atomic loads / stores: they bypass the non-coherent caches like
// the adds did, but - unlike read-modify-write atomics - pipeline: ALL loads of the NR records are issued before the first
// store, so a call costs one memory round trip (~3 us in the tail of a launch, where nothing else hides it) however many
// records it takes.  Callers therefore take a thread's records in one call.
template <int NR>
__device__ __forceinline__ void fx_take_n(FxAcc* bank, long i, long n, double (&out)[NR]) {
  long long hi[NR], lo[NR];
  unsigned long long bad[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    hi[q] = 0;
    lo[q] = 0;
    bad[q] = 0;
  }
#pragma unroll
  for (int r = 0; r < FX_REP; ++r)
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      unsigned long long* w = reinterpret_cast<unsigned long long*>(bank[(long)r * n + i + q].w);
      hi[q] += (long long)__hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lo[q] += (long long)__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bad[q] += __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
  for (int r = 0; r < FX_REP; ++r)
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      unsigned long long* w = reinterpret_cast<unsigned long long*>(bank[(long)r * n + i + q].w);
      __hip_atomic_store(w + 0, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(w + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(w + 2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const double v = (double)hi[q] * (1.0 / 1024.0) + (double)lo[q] * 0x1p-58;
    out[q] = bad[q] ? __builtin_nan("") : v;
  }
}
__device__ __forceinline__ double fx_take(FxAcc* bank, long i, long n) {
  double v[1];
  fx_take_n<1>(bank, i, n, v);
  return v[0];
}
// "Was this the last workgroup of the launch to get here?"  Every thread of every workgroup calls it after its fx_add
// calls (uniformly: it contains barriers).  Everything the workgroups exchange travels in device-scope ATOMICS (the
// accumulators, the ticket), so all that is needed is that a workgroup's adds have been performed before its ticket is
// drawn: each thread waits until its own outstanding vector-memory operations have been acknowledged (s_waitcnt
// vmcnt(0): on gfx9 the counter also covers atomics without return and drops when L2 has performed them), the barrier
// joins the threads, one thread draws the ticket.  Two things that do NOT work here: __threadfence() - at agent scope an L2
// write-back + invalidate on this multi-XCD part; in the epilogue of every conv workgroup it doubled the convolution's
// time (measured: conv_box 6.0 -> 12.1 ms per step) - and a workgroup-scope release fence, which compiles to nothing for
// global memory in this mode (checked in the ISA: the adds were still in flight at the barrier, a late add then landed
// after the finalising workgroup had read and reset the accumulator).  The workgroup that draws the last ticket reads the
// totals with agent-scope atomic loads (never cached) and re-arms the counter.  `counter`: a zero-initialised word.
__device__ __forceinline__ bool last_workgroup(unsigned* counter, unsigned nwg) {
  __shared__ unsigned s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(counter, 1u);
    s_last = (t == nwg - 1u);
    if (t == nwg - 1u) atomicExch(counter, 0u);
  }
  __syncthreads();
  return s_last != 0;
}

// LDS hand-over inside one wave (its LDS instructions execute in order; the fences keep the compiler from moving accesses)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic ONLY: waits for this wave's LDS instructions (lgkmcnt) and meets the other
// waves.  __syncthreads() also drains vmcnt - every global load in flight - which is exactly what a prefetch issued before
// the barrier must not pay.  Use only where no global store -> load hand-over between the waves crosses the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The same protocol run by ONE wave (the only wave of the workgroup that issued fx_add calls): no workgroup barrier, the
// other waves go on with their stores / leave.  Wave-uniform result.  Used where the statistics are a side job of a
// kernel whose workgroups would otherwise all stall ~2 us on the ticket's round trip (conv epilogue, norm reductions).
__device__ __forceinline__ bool last_workgroup_wave(unsigned* counter, unsigned nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned last = 0;
  if ((threadIdx.x & 63) == 0) {
    const unsigned t = atomicAdd(counter, 1u);
    last = t == nwg - 1u;
    if (last) atomicExch(counter, 0u);
  }
  return __builtin_amdgcn_readfirstlane((int)last) != 0;
}

// Stream-ordered zero fill by a kernel.  hipMemsetAsync is NOT used anywhere in this library: captured into a hipGraph it
// becomes a fill node whose pattern staging the runtime releases after capture - replays after later allocations then
// fill with whatever landed there (measured on ROCm 7.2 / MI355X, tools/probes/graph_memset_repro.py: NaNs in dbeta).
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void zero_fill_kernel(u32x4* __restrict__ p, size_t nvec, uint32_t* __restrict__ tail,
                                                        int ntail) {
  const size_t stride = (size_t)gridDim.x * 256;
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}
// bytes must be a multiple of 4 (every caller zeroes float / 64-bit counters)
inline hipError_t zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  char* c = (char*)ptr;
  size_t head = ((uintptr_t)c & 15) ? 16 - ((uintptr_t)c & 15) : 0;  // words up to the first 16-byte boundary
  if (head > bytes) head = bytes;
  if (head) {
    NNZ_LAUNCH(zero_fill_kernel<0>, dim3(1), dim3(256), 0, s, (u32x4*)nullptr, (size_t)0, (uint32_t*)c, (int)(head / 4));
    c += head;
    bytes -= head;
  }
  const size_t nvec = bytes / 16;
  const int ntail = (int)((bytes - nvec * 16) / 4);
  if (nvec || ntail) {
    size_t wg = (nvec + 255) / 256;
    if (wg > 4096) wg = 4096;
    if (wg < 1) wg = 1;
    NNZ_LAUNCH(zero_fill_kernel<0>, dim3((unsigned)wg), dim3(256), 0, s, (u32x4*)c, nvec, (uint32_t*)(c + nvec * 16),
               ntail);
  }
  return hipGetLastError();
}

// Deterministic second stage of a cross-workgroup sum: out[e] = sum_p part[p * stride + e], p = 0 .. nparts - 1 in a FIXED
// grouping (thread = (element, every 4th partial of its range), the four sums folded as (g0 + g1) + (g2 + g3)) - the form the
// convolution weight gradients use (conv_wgrad.hip wgrad_reduce_kernel).  The first stage stores each workgroup's partial with
// plain stores; nothing needs zero-filling and no float atomics are involved, so the result is bit-identical from run to run.
// Grid row y folds partials [y * per, (y + 1) * per) into out + y * E: many partials of few elements (a per-channel sum over a
// thousand tiles) go through two levels - fold_partials() does that when the caller's workspace has room behind the partials.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ part, int nparts, long stride, long E,
                                                            float* __restrict__ out, int per) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x;
  const int el = tid & 63, sg = tid >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  const int p0 = blockIdx.y * per;
  const int p1 = p0 + per < nparts ? p0 + per : nparts;
  float s0 = 0.f, s1 = 0.f;
  if (e < E) {
    int q = p0 + sg;
    for (; q + 4 < p1; q += 8) {  // two loads in flight
      s0 += part[(size_t)q * stride + e];
      s1 += part[(size_t)(q + 4) * stride + e];
    }
    if (q < p1) s0 += part[(size_t)q * stride + e];
  }
  red[sg][el] = s0 + s1;
  __syncthreads();
  if (sg == 0 && e < E) out[(size_t)blockIdx.y * E + e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}
constexpr int FOLD_GROUPS = 32;   // first-level groups of the two-level fold
// floats of scratch the two-level fold wants behind the partials (0: one level)
inline long fold_partials_scratch_floats(int nparts, long E) { return nparts > 8 * FOLD_GROUPS ? (long)FOLD_GROUPS * E : 0; }
inline hipError_t fold_partials(const float* part, int nparts, long stride, long E, float* out, hipStream_t s,
                                float* scratch = nullptr) {
  if (E <= 0) return hipSuccess;
  const unsigned gx = (unsigned)((E + 63) / 64);
  if (scratch && nparts > 8 * FOLD_GROUPS) {
    const int per = (nparts + FOLD_GROUPS - 1) / FOLD_GROUPS;
    const int groups = (nparts + per - 1) / per;
    NNZ_LAUNCH(fold_partials_kernel<0>, dim3(gx, groups), dim3(256), 0, s, part, nparts, stride, E, scratch, per);
    NNZ_LAUNCH(fold_partials_kernel<0>, dim3(gx, 1), dim3(256), 0, s, (const float*)scratch, groups, E, E, out, groups);
  } else {
    NNZ_LAUNCH(fold_partials_kernel<0>, dim3(gx, 1), dim3(256), 0, s, part, nparts, stride, E, out, nparts);
  }
  return hipGetLastError();
}

// Dynamic-LDS opt-in above 64 KB (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE function attribute.  One
// cache per kernel instantiation remembers the largest size registered on each device; atomics make concurrent host
// threads safe (the worst case is a redundant, idempotent hipFuncSetAttribute call).
struct DynLdsCache {
  static constexpr int MAX_DEV = 32;
  std::atomic<int> granted[MAX_DEV];
};
inline hipError_t ensure_dyn_lds(const void* kern, int lds_bytes, DynLdsCache& cache) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool cached = dev >= 0 && dev < DynLdsCache::MAX_DEV;
  if (cached && lds_bytes <= cache.granted[dev].load(std::memory_order_acquire)) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess || !cached) return e;
  int seen = cache.granted[dev].load(std::memory_order_relaxed);
  while (seen < lds_bytes && !cache.granted[dev].compare_exchange_weak(seen, lds_bytes, std::memory_order_release)) {
  }
  return hipSuccess;
}

}  // namespace nnz
