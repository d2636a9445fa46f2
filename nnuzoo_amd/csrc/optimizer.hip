// Fused optimizer tail of nnUNetTrainer.train_step (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:
// 1131-1139): grad_scaler.unscale_ -> clip_grad_norm_(12) -> SGD(momentum 0.99, nesterov, weight decay) step with the
// GradScaler's skip-on-inf, as two HBM-bound passes over the flat gradient arena the backward schedule fills
// (nnuzoo_amd/nets/plain_conv_unet.py):
//   pass 1  sum of squares + non-finite flag of the (still scaled) gradients               (1 read of the arena)
//   pass 2  g = grad * inv_scale * clip;  g += wd * p;  buf = mom * buf + g (buf = g on the first step);
//           p -= lr * (g + mom * buf)                                        (arena read, param + momentum read-write)
// replacing torch's foreach unscale (read+write), norm (read), clip multiply (read+write) and SGD (3 reads, 2 writes).
// Everything stays on the device: no host synchronisation for the inf check (torch's GradScaler.step does `.item()`).
#include "common.hpp"
#include <string.h>

namespace nnz {

struct SgdChunk {
  float* param;
  float* mom;
  long arena_off;  // element offset of this chunk's gradients in the arena
  int n;           // elements in this chunk
  int pad;
};

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out2,
                                                         FxAcc* acc, unsigned* counter) {
  __shared__ float red[4][2];
  float s = 0.f, bad = 0.f;
  const long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  // four independent 16-byte loads in flight per lane (the loop is latency-bound otherwise)
  const long stride = (long)gridDim.x * 256;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = g4[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s += v[k][e] * v[k][e];
        bad += (__builtin_isnan(v[k][e]) || __builtin_isinf(v[k][e])) ? 1.f : 0.f;
      }
  }
  for (; i < n4; i += stride) {
    const f32x4 v = g4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s += v[e] * v[e];
      bad += (__builtin_isnan(v[e]) || __builtin_isinf(v[e])) ? 1.f : 0.f;
    }
  }
  if (blockIdx.x == 0)
    for (long j = (n4 << 2) + threadIdx.x; j < n; j += 256) {
      const float v = g[j];
      s += v * v;
      bad += (__builtin_isnan(v) || __builtin_isinf(v)) ? 1.f : 0.f;
    }
  s = wave_sum(s);
  bad = wave_sum(bad);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = s;
    red[wave][1] = bad;
  }
  __syncthreads();
  if (acc) {
    // deterministic: fixed-point sum across workgroups (common.hpp), the last one writes the two floats.  A non-finite
    // gradient makes the block's partial non-finite, which poisons the accumulator: the total reads back NaN and the
    // count of bad entries below stays the inf-skip flag.  Wave 0 alone runs the protocol (the others leave).
    if (threadIdx.x >= 64) return;
    if (threadIdx.x == 0) {
      fx_add(acc, 0, 2, blockIdx.x, (double)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])));
      fx_add(acc, 1, 2, blockIdx.x, (double)((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])));
    }
    if (last_workgroup_wave(counter, gridDim.x) && threadIdx.x == 0) {
      out2[0] = (float)fx_take(acc, 0, 2);
      out2[1] = (float)fx_take(acc, 1, 2);
    }
    return;
  }
  if (threadIdx.x == 0) {
    atomicAdd(out2 + 0, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    const float b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    if (b > 0.f) atomicAdd(out2 + 1, b);
  }
}

__global__ __launch_bounds__(256) void sgd_nesterov_kernel(const SgdChunk* __restrict__ chunks,
                                                           const float* __restrict__ arena,
                                                           float* __restrict__ stats2,
                                                           const float* __restrict__ inv_scale_dev, float max_norm,
                                                           float lr, float momentum, float wd, int first_step) {
  // sum of squares of the scaled gradients; a non-finite sum also marks the step as skipped (overflowed squares)
  const float ss = stats2[0];
  const bool overflow = !(ss == ss) || __builtin_isinf(ss);
  if (stats2[1] > 0.f || overflow) {  // GradScaler semantics: leave params and momentum
    // finite gradients whose squares overflowed: the caller's found_inf flag (stats2[1]) must say "skipped" too, so
    // that the loss scale backs off exactly when this kernel skipped (every workgroup takes this branch: no race)
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) stats2[1] = 1.f;
    return;
  }
  const float inv_scale = inv_scale_dev ? inv_scale_dev[0] : 1.f;
  const float total_norm = sqrtf(ss) * inv_scale;
  float clip = max_norm / (total_norm + 1e-6f);
  clip = clip > 1.f ? 1.f : clip;
  const float mult = inv_scale * clip;
  const SgdChunk c = chunks[blockIdx.x];
  const float* g = arena + c.arena_off;
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float p = c.param[i];
    float gi = g[i] * mult + wd * p;
    float b = first_step ? gi : momentum * c.mom[i] + gi;
    c.mom[i] = b;
    gi = gi + momentum * b;
    c.param[i] = p - lr * gi;
  }
}

}  // namespace nnz

extern "C" int nnz_sgd_chunk_bytes(void) { return (int)sizeof(nnz::SgdChunk); }

extern "C" int nnz_sgd_chunk_fill(void* out_host, float* param, float* momentum, long arena_offset, int n) {
  using namespace nnz;
  if (!out_host || !param || !momentum || n < 1 || arena_offset < 0) return NNZ_EINVAL;
  SgdChunk c = {};
  c.param = param; c.mom = momentum; c.arena_off = arena_offset; c.n = n;
  memcpy(out_host, &c, sizeof(c));
  return NNZ_OK;
}

extern "C" int nnz_grad_sumsq_nonfinite(const float* grads, long n, float* out2_zeroed, void* stream) {
  using namespace nnz;
  if (!grads || !out2_zeroed || n < 1 || ((size_t)grads & 15)) return NNZ_EINVAL;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  NNZ_LAUNCH(grad_sumsq_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grads, n, out2_zeroed,
             (FxAcc*)nullptr, (unsigned*)nullptr);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// deterministic variant: `acc` = 2 zeroed fixed-point records (nnz_fxacc_bytes() each), `counter` = one zeroed 32-bit word,
// both left zero; out2 = {sum of squares, number of non-finite entries} is WRITTEN (needs no zero fill)
extern "C" int nnz_grad_sumsq_nonfinite_det(const float* grads, long n, float* out2, void* acc, void* counter,
                                            void* stream) {
  using namespace nnz;
  if (!grads || !out2 || !acc || !counter || n < 1 || ((size_t)grads & 15)) return NNZ_EINVAL;
  // every workgroup ends in two fixed-point adds on the SAME two records (~43 ns each, serialised per address, FX_REP
  // replicas): 2048 workgroups spent 86 us there for a 125 MB read.  512 workgroups x 4 loads in flight per lane.
  long blocks = (n / 16 + 255) / 256;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  NNZ_LAUNCH(grad_sumsq_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grads, n, out2, (FxAcc*)acc,
             (unsigned*)counter);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_sgd_nesterov_fused(const void* chunks_device, int nchunks, const float* arena, float* stats2,
                                      const float* inv_scale_device, float max_norm, float lr, float momentum,
                                      float weight_decay, int first_step, void* stream) {
  using namespace nnz;
  if (!chunks_device || nchunks < 1 || !arena || !stats2) return NNZ_EINVAL;
  NNZ_LAUNCH(sgd_nesterov_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const SgdChunk*)chunks_device, arena, stats2, inv_scale_device, max_norm, lr, momentum, weight_decay,
                     first_step);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
