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

// ---- fused AdamW tail of the zoo trainers (round 4) ------------------------------------------------------------------------------
// The X^2-Net plugins step AdamW(lr 1e-4, wd 5e-2, eps 1e-5) behind GradScaler.unscale_ + clip_grad_norm_(12)
// (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerM2Net.py:58-65, nnUNetTrainer.py:1131-1139).  With 1 500 - 4 100
// parameter tensors torch's multi-tensor path issues ~260 launches per step, each behind 60-100 us of host-side list handling:
// 8-17 ms of GPU idle time per step in the M2Net / SwT2Net / SSND2Net traces (profiles/r03_*_graph_kernels.txt).  Here the
// gradients stay where autograd (or the replayed hipGraph) leaves them and a DEVICE TABLE of chunks {param, grad, exp_avg,
// exp_avg_sq, n} - built once, the addresses are static under graph replay - drives two launches:
//   pass 1  sum of squares + non-finite count of all (still scaled) gradients, fixed-point across workgroups (deterministic);
//           its last workgroup also advances the per-parameter step counters when the step is going to be applied
//   pass 2  g = grad * inv_scale * clip;  p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//           p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (torch.optim.AdamW, amsgrad off)
// No host synchronisation: the inf / overflow skip is decided on the device like in the SGD tail above.
struct AdamChunk {
  float* param;
  const float* grad;
  float* m;
  float* v;
  int n;
  int owner;   // index of the parameter this chunk belongs to: its step counter is steps[owner]
};

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const AdamChunk* __restrict__ chunks, int nchunks,
                                                         float* __restrict__ out2, FxAcc* acc, unsigned* counter,
                                                         float* __restrict__ steps, int nsteps,
                                                         const float* __restrict__ inv_scale_dev) {
  __shared__ float red[4][2];
  // squares of the UNSCALED gradients (g / loss scale): with a gradient norm of ~3e4 (SSND2Net at initialisation) under a loss
  // scale >= 2048 a workgroup's partial sum of the scaled squares leaves the fixed-point accumulator's range (2^52) although
  // every fp32 gradient is finite - torch's unscale_ / clip / step would apply that step
  const float inv_scale = inv_scale_dev ? inv_scale_dev[0] : 1.f;
  float s = 0.f, bad = 0.f;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const AdamChunk ch = chunks[c];
    const bool vec = (((size_t)ch.grad) & 15) == 0;
    const int n4 = vec ? ch.n >> 2 : 0;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(ch.grad);
    for (int i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = g4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = v[e] * inv_scale;
        s += u * u;
        bad += (__builtin_isnan(v[e]) || __builtin_isinf(v[e])) ? 1.f : 0.f;
      }
    }
    for (int i = (n4 << 2) + threadIdx.x; i < ch.n; i += 256) {
      const float v = ch.grad[i];
      const float u = v * inv_scale;
      s += u * u;
      bad += (__builtin_isnan(v) || __builtin_isinf(v)) ? 1.f : 0.f;
    }
  }
  s = wave_sum(s);
  bad = wave_sum(bad);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = s;
    red[wave][1] = bad;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  if (threadIdx.x == 0) {
    // THREE records (round 6): {sum of squares, non-finite count, sum of squares x 2^-64}.  Record 0 resolves 3.5e-18 but a
    // workgroup partial >= 2^52 poisons it; the chaotic nets reach that with every gradient finite - SSND2NetP at 128^2 in an
    // fp32 step: gradient norm 2.6e9, sum of squares 7e18, where torch's clip_grad_norm_ (an fp32 sum) clips and steps
    // (tools/probes/ssnd2net_fp32_probe.py: the fused tail skipped all 24 steps).  Record 2 carries the same partials scaled
    // into range for every finite fp32 sum (resolution 2^6 in the unscaled sum - irrelevant where it is used, above 2^52) and
    // is read only when record 0 is poisoned.  Integer adds both: deterministic either way.
    const double part = (double)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
    fx_add(acc, 0, 3, blockIdx.x, part);
    fx_add(acc, 1, 3, blockIdx.x, (double)((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])));
    fx_add(acc, 2, 3, blockIdx.x, part * 0x1p-64);
  }
  if (last_workgroup_wave(counter, gridDim.x)) {
    float ss = 0.f, nb = 0.f;
    if (threadIdx.x == 0) {
      double t[3];
      fx_take_n<3>(acc, 0, 3, t);
      ss = (float)(t[0] == t[0] ? t[0] : t[2] * 0x1p64);
      nb = (float)t[1];
      const bool overflow = !(ss == ss) || __builtin_isinf(ss);
      if (overflow) nb = nb > 0.f ? nb : 1.f;   // finite gradients whose squares overflowed: skipped as well
      out2[0] = ss;
      out2[1] = nb;
    }
    nb = __shfl(nb, 0, 64);
    if (!(nb > 0.f))
      for (int i = threadIdx.x; i < nsteps; i += 64) steps[i] += 1.f;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk* __restrict__ chunks, const float* __restrict__ stats2,
                                                    const float* __restrict__ inv_scale_dev, float max_norm, float lr,
                                                    float b1, float b2, float omb1, float omb2, float eps, float wd,
                                                    const float* __restrict__ steps) {
  if (stats2[1] > 0.f) return;   // GradScaler semantics: parameters and moments untouched
  const float inv_scale = inv_scale_dev ? inv_scale_dev[0] : 1.f;
  const float total_norm = sqrtf(stats2[0]);      // pass 1 summed the squares of the unscaled gradients
  float clip = max_norm > 0.f ? max_norm / (total_norm + 1e-6f) : 1.f;
  clip = clip > 1.f ? 1.f : clip;
  const float mult = inv_scale * clip;
  // bias corrections in double, once per workgroup: 1 - 0.999^t in fp32 (fast-math powf) loses four digits at small t
  // The step counter is the chunk owner's own (torch keeps one per parameter): after a change of the member set - the encoder
  // of the Swin-UMamba plugins unfrozen at epoch 10, a checkpoint with heterogeneous counters - parameters at step 1 sit beside
  // parameters at step 2 500, and a shared t would give the new ones a first update 1 / (1 - beta1^2500) / (1 - beta1) too small
  // or the old ones one 10 x too large.
  const AdamChunk c = chunks[blockIdx.x];
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const double t = (double)steps[c.owner];      // already advanced by pass 1
    bc[0] = (float)(1.0 - pow((double)b1, t));
    bc[1] = (float)sqrt(1.0 - pow((double)b2, t));
  }
  __syncthreads();
  const float bc1 = bc[0], bc2s = bc[1];
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float g = c.grad[i] * mult;
    const float p = c.param[i] * decay;
    const float m = b1 * c.m[i] + omb1 * g;      // omb = 1 - beta formed in double by the host like torch does (1.f - 0.999f is
    const float v = b2 * c.v[i] + omb2 * g * g;  // off by 1.3e-5 relative: visible in exp_avg_sq after a few steps)
    c.m[i] = m;
    c.v[i] = v;
    c.param[i] = p - step_size * (m / (sqrtf(v) / bc2s + eps));
  }
}

}  // namespace nnz

extern "C" int nnz_adam_chunk_bytes(void) { return (int)sizeof(nnz::AdamChunk); }

// stats2 (2 floats) is written: {sum of squares of the UNSCALED gradients (g * inv_scale), > 0 if the step is skipped}; acc / counter: 3 zeroed
// fixed-point records + one zeroed word (left zero); steps: the parameters' step counters (fp32, one per parameter, advanced
// here when the step is applied); inv_scale_device: 1 / loss scale (1 float on the device) or NULL; max_norm <= 0: no clipping
extern "C" int nnz_adamw_fused(const void* chunks_device, int nchunks, float* stats2, void* acc, void* counter,
                               const float* inv_scale_device, float max_norm, float lr, double beta1d, double beta2d, float eps,
                               float weight_decay, float* steps, int nsteps, void* stream) {
  using namespace nnz;
  const float beta1 = (float)beta1d, beta2 = (float)beta2d;
  if (!chunks_device || nchunks < 1 || !stats2 || !acc || !counter || !steps || nsteps < 1) return NNZ_EINVAL;
  const int blocks = nchunks < 512 ? nchunks : 512;
  NNZ_LAUNCH(adam_sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunks_device, nchunks,
             stats2, (FxAcc*)acc, (unsigned*)counter, steps, nsteps, inv_scale_device);
  NNZ_LAUNCH(adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunks_device,
             (const float*)stats2, inv_scale_device, max_norm, lr, beta1, beta2, (float)(1.0 - (double)beta1d),
             (float)(1.0 - (double)beta2d), eps, weight_decay, (const float*)steps);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

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
