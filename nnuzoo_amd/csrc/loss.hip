// Fused soft-Dice + cross-entropy statistics over NC(D)HW logits (gfx950), forward and backward.
// Replaces, for one deep-supervision output, the chain softmax -> one-hot scatter -> 3 reductions (Dice) plus
// log_softmax + nll (CE) that the reference runs as separate full-resolution torch passes:
//   DC_and_CE_loss.forward               /root/reference/nnunetv2/training/loss/compound_losses.py:31-56
//   MemoryEfficientSoftDiceLoss.forward  /root/reference/nnunetv2/training/loss/dice.py:72-119
//   RobustCrossEntropyLoss.forward       /root/reference/nnunetv2/training/loss/robust_ce_loss.py:12-16
// HBM-bound: forward = one read of logits + target; backward = one read + one write.  All math in fp32 from the
// fp16 (autocast) or fp32 logits, exactly the dtype flow of the reference (softmax / CE run in fp32 under autocast).
//   sums[b] = { intersect[c] (C), sum_pred[c] (C), sum_gt[c] (C), ce_sum (1) }
//   coef[b] = { dL/d intersect[c] (C), dL/d sum_pred[c] (C), dL/d ce_sum (1) }
#include "common.hpp"

namespace nnz {

constexpr int LS_MAXC_BIG = 32;  // class-count buckets: <= 8 (the tuned instantiation) and <= 32 (same code, larger register arrays)

template <typename T>
struct LossArgs {
  const T* logits;      // [B][C][V]
  const int16_t* tgt;   // [B][V]
  float* sums;          // [B][3C+1]
  const float* coef;    // [B][2C+1]
  const float* gmul;    // optional device scalar multiplying every coefficient (upstream gradient of the loss)
  T* dlogits;           // [B][C][V]
  int B, C;
  long V;
  int vpb;
  int ignore;           // label value excluded from every sum and gradient (DC_and_CE_loss ignore_label), or -32768
  FxAcc* acc;           // optional [B][3C+1] fixed-point accumulators + launch counter (common.hpp): deterministic sums
  unsigned* counter;
};

template <typename T, int LS_MAXC>
__device__ __forceinline__ void softmax_at(const LossArgs<T>& a, long base, long v, float (&p)[LS_MAXC], float& lse) {
  float z[LS_MAXC];
  float m = -3.0e38f;
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < a.C) {
      z[c] = (float)a.logits[base + (long)c * a.V + v];
      m = fmaxf(m, z[c]);
    }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < a.C) {
      p[c] = __expf(z[c] - m);
      s += p[c];
    }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < a.C) p[c] *= inv;
  lse = m + __logf(s);
}

template <typename T, int LS_MAXC>
__global__ __launch_bounds__(256) void dc_ce_fwd_kernel(LossArgs<T> a) {
  __shared__ float lred[3 * LS_MAXC + 1];
  __shared__ float lwave[4][3 * LS_MAXC + 1];  // deterministic mode: one slot per wave, folded in wave order
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if (tid < 3 * LS_MAXC + 1) lred[tid] = 0.f;
  __syncthreads();
  const long v0 = (long)blockIdx.x * a.vpb;
  long v1 = v0 + a.vpb;
  if (v1 > a.V) v1 = a.V;
  const long base = (long)b * a.C * a.V;
  float inter[LS_MAXC], sp[LS_MAXC], sg[LS_MAXC], ce = 0.f;
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c) inter[c] = sp[c] = sg[c] = 0.f;
  for (long v = v0 + tid; v < v1; v += 256) {
    const int t = a.tgt[(long)b * a.V + v];
    if (t == a.ignore) continue;  // loss_mask of the Dice sums and ignore_index of the CE (compound_losses.py:38-52)
    float p[LS_MAXC], lse;
    softmax_at<T, LS_MAXC>(a, base, v, p, lse);
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c)
      if (c < a.C) {
        sp[c] += p[c];
        if (c == t) {
          inter[c] += p[c];
          sg[c] += 1.f;
          ce += lse - (float)a.logits[base + (long)c * a.V + v];
        }
      }
  }
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < a.C) {
      const float x0 = wave_sum(inter[c]), x1 = wave_sum(sp[c]), x2 = wave_sum(sg[c]);
      if ((tid & 63) == 0) {
        if (a.acc) {
          lwave[tid >> 6][c] = x0;
          lwave[tid >> 6][a.C + c] = x1;
          lwave[tid >> 6][2 * a.C + c] = x2;
        } else {
          atomicAdd(&lred[c], x0);
          atomicAdd(&lred[a.C + c], x1);
          atomicAdd(&lred[2 * a.C + c], x2);
        }
      }
    }
  const float xc = wave_sum(ce);
  if ((tid & 63) == 0) {
    if (a.acc)
      lwave[tid >> 6][3 * a.C] = xc;
    else
      atomicAdd(&lred[3 * a.C], xc);
  }
  __syncthreads();
  const int S = 3 * a.C + 1;
  if (a.acc) {
    if (tid < S)
      fx_add(a.acc, (long)b * S + tid, (long)a.B * S, blockIdx.x,
             (double)((lwave[0][tid] + lwave[1][tid]) + (lwave[2][tid] + lwave[3][tid])));
    if (last_workgroup(a.counter, gridDim.x * gridDim.y))
      for (int i = tid; i < a.B * S; i += 256) a.sums[i] = (float)fx_take(a.acc, i, (long)a.B * S);
    return;
  }
  if (tid < S) atomicAdd(a.sums + (long)b * S + tid, lred[tid]);
}

template <typename T, int LS_MAXC>
__global__ __launch_bounds__(256) void dc_ce_bwd_kernel(LossArgs<T> a) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const long v0 = (long)blockIdx.x * a.vpb;
  long v1 = v0 + a.vpb;
  if (v1 > a.V) v1 = a.V;
  const long base = (long)b * a.C * a.V;
  float gi[LS_MAXC], gp[LS_MAXC];
  const float gm = a.gmul ? a.gmul[0] : 1.f;
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c) {
    gi[c] = c < a.C ? gm * a.coef[(long)b * (2 * a.C + 1) + c] : 0.f;
    gp[c] = c < a.C ? gm * a.coef[(long)b * (2 * a.C + 1) + a.C + c] : 0.f;
  }
  const float gce = gm * a.coef[(long)b * (2 * a.C + 1) + 2 * a.C];
  for (long v = v0 + tid; v < v1; v += 256) {
    const int t = a.tgt[(long)b * a.V + v];
    if (t == a.ignore) {
#pragma unroll
      for (int c = 0; c < LS_MAXC; ++c)
        if (c < a.C) a.dlogits[base + (long)c * a.V + v] = (T)0.f;
      continue;
    }
    float p[LS_MAXC], lse;
    softmax_at<T, LS_MAXC>(a, base, v, p, lse);
    float S = 0.f;
    float ac[LS_MAXC];
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c)
      if (c < a.C) {
        ac[c] = gp[c] + (c == t ? gi[c] : 0.f);
        S += ac[c] * p[c];
      }
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c)
      if (c < a.C) {
        const float d = p[c] * (ac[c] - S) + gce * (p[c] - (c == t ? 1.f : 0.f));
        a.dlogits[base + (long)c * a.V + v] = (T)d;
      }
  }
}

template <typename T>
static int launch_loss(LossArgs<T> a, bool bwd, hipStream_t s) {
  if (a.C < 1 || a.C > LS_MAXC_BIG || a.B < 1 || a.V < 1) return NNZ_EINVAL;
  const bool big = a.C > 8;
  long vpb = (a.V * a.B + 2047) / 2048;
  if (vpb < 1024) vpb = 1024;
  if (vpb > a.V) vpb = a.V;
  a.vpb = (int)vpb;
  const int gx = (int)((a.V + vpb - 1) / vpb);
  if (!bwd) {
    if (!a.acc) {
      hipError_t e = nnz::zero_async(a.sums, sizeof(float) * a.B * (3 * a.C + 1), s);
      if (e != hipSuccess) return (int)e;
    }
    if (big)
      NNZ_LAUNCH((dc_ce_fwd_kernel<T, LS_MAXC_BIG>), dim3(gx, a.B), dim3(256), 0, s, a);
    else
      NNZ_LAUNCH((dc_ce_fwd_kernel<T, 8>), dim3(gx, a.B), dim3(256), 0, s, a);
  } else {
    if (big)
      NNZ_LAUNCH((dc_ce_bwd_kernel<T, LS_MAXC_BIG>), dim3(gx, a.B), dim3(256), 0, s, a);
    else
      NNZ_LAUNCH((dc_ce_bwd_kernel<T, 8>), dim3(gx, a.B), dim3(256), 0, s, a);
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// Online-Dice statistics of the validation step: argmax over the class axis (first maximum on ties, like
// torch.argmax) compared with the label map; one read of logits + target instead of argmax -> zeros -> scatter_ ->
// three products -> three reductions (nnUNetTrainer.validation_step, nnUNetTrainer.py:1185-1226 with
// get_tp_fp_fn_tn, training/loss/dice.py:122-180).  counts[c] = {tp, fp, fn} as exact integers.
template <typename T, int LS_MAXC>
__global__ __launch_bounds__(256) void argmax_stats_kernel(const T* __restrict__ logits, const int16_t* __restrict__ tgt,
                                                           unsigned long long* __restrict__ counts, int C, long V,
                                                           int vpb, int ignore) {
  __shared__ unsigned int lc[3 * LS_MAXC];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if (tid < 3 * LS_MAXC) lc[tid] = 0u;
  __syncthreads();
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  unsigned int tp[LS_MAXC], fp[LS_MAXC], fn[LS_MAXC];
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c) tp[c] = fp[c] = fn[c] = 0u;
  const long base = (long)b * C * V;
  for (long v = v0 + tid; v < v1; v += 256) {
    float best = (float)logits[base + v];
    int arg = 0;
#pragma unroll
    for (int c = 1; c < LS_MAXC; ++c)
      if (c < C) {
        const float z = (float)logits[base + (long)c * V + v];
        if (z > best) {
          best = z;
          arg = c;
        }
      }
    const int t = tgt[(long)b * V + v];
    if (t == ignore) continue;  // validation_step's mask (nnUNetTrainer.py:1209-1216): ignored voxels count nowhere
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c) {
      tp[c] += (arg == c) & (t == c);
      fp[c] += (arg == c) & (t != c);
      fn[c] += (arg != c) & (t == c);
    }
  }
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < C) {
      const unsigned int a0 = wave_sum_u32(tp[c]), a1 = wave_sum_u32(fp[c]), a2 = wave_sum_u32(fn[c]);
      if ((tid & 63) == 0) {
        atomicAdd(&lc[c * 3 + 0], a0);
        atomicAdd(&lc[c * 3 + 1], a1);
        atomicAdd(&lc[c * 3 + 2], a2);
      }
    }
  __syncthreads();
  if (tid < 3 * C && lc[tid]) atomicAdd(counts + tid, (unsigned long long)lc[tid]);
}

// ---- loss value + gradient coefficients from the sums, one tiny launch per deep-supervision output ------------------
// Replaces the ~25 element-wise torch launches per output that turn {intersect, sum_pred, sum_gt, ce_sum} into
// DC_and_CE_loss's value (dice.py:105-119: dc = (2I + s) / clip(G + P + s, 1e-8), loss = -mean(dc); CE mean) and, through
// autograd, into the coefficients the backward kernel needs.  loss_accum[0] += ds_weight * (w_ce * ce + w_dice * dice);
// coef[b] = ds_weight * {dL/dI[C], dL/dP[C], dL/dce_sum} (classes excluded by do_bg = false get 0).
struct FinalizeArgs {
  const float* sums;  // [B][3C+1]
  float* loss_accum;  // [1]
  float* coef;        // [B][2C+1]
  int B, C;
  long V;
  int batch_dice, do_bg, use_valid_count;
  float smooth, w_ce, w_dice, ds_weight;
};

__global__ __launch_bounds__(64) void dc_ce_finalize_kernel(FinalizeArgs a) {
  const int lane = threadIdx.x;
  const int C = a.C, S = 3 * C + 1, c0 = a.do_bg ? 0 : 1, KC = C - c0;
  float dice_sum = 0.f;
  if (a.batch_dice) {
    for (int c = c0 + lane; c < C; c += 64) {
      float I = 0.f, P = 0.f, G = 0.f;
      for (int b = 0; b < a.B; ++b) {
        I += a.sums[b * S + c];
        P += a.sums[b * S + C + c];
        G += a.sums[b * S + 2 * C + c];
      }
      const float raw = G + P + a.smooth, den = fmaxf(raw, 1e-8f);
      dice_sum += (2.f * I + a.smooth) / den;
      const float k = a.ds_weight * a.w_dice / (float)KC;
      const float dI = -2.f / den * k;
      const float dP = raw > 1e-8f ? (2.f * I + a.smooth) / (den * den) * k : 0.f;
      for (int b = 0; b < a.B; ++b) {
        a.coef[b * (2 * C + 1) + c] = dI;
        a.coef[b * (2 * C + 1) + C + c] = dP;
      }
    }
  } else {
    const int n = a.B * KC;
    for (int i = lane; i < n; i += 64) {
      const int b = i / KC, c = c0 + i % KC;
      const float I = a.sums[b * S + c], P = a.sums[b * S + C + c], G = a.sums[b * S + 2 * C + c];
      const float raw = G + P + a.smooth, den = fmaxf(raw, 1e-8f);
      dice_sum += (2.f * I + a.smooth) / den;
      const float k = a.ds_weight * a.w_dice / (float)n;
      a.coef[b * (2 * C + 1) + c] = -2.f / den * k;
      a.coef[b * (2 * C + 1) + C + c] = raw > 1e-8f ? (2.f * I + a.smooth) / (den * den) * k : 0.f;
    }
  }
  if (!a.do_bg)
    for (int b = lane; b < a.B; b += 64) a.coef[b * (2 * C + 1) + 0] = a.coef[b * (2 * C + 1) + C] = 0.f;
  dice_sum = wave_sum(dice_sum);
  float ce = 0.f, valid = 0.f;
  for (int b = lane; b < a.B; b += 64) {
    ce += a.sums[b * S + 3 * C];
    for (int c = 0; c < C; ++c) valid += a.sums[b * S + 2 * C + c];
  }
  ce = wave_sum(ce);
  valid = wave_sum(valid);
  const float nvox = a.use_valid_count ? fmaxf(valid, 1.f) : (float)a.B * (float)a.V;
  for (int b = lane; b < a.B; b += 64) a.coef[b * (2 * C + 1) + 2 * C] = a.ds_weight * a.w_ce / nvox;
  if (lane == 0) {
    const float dice = -dice_sum / (float)(a.batch_dice ? KC : a.B * KC);
    atomicAdd(a.loss_accum, a.ds_weight * (a.w_ce * ce / nvox + a.w_dice * dice));
  }
}

// ---- region-based training: sigmoid Dice + BCE-with-logits over one-hot region targets ------------------------------
// DC_and_BCE_loss.forward (/root/reference/nnunetv2/training/loss/compound_losses.py:59-109) with
// MemoryEfficientSoftDiceLoss(apply_nonlin=torch.sigmoid, do_bg=True) (dice.py:72-119): per (b, region c)
//   intersect = sum p*y*m, sum_pred = sum p*m, sum_gt = sum y*m, p = sigmoid(z), m = 1 - target[:, -1] (ignore channel)
// and bce_sum = sum_{c,v} m * (max(z,0) - z*y + log(1 + exp(-|z|))), mask_sum = sum_v m.
//   sums[b] = { intersect[C], sum_pred[C], sum_gt[C], bce_sum, mask_sum };   coef[b] = { dL/dintersect[C], dL/dsum_pred[C], dL/dbce_sum }
template <typename T>
struct RegionArgs {
  const T* logits;        // [B][C][V]
  const int16_t* tgt;     // [B][Ct][V], Ct = C (+1: last channel = ignore mask) values 0 / 1
  float* sums;            // [B][3C+2]
  const float* coef;      // [B][2C+1]
  T* dlogits;
  int B, C, Ct;
  long V;
  int vpb;
};

template <typename T, int LS_MAXC>
__global__ __launch_bounds__(256) void dc_bce_fwd_kernel(RegionArgs<T> a) {
  __shared__ float lred[3 * LS_MAXC + 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if (tid < 3 * LS_MAXC + 2) lred[tid] = 0.f;
  __syncthreads();
  const long v0 = (long)blockIdx.x * a.vpb;
  long v1 = v0 + a.vpb;
  if (v1 > a.V) v1 = a.V;
  const long base = (long)b * a.C * a.V, tbase = (long)b * a.Ct * a.V;
  float inter[LS_MAXC], sp[LS_MAXC], sg[LS_MAXC], bce = 0.f, ms = 0.f;
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c) inter[c] = sp[c] = sg[c] = 0.f;
  for (long v = v0 + tid; v < v1; v += 256) {
    const float m = a.Ct > a.C ? 1.f - (float)a.tgt[tbase + (long)a.C * a.V + v] : 1.f;
    ms += m;
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c)
      if (c < a.C) {
        const float z = (float)a.logits[base + (long)c * a.V + v];
        const float y = (float)a.tgt[tbase + (long)c * a.V + v];
        const float e = __expf(-fabsf(z));
        const float p = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        inter[c] += p * y * m;
        sp[c] += p * m;
        sg[c] += y * m;
        bce += m * (fmaxf(z, 0.f) - z * y + __logf(1.f + e));
      }
  }
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < a.C) {
      const float x0 = wave_sum(inter[c]), x1 = wave_sum(sp[c]), x2 = wave_sum(sg[c]);
      if ((tid & 63) == 0) {
        atomicAdd(&lred[c], x0);
        atomicAdd(&lred[a.C + c], x1);
        atomicAdd(&lred[2 * a.C + c], x2);
      }
    }
  const float xb = wave_sum(bce), xm = wave_sum(ms);
  if ((tid & 63) == 0) {
    atomicAdd(&lred[3 * a.C], xb);
    atomicAdd(&lred[3 * a.C + 1], xm);
  }
  __syncthreads();
  if (tid < 3 * a.C + 2) atomicAdd(a.sums + (long)b * (3 * a.C + 2) + tid, lred[tid]);
}

template <typename T, int LS_MAXC>
__global__ __launch_bounds__(256) void dc_bce_bwd_kernel(RegionArgs<T> a) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const long v0 = (long)blockIdx.x * a.vpb;
  long v1 = v0 + a.vpb;
  if (v1 > a.V) v1 = a.V;
  const long base = (long)b * a.C * a.V, tbase = (long)b * a.Ct * a.V;
  float gi[LS_MAXC], gp[LS_MAXC];
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c) {
    gi[c] = c < a.C ? a.coef[(long)b * (2 * a.C + 1) + c] : 0.f;
    gp[c] = c < a.C ? a.coef[(long)b * (2 * a.C + 1) + a.C + c] : 0.f;
  }
  const float gb = a.coef[(long)b * (2 * a.C + 1) + 2 * a.C];
  for (long v = v0 + tid; v < v1; v += 256) {
    const float m = a.Ct > a.C ? 1.f - (float)a.tgt[tbase + (long)a.C * a.V + v] : 1.f;
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c)
      if (c < a.C) {
        const float z = (float)a.logits[base + (long)c * a.V + v];
        const float y = (float)a.tgt[tbase + (long)c * a.V + v];
        const float e = __expf(-fabsf(z));
        const float p = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        const float dp = p * (1.f - p);
        a.dlogits[base + (long)c * a.V + v] = (T)(m * ((gi[c] * y + gp[c]) * dp + gb * (p - y)));
      }
  }
}

// validation statistics of region training: prediction = sigmoid(z) > 0.5 (z > 0) per region channel, masked
template <typename T, int LS_MAXC>
__global__ __launch_bounds__(256) void region_stats_kernel(const T* __restrict__ logits, const int16_t* __restrict__ tgt,
                                                           unsigned long long* __restrict__ counts, int C, int Ct, long V,
                                                           int vpb) {
  __shared__ unsigned int lc[3 * LS_MAXC];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if (tid < 3 * LS_MAXC) lc[tid] = 0u;
  __syncthreads();
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  unsigned int tp[LS_MAXC], fp[LS_MAXC], fn[LS_MAXC];
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c) tp[c] = fp[c] = fn[c] = 0u;
  const long base = (long)b * C * V, tbase = (long)b * Ct * V;
  for (long v = v0 + tid; v < v1; v += 256) {
    if (Ct > C && tgt[tbase + (long)C * V + v] != 0) continue;
#pragma unroll
    for (int c = 0; c < LS_MAXC; ++c)
      if (c < C) {
        const bool pr = (float)logits[base + (long)c * V + v] > 0.f;
        const bool y = tgt[tbase + (long)c * V + v] != 0;
        tp[c] += pr & y;
        fp[c] += pr & !y;
        fn[c] += !pr & y;
      }
  }
#pragma unroll
  for (int c = 0; c < LS_MAXC; ++c)
    if (c < C) {
      const unsigned int a0 = wave_sum_u32(tp[c]), a1 = wave_sum_u32(fp[c]), a2 = wave_sum_u32(fn[c]);
      if ((tid & 63) == 0) {
        atomicAdd(&lc[c * 3 + 0], a0);
        atomicAdd(&lc[c * 3 + 1], a1);
        atomicAdd(&lc[c * 3 + 2], a2);
      }
    }
  __syncthreads();
  if (tid < 3 * C && lc[tid]) atomicAdd(counts + tid, (unsigned long long)lc[tid]);
}

template <typename T>
static int launch_region(RegionArgs<T> a, int mode, void* counts, hipStream_t s) {
  if (a.C < 1 || a.C > LS_MAXC_BIG || a.B < 1 || a.V < 1 || (a.Ct != a.C && a.Ct != a.C + 1)) return NNZ_EINVAL;
  long vpb = (a.V * a.B + 2047) / 2048;
  if (vpb < 1024) vpb = 1024;
  if (vpb > a.V) vpb = a.V;
  a.vpb = (int)vpb;
  const int gx = (int)((a.V + vpb - 1) / vpb);
  const bool big = a.C > 8;
  if (mode == 0) {
    hipError_t e = nnz::zero_async(a.sums, sizeof(float) * a.B * (3 * a.C + 2), s);
    if (e != hipSuccess) return (int)e;
    if (big) NNZ_LAUNCH((dc_bce_fwd_kernel<T, LS_MAXC_BIG>), dim3(gx, a.B), dim3(256), 0, s, a);
    else NNZ_LAUNCH((dc_bce_fwd_kernel<T, 8>), dim3(gx, a.B), dim3(256), 0, s, a);
  } else if (mode == 1) {
    if (big) NNZ_LAUNCH((dc_bce_bwd_kernel<T, LS_MAXC_BIG>), dim3(gx, a.B), dim3(256), 0, s, a);
    else NNZ_LAUNCH((dc_bce_bwd_kernel<T, 8>), dim3(gx, a.B), dim3(256), 0, s, a);
  } else {
    hipError_t e = nnz::zero_async(counts, sizeof(unsigned long long) * 3 * a.C, s);
    if (e != hipSuccess) return (int)e;
    if (big)
      NNZ_LAUNCH((region_stats_kernel<T, LS_MAXC_BIG>), dim3(gx, a.B), dim3(256), 0, s, a.logits, a.tgt,
                         (unsigned long long*)counts, a.C, a.Ct, a.V, (int)vpb);
    else
      NNZ_LAUNCH((region_stats_kernel<T, 8>), dim3(gx, a.B), dim3(256), 0, s, a.logits, a.tgt,
                         (unsigned long long*)counts, a.C, a.Ct, a.V, (int)vpb);
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_dc_bce_loss_forward(const void* logits, int logits_is_f16, const int16_t* target_regions, float* sums,
                                       int B, int C, int Ct, long V, void* stream) {
  using namespace nnz;
  if (!logits || !target_regions || !sums) return NNZ_EINVAL;
  if (logits_is_f16) {
    RegionArgs<f16> a = {};
    a.logits = (const f16*)logits; a.tgt = target_regions; a.sums = sums; a.B = B; a.C = C; a.Ct = Ct; a.V = V;
    return launch_region(a, 0, nullptr, (hipStream_t)stream);
  }
  RegionArgs<float> a = {};
  a.logits = (const float*)logits; a.tgt = target_regions; a.sums = sums; a.B = B; a.C = C; a.Ct = Ct; a.V = V;
  return launch_region(a, 0, nullptr, (hipStream_t)stream);
}

extern "C" int nnz_dc_bce_loss_backward(const void* logits, int logits_is_f16, const int16_t* target_regions,
                                        const float* coef, void* dlogits, int B, int C, int Ct, long V, void* stream) {
  using namespace nnz;
  if (!logits || !target_regions || !coef || !dlogits) return NNZ_EINVAL;
  if (logits_is_f16) {
    RegionArgs<f16> a = {};
    a.logits = (const f16*)logits; a.tgt = target_regions; a.coef = coef; a.dlogits = (f16*)dlogits;
    a.B = B; a.C = C; a.Ct = Ct; a.V = V;
    return launch_region(a, 1, nullptr, (hipStream_t)stream);
  }
  RegionArgs<float> a = {};
  a.logits = (const float*)logits; a.tgt = target_regions; a.coef = coef; a.dlogits = (float*)dlogits;
  a.B = B; a.C = C; a.Ct = Ct; a.V = V;
  return launch_region(a, 1, nullptr, (hipStream_t)stream);
}

extern "C" int nnz_region_tp_fp_fn(const void* logits, int logits_is_f16, const int16_t* target_regions, void* counts_u64,
                                   int B, int C, int Ct, long V, void* stream) {
  using namespace nnz;
  if (!logits || !target_regions || !counts_u64) return NNZ_EINVAL;
  if (logits_is_f16) {
    RegionArgs<f16> a = {};
    a.logits = (const f16*)logits; a.tgt = target_regions; a.B = B; a.C = C; a.Ct = Ct; a.V = V;
    return launch_region(a, 2, counts_u64, (hipStream_t)stream);
  }
  RegionArgs<float> a = {};
  a.logits = (const float*)logits; a.tgt = target_regions; a.B = B; a.C = C; a.Ct = Ct; a.V = V;
  return launch_region(a, 2, counts_u64, (hipStream_t)stream);
}

extern "C" int nnz_argmax_tp_fp_fn(const void* logits, int logits_is_f16, const int16_t* target, void* counts_u64,
                                   int B, int C, long V, int ignore_label, void* stream) {
  using namespace nnz;
  if (!logits || !target || !counts_u64 || C < 1 || C > LS_MAXC_BIG || B < 1 || V < 1) return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = nnz::zero_async(counts_u64, sizeof(unsigned long long) * 3 * C, s);
  if (e != hipSuccess) return (int)e;
  long vpb = (V * B + 2047) / 2048;
  if (vpb < 2048) vpb = 2048;
  if (vpb > V) vpb = V;
  const int gx = (int)((V + vpb - 1) / vpb);
#define NNZ_ARGMAX(TT, MC)                                                                                       \
  NNZ_LAUNCH((argmax_stats_kernel<TT, MC>), dim3(gx, B), dim3(256), 0, s, (const TT*)logits, target, \
                     (unsigned long long*)counts_u64, C, V, (int)vpb, ignore_label)
  if (logits_is_f16) {
    if (C > 8) NNZ_ARGMAX(f16, LS_MAXC_BIG); else NNZ_ARGMAX(f16, 8);
  } else {
    if (C > 8) NNZ_ARGMAX(float, LS_MAXC_BIG); else NNZ_ARGMAX(float, 8);
  }
#undef NNZ_ARGMAX
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_dc_ce_loss_forward_det(const void* logits, int logits_is_f16, const int16_t* target, float* sums,
                                          int B, int C, long V, int ignore_label, void* acc, void* counter,
                                          void* stream);
extern "C" int nnz_dc_ce_loss_forward(const void* logits, int logits_is_f16, const int16_t* target, float* sums, int B,
                                      int C, long V, int ignore_label, void* stream) {
  return nnz_dc_ce_loss_forward_det(logits, logits_is_f16, target, sums, B, C, V, ignore_label, nullptr, nullptr, stream);
}

// acc / counter (both or neither): B * (3C + 1) zeroed fixed-point records (nnz_fxacc_bytes() each) + one zeroed 32-bit
// word, left zero by the launch: the sums are then bit-identical run to run (no float atomics, common.hpp)
extern "C" int nnz_dc_ce_loss_forward_det(const void* logits, int logits_is_f16, const int16_t* target, float* sums,
                                          int B, int C, long V, int ignore_label, void* acc, void* counter,
                                          void* stream) {
  using namespace nnz;
  if (!logits || !target || !sums || (!acc != !counter)) return NNZ_EINVAL;
  if (logits_is_f16) {
    LossArgs<f16> a = {};
    a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
    a.logits = (const f16*)logits; a.tgt = target; a.sums = sums; a.B = B; a.C = C; a.V = V; a.ignore = ignore_label;
    return launch_loss(a, false, (hipStream_t)stream);
  }
  LossArgs<float> a = {};
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
  a.logits = (const float*)logits; a.tgt = target; a.sums = sums; a.B = B; a.C = C; a.V = V; a.ignore = ignore_label;
  return launch_loss(a, false, (hipStream_t)stream);
}

extern "C" int nnz_dc_ce_loss_finalize(const float* sums, float* loss_accum, float* coef, int B, int C, long V,
                                       int batch_dice, int do_bg, float smooth, float weight_ce, float weight_dice,
                                       float ds_weight, int use_valid_count, void* stream) {
  using namespace nnz;
  if (!sums || !loss_accum || !coef || B < 1 || C < 1 || C > LS_MAXC_BIG || (!do_bg && C < 2)) return NNZ_EINVAL;
  FinalizeArgs a = {};
  a.sums = sums; a.loss_accum = loss_accum; a.coef = coef; a.B = B; a.C = C; a.V = V;
  a.batch_dice = batch_dice; a.do_bg = do_bg; a.use_valid_count = use_valid_count;
  a.smooth = smooth; a.w_ce = weight_ce; a.w_dice = weight_dice; a.ds_weight = ds_weight;
  NNZ_LAUNCH(dc_ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_dc_ce_loss_backward_scaled(const void* logits, int logits_is_f16, const int16_t* target,
                                              const float* coef, const float* gmul_device, void* dlogits, int B, int C,
                                              long V, int ignore_label, void* stream);

extern "C" int nnz_dc_ce_loss_backward(const void* logits, int logits_is_f16, const int16_t* target, const float* coef,
                                       void* dlogits, int B, int C, long V, int ignore_label, void* stream) {
  return nnz_dc_ce_loss_backward_scaled(logits, logits_is_f16, target, coef, nullptr, dlogits, B, C, V, ignore_label,
                                        stream);
}

extern "C" int nnz_dc_ce_loss_backward_scaled(const void* logits, int logits_is_f16, const int16_t* target,
                                              const float* coef, const float* gmul_device, void* dlogits, int B, int C,
                                              long V, int ignore_label, void* stream) {
  using namespace nnz;
  if (!logits || !target || !coef || !dlogits) return NNZ_EINVAL;
  if (logits_is_f16) {
    LossArgs<f16> a = {};
    a.logits = (const f16*)logits; a.tgt = target; a.coef = coef; a.gmul = gmul_device; a.dlogits = (f16*)dlogits; a.B = B; a.C = C; a.V = V; a.ignore = ignore_label;
    return launch_loss(a, true, (hipStream_t)stream);
  }
  LossArgs<float> a = {};
  a.logits = (const float*)logits; a.tgt = target; a.coef = coef; a.gmul = gmul_device; a.dlogits = (float*)dlogits; a.B = B; a.C = C; a.V = V; a.ignore = ignore_label;
  return launch_loss(a, true, (hipStream_t)stream);
}
