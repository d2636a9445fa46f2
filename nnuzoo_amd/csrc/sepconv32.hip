// The depthwise-separable conv -> BatchNorm -> ReLU unit of SwT2Net's RSU4F stages in fp32, channels-last (round 6).
//
// Reference: /root/reference/nnunetv2/nets/swt2net.py:17-31 `REBNCONV` = get_dwconv_layer (depthwise 3x3, then pointwise 1x1,
// both bias-free) -> nn.BatchNorm2d -> ReLU, eight of them per RSU4F (:873-905), three RSU4F stages in SwT2Net (512 / 1024
// channels on 32^2 and 16^2 maps at a 512^2 patch).  The reference's trainer runs them in fp32 (nnUNetTrainerSwT2Net.py:112-130, no
// autocast) through cuDNN; on this stack that is MIOpen: Winograd / implicit-GEMM solvers picked per process for the 1x1 (the
// SwT2Net step differed by 7 % from box to box on that choice alone, DESIGN 5 of round 5), NCHW <-> NHWC transposes around them,
// MIOpenBatchNorm{Fwd,Bwd}Spatial and ATen's depthwise kernels - ~140 library launches and 2.5 ms of a 56 ms step
// (profiles/r05_swt2net_graph_kernels.txt).
//
// Here the RSU4F keeps its activations token-major ([B, H, W, C] fp32, the layout the Swin stages around it already use):
//   * the pointwise 1x1 convolution IS a token Linear: csrc/dense32.hip (fp32 MFMA, its grouped weight gradients);
//   * the depthwise 3x3 and the BatchNorm + ReLU are the small kernels below.  With channels innermost a lane owns 4 channels of a
//     token (16-byte accesses, a wave covers 64 x 4 consecutive floats = whole lines); the maps are tiny (512 ... 2 048 tokens), so
//     these are latency-sized launches - the point is one launch per operator, no layout changes, no library.
// Everything reduces in a fixed order (no atomics): the unit is bit-reproducible from run to run.
#include "common.hpp"

namespace nnz {

struct DwArgs32 {
  const float* x;     // [B][H][W][C]
  const float* w;     // [C][3][3] (torch's [C, 1, 3, 3])
  const float* bias;  // [C] or null
  float* y;           // [B][H][W][C]
  int B, H, W, C;
};

// FLIP = false: y[t][c] = b[c] + sum_k w[c][k] x[t + off_k][c] (zero padding 1).  FLIP = true: the input gradient - the same
// sum with the kernel rotated by 180 degrees (and no bias).  Thread = one channel quad (fixed: its 36 weights sit in registers) x
// DW_TOK tokens, DW_TOK consecutive workgroup rows apart.  (First version: one token per thread with the weights fetched per
// token - 45 vector-memory instructions per output quad, 1.3 TB/s on the 512^2 x 32 stem.)
constexpr int DW_TOK = 4;
template <bool FLIP>
__global__ __launch_bounds__(256) void dw3x3_nhwc_kernel(DwArgs32 a, int cq, int tl) {
  // cq = channel quads per workgroup (<= 256), tl = 256 / cq token lanes
  const int C4 = a.C >> 2;
  const int ql = threadIdx.x % cq, tk = threadIdx.x / cq;
  const int q = blockIdx.y * cq + ql;
  if (q >= C4 || tk >= tl) return;
  const int c = q * 4;
  const long T = (long)a.B * a.H * a.W;
  float w[4][9];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 9; ++k) w[e][k] = a.w[(long)(c + e) * 9 + (FLIP ? 8 - k : k)];
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
  if (!FLIP && a.bias) b0 = *reinterpret_cast<const f32x4*>(a.bias + c);
#pragma unroll
  for (int j = 0; j < DW_TOK; ++j) {
    const long t = ((long)blockIdx.x * DW_TOK + j) * tl + tk;
    if (t >= T) break;
    const int xw = (int)(t % a.W);
    const int yh = (int)((t / a.W) % a.H);
    f32x4 acc = b0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = yh + ky - 1;
      if ((unsigned)yy >= (unsigned)a.H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = xw + kx - 1;
        if ((unsigned)xx >= (unsigned)a.W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + (t + (long)(ky - 1) * a.W + (kx - 1)) * a.C + c);
        const int k = ky * 3 + kx;
        acc[0] += w[0][k] * v[0];
        acc[1] += w[1][k] * v[1];
        acc[2] += w[2][k] * v[2];
        acc[3] += w[3][k] * v[3];
      }
    }
    *reinterpret_cast<f32x4*>(a.y + t * a.C + c) = acc;
  }
}

// Lane geometry of the reducing kernels below: a workgroup owns 16 consecutive channels - lane quad q = tid & 3 holds channels
// 4 q .. 4 q + 3 of a token as one 16-byte access - and 64 token groups tg = tid >> 2 (16 per wave).  Sums over tokens fold in
// three fixed stages: the lanes of a wave that share q (shuffles over lane bits 2 .. 5), the four waves through LDS, and - across
// workgroups - per-range partials summed by fold_partials().  (First version: 32 / 64 channels x 8 / 4 token groups, one 4-byte load
// per token and thread - 16 workgroups walking 64 tokens each for a 512 x 512 map: 96 - 190 us per launch where the bytes take 1.)
__device__ __forceinline__ float sep_wave_fold(float v) {
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// Weight gradient: dW[c][k] = sum_t dy[t][c] x[t + off_k][c].  Workgroup = 16 channels (blockIdx.x) x one range of 256 tokens
// (blockIdx.y): four tokens per thread; per-range partials [range][C][9] go to the workspace and fold_partials() sums the ranges.
__global__ __launch_bounds__(256) void dw3x3_nhwc_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ part, int B, int H, int W, int C,
                                                               long tokens_per_range) {
  __shared__ float red[4][16][9];
  const int q = threadIdx.x & 3, tg = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 16 + q * 4;
  const long T = (long)B * H * W;
  const long t0 = (long)blockIdx.y * tokens_per_range;
  const long t1 = t0 + tokens_per_range < T ? t0 + tokens_per_range : T;
  f32x4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (long t = t0 + tg; t < t1; t += 64) {
      const int xw = (int)(t % W);
      const int yh = (int)((t / W) % H);
      const f32x4 gy = *reinterpret_cast<const f32x4*>(dy + t * C + c);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = yh + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = xw + kx - 1;
          if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            acc[ky * 3 + kx] += gy * *reinterpret_cast<const f32x4*>(x + (t + (long)(ky - 1) * W + (kx - 1)) * C + c);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = sep_wave_fold(acc[k][e]);
      if ((threadIdx.x & 63) < 4) red[wave][q * 4 + e][k] = v;
    }
  __syncthreads();
  if (threadIdx.x < 144) {
    const int ch = threadIdx.x / 9, k = threadIdx.x - ch * 9;
    if (blockIdx.x * 16 + ch < C)
      part[((long)blockIdx.y * C + blockIdx.x * 16 + ch) * 9 + k] = (red[0][ch][k] + red[1][ch][k]) + (red[2][ch][k] + red[3][ch][k]);
  }
}

// ---- BatchNorm (training or running statistics) + ReLU on [T][C] -------------------------------------------------------------
// Workgroup = 16 channels x 64 token groups (geometry above), the whole token axis: these maps are 128 ... 2 048 tokens.  Training
// statistics are two-pass (mean, then squared deviations) like torch's (a sum of squares minus the squared mean loses the variance
// of a channel whose mean is large against its spread); the 16 channels x T tokens are re-read from L2.
struct BnArgs32 {
  const float* x;
  const float* dy;      // backward only
  const float* gamma;
  const float* beta;
  float* running_mean;  // forward, training: updated in place (momentum); eval: read
  float* running_var;
  float* mean;          // [C] saved statistics (forward writes, backward reads)
  float* rstd;
  float* y;             // forward output / backward dx
  float* dgamma;
  float* dbeta;
  long T;
  int C;
  int training;
  float momentum, eps;
};

// sum over all tokens of a per-thread f32x4 (the thread's four channels); every thread gets its channels' totals
__device__ __forceinline__ f32x4 bn_fold(float (*red)[16], int q, f32x4 v) {
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s = sep_wave_fold(v[e]);
    if ((threadIdx.x & 63) < 4) red[wave][q * 4 + e] = s;
  }
  __syncthreads();
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (red[0][q * 4 + e] + red[1][q * 4 + e]) + (red[2][q * 4 + e] + red[3][q * 4 + e]);
  __syncthreads();
  return o;
}

// RT = tokens a thread keeps in REGISTERS (T <= 64 RT): the thread's rows are loaded once - all loads in flight together, one
// memory round trip - and the mean, the squared deviations and the output come from registers.  RT = 0: the three passes re-read
// the rows (any T).  (The re-reading form was three to five DEPENDENT round trips per launch: ~20 us for a 512 x 512 map
// where MIOpen's kernel took 5.)
template <int RT>
__global__ __launch_bounds__(256) void bn_relu_nhwc_fwd_kernel(BnArgs32 a) {
  __shared__ float red[4][16];
  const int q = threadIdx.x & 3, tg = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + q * 4;
  const bool live = c < a.C;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 rows[RT > 0 ? RT : 1];
  if (RT > 0) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long t = tg + 64L * i;
      rows[i] = (live && t < a.T) ? *reinterpret_cast<const f32x4*>(a.x + t * a.C + c) : z;
    }
  }
  f32x4 mean = z, rstd = z;
  if (a.training) {
    f32x4 s = z;
    if (RT > 0) {
#pragma unroll
      for (int i = 0; i < RT; ++i) s += rows[i];
    } else if (live) {
      for (long t = tg; t < a.T; t += 64) s += *reinterpret_cast<const f32x4*>(a.x + t * a.C + c);
    }
    mean = bn_fold(red, q, s) * (1.f / (float)a.T);
    f32x4 d2 = z;
    if (RT > 0) {
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const f32x4 d = rows[i] - mean;
        if (tg + 64L * i < a.T) d2 += d * d;
      }
    } else if (live) {
      for (long t = tg; t < a.T; t += 64) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(a.x + t * a.C + c) - mean;
        d2 += d * d;
      }
    }
    const f32x4 var = bn_fold(red, q, d2) * (1.f / (float)a.T);
#pragma unroll
    for (int e = 0; e < 4; ++e) rstd[e] = 1.f / sqrtf(var[e] + a.eps);
    if (live && tg == 0) {
      const float k = a.T > 1 ? (float)a.T / (float)(a.T - 1) : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a.running_mean[c + e] = (1.f - a.momentum) * a.running_mean[c + e] + a.momentum * mean[e];
        a.running_var[c + e] = (1.f - a.momentum) * a.running_var[c + e] + a.momentum * var[e] * k;
      }
    }
  } else if (live) {
    mean = *reinterpret_cast<const f32x4*>(a.running_mean + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(a.running_var + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) rstd[e] = 1.f / sqrtf(rv[e] + a.eps);
  }
  if (!live) return;
  if (tg == 0) {
    *reinterpret_cast<f32x4*>(a.mean + c) = mean;
    *reinterpret_cast<f32x4*>(a.rstd + c) = rstd;
  }
  const f32x4 sc = rstd * *reinterpret_cast<const f32x4*>(a.gamma + c);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(a.beta + c) - mean * sc;
  if (RT > 0) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long t = tg + 64L * i;
      f32x4 v = rows[i] * sc + sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      if (t < a.T) *reinterpret_cast<f32x4*>(a.y + t * a.C + c) = v;
    }
  } else {
    for (long t = tg; t < a.T; t += 64) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.x + t * a.C + c) * sc + sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      *reinterpret_cast<f32x4*>(a.y + t * a.C + c) = v;
    }
  }
}

// dx = gamma rstd (g - mean(g) - xhat mean(g xhat)),  g = dy [pre > 0],  pre = xhat gamma + beta;  dgamma = sum g xhat,  dbeta = sum g
// RT as above: xhat and g of the thread's rows stay in registers between the reduction and the output pass
template <int RT>
__global__ __launch_bounds__(256) void bn_relu_nhwc_bwd_kernel(BnArgs32 a) {
  __shared__ float red[4][16];
  const int q = threadIdx.x & 3, tg = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + q * 4;
  const bool live = c < a.C;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 mean = live ? *reinterpret_cast<const f32x4*>(a.mean + c) : z, rstd = live ? *reinterpret_cast<const f32x4*>(a.rstd + c) : z;
  const f32x4 ga = live ? *reinterpret_cast<const f32x4*>(a.gamma + c) : z, be = live ? *reinterpret_cast<const f32x4*>(a.beta + c) : z;
  auto masked = [&](f32x4 xh, f32x4 gg) {
    const f32x4 pre = xh * ga + be;
#pragma unroll
    for (int e = 0; e < 4; ++e) gg[e] = pre[e] > 0.f ? gg[e] : 0.f;
    return gg;
  };
  f32x4 xh_r[RT > 0 ? RT : 1], g_r[RT > 0 ? RT : 1];
  f32x4 s1 = z, s2 = z;
  if (RT > 0) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long t = tg + 64L * i;
      const bool ok = live && t < a.T;
      xh_r[i] = ok ? *reinterpret_cast<const f32x4*>(a.x + t * a.C + c) : mean;
      g_r[i] = ok ? *reinterpret_cast<const f32x4*>(a.dy + t * a.C + c) : z;
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      xh_r[i] = (xh_r[i] - mean) * rstd;
      g_r[i] = masked(xh_r[i], g_r[i]);
      s1 += g_r[i];
      s2 += g_r[i] * xh_r[i];
    }
  } else if (live) {
    for (long t = tg; t < a.T; t += 64) {
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(a.x + t * a.C + c) - mean) * rstd;
      const f32x4 gg = masked(xh, *reinterpret_cast<const f32x4*>(a.dy + t * a.C + c));
      s1 += gg;
      s2 += gg * xh;
    }
  }
  const f32x4 sum_g = bn_fold(red, q, s1);
  const f32x4 sum_gx = bn_fold(red, q, s2);
  if (!live) return;
  if (tg == 0) {
    *reinterpret_cast<f32x4*>(a.dgamma + c) = sum_gx;
    *reinterpret_cast<f32x4*>(a.dbeta + c) = sum_g;
  }
  const float inv_t = 1.f / (float)a.T;
  const f32x4 m1 = sum_g * inv_t, m2 = sum_gx * inv_t, k = ga * rstd;
  if (RT > 0) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long t = tg + 64L * i;
      if (t < a.T) *reinterpret_cast<f32x4*>(a.y + t * a.C + c) = k * (g_r[i] - m1 - xh_r[i] * m2);
    }
  } else {
    for (long t = tg; t < a.T; t += 64) {
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(a.x + t * a.C + c) - mean) * rstd;
      const f32x4 gg = masked(xh, *reinterpret_cast<const f32x4*>(a.dy + t * a.C + c));
      *reinterpret_cast<f32x4*>(a.y + t * a.C + c) = k * (gg - m1 - xh * m2);
    }
  }
}

// ---- weight gradient of a pointwise convolution with few channels over MANY tokens -----------------------------------------
// dW[n][k] = sum_t dy[t][n] x[t][k], N, K <= 64, T = 32 768 ... 524 288: the stems and heads of the Swin U-net stages at
// full resolution (32 -> 32 at 512^2, 64 -> 64 at 256^2, ...).  A 64-wide MFMA tile is half empty at these widths and the grouped
// fp32 weight-gradient launch of csrc/dense32.hip spent ~140 us per such problem (~1 TB/s); the problem is a streaming reduction of
// 2 x T x 32 floats.  Here: a workgroup owns a range of 1 024 tokens, stages 64 tokens of dy and x at a time in LDS (coalesced
// 16-byte loads), a thread owns a 4 x 4 block of dW and a share of the 64 tokens (256 / (N K / 16) shares); shares fold through LDS
// in a fixed order, ranges through fold_partials(): deterministic.
constexpr int PW_TR = 1024;   // tokens per workgroup
// TA = float, or _Float16 rows of an fp16-autocast step (converted while staged; the sums are fp32 either way); BIAS: the partials
// carry N more floats behind the N x K block - db[n] = sum_t dy[t][n], summed by the threads that own a tile's first column block.
template <typename TA>
__device__ __forceinline__ f32x4 pw_ld4(const TA* p);
template <>
__device__ __forceinline__ f32x4 pw_ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 pw_ld4<_Float16>(const _Float16* p) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <typename TA, bool BIAS>
__global__ __launch_bounds__(256) void pw_wgrad_small_kernel(const TA* __restrict__ dy, const TA* __restrict__ x,
                                                             float* __restrict__ part, long T, int N, int K, int shares) {
  __shared__ __attribute__((aligned(16))) float sD[64 * 64];
  __shared__ __attribute__((aligned(16))) float sX[64 * 64];
  const int tid = threadIdx.x;
  const int tk = K >> 2, tiles = (N >> 2) * tk;
  const int tile = tid % tiles, sp = tid / tiles;
  const bool work = sp < shares;
  const int n0 = (tile / tk) * 4, k0 = (tile % tk) * 4;
  const long t0 = (long)blockIdx.x * PW_TR;
  const long t1 = t0 + PW_TR < T ? t0 + PW_TR : T;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool bias_owner = BIAS && k0 == 0;
  const long E = (long)N * K + (BIAS ? N : 0);
  const int n4 = N >> 2, k4 = K >> 2;
  // the next chunk's pieces are requested BEFORE the current chunk's products (register double buffer): without it every chunk
  // exposed a full memory round trip between its two barriers (16 per workgroup: ~100 us per launch where the bytes take 30)
  constexpr int PCS = 4;                       // 64 tokens x <= 16 quads = <= 1 024 pieces per operand: <= 4 per thread
  f32x4 rd[PCS], rx[PCS];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](long tc) {
    const int nt = t1 - tc < 64 ? (int)(t1 - tc) : 64;
#pragma unroll
    for (int j = 0; j < PCS; ++j) {
      const int i = tid + 256 * j;
      const int ttd = i / n4, qd = i - ttd * n4;
      rd[j] = (i < 64 * n4 && ttd < nt) ? pw_ld4<TA>(dy + (tc + ttd) * N + 4 * qd) : z4;
      const int ttx = i / k4, qx = i - ttx * k4;
      rx[j] = (i < 64 * k4 && ttx < nt) ? pw_ld4<TA>(x + (tc + ttx) * K + 4 * qx) : z4;
    }
  };
  if (t0 < t1) fetch(t0);
  for (long tc = t0; tc < t1; tc += 64) {
#pragma unroll
    for (int j = 0; j < PCS; ++j) {
      const int i = tid + 256 * j;
      if (i < 64 * n4) *reinterpret_cast<f32x4*>(sD + (i / n4) * N + 4 * (i % n4)) = rd[j];
      if (i < 64 * k4) *reinterpret_cast<f32x4*>(sX + (i / k4) * K + 4 * (i % k4)) = rx[j];
    }
    __syncthreads();
    if (tc + 64 < t1) fetch(tc + 64);
    if (work) {
      for (int tt = sp; tt < 64; tt += shares) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(sD + tt * N + n0);
        const f32x4 v = *reinterpret_cast<const f32x4*>(sX + tt * K + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += d[i] * v[j];
        if (bias_owner) bsum += d;
      }
    }
    __syncthreads();
  }
  // fold the token shares: share s of tile `tile` parks its block at sD[(s * tiles + tile) * 16 ...] (shares * tiles <= 256 blocks
  // of 16 floats = the 4 096 floats of sD), share 0 sums them in share order
  if (work) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(sD + (sp * tiles + tile) * 16 + 4 * i) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    if (bias_owner) *reinterpret_cast<f32x4*>(sX + (sp * n4 + (n0 >> 2)) * 4) = bsum;     // sX is free behind the loop's last barrier
  }
  __syncthreads();
  if (work && sp == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 s = *reinterpret_cast<const f32x4*>(sD + tile * 16 + 4 * i);
      for (int q = 1; q < shares; ++q) s += *reinterpret_cast<const f32x4*>(sD + (q * tiles + tile) * 16 + 4 * i);
      *reinterpret_cast<f32x4*>(part + (long)blockIdx.x * E + (long)(n0 + i) * K + k0) = s;
    }
    if (bias_owner) {
      f32x4 s = *reinterpret_cast<const f32x4*>(sX + (n0 >> 2) * 4);
      for (int q = 1; q < shares; ++q) s += *reinterpret_cast<const f32x4*>(sX + (q * n4 + (n0 >> 2)) * 4);
      *reinterpret_cast<f32x4*>(part + (long)blockIdx.x * E + (long)N * K + n0) = s;
    }
  }
}

// ---- 1x1 convolution to a handful of output channels: the side heads and the fuse convolution of the X^2-Nets --------------
// y[b][n][p] = bias[n] + sum_k W[n][k] x(b, p, k), N <= 8: /root/reference/nnunetv2/nets/swt2net.py:1021-1028 `side1 .. side6`
// (C -> classes, 1x1) and `outconv` (6 classes -> classes) of SwT2Net.  The output is NCHW ([B][N][P]: what the loss and the
// up-sampling of the deep-supervision outputs take); the input is addressed through (batch, position, channel) strides: token-major
// ([B][P][K], the layout the stages hand over: xsp = K, xsk = 1) or NCHW (the concatenated side outputs: xsp = 1, xsk = P).
// HBM-bound by construction; the weight gradient is per-range partials + fold_partials (deterministic).
constexpr int HD_MAXN = 8;
// T = float (the fp32 device step of SwT2Net) or _Float16 (activations of the fp16-autocast steps: M2Net's fuse convolution); the
// weights, the bias and every sum are fp32 either way.
template <typename T>
struct HeadArgsT {
  const T* x;
  const float* w;      // [N][K]
  const float* bias;   // [N] or null
  const T* dy;         // [B][N][P]
  T* y;                // forward: [B][N][P];  dgrad: dx in the layout of x
  long xsb, xsp, xsk;
  int B, N, K;
  long P;
};
template <typename T>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(HeadArgsT<T> a) {
  extern __shared__ float sw[];        // [N][K]
  for (int i = threadIdx.x; i < a.N * a.K; i += 256) sw[i] = a.w[i];
  __syncthreads();
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)a.B * a.P) return;
  const int b = (int)(t / a.P);
  const long p = t - (long)b * a.P;
  const T* xp = a.x + b * a.xsb + p * a.xsp;
  float acc[HD_MAXN];
#pragma unroll
  for (int n = 0; n < HD_MAXN; ++n) acc[n] = (n < a.N && a.bias) ? a.bias[n] : 0.f;
  if (sizeof(T) == 4 && a.xsk == 1 && (a.K & 3) == 0) {
    for (int k = 0; k < a.K; k += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xp) + k);
#pragma unroll
      for (int n = 0; n < HD_MAXN; ++n)
        if (n < a.N) {
          const float* wr = sw + n * a.K + k;
          acc[n] += wr[0] * v[0] + wr[1] * v[1] + wr[2] * v[2] + wr[3] * v[3];
        }
    }
  } else {
    for (int k = 0; k < a.K; ++k) {
      const float v = (float)xp[k * a.xsk];
#pragma unroll
      for (int n = 0; n < HD_MAXN; ++n)
        if (n < a.N) acc[n] += sw[n * a.K + k] * v;
    }
  }
#pragma unroll
  for (int n = 0; n < HD_MAXN; ++n)
    if (n < a.N) a.y[((long)b * a.N + n) * a.P + p] = (T)acc[n];
}
// dx(b, p, k) = sum_n W[n][k] dy[b][n][p]; thread = (position, channel) with the channel fastest (token-major) or the position
// fastest (NCHW)
template <typename T>
__global__ __launch_bounds__(256) void head1x1_dgrad_kernel(HeadArgsT<T> a) {
  extern __shared__ float sw[];
  for (int i = threadIdx.x; i < a.N * a.K; i += 256) sw[i] = a.w[i];
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.B * a.P * a.K;
  if (i >= total) return;
  long t;
  int k;
  if (a.xsk == 1) { t = i / a.K; k = (int)(i - t * a.K); }
  else { const long bk = i / a.P; const long p = i - bk * a.P; k = (int)(bk % a.K); t = (bk / a.K) * a.P + p; }
  const int b = (int)(t / a.P);
  const long p = t - (long)b * a.P;
  float v = 0.f;
#pragma unroll
  for (int n = 0; n < HD_MAXN; ++n)
    if (n < a.N) v += sw[n * a.K + k] * (float)a.dy[((long)b * a.N + n) * a.P + p];
  a.y[b * a.xsb + p * a.xsp + k * a.xsk] = (T)v;
}
// partial[range][n][k] = sum over the range's positions of dy[b][n][p] x(b, p, k); column K of a row = the bias partial sum dy.
// Workgroup = 256 channels (blockIdx.y) x one range of HD_TR positions (never across samples: ranges are cut per sample).
constexpr int HD_TR = 512;
// (tr = positions per range, <= HD_TR: 64 for these wide heads - their maps are small (128^2 ... 16^2), and with 512 positions per
//  range 64 workgroups walked 512 dependent-latency iterations each: 146 us per launch)
template <typename T>
__global__ __launch_bounds__(256) void head1x1_wgrad_kernel(HeadArgsT<T> a, float* __restrict__ part, int ranges_per_sample, int tr) {
  __shared__ float sdy[HD_MAXN][HD_TR];
  const int b = blockIdx.x / ranges_per_sample, r = blockIdx.x % ranges_per_sample;
  const long p0 = (long)r * tr;
  const int np = a.P - p0 < tr ? (int)(a.P - p0) : tr;
  for (int i = threadIdx.x; i < a.N * tr; i += 256) {
    const int n = i / tr, pp = i - n * tr;
    sdy[n][pp] = pp < np ? (float)a.dy[((long)b * a.N + n) * a.P + p0 + pp] : 0.f;
  }
  __syncthreads();
  const int k = blockIdx.y * 256 + threadIdx.x;
  float acc[HD_MAXN];
#pragma unroll
  for (int n = 0; n < HD_MAXN; ++n) acc[n] = 0.f;
  if (k < a.K) {
    const T* xp = a.x + b * a.xsb + p0 * a.xsp + k * a.xsk;
    for (int pp = 0; pp < np; ++pp) {
      const float v = (float)xp[pp * a.xsp];
#pragma unroll
      for (int n = 0; n < HD_MAXN; ++n)
        if (n < a.N) acc[n] += sdy[n][pp] * v;
    }
  } else if (k == a.K) {      // the bias column
    for (int pp = 0; pp < np; ++pp)
#pragma unroll
      for (int n = 0; n < HD_MAXN; ++n)
        if (n < a.N) acc[n] += sdy[n][pp];
  }
  if (k <= a.K) {
#pragma unroll
    for (int n = 0; n < HD_MAXN; ++n)
      if (n < a.N) part[((long)blockIdx.x * a.N + n) * (a.K + 1) + k] = acc[n];
  }
}

// The same partials for FEW input channels (K + 1 <= 128: the full-resolution heads - 32 channels over 262 144 positions per sample -
// and the 12-channel fuse convolution): with one thread per channel 13 ... 33 lanes of a workgroup worked (184 us per launch).  Here a
// chunk of 64 positions of x goes through LDS ([pos][K], loaded coalesced in either layout) and the threads are (channel, position
// group): Kp = K + 1 rounded up to a power of two, 256 / Kp groups that split the chunk's positions, folded through LDS at the end.
template <typename T>
__global__ __launch_bounds__(256) void head1x1_wgrad_small_kernel(HeadArgsT<T> a, float* __restrict__ part, int ranges_per_sample, int Kp) {
  __shared__ float sdy[HD_MAXN][64];
  __shared__ __attribute__((aligned(16))) float sx[64 * 128];       // [pos][K], reused as the fold buffer [g][n][Kp]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / ranges_per_sample, r = blockIdx.x % ranges_per_sample;
  const long p0 = (long)r * HD_TR;
  const int np = a.P - p0 < HD_TR ? (int)(a.P - p0) : HD_TR;
  const int k = tid % Kp, g = tid / Kp, G = 256 / Kp;
  float acc[HD_MAXN];
#pragma unroll
  for (int n = 0; n < HD_MAXN; ++n) acc[n] = 0.f;
  const T* xb = a.x + b * a.xsb;
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int nc = np - c0 < 64 ? np - c0 : 64;
    for (int i = tid; i < a.N * 64; i += 256) {
      const int n = i >> 6, pp = i & 63;
      sdy[n][pp] = pp < nc ? (float)a.dy[((long)b * a.N + n) * a.P + p0 + c0 + pp] : 0.f;
    }
    for (int i = tid; i < 64 * a.K; i += 256) {
      int pp, kk;
      if (a.xsk == 1) { pp = i / a.K; kk = i - pp * a.K; }
      else { kk = i >> 6; pp = i & 63; }
      sx[pp * a.K + kk] = pp < nc ? (float)xb[(p0 + c0 + pp) * a.xsp + kk * a.xsk] : 0.f;
    }
    __syncthreads();
    if (k <= a.K) {
      for (int pp = g; pp < 64; pp += G) {
        const float v = k < a.K ? sx[pp * a.K + k] : 1.f;          // column K: the bias gradient, sum of dy
#pragma unroll
        for (int n = 0; n < HD_MAXN; ++n)
          if (n < a.N) acc[n] += sdy[n][pp] * v;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < HD_MAXN; ++n)
    if (n < a.N) sx[(g * HD_MAXN + n) * Kp + k] = acc[n];
  __syncthreads();
  if (g == 0 && k <= a.K) {
#pragma unroll
    for (int n = 0; n < HD_MAXN; ++n)
      if (n < a.N) {
        float t = 0.f;
        for (int q = 0; q < G; ++q) t += sx[(q * HD_MAXN + n) * Kp + k];
        part[((long)blockIdx.x * a.N + n) * (a.K + 1) + k] = t;
      }
  }
}

}  // namespace nnz

// x, y: [B][H][W][C] fp32 (C a multiple of 4, 16-byte aligned); w: [C][3][3]; bias: [C] or NULL.  flip = 1: the input gradient
// (x = dy, y = dx; the bias is ignored).
extern "C" int nnz_dw3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                  int flip, void* stream) {
  using namespace nnz;
  if (!x || !w || !y || B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || ((size_t)x & 15) || ((size_t)y & 15)) return NNZ_EINVAL;
  DwArgs32 a = {x, w, flip ? nullptr : bias, y, B, H, W, C};
  const int C4 = C >> 2;
  const int cq = C4 < 256 ? C4 : 256;
  const int tl = 256 / cq;
  const long T = (long)B * H * W;
  const long bx = (T + (long)tl * DW_TOK - 1) / ((long)tl * DW_TOK);
  const int by = (C4 + cq - 1) / cq;
  if (bx > 0x7fffffffL || by > 65535) return NNZ_EINVAL;
  if (flip) NNZ_LAUNCH(dw3x3_nhwc_kernel<true>, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream, a, cq, tl);
  else NNZ_LAUNCH(dw3x3_nhwc_kernel<false>, dim3((unsigned)bx, by), dim3(256), 0, (hipStream_t)stream, a, cq, tl);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

static int dw3x3_ranges(long T) {
  long r = (T + 255) / 256;
  return (int)(r < 1 ? 1 : (r > 8192 ? 8192 : r));
}
extern "C" long nnz_dw3x3_nhwc_wgrad_workspace_floats(int B, int H, int W, int C) {
  if (B < 1 || H < 1 || W < 1 || C < 1) return 0;
  const int ranges = dw3x3_ranges((long)B * H * W);
  return (long)ranges * C * 9 + nnz::fold_partials_scratch_floats(ranges, (long)C * 9);
}
// dw: [C][3][3] is WRITTEN (no zero fill); workspace: nnz_dw3x3_nhwc_wgrad_workspace_floats floats.  Deterministic.
extern "C" int nnz_dw3x3_nhwc_wgrad_f32(const float* x, const float* dy, float* workspace, float* dw, int B, int H, int W, int C,
                                        void* stream) {
  using namespace nnz;
  if (!x || !dy || !workspace || !dw || B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || ((size_t)x & 15) || ((size_t)dy & 15))
    return NNZ_EINVAL;
  const long T = (long)B * H * W;
  const int ranges = dw3x3_ranges(T);
  const long per = (T + ranges - 1) / ranges;
  NNZ_LAUNCH(dw3x3_nhwc_wgrad_kernel, dim3((C + 15) / 16, ranges), dim3(256), 0, (hipStream_t)stream, x, dy, workspace, B, H, W, C,
             per);
  float* scratch = fold_partials_scratch_floats(ranges, (long)C * 9) ? workspace + (long)ranges * C * 9 : nullptr;
  hipError_t e = fold_partials(workspace, ranges, (long)C * 9, (long)C * 9, dw, (hipStream_t)stream, scratch);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// y = relu(batch_norm(x)) on [T][C] fp32, C % 4 == 0.  training != 0: batch statistics (biased variance for the normalisation, the running
// estimates updated in place with `momentum` and the unbiased variance - torch.nn.BatchNorm2d's rule); training == 0: the running
// estimates.  mean / rstd [C] are written for the backward.
extern "C" int nnz_bn_relu_nhwc_forward_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                            float* running_var, float* mean, float* rstd, float* y, long T, int C, int training,
                                            float momentum, float eps, void* stream) {
  using namespace nnz;
  if (!x || !gamma || !beta || !running_mean || !running_var || !mean || !rstd || !y || T < 1 || C < 4 || (C & 3)) return NNZ_EINVAL;
  BnArgs32 a = {};
  a.x = x; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var; a.mean = mean; a.rstd = rstd;
  a.y = y; a.T = T; a.C = C; a.training = training; a.momentum = momentum; a.eps = eps;
  const dim3 grid((C + 15) / 16);
  if (T <= 64 * 8) NNZ_LAUNCH(bn_relu_nhwc_fwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (T <= 64 * 32) NNZ_LAUNCH(bn_relu_nhwc_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else NNZ_LAUNCH(bn_relu_nhwc_fwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// backward of the training-mode unit: dx [T][C], dgamma / dbeta [C] are WRITTEN.  Deterministic.
extern "C" int nnz_bn_relu_nhwc_backward_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                             const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, long T,
                                             int C, void* stream) {
  using namespace nnz;
  if (!x || !dy || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || T < 1 || C < 4 || (C & 3)) return NNZ_EINVAL;
  BnArgs32 a = {};
  a.x = x; a.dy = dy; a.gamma = gamma; a.beta = beta; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
  a.y = dx; a.dgamma = dgamma; a.dbeta = dbeta; a.T = T; a.C = C;
  const dim3 grid((C + 15) / 16);
  if (T <= 64 * 8) NNZ_LAUNCH(bn_relu_nhwc_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (T <= 64 * 24) NNZ_LAUNCH(bn_relu_nhwc_bwd_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else NNZ_LAUNCH(bn_relu_nhwc_bwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// dW [N][K] = dy^T x over T tokens; N, K multiples of 4, <= 64; dy [T][N], x [T][K] fp32.  workspace:
// nnz_pw_wgrad_small_workspace_floats(T, N, K) floats.  dW is WRITTEN; deterministic (fixed-order folds).
extern "C" long nnz_pw_wgrad_small_workspace_floats_b(long T, int N, int K, int with_bias) {
  if (T < 1 || N < 4 || K < 4) return 0;
  const long ranges = (T + nnz::PW_TR - 1) / nnz::PW_TR;
  const long E = (long)N * K + (with_bias ? N : 0);
  return ranges * E + nnz::fold_partials_scratch_floats((int)ranges, (long)N * K);
}
extern "C" long nnz_pw_wgrad_small_workspace_floats(long T, int N, int K) { return nnz_pw_wgrad_small_workspace_floats_b(T, N, K, 0); }
template <typename TA>
static int pw_wgrad_small_t(const TA* dy, const TA* x, float* workspace, float* dW, float* db, long T, int N, int K, void* stream) {
  using namespace nnz;
  const size_t al = sizeof(TA) == 4 ? 15 : 7;
  if (!dy || !x || !workspace || !dW || T < 1 || N < 4 || K < 4 || N > 64 || K > 64 || (N & 3) || (K & 3) || ((size_t)dy & al) ||
      ((size_t)x & al))
    return NNZ_EINVAL;
  const long ranges = (T + PW_TR - 1) / PW_TR;
  if (ranges > 0x7fffffffL) return NNZ_EINVAL;
  const int tiles = (N >> 2) * (K >> 2);
  int shares = 256 / tiles;
  if (shares > 16) shares = 16;
  const long NK = (long)N * K, E = NK + (db ? N : 0);
  if (db) NNZ_LAUNCH((pw_wgrad_small_kernel<TA, true>), dim3((unsigned)ranges), dim3(256), 0, (hipStream_t)stream, dy, x, workspace, T, N, K, shares);
  else NNZ_LAUNCH((pw_wgrad_small_kernel<TA, false>), dim3((unsigned)ranges), dim3(256), 0, (hipStream_t)stream, dy, x, workspace, T, N, K, shares);
  float* scratch = fold_partials_scratch_floats((int)ranges, NK) ? workspace + ranges * E : nullptr;
  hipError_t e = fold_partials(workspace, (int)ranges, E, NK, dW, (hipStream_t)stream, scratch);
  if (e != hipSuccess) return (int)e;
  if (db) {
    e = fold_partials(workspace + NK, (int)ranges, E, (long)N, db, (hipStream_t)stream, scratch);
    if (e != hipSuccess) return (int)e;
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_pw_wgrad_small_f32(const float* dy, const float* x, float* workspace, float* dW, long T, int N, int K,
                                      void* stream) {
  return pw_wgrad_small_t<float>(dy, x, workspace, dW, nullptr, T, N, K, stream);
}
// the same for fp32 or fp16 (is_f16) token rows and, db != NULL, the bias gradient db [N] = sum_t dy[t] from the same pass
// (workspace: nnz_pw_wgrad_small_workspace_floats_b(T, N, K, db != NULL))
extern "C" int nnz_pw_wgrad_small(const void* dy, const void* x, int is_f16, float* workspace, float* dW, float* db, long T, int N,
                                  int K, void* stream) {
  return is_f16 ? pw_wgrad_small_t<_Float16>((const _Float16*)dy, (const _Float16*)x, workspace, dW, db, T, N, K, stream)
                : pw_wgrad_small_t<float>((const float*)dy, (const float*)x, workspace, dW, db, T, N, K, stream);
}

// 1x1 convolution to N <= 8 channels.  x: element (b, p, k) at x[b * xsb + p * xsp + k * xsk] (token-major: xsp = K, xsk = 1; NCHW:
// xsp = 1, xsk = P), y / dy: [B][N][P] contiguous, w [N][K], bias [N] or NULL; N * K <= 8 192.
static int head1x1_ok(int B, int N, int K, long P) { return B >= 1 && N >= 1 && N <= nnz::HD_MAXN && K >= 1 && (long)N * K <= 8192 && P >= 1; }
template <typename T>
static int head1x1_forward_t(const T* x, const float* w, const float* bias, T* y, int B, int N, int K, long P, long xsb, long xsp,
                             long xsk, void* stream) {
  using namespace nnz;
  if (!x || !w || !y || !head1x1_ok(B, N, K, P)) return NNZ_EINVAL;
  HeadArgsT<T> a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.xsb = xsb; a.xsp = xsp; a.xsk = xsk; a.B = B; a.N = N; a.K = K; a.P = P;
  const long blocks = ((long)B * P + 255) / 256;
  if (blocks > 0x7fffffffL) return NNZ_EINVAL;
  NNZ_LAUNCH(head1x1_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), N * K * (int)sizeof(float), (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_head1x1_forward_f32(const float* x, const float* w, const float* bias, float* y, int B, int N, int K, long P,
                                       long xsb, long xsp, long xsk, void* stream) {
  return head1x1_forward_t<float>(x, w, bias, y, B, N, K, P, xsb, xsp, xsk, stream);
}
// the same with fp16 activations in and out (x, y: IEEE half), fp32 weights / bias / sums: the fuse convolution of an fp16-autocast
// step (/root/reference/nnunetv2/nets/m2net.py:881 `outconv`)
extern "C" int nnz_head1x1_forward_f16(const void* x, const float* w, const float* bias, void* y, int B, int N, int K, long P,
                                       long xsb, long xsp, long xsk, void* stream) {
  return head1x1_forward_t<_Float16>((const _Float16*)x, w, bias, (_Float16*)y, B, N, K, P, xsb, xsp, xsk, stream);
}
// dx in the layout of x (every element written)
template <typename T>
static int head1x1_dgrad_t(const T* dy, const float* w, T* dx, int B, int N, int K, long P, long xsb, long xsp, long xsk, void* stream) {
  using namespace nnz;
  if (!dy || !w || !dx || !head1x1_ok(B, N, K, P)) return NNZ_EINVAL;
  HeadArgsT<T> a = {};
  a.dy = dy; a.w = w; a.y = dx; a.xsb = xsb; a.xsp = xsp; a.xsk = xsk; a.B = B; a.N = N; a.K = K; a.P = P;
  const long blocks = ((long)B * P * K + 255) / 256;
  if (blocks > 0x7fffffffL) return NNZ_EINVAL;
  NNZ_LAUNCH(head1x1_dgrad_kernel<T>, dim3((unsigned)blocks), dim3(256), N * K * (int)sizeof(float), (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_head1x1_dgrad_f32(const float* dy, const float* w, float* dx, int B, int N, int K, long P, long xsb, long xsp,
                                     long xsk, void* stream) {
  return head1x1_dgrad_t<float>(dy, w, dx, B, N, K, P, xsb, xsp, xsk, stream);
}
extern "C" int nnz_head1x1_dgrad_f16(const void* dy, const float* w, void* dx, int B, int N, int K, long P, long xsb, long xsp,
                                     long xsk, void* stream) {
  return head1x1_dgrad_t<_Float16>((const _Float16*)dy, w, (_Float16*)dx, B, N, K, P, xsb, xsp, xsk, stream);
}
static int head1x1_range(int K) { return K + 1 <= 128 ? nnz::HD_TR : 64; }
extern "C" long nnz_head1x1_wgrad_workspace_floats(int B, int N, int K, long P) {
  if (!head1x1_ok(B, N, K, P)) return 0;
  const int tr = head1x1_range(K);
  const long ranges = (long)B * ((P + tr - 1) / tr);
  return ranges * N * (K + 1) + nnz::fold_partials_scratch_floats((int)ranges, (long)N * (K + 1));
}
// dwb [N][K + 1] is WRITTEN: columns 0 .. K - 1 = dW, column K = db.  Deterministic.
template <typename T>
static int head1x1_wgrad_t(const T* x, const T* dy, float* workspace, float* dwb, int B, int N, int K, long P, long xsb, long xsp,
                           long xsk, void* stream) {
  using namespace nnz;
  if (!x || !dy || !workspace || !dwb || !head1x1_ok(B, N, K, P)) return NNZ_EINVAL;
  HeadArgsT<T> a = {};
  a.x = x; a.dy = dy; a.xsb = xsb; a.xsp = xsp; a.xsk = xsk; a.B = B; a.N = N; a.K = K; a.P = P;
  const int tr = head1x1_range(K);
  const int rps = (int)((P + tr - 1) / tr);
  const long ranges = (long)B * rps;
  if (ranges > 0x7fffffffL) return NNZ_EINVAL;
  if (K + 1 <= 128) {
    int Kp = 1;
    while (Kp < K + 1) Kp <<= 1;
    NNZ_LAUNCH(head1x1_wgrad_small_kernel<T>, dim3((unsigned)ranges), dim3(256), 0, (hipStream_t)stream, a, workspace, rps, Kp);
  } else {
    NNZ_LAUNCH(head1x1_wgrad_kernel<T>, dim3((unsigned)ranges, (K + 1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, workspace, rps, tr);
  }
  const long E = (long)N * (K + 1);
  float* scratch = fold_partials_scratch_floats((int)ranges, E) ? workspace + ranges * E : nullptr;
  hipError_t e = fold_partials(workspace, (int)ranges, E, E, dwb, (hipStream_t)stream, scratch);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_head1x1_wgrad_f32(const float* x, const float* dy, float* workspace, float* dwb, int B, int N, int K, long P,
                                     long xsb, long xsp, long xsk, void* stream) {
  return head1x1_wgrad_t<float>(x, dy, workspace, dwb, B, N, K, P, xsb, xsp, xsk, stream);
}
// fp16 x / dy, fp32 partial sums and result (same workspace size)
extern "C" int nnz_head1x1_wgrad_f16(const void* x, const void* dy, float* workspace, float* dwb, int B, int N, int K, long P,
                                     long xsb, long xsp, long xsk, void* stream) {
  return head1x1_wgrad_t<_Float16>((const _Float16*)x, (const _Float16*)dy, workspace, dwb, B, N, K, P, xsb, xsp, xsk, stream);
}
