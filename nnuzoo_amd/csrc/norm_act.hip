// InstanceNorm(affine) + LeakyReLU on channels-last fp16 activations, forward and backward (gfx950).
// Replaces nn.InstanceNorm3d/2d + nn.LeakyReLU(0.01, inplace) of every PlainConvUNet conv block (arch kwargs at
// /root/reference/nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:285-305:
// norm_op_kwargs {'eps': 1e-5, 'affine': True}, nonlin LeakyReLU).
//
// HBM-bound: every kernel moves 16 bytes per lane (8 fp16 channels of one voxel) and keeps the per-(n, c)
// statistics in fp32.  Statistics are biased (population) variance like torch's instance_norm.
//   stats   : x -> {sum, sumsq}[n][c]                                   (1 read)
//   apply   : y = lrelu((x - mean) * rstd * gamma + beta)               (1 read, 1 write)
//   bwd_red : {sum g', sum g' * xhat}[n][c], g' = g * lrelu'(y)         (2 reads)
//   bwd_app : dx = gamma * rstd * (g' - mean(g') - xhat * mean(g' xhat)) (2 reads, 1 write)
#include "common.hpp"

namespace nnz {

struct NormArgs {
  const f16* x;      // raw conv output [N][V][ldx]
  const f16* g;      // upstream gradient [N][V][ldg] (bwd only)
  f16* y;            // output [N][V][ldy] (y for fwd, dx for bwd)
  float* stats;      // [N][C][2] {sum, sumsq}
  float* red;        // [N][C][2] {sum g', sum g' xhat} (bwd)
  const float* gamma;
  const float* beta;
  float* dgamma;     // optional outputs of bwd apply: sum over n of red[n][c][1] / red[n][c][0]
  float* dbeta;
  int N, C, ldx, ldg, ldy;
  long V;
  int vpb;           // voxels per block
  float eps, slope;
  // deterministic variants (fixed-point accumulators, common.hpp): `nstat` [N][C][4] = {mean, rstd, rstd*gamma, beta -
  // mean*rstd*gamma} replaces stats + gamma + beta as the INPUT of modes 1-3 and is the OUTPUT of mode 0; `nred` [N][C][2]
  // = {mean g', mean g' xhat} is the output of mode 2 (with dgamma / dbeta) and the input of mode 3; `sums` [N][C][2] =
  // {sum, sumsq} is an alternative mode-0 output (plain moments, e.g. a bias gradient)
  FxAcc* acc;        // [N][C][2], zero between launches
  unsigned* counter;
  float* nstat;
  float* nred;
  float* sums;
};

__device__ __forceinline__ void load8(const f16* p, float (&v)[8]) {
  const f16x8 h = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
__device__ __forceinline__ void store8(f16* p, const float (&v)[8]) {
  f16x8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (f16)v[i];
  *reinterpret_cast<f16x8*>(p) = h;
}

// MODE 0: stats, 1: apply fwd, 2: bwd reduce, 3: bwd apply
// Thread = (row r, channel group cg of 8 fp16 = 16 bytes).  Each thread keeps UNR independent 16-byte loads in
// flight (the loop is latency-bound otherwise: Little's law needs ~64 KB in flight per CU for HBM3E); reductions
// go lane -> LDS slab [rows][2C] -> one column sum per thread -> one global atomic per (block, channel).
// (LDS float atomics are avoided: ds_add_f32 measured an order of magnitude slower than plain LDS traffic.)
constexpr int NRM_UNR = 4;

template <int MODE>
__global__ __launch_bounds__(256) void norm_kernel(NormArgs a) {
  extern __shared__ float lred[];  // [rows][2C] (MODE 0 / 2)
  const int CG = a.C >> 3;
  const int rows = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG;
  const int r = tid / CG;
  const int n = blockIdx.y;
  const long v0 = (long)blockIdx.x * a.vpb;
  long v1 = v0 + a.vpb;
  if (v1 > a.V) v1 = a.V;
  const bool active = r < rows;

  if (MODE == 3 && a.dgamma && !a.nred && blockIdx.x == 0 && blockIdx.y == 0) {
    // parameter gradients of the affine: one workgroup folds the per-sample reductions (saves a torch reduction and
    // two copies per conv block)
    for (int c = tid; c < a.C; c += 256) {
      float sg = 0.f, sb = 0.f;
      for (int nn = 0; nn < a.N; ++nn) {
        sb += a.red[((long)nn * a.C + c) * 2 + 0];
        sg += a.red[((long)nn * a.C + c) * 2 + 1];
      }
      a.dgamma[c] = sg;
      a.dbeta[c] = sb;
    }
  }

  float scale[8], shift[8], mean[8], rstd[8], m1[8], m2[8];
  if (MODE != 0 && active && a.nstat) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cg * 8 + i;
      const f32x4 t = *reinterpret_cast<const f32x4*>(a.nstat + ((long)n * a.C + c) * 4);
      mean[i] = t[0];
      rstd[i] = t[1];
      scale[i] = t[2];
      shift[i] = t[3];
      if (MODE == 3) {
        m1[i] = a.nred[((long)n * a.C + c) * 2 + 0];
        m2[i] = a.nred[((long)n * a.C + c) * 2 + 1];
      }
    }
  } else if (MODE != 0 && active) {
    const float invV = 1.f / (float)a.V;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cg * 8 + i;
      const float s = a.stats[((long)n * a.C + c) * 2 + 0];
      const float ss = a.stats[((long)n * a.C + c) * 2 + 1];
      const float mu = s * invV;
      float var = ss * invV - mu * mu;
      var = var < 0.f ? 0.f : var;
      mean[i] = mu;
      rstd[i] = rsqrtf(var + a.eps);
      scale[i] = rstd[i] * a.gamma[c];
      shift[i] = a.beta[c] - mu * scale[i];
      if (MODE == 3) {
        m1[i] = a.red[((long)n * a.C + c) * 2 + 0] * invV;
        m2[i] = a.red[((long)n * a.C + c) * 2 + 1] * invV;
      }
    }
  }

  float acc0[8], acc1[8], pilot[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc0[i] = acc1[i] = pilot[i] = 0.f;
  if (MODE == 0 && a.acc && active) {
    // moments about a pilot value (the block's first voxel) instead of about 0: no cancellation when |mean| >> std
    float t[8];
    load8(a.x + ((long)n * a.V + v0) * a.ldx + cg * 8, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) pilot[i] = t[i];
  }

  if (active) {
    const long base = (long)n * a.V;
    for (long v = v0 + r; v < v1; v += (long)rows * NRM_UNR) {
      f16x8 xh[NRM_UNR], gh[NRM_UNR];
#pragma unroll
      for (int k = 0; k < NRM_UNR; ++k) {
        const long vv = v + (long)k * rows;
        if (vv < v1) {
          xh[k] = *reinterpret_cast<const f16x8*>(a.x + (base + vv) * a.ldx + cg * 8);
          if (MODE >= 2) gh[k] = *reinterpret_cast<const f16x8*>(a.g + (base + vv) * a.ldg + cg * 8);
        }
      }
#pragma unroll
      for (int k = 0; k < NRM_UNR; ++k) {
        const long vv = v + (long)k * rows;
        if (vv >= v1) continue;
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = (float)xh[k][i] - pilot[i];
            acc0[i] += x;
            acc1[i] += x * x;
          }
        } else if (MODE == 1) {
          // (common.hpp norm_lrelu8: the rounding points of torch's autocast op sequence; shared with the consumer-side norm)
          const f16x8 o = __builtin_bit_cast(f16x8, norm_lrelu8(__builtin_bit_cast(u32x4, xh[k]), scale, shift, slope_pair(a.slope)));
          *reinterpret_cast<f16x8*>(a.y + (base + vv) * a.ldy + cg * 8) = o;
        } else {
          f16x8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = (float)xh[k][i], g = (float)gh[k][i];
            const float pre = x * scale[i] + shift[i];
            const float gp = pre > 0.f ? g : g * a.slope;
            const float xn = (x - mean[i]) * rstd[i];
            if (MODE == 2) {
              acc0[i] += gp;
              acc1[i] += gp * xn;
            } else {
              o[i] = (f16)(scale[i] * (gp - m1[i] - xn * m2[i]));
            }
          }
          if (MODE == 3) *reinterpret_cast<f16x8*>(a.y + (base + vv) * a.ldy + cg * 8) = o;
        }
      }
    }
  }

  if (MODE == 0 || MODE == 2) {
    const int C2 = 2 * a.C;
    // When the channel groups divide the wave (CG = 4 .. 64, a power of two - every layer but the 320-channel ones) the
    // rows a wave holds are folded with lane exchanges first and the slab has one row per WAVE: the lone finishing wave then
    // adds 4 values per channel instead of walking up to 64 dependent LDS reads per channel (3 us per workgroup, measured as
    // the tail of every reducing launch).  Fixed exchange order: deterministic.
    const bool fold = a.acc && CG <= 64 && (CG & (CG - 1)) == 0;
    int nrows = rows;
    if (fold) {
      for (int off = CG; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc0[i] += __shfl_xor(acc0[i], off, 64);
          acc1[i] += __shfl_xor(acc1[i], off, 64);
        }
      nrows = 4;
      if ((tid & 63) < CG) {
        const int w = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          lred[w * C2 + (cg * 8 + i) * 2 + 0] = acc0[i];
          lred[w * C2 + (cg * 8 + i) * 2 + 1] = acc1[i];
        }
      }
    } else if (active) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        lred[r * C2 + (cg * 8 + i) * 2 + 0] = acc0[i];
        lred[r * C2 + (cg * 8 + i) * 2 + 1] = acc1[i];
      }
    }
    __syncthreads();
    if (a.acc) {
      // deterministic: fixed-order fold inside the workgroup, fixed-point adds across workgroups, the last workgroup of
      // the launch turns the totals into the float tables.  Wave 0 does all of it - the other waves leave: no workgroup
      // waits on the ticket's round trip.
      if (tid >= 64) return;
      const long nrec = (long)a.N * a.C * 2;
      for (int c = tid; c < a.C; c += 64) {
        float s0 = 0.f, s1 = 0.f;
        for (int rr = 0; rr < nrows; ++rr) {
          s0 += lred[rr * C2 + c * 2 + 0];
          s1 += lred[rr * C2 + c * 2 + 1];
        }
        const long rec = ((long)n * a.C + c) * 2;
        if (MODE == 0) {
          const double cnt = (double)(v1 - v0);
          const double k = (double)(float)a.x[((long)n * a.V + v0) * a.ldx + c];
          fx_add(a.acc, rec, nrec, blockIdx.x, (double)s0 + cnt * k);
          fx_add(a.acc, rec + 1, nrec, blockIdx.x, (double)s1 + 2.0 * k * (double)s0 + cnt * k * k);
        } else {
          fx_add(a.acc, rec, nrec, blockIdx.x, (double)s0);
          fx_add(a.acc, rec + 1, nrec, blockIdx.x, (double)s1);
        }
      }
      if (!last_workgroup_wave(a.counter, gridDim.x * gridDim.y)) return;
      const double V = (double)a.V;
      if (MODE == 0) {
        for (int i = tid; i < a.N * a.C; i += 64) {
          double mom[2];
          fx_take_n<2>(a.acc, (long)i * 2, nrec, mom);
          const double sx = mom[0], sq = mom[1];
          if (a.sums) {
            a.sums[(long)i * 2 + 0] = (float)sx;
            a.sums[(long)i * 2 + 1] = (float)sq;
          }
          if (a.nstat) {
            const double mean = sx / V;
            double var = sq / V - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float rs = (float)(1.0 / sqrt(var + (double)a.eps));
            const int c = i % a.C;
            const float sc = rs * a.gamma[c];
            const f32x4 o = {(float)mean, rs, sc, a.beta[c] - (float)mean * sc};
            *reinterpret_cast<f32x4*>(a.nstat + (long)i * 4) = o;
          }
        }
      } else {
        // per sample the two means the apply pass needs; over the batch (fixed order) the affine's gradients.  One lane
        // per (sample, channel) takes both sums in ONE round trip; the sums over the batch go through LDS (the reduction
        // slab is free now) when they fit, else through a per-channel loop.
        double* sred = reinterpret_cast<double*>(lred);
        const bool via_lds = (size_t)a.N * a.C * 2 * sizeof(double) <= sizeof(float) * (size_t)rows * C2;
        wave_lds_sync();
        if (via_lds) {
          for (int i = tid; i < a.N * a.C; i += 64) {
            double r[2];
            fx_take_n<2>(a.acc, (long)i * 2, nrec, r);
            a.nred[(long)i * 2 + 0] = (float)(r[0] / V);
            a.nred[(long)i * 2 + 1] = (float)(r[1] / V);
            sred[(long)i * 2 + 0] = r[0];
            sred[(long)i * 2 + 1] = r[1];
          }
          wave_lds_sync();
          if (a.dgamma)
            for (int c = tid; c < a.C; c += 64) {
              double sg = 0.0, sb = 0.0;
              for (int nn = 0; nn < a.N; ++nn) {
                sb += sred[((long)nn * a.C + c) * 2 + 0];
                sg += sred[((long)nn * a.C + c) * 2 + 1];
              }
              a.dgamma[c] = (float)sg;
              a.dbeta[c] = (float)sb;
            }
        } else {
          for (int c = tid; c < a.C; c += 64) {
            double sg = 0.0, sb = 0.0;
            for (int nn = 0; nn < a.N; ++nn) {
              const long i = (long)nn * a.C + c;
              double r[2];
              fx_take_n<2>(a.acc, i * 2, nrec, r);
              a.nred[i * 2 + 0] = (float)(r[0] / V);
              a.nred[i * 2 + 1] = (float)(r[1] / V);
              sb += r[0];
              sg += r[1];
            }
            if (a.dgamma) {
              a.dgamma[c] = (float)sg;
              a.dbeta[c] = (float)sb;
            }
          }
        }
      }
      return;
    }
    float* dst = (MODE == 0 ? a.stats : a.red) + (long)n * C2;
    for (int i = tid; i < C2; i += 256) {
      float s = 0.f;
      for (int rr = 0; rr < rows; ++rr) s += lred[rr * C2 + i];
      atomicAdd(dst + i, s);
    }
  }
}

static int g_norm_blocks = 2048;   // nnz_norm_tuning(0, blocks): target number of workgroups per launch (A/B runs)
static int g_norm_blocks_red = 512;   // nnz_norm_tuning(1, blocks): the same for the reducing kernels (statistics, backward reduce)

template <int MODE>
static int launch_norm(NormArgs a, hipStream_t s, bool pre_zeroed = false) {
  if (a.C % 8 || a.C > 640 || a.C < 8) return NNZ_EINVAL;
  // ~2048 blocks (8 per CU) of >= 512 voxels; each block sweeps its voxel range with NRM_UNR loads in flight per lane
  // (tools/probes/norm_bw_probe.py on the 268 MB full-resolution tensor: the apply kernels gain 6-10 % from 4x more, smaller
  // workgroups - 5.05 -> 5.6 TB/s, torch's device copy does 5.2-5.4 - the reducing kernels lose 40 %: their per-block
  // LDS slab + atomics are a fixed cost)
  const long nblk = (MODE == 0 || MODE == 2) ? g_norm_blocks_red : 4L * g_norm_blocks;
  long vpb = (a.V * a.N + nblk - 1) / nblk;
  if (vpb < 512) vpb = 512;
  if (vpb > a.V) vpb = a.V;
  a.vpb = (int)vpb;
  const int gx = (int)((a.V + vpb - 1) / vpb);
  const int rows = 256 / (a.C >> 3);
  const size_t lds = (MODE == 0 || MODE == 2) ? sizeof(float) * rows * 2 * a.C : 0;
  if ((MODE == 0 || MODE == 2) && !pre_zeroed && !a.acc) {
    hipError_t e = nnz::zero_async(MODE == 0 ? a.stats : a.red, sizeof(float) * 2 * a.N * a.C, s);
    if (e != hipSuccess) return (int)e;
  }
  NNZ_LAUNCH(norm_kernel<MODE>, dim3(gx, a.N), dim3(256), lds, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_norm_tuning(int knob, int value) {
  if ((knob != 0 && knob != 1) || value < 64) return NNZ_EINVAL;
  (knob == 0 ? nnz::g_norm_blocks : nnz::g_norm_blocks_red) = value;
  return NNZ_OK;
}

extern "C" int nnz_instnorm_stats(const void* x, float* stats, int N, long V, int C, int ldx, int pre_zeroed,
                                  void* stream) {
  using namespace nnz;
  if (!x || !stats) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x;
  a.stats = stats;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx;
  return launch_norm<0>(a, (hipStream_t)stream, pre_zeroed != 0);
}

extern "C" int nnz_instnorm_lrelu_apply(const void* x, const float* stats, const float* gamma, const float* beta,
                                        void* y, int N, long V, int C, int ldx, int ldy, float eps, float slope,
                                        void* stream) {
  using namespace nnz;
  if (!x || !stats || !gamma || !beta || !y) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x;
  a.stats = const_cast<float*>(stats);
  a.gamma = gamma; a.beta = beta;
  a.y = (f16*)y;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx; a.ldy = ldy;
  a.eps = eps; a.slope = slope;
  return launch_norm<1>(a, (hipStream_t)stream);
}

extern "C" int nnz_instnorm_lrelu_bwd_reduce(const void* x, const void* g, const float* stats, const float* gamma,
                                             const float* beta, float* red, int N, long V, int C, int ldx, int ldg,
                                             float eps, float slope, int pre_zeroed, void* stream) {
  using namespace nnz;
  if (!x || !g || !stats || !gamma || !beta || !red) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x; a.g = (const f16*)g;
  a.stats = const_cast<float*>(stats);
  a.red = red;
  a.gamma = gamma; a.beta = beta;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx; a.ldg = ldg;
  a.eps = eps; a.slope = slope;
  return launch_norm<2>(a, (hipStream_t)stream, pre_zeroed != 0);
}

extern "C" int nnz_instnorm_lrelu_bwd_apply(const void* x, const void* g, const float* stats, const float* red,
                                            const float* gamma, const float* beta, void* dx, int N, long V, int C,
                                            int ldx, int ldg, int lddx, float eps, float slope, float* dgamma,
                                            float* dbeta, void* stream) {
  using namespace nnz;
  if (!x || !g || !stats || !red || !gamma || !beta || !dx || (!dgamma != !dbeta)) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x; a.g = (const f16*)g;
  a.stats = const_cast<float*>(stats);
  a.red = const_cast<float*>(red);
  a.gamma = gamma; a.beta = beta;
  a.y = (f16*)dx;
  a.dgamma = dgamma; a.dbeta = dbeta;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx; a.ldg = ldg; a.ldy = lddx;
  a.eps = eps; a.slope = slope;
  return launch_norm<3>(a, (hipStream_t)stream);
}


// ---- deterministic entry points (fixed-point accumulators; see common.hpp and NormArgs) ----------------------------------
// `acc`: N * C * 2 records of nnz_fxacc_bytes() bytes, `counter`: one 32-bit word; zero before the first launch, left zero.

// statistics of x: nstat[N][C][4] (needs gamma, beta, eps) and / or plain moments sums[N][C][2] = {sum, sumsq}
extern "C" int nnz_instnorm_stats_det(const void* x, int N, long V, int C, int ldx, void* acc, void* counter,
                                      const float* gamma, const float* beta, float eps, float* nstat, float* sums,
                                      void* stream) {
  using namespace nnz;
  if (!x || !acc || !counter || (!nstat && !sums) || (nstat && (!gamma || !beta))) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx;
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.nstat = nstat; a.sums = sums;
  return launch_norm<0>(a, (hipStream_t)stream, true);
}

// y = lrelu(x * nstat.scale + nstat.shift)
extern "C" int nnz_instnorm_lrelu_apply_tab(const void* x, const float* nstat, void* y, int N, long V, int C, int ldx,
                                            int ldy, float slope, void* stream) {
  using namespace nnz;
  if (!x || !nstat || !y) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x;
  a.nstat = const_cast<float*>(nstat);
  a.y = (f16*)y;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx; a.ldy = ldy;
  a.slope = slope;
  return launch_norm<1>(a, (hipStream_t)stream);
}

// backward of InstanceNorm(affine) + LeakyReLU in two launches: the reduction (nred, dgamma, dbeta) and the apply pass
extern "C" int nnz_instnorm_lrelu_bwd_tab(const void* x, const void* g, const float* nstat, void* acc, void* counter,
                                          float* nred, void* dx, int N, long V, int C, int ldx, int ldg, int lddx,
                                          float slope, float* dgamma, float* dbeta, void* stream) {
  using namespace nnz;
  if (!x || !g || !nstat || !acc || !counter || !nred || !dx || (!dgamma != !dbeta)) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x; a.g = (const f16*)g;
  a.nstat = const_cast<float*>(nstat);
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
  a.nred = nred;
  a.y = (f16*)dx;
  a.dgamma = dgamma; a.dbeta = dbeta;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx; a.ldg = ldg; a.ldy = lddx;
  a.slope = slope;
  const int rc = launch_norm<2>(a, (hipStream_t)stream, true);
  if (rc != NNZ_OK) return rc;
  a.acc = nullptr;
  return launch_norm<3>(a, (hipStream_t)stream);
}

// the apply launch alone: nred[N][C][2] already holds {mean g', mean g' xhat} (nnz_conv_tap_dgrad_normred wrote it from the
// epilogue of the convolution that produced g)
extern "C" int nnz_instnorm_lrelu_bwd_apply_tab(const void* x, const void* g, const float* nstat, const float* nred, void* dx,
                                                int N, long V, int C, int ldx, int ldg, int lddx, float slope,
                                                void* stream) {
  using namespace nnz;
  if (!x || !g || !nstat || !nred || !dx) return NNZ_EINVAL;
  NormArgs a = {};
  a.x = (const f16*)x; a.g = (const f16*)g;
  a.nstat = const_cast<float*>(nstat);
  a.nred = const_cast<float*>(nred);
  a.y = (f16*)dx;
  a.N = N; a.V = V; a.C = C; a.ldx = ldx; a.ldg = ldg; a.ldy = lddx;
  a.slope = slope;
  return launch_norm<3>(a, (hipStream_t)stream);
}

// BatchNorm (REBNCONV, nnuzoo_amd/rebnconv.py) in training mode: the per-sample sums of the conv epilogue -> the batch sums
// the apply kernel reads, and torch's running-estimate update (F.batch_norm: biased variance for the normalisation, unbiased
// n / (n - 1) for running_var, momentum-weighted) - ONE launch instead of ten element-wise ones per REBNCONV unit.
namespace nnz {
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ stats, int N, int C, float n,
                                                              float momentum, float* __restrict__ bstats,
                                                              float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f, q = 0.f;
  for (int i = 0; i < N; ++i) {
    s += stats[((long)i * C + c) * 2 + 0];
    q += stats[((long)i * C + c) * 2 + 1];
  }
  bstats[c * 2 + 0] = s;
  bstats[c * 2 + 1] = q;
  if (running_mean && running_var) {
    const float mean = s / n;
    float var = q / n - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float unb = n > 1.f ? n / (n - 1.f) : 1.f;
    running_mean[c] = running_mean[c] * (1.f - momentum) + mean * momentum;
    running_var[c] = running_var[c] * (1.f - momentum) + var * unb * momentum;
  }
}
}  // namespace nnz

extern "C" int nnz_bn_batch_stats_finish(const float* stats, int N, int C, float n, float momentum, float* bstats,
                                         float* running_mean, float* running_var, void* stream) {
  using namespace nnz;
  if (!stats || !bstats || N < 1 || C < 1 || !(n >= 1.f) || (running_mean == nullptr) != (running_var == nullptr))
    return NNZ_EINVAL;
  NNZ_LAUNCH(bn_stats_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, N, C, n, momentum,
             bstats, running_mean, running_var);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
