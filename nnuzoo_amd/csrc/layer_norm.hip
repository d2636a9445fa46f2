// LayerNorm over the last (channel) dimension of token-major tensors, forward + backward, for the VSS / Swin blocks of
// the zoo nets (reference: nn.LayerNorm at m2net.py:101,521; ssnd2net.py; swt2net.py:630-660 - `ln_1`, `out_norm`,
// patch merge/expand norms).  C is small there (16 .. 1024) and the row count huge (512^2 tokens x batch), which the
// generic kernels handle poorly (one workgroup per 16-float row); here a row lives on LPR = 4 .. 64 lanes of a wave
// (16 bytes per lane per pass), several rows per wave, reductions by xor-shuffles inside the lane group - one HBM
// pass forward, one backward, with dgamma / dbeta folded into the backward pass (per-lane partials over the rows the
// lane group visits, one LDS fold per workgroup, one atomic per channel per workgroup).
#include "common.hpp"

namespace nnz {

struct LnArgs {
  const void* x;      // [R][C] f16 or f32
  const float* gamma; // [C] or null
  const float* beta;  // [C] or null
  void* y;            // [R][C] f32, or f16 when y_is_f16 (the consumer is an autocast Linear: same rounding, no cast pass)
  float* mean;        // [R]
  float* rstd;        // [R]
  const void* dy;     // [R][C] f32 or f16 (dy_is_f16)
  void* dx;           // [R][C] same type as x
  const void* dres;   // backward, optional: [R][C] gradient (type of x) of the residual stream the normalised tensor was taken from -
                      // dx = dres + (LayerNorm backward), the sum autograd would make with an add launch of its own
  float* dgamma;      // [C] atomic, zeroed by the launcher; may be null
  float* dbeta;
  long R;
  int C;
  float eps;
  // optional gate (SS2D: out_norm(y) * silu(z), m2net.py:220): y_out = LN(x) * silu(z); z rows have their own stride (z is
  // one half of the in_proj output), dz has z's type and is written densely [R][C]
  const void* z;
  void* dz;
  long z_stride;
  int z_is_f16;
  int y_is_f16, dy_is_f16;
  FxAcc* acc;       // backward, deterministic: [2][C] fixed-point accumulators (common.hpp) + launch counter; dgamma / dbeta are
  unsigned* counter;  // then WRITTEN by the launch's last workgroup (no zero fill, no float atomics)
  float* zero_buf;  // forward: [2][C] buffer the matching backward will accumulate dgamma / dbeta into - zeroed here, so the
                    // backward needs no zeroing launch of its own (may be null)
  // round 5, backward of the fused Swin block (fp32): `part` != null - every workgroup WRITES its dgamma | dbeta partial sums to
  // part[blockIdx.x][2 C] and returns (a fold job of the pass's grouped launch sums them in workgroup order: no fixed-point
  // adds, no last-workgroup tail per launch).  pad_h > 0: the R rows are the tokens of the top / left padded grid
  // (pad_h + pad_y) x (pad_w + pad_x) (dy, mean, rstd are laid out on it); x, dres and dx live on the unpadded [B][pad_h][pad_w]
  // grid - padded tokens read x = 0, contribute to dgamma / dbeta, and their dx is dropped (the crop's backward).
  float* part;
  int pad_h, pad_w, pad_y, pad_x;
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }

template <class T>
__device__ __forceinline__ f32x4 ld4(const T* p);
template <>
__device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 ld4<f16>(const f16* p) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(f16* p, f32x4 v) {
  *reinterpret_cast<f16x4*>(p) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
}

__device__ __forceinline__ f32x4 ldz(const LnArgs& a, long r, int c) {
  return a.z_is_f16 ? ld4((const f16*)a.z + r * a.z_stride + c) : ld4((const float*)a.z + r * a.z_stride + c);
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// a lane group of LPR lanes owns a row; lane q of the group holds the channels 4(q + LPR i) .. +3, i < IT
template <class T, int LPR, int IT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs a) {
  constexpr int GPW = 256 / LPR;  // lane groups (rows in flight) per workgroup
  const int q = threadIdx.x % LPR, g = threadIdx.x / LPR;
  const int C = a.C;
  const float invC = 1.f / (float)C;
  f32x4 gm[IT], bt[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = 4 * (q + LPR * i);
    gm[i] = (a.gamma && c < C) ? ld4(a.gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    bt[i] = (a.beta && c < C) ? ld4(a.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.zero_buf && blockIdx.x == 0)
    for (int c = threadIdx.x; c < 2 * C; c += 256) a.zero_buf[c] = 0.f;
  for (long r = (long)blockIdx.x * GPW + g; r < a.R; r += (long)gridDim.x * GPW) {
    const T* xr = (const T*)a.x + r * C;
    f32x4 v[IT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = 4 * (q + LPR * i);
      v[i] = c < C ? ld4(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = group_sum<LPR>(s) * invC;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = 4 * (q + LPR * i);
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mu;
          ss += d * d;
        }
      }
    }
    const float rs = 1.f / sqrtf(group_sum<LPR>(ss) * invC + a.eps);
    const long yo = r * C;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = 4 * (q + LPR * i);
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * gm[i][e] + bt[i][e];
        if (a.z) {
          const f32x4 zv = ldz(a, r, c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= silu_f(zv[e]);
        }
        if (a.y_is_f16) st4((f16*)a.y + yo + c, o);
        else st4((float*)a.y + yo + c, o);
      }
    }
    if (q == 0) {
      a.mean[r] = mu;
      a.rstd[r] = rs;
    }
  }
}

template <class T, int LPR, int IT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnArgs a) {
  constexpr int GPW = 256 / LPR;
  __shared__ float fold[2][GPW][LPR * 4 * IT];
  const int q = threadIdx.x % LPR, g = threadIdx.x / LPR;
  const int C = a.C;
  const float invC = 1.f / (float)C;
  f32x4 gm[IT], bt[IT], dg[IT], db[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = 4 * (q + LPR * i);
    gm[i] = (a.gamma && c < C) ? ld4(a.gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    bt[i] = (a.z && a.beta && c < C) ? ld4(a.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long r = (long)blockIdx.x * GPW + g; r < a.R; r += (long)gridDim.x * GPW) {
    long sr = r;                                  // row of x / dres / dx
    if (a.pad_h > 0) {
      const int hp = a.pad_h + a.pad_y, wp = a.pad_w + a.pad_x;
      const int b = (int)(r / (hp * wp)), rem = (int)(r - (long)b * (hp * wp));
      const int y = rem / wp, x = rem - y * wp;
      sr = (y >= a.pad_y && x >= a.pad_x) ? ((long)b * a.pad_h + (y - a.pad_y)) * a.pad_w + (x - a.pad_x) : -1;
    }
    const T* xr = (const T*)a.x + (sr < 0 ? 0 : sr) * C;
    const long dyo = r * C;
    const float mu = a.mean[r], rs = a.rstd[r];
    f32x4 xh[IT], gy[IT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = 4 * (q + LPR * i);
      if (c < C) {
        const f32x4 xv = sr < 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : ld4(xr + c);
        f32x4 dv = a.dy_is_f16 ? ld4((const f16*)a.dy + dyo + c) : ld4((const float*)a.dy + dyo + c);
        if (a.z) {
          // out = n * silu(z), n = xhat * gamma + beta:  dn = dout * silu(z);  dz = dout * n * silu'(z)
          const f32x4 zv = ldz(a, r, c);
          f32x4 dzv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sg = 1.f / (1.f + __expf(-zv[e]));
            const float n = (xv[e] - mu) * rs * gm[i][e] + bt[i][e];
            dzv[e] = dv[e] * n * sg * (1.f + zv[e] * (1.f - sg));
            dv[e] *= zv[e] * sg;
          }
          if (a.z_is_f16) st4((f16*)a.dz + r * C + c, dzv);
          else st4((float*)a.dz + r * C + c, dzv);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          gy[i][e] = dv[e] * gm[i][e];
          s1 += gy[i][e];
          s2 += gy[i][e] * xh[i][e];
          dg[i][e] += dv[e] * xh[i][e];
          db[i][e] += dv[e];
        }
      } else {
        xh[i] = gy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float m1 = group_sum<LPR>(s1) * invC, m2 = group_sum<LPR>(s2) * invC;
    if (sr < 0) continue;                         // a padded token: its dx is cropped away
    T* dxr = (T*)a.dx + sr * C;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = 4 * (q + LPR * i);
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - m1 - xh[i][e] * m2);
        if (a.dres) {
          const f32x4 sk = ld4((const T*)a.dres + sr * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = sk[e] + o[e];
        }
        st4(dxr + c, o);
      }
    }
  }
  if (!a.dgamma && !a.dbeta && !a.part) return;
  // fold the lane groups' partials: fold[.][g][channel], then thread c sums over g
#pragma unroll
  for (int i = 0; i < IT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * (q + LPR * i) + e;
      fold[0][g][c] = dg[i][e];
      fold[1][g][c] = db[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg = 0.f, sb = 0.f;
    for (int gg = 0; gg < GPW; ++gg) {
      sg += fold[0][gg][c];
      sb += fold[1][gg][c];
    }
    if (a.part) {
      a.part[(long)blockIdx.x * 2 * C + c] = sg;
      a.part[(long)blockIdx.x * 2 * C + C + c] = sb;
    } else if (a.acc) {
      fx_add(a.acc, c, 2L * C, blockIdx.x, (double)sg);
      fx_add(a.acc, (long)C + c, 2L * C, blockIdx.x, (double)sb);
    } else {
      if (a.dgamma) atomicAdd(a.dgamma + c, sg);
      if (a.dbeta) atomicAdd(a.dbeta + c, sb);
    }
  }
  if (a.acc && last_workgroup(a.counter, gridDim.x))
    for (int c = threadIdx.x; c < C; c += 256) {
      const double g = fx_take(a.acc, c, 2L * C), b = fx_take(a.acc, (long)C + c, 2L * C);
      if (a.dgamma) a.dgamma[c] = (float)g;
      if (a.dbeta) a.dbeta[c] = (float)b;
    }
}

static int ln_lpr(int C) { return C <= 16 ? 4 : C <= 32 ? 8 : C <= 64 ? 16 : C <= 128 ? 32 : 64; }
static long ln_bwd_cap() {
  static const long cap = [] { const char* e = getenv("NNZ_LN_BWD_CAP"); return e && atol(e) > 0 ? atol(e) : 256L; }();
  return cap;
}
// workgroups of a backward launch over `rows` rows of C channels (= the number of dgamma | dbeta partials of the `part` mode)
static long ln_bwd_grid(long rows, int C) {
  const int gpw = 256 / ln_lpr(C);
  const long want = (rows + gpw - 1) / gpw;
  return want < ln_bwd_cap() ? (want < 1 ? 1 : want) : ln_bwd_cap();
}

template <class T, int LPR, int IT>
static int ln_launch(const LnArgs& a, bool bwd, hipStream_t s) {
  constexpr int GPW = 256 / LPR;
  long want = (a.R + GPW - 1) / GPW;
  // forward: enough workgroups to fill the chip several times; backward: one workgroup per CU, grid-striding over the rows -
  // every workgroup ends with 2 C fixed-point adds and the last one waits for all of them, so the tail grows with the grid
  // (SwT2Net step at 1024 / 512 / 256 / 128 / 64 workgroups: 84.1 / 82.6 / 81.6 / 81.3 / 82.5 ms, M2Net 87.8 / 86.9 / 86.8 /
  // 88.4 / 91.5 ms; NNZ_LN_BWD_CAP overrides)
  const long cap = bwd ? ln_bwd_cap() : 8192;
  const unsigned grid = (unsigned)(want < cap ? (want < 1 ? 1 : want) : cap);
  if (bwd)
    NNZ_LAUNCH((ln_bwd_kernel<T, LPR, IT>), dim3(grid), dim3(256), 0, s, a);
  else
    NNZ_LAUNCH((ln_fwd_kernel<T, LPR, IT>), dim3(grid), dim3(256), 0, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <class T>
static int ln_dispatch(const LnArgs& a, bool bwd, hipStream_t s) {
  const int C = a.C;
  if (C <= 16) return ln_launch<T, 4, 1>(a, bwd, s);
  if (C <= 32) return ln_launch<T, 8, 1>(a, bwd, s);
  if (C <= 64) return ln_launch<T, 16, 1>(a, bwd, s);
  if (C <= 128) return ln_launch<T, 32, 1>(a, bwd, s);
  if (C <= 256) return ln_launch<T, 64, 1>(a, bwd, s);
  if (C <= 512) return ln_launch<T, 64, 2>(a, bwd, s);
  if (C <= 1024) return ln_launch<T, 64, 4>(a, bwd, s);
  if (C <= 2048) return ln_launch<T, 64, 8>(a, bwd, s);
  return NNZ_EINVAL;
}

}  // namespace nnz

static int ln_forward_impl(const void* x, int x_is_f16, const float* gamma, const float* beta, const void* z,
                           int z_is_f16, long z_stride, void* y, int y_is_f16, float* mean, float* rstd, float* zero_2c,
                           long rows, int C, float eps, void* stream) {
  using namespace nnz;
  if (!x || !y || !mean || !rstd || rows < 0 || C < 4 || (C & 3) || C > 2048 || (z && (z_stride & 3))) return NNZ_EINVAL;
  if (rows == 0) return NNZ_OK;
  LnArgs a = {};
  a.x = x; a.gamma = gamma; a.beta = beta; a.y = y; a.mean = mean; a.rstd = rstd; a.R = rows; a.C = C; a.eps = eps;
  a.z = z; a.z_is_f16 = z_is_f16; a.z_stride = z_stride; a.y_is_f16 = y_is_f16; a.zero_buf = zero_2c;
  return x_is_f16 ? ln_dispatch<f16>(a, false, (hipStream_t)stream) : ln_dispatch<float>(a, false, (hipStream_t)stream);
}

static int ln_backward_impl(const void* x, int x_is_f16, const float* gamma, const float* beta, const void* z,
                            int z_is_f16, long z_stride, const float* mean, const float* rstd, const void* dy,
                            int dy_is_f16, void* dx, void* dz, float* dgamma, float* dbeta, int pre_zeroed, long rows,
                            int C, void* stream, void* acc = nullptr, void* counter = nullptr,
                            const void* dres = nullptr) {
  using namespace nnz;
  if (acc) pre_zeroed = 1;   // dgamma / dbeta are written, not accumulated
  if (!x || !mean || !rstd || !dy || !dx || rows < 0 || C < 4 || (C & 3) || C > 2048 || (z && (!dz || (z_stride & 3))))
    return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (pre_zeroed) {  // the forward launch zeroed the [2][C] buffer
  } else if (dgamma && dbeta == dgamma + C) {  // one buffer [2][C]: one launch
    if ((e = zero_async(dgamma, sizeof(float) * 2 * C, s)) != hipSuccess) return (int)e;
  } else {
    if (dgamma && (e = zero_async(dgamma, sizeof(float) * C, s)) != hipSuccess) return (int)e;
    if (dbeta && (e = zero_async(dbeta, sizeof(float) * C, s)) != hipSuccess) return (int)e;
  }
  if (rows == 0) return NNZ_OK;
  LnArgs a = {};
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = (float*)mean; a.rstd = (float*)rstd; a.dy = dy; a.dx = dx;
  a.dgamma = dgamma; a.dbeta = dbeta; a.R = rows; a.C = C;
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter; a.dres = dres;
  a.z = z; a.dz = dz; a.z_is_f16 = z_is_f16; a.z_stride = z_stride; a.dy_is_f16 = dy_is_f16;
  return x_is_f16 ? ln_dispatch<f16>(a, true, s) : ln_dispatch<float>(a, true, s);
}

extern "C" int nnz_layer_norm_gate_forward(const void* x, int x_is_f16, const float* gamma, const float* beta,
                                           const void* z, int z_is_f16, long z_row_stride, void* y, int y_is_f16,
                                           float* mean, float* rstd, float* zero_2c, long rows, int C, float eps,
                                           void* stream) {
  if (!z) return NNZ_EINVAL;
  return ln_forward_impl(x, x_is_f16, gamma, beta, z, z_is_f16, z_row_stride, y, y_is_f16, mean, rstd, zero_2c, rows, C,
                         eps, stream);
}

extern "C" int nnz_layer_norm_gate_backward(const void* x, int x_is_f16, const float* gamma, const float* beta,
                                            const void* z, int z_is_f16, long z_row_stride, const float* mean,
                                            const float* rstd, const void* dy, int dy_is_f16, void* dx, void* dz,
                                            float* dgamma, float* dbeta, int pre_zeroed, long rows, int C,
                                            void* stream) {
  if (!z) return NNZ_EINVAL;
  return ln_backward_impl(x, x_is_f16, gamma, beta, z, z_is_f16, z_row_stride, mean, rstd, dy, dy_is_f16, dx, dz, dgamma,
                          dbeta, pre_zeroed, rows, C, stream);
}

extern "C" int nnz_layer_norm_forward(const void* x, int x_is_f16, const float* gamma, const float* beta, void* y,
                                      int y_is_f16, float* mean, float* rstd, float* zero_2c, long rows, int C,
                                      float eps, void* stream) {
  return ln_forward_impl(x, x_is_f16, gamma, beta, nullptr, 0, 0, y, y_is_f16, mean, rstd, zero_2c, rows, C, eps, stream);
}

extern "C" int nnz_layer_norm_backward(const void* x, int x_is_f16, const float* gamma, const float* mean,
                                       const float* rstd, const void* dy, int dy_is_f16, void* dx, float* dgamma,
                                       float* dbeta, int pre_zeroed, long rows, int C, void* stream) {
  return ln_backward_impl(x, x_is_f16, gamma, nullptr, nullptr, 0, 0, mean, rstd, dy, dy_is_f16, dx, nullptr, dgamma,
                          dbeta, pre_zeroed, rows, C, stream);
}

// deterministic variants: acc = 2 * C zeroed records of nnz_fxacc_bytes() bytes, counter = one zeroed 32-bit word (both left
// zero); dgamma / dbeta are written by the launch's last workgroup - bit-identical run to run, no zero fill needed
extern "C" int nnz_layer_norm_backward_det(const void* x, int x_is_f16, const float* gamma, const float* mean,
                                           const float* rstd, const void* dy, int dy_is_f16, void* dx, float* dgamma,
                                           float* dbeta, void* acc, void* counter, long rows, int C, void* stream) {
  if (!acc || !counter) return NNZ_EINVAL;
  return ln_backward_impl(x, x_is_f16, gamma, nullptr, nullptr, 0, 0, mean, rstd, dy, dy_is_f16, dx, nullptr, dgamma,
                          dbeta, 1, rows, C, stream, acc, counter);
}
extern "C" int nnz_layer_norm_gate_backward_det(const void* x, int x_is_f16, const float* gamma, const float* beta,
                                                const void* z, int z_is_f16, long z_row_stride, const float* mean,
                                                const float* rstd, const void* dy, int dy_is_f16, void* dx, void* dz,
                                                float* dgamma, float* dbeta, void* acc, void* counter, long rows, int C,
                                                void* stream) {
  if (!z || !acc || !counter) return NNZ_EINVAL;
  return ln_backward_impl(x, x_is_f16, gamma, beta, z, z_is_f16, z_row_stride, mean, rstd, dy, dy_is_f16, dx, dz, dgamma,
                          dbeta, 1, rows, C, stream, acc, counter);
}

// nnz_layer_norm_backward_det with the residual stream's gradient added into dx (dres: [rows][C] of x's type, may alias nothing
// else; null = the plain backward): x -> (LayerNorm(x), x) is how every Swin / VSS block uses its norm, and the two
// gradients of x meet here instead of in an add launch
extern "C" int nnz_layer_norm_backward_det_res(const void* x, int x_is_f16, const float* gamma, const float* mean,
                                               const float* rstd, const void* dy, int dy_is_f16, const void* dres,
                                               void* dx, float* dgamma, float* dbeta, void* acc, void* counter,
                                               long rows, int C, void* stream) {
  if (!acc || !counter) return NNZ_EINVAL;
  return ln_backward_impl(x, x_is_f16, gamma, nullptr, nullptr, 0, 0, mean, rstd, dy, dy_is_f16, dx, nullptr, dgamma,
                          dbeta, 1, rows, C, stream, acc, counter, dres);
}

// ---- round 5: the LayerNorm backward of the fused Swin block (fp32) ---------------------------------------------------------
// dx = dres + LayerNorm-backward(dy) like nnz_layer_norm_backward_det_res, but (1) dgamma | dbeta leave the launch as
// per-workgroup partials part[nnz_layer_norm_backward_parts(rows, C)][2 C], to be summed in workgroup order by a fold job of the
// backward pass's grouped launch (nnz_dense32_group_fill_fold), and (2) with pad_h > 0 the rows are the tokens of the block's
// top / left padded grid (dy / mean / rstd on it) while x / dres / dx live on the unpadded [B][pad_h][pad_w] grid.
extern "C" long nnz_layer_norm_backward_parts(long rows, int C) {
  if (rows < 1 || C < 4 || C > 2048) return 0;
  return nnz::ln_bwd_grid(rows, C);
}
extern "C" int nnz_layer_norm_backward_partial(const float* x, const float* gamma, const float* mean, const float* rstd,
                                               const float* dy, const float* dres, float* dx, float* part, long rows, int C,
                                               int pad_h, int pad_w, int pad_y, int pad_x, void* stream) {
  using namespace nnz;
  if (!x || !mean || !rstd || !dy || !dx || !part || rows < 1 || C < 4 || (C & 3) || C > 2048) return NNZ_EINVAL;
  if (pad_h > 0 && (pad_w < 1 || pad_y < 0 || pad_x < 0 || rows % ((long)(pad_h + pad_y) * (pad_w + pad_x)))) return NNZ_EINVAL;
  LnArgs a = {};
  a.x = x; a.gamma = gamma; a.mean = (float*)mean; a.rstd = (float*)rstd; a.dy = dy; a.dx = dx; a.dres = dres; a.part = part;
  a.R = rows; a.C = C; a.pad_h = pad_h; a.pad_w = pad_w; a.pad_y = pad_y; a.pad_x = pad_x;
  return ln_dispatch<float>(a, true, (hipStream_t)stream);
}
