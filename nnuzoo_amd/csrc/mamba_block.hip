// The two element-wise pieces of the 1-D Mamba block around the selective scan (gfx950), forward and backward:
//   causal depthwise conv1d + SiLU   x (B, D, L) -> silu(bias[d] + sum_j w[d][j] * x[b, d, l - (W-1) + j])
//   SiLU gate                        out = y * z * sigmoid(z)
// Reference call sites: mamba_inner_ref (/root/reference/nnunetv2/nets/seg_mamba/selective_scan_interface.py:640-674:
// `causal_conv1d_fn(x, w, b, "silu")`, and `selective_scan_fn(..., z=z)` whose reference applies `out * F.silu(z)`,
// same file :140-148) and Mamba.forward (/root/reference/nnunetv2/nets/seg_mamba/mamba_simple.py:190-357, slow path
// :315-357: `self.act(self.conv1d(x)[..., :seqlen])`).  fp32 in / out like the reference's slow path.  HBM-bound.
#include "common.hpp"

namespace nnz {

constexpr int CC_MAXW = 8;
constexpr int CC_LPT = 8;  // outputs per thread (one pass over a row segment of 256 * 8 positions per block)

struct ConvArgs {
  const float* x;   // [B][D][L]
  const float* w;   // [D][W]
  const float* b;   // [D] or null
  float* y;         // forward output
  const float* dy;  // backward: gradient of the output
  float* dx;
  float* dw;        // [D][W], accumulated with atomics (zeroed by the launcher)
  float* db;        // [D] or null
  int B, D, L, W;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(256) void causal_conv1d_silu_fwd_kernel(ConvArgs a) {
  const int row = blockIdx.y;  // b * D + d
  const int d = row % a.D;
  const float* xr = a.x + (long)row * a.L;
  float* yr = a.y + (long)row * a.L;
  float w[CC_MAXW];
#pragma unroll
  for (int j = 0; j < CC_MAXW; ++j) w[j] = j < a.W ? a.w[d * a.W + j] : 0.f;
  const float bias = a.b ? a.b[d] : 0.f;
  const int l0 = (blockIdx.x * 256 + threadIdx.x) * CC_LPT;
  if (l0 >= a.L) return;
  // window x[l0 - (W-1) .. l0 + LPT - 1]
  float win[CC_LPT + CC_MAXW - 1];
#pragma unroll
  for (int i = 0; i < CC_LPT + CC_MAXW - 1; ++i) {
    const int l = l0 - (a.W - 1) + i;
    win[i] = (i < CC_LPT + a.W - 1 && l >= 0 && l < a.L) ? xr[l] : 0.f;
  }
#pragma unroll
  for (int o = 0; o < CC_LPT; ++o) {
    if (l0 + o >= a.L) break;
    float acc = bias;
#pragma unroll
    for (int j = 0; j < CC_MAXW; ++j)
      if (j < a.W) acc += w[j] * win[o + j];
    yr[l0 + o] = acc * sigmoidf_(acc);
  }
}

// dpre = dy * silu'(pre) recomputed from x; dx[l] = sum_j w[j] * dpre[l + (W-1) - j]; dw, db reduced per block.
__global__ __launch_bounds__(256) void causal_conv1d_silu_bwd_kernel(ConvArgs a) {
  __shared__ float red[4][CC_MAXW + 1];
  const int row = blockIdx.y;
  const int d = row % a.D;
  const float* xr = a.x + (long)row * a.L;
  const float* gr = a.dy + (long)row * a.L;
  float* dxr = a.dx + (long)row * a.L;
  float w[CC_MAXW];
#pragma unroll
  for (int j = 0; j < CC_MAXW; ++j) w[j] = j < a.W ? a.w[d * a.W + j] : 0.f;
  const float bias = a.b ? a.b[d] : 0.f;
  const int W1 = a.W - 1;
  const int l0 = (blockIdx.x * 256 + threadIdx.x) * CC_LPT;
  float dwl[CC_MAXW], dbl = 0.f;
#pragma unroll
  for (int j = 0; j < CC_MAXW; ++j) dwl[j] = 0.f;
  if (l0 < a.L) {
    // dpre is needed at l0 .. l0 + LPT - 1 + W1 (the outputs that read this thread's inputs); each dpre[l] needs
    // x[l - W1 .. l]: window x[l0 - W1 .. l0 + LPT - 1 + W1]
    float xw[CC_LPT + 2 * (CC_MAXW - 1)];
#pragma unroll
    for (int i = 0; i < CC_LPT + 2 * (CC_MAXW - 1); ++i) {
      const int l = l0 - W1 + i;
      xw[i] = (i < CC_LPT + 2 * W1 && l >= 0 && l < a.L) ? xr[l] : 0.f;
    }
    float dpre[CC_LPT + CC_MAXW - 1];
#pragma unroll
    for (int i = 0; i < CC_LPT + CC_MAXW - 1; ++i) {
      const int l = l0 + i;
      float v = 0.f;
      if (i < CC_LPT + W1 && l < a.L) {
        float pre = bias;
#pragma unroll
        for (int j = 0; j < CC_MAXW; ++j)
          if (j < a.W) pre += w[j] * xw[i + j];
        const float s = sigmoidf_(pre);
        v = gr[l] * (s + pre * s * (1.f - s));
      }
      dpre[i] = v;
    }
#pragma unroll
    for (int o = 0; o < CC_LPT; ++o) {
      if (l0 + o >= a.L) break;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < CC_MAXW; ++j)
        if (j < a.W) acc += w[j] * dpre[o + W1 - j];
      dxr[l0 + o] = acc;
      // this thread owns outputs l0 .. l0 + LPT - 1 for the weight / bias gradients
      dbl += dpre[o];
#pragma unroll
      for (int j = 0; j < CC_MAXW; ++j)
        if (j < a.W) dwl[j] += dpre[o] * xw[o + j];
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < CC_MAXW; ++j) {
    const float s = wave_sum(dwl[j]);
    if ((threadIdx.x & 63) == 0) red[wave][j] = s;
  }
  {
    const float s = wave_sum(dbl);
    if ((threadIdx.x & 63) == 0) red[wave][CC_MAXW] = s;
  }
  __syncthreads();
  if (threadIdx.x < a.W)
    atomicAdd(a.dw + d * a.W + threadIdx.x,
              red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (threadIdx.x == CC_MAXW && a.db)
    atomicAdd(a.db + d, red[0][CC_MAXW] + red[1][CC_MAXW] + red[2][CC_MAXW] + red[3][CC_MAXW]);
}

__global__ __launch_bounds__(256) void silu_gate_fwd_kernel(const float* __restrict__ y, const float* __restrict__ z,
                                                            float* __restrict__ out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float zz = z[i];
    out[i] = y[i] * zz * sigmoidf_(zz);
  }
}

__global__ __launch_bounds__(256) void silu_gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            const float* __restrict__ z, float* __restrict__ dy,
                                                            float* __restrict__ dz, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float zz = z[i], s = sigmoidf_(zz), g = dout[i];
    dy[i] = g * zz * s;
    dz[i] = g * y[i] * (s + zz * s * (1.f - s));
  }
}

}  // namespace nnz

extern "C" int nnz_causal_conv1d_silu_forward(const float* x, const float* w, const float* bias, float* y, int B, int D,
                                              int L, int W, void* stream) {
  using namespace nnz;
  if (!x || !w || !y || B < 1 || D < 1 || L < 1 || W < 1 || W > CC_MAXW) return NNZ_EINVAL;
  ConvArgs a = {};
  a.x = x; a.w = w; a.b = bias; a.y = y; a.B = B; a.D = D; a.L = L; a.W = W;
  const int gx = (L + 256 * CC_LPT - 1) / (256 * CC_LPT);
  NNZ_LAUNCH(causal_conv1d_silu_fwd_kernel, dim3(gx, B * D), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_causal_conv1d_silu_backward(const float* x, const float* w, const float* bias, const float* dy,
                                               float* dx, float* dw, float* dbias, int B, int D, int L, int W,
                                               void* stream) {
  using namespace nnz;
  if (!x || !w || !dy || !dx || !dw || B < 1 || D < 1 || L < 1 || W < 1 || W > CC_MAXW) return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = nnz::zero_async(dw, sizeof(float) * D * W, s);
  if (e != hipSuccess) return (int)e;
  if (dbias) {
    e = nnz::zero_async(dbias, sizeof(float) * D, s);
    if (e != hipSuccess) return (int)e;
  }
  ConvArgs a = {};
  a.x = x; a.w = w; a.b = bias; a.dy = dy; a.dx = dx; a.dw = dw; a.db = dbias; a.B = B; a.D = D; a.L = L; a.W = W;
  const int gx = (L + 256 * CC_LPT - 1) / (256 * CC_LPT);
  NNZ_LAUNCH(causal_conv1d_silu_bwd_kernel, dim3(gx, B * D), dim3(256), 0, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_silu_gate_forward(const float* y, const float* z, float* out, long n, void* stream) {
  using namespace nnz;
  if (!y || !z || !out || n < 1) return NNZ_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(silu_gate_fwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, y, z, out, n);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_silu_gate_backward(const float* dout, const float* y, const float* z, float* dy, float* dz, long n,
                                      void* stream) {
  using namespace nnz;
  if (!dout || !y || !z || !dy || !dz || n < 1) return NNZ_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(silu_gate_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dout, y, z, dy, dz, n);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
