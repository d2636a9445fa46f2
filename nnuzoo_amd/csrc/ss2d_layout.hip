// Layout kernels around the cross-scan mode of the selective scan (selective_scan.hip, ScanArgs::xs_*): together they
// replace the ~60 element-wise / copy launches per SS2D block with which the reference builds and undoes its four scan
// directions (m2net.py:170-206: stack of x and x^T, flip, cat; after the scan flip back, transpose back, 3 adds, and the
// permute to token-major) and their autograd mirrors.
//   prepare  : x (B, D, H, W) f16|f32        -> x2 [2][B][D][L] f32   (row-major tokens | column-major tokens)
//   merge    : y [B][4][D][L] (per direction, source order) -> out (B, H, W, D) f32 = y0 + y2 + (y1 + y3)^T
//   split    : dout (B, H, W, D) f32         -> dy2 [2][B][D][L]      (the gradient every direction of a source sees)
//   merge_dx : du [B][4][D][L] (optional) + dx2 [2][B][D][L] -> dx (B, D, H, W) in x's type
// All four are tiled transposes through LDS: every global access is a run of >= 16 consecutive floats.
#include "common.hpp"

namespace nnz {

constexpr int XT = 32;  // plane tile edge (prepare / merge_dx)

template <class T>
__global__ __launch_bounds__(256) void ss2d_prepare_kernel(const T* __restrict__ x, float* __restrict__ x2, int H, int W,
                                                           long plane_count) {
  __shared__ float tile[XT][XT + 1];
  const long plane = blockIdx.y;  // b * D + d
  const int tiles_w = (W + XT - 1) / XT;
  const int h0 = (blockIdx.x / tiles_w) * XT, w0 = (blockIdx.x % tiles_w) * XT;
  const long L = (long)H * W;
  const T* src = x + plane * L;
  float* rm = x2 + plane * L;
  float* cm = x2 + (plane_count + plane) * L;
  const int tx = threadIdx.x % XT, ty = threadIdx.x / XT;  // 32 x 8
#pragma unroll
  for (int i = 0; i < XT; i += 8) {
    const int h = h0 + ty + i, w = w0 + tx;
    float v = 0.f;
    if (h < H && w < W) {
      v = (float)src[(long)h * W + w];
      rm[(long)h * W + w] = v;
    }
    tile[ty + i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < XT; i += 8) {
    const int w = w0 + ty + i, h = h0 + tx;
    if (h < H && w < W) cm[(long)w * H + h] = tile[tx][ty + i];
  }
}

// dx[plane][h][w] = a0 + a1 + a2 at [h W + w]  +  (c0 + c1 + c2) at [w H + h]
template <class T>
__global__ __launch_bounds__(256) void ss2d_merge_dx_kernel(const float* __restrict__ du, const float* __restrict__ dx2,
                                                            T* __restrict__ dx, int D, int H, int W, long plane_count) {
  __shared__ float tile[XT][XT + 1];
  const long plane = blockIdx.y;
  const int b = plane / D, d = plane % D;
  const int tiles_w = (W + XT - 1) / XT;
  const int h0 = (blockIdx.x / tiles_w) * XT, w0 = (blockIdx.x % tiles_w) * XT;
  const long L = (long)H * W;
  const float* u0 = du + (((long)b * 4 + 0) * D + d) * L;
  const float* u1 = du + (((long)b * 4 + 1) * D + d) * L;
  const float* u2 = du + (((long)b * 4 + 2) * D + d) * L;
  const float* u3 = du + (((long)b * 4 + 3) * D + d) * L;
  const float* g0 = dx2 + plane * L;
  const float* g1 = dx2 + (plane_count + plane) * L;
  const int tx = threadIdx.x % XT, ty = threadIdx.x / XT;
#pragma unroll
  for (int i = 0; i < XT; i += 8) {
    const int w = w0 + ty + i, h = h0 + tx;
    float v = 0.f;
    if (h < H && w < W) {
      const long m = (long)w * H + h;
      v = g1[m];
      if (du) v += u1[m] + u3[m];
    }
    tile[tx][ty + i] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < XT; i += 8) {
    const int h = h0 + ty + i, w = w0 + tx;
    if (h < H && w < W) {
      const long m = (long)h * W + w;
      float v = tile[ty + i][tx] + g0[m];
      if (du) v += u0[m] + u2[m];
      dx[plane * L + m] = (T)v;
    }
  }
}

// token-major <-> per-direction channel-major: tile of 16 x 16 tokens x 16 channels
constexpr int MT = 16;
constexpr int MPITCH = MT * (MT + 1) + 1;  // channel pitch: distinct banks for the 16 channels of a token

// SPLIT = false: out[b][h][w][d] = y0 + y2 + (y1 + y3)^T;  SPLIT = true: dy2[0] = dout^T(tokens->channels), dy2[1] likewise
template <bool SPLIT>
__global__ __launch_bounds__(256) void ss2d_merge_kernel(const float* __restrict__ y, float* __restrict__ tok,
                                                         float* __restrict__ dy2, int Bt, int D, int H, int W) {
  __shared__ float T[MT * MPITCH];
  const int tiles_w = (W + MT - 1) / MT;
  const int h0 = (blockIdx.x / tiles_w) * MT, w0 = (blockIdx.x % tiles_w) * MT;
  const int d0 = blockIdx.y * MT, b = blockIdx.z;
  const long L = (long)H * W;
  const int lo = threadIdx.x % MT, hi = threadIdx.x / MT;
  if (!SPLIT) {
    // row-major sources: lanes along w
    for (int d = 0; d < MT; ++d) {
      const int h = h0 + hi, w = w0 + lo;
      float v = 0.f;
      if (h < H && w < W && d0 + d < D) {
        const long m = (long)h * W + w;
        v = y[(((long)b * 4 + 0) * D + d0 + d) * L + m] + y[(((long)b * 4 + 2) * D + d0 + d) * L + m];
      }
      T[d * MPITCH + hi * (MT + 1) + lo] = v;
    }
    __syncthreads();
    // column-major sources: lanes along h
    for (int d = 0; d < MT; ++d) {
      const int w = w0 + hi, h = h0 + lo;
      if (h < H && w < W && d0 + d < D) {
        const long m = (long)w * H + h;
        T[d * MPITCH + lo * (MT + 1) + hi] +=
            y[(((long)b * 4 + 1) * D + d0 + d) * L + m] + y[(((long)b * 4 + 3) * D + d0 + d) * L + m];
      }
    }
    __syncthreads();
    // token-major store: lanes along d
    for (int tkn = hi; tkn < MT * MT; tkn += 256 / MT) {
      const int th = tkn / MT, tw = tkn % MT;
      const int h = h0 + th, w = w0 + tw;
      if (h < H && w < W && d0 + lo < D) tok[((long)b * L + (long)h * W + w) * D + d0 + lo] = T[lo * MPITCH + th * (MT + 1) + tw];
    }
  } else {
    for (int tkn = hi; tkn < MT * MT; tkn += 256 / MT) {
      const int th = tkn / MT, tw = tkn % MT;
      const int h = h0 + th, w = w0 + tw;
      float v = 0.f;
      if (h < H && w < W && d0 + lo < D) v = tok[((long)b * L + (long)h * W + w) * D + d0 + lo];
      T[lo * MPITCH + th * (MT + 1) + tw] = v;
    }
    __syncthreads();
    const long src1 = (long)Bt * D * L;
    for (int d = 0; d < MT; ++d) {
      if (d0 + d >= D) break;
      float* rm = dy2 + ((long)b * D + d0 + d) * L;
      {
        const int h = h0 + hi, w = w0 + lo;
        if (h < H && w < W) rm[(long)h * W + w] = T[d * MPITCH + hi * (MT + 1) + lo];
      }
      {
        const int w = w0 + hi, h = h0 + lo;
        if (h < H && w < W) rm[src1 + (long)w * H + h] = T[d * MPITCH + lo * (MT + 1) + hi];
      }
    }
  }
}

}  // namespace nnz

extern "C" int nnz_ss2d_prepare(const void* x, int x_is_f16, float* x2, int Bt, int D, int H, int W, void* stream) {
  using namespace nnz;
  if (!x || !x2 || Bt < 1 || D < 1 || H < 1 || W < 1 || (long)Bt * D > 65535) return NNZ_EINVAL;
  const long planes = (long)Bt * D;
  dim3 grid(((H + XT - 1) / XT) * ((W + XT - 1) / XT), (unsigned)planes);
  if (x_is_f16)
    NNZ_LAUNCH(ss2d_prepare_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, x2, H, W, planes);
  else
    NNZ_LAUNCH(ss2d_prepare_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, x2, H, W,
                       planes);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_ss2d_merge_dx(const float* du, const float* dx2, void* dx, int dx_is_f16, int Bt, int D, int H, int W,
                                 void* stream) {
  using namespace nnz;
  if (!dx2 || !dx || Bt < 1 || D < 1 || H < 1 || W < 1 || (long)Bt * D > 65535) return NNZ_EINVAL;
  const long planes = (long)Bt * D;
  dim3 grid(((H + XT - 1) / XT) * ((W + XT - 1) / XT), (unsigned)planes);
  if (dx_is_f16)
    NNZ_LAUNCH(ss2d_merge_dx_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, du, dx2, (f16*)dx, D, H, W,
                       planes);
  else
    NNZ_LAUNCH(ss2d_merge_dx_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, du, dx2, (float*)dx, D, H, W,
                       planes);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_ss2d_merge(const float* y, float* out_tokens, int Bt, int D, int H, int W, void* stream) {
  using namespace nnz;
  if (!y || !out_tokens || Bt < 1 || D < 1 || H < 1 || W < 1 || Bt > 65535 || (D + MT - 1) / MT > 65535)
    return NNZ_EINVAL;
  dim3 grid(((H + MT - 1) / MT) * ((W + MT - 1) / MT), (D + MT - 1) / MT, Bt);
  NNZ_LAUNCH(ss2d_merge_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, y, out_tokens, (float*)nullptr,
                     Bt, D, H, W);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_ss2d_split(const float* dout_tokens, float* dy2, int Bt, int D, int H, int W, void* stream) {
  using namespace nnz;
  if (!dout_tokens || !dy2 || Bt < 1 || D < 1 || H < 1 || W < 1 || Bt > 65535 || (D + MT - 1) / MT > 65535)
    return NNZ_EINVAL;
  dim3 grid(((H + MT - 1) / MT) * ((W + MT - 1) / MT), (D + MT - 1) / MT, Bt);
  NNZ_LAUNCH(ss2d_merge_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                     const_cast<float*>(dout_tokens), dy2, Bt, D, H, W);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
