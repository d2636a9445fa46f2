// Depthwise 3x3 conv + SiLU of the SS2D block, fused with the layout change the scan needs.
// Reference (m2net.py:212-216): x, z = in_proj(x).chunk(2); x = act(conv2d(x.permute(0, 3, 1, 2).contiguous())) with a
// depthwise Conv2d(Di, Di, 3, padding 1, groups Di) - a permute copy, a library conv, a SiLU pass, and then (in
// forward_core) the transposed copy.  Here one kernel reads the token-major half of the in_proj output in place (row
// stride 2 Di) and writes both scan sources x2[0] (row-major tokens) and x2[1] (column-major tokens) in fp32; one
// backward kernel takes the gradient of x2, recomputes the pre-activation, and produces the token-major input gradient,
// dweight and dbias.  Tiles of 16 x 16 tokens x 16 channels through LDS; every global access is a run of >= 16 elements.
#include "common.hpp"

namespace nnz {

constexpr int DT = 16;            // tokens per tile edge
constexpr int DC = 16;            // channels per tile
constexpr int DPAD = 8;           // row padding (floats) of the token-major LDS images

struct DwArgs {
  const void* x;      // token-major [B][H][W][.] rows x_stride elements apart, f16 or f32; channels [0, D)
  const float* w;     // [D][3][3]
  const float* bias;  // [D] or null
  float* x2;          // [2][B][D][L]
  const float* dx2;   // [2][B][D][L]   (backward)
  void* dx;           // [B][H][W][D] dense, x's type
  float* dw;          // [D][9] atomic
  float* dbias;       // [D] atomic, may be null
  long x_stride;
  int x_is_f16;
  int B, D, H, W;
  float* part;        // null, or [B * tiles][D * 10]: two-stage (deterministic) weight / bias gradient
};

__device__ __forceinline__ float dw_load(const DwArgs& a, long off) {
  return a.x_is_f16 ? (float)((const f16*)a.x)[off] : ((const float*)a.x)[off];
}

// LDS image of the input tile with a halo of HALO tokens: [(DT + 2 HALO)^2][DC (+pad per row of tokens)]
template <int HALO>
__device__ __forceinline__ void dw_stage_x(const DwArgs& a, float* X, int b, int h0, int w0, int d0) {
  constexpr int E = DT + 2 * HALO;
  constexpr int PITCH = E * DC + DPAD;
  // whole 16-channel groups of 16-byte aligned rows: one 16-byte load per 8 (fp16) / 4 (fp32) channels of a token instead of one
  // load per element (round 5: the element-wise form issued 25 load + index sequences per thread in front of the halo-2 tile)
  const int cpp = a.x_is_f16 ? 8 : 4;                     // channels per 16-byte piece
  const bool vec = d0 + DC <= a.D && (a.x_stride % cpp) == 0 && (((size_t)a.x) & 15) == 0 && (d0 % cpp) == 0;
  if (vec) {
    const int ppt = DC / cpp;                               // pieces per token
    for (int i = threadIdx.x; i < E * E * ppt; i += 256) {
      const int pc = i % ppt, tx = (i / ppt) % E, ty = i / (ppt * E);
      const int h = h0 - HALO + ty, w = w0 - HALO + tx;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) {
        const long off = ((long)(b * a.H + h) * a.W + w) * a.x_stride + d0 + pc * cpp;
        if (a.x_is_f16) {
          const f16x8 t = *reinterpret_cast<const f16x8*>((const f16*)a.x + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
        } else {
          const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)a.x + off);
          v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        }
      }
      float* dst = X + ty * PITCH + tx * DC + pc * cpp;
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      if (a.x_is_f16) *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    return;
  }
  for (int i = threadIdx.x; i < E * E * DC; i += 256) {
    const int d = i % DC, tx = (i / DC) % E, ty = i / (DC * E);
    const int h = h0 - HALO + ty, w = w0 - HALO + tx;
    float v = 0.f;
    if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && d0 + d < a.D)
      v = dw_load(a, ((long)(b * a.H + h) * a.W + w) * a.x_stride + d0 + d);
    X[ty * PITCH + tx * DC + d] = v;
  }
}

__global__ __launch_bounds__(256) void dwconv_silu_fwd_kernel(DwArgs a) {
  constexpr int E = DT + 2, PITCH = E * DC + DPAD;
  __shared__ __attribute__((aligned(16))) float X[E * PITCH];
  __shared__ float T[DC * (DT * (DT + 1) + 1)];
  constexpr int TP = DT * (DT + 1) + 1;
  const int tiles_w = (a.W + DT - 1) / DT;
  const int h0 = (blockIdx.x / tiles_w) * DT, w0 = (blockIdx.x % tiles_w) * DT;
  const int d0 = blockIdx.y * DC, b = blockIdx.z;
  dw_stage_x<1>(a, X, b, h0, w0, d0);
  const int d = threadIdx.x % DC, th = threadIdx.x / DC;
  float wv[9], bv = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = d0 + d < a.D ? a.w[(long)(d0 + d) * 9 + t] : 0.f;
  if (a.bias && d0 + d < a.D) bv = a.bias[d0 + d];
  __syncthreads();
  for (int tw = 0; tw < DT; ++tw) {
    float acc = bv;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc += wv[i * 3 + j] * X[(th + i) * PITCH + (tw + j) * DC + d];
    T[d * TP + th * (DT + 1) + tw] = acc / (1.f + __expf(-acc));
  }
  __syncthreads();
  const long L = (long)a.H * a.W;
  const int lo = threadIdx.x % DT, hi = threadIdx.x / DT;
  for (int dd = 0; dd < DC; ++dd) {
    if (d0 + dd >= a.D) break;
    float* rm = a.x2 + ((long)b * a.D + d0 + dd) * L;
    float* cm = rm + (long)a.B * a.D * L;
    {
      const int h = h0 + hi, w = w0 + lo;
      if (h < a.H && w < a.W) rm[(long)h * a.W + w] = T[dd * TP + hi * (DT + 1) + lo];
    }
    {
      const int w = w0 + hi, h = h0 + lo;
      if (h < a.H && w < a.W) cm[(long)w * a.H + h] = T[dd * TP + lo * (DT + 1) + hi];
    }
  }
}

__global__ __launch_bounds__(256) void dwconv_silu_bwd_kernel(DwArgs a) {
  constexpr int E2 = DT + 4, P2 = E2 * DC + DPAD;  // input with halo 2
  constexpr int E1 = DT + 2, P1 = E1 * DC + DPAD;  // pre-activation gradient with halo 1
  __shared__ __attribute__((aligned(16))) float X[E2 * P2];
  __shared__ float G[E1 * P1];
  __shared__ float red[16][DC][10];
  const int tiles_w = (a.W + DT - 1) / DT;
  const int h0 = (blockIdx.x / tiles_w) * DT, w0 = (blockIdx.x % tiles_w) * DT;
  const int d0 = blockIdx.y * DC, b = blockIdx.z;
  const long L = (long)a.H * a.W;
  dw_stage_x<2>(a, X, b, h0, w0, d0);
  // gradient of the activation on the tile + halo 1: row-major source with lanes along w, column-major with lanes along h
  for (int i = threadIdx.x; i < DC * E1 * E1; i += 256) {
    const int tx = i % E1, ty = (i / E1) % E1, d = i / (E1 * E1);
    const int h = h0 - 1 + ty, w = w0 - 1 + tx;
    float v = 0.f;
    if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && d0 + d < a.D)
      v = a.dx2[((long)b * a.D + d0 + d) * L + (long)h * a.W + w];
    G[ty * P1 + tx * DC + d] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < DC * E1 * E1; i += 256) {
    const int ty = i % E1, tx = (i / E1) % E1, d = i / (E1 * E1);
    const int h = h0 - 1 + ty, w = w0 - 1 + tx;
    if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && d0 + d < a.D)
      G[ty * P1 + tx * DC + d] += a.dx2[((long)(a.B + b) * a.D + d0 + d) * L + (long)w * a.H + h];
  }
  const int d = threadIdx.x % DC, th = threadIdx.x / DC;
  float wv[9], bv = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = d0 + d < a.D ? a.w[(long)(d0 + d) * 9 + t] : 0.f;
  if (a.bias && d0 + d < a.D) bv = a.bias[d0 + d];
  __syncthreads();
  // G <- d(pre-activation) = d(act) * silu'(pre), pre recomputed from the input image; thread (d, row) walks its row(s) with
  // a 3 x 3 window of the image in registers: three LDS reads per position instead of nine (round 5)
  for (int ty = th; ty < E1; ty += 16) {
    float c0[3], c1[3], c2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      c0[i] = X[(ty + i) * P2 + 0 * DC + d];
      c1[i] = X[(ty + i) * P2 + 1 * DC + d];
    }
#pragma unroll
    for (int tx = 0; tx < E1; ++tx) {
      float pre = bv;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c2[i] = X[(ty + i) * P2 + (tx + 2) * DC + d];
        pre += wv[i * 3 + 0] * c0[i] + wv[i * 3 + 1] * c1[i] + wv[i * 3 + 2] * c2[i];
      }
      const float sg = 1.f / (1.f + __expf(-pre));
      G[ty * P1 + tx * DC + d] *= sg * (1.f + pre * (1.f - sg));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c0[i] = c1[i];
        c1[i] = c2[i];
      }
    }
  }
  __syncthreads();
  // input gradient on the tile and the weight / bias partials of the tile's own outputs
  float pw[9], pb = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) pw[t] = 0.f;
  const int h = h0 + th;
  {
    // windows in registers, sliding along the row: G rows th .. th + 2, columns tw .. tw + 2 (the data gradient reads them
    // mirrored); X rows th + 1 .. th + 3, columns tw + 1 .. tw + 3 (the taps of the tile's own output at (th, tw))
    float g0[3], g1[3], g2[3], x0[3], x1[3], x2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g0[i] = G[(th + i) * P1 + 0 * DC + d];
      g1[i] = G[(th + i) * P1 + 1 * DC + d];
      x0[i] = X[(th + 1 + i) * P2 + 1 * DC + d];
      x1[i] = X[(th + 1 + i) * P2 + 2 * DC + d];
    }
#pragma unroll
    for (int tw = 0; tw < DT; ++tw) {
      const int w = w0 + tw;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        g2[i] = G[(th + i) * P1 + (tw + 2) * DC + d];
        x2[i] = X[(th + 1 + i) * P2 + (tw + 3) * DC + d];
      }
      // dx[h][w] = sum_{i, j} w[i][j] dpre[h - i + 1][w - j + 1]: G row th + 2 - i, column tw + 2 - j
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        acc += wv[i * 3 + 0] * g2[2 - i] + wv[i * 3 + 1] * g1[2 - i] + wv[i * 3 + 2] * g0[2 - i];
      if (h < a.H && w < a.W && d0 + d < a.D) {
        const long o = ((long)(b * a.H + h) * a.W + w) * a.D + d0 + d;
        if (a.x_is_f16) ((f16*)a.dx)[o] = (f16)acc;
        else ((float*)a.dx)[o] = acc;
      }
      const float g = g1[1];  // dpre of this thread's own output (0 outside the image)
      pb += g;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        pw[i * 3 + 0] += g * x0[i];
        pw[i * 3 + 1] += g * x1[i];
        pw[i * 3 + 2] += g * x2[i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        g0[i] = g1[i]; g1[i] = g2[i];
        x0[i] = x1[i]; x1[i] = x2[i];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[th][d][t] = pw[t];
  red[th][d][9] = pb;
  __syncthreads();
  if (threadIdx.x < DC * 10) {
    const int dd = threadIdx.x / 10, t = threadIdx.x % 10;
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += red[r][dd][t];
    if (d0 + dd < a.D) {
      if (a.part) {   // two-stage mode: row (sample, tile) of the workspace, layout [D * 9 weights | D biases]
        float* prow = a.part + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * ((size_t)a.D * 10);
        prow[t < 9 ? (long)(d0 + dd) * 9 + t : (long)a.D * 9 + d0 + dd] = s;
      } else if (t < 9) {
        atomicAdd(a.dw + (long)(d0 + dd) * 9 + t, s);
      } else if (a.dbias) {
        atomicAdd(a.dbias + d0 + dd, s);
      }
    }
  }
}

}  // namespace nnz

extern "C" int nnz_ss2d_dwconv_silu_forward(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight,
                                            const float* bias, float* x2, int Bt, int D, int H, int W, void* stream) {
  using namespace nnz;
  if (!x_tokens || !weight || !x2 || Bt < 1 || D < 1 || H < 1 || W < 1 || Bt > 65535 || (D + DC - 1) / DC > 65535)
    return NNZ_EINVAL;
  DwArgs a = {};
  a.x = x_tokens; a.x_is_f16 = x_is_f16; a.x_stride = x_row_stride; a.w = weight; a.bias = bias; a.x2 = x2;
  a.B = Bt; a.D = D; a.H = H; a.W = W;
  dim3 grid(((H + DT - 1) / DT) * ((W + DT - 1) / DT), (D + DC - 1) / DC, Bt);
  NNZ_LAUNCH(dwconv_silu_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

static int dw_bwd_impl(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight, const float* bias,
                       const float* dx2, void* dx_tokens, float* dweight, float* dbias, float* workspace, long ws_floats,
                       int Bt, int D, int H, int W, void* stream) {
  using namespace nnz;
  if (!x_tokens || !weight || !dx2 || !dx_tokens || !dweight || Bt < 1 || D < 1 || H < 1 || W < 1 || Bt > 65535 ||
      (D + DC - 1) / DC > 65535)
    return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  dim3 grid(((H + DT - 1) / DT) * ((W + DT - 1) / DT), (D + DC - 1) / DC, Bt);
  const long nparts = (long)grid.x * Bt;
  if (workspace) {
    if (ws_floats < nparts * (long)D * 10 + fold_partials_scratch_floats((int)nparts, (long)D * 10)) return NNZ_EINVAL;
  } else if (dbias == dweight + (long)D * 9) {  // one buffer [D*9 + D]: one launch
    if ((e = zero_async(dweight, sizeof(float) * D * 10, s)) != hipSuccess) return (int)e;
  } else {
    if ((e = zero_async(dweight, sizeof(float) * D * 9, s)) != hipSuccess) return (int)e;
    if (dbias && (e = zero_async(dbias, sizeof(float) * D, s)) != hipSuccess) return (int)e;
  }
  DwArgs a = {};
  a.x = x_tokens; a.x_is_f16 = x_is_f16; a.x_stride = x_row_stride; a.w = weight; a.bias = bias; a.dx2 = dx2;
  a.dx = dx_tokens; a.dw = dweight; a.dbias = dbias; a.part = workspace;
  a.B = Bt; a.D = D; a.H = H; a.W = W;
  NNZ_LAUNCH(dwconv_silu_bwd_kernel, grid, dim3(256), 0, s, a);
  if (workspace) {
    const long row = (long)D * 10;
    float* scratch = fold_partials_scratch_floats((int)nparts, row) ? workspace + nparts * row : nullptr;
    if ((e = fold_partials(workspace, (int)nparts, row, (long)D * 9, dweight, s, scratch)) != hipSuccess) return (int)e;
    if (dbias && (e = fold_partials(workspace + (long)D * 9, (int)nparts, row, D, dbias, s, scratch)) != hipSuccess)
      return (int)e;
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// weight / bias gradient: nnz_ss2d_dwconv_silu_backward zero-fills dweight / dbias and the tiles add with fp32 atomics;
// nnz_ss2d_dwconv_silu_backward_ws WRITES them from per-tile partial rows in `workspace`
// (nnz_ss2d_dwconv_silu_backward_workspace_floats) folded in a fixed order: bit-reproducible.
extern "C" long nnz_ss2d_dwconv_silu_backward_workspace_floats(int Bt, int D, int H, int W) {
  using namespace nnz;
  if (Bt < 1 || D < 1 || H < 1 || W < 1) return 0;
  const long nparts = (long)((H + DT - 1) / DT) * ((W + DT - 1) / DT) * Bt;
  return nparts * (long)D * 10 + fold_partials_scratch_floats((int)nparts, (long)D * 10);
}
extern "C" int nnz_ss2d_dwconv_silu_backward(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight,
                                             const float* bias, const float* dx2, void* dx_tokens, float* dweight,
                                             float* dbias, int Bt, int D, int H, int W, void* stream) {
  return dw_bwd_impl(x_tokens, x_is_f16, x_row_stride, weight, bias, dx2, dx_tokens, dweight, dbias, nullptr, 0, Bt, D, H, W,
                     stream);
}
extern "C" int nnz_ss2d_dwconv_silu_backward_ws(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight,
                                                const float* bias, const float* dx2, void* dx_tokens, float* dweight,
                                                float* dbias, float* workspace, long ws_floats, int Bt, int D, int H, int W,
                                                void* stream) {
  if (!workspace) return NNZ_EINVAL;
  return dw_bwd_impl(x_tokens, x_is_f16, x_row_stride, weight, bias, dx2, dx_tokens, dweight, dbias, workspace, ws_floats, Bt,
                     D, H, W, stream);
}
