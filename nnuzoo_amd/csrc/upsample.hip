// Bilinear up-sampling of the deep-supervision side outputs to the full map and its adjoint
// (reference: `_upsample_like` = F.interpolate(src, size=tar.shape[2:], mode='bilinear'), /root/reference/nnunetv2/nets/m2net.py:33-36 (F.upsample),
// used at :948-950 for d2 .. d6 and inside every RSU block; align_corners=False: source index max(scale * (o + 0.5) - 0.5, 0) with
// scale = n_in / n_out, neighbours clamped to the last sample - ATen's area_pixel_compute_source_index).
// Forward: one thread per output element, 4 reads.  Backward: the ADJOINT as a gather - every input pixel sums the output pixels that
// read it, in a fixed order (ATen scatters with float atomics: 438 us per call on the 32 x side outputs and order-dependent sums; rounds
// 2-5 used two library GEMMs with the interpolation matrices).  A workgroup owns one input row y of one (sample, channel) plane: column
// sums over the output rows that touch y (coalesced along X) through LDS, then the same along X.  HBM-bound, ~2 reads of dOut.
#include "common.hpp"

namespace nnz {

struct UpArgs {
  const void* src;   // forward: in [B][h][w];   backward: dOut [B][H][W]
  void* dst;         // forward: out [B][H][W];  backward: dIn [B][h][w]
  int B, h, w, H, W;
  float sy, sx;      // h / H, w / W
};

__device__ __forceinline__ void up_source(float scale, int o, int n_in, int& i0, int& i1, float& l1) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_up_fwd_kernel(UpArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)a.H * a.W;
  if (i >= per * a.B) return;
  const int b = (int)(i / per);
  const int r = (int)(i - (long)b * per);
  const int Y = r / a.W, X = r - Y * a.W;
  int y0, y1, x0, x1;
  float ly, lx;
  up_source(a.sy, Y, a.h, y0, y1, ly);
  up_source(a.sx, X, a.w, x0, x1, lx);
  const T* p = (const T*)a.src + (long)b * a.h * a.w;
  const float v00 = (float)p[y0 * a.w + x0], v01 = (float)p[y0 * a.w + x1];
  const float v10 = (float)p[y1 * a.w + x0], v11 = (float)p[y1 * a.w + x1];
  const float o = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  ((T*)a.dst)[i] = (T)o;
}

// first / last output index whose two sources can include input index i (conservative by one on each side; the exact test is the weight)
__device__ __forceinline__ void up_range(float scale, int i, int n_out, int& lo, int& hi) {
  lo = (int)floorf(((float)i - 0.5f) / scale - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) / scale - 0.5f) + 1;
  if (lo < 0) lo = 0;
  if (hi > n_out - 1) hi = n_out - 1;
}
__device__ __forceinline__ float up_weight(float scale, int o, int n_in, int i) {
  int i0, i1;
  float l1;
  up_source(scale, o, n_in, i0, i1, l1);
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_up_bwd_kernel(UpArgs a) {
  extern __shared__ float col[];      // [W]
  const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const T* g = (const T*)a.src + (long)b * a.H * a.W;
  int lo, hi;
  up_range(a.sy, y, a.H, lo, hi);
  for (int X = tid; X < a.W; X += 256) {
    float acc = 0.f;
    for (int Y = lo; Y <= hi; ++Y) {
      const float wgt = up_weight(a.sy, Y, a.h, y);
      if (wgt != 0.f) acc += wgt * (float)g[(long)Y * a.W + X];
    }
    col[X] = acc;
  }
  __syncthreads();
  T* out = (T*)a.dst + ((long)b * a.h + y) * a.w;
  for (int x = tid; x < a.w; x += 256) {
    int xl, xh;
    up_range(a.sx, x, a.W, xl, xh);
    float acc = 0.f;
    for (int X = xl; X <= xh; ++X) {
      const float wgt = up_weight(a.sx, X, a.w, x);
      if (wgt != 0.f) acc += wgt * col[X];
    }
    out[x] = (T)acc;
  }
}

}  // namespace nnz

static int up_ok(const void* a, const void* b, int B, int h, int w, int H, int W) {
  return a && b && B >= 1 && B <= 65535 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && h <= 65535 && W <= 16384 &&
         (long)B * H * W < (1L << 40);
}
// src [B][h][w] -> dst [B][H][W], B = samples x channels; fp32 or IEEE half (is_f16), fp32 arithmetic
extern "C" int nnz_bilinear_up_forward(const void* src, void* dst, int is_f16, int B, int h, int w, int H, int W, void* stream) {
  using namespace nnz;
  if (!up_ok(src, dst, B, h, w, H, W)) return NNZ_EINVAL;
  UpArgs a = {src, dst, B, h, w, H, W, (float)h / (float)H, (float)w / (float)W};
  const long blocks = ((long)B * H * W + 255) / 256;
  if (blocks > 0x7fffffffL) return NNZ_EINVAL;
  if (is_f16) NNZ_LAUNCH(bilinear_up_fwd_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else NNZ_LAUNCH(bilinear_up_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
// dout [B][H][W] -> din [B][h][w] (written): the adjoint of the forward, sums in a fixed order
extern "C" int nnz_bilinear_up_backward(const void* dout, void* din, int is_f16, int B, int h, int w, int H, int W, void* stream) {
  using namespace nnz;
  if (!up_ok(dout, din, B, h, w, H, W)) return NNZ_EINVAL;
  UpArgs a = {dout, din, B, h, w, H, W, (float)h / (float)H, (float)w / (float)W};
  const int lds = W * (int)sizeof(float);
  if (is_f16) NNZ_LAUNCH(bilinear_up_bwd_kernel<_Float16>, dim3((unsigned)h, (unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
  else NNZ_LAUNCH(bilinear_up_bwd_kernel<float>, dim3((unsigned)h, (unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
