// Residual add with stochastic depth for the VSS / SSND blocks: out = input + x * mask[b] * scale
// (reference: `input + self.drop_path(self.self_attention(self.ln_1(input)))`, m2net.py:530 with timm's DropPath:
// mask = x.new_empty((B, 1, ..)).bernoulli_(keep); mask.div_(keep); x * mask).  The mask is drawn by the caller with
// the same torch call as the reference (same RNG stream); scaling, multiply and add are one pass here instead of three
// (and one instead of two in the backward: dx = dout * mask[b] * scale, d(input) = dout).
#include "common.hpp"

namespace nnz {

struct ResArgs {
  const void* input;  // [B][P] f16 or f32
  const void* x;      // [B][P] f16 or f32
  const void* mask;   // [B] f16 or f32 (0 / 1), may be null (no drop: mask = 1)
  void* out;          // [B][P] f32 unless both inputs are f16
  long P;             // elements per sample, multiple of 4
  int B;
  int in_f16, x_f16, mask_f16, out_f16;
  float scale;
  float mask_add;     // != 0: mask holds uniform draws r[b] and the 0 / 1 mask is floor(r[b] + mask_add) (timm-style DropPath)
};

__device__ __forceinline__ f32x4 ld4any(const void* p, long i, int is_f16) {
  if (is_f16) {
    const f16x4 h = *reinterpret_cast<const f16x4*>((const f16*)p + i);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  return *reinterpret_cast<const f32x4*>((const float*)p + i);
}
__device__ __forceinline__ void st4any(void* p, long i, int is_f16, f32x4 v) {
  if (is_f16) *reinterpret_cast<f16x4*>((f16*)p + i) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  else *reinterpret_cast<f32x4*>((float*)p + i) = v;
}

// BWD: out = dx = dout(input slot) * m   (x unused)
template <bool BWD>
__global__ __launch_bounds__(256) void residual_droppath_kernel(ResArgs a) {
  const int b = blockIdx.y;
  float m = a.scale;
  if (a.mask) {
    float mv = a.mask_f16 ? (float)((const f16*)a.mask)[b] : ((const float*)a.mask)[b];
    if (a.mask_add != 0.f) mv = floorf(mv + a.mask_add);
    m *= mv;
  }
  const long base = (long)b * a.P;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < a.P; i += (long)gridDim.x * 1024) {
    const f32x4 r = ld4any(a.input, base + i, a.in_f16);
    f32x4 o;
    if (BWD) {
      o = r * m;
    } else {
      const f32x4 xv = ld4any(a.x, base + i, a.x_f16);
      o = r + xv * m;
    }
    st4any(a.out, base + i, a.out_f16, o);
  }
}

static int res_launch(const ResArgs& a, bool bwd, hipStream_t s) {
  if (!a.input || !a.out || a.B < 1 || a.B > 65535 || a.P < 4 || (a.P & 3)) return NNZ_EINVAL;
  long wg = (a.P / 4 + 255) / 256;
  if (wg > 2048) wg = 2048;
  dim3 grid((unsigned)wg, a.B);
  if (bwd) NNZ_LAUNCH(residual_droppath_kernel<true>, grid, dim3(256), 0, s, a);
  else NNZ_LAUNCH(residual_droppath_kernel<false>, grid, dim3(256), 0, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_residual_droppath_forward(const void* input, int input_is_f16, const void* x, int x_is_f16,
                                             const void* mask, int mask_is_f16, float scale, void* out, int out_is_f16,
                                             int B, long per_sample, void* stream) {
  using namespace nnz;
  if (!x) return NNZ_EINVAL;
  ResArgs a = {};
  a.input = input; a.x = x; a.mask = mask; a.out = out; a.P = per_sample; a.B = B;
  a.in_f16 = input_is_f16; a.x_f16 = x_is_f16; a.mask_f16 = mask_is_f16; a.out_f16 = out_is_f16; a.scale = scale;
  return res_launch(a, false, (hipStream_t)stream);
}

extern "C" int nnz_residual_droppath_backward(const void* dout, int dout_is_f16, const void* mask, int mask_is_f16,
                                              float scale, void* dx, int dx_is_f16, int B, long per_sample,
                                              void* stream) {
  using namespace nnz;
  ResArgs a = {};
  a.input = dout; a.mask = mask; a.out = dx; a.P = per_sample; a.B = B;
  a.in_f16 = dout_is_f16; a.mask_f16 = mask_is_f16; a.out_f16 = dx_is_f16; a.scale = scale;
  return res_launch(a, true, (hipStream_t)stream);
}

// the same pair with the mask still to be made: `rand` = B fp32 uniform draws, mask[b] = floor(rand[b] + keep) - what the
// reference's DropPath computes with an add and a floor_ launch of its own (swt2net.py:379-388)
extern "C" int nnz_residual_droppath_rand_forward(const void* input, int input_is_f16, const void* x, int x_is_f16,
                                                  const float* rand, float keep, float scale, void* out,
                                                  int out_is_f16, int B, long per_sample, void* stream) {
  using namespace nnz;
  if (!x || !rand || !(keep > 0.f)) return NNZ_EINVAL;
  ResArgs a = {};
  a.input = input; a.x = x; a.mask = rand; a.out = out; a.P = per_sample; a.B = B; a.mask_add = keep;
  a.in_f16 = input_is_f16; a.x_f16 = x_is_f16; a.out_f16 = out_is_f16; a.scale = scale;
  return res_launch(a, false, (hipStream_t)stream);
}

extern "C" int nnz_residual_droppath_rand_backward(const void* dout, int dout_is_f16, const float* rand, float keep,
                                                   float scale, void* dx, int dx_is_f16, int B, long per_sample,
                                                   void* stream) {
  using namespace nnz;
  if (!rand || !(keep > 0.f)) return NNZ_EINVAL;
  ResArgs a = {};
  a.input = dout; a.mask = rand; a.out = dx; a.P = per_sample; a.B = B; a.mask_add = keep;
  a.in_f16 = dout_is_f16; a.out_f16 = dx_is_f16; a.scale = scale;
  return res_launch(a, true, (hipStream_t)stream);
}

// ---- top / left zero padding of a channels-last map and its inverse crop (SwinTransformerBlock.forward pads every block's
// input to a multiple of the 7-token window and crops the result: F.pad is a fill + a copy, the crop's backward another
// fill + copy - here each is one launch).  small: [B][H][W][C], big: [B][H + py][W + px][C], fp32, C % 4 == 0.
namespace nnz {
struct PadArgs {
  const float* src;
  float* dst;
  long rowC;      // W * C of the SMALL map
  long big_rowC;  // (W + px) * C
  long pxC;       // px * C
  int H, py, B;
};

// PAD: dst = big, every element written (zeros in the first py rows / first px columns); else dst = small = src[.., py:, px:, :]
template <bool PAD>
__global__ __launch_bounds__(256) void pad_crop_kernel(PadArgs a) {
  const int rows = PAD ? a.H + a.py : a.H;                  // rows of dst
  const long dst_rowC = PAD ? a.big_rowC : a.rowC;
  const long row = blockIdx.y;                              // b * rows + y
  const int b = (int)(row / rows), y = (int)(row % rows);
  float* d = a.dst + row * dst_rowC;
  if (PAD) {
    const bool live = y >= a.py;
    const float* sp = a.src + ((long)b * a.H + (y - a.py)) * a.rowC - a.pxC;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < dst_rowC; i += (long)gridDim.x * 1024) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live && i >= a.pxC) v = *reinterpret_cast<const f32x4*>(sp + i);
      *reinterpret_cast<f32x4*>(d + i) = v;
    }
  } else {
    const float* sp = a.src + ((long)b * (a.H + a.py) + (y + a.py)) * a.big_rowC + a.pxC;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < dst_rowC; i += (long)gridDim.x * 1024)
      *reinterpret_cast<f32x4*>(d + i) = *reinterpret_cast<const f32x4*>(sp + i);
  }
}
}  // namespace nnz

static int pad_crop_launch(const float* src, float* dst, int B, int H, int W, int C, int py, int px, bool pad, void* stream) {
  using namespace nnz;
  if (!src || !dst || B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || py < 0 || px < 0) return NNZ_EINVAL;
  PadArgs a = {};
  a.src = src; a.dst = dst; a.rowC = (long)W * C; a.big_rowC = (long)(W + px) * C; a.pxC = (long)px * C;
  a.H = H; a.py = py; a.B = B;
  const long rows = (long)B * (pad ? H + py : H);
  if (rows > 65535) return NNZ_EINVAL;
  const long rc = pad ? a.big_rowC : a.rowC;
  long gx = (rc / 4 + 255) / 256;
  if (gx > 64) gx = 64;
  if (pad) NNZ_LAUNCH(pad_crop_kernel<true>, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
  else NNZ_LAUNCH(pad_crop_kernel<false>, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_pad_top_left(const float* small, float* big, int B, int H, int W, int C, int py, int px, void* stream) {
  return pad_crop_launch(small, big, B, H, W, C, py, px, true, stream);
}
extern "C" int nnz_crop_top_left(const float* big, float* small, int B, int H, int W, int C, int py, int px, void* stream) {
  return pad_crop_launch(big, small, B, H, W, C, py, px, false, stream);
}
