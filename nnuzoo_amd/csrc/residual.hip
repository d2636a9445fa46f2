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
