// Sliding-window inference accumulation (gfx950): the arithmetic of
//   nnUNetPredictor._internal_maybe_mirror_and_predict   (/root/reference/nnunetv2/inference/predict_from_raw_data.py:549-564)
//   nnUNetPredictor._internal_predict_sliding_window_return_logits  (same file :566-643)
// after the network forward, as two HBM-bound kernels:
//   accumulate: p = out_0; for m >= 1: p = f16(p + flip_m(out_m)); p = f16(p / M)        (mirror TTA merge)
//               p = f16(p * gaussian); logits[sl] = f16(logits[sl] + p); n_pred[sl] = f16(n_pred[sl] + gaussian)
//   finalize  : logits = f16(logits / n_pred), inf flag
// Every intermediate is rounded to fp16 exactly where the reference's half tensors round (torch computes half ops in
// fp32 and rounds once), so the result is bit-identical to the reference loop given the same network outputs.  Tiles
// overlap, so tiles are accumulated one launch after another in slicer order (stream order = the reference's order).
#include "common.hpp"

namespace nnz {

struct SwArgs {
  const f16* preds;  // [M][K][td][th][tw]
  const f16* gauss;  // [td][th][tw] or null (weight 1)
  f16* logits;       // [K][D][H][W]
  f16* npred;        // [D][H][W]
  int M, K;
  int t[3], dims[3], off[3];
  int flip[8];       // bit a set: prediction m was computed on the input flipped along spatial axis a
};

__device__ __forceinline__ f16 hadd(f16 a, f16 b) { return (f16)((float)a + (float)b); }

__global__ __launch_bounds__(256) void sw_accumulate_kernel(SwArgs a) {
  const long tv = (long)a.t[0] * a.t[1] * a.t[2];
  const long iv = (long)a.dims[0] * a.dims[1] * a.dims[2];
  for (long v = blockIdx.x * 256L + threadIdx.x; v < tv; v += (long)gridDim.x * 256) {
    const int x = v % a.t[2];
    const int y = (v / a.t[2]) % a.t[1];
    const int z = v / ((long)a.t[2] * a.t[1]);
    const f16 g = a.gauss ? a.gauss[v] : (f16)1.0f;
    const long o = ((long)(a.off[0] + z) * a.dims[1] + (a.off[1] + y)) * a.dims[2] + (a.off[2] + x);
    for (int k = 0; k < a.K; ++k) {
      f16 p = a.preds[(long)k * tv + v];
      for (int m = 1; m < a.M; ++m) {
        const int fl = a.flip[m];
        const int zz = (fl & 1) ? a.t[0] - 1 - z : z;
        const int yy = (fl & 2) ? a.t[1] - 1 - y : y;
        const int xx = (fl & 4) ? a.t[2] - 1 - x : x;
        p = hadd(p, a.preds[((long)m * a.K + k) * tv + ((long)zz * a.t[1] + yy) * a.t[2] + xx]);
      }
      if (a.M > 1) p = (f16)((float)p / (float)a.M);
      if (a.gauss) p = (f16)((float)p * (float)g);
      f16* dst = a.logits + (long)k * iv + o;
      *dst = hadd(*dst, p);
    }
    a.npred[o] = hadd(a.npred[o], g);
  }
}

__global__ __launch_bounds__(256) void sw_finalize_kernel(f16* logits, const f16* npred, int K, long V, int* inf_flag) {
  const long total = (long)K * V;
  bool bad = false;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const f16 r = (f16)((float)logits[i] / (float)npred[i % V]);
    logits[i] = r;
    bad |= __builtin_isinf((float)r);
  }
  if (bad) atomicOr(inf_flag, 1);
}

}  // namespace nnz

extern "C" int nnz_sliding_window_accumulate(const void* preds_f16, int M, const int* flip_bits, const void* gaussian_f16,
                                             void* logits_f16, void* npred_f16, int K, const int* tile_dims,
                                             const int* image_dims, const int* offset, void* stream) {
  using namespace nnz;
  if (!preds_f16 || !logits_f16 || !npred_f16 || !tile_dims || !image_dims || !offset || M < 1 || M > 8 || K < 1 ||
      (M > 1 && !flip_bits))
    return NNZ_EINVAL;
  SwArgs a = {};
  a.preds = (const f16*)preds_f16;
  a.gauss = (const f16*)gaussian_f16;
  a.logits = (f16*)logits_f16;
  a.npred = (f16*)npred_f16;
  a.M = M;
  a.K = K;
  for (int i = 0; i < 3; ++i) {
    a.t[i] = tile_dims[i];
    a.dims[i] = image_dims[i];
    a.off[i] = offset[i];
    if (a.t[i] < 1 || a.off[i] < 0 || a.off[i] + a.t[i] > a.dims[i]) return NNZ_EINVAL;
  }
  for (int m = 0; m < M; ++m) a.flip[m] = flip_bits ? flip_bits[m] : 0;
  if (a.flip[0] != 0) return NNZ_EINVAL;  // prediction 0 is the un-mirrored one
  const long tv = (long)a.t[0] * a.t[1] * a.t[2];
  long blocks = (tv + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(sw_accumulate_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_sliding_window_finalize(void* logits_f16, const void* npred_f16, int K, long V, int* inf_flag,
                                           void* stream) {
  using namespace nnz;
  if (!logits_f16 || !npred_f16 || !inf_flag || K < 1 || V < 1) return NNZ_EINVAL;
  long blocks = ((long)K * V + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(sw_finalize_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (f16*)logits_f16,
                     (const f16*)npred_f16, K, V, inf_flag);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
