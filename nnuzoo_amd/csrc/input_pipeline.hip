// GPU-side input pipeline of the training step: the batch is cut out of cases that are RESIDENT in HBM
// (288 GB holds whole preprocessed datasets), so the per-step host work is choosing B bounding boxes - no worker processes,
// no host -> device copy of patches.
//   reference: nnUNetDataLoader.generate_train_batch, /root/reference/nnunetv2/training/dataloading/data_loader.py:180-259
//     data_all[j] = crop_and_pad_nd(data, bbox, 0);  seg_all[j] = crop_and_pad_nd(seg, bbox, -1)      (:207-218)
//   then, of the reference's transform chain, the two that only move voxels: MirrorTransform (flip of the patch along the
//   drawn axes, nnUNetTrainer.py:917-920) and DownsampleSegForDSTransform (one nearest-exact resampled target per
//   deep-supervision scale, :971).
// Kernels are pure gathers: HBM-bound, coalesced along the fastest axis, bit-exact integer / copy work.
#include "common.hpp"

namespace nnz {

constexpr int IP_MAXB = 16;   // samples per launch (their descriptors travel as kernel arguments: no table upload)

struct CropArgs {
  const void* src[IP_MAXB];   // case volume [C][sd][sh][sw]
  int shape[IP_MAXB][3];      // sd, sh, sw of the case
  int lb[IP_MAXB][3];         // lower corner of the bounding box in case coordinates (may be negative)
  int flip[IP_MAXB];          // bit a set: the patch is mirrored along axis a (0 = d, 1 = h, 2 = w)
  void* dst;                  // [nb][C][pd][ph][pw]
  int nb, C, pd, ph, pw;
  float pad_f;
  int pad_i;
};

template <typename T>
__global__ __launch_bounds__(256) void crop_pad_kernel(CropArgs a) {
  const int b = blockIdx.z, c = blockIdx.y;
  const long vol = (long)a.pd * a.ph * a.pw;
  const T* src = static_cast<const T*>(a.src[b]) + (long)c * a.shape[b][0] * a.shape[b][1] * a.shape[b][2];
  T* dst = static_cast<T*>(a.dst) + ((long)b * a.C + c) * vol;
  const T pad = sizeof(T) == 4 ? (T)a.pad_f : (T)a.pad_i;
  const int sd = a.shape[b][0], sh = a.shape[b][1], sw = a.shape[b][2];
  const int fl = a.flip[b];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long)gridDim.x * 256) {
    const int x = (int)(i % a.pw);
    const long r = i / a.pw;
    const int y = (int)(r % a.ph), z = (int)(r / a.ph);
    const int zz = a.lb[b][0] + ((fl & 1) ? a.pd - 1 - z : z);
    const int yy = a.lb[b][1] + ((fl & 2) ? a.ph - 1 - y : y);
    const int xx = a.lb[b][2] + ((fl & 4) ? a.pw - 1 - x : x);
    const bool in = zz >= 0 && zz < sd && yy >= 0 && yy < sh && xx >= 0 && xx < sw;
    dst[i] = in ? src[((long)zz * sh + yy) * sw + xx] : pad;
  }
}

// torch's 'nearest-exact' source index: min(floor((o + 0.5) * scale), in - 1), scale = in / out in float32.  This file is
// built WITHOUT fast-math (build.py STRICT_FP): an approximate 24 / 3 = 7.9999995 turns index 4 into 3.
__device__ __forceinline__ int nearest_exact(int o, float scale, int in) {
  const int s = (int)floorf(((float)o + 0.5f) * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ __launch_bounds__(256) void downsample_nearest_i16_kernel(const short* __restrict__ src, short* __restrict__ dst,
                                                                     long nc, int id, int ih, int iw, int od, int oh, int ow) {
  const float sd = (float)id / (float)od, sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;
  const long ovol = (long)od * oh * ow, total = nc * ovol;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / ovol, r = i % ovol;
    const int x = (int)(r % ow), y = (int)((r / ow) % oh), z = (int)(r / ((long)ow * oh));
    dst[i] = src[((n * id + nearest_exact(z, sd, id)) * ih + nearest_exact(y, sh, ih)) * iw + nearest_exact(x, sw, iw)];
  }
}

template <typename T>
static int crop_pad_launch(const void* const* src, const int* shapes, const int* lbs, const int* flips, void* dst, int B,
                           int C, int pd, int ph, int pw, float pad_f, int pad_i, hipStream_t s) {
  if (!src || !shapes || !lbs || !dst || B < 1 || C < 1 || pd < 1 || ph < 1 || pw < 1) return NNZ_EINVAL;
  const long vol = (long)pd * ph * pw;
  for (int b0 = 0; b0 < B; b0 += IP_MAXB) {
    CropArgs a = {};
    a.nb = B - b0 < IP_MAXB ? B - b0 : IP_MAXB;
    for (int j = 0; j < a.nb; ++j) {
      if (!src[b0 + j]) return NNZ_EINVAL;
      a.src[j] = src[b0 + j];
      for (int k = 0; k < 3; ++k) {
        a.shape[j][k] = shapes[(b0 + j) * 3 + k];
        a.lb[j][k] = lbs[(b0 + j) * 3 + k];
      }
      a.flip[j] = flips ? flips[b0 + j] : 0;
    }
    a.dst = static_cast<char*>(dst) + (long)b0 * C * vol * sizeof(T);
    a.C = C; a.pd = pd; a.ph = ph; a.pw = pw; a.pad_f = pad_f; a.pad_i = pad_i;
    long blocks = (vol + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    NNZ_LAUNCH(crop_pad_kernel<T>, dim3((unsigned)blocks, C, a.nb), dim3(256), 0, s, a);
    NNZ_LAUNCH_CHECK();
  }
  return NNZ_OK;
}

}  // namespace nnz

// src: B HOST-side entries, each a device pointer to a resident case [C][sd][sh][sw]; shapes, lbs: B x 3 host ints;
// flips: B host ints or NULL; dst: device [B][C][pd][ph][pw].  2-D data: pd = 1 and depth extent 1.
extern "C" int nnz_crop_pad_f32(const void* const* src, const int* shapes, const int* lbs, const int* flips, float* dst, int B,
                                int C, int pd, int ph, int pw, float pad_value, void* stream) {
  return nnz::crop_pad_launch<float>(src, shapes, lbs, flips, dst, B, C, pd, ph, pw, pad_value, 0, (hipStream_t)stream);
}
extern "C" int nnz_crop_pad_i16(const void* const* src, const int* shapes, const int* lbs, const int* flips, short* dst, int B,
                                int C, int pd, int ph, int pw, int pad_value, void* stream) {
  return nnz::crop_pad_launch<short>(src, shapes, lbs, flips, dst, B, C, pd, ph, pw, 0.f, pad_value, (hipStream_t)stream);
}
// src [nc][id][ih][iw] int16 -> dst [nc][od][oh][ow], torch interpolate(mode='nearest-exact') index rule
extern "C" int nnz_downsample_nearest_i16(const short* src, short* dst, long nc, int id, int ih, int iw, int od, int oh, int ow,
                                          void* stream) {
  using namespace nnz;
  if (!src || !dst || nc < 1 || id < 1 || ih < 1 || iw < 1 || od < 1 || oh < 1 || ow < 1) return NNZ_EINVAL;
  long blocks = (nc * od * oh * ow + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(downsample_nearest_i16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, nc, id, ih,
             iw, od, oh, ow);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
