// Weight layout plumbing for the tap-table convolutions:
//   nnz_pack_conv_weight   : torch fp32 parameter (any strides) -> Wp[R/16][C/32][T][32][16] fp16
//   nnz_unpack_conv_wgrad  : dW[T][A][B] fp32 -> torch-layout fp32 gradient (any strides)
// Parameters keep the reference's names and shapes (state_dict compatibility,
// /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1291-1352); these kernels are the only place
// that knows the packed layout.
#include "common.hpp"
#include <string.h>

namespace nnz {

struct KselTable {
  int ksel[32];
};

__global__ void pack_weight_kernel(const float* __restrict__ src, f16* __restrict__ dst, int R, int C, int T,
                                   long sr, long sc, long sk, KselTable tab) {
  // one thread per packed element
  const long total = (long)R * C * T;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r16 = i & 15;
    long q = i >> 4;
    const int c32 = q & 31;
    q >>= 5;
    const int t = q % T;
    q /= T;
    const int cb = q % (C >> 5);
    const int rb = q / (C >> 5);
    const int r = rb * 16 + r16, c = cb * 32 + c32;
    dst[i] = (f16)src[r * sr + c * sc + tab.ksel[t] * sk];
  }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ grad, int A, int B, int T,
                                    long sa, long sb, long sk, KselTable tab, int accumulate) {
  const long total = (long)T * A * B;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i % B;
    const long q = i / B;
    const int a = q % A;
    const int t = q / A;
    float* g = grad + a * sa + b * sb + tab.ksel[t] * sk;
    *g = accumulate ? *g + dw[i] : dw[i];
  }
}

// One launch packs every layer's weights: a device-resident job table (built once per plan: parameter and packed
// buffer addresses are stable across steps) replaces ~50 tiny launches per step.
struct PackJob {
  const float* src;
  f16* dst;
  int R, C, T;
  int pad;
  long sr, sc, sk;
  int ksel[32];
};

// One workgroup per (16 reduction channels) x (32 output channels) unit: its packed image is ONE contiguous block of
// T KiB.  The 16 x 32 x T source elements are read in source order (the faster-varying of the two channel strides
// innermost, taps contiguous for the torch conv layouts), converted and scattered into an LDS image of the block,
// which is then streamed out as whole 16-byte pieces (the previous wave-per-channel version wrote 32-byte runs at a
// 1 KiB stride and ran at 0.4 TB/s).
__global__ __launch_bounds__(256) void pack_weight_batched_kernel(const PackJob* __restrict__ jobs) {
  __shared__ __attribute__((aligned(16))) f16 img[32 * 512];
  const PackJob j = jobs[blockIdx.y];
  const int tid = threadIdx.x;
  const int CB = j.C >> 5;
  const int units = CB * (j.R >> 4);
  const int T = j.T;
  const int nelem = 512 * T;
  const bool r_inner = j.sr < j.sc;  // which channel index walks the smaller stride
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int cb = u % CB, rb = u / CB;
    const float* src = j.src + (long)(rb * 16) * j.sr + (long)(cb * 32) * j.sc;
    for (int e = tid; e < nelem; e += 256) {
      const int t = e % T;
      const int q = e / T;
      int r16, c32;
      if (r_inner) {
        r16 = q & 15;
        c32 = q >> 4;
      } else {
        c32 = q & 31;
        r16 = q >> 5;
      }
      img[t * 512 + c32 * 16 + r16] = (f16)src[(long)r16 * j.sr + (long)c32 * j.sc + (long)j.ksel[t] * j.sk];
    }
    __syncthreads();
    u32x4* dst = reinterpret_cast<u32x4*>(j.dst + (long)u * nelem);
    const u32x4* im = reinterpret_cast<const u32x4*>(img);
    for (int o = tid; o < nelem / 8; o += 256) dst[o] = im[o];
    __syncthreads();
  }
}

}  // namespace nnz

extern "C" int nnz_pack_job_bytes(void) { return (int)sizeof(nnz::PackJob); }

// host helper: serialise one job into `out` (nnz_pack_job_bytes() bytes) for upload into the device job table
extern "C" int nnz_pack_job_fill(void* out, const float* src, void* dst_f16, int R, int C, int T, long sr, long sc,
                                 long sk, const int* ksel) {
  using namespace nnz;
  if (!out || !src || !dst_f16 || !ksel || R % 16 || C % 32 || T < 1 || T > 32) return NNZ_EINVAL;
  PackJob j = {};
  j.src = src; j.dst = (f16*)dst_f16; j.R = R; j.C = C; j.T = T; j.sr = sr; j.sc = sc; j.sk = sk;
  for (int i = 0; i < 32; ++i) j.ksel[i] = i < T ? ksel[i] : 0;
  memcpy(out, &j, sizeof(j));
  return NNZ_OK;
}

extern "C" int nnz_pack_conv_weights_batched(const void* jobs_device, int njobs, void* stream) {
  using namespace nnz;
  if (!jobs_device || njobs < 1) return NNZ_EINVAL;
  NNZ_LAUNCH(pack_weight_batched_kernel, dim3(128, njobs), dim3(256), 0, (hipStream_t)stream,
                     (const PackJob*)jobs_device);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_pack_conv_weight(const float* src, void* dst_f16, int R, int C, int T, long sr, long sc, long sk,
                                    const int* ksel, void* stream) {
  using namespace nnz;
  if (!src || !dst_f16 || !ksel || R % 16 || C % 32 || T < 1 || T > 32) return NNZ_EINVAL;
  KselTable tab;
  for (int i = 0; i < 32; ++i) tab.ksel[i] = i < T ? ksel[i] : 0;
  const long total = (long)R * C * T;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  NNZ_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (f16*)dst_f16, R, C, T,
                     sr, sc, sk, tab);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_unpack_conv_wgrad(const float* dw, float* grad, int A, int B, int T, long sa, long sb, long sk,
                                     const int* ksel, int accumulate, void* stream) {
  using namespace nnz;
  if (!dw || !grad || !ksel || T < 1 || T > 32) return NNZ_EINVAL;
  KselTable tab;
  for (int i = 0; i < 32; ++i) tab.ksel[i] = i < T ? ksel[i] : 0;
  const long total = (long)A * B * T;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  NNZ_LAUNCH(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dw, grad, A, B, T, sa, sb,
                     sk, tab, accumulate);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
