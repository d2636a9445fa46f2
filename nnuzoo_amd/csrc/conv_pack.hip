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

// ---- both packed forms of every layer from ONE read of the parameters (round 3) ------------------------------------------
// The forward convolution wants Wf[X/16][Y/32][T][32 y][16 x] (X = input channels = its reduction), the data gradient
// Wb[Y/16][X/32][T][32 x][16 y] with its own tap order.  Rounds 1-2 packed them in two launches (start of forward / of
// backward), each re-reading all 125 MB of fp32 parameters with 4-byte loads: 0.39 ms per step at ~1 TB/s.  Here a
// workgroup owns one 32 x 32 channel block of one layer for ALL kernel positions: it streams the block in with 16-byte
// loads along the parameter's contiguous runs (32 inner channels x nk taps = one run per outer channel), converts to fp16
// into one LDS image [k][x][y] and streams both packed blocks out as whole 16-byte pieces.
struct DualPackJob {
  const float* src;
  f16* dst_f;
  f16* dst_b;
  int X, Y, nk, x_inner;  // x_inner: the X axis has stride nk (conv weights (Y, X, k)); else the Y axis has (convT (X, Y, k))
  int first_block;        // prefix sum of (X/32) * (Y/32) over the jobs before this one
  int pad;
  int ksel_f[32];
  int ksel_b[32];
};

constexpr int DP_PITCH = 40;  // f16 per x row of the LDS image: 80 bytes keeps the 8-y pieces 16-byte aligned

__global__ __launch_bounds__(256) void pack_dual_kernel(const DualPackJob* __restrict__ jobs, int njobs) {
  extern __shared__ __attribute__((aligned(16))) f16 img[];  // [nk][32 x][DP_PITCH]
  __shared__ int s_job;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].first_block <= (int)blockIdx.x) ++j;
    s_job = j;
  }
  __syncthreads();
  const DualPackJob& J = jobs[s_job];
  const int nk = J.nk, X = J.X, Y = J.Y;
  const int blk = blockIdx.x - J.first_block;
  const int nyb = Y >> 5;
  const int xb = blk / nyb, yb = blk % nyb;
  const int inner_n = J.x_inner ? X : Y;                 // channels along the contiguous axis
  const int ib = J.x_inner ? xb : yb, ob = J.x_inner ? yb : xb;
  const int row_f4 = 8 * nk;                             // float4 per outer row of the block (32 * nk floats)
  const int total = 32 * row_f4;
  const float* base = J.src + ((size_t)(ob * 32) * inner_n + ib * 32) * nk;
  for (int p0 = tid; p0 < total; p0 += 4 * 256) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * 256;
      if (p < total) {
        const int row = p / row_f4, q = p - row * row_f4;
        v[u] = *reinterpret_cast<const f32x4*>(base + (size_t)row * inner_n * nk + 4 * q);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * 256;
      if (p < total) {
        const int row = p / row_f4, q = p - row * row_f4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 4 * q + e;
          const int inner = f / nk, kk = f - inner * nk;
          const int x = J.x_inner ? inner : row, y = J.x_inner ? row : inner;
          img[(kk * 32 + x) * DP_PITCH + y] = (f16)v[u][e];
        }
      }
    }
  }
  __syncthreads();
  const int npiece = nk * 128;  // 16-byte pieces per packed form: 2 halves x nk x 32 x 2
  // forward form: unit (xb * 2 + xh, yb), piece = [t][y][xo] holding x = xh * 16 + xo * 8 .. + 7 (stride DP_PITCH in LDS)
  for (int p = tid; p < npiece; p += 256) {
    const int xo = p & 1, y = (p >> 1) & 31, r = p >> 6;
    const int t = r % nk, xh = r / nk;
    const f16* s0 = img + (J.ksel_f[t] * 32 + xh * 16 + xo * 8) * DP_PITCH + y;
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = s0[j * DP_PITCH];
    f16* d = J.dst_f + ((size_t)((xb * 2 + xh) * nyb + yb) * nk + t) * 512 + y * 16 + xo * 8;
    *reinterpret_cast<f16x8*>(d) = o;
  }
  // data-gradient form: unit (yb * 2 + yh, xb), piece = [t][x][yo] holding y = yh * 16 + yo * 8 .. + 7 (contiguous in LDS)
  const int nxb = X >> 5;
  for (int p = tid; p < npiece; p += 256) {
    const int yo = p & 1, x = (p >> 1) & 31, r = p >> 6;
    const int t = r % nk, yh = r / nk;
    const f16x8 o = *reinterpret_cast<const f16x8*>(img + (J.ksel_b[t] * 32 + x) * DP_PITCH + yh * 16 + yo * 8);
    f16* d = J.dst_b + ((size_t)((yb * 2 + yh) * nxb + xb) * nk + t) * 512 + x * 16 + yo * 8;
    *reinterpret_cast<f16x8*>(d) = o;
  }
}

}  // namespace nnz

extern "C" int nnz_pack_dual_job_bytes(void) { return (int)sizeof(nnz::DualPackJob); }

// host helper: serialise one job (first_block = number of 32 x 32 blocks of all earlier jobs)
extern "C" int nnz_pack_dual_job_fill(void* out, const float* src, void* dst_fwd_f16, void* dst_dgrad_f16, int X, int Y,
                                      int nk, int x_inner, int first_block, const int* ksel_fwd, const int* ksel_dgrad) {
  using namespace nnz;
  if (!out || !src || !dst_fwd_f16 || !dst_dgrad_f16 || !ksel_fwd || !ksel_dgrad || X % 32 || Y % 32 || nk < 1 || nk > 27)
    return NNZ_EINVAL;
  DualPackJob j = {};
  j.src = src; j.dst_f = (f16*)dst_fwd_f16; j.dst_b = (f16*)dst_dgrad_f16;
  j.X = X; j.Y = Y; j.nk = nk; j.x_inner = x_inner; j.first_block = first_block;
  for (int i = 0; i < 32; ++i) {
    j.ksel_f[i] = i < nk ? ksel_fwd[i] : 0;
    j.ksel_b[i] = i < nk ? ksel_dgrad[i] : 0;
    if (j.ksel_f[i] < 0 || j.ksel_f[i] >= nk || j.ksel_b[i] < 0 || j.ksel_b[i] >= nk) return NNZ_EINVAL;
  }
  memcpy(out, &j, sizeof(j));
  return NNZ_OK;
}

extern "C" int nnz_pack_dual_batched(const void* jobs_device, int njobs, int total_blocks, int max_nk, void* stream) {
  using namespace nnz;
  if (!jobs_device || njobs < 1 || total_blocks < 1 || max_nk < 1 || max_nk > 27) return NNZ_EINVAL;
  const int lds = max_nk * 32 * DP_PITCH * 2;  // <= 69 120 bytes
  static DynLdsCache lds_cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(pack_dual_kernel), lds, lds_cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(pack_dual_kernel, dim3(total_blocks), dim3(256), lds, (hipStream_t)stream,
             (const DualPackJob*)jobs_device, njobs);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

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
