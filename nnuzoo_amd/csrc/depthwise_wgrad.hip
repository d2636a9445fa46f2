// Weight (and bias) gradient of the depthwise 3x3 convolutions of the 2-D zoo nets (stride 1, padding = dilation):
//     dW[c][ky][kx] = sum_{b, y, x} dy[b][c][y][x] * in[b][c][y + (ky-1) d][x + (kx-1) d],     db[c] = sum dy[b][c]
// Layers: the depthwise half of get_dwconv_layer and the `Convolution(..., groups=C)` blocks of
// /root/reference/nnunetv2/nets/ssnd2net.py (GSC / SSND convnd) and nets/light_mamba2net.py (ResMambaBlock, GSC), which the
// reference runs through cuDNN's grouped convolution.
//
// Why a kernel of its own: both library routes are far off the memory roofline for this shape of problem - MIOpen turns it
// into a batched GEMM (72 ms for 32 channels @ 512^2), ATen's direct kernel (what round 2 first switched to) gives every
// (channel, tap) ONE workgroup that walks the whole batch x plane: 110-126 us per call on average, 350 calls = 38 ms of the
// 266 ms SSND2Net step and 28 of the 297 ms LightMamba2Net step (profiles/r02_ssnd2net_graph_kernels.txt).  The problem is
// a streaming reduction: each input and gradient plane is needed once.
//
// Mapping: workgroup = one band of rows of one (batch, channel) plane; a thread walks the band's pixels (4 at a time when
// the plane allows vector loads) with stride 256 (coalesced along x), reads dy once and its 9 neighbours of `in` (L1 / L2 hits: the band's rows are re-read by the same
// workgroup only), keeps 9 + 1 fp32 sums; wave reduction, four waves through LDS, one partial per (channel, band), then a
// second tiny kernel sums the partials in a fixed order (deterministic: no atomics).  HBM-bound by construction:
// algorithmic bytes = 2 planes read once = 2 * B * C * H * W * sizeof(T).
#include "common.hpp"

namespace nnz {

constexpr int DW_ACC = 10;   // 9 taps + bias

// 4 consecutive pixels of a row as floats (one 16-byte / 8-byte load)
__device__ __forceinline__ void dw_load4(const float* p, float (&v)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void dw_load4(const f16* p, float (&v)[4]) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 t = *reinterpret_cast<const h4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

// VEC (dilation 1, W a multiple of 4, 16-byte aligned planes): a thread owns 4 consecutive pixels - per input row one vector
// load and two halo scalars instead of 12 scalar loads (the scalar form is bound by load instructions, not by bytes:
// 96 us for 32 channels @ 512^2 fp16 where the bytes take 8)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void dw_wgrad_partial_kernel(const T* __restrict__ in, const T* __restrict__ dy,
                                                               float* __restrict__ part, int C, int H, int W, int dil,
                                                               int bands, int rows_per_band) {
  __shared__ float red[4][DW_ACC];
  const int c = blockIdx.y;
  const int b = blockIdx.x / bands, band = blockIdx.x % bands;
  const int y0 = band * rows_per_band;
  const int y1 = y0 + rows_per_band < H ? y0 + rows_per_band : H;
  const long plane = ((long)b * C + c) * H * W;
  const T* ip = in + plane;
  const T* gp = dy + plane;
  float acc[DW_ACC];
#pragma unroll
  for (int t = 0; t < DW_ACC; ++t) acc[t] = 0.f;
  if (VEC) {
    const int W4 = W >> 2;
    const int n4 = (y1 - y0) * W4;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const int yl = i / W4;
      const int x = (i - yl * W4) << 2;
      const int y = y0 + yl;
      float g[4];
      dw_load4(gp + (long)y * W + x, g);
      acc[9] += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        const bool yin = yy >= 0 && yy < H;
        const T* row = ip + (long)(yin ? yy : y) * W;
        float v[6], m[4];
        dw_load4(row + x, m);
        const float lft = x > 0 ? (float)row[x - 1] : 0.f;
        const float rgt = x + 4 < W ? (float)row[x + 4] : 0.f;
        v[0] = yin ? lft : 0.f;
        v[5] = yin ? rgt : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[1 + j] = yin ? m[j] : 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] += g[j] * v[j + kx];
      }
    }
  } else {
    const int npx = (y1 - y0) * W;
    for (int i = threadIdx.x; i < npx; i += 256) {
      const int yl = i / W;
      const int x = i - yl * W;
      const int y = y0 + yl;
      const float g = (float)gp[(long)y * W + x];
      acc[9] += g;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + (ky - 1) * dil;
        const bool yin = yy >= 0 && yy < H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = x + (kx - 1) * dil;
          const bool ok = yin && xx >= 0 && xx < W;
          const float v = ok ? (float)ip[(long)(ok ? yy : 0) * W + (ok ? xx : 0)] : 0.f;
          acc[ky * 3 + kx] += g * v;
        }
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < DW_ACC; ++t) {
    const float s = wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < DW_ACC) {
    const int t = threadIdx.x;
    part[((long)c * gridDim.x + blockIdx.x) * DW_ACC + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
}

// one wave per channel: lanes stride over the channel's partials, fixed order -> the same bits every run
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                              float* __restrict__ db, int C, int nsplit) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  float acc[DW_ACC];
#pragma unroll
  for (int t = 0; t < DW_ACC; ++t) acc[t] = 0.f;
  for (int s = lane; s < nsplit; s += 64) {
#pragma unroll
    for (int t = 0; t < DW_ACC; ++t) acc[t] += part[((long)c * nsplit + s) * DW_ACC + t];
  }
#pragma unroll
  for (int t = 0; t < DW_ACC; ++t) {
    const float s = wave_sum(acc[t]);
    if (lane == 0) {
      if (t < 9) dw[c * 9 + t] = s;
      else if (db) db[c] = s;
    }
  }
}

// bands per plane: enough workgroups to fill the chip (>= ~2048), at least 2048 pixels per band (the block reduction at
// the end of a workgroup is a fixed cost: 1-pixel-per-thread workgroups ran at 17 us for 1 M pixels)
static int dw_bands(int B, int C, int H, int W) {
  int bands = 1;
  while ((long)B * C * bands < 2048 && bands * 2 <= H && ((long)H / (bands * 2)) * W >= 2048) bands *= 2;
  return bands;
}

}  // namespace nnz

extern "C" long nnz_dwconv2d_wgrad_workspace_floats(int B, int C, int H, int W) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
  return (long)C * B * nnz::dw_bands(B, C, H, W) * nnz::DW_ACC;
}

// in, dy: [B][C][H][W] contiguous, fp16 (is_f16) or fp32; dw: [C][3][3] fp32; db: [C] fp32 or NULL;
// workspace: nnz_dwconv2d_wgrad_workspace_floats floats
extern "C" int nnz_dwconv2d_wgrad(const void* in, const void* dy, int is_f16, float* workspace, float* dw, float* db, int B,
                                  int C, int H, int W, int dilation, void* stream) {
  using namespace nnz;
  if (!in || !dy || !workspace || !dw || B < 1 || C < 1 || C > 65535 || H < 1 || W < 1 || dilation < 1 ||
      (long)H * W > (1L << 30))
    return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int bands = dw_bands(B, C, H, W);
  const int rows_per_band = (H + bands - 1) / bands;
  const dim3 grid((unsigned)(B * bands), (unsigned)C);
  const bool vec = dilation == 1 && W % 4 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)dy % 16) == 0;
#define DW_GO(T, V)                                                                                               \
  NNZ_LAUNCH((dw_wgrad_partial_kernel<T, V>), grid, dim3(256), 0, s, static_cast<const T*>(in), static_cast<const T*>(dy), \
             workspace, C, H, W, dilation, bands, rows_per_band)
  if (is_f16) {
    if (vec) DW_GO(f16, true); else DW_GO(f16, false);
  } else {
    if (vec) DW_GO(float, true); else DW_GO(float, false);
  }
#undef DW_GO
  NNZ_LAUNCH_CHECK();
  NNZ_LAUNCH(dw_wgrad_reduce_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, workspace, dw, db, C, B * bands);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
