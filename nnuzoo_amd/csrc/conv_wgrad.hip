// Weight-gradient contraction of the tap-table convolution on MFMA (gfx950):
//   dW[widx_t][a][b] = sum_n sum_m  P[n, m*IS + off_t, a] * Q[n, m*OS + ooff, b]
// P is the "boxed" operand (the one the taps shift: the layer input X for a convolution, the output
// gradient for a transposed convolution), Q the plain one.  Replaces the bwd-weight kernels torch
// autograd runs for nn.Conv3d / nn.ConvTranspose3d inside the reference's training step
// (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1112-1144, `.backward()`).
//
// Design (MI355X-first):
//   * The contraction index is the VOXEL, which is the strided axis of a channels-last tensor, so both
//     MFMA operands are fetched with ds_read_b64_tr_b16 (hardware-transposed LDS reads) from plain
//     [voxel][32 ch] images: four consecutive voxels x 32 channels = one conflict-free 256-byte row.
//   * A workgroup owns a (32 boxed-channel x 32 plain-channel) block pair for ALL taps: wave w keeps the
//     accumulators of taps w, w+4, ... (<= 7 x 16 registers) for the whole launch and walks a strided
//     share of the m-tiles (persistent), so the only cross-workgroup reduction is one fp32 atomic add of
//     the [T][32][32] block per workgroup (a few MB chip-wide instead of GBs; Guideline 12).
//   * Next tile's box/tile are prefetched into registers while the current one is in the MFMA loop.
#include "common.hpp"
#include "conv_params.h"

namespace nnz {

struct WgradDev {
  const f16* p;  // boxed operand
  const f16* q;  // plain operand
  float* dw;     // [T][A][B] fp32, zeroed by the launcher
  nnz_conv_desc d;  // Cin = A (boxed channels), Cout = B (plain channels), ldi/ldo their strides
  int tiles[3];
  int ntiles;   // per launch: N * tiles
  int splits;
  int pairs_b;  // B/32
};

template <int TD, int TH, int TW, int IS, int EXT>
struct WgCfg {
  static constexpr int BD = (TD - 1) * IS + EXT + 1;
  static constexpr int BH = (TH - 1) * IS + EXT + 1;
  static constexpr int BW = (TW - 1) * IS + EXT + 1;
  static constexpr int NVOX = TD * TH * TW;
  static constexpr int BOX_BYTES = BD * BH * BW * 64;
  static constexpr int Q_BYTES = NVOX * 64;
  static constexpr int NBOXLOAD = BD * BH * BW * 4;
  static constexpr int NQLOAD = NVOX * 4;
  static constexpr int LPT_BOX = (NBOXLOAD + 255) / 256;
  static constexpr int LPT_Q = (NQLOAD + 255) / 256;
  static constexpr int KB = NVOX / 16;
  static constexpr int LDS_BYTES = BOX_BYTES + Q_BYTES;
  static_assert(TW == 8, "k-block map assumes TW == 8");
};

template <int TD, int TH, int TW, int IS, int EXT, int MAXT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradDev p) {
  using C = WgCfg<TD, TH, TW, IS, EXT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* box = smem;
  char* qt = smem + C::BOX_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;

  const int pair = blockIdx.x / p.splits;
  const int split = blockIdx.x % p.splits;
  const int a0 = (pair / p.pairs_b) * 32;
  const int b0 = (pair % p.pairs_b) * 32;

  const int T = p.d.ntaps_total;
  const int Di = p.d.in_dims[0], Hi = p.d.in_dims[1], Wi = p.d.in_dims[2];
  const int Do = p.d.out_dims[0], Ho = p.d.out_dims[1], Wo = p.d.out_dims[2];
  const int tiles_per_n = p.tiles[0] * p.tiles[1] * p.tiles[2];

  // taps of this wave: wave, wave+4, ...  (MAXT = 7 for 27 taps, 2 for the 8 taps of the k2s2 transpose)
  int tap_off[MAXT];
  int ntw = 0;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + 4 * i;
    tap_off[i] = 0;
    if (t < T) {
      const nnz_conv_tap tp = p.d.taps[t];
      tap_off[i] = (((tp.off[0] - p.d.lo[0]) * C::BH + (tp.off[1] - p.d.lo[1])) * C::BW + (tp.off[2] - p.d.lo[2])) * 64;
      ntw = i + 1;
    }
  }

  // tr16 addressing: lane supplies row (voxel) q = (lane&15)>>2 of its group's 4-row block, channel quad
  // 16*((lane>>4)&1) + 4*(lane&3); the k index inside a 16-voxel block is 8*hh + 4*s + q.
  const int qrow = (lane & 15) >> 2;
  const int chan_byte = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  f32x16 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  u32x4 breg[C::LPT_BOX];
  u32x4 qreg[C::LPT_Q];

  auto issue_loads = [&](int tile) {
    const int n = tile / tiles_per_n;
    int r = tile - n * tiles_per_n;
    const int tw_i = r % p.tiles[2];
    r /= p.tiles[2];
    const int th_i = r % p.tiles[1];
    const int td_i = r / p.tiles[1];
    const int m0d = td_i * TD, m0h = th_i * TH, m0w = tw_i * TW;
    const int lod = m0d * IS + p.d.lo[0], loh = m0h * IS + p.d.lo[1], low = m0w * IS + p.d.lo[2];
#pragma unroll
    for (int i = 0; i < C::LPT_BOX; ++i) {
      const int c = tid + i * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < C::NBOXLOAD) {
        const int part = c & 3;
        const int s = c >> 2;
        const int bw = s % C::BW;
        const int bh = (s / C::BW) % C::BH;
        const int bd = s / (C::BW * C::BH);
        const int id = lod + bd, ih = loh + bh, iw = low + bw;
        if ((unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi)
          v = *reinterpret_cast<const u32x4*>(p.p + ((size_t)((n * Di + id) * Hi + ih) * Wi + iw) * p.d.ldi + a0 +
                                              part * 8);
      }
      breg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < C::LPT_Q; ++i) {
      const int c = tid + i * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (c < C::NQLOAD) {
        const int part = c & 3;
        const int s = c >> 2;
        const int tw = s % TW, th = (s / TW) % TH, td = s / (TW * TH);
        const int md = m0d + td, mh = m0h + th, mw = m0w + tw;
        const int od = md * p.d.out_stride + p.d.groups[0].ooff[0];
        const int oh = mh * p.d.out_stride + p.d.groups[0].ooff[1];
        const int ow = mw * p.d.out_stride + p.d.groups[0].ooff[2];
        if (md < p.d.m_dims[0] && mh < p.d.m_dims[1] && mw < p.d.m_dims[2] && od < Do && oh < Ho && ow < Wo)
          v = *reinterpret_cast<const u32x4*>(p.q + ((size_t)((n * Do + od) * Ho + oh) * Wo + ow) * p.d.ldo + b0 +
                                              part * 8);
      }
      qreg[i] = v;
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int i = 0; i < C::LPT_BOX; ++i) {
      const int c = tid + i * 256;
      if (c < C::NBOXLOAD) *reinterpret_cast<u32x4*>(box + c * 16) = breg[i];
    }
#pragma unroll
    for (int i = 0; i < C::LPT_Q; ++i) {
      const int c = tid + i * 256;
      if (c < C::NQLOAD) *reinterpret_cast<u32x4*>(qt + c * 16) = qreg[i];
    }
  };

  int tile = split;
  if (tile < p.ntiles) issue_loads(tile);
  for (; tile < p.ntiles; tile += p.splits) {
    __syncthreads();
    write_lds();
    __syncthreads();
    if (tile + p.splits < p.ntiles) issue_loads(tile + p.splits);

    for (int kb = 0; kb < C::KB; ++kb) {
      // voxel of (kb, kk = 8*hh + 4*s + qrow): two h-rows of 8 w per k-block
      const int v0 = kb * 16 + 8 * hh;  // first voxel of this lane-half's row; tw = 4*s + qrow
      const int th = (v0 / TW) % TH, td = v0 / (TW * TH);
      const int qbase = (v0 + qrow) * 64 + chan_byte;
      const i16x4 q0 = lds_read_tr16(qt + qbase);
      const i16x4 q1 = lds_read_tr16(qt + qbase + 4 * 64);
      f16x8 bq;
      {
        union { i16x4 v[2]; f16x8 h; } u;
        u.v[0] = q0;
        u.v[1] = q1;
        bq = u.h;
      }
      const int bbase = ((((td * IS) * C::BH + th * IS) * C::BW) + qrow * IS) * 64 + chan_byte;
      // branch-free over the wave's MAXT tap slots (a slot beyond the wave's taps re-reads tap offset 0 and is
      // never flushed): all 2*MAXT transposed reads issue back to back ahead of the MFMAs instead of one
      // read->wait->MFMA chain per tap (SQ_WAIT_INST_ANY was 36-58 % of the wave cycles with the guarded loop)
      f16x8 xa[MAXT];
#pragma unroll
      for (int i = 0; i < MAXT; ++i) {
        union { i16x4 v[2]; f16x8 h; } u;
        u.v[0] = lds_read_tr16(box + bbase + tap_off[i]);
        u.v[1] = lds_read_tr16(box + bbase + tap_off[i] + 4 * IS * 64);
        xa[i] = u.h;
      }
#pragma unroll
      for (int i = 0; i < MAXT; ++i) acc[i] = mfma32(xa[i], bq, acc[i]);
    }
  }

  // ---- flush: D[row = a][col = b]; row = (r&3) + 8(r>>2) + 4hh, col = lane&31 -----------------------
  const int A = p.d.Cin, B = p.d.Cout;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + 4 * i;
    if (i < ntw) {
      const int widx = p.d.taps[t].widx;
      float* dst = p.dw + ((size_t)widx * A + a0) * B + b0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        atomicAdd(dst + (size_t)row * B, acc[i][r]);
      }
    }
  }
}

template <int TD, int TH, int TW, int IS, int EXT, int MAXT>
static int launch_wg_t(const WgradDev& base, hipStream_t stream, bool pre_zeroed) {
  using C = WgCfg<TD, TH, TW, IS, EXT>;
  WgradDev p = base;
  p.tiles[0] = (p.d.m_dims[0] + TD - 1) / TD;
  p.tiles[1] = (p.d.m_dims[1] + TH - 1) / TH;
  p.tiles[2] = (p.d.m_dims[2] + TW - 1) / TW;
  p.ntiles = p.d.N * p.tiles[0] * p.tiles[1] * p.tiles[2];
  p.pairs_b = p.d.Cout / 32;
  const int pairs = (p.d.Cin / 32) * p.pairs_b;
  int splits = 1024 / pairs;
  if (splits < 1) splits = 1;
  if (splits > p.ntiles) splits = p.ntiles;
  p.splits = splits;
  auto kern = conv_wgrad_kernel<TD, TH, TW, IS, EXT, MAXT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (!pre_zeroed) {
    hipError_t e = hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.d.ntaps_total * p.d.Cin * p.d.Cout, stream);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(pairs * splits), dim3(256), C::LDS_BYTES, stream, p);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <int TD, int TH, int TW, int IS, int EXT>
static int launch_wg(const WgradDev& p, hipStream_t stream, bool pre_zeroed) {
  const int per_wave = (p.d.ntaps_total + 3) / 4;
  if (per_wave <= 2) return launch_wg_t<TD, TH, TW, IS, EXT, 2>(p, stream, pre_zeroed);
  return launch_wg_t<TD, TH, TW, IS, EXT, 7>(p, stream, pre_zeroed);
}

}  // namespace nnz

extern "C" int nnz_conv_tap_wgrad(const void* boxed, const void* plain, float* dw, const nnz_conv_desc* desc,
                                  int pre_zeroed, void* stream) {
  using namespace nnz;
  if (!boxed || !plain || !dw || !desc) return NNZ_EINVAL;
  const nnz_conv_desc& d = *desc;
  if (d.Cin % 32 || d.Cout % 32 || d.ngroups != 1 || d.ntaps_total > 28 || d.ntaps_total < 1 || d.ldi % 8 ||
      d.ldo % 8 || (d.in_stride != 1 && d.in_stride != 2) || d.ext < 0 || d.ext > 2)
    return NNZ_EINVAL;
  WgradDev p;
  p.p = (const f16*)boxed;
  p.q = (const f16*)plain;
  p.dw = dw;
  p.d = d;
  hipStream_t s = (hipStream_t)stream;
  if (d.in_stride == 1) {
    if (d.ext == 0) return launch_wg<4, 8, 8, 1, 0>(p, s, pre_zeroed != 0);
    if (d.ext == 1) return launch_wg<4, 8, 8, 1, 1>(p, s, pre_zeroed != 0);
    return launch_wg<4, 8, 8, 1, 2>(p, s, pre_zeroed != 0);
  } else {
    if (d.ext <= 1) return launch_wg<2, 4, 8, 2, 1>(p, s, pre_zeroed != 0);
    return launch_wg<2, 4, 8, 2, 2>(p, s, pre_zeroed != 0);
  }
}
