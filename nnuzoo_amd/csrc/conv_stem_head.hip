// The two HBM-bound ends of the PlainConvUNet that do not fit an MFMA tile (gfx950):
//   * stem : Conv3d(1 -> 32, k3, p1) on the fp32 input patch, forward + weight gradient
//            (first block of encoder stage 0; no data gradient is needed for the network input)
//   * head : 1x1x1 segmentation convs (C -> K classes, K <= 8) writing NCDHW logits, forward,
//            data gradient and weight/bias gradient (deep-supervision outputs,
//            /root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1010-1022 toggles them)
// Arithmetic follows autocast: operands rounded to fp16, fp32 accumulate.
#include "common.hpp"

namespace nnz {

// ------------------------------------------------------------------------------------------------ stem
constexpr int ST_TD = 4, ST_TH = 8, ST_TW = 8;
constexpr int ST_BD = ST_TD + 2, ST_BH = ST_TH + 2, ST_BW = ST_TW + 2;
constexpr int ST_CO = 32;

struct StemArgs {
  const float* x;   // [N][D][H][W] fp32
  const float* w;   // [32][27] fp32 (torch (32,1,3,3,3))
  const float* b;   // [32]
  f16* y;           // [N][V][ldy]
  const f16* dy;    // [N][V][lddy]   (wgrad)
  float* dw;        // [32][27] fp32  (wgrad, zeroed by launcher)
  int N, D, H, W, ldy, lddy;
  int tiles[3];
  FxAcc* acc;        // deterministic weight gradient: 27 * 32 fixed-point accumulators (common.hpp) + launch counter
  unsigned* counter;
};

// Stem forward on MFMA (a VALU version issued 864 v_fma + 216 broadcast LDS reads per voxel and ran 4x off the HBM
// bound): D[row = cout][col = voxel] = W[cout][tap] * X[tap][voxel] with the 27 taps padded to K = 32
// (two v_mfma_f32_32x32x16_f16 per 32 voxels).  A = the whole weight (32 x 32 fp16: two 16-byte registers per lane,
// loaded once), B is gathered from the fp16 x tile in LDS: lane (voxel, hh) reads its 8 taps as 2-byte loads.  The
// output tile goes through an LDS image so that stores are whole 16-byte pieces (full 64-byte voxel rows).
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(StemArgs a) {
  constexpr int PW = 16;                                   // w pitch of the x tile (halves)
  constexpr int ROWB = ST_CO * 2 + 16;                     // bytes per voxel row of the output image
  __shared__ __attribute__((aligned(16))) f16 xt[ST_BD * ST_BH * PW];
  __shared__ __attribute__((aligned(16))) char img[ST_TD * ST_TH * ST_TW * ROWB];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int bx = blockIdx.x;
  const int tw_i = bx % a.tiles[2];
  bx /= a.tiles[2];
  const int th_i = bx % a.tiles[1];
  const int td_i = bx / a.tiles[1];
  const int n = blockIdx.y;
  const int m0d = td_i * ST_TD, m0h = th_i * ST_TH, m0w = tw_i * ST_TW;
  // x tile (autocast rounding of the input), zero outside the volume
  for (int i = tid; i < ST_BD * ST_BH * ST_BW; i += 256) {
    const int bw = i % ST_BW, bh = (i / ST_BW) % ST_BH, bd = i / (ST_BW * ST_BH);
    const int id = m0d + bd - 1, ih = m0h + bh - 1, iw = m0w + bw - 1;
    float v = 0.f;
    if ((unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
      v = a.x[(((long)n * a.D + id) * a.H + ih) * a.W + iw];
    xt[(bd * ST_BH + bh) * PW + bw] = (f16)v;
  }
  // A operand: row = cout (l31), k = 8*hh + j (+16 for the second MFMA); taps >= 27 are zero
  f16x8 wa[2];
  int toff[2][8];  // LDS element offset of tap k relative to the lane's voxel
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = m * 16 + 8 * hh + j;
      wa[m][j] = k < 27 ? (f16)a.w[l31 * 27 + k] : (f16)0.f;
      const int kk = k < 27 ? k : 0;
      toff[m][j] = ((kk / 9) * ST_BH + (kk / 3) % 3) * PW + kk % 3;
    }
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = a.b ? a.b[(r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
  __syncthreads();
  // each wave: 2 groups of 32 voxels (the tile has 8)
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const int v = (wave * 2 + gi) * 32 + l31;
    const int tw = v % ST_TW, th = (v / ST_TW) % ST_TH, td = v / (ST_TW * ST_TH);
    const int vbase = (td * ST_BH + th) * PW + tw;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[r];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f16x8 xb;
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = xt[vbase + toff[m][j]];
      acc = mfma32(wa[m], xb, acc);
    }
    // lane holds couts (r&3) + 8(r>>2) + 4hh of voxel v
    char* row = img + v * ROWB + (4 * hh) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x4 o = {(f16)acc[4 * q + 0], (f16)acc[4 * q + 1], (f16)acc[4 * q + 2], (f16)acc[4 * q + 3]};
      *reinterpret_cast<f16x4*>(row + 16 * q) = o;
    }
  }
  __syncthreads();
  for (int c = tid; c < ST_TD * ST_TH * ST_TW * 4; c += 256) {
    const int v = c >> 2, part = c & 3;
    const int tw = v % ST_TW, th = (v / ST_TW) % ST_TH, td = v / (ST_TW * ST_TH);
    const int od = m0d + td, oh = m0h + th, ow = m0w + tw;
    if (od < a.D && oh < a.H && ow < a.W)
      *reinterpret_cast<f16x8*>(a.y + ((((long)n * a.D + od) * a.H + oh) * a.W + ow) * a.ldy + part * 8) =
          *reinterpret_cast<const f16x8*>(img + v * ROWB + part * 16);
  }
}

// dW[co][t] = sum_v dy[v][co] * x[v + off_t] as an MFMA contraction over voxels (persistent over tiles):
//   D[row = tap t (27 of 32)][col = co] += A[t][k] * B[k][co],  k = 16 voxels (two h-rows of 8 w)
//   A: lane (t, hh) gathers x[v_k + off_t] for its 8 voxels (8 consecutive w of the fp16 x tile, 2-byte reads);
//   B: dy tile [voxel][32 co] through the transposed LDS read (same addressing as conv_wgrad.hip).
// HBM-bound (one read of dy): the VALU version it replaces ran at 0.45 TB/s.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemArgs a, int ntiles) {
  __shared__ __attribute__((aligned(16))) f16 xt[ST_BD * ST_BH * ST_BW + 8];
  __shared__ __attribute__((aligned(16))) f16 dyt[ST_TD * ST_TH * ST_TW * ST_CO];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;
  const int t = lane & 31;
  const int toff = t < 27 ? ((t / 9) * ST_BH + (t / 3) % 3) * ST_BW + t % 3 : 0;
  const int qrow = (lane & 15) >> 2;
  const int chan_byte = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int tiles_per_n = a.tiles[0] * a.tiles[1] * a.tiles[2];
  // register prefetch of the next tile (x: 3 values, dy: 4 x 16 bytes per thread) under the current tile's MFMAs: the
  // loop used to expose the full global-load latency for 4 MFMAs of work per wave
  constexpr int XL = (ST_BD * ST_BH * ST_BW + 255) / 256;
  float xreg[XL];
  u32x4 dreg[4];
  auto prefetch = [&](int tile) {
    const int n = tile / tiles_per_n;
    int r = tile % tiles_per_n;
    const int tw_i = r % a.tiles[2];
    r /= a.tiles[2];
    const int th_i = r % a.tiles[1];
    const int td_i = r / a.tiles[1];
    const int m0d = td_i * ST_TD, m0h = th_i * ST_TH, m0w = tw_i * ST_TW;
#pragma unroll
    for (int k = 0; k < XL; ++k) {
      const int i = tid + k * 256;
      const int bw = i % ST_BW, bh = (i / ST_BW) % ST_BH, bd = i / (ST_BW * ST_BH);
      const int id = m0d + bd - 1, ih = m0h + bh - 1, iw = m0w + bw - 1;
      float v = 0.f;
      if (i < ST_BD * ST_BH * ST_BW && (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H &&
          (unsigned)iw < (unsigned)a.W)
        v = a.x[(((long)n * a.D + id) * a.H + ih) * a.W + iw];
      xreg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int part = i & 3, v = i >> 2;
      const int tw = v % ST_TW, th = (v / ST_TW) % ST_TH, td = v / (ST_TW * ST_TH);
      const int od = m0d + td, oh = m0h + th, ow = m0w + tw;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (od < a.D && oh < a.H && ow < a.W)
        val = *reinterpret_cast<const u32x4*>(a.dy + ((((long)n * a.D + od) * a.H + oh) * a.W + ow) * a.lddy + part * 8);
      dreg[k] = val;
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XL; ++k) {
      const int i = tid + k * 256;
      if (i < ST_BD * ST_BH * ST_BW) xt[i] = (f16)xreg[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      *reinterpret_cast<u32x4*>(dyt + (i >> 2) * ST_CO + (i & 3) * 8) = dreg[k];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const int kb = wave * 4 + kq;        // 16 k-blocks of 16 voxels, 4 per wave
      const int v0 = kb * 16 + 8 * hh;     // first voxel of this lane half's h-row (tw = 0)
      const int th = (v0 / ST_TW) % ST_TH, td = v0 / (ST_TW * ST_TH);
      const char* qb = reinterpret_cast<const char*>(dyt) + (v0 + qrow) * 64 + chan_byte;
      union { i16x4 v[2]; f16x8 h; } ub;
      ub.v[0] = lds_read_tr16(qb);
      ub.v[1] = lds_read_tr16(qb + 4 * 64);
      const f16* xr = xt + (td * ST_BH + th) * ST_BW + toff;
      f16x8 av;
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = xr[j];
      acc = mfma32(av, ub.h, acc);
    }
  }
  // D[row = t][col = co]: row = (r&3) + 8(r>>2) + 4hh, col = lane & 31.  The 4 waves' partial tiles are summed
  // through LDS (reusing the dy tile) so that the workgroup issues ONE atomic per element: 864 hot addresses
  // shared by every workgroup serialise badly otherwise (MI355X_MICROARCH.md, global float atomics, contention row).
  __syncthreads();
  float* red = reinterpret_cast<float*>(dyt);  // [4 waves][32 t][32 co] fp32 = 16 KB
  const int co = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int tt = (r & 3) + 8 * (r >> 2) + 4 * hh;
    red[(wave * 32 + tt) * 32 + co] = acc[r];
  }
  __syncthreads();
  for (int i = tid; i < 27 * 32; i += 256) {
    const int tt = i / 32, c = i % 32;
    const float s = red[(0 * 32 + tt) * 32 + c] + red[(1 * 32 + tt) * 32 + c] + red[(2 * 32 + tt) * 32 + c] +
                    red[(3 * 32 + tt) * 32 + c];
    if (a.acc)
      fx_add(a.acc, c * 27 + tt, 27 * 32, blockIdx.x, (double)s);
    else
      atomicAdd(a.dw + c * 27 + tt, s);
  }
  if (a.acc && last_workgroup(a.counter, gridDim.x))
    for (int i = tid; i < 27 * 32; i += 256) a.dw[i] = (float)fx_take(a.acc, i, 27 * 32);
}

// ------------------------------------------------------------------------------------------------ head
constexpr int HD_MAXK = 32;  // classes: <= 8 run the register-resident kernels, 9..32 the row-per-thread ones
constexpr int HD_UNR = 4;  // rows in flight per thread in the coalesced head kernels

struct HeadArgs {
  const f16* x;      // [N][V][ldx], C channels
  const float* w;    // [K][C] fp32
  const float* b;    // [K]
  f16* logits;       // [N][K][V] (NCDHW)
  const f16* dl;     // [N][K][V] gradient of logits
  f16* dx;           // [N][V][lddx]
  float* dw;         // [K][C] (zeroed by launcher)
  float* db;         // [K]
  int N, C, K, ldx, lddx;
  long V;
  int Ktot, k0;      // weight-gradient launches cover classes k0 .. k0 + K - 1 of Ktot
  FxAcc* acc;        // deterministic weight gradient: K * C + K fixed-point accumulators (common.hpp) + launch counter
  unsigned* counter;
  // consumer-side InstanceNorm + LeakyReLU (nnz_seg_head_forward_innorm / nnz_seg_head_wgrad_innorm): x is the RAW conv
  // output of the decoder stage's last block; every 16-byte piece is normalised right after its load with the block's table
  // in_tab[N][C][4] = {mean, rstd, scale, shift} - y = lrelu(x * scale + shift), rounded once to fp16 like the apply pass of
  // norm_act.hip - so the activated tensor is never written.  The forward launches then take one sample per grid row.
  const float* in_tab;
  float in_slope;
};

// the thread's 8 channels of sample n: scale / shift registers
__device__ __forceinline__ void head_tab8(const HeadArgs& a, long n, int c0, float (&sc)[8], float (&sh)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x2 t = *reinterpret_cast<const f32x2*>(a.in_tab + ((size_t)n * a.C + c0 + i) * 4 + 2);
    sc[i] = t[0];
    sh[i] = t[1];
  }
}
__device__ __forceinline__ f16x8 head_norm8(f16x8 h, const float (&sc)[8], const float (&sh)[8], float slope) {
  return __builtin_bit_cast(f16x8, norm_lrelu8(__builtin_bit_cast(u32x4, h), sc, sh, slope_pair(slope)));
}

template <int MAXK>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
  extern __shared__ float wsm[];  // [K][C]
  for (int i = threadIdx.x; i < a.K * a.C; i += 256) wsm[i] = (float)(f16)a.w[i];
  __syncthreads();
  const long total = (long)a.N * a.V;
  for (long row = blockIdx.x * 256L + threadIdx.x; row < total; row += (long)gridDim.x * 256) {
    const long n = row / a.V, v = row % a.V;
    float acc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[k] = (k < a.K && a.b) ? a.b[k] : 0.f;
    const f16* xp = a.x + row * a.ldx;
    for (int c8 = 0; c8 < a.C; c8 += 8) {
      f16x8 h = *reinterpret_cast<const f16x8*>(xp + c8);
      if (a.in_tab) {   // (this kernel serves the C = 320 levels: a few hundred rows)
        float sc[8], sh[8];
        head_tab8(a, n, c8, sc, sh);
        h = head_norm8(h, sc, sh, a.in_slope);
      }
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < a.K) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[k] += (float)h[i] * wsm[k * a.C + c8 + i];
        }
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < a.K) a.logits[(n * a.K + k) * a.V + v] = (f16)acc[k];
  }
}

template <int MAXK>
__global__ __launch_bounds__(256) void head_dgrad_kernel(HeadArgs a, int accumulate) {
  extern __shared__ float wsm[];
  for (int i = threadIdx.x; i < a.K * a.C; i += 256) wsm[i] = (float)(f16)a.w[i];
  __syncthreads();
  const long total = (long)a.N * a.V;
  for (long row = blockIdx.x * 256L + threadIdx.x; row < total; row += (long)gridDim.x * 256) {
    const long n = row / a.V, v = row % a.V;
    float g[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) g[k] = k < a.K ? (float)a.dl[(n * a.K + k) * a.V + v] : 0.f;
    f16* dp = a.dx + row * a.lddx;
    for (int c8 = 0; c8 < a.C; c8 += 8) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < a.K) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += g[k] * wsm[k * a.C + c8 + i];
        }
      f16x8 h;
      if (accumulate) {
        const f16x8 old = *reinterpret_cast<const f16x8*>(dp + c8);
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (f16)(o[i] + (float)old[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (f16)o[i];
      }
      *reinterpret_cast<f16x8*>(dp + c8) = h;
    }
  }
}

// ---- coalesced variants for power-of-two channel-group counts (C = 32 ... 256) ---------------------------------------
// Thread = (row r, channel group cg of 8 fp16 = 16 bytes): a wave's load instruction covers 1 KiB of consecutive
// memory, HD_UNR rows are in flight per thread (the loop is latency-bound otherwise), the K x 8 weights of the
// thread's channels sit in registers.  The row-per-thread kernels above stay for C = 320 (8^3 voxels: negligible).
template <int K>
__global__ __launch_bounds__(256) void head_fwd_cg_kernel(HeadArgs a, int cgs) {
  const int tid = threadIdx.x;
  const int cg = tid & (cgs - 1);
  const int rows = 256 / cgs;
  const int r = tid / cgs;
  float w[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) w[k][i] = (float)(f16)a.w[k * a.C + cg * 8 + i];
  float bias[K];
#pragma unroll
  for (int k = 0; k < K; ++k) bias[k] = a.b ? a.b[k] : 0.f;
  // consumer-side norm: grid row = sample, the thread's {scale, shift} pairs stay in registers
  float sc[8], sh[8];
  if (a.in_tab) head_tab8(a, blockIdx.y, cg * 8, sc, sh);
  const long base = a.in_tab ? (long)blockIdx.y * a.V : 0;
  const long total = a.in_tab ? base + a.V : (long)a.N * a.V;
  for (long row0 = base + (long)blockIdx.x * rows * HD_UNR; row0 < total; row0 += (long)gridDim.x * rows * HD_UNR) {
    f16x8 h[HD_UNR];
#pragma unroll
    for (int u = 0; u < HD_UNR; ++u) {
      const long row = row0 + u * rows + r;
      if (row < total) h[u] = *reinterpret_cast<const f16x8*>(a.x + row * a.ldx + cg * 8);
    }
    if (a.in_tab) {
#pragma unroll
      for (int u = 0; u < HD_UNR; ++u) h[u] = head_norm8(h[u], sc, sh, a.in_slope);
    }
#pragma unroll
    for (int u = 0; u < HD_UNR; ++u) {
      const long row = row0 + u * rows + r;
      float acc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        acc[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[k] += (float)h[u][i] * w[k][i];
      }
      // fold the channel groups of the row (consecutive lanes)
      for (int o = 1; o < cgs; o <<= 1)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
      if (cg == 0 && row < total) {
        const long n = row / a.V, v = row - n * a.V;
#pragma unroll
        for (int k = 0; k < K; ++k) a.logits[(n * K + k) * a.V + v] = (f16)(acc[k] + bias[k]);
      }
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void head_dgrad_cg_kernel(HeadArgs a, int cgs, int accumulate) {
  const int tid = threadIdx.x;
  const int cg = tid & (cgs - 1);
  const int rows = 256 / cgs;
  const int r = tid / cgs;
  float w[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) w[k][i] = (float)(f16)a.w[k * a.C + cg * 8 + i];
  const long total = (long)a.N * a.V;
  for (long row0 = (long)blockIdx.x * rows * HD_UNR; row0 < total; row0 += (long)gridDim.x * rows * HD_UNR) {
    float g[HD_UNR][K];
    f16x8 old[HD_UNR];
#pragma unroll
    for (int u = 0; u < HD_UNR; ++u) {
      const long row = row0 + u * rows + r;
      if (row < total) {
        const long n = row / a.V, v = row - n * a.V;
#pragma unroll
        for (int k = 0; k < K; ++k) g[u][k] = (float)a.dl[(n * K + k) * a.V + v];
        if (accumulate) old[u] = *reinterpret_cast<const f16x8*>(a.dx + row * a.lddx + cg * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < HD_UNR; ++u) {
      const long row = row0 + u * rows + r;
      if (row >= total) continue;
      f16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = 0.f;  // same summation order over k as the row-per-thread kernel, accumulate added last
#pragma unroll
        for (int k = 0; k < K; ++k) t += g[u][k] * w[k][i];
        o[i] = (f16)(accumulate ? t + (float)old[u][i] : t);
      }
      *reinterpret_cast<f16x8*>(a.dx + row * a.lddx + cg * 8) = o;
    }
  }
}

// dW[k][c] = sum_{n,v} dl[n][k][v] * x[n][v][c];  db[k] = sum dl.   thread = (row r, channel group cg)
template <int K>
__global__ __launch_bounds__(256) void head_wgrad_kernel(HeadArgs a, int vpb) {
  extern __shared__ float lred[];  // [rows][K*C + K]
  const int CG = a.C >> 3;
  const int rows = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, r = tid / CG;
  const int n = blockIdx.y;
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > a.V) v1 = a.V;
  float acc[K][8];
  float accb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    accb[k] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[k][i] = 0.f;
  }
  if (r < rows) {
    const f16* xp = a.x + (long)n * a.V * a.ldx + cg * 8;
    const f16* gp = a.dl + ((long)n * a.Ktot + a.k0) * a.V;
    float sc[8], sh[8];
    if (a.in_tab) head_tab8(a, n, cg * 8, sc, sh);
    for (long v = v0 + r; v < v1; v += (long)rows * HD_UNR) {
      // HD_UNR rows in flight per thread (one 16-byte load + K 2-byte loads each): the loop was latency-bound
      f16x8 h[HD_UNR];
      float gk[HD_UNR][K];
#pragma unroll
      for (int u = 0; u < HD_UNR; ++u) {
        const long vv = v + (long)u * rows;
        if (vv < v1) {
          h[u] = *reinterpret_cast<const f16x8*>(xp + vv * a.ldx);
#pragma unroll
          for (int k = 0; k < K; ++k) gk[u][k] = (float)gp[(long)k * a.V + vv];
        }
      }
      if (a.in_tab) {
#pragma unroll
        for (int u = 0; u < HD_UNR; ++u) h[u] = head_norm8(h[u], sc, sh, a.in_slope);
      }
#pragma unroll
      for (int u = 0; u < HD_UNR; ++u) {
        if (v + (long)u * rows >= v1) continue;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          accb[k] += gk[u][k];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[k][i] += gk[u][k] * (float)h[u][i];
        }
      }
    }
  }
  // block reduction through an LDS slab [rows][K*C + K] and one column sum per thread (LDS float atomics with 64
  // threads per address serialised: that tail was longer than the streaming loop)
  const int KC = K * a.C + K;
  if (r < rows) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) lred[r * KC + k * a.C + cg * 8 + i] = acc[k][i];
      if (cg == 0) lred[r * KC + K * a.C + k] = accb[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < KC; i += 256) {
    float t = 0.f;
    for (int rr = 0; rr < rows; ++rr) t += lred[rr * KC + i];
    if (a.acc)
      fx_add(a.acc, i, KC, blockIdx.x, (double)t);
    else if (i < K * a.C)
      atomicAdd(a.dw + (long)a.k0 * a.C + i, t);
    else
      atomicAdd(a.db + a.k0 + (i - K * a.C), t);
  }
  if (a.acc && last_workgroup(a.counter, gridDim.x * gridDim.y))
    for (int i = tid; i < KC; i += 256) {
      const float t = (float)fx_take(a.acc, i, KC);
      if (i < K * a.C)
        a.dw[(long)a.k0 * a.C + i] = t;
      else
        a.db[a.k0 + (i - K * a.C)] = t;
    }
}

}  // namespace nnz

extern "C" int nnz_stem_conv_forward(const float* x, const float* w, const float* bias, void* y, int N, int D, int H,
                                     int W, int Cout, int ldy, void* stream) {
  using namespace nnz;
  if (!x || !w || !y || Cout != ST_CO || ldy % 8) return NNZ_EINVAL;
  StemArgs a = {};
  a.x = x; a.w = w; a.b = bias; a.y = (f16*)y;
  a.N = N; a.D = D; a.H = H; a.W = W; a.ldy = ldy;
  a.tiles[0] = (D + ST_TD - 1) / ST_TD;
  a.tiles[1] = (H + ST_TH - 1) / ST_TH;
  a.tiles[2] = (W + ST_TW - 1) / ST_TW;
  NNZ_LAUNCH(stem_fwd_mfma_kernel, dim3(a.tiles[0] * a.tiles[1] * a.tiles[2], N), dim3(256), 0,
                     (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_stem_conv_wgrad_det(const float* x, const void* dy, float* dw, int N, int D, int H, int W, int Cout,
                                       int lddy, void* acc, void* counter, void* stream);
extern "C" int nnz_stem_conv_wgrad(const float* x, const void* dy, float* dw, int N, int D, int H, int W, int Cout,
                                   int lddy, void* stream) {
  return nnz_stem_conv_wgrad_det(x, dy, dw, N, D, H, W, Cout, lddy, nullptr, nullptr, stream);
}

// acc / counter (both or neither): >= 864 zeroed fixed-point records (nnz_fxacc_bytes() each) + one zeroed 32-bit word,
// left zero; with them the gradient is bit-identical run to run (no float atomics) and dw needs no zero fill
extern "C" int nnz_stem_conv_wgrad_det(const float* x, const void* dy, float* dw, int N, int D, int H, int W, int Cout,
                                       int lddy, void* acc, void* counter, void* stream) {
  using namespace nnz;
  if (!x || !dy || !dw || Cout != ST_CO || lddy % 8 || (!acc != !counter)) return NNZ_EINVAL;
  StemArgs a = {};
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
  a.x = x; a.dy = (const f16*)dy; a.dw = dw;
  a.N = N; a.D = D; a.H = H; a.W = W; a.lddy = lddy;
  a.tiles[0] = (D + ST_TD - 1) / ST_TD;
  a.tiles[1] = (H + ST_TH - 1) / ST_TH;
  a.tiles[2] = (W + ST_TW - 1) / ST_TW;
  const int ntiles = N * a.tiles[0] * a.tiles[1] * a.tiles[2];
  if (!acc) {
    hipError_t e = nnz::zero_async(dw, sizeof(float) * ST_CO * 27, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  const int grid = ntiles < 512 ? ntiles : 512;  // persistent: 2 workgroups per CU, one atomic per element each
  NNZ_LAUNCH(stem_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, ntiles);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

namespace nnz {
static inline bool head_cg_ok(int C) {
  const int cgs = C >> 3;
  return C % 8 == 0 && cgs >= 1 && cgs <= 32 && (cgs & (cgs - 1)) == 0;
}
template <int K>
static void launch_head_fwd_cg(const HeadArgs& a, hipStream_t s) {
  const int cgs = a.C >> 3, rows = 256 / cgs;
  const long nrows = a.in_tab ? a.V : (long)a.N * a.V;   // consumer-side norm: one sample per grid row
  long blocks = (nrows + rows * HD_UNR - 1) / (rows * HD_UNR);
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(head_fwd_cg_kernel<K>, dim3((int)blocks, a.in_tab ? a.N : 1), dim3(256), 0, s, a, cgs);
}
template <int K>
static void launch_head_dgrad_cg(const HeadArgs& a, int accumulate, hipStream_t s) {
  const int cgs = a.C >> 3, rows = 256 / cgs;
  long blocks = ((long)a.N * a.V + rows * HD_UNR - 1) / (rows * HD_UNR);
  if (blocks > 8192) blocks = 8192;
  NNZ_LAUNCH(head_dgrad_cg_kernel<K>, dim3((int)blocks), dim3(256), 0, s, a, cgs, accumulate);
}
#define NNZ_HEAD_K_SWITCH(K, CALL) \
  switch (K) {                      \
    case 1: CALL(1); break;         \
    case 2: CALL(2); break;         \
    case 3: CALL(3); break;         \
    case 4: CALL(4); break;         \
    case 5: CALL(5); break;         \
    case 6: CALL(6); break;         \
    case 7: CALL(7); break;         \
    default: CALL(8); break;        \
  }
}  // namespace nnz

extern "C" int nnz_seg_head_forward_innorm(const void* x_raw, const float* in_tab, float in_slope, const float* w,
                                           const float* bias, void* logits, int N, long V, int C, int K, int ldx,
                                           void* stream);
extern "C" int nnz_seg_head_forward(const void* x, const float* w, const float* bias, void* logits, int N, long V,
                                    int C, int K, int ldx, void* stream) {
  return nnz_seg_head_forward_innorm(x, nullptr, 0.f, w, bias, logits, N, V, C, K, ldx, stream);
}

// 1x1 segmentation head on the RAW conv output of the decoder stage's last block: InstanceNorm + LeakyReLU are applied to
// every piece as it is loaded (HeadArgs::in_tab); in_tab == NULL is nnz_seg_head_forward.
extern "C" int nnz_seg_head_forward_innorm(const void* x, const float* in_tab, float in_slope, const float* w,
                                           const float* bias, void* logits, int N, long V, int C, int K, int ldx,
                                           void* stream) {
  using namespace nnz;
  if (!x || !w || !logits || K < 1 || K > HD_MAXK || C % 8 || ldx % 8) return NNZ_EINVAL;
  HeadArgs a = {};
  a.in_tab = in_tab; a.in_slope = in_slope;
  a.x = (const f16*)x; a.w = w; a.b = bias; a.logits = (f16*)logits;
  a.N = N; a.V = V; a.C = C; a.K = K; a.ldx = ldx;
  if (head_cg_ok(C) && K <= 8) {
#define NNZ_CALL(KK) launch_head_fwd_cg<KK>(a, (hipStream_t)stream)
    NNZ_HEAD_K_SWITCH(K, NNZ_CALL)
#undef NNZ_CALL
    NNZ_LAUNCH_CHECK();
    return NNZ_OK;
  }
  long blocks = (N * V + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (K <= 8)
    NNZ_LAUNCH(head_fwd_kernel<8>, dim3((int)blocks), dim3(256), sizeof(float) * K * C, (hipStream_t)stream, a);
  else
    NNZ_LAUNCH(head_fwd_kernel<HD_MAXK>, dim3((int)blocks), dim3(256), sizeof(float) * K * C,
                       (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_seg_head_dgrad(const void* dlogits, const float* w, void* dx, int N, long V, int C, int K, int lddx,
                                  int accumulate, void* stream) {
  using namespace nnz;
  if (!dlogits || !w || !dx || K < 1 || K > HD_MAXK || C % 8 || lddx % 8) return NNZ_EINVAL;
  HeadArgs a = {};
  a.dl = (const f16*)dlogits; a.w = w; a.dx = (f16*)dx;
  a.N = N; a.V = V; a.C = C; a.K = K; a.lddx = lddx;
  if (head_cg_ok(C) && K <= 8) {
#define NNZ_CALL(KK) launch_head_dgrad_cg<KK>(a, accumulate, (hipStream_t)stream)
    NNZ_HEAD_K_SWITCH(K, NNZ_CALL)
#undef NNZ_CALL
    NNZ_LAUNCH_CHECK();
    return NNZ_OK;
  }
  long blocks = (N * V + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (K <= 8)
    NNZ_LAUNCH(head_dgrad_kernel<8>, dim3((int)blocks), dim3(256), sizeof(float) * K * C, (hipStream_t)stream,
                       a, accumulate);
  else
    NNZ_LAUNCH(head_dgrad_kernel<HD_MAXK>, dim3((int)blocks), dim3(256), sizeof(float) * K * C,
                       (hipStream_t)stream, a, accumulate);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_seg_head_wgrad_det(const void* x, const void* dlogits, float* dw, float* db, int N, long V, int C,
                                      int K, int ldx, void* acc, void* counter, void* stream);
extern "C" int nnz_seg_head_wgrad_innorm(const void* x_raw, const float* in_tab, float in_slope, const void* dlogits,
                                         float* dw, float* db, int N, long V, int C, int K, int ldx, void* acc,
                                         void* counter, void* stream);
extern "C" int nnz_seg_head_wgrad(const void* x, const void* dlogits, float* dw, float* db, int N, long V, int C, int K,
                                  int ldx, void* stream) {
  return nnz_seg_head_wgrad_det(x, dlogits, dw, db, N, V, C, K, ldx, nullptr, nullptr, stream);
}

// acc / counter (both or neither): >= 8 * (C + 1) zeroed fixed-point records + one zeroed 32-bit word, left zero; with
// them dw / db are bit-identical run to run (no float atomics) and need no zero fill
extern "C" int nnz_seg_head_wgrad_det(const void* x, const void* dlogits, float* dw, float* db, int N, long V, int C,
                                      int K, int ldx, void* acc, void* counter, void* stream) {
  return nnz_seg_head_wgrad_innorm(x, nullptr, 0.f, dlogits, dw, db, N, V, C, K, ldx, acc, counter, stream);
}

// ... with x the RAW conv output of the stage's last block (HeadArgs::in_tab); in_tab == NULL is nnz_seg_head_wgrad_det
extern "C" int nnz_seg_head_wgrad_innorm(const void* x, const float* in_tab, float in_slope, const void* dlogits, float* dw,
                                         float* db, int N, long V, int C, int K, int ldx, void* acc, void* counter,
                                         void* stream) {
  using namespace nnz;
  if (!x || !dlogits || !dw || !db || K < 1 || K > HD_MAXK || C % 8 || C > 640 || ldx % 8 || (!acc != !counter))
    return NNZ_EINVAL;
  HeadArgs a = {};
  a.in_tab = in_tab; a.in_slope = in_slope;
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
  a.x = (const f16*)x; a.dl = (const f16*)dlogits; a.dw = dw; a.db = db;
  a.N = N; a.V = V; a.C = C; a.K = K; a.ldx = ldx;
  hipStream_t s = (hipStream_t)stream;
  if (!acc) {
    hipError_t e = nnz::zero_async(dw, sizeof(float) * K * C, s);
    if (e != hipSuccess) return (int)e;
    e = nnz::zero_async(db, sizeof(float) * K, s);
    if (e != hipSuccess) return (int)e;
  }
  long vpb = (V * N + 1023) / 1024;
  if (vpb < 256) vpb = 256;
  if (vpb > V) vpb = V;
  const int gx = (int)((V + vpb - 1) / vpb);
  const int rows = 256 / (C >> 3) < 1 ? 1 : 256 / (C >> 3);
  // classes in groups of <= 8 (the accumulators of a group live in registers)
  int kmax = (int)((64 * 1024 / sizeof(float)) / ((size_t)rows * (C + 1)));  // classes per launch the LDS slab holds
  if (kmax > 8) kmax = 8;
  if (kmax < 1) return NNZ_EINVAL;
  for (int k0 = 0; k0 < K; k0 += kmax) {
    const int kg = K - k0 < kmax ? K - k0 : kmax;
    a.K = kg; a.Ktot = K; a.k0 = k0;
    const size_t lds = sizeof(float) * rows * (kg * C + kg);
    if (lds > 64 * 1024) return NNZ_EINVAL;  // K*C beyond the slab: not a segmentation head
#define NNZ_CALL(KK) NNZ_LAUNCH(head_wgrad_kernel<KK>, dim3(gx, N), dim3(256), lds, s, a, (int)vpb)
    NNZ_HEAD_K_SWITCH(kg, NNZ_CALL)
#undef NNZ_CALL
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
