// x_proj of the SS2D / cross-scan block on channel-major fp32 activations (gfx950).
// Reference: the two einsums of SS2D.forward_core, /root/reference/nnunetv2/nets/m2net.py:179-184
//   x_dbl = einsum("b k d l, k c d -> b k c l", xs, x_proj_weight)   (dt, B, C rows of each direction)
// In the cross-scan formulation (nnuzoo_amd/ss2d_scan.py) directions k = s + 2j share source s (row- / column-major
// tokens), so the projection is  P[s][b][c][l] = sum_d Wst[s][c][d] x2[s][b][d][l],  c < C2 = 2 (R + 2N) <= 80 rows,
// Di = 32..256 input channels, L up to 262 144 tokens.  As GEMMs these are (C2 x Di) x (Di x L): a library call with
// M = 66..80 runs a 16x16 macro tile at ~1 ms (10 calls per M2Net step at 512^2) where the data is 0.1-0.4 GB
// (25-70 us at HBM speed), and each call costs ~28 us of host dispatch (660 mm / bmm calls per step).
//
// Layout-native design: the token axis is contiguous, so LANES = TOKENS and every global access is a coalesced 256-byte
// run per wave; the small weight matrix is staged (32 input channels at a time) into LDS and read as wave-uniform
// broadcasts; a thread keeps all C2 outputs of its token in registers.  fp32 FMAs throughout (the scan is fp32 in the
// reference: `.float()` at m2net.py:185-191).
//   xproj_fwd  : P   = W x2
//   xproj_bwd_x: dx2 = W^T dP + du[k = s] + du[k = s + 2]            (the scans' own input gradients folded in)
//   xproj_bwd_w: dW  = sum_{b, l} dP x2^T                             (register tiles, token ranges over workgroups)
#include "common.hpp"

namespace nnz {

constexpr int XP_CMAX = 80;  // 2 (R + 32), R <= 8
constexpr int XP_DC = 32;    // input channels per staged weight chunk

struct XpArgs {
  const float* x2;   // [2][B][Di][L]
  const float* W;    // [2][C2][Di]
  float* P;          // [2][B][C2][L]
  const float* dP;   // [2][B][C2][L]
  const float* du;   // [B][4][Di][L]
  float* dx2;        // [2][B][Di][L]
  float* dW;         // [2][C2][Di] (atomic, pre-zeroed)
  int B, Di, C2;
  long L;
  long tokens_per_wg;
  // Cp > 0: W / dW are the module's own x_proj_weight layout [K = 4][Cp][Di] (direction k = s + 2 j holds rows
  // [j Cp, (j + 1) Cp) of source s's stacked matrix, C2 = 2 Cp) - no stacked copy of the weight per call, no un-stacking
  // copy of its gradient; Cp == 0: W / dW are [2][C2][Di]
  int Cp;
  float* part;       // null, or [workgroups along x][2 * C2 * Di]: two-stage (deterministic) weight gradient
};

// row c2 of source s's projection matrix
__device__ __forceinline__ long xp_row(const XpArgs& a, int s, int c2) {
  if (a.Cp > 0) return ((long)(s + 2 * (c2 / a.Cp)) * a.Cp + c2 % a.Cp) * a.Di;
  return ((long)s * a.C2 + c2) * a.Di;
}

// CPT = outputs per thread: a thread's work is CPT * Di FMAs whatever L is, so the deep levels (Di = 256 ... 512, L = 256 ...
// 4 096 tokens: 4-64 workgroups) ran ~55 us on pure loop latency (r03_m2net_kernel_histogram: median launch 16 workgroups,
// 55 us); there the outputs are split into 80 / CPT groups over blockIdx.z - more, shorter threads (x2 is re-read per group
// from L2).
template <int CPT>
__global__ __launch_bounds__(256) void xproj_fwd_kernel(XpArgs a) {
  __shared__ __attribute__((aligned(16))) float sW[CPT * XP_DC];  // [c][32 d] of the current chunk
  const int sb = blockIdx.y;                 // s * B + b
  const int s = sb / a.B;
  const int cbase = blockIdx.z * CPT;
  if (cbase >= a.C2) return;
  const long l = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = l < a.L;
  const float* x = a.x2 + (long)sb * a.Di * a.L + (ok ? l : 0);
  float acc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
  for (int d0 = 0; d0 < a.Di; d0 += XP_DC) {
    __syncthreads();
    for (int e = threadIdx.x; e < CPT * XP_DC; e += 256) {
      const int c = e / XP_DC, d = e % XP_DC;
      sW[e] = cbase + c < a.C2 ? a.W[xp_row(a, s, cbase + c) + d0 + d] : 0.f;
    }
    float xv[XP_DC];
#pragma unroll
    for (int d = 0; d < XP_DC; ++d) xv[d] = ok ? x[(long)(d0 + d) * a.L] : 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
#pragma unroll
      for (int d4 = 0; d4 < XP_DC; d4 += 4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sW + c * XP_DC + d4);  // wave-uniform address: broadcast
        acc[c] += w[0] * xv[d4] + w[1] * xv[d4 + 1] + w[2] * xv[d4 + 2] + w[3] * xv[d4 + 3];
      }
    }
  }
  if (ok) {
    float* P = a.P + (long)sb * a.C2 * a.L + l;
#pragma unroll
    for (int c = 0; c < CPT; ++c)
      if (cbase + c < a.C2) P[(long)(cbase + c) * a.L] = acc[c];
  }
}

__global__ __launch_bounds__(256) void xproj_bwd_x_kernel(XpArgs a) {
  __shared__ __attribute__((aligned(16))) float sWt[XP_DC * XP_CMAX];  // [32 d][c] of the current chunk (transposed)
  const int sb = blockIdx.y;
  const int s = sb / a.B, b = sb % a.B;
  const long l = (long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = l < a.L;
  const float* dP = a.dP + (long)sb * a.C2 * a.L + (ok ? l : 0);
  float g[XP_CMAX];
#pragma unroll
  for (int c = 0; c < XP_CMAX; ++c) g[c] = (ok && c < a.C2) ? dP[(long)c * a.L] : 0.f;
  // du of the two directions of this source: k = s and k = s + 2 in [B][4][Di][L]
  const float* du0 = a.du + (((long)b * 4 + s) * a.Di) * a.L + (ok ? l : 0);
  const float* du1 = a.du + (((long)b * 4 + s + 2) * a.Di) * a.L + (ok ? l : 0);
  float* dx = a.dx2 + (long)sb * a.Di * a.L + l;
  // blockIdx.z = a slice of the Di output channels (the deep levels: few tokens, many channels - see xproj_fwd_kernel)
  const int dper = a.Di / gridDim.z;
  for (int d0 = blockIdx.z * dper; d0 < (blockIdx.z + 1) * dper; d0 += XP_DC) {
    __syncthreads();
    for (int e = threadIdx.x; e < XP_DC * XP_CMAX; e += 256) {
      const int c = e / XP_DC, d = e % XP_DC;     // read W[c][d0 + d] with d fastest (coalesced), store transposed
      sWt[d * XP_CMAX + c] = c < a.C2 ? a.W[xp_row(a, s, c) + d0 + d] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int d = 0; d < XP_DC; ++d) {
      float v = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < XP_CMAX; c4 += 4) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sWt + d * XP_CMAX + c4);
        v += w[0] * g[c4] + w[1] * g[c4 + 1] + w[2] * g[c4 + 2] + w[3] * g[c4 + 3];
      }
      if (ok) {
        const long o = (long)(d0 + d) * a.L;
        dx[o] = v + du0[o] + du1[o];
      }
    }
  }
}

// ---- round 5: forward and backward-x on the fp32 matrix cores ------------------------------------------------------------------
// The FMA kernels above read their weights as wave-uniform LDS broadcasts: one ds_read_b128 per four FMAs, and with four SIMDs
// issuing them the LDS pipe, not the FMAs and not HBM, sets the pace (forward 182 us for 32 channels @ 512^2, batch 2, where the
// bytes take 75; backward-x 268 us for 118).  v_mfma_f32_32x32x2_f32 (exact fp32, the FMA rate) takes the weights as its A
// operand - ONE register per (row tile, step) and lane, fetched once per 64 tokens - and tokens sit on the lanes:
//   forward     P[c][l]  = sum_d W[c][d] x[d][l]      rows c (3 tiles of 32 cover C2 <= 80), contraction d, 16 steps per 32 channels
//   backward-x  dx[d][l] = sum_c W[c][d] dP[c][l]     rows d (one tile per 32 channels), contraction c (C2 / 2 steps)
// A wave owns 64 consecutive tokens as TWO column tiles: lane (l31, hh) holds tokens 2 l31 and 2 l31 + 1 of contraction row
// 2 step + hh, i.e. one 8-byte load per step (256 contiguous bytes per half-wave), and stores token pairs the same way.
// Needs an even L (8-byte alignment of every row); odd L keeps the FMA kernels (NNZ_XPROJ_MFMA=0 forces them).
__device__ __forceinline__ f32x16 xp_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int xp_crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// NRT row tiles per workgroup (3: all of C2 <= 96 in one go; 1: blockIdx.z picks the tile - few-token levels)
template <int NRT>
__global__ __launch_bounds__(256) void xproj_fwd_mfma_kernel(XpArgs a) {
  constexpr int WP = 97;                                 // pitch of the image: the transposing stores hit 32 different banks
  __shared__ float sW[XP_DC * WP];                       // [d of the chunk][c], zero beyond C2
  const int sb = blockIdx.y, s = sb / a.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int rt0 = NRT == 3 ? 0 : blockIdx.z;
  if (rt0 * 32 >= a.C2) return;
  const long tok = (long)blockIdx.x * 256 + wave * 64 + 2 * l31;   // this lane's token pair
  const bool ok = tok < a.L;                                        // L is even: a pair is inside or outside as a whole
  const float* x = a.x2 + (long)sb * a.Di * a.L + (ok ? tok : 0);
  f32x16 acc[NRT][2];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
  for (int d0 = 0; d0 < a.Di; d0 += XP_DC) {
    f32x2 xv[XP_DC / 2];
#pragma unroll
    for (int st = 0; st < XP_DC / 2; ++st)
      xv[st] = ok ? *reinterpret_cast<const f32x2*>(x + (long)(d0 + 2 * st + hh) * a.L) : f32x2{0.f, 0.f};
    __syncthreads();
    for (int e = tid; e < XP_DC * 96; e += 256) {        // W[c][d0 + d], d fastest in memory; image [d][c]
      const int c = e / XP_DC, d = e % XP_DC;
      sW[d * WP + c] = c < a.C2 ? a.W[xp_row(a, s, c) + d0 + d] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < XP_DC / 2; ++st) {
#pragma unroll
      for (int i = 0; i < NRT; ++i) {
        const float w = sW[(2 * st + hh) * WP + (rt0 + i) * 32 + l31];      // 32 consecutive words per half-wave
        acc[i][0] = xp_mfma(w, xv[st][0], acc[i][0]);
        acc[i][1] = xp_mfma(w, xv[st][1], acc[i][1]);
      }
    }
  }
  if (!ok) return;
  float* P = a.P + (long)sb * a.C2 * a.L + tok;
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (rt0 + i) * 32 + xp_crow(r, hh);
      if (c < a.C2) *reinterpret_cast<f32x2*>(P + (long)c * a.L) = f32x2{acc[i][0][r], acc[i][1][r]};
    }
}

// blockIdx.z = a slice of the Di output channels (multiples of 32)
__global__ __launch_bounds__(256) void xproj_bwd_x_mfma_kernel(XpArgs a) {
  __shared__ float sW[XP_CMAX * XP_DC];                  // [c][d of the chunk]: W's own layout, rows beyond C2 zero
  const int sb = blockIdx.y, s = sb / a.B, b = sb % a.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long tok = (long)blockIdx.x * 256 + wave * 64 + 2 * l31;
  const bool ok = tok < a.L;
  const float* dP = a.dP + (long)sb * a.C2 * a.L + (ok ? tok : 0);
  f32x2 g[XP_CMAX / 2];                                   // dP[c = 2 step + hh][token pair]
#pragma unroll
  for (int st = 0; st < XP_CMAX / 2; ++st) {
    const int c = 2 * st + hh;
    g[st] = (ok && c < a.C2) ? *reinterpret_cast<const f32x2*>(dP + (long)c * a.L) : f32x2{0.f, 0.f};
  }
  const float* du0 = a.du + (((long)b * 4 + s) * a.Di) * a.L + (ok ? tok : 0);
  const float* du1 = a.du + (((long)b * 4 + s + 2) * a.Di) * a.L + (ok ? tok : 0);
  float* dx = a.dx2 + (long)sb * a.Di * a.L + tok;
  const int dper = a.Di / gridDim.z;
  for (int d0 = blockIdx.z * dper; d0 < (blockIdx.z + 1) * dper; d0 += XP_DC) {
    __syncthreads();
    for (int e = tid; e < XP_CMAX * XP_DC; e += 256) {
      const int c = e / XP_DC, d = e % XP_DC;
      sW[e] = c < a.C2 ? a.W[xp_row(a, s, c) + d0 + d] : 0.f;
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int st = 0; st < XP_CMAX / 2; ++st) {
      const float w = sW[(2 * st + hh) * XP_DC + l31];    // A[row d = l31][k = c]: 32 consecutive words per half-wave
      acc[0] = xp_mfma(w, g[st][0], acc[0]);
      acc[1] = xp_mfma(w, g[st][1], acc[1]);
    }
    if (ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long o = (long)(d0 + xp_crow(r, hh)) * a.L;
        const f32x2 u0 = *reinterpret_cast<const f32x2*>(du0 + o), u1 = *reinterpret_cast<const f32x2*>(du1 + o);
        *reinterpret_cast<f32x2*>(dx + o) = f32x2{acc[0][r] + u0[0] + u1[0], acc[1][r] + u0[1] + u1[1]};
      }
    }
  }
}

// dW[s][c][d] = sum over (b, l) of dP[c][l] x2[d][l].  Workgroup = one source s and a token range; thread = (token slice,
// 8 x 8 block of dW); rows are staged as [row][64 tokens] tiles and read 4 tokens (16 bytes) at a time.
constexpr int XPW_TOK = 64;

__device__ __forceinline__ void xproj_bwd_w_body(const XpArgs& a, const unsigned bx, const int s) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int C2p = (a.C2 + 7) & ~7;
  const int nbk = (C2p >> 3) * (a.Di >> 3);
  const int TSL = 256 / nbk;  // >= 1 (launcher checks nbk <= 256)
  float* sP = smem_f;                         // [C2p][64]
  float* sX = sP + C2p * XPW_TOK;             // [Di][64]
  float* sred = sX + a.Di * XPW_TOK;          // [TSL][C2p * Di]
  const int tid = threadIdx.x;
  const int blk = tid % nbk, ts = tid / nbk;
  const bool active = ts < TSL;
  const int c0 = (blk / (a.Di >> 3)) * 8, d0 = (blk % (a.Di >> 3)) * 8;
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  const long T = (long)a.B * a.L;             // tokens of this source, (b, l) flattened: rows are [b][row][l]
  const long t_begin = (long)bx * a.tokens_per_wg;
  long t_end = t_begin + a.tokens_per_wg;
  if (t_end > T) t_end = T;
  for (long tb = t_begin; tb < t_end; tb += XPW_TOK) {
    // a 64-token round never straddles two samples: tokens_per_wg and L are multiples of 64 (launcher)
    const int b = (int)(tb / a.L);
    const long l0 = tb - (long)b * a.L;
    const float* dPb = a.dP + ((long)(s * a.B + b) * a.C2) * a.L + l0;
    const float* xb = a.x2 + ((long)(s * a.B + b) * a.Di) * a.L + l0;
    __syncthreads();
    for (int e = tid; e < C2p * (XPW_TOK / 4); e += 256) {
      const int c = e / (XPW_TOK / 4), q = e % (XPW_TOK / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < a.C2) v = *reinterpret_cast<const f32x4*>(dPb + (long)c * a.L + q * 4);
      *reinterpret_cast<f32x4*>(sP + c * XPW_TOK + q * 4) = v;
    }
    for (int e = tid; e < a.Di * (XPW_TOK / 4); e += 256) {
      const int d = e / (XPW_TOK / 4), q = e % (XPW_TOK / 4);
      *reinterpret_cast<f32x4*>(sX + d * XPW_TOK + q * 4) =
          *reinterpret_cast<const f32x4*>(xb + (long)d * a.L + q * 4);
    }
    __syncthreads();
    if (active) {
      for (int q = ts; q < XPW_TOK / 4; q += TSL) {
        f32x4 pv[8], xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          pv[i] = *reinterpret_cast<const f32x4*>(sP + (c0 + i) * XPW_TOK + q * 4);
          xv[i] = *reinterpret_cast<const f32x4*>(sX + (d0 + i) * XPW_TOK + q * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[i][j] += pv[i][0] * xv[j][0] + pv[i][1] * xv[j][1] + pv[i][2] * xv[j][2] + pv[i][3] * xv[j][3];
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sred[(long)ts * C2p * a.Di + (c0 + i) * a.Di + d0 + j] = acc[i][j];
  }
  __syncthreads();
  for (int e = tid; e < a.C2 * a.Di; e += 256) {
    float v = 0.f;
    for (int q = 0; q < TSL; ++q) v += sred[(long)q * C2p * a.Di + e];
    const long o = xp_row(a, s, e / a.Di) + e % a.Di;
    if (a.part) a.part[(size_t)bx * 2 * a.C2 * a.Di + o] = v;   // plain store; nnz::fold_partials sums the rows in order
    else atomicAdd(a.dW + o, v);
  }
}

__global__ __launch_bounds__(256) void xproj_bwd_w_kernel(XpArgs a) { xproj_bwd_w_body(a, blockIdx.x, blockIdx.y); }

// grouped form (round 5): the x_proj weight gradients of every SS2D block of a backward pass in ONE launch over a job table
// (66 launches of 11-400 us per M2Net step), partial matrices per workgroup, one fold launch (nnz_group_fold_launch) - see
// csrc/token_linear.hip tl_wgrad_group_kernel for the protocol.  The grouped launch runs the product on the fp32 matrix cores:
//   dW[c][d] = sum_l dP[c][l] x2[d][l]: rows c (<= 3 tiles of 32), columns d (Di / 32 tiles), contraction = tokens, which are
//   CONTIGUOUS in both operands - both tiles go to LDS as [row][64 tokens] (pitch 65: lane = row reads hit 64 banks) and a lane's
//   MFMA operand of step st is one word, image[row][2 st + hh].  A workgroup covers <= 16 blocks of 32 x 32 (4 per wave;
//   <= 4 blocks: the waves split the round's 32 steps instead and are summed through LDS in wave order), blockIdx picks the
//   source, the token range and the block group.  (The FMA kernel above: 8 x 8 register tiles + a [slices][C2 Di] LDS fold,
//   3.5 ms for the 66 problems of an M2Net step even as one launch.)
constexpr int XPM_BPW = 4;                      // blocks per wave
constexpr int XPM_PITCH = XPW_TOK + 1;
__device__ __forceinline__ void xproj_bwd_w_mfma_body(const XpArgs& a, const unsigned bx, const int s, const int ygroup) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int CB = (a.C2 + 31) >> 5, DB = a.Di >> 5, NBK = CB * DB;        // blocks bi = db * CB + cb
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int g_begin = ygroup * 4 * XPM_BPW;
  const int g_n = NBK - g_begin < 4 * XPM_BPW ? NBK - g_begin : 4 * XPM_BPW;
  const int db_lo = g_begin / CB, db_hi = (g_begin + g_n - 1) / CB;       // d blocks this group touches
  const int nd = (db_hi - db_lo + 1) * 32;
  float* sP = smem_f;                          // [CB * 32][65]  (rows beyond C2: zero)
  float* sX = sP + CB * 32 * XPM_PITCH;        // [nd][65]
  const bool split_steps = g_n <= XPM_BPW;
  const int per = (g_n + 3) >> 2;
  const int b_begin = g_begin + (split_steps ? 0 : wave * per);
  int nblk = split_steps ? g_n : (g_n - wave * per < per ? g_n - wave * per : per);
  if (nblk < 0) nblk = 0;
  const int st_begin = split_steps ? wave * 8 : 0, st_end = split_steps ? wave * 8 + 8 : 32;
  f32x16 acc[XPM_BPW];
#pragma unroll
  for (int j = 0; j < XPM_BPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const long T = (long)a.B * a.L;
  const long t_begin = (long)bx * a.tokens_per_wg;
  long t_end = t_begin + a.tokens_per_wg;
  if (t_end > T) t_end = T;
  for (long tb = t_begin; tb < t_end; tb += XPW_TOK) {
    const int b = (int)(tb / a.L);               // a 64-token round never straddles two samples (launcher)
    const long l0 = tb - (long)b * a.L;
    const float* dPb = a.dP + ((long)(s * a.B + b) * a.C2) * a.L + l0;
    const float* xb = a.x2 + ((long)(s * a.B + b) * a.Di + db_lo * 32) * a.L + l0;
    __syncthreads();
    for (int e = tid; e < CB * 32 * (XPW_TOK / 4); e += 256) {
      const int c = e / (XPW_TOK / 4), q = e % (XPW_TOK / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < a.C2) v = *reinterpret_cast<const f32x4*>(dPb + (long)c * a.L + q * 4);
      float* d = sP + c * XPM_PITCH + q * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    for (int e = tid; e < nd * (XPW_TOK / 4); e += 256) {
      const int dd = e / (XPW_TOK / 4), q = e % (XPW_TOK / 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)dd * a.L + q * 4);
      float* d = sX + dd * XPM_PITCH + q * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
#pragma unroll 4
    for (int st = st_begin; st < st_end; ++st) {
      const int k = 2 * st + hh;
#pragma unroll
      for (int j = 0; j < XPM_BPW; ++j) {
        if (j < nblk) {                          // wave-uniform
          const int bi = b_begin + j;
          const int db = bi / CB, cb = bi - db * CB;
          const float av = sP[(cb * 32 + l31) * XPM_PITCH + k];
          const float bv = sX[((db - db_lo) * 32 + l31) * XPM_PITCH + k];
          acc[j] = xp_mfma(av, bv, acc[j]);
        }
      }
    }
  }
  // ---- the workgroup's partial matrix: rows of the module's weight layout -------------------------------------------------------
  float* prow = a.part + (size_t)bx * 2 * a.C2 * a.Di;
  __syncthreads();
  float* red = smem_f;                           // [4][32][32] (the images are dead)
  if (split_steps) {
#pragma unroll
    for (int j = 0; j < XPM_BPW; ++j) {
      if (j < nblk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + xp_crow(r, hh)) * 32 + l31] = acc[j][r];
        __syncthreads();
        const int bi = b_begin + j, db = bi / CB, cb = bi - db * CB;
        for (int e = tid; e < 1024; e += 256) {
          const int c = cb * 32 + (e >> 5), d = db * 32 + (e & 31);
          const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
          if (c < a.C2) prow[xp_row(a, s, c) + d] = v;
        }
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < XPM_BPW; ++j) {
      if (j < nblk) {
        const int bi = b_begin + j, db = bi / CB, cb = bi - db * CB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = cb * 32 + xp_crow(r, hh);
          if (c < a.C2) prow[xp_row(a, s, c) + db * 32 + l31] = acc[j][r];
        }
      }
    }
  }
}
static int xpm_ygroups(int Di, int C2) { return (((C2 + 31) / 32) * (Di / 32) + 4 * nnz::XPM_BPW - 1) / (4 * nnz::XPM_BPW); }
static size_t xpm_lds_bytes(int Di, int C2) {
  const int CB = (C2 + 31) / 32;
  const int DBg = (4 * nnz::XPM_BPW + CB - 1) / CB + 1;                     // d blocks one group can touch (upper bound)
  const int DB = Di / 32;
  const size_t img = sizeof(float) * (size_t)nnz::XPM_PITCH * 32 * (CB + (DBg < DB ? DBg : DB));
  return img > 16384 ? img : 16384;
}

struct XpJob {
  XpArgs a;
  int wg_begin;
  int ygroups;
};
__global__ __launch_bounds__(256, 2) void xproj_bwd_w_group_kernel(const XpJob* __restrict__ jobs, const int* __restrict__ wg_job) {
  const int j = __builtin_amdgcn_readfirstlane(wg_job[blockIdx.x]);
  const XpJob* jp = jobs + j;
  const XpArgs a = jp->a;
  unsigned local = blockIdx.x - (unsigned)jp->wg_begin;
  const unsigned yg = (unsigned)jp->ygroups;
  const int ygroup = (int)(local % yg);
  local /= yg;
  xproj_bwd_w_mfma_body(a, local >> 1, (int)(local & 1), ygroup);
}

static bool xp_shape_ok(int B, int Di, int C2, long L) {
  return B >= 1 && Di >= XP_DC && Di % XP_DC == 0 && Di <= 1024 && C2 >= 8 && C2 <= XP_CMAX && L >= 1;
}

}  // namespace nnz

// cp: 0 = W is [2][C2][Di]; > 0 = W is the module's [4][cp][Di] x_proj_weight (C2 must be 2 * cp), see XpArgs::Cp
extern "C" int nnz_ss2d_xproj_forward(const float* x2, const float* W, float* P, int B, int Di, int C2, long L, int cp,
                                      void* stream) {
  using namespace nnz;
  if (!x2 || !W || !P || !xp_shape_ok(B, Di, C2, L) || cp < 0 || (cp > 0 && 2 * cp != C2)) return NNZ_EINVAL;
  XpArgs a = {};
  a.x2 = x2; a.W = W; a.P = P; a.B = B; a.Di = Di; a.C2 = C2; a.L = L; a.Cp = cp;
  // output groups until the launch has ~256 workgroups
  const long base_wgs = ((L + 255) / 256) * 2 * B;
  const dim3 g1((unsigned)((L + 255) / 256), 2 * B, 1);
  static const bool use_mfma = [] { const char* e = getenv("NNZ_XPROJ_MFMA"); return !e || atoi(e) != 0; }();
  if (use_mfma && (L & 1) == 0) {      // matrix-core form (round 5); row tiles over blockIdx.z where the tokens alone are few
    if (base_wgs >= 128) NNZ_LAUNCH(xproj_fwd_mfma_kernel<3>, g1, dim3(256), 0, (hipStream_t)stream, a);
    else NNZ_LAUNCH(xproj_fwd_mfma_kernel<1>, dim3(g1.x, g1.y, (C2 + 31) / 32), dim3(256), 0, (hipStream_t)stream, a);
    NNZ_LAUNCH_CHECK();
    return NNZ_OK;
  }
  if (base_wgs >= 128) {
    NNZ_LAUNCH(xproj_fwd_kernel<XP_CMAX>, g1, dim3(256), 0, (hipStream_t)stream, a);
  } else if (base_wgs >= 64) {
    NNZ_LAUNCH(xproj_fwd_kernel<40>, dim3(g1.x, g1.y, (C2 + 39) / 40), dim3(256), 0, (hipStream_t)stream, a);
  } else if (base_wgs >= 32) {
    NNZ_LAUNCH(xproj_fwd_kernel<20>, dim3(g1.x, g1.y, (C2 + 19) / 20), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    NNZ_LAUNCH(xproj_fwd_kernel<10>, dim3(g1.x, g1.y, (C2 + 9) / 10), dim3(256), 0, (hipStream_t)stream, a);
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_ss2d_xproj_backward_x(const float* dP, const float* W, const float* du, float* dx2, int B, int Di,
                                         int C2, long L, int cp, void* stream) {
  using namespace nnz;
  if (!dP || !W || !du || !dx2 || !xp_shape_ok(B, Di, C2, L) || cp < 0 || (cp > 0 && 2 * cp != C2)) return NNZ_EINVAL;
  XpArgs a = {};
  a.dP = dP; a.W = W; a.du = du; a.dx2 = dx2; a.B = B; a.Di = Di; a.C2 = C2; a.L = L; a.Cp = cp;
  // channel slices (multiples of the 32-channel chunk) until the launch has ~256 workgroups
  const long base_wgs = ((L + 255) / 256) * 2 * B;
  int slices = 1;
  while (slices < 16 && base_wgs * slices < 256 && (Di / XP_DC) % (slices * 2) == 0) slices *= 2;
  static const bool use_mfma = [] { const char* e = getenv("NNZ_XPROJ_MFMA"); return !e || atoi(e) != 0; }();
  if (use_mfma && (L & 1) == 0)
    NNZ_LAUNCH(xproj_bwd_x_mfma_kernel, dim3((unsigned)((L + 255) / 256), 2 * B, slices), dim3(256), 0, (hipStream_t)stream, a);
  else
    NNZ_LAUNCH(xproj_bwd_x_kernel, dim3((unsigned)((L + 255) / 256), 2 * B, slices), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// dW[2][C2][Di] += per-source token contraction.  Needs L % 64 == 0 and (ceil8(C2)/8)(Di/8) <= 256; returns -22 otherwise (the
// caller keeps the library path for those shapes).  nnz_ss2d_xproj_backward_w: dW pre-zeroed, fp32 atomics over the token
// ranges; nnz_ss2d_xproj_backward_w_ws: dW is WRITTEN - one partial matrix per workgroup in `workspace`
// (nnz_ss2d_xproj_backward_w_workspace_floats) + a fixed-order fold: bit-reproducible.
static long xp_bwd_w_tokens_per_wg(long T) {
  long tpw = (T + 255) / 256;                              // ~256 workgroups per source (512 in all)
  return (tpw + nnz::XPW_TOK - 1) / nnz::XPW_TOK * nnz::XPW_TOK;
}
extern "C" long nnz_ss2d_xproj_backward_w_workspace_floats(int B, int Di, int C2, long L) {
  if (B < 1 || L < 1) return 0;
  const long T = (long)B * L, tpw = xp_bwd_w_tokens_per_wg(T);
  return ((T + tpw - 1) / tpw) * 2L * C2 * Di;
}
static int xp_bwd_w_impl(const float* dP, const float* x2, float* dW, float* workspace, long ws_floats, int B, int Di, int C2,
                         long L, int cp, void* stream) {
  using namespace nnz;
  if (!dP || !x2 || !dW || !xp_shape_ok(B, Di, C2, L) || (L % XPW_TOK) || cp < 0 || (cp > 0 && 2 * cp != C2))
    return NNZ_EINVAL;
  const int C2p = (C2 + 7) & ~7;
  const int nbk = (C2p >> 3) * (Di >> 3);
  if (nbk > 256) return NNZ_EINVAL;
  const int TSL = 256 / nbk;
  XpArgs a = {};
  a.dP = dP; a.x2 = x2; a.dW = dW; a.B = B; a.Di = Di; a.C2 = C2; a.L = L; a.Cp = cp;
  const long T = (long)B * L;
  const long tpw = xp_bwd_w_tokens_per_wg(T);
  a.tokens_per_wg = tpw;
  const long wgs = (T + tpw - 1) / tpw;
  if (workspace) {
    if (ws_floats < wgs * 2L * C2 * Di) return NNZ_EINVAL;
    a.part = workspace;
  }
  const size_t lds = sizeof(float) * ((size_t)(C2p + Di) * XPW_TOK + (size_t)TSL * C2p * Di);
  if (lds > 160 * 1024) return NNZ_EINVAL;
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(xproj_bwd_w_kernel), (int)lds, cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(xproj_bwd_w_kernel, dim3((unsigned)wgs, 2), dim3(256), lds, (hipStream_t)stream, a);
  if (workspace) {
    e = fold_partials(workspace, (int)wgs, 2L * C2 * Di, 2L * C2 * Di, dW, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_ss2d_xproj_backward_w(const float* dP, const float* x2, float* dW, int B, int Di, int C2, long L,
                                         int cp, void* stream) {
  return xp_bwd_w_impl(dP, x2, dW, nullptr, 0, B, Di, C2, L, cp, stream);
}
extern "C" int nnz_ss2d_xproj_backward_w_ws(const float* dP, const float* x2, float* dW, float* workspace, long ws_floats,
                                            int B, int Di, int C2, long L, int cp, void* stream) {
  if (!workspace) return NNZ_EINVAL;
  return xp_bwd_w_impl(dP, x2, dW, workspace, ws_floats, B, Di, C2, L, cp, stream);
}

// ---- grouped x_proj weight gradients: _plan gives workgroups (both sources), dynamic LDS bytes and workspace floats (one partial
// [2][C2][Di] matrix per token range); _fill writes a HOST record; fold: rows of 2 C2 Di floats, parts = workgroups / 2.
extern "C" int nnz_ss2d_xproj_backward_w_group_record_bytes(void) { return (int)sizeof(nnz::XpJob); }
extern "C" int nnz_ss2d_xproj_backward_w_group_plan(int B, int Di, int C2, long L, int* wgs, int* lds_bytes, long* ws_floats) {
  using namespace nnz;
  if (!xp_shape_ok(B, Di, C2, L) || (L % XPW_TOK) || !wgs || !lds_bytes || !ws_floats) return NNZ_EINVAL;
  const long T = (long)B * L, tpw = xp_bwd_w_tokens_per_wg(T);
  const long ranges = (T + tpw - 1) / tpw;
  const size_t lds = xpm_lds_bytes(Di, C2);
  if (lds > 160 * 1024) return NNZ_EINVAL;
  *wgs = (int)(2 * ranges * xpm_ygroups(Di, C2));          // sources x token ranges x block groups
  *lds_bytes = (int)lds;
  *ws_floats = ranges * 2L * C2 * Di;
  return NNZ_OK;
}
extern "C" int nnz_ss2d_xproj_backward_w_group_fill(void* job_host, const float* dP, const float* x2, float* workspace, int B,
                                                    int Di, int C2, long L, int cp, int wg_begin) {
  using namespace nnz;
  if (!job_host || !dP || !x2 || !workspace || !xp_shape_ok(B, Di, C2, L) || (L % XPW_TOK) || cp < 0 || (cp > 0 && 2 * cp != C2) ||
      wg_begin < 0)
    return NNZ_EINVAL;
  XpJob j = {};
  j.a.dP = dP; j.a.x2 = x2; j.a.B = B; j.a.Di = Di; j.a.C2 = C2; j.a.L = L; j.a.Cp = cp;
  j.a.tokens_per_wg = xp_bwd_w_tokens_per_wg((long)B * L);
  j.a.part = workspace;
  j.wg_begin = wg_begin;
  j.ygroups = xpm_ygroups(Di, C2);
  *reinterpret_cast<XpJob*>(job_host) = j;
  return NNZ_OK;
}
extern "C" int nnz_ss2d_xproj_backward_w_group_launch(const void* jobs_dev, const int* wg_job_dev, int total_wgs,
                                                      int max_lds_bytes, void* stream) {
  using namespace nnz;
  if (!jobs_dev || !wg_job_dev || total_wgs < 1 || max_lds_bytes < 1 || max_lds_bytes > 160 * 1024) return NNZ_EINVAL;
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(xproj_bwd_w_group_kernel), max_lds_bytes, cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(xproj_bwd_w_group_kernel, dim3((unsigned)total_wgs), dim3(256), max_lds_bytes, (hipStream_t)stream,
             (const XpJob*)jobs_dev, wg_job_dev);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
