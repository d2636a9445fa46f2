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
// (66 launches of 11-400 us per M2Net step, most of them a few hundred short workgroups), partial matrices per workgroup, one
// fold launch (nnz_group_fold_launch) - see csrc/token_linear.hip tl_wgrad_group_kernel for the protocol
struct XpJob {
  XpArgs a;
  int wg_begin;
  int pad;
};
__global__ __launch_bounds__(256) void xproj_bwd_w_group_kernel(const XpJob* __restrict__ jobs, const int* __restrict__ wg_job) {
  const int j = __builtin_amdgcn_readfirstlane(wg_job[blockIdx.x]);
  const XpJob* jp = jobs + j;
  const XpArgs a = jp->a;
  const unsigned local = blockIdx.x - (unsigned)jp->wg_begin;
  xproj_bwd_w_body(a, local >> 1, (int)(local & 1));
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
  const int C2p = (C2 + 7) & ~7;
  const int nbk = (C2p >> 3) * (Di >> 3);
  if (nbk > 256) return NNZ_EINVAL;
  const int TSL = 256 / nbk;
  const long T = (long)B * L, tpw = xp_bwd_w_tokens_per_wg(T);
  const long ranges = (T + tpw - 1) / tpw;
  const size_t lds = sizeof(float) * ((size_t)(C2p + Di) * XPW_TOK + (size_t)TSL * C2p * Di);
  if (lds > 160 * 1024) return NNZ_EINVAL;
  *wgs = (int)(2 * ranges);
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
