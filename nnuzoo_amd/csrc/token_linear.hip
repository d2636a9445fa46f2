// Token-major Linear layers with very many tokens and few features (gfx950): y[T][N] = x[T][K] W[N][K]^T + b.
// These are the in_proj / out_proj / patch-merge `reduction` / patch `expand` Linears of the VSS / SSND blocks
// (/root/reference/nnunetv2/nets/m2net.py:97,103,258,300; ssnd2net.py: same layers) under the autocast step of
// nnUNetTrainerM2Net / SSND2Net (fp16 operands, fp32 accumulate, fp16 result): 16..256 features, up to 5 x 10^5 tokens.
// A GEMM library sees M = tokens, N <= 256, K <= 256 and answers with a 64x64 or 16x16 macro tile and a K loop: 0.2-1 ms
// per call where the data is 20-90 MB (15-60 us at HBM speed); the M2Net step made 480 such calls (30 ms of 144).
//
// Design (memory-bound, so: one pass over x, one over y, weights never re-read from HBM):
//   * forward and input gradient are ONE kernel: out[T][Mo] = in[T][Kr] A[Mo][Kr]^T with A = W (forward) or A = W^T
//     (input gradient).  A is tiny: it is staged once per workgroup into LDS as fp16 MFMA fragments straight from the fp32
//     master parameter (no separate cast launch, no fp16 weight copy in HBM).
//   * v_mfma_f32_32x32x16_f16 with A as the A operand (rows = output features) and TOKENS on the lanes (B operand): a
//     lane's B fragment is 8 consecutive fp16 of its token's row = one 16-byte global load, no LDS staging for activations;
//     waves are persistent over 32-token tiles and prefetch the next tile's rows under the current tile's MFMAs.
//   * D comes out feature-major per lane (4 consecutive features of one token per accumulator quad); the tile is
//     transposed through a per-wave LDS image so that global stores are whole 16-byte pieces in token-major order.
//   * weight gradient dW[N][K] = sum_t dy[t][n] x[t][k] (+ bias gradient): 2 T N K <= ~1 GFLOP per call - plain fp32 FMAs
//     on register tiles (8 x 8 per thread), token ranges split over workgroups, one fp32 atomic per entry and workgroup.
#include "common.hpp"

namespace nnz {

struct TlArgs {
  const f16* in;      // [T][Kr]
  const float* W;     // fp32 master parameter [N][K] (torch layout)
  const float* bias;  // [Mo] or null
  f16* out;           // [T][Mo_real]
  long T;
  int Kr, Mo_real;    // reduction length, real number of output features (Mo_real <= 32 * MB)
  int transposed;     // 0: A[m][kk] = W[m][kk] (ld = Kr); 1: A[m][kk] = W[kk][m] (ld = Mo_real)
  int ntiles;
};

template <int MB, int KS>
__global__ __launch_bounds__(256) void tl_fwd_kernel(TlArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Mo = MB * 32, Kr = KS * 16;
  constexpr int ROWB = Mo * 2 + 16;  // per-wave output image: bytes per token row (+16 spreads the b64 writes over banks)
  char* sA = smem;                            // [MB][KS][32 rows][32 B], 16-byte halves swizzled by (row >> 4) & 1
  char* sO = smem + MB * KS * 1024;           // [4 waves][32 tokens][ROWB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;

  // ---- stage A (fp32 master -> fp16 fragments) -------------------------------------------------------------------
  // 16-byte global loads (four consecutive elements of the contiguous axis) - the element-by-element version made this
  // prologue 59 us for the 256 x 128 / 128 x 256 weights of the deep levels, whatever the token count
  // (profiles/r03_m2net_kernel_histogram.txt: tl_fwd_kernel<8, 8> / <4, 16> flat at 57-68 us for 16-256 workgroups)
  auto put = [&](int m, int kk, float v) {
    const int mb = m >> 5, row = m & 31, ks = kk >> 4, k16 = kk & 15;
    const int half = (k16 >> 3) ^ ((row >> 4) & 1);
    *reinterpret_cast<f16*>(sA + ((mb * KS + ks) * 32 + row) * 32 + half * 16 + (k16 & 7) * 2) = (f16)v;
  };
  if (!a.transposed) {
    for (int e4 = tid; e4 < Mo * Kr / 4; e4 += 256) {
      const int m = e4 / (Kr / 4), kk = (e4 % (Kr / 4)) * 4;     // W[m][kk .. kk + 3], kk contiguous
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.Mo_real) v = *reinterpret_cast<const f32x4*>(a.W + (long)m * Kr + kk);
      const int mb = m >> 5, row = m & 31, ks = kk >> 4, k16 = kk & 15;
      const int half = (k16 >> 3) ^ ((row >> 4) & 1);
      const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(sA + ((mb * KS + ks) * 32 + row) * 32 + half * 16 + (k16 & 7) * 2) = h;
    }
  } else if ((a.Mo_real & 3) == 0) {
    for (int e4 = tid; e4 < Mo * Kr / 4; e4 += 256) {
      const int kk = e4 / (Mo / 4), m = (e4 % (Mo / 4)) * 4;     // W[kk][m .. m + 3], m contiguous
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.Mo_real) v = *reinterpret_cast<const f32x4*>(a.W + (long)kk * a.Mo_real + m);
#pragma unroll
      for (int q = 0; q < 4; ++q) put(m + q, kk, v[q]);
    }
  } else {
    for (int e = tid; e < Mo * Kr; e += 256) {
      const int kk = e / Mo, m = e % Mo;
      put(m, kk, m < a.Mo_real ? a.W[(long)kk * a.Mo_real + m] : 0.f);
    }
  }
  __syncthreads();
  const int a_lane = l31 * 32 + ((hh ^ ((l31 >> 4) & 1)) << 4);
  char* sOw = sO + wave * 32 * ROWB;

  const int nwaves = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  auto load_tile = [&](int tl, f16x8 (&b)[KS]) {
    const long t = (long)tl * 32 + l31;
    const bool ok = tl < a.ntiles && t < a.T;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) v = *reinterpret_cast<const f16x8*>(a.in + t * Kr + ks * 16 + hh * 8);
      b[ks] = v;
    }
  };
  f16x8 bcur[KS], bnxt[KS];
  load_tile(tile, bcur);
  for (; tile < a.ntiles; tile += nwaves) {
    load_tile(tile + nwaves, bnxt);
    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(sA + (mb * KS + ks) * 1024 + a_lane);
        acc[mb] = mfma32(af, bcur[ks], acc[mb]);
      }
    }
    // ---- epilogue: D[row = feature][col = token]; lane holds features (r&3) + 8(r>>2) + 4hh of token l31 ----------
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        const int f0 = mb * 32 + 8 * q + 4 * hh;
        if (a.bias && f0 + 3 < a.Mo_real) bv = *reinterpret_cast<const f32x4*>(a.bias + f0);
        f16x4 o = {(f16)(acc[mb][4 * q + 0] + bv[0]), (f16)(acc[mb][4 * q + 1] + bv[1]),
                   (f16)(acc[mb][4 * q + 2] + bv[2]), (f16)(acc[mb][4 * q + 3] + bv[3])};
        *reinterpret_cast<f16x4*>(sOw + l31 * ROWB + f0 * 2) = o;
      }
    // same-wave LDS: program order is enough (no barrier; the image is private to the wave)
    const int ppr = a.Mo_real >> 3;  // 16-byte pieces per token row
    const long tbase = (long)tile * 32;
    for (int p = lane; p < 32 * ppr; p += 64) {
      const int tk = p / ppr, c8 = p % ppr;
      if (tbase + tk < a.T)
        *reinterpret_cast<f16x8*>(a.out + (tbase + tk) * a.Mo_real + c8 * 8) =
            *reinterpret_cast<const f16x8*>(sOw + tk * ROWB + c8 * 16);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnxt[ks];
  }
}

template <int MB, int KS>
static int launch_tl(const TlArgs& a, hipStream_t s) {
  const int lds = MB * KS * 1024 + 4 * 32 * (MB * 64 + 16);
  auto kern = tl_fwd_kernel<MB, KS>;
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, cache);
  if (e != hipSuccess) return (int)e;
  int wgs = (a.ntiles + 3) / 4;
  const int per_cu = lds > 80 * 1024 ? 1 : lds > 40 * 1024 ? 2 : 4;
  if (wgs > 256 * per_cu) wgs = 256 * per_cu;  // persistent waves: a few workgroups per CU
  NNZ_LAUNCH(kern, dim3(wgs), dim3(256), lds, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <int MB>
static int dispatch_ks(const TlArgs& a, int ks, hipStream_t s) {
  switch (ks) {
    case 1: return launch_tl<MB, 1>(a, s);
    case 2: return launch_tl<MB, 2>(a, s);
    case 4: return launch_tl<MB, 4>(a, s);
    case 8: return launch_tl<MB, 8>(a, s);
    case 16: return launch_tl<MB, 16>(a, s);
  }
  return NNZ_EINVAL;
}

// ---- weight (+ bias) gradient ------------------------------------------------------------------------------------
struct TlWgArgs {
  const f16* dy;  // [T][N]
  const f16* x;   // [T][K]
  float* dW;      // [N][K] fp32, pre-zeroed (atomic accumulation over workgroups)
  float* db;      // [N] or null, pre-zeroed
  long T;
  int N, K;
  long tokens_per_wg;
  float* part;    // null, or [workgroups][N * K + N]: two-stage (deterministic) mode
};

// thread = (token slice ts, 8x8 block of dW); TSL token slices of the workgroup run in parallel over its token range and
// are folded through LDS at the end.  Rows of dy / x are broadcast LDS reads (16 bytes = 8 fp16 per operand and token).
constexpr int TLW_TOK = 64;  // tokens staged per round

__device__ __forceinline__ void tl_wgrad_body(const TlWgArgs& a, const unsigned bx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, K = a.K;
  const int nbk = (N >> 3) * (K >> 3);         // 8x8 blocks of dW
  const int TSL = 256 / nbk > 0 ? 256 / nbk : 1;  // token slices (>= 1); nbk <= 256 is checked by the launcher
  f16* sdy = reinterpret_cast<f16*>(smem);                      // [TLW_TOK][N]
  f16* sx = sdy + TLW_TOK * N;                                  // [TLW_TOK][K]
  float* sred = reinterpret_cast<float*>(sx + TLW_TOK * K);     // [TSL][N*K] fold buffer (+ [TSL][N] bias)
  const int tid = threadIdx.x;
  const int blk = tid % nbk, ts = tid / nbk;
  const bool active = ts < TSL;
  const int n0 = (blk / (K >> 3)) * 8, k0 = (blk % (K >> 3)) * 8;
  float acc[8][8];
  float accb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    accb[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  }
  const long t_begin = (long)bx * a.tokens_per_wg;
  long t_end = t_begin + a.tokens_per_wg;
  if (t_end > a.T) t_end = a.T;
  const int pn = N >> 3, pk = K >> 3;  // 16-byte pieces per row
  for (long tb = t_begin; tb < t_end; tb += TLW_TOK) {
    __syncthreads();
    const int nt = (int)((t_end - tb) < TLW_TOK ? (t_end - tb) : TLW_TOK);
    for (int p = tid; p < nt * pn; p += 256)
      reinterpret_cast<f16x8*>(sdy)[p] = *reinterpret_cast<const f16x8*>(a.dy + tb * N + (long)p * 8);
    for (int p = tid; p < nt * pk; p += 256)
      reinterpret_cast<f16x8*>(sx)[p] = *reinterpret_cast<const f16x8*>(a.x + tb * K + (long)p * 8);
    __syncthreads();
    if (active) {
      for (int tt = ts; tt < nt; tt += TSL) {
        const f16x8 d8 = *reinterpret_cast<const f16x8*>(sdy + tt * N + n0);
        const f16x8 x8 = *reinterpret_cast<const f16x8*>(sx + tt * K + k0);
        float df[8], xf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          df[i] = (float)d8[i];
          xf[i] = (float)x8[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (k0 == 0) accb[i] += df[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] += df[i] * xf[j];
        }
      }
    }
  }
  // ---- fold the token slices, one atomic per entry -----------------------------------------------------------------
  __syncthreads();
  if (active) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sred[(long)ts * N * K + (n0 + i) * K + k0 + j] = acc[i][j];
    if (k0 == 0)
#pragma unroll
      for (int i = 0; i < 8; ++i) sred[(long)TSL * N * K + ts * N + n0 + i] = accb[i];
  }
  __syncthreads();
  // a.part (nnz_token_linear_wgrad_ws): the workgroup's sums go to its own row of the workspace with plain stores and a second
  // kernel folds the rows in a fixed order - bit-reproducible; otherwise one fp32 atomic per entry into the pre-zeroed dW / db
  float* prow = a.part ? a.part + (size_t)bx * ((size_t)N * K + N) : nullptr;
  for (int e = tid; e < N * K; e += 256) {
    float v = 0.f;
    for (int s_ = 0; s_ < TSL; ++s_) v += sred[(long)s_ * N * K + e];
    if (prow) prow[e] = v;
    else atomicAdd(a.dW + e, v);
  }
  if (a.db || prow)
    for (int e = tid; e < N; e += 256) {
      float v = 0.f;
      for (int s_ = 0; s_ < TSL; ++s_) v += sred[(long)TSL * N * K + s_ * N + e];
      if (prow) prow[(size_t)N * K + e] = v;
      else atomicAdd(a.db + e, v);
    }
}

__global__ __launch_bounds__(256) void tl_wgrad_kernel(TlWgArgs a) { tl_wgrad_body(a, blockIdx.x); }

// ---- grouped weight gradients (round 5): ALL weight gradients of the fp16 token Linears of one backward pass in one launch -----
// A Mamba-net step holds ~160 of them (in_proj / out_proj / patch merge / expand of every VSS block; 4 ms of an 83 ms M2Net step
// as 160 launches of ~21 us each, almost all of it launch latency at a few hundred workgroups of a few microseconds).  They are
// not on the data-gradient chain: the autograd nodes queue (dy, x, dW, db) and the end of the pass runs every problem in ONE launch
// over a job table (workgroup -> job map in device memory), every workgroup writing its partial block to its job's slice of the
// workspace, and ONE fold launch sums the blocks in workgroup order (nnz_group_fold_launch): no zero fills, no float atomics -
// bit-identical from run to run.
// ---- the same product on the matrix cores (round 5; the grouped launch runs this body) ---------------------------------------
// The FMA kernel above keeps 8 x 8 register blocks per thread and folds 256 / (N K / 64) token slices through a [slices][N K]
// LDS buffer - 67 KB for a 16 x 32 weight, two workgroups per CU, a barrier pair per 64 tokens and 128 FMAs per thread between
// them: grouped into one launch the ~160 problems of an M2Net step still took 3.1 ms (profiles/r05_m2net_graph_kernels_grouped_first.txt).
// dW[n][k] = sum_t dy[t][n] x[t][k] is a GEMM whose contraction index (the token) is the SLOW axis of both operands - exactly what
// ds_read_b64_tr_b16 is for: both token-major fp16 tiles go to LDS as they are (16-byte pieces), and the MFMA fragments
// (feature on the lane, 8 consecutive tokens in the lane's elements) are transposed reads of them (as in csrc/conv_wgrad.hip).
//   * the weight matrix is cut into 32 x 32 blocks (v_mfma_f32_32x32x16_f16, fp32 accumulate); a workgroup covers <= 32 blocks
//     (blockIdx y picks the group for the 256 x 128 / 256 x 256 weights), a token range of tokens_per_wg tokens, 64 per round;
//   * <= 8 blocks: the four waves split the round's four 16-token steps and their accumulators are summed through LDS at the
//     end (in wave order); more: a wave owns <= 8 blocks and runs all four steps;
//   * image pitch = 16 words (mod 32): the four rows x two 16-column groups of a 32-lane half land on 32 different banks;
//   * bias gradient: column sums of the dy tile by 16-byte pieces on the vector ALU (four per thread and round at N = 256);
//   * the workgroup's block goes to its row of the workspace (layout dW | db as before) - the fold kernel sums the rows in order.
__device__ __forceinline__ int tlw_pitch_words(int cols) {          // cols = features rounded up to 32
  const int w = cols >> 1;
  return ((cols >> 5) & 1) ? w : w + 16;
}
__device__ __forceinline__ f16x8 tlw_frag(const char* img, int pitch_bytes, int row0, int colbyte0, int lane) {
  const int q = (lane >> 2) & 3, p = lane & 3, g = (lane >> 4) & 1, hh = lane >> 5;
  const char* a = img + (row0 + 8 * hh + q) * pitch_bytes + colbyte0 + g * 32 + p * 8;
  union { i16x4 v[2]; f16x8 h; } u;
  u.v[0] = lds_read_tr16(a);
  u.v[1] = lds_read_tr16(a + 4 * pitch_bytes);
  return u.h;
}

constexpr int TLM_TOK = 64;        // tokens per round (= TLW_TOK: the token ranges of tl_wgrad_tokens_per_wg are multiples of it)
constexpr int TLM_BPW = 8;         // blocks per wave
__device__ __forceinline__ void tl_wgrad_mfma_body(const TlWgArgs& a, const unsigned bx, const int ygroup) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, K = a.K;
  const int NB = (N + 31) >> 5, KB = (K + 31) >> 5, NBK = NB * KB;
  const int pn = tlw_pitch_words(NB * 32) * 4, pk = tlw_pitch_words(KB * 32) * 4;      // bytes
  char* sdy = smem;
  char* sx = smem + TLM_TOK * pn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  // this workgroup's blocks, and this wave's share of them / of the round's four 16-token steps
  const int g_begin = ygroup * 4 * TLM_BPW;
  const int g_n = NBK - g_begin < 4 * TLM_BPW ? NBK - g_begin : 4 * TLM_BPW;
  const bool split_steps = g_n <= TLM_BPW;                 // few blocks: waves take steps, not blocks
  const int per = (g_n + 3) >> 2;
  const int b_begin = g_begin + (split_steps ? 0 : wave * per);
  int nblk = split_steps ? g_n : (g_n - wave * per < per ? g_n - wave * per : per);
  if (nblk < 0) nblk = 0;
  const int ks_begin = split_steps ? wave : 0, ks_end = split_steps ? wave + 1 : 4;
  f32x16 acc[TLM_BPW];
#pragma unroll
  for (int j = 0; j < TLM_BPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // bias: thread = (16-byte piece of the row, token lane)
  const int ppn = N >> 3, ppk = K >> 3;
  const int bp = tid % ppn, btl = tid / ppn, bstep = 256 / ppn;
  float accb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) accb[i] = 0.f;
  // zero the images once: padding columns and rows beyond a short last round stay zero / are rewritten below
  for (int e = tid * 16; e < TLM_TOK * (pn + pk); e += 256 * 16) *reinterpret_cast<u32x4*>(smem + e) = u32x4{0, 0, 0, 0};
  const long t_begin = (long)bx * a.tokens_per_wg;
  long t_end = t_begin + a.tokens_per_wg;
  if (t_end > a.T) t_end = a.T;
  for (long tb = t_begin; tb < t_end; tb += TLM_TOK) {
    __syncthreads();
    const int nt = (int)((t_end - tb) < TLM_TOK ? (t_end - tb) : TLM_TOK);
    for (int p = tid; p < TLM_TOK * ppn; p += 256) {
      const int t = p / ppn, c = p - t * ppn;
      u32x4 v = {0, 0, 0, 0};
      if (t < nt) v = *reinterpret_cast<const u32x4*>(a.dy + (tb + t) * N + c * 8);
      *reinterpret_cast<u32x4*>(sdy + t * pn + c * 16) = v;
    }
    for (int p = tid; p < TLM_TOK * ppk; p += 256) {
      const int t = p / ppk, c = p - t * ppk;
      u32x4 v = {0, 0, 0, 0};
      if (t < nt) v = *reinterpret_cast<const u32x4*>(a.x + (tb + t) * K + c * 8);
      *reinterpret_cast<u32x4*>(sx + t * pk + c * 16) = v;
    }
    __syncthreads();
    if (ygroup == 0 && btl < TLM_TOK) {
      for (int t = btl; t < TLM_TOK; t += bstep) {
        const f16x8 d8 = *reinterpret_cast<const f16x8*>(sdy + t * pn + bp * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) accb[i] += (float)d8[i];
      }
    }
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      int nb_cur = -1;
      f16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < TLM_BPW; ++j) {
        if (j < nblk) {                                    // wave-uniform
          const int bi = b_begin + j;
          const int nb = bi / KB, kb = bi - nb * KB;
          if (nb != nb_cur) {
            af = tlw_frag(sdy, pn, 16 * ks, nb * 64, lane);
            nb_cur = nb;
          }
          const f16x8 bf = tlw_frag(sx, pk, 16 * ks, kb * 64, lane);
          acc[j] = mfma32(af, bf, acc[j]);
        }
      }
    }
  }
  // ---- results: dW block (row n = 32 nb + crow(r, hh), column k = 32 kb + l31), bias -------------------------------------------
  float* prow = a.part + (size_t)bx * ((size_t)N * K + N);
  __syncthreads();                                         // the images are dead: their space carries the cross-wave sums
  float* red = reinterpret_cast<float*>(smem);
  if (split_steps) {
#pragma unroll
    for (int j = 0; j < TLM_BPW; ++j) {
      if (j < nblk) {                                      // workgroup-uniform (split_steps: every wave has the same blocks)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + l31] = acc[j][r];
        __syncthreads();
        const int bi = b_begin + j, nb = bi / KB, kb = bi - nb * KB;
        for (int e = tid; e < 1024; e += 256) {
          const int n = nb * 32 + (e >> 5), k = kb * 32 + (e & 31);
          const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
          if (n < N && k < K) prow[(size_t)n * K + k] = v;
        }
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TLM_BPW; ++j) {
      if (j < nblk) {
        const int bi = b_begin + j, nb = bi / KB, kb = bi - nb * KB;
        const int k = kb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (n < N && k < K) prow[(size_t)n * K + k] = acc[j][r];
        }
      }
    }
  }
  if (ygroup == 0) {                                       // bias: fold the token lanes of every piece in lane order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tid * 8 + i] = accb[i];
    __syncthreads();
    for (int c = tid; c < N; c += 256) {
      const int piece = c >> 3, i = c & 7;
      float v = 0.f;
      for (int tl = 0; tl * ppn + piece < 256; ++tl) v += red[(tl * ppn + piece) * 8 + i];
      prow[(size_t)N * K + c] = v;
    }
  }
}
// block groups (blockIdx y) and dynamic LDS of the matrix-core body
static int tlm_ygroups(int N, int K) { return (((N + 31) / 32) * ((K + 31) / 32) + 4 * nnz::TLM_BPW - 1) / (4 * nnz::TLM_BPW); }
static int tlm_pitch_words_host(int cols) { return (((cols >> 5) & 1) ? (cols >> 1) : (cols >> 1) + 16); }
static size_t tlm_lds_bytes(int N, int K) {
  const size_t img = (size_t)nnz::TLM_TOK * 4 * (tlm_pitch_words_host((N + 31) / 32 * 32) + tlm_pitch_words_host((K + 31) / 32 * 32));
  return img > 16384 ? img : 16384;                        // >= the [4][32][32] cross-wave buffer and the [256][8] bias buffer
}

struct TlWgJob {
  TlWgArgs a;
  int wg_begin;
  int ygroups;
};
__global__ __launch_bounds__(256, 2) void tl_wgrad_group_kernel(const TlWgJob* __restrict__ jobs, const int* __restrict__ wg_job) {
  const int j = __builtin_amdgcn_readfirstlane(wg_job[blockIdx.x]);
  const TlWgJob* jp = jobs + j;
  const TlWgArgs a = jp->a;
  const unsigned local = blockIdx.x - (unsigned)jp->wg_begin;
  const unsigned yg = (unsigned)jp->ygroups;
  tl_wgrad_mfma_body(a, local / yg, (int)(local % yg));
}

}  // namespace nnz

// out[T][Mo] (f16) = in[T][Kr] (f16) A^T + bias, A = W (transposed = 0: W is [Mo][Kr]) or W^T (transposed = 1: W is
// [Kr][Mo]); W fp32.  Kr in {16, 32, 64, 128, 256}; Mo % 8 == 0, Mo <= 256, Mo padded to {32, 64, 128, 256} internally.
extern "C" int nnz_token_linear_forward(const void* in_f16, const float* W, const float* bias, void* out_f16, long T,
                                        int Kr, int Mo, int transposed, void* stream) {
  using namespace nnz;
  if (!in_f16 || !W || !out_f16 || T < 1 || Mo < 8 || Mo > 256 || (Mo & 7) || T > (1L << 36)) return NNZ_EINVAL;
  if (Kr != 16 && Kr != 32 && Kr != 64 && Kr != 128 && Kr != 256) return NNZ_EINVAL;
  const int mb = Mo <= 32 ? 1 : Mo <= 64 ? 2 : Mo <= 128 ? 4 : 8;
  if (mb * 32 * Kr > 65536) return NNZ_EINVAL;  // A fragments in LDS: at most 128 KB of fp16 incl. the output images
  // register budget: MB accumulators (16 VGPRs each) + two sets of KS token fragments (4 VGPRs each)
  if (mb * 16 + 2 * (Kr / 16) * 4 > 208) return NNZ_EINVAL;
  TlArgs a = {};
  a.in = (const f16*)in_f16; a.W = W; a.bias = bias; a.out = (f16*)out_f16;
  a.T = T; a.Kr = Kr; a.Mo_real = Mo; a.transposed = transposed;
  a.ntiles = (int)((T + 31) / 32);
  hipStream_t s = (hipStream_t)stream;
  switch (mb) {
    case 1: return dispatch_ks<1>(a, Kr / 16, s);
    case 2: return dispatch_ks<2>(a, Kr / 16, s);
    case 4: return dispatch_ks<4>(a, Kr / 16, s);
    case 8: return dispatch_ks<8>(a, Kr / 16, s);
  }
  return NNZ_EINVAL;
}

// 1 if nnz_token_linear_forward supports (Kr, Mo), else 0 (the host falls back to the library GEMM)
extern "C" int nnz_token_linear_supported(int Kr, int Mo) {
  if (Kr != 16 && Kr != 32 && Kr != 64 && Kr != 128 && Kr != 256) return 0;
  if (Mo < 8 || Mo > 256 || (Mo & 7)) return 0;
  const int mb = Mo <= 32 ? 1 : Mo <= 64 ? 2 : Mo <= 128 ? 4 : 8;
  if (mb * 32 * Kr > 65536) return 0;
  if (mb * 16 + 2 * (Kr / 16) * 4 > 208) return 0;
  return 1;
}

// dW[N][K] += sum_t dy[t][n] x[t][k];  db[n] += sum_t dy[t][n]  (N, K multiples of 8, (N/8)(K/8) <= 256).
// nnz_token_linear_wgrad: dW / db pre-zeroed fp32, workgroups add with fp32 atomics (the sum's low bits depend on arrival order).
// nnz_token_linear_wgrad_ws: dW / db are WRITTEN (no pre-zeroing); `workspace` of nnz_token_linear_wgrad_workspace_floats(T, N, K)
// floats receives one partial block per workgroup and a second kernel folds them in a fixed order: bit-reproducible.
static long tl_wgrad_tokens_per_wg(long T) {
  // ~512 workgroups (two per CU) whatever T is: the low-resolution levels have few tokens but large N x K, and a
  // workgroup's serial loop is tokens x 64 FMAs per thread (the first version fixed 2048 tokens per workgroup: 16
  // workgroups and 280 us for T = 32 768, N K = 16 384).
  long tpw = (T + 511) / 512;
  return (tpw + nnz::TLW_TOK - 1) / nnz::TLW_TOK * nnz::TLW_TOK;
}
extern "C" long nnz_token_linear_wgrad_workspace_floats(long T, int N, int K) {
  if (T < 1 || N < 8 || K < 8) return 0;
  const long tpw = tl_wgrad_tokens_per_wg(T);
  return ((T + tpw - 1) / tpw) * ((long)N * K + N);
}
static int tl_wgrad_impl(const void* dy_f16, const void* x_f16, float* dW, float* db, float* workspace, long ws_floats,
                         long T, int N, int K, void* stream) {
  using namespace nnz;
  if (!dy_f16 || !x_f16 || !dW || T < 1 || (N & 7) || (K & 7) || N < 8 || K < 8) return NNZ_EINVAL;
  const int nbk = (N >> 3) * (K >> 3);
  if (nbk > 256) return NNZ_EINVAL;
  const int TSL = 256 / nbk;
  TlWgArgs a = {};
  a.dy = (const f16*)dy_f16; a.x = (const f16*)x_f16; a.dW = dW; a.db = db;
  a.T = T; a.N = N; a.K = K;
  const long tpw = tl_wgrad_tokens_per_wg(T);
  a.tokens_per_wg = tpw;
  long wgs = (T + tpw - 1) / tpw;
  if (workspace) {
    if (ws_floats < wgs * ((long)N * K + N)) return NNZ_EINVAL;
    a.part = workspace;
  }
  const size_t lds = (size_t)TLW_TOK * (N + K) * 2 + (size_t)TSL * (N * K + N) * 4;
  if (lds > 160 * 1024) return NNZ_EINVAL;
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(tl_wgrad_kernel), (int)lds, cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(tl_wgrad_kernel, dim3((unsigned)wgs), dim3(256), lds, (hipStream_t)stream, a);
  if (workspace) {
    const long row = (long)N * K + N;
    e = fold_partials(workspace, (int)wgs, row, (long)N * K, dW, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (db) {
      e = fold_partials(workspace + (size_t)N * K, (int)wgs, row, N, db, (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
    }
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_token_linear_wgrad(const void* dy_f16, const void* x_f16, float* dW, float* db, long T, int N, int K,
                                      void* stream) {
  return tl_wgrad_impl(dy_f16, x_f16, dW, db, nullptr, 0, T, N, K, stream);
}
extern "C" int nnz_token_linear_wgrad_ws(const void* dy_f16, const void* x_f16, float* dW, float* db, float* workspace,
                                         long ws_floats, long T, int N, int K, void* stream) {
  if (!workspace) return NNZ_EINVAL;
  return tl_wgrad_impl(dy_f16, x_f16, dW, db, workspace, ws_floats, T, N, K, stream);
}

// ---- grouped form (see tl_wgrad_group_kernel).  Host protocol as for nnz_dense32_group_*: plan every queued problem (workgroups,
// dynamic LDS bytes, workspace floats = workgroups * (N K + N)), lay the jobs out back to back, fill one record per job into a
// HOST table of nnz_token_linear_wgrad_group_record_bytes() bytes each, build the int32 map workgroup -> job, copy both to the
// device, launch with the LARGEST LDS size of the group; then fold each job's partial blocks (rows of N K + N floats: dW then db;
// their number is ws_floats / (N K + N) - the plan's workgroup count also carries the block groups of wide weights)
// with fold records (nnz_dense32_group_fill_fold) and nnz_group_fold_launch.  N, K <= 256, multiples of 8.
extern "C" int nnz_token_linear_wgrad_group_record_bytes(void) { return (int)sizeof(nnz::TlWgJob); }
extern "C" int nnz_token_linear_wgrad_group_plan(long T, int N, int K, int* wgs, int* lds_bytes, long* ws_floats) {
  using namespace nnz;
  if (T < 1 || (N & 7) || (K & 7) || N < 8 || K < 8 || !wgs || !lds_bytes || !ws_floats) return NNZ_EINVAL;
  if (N > 256 || K > 256) return NNZ_EINVAL;
  const long tpw = tl_wgrad_tokens_per_wg(T);
  const long w = (T + tpw - 1) / tpw;                      // token ranges = partial blocks to fold
  const size_t lds = tlm_lds_bytes(N, K);
  const int yg = tlm_ygroups(N, K);
  if (lds > 160 * 1024 || w * yg > (1L << 30)) return NNZ_EINVAL;
  *wgs = (int)(w * yg);                                    // workgroups: token ranges x block groups
  *lds_bytes = (int)lds;
  *ws_floats = w * ((long)N * K + N);
  return NNZ_OK;
}
extern "C" int nnz_token_linear_wgrad_group_fill(void* job_host, const void* dy_f16, const void* x_f16, float* workspace, long T,
                                                 int N, int K, int wg_begin) {
  using namespace nnz;
  if (!job_host || !dy_f16 || !x_f16 || !workspace || T < 1 || (N & 7) || (K & 7) || N < 8 || K < 8 || wg_begin < 0)
    return NNZ_EINVAL;
  TlWgJob j = {};
  j.a.dy = (const f16*)dy_f16; j.a.x = (const f16*)x_f16; j.a.T = T; j.a.N = N; j.a.K = K;
  j.a.tokens_per_wg = tl_wgrad_tokens_per_wg(T);
  j.a.part = workspace;
  j.wg_begin = wg_begin;
  j.ygroups = tlm_ygroups(N, K);
  *reinterpret_cast<TlWgJob*>(job_host) = j;
  return NNZ_OK;
}
extern "C" int nnz_token_linear_wgrad_group_launch(const void* jobs_dev, const int* wg_job_dev, int total_wgs, int max_lds_bytes,
                                                   void* stream) {
  using namespace nnz;
  if (!jobs_dev || !wg_job_dev || total_wgs < 1 || max_lds_bytes < 1 || max_lds_bytes > 160 * 1024) return NNZ_EINVAL;
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(tl_wgrad_group_kernel), max_lds_bytes, cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(tl_wgrad_group_kernel, dim3((unsigned)total_wgs), dim3(256), max_lds_bytes, (hipStream_t)stream,
             (const TlWgJob*)jobs_dev, wg_job_dev);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
