// fp32 token-major Linear layers on the fp32 matrix cores (gfx950, v_mfma_f32_32x32x2_f32): forward (+ bias, + GELU),
// input gradient (+ GELU'), weight + bias gradient.  The Swin / ViT trainers of the reference run WITHOUT autocast
// (/root/reference/nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwT2Net.py:112-130), so the Linear layers that hold 80 %
// of SwT2Net's FLOPs - qkv / proj (/root/reference/nnunetv2/nets/swt2net.py:584-619), the Mlp (:496-515), patch merging /
// expanding and the skip fusions (:843-868) - are exact fp32 contractions.  Rounds 1-2 sent them to the GEMM library:
// per Linear and step one forward GEMM, one input-gradient GEMM, a chunked batched GEMM + sum for the weight gradient, a
// reduce launch for the bias gradient, plus element-wise launches for bias / GELU / GELU' - ~9 000 small kernels per step.
//
// One kernel template serves the three products; what differs is only how an operand tile reaches LDS:
//   forward   y[t][n]  = sum_k x[t][k]  W[n][k]   rows = tokens (x, k contiguous)   cols = features (W, k contiguous)
//   dgrad     dx[t][k] = sum_n dy[t][n] W[n][k]   rows = tokens (dy, n contiguous)  cols = k (W read transposed)
//   wgrad     dW[n][k] = sum_t dy[t][n] x[t][k]   rows = n (dy read transposed)     cols = k (x read transposed)
// Lane half hh takes contraction steps 4hh .. 4hh+3 of every 8 - any assignment of steps to MFMA issues is valid as long as
// both operands use the same one - so an operand whose contraction index is contiguous in memory is fetched from its
// [row][step] LDS image with one ds_read_b128 per four MFMAs; an operand whose ROW index is contiguous keeps that layout in
// LDS ([step][row]: 16-byte staging writes, no scattered transposition) and is fetched with four conflict-free ds_read_b32.  Rows (tokens) are the MFMA A operand and features sit
// on the lanes, so an accumulator register is 32 consecutive features of one token: stores are 128-byte runs without a
// transpose.  4 waves per workgroup in a 2 x 2 grid, each wave WM x WN MFMA tiles (up to 64 x 64 outputs = 64 accumulator
// registers); the next contraction block is prefetched into registers under the MFMAs of the current one.
// The weight gradient splits the token axis over workgroups; partials go to a workspace and a second kernel folds them in
// a fixed order (deterministic: no float atomics), the bias gradient rides along as the column sums of the dy tile.
// (Folding inside the same launch - the split that finishes last re-reads its tile's partials - was tried: the partials
// then have to travel through agent-scope 4-byte stores and loads, which made the SwT2Net step 40 % slower.)
#include "common.hpp"

namespace nnz {

constexpr int D32_BK = 32;          // contraction steps per LDS block of the large tiles (64 for the 64 x 64 tile)

__device__ __forceinline__ f32x16 mfma_f32x(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// tanh-free exact GELU (torch.nn.GELU default, approximate='none'): 0.5 x (1 + erf(x / sqrt 2))
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  return cdf + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

struct D32Args {
  const float* A;       // row operand source
  const float* B;       // column operand source
  float* out;           // [rows][ldo]
  float* out2;          // forward + GELU: the activated copy (out keeps the pre-activation for the backward)
  const float* bias;    // per column, or null
  const float* aux;     // dgrad + GELU': pre-activation h[rows][cols] (same layout as out)
  float* part_db;       // wgrad: [splits][rows] column sums of dy (null: none)
  long a_rs, a_cs;      // element (row r, contraction c) of A at A[r * a_rs + c * a_cs]   (one of the two strides is 1)
  long b_rs, b_cs;      // element (col j, contraction c) of B at B[j * b_rs + c * b_cs]
  long ldo;
  int rows, cols, kc;   // problem size: rows x cols outputs, contraction length kc (per split for wgrad)
  int kc_total;         // wgrad: full contraction length (tokens); splits cut it into kc-sized ranges
  long split_stride;    // wgrad / split-K: out + split * split_stride
  int epi;              // 0 none / bias, 1 GELU (dual output), 2 multiply by GELU'(aux)
  // ---- fused pieces of the Swin block (round 5); all zero = the plain product -------------------------------------------
  // LayerNorm prologue (forward kernels instantiated with LN = true): the A rows are normalised while they are staged,
  //   row statistics by a pre-pass of the workgroup over its rows (two passes: mean, then centred squares - what
  //   nnz_layer_norm_forward computes); the first column tile also writes mean / rstd / the normalised rows (ln_y, the x
  //   operand of the weight gradient and what the LayerNorm backward needs)
  const float* ln_gamma;
  const float* ln_beta;
  float* ln_mean;       // [rows]
  float* ln_rstd;       // [rows]
  float* ln_y;          // [rows][kc_full] or null
  float ln_eps;
  // pad gather (LN kernels): pad_h > 0 - A row r is token (b, y, x) of the top / left padded grid (pad_h + pad_y) x (pad_w + pad_x);
  //   it reads source token (b, y - pad_y, x - pad_x) of the [B][pad_h][pad_w] tensor, or zeros (SwinTransformerBlock's F.pad,
  //   swt2net.py:643-645, folded into the addressing)
  int pad_h, pad_w, pad_y, pad_x;
  // epilogue: out = res + s_b (acc + bias) [* GELU'(aux)]; s_b = dp_inv * floor(dp_keep + dp_rand[b]), b = row / dp_rps - the
  //   residual add and the reference's DropPath (swt2net.py:379-388) without launches of their own.  In the weight gradient
  //   dp_* scales the dy operand per token instead (the gradient of the dropped branch).
  const float* res;
  const float* dp_rand;
  float dp_keep, dp_inv;
  int dp_rps, dp_nb;
  int ksplit;           // forward / dgrad: > 1 - split-K, raw partials to out + split * split_stride, epilogue in the fold kernel
  int kc_full;          // forward / dgrad with ksplit > 1: the whole contraction length (kc = per split)
  // fp16 operands / outputs (the token Linears of the autocast nets that the fp16 token kernel does not take: activations and
  // their gradients stay fp16 in HBM - no cast launches around the product -, converted while they are staged; weights,
  // accumulators, weight gradients and split-K partials are fp32).  A / B / out then point at _Float16.
  int a_half, b_half, out_half;
};

// drop-path scale of sample b
__device__ __forceinline__ float d32_dp_scale(const D32Args& a, int b) {
  b = b < a.dp_nb ? b : a.dp_nb - 1;
  return a.dp_inv * floorf(a.dp_rand[b] + a.dp_keep);
}

// the epilogue of one output element (shared by the tile kernels and the split-K fold)
template <bool OUT_HALF = false>
__device__ __forceinline__ void d32_epilogue(const D32Args& a, float v, float bv, float sc, long o) {
  v = (v + bv) * sc;
  if (a.epi == 2) v *= gelu_grad_f(a.aux[o]);
  if (a.res) v += a.res[o];
  if constexpr (OUT_HALF) reinterpret_cast<_Float16*>(a.out)[o] = (_Float16)v;
  else a.out[o] = v;
  if (a.epi == 1) a.out2[o] = gelu_f(v);
}

// An operand tile of NR rows x D32_BK contraction steps.  Global loads are always 16 bytes along the axis that is
// contiguous in MEMORY, and the LDS image keeps that axis contiguous too (16-byte LDS writes, no scattered transposition):
//   CONTIG  (contraction contiguous: x / dy rows, W in the forward)   image [row][step], pitch 36: a lane's four steps of
//           its row are one ds_read_b128;
//   !CONTIG (row index contiguous: W in the input gradient, dy / x in the weight gradient)   image [step][row], pitch
//           NR + 4: a lane's four steps are four ds_read_b32 whose 32 lanes read consecutive words (conflict-free).
// LN (CONTIG tiles of the forward's A operand): rows come through the workgroup's row table (source token or -1 = a padded
// token, all zeros) and are normalised on their way into LDS with the row statistics of the pre-pass; a thread's pieces of
// one block share their column (256 threads / 16 pieces per row), so gamma / beta are ONE 16-byte load each per block.
struct D32Ln {
  const int* src_row;     // LDS [NR]: source row of tile row r, -1 = zeros
  const float* mean;      // LDS [NR]
  const float* rstd;      // LDS [NR]
};
template <int NR, bool CONTIG, int BK>
struct D32Tile {
  static constexpr int D32_BK = BK;
  static constexpr int D32_PITCH = BK + 4;                             // floats per row of the [row][step] image (16-byte aligned)
  static constexpr int TP = NR + 4;                                    // pitch of the [step][row] image
  static constexpr int FLOATS = CONTIG ? NR * D32_PITCH : D32_BK * TP;
  static constexpr int NV = NR * D32_BK / 4;                           // 16-byte pieces per tile
  static constexpr int LPT = (NV + 255) / 256;
  f32x4 reg[LPT];
  f32x4 gm, bt;                                                        // LN: gamma / beta of this thread's column piece
  // half: src points at _Float16 (same strides in elements); four values are one 8-byte load.  The raw halves ride in the
  // first two words of reg[i] and are converted in store(): a conversion here made every load of the prefetch wait for its data
  // on the spot (the step's loads ran one after the other: grouped weight gradients 2.3 -> 3.2 ms per M2Net pass)
  // (HALF is a template parameter: as a run-time flag the two forms of every load sat in branches of their own, the prefetch
  //  loads of a step no longer went out back to back, and the fp32 SwT2Net step lost 5.6 ms of 57.6)
  template <bool half = false>
  __device__ __forceinline__ void load(const float* src, long rs, long cs, int r0, int nrows, int c0, int nc, int tid) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const _Float16* srch = reinterpret_cast<const _Float16*>(src);
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int p = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < NV) {
        if (CONTIG) {
          const int r = p / (D32_BK / 4), c4 = (p % (D32_BK / 4)) * 4;
          if (r0 + r < nrows && c0 + c4 < nc) {
            const long o = (long)(r0 + r) * rs + (c0 + c4);
            if constexpr (half) {
              const f32x2_t raw = *reinterpret_cast<const f32x2_t*>(srch + o);
              v[0] = raw[0];
              v[1] = raw[1];
            } else {
              v = *reinterpret_cast<const f32x4*>(src + o);
            }
          }
        } else {
          const int c = p / (NR / 4), r4 = (p % (NR / 4)) * 4;         // consecutive lanes: consecutive rows of one step
          if (c0 + c < nc) {
            const long o = (long)(c0 + c) * cs + (r0 + r4);
            if (r0 + r4 + 3 < nrows) {
              if constexpr (half) {
                const f32x2_t raw = *reinterpret_cast<const f32x2_t*>(srch + o);
                v[0] = raw[0];
                v[1] = raw[1];
              } else {
                v = *reinterpret_cast<const f32x4*>(src + o);
              }
            } else if constexpr (half) {
              _Float16 hv[4] = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (r0 + r4 + e < nrows) hv[e] = srch[o + e];
              typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
              const f32x2_t raw = __builtin_bit_cast(f32x2_t, h16x4{hv[0], hv[1], hv[2], hv[3]});
              v[0] = raw[0];
              v[1] = raw[1];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (r0 + r4 + e < nrows) v[e] = src[o + e];
            }
          }
        }
      }
      reg[i] = v;
    }
  }
  // the fp32 values of piece i (raw halves of a half operand are converted here)
  template <bool half>
  __device__ __forceinline__ f32x4 piece(int i) const {
    if constexpr (!half) return reg[i];
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    const h16x4 hv = __builtin_bit_cast(h16x4, f32x2_t{reg[i][0], reg[i][1]});
    return f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
  }
  // weight gradient, dy operand of a drop-path branch: step (= token) c0 + c is scaled by its sample's factor
  __device__ __forceinline__ void scale_steps(const D32Args& a, int c0, int nc, int tid) {
    const int b0 = c0 / a.dp_rps;                                      // wave-uniform
    const float s0 = d32_dp_scale(a, b0);
    const int e1 = (b0 + 1) * a.dp_rps;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int p = tid + i * 256;
      if (p < NV) {
        const int c = c0 + p / (NR / 4);
        float sc = s0;
        if (c >= e1) sc = d32_dp_scale(a, c / a.dp_rps);               // rare: a block that crosses into the next sample(s)
        reg[i] = reg[i] * sc;
      }
    }
  }
  // LN tiles: rows through the row table; gamma / beta of the thread's column piece ride along
  __device__ __forceinline__ void load_ln(const D32Args& a, const D32Ln& ln, int r0, int c0, int nc, int tid) {
    static_assert(CONTIG, "LayerNorm prologue: contraction-contiguous rows only");
    const int c4 = (tid % (D32_BK / 4)) * 4;
    const bool cin = c0 + c4 < nc;
    gm = (cin && a.ln_gamma) ? *reinterpret_cast<const f32x4*>(a.ln_gamma + c0 + c4) : f32x4{cin ? 1.f : 0.f, cin ? 1.f : 0.f, cin ? 1.f : 0.f, cin ? 1.f : 0.f};
    bt = (cin && a.ln_beta) ? *reinterpret_cast<const f32x4*>(a.ln_beta + c0 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int p = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < NV) {
        const int r = p / (D32_BK / 4);
        const int sr = ln.src_row[r];
        if (sr >= 0 && cin) v = *reinterpret_cast<const f32x4*>(a.A + (long)sr * a.a_rs + (c0 + c4));
      }
      reg[i] = v;
    }
  }
  template <bool half = false>
  __device__ __forceinline__ void store(float* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int p = tid + i * 256;
      if (p < NV) {
        if (CONTIG) {
          const int r = p / (D32_BK / 4), c4 = (p % (D32_BK / 4)) * 4;
          *reinterpret_cast<f32x4*>(tile + r * D32_PITCH + c4) = piece<half>(i);
        } else {
          const int c = p / (NR / 4), r4 = (p % (NR / 4)) * 4;
          *reinterpret_cast<f32x4*>(tile + c * TP + r4) = piece<half>(i);
        }
      }
    }
  }
  // normalise, hand to LDS, and (first column tile) write the normalised rows out
  __device__ __forceinline__ void store_ln(const D32Args& a, const D32Ln& ln, float* tile, int r0, int c0, int nc, int tid,
                                           bool write_y) const {
    const int c4 = (tid % (D32_BK / 4)) * 4;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int p = tid + i * 256;
      if (p < NV) {
        const int r = p / (D32_BK / 4);
        const float mu = ln.mean[r], rs = ln.rstd[r];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (reg[i][e] - mu) * rs * gm[e] + bt[e];
        if (r0 + r >= a.rows) o = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(tile + r * D32_PITCH + c4) = o;
        if (write_y && r0 + r < a.rows && c0 + c4 < nc)
          *reinterpret_cast<f32x4*>(a.ln_y + (long)(r0 + r) * a.kc_full + (c0 + c4)) = o;
      }
    }
  }
  // the lane's four contraction steps kb * 8 + hh * 4 + {0..3} of row `row`
  static __device__ __forceinline__ f32x4 frag(const float* tile, int row, int kb, int hh) {
    if (CONTIG) return *reinterpret_cast<const f32x4*>(tile + row * D32_PITCH + kb * 8 + hh * 4);
    const float* p = tile + (kb * 8 + hh * 4) * TP + row;
    f32x4 v = {p[0], p[TP], p[2 * TP], p[3 * TP]};
    return v;
  }
};

// WM x WN MFMA tiles per wave; workgroup tile = (64 WM) rows x (64 WN) cols.  The contraction runs in blocks of BK steps
// with a prefetch distance of TWO blocks (two register sets, the loop unrolled by two): most calls of the Swin / ViT nets
// are small (a few hundred workgroups, 12-96 blocks each), so a block's global-load latency must hide behind the LDS
// hand-over and the MFMAs of two neighbouring blocks, not one.
// `split`: weight gradient - the token range; forward / input gradient with a.ksplit > 1 - the range of the contraction
// (split-K: skinny products of the deep Swin levels, 2 x 12 tiles of 64 x 64 over a contraction of 3 072, are otherwise one
// 48-block chain per workgroup on a tenth of the chip); the partials are folded in split order by dense32_splitk_fold_kernel.
template <int WM, int WN, bool A_CONTIG, bool B_CONTIG, bool WGRAD, bool LN, bool H16 = false>
__device__ __forceinline__ void dense32_body(const D32Args& a, const int tile, const int split) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int D32_BK = (WM == 1 && WN == 1) ? 64 : 32;
  using TA = D32Tile<BM, A_CONTIG, D32_BK>;
  using TB = D32Tile<BN, B_CONTIG, D32_BK>;
  __shared__ __attribute__((aligned(16))) float sA[TA::FLOATS];
  __shared__ __attribute__((aligned(16))) float sB[TB::FLOATS];
  __shared__ float sMean[LN ? BM : 1], sRstd[LN ? BM : 1];
  __shared__ int sSrc[LN ? BM : 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;           // wave grid 2 x 2
  const int ntc = (a.cols + BN - 1) / BN;
  const int r0 = (tile / ntc) * BM, c0 = (tile % ntc) * BN;
  int k_begin = 0, k_end = a.kc;
  const bool ksplit = !WGRAD && a.ksplit > 1;
  if (WGRAD) {
    k_begin = split * a.kc;
    k_end = k_begin + a.kc < a.kc_total ? k_begin + a.kc : a.kc_total;
  } else if (ksplit) {
    k_begin = split * a.kc;
    k_end = k_begin + a.kc < a.kc_full ? k_begin + a.kc : a.kc_full;
  }
  const bool wscale = WGRAD && a.dp_rand != nullptr;
  D32Ln ln = {sSrc, sMean, sRstd};

  if (LN) {
    // ---- pre-pass: source row and LayerNorm statistics of the tile's rows; TPR threads per row ------------------------
    constexpr int TPR = 256 / BM;
    const int r = tid / TPR, part = tid % TPR;
    const int row = r0 + r;
    int sr = -1;
    if (row < a.rows) {
      sr = row;
      if (a.pad_h > 0) {
        const int hp = a.pad_h + a.pad_y, wp = a.pad_w + a.pad_x;
        const int b = row / (hp * wp), rem = row - b * (hp * wp);
        const int y = rem / wp, x = rem - y * wp;
        sr = (y >= a.pad_y && x >= a.pad_x) ? (b * a.pad_h + (y - a.pad_y)) * a.pad_w + (x - a.pad_x) : -1;
      }
    }
    const int K = a.kc_full;
    float mu = 0.f, rs = 0.f;
    if (sr >= 0) {
      const float* xr = a.A + (long)sr * a.a_rs;
      float s0 = 0.f;
      for (int c = 4 * part; c < K; c += 4 * TPR) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        s0 += (v[0] + v[1]) + (v[2] + v[3]);
      }
#pragma unroll
      for (int o = TPR / 2; o > 0; o >>= 1) s0 += __shfl_xor(s0, o, 64);
      mu = s0 / (float)K;
      float ss = 0.f;
      for (int c = 4 * part; c < K; c += 4 * TPR) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - mu;
          ss += d * d;
        }
      }
#pragma unroll
      for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      rs = 1.f / sqrtf(ss / (float)K + a.ln_eps);
    } else {
      rs = 1.f / sqrtf(a.ln_eps);                    // a zero row: mean 0, variance 0 - the normalised row is beta
    }
    if (part == 0) {
      sSrc[r] = sr;
      sMean[r] = mu;
      sRstd[r] = rs;
      if (c0 == 0 && split == 0 && row < a.rows) {
        a.ln_mean[row] = mu;
        a.ln_rstd[row] = rs;
      }
    }
    __syncthreads();
  }
  const bool write_y = LN && c0 == 0 && a.ln_y != nullptr;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float dbsum = 0.f;  // wgrad: this thread's row of the dy tile (tid < BM), summed over the contraction

  TA stA0, stA1;
  TB stB0, stB1;
  auto compute = [&]() {
    if (WGRAD && a.part_db && c0 == 0 && tid < BM) {
#pragma unroll
      for (int q = 0; q < D32_BK; ++q) dbsum += A_CONTIG ? sA[tid * TA::D32_PITCH + q] : sA[q * TA::TP + tid];
    }
#pragma unroll
    for (int kb = 0; kb < D32_BK / 8; ++kb) {
      f32x4 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[i] = TA::frag(sA, (wr * WM + i) * 32 + l31, kb, hh);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[j] = TB::frag(sB, (wc * WN + j) * 32 + l31, kb, hh);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = mfma_f32x(fa[i][s], fb[j][s], acc[i][j]);
    }
  };
  auto loadA = [&](TA& t, int k0) {
    if constexpr (LN) {
      t.load_ln(a, ln, r0, k0, k_end, tid);
    } else {
      t.template load<H16>(a.A, a.a_rs, a.a_cs, r0, a.rows, k0, k_end, tid);
      if constexpr (WGRAD && !A_CONTIG) {
        if (wscale) t.scale_steps(a, k0, k_end, tid);
      }
    }
  };
  auto storeA = [&](const TA& t, int k0) {
    if constexpr (LN) t.store_ln(a, ln, sA, r0, k0, k_end, tid, write_y);
    else t.template store<H16>(sA, tid);
  };
  loadA(stA0, k_begin);
  stB0.template load<H16 && WGRAD>(a.B, a.b_rs, a.b_cs, c0, a.cols, k_begin, k_end, tid);
  if (k_begin + D32_BK < k_end) {
    loadA(stA1, k_begin + D32_BK);
    stB1.template load<H16 && WGRAD>(a.B, a.b_rs, a.b_cs, c0, a.cols, k_begin + D32_BK, k_end, tid);
  }
  for (int k0 = k_begin; k0 < k_end; k0 += 2 * D32_BK) {
    __syncthreads();
    storeA(stA0, k0);
    stB0.template store<H16 && WGRAD>(sB, tid);
    __syncthreads();
    if (k0 + 2 * D32_BK < k_end) {
      loadA(stA0, k0 + 2 * D32_BK);
      stB0.template load<H16 && WGRAD>(a.B, a.b_rs, a.b_cs, c0, a.cols, k0 + 2 * D32_BK, k_end, tid);
    }
    compute();
    if (k0 + D32_BK < k_end) {
      __syncthreads();
      storeA(stA1, k0 + D32_BK);
      stB1.template store<H16 && WGRAD>(sB, tid);
      __syncthreads();
      if (k0 + 3 * D32_BK < k_end) {
        loadA(stA1, k0 + 3 * D32_BK);
        stB1.template load<H16 && WGRAD>(a.B, a.b_rs, a.b_cs, c0, a.cols, k0 + 3 * D32_BK, k_end, tid);
      }
      compute();
    }
  }

  // ---- epilogue: acc[i][j][r] = output (row = r0 + (wr WM + i) 32 + (r&3) + 8 (r>>2) + 4 hh, col = c0 + (wc WN + j) 32 + l31)
  if (WGRAD || ksplit) {     // raw partial (or the whole weight gradient): no bias, no activation
    float* out = a.out + (long)split * a.split_stride;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = c0 + (wc * WN + j) * 32 + l31;
      if (col >= a.cols) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + (wr * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (row < a.rows) out[(long)row * a.ldo + col] = acc[i][j][r];
        }
    }
    if (WGRAD && a.part_db && c0 == 0 && tid < BM && r0 + tid < a.rows) a.part_db[(long)split * a.rows + r0 + tid] = dbsum;
    return;
  }
  // drop-path scale per row: the tile's rows span at most three samples when a sample has >= 64 rows (b0, b0 + 1, b0 + 2)
  float s0 = 1.f, s1 = 1.f, s2 = 1.f;
  int e1 = 0x7fffffff, e2 = 0x7fffffff;
  const bool dp = a.dp_rand != nullptr;
  const bool dp_div = dp && a.dp_rps < 64;
  if (dp && !dp_div) {
    const int b0 = r0 / a.dp_rps;
    s0 = d32_dp_scale(a, b0);
    s1 = d32_dp_scale(a, b0 + 1);
    s2 = d32_dp_scale(a, b0 + 2);
    e1 = (b0 + 1) * a.dp_rps;
    e2 = e1 + a.dp_rps;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = c0 + (wc * WN + j) * 32 + l31;
    if (col >= a.cols) continue;
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + (wr * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row >= a.rows) continue;
        float sc = row >= e2 ? s2 : (row >= e1 ? s1 : s0);
        if (dp_div) sc = d32_dp_scale(a, row / a.dp_rps);
        d32_epilogue<H16 && !WGRAD>(a, acc[i][j][r], bv, sc, (long)row * a.ldo + col);
      }
  }
}

template <int WM, int WN, bool A_CONTIG, bool B_CONTIG, bool WGRAD, bool LN = false, bool H16 = false>
__global__ __launch_bounds__(256, 2) void dense32_kernel(D32Args a) {
  dense32_body<WM, WN, A_CONTIG, B_CONTIG, WGRAD, LN, H16>(a, blockIdx.x, blockIdx.y);
}

// split-K second stage: out[row][col] = epilogue(sum_s part[s][row][col]) in split order (bit-identical run to run)
__global__ __launch_bounds__(256) void dense32_splitk_fold_kernel(D32Args a, const float* __restrict__ part) {
  const long n4 = (long)a.rows * a.cols / 4;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < n4; p += (long)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(part + 4 * p);
    for (int q = 1; q < a.ksplit; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(part + (long)q * a.split_stride + 4 * p);
      v += t;
    }
    const long e = 4 * p;
    const int row = (int)(e / a.cols), col = (int)(e - (long)row * a.cols);
    const float sc = a.dp_rand ? d32_dp_scale(a, row / a.dp_rps) : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a.out_half) d32_epilogue<true>(a, v[i], a.bias ? a.bias[col + i] : 0.f, sc, (long)row * a.ldo + col + i);
      else d32_epilogue<false>(a, v[i], a.bias ? a.bias[col + i] : 0.f, sc, (long)row * a.ldo + col + i);
    }
  }
}

// ---- grouped weight gradients: ALL weight gradients of a backward pass in one launch ------------------------------------
// The Swin / ViT steps hold ~670 Linear layers whose weight gradients (T = 882 ... 35 378 tokens, K, N = 96 ... 3072) are
// 18-30 us launches of a few hundred workgroups each plus a fold launch: 21 ms of a 104 ms SwT2Net step, latency not MFMA
// rate.  They do not sit on the data-gradient chain, so the autograd nodes only queue (dy, x, dW, db) and the end of the
// backward pass runs every queued problem in ONE launch of 64 x 64 tiles (job table + workgroup -> job map in device
// memory) and ONE fold launch.  Same arithmetic and split rule as nnz_dense32_wgrad: bit-identical results.
struct D32Job {
  D32Args a;
  int wg_begin;   // first workgroup of this job in the grouped launch
  int ntiles;     // tiles of the weight matrix (64 x 64 or 128 x 128 by launch class); workgroup (wg - wg_begin) = split * ntiles + tile
};
struct D32FoldJob {
  const float* part;
  float* dst;
  const float* part_db;
  float* db;
  long n;
  int splits, nb;
  int blk_begin;  // first 256-thread block of this job in the grouped fold launch
  int pad;
};
// sum of `splits` partials of element i (stride n) in split order; eight independent loads in flight (LayerNorm fold jobs have
// up to 256 parts).  ONE function for every fold kernel, so that the grouped and the per-layer folds give the same bits.
__device__ __forceinline__ float d32_fold_sum(const float* __restrict__ part, long n, int splits, long i) {
#pragma clang fp reassociate(off)      // the sum runs in split order whatever -ffast-math would like to do with it
  float s = 0.f;
  int q = 0;
  for (; q + 8 <= splits; q += 8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = part[(long)(q + e) * n + i];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += t[e];
  }
  for (; q < splits; ++q) s += part[(long)q * n + i];
  return s;
}

template <int WM, int WN, bool H16 = false>
__global__ __launch_bounds__(256, 2) void dense32_group_wgrad_kernel(const D32Job* __restrict__ jobs,
                                                                     const int* __restrict__ wg_job) {
  const int j = __builtin_amdgcn_readfirstlane(wg_job[blockIdx.x]);
  const D32Job* jp = jobs + j;
  const D32Args a = jp->a;
  const int local = (int)blockIdx.x - jp->wg_begin;
  const int nt = jp->ntiles;
  dense32_body<WM, WN, false, false, true, false, H16>(a, local % nt, local / nt);
}
__global__ __launch_bounds__(256) void dense32_group_fold_kernel(const D32FoldJob* __restrict__ jobs,
                                                                 const int* __restrict__ blk_job) {
  const int j = __builtin_amdgcn_readfirstlane(blk_job[blockIdx.x]);
  const D32FoldJob f = jobs[j];
  const long i = (long)((int)blockIdx.x - f.blk_begin) * 256 + threadIdx.x;
  if (i < f.n) {
    f.dst[i] = d32_fold_sum(f.part, f.n, f.splits, i);
  } else if (f.db && i - f.n < f.nb) {
    const long jj = i - f.n;
    f.db[jj] = d32_fold_sum(f.part_db, f.nb, f.splits, jj);
  }
}

// fold the weight-gradient partials in split order: dW[e] = sum_s part[s][e], db[n] = sum_s part_db[s][n]
__global__ __launch_bounds__(256) void dense32_fold_kernel(const float* __restrict__ part, int splits, long n,
                                                           float* __restrict__ dst, const float* __restrict__ part_db,
                                                           int nb, float* __restrict__ db) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    dst[i] = d32_fold_sum(part, n, splits, i);
  } else if (db && i - n < nb) {
    const long j = i - n;
    db[j] = d32_fold_sum(part_db, nb, splits, j);
  }
}

template <bool AC, bool BC, bool WG, bool LN = false, bool H16 = false>
static int d32_launch(const D32Args& a, int splits, hipStream_t s) {
  // tile choice: the largest tile that still gives the chip ~2 workgroups per CU
  auto wgs = [&](int bm, int bn) { return (long)((a.rows + bm - 1) / bm) * ((a.cols + bn - 1) / bn) * splits; };
  if (a.cols > 64 && a.rows > 64 && wgs(128, 128) >= 256) {
    NNZ_LAUNCH((dense32_kernel<2, 2, AC, BC, WG, LN, H16>), dim3((unsigned)wgs(128, 128) / splits, splits), dim3(256), 0, s, a);
  } else if (a.cols > 64 && wgs(64, 128) >= 192) {
    NNZ_LAUNCH((dense32_kernel<1, 2, AC, BC, WG, LN, H16>), dim3((unsigned)wgs(64, 128) / splits, splits), dim3(256), 0, s, a);
  } else {
    NNZ_LAUNCH((dense32_kernel<1, 1, AC, BC, WG, LN, H16>), dim3((unsigned)wgs(64, 64) / splits, splits), dim3(256), 0, s, a);
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// split-K of the forward / input gradient: only products whose 64 x 64 tiling leaves most of the chip idle (< 192 workgroups)
// and whose contraction is long enough for the cut to pay for its fold launch (>= 768: a 64-step block costs a workgroup ~1.4 us,
// the fold ~5 us - profiles/r05_swin_ops_before.txt; with the first threshold of 256 the 384 ... 512-step products of the
// 16^2 / 32^2 levels were split for no gain and the SwT2Net step carried 613 fold launches); enough splits for ~384 workgroups,
// each at least 128 steps, at most 16.  per = steps per split (a multiple of the 64-step LDS block).  Returns the splits (1 = none).
static int d32_ksplits(long T, int K, int N, int* per_out) {
  const long tiles = ((T + 63) / 64) * ((N + 63) / 64);
  int splits = 1;
  static const int kmin = [] { const char* e = getenv("NNZ_D32_SPLITK_MIN"); return e && atoi(e) > 0 ? atoi(e) : 768; }();
  if (tiles < 192 && K >= kmin) {
    splits = (int)((384 + tiles - 1) / tiles);
    if (splits > K / 128) splits = K / 128;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
  }
  int per = (K + splits - 1) / splits;
  per = (per + 63) / 64 * 64;
  if (per_out) *per_out = per;
  return (K + per - 1) / per;
}

// one forward / input-gradient product with everything optional: `a` describes the whole problem (kc = the contraction)
template <bool BC, bool LN, bool H16 = false>
static int d32_run(D32Args a, float* workspace, hipStream_t s) {
  int per = 0;
  const int K = a.kc;
  a.kc_full = K;
  const int splits = workspace ? d32_ksplits(a.rows, K, a.cols, &per) : 1;
  if (splits <= 1) {
    a.ksplit = 1;
    return d32_launch<true, BC, false, LN, H16>(a, 1, s);
  }
  D32Args t = a;                         // the tile launch: raw partials [split][rows][cols] in the workspace
  t.ksplit = splits; t.kc = per;
  t.out = workspace; t.out_half = 0; t.ldo = a.cols; t.split_stride = (long)a.rows * a.cols;
  const long tiles = (long)((a.rows + 63) / 64) * ((a.cols + 63) / 64);
  NNZ_LAUNCH((dense32_kernel<1, 1, true, BC, false, LN, H16>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, s, t);
  D32Args f = a;                         // the fold: sums the partials in split order, then the epilogue
  f.ksplit = splits; f.split_stride = t.split_stride;
  const long n4 = (long)a.rows * a.cols / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(dense32_splitk_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, s, f, (const float*)workspace);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

// y[T][N] = x[T][K] W[N][K]^T + bias;  gelu != 0: y keeps the pre-activation, y_act = GELU(y) (exact erf form)
extern "C" int nnz_dense32_forward(const float* x, const float* W, const float* bias, float* y, float* y_act, long T, int K,
                                   int N, int gelu, void* stream) {
  using namespace nnz;
  if (!x || !W || !y || T < 1 || T > (1L << 30) || K < 4 || N < 1 || (K & 3) || (gelu && !y_act)) return NNZ_EINVAL;
  D32Args a = {};
  a.A = x; a.a_rs = K; a.a_cs = 1;
  a.B = W; a.b_rs = K; a.b_cs = 1;
  a.out = y; a.out2 = y_act; a.bias = bias; a.ldo = N;
  a.rows = (int)T; a.cols = N; a.kc = K; a.epi = gelu ? 1 : 0;
  return d32_launch<true, true, false>(a, 1, (hipStream_t)stream);
}

// dx[T][K] = dy[T][N] W[N][K]  (* GELU'(h[T][K]) when h is given: the gradient w.r.t. the pre-activation of the layer below)
extern "C" int nnz_dense32_dgrad(const float* dy, const float* W, const float* h, float* dx, long T, int K, int N,
                                 void* stream) {
  using namespace nnz;
  if (!dy || !W || !dx || T < 1 || T > (1L << 30) || K < 1 || N < 4 || (N & 3) || (K & 3)) return NNZ_EINVAL;
  D32Args a = {};
  a.A = dy; a.a_rs = N; a.a_cs = 1;
  a.B = W; a.b_rs = 1; a.b_cs = K;      // column j = input feature k: W[n][k], contraction n strided by K, rows contiguous
  a.out = dx; a.aux = h; a.ldo = K;
  a.rows = (int)T; a.cols = K; a.kc = N; a.epi = h ? 2 : 0;
  return d32_launch<true, false, false>(a, 1, (hipStream_t)stream);
}

// ---- the same two products with the fused pieces of a Swin block (round 5) and split-K for skinny shapes ------------------
// workspace: nnz_dense32_splitk_workspace_floats(T, K, N) floats (contraction K, N output columns), or NULL = never split.
extern "C" long nnz_dense32_splitk_workspace_floats(long T, int contraction, int out_cols) {
  if (T < 1 || contraction < 1 || out_cols < 1) return 0;
  const int splits = nnz::d32_ksplits(T, contraction, out_cols, nullptr);
  return splits > 1 ? (long)splits * T * out_cols : 0;
}

// y = epilogue(LN?(x) W^T + bias).  T rows of the OUTPUT (with the pad gather: tokens of the padded grid).
//   ln_gamma != NULL or ln_mean != NULL: LayerNorm prologue over K (eps ln_eps): writes ln_mean / ln_rstd [T] and, when ln_y is
//     given, the normalised rows [T][K]; pad_h > 0: x is [B][pad_h][pad_w][K], row r is a token of the top / left padded grid
//     (pad_h + pad_y) x (pad_w + pad_x) - padded tokens are zero rows (their normalised value is beta);
//   res != NULL: y = res + s (x W^T + bias), res laid out like y;  dp_rand != NULL: s = floor(dp_keep + dp_rand[row / dp_rps])
//     / dp_keep (dp_nb samples);  gelu as in nnz_dense32_forward.
extern "C" int nnz_dense32_forward_fused(const float* x, const float* W, const float* bias, float* y, float* y_act, long T,
                                         int K, int N, int gelu, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                         float* ln_mean, float* ln_rstd, float* ln_y, int pad_h, int pad_w, int pad_y,
                                         int pad_x, const float* res, const float* dp_rand, float dp_keep, int dp_rps,
                                         int dp_nb, float* workspace, void* stream) {
  using namespace nnz;
  if (!x || !W || !y || T < 1 || T > (1L << 30) || K < 4 || N < 4 || (K & 3) || (N & 3) || (gelu && !y_act)) return NNZ_EINVAL;
  const bool ln = ln_mean != nullptr;
  if (ln && !ln_rstd) return NNZ_EINVAL;
  if (!ln && (pad_h > 0 || ln_gamma || ln_beta || ln_y)) return NNZ_EINVAL;
  if (pad_h > 0 && (pad_w < 1 || pad_y < 0 || pad_x < 0 || T % ((long)(pad_h + pad_y) * (pad_w + pad_x)))) return NNZ_EINVAL;
  if (dp_rand && (!(dp_keep > 0.f) || dp_rps < 1 || dp_nb < 1)) return NNZ_EINVAL;
  D32Args a = {};
  a.A = x; a.a_rs = K; a.a_cs = 1;
  a.B = W; a.b_rs = K; a.b_cs = 1;
  a.out = y; a.out2 = y_act; a.bias = bias; a.ldo = N;
  a.rows = (int)T; a.cols = N; a.kc = K; a.epi = gelu ? 1 : 0;
  a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.ln_eps = ln_eps; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_y = ln_y;
  a.pad_h = pad_h; a.pad_w = pad_w; a.pad_y = pad_y; a.pad_x = pad_x;
  a.res = res; a.dp_rand = dp_rand; a.dp_keep = dp_keep; a.dp_inv = dp_rand ? 1.f / dp_keep : 1.f; a.dp_rps = dp_rps;
  a.dp_nb = dp_nb;
  return ln ? d32_run<true, true>(a, workspace, (hipStream_t)stream) : d32_run<true, false>(a, workspace, (hipStream_t)stream);
}

// dx = s (dy W) [* GELU'(h)]: the input gradient of a branch that was added through DropPath (s as above, NULL = 1)
extern "C" int nnz_dense32_dgrad_fused(const float* dy, const float* W, const float* h, float* dx, long T, int K, int N,
                                       const float* dp_rand, float dp_keep, int dp_rps, int dp_nb, float* workspace,
                                       void* stream) {
  using namespace nnz;
  if (!dy || !W || !dx || T < 1 || T > (1L << 30) || K < 4 || N < 4 || (N & 3) || (K & 3)) return NNZ_EINVAL;
  if (dp_rand && (!(dp_keep > 0.f) || dp_rps < 1 || dp_nb < 1)) return NNZ_EINVAL;
  D32Args a = {};
  a.A = dy; a.a_rs = N; a.a_cs = 1;
  a.B = W; a.b_rs = 1; a.b_cs = K;
  a.out = dx; a.aux = h; a.ldo = K;
  a.rows = (int)T; a.cols = K; a.kc = N; a.epi = h ? 2 : 0;
  a.dp_rand = dp_rand; a.dp_keep = dp_keep; a.dp_inv = dp_rand ? 1.f / dp_keep : 1.f; a.dp_rps = dp_rps; a.dp_nb = dp_nb;
  return d32_run<false, false>(a, workspace, (hipStream_t)stream);
}

// The same two products for fp16 activations (x, y, dy, dx are _Float16; W, bias fp32; fp32 accumulation, split-K partials
// fp32): y = fp16(x W^T + bias), dx = fp16(dy W).  What torch.autocast's F.linear computes with an fp16 copy of W, at fp32
// weight precision and without the casts' launches.
extern "C" int nnz_dense32_forward_h16(const void* x, const float* W, const float* bias, void* y, long T, int K, int N,
                                       float* workspace, void* stream) {
  using namespace nnz;
  if (!x || !W || !y || T < 1 || T > (1L << 30) || K < 4 || N < 4 || (K & 3) || (N & 3)) return NNZ_EINVAL;
  D32Args a = {};
  a.A = reinterpret_cast<const float*>(x); a.a_rs = K; a.a_cs = 1; a.a_half = 1;
  a.B = W; a.b_rs = K; a.b_cs = 1;
  a.out = reinterpret_cast<float*>(y); a.out_half = 1; a.bias = bias; a.ldo = N;
  a.rows = (int)T; a.cols = N; a.kc = K; a.epi = 0;
  a.dp_inv = 1.f; a.dp_keep = 1.f; a.dp_rps = 1; a.dp_nb = 1;
  return d32_run<true, false, true>(a, workspace, (hipStream_t)stream);
}
extern "C" int nnz_dense32_dgrad_h16(const void* dy, const float* W, void* dx, long T, int K, int N, float* workspace,
                                     void* stream) {
  using namespace nnz;
  if (!dy || !W || !dx || T < 1 || T > (1L << 30) || K < 4 || N < 4 || (N & 3) || (K & 3)) return NNZ_EINVAL;
  D32Args a = {};
  a.A = reinterpret_cast<const float*>(dy); a.a_rs = N; a.a_cs = 1; a.a_half = 1;
  a.B = W; a.b_rs = 1; a.b_cs = K;
  a.out = reinterpret_cast<float*>(dx); a.out_half = 1; a.ldo = K;
  a.rows = (int)T; a.cols = K; a.kc = N; a.epi = 0;
  a.dp_inv = 1.f; a.dp_keep = 1.f; a.dp_rps = 1; a.dp_nb = 1;
  return d32_run<false, false, true>(a, workspace, (hipStream_t)stream);
}

// token splits of the weight gradient: none when the weight matrix alone gives >= 256 tiles of 64 x 64, otherwise enough
// to reach ~768 workgroups, at least 64 tokens each, at most 16 splits; `per` = tokens per split (a multiple of the contraction block)
static long d32_wgrad_splits(long T, int K, int N, long* per_out) {
  const long tiles = (long)((N + 63) / 64) * ((K + 63) / 64);
  long splits = tiles >= 256 ? 1 : (768 + tiles - 1) / tiles;
  const long max_by_tokens = (T + 63) / 64;
  if (splits > max_by_tokens) splits = max_by_tokens;
  if (splits > 16) splits = 16;
  if (splits < 1) splits = 1;
  long per = (T + splits - 1) / splits;
  per = (per + 63) / 64 * 64;   // a multiple of every tile's contraction block
  if (per_out) *per_out = per;
  return (T + per - 1) / per;
}

extern "C" long nnz_dense32_wgrad_workspace_floats(long T, int K, int N) {
  if (T < 1 || K < 1 || N < 1) return 0;
  const long splits = d32_wgrad_splits(T, K, N, nullptr);
  return splits > 1 ? splits * ((long)N * K + N) : 1;
}

// dW[N][K] = sum_t dy[t][n] x[t][k];  db[n] = sum_t dy[t][n] (db may be null).  Deterministic: token splits write partials
// into `workspace` (nnz_dense32_wgrad_workspace_floats), folded in split order.
extern "C" int nnz_dense32_wgrad(const float* dy, const float* x, float* dW, float* db, float* workspace, long T, int K,
                                 int N, void* stream) {
  using namespace nnz;
  if (!dy || !x || !dW || !workspace || T < 1 || T > (1L << 30) || (K & 3) || (N & 3) || K < 4 || N < 4) return NNZ_EINVAL;
  long per = 0;
  const long splits = d32_wgrad_splits(T, K, N, &per);
  D32Args a = {};
  a.A = dy; a.a_rs = 1; a.a_cs = N;     // row = n, contraction = token t (stride N)
  a.B = x; a.b_rs = 1; a.b_cs = K;      // col = k, contraction = token t (stride K)
  a.ldo = K; a.rows = N; a.cols = K; a.kc = (int)per; a.kc_total = (int)T;
  hipStream_t s = (hipStream_t)stream;
  if (splits == 1) {
    a.out = dW; a.split_stride = 0; a.part_db = db;
    return d32_launch<false, false, true>(a, 1, s);
  }
  a.out = workspace; a.split_stride = (long)N * K;
  a.part_db = db ? workspace + splits * (long)N * K : nullptr;
  const int rc = d32_launch<false, false, true>(a, (int)splits, s);
  if (rc != NNZ_OK) return rc;
  const long n = (long)N * K;
  const long total = n + (db ? N : 0);
  NNZ_LAUNCH(dense32_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)workspace,
             (int)splits, n, dW, (const float*)a.part_db, N, db);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// ---- grouped weight gradients (see dense32_group_wgrad_kernel) -------------------------------------------------------------
// Host protocol: for every queued problem nnz_dense32_group_plan gives its workgroups, fold blocks (0: written directly) and
// workspace floats; the caller lays the jobs out back to back (wg_begin / blk_begin = running sums), fills one record per job
// with nnz_dense32_group_fill into HOST tables of nnz_dense32_group_record_bytes(0 / 1) bytes per record, builds the two
// int32 maps (workgroup -> job, fold block -> fold job), copies the four arrays to the device and calls
// nnz_dense32_group_launch.
extern "C" int nnz_dense32_group_record_bytes(int which) {
  return which == 0 ? (int)sizeof(nnz::D32Job) : (int)sizeof(nnz::D32FoldJob);
}
// launch class of a job: 1 = 128 x 128 tiles (weight matrices that still give >= 256 workgroups with them), 0 = 64 x 64; the
// token splits do not depend on it, so the results are bit-identical either way
static int d32_group_class(long T, int K, int N) {
  const long splits = d32_wgrad_splits(T, K, N, nullptr);
  return (N > 64 && K > 64 && (long)((N + 127) / 128) * ((K + 127) / 128) * splits >= 256) ? 1 : 0;
}
extern "C" int nnz_dense32_group_class(long T, int K, int N) {
  if (T < 1 || K < 4 || N < 4) return NNZ_EINVAL;
  return d32_group_class(T, K, N);
}
extern "C" int nnz_dense32_group_plan(long T, int K, int N, int* wgs, int* fold_blocks, long* ws_floats) {
  if (T < 1 || T > (1L << 30) || (K & 3) || (N & 3) || K < 4 || N < 4 || !wgs || !fold_blocks || !ws_floats) return NNZ_EINVAL;
  const long splits = d32_wgrad_splits(T, K, N, nullptr);
  const int tsz = d32_group_class(T, K, N) ? 128 : 64;
  const long tiles = (long)((N + tsz - 1) / tsz) * ((K + tsz - 1) / tsz);
  *wgs = (int)(tiles * splits);
  *fold_blocks = splits > 1 ? (int)(((long)N * K + N + 255) / 256) : 0;
  *ws_floats = splits > 1 ? splits * ((long)N * K + N) : 0;
  return NNZ_OK;
}
extern "C" int nnz_dense32_group_fill(void* job_host, void* fold_host, const float* dy, const float* x, float* dW, float* db,
                                      float* workspace, long T, int K, int N, int wg_begin, int blk_begin) {
  using namespace nnz;
  if (!job_host || !dy || !x || !dW || T < 1 || T > (1L << 30) || (K & 3) || (N & 3) || K < 4 || N < 4) return NNZ_EINVAL;
  long per = 0;
  const long splits = d32_wgrad_splits(T, K, N, &per);
  if (splits > 1 && (!workspace || !fold_host)) return NNZ_EINVAL;
  D32Job j = {};
  D32Args& a = j.a;
  a.A = dy; a.a_rs = 1; a.a_cs = N;
  a.B = x; a.b_rs = 1; a.b_cs = K;
  a.ldo = K; a.rows = N; a.cols = K; a.kc = (int)per; a.kc_total = (int)T;
  j.wg_begin = wg_begin;
  const int tsz = d32_group_class(T, K, N) ? 128 : 64;
  j.ntiles = ((N + tsz - 1) / tsz) * ((K + tsz - 1) / tsz);
  if (splits == 1) {
    a.out = dW; a.split_stride = 0; a.part_db = db;
  } else {
    a.out = workspace; a.split_stride = (long)N * K;
    a.part_db = db ? workspace + splits * (long)N * K : nullptr;
    D32FoldJob f = {};
    f.part = workspace; f.dst = dW; f.part_db = a.part_db; f.db = db;
    f.n = (long)N * K; f.splits = (int)splits; f.nb = N; f.blk_begin = blk_begin;
    *reinterpret_cast<D32FoldJob*>(fold_host) = f;
  }
  *reinterpret_cast<D32Job*>(job_host) = j;
  return NNZ_OK;
}
// the same record for a branch that was added through DropPath: the dy operand is scaled per token by
// s = floor(dp_keep + dp_rand[token / dp_rps]) / dp_keep while it is staged (dp_nb samples) - dW = sum_t s_t dy[t]^T x[t], and the
// bias gradient (the column sums of the scaled tile) likewise
extern "C" int nnz_dense32_group_fill_scaled(void* job_host, void* fold_host, const float* dy, const float* x, float* dW,
                                             float* db, float* workspace, long T, int K, int N, int wg_begin, int blk_begin,
                                             const float* dp_rand, float dp_keep, int dp_rps, int dp_nb) {
  using namespace nnz;
  if (dp_rand && (!(dp_keep > 0.f) || dp_rps < 1 || dp_nb < 1)) return NNZ_EINVAL;
  const int rc = nnz_dense32_group_fill(job_host, fold_host, dy, x, dW, db, workspace, T, K, N, wg_begin, blk_begin);
  if (rc != NNZ_OK) return rc;
  D32Job* j = reinterpret_cast<D32Job*>(job_host);
  j->a.dp_rand = dp_rand; j->a.dp_keep = dp_keep; j->a.dp_inv = dp_rand ? 1.f / dp_keep : 1.f; j->a.dp_rps = dp_rps;
  j->a.dp_nb = dp_nb;
  return NNZ_OK;
}
// the same record with dy and x as _Float16 (dW, db, partials fp32)
extern "C" int nnz_dense32_group_fill_h16(void* job_host, void* fold_host, const void* dy, const void* x, float* dW, float* db,
                                          float* workspace, long T, int K, int N, int wg_begin, int blk_begin) {
  using namespace nnz;
  const int rc = nnz_dense32_group_fill(job_host, fold_host, reinterpret_cast<const float*>(dy),
                                        reinterpret_cast<const float*>(x), dW, db, workspace, T, K, N, wg_begin, blk_begin);
  if (rc != NNZ_OK) return rc;
  D32Job* j = reinterpret_cast<D32Job*>(job_host);
  j->a.a_half = 1;
  j->a.b_half = 1;
  return NNZ_OK;
}
// a fold-only record (no product): dst[i] = sum_{q < parts} part[q * n + i], i < n, in part order - the LayerNorm backward of the
// fused Swin block leaves its per-workgroup dgamma | dbeta partials ([parts][2 C]) to this launch instead of ending every
// backward launch with fixed-point adds and a last-workgroup pass
extern "C" int nnz_dense32_group_fill_fold(void* fold_host, const float* part, float* dst, long n, int parts, int blk_begin) {
  using namespace nnz;
  if (!fold_host || !part || !dst || n < 1 || parts < 1 || blk_begin < 0) return NNZ_EINVAL;
  D32FoldJob f = {};
  f.part = part; f.dst = dst; f.n = n; f.splits = parts; f.nb = 0; f.blk_begin = blk_begin;
  *reinterpret_cast<D32FoldJob*>(fold_host) = f;
  return NNZ_OK;
}
// all jobs of one launch must be of the same class (nnz_dense32_group_class)
extern "C" int nnz_dense32_group_launch(const void* jobs_dev, const int* wg_job_dev, int total_wgs, const void* fold_dev,
                                        const int* blk_job_dev, int total_blks, int tile_class, void* stream) {
  using namespace nnz;
  if (!jobs_dev || !wg_job_dev || total_wgs < 1 || total_blks < 0 || (total_blks > 0 && (!fold_dev || !blk_job_dev)))
    return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // tile_class bit 0: 128 x 128 tiles; bit 1: the jobs' dy / x are _Float16 (nnz_dense32_group_fill_h16) - one launch holds jobs
  // of one kind (the operand type is a template parameter of the kernel: see D32Tile::load)
  if (tile_class == 3) {
    NNZ_LAUNCH((dense32_group_wgrad_kernel<2, 2, true>), dim3((unsigned)total_wgs), dim3(256), 0, s, (const D32Job*)jobs_dev,
               wg_job_dev);
  } else if (tile_class == 2) {
    NNZ_LAUNCH((dense32_group_wgrad_kernel<1, 1, true>), dim3((unsigned)total_wgs), dim3(256), 0, s, (const D32Job*)jobs_dev,
               wg_job_dev);
  } else if (tile_class == 1) {
    NNZ_LAUNCH((dense32_group_wgrad_kernel<2, 2>), dim3((unsigned)total_wgs), dim3(256), 0, s, (const D32Job*)jobs_dev,
               wg_job_dev);
  } else {
    NNZ_LAUNCH((dense32_group_wgrad_kernel<1, 1>), dim3((unsigned)total_wgs), dim3(256), 0, s, (const D32Job*)jobs_dev,
               wg_job_dev);
  }
  if (total_blks > 0)
    NNZ_LAUNCH(dense32_group_fold_kernel, dim3((unsigned)total_blks), dim3(256), 0, s, (const D32FoldJob*)fold_dev,
               blk_job_dev);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// fold records alone (nnz_dense32_group_fill_fold), for grouped launches of other kernel families (csrc/token_linear.hip)
extern "C" int nnz_group_fold_launch(const void* fold_dev, const int* blk_job_dev, int total_blks, void* stream) {
  using namespace nnz;
  if (!fold_dev || !blk_job_dev || total_blks < 1) return NNZ_EINVAL;
  NNZ_LAUNCH(dense32_group_fold_kernel, dim3((unsigned)total_blks), dim3(256), 0, (hipStream_t)stream,
             (const D32FoldJob*)fold_dev, blk_job_dev);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
