// Swin window attention core on MFMA (gfx950), fp32, forward + backward.
// Replaces, inside WindowAttention.forward of the reference (/root/reference/nnunetv2/nets/swt2net.py:584-619,
// byte-identical copy nets/swt.py:346-383), everything between the qkv Linear and the proj Linear:
//   cyclic roll by -window/2 (shifted blocks) -> 7x7 window partition -> split heads ->
//   softmax(q*scale @ k^T + rel_pos_bias[h] (+ -100 region mask, :559-582)) @ v -> merge heads -> un-partition -> roll back.
// Because the qkv / proj Linears act per token they commute with the token permutations, so the kernel reads q, k, v
// straight out of the (B, H, W, 3C) qkv tensor with the roll + partition folded into its index math and writes the
// merged-head result to the (B, H, W, C) position it came from: the four rearrange/roll copies of the reference
// (each a full read + write of the activation) never exist.  The shift mask is computed from region ids on the fly.
//
// One wave per (window, head).  v_mfma_f32_32x32x2_f32 (exact fp32, matching the reference's fp32 trainer
// nnUNetTrainerSwT2Net.train_step, no autocast).  The score tile is computed TRANSPOSED (keys on rows, queries on
// lanes): a query's 64 (49 + pad) scores then live in two lanes' registers, so softmax needs one lane exchange
// instead of a 32-lane reduction, and P^T feeds the PV MFMA as its B operand with no data movement
// (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand", T12).
#include "common.hpp"

namespace nnz {

constexpr int WA_L = 49;   // tokens per window
constexpr int WA_WS = 7;
constexpr int WA_LD = 33;  // LDS row stride (floats): head_dim <= 32, +1 pad -> conflict-free column reads

struct AttnArgs {
  const float* qkv;    // [B][H][W][3C]
  const float* bias;   // relative_position_bias_table [(2*7-1)^2][heads]
  const int* bidx;     // relative_position_index [49][49] (int32)
  float* out;          // [B][H][W][C]
  const float* dout;   // [B][H][W][C]
  float* dqkv;         // [B][H][W][3C]
  float* dbias;        // gradient of the table [169][heads] (atomic, zeroed by launcher)
  int B, H, W, C, heads, hd, shift;
  float scale;
};

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

struct TokenMap {
  long base;   // element offset of the token's (b, y, x) position, in units of "C" (multiply by row length)
  int region;  // shift-mask region id (0 when not shifted)
};

__device__ __forceinline__ TokenMap token_map(const AttnArgs& a, int win, int l) {
  const int nww = a.W / WA_WS, nwh = a.H / WA_WS;
  const int b = win / (nwh * nww);
  const int wi = (win / nww) % nwh, wj = win % nww;
  const int r = wi * WA_WS + l / WA_WS, c = wj * WA_WS + l % WA_WS;  // coordinates in the rolled image
  int y = r + a.shift, x = c + a.shift;
  if (y >= a.H) y -= a.H;
  if (x >= a.W) x -= a.W;
  TokenMap m;
  m.base = ((long)b * a.H + y) * a.W + x;
  m.region = 0;
  if (a.shift > 0) {
    const int hid = r < a.H - WA_WS ? 0 : (r < a.H - a.shift ? 1 : 2);
    const int wid = c < a.W - WA_WS ? 0 : (c < a.W - a.shift ? 1 : 2);
    m.region = 3 * hid + wid;
  }
  return m;
}

// stage one [49][hd] operand (rows >= 49 and columns >= hd zero-filled) into an LDS image [64][WA_LD]
__device__ __forceinline__ void stage(const AttnArgs& a, const float* src, int row_len, int ch0, int win, float mul,
                                      float* dst, int lane) {
  for (int i = lane; i < 64 * WA_LD; i += 64) dst[i] = 0.f;
  __syncthreads();
  if ((a.hd & 3) == 0) {
    const int q4 = a.hd >> 2;
    for (int i = lane; i < WA_L * q4; i += 64) {
      const int l = i / q4, c4 = (i % q4) * 4;
      const TokenMap m = token_map(a, win, l);
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + m.base * row_len + ch0 + c4);
      float* d = dst + l * WA_LD + c4;
      d[0] = v[0] * mul; d[1] = v[1] * mul; d[2] = v[2] * mul; d[3] = v[3] * mul;
    }
  } else {  // head_dim not a multiple of 4: scalar staging
    for (int i = lane; i < WA_L * a.hd; i += 64) {
      const int l = i / a.hd, c = i % a.hd;
      dst[l * WA_LD + c] = src[token_map(a, win, l).base * row_len + ch0 + c] * mul;
    }
  }
}

// store the 16 accumulator rows of one lane (channels 8*g4 + 4*hh + e) of a [channel][token] tile to dst[channel]
__device__ __forceinline__ void store_cols(float* dst, const f32x16& o, int hh, int hd, float mul) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int c0 = 8 * g4 + 4 * hh;
    if ((hd & 3) == 0) {
      if (c0 < hd) {
        f32x4 v = {o[4 * g4] * mul, o[4 * g4 + 1] * mul, o[4 * g4 + 2] * mul, o[4 * g4 + 3] * mul};
        *reinterpret_cast<f32x4*>(dst + c0) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < hd) dst[c0 + e] = o[4 * g4 + e] * mul;
    }
  }
}

// region id of every token of the window (0 when not shifted), computed once per workgroup
__device__ __forceinline__ void stage_regions(const AttnArgs& a, int win, int* sreg, int lane) {
  sreg[lane] = (a.shift > 0 && lane < WA_L) ? token_map(a, win, lane).region : 0;
}

// S^T = K Q^T (+ bias^T + mask), softmax over keys per query.  On return p[tk][tq][r] = P[query][key] for
// query = tq*32 + (lane&31), key = tk*32 + crow(r, lane>>5); also returns nothing else (m/l are folded in).
__device__ __forceinline__ void scores_T(const AttnArgs& a, const float* sq, const float* sk, const int* sreg,
                                         int head, int lane, f32x16 (&p)[2][2]) {
  const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[x][y][r] = 0.f;
  for (int kk = 0; kk < a.hd / 2; ++kk) {
    const int c = 2 * kk + hh;
    const float k0 = sk[l31 * WA_LD + c], k1 = sk[(32 + l31) * WA_LD + c];
    const float q0 = sq[l31 * WA_LD + c], q1 = sq[(32 + l31) * WA_LD + c];
    p[0][0] = mfma_f32(k0, q0, p[0][0]);
    p[0][1] = mfma_f32(k0, q1, p[0][1]);
    p[1][0] = mfma_f32(k1, q0, p[1][0]);
    p[1][1] = mfma_f32(k1, q1, p[1][1]);
  }
  const float* bias = a.bias + head;
#pragma unroll
  for (int tq = 0; tq < 2; ++tq) {
    const int i = tq * 32 + l31;
    const int ic = i < WA_L ? i : WA_L - 1;
    const int reg_i = sreg[ic];
    float m = -3.0e38f;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tk * 32 + crow(r, hh);
        float s = -3.0e38f;
        if (j < WA_L) {
          s = p[tk][tq][r] + bias[a.bidx[ic * WA_L + j] * a.heads];
          if (sreg[j] != reg_i) s += -100.f;
        }
        p[tk][tq][r] = s;
        m = fmaxf(m, s);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tk * 32 + crow(r, hh);
        const float e = j < WA_L ? __expf(p[tk][tq][r] - m) : 0.f;
        p[tk][tq][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[tk][tq][r] *= inv;
  }
}

__global__ __launch_bounds__(64) void win_attn_fwd_kernel(AttnArgs a) {
  __shared__ float sq[64 * WA_LD], sk[64 * WA_LD], sv[64 * WA_LD];
  __shared__ int sreg[64];
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hh = lane >> 5;
  const int win = blockIdx.x, head = blockIdx.y;
  const int C3 = 3 * a.C;
  stage_regions(a, win, sreg, lane);
  stage(a, a.qkv, C3, head * a.hd, win, a.scale, sq, lane);
  stage(a, a.qkv, C3, a.C + head * a.hd, win, 1.f, sk, lane);
  stage(a, a.qkv, C3, 2 * a.C + head * a.hd, win, 1.f, sv, lane);
  __syncthreads();
  f32x16 p[2][2];
  scores_T(a, sq, sk, sreg, head, lane, p);
  // O^T[p][query] = sum_key V[key][p] * P^T[key][query]
#pragma unroll
  for (int tq = 0; tq < 2; ++tq) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) o = mfma_f32(sv[(tk * 32 + crow(r, hh)) * WA_LD + l31], p[tk][tq][r], o);
    const int i = tq * 32 + l31;
    if (i < WA_L) {
      store_cols(a.out + token_map(a, win, i).base * a.C + head * a.hd, o, hh, a.hd, 1.f);
    }
  }
}

// backward.  Pass A (transposed orientation, queries on lanes): P^T, dP^T, delta, dS^T -> dQ.
//            Pass B (queries on rows): P, dP, dS -> dV, dK, dbias.  Row statistics cross from A to B through LDS.
__global__ __launch_bounds__(64) void win_attn_bwd_kernel(AttnArgs a) {
  __shared__ float sq[64 * WA_LD], sk[64 * WA_LD], sv[64 * WA_LD], sdo[64 * WA_LD];
  __shared__ float srow[3][64];  // per query: row max m, 1/sum, delta
  __shared__ int sreg[64];
  // gradient of the relative-position bias table: summed per (window, head) in LDS first (for a fixed query the 49 keys
  // hit 49 different table entries, so the LDS atomics of a step do not collide), then ONE global atomic per table entry.
  // The first version issued all 2401 atomics per (window, head) straight at the (169, heads) table: ~10 000 colliding
  // atomics per address made the backward 16x slower than the forward.
  constexpr int NBIAS = (2 * WA_WS - 1) * (2 * WA_WS - 1);
  __shared__ float sdb[NBIAS];
  const int lane = threadIdx.x;
  const int l31 = lane & 31, hh = lane >> 5;
  const int win = blockIdx.x, head = blockIdx.y;
  const int C3 = 3 * a.C;
  for (int i = lane; i < NBIAS; i += 64) sdb[i] = 0.f;
  stage_regions(a, win, sreg, lane);
  stage(a, a.qkv, C3, head * a.hd, win, a.scale, sq, lane);
  stage(a, a.qkv, C3, a.C + head * a.hd, win, 1.f, sk, lane);
  stage(a, a.qkv, C3, 2 * a.C + head * a.hd, win, 1.f, sv, lane);
  stage(a, a.dout, a.C, head * a.hd, win, 1.f, sdo, lane);
  __syncthreads();
  const float* bias = a.bias + head;

  // ---------------- pass A ------------------------------------------------------------------------------------
  {
    f32x16 s[2][2], dp[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[x][y][r] = 0.f;
          dp[x][y][r] = 0.f;
        }
    for (int kk = 0; kk < a.hd / 2; ++kk) {
      const int c = 2 * kk + hh;
      const float k0 = sk[l31 * WA_LD + c], k1 = sk[(32 + l31) * WA_LD + c];
      const float q0 = sq[l31 * WA_LD + c], q1 = sq[(32 + l31) * WA_LD + c];
      const float v0 = sv[l31 * WA_LD + c], v1 = sv[(32 + l31) * WA_LD + c];
      const float g0 = sdo[l31 * WA_LD + c], g1 = sdo[(32 + l31) * WA_LD + c];
      s[0][0] = mfma_f32(k0, q0, s[0][0]);
      s[0][1] = mfma_f32(k0, q1, s[0][1]);
      s[1][0] = mfma_f32(k1, q0, s[1][0]);
      s[1][1] = mfma_f32(k1, q1, s[1][1]);
      dp[0][0] = mfma_f32(v0, g0, dp[0][0]);  // dP^T[key][query] = sum_c V[key][c] dO[query][c]
      dp[0][1] = mfma_f32(v0, g1, dp[0][1]);
      dp[1][0] = mfma_f32(v1, g0, dp[1][0]);
      dp[1][1] = mfma_f32(v1, g1, dp[1][1]);
    }
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      const int i = tq * 32 + l31;
      const int ic = i < WA_L ? i : WA_L - 1;
      const int reg_i = sreg[ic];
      float m = -3.0e38f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = tk * 32 + crow(r, hh);
          float x = -3.0e38f;
          if (j < WA_L) {
            x = s[tk][tq][r] + bias[a.bidx[ic * WA_L + j] * a.heads];
            if (sreg[j] != reg_i) x += -100.f;
          }
          s[tk][tq][r] = x;
          m = fmaxf(m, x);
        }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = tk * 32 + crow(r, hh);
          const float e = j < WA_L ? __expf(s[tk][tq][r] - m) : 0.f;
          s[tk][tq][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      float delta = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[tk][tq][r] *= inv;  // P^T
          delta += s[tk][tq][r] * dp[tk][tq][r];
        }
      delta += __shfl_xor(delta, 32, 64);
      if (hh == 0) {
        srow[0][i] = m;
        srow[1][i] = inv;
        srow[2][i] = delta;
      }
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tk][tq][r] *= (dp[tk][tq][r] - delta);  // dS^T
      // dQ^T[c][query] = scale * sum_key K[key][c] dS^T[key][query]
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma_f32(sk[(tk * 32 + crow(r, hh)) * WA_LD + l31], s[tk][tq][r], o);
      if (i < WA_L) {
        store_cols(a.dqkv + token_map(a, win, i).base * C3 + head * a.hd, o, hh, a.hd, a.scale);
      }
    }
  }
  __syncthreads();

  // ---------------- pass B: queries on rows, keys on lanes ------------------------------------------------------
  {
    f32x16 s[2][2], dp[2][2];  // [tq][tk]: row = query tq*32 + crow(r, hh), col = key tk*32 + l31
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[x][y][r] = 0.f;
          dp[x][y][r] = 0.f;
        }
    for (int kk = 0; kk < a.hd / 2; ++kk) {
      const int c = 2 * kk + hh;
      const float k0 = sk[l31 * WA_LD + c], k1 = sk[(32 + l31) * WA_LD + c];
      const float q0 = sq[l31 * WA_LD + c], q1 = sq[(32 + l31) * WA_LD + c];
      const float v0 = sv[l31 * WA_LD + c], v1 = sv[(32 + l31) * WA_LD + c];
      const float g0 = sdo[l31 * WA_LD + c], g1 = sdo[(32 + l31) * WA_LD + c];
      s[0][0] = mfma_f32(q0, k0, s[0][0]);
      s[0][1] = mfma_f32(q0, k1, s[0][1]);
      s[1][0] = mfma_f32(q1, k0, s[1][0]);
      s[1][1] = mfma_f32(q1, k1, s[1][1]);
      dp[0][0] = mfma_f32(g0, v0, dp[0][0]);  // dP[query][key] = sum_c dO[query][c] V[key][c]
      dp[0][1] = mfma_f32(g0, v1, dp[0][1]);
      dp[1][0] = mfma_f32(g1, v0, dp[1][0]);
      dp[1][1] = mfma_f32(g1, v1, dp[1][1]);
    }
    float* dbias = a.dbias + head;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      const int j = tk * 32 + l31;
      const int jc = j < WA_L ? j : WA_L - 1;
      const int reg_j = sreg[jc];
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = tq * 32 + crow(r, hh);
          float pv = 0.f, ds = 0.f;
          if (i < WA_L && j < WA_L) {
            const int bx = a.bidx[i * WA_L + j];
            float x = s[tq][tk][r] + bias[bx * a.heads];
            if (sreg[i] != reg_j) x += -100.f;
            pv = __expf(x - srow[0][i]) * srow[1][i];
            ds = pv * (dp[tq][tk][r] - srow[2][i]);
            __hip_atomic_fetch_add(sdb + bx, ds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          s[tq][tk][r] = pv;   // P[query][key]
          dp[tq][tk][r] = ds;  // dS[query][key]
        }
      // dV^T[c][key] = sum_query dO[query][c] P[query][key];  dK^T[c][key] = sum_query Qs[query][c] dS[query][key]
      f32x16 ov, ok;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ov[r] = 0.f;
        ok[r] = 0.f;
      }
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = tq * 32 + crow(r, hh);
          ov = mfma_f32(sdo[i * WA_LD + l31], s[tq][tk][r], ov);
          ok = mfma_f32(sq[i * WA_LD + l31], dp[tq][tk][r], ok);
        }
      if (j < WA_L) {
        float* dst = a.dqkv + token_map(a, win, j).base * C3 + head * a.hd;
        store_cols(dst + a.C, ok, hh, a.hd, 1.f);
        store_cols(dst + 2 * a.C, ov, hh, a.hd, 1.f);
      }
    }
    __syncthreads();
    for (int i = lane; i < NBIAS; i += 64) atomicAdd(dbias + i * a.heads, sdb[i]);
  }
}

static int check(const AttnArgs& a) {
  if (a.B < 1 || a.H % WA_WS || a.W % WA_WS || a.heads < 1 || a.C != a.heads * a.hd || a.hd % 2 || a.hd > 32 ||
      a.hd < 2 || (a.shift != 0 && a.shift != WA_WS / 2))
    return NNZ_EINVAL;
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_window_attention_forward(const float* qkv, const float* bias_table, const int* bias_index, float* out,
                                            int B, int H, int W, int C, int heads, int shift, float scale,
                                            void* stream) {
  using namespace nnz;
  if (!qkv || !bias_table || !bias_index || !out) return NNZ_EINVAL;
  AttnArgs a = {};
  a.qkv = qkv; a.bias = bias_table; a.bidx = bias_index; a.out = out;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.hd = heads > 0 ? C / heads : 0; a.shift = shift; a.scale = scale;
  if (int rc = check(a)) return rc;
  const int nwin = B * (H / WA_WS) * (W / WA_WS);
  NNZ_LAUNCH(win_attn_fwd_kernel, dim3(nwin, heads), dim3(64), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_window_attention_backward(const float* qkv, const float* bias_table, const int* bias_index,
                                             const float* dout, float* dqkv, float* dbias_table, int B, int H, int W,
                                             int C, int heads, int shift, float scale, void* stream) {
  using namespace nnz;
  if (!qkv || !bias_table || !bias_index || !dout || !dqkv || !dbias_table) return NNZ_EINVAL;
  AttnArgs a = {};
  a.qkv = qkv; a.bias = bias_table; a.bidx = bias_index; a.dout = dout; a.dqkv = dqkv; a.dbias = dbias_table;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.hd = heads > 0 ? C / heads : 0; a.shift = shift; a.scale = scale;
  if (int rc = check(a)) return rc;
  hipError_t e = nnz::zero_async(dbias_table, sizeof(float) * heads * (2 * WA_WS - 1) * (2 * WA_WS - 1),
                                (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  const int nwin = B * (H / WA_WS) * (W / WA_WS);
  NNZ_LAUNCH(win_attn_bwd_kernel, dim3(nwin, heads), dim3(64), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
