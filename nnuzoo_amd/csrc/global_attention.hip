// Global multi-head self-attention core of the ViT blocks (monai SABlock as bound by the reference at
// /root/reference/nnunetv2/nets/unetr2net.py:10,1414-1428): out = softmax(q k^T * scale) v over ALL L patch tokens of a
// sample (L <= 1024 in the zoo: 512^2 / 16^2), head_dim 8 / 16 / 32, straight from the packed projection
// qkv [B][L][3][H][D] to the merged-head result [B][L][H * D].  Forward + backward, fp32 on v_mfma_f32_32x32x2_f32.
// Round 2 forwarded this to torch's scaled_dot_product_attention (library dispatch); this is the hand-written path.
//
// Flash-style (the L x L score matrix never exists), same operand tricks as csrc/window_attention.hip:
//   * scores are computed TRANSPOSED (keys on MFMA rows, queries on the lanes): a query's statistics live in one lane
//     pair, and P^T is the B operand of the P V product without moving data;
//   * the contraction over channels may visit the channels in any order, so lane half hh owns channels
//     [hh D/2, (hh+1) D/2) of its token's row: row operands are plain reads of a [token][channel] LDS image;
//   * a workgroup = 4 waves = 128 queries (or keys) of one (sample, head); the other side streams through LDS in blocks of
//     64 tokens that all four waves share.
// Backward without atomics (deterministic): one kernel owns 128 KEYS per workgroup and sweeps the queries (dK, dV), a second
// owns 128 QUERIES and sweeps the keys (dQ); both recompute P from the saved log-sum-exp, delta = rowsum(dO * O) is formed
// while the dO block is staged.
#include "common.hpp"

namespace nnz {

constexpr int GA_LD = 33;   // LDS row stride of a [token][channel] image (D <= 32, +1 pad)
constexpr int GA_KB = 64;   // tokens per streamed block

struct GAttnArgs {
  const float* qkv;   // [B][L][3][H][D]
  float* out;         // [B][L][H*D]
  float* lse;         // [B][H][L]  log-sum-exp of the scaled scores of every query
  const float* dout;  // [B][L][H*D]
  float* dqkv;        // [B][L][3][H][D]
  int B, L, H, D;
  float scale;
};

__device__ __forceinline__ f32x16 ga_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int ga_crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// stage rows [t0, t0 + 64) of one [L][D] operand (row stride `row_len`, tokens >= L zero-filled) into an LDS image
__device__ __forceinline__ void ga_stage(const float* src, long row_len, int D, int t0, int L, float mul, float* dst,
                                         int tid) {
  if ((D & 3) == 0) {
    const int q4 = D >> 2;
    for (int i = tid; i < GA_KB * q4; i += 256) {
      const int l = i / q4, c4 = (i - l * q4) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t0 + l < L) v = *reinterpret_cast<const f32x4*>(src + (long)(t0 + l) * row_len + c4);
      float* d = dst + l * GA_LD + c4;
      d[0] = v[0] * mul; d[1] = v[1] * mul; d[2] = v[2] * mul; d[3] = v[3] * mul;
    }
  } else {
    for (int i = tid; i < GA_KB * D; i += 256) {
      const int l = i / D, c = i - l * D;
      dst[l * GA_LD + c] = t0 + l < L ? src[(long)(t0 + l) * row_len + c] * mul : 0.f;
    }
  }
}
// row operand of a 32-token tile of an image: lane (token l31, half hh) gets channels hh D/2 + s
__device__ __forceinline__ void ga_rows(const float* img, int D, int tile, int l31, int hh, float (&v)[16]) {
  const float* p = img + (tile * 32 + l31) * GA_LD + hh * (D >> 1);
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = s < (D >> 1) ? p[s] : 0.f;
}
// the same straight from global memory (the side a wave keeps for its whole run)
__device__ __forceinline__ void ga_rows_global(const float* src, long row_len, int D, int t, int L, int hh, float mul,
                                               float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = 0.f;
  if (t < L) {
    const float* p = src + (long)t * row_len + hh * (D >> 1);
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < (D >> 1)) v[s] = p[s] * mul;
  }
}
// column operand: step r needs X[token tile*32 + crow(r, hh)][channel l31]
__device__ __forceinline__ void ga_cols(const float* img, int D, int tile, int l31, int hh, float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = l31 < D ? img[(tile * 32 + ga_crow(r, hh)) * GA_LD + l31] : 0.f;
}
__device__ __forceinline__ void ga_store_cols(float* dst, const f32x16& o, int hh, int D, float mul) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int c0 = 8 * g4 + 4 * hh;
    if ((D & 3) == 0) {
      if (c0 < D) {
        f32x4 v = {o[4 * g4] * mul, o[4 * g4 + 1] * mul, o[4 * g4 + 2] * mul, o[4 * g4 + 3] * mul};
        *reinterpret_cast<f32x4*>(dst + c0) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < D) dst[c0 + e] = o[4 * g4 + e] * mul;
    }
  }
}

// ---- forward: workgroup = (128 queries, head, sample); keys / values stream through LDS ---------------------------------
__global__ __launch_bounds__(256) void gattn_fwd_kernel(GAttnArgs a) {
  __shared__ float sk[GA_KB * GA_LD], sv[GA_KB * GA_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int D = a.D, steps = D >> 1;
  const long row = 3L * a.H * D;
  const float* base = a.qkv + (long)b * a.L * row + (long)h * D;
  const int q = blockIdx.x * 128 + wave * 32 + l31;  // this lane's query
  float qv[16];
  ga_rows_global(base, row, D, q, a.L, hh, a.scale, qv);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -3.0e38f, lsum = 0.f;
  for (int k0 = 0; k0 < a.L; k0 += GA_KB) {
    __syncthreads();
    ga_stage(base + (long)a.H * D, row, D, k0, a.L, 1.f, sk, tid);
    ga_stage(base + 2L * a.H * D, row, D, k0, a.L, 1.f, sv, tid);
    __syncthreads();
    f32x16 s[2];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      float kv[16];
      ga_rows(sk, D, tk, l31, hh, kv);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[tk][r] = 0.f;
#pragma unroll
      for (int st = 0; st < 16; ++st)
        if (st < steps) s[tk] = ga_mfma(kv[st], qv[st], s[tk]);
    }
    float bm = -3.0e38f;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = k0 + tk * 32 + ga_crow(r, hh) < a.L;
        s[tk][r] = ok ? s[tk][r] : -3.0e38f;
        bm = fmaxf(bm, s[tk][r]);
      }
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m, bm);
    const float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = k0 + tk * 32 + ga_crow(r, hh) < a.L;
        const float e = ok ? __expf(s[tk][r] - mn) : 0.f;
        s[tk][r] = e;
        ps += e;
      }
    lsum = lsum * alpha + ps;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      float vc[16];
      ga_cols(sv, D, tk, l31, hh, vc);
#pragma unroll
      for (int r = 0; r < 16; ++r) o = ga_mfma(vc[r], s[tk][r], o);
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  if (q < a.L) {
    ga_store_cols(a.out + ((long)b * a.L + q) * a.H * D + (long)h * D, o, hh, D, 1.f / lsum);
    if (hh == 0) a.lse[((long)b * a.H + h) * a.L + q] = m + __logf(lsum);
  }
}

// stage a 64-query block of Q * scale and dO, its log-sum-exp and delta = rowsum(dO * O) (shared by the four waves)
__device__ __forceinline__ void ga_stage_queries(const GAttnArgs& a, int b, int h, int q0, float* sq, float* sdo,
                                                 float* slse, float* sdelta, int tid) {
  const int D = a.D;
  const long row = 3L * a.H * D, orow = (long)a.H * D;
  const float* qb = a.qkv + (long)b * a.L * row + (long)h * D;
  const float* gb = a.dout + (long)b * a.L * orow + (long)h * D;
  const float* ob = a.out + (long)b * a.L * orow + (long)h * D;
  if (sq) ga_stage(qb, row, D, q0, a.L, a.scale, sq, tid);
  ga_stage(gb, orow, D, q0, a.L, 1.f, sdo, tid);
  if (tid < GA_KB) {
    const int q = q0 + tid;
    float dl = 0.f, ls = 0.f;
    if (q < a.L) {
      for (int c = 0; c < D; ++c) dl += gb[(long)q * orow + c] * ob[(long)q * orow + c];
      ls = a.lse[((long)b * a.H + h) * a.L + q];
    }
    sdelta[tid] = dl;
    slse[tid] = ls;
  }
}

// ---- backward, keys side: workgroup = (128 keys, head, sample) sweeps the queries; dK and dV without atomics -------------
__global__ __launch_bounds__(256) void gattn_bwd_kv_kernel(GAttnArgs a) {
  __shared__ float sq[GA_KB * GA_LD], sdo[GA_KB * GA_LD], slse[GA_KB], sdelta[GA_KB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int D = a.D, steps = D >> 1;
  const long row = 3L * a.H * D;
  const float* base = a.qkv + (long)b * a.L * row + (long)h * D;
  const int key = blockIdx.x * 128 + wave * 32 + l31;  // this lane's key
  float kv[16], vv[16];
  ga_rows_global(base + (long)a.H * D, row, D, key, a.L, hh, 1.f, kv);
  ga_rows_global(base + 2L * a.H * D, row, D, key, a.L, hh, 1.f, vv);
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dk[r] = 0.f;
    dv[r] = 0.f;
  }
  for (int q0 = 0; q0 < a.L; q0 += GA_KB) {
    __syncthreads();
    ga_stage_queries(a, b, h, q0, sq, sdo, slse, sdelta, tid);
    __syncthreads();
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      // S[query][key] and dP[query][key]: queries on rows (row operands from the images), this wave's keys on the lanes
      float qr[16], gr[16];
      ga_rows(sq, D, tq, l31, hh, qr);
      ga_rows(sdo, D, tq, l31, hh, gr);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = 0.f;
        dp[r] = 0.f;
      }
#pragma unroll
      for (int st = 0; st < 16; ++st)
        if (st < steps) {
          s = ga_mfma(qr[st], kv[st], s);
          dp = ga_mfma(gr[st], vv[st], dp);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = tq * 32 + ga_crow(r, hh);
        const bool ok = q0 + ql < a.L && key < a.L;
        const float p = ok ? __expf(s[r] - slse[ql]) : 0.f;
        s[r] = p;                               // P[query][key]
        dp[r] = p * (dp[r] - sdelta[ql]);       // dS[query][key]
      }
      float gc[16], qc[16];
      ga_cols(sdo, D, tq, l31, hh, gc);
      ga_cols(sq, D, tq, l31, hh, qc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dv = ga_mfma(gc[r], s[r], dv);          // dV^T[c][key] += dO[query][c] P[query][key]
        dk = ga_mfma(qc[r], dp[r], dk);         // dK^T[c][key] += (Q scale)[query][c] dS[query][key]
      }
    }
  }
  if (key < a.L) {
    float* dst = a.dqkv + ((long)b * a.L + key) * row + (long)h * D;
    ga_store_cols(dst + (long)a.H * D, dk, hh, D, 1.f);
    ga_store_cols(dst + 2L * a.H * D, dv, hh, D, 1.f);
  }
}

// ---- backward, queries side: workgroup = (128 queries, head, sample) sweeps the keys; dQ ---------------------------------
__global__ __launch_bounds__(256) void gattn_bwd_q_kernel(GAttnArgs a) {
  __shared__ float sk[GA_KB * GA_LD], sv[GA_KB * GA_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int D = a.D, steps = D >> 1;
  const long row = 3L * a.H * D, orow = (long)a.H * D;
  const float* base = a.qkv + (long)b * a.L * row + (long)h * D;
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  float qv[16], gv[16];
  ga_rows_global(base, row, D, q, a.L, hh, a.scale, qv);
  ga_rows_global(a.dout + (long)b * a.L * orow + (long)h * D, orow, D, q, a.L, hh, 1.f, gv);
  float lse = 0.f, delta = 0.f;
  if (q < a.L) {
    lse = a.lse[((long)b * a.H + h) * a.L + q];
    const float* gb = a.dout + ((long)b * a.L + q) * orow + (long)h * D;
    const float* ob = a.out + ((long)b * a.L + q) * orow + (long)h * D;
    for (int c = hh * steps; c < (hh + 1) * steps; ++c) delta += gb[c] * ob[c];
  }
  delta += __shfl_xor(delta, 32, 64);
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  for (int k0 = 0; k0 < a.L; k0 += GA_KB) {
    __syncthreads();
    ga_stage(base + (long)a.H * D, row, D, k0, a.L, 1.f, sk, tid);
    ga_stage(base + 2L * a.H * D, row, D, k0, a.L, 1.f, sv, tid);
    __syncthreads();
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      float kr[16], vr[16];
      ga_rows(sk, D, tk, l31, hh, kr);
      ga_rows(sv, D, tk, l31, hh, vr);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = 0.f;
        dp[r] = 0.f;
      }
#pragma unroll
      for (int st = 0; st < 16; ++st)
        if (st < steps) {
          s = ga_mfma(kr[st], qv[st], s);       // S^T[key][query]
          dp = ga_mfma(vr[st], gv[st], dp);     // dP^T[key][query] = sum_c V[key][c] dO[query][c]
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = k0 + tk * 32 + ga_crow(r, hh) < a.L && q < a.L;
        const float p = ok ? __expf(s[r] - lse) : 0.f;
        s[r] = p * (dp[r] - delta);             // dS^T[key][query]
      }
      float kc[16];
      ga_cols(sk, D, tk, l31, hh, kc);
#pragma unroll
      for (int r = 0; r < 16; ++r) dq = ga_mfma(kc[r], s[r], dq);   // dQ^T[c][query] += K[key][c] dS^T[key][query]
    }
  }
  if (q < a.L) ga_store_cols(a.dqkv + ((long)b * a.L + q) * row + (long)h * D, dq, hh, D, a.scale);
}

static int ga_check(const GAttnArgs& a) {
  if (a.B < 1 || a.L < 1 || a.L > 65536 || a.H < 1 || a.H > 65535 || a.B > 65535 || a.D < 2 || a.D > 32 || (a.D & 1))
    return NNZ_EINVAL;
  return NNZ_OK;
}

}  // namespace nnz

// out[b][l][h*D + c] = sum_j softmax_j(scale * q[b,l,h] . k[b,j,h]) v[b,j,h][c];  lse[b][h][l] saved for the backward
extern "C" int nnz_global_attention_forward(const float* qkv, float* out, float* lse, int B, int L, int H, int D,
                                            float scale, void* stream) {
  using namespace nnz;
  if (!qkv || !out || !lse) return NNZ_EINVAL;
  GAttnArgs a = {};
  a.qkv = qkv; a.out = out; a.lse = lse; a.B = B; a.L = L; a.H = H; a.D = D; a.scale = scale;
  if (int rc = ga_check(a)) return rc;
  NNZ_LAUNCH(gattn_fwd_kernel, dim3((L + 127) / 128, H, B), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// dqkv (all three parts WRITTEN, deterministic: no atomics) from qkv, the forward's out / lse and dout
extern "C" int nnz_global_attention_backward(const float* qkv, const float* out, const float* lse, const float* dout,
                                             float* dqkv, int B, int L, int H, int D, float scale, void* stream) {
  using namespace nnz;
  if (!qkv || !out || !lse || !dout || !dqkv) return NNZ_EINVAL;
  GAttnArgs a = {};
  a.qkv = qkv; a.out = const_cast<float*>(out); a.lse = const_cast<float*>(lse); a.dout = dout; a.dqkv = dqkv;
  a.B = B; a.L = L; a.H = H; a.D = D; a.scale = scale;
  if (int rc = ga_check(a)) return rc;
  NNZ_LAUNCH(gattn_bwd_kv_kernel, dim3((L + 127) / 128, H, B), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH(gattn_bwd_q_kernel, dim3((L + 127) / 128, H, B), dim3(256), 0, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
