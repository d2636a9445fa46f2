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
// v_mfma_f32_32x32x2_f32 (exact fp32, matching the reference's fp32 trainer nnUNetTrainerSwT2Net.train_step, no autocast).
// The score tile is computed TRANSPOSED (keys on rows, queries on lanes): a query's 64 (49 + pad) scores then live in two
// lanes' registers, so softmax needs one lane exchange instead of a 32-lane reduction, and P^T feeds the PV MFMA as its B
// operand with no data movement (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand", T12).
//
// Round 3 (the round-2 kernels ran one wave per workgroup, staged q / k / v through 25-34 KB of LDS per wave - one wave
// per SIMD - and fetched the relative-position bias with two dependent GLOBAL loads per score: 26 us forward / 75 us
// backward per call at ~0.02 of the fp32 MFMA rate):
//   * a workgroup = 4 waves = one HEAD and a run of windows: the head's 49 x 49 bias matrix is gathered ONCE into LDS
//     (transposed, so that the lanes of a score register read consecutive words) and serves every window of the run;
//   * every operand of a (window, head) is fetched from global memory ONCE, as 16-byte row loads into wave-private
//     49-row LDS images (a first attempt without any staging made ~36 dependent L2 round trips per window and was slower
//     than round 2); the contraction over channels may visit the channels in any order as long as both operands agree, so
//     lane half hh simply owns channels [hh hd/2, (hh+1) hd/2) of its token's image row: no transposed copies;
//   * operands needed with the TOKEN index in the contraction (V in the forward; K, Q, dO in the backward) are read per
//     MFMA step as one row segment of the image (lane = channel: conflict-free);
//   * the bias-table gradient is DETERMINISTIC: dS goes through LDS once per key tile, each of the 169 table entries
//     (entry = displacement (yi - yj, xi - xj), the layout of the reference's relative_position_index, swt2net.py:545)
//     sums its (7 - |dy|)(7 - |dx|) members in a fixed order, waves are folded in wave order and workgroups add
//     fixed-point integers (common.hpp FxAcc) - no float atomics anywhere.
#include "common.hpp"

// -DNNZ_WA_TIMESTAMPS=1 (tools/probes/wa_phase_probe.py builds its own library with it; never the shipped one): thread 0 of every
// workgroup of win_attn_bwd_pair_kernel records s_memtime at its phase boundaries into g_wa_ts[workgroup][16]
#ifndef NNZ_WA_TIMESTAMPS
#define NNZ_WA_TIMESTAMPS 0
#endif
#if NNZ_WA_TIMESTAMPS
__device__ unsigned long long* g_wa_ts_dev = nullptr;
#define NNZ_WA_TS(slot)                                                                                        \
  do {                                                                                                         \
    if (g_wa_ts_dev && threadIdx.x == 0) g_wa_ts_dev[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define NNZ_WA_TS(slot) do {} while (0)
#endif

namespace nnz {

constexpr int WA_L = 49;   // tokens per window
constexpr int WA_WS = 7;
constexpr int WA_LD = 36;  // LDS row stride (floats) of a [token][channel] image: head_dim <= 32; 36 keeps rows 16-byte aligned
                           // (one ds_write_b128 per staged piece, ds_read_b128 row operands: conflict-free in the 16-lane groups
                           // of a b128 access, 36 l mod 64 being 16 different multiples of 4) - round 5, was 33 with 4-byte accesses
constexpr int WA_NBIAS = (2 * WA_WS - 1) * (2 * WA_WS - 1);
constexpr int WA_BP = 64;  // pitch of the transposed bias matrix sbT[key][query]
constexpr int WA_NF_SLOT = 170;  // backward pair kernel: entry of the [176][2] fixed-point table (169 used) that flags a non-finite dS

struct AttnArgs {
  const float* qkv;    // [B][H][W][3C]
  const float* bias;   // relative_position_bias_table [(2*7-1)^2][heads]
  const int* bidx;     // relative_position_index [49][49] (int32)
  float* out;          // [B][H][W][C]
  const float* dout;   // [B][H][W][C]
  float* dqkv;         // [B][H][W][3C]
  float* dbias;        // gradient of the table [169][heads] (written by the launch's last workgroup)
  FxAcc* acc;          // [heads][169] fixed-point accumulators + launch counter (backward)
  unsigned* counter;
  int B, H, W, C, heads, hd, shift;
  int nwin, wpb;       // windows in total, windows per workgroup
  float scale;
  // round 5: H x W is the block's top / left padded grid (swt2net.py:643-645) and `out` / `dout` live on the UNPADDED grid
  // (H - py) x (W - px): the block's crop (:660) and its backward (zero rows for padded tokens) are folded into the addressing.
  // py = px = 0: out / dout on the same grid as qkv.
  int py, px;
  // round 5: dpart != null - every workgroup WRITES its share of the bias-table gradient to dpart[blockIdx.x][169][heads] and
  // is done (a fold record of the backward pass's grouped launch sums the shares in workgroup order); null: fixed-point adds +
  // the head's last workgroup writes dbias
  float* dpart;
};

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// token l of window `win`: element offset of its (b, y, x) position in units of one token row, and its shift-mask region
__device__ __forceinline__ void token_map(const AttnArgs& a, int win, int l, int& base, int& region, int& ubase) {
  const int nww = a.W / WA_WS, nwh = a.H / WA_WS;
  const int b = win / (nwh * nww);
  const int wi = (win / nww) % nwh, wj = win % nww;
  const int r = wi * WA_WS + l / WA_WS, c = wj * WA_WS + l % WA_WS;  // coordinates in the rolled image
  int y = r + a.shift, x = c + a.shift;
  if (y >= a.H) y -= a.H;
  if (x >= a.W) x -= a.W;
  base = (b * a.H + y) * a.W + x;
  ubase = (y >= a.py && x >= a.px) ? (b * (a.H - a.py) + (y - a.py)) * (a.W - a.px) + (x - a.px) : -1;
  region = 0;
  if (a.shift > 0) {
    const int hid = r < a.H - WA_WS ? 0 : (r < a.H - a.shift ? 1 : 2);
    const int wid = c < a.W - WA_WS ? 0 : (c < a.W - a.shift ? 1 : 2);
    region = 3 * hid + wid;
  }
}

// One [49][hd] operand of the window -> a wave-private LDS image [49][WA_LD] (16-byte global loads along the token's row).
// Every operand is fetched from global memory exactly once per (window, head): the round-3 first attempt read row and
// column operands straight from global memory step by step and spent its time in ~36 dependent L2 round trips per window.
__device__ __forceinline__ void stage_image(const float* src, long row_len, int ch0, int hd, const int* stok, float mul,
                                            float* dst, int lane) {
  if ((hd & 3) == 0) {
    const int q4 = hd >> 2;
    for (int i = lane; i < WA_L * q4; i += 64) {
      const int l = i / q4, c4 = (i - l * q4) * 4;
      const int t = stok[l];
      const f32x4 v = t >= 0 ? *reinterpret_cast<const f32x4*>(src + (long)t * row_len + ch0 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(dst + l * WA_LD + c4) = v * mul;
    }
  } else {
    for (int i = lane; i < WA_L * hd; i += 64) {
      const int l = i / hd, c = i - l * hd;
      const int t = stok[l];
      dst[l * WA_LD + c] = t >= 0 ? src[(long)t * row_len + ch0 + c] * mul : 0.f;
    }
  }
}

// The same staging in two halves, for head dims 4 / 8 / 16 / 32 (SwT2Net: 32 at every level).  stage_image's loop has a run-time
// trip count: every iteration is token-index read (LDS round trip) -> global load -> s_waitcnt vmcnt(0) -> LDS write, i.e. the seven
// pieces of an image were seven SERIAL L2 / HBM round trips (round 5, found in the ISA; measured worth: forward launches 5 - 12 %,
// the ~10 us launch floor itself did not move).  Here all pieces of all images of a wave are requested first -
// unconditionally: a padded token reads token 0 and is zeroed at the LDS write, so that no load sits in a branch of its own - and
// written afterwards: one round trip.
constexpr int WA_SP = (WA_L * 8 + 63) / 64;          // 16-byte pieces per lane of one image at hd = 32
__device__ __forceinline__ bool wa_split_ok(int hd) { return hd == 4 || hd == 8 || hd == 16 || hd == 32; }
// pieces K0 .. K1-1 of an image: requests (v), then - after the caller has requested everything it wants in flight - the writes
template <int K0, int K1>
__device__ __forceinline__ void fetch_image(const float* src, long row_len, int ch0, int hd, const int* stok,
                                            f32x4 (&v)[WA_SP], int lane) {
  const int q4 = hd >> 2, total = WA_L * q4, sh = 31 - __builtin_clz(q4);
  int t[WA_SP];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int i = lane + 64 * k;
    t[k] = stok[(i < total ? i : lane) >> sh];
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int i = lane + 64 * k;
    const int c4 = ((i < total ? i : lane) & (q4 - 1)) * 4;
    // 32-bit element offset from a wave-uniform base (the tensors of a launch are far below 2^32 bytes): one address register per
    // piece instead of a 64-bit pair and a 64-bit multiply
    const unsigned off = (unsigned)(t[k] >= 0 ? t[k] : 0) * (unsigned)row_len + (unsigned)(ch0 + c4);
    v[k] = *reinterpret_cast<const f32x4*>(src + off);
  }
}
template <int K0, int K1>
__device__ __forceinline__ void commit_image(const f32x4 (&v)[WA_SP], int hd, const int* stok, float mul, float* dst, int lane) {
  const int q4 = hd >> 2, total = WA_L * q4, sh = 31 - __builtin_clz(q4);
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int i = lane + 64 * k;
    if (i < total) {
      const int l = i >> sh, c4 = (i & (q4 - 1)) * 4;
      *reinterpret_cast<f32x4*>(dst + l * WA_LD + c4) = stok[l] >= 0 ? v[k] * mul : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}
// two images of one wave.  ROUNDS = 1: all 14 pieces in flight at once; 2: pieces 0..3 of both, then 4..6 (the pair kernel of the
// backward sits at its 256-register cap: 56 registers of pieces in flight spilled)
template <int ROUNDS>
__device__ __forceinline__ void stage_two(const float* srcA, long rlA, int chA, const int* stokA, float mulA, float* dstA,
                                          const float* srcB, long rlB, int chB, const int* stokB, float mulB, float* dstB,
                                          int hd, int lane) {
  if (wa_split_ok(hd)) {
    f32x4 pa[WA_SP], pb[WA_SP];
    constexpr int H = ROUNDS == 1 ? WA_SP : 4;
    fetch_image<0, H>(srcA, rlA, chA, hd, stokA, pa, lane);
    fetch_image<0, H>(srcB, rlB, chB, hd, stokB, pb, lane);
    commit_image<0, H>(pa, hd, stokA, mulA, dstA, lane);
    commit_image<0, H>(pb, hd, stokB, mulB, dstB, lane);
    if constexpr (ROUNDS == 2) {
      fetch_image<H, WA_SP>(srcA, rlA, chA, hd, stokA, pa, lane);
      fetch_image<H, WA_SP>(srcB, rlB, chB, hd, stokB, pb, lane);
      commit_image<H, WA_SP>(pa, hd, stokA, mulA, dstA, lane);
      commit_image<H, WA_SP>(pb, hd, stokB, mulB, dstB, lane);
    }
  } else {
    stage_image(srcA, rlA, chA, hd, stokA, mulA, dstA, lane);
    stage_image(srcB, rlB, chB, hd, stokB, mulB, dstB, lane);
  }
}
__device__ __forceinline__ void stage_one(const float* src, long rl, int ch, const int* stok, float mul, float* dst, int hd, int lane) {
  if (wa_split_ok(hd)) {
    f32x4 pa[WA_SP];
    fetch_image<0, WA_SP>(src, rl, ch, hd, stok, pa, lane);
    commit_image<0, WA_SP>(pa, hd, stok, mul, dst, lane);
  } else {
    stage_image(src, rl, ch, hd, stok, mul, dst, lane);
  }
}

// "row" operand of one 32-token tile from an image: lane (token l31, half hh) gets channels hh*hd/2 + s, s = 0 .. hd/2 - 1
// of its token's row (zeros for tokens >= 49): the contraction over channels visits them in this order for BOTH operands.
// Head dims that are multiples of 8 read whole 16-byte pieces (pitch 36); others word by word.
__device__ __forceinline__ void load_rows(const float* img, int hd, int tile, int l31, int hh, float (&v)[16]) {
  const int l = tile * 32 + l31;
  const int half = hd >> 1;
  const float* p = img + (l < WA_L ? l : 0) * WA_LD + hh * half;
  if ((hd & 7) == 0) {                       // half a multiple of 4: the lane's channels are whole 16-byte pieces
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (l < WA_L && 4 * q < half) t = *reinterpret_cast<const f32x4*>(p + 4 * q);
      v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = (l < WA_L && s < half) ? p[s] : 0.f;
  }
}

// the same row operand straight from global memory (forward: Q is used once per window, so it skips the image and the
// forward's LDS drops to two images per wave = two workgroups per CU)
__device__ __forceinline__ void load_rows_global(const float* src, long row_len, int ch0, int hd, const int* stok, int tile,
                                                 int l31, int hh, float mul, float (&v)[16]) {
  const int l = tile * 32 + l31;
  const int half = hd >> 1;
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = 0.f;
  if (l < WA_L) {
    const float* p = src + (long)stok[l] * row_len + ch0 + hh * half;
    if ((hd & 7) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * q < half) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
          v[4 * q] = t[0] * mul; v[4 * q + 1] = t[1] * mul; v[4 * q + 2] = t[2] * mul; v[4 * q + 3] = t[3] * mul;
        }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < half) v[s] = p[s] * mul;
    }
  }
}

// "column" operand: step r of a 32-token tile needs X[token tile*32 + crow(r, hh)][channel l31] - the token index is the
// contraction index of the MFMA, the lane is the channel (consecutive words of one image row: conflict-free)
__device__ __forceinline__ void load_cols(const float* img, int hd, int tile, int l31, int hh, float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = tile * 32 + crow(r, hh);
    v[r] = (j < WA_L && l31 < hd) ? img[j * WA_LD + l31] : 0.f;
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

// the head's bias matrix, transposed: sbT[key j][query i] = table[index[i][j]][head]   (once per workgroup).
// Round 5: the head's 169 table entries go to LDS first (`stab`, any 169 floats of LDS that nothing else uses before the caller's
// next barrier) and the 2 401 index words are plain coalesced loads - two INDEPENDENT global round trips instead of ten rounds of
// index load -> dependent table load (the gather alone was a third of the 12 us / 36 us floor of a launch).
__device__ __forceinline__ void stage_bias(const AttnArgs& a, int head, float* sbT, float* stab, int tid, int nthreads) {
  int idx[(WA_L * WA_L + 255) / 256];
#pragma unroll
  for (int q = 0; q < (WA_L * WA_L + 255) / 256; ++q) {
    const int e = tid + q * 256;
    idx[q] = e < WA_L * WA_L ? a.bidx[e] : 0;
  }
  for (int e = tid; e < WA_NBIAS; e += nthreads) stab[e] = a.bias[e * a.heads + head];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < (WA_L * WA_L + 255) / 256; ++q) {
    const int e = tid + q * 256;
    if (e < WA_L * WA_L) {
      const int i = e / WA_L, j = e - i * WA_L;
      sbT[j * WA_BP + i] = stab[idx[q]];
    }
  }
}

// wave-private LDS hand-over (one wave writes, the same wave's other lanes read): LDS instructions of a wave execute in
// order; the fence keeps the compiler from moving accesses across
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// S^T tiles of one query tile tq: s[tk] (keys tk*32 + crow(r, hh) on rows, query tq*32 + l31 on the lane) from the row
// operands, + bias + mask, softmax over the keys.  Returns P^T in s; m / inv are the query's row max and 1 / sum.
__device__ __forceinline__ void scores_T(const float (&kv)[2][16], const float (&qv)[16], int steps, const float* sbT,
                                         const int* sreg, int shift, int tq, int l31, int hh, f32x16 (&s)[2], float& m_out,
                                         float& inv_out) {
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[x][r] = 0.f;
#pragma unroll
  for (int st = 0; st < 16; ++st)
    if (st < steps) {
      s[0] = mfma_f32(kv[0][st], qv[st], s[0]);
      s[1] = mfma_f32(kv[1][st], qv[st], s[1]);
    }
  const int i = tq * 32 + l31;
  const int ic = i < WA_L ? i : WA_L - 1;
  const int reg_i = shift ? sreg[ic] : 0;
  float m = -3.0e38f;
#pragma unroll
  for (int tk = 0; tk < 2; ++tk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = tk * 32 + crow(r, hh);
      float x = -3.0e38f;
      if (j < WA_L) {
        x = s[tk][r] + sbT[j * WA_BP + ic];
        if (shift && sreg[j] != reg_i) x += -100.f;
      }
      s[tk][r] = x;
      m = fmaxf(m, x);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int tk = 0; tk < 2; ++tk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = tk * 32 + crow(r, hh);
      const float e = j < WA_L ? __expf(s[tk][r] - m) : 0.f;
      s[tk][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int tk = 0; tk < 2; ++tk)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[tk][r] *= inv;
  m_out = m;
  inv_out = inv;
}

constexpr int WA_IMG = WA_L * WA_LD;  // floats of one [49][WA_LD] image
static_assert(WA_IMG % 4 == 0, "images must keep 16-byte alignment");

__global__ __launch_bounds__(256) void win_attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sbT = smem;                                      // [49][64]
  float* wbase = smem + WA_L * WA_BP;                     // per wave: sk, sv images | stok, sreg, sutok
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y;
  const int C3 = 3 * a.C, hd = a.hd, steps = hd >> 1;
  float* sk = wbase + wave * (2 * WA_IMG + 192);
  float* sv = sk + WA_IMG;
  int* stok = reinterpret_cast<int*>(sv + WA_IMG);
  int* sreg = stok + 64;
  int* sutok = sreg + 64;      // token index on the unpadded grid of out / dout, -1 = a padded token
  stage_bias(a, head, sbT, wbase, tid, 256);   // wbase: image space, unused until the barrier below
  __syncthreads();
  const int w_end = (blockIdx.x + 1) * a.wpb < a.nwin ? (blockIdx.x + 1) * a.wpb : a.nwin;
  for (int win = blockIdx.x * a.wpb + wave; win < w_end; win += 4) {
    wave_sync();  // the previous window's readers are done with the images / stok / sreg
    {
      int base = 0, region = 0, ubase = -1;
      if (lane < WA_L) token_map(a, win, lane, base, region, ubase);
      stok[lane] = base;
      sreg[lane] = region;
      sutok[lane] = ubase;
    }
    wave_sync();
    float qv[2][16];
    load_rows_global(a.qkv, C3, head * hd, hd, stok, 0, l31, hh, a.scale, qv[0]);
    load_rows_global(a.qkv, C3, head * hd, hd, stok, 1, l31, hh, a.scale, qv[1]);
    stage_two<1>(a.qkv, C3, a.C + head * hd, stok, 1.f, sk, a.qkv, C3, 2 * a.C + head * hd, stok, 1.f, sv, hd, lane);
    wave_sync();
    float kv[2][16];
    load_rows(sk, hd, 0, l31, hh, kv[0]);
    load_rows(sk, hd, 1, l31, hh, kv[1]);
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
      f32x16 p[2];
      float m, inv;
      scores_T(kv, qv[tq], steps, sbT, sreg, a.shift, tq, l31, hh, p, m, inv);
      // O^T[channel][query] = sum_key V[key][channel] P^T[key][query]
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        float vc[16];
        load_cols(sv, hd, tk, l31, hh, vc);
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma_f32(vc[r], p[tk][r], o);
      }
      const int i = tq * 32 + l31;
      if (i < WA_L && sutok[i] >= 0) store_cols(a.out + (long)sutok[i] * a.C + head * hd, o, hh, hd, 1.f);
    }
  }
}

// forward with TWO waves per window (see win_attn_bwd_pair_kernel): role 0 stages K, role 1 stages V, each takes one query
// tile.  Two pairs per workgroup, 14 KB of images per pair.
__global__ __launch_bounds__(256) void win_attn_fwd_pair_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sbT = smem;                                      // [49][64]
  float* wbase = smem + WA_L * WA_BP;                     // per wave: sk, sv images | stok, sreg, sutok
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y;
  const int C3 = 3 * a.C, hd = a.hd, steps = hd >> 1;
  const int pair = wave >> 1, role = wave & 1;
  float* sk = wbase + pair * (2 * WA_IMG + 192);
  float* sv = sk + WA_IMG;
  int* stok = reinterpret_cast<int*>(sv + WA_IMG);
  int* sreg = stok + 64;
  int* sutok = sreg + 64;      // token index on the unpadded grid of out / dout, -1 = a padded token
  stage_bias(a, head, sbT, wbase, tid, 256);   // wbase: image space, unused until the barrier below
  __syncthreads();
  const int w_end = (blockIdx.x + 1) * a.wpb < a.nwin ? (blockIdx.x + 1) * a.wpb : a.nwin;
  const int iters = (a.wpb + 1) / 2;
  for (int it = 0; it < iters; ++it) {
    const int win = blockIdx.x * a.wpb + pair + 2 * it;
    const bool live = win < w_end;
    lds_barrier();  // the previous window's readers are done with the images / stok / sreg
    if (live && role == 0) {
      int base = 0, region = 0, ubase = -1;
      if (lane < WA_L) token_map(a, win, lane, base, region, ubase);
      stok[lane] = base;
      sreg[lane] = region;
      sutok[lane] = ubase;
    }
    lds_barrier();
    float qv[16];
    if (live) {
      load_rows_global(a.qkv, C3, head * hd, hd, stok, role, l31, hh, a.scale, qv);
      stage_one(a.qkv, C3, (role == 0 ? a.C : 2 * a.C) + head * hd, stok, 1.f, role == 0 ? sk : sv, hd, lane);
    }
    lds_barrier();
    if (!live) continue;
    float kv[2][16];
    load_rows(sk, hd, 0, l31, hh, kv[0]);
    load_rows(sk, hd, 1, l31, hh, kv[1]);
    {
      const int tq = role;
      f32x16 p[2];
      float m, inv;
      scores_T(kv, qv, steps, sbT, sreg, a.shift, tq, l31, hh, p, m, inv);
      // O^T[channel][query] = sum_key V[key][channel] P^T[key][query]
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        float vc[16];
        load_cols(sv, hd, tk, l31, hh, vc);
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma_f32(vc[r], p[tk][r], o);
      }
      const int i = tq * 32 + l31;
      if (i < WA_L && sutok[i] >= 0) store_cols(a.out + (long)sutok[i] * a.C + head * hd, o, hh, hd, 1.f);
    }
  }
}

// backward.  Pass A (transposed orientation, queries on lanes): P^T, dP^T, delta, dS^T -> dQ.
//            Pass B (queries on rows, keys on lanes): P, dP, dS -> dV, dK, and dS -> the bias-table gradient.
// Row statistics cross from A to B through LDS.  LDS per workgroup: the head's bias matrix and per wave four operand images,
// one [49][33] dS tile (its own pitch, WA_DSP) for the bias-gradient sums, the row statistics and the token map (~150 KB: one workgroup per CU).
constexpr int WA_DSP = 33;                                    // pitch of the per-key-tile dS image [query][key - 32 tk]
constexpr int WA_DS_FLOATS = (WA_L * WA_DSP + 3) / 4 * 4;      // the dS tile, rounded so that what follows stays 16-byte aligned
constexpr int WA_WAVE_FLOATS = 4 * WA_IMG + WA_DS_FLOATS + 3 * 64 + 3 * 64;  // sq sk sv sdo | sds | srow[3][64] | stok, sreg, sutok

__global__ __launch_bounds__(256) void win_attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sbT = smem;                                   // [49][64]
  float* sdb_all = smem + WA_L * WA_BP;                // [4 waves][176]
  float* wbase = sdb_all + 4 * 176;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y;
  const int C3 = 3 * a.C, hd = a.hd, steps = hd >> 1;
  float* sq = wbase + wave * WA_WAVE_FLOATS;           // Q * scale, K, V, dO images [49][33]
  float* sk = sq + WA_IMG;
  float* sv = sk + WA_IMG;
  float* sdo = sv + WA_IMG;
  float* sds = sdo + WA_IMG;                           // dS[query][key - 32 tk] of the current key tile
  float* srow = sds + WA_DS_FLOATS;                    // [3][64]: row max, 1 / sum, delta per query
  int* stok = reinterpret_cast<int*>(srow + 3 * 64);
  int* sreg = stok + 64;
  int* sutok = sreg + 64;      // token index on the unpadded grid of out / dout, -1 = a padded token
  float* sdb = sdb_all + wave * 176;
  const int qo = head * hd, ko = a.C + head * hd, vo = 2 * a.C + head * hd;
  stage_bias(a, head, sbT, wbase, tid, 256);   // wbase: image space, unused until the barrier below
  // the wave's share of the bias-table gradient: lane owns table entries lane, lane + 64, lane + 128
  float db_acc[3] = {0.f, 0.f, 0.f};
  __syncthreads();
  const int w_end = (blockIdx.x + 1) * a.wpb < a.nwin ? (blockIdx.x + 1) * a.wpb : a.nwin;
  for (int win = blockIdx.x * a.wpb + wave; win < w_end; win += 4) {
    wave_sync();
    {
      int base = 0, region = 0, ubase = -1;
      if (lane < WA_L) token_map(a, win, lane, base, region, ubase);
      stok[lane] = base;
      sreg[lane] = region;
      sutok[lane] = ubase;
    }
    wave_sync();
    stage_image(a.qkv, C3, qo, hd, stok, a.scale, sq, lane);
    stage_image(a.qkv, C3, ko, hd, stok, 1.f, sk, lane);
    stage_image(a.qkv, C3, vo, hd, stok, 1.f, sv, lane);
    stage_image(a.dout, a.C, qo, hd, sutok, 1.f, sdo, lane);
    wave_sync();

    // ---------------- pass A: keys on rows, queries on lanes --------------------------------------------------------
#pragma unroll 1
    for (int tq = 0; tq < 2; ++tq) {
      float qv[16], gv[16];
      load_rows(sq, hd, tq, l31, hh, qv);
      load_rows(sdo, hd, tq, l31, hh, gv);
      f32x16 s[2], dp[2];
      float m, inv;
      {
        float kv[2][16];
        load_rows(sk, hd, 0, l31, hh, kv[0]);
        load_rows(sk, hd, 1, l31, hh, kv[1]);
        scores_T(kv, qv, steps, sbT, sreg, a.shift, tq, l31, hh, s, m, inv);  // s = P^T
      }
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        float vv[16];
        load_rows(sv, hd, tk, l31, hh, vv);
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[tk][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 16; ++st)
          if (st < steps) dp[tk] = mfma_f32(vv[st], gv[st], dp[tk]);  // dP^T[key][query] = sum_c V[key][c] dO[query][c]
      }
      float delta = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += s[tk][r] * dp[tk][r];
      delta += __shfl_xor(delta, 32, 64);
      const int i = tq * 32 + l31;
      if (hh == 0) {
        srow[i] = m;
        srow[64 + i] = inv;
        srow[128 + i] = delta;
      }
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tk][r] *= (dp[tk][r] - delta);  // dS^T
      // dQ^T[c][query] = scale * sum_key K[key][c] dS^T[key][query]
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        float kc[16];
        load_cols(sk, hd, tk, l31, hh, kc);
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma_f32(kc[r], s[tk][r], o);
      }
      if (i < WA_L) store_cols(a.dqkv + (long)stok[i] * C3 + qo, o, hh, hd, a.scale);
    }
    wave_sync();  // srow is complete

    // ---------------- pass B: queries on rows, keys on lanes ---------------------------------------------------------
    // the row operands with roles swapped: A = Q (rows = queries), B = K (cols = keys); dP: A = dO, B = V
#pragma unroll 1
    for (int tk = 0; tk < 2; ++tk) {
      f32x16 s[2], dp[2];  // [tq]
      {
        float kv[16], vv[16];
        load_rows(sk, hd, tk, l31, hh, kv);
        load_rows(sv, hd, tk, l31, hh, vv);
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
          float qr[16], gr[16];
          load_rows(sq, hd, tq, l31, hh, qr);
          load_rows(sdo, hd, tq, l31, hh, gr);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s[tq][r] = 0.f;
            dp[tq][r] = 0.f;
          }
#pragma unroll
          for (int st = 0; st < 16; ++st)
            if (st < steps) {
              s[tq] = mfma_f32(qr[st], kv[st], s[tq]);
              dp[tq] = mfma_f32(gr[st], vv[st], dp[tq]);
            }
        }
      }
      const int j = tk * 32 + l31;
      const int jc = j < WA_L ? j : WA_L - 1;
      const int reg_j = a.shift ? sreg[jc] : 0;
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = tq * 32 + crow(r, hh);
          float pv = 0.f, ds = 0.f;
          if (i < WA_L && j < WA_L) {
            float x = s[tq][r] + sbT[j * WA_BP + i];
            if (a.shift && sreg[i] != reg_j) x += -100.f;
            pv = __expf(x - srow[i]) * srow[64 + i];
            ds = pv * (dp[tq][r] - srow[128 + i]);
          }
          s[tq][r] = pv;    // P[query][key]
          dp[tq][r] = ds;   // dS[query][key]
          if (i < WA_L) sds[i * WA_DSP + l31] = ds;
        }
      // dV^T[c][key] = sum_query dO[query][c] P[query][key];  dK^T[c][key] = sum_query Qs[query][c] dS[query][key]
      f32x16 ov, ok;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ov[r] = 0.f;
        ok[r] = 0.f;
      }
#pragma unroll
      for (int tq = 0; tq < 2; ++tq) {
        float gc[16], qc[16];
        load_cols(sdo, hd, tq, l31, hh, gc);
        load_cols(sq, hd, tq, l31, hh, qc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ov = mfma_f32(gc[r], s[tq][r], ov);
          ok = mfma_f32(qc[r], dp[tq][r], ok);
        }
      }
      if (j < WA_L) {
        float* dst = a.dqkv + (long)stok[j] * C3;
        store_cols(dst + ko, ok, hh, hd, 1.f);
        store_cols(dst + vo, ov, hh, hd, 1.f);
      }
      // bias-table gradient, this key tile's share: every table entry sums its members (i, j) in raster order of i
      wave_sync();
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int bx = lane + 64 * e;
        if (bx < WA_NBIAS) {
          // entry bx = (yi - yj + 6) * 13 + (xi - xj + 6): all (i, j) with that displacement
          const int dy = bx / 13 - 6, dx = bx % 13 - 6;
          const int y0 = dy > 0 ? dy : 0, y1 = dy < 0 ? WA_WS + dy : WA_WS;
          const int x0 = dx > 0 ? dx : 0, x1 = dx < 0 ? WA_WS + dx : WA_WS;
          // (round 5, phase probe: this loop was a chain of up to ~90 dependent LDS reads on the lane that owns displacement (0, 0).
          //  The seven reads of a row are now independent - invalid members read column 0 and add 0.0f, so the sum is the same
          //  bits as before: members in (yi, xi) order)
          float t = 0.f;
          for (int yi = y0; yi < y1; ++yi) {
            float v[WA_WS];
#pragma unroll
            for (int xi = 0; xi < WA_WS; ++xi) {
              const int jj = (yi - dy) * WA_WS + (xi - dx) - 32 * tk;
              const bool ok = xi >= x0 && xi < x1 && jj >= 0 && jj < 32;
              const float r = sds[(yi * WA_WS + xi) * WA_DSP + (ok ? jj : 0)];
              v[xi] = ok ? r : 0.f;
            }
#pragma unroll
            for (int xi = 0; xi < WA_WS; ++xi) t += v[xi];
          }
          db_acc[e] += t;
        }
      }
      wave_sync();  // the next key tile overwrites sds
    }
  }
  // ---- fold the four waves in wave order, one fixed-point add per table entry and workgroup, last workgroup writes ------
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (lane + 64 * e < WA_NBIAS) sdb[lane + 64 * e] = db_acc[e];
  __syncthreads();
  if (a.dpart) {      // deferred fold (fused Swin block): no atomics, no ticket
    if (tid < WA_NBIAS)
      a.dpart[((long)blockIdx.x * WA_NBIAS + tid) * a.heads + head] =
          (sdb_all[tid] + sdb_all[176 + tid]) + (sdb_all[2 * 176 + tid] + sdb_all[3 * 176 + tid]);
    return;
  }
  if (tid < WA_NBIAS) {
    const float t = (sdb_all[tid] + sdb_all[176 + tid]) + (sdb_all[2 * 176 + tid] + sdb_all[3 * 176 + tid]);
    fx_add(a.acc, (long)head * WA_NBIAS + tid, (long)a.heads * WA_NBIAS, blockIdx.x, (double)t);
  }
  // One ticket per HEAD (its counter lives behind the accumulator bank): the head's last workgroup reads its 169 entries -
  // one take per thread, a single memory round trip.  (One ticket per launch made the last workgroup take all 169 * heads
  // entries, 16 dependent round trips per thread at 24 heads: a fixed 65 us per launch - tools/bench_window_attention.py
  // measured 86 us for 48 (window, head) problems as for 600.)
  const long nrec = (long)a.heads * WA_NBIAS;
  unsigned* ticket = reinterpret_cast<unsigned*>(a.acc[(long)FX_REP * nrec + head].w);
  if (last_workgroup(ticket, gridDim.x) && tid < WA_NBIAS)
    a.dbias[tid * a.heads + head] = (float)fx_take(a.acc, (long)head * WA_NBIAS + tid, nrec);
}

// The same backward with TWO waves per window (launches of fewer (window, head) problems than SIMDs: 120 of the 144 launches of
// a SwT2Net pass sit on the latency of one window's chain - tools/bench_window_attention.py: 62-68 us for 48 ... 600
// problems).  Wave pair p = wave >> 1 owns a window; role = wave & 1 stages half of the images (Q, K / V, dO), takes query
// tile `role` in pass A and key tile `role` in pass B (each with its own dS tile).  The pair meets at workgroup barriers
// (both pairs run the same number of iterations; a pair without a window idles through them).
static_assert(WA_L * WA_DSP <= WA_IMG, "a dS tile must fit the image it replaces");
constexpr int WA_PAIR_FLOATS = 4 * WA_IMG + 3 * 64 + 3 * 64;  // sq sk sv sdo (sk / sv double as the dS tiles) | srow[3][64] | stok, sreg, sutok

__global__ __launch_bounds__(256, 2) void win_attn_bwd_pair_kernel(AttnArgs a) {
  NNZ_WA_TS(0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sbT = smem;                                   // [49][64]
  float* sdb_all = smem + WA_L * WA_BP;                // [4 waves][176]
  float* wbase = sdb_all + 4 * 176;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y;
  const int C3 = 3 * a.C, hd = a.hd, steps = hd >> 1;
  const int pair = wave >> 1, role = wave & 1;
  float* sq = wbase + pair * WA_PAIR_FLOATS;           // Q * scale, K, V, dO images [49][33], shared by the pair
  float* sk = sq + WA_IMG;
  float* sv = sk + WA_IMG;
  float* sdo = sv + WA_IMG;
  // dS[query][key - 32 role] of this wave's key tile lives ON the K (role 0) / V (role 1) image: both are dead once the pair
  // has taken its K / V row operands of pass B (barrier below) - WA_PAIR_FLOATS per pair (four [49][36] images + rows + token tables: 29.8 KB) instead of five images, two workgroups per CU
  float* sds = role == 0 ? sk : sv;
  float* srow = sdo + WA_IMG;                          // [3][64]: row max, 1 / sum, delta per query
  int* stok = reinterpret_cast<int*>(srow + 3 * 64);
  int* sreg = stok + 64;
  int* sutok = sreg + 64;      // token index on the unpadded grid of out / dout, -1 = a padded token
  const int qo = head * hd, ko = a.C + head * hd, vo = 2 * a.C + head * hd;
  NNZ_WA_TS(1);
  stage_bias(a, head, sbT, wbase, tid, 256);   // wbase: image space, unused until the barrier below
  // the wave's share of the bias-table gradient: lane owns table entries lane, lane + 64, lane + 128
  // the workgroup's share of the bias-table gradient: ONE table of 169 fixed-point sums in LDS, two 64-bit words per entry like
  // common.hpp's FxAcc - coarse (units of 2^-10: |v| < 2^52) and fine (the remainder in units of 2^-58) - so that gradients of any
  // fp32 magnitude keep 48 bits below the coarse unit (a single 2^-40 word saturated at |dS| > 8.4e6: the deep decoder stages of
  // tests/test_u2net_swt.py reach that).  Integer adds are associative: the order the four waves' adds arrive in does not matter,
  // the result is bit-identical run to run.  Every dS value goes to its entry with two ds_add_u64 (the 64 lanes of an instruction
  // hold distinct keys: at most 2-way conflicts between the half-waves' rows) - 64 instructions per wave and window instead of the dS
  // tile through LDS and a loop of per-entry sums (phase probe, round 5: that loop was the largest single piece of pass B)
  unsigned long long* sfx = reinterpret_cast<unsigned long long*>(sdb_all);       // [176][2]
  for (int e = tid; e < 2 * 176; e += 256) sfx[e] = 0ull;
  __syncthreads();
  const int w_end = (blockIdx.x + 1) * a.wpb < a.nwin ? (blockIdx.x + 1) * a.wpb : a.nwin;
  const int iters = (a.wpb + 1) / 2;
  NNZ_WA_TS(2);
  const int lane_outer = lane;
  for (int it = 0; it < iters; ++it) {
    // lane-derived values are recomputed per window: hoisted out of this loop they (LDS offsets of every row / column read
    // below) held enough registers across it that the image pieces in flight spilled
    int lane = lane_outer;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, hh = lane >> 5;
    const int win = blockIdx.x * a.wpb + pair + 2 * it;
    const bool live = win < w_end;
    lds_barrier();   // the previous window's readers are done with the images / stok / sreg
    if (live && role == 0) {
      int base = 0, region = 0, ubase = -1;
      if (lane < WA_L) token_map(a, win, lane, base, region, ubase);
      stok[lane] = base;
      sreg[lane] = region;
      sutok[lane] = ubase;
    }
    lds_barrier();
    if (live) {
      // (batched staging in ONE round - all 14 pieces of the wave's two images in flight, stage_two<1>; two rounds of 8 + 6 pieces
      //  were tried when the kernel sat at its 256-register cap and were no faster)
      stage_two<1>(a.qkv, C3, role == 0 ? qo : vo, stok, role == 0 ? a.scale : 1.f, role == 0 ? sq : sv,
                   role == 0 ? a.qkv : a.dout, role == 0 ? (long)C3 : (long)a.C, role == 0 ? ko : qo, role == 0 ? stok : sutok, 1.f,
                   role == 0 ? sk : sdo, hd, lane);
    }
    lds_barrier();

    if (it == 0) NNZ_WA_TS(3);
    // ---------------- pass A: keys on rows, queries on lanes --------------------------------------------------------
    if (live) {
      const int tq = role;
      float qv[16], gv[16];
      load_rows(sq, hd, tq, l31, hh, qv);
      load_rows(sdo, hd, tq, l31, hh, gv);
      f32x16 s[2], dp[2];
      float m, inv;
      {
        float kv[2][16];
        load_rows(sk, hd, 0, l31, hh, kv[0]);
        load_rows(sk, hd, 1, l31, hh, kv[1]);
        scores_T(kv, qv, steps, sbT, sreg, a.shift, tq, l31, hh, s, m, inv);  // s = P^T
      }
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        float vv[16];
        load_rows(sv, hd, tk, l31, hh, vv);
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[tk][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 16; ++st)
          if (st < steps) dp[tk] = mfma_f32(vv[st], gv[st], dp[tk]);  // dP^T[key][query] = sum_c V[key][c] dO[query][c]
      }
      float delta = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += s[tk][r] * dp[tk][r];
      delta += __shfl_xor(delta, 32, 64);
      const int i = tq * 32 + l31;
      if (hh == 0) {
        srow[i] = m;
        srow[64 + i] = inv;
        srow[128 + i] = delta;
      }
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tk][r] *= (dp[tk][r] - delta);  // dS^T
      // dQ^T[c][query] = scale * sum_key K[key][c] dS^T[key][query]
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        float kc[16];
        load_cols(sk, hd, tk, l31, hh, kc);
#pragma unroll
        for (int r = 0; r < 16; ++r) o = mfma_f32(kc[r], s[tk][r], o);
      }
      if (i < WA_L) store_cols(a.dqkv + (long)stok[i] * C3 + qo, o, hh, hd, a.scale);
    }
    if (it == 0) NNZ_WA_TS(4);
    lds_barrier();  // srow is complete (both query tiles)

    // ---------------- pass B: queries on rows, keys on lanes ---------------------------------------------------------
    // the row operands with roles swapped: A = Q (rows = queries), B = K (cols = keys); dP: A = dO, B = V
    float kvB[16], vvB[16];
    if (live) {
      load_rows(sk, hd, role, l31, hh, kvB);
      load_rows(sv, hd, role, l31, hh, vvB);
    }
    lds_barrier();  // both waves hold their K / V rows: the two images become the dS tiles
    if (it == 0) NNZ_WA_TS(5);
    if (live) {
      const int tk = role;
      f32x16 s[2], dp[2];  // [tq]
      {
        float (&kv)[16] = kvB;
        float (&vv)[16] = vvB;
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
          float qr[16], gr[16];
          load_rows(sq, hd, tq, l31, hh, qr);
          load_rows(sdo, hd, tq, l31, hh, gr);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s[tq][r] = 0.f;
            dp[tq][r] = 0.f;
          }
#pragma unroll
          for (int st = 0; st < 16; ++st)
            if (st < steps) {
              s[tq] = mfma_f32(qr[st], kv[st], s[tq]);
              dp[tq] = mfma_f32(gr[st], vv[st], dp[tq]);
            }
        }
      }
      const int j = tk * 32 + l31;
      const int jc = j < WA_L ? j : WA_L - 1;
      const int yj = (jc * 37) >> 8, xj = jc - 7 * yj;
      const int reg_j = a.shift ? sreg[jc] : 0;
      bool nonfinite = false;
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = tq * 32 + crow(r, hh);
          float pv = 0.f, ds = 0.f;
          if (i < WA_L && j < WA_L) {
            float x = s[tq][r] + sbT[j * WA_BP + i];
            if (a.shift && sreg[i] != reg_j) x += -100.f;
            pv = __expf(x - srow[i]) * srow[64 + i];
            ds = pv * (dp[tq][r] - srow[128 + i]);
          }
          s[tq][r] = pv;    // P[query][key]
          dp[tq][r] = ds;   // dS[query][key]
          nonfinite |= !(fabsf(ds) < 3.0e38f);        // NaN / Inf: the integer conversions below would turn them into finite sums
          if (i < WA_L && j < WA_L) {
            // entry (yi - yj + 6) * 13 + (xi - xj + 6); i / 7 = (i * 37) >> 8 for i < 64
            const int yi = (i * 37) >> 8, xi = i - 7 * yi;
            const float hi = rintf(ds * 0x1p10f);                       // exact: |ds| < 2^52 / 2^10 is far beyond fp32 gradients that matter
            const float lo = (ds - hi * 0x1p-10f) * 0x1p58f;            // |remainder| <= 2^-11: fits 2^47
            unsigned long long* q = sfx + 2 * ((yi - yj + 6) * 13 + (xi - xj + 6));
            atomicAdd(q, (unsigned long long)(long long)__float2ll_rn(hi));
            atomicAdd(q + 1, (unsigned long long)(long long)__float2ll_rn(lo));
          }
        }
      if (nonfinite) sfx[2 * WA_NF_SLOT] = 1ull;      // (same value from every writer: a plain store)
      // dV^T[c][key] = sum_query dO[query][c] P[query][key];  dK^T[c][key] = sum_query Qs[query][c] dS[query][key]
      f32x16 ov, ok;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ov[r] = 0.f;
        ok[r] = 0.f;
      }
#pragma unroll
      for (int tq = 0; tq < 2; ++tq) {
        float gc[16], qc[16];
        load_cols(sdo, hd, tq, l31, hh, gc);
        load_cols(sq, hd, tq, l31, hh, qc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ov = mfma_f32(gc[r], s[tq][r], ov);
          ok = mfma_f32(qc[r], dp[tq][r], ok);
        }
      }
      if (j < WA_L) {
        float* dst = a.dqkv + (long)stok[j] * C3;
        store_cols(dst + ko, ok, hh, hd, 1.f);
        store_cols(dst + vo, ov, hh, hd, 1.f);
      }
    }
  }
  NNZ_WA_TS(6);
  // ---- fold the four waves in wave order, one fixed-point add per table entry and workgroup, last workgroup writes ------
  __syncthreads();
  // A non-finite dS anywhere in the workgroup makes its whole share NaN (the float sum this table replaces carried
  // NaN / Inf into relative_position_bias_table.grad, and the fp32 Swin trainers run without a GradScaler - nothing else would flag
  // the step for that parameter); fx_add counts non-finite partials, so the fixed-point path propagates it too.
  const bool wg_nonfinite = sfx[2 * WA_NF_SLOT] != 0ull;
  if (a.dpart) {      // deferred fold (fused Swin block): no atomics, no ticket
    if (tid < WA_NBIAS)
      a.dpart[((long)blockIdx.x * WA_NBIAS + tid) * a.heads + head] = wg_nonfinite ? __builtin_nanf("") :
          (float)((double)(long long)sfx[2 * tid] * 0x1p-10 + (double)(long long)sfx[2 * tid + 1] * 0x1p-58);
    return;
  }
  if (tid < WA_NBIAS) {
    const float t = wg_nonfinite ? __builtin_nanf("") :
        (float)((double)(long long)sfx[2 * tid] * 0x1p-10 + (double)(long long)sfx[2 * tid + 1] * 0x1p-58);
    fx_add(a.acc, (long)head * WA_NBIAS + tid, (long)a.heads * WA_NBIAS, blockIdx.x, (double)t);
  }
  // One ticket per HEAD (its counter lives behind the accumulator bank): the head's last workgroup reads its 169 entries -
  // one take per thread, a single memory round trip.  (One ticket per launch made the last workgroup take all 169 * heads
  // entries, 16 dependent round trips per thread at 24 heads: a fixed 65 us per launch - tools/bench_window_attention.py
  // measured 86 us for 48 (window, head) problems as for 600.)
  const long nrec = (long)a.heads * WA_NBIAS;
  unsigned* ticket = reinterpret_cast<unsigned*>(a.acc[(long)FX_REP * nrec + head].w);
  if (last_workgroup(ticket, gridDim.x) && tid < WA_NBIAS)
    a.dbias[tid * a.heads + head] = (float)fx_take(a.acc, (long)head * WA_NBIAS + tid, nrec);
  NNZ_WA_TS(7);
}

static int check(const AttnArgs& a) {
  if (a.B < 1 || a.H % WA_WS || a.W % WA_WS || a.heads < 1 || a.C != a.heads * a.hd || a.hd % 2 || a.hd > 32 ||
      a.hd < 2 || (a.shift != 0 && a.shift != WA_WS / 2) || (long)a.B * a.H * a.W * 3 * a.C >= (1L << 32))   // fetch_image: 32-bit element offsets
    return NNZ_EINVAL;
  return NNZ_OK;
}

// windows per workgroup: a multiple of 4 (one per wave and round) that still leaves >= ~768 workgroups, at most 16
static int windows_per_wg(int nwin, int heads) {
  int wpb = 4;
  while (wpb < 16 && (long)((nwin + 2 * wpb - 1) / (2 * wpb)) * heads >= 768) wpb *= 2;
  return wpb;
}

}  // namespace nnz

#if NNZ_WA_TIMESTAMPS
extern "C" int wa_probe_set_timestamps(void* buf) {   // (probe builds only: not part of the C-ABI of include/nnuzoo_hip.h)
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wa_ts_dev), &buf, sizeof(buf));
}
#endif

static int wa_forward(const float* qkv, const float* bias_table, const int* bias_index, float* out, int B, int H, int W, int C,
                      int heads, int shift, float scale, int py, int px, void* stream) {
  using namespace nnz;
  if (!qkv || !bias_table || !bias_index || !out || py < 0 || px < 0 || py >= H || px >= W) return NNZ_EINVAL;
  AttnArgs a = {};
  a.qkv = qkv; a.bias = bias_table; a.bidx = bias_index; a.out = out;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.hd = heads > 0 ? C / heads : 0; a.shift = shift; a.scale = scale;
  a.py = py; a.px = px;
  if (int rc = check(a)) return rc;
  a.nwin = B * (H / WA_WS) * (W / WA_WS);
  static const int pair_mode = [] { const char* v = getenv("NNZ_WA_PAIR"); return v ? atoi(v) : 1; }();
  if (pair_mode) {
    a.wpb = a.nwin < 2 ? 1 : 2;
    while (a.wpb < 8 && (long)((a.nwin + 2 * a.wpb - 1) / (2 * a.wpb)) * heads >= 2048) a.wpb *= 2;
    const int ldsp = (WA_L * WA_BP + 2 * (2 * WA_IMG + 192)) * (int)sizeof(float);
    static DynLdsCache cachep;
    hipError_t ep = ensure_dyn_lds(reinterpret_cast<const void*>(win_attn_fwd_pair_kernel), ldsp, cachep);
    if (ep != hipSuccess) return (int)ep;
    NNZ_LAUNCH(win_attn_fwd_pair_kernel, dim3((a.nwin + a.wpb - 1) / a.wpb, heads), dim3(256), ldsp, (hipStream_t)stream, a);
    NNZ_LAUNCH_CHECK();
    return NNZ_OK;
  }
  a.wpb = windows_per_wg(a.nwin, heads);
  const int lds = (WA_L * WA_BP + 4 * (2 * WA_IMG + 192)) * (int)sizeof(float);
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(win_attn_fwd_kernel), lds, cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(win_attn_fwd_kernel, dim3((a.nwin + a.wpb - 1) / a.wpb, heads), dim3(256), lds, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// acc: heads * 169 zeroed fixed-point records (nnz_fxacc_bytes() each) followed by `heads` zeroed 32-byte ticket records
// (i.e. at least heads * 170 records of nnz_fxacc_bytes()); all left zero.  `counter` is unused (kept in the signature).
// dbias_table is WRITTEN (no zero fill needed) and bit-identical run to run.
// workgroups along x of a backward launch (= the number of bias-gradient shares in `dpart` mode)
static int wa_backward_groups(int nwin, int heads) {
  static const int pair_mode = [] { const char* v = getenv("NNZ_WA_PAIR"); return v ? atoi(v) : 1; }();
  int wpb;
  if (pair_mode) {
    wpb = nwin < 2 ? 1 : 2;
    while (wpb < 8 && (long)((nwin + 2 * wpb - 1) / (2 * wpb)) * heads >= 1024) wpb *= 2;
  } else {
    wpb = nnz::windows_per_wg(nwin, heads);
  }
  return (nwin + wpb - 1) / wpb;
}

static int wa_backward(const float* qkv, const float* bias_table, const int* bias_index, const float* dout, float* dqkv,
                       float* dbias_table, void* acc, void* counter, int B, int H, int W, int C, int heads, int shift,
                       float scale, int py, int px, void* stream, float* dpart = nullptr) {
  using namespace nnz;
  if (!qkv || !bias_table || !bias_index || !dout || !dqkv || (!dpart && (!dbias_table || !acc)) || py < 0 || px < 0 || py >= H ||
      px >= W)
    return NNZ_EINVAL;
  AttnArgs a = {};
  a.qkv = qkv; a.bias = bias_table; a.bidx = bias_index; a.dout = dout; a.dqkv = dqkv; a.dbias = dbias_table;
  a.acc = (FxAcc*)acc; a.counter = (unsigned*)counter;
  a.B = B; a.H = H; a.W = W; a.C = C; a.heads = heads; a.hd = heads > 0 ? C / heads : 0; a.shift = shift; a.scale = scale;
  a.py = py; a.px = px; a.dpart = dpart;
  if (int rc = check(a)) return rc;
  a.nwin = B * (H / WA_WS) * (W / WA_WS);
  a.wpb = windows_per_wg(a.nwin, heads);
  // two waves per window (win_attn_bwd_pair_kernel) at every size: with the dS tiles on the K / V images a pair needs 27 KB
  // of LDS, two workgroups = four windows = eight waves fit a CU, and a window's chain is 36-39 us instead of 62-68
  // (tools/bench_window_attention.py, NNZ_WA_PAIR=0 selects the one-wave kernel: 2 166 problems 207 -> 142 us, 1 200: 137 ->
  // 96, 600: 67 -> 48, <= 432: 65 -> 37)
  static const int pair_mode = [] { const char* v = getenv("NNZ_WA_PAIR"); return v ? atoi(v) : 1; }();
  if (pair_mode) {
    a.wpb = a.nwin < 2 ? 1 : 2;
    while (a.wpb < 8 && (long)((a.nwin + 2 * a.wpb - 1) / (2 * a.wpb)) * heads >= 1024) a.wpb *= 2;
    const int ldsp = (WA_L * WA_BP + 4 * 176 + 2 * WA_PAIR_FLOATS) * (int)sizeof(float);
    static DynLdsCache cachep;
    hipError_t ep = ensure_dyn_lds(reinterpret_cast<const void*>(win_attn_bwd_pair_kernel), ldsp, cachep);
    if (ep != hipSuccess) return (int)ep;
    NNZ_LAUNCH(win_attn_bwd_pair_kernel, dim3((a.nwin + a.wpb - 1) / a.wpb, heads), dim3(256), ldsp, (hipStream_t)stream, a);
    NNZ_LAUNCH_CHECK();
    return NNZ_OK;
  }
  const int lds = (WA_L * WA_BP + 4 * 176 + 4 * WA_WAVE_FLOATS) * (int)sizeof(float);
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(win_attn_bwd_kernel), lds, cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH(win_attn_bwd_kernel, dim3((a.nwin + a.wpb - 1) / a.wpb, heads), dim3(256), lds, (hipStream_t)stream, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

extern "C" int nnz_window_attention_forward(const float* qkv, const float* bias_table, const int* bias_index, float* out,
                                            int B, int H, int W, int C, int heads, int shift, float scale,
                                            void* stream) {
  return wa_forward(qkv, bias_table, bias_index, out, B, H, W, C, heads, shift, scale, 0, 0, stream);
}
extern "C" int nnz_window_attention_backward(const float* qkv, const float* bias_table, const int* bias_index,
                                             const float* dout, float* dqkv, float* dbias_table, void* acc, void* counter,
                                             int B, int H, int W, int C, int heads, int shift, float scale, void* stream) {
  return wa_backward(qkv, bias_table, bias_index, dout, dqkv, dbias_table, acc, counter, B, H, W, C, heads, shift, scale, 0, 0,
                     stream);
}
// round 5: the same pair inside the fused Swin block - qkv / dqkv on the block's top / left padded grid H x W, out / dout on the
// unpadded grid (H - py) x (W - px): the rows of padded tokens are never written (forward) and read as zeros (backward)
extern "C" int nnz_window_attention_forward_pad(const float* qkv, const float* bias_table, const int* bias_index, float* out,
                                                int B, int H, int W, int C, int heads, int shift, float scale, int py,
                                                int px, void* stream) {
  return wa_forward(qkv, bias_table, bias_index, out, B, H, W, C, heads, shift, scale, py, px, stream);
}
extern "C" int nnz_window_attention_backward_pad(const float* qkv, const float* bias_table, const int* bias_index,
                                                 const float* dout, float* dqkv, float* dbias_table, void* acc,
                                                 void* counter, int B, int H, int W, int C, int heads, int shift,
                                                 float scale, int py, int px, void* stream) {
  return wa_backward(qkv, bias_table, bias_index, dout, dqkv, dbias_table, acc, counter, B, H, W, C, heads, shift, scale, py,
                     px, stream);
}
// the backward with the bias-table gradient left as per-workgroup shares: dpart[nnz_window_attention_backward_parts(B, H, W,
// heads)][169][heads], summed in share order by a fold record (nnz_dense32_group_fill_fold, n = 169 * heads)
extern "C" long nnz_window_attention_backward_parts(int B, int H, int W, int heads) {
  if (B < 1 || H < 7 || W < 7 || H % 7 || W % 7 || heads < 1) return 0;
  return wa_backward_groups(B * (H / 7) * (W / 7), heads);
}
extern "C" int nnz_window_attention_backward_partial(const float* qkv, const float* bias_table, const int* bias_index,
                                                     const float* dout, float* dqkv, float* dpart, int B, int H, int W, int C,
                                                     int heads, int shift, float scale, int py, int px, void* stream) {
  if (!dpart) return NNZ_EINVAL;
  return wa_backward(qkv, bias_table, bias_index, dout, dqkv, nullptr, nullptr, nullptr, B, H, W, C, heads, shift, scale, py, px,
                     stream, dpart);
}
