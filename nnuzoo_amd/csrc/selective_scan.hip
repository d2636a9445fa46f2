// Selective scan (Mamba S6 recurrence) forward + backward as a wavefront-parallel chunk scan on gfx950.
// Replaces mamba_ssm's `selective_scan_cuda.fwd/bwd` behind `selective_scan_fn` as called by the reference's SS2D /
// SSND blocks (/root/reference/nnunetv2/nets/m2net.py:193-199, ssnd2net.py:271-277; semantics = the in-tree
// selective_scan_ref, nets/seg_mamba/selective_scan_interface.py:86-152): real A, B/C of shape (B, K, N, L), z = None,
//     dl = softplus(delta + bias);  h_t = exp(dl_t A) h_{t-1} + dl_t B_t u_t;  y_t = <h_t, C_t> + D u_t.
//
// Design (MI355X-first; HBM-bound, fp32, N = 16):
//   * Time on the lanes: a wave owns 256 consecutive steps of one (b, d) row (4 per lane), so u / delta / y move as
//     fully coalesced 16-byte-per-lane accesses; the affine pairs (a, b) are combined in-lane, then across the 64
//     lanes by a 6-step DPP scan (row_shr 1/2/4/8, row_bcast15, row_bcast31: no LDS traffic, no bpermute).
//   * L up to 262 144 with only a few hundred (b, d) rows: the sequence is cut into 256-step chunks that run in
//     parallel; pass 1 reduces every chunk to its affine summary, a tiny carry kernel scans the summaries, pass 3
//     replays the chunk from its true entry state.  The entry states are kept as the backward's checkpoints.
//   * B_t / C_t are shared by all D channels of a direction group: a workgroup stages the chunk's [16][256] B and C
//     tiles in LDS once and loops its 4 waves over the group's channels (the stock kernel re-reads them per channel).
//   * Backward: same structure run in reverse for g_t = C_t dy_t + a_{t+1} g_{t+1}; h is recomputed per chunk from
//     the checkpoint; dB / dC are reduced over the group's channels in LDS and leave the workgroup as one atomic
//     add per element; dA / dD / dbias by wave reduction + one atomic per (row, chunk).
#include "common.hpp"

namespace nnz {

constexpr int SS_N = 16;
constexpr int SS_KI = 4;
constexpr int SS_CL = 64 * SS_KI;  // chunk length
constexpr int SS_NW = 4;           // waves per workgroup

struct ScanArgs {
  const float* u;      // [B][KD][L]
  const float* delta;  // [B][KD][L]
  const float* A;      // [KD][N]
  const float* Bm;     // [B][K][N][L]
  const float* Cm;     // [B][K][N][L]
  const float* D;      // [KD] or null
  const float* bias;   // [KD] or null
  float* y;            // [B][KD][L]
  float* P;            // [rows][N][nchunks]   chunk summaries (multiplier)
  float* S;            // [rows][N][nchunks]   chunk summaries (offset)
  float* Hin;          // [rows][nchunks][N]   state entering each chunk (forward checkpoints)
  float* Gin;          // [rows][nchunks][N]   gradient state entering each chunk from the right
  const float* dy;     // [B][KD][L]
  float* du;
  float* ddelta;
  float* dA;           // [KD][N]   (atomic, zeroed by launcher)
  float* dB;           // [B][K][N][L] (atomic)
  float* dC;
  float* dD;           // [KD]
  float* dbias;        // [KD]
  int Bt, K, Dg, KD, L, nchunks;
  int rows_per_wg;     // multiple of SS_NW, divides Dg
  int softplus;
  // ---- cross-scan mode (XS; the SS2D block, K = 4): the four directions read ONE input through index arithmetic
  // instead of four materialised copies.  Direction k = s + 2j: source s (0: row-major tokens, 1: column-major
  // tokens, i.e. the transposed image), reversed in time when j = 1.  u / dy: [2][B][Dg][L] by source; the projections
  // P = [W_s ; W_{s+2}] x_s: [2][B][2 Cp][L], direction k's rows at j*Cp: R dt rows, N B rows, N C rows (Cp = R + 2N);
  // delta is formed in the kernel from the R dt rows and Wdt[k][d][R] (no [B][KD][L] delta / ddelta tensors);
  // y / du: [B][KD][L] per direction in the source's (un-reversed) token order; dP like P (accumulated).
  const float* xs_P;
  const float* xs_Wdt;  // [KD][R]
  int a_is_log;         // A holds A_log: the kernels use -exp(A_log) and return dA_log = dA * A (m2net.py:196)
  float* xs_dP;
  int R, Cp;
  // round 5: where several workgroups share one (batch, direction) group - channel chunks of a wide group - each writes its dB /
  // dC / d dt rows to a slab of its own (slab + part * slab_stride, laid out like xs_dP, or dB then dC in plain mode) and a fold
  // launch sums the slabs in part order: bit-identical from run to run, no zero fill, no float atomics (which run at 1.3 TB/s
  // on this chip - the slabs cost no more).  null: the atomics of rounds 1-4 (nnz_scan_tuning knob 4 = 0).
  float* slab;
  long slab_stride;
};

constexpr int SS_RMAX = 8;          // largest dt_rank of the cross-scan mode
constexpr int SS_DTP = SS_CL + 4;   // pitch of the staged dt tile: one column more than the chunk (dl of the next step)

// ---- DPP helpers -----------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                CTRL, ROW_MASK, 0xF, false));
}

// inclusive scan over the 64 lanes of the affine maps x -> a x + b (lower lanes applied first)
__device__ __forceinline__ void wave_scan_affine(float& a, float& b) {
#define NNZ_STEP(CTRL, MASK)                       \
  {                                                \
    const float pa = dpp<CTRL, MASK>(1.f, a);      \
    const float pb = dpp<CTRL, MASK>(0.f, b);      \
    b = a * pb + b;                                \
    a = a * pa;                                    \
  }
  NNZ_STEP(0x111, 0xF)  // row_shr:1
  NNZ_STEP(0x112, 0xF)  // row_shr:2
  NNZ_STEP(0x114, 0xF)  // row_shr:4
  NNZ_STEP(0x118, 0xF)  // row_shr:8
  NNZ_STEP(0x142, 0xA)  // row_bcast:15 -> rows 1, 3
  NNZ_STEP(0x143, 0xC)  // row_bcast:31 -> rows 2, 3
#undef NNZ_STEP
}

// inclusive SUFFIX scan (higher lanes applied first): lane l gets F_l o F_{l+1} o ... o F_63.
// DPP only (the first version used 12 ds_bpermute shuffles per call, whose LDS round trips were the dependent chain of the
// backward's state loop): row_shl 1/2/4/8 inside the 16-lane rows, then the rows' totals (lanes 16, 32, 48) are read
// with v_readlane and composed as wave-uniform values - there is no downward row broadcast in DPP.
__device__ __forceinline__ void wave_rscan_affine(float& a, float& b, int lane) {
#define NNZ_RSTEP(CTRL)                            \
  {                                                \
    const float pa = dpp<CTRL, 0xF>(1.f, a);       \
    const float pb = dpp<CTRL, 0xF>(0.f, b);       \
    b = a * pb + b;                                \
    a = a * pa;                                    \
  }
  NNZ_RSTEP(0x101)  // row_shl:1
  NNZ_RSTEP(0x102)  // row_shl:2
  NNZ_RSTEP(0x104)  // row_shl:4
  NNZ_RSTEP(0x108)  // row_shl:8
#undef NNZ_RSTEP
  const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 16));
  const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 16));
  const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 32));
  const float b2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 32));
  const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 48));
  const float b3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 48));
  // what follows each row: U2 = T3, U1 = T2 o T3, U0 = T1 o T2 o T3 (T_r = total of row r)
  const float u1a = a2 * a3, u1b = a2 * b3 + b2;
  const float u0a = a1 * u1a, u0b = a1 * u1b + b1;
  const int row = lane >> 4;
  const float ua = row == 0 ? u0a : row == 1 ? u1a : row == 2 ? a3 : 1.f;
  const float ub = row == 0 ? u0b : row == 1 ? u1b : row == 2 ? b3 : 0.f;
  b = a * ub + b;
  a = a * ua;
}

// value of the previous / next lane (wave_shr:1 / wave_shl:1); lane 0 / lane 63 receive `edge`
__device__ __forceinline__ float lane_prev(float v, float edge) { return dpp<0x138, 0xF>(edge, v); }
__device__ __forceinline__ float lane_next(float v, float edge) { return dpp<0x130, 0xF>(edge, v); }

// wave-wide sum by DPP (row_shr 1/2/4/8 + row_bcast15/31): the total arrives in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp<0x111, 0xF>(0.f, v);
  v += dpp<0x112, 0xF>(0.f, v);
  v += dpp<0x114, 0xF>(0.f, v);
  v += dpp<0x118, 0xF>(0.f, v);
  v += dpp<0x142, 0xA>(0.f, v);
  v += dpp<0x143, 0xC>(0.f, v);
  return v;
}

// softplus(x) = max(x, 0) + log1p(exp(-|x|)); log1p by its series for tiny arguments (keeps relative accuracy
// where exp(-|x|) << 1) and by the fast log otherwise.  The libm log1pf expands to ~100 instructions on gfx950.
__device__ __forceinline__ float softplus_f(float x) {
  const float e = __expf(-fabsf(x));
  const float l = e < 1.0e-3f ? e * (1.f - e * (0.5f - 0.33333334f * e)) : __logf(1.f + e);
  return fmaxf(x, 0.f) + l;
}

// load 4 consecutive elements of a row starting at t (vector when the row is 16-byte aligned and fully in range)
__device__ __forceinline__ void load4(const float* row, int t, int L, bool vec, float (&v)[SS_KI]) {
  if (vec && t + SS_KI <= L) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + t);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  } else {
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) v[i] = (t + i < L) ? row[t + i] : 0.f;
  }
}
__device__ __forceinline__ void store4(float* row, int t, int L, bool vec, const float (&v)[SS_KI]) {
  if (vec && t + SS_KI <= L) {
    f32x4 x = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(row + t) = x;
  } else {
#pragma unroll
    for (int i = 0; i < SS_KI; ++i)
      if (t + i < L) row[t + i] = v[i];
  }
}

// time-reversed rows: logical step t lives at memory position L-1-t, so the 4 steps of a lane are 4 consecutive
// elements read back to front (still one 16-byte access when L % 4 == 0)
__device__ __forceinline__ void load4x(const float* row, int t, int L, bool vec, bool REV, float (&v)[SS_KI]) {
  if (!REV) {
    load4(row, t, L, vec, v);
  } else if (vec && t + SS_KI <= L) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + (L - t - SS_KI));
    v[0] = x[3]; v[1] = x[2]; v[2] = x[1]; v[3] = x[0];
  } else {
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) v[i] = (t + i < L) ? row[L - 1 - t - i] : 0.f;
  }
}
__device__ __forceinline__ void store4x(float* row, int t, int L, bool vec, bool REV, const float (&v)[SS_KI]) {
  if (!REV) {
    store4(row, t, L, vec, v);
  } else if (vec && t + SS_KI <= L) {
    f32x4 x = {v[3], v[2], v[1], v[0]};
    *reinterpret_cast<f32x4*>(row + (L - t - SS_KI)) = x;
  } else {
#pragma unroll
    for (int i = 0; i < SS_KI; ++i)
      if (t + i < L) row[L - 1 - t - i] = v[i];
  }
}
// [rows][ncols] tile of rows with stride L starting at logical step t0, time-reversed or not, into dst[rows][pitch]
__device__ __forceinline__ void stage_rows(const float* src, float* dst, int rows, int ncols, int pitch, int t0, int L,
                                           bool REV) {
  for (int i = threadIdx.x; i < rows * ncols; i += SS_NW * 64) {
    const int n = i / ncols, tt = i % ncols;
    const int t = t0 + tt;
    dst[n * pitch + tt] = t < L ? src[(long)n * L + (REV ? L - 1 - t : t)] : 0.f;
  }
}

// stage the [N][CL] tile of one (b, k) group for chunk c into LDS (zero beyond L)
__device__ __forceinline__ void stage_tile(const float* src /* [N][L] */, float* dst /* [N][CL] */, int t0, int L) {
  for (int i = threadIdx.x; i < SS_N * SS_CL; i += SS_NW * 64) {
    const int n = i / SS_CL, tt = i % SS_CL;
    const int t = t0 + tt;
    dst[i] = t < L ? src[(long)n * L + t] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// per-row inputs of one wave, prefetched one row ahead (the loads of row r+1 fly under the n-loop of row r)
// ---------------------------------------------------------------------------------------------------------------
struct RowIn {
  float u[SS_KI], dl[SS_KI], dy[SS_KI];
  float dl_next;  // raw delta of the step after the lane's last item (backward only; lane 63: next chunk's first)
  float aux;      // lanes 0-15: A[kd][n]; 16-31: Hin[n]; 32-47: Gin[n]  (one coalesced load per wave)
  float bias, Dv;
};

template <bool BWD, bool FINAL, bool XS>
__device__ __forceinline__ void load_row(const ScanArgs& a, int b, int k, int r, int c, int t, int lane, bool vec,
                                         RowIn& x) {
  const int kd = k * a.Dg + r;
  const long row = (long)b * a.KD + kd;
  if (!XS) {
    load4(a.delta + row * a.L, t, a.L, vec, x.dl);
    if (!BWD || FINAL) load4(a.u + row * a.L, t, a.L, vec, x.u);
    if (BWD) {
      load4(a.dy + row * a.L, t, a.L, vec, x.dy);
      const int tn = t + SS_KI;
      x.dl_next = tn < a.L ? a.delta[row * a.L + tn] : 0.f;
    }
  } else {
    // shared input of the four directions: source k & 1, reversed for k >= 2; delta comes from the staged dt tile
    const bool rev = k >= 2;
    const long srow = ((long)(k & 1) * a.Bt + b) * a.Dg + r;
    if (!BWD || FINAL) load4x(a.u + srow * a.L, t, a.L, vec, rev, x.u);
    if (BWD) load4x(a.dy + srow * a.L, t, a.L, vec, rev, x.dy);
  }
  const int n = lane & 15;
  float v = 0.f;
  if (lane < 16) {
    v = a.A[(long)kd * SS_N + n];
    if (XS && a.a_is_log) v = -__expf(v);
  }
  else if (lane < 32) { if (FINAL) v = a.Hin[(row * a.nchunks + c) * SS_N + n]; }
  else if (lane < 48) { if (BWD && FINAL) v = a.Gin[(row * a.nchunks + c) * SS_N + n]; }
  else if (XS && lane - 48 < a.R) v = a.xs_Wdt[(long)kd * a.R + (lane - 48)];
  x.aux = v;
  x.bias = a.bias ? a.bias[kd] : 0.f;
  x.Dv = a.D ? a.D[kd] : 0.f;
}

// cross-scan: raw delta of the lane's 4 steps (and of the step after them) = Wdt[k][d][:] . dt[:, t] from the LDS tile
__device__ __forceinline__ void xs_delta(const float* sDt, const float* wrow /* LDS: Wdt[kd][0..R) */, int R, int lane,
                                         float (&dl)[SS_KI], float& dl_next) {
#pragma unroll
  for (int i = 0; i < SS_KI; ++i) dl[i] = 0.f;
  dl_next = 0.f;
  for (int r = 0; r < R; ++r) {
    const float w = wrow[r];
    const f32x4 dv = *reinterpret_cast<const f32x4*>(sDt + r * SS_DTP + lane * SS_KI);
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) dl[i] += w * dv[i];
    dl_next += w * sDt[r * SS_DTP + lane * SS_KI + SS_KI];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward: FINAL = false -> chunk summaries (P, S); FINAL = true -> y from the entry state Hin
// ---------------------------------------------------------------------------------------------------------------
template <bool FINAL, bool XS>
__global__ __launch_bounds__(SS_NW * 64) void scan_fwd_kernel(ScanArgs a) {
  __shared__ __attribute__((aligned(16))) float sB[SS_N * SS_CL];
  __shared__ __attribute__((aligned(16))) float sC[FINAL ? SS_N * SS_CL : 4];
  __shared__ __attribute__((aligned(16))) float sDt[XS ? SS_RMAX * SS_DTP : 4];
  __shared__ float sw[SS_NW][64];  // per wave: A row, entry state, (XS) Wdt row
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x;                  // chunk
  const int wgs_per_group = a.Dg / a.rows_per_wg;
  const int grp = blockIdx.y / wgs_per_group;  // b*K + k
  const int sub = blockIdx.y % wgs_per_group;
  const int b = grp / a.K, k = grp % a.K;
  const int t0 = c * SS_CL;
  const int t = t0 + lane * SS_KI;
  const bool vec = (a.L & 3) == 0;
  const int r_end = (sub + 1) * a.rows_per_wg;

  const bool rev = XS && k >= 2;

  RowIn cur, nxt;
  int r = sub * a.rows_per_wg + wave;
  load_row<false, FINAL, XS>(a, b, k, r, c, t, lane, vec, cur);
  if (!XS) {
    stage_tile(a.Bm + (long)grp * SS_N * a.L, sB, t0, a.L);
    if (FINAL) stage_tile(a.Cm + (long)grp * SS_N * a.L, sC, t0, a.L);
  } else {
    const float* Pk = a.xs_P + ((((long)(k & 1) * a.Bt + b) * 2 + (k >> 1)) * a.Cp) * a.L;
    stage_rows(Pk, sDt, a.R, SS_CL + 1, SS_DTP, t0, a.L, rev);
    stage_rows(Pk + (long)a.R * a.L, sB, SS_N, SS_CL, SS_CL, t0, a.L, rev);
    if (FINAL) stage_rows(Pk + (long)(a.R + SS_N) * a.L, sC, SS_N, SS_CL, SS_CL, t0, a.L, rev);
  }
  __syncthreads();

  for (; r < r_end; r += SS_NW) {
    const int kd = k * a.Dg + r;
    const long row = (long)b * a.KD + kd;
    if (r + SS_NW < r_end) load_row<false, FINAL, XS>(a, b, k, r + SS_NW, c, t, lane, vec, nxt);
    sw[wave][lane] = cur.aux;  // same-wave LDS: program order is enough
    if (XS) {
      float unused;
      xs_delta(sDt, &sw[wave][48], a.R, lane, cur.dl, unused);
    }
    float u[SS_KI], dl[SS_KI], yv[SS_KI];
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) {
      float d = cur.dl[i] + cur.bias;
      if (a.softplus) d = softplus_f(d);
      dl[i] = (t + i < a.L) ? d : 0.f;  // identity step beyond L
      u[i] = cur.u[i];
      yv[i] = cur.Dv * u[i];
    }
#pragma unroll 4
    for (int n = 0; n < SS_N; ++n) {
      const float An = sw[wave][n];
      const f32x4 Bv = *reinterpret_cast<const f32x4*>(sB + n * SS_CL + lane * SS_KI);
      float ai[SS_KI], bi[SS_KI];
#pragma unroll
      for (int i = 0; i < SS_KI; ++i) {
        ai[i] = __expf(dl[i] * An);
        bi[i] = dl[i] * u[i] * Bv[i];
      }
      float pa = ai[0], pb = bi[0];
#pragma unroll
      for (int i = 1; i < SS_KI; ++i) {
        pb = ai[i] * pb + bi[i];
        pa *= ai[i];
      }
      wave_scan_affine(pa, pb);
      if (!FINAL) {
        if (lane == 63) {
          a.P[(row * SS_N + n) * a.nchunks + c] = pa;
          a.S[(row * SS_N + n) * a.nchunks + c] = pb;
        }
      } else {
        const float ea = lane_prev(pa, 1.f), eb = lane_prev(pb, 0.f);  // lane 0: identity
        float h = ea * sw[wave][16 + n] + eb;
        const f32x4 Cv = *reinterpret_cast<const f32x4*>(sC + n * SS_CL + lane * SS_KI);
#pragma unroll
        for (int i = 0; i < SS_KI; ++i) {
          h = ai[i] * h + bi[i];
          yv[i] += Cv[i] * h;
        }
      }
    }
    if (FINAL) store4x(a.y + row * a.L, t, a.L, vec, rev, yv);
    cur = nxt;
  }
}

// carry over the chunk summaries.  One wave per (row, n); lanes split the chunks.
//   forward : X[0] = 0, X[c+1] = P[c] X[c] + S[c]          -> out[row][c][n] = X[c]
//   reverse : X[nch-1] = 0, X[c-1] = P[c] X[c] + S[c]      -> out[row][c][n] = X[c]
template <bool REVERSE>
__global__ __launch_bounds__(256) void scan_carry_kernel(const float* __restrict__ P, const float* __restrict__ S,
                                                         float* __restrict__ out, long rows_n, int nchunks) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (row * N + n)
  if (w >= rows_n) return;
  const long row = w / SS_N;
  const int n = w % SS_N;
  const float* p = P + w * nchunks;
  const float* s = S + w * nchunks;
  const int cpl = (nchunks + 63) / 64;
  // position j = 0 .. nchunks-1 in processing order; chunk index c = REVERSE ? nchunks-1-j : j
  const int j0 = lane * cpl;
  float pa = 1.f, pb = 0.f;
  for (int j = j0; j < j0 + cpl && j < nchunks; ++j) {
    const int c = REVERSE ? nchunks - 1 - j : j;
    pb = p[c] * pb + s[c];
    pa = p[c] * pa;
  }
  wave_scan_affine(pa, pb);
  float eb = __shfl_up(pb, 1, 64);
  if (lane == 0) eb = 0.f;
  float x = eb;  // X at processing position j0 (entry value is 0, so only the offset part matters)
  for (int j = j0; j < j0 + cpl && j < nchunks; ++j) {
    const int c = REVERSE ? nchunks - 1 - j : j;
    out[(row * nchunks + c) * SS_N + n] = x;
    x = p[c] * x + s[c];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: FINAL = false -> reverse chunk summaries; FINAL = true -> all gradients
// reverse recurrence  g_t = a_{t+1} g_{t+1} + C_t dy_t
// ---------------------------------------------------------------------------------------------------------------
template <bool FINAL, bool XS>
__global__ __launch_bounds__(SS_NW * 64) void scan_bwd_kernel(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  float* sC = dyn_lds;                                   // [N][CL]
  float* sw = sC + SS_N * SS_CL;                         // [NW][64] per wave: A row, Hin, Gin, (XS) Wdt row
  float* sB = sw + SS_NW * 64;                           // FINAL only: [N][CL]
  float* sdB = sB + SS_N * SS_CL;                        // FINAL only
  float* sdC = sdB + SS_N * SS_CL;                       // FINAL only
  float* sDt = FINAL ? sdC + SS_N * SS_CL : sB;          // XS only: [R][DTP] staged dt rows
  float* sdDt = sDt + a.R * SS_DTP;                      // XS && FINAL only: [R][CL] gradient of the dt rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x;
  const int wgs_per_group = a.Dg / a.rows_per_wg;
  const int grp = blockIdx.y / wgs_per_group;
  const int sub = blockIdx.y % wgs_per_group;
  const int b = grp / a.K, k = grp % a.K;
  const int t0 = c * SS_CL;
  const int t = t0 + lane * SS_KI;
  const bool vec = (a.L & 3) == 0;
  const int r_end = (sub + 1) * a.rows_per_wg;
  float* swv = sw + wave * 64;

  const bool rev = XS && k >= 2;
  const long pk_off = XS ? ((((long)(k & 1) * a.Bt + b) * 2 + (k >> 1)) * a.Cp) * a.L : 0;

  RowIn cur, nxt;
  int r = sub * a.rows_per_wg + wave;
  load_row<true, FINAL, XS>(a, b, k, r, c, t, lane, vec, cur);
  if (!XS) {
    stage_tile(a.Cm + (long)grp * SS_N * a.L, sC, t0, a.L);
    if (FINAL) stage_tile(a.Bm + (long)grp * SS_N * a.L, sB, t0, a.L);
  } else {
    const float* Pk = a.xs_P + pk_off;
    stage_rows(Pk, sDt, a.R, SS_CL + 1, SS_DTP, t0, a.L, rev);
    stage_rows(Pk + (long)(a.R + SS_N) * a.L, sC, SS_N, SS_CL, SS_CL, t0, a.L, rev);
    if (FINAL) stage_rows(Pk + (long)a.R * a.L, sB, SS_N, SS_CL, SS_CL, t0, a.L, rev);
  }
  if (FINAL) {
    for (int i = threadIdx.x; i < SS_N * SS_CL; i += SS_NW * 64) {
      sdB[i] = 0.f;
      sdC[i] = 0.f;
    }
  }
  __syncthreads();
  // cross-scan: d dt[r][t] = sum_d Wdt[kd][r] dd_d[t] summed over the wave's channels in registers (R <= 8 rows x 4 steps),
  // folded over the four waves once per workgroup (ds_add_f32 per row and lane was the slow part of this mode)
  float accDt[XS && FINAL ? SS_RMAX : 1][SS_KI];
#pragma unroll
  for (int rr = 0; rr < (XS && FINAL ? SS_RMAX : 1); ++rr)
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) accDt[rr][i] = 0.f;

  for (; r < r_end; r += SS_NW) {
    const int kd = k * a.Dg + r;
    const long row = (long)b * a.KD + kd;
    if (r + SS_NW < r_end) load_row<true, FINAL, XS>(a, b, k, r + SS_NW, c, t, lane, vec, nxt);
    swv[lane] = cur.aux;  // same-wave LDS: program order is enough
    if (XS) xs_delta(sDt, swv + 48, a.R, lane, cur.dl, cur.dl_next);
    float u[SS_KI], draw[SS_KI], dl[SS_KI], dyv[SS_KI];
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) {
      float d = cur.dl[i] + cur.bias;
      draw[i] = d;
      if (a.softplus) d = softplus_f(d);
      dl[i] = (t + i < a.L) ? d : 0.f;
      dyv[i] = cur.dy[i];
      u[i] = FINAL ? cur.u[i] : 0.f;
    }
    // dl of the step right after this lane's items (first item of lane+1; for lane 63 the next chunk's first step)
    float dl_next = 0.f;
    if (t + SS_KI < a.L) {
      dl_next = cur.dl_next + cur.bias;
      if (a.softplus) dl_next = softplus_f(dl_next);
    }
    float duv[SS_KI], ddl[SS_KI];
#pragma unroll
    for (int i = 0; i < SS_KI; ++i) {
      duv[i] = cur.Dv * dyv[i];
      ddl[i] = 0.f;
    }
    const float* Arow = swv;       // LDS broadcasts: nothing inside the n-loop waits on vector memory
    const float* hin = swv + 16;
    const float* gin = swv + 32;
    // The 4 waves of the workgroup walk the states in a skewed order (wave w starts at n = 4w) and meet at a
    // barrier after every 4th state: at any moment they update DIFFERENT rows of the dB/dC accumulation tiles, so the
    // cross-channel reduction is a plain LDS read-modify-write (ds_add_f32 atomics measured ~4x slower in total).
    for (int j = 0; j < SS_N; ++j) {
      const int n = FINAL ? ((j + 4 * wave) & (SS_N - 1)) : j;
      const float An = Arow[n];
      const f32x4 Cv = *reinterpret_cast<const f32x4*>(sC + n * SS_CL + lane * SS_KI);
      float ai[SS_KI + 1];
#pragma unroll
      for (int i = 0; i < SS_KI; ++i) ai[i] = __expf(dl[i] * An);
      ai[SS_KI] = __expf(dl_next * An);
      // F(g) = g of this lane's first item as a function of the g entering after its last item:
      //   G_i(g) = ai[i+1] * g + C_i dy_i,  F = G_0 o G_1 o ... o G_{K-1}, built from the inside (i = K-1) outwards
      float ra, rb;
      {
        float fa = 1.f, fb = 0.f;  // identity, will become F
#pragma unroll
        for (int i = SS_KI - 1; i >= 0; --i) {
          // F_new(g) = G_i(F_old(g))
          fb = ai[i + 1] * fb + Cv[i] * dyv[i];
          fa = ai[i + 1] * fa;
        }
        ra = fa;
        rb = fb;
      }
      if (!FINAL) {
        wave_rscan_affine(ra, rb, lane);
        if (lane == 0) {
          a.P[(row * SS_N + n) * a.nchunks + c] = ra;
          a.S[(row * SS_N + n) * a.nchunks + c] = rb;
        }
        continue;
      }
      // ---- forward recompute of h over the chunk ------------------------------------------------------------
      const f32x4 Bv = *reinterpret_cast<const f32x4*>(sB + n * SS_CL + lane * SS_KI);
      float bi[SS_KI];
#pragma unroll
      for (int i = 0; i < SS_KI; ++i) bi[i] = dl[i] * u[i] * Bv[i];
      float pa = ai[0], pb = bi[0];
#pragma unroll
      for (int i = 1; i < SS_KI; ++i) {
        pb = ai[i] * pb + bi[i];
        pa *= ai[i];
      }
      wave_scan_affine(pa, pb);
      const float ea = lane_prev(pa, 1.f), eb = lane_prev(pb, 0.f);  // lane 0: identity
      float hprev[SS_KI], hcur[SS_KI];
      float h = ea * hin[n] + eb;
#pragma unroll
      for (int i = 0; i < SS_KI; ++i) {
        hprev[i] = h;
        h = ai[i] * h + bi[i];
        hcur[i] = h;
      }
      // ---- reverse scan of g -------------------------------------------------------------------------------
      wave_rscan_affine(ra, rb, lane);
      const float xa = lane_next(ra, 1.f), xb = lane_next(rb, 0.f);  // lane 63: identity
      float g = xa * gin[n] + xb;  // g of the first item of lane+1 (or of the next chunk)
      float dAn = 0.f;
      float dBv[SS_KI], dCv[SS_KI];
#pragma unroll
      for (int i = SS_KI - 1; i >= 0; --i) {
        g = ai[i + 1] * g + Cv[i] * dyv[i];  // g_i
        dCv[i] = dyv[i] * hcur[i];
        dBv[i] = g * dl[i] * u[i];
        duv[i] += g * dl[i] * Bv[i];
        const float gah = g * ai[i] * hprev[i];
        ddl[i] += g * Bv[i] * u[i] + An * gah;
        dAn += dl[i] * gah;
      }
      {
        f32x4* pB = reinterpret_cast<f32x4*>(sdB + n * SS_CL + lane * SS_KI);
        f32x4* pC = reinterpret_cast<f32x4*>(sdC + n * SS_CL + lane * SS_KI);
        f32x4 vB = *pB, vC = *pC;
#pragma unroll
        for (int i = 0; i < SS_KI; ++i) {
          vB[i] += dBv[i];
          vC[i] += dCv[i];
        }
        *pB = vB;
        *pC = vC;
      }
      // per-(row, chunk) partial of dA goes to the (now free) summary workspace with a plain store; a finalize
      // kernel sums over chunks and batch: no contended atomic (and no vector-memory wait) inside this loop
      dAn = wave_sum_to_lane63(dAn);
      if (lane == 63) a.P[(row * SS_N + n) * a.nchunks + c] = dAn;
      // keeps the waves' skewed n-order disjoint (all waves run the same trip counts).  A barrier every 4th state is
      // enough: between two barriers wave w touches rows 4(k + w) .. 4(k + w) + 3 (mod 16), disjoint for the four waves
      if ((j & 3) == 3) __syncthreads();
    }
    if (FINAL) {
      float sdb = 0.f, sdD = 0.f;
      float dd[SS_KI];
#pragma unroll
      for (int i = 0; i < SS_KI; ++i) {
        float gsp = 1.f;
        if (a.softplus) gsp = draw[i] > 20.f ? 1.f : 1.f / (1.f + __expf(-draw[i]));
        dd[i] = (t + i < a.L) ? ddl[i] * gsp : 0.f;
        sdb += dd[i];
        sdD += dyv[i] * u[i];
      }
      store4x(a.du + row * a.L, t, a.L, vec, rev, duv);
      if (!XS) {
        store4(a.ddelta + row * a.L, t, a.L, vec, dd);
      } else {
        // delta = Wdt[kd][:] . dt[:, t]:  dWdt[kd][r] = sum_t dd_t dt[r][t] (per-(row, chunk) partial, summed by the
        // finalize kernel);  d dt[r][t] += Wdt[kd][r] dd_t, summed over this wave's channels in registers
#pragma unroll
        for (int rr = 0; rr < SS_RMAX; ++rr) {
          if (rr < a.R) {  // wave-uniform
            const f32x4 dv = *reinterpret_cast<const f32x4*>(sDt + rr * SS_DTP + lane * SS_KI);
            float sw_ = 0.f;
            const float w = swv[48 + rr];
#pragma unroll
            for (int i = 0; i < SS_KI; ++i) {
              sw_ += dd[i] * dv[i];
              accDt[rr][i] += w * dd[i];
            }
            sw_ = wave_sum_to_lane63(sw_);
            if (lane == 63) a.S[(row * SS_N + 2 + rr) * a.nchunks + c] = sw_;
          }
        }
      }
      sdb = wave_sum(sdb);
      sdD = wave_sum(sdD);
      if (lane == 0) {
        a.S[(row * SS_N + 0) * a.nchunks + c] = sdb;
        a.S[(row * SS_N + 1) * a.nchunks + c] = sdD;
      }
    }
    cur = nxt;
  }
  if (FINAL) {
    __syncthreads();
    // (slab mode: this workgroup's own copy of the tile - plain stores, folded afterwards)
    float* dPbase = a.slab ? a.slab + (long)sub * a.slab_stride : a.xs_dP;
    float* dBbase = a.slab ? a.slab + (long)sub * a.slab_stride : a.dB;
    float* dCbase = a.slab ? dBbase + (long)a.Bt * a.K * SS_N * a.L : a.dC;
    float* gB = XS ? dPbase + pk_off + (long)a.R * a.L : dBbase + (long)grp * SS_N * a.L;
    float* gC = XS ? dPbase + pk_off + (long)(a.R + SS_N) * a.L : dCbase + (long)grp * SS_N * a.L;
    const bool sole = wgs_per_group == 1 || a.slab != nullptr;  // the only writer of the tile: plain stores
    for (int i = threadIdx.x; i < SS_N * SS_CL; i += SS_NW * 64) {
      const int n = i / SS_CL, tt = i % SS_CL;
      const int li = i;
      if (t0 + tt < a.L) {
        const long m = (long)n * a.L + (rev ? a.L - 1 - (t0 + tt) : t0 + tt);
        if (sole) {
          gB[m] = sdB[li];
          gC[m] = sdC[li];
        } else {
          atomicAdd(gB + m, sdB[li]);
          atomicAdd(gC + m, sdC[li]);
        }
      }
    }
    if (XS) {
      // fold the four waves' register sums into the LDS tile: wave 0 stores, waves 1..3 add in turn
      for (int w = 0; w < SS_NW; ++w) {
        if (wave == w) {
#pragma unroll
          for (int rr = 0; rr < SS_RMAX; ++rr) {
            if (rr < a.R) {
              f32x4* pT = reinterpret_cast<f32x4*>(sdDt + rr * SS_CL + lane * SS_KI);
              f32x4 v = {accDt[rr][0], accDt[rr][1], accDt[rr][2], accDt[rr][3]};
              if (w > 0) v += *pT;
              *pT = v;
            }
          }
        }
        __syncthreads();
      }
      float* gT = dPbase + pk_off;
      for (int i = threadIdx.x; i < a.R * SS_CL; i += SS_NW * 64) {
        const int rr = i / SS_CL, tt = i % SS_CL;
        if (t0 + tt < a.L) {
          const long m = (long)rr * a.L + (rev ? a.L - 1 - (t0 + tt) : t0 + tt);
          if (sole) gT[m] = sdDt[i];
          else atomicAdd(gT + m, sdDt[i]);
        }
      }
    }
  }
}

// dA[kd][n] = sum_{b, c} P[(b*KD + kd)*N + n][c];  dbias / dD from rows 0 / 1 of S.  One wave per (kd, n).
__global__ __launch_bounds__(256) void scan_bwd_finalize_kernel(const float* __restrict__ P, const float* __restrict__ S,
                                                                float* __restrict__ dA, float* __restrict__ dbias,
                                                                float* __restrict__ dD, int Bt, int KD, int nchunks,
                                                                float* __restrict__ dWdt, int R,
                                                                const float* __restrict__ Alog) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)KD * SS_N) return;
  const int kd = w / SS_N, n = w % SS_N;
  float sa = 0.f, s0 = 0.f;
  for (int b = 0; b < Bt; ++b) {
    const long base = (((long)b * KD + kd) * SS_N + n) * nchunks;
    for (int c = lane; c < nchunks; c += 64) {
      sa += P[base + c];
      if (n < 2 + R) s0 += S[base + c];
    }
  }
  sa = wave_sum(sa);
  s0 = wave_sum(s0);
  if (lane == 0) {
    dA[w] = Alog ? sa * -__expf(Alog[w]) : sa;
    if (n == 0 && dbias) dbias[kd] = s0;
    if (n == 1 && dD) dD[kd] = s0;
    if (n >= 2 && n < 2 + R && dWdt) dWdt[(long)kd * R + (n - 2)] = s0;   // cross-scan mode: rows 2 .. 2+R of S
  }
}

constexpr int SS_BWD_LDS_SUMMARY = (SS_N * SS_CL + SS_NW * 64) * 4;
constexpr int SS_BWD_LDS_FINAL = (4 * SS_N * SS_CL + SS_NW * 64) * 4;

static int pick_rows_per_wg(int Dg, long groups_chunks) {
  // whole group per workgroup when there are plenty of (group, chunk) pairs, else split the group's channels
  int r = Dg;
  while (r > SS_NW && groups_chunks * (Dg / r) < 1024 && (r / 2) % SS_NW == 0) r /= 2;
  return r;
}

static int check(const ScanArgs& a) {
  if (a.Bt < 1 || a.K < 1 || a.Dg < SS_NW || a.Dg % SS_NW || a.KD != a.K * a.Dg || a.L < 1) return NNZ_EINVAL;
  return NNZ_OK;
}

}  // namespace nnz

#include "ss2d_scan_rl.hpp"

namespace nnz {

// Tuning knobs of the cross-scan (diagnostics / A-B runs; nnz_scan_tuning(knob, value)):
//   0  channels-on-lanes kernels (ss2d_scan_rl.hpp) for the cross-scan forward AND backward   (default 1)
//   1  forced sub-blocks per chunk (4 / 8 / .. / 64 = chunks of 64 .. 1024 steps), 0 = by size   (default 0)
//   2  smallest problem (batch x 4 Dg x L row-steps) that takes the channels-on-lanes kernels   (default 2 M)
//   3  (read-only use) number of channels-on-lanes launches so far: lets a test assert which generation ran
//   4  deterministic slabs + fold for the shared dB / dC / d dt tiles (ScanArgs::slab) instead of fp32 atomics   (default 1)
static int g_scan_tuning[5] = {1, 0, 2 << 20, 0, 1};

constexpr int RL_MIN_CLB = 4;   // shortest chunk: 64 steps (sizes of the state / workspace buffers assume it)

// sub-blocks per chunk: the longest chunk (<= 1024 steps) that divides L and still gives every SIMD of the chip (1024)
// two waves; 0 = shape not supported by the channels-on-lanes kernels
static int rl_pick_clb(const ScanArgs& a) {
  if (a.L % (RL_MIN_CLB * RL_T) != 0 || !(a.Dg == 32 || a.Dg % 64 == 0) || a.R > SS_RMAX) return 0;
  // small problems (fewer than ~2 M row-steps: 32x32 tokens at 128 channels and below) leave most SIMDs without a wave in
  // this mapping; the time-on-lanes kernels split them finer (tools/probes/scan_threshold_probe.py: break-even between 1 M and 2 M)
  if ((long)a.Bt * a.KD * a.L < (long)g_scan_tuning[2]) return 0;
  const int forced = g_scan_tuning[1];
  if (forced >= RL_MIN_CLB && (forced & (forced - 1)) == 0 && a.L % (forced * RL_T) == 0) return forced;
  const long lanes = (long)a.Bt * a.KD;
  int best = RL_MIN_CLB;
  for (int clb = RL_MIN_CLB; clb <= 64; clb *= 2) {
    if (a.L % (clb * RL_T) != 0) break;
    if (clb > RL_MIN_CLB && lanes * (a.L / (clb * RL_T)) / 64 < 2048) break;
    best = clb;
  }
  return best;
}
static bool rl_ok(const ScanArgs& a) { return g_scan_tuning[0] != 0 && check(a) == NNZ_OK && rl_pick_clb(a) != 0; }
static long rl_nch_max(int L) { return (L + RL_MIN_CLB * RL_T - 1) / (RL_MIN_CLB * RL_T); }

static void rl_setup(ScanArgs& a, int& clb, dim3& grid, float* chunk_state, float* workspace, float*& Hck) {
  clb = rl_pick_clb(a);
  a.nchunks = (a.L + clb * RL_T - 1) / (clb * RL_T);
  const long rows = (long)a.Bt * a.KD;
  a.P = workspace;
  a.S = workspace + rows * SS_N * a.nchunks;
  a.Hin = chunk_state;
  Hck = chunk_state + rows * SS_N * rl_nch_max(a.L);
  const int Dl = a.Dg < 64 ? a.Dg : 64;
  const int slots = 64 / Dl;
  grid = dim3((a.nchunks + slots - 1) / slots, a.Bt * a.K * (a.Dg / Dl));
}

// floats of P + S at the front of the workspace (both generations); the slabs of the deterministic mode follow
static long ws_ps_floats(int Bt, int KD, int L) { return 2L * Bt * KD * SS_N * rl_nch_max(L); }
// one slab = one copy of what the backward accumulates across workgroups: dP (cross-scan) or dB | dC (plain)
static long slab_floats(const ScanArgs& a, bool xs) {
  return xs ? 4L * a.Bt * a.Cp * a.L : 2L * a.Bt * a.K * SS_N * a.L;
}
// point the kernel at `parts` slabs behind P / S; returns false (atomics) when the knob is off
static bool slab_setup(ScanArgs& a, bool xs, int parts, float* workspace) {
  if (!g_scan_tuning[4] || parts < 2) return false;
  a.slab = workspace + ws_ps_floats(a.Bt, a.KD, a.L);
  a.slab_stride = slab_floats(a, xs);
  return true;
}
static hipError_t slab_fold(const ScanArgs& a, bool xs, int parts, hipStream_t s) {
  if (xs) return fold_partials(a.slab, parts, a.slab_stride, a.slab_stride, a.xs_dP, s);
  const long nB = (long)a.Bt * a.K * SS_N * a.L;
  hipError_t e = fold_partials(a.slab, parts, a.slab_stride, nB, a.dB, s);
  if (e != hipSuccess) return e;
  return fold_partials(a.slab + nB, parts, a.slab_stride, nB, a.dC, s);
}

template <bool XS>
static int rl_forward(ScanArgs& a, float* chunk_state, float* workspace, hipStream_t s) {
  ++g_scan_tuning[3];
  int clb;
  dim3 grid;
  float* Hck;
  rl_setup(a, clb, grid, chunk_state, workspace, Hck);
  const long rows = (long)a.Bt * a.KD;
  if (a.nchunks > 1) {
    NNZ_LAUNCH(xs_rl_fwd_summary_kernel<XS>, grid, dim3(64), 0, s, a, clb);
    NNZ_LAUNCH_CHECK();
    const long rows_n = rows * SS_N;
    NNZ_LAUNCH(scan_carry_kernel<false>, dim3((unsigned)((rows_n + 3) / 4)), dim3(256), 0, s, a.P, a.S, a.Hin, rows_n,
               a.nchunks);
    NNZ_LAUNCH_CHECK();
  } else {
    hipError_t e = nnz::zero_async(a.Hin, sizeof(float) * rows * SS_N, s);
    if (e != hipSuccess) return (int)e;
  }
  NNZ_LAUNCH(xs_rl_fwd_final_kernel<XS>, grid, dim3(64), 0, s, a, clb, Hck);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <bool XS>
static int rl_backward(ScanArgs& a, const float* chunk_state, float* grad_state, float* workspace, float* dWdt,
                       hipStream_t s) {
  ++g_scan_tuning[3];
  int clb;
  dim3 grid;
  float* Hck;
  rl_setup(a, clb, grid, const_cast<float*>(chunk_state), workspace, Hck);
  a.Gin = grad_state;
  const long rows = (long)a.Bt * a.KD;
  const bool shared_dp = a.Dg > 64;     // several waves (channel groups of 64) write one dP tile
  const bool slabs = shared_dp && slab_setup(a, XS, a.Dg / 64, workspace);
  const int atomic_dp = shared_dp && !slabs;
  hipError_t e;
  if (atomic_dp) {
    if (XS) {
      if ((e = nnz::zero_async(a.xs_dP, sizeof(float) * 2L * a.Bt * 2 * a.Cp * a.L, s)) != hipSuccess) return (int)e;
    } else {
      if ((e = nnz::zero_async(a.dB, sizeof(float) * (long)a.Bt * a.K * SS_N * a.L, s)) != hipSuccess) return (int)e;
      if ((e = nnz::zero_async(a.dC, sizeof(float) * (long)a.Bt * a.K * SS_N * a.L, s)) != hipSuccess) return (int)e;
    }
  }
  if (a.nchunks > 1) {
    NNZ_LAUNCH(xs_rl_bwd_summary_kernel<XS>, grid, dim3(64), 0, s, a, clb);
    NNZ_LAUNCH_CHECK();
    const long rows_n = rows * SS_N;
    NNZ_LAUNCH(scan_carry_kernel<true>, dim3((unsigned)((rows_n + 3) / 4)), dim3(256), 0, s, a.P, a.S, a.Gin, rows_n,
               a.nchunks);
    NNZ_LAUNCH_CHECK();
  } else {
    if ((e = nnz::zero_async(a.Gin, sizeof(float) * rows * SS_N, s)) != hipSuccess) return (int)e;
  }
  NNZ_LAUNCH(xs_rl_bwd_kernel<XS>, grid, dim3(64), 0, s, a, clb, Hck, atomic_dp);
  NNZ_LAUNCH_CHECK();
  if (slabs && (e = slab_fold(a, XS, a.Dg / 64, s)) != hipSuccess) return (int)e;
  NNZ_LAUNCH(scan_bwd_finalize_kernel, dim3((unsigned)((a.KD * SS_N + 3) / 4)), dim3(256), 0, s, a.P, a.S, a.dA, a.dbias,
             a.dD, a.Bt, a.KD, a.nchunks, dWdt, a.R, a.a_is_log ? a.A : (const float*)nullptr);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_scan_tuning(int knob, int value) {
  if (knob < 0 || knob >= 5) return NNZ_EINVAL;
  nnz::g_scan_tuning[knob] = value;
  return NNZ_OK;
}
extern "C" int nnz_scan_tuning_get(int knob) { return (knob < 0 || knob >= 5) ? -1 : nnz::g_scan_tuning[knob]; }

// buffer sizes of the cross-scan entry points (cover both kernel generations)
extern "C" long nnz_ss2d_scan_state_floats(int Bt, int Dg, int L) {
  const long rows = (long)Bt * 4 * Dg;
  return rows * nnz::SS_N * nnz::rl_nch_max(L) + rows * (((long)L + nnz::RL_T - 1) / nnz::RL_T) * nnz::SS_N;
}
extern "C" long nnz_ss2d_scan_grad_state_floats(int Bt, int Dg, int L) {
  return (long)Bt * 4 * Dg * nnz::SS_N * nnz::rl_nch_max(L);
}
// most slabs either generation can ask for on this shape (ScanArgs::slab): channel groups of 64 (channels-on-lanes) or the
// time-on-lanes split of a group over workgroups
static long scan_slab_parts(int Bt, int K, int Dg, int L) {
  const long p2 = Dg > 64 ? Dg / 64 : 1;
  const long p1 = Dg / nnz::pick_rows_per_wg(Dg, (long)Bt * K * ((L + nnz::SS_CL - 1) / nnz::SS_CL));
  const long p = p1 > p2 ? p1 : p2;
  return p >= 2 ? p : 0;
}
extern "C" long nnz_ss2d_scan_workspace_floats(int Bt, int Dg, int L) {
  if (Bt < 1 || Dg < nnz::SS_NW || L < 1) return 0;
  return 2L * Bt * 4 * Dg * nnz::SS_N * nnz::rl_nch_max(L) +
         scan_slab_parts(Bt, 4, Dg, L) * 4L * Bt * (nnz::SS_RMAX + 2 * nnz::SS_N) * L;
}

// buffer sizes of the plain entry points (cover both kernel generations: chunks as short as 64 steps, and the forward's
// 16-step checkpoints behind the chunk-entry states)
extern "C" long nnz_selective_scan_workspace_floats(int Bt, int KD, int L) {
  // P and S; behind them the slabs of the deterministic dB / dC fold.  K is not an argument here: with K >= 1 groups of KD / K
  // channels the slab count is at most KD / SS_NW and one slab holds 2 Bt K N L floats, K (Dg / rows_per_wg) <= KD / SS_NW.
  if (Bt < 1 || KD < 1 || L < 1) return 0;
  return 2L * Bt * KD * nnz::SS_N * nnz::rl_nch_max(L) + 2L * Bt * (KD / nnz::SS_NW + 1) * nnz::SS_N * L;
}
extern "C" long nnz_selective_scan_state_floats(int Bt, int KD, int L) {
  const long rows = (long)Bt * KD;
  return rows * nnz::SS_N * nnz::rl_nch_max(L) + rows * (((long)L + nnz::RL_T - 1) / nnz::RL_T) * nnz::SS_N;  // Hin, Hck
}
extern "C" long nnz_selective_scan_grad_state_floats(int Bt, int KD, int L) {
  return (long)Bt * KD * nnz::SS_N * nnz::rl_nch_max(L);  // Gin
}

namespace nnz {

template <bool XS>
static int scan_forward_impl(ScanArgs& a, float* chunk_state, float* workspace, hipStream_t s) {
  if (int rc = check(a)) return rc;
  a.nchunks = (a.L + SS_CL - 1) / SS_CL;
  const long rows = (long)a.Bt * a.KD;
  a.P = workspace;
  a.S = workspace + rows * SS_N * a.nchunks;
  a.Hin = chunk_state;
  a.rows_per_wg = pick_rows_per_wg(a.Dg, (long)a.Bt * a.K * a.nchunks);
  dim3 grid(a.nchunks, a.Bt * a.K * (a.Dg / a.rows_per_wg));
  if (a.nchunks > 1) {
    NNZ_LAUNCH((scan_fwd_kernel<false, XS>), grid, dim3(SS_NW * 64), 0, s, a);
    NNZ_LAUNCH_CHECK();
    const long rows_n = rows * SS_N;
    NNZ_LAUNCH(scan_carry_kernel<false>, dim3((unsigned)((rows_n + 3) / 4)), dim3(256), 0, s, a.P, a.S, a.Hin,
                       rows_n, a.nchunks);
    NNZ_LAUNCH_CHECK();
  } else {
    hipError_t e = nnz::zero_async(a.Hin, sizeof(float) * rows * SS_N, s);
    if (e != hipSuccess) return (int)e;
  }
  NNZ_LAUNCH((scan_fwd_kernel<true, XS>), grid, dim3(SS_NW * 64), 0, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <bool XS>
static int scan_backward_impl(ScanArgs& a, const float* chunk_state, float* grad_state, float* workspace, float* dWdt,
                              hipStream_t s) {
  if (int rc = check(a)) return rc;
  a.nchunks = (a.L + SS_CL - 1) / SS_CL;
  const long rows = (long)a.Bt * a.KD;
  a.P = workspace;
  a.S = workspace + rows * SS_N * a.nchunks;
  a.Hin = const_cast<float*>(chunk_state);
  a.Gin = grad_state;
  a.rows_per_wg = pick_rows_per_wg(a.Dg, (long)a.Bt * a.K * a.nchunks);
  hipError_t e;
  const int parts1 = a.Dg / a.rows_per_wg;
  const bool slabs = slab_setup(a, XS, parts1, workspace);
  if (a.rows_per_wg != a.Dg && !slabs) {  // several workgroups add into one dB/dC (dP) tile
    if (XS) {
      if ((e = nnz::zero_async(a.xs_dP, sizeof(float) * 2L * a.Bt * 2 * a.Cp * a.L, s)) != hipSuccess) return (int)e;
    } else {
      if ((e = nnz::zero_async(a.dB, sizeof(float) * (long)a.Bt * a.K * SS_N * a.L, s)) != hipSuccess) return (int)e;
      if ((e = nnz::zero_async(a.dC, sizeof(float) * (long)a.Bt * a.K * SS_N * a.L, s)) != hipSuccess) return (int)e;
    }
  }
  const int lds_summary = SS_BWD_LDS_SUMMARY + (XS ? a.R * SS_DTP * 4 : 0);
  const int lds_final = SS_BWD_LDS_FINAL + (XS ? a.R * (SS_DTP + SS_CL) * 4 : 0);
  dim3 grid(a.nchunks, a.Bt * a.K * (a.Dg / a.rows_per_wg));
  if (a.nchunks > 1) {
    NNZ_LAUNCH((scan_bwd_kernel<false, XS>), grid, dim3(SS_NW * 64), lds_summary, s, a);
    NNZ_LAUNCH_CHECK();
    const long rows_n = rows * SS_N;
    NNZ_LAUNCH(scan_carry_kernel<true>, dim3((unsigned)((rows_n + 3) / 4)), dim3(256), 0, s, a.P, a.S, a.Gin,
                       rows_n, a.nchunks);
    NNZ_LAUNCH_CHECK();
  } else {
    if ((e = nnz::zero_async(a.Gin, sizeof(float) * rows * SS_N, s)) != hipSuccess) return (int)e;
  }
  static DynLdsCache lds_cache;  // per instantiation (XS or not), per device
  e = ensure_dyn_lds(reinterpret_cast<const void*>(scan_bwd_kernel<true, XS>), lds_final, lds_cache);
  if (e != hipSuccess) return (int)e;
  NNZ_LAUNCH((scan_bwd_kernel<true, XS>), grid, dim3(SS_NW * 64), lds_final, s, a);
  NNZ_LAUNCH_CHECK();
  if (slabs && (e = slab_fold(a, XS, parts1, s)) != hipSuccess) return (int)e;
  NNZ_LAUNCH(scan_bwd_finalize_kernel, dim3((unsigned)((a.KD * SS_N + 3) / 4)), dim3(256), 0, s, a.P, a.S, a.dA,
                     a.dbias, a.dD, a.Bt, a.KD, a.nchunks, dWdt, XS ? a.R : 0,
                     (XS && a.a_is_log) ? a.A : (const float*)nullptr);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_selective_scan_forward(const float* u, const float* delta, const float* A, const float* Bm,
                                          const float* Cm, const float* D, const float* delta_bias, float* y,
                                          float* chunk_state, float* workspace, int Bt, int K, int Dg, int N, int L,
                                          int delta_softplus, void* stream) {
  using namespace nnz;
  if (!u || !delta || !A || !Bm || !Cm || !y || !chunk_state || !workspace || N != SS_N) return NNZ_EINVAL;
  ScanArgs a = {};
  a.u = u; a.delta = delta; a.A = A; a.Bm = Bm; a.Cm = Cm; a.D = D; a.bias = delta_bias; a.y = y;
  a.Bt = Bt; a.K = K; a.Dg = Dg; a.KD = K * Dg; a.L = L; a.softplus = delta_softplus;
  if (rl_ok(a)) return rl_forward<false>(a, chunk_state, workspace, (hipStream_t)stream);
  return scan_forward_impl<false>(a, chunk_state, workspace, (hipStream_t)stream);
}

extern "C" int nnz_selective_scan_backward(const float* u, const float* delta, const float* A, const float* Bm,
                                           const float* Cm, const float* D, const float* delta_bias, const float* dy,
                                           const float* chunk_state, float* grad_state, float* workspace, float* du,
                                           float* ddelta, float* dA, float* dB, float* dC, float* dD, float* dbias,
                                           int Bt, int K, int Dg, int N, int L, int delta_softplus, void* stream) {
  using namespace nnz;
  if (!u || !delta || !A || !Bm || !Cm || !dy || !chunk_state || !grad_state || !workspace || !du || !ddelta || !dA ||
      !dB || !dC || N != SS_N)
    return NNZ_EINVAL;
  ScanArgs a = {};
  a.u = u; a.delta = delta; a.A = A; a.Bm = Bm; a.Cm = Cm; a.D = D; a.bias = delta_bias; a.dy = dy;
  a.du = du; a.ddelta = ddelta; a.dA = dA; a.dB = dB; a.dC = dC; a.dD = dD; a.dbias = dbias;
  a.Bt = Bt; a.K = K; a.Dg = Dg; a.KD = K * Dg; a.L = L; a.softplus = delta_softplus;
  if (rl_ok(a)) return rl_backward<false>(a, chunk_state, grad_state, workspace, nullptr, (hipStream_t)stream);
  return scan_backward_impl<false>(a, chunk_state, grad_state, workspace, nullptr, (hipStream_t)stream);
}

// ---- cross-scan entry points (SS2D: 4 directions over one (B, Dg, H, W) input; see ScanArgs) -------------------------
extern "C" int nnz_ss2d_scan_forward(const float* x2, const float* P, const float* Wdt, const float* A, const float* D,
                                     const float* delta_bias, float* y, float* chunk_state, float* workspace, int Bt,
                                     int Dg, int R, int L, int delta_softplus, int a_is_log, void* stream) {
  using namespace nnz;
  if (!x2 || !P || !Wdt || !A || !y || !chunk_state || !workspace || R < 1 || R > SS_RMAX) return NNZ_EINVAL;
  ScanArgs a = {};
  a.u = x2; a.xs_P = P; a.xs_Wdt = Wdt; a.A = A; a.D = D; a.bias = delta_bias; a.y = y;
  a.R = R; a.Cp = R + 2 * SS_N;
  a.Bt = Bt; a.K = 4; a.Dg = Dg; a.KD = 4 * Dg; a.L = L; a.softplus = delta_softplus; a.a_is_log = a_is_log;
  if (rl_ok(a)) return rl_forward<true>(a, chunk_state, workspace, (hipStream_t)stream);
  return scan_forward_impl<true>(a, chunk_state, workspace, (hipStream_t)stream);
}

extern "C" int nnz_ss2d_scan_backward(const float* x2, const float* P, const float* Wdt, const float* A, const float* D,
                                      const float* delta_bias, const float* dy2, const float* chunk_state,
                                      float* grad_state, float* workspace, float* du, float* dP, float* dWdt, float* dA,
                                      float* dD, float* dbias, int Bt, int Dg, int R, int L, int delta_softplus,
                                      int a_is_log, void* stream) {
  using namespace nnz;
  if (!x2 || !P || !Wdt || !A || !dy2 || !chunk_state || !grad_state || !workspace || !du || !dP || !dWdt || !dA ||
      R < 1 || R > SS_RMAX)
    return NNZ_EINVAL;
  ScanArgs a = {};
  a.u = x2; a.xs_P = P; a.xs_Wdt = Wdt; a.A = A; a.D = D; a.bias = delta_bias; a.dy = dy2;
  a.du = du; a.xs_dP = dP; a.dA = dA; a.dD = dD; a.dbias = dbias;
  a.R = R; a.Cp = R + 2 * SS_N;
  a.Bt = Bt; a.K = 4; a.Dg = Dg; a.KD = 4 * Dg; a.L = L; a.softplus = delta_softplus; a.a_is_log = a_is_log;
  if (rl_ok(a)) return rl_backward<true>(a, chunk_state, grad_state, workspace, dWdt, (hipStream_t)stream);
  return scan_backward_impl<true>(a, chunk_state, grad_state, workspace, dWdt, (hipStream_t)stream);
}
