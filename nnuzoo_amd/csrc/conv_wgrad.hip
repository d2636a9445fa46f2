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

// -DNNZ_WGRAD_TIMESTAMPS=1 (tools/probes/conv_phase_probe.py --wgrad; never the shipped library): thread 0 of every workgroup
// accumulates s_memtime differences per phase over its tiles and writes ts[workgroup][8] = {set-up, wait + stage, issuing the next
// tile's loads, MFMA loop, flush, tiles} at the end - buffer address = conv tuning knobs 12 (low) / 13 (high)
#ifndef NNZ_WGRAD_TIMESTAMPS
#define NNZ_WGRAD_TIMESTAMPS 0
#endif
#ifndef NNZ_SETPRIO
#define NNZ_SETPRIO 0   // experiment: wave priority raised over the MFMA sequence (measured: see profiles/r04_conv_ab_same_box.txt)
#endif
#ifndef NNZ_WGRAD_AHEAD
#define NNZ_WGRAD_AHEAD 3   // fragments requested ahead of the MFMA that consumes them (AHEAD + 1 register buffers)
#endif
extern "C" int nnz_conv_tuning_get(int knob);
struct WgradDev {
  const f16* p;  // boxed operand
  const f16* q;  // plain operand
  float* dw;     // [T][A][B] fp32, zeroed by the launcher (atomic mode)
  float* part;   // [splits][T][A][B] fp32 partial blocks, plain stores (two-stage mode); null = atomic mode
  long tab;      // T*A*B
  nnz_conv_desc d;  // Cin = A (boxed channels), Cout = B (plain channels), ldi/ldo their strides
  int tiles[3];
  int ntiles;   // per launch: N * tiles
  int splits;
  int pairs_b;  // B/32
  // Consumer-side InstanceNorm + LeakyReLU (nnz_conv_tap_wgrad_to_grad_innorm): an operand that is the RAW conv output of
  // its producer block is normalised while its tile is staged, y = lrelu(x * scale + shift), {scale, shift} =
  // tab[n][c - c0][2..3]; channels below c0 (the transposed-conv half of a cat buffer) pass unchanged; padding stays zero.
  // Same arithmetic and single fp16 rounding as the apply pass of norm_act.hip: the LDS images hold the bits the
  // materialised activation would.  Tables of this workgroup's 32 + 32 channels for all N samples sit behind the tiles.
  const float* p_tab;
  const float* q_tab;
  int p_c0, q_c0;
  float p_slope, q_slope;
  unsigned p_bytes, q_bytes;  // extents of the operands for the staging loads' buffer descriptors (launcher; < 2^32 - 16)
#if NNZ_WGRAD_TIMESTAMPS
  unsigned long long* ts;
#endif
};

// Box geometry policies (same split as conv_fprop.hip): compile-time isotropic stride/extent for the 3-D plans the
// bench runs, per-axis run-time values for 2-D (depth-1) and anisotropic plans.
template <int IS, int EXT>
struct WGeoIso {
  static constexpr bool kStatic = true;
  static constexpr int kIS = IS, kEXT = EXT;
  __host__ __device__ explicit WGeoIso(const nnz_conv_desc&) {}
  __host__ __device__ constexpr int is(int) const { return IS; }
  __host__ __device__ constexpr int ext(int) const { return EXT; }
};
struct WGeoDyn {
  static constexpr bool kStatic = false;
  static constexpr int kIS = 0, kEXT = 0;
  int s[3], e[3];
  __host__ __device__ explicit WGeoDyn(const nnz_conv_desc& d)
      : s{d.in_stride[0], d.in_stride[1], d.in_stride[2]}, e{d.ext[0], d.ext[1], d.ext[2]} {}
  __host__ __device__ int is(int a) const { return s[a]; }
  __host__ __device__ int ext(int a) const { return e[a]; }
};

template <int TD, int TH, int TW, class G>
struct WBoxGeom {
  int BD, BH, BW, BOX_BYTES, NBOXLOAD;
  __host__ __device__ explicit WBoxGeom(const G& g) {
    BD = (TD - 1) * g.is(0) + g.ext(0) + 1;
    BH = (TH - 1) * g.is(1) + g.ext(1) + 1;
    BW = (TW - 1) * g.is(2) + g.ext(2) + 1;
    BOX_BYTES = BD * BH * BW * 64;
    NBOXLOAD = BD * BH * BW * 4;
  }
};

template <int TD, int TH, int TW>
struct WgCfg {
  static constexpr int NVOX = TD * TH * TW;
  static constexpr int Q_BYTES = NVOX * 64;
  static constexpr int NQLOAD = NVOX * 4;
  static constexpr int LPT_Q = (NQLOAD + 255) / 256;
  static constexpr int KB = NVOX / 16;
  static_assert(TW == 8, "k-block map assumes TW == 8");
};

template <int TD, int TH, int TW, int LPT_BOX, int MAXT, class G>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradDev p) {
  using C = WgCfg<TD, TH, TW>;
  const G geo(p.d);
  const WBoxGeom<TD, TH, TW, G> bgd(geo);
  // compile-time constants for the isotropic policy, run-time values otherwise
  constexpr bool ST = G::kStatic;
  const int gBD = ST ? (TD - 1) * G::kIS + G::kEXT + 1 : bgd.BD;
  const int gBH = ST ? (TH - 1) * G::kIS + G::kEXT + 1 : bgd.BH;
  const int gBW = ST ? (TW - 1) * G::kIS + G::kEXT + 1 : bgd.BW;
  const int gBOX_BYTES = gBD * gBH * gBW * 64;
  const int gNBOXLOAD = gBD * gBH * gBW * 4;
  const int ISD = ST ? G::kIS : geo.is(0), ISH = ST ? G::kIS : geo.is(1), ISW = ST ? G::kIS : geo.is(2);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* box = smem;
  char* qt = smem + gBOX_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5;

  const int pair = blockIdx.x / p.splits;
  const int split = blockIdx.x % p.splits;
  const int a0 = (pair / p.pairs_b) * 32;
  const int b0 = (pair % p.pairs_b) * 32;

  const int T = p.d.ntaps_total;
  // everything the per-tile prefetch needs, pinned in SGPRs (see pin_uniform)
  const int Di = pin_uniform(p.d.in_dims[0]), Hi = pin_uniform(p.d.in_dims[1]), Wi = pin_uniform(p.d.in_dims[2]);
  const int Do = pin_uniform(p.d.out_dims[0]), Ho = pin_uniform(p.d.out_dims[1]), Wo = pin_uniform(p.d.out_dims[2]);
  const int Dm = pin_uniform(p.d.m_dims[0]), Hm = pin_uniform(p.d.m_dims[1]), Wm = pin_uniform(p.d.m_dims[2]);
  const int tiles1 = pin_uniform(p.tiles[1]), tiles2 = pin_uniform(p.tiles[2]);
  const int tiles_per_n = pin_uniform(p.tiles[0] * p.tiles[1] * p.tiles[2]);
  const int ldi = pin_uniform(p.d.ldi), ldo = pin_uniform(p.d.ldo);
  const int lo0 = pin_uniform(p.d.lo[0]), lo1 = pin_uniform(p.d.lo[1]), lo2 = pin_uniform(p.d.lo[2]);
  const int os0 = pin_uniform(p.d.out_stride[0]), os1 = pin_uniform(p.d.out_stride[1]),
            os2 = pin_uniform(p.d.out_stride[2]);
  const int oo0 = pin_uniform(p.d.groups[0].ooff[0]), oo1 = pin_uniform(p.d.groups[0].ooff[1]),
            oo2 = pin_uniform(p.d.groups[0].ooff[2]);
  const f16* Pp = pin_uniform(p.p) + a0;
  const f16* Qp = pin_uniform(p.q) + b0;
  const int ntiles = pin_uniform(p.ntiles), splits = pin_uniform(p.splits);

  // consumer-side norm tables: [N][32] {scale, shift} per operand, behind the Q tile; visible after the loop's first barrier
  float* ptab = reinterpret_cast<float*>(smem + gBOX_BYTES + C::Q_BYTES);
  float* qtab = ptab + p.d.N * 64;
  const bool pnorm = p.p_tab && a0 >= p.p_c0, qnorm = p.q_tab && b0 >= p.q_c0;
  if (pnorm)
    for (int i = tid; i < p.d.N * 32; i += 256)
      *reinterpret_cast<f32x2*>(ptab + 2 * i) = *reinterpret_cast<const f32x2*>(
          p.p_tab + ((size_t)(i >> 5) * (p.d.Cin - p.p_c0) + (a0 - p.p_c0) + (i & 31)) * 4 + 2);
  if (qnorm)
    for (int i = tid; i < p.d.N * 32; i += 256)
      *reinterpret_cast<f32x2*>(qtab + 2 * i) = *reinterpret_cast<const f32x2*>(
          p.q_tab + ((size_t)(i >> 5) * (p.d.Cout - p.q_c0) + (b0 - p.q_c0) + (i & 31)) * 4 + 2);

  // taps of this wave: wave, wave+4, ...  (MAXT = 7 for 27 taps, 2 for the 8 taps of the k2s2 transpose)
  int tap_off[MAXT];
  int ntw = 0;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + 4 * i;
    tap_off[i] = 0;
    if (t < T) {
      const nnz_conv_tap tp = p.d.taps[t];
      // W-stride 2: the LDS image holds the even box columns first, then the odd ones, so that the four voxels of a
      // transposed read are adjacent 64-byte rows again (interleaved they were 2-way bank conflicts)
      const int o2 = tp.off[2] - p.d.lo[2];
      const int col = ISW == 2 ? (o2 >> 1) + (o2 & 1) * ((gBW + 1) >> 1) : o2;
      tap_off[i] = (((tp.off[0] - p.d.lo[0]) * gBH + (tp.off[1] - p.d.lo[1])) * gBW + col) * 64;
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

  // staging loads = buffer loads: pieces outside the volume carry an offset beyond the extent and come back as zeros - no
  // exec-masked branch per piece (see conv_fprop.hip)
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(Pp), 0, (int)p.p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(Qp), 0, (int)p.q_bytes, 0x00020000);
  u32x4 breg[LPT_BOX];
  u32x4 qreg[C::LPT_Q];
  unsigned bmask = 0, qmask = 0;   // pieces of the prefetched tile that came from memory (the others are zero padding)
  bool ppad = false, qpad = false;  // the prefetched box / tile has pieces outside the volume (uniform over the workgroup)

  // The prefetch of a tile = prep_tile (uniform values of the tile) + one issue_piece per 16-byte piece of the thread.  Inside the
  // tile loop the pieces of the NEXT tile are issued one by one between the MFMAs of the current one: issued as a block of 14
  // in front of the MFMA loop they took 3 000-3 700 clocks per tile waiting for vector-memory queue slots - not for their address
  // arithmetic, see profiles/r04_wgrad_phases.txt - during which the wave issued nothing else.
  int t_n = 0, t_m0d = 0, t_m0h = 0, t_m0w = 0, t_lod = 0, t_loh = 0, t_low = 0;
  bool t_live = false;
  auto prep_tile = [&](int tile, bool live) {
    bmask = qmask = 0;
    t_live = live;
    t_n = tile / tiles_per_n;
    int r = tile - t_n * tiles_per_n;
    const int tw_i = r % tiles2;
    r /= tiles2;
    const int th_i = r % tiles1;
    const int td_i = r / tiles1;
    t_m0d = td_i * TD, t_m0h = th_i * TH, t_m0w = tw_i * TW;
    t_lod = t_m0d * ISD + lo0, t_loh = t_m0h * ISH + lo1, t_low = t_m0w * ISW + lo2;
    ppad = t_lod < 0 || t_loh < 0 || t_low < 0 || t_lod + gBD > Di || t_loh + gBH > Hi || t_low + gBW > Wi;
    qpad = t_m0d + TD > Dm || t_m0h + TH > Hm || t_m0w + TW > Wm || (t_m0d + TD - 1) * os0 + oo0 >= Do ||
           (t_m0h + TH - 1) * os1 + oo1 >= Ho || (t_m0w + TW - 1) * os2 + oo2 >= Wo;
  };
  constexpr int NPIECES = LPT_BOX + C::LPT_Q;
  auto issue_piece = [&](int j) {   // j is a compile-time constant at every call site
    if (j < LPT_BOX) {
      const int i = j;
      const int c = tid + i * 256;
      const int part = c & 3;
      const int s = c >> 2;
      const int bwl = s % gBW;  // column of the LDS image
      const int bh = (s / gBW) % gBH;
      const int bd = s / (gBW * gBH);
      const int hw = (gBW + 1) >> 1;
      const int bw = ISW == 2 ? (bwl < hw ? 2 * bwl : 2 * (bwl - hw) + 1) : bwl;
      const int id = t_lod + bd, ih = t_loh + bh, iw = t_low + bw;
      const bool ok = t_live & (c < gNBOXLOAD) & ((unsigned)id < (unsigned)Di) & ((unsigned)ih < (unsigned)Hi) & ((unsigned)iw < (unsigned)Wi);
      const unsigned off = ((unsigned)((((t_n * Di + id) * Hi + ih) * Wi + iw) * ldi) + part * 8) * 2u;
      breg[i] = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, (int)(ok ? off : OOB), 0, 0);
      bmask |= ok ? 1u << i : 0u;
    } else {
      const int i = j - LPT_BOX;
      const int c = tid + i * 256;
      const int part = c & 3;
      const int s = c >> 2;
      const int tw = s % TW, th = (s / TW) % TH, td = s / (TW * TH);
      const int md = t_m0d + td, mh = t_m0h + th, mw = t_m0w + tw;
      const int od = md * os0 + oo0;
      const int oh = mh * os1 + oo1;
      const int ow = mw * os2 + oo2;
      const bool ok = t_live & (c < C::NQLOAD) & (md < Dm) & (mh < Hm) & (mw < Wm) & (od < Do) & (oh < Ho) & (ow < Wo);
      const unsigned off = ((unsigned)((((t_n * Do + od) * Ho + oh) * Wo + ow) * ldo) + part * 8) * 2u;
      qreg[i] = __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, (int)(ok ? off : OOB), 0, 0);
      qmask |= ok ? 1u << i : 0u;
    }
  };
  // Consumer-side norm of the staged tile between the two barriers, CHANNEL PAIR by channel pair (every piece of a thread holds
  // the same 8 channels: c = tid + 256 i -> part = tid & 3; dword q of a piece = channels 2q, 2q + 1 of that part): one 16-byte
  // table read per pair, then 4 VALU operations per piece (common.hpp norm_lrelu_pair).  Padding pieces are normalised like
  // the others and zeroed again when the tile has any (ppad / qpad).  The caller makes the normalised operand the PLAIN one
  // where it can (conv_plan.conv_wgrad_flipped): 4 halo-free pieces per tile instead of 10 boxed ones.
  // (Measured and dropped: the same work inside the MFMA loop of the previous tile, piece by piece or as a software pipeline
  //  over the last k-blocks - +12 % on the kernel either way against +7 % here; the loop is issue-bound per wave, and the extra
  //  live state pushed the 4 x 8 x 8 instantiation into scratch.)
  auto norm_regs = [&](const float* tp, float slope, auto& regs, int npieces) {
    const nnz_h2 sl = slope_pair(slope);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(tp + 4 * q);
#pragma unroll
      for (int i = 0; i < LPT_BOX + C::LPT_Q; ++i)
        if (i < npieces) regs[i < npieces ? i : 0][q] = norm_lrelu_pair(regs[i < npieces ? i : 0][q], t[0], t[1], t[2], t[3], sl);
    }
  };
  auto zero_padding = [&]() {
    if (pnorm && ppad) {
#pragma unroll
      for (int i = 0; i < LPT_BOX; ++i)
        if (!((bmask >> i) & 1u)) breg[i] = u32x4{0u, 0u, 0u, 0u};
    }
    if (qnorm && qpad) {
#pragma unroll
      for (int i = 0; i < C::LPT_Q; ++i)
        if (!((qmask >> i) & 1u)) qreg[i] = u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto write_lds = [&](int tile) {
    if (pnorm || qnorm) {
      const int n = tile / tiles_per_n;
      if (pnorm) norm_regs(ptab + n * 64 + (tid & 3) * 16, p.p_slope, breg, LPT_BOX);
      if (qnorm) norm_regs(qtab + n * 64 + (tid & 3) * 16, p.q_slope, qreg, C::LPT_Q);
      zero_padding();
    }
#pragma unroll
    for (int i = 0; i < LPT_BOX; ++i) {
      const int c = tid + i * 256;
      if (c < gNBOXLOAD) *reinterpret_cast<u32x4*>(box + c * 16) = breg[i];
    }
#pragma unroll
    for (int i = 0; i < C::LPT_Q; ++i) {
      const int c = tid + i * 256;
      if (c < C::NQLOAD) *reinterpret_cast<u32x4*>(qt + c * 16) = qreg[i];
    }
  };

#if NNZ_WGRAD_TIMESTAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[5] = {0, 0, 0, 0, 0}, n_tiles = 0;
  const unsigned long long t_start = wall_clock64();
#define NNZ_WTS(k) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[k] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define NNZ_WTS(k) do {} while (0)
#endif
  int tile = split;
  prep_tile(tile < ntiles ? tile : 0, tile < ntiles);
#pragma unroll
  for (int j = 0; j < NPIECES; ++j) issue_piece(j);
  NNZ_WTS(0);
  for (; tile < ntiles; tile += splits) {
    __syncthreads();
    write_lds(tile);
    __syncthreads();
    NNZ_WTS(1);
    // a workgroup's last tile "prefetches" with every piece marked absent (offset beyond the extent: no memory traffic)
    prep_tile(tile + splits < ntiles ? tile + splits : 0, tile + splits < ntiles);
    NNZ_WTS(2);

    // The tile's KB k-blocks x MAXT tap slots as ONE sequence of fragments f = kb * MAXT + i, software-pipelined by hand: fragment
    // f + AHEAD (3) is requested (two transposed reads into buffer (f + AHEAD) % (AHEAD + 1)) right before the MFMA of fragment f
    // is issued, so every MFMA waits for reads that are three MFMAs old (four: measured, no gain).  Left to itself the scheduler kept two fragments in flight and waited with
    // lgkmcnt(1) / (2) in front of every MFMA, draining to lgkmcnt(0) every second k-block: the loop took 7 000 clocks per tile
    // for 3 600 clocks of MFMA issue, with or without a second workgroup on the CU (profiles/r04_wgrad_phases.txt).
    // Voxel of (kb, kk = 8*hh + 4*s + qrow): two h-rows of 8 w per k-block -> td = kb / (TH/2), th = 2 (kb % (TH/2)) + hh.
    // A tap slot beyond the wave's taps re-reads tap offset 0 and is never flushed.
    {
      constexpr int S = C::KB * MAXT, AHEAD = NNZ_WGRAD_AHEAD, NXB = AHEAD + 1;
      constexpr int NBQ = MAXT > AHEAD ? 2 : 4;   // k-blocks whose plain-operand fragment is live at once: the current one and those the look-ahead reaches
      static_assert(MAXT > AHEAD || (AHEAD + MAXT - 1) / MAXT + 1 <= NBQ, "plain-operand buffers");
      static_assert(TH % 2 == 0 && S > AHEAD, "k-block map");
      static_assert(NPIECES <= 2 * S, "at most two pieces of the next tile per fragment");
      const char* xlane = box + ((hh * ISH * gBW + qrow) * 64 + chan_byte);
      const char* qlane = qt + ((8 * hh + qrow) * 64 + chan_byte);
      auto rd_q = [&](int kb) {
        union { i16x4 v[2]; f16x8 h; } u;
        u.v[0] = lds_read_tr16(qlane + kb * 1024);
        u.v[1] = lds_read_tr16(qlane + kb * 1024 + 4 * 64);
        return u.h;
      };
      auto rd_x = [&](int f) {
        const int kb = f / MAXT, i = f % MAXT;
        const int koff = (((kb / (TH / 2)) * ISD * gBH + 2 * (kb % (TH / 2)) * ISH) * gBW) * 64;
        union { i16x4 v[2]; f16x8 h; } u;
        u.v[0] = lds_read_tr16(xlane + koff + tap_off[i]);
        u.v[1] = lds_read_tr16(xlane + koff + tap_off[i] + 4 * 64);
        return u.h;
      };
      f16x8 xb[NXB], bqb[NBQ];
#if NNZ_SETPRIO
      __builtin_amdgcn_s_setprio(NNZ_SETPRIO);
#endif
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) {
        if (f % MAXT == 0) bqb[(f / MAXT) % NBQ] = rd_q(f / MAXT);
        xb[f % NXB] = rd_x(f);
      }
#pragma clang loop unroll(full)
      for (int f = 0; f < S; ++f) {
        const int kb = f / MAXT, i = f % MAXT;
        if (f + AHEAD < S) {
          if ((f + AHEAD) % MAXT == 0) bqb[((f + AHEAD) / MAXT) % NBQ] = rd_q((f + AHEAD) / MAXT);
          xb[(f + AHEAD) % NXB] = rd_x(f + AHEAD);
          __builtin_amdgcn_sched_barrier(0);   // the reads go out BEFORE the MFMA (the scheduler put them behind it)
        }
        acc[i] = mfma32(xb[f % NXB], bqb[kb % NBQ], acc[i]);
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's pieces, spread evenly over the sequence
        if (((f + 1) * NPIECES) / S > (f * NPIECES) / S) issue_piece((f * NPIECES) / S);
        if (((f + 1) * NPIECES) / S > (f * NPIECES) / S + 1) issue_piece((f * NPIECES) / S + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#if NNZ_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
#if NNZ_WGRAD_TIMESTAMPS
    n_tiles += 1;
#endif
    NNZ_WTS(3);
  }

  // ---- flush: D[row = a][col = b]; row = (r&3) + 8(r>>2) + 4hh, col = lane&31 -----------------------
  const int A = p.d.Cin, B = p.d.Cout;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = wave + 4 * i;
    if (i < ntw) {
      const int widx = p.d.taps[t].widx;
      if (p.part) {
        // two-stage mode: every (pair, split) workgroup owns its 32x32 block of part[split]: plain 128-byte row
        // stores, summed in fixed order by wgrad_reduce_kernel (deterministic, no zero-fill, no atomics)
        float* dst = p.part + (size_t)split * p.tab + ((size_t)widx * A + a0) * B + b0 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          dst[(size_t)row * B] = acc[i][r];
        }
      } else {
        float* dst = p.dw + ((size_t)widx * A + a0) * B + b0 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          atomicAdd(dst + (size_t)row * B, acc[i][r]);
        }
      }
    }
  }
#if NNZ_WGRAD_TIMESTAMPS
  NNZ_WTS(4);
  if (p.ts && threadIdx.x == 0) {
    unsigned long long* o = p.ts + (size_t)blockIdx.x * 8;
    for (int k = 0; k < 5; ++k) o[k] = t_acc[k];
    o[5] = n_tiles;
    // where the workgroup ran: HW_ID (id 4: cu_id bits 11:8, sh_id 12, se_id 15:13) and XCC_ID (id 20), full registers
    o[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    o[7] = wall_clock64() - t_start;   // lifetime on the constant 100 MHz clock -> in-kernel shader clock = sum of the phases / this
  }
#endif
}

struct WgLaunchOpt {
  bool pre_zeroed = false;
  long max_wgs = 4096;     // two-stage mode: bounded by the partial workspace
  int* splits_used = nullptr;
};

template <int TD, int TH, int TW, int LPT_BOX, int MAXT, class G>
static int launch_wg_t(const WgradDev& base, hipStream_t stream, const WgLaunchOpt& opt) {
  const bool pre_zeroed = opt.pre_zeroed;
  using C = WgCfg<TD, TH, TW>;
  WgradDev p = base;
  const G geo(p.d);
  const WBoxGeom<TD, TH, TW, G> bg(geo);
  if (bg.NBOXLOAD > LPT_BOX * 256) return NNZ_EINVAL;  // register staging cannot hold this box
  const int lds = bg.BOX_BYTES + C::Q_BYTES + ((p.p_tab || p.q_tab) ? p.d.N * 512 : 0);
  if (lds > 160 * 1024) return NNZ_EINVAL;
  p.tiles[0] = (p.d.m_dims[0] + TD - 1) / TD;
  p.tiles[1] = (p.d.m_dims[1] + TH - 1) / TH;
  p.tiles[2] = (p.d.m_dims[2] + TW - 1) / TW;
  p.ntiles = p.d.N * p.tiles[0] * p.tiles[1] * p.tiles[2];
  p.pairs_b = p.d.Cout / 32;
  const int pairs = (p.d.Cin / 32) * p.pairs_b;
  // Split choice by a two-term cost model (measured, round 1): a workgroup spends ~4 us per 4x8x8x27-tap tile when two
  // workgroups share a CU (512 resident chip-wide), and the final fp32 atomic flush of its [T][32][32] block moves
  // T*4 KiB at ~1.1 TB/s chip-wide.  More splits = fewer tiles each but more flush traffic: 1024 workgroups made the
  // 128..320-channel layers flush-bound (113 MB of atomics for a 30 us contraction).
  int best = 1;
  double best_cost = 1e30;
  const double t_tile = 4.0 * (C::KB / 16.0) * (MAXT / 7.0);
  // (two-stage mode: the block is stored once and read once at ~5 TB/s instead of atomically added at 1.1 TB/s)
  const double t_flush = p.part ? p.d.ntaps_total * 4096.0 * 2 / 5.0e6 : p.d.ntaps_total * 4096.0 / 1.1e6;
  for (int s = 1; s <= p.ntiles && (long)s * pairs <= opt.max_wgs; s *= 2) {
    const long wgs = (long)s * pairs;
    const double rounds = (double)((wgs + 511) / 512);
    const double cost = rounds * ((p.ntiles + s - 1) / s) * t_tile + wgs * t_flush;
    if (cost < best_cost) {
      best_cost = cost;
      best = s;
    }
  }
  p.splits = best;
  const int splits = best;
  // Measured on the finished loop and dropped (profiles/r04_wgrad_phases.txt): (1) idling the second workgroup of every CU -
  // blockIdx b and b + 256, confirmed from HW_ID - for half an MFMA loop at its start so that the two alternate between MFMA loop
  // and staging; (2) an XCD-aware tile order (each XCD walks a contiguous eighth of every round of tiles, so that a box's halo is
  // found in the XCD's own L2).  Neither moved the kernel or the step.
#if NNZ_WGRAD_TIMESTAMPS
  p.ts = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)nnz_conv_tuning_get(13) << 32) | (unsigned)nnz_conv_tuning_get(12));
#endif
  {
    const unsigned long long pb = 2ull * p.d.N * p.d.in_dims[0] * p.d.in_dims[1] * p.d.in_dims[2] * (unsigned long long)p.d.ldi;
    const unsigned long long qb = 2ull * p.d.N * p.d.out_dims[0] * p.d.out_dims[1] * p.d.out_dims[2] * (unsigned long long)p.d.ldo;
    if (pb >= 0xFFFFFFF0ull || qb >= 0xFFFFFFF0ull) return NNZ_EINVAL;   // 32-bit byte offsets in the staging loads
    p.p_bytes = (unsigned)pb;
    p.q_bytes = (unsigned)qb;
  }
  auto kern = conv_wgrad_kernel<TD, TH, TW, LPT_BOX, MAXT, G>;
  static DynLdsCache lds_cache;  // per instantiation, per device
  {
    hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_cache);
    if (e != hipSuccess) return (int)e;
  }
  if (!p.part && !pre_zeroed) {
    hipError_t e = nnz::zero_async(p.dw, sizeof(float) * (size_t)p.d.ntaps_total * p.d.Cin * p.d.Cout, stream);
    if (e != hipSuccess) return (int)e;
  }
  if (opt.splits_used) *opt.splits_used = splits;
  NNZ_LAUNCH(kern, dim3(pairs * splits), dim3(256), lds, stream, p);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <int TD, int TH, int TW, int LPT_BOX, class G>
static int launch_wg(const WgradDev& p, hipStream_t stream, const WgLaunchOpt& opt) {
  const int per_wave = (p.d.ntaps_total + 3) / 4;
  if (per_wave <= 2) return launch_wg_t<TD, TH, TW, LPT_BOX, 2, G>(p, stream, opt);
  return launch_wg_t<TD, TH, TW, LPT_BOX, 7, G>(p, stream, opt);
}

template <int TD, int TH, int TW, int IS, int EXT>
static int launch_wg_iso(const WgradDev& p, hipStream_t stream, const WgLaunchOpt& opt) {
  constexpr int lpt =
      (((TD - 1) * IS + EXT + 1) * ((TH - 1) * IS + EXT + 1) * ((TW - 1) * IS + EXT + 1) * 4 + 255) / 256;
  return launch_wg<TD, TH, TW, lpt, WGeoIso<IS, EXT>>(p, stream, opt);
}

// geometry dispatch shared by the atomic and the two-stage entry points
static int launch_wgrad_any(const WgradDev& p, hipStream_t s, const WgLaunchOpt& o) {
  const nnz_conv_desc& d = p.d;
  const bool iso = d.in_stride[0] == d.in_stride[1] && d.in_stride[1] == d.in_stride[2] && d.ext[0] == d.ext[1] &&
                   d.ext[1] == d.ext[2] && d.m_dims[0] > 1;
  const bool strided = d.in_stride[0] == 2 || d.in_stride[1] == 2 || d.in_stride[2] == 2;
  if (iso) {
    const int ext = d.ext[0];
    if (d.in_stride[0] == 1)
      return ext == 0 ? launch_wg_iso<4, 8, 8, 1, 0>(p, s, o)
                      : ext == 1 ? launch_wg_iso<4, 8, 8, 1, 1>(p, s, o) : launch_wg_iso<4, 8, 8, 1, 2>(p, s, o);
    return ext <= 1 ? launch_wg_iso<2, 4, 8, 2, 1>(p, s, o) : launch_wg_iso<2, 4, 8, 2, 2>(p, s, o);
  }
  // per-axis geometry: 2-D plans (depth-1 volumes -> flat tiles) and anisotropic 3-D plans
  if (d.m_dims[0] == 1 && d.in_dims[0] == 1 && d.out_dims[0] == 1) {
    // dilated 3x3 (tap offsets up to +-8): the 1x16x8 tile with 12 staging pieces per thread holds its (16 + 16) x (8 + 16) box
    if (d.ext[1] > 2 || d.ext[2] > 2) return strided ? NNZ_EINVAL : launch_wg<1, 16, 8, 12, WGeoDyn>(p, s, o);
    return strided ? launch_wg<1, 16, 8, 9, WGeoDyn>(p, s, o) : launch_wg<1, 32, 8, 6, WGeoDyn>(p, s, o);
  }
  return strided ? launch_wg<2, 4, 8, 12, WGeoDyn>(p, s, o) : launch_wg<4, 8, 8, 10, WGeoDyn>(p, s, o);
}

struct WgKsel {
  int k[32];
};

// Stage 2a (many splits, few channels): tmp[e] = sum_s part[s][e], s in fixed order; thread = (element, split group of 4),
// coalesced over the [t][a][b] element index.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, long tab,
                                                           float* __restrict__ tmp) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x;
  const int el = tid & 63, sg = tid >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s0 = 0.f, s1 = 0.f;
  if (e < tab) {
    int s = sg;
    for (; s + 4 < splits; s += 8) {  // two loads in flight
      s0 += part[(size_t)s * tab + e];
      s1 += part[(size_t)(s + 4) * tab + e];
    }
    if (s < splits) s0 += part[(size_t)s * tab + e];
  }
  red[sg][el] = s0 + s1;
  __syncthreads();
  if (sg == 0 && e < tab) tmp[e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// Stage 2b: grad[a*sa + b*sb + ksel[t]*sk] (+)= sum_s part[s][t][a][b] for one (a, 32 b) column block and all taps,
// transposed through LDS: reads are 128-byte rows of the [t][a][b] image, writes are runs of T consecutive taps of the
// torch layout (sk = 1), instead of one 4-byte write per 64-byte sector.
__global__ __launch_bounds__(256) void wgrad_to_grad_kernel(const float* __restrict__ part, int splits, long tab,
                                                            float* __restrict__ grad, int A, int B, int T, long sa,
                                                            long sb, long sk, WgKsel ks, int accumulate) {
  __shared__ float tile[32 * 33];
  const int tid = threadIdx.x;
  const int a = blockIdx.x % A;
  const int b0 = (blockIdx.x / A) * 32;
  const int n = T * 32;
  for (int i = tid; i < n; i += 256) {
    const int t = i >> 5, bb = i & 31;
    const size_t e = ((size_t)t * A + a) * B + b0 + bb;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[(size_t)s * tab + e];
    tile[t * 33 + bb] = v;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int bb = i / T, t = i - bb * T;
    float* g = grad + a * sa + (long)(b0 + bb) * sb + ks.k[t] * sk;
    const float v = tile[t * 33 + bb];
    *g = accumulate ? *g + v : v;
  }
}

}  // namespace nnz

extern "C" int nnz_conv_tap_wgrad(const void* boxed, const void* plain, float* dw, const nnz_conv_desc* desc,
                                  int pre_zeroed, void* stream) {
  using namespace nnz;
  if (!boxed || !plain || !dw || !desc) return NNZ_EINVAL;
  const nnz_conv_desc& d = *desc;
  if (d.Cin % 32 || d.Cout % 32 || d.ngroups != 1 || d.ntaps_total > 28 || d.ntaps_total < 1 || d.ldi % 8 || d.ldo % 8)
    return NNZ_EINVAL;
  for (int a = 0; a < 3; ++a)
    if ((d.in_stride[a] != 1 && d.in_stride[a] != 2) || (d.out_stride[a] != 1 && d.out_stride[a] != 2) ||
        d.ext[a] < 0 || d.ext[a] > ((d.m_dims[0] == 1 && d.in_dims[0] == 1 && d.out_dims[0] == 1 && a > 0) ? 16 : 2))
      return NNZ_EINVAL;
  // 32-bit voxel offsets inside the kernel: batches beyond 2^31 elements per tensor run as sample chunks that
  // accumulate into the same dW (zeroed once)
  const long p_per_n = (long)d.in_dims[0] * d.in_dims[1] * d.in_dims[2] * d.ldi;
  const long q_per_n = (long)d.out_dims[0] * d.out_dims[1] * d.out_dims[2] * d.ldo;
  const long per_n = p_per_n > q_per_n ? p_per_n : q_per_n;
  if (per_n >= (1L << 31)) return NNZ_EINVAL;
  int chunk = (int)(((1L << 31) - 1) / per_n);
  if (chunk > d.N) chunk = d.N;
  hipStream_t s = (hipStream_t)stream;
  for (int n0 = 0; n0 < d.N; n0 += chunk) {
    WgradDev p = {};
    p.p = (const f16*)boxed + (size_t)n0 * p_per_n;
    p.q = (const f16*)plain + (size_t)n0 * q_per_n;
    p.dw = dw;
    p.d = d;
    p.d.N = d.N - n0 < chunk ? d.N - n0 : chunk;
    WgLaunchOpt o;
    o.pre_zeroed = pre_zeroed != 0 || n0 > 0;
    const int rc = launch_wgrad_any(p, s, o);
    if (rc != NNZ_OK) return rc;
  }
  return NNZ_OK;
}

extern "C" long nnz_conv_tap_wgrad_workspace_floats(const nnz_conv_desc* desc) {
  if (!desc) return 0;
  // up to 1024 workgroups' [T][32][32] blocks (>= one full [T][A][B] image for every channel-pair count)
  const long blk = (long)desc->ntaps_total * 1024;
  const long pairs = (long)(desc->Cin / 32) * (desc->Cout / 32);
  const long wgs = pairs > 1024 ? pairs : 1024;
  return wgs * blk + pairs * blk;  // partial blocks + one reduced [T][A][B] image
}

struct WgNormArgs {
  const float* p_tab;
  int p_c0;
  float p_slope;
  const float* q_tab;
  int q_c0;
  float q_slope;
};
static int wgrad_to_grad_impl(const void* boxed, const void* plain, float* workspace, long ws_floats, float* grad, long sa,
                              long sb, long sk, const int* ksel, int accumulate, const nnz_conv_desc* desc,
                              const WgNormArgs* nm, void* stream);

extern "C" int nnz_conv_tap_wgrad_to_grad(const void* boxed, const void* plain, float* workspace, long ws_floats,
                                          float* grad, long sa, long sb, long sk, const int* ksel, int accumulate,
                                          const nnz_conv_desc* desc, void* stream) {
  return wgrad_to_grad_impl(boxed, plain, workspace, ws_floats, grad, sa, sb, sk, ksel, accumulate, desc, nullptr, stream);
}

// ... with operands that are RAW conv outputs of their producer blocks (WgradDev::p_tab): the layer input of a convolution
// (boxed operand), the lower-resolution activation of a transposed convolution (plain operand).  tab == NULL: the operand
// is used as it is.  c0 multiples of 32.  Reference ops: the bwd-weight kernels of nn.Conv3d / nn.ConvTranspose3d fed by
// InstanceNorm + LeakyReLU outputs (default_experiment_planner.py:285-305).
extern "C" int nnz_conv_tap_wgrad_to_grad_innorm(const void* boxed, const void* plain, float* workspace, long ws_floats,
                                                 float* grad, long sa, long sb, long sk, const int* ksel, int accumulate,
                                                 const nnz_conv_desc* desc, const float* boxed_tab, int boxed_c0,
                                                 float boxed_slope, const float* plain_tab, int plain_c0,
                                                 float plain_slope, void* stream) {
  if (!desc || (!boxed_tab && !plain_tab)) return NNZ_EINVAL;
  if ((boxed_tab && (boxed_c0 < 0 || boxed_c0 % 32 || boxed_c0 >= desc->Cin)) ||
      (plain_tab && (plain_c0 < 0 || plain_c0 % 32 || plain_c0 >= desc->Cout)) || desc->N > 64)
    return NNZ_EINVAL;
  WgNormArgs nm = {boxed_tab, boxed_c0, boxed_slope, plain_tab, plain_c0, plain_slope};
  return wgrad_to_grad_impl(boxed, plain, workspace, ws_floats, grad, sa, sb, sk, ksel, accumulate, desc, &nm, stream);
}

static int wgrad_to_grad_impl(const void* boxed, const void* plain, float* workspace, long ws_floats, float* grad, long sa,
                              long sb, long sk, const int* ksel, int accumulate, const nnz_conv_desc* desc,
                              const WgNormArgs* nm, void* stream) {
  using namespace nnz;
  if (!boxed || !plain || !workspace || !grad || !ksel || !desc) return NNZ_EINVAL;
  const nnz_conv_desc& d = *desc;
  if (d.Cin % 32 || d.Cout % 32 || d.ngroups != 1 || d.ntaps_total > 28 || d.ntaps_total < 1 || d.ldi % 8 || d.ldo % 8)
    return NNZ_EINVAL;
  for (int a = 0; a < 3; ++a)
    if ((d.in_stride[a] != 1 && d.in_stride[a] != 2) || (d.out_stride[a] != 1 && d.out_stride[a] != 2) ||
        d.ext[a] < 0 || d.ext[a] > ((d.m_dims[0] == 1 && d.in_dims[0] == 1 && d.out_dims[0] == 1 && a > 0) ? 16 : 2))
      return NNZ_EINVAL;
  const long tab = (long)d.ntaps_total * d.Cin * d.Cout;
  const long pairs = (long)(d.Cin / 32) * (d.Cout / 32);
  if (ws_floats < 2 * tab) return NNZ_EINVAL;
  const long p_per_n = (long)d.in_dims[0] * d.in_dims[1] * d.in_dims[2] * d.ldi;
  const long q_per_n = (long)d.out_dims[0] * d.out_dims[1] * d.out_dims[2] * d.ldo;
  const long per_n = p_per_n > q_per_n ? p_per_n : q_per_n;
  if (per_n >= (1L << 31)) return NNZ_EINVAL;
  int chunk = (int)(((1L << 31) - 1) / per_n);
  if (chunk > d.N) chunk = d.N;
  hipStream_t s = (hipStream_t)stream;
  WgKsel ks;
  for (int i = 0; i < 32; ++i) ks.k[i] = i < d.ntaps_total ? ksel[i] : 0;
  int splits = 1;
  if (chunk >= d.N) {
    // two-stage: partial blocks in the workspace, then the fixed-order reduction straight into the torch-layout grad
    WgradDev p = {};
    p.p = (const f16*)boxed; p.q = (const f16*)plain; p.part = workspace; p.tab = tab; p.d = d;
    if (nm) {
      p.p_tab = nm->p_tab; p.p_c0 = nm->p_c0; p.p_slope = nm->p_slope;
      p.q_tab = nm->q_tab; p.q_c0 = nm->q_c0; p.q_slope = nm->q_slope;
    }
    WgLaunchOpt o;
    o.max_wgs = (ws_floats - tab) / ((long)d.ntaps_total * 1024);  // workgroups whose blocks fit beside the temp image
    if (o.max_wgs < pairs) return NNZ_EINVAL;
    o.splits_used = &splits;
    const int rc = launch_wgrad_any(p, s, o);
    if (rc != NNZ_OK) return rc;
  } else {
    if (nm) return NNZ_EINVAL;   // (the chunked path keeps materialised operands)
    // batch beyond 2^31 elements: sample chunks accumulate atomically into one [T][A][B] image in the workspace
    hipError_t e = nnz::zero_async(workspace, sizeof(float) * tab, s);
    if (e != hipSuccess) return (int)e;
    for (int n0 = 0; n0 < d.N; n0 += chunk) {
      WgradDev p = {};
      p.p = (const f16*)boxed + (size_t)n0 * p_per_n;
      p.q = (const f16*)plain + (size_t)n0 * q_per_n;
      p.dw = workspace; p.d = d;
      p.d.N = d.N - n0 < chunk ? d.N - n0 : chunk;
      WgLaunchOpt o;
      o.pre_zeroed = true;
      const int rc = launch_wgrad_any(p, s, o);
      if (rc != NNZ_OK) return rc;
    }
  }
  const float* src = workspace;
  if (splits > 8) {
    // many splits (small channel counts: tab is small): element-parallel pre-reduction into the temp image
    float* tmp = workspace + (ws_floats - tab);
    const long blocks = (tab + 63) / 64;
    NNZ_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, splits, tab,
                       tmp);
    src = tmp;
    splits = 1;
  }
  NNZ_LAUNCH(wgrad_to_grad_kernel, dim3((unsigned)(d.Cin * (d.Cout / 32))), dim3(256), 0, s, src, splits, tab,
                     grad, d.Cin, d.Cout, d.ntaps_total, sa, sb, sk, ks, accumulate);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
