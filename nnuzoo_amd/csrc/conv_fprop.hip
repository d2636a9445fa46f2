// Tap-table implicit-GEMM convolution on MFMA (gfx950), channels-last fp16 activations, fp32 accumulate.
//
// One kernel family serves every dense contraction of the PlainConvUNet forward/backward except the
// weight gradients (conv_wgrad.hip) and the Cin=1 stem (conv_stem.hip):
//   * k3 s1 / k3 s2 forward convolutions                       (1 group, 27 taps)
//   * data gradient of a k3 s1 convolution                      (1 group, 27 flipped taps)
//   * data gradient of a k3 s2 convolution                      (8 output-parity groups, 1..8 taps each)
//   * k2 s2 transposed convolution forward                      (8 groups of 1 tap, out_stride 2)
//   * data gradient of the k2 s2 transposed convolution         (1 group, 8 taps, in_stride 2)
//   * 1x1 convolutions                                           (1 group, 1 tap)
// Reference op being replaced: torch.nn.Conv3d / ConvTranspose3d as wired by
// dynamic_network_architectures.PlainConvUNet (instantiated at
// /root/reference/nnunetv2/utilities/get_network_from_plans.py:27-57).
//
// Design (MI355X-first):
//   * A workgroup owns an m-tile of TDxTHxTW voxels and NB*32 output channels.  Per 16-channel slice of
//     the reduction it stages (a) the input *box* (tile + halo) and (b) all taps' weights for that slice
//     into LDS once; all taps then read shifted windows of the box from LDS, so an input element crosses
//     the L2->LDS path once per (tile, Cout-block) instead of 27 times.
//   * v_mfma_f32_32x32x16_f16 with the weights as the A operand (rows = cout) and voxels on the lanes
//     (B operand): every lane ends up with 4 consecutive couts of ONE voxel per accumulator quad, so the
//     epilogue stores 8-byte channel runs with no LDS transpose.
//   * 32-byte LDS rows (16 ch) for box and weights; a 1-bit XOR on the 16-byte half (box: h parity,
//     weights: (cout>>4)&1) makes every ds_read_b128 lane group hit 16 distinct slots.
//   * Next slice's global loads are issued before the MFMA loop of the current one (register staging,
//     write after the barrier), >=2 workgroups per CU cover the barriers.
//   * 1-D grid with a bijective XCD remap: consecutive m-tiles (which share halo and weights) run on
//     one XCD's L2.
#include "common.hpp"
#include "conv_params.h"
#include <stddef.h>

namespace nnz {

struct TapDev {
  int lds_off;  // byte offset of the tap inside the box
  int hpar;     // parity of the tap's h offset (selects the swizzled half)
};

// -DNNZ_CONV_TIMESTAMPS=1 (tools/probes/conv_phase_probe.py builds its own library with it; never the shipped one): thread 0 of
// every workgroup records s_memtime at the phase boundaries into ts[workgroup][16] - buffer address = knobs 12 (low) / 13 (high)
#ifndef NNZ_CONV_TIMESTAMPS
#define NNZ_CONV_TIMESTAMPS 0
#endif
#if NNZ_CONV_TIMESTAMPS
#define NNZ_TS(slot)                                                                                   \
  do {                                                                                                 \
    if (p.ts && threadIdx.x == 0 && (slot) < 16) {                                                     \
      p.ts[(long)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime();                             \
      /* slot 15 (launches of <= 2 slices): the workgroup's lifetime on the constant 100 MHz clock -> in-kernel shader clock */ \
      if ((slot) == 0) p.ts[(long)blockIdx.x * 16 + 15] = wall_clock64();                              \
      if ((slot) == 14) p.ts[(long)blockIdx.x * 16 + 15] = wall_clock64() - p.ts[(long)blockIdx.x * 16 + 15]; \
    }                                                                                                  \
  } while (0)
#else
#define NNZ_TS(slot) do {} while (0)
#endif

struct ConvDev {
  const f16* in;
  f16* out;
  const f16* w;
  const float* bias;
  float* stats;  // optional [N][Cout][2] {sum, sumsq} of the fp16 outputs (InstanceNorm statistics), pre-zeroed
  // deterministic statistics (nnz_conv_tap_forward_norm): fixed-point accumulators [N][Cout][2] {sum, sumsq}, zero
  // between launches; the last workgroup turns them into nstat[N][Cout][4] = {mean, rstd, rstd*gamma, beta - mean*rstd*gamma}
  FxAcc* acc;
  float* nstat;
  unsigned* counter;
  const float* gamma;
  const float* beta;
  float eps;
  // Data-gradient launches (nnz_conv_tap_dgrad_normred): the tile this launch writes is g = dL/d(activation) of the layer
  // BELOW, i.e. the input of that layer's InstanceNorm + LeakyReLU backward.  With `bx` (that layer's fp16 conv output
  // [N][out voxels][ldbx]) and `bstat` (its table [N][Cout][4]) the epilogue also forms the two per-(sample, channel)
  // reductions of that backward - sum g' and sum g' xhat, g' = g * lrelu'(pre) - on the tile it holds, adds them to the
  // fixed-point accumulators and the launch's last workgroup writes nred[N][Cout][2] = {mean g', mean g' xhat} and the
  // affine's gradients: the separate reducing pass over (x, g) of norm_act.hip MODE 2 (two full reads) disappears.
  const f16* bx;
  const float* bstat;
  float* nred;
  float* dgamma;
  float* dbeta;
  float slope;
  int ldbx;
#if NNZ_CONV_TIMESTAMPS
  unsigned long long* ts;
#endif
  // ceil(2^32 / d) of the divisors of the workgroup-index decode (launcher): q = umulhi(n, m) is exact while n * d < 2^32
  unsigned mg_gy, mg_gxw, mg_nsplit, mg_ngroups, mg_t2, mg_t1;
  unsigned in_bytes, w_bytes, bx_bytes;  // extents of `in` / `w` / `bx` for the buffer descriptors of the staging loads (launcher; < 2^32 - 16)
  int mfma_moments;  // knob 11: forward statistics of full tiles on the matrix cores
  int sep_finish;  // knob 10: no ticket in the epilogue, conv_stats_finish_kernel / conv_normred_finish_kernel follow the launch
  int dbg;  // nnz_conv_tuning(6, bits): epilogue experiments (tools/probes/normred_epilogue_probe.py); 0 in production
  // split-K over the 16-channel slices of the reduction (the <= 8^3 levels: 10-40 workgroups each walking 20-40 slices of
  // 55 KB of weights were 55 us of pure latency per launch): workgroup (.., split) covers slices [split * kper, ...) and
  // stores its fp32 accumulators to part[split][n][voxel][cout]; conv_splitk_finish_kernel folds the splits in order
  float* part;
  long ws_floats;
  int nsplit, kper;
  nnz_conv_desc d;
  int tiles[3];
  int gx, gy, gz;
  int cout_fastest;  // workgroup order: cout block index fastest (1) or m-tile index fastest (0)
  // knob 14 (round 6): m-tiles in BRICK order - 4 x 4 x 4 bricks of tiles, bricks in (d, h, w) order - instead of the linear
  // (d, h, w) order.  The 64 workgroups an XCD runs at a time (32 CUs x 2) are then one brick (or half of one when two cout
  // blocks share a tile) whose boxes overlap on all three axes: the halo a tile shares with its d-neighbours is fetched while the
  // neighbour is resident instead of 256 tiles later, when the XCD's 4 MB L2 has long dropped it (a 16 x 16 plane of 8^3 tiles
  // at 32 channels is 8.4 MB).  Launcher: only when all three tile counts are multiples of 4.
  int brick;
  unsigned mg_b2, mg_b1;   // reciprocals of tiles[2] / 4 and tiles[1] / 4
  // knob 15 (round 6): the tap GROUPS of a multi-group launch (the eight output-parity classes of a stride-2 data gradient, the
  // eight positions of a transposed convolution) as the FASTEST index of the workgroup order instead of the slowest: the groups of
  // one m-tile read the same input box (with different taps) and interleave their output voxels on the same cache lines, and were
  // 1/8 of the launch apart - the 32 -> 64 stride-2 data gradient at 128^3 fetched 2.0 GB for a 67 MB input + 268 MB of layer-below
  // rows (profiles/r06_pmc_fetch_size.csv) and ran at the fabric's bandwidth.
  int group_fastest;
  // Consumer-side InstanceNorm + LeakyReLU (nnz_conv_tap_forward_innorm): `in` is the RAW conv output of the producer block(s)
  // and is normalised while the box is staged, y = lrelu(x * scale + shift) with {scale, shift} = in_tab[n][c - in_c0][2..3]
  // (the producer's table, written by its own launch's last workgroup); channels [0, in_c0) - the transposed-conv half of a
  // cat buffer - pass unchanged.  The separate apply pass (norm_act.hip MODE 1: one read + one write of every activation)
  // and the activated tensor itself disappear; the arithmetic is that pass's (common.hpp norm_lrelu8: fp32 FMA rounded to
  // fp16, LeakyReLU in fp16), so the box holds the same bits the materialised activation would.  Padding voxels stay zero (the reference pads the
  // ACTIVATION with zeros).
  const float* in_tab;
  int in_c0;
  float in_slope;
  int intab_off;   // byte offset of the LDS table (launcher)
  int tiles_per_wg;  // PERSIST instantiations: consecutive m-tiles (along W) per workgroup; gx counts workgroups' first tiles
  // Tap tables made by the launcher (round 4): tapenc[t] = box byte offset of tap t | h-offset parity << 4, rowenc[r] the
  // same for the nine (kh, kw) rows of the depth-reuse loop.  Lane t fetches its entry with ONE vector load from the kernel
  // argument segment.  (Before, every wave built them in a 27 + 9 iteration loop of dependent scalar loads from the
  // descriptor: ~36 scalar-cache round trips at the head of every workgroup.)
  int tapenc[NNZ_MAX_TAPS];
  int rowenc[12];
};

// Box geometry policies.  GeoIso: input stride and tap extent are compile-time and equal on the three axes (the
// isotropic 3-D plans: every box dimension folds to a constant).  GeoDyn: per-axis values from the descriptor
// (2-D plans = depth-1 volumes, anisotropic 3-D plans such as k(1,3,3) s(1,2,2)); the geometry only enters the
// set-up code, the per-tap loop sees two extra scalar multiplies.
template <int IS, int EXT>
struct GeoIso {
  __host__ __device__ explicit GeoIso(const nnz_conv_desc&) {}
  __host__ __device__ constexpr int is(int) const { return IS; }
  __host__ __device__ constexpr int ext(int) const { return EXT; }
};
struct GeoDyn {
  int s[3], e[3];
  __host__ __device__ explicit GeoDyn(const nnz_conv_desc& d)
      : s{d.in_stride[0], d.in_stride[1], d.in_stride[2]}, e{d.ext[0], d.ext[1], d.ext[2]} {}
  __host__ __device__ int is(int a) const { return s[a]; }
  __host__ __device__ int ext(int a) const { return e[a]; }
};

template <int TD, int TH, int TW, class G>
struct BoxGeom {
  int BD, BH, BW, PW, BOX_BYTES, NBOXLOAD;
  __host__ __device__ explicit BoxGeom(const G& g) {
    BD = (TD - 1) * g.is(0) + g.ext(0) + 1;
    BH = (TH - 1) * g.is(1) + g.ext(1) + 1;
    BW = (TW - 1) * g.is(2) + g.ext(2) + 1;
    // row pitch in voxels: a multiple of 4 keeps the h-rows of a stride-1 tap read on alternating bank halves; with
    // W-stride 2 (de-interleaved columns) an even pitch is conflict-free and lets the 2x4x8 stride-2 box + weights
    // fit twice into a CU's LDS (81.2 KB instead of 84.1 KB per workgroup)
    PW = g.is(2) == 2 ? (BW + 1) & ~1 : (BW + 3) & ~3;
    BOX_BYTES = BD * BH * PW * 32;
    NBOXLOAD = BD * BH * BW * 2;  // 16-byte pieces
  }
};

template <int TD, int TH, int TW, int NB>
struct ConvCfg {
  static constexpr int LPT_W = (NB * 27 * 64 + 255) / 256;
  static constexpr int MB = TD * TH * TW / 32;
  static constexpr int WAVES_M = MB >= 4 ? 4 : MB;
  static constexpr int WAVES_N = 4 / WAVES_M;
  static constexpr int WM = MB / WAVES_M;
  static constexpr int WN = NB / WAVES_N;
  static_assert(TW == 8, "lane->voxel map assumes TW == 8");
  static_assert(NB % WAVES_N == 0 && WN >= 1, "NB must cover the N-split of the waves");
  static constexpr int OUT_BYTES = TD * TH * TW * (NB * 64 + 16);  // epilogue staging image
};

// Tuning knobs (diagnostics / per-layer experiments; defaults are the measured best).  nnz_conv_tuning(knob, value):
//   0  depth-reuse loop for k3 s1 layers with Cout % 64 != 0 on >= 64^3 grids          (default 1)
//   1  depth-reuse loop (one 32-cout block per workgroup) also for Cout % 64 == 0      (default 1)
//      (with the m-tile-fastest order a 2-slice layer paid the doubled box fetch of two workgroups per tile more than
//      it gained - dec0.0's data gradient, 32 -> 64 channels: 760 -> 680 TFLOP/s; with the cout-fastest order of knob 3
//      the second workgroup finds the box in L2 and the same layer gains: 628 -> 720 on one box)
//   2  smallest m-grid edge (cube root of the voxel count) that takes the depth-reuse loop (default 16)
//   3  workgroup order: cout block fastest (1) / m-tile fastest (0)                        (default 1)
//   4  smallest Cin for knob 1                                                          (default 32)
//   5  1 disables split-K over the reduction slices (the <= 8^3 levels; needs the *_ws entry points)  (default 0)
//   6  epilogue experiments of the fused norm-backward launches (ConvDev::dbg)         (default 0)
//   7  stride-2 forward convolutions on >= 32^3 output grids take the 4x8x8 x 64-cout tile, one workgroup per CU   (default 1:
//      32 -> 64 @128^3 0.238 -> 0.207 ms, 64 -> 128 @64^3 0.102 -> 0.093 ms, tools/bench_conv_layers.py --tuning 7=0/1)
//   8  split-K applies to launches of fewer than this many workgroups                                                (default 128)
//   9  depth-reuse launches: most consecutive W tiles one (persistent) workgroup walks (power of two; 1 = off)    (default 4)
//  10  the fixed-point statistics / norm-backward reductions of a launch are turned into their tables by a separate finishing
//      kernel (1) instead of by the launch's last workgroup behind a ticket (0)                                       (default 1:
//      tools/probes/conv_phase_probe.py - wave 0 of EVERY workgroup sat ~10 000 cycles, a quarter of a 2-slice workgroup's
//      lifetime, in the two dependent memory round trips of the protocol: adds acknowledged, ticket returned)
//  11  InstanceNorm statistics of the forward epilogue: tiles inside the volume form their moments with MFMAs (1) / VALU sums and
//      lane shuffles everywhere (0)                                                                                    (default 1:
//      +1.4 % on the training step, same box.  The pilot subtraction runs in packed fp16 - exact when the spread is small
//      against the mean (the cancellation-prone case), rounded to 11 bits otherwise - so the launch keeps THREE sums per (sample,
//      channel): the consistent pair of the rounded deviations for the variance and the exact sum of x for the mean.  Table
//      against float64 on the stored outputs: mean 3e-6 like the VALU form, rstd within 1e-4 relative (VALU form 3e-6) -
//      tests/test_determinism_gpu.py test_conv_epilogue_table_{large,small}_mean hold both forms.)
//  12, 13  -DNNZ_CONV_TIMESTAMPS builds only: low / high half of the timestamp buffer's address
//  14  m-tiles in 4 x 4 x 4 brick order where all three tile counts are multiples of 4 (ConvDev::brick)               (default 1)
//  15  tap groups of a multi-group launch as the fastest workgroup index (ConvDev::group_fastest)                         (default 1)
#ifndef NNZ_SETPRIO
#define NNZ_SETPRIO 0   // experiment: wave priority raised over the depth-reuse MFMA loop
#endif
#ifndef NNZ_DRE_PERSIST
#define NNZ_DRE_PERSIST false
#endif
#ifndef NNZ_S2_PERSIST
#define NNZ_S2_PERSIST false
#endif
static int g_tuning[16] = {1, 1, 16, 1, 32, 0, 0, 1, 128, 4, 1, 1, 0, 0, 1, 1};

// LPT_BOX: 16-byte box pieces per thread (register staging bound; the launcher checks it covers the geometry)
// DRE ("depth reuse", k3 s1 tables only, 8x8x8 x 32-cout tile): a wave owns four consecutive depth planes of one h-half.
// One voxel fragment of INPUT plane d0 + p then feeds the MFMAs of the three depth taps (output planes p, p - 1, p - 2)
// instead of being re-read from LDS for each: 6 voxel + 3 weight fragments per (kh, kw) for 12 MFMAs = 0.75 KB of LDS
// reads per MFMA against 1.25 KB in the tap-by-tap loop.  With Cout = 32 there is only one N block, so this is the only
// reuse a voxel fragment can get; it takes the LDS array from ~75 % to ~50 % busy on the full-resolution layers.
// MINB: workgroups per CU the register budget is sized for (2: <= 256 VGPRs; 1: the big stride-2 tile, whose 142 KB of box +
// weights allow one workgroup per CU anyway, may use the whole file)
// PERSIST: a workgroup walks `tiles_per_wg` consecutive m-tiles along W (same sample, same cout block).  The per-workgroup set-up
// (argument loads, tap tables, fragment addresses: ~2 000 instructions, ~1/4 of a 2-slice workgroup's time) is paid once, and
// the next tile's first slice is requested while the current tile's last slice is still in its MFMA loop (forward launches: the
// epilogue does not need the staging registers), so its HBM latency hides behind that loop and the epilogue.
template <int TD, int TH, int TW, int NB, int LPT_BOX, class G, bool DRE = false, bool DFLIP = false, int MINB = 2,
          bool PERSIST = false>
__global__ __launch_bounds__(256, MINB) void conv_box_kernel(ConvDev p) {
  using C = ConvCfg<TD, TH, TW, NB>;
  NNZ_TS(0);
  static_assert(!DRE || (TD == 8 && TH == 8 && TW == 8 && NB == 1), "depth-reuse loop: 8x8x8 tile, one cout block");
  const G geo(p.d);
  const BoxGeom<TD, TH, TW, G> bg(geo);
  const int ISD = geo.is(0), ISH = geo.is(1), ISW = geo.is(2);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* box = smem;
  char* wl = smem + bg.BOX_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hh = lane >> 5;

  // ---- workgroup -> (tile, cout block, sample, group) ---------------------------------------------
  // Every kernel argument the decode and the staging addresses need is fetched in ONE batch here (the empty asm consumes them
  // all, so their scalar loads are issued together and waited for once) - left to itself the compiler loads each next to its
  // first use: eight dependent scalar round trips in front of the first global load.  The divisions of the decode use the
  // launcher's reciprocals (one s_mul_hi each instead of a ~25-instruction float-reciprocal sequence, ten times).
  const int TPW = PERSIST ? p.tiles_per_wg : 1;     // tiles per workgroup
  const int gx_ = p.gx, gy_ = p.gy, gz_ = p.gz, nsplit_ = p.nsplit, ngroups_ = p.d.ngroups, cf_ = p.cout_fastest;
  const int t1_ = p.tiles[1], t2_ = p.tiles[2];
  const unsigned mgy = p.mg_gy, mgx = p.mg_gxw, mgs = p.mg_nsplit, mgg = p.mg_ngroups, mgt2 = p.mg_t2, mgt1 = p.mg_t1;
  const unsigned mgb2 = p.mg_b2, mgb1 = p.mg_b1;
  const int brick_ = p.brick, gfast_ = p.group_fastest;
  const int lo0_ = p.d.lo[0], lo1_ = p.d.lo[1], lo2_ = p.d.lo[2], ldi_ = p.d.ldi;
  const int Cin = p.d.Cin, Cout = p.d.Cout, T = p.d.ntaps_total;
  const int Di = p.d.in_dims[0], Hi = p.d.in_dims[1], Wi = p.d.in_dims[2];
  const f16* in_ = p.in;
  const f16* w_ = p.w;
  const unsigned in_bytes_ = p.in_bytes, w_bytes_ = p.w_bytes;
  asm volatile("" ::"s"(in_), "s"(w_), "s"(in_bytes_), "s"(w_bytes_));
  asm volatile("" ::"s"(gx_), "s"(gy_), "s"(gz_), "s"(nsplit_), "s"(ngroups_), "s"(cf_), "s"(t1_), "s"(t2_), "s"(mgy), "s"(mgx),
               "s"(mgs), "s"(mgg), "s"(mgt2), "s"(mgt1), "s"(lo0_), "s"(lo1_), "s"(lo2_), "s"(ldi_), "s"(Cin), "s"(Cout), "s"(T),
               "s"(Di), "s"(Hi), "s"(Wi), "s"(TPW), "s"(mgb2), "s"(mgb1), "s"(brick_), "s"(gfast_));
  auto udiv = [](unsigned n, int d, unsigned m) -> unsigned { return d == 1 ? n : __umulhi(n, m); };
  const int gxw = PERSIST ? gx_ / TPW : gx_;          // workgroups along the m-tile index
  const unsigned mgxw = mgx;                          // (the launcher's reciprocal is that of gx / tiles_per_wg)
  const unsigned nwg = (unsigned)gxw * gy_ * gz_;
  unsigned lin = xcd_remap(blockIdx.x, nwg);
  int bx, by;
  int g_fast = -1;
  if (gfast_) {
    const unsigned q = udiv(lin, ngroups_, mgg);
    g_fast = lin - q * ngroups_;
    lin = q;
  }
  if (cf_) {
    // the cout blocks of one m-tile are neighbours in launch order (same XCD, same time): the second .. gy-th of them
    // find the tile's input box in L2 instead of HBM
    unsigned q = udiv(lin, gy_, mgy);
    by = lin - q * gy_;
    lin = q;
    q = udiv(lin, gxw, mgxw);
    bx = lin - q * gxw;
    lin = q;
  } else {
    unsigned q = udiv(lin, gxw, mgxw);
    bx = lin - q * gxw;
    lin = q;
    q = udiv(lin, gy_, mgy);
    by = lin - q * gy_;
    lin = q;
  }
  bx *= TPW;
  int bz = lin;
  int split = 0;
  if (nsplit_ > 1) {
    const unsigned q = udiv(bz, nsplit_, mgs);
    split = bz - q * nsplit_;
    bz = q;
  }
  const int n = gfast_ ? bz : (int)udiv(bz, ngroups_, mgg);
  const int g = gfast_ ? g_fast : bz - n * ngroups_;
  int tw_i, td_i, th_i;
  if (!PERSIST && brick_) {
    const unsigned b = (unsigned)bx >> 6, r = (unsigned)bx & 63u;
    const int nb2 = t2_ >> 2, nb1 = t1_ >> 2;
    const unsigned qb = udiv(b, nb2, mgb2);
    const int bw = b - qb * nb2;
    const int bd = udiv(qb, nb1, mgb1);
    const int bh = qb - bd * nb1;
    tw_i = bw * 4 + (r & 3);
    th_i = bh * 4 + ((r >> 2) & 3);
    td_i = bd * 4 + (r >> 4);
  } else {
    const unsigned q2 = udiv(bx, t2_, mgt2);
    tw_i = bx - q2 * t2_;
    td_i = udiv(q2, t1_, mgt1);
    th_i = q2 - td_i * t1_;
  }
  const int m0d = td_i * TD, m0h = th_i * TH;
  int m0w = tw_i * TW;      // (PERSIST: advances by TW per tile; the launcher guarantees the run stays inside one W row)
  const int cb0 = by * NB;  // first 32-wide cout block

  const nnz_conv_group grp = p.d.groups[g];
  const int nt = grp.ntaps;
  const int tb = grp.tap_begin;
  // The scalars the epilogue's index arithmetic uses, fetched HERE and pinned in SGPRs.  Left to itself the compiler re-reads a
  // kernel argument next to every use - inside the short-circuit bounds checks of the epilogue that was a scalar load plus a full
  // lgkmcnt(0) wait per comparison: five dependent round trips in front of each of the eight layer-below loads of the fused
  // norm-backward launches (tools/probes/conv_phase_probe.py: 10 000 cycles per workgroup between the last MFMA and the
  // accumulators' LDS image).
  const int mD0 = pin_uniform(p.d.m_dims[0]), mD1 = pin_uniform(p.d.m_dims[1]), mD2 = pin_uniform(p.d.m_dims[2]);
  const int oD0 = pin_uniform(p.d.out_dims[0]), oD1 = pin_uniform(p.d.out_dims[1]), oD2 = pin_uniform(p.d.out_dims[2]);
  const int oS0 = pin_uniform(p.d.out_stride[0]), oS1 = pin_uniform(p.d.out_stride[1]), oS2 = pin_uniform(p.d.out_stride[2]);
  const int oO0 = pin_uniform(grp.ooff[0]), oO1 = pin_uniform(grp.ooff[1]), oO2 = pin_uniform(grp.ooff[2]);
  const int ldo_ = pin_uniform(p.d.ldo), ldbx_ = pin_uniform(p.ldbx), accum_ = pin_uniform(p.d.accumulate);
  const int dbg_ = pin_uniform(p.dbg);

  // ---- per-thread staging addresses (independent of the channel slice) ----------------------------
  // The staging loads are BUFFER loads (descriptor = base + extent in SGPRs, 32-bit byte offset per lane, the slice's offset in
  // the scalar offset operand): a piece outside the volume carries an offset beyond the extent and the range check returns
  // zeros - one instruction per piece.  (As flat loads every piece was an exec-masked branch around a 64-bit address add, ~10
  // instructions each: tools/probes/conv_phase_probe.py priced the 15 loads of a slice at ~1 600 cycles in front of the MFMA
  // loop, a quarter of a slice's time, and 37 VGPRs + 14 SGPR pairs of addresses and masks.)
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(in_), 0, (int)in_bytes_, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(w_), 0, (int)w_bytes_, 0x00020000);
  const __amdgpu_buffer_rsrc_t bx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.bx), 0, (int)p.bx_bytes, 0x00020000);
  unsigned box_goff[LPT_BOX];  // byte offset into `in` (without channel slice), OOB = zero fill
  int box_loff[LPT_BOX];  // LDS byte offset, -1 = nothing to do
  {
    const int lod = m0d * ISD + lo0_, loh = m0h * ISH + lo1_, low = m0w * ISW + lo2_;
#pragma unroll
    for (int i = 0; i < LPT_BOX; ++i) {
      const int c = tid + i * 256;
      box_goff[i] = OOB;
      box_loff[i] = -1;
      if (c < bg.NBOXLOAD) {
        const int half = c & 1;
        const int s = c >> 1;
        const int bw = s % bg.BW;
        const int bh = (s / bg.BW) % bg.BH;
        const int bd = s / (bg.BW * bg.BH);
        const int id = lod + bd, ih = loh + bh, iw = low + bw;
        // LDS image: W-stride 2 de-interleaves the columns (even columns first) so that a tap's 8 lanes read 8
        // adjacent rows again; the half swizzle follows the row index the lanes step through (bh, or bh/2 for
        // H-stride 2).  Without this the stride-2 reads were 4-way bank conflicts (SQ_LDS_BANK_CONFLICT 48 %).
        const int bwp = ISW == 2 ? (bw >> 1) + (bw & 1) * ((bg.BW + 1) >> 1) : bw;
        const int swz = (ISH == 2 ? bh >> 1 : bh) & 1;
        box_loff[i] = ((bd * bg.BH + bh) * bg.PW + bwp) * 32 + ((half ^ swz) << 4);
        if ((unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi)
          box_goff[i] = ((unsigned)((((n * Di + id) * Hi + ih) * Wi + iw) * ldi_) + half * 8) * 2u;
      }
    }
  }
  // PERSIST: the staging addresses of the tile that starts at W position `w0` - the piece's box coordinates are decoded from
  // its LDS offset (stride-1 geometry only: bwp = bw), validity and global offset as above
  auto set_box_goff = [&](int w0) {
    const int lod = m0d * ISD + lo0_, loh = m0h * ISH + lo1_, low = w0 * ISW + lo2_;
#pragma unroll
    for (int i = 0; i < LPT_BOX; ++i) {
      box_goff[i] = OOB;
      if (box_loff[i] >= 0) {
        // (opaque copy: the decode is invariant across the tile loop and would otherwise be hoisted into ~32 live registers)
        int lo_ = box_loff[i];
        asm volatile("" : "+v"(lo_));
        const int r = lo_ >> 5;
        const int bw = r % bg.PW, bh = (r / bg.PW) % bg.BH, bd = r / (bg.PW * bg.BH);
        const int half = ((lo_ >> 4) & 1) ^ (bh & 1);
        const int id = lod + bd, ih = loh + bh, iw = low + bw;
        if ((unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi)
          box_goff[i] = ((unsigned)((((n * Di + id) * Hi + ih) * Wi + iw) * ldi_) + half * 8) * 2u;
      }
    }
  };
  const int nwchunks = NB * nt * 64;
  unsigned w_goff[C::LPT_W];  // byte offset of the thread's weight pieces inside one slice's block, OOB = none
#pragma unroll
  for (int i = 0; i < C::LPT_W; ++i) {
    const int c = tid + i * 256;
    w_goff[i] = OOB;
    if (c < nwchunks) {
      const int nb = c / (nt * 64);
      const int r = c - nb * nt * 64;
      w_goff[i] = (unsigned)(((cb0 + nb) * T + tb) * 512 + r * 8) * 2u;
    }
  }
  const unsigned w_slice_bytes = (unsigned)((Cout >> 5) * T) * 1024u;

  // ---- per-lane fragment addresses ----------------------------------------------------------------
  const int wm = wave % C::WAVES_M;
  const int wn = wave / C::WAVES_M;
  int vox_off[C::WM];  // LDS byte offset of the lane's voxel row for a tap of even h offset (odd: ^ 16)
  int out_vox[C::WM];  // element offset of the lane's output voxel, -1 if masked
  // tile-linear voxel index of fragment i of this wave.  Default: fragments are consecutive runs of 32 voxels.  DRE:
  // fragment i = depth plane 4 (wm >> 1) + i, h rows 4 (wm & 1) .. + 3.
  auto frag_voxel = [&](int i) -> int {
    if constexpr (DRE) return ((4 * (wm >> 1) + i) * TH + 4 * (wm & 1) + (l31 >> 3)) * TW + (l31 & 7);
    return (wm * C::WM + i) * 32 + l31;
  };
  auto set_out_vox = [&]() {
#pragma unroll
    for (int i = 0; i < C::WM; ++i) {
      const int v = frag_voxel(i);
      const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
      const int md = m0d + td, mh = m0h + th, mw = m0w + tw;
      const int od = md * oS0 + oO0;
      const int oh = mh * oS1 + oO1;
      const int ow = mw * oS2 + oO2;
      const bool ok = md < mD0 && mh < mD1 && mw < mD2 &&
                      od < oD0 && oh < oD1 && ow < oD2;
      out_vox[i] = ok ? (((n * oD0 + od) * oD1 + oh) * oD2 + ow) * ldo_
                      : -1;
    }
  };
#pragma unroll
  for (int i = 0; i < C::WM; ++i) {
    const int v = frag_voxel(i);
    const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
    const int base = (((td * ISD) * bg.BH + th * ISH) * bg.PW + tw) * 32;  // column tw of the (de-interleaved) image
    const int f = th & 1;
    vox_off[i] = base + ((hh ^ f) << 4);
  }
  set_out_vox();
  // weights: lane reads row (l&31) of [nb][t][32][32B]
  const int w_lane = l31 * 32 + ((hh ^ ((lane >> 4) & 1)) << 4);

  f32x16 acc[C::WN][C::WM];
#pragma unroll
  for (int a = 0; a < C::WN; ++a)
#pragma unroll
    for (int b = 0; b < C::WM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // Tap table in a VGPR: lane t holds tap t's box offset (bytes, multiple of 32) | h-offset parity (bit 4: the
  // swizzle flips the 16-byte half).  The tap loop fetches it with v_readlane - no scalar loads (their lgkmcnt(0)
  // would drain the LDS queue) and no memory at all inside the MFMA loop.
  typedef const int __attribute__((address_space(4))) * karg_ptr;                    // ConvDev is the only parameter
  const karg_ptr kargs = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  int tap_tab = 0;
  if (lane < nt) tap_tab = kargs[offsetof(ConvDev, tapenc) / 4 + tb + lane];

  u32x4 breg[LPT_BOX];
  u32x4 wreg[C::LPT_W];
  const int nkc = Cin >> 4;
  // consumer-side norm: {scale, shift} of a slice's 16 channels live in a two-slot LDS table behind box + weights.  Threads
  // 0..15 fetch slice kc + 1's pairs one slice ahead (tabreg) and drop them into slot (kc + 1) & 1 while everybody stages
  // slice kc from slot kc & 1 - two barriers separate every write of a slot from its reads, no extra barrier needed.
  float* intab = reinterpret_cast<float*>(smem + p.intab_off);
  f32x2 tabreg = {1.f, 0.f};
  auto load_tab = [&](int kc) {
    if (p.in_tab && tid < 16) {
      const int c = kc * 16 + tid;
      tabreg = f32x2{1.f, 0.f};
      if (c >= p.in_c0)
        tabreg = *reinterpret_cast<const f32x2*>(p.in_tab + ((size_t)n * (Cin - p.in_c0) + (c - p.in_c0)) * 4 + 2);
    }
  };
  auto store_tab = [&](int kc) {
    if (p.in_tab && tid < 16) *reinterpret_cast<f32x2*>(intab + (kc & 1) * 32 + 2 * tid) = tabreg;
  };
  auto issue_loads = [&](int kc) {
#pragma unroll
    for (int i = 0; i < LPT_BOX; ++i)
      breg[i] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)box_goff[i], kc * 32, 0);
#pragma unroll
    for (int i = 0; i < C::LPT_W; ++i)
      wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)w_goff[i], (int)((unsigned)kc * w_slice_bytes), 0);
  };
  // Consumer-side norm of the staged pieces between the slice's two barriers, CHANNEL PAIR by channel pair (every piece of a
  // thread holds the same 8 channels of the slice: c = tid + 256 i -> half = tid & 1; dword q of every piece = channels 2q,
  // 2q + 1 of that half): one 16-byte table read per pair, then 4 VALU operations per piece (common.hpp norm_lrelu_pair).
  // Padding pieces are normalised like the others and zeroed again when the box is not interior.
  // Measured alternatives (tools/bench_conv_layers.py --innorm, 32 -> 32 @128^3 forward, 0.291 ms without the norm): fp32
  // convert / select / convert arithmetic between the barriers 0.34 ms; this form +8 %; the same work moved INSIDE the MFMA
  // loop of the previous slice - piece by piece (+15 %) or as a software pipeline over the loop's last steps with the table
  // read one step ahead (+13 %) - is slower than between the barriers: the loop is issue-bound per wave (two waves per SIMD,
  // LDS fragment reads + MFMAs back to back), so every extra instruction costs its issue slot wherever it sits, and inside
  // the loop it also lengthens the read -> MFMA dependency chains.
  auto norm_pair = [&](int kc, int q) {
    const f32x4 tq = *reinterpret_cast<const f32x4*>(intab + (kc & 1) * 32 + (tid & 1) * 16 + 4 * q);
    const nnz_h2 sl = slope_pair(kc * 16 < p.in_c0 ? 1.f : p.in_slope);
#pragma unroll
    for (int i = 0; i < LPT_BOX; ++i) breg[i][q] = norm_lrelu_pair(breg[i][q], tq[0], tq[1], tq[2], tq[3], sl);
  };
  bool padded_box;   // some piece of this thread's box lies outside the volume (uniform over the workgroup's tile)
  auto set_padded_box = [&](int w0) {
    const int lod = m0d * ISD + lo0_, loh = m0h * ISH + lo1_, low = w0 * ISW + lo2_;
    padded_box = lod < 0 || loh < 0 || low < 0 || lod + bg.BD > Di || loh + bg.BH > Hi || low + bg.BW > Wi;
  };
  set_padded_box(m0w);
  auto write_lds = [&](int kc) {
    if (p.in_tab) {
#pragma unroll
      for (int q = 0; q < 4; ++q) norm_pair(kc, q);
      if (padded_box) {
#pragma unroll
        for (int i = 0; i < LPT_BOX; ++i)
          if (box_goff[i] == OOB) breg[i] = u32x4{0u, 0u, 0u, 0u};
      }
      store_tab(kc + 1);
    }
#pragma unroll
    for (int i = 0; i < LPT_BOX; ++i)
      if (box_loff[i] >= 0) *reinterpret_cast<u32x4*>(box + box_loff[i]) = breg[i];
#pragma unroll
    for (int i = 0; i < C::LPT_W; ++i) {
      const int c = tid + i * 256;
      if (c < nwchunks) {
        const int half = c & 1;
        const int row = c >> 1;  // (nb*nt + t)*32 + co
        const int co = row & 31;
        *reinterpret_cast<u32x4*>(wl + row * 32 + ((half ^ ((co >> 4) & 1)) << 4)) = wreg[i];
      }
    }
  };
  // fragments of tap t: A = weights (rows = cout), B = box window (voxels on the lanes)
  auto load_frags = [&](int t, f16x8 (&a)[C::WN], f16x8 (&b)[C::WM]) {
    const int enc = __builtin_amdgcn_readlane(tap_tab, t);
    const int toff = enc & ~16, flip = enc & 16;
#pragma unroll
    for (int i = 0; i < C::WN; ++i)
      a[i] = *reinterpret_cast<const f16x8*>(wl + (((wn * C::WN + i) * nt + t) << 10) + w_lane);
#pragma unroll
    for (int i = 0; i < C::WM; ++i) b[i] = *reinterpret_cast<const f16x8*>(box + ((vox_off[i] + toff) ^ flip));
  };
  auto mfma_all = [&](const f16x8 (&a)[C::WN], const f16x8 (&b)[C::WM]) {
#pragma unroll
    for (int i = 0; i < C::WN; ++i)
#pragma unroll
      for (int j = 0; j < C::WM; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
  };

  // depth-reuse loop: per (kh, kw) row r the box offset without its depth part (lane r of rtab)
  int rtab = 0;
  if constexpr (DRE) {
    if (lane < 9) rtab = kargs[offsetof(ConvDev, rowenc) / 4 + lane];
  }
  const int plane_bytes = bg.BH * bg.PW * 32;
  // addresses: flip only toggles bit 4 of the lane's base (every other term is a multiple of 32), so the six planes and
  // the three depth taps of a row are immediate offsets from ONE voxel base and ONE weight base per row
  const int vb_even = vox_off[0], vb_odd = vox_off[0] ^ 16;
  // a row (kh, kw) is processed in two halves so that only three voxel fragments per register set are live:
  //   half A: weights of the three depth taps + input planes 0..2  -> 6 MFMAs;   half B: input planes 3..5 -> 6 MFMAs
  auto load_a = [&](int r, f16x8 (&a)[3], f16x8 (&x)[3]) {
    const int enc = __builtin_amdgcn_readlane(rtab, r);
    const char* wrow = wl + w_lane + (r << 10);
    const char* xrow = box + ((enc & 16) ? vb_odd : vb_even) + (enc & ~16);
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const f16x8*>(wrow + (DFLIP ? 2 - q : q) * 9 * 1024);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const f16x8*>(xrow + pl * plane_bytes);
  };
  auto load_b = [&](int r, f16x8 (&x)[3]) {
    const int enc = __builtin_amdgcn_readlane(rtab, r);
    const char* xrow = box + ((enc & 16) ? vb_odd : vb_even) + (enc & ~16);
#pragma unroll
    for (int pl = 3; pl < 6; ++pl) x[pl - 3] = *reinterpret_cast<const f16x8*>(xrow + pl * plane_bytes);
  };
  auto mfma_half = [&](const f16x8 (&a)[3], const f16x8 (&x)[3], int pl0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int j = pl0 + i - q;  // output plane of (input plane pl0 + i, depth offset q)
        if (j >= 0 && j < 4) acc[0][j] = mfma32(a[q], x[i], acc[0][j]);
      }
  };

  constexpr int PPV = NB * 4;  // 16-byte pieces per voxel
  // fused norm-backward reductions (ConvDev::bx): the loads of the layer-below tile are issued at the start of the epilogue,
  // before the accumulators go through LDS, INTO the staging registers of the main loop (free now).  The epilogue's barriers
  // are lds_barrier(): __syncthreads() would drain these loads.  (Tried and dropped, tools/probes/normred_epilogue_probe.py:
  // issuing them inside the last reduction slice - no gain, and separate destination arrays made the allocator spill 550
  // registers; a one-workgroup finishing kernel instead of the ticket protocol - no gain; 16 accumulator replicas - the
  // finishing reads cost more than the contention they remove.)
  constexpr int BX_PARTS = 256 / PPV;
  constexpr int BX_NIT = (TD * TH * TW + BX_PARTS - 1) / BX_PARTS;
  static_assert(BX_NIT <= LPT_BOX + C::LPT_W, "layer-below tile must fit the staging registers");
  const bool bxmode = p.acc && p.bx;
  const int kc0 = nsplit_ > 1 ? split * p.kper : 0;
  const int kc1 = nsplit_ > 1 ? (kc0 + p.kper < nkc ? kc0 + p.kper : nkc) : nkc;
  NNZ_TS(1);
  // (table first: store_tab below then waits for the OLDEST load only - behind the slice's 15 loads it waited for all of them,
  //  and the barrier + normalisation of slice 0 could not start before the last piece had landed)
  load_tab(kc0);
  issue_loads(kc0);
  // (requesting the next tile's first slice inside this tile's last MFMA loop keeps 32 staging registers alive through the
  //  epilogue, which then spills them: measured slower than the late request below)
  constexpr bool EARLY_PREFETCH = false;
  bool prefetched = false;   // PERSIST: the next tile's first slice was requested inside this tile's last MFMA loop
  const int tid_outer = tid;
  auto tile_body = [&](bool last_tile) {
  // PERSIST: everything the epilogue derives from the thread index is recomputed per tile from an opaque copy - left to itself
  // the compiler hoists those (loop-invariant) values out of the tile loop and spills ~180 registers (measured: 2.5x slower)
  int tid = tid_outer;
  if constexpr (PERSIST) asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hh = lane >> 5;
  const int wm = wave % C::WAVES_M;
  const int wn = wave / C::WAVES_M;
  (void)l31; (void)wm; (void)wn;
  auto bx_index = [&](int k) -> long {   // row index of item k of this thread in the output / layer-below tensors, -1: none
    const int v = tid / PPV + k * BX_PARTS;
    const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
    const int md = m0d + td, mh = m0h + th, mw = m0w + tw;
    const int od = md * oS0 + oO0;
    const int oh = mh * oS1 + oO1;
    const int ow = mw * oS2 + oO2;
    const bool ok = (v < TD * TH * TW) & (md < mD0) & (mh < mD1) & (mw < mD2) & (od < oD0) & (oh < oD1) & (ow < oD2);
    return ok ? ((long)((n * oD0 + od) * oD1 + oh) * oD2 + ow) : -1;
  };
  auto issue_bx = [&]() {
    const int co = cb0 * 32 + (tid % PPV) * 8;
#pragma unroll
    for (int k = 0; k < BX_NIT; ++k) {
      const long idx = bx_index(k);
      // buffer load like the staging loads: rows outside the tile carry an offset beyond the extent and come back as zeros
      const unsigned off = idx >= 0 ? ((unsigned)idx * (unsigned)ldbx_ + (unsigned)co) * 2u : OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc, (int)off, 0, 0);
      if (k < LPT_BOX) breg[k < LPT_BOX ? k : 0] = v;
      else wreg[k >= LPT_BOX ? k - LPT_BOX : 0] = v;
    }
  };
#pragma unroll
  for (int a = 0; a < C::WN; ++a)
#pragma unroll
    for (int b = 0; b < C::WM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  store_tab(kc0);
  if (kc0 + 1 < kc1) load_tab(kc0 + 1);
  prefetched = false;
  for (int kc = kc0; kc < kc1; ++kc) {
    __syncthreads();  // all waves finished reading the previous slice
    write_lds(kc);
    __syncthreads();
    NNZ_TS(2 + 2 * (kc - kc0) < 12 ? 2 + 2 * (kc - kc0) : 15);
    // (table first: its destination register is rewritten here, and the compiler guards that with a wait for every load in
    //  flight - before the slice's 15 loads are issued that wait is free, after them it would hold wave 0 until they land)
    if (kc + 2 < kc1) load_tab(kc + 2);
    if (kc + 1 < kc1) issue_loads(kc + 1);
    // fused norm-backward reductions: the layer-below tile is requested under the LAST slice's MFMA loop - the staging registers
    // are free there (no next slice to prefetch) and the epilogue finds the rows landed instead of waiting a memory round trip
    // (tools/probes/conv_phase_probe.py: ~5 000 cycles per workgroup between the last MFMA and the accumulators' LDS image)
    // (depth-reuse instantiations only: elsewhere the longer live ranges cost a wave of occupancy or spill)
    else if (DRE && bxmode) issue_bx();
    if (PERSIST && EARLY_PREFETCH && kc + 1 == kc1 && !last_tile && !bxmode) {
      // forward launches: the epilogue leaves the staging registers alone - request the next tile's first slice now
      set_box_goff(m0w + TW);
      set_padded_box(m0w + TW);
      load_tab(kc0);
      issue_loads(kc0);
      prefetched = true;
    }

    if constexpr (DRE) {
      // Software pipeline by half rows: the reads of the NEXT half are issued as a group before the six MFMAs of the
      // current one, so every MFMA's operands were requested ~200 cycles earlier.  The scheduling barriers keep the
      // compiler from sinking the reads next to their uses (it does, to save registers, and then every MFMA waits on
      // lgkmcnt(0) for a read issued just before it).
      f16x8 a0[3], a1[3], xa[3], xb[3];
#if NNZ_SETPRIO
      __builtin_amdgcn_s_setprio(NNZ_SETPRIO);
#endif
      load_a(0, a0, xa);
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        load_b(r, xb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, xa, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(r + 1, a1, xa);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, xb, 3);
        __builtin_amdgcn_sched_barrier(0);
        load_b(r + 1, xb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, xa, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(r + 2, a0, xa);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, xb, 3);
        __builtin_amdgcn_sched_barrier(0);
      }
      load_b(8, xb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(a0, xa, 0);
      mfma_half(a0, xb, 3);
#if NNZ_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    } else {
      // software pipeline over the taps: the LDS reads of tap t+1 are in flight while tap t's MFMAs issue
      f16x8 a0[C::WN], b0[C::WM], a1[C::WN], b1[C::WM];
      load_frags(0, a0, b0);
      int t = 0;
      for (; t + 2 <= nt; t += 2) {
        load_frags(t + 1, a1, b1);
        mfma_all(a0, b0);
        if (t + 2 < nt) load_frags(t + 2, a0, b0);
        mfma_all(a1, b1);
      }
      if (t < nt) mfma_all(a0, b0);
    }
    NNZ_TS(3 + 2 * (kc - kc0) < 12 ? 3 + 2 * (kc - kc0) : 15);
  }

  if (p.part) {
    // split-K: the lane's accumulator quads are 4 consecutive couts of one voxel -> 16-byte fp32 stores, no transposition
    const long vox_per_n = (long)oD0 * oD1 * oD2;
#pragma unroll
    for (int j = 0; j < C::WM; ++j) {
      const int v = frag_voxel(j);
      const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
      const int md = m0d + td, mh = m0h + th, mw = m0w + tw;
      const int od = md * oS0 + oO0;
      const int oh = mh * oS1 + oO1;
      const int ow = mw * oS2 + oO2;
      if (!(md < mD0 && mh < mD1 && mw < mD2 && od < oD0 &&
            oh < oD1 && ow < oD2))
        continue;
      const long vox = ((long)od * oD1 + oh) * oD2 + ow;
      float* dst = p.part + (((long)split * p.d.N + n) * vox_per_n + vox) * Cout;
#pragma unroll
      for (int i = 0; i < C::WN; ++i) {
        const int nbl = wn * C::WN + i;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4*>(dst + (cb0 + nbl) * 32 + 4 * hh + 8 * q) = o;
        }
      }
    }
    return;
  }
  // ---- epilogue: D[row = cout][col = voxel]; lane holds couts (r&3) + 8(r>>2) + 4hh of its voxel -----
  // The tile is transposed through LDS (the box/weight region is free now) so that global stores are whole
  // 16-byte pieces in voxel-major order: a wave writes full 128-byte lines.  (Direct 8-byte stores from the
  // accumulator layout measured WRITE_SIZE = 1.5x the output bytes: partial-line writes.)
  constexpr int ROWB = NB * 64 + 16;  // LDS bytes per voxel row (+16: spreads the b64 writes over the banks)
  static_assert(TD * TH * TW * ROWB <= C::OUT_BYTES, "output image must fit the staging LDS");
  // the old gradient (accumulating launches), the layer-below table and the row indices: requested before the staging where
  // the accumulators leave room for them (one cout block per workgroup), after it otherwise (the two-block tiles sit at 256
  // registers: asking earlier spilled 22 of them)
  f16x8 bold[BX_NIT];
  f32x4 btab[8];
  long boidx[BX_NIT];
  auto bx_side_loads = [&]() {
    const int co = cb0 * 32 + (tid % PPV) * 8;
#pragma unroll
    for (int k = 0; k < BX_NIT; ++k) {
      boidx[k] = bx_index(k);
      if (accum_ && boidx[k] >= 0) bold[k] = *reinterpret_cast<const f16x8*>(p.out + boidx[k] * ldo_ + co);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) btab[e] = *reinterpret_cast<const f32x4*>(p.bstat + ((size_t)n * Cout + co + e) * 4);
  };
  if (!DRE && bxmode) issue_bx();
  if (NB == 1 && bxmode) bx_side_loads();
  NNZ_TS(9);
  lds_barrier();  // every wave is done reading box / weights (LDS-only barrier: global loads stay in flight)
#pragma unroll
  for (int i = 0; i < C::WN; ++i) {
    const int nbl = wn * C::WN + i;
#pragma unroll
    for (int j = 0; j < C::WM; ++j) {
      const int v = frag_voxel(j);
      char* row = smem + v * ROWB + (nbl * 32 + 4 * hh) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};  // bias joins in fp32, before the single rounding to fp16
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + (cb0 + nbl) * 32 + 4 * hh + 8 * q);
        f16x4 o = {(f16)(acc[i][j][4 * q + 0] + bv[0]), (f16)(acc[i][j][4 * q + 1] + bv[1]),
                   (f16)(acc[i][j][4 * q + 2] + bv[2]), (f16)(acc[i][j][4 * q + 3] + bv[3])};
        *reinterpret_cast<f16x4*>(row + 16 * q) = o;
      }
    }
  }
  lds_barrier();
  NNZ_TS(12);
  if (NB != 1 && bxmode) bx_side_loads();
  constexpr int NPIECE = TD * TH * TW * PPV;
  // knob 10: wave 0's fixed-point adds are issued AFTER its share of the output stores.  In front of them the 128 adds (all
  // workgroups of the launch hit the same 256 words) backed up the wave's memory queue and its stores - the workgroup's last -
  // left twice as late (tools/probes/conv_phase_probe.py: 5 900 against 2 900 cycles)
  long fx_rec = -1;
  double fx_v1 = 0.0, fx_v2 = 0.0, fx_v3 = 0.0;
  if (p.stats) {
    // InstanceNorm statistics of this tile from the fp16 image (what the normalisation will read): a thread sums 8
    // channels over its share of the voxels, a [parts][2*NC] slab behind the image folds the shares, one atomic per
    // (tile, channel, moment).  Saves the separate read of the whole conv output.
    constexpr int NC = NB * 32, NVOX = TD * TH * TW;
    constexpr int PARTS = 256 / PPV;
    float* slab = reinterpret_cast<float*>(smem + NVOX * ROWB);
    const int part = tid / PPV, c8 = tid % PPV;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    for (int v = part; v < NVOX; v += PARTS) {
      const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
      if (m0d + td < mD0 && m0h + th < mD1 && m0w + tw < mD2) {
        const f16x8 val = *reinterpret_cast<const f16x8*>(smem + v * ROWB + c8 * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (float)val[e];
          s1[e] += x;
          s2[e] += x * x;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      slab[part * (2 * NC) + (c8 * 8 + e) * 2 + 0] = s1[e];
      slab[part * (2 * NC) + (c8 * 8 + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    if (tid < 2 * NC) {
      float t = 0.f;
      for (int q = 0; q < PARTS; ++q) t += slab[q * (2 * NC) + tid];
      atomicAdd(p.stats + ((size_t)n * Cout + cb0 * 32) * 2 + tid, t);
    }
  } else if (bxmode) {
    // InstanceNorm + LeakyReLU backward reductions of the layer below on this tile (see ConvDev::bx).  Thread = (voxel share,
    // 8-channel group): the same (voxel, piece) items it would store in the loop at the end, so the final fp16 values
    // (after the optional accumulate) are formed once, used for the sums and stored from here.  Packed fp32 arithmetic
    // (two channels per instruction): the sums are ~10 VALU operations per element on 16 elements x 8 voxels per lane.
    constexpr int NC = NB * 32, NVOX = TD * TH * TW;
    float* slab = reinterpret_cast<float*>(smem + NVOX * ROWB);
    const int part = tid / PPV, c8 = tid % PPV;
    const int co = cb0 * 32 + c8 * 8;
    f32x2 mean2[4], scale2[4], shift2[4], s1[4], s2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 t0 = btab[2 * e], t1 = btab[2 * e + 1];
      mean2[e] = f32x2{t0[0], t1[0]};
      scale2[e] = f32x2{t0[2], t1[2]};
      shift2[e] = f32x2{t0[3], t1[3]};
      s1[e] = s2[e] = f32x2{0.f, 0.f};
    }
    const f32x2 slope2 = {p.slope, p.slope};
#pragma unroll
    for (int k = 0; k < BX_NIT; ++k) {
      if (boidx[k] < 0) continue;
      const int v = part + k * BX_PARTS;
      f16x8 val = *reinterpret_cast<const f16x8*>(smem + v * ROWB + c8 * 16);
      if (accum_) {
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = (f16)((float)val[e] + (float)bold[k][e]);
      }
      *reinterpret_cast<f16x8*>(p.out + boidx[k] * ldo_ + co) = val;
      if (dbg_ & 2) continue;
      const u32x4 xraw = k < LPT_BOX ? breg[k < LPT_BOX ? k : 0] : wreg[k >= LPT_BOX ? k - LPT_BOX : 0];
      const f16x8 xk = __builtin_bit_cast(f16x8, xraw);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 x = {(float)xk[2 * e], (float)xk[2 * e + 1]};
        const f32x2 g = {(float)val[2 * e], (float)val[2 * e + 1]};
        const f32x2 pre = x * scale2[e] + shift2[e];
        const f32x2 gs = g * slope2;
        const f32x2 gp = {pre[0] > 0.f ? g[0] : gs[0], pre[1] > 0.f ? g[1] : gs[1]};
        s1[e] += gp;
        s2[e] += gp * (x - mean2[e]);   // times rstd once per channel, below
      }
    }
    NNZ_TS(10);
#pragma unroll
    for (int off = PPV; off < 64; off <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e][0] += __shfl_xor(s1[e][0], off, 64);
        s1[e][1] += __shfl_xor(s1[e][1], off, 64);
        s2[e][0] += __shfl_xor(s2[e][0], off, 64);
        s2[e][1] += __shfl_xor(s2[e][1], off, 64);
      }
    if (lane < PPV) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<f32x4*>(slab + wave * (2 * NC) + (lane * 8 + 2 * e) * 2) =
            f32x4{s1[e][0], s2[e][0] * btab[2 * e][1], s1[e][1], s2[e][1] * btab[2 * e + 1][1]};   // s2 *= rstd
      }
    }
    __syncthreads();
    NNZ_TS(11);
    if (wave != 0 || (dbg_ & 4)) return;
    const long nrec = (long)p.d.N * Cout * 2;
    for (int c = lane; c < NC; c += 64) {
      const float S1 = (slab[c * 2] + slab[2 * NC + c * 2]) + (slab[4 * NC + c * 2] + slab[6 * NC + c * 2]);
      const float S2 = (slab[c * 2 + 1] + slab[2 * NC + c * 2 + 1]) + (slab[4 * NC + c * 2 + 1] + slab[6 * NC + c * 2 + 1]);
      const long ch = (long)n * Cout + cb0 * 32 + c;
      fx_add(p.acc, ch * 2, nrec, blockIdx.x, (double)S1);
      fx_add(p.acc, ch * 2 + 1, nrec, blockIdx.x, (double)S2);
    }
    NNZ_TS(13);
    if (!p.sep_finish && last_tile && last_workgroup_wave(p.counter, nwg)) {
      const double V = (double)oD0 * oD1 * oD2;
      for (int c = lane; c < Cout; c += 64) {
        double sg = 0.0, sb = 0.0;
        for (int nn = 0; nn < p.d.N; ++nn) {
          const long i = (long)nn * Cout + c;
          double r[2];
          fx_take_n<2>(p.acc, i * 2, nrec, r);
          p.nred[i * 2 + 0] = (float)(r[0] / V);
          p.nred[i * 2 + 1] = (float)(r[1] / V);
          sb += r[0];
          sg += r[1];
        }
        if (p.dgamma) {
          p.dgamma[c] = (float)sg;
          p.dbeta[c] = (float)sb;
        }
      }
    }
    NNZ_TS(14);
    return;
  } else if (p.acc) {
    // Deterministic and cancellation-free variant: moments about a PILOT value per channel (the tile's first voxel), folded
    // in a fixed order inside the workgroup, re-centred in double and added to the sample's fixed-point accumulators (integer
    // adds commute: the result does not depend on which workgroup finishes first).  sum x^2 is exact to ~1e-16 relative, so
    // var = E[x^2] - mean^2 formed in double by the finalising workgroup stays good for |mean| / std up to ~1e4.
    constexpr int NC = NB * 32, NVOX = TD * TH * TW;
    constexpr int PARTS = 256 / PPV;
    static_assert(PARTS * 2 * NC >= 12 * NC, "the statistics slab holds the four waves' pairs and their third sums");
    float* slab = reinterpret_cast<float*>(smem + NVOX * ROWB);
    const int RPC = p.mfma_moments ? 3 : 2;   // accumulator records per (sample, channel)
    const bool full_tile = m0d + TD <= mD0 && m0h + TH <= mD1 && m0w + TW <= mD2;
    if (full_tile && p.mfma_moments) {
      // Tiles inside the volume (all but the last ones of an axis): the moments on the matrix cores.  A wave takes NVOX / 4
      // voxels in 16-voxel steps; the transposed LDS read (ds_read_b64_tr_b16, as in conv_wgrad.hip) delivers the image as an
      // MFMA fragment F[channel][voxel] - the same registers serve as A and as B operand - the pilot is subtracted in packed
      // fp16 (d = x - K: exact while x and K lie within a factor of two of each other, i.e. in the cancellation-prone case of
      // a large mean; rounded to 11 bits otherwise - which is why the mean comes from a third, exact sum, see knob 11), and
      //   D2 = F F^T (diagonal: sum d^2)      D1 = F 1 (any column: sum d)
      // accumulate in fp32 in the MFMA's fixed order.  16 transposed reads + 32 packed subtractions + 16 MFMAs per wave replace
      // ~360 VALU instructions of convert / subtract / add / FMA per thread and the 64-shuffle lane fold
      // (tools/probes/conv_phase_probe.py: 4 600 + 2 300 cycles per workgroup).  Channel c's two sums sit in lane (c, hh =
      // (c >> 2) & 1), accumulator register (c & 3) + 4 (c >> 3).
      constexpr int KB_W = NVOX / 64;   // 16-voxel steps per wave
      static_assert(NVOX % 64 == 0, "voxel steps must split over the four waves");
      const int qrow = (lane & 15) >> 2;
      const int chan_byte = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
      const f16x8 ones = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f16 kh = *reinterpret_cast<const f16*>(smem + (nb * 32 + l31) * 2);   // pilot of the lane's channel
        const nnz_h2 k2 = {kh, kh};
        f32x16 d1, d2, dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[r] = d2[r] = dx[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB_W; ++kb) {
          const int v0 = (wave * KB_W + kb) * 16 + 8 * hh;
          const char* src = smem + (v0 + qrow) * ROWB + nb * 64 + chan_byte;
          union { i16x4 v[2]; nnz_h2 h[4]; f16x8 f; } u;
          u.v[0] = lds_read_tr16(src);
          u.v[1] = lds_read_tr16(src + 4 * ROWB);
          dx = mfma32(u.f, ones, dx);        // sum x of the UNROUNDED values: the table's mean stays exact
#pragma unroll
          for (int q = 0; q < 4; ++q) u.h[q] = u.h[q] - k2;
          d2 = mfma32(u.f, u.f, d2);
          d1 = mfma32(u.f, ones, d1);
        }
        const int rsel = (l31 & 3) + 4 * (l31 >> 3);
        float s1 = 0.f, s2 = 0.f, sx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r == rsel) {
            s1 = d1[r];
            s2 = d2[r];
            sx = dx[r];
          }
        if (hh == ((l31 >> 2) & 1)) {
          *reinterpret_cast<f32x2*>(slab + wave * (2 * NC) + (nb * 32 + l31) * 2) = f32x2{s1, s2};
          slab[8 * NC + wave * NC + nb * 32 + l31] = sx;       // third sums behind the four waves' pairs
        }
      }
      NNZ_TS(6);
    } else {
    const int part = tid / PPV, c8 = tid % PPV;
    float K[8], s1[8], s2[8];
    {
      const f16x8 k0 = *reinterpret_cast<const f16x8*>(smem + c8 * 16);  // voxel (0, 0, 0) of the tile: always valid
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        K[e] = (float)k0[e];
        s1[e] = s2[e] = 0.f;
      }
    }
    for (int v = part; v < NVOX; v += PARTS) {
      const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
      if (m0d + td < mD0 && m0h + th < mD1 && m0w + tw < mD2) {
        const f16x8 val = *reinterpret_cast<const f16x8*>(smem + v * ROWB + c8 * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (float)val[e] - K[e];
          s1[e] += x;
          s2[e] += x * x;
        }
      }
    }
    NNZ_TS(6);   // (sub-phase stamps 6..11 are meaningful for <= 2-slice launches only: deeper ones use these slots for slices)
    // fold the wave's 64 / PPV voxel shares with lane exchanges (fixed order), one slab row per wave, then wave 0 alone
    // finishes: per channel the four waves' sums, the double re-centring, the fixed-point adds, the ticket and - if this
    // was the launch's last workgroup - the table.  Waves 1-3 go straight on to the output stores: nobody waits for the
    // ticket's round trip (the first version stalled every workgroup ~2 us there: +31 us per full-resolution launch).
    constexpr int PPW = 64 / PPV;  // voxel shares per wave
#pragma unroll
    for (int off = PPV; off < 64; off <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], off, 64);
        s2[e] += __shfl_xor(s2[e], off, 64);
      }
    static_assert(PPW * PPV == 64, "lane = (share, channel group)");
    if (lane < PPV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        slab[wave * (2 * NC) + (lane * 8 + e) * 2 + 0] = s1[e];
        slab[wave * (2 * NC) + (lane * 8 + e) * 2 + 1] = s2[e];
      }
    }
    }   // VALU path
    __syncthreads();
    NNZ_TS(7);
    if (wave == 0) {
      for (int c = lane; c < NC; c += 64) {
        const float S1 = (slab[c * 2] + slab[2 * NC + c * 2]) + (slab[4 * NC + c * 2] + slab[6 * NC + c * 2]);
        const float S2 = (slab[c * 2 + 1] + slab[2 * NC + c * 2 + 1]) + (slab[4 * NC + c * 2 + 1] + slab[6 * NC + c * 2 + 1]);
        const int cd = mD0 - m0d < TD ? mD0 - m0d : TD;
        const int ch = mD1 - m0h < TH ? mD1 - m0h : TH;
        const int cw = mD2 - m0w < TW ? mD2 - m0w : TW;
        const double cnt = (double)(cd * ch * cw);
        const double k = (double)(float)*reinterpret_cast<const f16*>(smem + c * 2);
        // knob 11: THREE sums per (sample, channel) - sum and sum of squares of the pilot-centred values as the tiles formed
        // them (fp16-rounded deviations on the MFMA path: a consistent pair, so the variance keeps its precision) and the exact
        // sum of x for the mean (MFMA tiles: F 1 on the unrounded fragment; VALU tiles: the same number as the first sum)
        const long rec = ((long)n * Cout + cb0 * 32 + c) * RPC, nrec = (long)p.d.N * Cout * RPC;
        const double v1 = (double)S1 + cnt * k, v2 = (double)S2 + 2.0 * k * (double)S1 + cnt * k * k;
        double v3 = v1;
        if (full_tile && p.mfma_moments)
          v3 = (double)((slab[8 * NC + c] + slab[9 * NC + c]) + (slab[10 * NC + c] + slab[11 * NC + c]));
        if (p.sep_finish) {   // (NC <= 64: one channel per lane) the adds follow this wave's stores, see below
          fx_rec = rec;
          fx_v1 = v1;
          fx_v2 = v2;
          fx_v3 = v3;
        } else {
          fx_add(p.acc, rec, nrec, blockIdx.x, v1);
          fx_add(p.acc, rec + 1, nrec, blockIdx.x, v2);
          if (RPC == 3) fx_add(p.acc, rec + 2, nrec, blockIdx.x, v3);
        }
      }
      NNZ_TS(8);
      if (!p.sep_finish && last_tile && last_workgroup_wave(p.counter, nwg)) {
        const double V = (double)mD0 * mD1 * mD2;
        const long nrec = (long)p.d.N * Cout * RPC;
        for (int i = lane; i < p.d.N * Cout; i += 64) {
          double mom[3];
          if (RPC == 3) {
            fx_take_n<3>(p.acc, (long)i * 3, nrec, mom);
          } else {
            double m2[2];
            fx_take_n<2>(p.acc, (long)i * 2, nrec, m2);
            mom[0] = m2[0]; mom[1] = m2[1]; mom[2] = m2[0];
          }
          const double mean = mom[2] / V, mean_r = mom[0] / V;
          double var = mom[1] / V - mean_r * mean_r;
          var = var < 0.0 ? 0.0 : var;  // (NaN stays NaN)
          const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
          const int c = i % Cout;
          const float sc = rstd * p.gamma[c];
          const f32x4 o = {(float)mean, rstd, sc, p.beta[c] - (float)mean * sc};
          *reinterpret_cast<f32x4*>(p.nstat + (size_t)i * 4) = o;
        }
      }
    }
  }
  NNZ_TS(13);
#pragma unroll 2
  for (int c = tid; c < NPIECE; c += 256) {
    const int v = c / PPV, part = c % PPV;
    const int tw = v % TW, th = (v / TW) % TH, td = v / (TW * TH);
    const int md = m0d + td, mh = m0h + th, mw = m0w + tw;
    const int od = md * oS0 + oO0;
    const int oh = mh * oS1 + oO1;
    const int ow = mw * oS2 + oO2;
    if (!(md < mD0 && mh < mD1 && mw < mD2 && od < oD0 &&
          oh < oD1 && ow < oD2))
      continue;
    const int co = cb0 * 32 + part * 8;
    f16* dst = p.out + ((size_t)((n * oD0 + od) * oD1 + oh) * oD2 + ow) * ldo_ + co;
    f16x8 val = *reinterpret_cast<const f16x8*>(smem + v * ROWB + part * 16);
    if (accum_) {
      const f16x8 old = *reinterpret_cast<const f16x8*>(dst);
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = (f16)((float)val[e] + (float)old[e]);
    }
    *reinterpret_cast<f16x8*>(dst) = val;
  }
  if (fx_rec >= 0) {
    const int rpc = p.mfma_moments ? 3 : 2;
    const long nrec = (long)p.d.N * Cout * rpc;
    fx_add(p.acc, fx_rec, nrec, blockIdx.x, fx_v1);
    fx_add(p.acc, fx_rec + 1, nrec, blockIdx.x, fx_v2);
    if (rpc == 3) fx_add(p.acc, fx_rec + 2, nrec, blockIdx.x, fx_v3);
  }
  NNZ_TS(14);
  };   // tile_body

  for (int it = 0; it < TPW; ++it) {
    const bool last_tile = it + 1 == TPW;
    tile_body(last_tile);
    if (!last_tile) {
      if (!prefetched) {   // data-gradient launches with the fused reductions: the epilogue used the staging registers
        set_box_goff(m0w + TW);
        set_padded_box(m0w + TW);
        load_tab(kc0);
        issue_loads(kc0);
      }
      m0w += TW;
    }
  }
}

// Knob 10: what the launch's last workgroup did behind its ticket, as kernels of their own (same arithmetic, same bits).  The
// kernel boundary orders them behind every workgroup's adds; one thread per record, so the read-and-reset round trips of all
// records overlap instead of queueing in one wave (Cout = 320: ten dependent round trips in the launch's tail before).
struct StatsFinish {
  FxAcc* acc;
  float* nstat;
  const float* gamma;
  const float* beta;
  float eps;
  int N, Cout;
  double V;
  int rpc;   // records per (sample, channel): 2, or 3 with the exact sum of x behind the pilot-centred pair (knob 11)
};
__global__ __launch_bounds__(64) void conv_stats_finish_kernel(StatsFinish a) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.N * a.Cout) return;
  const long nrec = (long)a.N * a.Cout * a.rpc;
  double mom[3];
  if (a.rpc == 3) {
    fx_take_n<3>(a.acc, (long)i * 3, nrec, mom);
  } else {
    double m2[2];
    fx_take_n<2>(a.acc, (long)i * 2, nrec, m2);
    mom[0] = m2[0]; mom[1] = m2[1]; mom[2] = m2[0];
  }
  const double mean = mom[2] / a.V, mean_r = mom[0] / a.V;
  double var = mom[1] / a.V - mean_r * mean_r;
  var = var < 0.0 ? 0.0 : var;  // (NaN stays NaN)
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const int c = i % a.Cout;
  const float sc = rstd * a.gamma[c];
  const f32x4 o = {(float)mean, rstd, sc, a.beta[c] - (float)mean * sc};
  *reinterpret_cast<f32x4*>(a.nstat + (size_t)i * 4) = o;
}
struct NormRedFinish {
  FxAcc* acc;
  float* nred;
  float* dgamma;
  float* dbeta;
  int N, Cout;
  double V;
};
__global__ __launch_bounds__(64) void conv_normred_finish_kernel(NormRedFinish a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.Cout) return;
  const long nrec = (long)a.N * a.Cout * 2;
  double sg = 0.0, sb = 0.0;
  for (int nn = 0; nn < a.N; ++nn) {
    const long i = (long)nn * a.Cout + c;
    double r[2];
    fx_take_n<2>(a.acc, i * 2, nrec, r);
    a.nred[i * 2 + 0] = (float)(r[0] / a.V);
    a.nred[i * 2 + 1] = (float)(r[1] / a.V);
    sb += r[0];
    sg += r[1];
  }
  if (a.dgamma) {
    a.dgamma[c] = (float)sg;
    a.dbeta[c] = (float)sb;
  }
}

// Split-K epilogue: out[n][v][c] (+)= bias[c] + sum_s part[s][n][v][c] in split order, rounded once to fp16; a workgroup
// owns ALL voxels of one (sample, 32-channel block), so the InstanceNorm table of these small layers needs no cross-
// workgroup sum at all: per-thread (count, mean, M2) triples are merged pairwise in a fixed tree (Chan), in double.
struct SplitKFinish {
  const float* part;
  const float* bias;
  f16* out;
  float* nstat;
  const float* gamma;
  const float* beta;
  float eps;
  int nsplit, N, Cout, ldo, accumulate;
  long V;
};
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(SplitKFinish a) {
  __shared__ double sm[32][32], sq[32][32];
  __shared__ float scnt[32][32];
  const int tid = threadIdx.x;
  const int c4 = (tid & 7) * 4, vl = tid >> 3;   // 8 threads cover the block's 32 channels; 32 voxel lanes
  const int cb = blockIdx.x, n = blockIdx.y;
  const int c0 = cb * 32 + c4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
  float cnt = 0.f, mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
  const long slab = (long)a.N * a.V * a.Cout;
  for (long v = vl; v < a.V; v += 32) {
    const float* src = a.part + ((long)n * a.V + v) * a.Cout + c0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < a.nsplit; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + q * slab);
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    f16* dst = a.out + ((long)n * a.V + v) * a.ldo + c0;
    f16x4 o = {(f16)(s[0] + bv[0]), (f16)(s[1] + bv[1]), (f16)(s[2] + bv[2]), (f16)(s[3] + bv[3])};
    if (a.accumulate) {
      const f16x4 old = *reinterpret_cast<const f16x4*>(dst);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)((float)o[e] + (float)old[e]);
    }
    *reinterpret_cast<f16x4*>(dst) = o;
    if (a.nstat) {  // Welford on the stored fp16 values
      cnt += 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = (float)o[e], d = x - mean[e];
        mean[e] += d / cnt;
        m2[e] += d * (x - mean[e]);
      }
    }
  }
  if (!a.nstat) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sm[vl][c4 + e] = (double)mean[e];
    sq[vl][c4 + e] = (double)m2[e];
  }
  scnt[vl][c4] = cnt;
  __syncthreads();
  if (tid < 32) {
    const int c = tid;
    double M = 0.0, Q = 0.0, Nn = 0.0;
    for (int l = 0; l < 32; ++l) {   // fixed order
      const double nb = (double)scnt[l][c & ~3];
      if (nb == 0.0) continue;
      const double d = sm[l][c] - M, tot = Nn + nb;
      M += d * nb / tot;
      Q += sq[l][c] + d * d * Nn * nb / tot;
      Nn = tot;
    }
    const double var = Nn > 0.0 ? Q / Nn : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const int cg = cb * 32 + c;
    const float sc = rstd * a.gamma[cg];
    const f32x4 o = {(float)M, rstd, sc, a.beta[cg] - (float)M * sc};
    *reinterpret_cast<f32x4*>(a.nstat + ((long)n * a.Cout + cg) * 4) = o;
  }
}

// split-K decision: few workgroups, many slices, one tap group with unit output stride, and a workspace that holds the partials
static int splitk_plan(const ConvDev& p, long ws_floats, int wgs_base, int* kper) {
  const nnz_conv_desc& d = p.d;
  const int nkc = d.Cin / 16;
  if (g_tuning[5] || !p.part || d.ngroups != 1 || d.out_stride[0] != 1 || d.out_stride[1] != 1 || d.out_stride[2] != 1 ||
      nkc < 8 || wgs_base >= g_tuning[8] || p.stats || p.bx)
    return 1;
  int splits = (384 + wgs_base - 1) / wgs_base;
  if (splits > nkc / 2) splits = nkc / 2;          // at least two slices per workgroup
  const long per_split = (long)d.N * d.out_dims[0] * d.out_dims[1] * d.out_dims[2] * d.Cout;
  if ((long)splits * per_split > ws_floats) splits = (int)(ws_floats / per_split);
  if (splits < 2) return 1;
  *kper = (nkc + splits - 1) / splits;
  return (nkc + *kper - 1) / *kper;
}

template <int TD, int TH, int TW, int NB, int LPT_BOX, class G, bool DRE = false, bool DFLIP = false, int MINB = 2,
          bool PERSIST = false>
static int launch_cfg(const ConvDev& base, hipStream_t stream) {
  using C = ConvCfg<TD, TH, TW, NB>;
  ConvDev p = base;
  const G geo(p.d);
  const BoxGeom<TD, TH, TW, G> bg(geo);
  if (bg.NBOXLOAD > LPT_BOX * 256) return NNZ_EINVAL;  // register staging cannot hold this box
  int maxnt = 1;  // weights region: all taps of the largest group, one 16-channel slice
  for (int g = 0; g < p.d.ngroups; ++g) maxnt = p.d.groups[g].ntaps > maxnt ? p.d.groups[g].ntaps : maxnt;
  const int wbytes = NB * maxnt * 1024;
  int lds = bg.BOX_BYTES + wbytes > C::OUT_BYTES ? bg.BOX_BYTES + wbytes : C::OUT_BYTES;
  if (p.stats || p.acc) {  // statistics slab [256 / (4 NB)][2 * 32 NB] floats behind the output image
    const int need = C::OUT_BYTES + (256 / (NB * 4)) * (2 * NB * 32) * 4;
    lds = lds > need ? lds : need;
  }
  p.intab_off = (bg.BOX_BYTES + wbytes + 15) & ~15;
  if (p.in_tab) {
    const int need = p.intab_off + 2 * 16 * 8;
    lds = lds > need ? lds : need;
  }
  if (lds > 160 * 1024) return NNZ_EINVAL;
  p.tiles[0] = (p.d.m_dims[0] + TD - 1) / TD;
  p.tiles[1] = (p.d.m_dims[1] + TH - 1) / TH;
  p.tiles[2] = (p.d.m_dims[2] + TW - 1) / TW;
  p.gx = p.tiles[0] * p.tiles[1] * p.tiles[2];
  p.gy = p.d.Cout / (32 * NB);
  p.kper = 0;
  p.nsplit = splitk_plan(p, p.ws_floats, p.gx * p.gy * p.d.N * p.d.ngroups, &p.kper);
  if (p.nsplit <= 1) p.part = nullptr;
  p.gz = p.d.N * p.d.ngroups * (p.nsplit > 1 ? p.nsplit : 1);
  p.cout_fastest = g_tuning[3] && p.gy > 1;
  {
    const int ISH = geo.is(1), ISW = geo.is(2);
    for (int t = 0; t < p.d.ntaps_total && t < NNZ_MAX_TAPS; ++t) {
      const nnz_conv_tap& tp = p.d.taps[t];
      const int o0 = tp.off[0] - p.d.lo[0], o1 = tp.off[1] - p.d.lo[1], o2 = tp.off[2] - p.d.lo[2];
      const int col = ISW == 2 ? (o2 >> 1) + (o2 & 1) * ((bg.BW + 1) >> 1) : o2;
      const int flip = (ISH == 2 ? o1 >> 1 : o1) & 1;
      p.tapenc[t] = (((o0 * bg.BH + o1) * bg.PW + col) * 32) | (flip << 4);
    }
    for (int r = 0; r < 9 && r < p.d.ntaps_total; ++r) {   // depth-reuse loop: rows of the FIRST tap group
      const nnz_conv_tap& tp = p.d.taps[p.d.groups[0].tap_begin + r];
      const int o1 = tp.off[1] - p.d.lo[1], o2 = tp.off[2] - p.d.lo[2];
      p.rowenc[r] = ((o1 * bg.PW + o2) * 32) | ((o1 & 1) << 4);
    }
  }
  // PERSIST: as many consecutive W tiles per workgroup as still leave >= 1024 workgroups (two rounds of the 512 resident
  // slots), never across a W row; knob 9 caps it (1 = one tile per workgroup)
  p.tiles_per_wg = 1;
  if (PERSIST && p.nsplit <= 1) {
    for (int t = g_tuning[9]; t > 1; t >>= 1)
      if (p.tiles[2] % t == 0 && (long)(p.gx / t) * p.gy * p.gz >= 1024) {
        p.tiles_per_wg = t;
        break;
      }
  }
  auto kern = conv_box_kernel<TD, TH, TW, NB, LPT_BOX, G, DRE, DFLIP, MINB, PERSIST>;
  static DynLdsCache lds_cache;  // per instantiation, per device
  {
    hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_cache);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned nwg = (unsigned)(p.gx / p.tiles_per_wg) * p.gy * p.gz;
  p.sep_finish = g_tuning[10] && p.acc && p.nsplit <= 1;
  p.mfma_moments = g_tuning[11];
  {
    auto magic = [](int d) -> unsigned { return d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u; };
    const int gxw = p.gx / p.tiles_per_wg;
    const int dmax = gxw > p.gy ? gxw : p.gy;
    if ((unsigned long long)nwg * (unsigned long long)(dmax > p.gz ? dmax : p.gz) >= (1ull << 32)) return NNZ_EINVAL;
    p.mg_gy = magic(p.gy); p.mg_gxw = magic(gxw); p.mg_nsplit = magic(p.nsplit); p.mg_ngroups = magic(p.d.ngroups);
    p.mg_t2 = magic(p.tiles[2]); p.mg_t1 = magic(p.tiles[1]);
    p.brick = g_tuning[14] && !PERSIST && p.tiles[0] % 4 == 0 && p.tiles[1] % 4 == 0 && p.tiles[2] % 4 == 0;
    p.mg_b2 = magic(p.tiles[2] / 4); p.mg_b1 = magic(p.tiles[1] / 4);
    p.group_fastest = g_tuning[15] && p.d.ngroups > 1 && p.nsplit <= 1;
  }
  {
    const unsigned long long ib = 2ull * p.d.N * p.d.in_dims[0] * p.d.in_dims[1] * p.d.in_dims[2] * (unsigned long long)p.d.ldi;
    const unsigned long long wb = 2ull * (p.d.Cin / 16) * (p.d.Cout / 32) * (unsigned long long)p.d.ntaps_total * 512ull;
    if (ib >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return NNZ_EINVAL;   // 32-bit byte offsets in the staging loads
    const unsigned long long bb = p.bx ? 2ull * p.d.N * p.d.out_dims[0] * p.d.out_dims[1] * p.d.out_dims[2] * (unsigned long long)p.ldbx : 0ull;
    if (bb >= 0xFFFFFFF0ull) return NNZ_EINVAL;
    p.in_bytes = (unsigned)ib;
    p.w_bytes = (unsigned)wb;
    p.bx_bytes = (unsigned)bb;
  }
  NNZ_LAUNCH(kern, dim3(nwg), dim3(256), lds, stream, p);
  if (p.sep_finish && p.bx) {
    NormRedFinish f = {p.acc, p.nred, p.dgamma, p.dbeta, p.d.N, p.d.Cout,
                       (double)p.d.out_dims[0] * p.d.out_dims[1] * p.d.out_dims[2]};
    NNZ_LAUNCH(conv_normred_finish_kernel, dim3((p.d.Cout + 63) / 64), dim3(64), 0, stream, f);
  } else if (p.sep_finish) {
    StatsFinish f = {p.acc, p.nstat, p.gamma, p.beta, p.eps, p.d.N, p.d.Cout,
                     (double)p.d.m_dims[0] * p.d.m_dims[1] * p.d.m_dims[2], p.mfma_moments ? 3 : 2};
    NNZ_LAUNCH(conv_stats_finish_kernel, dim3((p.d.N * p.d.Cout + 63) / 64), dim3(64), 0, stream, f);
  }
  if (p.nsplit > 1) {
    SplitKFinish f = {};
    f.part = p.part; f.bias = p.bias; f.out = p.out; f.nstat = p.acc ? p.nstat : nullptr;
    f.gamma = p.gamma; f.beta = p.beta; f.eps = p.eps;
    f.nsplit = p.nsplit; f.N = p.d.N; f.Cout = p.d.Cout; f.ldo = p.d.ldo; f.accumulate = p.d.accumulate;
    f.V = (long)p.d.out_dims[0] * p.d.out_dims[1] * p.d.out_dims[2];
    NNZ_LAUNCH(conv_splitk_finish_kernel, dim3(p.d.Cout / 32, p.d.N), dim3(256), 0, stream, f);
  }
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

template <int TD, int TH, int TW, int IS, int EXT>
constexpr int iso_lpt() {
  return (((TD - 1) * IS + EXT + 1) * ((TH - 1) * IS + EXT + 1) * ((TW - 1) * IS + EXT + 1) * 2 + 255) / 256;
}
template <int TD, int TH, int TW, int NB, int IS, int EXT>
static int launch_iso(const ConvDev& p, hipStream_t stream) {
  return launch_cfg<TD, TH, TW, NB, iso_lpt<TD, TH, TW, IS, EXT>(), GeoIso<IS, EXT>>(p, stream);
}

// k3 s1 table in depth-major tap order whose three depth taps of every (kh, kw) row share their in-plane offset and sit
// on box planes (0, 1, 2) [forward] or (2, 1, 0) [data gradient]: what the depth-reuse loop assumes.  Returns -1 if not.
static int depth_reuse_flip(const nnz_conv_desc& d) {
  if (d.ngroups != 1 || d.groups[0].ntaps != 27 || d.groups[0].tap_begin != 0) return -1;
  for (int a = 0; a < 3; ++a)
    if (d.in_stride[a] != 1 || d.out_stride[a] != 1 || d.ext[a] != 2) return -1;
  const int first = d.taps[0].off[0] - d.lo[0];
  if (first != 0 && first != 2) return -1;
  const int flip = first == 2;
  for (int r = 0; r < 9; ++r)
    for (int q = 0; q < 3; ++q) {
      const nnz_conv_tap& t = d.taps[q * 9 + r];
      if (t.off[1] != d.taps[r].off[1] || t.off[2] != d.taps[r].off[2]) return -1;
      if (t.off[0] - d.lo[0] != (flip ? 2 - q : q)) return -1;
    }
  return flip;
}

// isotropic 3-D plans (the tuned path: every box dimension is a compile-time constant)
template <int IS, int EXT>
static int launch_tile(const ConvDev& p, hipStream_t stream) {
  const long mvox = (long)p.d.m_dims[0] * p.d.m_dims[1] * p.d.m_dims[2];
  const bool nb2 = (p.d.Cout % 64) == 0;
  if constexpr (IS == 1) {
    // big tile for the 32-channel full-resolution layers (more weight reuse per wave), otherwise 4x8x8
    if constexpr (EXT == 2) {
      const int flip = depth_reuse_flip(p.d);
      const long edge = g_tuning[2];
      if (flip >= 0 && mvox >= edge * edge * edge && ((!nb2 && g_tuning[0]) || (nb2 && g_tuning[1] && p.d.Cin >= g_tuning[4]))) {
        // (PERSIST = true - one workgroup walking up to four consecutive W tiles - is built and bit-exact, and slower: the tile
        //  loop makes the allocator spill 300-720 bytes per lane (next_free_vgpr 256 + scratch) whatever is done against
        //  hoisting (opaque thread index inside the tile body, late instead of early prefetch); 32 -> 32 @128^3 forward
        //  0.29 ms one tile per workgroup, 0.40-0.56 ms persistent.  tools/bench_conv_layers.py --tuning 9=T with the
        //  template flag flipped reproduces it.)
        if (flip) return launch_cfg<8, 8, 8, 1, iso_lpt<8, 8, 8, IS, EXT>(), GeoIso<IS, EXT>, true, true, 2, NNZ_DRE_PERSIST>(p, stream);
        return launch_cfg<8, 8, 8, 1, iso_lpt<8, 8, 8, IS, EXT>(), GeoIso<IS, EXT>, true, false, 2, NNZ_DRE_PERSIST>(p, stream);
      }
    }
    if (!nb2) {
      if (mvox >= 64 * 64 * 64) return launch_iso<8, 8, 8, 1, IS, EXT>(p, stream);
      return launch_iso<4, 8, 8, 1, IS, EXT>(p, stream);
    }
    // deep, spatially small levels (<= 16^3): the 4x8x8 tile leaves most of the 256 CUs idle, so cut the m-tile
    // to 2x4x8 (4x the workgroups, one 32x32 MFMA tile per wave)
    const long wgs488 = (long)p.d.N * p.d.ngroups * (p.d.Cout / 64) * ((p.d.m_dims[0] + 3) / 4) *
                        ((p.d.m_dims[1] + 7) / 8) * ((p.d.m_dims[2] + 7) / 8);
    if (wgs488 < 256) return launch_iso<2, 4, 8, 2, IS, EXT>(p, stream);
    // (an 8x8x8 x 64-cout tile was tried for the large levels to halve the weight staging per voxel: 210 VGPRs,
    //  spills and 1 workgroup per CU made it 7-13 % slower than 4x8x8 - measured, round 1)
    return launch_iso<4, 8, 8, 2, IS, EXT>(p, stream);
  } else {
    // stride 2, large grids (knob 7): 4x8x8 voxels x 64 couts per workgroup - four times the MFMA work per staged slice and
    // half the LDS fragment traffic per MFMA (2 x 2 register tile) of the 2x4x8 tile, at one workgroup per CU (142 KB of LDS)
    if (g_tuning[7] && mvox >= 32L * 32 * 32)
      return launch_cfg<4, 8, 8, 2, iso_lpt<4, 8, 8, IS, EXT>(), GeoIso<IS, EXT>, false, false, 1, NNZ_S2_PERSIST>(p, stream);
    return launch_cfg<2, 4, 8, 2, iso_lpt<2, 4, 8, IS, EXT>(), GeoIso<IS, EXT>, false, false, 2, NNZ_S2_PERSIST>(p, stream);  // N-split over waves: needs Cout % 64 == 0
  }
}

// per-axis geometry: 2-D plans (depth-1 volumes -> flat tiles) and anisotropic 3-D plans
static int launch_dyn(const ConvDev& p, hipStream_t stream) {
  const nnz_conv_desc& d = p.d;
  const bool nb2 = (d.Cout % 64) == 0;
  const bool strided = d.in_stride[0] == 2 || d.in_stride[1] == 2 || d.in_stride[2] == 2;
  if (d.m_dims[0] == 1 && d.in_dims[0] == 1 && d.out_dims[0] == 1) {
    // dilated 3x3 (REBNCONV, dilation 2 / 4 / 8: tap offsets up to +-8, box = tile + 2 * dilation): the 1x8x8 tile with
    // room for a 24 x 24 box in the staging registers (the maps are 32^2 / 16^2: one wave of workgroups either way)
    if (d.ext[1] > 2 || d.ext[2] > 2) {
      if (strided) return NNZ_EINVAL;
      return nb2 ? launch_cfg<1, 8, 8, 2, 5, GeoDyn>(p, stream) : launch_cfg<1, 16, 8, 1, 6, GeoDyn>(p, stream);
    }
    // flat tiles: 1x32x8 voxels (stride 1), 1x16x8 (stride 2: the box is (2*TH+1) x 17), 1x8x8 for small maps
    const long wgs = (long)d.N * d.ngroups * (d.Cout / (nb2 ? 64 : 32)) * ((d.m_dims[1] + 31) / 32) *
                     ((d.m_dims[2] + 7) / 8);
    if (nb2 && wgs < 256) return launch_cfg<1, 8, 8, 2, 3, GeoDyn>(p, stream);
    if (strided) {
      if (nb2) return launch_cfg<1, 16, 8, 2, 5, GeoDyn>(p, stream);
      return launch_cfg<1, 16, 8, 1, 5, GeoDyn>(p, stream);
    }
    if (nb2) return launch_cfg<1, 32, 8, 2, 3, GeoDyn>(p, stream);
    return launch_cfg<1, 32, 8, 1, 3, GeoDyn>(p, stream);
  }
  if (strided) {
    if (!nb2) return NNZ_EINVAL;
    return launch_cfg<2, 4, 8, 2, 6, GeoDyn>(p, stream);
  }
  if (!nb2) return launch_cfg<4, 8, 8, 1, 5, GeoDyn>(p, stream);
  const long wgs488 = (long)d.N * d.ngroups * (d.Cout / 64) * ((d.m_dims[0] + 3) / 4) * ((d.m_dims[1] + 7) / 8) *
                      ((d.m_dims[2] + 7) / 8);
  if (wgs488 < 256) return launch_cfg<2, 4, 8, 2, 6, GeoDyn>(p, stream);
  return launch_cfg<4, 8, 8, 2, 5, GeoDyn>(p, stream);
}

}  // namespace nnz

extern "C" int nnz_conv_tap_forward_stats(const void* in, void* out, const void* w_packed, const float* bias,
                                          const nnz_conv_desc* desc, float* stats, void* stream);
struct NormRedArgs {  // see ConvDev::bx
  const void* bx;
  const float* bstat;
  float* nred;
  float* dgamma;
  float* dbeta;
  float slope;
  int ldbx;
};
struct InNormArgs {  // see ConvDev::in_tab
  const float* tab;
  int c0;
  float slope;
};
static int conv_tap_forward_impl(const void* in, void* out, const void* w_packed, const float* bias,
                                 const nnz_conv_desc* desc, float* stats, void* acc, unsigned* counter,
                                 const float* gamma, const float* beta, float eps, float* nstat, void* stream,
                                 float* workspace = nullptr, long ws_floats = 0, const NormRedArgs* nr = nullptr,
                                 const InNormArgs* inn = nullptr);

// read-back (tests restore a knob they changed; conv_wgrad.hip's experiment build reads the timestamp buffer's address)
extern "C" int nnz_conv_tuning_get(int knob) { return knob >= 0 && knob < 16 ? nnz::g_tuning[knob] : 0; }

extern "C" int nnz_conv_tuning(int knob, int value) {
  if (knob < 0 || knob >= 16) return NNZ_EINVAL;
  nnz::g_tuning[knob] = value;
  return NNZ_OK;
}

extern "C" int nnz_conv_tap_forward(const void* in, void* out, const void* w_packed, const float* bias,
                                    const nnz_conv_desc* desc, void* stream) {
  return nnz_conv_tap_forward_stats(in, out, w_packed, bias, desc, nullptr, stream);
}

extern "C" int nnz_conv_tap_forward_stats(const void* in, void* out, const void* w_packed, const float* bias,
                                          const nnz_conv_desc* desc, float* stats, void* stream) {
  return conv_tap_forward_impl(in, out, w_packed, bias, desc, stats, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, stream);
}

// Forward convolution whose epilogue also produces the InstanceNorm table of its own output - deterministic (fixed-point
// accumulators, see common.hpp) and free of the sumsq/V - mean^2 cancellation in fp32: nstat[N][Cout][4] = {mean, rstd,
// rstd * gamma, beta - mean * rstd * gamma}, written by the launch's last workgroup.  `acc`: N * Cout * 2 records of
// nnz_fxacc_bytes() bytes, `counter`: one 32-bit word; both zero before the first launch and left zero by every launch.
extern "C" int nnz_conv_tap_forward_norm(const void* in, void* out, const void* w_packed, const float* bias,
                                         const nnz_conv_desc* desc, void* acc, void* counter, const float* gamma,
                                         const float* beta, float eps, float* nstat, void* stream) {
  if (!acc || !counter || !gamma || !beta || !nstat) return NNZ_EINVAL;
  return conv_tap_forward_impl(in, out, w_packed, bias, desc, nullptr, acc, (unsigned*)counter, gamma, beta, eps, nstat, stream);
}

// bytes of ONE logical accumulator record as the *_det / *_norm entry points count them (all its replicas)
// The same two entry points with a caller-provided fp32 workspace: layers with few output tiles and a long reduction (the
// <= 8^3 levels of the 3-D nets) then split the reduction over workgroups (split-K over 16-channel slices), partials in the
// workspace, folded in split order by a finishing kernel that also writes the InstanceNorm table (norm variant).  Results are
// deterministic; without a workspace (or for layers that do not qualify) the call is the plain one.
extern "C" int nnz_conv_tap_forward_ws(const void* in, void* out, const void* w_packed, const float* bias,
                                       const nnz_conv_desc* desc, float* workspace, long ws_floats, void* stream) {
  return conv_tap_forward_impl(in, out, w_packed, bias, desc, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, stream,
                               workspace, ws_floats);
}
extern "C" int nnz_conv_tap_forward_norm_ws(const void* in, void* out, const void* w_packed, const float* bias,
                                            const nnz_conv_desc* desc, void* acc, void* counter, const float* gamma,
                                            const float* beta, float eps, float* nstat, float* workspace, long ws_floats,
                                            void* stream) {
  if (!acc || !counter || !gamma || !beta || !nstat) return NNZ_EINVAL;
  return conv_tap_forward_impl(in, out, w_packed, bias, desc, nullptr, acc, (unsigned*)counter, gamma, beta, eps, nstat, stream,
                               workspace, ws_floats);
}

// Data-gradient launch (any dgrad table: stride-1, the phase groups of a stride-2 layer, accumulating or not) that also
// closes the reductions of the InstanceNorm(affine) + LeakyReLU backward of the layer BELOW: `out` is dL/d(activation) of
// that layer, `x_raw` its fp16 conv output [N][out voxels][ld_x], `nstat` its table [N][Cout][4] (Cout = desc->Cout).
// Writes nred[N][Cout][2] = {mean g', mean g' xhat} and, if given, dgamma / dbeta [Cout] - what
// nnz_instnorm_lrelu_bwd_tab's reducing launch would produce; the caller then runs only its apply launch
// (nnz_instnorm_lrelu_bwd_apply_tab).  Deterministic (fixed-point sums).  `acc`: N * Cout * 2 records of nnz_fxacc_bytes(),
// `counter`: one word; zero before, left zero.  The whole batch must fit one launch (< 2^31 elements per tensor).
extern "C" int nnz_conv_tap_dgrad_normred(const void* in, void* out, const void* w_packed, const nnz_conv_desc* desc,
                                          const void* x_raw, int ld_x, const float* nstat, float slope, void* acc,
                                          void* counter, float* nred, float* dgamma, float* dbeta, void* stream) {
  NormRedArgs nr = {x_raw, nstat, nred, dgamma, dbeta, slope, ld_x};
  return conv_tap_forward_impl(in, out, w_packed, nullptr, desc, nullptr, acc, (unsigned*)counter, nullptr, nullptr, 0.f,
                               nullptr, stream, nullptr, 0, &nr);
}

// Forward convolution whose INPUT is the raw (pre-norm) conv output of the producer block: InstanceNorm(affine) + LeakyReLU
// are applied while the input box is staged (ConvDev::in_tab), so the activated tensor never exists in HBM - the consumer
// side of the "conv + norm + act" fusion (reference op sequence Conv -> InstanceNorm -> LeakyReLU,
// /root/reference/nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:285-305).
//   in_tab   [N][Cin - in_c0][4]   the producer's table {mean, rstd, scale, shift} (nnz_conv_tap_forward_norm* output)
//   in_c0    channels [0, in_c0) of `in` are used as they are (cat buffer: transposed-conv half), multiple of 16
//   in_slope LeakyReLU slope of the producer block
// acc / counter / gamma / beta / nstat as in nnz_conv_tap_forward_norm_ws (all NULL: no statistics of the output, e.g. the
// one-tap groups of a transposed convolution); workspace as in nnz_conv_tap_forward_ws.
extern "C" int nnz_conv_tap_forward_innorm(const void* in, void* out, const void* w_packed, const float* bias,
                                           const nnz_conv_desc* desc, const float* in_tab, int in_c0, float in_slope,
                                           void* acc, void* counter, const float* gamma, const float* beta, float eps,
                                           float* nstat, float* workspace, long ws_floats, void* stream) {
  if (!in_tab) return NNZ_EINVAL;
  const bool any = acc || counter || gamma || beta || nstat;
  if (any && (!acc || !counter || !gamma || !beta || !nstat)) return NNZ_EINVAL;
  InNormArgs inn = {in_tab, in_c0, in_slope};
  return conv_tap_forward_impl(in, out, w_packed, bias, desc, nullptr, acc, (unsigned*)counter, gamma, beta, eps, nstat, stream,
                               workspace, ws_floats, nullptr, &inn);
}

extern "C" int nnz_fxacc_bytes(void) { return (int)sizeof(nnz::FxAcc) * nnz::FX_REP; }

static int conv_tap_forward_impl(const void* in, void* out, const void* w_packed, const float* bias,
                                 const nnz_conv_desc* desc, float* stats, void* acc, unsigned* counter,
                                 const float* gamma, const float* beta, float eps, float* nstat, void* stream,
                                 float* workspace, long ws_floats, const NormRedArgs* nr, const InNormArgs* inn) {
  using namespace nnz;
  if (!in || !out || !w_packed || !desc) return NNZ_EINVAL;
  if (inn && inn->tab && (inn->c0 < 0 || inn->c0 % 16 || inn->c0 >= desc->Cin)) return NNZ_EINVAL;
  const nnz_conv_desc& d = *desc;
  // fused statistics: plain forward convolutions only (one group, output written once, unit output stride)
  if ((stats || acc) && !nr &&
      (d.ngroups != 1 || d.accumulate || d.out_stride[0] != 1 || d.out_stride[1] != 1 || d.out_stride[2] != 1))
    return NNZ_EINVAL;
  if (nr && (!acc || !counter || !nr->bx || !nr->bstat || !nr->nred || nr->ldbx % 8 || nr->ldbx < d.Cout ||
             (nr->dgamma == nullptr) != (nr->dbeta == nullptr)))
    return NNZ_EINVAL;
  if (d.Cin % 32 || d.Cout % 32 || d.ngroups < 1 || d.ngroups > NNZ_MAX_GROUPS || d.ntaps_total > NNZ_MAX_TAPS ||
      d.ldi % 8 || d.ldo % 8)
    return NNZ_EINVAL;
  // tap extent per axis: 0..2 for the k1 / k3 layers; up to 16 (dilation 8) on the H / W axes of 2-D layers
  const bool flat = d.m_dims[0] == 1 && d.in_dims[0] == 1 && d.out_dims[0] == 1;
  for (int a = 0; a < 3; ++a)
    if ((d.in_stride[a] != 1 && d.in_stride[a] != 2) || (d.out_stride[a] != 1 && d.out_stride[a] != 2) ||
        d.ext[a] < 0 || d.ext[a] > ((flat && a > 0) ? 16 : 2))
      return NNZ_EINVAL;
  for (int g = 0; g < d.ngroups; ++g)
    if (d.groups[g].ntaps > 27 || d.groups[g].ntaps < 1) return NNZ_EINVAL;
  // The kernels index voxels with 32-bit element offsets: batches whose tensors pass 2^31 elements (288 GB of HBM make
  // batch 16+ of 128^3 x 64 channels realistic) are launched in sample chunks that stay below it.
  const long in_per_n = (long)d.in_dims[0] * d.in_dims[1] * d.in_dims[2] * d.ldi;
  const long out_per_n = (long)d.out_dims[0] * d.out_dims[1] * d.out_dims[2] * d.ldo;
  const long per_n = in_per_n > out_per_n ? in_per_n : out_per_n;
  if (per_n >= (1L << 31)) return NNZ_EINVAL;
  int chunk = (int)(((1L << 31) - 1) / per_n);
  if (chunk > d.N) chunk = d.N;
  if (nr && chunk < d.N) return NNZ_EINVAL;  // the fused reductions close over the whole batch in one launch
  hipStream_t s = (hipStream_t)stream;
  const bool iso = d.in_stride[0] == d.in_stride[1] && d.in_stride[1] == d.in_stride[2] && d.ext[0] == d.ext[1] &&
                   d.ext[1] == d.ext[2] && d.m_dims[0] > 1;
  for (int n0 = 0; n0 < d.N; n0 += chunk) {
    ConvDev p;
    p.in = (const f16*)in + (size_t)n0 * in_per_n;
    p.out = (f16*)out + (size_t)n0 * out_per_n;
    p.w = (const f16*)w_packed;
    p.bias = bias;
    p.stats = stats ? stats + (size_t)n0 * d.Cout * 2 : nullptr;
    p.acc = (FxAcc*)acc;  // every sample chunk is a launch of its own: the bank is indexed from 0 again
    p.nstat = nstat ? nstat + (size_t)n0 * d.Cout * 4 : nullptr;
    p.counter = counter;
    p.gamma = gamma;
    p.beta = beta;
    p.eps = eps;
    p.bx = nr ? (const f16*)nr->bx : nullptr;
    p.bstat = nr ? nr->bstat : nullptr;
    p.nred = nr ? nr->nred : nullptr;
    p.dgamma = nr ? nr->dgamma : nullptr;
    p.dbeta = nr ? nr->dbeta : nullptr;
    p.slope = nr ? nr->slope : 0.f;
    p.ldbx = nr ? nr->ldbx : 0;
    p.dbg = g_tuning[6];
#if NNZ_CONV_TIMESTAMPS
    p.ts = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)g_tuning[13] << 32) | (unsigned)g_tuning[12]);
#endif
    p.in_tab = inn && inn->tab ? inn->tab + (size_t)n0 * (d.Cin - inn->c0) * 4 : nullptr;
    p.in_c0 = inn ? inn->c0 : 0;
    p.in_slope = inn ? inn->slope : 0.f;
    p.intab_off = 0;
    p.part = workspace;
    p.ws_floats = workspace ? ws_floats : 0;
    p.nsplit = 1;
    p.kper = 0;
    p.d = d;
    p.d.N = d.N - n0 < chunk ? d.N - n0 : chunk;
    int rc;
    if (!iso) {
      rc = launch_dyn(p, s);
    } else {
      const int is = d.in_stride[0], ext = d.ext[0];
      if (is == 1) {
        rc = ext == 0 ? launch_tile<1, 0>(p, s) : ext == 1 ? launch_tile<1, 1>(p, s) : launch_tile<1, 2>(p, s);
      } else {
        if (d.Cout % 64) return NNZ_EINVAL;  // the IS=2 tiles split N over waves
        rc = ext <= 1 ? launch_tile<2, 1>(p, s) : launch_tile<2, 2>(p, s);
      }
    }
    if (rc != NNZ_OK) return rc;
  }
  return NNZ_OK;
}
