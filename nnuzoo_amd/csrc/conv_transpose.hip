// ConvTranspose with kernel = stride (the up-sampling of every PlainConvUNet decoder stage: nn.ConvTranspose3d / 2d built by
// dynamic_network_architectures' UNetDecoder from the planner's strides, default_experiment_planner.py:285-305), forward
// and data gradient, for the FULL-RESOLUTION stages.
//
// Why a kernel of its own: with kernel = stride every output voxel has exactly one input voxel and one weight tap
// (its parity p): out[s m + p][co] = b[co] + sum_ci in[m][ci] W[ci][co][p].  That is 8 small GEMMs (K = Cin) over a tensor
// that is written once - 8.6 GFLOP against 335 MB for the 64 -> 32 channel stage at 128^3: HBM-bound.  Run as eight one-tap
// groups of the tap-table conv_box kernel it paid that kernel's per-tile prologue / epilogue for 32 MFMAs of work and moved
// data at ~1 TB/s (0.32 ms forward + 0.20 ms data gradient per step for that one stage, profiles/r02_bench_n1_kernel_stats.csv).
// Here: all P x Cin x Cout weights live in LDS as ready-made MFMA A fragments (converted from the fp32 master by the
// workgroup itself: no packed copy), workgroups are persistent over tiles of 32 input voxels per wave, the input voxel's
// channel row is the B operand straight from global memory (16 bytes per lane), each parity's [32 voxels][Cout] result is
// transposed through LDS and leaves as whole 16-byte pieces of the output rows.
//   data gradient: dIn[m][ci] = sum_p sum_co dOut[s m + p][co] W[ci][co][p] - the same structure with K = P x Cout.
// Deeper stages (Cin >= 256: the weights no longer fit LDS, and the tensors are tiny) stay on conv_box.
#include "common.hpp"

namespace nnz {

struct ConvTArgs {
  const f16* in;      // fwd: x [N][Vin][ldi] (Cin ch);   dgrad: dOut [N][Vout][ldi] (Cout ch)
  const float* W;     // [Cin][Cout][P] fp32 (torch ConvTranspose layout)
  const float* bias;  // [Cout] or null (fwd)
  f16* out;           // fwd: [N][Vout][ldo] (Cout ch);  dgrad: dIn [N][Vin][ldo] (Cin ch)
  int N, Di, Hi, Wi, Cin, Cout, sd, sh, sw, ldi, ldo;
  long ntiles;        // ceil(N * Vin / 32)
  // forward only - consumer-side InstanceNorm + LeakyReLU (nnz_convT_forward_innorm): `in` is the RAW conv output of the
  // block below; its table in_tab[N][Cin][4] = {mean, rstd, scale, shift} is staged as [N][Cin] {scale, shift} pairs in LDS
  // and every B fragment is normalised in registers (fp32 FMA, LeakyReLU, one rounding to fp16 - the bits the apply pass of
  // norm_act.hip would have written)
  const float* in_tab;
  float in_slope;
};

constexpr int CT_PAD = 8;   // f16 of padding per staged output row

// D tile of mfma32: lane (col = lane & 31, hh = lane >> 5) holds rows 8 q + 4 hh + j in acc[4 q + j]
__device__ __forceinline__ void ct_spill(const f32x16& acc, const float* bias_rows /* or null */, f16* row, int hh) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f16x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = acc[4 * q + j];
      if (bias_rows) x += bias_rows[8 * q + 4 * hh + j];
      v[j] = (f16)x;
    }
    *reinterpret_cast<f16x4*>(row + 8 * q + 4 * hh) = v;
  }
}

template <bool DGRAD>
__global__ __launch_bounds__(256) void convT_kernel(ConvTArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ct_lds[];
  const int P = a.sd * a.sh * a.sw;
  const int M = DGRAD ? a.Cin : a.Cout;            // rows of the result (channels written)
  const int MB = M / 32;
  const int KS = (DGRAD ? a.Cout : a.Cin) / 16;    // 16-channel K slices per parity
  f16x8* sW = reinterpret_cast<f16x8*>(ct_lds);    // fwd [P][MB][KS][64], dgrad [MB][P][KS][64]
  const int nfrag = P * MB * KS * 64;
  const int pitch = M + CT_PAD;
  f16* sOut = reinterpret_cast<f16*>(ct_lds + (size_t)nfrag * 16);           // [4 waves][32][pitch]
  long* sOv = reinterpret_cast<long*>(sOut + 4 * 32 * pitch);               // [4 waves][32] output voxel of each column
  float* sBias = reinterpret_cast<float*>(sOv + 4 * 32);                     // [Cout] (fwd)
  float* sTab = sBias + a.Cout;                                              // [N][Cin][2] (fwd, consumer-side norm)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hh = lane >> 5;

  // ---- weights -> MFMA A fragments (row = lane & 31 of the block, k = 8 (lane >> 5) + j).  One (ci, co) pair at a time: its
  // P taps are contiguous in the fp32 parameter (one or two 16-byte loads), each lands in its parity's fragment
  {
    f16* sWh = reinterpret_cast<f16*>(sW);
    const int pairs = a.Cin * a.Cout;
    for (int e = tid; e < pairs; e += 256) {
      const int ci = e / a.Cout, co = e % a.Cout;
      const int row = DGRAD ? ci : co, k = DGRAD ? co : ci;
      const int mb = row >> 5, l = (row & 31) + 32 * ((k >> 3) & 1), ks = k >> 4, j = k & 7;
      const float* w = a.W + (long)e * P;
      float wv[8];
      if (P == 8) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w), w1 = *reinterpret_cast<const f32x4*>(w + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { wv[q] = w0[q]; wv[4 + q] = w1[q]; }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = q < P ? w[q] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if (p < P) {
          const long f = DGRAD ? ((long)(mb * P + p) * KS + ks) * 64 + l : ((long)(p * MB + mb) * KS + ks) * 64 + l;
          sWh[f * 8 + j] = (f16)wv[p];
        }
      }
    }
  }
  if (!DGRAD)
    for (int c = tid; c < a.Cout; c += 256) sBias[c] = a.bias ? a.bias[c] : 0.f;
  if (!DGRAD && a.in_tab)
    for (int i = tid; i < a.N * a.Cin; i += 256)
      *reinterpret_cast<f32x2*>(sTab + 2 * i) = *reinterpret_cast<const f32x2*>(a.in_tab + (size_t)i * 4 + 2);
  __syncthreads();

  const long Vin = (long)a.Di * a.Hi * a.Wi;
  const int Do = a.Di * a.sd, Ho = a.Hi * a.sh, Wo = a.Wi * a.sw;
  const long total = (long)a.N * Vin;
  f16* myOut = sOut + wave * 32 * pitch;
  long* myOv = sOv + wave * 32;
  const int ppv = M / 8;                              // 16-byte pieces per result row

  for (long tile = (long)blockIdx.x * 4 + wave; tile < a.ntiles; tile += (long)gridDim.x * 4) {
    const long v = tile * 32 + col;                   // input voxel of this lane's column
    const bool valid = v < total;
    const long vv = valid ? v : total - 1;
    const int n = (int)(vv / Vin);
    long r = vv % Vin;
    const int x = (int)(r % a.Wi);
    r /= a.Wi;
    const int y = (int)(r % a.Hi), z = (int)(r / a.Hi);
    if (!DGRAD) {
      // B fragments: the input voxel's channels, 8 per lane-half and K slice
      f16x8 bf[20];
#pragma unroll
      for (int ks = 0; ks < 20; ++ks)
        if (ks < KS) bf[ks] = *reinterpret_cast<const f16x8*>(a.in + vv * a.ldi + ks * 16 + 8 * hh);
      if (a.in_tab) {
        const float* tn = sTab + ((long)n * a.Cin + 8 * hh) * 2;
#pragma unroll
        for (int ks = 0; ks < 20; ++ks)
          if (ks < KS) {
            bf[ks] = __builtin_bit_cast(f16x8, norm_lrelu8_tab(__builtin_bit_cast(u32x4, bf[ks]), tn + ks * 32,
                                                               slope_pair(a.in_slope)));
          }
      }
      for (int p = 0; p < P; ++p) {
        const int px = p % a.sw, py = (p / a.sw) % a.sh, pz = p / (a.sw * a.sh);
        const long ov = (((long)n * Do + a.sd * z + pz) * Ho + a.sh * y + py) * Wo + a.sw * x + px;
        if (hh == 0) myOv[col] = valid ? ov : -1;
        for (int mb = 0; mb < MB; ++mb) {
          f32x16 acc = {};
          const f16x8* wf = sW + ((long)(p * MB + mb) * KS) * 64 + lane;
#pragma unroll
          for (int ks = 0; ks < 20; ++ks)
            if (ks < KS) acc = mfma32(wf[ks * 64], bf[ks], acc);
          ct_spill(acc, sBias + mb * 32, myOut + col * pitch + mb * 32, hh);
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < 32 * ppv; i += 64) {   // whole 16-byte pieces of the output rows
          const int c = i / ppv, c8 = i % ppv;
          const long o = myOv[c];
          if (o >= 0)
            *reinterpret_cast<f16x8*>(a.out + o * a.ldo + c8 * 8) = *reinterpret_cast<const f16x8*>(myOut + c * pitch + c8 * 8);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      // B fragments: the 2x2x2 children's gradient channels, per parity and K slice (P * KS <= 32)
      f16x8 bf[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (i < P * KS) {
          const int p = i / KS, ks = i % KS;
          const int px = p % a.sw, py = (p / a.sw) % a.sh, pz = p / (a.sw * a.sh);
          const long ov = (((long)n * Do + a.sd * z + pz) * Ho + a.sh * y + py) * Wo + a.sw * x + px;
          bf[i] = *reinterpret_cast<const f16x8*>(a.in + ov * a.ldi + ks * 16 + 8 * hh);
        }
      }
      if (hh == 0) myOv[col] = valid ? vv : -1;
      for (int mb = 0; mb < MB; ++mb) {
        f32x16 acc = {};
        const f16x8* wf = sW + ((long)mb * P * KS) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < P * KS) acc = mfma32(wf[i * 64], bf[i], acc);
        ct_spill(acc, nullptr, myOut + col * pitch + mb * 32, hh);
      }
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < 32 * ppv; i += 64) {
        const int c = i / ppv, c8 = i % ppv;
        const long o = myOv[c];
        if (o >= 0)
          *reinterpret_cast<f16x8*>(a.out + o * a.ldo + c8 * 8) = *reinterpret_cast<const f16x8*>(myOut + c * pitch + c8 * 8);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

static size_t ct_lds_bytes(int Cin, int Cout, int P, bool dgrad, int tabN = 0) {
  const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  return (size_t)P * (M / 32) * (K / 16) * 64 * 16 + (size_t)4 * 32 * (M + CT_PAD) * 2 + 4 * 32 * 8 + (size_t)Cout * 4 +
         (size_t)tabN * Cin * 8;
}

static bool ct_supported(int Cin, int Cout, int sd, int sh, int sw, bool dgrad) {
  const int P = sd * sh * sw;
  if (sd < 1 || sd > 2 || sh < 1 || sh > 2 || sw < 1 || sw > 2 || P < 2) return false;
  const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  if (M % 32 || K % 16 || M > 256) return false;
  if (!dgrad && K / 16 > 20) return false;
  if (dgrad && P * (K / 16) > 32) return false;
  return ct_lds_bytes(Cin, Cout, P, dgrad) <= 150 * 1024;
}

template <bool DGRAD>
static int ct_launch(ConvTArgs a, hipStream_t s) {
  if (!a.in || !a.W || !a.out || a.N < 1 || a.Di < 1 || a.Hi < 1 || a.Wi < 1) return NNZ_EINVAL;
  if (!ct_supported(a.Cin, a.Cout, a.sd, a.sh, a.sw, DGRAD)) return NNZ_EINVAL;
  if (a.ldi % 8 || a.ldo % 8) return NNZ_EINVAL;
  const long total = (long)a.N * a.Di * a.Hi * a.Wi;
  a.ntiles = (total + 31) / 32;
  if (a.in_tab && (DGRAD || a.Cout % 4)) return NNZ_EINVAL;
  const size_t lds = ct_lds_bytes(a.Cin, a.Cout, a.sd * a.sh * a.sw, DGRAD, a.in_tab ? a.N : 0);
  if (lds > 160 * 1024) return NNZ_EINVAL;
  static DynLdsCache cache;
  hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(convT_kernel<DGRAD>), (int)lds, cache);
  if (e != hipSuccess) return (int)e;
  // persistent workgroups: as many as fit the chip at this LDS size (160 KB per CU), at most one per 4 tiles
  int per_cu = (int)((160 * 1024) / (lds + 1024));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;     // few long-lived workgroups: every workgroup converts the whole weight set once
  long blocks = 256L * per_cu;
  const long need = (a.ntiles + 3) / 4;
  if (blocks > need) blocks = need;
  NNZ_LAUNCH(convT_kernel<DGRAD>, dim3((unsigned)blocks), dim3(256), lds, s, a);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

}  // namespace nnz

extern "C" int nnz_convT_supported(int Cin, int Cout, int sd, int sh, int sw, int dgrad) {
  return nnz::ct_supported(Cin, Cout, sd, sh, sw, dgrad != 0) ? 1 : 0;
}

// out[n][(s m + p)][:Cout] (row stride ldo) = bias + in[n][m][:Cin] (row stride ldi) W[:, :, p]
extern "C" int nnz_convT_forward(const void* in, const float* W, const float* bias, void* out, int N, int Di, int Hi, int Wi,
                                 int Cin, int Cout, int sd, int sh, int sw, int ldi, int ldo, void* stream) {
  nnz::ConvTArgs a = {};
  a.in = (const f16*)in; a.W = W; a.bias = bias; a.out = (f16*)out;
  a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.sd = sd; a.sh = sh; a.sw = sw; a.ldi = ldi; a.ldo = ldo;
  return nnz::ct_launch<false>(a, (hipStream_t)stream);
}

// ... with `in` the RAW conv output of the block below, normalised on the fly (ConvTArgs::in_tab); reference ops:
// InstanceNorm3d + LeakyReLU feeding nn.ConvTranspose3d in the decoder (default_experiment_planner.py:285-305)
extern "C" int nnz_convT_forward_innorm(const void* in, const float* in_tab, float in_slope, const float* W, const float* bias,
                                        void* out, int N, int Di, int Hi, int Wi, int Cin, int Cout, int sd, int sh, int sw,
                                        int ldi, int ldo, void* stream) {
  if (!in_tab) return NNZ_EINVAL;
  nnz::ConvTArgs a = {};
  a.in = (const f16*)in; a.W = W; a.bias = bias; a.out = (f16*)out; a.in_tab = in_tab; a.in_slope = in_slope;
  a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.sd = sd; a.sh = sh; a.sw = sw; a.ldi = ldi; a.ldo = ldo;
  return nnz::ct_launch<false>(a, (hipStream_t)stream);
}

// din[n][m][:Cin] (row stride ldo) = sum_p dout[n][(s m + p)][:Cout] (row stride ldi) W[:, :, p]^T
extern "C" int nnz_convT_dgrad(const void* dout, const float* W, void* din, int N, int Di, int Hi, int Wi, int Cin, int Cout,
                               int sd, int sh, int sw, int ldi, int ldo, void* stream) {
  nnz::ConvTArgs a = {};
  a.in = (const f16*)dout; a.W = W; a.out = (f16*)din;
  a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.sd = sd; a.sh = sh; a.sw = sw; a.ldi = ldi; a.ldo = ldo;
  return nnz::ct_launch<true>(a, (hipStream_t)stream);
}
