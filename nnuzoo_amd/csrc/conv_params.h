// Plain-C parameter blocks shared by the host launchers and the device kernels of the conv path.
// (No torch types, no C++ types: these structs are part of the C-ABI, see include/nnuzoo_hip.h.)
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNZ_MAX_GROUPS 8
#define NNZ_MAX_TAPS 32

// One tap of a (possibly strided / transposed) convolution:
//   input voxel  = m * in_stride + off      (per axis, m = position on the launch's m-grid; strides are per axis
//                                            so that 2-D (D = 1) and anisotropic 3-D plans use the same tables)
//   widx         = index of the [Cin x Cout] weight slice this tap multiplies with
typedef struct nnz_conv_tap {
  int32_t off[3];
  int32_t widx;
} nnz_conv_tap;

// A tap group writes one phase of the output:   output voxel = m * out_stride + ooff.
// Ordinary convolutions have one group holding all taps; a k2s2 transposed convolution has 8 groups of
// one tap; the data-gradient of a k3s2 convolution has 8 groups (one per output parity) of 1..8 taps.
typedef struct nnz_conv_group {
  int32_t ooff[3];
  int32_t tap_begin;
  int32_t ntaps;
} nnz_conv_group;

// Generic "tap table" convolution on channels-last (N, D, H, W, C) fp16 tensors:
//   out[n, m*OS + ooff_g, co] (+)= bias[co] + sum_{t in g} sum_ci in[n, m*IS + off_t, ci] * W[widx_t][ci][co]
// Packed weight layout (see nnz_pack_conv_weight): Wp[Cin/16][Cout/32][T][32][16] fp16.
typedef struct nnz_conv_desc {
  int32_t N;
  int32_t in_dims[3];   // Di, Hi, Wi
  int32_t out_dims[3];  // Do, Ho, Wo
  int32_t m_dims[3];    // Dm, Hm, Wm  (launch grid in "m" space)
  int32_t Cin, Cout;    // both multiples of 32
  int32_t ldi, ldo;     // channel strides (elements per voxel) of in / out, >= Cin / Cout
  int32_t in_stride[3];   // IS per axis: 1 or 2
  int32_t out_stride[3];  // OS per axis: 1 or 2
  int32_t ext[3];         // max_t off - min_t off over all groups, per axis (0, 1 or 2)
  int32_t lo[3];        // min_t off per axis (the box origin relative to m*IS)
  int32_t ntaps_total;  // T
  int32_t ngroups;
  int32_t accumulate;   // 1: out += result (read-modify-write), 0: out = result
  nnz_conv_group groups[NNZ_MAX_GROUPS];
  nnz_conv_tap taps[NNZ_MAX_TAPS];
} nnz_conv_desc;

#ifdef __cplusplus
}
#endif
