// Device-side training augmentations: the interpolating and the intensity transforms of the
// reference's training chain, nnUNetTrainer.get_training_transforms (/root/reference/nnunetv2/training/nnUNetTrainer/
// nnUNetTrainer.py:825-973), applied to a batch that is already resident in HBM (dataloading/device_loader.py cut it there):
//   SpatialTransform (rotation p 0.2, isotropic scaling p 0.2 in (0.7, 1.4), no elastic deformation; :845-852)
//   GaussianNoiseTransform (variance in (0, 0.1), :857-863)         MultiplicativeBrightnessTransform ((0.75, 1.25), :872-878)
//   ContrastTransform ((0.75, 1.25), preserve_range, :879-886)      GammaTransform ((0.7, 1.5), retain_stats, inverted / plain, :897-914)
// The transform classes live in batchgeneratorsv2 (pyproject.toml:51 ">=0.2"), which is neither in /root/reference nor in this
// image: the arithmetic below restates the published algorithm of each transform from its name and the call site's parameters -
// PARITY UNPINNED - and is tested against plain torch fp32 formulations of the same arithmetic
// (tests/test_device_augment_gpu.py).  Also here: GaussianBlurTransform (:864-871, separable, edge-replicating) and
// SimulateLowResolutionTransform (:887-896, nearest down / linear up in one gather pass; the package interpolates the way up with a cubic).
// Everything here is HBM-bound streaming over a 2 x C x 128^3 batch (tens of MB): one launch per transform, statistics as a
// fixed-order two-stage reduction (no float atomics).  Built without fast-math (build.py STRICT_FP): powf / sqrtf / division as written.
#include "common.hpp"

namespace nnz {

constexpr int AUG_MAXB = 16;     // samples per launch: their matrices travel as kernel arguments

struct AffineArgs {
  const void* src;             // [B][C][D][H][W]
  void* dst;                   // same shape
  float m[AUG_MAXB][12];       // per sample: source index = M (out index - centre) + centre + shift, rows (z, y, x), 3 x 4 row-major
  int B, C, D, H, W;
  float pad_f;
  int pad_i;
};

__device__ __forceinline__ void aug_src_coord(const float* m, float cz, float cy, float cx, int z, int y, int x, float& sz,
                                              float& sy, float& sx) {
  const float dz = (float)z - cz, dy = (float)y - cy, dx = (float)x - cx;
  sz = m[0] * dz + m[1] * dy + m[2] * dx + m[3] + cz;
  sy = m[4] * dz + m[5] * dy + m[6] * dx + m[7] + cy;
  sx = m[8] * dz + m[9] * dy + m[10] * dx + m[11] + cx;
}

// data: trilinear (bilinear when D = 1), samples outside the volume contribute `pad`
__global__ __launch_bounds__(256) void aug_affine_f32_kernel(AffineArgs a) {
  const int b = blockIdx.y;
  const long vol = (long)a.D * a.H * a.W;
  const float* m = a.m[b];
  const float cz = 0.5f * (float)(a.D - 1), cy = 0.5f * (float)(a.H - 1), cx = 0.5f * (float)(a.W - 1);
  const float* src = static_cast<const float*>(a.src) + (long)b * a.C * vol;
  float* dst = static_cast<float*>(a.dst) + (long)b * a.C * vol;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long)gridDim.x * 256) {
    const int x = (int)(i % a.W);
    const long r = i / a.W;
    const int y = (int)(r % a.H), z = (int)(r / a.H);
    float sz, sy, sx;
    aug_src_coord(m, cz, cy, cx, z, y, x, sz, sy, sx);
    const float fz = floorf(sz), fy = floorf(sy), fx = floorf(sx);
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const float tz = a.D > 1 ? sz - fz : 0.f, ty = sy - fy, tx = sx - fx;
    for (int c = 0; c < a.C; ++c) {
      const float* s = src + (long)c * vol;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int oz = q >> 2, oy = (q >> 1) & 1, ox = q & 1;
        const float w = (oz ? tz : 1.f - tz) * (oy ? ty : 1.f - ty) * (ox ? tx : 1.f - tx);
        if (w == 0.f) continue;
        const int zz = z0 + oz, yy = y0 + oy, xx = x0 + ox;
        const bool in = zz >= 0 && zz < a.D && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        acc += w * (in ? s[((long)zz * a.H + yy) * a.W + xx] : a.pad_f);
      }
      dst[(long)c * vol + i] = acc;
    }
  }
}

// segmentation: nearest neighbour (round half up), `pad` outside
__global__ __launch_bounds__(256) void aug_affine_i16_kernel(AffineArgs a) {
  const int b = blockIdx.y;
  const long vol = (long)a.D * a.H * a.W;
  const float* m = a.m[b];
  const float cz = 0.5f * (float)(a.D - 1), cy = 0.5f * (float)(a.H - 1), cx = 0.5f * (float)(a.W - 1);
  const short* src = static_cast<const short*>(a.src) + (long)b * a.C * vol;
  short* dst = static_cast<short*>(a.dst) + (long)b * a.C * vol;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long)gridDim.x * 256) {
    const int x = (int)(i % a.W);
    const long r = i / a.W;
    const int y = (int)(r % a.H), z = (int)(r / a.H);
    float sz, sy, sx;
    aug_src_coord(m, cz, cy, cx, z, y, x, sz, sy, sx);
    const int zz = a.D > 1 ? (int)floorf(sz + 0.5f) : 0, yy = (int)floorf(sy + 0.5f), xx = (int)floorf(sx + 0.5f);
    const bool in = zz >= 0 && zz < a.D && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    for (int c = 0; c < a.C; ++c)
      dst[(long)c * vol + i] = in ? src[(long)c * vol + ((long)zz * a.H + yy) * a.W + xx] : (short)a.pad_i;
  }
}

// ---- per-(sample, channel) statistics: {sum, sum of squares, min, max} ---------------------------------------------------
constexpr int AUG_SB = 64;       // partial blocks per (sample, channel)

__global__ __launch_bounds__(256) void aug_stats_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
  const long bc = blockIdx.y;
  const float* p = x + bc * n;
  double s = 0.0, ss = 0.0;
  float mn = INFINITY, mx = -INFINITY;
  const long per = (n + AUG_SB - 1) / AUG_SB;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = p[i];
    s += (double)v;
    ss += (double)v * (double)v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  __shared__ double sh_s[256], sh_ss[256];
  __shared__ float sh_mn[256], sh_mx[256];
  sh_s[threadIdx.x] = s; sh_ss[threadIdx.x] = ss; sh_mn[threadIdx.x] = mn; sh_mx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + o];
      sh_ss[threadIdx.x] += sh_ss[threadIdx.x + o];
      sh_mn[threadIdx.x] = fminf(sh_mn[threadIdx.x], sh_mn[threadIdx.x + o]);
      sh_mx[threadIdx.x] = fmaxf(sh_mx[threadIdx.x], sh_mx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* q = reinterpret_cast<double*>(part) + (bc * AUG_SB + blockIdx.x) * 4;
    q[0] = sh_s[0]; q[1] = sh_ss[0]; q[2] = (double)sh_mn[0]; q[3] = (double)sh_mx[0];
  }
}
// stats[bc] = {mean, standard deviation (population), min, max}
__global__ __launch_bounds__(64) void aug_stats_final_kernel(const float* __restrict__ part, long n, int nbc, float* __restrict__ stats) {
  const int bc = blockIdx.x * 64 + threadIdx.x;
  if (bc >= nbc) return;
  const double* q = reinterpret_cast<const double*>(part) + (long)bc * AUG_SB * 4;
  double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < AUG_SB; ++k) {
    s += q[4 * k];
    ss += q[4 * k + 1];
    mn = q[4 * k + 2] < mn ? q[4 * k + 2] : mn;
    mx = q[4 * k + 3] > mx ? q[4 * k + 3] : mx;
  }
  const double mean = s / (double)n;
  double var = ss / (double)n - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[4 * bc + 0] = (float)mean;
  stats[4 * bc + 1] = (float)sqrt(var);
  stats[4 * bc + 2] = (float)mn;
  stats[4 * bc + 3] = (float)mx;
}

// ---- element-wise transforms, in place, one record per (sample, channel) ---------------------------------------------------
// op 0  noise     v += p0 * N(0, 1)                       (p0 = sigma; draws: counter-based, keyed by (seed, bc, element))
// op 1  linear    v = p0 * v + p1                         (brightness: p1 = 0; inversion: p0 = -1)
// op 2  contrast  v = clamp((v - mean) p0 + mean, min, max)            statistics of `sa` (the tensor as it is)
// op 3  gamma     v = ((v - min) / max(max - min, 1e-7)) ^ p0 * (max - min) + min         statistics of `sa`
// op 4  restore   v = (v - mean_a) / max(std_a, 1e-7) * std_b + mean_b     sa = statistics now, sb = statistics to restore
// rec[bc] = {active (0 / 1), p0, p1, unused}
__device__ __forceinline__ unsigned aug_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float aug_gauss(unsigned seed, unsigned bc, unsigned long i) {
  const unsigned k = aug_hash(seed ^ (bc * 0x9e3779b9U));
  const unsigned a = aug_hash(k ^ (unsigned)i ^ aug_hash((unsigned)(i >> 32) + 0x632be5abU));
  const unsigned b = aug_hash(a + 0x85ebca6bU);
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);       // (0, 1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);                // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ __launch_bounds__(256) void aug_intensity_kernel(float* __restrict__ x, long n, int op, const float* __restrict__ rec,
                                                            const float* __restrict__ sa, const float* __restrict__ sb,
                                                            unsigned seed) {
  const int bc = blockIdx.y;
  const float active = rec[4 * bc], p0 = rec[4 * bc + 1], p1 = rec[4 * bc + 2];
  if (active == 0.f) return;
  float* p = x + (long)bc * n;
  float mean = 0.f, sd = 0.f, mn = 0.f, mx = 0.f, mean_b = 0.f, sd_b = 0.f;
  if (sa) { mean = sa[4 * bc]; sd = sa[4 * bc + 1]; mn = sa[4 * bc + 2]; mx = sa[4 * bc + 3]; }
  if (sb) { mean_b = sb[4 * bc]; sd_b = sb[4 * bc + 1]; }
  const float rng = mx - mn;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = p[i];
    switch (op) {
      case 0: v += p0 * aug_gauss(seed, (unsigned)bc, (unsigned long)i); break;
      case 1: v = p0 * v + p1; break;
      case 2: v = fminf(fmaxf((v - mean) * p0 + mean, mn), mx); break;
      case 3: v = powf(fmaxf((v - mn) / fmaxf(rng, 1e-7f), 0.f), p0) * rng + mn; break;
      default: v = (v - mean) / fmaxf(sd, 1e-7f) * sd_b + mean_b; break;
    }
    p[i] = v;
  }
}

// ---- GaussianBlurTransform (:864-871): separable Gaussian, one axis per launch, sigma per (sample, channel) and axis ----------
// rec[bc] = {active, sigma_z, sigma_y, sigma_x}; taps k = -R .. R with R = ceil(3 sigma) (<= AUG_BLUR_R), weights exp(-k^2 / 2 sigma^2)
// normalised over the taps; the border repeats the edge voxel.  axis: 0 = z, 1 = y, 2 = x.  src != dst.
constexpr int AUG_BLUR_R = 4;
__global__ __launch_bounds__(256) void aug_blur_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W,
                                                            int axis, const float* __restrict__ rec) {
  const int bc = blockIdx.y;
  const long n = (long)D * H * W;
  const float* s = src + (long)bc * n;
  float* d = dst + (long)bc * n;
  const float sigma = rec[4 * bc + 1 + axis];
  const bool on = rec[4 * bc] != 0.f && sigma > 0.f && (axis == 0 ? D : (axis == 1 ? H : W)) > 1;
  float w[2 * AUG_BLUR_R + 1];
  int R = 0;
  if (on) {
    R = (int)ceilf(3.f * sigma);
    R = R < 1 ? 1 : (R > AUG_BLUR_R ? AUG_BLUR_R : R);
    float tot = 0.f;
    for (int k = -R; k <= R; ++k) {
      w[k + AUG_BLUR_R] = expf(-(float)(k * k) / (2.f * sigma * sigma));
      tot += w[k + AUG_BLUR_R];
    }
    for (int k = -R; k <= R; ++k) w[k + AUG_BLUR_R] /= tot;
  }
  const int len = axis == 0 ? D : (axis == 1 ? H : W);
  const long stride = axis == 0 ? (long)H * W : (axis == 1 ? W : 1);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (!on) { d[i] = s[i]; continue; }
    const int pos = (int)((i / stride) % len);
    float acc = 0.f;
    for (int k = -R; k <= R; ++k) {
      int q = pos + k;
      q = q < 0 ? 0 : (q >= len ? len - 1 : q);
      acc += w[k + AUG_BLUR_R] * s[i + (long)(q - pos) * stride];
    }
    d[i] = acc;
  }
}

// ---- SimulateLowResolutionTransform (:887-896): nearest down-sampling to round(size * scale) per axis, linear up-sampling back -
// rec[bc] = {active, scale, -, -} (axes synchronised; 2-D / dummy-2-D: the z axis keeps its size: `keep_z`).  One gather pass: the
// low-resolution voxel j of an axis of n voxels sits at (j + 0.5) n / m - 0.5 and holds the original voxel floor((j + 0.5) n / m).
__device__ __forceinline__ void aug_lr_axis(int i, int n, int m, int& j0, int& j1, float& t) {
  float p = ((float)i + 0.5f) * (float)m / (float)n - 0.5f;       // position of output voxel i on the low-resolution grid
  p = p < 0.f ? 0.f : (p > (float)(m - 1) ? (float)(m - 1) : p);
  j0 = (int)floorf(p);
  j1 = j0 + 1 < m ? j0 + 1 : m - 1;
  t = p - (float)j0;
}
__device__ __forceinline__ int aug_lr_src(int j, int n, int m) {
  const int s = (int)floorf(((float)j + 0.5f) * (float)n / (float)m);
  return s < n - 1 ? s : n - 1;
}
__global__ __launch_bounds__(256) void aug_lowres_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W,
                                                         int keep_z, const float* __restrict__ rec) {
  const int bc = blockIdx.y;
  const long n = (long)D * H * W;
  const float* s = src + (long)bc * n;
  float* d = dst + (long)bc * n;
  const float scale = rec[4 * bc + 1];
  const bool on = rec[4 * bc] != 0.f && scale > 0.f && scale < 1.f;
  int mz = D, my = H, mx = W;
  if (on) {
    if (!keep_z && D > 1) { mz = (int)rintf((float)D * scale); mz = mz < 1 ? 1 : mz; }
    my = (int)rintf((float)H * scale); my = my < 1 ? 1 : my;
    mx = (int)rintf((float)W * scale); mx = mx < 1 ? 1 : mx;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (!on) { d[i] = s[i]; continue; }
    const int x = (int)(i % W);
    const long r = i / W;
    const int y = (int)(r % H), z = (int)(r / H);
    int z0, z1, y0, y1, x0, x1;
    float tz, ty, tx;
    aug_lr_axis(z, D, mz, z0, z1, tz);
    aug_lr_axis(y, H, my, y0, y1, ty);
    aug_lr_axis(x, W, mx, x0, x1, tx);
    const int sz[2] = {aug_lr_src(z0, D, mz), aug_lr_src(z1, D, mz)};
    const int sy[2] = {aug_lr_src(y0, H, my), aug_lr_src(y1, H, my)};
    const int sx[2] = {aug_lr_src(x0, W, mx), aug_lr_src(x1, W, mx)};
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int oz = q >> 2, oy = (q >> 1) & 1, ox = q & 1;
      const float wgt = (oz ? tz : 1.f - tz) * (oy ? ty : 1.f - ty) * (ox ? tx : 1.f - tx);
      if (wgt == 0.f) continue;
      acc += wgt * s[((long)sz[oz] * H + sy[oy]) * W + sx[ox]];
    }
    d[i] = acc;
  }
}

// seg: every `from` becomes `to` (RemoveLabelTansform(-1, 0), nnUNetTrainer.py:929-931)
__global__ __launch_bounds__(256) void aug_relabel_i16_kernel(short* __restrict__ x, long n, int from, int to) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    if (x[i] == (short)from) x[i] = (short)to;
}

template <typename T>
static int affine_launch(const void* src, void* dst, const float* mats, int B, int C, int D, int H, int W, float pad_f, int pad_i,
                         hipStream_t s) {
  if (!src || !dst || src == dst || !mats || B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return NNZ_EINVAL;
  const long vol = (long)D * H * W;
  for (int b0 = 0; b0 < B; b0 += AUG_MAXB) {
    AffineArgs a = {};
    const int nb = B - b0 < AUG_MAXB ? B - b0 : AUG_MAXB;
    a.src = static_cast<const char*>(src) + (long)b0 * C * vol * sizeof(T);
    a.dst = static_cast<char*>(dst) + (long)b0 * C * vol * sizeof(T);
    for (int j = 0; j < nb; ++j)
      for (int k = 0; k < 12; ++k) a.m[j][k] = mats[(b0 + j) * 12 + k];
    a.B = nb; a.C = C; a.D = D; a.H = H; a.W = W; a.pad_f = pad_f; a.pad_i = pad_i;
    long blocks = (vol + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (sizeof(T) == 4) NNZ_LAUNCH(aug_affine_f32_kernel, dim3((unsigned)blocks, nb), dim3(256), 0, s, a);
    else NNZ_LAUNCH(aug_affine_i16_kernel, dim3((unsigned)blocks, nb), dim3(256), 0, s, a);
    NNZ_LAUNCH_CHECK();
  }
  return NNZ_OK;
}

// ---- label-side transforms of the training chain whose arithmetic the REFERENCE defines itself (round 6) -------------------
// /root/reference/nnunetv2/training/data_augmentation/custom_transforms/: region_based_training.py:7-39
// (ConvertSegmentationToRegionsTransform), cascade_transforms.py:10-39 (MoveSegAsOneHotToData), masking.py:6-24 (MaskTransform);
// the trainer chain calls their batchgeneratorsv2 namesakes at nnUNetTrainer.py:921-927, 932-939, 961-969.  Element-wise on the
// resident batch; label lists travel as kernel arguments (<= AUG_MAXL entries).
constexpr int AUG_MAXL = 64;
struct LabelTable {
  int n;                   // regions / labels
  int begin[AUG_MAXL + 1]; // regions: labels [begin[r], begin[r + 1]) belong to region r
  int label[AUG_MAXL];
};
// out[b][r][i] = 1 where seg[b][seg_channel][i] is one of region r's labels, else 0   (seg [B][Cs][n], out [B][R][n], int16)
__global__ __launch_bounds__(256) void aug_regions_kernel(const short* __restrict__ seg, short* __restrict__ out, long n, int Cs,
                                                          int seg_channel, LabelTable t) {
  const int b = blockIdx.y;
  const short* sp = seg + ((long)b * Cs + seg_channel) * n;
  short* op = out + (long)b * t.n * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int v = sp[i];
    for (int r = 0; r < t.n; ++r) {
      short hit = 0;
      for (int k = t.begin[r]; k < t.begin[r + 1]; ++k) hit |= (short)(t.label[k] == v);
      op[(long)r * n + i] = hit;
    }
  }
}
// data[b][c0 + k][i] = (seg[b][seg_channel][i] == label[k]) as float, k < K   (data [B][Cd][n] float, seg [B][Cs][n] int16)
__global__ __launch_bounds__(256) void aug_onehot_kernel(const short* __restrict__ seg, float* __restrict__ data, long n, int Cs,
                                                         int seg_channel, int Cd, int c0, LabelTable t) {
  const int b = blockIdx.y;
  const short* sp = seg + ((long)b * Cs + seg_channel) * n;
  float* dp = data + ((long)b * Cd + c0) * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int v = sp[i];
    for (int k = 0; k < t.n; ++k) dp[(long)k * n + i] = t.label[k] == v ? 1.f : 0.f;
  }
}
// data[b][c][i] = value where seg[b][mask_channel][i] < 0, for the channels whose bit is set in `channels`
__global__ __launch_bounds__(256) void aug_mask_kernel(float* __restrict__ data, const short* __restrict__ seg, long n, int Cd, int Cs,
                                                       int mask_channel, unsigned long long channels, float value) {
  const int b = blockIdx.y;
  const short* sp = seg + ((long)b * Cs + mask_channel) * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (sp[i] < 0)
      for (int c = 0; c < Cd; ++c)
        if ((channels >> c) & 1ull) data[((long)b * Cd + c) * n + i] = value;
  }
}

}  // namespace nnz

// dst[b][c][o] = interpolation of src[b][c] at M_b (o - centre) + centre + shift_b (index space, centre = (size - 1) / 2 per axis);
// mats: B x 12 HOST floats (3 x 4 row-major, rows z, y, x; last column = shift).  2-D: D = 1 (the z row / column are ignored).
// f32: trilinear with `pad_value` outside;  i16: nearest with `pad_value` outside (the reference pads segmentations with -1).
extern "C" int nnz_aug_affine_f32(const float* src, float* dst, const float* mats, int B, int C, int D, int H, int W,
                                  float pad_value, void* stream) {
  return nnz::affine_launch<float>(src, dst, mats, B, C, D, H, W, pad_value, 0, (hipStream_t)stream);
}
extern "C" int nnz_aug_affine_i16(const short* src, short* dst, const float* mats, int B, int C, int D, int H, int W,
                                  int pad_value, void* stream) {
  return nnz::affine_launch<short>(src, dst, mats, B, C, D, H, W, 0.f, pad_value, (hipStream_t)stream);
}
// floats of the partials buffer of nnz_aug_stats_f32 for nbc (sample, channel) pairs
extern "C" long nnz_aug_stats_workspace_floats(int nbc) { return nbc < 1 ? 0 : (long)nbc * nnz::AUG_SB * 8; }
// stats[bc] = {mean, population standard deviation, min, max} of x[bc][0 .. n), fixed summation order (bit-identical run to run)
extern "C" int nnz_aug_stats_f32(const float* x, long n, int nbc, float* workspace, float* stats, void* stream) {
  using namespace nnz;
  if (!x || !workspace || !stats || n < 1 || nbc < 1) return NNZ_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  NNZ_LAUNCH(aug_stats_partial_kernel, dim3(AUG_SB, (unsigned)nbc), dim3(256), 0, s, x, n, workspace);
  NNZ_LAUNCH(aug_stats_final_kernel, dim3((unsigned)((nbc + 63) / 64)), dim3(64), 0, s, (const float*)workspace, n, nbc, stats);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
// in place over x[nbc][n]; rec: DEVICE [nbc][4] = {active, p0, p1, -}; stats_a / stats_b: device [nbc][4] from nnz_aug_stats_f32
// (may be NULL for ops that do not read them); op: 0 noise, 1 linear, 2 contrast, 3 gamma, 4 restore statistics (see above)
extern "C" int nnz_aug_intensity_f32(float* x, long n, int nbc, int op, const float* rec, const float* stats_a,
                                     const float* stats_b, int seed, void* stream) {
  using namespace nnz;
  if (!x || !rec || n < 1 || nbc < 1 || op < 0 || op > 4) return NNZ_EINVAL;
  if ((op == 2 || op == 3 || op == 4) && !stats_a) return NNZ_EINVAL;
  if (op == 4 && !stats_b) return NNZ_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(aug_intensity_kernel, dim3((unsigned)blocks, (unsigned)nbc), dim3(256), 0, (hipStream_t)stream, x, n, op, rec,
             stats_a, stats_b, (unsigned)seed);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
// one axis of the separable Gaussian blur: dst[bc] = blur_axis(src[bc]) for active rows, a copy otherwise; rec = device [nbc][4]
// {active, sigma_z, sigma_y, sigma_x}; src != dst; x is [nbc][D][H][W]
extern "C" int nnz_aug_blur_axis_f32(const float* src, float* dst, int nbc, int D, int H, int W, int axis, const float* rec,
                                     void* stream) {
  using namespace nnz;
  if (!src || !dst || src == dst || !rec || nbc < 1 || D < 1 || H < 1 || W < 1 || axis < 0 || axis > 2) return NNZ_EINVAL;
  long blocks = ((long)D * H * W + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(aug_blur_axis_kernel, dim3((unsigned)blocks, (unsigned)nbc), dim3(256), 0, (hipStream_t)stream, src, dst, D, H, W, axis,
             rec);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
// low-resolution simulation: rec = device [nbc][4] {active, scale in (0, 1), -, -}; keep_z != 0: the z axis is not resampled
extern "C" int nnz_aug_lowres_f32(const float* src, float* dst, int nbc, int D, int H, int W, int keep_z, const float* rec,
                                  void* stream) {
  using namespace nnz;
  if (!src || !dst || src == dst || !rec || nbc < 1 || D < 1 || H < 1 || W < 1) return NNZ_EINVAL;
  long blocks = ((long)D * H * W + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(aug_lowres_kernel, dim3((unsigned)blocks, (unsigned)nbc), dim3(256), 0, (hipStream_t)stream, src, dst, D, H, W, keep_z,
             rec);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
extern "C" int nnz_aug_relabel_i16(short* x, long n, int from, int to, void* stream) {
  using namespace nnz;
  if (!x || n < 1) return NNZ_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  NNZ_LAUNCH(aug_relabel_i16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, from, to);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}

// ConvertSegmentationToRegionsTransform: out [B][R][n] int16 (written); region r = labels[begin[r] .. begin[r + 1]); host arrays
extern "C" int nnz_aug_seg_to_regions_i16(const short* seg, short* out, int B, int Cs, int seg_channel, long n, const int* begin,
                                          const int* labels, int R, void* stream) {
  using namespace nnz;
  if (!seg || !out || !begin || !labels || B < 1 || Cs < 1 || seg_channel < 0 || seg_channel >= Cs || n < 1 || R < 1 || R > AUG_MAXL ||
      begin[R] > AUG_MAXL || begin[0] != 0)
    return NNZ_EINVAL;
  LabelTable t = {};
  t.n = R;
  for (int r = 0; r <= R; ++r) t.begin[r] = begin[r];
  for (int k = 0; k < begin[R]; ++k) t.label[k] = labels[k];
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(aug_regions_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, seg, out, n, Cs, seg_channel, t);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
// MoveSegAsOneHotToData: channels c0 .. c0 + K - 1 of data [B][Cd][n] are written with the one-hot encoding of
// seg[:, seg_channel] over `labels` (K host ints); the caller allocates the widened data tensor and copies the image channels
extern "C" int nnz_aug_seg_onehot_to_data_f32(const short* seg, float* data, int B, int Cs, int seg_channel, int Cd, int c0, long n,
                                              const int* labels, int K, void* stream) {
  using namespace nnz;
  if (!seg || !data || !labels || B < 1 || Cs < 1 || seg_channel < 0 || seg_channel >= Cs || K < 1 || K > AUG_MAXL || c0 < 0 ||
      c0 + K > Cd || n < 1)
    return NNZ_EINVAL;
  LabelTable t = {};
  t.n = K;
  for (int k = 0; k < K; ++k) t.label[k] = labels[k];
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(aug_onehot_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, seg, data, n, Cs, seg_channel, Cd,
             c0, t);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
// MaskTransform: data[:, c][seg[:, mask_channel] < 0] = value for the channels in the bit set `channels` (bit c = channel c, Cd <= 63); in place
extern "C" int nnz_aug_mask_outside_f32(float* data, const short* seg, int B, int Cd, int Cs, int mask_channel, long n,
                                        long channels, float value, void* stream) {
  using namespace nnz;
  if (!data || !seg || B < 1 || Cd < 1 || Cd > 63 || Cs < 1 || mask_channel < 0 || mask_channel >= Cs || n < 1) return NNZ_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  NNZ_LAUNCH(aug_mask_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, data, seg, n, Cd, Cs, mask_channel,
             (unsigned long long)channels, value);
  NNZ_LAUNCH_CHECK();
  return NNZ_OK;
}
