// train_ops.h -- the HBM-bound kernels of one TrackNet training step (train.py:84-96):
//   BatchNorm2d in training mode (model.py:9): batch statistics, running-stat update, normalise+ReLU, backward
//   WBCELoss forward + closed-form backward (utils/metric.py:3-20)
//   head (1x1 conv + sigmoid) backward, 2x2 max-pool backward (+ skip-gradient add), nearest-upsample backward,
//   sample mixup (train.py:32-40).
// All reductions are two-stage and deterministic: fixed-shape partials (fp64) written by the first kernel,
// summed in a fixed order by a finalize kernel -- no atomics, no zero-initialised accumulators.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

typedef float t_f32x4 __attribute__((ext_vector_type(4)));
typedef float t_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kRedSplit = 64;   // partial sums per channel / per sample

// Sum `v` over the 256 threads of a workgroup; result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* red /* [4] LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- BatchNorm (training) forward ------------------------------------------------------------------------------
// partial[c][s] = (sum, sumsq) of z[:, c, :] over the s-th slice of the N*HW elements.  grid = (kRedSplit, C)
inline __global__ void __launch_bounds__(256) bn_stats_partial_kernel(const float* __restrict__ z, double* __restrict__ partial,
                                                               int N, int C, int HW) {
  __shared__ double red[4];
  const int c = blockIdx.y, s = blockIdx.x;
  const int hw4 = HW >> 2;
  const long total4 = (long)N * hw4;
  double s1 = 0.0, s2 = 0.0;
  for (long t = (long)s * 256 + threadIdx.x; t < total4; t += (long)kRedSplit * 256) {
    const int n = (int)(t / hw4);
    const int p = (int)(t - (long)n * hw4) << 2;
    const t_f32x4 v = *reinterpret_cast<const t_f32x4*>(z + ((size_t)n * C + c) * HW + p);
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1 += (double)v[k]; s2 += (double)v[k] * (double)v[k]; }
  }
  const double a = block_sum_256(s1, red);
  const double b = block_sum_256(s2, red);
  if (threadIdx.x == 0) { partial[((size_t)c * kRedSplit + s) * 2] = a; partial[((size_t)c * kRedSplit + s) * 2 + 1] = b; }
}

// partial[c][s] = sum over tiles s, s + kRedSplit, ... of the (sum, sum of squares) pairs a convolution epilogue left per channel
// and pixel tile (tile_stats [C][n_tiles][2], conv3x3_wino3_mfma.h): the input of bn_stats_finalize_kernel without a pass over z.
// grid = (kRedSplit, C), 64 threads: a wave strides over its slice, then a fixed-order butterfly.
inline __global__ void __launch_bounds__(64) bn_tile_stats_reduce_kernel(const double* __restrict__ tile_stats, double* __restrict__ partial,
                                                                  long n_tiles) {
  const int c = blockIdx.y, s = blockIdx.x;
  const double* src = tile_stats + (size_t)c * n_tiles * 2;
  double s1 = 0.0, s2 = 0.0;
  for (long t = (long)s * 64 + threadIdx.x; t < n_tiles; t += (long)kRedSplit * 64) { s1 += src[2 * t]; s2 += src[2 * t + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
  if (threadIdx.x == 0) { partial[((size_t)c * kRedSplit + s) * 2] = s1; partial[((size_t)c * kRedSplit + s) * 2 + 1] = s2; }
}

// The ReLU(BN(z)) scale as both directions evaluate it: y = fmaf(z - mean, bn_scale(gamma, invstd), beta)
__device__ __forceinline__ float bn_scale(float gamma, float invstd) { return (float)((double)gamma * (double)invstd); }

// mean / biased var -> (scale, shift) for the normalise pass, saved (mean, invstd) for backward, running-stat update
// with the UNBIASED variance.  One thread per channel.
__device__ __forceinline__ void bn_stats_finalize_channel(double s1, double s2, int c, const float* __restrict__ gamma, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float eps, float momentum, long count,
                                                          float* __restrict__ scale, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const double n = (double)count;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)invstd;
  scale[c] = bn_scale(gamma[c], save_invstd[c]);      // from the SAVED fp32 invstd: the backward pass recomputes exactly this value
  const double unbiased = count > 1 ? var * (n / (n - 1.0)) : var;
  running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
  running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
}
inline __global__ void bn_stats_finalize_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, float eps, float momentum, long count,
                                         float* __restrict__ scale, float* __restrict__ save_mean,
                                         float* __restrict__ save_invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < kRedSplit; ++s) { s1 += partial[((size_t)c * kRedSplit + s) * 2]; s2 += partial[((size_t)c * kRedSplit + s) * 2 + 1]; }
  (void)beta;
  bn_stats_finalize_channel(s1, s2, c, gamma, running_mean, running_var, eps, momentum, count, scale, save_mean, save_invstd);
}

// Round 6: bn_tile_stats_reduce_kernel + a finalize kernel as ONE launch (a training step made 27 + 27 of these 5-us launches on its critical
// chain).  grid = C, 1024 threads: wave w forms the partial sums of the slices s = w, w + 16, w + 32, w + 48 exactly as workgroup (s, c) of
// bn_tile_stats_reduce_kernel does (same strides, same butterfly), thread 0 then adds the kRedSplit partials in order like the finalize
// kernels: the same additions in the same order -- bit-identical to the two launches.  FIN: the per-channel epilogue.
template <class FIN>
__device__ __forceinline__ void bn_tile_stats_fold(const double* __restrict__ tile_stats, long n_tiles, FIN&& fin) {
  __shared__ double part_s[kRedSplit * 2];
  static_assert(kRedSplit == 64, "sixteen waves x four slices");
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* src = tile_stats + (size_t)c * n_tiles * 2;
  for (int s = wv; s < kRedSplit; s += 16) {
    double s1 = 0.0, s2 = 0.0;
    for (long t = (long)s * 64 + lane; t < n_tiles; t += (long)kRedSplit * 64) { s1 += src[2 * t]; s2 += src[2 * t + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
    if (lane == 0) { part_s[2 * s] = s1; part_s[2 * s + 1] = s2; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < kRedSplit; ++s) { s1 += part_s[2 * s]; s2 += part_s[2 * s + 1]; }
    fin(s1, s2, c);
  }
}
inline __global__ void __launch_bounds__(1024) bn_tile_stats_finalize_kernel(const double* __restrict__ tile_stats, long n_tiles, const float* __restrict__ gamma,
                                                                      float* __restrict__ running_mean, float* __restrict__ running_var, float eps,
                                                                      float momentum, long count, float* __restrict__ scale,
                                                                      float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  bn_tile_stats_fold(tile_stats, n_tiles, [&](double s1, double s2, int c) {
    bn_stats_finalize_channel(s1, s2, c, gamma, running_mean, running_var, eps, momentum, count, scale, save_mean, save_invstd);
  });
}

// a = max((z - mean[c])*scale[c] + beta[c], 0)  -- subtract first, like the reference: no cancellation when |mean| >> std
inline __global__ void __launch_bounds__(256) bn_apply_relu_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ a, long NC, int C, int HW) {
  const int hw4 = HW >> 2;
  const long total4 = NC * hw4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (long)gridDim.x * blockDim.x) {
    const long nc = t / hw4;
    const int c = (int)(nc % C);
    const float mu = mean[c], sc = scale[c], sh = shift[c];
    t_f32x4 v = *reinterpret_cast<const t_f32x4*>(z + t * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float y = fmaf(v[k] - mu, sc, sh); v[k] = y > 0.0f ? y : 0.0f; }
    *reinterpret_cast<t_f32x4*>(a + t * 4) = v;
  }
}

// The same pass for a block whose output is pooled next (model.py:47-48, 50-51, 53-54: a down block's last layer): it also writes
// MaxPool2d(2, 2) of the a it forms -- pointwise.h's maxpool2x2_kernel without its read of a.  Thread = two windows (2 rows x 4 columns).
// a is never NaN here (a NaN y fails y > 0), so the window maximum needs no order.  Needs W % 4 == 0, H % 2 == 0.
inline __global__ void __launch_bounds__(256) bn_apply_relu_pool_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float* __restrict__ a, float* __restrict__ pooled, long NC, int C, int H, int W) {
  const int Ho = H >> 1, W4 = W >> 2, Wo = W >> 1;
  const long total = NC * Ho * W4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int q = (int)(t % W4);
    const long u = t / W4;
    const int oh = (int)(u % Ho);
    const long nc = u / Ho;
    const int c = (int)(nc % C);
    const float mu = mean[c], sc = scale[c], sh = shift[c];
    const size_t i0 = ((size_t)nc * H + 2 * oh) * W + 4 * q;
    t_f32x4 v0 = *reinterpret_cast<const t_f32x4*>(z + i0), v1 = *reinterpret_cast<const t_f32x4*>(z + i0 + W);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float y0 = fmaf(v0[k] - mu, sc, sh), y1 = fmaf(v1[k] - mu, sc, sh);
      v0[k] = y0 > 0.0f ? y0 : 0.0f;
      v1[k] = y1 > 0.0f ? y1 : 0.0f;
    }
    *reinterpret_cast<t_f32x4*>(a + i0) = v0;
    *reinterpret_cast<t_f32x4*>(a + i0 + W) = v1;
    t_f32x2 o;
    o[0] = fmaxf(fmaxf(v0[0], v0[1]), fmaxf(v1[0], v1[1]));
    o[1] = fmaxf(fmaxf(v0[2], v0[3]), fmaxf(v1[2], v1[3]));
    *reinterpret_cast<t_f32x2*>(pooled + ((size_t)nc * Ho + oh) * Wo + 2 * q) = o;
  }
}

// ---- BatchNorm + ReLU backward ---------------------------------------------------------------------------------
// g = dA * (a > 0);  partial[c][s] = (sum g, sum g*xhat), xhat = (z - mean) * invstd.      grid = (kRedSplit, C)
// FROM_Z: the mask is recomputed from z with the forward's own expression (bit-identical to a > 0) instead of reading a:
// two tensors per pass instead of three.
template <bool FROM_Z>
__global__ void __launch_bounds__(256) bn_relu_bwd_partial_kernel(const float* __restrict__ dA, const float* __restrict__ a,
                                                                  const float* __restrict__ z, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, double* __restrict__ partial,
                                                                  int N, int C, int HW) {
  __shared__ double red[4];
  const int c = blockIdx.y, s = blockIdx.x;
  const int hw4 = HW >> 2;
  const long total4 = (long)N * hw4;
  const float mu = mean[c], is = invstd[c];
  const float sc = FROM_Z ? bn_scale(gamma[c], is) : 0.0f, sh = FROM_Z ? beta[c] : 0.0f;
  double s1 = 0.0, s2 = 0.0;
  for (long t = (long)s * 256 + threadIdx.x; t < total4; t += (long)kRedSplit * 256) {
    const int n = (int)(t / hw4);
    const int p = (int)(t - (long)n * hw4) << 2;
    const size_t off = ((size_t)n * C + c) * HW + p;
    const t_f32x4 g = *reinterpret_cast<const t_f32x4*>(dA + off);
    const t_f32x4 zv = *reinterpret_cast<const t_f32x4*>(z + off);
    t_f32x4 av;
    if (FROM_Z) {
#pragma unroll
      for (int k = 0; k < 4; ++k) av[k] = fmaf(zv[k] - mu, sc, sh);
    } else {
      av = *reinterpret_cast<const t_f32x4*>(a + off);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = av[k] > 0.0f ? g[k] : 0.0f;
      s1 += (double)gk;
      s2 += (double)gk * (double)((zv[k] - mu) * is);
    }
  }
  const double r1 = block_sum_256(s1, red);
  const double r2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) { partial[((size_t)c * kRedSplit + s) * 2] = r1; partial[((size_t)c * kRedSplit + s) * 2 + 1] = r2; }
}

// c4[c] = (mean, invstd, gamma * invstd exactly as the forward / backward passes round it, beta): what the data-gradient epilogue
// that takes BatchNorm backward's sums (WinoArgs::bn_c4) reads per channel.
inline __global__ void bn_bwd_consts_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ c4, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  c4[4 * c] = mean[c];
  c4[4 * c + 1] = invstd[c];
  c4[4 * c + 2] = bn_scale(gamma[c], invstd[c]);
  c4[4 * c + 3] = beta[c];
}

// dbeta = sum g, dgamma = sum g*xhat; coefficients of the apply pass:  dZ = k0*g - k1 - k2*xhat
__device__ __forceinline__ void bn_relu_bwd_finalize_channel(double s1, double s2, int c, int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ invstd, long count, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef) {
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  const double k0 = (double)gamma[c] * (double)invstd[c];
  coef[c] = (float)k0;
  coef[C + c] = (float)(k0 * s1 / (double)count);
  coef[2 * C + c] = (float)(k0 * s2 / (double)count);
}
inline __global__ void bn_relu_bwd_finalize_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                            const float* __restrict__ invstd, long count, float* __restrict__ dgamma,
                                            float* __restrict__ dbeta, float* __restrict__ coef /* [3][C] */, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < kRedSplit; ++s) { s1 += partial[((size_t)c * kRedSplit + s) * 2]; s2 += partial[((size_t)c * kRedSplit + s) * 2 + 1]; }
  bn_relu_bwd_finalize_channel(s1, s2, c, C, gamma, invstd, count, dgamma, dbeta, coef);
}
// (the tile-statistics route: reduce + finalize in one launch, bit-identical -- see bn_tile_stats_fold)
inline __global__ void __launch_bounds__(1024) bn_tile_stats_bwd_finalize_kernel(const double* __restrict__ tile_stats, long n_tiles, const float* __restrict__ gamma,
                                                                          const float* __restrict__ invstd, long count, float* __restrict__ dgamma,
                                                                          float* __restrict__ dbeta, float* __restrict__ coef /* [3][C] */, int C) {
  bn_tile_stats_fold(tile_stats, n_tiles, [&](double s1, double s2, int c) {
    bn_relu_bwd_finalize_channel(s1, s2, c, C, gamma, invstd, count, dgamma, dbeta, coef);
  });
}

// dZ = k0*g - k1 - k2*xhat  (written over dA's buffer is allowed: dZ may alias dA)
template <bool FROM_Z>
__global__ void __launch_bounds__(256) bn_relu_bwd_apply_kernel(const float* dA, const float* __restrict__ a,
                                                                const float* __restrict__ z, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ coef,
                                                                float* dZ, long NC, int C, int HW) {
  const int hw4 = HW >> 2;
  const long total4 = NC * hw4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (long)gridDim.x * blockDim.x) {
    const long nc = t / hw4;
    const int c = (int)(nc % C);
    const float mu = mean[c], is = invstd[c], k0 = coef[c], k1 = coef[C + c], k2 = coef[2 * C + c];
    const t_f32x4 g = *reinterpret_cast<const t_f32x4*>(dA + t * 4);
    const t_f32x4 zv = *reinterpret_cast<const t_f32x4*>(z + t * 4);
    t_f32x4 av;
    if (FROM_Z) {
      const float sc = bn_scale(gamma[c], is), sh = beta[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) av[k] = fmaf(zv[k] - mu, sc, sh);
    } else {
      av = *reinterpret_cast<const t_f32x4*>(a + t * 4);
    }
    t_f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = av[k] > 0.0f ? g[k] : 0.0f;
      o[k] = k0 * gk - k1 - k2 * ((zv[k] - mu) * is);
    }
    *reinterpret_cast<t_f32x4*>(dZ + t * 4) = o;
  }
}

// ---- WBCE (utils/metric.py:15-20) -------------------------------------------------------------------------------
__device__ __forceinline__ float wbce_elem(float p, float y) {
  const float q = 1.0f - p;
  const float pc = fminf(fmaxf(p, 1e-7f), 1.0f), qc = fminf(fmaxf(q, 1e-7f), 1.0f);
  return -(q * q * y * logf(pc) + p * p * (1.0f - y) * logf(qc));
}

// partial[n][s] = sum of the element losses of sample n over slice s.     grid = (kRedSplit, N)
inline __global__ void __launch_bounds__(256) wbce_partial_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                           double* __restrict__ partial, long per_sample) {
  __shared__ double red[4];
  const int n = blockIdx.y, s = blockIdx.x;
  const float* pp = p + (size_t)n * per_sample;
  const float* yy = y + (size_t)n * per_sample;
  double acc = 0.0;
  for (long t = (long)s * 256 + threadIdx.x; t < per_sample; t += (long)kRedSplit * 256) acc += (double)wbce_elem(pp[t], yy[t]);
  const double r = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[(size_t)n * kRedSplit + s] = r;
}

// reduce != 0: out[0] = mean over everything; else out[n] = per-sample mean   (utils/metric.py:17-20)
inline __global__ void wbce_finalize_kernel(const double* __restrict__ partial, float* __restrict__ out, int N, long per_sample, int reduce) {
  if (reduce) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      double t = 0.0;
      for (int n = 0; n < N; ++n) for (int s = 0; s < kRedSplit; ++s) t += partial[(size_t)n * kRedSplit + s];
      out[0] = (float)(t / ((double)N * (double)per_sample));
    }
  } else {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) {
      double t = 0.0;
      for (int s = 0; s < kRedSplit; ++s) t += partial[(size_t)n * kRedSplit + s];
      out[n] = (float)(t / (double)per_sample);
    }
  }
}

// d(elem)/dp in closed form (the clamp's gradient is 0 outside [1e-7, 1])
__device__ __forceinline__ float wbce_elem_grad(float pv, float yv) {
  const float q = 1.0f - pv;
  const float pc = fminf(fmaxf(pv, 1e-7f), 1.0f), qc = fminf(fmaxf(q, 1e-7f), 1.0f);
  const float in_p = (pv >= 1e-7f && pv <= 1.0f) ? 1.0f : 0.0f;
  const float in_q = (q >= 1e-7f && q <= 1.0f) ? 1.0f : 0.0f;
  return -(-2.0f * q * yv * logf(pc) + q * q * yv * in_p / pc + 2.0f * pv * (1.0f - yv) * logf(qc) - pv * pv * (1.0f - yv) * in_q / qc);
}

// dL/dp = upstream[n or 0] * d(elem)/dp / denom
inline __global__ void __launch_bounds__(256) wbce_backward_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                            const float* __restrict__ upstream, int upstream_per_sample,
                                                            float inv_denom, float* __restrict__ dp, long per_sample, long total) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const float g = wbce_elem_grad(p[t], y[t]);
    const float up = upstream[upstream_per_sample ? (int)(t / per_sample) : 0];
    dp[t] = g * up * inv_denom;
  }
}

// ---- head forward with the loss: p = sigmoid(W a + b) written once, WBCE partial sums taken from the registers that hold it ---
// grid = (kHeadLossSplit, N): block (s, n) walks 4-pixel groups s, s + kHeadLossSplit, ... of sample n; partial[n][s] in fp64
// (fixed assignment: deterministic).  C input planes, L <= LT maps.
constexpr int kHeadLossSplit = 256;
template <int LT>
__global__ void __launch_bounds__(256) head1x1_sigmoid_wbce_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ b, const float* __restrict__ y,
                                                                   float* __restrict__ pout, double* __restrict__ partial, int C, int L, int HW) {
  __shared__ double red[4];
  const int n = blockIdx.y, hw4 = HW >> 2;
  double acc_loss = 0.0;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < hw4; t += kHeadLossSplit * 256) {
    const int px = t << 2;
    const float* xb = x + (size_t)n * C * HW + px;
    t_f32x4 acc[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) acc[l] = (t_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const t_f32x4 v = *reinterpret_cast<const t_f32x4*>(xb + (size_t)c * HW);
#pragma unroll
      for (int l = 0; l < LT; ++l) {
        const float wl = l < L ? w[l * C + c] : 0.0f;
        acc[l] += wl * v;
      }
    }
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      if (l < L) {
        const size_t o = ((size_t)n * L + l) * HW + px;
        t_f32x4 z = acc[l] + b[l];
        const t_f32x4 yy = *reinterpret_cast<const t_f32x4*>(y + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          z[k] = 1.0f / (1.0f + expf(-z[k]));
          acc_loss += (double)wbce_elem(z[k], yy[k]);
        }
        *reinterpret_cast<t_f32x4*>(pout + o) = z;
      }
    }
  }
  const double r = block_sum_256(acc_loss, red);
  if (threadIdx.x == 0) partial[(size_t)n * kHeadLossSplit + blockIdx.x] = r;
}

// the matching finalize: out[0] = mean over everything (reduce) or out[n] = per-sample means
inline __global__ void head_wbce_finalize_kernel(const double* __restrict__ partial, float* __restrict__ out, int N, long per_sample, int reduce) {
  if (reduce) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      double t = 0.0;
      for (int n = 0; n < N; ++n) for (int s = 0; s < kHeadLossSplit; ++s) t += partial[(size_t)n * kHeadLossSplit + s];
      out[0] = (float)(t / ((double)N * (double)per_sample));
    }
  } else {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) {
      double t = 0.0;
      for (int s = 0; s < kHeadLossSplit; ++s) t += partial[(size_t)n * kHeadLossSplit + s];
      out[n] = (float)(t / (double)per_sample);
    }
  }
}

// ---- head backward: p = sigmoid(W a + b) ------------------------------------------------------------------------
// dz = dP * p * (1-p);  dA[c] = sum_l W[l][c] dz[l];  dW[l][c] = sum dz[l]*a[c];  db[l] = sum dz[l]
// One workgroup walks pixel tiles of P = 128; partial dW/db per workgroup -> finalize.   L <= 16, C == 64.
// LDS traffic is what bounds it, so each staged value is read once per thread that needs it:
//   dA: a thread owns one pixel, holds its dz[0..L) in registers and takes the filter column of its (wave-uniform)
//       channel as 16-byte LDS broadcasts (scalar global loads instead were measured slower: their latency is exposed);
//   dW: a thread owns channel c for ALL l over a quarter of the tile's pixels: one a read and L/4 16-byte broadcast reads of
//       dz per pixel feed L FMAs; the four pixel quarters (waves) are summed in fixed order at the end.
// FUSED_WBCE: sigmoid + WBCELoss fused into the head (the north-star's wording): `dP` then holds the TARGETS y and dL/dp is formed on
// the fly from (p, y) and the loss node's upstream gradient -- the dP tensor (N*L*H*W floats) is neither written nor read.
constexpr int kHeadP = 128, kHeadC = 64, kHeadLMax = 16;
template <bool FUSED_WBCE>
__global__ void __launch_bounds__(256) head_backward_kernel(const float* __restrict__ dP, const float* __restrict__ p,
                                                            const float* __restrict__ a, const float* __restrict__ w,
                                                            float* __restrict__ dA, float* __restrict__ part /* [grid][L*C + L] */,
                                                            int N, int L, int HW, const float* __restrict__ upstream, int upstream_per_sample,
                                                            float inv_denom) {
  __shared__ __attribute__((aligned(16))) float dz_s[kHeadP * kHeadLMax];       // [px][l], l padded to 16 (zeros)
  __shared__ float a_s[kHeadC * (kHeadP + 1)];
  __shared__ __attribute__((aligned(16))) float w_s[kHeadC * kHeadLMax];       // [c][l]: the filter column of a channel, l padded (zeros)
  const int tid = threadIdx.x;
  for (int i = tid; i < kHeadC * kHeadLMax; i += 256) { const int c = i / kHeadLMax, l = i - c * kHeadLMax; w_s[i] = l < L ? w[l * kHeadC + c] : 0.0f; }
  const int tilesPer = (HW + kHeadP - 1) / kHeadP;
  const long nTiles = (long)N * tilesPer;
  const int c_own = tid & 63, pq = tid >> 6;           // dW ownership: channel c_own, pixels [32 pq, 32 pq + 32) of every tile
  float accW[kHeadLMax];
#pragma unroll
  for (int l = 0; l < kHeadLMax; ++l) accW[l] = 0.0f;
  float accB = 0.0f;                                     // thread tid < L owns db[tid]
  for (int i = tid; i < kHeadP * kHeadLMax; i += 256) dz_s[i] = 0.0f;           // the padding lanes l >= L stay zero
  // A tile travels global -> registers -> LDS; the loads of tile t+1 are issued before the arithmetic of tile t, so their
  // latency hides behind it.  Thread slots: 8 pieces of 4 pixels of `a` (channel = slot / 32), L/2 (l, px) pairs of dz.
  constexpr int NA = kHeadC * kHeadP / 4 / 256, NZ = kHeadLMax * kHeadP / 256;
  const bool vec_ok = (HW & 3) == 0;
  t_f32x4 ra[NA];
  float rz[NZ];
  auto fetch = [&](long tile) {
    const int n = (int)(tile / tilesPer);
    const int p0 = (int)(tile - (long)n * tilesPer) * kHeadP;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = tid + j * 256;
      const int c = e >> 5, px = (e & 31) * 4;
      const float* src = a + ((size_t)n * kHeadC + c) * HW + p0 + px;
      if (vec_ok && p0 + px + 3 < HW) ra[j] = *reinterpret_cast<const t_f32x4*>(src);
      else
#pragma unroll
        for (int k = 0; k < 4; ++k) ra[j][k] = (p0 + px + k < HW) ? src[k] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
      const int i = tid + j * 256;
      const int l = i / kHeadP, px = i - l * kHeadP;
      float v = 0.0f;
      if (l < L && p0 + px < HW) {
        const size_t o = ((size_t)n * L + l) * HW + p0 + px;
        const float pv = p[o];
        float dpv;
        if (FUSED_WBCE) dpv = wbce_elem_grad(pv, dP[o]) * upstream[upstream_per_sample ? n : 0] * inv_denom;      // = wbce_backward_kernel's dp
        else dpv = dP[o];
        v = dpv * pv * (1.0f - pv);
      }
      rz[j] = v;
    }
  };
  auto publish = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = tid + j * 256;
      float* d = a_s + (e >> 5) * (kHeadP + 1) + (e & 31) * 4;
      d[0] = ra[j][0]; d[1] = ra[j][1]; d[2] = ra[j][2]; d[3] = ra[j][3];
    }
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
      const int i = tid + j * 256;
      const int l = i / kHeadP, px = i - l * kHeadP;
      if (l < L) dz_s[px * kHeadLMax + l] = rz[j];
    }
  };
  if ((long)blockIdx.x < nTiles) fetch(blockIdx.x);
  for (long tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    const int n = (int)(tile / tilesPer);
    const int p0 = (int)(tile - (long)n * tilesPer) * kHeadP;
    __syncthreads();                                      // the previous tile's readers are done
    publish();
    __syncthreads();
    if (tile + gridDim.x < nTiles) fetch(tile + gridDim.x);
    // dA: thread -> pixel px = tid % 128, channels c = (tid/128) + 2k (uniform per wave: scalar filter loads)
    {
      const int px = tid & (kHeadP - 1);
      float dzr[kHeadLMax];
#pragma unroll
      for (int q = 0; q < kHeadLMax / 4; ++q) {
        const t_f32x4 v = *reinterpret_cast<const t_f32x4*>(dz_s + px * kHeadLMax + 4 * q);
        dzr[4 * q] = v[0]; dzr[4 * q + 1] = v[1]; dzr[4 * q + 2] = v[2]; dzr[4 * q + 3] = v[3];
      }
      const int ch = __builtin_amdgcn_readfirstlane(tid >> 7);
      if (p0 + px < HW) {
        float* o = dA + (size_t)n * kHeadC * HW + p0 + px;
#pragma unroll 4
        for (int c = ch; c < kHeadC; c += 2) {
          float s = 0.0f;
#pragma unroll
          for (int q = 0; q < kHeadLMax / 4; ++q) {
            if (4 * q < L) {                              // wave-uniform address: a 16-byte broadcast read per four maps
              const t_f32x4 wv = *reinterpret_cast<const t_f32x4*>(w_s + c * kHeadLMax + 4 * q);
              s = fmaf(wv[0], dzr[4 * q], s); s = fmaf(wv[1], dzr[4 * q + 1], s);
              s = fmaf(wv[2], dzr[4 * q + 2], s); s = fmaf(wv[3], dzr[4 * q + 3], s);
            }
          }
          o[(size_t)c * HW] = s;
        }
      }
    }
    // dW / db partials
    {
      const float* arow = a_s + c_own * (kHeadP + 1) + pq * 32;
      const float* dzq = dz_s + pq * 32 * kHeadLMax;
      for (int px = 0; px < 32; ++px) {
        const float av = arow[px];
#pragma unroll
        for (int q = 0; q < kHeadLMax / 4; ++q) {
          if (4 * q < L) {
            const t_f32x4 v = *reinterpret_cast<const t_f32x4*>(dzq + px * kHeadLMax + 4 * q);
            accW[4 * q] = fmaf(v[0], av, accW[4 * q]); accW[4 * q + 1] = fmaf(v[1], av, accW[4 * q + 1]);
            accW[4 * q + 2] = fmaf(v[2], av, accW[4 * q + 2]); accW[4 * q + 3] = fmaf(v[3], av, accW[4 * q + 3]);
          }
        }
      }
    }
    if (tid < L) { float s = accB; for (int px = 0; px < kHeadP; ++px) s += dz_s[px * kHeadLMax + tid]; accB = s; }
  }
  // fold the four pixel quarters in the fixed order pq = 0..3 (a_s is free now)
  __syncthreads();
  float* red = a_s;                                      // [pq][l][c]
#pragma unroll
  for (int l = 0; l < kHeadLMax; ++l) red[(pq * kHeadLMax + l) * kHeadC + c_own] = accW[l];
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * (L * kHeadC + L);
  for (int i = tid; i < L * kHeadC; i += 256) {
    const int l = i / kHeadC, c = i - l * kHeadC;
    out[i] = ((red[(0 * kHeadLMax + l) * kHeadC + c] + red[(1 * kHeadLMax + l) * kHeadC + c]) + red[(2 * kHeadLMax + l) * kHeadC + c]) +
             red[(3 * kHeadLMax + l) * kHeadC + c];
  }
  if (tid < L) out[L * kHeadC + tid] = accB;
}

// Round 6: the same three gradients on the matrix pipe (the kernel above took 0.47 ms of a training step's critical chain for 0.94 GB of traffic --
// its 1024 multiply-adds per pixel went through LDS broadcasts).  MFMA 16x16x4 (fp32: an fmaf chain in k order, bitwise), NO LDS in the loop: a wave
// owns groups of 16 pixels (HW % 16 == 0, C = 64, L <= 8):
//   * dz[px][l] is formed ONCE per group in the weight gradient's operand layout -- lane (l = lane & 15 < L, pixel quarter kq = lane >> 4): a
//     16-byte read of p (and of y / dP) = the four pixels 4 kq .. 4 kq + 3 of map l -- and handed to the data gradient's layout (lane = pixel,
//     kq = map) by four ds_bpermute (no LDS allocation) + one select;
//   * dA[px][c] = sum_l dz[px][l] w[l][c]: A = dz (M = pixel, K = map), B = w (loop-invariant registers), 4 channel blocks x <= 2 K steps; a lane's
//     four accumulator registers are four CONSECUTIVE pixels of one channel: one 16-byte store (the same fmaf order over l as the kernel above);
//   * dW[l][c] += sum_px dz[px][l] a[c][px]: A = dz (M = map, K = pixel), B = a read as 16 bytes along the pixels (K order = the read's
//     component: any order, both operands agree), 4 channel blocks x 4 K steps, accumulators live across the wave's groups; db from the same dz.
// The four waves' dW / db are folded in a fixed order into part[block] (then sum_partials_wave_kernel, as before).  Deterministic.
template <bool FUSED_WBCE>
__global__ void __launch_bounds__(256) head_backward_mfma_kernel(const float* __restrict__ dP, const float* __restrict__ p,
                                                                 const float* __restrict__ a, const float* __restrict__ w,
                                                                 float* __restrict__ dA, float* __restrict__ part /* [grid][L*C + L] */,
                                                                 int N, int L, int HW, const float* __restrict__ upstream, int upstream_per_sample,
                                                                 float inv_denom) {
  __shared__ float red[4 * (8 * kHeadC + 8)];            // [wave][l 8][c 64] + [wave][l 8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int KS = (L + 3) >> 2;                           // K steps of the data gradient (maps in fours)
  // B operand of the data gradient: w[l = 4 kk + kq][c = 16 cb + m] (zero beyond L)
  float wB[4][2];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int l = 4 * kk + kq;
      wB[cb][kk] = l < L ? w[l * kHeadC + 16 * cb + m] : 0.0f;
    }
  t_f32x4 accW[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) accW[cb] = t_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float accB = 0.0f;
  const long gpi = HW >> 4;                              // groups per image
  const long groups = (long)N * gpi;
  const long stride = (long)gridDim.x * 4;
  for (long g = (long)blockIdx.x * 4 + wave; g < groups; g += stride) {
    const int n = (int)(g / gpi);
    const int p0 = (int)(g - (long)n * gpi) << 4;
    // ---- dz in the weight gradient's layout: lane (map m, pixels 4 kq .. 4 kq + 3)
    t_f32x4 dzW = {0.0f, 0.0f, 0.0f, 0.0f};
    if (m < L) {
      const size_t o = ((size_t)n * L + m) * HW + p0 + 4 * kq;
      const t_f32x4 pv = *reinterpret_cast<const t_f32x4*>(p + o);
      const t_f32x4 yv = *reinterpret_cast<const t_f32x4*>(dP + o);
      const float up = FUSED_WBCE ? upstream[upstream_per_sample ? n : 0] : 1.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float dpv = FUSED_WBCE ? wbce_elem_grad(pv[t], yv[t]) * up * inv_denom : yv[t];      // (the kernel above's expression, product for product)
        dzW[t] = dpv * pv[t] * (1.0f - pv[t]);
      }
      accB += (dzW[0] + dzW[1]) + (dzW[2] + dzW[3]);
    }
    // ---- the activation rows of the group: lane (channel 16 cb + m, pixels 4 kq .. 4 kq + 3)
    t_f32x4 a4[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) a4[cb] = *reinterpret_cast<const t_f32x4*>(a + ((size_t)n * kHeadC + 16 * cb + m) * HW + p0 + 4 * kq);
    // ---- data gradient: A[pixel m][map 4 kk + kq] = dzW of lane (map) + 16 (m >> 2), component m & 3
    float dzA[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int src = (4 * kk + kq) + 16 * (m >> 2);
      const float v0 = __shfl(dzW[0], src, 64), v1 = __shfl(dzW[1], src, 64), v2 = __shfl(dzW[2], src, 64), v3 = __shfl(dzW[3], src, 64);
      const int t = m & 3;
      dzA[kk] = t == 0 ? v0 : (t == 1 ? v1 : (t == 2 ? v2 : v3));
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      t_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dzA[0], wB[cb][0], acc, 0, 0, 0);
      if (KS > 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dzA[1], wB[cb][1], acc, 0, 0, 0);
      // acc[r] = dA[pixel 4 kq + r][channel 16 cb + m]
      *reinterpret_cast<t_f32x4*>(dA + ((size_t)n * kHeadC + 16 * cb + m) * HW + p0 + 4 * kq) = acc;
    }
    // ---- weight gradient: K step t = the t-th pixel of every lane's four
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int t = 0; t < 4; ++t) accW[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzW[t], a4[cb][t], accW[cb], 0, 0, 0);
  }
  // ---- fold: accW[cb][r] = dW[map 4 kq + r][channel 16 cb + m] of this wave; db: the four pixel quarters of map m
  accB += __shfl_xor(accB, 16, 64);
  accB += __shfl_xor(accB, 32, 64);
  float* rw = red + wave * (8 * kHeadC + 8);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int l = 4 * kq + r;
      if (l < 8) rw[l * kHeadC + 16 * cb + m] = accW[cb][r];
    }
  if (lane < 8) rw[8 * kHeadC + lane] = accB;
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * (L * kHeadC + L);
  constexpr int WS = 8 * kHeadC + 8;
  for (int i = tid; i < L * kHeadC; i += 256) out[i] = ((red[i] + red[WS + i]) + red[2 * WS + i]) + red[3 * WS + i];
  if (tid < L) out[L * kHeadC + tid] = ((red[8 * kHeadC + tid] + red[WS + 8 * kHeadC + tid]) + red[2 * WS + 8 * kHeadC + tid]) + red[3 * WS + 8 * kHeadC + tid];
}

// out[i] = sum_b part[b][i] in fixed order (double accumulate).  Used for head dW/db and the wgrad split-K slabs.
inline __global__ void __launch_bounds__(256) sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, long n, int nparts) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < nparts; ++b) s += (double)part[(size_t)b * n + i];
    out[i] = (float)s;
  }
}

inline __global__ void __launch_bounds__(256) fill_zero_kernel(float* __restrict__ p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.0f;
}

// Same sum (same order, so the same bits), 16 bytes per lane and four slabs in flight: n % 4 == 0, 16-byte aligned.
typedef float to_f32x4 __attribute__((ext_vector_type(4)));
inline __global__ void __launch_bounds__(256) sum_partials_vec4_kernel(const float* __restrict__ part, float* __restrict__ out, long n4,
                                                                int nparts) {
  const to_f32x4* src = reinterpret_cast<const to_f32x4*>(part);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 4 <= nparts; b += 4) {
      const to_f32x4 v0 = src[(size_t)b * n4 + i], v1 = src[(size_t)(b + 1) * n4 + i];
      const to_f32x4 v2 = src[(size_t)(b + 2) * n4 + i], v3 = src[(size_t)(b + 3) * n4 + i];
      s0 += (double)v0[0]; s1 += (double)v0[1]; s2 += (double)v0[2]; s3 += (double)v0[3];
      s0 += (double)v1[0]; s1 += (double)v1[1]; s2 += (double)v1[2]; s3 += (double)v1[3];
      s0 += (double)v2[0]; s1 += (double)v2[1]; s2 += (double)v2[2]; s3 += (double)v2[3];
      s0 += (double)v3[0]; s1 += (double)v3[1]; s2 += (double)v3[2]; s3 += (double)v3[3];
    }
    for (; b < nparts; ++b) {
      const to_f32x4 v = src[(size_t)b * n4 + i];
      s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
    }
    to_f32x4 o; o[0] = (float)s0; o[1] = (float)s1; o[2] = (float)s2; o[3] = (float)s3;
    reinterpret_cast<to_f32x4*>(out)[i] = o;
  }
}

// out[i] = sum_b part[b*stride + offset + i]  (fixed order, double accumulate)
inline __global__ void __launch_bounds__(256) sum_partials_strided_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                                   int nparts, long stride, long offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < nparts; ++b) s += (double)part[(size_t)b * stride + offset + i];
    out[i] = (float)s;
  }
}

// The same strided sum for FEW outputs and MANY parts (head dW/db: 520 outputs x 1024 workgroup partials): one wave per
// output, lane l adds parts l, l+64, ... in order, then the 64 lane sums are folded in lane order -- a fixed summation
// tree, so the result is deterministic (it differs from the serial order above only in association).
inline __global__ void __launch_bounds__(256) sum_partials_wave_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                                int nparts, long stride, long offset) {
  const int lane = threadIdx.x & 63;
  const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= n) return;                                      // whole wave leaves together
  double s = 0.0;
  for (int b = lane; b < nparts; b += 64) s += (double)part[(size_t)b * stride + offset + o];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if (lane == 0) out[o] = (float)s;
}

// ---- pooling / upsampling backward ------------------------------------------------------------------------------
// dx = dskip + route(dpool): the FIRST maximum of each 2x2 window (row-major, strict '>') receives the pooled gradient.
inline __global__ void __launch_bounds__(256) maxpool2x2_bwd_add_kernel(const float* __restrict__ x, const float* __restrict__ dpool,
                                                                 const float* __restrict__ dskip, float* __restrict__ dx,
                                                                 long NC, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = NC * Ho * Wo;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ow = (int)(t % Wo);
    const long u = t / Wo;
    const int oh = (int)(u % Ho);
    const long nc = u / Ho;
    const size_t i00 = ((size_t)nc * H + 2 * oh) * W + 2 * ow;
    const float v0 = x[i00], v1 = x[i00 + 1], v2 = x[i00 + W], v3 = x[i00 + W + 1];
    int am = 0; float m = v0;
    if (v1 > m || v1 != v1) { m = v1; am = 1; }
    if (v2 > m || v2 != v2) { m = v2; am = 2; }
    if (v3 > m || v3 != v3) { m = v3; am = 3; }
    const float g = dpool[t];
    const bool has = dskip != nullptr;
    dx[i00] = (has ? dskip[i00] : 0.0f) + (am == 0 ? g : 0.0f);
    dx[i00 + 1] = (has ? dskip[i00 + 1] : 0.0f) + (am == 1 ? g : 0.0f);
    dx[i00 + W] = (has ? dskip[i00 + W] : 0.0f) + (am == 2 ? g : 0.0f);
    dx[i00 + W + 1] = (has ? dskip[i00 + W + 1] : 0.0f) + (am == 3 ? g : 0.0f);
  }
}

// Round 6: the same routing, for a pooled tensor that is a = ReLU(BN(z)) of a block this step normalised (model.py:9-10 in front of model.py:48,
// 51, 54), which ALSO takes that block's two BatchNorm + ReLU backward sums of the gradient it writes: tile_stats[c][s] = (sum g, sum g * zhat)
// over slice s of channel c, g = dx where a > 0 (bn_tile_stats_bwd_finalize_kernel's input: the separate sums pass over (dA, z) is gone).  a is
// recomputed from z with the forward's own expression (bn_apply_relu_kernel: bit-identical, so the first maximum is the same one) -- the pass
// reads z instead of a: the sums cost no HBM traffic.  grid = (n_slices, C), 256 threads; a thread = two windows (2 rows x 4 columns).
// Needs W % 4 == 0, H % 2 == 0, 16-byte aligned z / dskip / dx, 8-byte aligned dpool.
inline __global__ void __launch_bounds__(256) maxpool2x2_bwd_add_bnsums_kernel(const float* __restrict__ z, const float* __restrict__ dpool,
                                                                        const float* __restrict__ dskip, float* __restrict__ dx,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        double* __restrict__ tile_stats, int N, int C, int H, int W) {
  __shared__ double red[4];
  const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
  const int Ho = H >> 1, W4 = W >> 2, Wo = W >> 1;
  const int per_n = Ho * W4;
  const long total = (long)N * per_n;
  const float mu = mean[c], is = invstd[c], sc = bn_scale(gamma[c], is), sh = beta[c];
  const bool has = dskip != nullptr;
  double s1 = 0.0, s2 = 0.0;
  for (long t = (long)s * 256 + threadIdx.x; t < total; t += (long)S * 256) {
    const int n = (int)(t / per_n);
    const int r = (int)(t - (long)n * per_n);
    const int oh = r / W4, q = r - oh * W4;
    const size_t plane = (size_t)n * C + c;
    const size_t i0 = (plane * H + 2 * oh) * W + 4 * q;
    const t_f32x4 z0 = *reinterpret_cast<const t_f32x4*>(z + i0), z1 = *reinterpret_cast<const t_f32x4*>(z + i0 + W);
    const t_f32x2 gp = *reinterpret_cast<const t_f32x2*>(dpool + (plane * Ho + oh) * Wo + 2 * q);
    t_f32x4 d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (has) { d0 = *reinterpret_cast<const t_f32x4*>(dskip + i0); d1 = *reinterpret_cast<const t_f32x4*>(dskip + i0 + W); }
    t_f32x4 a0, a1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float y0 = fmaf(z0[k] - mu, sc, sh), y1 = fmaf(z1[k] - mu, sc, sh);
      a0[k] = y0 > 0.0f ? y0 : 0.0f;
      a1[k] = y1 > 0.0f ? y1 : 0.0f;
    }
#pragma unroll
    for (int wnd = 0; wnd < 2; ++wnd) {
      const float v0 = a0[2 * wnd], v1 = a0[2 * wnd + 1], v2 = a1[2 * wnd], v3 = a1[2 * wnd + 1];
      int am = 0; float m = v0;                           // (a is never NaN here: a NaN y fails y > 0)
      if (v1 > m) { m = v1; am = 1; }
      if (v2 > m) { m = v2; am = 2; }
      if (v3 > m) { m = v3; am = 3; }
      const float g = gp[wnd];
      d0[2 * wnd] += am == 0 ? g : 0.0f;
      d0[2 * wnd + 1] += am == 1 ? g : 0.0f;
      d1[2 * wnd] += am == 2 ? g : 0.0f;
      d1[2 * wnd + 1] += am == 3 ? g : 0.0f;
    }
    *reinterpret_cast<t_f32x4*>(dx + i0) = d0;
    *reinterpret_cast<t_f32x4*>(dx + i0 + W) = d1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g0 = a0[k] > 0.0f ? d0[k] : 0.0f, g1 = a1[k] > 0.0f ? d1[k] : 0.0f;
      s1 += (double)g0;
      s2 += (double)g0 * (double)((z0[k] - mu) * is);
      s1 += (double)g1;
      s2 += (double)g1 * (double)((z1[k] - mu) * is);
    }
  }
  const double r1 = block_sum_256(s1, red);
  const double r2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) { tile_stats[((size_t)c * S + s) * 2] = r1; tile_stats[((size_t)c * S + s) * 2 + 1] = r2; }
}

// nearest 2x upsample backward: d_lo[h][w] = sum of the 2x2 block of d_hi
inline __global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const float* __restrict__ d_hi, float* __restrict__ d_lo,
                                                             long NC, int Hl, int Wl) {
  const long total = NC * Hl * Wl;
  const int Wh = 2 * Wl;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int w = (int)(t % Wl);
    const long u = t / Wl;
    const int h = (int)(u % Hl);
    const long nc = u / Hl;
    const size_t i = ((size_t)nc * 2 * Hl + 2 * h) * Wh + 2 * w;
    d_lo[t] = (d_hi[i] + d_hi[i + 1]) + (d_hi[i + Wh] + d_hi[i + Wh + 1]);
  }
}

// ---- mixup (train.py:32-40): out[n] = x[n]*lam[n] + x[perm[n]]*(1-lam[n]) -----------------------------------------
inline __global__ void __launch_bounds__(256) mixup_kernel(const float* __restrict__ x, const float* __restrict__ lam,
                                                    const int* __restrict__ perm, float* __restrict__ out, int N, long per_sample) {
  const long per4 = per_sample >> 2;
  const long total4 = (long)N * per4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (long)gridDim.x * blockDim.x) {
    const int n = (int)(t / per4);
    const long o = (t - (long)n * per4) << 2;
    const float l = lam[n];
    const t_f32x4 a = *reinterpret_cast<const t_f32x4*>(x + (size_t)n * per_sample + o);
    const t_f32x4 b = *reinterpret_cast<const t_f32x4*>(x + (size_t)perm[n] * per_sample + o);
    *reinterpret_cast<t_f32x4*>(out + (size_t)n * per_sample + o) = a * l + b * (1.0f - l);
  }
}

}  // namespace tnv3
