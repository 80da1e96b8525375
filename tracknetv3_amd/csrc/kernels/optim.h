// optim.h -- the optimiser side of a training step on the device, so that a step issues no host-side
// tensor arithmetic, no H2D copy and no host sync:
//   * multi-tensor Adam (torch.optim.Adam defaults, train.py:242) over ALL parameter tensors in ONE launch, with the
//     gradient clipping of torch.nn.utils.clip_grad_norm_ (train.py:165) applied on the fly and an optional zero-grad;
//   * the global gradient norm behind that clipping (two small launches: fixed-shape fp64 partials, fixed-order finalize);
//   * multi-tensor SGD with momentum (train.py:244);
//   * the mixup draws of train.py:33-36 -- lambda ~ Beta(alpha, alpha) folded to >= 0.5 and a random partner permutation --
//     from a counter-based Philox4x32-10 generator (Beta through two Gammas, Marsaglia-Tsang).
// The tensor table travels in the kernel-argument buffer (<= 4 KB): no device-side pointer array to fill, hence no copy.
// Arithmetic follows torch's foreach implementation operation by operation (each line below is one rounded fp32 op of
// torch/optim/adam.py::_multi_tensor_adam), so parameters track torch.optim.Adam to the last bit or two.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

constexpr int kOptMaxTensors = 64;       // tensors per launch (the host side splits longer lists)
constexpr int kOptChunk = 4096;          // elements per workgroup (256 threads x 16)
constexpr int kNormSplit = 256;          // partial sums of squares of the gradient-norm reduction

struct OptTensorTable {
  float* param[kOptMaxTensors];
  float* grad[kOptMaxTensors];
  float* s0[kOptMaxTensors];             // Adam: exp_avg            SGD: momentum buffer
  float* s1[kOptMaxTensors];             // Adam: exp_avg_sq         SGD: unused
  int first_chunk[kOptMaxTensors + 1];   // prefix sum of ceil(numel / kOptChunk)
  long numel[kOptMaxTensors];
  int count;
};

__device__ __forceinline__ int opt_find_tensor(const OptTensorTable& t, int chunk) {
  int lo = 0, hi = t.count;              // first_chunk[lo] <= chunk < first_chunk[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.first_chunk[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- global gradient norm (clip_grad_norm_: norm of the per-tensor L2 norms == sqrt of the sum of all squares)
// partial[b] = sum of g^2 over the chunks b, b + kNormSplit, ... (fp64); fixed assignment => deterministic
inline __global__ void __launch_bounds__(256) grad_sumsq_partial_kernel(const OptTensorTable t, double* __restrict__ partial) {
  __shared__ double red[4];
  const int total = t.first_chunk[t.count];
  double s = 0.0;
  for (int c = blockIdx.x; c < total; c += kNormSplit) {
    const int k = opt_find_tensor(t, c);
    const long base = (long)(c - t.first_chunk[k]) * kOptChunk;
    const float* g = t.grad[k];
    const long n = t.numel[k];
#pragma unroll 4
    for (int i = threadIdx.x; i < kOptChunk; i += 256) {
      const long e = base + i;
      if (e < n) { const double v = (double)g[e]; s += v * v; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (total_norm + 1e-6))   (clip_grad_norm_'s formula)
inline __global__ void grad_norm_finalize_kernel(const double* __restrict__ partial, int nparts, float max_norm, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int b = 0; b < nparts; ++b) s += partial[b];
  const float total = (float)sqrt(s);
  float coef = max_norm / (total + 1e-6f);
  out[0] = total;
  out[1] = coef < 1.0f ? coef : 1.0f;
}

// One element of torch's foreach Adam.  torch runs these as SEPARATE kernels, so products and sums of different lines never
// fuse there; inside one ATen functor the compiler contracts a*b+c (clang's HIP default), which the explicit fmaf calls
// restate.  Contraction across the lines below is switched off so that the rounding points are torch's.
struct AdamScalars { float one_minus_beta1, beta2, one_minus_beta2, neg_step_size, bc2_sqrt, eps, weight_decay; };
__device__ __forceinline__ void adam_element(float& par, float& grad, float& mm, float& vv, const AdamScalars& a, float coef, bool do_clip) {
#pragma clang fp contract(off)
  if (do_clip) grad = grad * coef;                                     // clip_grad_norm_: g.mul_(clip_coef_clamped)
  if (a.weight_decay != 0.0f) grad = fmaf(a.weight_decay, par, grad);  // grad.add(param, alpha=weight_decay)
  mm = fmaf(a.one_minus_beta1, grad - mm, mm);                         // exp_avg.lerp_(grad, 1 - beta1): self + w * (end - self)
  vv = vv * a.beta2;                                                   // exp_avg_sq.mul_(beta2)
  vv = fmaf(a.one_minus_beta2, grad * grad, vv);                       // .addcmul_(grad, grad, value): self + value * (t1 * t2)
  float denom = sqrtf(vv);                                             // exp_avg_sq.sqrt()
  denom = denom / a.bc2_sqrt;                                          // .div_(bias_correction2_sqrt)
  denom = denom + a.eps;                                               // .add_(eps)
  par = fmaf(a.neg_step_size, mm / denom, par);                        // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// ---- Adam, every tensor of the table in one launch.  clip: device pointer to the clip coefficient (or nullptr = 1).
// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t): computed on the host in double, rounded to float -- exactly
// what torch hands its foreach kernels.
inline __global__ void __launch_bounds__(256) adam_multi_kernel(const OptTensorTable t, const AdamScalars a, const float* __restrict__ clip,
                                                                int zero_grad) {
  const int k = opt_find_tensor(t, blockIdx.x);
  const long base = (long)(blockIdx.x - t.first_chunk[k]) * kOptChunk;
  float* __restrict__ p = t.param[k];
  float* __restrict__ g = t.grad[k];
  float* __restrict__ m = t.s0[k];
  float* __restrict__ v = t.s1[k];
  const long n = t.numel[k];
  const float coef = clip ? clip[0] : 1.0f;
  const bool do_clip = clip != nullptr;
#pragma unroll 4
  for (int i = threadIdx.x; i < kOptChunk; i += 256) {
    const long e = base + i;
    if (e >= n) break;
    float grad = g[e], par = p[e], mm = m[e], vv = v[e];
    const float clipped = do_clip ? grad * coef : grad;
    adam_element(par, grad, mm, vv, a, coef, do_clip);
    m[e] = mm;
    v[e] = vv;
    p[e] = par;
    if (zero_grad) g[e] = 0.0f;
    else if (do_clip) g[e] = clipped;                              // the clipped gradient is what the caller sees afterwards
  }
}

__device__ __forceinline__ float sgd_element(float par, float grad, float& buf, float lr, float momentum, float weight_decay,
                                             int first_step, float coef, bool do_clip) {
#pragma clang fp contract(off)
  if (do_clip) grad = grad * coef;
  if (weight_decay != 0.0f) grad = fmaf(weight_decay, par, grad);      // grad.add(param, alpha=weight_decay)
  if (momentum != 0.0f) {
    if (first_step) buf = grad;                                        // buf = clone(grad)
    else { buf = buf * momentum; buf = buf + grad; }                   // buf.mul_(momentum).add_(grad, alpha=1 - dampening)
    grad = buf;
  }
  return fmaf(-lr, grad, par);                                         // param.add_(grad, alpha=-lr)
}

// ---- SGD (+ momentum, dampening 0, no Nesterov): buf = g (first step) | momentum * buf + g;  p -= lr * buf
inline __global__ void __launch_bounds__(256) sgd_multi_kernel(const OptTensorTable t, float lr, float momentum, float weight_decay,
                                                               int first_step, const float* __restrict__ clip, int zero_grad) {
  const int k = opt_find_tensor(t, blockIdx.x);
  const long base = (long)(blockIdx.x - t.first_chunk[k]) * kOptChunk;
  float* __restrict__ p = t.param[k];
  float* __restrict__ g = t.grad[k];
  float* __restrict__ buf = t.s0[k];
  const long n = t.numel[k];
  const float coef = clip ? clip[0] : 1.0f;
#pragma unroll 4
  for (int i = threadIdx.x; i < kOptChunk; i += 256) {
    const long e = base + i;
    if (e >= n) break;
    float grad = g[e];
    const float par = p[e];
    float bv = momentum != 0.0f && !first_step ? buf[e] : 0.0f;
    p[e] = sgd_element(par, grad, bv, lr, momentum, weight_decay, first_step, coef, clip != nullptr);
    if (momentum != 0.0f) buf[e] = bv;
    if (zero_grad) g[e] = 0.0f;
    else if (clip != nullptr) g[e] = grad * coef;                     // as clip_grad_norm_ does: the caller sees the clipped gradient
  }
}

// ---- mixup draws on the device (train.py:33-36).  Philox4x32-10 (Salmon et al. 2011), key = seed, counter = (step, lane, draw).
struct Philox {
  uint32_t c[4], k[2];
  __device__ __forceinline__ static uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
  __device__ __forceinline__ void round() {
    const uint32_t hi0 = mulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = mulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  __device__ __forceinline__ void run(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
    k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
    c[0] = (uint32_t)ctr_lo; c[1] = (uint32_t)(ctr_lo >> 32); c[2] = (uint32_t)ctr_hi; c[3] = (uint32_t)(ctr_hi >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) round();
  }
};

struct PhiloxStream {                      // a private sequence of uniforms for one (seed, step, lane)
  uint64_t seed, step;
  uint32_t lane, block, have;
  Philox px;
  __device__ __forceinline__ uint32_t next_u32() {
    if (have == 0) { px.run(seed, step, ((uint64_t)lane << 32) | block); ++block; have = 4; }
    const uint32_t r = have == 4 ? px.c[0] : (have == 3 ? px.c[1] : (have == 2 ? px.c[2] : px.c[3]));   // no dynamic indexing: stays in registers
    --have;
    return r;
  }
  __device__ __forceinline__ float uniform() { return ((float)(next_u32() >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)
  __device__ __forceinline__ float normal() {                                                                     // Box-Muller
    const float u1 = uniform(), u2 = uniform();
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);     // hardware log2 / cos: plenty for a sampler, no scratch
  }
};

// Gamma(shape a, scale 1): Marsaglia & Tsang (2000) for a >= 1; for a < 1, Gamma(a) = Gamma(a + 1) * U^(1/a)
__device__ __forceinline__ float philox_gamma(PhiloxStream& s, float a) {
  const float boost = a < 1.0f ? __expf(__logf(s.uniform()) / a) : 1.0f;
  if (a < 1.0f) a += 1.0f;
  const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
  for (int it = 0; it < 64; ++it) {
    const float x = s.normal();
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    const float u = s.uniform();
    if (u < 1.0f - 0.0331f * x * x * x * x || __logf(u) < 0.5f * x * x + d * (1.0f - v + __logf(v))) return boost * d * v;
  }
  return boost * d;                                             // (never reached in practice: acceptance > 95 % per trial)
}

// lam[n] = max(B, 1 - B), B ~ Beta(alpha, alpha);  perm = a uniformly random permutation of 0..n-1 (Fisher-Yates, one lane).
// One workgroup of 256 threads; n <= 65536 samples per call.
inline __global__ void __launch_bounds__(256) mixup_draw_kernel(float* __restrict__ lam, int* __restrict__ perm, int n, float alpha,
                                                                unsigned long long seed, unsigned long long step) {
  for (int i = threadIdx.x; i < n; i += 256) {
    PhiloxStream s{seed, step, (uint32_t)i + 1u, 0u, 0u, {}};
    const float x = philox_gamma(s, alpha), y = philox_gamma(s, alpha);
    const float b = (x + y) > 0.0f ? x / (x + y) : 0.5f;
    lam[i] = b > 1.0f - b ? b : 1.0f - b;
  }
  if (threadIdx.x == 0) {
    PhiloxStream s{seed, step, 0u, 0u, 0u, {}};
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int i = n - 1; i > 0; --i) {
      // unbiased index in [0, i]: rejection on the top of the 32-bit range
      const uint32_t bound = (uint32_t)i + 1u, limit = 0xFFFFFFFFu - (0xFFFFFFFFu % bound + 1u) % bound;
      uint32_t r = s.next_u32();
      while (r > limit) r = s.next_u32();
      const int j = (int)(r % bound);
      const int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
    }
  }
}

}  // namespace tnv3
