// pointwise.h -- the HBM-bound pieces of TrackNet.forward (model.py:57-73):
//   * head: nn.Conv2d(64, L, (1,1)) + bias -> nn.Sigmoid          (model.py:54-55,71-72)
//   * nn.MaxPool2d((2,2), stride=(2,2))                             (model.py:59,61,63)
// One pass over the data each, 16-byte accesses, no reuse -> bounded by HBM bandwidth (~8 TB/s).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float pw_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// y[n][l][p] = sigmoid(b[l] + sum_c w[l][c] * x[n][c][p]);  each thread owns 4 consecutive pixels.
// LT outputs are accumulated per pass over the C input planes (L <= LT: a single pass).
template <int LT>
__global__ void __launch_bounds__(256) head1x1_sigmoid_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int N, int C, int L, int HW, int apply_sigmoid) {
  const int hw4 = HW >> 2;
  const long total = (long)N * hw4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int n = (int)(t / hw4);
    const int p = (int)(t - (long)n * hw4) << 2;
    const float* xb = x + (size_t)n * C * HW + p;
    for (int l0 = 0; l0 < L; l0 += LT) {
      pw_f32x4 acc[LT];
#pragma unroll
      for (int l = 0; l < LT; ++l) acc[l] = (pw_f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < C; ++c) {
        const pw_f32x4 v = *reinterpret_cast<const pw_f32x4*>(xb + (size_t)c * HW);
#pragma unroll
        for (int l = 0; l < LT; ++l) {
          const float wl = (l0 + l < L) ? w[(l0 + l) * C + c] : 0.0f;   // wave-uniform -> scalar load
          acc[l] += wl * v;
        }
      }
#pragma unroll
      for (int l = 0; l < LT; ++l) {
        if (l0 + l < L) {
          pw_f32x4 z = acc[l] + b[l0 + l];
          if (apply_sigmoid) {
            z[0] = sigmoidf_(z[0]); z[1] = sigmoidf_(z[1]); z[2] = sigmoidf_(z[2]); z[3] = sigmoidf_(z[3]);
          }
          *reinterpret_cast<pw_f32x4*>(y + ((size_t)n * L + l0 + l) * HW + p) = z;
        }
      }
    }
  }
}

// 2x2 / stride-2 max pooling over [NC][H][W] planes; each thread produces two horizontally adjacent outputs
// from two 16-byte row reads.  Comparison order = row-major window order with strict '>' (PyTorch's tie rule;
// immaterial here).  NaN propagates like torch (a NaN input wins).
inline __global__ void __launch_bounds__(256) maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         long NC, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1, Wo2 = Wo >> 1;
  const long total = NC * Ho * Wo2;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int q = (int)(t % Wo2);
    const long u = t / Wo2;
    const int oh = (int)(u % Ho);
    const long nc = u / Ho;
    const float* r0 = x + ((size_t)nc * H + 2 * oh) * W + 4 * q;
    const pw_f32x4 a = *reinterpret_cast<const pw_f32x4*>(r0);
    const pw_f32x4 c = *reinterpret_cast<const pw_f32x4*>(r0 + W);
    auto mx = [](float m, float v) { return (v > m || v != v) ? v : m; };
    f32x2 o;
    o[0] = mx(mx(mx(a[0], a[1]), c[0]), c[1]);
    o[1] = mx(mx(mx(a[2], a[3]), c[2]), c[3]);
    *reinterpret_cast<f32x2*>(y + ((size_t)nc * Ho + oh) * Wo + 2 * q) = o;
  }
}

// Diagnostic: sustained v_mfma_f32_32x32x2_f32 rate of the chip as it is clocked under load -- every wave issues
// `iters` x 8 independent accumulator chains from registers only (no LDS, no global traffic in the loop).  Used by
// scripts/microbench.py to put the conv kernels' TFLOP/s next to what the matrix pipe delivers at the same time.
inline __global__ void __launch_bounds__(256) mfma_f32_probe_kernel(float* __restrict__ out, int iters, float a0, float b0) {
  typedef float pf32x16 __attribute__((ext_vector_type(16)));
  pf32x16 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  float a = a0 + (float)(threadIdx.x & 7) * 1e-3f, b = b0 - (float)(threadIdx.x & 3) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace tnv3
