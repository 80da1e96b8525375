// preproc.h -- frame preprocessing in front of TrackNet, integer / byte work bound by HBM:
//   * Pillow's 8-bit two-pass BICUBIC resample (`Image.resize`, dataset.py:450,629): horizontal pass u8 -> u8, vertical
//     pass u8 -> u8 fused with `np.moveaxis(img, -1, 0)` and `frames /= 255.` (dataset.py:451-459) -> fp32 CHW.
//     Fixed-point: acc = (1 << 21) + sum_k pixel[k] * coeff[k] (coeff scaled by 2^22), clip8(acc >> 22) -- bit-exact with
//     src/libImaging/Resample.c.  Coefficient tables come from the host (they depend only on the two sizes).
//   * temporal median background (`np.median(frame_arr, 0).astype('uint8')`, dataset.py:101-105): per byte position a
//     256-bin histogram over the T frames in LDS, then the two middle order statistics, (a + b) >> 1.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

constexpr int kResampleBits = 22;
constexpr int kResampleMaxRowBytes = 32 * 1024;     // one source row (W*C bytes) staged in LDS

__device__ __forceinline__ unsigned char resample_clip8(int acc) {
  const int v = acc >> kResampleBits;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// One workgroup per (frame, source row).  src [F][H][W][C] u8 -> dst [F][H][OW][C] u8.
__global__ void __launch_bounds__(256) resample_h_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                            const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                            const int* __restrict__ kk, int ksize, int H, int W, int C, int OW) {
  __shared__ unsigned char row_s[kResampleMaxRowBytes];
  const size_t row = blockIdx.x;                       // f * H + y
  const unsigned char* s = src + row * (size_t)W * C;
  for (int i = threadIdx.x; i < W * C; i += 256) row_s[i] = s[i];
  __syncthreads();
  unsigned char* d = dst + row * (size_t)OW * C;
  for (int o = threadIdx.x; o < OW * C; o += 256) {
    const int xx = o / C, c = o - xx * C;
    const int x0 = xmin[xx], n = xcnt[xx];
    const int* k = kk + (size_t)xx * ksize;
    int acc = 1 << (kResampleBits - 1);
    for (int x = 0; x < n; ++x) acc += (int)row_s[(x0 + x) * C + c] * k[x];
    d[o] = resample_clip8(acc);
  }
}

// One workgroup per (frame, output row).  src [F][H][OW][C] u8 -> dst_f32 [F][C][OH][OW] = lut[u8]  and/or  dst_u8 [F][OH][OW][C].
__global__ void __launch_bounds__(256) resample_v_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst_f32,
                                                            unsigned char* __restrict__ dst_u8, const int* __restrict__ ymin,
                                                            const int* __restrict__ ycnt, const int* __restrict__ kk, int ksize,
                                                            const float* __restrict__ lut, int H, int OW, int C, int OH) {
  const int f = blockIdx.x / OH, yy = blockIdx.x - f * OH;
  const int y0 = ymin[yy], n = ycnt[yy];
  const int* k = kk + (size_t)yy * ksize;
  const unsigned char* s = src + ((size_t)f * H + y0) * OW * C;
  for (int o = threadIdx.x; o < OW * C; o += 256) {
    int acc = 1 << (kResampleBits - 1);
    for (int y = 0; y < n; ++y) acc += (int)s[(size_t)y * OW * C + o] * k[y];
    const unsigned char v = resample_clip8(acc);
    const int xx = o / C, c = o - xx * C;
    if (dst_u8) dst_u8[((size_t)f * OH + yy) * OW * C + o] = v;
    if (dst_f32) dst_f32[(((size_t)f * C + c) * OH + yy) * OW + xx] = lut[v];
  }
}

// Median over T frames of P bytes each: frames [T][P] u8 -> med [P] u8 = floor((v[(T-1)/2] + v[T/2]) / 2).
// 128 threads x 256 32-bit bins = 128 KB of LDS; bin-major layout keeps the 32 lanes of a half-wave on 32 banks.
// med2 (optional): a + b as uint16 = twice the float median (np.median of an even count ends in .5): what the
// difference-frame modes subtract (dataset.py:439) without leaving integer arithmetic.
__global__ void __launch_bounds__(128) median_u8_kernel(const unsigned char* __restrict__ frames, unsigned char* __restrict__ med,
                                                        unsigned short* __restrict__ med2, int T, long P) {
  __shared__ unsigned int hist_s[256 * 128];          // [bin][thread]: 128 KB of the CU's 160 KB
  const int tid = threadIdx.x;
  const long p = (long)blockIdx.x * 128 + tid;
  for (int b = 0; b < 256; ++b) hist_s[b * 128 + tid] = 0u;
  if (p < P) {
    const unsigned char* f = frames + p;
    int t = 0;
    for (; t + 8 <= T; t += 8) {
      unsigned char v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = f[(size_t)(t + u) * P];
#pragma unroll
      for (int u = 0; u < 8; ++u) hist_s[(int)v[u] * 128 + tid] += 1u;
    }
    for (; t < T; ++t) hist_s[(int)f[(size_t)t * P] * 128 + tid] += 1u;
    const unsigned r0 = (unsigned)((T - 1) / 2), r1 = (unsigned)(T / 2);
    unsigned cum = 0;
    int a = -1, b2 = -1;
    for (int b = 0; b < 256; ++b) {
      cum += hist_s[b * 128 + tid];
      if (a < 0 && cum > r0) a = b;
      if (b2 < 0 && cum > r1) { b2 = b; break; }
    }
    if (med) med[p] = (unsigned char)((a + b2) >> 1);
    if (med2) med2[p] = (unsigned short)(a + b2);
  }
}

// Difference frame of bg_mode 'subtract' / 'subtract_concat' (dataset.py:439, 443):
//   out[f][p] = uint8( sum_c |frame[f][p][c] - median[p][c]| )   with a float64 median -> truncate, then wrap mod 256.
// In integers: s2 = sum_c |2*v - med2|,  out = (s2 >> 1) & 255.   frames [F][P][3] u8, med2 [P][3] u16 -> out [F][P] u8.
__global__ void __launch_bounds__(256) absdiff_sum_u8_kernel(const unsigned char* __restrict__ frames,
                                                             const unsigned short* __restrict__ med2,
                                                             unsigned char* __restrict__ out, int F, long P) {
  const long total = (long)F * P;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long p = t % P;
    const unsigned char* v = frames + t * 3;
    const unsigned short* m = med2 + p * 3;
    int s2 = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const int d = 2 * (int)v[c] - (int)m[c]; s2 += d < 0 ? -d : d; }
    out[t] = (unsigned char)((s2 >> 1) & 255);
  }
}

}  // namespace tnv3
