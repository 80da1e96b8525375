// wgrad3x3_mfma.h -- weight gradient of the 3x3 convolution on the fp32 matrix cores.
//
//   dW[co][ci][kh][kw] = sum_{n,h,w} dZ[n][co][h][w] * X[n][ci][h+kh-1][w+kw-1]          (X zero-padded)
//
// What autograd derives for nn.Conv2d(k=3, padding='same', bias=False) in Conv2DBlock (model.py:8) when
// train.py:95 calls loss.backward().  GEMM view: M = co, N = ci (one 32-wide N-tile per filter tap), K = pixels.
//   MFMA 32x32x2: lane l supplies A[co = l&31][pixel p + (l>>5)] and B[pixel p + (l>>5)][ci = l&31].
//   LDS: dZ tile [co][TR*TC] (row stride odd -> the 32 lanes of a half-wave hit 32 banks), X halo tile
//   [ci][TR+2][TC+2] (plane stride odd, same reason).  Each wave owns one 32-channel M-tile x one 32-channel ci
//   group and keeps all 9 taps in registers (9 x 16 accumulators): 1 A read + 9 B reads per 9 MFMAs.
// K (all pixels of the batch) is split over `splitK` workgroups per (co, ci) block; each writes its partial slab
// part[ks][Cout][Cin][9]; sum_partials_kernel adds the slabs in a fixed order (deterministic, no atomics).
// X may be the two-source, nearest-upsampled concat input of a decoder-entry layer (same loader as the forward).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct WgradArgs {
  const float* src0;   // X part 1: [N][C0][H][W] or [N][C0][H/2][W/2] when up0
  const float* src1;   // X part 2: [N][C1][H][W] or nullptr
  const float* dz;     // [N][Cout][H][W]
  float* part;         // [splitK][Cout][C0+C1][9]
  int N, C0, C1, Cout, H, W, up0, splitK;
};

template <int WM_, int WC_, int TR_ = 2, int TC_ = 32>
struct WgradCfg {
  static constexpr int WM = WM_, WC = WC_, TR = TR_, TC = TC_;
  static constexpr int NT = WM * WC * 64;
  static constexpr int MB = WM * 32, CB = WC * 32;
  static constexpr int PIX = TR * TC, PIXP = PIX + 1;
  static constexpr int TRp = TR + 2, TCp = TC + 2, PLANE = TRp * TCp, PLANEP = PLANE | 1;
  static constexpr int DZ_FLOATS = MB * PIXP, X_FLOATS = CB * PLANEP;
  static constexpr int LDS_BYTES = (DZ_FLOATS + X_FLOATS) * 4;
  static_assert(TC % 2 == 0, "one MFMA consumes two horizontally adjacent pixels");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT, 2) wgrad3x3_mfma_kernel(const WgradArgs a) {
  constexpr int WC = Cfg::WC, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT, MB = Cfg::MB, CB = Cfg::CB;
  constexpr int PIX = Cfg::PIX, PIXP = Cfg::PIXP, TRp = Cfg::TRp, TCp = Cfg::TCp, PLANE = Cfg::PLANE, PLANEP = Cfg::PLANEP;
  __shared__ float lds[Cfg::DZ_FLOATS + Cfg::X_FLOATS];
  float* dz_s = lds;
  float* x_s = lds + Cfg::DZ_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave % WC, wm = wave / WC;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cout = a.Cout, C0 = a.C0, C1 = a.C1, Cin = C0 + C1;
  const int HW = H * W;
  const int H0 = a.up0 ? (H >> 1) : H, W0 = a.up0 ? (W >> 1) : W, HW0 = H0 * W0;

  const int nCB = (Cin + CB - 1) / CB;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * MB, ci0 = cb * CB;

  const int tilesH = (H + TR - 1) / TR, tilesW = (W + TC - 1) / TC;
  const int nTiles = a.N * tilesH * tilesW;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int a_off = (wm * 32 + bl) * PIXP + half;
  const int b_off = (wc * 32 + bl) * PLANEP + half;

  for (int tile = ks; tile < nTiles; tile += a.splitK) {
    const int n = tile / (tilesH * tilesW);
    const int trem = tile - n * (tilesH * tilesW);
    const int h0 = (trem / tilesW) * TR, w0 = (trem % tilesW) * TC;
    __syncthreads();
    // ---- stage dZ tile [MB][TR][TC]
    for (int idx = tid; idx < MB * PIX; idx += NT) {
      const int col = idx % TC, t2 = idx / TC;
      const int r = t2 % TR, co_l = t2 / TR;
      const int co = co0 + co_l, gh = h0 + r, gw = w0 + col;
      float v = 0.0f;
      if (co < Cout && gh < H && gw < W) v = a.dz[((size_t)n * Cout + co) * HW + gh * W + gw];
      dz_s[co_l * PIXP + r * TC + col] = v;
    }
    // ---- stage X halo tile [CB][TR+2][TC+2]
    for (int idx = tid; idx < CB * PLANE; idx += NT) {
      const int ci_l = idx / PLANE, rr = idx - ci_l * PLANE;
      const int tr = rr / TCp, tc = rr - tr * TCp;
      const int ci = ci0 + ci_l, gh = h0 - 1 + tr, gw = w0 - 1 + tc;
      float v = 0.0f;
      if (ci < Cin && gh >= 0 && gh < H && gw >= 0 && gw < W) {
        if (ci < C0) v = a.up0 ? a.src0[((size_t)n * C0 + ci) * HW0 + (gh >> 1) * W0 + (gw >> 1)]
                               : a.src0[((size_t)n * C0 + ci) * HW + gh * W + gw];
        else v = a.src1[((size_t)n * C1 + (ci - C0)) * HW + gh * W + gw];
      }
      x_s[ci_l * PLANEP + tr * TCp + tc] = v;
    }
    __syncthreads();
    // ---- MFMA: K-step = two horizontally adjacent pixels of one tile row
    const float* A = dz_s + a_off;
    const float* B = x_s + b_off;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
#pragma unroll
      for (int j = 0; j < TC / 2; ++j) {
        const float av = A[r * TC + 2 * j];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int kh = tap / 3, kw = tap - 3 * kh;
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, B[(r + kh) * TCp + 2 * j + kw], acc[tap], 0, 0, 0);
        }
      }
    }
  }

  // ---- partial slab: part[ks][co][ci][tap]
  float* slab = a.part + (size_t)ks * Cout * Cin * 9;
  const int ci = ci0 + wc * 32 + bl;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (co < Cout && ci < Cin) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) slab[((size_t)co * Cin + ci) * 9 + tap] = acc[tap][r];
    }
  }
}

}  // namespace tnv3
