// wgrad_wino_mfma.h -- weight gradient of the plain 3x3 'same' convolution in Winograd F(2x2, 3x3) form.
//
// With the forward written as  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A  per 2x2 output tile, the gradient of the
// transformed filter U = G g G^T is
//     dU_xi[co][ci] = sum_{n, tiles}  Yh_xi[co][tile] * V_xi[ci][tile],    Yh = A dY A^T (4x4 from the 2x2 tile of dZ),  V = B^T d B
// -- sixteen independent GEMMs with K = all tiles of the batch, 16 instead of 36 multiply-adds per (co, ci, tile) -- and
//     dW = G^T dU G                                                        (wgrad_wino_fold_kernel, after the split-K sum).
// Same gradient as autograd of nn.Conv2d in exact arithmetic (train.py:95); fp32 rounding of the +-1, 1/2 transforms as in
// the forward kernel (conv3x3_wino_mfma.h).
//
// MFMA 32x32x2: M = 32 co, N = 32 ci, K = 2 tiles; a wave keeps all 16 xi of its 32 x 32 block of dU (256 accumulator
// registers, one wave per SIMD); a workgroup = 64 co x 64 ci walks its share of the K range in chunks of 8 tiles (a 2 x 16
// pixel strip): the raw strips of dZ and X arrive by LDS DMA two chunks ahead, the threads transform them into the LDS
// operands Yh[xi][co][tile], V[xi][ci][tile] (channel stride 9: conflict-free), then 4 x 16 MFMAs per wave.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct WgradWinoArgs {
  const float* x;       // [N][Cin][H][W]
  const float* dz;      // [N][Cout][H][W]
  const float* zeros;   // >= 64 zero floats
  float* part;          // [splitK][16][Cout][Cin]
  int N, Cin, Cout, H, W, splitK;
};

struct WgradWinoCfg {
  static constexpr int NT = 256, MB = 64, CB = 64, TCH = 8;            // 8 tiles (2 x 16 pixels) per chunk
  static constexpr int TS = TCH + 1;                                    // channel stride of the transformed operands
  static constexpr int OP_FLOATS = 16 * 64 * TS;                        // one transformed operand: [xi][channel][tile]
  static constexpr int DZ_RAW = 64 * 2 * 16;                            // [co][2 rows][16 px]
  static constexpr int XW = 24, X_RAW = 64 * 4 * XW;                    // [ci][4 rows][24 px: columns 16j-4 .. 16j+19]
  static constexpr int RAW_STAGE = DZ_RAW + X_RAW;
  static constexpr int NDZ = DZ_RAW / 4 / NT, NX = X_RAW / 4 / NT;      // 16-byte DMA pieces per thread and chunk: 2 + 6
  static constexpr int DMA_PER_CHUNK = NDZ + NX;
  static constexpr int LDS_FLOATS = 2 * OP_FLOATS + 2 * RAW_STAGE;
  static_assert(DZ_RAW % (4 * NT) == 0 && X_RAW % (4 * NT) == 0, "pieces must deal evenly");
};

inline __global__ void __launch_bounds__(WgradWinoCfg::NT) wgrad_wino_mfma_kernel(const WgradWinoArgs a) {
  using Cfg = WgradWinoCfg;
  constexpr int NT = Cfg::NT, TS = Cfg::TS, XW = Cfg::XW;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;
  float* v_s = lds + Cfg::OP_FLOATS;
  float* raw_s = lds + 2 * Cfg::OP_FLOATS;                 // two stages of [dz strip | x strip]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;                  // wm: co half, wn: ci half
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int nCB = Cin / 64;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * 64, ci0 = cb * 64;
  const int segW = W / 16, rowsT = H / 2;
  const int nChunksAll = a.N * rowsT * segW;
  // this workgroup's chunks: ks, ks + splitK, ...
  const int nMine = nChunksAll > ks ? (nChunksAll - ks + a.splitK - 1) / a.splitK : 0;

  const float* zsrc = a.zeros + (lane & 15) * 4;
  const int wbase = wave * 64;
  auto dma_chunk = [&](int q, int stage) {                  // q-th chunk of this workgroup
    const int c = ks + q * a.splitK;
    const int n = c / (rowsT * segW);
    const int rem = c - n * (rowsT * segW);
    const int i = rem / segW, j = rem - i * segW;           // tile row, 16-pixel segment
    float* rs = raw_s + stage * Cfg::RAW_STAGE;
    const float* dzb = a.dz + ((size_t)n * Cout + co0) * HW + (size_t)(2 * i) * W + 16 * j;
#pragma unroll
    for (int p = 0; p < Cfg::NDZ; ++p) {
      const int e = tid + p * NT;                           // piece of [co][row][4 pieces]
      const int q4 = e & 3, r = (e >> 2) & 1, co = e >> 3;
      lds_dma16(dzb + (size_t)co * HW + r * W + 4 * q4, rs + (p * NT + wbase) * 4);
    }
    const float* xb = a.x + ((size_t)n * Cin + ci0) * HW;
#pragma unroll
    for (int p = 0; p < Cfg::NX; ++p) {
      const int e = tid + p * NT;                           // piece of [ci][4 rows][6 pieces]
      const int q6 = e % 6, t2 = e / 6;
      const int r = t2 & 3, ci = t2 >> 2;
      const int gh = 2 * i - 1 + r, gw = 16 * j - 4 + 4 * q6;
      const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
      lds_dma16(ok ? xb + (size_t)ci * HW + (size_t)gh * W + gw : zsrc, rs + Cfg::DZ_RAW + (p * NT + wbase) * 4);
    }
  };
  // raw strips -> Yh[xi][co][t], V[xi][ci][t]: thread handles patches p = tid, tid + 256 of each operand (64 channels x 8 tiles)
  auto transform = [&](int stage) {
    const float* rs = raw_s + stage * Cfg::RAW_STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = tid + q * NT;
      const int t = p & 7, ch = p >> 3;
      {   // Yh = A dY A^T,  A = [1 0; 1 1; 1 -1; 0 -1]
        const float* d = rs + ch * 32 + 2 * t;
        const float y00 = d[0], y01 = d[1], y10 = d[16], y11 = d[17];
        const float r0[2] = {y00, y01}, r1[2] = {y00 + y10, y01 + y11}, r2[2] = {y00 - y10, y01 - y11}, r3[2] = {-y10, -y11};
        float* o = yh_s + ch * TS + t;
        const float* rr[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[(i * 4 + 0) * 64 * TS] = rr[i][0];
          o[(i * 4 + 1) * 64 * TS] = rr[i][0] + rr[i][1];
          o[(i * 4 + 2) * 64 * TS] = rr[i][0] - rr[i][1];
          o[(i * 4 + 3) * 64 * TS] = -rr[i][1];
        }
      }
      {   // V = B^T d B on the 4x4 input patch: strip rows 0..3, columns 2t+3 .. 2t+6 (strip column 0 = image column 16j-4)
        const float* d = rs + Cfg::DZ_RAW + ch * (4 * XW) + 2 * t + 3;
        float e[4][4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
          const float d0 = d[jx], d1 = d[XW + jx], d2 = d[2 * XW + jx], d3 = d[3 * XW + jx];
          e[0][jx] = d0 - d2; e[1][jx] = d1 + d2; e[2][jx] = d2 - d1; e[3][jx] = d1 - d3;
        }
        float* o = v_s + ch * TS + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[(r * 4 + 0) * 64 * TS] = e[r][0] - e[r][2];
          o[(r * 4 + 1) * 64 * TS] = e[r][1] + e[r][2];
          o[(r * 4 + 2) * 64 * TS] = e[r][2] - e[r][1];
          o[(r * 4 + 3) * 64 * TS] = e[r][1] - e[r][3];
        }
      }
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;

  const int a_off = (wm * 32 + bl) * TS + half;
  const int b_off = (wn * 32 + bl) * TS + half;
  auto wait_landed = [&](bool newest_in_flight) {
    if (newest_in_flight) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(Cfg::DMA_PER_CHUNK));
    else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
  };
  auto publish_lds = [&]() {
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  if (nMine > 0) {
    dma_chunk(0, 0);
    if (nMine > 1) dma_chunk(1, 1);
    wait_landed(nMine > 1);
    __builtin_amdgcn_s_barrier();
    transform(0);
    publish_lds();
  }
  for (int q = 0; q < nMine; ++q) {
    if (q + 2 < nMine) dma_chunk(q + 2, q & 1);            // stage q&1 held the strips of chunk q (already transformed)
    const float* A = yh_s + a_off;
    const float* B = v_s + b_off;
    constexpr int NSTEP = 4 * 16;                           // (tile pair, xi)
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int tp = s >> 4, xi = s & 15;
      av[s % RING] = A[xi * 64 * TS + 2 * tp];
      bv[s % RING] = B[xi * 64 * TS + 2 * tp];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc[s & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], acc[s & 15], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    if (q + 1 < nMine) {
      wait_landed(q + 2 < nMine);
      __builtin_amdgcn_s_barrier();
      transform((q + 1) & 1);
      publish_lds();
    }
  }

  // partial slab: part[ks][xi][co][ci]
  float* slab = a.part + (size_t)ks * 16 * Cout * Cin;
  const int ci = ci0 + wn * 32 + bl;
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      slab[((size_t)xi * Cout + co) * Cin + ci] = acc[xi][r];
    }
}

// dW[co][ci][3][3] = G^T (sum_ks part[ks][.][co][ci]) G.
// A block folds 16 (co, ci) elements: thread = (element el, xi row q, quarter p of the ks range) sums four xi over its
// quarter in fp64; the quarters are combined in the fixed order p = 0..3 through LDS (deterministic), then 48 threads
// apply G^T . G, three outputs each.  One thread per element (the first version) left a 128 x 128 layer with 64 blocks
// walking 1024 dependent-latency loads each: 0.3 ms per launch for 67 MB.
inline __global__ void __launch_bounds__(256) wgrad_wino_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout, int Cin,
                                                              int splitK) {
  __shared__ double red[4][16][16];                  // [p][xi][el]
  __shared__ float us[16][16];                       // [xi][el]
  const long n = (long)Cout * Cin;
  const int tid = threadIdx.x, el = tid & 15, q = (tid >> 4) & 3, p = tid >> 6;
  const int kq = (splitK + 3) / 4;
  const int k0 = p * kq, k1 = (k0 + kq < splitK) ? k0 + kq : splitK;
  for (long e0 = (long)blockIdx.x * 16; e0 < n; e0 += (long)gridDim.x * 16) {
    const long e = e0 + el;
    const bool live = e < n;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (live)
      for (int k = k0; k < k1; ++k) {
        const float* src = part + ((size_t)k * 16 + 4 * q) * n + e;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += (double)src[(size_t)j * n];
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[p][4 * q + j][el] = s[j];
    __syncthreads();
    if (p == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xi = 4 * q + j;
        us[xi][el] = (float)(((red[0][xi][el] + red[1][xi][el]) + red[2][xi][el]) + red[3][xi][el]);
      }
    }
    __syncthreads();
    if (p == 0 && q < 3 && live) {
      // t = G^T u (3x4), dW = t G (3x3);  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]; this thread: row r = q of dW
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u0 = us[j][el], u1 = us[4 + j][el], u2 = us[8 + j][el], u3 = us[12 + j][el];
        t[j] = q == 0 ? u0 + 0.5f * (u1 + u2) : (q == 1 ? 0.5f * (u1 - u2) : 0.5f * (u1 + u2) + u3);
      }
      float* o = dw + e * 9 + q * 3;
      o[0] = t[0] + 0.5f * (t[1] + t[2]);
      o[1] = 0.5f * (t[1] - t[2]);
      o[2] = 0.5f * (t[1] + t[2]) + t[3];
    }
    __syncthreads();
  }
}

}  // namespace tnv3
