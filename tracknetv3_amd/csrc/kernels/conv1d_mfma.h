// conv1d_mfma.h -- the dense layers of InpaintNet (model.py:76-87, 103-110: Conv1d(k=3, padding='same') + LeakyReLU with
// 32..384 input and 32..256 output channels, sequences of L = 16) on the fp32 matrix cores.
//
// GEMM view per layer: M = output channels, N = positions (16 per sequence, all sequences of the batch), K = Cin x 3.
//   MFMA 32x32x2: an N-tile is 32 positions = two whole sequences; a K-pair is two input channels at one tap.
//   lane l supplies A[co = l&31][ci + (l>>5)][tap] and B[ci + (l>>5)][position (l&31) + tap - 1].
//   'same' padding needs no halo in LDS: the out-of-sequence taps (position 0 at tap 0, position 15 at tap 2) are
//   zeroed in the lane, so sequences sit densely ([ci][seq*16 + pos]) and the B reads of a half-wave are 32 consecutive
//   floats.  The filter is staged as stored in the state_dict ([co][ci][3], rows of 24 floats per 8-channel chunk,
//   padded to 25 -> conflict-free column reads); no pre-packing, the C ABI takes the nn.Conv1d weight itself.
// Channel concats (model.py:120,122,124) are two source pointers; a chunk of 8 channels always comes from one of them.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "conv1d_k3.h"
#include "conv3x3_mfma.h"

namespace tnv3 {

template <int MT_, int NTW_, int WM_, int WN_, int CC_ = 8, int WK_ = 1>
struct Conv1dMfmaCfg {
  static constexpr int MT = MT_, NTW = NTW_, WM = WM_, WN = WN_;
  static constexpr int WK = WK_;                     // waves that split the K range of a stage (small batches: a 32 x 32 tile per
                                                     // workgroup spreads few sequences over many CUs, its four waves share the chain)
  static constexpr int NT = WM * WN * WK * 64;
  static constexpr int MB = MT * 32 * WM;            // output channels per workgroup
  static constexpr int NB = NTW * 32 * WN;           // positions per workgroup
  static constexpr int SB = NB / 16;                 // sequences per workgroup
  static constexpr int CC = CC_;                     // input channels per pipeline stage (8: throughput; 32: few workgroups,
                                                     // where the chain of load -> barrier -> MFMA stages is the latency)
  static constexpr int WROW = CC * 3 + 1;            // padded filter row
  static constexpr int INP = NB + 8;                 // padded input row: 4 floats of slack on both sides
  static constexpr int W_FLOATS = MB * WROW, IN_FLOATS = CC * INP;
  static constexpr int BUF_FLOATS = ((W_FLOATS + 3) / 4) * 4 + IN_FLOATS;
  static constexpr int WQ = CC * 3 / 4;              // 16-byte pieces per filter row and stage
  static constexpr int NW4 = (MB * WQ + NT - 1) / NT; // 16-byte filter loads per thread per stage
  static constexpr int NI4 = (CC * SB * 4 + NT - 1) / NT;
  static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
  static constexpr int CW = CC / WK;                 // channels of a stage one wave contracts
  static_assert(CC % (2 * WK) == 0, "each K-split wave needs whole channel pairs");
  static_assert(WK == 1 || WK * WM * WN * MT * NTW * 16 * 64 <= 2 * BUF_FLOATS, "the K-split reduction reuses the stage buffers");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv1d_k3_mfma_kernel(const Conv1dArgs a) {
  constexpr int MT = Cfg::MT, NTW = Cfg::NTW, WN = Cfg::WN, NT = Cfg::NT, MB = Cfg::MB, SB = Cfg::SB, CC = Cfg::CC;
  constexpr int WROW = Cfg::WROW, INP = Cfg::INP, L = 16;
  constexpr int IN_OFF = ((Cfg::W_FLOATS + 3) / 4) * 4;
  __shared__ __attribute__((aligned(16))) float lds[2 * Cfg::BUF_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = (wave / WN) % Cfg::WM, wk = wave / (WN * Cfg::WM);
  const int half = lane >> 5, bl = lane & 31;
  const int Cin = a.C0 + a.C1, Cout = a.Cout;
  const int nMB = (Cout + MB - 1) / MB;
  const int mb = blockIdx.x % nMB;                   // the M-blocks of one sequence group are neighbours: inputs hit L2
  const long n0 = (long)(blockIdx.x / nMB) * SB;
  const int co0 = mb * MB;

  f32x4 rw[Cfg::NW4], ri[Cfg::NI4];
  auto load_stage = [&](int k) {
    const int c0 = k * CC;
#pragma unroll
    for (int i = 0; i < Cfg::NW4; ++i) {
      const int idx = tid + i * NT;
      const int q = idx % Cfg::WQ, co_l = idx / Cfg::WQ;
      const bool ok = idx < MB * Cfg::WQ && co0 + co_l < Cout;
      const float* src = a.w + ((size_t)(ok ? co0 + co_l : 0) * Cin + c0) * 3 + 4 * q;
      rw[i] = *reinterpret_cast<const f32x4*>(src);
      if (!ok) rw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool first = c0 < a.C0;                    // workgroup-uniform: C0 % CC == 0
    const float* sbase = first ? a.src0 : a.src1;
    const int Cs = first ? a.C0 : a.C1, cb = first ? c0 : c0 - a.C0;
#pragma unroll
    for (int i = 0; i < Cfg::NI4; ++i) {
      const int idx = tid + i * NT;
      const int q4 = idx & 3, c = (idx >> 2) % CC, s = idx / (4 * CC);
      const bool ok = idx < CC * SB * 4 && n0 + s < a.N;
      const float* src = sbase + ((size_t)(ok ? n0 + s : 0) * Cs + cb + c) * L + 4 * q4;
      ri[i] = *reinterpret_cast<const f32x4*>(src);
      if (!ok) ri[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_stage = [&](int buf) {
    float* w_s = lds + buf * Cfg::BUF_FLOATS;
    float* in_s = w_s + IN_OFF;
#pragma unroll
    for (int i = 0; i < Cfg::NW4; ++i) {
      const int idx = tid + i * NT;
      if (idx < MB * Cfg::WQ) {
        float* d = w_s + (idx / Cfg::WQ) * WROW + 4 * (idx % Cfg::WQ);
        d[0] = rw[i][0]; d[1] = rw[i][1]; d[2] = rw[i][2]; d[3] = rw[i][3];
      }
    }
#pragma unroll
    for (int i = 0; i < Cfg::NI4; ++i) {
      const int idx = tid + i * NT;
      if (idx < CC * SB * 4) {
        const int q4 = idx & 3, c = (idx >> 2) % CC, s = idx / (4 * CC);
        *reinterpret_cast<f32x4*>(in_s + c * INP + 4 + s * L + 4 * q4) = ri[i];
      }
    }
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.0f;

  const bool z_first = (bl & 15) == 0, z_last = (bl & 15) == 15;
  const int a_off = (wm * MT * 32 + bl) * WROW + (wk * Cfg::CW + half) * 3;
  const int b_off = IN_OFF + (wk * Cfg::CW + half) * INP + 4 + wn * NTW * 32 + bl - 1;

  const int nChunks = Cin / CC;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int k = 0; k < nChunks; ++k) {
    const int buf = k & 1;
    if (k + 1 < nChunks) load_stage(k + 1);
    const float* A = lds + buf * Cfg::BUF_FLOATS + a_off;
    const float* B = lds + buf * Cfg::BUF_FLOATS + b_off;
    constexpr int NSTEP = (Cfg::CW / 2) * 3;
    float av[2][MT], bv[2][NTW];
    auto read_step = [&](int s, float (&ar)[MT], float (&br)[NTW]) {
      const int cp = s / 3, tap = s - 3 * cp;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ar[mt] = A[mt * 32 * WROW + cp * 6 + tap];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const float v = B[cp * 2 * INP + j * 32 + tap];
        br[j] = ((tap == 0 && z_first) || (tap == 2 && z_last)) ? 0.0f : v;
      }
    };
    read_step(0, av[0], bv[0]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 1 < NSTEP) read_step(s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][mt], bv[s & 1][j], acc[mt][j], 0, 0, 0);
    }
    if (k + 1 < nChunks) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- K-split waves: fold the partial accumulators through LDS (fixed order wk = 1, 2, ...: deterministic)
  if (Cfg::WK > 1) {
    constexpr int TILE = MT * NTW * 16 * 64;                       // floats one wave holds
    float* red = lds;                                              // stage buffers are free after the last barrier
    const int slot = ((wk * Cfg::WM + wm) * WN + wn) * TILE + lane;
    if (wk > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[slot + ((mt * NTW + j) * 16 + r) * 64] = acc[mt][j][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int q = 1; q < Cfg::WK; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[mt][j][r] += red[((q * Cfg::WM + wm) * WN + wn) * TILE + lane + ((mt * NTW + j) * 16 + r) * 64];
  }

  // ---- epilogue: bias + activation, [N][Cout][16]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= Cout) continue;
      const float bias = a.b[co];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int pg = (wn * NTW + j) * 32 + bl;
        const long n = n0 + (pg >> 4);
        if (n >= a.N) continue;
        float v = acc[mt][j][r] + bias;
        if (a.act == 1) v = v > 0.0f ? v : 0.01f * v;
        else if (a.act == 2) v = 1.0f / (1.0f + expf(-v));
        a.dst[((size_t)n * Cout + co) * L + (pg & 15)] = v;
      }
    }
  }
}

}  // namespace tnv3
