// conv_up2x_mfma.h -- the upsampled half of a decoder-entry layer (model.py:65,67,69: Conv2DBlock applied to
// torch.cat([nn.Upsample(scale_factor=2)(x), skip], dim=1)) computed AT THE LOW RESOLUTION.
//
// A 3x3 'same' convolution over a nearest-2x-upsampled tensor only ever sees 2x2 distinct source pixels per output
// pixel: for output row h = 2i + ph the taps kh = 0,1,2 read low-res rows {i-1, i, i} (ph = 0) or {i, i, i+1} (ph = 1),
// and likewise for columns.  Summing the filter taps that hit the same source pixel,
//     Wq[cls = 2ph+pw][a][b] = sum_{kh in R[ph][a]} sum_{kw in R[pw][b]} W[kh][kw],   R[0] = {{0},{1,2}},  R[1] = {{0,1},{2}},
// turns the upsampled part of the layer into four 2x2 correlations on the low-res tensor (one per output parity class):
//     P[co][2i+ph][2j+pw] = sum_{ci,a,b} Wq[cls][ci][a][b][co] * Xlow[ci][i + a + ph - 1][j + b + pw - 1]
// i.e. 4/9 of the multiply-adds, and the low-res tensor is read once instead of four times.  The skip half of the layer
// stays a regular 3x3 convolution (conv3x3_mfma_kernel) that takes P as its `addend` and applies BN + ReLU.
// Same algebra as the reference in exact arithmetic; in fp32 the pre-summed taps differ from the tap-by-tap chain by
// ordinary rounding (a few 1e-7 relative), far inside the 1e-4 heat-map bar.
//
// MFMA 32x32x2 mapping: an N-tile is 32 output pixels of one full-res row with the SAME column parity (w = 2j + pw,
// j = 32 consecutive low-res columns), so all its lanes share the class filter and the B operand is 32 consecutive
// low-res floats of the LDS halo tile; a wave owns both column parities of its rows and stores (pw = 0, pw = 1) as one
// 8-byte pair per lane -> contiguous 256-byte row segments per half-wave.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct ConvUp2xArgs {
  const float* src;     // [N][C0][HL][WL]   low-resolution tensor (the operand of nn.Upsample)
  const float* wq;      // [C0_pad][4][4][Cout]  pre-summed class filters (pack_up2x_weights_kernel), C0_pad = roundup(C0, CC)
  float* dst;           // [N][Cout][2*HL][2*WL]  partial sums P (no bias / BN / activation)
  int N, C0, Cout, HL, WL;
};

template <int MT_, int WM_, int WN_, int TRL_, int CC_>
struct ConvUp2xCfg {
  static constexpr int MT = MT_, WM = WM_, WN = WN_, TRL = TRL_, CC = CC_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int MB = MT * 32 * WM;
  static constexpr int NTILES = 4 * TRL;                 // (full-res row, column parity) pairs of a 2*TRL x 64 output tile
  static constexpr int NTW = NTILES / WN;                // per wave: whole rows, both parities
  static_assert(NTILES % WN == 0 && NTW % 2 == 0, "a wave owns both column parities of its rows");
  static_assert(CC % 2 == 0, "one MFMA = 2 channels");
  static constexpr int TRp = TRL + 2, TCp = 34, PLANE = TRp * TCp;
  static constexpr int E_IN = CC * PLANE, NIN = (E_IN + NT - 1) / NT, IN_FLOATS = NIN * NT;
  static constexpr int KROWS = CC * 16;                  // (channel, class, tap) rows of one stage
  static constexpr int W_FLOATS = KROWS * MB, E_W4 = W_FLOATS / 4, NW4 = (E_W4 + NT - 1) / NT;
  static_assert(E_W4 % NT == 0, "the filter panel must deal evenly");
  static constexpr int BUF_FLOATS = W_FLOATS + IN_FLOATS;
  static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv_up2x_mfma_kernel(const ConvUp2xArgs a) {
  constexpr int MT = Cfg::MT, WN = Cfg::WN, TRL = Cfg::TRL, CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, NTW = Cfg::NTW;
  constexpr int TCp = Cfg::TCp, PLANE = Cfg::PLANE, NIN = Cfg::NIN, NW4 = Cfg::NW4, KROWS = Cfg::KROWS;
  __shared__ __attribute__((aligned(16))) float lds[2 * Cfg::BUF_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int half = lane >> 5, bl = lane & 31;
  const int HL = a.HL, WL = a.WL, C0 = a.C0, Cout = a.Cout;
  const int tilesH = (HL + TRL - 1) / TRL, tilesW = (WL + 31) / 32;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  int mb, pt;
  if (!conv_block_map(blockIdx.x, nMB, nPT, mb, pt)) return;
  const int n = pt / (tilesH * tilesW);
  const int trem = pt - n * (tilesH * tilesW);
  const int i0 = (trem / tilesW) * TRL, j0 = (trem % tilesW) * 32;
  const int m0 = mb * MB;
  const int HWL = HL * WL;

  // staging slots of the [CC][TRL+2][34] halo tile: offset inside one low-res plane, or -1 (zero padding / unused)
  int so[NIN], sc[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int e = tid + i * NT;
    const int c = e / PLANE, r = e - c * PLANE;
    const int tr = r / TCp, tc = r - tr * TCp;
    const int gi = i0 - 1 + tr, gj = j0 - 1 + tc;
    const bool ok = e < Cfg::E_IN && gi >= 0 && gi < HL && gj >= 0 && gj < WL;
    so[i] = ok ? gi * WL + gj : -1;
    sc[i] = c;
  }
  float rin[NIN];
  f32x4 rw[NW4];
  auto load_stage = [&](int k) {
    const int cbeg = k * CC;
    const float* base = a.src + ((size_t)n * C0 + cbeg) * HWL;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const bool ok = so[i] >= 0 && cbeg + sc[i] < C0;
      rin[i] = base[ok ? sc[i] * HWL + so[i] : 0];
    }
    const float* wsrc = a.wq + (size_t)k * KROWS * Cout + m0;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * NT;
      const int krow = e4 / (MB / 4), m4 = e4 - krow * (MB / 4);
      rw[i] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)krow * Cout + m4 * 4);
    }
  };
  auto store_stage = [&](int buf, int k) {
    float* lw = lds + buf * Cfg::BUF_FLOATS;
    float* li = lw + Cfg::W_FLOATS;
#pragma unroll
    for (int i = 0; i < NW4; ++i) *reinterpret_cast<f32x4*>(lw + (tid + i * NT) * 4) = rw[i];
    const int cbeg = k * CC;
#pragma unroll
    for (int i = 0; i < NIN; ++i) li[tid + i * NT] = (so[i] >= 0 && cbeg + sc[i] < C0) ? rin[i] : 0.0f;
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.0f;

  // N-tile t of the workgroup = (full-res row rf = t >> 1 of the tile, column parity pw = t & 1); wave wn owns
  // t = wn*NTW .. wn*NTW + NTW-1.  Row rf = 2*il + ph reads halo rows il + ph + a and columns bl + pw + b (a, b = tap bits),
  // with the class filter (ph, pw).  Per N-tile that is one wave-uniform offset on each side, the rest are immediates.
  int aoff[NTW], boff[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int t = wn * NTW + j;
    const int rf = t >> 1, pw = t & 1, il = rf >> 1, ph = rf & 1;
    aoff[j] = half * 16 * MB + ((ph * 2 + pw) * 4) * MB + wm * (MT * 32) + bl;
    boff[j] = Cfg::W_FLOATS + half * PLANE + (il + ph) * TCp + pw + bl;
  }

  const int nChunks = (C0 + CC - 1) / CC;
  load_stage(0);
  store_stage(0, 0);
  __syncthreads();
  for (int k = 0; k < nChunks; ++k) {
    const int buf = k & 1;
    if (k + 1 < nChunks) load_stage(k + 1);
    const float* S = lds + buf * Cfg::BUF_FLOATS;
    constexpr int NSTEP = (CC / 2) * 4;                  // (channel pair, tap) steps of this stage
    float av[2][MT][NTW], bv[2][NTW];
    auto read_step = [&](int s, float (&ar)[MT][NTW], float (&br)[NTW]) {
      const int cp = s >> 2, tap = s & 3;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        br[j] = S[boff[j] + (2 * cp) * PLANE + (tap >> 1) * TCp + (tap & 1)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ar[mt][j] = S[aoff[j] + ((2 * cp) * 16 + tap) * MB + mt * 32];
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
          acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][mt][j], bv[s & 1][j], acc[mt][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (MT + 1) * NTW, 0);   // DS reads of the next step
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);         // MFMAs of this step
    }
    if (k + 1 < nChunks) store_stage(buf ^ 1, k + 1);
    __syncthreads();
  }

  // ---- epilogue: the two column parities of a row are interleaved into 8-byte pairs
  const int H = 2 * HL, W = 2 * WL;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wm * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float* dplane = a.dst + ((size_t)n * Cout + co) * H * W;
#pragma unroll
      for (int jp = 0; jp < NTW / 2; ++jp) {
        const int rf = (wn * NTW) / 2 + jp;              // full-res row inside the tile
        const int oh = 2 * i0 + rf, ow = 2 * (j0 + bl);
        if (oh < H && ow < W) {
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          f32x2_t v; v[0] = acc[mt][2 * jp][r]; v[1] = acc[mt][2 * jp + 1][r];
          *reinterpret_cast<f32x2_t*>(dplane + (size_t)oh * W + ow) = v;
        }
      }
    }
  }
}

// Wq[C0_pad][cls][a*2+b][Cout] from the layer's nn.Conv2d weight W[Cout][Cin][3][3] (first C0 input channels), + zero rows.
inline __global__ void pack_up2x_weights_kernel(const float* __restrict__ w, float* __restrict__ wq, int Cout, int Cin, int C0, int C0pad) {
  const long total = (long)C0pad * 16 * Cout;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int co = (int)(e % Cout);
    const long t = e / Cout;
    const int tap = (int)(t & 3), cls = (int)((t >> 2) & 3);
    const int ci = (int)(t >> 4);
    float v = 0.0f;
    if (ci < C0) {
      const int ph = cls >> 1, pw = cls & 1, ta = tap >> 1, tb = tap & 1;
      // rows kh hitting low-res row offset `ta` for output parity ph:  ph=0: {0} | {1,2};  ph=1: {0,1} | {2}
      const int kh0 = ph == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), kh1 = ph == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
      const int kw0 = pw == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), kw1 = pw == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
      const float* wk = w + ((long)co * Cin + ci) * 9;
      for (int kh = kh0; kh <= kh1; ++kh)
        for (int kw = kw0; kw <= kw1; ++kw) v += wk[kh * 3 + kw];
    }
    wq[e] = v;
  }
}



// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of the same half-layer, also at the low resolution.  Transposing
//     P[co][2i+ph][2j+pw] = sum Wq[cls][ci][a][b][co] * Xlow[ci][i+a+ph-1][j+b+pw-1]
// gives        dXlow[ci][u][v] = sum_{co} sum_{dr,dc in {-1,0,1,2}} G[co][dr][dc][ci] * dZ[co][2u+dr][2v+dc]
// with G[co][dr][dc][ci] = Wq[(ph,pw)][ci][a][b][co] for dr = 2 - 2a - ph, dc = 2 - 2b - pw: a 4x4, stride-2 correlation of
// the full-resolution dZ -- 4/9 of the multiply-adds of "3x3 data gradient at full resolution, then sum the 2x2 blocks"
// (autograd of nn.Upsample + Conv2d), and the full-resolution gradient of the upsampled tensor is never written.
// The dZ halo tile is staged de-interleaved by column parity, so the stride-2 B operand is 32 consecutive floats again.
struct DgradUp2xArgs {
  const float* dz;      // [N][Cout][2*HL][2*WL]
  const float* g;       // [Cout_pad][16][C0]   (pack_dgrad_up2x_weights_kernel), Cout_pad = roundup(Cout, CC)
  float* dst;           // [N][C0][HL][WL]
  int N, C0, Cout, HL, WL;
};

template <int MT_, int WM_, int WN_, int NTW_, int CC_>
struct DgradUp2xCfg {
  static constexpr int MT = MT_, WM = WM_, WN = WN_, NTW = NTW_, CC = CC_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int MB = MT * 32 * WM;                // input channels (ci) per workgroup
  static constexpr int TRL = WN * NTW;                   // low-res output rows per workgroup (32 columns wide)
  static constexpr int ROWS = 2 * TRL + 2, HALF = 34, ROWF = 2 * HALF, PLANE = ROWS * ROWF;   // dZ tile: [row][parity][34]
  static constexpr int COLS = 66;                        // full-res columns 2*v0-1 .. 2*v0+64
  static constexpr int E_IN = CC * ROWS * COLS, NIN = (E_IN + NT - 1) / NT;
  static constexpr int IN_FLOATS = CC * PLANE;
  static constexpr int KROWS = CC * 16;
  static constexpr int W_FLOATS = KROWS * MB, E_W4 = W_FLOATS / 4, NW4 = (E_W4 + NT - 1) / NT;
  static_assert(E_W4 % NT == 0, "the filter panel must deal evenly");
  static_assert(CC % 2 == 0, "one MFMA = 2 channels");
  static constexpr int BUF_FLOATS = W_FLOATS + IN_FLOATS;
  static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) dgrad_up2x_mfma_kernel(const DgradUp2xArgs a) {
  constexpr int MT = Cfg::MT, WN = Cfg::WN, NTW = Cfg::NTW, CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TRL = Cfg::TRL;
  constexpr int ROWS = Cfg::ROWS, HALF = Cfg::HALF, ROWF = Cfg::ROWF, PLANE = Cfg::PLANE, COLS = Cfg::COLS;
  constexpr int NIN = Cfg::NIN, NW4 = Cfg::NW4, KROWS = Cfg::KROWS;
  __shared__ __attribute__((aligned(16))) float lds[2 * Cfg::BUF_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int half = lane >> 5, bl = lane & 31;
  const int HL = a.HL, WL = a.WL, C0 = a.C0, Cout = a.Cout;
  const int H = 2 * HL, W = 2 * WL, HW = H * W;
  const int tilesH = (HL + TRL - 1) / TRL, tilesW = (WL + 31) / 32;
  const int nPT = a.N * tilesH * tilesW, nMB = (C0 + MB - 1) / MB;
  int mb, pt;
  if (!conv_block_map(blockIdx.x, nMB, nPT, mb, pt)) return;
  const int n = pt / (tilesH * tilesW);
  const int trem = pt - n * (tilesH * tilesW);
  const int u0 = (trem / tilesW) * TRL, v0 = (trem % tilesW) * 32;
  const int m0 = mb * MB;

  // staging slots: element e of [CC][ROWS][66 full-res columns] -> global offset inside one dZ plane (or -1) and LDS slot
  int so[NIN], sl[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int e = tid + i * NT;
    const int c = e / (ROWS * COLS), r = e - c * (ROWS * COLS);
    const int tr = r / COLS, cc = r - tr * COLS;
    const int gr = 2 * u0 - 1 + tr, gc = 2 * v0 - 1 + cc;
    const bool ok = e < Cfg::E_IN && gr >= 0 && gr < H && gc >= 0 && gc < W;
    so[i] = ok ? gr * W + gc : -1;
    // cc even -> odd full-res column (plane 1), index cc/2;  cc odd -> even column (plane 0), index (cc-1)/2
    sl[i] = e < Cfg::E_IN ? (c * PLANE + tr * ROWF + ((cc & 1) ? 0 : HALF) + (cc >> 1)) | (c << 24) : -1;
  }
  float rin[NIN];
  f32x4 rw[NW4];
  auto load_stage = [&](int k) {
    const int cbeg = k * CC;
    const float* base = a.dz + ((size_t)n * Cout + cbeg) * HW;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int c = sl[i] >> 24;
      const bool ok = so[i] >= 0 && cbeg + c < Cout;
      rin[i] = base[ok ? (size_t)c * HW + so[i] : 0];
    }
    const float* wsrc = a.g + (size_t)k * KROWS * C0 + m0;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * NT;
      const int krow = e4 / (MB / 4), m4 = e4 - krow * (MB / 4);
      const bool ok = m0 + m4 * 4 < C0;                         // C0 % 4 == 0: a 16-byte group is all-in or all-out
      rw[i] = *reinterpret_cast<const f32x4*>(wsrc + (ok ? (size_t)krow * C0 + m4 * 4 : 0));
      if (!ok) rw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_stage = [&](int buf, int k) {
    float* lw = lds + buf * Cfg::BUF_FLOATS;
    float* li = lw + Cfg::W_FLOATS;
#pragma unroll
    for (int i = 0; i < NW4; ++i) *reinterpret_cast<f32x4*>(lw + (tid + i * NT) * 4) = rw[i];
    const int cbeg = k * CC;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      if (sl[i] >= 0) li[sl[i] & 0xFFFFFF] = (so[i] >= 0 && cbeg + (sl[i] >> 24) < Cout) ? rin[i] : 0.0f;
    }
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.0f;

  // tap (dri, dci) = (dr + 1, dc + 1): tile row 2*(u - u0) + dri, plane (dci even -> odd columns: plane 1), index (v - v0) + (dci >> 1)
  const int a_off = half * 16 * MB + wm * (MT * 32) + bl;
  const int b_off = Cfg::W_FLOATS + half * PLANE + bl;

  const int nChunks = (Cout + CC - 1) / CC;
  load_stage(0);
  store_stage(0, 0);
  __syncthreads();
  for (int k = 0; k < nChunks; ++k) {
    const int buf = k & 1;
    if (k + 1 < nChunks) load_stage(k + 1);
    const float* S = lds + buf * Cfg::BUF_FLOATS;
    constexpr int NSTEP = (CC / 2) * 16;
    float av[2][MT], bv[2][NTW];
    auto read_step = [&](int s, float (&ar)[MT], float (&br)[NTW]) {
      const int cp = s >> 4, tap = s & 15;
      const int dri = tap >> 2, dci = tap & 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ar[mt] = S[a_off + ((2 * cp) * 16 + tap) * MB + mt * 32];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int ul = wn * NTW + j;                                 // low-res row of this N-tile inside the workgroup tile
        br[j] = S[b_off + (2 * cp) * PLANE + (2 * ul + dri) * ROWF + ((dci & 1) ? 0 : HALF) + (dci >> 1)];
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
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
    }
    if (k + 1 < nChunks) store_stage(buf ^ 1, k + 1);
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = m0 + wm * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (ci >= C0) continue;
      float* dplane = a.dst + ((size_t)n * C0 + ci) * HL * WL;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int u = u0 + wn * NTW + j, v = v0 + bl;
        if (u < HL && v < WL) dplane[(size_t)u * WL + v] = acc[mt][j][r];
      }
    }
  }
}

// G[Cout_pad][dri*4+dci][C0] from W[Cout][Cin][3][3] (first C0 input channels): dr = dri - 1 = 2 - 2a - ph, dc likewise.
inline __global__ void pack_dgrad_up2x_weights_kernel(const float* __restrict__ w, float* __restrict__ g, int Cout, int Cin, int C0, int CoutPad) {
  const long total = (long)CoutPad * 16 * C0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(e % C0);
    const long t = e / C0;
    const int tap = (int)(t & 15), co = (int)(t >> 4);
    float v = 0.0f;
    if (co < Cout) {
      const int dri = tap >> 2, dci = tap & 3;
      // dri 0..3  <->  (a, ph) = (1,1), (1,0), (0,1), (0,0);  kernel rows kh in R[ph][a]:  R[0] = {{0},{1,2}},  R[1] = {{0,1},{2}}
      const int ta = dri < 2 ? 1 : 0, ph = (dri & 1) ? 0 : 1;
      const int tb = dci < 2 ? 1 : 0, pw = (dci & 1) ? 0 : 1;
      const int kh0 = ph == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), kh1 = ph == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
      const int kw0 = pw == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), kw1 = pw == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
      const float* wk = w + ((long)co * Cin + ci) * 9;
      for (int kh = kh0; kh <= kh1; ++kh)
        for (int kw = kw0; kw <= kw1; ++kw) v += wk[kh * 3 + kw];
    }
    g[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the half-layer at the low resolution:
//     D[co][ci][dr][dc] = sum_{n,u,v} dZ[co][2u+dr][2v+dc] * Xlow[ci][u][v],   dr, dc in {-1, 0, 1, 2}
// (16 taps per low-res pixel instead of the 36 of "3x3 weight gradient over the upsampled tensor"), folded back with
//     dW[co][ci][kh][kw] = sum_{dr in S(kh)} sum_{dc in S(kw)} D[co][ci][dr][dc],   S(0) = {2, 1}, S(1) = {0, 1}, S(2) = {0, -1}.
// dZ is first split into its four parity images Z[pr][pc][..][i][j] = dZ[..][2i+pr][2j+pc] (one HBM pass); against the image
// (pr, pc) the taps dr = 2 - pr - 2*th, dc = 2 - pc - 2*tw (th, tw in {0, 1}) are the 2x2 window (kh0, kw0) = (pr, pc) of an
// ordinary 3x3 weight gradient of (Xlow, Z[pr][pc]) -- wgrad3x3_mfma_kernel<WgradCfg<.., NTAP = 4>>.

// zp[(pr*2+pc)][plane][i][j] = dz[plane][2i+pr][2j+pc];  planes = N*Cout, W % 4 == 0
inline __global__ void __launch_bounds__(256) space_to_depth2_kernel(const float* __restrict__ dz, float* __restrict__ zp, long planes, int H, int W) {
  typedef float s2d_f2 __attribute__((ext_vector_type(2)));
  const int HL = H >> 1, WL = W >> 1, W4 = W >> 2;
  const long total = planes * H * W4;
  const size_t img = (size_t)planes * HL * WL;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int q = (int)(t % W4);
    const long u = t / W4;
    const int h = (int)(u % H);
    const long pl = u / H;
    const f32x4 v = *reinterpret_cast<const f32x4*>(dz + ((size_t)pl * H + h) * W + 4 * q);
    const size_t o = ((size_t)pl * HL + (h >> 1)) * WL + 2 * q;
    s2d_f2 e, d; e[0] = v[0]; e[1] = v[2]; d[0] = v[1]; d[1] = v[3];
    *reinterpret_cast<s2d_f2*>(zp + ((h & 1) * 2 + 0) * img + o) = e;
    *reinterpret_cast<s2d_f2*>(zp + ((h & 1) * 2 + 1) * img + o) = d;
  }
}

// dW[Cout][C0+C1][3][3]: channels < C0 folded from D4[img = pr*2+pc][Cout][C0][th*2+tw], the rest copied from dw_skip[Cout][C1][9]
inline __global__ void __launch_bounds__(256) wgrad_up2x_assemble_kernel(const float* __restrict__ d4, const float* __restrict__ dw_skip,
                                                                  float* __restrict__ dw, int Cout, int C0, int C1) {
  const int Cin = C0 + C1;
  const long total = (long)Cout * Cin * 9;
  const size_t img = (size_t)Cout * C0 * 4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % 9);
    const long t = e / 9;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    float v;
    if (ci >= C0) {
      v = dw_skip[((size_t)co * C1 + (ci - C0)) * 9 + tap];
    } else {
      const int kh = tap / 3, kw = tap - 3 * kh;
      // S(k) as (parity image index p, window tap t) pairs: dr = 2 -> (0,0), 0 -> (0,1), 1 -> (1,0), -1 -> (1,1)
      const int rp[2] = {kh == 0 ? 0 : (kh == 1 ? 0 : 0), kh == 0 ? 1 : (kh == 1 ? 1 : 1)};
      const int rt[2] = {kh == 0 ? 0 : (kh == 1 ? 1 : 1), kh == 0 ? 0 : (kh == 1 ? 0 : 1)};
      const int cp[2] = {0, 1};
      const int ct[2] = {kw == 0 ? 0 : 1, kw == 0 ? 0 : (kw == 1 ? 0 : 1)};
      v = 0.0f;
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          v += d4[(size_t)(rp[i] * 2 + cp[j]) * img + ((size_t)co * C0 + ci) * 4 + rt[i] * 2 + ct[j]];
    }
    dw[e] = v;
  }
}

}  // namespace tnv3
