// conv3x3_wino_mfma.h -- Winograd F(2x2, 3x3) form of the plain 3x3 'same' convolution + eval-mode BN + ReLU
// (model.py:4-16 in eval mode), fused in one kernel on the fp32 matrix cores.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, d = its 4x4 input patch (Lavin & Gray 2015)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// 16 multiply-adds per (co, ci, tile) instead of 36: the contraction becomes 16 independent GEMMs (one per transform
// coefficient xi): M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile].  Same function as the reference in exact
// arithmetic; in fp32 the +-1, 1/2 transforms cost ~3x the rounding noise of the direct chain (6e-7 vs 2e-7 of the
// output scale per layer, measured) -- two orders below the 1e-4 heat-map bar.
//
// Two kernels here (and the third-generation kernel of conv3x3_wino3_mfma.h, variants 3-5) share the tile (64 output channels x 64 tiles = 4 x 64 pixels per workgroup, 8-channel chunks) and the
// packed filter panel:
//   conv3x3_wino_split_mfma_kernel  the production one (tnv3_conv3x3_wino_variant 2): eight waves, the 16 xi split over
//                                   two wave groups (128 accumulators per wave, two waves per SIMD), one barrier per chunk;
//   conv3x3_wino_mfma_kernel        its predecessor (variant 0): four waves, one per SIMD, described first below.
// (Variant 1 -- the patch transform interleaved into the MFMA stream, 20 % slower in round 1 -- was removed in round 2.)
// Mapping (first kernel): MFMA 32x32x2 with M = 32 output channels, N = 32 tiles (2 tile rows x 16 tile columns = 4 x 32 pixels),
// K = 2 input channels; a wave keeps ALL 16 xi of its 32 x 32 tile (256 accumulator registers, one wave per SIMD) so the
// inverse transform happens in registers.  Per chunk of CC channels a workgroup stages the raw halo tile and the
// pre-transformed filter panel U (conv3x3_wino_pack_kernel), transforms the patches cooperatively into the LDS operand
// V[ci][xi][tile] (adds only), then runs CC/2 x 16 MFMAs per wave with both operands read as 32 consecutive floats.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct WinoArgs {
  const float* src;     // [N][Cin][H][W]
  const float* u;       // [Cin_pad][16][Cout]   G g G^T per (ci, co), Cin_pad = roundup(Cin, CC), padding rows zero, then
                        // kPackZeroTail zeros (`zeros` points there: the padding source of the LDS-DMA loader)
  const float* zeros;
  const float* addend;  // optional [N][Cout][H][W] added before the affine (decoder-entry layers) or nullptr
  const float* mean;    // [Cout] or nullptr:  y = (acc - mean) * scale + shift  (eval-mode BN, reference's operation order)
  const float* scale;
  const float* shift;
  float* dst;           // [N][Cout][H][W]
  int N, Cin, Cout, H, W, relu;
  double* stats;        // optional (kernel variant 3 / 4 only) [Cout][nTiles][2]: per output channel and pixel tile (4 x 64 pixels) the sum
                        // and the sum of squares of the values written to dst -- the batch statistics of training-mode BatchNorm taken
                        // from the epilogue's registers (wave butterfly + fixed-order LDS fold: deterministic)
  // Data-gradient launches only (with `stats`): the values written to dst are dA, the gradient at the ACTIVATION of the previous
  // Conv2DBlock; bn_z = that block's raw convolution output [N][Cout][H][W], bn_c4 = its per-channel constants [Cout][4] =
  // (mean, invstd, gamma * invstd as the forward rounds it, beta).  The statistics then are the two sums of BatchNorm + ReLU's
  // backward (model.py:9-10): sum g and sum g * xhat with g = dA * [BN(z) > 0], xhat = (z - mean) * invstd -- the pass
  // bn_relu_bwd_partial_kernel would make over dA and z, taken from the epilogue's registers instead.
  const float* bn_z;
  const float* bn_c4;
  float* pool_dst;      // optional (conv3x3_wino43s_kernel only) [N][Cout][H/2][W/2]: MaxPool2d(2, 2) of the values written to dst (model.py:59,61,63),
                        // taken from the write-out's registers
};

template <int WM_, int WN_, int CC_, int DIAG_ = 0>
struct WinoCfg {
  static constexpr int WM = WM_, WN = WN_, CC = CC_;
  static constexpr int DIAG = DIAG_;   // timing twins (WRONG results): 1 = no DMA after the prologue, 2 = + no patch transform, 3 = + no barriers
  static constexpr int NT = WM * WN * 64;
  static constexpr int MB = 32 * WM;                 // output channels per workgroup
  static constexpr int TB = 32 * WN;                 // 2x2 tiles per workgroup: 2 tile rows x (16*WN) tile columns
  static constexpr int PW = 32 * WN;                 // pixel columns per workgroup (4 pixel rows)
  static constexpr int RW = PW + 8, RAWP = 6 * RW;   // raw halo tile per channel: 6 rows x (PW + 8): columns w0-4 .. w0+PW+3, staged
                                                     // as 16-byte pieces (each fully inside or fully outside the image: W % 4 == 0)
  static constexpr int RAW_FLOATS = CC * RAWP;
  static constexpr int U_FLOATS = CC * 16 * MB, V_FLOATS = CC * 16 * TB;
  static constexpr int NU4 = U_FLOATS / 4 / NT;      // 16-byte filter loads per thread and chunk
  static constexpr int NRAW = (RAW_FLOATS / 4 + NT - 1) / NT;   // 16-byte raw pieces per thread and chunk
  static constexpr int NPAIR = (CC * TB + NT - 1) / NT;   // (channel, tile) patches each thread transforms per chunk
  static_assert((U_FLOATS / 4) % NT == 0, "filter panel must deal evenly");
  static_assert(CC % 2 == 0, "one MFMA = 2 channels");
  // Staging is LDS DMA with a prefetch distance of TWO chunks: at one wave per SIMD (256 accumulator registers) nothing
  // else hides the L2 / HBM latency, and one chunk of MFMAs (CC/2 x 16 x 64 cycles = 1.7 us) is shorter than it.
  // Three filter stages, two raw stages (every wave issues the same number of DMAs per chunk: counted vmcnt), one V.
  static constexpr int RAW_STAGE = NRAW * NT * 4;    // padded so that all NRAW pieces of every wave land inside the stage
  static constexpr int DMA_PER_CHUNK = NU4 + NRAW;
  static constexpr int LDS_FLOATS = 3 * U_FLOATS + V_FLOATS + 2 * RAW_STAGE;
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv3x3_wino_mfma_kernel(const WinoArgs a) {
  constexpr int WN = Cfg::WN, CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TB = Cfg::TB, PW = Cfg::PW, RW = Cfg::RW, RAWP = Cfg::RAWP;
  constexpr int NU4 = Cfg::NU4, NRAW = Cfg::NRAW, NPAIR = Cfg::NPAIR;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* u_s = lds;                                   // three stages
  float* v_s = lds + 3 * Cfg::U_FLOATS;
  float* raw_s = v_s + Cfg::V_FLOATS;                 // two stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = H / 4, tilesW = W / PW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  int mb, pt;
  if (!conv_block_map(blockIdx.x, nMB, nPT, mb, pt)) return;
  const int n = pt / (tilesH * tilesW);
  const int trem = pt - n * (tilesH * tilesW);
  const int h0 = (trem / tilesW) * 4, w0 = (trem % tilesW) * PW;
  const int m0 = mb * MB;

  // raw staging slots: 16-byte piece e of [CC][6][RW/4] -> offset in one input plane or -1 (zero padding / unused slot)
  int so[NRAW];
#pragma unroll
  for (int i = 0; i < NRAW; ++i) {
    const int e = tid + i * NT;
    const int r = e % (RAWP / 4);
    const int tr = r / (RW / 4), q = r - tr * (RW / 4);
    const int gh = h0 - 1 + tr, gw = w0 - 4 + 4 * q;
    so[i] = (e < Cfg::RAW_FLOATS / 4 && gh >= 0 && gh < H && gw >= 0 && gw < W) ? gh * W + gw : -1;
  }
  const float* zsrc = a.zeros + (lane & 15) * 4;
  const int wbase = wave * 64;
  // DMA of chunk k: filter panel -> u_s[ustage], raw halo tile -> raw_s[rstage]; NU4 + NRAW instructions per wave, always
  auto dma_stage = [&](int k, int ustage, int rstage) {
    float* us = u_s + ustage * Cfg::U_FLOATS;
    const float* usrc = a.u + (size_t)k * CC * 16 * Cout + m0;
#pragma unroll
    for (int i = 0; i < NU4; ++i) {
      const int e4 = tid + i * NT;
      const int row = e4 / (MB / 4), m4 = e4 - row * (MB / 4);
      lds_dma16(usrc + (size_t)row * Cout + m4 * 4, us + (i * NT + wbase) * 4);
    }
    float* rs = raw_s + rstage * Cfg::RAW_STAGE;
    const float* base = a.src + ((size_t)n * Cin + k * CC) * HW;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int c = (tid + i * NT) / (RAWP / 4);
      const bool ok = so[i] >= 0 && k * CC + c < Cin;
      lds_dma16(ok ? base + (size_t)c * HW + so[i] : zsrc, rs + (i * NT + wbase) * 4);
    }
  };
  // V[c][xi][t] = (B^T d B)[xi] for the 4x4 patch of tile t = (tr, tc): raw rows 2tr .. 2tr+3, columns 2tc .. 2tc+3
  auto transform = [&](int rstage) {
    const float* rs = raw_s + rstage * Cfg::RAW_STAGE;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int p = tid + i * NT;
      if (p >= CC * TB) break;
      const int c = p / TB, t = p - c * TB;
      const int tr = t / (TB / 2), tc = t - tr * (TB / 2);
      const float* d = rs + c * RAWP + (2 * tr) * RW + 2 * tc + 3;       // tile column 0 = image column w0-1 = staged column 3
      float e[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = d[j], d1 = d[RW + j], d2 = d[2 * RW + j], d3 = d[3 * RW + j];
        e[0][j] = d0 - d2; e[1][j] = d1 + d2; e[2][j] = d2 - d1; e[3][j] = d1 - d3;
      }
      float* v = v_s + (c * 16) * TB + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[(r * 4 + 0) * TB] = e[r][0] - e[r][2];
        v[(r * 4 + 1) * TB] = e[r][1] + e[r][2];
        v[(r * 4 + 2) * TB] = e[r][2] - e[r][1];
        v[(r * 4 + 3) * TB] = e[r][1] - e[r][3];
      }
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;

  const int a_off = half * 16 * MB + wm * 32 + bl;
  const int b_off = half * 16 * TB + wn * 32 + bl;

  const int nChunks = (Cin + CC - 1) / CC;
  auto wait_landed = [&](bool newest_in_flight) {       // everything but (optionally) the newest chunk's DMAs has landed
    if (newest_in_flight) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(Cfg::DMA_PER_CHUNK));
    else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
  };
  // Barriers are raw s_barrier: __syncthreads() would make the compiler drain vmcnt to 0 and with it the DMAs of the chunk
  // that is supposed to stay in flight.  LDS writes of the transform are published with an explicit lgkmcnt(0).
  auto publish_lds = [&]() {
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };
  dma_stage(0, 0, 0);
  if (nChunks > 1) dma_stage(1, 1, 1);
  wait_landed(nChunks > 1);
  __builtin_amdgcn_s_barrier();
  transform(0);
  publish_lds();
  for (int k = 0; k < nChunks; ++k) {
    // stage (k+2)%3 held the filters of chunk k-1 and raw stage k%2 the patches of chunk k: both were released by the barriers below
    if (k + 2 < nChunks && Cfg::DIAG < 1) dma_stage(k + 2, (k + 2) % 3, k & 1);
    const float* A = u_s + (k % 3) * Cfg::U_FLOATS + a_off;
    const float* B = v_s + b_off;
    constexpr int NSTEP = (CC / 2) * 16;                 // (channel pair, xi): one MFMA each
    // One MFMA (64 cycles) per step is shorter than the LDS round trip, so the operand reads run PF steps ahead (ring of
    // PF + 1 register pairs); sched_group_barrier pins "two DS reads, one MFMA" so the compiler keeps that distance.
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int cp = s >> 4, xi = s & 15;
      av[s % RING] = A[(2 * cp * 16 + xi) * MB];
      bv[s % RING] = B[(2 * cp * 16 + xi) * TB];
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
    if (k + 1 < nChunks) {
      if (Cfg::DIAG < 1) wait_landed(k + 2 < nChunks);   // chunk k+1 is in LDS (this wave's pieces) ...
      if (Cfg::DIAG < 3) __builtin_amdgcn_s_barrier();   // ... and everybody's; V and filter stage k%3 are no longer read
      if (Cfg::DIAG < 2) transform((k + 1) & 1);
      if (Cfg::DIAG < 3) publish_lds();
    }
  }

  // ---- inverse transform A^T M A in registers, affine + ReLU, two 8-byte stores per (channel, tile)
  const bool has_affine = a.scale != nullptr;
  const int t = wn * 32 + bl;
  const int tr = t / (TB / 2), tc = t - tr * (TB / 2);
  const int oh = h0 + 2 * tr, ow = w0 + 2 * tc;
  typedef float wf2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    float tt[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float m0j = acc[j][r], m1j = acc[4 + j][r], m2j = acc[8 + j][r], m3j = acc[12 + j][r];
      tt[0][j] = m0j + m1j + m2j;
      tt[1][j] = m1j - m2j - m3j;
    }
    float mu = 0.0f, sc = 1.0f, sh = 0.0f;
    if (has_affine) { mu = a.mean ? a.mean[co] : 0.0f; sc = a.scale[co]; sh = a.shift[co]; }
    const size_t plane = ((size_t)n * Cout + co) * HW;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      wf2 v;
      v[0] = tt[y][0] + tt[y][1] + tt[y][2];
      v[1] = tt[y][1] - tt[y][2] - tt[y][3];
      const size_t o = plane + (size_t)(oh + y) * W + ow;
      if (a.addend) { const wf2 ad = *reinterpret_cast<const wf2*>(a.addend + o); v[0] += ad[0]; v[1] += ad[1]; }
      if (has_affine) { v[0] = (v[0] - mu) * sc + sh; v[1] = (v[1] - mu) * sc + sh; }
      if (a.relu) { v[0] = v[0] > 0.0f ? v[0] : 0.0f; v[1] = v[1] > 0.0f ? v[1] : 0.0f; }
      *reinterpret_cast<wf2*>(a.dst + o) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// xi-split kernel (the production one): the same 64-channel x 64-tile workgroup tile, but EIGHT waves.  Waves 0..3
// ("group 0") accumulate the transform rows xi = 0..7 of their 32 x 32 sub-tile, waves 4..7 ("group 1") the rows
// xi = 8..15: 128 accumulator registers per wave, so two waves share a SIMD.  Within a chunk the groups run in opposite
// order -- group 0 issues the chunk's LDS DMAs, transforms its half of the next chunk's patches and THEN runs its 32
// MFMAs; group 1 runs its MFMAs FIRST and transforms afterwards -- so on every SIMD one wave can feed the matrix pipe
// while the other does the DMA bookkeeping, the adds and the LDS traffic of the patch transform.  The next chunk's V
// goes to a second V stage (nobody waits for the current one to be drained), which leaves ONE barrier per chunk.
// Everything a chunk issues by DMA (filters of chunk k+1, raw tile of chunk k+2) has the whole chunk to land: plain
// vmcnt(0) before the barrier.
// LDS: 2 filter stages + 2 V stages + 2 raw stages = 160 KB.  The two halves of A^T M A meet through LDS at the end:
// group g finishes output row g of every 2 x 2 tile.
// Measured (scripts/wino_diag.py, profiles/r01_wino_split_diag.json): 5-6 % faster than the one-wave-per-SIMD kernel
// above, NOT the 1.4x a perfect overlap would give: its timing twins show that the DMA pieces and the transform still
// cost (nearly) their full time next to another wave's MFMAs, additively, and neither a deeper operand ring nor wave
// priorities change that -- consistent with the fp32 MFMA sharing issue / execution resources with the other
// instruction classes of its SIMD, in which case the lever is the NUMBER of non-MFMA instructions, not their placement.
// Staging through registers instead (global_load_dwordx4 by every thread at the start of a chunk, ds_write_b128 at its
// end; built and measured) is 5 % slower than the LDS DMA: twice the instructions for the same bytes.
template <int CC_, int DIAG_ = 0>
struct WinoSplitCfg {
  static constexpr int WM = 2, WN = 2, CC = CC_;
  static constexpr int DIAG = DIAG_;   // timing twins (WRONG results): 1 = no DMA after the prologue, 2 = + no patch transform, 3 = + no barriers;
                                       // with DMA: 4 = no patch transform, 5 = transform without its V writes, 6 = without its raw reads
  static constexpr int NT = 2 * WM * WN * 64;
  static constexpr int MB = 32 * WM, TB = 32 * WN, PW = 32 * WN;
  static constexpr int RW = PW + 8, RAWP = 6 * RW;
  static constexpr int RAW_FLOATS = CC * RAWP;
  static constexpr int U_FLOATS = CC * 16 * MB, V_FLOATS = CC * 16 * TB;
  static constexpr int NTD = NT / 2;                 // group 0 issues every DMA of a chunk (while group 1 is already in its MFMAs)
  static constexpr int NU4 = U_FLOATS / 4 / NTD;
  static constexpr int NRAW = (RAW_FLOATS / 4 + NTD - 1) / NTD;
  static constexpr int RAW_STAGE = NRAW * NTD * 4;
  static constexpr int LDS_FLOATS = 2 * U_FLOATS + 2 * V_FLOATS + 2 * RAW_STAGE;
  static constexpr int XCH_FLOATS = (NT / 64) * 32 * 64;            // epilogue exchange: 32 floats per lane
  static_assert((U_FLOATS / 4) % NTD == 0, "filter panel must deal evenly");
  static_assert(CC * TB == NT, "one patch per thread and chunk");
  static_assert(XCH_FLOATS <= 2 * U_FLOATS, "the exchange reuses the filter stages");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv3x3_wino_split_mfma_kernel(const WinoArgs a) {
  constexpr int WN = Cfg::WN, CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TB = Cfg::TB, PW = Cfg::PW, RW = Cfg::RW, RAWP = Cfg::RAWP;
  constexpr int NU4 = Cfg::NU4, NRAW = Cfg::NRAW, NTD = Cfg::NTD;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* u_s = lds;                                   // two stages
  float* v_s = lds + 2 * Cfg::U_FLOATS;               // two stages
  float* raw_s = v_s + 2 * Cfg::V_FLOATS;             // two stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wn = wq % WN, wm = wq / WN;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = H / 4, tilesW = W / PW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  int mb, pt;
  if (!conv_block_map(blockIdx.x, nMB, nPT, mb, pt)) return;
  const int n = pt / (tilesH * tilesW);
  const int trem = pt - n * (tilesH * tilesW);
  const int h0 = (trem / tilesW) * 4, w0 = (trem % tilesW) * PW;
  const int m0 = mb * MB;

  int so[NRAW];
#pragma unroll
  for (int i = 0; i < NRAW; ++i) {                    // DMA slots of group 0 (tid < NTD); group 1 never uses them
    const int e = (tid & (NTD - 1)) + i * NTD;
    const int r = e % (RAWP / 4);
    const int tr = r / (RW / 4), q = r - tr * (RW / 4);
    const int gh = h0 - 1 + tr, gw = w0 - 4 + 4 * q;
    so[i] = (e < Cfg::RAW_FLOATS / 4 && gh >= 0 && gh < H && gw >= 0 && gw < W) ? gh * W + gw : -1;
  }
  const float* zsrc = a.zeros + (lane & 15) * 4;
  const int wbase = wave * 64;
  auto dma_u = [&](int k) {
    float* us = u_s + (k & 1) * Cfg::U_FLOATS;
    const float* usrc = a.u + (size_t)k * CC * 16 * Cout + m0;
#pragma unroll
    for (int i = 0; i < NU4; ++i) {
      const int e4 = tid + i * NTD;
      const int row = e4 / (MB / 4), m4 = e4 - row * (MB / 4);
      lds_dma16(usrc + (size_t)row * Cout + m4 * 4, us + (i * NTD + wbase) * 4);
    }
  };
  auto dma_raw = [&](int k) {
    float* rs = raw_s + (k & 1) * Cfg::RAW_STAGE;
    const float* base = a.src + ((size_t)n * Cin + k * CC) * HW;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int c = (tid + i * NTD) / (RAWP / 4);
      const bool ok = so[i] >= 0 && k * CC + c < Cin;
      lds_dma16(ok ? base + (size_t)c * HW + so[i] : zsrc, rs + (i * NTD + wbase) * 4);
    }
  };
  // this thread's patch of a chunk: channel tid / TB, tile tid % TB  (group 0 = channels 0..CC/2-1, group 1 the rest)
  const int pc = tid / TB, ptile = tid - pc * TB;
  const int ptr_ = ptile / (TB / 2), ptc = ptile - ptr_ * (TB / 2);
  const int t_src = pc * RAWP + (2 * ptr_) * RW + 2 * ptc + 3;
  const int t_dst = (pc * 16) * TB + ptile;
  auto transform = [&](int stage) {                     // raw stage -> V stage of the same parity
    const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src;
    float e[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d0, d1, d2, d3;
      if (Cfg::DIAG == 6) { d0 = (float)(tid + j); d1 = (float)(tid - j); d2 = (float)(tid * j); d3 = (float)(stage + j); }
      else { d0 = d[j]; d1 = d[RW + j]; d2 = d[2 * RW + j]; d3 = d[3 * RW + j]; }
      e[0][j] = d0 - d2; e[1][j] = d1 + d2; e[2][j] = d2 - d1; e[3][j] = d1 - d3;
    }
    float* v = v_s + stage * Cfg::V_FLOATS + t_dst;
    if (Cfg::DIAG == 5 && a.N > 0) {                    // keep the reads and adds alive without storing
      float sum = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += (e[r][0] - e[r][2]) * (e[r][1] + e[r][2]) + (e[r][2] - e[r][1]) * (e[r][1] - e[r][3]);
      if (sum == 1234.5678f) v[0] = sum;
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[(r * 4 + 0) * TB] = e[r][0] - e[r][2];
      v[(r * 4 + 1) * TB] = e[r][1] + e[r][2];
      v[(r * 4 + 2) * TB] = e[r][2] - e[r][1];
      v[(r * 4 + 3) * TB] = e[r][1] - e[r][3];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;

  const int a_off = (half * 16 + grp * 8) * MB + wm * 32 + bl;
  const int b_off = (half * 16 + grp * 8) * TB + wn * 32 + bl;
  auto mfma_chunk = [&](int k) {
    const float* A = u_s + (k & 1) * Cfg::U_FLOATS + a_off;
    const float* B = v_s + (k & 1) * Cfg::V_FLOATS + b_off;
    constexpr int NSTEP = (CC / 2) * 8;                  // (channel pair, xi of this group): one MFMA each
    constexpr int PF = 4, RING = PF + 1;                // deeper rings / raised wave priority measured: no effect
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int cp = s >> 3, x = s & 7;
      av[s % RING] = A[(2 * cp * 16 + x) * MB];
      bv[s % RING] = B[(2 * cp * 16 + x) * TB];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], acc[s & 7], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  auto chunk_barrier = [&]() {                          // DMAs landed, V writes visible, everybody done with the old stages
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  const int nChunks = (Cin + CC - 1) / CC;
  // DIAG 7 (libtnv3_diag.so): every wave accumulates, in SGPRs only (no register pressure on the vector side), the s_memtime
  // cycles it spends in each phase of the chunk loop; at the end the wave of the mid-grid workgroup writes its six totals to
  // dst (uint64 [wave][8]: 6 phase totals, chunk count, 0).  The output is garbage by design.
  unsigned long long t_acc0 = 0, t_acc1 = 0, t_acc2 = 0, t_acc3 = 0, t_acc4 = 0, t_acc5 = 0, t_last = 0;
  auto stamp = [&](int slot) {
    if constexpr (Cfg::DIAG == 7) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      const unsigned long long d = now - t_last;
      t_last = now;
      if (slot == 0) t_acc0 += d; else if (slot == 1) t_acc1 += d; else if (slot == 2) t_acc2 += d;
      else if (slot == 3) t_acc3 += d; else if (slot == 4) t_acc4 += d; else if (slot == 5) t_acc5 += d;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (grp == 0) {
    dma_u(0);
    dma_raw(0);
    if (nChunks > 1) dma_raw(1);
  }
  chunk_barrier();
  transform(0);
  chunk_barrier();
  for (int k = 0; k < nChunks; ++k) {
    // free since the barrier that ended chunk k-1: filter stage (k+1)&1 (held chunk k-1) and raw stage k&1 (held chunk k,
    // transformed during chunk k-1)
    constexpr bool kDma = Cfg::DIAG < 1 || Cfg::DIAG > 3, kTransform = Cfg::DIAG < 2 || Cfg::DIAG > 4;
    const bool more = k + 1 < nChunks && kTransform;
    stamp(0);                                              // slot 0: loop overhead since the barrier release
    if (grp == 0) {
      if (k + 1 < nChunks && kDma) dma_u(k + 1);
      if (k + 2 < nChunks && kDma) dma_raw(k + 2);
      stamp(1);                                            // grp 0: DMA issue
      if (more) transform((k + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      stamp(2);                                            // grp 0: patch transform
      mfma_chunk(k);
      stamp(3);                                            // grp 0: MFMAs
    } else {
      mfma_chunk(k);
      __builtin_amdgcn_sched_barrier(0);
      stamp(1);                                            // grp 1: MFMAs
      if (more) transform((k + 1) & 1);
      stamp(2);                                            // grp 1: patch transform
    }
    if (Cfg::DIAG == 7) {
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
      stamp(4);                                            // own DMAs landed, own LDS writes done
      __builtin_amdgcn_s_barrier();
      stamp(5);                                            // waiting for the other waves
    } else if (Cfg::DIAG != 3) chunk_barrier();
  }
  if constexpr (Cfg::DIAG == 7) {
    float keep = 0.0f;                                     // the accumulators must stay live or the MFMAs are dead code
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) keep += acc[x][r];
    if (blockIdx.x == gridDim.x / 2) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dst) + wave * 8;
      if (lane == 0) { o[0] = t_acc0; o[1] = t_acc1; o[2] = t_acc2; o[3] = t_acc3; o[4] = t_acc4; o[5] = t_acc5; o[6] = (unsigned long long)nChunks; }
      if (keep == 1234.5678f) o[7] = 1;
    }
    return;
  }

  // ---- inverse transform: rows of A^T M (this group's two xi rows), columns, then the halves meet through LDS
  float pv[16][2][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float tt[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = acc[j][r], hi = acc[4 + j][r];    // M rows 2*grp and 2*grp + 1
      tt[0][j] = grp ? lo : lo + hi;                     // A^T row 0 = [1 1 1 0]
      tt[1][j] = grp ? -lo - hi : hi;                    // A^T row 1 = [0 1 -1 -1]
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      pv[r][y][0] = tt[y][0] + tt[y][1] + tt[y][2];
      pv[r][y][1] = tt[y][1] - tt[y][2] - tt[y][3];
    }
  }
  float* xch = lds;                                      // all stages are free after the last chunk barrier
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int x = 0; x < 2; ++x) xch[(wave * 32 + r * 2 + x) * 64 + lane] = grp ? pv[r][0][x] : pv[r][1][x];
  __syncthreads();
  const bool has_affine = a.scale != nullptr;
  const int t = wn * 32 + bl;
  const int tr = t / (TB / 2), tc = t - tr * (TB / 2);
  const int oh = h0 + 2 * tr + grp, ow = w0 + 2 * tc;    // group g finishes output row g of the tile
  typedef float wf2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    wf2 v;
    v[0] = (grp ? pv[r][1][0] : pv[r][0][0]) + xch[((wave ^ 4) * 32 + r * 2 + 0) * 64 + lane];
    v[1] = (grp ? pv[r][1][1] : pv[r][0][1]) + xch[((wave ^ 4) * 32 + r * 2 + 1) * 64 + lane];
    float mu = 0.0f, sc = 1.0f, sh = 0.0f;
    if (has_affine) { mu = a.mean ? a.mean[co] : 0.0f; sc = a.scale[co]; sh = a.shift[co]; }
    const size_t o = ((size_t)n * Cout + co) * HW + (size_t)oh * W + ow;
    if (a.addend) { const wf2 ad = *reinterpret_cast<const wf2*>(a.addend + o); v[0] += ad[0]; v[1] += ad[1]; }
    if (has_affine) { v[0] = (v[0] - mu) * sc + sh; v[1] = (v[1] - mu) * sc + sh; }
    if (a.relu) { v[0] = v[0] > 0.0f ? v[0] : 0.0f; v[1] = v[1] > 0.0f ? v[1] : 0.0f; }
    *reinterpret_cast<wf2*>(a.dst + o) = v;
  }
}

// U[Cin_pad][xi = i*4+j][Cout] = (G g G^T)[i][j] from W[Cout][Cin][3][3]; rows ci >= Cin are zero
// The filter of logical (co, ci) starts at w + co * s_co + ci * s_ci; flip reads its taps back to front (the data
// gradient's filter w'[ci][co][kh][kw] = w[co][ci][2-kh][2-kw] is s_co = 9, s_ci = Cin_w * 9, flip = 1 on the same tensor).
// layout 0: U[ci][xi][Cout];  layout 1 ("quad", conv3x3_wino3_mfma.h): U[ci / 2][xi / 4][ci % 2][Cout][xi % 4] (CinPad even);
// layout 2 (conv3x3_wino6_mfma.h: the A operand in the order its lanes load it, Cout % 32 == 0, CinPad % 8 == 0):
//   U[co / 32][ci / 8][xi / 8][q = (ci % 8 / 2) * 2 + (xi % 8) / 4][lane = (ci % 2) * 32 + co % 32][xi % 4]
// (elements e0, e0 + stride, ... of one panel; shared by the one-panel kernel and the table-driven one)
__device__ __forceinline__ void conv3x3_wino_pack_elements(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int CinPad,
                                                           long s_co, long s_ci, int flip, int layout, long e0, long stride) {
  const long body = (long)CinPad * 16 * Cout, total = body + kPackZeroTail;
  for (long e = e0; e < total; e += stride) {
    if (e >= body) { u[e] = 0.0f; continue; }
    int co, xi, ci;
    if (layout == 2) {
      const int j = (int)(e & 3), ln = (int)((e >> 2) & 63), q = (int)((e >> 8) & 7), g = (int)((e >> 11) & 1);
      const long rest = e >> 12;                          // mb32 * (CinPad / 8) + chunk
      const int nch = CinPad / 8;
      const int k = (int)(rest % nch), mb32 = (int)(rest / nch);
      ci = 8 * k + 2 * (q >> 1) + (ln >> 5);
      xi = 8 * g + 4 * (q & 1) + j;
      co = 32 * mb32 + (ln & 31);
    } else if (layout == 1) {
      const int x = (int)(e & 3);
      co = (int)((e >> 2) % Cout);
      const long t = (e >> 2) / Cout;                     // ((pair * 4 + row) * 2 + parity)
      xi = (int)((t >> 1) & 3) * 4 + x;
      ci = (int)(t >> 3) * 2 + (int)(t & 1);
    } else {
      co = (int)(e % Cout);
      const long t = e / Cout;
      xi = (int)(t & 15);
      ci = (int)(t >> 4);
    }
    float v = 0.0f;
    if (ci < Cin) {
      const float* g = w + (long)co * s_co + (long)ci * s_ci;
      const int i = xi >> 2, j = xi & 3;
      // row i of G applied to the filter rows, then row j of G to the columns
      float rowv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float g0 = flip ? g[8 - c] : g[c], g1 = flip ? g[5 - c] : g[3 + c], g2 = flip ? g[2 - c] : g[6 + c];
        rowv[c] = i == 0 ? g0 : (i == 1 ? 0.5f * (g0 + g1 + g2) : (i == 2 ? 0.5f * (g0 - g1 + g2) : g2));
      }
      v = j == 0 ? rowv[0] : (j == 1 ? 0.5f * (rowv[0] + rowv[1] + rowv[2]) : (j == 2 ? 0.5f * (rowv[0] - rowv[1] + rowv[2]) : rowv[2]));
    }
    u[e] = v;
  }
}


inline __global__ void conv3x3_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int CinPad, long s_co,
                                         long s_ci, int flip, int layout) {
  conv3x3_wino_pack_elements(w, u, Cout, Cin, CinPad, s_co, s_ci, flip, layout, (long)blockIdx.x * blockDim.x + threadIdx.x,
                             (long)gridDim.x * blockDim.x);
}

// Every stale filter panel of a training step in ONE launch (tnv3_conv3x3_wino_pack_multi): the step re-packs ~33 panels after each
// optimiser step -- forward and data-gradient filters of every layer -- as 14-us launches strung along the main stream; the table
// (kernel argument, by value) gives each panel a run of blocks.  Same arithmetic per element: the same bits as the one-panel kernel.
constexpr int kWinoPackMaxItems = 40;
struct WinoPackTable {
  const float* w[kWinoPackMaxItems];
  float* u[kWinoPackMaxItems];
  long s_co[kWinoPackMaxItems], s_ci[kWinoPackMaxItems];
  int cout[kWinoPackMaxItems], cin[kWinoPackMaxItems], cpad[kWinoPackMaxItems], flip[kWinoPackMaxItems], layout[kWinoPackMaxItems];
  int first_block[kWinoPackMaxItems + 1];      // prefix sum of the panels' block counts
  int count;
};
inline __global__ void __launch_bounds__(256) conv3x3_wino_pack_multi_kernel(const WinoPackTable t) {
  int lo = 0, hi = t.count;                    // first_block[lo] <= blockIdx.x < first_block[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const int k = lo;
  const long nb = t.first_block[k + 1] - t.first_block[k];
  conv3x3_wino_pack_elements(t.w[k], t.u[k], t.cout[k], t.cin[k], t.cpad[k], t.s_co[k], t.s_ci[k], t.flip[k], t.layout[k],
                             (long)((int)blockIdx.x - t.first_block[k]) * 256 + threadIdx.x, nb * 256);
}

}  // namespace tnv3
