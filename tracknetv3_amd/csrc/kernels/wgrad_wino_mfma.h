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

#include <type_traits>

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

#ifdef TNV3_DIAG      // the first kernel (variant 0): a measurement twin of libtnv3_diag.so since ABI 5 -- a non-template kernel is emitted wherever this
                      // header is included, so the definition itself is guarded
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
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Second generation (variant 1 of tnv3_conv3x3_wgrad_wino): the same GEMMs, the same K order per accumulator -- hence the
// same bits -- restructured the way the forward kernel was (conv3x3_wino_mfma.h -> conv3x3_wino3_mfma.h):
//   * 512 threads, two waves per SIMD: wave group g keeps transform rows 2g, 2g+1 (8 of the 16 xi, 128 accumulator registers)
//     of the 64 x 64 block and PRODUCES those rows of both operands itself, so the groups share nothing but the raw strips.
//     They run half a period apart -- while one group's waves stream their 32 MFMAs of a chunk, the other group transforms
//     its next chunk -- instead of the whole workgroup alternating between a transform phase (matrix pipe idle) and an MFMA
//     phase: the first kernel spent 3.6 LDS instructions per MFMA serially with the MFMAs and reached ~35 % of the fp32 peak.
//   * a thread transforms a horizontal tile PAIR: dZ rows as two 16-byte reads, X rows as 2 x 16 bytes + 4 (the forward
//     kernel's paired patch transform) instead of 20 scalar reads per patch; results leave as 8-byte pairs.
//   * LDS-DMA through buffer descriptors: per-lane offsets are chunk-invariant, the strip position is one scalar add, image
//     borders are the hardware's out-of-range zeros (no per-piece address arithmetic, no zero page), and the chunk walk
//     (image, tile row, 16-pixel segment) advances by adds and carries instead of two divisions per chunk.
struct WgradWino2Cfg {
  static constexpr int NT = 512, TCH = 8, TS = TCH + 1;
  static constexpr int OP_FLOATS = 16 * 64 * TS;                        // one transformed operand: [xi][channel][tile]
  static constexpr int DZ_RAW = 64 * 2 * 16;                            // [co][2 rows][16 px]
  static constexpr int XW = 24, X_RAW = 64 * 4 * XW;                    // [ci][4 rows][24 px: columns 16j-4 .. 16j+19]
  static constexpr int RAW_STAGE = DZ_RAW + X_RAW;
  static constexpr int NX = X_RAW / 4 / NT;                             // x pieces per thread and chunk (3); one dZ piece
  static constexpr int LDS_FLOATS = 2 * OP_FLOATS + 2 * RAW_STAGE;
  static_assert(DZ_RAW / 4 == NT && X_RAW % (4 * NT) == 0, "pieces must deal evenly");
};

// SPLIT (variant 7): as in kernel 3's variant 4, the group in its MFMA phase also transforms Yh for the OTHER group inside its own
// MFMA stream and the transforming group does V only -- here with kernel 2's 4-byte operand ring, i.e. inside its footprint (the
// registers and LDS it leaves are what lets the other stream's BatchNorm passes run beside it, §3.1f of DESIGN.md).
template <int SPLIT>
__global__ void __launch_bounds__(WgradWino2Cfg::NT) wgrad_wino2_mfma_kernel(const WgradWinoArgs a) {
  using Cfg = WgradWino2Cfg;
  constexpr int NT = Cfg::NT, TS = Cfg::TS, XW = Cfg::XW, NX = Cfg::NX;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;
  float* v_s = lds + Cfg::OP_FLOATS;
  float* raw_s = lds + 2 * Cfg::OP_FLOATS;                 // two stages of [dz strip | x strip]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wn = wq & 1, wm = wq >> 1;                      // wm: co half, wn: ci half
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int nCB = Cin / 64;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * 64, ci0 = cb * 64;
  const int segW = W / 16, rowsT = H / 2;
  const int nChunksAll = a.N * rowsT * segW;
  const int nMine = nChunksAll > ks ? (nChunksAll - ks + a.splitK - 1) / a.splitK : 0;      // chunks ks, ks + splitK, ...

  // ---- the DMA cursor: (image, tile row, segment) of the next chunk to fetch; one step = splitK chunks
  const int per_img = rowsT * segW;
  int c_n = ks / per_img, c_i = (ks - c_n * per_img) / segW, c_j = ks - c_n * per_img - c_i * segW;
  const int d_n = a.splitK / per_img, d_i = (a.splitK - d_n * per_img) / segW, d_j = a.splitK - d_n * per_img - d_i * segW;

  // ---- chunk-invariant per-lane byte offsets, relative to the strip origin (dZ: row 2i, column 16j; X: row 2i-1, column 16j-4)
  unsigned vo_dz, vo_x[NX], edge[NX];                      // edge bits: piece lies in strip row 0 / row 3 / column piece 0 / column piece 5
  {
    const int q4 = tid & 3, r = (tid >> 2) & 1, co = tid >> 3;            // dZ piece of [co][2 rows][4 pieces]
    vo_dz = (unsigned)(co * HW + r * W + 4 * q4) * 4u;
#pragma unroll
    for (int p = 0; p < NX; ++p) {
      const int e = tid + p * NT;                                          // X piece of [ci][4 rows][6 pieces]
      const int q6 = e % 6, t2 = e / 6;
      const int r4 = t2 & 3, ci = t2 >> 2;
      vo_x[p] = (unsigned)(ci * HW + r4 * W + 4 * q6) * 4u;
      edge[p] = (r4 == 0 ? 1u : 0u) | (r4 == 3 ? 2u : 0u) | (q6 == 0 ? 4u : 0u) | (q6 == 5 ? 8u : 0u);
    }
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);             // scalar: the LDS-DMA destinations (M0) stay on the SALU
  const unsigned planes_dz = 64u * (unsigned)HW * 4u;                      // 64 channel planes of one image (< 2^31: host check)
  auto dma_chunk = [&](int stage) {                                        // the cursor's chunk -> raw stage; then the cursor advances
    const tnv3_rsrc_t r_dz = tnv3_make_rsrc(a.dz + ((size_t)c_n * Cout + co0) * HW, planes_dz);
    const tnv3_rsrc_t r_x = tnv3_make_rsrc(a.x + ((size_t)c_n * Cin + ci0) * HW, planes_dz);
    const int off_dz = (2 * c_i * W + 16 * c_j) * 4;
    const int off_x = ((2 * c_i - 1) * W + 16 * c_j - 4) * 4;            // negative at the top-left corner: only out-of-image pieces
    const unsigned border = (c_i == 0 ? 1u : 0u) | (c_i == rowsT - 1 ? 2u : 0u) | (c_j == 0 ? 4u : 0u) | (c_j == segW - 1 ? 8u : 0u);
    float* rs = raw_s + stage * Cfg::RAW_STAGE;
    tnv3_buf_dma16(r_dz, rs + wbase * 4, vo_dz + (unsigned)off_dz);
#pragma unroll
    for (int p = 0; p < NX; ++p)
      tnv3_buf_dma16(r_x, rs + Cfg::DZ_RAW + (p * NT + wbase) * 4, (edge[p] & border) ? kDmaOob : vo_x[p] + (unsigned)off_x);
    c_j += d_j; if (c_j >= segW) { c_j -= segW; ++c_i; }
    c_i += d_i; if (c_i >= rowsT) { c_i -= rowsT; ++c_n; }
    c_n += d_n;
  };

  // ---- transform of a tile pair: thread (channel ch, pair tp) of its group -> rows 2*grp, 2*grp+1 of Yh and V, tiles 2tp, 2tp+1
  const int tg = tid & 255, ch = tg >> 2, tp = tg & 3;
  auto yh_transform = [&](int stage, auto gc) {
    constexpr int G = decltype(gc)::value;
    const float* rs = raw_s + stage * Cfg::RAW_STAGE;
    {   // Yh = A dY A^T,  A = [1 0; 1 1; 1 -1; 0 -1]: rows (y0, y0 + y1 | y0 - y1, -y1), the same along the columns
      const f32x4 ya = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 4 * tp);          // dZ row 2i,   columns 4tp .. 4tp+3
      const f32x4 yb = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 16 + 4 * tp);     // dZ row 2i+1
      float* o = yh_s + (G * 8) * 64 * TS + ch * TS + 2 * tp;
#pragma unroll
      for (int i = 0; i < 2; ++i) {                                                      // transform row 2G + i
        float rr[2][2];                                                                  // [tile][column of the 2x2 block]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float y00 = ya[2 * t], y01 = ya[2 * t + 1], y10 = yb[2 * t], y11 = yb[2 * t + 1];
          if (G == 0) { rr[t][0] = i == 0 ? y00 : y00 + y10; rr[t][1] = i == 0 ? y01 : y01 + y11; }
          else { rr[t][0] = i == 0 ? y00 - y10 : -y10; rr[t][1] = i == 0 ? y01 - y11 : -y11; }
        }
        float* oi = o + (i * 4) * 64 * TS;
        oi[0 * 64 * TS] = rr[0][0];            oi[0 * 64 * TS + 1] = rr[1][0];
        oi[1 * 64 * TS] = rr[0][0] + rr[0][1]; oi[1 * 64 * TS + 1] = rr[1][0] + rr[1][1];
        oi[2 * 64 * TS] = rr[0][0] - rr[0][1]; oi[2 * 64 * TS + 1] = rr[1][0] - rr[1][1];
        oi[3 * 64 * TS] = -rr[0][1];           oi[3 * 64 * TS + 1] = -rr[1][1];
      }
    }
  };
  auto transform = [&](int stage, auto gc, bool with_yh) {          // with_yh: a compile-time constant at every call site
    constexpr int G = decltype(gc)::value;
    const float* rs = raw_s + stage * Cfg::RAW_STAGE;
    if (with_yh) yh_transform(stage, gc);
    {   // V = B^T d B: strip rows G .. G+2 (patch rows d0,d1,d2 | d1,d2,d3), patch columns 4tp+3 .. 4tp+8 of the 24-float strip row
      const float* d = rs + Cfg::DZ_RAW + ch * (4 * XW) + G * XW + 4 * tp;
      float x[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * XW);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * XW + 4);
        const float q2 = d[r * XW + 8];
        x[r][0] = q0[3]; x[r][1] = q1[0]; x[r][2] = q1[1]; x[r][3] = q1[2]; x[r][4] = q1[3]; x[r][5] = q2;
      }
      float e[2][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        e[0][j] = G ? x[1][j] - x[0][j] : x[0][j] - x[2][j];      // d2 - d1      | d0 - d2
        e[1][j] = G ? x[0][j] - x[2][j] : x[1][j] + x[2][j];      // d1 - d3      | d1 + d2
      }
      float* o = v_s + (G * 8) * 64 * TS + ch * TS + 2 * tp;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float* orow = o + (r * 4) * 64 * TS;
        orow[0 * 64 * TS] = e[r][0] - e[r][2]; orow[0 * 64 * TS + 1] = e[r][2] - e[r][4];
        orow[1 * 64 * TS] = e[r][1] + e[r][2]; orow[1 * 64 * TS + 1] = e[r][3] + e[r][4];
        orow[2 * 64 * TS] = e[r][2] - e[r][1]; orow[2 * 64 * TS + 1] = e[r][4] - e[r][3];
        orow[3 * 64 * TS] = e[r][1] - e[r][3]; orow[3 * 64 * TS + 1] = e[r][3] - e[r][5];
      }
    }
  };
  const int sgrp = __builtin_amdgcn_readfirstlane(grp);                    // wave-uniform: scalar branches, no per-element selects

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  const float* A = yh_s + (grp * 8) * 64 * TS + (wm * 32 + bl) * TS + half;
  const float* B = v_s + (grp * 8) * 64 * TS + (wn * 32 + bl) * TS + half;
  // 4 tile pairs x this group's 8 xi; per accumulator the K order of kernel 1.  SPLIT: behind the first 12 steps, the Yh transform of
  // the other group's rows from raw stage `yh_stage` (after the last chunk: of a stale strip, into rows nobody reads -- unconditional,
  // so that the MFMA stream stays straight-line code)
  auto mfma_chunk = [&](int yh_stage, auto gother) {
    constexpr int NSTEP = 4 * 8;
    constexpr int PF = 4, RING = PF + 1;
    constexpr int CUT = SPLIT ? 12 : NSTEP;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int t2 = s >> 3, xi = s & 7;
      av[s % RING] = A[xi * 64 * TS + 2 * t2];
      bv[s % RING] = B[xi * 64 * TS + 2 * t2];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < CUT; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], acc[s & 7], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    if constexpr (SPLIT) {
      __builtin_amdgcn_sched_barrier(0);
      yh_transform(yh_stage, gother);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = CUT; s < NSTEP; ++s) {
        if (s + PF < NSTEP) read_step(s + PF);
        acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], acc[s & 7], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
  };
  auto phase_end = [&](bool dma_too) {                      // own LDS traffic (and, at the end of an odd phase, own DMAs) done; everybody
    if (dma_too) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // Phases (one workgroup barrier each); group 0 runs half a period ahead of group 1:
  //   fill      DMA(0), DMA(1);  group 0: T(0)
  //   odd  q    group 0: MFMA(q)        group 1: T(q)          -- ends with the DMAs issued one phase earlier landed
  //   even q    all: issue DMA(q+2) into stage q & 1 (both groups are done with the strips of chunk q)
  //             group 0: T(q+1)         group 1: MFMA(q)
  if (nMine > 0) {
    dma_chunk(0);
    if (nMine > 1) dma_chunk(1);
  }
  phase_end(true);
  // (one loop per group, each with a single MFMA site: the accumulators stay in place; both execute the same barriers)
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  if (sgrp == 0) {
    if (nMine > 0) transform(0, G0{}, true);
    phase_end(false);
    for (int q = 0; q < nMine; ++q) {
      mfma_chunk(q & 1, G1{});                              // SPLIT: + Yh rows 2, 3 of chunk q
      phase_end(true);
      if (q + 2 < nMine) dma_chunk(q & 1);
      if (q + 1 < nMine) transform((q + 1) & 1, G0{}, !SPLIT);
      phase_end(false);
    }
  } else {
    phase_end(false);
    for (int q = 0; q < nMine; ++q) {
      transform(q & 1, G1{}, !SPLIT);
      phase_end(true);
      if (q + 2 < nMine) dma_chunk(q & 1);
      mfma_chunk((q + 1) & 1, G0{});                        // SPLIT: + Yh rows 0, 1 of chunk q+1
      phase_end(false);
    }
  }

  // partial slab: part[ks][xi][co][ci]
  float* slab = a.part + (size_t)ks * 16 * Cout * Cin;
  const int ci = ci0 + wn * 32 + bl;
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      slab[((size_t)(grp * 8 + x) * Cout + co) * Cin + ci] = acc[x][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Third generation (variants 2 and 3 of tnv3_conv3x3_wgrad_wino): the schedule of kernel 2 -- two wave groups half a period
// apart, each producing and consuming its own transform rows -- and the same K order per accumulator (the same bits), with the
// two things kernel 2's counters and ISA pointed at:
//   * operands as 16-byte reads.  A transformed operand row is [xi][64 channels][8 tile positions] with the chunk's tiles split
//     by parity -- positions 0..3 hold tiles 0, 2, 4, 6, positions 4..7 tiles 1, 3, 5, 7 -- so lane (channel, k = half) fetches
//     the k-th tile of all four K steps of a chunk with ONE ds_read_b128: 16 reads per wave and chunk instead of 64 ds_read_b32
//     whose three-deep register ring (two MFMAs of lookahead) was shorter than the LDS latency under the other group's
//     transform traffic.  The channel's two 16-byte blocks sit at block (2 c + b + g) mod 16 of its 8-channel group (c = channel
//     in the group, g = parity of the group): reads by 16 consecutive channels and the transform's 4-byte writes both touch
//     every bank exactly once, with no padding (the stride-9 rows of kernel 2 needed none either but could only be read 4 bytes
//     at a time).
//   * NST raw stages instead of two (the compact rows free 8 KB; NST = 3 fills the 160 KB exactly): a strip is fetched
//     NST - 1 chunks ahead.  On the 64-channel layers every strip is an HBM miss (one (co, ci) block: nothing is shared between
//     workgroups), and two stages gave the DMA two phases (~1.7 us at the MFMA rate) to land.
template <int NST_ = 3, int DIAG_ = 0, int SPLIT_ = 0>
struct WgradWino3Cfg {
  static constexpr int NST = NST_;
  static constexpr int DIAG = DIAG_;                                    // timing twins (WRONG results; libtnv3_diag.so): 1 no transform, 2 no DMA, 3 no MFMAs
  // SPLIT (variant 4): the transform work is dealt by what it costs where.  Beside the partner's MFMA stream a wave issues ~4 vector
  // or one 16-byte LDS instruction per MFMA (r02_mfma_f32_coissue.json), and an LDS-DMA piece takes two MFMA times, so a group's
  // whole transform (60 vector + 10 LDS-read instructions + 4 DMA pieces) took ~3800 cycles beside 2100 cycles of MFMAs.  With SPLIT the
  // group in its MFMA phase also transforms Yh FOR THE OTHER GROUP (2 raw reads, ~16 adds, 16 stores inside its own stream at ~5
  // cycles each) and issues no DMA; the transforming group does V only and issues its half of a strip's DMA pieces in its own
  // transform phase (group 0 in the even phase as before, group 1 one phase later: three raw stages needed).
  static constexpr int SPLIT = SPLIT_;
  static constexpr int NT = 512, TCH = 8;
  static constexpr int XS = 64 * TCH;                                   // one xi row of a transformed operand
  static constexpr int OP_FLOATS = 16 * XS;
  static constexpr int DZ_RAW = 64 * 2 * 16;                            // [co][2 rows][16 px]
  static constexpr int XW = 24, X_RAW = 64 * 4 * XW;                    // [ci][4 rows][24 px: columns 16j-4 .. 16j+19]
  static constexpr int RAW_STAGE = DZ_RAW + X_RAW;
  static constexpr int NX = X_RAW / 4 / NT;                             // x pieces per thread and chunk (3); one dZ piece
  static constexpr int DMA_PER_CHUNK = 1 + NX;
  static constexpr int LDS_FLOATS = 2 * OP_FLOATS + NST * RAW_STAGE;
  static_assert(DZ_RAW / 4 == NT && X_RAW % (4 * NT) == 0, "pieces must deal evenly");
  static_assert(NST >= 2 && NST <= 3 && LDS_FLOATS * 4 <= 160 * 1024, "LDS budget (and the counted waits below know two or three stages)");
  static_assert(!SPLIT || NST == 3, "the split schedule issues group 1's DMA pieces a phase later: three raw stages");
};

// float offset of 16-byte block `blk` (0: even tiles, 1: odd tiles) of channel `ch` inside an xi row
__device__ __forceinline__ int wgrad_wino3_op_off(int ch, int blk) {
  const int g = ch >> 3;
  return g * 64 + ((2 * (ch & 7) + (g & 1) + blk) & 15) * 4;
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) wgrad_wino3_mfma_kernel(const WgradWinoArgs a) {
  constexpr int NT = Cfg::NT, XS = Cfg::XS, XW = Cfg::XW, NX = Cfg::NX, NST = Cfg::NST;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;
  float* v_s = lds + Cfg::OP_FLOATS;
  float* raw_s = lds + 2 * Cfg::OP_FLOATS;                 // NST stages of [dz strip | x strip]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wn = wq & 1, wm = wq >> 1;                      // wm: co half, wn: ci half
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int nCB = Cin / 64;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * 64, ci0 = cb * 64;
  const int segW = W / 16, rowsT = H / 2;
  const int nChunksAll = a.N * rowsT * segW;
  const int nMine = nChunksAll > ks ? (nChunksAll - ks + a.splitK - 1) / a.splitK : 0;      // chunks ks, ks + splitK, ...

  // ---- the DMA cursor: (image, tile row, segment) of the next chunk to fetch; one step = splitK chunks
  const int per_img = rowsT * segW;
  int c_n = ks / per_img, c_i = (ks - c_n * per_img) / segW, c_j = ks - c_n * per_img - c_i * segW;
  const int d_n = a.splitK / per_img, d_i = (a.splitK - d_n * per_img) / segW, d_j = a.splitK - d_n * per_img - d_i * segW;

  // ---- chunk-invariant per-lane byte offsets, relative to the strip origin (dZ: row 2i, column 16j; X: row 2i-1, column 16j-4)
  unsigned vo_dz, vo_x[NX], edge[NX];                      // edge bits: piece lies in strip row 0 / row 3 / column piece 0 / column piece 5
  {
    const int q4 = tid & 3, r = (tid >> 2) & 1, co = tid >> 3;            // dZ piece of [co][2 rows][4 pieces]
    vo_dz = (unsigned)(co * HW + r * W + 4 * q4) * 4u;
#pragma unroll
    for (int p = 0; p < NX; ++p) {
      const int e = tid + p * NT;                                          // X piece of [ci][4 rows][6 pieces]
      const int q6 = e % 6, t2 = e / 6;
      const int r4 = t2 & 3, ci = t2 >> 2;
      vo_x[p] = (unsigned)(ci * HW + r4 * W + 4 * q6) * 4u;
      edge[p] = (r4 == 0 ? 1u : 0u) | (r4 == 3 ? 2u : 0u) | (q6 == 0 ? 4u : 0u) | (q6 == 5 ? 8u : 0u);
    }
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);             // scalar: the LDS-DMA destinations (M0) stay on the SALU
  const unsigned planes_dz = 64u * (unsigned)HW * 4u;                      // 64 channel planes of one image (< 2^31: host check)
  auto dma_chunk = [&](int stage) {                                        // the cursor's chunk -> raw stage; then the cursor advances
    if constexpr (Cfg::DIAG != 2) {
      const tnv3_rsrc_t r_dz = tnv3_make_rsrc(a.dz + ((size_t)c_n * Cout + co0) * HW, planes_dz);
      const tnv3_rsrc_t r_x = tnv3_make_rsrc(a.x + ((size_t)c_n * Cin + ci0) * HW, planes_dz);
      const int off_dz = (2 * c_i * W + 16 * c_j) * 4;
      const int off_x = ((2 * c_i - 1) * W + 16 * c_j - 4) * 4;          // negative at the top-left corner: only out-of-image pieces
      const unsigned border = (c_i == 0 ? 1u : 0u) | (c_i == rowsT - 1 ? 2u : 0u) | (c_j == 0 ? 4u : 0u) | (c_j == segW - 1 ? 8u : 0u);
      float* rs = raw_s + stage * Cfg::RAW_STAGE;
      tnv3_buf_dma16(r_dz, rs + wbase * 4, vo_dz + (unsigned)off_dz);
#pragma unroll
      for (int p = 0; p < NX; ++p)
        tnv3_buf_dma16(r_x, rs + Cfg::DZ_RAW + (p * NT + wbase) * 4, (edge[p] & border) ? kDmaOob : vo_x[p] + (unsigned)off_x);
    }
    c_j += d_j; if (c_j >= segW) { c_j -= segW; ++c_i; }
    c_i += d_i; if (c_i >= rowsT) { c_i -= rowsT; ++c_n; }
    c_n += d_n;
  };

  // ---- transforms of a tile pair: thread (channel ch, pair tp) -> rows 2G, 2G+1 of Yh / of V, tiles 2tp, 2tp+1
  const int tg = tid & 255, ch = tg >> 2, tp = tg & 3;
  const int w_even = wgrad_wino3_op_off(ch, 0) + tp, w_odd = wgrad_wino3_op_off(ch, 1) + tp;     // tile 2tp / tile 2tp+1 of this channel
  // Yh = A dY A^T,  A = [1 0; 1 1; 1 -1; 0 -1]: rows (y0, y0 + y1 | y0 - y1, -y1), the same along the columns
  auto yh_read = [&](int stage, f32x4& ya, f32x4& yb) {
    const float* rs = raw_s + stage * Cfg::RAW_STAGE;
    ya = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 4 * tp);          // dZ row 2i,   columns 4tp .. 4tp+3
    yb = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 16 + 4 * tp);     // dZ row 2i+1
  };
  auto yh_finish = [&](const f32x4& ya, const f32x4& yb, auto gc) {
    constexpr int G = decltype(gc)::value;
    float* oe = yh_s + (G * 8) * XS + w_even;
    float* oo = yh_s + (G * 8) * XS + w_odd;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                                      // transform row 2G + i
      float rr[2][2];                                                                  // [tile][column of the 2x2 block]
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float y00 = ya[2 * t], y01 = ya[2 * t + 1], y10 = yb[2 * t], y11 = yb[2 * t + 1];
        if (G == 0) { rr[t][0] = i == 0 ? y00 : y00 + y10; rr[t][1] = i == 0 ? y01 : y01 + y11; }
        else { rr[t][0] = i == 0 ? y00 - y10 : -y10; rr[t][1] = i == 0 ? y01 - y11 : -y11; }
      }
      oe[(i * 4 + 0) * XS] = rr[0][0];            oo[(i * 4 + 0) * XS] = rr[1][0];
      oe[(i * 4 + 1) * XS] = rr[0][0] + rr[0][1]; oo[(i * 4 + 1) * XS] = rr[1][0] + rr[1][1];
      oe[(i * 4 + 2) * XS] = rr[0][0] - rr[0][1]; oo[(i * 4 + 2) * XS] = rr[1][0] - rr[1][1];
      oe[(i * 4 + 3) * XS] = -rr[0][1];           oo[(i * 4 + 3) * XS] = -rr[1][1];
    }
  };
  // V = B^T d B: strip rows G .. G+2 (patch rows d0,d1,d2 | d1,d2,d3), patch columns 4tp+3 .. 4tp+8 of the 24-float strip row
  auto v_transform = [&](int stage, auto gc) {
    constexpr int G = decltype(gc)::value;
    const float* d = raw_s + stage * Cfg::RAW_STAGE + Cfg::DZ_RAW + ch * (4 * XW) + G * XW + 4 * tp;
    float x[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * XW);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * XW + 4);
      const float q2 = d[r * XW + 8];
      x[r][0] = q0[3]; x[r][1] = q1[0]; x[r][2] = q1[1]; x[r][3] = q1[2]; x[r][4] = q1[3]; x[r][5] = q2;
    }
    float e[2][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      e[0][j] = G ? x[1][j] - x[0][j] : x[0][j] - x[2][j];      // d2 - d1      | d0 - d2
      e[1][j] = G ? x[0][j] - x[2][j] : x[1][j] + x[2][j];      // d1 - d3      | d1 + d2
    }
    float* oe = v_s + (G * 8) * XS + w_even;
    float* oo = v_s + (G * 8) * XS + w_odd;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      oe[(r * 4 + 0) * XS] = e[r][0] - e[r][2]; oo[(r * 4 + 0) * XS] = e[r][2] - e[r][4];
      oe[(r * 4 + 1) * XS] = e[r][1] + e[r][2]; oo[(r * 4 + 1) * XS] = e[r][3] + e[r][4];
      oe[(r * 4 + 2) * XS] = e[r][2] - e[r][1]; oo[(r * 4 + 2) * XS] = e[r][4] - e[r][3];
      oe[(r * 4 + 3) * XS] = e[r][1] - e[r][3]; oo[(r * 4 + 3) * XS] = e[r][3] - e[r][5];
    }
  };
  auto transform = [&](int stage, auto gc, bool with_yh) {          // with_yh is a compile-time constant at every call site
    if constexpr (Cfg::DIAG == 1) return;
    if (with_yh) {
      f32x4 ya, yb;
      yh_read(stage, ya, yb);
      yh_finish(ya, yb, gc);
    }
    v_transform(stage, gc);
  };
  const int sgrp = __builtin_amdgcn_readfirstlane(grp);                    // wave-uniform: scalar branches, no per-element selects

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  const float* A = yh_s + (grp * 8) * XS + wgrad_wino3_op_off(wm * 32 + bl, half);
  const float* B = v_s + (grp * 8) * XS + wgrad_wino3_op_off(wn * 32 + bl, half);
  // this group's 8 xi x 4 K steps (per accumulator the K order of kernels 1 and 2); with SPLIT, behind the first K step, the Yh
  // transform of the OTHER group's rows from raw stage `yh_stage` (after the last chunk: of a stale strip, into rows nobody reads
  // any more -- unconditional, so that the MFMA stream stays one basic block)
  auto mfma_chunk = [&](int yh_stage, auto gother) {
    f32x4 av[8], bv[8], ya, yb;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      av[x] = *reinterpret_cast<const f32x4*>(A + x * XS);
      bv[x] = *reinterpret_cast<const f32x4*>(B + x * XS);
    }
    constexpr bool do_yh = Cfg::SPLIT && Cfg::DIAG != 1;
    if (do_yh) yh_read(yh_stage, ya, yb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Cfg::DIAG == 3) {
#pragma unroll
      for (int x = 0; x < 8; ++x) asm volatile("" ::"v"(av[x]), "v"(bv[x]));
      if (do_yh) yh_finish(ya, yb, gother);
    } else {
#pragma unroll
      for (int x = 0; x < 8; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][0], bv[x][0], acc[x], 0, 0, 0);
      if (Cfg::SPLIT) {
        __builtin_amdgcn_sched_barrier(0);
        if (do_yh) yh_finish(ya, yb, gother);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t2 = 1; t2 < 4; ++t2)
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][t2], bv[x][t2], acc[x], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // own LDS traffic done; with `landed`, also this wave's DMA pieces of the chunk the next transform reads (the `later` chunks
  // issued after it may still be in flight)
  auto phase_end = [&](bool landed, int later) {
    if (landed) {
      if (NST >= 3 && later >= 1) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only((NST - 2) * Cfg::DMA_PER_CHUNK));
      else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    }
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // Phases (one workgroup barrier each); group 0 runs half a period ahead of group 1:
  //   fill      DMA(0) .. DMA(NST-1);  group 0: T(0)
  //   odd  q    group 0: MFMA(q)        group 1: T(q)          -- ends with the strips of chunk q+1 landed
  //   even q    all: issue DMA(q+NST) into stage q % NST (both groups are done with the strips of chunk q)
  //             group 0: T(q+1)         group 1: MFMA(q)
  // SPLIT: T = the V transform only; the group in its MFMA phase does the other group's Yh (group 0 in the odd phase: Yh rows 2, 3
  // of chunk q; group 1 in the even phase: Yh rows 0, 1 of chunk q+1); group 1 issues its pieces of DMA(q+2) in the odd phase
  // (stage (q-1) % 3, free since the odd phase of chunk q-1) instead of DMA(q+3) in the even one.
#pragma unroll
  for (int s = 0; s < NST; ++s)
    if (s < nMine) dma_chunk(s);
  // chunk 0 landed: at most the NST-1 later fills stay in flight
  if (NST >= 3 && nMine >= NST) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only((NST - 1) * Cfg::DMA_PER_CHUNK));
  else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
  __builtin_amdgcn_s_barrier();
  int st = 0, st1 = 1, st2 = NST > 2 ? 2 : 0;               // stages of chunks q, q+1, q+2 (= q-1 with three stages)
  auto next_stage = [&]() { const int t = st; st = st1; st1 = NST > 2 ? st2 : t; st2 = t; };
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  // (one loop per group, each with a single MFMA site: the accumulators stay in place; both execute the same barriers)
  if (sgrp == 0) {
    if (nMine > 0) transform(0, G0{}, true);                              // (the fill is not worth splitting)
    phase_end(false, 0);
    for (int q = 0; q < nMine; ++q) {
      mfma_chunk(st, G1{});                                               // SPLIT: + Yh rows 2, 3 of chunk q
      phase_end(true, q + 2 < nMine ? 1 : 0);
      if (q + NST < nMine) dma_chunk(st);
      if (q + 1 < nMine) transform(st1, G0{}, !Cfg::SPLIT);
      phase_end(false, 0);
      next_stage();
    }
  } else {
    phase_end(false, 0);
    for (int q = 0; q < nMine; ++q) {
      if (Cfg::SPLIT && q >= 1 && q + 2 < nMine) dma_chunk(st2);
      transform(st, G1{}, !Cfg::SPLIT);
      phase_end(true, q + 2 < nMine ? 1 : 0);
      if (!Cfg::SPLIT && q + NST < nMine) dma_chunk(st);
      mfma_chunk(st1, G0{});                                              // SPLIT: + Yh rows 0, 1 of chunk q+1
      phase_end(false, 0);
      next_stage();
    }
  }

  // partial slab: part[ks][xi][co][ci]
  float* slab = a.part + (size_t)ks * 16 * Cout * Cin;
  const int ci = ci0 + wn * 32 + bl;
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      slab[((size_t)(grp * 8 + x) * Cout + co) * Cin + ci] = acc[x][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fourth generation (variant 5): no roles.  Kernels 2 and 3 alternate each wave group between an MFMA phase and a transform phase
// half a period apart; but beside its SIMD partner's MFMA stream a wave issues only ~4 vector or one 16-byte LDS instruction per MFMA
// (profiles/r02_mfma_f32_coissue.json), so a group's transform of BOTH operands (60 vector + 10 LDS-read instructions + 4 DMA pieces)
// took nearly twice as long as the other group's 32 MFMAs and the phases -- not the matrix pipe -- set the pace (timing twins
// 101-103, 111-113: transforms and MFMAs nearly add).  Inside a wave that streams MFMAs an extra instruction costs ~5 cycles, and
// the partner's MFMAs fill even that.  So here all eight waves run the same program: every wave streams its 32 MFMAs of chunk q
// and, between them, transforms its tile pair of chunk q+1:
//     barrier X        operands of chunk q complete in LDS, strips of chunk q+1 landed
//     16 operand reads (all four K steps), the dZ rows of chunk q+1; K step 0 of all eight xi
//     barrier Y        every wave holds its operands in registers: the operand buffers are free
//     K steps 1-3, xi pair by xi pair (the operand registers of a finished pair take the X rows of chunk q+1), interleaved with the
//     Yh and V transforms of chunk q+1 -- results stored straight into the operand buffers -- and the four DMA pieces of chunk
//     q+NST into the stage chunk q's strips left (always issued: beyond the last chunk every lane is out of range, so the waits
//     are constant counts)
// Per accumulator the K order is that of kernels 1-3: the same bits.  Operand layout and raw stages as in kernel 3.  Unlike them
// it takes any Cin (a partial last block of 64 input channels), which gives the stem layer (Cin = 27) the Winograd form too.
template <int NST_ = 3, int DIAG_ = 0>
struct WgradWino5Cfg : WgradWino3Cfg<NST_, DIAG_, 0> {};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) wgrad_wino5_mfma_kernel(const WgradWinoArgs a) {
  constexpr int NT = Cfg::NT, XS = Cfg::XS, XW = Cfg::XW, NX = Cfg::NX, NST = Cfg::NST;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;
  float* v_s = lds + Cfg::OP_FLOATS;
  float* raw_s = lds + 2 * Cfg::OP_FLOATS;                 // NST stages of [dz strip | x strip]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wn = wq & 1, wm = wq >> 1;                      // wm: co half, wn: ci half
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int nCB = (Cin + 63) / 64;                         // the last ci block may be partial (the stem: Cin = 27): its missing channels
                                                            // are out of the X descriptor's range (zeros) and are not stored
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * 64, ci0 = cb * 64;
  const int segW = W / 16, rowsT = H / 2;
  const int nChunksAll = a.N * rowsT * segW;
  const int nMine = nChunksAll > ks ? (nChunksAll - ks + a.splitK - 1) / a.splitK : 0;      // chunks ks, ks + splitK, ...

  // ---- the DMA cursor: (image, tile row, segment) of the next chunk to fetch; one step = splitK chunks
  const int per_img = rowsT * segW;
  int c_n = ks / per_img, c_i = (ks - c_n * per_img) / segW, c_j = ks - c_n * per_img - c_i * segW;
  const int d_n = a.splitK / per_img, d_i = (a.splitK - d_n * per_img) / segW, d_j = a.splitK - d_n * per_img - d_i * segW;
  int left = nMine;                                         // chunks not yet requested

  unsigned vo_dz, vo_x[NX], edge[NX];                      // as in kernel 3
  {
    const int q4 = tid & 3, r = (tid >> 2) & 1, co = tid >> 3;
    vo_dz = (unsigned)(co * HW + r * W + 4 * q4) * 4u;
#pragma unroll
    for (int p = 0; p < NX; ++p) {
      const int e = tid + p * NT;
      const int q6 = e % 6, t2 = e / 6;
      const int r4 = t2 & 3, ci = t2 >> 2;
      vo_x[p] = (unsigned)(ci * HW + r4 * W + 4 * q6) * 4u;
      edge[p] = (r4 == 0 ? 1u : 0u) | (r4 == 3 ? 2u : 0u) | (q6 == 0 ? 4u : 0u) | (q6 == 5 ? 8u : 0u);
    }
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);
  const unsigned planes_dz = 64u * (unsigned)HW * 4u;
  const unsigned planes_x = (unsigned)(Cin - ci0 < 64 ? Cin - ci0 : 64) * (unsigned)HW * 4u;
  // the cursor's chunk: descriptors and strip offsets now (scalar), the four pieces wherever the caller puts them; past the last
  // chunk (`left` <= 0) every lane is out of range: the stage is filled with zeros nobody reads
  struct DmaPlan { tnv3_rsrc_t r_dz, r_x; unsigned off_dz, off_x, border; float* rs; };
  auto dma_plan = [&](int stage) {
    DmaPlan d;
    const bool dead = left <= 0;
    --left;
    d.r_dz = tnv3_make_rsrc(a.dz + ((size_t)(dead ? 0 : c_n) * Cout + co0) * HW, planes_dz);
    d.r_x = tnv3_make_rsrc(a.x + ((size_t)(dead ? 0 : c_n) * Cin + ci0) * HW, planes_x);
    d.off_dz = (unsigned)((2 * c_i * W + 16 * c_j) * 4);
    d.off_x = (unsigned)(((2 * c_i - 1) * W + 16 * c_j - 4) * 4);        // negative at the top-left corner: only out-of-image pieces
    d.border = dead ? 15u : ((c_i == 0 ? 1u : 0u) | (c_i == rowsT - 1 ? 2u : 0u) | (c_j == 0 ? 4u : 0u) | (c_j == segW - 1 ? 8u : 0u));
    if (dead) d.off_dz = kDmaOob;
    d.rs = raw_s + stage * Cfg::RAW_STAGE;
    c_j += d_j; if (c_j >= segW) { c_j -= segW; ++c_i; }
    c_i += d_i; if (c_i >= rowsT) { c_i -= rowsT; ++c_n; }
    c_n += d_n;
    return d;
  };
  auto dma_piece = [&](const DmaPlan& d, int p) {          // p = 0: the dZ piece; 1 .. NX: the X pieces
    if constexpr (Cfg::DIAG == 2) return;
    if (p == 0) tnv3_buf_dma16(d.r_dz, d.rs + wbase * 4, d.off_dz == kDmaOob ? kDmaOob : vo_dz + d.off_dz);
    else tnv3_buf_dma16(d.r_x, d.rs + Cfg::DZ_RAW + ((p - 1) * NT + wbase) * 4,
                        ((edge[p - 1] & d.border) || d.border == 15u) ? kDmaOob : vo_x[p - 1] + d.off_x);
  };

  // ---- transforms of a tile pair: thread (channel ch, pair tp) of its group -> rows 2*grp, 2*grp+1 of Yh and V, tiles 2tp, 2tp+1
  const int tg = tid & 255, ch = tg >> 2, tp = tg & 3;
  const int w_even = wgrad_wino3_op_off(ch, 0) + tp, w_odd = wgrad_wino3_op_off(ch, 1) + tp;
  auto yh_read = [&](int stage, f32x4& ya, f32x4& yb) {
    const float* rs = raw_s + stage * Cfg::RAW_STAGE;
    ya = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 4 * tp);          // dZ row 2i,   columns 4tp .. 4tp+3
    yb = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 16 + 4 * tp);     // dZ row 2i+1
  };
  auto yh_finish = [&](const f32x4& ya, const f32x4& yb, auto gc) {     // Yh = A dY A^T, rows 2G, 2G+1 (kernel 3)
    if constexpr (Cfg::DIAG == 1) return;
    constexpr int G = decltype(gc)::value;
    float* oe = yh_s + (G * 8) * XS + w_even;
    float* oo = yh_s + (G * 8) * XS + w_odd;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rr[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float y00 = ya[2 * t], y01 = ya[2 * t + 1], y10 = yb[2 * t], y11 = yb[2 * t + 1];
        if (G == 0) { rr[t][0] = i == 0 ? y00 : y00 + y10; rr[t][1] = i == 0 ? y01 : y01 + y11; }
        else { rr[t][0] = i == 0 ? y00 - y10 : -y10; rr[t][1] = i == 0 ? y01 - y11 : -y11; }
      }
      oe[(i * 4 + 0) * XS] = rr[0][0];            oo[(i * 4 + 0) * XS] = rr[1][0];
      oe[(i * 4 + 1) * XS] = rr[0][0] + rr[0][1]; oo[(i * 4 + 1) * XS] = rr[1][0] + rr[1][1];
      oe[(i * 4 + 2) * XS] = rr[0][0] - rr[0][1]; oo[(i * 4 + 2) * XS] = rr[1][0] - rr[1][1];
      oe[(i * 4 + 3) * XS] = -rr[0][1];           oo[(i * 4 + 3) * XS] = -rr[1][1];
    }
  };
  struct XRows { f32x4 q0[3], q1[3]; float q2[3]; };
  auto v_read = [&](int stage, XRows& xr, auto gc) {       // strip rows G .. G+2, patch columns 4tp+3 .. 4tp+8 of the 24-float strip row
    constexpr int G = decltype(gc)::value;
    const float* d = raw_s + stage * Cfg::RAW_STAGE + Cfg::DZ_RAW + ch * (4 * XW) + G * XW + 4 * tp;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      xr.q0[r] = *reinterpret_cast<const f32x4*>(d + r * XW);
      xr.q1[r] = *reinterpret_cast<const f32x4*>(d + r * XW + 4);
      xr.q2[r] = d[r * XW + 8];
    }
  };
  auto v_finish = [&](const XRows& xr, auto gc) {          // V = B^T d B, rows 2G, 2G+1 (kernel 3)
    if constexpr (Cfg::DIAG == 1) return;
    constexpr int G = decltype(gc)::value;
    float x[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      x[r][0] = xr.q0[r][3]; x[r][1] = xr.q1[r][0]; x[r][2] = xr.q1[r][1]; x[r][3] = xr.q1[r][2]; x[r][4] = xr.q1[r][3]; x[r][5] = xr.q2[r];
    }
    float e[2][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      e[0][j] = G ? x[1][j] - x[0][j] : x[0][j] - x[2][j];      // d2 - d1      | d0 - d2
      e[1][j] = G ? x[0][j] - x[2][j] : x[1][j] + x[2][j];      // d1 - d3      | d1 + d2
    }
    float* oe = v_s + (G * 8) * XS + w_even;
    float* oo = v_s + (G * 8) * XS + w_odd;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      oe[(r * 4 + 0) * XS] = e[r][0] - e[r][2]; oo[(r * 4 + 0) * XS] = e[r][2] - e[r][4];
      oe[(r * 4 + 1) * XS] = e[r][1] + e[r][2]; oo[(r * 4 + 1) * XS] = e[r][3] + e[r][4];
      oe[(r * 4 + 2) * XS] = e[r][2] - e[r][1]; oo[(r * 4 + 2) * XS] = e[r][4] - e[r][3];
      oe[(r * 4 + 3) * XS] = e[r][1] - e[r][3]; oo[(r * 4 + 3) * XS] = e[r][3] - e[r][5];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  const float* A = yh_s + (grp * 8) * XS + wgrad_wino3_op_off(wm * 32 + bl, half);
  const float* B = v_s + (grp * 8) * XS + wgrad_wino3_op_off(wn * 32 + bl, half);
  constexpr int kLater = (NST - 2) * Cfg::DMA_PER_CHUNK;   // DMA pieces that may stay in flight when the next chunk's strips must be in

  // ---- fill: strips of chunks 0 .. NST-1; everybody transforms chunk 0
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    const DmaPlan d = dma_plan(s);
#pragma unroll
    for (int p = 0; p <= NX; ++p) dma_piece(d, p);
  }
  __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only((NST - 1) * Cfg::DMA_PER_CHUNK));
  __builtin_amdgcn_s_barrier();
  const int sgrp = __builtin_amdgcn_readfirstlane(grp);                    // wave-uniform: one scalar branch around the two loops
  int st = 0, st1 = 1, st2 = NST > 2 ? 2 : 0;               // stages of chunks q, q+1, q+2
  auto next_stage = [&]() { const int t = st; st = st1; st1 = NST > 2 ? st2 : t; st2 = t; };
  auto run = [&](auto gc) {
    {
      f32x4 ya, yb;
      XRows xr;
      yh_read(0, ya, yb);
      v_read(0, xr, gc);
      yh_finish(ya, yb, gc);
      v_finish(xr, gc);
    }
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(kLater));                  // chunk 1 landed
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();                                         // barrier X
    for (int q = 0; q < nMine; ++q) {
      f32x4 av[8], bv[8], ya, yb;
      XRows xr;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        av[x] = *reinterpret_cast<const f32x4*>(A + x * XS);
        bv[x] = *reinterpret_cast<const f32x4*>(B + x * XS);
      }
      yh_read(st1, ya, yb);
      const DmaPlan d = dma_plan(st);                                     // chunk q+NST -> the stage of chunk q (read one iteration ago)
      if constexpr (NST == 2) {                                           // two stages: the strips are due at the end of THIS iteration
#pragma unroll
        for (int p = 0; p <= NX; ++p) dma_piece(d, p);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Cfg::DIAG != 3) {
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][0], bv[x][0], acc[x], 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
      __builtin_amdgcn_s_barrier();                                       // barrier Y: all operand reads of chunk q are done
      __builtin_amdgcn_sched_barrier(0);
      auto pair_mfmas = [&](int x0) {                                     // K steps 1..3 of xi x0, x0+1, alternating accumulators
        if constexpr (Cfg::DIAG != 3) {
#pragma unroll
          for (int t2 = 1; t2 < 4; ++t2)
#pragma unroll
            for (int x = x0; x < x0 + 2; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][t2], bv[x][t2], acc[x], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(av[x0]), "v"(bv[x0]), "v"(av[x0 + 1]), "v"(bv[x0 + 1]));
        }
      };
      // xi 0, 1 + the Yh transform
      pair_mfmas(0);
      yh_finish(ya, yb, gc);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                // VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                // DS write
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NST > 2) dma_piece(d, 0);
      v_read(st1, xr, gc);                                                // (their registers: the operands of xi 0, 1 and the dZ rows)
      __builtin_amdgcn_sched_barrier(0);
      // xi 2, 3 while the X rows arrive
      pair_mfmas(2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NST > 2) {
        dma_piece(d, 1);
        dma_piece(d, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      // xi 4 .. 7 + the V transform
      pair_mfmas(4);
      pair_mfmas(6);
      v_finish(xr, gc);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NST > 2) dma_piece(d, 3);
      static_assert(NX == 3, "piece placement above");
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(kLater));                // strips of chunk q+2 landed (this wave's pieces)
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));                   // this wave's operand stores of chunk q+1 done
      __builtin_amdgcn_s_barrier();                                       // barrier X
      next_stage();
    }
  };
  if (sgrp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // partial slab: part[ks][xi][co][ci]
  float* slab = a.part + (size_t)ks * 16 * Cout * Cin;
  const int ci = ci0 + wn * 32 + bl;
  if (ci < Cin) {
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        slab[((size_t)(grp * 8 + x) * Cout + co) * Cin + ci] = acc[x][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the UPSAMPLED half of a decoder-entry layer (model.py:65,67,69) in the 9-GEMM Winograd form of
// conv_up2x_wino_mfma.h.  Forward: M_xi = sum_ci U'_xi V'_xi for the nine xi at transform rows / columns {0, 1, 3}, U' = G' g G'^T
// (G' = [1 0 0; 1 1 1; 0 0 1]), V' from the 3x3 low-resolution neighbourhood (rows L[i-1] - L[i], L[i], L[i] - L[i+1], then the
// same along the columns), Y = A^T M A.  Hence
//     dU'_xi[co][ci] = sum_{n, low-res pixels}  Yh_xi[co][p] * V'_xi[ci][p],     Yh = rows / columns {0, 1, 3} of A dY A^T
//     dg = G'^T dU' G'                                                           (wgrad_up2x_wino_fold_kernel)
// -- nine GEMMs with K = the low-resolution pixels of the batch: 9 multiply-adds per (co, ci, low-res pixel) instead of the 16 of
// the four 2x2-window launches (and no parity split of dZ, no per-parity reductions).
// Kernel: the structure of wgrad_wino2_mfma_kernel -- 512 threads, two wave groups half a period apart, buffer-descriptor LDS-DMA,
// strip cursor without divisions -- with a 64 co x 128 ci block: nine accumulators per 32 x 32 block fit one wave (144 registers),
// so a wave owns all xi of its block; group g takes ci 64g .. 64g+63 and keeps its own copy of both transformed operands (Yh of
// all 64 co: 12 adds per tile pair, cheaper than sharing it across the groups' phases).  A chunk = 8 low-resolution pixels of one
// row (2 x 16 pixels of dZ): 36 MFMAs per wave.
struct WgradUp2xWinoArgs {
  const float* x_low;   // [N][C0][Hl][Wl]
  const float* dz;      // [N][Cout][2 Hl][2 Wl]
  float* part;          // [splitK][9][Cout][C0]
  int N, C0, Cout, Hl, Wl, splitK;
};

struct WgradUp2xWinoCfg {
  static constexpr int NT = 512, TCH = 8, TS = TCH + 1, NXI = 9, CB = 128;
  static constexpr int OP_FLOATS = NXI * 64 * TS;                       // one transformed operand of one group: [xi][channel][tile]
  static constexpr int DZ_RAW = 64 * 2 * 16;                            // [co][2 rows][16 px]
  static constexpr int XW = 16, X_RAW = CB * 3 * XW;                    // [ci][3 low rows][16 low px: columns 8j-4 .. 8j+11]
  static constexpr int RAW_STAGE = DZ_RAW + X_RAW;
  static constexpr int NX = X_RAW / 4 / NT;                             // x pieces per thread and chunk (3); one dZ piece
  static constexpr int LDS_FLOATS = 4 * OP_FLOATS + 2 * RAW_STAGE;
  static_assert(DZ_RAW / 4 == NT && X_RAW % (4 * NT) == 0, "pieces must deal evenly");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

inline __global__ void __launch_bounds__(WgradUp2xWinoCfg::NT) wgrad_up2x_wino_mfma_kernel(const WgradUp2xWinoArgs a) {
  using Cfg = WgradUp2xWinoCfg;
  constexpr int NT = Cfg::NT, TS = Cfg::TS, XW = Cfg::XW, NX = Cfg::NX, NXI = Cfg::NXI, CB = Cfg::CB;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wm = wq & 1, wn = wq >> 1;                      // wm: co half; wn: 32-channel half of the group's 64 ci
  const int half = lane >> 5, bl = lane & 31;
  float* yh_s = lds + grp * 2 * Cfg::OP_FLOATS;             // this group's operands
  float* v_s = yh_s + Cfg::OP_FLOATS;
  float* raw_s = lds + 4 * Cfg::OP_FLOATS;                  // two stages of [dz strip | x strip], shared by the groups
  const int Hl = a.Hl, Wl = a.Wl, C0 = a.C0, Cout = a.Cout, HWl = Hl * Wl, W = 2 * Wl, HW = 4 * HWl;
  const int nCB = C0 / CB;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * 64, ci0 = cb * CB;
  const int segW = Wl / 8;
  const int nChunksAll = a.N * Hl * segW;
  const int nMine = nChunksAll > ks ? (nChunksAll - ks + a.splitK - 1) / a.splitK : 0;      // chunks ks, ks + splitK, ...

  // ---- the DMA cursor: (image, low-res row, 8-pixel segment) of the next chunk to fetch; one step = splitK chunks
  const int per_img = Hl * segW;
  int c_n = ks / per_img, c_i = (ks - c_n * per_img) / segW, c_j = ks - c_n * per_img - c_i * segW;
  const int d_n = a.splitK / per_img, d_i = (a.splitK - d_n * per_img) / segW, d_j = a.splitK - d_n * per_img - d_i * segW;

  unsigned vo_dz, vo_x[NX], edge[NX];                      // edge bits: piece lies in strip row 0 / row 2 / column piece 0 / column piece 3
  {
    const int q4 = tid & 3, r = (tid >> 2) & 1, co = tid >> 3;            // dZ piece of [co][2 rows][4 pieces]
    vo_dz = (unsigned)(co * HW + r * W + 4 * q4) * 4u;
#pragma unroll
    for (int p = 0; p < NX; ++p) {
      const int e = tid + p * NT;                                          // X piece of [ci][3 rows][4 pieces]
      const int q = e & 3, t2 = e >> 2;
      const int r3 = t2 % 3, ci = t2 / 3;
      vo_x[p] = (unsigned)(ci * HWl + r3 * Wl + 4 * q) * 4u;
      edge[p] = (r3 == 0 ? 1u : 0u) | (r3 == 2 ? 2u : 0u) | (q == 0 ? 4u : 0u) | (q == 3 ? 8u : 0u);
    }
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);             // scalar: the LDS-DMA destinations (M0) stay on the SALU
  auto dma_chunk = [&](int stage) {                                        // the cursor's chunk -> raw stage; then the cursor advances
    const tnv3_rsrc_t r_dz = tnv3_make_rsrc(a.dz + ((size_t)c_n * Cout + co0) * HW, 64u * (unsigned)HW * 4u);
    const tnv3_rsrc_t r_x = tnv3_make_rsrc(a.x_low + ((size_t)c_n * C0 + ci0) * HWl, (unsigned)CB * (unsigned)HWl * 4u);
    const int off_dz = (2 * c_i * W + 16 * c_j) * 4;
    const int off_x = ((c_i - 1) * Wl + 8 * c_j - 4) * 4;                // negative at the top-left corner: only out-of-image pieces
    const unsigned border = (c_i == 0 ? 1u : 0u) | (c_i == Hl - 1 ? 2u : 0u) | (c_j == 0 ? 4u : 0u) | (c_j == segW - 1 ? 8u : 0u);
    float* rs = raw_s + stage * Cfg::RAW_STAGE;
    tnv3_buf_dma16(r_dz, rs + wbase * 4, vo_dz + (unsigned)off_dz);
#pragma unroll
    for (int p = 0; p < NX; ++p)
      tnv3_buf_dma16(r_x, rs + Cfg::DZ_RAW + (p * NT + wbase) * 4, (edge[p] & border) ? kDmaOob : vo_x[p] + (unsigned)off_x);
    c_j += d_j; if (c_j >= segW) { c_j -= segW; ++c_i; }
    c_i += d_i; if (c_i >= Hl) { c_i -= Hl; ++c_n; }
    c_n += d_n;
  };

  // ---- transforms of a low-res pixel pair (2tp, 2tp+1): thread (channel ch, pair tp) of its group -> Yh (co = ch) and V' (ci = 64 grp + ch)
  const int tg = tid & 255, ch = tg >> 2, tp = tg & 3;
  const bool odd = (tp & 1) != 0;
  auto transform = [&](int stage) {
    const float* rs = raw_s + stage * Cfg::RAW_STAGE;
    {   // Yh = rows / columns {0, 1, 3} of A dY A^T,  A = [1 0; 1 1; 1 -1; 0 -1]: (y0, y0 + y1, -y1), the same along the columns
      const f32x4 ya = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 4 * tp);          // dZ row 2i,   columns 4tp .. 4tp+3
      const f32x4 yb = *reinterpret_cast<const f32x4*>(rs + ch * 32 + 16 + 4 * tp);     // dZ row 2i+1
      float* o = yh_s + ch * TS + 2 * tp;
#pragma unroll
      for (int ra = 0; ra < 3; ++ra) {
        float rr[2][2];                                                                  // [pixel of the pair][column of its 2x2 block]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float y00 = ya[2 * t], y01 = ya[2 * t + 1], y10 = yb[2 * t], y11 = yb[2 * t + 1];
          rr[t][0] = ra == 0 ? y00 : (ra == 1 ? y00 + y10 : -y10);
          rr[t][1] = ra == 0 ? y01 : (ra == 1 ? y01 + y11 : -y11);
        }
        float* oi = o + (ra * 3) * 64 * TS;
        oi[0 * 64 * TS] = rr[0][0];            oi[0 * 64 * TS + 1] = rr[1][0];
        oi[1 * 64 * TS] = rr[0][0] + rr[0][1]; oi[1 * 64 * TS + 1] = rr[1][0] + rr[1][1];
        oi[2 * 64 * TS] = -rr[0][1];           oi[2 * 64 * TS + 1] = -rr[1][1];
      }
    }
    {   // V': low rows i-1, i, i+1 (strip rows 0..2), low columns 8j + 2tp - 1 .. 8j + 2tp + 2 = strip columns 2tp+3 .. 2tp+6
      const float* d = rs + Cfg::DZ_RAW + (grp * 64 + ch) * (3 * XW) + 4 * (tp >> 1);
      float x[3][4];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * XW);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * XW + 4);
        const float q2 = d[r * XW + 8];
        x[r][0] = odd ? q1[1] : q0[3]; x[r][1] = odd ? q1[2] : q1[0]; x[r][2] = odd ? q1[3] : q1[1]; x[r][3] = odd ? q2 : q1[2];
      }
      float* o = v_s + ch * TS + 2 * tp;
#pragma unroll
      for (int ra = 0; ra < 3; ++ra) {
        float rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[j] = ra == 0 ? x[0][j] - x[1][j] : (ra == 1 ? x[1][j] : x[1][j] - x[2][j]);
        float* oi = o + (ra * 3) * 64 * TS;
        oi[0 * 64 * TS] = rr[0] - rr[1]; oi[0 * 64 * TS + 1] = rr[1] - rr[2];
        oi[1 * 64 * TS] = rr[1];         oi[1 * 64 * TS + 1] = rr[2];
        oi[2 * 64 * TS] = rr[1] - rr[2]; oi[2 * 64 * TS + 1] = rr[2] - rr[3];
      }
    }
  };

  f32x16 acc[NXI];
#pragma unroll
  for (int x = 0; x < NXI; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  const float* A = yh_s + (wm * 32 + bl) * TS + half;
  const float* B = v_s + (wn * 32 + bl) * TS + half;
  auto mfma_chunk = [&]() {                                 // 4 pixel pairs x 9 xi
    constexpr int NSTEP = 4 * NXI;
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int t2 = s / NXI, xi = s - t2 * NXI;
      av[s % RING] = A[xi * 64 * TS + 2 * t2];
      bv[s % RING] = B[xi * 64 * TS + 2 * t2];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc[s % NXI] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], acc[s % NXI], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  auto phase_end = [&](bool dma_too) {
    if (dma_too) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // Phases as in wgrad_wino2_mfma_kernel: group 0 runs half a period ahead of group 1; one loop per group (one MFMA site each).
  const int sgrp = __builtin_amdgcn_readfirstlane(grp);
  if (nMine > 0) {
    dma_chunk(0);
    if (nMine > 1) dma_chunk(1);
  }
  phase_end(true);
  if (sgrp == 0) {
    if (nMine > 0) transform(0);
    phase_end(false);
    for (int q = 0; q < nMine; ++q) {
      mfma_chunk();
      phase_end(true);
      if (q + 2 < nMine) dma_chunk(q & 1);
      if (q + 1 < nMine) transform((q + 1) & 1);
      phase_end(false);
    }
  } else {
    phase_end(false);
    for (int q = 0; q < nMine; ++q) {
      transform(q & 1);
      phase_end(true);
      if (q + 2 < nMine) dma_chunk(q & 1);
      mfma_chunk();
      phase_end(false);
    }
  }

  // partial slab: part[ks][xi][co][ci]
  float* slab = a.part + (size_t)ks * NXI * Cout * C0;
  const int ci = ci0 + grp * 64 + wn * 32 + bl;
#pragma unroll
  for (int x = 0; x < NXI; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      slab[((size_t)x * Cout + co) * C0 + ci] = acc[x][r];
    }
}

// d9[co][ci][3][3] = G'^T (sum_ks part[ks][.][co][ci]) G',  G' = [1 0 0; 1 1 1; 0 0 1]: the slabs are added in the fixed order
// ks = 0, 1, ... in fp64 (deterministic), one thread per (co, ci).
inline __global__ void __launch_bounds__(256) wgrad_up2x_wino_fold_kernel(const float* __restrict__ part, float* __restrict__ d9, int Cout, int C0,
                                                                       int splitK) {
  const long n = (long)Cout * C0;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    double u[9];
#pragma unroll
    for (int x = 0; x < 9; ++x) u[x] = 0.0;
    for (int k = 0; k < splitK; ++k) {
      const float* src = part + (size_t)k * 9 * n + e;
#pragma unroll
      for (int x = 0; x < 9; ++x) u[x] += (double)src[(size_t)x * n];
    }
    float uf[3][3], t[3][3];
#pragma unroll
    for (int x = 0; x < 9; ++x) uf[x / 3][x % 3] = (float)u[x];
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2) { t[0][b2] = uf[0][b2] + uf[1][b2]; t[1][b2] = uf[1][b2]; t[2][b2] = uf[1][b2] + uf[2][b2]; }
    float* o = d9 + e * 9;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) { o[kh * 3 + 0] = t[kh][0] + t[kh][1]; o[kh * 3 + 1] = t[kh][1]; o[kh * 3 + 2] = t[kh][1] + t[kh][2]; }
  }
}

// dW[Cout][C0+C1][9]: input channels < C0 from d9[Cout][C0][9], the rest from dw_skip[Cout][C1][9]
inline __global__ void __launch_bounds__(256) wgrad_up2x_join_kernel(const float* __restrict__ d9, const float* __restrict__ dw_skip,
                                                                  float* __restrict__ dw, int Cout, int C0, int C1) {
  const int Cin = C0 + C1;
  const long total = (long)Cout * Cin * 9;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int tap = (int)(e % 9);
    const long t = e / 9;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    dw[e] = ci < C0 ? d9[((size_t)co * C0 + ci) * 9 + tap] : dw_skip[((size_t)co * C1 + (ci - C0)) * 9 + tap];
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
