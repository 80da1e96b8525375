// conv3x3_wino43s_mfma.h -- Winograd F(4x4, 3x3) with ALL 36 transform coefficients of an output block in ONE wave
// (tnv3_conv3x3_wino43_forward, kernel variant 0; conv3x3_wino43_mfma.h is its predecessor, variant 1, kept as the A/B twin).
//
// Why.  The 32x32x2 kernel gives a wave nine of the 36 coefficients xi (9 x 16 accumulator registers), so the output transform
// A^T M A needs the partial sums of four waves: 96 KB through LDS per round, eight barriers, 1750 instructions per wave and tile --
// 15-19 k cycles of write-out per tile against 5.8 k per chunk of useful work (profiles/r03_wino43_timeline.json).  With
// v_mfma_f32_16x16x4_f32 one accumulator is FOUR registers: a wave holds all 36 xi of a (16 channels x 16 tiles) block in 144, the
// output transform is per-lane register arithmetic, and nothing is exchanged.
//
// Mapping.  512 threads, two waves per SIMD (what a wave issues besides its fp32 MFMAs is hidden only behind the MFMAs of the OTHER
// wave of its SIMD -- profiles/r02_mfma_f32_coissue.json; the first form of this kernel, one 512-register wave per SIMD with two
// accumulator sets, measured 6540 cycles per step against 4670 for its MFMAs alone: profiles/r04_wino43s_one_wave_per_simd_twins.json).
// Wave = 16 output channels x 16 tiles of 4x4 (one tile row of 64 pixels) x 36 xi.  Two workgroup geometries (template CBW):
//   CBW = 4:  64 channels x 2 tile rows (8 x 64 pixels); waves w, w + 4 (one SIMD) share the channel block = the same A quads
//   CBW = 8: 128 channels x 1 tile row  (4 x 64 pixels): ONE patch transform feeds 128 output channels -- half the transform
//             instructions per MFMA; the two wave groups transform on alternate steps.
//   MFMA 16x16x4: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D reg r = row 4 (l >> 4) + r, col l & 15.
//   M = 16 output channels, N = 16 tile columns, K = 4 input channels; a chunk = 8 input channels = two K steps s; lane group
//   g = l >> 4 takes input channels 2 g + s.
//   * A operand (U = G g G^T) from L2 into registers: panel [16-channel block][chunk][xi pair 18][lane][4 = (s, xi & 1)], one
//     16-byte load per xi pair = 4 MFMAs; ring of AR quads, AR - 1 pairs ahead.
//   * B operand (V = B^T d B) from LDS: V[stage 3][xi pair 18][tile row][g 4][tile column 16][4 = (s, xi & 1)], one ds_read_b128
//     per pair: a wave reads 1 KB contiguous, conflict-free; ring of three.
//   * The raw halo tile arrives by LDS-DMA as 17 pieces of four columns per row STARTING ONE COLUMN LEFT of the tile (w0 - 1 ..
//     w0 + 66; the global address of a piece is 4-byte, not 16-byte aligned): the 6 columns of tile column tc's patch are piece tc
//     and the first half of piece tc + 1 -- one ds_read_b128 + one ds_read_b64 per raw row instead of one 16-byte and two
//     conflicting 4-byte reads (29 % of the predecessor's LDS cycles were bank conflicts of those).  What the shift costs: the
//     column left of the image and the one right of it are real memory (the neighbouring row's ends), so the lanes of tile column 0
//     / 15 of a border tile zero their first / last transformed column (three v_cndmask each per half patch), and the ONE piece per
//     image that would start before the image (channel 0, row 0) is patched by one lane from a guarded load.  A piece that runs
//     over the END of the image keeps its leading in-range dwords (raw buffers are range-checked per dword).
//   * Thread = half a patch (channel, tile, row half rh): five raw rows -> three rows of B^T d B -> 9 ds_write_b64 (a lane's two
//     values of a xi pair are adjacent).  Lane order (tc & 7, s, tc >> 3, g & 1) makes raw reads and V stores conflict-free in the
//     hardware's lane groups.
//   * Pipeline: step = one chunk of one tile; the stream of steps runs ACROSS tile boundaries.  Step sigma: MFMAs on V(sigma),
//     transform raw(sigma + 2) -> V(sigma + 2), DMA raw(sigma + 3); three V stages and two raw stages; ONE barrier per step, and
//     the B operands of the next step's first pairs are read before it (V(sigma + 1) was complete one barrier earlier).
//   * A step is 36 slots of two MFMAs, each followed by at most one piece of other work (sched_barrier fences).
//   * Write-out: per lane 4 channels x (4x4 pixels): A^T M A in registers (110 operations per channel), addend / BatchNorm / ReLU,
//     16-byte stores (16 lanes = 256 contiguous bytes of an output row).  No LDS, no barrier.
// Needs Cout % 64 == 0 (CBW = 8: % 128), H % 4 == 0, W % 64 == 0 (CBW = 4, H % 8 == 4: the last tile row's lower half is outside the
// image).  Deterministic; not bit-identical to the 32x32x2 kernel (the output transform adds in another order).
#pragma once
#include <type_traits>
#include <utility>
#include "conv3x3_wino43_mfma.h"

namespace tnv3 {

struct Wino43SBase {
  static constexpr int NT = 512, TW = 64;
  static constexpr int B_RING = 3, B_DIST = 2;
  static constexpr int NR = 2;
  static constexpr int CC = 8;                            // the panel's chunk (the kernels' step: CC * KB channels)
};
// MODE 0: the plain layer (36 xi = 18 pairs; raw tile of the input itself).  MODE 1: the UPSAMPLED half of a decoder-entry layer
// (model.py:65,67,69: conv3x3 over nn.Upsample(2)(x_low)) computed from the low-resolution tensor -- 25 xi = 13 pairs (the last one
// half empty), see the section "upsampled half" below.  MODE 2 (round 5): that half's DATA GRADIENT -- the raw tile is dZ at the full
// resolution (MODE 0's loader), 25 xi (MODE 1's operand pairs and accumulators), and the write-out sums each 2x2 block of the 4x4
// tile into the low-resolution gradient; see the section "data gradient of the upsampled half".
template <int CBW_, int MODE_ = 0>
struct Wino43SCfg : Wino43SBase {
  static_assert(CBW_ == 4 || CBW_ == 8, "16-channel blocks per workgroup");
  static constexpr int MODE = MODE_;
  static constexpr int NXI = MODE_ ? 25 : 36, PAIRS = (NXI + 1) / 2;
  static constexpr int A_CHUNK_FLOATS = PAIRS * 64 * 4;   // one (16-channel block, chunk of 8 input channels) of the panel
  static constexpr bool LOW = MODE_ == 1;                 // the SOURCE is the low-resolution tensor
  static constexpr int RQ = LOW ? 9 : 17;                 // pieces per raw row: 17 from column w0 - 1 / 9 from low-resolution column w0 / 2 - 1
  static constexpr int CBW = CBW_, TRW = 8 / CBW_;        // tile rows per workgroup
  static constexpr int KB = CBW_ / 4;                     // 8-channel blocks per step: 1 (64 x 2 rows) / 2 (128 x 1 row: 16 channels per step,
  static constexpr int SC = 8 * KB;                       // so that every thread still has half a patch to transform in every step)
  static constexpr int NP = PAIRS * KB;                   // operand pairs (xi pair, 8-channel block) = A quads = B quads per wave and step
  static constexpr int NV = CBW_ == 4 ? 3 : 2;            // V stages: 3 = the transform runs two steps ahead and the next step's first B quads are
                                                          // read before the step's barrier; 2 (LDS: 16-channel stages) = one step ahead
  static constexpr int MB = 16 * CBW, TB = 16 * TRW, TH = 4 * TRW;
  static constexpr int RROWS = LOW ? TH / 2 + 2 : TH + 2;        // raw halo tile per channel: rows h0-1 .. h0+TH / low-resolution rows h0/2-1 .. h0/2+TH/2
  // pieces per channel plane.  MODE 0: a multiple of 16 pieces, so that the two channels a 16-lane group of a ds_read_b128 touches fall into
  // disjoint bank ranges; MODE 1 (8-byte reads, 32-lane groups): 8 mod 16 pieces = 32 mod 64 banks between the group's two channels
  static constexpr int RPLANE = LOW ? (RROWS * RQ + 7) / 16 * 16 + 8 : (RROWS * RQ + 15) / 16 * 16;
  static constexpr int RAW_SLOTS = SC * RPLANE;           // pieces per stage: 1408 (2.75 per thread) / 1792 (3.5)
  static constexpr int RAW_STAGE = RAW_SLOTS * 4;         // floats
  static constexpr int NDMA = (RAW_SLOTS + NT - 1) / NT;  // DMA instructions per wave and step; the last one: the first DMA_LAST_WAVES waves
  static constexpr int DMA_LAST_WAVES = (RAW_SLOTS - (NDMA - 1) * NT + 63) / 64;      // (a partial last wave: its surplus lanes are out of range)
  static constexpr int V_PAIR = TRW * 256;                // floats per operand pair: [tile row][g 4][tile column 16][4]
  static constexpr int V_STAGE = NP * V_PAIR;             // 36 KB
  static constexpr int LDS_FLOATS = NV * V_STAGE + NR * RAW_STAGE;      // 155,648 / 131,072 bytes
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
  static constexpr bool B_ACROSS = NV == 3 && NP % B_RING == 0;      // the next step's first B quads are read before this step's barrier
};

inline size_t conv3x3_wino43s_packed_floats(int cin, int cout) {
  if (cin <= 0 || cout <= 0 || cout % 16) return 0;
  return (size_t)(cout / 16) * ((cin + Wino43SBase::CC - 1) / Wino43SBase::CC) * (18 * 64 * 4) + kPackZeroTail;
}

// w[..][3][3] -> panel u[co / 16][chunk][pair = 3 i + j / 2][lane = (g = ci % 8 / 2) * 16 + co % 16][(s = ci % 2) * 2 + j % 2] for
// xi = (i, j), zero for ci >= Cin; then kPackZeroTail zeros.  One work item = one (16-channel block, chunk, lane): the 3x3 filters of
// its two input channels once, 18 float4.
inline long conv3x3_wino43s_pack_items(int Cout, int Cin) { return (long)(Cout / 16) * ((Cin + 7) / 8) * 64 + kPackZeroTail / 4; }
__device__ __forceinline__ void conv3x3_wino43s_pack_elements(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, long s_co, long s_ci,
                                                              int flip, long q0, long stride) {
  const int nch = (Cin + 7) / 8;
  const long quads = (long)(Cout / 16) * nch * 64;
  f32x4* u4 = reinterpret_cast<f32x4*>(u);
  for (long q = q0; q < quads + kPackZeroTail / 4; q += stride) {
    if (q >= quads) { u4[quads * 18 + (q - quads)] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; continue; }
    const int ln = (int)(q & 63);
    const int r = (int)(q >> 6), k = r % nch, cb = r / nch;
    const int co = 16 * cb + (ln & 15), g = ln >> 4;
    float rowv[2][6][3];                               // [s][row i of G applied down the filter's columns][column c]
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int ci = 8 * k + 2 * g + sx;
      float f[9];
      if (ci < Cin) {
        const float* gp = w + (long)co * s_co + (long)ci * s_ci;
#pragma unroll
        for (int t = 0; t < 9; ++t) f[t] = flip ? gp[8 - t] : gp[t];
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) f[t] = 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) rowv[sx][i][c] = wino43_g_row(i, f[c], f[3 + c], f[6 + c]);
    }
    f32x4* dst = u4 + ((long)r * 18) * 64 + ln;
#pragma unroll
    for (int pr = 0; pr < 18; ++pr) {
      const int i = pr / 3, j0 = 2 * (pr % 3);
      f32x4 v;
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        v[2 * sx] = wino43_g_row(j0, rowv[sx][i][0], rowv[sx][i][1], rowv[sx][i][2]);
        v[2 * sx + 1] = wino43_g_row(j0 + 1, rowv[sx][i][0], rowv[sx][i][1], rowv[sx][i][2]);
      }
      dst[pr * 64] = v;
    }
  }
}
inline __global__ void __launch_bounds__(256) conv3x3_wino43s_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin,
                                                                         long s_co, long s_ci, int flip) {
  conv3x3_wino43s_pack_elements(w, u, Cout, Cin, s_co, s_ci, flip, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// ---- upsampled half (MODE 1).  Interpolation points (0, +-1, +-b, inf) with b^2 = kU43B: the upsampled signal's polynomial is
//      (1 + x) l(x^2), so the point -1 vanishes for ANY b and the rows of +-b stay proportional -- b is free, and it sets the precision:
//      the row of the point 0 / inf of B^T E is (B a - (1 + B) b_ + c) / (B b_ - (1 + B) c + e), B = b^2 (amplification 2 (1 + B): 10 with
//      Lavin's b = 2), and the columns of +-b of A^T carry b^2 / b^3 into output rows 2 / 3.  Swept in fp32 emulation on upsampled post-ReLU
//      inputs (tests/studies/up2x_points_study.py, profiles/r05_up2x_points_study.json): max error vs fp64 relative to the output scale
//      0.9-1.0e-5 at B = 4 (Lavin), 6-8e-6 for B in 2 .. 3.5 (rms 5.4e-7 -> 5.0e-7; the error sits in output row / column 3), 1.0-1.3e-5
//      at B = 1.5 (the points +-1 and +-b close in: G's rows grow as 1 / (B - 1)); the plain layers' set (0, +-3/4, +-3/2, inf), which has no
//      point -1 and keeps all 36 products, reaches 2.4e-6.  kU43B = 2.75 (exact in binary, as is 1 + B).  U'[i'][j'] = G' g G'^T over the indices (0, 1, e, o, 5) with G' = [1/B 0 0; (1 1 1) / (1 - B); -(1 B B) / (B (B - 1));
//      -(1 1 B) / (B - 1); 0 0 1] -- rows 0 and infinity of Toom-Cook's G, twice its row of the point 1 (the patch transform carries
//      c - B b_, half of B^T's row), and the two combinations of G[+b], G[-b] that meet the SAME transformed value b_ - c for the even / odd
//      output rows (A^T's columns of +-b differ in the sign of the odd rows).
//      Panel u[co / 16][chunk][pair 13][lane = (ci % 8 / 2) * 16 + co % 16][(ci % 2) * 2 + slot]: pairs 0-4 = ((0, j'), (1, j')),
//      5-9 = ((e, j'), (o, j')), 10-12 = ((5, 0), (5, 1)), ((5, e), (5, o)), ((5, 5), zero); the first c0 input channels of w[Cout][Cin][3][3].
constexpr float kU43B = 2.75f, kU43B1 = 1.0f + kU43B;
__device__ __forceinline__ float wino43u_g_row(int i, float g0, float g1, float g2) {
  switch (i) {
    case 0: return (1.0f / kU43B) * g0;
    case 1: return (1.0f / (1.0f - kU43B)) * ((g0 + g1) + g2);
    case 2: return (-1.0f / (kU43B * (kU43B - 1.0f))) * g0 + (-1.0f / (kU43B - 1.0f)) * (g1 + g2);
    case 3: return (-1.0f / (kU43B - 1.0f)) * (g0 + g1) + (-kU43B / (kU43B - 1.0f)) * g2;
    default: return g2;
  }
}
inline size_t conv_up2x_wino43_packed_floats(int c0, int cout) {
  if (c0 <= 0 || cout <= 0 || cout % 16) return 0;
  return (size_t)(cout / 16) * ((c0 + 7) / 8) * (13 * 64 * 4) + kPackZeroTail;
}
inline long conv_up2x_wino43_pack_items(int Cout, int c0) { return (long)(Cout / 16) * ((c0 + 7) / 8) * 64 + kPackZeroTail / 4; }
inline __global__ void __launch_bounds__(256) conv_up2x_wino43_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int c0) {
  const int nch = (c0 + 7) / 8;
  const long quads = (long)(Cout / 16) * nch * 64;
  f32x4* u4 = reinterpret_cast<f32x4*>(u);
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads + kPackZeroTail / 4; q += (long)gridDim.x * 256) {
    if (q >= quads) { u4[quads * 13 + (q - quads)] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; continue; }
    const int ln = (int)(q & 63);
    const int r = (int)(q >> 6), k = r % nch, cb = r / nch;
    const int co = 16 * cb + (ln & 15), g = ln >> 4;
    float uu[2][5][5];                                  // [s][i'][j']
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int ci = 8 * k + 2 * g + sx;
      float f[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) f[t] = ci < c0 ? w[((size_t)co * Cin + ci) * 9 + t] : 0.0f;
      float rowv[5][3];                                 // G' applied down the filter's columns
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) rowv[i][c] = wino43u_g_row(i, f[c], f[3 + c], f[6 + c]);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) uu[sx][i][j] = wino43u_g_row(j, rowv[i][0], rowv[i][1], rowv[i][2]);
    }
    f32x4* dst = u4 + ((long)r * 13) * 64 + ln;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      dst[j * 64] = f32x4{uu[0][0][j], uu[0][1][j], uu[1][0][j], uu[1][1][j]};
      dst[(5 + j) * 64] = f32x4{uu[0][2][j], uu[0][3][j], uu[1][2][j], uu[1][3][j]};
    }
    dst[10 * 64] = f32x4{uu[0][4][0], uu[0][4][1], uu[1][4][0], uu[1][4][1]};
    dst[11 * 64] = f32x4{uu[0][4][2], uu[0][4][3], uu[1][4][2], uu[1][4][3]};
    dst[12 * 64] = f32x4{uu[0][4][4], 0.0f, uu[1][4][4], 0.0f};
  }
}

// ---- data gradient of the upsampled half (MODE 2).  Points (0, +-1, +-b, inf), b = kD43b = 3/2 (B = b^2 = 9/4: every constant of the
//      transforms an exact binary fraction; the plain layers' larger point is 3/2 too).  U''[i'][j'] = G'' w~ G''^T over the indices
//      (0, 1, b, -b, 5), w~[ci][co][kh][kw] = w[co][ci][2 - kh][2 - kw], with Toom-Cook's G rows scaled by the block sum's column factors:
//      G'' = [1/B 0 0; (1 1 1) / (1 - B); (1 + b) (1 b B) / (2 B (B - 1)); (1 - b) (1 -b B) / (2 B (B - 1)); 0 0 1].
//      Panel u[ci / 16][chunk over co][pair 13][lane = (co % 8 / 2) * 16 + ci % 16][(co % 2) * 2 + slot], pairs as in MODE 1 with (e, o) -> (b, -b).
constexpr float kD43b = 1.5f, kD43B = kD43b * kD43b, kD43B1 = 1.0f + kD43B;
__device__ __forceinline__ float wino43d_g_row(int i, float g0, float g1, float g2) {
  constexpr float nb = 1.0f / (2.0f * kD43B * (kD43B - 1.0f));
  switch (i) {
    case 0: return (1.0f / kD43B) * g0;
    case 1: return (1.0f / (1.0f - kD43B)) * ((g0 + g1) + g2);
    case 2: return ((1.0f + kD43b) * nb) * ((g0 + kD43b * g1) + kD43B * g2);
    case 3: return ((1.0f - kD43b) * nb) * ((g0 - kD43b * g1) + kD43B * g2);
    default: return g2;
  }
}
inline size_t dgrad_up2x_wino43_packed_floats(int c0, int cout) {
  if (c0 <= 0 || cout <= 0 || c0 % 16) return 0;
  return (size_t)(c0 / 16) * ((cout + 7) / 8) * (13 * 64 * 4) + kPackZeroTail;
}
inline long dgrad_up2x_wino43_pack_items(int Cout, int c0) { return (long)(c0 / 16) * ((Cout + 7) / 8) * 64 + kPackZeroTail / 4; }
inline __global__ void __launch_bounds__(256) dgrad_up2x_wino43_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int c0) {
  const int nch = (Cout + 7) / 8;
  const long quads = (long)(c0 / 16) * nch * 64;
  f32x4* u4 = reinterpret_cast<f32x4*>(u);
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads + kPackZeroTail / 4; q += (long)gridDim.x * 256) {
    if (q >= quads) { u4[quads * 13 + (q - quads)] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; continue; }
    const int ln = (int)(q & 63);
    const int r = (int)(q >> 6), k = r % nch, cb = r / nch;
    const int ci = 16 * cb + (ln & 15), g = ln >> 4;      // the kernel's output channel = an input channel of the layer
    float uu[2][5][5];                                  // [s][i'][j']
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int co = 8 * k + 2 * g + sx;                // the kernel's contraction channel = an output channel of the layer
      float f[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) f[t] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + (8 - t)] : 0.0f;      // flipped taps
      float rowv[5][3];
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) rowv[i][c] = wino43d_g_row(i, f[c], f[3 + c], f[6 + c]);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) uu[sx][i][j] = wino43d_g_row(j, rowv[i][0], rowv[i][1], rowv[i][2]);
    }
    f32x4* dst = u4 + ((long)r * 13) * 64 + ln;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      dst[j * 64] = f32x4{uu[0][0][j], uu[0][1][j], uu[1][0][j], uu[1][1][j]};
      dst[(5 + j) * 64] = f32x4{uu[0][2][j], uu[0][3][j], uu[1][2][j], uu[1][3][j]};
    }
    dst[10 * 64] = f32x4{uu[0][4][0], uu[0][4][1], uu[1][4][0], uu[1][4][1]};
    dst[11 * 64] = f32x4{uu[0][4][2], uu[0][4][3], uu[1][4][2], uu[1][4][3]};
    dst[12 * 64] = f32x4{uu[0][4][4], 0.0f, uu[1][4][4], 0.0f};
  }
}

// Table-driven pack of panels of BOTH Winograd forms in one launch (layout 0-2: F(2x2) panels, conv3x3_wino_mfma.h; 3: F(4x4) panels of the 32x32x2 kernel; 4: of the 16x16x4 kernel)
inline __global__ void __launch_bounds__(256) conv3x3_wino_pack_multi43_kernel(const WinoPackTable t) {
  int lo = 0, hi = t.count;                    // first_block[lo] <= blockIdx.x < first_block[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const int k = lo;
  const long nb = t.first_block[k + 1] - t.first_block[k];
  const long e0 = (long)((int)blockIdx.x - t.first_block[k]) * 256 + threadIdx.x;
  if (t.layout[k] == 4) conv3x3_wino43s_pack_elements(t.w[k], t.u[k], t.cout[k], t.cin[k], t.s_co[k], t.s_ci[k], t.flip[k], e0, nb * 256);
  else if (t.layout[k] == 3) conv3x3_wino43_pack_elements(t.w[k], t.u[k], t.cout[k], t.cin[k], t.s_co[k], t.s_ci[k], t.flip[k], e0, nb * 256);
  else conv3x3_wino_pack_elements(t.w[k], t.u[k], t.cout[k], t.cin[k], t.cpad[k], t.s_co[k], t.s_ci[k], t.flip[k], t.layout[k], e0, nb * 256);
}

template <int I, int N, class F>
__device__ __forceinline__ void wino43s_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wino43s_for<I + 1, N>(f);
  }
}

// The full 1-D output transform A^T (conv3x3_wino43_mfma.h) of six coefficients -> four outputs, 11 operations
__device__ __forceinline__ void wino43s_at6(float m0, float m1, float m2, float m3, float m4, float m5, float (&o)[4]) {
  const float p1 = m1 + m2, q1 = m1 - m2, p2 = m3 + m4, q2 = m3 - m4;
  o[0] = fmaf(0.5f / kW43S, p2, fmaf(1.0f / kW43S, p1, m0));
  o[1] = q1 + q2;
  o[2] = fmaf(kW43_2S, p2, kW43S * p1);
  o[3] = fmaf(kW43_4S2, q2, fmaf(kW43S2, q1, m5));
}

// CBW: geometry (above).  GROW, TS: the step's schedule.  A quad (one per operand pair, NP per step) is named by its pair; the quads of
// the next step's first GROW pairs are requested in the LAST slots of a step -- when the patch transform (slots TS .. TS + 16) has
// released its registers -- and the others five pairs ahead of their MFMAs.  Why: vmcnt retires in order, so a filter load issued after
// the step's raw DMA (slots 0 ..: HBM latency, ~3700 cycles at 288 x 512 under load) cannot be consumed before that DMA has landed; with
// a uniform five-pair ring the MFMAs of pair 5 waited for it (+2100 cycles per step at 288 x 512, +1450 at 144 x 256, nothing where
// the input sits in L2: profiles/r04_wino43s_two_waves_twins_before_grow.json).  Now the first load younger than the DMA feeds pair GROW.
// TL = 1 (libtnv3_diag.so only): s_memtime totals of one mid-grid workgroup, [wave 8][8] uint64 to a.stats: 0 prologue, 1 steps,
// 2 write-outs, 3 steps walked, 4 tiles walked.  DG (diag only, WRONG results): timing twins -- bit 0 no raw DMA after the prologue,
// bit 1 no patch transform, bit 2 no A loads (the ring keeps the prologue's quads), bit 3 no B reads, bit 4 no MFMAs, bit 5 no output stores.
// POOL = 1: a.pool_dst receives MaxPool2d(2, 2) of the block.
template <int CBW, int STATS = 0, int GROW = 10, int TS = 10, int TL = 0, int DG = 0, int POOL = 0, int MODE = 0>
__global__ void __launch_bounds__(Wino43SBase::NT) conv3x3_wino43s_kernel(const WinoArgs a) {
  using Cfg = Wino43SCfg<CBW, MODE>;
  static_assert(MODE == 0 || (POOL == 0 && (STATS == 0 || (MODE == 2 && STATS == 2))),
                "the upsampled half (and its data gradient) write plain sums; the data gradient may take the BatchNorm-backward sums of what it writes");
  static_assert(STATS >= 0 && STATS <= 2 && (STATS != 2 || POOL == 0), "STATS: 0 none, 1 BatchNorm forward statistics, 2 BatchNorm backward sums (data gradient)");
  constexpr bool LOW = Cfg::LOW;
  constexpr int SC = Cfg::SC, KB = Cfg::KB, NP = Cfg::NP, NV = Cfg::NV, NSLOT = 2 * NP, PAIRS = Cfg::PAIRS, NXI = Cfg::NXI;
  constexpr int NT = Cfg::NT, MB = Cfg::MB, V_STAGE = Cfg::V_STAGE, RAW_STAGE = Cfg::RAW_STAGE, V_PAIR = Cfg::V_PAIR;
  constexpr int A_DIST = 5, RQ = Cfg::RQ, T_PIECES = MODE == 1 ? 5 : (MODE == 2 ? 11 : 17), GPS = 3;      // GPS: grow loads per slot
  constexpr bool B_ACROSS = Cfg::B_ACROSS;
  constexpr int GS = NSLOT - (GROW + GPS - 1) / GPS;                  // first slot of the grow phase
  constexpr int TD = NV - 1;                                          // the transform's lead over the MFMAs, in steps
  static_assert(GROW >= A_DIST && GROW <= 16 && TS >= Cfg::NDMA && TS + T_PIECES <= NSLOT, "step schedule");
  static_assert((GROW - 1) < (GS >> 1), "a grow load refills the quad of a pair this step has finished");
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* v_s = lds;
  float* raw_s = lds + NV * V_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;      // (H, W: the tile grid's = the full resolution; MODE 1 reads its source, MODE 2
  const int SH = H >> LOW, SW = W >> LOW, SHW = SH * SW;                   //  writes its result, at half of both)
  const int tilesH = (H + Cfg::TH - 1) / Cfg::TH, tilesW = W / Cfg::TW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (Cin + SC - 1) / SC;               // steps per tile
  const int nCh8 = (Cin + 7) / 8;                        // the panel's chunks

  // Three cursors walk the same list of (tile, chunk) steps: M (the MFMAs), T (the patch transform: TD steps ahead), D (the raw DMA: TD + 1
  // ahead).  Their tile walks are copies; only the mutable fields cost scalar registers.  (The filter loads run one step ahead of M: the
  // next step's panel slice follows from M's own fields.)
  ConvTileWalk wM, wT, wD;
  wM.init(blockIdx.x, gridDim.x, nMB, nPT, tilesH, tilesW);
  if (!wM.valid) return;
  wT = wM; wD = wM;
  int kM = 0, kT = 0, kD = 0;

  unsigned long long tl_acc[3] = {0, 0, 0}, tl_last = 0;
  int tl_steps = 0, tl_tiles = 0;
  auto tl_stamp = [&](int slot) {
    if constexpr (TL != 0) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tl_acc[slot] += now - tl_last;
      tl_last = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (TL != 0) tl_last = __builtin_amdgcn_s_memtime();

  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);
  // ---- MFMA role: wave = (16-channel block cb, tile row tr)
  const int cb = swave % CBW, tr = swave / CBW;
  const int b_lane = tr * 256 + lane * 4;                                                 // + pair * V_PAIR
  const unsigned a_lane_b = (unsigned)lane * 16u;
  const tnv3_rsrc_t r_panel = tnv3_make_rsrc(a.u, (unsigned)((size_t)(Cout / 16) * nCh8 * Cfg::A_CHUNK_FLOATS * 4));

  f32x4 acc[2 * PAIRS];                                 // [xi], register r = channel 4 (lane >> 4) + r
  f32x4 aq[NP], bq[Cfg::B_RING];

  // ---- D cursor: per-tile piece offsets.  Slot e = tid + i * 512 -> (channel c, row, piece q) of [SC][RPLANE]; the pad pieces of a
  //      plane, rows outside the image and channels >= Cin read out of the descriptor's range = zeros.
  unsigned voD[Cfg::NDMA];
  tnv3_rsrc_t r_srcD;
  auto set_d = [&]() {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);
    const int h0 = (wD.trow * Cfg::TH) >> LOW, w0 = (wD.tcol * Cfg::TW) >> LOW;      // in the source's pixels
#pragma unroll
    for (int i = 0; i < Cfg::NDMA; ++i) {
      const int e = t_op + i * NT;
      const int c = e / Cfg::RPLANE, rem = e - c * Cfg::RPLANE;
      const int r = rem / RQ, q = rem - r * RQ;
      const int gh = h0 - 1 + r, gw = w0 - 1 + 4 * q;
      const bool ok = wD.valid && e < Cfg::RAW_SLOTS && rem < Cfg::RROWS * RQ && gh >= 0 && gh < SH;
      voD[i] = ok ? (unsigned)(c * SHW + gh * SW + gw) * 4u : kDmaOob;      // (channel 0, row 0, column -1: wraps beyond the range = zeros; see fix_corner)
    }
    const int nn = wD.valid ? wD.n : 0;
    r_srcD = tnv3_make_rsrc(a.src + (size_t)nn * Cin * SHW, (unsigned)Cin * (unsigned)SHW * 4u);
  };
  auto dma_piece = [&](int i, int stage) {              // piece i of chunk kD of the D tile -> raw stage
    if (i < Cfg::NDMA - 1 || swave < Cfg::DMA_LAST_WAVES)
      tnv3_buf_dma16(r_srcD, raw_s + stage * RAW_STAGE + (i * NT + wbase) * 4, voD[i] + (unsigned)kD * (unsigned)(SC * 4) * (unsigned)SHW);
  };
  auto adv_d = [&]() {
    if (++kD >= nChunks) { kD = 0; wD.next(); set_d(); }
  };
  auto full_barrier = [&]() {
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // Everything below is instantiated per wave group (waves 0-3 / 4-7: the two waves of every SIMD) and selected by ONE scalar branch: a
  // group is one row half RH of the patch transform (its code differs per half), and a branch inside the step loop would put the
  // accumulators through phi copies.
  auto body = [&](auto grpc) {
  constexpr int RH = decltype(grpc)::value;
  // ---- transform role: thread = half patch (channel ci = 8 kb + 2 g + s, tile row ttr, tile column tc, row half RH); lane bits (tc & 7, s,
  //      tc >> 3, g & 1), wave bits (g >> 1, ttr | kb, RH): the 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...) then hold
  //      tile columns {0-3, 12-15} of one channel and {4-11} of its neighbour (planes a multiple of 16 pieces apart: 16 different
  //      16-byte bank groups), and the 16 lanes of a ds_write_b64 group hold 8 tile columns x 2 s = 32 different banks.
  const int t_tc = (lane & 7) | ((lane >> 1) & 8), t_s = (lane >> 3) & 1, t_g = ((lane >> 5) & 1) | ((swave & 1) << 1);
  const int t_x = (swave >> 1) & 1, t_tr = CBW == 4 ? t_x : 0, t_kb = CBW == 4 ? 0 : t_x;
  const int t_ci = 8 * t_kb + 2 * t_g + t_s;
  // MODE 0: raw rows 4 ttr + RH .., piece tc (+ row * 68 floats; second piece + 4); V pairs 9 RH ..
  // MODE 1: low-resolution raw rows 2 ttr + RH .., floats 2 tc .. 2 tc + 3 (+ row * 36 floats); V pairs 0-4 (RH 0) / 5-12 (RH 1)
  // MODE 2: MODE 0's raw rows, MODE 1's V pairs
  const int t_src = t_ci * (Cfg::RPLANE * 4) + (LOW ? ((2 * t_tr + RH) * RQ) * 4 + 2 * t_tc : ((4 * t_tr + RH) * RQ + t_tc) * 4);
  const int t_dst = (t_kb * PAIRS + (MODE ? 5 : 9) * RH) * V_PAIR + t_tr * 256 + t_g * 64 + t_tc * 4 + t_s * 2;         // + pair * V_PAIR

  // ---- T cursor: the transform of half a patch, in pieces (the step places one piece behind an MFMA slot)
  float* t_raw = nullptr;      // raw stage + t_src
  float* t_v = nullptr;        // V stage + t_dst
  bool zl = false, zr = false, fix_corner = false;
  auto set_t = [&](int raw_stage, int v_stage) {
    t_raw = raw_s + raw_stage * RAW_STAGE + t_src;
    t_v = v_s + v_stage * V_STAGE + t_dst;
    zl = wT.tcol == 0 && t_tc == 0;                      // the column left of the image: zero, but the shifted piece holds the previous row's end
    zr = wT.tcol == tilesW - 1 && t_tc == 15;            // the column right of it
    fix_corner = wT.valid && wT.trow == 0 && wT.tcol == 0 && kT == 0;
  };
  f32x4 tq0[5];
  wf2 tq1[5];
  float tt[3][6];
  wf2 ulo[3], uhi[3];                                    // MODE 1: the thread's three low-resolution rows, columns (-1, 0) and (1, 2) of the patch
  float uc[2][4], uw[2][4];
  auto t_piece = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    if constexpr (MODE == 1) {
      // ---- upsampled half.  The 6 x 6 patch of the upsampled tensor is E l E^T of a 4 x 4 low-resolution patch l (rows / columns
      //      a, b, b, c, c, e).  With the points (0, +-1, +-sqrt(B), inf), B = kU43B, B^T E has the rows (B a - (1 + B) b + c), 2 (c - B b), 0,
      //      -(1 + sqrt B) (b - c), (sqrt B - 1) (b - c), (B b - (1 + B) c + e): the point -1 vanishes, +-sqrt B are proportional.  Five products per
      //      axis remain -- indices (0, 1, e, o, 5), where e / o carry the SAME transformed value b - c against two filters (for the even / the
      //      odd output rows: A^T's columns of +-sqrt B differ in the sign of the odd rows) -- 25 of the 36.  Row half RH = 0: indices 0, 1 from
      //      rows (a, b, c); RH = 1: e / o, 5 from (b, c, e).
      if constexpr (P == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          ulo[r] = *reinterpret_cast<const wf2*>(t_raw + r * (RQ * 4));
          uhi[r] = *reinterpret_cast<const wf2*>(t_raw + r * (RQ * 4) + 2);
        }
        if (fix_corner) {                                // the piece before the image's first element (channel 0, low-resolution row 0)
          if (lane < 2 && (swave & 3) == 0) {            // (that piece = columns -1 .. 2: all of tile column 0's row and the first half of tile column 1's)
            const tnv3_rsrc_t ri = tnv3_make_rsrc(a.src + (size_t)wT.n * Cin * SHW, (unsigned)Cin * (unsigned)SHW * 4u);
            const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
            const bool l0 = lane == 0;                   // (selects, not branches: the compiler turned the branchy form into an indexed scratch store)
            ulo[1 - RH] = wf2{l0 ? 0.0f : x[1], l0 ? x[0] : x[2]};
            uhi[1 - RH] = wf2{l0 ? x[1] : uhi[1 - RH][0], l0 ? x[2] : uhi[1 - RH][1]};
          }
        }
      } else if constexpr (P == 1) {                     // down the patch's four columns
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float r0 = x < 2 ? ulo[0][x & 1] : uhi[0][x & 1], r1 = x < 2 ? ulo[1][x & 1] : uhi[1][x & 1], r2 = x < 2 ? ulo[2][x & 1] : uhi[2][x & 1];
          float c0, c1;
          if constexpr (RH == 0) { c0 = fmaf(kU43B, r0, fmaf(-kU43B1, r1, r2)); c1 = fmaf(-kU43B, r1, r2); }      // (a, b, c) -> B a - (1 + B) b + c,  c - B b
          else { c0 = r0 - r1; c1 = fmaf(kU43B, r0, fmaf(-kU43B1, r1, r2)); }                                   // (b, c, e) -> b - c,  B b - (1 + B) c + e
          if (x == 0) { c0 = zl ? 0.0f : c0; c1 = zl ? 0.0f : c1; }
          if (x == 3) { c0 = zr ? 0.0f : c0; c1 = zr ? 0.0f : c1; }
          uc[0][x] = c0; uc[1][x] = c1;
        }
      } else if constexpr (P == 2 || P == 3) {           // along a row: (p0 .. p3) -> the values of indices 0, 1, e = o, 5
        constexpr int k = P - 2;
        const float p0 = uc[k][0], p1 = uc[k][1], p2 = uc[k][2], p3 = uc[k][3];
        uw[k][0] = fmaf(kU43B, p0, fmaf(-kU43B1, p1, p2));
        uw[k][1] = fmaf(-kU43B, p1, p2);
        uw[k][2] = p1 - p2;
        uw[k][3] = fmaf(kU43B, p1, fmaf(-kU43B1, p2, p3));
      } else {                                           // V pairs: (0, j') with (1, j') -- (e, j') with (o, j') -- row 5 in three pairs
        auto jv = [](int j) constexpr { return j < 2 ? j : (j < 4 ? 2 : 3); };      // j' = (0, 1, e, o, 5) -> the row pass's value
        if constexpr (RH == 0) {
#pragma unroll
          for (int j = 0; j < 5; ++j) *reinterpret_cast<wf2*>(t_v + j * V_PAIR) = wf2{uw[0][jv(j)], uw[1][jv(j)]};
        } else {
#pragma unroll
          for (int j = 0; j < 5; ++j) *reinterpret_cast<wf2*>(t_v + j * V_PAIR) = wf2{uw[0][jv(j)], uw[0][jv(j)]};
          *reinterpret_cast<wf2*>(t_v + 5 * V_PAIR) = wf2{uw[1][0], uw[1][1]};
          *reinterpret_cast<wf2*>(t_v + 6 * V_PAIR) = wf2{uw[1][2], uw[1][2]};
          *reinterpret_cast<wf2*>(t_v + 7 * V_PAIR) = wf2{uw[1][3], 0.0f};
        }
      }
    } else if constexpr (MODE == 2) {
      // ---- data gradient of the upsampled half.  dX_low = 2x2 block sums of the plain 3x3 correlation of dZ with the transposed, flipped
      //      filter.  In F(4x4) form over the points (0, +-1, +-b, inf) the block sum P A^T (P = [1 1 0 0; 0 0 1 1]) has a ZERO column at the point
      //      -1 (A^T's column (1, -1, 1, -1)): five products per axis remain, indices (0, 1, b, -b, 5).  b = kD43b = 3/2, B = b^2.  B^T's rows:
      //      (B d0 - (1 + B) d2 + d4), (d4 - B d2) + (d3 - B d1), (d4 - d2) +- b (d3 - d1), (B d1 - (1 + B) d3 + d5).  Row half 0: rows 0, 1 from raw
      //      rows d0 .. d4; row half 1: rows b, -b, 5 from d1 .. d5.  The filter rows carry P A^T's column factors (2, 1 + b, 1 - b), so that
      //      out0 = m0 + m1 + (mb + m-b), out1 = m1 + B (mb + m-b) + m5 (write-out below).
      if constexpr (P == 0) {
#pragma unroll
        for (int r = 0; r < 5; ++r) tq0[r] = *reinterpret_cast<const f32x4*>(t_raw + r * (RQ * 4));
        if (fix_corner) {
          if (lane == 0 && (swave & 3) == 0) {
            const tnv3_rsrc_t ri = tnv3_make_rsrc(a.src + (size_t)wT.n * Cin * HW, (unsigned)Cin * (unsigned)HW * 4u);
            const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
            tq0[1 - RH] = f32x4{0.0f, x[0], x[1], x[2]};
          }
        }
      } else if constexpr (P == 5) {
#pragma unroll
        for (int r = 0; r < 5; ++r) tq1[r] = *reinterpret_cast<const wf2*>(t_raw + r * (RQ * 4) + 4);
      } else if constexpr (P < 8) {                      // first pass, down patch column c
        constexpr int c = P < 5 ? P - 1 : P - 2;
        float x[5], o[3];
#pragma unroll
        for (int r = 0; r < 5; ++r) x[r] = c < 4 ? tq0[r][c < 4 ? c : 0] : tq1[r][c < 4 ? 0 : c - 4];
        if constexpr (RH == 0) {                         // x = d0 .. d4
          o[0] = fmaf(kD43B, x[0], fmaf(-kD43B1, x[2], x[4]));
          o[1] = fmaf(-kD43B, x[2], x[4]) + fmaf(-kD43B, x[1], x[3]);
          o[2] = 0.0f;
        } else {                                         // x = d1 .. d5
          const float e = x[3] - x[1], f = x[2] - x[0];
          o[0] = fmaf(kD43b, f, e);
          o[1] = fmaf(-kD43b, f, e);
          o[2] = fmaf(kD43B, x[0], fmaf(-kD43B1, x[2], x[4]));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float v = o[r];
          if constexpr (c == 0) v = zl ? 0.0f : v;
          if constexpr (c == 5) v = zr ? 0.0f : v;
          tt[r][c] = v;
        }
      } else {                                           // second pass along row r of the half: five values, straight into V
        constexpr int r = P - 8;
        if constexpr (RH == 1 || r < 2) {
          const float(&d)[6] = tt[r];
          const float e = d[4] - d[2], f = d[3] - d[1];
          uw[0][0] = fmaf(kD43B, d[0], fmaf(-kD43B1, d[2], d[4]));
          uw[0][1] = fmaf(-kD43B, d[2], d[4]) + fmaf(-kD43B, d[1], d[3]);
          uw[0][2] = fmaf(kD43b, f, e);
          uw[0][3] = fmaf(-kD43b, f, e);
          const float v5 = fmaf(kD43B, d[1], fmaf(-kD43B1, d[3], d[5]));
          if constexpr (r == 0) {                        // rows 0 / b: the first floats of pairs 0-4 (row half 0) / 5-9 (row half 1); parked until row 1 / -b
#pragma unroll
            for (int j = 0; j < 4; ++j) uc[0][j] = uw[0][j];
            uc[1][0] = v5;
          } else if constexpr (r == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<wf2*>(t_v + j * V_PAIR) = wf2{uc[0][j], uw[0][j]};
            *reinterpret_cast<wf2*>(t_v + 4 * V_PAIR) = wf2{uc[1][0], v5};
          } else {                                       // row 5 (row half 1 only): pairs 10-12
            *reinterpret_cast<wf2*>(t_v + 5 * V_PAIR) = wf2{uw[0][0], uw[0][1]};
            *reinterpret_cast<wf2*>(t_v + 6 * V_PAIR) = wf2{uw[0][2], uw[0][3]};
            *reinterpret_cast<wf2*>(t_v + 7 * V_PAIR) = wf2{v5, 0.0f};
          }
        }
      }
    } else if constexpr (P == 0) {                       // raw rows RH .. RH + 4 of the patch, columns 0-3
#pragma unroll
      for (int r = 0; r < 5; ++r) tq0[r] = *reinterpret_cast<const f32x4*>(t_raw + r * (RQ * 4));
      if (fix_corner) {                                  // scalar branch, taken once per image: the piece before the image's first element
        if (lane == 0 && (swave & 3) == 0) {             // (patch row 1 of channel 0, tile (0, 0): BOTH row halves read it -- as their row 1 / row 0)
          const tnv3_rsrc_t ri = tnv3_make_rsrc(a.src + (size_t)wT.n * Cin * HW, (unsigned)Cin * (unsigned)HW * 4u);
          const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
          tq0[1 - RH] = f32x4{0.0f, x[0], x[1], x[2]};
        }
      }
    } else if constexpr (P == 5) {                       // (patch columns 4, 5: read once columns 0-3 have released their registers)
#pragma unroll
      for (int r = 0; r < 5; ++r) tq1[r] = *reinterpret_cast<const wf2*>(t_raw + r * (RQ * 4) + 4);
    } else if constexpr (P < 8) {                        // first pass, down patch column c: rows 3 RH .. 3 RH + 2 of B^T d
      constexpr int c = P < 5 ? P - 1 : P - 2;
      float x[5], o[3];
#pragma unroll
      for (int r = 0; r < 5; ++r) x[r] = c < 4 ? tq0[r][c < 4 ? c : 0] : tq1[r][c < 4 ? 0 : c - 4];
      if constexpr (RH == 1) {
        const float d[6] = {0.0f, x[0], x[1], x[2], x[3], x[4]};
        wino43_bt_half<1>(d, o);
      } else {
        const float d[6] = {x[0], x[1], x[2], x[3], x[4], 0.0f};
        wino43_bt_half<0>(d, o);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float v = o[r];
        if constexpr (c == 0) v = zl ? 0.0f : v;
        if constexpr (c == 5) v = zr ? 0.0f : v;
        tt[r][c] = v;
      }
    } else {                                             // second pass along row 3 RH + r, results straight into V: pair 3 (3 RH + r) + j / 2
      constexpr int r = (P - 8) / 3, sub = (P - 8) % 3;
      const float(&d)[6] = tt[r];
      float* vd = t_v + (3 * r) * V_PAIR;
      if constexpr (sub == 0) {
        const float o0 = fmaf(kW43_4S4, d[0], fmaf(-kW43_5S2, d[2], d[4]));
        const float aa = fmaf(-kW43_4S2, d[2], d[4]), bb = fmaf(-kW43_4S2, d[1], d[3]);
        const float o1 = fmaf(kW43S, bb, aa);
        tt[r][0] = fmaf(-kW43S, bb, aa);                 // o2, parked in the consumed slot until the next piece
        *reinterpret_cast<wf2*>(vd) = wf2{o0, o1};
      } else if constexpr (sub == 1) {
        const float cc = fmaf(-kW43S2, d[2], d[4]), ee = fmaf(-kW43S2, d[1], d[3]);
        const float o3 = fmaf(kW43_2S, ee, cc);
        const float o4 = fmaf(-kW43_2S, ee, cc);
        *reinterpret_cast<wf2*>(vd + V_PAIR) = wf2{d[0], o3};
        tt[r][0] = o4;
      } else {
        const float o5 = fmaf(kW43_4S4, d[1], fmaf(-kW43_5S2, d[3], d[5]));
        *reinterpret_cast<wf2*>(vd + 2 * V_PAIR) = wf2{d[0], o5};
      }
    }
  };
  auto adv_t = [&]() {
    if (++kT >= nChunks) { kT = 0; wT.next(); }
  };

  // ---- A / B operand streams.  Operand pair q = kb * 18 + p of step k: the panel's chunk k * KB + kb (beyond the last: out of the
  //      descriptor's range = zeros), xi pair p.
  auto a_base = [&](int mb, int k) -> unsigned {        // byte offset of this wave's (16-channel block, first chunk of step k) in the panel
    return (unsigned)((mb * CBW + cb) * nCh8 + k * KB) * (unsigned)(Cfg::A_CHUNK_FLOATS * 4);
  };
  // (a chunk beyond the panel's last -- the second half of a 16-channel step when Cin % 16 <= 8 -- must read zeros: through the LANE offset,
  //  the scalar offset of a buffer access is not range-checked)
  auto a_lane_of = [&](int k, int q) -> unsigned { return (KB == 1 || q < PAIRS || k * KB + 1 < nCh8) ? a_lane_b : kDmaOob; };
  // Quad q of a step's panel slice sits 1024 q bytes behind the slice's start: one scalar offset per quad (36 scalar adds per step).
  // Round 6 measured the alternative -- (q & 3) * 1024 in the instruction's immediate offset, only (q >> 2) * 4096 through the scalar offset:
  // 9 adds per step -- and found nothing (profiles/r06_a_imm_offsets_ab.json: 0.0 to +2 % per launch over the eight plain-layer shapes,
  // bit-identical; the weight-gradient kernel's per-instruction cost, DESIGN 3.1k, does not carry over: here the scalar adds sit in the shadow
  // of the MFMAs they are interleaved with), and the 128-channel instantiations with a statistics epilogue lose two registers to it (they
  // sit at exactly 256: a spill, and a spill inside a counted-vmcnt region is a wrong result, 3.1g).  Kept as a build switch for the record.
#ifndef TNV3_A_IMM_OFFSETS
#define TNV3_A_IMM_OFFSETS 0
#endif
  constexpr bool A_IMM = TNV3_A_IMM_OFFSETS != 0 && !(CBW == 8 && STATS != 0);
  auto a_load = [&](unsigned lane_off, unsigned so_base, int q) -> f32x4 {
    if constexpr (A_IMM) return tnv3_buf_load_f4(r_panel, lane_off + (unsigned)(q & 3) * 1024u, so_base + (unsigned)(q >> 2) * 4096u);
    else return tnv3_buf_load_f4(r_panel, lane_off, so_base + (unsigned)q * 1024u);
  };
  auto next_mb = [&]() -> int {                          // the channel block of the tile after M's (the walk's rule)
    int mb = wM.mb + wM.d_mb;
    if (mb >= nMB) mb -= nMB;
    return mb;
  };

  // ---- prologue: the first TD + 1 raw tiles, the first TD V stages, the operand quads of step 0
  int sv = 0, sr = 0;                                    // V stage of step sigma = sigma % NV, raw stage of raw(sigma) = sigma & 1
  set_d();
#pragma unroll
  for (int i = 0; i < Cfg::NDMA; ++i) dma_piece(i, 0);
  adv_d();
#pragma unroll
  for (int i = 0; i < Cfg::NDMA; ++i) dma_piece(i, 1);
  adv_d();
  full_barrier();
  set_t(0, 0);
  wino43s_for<0, T_PIECES>(t_piece);
  adv_t();
  if constexpr (NV == 3) {
    full_barrier();
#pragma unroll
    for (int i = 0; i < Cfg::NDMA; ++i) dma_piece(i, 0);
    adv_d();
    set_t(1, 1);
    wino43s_for<0, T_PIECES>(t_piece);
    adv_t();
  }
  {
    const unsigned s0 = a_base(wM.mb, 0);
#pragma unroll
    for (int q = 0; q < GROW; ++q) aq[q] = a_load(a_lane_of(0, q), s0, q);
  }
  full_barrier();
  if constexpr (B_ACROSS) {
#pragma unroll
    for (int q = 0; q < Cfg::B_DIST; ++q) bq[q] = *reinterpret_cast<const f32x4*>(v_s + q * V_PAIR + b_lane);
  }
  tl_stamp(0);
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

  // ---- one step: chunk kM of the M tile.  2 NP slots of two MFMAs (operand pair q, K step s).  Slot 2 q first issues the B read of pair
  //      q + 2 and the A load of pair q + 5 (if that is one of this step's pairs GROW .. NP - 1); slots 0 .. carry the DMA pieces, TS ..
  //      the transform pieces, GS .. the A loads of the next step's pairs 0 .. GROW - 1.
  auto step = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    int svn, svt;
    if constexpr (NV == 3) { svn = sv == 2 ? 0 : sv + 1; svt = svn == 2 ? 0 : svn + 1; } else { svn = sv ^ 1; svt = svn; }
    const float* bM = v_s + sv * V_STAGE + b_lane;
    const float* bN = v_s + svn * V_STAGE + b_lane;
    const bool last_k = kM + 1 >= nChunks;
    const unsigned soM = a_base(wM.mb, kM), soA = last_k ? a_base(next_mb(), 0) : a_base(wM.mb, kM + 1);
    const int kA = last_k ? 0 : kM + 1;                  // (past the last step: a slice nobody uses, inside the panel)
    const unsigned a_lane_hi = a_lane_of(kM, PAIRS);     // lane offset of this step's second 8-channel block
    const int srt = (sr + TD) & 1, srd = srt ^ 1;        // raw stages of the transform's / the DMA's step
    set_t(srt, svt);
    if constexpr (!B_ACROSS) {                           // (no V stage to spare, or a pair count the ring does not divide: this step's first B quads after the barrier)
#pragma unroll
      for (int q = 0; q < Cfg::B_DIST; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bM + q * V_PAIR);
    }
    wino43s_for<0, NSLOT>([&](auto ic) {
      constexpr int IDX = decltype(ic)::value, q = IDX >> 1, s = IDX & 1, p = q % PAIRS;
      if constexpr (s == 0) {
        constexpr int qb = q + Cfg::B_DIST, qa = q + A_DIST;
        if constexpr ((DG & 8) == 0 && (qb < NP || B_ACROSS))
          bq[qb % Cfg::B_RING] = *reinterpret_cast<const f32x4*>((qb < NP ? bM : bN) + (qb % NP) * V_PAIR);
        if constexpr ((DG & 4) == 0 && qa >= GROW && qa < NP) aq[qa] = a_load(qa < PAIRS ? a_lane_b : a_lane_hi, soM, qa);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4& av = aq[q];
      const f32x4& bv = bq[q % Cfg::B_RING];
      constexpr bool ZERO = FIRST && q < PAIRS && s == 0;   // a tile's first K step starts the accumulator
      if constexpr ((DG & 16) != 0) {
        if constexpr (ZERO) { acc[2 * p] = f32x4{av[0], bv[0], 0.0f, 0.0f}; acc[2 * p + 1] = f32x4{av[1], bv[1], 0.0f, 0.0f}; }
      } else {
        acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2 * s], bv[2 * s], ZERO ? zero4 : acc[2 * p], 0, 0, 0);
        if constexpr (2 * p + 1 < NXI)                   // (25 xi: the last pair is half empty)
          acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2 * s + 1], bv[2 * s + 1], ZERO ? zero4 : acc[2 * p + 1], 0, 0, 0);
      }
      if constexpr (IDX < Cfg::NDMA) { if constexpr ((DG & 1) == 0) dma_piece(IDX, srd); }
      if constexpr (IDX >= TS && IDX - TS < T_PIECES) { if constexpr ((DG & 2) == 0) t_piece(std::integral_constant<int, IDX - TS>{}); }
      if constexpr (IDX >= GS && (DG & 4) == 0) {
#pragma unroll
        for (int j = (IDX - GS) * GPS; j < (IDX - GS + 1) * GPS && j < GROW; ++j) aq[j] = a_load(a_lane_of(kA, j), soA, j);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // this wave's raw pieces of the DMA's step have landed (they are OLDER than the step's NP A loads), its V stores are done
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only((DG & 4) ? 0 : NP));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    sv = svn; sr ^= 1;
    adv_d();
    adv_t();
    if constexpr (TL != 0) ++tl_steps;
  };

  // ---- write-out of this wave's block of the M tile
  auto writeout = [&]() {
    const bool has_affine = !STATS && a.scale != nullptr, has_mean = !STATS && a.mean != nullptr, has_addend = a.addend != nullptr;
    int ln_w = lane;
    TNV3_OPAQUE_V(ln_w);
    const int g = ln_w >> 4, tc = ln_w & 15;
    const int e_m0 = wM.mb * MB + 16 * cb;               // this wave's first channel
    const int oh = wM.trow * Cfg::TH + 4 * tr, ow = wM.tcol * Cfg::TW + 4 * tc;
    const size_t plane0 = ((size_t)wM.n * Cout + e_m0) * HW;
    const unsigned planes_b = 16u * (unsigned)HW * 4u;
    const tnv3_rsrc_t r_dst = tnv3_make_rsrc(a.dst + plane0, planes_b);
    if constexpr (MODE == 2) {
      // data gradient of the upsampled half: dX_low (2x2 per tile) = Q M Q^T over the indices (0, 1, b, -b, 5) with Q = [1 1 1 1 0; 0 1 B B 1];
      // M[i'][j'] = acc[2 j' + i'] (i' = 0, 1), acc[10 + 2 j' + (i' - 2)] (i' = b, -b), acc[20 + j'] (i' = 5).  dst = [N][Cout][H / 2][W / 2].
      const int LW = W >> 1, LHW = HW >> 2;
      const tnv3_rsrc_t r_low = tnv3_make_rsrc(a.dst + ((size_t)wM.n * Cout + e_m0) * LHW, 16u * (unsigned)LHW * 4u);
      const unsigned lane_off_l = oh < H ? (unsigned)((4 * g) * LHW + (oh >> 1) * LW + (ow >> 1)) * 4u : kDmaOob;
      // STATS == 2 (round 6): dX_low IS dA of the block whose activation the decoder entry upsamples (model.py:64-69: nothing else reads it), so this
      // write-out takes that block's two BatchNorm + ReLU backward sums exactly as the plain data gradient's does below: bn_z = its raw convolution
      // output at the LOW resolution (two 8-byte rows per channel and lane), constants in a.mean / a.scale / a.shift / a.bn_c4; one statistics tile
      // = this wave's tile row = 2 x 32 low-resolution pixels, a.N * (H / 4) * tilesW of them per channel.
      constexpr bool BNB2 = STATS == 2;
      const tnv3_rsrc_t r_zl = tnv3_make_rsrc((BNB2 ? a.bn_z : a.dst) + ((size_t)wM.n * Cout + e_m0) * LHW, 16u * (unsigned)LHW * 4u);
      float c_mu_n = 0.0f, c_is_n = 0.0f, c_ga_n = 0.0f, c_be_n = 0.0f;
      auto load_c = [&](int r) {
        const int ch = e_m0 + 4 * g + r;
        c_mu_n = a.mean[ch]; c_is_n = a.scale[ch]; c_ga_n = a.shift[ch]; c_be_n = a.bn_c4[ch];
      };
      if constexpr (BNB2) load_c(0);
      auto q2 = [](float q0, float q1, float qb, float qm, float q5, float (&o)[2]) {
        const float sb = qb + qm;
        o[0] = (q0 + q1) + sb;
        o[1] = fmaf(kD43B, sb, q1) + q5;
      };
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float wv[2][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float o[2];
          q2(acc[2 * j][r], acc[2 * j + 1][r], acc[10 + 2 * j][r], acc[11 + 2 * j][r], acc[20 + j][r], o);
          wv[0][j] = o[0]; wv[1][j] = o[1];
        }
        const float c_mu = c_mu_n, c_is = c_is_n, c_sc = (float)((double)c_ga_n * (double)c_is_n), c_be = c_be_n;      // (bn_scale: the forward's rounding)
        wf2 zr[2] = {wf2{0.0f, 0.0f}, wf2{0.0f, 0.0f}};
        if constexpr (BNB2) {
          if (r < 3) load_c(r + 1);
#pragma unroll
          for (int ar = 0; ar < 2; ++ar) zr[ar] = tnv3_buf_load_f2(r_zl, lane_off_l, (unsigned)r * (unsigned)LHW * 4u + (unsigned)(ar * LW) * 4u);
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int ar = 0; ar < 2; ++ar) {
          float o[2];
          q2(wv[ar][0], wv[ar][1], wv[ar][2], wv[ar][3], wv[ar][4], o);
          tnv3_buf_store_f2(r_low, (DG & 32) ? kDmaOob : lane_off_l, (unsigned)r * (unsigned)LHW * 4u + (unsigned)(ar * LW) * 4u, wf2{o[0], o[1]});
          if constexpr (BNB2) {
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
              const float zc = zr[ar][b2] - c_mu;
              const float gk = fmaf(zc, c_sc, c_be) > 0.0f ? o[b2] : 0.0f;
              s1 += (double)gk;
              s2 += (double)gk * (double)(zc * c_is);
            }
          }
        }
        if constexpr (BNB2) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          if (tc == 0 && oh < H) {
            const long st_tile = ((long)wM.n * (H >> 2) + (oh >> 2)) * tilesW + wM.tcol;
            double* o = a.stats + ((size_t)(e_m0 + 4 * g + r) * ((size_t)a.N * (H >> 2) * tilesW) + st_tile) * 2;
            o[0] = s1;
            o[1] = s2;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (MODE == 1) {
      // upsampled half: Y = A'^T M' A' over the indices (0, 1, e, o, 5) with A'^T = [1 1 1 0 0; 0 1 0 1 0; 0 1 B 0 0; 0 1 0 B 1], B = kU43B; M'[i'][j'] =
      // acc[2 j' + i'] (i' = 0, 1), acc[10 + 2 j' + (i' - 2)] (i' = e, o), acc[20 + j'] (i' = 5).  Plain partial sums: the skip half's launch adds them.
      const unsigned lane_off_u = oh < H ? (unsigned)((4 * g) * HW + oh * W + ow) * 4u : kDmaOob;
      auto at5 = [](float q0, float q1, float qe, float qo, float q5, float (&o)[4]) {
        o[0] = (q0 + q1) + qe;
        o[1] = q1 + qo;
        o[2] = fmaf(kU43B, qe, q1);
        o[3] = fmaf(kU43B, qo, q1) + q5;
      };
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float wv[4][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float o[4];
          at5(acc[2 * j][r], acc[2 * j + 1][r], acc[10 + 2 * j][r], acc[11 + 2 * j][r], acc[20 + j][r], o);
          wv[0][j] = o[0]; wv[1][j] = o[1]; wv[2][j] = o[2]; wv[3][j] = o[3];
        }
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) {
          float o[4];
          at5(wv[ar][0], wv[ar][1], wv[ar][2], wv[ar][3], wv[ar][4], o);
          tnv3_buf_store_f4(r_dst, (DG & 32) ? kDmaOob : lane_off_u, (unsigned)r * (unsigned)HW * 4u + (unsigned)(ar * W) * 4u, f32x4{o[0], o[1], o[2], o[3]});
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    // STATS == 2 (round 6): a DATA-GRADIENT launch whose outputs are dA, the gradient at the activation of the previous Conv2DBlock, and which also
    // takes that block's two BatchNorm + ReLU backward sums (model.py:9-10) from the registers it writes dA from: bn_z = that block's raw
    // convolution output (read here like an addend: four 16-byte rows per channel), per channel mean / invstd / gamma / beta in a.mean /
    // a.scale / a.shift / a.bn_c4.  g = dA [BN(z) > 0] with the forward's own expression (bit-identical mask), xhat = (z - mean) invstd --
    // bn_relu_bwd_partial_kernel's arithmetic, element for element; only the order of the fp64 sums differs (per tile, then the tiles in order).
    constexpr bool BNB = STATS == 2;
    const tnv3_rsrc_t r_add = tnv3_make_rsrc(BNB ? a.bn_z + plane0 : (has_addend ? a.addend + plane0 : a.dst + plane0), planes_b);
    const unsigned lane_off_b = oh < H ? (unsigned)((4 * g) * HW + oh * W + ow) * 4u : kDmaOob;      // (a tile row below the image: loads give 0, stores are dropped)
    // MaxPool2d(2, 2) of the block as a second output (the down blocks' last layers): a lane's 4x4 pixels are 2x2 pooled ones
    constexpr bool has_pool = POOL != 0;               // (its own instantiation: the descriptor and the kept row would cost the plain kernel spills)
    const tnv3_rsrc_t r_pool = tnv3_make_rsrc(has_pool ? a.pool_dst + plane0 / 4 : a.dst + plane0, planes_b / 4);
    // The epilogue's options are wave-uniform run-time arguments.  Tested per row of four outputs they cost ~30 instructions of selects
    // between the variants (51 per row against the transform's 11: 920 of a 64-channel tile's 1960 vector instructions per thread,
    // 3.4 per MFMA in profiles/r04_infer_sq_summary.json), and one instantiated copy of the write-out per combination costs the kernels
    // 20-170 spilled registers.  So the arithmetic is unconditional on NEUTRAL constants instead: mean 0 / scale 1 / shift 0 without
    // BatchNorm, and the ReLU as an integer max with 0 or INT_MIN (one instruction; -0 and negative NaNs -> +0).
    {
      constexpr bool AFF = !STATS;
      int relu_floor = (!STATS && a.relu) ? 0 : (int)0x80000000;
      TNV3_OPAQUE_S(relu_floor);                         // (or the compiler turns the max back into max-with-0 + a select on a.relu)
      auto pool_store = [&](int ar, unsigned ch_b, const f32x4& v, wf2& pv) {      // (maxpool2x2_kernel's comparison order and NaN rule: bit-identical to the separate pass)
        auto mx = [](float m, float x) { return (x > m || x != x) ? x : m; };
        if ((ar & 1) == 0) pv = wf2{mx(v[0], v[1]), mx(v[2], v[3])};
        else {
          const unsigned pool_off_b = oh < H ? (lane_off_b >> 2) + (unsigned)ow : kDmaOob;      // bytes: ((4 g) HW / 4 + (oh / 2) (W / 2) + ow / 2) * 4 = (4 g) HW + oh W + 2 ow
          tnv3_buf_store_f2(r_pool, pool_off_b, ch_b / 4 + (unsigned)((ar >> 1) * (W >> 1)) * 4u, wf2{mx(mx(pv[0], v[0]), v[1]), mx(mx(pv[1], v[2]), v[3])});
        }
      };
      // per-channel constants: requested ONE channel ahead of their use (all four ahead of the loop held 12 registers; requested where they
      // are used, each channel waited a global-load latency for them)
      float mu_n = 0.0f, sc_n = 1.0f, sh_n = 0.0f;
      auto load_consts = [&](int r) {
        if (has_affine) {
          sc_n = a.scale[e_m0 + 4 * g + r];
          sh_n = a.shift[e_m0 + 4 * g + r];
          if (has_mean) mu_n = a.mean[e_m0 + 4 * g + r];
        }
      };
      if constexpr (AFF) load_consts(0);
      float b_mu_n = 0.0f, b_is_n = 0.0f, b_ga_n = 0.0f, b_be_n = 0.0f;      // STATS == 2: the previous block's BatchNorm constants, one channel ahead
      auto load_bn_consts = [&](int r) {
        const int ch = e_m0 + 4 * g + r;
        b_mu_n = a.mean[ch]; b_is_n = a.scale[ch]; b_ga_n = a.shift[ch]; b_be_n = a.bn_c4[ch];
      };
      if constexpr (BNB) load_bn_consts(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned ch_b = (unsigned)r * (unsigned)HW * 4u;
        const float mu = mu_n, sc = sc_n, sh = sh_n;
        if constexpr (AFF) {
          if (r < 3) load_consts(r + 1);
        }
        const float b_mu = b_mu_n, b_is = b_is_n, b_sc = (float)((double)b_ga_n * (double)b_is_n), b_be = b_be_n;      // (bn_scale: the forward's rounding)
        if constexpr (BNB) {
          if (r < 3) load_bn_consts(r + 1);
        }
        f32x4 ad[4];
        if (BNB || has_addend) {
#pragma unroll
          for (int ar = 0; ar < 4; ++ar) ad[ar] = tnv3_buf_load_f4(r_add, lane_off_b, ch_b + (unsigned)(ar * W) * 4u);
        }
        float wv[4][6];                                    // W[a][j] = sum_i A^T[a][i] M[i][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          float o[4];
          wino43s_at6(acc[j][r], acc[6 + j][r], acc[12 + j][r], acc[18 + j][r], acc[24 + j][r], acc[30 + j][r], o);
          wv[0][j] = o[0]; wv[1][j] = o[1]; wv[2][j] = o[2]; wv[3][j] = o[3];
        }
        double s1 = 0.0, s2 = 0.0;
        wf2 pv = {0.0f, 0.0f};
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) {
          float o[4];
          wino43s_at6(wv[ar][0], wv[ar][1], wv[ar][2], wv[ar][3], wv[ar][4], wv[ar][5], o);
          f32x4 v = {o[0], o[1], o[2], o[3]};
          if (!BNB && has_addend) {
            TNV3_NO_IF_CONVERSION();                     // a real scalar branch: as a select the four adds cost four more instructions per row
            v += ad[ar];
          }
          if constexpr (AFF) {
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) v[b2] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, (v[b2] - mu) * sc + sh), relu_floor));
          }
          tnv3_buf_store_f4(r_dst, (DG & 32) ? kDmaOob : lane_off_b, ch_b + (unsigned)(ar * W) * 4u, v);
          if (has_pool) pool_store(ar, ch_b, v, pv);
          if constexpr (STATS == 1) {
            s1 += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
            s2 += ((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1]) + ((double)v[2] * (double)v[2] + (double)v[3] * (double)v[3]);
          }
          if constexpr (BNB) {
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) {
              const float zc = ad[ar][b2] - b_mu;
              const float gk = fmaf(zc, b_sc, b_be) > 0.0f ? v[b2] : 0.0f;
              s1 += (double)gk;
              s2 += (double)gk * (double)(zc * b_is);
            }
          }
        }
        if constexpr (STATS) {
          // BatchNorm batch statistics (model.py:9 in training mode) from the epilogue's registers: the 16 lanes of a lane group hold the 16
          // tile columns of channel 4 g + r.  Butterflies over lane bits 3..0 in a fixed order, fp64: deterministic.  One statistics tile =
          // 4 x 64 pixels (this wave's tile row).
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          if (tc == 0 && oh < H) {
            const long st_tile = ((long)wM.n * (H >> 2) + (oh >> 2)) * tilesW + wM.tcol;
            double* o = a.stats + ((size_t)(e_m0 + 4 * g + r) * ((size_t)a.N * (H >> 2) * tilesW) + st_tile) * 2;
            o[0] = s1;
            o[1] = s2;
          }
        }
        __builtin_amdgcn_sched_barrier(0);                // one channel at a time
      }
    }
    }
  };

  for (;;) {                                             // one pass per workgroup tile
    step(std::true_type{});
    ++kM;
    for (; kM < nChunks; ++kM) step(std::false_type{});
    tl_stamp(1);
    writeout();
    __builtin_amdgcn_sched_barrier(0);
    tl_stamp(2);
    if constexpr (TL != 0) ++tl_tiles;
    kM = 0;
    wM.next();
    if (!wM.valid) break;
  }
  };
  if (swave >> 2) body(std::integral_constant<int, 1>{}); else body(std::integral_constant<int, 0>{});
  if constexpr (TL != 0) {
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.stats) + wave * 8;
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = tl_acc[i];
      o[3] = (unsigned long long)tl_steps;
      o[4] = (unsigned long long)tl_tiles;
      o[5] = o[6] = o[7] = 0;
    }
  }
}

}  // namespace tnv3
