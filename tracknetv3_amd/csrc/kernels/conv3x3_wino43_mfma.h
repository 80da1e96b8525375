// conv3x3_wino43_mfma.h -- the plain 3x3 'same' convolution in fused Winograd F(4x4, 3x3) form (tnv3_conv3x3_wino43_forward).
//
// F(2x2, 3x3) (conv3x3_wino*_mfma.h) multiplies 16 transform coefficients per 2x2 output tile -- 4 per output pixel instead of the
// direct form's 9; F(4x4, 3x3) multiplies 36 per 4x4 tile -- 2.25 per pixel, 1.78x fewer MFMAs again.  Its transforms are no longer
// adds only (B^T has the entries 4, -5, 2; G has 1/4, 1/6, 1/24; A^T has 2, 4, 8), which costs accuracy: on the whole TrackNet(27 -> 8)
// forward at 288 x 512 the heat maps deviate from the fp64 forward by 3.1e-6 (F(2x2): 1.4e-6, direct fp32: see
// profiles/r03_wino_f43_precision.json; the path's bar is 1e-4).  Same function as tnv3_conv3x3_wino_forward up to that rounding.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          d: 6x6 input patch (stride 4), Y: 4x4 outputs, 36 products per (co, ci, tile)
//
// Mapping.  Workgroup tile = 64 output channels x 32 tiles (2 tile rows x 16 tile columns = 8 x 64 pixels), 512 threads.  The 36
// transform coefficients xi = (i, j) are dealt in four 3x3 blocks xg = (i / 3, j / 3); wave = (xg, 32-channel block wm) keeps the nine
// xi of its block for 32 channels x 32 tiles: 144 accumulator registers (MFMA 32x32x2: M = channels, N = tiles, K = input channel pairs).
//   * A operand (U = G g G^T) straight from L2 into registers, as in conv3x3_wino6_mfma.h: the panel is packed in lane order
//     [32-channel block][chunk of 8 ci][xg][x9][lane = (k half, channel)][4 K steps]: nine 16-byte loads per wave and chunk, each
//     quad re-loaded in place behind the MFMAs that consumed it.
//   * B operand V = B^T d B in LDS as V[xg][x9][tile][12: (k half) x 4 K steps + pad]: ONE ds_read_b128 feeds the four K steps of a xi
//     (tile stride 12 floats: conflict-free in the hardware's 16-lane groups); two stages (chunk k is consumed while k+1 is made).
//   * no roles (wgrad_wino5_mfma_kernel's lesson): every wave streams its 36 MFMAs of chunk k and, between them, transforms HALF a
//     patch of chunk k+1 -- thread = (channel, tile, row half rh): three rows of B^T d B from five raw rows, 72 fused multiply-adds,
//     15 LDS reads, 18 stores -- and issues its LDS-DMA pieces of chunk k+2.
//   * write-out: every wave turns its 3x3 block of M into its PARTIAL 4x4 output (A^T[:, I] M_IJ A[J, :]), the four xg waves of a channel
//     block meet through LDS (in four rounds of four channels), wave xg finishes output row xg of every tile: 16-byte stores.
// Needs Cout % 64 == 0, H % 4 == 0, W % 64 == 0.  Not bit-identical to the F(2x2) kernels (another factorisation); deterministic.
#pragma once
#include <type_traits>
#include "conv3x3_wino3_mfma.h"

namespace tnv3 {

struct Wino43Cfg {
  static constexpr int CC = 8, NT = 512, MB = 64, TB = 32;
  static constexpr int TH = 8, TW = 64;                    // pixels per workgroup tile
  static constexpr int RROWS = 10, RW = 80;                 // raw halo tile per channel: rows h0-1 .. h0+8; 80 floats per row = 20 pieces, the
                                                            // first 18 = columns w0-4 .. w0+67 (row stride 80: the transform's 16-byte reads
                                                            // of tile rows 0 / 1 then fall into disjoint bank ranges)
  static constexpr int RAWP = RROWS * RW;                   // 800 floats per channel
  static constexpr int RAW_STAGE = CC * RAWP;               // 6400 floats = 1600 pieces: three per thread + one more for wave 0
  static constexpr int VT = 12;                             // floats per (xi, tile): [k half][4 K steps] + 4 pad
  static constexpr int V_STAGE = 36 * TB * VT;              // 13824 floats
  static constexpr int LDS_FLOATS = 2 * V_STAGE + 2 * RAW_STAGE;
  static constexpr int A_CHUNK_FLOATS = 4 * 9 * 64 * 4;     // one (32-channel block, chunk) of the panel: [xg][x9][lane][4]
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

inline size_t conv3x3_wino43_packed_floats(int cin, int cout) {
  if (cin <= 0 || cout <= 0 || cout % 32) return 0;
  return (size_t)(cout / 32) * ((cin + Wino43Cfg::CC - 1) / Wino43Cfg::CC) * Wino43Cfg::A_CHUNK_FLOATS + kPackZeroTail;
}

// Interpolation points (0, +-s, +-2s, inf) with s = 3/4 instead of Lavin's s = 1.  Scaling the points changes nothing but the
// constants of the transforms (the input transform keeps its instruction count, the output transform of the (0, +-s) half gains two
// multiplies), yet it balances the magnitudes the transforms mix: fp32 error of a 64-channel layer 2.95e-7 -> 1.42e-7 rms
// (4.97e-6 -> 1.46e-6 max) of the output scale, the network's eval heat maps 3.1e-6 -> 1.6e-6 from the fp64 forward (direct fp32:
// 1.1e-6), training-mode heat maps at 288x512 9.1e-5 -> 3.9e-5 (profiles/r03_wino_f43_precision.json; a scan of symmetric pairs
// (a, b) puts the optimum at a = 0.65-0.75, b = 1.4-1.6 -- (3/4, 3/2) is within 7 % of it and every constant below is an exact
// binary fraction).  kW43S = 1 gives Lavin's points back (with the columns of +-2 in A^T halved and their rows in G doubled).
constexpr float kW43S = 0.75f;
constexpr float kW43S2 = kW43S * kW43S, kW43S3 = kW43S2 * kW43S, kW43S4 = kW43S2 * kW43S2;
// B^T (monic rows, the Toom-Cook construction): [4s^4 0 -5s^2 0 1 0; 0 -4s^3 -4s^2 s 1 0; 0 4s^3 -4s^2 -s 1 0; 0 -2s^3 -s^2 2s 1 0;
//                                                0 2s^3 -s^2 -2s 1 0; 0 4s^4 0 -5s^2 0 1]
constexpr float kW43_4S4 = 4.0f * kW43S4, kW43_5S2 = 5.0f * kW43S2, kW43_4S2 = 4.0f * kW43S2, kW43_2S = 2.0f * kW43S;

// G: row of point p = |p| [1 p p^2] / prod_{q != p} (p - q) over the finite points (the factor |p|: A^T's column is divided by it; the row
// of 0 has none), the row of infinity [0 0 1].  Row i applied to (g0, g1, g2):
__device__ __forceinline__ float wino43_g_row(int i, float g0, float g1, float g2) {
  constexpr float n0 = 1.0f / (4.0f * kW43S4), n1 = -kW43S / (6.0f * kW43S4), n2 = kW43_2S / (24.0f * kW43S4);
  switch (i) {
    case 0: return n0 * g0;
    case 1: return n1 * ((g0 + kW43S * g1) + kW43S2 * g2);
    case 2: return n1 * ((g0 - kW43S * g1) + kW43S2 * g2);
    case 3: return n2 * g0 + ((n2 * kW43_2S) * g1 + (n2 * kW43_4S2) * g2);
    case 4: return n2 * g0 + ((-n2 * kW43_2S) * g1 + (n2 * kW43_4S2) * g2);
    default: return g2;
  }
}

// w[..][3][3] (strides s_co / s_ci floats between output / input channels; flip: taps reversed -- the data gradient's filter) ->
// panel u[co / 32][chunk][xg][x9][lane = (ci % 2) * 32 + co % 32][ci % 8 / 2], zero for ci >= Cin; then kPackZeroTail zeros.
// One work item = one (32-channel block, chunk, lane): it reads the 3x3 filters of the lane's four input channels ONCE, forms all 36
// G g G^T entries of each and writes 36 float4 -- a wave's 64 lanes store 1 KB contiguous per (xg, x9).  (The first form took one item
// per output float: 36 x the filter reads, three 64-bit divisions and two run-time switches each -- 0.79 ms for the ~40 panels of a
// training step, profiles/r03_train_kernel_stats.csv; shared by the one-panel kernel and the table-driven one below.)
inline long conv3x3_wino43_pack_items(int Cout, int Cin) { return (long)(Cout / 32) * ((Cin + 7) / 8) * 64 + kPackZeroTail / 4; }
__device__ __forceinline__ void conv3x3_wino43_pack_elements(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, long s_co, long s_ci,
                                                             int flip, long q0, long stride) {
  static_assert(kPackZeroTail % 4 == 0, "the zero tail is written as float4");
  const int nch = (Cin + 7) / 8;
  const long quads = (long)(Cout / 32) * nch * 64;
  f32x4* u4 = reinterpret_cast<f32x4*>(u);
  for (long q = q0; q < quads + kPackZeroTail / 4; q += stride) {
    if (q >= quads) { u4[quads * 36 + (q - quads)] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; continue; }
    const int ln = (int)(q & 63);
    const int r = (int)(q >> 6), k = r % nch, mb32 = r / nch;
    const int co = 32 * mb32 + (ln & 31);
    float rowv[4][6][3];                               // [channel s][row i of G applied down the filter's columns][column c]
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
      const int ci = 8 * k + 2 * sx + (ln >> 5);
      float f[9];
      if (ci < Cin) {
        const float* g = w + (long)co * s_co + (long)ci * s_ci;
#pragma unroll
        for (int t = 0; t < 9; ++t) f[t] = flip ? g[8 - t] : g[t];
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) f[t] = 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) rowv[sx][i][c] = wino43_g_row(i, f[c], f[3 + c], f[6 + c]);
    }
    f32x4* dst = u4 + ((long)r * 36) * 64 + ln;        // (xg, x9) planes of 64 float4
#pragma unroll
    for (int xg = 0; xg < 4; ++xg)
#pragma unroll
      for (int x9 = 0; x9 < 9; ++x9) {
        const int i = 3 * (xg >> 1) + x9 / 3, j = 3 * (xg & 1) + x9 % 3;
        f32x4 v;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) v[sx] = wino43_g_row(j, rowv[sx][i][0], rowv[sx][i][1], rowv[sx][i][2]);
        dst[(xg * 9 + x9) * 64] = v;
      }
  }
}
#ifdef TNV3_DIAG      // (a non-template kernel is emitted wherever this header is included: the twin stays out of the product objects)
inline __global__ void __launch_bounds__(256) conv3x3_wino43_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin,
                                                                        long s_co, long s_ci, int flip) {
  conv3x3_wino43_pack_elements(w, u, Cout, Cin, s_co, s_ci, flip, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}
#endif
// One 1-D input transform B^T applied to six values, the three outputs of one half: rows 0..2 (RH = 0) or 3..5 (RH = 1).
//   (B^T above; Lavin's for s = 1: [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1])
template <int RH>
__device__ __forceinline__ void wino43_bt_half(const float (&d)[6], float (&t)[3]) {
  if (RH == 0) {
    t[0] = fmaf(kW43_4S4, d[0], fmaf(-kW43_5S2, d[2], d[4]));
    const float a = fmaf(-kW43_4S2, d[2], d[4]), b = fmaf(-kW43_4S2, d[1], d[3]);
    t[1] = fmaf(kW43S, b, a);
    t[2] = fmaf(-kW43S, b, a);
  } else {
    const float c = fmaf(-kW43S2, d[2], d[4]), e = fmaf(-kW43S2, d[1], d[3]);
    t[0] = fmaf(kW43_2S, e, c);
    t[1] = fmaf(-kW43_2S, e, c);
    t[2] = fmaf(kW43_4S4, d[1], fmaf(-kW43_5S2, d[3], d[5]));
  }
}
// ... and all six outputs (the second pass along a row of the half-transformed patch)
__device__ __forceinline__ void wino43_bt_full(const float (&d)[6], float (&t)[6]) {
  t[0] = fmaf(kW43_4S4, d[0], fmaf(-kW43_5S2, d[2], d[4]));
  const float a = fmaf(-kW43_4S2, d[2], d[4]), b = fmaf(-kW43_4S2, d[1], d[3]);
  t[1] = fmaf(kW43S, b, a);
  t[2] = fmaf(-kW43S, b, a);
  const float c = fmaf(-kW43S2, d[2], d[4]), e = fmaf(-kW43S2, d[1], d[3]);
  t[3] = fmaf(kW43_2S, e, c);
  t[4] = fmaf(-kW43_2S, e, c);
  t[5] = fmaf(kW43_4S4, d[1], fmaf(-kW43_5S2, d[3], d[5]));
}

// Output transform: three M values of one half (J = 0: coefficients 0..2, J = 1: 3..5) -> their contribution to the four outputs.
//   A^T = [1 1/s 1/s 1/(2s) 1/(2s) 0; 0 1 -1 1 -1 0; 0 s s 2s 2s 0; 0 s^2 -s^2 4s^2 -4s^2 1]: column of point p = [1 p p^2 p^3] / |p| (G's row
//   carries the |p|); Lavin's for s = 1 is the same with the columns of +-2 unscaled: [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
template <int J>
__device__ __forceinline__ void wino43_at_half(float m0, float m1, float m2, float (&o)[4]) {
  if (J == 0) {
    const float p = m1 + m2, q = m1 - m2;
    o[0] = fmaf(1.0f / kW43S, p, m0); o[1] = q; o[2] = kW43S * p; o[3] = kW43S2 * q;
  } else {
    const float p = m0 + m1, q = m0 - m1;
    o[0] = (0.5f / kW43S) * p; o[1] = q; o[2] = kW43_2S * p; o[3] = fmaf(kW43_4S2, q, m2);
  }
}

typedef float wf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wf2 wino43_fma2(float k, wf2 x, wf2 y) { return __builtin_elementwise_fma(wf2{k, k}, x, y); }      // v_pk_fma_f32
// The 1-D input transforms on two lines at once (two patch columns in the first pass, two rows in the second): packed fp32 instructions
template <int RH>
__device__ __forceinline__ void wino43_bt_half2(const wf2 (&d)[6], wf2 (&t)[3]) {
  if (RH == 0) {
    t[0] = wino43_fma2(kW43_4S4, d[0], wino43_fma2(-kW43_5S2, d[2], d[4]));
    const wf2 a = wino43_fma2(-kW43_4S2, d[2], d[4]), b = wino43_fma2(-kW43_4S2, d[1], d[3]);
    t[1] = wino43_fma2(kW43S, b, a);
    t[2] = wino43_fma2(-kW43S, b, a);
  } else {
    const wf2 c = wino43_fma2(-kW43S2, d[2], d[4]), e = wino43_fma2(-kW43S2, d[1], d[3]);
    t[0] = wino43_fma2(kW43_2S, e, c);
    t[1] = wino43_fma2(-kW43_2S, e, c);
    t[2] = wino43_fma2(kW43_4S4, d[1], wino43_fma2(-kW43_5S2, d[3], d[5]));
  }
}
__device__ __forceinline__ void wino43_bt_full2(const wf2 (&d)[6], wf2 (&t)[6]) {
  t[0] = wino43_fma2(kW43_4S4, d[0], wino43_fma2(-kW43_5S2, d[2], d[4]));
  const wf2 a = wino43_fma2(-kW43_4S2, d[2], d[4]), b = wino43_fma2(-kW43_4S2, d[1], d[3]);
  t[1] = wino43_fma2(kW43S, b, a);
  t[2] = wino43_fma2(-kW43S, b, a);
  const wf2 c = wino43_fma2(-kW43S2, d[2], d[4]), e = wino43_fma2(-kW43S2, d[1], d[3]);
  t[3] = wino43_fma2(kW43_2S, e, c);
  t[4] = wino43_fma2(-kW43_2S, e, c);
  t[5] = wino43_fma2(kW43_4S4, d[1], wino43_fma2(-kW43_5S2, d[3], d[5]));
}
// ... on a PAIR of channels at once (two-wide vectors: packed fp32 instructions)
template <int J>
__device__ __forceinline__ void wino43_at_half2(wf2 m0, wf2 m1, wf2 m2, wf2 (&o)[4]) {
  if (J == 0) {
    const wf2 p = m1 + m2, q = m1 - m2;
    o[0] = wino43_fma2(1.0f / kW43S, p, m0); o[1] = q; o[2] = kW43S * p; o[3] = kW43S2 * q;
  } else {
    const wf2 p = m0 + m1, q = m0 - m1;
    o[0] = (0.5f / kW43S) * p; o[1] = q; o[2] = kW43_2S * p;
    o[3] = wino43_fma2(kW43_4S2, q, m2);
  }
}

// TL = 1 (libtnv3_diag.so only): s_memtime totals per phase of one mid-grid workgroup, written as [wave 8][8] uint64 to a.stats
// (tile fill: DMA wait + first transform + two barriers; chunk loop; next tile's offsets + raw issue; write-out; A issue + loop tail;
// chunks; tiles) -- results stay correct.  TL = 2: the same with the output stores out of range (dropped by the descriptor's bounds
// check), 3: without the LDS exchange of the write-out (no ds_write / ds_read of partial sums; barriers and arithmetic kept), 4: both --
// WRONG results: what the write-out's 15-19 k cycles per tile are made of.
template <int EARLY, int PACKED_T = 0, int STATS = 0, int TL = 0>
__global__ void __launch_bounds__(Wino43Cfg::NT) conv3x3_wino43_kernel(const WinoArgs a) {
  using Cfg = Wino43Cfg;
  constexpr int CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TB = Cfg::TB, RW = Cfg::RW, RAWP = Cfg::RAWP, VT = Cfg::VT;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* v_s = lds;                                   // two stages each
  float* raw_s = lds + 2 * Cfg::V_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int xg = swave & 3, wm = swave >> 2;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = (H + Cfg::TH - 1) / Cfg::TH, tilesW = W / Cfg::TW;      // H % 8 == 4: the last tile row's lower half is outside the image
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (Cin + CC - 1) / CC;

  ConvTileWalk walk;
  walk.init(blockIdx.x, gridDim.x, nMB, nPT, tilesH, tilesW);
  if (!walk.valid) return;
  unsigned long long tl_acc[5] = {0, 0, 0, 0, 0}, tl_last = 0;
  int tl_chunks = 0, tl_tiles = 0;
  auto tl_stamp = [&](int slot) {
    if constexpr (TL != 0) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tl_acc[slot] += now - tl_last;
      tl_last = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);      // scalar: the LDS-DMA destination (M0) stays on the SALU
  const size_t x_step = (size_t)CC * HW;
  // ---- transform roles: thread = (channel t_c, tile t_t, row half rh); rh is wave-uniform (waves 0-3 / 4-7)
  const int rh = swave >> 2;
  // A half-wave (the lane group of a 4-byte LDS store: 32 banks) takes all 32 tiles, each with ONE of the chunk's channels, rotated by the
  // tile: position j = c >> 1 of the tile's 16-byte quad = (t + t / 8 + half-wave) mod 4, so the 32 stores of an instruction hit 8 quads x
  // 4 positions = 32 different banks (a fixed channel per half-wave put them on 8: 4-way conflicts, 54 % of the LDS cycles,
  // profiles/r03_infer_sq_summary.json).  The 16-byte raw reads stay conflict-free: the channel's plane offset is 32 (c & 1) banks, the same for the whole half-wave.
  const int tp = tid & 255, t_g = tp >> 5, t_t = tp & 31, t_tr = t_t >> 4, t_tc = t_t & 15;
  const int t_c = 2 * (((t_t & 3) + (t_t >> 3) + t_g) & 3) + (t_g >> 2);
  const int t_src = t_c * RAWP + (4 * t_tr) * RW + 4 * t_tc;        // + row * RW: 16-byte aligned
  const int t_dst = t_t * VT + (t_c & 1) * 4 + (t_c >> 1);           // + (xg * 9 + x9) * TB * VT
  // ---- MFMA operands
  const int b_off = (xg * 9) * TB * VT + bl * VT + half * 4;        // + x9 * TB * VT
  const unsigned a_lane_b = (unsigned)lane * 16u;
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;

  // Everything below is instantiated twice, for the wave's row half RH (waves 0-3 / 4-7), and selected by ONE scalar branch around
  // the whole tile loop: the transform's code differs per half, and a branch inside the chunk loop would put the accumulators through
  // phi copies between the two arms.
  auto body = [&](auto rhc) {
  constexpr int RH = decltype(rhc)::value;
  f32x4 av[9];
  f32x16 acc[9];
  // ---- per-tile state of the loaders: raw LDS-DMA offsets (slot e = tid + i * 512 -> (channel, row, piece) of [CC][10][20]; pieces 18,
  //      19 of a row and everything outside the image read out of range = 0), the image's input planes, this wave's panel slice
  unsigned vo[4];
  const float* xp;
  const float* a_tile;
  auto set_tile = [&](int t_n, int t_h0, int t_w0, int t_m0) {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);                                  // recomputed per tile; nothing of it stays live across the chunk loop
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t_op + i * NT;
      const int c = e / 200, rem = e - c * 200;
      const int r = rem / 20, q = rem - r * 20;
      const int gh = t_h0 - 1 + r, gw = t_w0 - 4 + 4 * q;
      const bool ok = e < 1600 && q < 18 && gh >= 0 && gh < H && gw >= 0 && gw < W;
      vo[i] = ok ? (unsigned)(c * HW + gh * W + gw) * 4u : kDmaOob;
    }
    xp = a.src + (size_t)t_n * Cin * HW;
    a_tile = a.u + ((size_t)(t_m0 / 32 + wm) * nChunks) * Cfg::A_CHUNK_FLOATS + xg * (9 * 256);
  };
  auto dma_chunk = [&](int k, int stage) {              // chunk k of the loaders' tile -> raw stage
    const int cvalid = Cin - k * CC;
    const tnv3_rsrc_t rr = tnv3_make_rsrc(xp + (size_t)k * x_step, (unsigned)(cvalid < CC ? cvalid : CC) * (unsigned)HW * 4u);
    float* rs = raw_s + stage * Cfg::RAW_STAGE;
#pragma unroll
    for (int i = 0; i < 3; ++i) tnv3_buf_dma16(rr, rs + (i * NT + wbase) * 4, vo[i]);
    if (swave == 0) tnv3_buf_dma16(rr, rs + (3 * NT + wbase) * 4, vo[3]);       // slots 1536 .. 1599
  };
  auto load_a = [&](int k, int x9) {
    const tnv3_rsrc_t ra = tnv3_make_rsrc(a_tile + (size_t)k * Cfg::A_CHUNK_FLOATS, 9u * 1024u);
    av[x9] = tnv3_buf_load_f4(ra, a_lane_b, (unsigned)x9 * 1024u);
  };
  // pipeline fill of the loaders' tile: raw tiles of chunks 0 and 1 (HBM / L2 latency: requested BEFORE the previous tile's write-out) ...
  auto issue_raw = [&]() {
    dma_chunk(0, 0);
    if (nChunks > 1) dma_chunk(1, 1);
  };
  // ... and the A quads of chunk 0 (L2 hits; their 36 registers would be live through the write-out: requested after it)
  auto issue_a = [&]() {
#pragma unroll
    for (int x = 0; x < 9; ++x) load_a(0, x);
  };
  set_tile(walk.n, walk.trow * Cfg::TH, walk.tcol * Cfg::TW, walk.mb * MB);
  issue_raw();
  issue_a();
  if constexpr (TL != 0) tl_last = __builtin_amdgcn_s_memtime();
  for (;;) {                                            // one pass per workgroup tile
    const int e_n = walk.n, e_h0 = walk.trow * Cfg::TH, e_w0 = walk.tcol * Cfg::TW, e_m0 = walk.mb * MB, e_pt = walk.pt;
    // ---- the half-patch transform of one chunk (raw stage -> V stage of the same parity), in pieces that the chunk loop places between
    //      its MFMA groups: rows 3 RH .. 3 RH + 2 of B^T d B from raw rows RH .. RH + 4 of the patch
    float t[3][6];
    auto column = [&](int c, float x0, float x1, float x2, float x3, float x4) {      // the 1-D transform down patch column c
      float col[6], o[3];
      if (RH == 0) { col[0] = x0; col[1] = x1; col[2] = x2; col[3] = x3; col[4] = x4; col[5] = 0.0f; }
      else { col[0] = 0.0f; col[1] = x0; col[2] = x1; col[3] = x2; col[4] = x3; col[5] = x4; }
      wino43_bt_half<RH>(col, o);
      t[0][c] = o[0]; t[1][c] = o[1]; t[2][c] = o[2];
    };
    auto t_read_pair = [&](int stage, int c, wf2 (&q)[5]) {      // patch columns c, c + 1 (c = 1 or 3): one 8-byte read per raw row
      const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src + RH * RW;
#pragma unroll
      for (int r = 0; r < 5; ++r) q[r] = *reinterpret_cast<const wf2*>(d + r * RW + 3 + c);
    };
    auto column2 = [&](int ca, int cb, wf2 x0, wf2 x1, wf2 x2, wf2 x3, wf2 x4) {      // two patch columns at once
      wf2 col[6], o[3];
      const wf2 z = {0.0f, 0.0f};
      if (RH == 0) { col[0] = x0; col[1] = x1; col[2] = x2; col[3] = x3; col[4] = x4; col[5] = z; }
      else { col[0] = z; col[1] = x0; col[2] = x1; col[3] = x2; col[4] = x3; col[5] = x4; }
      wino43_bt_half2<RH>(col, o);
#pragma unroll
      for (int k = 0; k < 3; ++k) { t[k][ca] = o[k][0]; t[k][cb] = o[k][1]; }
    };
    auto t_cols_pair = [&](int c, const wf2 (&q)[5]) { column2(c, c + 1, q[0], q[1], q[2], q[3], q[4]); };
    // (the scalar form: patch columns 1..4 as one 16-byte read per raw row, four 1-D transforms)
    auto t_read_mid = [&](int stage, f32x4 (&q)[5]) {
      const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src + RH * RW;
#pragma unroll
      for (int r = 0; r < 5; ++r) q[r] = *reinterpret_cast<const f32x4*>(d + r * RW + 4);
    };
    auto t_cols_mid = [&](const f32x4 (&q)[5]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) column(1 + c, q[0][c], q[1][c], q[2][c], q[3][c], q[4][c]);
    };
    auto t_read_edge = [&](int stage, float (&e0)[5], float (&e5)[5]) {      // patch columns 0 and 5
      const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src + RH * RW;
#pragma unroll
      // (4-byte reads at a stride of four floats: 8 of the 32 banks, 4-way conflicts.  As 8-byte reads of (column -1, 0) and (5, 6) --
      // 64 banks, 2-way -- the kernel measured 1-3 % SLOWER on every shape: twice the return data for the same two values.)
      for (int r = 0; r < 5; ++r) { e0[r] = d[r * RW + 3]; e5[r] = d[r * RW + 8]; }
    };
    auto t_cols_edge = [&](const float (&e0)[5], const float (&e5)[5]) {
      if (PACKED_T) {
        column2(0, 5, wf2{e0[0], e5[0]}, wf2{e0[1], e5[1]}, wf2{e0[2], e5[2]}, wf2{e0[3], e5[3]}, wf2{e0[4], e5[4]});
      } else {
        column(0, e0[0], e0[1], e0[2], e0[3], e0[4]);
        column(5, e5[0], e5[1], e5[2], e5[3], e5[4]);
      }
    };
    auto t_rows01 = [&](int stage) {                     // second pass along rows 0 and 1 at once
      float* vdst = v_s + stage * Cfg::V_STAGE + t_dst;
      wf2 d[6], o[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) d[c] = wf2{t[0][c], t[1][c]};
      wino43_bt_full2(d, o);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        vdst[((2 * RH + j / 3) * 9 + j % 3) * (TB * VT)] = o[j][0];
        vdst[((2 * RH + j / 3) * 9 + 3 + j % 3) * (TB * VT)] = o[j][1];
      }
    };
    auto t_row = [&](int stage, int r) {                 // second pass along row r, results straight into V: xi = (3 RH + r, j)
      float* vdst = v_s + stage * Cfg::V_STAGE + t_dst;
      float o[6];
      wino43_bt_full(t[r], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) vdst[((2 * RH + j / 3) * 9 + 3 * r + j % 3) * (TB * VT)] = o[j];      // block (RH, j / 3), x9 = 3 r + j % 3
    };
    auto transform_all = [&](int stage) {
      f32x4 q[5];
      float e0[5], e5[5];
      t_read_mid(stage, q);
      t_cols_mid(q);
      t_read_edge(stage, e0, e5);
      t_cols_edge(e0, e5);
#pragma unroll
      for (int r = 0; r < 3; ++r) t_row(stage, r);
    };
    auto full_barrier = [&]() {
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
      __builtin_amdgcn_s_barrier();
    };

    // ---- V of chunk 0 (its raw tiles and the A quads were requested before the previous tile's write-out)
    full_barrier();
    transform_all(0);
    full_barrier();
    tl_stamp(0);

    // One chunk: the 36 MFMAs of chunk k (V stage k & 1) in five xi groups, and between the groups the half-patch transform of chunk
    // k+1 (raw stage -> V stage of the other parity; after the last chunk: of a stale raw stage into a V stage nobody reads --
    // unconditional, so that the stream stays straight-line code); the raw tile of chunk k+2 goes into raw stage k & 1, whose rows were
    // consumed one chunk ago.
    auto chunk_body = [&](int k, auto first_c) {
      constexpr bool FIRST = decltype(first_c)::value;    // the tile's first chunk starts the accumulators from zero
      const int sc = k & 1, sn = sc ^ 1;
      const float* B = v_s + sc * Cfg::V_STAGE + b_off;
      const int knext = k + 1 < nChunks ? k + 1 : k;      // (the last chunk re-loads its own A: unconditional loads)
      f32x4 bq[4];                                      // ring: the quads of xi pair p in bq[2 (p & 1)], bq[2 (p & 1) + 1]
      auto read_pair = [&](int p) {
        bq[2 * (p & 1)] = *reinterpret_cast<const f32x4*>(B + (2 * p) * (TB * VT));
        if (p < 4) bq[2 * (p & 1) + 1] = *reinterpret_cast<const f32x4*>(B + (2 * p + 1) * (TB * VT));
      };
      auto mfma_pair = [&](int p) {
        const int x0 = 2 * p, x1 = p < 4 ? 2 * p + 1 : -1;
        if (p + 1 < 5) read_pair(p + 1);                 // fenced: the next pair's reads go out before this pair's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[x0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x0][s], bq[2 * (p & 1)][s], FIRST && s == 0 ? zero16 : acc[x0], 0, 0, 0);
          if (x1 >= 0) acc[x1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x1][s], bq[2 * (p & 1) + 1][s], FIRST && s == 0 ? zero16 : acc[x1], 0, 0, 0);
        }
        load_a(knext, x0);
        if (x1 >= 0) load_a(knext, x1);
        __builtin_amdgcn_sched_barrier(0);
      };
      float e0[5], e5[5];
      read_pair(0);
      if (k + 2 < nChunks) dma_chunk(k + 2, sc);
      if constexpr (PACKED_T) {                           // (8-byte reads, two-wide arithmetic: measured 2-4 % slower on the deep layers)
        wf2 q[5];
        t_read_pair(sn, 1, q);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(0);
        t_cols_pair(1, q);
        t_read_pair(sn, 3, q);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(1);
        t_cols_pair(3, q);
        t_read_edge(sn, e0, e5);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(2);
        t_cols_edge(e0, e5);
        t_rows01(sn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(3);
        t_row(sn, 2);
      } else {
        f32x4 q[5];
        t_read_mid(sn, q);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(0);
        t_cols_mid(q);
        t_read_edge(sn, e0, e5);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(1);
        t_cols_edge(e0, e5);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(2);
        t_row(sn, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(3);
        t_row(sn, 1);
        t_row(sn, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair(4);
      // this wave's raw pieces of chunk k+2 have landed (they are OLDER than the nine A loads issued since), its V stores are done
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(9));
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
      __builtin_amdgcn_s_barrier();
    };
    chunk_body(0, std::true_type{});
    for (int k = 1; k < nChunks; ++k) chunk_body(k, std::false_type{});
    tl_stamp(1);
    if constexpr (TL != 0) { tl_chunks += nChunks; ++tl_tiles; }

    // ---- the NEXT tile's pipeline fill goes out now: the raw stages and the A registers are free, and the write-out below (which
    //      exchanges through the V stages) hides the latency of these loads
    walk.next();
    const bool have_next = walk.valid;
    if (EARLY && have_next) {
      set_tile(walk.n, walk.trow * Cfg::TH, walk.tcol * Cfg::TW, walk.mb * MB);
      issue_raw();
    }
    tl_stamp(2);

    // ---- write-out.  This wave's partial 4x4 output per (channel r of the lane, tile bl): P = A^T[:, I] M_IJ A[J, :] with
    //      I = rows 3 (xg >> 1) .., J = columns 3 (xg & 1) ..; the four xg waves of a channel block exchange through LDS, wave xg
    //      finishes output row xg.  Four rounds of four channels r (the exchange of a round: 8 waves x 3 rows x 4 r x 16 bytes x 64
    //      lanes = 96 KB, in the V stages); the arithmetic runs on channel PAIRS (r, r + 1: adjacent accumulator registers) as
    //      two-wide vectors -- v_pk_add_f32 / v_pk_fma_f32, half the vector instructions.
    float* xch = lds;                                    // [dst wave 8][src slot 3][pair 2][column half 2][lane 64][4]
    // (the statistics variant writes the raw convolution: the host refuses affine / ReLU arguments with it, so their code is not compiled in)
    const bool has_affine = !STATS && a.scale != nullptr, has_mean = !STATS && a.mean != nullptr, has_addend = a.addend != nullptr;
    int ln_w = lane;
    TNV3_OPAQUE_V(ln_w);                                // the write-out's lane arithmetic is redone per tile, not kept live across the chunk loop
    const int bl_w = ln_w & 31, half_w = ln_w >> 5;
    const int t_r = bl_w >> 4, t_col = bl_w & 15;
    const size_t plane0 = ((size_t)e_n * Cout + e_m0) * HW;
    const unsigned planes_b = (unsigned)MB * (unsigned)HW * 4u;
    const tnv3_rsrc_t r_dst = tnv3_make_rsrc(a.dst + plane0, planes_b);
    const tnv3_rsrc_t r_add = tnv3_make_rsrc(has_addend ? a.addend + plane0 : a.dst + plane0, planes_b);
    auto chan_off = [&](int r) -> unsigned { return (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)HW * 4u; };
    auto writeout = [&](auto xgc) {
      constexpr int XG = decltype(xgc)::value, XI = XG >> 1, XJ = XG & 1;
      const int oh = e_h0 + 4 * t_r + XG, ow = e_w0 + 4 * t_col;
      // (wm == RH: both are wave >> 2; an output row below the image: out of the descriptor's range -- loads give 0, stores are dropped)
      const unsigned lane_off_b = (oh < H && TL != 2 && TL != 4) ? (unsigned)((RH * 32 + 4 * half_w) * HW + oh * W + ow) * 4u : kDmaOob;
      const int c4 = e_m0 + RH * 32 + 4 * half_w;          // channel of r = 0; r -> c4 + (r & 3) + 8 * (r >> 2)
      double* red = reinterpret_cast<double*>(lds + 8 * 3 * 2 * 2 * 64 * 4);      // STATS: [wave 8][half 2][16 r][2] behind the exchange region (4 KB of the V stages' last 12)
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) {                   // channels r = 4 rd .. 4 rd + 3 of the lane's sixteen
        // this round's loads first: their latency hides behind the partial transforms and the exchange
        f32x4 ad[4], mu4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, sc4 = f32x4{1.0f, 1.0f, 1.0f, 1.0f}, sh4 = mu4;
        if (has_addend) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) ad[rr] = tnv3_buf_load_f4(r_add, lane_off_b, chan_off(4 * rd + rr));
        }
        if (has_affine) {
          sc4 = *reinterpret_cast<const f32x4*>(a.scale + c4 + 8 * rd);
          sh4 = *reinterpret_cast<const f32x4*>(a.shift + c4 + 8 * rd);
          if (has_mean) mu4 = *reinterpret_cast<const f32x4*>(a.mean + c4 + 8 * rd);
        }
        double q1[STATS ? 4 : 1], q2[STATS ? 4 : 1];      // STATS: this lane's four pixels per channel of the round -- sum, sum of squares
        wf2 own[2][4];                                   // [pair][output column]: this wave's own share of output row XG
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int r0 = 4 * rd + 2 * pp;
          wf2 wv[3][4];                                  // W[i][b] = sum_j M[i][j] A^T[b][j]
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const wf2 m0 = {acc[3 * i][r0], acc[3 * i][r0 + 1]}, m1 = {acc[3 * i + 1][r0], acc[3 * i + 1][r0 + 1]},
                      m2 = {acc[3 * i + 2][r0], acc[3 * i + 2][r0 + 1]};
            wino43_at_half2<XJ>(m0, m1, m2, wv[i]);
          }
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {               // output columns 2 hb, 2 hb + 1 (half of P at a time: sixteen registers less)
            wf2 pr[4][2];                                // P[a][b] = sum_i A^T[a][i] W[i][b]
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              wf2 o[4];
              wino43_at_half2<XI>(wv[0][2 * hb + bb], wv[1][2 * hb + bb], wv[2][2 * hb + bb], o);
              pr[0][bb] = o[0]; pr[1][bb] = o[1]; pr[2][bb] = o[2]; pr[3][bb] = o[3];
            }
#pragma unroll
            for (int arow = 0; arow < 4; ++arow) {
              if (arow == XG) {
                own[pp][2 * hb] = pr[arow][0];
                own[pp][2 * hb + 1] = pr[arow][1];
              } else {                                   // to wave (arow, wm): slot = this wave's rank among its three senders
                const int slot = XG < arow ? XG : XG - 1;
                float* dst = xch + ((((((RH * 4 + arow) * 3 + slot) * 2 + pp) * 2) + hb) * 64 + ln_w) * 4;
                if (TL < 3) *reinterpret_cast<f32x4*>(dst) = f32x4{pr[arow][0][0], pr[arow][0][1], pr[arow][1][0], pr[arow][1][1]};
                else own[pp][2 * hb] += pr[arow][0] + pr[arow][1];      // (keeps the arithmetic alive)
              }
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          wf2 v2[4];
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) v2[b2] = own[pp][b2];
#pragma unroll
          for (int slot = 0; slot < 3; ++slot) {
            const float* src = xch + (((((RH * 4 + XG) * 3 + slot) * 2 + pp) * 2) * 64 + ln_w) * 4;
            if (TL >= 3) continue;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(src), g1 = *reinterpret_cast<const f32x4*>(src + 64 * 4);
            v2[0] += wf2{g0[0], g0[1]}; v2[1] += wf2{g0[2], g0[3]}; v2[2] += wf2{g1[0], g1[1]}; v2[3] += wf2{g1[2], g1[3]};
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int rr = 2 * pp + j, r = 4 * rd + rr;
            f32x4 v = {v2[0][j], v2[1][j], v2[2][j], v2[3][j]};
            if (has_addend) v += ad[rr];
            if (has_affine) {
              const float mu = mu4[rr], sc = sc4[rr], sh = sh4[rr];
#pragma unroll
              for (int b2 = 0; b2 < 4; ++b2) v[b2] = (v[b2] - mu) * sc + sh;
            }
            if (!STATS && a.relu) {
#pragma unroll
              for (int b2 = 0; b2 < 4; ++b2) v[b2] = v[b2] > 0.0f ? v[b2] : 0.0f;
            }
            tnv3_buf_store_f4(r_dst, lane_off_b, chan_off(r), v);
            if constexpr (STATS) {                        // (rows below the image do not count)
              const bool in = oh < H;
              q1[rr] = in ? ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]) : 0.0;
              q2[rr] = in ? ((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1]) + ((double)v[2] * (double)v[2] + (double)v[3] * (double)v[3]) : 0.0;
            }
          }
        }
        if constexpr (STATS) {
          // BatchNorm batch statistics from the epilogue's registers (model.py:9 in training mode): a half-wave holds 32 tiles x 4 pixels
          // of the round's four channels.  Reduce-scatter over lane bits 4, 3 (the lane keeps the channels its bits select and adds the
          // partner's copy), then plain exchanges over bits 2, 1, 0: every lane of an 8-lane run ends up with the half-wave's sum of
          // channel 2 * bit4 + bit3 of the round.  The four xg waves (output rows) of a channel block are folded through LDS after the
          // last round, in a fixed order.  fp64: deterministic.
          {
            const bool up = (bl_w & 16) != 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const double s1 = up ? q1[i] : q1[i + 2], k1 = up ? q1[i + 2] : q1[i];
              const double s2 = up ? q2[i] : q2[i + 2], k2 = up ? q2[i + 2] : q2[i];
              q1[i] = k1 + __shfl_xor(s1, 16, 64);
              q2[i] = k2 + __shfl_xor(s2, 16, 64);
            }
          }
          {
            const bool up = (bl_w & 8) != 0;
            const double s1 = up ? q1[0] : q1[1], k1 = up ? q1[1] : q1[0];
            const double s2 = up ? q2[0] : q2[1], k2 = up ? q2[1] : q2[0];
            q1[0] = k1 + __shfl_xor(s1, 8, 64);
            q2[0] = k2 + __shfl_xor(s2, 8, 64);
          }
#pragma unroll
          for (int o = 4; o > 0; o >>= 1) {
            q1[0] += __shfl_xor(q1[0], o, 64);
            q2[0] += __shfl_xor(q2[0], o, 64);
          }
          if ((bl_w & 7) == 0) {
            const int rr = 2 * ((bl_w >> 4) & 1) + ((bl_w >> 3) & 1);
            double* d = red + (((RH * 4 + XG) * 2 + half_w) * 16 + 4 * rd + rr) * 2;
            d[0] = q1[0];
            d[1] = q2[0];
          }
        }
        __syncthreads();
      }
    };
    switch (xg) {                                        // wave-uniform: one scalar branch around the whole write-out
      case 0: writeout(std::integral_constant<int, 0>{}); break;
      case 1: writeout(std::integral_constant<int, 1>{}); break;
      case 2: writeout(std::integral_constant<int, 2>{}); break;
      default: writeout(std::integral_constant<int, 3>{}); break;
    }
    if constexpr (STATS) {
      __syncthreads();
      if (tid < MB) {                                      // channel tid of this block: fold the four waves (output rows) that own its pixels
        const double* red = reinterpret_cast<const double*>(lds + 8 * 3 * 2 * 2 * 64 * 4);
        const int cwm = tid >> 5, q = tid & 31;
        const int chalf = (q >> 2) & 1, cr = (q & 3) + 4 * (q >> 3);
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double* d = red + (((cwm * 4 + g) * 2 + chalf) * 16 + cr) * 2;
          s1 += d[0];
          s2 += d[1];
        }
        double* o = a.stats + ((size_t)(e_m0 + tid) * nPT + e_pt) * 2;
        o[0] = s1;
        o[1] = s2;
      }
      __syncthreads();                                     // the fold has read `red`: the next tile's transform may overwrite the V stages
    }

    tl_stamp(3);
    if (!have_next) break;
    if (!EARLY) {
      set_tile(walk.n, walk.trow * Cfg::TH, walk.tcol * Cfg::TW, walk.mb * MB);
      issue_raw();
    }
    issue_a();
    tl_stamp(4);
  }
  if constexpr (TL != 0) {
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.stats) + wave * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = tl_acc[i];
      o[5] = (unsigned long long)tl_chunks;
      o[6] = (unsigned long long)tl_tiles;
      o[7] = 0;
    }
  }
  };
  if (rh) body(std::integral_constant<int, 1>{}); else body(std::integral_constant<int, 0>{});
}

}  // namespace tnv3
