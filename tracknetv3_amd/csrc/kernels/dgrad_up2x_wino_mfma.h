// dgrad_up2x_wino_mfma.h -- data gradient of the upsampled half of a decoder-entry layer (model.py:65,67,69), straight at the low
// resolution, as ONE GEMM with K = 9 * Cout.
//
// The gradient w.r.t. the upsampled tensor is a plain 3x3 'same' correlation of dZ with the transposed, flipped filter
// w~[ci][co][kh][kw] = w[co][ci][2-kh][2-kw]; in Winograd F(2x2, 3x3) form its 2x2 tile at (2i, 2j) is Y = A^T M A with
// M_xi = sum_co (G w~ G^T)_xi * (B^T d B)_xi, d = the 4x4 patch of dZ around the tile.  nn.Upsample's backward SUMS that tile:
//     dX_low[ci][i][j] = 1^T Y 1 = c^T M c,        c = A 1 = (1, 2, 0, -1)
// -- transform row / column 2 drops out (the mirror image of the forward's vanishing row d2 - d1, conv_up2x_wino_mfma.h), and what
// is left is a single sum over (co, xi):
//     dX_low[ci][p] = sum_{co, xi in {0,1,3}^2}  U''_xi[co][ci] * V_xi[co][p],     U'' = c_a c_b (G w~ G^T)[a][b],  V = (B^T d B)[a][b]
// With c folded into G -- G'' = [1 0 0; 1 1 1; 0 0 -1] -- U'' is a table of signed tap sums (no 1/2), and V needs adds only:
// rows d0 - d2, d1 + d2, d1 - d3 of the patch, then the same along the columns.  9 multiply-adds per (ci, co, low-res pixel)
// instead of the 16 of dgrad_up2x_mfma_kernel's 4x4 stride-2 correlation, ONE accumulator per output (the nine GEMMs share it),
// no output transform.  Same gradient in exact arithmetic; fp32 rounding as in the other Winograd kernels.
//
// Kernel: the streaming persistent skeleton of conv3x3_wino3_mfma.h / conv_up2x_wino_mfma.h.  A tile = 128 input channels (ci) x
// 64 low-res pixels (two rows of 32); per chunk of 8 output channels (72 K rows): the filter panel U''[72][128] and the raw dZ
// tile [8][6 rows][72 columns] arrive by LDS-DMA through buffer descriptors, wave group g (= low-res row g) transforms its row's
// patches into V[72][64] and runs 36 MFMAs per wave on its 32 x 32 block (one accumulator block per wave).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct DgradUp2xWinoArgs {
  const float* dz;     // [N][Cout][2 Hl][2 Wl]
  const float* u;      // [round_up(Cout, 8)][9][C0]   U'' (dgrad_up2x_wino_pack_kernel)
  float* dst;          // [N][C0][Hl][Wl]   gradient w.r.t. the low-resolution operand of nn.Upsample(2)
  int N, C0, Cout, Hl, Wl;
  int first;           // the wave group that runs its MFMAs first in a chunk: 0 = the older waves (production), 1 = round 2's order
};

// w[Cout][Cin][3][3] (its first c0 input channels) -> u[copad][9][c0], xi = 3 * a + b over transform rows / columns (0, 1, 3)
inline __global__ void __launch_bounds__(256) dgrad_up2x_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin,
                                                                       int c0, int copad) {
  const long total = (long)copad * c0;
  if (blockIdx.x == 0 && threadIdx.x < kPackZeroTail) u[(size_t)copad * 9 * c0 + threadIdx.x] = 0.0f;        // the zero tail behind the panel
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int ci = (int)(e % c0), co = (int)(e / c0);
    float g[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + k] : 0.0f;
    float t[3][3];                                      // rows of G'' w~ (w~ row kh' = w row 2 - kh'):  w2,  w0 + w1 + w2,  -w0
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { t[0][kw] = g[2][kw]; t[1][kw] = (g[0][kw] + g[1][kw]) + g[2][kw]; t[2][kw] = -g[0][kw]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) {                       // columns the same way (w~ column kw' = w column 2 - kw')
      float* o = u + ((size_t)co * 9 + a * 3) * c0 + ci;
      o[0] = t[a][2];
      o[(size_t)c0] = (t[a][0] + t[a][1]) + t[a][2];
      o[2 * (size_t)c0] = -t[a][0];
    }
  }
}

struct DgradUp2xWinoCfg {
  static constexpr int CC = 8, NT = 512, MB = 128, TWL = 32, NXI = 9;    // a tile: 2 low-res rows x TWL low-res columns = 64 pixels
  static constexpr int PIX = 2 * TWL, KR = CC * NXI;                     // 72 K rows per chunk
  static constexpr int U_FLOATS = KR * MB;                               // 9216: 2304 pieces = 4 rounds of 512 + one of 256 (waves 0-3)
  static constexpr int NU4 = 5;
  static constexpr int V_STAGE = KR * PIX;                               // 4608
  static constexpr int RW = 2 * TWL + 8, RAW_FLOATS = CC * 6 * RW;       // [co][6 rows][72: high-res columns 2 j0 - 4 .. 2 j0 + 67]
  static constexpr int RAW_PIECES = RAW_FLOATS / 4;                      // 864 = 512 + 352 -> the second round runs on waves 0-5 (384 slots)
  static constexpr int RAW_STAGE = (512 + 384) * 4;
  static constexpr int LDS_FLOATS = 2 * U_FLOATS + 2 * V_STAGE + 2 * RAW_STAGE;
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

inline __global__ void __launch_bounds__(DgradUp2xWinoCfg::NT) dgrad_up2x_wino_stream_kernel(const DgradUp2xWinoArgs a) {
  using Cfg = DgradUp2xWinoCfg;
  constexpr int CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TWL = Cfg::TWL, NXI = Cfg::NXI, PIX = Cfg::PIX, KR = Cfg::KR, RW = Cfg::RW;
  constexpr int NU4 = Cfg::NU4;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* u_s = lds;                                   // two stages each
  float* v_s = lds + 2 * Cfg::U_FLOATS;
  float* raw_s = v_s + 2 * Cfg::V_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;           // group g: low-res row g of the tile; wq: 32-channel block of the 128 ci
  const int half = lane >> 5, bl = lane & 31;
  const int Hl = a.Hl, Wl = a.Wl, C0 = a.C0, Cout = a.Cout, HWl = Hl * Wl;
  const int W = 2 * Wl, H = 2 * Hl, HW = 4 * HWl;
  const int tilesH = Hl / 2, tilesW = Wl / TWL;
  const int nPT = a.N * tilesH * tilesW, nMB = C0 / MB;
  const int nChunks = (Cout + CC - 1) / CC;            // >= 2 (host)

  ConvTileWalk walk;                                  // always one tile ahead of the one being computed
  walk.init(blockIdx.x, gridDim.x, nMB, nPT, tilesH, tilesW);
  if (!walk.valid) return;

  // ---- per-lane DMA offsets
  unsigned vo_u[NU4], vo_r[2], vo_rn[2];
#pragma unroll
  for (int i = 0; i < NU4; ++i) {                     // piece e of [72 rows][MB / 4]: 16 bytes of 128 input channels
    const int e = tid + i * NT;
    const int row = e / (MB / 4), m4 = e - row * (MB / 4);
    vo_u[i] = (unsigned)(row * C0 + m4 * 4) * 4u;     // (round 4 runs on waves 0-3 only: e < 2304)
  }
  auto raw_offsets = [&](unsigned (&vo)[2], int i0, int j0) {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);                              // recomputed per tile; nothing of it stays live across the chunk loop
#pragma unroll
    for (int i = 0; i < 2; ++i) {                     // raw piece e of [CC][6 rows][RW / 4]; rows / columns outside the image read as zeros
      const int e = t_op + i * NT;
      const int c = e / (6 * RW / 4), r = e - c * (6 * RW / 4);
      const int tr = r / (RW / 4), q = r - tr * (RW / 4);
      const int gh = 2 * i0 - 1 + tr, gw = 2 * j0 - 4 + 4 * q;
      const bool ok = e < Cfg::RAW_PIECES && gh >= 0 && gh < H && gw >= 0 && gw < W;
      vo[i] = ok ? (unsigned)(c * HW + gh * W + gw) * 4u : kDmaOob;
    }
  };
  int c_n = walk.n, c_i0 = walk.trow * 2, c_j0 = walk.tcol * TWL, c_m0 = walk.mb * MB;
  raw_offsets(vo_r, c_i0, c_j0);
  walk.next();
  bool have_next = walk.valid;
  int n_n = walk.n, n_i0 = walk.trow * 2, n_j0 = walk.tcol * TWL, n_m0 = walk.mb * MB;
  if (have_next) raw_offsets(vo_rn, n_i0, n_j0);

  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int wbase = swave * 64;                       // scalar: the LDS-DMA destinations (M0) stay on the SALU
  const size_t u_step = (size_t)KR * C0, z_step = (size_t)CC * HW;         // floats per chunk
  auto dma_u = [&](const float* up, int su) {         // filter chunk at `up` (72 rows of C0 floats, this tile's 128 channels first)
    const tnv3_rsrc_t ru = tnv3_make_rsrc(up, (unsigned)(KR * C0) * 4u);
    float* us = u_s + su * Cfg::U_FLOATS;
#pragma unroll
    for (int i = 0; i < NU4 - 1; ++i) tnv3_buf_dma16(ru, us + (i * NT + wbase) * 4, vo_u[i]);
    if (swave < 4) tnv3_buf_dma16(ru, us + ((NU4 - 1) * NT + wbase) * 4, vo_u[NU4 - 1]);
  };
  auto dma_r = [&](const float* zp, int cvalid, const unsigned (&vo)[2], int sr) {        // raw dZ tile of a chunk -> raw stage sr
    const tnv3_rsrc_t rr = tnv3_make_rsrc(zp, (unsigned)(cvalid < CC ? cvalid : CC) * (unsigned)HW * 4u);   // channels past Cout: zero
    float* rs = raw_s + sr * Cfg::RAW_STAGE;
    tnv3_buf_dma16(rr, rs + wbase * 4, vo[0]);
    if (swave < 6) tnv3_buf_dma16(rr, rs + (NT + wbase) * 4, vo[1]);
  };
  const float* c_u = a.u + c_m0;
  const float* c_z = a.dz + (size_t)c_n * Cout * HW;
  const float* n_u = a.u + n_m0;
  const float* n_z = a.dz + (size_t)n_n * Cout * HW;

  // ---- patch transform: threads 0..127 of a group: (channel pc, pixel pair pp of the group's low-res row) -> the 9 V values of both
  const int tg = tid & 255, pc = (tg >> 4) & 7, pp = tg & 15;
  const bool t_active = tg < 128;
  const int t_src = pc * (6 * RW) + (2 * grp) * RW + 4 * pp;          // raw rows 2g .. 2g+3 = high rows 2(i0+g)-1 .. 2(i0+g)+2
  const int t_dst = pc * NXI * PIX + grp * TWL + 2 * pp;
  typedef float wf2 __attribute__((ext_vector_type(2)));
  auto transform = [&](int stage) {                     // raw stage -> V stage of the same parity
    if (!t_active) return;
    const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src;
    float x[4][6];                                      // high-res columns 2j-1 .. 2j+4 of the pair (j, j+1): raw columns 4pp+3 .. 4pp+8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * RW);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * RW + 4);
      const float q2 = d[r * RW + 8];
      x[r][0] = q0[3]; x[r][1] = q1[0]; x[r][2] = q1[1]; x[r][3] = q1[2]; x[r][4] = q1[3]; x[r][5] = q2;
    }
    float* v = v_s + stage * Cfg::V_STAGE + t_dst;
#pragma unroll
    for (int ra = 0; ra < 3; ++ra) {                    // transform rows 0, 1, 3:  d0 - d2,  d1 + d2,  d1 - d3
      float rr[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) rr[j] = ra == 0 ? x[0][j] - x[2][j] : (ra == 1 ? x[1][j] + x[2][j] : x[1][j] - x[3][j]);
      wf2 o;                                            // (pixel 2pp, pixel 2pp+1) per transform column 0, 1, 3
      o[0] = rr[0] - rr[2]; o[1] = rr[2] - rr[4]; *reinterpret_cast<wf2*>(v + (ra * 3 + 0) * PIX) = o;
      o[0] = rr[1] + rr[2]; o[1] = rr[3] + rr[4]; *reinterpret_cast<wf2*>(v + (ra * 3 + 1) * PIX) = o;
      o[0] = rr[1] - rr[3]; o[1] = rr[3] - rr[5]; *reinterpret_cast<wf2*>(v + (ra * 3 + 2) * PIX) = o;
    }
  };

  f32x16 acc;
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
  const int a_off = half * MB + wq * 32 + bl;
  const int b_off = half * PIX + grp * TWL + bl;
  auto mfma_chunk = [&](int stage, auto first_c) {      // first_c: the tile's first chunk starts the accumulator from the inline zero
    constexpr bool FIRST = decltype(first_c)::value;
    const float* A = u_s + stage * Cfg::U_FLOATS + a_off;
    const float* B = v_s + stage * Cfg::V_STAGE + b_off;
    constexpr int NSTEP = KR / 2;                        // 36 K pairs
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      av[s % RING] = A[(2 * s) * MB];
      bv[s % RING] = B[(2 * s) * PIX];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], FIRST && s == 0 ? zero16 : acc, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  auto chunk_barrier = [&]() {                          // own DMAs landed, own V writes done, everybody finished with the old stages
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  int gs = 0;                                          // chunks done so far: the current chunk uses stage gs & 1
  const float* pu;
  const float* pz;
  int pz_left;
  auto chunk_body = [&](auto first_c, auto where_c) {   // see conv3x3_wino_stream_mfma_kernel
    constexpr int WHERE = decltype(where_c)::value;
    const int sc = gs & 1, sn = sc ^ 1;
    const bool ahead = WHERE != 2 || have_next;
    auto dmas = [&]() {
      if constexpr (WHERE == 0) {
        dma_u(pu, sn);
        dma_r(pz, pz_left, vo_r, sc);
      } else if constexpr (WHERE == 1) {
        dma_u(pu, sn);
        if (have_next) dma_r(n_z, Cout, vo_rn, sc);
      } else if (have_next) {
        dma_u(n_u, sn);
        dma_r(n_z + z_step, Cout - CC, vo_rn, sc);
      }
    };
    if (grp != a.first) {                               // transform-first group: DMAs, transform, MFMAs;  the other: MFMAs, DMAs, transform
      dmas();
      if (ahead) transform(sn);
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(sc, first_c);
    } else {
      mfma_chunk(sc, first_c);
      __builtin_amdgcn_sched_barrier(0);
      dmas();
      if (ahead) transform(sn);
      __builtin_amdgcn_sched_barrier(0);
    }
    pu += u_step; pz += z_step; pz_left -= CC;
    chunk_barrier();
    ++gs;
  };
  typedef std::integral_constant<int, 0> in_tile_t;
  typedef std::integral_constant<int, 1> second_to_last_t;
  typedef std::integral_constant<int, 2> last_t;

  // pipeline fill (once per workgroup): filters of chunk 0, raw tiles of chunks 0 and 1, V of chunk 0
  dma_u(c_u, 0);
  dma_r(c_z, Cout, vo_r, 0);
  dma_r(c_z + z_step, Cout - CC, vo_r, 1);
  chunk_barrier();
  transform(0);
  chunk_barrier();
  for (;;) {                                            // one pass per tile
    pu = c_u + u_step; pz = c_z + 2 * z_step; pz_left = Cout - 2 * CC;
    if (nChunks == 2) {
      chunk_body(std::true_type{}, second_to_last_t{});
    } else {
      chunk_body(std::true_type{}, in_tile_t{});
      for (int k = 1; k < nChunks - 2; ++k) chunk_body(std::false_type{}, in_tile_t{});
      chunk_body(std::false_type{}, second_to_last_t{});
    }
    chunk_body(std::false_type{}, last_t{});

    // ---- write-out (no transform, no LDS): lane = low-res pixel (c_i0 + grp, c_j0 + bl), 16 input channels
    {
      int tid_e = threadIdx.x;
      TNV3_OPAQUE_V(tid_e);
      const int e_lane = tid_e & 63, e_wave = tid_e >> 6, e_grp = e_wave >> 2, e_wq = e_wave & 3;
      const int e_half = e_lane >> 5, e_bl = e_lane & 31;
      const unsigned lane_off_b = (unsigned)((e_wq * 32 + 4 * e_half) * HWl + (c_i0 + e_grp) * Wl + c_j0 + e_bl) * 4u;
      float* base = a.dst + ((size_t)c_n * C0 + c_m0) * HWl;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float*>(reinterpret_cast<char*>(base + (size_t)((r & 3) + 8 * (r >> 2)) * HWl) + lane_off_b) = acc[r];
    }
    if (!have_next) break;
    c_n = n_n; c_i0 = n_i0; c_j0 = n_j0; c_m0 = n_m0; c_u = n_u; c_z = n_z;
    vo_r[0] = vo_rn[0]; vo_r[1] = vo_rn[1];
    walk.next();
    have_next = walk.valid;
    n_n = walk.n; n_i0 = walk.trow * 2; n_j0 = walk.tcol * TWL; n_m0 = walk.mb * MB;
    n_u = a.u + n_m0;
    n_z = a.dz + (size_t)n_n * Cout * HW;
    if (have_next) raw_offsets(vo_rn, n_i0, n_j0);
  }
}

}  // namespace tnv3
