// conv_up2x_wino_mfma.h -- the upsampled half of a decoder-entry layer (model.py:65,67,69: conv3x3 over
// cat([Upsample(2)(x), skip])) in Winograd F(2x2, 3x3) form, where the nearest-2x upsampling makes 7 of the 16 transform
// coefficients vanish identically.
//
// For the 2x2 output tile at (2i, 2j) the 4x4 input patch of the UPSAMPLED tensor has rows (L[i-1], L[i], L[i], L[i+1]) of the
// low-resolution tensor L (zero outside it: the upsampled image's zero padding), and the same structure along the columns.
// With B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] the rows of B^T d are
//     d0 - d2 = L[i-1] - L[i],      d1 + d2 = 2 L[i],      d2 - d1 = 0,      d1 - d3 = L[i] - L[i+1]
// so V = B^T d B is non-zero only at transform rows / columns {0, 1, 3}: NINE of the sixteen GEMMs
//     M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]
// remain, 9 multiply-adds per low-resolution pixel and channel pair instead of the 16 of conv_up2x_mfma.h's pre-summed 2x2
// class filters (and the 36 of the reference's direct form).  The factor 2 of row / column 1 moves into the filter: with
// G' = [1 0 0; 1 1 1; 0 0 1] (rows 0, 1, 3 of G, row 1 doubled) U' = G' g G'^T is a table of plain tap SUMS -- no 1/2 anywhere --
// and V' uses R0 = L[i-1] - L[i], R1 = L[i], R3 = L[i] - L[i+1] (then the same along the columns): adds only.
// Output: Y = A^T M A with M's row / column 2 zero:  Y[0][x] = T0[x] + T1[x], Y[1][x] = T1[x] - T3[x],
// Ta[0] = M[a][0] + M[a][1], Ta[1] = M[a][1] - M[a][3].  Same function as the reference in exact arithmetic; fp32 rounding as
// in the other Winograd kernels (+-1 transforms only).
//
// Kernel: the streaming persistent form of conv3x3_wino3_mfma.h (one workgroup per CU walks the tile list, the chunk pipeline
// runs through the tile boundaries, LDS-DMA through buffer descriptors), with a tile of 64 output channels x 128 tiles (two
// low-resolution rows of 64 pixels = 4 x 128 output pixels).  Nine accumulators per 32 x 32 block fit one wave (144 registers,
// two waves per SIMD), so every wave owns ALL xi of its block: wave group g = low-resolution row g of the tile (it transforms
// that row's patches and consumes them), and the output transform happens in registers -- no exchange through LDS, no barrier in
// the write-out.  Per 8-channel chunk a wave runs 36 MFMAs against 3.4 DMA pieces and one tile-pair transform per thread (the
// 16-xi kernel: 32 MFMAs, 6 pieces): less non-MFMA work per MFMA on top of 9/16 of the MFMAs.
// The result is the layer's partial sum P (no BN / ReLU): the skip half's launch takes it as `addend` (tnv3_conv3x3_wino_forward).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct ConvUp2xWinoArgs {
  const float* src;    // [N][C0][Hl][Wl]   low-resolution operand of nn.Upsample(2)
  const float* u;      // [round_up(C0, 8)][9][Cout]   U' (conv_up2x_wino_pack_kernel)
  float* dst;          // [N][Cout][2 Hl][2 Wl]   partial sums
  int N, C0, Cout, Hl, Wl;
  int first;           // the wave group that runs its MFMAs FIRST in a chunk (the other one starts with DMAs + transform): 0 = the older
                       // waves of each SIMD (production, see WinoV3Cfg::SWAP), 1 = round 2's order
};

// w[Cout][Cin][3][3] (the first c0 input channels) -> u[c0pad][9][Cout], xi = 3 * a + b over transform rows / columns (0, 1, 3).
inline __global__ void __launch_bounds__(256) conv_up2x_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin,
                                                                      int c0, int c0pad) {
  const long total = (long)c0pad * Cout;
  if (blockIdx.x == 0 && threadIdx.x < kPackZeroTail) u[(size_t)c0pad * 9 * Cout + threadIdx.x] = 0.0f;      // the zero tail behind the panel
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int co = (int)(e % Cout), ci = (int)(e / Cout);
    float g[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = ci < c0 ? w[((size_t)co * Cin + ci) * 9 + k] : 0.0f;
    float t[3][3];                                      // rows of G' g:  g0,  g0 + g1 + g2,  g2
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { t[0][kw] = g[0][kw]; t[1][kw] = (g[0][kw] + g[1][kw]) + g[2][kw]; t[2][kw] = g[2][kw]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float* o = u + ((size_t)ci * 9 + a * 3) * Cout + co;
      o[0] = t[a][0];
      o[(size_t)Cout] = (t[a][0] + t[a][1]) + t[a][2];
      o[2 * (size_t)Cout] = t[a][2];
    }
  }
}

struct ConvUp2xWinoCfg {
  static constexpr int CC = 8, NT = 512, MB = 64, TW = 64, NXI = 9;      // a tile: 2 low-res rows x TW low-res columns = 128 tiles
  static constexpr int TILES = 2 * TW;
  static constexpr int U_FLOATS = CC * NXI * MB;                         // 4608
  static constexpr int NU4 = 3, U_STAGE = NU4 * NT * 4;                  // 1152 pieces dealt as 3 per thread (the last 384 slots: padding)
  static constexpr int V_STAGE = CC * NXI * TILES;                       // 9216
  static constexpr int RW = TW + 8, RAW_FLOATS = CC * 4 * RW;            // [c][4 rows][72: columns j0-4 .. j0+67]
  static constexpr int NRAW = 2, RAW_STAGE = NRAW * NT * 4;              // 576 pieces dealt as 2 per thread
  static constexpr int LDS_FLOATS = 2 * U_STAGE + 2 * V_STAGE + 2 * RAW_STAGE;
  static_assert(U_FLOATS / 4 <= NU4 * NT && RAW_FLOATS / 4 <= NRAW * NT, "every piece has a slot");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

inline __global__ void __launch_bounds__(ConvUp2xWinoCfg::NT) conv_up2x_wino_stream_kernel(const ConvUp2xWinoArgs a) {
  using Cfg = ConvUp2xWinoCfg;
  constexpr int CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TW = Cfg::TW, NXI = Cfg::NXI, TILES = Cfg::TILES, RW = Cfg::RW;
  constexpr int NU4 = Cfg::NU4, NRAW = Cfg::NRAW;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* u_s = lds;                                   // two stages each
  float* v_s = lds + 2 * Cfg::U_STAGE;
  float* raw_s = v_s + 2 * Cfg::V_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;           // group g: low-res row g of the tile
  const int wn = wq & 1, wm = wq >> 1;                // wn: 32-tile half of the row, wm: 32-channel half
  const int half = lane >> 5, bl = lane & 31;
  const int Hl = a.Hl, Wl = a.Wl, C0 = a.C0, Cout = a.Cout, HWl = Hl * Wl;
  const int W = 2 * Wl, HW = 4 * HWl;
  const int tilesH = Hl / 2, tilesW = Wl / TW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (C0 + CC - 1) / CC;              // >= 2 (host)

  ConvTileWalk walk;                                  // always one tile ahead of the one being computed
  walk.init(blockIdx.x, gridDim.x, nMB, nPT, tilesH, tilesW);
  if (!walk.valid) return;

  // ---- per-lane DMA offsets: the filter pieces' are the same for every tile and chunk, the raw pieces' follow the tile
  unsigned vo_u[NU4], vo_r[NRAW], vo_rn[NRAW];
#pragma unroll
  for (int i = 0; i < NU4; ++i) {                     // piece e of [CC * 9 rows][MB / 4]: row = (ci, xi), 16 bytes of 64 channels
    const int e = tid + i * NT;
    const int row = e / (MB / 4), m4 = e - row * (MB / 4);
    vo_u[i] = e < Cfg::U_FLOATS / 4 ? (unsigned)(row * Cout + m4 * 4) * 4u : kDmaOob;
  }
  auto raw_offsets = [&](unsigned (&vo)[NRAW], int i0, int j0) {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);                              // recomputed per tile; nothing of it stays live across the chunk loop
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {                  // raw piece e of [CC][4 rows][RW / 4]; rows / columns outside the image read as zeros
      const int e = t_op + i * NT;
      const int c = e / (4 * RW / 4), r = e - c * (4 * RW / 4);
      const int tr = r / (RW / 4), q = r - tr * (RW / 4);
      const int gh = i0 - 1 + tr, gw = j0 - 4 + 4 * q;
      const bool ok = e < Cfg::RAW_FLOATS / 4 && gh >= 0 && gh < Hl && gw >= 0 && gw < Wl;
      vo[i] = ok ? (unsigned)(c * HWl + gh * Wl + gw) * 4u : kDmaOob;
    }
  };
  int c_n = walk.n, c_i0 = walk.trow * 2, c_j0 = walk.tcol * TW, c_m0 = walk.mb * MB;
  raw_offsets(vo_r, c_i0, c_j0);
  walk.next();
  bool have_next = walk.valid;
  int n_n = walk.n, n_i0 = walk.trow * 2, n_j0 = walk.tcol * TW, n_m0 = walk.mb * MB;
  if (have_next) raw_offsets(vo_rn, n_i0, n_j0);

  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);      // scalar: the LDS-DMA destinations (M0) stay on the SALU
  const size_t u_step = (size_t)CC * NXI * Cout, x_step = (size_t)CC * HWl;     // floats per chunk
  auto dma_u = [&](const float* up, int su) {         // filter chunk at `up` (CC * 9 rows of Cout floats, this tile's 64 channels first)
    const tnv3_rsrc_t ru = tnv3_make_rsrc(up, (unsigned)(CC * NXI * Cout) * 4u);
    float* us = u_s + su * Cfg::U_STAGE;
#pragma unroll
    for (int i = 0; i < NU4; ++i) tnv3_buf_dma16(ru, us + (i * NT + wbase) * 4, vo_u[i]);
  };
  auto dma_r = [&](const float* xp, int cvalid, const unsigned (&vo)[NRAW], int sr) {     // raw low-res tile of a chunk -> raw stage sr
    const tnv3_rsrc_t rr = tnv3_make_rsrc(xp, (unsigned)(cvalid < CC ? cvalid : CC) * (unsigned)HWl * 4u);   // channels past C0: zero
    float* rs = raw_s + sr * Cfg::RAW_STAGE;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) tnv3_buf_dma16(rr, rs + (i * NT + wbase) * 4, vo[i]);
  };
  const float* c_u = a.u + c_m0;
  const float* c_x = a.src + (size_t)c_n * C0 * HWl;
  const float* n_u = a.u + n_m0;
  const float* n_x = a.src + (size_t)n_n * C0 * HWl;

  // ---- patch transform: thread (channel pc, tile pair pp) of its group's row -> the 9 V' values of tiles 2pp, 2pp+1
  const int tg = tid & 255, pc = tg >> 5, pp = tg & 31;
  const int t_src = pc * (4 * RW) + grp * RW + 4 * (pp >> 1);       // raw rows grp, grp+1, grp+2 = low rows i0+grp-1 .. i0+grp+1
  const bool odd = (pp & 1) != 0;
  const int t_dst = pc * NXI * TILES + grp * TW + 2 * pp;
  typedef float wf2 __attribute__((ext_vector_type(2)));
  auto transform = [&](int stage) {                     // raw stage -> V stage of the same parity
    const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src;
    float x[3][4];                                      // low columns j-1 .. j+2 of the pair (j, j+1): raw columns 2pp+3 .. 2pp+6
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * RW);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * RW + 4);
      const float q2 = d[r * RW + 8];
      x[r][0] = odd ? q1[1] : q0[3]; x[r][1] = odd ? q1[2] : q1[0]; x[r][2] = odd ? q1[3] : q1[1]; x[r][3] = odd ? q2 : q1[2];
    }
    float* v = v_s + stage * Cfg::V_STAGE + t_dst;
#pragma unroll
    for (int ra = 0; ra < 3; ++ra) {                    // transform rows 0, 1, 3:  L[i-1] - L[i],  L[i],  L[i] - L[i+1]
      float rr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) rr[j] = ra == 0 ? x[0][j] - x[1][j] : (ra == 1 ? x[1][j] : x[1][j] - x[2][j]);
      wf2 o;                                            // (tile 2pp, tile 2pp+1) per transform column 0, 1, 3
      o[0] = rr[0] - rr[1]; o[1] = rr[1] - rr[2]; *reinterpret_cast<wf2*>(v + (ra * 3 + 0) * TILES) = o;
      o[0] = rr[1];         o[1] = rr[2];         *reinterpret_cast<wf2*>(v + (ra * 3 + 1) * TILES) = o;
      o[0] = rr[1] - rr[2]; o[1] = rr[2] - rr[3]; *reinterpret_cast<wf2*>(v + (ra * 3 + 2) * TILES) = o;
    }
  };

  f32x16 acc[NXI];
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
  const int a_off = (half * NXI) * MB + wm * 32 + bl;
  const int b_off = (half * NXI) * TILES + grp * TW + wn * 32 + bl;
  auto mfma_chunk = [&](int stage, auto first_c) {      // first_c: the tile's first chunk starts every accumulator from the inline zero
    constexpr bool FIRST = decltype(first_c)::value;
    const float* A = u_s + stage * Cfg::U_STAGE + a_off;
    const float* B = v_s + stage * Cfg::V_STAGE + b_off;
    constexpr int NSTEP = (CC / 2) * NXI;                // (channel pair, xi): one MFMA each
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int cp = s / NXI, x = s - cp * NXI;
      av[s % RING] = A[(2 * cp * NXI + x) * MB];
      bv[s % RING] = B[(2 * cp * NXI + x) * TILES];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc[s % NXI] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], FIRST && s < NXI ? zero16 : acc[s % NXI], 0, 0, 0);
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
  const float* pu;                                     // filters one chunk ahead / raw tile two chunks ahead, inside the current tile
  const float* px;
  int px_left;
  // One chunk (see conv3x3_wino_stream_mfma_kernel): WHERE 0 = the chunks ahead lie inside this tile, 1 = second-to-last chunk,
  // 2 = last chunk of the tile.
  auto chunk_body = [&](auto first_c, auto where_c) {
    constexpr int WHERE = decltype(where_c)::value;
    const int sc = gs & 1, sn = sc ^ 1;
    const bool ahead = WHERE != 2 || have_next;
    auto dmas = [&]() {
      if constexpr (WHERE == 0) {
        dma_u(pu, sn);
        dma_r(px, px_left, vo_r, sc);
      } else if constexpr (WHERE == 1) {
        dma_u(pu, sn);
        if (have_next) dma_r(n_x, C0, vo_rn, sc);
      } else if (have_next) {
        dma_u(n_u, sn);
        dma_r(n_x + x_step, C0 - CC, vo_rn, sc);
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
    pu += u_step; px += x_step; px_left -= CC;
    chunk_barrier();
    ++gs;
  };
  typedef std::integral_constant<int, 0> in_tile_t;
  typedef std::integral_constant<int, 1> second_to_last_t;
  typedef std::integral_constant<int, 2> last_t;

  // pipeline fill (once per workgroup): filters of chunk 0, raw tiles of chunks 0 and 1, V of chunk 0
  dma_u(c_u, 0);
  dma_r(c_x, C0, vo_r, 0);
  dma_r(c_x + x_step, C0 - CC, vo_r, 1);
  chunk_barrier();
  transform(0);
  chunk_barrier();
  for (;;) {                                            // one pass per tile
    pu = c_u + u_step; px = c_x + 2 * x_step; px_left = C0 - 2 * CC;
    if (nChunks == 2) {
      chunk_body(std::true_type{}, second_to_last_t{});
    } else {
      chunk_body(std::true_type{}, in_tile_t{});
      for (int k = 1; k < nChunks - 2; ++k) chunk_body(std::false_type{}, in_tile_t{});
      chunk_body(std::false_type{}, second_to_last_t{});
    }
    chunk_body(std::false_type{}, last_t{});

    // ---- output transform in registers + write-out (no LDS): lane = low-res pixel (c_i0 + grp, c_j0 + wn * 32 + bl), 16 channels
    {
      int tid_e = threadIdx.x;
      TNV3_OPAQUE_V(tid_e);                             // redone per tile, not kept live across the chunk loop
      const int e_lane = tid_e & 63, e_wave = tid_e >> 6, e_grp = e_wave >> 2, e_wq = e_wave & 3;
      const int e_wn = e_wq & 1, e_wm = e_wq >> 1, e_half = e_lane >> 5, e_bl = e_lane & 31;
      const int oh = 2 * (c_i0 + e_grp), ow = 2 * (c_j0 + e_wn * 32 + e_bl);
      const unsigned lane_off_b = (unsigned)((e_wm * 32 + 4 * e_half) * HW + oh * W + ow) * 4u;    // bytes from plane (n, m0): < 2^31 (host)
      const tnv3_rsrc_t r_dst = tnv3_make_rsrc(a.dst + ((size_t)c_n * Cout + c_m0) * HW, (unsigned)MB * (unsigned)HW * 4u);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float t[3][2];                                  // Ta[x] of transform rows a = 0, 1, 3
#pragma unroll
        for (int ra = 0; ra < 3; ++ra) {
          t[ra][0] = acc[ra * 3 + 0][r] + acc[ra * 3 + 1][r];
          t[ra][1] = acc[ra * 3 + 1][r] - acc[ra * 3 + 2][r];
        }
        tnv3_f2 y0, y1;
        y0[0] = t[0][0] + t[1][0]; y0[1] = t[0][1] + t[1][1];
        y1[0] = t[1][0] - t[2][0]; y1[1] = t[1][1] - t[2][1];
        const unsigned ch = (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)HW * 4u;
        tnv3_buf_store_f2(r_dst, lane_off_b, ch, y0);
        tnv3_buf_store_f2(r_dst, lane_off_b + (unsigned)W * 4u, ch, y1);
      }
    }
    if (!have_next) break;
    c_n = n_n; c_i0 = n_i0; c_j0 = n_j0; c_m0 = n_m0; c_u = n_u; c_x = n_x;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) vo_r[i] = vo_rn[i];
    walk.next();
    have_next = walk.valid;
    n_n = walk.n; n_i0 = walk.trow * 2; n_j0 = walk.tcol * TW; n_m0 = walk.mb * MB;
    n_u = a.u + n_m0;
    n_x = a.src + (size_t)n_n * C0 * HWl;
    if (have_next) raw_offsets(vo_rn, n_i0, n_j0);
  }
}

}  // namespace tnv3
