// wgrad_wino43_mfma.h -- weight gradient of a plain 3x3 layer in Winograd F(4x4, 3x3) form (tnv3_conv3x3_wgrad_wino, kernel variant 8).
//
// With the forward Y = A^T [ (G g G^T) .* (B^T d B) ] A (conv3x3_wino43_mfma.h: interpolation points (0, +-3/4, +-3/2, inf)) the
// gradient with respect to the filter is
//     dg[co][ci] = G^T [ sum over images and 4x4 tiles of  (A dY A^T) .* (B^T d B) ] G
// -- 36 independent GEMMs  S_xi[co][ci] = sum_tiles Yh_xi[co][tile] * V_xi[ci][tile]  with K = all tiles of the batch, 36 products per
// (co, ci, 4x4 tile) = 2.25 per pixel where F(2x2) (wgrad_wino_mfma.h) multiplies 4 and the direct form 9.  Precision: the weight
// gradient is a leaf (nothing amplifies its rounding); rms 3.4e-7 / max 2.1e-6 of max|dW| against fp64 autograd at batch 10
// (profiles/r03_wino_f43_wgrad_precision.json; F(2x2): 1.2e-7 / 6e-7, the direct fp32 form 1.4e-7 / 7e-7).
//
// Mapping (the 16x16x4 structure of conv3x3_wino43s_mfma.h).  MFMA 16x16x4: M = 16 output channels, N = 16 input channels, K = 4 tiles
// (a strip of 4 x 16 pixels = one STEP); a wave keeps all 36 xi of its 16 x 16 block of S: 144 accumulator registers, two waves per
// SIMD.  Workgroup = 64 co x 32 ci (waves = 4 co blocks x 2 ci blocks) x a contiguous share of the strips (split-K); 512 threads.
//   * Both operands are transformed per step and meet in LDS as 16-byte quads of four xi -- quad q = 3 i' + m holds xi = (i', 2 m),
//     (i', 2 m + 1), (i' + 3, 2 m), (i' + 3, 2 m + 1) -- in MFMA lane order (lane = tile * 16 + channel): one ds_read_b128 per operand
//     and quad = four MFMAs; two stages (the transform of step sigma + 1 runs between the MFMAs of step sigma).
//   * Yh = A dY A^T: waves 0-3, thread = (co, tile): the 4x4 tile of dZ comes STRAIGHT from global memory into registers (four
//     16-byte loads, requested one step ahead), 100 operations, nine 16-byte stores.
//   * V = B^T d B: waves 4-7, thread = (ci, tile, row half): the raw strip of X (32 channels x 6 rows x 20 columns from column
//     16 k - 1: pieces whose global address is 4-byte aligned, as in the forward kernel) arrives by LDS-DMA two steps ahead; 72
//     operations, nine 8-byte stores.  The columns left / right of the image are zeroed after the first pass, the one piece per image
//     that would start before it (channel 0, row 0) is patched from a guarded load.
//   * Epilogue: G^T S G per (co, ci) in registers (90 operations), the 3x3 result to the workgroup's slab [k][tap][co][ci]; the slabs
//     are summed in fp64, in a fixed order, by wgrad_wino43_fold_kernel.  Deterministic.
// Needs Cout % 64 == 0, H % 4 == 0, W % 16 == 0; any Cin (a partial last block of 32 input channels reads zeros).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "conv3x3_wino43s_mfma.h"
#include "wgrad_wino_mfma.h"

namespace tnv3 {

template <int RAWN_>
struct WgradWino43CfgT {
  static constexpr int NT = 512, MB = 64, CB = 32, RAWN = RAWN_;
  static constexpr int YH_STAGE = 9 * 4 * 64 * 4;         // [quad 9][co block 4][lane 64][4]: 36 KB
  static constexpr int V_STAGE = 9 * 2 * 64 * 4;          // [quad 9][ci block 2][lane 64][4]: 18 KB
  static constexpr int RQ = 5;                            // pieces per raw row: columns 16 k - 1 .. 16 k + 18
  static constexpr int RPLANE = 34;                       // pieces per channel plane (6 rows x 5 = 30 used): 2 mod 16, so that the 16 lanes of a
                                                          // ds_read_b128 group (channels {0-3, 12-15} of one tile, {4-11} of the next) hit 16 different 16-byte bank groups
  static constexpr int RAW_SLOTS = CB * RPLANE;           // 1088 pieces per stage
  static constexpr int RAW_STAGE = RAW_SLOTS * 4;         // floats
  static constexpr int NDMA = (RAW_SLOTS + NT - 1) / NT;  // 3 (the third: wave 0)
  static constexpr int DMA_LAST_WAVES = (RAW_SLOTS - (NDMA - 1) * NT + 63) / 64;
  static constexpr int LDS_FLOATS = 2 * (YH_STAGE + V_STAGE) + RAWN * RAW_STAGE;   // 145,408 bytes with two raw stages, 162,816 with three
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};
using WgradWino43Cfg = WgradWino43CfgT<2>;
// Compile-time switches of the kernel body.  The product kernel is <0>; everything else is a measurement twin of libtnv3_diag.so
// (bits 0-4 and 7 switch parts of a step off: WRONG results) or a candidate schedule (bits 5, 8, 9: results unchanged).
struct WgradWino43Sw {
  enum : int { NoDma = 1, NoDy = 2, NoYh = 4, NoV = 8, NoMfma = 16, EarlyDy = 32, Timeline = 64, NoOpReads = 128, Raw3 = 256, EarlyDy2 = 512 };
};

// z = A x along one axis: A = (A^T)^T of conv3x3_wino43_mfma.h, rows [1 0 0 0; 1/s 1 s s^2; 1/s -1 s -s^2; 1/(2s) 1 2s 4s^2; 1/(2s) -1 2s -4s^2; 0 0 0 1]
__device__ __forceinline__ void wino43_a6(float x0, float x1, float x2, float x3, float (&z)[6]) {
  const float e1 = fmaf(kW43S, x2, (1.0f / kW43S) * x0), o1 = fmaf(kW43S2, x3, x1);
  const float e2 = fmaf(kW43_2S, x2, (0.5f / kW43S) * x0), o2 = fmaf(kW43_4S2, x3, x1);
  z[0] = x0; z[1] = e1 + o1; z[2] = e1 - o1; z[3] = e2 + o2; z[4] = e2 - o2; z[5] = x3;
}
// o = G^T m along one axis (G of wino43_g_row: rows n0 [1 0 0]; n1 [1 +-s s^2]; n2 [1 +-2s 4s^2]; [0 0 1])
__device__ __forceinline__ void wino43_gt3(float m0, float m1, float m2, float m3, float m4, float m5, float (&o)[3]) {
  constexpr float n0 = 1.0f / (4.0f * kW43S4), n1 = -kW43S / (6.0f * kW43S4), n2 = kW43_2S / (24.0f * kW43S4);
  const float p1 = m1 + m2, q1 = m1 - m2, p2 = m3 + m4, q2 = m3 - m4;
  o[0] = fmaf(n0, m0, fmaf(n1, p1, n2 * p2));
  o[1] = fmaf(n1 * kW43S, q1, (n2 * kW43_2S) * q2);
  o[2] = fmaf(n1 * kW43S2, p1, fmaf(n2 * kW43_4S2, p2, m5));
}

// part: [splitK][9 taps][Cout][Cin]
template <int SW>
__device__ __forceinline__ void wgrad_wino43_body(const WgradWinoArgs& a) {
  using Sw = WgradWino43Sw;
  constexpr bool kRaw3 = (SW & Sw::Raw3) != 0, kEarly = (SW & Sw::EarlyDy) != 0, kEarly2 = (SW & Sw::EarlyDy2) != 0, kTl = (SW & Sw::Timeline) != 0;
  constexpr bool kDma = !(SW & Sw::NoDma), kDy = !(SW & Sw::NoDy), kYh = !(SW & Sw::NoYh), kV = !(SW & Sw::NoV), kMfma = !(SW & Sw::NoMfma),
                 kOpReads = !(SW & Sw::NoOpReads);
  using Cfg = WgradWino43CfgT<kRaw3 ? 3 : 2>;
  constexpr int RQ = Cfg::RQ, YH = Cfg::YH_STAGE, VS = Cfg::V_STAGE, RAW_STAGE = Cfg::RAW_STAGE, RAWN = Cfg::RAWN;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;                                    // two stages
  float* v_s = lds + 2 * YH;
  float* raw_s = lds + 2 * (YH + VS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int nIB = (Cin + Cfg::CB - 1) / Cfg::CB;
  const int kW = W >> 4, nTR = H >> 2, kpi = nTR * kW;  // strips per tile row, tile rows, strips per image
  const long strips = (long)a.N * kpi;
  // block -> (co block, ci block, K share)
  const int b = blockIdx.x;
  const int ks = b % a.splitK, bb = b / a.splitK, ibk = bb % nIB, mbk = bb / nIB;
  const long e0 = strips * ks / a.splitK, e1 = strips * (ks + 1) / a.splitK;
  const int nsteps = (int)(e1 - e0);
  const int co0 = mbk * Cfg::MB, ci0 = ibk * Cfg::CB;

  // MFMA role: wave = (co block cb, ci block ib)
  const int cb = swave & 3, ib = swave >> 2;
  const int a_lane = cb * 256 + lane * 4, b_lane = ib * 256 + lane * 4;      // + quad * 1024 / + quad * 512
  f32x4 acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // A strip cursor: (image, tile row, strip column) of a flat strip index and what a step derives from them, advanced WITHOUT divisions or
  // multiplications -- `off` = the strip's origin 4 tr W + 16 kc (floats inside a channel plane), `img` = the image's element offset
  // n Cin HW / n Cout HW kept as running pointers.  One s_add / s_cmp / branch per cursor on the common path.
  struct Cur { int tr, kc, off; const float* ximg; const float* zimg; int left; };      // left: strips of the slice from this one on (<= 0: past the end)
  auto cur_at = [&](long e) {
    Cur c;
    const int n = (int)(e / kpi);
    const int rem = (int)(e - (long)n * kpi);
    c.tr = rem / kW;
    c.kc = rem - c.tr * kW;
    c.off = 4 * c.tr * W + 16 * c.kc;
    c.ximg = a.x + (size_t)n * Cin * HW;
    c.zimg = a.dz + (size_t)n * Cout * HW;
    c.left = nsteps;
    return c;
  };
  auto cur_next = [&](Cur& c) {
    --c.left;
    c.off += 16;
    if (++c.kc >= kW) {
      TNV3_NO_IF_CONVERSION();
      c.kc = 0;
      c.off += 3 * W;
      if (++c.tr >= nTR) { c.tr = 0; c.off = 0; c.ximg += (size_t)Cin * HW; c.zimg += (size_t)Cout * HW; }
    }
  };

  // T: the transforms' strip (one ahead of the MFMAs'); Y: the dY loads' (two ahead); D: the X DMA's (RAWN ahead)
  Cur cT = cur_at(e0), cY = cT, cD = cT;

  // Everything below is instantiated per wave group: waves 0-3 transform dY (Yh), waves 4-5 / 6-7 the upper / lower half patches of X (V).
  auto body = [&](auto grpc, auto rhc) {
  constexpr int GRP = decltype(grpc)::value, RH = decltype(rhc)::value;

  // ---- X raw DMA (waves 4-7: the group with the lighter transform): slot e = (tid - 256) + i * 256 -> (channel ch, row r, piece q) of
  // [32][RPLANE]; a slot's offset inside the image is a per-lane constant + the strip's origin, so a step costs one v_add per piece (and a
  // few selects in the strips of the first / last tile row, whose rows above / below the image must read as zeros).
  constexpr int NDMA = (Cfg::RAW_SLOTS + 255) / 256;                          // 5 (the fifth: wave 4)
  constexpr int DMA_LAST_WAVES = (Cfg::RAW_SLOTS - (NDMA - 1) * 256 + 63) / 64;
  unsigned dma_c[NDMA];
  unsigned dma_rows = 0;                                                      // 3 bits per piece: its raw row r (7: a padding slot)
  const int wbase1 = __builtin_amdgcn_readfirstlane((wave & 3) * 64);
  if constexpr (GRP == 1) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int e = (tid & 255) + i * 256;
      const int ch = e / Cfg::RPLANE, rem = e - ch * Cfg::RPLANE;
      const int r = rem / RQ, q = rem - r * RQ;
      const bool ok = e < Cfg::RAW_SLOTS && rem < 6 * RQ;
      dma_c[i] = ok ? (unsigned)((ci0 + ch) * HW + r * W + 4 * q) * 4u : kDmaOob;      // (channels >= Cin: beyond the descriptor's range = zeros)
      dma_rows |= (unsigned)(ok ? r : 7) << (3 * i);
    }
  }
  auto dma_x = [&](const Cur& c, int stage) {
    if constexpr (GRP == 1) {
      const bool live = c.left > 0;
      const tnv3_rsrc_t rx = tnv3_make_rsrc(c.ximg, (unsigned)Cin * (unsigned)HW * 4u);
      // origin of the raw strip: row 4 tr - 1, column 16 kc - 1 (negative in the first tile row: wrap-around arithmetic; what stays negative -- the
      // piece before the image's first element -- lands beyond the descriptor's range = zeros, and v_piece patches it)
      const unsigned o4 = (unsigned)(c.off - W - 1) * 4u;
      const bool edge = c.tr == 0 || c.tr == nTR - 1 || !live;
      float* dst = raw_s + stage * RAW_STAGE + wbase1 * 4;
      if (!edge) {                                        // the common step: one v_add per piece, nothing else between the DMA instructions
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
          if (i < NDMA - 1 || (swave & 3) < DMA_LAST_WAVES) tnv3_buf_dma16(rx, dst + i * 1024, dma_c[i] + o4);
      } else {
        TNV3_NO_IF_CONVERSION();
        unsigned rows = dma_rows;
        TNV3_OPAQUE_V(rows);                              // (the row tests are made here, not hoisted into ten loop-invariant lane masks)
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
          if (i < NDMA - 1 || (swave & 3) < DMA_LAST_WAVES) {
            const unsigned r = (rows >> (3 * i)) & 7u;
            const bool out = !live || (c.tr == 0 && r == 0) || (c.tr == nTR - 1 && r == 5);
            tnv3_buf_dma16(rx, dst + i * 1024, out ? kDmaOob : dma_c[i] + o4);      // (a padding slot stays out of range: 2^31 - 4 (W + 1) >= the descriptor's size, wgrad_wino43_supported)
          }
        }
      }
    }
  };

  // ---- group 0: thread = (co = co block `swave`, lane & 15; tile = lane >> 4)
  f32x4 dy[4];
  const unsigned dy_c = (unsigned)((co0 + 16 * (swave & 3) + (lane & 15)) * HW + 4 * (lane >> 4)) * 4u;
  auto load_dy = [&](const Cur& c) {
    if constexpr (GRP == 0) {
      const bool live = c.left > 0;                                           // (past the end of the slice: any in-range tile; never multiplied)
      const tnv3_rsrc_t rz = tnv3_make_rsrc(live ? c.zimg : a.dz, (unsigned)Cout * (unsigned)HW * 4u);
      const unsigned so = live ? (unsigned)c.off * 4u : 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) dy[r] = tnv3_buf_load_f4(rz, dy_c, so + (unsigned)(r * W) * 4u);
    }
  };
  float ty[6][4];
  auto yh_piece = [&](auto pc, float* dst) {           // dst = yh stage + (swave & 3) * 256 + lane * 4
    constexpr int P = decltype(pc)::value;
    if constexpr (P < 4) {                              // first pass down the tile's column P
      float z[6];
      wino43_a6(dy[0][P], dy[1][P], dy[2][P], dy[3][P], z);
#pragma unroll
      for (int i = 0; i < 6; ++i) ty[i][P] = z[i];
    } else if constexpr (P < 7) {                       // second pass along rows i' and i' + 3: quads 3 i' .. 3 i' + 2
      constexpr int ip = P - 4;
      float za[6], zb[6];
      wino43_a6(ty[ip][0], ty[ip][1], ty[ip][2], ty[ip][3], za);
      wino43_a6(ty[ip + 3][0], ty[ip + 3][1], ty[ip + 3][2], ty[ip + 3][3], zb);
#pragma unroll
      for (int m = 0; m < 3; ++m) *reinterpret_cast<f32x4*>(dst + (3 * ip + m) * 1024) = f32x4{za[2 * m], za[2 * m + 1], zb[2 * m], zb[2 * m + 1]};
    }
  };
  // ---- group 1: thread = (ci block swave & 1, ci = lane & 15; tile = lane >> 4; row half RH: a compile-time parameter of the instantiation)
  const int v_ib = swave & 1;
  const int v_t = lane >> 4, v_ci = 16 * v_ib + (lane & 15);
  const int v_src = v_ci * (Cfg::RPLANE * 4) + (RH * RQ + v_t) * 4;        // + row * 20 floats; second piece + 4
  const int v_dst = v_ib * 256 + lane * 4 + 2 * RH;                        // + quad * 512
  f32x4 tq0[5];
  wf2 tq1[5];
  float tt[3][6];
  bool zl = false, zr = false, fix_corner = false;
  auto v_piece = [&](auto pc, const float* raw, float* dst, const Cur& c) {
    constexpr int P = decltype(pc)::value;
    if constexpr (P == 0) {
#pragma unroll
      for (int r = 0; r < 5; ++r) tq0[r] = *reinterpret_cast<const f32x4*>(raw + r * (RQ * 4));
      if (fix_corner) {                                 // the piece before the image's first element (channel 0, row 0): patch row 1 of tile 0, both row halves
        TNV3_NO_IF_CONVERSION();
        if ((lane & 15) == 0 && v_t == 0 && v_ib == 0) {
          const tnv3_rsrc_t ri = tnv3_make_rsrc(c.ximg, (unsigned)Cin * (unsigned)HW * 4u);
          const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
          const f32x4 fx = f32x4{0.0f, x[0], x[1], x[2]};
          if constexpr (RH) tq0[0] = fx; else tq0[1] = fx;
        }
      }
    } else if constexpr (P == 1) {
#pragma unroll
      for (int r = 0; r < 5; ++r) tq1[r] = *reinterpret_cast<const wf2*>(raw + r * (RQ * 4) + 4);
    } else if constexpr (P < 8) {                       // first pass, down patch column cc
      constexpr int cc = P - 2;
      float x[5], o[3];
#pragma unroll
      for (int r = 0; r < 5; ++r) x[r] = cc < 4 ? tq0[r][cc < 4 ? cc : 0] : tq1[r][cc < 4 ? 0 : cc - 4];
      if constexpr (RH) {
        const float d[6] = {0.0f, x[0], x[1], x[2], x[3], x[4]};
        wino43_bt_half<1>(d, o);
      } else {
        const float d[6] = {x[0], x[1], x[2], x[3], x[4], 0.0f};
        wino43_bt_half<0>(d, o);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float v = o[r];
        if constexpr (cc == 0) v = zl ? 0.0f : v;
        if constexpr (cc == 5) v = zr ? 0.0f : v;
        tt[r][cc] = v;
      }
    } else if constexpr (P < 11) {                      // second pass along row 3 RH + r: quads 3 r .. 3 r + 2, floats 2 RH, 2 RH + 1
      constexpr int r = P - 8;
      float o[6];
      wino43_bt_full(tt[r], o);
#pragma unroll
      for (int m = 0; m < 3; ++m) *reinterpret_cast<wf2*>(dst + (3 * r + m) * 512) = wf2{o[2 * m], o[2 * m + 1]};
    }
  };
  constexpr int NPIECE = GRP == 0 ? 7 : 11;
  auto set_v_flags = [&](const Cur& c) {
    zl = c.kc == 0 && v_t == 0;
    zr = c.kc == kW - 1 && v_t == 3;
    fix_corner = c.left > 0 && c.off == 0 && ci0 == 0;
  };
  auto transform_all = [&](int stage, int raw_stage, const Cur& c) {      // (prologue: not interleaved)
    if constexpr (GRP == 0) {
      float* dst = yh_s + stage * YH + (swave & 3) * 256 + lane * 4;
      wino43s_for<0, NPIECE>([&](auto pc) { yh_piece(pc, dst); });
    } else {
      set_v_flags(c);
      const float* raw = raw_s + raw_stage * RAW_STAGE + v_src;
      float* dst = v_s + stage * VS + v_dst;
      wino43s_for<0, NPIECE>([&](auto pc) { v_piece(pc, raw, dst, c); });
    }
  };
  auto full_barrier = [&]() {
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: operands of step 0 in stage 0, the loads of the next steps under way
  dma_x(cD, 0);
  cur_next(cD);
  load_dy(cY);
  cur_next(cY);
  full_barrier();
  transform_all(0, 0, cT);
  cur_next(cT);
#pragma unroll
  for (int rs = 1; rs < RAWN; ++rs) {
    dma_x(cD, rs);
    cur_next(cD);
  }
  load_dy(cY);                                          // (the transform above has consumed step 0's tile)
  cur_next(cY);
  full_barrier();

  unsigned long long tl_busy = 0, tl_wait = 0, tl_prev = 0;
  if constexpr (kTl) tl_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long tl_first = tl_prev;

  // ---- steps
  int rcur = 0;                                         // raw stage of the step whose MFMAs run: consumed, the DMA's target
  for (int sg = 0; sg < nsteps; ++sg) {
    const int st = sg & 1, sn = st ^ 1;
    const int rnext = rcur + 1 < RAWN ? rcur + 1 : 0;
    const float* A = yh_s + st * YH + a_lane;
    const float* B = v_s + st * VS + b_lane;
    const float* raw = raw_s + rnext * RAW_STAGE + v_src;   // raw(sigma + 1)
    float* ydst = yh_s + sn * YH + (swave & 3) * 256 + lane * 4;
    float* vdst = v_s + sn * VS + v_dst;
    if constexpr (GRP == 1) set_v_flags(cT);
    f32x4 aq[2], bq[2];
    if constexpr (kOpReads) {
      aq[0] = *reinterpret_cast<const f32x4*>(A);
      bq[0] = *reinterpret_cast<const f32x4*>(B);
    } else {
      aq[0] = aq[1] = bq[0] = bq[1] = f32x4{1.0f, 0.5f, 0.25f, 2.0f};
    }
    // raw(sigma + RAWN) -> the raw stage whose strip the transform of the previous step has consumed
    if constexpr (kDma) dma_x(cD, rcur);
    wino43s_for<0, 9>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q + 1 < 9 && kOpReads) {
        aq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(A + (q + 1) * 1024);
        bq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(B + (q + 1) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kMfma) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q & 1][e], bq[q & 1][e], acc[4 * q + e], 0, 0, 0);
      } else {
        if (sg == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * q + e] = aq[q & 1] + bq[q & 1];
        }
#ifndef TNV3_EMU
        asm volatile("" : : "v"(aq[q & 1]), "v"(bq[q & 1]));      // (the operand reads stay)
#endif
      }
      // the transform of step sigma + 1, a piece (or two) behind every quad
      if constexpr (GRP == 0) {
        if constexpr (kYh) {
          if constexpr (kEarly2) {                      // the column passes behind the first two quads: dy is free after quad 1
            if constexpr (q < 2) {
              yh_piece(std::integral_constant<int, 2 * q>{}, ydst);
              yh_piece(std::integral_constant<int, 2 * q + 1>{}, ydst);
            } else if constexpr (q < 5) {
              yh_piece(std::integral_constant<int, q + 2>{}, ydst);
            }
          } else {
            if constexpr (q < 7) yh_piece(std::integral_constant<int, q>{}, ydst);
          }
        }
        if constexpr (kDy && ((kEarly2 && q == 1) || (kEarly && !kEarly2 && q == 3))) load_dy(cY);
      } else if constexpr (kV) {
        if constexpr (q < 2) {                          // eleven pieces behind nine quads, in order: the two reads, then one piece each
          v_piece(std::integral_constant<int, 2 * q>{}, raw, vdst, cT);
          v_piece(std::integral_constant<int, 2 * q + 1>{}, raw, vdst, cT);
        } else {
          v_piece(std::integral_constant<int, q + 2>{}, raw, vdst, cT);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    unsigned long long tl_b = 0;
    if constexpr (kTl) tl_b = __builtin_amdgcn_s_memtime();
    // End of the step.  Waves 4-7: their pieces of raw(sigma + 2) -- what the next step transforms -- must have landed: with two raw stages
    // that is this step's DMA (vmcnt 0), with three the previous step's (vmcnt retires in order: all but this step's pieces).  Waves 0-3 issue
    // no DMA; their dY loads are waited for where the tile is first read (the compiler's own count).
    if constexpr (GRP == 0) {
      if constexpr (kDy && !kEarly && !kEarly2) load_dy(cY);
    } else {
      if constexpr (kRaw3 && kDma) {
        if ((swave & 3) < DMA_LAST_WAVES) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(NDMA));
        else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(NDMA - 1));
      } else {
        __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
      }
    }
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kTl) {
      const unsigned long long tl_c = __builtin_amdgcn_s_memtime();
      tl_busy += tl_b - tl_prev;
      tl_wait += tl_c - tl_b;
      tl_prev = tl_c;
    }
    cur_next(cT);
    cur_next(cY);
    cur_next(cD);
    rcur = rnext;
  }
  if constexpr (kTl) {
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* tl = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.zeros)) + swave * 8;
      tl[0] = tl_busy; tl[1] = tl_wait; tl[2] = (unsigned long long)nsteps; tl[3] = tl_prev - tl_first;
    }
  }

  // ---- epilogue: dg = G^T S G per (co, ci); acc[4 q + e] = S[(q / 3) + 3 (e >> 1)][2 (q % 3) + (e & 1)]; the slab's tap planes
  {
    const int g = lane >> 4, ci = ci0 + 16 * ib + (lane & 15);
    float* slab = a.part + (size_t)ks * 9 * Cout * Cin;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * cb + 4 * g + r;
      float p[3][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float m[6], o[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = acc[4 * (3 * (i % 3) + j / 2) + 2 * (i / 3) + (j & 1)][r];
        wino43_gt3(m[0], m[1], m[2], m[3], m[4], m[5], o);
        p[0][j] = o[0]; p[1][j] = o[1]; p[2][j] = o[2];
      }
#pragma unroll
      for (int aa = 0; aa < 3; ++aa) {
        float o[3];
        wino43_gt3(p[aa][0], p[aa][1], p[aa][2], p[aa][3], p[aa][4], p[aa][5], o);
        if (ci < Cin) {
#pragma unroll
          for (int bb2 = 0; bb2 < 3; ++bb2) slab[((size_t)(3 * aa + bb2) * Cout + co) * Cin + ci] = o[bb2];
        }
      }
    }
  }
  };
  if (swave < 4) body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  else if (swave < 6) body(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
}

inline __global__ void __launch_bounds__(WgradWino43Cfg::NT) wgrad_wino43_kernel(const WgradWinoArgs a) { wgrad_wino43_body<0>(a); }
#ifdef TNV3_DIAG
template <int SW>
__global__ void __launch_bounds__(WgradWino43Cfg::NT) wgrad_wino43_twin_kernel(const WgradWinoArgs a) { wgrad_wino43_body<SW>(a); }
#endif

// dw[co][ci][tap] = sum over the K shares of part[k][tap][co][ci]: fp64, in a fixed order (four contiguous quarters of the shares,
// then the quarters in order) -- deterministic.  Block = 64 elements x 4 quarters: the loads of a wave are contiguous.
inline __global__ void __launch_bounds__(256) wgrad_wino43_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout, int Cin, int splitK) {
  __shared__ double red[4][64];
  const long n = (long)Cout * Cin, total = 9 * n;
  const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int kq = (splitK + 3) / 4;
  const int k0 = p * kq, k1 = (k0 + kq < splitK) ? k0 + kq : splitK;
  for (long e0 = (long)blockIdx.x * 64; e0 < total; e0 += (long)gridDim.x * 64) {
    const long e = e0 + el;
    double s = 0.0;
    if (e < total) {
      int k = k0;
      for (; k + 4 <= k1; k += 4) {
        const float v0 = part[(size_t)k * total + e], v1 = part[(size_t)(k + 1) * total + e], v2 = part[(size_t)(k + 2) * total + e], v3 = part[(size_t)(k + 3) * total + e];
        s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
      }
      for (; k < k1; ++k) s += (double)part[(size_t)k * total + e];
    }
    red[p][el] = s;
    __syncthreads();
    if (p == 0 && e < total) {
      const double t = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
      const int tap = (int)(e / n);
      const long cc = e - (long)tap * n;
      dw[cc * 9 + tap] = (float)t;
    }
    __syncthreads();
  }
}

}  // namespace tnv3
