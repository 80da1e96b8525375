// wgrad_up2x_wino43_mfma.h -- weight gradient of the UPSAMPLED half of a decoder-entry layer (model.py:65,67,69: conv3x3 over
// nn.Upsample(2)(x_low)) in the 25-of-36 Winograd F(4x4, 3x3) form, from the low-resolution tensor (tnv3_conv3x3_wgrad_up2x, round 5).
//
// The forward of that half is Y = A'^T [ (G' g G'^T) .* (B' l B'^T) ] A' over the five indices (0, 1, e, o, 5) per axis
// (conv3x3_wino43s_mfma.h, MODE 1: interpolation points (0, +-1, +-sqrt B, inf), B = kU43B; the point -1 vanishes because the upsampled
// signal's polynomial is (1 + x) l(x^2)); l = the 4x4 LOW-resolution patch under a 4x4 output tile.  Its adjoint in g:
//     dg[co][ci] = G'^T [ sum over images and 4x4 tiles of  (A' dY A'^T) .* (B' l B'^T) ] G'
// -- 25 GEMMs S_xi[co][ci] = sum_tiles Yh_xi[co][tile] * V_xi[ci][tile] with K = all tiles of the batch: 25 products per (co, ci, 4x4
// tile) = 6.25 per low-resolution pixel where the 9-GEMM F(2x2) form (wgrad_up2x_wino_mfma_kernel) multiplies 9 and the direct form 36.
// Both transforms are a handful of adds: A' = [1 0 0 0; 1 1 1 1; 1 0 B 0; 0 1 0 B; 0 0 0 1] (5 operations per column), B' as in MODE 1.
//
// Mapping: wgrad_wino43_mfma.h's.  MFMA 16x16x4: M = 16 output channels, N = 16 input channels, K = 4 tiles (a strip of 4 x 16 output
// pixels = 2 x 8 low-resolution pixels = one STEP); a wave keeps all 25 xi of its 16 x 16 block (100 accumulator registers), two waves
// per SIMD; workgroup = 64 co x 32 ci x a contiguous share of the strips (split-K).  xi = 5 i' + j' over (0, 1, e, o, 5)^2 in quads of
// four (seven quads, the last holds xi 24 alone) in MFMA lane order.
//   * Yh = A' dY A'^T: waves 0-3, thread = (co, tile): the 4x4 tile of dZ straight from global memory into registers, seven 16-byte stores.
//   * V = B' l B'^T: waves 4-7, thread = (ci, tile, row half): the raw low-resolution strip (32 channels x 4 rows x 12 columns from
//     column 8 k - 1) arrives by LDS-DMA two steps ahead.  Row half 0 = indices 0, 1 (xi 0-9), row half 1 = e, o (the same value twice)
//     and 5 (xi 10-24).  The columns left / right of the image are zeroed after the first pass, the one piece per image that would start
//     before it (channel 0, row 0) is patched from a guarded load.
//   * Epilogue: G'^T S G' per (co, ci) in registers, the 3x3 result to the workgroup's slab [k][tap][co][ci]; the slabs are summed in
//     fp64, in a fixed order, by wgrad_wino43_fold_kernel (which writes the [Cout][C0][9] block wgrad_up2x_join_kernel reads).
// Needs Cout % 64 == 0, Hl % 2 == 0, Wl % 8 == 0; any C0 (a partial last block of 32 input channels reads zeros).  Deterministic.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "wgrad_wino43_mfma.h"

namespace tnv3 {

struct WgradUp2xWino43Cfg {
  static constexpr int NT = 512, MB = 64, CB = 32, NQ = 7;
  static constexpr int YH_STAGE = NQ * 4 * 64 * 4;        // [quad 7][co block 4][lane 64][4]: 28 KB
  static constexpr int V_STAGE = NQ * 2 * 64 * 4;         // [quad 7][ci block 2][lane 64][4]: 14 KB
  static constexpr int RQ = 3;                            // pieces per raw row: low-resolution columns 8 k - 1 .. 8 k + 10
  static constexpr int RPLANE = 13;                       // pieces per channel plane (4 rows x 3 = 12 used): 52 floats -- the 16 channels x 2 tiles of a
                                                          // 32-lane ds_read_b64 group hit 32 different 8-byte bank pairs
  static constexpr int RAW_SLOTS = CB * RPLANE;           // 416 pieces per stage: one DMA instruction on waves 0-6
  static constexpr int DMA_WAVES = (RAW_SLOTS + 63) / 64;
  static constexpr int RAW_STAGE = DMA_WAVES * 64 * 4;    // floats (whole waves: the surplus lanes of wave 6 write zeros behind the last plane)
  static constexpr int LDS_FLOATS = 2 * (YH_STAGE + V_STAGE) + 2 * RAW_STAGE;      // 100,352 bytes
  static_assert(RAW_SLOTS <= NT && LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

// z = A' x along one axis (indices 0, 1, e, o, 5)
__device__ __forceinline__ void wino43u_a5(float x0, float x1, float x2, float x3, float (&z)[5]) {
  z[0] = x0;
  z[1] = (x0 + x1) + (x2 + x3);
  z[2] = fmaf(kU43B, x2, x0);
  z[3] = fmaf(kU43B, x3, x1);
  z[4] = x3;
}
// o = G'^T m along one axis (G' of wino43u_g_row)
__device__ __forceinline__ void wino43u_gt3(float m0, float m1, float me, float mo, float m5, float (&o)[3]) {
  constexpr float r0 = 1.0f / kU43B, r1 = 1.0f / (1.0f - kU43B), re0 = -1.0f / (kU43B * (kU43B - 1.0f)), rb = -1.0f / (kU43B - 1.0f),
                  rbb = -kU43B / (kU43B - 1.0f);
  const float t1 = r1 * m1;
  o[0] = fmaf(r0, m0, fmaf(re0, me, fmaf(rb, mo, t1)));
  o[1] = fmaf(rb, me + mo, t1);
  o[2] = fmaf(rb, me, fmaf(rbb, mo, t1)) + m5;
}

// part: [splitK][9 taps][Cout][C0]
// Round 6: the step's bookkeeping follows wgrad_wino43_mfma.h's diet (every instruction of a step costs ~4.6 cycles beside 1600 cycles of MFMAs
// per SIMD): the DMA slot's address is a per-lane constant + the strip's origin, the strip cursors advance by additions (no division, no
// multiplication by the image number), the row half of the V transform is a compile-time parameter of the wave group's instantiation.
inline __global__ void __launch_bounds__(WgradUp2xWino43Cfg::NT) wgrad_up2x_wino43_kernel(const WgradUp2xWinoArgs a) {
  using Cfg = WgradUp2xWino43Cfg;
  constexpr int RQ = Cfg::RQ, YH = Cfg::YH_STAGE, VS = Cfg::V_STAGE, RAW_STAGE = Cfg::RAW_STAGE, NQ = Cfg::NQ;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;                                    // two stages
  float* v_s = lds + 2 * YH;
  float* raw_s = lds + 2 * (YH + VS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int Hl = a.Hl, Wl = a.Wl, C0 = a.C0, Cout = a.Cout, LHW = Hl * Wl;
  const int H = 2 * Hl, W = 2 * Wl, HW = H * W;
  const int nIB = (C0 + Cfg::CB - 1) / Cfg::CB;
  const int kW = W >> 4, nTR = H >> 2, kpi = nTR * kW;  // strips per tile row, tile rows, strips per image
  const long strips = (long)a.N * kpi;
  const int b = blockIdx.x;
  const int ks = b % a.splitK, bb = b / a.splitK, ibk = bb % nIB, mbk = bb / nIB;
  const long e0 = strips * ks / a.splitK, e1 = strips * (ks + 1) / a.splitK;
  const int nsteps = (int)(e1 - e0);
  const int co0 = mbk * Cfg::MB, ci0 = ibk * Cfg::CB;

  const int cb = swave & 3, ib = swave >> 2;            // MFMA role: wave = (co block cb, ci block ib)
  const int a_lane = cb * 256 + lane * 4, b_lane = ib * 256 + lane * 4;      // + quad * 1024 / + quad * 512
  f32x4 acc[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);

  // strip cursor: `off` = 4 tr W + 16 kc (the strip's origin in a plane of dZ), `offl` = 2 tr Wl + 8 kc (in a plane of x_low), the images as
  // running pointers, `left` = strips of the slice from this one on
  struct Cur { int tr, kc, off, offl; const float* ximg; const float* zimg; int left; };
  auto cur_at = [&](long e) {
    Cur c;
    const int n = (int)(e / kpi);
    const int rem = (int)(e - (long)n * kpi);
    c.tr = rem / kW;
    c.kc = rem - c.tr * kW;
    c.off = 4 * c.tr * W + 16 * c.kc;
    c.offl = 2 * c.tr * Wl + 8 * c.kc;
    c.ximg = a.x_low + (size_t)n * C0 * LHW;
    c.zimg = a.dz + (size_t)n * Cout * HW;
    c.left = nsteps;
    return c;
  };
  auto cur_next = [&](Cur& c) {
    --c.left;
    c.off += 16;
    c.offl += 8;
    if (++c.kc >= kW) {
      TNV3_NO_IF_CONVERSION();
      c.kc = 0;
      c.off += 3 * W;                                   // (16 kW = W)
      c.offl += Wl;                                     // (8 kW = Wl)
      if (++c.tr >= nTR) { c.tr = 0; c.off = 0; c.offl = 0; c.ximg += (size_t)C0 * LHW; c.zimg += (size_t)Cout * HW; }
    }
  };

  // ---- x_low raw DMA (waves 0-6, one piece each): slot e = tid -> (channel ch, low-resolution row r of 2 tr - 1 .. 2 tr + 2, piece q) of [32][RPLANE]
  unsigned dma_c = kDmaOob;
  int dma_r = 7;
  {
    const int e = tid;
    const int ch = e / Cfg::RPLANE, rem = e - ch * Cfg::RPLANE;
    const int r = rem / RQ, q = rem - r * RQ;
    if (e < Cfg::RAW_SLOTS && rem < 4 * RQ) {
      dma_c = (unsigned)((ci0 + ch) * LHW + r * Wl + 4 * q) * 4u;      // (channels >= C0: beyond the descriptor's range = zeros)
      dma_r = r;
    }
  }
  auto dma_x = [&](const Cur& c, int stage) {
    if (swave < Cfg::DMA_WAVES) {
      const bool live = c.left > 0;
      const tnv3_rsrc_t rx = tnv3_make_rsrc(c.ximg, (unsigned)C0 * (unsigned)LHW * 4u);
      // origin: low-resolution row 2 tr - 1, column 8 kc - 1 (wrap-around arithmetic: what stays negative lands beyond the descriptor's range =
      // zeros; a padding slot at 2^31 + origin stays out of range as well: wgrad_up2x_wino43_supported keeps an image 4 (Wl + 1) bytes below 2^31)
      unsigned vo = dma_c + (unsigned)(c.offl - Wl - 1) * 4u;
      if (c.tr == 0 || c.tr == nTR - 1 || !live) {
        TNV3_NO_IF_CONVERSION();
        if (!live || (c.tr == 0 && dma_r == 0) || (c.tr == nTR - 1 && dma_r == 3)) vo = kDmaOob;
      }
      tnv3_buf_dma16(rx, raw_s + stage * RAW_STAGE + wbase * 4, vo);
    }
  };

  Cur cT = cur_at(e0), cD = cT;                         // T: the transforms' strip (one ahead of the MFMAs'); D: the loads' (two ahead)

  auto body = [&](auto grpc, auto rhc) {
  constexpr int GRP = decltype(grpc)::value, RH = decltype(rhc)::value;
  // ---- group 0: thread = (co = co block `swave`, lane & 15; tile = lane >> 4)
  f32x4 dy[4];
  const unsigned dy_c = (unsigned)((co0 + 16 * (swave & 3) + (lane & 15)) * HW + 4 * (lane >> 4)) * 4u;
  auto load_dy = [&](const Cur& c) {
    if constexpr (GRP == 0) {
      const bool live = c.left > 0;                     // (past the end of the slice: any in-range tile; never multiplied)
      const tnv3_rsrc_t rz = tnv3_make_rsrc(live ? c.zimg : a.dz, (unsigned)Cout * (unsigned)HW * 4u);
      const unsigned so = live ? (unsigned)c.off * 4u : 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) dy[r] = tnv3_buf_load_f4(rz, dy_c, so + (unsigned)(r * W) * 4u);
    }
  };
  float ty[5][4];
  float yv[25];
  auto yh_piece = [&](auto pc, float* dst) {           // dst = yh stage + (swave & 3) * 256 + lane * 4
    constexpr int P = decltype(pc)::value;
    if constexpr (P < 2) {                              // first pass down the tile's columns 2 P, 2 P + 1
#pragma unroll
      for (int cc = 2 * P; cc < 2 * P + 2; ++cc) {
        float z[5];
        wino43u_a5(dy[0][cc], dy[1][cc], dy[2][cc], dy[3][cc], z);
#pragma unroll
        for (int i = 0; i < 5; ++i) ty[i][cc] = z[i];
      }
    } else if constexpr (P == 2) {                      // second pass along the five rows
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        float z[5];
        wino43u_a5(ty[i][0], ty[i][1], ty[i][2], ty[i][3], z);
#pragma unroll
        for (int j = 0; j < 5; ++j) yv[5 * i + j] = z[j];
      }
    } else if constexpr (P < 3 + NQ) {                  // one quad per piece
      constexpr int q = P - 3;
      *reinterpret_cast<f32x4*>(dst + q * 1024) = f32x4{yv[4 * q], q < 6 ? yv[4 * q + 1] : 0.0f, q < 6 ? yv[4 * q + 2] : 0.0f, q < 6 ? yv[4 * q + 3] : 0.0f};
    }
  };
  // ---- group 1: thread = (ci block swave & 1, ci = lane & 15; tile = lane >> 4; row half RH: a compile-time parameter of the instantiation)
  const int v_ib = swave & 1;
  const int v_t = lane >> 4, v_ci = 16 * v_ib + (lane & 15);
  const int v_src = v_ci * (Cfg::RPLANE * 4) + (RH * RQ) * 4 + 2 * v_t;         // + row * 12 floats; columns 2, 3 of the patch at + 2
  const int v_dst = v_ib * 256 + lane * 4;                                     // + quad * 512
  wf2 ulo[3], uhi[3];
  float uc[2][4], uw[2][4];
  bool zl = false, zr = false, fix_corner = false;
  auto v_piece = [&](auto pc, const float* raw, float* dst, const Cur& c) {
    constexpr int P = decltype(pc)::value;
    if constexpr (P == 0) {                             // the thread's three low-resolution rows: (a, b, c) for row half 0, (b, c, e) for 1
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        ulo[r] = *reinterpret_cast<const wf2*>(raw + r * (RQ * 4));
        uhi[r] = *reinterpret_cast<const wf2*>(raw + r * (RQ * 4) + 2);
      }
      if (fix_corner) {                                 // the piece before the image's first element (channel 0, low-resolution row 0, columns -1 .. 2):
        TNV3_NO_IF_CONVERSION();
        if ((lane & 15) == 0 && v_t < 2 && v_ib == 0) { // all of tile 0's row and the first half of tile 1's; row 0 is patch row 1 (half 0) / 0 (half 1)
          const tnv3_rsrc_t ri = tnv3_make_rsrc(c.ximg, (unsigned)C0 * (unsigned)LHW * 4u);
          const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
          const bool t0 = v_t == 0;
          const wf2 nlo = wf2{t0 ? 0.0f : x[1], t0 ? x[0] : x[2]};
          if constexpr (RH) { ulo[0] = nlo; if (t0) uhi[0] = wf2{x[1], x[2]}; }
          else { ulo[1] = nlo; if (t0) uhi[1] = wf2{x[1], x[2]}; }
        }
      }
    } else if constexpr (P == 1) {                      // down the patch's four columns
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float r0 = x < 2 ? ulo[0][x & 1] : uhi[0][x & 1], r1 = x < 2 ? ulo[1][x & 1] : uhi[1][x & 1], r2 = x < 2 ? ulo[2][x & 1] : uhi[2][x & 1];
        float c0, c1;
        if constexpr (RH == 0) { c0 = fmaf(kU43B, r0, fmaf(-kU43B1, r1, r2)); c1 = fmaf(-kU43B, r1, r2); }      // (a, b, c) -> B a - (1 + B) b + c,  c - B b
        else { c0 = r0 - r1; c1 = fmaf(kU43B, r0, fmaf(-kU43B1, r1, r2)); }                                      // (b, c, e) -> b - c,  B b - (1 + B) c + e
        if (x == 0) { c0 = zl ? 0.0f : c0; c1 = zl ? 0.0f : c1; }
        if (x == 3) { c0 = zr ? 0.0f : c0; c1 = zr ? 0.0f : c1; }
        uc[0][x] = c0; uc[1][x] = c1;
      }
    } else if constexpr (P == 2 || P == 3) {            // along a row: the values of indices 0, 1, e = o, 5
      constexpr int k = P - 2;
      const float p0 = uc[k][0], p1 = uc[k][1], p2 = uc[k][2], p3 = uc[k][3];
      uw[k][0] = fmaf(kU43B, p0, fmaf(-kU43B1, p1, p2));
      uw[k][1] = fmaf(-kU43B, p1, p2);
      uw[k][2] = p1 - p2;
      uw[k][3] = fmaf(kU43B, p1, fmaf(-kU43B1, p2, p3));
    } else if constexpr (P == 4) {                      // xi = 5 i' + j'; j' = (0, 1, e, o, 5) -> the row pass's values (0, 1, 2, 2, 3)
      const float* r0 = uw[0];
      const float* r1 = uw[1];
      if constexpr (RH == 0) {                          // rows i' = 0 (xi 0-4) and 1 (xi 5-9): quads 0, 1 and the first half of quad 2
        *reinterpret_cast<f32x4*>(dst) = f32x4{r0[0], r0[1], r0[2], r0[2]};
        *reinterpret_cast<f32x4*>(dst + 512) = f32x4{r0[3], r1[0], r1[1], r1[2]};
        *reinterpret_cast<wf2*>(dst + 2 * 512) = wf2{r1[2], r1[3]};
      } else {                                          // rows i' = e, o (both r0: xi 10-14, 15-19) and 5 (r1: xi 20-24)
        *reinterpret_cast<wf2*>(dst + 2 * 512 + 2) = wf2{r0[0], r0[1]};
        *reinterpret_cast<f32x4*>(dst + 3 * 512) = f32x4{r0[2], r0[2], r0[3], r0[0]};
        *reinterpret_cast<f32x4*>(dst + 4 * 512) = f32x4{r0[1], r0[2], r0[2], r0[3]};
        *reinterpret_cast<f32x4*>(dst + 5 * 512) = f32x4{r1[0], r1[1], r1[2], r1[2]};
        *reinterpret_cast<f32x4*>(dst + 6 * 512) = f32x4{r1[3], 0.0f, 0.0f, 0.0f};
      }
    }
  };
  constexpr int NPIECE = GRP == 0 ? 3 + NQ : 5;
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

  // ---- prologue: operands of step 0 in stage 0, the loads of step 1 under way
  dma_x(cD, 0);
  load_dy(cD);
  full_barrier();
  transform_all(0, 0, cT);
  cur_next(cT);
  cur_next(cD);
  dma_x(cD, 1);
  load_dy(cD);
  cur_next(cD);
  full_barrier();

  // ---- steps
  for (int sg = 0; sg < nsteps; ++sg) {
    const int st = sg & 1, sn = st ^ 1;
    const float* A = yh_s + st * YH + a_lane;
    const float* B = v_s + st * VS + b_lane;
    const float* raw = raw_s + sn * RAW_STAGE + v_src;   // raw(sigma + 1): requested one step ago
    float* ydst = yh_s + sn * YH + (swave & 3) * 256 + lane * 4;
    float* vdst = v_s + sn * VS + v_dst;
    if constexpr (GRP == 1) set_v_flags(cT);
    f32x4 aq[2], bq[2];
    aq[0] = *reinterpret_cast<const f32x4*>(A);
    bq[0] = *reinterpret_cast<const f32x4*>(B);
    dma_x(cD, st);                                      // raw(sigma + 2) -> the raw stage whose strip the transform of the previous step has consumed
    wino43s_for<0, NQ>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q + 1 < NQ) {
        aq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(A + (q + 1) * 1024);
        bq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(B + (q + 1) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < 25)
          acc[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q & 1][e], bq[q & 1][e], acc[4 * q + e], 0, 0, 0);
      // the transform of step sigma + 1 behind the quads: ten pieces (group 0) / five (group 1) over seven slots
      if constexpr (GRP == 0) {
        if constexpr (q < 3) {
          yh_piece(std::integral_constant<int, q>{}, ydst);
        } else {
          yh_piece(std::integral_constant<int, 2 * q - 3>{}, ydst);                              // quads 0-1, 2-3, 4-5, 6
          if constexpr (2 * q - 2 < 3 + NQ) yh_piece(std::integral_constant<int, 2 * q - 2>{}, ydst);
        }
      } else {
        if constexpr (q < 5) v_piece(std::integral_constant<int, q>{}, raw, vdst, cT);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (GRP == 0) {
      // this wave's x piece (older than the four dY loads requested now) has landed once at most four loads are in flight
      load_dy(cD);
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(4));
    } else {
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    }
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur_next(cT);
    cur_next(cD);
  }

  // ---- epilogue: dg = G'^T S G' per (co, ci); acc[5 i' + j'] = S[i'][j']; the slab's tap planes
  {
    const int g = lane >> 4, ci = ci0 + 16 * ib + (lane & 15);
    float* slab = a.part + (size_t)ks * 9 * Cout * C0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * cb + 4 * g + r;
      float p[3][5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float o[3];
        wino43u_gt3(acc[j][r], acc[5 + j][r], acc[10 + j][r], acc[15 + j][r], acc[20 + j][r], o);
        p[0][j] = o[0]; p[1][j] = o[1]; p[2][j] = o[2];
      }
#pragma unroll
      for (int aa = 0; aa < 3; ++aa) {
        float o[3];
        wino43u_gt3(p[aa][0], p[aa][1], p[aa][2], p[aa][3], p[aa][4], o);
        if (ci < C0) {
#pragma unroll
          for (int bb2 = 0; bb2 < 3; ++bb2) slab[((size_t)(3 * aa + bb2) * Cout + co) * C0 + ci] = o[bb2];
        }
      }
    }
  }
  };
  if (swave < 4) body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  else if (swave < 6) body(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
}

}  // namespace tnv3
