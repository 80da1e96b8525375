// wgrad_wino43_r5_mfma.h -- libtnv3_diag.so / the emulator only: round 5's schedule of the F(4x4) weight-gradient kernel, kept as the same-session
// A/B reference of kernels/wgrad_wino43_mfma.h (whose header describes the algorithm; results are bit-identical).  What round 6 changed and why
// (profiles/r06_wgrad43_twins.json): with two waves per SIMD every instruction of a step -- vector, scalar, LDS, branch -- costs ~4.6 cycles of a
// step that holds 2304 cycles of MFMAs, and this schedule spent ~330 of them per wave: the X DMA's slot -> address arithmetic redone per step
// (~24 vector instructions per piece), three strip cursors with their carries and image multiplications, a run-time row half (12 scalar
// branches per step in the V transform).  Switches: WgradWino43Sw bits 0-9 as in the main header.
#pragma once
#include "wgrad_wino43_mfma.h"

#ifdef TNV3_DIAG
namespace tnv3 {

// part: [splitK][9 taps][Cout][Cin]
template <int SW>
__device__ __forceinline__ void wgrad_wino43_r5_body(const WgradWinoArgs& a) {
  using Sw = WgradWino43Sw;
  constexpr bool kRaw3 = (SW & Sw::Raw3) != 0, kEarly = (SW & Sw::EarlyDy) != 0, kEarly2 = (SW & Sw::EarlyDy2) != 0, kTl = (SW & Sw::Timeline) != 0;
  constexpr bool kDma = !(SW & Sw::NoDma), kDy = !(SW & Sw::NoDy), kYh = !(SW & Sw::NoYh), kV = !(SW & Sw::NoV), kMfma = !(SW & Sw::NoMfma),
                 kOpReads = !(SW & Sw::NoOpReads);
  using Cfg = WgradWino43CfgT<kRaw3 ? 3 : 2>;
  constexpr int NT = Cfg::NT, RQ = Cfg::RQ, YH = Cfg::YH_STAGE, VS = Cfg::V_STAGE, RAW_STAGE = Cfg::RAW_STAGE, RAWN = Cfg::RAWN;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* yh_s = lds;                                    // two stages
  float* v_s = lds + 2 * YH;
  float* raw_s = lds + 2 * (YH + VS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int nIB = (Cin + Cfg::CB - 1) / Cfg::CB, nMB = Cout / Cfg::MB;
  const int kW = W >> 4, kpi = (H >> 2) * kW;           // strips per tile row / per image
  const long strips = (long)a.N * kpi;
  // block -> (co block, ci block, K share)
  const int b = blockIdx.x;
  const int ks = b % a.splitK, bb = b / a.splitK, ibk = bb % nIB, mbk = bb / nIB;
  const long e0 = strips * ks / a.splitK, e1 = strips * (ks + 1) / a.splitK;
  const int nsteps = (int)(e1 - e0);
  const int co0 = mbk * Cfg::MB, ci0 = ibk * Cfg::CB;
  (void)nMB;

  // MFMA role: wave = (co block cb, ci block ib)
  const int cb = swave & 3, ib = swave >> 2;
  const int a_lane = cb * 256 + lane * 4, b_lane = ib * 256 + lane * 4;      // + quad * 1024 / + quad * 512
  f32x4 acc[36];
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);

  // strip cursors: (image, tile row, strip column) of a flat strip index, advanced without divisions
  struct Cur { int n, tr, kc; };
  auto cur_at = [&](long e) {
    Cur c;
    c.n = (int)(e / kpi);
    const int rem = (int)(e - (long)c.n * kpi);
    c.tr = rem / kW;
    c.kc = rem - c.tr * kW;
    return c;
  };
  auto cur_next = [&](Cur& c) {
    if (++c.kc >= kW) { c.kc = 0; if (++c.tr >= (H >> 2)) { c.tr = 0; ++c.n; } }
  };

  // ---- X raw DMA: slot e = tid + i * 512 -> (channel c, row, piece q) of [32][RPLANE]
  auto dma_x = [&](const Cur& c, int stage, bool live) {
    const tnv3_rsrc_t rx = tnv3_make_rsrc(a.x + (size_t)(live ? c.n : 0) * Cin * HW, (unsigned)Cin * (unsigned)HW * 4u);
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);
#pragma unroll
    for (int i = 0; i < Cfg::NDMA; ++i) {
      if (i < Cfg::NDMA - 1 || swave < Cfg::DMA_LAST_WAVES) {
        const int e = t_op + i * NT;
        const int ch = e / Cfg::RPLANE, rem = e - ch * Cfg::RPLANE;
        const int r = rem / RQ, q = rem - r * RQ;
        const int gh = 4 * c.tr - 1 + r, gw = 16 * c.kc - 1 + 4 * q;
        const bool ok = live && e < Cfg::RAW_SLOTS && rem < 6 * RQ && gh >= 0 && gh < H;
        const unsigned vo = ok ? (unsigned)((ci0 + ch) * HW + gh * W + gw) * 4u : kDmaOob;      // (channels >= Cin: beyond the descriptor's range = zeros)
        tnv3_buf_dma16(rx, raw_s + stage * RAW_STAGE + (i * NT + wbase) * 4, vo);
      }
    }
  };

  // T: the transforms' strip (one ahead of the MFMAs'); Y: the dY loads' (two ahead); D: the X DMA's (RAWN ahead)
  Cur cT = cur_at(e0), cY = cT, cD = cT;
  int sT = 0, sY = 0, sD = 0;                           // steps the cursors are ahead of the slice start

  // Everything below is instantiated per wave group: waves 0-3 transform dY (Yh), waves 4-7 the patches of X (V).
  auto body = [&](auto grpc) {
  constexpr int GRP = decltype(grpc)::value;
  // ---- group 0: thread = (co = co block `swave`, lane & 15; tile = lane >> 4)
  f32x4 dy[4];
  auto load_dy = [&](const Cur& c, bool live) {
    const tnv3_rsrc_t rz = tnv3_make_rsrc(a.dz + (size_t)(live ? c.n : 0) * Cout * HW, (unsigned)Cout * (unsigned)HW * 4u);
    const unsigned vo = live ? (unsigned)((co0 + 16 * (swave & 3) + (lane & 15)) * HW + (4 * c.tr) * W + 16 * c.kc + 4 * (lane >> 4)) * 4u : kDmaOob;
#pragma unroll
    for (int r = 0; r < 4; ++r) dy[r] = tnv3_buf_load_f4(rz, vo, (unsigned)(r * W) * 4u);
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
  // ---- group 1: thread = (ci block (swave - 4) & 1, ci = lane & 15; tile = lane >> 4; row half RH = (swave - 4) >> 1)
  const int v_ib = swave & 1, v_rh = (swave >> 1) & 1;
  const int v_t = lane >> 4, v_ci = 16 * v_ib + (lane & 15);
  const int v_src = v_ci * (Cfg::RPLANE * 4) + (v_rh * RQ + v_t) * 4;      // + row * 20 floats; second piece + 4
  const int v_dst = v_ib * 256 + lane * 4 + 2 * v_rh;                      // + quad * 512
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
        if ((lane & 15) == 0 && v_t == 0 && v_ib == 0) {
          const tnv3_rsrc_t ri = tnv3_make_rsrc(a.x + (size_t)c.n * Cin * HW, (unsigned)Cin * (unsigned)HW * 4u);
          const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
          const f32x4 fx = f32x4{0.0f, x[0], x[1], x[2]};
          if (v_rh) tq0[0] = fx; else tq0[1] = fx;
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
      if (v_rh) {
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
  auto set_v_flags = [&](const Cur& c, bool live, bool first_of_image_channel0) {
    zl = c.kc == 0 && v_t == 0;
    zr = c.kc == kW - 1 && v_t == 3;
    fix_corner = live && first_of_image_channel0;
  };
  auto transform_all = [&](int stage, int raw_stage, const Cur& c, bool live) {      // (prologue: not interleaved)
    if constexpr (GRP == 0) {
      float* dst = yh_s + stage * YH + (swave & 3) * 256 + lane * 4;
      wino43s_for<0, NPIECE>([&](auto pc) { yh_piece(pc, dst); });
    } else {
      set_v_flags(c, live, c.tr == 0 && c.kc == 0 && ci0 == 0);
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
  dma_x(cD, 0, sD < nsteps);
  cur_next(cD); ++sD;
  if constexpr (GRP == 0) load_dy(cY, sY < nsteps);
  cur_next(cY); ++sY;
  full_barrier();
  transform_all(0, 0, cT, sT < nsteps);
  cur_next(cT); ++sT;
#pragma unroll
  for (int rs = 1; rs < RAWN; ++rs) {
    dma_x(cD, rs, sD < nsteps);
    cur_next(cD); ++sD;
  }
  if constexpr (GRP == 0) load_dy(cY, sY < nsteps);      // (the transform above has consumed step 0's tile)
  cur_next(cY); ++sY;
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
    const bool liveT = sT < nsteps, liveY = sY < nsteps, liveD = sD < nsteps;
    const float* raw = raw_s + rnext * RAW_STAGE + v_src;   // raw(sigma + 1)
    float* ydst = yh_s + sn * YH + (swave & 3) * 256 + lane * 4;
    float* vdst = v_s + sn * VS + v_dst;
    if constexpr (GRP == 1) set_v_flags(cT, liveT, cT.tr == 0 && cT.kc == 0 && ci0 == 0);
    f32x4 aq[2], bq[2];
    if constexpr (kOpReads) {
      aq[0] = *reinterpret_cast<const f32x4*>(A);
      bq[0] = *reinterpret_cast<const f32x4*>(B);
    } else {
      aq[0] = aq[1] = bq[0] = bq[1] = f32x4{1.0f, 0.5f, 0.25f, 2.0f};
    }
    // raw(sigma + RAWN) -> the raw stage whose strip the transform of the previous step has consumed
    if constexpr (kDma) dma_x(cD, rcur, liveD);
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
          acc[4 * q + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q & 1][e], bq[q & 1][e], sg == 0 ? zero4 : acc[4 * q + e], 0, 0, 0);
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
        if constexpr (kDy && ((kEarly2 && q == 1) || (kEarly && !kEarly2 && q == 3))) load_dy(cY, liveY);
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
    // End of the step: this wave's pieces of raw(sigma + 2) -- what the next step transforms -- must have landed.  vmcnt retires in order:
    // with two raw stages they are this step's DMA (older than the dY loads of this step only); with three, the previous step's (older
    // than this step's DMA as well).
    constexpr int kDyN = kDy ? 4 : 0;
    if constexpr (GRP == 0) {
      if constexpr (kDy && !kEarly && !kEarly2) load_dy(cY, liveY);
      if constexpr (kRaw3 && kDma) {
        if (swave < Cfg::DMA_LAST_WAVES) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(kDyN + Cfg::NDMA));
        else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(kDyN + Cfg::NDMA - 1));
      } else {
        __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(kDyN));
      }
    } else {
      if constexpr (kRaw3 && kDma) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(Cfg::NDMA - 1));      // (waves 4-7 issue NDMA - 1 pieces)
      else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
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
    cur_next(cT); ++sT;
    cur_next(cY); ++sY;
    cur_next(cD); ++sD;
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
        for (int i = 0; i < 6; ++i) m[i] = nsteps > 0 ? acc[4 * (3 * (i % 3) + j / 2) + 2 * (i / 3) + (j & 1)][r] : 0.0f;
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
  if (swave >> 2) body(std::integral_constant<int, 1>{}); else body(std::integral_constant<int, 0>{});
}

template <int SW>
__global__ void __launch_bounds__(WgradWino43Cfg::NT) wgrad_wino43_r5_twin_kernel(const WgradWinoArgs a) { wgrad_wino43_r5_body<SW>(a); }

}  // namespace tnv3
#endif
