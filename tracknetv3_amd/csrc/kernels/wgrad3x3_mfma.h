// wgrad3x3_mfma.h -- weight gradient of the 3x3 convolution on the fp32 matrix cores.
//
//   dW[co][ci][kh][kw] = sum_{n,h,w} dZ[n][co][h][w] * X[n][ci][h+kh-1][w+kw-1]          (X zero-padded)
//
// What autograd derives for nn.Conv2d(k=3, padding='same', bias=False) in Conv2DBlock (model.py:8) when
// train.py:95 calls loss.backward().  GEMM view: M = co, N = ci (one 32-wide N-tile per filter tap), K = pixels.
//   MFMA 32x32x2: lane l supplies A[co = l&31][pixel p + (l>>5)] and B[pixel p + (l>>5)][ci = l&31].
//   LDS: dZ tile [co][TR*TC] (row stride odd -> the 32 lanes of a half-wave hit 32 banks), X halo tile
//   [ci][TR+2][TC+2] (plane stride odd, same reason).  Each wave owns one 32-channel M-tile x one 32-channel ci
//   group and keeps all 9 taps in registers (9 x 16 accumulators): 1 A read + 9 B reads per 9 MFMAs.
// K (all pixels of the batch) is split over `splitK` workgroups per (co, ci) block; each writes its partial slab
// part[ks][Cout][Cin][9]; sum_partials_kernel adds the slabs in a fixed order (deterministic, no atomics).
// X may be the two-source, nearest-upsampled concat input of a decoder-entry layer (same loader as the forward).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "conv3x3_mfma.h"

namespace tnv3 {

struct WgradArgs {
  const float* src0;   // X part 1: [N][C0][H][W] or [N][C0][H/2][W/2] when up0
  const float* src1;   // X part 2: [N][C1][H][W] or nullptr
  const float* dz;     // [N][Cout][H][W]
  float* part;         // [splitK][Cout][C0+C1][9]
  int N, C0, C1, Cout, H, W, up0, splitK;
  const float* zeros;  // >= 64 zero floats (source of the padding for the LDS-DMA kernel)
  int kh0, kw0;        // WgradCfg<..., NTAP = 4> only: the 2x2 subset of taps {kh0, kh0+1} x {kw0, kw0+1} (slab [..][4])
};

template <int WM_, int WC_, int TR_ = 4, int TC_ = 32, int NTAP_ = 9>
struct WgradCfg {
  static constexpr int WM = WM_, WC = WC_, TR = TR_, TC = TC_;
  static constexpr int NTAP = NTAP_, TSIDE = NTAP_ == 9 ? 3 : 2;   // 9: the 3x3 filter; 4: a 2x2 window of it at (kh0, kw0) --
                                                                  // the per-parity pieces of a decoder-entry layer's low-res wgrad
  static_assert(NTAP_ == 9 || NTAP_ == 4, "tap sets");
  static constexpr int NT = WM * WC * 64;
  static constexpr int MB = WM * 32, CB = WC * 32;
  static constexpr int PIX = TR * TC, PIXP = PIX + 1;
  static constexpr int TRp = TR + 2, TCp = TC + 2, PLANE = TRp * TCp, PLANEP = PLANE | 1;
  static constexpr int DZ_FLOATS = MB * PIXP, X_FLOATS = CB * PLANEP;
  static constexpr int LDS_BYTES = (DZ_FLOATS + X_FLOATS) * 4;
  static_assert(TC % 2 == 0, "one MFMA consumes two horizontally adjacent pixels");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) wgrad3x3_mfma_kernel(const WgradArgs a) {
  constexpr int WC = Cfg::WC, TR = Cfg::TR, TC = Cfg::TC, NT = Cfg::NT, MB = Cfg::MB, CB = Cfg::CB;
  constexpr int PIX = Cfg::PIX, PIXP = Cfg::PIXP, TRp = Cfg::TRp, TCp = Cfg::TCp, PLANE = Cfg::PLANE, PLANEP = Cfg::PLANEP;
  constexpr int NDZ4 = (MB * PIX / 4 + NT - 1) / NT;     // 16-byte dZ loads per thread per tile
  constexpr int NX = CB;                                 // X halo elements per thread per tile (one per channel)
  static_assert((MB * PIX / 4) % NT == 0, "dZ tile must divide evenly over the workgroup");
  __shared__ float lds[Cfg::DZ_FLOATS + Cfg::X_FLOATS];
  float* dz_s = lds;
  float* x_s = lds + Cfg::DZ_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave % WC, wm = wave / WC;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cout = a.Cout, C0 = a.C0, C1 = a.C1, Cin = C0 + C1;
  const int HW = H * W;
  const int H0 = a.up0 ? (H >> 1) : H, W0 = a.up0 ? (W >> 1) : W, HW0 = H0 * W0;

  const int nCB = (Cin + CB - 1) / CB;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * MB, ci0 = cb * CB;

  const int tilesH = (H + TR - 1) / TR, tilesW = (W + TC - 1) / TC;
  const int nTiles = a.N * tilesH * tilesW;

  // ---- per-thread staging roles (tile independent)
  // dZ: 16-byte group g = tid + i*NT of [MB][TR][TC/4].   X: thread tid < PLANE owns ONE (row, col) of the halo plane and
  // walks the CB channels (i = channel): no per-slot coordinates to keep, two instructions per element.
  static_assert(PLANE <= NT, "one thread per halo-plane element");
  const int x_tr = tid / TCp, x_tc = tid - x_tr * TCp;
  const bool x_thread = tid < PLANE;

  f32x4 rdz[NDZ4];
  float rx[NX];
  unsigned dz_ok = 0;                           // validity bits of the dZ groups currently held in registers
  bool x_ok = false;                            // this thread's halo position is inside the image
  static_assert(NDZ4 <= 32, "validity mask is 32-bit");

  // The whole ci block comes from ONE source (host guarantees C0 % CB == 0 for a two-source input), so the source,
  // its plane geometry and the upsample flag are workgroup-uniform: one branch per tile, none per element.
  const bool blk0 = ci0 < C0;
  const bool up = blk0 && a.up0;
  const float* xsrc = blk0 ? a.src0 : a.src1;
  const int Cs = blk0 ? C0 : C1, cib = blk0 ? ci0 : ci0 - C0;
  const int Ws = up ? W0 : W, HWs = up ? HW0 : HW;

  auto load_tile = [&](int tile) {
    const int n = tile / (tilesH * tilesW);
    const int trem = tile - n * (tilesH * tilesW);
    const int h0 = (trem / tilesW) * TR, w0 = (trem % tilesW) * TC;
    const float* dzn = a.dz + (size_t)n * Cout * HW;
    dz_ok = 0;
#pragma unroll
    for (int i = 0; i < NDZ4; ++i) {
      const int g = tid + i * NT;
      const int c4 = g % (TC / 4), t2 = g / (TC / 4);
      const int r = t2 % TR, co_l = t2 / TR;
      const int co = co0 + co_l, gh = h0 + r, gw = w0 + 4 * c4;
      const bool ok = co < Cout && gh < H && gw < W;            // W % 4 == 0: a 16-byte group is all-in or all-out
      const int off = ok ? (co * HW + gh * W + gw) : 0;
      rdz[i] = *reinterpret_cast<const f32x4*>(dzn + off);
      dz_ok |= ok ? (1u << i) : 0u;
    }
    const int gh = h0 - 1 + x_tr, gw = w0 - 1 + x_tc;
    x_ok = x_thread && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
    const int sp_off = x_ok ? (up ? (gh >> 1) * Ws + (gw >> 1) : gh * Ws + gw) : 0;
    const float* xn = xsrc + ((size_t)n * Cs + cib) * HWs + sp_off;
    const int nch = Cs - cib;                                    // channels of this source from cib on (>= 1)
#pragma unroll
    for (int i = 0; i < NX; ++i) rx[i] = xn[(size_t)(i < nch ? i : 0) * HWs];
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NDZ4; ++i) {
      const int g = tid + i * NT;
      const int c4 = g % (TC / 4), t2 = g / (TC / 4);
      const int r = t2 % TR, co_l = t2 / TR;
      float* d = dz_s + co_l * PIXP + r * TC + 4 * c4;          // odd row stride: four scalar stores
      const bool ok = (dz_ok >> i) & 1u;
      d[0] = ok ? rdz[i][0] : 0.0f; d[1] = ok ? rdz[i][1] : 0.0f; d[2] = ok ? rdz[i][2] : 0.0f; d[3] = ok ? rdz[i][3] : 0.0f;
    }
    if (x_thread) {
      const int nch = Cs - cib;
#pragma unroll
      for (int i = 0; i < NX; ++i) x_s[i * PLANEP + tid] = (x_ok && i < nch) ? rx[i] : 0.0f;
    }
  };

  constexpr int NTAP = Cfg::NTAP, TSIDE = Cfg::TSIDE;
  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int a_off = (wm * 32 + bl) * PIXP + half;
  const int b_off = (wc * 32 + bl) * PLANEP + half + (NTAP == 9 ? 0 : a.kh0 * TCp + a.kw0);

  if (ks < nTiles) load_tile(ks);
  for (int tile = ks; tile < nTiles; tile += a.splitK) {
    __syncthreads();                            // every wave is done reading the previous tile
    store_tile();
    __syncthreads();
    if (tile + a.splitK < nTiles) load_tile(tile + a.splitK);   // in flight during the MFMA block below
    // ---- MFMA: K-step = two horizontally adjacent pixels of one tile row
    const float* A = dz_s + a_off;
    const float* B = x_s + b_off;
    // Per tile row: 16 K-steps (two adjacent pixels each), fully unrolled; operands of step s+1 are read before the
    // nine MFMAs of step s (static double buffer), as in the forward kernel.  The row loop stays rolled (code size /
    // compile time) and the operand pipeline is carried across it: the last step of row r prefetches step 0 of row r+1.
    constexpr int NSTEP = TC / 2;
    static_assert(NSTEP % 2 == 0, "ring slot of the carried prefetch");
    float av[2], bv[2][NTAP];
    auto read_step = [&](int r, int s, float& ar, float (&br)[NTAP]) {
      ar = A[r * TC + 2 * s];
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) br[tap] = B[(r + tap / TSIDE) * TCp + 2 * s + (tap % TSIDE)];
    };
    read_step(0, 0, av[0], bv[0]);
#pragma unroll 1
    for (int r = 0; r < TR; ++r) {
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s + 1 < NSTEP) read_step(r, s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
        else if (r + 1 < TR) read_step(r + 1, 0, av[0], bv[0]);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap)
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][tap], acc[tap], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NTAP + 1, 0);   // DS reads of the next step
        __builtin_amdgcn_sched_group_barrier(0x008, NTAP, 0);       // MFMAs of step s
      }
    }
  }

  // ---- partial slab: part[ks][co][ci][tap]
  float* slab = a.part + (size_t)ks * Cout * Cin * NTAP;
  const int ci = ci0 + wc * 32 + bl;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (co < Cout && ci < Cin) {
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) slab[((size_t)co * Cin + ci) * NTAP + tap] = acc[tap][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant.  With one wave per SIMD (144 accumulator registers) nothing hides the instructions of a wave but its
// own MFMAs: the register-staged kernel above issues the ~800 address / load / ds_write instructions of a tile in two
// bursts during which the matrix pipe idles.  Here both tiles go global -> LDS by DMA (no staging registers, no
// ds_write), the LDS is double-buffered, and the DMAs of tile t+1 are issued two per K-step inside the fully unrolled
// MFMA stream of tile t, i.e. in the shadow of the 64-cycle MFMAs.  Tiles are TR = 2 rows (two 50 KB stages fit).
//   work item = one wave-wide DMA of 64 floats:  dZ row (co, 64 pixels)   -> dz_s[co][0..63]
//                                                X plane third (ci, g)    -> x_s[ci][64g .. 64g+63]   (halo plane 4x34 = 136)
//   items are dealt to the waves round-robin with a static (item -> kind, group) map, so no branch splits the MFMA block.
template <int WM_, int WC_>
struct WgradDmaCfg {
  static constexpr int WM = WM_, WC = WC_, TR = 2, TC = 32;
  static constexpr int NW = WM * WC, NT = NW * 64;
  static constexpr int MB = WM * 32, CB = WC * 32;
  static constexpr int PIX = TR * TC, PIXP = PIX + 1;
  static constexpr int TRp = TR + 2, TCp = TC + 2, PLANE = TRp * TCp, XG = (PLANE + 63) / 64, PLANEP = XG * 64 + 1;
  static constexpr int DZ_FLOATS = MB * PIXP, X_FLOATS = CB * PLANEP;
  static constexpr int BUF_FLOATS = DZ_FLOATS + X_FLOATS;
  static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
  static constexpr int DZ_IPW = MB / NW, X_IPW = XG * CB / NW, IPW = DZ_IPW + X_IPW;    // DMA items per wave per tile
  static constexpr int KSTEPS = PIX / 2, SLOTS = (IPW + KSTEPS - 1) / KSTEPS;
  static_assert(PIX == 64, "one dZ row of the tile = one wave-wide DMA");
  static_assert(MB % NW == 0 && CB % NW == 0, "items must deal evenly");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) wgrad3x3_dma_kernel(const WgradArgs a) {
  constexpr int WC = Cfg::WC, TR = Cfg::TR, TC = Cfg::TC, MB = Cfg::MB, CB = Cfg::CB, NW = Cfg::NW;
  constexpr int PIXP = Cfg::PIXP, TCp = Cfg::TCp, PLANE = Cfg::PLANE, PLANEP = Cfg::PLANEP, XG = Cfg::XG;
  __shared__ float lds[2 * Cfg::BUF_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave % WC, wm = wave / WC;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cout = a.Cout, C0 = a.C0, C1 = a.C1, Cin = C0 + C1;
  const int HW = H * W;
  const int H0 = a.up0 ? (H >> 1) : H, W0 = a.up0 ? (W >> 1) : W, HW0 = H0 * W0;

  const int nCB = (Cin + CB - 1) / CB;
  int b = blockIdx.x;
  const int ks = b % a.splitK; b /= a.splitK;
  const int cb = b % nCB, mb = b / nCB;
  const int co0 = mb * MB, ci0 = cb * CB;
  const int tilesH = (H + TR - 1) / TR, tilesW = (W + TC - 1) / TC;
  const int nTiles = a.N * tilesH * tilesW;

  const bool blk0 = ci0 < C0;                     // the ci block comes from one source (host: C0 % CB == 0)
  const bool up = blk0 && a.up0;
  const float* xsrc = blk0 ? a.src0 : a.src1;
  const int Cs = blk0 ? C0 : C1, cib = blk0 ? ci0 : ci0 - C0;
  const int Ws = up ? W0 : W, HWs = up ? HW0 : HW;
  const int nch = Cs - cib;                       // channels this source still has from cib on
  const float* zsrc = a.zeros + lane;

  // tile-independent lane roles: pixel (lane>>5, lane&31) of a dZ row; element g*64+lane of the halo plane
  const int dz_rel = (lane >> 5) * W + (lane & 31);
  int x_tr[XG], x_tc[XG];
#pragma unroll
  for (int g = 0; g < XG; ++g) { const int e = g * 64 + lane; x_tr[g] = e / TCp; x_tc[g] = e - x_tr[g] * TCp; }

  // per-tile staging state (of the tile being fetched)
  const float* dzt = nullptr;
  const float* xt = nullptr;
  bool dz_ok = false;
  int x_off[XG];
  auto prepare = [&](int tile) {
    const int n = tile / (tilesH * tilesW);
    const int trem = tile - n * (tilesH * tilesW);
    const int h0 = (trem / tilesW) * TR, w0 = (trem % tilesW) * TC;
    dzt = a.dz + ((size_t)n * Cout + co0) * HW + (size_t)h0 * W + w0 + dz_rel;
    dz_ok = (h0 + (lane >> 5) < H) && (w0 + (lane & 31) < W);
    xt = xsrc + ((size_t)n * Cs + cib) * HWs;
#pragma unroll
    for (int g = 0; g < XG; ++g) {
      const int gh = h0 - 1 + x_tr[g], gw = w0 - 1 + x_tc[g];
      const bool ok = g * 64 + lane < PLANE && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      x_off[g] = ok ? (up ? (gh >> 1) * Ws + (gw >> 1) : gh * Ws + gw) : -1;
    }
  };
  // item j of this wave (static j -> static kind / plane third)
  auto issue = [&](int j, float* stage) {
    if (j < Cfg::DZ_IPW) {
      const int co_l = j * NW + wave;
      const bool ok = dz_ok && co0 + co_l < Cout;
      lds_dma4(ok ? dzt + (size_t)co_l * HW : zsrc, stage + co_l * PIXP);
    } else {
      const int k = j - Cfg::DZ_IPW;
      const int g = k % XG, ci_l = (k / XG) * NW + wave;
      const bool ok = x_off[g] >= 0 && ci_l < nch;
      lds_dma4(ok ? xt + (size_t)ci_l * HWs + x_off[g] : zsrc, stage + Cfg::DZ_FLOATS + ci_l * PLANEP + g * 64);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int a_off = (wm * 32 + bl) * PIXP + half;
  const int b_off = Cfg::DZ_FLOATS + (wc * 32 + bl) * PLANEP + half;

  // One tile: 32 K-steps (two adjacent pixels each) x 9 taps, fully unrolled; the operands of step s+1 are read before
  // the MFMAs of step s; with FETCH the DMAs of the next tile ride along, SLOTS per K-step.
  auto tile_mfma = [&](const float* cur, float* nxt, auto fetch) {
    constexpr bool FETCH = decltype(fetch)::value;
    const float* A = cur + a_off;
    const float* B = cur + b_off;
    constexpr int NSTEP = Cfg::KSTEPS, SPR = TC / 2;
    float av[2], bv[2][9];
    auto read_step = [&](int st, float& ar, float (&br)[9]) {
      const int r = st / SPR, s = st - r * SPR;
      ar = A[r * TC + 2 * s];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) br[tap] = B[(r + tap / 3) * TCp + 2 * s + (tap % 3)];
    };
    read_step(0, av[0], bv[0]);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      if (st + 1 < NSTEP) read_step(st + 1, av[(st + 1) & 1], bv[(st + 1) & 1]);
      if (FETCH) {
#pragma unroll
        for (int q = 0; q < Cfg::SLOTS; ++q)
          if (st * Cfg::SLOTS + q < Cfg::IPW) issue(st * Cfg::SLOTS + q, nxt);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1], bv[st & 1][tap], acc[tap], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);             // DS reads of the next step
      if (FETCH) __builtin_amdgcn_sched_group_barrier(0x020, Cfg::SLOTS, 0);   // the DMAs riding on this step
      __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);              // MFMAs of this step
    }
  };

  if (ks < nTiles) {
    prepare(ks);
#pragma unroll
    for (int j = 0; j < Cfg::IPW; ++j) issue(j, lds);
  }
  int buf = 0;
  for (int tile = ks; tile < nTiles; tile += a.splitK) {
    __syncthreads();          // this stage has landed (vmcnt drained) for every wave; the other stage is no longer read
    float* cur = lds + buf * Cfg::BUF_FLOATS;
    float* nxt = lds + (buf ^ 1) * Cfg::BUF_FLOATS;
    if (tile + a.splitK < nTiles) {
      prepare(tile + a.splitK);
      tile_mfma(cur, nxt, std::true_type{});
    } else {
      tile_mfma(cur, nxt, std::false_type{});
    }
    buf ^= 1;
  }

  float* slab = a.part + (size_t)ks * Cout * Cin * 9;
  const int ci = ci0 + wc * 32 + bl;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (co < Cout && ci < Cin) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) slab[((size_t)co * Cin + ci) * 9 + tap] = acc[tap][r];
    }
  }
}

}  // namespace tnv3
