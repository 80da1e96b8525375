// conv3x3_wino43s_mfma.h -- Winograd F(4x4, 3x3) with ALL 36 transform coefficients of an output block in ONE wave
// (tnv3_conv3x3_wino43_forward, kernel variant 0; conv3x3_wino43_mfma.h is its predecessor, variant 1, kept as the A/B twin).
//
// Why.  The 32x32x2 kernel gives a wave nine of the 36 coefficients xi (9 x 16 accumulator registers), so the output transform
// A^T M A needs the partial sums of four waves: 96 KB through LDS per round, eight barriers, 1750 instructions per wave and tile --
// 15-19 k cycles of write-out per tile against 5.8 k per chunk of useful work (profiles/r03_wino43_timeline.json).  With
// v_mfma_f32_16x16x4_f32 one accumulator is FOUR registers: a wave holds all 36 xi of a (16 channels x 16 tiles) block in 144, the
// output transform is per-lane register arithmetic, and nothing is exchanged.
//
// Mapping.  One wave per SIMD (256 threads, 512 registers per lane): workgroup tile = 64 output channels x 32 tiles of 4x4 (2 tile
// rows x 16 tile columns = 8 x 64 pixels); wave w = 16 channels x BOTH tile rows = two accumulator sets of 36 x 4 registers.  288
// accumulators do not fit the 256 VGPRs an instruction can name, so the MFMAs are written as inline asm with the register file
// chosen per accumulator: set 0 and most of set 1 in AGPRs ("+a"), the rest of set 1 in VGPRs ("+v") -- the compiler allocates, it
// never copies (tests/test_kernel_resources.py: no v_accvgpr_* inside the chunk loop, no spills).
//   MFMA 16x16x4: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D reg r = row 4 (l >> 4) + r, col l & 15.
//   M = 16 output channels, N = 16 tile columns, K = 4 input channels; a chunk = 8 input channels = two K steps s; lane group
//   g = l >> 4 takes input channels 2 g + s.
//   * A operand (U = G g G^T) from L2 into registers: panel [16-channel block][chunk][xi pair 18][lane][4 = (s, xi & 1)], one
//     16-byte load per xi pair = 8 MFMAs (both tile rows share it); ring of six quads, five pairs (~1300 cycles) ahead.
//   * B operand (V = B^T d B) from LDS: V[stage 3][xi pair 18][tile row 2][g 4][tile column 16][4 = (s, xi & 1)], one
//     ds_read_b128 per (pair, tile row): a wave reads 1 KB contiguous, conflict-free; ring of three pairs.
//   * The raw halo tile arrives by LDS-DMA as 17 pieces of four columns per row STARTING ONE COLUMN LEFT of the tile (w0 - 1 ..
//     w0 + 66; the global address of a piece is 4-byte, not 16-byte aligned): the 6 columns of tile column tc's patch are piece tc
//     and the first half of piece tc + 1 -- one ds_read_b128 + one ds_read_b64 per raw row instead of one 16-byte and two
//     conflicting 4-byte reads (29 % of the predecessor's LDS cycles were bank conflicts of those).  What the shift costs: the
//     column left of the image and the one right of it are real memory (the neighbouring row's ends), so the lanes of tile column 0
//     / 15 of a border tile zero their first / last transformed column (three v_cndmask each per half patch), and the ONE piece per
//     image that would start before the image (channel 0, row 0) is patched by one lane from a guarded load.
//   * Thread = one patch (channel, tile): six raw rows -> B^T d B -> 18 ds_write_b64 (a lane's two values of a xi pair are
//     adjacent).  Lane order (tc & 7, s, tc >> 3, g & 1) makes raw reads and V stores conflict-free in the hardware's lane groups.
//   * Pipeline: step = one chunk of one tile; the stream of steps runs ACROSS tile boundaries.  Step sigma: MFMAs on V(sigma),
//     transform raw(sigma + 2) -> V(sigma + 2), DMA raw(sigma + 3); three V stages and two raw stages; ONE barrier per step, and
//     the B operands of the next step's first pairs are read before it (V(sigma + 1) was complete one barrier earlier).
//   * A step is 72 slots of two MFMAs; behind each slot sits a piece of at most ~8 other instructions (sched_barrier fences):
//     the matrix pipe runs one 32-cycle MFMA after the other while the wave issues the pieces in its shadow.
//   * Write-out: per lane 4 channels x (4x4 pixels) per tile row: A^T M A in registers (110 operations per channel), addend /
//     BatchNorm / ReLU, 16-byte stores (16 lanes = 256 contiguous bytes of an output row).  No LDS, no barrier.
// Needs Cout % 64 == 0, H % 4 == 0, W % 64 == 0 (H % 8 == 4: the last tile row's lower half is outside the image).
// Deterministic; not bit-identical to the 32x32x2 kernel (the output transform adds in another order).
#pragma once
#include <type_traits>
#include <utility>
#include "conv3x3_wino43_mfma.h"

namespace tnv3 {

struct Wino43SCfg {
  static constexpr int CC = 8, NT = 256, MB = 64, TB = 32;
  static constexpr int TH = 8, TW = 64;
  static constexpr int RROWS = 10, RQ = 17;               // raw halo tile per channel: rows h0-1 .. h0+8, 17 pieces from column w0-1
  static constexpr int RPLANE = 176;                      // pieces per channel plane (170 used): a multiple of 16 pieces, so that the two
                                                          // channels a 16-lane read group touches fall into disjoint bank ranges
  static constexpr int RAW_SLOTS = CC * RPLANE;           // 1408 pieces per stage: 5.5 per thread
  static constexpr int RAW_STAGE = RAW_SLOTS * 4;         // floats
  static constexpr int NDMA = 6;                          // DMA instructions per wave and step (the sixth: waves 0 and 1)
  static constexpr int PAIRS = 18;
  static constexpr int V_PAIR = 2 * 4 * 16 * 4;           // floats per xi pair: [tile row][g][tile column][4]
  static constexpr int V_STAGE = PAIRS * V_PAIR;          // 9216 floats = 36 KB
  static constexpr int NV = 3, NR = 2;
  static constexpr int LDS_FLOATS = NV * V_STAGE + NR * RAW_STAGE;      // 155,648 bytes
  static constexpr int A_CHUNK_FLOATS = PAIRS * 64 * 4;   // one (16-channel block, chunk) of the panel
  static constexpr int B_RING = 3, B_DIST = 2;            // (the A ring's size is a kernel template parameter: 6, 9 or 18 quads)
  static constexpr int ACC1_V_FROM = 14;                  // xi pairs >= this keep tile row 1's accumulators in VGPRs (8 xi x 4 = 32)
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
  static_assert(PAIRS % B_RING == 0, "rings are indexed statically");
};

inline size_t conv3x3_wino43s_packed_floats(int cin, int cout) {
  if (cin <= 0 || cout <= 0 || cout % 16) return 0;
  return (size_t)(cout / 16) * ((cin + Wino43SCfg::CC - 1) / Wino43SCfg::CC) * Wino43SCfg::A_CHUNK_FLOATS + kPackZeroTail;
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

// Two MFMAs 16x16x4 (fp32) on two accumulators of one register file: "a" = AGPR, "v" = VGPR.  ZERO: the accumulators start from
// the inline constant 0 (a tile's first K step: no zeroing pass, no v_accvgpr_write -> MFMA hazard).  The two are independent; a
// dependent pair (the same accumulator's next K step) is always at least one slot away.
// Hazards hipcc does not pad for an asm statement (cdna_hip_programming.md 5.7): the A / B operands here come from buffer / LDS
// loads (s_waitcnt, which the compiler does insert for asm inputs), never from a VALU write in the two instructions before
// (tests/test_kernel_resources.py scans for it); the accumulators are read by other instructions only in the write-out, more than
// 12 wait states after the last MFMA (an s_nop fence sits there).
template <bool ZERO, bool VCLS>
__device__ __forceinline__ void wino43s_mfma2(f32x4& c0, f32x4& c1, float a0, float a1, float b0, float b1) {
#ifdef TNV3_EMU
  const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, ZERO ? z : c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, ZERO ? z : c1, 0, 0, 0);
#else
  if constexpr (ZERO) {
    if constexpr (VCLS)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, 0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %5, 0"
                   : "=&v"(c0), "=&v"(c1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    else
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, 0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %5, 0"
                   : "=a"(c0), "=a"(c1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
  } else {
    if constexpr (VCLS)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %5, %1"
                   : "+v"(c0), "+v"(c1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    else
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %5, %1"
                   : "+a"(c0), "+a"(c1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
  }
#endif
}

// One component of an accumulator for ordinary arithmetic.  From an AGPR accumulator hipcc would copy the whole 128-bit tuple for
// every component it extracts (144 registers for one channel's 36 values); the explicit read names the one register.
template <bool VCLS>
__device__ __forceinline__ float wino43s_acc_read(const f32x4& c, int r) {
#ifdef TNV3_EMU
  return c[r];
#else
  if constexpr (VCLS) return c[r];
  else {
    float x;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(c[r]));
    return x;
  }
#endif
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

// AR: quads in the A ring (a quad is requested AR - 1 xi pairs = (AR - 1) x 256 matrix-pipe cycles before its MFMAs).
// TL = 1 (libtnv3_diag.so only): s_memtime totals of one mid-grid workgroup, [wave 4][8] uint64 to a.stats: 0 prologue, 1 steps,
// 2 write-outs, 3 steps walked, 4 tiles walked.  DG (diag only, WRONG results): timing twins -- bit 0 no raw DMA after the prologue,
// bit 1 no patch transform, bit 2 no A loads (the ring keeps the prologue's quads), bit 3 no B reads, bit 4 no MFMAs, bit 5 no output stores.
template <int STATS = 0, int AR = 6, int TL = 0, int DG = 0>
__global__ void __launch_bounds__(Wino43SCfg::NT) conv3x3_wino43s_kernel(const WinoArgs a) {
  using Cfg = Wino43SCfg;
  constexpr int A_RING = AR, A_DIST = AR - 1;
  static_assert(Cfg::PAIRS % A_RING == 0, "the A ring is indexed statically");
  constexpr int CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, V_STAGE = Cfg::V_STAGE, RAW_STAGE = Cfg::RAW_STAGE, V_PAIR = Cfg::V_PAIR;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* v_s = lds;
  float* raw_s = lds + Cfg::NV * V_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = (H + Cfg::TH - 1) / Cfg::TH, tilesW = W / Cfg::TW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (Cin + CC - 1) / CC;

  // Three cursors walk the same list of (tile, chunk) steps: M (the MFMAs), T (the patch transform: two steps ahead), D (the raw DMA: three
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
  // ---- transform role: thread = patch (channel ci = 2 g + s, tile row tr, tile column tc); lane bits (tc & 7, s, tc >> 3, g & 1), wave
  //      bits (g >> 1, tr): the 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...) then hold tile columns {0-3, 12-15} of one
  //      channel and {4-11} of its neighbour (planes a multiple of 16 pieces apart: 16 different 16-byte bank groups), and the 16 lanes
  //      of a ds_write_b64 group hold 8 tile columns x 2 s = 32 different banks.
  const int t_tc = (lane & 7) | ((lane >> 1) & 8), t_s = (lane >> 3) & 1, t_g = ((lane >> 5) & 1) | ((swave & 1) << 1), t_tr = swave >> 1;
  const int t_ci = 2 * t_g + t_s;
  const int t_src = t_ci * (Cfg::RPLANE * 4) + ((4 * t_tr) * Cfg::RQ + t_tc) * 4;      // + row * 68 floats; second piece + 4
  const int t_dst = t_tr * 256 + t_g * 64 + t_tc * 4 + t_s * 2;                          // + pair * V_PAIR
  const int b_lane = lane * 4;                                                           // + pair * V_PAIR + ts * 256
  const unsigned a_lane_b = (unsigned)lane * 16u;
  const tnv3_rsrc_t r_panel = tnv3_make_rsrc(a.u, (unsigned)((size_t)(Cout / 16) * nChunks * Cfg::A_CHUNK_FLOATS * 4));

  f32x4 acc0[36], acc1[36];                             // tile row 0 / 1: [xi = 6 i + j], register r = channel 4 (lane >> 4) + r
  f32x4 aq[A_RING], bq[Cfg::B_RING][2];

  // ---- D cursor: per-tile piece offsets.  Slot e = tid + i * 256 -> (channel c, row, piece q) of [CC][RPLANE]; pieces 170 .. 175 of a
  //      plane, rows outside the image and channels >= Cin read out of the descriptor's range = zeros.
  unsigned voD[Cfg::NDMA];
  tnv3_rsrc_t r_srcD;
  auto set_d = [&]() {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);
    const int h0 = wD.trow * Cfg::TH, w0 = wD.tcol * Cfg::TW;
#pragma unroll
    for (int i = 0; i < Cfg::NDMA; ++i) {
      const int e = t_op + i * NT;
      const int c = e / Cfg::RPLANE, rem = e - c * Cfg::RPLANE;
      const int r = rem / Cfg::RQ, q = rem - r * Cfg::RQ;
      const int gh = h0 - 1 + r, gw = w0 - 1 + 4 * q;
      const bool ok = wD.valid && e < Cfg::RAW_SLOTS && rem < Cfg::RROWS * Cfg::RQ && gh >= 0 && gh < H;
      voD[i] = ok ? (unsigned)(c * HW + gh * W + gw) * 4u : kDmaOob;      // (channel 0, row 0, column -1: wraps beyond the range = zeros; see fix_corner)
    }
    const int nn = wD.valid ? wD.n : 0;
    r_srcD = tnv3_make_rsrc(a.src + (size_t)nn * Cin * HW, (unsigned)Cin * (unsigned)HW * 4u);
  };
  auto dma_piece = [&](int i, int stage) {              // piece i of chunk kD of the D tile -> raw stage
    if (i < Cfg::NDMA - 1 || swave < 2)
      tnv3_buf_dma16(r_srcD, raw_s + stage * RAW_STAGE + (i * NT + wbase) * 4, voD[i] + (unsigned)kD * (unsigned)(CC * 4) * (unsigned)HW);
  };
  auto adv_d = [&]() {
    if (++kD >= nChunks) { kD = 0; wD.next(); set_d(); }
  };

  // ---- T cursor: the transform of one patch, in pieces (the step places one piece behind each MFMA slot)
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
  f32x4 tq0[6];
  wf2 tq1[6];
  float tt[6][6];
  auto t_piece = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    if constexpr (P == 0) {
#pragma unroll
      for (int r = 0; r < 6; ++r) tq0[r] = *reinterpret_cast<const f32x4*>(t_raw + r * (Cfg::RQ * 4));
      if (fix_corner) {                                  // scalar branch, taken once per image: the piece before the image's first element
        if (tid == 0) {
          const tnv3_rsrc_t ri = tnv3_make_rsrc(a.src + (size_t)wT.n * Cin * HW, (unsigned)Cin * (unsigned)HW * 4u);
          const f32x4 x = tnv3_buf_load_f4(ri, 0u, 0u);
          tq0[1] = f32x4{0.0f, x[0], x[1], x[2]};
        }
      }
    } else if constexpr (P == 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) tq1[r] = *reinterpret_cast<const wf2*>(t_raw + r * (Cfg::RQ * 4) + 4);
    } else if constexpr (P < 14) {                       // first pass, down patch column c: rows 0-2 (h = 0) or 3-5 (h = 1) of B^T d
      constexpr int c = (P - 2) >> 1, h = (P - 2) & 1;
      float d[6], o[3];
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r] = c < 4 ? tq0[r][c < 4 ? c : 0] : tq1[r][c < 4 ? 0 : c - 4];
      if constexpr (h == 0) wino43_bt_half<0>(d, o); else wino43_bt_half<1>(d, o);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float v = o[r];
        if constexpr (c == 0) v = zl ? 0.0f : v;
        if constexpr (c == 5) v = zr ? 0.0f : v;
        tt[3 * h + r][c] = v;
      }
    } else {                                             // second pass along row i, results straight into V: xi = (i, j), pair 3 i + j / 2
      constexpr int i = (P - 14) / 3, sub = (P - 14) % 3;
      const float(&d)[6] = tt[i];
      float* vd = t_v + (3 * i) * V_PAIR;
      if constexpr (sub == 0) {
        const float o0 = fmaf(kW43_4S4, d[0], fmaf(-kW43_5S2, d[2], d[4]));
        const float aa = fmaf(-kW43_4S2, d[2], d[4]), bb = fmaf(-kW43_4S2, d[1], d[3]);
        const float o1 = fmaf(kW43S, bb, aa);
        tt[i][0] = fmaf(-kW43S, bb, aa);                 // o2, parked in the consumed slot until the next piece
        *reinterpret_cast<wf2*>(vd) = wf2{o0, o1};
      } else if constexpr (sub == 1) {
        const float cc = fmaf(-kW43S2, d[2], d[4]), ee = fmaf(-kW43S2, d[1], d[3]);
        const float o3 = fmaf(kW43_2S, ee, cc);
        const float o4 = fmaf(-kW43_2S, ee, cc);
        *reinterpret_cast<wf2*>(vd + V_PAIR) = wf2{d[0], o3};
        tt[i][0] = o4;
      } else {
        const float o5 = fmaf(kW43_4S4, d[1], fmaf(-kW43_5S2, d[3], d[5]));
        *reinterpret_cast<wf2*>(vd + 2 * V_PAIR) = wf2{d[0], o5};
      }
    }
  };
  constexpr int T_PIECES = 32;
  auto adv_t = [&]() {
    if (++kT >= nChunks) { kT = 0; wT.next(); }
  };

  // ---- A / B operand streams
  auto a_soff = [&](int mb, int k) -> unsigned {      // byte offset of this wave's (16-channel block, chunk) in the panel
    return (unsigned)((mb * 4 + swave) * nChunks + k) * (unsigned)(Cfg::A_CHUNK_FLOATS * 4);
  };
  auto a_soff_next = [&]() -> unsigned {               // ... of the step after M's: the next chunk, or the next tile's first (the walk's channel-block rule)
    if (kM + 1 < nChunks) return a_soff(wM.mb, kM + 1);
    int mb = wM.mb + wM.d_mb;
    if (mb >= nMB) mb -= nMB;
    return a_soff(mb, 0);
  };
  auto full_barrier = [&]() {
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: raw(0), raw(1) -> V(0); raw(2); V(1); the operand rings of step 0
  int sv = 0, sr = 0;                                    // V stage of step sigma = sigma % 3, raw stage of raw(sigma) = sigma & 1
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
  full_barrier();
#pragma unroll
  for (int i = 0; i < Cfg::NDMA; ++i) dma_piece(i, 0);
  adv_d();
  set_t(1, 1);
  wino43s_for<0, T_PIECES>(t_piece);
  adv_t();
  {
    const unsigned s0 = a_soff(wM.mb, 0);
#pragma unroll
    for (int p = 0; p < A_DIST; ++p) aq[p] = tnv3_buf_load_f4(r_panel, a_lane_b, s0 + (unsigned)p * 1024u);
  }
  full_barrier();
#pragma unroll
  for (int p = 0; p < Cfg::B_DIST; ++p)
#pragma unroll
    for (int ts = 0; ts < 2; ++ts) bq[p][ts] = *reinterpret_cast<const f32x4*>(v_s + p * V_PAIR + ts * 256 + b_lane);
  tl_stamp(0);

  // ---- one step: chunk kM of the M tile.  72 slots of two MFMAs (xi pair p, K step s, tile row ts); slot 4 p first issues the B reads of
  //      pair p + 2 and the A load of pair p + 5 (beyond pair 17: the next step's), every other slot is followed by one DMA / transform piece.
  auto step = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    const int svn = sv == 2 ? 0 : sv + 1, svt = svn == 2 ? 0 : svn + 1;
    const float* bM = v_s + sv * V_STAGE + b_lane;
    const float* bN = v_s + svn * V_STAGE + b_lane;
    const unsigned soM = a_soff(wM.mb, kM), soA = a_soff_next();      // (past the last step: a slice nobody uses, inside the panel)
    set_t(sr, svt);
    const int srd = sr ^ 1;
    wino43s_for<0, 72>([&](auto ic) {
      constexpr int IDX = decltype(ic)::value, p = IDX >> 2, s = (IDX >> 1) & 1, ts = IDX & 1;
      if constexpr ((IDX & 3) == 0) {
        constexpr int pb = p + Cfg::B_DIST, pa = p + A_DIST;
        if constexpr ((DG & 8) == 0) {
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2)
            bq[pb % Cfg::B_RING][t2] = *reinterpret_cast<const f32x4*>((pb < 18 ? bM : bN) + (pb % 18) * V_PAIR + t2 * 256);
        }
        if constexpr ((DG & 4) == 0)
          aq[pa % A_RING] = tnv3_buf_load_f4(r_panel, a_lane_b, (pa < 18 ? soM : soA) + (unsigned)(pa % 18) * 1024u);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4& av = aq[p % A_RING];
      const f32x4& bv = bq[p % Cfg::B_RING][ts];
      if constexpr ((DG & 16) != 0) {
        if constexpr (FIRST && s == 0) { (ts == 0 ? acc0 : acc1)[2 * p] = f32x4{av[0], bv[0], 0.0f, 0.0f}; (ts == 0 ? acc0 : acc1)[2 * p + 1] = f32x4{av[1], bv[1], 0.0f, 0.0f}; }
      } else if constexpr (ts == 0)
        wino43s_mfma2<FIRST && s == 0, false>(acc0[2 * p], acc0[2 * p + 1], av[2 * s], av[2 * s + 1], bv[2 * s], bv[2 * s + 1]);
      else
        wino43s_mfma2<FIRST && s == 0, (p >= Cfg::ACC1_V_FROM)>(acc1[2 * p], acc1[2 * p + 1], av[2 * s], av[2 * s + 1], bv[2 * s], bv[2 * s + 1]);
      if constexpr ((IDX & 3) != 0) {
        constexpr int o = IDX - (IDX >> 2) - 1;          // ordinal among the slots that carry a piece: 0 .. 53
        if constexpr (o < Cfg::NDMA) { if constexpr ((DG & 1) == 0) dma_piece(o, srd); }
        else if constexpr (o - Cfg::NDMA < T_PIECES) { if constexpr ((DG & 2) == 0) t_piece(std::integral_constant<int, o - Cfg::NDMA>{}); }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // this wave's raw pieces of step sigma + 3 have landed (they are OLDER than the A loads still in flight), its V stores are done
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only((DG & 4) ? 0 : A_DIST));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    sv = svn; sr = srd;
    adv_d();
    adv_t();
    if constexpr (TL != 0) ++tl_steps;
  };

  // ---- write-out of one tile row of the M tile
  const int w_g = lane >> 4, w_tc = lane & 15;
  auto writeout = [&](auto tsc) {
    constexpr int TS = decltype(tsc)::value;
    const f32x4(&acc)[36] = TS == 0 ? acc0 : acc1;
    const bool has_affine = !STATS && a.scale != nullptr, has_mean = !STATS && a.mean != nullptr, has_addend = a.addend != nullptr;
    int ln_w = lane;
    TNV3_OPAQUE_V(ln_w);
    const int g = ln_w >> 4, tc = ln_w & 15;
    const int e_m0 = wM.mb * MB + 16 * swave;            // this wave's first channel
    const int oh = wM.trow * Cfg::TH + 4 * TS, ow = wM.tcol * Cfg::TW + 4 * tc;
    const size_t plane0 = ((size_t)wM.n * Cout + e_m0) * HW;
    const unsigned planes_b = 16u * (unsigned)HW * 4u;
    const tnv3_rsrc_t r_dst = tnv3_make_rsrc(a.dst + plane0, planes_b);
    const tnv3_rsrc_t r_add = tnv3_make_rsrc(has_addend ? a.addend + plane0 : a.dst + plane0, planes_b);
    const unsigned lane_off_b = oh < H ? (unsigned)((4 * g) * HW + oh * W + ow) * 4u : kDmaOob;      // (a tile row below the image: loads give 0, stores are dropped)
    f32x4 mu4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, sc4 = f32x4{1.0f, 1.0f, 1.0f, 1.0f}, sh4 = mu4;
    if (has_affine) {
      sc4 = *reinterpret_cast<const f32x4*>(a.scale + e_m0 + 4 * g);
      sh4 = *reinterpret_cast<const f32x4*>(a.shift + e_m0 + 4 * g);
      if (has_mean) mu4 = *reinterpret_cast<const f32x4*>(a.mean + e_m0 + 4 * g);
    }
    double q1[STATS ? 4 : 1], q2[STATS ? 4 : 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned ch_b = (unsigned)r * (unsigned)HW * 4u;
      f32x4 ad[4];
      if (has_addend) {
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) ad[ar] = tnv3_buf_load_f4(r_add, lane_off_b, ch_b + (unsigned)(ar * W) * 4u);
      }
      float wv[4][6];                                    // W[a][j] = sum_i A^T[a][i] M[i][j]
      wino43s_for<0, 6>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        float m[6], o[4];
        wino43s_for<0, 6>([&](auto ic) {
          constexpr int i = decltype(ic)::value, xi = 6 * i + j;
          m[i] = wino43s_acc_read<(TS == 1 && xi / 2 >= Cfg::ACC1_V_FROM)>(acc[xi], r);
        });
        wino43s_at6(m[0], m[1], m[2], m[3], m[4], m[5], o);
        wv[0][j] = o[0]; wv[1][j] = o[1]; wv[2][j] = o[2]; wv[3][j] = o[3];
      });
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int ar = 0; ar < 4; ++ar) {
        float o[4];
        wino43s_at6(wv[ar][0], wv[ar][1], wv[ar][2], wv[ar][3], wv[ar][4], wv[ar][5], o);
        f32x4 v = {o[0], o[1], o[2], o[3]};
        if (has_addend) v += ad[ar];
        if (has_affine) {
          const float mu = mu4[r], sc = sc4[r], sh = sh4[r];
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) v[b2] = (v[b2] - mu) * sc + sh;
        }
        if (!STATS && a.relu) {
#pragma unroll
          for (int b2 = 0; b2 < 4; ++b2) v[b2] = v[b2] > 0.0f ? v[b2] : 0.0f;
        }
        tnv3_buf_store_f4(r_dst, (DG & 32) ? kDmaOob : lane_off_b, ch_b + (unsigned)(ar * W) * 4u, v);
        if constexpr (STATS) {
          s1 += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
          s2 += ((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1]) + ((double)v[2] * (double)v[2] + (double)v[3] * (double)v[3]);
        }
      }
      if constexpr (STATS) { q1[r] = oh < H ? s1 : 0.0; q2[r] = oh < H ? s2 : 0.0; }
      __builtin_amdgcn_sched_barrier(0);                // one channel at a time: 36 accumulator reads, not 144, live at once
    }
    if constexpr (STATS) {
      // BatchNorm batch statistics (model.py:9 in training mode) from the epilogue's registers: the 16 lanes of a lane group hold the 16
      // tile columns of the group's four channels.  Butterflies over lane bits 3..0 in a fixed order, fp64: deterministic.  One
      // statistics tile = 4 x 64 pixels (tile row TS of the workgroup tile).
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          q1[r] += __shfl_xor(q1[r], o, 64);
          q2[r] += __shfl_xor(q2[r], o, 64);
        }
      }
      if (tc < 4) {                                      // lane tc of the group writes channel 4 g + tc
        const double v1 = tc == 0 ? q1[0] : tc == 1 ? q1[1] : tc == 2 ? q1[2] : q1[3];
        const double v2 = tc == 0 ? q2[0] : tc == 1 ? q2[1] : tc == 2 ? q2[2] : q2[3];
        const long st_tile = ((long)wM.n * (2 * tilesH) + 2 * wM.trow + TS) * tilesW + wM.tcol;
        double* o = a.stats + ((size_t)(e_m0 + 4 * g + tc) * ((size_t)a.N * 2 * tilesH * tilesW) + st_tile) * 2;
        o[0] = v1;
        o[1] = v2;
      }
    }
  };

  for (;;) {                                             // one pass per workgroup tile
    step(std::true_type{});
    ++kM;
    for (; kM < nChunks; ++kM) step(std::false_type{});
    tl_stamp(1);
#ifndef TNV3_EMU
    // MFMA result -> VALU read: 12 wait states (8-pass XDL).  The accumulators of the last two xi pairs pass THROUGH the statement, so
    // no read of them can be placed above it (every other accumulator's last MFMA is at least eight MFMAs older).
    asm volatile("s_nop 15" : "+a"(acc0[32]), "+a"(acc0[33]), "+a"(acc0[34]), "+a"(acc0[35]), "+v"(acc1[32]), "+v"(acc1[33]), "+v"(acc1[34]), "+v"(acc1[35]) : : "memory");
    static_assert(Cfg::ACC1_V_FROM <= 16, "the fence names tile row 1's last accumulators as VGPRs");
#endif
    __builtin_amdgcn_sched_barrier(0);
    writeout(std::integral_constant<int, 0>{});
    writeout(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    tl_stamp(2);
    if constexpr (TL != 0) ++tl_tiles;
    kM = 0;
    wM.next();
    if (!wM.valid) break;
  }
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
  (void)w_g; (void)w_tc;
}

}  // namespace tnv3
