// conv3x3_wino3_mfma.h -- third generation of the fused Winograd F(2x2, 3x3) forward kernel (variant 3 of
// tnv3_conv3x3_wino_forward).  Same tile, same LDS operands, same MFMA stream and epilogue as the xi-split kernel
// (conv3x3_wino_split_mfma_kernel, conv3x3_wino_mfma.h) -- and therefore bit-identical results -- but the non-MFMA work of
// a chunk is re-balanced after a cycle-level timeline of that kernel (s_memtime per phase, profiles/r02_wino_timeline*.json)
// showed what its counters only hinted at: the matrix pipe idles because ONE wave group's serial chain
// [issue 12 LDS-DMA pieces: 2700 cycles] -> [patch transform: 930] -> [32 MFMAs: 2150] is 5800 cycles long while the other
// group finishes its [MFMAs: 2200] -> [transform: 1150] in 3400 and then waits 2200 cycles at the chunk barrier.
//
//   1. DMA by buffer_load ... lds with a per-chunk scalar descriptor: every per-lane offset is chunk-invariant (6 VGPRs), the
//      chunk advance and the channel bound live in the descriptor (SALU), zero padding comes from the hardware range check
//      (offset >= num_records -> 0) -- no per-piece address arithmetic on the vector ALU (was ~86 VALU per chunk).
//   2. DMA issue split evenly over all eight waves (6 pieces each): group 0 issues its half at the start of the chunk, group 1
//      after its MFMAs -- both groups now carry the same non-MFMA load.
//   3. Patch transform by transform-row pairs: a thread produces, for TWO horizontally adjacent tiles, the two rows of
//      B^T d B that its OWN wave group consumes (xi = 8g .. 8g+7): 3 raw rows x (2 x ds_read_b128 + ds_read_b32) instead
//      of 16 stride-2 ds_read_b32 (2-way bank conflicts), 28 instead of 32 adds, 8 x ds_write_b64 instead of 16 x
//      ds_write_b32.  Same operations per element as before => the same bits.
//
// What that bought (MI355X, batch 10): 0-8 %, because the premise "one wave's VALU / LDS / DMA work hides under the other
// wave's MFMAs" is false for the fp32 MFMA.  A microbenchmark (kernels/coissue_probe.h, profiles/r02_mfma_f32_coissue.json)
// measured it directly: beside a partner wave that streams v_mfma_f32_32x32x2_f32 a wave gets ONE short burst (4 VALU, or 2
// ds_read_b32, or 1 ds_read_b128, or half an LDS-DMA piece) per 64-cycle MFMA, and inside the MFMA wave every extra instruction
// costs ~5 cycles of matrix-pipe time (MFMA + 2 VALU: 76 cycles per step instead of 66; + 2 ds_read_b32: 77).  The SIMD is
// time-multiplexed: a chunk costs its MFMAs plus ~5 cycles for EVERY other instruction either wave executes, wherever it is
// placed (which is why priorities, orders and DMA splits -- all measured, profiles/r02_wino_timeline.json -- change nothing).
// The lever is the instruction COUNT.  Hence the QUAD form (variant 4):
//   4. operand layouts with the four xi of a transform row adjacent -- U[pair][row][parity][co][4], V[pair][row][parity][tile][4]
//      -- so that ONE conflict-free ds_read_b128 per operand feeds FOUR MFMAs (0.5 instead of 2 LDS reads per MFMA), the
//      transform stores rows as ds_write_b128 (4 instead of 8 stores), and the LDS-DMA destinations are scalar (2 instead of
//      4 instructions per piece).  Every accumulator still sees its channel pairs in the same order => the same bits.
#pragma once
#include <type_traits>
#include "conv3x3_wino_mfma.h"

namespace tnv3 {

template <int CC_, int DIAG_ = 0, int DMA_MODE_ = 0, int PRIO_ = 0, int QUAD_ = 0, int SYM_ = 0, int PERSIST_ = 0, int SWAP_ = 0>
struct WinoV3Cfg {
  // Streaming kernel only.  Which wave group runs its MFMAs FIRST in a chunk: 0 = group 1 (waves 4-7), 1 = group 0 (waves 0-3) -- the
  // OLDER waves of each SIMD, which the matrix pipe's arbitration favours (69 : 31, kernels/coissue_probe.h) when two MFMA streams
  // compete.  Round 3's timeline of the 128-channel kernel (profiles/r03_wino6_timeline*.json) showed that with the younger waves
  // first, their MFMA phase is stretched by the late-starting older waves' MFMAs until BOTH end together -- and the first group's
  // patch transform then runs with the pipe idle; with the older waves first they finish early and transform under the others' MFMAs.
  static constexpr int SWAP = SWAP_;
  // 1: persistent workgroups (variant 5).  The launch has one workgroup per CU and each walks the tile list with the grid as its
  // stride (same XCD as the one-tile-per-workgroup launch gives that tile).  One workgroup fills a CU (512 threads x 256
  // registers, 150 KB of LDS), so nothing of a tile's fixed cost -- workgroup launch, index set-up, 128 accumulator writes,
  // the first DMA's round trip to HBM, the output transform and its stores: ~18 000 cycles, i.e. three chunk periods, 29 % of
  // a 64-channel layer -- overlaps anything.  The loop removes the launch and set-up, issues the NEXT tile's first DMAs before
  // the output transform of the current one (the exchange buffer moves from the filter stages to the V stages for that) and
  // starts each accumulator from the MFMA's inline zero.  Same arithmetic in the same order => the same bits.
  static constexpr int PERSIST = PERSIST_;
  // 1: both wave groups run the SAME program per chunk -- MFMAs, then their DMA pieces, then their transform rows -- instead of
  // opposite orders.  Non-MFMA work of one wave does not hide under the other wave's fp32 MFMAs anyway (header comment), and
  // next to an MFMA stream it crawls (one instruction per MFMA slot: a 700-cycle transform takes 2800); in lockstep both waves
  // of a SIMD share the matrix pipe for 2 x 32 MFMAs and then do their other work at full speed.
  static constexpr int SYM = SYM_;
  static constexpr int QUAD = QUAD_;                 // 1: quad operand layouts (filters packed with layout 1), see the header comment
  // 1: the wave raises its priority (s_setprio 3) for its DMA-issue and patch-transform phases and drops it for its MFMAs.
  // The timeline (profiles/r02_wino_timeline.json) shows that with equal priorities a wave's VALU / LDS / VMEM instructions
  // barely issue while its SIMD partner streams MFMAs (a 700-cycle transform takes 2500); an MFMA needs one issue slot per 64
  // cycles, so the matrix pipe loses nothing when everything else goes first.
  static constexpr int PRIO = PRIO_;
  static constexpr int WM = 2, WN = 2, CC = CC_;
  static constexpr int DIAG = DIAG_;                 // 7: s_memtime phase totals of one workgroup written to dst (libtnv3_diag.so)
  // who issues the chunk's LDS-DMA pieces, and when: 0 = every thread its own pieces, group 0 at the start of the chunk and
  // group 1 after its MFMAs; 1 = group 1 issues ALL pieces after its MFMAs; 2 = every thread its own pieces, both groups
  // after their MFMAs
  static constexpr int DMA_MODE = DMA_MODE_;
  static constexpr int NTD = DMA_MODE_ == 1 ? WM * WN * 64 : 2 * WM * WN * 64;     // threads that issue DMA
  static constexpr int NT = 2 * WM * WN * 64;        // 512 threads: waves 0-3 = xi group 0, waves 4-7 = xi group 1
  static constexpr int MB = 32 * WM, TB = 32 * WN, PW = 32 * WN;
  static constexpr int RW = PW + 8, RAWP = 6 * RW;
  static constexpr int RAW_FLOATS = CC * RAWP;
  static constexpr int U_FLOATS = CC * 16 * MB, V_FLOATS = CC * 16 * TB;
  static constexpr int NU4 = U_FLOATS / 4 / NTD;                         // filter pieces per issuing thread and chunk (4 or 8)
  static constexpr int NRAW = (RAW_FLOATS / 4 + NTD - 1) / NTD;          // raw pieces per issuing thread and chunk (2 or 4)
  static constexpr int RAW_STAGE = NRAW * NTD * 4;
  static constexpr int LDS_FLOATS = 2 * U_FLOATS + 2 * V_FLOATS + 2 * RAW_STAGE;
  static constexpr int XCH_FLOATS = (NT / 64) * 32 * 64;
  static_assert((U_FLOATS / 4) % NTD == 0, "filter panel must deal evenly");
  static_assert(CC * (TB / 2) == NT / 2, "one tile pair per thread, group and chunk");
  static_assert(XCH_FLOATS <= 2 * U_FLOATS, "the exchange reuses the filter stages");
  static_assert((RW % 4) == 0 && (RAWP % 4) == 0, "16-byte rows");
  static_assert(!QUAD_ || (DMA_MODE_ == 0 && CC_ == 8), "the quad form is built for 8-channel chunks, every thread moving its own pieces");
};

// ---- write-out of a finished 64-channel x 64-tile accumulator block (shared by the kernels of this file)
// acc: this wave's 8 accumulators (xi rows 2*grp, 2*grp+1 of its 32 x 32 block);  xch_g0 / xch_g1: 32 KB of free LDS each -- the
// partial sums group 0 / group 1 hand to the partner group;  e_*: the tile (image, first row / column, first channel, tile index).
// STREAM: the caller goes straight on to its next chunk, so the function ends with everybody done reading the regions.
template <class Cfg, bool OPAQUE, bool STREAM, class StampMid>
__device__ __forceinline__ void wino3_writeout(const WinoArgs& a, const f32x16 (&acc)[8], float* xch_g0, float* xch_g1, int e_n, int e_h0, int e_w0,
                                               int e_m0, int e_pt, int nPT, StampMid&& stamp_mid) {
  constexpr int WN = Cfg::WN, MB = Cfg::MB, TB = Cfg::TB;
  const int H = a.H, W = a.W, Cout = a.Cout, HW = H * W;
  (void)H;
  // ---- inverse transform: rows of A^T M (this group's two xi rows), columns, then the halves meet through LDS
  int tid_e = threadIdx.x;
  if constexpr (OPAQUE) TNV3_OPAQUE_V(tid_e);           // the write-out's lane arithmetic is redone per tile, not kept live across the MFMA loop
  const int tid = tid_e, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, wq = wave & 3;
  const int wn = wq % WN, wm = wq / WN, half = lane >> 5, bl = lane & 31;
  // group g finishes output row g of every tile: `own` is its partial sum of that row; its partial sum of the other row goes
  // to the partner wave through LDS.  The group is wave-uniform: one scalar branch, no per-element selects.
  float* xch_wr = (grp ? xch_g1 : xch_g0) + wq * (32 * 64) + lane;       // this wave's 8 KB slot of its group's region
  const float* xch_rd = (grp ? xch_g0 : xch_g1) + wq * (32 * 64) + lane;  // the partner wave's slot
  float own[16][2];
  auto out_rows = [&](auto gc) {
    constexpr int G = decltype(gc)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float tt[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = acc[j][r], hi = acc[4 + j][r];  // M rows 2*grp and 2*grp + 1
        tt[0][j] = G ? lo : lo + hi;                     // A^T row 0 = [1 1 1 0]
        tt[1][j] = G ? -lo - hi : hi;                    // A^T row 1 = [0 1 -1 -1]
      }
      float pv[2][2];
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        pv[y][0] = tt[y][0] + tt[y][1] + tt[y][2];
        pv[y][1] = tt[y][1] - tt[y][2] - tt[y][3];
      }
      own[r][0] = pv[G][0]; own[r][1] = pv[G][1];
#pragma unroll
      for (int x = 0; x < 2; ++x) xch_wr[(r * 2 + x) * 64] = pv[1 - G][x];
    }
  };
  if (__builtin_amdgcn_readfirstlane(grp)) out_rows(std::true_type{}); else out_rows(std::false_type{});
  __syncthreads();
  stamp_mid();
  // write-out: every load this lane needs -- the partner's 32 partial sums, the 16 channels' BatchNorm constants (four 16-byte
  // loads each: the lane's channels are 4 runs of 4), the addend -- is issued BEFORE the first use, and the 16 stores go out
  // back to back through one scalar base per channel + one 32-bit lane offset (no per-store 64-bit address arithmetic, no
  // s_waitcnt between a store and the next channel's loads: that chain used to cost 6000 cycles per tile).
  const bool has_affine = a.scale != nullptr, has_mean = a.mean != nullptr, has_addend = a.addend != nullptr;
  const int t = wn * 32 + bl;
  const int tr = t / (TB / 2), tc = t - tr * (TB / 2);
  const int oh = e_h0 + 2 * tr + grp, ow = e_w0 + 2 * tc;
  // bytes from channel plane (n, m0) to this lane's pixel pair (< 2^31: host check): the one per-lane part of every address below.
  // The 64 planes of the tile's channel block sit behind one buffer descriptor; channel r of the lane's run is a scalar offset.
  const unsigned lane_off_b = (unsigned)((wm * 32 + 4 * half) * HW + oh * W + ow) * 4u;
  const size_t plane0 = ((size_t)e_n * Cout + e_m0) * HW;
  const unsigned planes_b = (unsigned)MB * (unsigned)HW * 4u;
  const tnv3_rsrc_t r_dst = tnv3_make_rsrc(a.dst + plane0, planes_b);
  auto chan_off = [&](int r) -> unsigned { return (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)HW * 4u; };
  float got[16][2];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int x = 0; x < 2; ++x) got[r][x] = xch_rd[(r * 2 + x) * 64];
  typedef float wf2 __attribute__((ext_vector_type(2)));
  wf2 ad[16];
  if (has_addend) {
    const tnv3_rsrc_t r_add = tnv3_make_rsrc(a.addend + plane0, planes_b);
#pragma unroll
    for (int r = 0; r < 16; ++r) ad[r] = tnv3_buf_load_f2(r_add, lane_off_b, chan_off(r));
  }
  const bool want_stats = a.stats != nullptr;            // wave-uniform (kernel argument); never together with the affine (host check)
  if (want_stats) {                                      // training forward: raw convolution (+ addend) out, statistics from the same registers
    double q1[16], q2[16];                               // this lane's two pixels per channel r: sum, sum of squares
    const bool bn_bwd = a.bn_z != nullptr;               // wave-uniform: data-gradient launch that also takes BatchNorm backward's two sums
    if (bn_bwd) {                                        // (no addend in that mode, host check: its registers hold the z pairs)
      const tnv3_rsrc_t r_z = tnv3_make_rsrc(a.bn_z + plane0, planes_b);
#pragma unroll
      for (int r = 0; r < 16; ++r) ad[r] = tnv3_buf_load_f2(r_z, lane_off_b, chan_off(r));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      wf2 v;
      v[0] = own[r][0] + got[r][0];
      v[1] = own[r][1] + got[r][1];
      if (has_addend) { v[0] += ad[r][0]; v[1] += ad[r][1]; }
      if (a.relu) { v[0] = v[0] > 0.0f ? v[0] : 0.0f; v[1] = v[1] > 0.0f ? v[1] : 0.0f; }
      tnv3_buf_store_f2(r_dst, lane_off_b, chan_off(r), v);
      if (!bn_bwd) {
        q1[r] = (double)v[0] + (double)v[1];
        q2[r] = (double)v[0] * (double)v[0] + (double)v[1] * (double)v[1];
      } else {                                           // g = dA * [BN(z) > 0] (the forward's own expression: bit-identical mask), xhat = (z - mean) * invstd
        const int ch = e_m0 + wm * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.bn_c4 + 4 * (size_t)ch);
        const float g0 = fmaf(ad[r][0] - c4[0], c4[2], c4[3]) > 0.0f ? v[0] : 0.0f;
        const float g1 = fmaf(ad[r][1] - c4[0], c4[2], c4[3]) > 0.0f ? v[1] : 0.0f;
        q1[r] = (double)g0 + (double)g1;
        q2[r] = (double)g0 * (double)((ad[r][0] - c4[0]) * c4[1]) + (double)g1 * (double)((ad[r][1] - c4[0]) * c4[1]);
      }
    }
    // BatchNorm batch statistics from the epilogue's registers (model.py:9 in training mode): a half-wave holds 32 tiles x 2 pixels
    // of 16 channels.  Reduce-scatter butterfly over the 32 lanes: at offset o the lane keeps the half of its channel list its
    // bit selects and adds the partner's copy of that half -- 8 + 4 + 2 + 1 exchanges, then one plain exchange at offset 1;
    // lane bl ends up with channel index r = bl >> 1 summed over the half-wave.  fp64, fixed order: deterministic.
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const int o = 16 >> step, cnt = 8 >> step;         // cnt values survive this step
      const bool up = (bl & o) != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < cnt) {
          const double s1 = up ? q1[i] : q1[i + cnt], k1 = up ? q1[i + cnt] : q1[i];
          const double s2 = up ? q2[i] : q2[i + cnt], k2 = up ? q2[i + cnt] : q2[i];
          q1[i] = k1 + __shfl_xor(s1, o, 64);
          q2[i] = k2 + __shfl_xor(s2, o, 64);
        }
      }
    }
    q1[0] += __shfl_xor(q1[0], 1, 64);
    q2[0] += __shfl_xor(q2[0], 1, 64);
    __syncthreads();                                       // everybody has read the exchange buffer: LDS is free again
    double* red = reinterpret_cast<double*>(xch_g0);       // [wave][half][16 r][2]
    if ((bl & 1) == 0) {
      double* d = red + ((wave * 2 + half) * 16 + (bl >> 1)) * 2;
      d[0] = q1[0];
      d[1] = q2[0];
    }
    __syncthreads();
    if (tid < MB) {                                        // channel tid of this block: fold the four waves that own its pixels
      const int cwm = tid >> 5, q = tid & 31;
      const int chalf = (q >> 2) & 1, cr = (q & 3) + 4 * (q >> 3);
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int x = 0; x < WN; ++x) {
          const double* d = red + (((g * 4 + cwm * WN + x) * 2 + chalf) * 16 + cr) * 2;
          s1 += d[0];
          s2 += d[1];
        }
      double* o = a.stats + ((size_t)(e_m0 + tid) * nPT + e_pt) * 2;
      o[0] = s1;
      o[1] = s2;
    }
    if constexpr (STREAM) __syncthreads();                 // the fold has read `red`: the next chunk may overwrite the regions
  } else {
    if constexpr (STREAM) __syncthreads();                 // (after the waits the compiler puts before a barrier) the partial sums are in registers:
                                                           // the next chunk's DMAs and transform may overwrite the regions
    f32x4 mu4[4], sc4[4], sh4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mu4[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; sc4[q] = f32x4{1.0f, 1.0f, 1.0f, 1.0f}; sh4[q] = mu4[q];
    }
    if (has_affine) {
      const int c4 = e_m0 + wm * 32 + 4 * half;          // channel of r = 0; r -> c4 + (r & 3) + 8 * (r >> 2)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sc4[q] = *reinterpret_cast<const f32x4*>(a.scale + c4 + 8 * q);
        sh4[q] = *reinterpret_cast<const f32x4*>(a.shift + c4 + 8 * q);
      }
      if (has_mean) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mu4[q] = *reinterpret_cast<const f32x4*>(a.mean + c4 + 8 * q);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      wf2 v;
      v[0] = own[r][0] + got[r][0];
      v[1] = own[r][1] + got[r][1];
      if (has_addend) { v[0] += ad[r][0]; v[1] += ad[r][1]; }
      if (has_affine) {
        const float mu = mu4[r >> 2][r & 3], sc = sc4[r >> 2][r & 3], sh = sh4[r >> 2][r & 3];
        v[0] = (v[0] - mu) * sc + sh; v[1] = (v[1] - mu) * sc + sh;
      }
      if (a.relu) { v[0] = v[0] > 0.0f ? v[0] : 0.0f; v[1] = v[1] > 0.0f ? v[1] : 0.0f; }
      tnv3_buf_store_f2(r_dst, lane_off_b, chan_off(r), v);
    }
  }
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv3x3_wino_v3_mfma_kernel(const WinoArgs a) {
  constexpr int WN = Cfg::WN, CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TB = Cfg::TB, PW = Cfg::PW, RW = Cfg::RW, RAWP = Cfg::RAWP;
  constexpr int NU4 = Cfg::NU4, NRAW = Cfg::NRAW, NTD = Cfg::NTD, DMA_MODE = Cfg::DMA_MODE;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* u_s = lds;                                   // two stages
  float* v_s = lds + 2 * Cfg::U_FLOATS;               // two stages
  float* raw_s = v_s + 2 * Cfg::V_FLOATS;             // two stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wn = wq % WN, wm = wq / WN;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = H / 4, tilesW = W / PW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (Cin + CC - 1) / CC;
  // tile list: entry b of the XCD-aware order (conv_block_map); a persistent workgroup takes b = blockIdx.x, + gridDim.x, ...
  const int nItems = Cfg::PERSIST ? conv_grid_blocks(nMB, nPT) : 0;
  int item = blockIdx.x;
  int mb, pt;
  if constexpr (Cfg::PERSIST) {
    while (item < nItems && !conv_block_map(item, nMB, nPT, mb, pt)) item += gridDim.x;
    if (item >= nItems) return;
  } else {
    if (!conv_block_map(item, nMB, nPT, mb, pt)) return;
  }
  int n = pt / (tilesH * tilesW);
  int trem = pt - n * (tilesH * tilesW);
  int h0 = (trem / tilesW) * 4, w0 = (trem % tilesW) * PW;
  int m0 = mb * MB;

  // ---- chunk-invariant per-lane byte offsets of this thread's DMA pieces
  unsigned vo_u[NU4], vo_r[NRAW];
#pragma unroll
  for (int i = 0; i < NU4; ++i) {                     // filter piece e4 of [CC*16 rows][MB/4]: row = (ci, xi), 16 bytes of 64 channels
    const int e4 = (tid & (NTD - 1)) + i * NTD;
    if (Cfg::QUAD) {                                    // packed [pair][row][parity][Cout][4]: piece e4 = (prow = (pair, row, parity), co)
      const int prow = e4 / MB, co = e4 - prow * MB;
      vo_u[i] = (unsigned)(prow * Cout + co) * 16u;
    } else {
      const int row = e4 / (MB / 4), m4 = e4 - row * (MB / 4);
      vo_u[i] = (unsigned)(row * Cout + m4 * 4) * 4u;
    }
  }
  const float* u_base;
  const float* x_base;
  auto set_item = [&]() {                               // what changes from tile to tile: the raw pieces' offsets and the two bases
    int t_op = tid;
    if constexpr (Cfg::PERSIST) TNV3_OPAQUE_V(t_op);
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {                  // raw piece e of [CC][6][RW/4]; padding / unused slots read out of range (= 0)
      const int e = (t_op & (NTD - 1)) + i * NTD;
      const int c = e / (RAWP / 4), r = e - c * (RAWP / 4);
      const int tr = r / (RW / 4), q = r - tr * (RW / 4);
      const int gh = h0 - 1 + tr, gw = w0 - 4 + 4 * q;
      const bool ok = e < Cfg::RAW_FLOATS / 4 && gh >= 0 && gh < H && gw >= 0 && gw < W;
      vo_r[i] = ok ? (unsigned)(c * HW + gh * W + gw) * 4u : kDmaOob;
    }
    u_base = a.u + (Cfg::QUAD ? (size_t)m0 * 4 : (size_t)m0);
    x_base = a.src + (size_t)n * Cin * HW;
  };
  set_item();
  const int wbase = __builtin_amdgcn_readfirstlane((wave & (NTD / 64 - 1)) * 64);      // scalar: the LDS-DMA destinations (M0) stay on the SALU
  // half h (0: issued by group 0 at the start of the chunk, 1: by group 1 after its MFMAs) of the DMAs of one chunk:
  // filters of chunk ku -> stage ku & 1, raw tile of chunk kr -> stage kr & 1.  A thread moves its own NU4 + NRAW pieces.
  auto dma_chunk = [&](int ku, int kr) {
    if (ku < nChunks) {
      const tnv3_rsrc_t ru = tnv3_make_rsrc(u_base + (size_t)ku * CC * 16 * Cout, (unsigned)(CC * 16 * Cout) * 4u);
      float* us = u_s + (ku & 1) * Cfg::U_FLOATS;
#pragma unroll
      for (int i = 0; i < NU4; ++i) tnv3_buf_dma16(ru, us + (i * NTD + wbase) * 4, vo_u[i]);
    }
    if (kr < nChunks) {
      const int cleft = Cin - kr * CC;                // channels past Cin lie beyond num_records: zero
      const tnv3_rsrc_t rr = tnv3_make_rsrc(x_base + (size_t)kr * CC * HW, (unsigned)(cleft < CC ? cleft : CC) * (unsigned)HW * 4u);
      float* rs = raw_s + (kr & 1) * Cfg::RAW_STAGE;
#pragma unroll
      for (int i = 0; i < NRAW; ++i) tnv3_buf_dma16(rr, rs + (i * NTD + wbase) * 4, vo_r[i]);
    }
  };

  // ---- patch transform: thread -> (channel c, tile row tr, tile pair pj), its group's two transform rows
  const int tg = tid & (NT / 2 - 1);
  const int pc = tg / (TB / 2), prem = tg - pc * (TB / 2);
  const int ptr_ = prem / (TB / 4), pj = prem - ptr_ * (TB / 4);
  const int t_src = pc * RAWP + (2 * ptr_ + grp) * RW + 4 * pj;                 // first of three raw rows, 16-byte aligned
  const int t_dst = Cfg::QUAD ? ((((pc >> 1) * 4 + 2 * grp) * 2 + (pc & 1)) * TB + ptr_ * (TB / 2) + 2 * pj) * 4     // [pair][row][parity][tile][4]
                              : (pc * 16 + grp * 8) * TB + ptr_ * (TB / 2) + 2 * pj;   // xi = 8*grp .. 8*grp+7, tiles (2pj, 2pj+1)
  typedef float wf2 __attribute__((ext_vector_type(2)));
  auto transform = [&](int stage) {                     // raw stage -> V stage of the same parity
    const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src;
    float x[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {                       // patch columns 4pj+3 .. 4pj+8 of raw row r
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * RW);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * RW + 4);
      const float q2 = d[r * RW + 8];
      x[r][0] = q0[3]; x[r][1] = q1[0]; x[r][2] = q1[1]; x[r][3] = q1[2]; x[r][4] = q1[3]; x[r][5] = q2;
    }
    float e[2][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      // group 0 holds patch rows d0,d1,d2 -> (B^T d) rows 0,1;  group 1 holds d1,d2,d3 -> rows 2,3
      e[0][j] = grp ? x[1][j] - x[0][j] : x[0][j] - x[2][j];      // d2 - d1      | d0 - d2
      e[1][j] = grp ? x[0][j] - x[2][j] : x[1][j] + x[2][j];      // d1 - d3      | d1 + d2
    }
    float* v = v_s + stage * Cfg::V_FLOATS + t_dst;
    if constexpr (Cfg::QUAD) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {                       // transform row 2*grp + r of both tiles: four xi each, one 16-byte store
        f32x4 o;
        o[0] = e[r][0] - e[r][2]; o[1] = e[r][1] + e[r][2]; o[2] = e[r][2] - e[r][1]; o[3] = e[r][1] - e[r][3];
        *reinterpret_cast<f32x4*>(v + r * (2 * TB * 4)) = o;
        o[0] = e[r][2] - e[r][4]; o[1] = e[r][3] + e[r][4]; o[2] = e[r][4] - e[r][3]; o[3] = e[r][3] - e[r][5];
        *reinterpret_cast<f32x4*>(v + r * (2 * TB * 4) + 4) = o;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      wf2 o;
      o[0] = e[r][0] - e[r][2]; o[1] = e[r][2] - e[r][4]; *reinterpret_cast<wf2*>(v + (r * 4 + 0) * TB) = o;
      o[0] = e[r][1] + e[r][2]; o[1] = e[r][3] + e[r][4]; *reinterpret_cast<wf2*>(v + (r * 4 + 1) * TB) = o;
      o[0] = e[r][2] - e[r][1]; o[1] = e[r][4] - e[r][3]; *reinterpret_cast<wf2*>(v + (r * 4 + 2) * TB) = o;
      o[0] = e[r][1] - e[r][3]; o[1] = e[r][3] - e[r][5]; *reinterpret_cast<wf2*>(v + (r * 4 + 3) * TB) = o;
    }
  };

  f32x16 acc[8];
  if constexpr (!Cfg::PERSIST) {
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  }
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;

  const int a_off = (half * 16 + grp * 8) * MB + wm * 32 + bl;
  const int b_off = (half * 16 + grp * 8) * TB + wn * 32 + bl;
  // quad form: float offsets of this lane's 16-byte operand groups inside a stage: [pair][row][parity][64][4], row = 2*grp + j
  const int aq_off = ((2 * grp * 2 + half) * MB + wm * 32 + bl) * 4;
  const int bq_off = ((2 * grp * 2 + half) * TB + wn * 32 + bl) * 4;
  auto mfma_chunk = [&](int k, auto first_c) {          // first_c: the tile's first chunk starts every accumulator from the inline zero
    constexpr bool FIRST = decltype(first_c)::value;
    if constexpr (Cfg::QUAD) {
      const float* A = u_s + (k & 1) * Cfg::U_FLOATS + aq_off;
      const float* B = v_s + (k & 1) * Cfg::V_FLOATS + bq_off;
      constexpr int PSTR = 4 * 2 * MB * 4, RSTR = 2 * MB * 4;     // floats per channel pair / per transform row (MB == TB)
      f32x4 av[2][2], bv[2][2];                            // [ring][row j]: one 16-byte read feeds the four xi of a row
      auto read_pair = [&](int cp) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          av[cp & 1][j] = *reinterpret_cast<const f32x4*>(A + cp * PSTR + j * RSTR);
          bv[cp & 1][j] = *reinterpret_cast<const f32x4*>(B + cp * PSTR + j * RSTR);
        }
      };
      read_pair(0);
#pragma unroll
      for (int cp = 0; cp < CC / 2; ++cp) {
        if (cp + 1 < CC / 2) read_pair(cp + 1);            // the next pair's four reads go out BEFORE this pair's eight MFMAs
        __builtin_amdgcn_sched_barrier(0);                 // (the scheduler would otherwise sink them next to their use)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x)
            acc[j * 4 + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cp & 1][j][x], bv[cp & 1][j][x], FIRST && cp == 0 ? zero16 : acc[j * 4 + x], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    const float* A = u_s + (k & 1) * Cfg::U_FLOATS + a_off;
    const float* B = v_s + (k & 1) * Cfg::V_FLOATS + b_off;
    constexpr int NSTEP = (CC / 2) * 8;                  // (channel pair, xi of this group): one MFMA each
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int cp = s >> 3, x = s & 7;
      av[s % RING] = A[(2 * cp * 16 + x) * MB];
      bv[s % RING] = B[(2 * cp * 16 + x) * TB];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], FIRST && s < 8 ? zero16 : acc[s & 7], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  auto chunk_barrier = [&]() {                          // own DMAs landed, own V writes done, everybody finished with the old stages
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  unsigned long long t_acc0 = 0, t_acc1 = 0, t_acc2 = 0, t_acc3 = 0, t_acc4 = 0, t_acc5 = 0, t_last = 0;
  auto stamp_to = [&](int slot) {
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      const unsigned long long dcy = now - t_last;
      t_last = now;
      if (slot == 0) t_acc0 += dcy; else if (slot == 1) t_acc1 += dcy; else if (slot == 2) t_acc2 += dcy;
      else if (slot == 3) t_acc3 += dcy; else if (slot == 4) t_acc4 += dcy; else if (slot == 5) t_acc5 += dcy;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stamp = [&](int slot) {
    if constexpr (Cfg::DIAG == 7) stamp_to(slot);
  };

  auto stamp8 = [&](int slot) {                         // DIAG 8: per-tile fixed-cost phases of a persistent workgroup
    if constexpr (Cfg::DIAG == 8) stamp_to(slot);
  };
  // first DMAs of a tile: filters of chunk 0, raw tiles of chunks 0 and 1 (every thread its own pieces)
  auto issue_first = [&]() {
    if (DMA_MODE != 1 || grp == 1) {
      dma_chunk(0, 0);
      dma_chunk(nChunks, 1);
    }
  };
  auto chunk_body = [&](int k, auto first_c) {
    // free since the barrier that ended chunk k-1: filter stage (k+1)&1 (held chunk k-1), raw stage k&1 (held chunk k,
    // transformed during chunk k-1) and V stage (k+1)&1 (read by the MFMAs of chunk k-1)
    const bool more = k + 1 < nChunks;
    stamp(0);
    if (grp == 0 && !Cfg::SYM) {
      if (Cfg::PRIO) __builtin_amdgcn_s_setprio(3);
      // each thread owns NU4 + NRAW pieces of the chunk; group 0's threads issue theirs now, group 1's threads theirs after
      // the MFMAs -- half of the chunk's bytes each.
      if (DMA_MODE == 0) dma_chunk(k + 1, k + 2);
      stamp(1);
      if (more) transform((k + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (Cfg::PRIO) __builtin_amdgcn_s_setprio(0);
      stamp(2);
      mfma_chunk(k, first_c);
      stamp(3);
      if (DMA_MODE == 2) { __builtin_amdgcn_sched_barrier(0); dma_chunk(k + 1, k + 2); }
    } else {
      mfma_chunk(k, first_c);
      __builtin_amdgcn_sched_barrier(0);
      if (Cfg::PRIO) __builtin_amdgcn_s_setprio(3);
      stamp(1);
      dma_chunk(k + 1, k + 2);
      stamp(2);
      if (more) transform((k + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (Cfg::PRIO) __builtin_amdgcn_s_setprio(0);
      stamp(3);
    }
    if constexpr (Cfg::DIAG == 7) {
      __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
      stamp(4);
      __builtin_amdgcn_s_barrier();
      stamp(5);
    } else {
      chunk_barrier();
    }
  };

  unsigned long long t_begin = 0;
  int n_items_done = 0;
  if constexpr (Cfg::DIAG == 8) { t_begin = __builtin_amdgcn_s_memtime(); t_last = t_begin; }
  issue_first();
  for (;;) {                                            // one pass per tile; a non-persistent workgroup leaves after the first
  chunk_barrier();                                      // first DMAs landed; (persistent) everybody is done with the exchange buffer
  stamp8(0);
  transform(0);
  chunk_barrier();
  stamp8(1);
  if constexpr (Cfg::PERSIST) {
    chunk_body(0, std::true_type{});
    for (int k = 1; k < nChunks; ++k) chunk_body(k, std::false_type{});
  } else {
    for (int k = 0; k < nChunks; ++k) chunk_body(k, std::false_type{});
  }
  stamp8(2);
  if constexpr (Cfg::DIAG == 7) {
    float keep = 0.0f;                                     // the accumulators must stay live or the MFMAs are dead code
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) keep += acc[x][r];
    if (blockIdx.x == gridDim.x / 2) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dst) + wave * 8;
      if (lane == 0) { o[0] = t_acc0; o[1] = t_acc1; o[2] = t_acc2; o[3] = t_acc3; o[4] = t_acc4; o[5] = t_acc5; o[6] = (unsigned long long)nChunks; }
      if (keep == 1234.5678f) o[7] = 1;
    }
    return;
  }

  // ---- the tile whose accumulators are complete; a persistent workgroup moves on to its next tile and gets that one's first
  //      DMAs going (filter stage 0, both raw stages: all free since the last chunk barrier) before it writes this one out
  const int e_n = n, e_h0 = h0, e_w0 = w0, e_m0 = m0, e_pt = pt;
  bool have_next = false;
  if constexpr (Cfg::PERSIST) {
    item += gridDim.x;
    while (item < nItems && !conv_block_map(item, nMB, nPT, mb, pt)) item += gridDim.x;
    have_next = item < nItems;
    if (have_next) {
      n = pt / (tilesH * tilesW);
      trem = pt - n * (tilesH * tilesW);
      h0 = (trem / tilesW) * 4; w0 = (trem % tilesW) * PW;
      m0 = mb * MB;
      set_item();
      issue_first();
    }
  }
  stamp8(3);

  // ---- inverse transform + write-out (wino3_writeout); the exchange regions: all stages are free after the last chunk barrier,
  //      except that the persistent form has the next tile's filters on their way into the filter stages
  {
    float* xch = Cfg::PERSIST ? v_s : lds;
    wino3_writeout<Cfg, Cfg::PERSIST != 0, false>(a, acc, xch, xch + 4 * 32 * 64, e_n, e_h0, e_w0, e_m0, e_pt, nPT, [&]() { stamp8(4); });
  }
  stamp8(5);
  ++n_items_done;
  if (!have_next) break;
  }                                                     // tile loop
  if constexpr (Cfg::DIAG == 8) {                       // phase totals of one workgroup, behind the N-th image of dst (the caller allocates N + 1)
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dst + (size_t)a.N * Cout * HW) + wave * 8;
      o[0] = t_acc0; o[1] = t_acc1; o[2] = t_acc2; o[3] = t_acc3; o[4] = t_acc4; o[5] = t_acc5;
      o[6] = (unsigned long long)n_items_done; o[7] = __builtin_amdgcn_s_memtime() - t_begin;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Streaming kernel (variant 5): persistent workgroups whose chunk pipeline runs THROUGH the tile boundaries.
//
// A workgroup of the kernel above fills its CU (512 threads x 256 registers, 160 KB of LDS), so nothing of a tile's fixed cost
// overlaps anything: workgroup launch (~7000 cycles seen from outside), index set-up, 128 accumulator writes, the first DMAs'
// round trip to HBM (1300-4600), the first patch transform (1200), the output transform and its stores (profiles/
// r02_wino_fixed_cost.json: ~14 000 cycles per tile inside the kernel, i.e. 2.5 chunk periods -- a quarter of a 64-channel
// layer).  Here one workgroup per CU walks the tile list (ConvTileWalk: no divisions per tile) and treats the chunks of
// consecutive tiles as ONE stream: stage parity follows a running chunk counter, and at the last chunks of a tile the
// "next chunk" DMAs (filters one chunk ahead, raw tile two ahead) and the patch transform simply belong to the next tile.
// At a tile boundary everything the next tile's first chunk needs is already in LDS; the only per-tile cost left is the output
// transform + write-out (wino3_writeout), which uses the two stages the last chunk just released.  Accumulators start from
// the MFMA's inline zero (no 128 register writes).  Same arithmetic in the same order as variants 2 / 3 => the same bits.
// Needs Cin > 8 (two chunks per tile; the host falls back to variant 3 otherwise).
template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv3x3_wino_stream_mfma_kernel(const WinoArgs a) {
  static_assert(!Cfg::QUAD && !Cfg::SYM && Cfg::DMA_MODE == 0, "the streaming kernel is built on the variant-3 schedule");
  constexpr int CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TB = Cfg::TB, PW = Cfg::PW, RW = Cfg::RW, RAWP = Cfg::RAWP;
  constexpr int NU4 = Cfg::NU4, NRAW = Cfg::NRAW, NTD = Cfg::NTD, WN = Cfg::WN;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* u_s = lds;                                   // two stages each
  float* v_s = lds + 2 * Cfg::U_FLOATS;
  float* raw_s = v_s + 2 * Cfg::V_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wq = wave & 3;
  const int wn = wq % WN, wm = wq / WN;
  const int half = lane >> 5, bl = lane & 31;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = H / 4, tilesW = W / PW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (Cin + CC - 1) / CC;             // >= 2 (host)

  ConvTileWalk walk;                                  // always one tile ahead of the one being computed
  walk.init(blockIdx.x, gridDim.x, nMB, nPT, tilesH, tilesW);
  if (!walk.valid) return;

  // ---- per-lane DMA offsets: the filter pieces' are the same for every tile and chunk, the raw pieces' follow the tile
  unsigned vo_u[NU4], vo_r[NRAW], vo_rn[NRAW];
#pragma unroll
  for (int i = 0; i < NU4; ++i) {                     // filter piece e4 of [CC*16 rows][MB/4]: row = (ci, xi), 16 bytes of 64 channels
    const int e4 = tid + i * NTD;
    const int row = e4 / (MB / 4), m4 = e4 - row * (MB / 4);
    vo_u[i] = (unsigned)(row * Cout + m4 * 4) * 4u;
  }
  auto raw_offsets = [&](unsigned (&vo)[NRAW], int h0, int w0) {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);                              // recomputed per tile; nothing of it stays live across the chunk loop
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {                  // raw piece e of [CC][6][RW/4]; padding / unused slots read out of range (= 0)
      const int e = t_op + i * NTD;
      const int c = e / (RAWP / 4), r = e - c * (RAWP / 4);
      const int tr = r / (RW / 4), q = r - tr * (RW / 4);
      const int gh = h0 - 1 + tr, gw = w0 - 4 + 4 * q;
      const bool ok = e < Cfg::RAW_FLOATS / 4 && gh >= 0 && gh < H && gw >= 0 && gw < W;
      vo[i] = ok ? (unsigned)(c * HW + gh * W + gw) * 4u : kDmaOob;
    }
  };
  // the tile being computed and the one after it
  int c_n = walk.n, c_h0 = walk.trow * 4, c_w0 = walk.tcol * PW, c_m0 = walk.mb * MB, c_pt = walk.pt;
  raw_offsets(vo_r, c_h0, c_w0);
  walk.next();
  bool have_next = walk.valid;
  int n_n = walk.n, n_h0 = walk.trow * 4, n_w0 = walk.tcol * PW, n_m0 = walk.mb * MB, n_pt = walk.pt;
  if (have_next) raw_offsets(vo_rn, n_h0, n_w0);

  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);      // scalar: the LDS-DMA destinations (M0) stay on the SALU
  // filter chunk at `up` (CC*16 rows of Cout floats, this tile's 64 channels first) -> filter stage su
  const size_t u_step = (size_t)CC * 16 * Cout, x_step = (size_t)CC * HW;       // floats per chunk
  auto dma_u = [&](const float* up, int su) {
    const tnv3_rsrc_t ru = tnv3_make_rsrc(up, (unsigned)(CC * 16 * Cout) * 4u);
    float* us = u_s + su * Cfg::U_FLOATS;
#pragma unroll
    for (int i = 0; i < NU4; ++i) tnv3_buf_dma16(ru, us + (i * NTD + wbase) * 4, vo_u[i]);
  };
  // raw tile of the chunk whose first channel plane is `xp` (cvalid of its CC channels exist; per-lane offsets vo) -> raw stage sr
  auto dma_r = [&](const float* xp, int cvalid, const unsigned (&vo)[NRAW], int sr) {
    const tnv3_rsrc_t rr = tnv3_make_rsrc(xp, (unsigned)(cvalid < CC ? cvalid : CC) * (unsigned)HW * 4u);   // channels past Cin: beyond num_records, zero
    float* rs = raw_s + sr * Cfg::RAW_STAGE;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) tnv3_buf_dma16(rr, rs + (i * NTD + wbase) * 4, vo[i]);
  };
  // scalar bases of the two tiles' filters and images
  const float* c_u = a.u + c_m0;
  const float* c_x = a.src + (size_t)c_n * Cin * HW;
  const float* n_u = a.u + n_m0;
  const float* n_x = a.src + (size_t)n_n * Cin * HW;

  // ---- patch transform: thread -> (channel c, tile row tr, tile pair pj), its group's two transform rows (as in the kernel above)
  const int tg = tid & (NT / 2 - 1);
  const int pc = tg / (TB / 2), prem = tg - pc * (TB / 2);
  const int ptr_ = prem / (TB / 4), pj = prem - ptr_ * (TB / 4);
  const int t_src = pc * RAWP + (2 * ptr_ + grp) * RW + 4 * pj;
  const int t_dst = (pc * 16 + grp * 8) * TB + ptr_ * (TB / 2) + 2 * pj;
  typedef float wf2 __attribute__((ext_vector_type(2)));
  // The transform in two parts, so that the LDS-DMA issue of the chunk can sit between the patch reads and their first use
  // (the reads' latency then hides behind the DMA instructions instead of stalling the wave right after them).
  float tx[3][6];
  auto transform_read = [&](int stage) {                // patch columns 4pj+3 .. 4pj+8 of the three raw rows this group needs
    if constexpr (Cfg::DIAG == 10 || Cfg::DIAG == 13) return;       // timing experiments (wrong results): no patch transform at all
    const float* d = raw_s + stage * Cfg::RAW_STAGE + t_src;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(d + r * RW);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(d + r * RW + 4);
      const float q2 = d[r * RW + 8];
      tx[r][0] = q0[3]; tx[r][1] = q1[0]; tx[r][2] = q1[1]; tx[r][3] = q1[2]; tx[r][4] = q1[3]; tx[r][5] = q2;
    }
  };
  auto transform_finish = [&](int stage) {              // raw stage -> V stage of the same parity
    if constexpr (Cfg::DIAG == 10 || Cfg::DIAG == 13) return;
    float e[2][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      e[0][j] = grp ? tx[1][j] - tx[0][j] : tx[0][j] - tx[2][j];
      e[1][j] = grp ? tx[0][j] - tx[2][j] : tx[1][j] + tx[2][j];
    }
    float* v = v_s + stage * Cfg::V_FLOATS + t_dst;
    if constexpr (Cfg::DIAG == 9) {                       // timing experiment (wrong results): the same loads and stores, no arithmetic
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        wf2 o;
        o[0] = tx[r][0]; o[1] = tx[r][1]; *reinterpret_cast<wf2*>(v + (r * 4 + 0) * TB) = o;
        o[0] = tx[r][2]; o[1] = tx[r][3]; *reinterpret_cast<wf2*>(v + (r * 4 + 1) * TB) = o;
        o[0] = tx[r][4]; o[1] = tx[r][5]; *reinterpret_cast<wf2*>(v + (r * 4 + 2) * TB) = o;
        o[0] = tx[2][r]; o[1] = tx[2][r + 2]; *reinterpret_cast<wf2*>(v + (r * 4 + 3) * TB) = o;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      wf2 o;
      o[0] = e[r][0] - e[r][2]; o[1] = e[r][2] - e[r][4]; *reinterpret_cast<wf2*>(v + (r * 4 + 0) * TB) = o;
      o[0] = e[r][1] + e[r][2]; o[1] = e[r][3] + e[r][4]; *reinterpret_cast<wf2*>(v + (r * 4 + 1) * TB) = o;
      o[0] = e[r][2] - e[r][1]; o[1] = e[r][4] - e[r][3]; *reinterpret_cast<wf2*>(v + (r * 4 + 2) * TB) = o;
      o[0] = e[r][1] - e[r][3]; o[1] = e[r][3] - e[r][5]; *reinterpret_cast<wf2*>(v + (r * 4 + 3) * TB) = o;
    }
  };
  auto transform = [&](int stage) { transform_read(stage); transform_finish(stage); };

  f32x16 acc[8];
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
  const int a_off = (half * 16 + grp * 8) * MB + wm * 32 + bl;
  const int b_off = (half * 16 + grp * 8) * TB + wn * 32 + bl;
  auto mfma_chunk = [&](int stage, auto first_c) {      // first_c: the tile's first chunk starts every accumulator from the inline zero
    constexpr bool FIRST = decltype(first_c)::value;
    const float* A = u_s + stage * Cfg::U_FLOATS + a_off;
    const float* B = v_s + stage * Cfg::V_FLOATS + b_off;
    constexpr int NSTEP = (CC / 2) * 8;                  // (channel pair, xi of this group): one MFMA each
    constexpr int PF = 4, RING = PF + 1;
    float av[RING], bv[RING];
    auto read_step = [&](int s) {
      const int cp = s >> 3, x = s & 7;
      av[s % RING] = A[(2 * cp * 16 + x) * MB];
      bv[s % RING] = B[(2 * cp * 16 + x) * TB];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if constexpr (Cfg::DIAG != 12 && Cfg::DIAG != 13) {            // (12 / 13: timing experiments, the MFMAs reuse the first operands)
        if (s + PF < NSTEP) read_step(s + PF);
      }
      acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING], bv[s % RING], FIRST && s < 8 ? zero16 : acc[s & 7], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  auto chunk_barrier = [&]() {                          // own DMAs landed, own V writes done, everybody finished with the old stages
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  unsigned long long t_acc0 = 0, t_acc1 = 0, t_acc2 = 0, t_acc3 = 0, t_acc4 = 0, t_acc5 = 0, t_last = 0, t_begin = 0;
  int n_items_done = 0;
  auto stamp8 = [&](int slot) {                         // DIAG 8: phase totals of one workgroup
    if constexpr (Cfg::DIAG == 8) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      const unsigned long long dcy = now - t_last;
      t_last = now;
      if (slot == 0) t_acc0 += dcy; else if (slot == 1) t_acc1 += dcy; else if (slot == 2) t_acc2 += dcy;
      else if (slot == 3) t_acc3 += dcy; else if (slot == 4) t_acc4 += dcy; else if (slot == 5) t_acc5 += dcy;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int gs = 0;                                          // chunks done so far: the current chunk uses stage gs & 1
  const float* pu;                                     // filters one chunk ahead / raw tile two chunks ahead, inside the current tile
  const float* px;
  int px_left;                                         // channels from px's chunk to Cin
  // One chunk.  Free since the barrier that ended the previous chunk: filter stage sn (held the previous chunk), raw stage sc
  // (holds this chunk's raw tile, transformed during the previous chunk) and V stage sn.  "One chunk ahead" (filters, patch
  // transform) and "two chunks ahead" (raw tile) run on into the next tile when this one ends -- WHERE: 0 = both inside this tile
  // (no conditions in the steady loop), 1 = the tile's second-to-last chunk (raw tile of the next tile's chunk 0), 2 = its last
  // chunk (filters, transform: next tile's chunk 0; raw tile: its chunk 1).
  auto chunk_body = [&](auto first_c, auto where_c) {
    constexpr int WHERE = decltype(where_c)::value;
    const int sc = gs & 1, sn = sc ^ 1;
    const bool ahead = WHERE != 2 || have_next;         // is there a chunk after this one at all
    auto dmas = [&]() {                                 // this thread's NU4 + NRAW pieces
      if constexpr (Cfg::DIAG == 11 || Cfg::DIAG == 13) return;     // timing experiments (wrong results): no LDS-DMA in the chunk loop
      if constexpr (WHERE == 0) {
        dma_u(pu, sn);
        dma_r(px, px_left, vo_r, sc);
      } else if constexpr (WHERE == 1) {
        dma_u(pu, sn);
        if (have_next) dma_r(n_x, Cin, vo_rn, sc);
      } else if (have_next) {
        dma_u(n_u, sn);
        dma_r(n_x + x_step, Cin - CC, vo_rn, sc);
      }
    };
    if (grp == Cfg::SWAP) {                             // transform-first group: patch reads, DMAs, transform, MFMAs;  the other: MFMAs, then the same
      if (ahead) transform_read(sn);
      __builtin_amdgcn_sched_barrier(0);
      dmas();
      __builtin_amdgcn_sched_barrier(0);
      if (ahead) transform_finish(sn);
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(sc, first_c);
    } else {
      mfma_chunk(sc, first_c);
      __builtin_amdgcn_sched_barrier(0);
      if (ahead) transform_read(sn);
      __builtin_amdgcn_sched_barrier(0);
      dmas();
      __builtin_amdgcn_sched_barrier(0);
      if (ahead) transform_finish(sn);
      __builtin_amdgcn_sched_barrier(0);
    }
    pu += u_step; px += x_step; px_left -= CC;
    chunk_barrier();
    ++gs;
  };
  typedef std::integral_constant<int, 0> in_tile_t;
  typedef std::integral_constant<int, 1> second_to_last_t;
  typedef std::integral_constant<int, 2> last_t;

  if constexpr (Cfg::DIAG == 8) { t_begin = __builtin_amdgcn_s_memtime(); t_last = t_begin; }
  // pipeline fill (once per workgroup): filters of chunk 0, raw tiles of chunks 0 and 1, V of chunk 0
  dma_u(c_u, 0);
  dma_r(c_x, Cin, vo_r, 0);
  dma_r(c_x + x_step, Cin - CC, vo_r, 1);
  chunk_barrier();
  transform(0);
  chunk_barrier();
  stamp8(0);
  for (;;) {                                            // one pass per tile
    stamp8(3);
    pu = c_u + u_step; px = c_x + 2 * x_step; px_left = Cin - 2 * CC;
    if (nChunks == 2) {
      chunk_body(std::true_type{}, second_to_last_t{});
    } else {
      chunk_body(std::true_type{}, in_tile_t{});
      for (int k = 1; k < nChunks - 2; ++k) chunk_body(std::false_type{}, in_tile_t{});
      chunk_body(std::false_type{}, second_to_last_t{});
    }
    chunk_body(std::false_type{}, last_t{});
    stamp8(2);
    // the stages the last chunk used (parity of gs - 1) are free; the other filter / V stages already hold the next tile's chunk 0
    const int sf = (gs & 1) ^ 1;
    wino3_writeout<Cfg, true, true>(a, acc, u_s + sf * Cfg::U_FLOATS, v_s + sf * Cfg::V_FLOATS, c_n, c_h0, c_w0, c_m0, c_pt, nPT, [&]() { stamp8(4); });
    stamp8(5);
    ++n_items_done;
    if (!have_next) break;
    c_n = n_n; c_h0 = n_h0; c_w0 = n_w0; c_m0 = n_m0; c_pt = n_pt; c_u = n_u; c_x = n_x;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) vo_r[i] = vo_rn[i];
    walk.next();
    have_next = walk.valid;
    n_n = walk.n; n_h0 = walk.trow * 4; n_w0 = walk.tcol * PW; n_m0 = walk.mb * MB; n_pt = walk.pt;
    n_u = a.u + n_m0;
    n_x = a.src + (size_t)n_n * Cin * HW;
    if (have_next) raw_offsets(vo_rn, n_h0, n_w0);
  }
  if constexpr (Cfg::DIAG == 8) {                       // phase totals of one workgroup, behind the N-th image of dst (the caller allocates N + 1)
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dst + (size_t)a.N * Cout * HW) + wave * 8;
      o[0] = t_acc0; o[1] = t_acc1; o[2] = t_acc2; o[3] = t_acc3; o[4] = t_acc4; o[5] = t_acc5;
      o[6] = (unsigned long long)n_items_done; o[7] = __builtin_amdgcn_s_memtime() - t_begin;
    }
  }
}

}  // namespace tnv3
