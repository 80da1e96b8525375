// conv3x3_wino6_mfma.h -- fourth generation of the fused Winograd F(2x2, 3x3) forward / data-gradient kernel (variant 6 of
// tnv3_conv3x3_wino_forward): the streaming persistent kernel of conv3x3_wino3_mfma.h re-tiled to 128 OUTPUT CHANNELS x 32 tiles
// per workgroup, with the filter operand read straight from L2 into registers.
//
// Why.  The streaming kernel (variant 5) keeps the matrix pipe busy 62 % of the time; its timing twins attribute the rest to the
// patch transform's LDS round trip (16 %), the LDS-DMA issue (6 %) and their interaction (6 %), and the co-issue probe says the
// only lever is the NUMBER of non-MFMA instructions per MFMA (every VALU / LDS / VMEM instruction of either wave of a SIMD costs
// matrix-pipe time, wherever it is placed).  With a 64-channel x 64-tile workgroup tile the B operand V = B^T d B of a chunk
// (8 channels x 16 xi x 64 tiles) feeds 64 output channels; it is re-made by every 64-channel block (Cout = 512: eight times --
// also the 2.4x input over-fetch of the round-2 review).  Here:
//   * tile = 128 channels x 32 tiles (4 x 32 pixels): V of a chunk is half the size and feeds twice the channels -- per MFMA half
//     the raw LDS-DMA, half the patch-transform reads / adds / writes, half the input traffic;
//   * the waves are 4 (channel blocks of 32) x 2 (xi groups): no two waves share an A operand, so staging the filter panel in LDS
//     buys nothing.  Each lane fetches its own A values with eight 16-byte global loads per chunk from a panel packed in exactly
//     that order (layout 2: [32-channel block][chunk][xi group][q][lane][4], 1 KB per wave-load, L2-resident: every workgroup of a
//     channel block reads the same 16 KB per chunk), IN PLACE: the load of chunk k+1's q-th quad is issued right behind the four
//     MFMAs that consumed chunk k's -- 32 registers, no LDS stage, no ds_read for A, no M0 traffic;
//   * per wave and chunk (32 MFMAs): 32 ds_read_b32 (B) + 6 LDS reads / 14 adds / 4 ds_write_b64 (transform) + 1 LDS-DMA piece +
//     8 global loads -- 42 LDS + ~14 VALU + 9 VMEM against 81 LDS + ~45 VALU + 6 VMEM in variant 5;
//   * the patch transform is split by TRANSFORM ROW: a wave produces one row R of B^T d B (two raw rows in, four xi out, for a pair
//     of horizontally adjacent tiles per thread), so R -- and with it the sign pattern and the raw-row offsets -- is wave-uniform;
//     the lane -> (channel, tile row, tile pair) map is chosen so that every ds_read_b128 group of 16 lanes covers 64 distinct banks.
// Same arithmetic per element in the same order as variants 2-5 (products and K order per accumulator, patch transform, output
// transform) => BIT-IDENTICAL results (tested on the emulator and the GPU).  LDS: 2 V + 2 raw stages + a dedicated 64 KB exchange
// region for the write-out = 113 KB.  Needs Cout % 128 == 0, W % 32 == 0, H % 4 == 0, Cin > 8.
#pragma once
#include <type_traits>
#include "conv3x3_wino3_mfma.h"

namespace tnv3 {

constexpr int kWinoCinPadK = 24;     // = kWinoCinPad (tnv3_impl.h): packed filter rows are padded to a multiple of 24 input channels

template <int DIAG_ = 0, int SWAP_ = 0, int PRIO_ = 0, int QB_ = 0, int ALATE_ = 0, int WN_ = 1>
struct WinoV6Cfg {
  // WN = 1: 128 channels x 32 tiles (4 x 32 pixels) per workgroup -- variant 6.  WN = 2: 64 channels x 64 tiles (4 x 64 pixels) -- variant 7,
  // the same kernel for layers with 64 output channels (the four 288 x 512 layers): the two waves of a channel block each load their own
  // copy of the A quads (L1 hits), V of a chunk feeds 64 channels as in variant 5, but there is no filter stage in LDS, no ds_read for A,
  // and the patch transform is the row-split, conflict-free one (two tile pairs per thread).
  // 1: "quad" V layout V[c][transform row R][tile][4] (the four xi of a row adjacent): ONE conflict-free ds_read_b128 feeds the B
  // operand of four MFMAs (8 instead of 32 LDS reads inside a wave's MFMA phase) and the transform stores a tile's row as one
  // ds_write_b128.  Every accumulator still sees its channel pairs in the same order => the same bits.
  static constexpr int QB = QB_;
  // 1: the eight A loads of the next chunk are issued in one burst right BEHIND the wave's MFMA phase instead of inside it.
  static constexpr int ALATE = ALATE_;
  // Which wave group runs its MFMAs FIRST in a chunk (the other one starts with the patch transform): 0 = group 1 (waves 4-7),
  // 1 = group 0 (waves 0-3, the OLDER waves of a SIMD, which the matrix pipe's arbitration favours when two MFMA streams compete).
  static constexpr int SWAP = SWAP_;
  static constexpr int PRIO = PRIO_;                 // > 0: the MFMA-first group raises its priority (s_setprio) for its MFMA phase
  static constexpr int DIAG = DIAG_;                 // timing twins (WRONG results; libtnv3_diag.so): 10 no patch transform, 11 no raw DMA,
                                                     // 14 no A loads in the chunk loop, 13 none of the three
  static constexpr int WN = WN_, WM = 4 / WN_, CC = 8;
  static_assert(WN_ == 1 || WN_ == 2, "tile shapes");
  static constexpr int NT = 2 * WM * WN * 64;        // 512 threads: waves 0-3 = xi group 0 (channel block wq / WN, tile half wq % WN), waves 4-7 = xi group 1
  static constexpr int MB = 32 * WM, TB = 32 * WN, PW = 32 * WN;
  static constexpr int RW = PW + 8, RAWP = 6 * RW;   // raw halo tile per channel: 6 rows x (PW + 8) floats (columns w0-4 .. w0+PW+3)
  static constexpr int RAW_FLOATS = CC * RAWP;       // WN = 1: 480 16-byte pieces (one per thread, 32 idle); WN = 2: 864 (two per thread)
  static constexpr int NRAW = WN;
  static constexpr int RAW_STAGE = NRAW * NT * 4;
  static constexpr int VC = 16 * TB + (WN == 1 ? 16 : 0);   // channel stride of V (WN = 1: padded for the 8-byte stores of the non-quad form; WN = 2: LDS is full)
  static constexpr int V_FLOATS = CC * VC;
  static constexpr int XCH_FLOATS = (NT / 64) * 32 * 64;
  static constexpr int LDS_FLOATS = 2 * V_FLOATS + 2 * RAW_STAGE + XCH_FLOATS;
  static constexpr int A_CHUNK_FLOATS = 2 * 8 * 64 * 4;      // one (32-channel block, chunk) of the layout-2 panel: [xi group][q][lane][4]
  static_assert(RAW_FLOATS / 4 <= NRAW * NT, "every raw piece has a slot");
  static_assert(CC * 4 * 2 * (TB / 4) == NT * WN, "WN (channel, transform row, tile row, tile pair) tasks per thread and chunk");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
  static_assert(WN == 1 || QB_ == 1, "the 64-channel form is built on the quad V layout");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) conv3x3_wino_a128_stream_kernel(const WinoArgs a) {
  constexpr int CC = Cfg::CC, NT = Cfg::NT, MB = Cfg::MB, TB = Cfg::TB, PW = Cfg::PW, RW = Cfg::RW, RAWP = Cfg::RAWP, VC = Cfg::VC;
  __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
  float* v_s = lds;                                   // two stages each
  float* raw_s = lds + 2 * Cfg::V_FLOATS;
  float* xch_s = raw_s + 2 * Cfg::RAW_STAGE;          // write-out exchange: group 0's 32 KB, then group 1's

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wq = __builtin_amdgcn_readfirstlane(wave & 3);
  // The two groups' programs are selected on this PER-LANE copy of the group: the compiler then predicates both programs in one
  // straight line (the inactive one is skipped through EXEC) instead of building a scalar diamond -- whose phi nodes made it copy
  // and spill the 128 accumulators and the 32 filter registers between the two MFMA sites (700+ spills).
  const int grp_v = wave >> 2;
  const int half = lane >> 5, bl = lane & 31;
  constexpr int WN = Cfg::WN, NRAW = Cfg::NRAW;
  const int wm = wq / WN, wn = wq % WN;               // MFMA roles: 32-channel block, 32-tile half (WN = 2: tile row) of the workgroup tile
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
  const int tilesH = H / 4, tilesW = W / PW;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  const int nChunks = (Cin + CC - 1) / CC;             // >= 2 (host)
  const int nChunksPad = (Cin + kWinoCinPadK - 1) / kWinoCinPadK * (kWinoCinPadK / CC);      // chunks per 32-channel block in the packed panel

  ConvTileWalk walk;                                  // always one tile ahead of the one being computed
  walk.init(blockIdx.x, gridDim.x, nMB, nPT, tilesH, tilesW);
  if (!walk.valid) return;

  // ---- raw LDS-DMA: piece e = tid of [CC][6][RW/4]; padding pieces and the 32 spare threads read out of range (= 0)
  unsigned vo_r[NRAW], vo_rn[NRAW];
  auto raw_offset = [&](unsigned (&vo)[NRAW], int h0, int w0) {
    int t_op = tid;
    TNV3_OPAQUE_V(t_op);                              // recomputed per tile; nothing of it stays live across the chunk loop
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      const int e = t_op + i * NT;
      const int c = e / (RAWP / 4), r = e - c * (RAWP / 4);
      const int tr = r / (RW / 4), q = r - tr * (RW / 4);
      const int gh = h0 - 1 + tr, gw = w0 - 4 + 4 * q;
      const bool ok = e < Cfg::RAW_FLOATS / 4 && gh >= 0 && gh < H && gw >= 0 && gw < W;
      vo[i] = ok ? (unsigned)(c * HW + gh * W + gw) * 4u : kDmaOob;
    }
  };
  int c_n = walk.n, c_h0 = walk.trow * 4, c_w0 = walk.tcol * PW, c_m0 = walk.mb * MB, c_pt = walk.pt;
  raw_offset(vo_r, c_h0, c_w0);
  walk.next();
  bool have_next = walk.valid;
  int n_n = walk.n, n_h0 = walk.trow * 4, n_w0 = walk.tcol * PW, n_m0 = walk.mb * MB, n_pt = walk.pt;
#pragma unroll
  for (int i = 0; i < NRAW; ++i) vo_rn[i] = kDmaOob;
  if (have_next) raw_offset(vo_rn, n_h0, n_w0);

  const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);      // scalar: the LDS-DMA destination (M0) stays on the SALU
  const size_t x_step = (size_t)CC * HW;                            // floats per chunk of the input
  auto dma_r = [&](const float* xp, int cvalid, const unsigned (&vo)[NRAW], int sr) {
    if constexpr (Cfg::DIAG == 11 || Cfg::DIAG == 13) return;
    const tnv3_rsrc_t rr = tnv3_make_rsrc(xp, (unsigned)(cvalid < CC ? cvalid : CC) * (unsigned)HW * 4u);   // channels past Cin: beyond num_records, zero
#pragma unroll
    for (int i = 0; i < NRAW; ++i) tnv3_buf_dma16(rr, raw_s + sr * Cfg::RAW_STAGE + (i * NT + wbase) * 4, vo[i]);
  };
  // ---- A operand: this wave's slice of the layout-2 panel, chunk by chunk: [q = channel pair * 2 + xi quad][lane][4]
  const size_t a_step = Cfg::A_CHUNK_FLOATS;                         // floats from one chunk of a 32-channel block to the next
  auto a_base = [&](int m0) -> const float* {
    return a.u + ((size_t)(m0 / 32 + wm) * nChunksPad) * Cfg::A_CHUNK_FLOATS + grp * (8 * 64 * 4);
  };
  const float* c_a = a_base(c_m0);
  const float* n_a = a_base(n_m0);
  const float* c_x = a.src + (size_t)c_n * Cin * HW;
  const float* n_x = a.src + (size_t)n_n * Cin * HW;
  f32x4 av[8];
  // through a buffer descriptor on the (scalar) slice base: the per-lane part of the address is ONE 32-bit register (lane * 16) and the
  // quad index an immediate -- 64-bit per-lane pointers cost three register pairs and six VALU additions per chunk
  const unsigned a_lane_b = (unsigned)lane * 16u;
  auto load_a = [&](const float* p, int q) {
    const tnv3_rsrc_t ra = tnv3_make_rsrc(p, 8u * 64u * 16u);
    av[q] = tnv3_buf_load_f4(ra, a_lane_b, (unsigned)q * 1024u);
  };

  // ---- patch transform: wave -> transform row R = 2 grp + (wq & 1) and channel half cs = wq >> 1; lane -> (channel, tile row,
  //      tile pair).  Lanes 4r .. 4r+3 are one run of four consecutive tile pairs; ds_read_b128 serves the lane groups
  //      {runs 0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}: the runs of one group take the four channels of one (tile row,
  //      pair half), whose raw tiles lie 240 floats apart -- 16-bank ranges 0, 48, 32, 16: no conflicts.
  const int run = lane >> 2, r7 = run & 7;
  const int t_tr = run >> 3, t_c = (wq >> 1) * 4 + (r7 >> 1);
  const int t_pj = 4 * (__builtin_popcount(r7) & 1) + (lane & 3);
  const int tR = 2 * grp + (wq & 1);                                    // wave-uniform
  const int rowA = tR == 0 ? 0 : (tR == 2 ? 2 : 1), rowB = tR == 0 ? 2 : (tR == 1 ? 2 : (tR == 2 ? 1 : 3));     // e = d[rowA] -/+ d[rowB]
  const int t_srcA = t_c * RAWP + (2 * t_tr + rowA) * RW + 4 * t_pj;     // 16-byte aligned
  const int t_srcB = t_c * RAWP + (2 * t_tr + rowB) * RW + 4 * t_pj;
  const int t_dst = Cfg::QB ? t_c * VC + tR * (TB * 4) + (t_tr * (TB / 2) + 2 * t_pj) * 4       // [c][R][tile][4]
                            : t_c * VC + (tR * 4) * TB + t_tr * (TB / 2) + 2 * t_pj;
  typedef float wf2 __attribute__((ext_vector_type(2)));
  // WN tasks per thread: tile pairs t_pj and (WN = 2) t_pj + 8 of the same (channel, tile row) -- 32 floats further along the raw rows
  // (the second task is read and finished after the first one's stores: twelve live patch registers, not twenty-four)
  float txa[1][6], txb[1][6];
  auto transform_read_k = [&](int stage, int kk) {      // patch columns 4pj+3 .. 4pj+8 of the two raw rows this transform row needs
    if constexpr (Cfg::DIAG == 10 || Cfg::DIAG == 13) return;
    const float* d = raw_s + stage * Cfg::RAW_STAGE + 32 * kk;
#pragma unroll
    for (int k = 0; k < 1; ++k) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(d + t_srcA + 32 * k), a1 = *reinterpret_cast<const f32x4*>(d + t_srcA + 32 * k + 4);
      const wf2 a2 = *reinterpret_cast<const wf2*>(d + t_srcA + 32 * k + 8);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(d + t_srcB + 32 * k), b1 = *reinterpret_cast<const f32x4*>(d + t_srcB + 32 * k + 4);
      const wf2 b2 = *reinterpret_cast<const wf2*>(d + t_srcB + 32 * k + 8);
      txa[k][0] = a0[3]; txa[k][1] = a1[0]; txa[k][2] = a1[1]; txa[k][3] = a1[2]; txa[k][4] = a1[3]; txa[k][5] = a2[0];
      txb[k][0] = b0[3]; txb[k][1] = b1[0]; txb[k][2] = b1[1]; txb[k][3] = b1[2]; txb[k][4] = b1[3]; txb[k][5] = b2[0];
    }
  };
  auto transform_finish_k = [&](int stage, int kk) {    // raw stage -> V stage of the same parity
    if constexpr (Cfg::DIAG == 10 || Cfg::DIAG == 13) return;
#pragma unroll
    for (int k = 0; k < 1; ++k) {
      float e[6];
      if (tR == 1) {                                      // (B^T d) row 1 = d1 + d2; rows 0, 2, 3 = d0 - d2, d2 - d1, d1 - d3
#pragma unroll
        for (int j = 0; j < 6; ++j) e[j] = txa[k][j] + txb[k][j];
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) e[j] = txa[k][j] - txb[k][j];
      }
      if constexpr (Cfg::QB) {                            // the four xi of this row, tile 2pj then tile 2pj + 1: two 16-byte stores
        float* v = v_s + stage * Cfg::V_FLOATS + t_dst + 64 * kk;     // 16 tiles x 4 floats further
        f32x4 o4;
        o4[0] = e[0] - e[2]; o4[1] = e[1] + e[2]; o4[2] = e[2] - e[1]; o4[3] = e[1] - e[3];
        *reinterpret_cast<f32x4*>(v) = o4;
        o4[0] = e[2] - e[4]; o4[1] = e[3] + e[4]; o4[2] = e[4] - e[3]; o4[3] = e[3] - e[5];
        *reinterpret_cast<f32x4*>(v + 4) = o4;
      } else {
        float* v = v_s + stage * Cfg::V_FLOATS + t_dst + 16 * kk;
        wf2 o;
        o[0] = e[0] - e[2]; o[1] = e[2] - e[4]; *reinterpret_cast<wf2*>(v + 0 * TB) = o;
        o[0] = e[1] + e[2]; o[1] = e[3] + e[4]; *reinterpret_cast<wf2*>(v + 1 * TB) = o;
        o[0] = e[2] - e[1]; o[1] = e[4] - e[3]; *reinterpret_cast<wf2*>(v + 2 * TB) = o;
        o[0] = e[1] - e[3]; o[1] = e[3] - e[5]; *reinterpret_cast<wf2*>(v + 3 * TB) = o;
      }
    }
  };

  auto transform_read = [&](int stage) { transform_read_k(stage, 0); };
  auto transform_finish = [&](int stage) {              // finishes task 0, then (WN = 2) reads and finishes task 1
    transform_finish_k(stage, 0);
    if constexpr (WN == 2) {
      transform_read_k(stage, 1);
      transform_finish_k(stage, 1);
    }
  };
  f32x16 acc[8];
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
  const int b_off = Cfg::QB ? half * VC + (2 * grp) * (TB * 4) + (wn * 32 + bl) * 4 : half * VC + (grp * 8) * TB + wn * 32 + bl;
  // One chunk of MFMAs: step s = (channel pair cp, xi x of this group); A from registers, B from the V stage four steps ahead.
  // Right behind the four MFMAs that consumed a quad, its registers are re-loaded with the NEXT chunk's quad from `anext` (always a
  // valid address: the last chunk of the last tile re-reads its own -- unconditional loads keep the MFMA stream one basic block).
  auto mfma_chunk = [&](int stage, auto first_c, const float* anext) {
    constexpr bool FIRST = decltype(first_c)::value;
    const float* B = v_s + stage * Cfg::V_FLOATS + b_off;
    constexpr int NSTEP = (CC / 2) * 8;
    if constexpr (Cfg::QB) {                              // step s = (cp, x): B quad (cp, x >> 2) = one 16-byte read, two quads ahead
      constexpr int BR = WN == 2 ? 2 : 3;                 // quads in flight (the 64-channel form has no registers to spare: one quad = four MFMAs ahead)
      f32x4 bq[BR];
      auto read_quad = [&](int qd) { bq[qd % BR] = *reinterpret_cast<const f32x4*>(B + (2 * (qd >> 1)) * VC + (qd & 1) * (TB * 4)); };
      read_quad(0);
      if constexpr (BR == 3) read_quad(1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        // full fences: the read of quad q + 2 goes out BEFORE the four MFMAs of quad q (with class-only group barriers the scheduler
        // satisfied "one LDS read per group" with the read needed NEXT and the ring collapsed to read -> wait -> use)
        if (q + BR - 1 < 8) read_quad(q + BR - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int s = 4 * q + j;
          acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][j], bq[q % BR][j], FIRST && s < 8 ? zero16 : acc[s & 7], 0, 0, 0);
        }
        if constexpr (Cfg::DIAG != 14 && Cfg::DIAG != 13 && !Cfg::ALATE) load_a(anext, q);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (Cfg::DIAG != 14 && Cfg::DIAG != 13 && Cfg::ALATE) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) load_a(anext, q);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    constexpr int PF = 4, RING = PF + 1;
    float bv[RING];
    auto read_step = [&](int s) {
      const int cp = s >> 3, x = s & 7;
      bv[s % RING] = B[(2 * cp) * VC + x * TB];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_step(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + PF < NSTEP) read_step(s + PF);
      const int q = (s >> 3) * 2 + ((s & 7) >> 2);
      acc[s & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][s & 3], bv[s % RING], FIRST && s < 8 ? zero16 : acc[s & 7], 0, 0, 0);
      // pin the issue order (one B read, one MFMA, and behind a quad's last MFMA its re-load): left alone the scheduler hoists the
      // eight loads to the top of the block, where the old quads are still live -- 32 extra registers, i.e. spills
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if ((s & 3) == 3) {
        if constexpr (Cfg::DIAG != 14 && Cfg::DIAG != 13 && !Cfg::ALATE) {
          load_a(anext, q);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
    if constexpr (Cfg::DIAG != 14 && Cfg::DIAG != 13 && Cfg::ALATE) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; ++q) load_a(anext, q);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // End of a chunk: this wave's raw piece has landed (it is OLDER than the A loads the wave issued inside its MFMA stream when the
  // wave is of group 0 -- patch reads, DMA, transform, MFMAs -- so eight loads may stay in flight; group 1 issues its DMA last),
  // its V writes are done, and everybody has finished with the old stages.
  auto chunk_barrier = [&]() {
    if (grp == Cfg::SWAP) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(8));
    else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };
  auto full_barrier = [&]() {
    __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
    __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
    __builtin_amdgcn_s_barrier();
  };

  // DIAG 7: s_memtime phase totals of one mid-grid workgroup (results stay correct; totals behind the N-th image of dst)
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
  auto stamp = [&](int slot) {
    if constexpr (Cfg::DIAG == 7) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      t_acc[slot] += now - t_last;
      t_last = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int gs = 0;                                          // chunks done so far: the current chunk uses stage gs & 1
  const float* pa;                                     // A one chunk ahead / raw tile two chunks ahead, inside the current tile
  const float* px;
  int px_left;
  // One chunk (see conv3x3_wino_stream_mfma_kernel): WHERE 0 = "one ahead" and "two ahead" both inside this tile, 1 = the tile's
  // second-to-last chunk (raw tile of the next tile's chunk 0), 2 = its last chunk (A and transform: next tile's chunk 0; raw tile:
  // its chunk 1).
  auto chunk_body = [&](auto first_c, auto where_c) {
    constexpr int WHERE = decltype(where_c)::value;
    const int sc = gs & 1, sn = sc ^ 1;
    const bool ahead = WHERE != 2 || have_next;
    const float* anext = WHERE == 2 ? (have_next ? n_a : c_a) : pa;
    auto dma = [&]() {
      if constexpr (WHERE == 0) dma_r(px, px_left, vo_r, sc);
      else if constexpr (WHERE == 1) { if (have_next) dma_r(n_x, Cin, vo_rn, sc); }
      else if (have_next) dma_r(n_x + x_step, Cin - CC, vo_rn, sc);
    };
    stamp(0);
    if (grp_v == Cfg::SWAP) {                           // transform-first group: patch reads, DMA, transform, MFMAs;  the other: MFMAs, then the same
      if (ahead) transform_read(sn);
      __builtin_amdgcn_sched_barrier(0);
      dma();
      __builtin_amdgcn_sched_barrier(0);
      stamp(1);
      if (ahead) transform_finish(sn);
      __builtin_amdgcn_sched_barrier(0);
      stamp(2);
      mfma_chunk(sc, first_c, anext);
      stamp(3);
    } else {
      if constexpr (Cfg::PRIO > 0) __builtin_amdgcn_s_setprio(Cfg::PRIO);
      mfma_chunk(sc, first_c, anext);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Cfg::PRIO > 0) __builtin_amdgcn_s_setprio(0);
      stamp(1);
      if (ahead) transform_read(sn);
      __builtin_amdgcn_sched_barrier(0);
      dma();
      __builtin_amdgcn_sched_barrier(0);
      stamp(2);
      if (ahead) transform_finish(sn);
      __builtin_amdgcn_sched_barrier(0);
      stamp(3);
    }
    pa += a_step; px += x_step; px_left -= CC;
    if constexpr (Cfg::DIAG == 7) {
      if (grp == Cfg::SWAP) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(8));
      else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
      __builtin_amdgcn_s_waitcnt(tnv3_lgkmcnt_only(0));
      stamp(4);
      __builtin_amdgcn_s_barrier();
      stamp(5);
    } else {
      chunk_barrier();
    }
    ++gs;
  };
  typedef std::integral_constant<int, 0> in_tile_t;
  typedef std::integral_constant<int, 1> second_to_last_t;
  typedef std::integral_constant<int, 2> last_t;

  // pipeline fill (once per workgroup): A of chunk 0, raw tiles of chunks 0 and 1, V of chunk 0
#pragma unroll
  for (int q = 0; q < 8; ++q) load_a(c_a, q);
  dma_r(c_x, Cin, vo_r, 0);
  dma_r(c_x + x_step, Cin - CC, vo_r, 1);
  full_barrier();
  transform_read(0);
  transform_finish(0);
  full_barrier();
  if constexpr (Cfg::DIAG == 7) t_last = __builtin_amdgcn_s_memtime();
  int n_tiles_done = 0;
  for (;;) {                                            // one pass per tile
    pa = c_a + a_step; px = c_x + 2 * x_step; px_left = Cin - 2 * CC;
    if (nChunks == 2) {
      chunk_body(std::true_type{}, second_to_last_t{});
    } else {
      chunk_body(std::true_type{}, in_tile_t{});
      for (int k = 1; k < nChunks - 2; ++k) chunk_body(std::false_type{}, in_tile_t{});
      chunk_body(std::false_type{}, second_to_last_t{});
    }
    chunk_body(std::false_type{}, last_t{});
    wino3_writeout<Cfg, true, true>(a, acc, xch_s, xch_s + 4 * 32 * 64, c_n, c_h0, c_w0, c_m0, c_pt, nPT, []() {});
    stamp(6);
    ++n_tiles_done;
    if (!have_next) break;
    c_n = n_n; c_h0 = n_h0; c_w0 = n_w0; c_m0 = n_m0; c_pt = n_pt; c_a = n_a; c_x = n_x;
#pragma unroll
    for (int i = 0; i < NRAW; ++i) vo_r[i] = vo_rn[i];
    walk.next();
    have_next = walk.valid;
    n_n = walk.n; n_h0 = walk.trow * 4; n_w0 = walk.tcol * PW; n_m0 = walk.mb * MB; n_pt = walk.pt;
    n_a = a_base(n_m0);
    n_x = a.src + (size_t)n_n * Cin * HW;
    if (have_next) raw_offset(vo_rn, n_h0, n_w0);
    stamp(7);
  }
  if constexpr (Cfg::DIAG == 7) {                       // [wave][10]: six chunk phases, write-out, tile advance, chunks, tiles
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dst + (size_t)a.N * Cout * HW) + wave * 10;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = t_acc[i];
      o[8] = (unsigned long long)gs; o[9] = (unsigned long long)n_tiles_done;
    }
  }
}

}  // namespace tnv3
