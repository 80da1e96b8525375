// inpaint_fused.h -- InpaintNet.forward (model.py:113-129) as ONE persistent kernel: a workgroup carries a sequence through
// all nine Conv1d layers with every activation (the three skip tensors included) resident in LDS, so the network is one
// launch instead of nine and no activation ever touches HBM (SURVEY 2.1, row "InpaintNet as one fused kernel").
//
//   * The seven dense layers (32->64 ... 384->128 ... 96->32, 99.9 % of the FLOPs) run on v_mfma_f32_16x16x4_f32: M = 16 output
//     channels, N = the 16 positions of the sequence, K = 4 (input channel, tap) pairs.  'same' padding needs no masking: every
//     activation row is stored with a zero halo column on each side ([C][20] floats: columns 0 and 17 zero, data in 1..16;
//     20 keeps the four K-groups of an operand read on disjoint banks).
//   * Filters are re-packed once per weight version (tnv3_inpaintnet_pack) into the order the lanes consume them -- K tap-major
//     (k = tap * Cin + ci) so that an MFMA's four K indices are four consecutive channels at ONE tap, i.e. four immediate
//     offsets of one LDS base -- and stream from L2 straight into registers (one coalesced 1 KiB global_load_dwordx4 per
//     wave and four MFMAs), prefetched a ring ahead; they never pass through LDS.  Concats are two source rows, never built.
//   * The 3->32 stem (K = 9) and the 32->2 head + sigmoid (+ the output permute) are vector code: 0.1 % of the work.
//   * A wave owns Cout/64 channel blocks of a layer and works on all of them at once (shared activation reads, independent
//     accumulators); with one block it splits K over two accumulators so that no MFMA waits for its predecessor.
// One sequence costs 8112 MFMAs = 65 k matrix-pipe cycles on the CU's four SIMDs (27 us): that is the latency floor of a
// batch of <= 256 sequences (one per CU), against nine dependent launches before.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "conv3x3_mfma.h"

namespace tnv3 {

typedef float if_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIfRow = 20;                       // floats per activation row in LDS: halo + 16 positions + halo + 2 pad
constexpr int kIfL = 16;                         // sequence length the fused kernel is built for

// dense layers in forward order: (C0 = channels of the first source, C1 = of the concatenated second one, Cout)
struct IfLayer { int c0, c1, cout; };
constexpr IfLayer kIfLayers[7] = {{32, 0, 64}, {64, 0, 128}, {128, 0, 256}, {256, 0, 256}, {256, 128, 128}, {128, 64, 64}, {64, 32, 32}};
constexpr int if_layer_floats(int i) { return (kIfLayers[i].c0 + kIfLayers[i].c1) * 3 * kIfLayers[i].cout; }
constexpr int if_layer_offset(int i) { return i == 0 ? 0 : if_layer_offset(i - 1) + if_layer_floats(i - 1); }
// packed buffer: [7 dense filters in lane order][stem filter 32x3x3 as stored][head filter 2x32x3 as stored][9 biases]
constexpr int kIfStemOff = if_layer_offset(6) + if_layer_floats(6);
constexpr int kIfHeadOff = kIfStemOff + 32 * 3 * 3;
constexpr int kIfBiasOff = kIfHeadOff + 2 * 32 * 3;
constexpr int kIfBiasCount = 32 + 64 + 128 + 256 + 256 + 128 + 64 + 32 + 2;
constexpr int kIfPackedFloats = kIfBiasOff + kIfBiasCount;
// bias offsets in network order: down_1, down_2, down_3, buttleneck.conv_1, .conv_2, up_1, up_2, up_3, predictor
constexpr int kIfBiasAt[9] = {0, 32, 96, 224, 480, 736, 864, 928, 960};

struct InpaintPackArgs {
  const float* w[9];          // the nine nn.Conv1d weights [Cout][Cin][3] in network order (state_dict tensors)
  const float* b[9];
  float* packed;
};

// Wp[block][chunk][lane][j] = W[co = 16*block + (lane & 15)][ci = 16*(chunk % (Cin/16)) + 4*(lane >> 4) + j][tap = chunk / (Cin/16)]
inline __global__ void __launch_bounds__(256) inpaint_pack_kernel(const InpaintPackArgs a) {
  const int stride = gridDim.x * blockDim.x;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kIfPackedFloats; e += stride) {
    float v;
    if (e < kIfStemOff) {
      int li = 0, cin = kIfLayers[0].c0 + kIfLayers[0].c1, base = 0;               // (constant indices only: the tables live in host memory)
#pragma unroll
      for (int i = 1; i < 7; ++i)
        if (e >= if_layer_offset(i)) { li = i; cin = kIfLayers[i].c0 + kIfLayers[i].c1; base = if_layer_offset(i); }
      const int nch = 3 * cin / 16;
      const int r = e - base;
      const int j = r & 3, lane = (r >> 2) & 63, q = (r >> 8) % nch, blk = (r >> 8) / nch;
      const int tap = q / (cin / 16), ci = (q % (cin / 16)) * 16 + 4 * (lane >> 4) + j, co = blk * 16 + (lane & 15);
      v = a.w[li + 1][((size_t)co * cin + ci) * 3 + tap];
    } else if (e < kIfHeadOff) {
      v = a.w[0][e - kIfStemOff];
    } else if (e < kIfBiasOff) {
      v = a.w[8][e - kIfHeadOff];
    } else {
      const int r = e - kIfBiasOff;
      int li = 0, base = 0;
#pragma unroll
      for (int i = 1; i < 9; ++i)
        if (r >= kIfBiasAt[i]) { li = i; base = kIfBiasAt[i]; }
      v = a.b[li][r - base];
    }
    a.packed[e] = v;
  }
}

// One dense layer for the workgroup's sequence.  src0 / src1: LDS activation rows of the two concatenated sources; every output
// element (channel co, position n) is handed to epi(co, n, value).  BPW = channel blocks a wave works on at once (ceil(Cout / 64)).
template <int C0, int C1, int COUT, int NW = 4, class Epi>
__device__ __forceinline__ void if_dense_core(const float* __restrict__ wp, const float* src0, const float* src1, int wave, int lane_in, Epi&& epi) {
  int lane = lane_in;
  TNV3_OPAQUE_V(lane);        // every per-lane address of this layer is formed HERE, not hoisted to the top of the kernel and kept live
                              // through all nine layers (the nine layers' filter / operand / epilogue addresses cost 60+ registers)
  constexpr int CIN = C0 + C1, NCH = 3 * CIN / 16, NB = COUT / 16, CPT = CIN / 16;     // chunks of 16 K; chunks per tap
  constexpr int BPW = (NB + NW - 1) / NW;                                             // blocks per wave (NW = 4, NB = 6: three waves x 2; NB = 2: two waves x 1)
  static_assert(NB % BPW == 0, "the waves' block groups must tile the layer");
  constexpr int NACC = BPW == 1 ? 2 : BPW;                                            // BPW == 1: split K over two accumulators
  constexpr int RING = BPW >= 4 ? 2 : 3;                                              // filter chunks in flight per block
  static_assert(NCH % RING == 0 && C0 % 16 == 0 && C1 % 16 == 0, "layer shape");
  if (wave * BPW >= NB) return;                                                       // 96 -> 32: two blocks, waves 2 and 3 rest
  const int n = lane & 15, kq = lane >> 4;
  const if_f32x4* wq = reinterpret_cast<const if_f32x4*>(wp) + (size_t)(wave * BPW) * NCH * 64 + lane;
  if_f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = if_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if_f32x4 ring[RING][BPW];
  auto fetch = [&](int q, int slot) {
#pragma unroll
    for (int b = 0; b < BPW; ++b) ring[slot][b] = wq[((size_t)b * NCH + q) * 64];
  };
#pragma unroll
  for (int s = 0; s < RING - 1; ++s) fetch(s, s);
  const int boff = (4 * kq) * kIfRow + n;                                             // lane part of the operand address
  // chunk q = (tap, 16 channels starting at c): walked incrementally (wave-uniform scalars), sources switch at c == C0
  int tap = 0, c = 0;
  auto operands = [&](float (&bv)[4]) {
    const float* src = (c < C0 ? src0 + c * kIfRow : src1 + (c - C0) * kIfRow) + boff + tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = src[j * kIfRow];
    c += 16;
    if (c == CIN) { c = 0; ++tap; }
  };
  float bcur[4], bnext[4];
  operands(bcur);
#pragma unroll 1
  for (int q0 = 0; q0 < NCH; q0 += RING) {                                            // (kept rolled: the ring makes RING steps one period)
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      const int q = q0 + s;
      if (q + RING - 1 < NCH) fetch(q + RING - 1, (s + RING - 1) % RING);             // filters RING - 1 chunks ahead (L2 latency)
      if (q + 1 < NCH) operands(bnext);                                               // activations one chunk ahead (LDS latency)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          const int ai = BPW == 1 ? (j & 1) : b;
          acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s][b][j], bcur[j], acc[ai], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) bcur[j] = bnext[j];
    }
  }
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const if_f32x4 r4 = BPW == 1 ? if_f32x4{acc[0][0] + acc[1][0], acc[0][1] + acc[1][1], acc[0][2] + acc[1][2], acc[0][3] + acc[1][3]} : acc[b];
#pragma unroll
    for (int r = 0; r < 4; ++r) epi((wave * BPW + b) * 16 + 4 * kq + r, n, r4[r]);     // D layout: row = 4 * (lane >> 4) + r, column = lane & 15
  }
}

// Forward layer: + bias, LeakyReLU(0.01) (model.py:81), rows of the output in LDS; gsave != nullptr: the activation also goes to
// HBM as [COUT][16] (saved for the backward pass of a training step).
template <int C0, int C1, int COUT, int NW = 4>
__device__ __forceinline__ void if_dense_layer(const float* __restrict__ wp, const float* __restrict__ bias, const float* src0,
                                               const float* src1, float* dst, int wave, int lane, float* gsave = nullptr) {
  if_dense_core<C0, C1, COUT, NW>(wp, src0, src1, wave, lane, [&](int co, int n, float acc) {
    float v = acc + bias[co];
    v = v > 0.0f ? v : 0.01f * v;
    dst[co * kIfRow + 1 + n] = v;
    if (gsave) gsave[co * kIfL + n] = v;
  });
}

constexpr int kIfRows = 4 + 32 + 64 + 128 + 256 + 256;                               // input(3, padded to 4), x1, x2, x3, ping, pong
constexpr int kIfLdsFloats = kIfRows * kIfRow;

// x [N][16][2], m [N][16][1] -> out [N][16][2]; grid-stride over sequences (one sequence per workgroup at a time)
// acts != nullptr (training forward): the eight hidden activations of every sequence are also written to HBM,
// acts[seq][kItActCh][16] in the order x1, x2, x3, b1, b2, u1, u2, u3 (inpaint_fused_train.h reads them in the backward pass).
constexpr int kItActCh = 32 + 64 + 128 + 256 + 256 + 128 + 64 + 32;                 // 960
constexpr int kItActOff[8] = {0, 32, 96, 224, 480, 736, 864, 928};
// NW = waves per workgroup: 4 (two workgroups per CU: the throughput form) or 8 (one sequence on two waves per SIMD: the nine
// dependent layers' MFMA / LDS latencies overlap between the two waves -- the small-batch latency form, chosen by the host when the
// batch leaves CUs free anyway).
template <int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) inpaintnet_fused_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                                                 const float* __restrict__ packed, float* __restrict__ out, int N,
                                                                                 float* __restrict__ acts) {
  constexpr int NT = 64 * NW;
  __shared__ __attribute__((aligned(16))) float lds[kIfLdsFloats];
  float* in0 = lds;
  float* x1 = in0 + 4 * kIfRow;
  float* x2 = x1 + 32 * kIfRow;
  float* x3 = x2 + 64 * kIfRow;
  float* pa = x3 + 128 * kIfRow;
  float* pb = pa + 256 * kIfRow;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* bias = packed + kIfBiasOff;
  for (int i = tid; i < kIfLdsFloats; i += NT) lds[i] = 0.0f;                      // the halo columns stay zero for good
  for (int seq = blockIdx.x; seq < N; seq += gridDim.x) {
    __syncthreads();
    if (tid < 48) {                                                                   // cat([coor, mask], 2).permute(0, 2, 1): rows x, y, mask
      const int p = tid / 3, c = tid - p * 3;
      in0[c * kIfRow + 1 + p] = c < 2 ? x[((size_t)seq * kIfL + p) * 2 + c] : m[(size_t)seq * kIfL + p];
    }
    __syncthreads();
    {                                                                                 // down_1: 3 -> 32, 512 outputs over the workgroup
      const float* w = packed + kIfStemOff;
#pragma unroll
      for (int o = 0; o < 512 / NT; ++o) {
        const int e = tid * (512 / NT) + o, co = e >> 4, p = e & 15;
        float s = bias[kIfBiasAt[0] + co];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
          for (int k = 0; k < 3; ++k) s = fmaf(w[(co * 3 + ci) * 3 + k], in0[ci * kIfRow + p + k], s);
        s = s > 0.0f ? s : 0.01f * s;
        x1[co * kIfRow + 1 + p] = s;
        if (acts) acts[((size_t)seq * kItActCh + kItActOff[0] + co) * kIfL + p] = s;
      }
    }
    float* ga = acts ? acts + (size_t)seq * kItActCh * kIfL : nullptr;
    auto gs = [&](int i) -> float* { return ga ? ga + kItActOff[i] * kIfL : nullptr; };
    __syncthreads();
    if_dense_layer<32, 0, 64, NW>(packed + if_layer_offset(0), bias + kIfBiasAt[1], x1, x1, x2, wave, lane, gs(1));
    __syncthreads();
    if_dense_layer<64, 0, 128, NW>(packed + if_layer_offset(1), bias + kIfBiasAt[2], x2, x2, x3, wave, lane, gs(2));
    __syncthreads();
    if_dense_layer<128, 0, 256, NW>(packed + if_layer_offset(2), bias + kIfBiasAt[3], x3, x3, pa, wave, lane, gs(3));
    __syncthreads();
    if_dense_layer<256, 0, 256, NW>(packed + if_layer_offset(3), bias + kIfBiasAt[4], pa, pa, pb, wave, lane, gs(4));
    __syncthreads();
    if_dense_layer<256, 128, 128, NW>(packed + if_layer_offset(4), bias + kIfBiasAt[5], pb, x3, pa, wave, lane, gs(5));      // cat([x, x3], 1)
    __syncthreads();
    if_dense_layer<128, 64, 64, NW>(packed + if_layer_offset(5), bias + kIfBiasAt[6], pa, x2, pb, wave, lane, gs(6));        // cat([x, x2], 1)
    __syncthreads();
    if_dense_layer<64, 32, 32, NW>(packed + if_layer_offset(6), bias + kIfBiasAt[7], pb, x1, pa, wave, lane, gs(7));         // cat([x, x1], 1)
    __syncthreads();
    if (tid < 32) {                                                                   // predictor 32 -> 2, sigmoid, permute back to [L][2]
      const float* w = packed + kIfHeadOff;
      const int p = tid >> 1, co = tid & 1;
      float s = bias[kIfBiasAt[8] + co];
      for (int ci = 0; ci < 32; ++ci)
#pragma unroll
        for (int k = 0; k < 3; ++k) s = fmaf(w[(co * 32 + ci) * 3 + k], pa[ci * kIfRow + p + k], s);
      out[((size_t)seq * kIfL + p) * 2 + co] = 1.0f / (1.0f + expf(-s));
    }
  }
}

}  // namespace tnv3
