// conv1d_k3.h -- InpaintNet's building block: nn.Conv1d(k=3, padding='same', bias=True) + LeakyReLU(0.01)
// (model.py:76-87), the channel concats of InpaintNet.forward (model.py:120,122,124), the input assembly
// cat([coor, mask], dim=2).permute(0,2,1) (model.py:114-115) and the final sigmoid + permute (model.py:126-128).
//
// Work decomposition: one workgroup = S sequences x COB output channels x LT positions; thread = (channel, sequence)
// with LT accumulators.  Input chunks [S][CK][LT+2] and weight chunks [CK*3][COB] are staged through LDS; the LT+2
// input values of a row are read as 16-byte LDS broadcasts (all lanes of a half-wave share the sequence), the three
// weights as conflict-free consecutive-lane reads.  fp32 vector FMA: 16.6 MFLOP per 16-step sequence,
// weights (2 MB) stay L2-resident.  Concat inputs are two source pointers, never materialised.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

typedef float c1_f32x4 __attribute__((ext_vector_type(4)));

struct Conv1dArgs {
  const float* src0;   // [N][C0][L], or [N][L][C0] when src_nlc
  const float* src1;   // [N][C1][L] / [N][L][C1] or nullptr; channels follow src0's (torch.cat([a, b], dim=1))
  const float* w;      // [Cout][C0+C1][3]   (the nn.Conv1d weight as stored in the state_dict)
  const float* b;      // [Cout]
  float* dst;          // [N][Cout][L], or [N][L][Cout] when dst_nlc
  int N, C0, C1, Cout, L;
  int src_nlc, dst_nlc;
  int act;             // 0: none, 1: LeakyReLU(0.01), 2: sigmoid
};

template <int S, int COB, int CK, int LT>
__global__ void __launch_bounds__(S* COB) conv1d_k3_kernel(const Conv1dArgs a) {
  constexpr int NT = S * COB;
  constexpr int LP = LT + 2;          // staged positions per row (halo 1 each side)
  constexpr int LTP = LT + 4;         // LDS row stride (16-byte aligned rows)
  constexpr int WS = COB + 1;         // padded weight row (conflict-free transposed store)
  static_assert(LT % 4 == 0 && COB == 32, "thread map assumes 32 output channels per half-wave");
  __shared__ __attribute__((aligned(16))) float in_s[S * CK * LTP];
  __shared__ float w_s[CK * 3 * WS];

  const int tid = threadIdx.x;
  const int co_l = tid % COB, s_l = tid / COB;
  const int n0 = blockIdx.x * S, co0 = blockIdx.y * COB, l0 = blockIdx.z * LT;
  const int Cin = a.C0 + a.C1, L = a.L;

  float acc[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) acc[l] = 0.0f;

  for (int c0 = 0; c0 < Cin; c0 += CK) {
    __syncthreads();
    // ---- stage inputs: element (s, c, p) <- src[n0+s][c0+c][l0-1+p], zero outside the sequence / batch / channels
    for (int idx = tid; idx < S * CK * LP; idx += NT) {
      const int p = idx % LP, t = idx / LP;
      const int c = t % CK, s = t / CK;
      const int n = n0 + s, ch = c0 + c, l = l0 - 1 + p;
      float v = 0.0f;
      if (n < a.N && ch < Cin && l >= 0 && l < L) {
        if (ch < a.C0) v = a.src_nlc ? a.src0[((size_t)n * L + l) * a.C0 + ch] : a.src0[((size_t)n * a.C0 + ch) * L + l];
        else v = a.src_nlc ? a.src1[((size_t)n * L + l) * a.C1 + (ch - a.C0)] : a.src1[((size_t)n * a.C1 + (ch - a.C0)) * L + l];
      }
      in_s[(s * CK + c) * LTP + p] = v;
    }
    // ---- stage weights transposed: w_s[(ci*3+k)][co] <- w[co0+co][c0+ci][k]
    for (int idx = tid; idx < COB * CK * 3; idx += NT) {
      const int j = idx % (CK * 3), co = idx / (CK * 3);
      float v = 0.0f;
      if (co0 + co < a.Cout && c0 + j / 3 < Cin) v = a.w[((size_t)(co0 + co) * Cin + c0) * 3 + j];
      w_s[j * WS + co] = v;
    }
    __syncthreads();
    // ---- accumulate
#pragma unroll 4
    for (int c = 0; c < CK; ++c) {
      float x[LTP];
      const c1_f32x4* row = reinterpret_cast<const c1_f32x4*>(in_s + (s_l * CK + c) * LTP);
#pragma unroll
      for (int q = 0; q < LTP / 4; ++q) {
        const c1_f32x4 v = row[q];
        x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
      }
      const float w0 = w_s[(c * 3 + 0) * WS + co_l], w1 = w_s[(c * 3 + 1) * WS + co_l], w2 = w_s[(c * 3 + 2) * WS + co_l];
#pragma unroll
      for (int l = 0; l < LT; ++l) acc[l] = fmaf(w2, x[l + 2], fmaf(w1, x[l + 1], fmaf(w0, x[l], acc[l])));
    }
  }

  const int n = n0 + s_l, co = co0 + co_l;
  if (n < a.N && co < a.Cout) {
    const float bias = a.b[co];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      if (l0 + l < L) {
        float v = acc[l] + bias;
        if (a.act == 1) v = v > 0.0f ? v : 0.01f * v;
        else if (a.act == 2) v = 1.0f / (1.0f + expf(-v));
        if (a.dst_nlc) a.dst[((size_t)n * L + l0 + l) * a.Cout + co] = v;
        else a.dst[((size_t)n * a.Cout + co) * L + l0 + l] = v;
      }
    }
  }
}

}  // namespace tnv3
