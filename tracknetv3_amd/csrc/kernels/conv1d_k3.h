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
  // ---- data-gradient mode (backward of the same layer): the kernel then computes dX = conv1d(dPre, W^T flipped)
  int wt;              // 1: w is the FORWARD filter [Cin_k = C0+C1][Cout_m = Cout][3] read as w'[m][kc][k] = w[kc][m][2-k]
  float* dst1;         // optional: output channels >= csplit go to dst1 [N][Cout-csplit][L] (gradient of a concat)
  int csplit;
  int accumulate;      // bit 0: dst += result, bit 1: dst1 += result (a tensor consumed twice receives two gradients)
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
      if (co0 + co < a.Cout && c0 + j / 3 < Cin)
        v = a.wt ? a.w[((size_t)(c0 + j / 3) * a.Cout + (co0 + co)) * 3 + (2 - j % 3)]
                 : a.w[((size_t)(co0 + co) * Cin + c0) * 3 + j];
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
    const float bias = a.b ? a.b[co] : 0.0f;
    const bool second = a.dst1 != nullptr && co >= a.csplit;
    float* dbase = second ? a.dst1 : a.dst;
    const int cdst = a.dst1 ? (second ? a.Cout - a.csplit : a.csplit) : a.Cout;
    const int cod = second ? co - a.csplit : co;
    const bool accum = second ? (a.accumulate & 2) : (a.accumulate & 1);
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      if (l0 + l < L) {
        float v = acc[l] + bias;
        if (a.act == 1) v = v > 0.0f ? v : 0.01f * v;
        else if (a.act == 2) v = 1.0f / (1.0f + expf(-v));
        float* d = a.dst_nlc ? dbase + ((size_t)n * L + l0 + l) * cdst + cod : dbase + ((size_t)n * cdst + cod) * L + l0 + l;
        *d = accum ? *d + v : v;
      }
    }
  }
}

// dPre = dOut * act'(out):  LeakyReLU'(0.01) from the sign of the output, sigmoid' = o(1-o).   (autograd of model.py:81,111)
// nlc != 0: dOut / out are [N][L][C] (the network output), dPre is always [N][C][L].
inline __global__ void __launch_bounds__(256) conv1d_act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                             float* __restrict__ dpre, int N, int C, int L, int act, int nlc) {
  const long total = (long)N * C * L;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int l = (int)(t % L);
    const long u = t / L;
    const int c = (int)(u % C);
    const long n = u / C;
    const size_t src = nlc ? ((size_t)n * L + l) * C + c : (size_t)t;
    const float o = out[src], g = dout[src];
    float d = g;
    if (act == 1) d = o > 0.0f ? g : 0.01f * g;
    else if (act == 2) d = g * o * (1.0f - o);
    dpre[t] = d;
  }
}

// Weight / bias gradient:  dW[co][ci][k] = sum_{n,l} dPre[n][co][l] * X[n][ci][l+k-1],  db[co] = sum_{n,l} dPre[n][co][l]
// grid = (ci blocks of 32, co blocks of 32, nsplit): each workgroup walks its share of the sequences in groups of S,
// stages dPre [S][32][L] and X [S][32][L+2] in LDS, thread = (co, 4 input channels) with 12 accumulators, and writes its
// partial slab part[split][Cout*Cin*3 + Cout]; sum_partials_kernel reduces the slabs in a fixed order.
template <int S, int LMAX>
__global__ void __launch_bounds__(256) conv1d_wgrad_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                                           const float* __restrict__ dpre, float* __restrict__ part, int N,
                                                           int C0, int C1, int Cout, int L, int src_nlc) {
  __shared__ float g_s[S * 32 * (LMAX + 1)];
  __shared__ float x_s[S * 32 * (LMAX + 2)];
  const int tid = threadIdx.x;
  const int co_l = tid & 31, cq = tid >> 5;              // 8 groups of 4 input channels
  const int Cin = C0 + C1;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int nsplit = gridDim.z, sp = blockIdx.z;
  float acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = acc[i][1] = acc[i][2] = 0.0f; }
  float accb = 0.0f;
  const int LP = L + 2, GS = L | 1;       // odd dPre row stride: the 32 channel-lanes hit 32 banks
  for (int n0 = sp * S; n0 < N; n0 += nsplit * S) {
    __syncthreads();
    for (int idx = tid; idx < S * 32 * L; idx += 256) {
      const int l = idx % L, t = idx / L;
      const int c = t % 32, s = t / 32;
      const int n = n0 + s, co = co0 + c;
      g_s[t * GS + l] = (n < N && co < Cout) ? dpre[((size_t)n * Cout + co) * L + l] : 0.0f;
    }
    for (int idx = tid; idx < S * 32 * LP; idx += 256) {
      const int p = idx % LP, t = idx / LP;
      const int c = t % 32, s = t / 32;
      const int n = n0 + s, ch = ci0 + c, l = p - 1;
      float v = 0.0f;
      if (n < N && ch < Cin && l >= 0 && l < L) {
        if (ch < C0) v = src_nlc ? src0[((size_t)n * L + l) * C0 + ch] : src0[((size_t)n * C0 + ch) * L + l];
        else v = src_nlc ? src1[((size_t)n * L + l) * C1 + (ch - C0)] : src1[((size_t)n * C1 + (ch - C0)) * L + l];
      }
      x_s[idx] = v;
    }
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      for (int l = 0; l < L; ++l) {
        const float g = g_s[(s * 32 + co_l) * GS + l];
        if (cq == 0) accb += g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* xr = x_s + (s * 32 + cq * 4 + i) * LP + l;     // broadcast within the 32 lanes sharing cq
          acc[i][0] = fmaf(g, xr[0], acc[i][0]);
          acc[i][1] = fmaf(g, xr[1], acc[i][1]);
          acc[i][2] = fmaf(g, xr[2], acc[i][2]);
        }
      }
    }
  }
  float* slab = part + (size_t)sp * ((size_t)Cout * Cin * 3 + Cout);
  const int co = co0 + co_l;
  if (co < Cout) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = ci0 + cq * 4 + i;
      if (ci < Cin) { float* d = slab + ((size_t)co * Cin + ci) * 3; d[0] = acc[i][0]; d[1] = acc[i][1]; d[2] = acc[i][2]; }
    }
    if (cq == 0 && blockIdx.x == 0) slab[(size_t)Cout * Cin * 3 + co] = accb;
  }
}

}  // namespace tnv3
