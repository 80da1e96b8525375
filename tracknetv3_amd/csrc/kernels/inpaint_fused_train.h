// inpaint_fused_train.h -- the backward pass of an InpaintNet training step (train.py:147-166 through model.py:113-129) in TWO
// launches instead of the ~26 of the per-layer path (nine activation-derivative passes, eight data gradients, nine weight
// gradients + their reductions), on the matrix cores:
//
//   inpaintnet_fused_dgrad_kernel  one workgroup carries a sequence's gradient backwards through all nine layers with every
//                                  gradient row resident in LDS (the mirror image of inpaintnet_fused_kernel): the seven dense
//                                  data gradients are v_mfma_f32_16x16x4_f32 GEMMs over transposed / tap-flipped filters packed in
//                                  lane order (inpaint_pack_t_kernel); LeakyReLU / sigmoid derivatives, the concat splits and the
//                                  skip-gradient additions happen in the epilogues.  Output: dPre[seq][962][16], the gradient at
//                                  every layer's pre-activation (960 hidden rows in the activation order + the predictor's 2).
//   inpaintnet_wgrad_all_kernel    dW and db of ALL nine layers in one launch: one wave per (layer, 16 output channels, 16 input
//                                  channels) runs the three taps' GEMMs dW_t[co][ci] = sum_{n,p} dPre[n][co][p] * in[n][ci][p+t-1]
//                                  over the whole batch (MFMA K = positions; fixed order => deterministic), straight from L2.
// The training forward is inpaintnet_fused_kernel with its `acts` argument set (activations saved as [seq][960][16]).
// Same functions as conv1d_k3.h's kernels up to fp32 summation order (tests: <= 2e-5 of each gradient's scale vs fp64 autograd).
#pragma once
#include "inpaint_fused.h"

namespace tnv3 {

constexpr int kItPreCh = kItActCh + 2;                       // 962: hidden rows in activation order, then the predictor's two rows
constexpr int kItParamFloats = 520610;                       // all 18 tensors in state_dict order (weight, bias per layer)
// forward order: down_1, down_2, down_3, buttleneck.conv_1, .conv_2, up_1, up_2, up_3, predictor
constexpr int kItCout[9] = {32, 64, 128, 256, 256, 128, 64, 32, 2};
constexpr int kItCin[9] = {3, 32, 64, 128, 256, 384, 192, 96, 32};
constexpr int it_w_off(int i) { return i == 0 ? 0 : it_w_off(i - 1) + kItCout[i - 1] * kItCin[i - 1] * 3 + kItCout[i - 1]; }   // weight of layer i
constexpr int it_b_off(int i) { return it_w_off(i) + kItCout[i] * kItCin[i] * 3; }
static_assert(it_b_off(8) + 2 == kItParamFloats, "InpaintNet parameter count");
// (tables, not calls: a constexpr function called with a loop variable in device code is emitted as a real -- here recursive -- call)
constexpr int kItWOff[9] = {it_w_off(0), it_w_off(1), it_w_off(2), it_w_off(3), it_w_off(4), it_w_off(5), it_w_off(6), it_w_off(7), it_w_off(8)};
constexpr int kItBOff[9] = {it_b_off(0), it_b_off(1), it_b_off(2), it_b_off(3), it_b_off(4), it_b_off(5), it_b_off(6), it_b_off(7), it_b_off(8)};

// ---- transposed, tap-flipped filters of the seven dense layers in the forward kernel's lane order: the data gradient of
//      y[co][p] = sum W[co][ci][k] x[ci][p+k-1] is dx[ci][p] = sum_{co,t} Wt[ci][co][t] dy[co][p+t-1] with Wt[ci][co][t] = W[co][ci][2-t]
//      -- the same 'same'-padded convolution with M = Cin, K = (tap, Cout).  Layer i (dense index 0..6 = forward layers 1..7):
//      Wtp[block][chunk][lane][j] = W[co = 16*(chunk % (Cout/16)) + 4*(lane >> 4) + j][ci = 16*block + (lane & 15)][2 - chunk / (Cout/16)]
struct InpaintPackTArgs {
  const float* w[7];          // the nn.Conv1d weights [Cout][Cin][3] of forward layers 1..7
  float* packed_t;            // kIfStemOff floats (same per-layer offsets as the forward pack: if_layer_offset)
};
inline __global__ void __launch_bounds__(256) inpaint_pack_t_kernel(const InpaintPackTArgs a) {
  const int stride = gridDim.x * blockDim.x;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kIfStemOff; e += stride) {
    int li = 0, cin = kIfLayers[0].c0 + kIfLayers[0].c1, cout = kIfLayers[0].cout, base = 0;
#pragma unroll
    for (int i = 1; i < 7; ++i)
      if (e >= if_layer_offset(i)) { li = i; cin = kIfLayers[i].c0 + kIfLayers[i].c1; cout = kIfLayers[i].cout; base = if_layer_offset(i); }
    const int nch = 3 * cout / 16;                            // K chunks of the data gradient: (tap, 16 forward output channels)
    const int r = e - base;
    const int j = r & 3, lane = (r >> 2) & 63, q = (r >> 8) % nch, blk = (r >> 8) / nch;
    const int t = q / (cout / 16), co = (q % (cout / 16)) * 16 + 4 * (lane >> 4) + j, ci = blk * 16 + (lane & 15);
    a.packed_t[e] = a.w[li][((size_t)co * cin + ci) * 3 + (2 - t)];
  }
}

constexpr int kItRows = 256 + 256 + 32 + 64 + 128;          // ping, pong, skip gradients of x1, x2, x3
constexpr int kItLdsFloats = kItRows * kIfRow;

// dout, out [N][16][2] (gradient at / value of the sigmoid output), acts [N][960][16], packed_t (inpaint_pack_t_kernel), packed (the
// forward pack: the predictor's filter is read from it as stored) -> dpre [N][962][16]
inline __global__ void __launch_bounds__(256, 2) inpaintnet_fused_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                            const float* __restrict__ acts, const float* __restrict__ packed_t,
                                                                            const float* __restrict__ packed, float* __restrict__ dpre, int N) {
  __shared__ __attribute__((aligned(16))) float lds[kItLdsFloats];
  float* ga = lds;
  float* gb = ga + 256 * kIfRow;
  float* s1 = gb + 256 * kIfRow;
  float* s2 = s1 + 32 * kIfRow;
  float* s3 = s2 + 64 * kIfRow;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kItLdsFloats; i += 256) lds[i] = 0.0f;                      // the halo columns stay zero for good
  for (int seq = blockIdx.x; seq < N; seq += gridDim.x) {
    const float* act = acts + (size_t)seq * kItActCh * kIfL;
    float* dp = dpre + (size_t)seq * kItPreCh * kIfL;
    // epilogues: `main` rows get the activation derivative of the layer that produced them (LeakyReLU: 1 or 0.01 by the sign of the
    // saved activation) and become that layer's dPre -- to LDS (next data gradient's source) and to HBM (weight gradient);
    // `skip` rows (the concat's second operand) are parked raw and added where that tensor's own gradient arrives.
    auto lrelu_d = [](float a) { return a > 0.0f ? 1.0f : 0.01f; };
    __syncthreads();
    if (tid < 32) {                                                                   // predictor: dPre = dOut * out * (1 - out)
      const int p = tid >> 1, c = tid & 1;
      const float o = out[((size_t)seq * kIfL + p) * 2 + c];
      const float g = dout[((size_t)seq * kIfL + p) * 2 + c] * (o * (1.0f - o));
      ga[c * kIfRow + 1 + p] = g;
      dp[(kItActCh + c) * kIfL + p] = g;
    }
    __syncthreads();
    {                                                                                 // predictor's data gradient 2 -> 32 (K = 6: vector code), x LeakyReLU'(u3)
      const float* w = packed + kIfHeadOff;                                           // [2][32][3] as stored
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int e = tid * 2 + o, ci = e >> 4, p = e & 15;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int k = 0; k < 3; ++k) s = fmaf(w[(c * 32 + ci) * 3 + k], ga[c * kIfRow + p + 2 - k], s);
        s *= lrelu_d(act[(kItActOff[7] + ci) * kIfL + p]);
        gb[ci * kIfRow + 1 + p] = s;
        dp[(kItActOff[7] + ci) * kIfL + p] = s;
      }
    }
    __syncthreads();
    // up_3 = conv(cat([u2, x1])): 32 -> 64 (u2) | 32 (x1)
    if_dense_core<32, 0, 96>(packed_t + if_layer_offset(6), gb, gb, wave, lane, [&](int c, int n, float v) {
      if (c < 64) { v *= lrelu_d(act[(kItActOff[6] + c) * kIfL + n]); ga[c * kIfRow + 1 + n] = v; dp[(kItActOff[6] + c) * kIfL + n] = v; }
      else s1[(c - 64) * kIfRow + 1 + n] = v;
    });
    __syncthreads();
    // up_2 = conv(cat([u1, x2])): 64 -> 128 (u1) | 64 (x2)
    if_dense_core<64, 0, 192>(packed_t + if_layer_offset(5), ga, ga, wave, lane, [&](int c, int n, float v) {
      if (c < 128) { v *= lrelu_d(act[(kItActOff[5] + c) * kIfL + n]); gb[c * kIfRow + 1 + n] = v; dp[(kItActOff[5] + c) * kIfL + n] = v; }
      else s2[(c - 128) * kIfRow + 1 + n] = v;
    });
    __syncthreads();
    // up_1 = conv(cat([b2, x3])): 128 -> 256 (b2) | 128 (x3)
    if_dense_core<128, 0, 384>(packed_t + if_layer_offset(4), gb, gb, wave, lane, [&](int c, int n, float v) {
      if (c < 256) { v *= lrelu_d(act[(kItActOff[4] + c) * kIfL + n]); ga[c * kIfRow + 1 + n] = v; dp[(kItActOff[4] + c) * kIfL + n] = v; }
      else s3[(c - 256) * kIfRow + 1 + n] = v;
    });
    __syncthreads();
    // buttleneck.conv_2: 256 -> 256 (b1)
    if_dense_core<256, 0, 256>(packed_t + if_layer_offset(3), ga, ga, wave, lane, [&](int c, int n, float v) {
      v *= lrelu_d(act[(kItActOff[3] + c) * kIfL + n]); gb[c * kIfRow + 1 + n] = v; dp[(kItActOff[3] + c) * kIfL + n] = v;
    });
    __syncthreads();
    // buttleneck.conv_1: 256 -> 128 (x3), + the skip gradient parked by up_1
    if_dense_core<256, 0, 128>(packed_t + if_layer_offset(2), gb, gb, wave, lane, [&](int c, int n, float v) {
      v = (v + s3[c * kIfRow + 1 + n]) * lrelu_d(act[(kItActOff[2] + c) * kIfL + n]); ga[c * kIfRow + 1 + n] = v; dp[(kItActOff[2] + c) * kIfL + n] = v;
    });
    __syncthreads();
    // down_3: 128 -> 64 (x2), + up_2's skip gradient
    if_dense_core<128, 0, 64>(packed_t + if_layer_offset(1), ga, ga, wave, lane, [&](int c, int n, float v) {
      v = (v + s2[c * kIfRow + 1 + n]) * lrelu_d(act[(kItActOff[1] + c) * kIfL + n]); gb[c * kIfRow + 1 + n] = v; dp[(kItActOff[1] + c) * kIfL + n] = v;
    });
    __syncthreads();
    // down_2: 64 -> 32 (x1), + up_3's skip gradient; down_1's inputs need no gradient
    if_dense_core<64, 0, 32>(packed_t + if_layer_offset(0), gb, gb, wave, lane, [&](int c, int n, float v) {
      v = (v + s1[c * kIfRow + 1 + n]) * lrelu_d(act[(kItActOff[0] + c) * kIfL + n]); dp[(kItActOff[0] + c) * kIfL + n] = v;
    });
  }
}

// ---- weight / bias gradients of all nine layers: grads[kItParamFloats] in state_dict order
// dense item table (forward layers 1..7): first item, ci blocks, dPre row offset, input rows = (acts offset a0, channels c0 | a1)
struct ItWgradLayer { int first, cib, layer, pre, a0, c0, a1; };
constexpr ItWgradLayer kItWg[7] = {{0, 2, 1, 32, 0, 32, 0},             // down_2: in x1
                                   {8, 4, 2, 96, 32, 64, 0},            // down_3: in x2
                                   {40, 8, 3, 224, 96, 128, 0},         // buttleneck.conv_1: in x3
                                   {168, 16, 4, 480, 224, 256, 0},      // buttleneck.conv_2: in b1
                                   {424, 24, 5, 736, 480, 256, 96},     // up_1: in cat([b2, x3])
                                   {616, 12, 6, 864, 736, 128, 32},     // up_2: in cat([u1, x2])
                                   {664, 6, 7, 928, 864, 64, 0}};       // up_3: in cat([u2, x1])
constexpr int kItWgItems = 676;                                         // sum over dense layers of (Cout / 16) * (Cin / 16)
constexpr int kItWgBlocks = kItWgItems / 4 + 2;                         // four items (waves) per workgroup, + stem, + head

inline __global__ void __launch_bounds__(256) inpaintnet_wgrad_all_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                                        const float* __restrict__ acts, const float* __restrict__ dpre,
                                                                        float* __restrict__ grads, int N) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == kItWgItems / 4) {                                   // down_1: dW [32][3][3], db [32]; in = cat([x, m], 2) read as stored
    for (int e = tid; e < 288 + 32; e += 256) {
      float s = 0.0f;
      if (e < 288) {
        const int co = e / 9, ci = (e / 3) % 3, k = e % 3;
        for (int n = 0; n < N; ++n) {
          const float* d = dpre + ((size_t)n * kItPreCh + co) * kIfL;
          for (int p = 0; p < kIfL; ++p) {
            const int q = p + k - 1;
            if (q < 0 || q >= kIfL) continue;
            const float v = ci < 2 ? x[((size_t)n * kIfL + q) * 2 + ci] : m[(size_t)n * kIfL + q];
            s = fmaf(d[p], v, s);
          }
        }
        grads[kItWOff[0] + e] = s;
      } else {
        const int co = e - 288;
        for (int n = 0; n < N; ++n) {
          const float* d = dpre + ((size_t)n * kItPreCh + co) * kIfL;
          for (int p = 0; p < kIfL; ++p) s += d[p];
        }
        grads[kItBOff[0] + co] = s;
      }
    }
    return;
  }
  if (blockIdx.x == kItWgItems / 4 + 1) {                               // predictor: dW [2][32][3], db [2]; in = u3
    for (int e = tid; e < 192 + 2; e += 256) {
      float s = 0.0f;
      if (e < 192) {
        const int co = e / 96, ci = (e / 3) % 32, k = e % 3;
        for (int n = 0; n < N; ++n) {
          const float* d = dpre + ((size_t)n * kItPreCh + kItActCh + co) * kIfL;
          const float* a = acts + ((size_t)n * kItActCh + kItActOff[7] + ci) * kIfL;
          for (int p = 0; p < kIfL; ++p) {
            const int q = p + k - 1;
            if (q >= 0 && q < kIfL) s = fmaf(d[p], a[q], s);
          }
        }
        grads[kItWOff[8] + e] = s;
      } else {
        const int co = e - 192;
        for (int n = 0; n < N; ++n) {
          const float* d = dpre + ((size_t)n * kItPreCh + kItActCh + co) * kIfL;
          for (int p = 0; p < kIfL; ++p) s += d[p];
        }
        grads[kItBOff[8] + co] = s;
      }
    }
    return;
  }
  // ---- dense layers: this wave's item = (layer, 16 output channels, 16 input channels), all three taps
  const int item = blockIdx.x * 4 + wave;
  int li = 0;
#pragma unroll
  for (int i = 1; i < 7; ++i)
    if (item >= kItWg[i].first) li = i;
  int first = 0, cib = 0, layer = 0, pre = 0, a0 = 0, c0 = 0, a1 = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {                                          // (selects on constants: the table itself lives in host memory)
    const bool hit = i == li;
    first = hit ? kItWg[i].first : first; cib = hit ? kItWg[i].cib : cib; layer = hit ? kItWg[i].layer : layer; pre = hit ? kItWg[i].pre : pre;
    a0 = hit ? kItWg[i].a0 : a0; c0 = hit ? kItWg[i].c0 : c0; a1 = hit ? kItWg[i].a1 : a1;
  }
  const int r = item - first, cob = r / cib, cb = r - cob * cib;
  const int cin = cib * 16;
  const int ci0 = cb * 16;
  const int in_row = ci0 < c0 ? a0 + ci0 : a1 + (ci0 - c0);              // 16-channel blocks never straddle the concat (c0 % 16 == 0)
  const int m16 = lane & 15, kq = lane >> 4;
  // MFMA K index kq at step j <-> position 4*kq + j (a permutation of the 16 positions over the four steps: it is a plain sum)
  const float* dptr = dpre + ((size_t)(pre + cob * 16 + m16)) * kIfL + 4 * kq;
  const float* iptr = acts + ((size_t)(in_row + m16)) * kIfL + 4 * kq;
  if_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0, acc2 = acc0;
  float bsum = 0.0f;
  for (int n = 0; n < N; ++n) {
    const if_f32x4 dq = *reinterpret_cast<const if_f32x4*>(dptr + (size_t)n * kItPreCh * kIfL);
    const float* ip = iptr + (size_t)n * kItActCh * kIfL;
    const if_f32x4 q = *reinterpret_cast<const if_f32x4*>(ip);
    const float left = kq > 0 ? ip[-1] : 0.0f, right = kq < 3 ? ip[4] : 0.0f;
    const if_f32x4 b0 = {left, q[0], q[1], q[2]}, b2 = {q[1], q[2], q[3], right};
#define TNV3_IT_STEP(j)                                                                \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[j], b0[j], acc0, 0, 0, 0);          \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[j], q[j], acc1, 0, 0, 0);           \
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[j], b2[j], acc2, 0, 0, 0);
    TNV3_IT_STEP(0) TNV3_IT_STEP(1) TNV3_IT_STEP(2) TNV3_IT_STEP(3)
#undef TNV3_IT_STEP
    bsum += (dq[0] + dq[1]) + (dq[2] + dq[3]);
  }
  int wo = 0, bo = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i == layer) { wo = kItWOff[i]; bo = kItBOff[i]; }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {                                       // D layout: row (co) = 4 * (lane >> 4) + r, column (ci) = lane & 15
    const int co = cob * 16 + 4 * kq + rr;
    float* g = grads + wo + ((size_t)co * cin + ci0 + m16) * 3;
    g[0] = acc0[rr]; g[1] = acc1[rr]; g[2] = acc2[rr];
  }
  if (cb == 0) {                                                         // bias gradient of these 16 channels: fold the four position quads
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (kq == 0) grads[bo + cob * 16 + m16] = bsum;
  }
}

}  // namespace tnv3
