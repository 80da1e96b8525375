// tnv3_impl.h -- host-side argument checking and kernel dispatch behind the C ABI (include/tracknetv3_hip.h).
// Templated on a Launcher so that the CPU test-suite can drive the very same dispatch code through the SIMT
// emulator (tests/emu); the shipped library instantiates it with the HIP launcher only (tnv3_capi.hip).
#pragma once
#include <stdio.h>
#include <string.h>

#include "kernels/conv3x3_mfma.h"
#include "kernels/conv_up2x_mfma.h"
#include "kernels/conv_up2x_wino_mfma.h"
#include "kernels/dgrad_up2x_wino_mfma.h"
#include "kernels/conv3x3_wino_mfma.h"
#include "kernels/conv3x3_wino3_mfma.h"
#include "kernels/conv3x3_wino6_mfma.h"
#include "kernels/conv3x3_wino43_mfma.h"
#include "kernels/conv3x3_wino43s_mfma.h"
#include "kernels/conv1d_k3.h"
#include "kernels/conv1d_mfma.h"
#include "kernels/inpaint_fused.h"
#include "kernels/inpaint_fused_train.h"
#include "kernels/pointwise.h"
#include "kernels/postproc.h"
#include "kernels/preproc.h"
#include "kernels/optim.h"
#include "kernels/train_ops.h"
#include "kernels/wgrad3x3_mfma.h"
#include "kernels/wgrad_wino_mfma.h"
#include "kernels/wgrad_wino43_mfma.h"
#include "kernels/wgrad_wino43_r5_mfma.h"
#include "kernels/wgrad_up2x_wino43_mfma.h"

namespace tnv3 {

// Since ABI 6 the product library carries, per kernel family, what dispatches by default plus one fallback per shape class; every generation that
// was measured and rejected is a measurement twin of libtnv3_diag.so (built from this file with -DTNV3_DIAG) and is refused here.
#ifdef TNV3_DIAG
constexpr bool kTwins = true;
#else
constexpr bool kTwins = false;
#endif


inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
#define TNV3_FAIL(code, ...)                               \
  do {                                                     \
    snprintf(::tnv3::err_buf(), 512, __VA_ARGS__);         \
    return (code);                                         \
  } while (0)

// ---- compiled conv tile configurations: <MT, NTW, WM, WN, TR, TC, CC>
using ConvC0 = ConvCfg<2, 2, 1, 4, 8, 32, 8>;    // 64 ch x ( 8x32 px)
using ConvC1 = ConvCfg<2, 4, 1, 4, 16, 32, 8>;   // 64 ch x (16x32 px)
using ConvC2 = ConvCfg<4, 2, 1, 4, 8, 32, 4>;    // 128 ch x ( 8x32 px), wave = 128 ch x 2 rows
using ConvC3 = ConvCfg<2, 4, 2, 2, 8, 32, 4>;    // 128 ch x ( 8x32 px), wave =  64 ch x 4 rows
using ConvC4 = ConvCfg<2, 2, 2, 2, 4, 32, 4>;    // 128 ch x ( 4x32 px)
using ConvC5 = ConvCfg<2, 2, 1, 4, 4, 64, 8>;    // 64 ch x ( 4x64 px)
using ConvC6 = ConvCfg<4, 2, 1, 4, 8, 32, 8>;    // 128 ch x ( 8x32 px), CC = 8
using ConvC7 = ConvCfg<2, 1, 1, 4, 4, 32, 8>;    // 64 ch x ( 4x32 px): 3 waves/SIMD
using ConvC8 = ConvCfg<2, 1, 1, 8, 8, 32, 8>;    // 64 ch x ( 8x32 px), 512 threads: cfg 7's wave tile, weight panel shared by 8 waves
using ConvC9 = ConvCfg<2, 1, 2, 4, 4, 32, 4>;    // 128 ch x ( 4x32 px), 512 threads: input tile shared by two channel halves
using ConvC10 = ConvCfg<2, 1, 1, 8, 8, 32, 8, 1, 2, 1>;   // cfg 8 + 2-step operand prefetch + setprio around the MFMA block
using ConvC11 = ConvCfg<2, 1, 2, 4, 4, 32, 4, 1, 2, 1>;   // cfg 9 + both (+1.5 % measured)
using ConvC12 = ConvCfg<2, 1, 1, 4, 4, 32, 8, 1, 2, 1>;   // cfg 7 + both
using ConvC13 = ConvCfg<2, 1, 1, 4, 4, 32, 8, 1, 2, 1, 1>;   // cfg 12 with LDS-DMA staging
using ConvC14 = ConvCfg<2, 1, 1, 8, 8, 32, 8, 1, 2, 1, 1>;   // cfg 10 with LDS-DMA staging
using ConvC15 = ConvCfg<2, 1, 2, 4, 4, 32, 4, 1, 2, 1, 1>;   // cfg 11 with LDS-DMA staging
using ConvC16 = ConvCfg<2, 1, 1, 4, 4, 32, 4, 1, 2, 1, 2>;   // cfg 12 tile, CC 4, 3-stage LDS-DMA + counted vmcnt
using ConvC17 = ConvCfg<2, 1, 2, 4, 4, 32, 4, 1, 2, 1, 2>;   // cfg 11 tile,       3-stage LDS-DMA + counted vmcnt
using ConvC18 = ConvCfg<2, 1, 1, 8, 8, 32, 4, 1, 2, 1, 2>;   // cfg 10 tile, CC 4, 3-stage LDS-DMA + counted vmcnt
constexpr int kNumConvConfigs = 19;
#ifdef TNV3_DIAG
// diagnostic twins (libtnv3_diag.so only: tnv3_diag_conv3x3_forward): same geometry, runtime `diag` honoured
using ConvD10 = ConvCfg<2, 1, 1, 8, 8, 32, 8, 1, 2, 1, 0, 1>;
using ConvD11 = ConvCfg<2, 1, 2, 4, 4, 32, 4, 1, 2, 1, 0, 1>;
using ConvD12 = ConvCfg<2, 1, 1, 4, 4, 32, 8, 1, 2, 1, 0, 1>;
#endif

struct ConvCfgInfo { int MB, TR, TC, CC, NT, LDS; };
template <class C> constexpr ConvCfgInfo cfg_info() { return {C::MB, C::TR, C::TC, C::CC, C::NT, C::LDS_BYTES}; }
inline ConvCfgInfo conv_cfg_info(int cfg) {
  switch (cfg) {
    case 0: return cfg_info<ConvC0>();
    case 1: return cfg_info<ConvC1>();
    case 2: return cfg_info<ConvC2>();
    case 3: return cfg_info<ConvC3>();
    case 4: return cfg_info<ConvC4>();
    case 5: return cfg_info<ConvC5>();
    case 6: return cfg_info<ConvC6>();
    case 7: return cfg_info<ConvC7>();
    case 8: return cfg_info<ConvC8>();
    case 9: return cfg_info<ConvC9>();
    case 10: return cfg_info<ConvC10>();
    case 11: return cfg_info<ConvC11>();
    case 12: return cfg_info<ConvC12>();
    case 13: return cfg_info<ConvC13>();
    case 14: return cfg_info<ConvC14>();
    case 15: return cfg_info<ConvC15>();
    case 16: return cfg_info<ConvC16>();
    case 17: return cfg_info<ConvC17>();
    case 18: return cfg_info<ConvC18>();
    default: return {0, 0, 0, 0, 0, 0};
  }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Compute units of the device the CURRENT CALL runs on (256 on an unpartitioned MI355X: 8 XCDs x 32 CUs).  Per thread: the
// C-ABI translation unit sets it at the top of every entry point from the stream's device (hipDeviceGetAttribute, cached per
// device); the emulator keeps the default.
inline int& num_cus() { static thread_local int n = 256; return n; }

// Library default when the caller passes cfg = -1 (overridden per layer by the tuned table on the Python side, which
// holds the measured winners for batch 10).  For every other batch size the choice follows the launch-tail model that
// the sweeps at batch 1 / 5 / 16 / 32 confirmed (profiles/r01_conv_tile_sweep_batch_1_5_16.md): a launch of B equal
// workgroups keeps the CUs busy for (B/CUs)/ceil(B/CUs) of its time (B/CUs when there are fewer workgroups than CUs);
// the 128-channel tile (cfg 11) is ~4.5 % more efficient per FLOP than the 64-channel one (cfg 12) and loses whenever
// halving the tile wins more than that back.
inline double conv_tail_factor(long blocks, int cus) {
  if (blocks < cus) return (double)blocks / cus;
  const long rounds = (blocks + cus - 1) / cus;
  return (double)blocks / ((double)rounds * cus);
}
inline int conv_auto_config(int N, int Cin, int Cout, int H, int W) {
  const int cus = num_cus();
  if (Cin <= 32 && Cout % 64 == 0) return 16;                 // stem: K = 9 * Cin is short, the 3-stage LDS-DMA pipeline wins
  if (Cout % 128 == 0) {
    const long b11 = (long)N * ((H + 3) / 4) * ((W + 31) / 32) * (Cout / 128);
    if (b11 < 2l * cus) return 12;
    return 0.955 * conv_tail_factor(2 * b11, cus) > conv_tail_factor(b11, cus) ? 12 : 11;
  }
  const long b14 = (long)N * ((H + 7) / 8) * ((W + 31) / 32) * ((Cout + 63) / 64);
  return b14 >= 2l * cus ? 14 : 12;                           // 64-channel layers: 8x32-pixel tiles once they fill the chip
}

template <class Cfg, class Launcher>
int launch_conv_cfg(Launcher& L, const Conv3x3Args& a) {
  if (a.Cout % Cfg::MB) TNV3_FAIL(-1, "conv3x3: Cout=%d not a multiple of the config's channel block %d", a.Cout, Cfg::MB);
  if (a.C1 > 0 && (a.C0 % Cfg::CC)) TNV3_FAIL(-1, "conv3x3: C0=%d must be a multiple of %d for a two-source input", a.C0, Cfg::CC);
  const int tilesH = (a.H + Cfg::TR - 1) / Cfg::TR, tilesW = (a.W + Cfg::TC - 1) / Cfg::TC;
  const long nPT = (long)a.N * tilesH * tilesW;
  if (nPT > (1l << 28)) TNV3_FAIL(-1, "conv3x3: too many pixel tiles");
  const int grid = conv_grid_blocks(a.Cout / Cfg::MB, (int)nPT);
  return L.launch(conv3x3_mfma_kernel<Cfg>, grid, Cfg::NT, a);
}

template <class Launcher>
int conv3x3_forward_impl(Launcher& L, const float* src0, const float* src1, const float* wpack, const float* mean,
                         const float* scale, const float* shift, float* dst, int n, int c0, int c1, int cout, int h, int w, int up0,
                         int relu, int cfg, float* dst1 = nullptr, int csplit = 0, int diag = 0, const float* addend = nullptr) {
  if (!src0 || !wpack || !dst) TNV3_FAIL(-1, "conv3x3: null pointer");
  if (n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || h <= 0 || w <= 0) TNV3_FAIL(-1, "conv3x3: non-positive dimension");
  if ((c1 > 0) != (src1 != nullptr)) TNV3_FAIL(-1, "conv3x3: src1 / c1 mismatch");
  if ((scale == nullptr) != (shift == nullptr)) TNV3_FAIL(-1, "conv3x3: scale and shift must both be given or both be NULL");
  if (mean && !scale) TNV3_FAIL(-1, "conv3x3: mean given without scale/shift");
  if (cout % 64) TNV3_FAIL(-1, "conv3x3: Cout=%d must be a multiple of 64", cout);
  if (h >= 8192 || w >= 8192) TNV3_FAIL(-1, "conv3x3: H,W must be < 8192");
  if (c1 > 0 && (c0 % 32)) TNV3_FAIL(-1, "conv3x3: two-source input needs C0 %% 32 == 0 (got %d)", c0);
  if (up0 && ((h | w) & 1)) TNV3_FAIL(-1, "conv3x3: upsampled source needs even H,W");
  if (cfg < 0) cfg = conv_auto_config(n, c0 + c1, cout, h, w);
  if (dst1 && (csplit <= 0 || csplit >= cout)) TNV3_FAIL(-1, "conv3x3: bad output split %d of %d", csplit, cout);
  if (dst1 && addend) TNV3_FAIL(-1, "conv3x3: addend cannot be combined with a split destination");
  const float* zeros = wpack + (size_t)round_up(c0 + c1, 32) * 9 * cout;       // the packed filter's zero tail
  Conv3x3Args a{src0, src1, wpack, zeros, mean, scale, shift, dst, dst1, csplit, n, c0, c1, cout, h, w, up0 ? 1 : 0, relu ? 1 : 0, addend, diag};
  if (diag) {
#ifdef TNV3_DIAG
    switch (cfg) {
      case 10: return launch_conv_cfg<ConvD10>(L, a);
      case 11: return launch_conv_cfg<ConvD11>(L, a);
      case 12: return launch_conv_cfg<ConvD12>(L, a);
      default: TNV3_FAIL(-1, "conv3x3 diagnostics exist for configs 10, 11, 12 only (got %d)", cfg);
    }
#else
    TNV3_FAIL(-1, "conv3x3: the diagnostic twins live in libtnv3_diag.so");
#endif
  }
  switch (cfg) {
    case 0: return launch_conv_cfg<ConvC0>(L, a);
    case 1: return launch_conv_cfg<ConvC1>(L, a);
    case 2: return launch_conv_cfg<ConvC2>(L, a);
    case 3: return launch_conv_cfg<ConvC3>(L, a);
    case 4: return launch_conv_cfg<ConvC4>(L, a);
    case 5: return launch_conv_cfg<ConvC5>(L, a);
    case 6: return launch_conv_cfg<ConvC6>(L, a);
    case 7: return launch_conv_cfg<ConvC7>(L, a);
    case 8: return launch_conv_cfg<ConvC8>(L, a);
    case 9: return launch_conv_cfg<ConvC9>(L, a);
    case 10: return launch_conv_cfg<ConvC10>(L, a);
    case 11: return launch_conv_cfg<ConvC11>(L, a);
    case 12: return launch_conv_cfg<ConvC12>(L, a);
    case 13: return launch_conv_cfg<ConvC13>(L, a);
    case 14: return launch_conv_cfg<ConvC14>(L, a);
    case 15: return launch_conv_cfg<ConvC15>(L, a);
    case 16: return launch_conv_cfg<ConvC16>(L, a);
    case 17: return launch_conv_cfg<ConvC17>(L, a);
    case 18: return launch_conv_cfg<ConvC18>(L, a);
    default: TNV3_FAIL(-1, "conv3x3: unknown config %d", cfg);
  }
}

inline size_t conv3x3_packed_floats(int cout, int cin, int transpose_flip) {
  const int K = transpose_flip ? cout : cin, M = transpose_flip ? cin : cout;
  return (size_t)round_up(K, 32) * 9 * M + kPackZeroTail;
}

template <class Launcher>
int pack_conv3x3_weights_impl(Launcher& L, const float* w, float* wpack, int cout, int cin, int transpose_flip) {
  if (!w || !wpack || cout <= 0 || cin <= 0) TNV3_FAIL(-1, "pack_conv3x3_weights: bad argument");
  const int K = transpose_flip ? cout : cin;
  const size_t total = conv3x3_packed_floats(cout, cin, transpose_flip);
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  return L.launch(pack_conv3x3_weights_kernel, grid, 256, w, wpack, cout, cin, round_up(K, 32), transpose_flip ? 1 : 0);
}

template <class Launcher>
int bn_eval_scale_impl(Launcher& L, const float* g, const float* rv, float eps, float* scale, int c) {
  if (!g || !rv || !scale || c <= 0) TNV3_FAIL(-1, "bn_eval_scale: bad argument");
  return L.launch(bn_eval_scale_kernel, (c + 255) / 256, 256, g, rv, eps, scale, c);
}

template <class Launcher>
int head1x1_impl(Launcher& L, const float* x, const float* w, const float* b, float* y, int n, int c, int l, int hw,
                 int apply_sigmoid) {
  if (!x || !w || !b || !y || n <= 0 || c <= 0 || l <= 0 || hw <= 0) TNV3_FAIL(-1, "head1x1: bad argument");
  if (hw % 4) TNV3_FAIL(-1, "head1x1: H*W=%d must be a multiple of 4", hw);
  const long total = (long)n * (hw / 4);
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  return L.launch(head1x1_sigmoid_kernel<8>, grid, 256, x, w, b, y, n, c, l, hw, apply_sigmoid ? 1 : 0);
}

template <class Launcher>
int maxpool2x2_impl(Launcher& L, const float* x, float* y, long nc, int h, int w) {
  if (!x || !y || nc <= 0 || h <= 0 || w <= 0) TNV3_FAIL(-1, "maxpool2x2: bad argument");
  if ((h % 2) || (w % 4)) TNV3_FAIL(-1, "maxpool2x2: needs H %% 2 == 0 and W %% 4 == 0 (got %dx%d)", h, w);
  const long total = nc * (h / 2) * (w / 4);
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  return L.launch(maxpool2x2_kernel, grid, 256, x, y, nc, h, w);
}


using Conv1dM128 = Conv1dMfmaCfg<2, 2, 2, 2>;   // 128 channels x 8 sequences
using Conv1dM64 = Conv1dMfmaCfg<2, 2, 1, 4>;    //  64 channels x 16 sequences
using Conv1dM32 = Conv1dMfmaCfg<1, 2, 1, 4>;    //  32 channels x 16 sequences
using Conv1dLat = Conv1dMfmaCfg<1, 1, 1, 1, 32, 4>; // small batches: 32 channels x 2 sequences per workgroup, four waves split K

template <class Launcher>
int conv1d_k3_impl(Launcher& L, const float* src0, const float* src1, const float* w, const float* b, float* dst, int n,
                   int c0, int c1, int cout, int l, int src_nlc, int dst_nlc, int act) {
  if (!src0 || !w || !b || !dst || n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || l <= 0) TNV3_FAIL(-1, "conv1d_k3: bad argument");
  if ((c1 > 0) != (src1 != nullptr)) TNV3_FAIL(-1, "conv1d_k3: src1 / c1 mismatch");
  if (act < 0 || act > 2) TNV3_FAIL(-1, "conv1d_k3: unknown activation %d", act);
  constexpr int S = 8, COB = 32, CK = 32, LT = 16;
  Conv1dArgs a{src0, src1, w, b, dst, n, c0, c1, cout, l, src_nlc ? 1 : 0, dst_nlc ? 1 : 0, act, 0, nullptr, 0, 0};
  // The dense layers (L = 16, channel counts in multiples of 8 / 32, channel-major tensors) go to the matrix cores.
  if (l == 16 && !src_nlc && !dst_nlc && cout % 32 == 0 && c0 % 8 == 0 && c1 % 8 == 0 &&
      (((uintptr_t)src0 | (uintptr_t)src1 | (uintptr_t)w) & 15) == 0) {
    auto go = [&](auto cfg) -> int {
      using Cfg = decltype(cfg);
      const long blocks = ((long)n + Cfg::SB - 1) / Cfg::SB * ((cout + Cfg::MB - 1) / Cfg::MB);
      if (blocks > 0x7fffffffl) TNV3_FAIL(-1, "conv1d_k3: batch too large");
      return L.launch(conv1d_k3_mfma_kernel<Cfg>, (int)blocks, Cfg::NT, a);
    };
    // Few sequences: the big tiles would occupy a handful of CUs for K/2 x 64 cycles each (61 us for the 384 -> 128 layer);
    // 32 x 32 tiles with the K range split over four waves put the same work on up to 4 x num_cus() SIMDs.
    if (c0 % 32 == 0 && c1 % 32 == 0 && ((long)n + 1) / 2 * (cout / 32) <= 4l * num_cus()) return go(Conv1dLat{});
    if (cout % 128 == 0) return go(Conv1dM128{});
    if (cout % 64 == 0) return go(Conv1dM64{});
    return go(Conv1dM32{});
  }
  const long gx = (n + S - 1) / S;
  if (gx > 0x7fffffffl) TNV3_FAIL(-1, "conv1d_k3: batch too large");
  return L.launch3(conv1d_k3_kernel<S, COB, CK, LT>, (int)gx, (cout + COB - 1) / COB, (l + LT - 1) / LT, S * COB, a);
}

// ---- InpaintNet as one persistent kernel (kernels/inpaint_fused.h)
inline size_t inpaintnet_packed_floats() { return (size_t)kIfPackedFloats; }

template <class Launcher>
int inpaintnet_pack_impl(Launcher& L, const float* const* w9, const float* const* b9, float* packed) {
  if (!w9 || !b9 || !packed) TNV3_FAIL(-1, "inpaintnet_pack: bad argument");
  InpaintPackArgs a;
  for (int i = 0; i < 9; ++i) {
    if (!w9[i] || !b9[i]) TNV3_FAIL(-1, "inpaintnet_pack: layer %d has a NULL tensor", i);
    a.w[i] = w9[i];
    a.b[i] = b9[i];
  }
  a.packed = packed;
  return L.launch(inpaint_pack_kernel, (kIfPackedFloats + 255) / 256, 256, a);
}

template <class Launcher>
int inpaintnet_fused_forward_impl(Launcher& L, const float* x, const float* m, const float* packed, float* out, int n, int l) {
  if (!x || !m || !packed || !out || n <= 0) TNV3_FAIL(-1, "inpaintnet_fused_forward: bad argument");
  if (l != kIfL) TNV3_FAIL(-1, "inpaintnet_fused_forward: built for sequences of %d positions (got %d)", kIfL, l);
  if (((uintptr_t)packed) & 15) TNV3_FAIL(-1, "inpaintnet_fused_forward: the packed parameters must be 16-byte aligned");
  if (n <= num_cus())                                    // latency form: one sequence per CU on eight waves (two per SIMD)
    return L.launch(inpaintnet_fused_kernel<8>, n, 512, x, m, packed, out, n, (float*)nullptr);
  const int cap = 2 * num_cus();                         // two resident workgroups per CU (59 KB of LDS each); grid-stride beyond
  return L.launch(inpaintnet_fused_kernel<4>, n < cap ? n : cap, 256, x, m, packed, out, n, (float*)nullptr);
}

// ---- InpaintNet training step in three launches (kernels/inpaint_fused_train.h)
inline size_t inpaintnet_packed_t_floats() { return (size_t)kIfStemOff; }
inline size_t inpaintnet_act_floats(int n) { return n <= 0 ? 0 : (size_t)n * kItActCh * kIfL; }
inline size_t inpaintnet_dpre_floats(int n) { return n <= 0 ? 0 : (size_t)n * kItPreCh * kIfL; }
inline size_t inpaintnet_param_floats() { return (size_t)kItParamFloats; }

template <class Launcher>
int inpaintnet_pack_t_impl(Launcher& L, const float* const* w9, float* packed_t) {
  if (!w9 || !packed_t) TNV3_FAIL(-1, "inpaintnet_pack_t: bad argument");
  InpaintPackTArgs a;
  for (int i = 0; i < 7; ++i) {
    if (!w9[i + 1]) TNV3_FAIL(-1, "inpaintnet_pack_t: layer %d has a NULL tensor", i + 1);
    a.w[i] = w9[i + 1];
  }
  a.packed_t = packed_t;
  return L.launch(inpaint_pack_t_kernel, (kIfStemOff + 255) / 256, 256, a);
}

template <class Launcher>
int inpaintnet_fused_forward_train_impl(Launcher& L, const float* x, const float* m, const float* packed, float* out, float* acts, int n, int l) {
  if (!x || !m || !packed || !out || !acts || n <= 0) TNV3_FAIL(-1, "inpaintnet_fused_forward_train: bad argument");
  if (l != kIfL) TNV3_FAIL(-1, "inpaintnet_fused_forward_train: built for sequences of %d positions (got %d)", kIfL, l);
  if (((uintptr_t)packed) & 15) TNV3_FAIL(-1, "inpaintnet_fused_forward_train: the packed parameters must be 16-byte aligned");
  if (n <= num_cus()) return L.launch(inpaintnet_fused_kernel<8>, n, 512, x, m, packed, out, n, acts);
  const int cap = 2 * num_cus();
  return L.launch(inpaintnet_fused_kernel<4>, n < cap ? n : cap, 256, x, m, packed, out, n, acts);
}

template <class Launcher>
int inpaintnet_fused_backward_impl(Launcher& L, const float* x, const float* m, const float* dout, const float* out, const float* acts,
                                   const float* packed, const float* packed_t, float* dpre, float* grads, int n, int l) {
  if (!x || !m || !dout || !out || !acts || !packed || !packed_t || !dpre || !grads || n <= 0) TNV3_FAIL(-1, "inpaintnet_fused_backward: bad argument");
  if (l != kIfL) TNV3_FAIL(-1, "inpaintnet_fused_backward: built for sequences of %d positions (got %d)", kIfL, l);
  if ((((uintptr_t)packed | (uintptr_t)packed_t | (uintptr_t)acts | (uintptr_t)dpre) & 15) != 0)
    TNV3_FAIL(-1, "inpaintnet_fused_backward: packed filters, activations and dPre must be 16-byte aligned");
  const int cap = 2 * num_cus();
  int rc = L.launch(inpaintnet_fused_dgrad_kernel, n < cap ? n : cap, 256, dout, out, acts, packed_t, packed, dpre, n);
  if (rc) return rc;
  return L.launch(inpaintnet_wgrad_all_kernel, kItWgBlocks, 256, x, m, acts, (const float*)dpre, grads, n);
}

template <class Launcher>
int ensemble_frames_impl(Launcher& L, const float* win, int n_local, long s_base, int l, int e, const float* weight, long t0,
                         int n_frames, long num_sample, int sum_order, float* out) {
  if (!win || !weight || !out || n_local <= 0 || l <= 0 || e <= 0 || n_frames <= 0 || num_sample <= 0 || sum_order < 0 || sum_order > 1)
    TNV3_FAIL(-1, "ensemble_frames: bad argument");
  for (long t = t0; t < t0 + n_frames; t += (n_frames > 1 ? n_frames - 1 : 1)) {   // first and last frame bound the need
    const long lo = t - l + 1 > 0 ? t - l + 1 : 0, hi = t < num_sample - 1 ? t : num_sample - 1;
    if (lo <= hi && (lo < s_base || hi >= s_base + n_local))
      TNV3_FAIL(-1, "ensemble_frames: frame %ld needs windows [%ld,%ld] but [%ld,%ld) are resident", t, lo, hi, s_base, s_base + n_local);
  }
  const long total = (long)n_frames * e;
  const int grid = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  return L.launch(ensemble_frames_kernel, grid, 256, win, n_local, s_base, l, e, weight, t0, n_frames, num_sample, sum_order, out);
}

inline size_t peakfind_workspace_bytes(int frames, int h, int w) {
  if (frames <= 0 || h <= 0 || w <= 0) return 0;
  const size_t hw = (size_t)h * w;
  return (size_t)((frames + 7) / 8) * 8 * sizeof(unsigned long long) + (size_t)frames * 5 * hw * sizeof(int);
}

template <class Launcher>
int heatmap_peakfind_impl(Launcher& L, const float* heat, float thr, int tie_last_wins, int32_t* out_bbox, void* ws,
                          size_t ws_bytes, int frames, int h, int w) {
  if (!heat || !out_bbox || !ws || frames <= 0 || h <= 0 || w <= 0) TNV3_FAIL(-1, "heatmap_peakfind: bad argument");
  if ((long)h * w >= (1l << 31) - 2) TNV3_FAIL(-1, "heatmap_peakfind: map too large");
  if (frames > 65535) TNV3_FAIL(-1, "heatmap_peakfind: at most 65535 maps per call");
  if (ws_bytes < peakfind_workspace_bytes(frames, h, w)) TNV3_FAIL(-1, "heatmap_peakfind: workspace too small");
  if (((uintptr_t)ws) & 7) TNV3_FAIL(-1, "heatmap_peakfind: workspace must be 8-byte aligned");
  const size_t hw = (size_t)h * w;
  unsigned long long* best = (unsigned long long*)ws;
  int* label = (int*)(best + ((frames + 7) / 8) * 8);
  int* box = label + (size_t)frames * hw;
  const int gx = (int)((hw + 255) / 256 > 1024 ? 1024 : (hw + 255) / 256);
  int rc;
  if ((rc = L.launch3(ccl_init_kernel, gx, frames, 1, 256, heat, thr, label, box, best, h, w))) return rc;
  if ((rc = L.launch3(ccl_merge_kernel, gx, frames, 1, 256, label, h, w))) return rc;
  if ((rc = L.launch3(ccl_box_kernel, gx, frames, 1, 256, label, box, h, w))) return rc;
  if ((rc = L.launch3(ccl_select_kernel, gx, frames, 1, 256, (const int*)label, (const int*)box, best, h, w, tie_last_wins ? 1 : 0))) return rc;
  return L.launch(ccl_emit_kernel, (frames + 63) / 64, 64, (const int*)box, (const unsigned long long*)best, (int*)out_bbox, frames, h, w,
                  tie_last_wins ? 1 : 0);
}

template <class Launcher>
int heatmap_box_max_impl(Launcher& L, const float* heat, const int32_t* boxes, float* out, int frames, int h, int w) {
  if (!heat || !out || frames <= 0 || h <= 0 || w <= 0) TNV3_FAIL(-1, "heatmap_box_max: bad argument");
  if ((long)h * w >= (1l << 31)) TNV3_FAIL(-1, "heatmap_box_max: map too large");
  return L.launch(heatmap_box_max_kernel, frames, 256, heat, (const int*)boxes, out, h, w);
}

template <class Launcher>
int mfma_probe_impl(Launcher& L, float* out, int blocks, int iters) {
  if (!out || blocks <= 0 || iters <= 0) TNV3_FAIL(-1, "mfma_probe: bad argument");
  return L.launch(mfma_f32_probe_kernel, blocks, 256, out, iters, 0.5f, 0.25f);
}

// ------------------------------------------------------------------------------------------ training entry points
inline int grid_for(long items, int per_block = 256, int cap = 65536) {
  const long g = (items + per_block - 1) / per_block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

inline size_t bn_workspace_bytes(int c) { return c <= 0 ? 0 : (size_t)c * kRedSplit * 2 * sizeof(double) + (size_t)c * 3 * sizeof(float); }

template <class Launcher>
int bn_train_forward_impl(Launcher& L, const float* z, const float* gamma, const float* beta, float* rm, float* rv, float eps,
                          float momentum, float* a, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, int n,
                          int c, int hw) {
  if (!z || !gamma || !beta || !rm || !rv || !a || !save_mean || !save_invstd || !ws || n <= 0 || c <= 0 || hw <= 0)
    TNV3_FAIL(-1, "bn_train_forward: bad argument");
  if (hw % 4) TNV3_FAIL(-1, "bn_train_forward: H*W must be a multiple of 4");
  if (ws_bytes < bn_workspace_bytes(c) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "bn_train_forward: workspace too small / misaligned");
  double* partial = (double*)ws;
  float* scale = (float*)(partial + (size_t)c * kRedSplit * 2);
  int rc;
  if ((rc = L.launch3(bn_stats_partial_kernel, kRedSplit, c, 1, 256, z, partial, n, c, hw))) return rc;
  if ((rc = L.launch(bn_stats_finalize_kernel, (c + 63) / 64, 64, (const double*)partial, gamma, beta, rm, rv, eps, momentum,
                     (long)n * hw, scale, save_mean, save_invstd, c))) return rc;
  return L.launch(bn_apply_relu_kernel, grid_for((long)n * c * (hw / 4)), 256, z, (const float*)save_mean, (const float*)scale, beta, a,
                  (long)n * c, c, hw);
}

// Training-mode BatchNorm whose batch statistics were taken in the producing convolution's epilogue (tnv3_conv3x3_wino_forward_stats):
// tile_stats [C][n_tiles][2] doubles -> fixed-order sums per channel, then exactly bn_train_forward's finalize + apply.
// pixel tiles per channel in the statistics buffer: 4 x 64 pixels (variants 3-5) or 4 x 32 (variant 6)
inline long conv3x3_wino_stats_tiles(int n, int h, int w, int variant = 5) {
  const int pw = variant == 6 ? 32 : 64;
  return (n <= 0 || h % 4 || w % pw) ? 0 : (long)n * (h / 4) * (w / pw);
}

template <class Launcher>
int bn_train_forward_tiles_impl(Launcher& L, const float* z, const double* tile_stats, long n_tiles, const float* gamma, const float* beta,
                                float* rm, float* rv, float eps, float momentum, float* a, float* save_mean, float* save_invstd, void* ws,
                                size_t ws_bytes, int n, int c, int hw, float* pooled = nullptr, int h = 0, int w = 0) {
  // pooled (round 6, tnv3_bn_train_forward_tiles_pool): MaxPool2d(2, 2) of a [n][c][h / 2][w / 2], written by the normalise + ReLU pass itself
  if (!z || !tile_stats || !gamma || !beta || !rm || !rv || !a || !save_mean || !save_invstd || !ws || n <= 0 || c <= 0 || hw <= 0 || n_tiles <= 0)
    TNV3_FAIL(-1, "bn_train_forward_tiles: bad argument");
  if (hw % 4) TNV3_FAIL(-1, "bn_train_forward_tiles: H*W must be a multiple of 4");
  if (pooled && (h <= 0 || w <= 0 || (h & 1) || (w & 3) || (long)h * w != hw || (((uintptr_t)pooled) & 7) || (((uintptr_t)z | (uintptr_t)a) & 15)))
    TNV3_FAIL(-1, "bn_train_forward_tiles_pool: needs H %% 2 == 0, W %% 4 == 0, 16-byte aligned z / a and an 8-byte aligned pooled output");
  if (ws_bytes < bn_workspace_bytes(c) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "bn_train_forward_tiles: workspace too small / misaligned");
  double* partial = (double*)ws;
  float* scale = (float*)(partial + (size_t)c * kRedSplit * 2);
  int rc;
  // (one launch: the tiles' fixed-order sums and the finalize -- bit-identical to bn_tile_stats_reduce_kernel + bn_stats_finalize_kernel)
  if ((rc = L.launch(bn_tile_stats_finalize_kernel, c, 1024, tile_stats, n_tiles, gamma, rm, rv, eps, momentum, (long)n * hw, scale, save_mean,
                     save_invstd))) return rc;
  if (pooled)
    return L.launch(bn_apply_relu_pool_kernel, grid_for((long)n * c * (hw / 8)), 256, z, (const float*)save_mean, (const float*)scale, beta, a, pooled,
                    (long)n * c, c, h, w);
  return L.launch(bn_apply_relu_kernel, grid_for((long)n * c * (hw / 4)), 256, z, (const float*)save_mean, (const float*)scale, beta, a,
                  (long)n * c, c, hw);
}

template <class Launcher>
int bn_relu_backward_impl(Launcher& L, const float* da, const float* a, const float* z, const float* gamma, const float* beta,
                          const float* mean, const float* invstd, float* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                          int n, int c, int hw) {
  if (!da || (!a && !beta) || !z || !gamma || !mean || !invstd || !dz || !dgamma || !dbeta || !ws || n <= 0 || c <= 0 || hw <= 0)
    TNV3_FAIL(-1, "bn_relu_backward: bad argument (a or beta must be given)");
  if (hw % 4) TNV3_FAIL(-1, "bn_relu_backward: H*W must be a multiple of 4");
  if (ws_bytes < bn_workspace_bytes(c) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "bn_relu_backward: workspace too small / misaligned");
  double* partial = (double*)ws;
  float* coef = (float*)(partial + (size_t)c * kRedSplit * 2);
  int rc;
  const bool from_z = a == nullptr;                      // recompute the ReLU mask from z (two tensors per pass instead of three)
  if ((rc = from_z ? L.launch3(bn_relu_bwd_partial_kernel<true>, kRedSplit, c, 1, 256, da, a, z, gamma, beta, mean, invstd, partial, n, c, hw)
                   : L.launch3(bn_relu_bwd_partial_kernel<false>, kRedSplit, c, 1, 256, da, a, z, gamma, beta, mean, invstd, partial, n, c, hw)))
    return rc;
  if ((rc = L.launch(bn_relu_bwd_finalize_kernel, (c + 63) / 64, 64, (const double*)partial, gamma, invstd, (long)n * hw, dgamma,
                     dbeta, coef, c))) return rc;
  return from_z ? L.launch(bn_relu_bwd_apply_kernel<true>, grid_for((long)n * c * (hw / 4)), 256, da, a, z, gamma, beta, mean, invstd,
                           (const float*)coef, dz, (long)n * c, c, hw)
                : L.launch(bn_relu_bwd_apply_kernel<false>, grid_for((long)n * c * (hw / 4)), 256, da, a, z, gamma, beta, mean, invstd,
                           (const float*)coef, dz, (long)n * c, c, hw);
}

// BatchNorm + ReLU backward whose two per-channel sums were taken in the epilogue of the data-gradient launch that produced dA
// (tnv3_conv3x3_wino_dgrad_bnstats): tile_stats [C][n_tiles][2] doubles -> fixed-order sums, then bn_relu_backward's finalize +
// apply -- one pass over (dA, z) instead of two.
template <class Launcher>
int bn_relu_backward_tiles_impl(Launcher& L, const float* da, const float* z, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, const double* tile_stats, long n_tiles, float* dz, float* dgamma, float* dbeta, void* ws,
                                size_t ws_bytes, int n, int c, int hw) {
  if (!da || !beta || !z || !gamma || !mean || !invstd || !tile_stats || !dz || !dgamma || !dbeta || !ws || n <= 0 || c <= 0 || hw <= 0 || n_tiles <= 0)
    TNV3_FAIL(-1, "bn_relu_backward_tiles: bad argument");
  if (hw % 4) TNV3_FAIL(-1, "bn_relu_backward_tiles: H*W must be a multiple of 4");
  if (ws_bytes < bn_workspace_bytes(c) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "bn_relu_backward_tiles: workspace too small / misaligned");
  double* partial = (double*)ws;
  float* coef = (float*)(partial + (size_t)c * kRedSplit * 2);
  int rc;
  if ((rc = L.launch(bn_tile_stats_bwd_finalize_kernel, c, 1024, tile_stats, n_tiles, gamma, invstd, (long)n * hw, dgamma, dbeta, coef, c))) return rc;
  return L.launch(bn_relu_bwd_apply_kernel<true>, grid_for((long)n * c * (hw / 4)), 256, da, (const float*)nullptr, z, gamma, beta, mean, invstd,
                  (const float*)coef, dz, (long)n * c, c, hw);
}

template <class Launcher>
int bn_bwd_consts_impl(Launcher& L, const float* mean, const float* invstd, const float* gamma, const float* beta, float* c4, int c) {
  if (!mean || !invstd || !gamma || !beta || !c4 || c <= 0) TNV3_FAIL(-1, "bn_bwd_consts: bad argument");
  return L.launch(bn_bwd_consts_kernel, (c + 63) / 64, 64, mean, invstd, gamma, beta, c4, c);
}

// ---- Winograd F(2x2, 3x3) form of the plain eval-mode layer (kernels/conv3x3_wino_mfma.h)
using WinoA = WinoCfg<2, 2, 8>;              // 64 channels x 64 tiles (4 x 64 pixels), 256 threads, 8-channel chunks
using WinoSplit = WinoSplitCfg<8>;          // same tile, 512 threads: two wave groups split the 16 transform rows (2 waves / SIMD)
using WinoV3 = WinoV3Cfg<8>;                // same tile and operands as WinoSplit; balanced DMA issue, buffer-descriptor DMA, paired transform
using WinoV4 = WinoV3Cfg<8, 0, 0, 0, 1>;    // + quad operand layouts (filter pack layout 1): 0.5 instead of 2 LDS reads per MFMA
using WinoV5 = WinoV3Cfg<8, 0, 0, 0, 0, 0, 0, 1>;   // persistent workgroups, the chunk pipeline running through the tile boundaries
                                                     // (conv3x3_wino_stream_mfma_kernel); the older waves (group 0) run their MFMAs first
// persistent launch: one workgroup per CU, a whole number of XCD rounds so that a workgroup's tiles all map to its XCD
inline int wino_persistent_grid(int items) { const int g = std::max(8, num_cus() / 8 * 8); return items < g ? items : g; }
// filter pack layout a kernel variant expects (tnv3_conv3x3_wino_layout)
inline int conv3x3_wino_layout(int variant) { return (variant == 6 || variant == 7 || (variant >= 100 && variant < 130)) ? 2 : ((variant == 4 || variant == 47 || variant == 44) ? 1 : 0); }
// kernel variants whose epilogue can emit the BatchNorm batch statistics (tnv3_conv3x3_wino_has_stats)
inline bool conv3x3_wino_has_stats(int variant) { return variant == 3 || variant == 4 || variant == 5 || variant == 6 || variant == 7; }
constexpr int kWinoCinPad = 24;              // filter rows are padded to a multiple of both chunk sizes
constexpr int kWinoDefaultVariant = 5;       // per-call `variant`: 5 streaming persistent kernel, 6 its 128-channel form with the filter operand
                                             // straight from L2 (conv3x3_wino6_mfma.h), 3 WinoV3, 2 WinoSplit, 4 WinoV4 (quad layouts),
                                             // 0 WinoA (one wave / SIMD); -1 = conv3x3_wino_pick(cin, cout)
static_assert(kWinoCinPad == kWinoCinPadK, "conv3x3_wino6_mfma.h carries its own copy of the filter row padding");
// What `variant` -1 means for a layer: by CHANNEL counts only, so that a filter panel packed ahead of the first forward (its
// layout follows the variant) is the one every later call of that layer reads, whatever the image size.
inline int conv3x3_wino_pick(int cin, int cout) {
  if (cin <= WinoV6Cfg<>::CC) return kWinoDefaultVariant;
  // (7, the same kernel with a 64-channel x 64-tile workgroup tile, is parity-green but 2-5 % slower than 5 on every shape,
  //  profiles/r03_wino7_ab.json: the gain of 6 is the 128-row tile -- half the patch transforms per MFMA -- not the operand path)
  return cout % 128 == 0 ? 6 : kWinoDefaultVariant;
}
inline size_t conv3x3_wino_packed_floats(int cin, int cout) {
  if (cin <= 0 || cout <= 0) return 0;
  return (size_t)round_up(cin, kWinoCinPad) * 16 * cout + kPackZeroTail;
}
inline bool conv3x3_wino_supported(int cin, int cout, int h, int w) {
  return cin > 0 && cout > 0 && cout % WinoA::MB == 0 && h % 4 == 0 && w % WinoA::PW == 0;
}

// Packs input channels c_from .. c_from + c_count - 1 of the nn.Conv2d weight w[cout_w][cin_w][3][3]: as the forward filter
// (Cout = cout_w, Cin = c_count) or, transpose_flip, as the data gradient's filter (Cout = c_count, Cin = cout_w).
template <class Launcher>
int conv3x3_wino_pack_view_impl(Launcher& L, const float* w, float* u, int cout_w, int cin_w, int c_from, int c_count, int transpose_flip,
                                int layout = 0) {
  if (!w || !u || cout_w <= 0 || cin_w <= 0 || c_from < 0 || c_count <= 0 || c_from + c_count > cin_w || layout < 0 || layout > 2)
    TNV3_FAIL(-1, "conv3x3_wino_pack: bad argument");
  const int cout = transpose_flip ? c_count : cout_w, cin = transpose_flip ? cout_w : c_count;
  if (layout == 2 && cout % 32) TNV3_FAIL(-1, "conv3x3_wino_pack: layout 2 needs Cout %% 32 == 0 (got %d)", cout);
  const long s_w_co = (long)cin_w * 9, s_w_ci = 9;
  const int cpad = round_up(cin, kWinoCinPad);
  const long total = (long)cpad * 16 * cout + kPackZeroTail;
  return L.launch(conv3x3_wino_pack_kernel, (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256), 256, w + (size_t)c_from * 9, u,
                  cout, cin, cpad, transpose_flip ? s_w_ci : s_w_co, transpose_flip ? s_w_co : s_w_ci, transpose_flip ? 1 : 0, layout);
}
// One launch for a list of panels (the C ABI's tnv3_wino_pack_item: the arguments of conv3x3_wino_pack_view_impl per panel).
struct WinoPackItem { const float* w; float* u; int cout_w, cin_w, c_from, c_count, transpose_flip, layout; };
template <class Launcher>
int conv3x3_wino_pack_multi_impl(Launcher& L, const WinoPackItem* items, int count) {
  if (!items || count < 0) TNV3_FAIL(-1, "conv3x3_wino_pack_multi: bad argument");
  for (int base = 0; base < count; base += kWinoPackMaxItems) {
    WinoPackTable t;
    t.count = count - base < kWinoPackMaxItems ? count - base : kWinoPackMaxItems;
    t.first_block[0] = 0;
    for (int k = 0; k < t.count; ++k) {
      const WinoPackItem& it = items[base + k];
      if (!it.w || !it.u || it.cout_w <= 0 || it.cin_w <= 0 || it.c_from < 0 || it.c_count <= 0 || it.c_from + it.c_count > it.cin_w || it.layout < 0 ||
          it.layout > 4)
        TNV3_FAIL(-1, "conv3x3_wino_pack_multi: bad item %d", base + k);
      const int cout = it.transpose_flip ? it.c_count : it.cout_w, cin = it.transpose_flip ? it.cout_w : it.c_count;
      if (it.layout == 3 && !kTwins) TNV3_FAIL(-1, "conv3x3_wino_pack_multi: layout 3 (the 32x32x2 F(4x4) kernel's panel) left the product library with ABI 6 (item %d)", base + k);
      if ((it.layout == 2 || it.layout == 3) && cout % 32) TNV3_FAIL(-1, "conv3x3_wino_pack_multi: layouts 2 and 3 need Cout %% 32 == 0 (item %d: %d)", base + k, cout);
      if (it.layout == 4 && cout % 16) TNV3_FAIL(-1, "conv3x3_wino_pack_multi: layout 4 needs Cout %% 16 == 0 (item %d: %d)", base + k, cout);
      const long s_w_co = (long)it.cin_w * 9, s_w_ci = 9;
      t.w[k] = it.w + (size_t)it.c_from * 9;
      t.u[k] = it.u;
      t.cout[k] = cout; t.cin[k] = cin; t.cpad[k] = round_up(cin, kWinoCinPad);
      t.s_co[k] = it.transpose_flip ? s_w_ci : s_w_co;
      t.s_ci[k] = it.transpose_flip ? s_w_co : s_w_ci;
      t.flip[k] = it.transpose_flip ? 1 : 0;
      t.layout[k] = it.layout;
      if (it.layout >= 3 && (((uintptr_t)it.u) & 15)) TNV3_FAIL(-1, "conv3x3_wino_pack_multi: an F(4x4) panel must be 16-byte aligned (item %d)", base + k);
      // work items: layout 3 (an F(4x4) panel) one per (32-channel block, chunk, lane) = 36 float4; the F(2x2) layouts one per float
      const long total = it.layout == 4 ? conv3x3_wino43s_pack_items(cout, cin)
                         : it.layout == 3 ? conv3x3_wino43_pack_items(cout, cin) : (long)t.cpad[k] * 16 * cout + kPackZeroTail;
      const long blocks = (total + 255) / 256;
      t.first_block[k + 1] = t.first_block[k] + (int)(blocks > 2048 ? 2048 : blocks);      // four elements per thread at most times 2048 blocks: grid-stride beyond
    }
    for (int k = t.count; k < kWinoPackMaxItems; ++k) {
      t.w[k] = nullptr; t.u[k] = nullptr; t.s_co[k] = t.s_ci[k] = 0; t.cout[k] = t.cin[k] = t.cpad[k] = t.flip[k] = t.layout[k] = 0;
      t.first_block[k + 1] = t.first_block[t.count];
    }
    if (t.count == 0) continue;
    const int rc = L.launch(conv3x3_wino_pack_multi43_kernel, t.first_block[t.count], 256, t);
    if (rc) return rc;
  }
  return 0;
}

template <class Launcher>
int conv3x3_wino_pack_impl(Launcher& L, const float* w, float* u, int cout, int cin, int layout = 0) {
  return conv3x3_wino_pack_view_impl(L, w, u, cout, cin, 0, cin, 0, layout);
}

// ---- Winograd F(4x4, 3x3) form: 36 products per 4x4 output tile.  `variant` 0 = kernels/conv3x3_wino43s_mfma.h (16x16x4 MFMAs, all 36 xi
//      of a block in one wave, write-out in registers; the 128-channel workgroup geometry where Cout % 128 == 0, else the 64-channel one),
//      2 = the same kernel, 64-channel geometry always; 1 = kernels/conv3x3_wino43_mfma.h (32x32x2, four waves per xi block).
//      Variants 0 and 2 read one panel layout, 1 another: pack and run with the same variant.
constexpr int kWino43Variants = 3;

constexpr int kWino43SGrow = 9;              // the 16x16x4 kernel's step schedule (conv3x3_wino43s_kernel<.., GROW, TS>): filter quads of the next step
constexpr int kWino43STs = 10;               // requested at the end of a step; first slot of the patch transform
inline bool conv3x3_wino43_supported(int cin, int cout, int h, int w) {
  return cin > 0 && cout > 0 && cout % Wino43Cfg::MB == 0 && h % 4 == 0 && w % Wino43Cfg::TW == 0;      // (H % 8 == 4: a half-empty last tile row)
}
static_assert(Wino43Cfg::MB == Wino43SCfg<4>::MB && Wino43Cfg::TW == Wino43SCfg<4>::TW && Wino43Cfg::TH == Wino43SCfg<4>::TH, "the F(4x4) kernels share their shape rules");
inline size_t conv3x3_wino43_packed_floats_v(int cin, int cout, int variant) {
  return (variant == 0 || variant == 2) ? conv3x3_wino43s_packed_floats(cin, cout) : variant == 1 ? conv3x3_wino43_packed_floats(cin, cout) : 0;
}
// Panel of input channels c_from .. c_from + c_count - 1 of the nn.Conv2d weight w[cout_w][cin_w][3][3]: the forward filter
// (Cout = cout_w, Cin = c_count) or, transpose_flip, the data gradient's (Cout = c_count, Cin = cout_w).
template <class Launcher>
int conv3x3_wino43_pack_impl(Launcher& L, const float* w, float* u, int cout_w, int cin_w, int c_from, int c_count, int transpose_flip, int variant) {
  if (!w || !u || cout_w <= 0 || cin_w <= 0 || c_from < 0 || c_count <= 0 || c_from + c_count > cin_w) TNV3_FAIL(-1, "conv3x3_wino43_pack: bad argument");
  if (variant < 0 || variant >= kWino43Variants) TNV3_FAIL(-1, "conv3x3_wino43_pack: unknown kernel variant %d", variant);
  if (variant == 1 && !kTwins) TNV3_FAIL(-1, "conv3x3_wino43_pack: kernel variant 1 (32x32x2) is a measurement twin of libtnv3_diag.so since ABI 6 (dispatchable: 0, 2)");
  const bool s16 = variant != 1;
  const int cout = transpose_flip ? c_count : cout_w, cin = transpose_flip ? cout_w : c_count;
  if (cout % (s16 ? 16 : 32)) TNV3_FAIL(-1, "conv3x3_wino43_pack: needs Cout %% %d == 0 (got %d)", s16 ? 16 : 32, cout);
  const long s_w_co = (long)cin_w * 9, s_w_ci = 9;
  if (((uintptr_t)u) & 15) TNV3_FAIL(-1, "conv3x3_wino43_pack: the panel must be 16-byte aligned");
  const long total = s16 ? conv3x3_wino43s_pack_items(cout, cin) : conv3x3_wino43_pack_items(cout, cin);      // one work item per (channel block, chunk, lane)
  const int grid = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  if (s16)
    return L.launch(conv3x3_wino43s_pack_kernel, grid, 256, w + (size_t)c_from * 9, u, cout, cin, transpose_flip ? s_w_ci : s_w_co,
                    transpose_flip ? s_w_co : s_w_ci, transpose_flip ? 1 : 0);
#ifdef TNV3_DIAG
  return L.launch(conv3x3_wino43_pack_kernel, grid, 256, w + (size_t)c_from * 9, u, cout, cin, transpose_flip ? s_w_ci : s_w_co,
                  transpose_flip ? s_w_co : s_w_ci, transpose_flip ? 1 : 0);
#else
  return -1;
#endif
}
// statistics tiles: variants 0 / 2 one per 4 x 64 pixels, variant 1 one per 8 x 64
inline long conv3x3_wino43_stats_tiles(int n, int h, int w, int variant) {
  if (n <= 0 || h % 4 || w % Wino43Cfg::TW || variant < 0 || variant >= kWino43Variants) return 0;
  return variant == 1 ? (long)n * ((h + Wino43Cfg::TH - 1) / Wino43Cfg::TH) * (w / Wino43Cfg::TW) : (long)n * (h / 4) * (w / Wino43Cfg::TW);
}
inline bool conv3x3_wino43s_wide(int cout, int variant) { return variant == 0 && cout % 128 == 0; }      // the 128-channel workgroup geometry
template <class Launcher>
int conv3x3_wino43_forward_impl(Launcher& L, const float* src, const float* u, const float* addend, const float* mean, const float* scale,
                                const float* shift, float* dst, int n, int cin, int cout, int h, int w, int relu, int variant, double* stats = nullptr,
                                float* pool_dst = nullptr, const float* bn_z = nullptr, const float* bn_beta = nullptr) {
  // bn_z (round 6): the launch is a DATA GRADIENT that also takes the previous block's BatchNorm + ReLU backward sums from its write-out
  // (conv3x3_wino43s_kernel<.., STATS = 2>): mean / scale / shift then carry that block's saved mean / invstd / gamma, bn_beta its beta.
  if (!src || !u || !dst || n <= 0) TNV3_FAIL(-1, "conv3x3_wino43: bad argument");
  if (bn_z) {
    if (!stats || !mean || !scale || !shift || !bn_beta || addend || pool_dst || relu || variant == 1)
      TNV3_FAIL(-1, "conv3x3_wino43 (BatchNorm-backward sums): needs the sums' buffer, mean / invstd / gamma / beta, and no addend / pool / ReLU (kernel variants 0 / 2)");
    if (((uintptr_t)bn_z & 15) || ((uintptr_t)bn_beta & 3)) TNV3_FAIL(-1, "conv3x3_wino43 (BatchNorm-backward sums): z must be 16-byte aligned, beta 4-byte");
  }
  if (pool_dst && (variant == 1 || stats)) TNV3_FAIL(-1, "conv3x3_wino43: the pooled second output belongs to kernel variants 0 / 2 without statistics");
  if (pool_dst && (((uintptr_t)pool_dst) & 7)) TNV3_FAIL(-1, "conv3x3_wino43: the pooled output must be 8-byte aligned");
  if (variant < 0 || variant >= kWino43Variants) TNV3_FAIL(-1, "conv3x3_wino43: unknown kernel variant %d", variant);
  if (variant == 1 && !kTwins) TNV3_FAIL(-1, "conv3x3_wino43: kernel variant 1 (32x32x2) is a measurement twin of libtnv3_diag.so since ABI 6 (dispatchable: 0, 2)");
  if (stats && !bn_z && (scale || shift || mean || relu)) TNV3_FAIL(-1, "conv3x3_wino43: the batch-statistics epilogue writes the raw convolution (no affine, no ReLU)");
  if (stats && (((uintptr_t)stats) & 7)) TNV3_FAIL(-1, "conv3x3_wino43: statistics buffer must be 8-byte aligned");
  if (!conv3x3_wino43_supported(cin, cout, h, w))
    TNV3_FAIL(-1, "conv3x3_wino43: needs Cout %% 64 == 0, H %% 4 == 0, W %% 64 == 0 (got %d -> %d, %dx%d)", cin, cout, h, w);
  if ((scale == nullptr) != (shift == nullptr) || (mean && !scale)) TNV3_FAIL(-1, "conv3x3_wino43: inconsistent affine arguments");
  if ((long)cin * h * w * 4 >= (1l << 31) || (long)Wino43Cfg::MB * h * w * 4 >= (1l << 31))
    TNV3_FAIL(-1, "conv3x3_wino43: one sample of the input / 64 output planes must stay below 2 GiB");
  if ((((uintptr_t)u | (uintptr_t)src | (uintptr_t)dst | (uintptr_t)addend) & 15) != 0)
    TNV3_FAIL(-1, "conv3x3_wino43: the panel, the input, the output and the addend must be 16-byte aligned");
  // (the 16x16x4 kernel reads mean / scale / shift as per-lane scalars: views at odd offsets of a flattened parameter buffer are fine;
  //  the 32x32x2 twin reads them as float4)
  if ((((uintptr_t)mean | (uintptr_t)scale | (uintptr_t)shift) & (variant == 1 ? 15 : 3)) != 0)
    TNV3_FAIL(-1, "conv3x3_wino43: mean / scale / shift must be 4-byte aligned");
  if (conv3x3_wino43_packed_floats_v(cin, cout, variant) * 4 >= (1ul << 31)) TNV3_FAIL(-1, "conv3x3_wino43: the filter panel must stay below 2 GiB");
  WinoArgs a{src, u, u, addend, mean, scale, shift, dst, n, cin, cout, h, w, relu ? 1 : 0, stats, bn_z, bn_beta, pool_dst};
  if (variant != 1) {
    const bool wide = conv3x3_wino43s_wide(cout, variant);
    const long npt = wide ? (long)n * (h / 4) * (w / 64) : (long)n * ((h + 7) / 8) * (w / 64);
    if (npt > (1l << 28)) TNV3_FAIL(-1, "conv3x3_wino43: too many pixel tiles");
    const int grid = wino_persistent_grid(conv_grid_blocks(cout / (wide ? 128 : 64), (int)npt));
    if (bn_z) {
      if (wide) return L.launch(conv3x3_wino43s_kernel<8, 2, kWino43SGrow, kWino43STs>, grid, Wino43SBase::NT, a);
      return L.launch(conv3x3_wino43s_kernel<4, 2, kWino43SGrow, kWino43STs>, grid, Wino43SBase::NT, a);
    }
    if (wide) {
      if (stats) return L.launch(conv3x3_wino43s_kernel<8, 1, kWino43SGrow, kWino43STs>, grid, Wino43SBase::NT, a);
      if (pool_dst) return L.launch(conv3x3_wino43s_kernel<8, 0, kWino43SGrow, kWino43STs, 0, 0, 1>, grid, Wino43SBase::NT, a);
      return L.launch(conv3x3_wino43s_kernel<8, 0, kWino43SGrow, kWino43STs>, grid, Wino43SBase::NT, a);
    }
    if (stats) return L.launch(conv3x3_wino43s_kernel<4, 1, kWino43SGrow, kWino43STs>, grid, Wino43SBase::NT, a);
    if (pool_dst) return L.launch(conv3x3_wino43s_kernel<4, 0, kWino43SGrow, kWino43STs, 0, 0, 1>, grid, Wino43SBase::NT, a);
    return L.launch(conv3x3_wino43s_kernel<4, 0, kWino43SGrow, kWino43STs>, grid, Wino43SBase::NT, a);
  }
#ifdef TNV3_DIAG
  const long npt = (long)n * ((h + Wino43Cfg::TH - 1) / Wino43Cfg::TH) * (w / Wino43Cfg::TW);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "conv3x3_wino43: too many pixel tiles");
  const int grid = wino_persistent_grid(conv_grid_blocks(cout / Wino43Cfg::MB, (int)npt));
  // <1, 0>: next tile's raw fill before the write-out, scalar input transform (the two-wide form <1, 1> measured 1-5 % slower on every
  // shape, profiles/r03_wino43_transform_ab.txt; filling after the write-out <0, 0> 1-1.5 % slower)
  if (stats) return L.launch(conv3x3_wino43_kernel<1, 0, 1>, grid, Wino43Cfg::NT, a);
  return L.launch(conv3x3_wino43_kernel<1, 0>, grid, Wino43Cfg::NT, a);
#else
  return -1;
#endif
}

#ifdef TNV3_DIAG
// Timeline / timing twins of the 16x16x4 F(4x4) kernel: the plain forward + [8 waves][8] uint64 of s_memtime totals (prologue, steps,
// write-outs, steps walked, tiles walked) in tl_out.  cbw: 4 / 8 (geometry); grow, ts: the step schedule; mask: conv3x3_wino43s_kernel's
// DG bits (0 = the product kernel's work, results correct).
template <class Launcher>
int conv3x3_wino43s_timeline_impl(Launcher& L, const float* src, const float* u, float* dst, unsigned long long* tl_out, int n, int cin, int cout,
                                  int h, int w, int cbw, int grow, int ts, int mask) {
  if (!src || !u || !dst || !tl_out || n <= 0) TNV3_FAIL(-1, "conv3x3_wino43s_timeline: bad argument");
  if (!conv3x3_wino43_supported(cin, cout, h, w) || (cbw == 8 && cout % 128)) TNV3_FAIL(-1, "conv3x3_wino43s_timeline: unsupported shape");
  if ((long)cin * h * w * 4 >= (1l << 31) || (long)Wino43Cfg::MB * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino43s_timeline: sample too large");
  WinoArgs a{src, u, u, nullptr, nullptr, nullptr, nullptr, dst, n, cin, cout, h, w, 0, reinterpret_cast<double*>(tl_out), nullptr, nullptr};
  const bool wide = cbw == 8;
  const long npt = wide ? (long)n * (h / 4) * (w / 64) : (long)n * ((h + 7) / 8) * (w / 64);
  const int grid = wino_persistent_grid(conv_grid_blocks(cout / (wide ? 128 : 64), (int)npt));
#define TNV3_W43S_TWIN(C, G, T, M) if (cbw == C && grow == G && ts == T && mask == M) return L.launch(conv3x3_wino43s_kernel<C, 0, G, T, 1, M>, grid, Wino43SBase::NT, a)
  TNV3_W43S_TWIN(4, 10, 10, 0); TNV3_W43S_TWIN(4, 5, 3, 0); TNV3_W43S_TWIN(4, 13, 12, 0); TNV3_W43S_TWIN(4, 8, 8, 0); TNV3_W43S_TWIN(4, 12, 4, 0);
  TNV3_W43S_TWIN(8, 10, 10, 0); TNV3_W43S_TWIN(8, 5, 4, 0); TNV3_W43S_TWIN(8, 13, 12, 0); TNV3_W43S_TWIN(8, 10, 28, 0); TNV3_W43S_TWIN(8, 16, 40, 0);
  TNV3_W43S_TWIN(4, 10, 10, 1); TNV3_W43S_TWIN(4, 10, 10, 2); TNV3_W43S_TWIN(4, 10, 10, 3); TNV3_W43S_TWIN(4, 10, 10, 4); TNV3_W43S_TWIN(4, 10, 10, 15);
  TNV3_W43S_TWIN(4, 10, 10, 16);
  TNV3_W43S_TWIN(8, 10, 10, 1); TNV3_W43S_TWIN(8, 10, 10, 2); TNV3_W43S_TWIN(8, 10, 10, 3); TNV3_W43S_TWIN(8, 10, 10, 4); TNV3_W43S_TWIN(8, 10, 10, 15);
  TNV3_W43S_TWIN(8, 10, 10, 16);
#undef TNV3_W43S_TWIN
  TNV3_FAIL(-1, "conv3x3_wino43s_timeline: no twin for geometry %d / grow %d / ts %d / mask %d", cbw, grow, ts, mask);
}

// Timeline twin of the F(4x4) kernel: the plain forward (results correct) + [8 waves][8] uint64 of s_memtime totals per phase in tl_out
template <class Launcher>
int conv3x3_wino43_timeline_impl(Launcher& L, const float* src, const float* u, float* dst, unsigned long long* tl_out, int n, int cin, int cout,
                                 int h, int w, int variant) {
  if (!src || !u || !dst || !tl_out || n <= 0) TNV3_FAIL(-1, "conv3x3_wino43_timeline: bad argument");
  if (!conv3x3_wino43_supported(cin, cout, h, w)) TNV3_FAIL(-1, "conv3x3_wino43_timeline: unsupported shape");
  if ((long)cin * h * w * 4 >= (1l << 31) || (long)Wino43Cfg::MB * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino43_timeline: sample too large");
  WinoArgs a{src, u, u, nullptr, nullptr, nullptr, nullptr, dst, n, cin, cout, h, w, 0, reinterpret_cast<double*>(tl_out), nullptr, nullptr};
  const long npt = (long)n * ((h + Wino43Cfg::TH - 1) / Wino43Cfg::TH) * (w / Wino43Cfg::TW);
  const int grid = wino_persistent_grid(conv_grid_blocks(cout / Wino43Cfg::MB, (int)npt));
  switch (variant) {
    case 1: return L.launch(conv3x3_wino43_kernel<1, 0, 0, 1>, grid, Wino43Cfg::NT, a);
    case 2: return L.launch(conv3x3_wino43_kernel<1, 0, 0, 2>, grid, Wino43Cfg::NT, a);      // no output stores (WRONG results)
    case 3: return L.launch(conv3x3_wino43_kernel<1, 0, 0, 3>, grid, Wino43Cfg::NT, a);      // no LDS exchange (WRONG results)
    case 4: return L.launch(conv3x3_wino43_kernel<1, 0, 0, 4>, grid, Wino43Cfg::NT, a);      // neither
    default: TNV3_FAIL(-1, "conv3x3_wino43_timeline: variant 1..4");
  }
}
#endif

template <class Launcher>
int conv3x3_wino_forward_impl(Launcher& L, const float* src, const float* u, const float* addend, const float* mean, const float* scale,
                              const float* shift, float* dst, int n, int cin, int cout, int h, int w, int relu, int variant = -1,
                              double* stats = nullptr, const float* bn_z = nullptr, const float* bn_c4 = nullptr) {
  if (!src || !u || !dst || n <= 0) TNV3_FAIL(-1, "conv3x3_wino: bad argument");
  if (variant < 0) variant = conv3x3_wino_pick(cin, cout);
  if ((variant == 3 || variant == 7) && !kTwins)
    TNV3_FAIL(-1, "conv3x3_wino: kernel variant %d is a measurement twin of libtnv3_diag.so since ABI 6 (dispatchable: 5, 6; -1 picks between them)", variant);
  if (stats && !conv3x3_wino_has_stats(variant)) TNV3_FAIL(-1, "conv3x3_wino: the batch-statistics epilogue exists in kernel variants 3, 4 and 5");
  if (stats && (scale || shift || mean)) TNV3_FAIL(-1, "conv3x3_wino: the batch-statistics epilogue writes the raw convolution (no affine)");
  if (stats && (((uintptr_t)stats) & 7)) TNV3_FAIL(-1, "conv3x3_wino: statistics buffer must be 8-byte aligned");
  const bool is_v6 = variant == 6 || (variant >= 100 && variant < 120);
  const bool is_v7 = variant == 7 || (variant >= 120 && variant < 130);
  if (is_v7) {     // the 64-channel form of kernel 6: 64 channels x (4 x 64 pixels) per workgroup, filters packed with layout 2
    using V7 = WinoV6Cfg<0, 1, 0, 1, 0, 2>;
    if (stats && (scale || shift || mean)) TNV3_FAIL(-1, "conv3x3_wino: the batch-statistics epilogue writes the raw convolution (no affine)");
    if (h % 4 || cout % V7::MB || w % V7::PW || cin <= V7::CC)
      TNV3_FAIL(-1, "conv3x3_wino (variant 7): needs Cout %% %d == 0, H %% 4 == 0, W %% %d == 0, Cin > %d (got %d -> %d, %dx%d)", V7::MB, V7::PW, V7::CC, cin, cout, h, w);
    if ((scale == nullptr) != (shift == nullptr) || (mean && !scale)) TNV3_FAIL(-1, "conv3x3_wino: inconsistent affine arguments");
    if ((long)cin * h * w * 4 >= (1l << 31) || (long)V7::MB * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino (variant 7): one sample of the input / 64 output planes must stay below 2 GiB");
    if ((bn_z != nullptr) != (bn_c4 != nullptr) || (bn_z && (!stats || addend || relu))) TNV3_FAIL(-1, "conv3x3_wino: inconsistent BatchNorm-backward statistics arguments");
    if ((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)u | (uintptr_t)bn_c4) & 15) != 0)
      TNV3_FAIL(-1, "conv3x3_wino (variant 7): filters / mean / scale / shift must be 16-byte aligned");
    const float* zeros7 = u + (size_t)round_up(cin, kWinoCinPad) * 16 * cout;
    WinoArgs a7{src, u, zeros7, addend, mean, scale, shift, dst, n, cin, cout, h, w, relu ? 1 : 0, stats, bn_z, bn_c4};
    const long npt7 = (long)n * (h / 4) * (w / V7::PW);
    if (npt7 > (1l << 28)) TNV3_FAIL(-1, "conv3x3_wino: too many pixel tiles");
    const int grid7 = wino_persistent_grid(conv_grid_blocks(cout / V7::MB, (int)npt7));
#ifdef TNV3_DIAG
    if (variant == 121) return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 0, 0, 1, 0, 2>>, grid7, V7::NT, a7);     // younger waves' MFMAs first
    if (variant != 7) TNV3_FAIL(-1, "conv3x3_wino: unknown kernel variant %d", variant);
    return L.launch(conv3x3_wino_a128_stream_kernel<V7>, grid7, V7::NT, a7);
#else
    (void)grid7;
    TNV3_FAIL(-1, "conv3x3_wino: unknown kernel variant %d", variant);
#endif
  }
  if (is_v6 ? (h % 4 != 0) : !conv3x3_wino_supported(cin, cout, h, w))
    TNV3_FAIL(-1, "conv3x3_wino: needs Cout %% %d == 0, H %% 4 == 0, W %% %d == 0 (got Cout=%d, %dx%d)", WinoA::MB, WinoA::PW, cout, h, w);
  if ((scale == nullptr) != (shift == nullptr) || (mean && !scale)) TNV3_FAIL(-1, "conv3x3_wino: inconsistent affine arguments");
  const float* zeros = u + (size_t)round_up(cin, kWinoCinPad) * 16 * cout;
  if ((bn_z != nullptr) != (bn_c4 != nullptr) || (bn_z && (!stats || addend || relu)))
    TNV3_FAIL(-1, "conv3x3_wino: the BatchNorm-backward statistics need z AND its constants, the statistics buffer, and no addend / ReLU");
  if (bn_c4 && (((uintptr_t)bn_c4) & 15)) TNV3_FAIL(-1, "conv3x3_wino: the BatchNorm constants must be 16-byte aligned");
  WinoArgs a{src, u, zeros, addend, mean, scale, shift, dst, n, cin, cout, h, w, relu ? 1 : 0, stats, bn_z, bn_c4};
  if (is_v6) {     // 128 channels x (4 x 32 pixels) per workgroup, filters packed with layout 2
    using V6 = WinoV6Cfg<0, 1, 0, 1>;    // production: the older waves (group 0) run their MFMAs first, B operand as 16-byte quads (profiles/r03_wino6_*)
    if (cout % V6::MB || w % V6::PW || cin <= V6::CC)
      TNV3_FAIL(-1, "conv3x3_wino (variant 6): needs Cout %% %d == 0, W %% %d == 0, Cin > %d (got %d -> %d, %dx%d)", V6::MB, V6::PW, V6::CC, cin, cout, h, w);
    if ((long)cin * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino (variant 6): one sample of the input must stay below 2 GiB");
    if ((long)V6::MB * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino (variant 6): 128 output planes must stay below 2 GiB");
    if ((((uintptr_t)a.scale | (uintptr_t)a.shift | (uintptr_t)a.mean | (uintptr_t)a.u) & 15) != 0)
      TNV3_FAIL(-1, "conv3x3_wino (variant 6): filters / mean / scale / shift must be 16-byte aligned");
    const long npt6 = (long)n * (h / 4) * (w / V6::PW);
    if (npt6 > (1l << 28)) TNV3_FAIL(-1, "conv3x3_wino: too many pixel tiles");
    const int grid6 = wino_persistent_grid(conv_grid_blocks(cout / V6::MB, (int)npt6));
#ifdef TNV3_DIAG
    switch (variant) {                // timing twins (wrong results by design): no patch transform / no raw DMA / none of the three / no A loads
      case 100: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<10>>, grid6, V6::NT, a);
      case 101: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<11>>, grid6, V6::NT, a);
      case 103: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<13>>, grid6, V6::NT, a);
      case 104: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<14>>, grid6, V6::NT, a);
      case 107: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<7>>, grid6, V6::NT, a);     // timeline (dst needs N + 1 images)
      // schedule experiments (results correct): 106 the YOUNGER waves (group 1) run their MFMAs first (6 has the older ones first);
      // 108 = 6 + raised priority in that MFMA phase; 109 = 106 + raised priority in group 1's MFMA phase; 105 = timeline of 6, 107 of 106
      case 106: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 0>>, grid6, V6::NT, a);     // the YOUNGER waves first (round 3's first form)
      case 108: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 1, 2>>, grid6, V6::NT, a);
      case 109: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 0, 2>>, grid6, V6::NT, a);
      case 105: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<7, 1>>, grid6, V6::NT, a);
      // 102: 6 with 4-byte B reads (6 reads 16-byte quads: 0.2-2 % faster once its prefetch was fenced);  110 = 102 + A loads behind the
      // MFMA phase;  111 = 6 + A loads behind the MFMA phase;  112 = 111 + priority;  113 = timeline of 111 (110-112 measured slower)
      case 102: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 1, 0, 0, 0>>, grid6, V6::NT, a);     // 6 with 4-byte B reads (round 3's second form)
      case 110: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 1, 0, 0, 1>>, grid6, V6::NT, a);
      case 111: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 1, 0, 1, 1>>, grid6, V6::NT, a);
      case 112: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<0, 1, 2, 1, 1>>, grid6, V6::NT, a);
      case 113: return L.launch(conv3x3_wino_a128_stream_kernel<WinoV6Cfg<7, 1, 0, 1, 1>>, grid6, V6::NT, a);
      default: break;
    }
#endif
    if (variant != 6) TNV3_FAIL(-1, "conv3x3_wino: unknown kernel variant %d", variant);
    return L.launch(conv3x3_wino_a128_stream_kernel<V6>, grid6, V6::NT, a);
  }
  const long npt = (long)n * (h / 4) * (w / WinoA::PW);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "conv3x3_wino: too many pixel tiles");
#ifdef TNV3_DIAG
  switch (variant) {                 // 11..13 / 21..26: timing twins of WinoA / WinoSplit (wrong results by design; libtnv3_diag.so only)
    case 11: return L.launch(conv3x3_wino_mfma_kernel<WinoCfg<2, 2, 8, 1>>, conv_grid_blocks(cout / WinoA::MB, (int)npt), WinoA::NT, a);
    case 12: return L.launch(conv3x3_wino_mfma_kernel<WinoCfg<2, 2, 8, 2>>, conv_grid_blocks(cout / WinoA::MB, (int)npt), WinoA::NT, a);
    case 13: return L.launch(conv3x3_wino_mfma_kernel<WinoCfg<2, 2, 8, 3>>, conv_grid_blocks(cout / WinoA::MB, (int)npt), WinoA::NT, a);
    case 21: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 1>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 22: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 2>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 23: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 3>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 24: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 4>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 25: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 5>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 26: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 6>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 27: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplitCfg<8, 7>>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    default: break;
  }
#endif
  if (variant == 3 || variant == 4 || variant == 5 || variant >= 30) {
    if ((long)cin * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino (variants 3, 4): one sample of the input must stay below 2 GiB");
    if ((long)cout * 16 * 8 * 16 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino (variants 3, 4): Cout too large");
    if ((long)WinoV3::MB * h * w * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wino (variants 3-5): 64 output planes must stay below 2 GiB");
    if ((((uintptr_t)a.scale | (uintptr_t)a.shift | (uintptr_t)a.mean) & 15) != 0)
      TNV3_FAIL(-1, "conv3x3_wino (variants 3-5): mean / scale / shift must be 16-byte aligned");
  }
#ifdef TNV3_DIAG
  // timeline twins (phase totals instead of results) and the measured-and-rejected schedules of kernel 3 (results correct):
  // 37 / 47: timelines of variants 3 / 4;  41: group 1 issues all DMAs after its MFMAs;  51: both groups after their MFMAs;
  // 61 / 71: variant 3 / 41 with raised priority in the DMA / transform phases
  const int grid3 = conv_grid_blocks(cout / WinoV3::MB, (int)npt);
  switch (variant) {
    case 37: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 7>>, grid3, WinoV3::NT, a);
    case 47: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 7, 0, 0, 1>>, grid3, WinoV3::NT, a);
    case 41: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 1>>, grid3, WinoV3::NT, a);
    case 33: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 0, 0, 0, 1>>, grid3, WinoV3::NT, a);     // variant 3, lockstep groups
    case 44: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 0, 0, 1, 1>>, grid3, WinoV3::NT, a);     // variant 4, lockstep groups
    case 38: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 7, 0, 0, 0, 1>>, grid3, WinoV3::NT, a);     // timeline of 33
    case 51: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 2>>, grid3, WinoV3::NT, a);
    case 61: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 0, 1>>, grid3, WinoV3::NT, a);
    case 71: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 1, 1>>, grid3, WinoV3::NT, a);
    // 83 / 85: variants 3 / 5 with the per-tile fixed-cost phases timed (results correct; totals behind the N-th image of dst)
    case 83: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 8>>, grid3, WinoV3::NT, a);
    case 85: if (cin <= WinoV3::CC) break;
             return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 8>>, wino_persistent_grid(grid3), WinoV3::NT, a);
    // 95 ... 99: timing twins of 5 (wrong results): patch transform without its arithmetic / no patch transform / no LDS-DMA in the
    // chunk loop / no operand reads in the MFMA stream / none of the three
    case 95: case 96: case 97: case 98: case 99:
      if (cin <= WinoV3::CC) break;
      if (variant == 95) return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 9>>, wino_persistent_grid(grid3), WinoV3::NT, a);
      if (variant == 96) return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 10>>, wino_persistent_grid(grid3), WinoV3::NT, a);
      if (variant == 97) return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 11>>, wino_persistent_grid(grid3), WinoV3::NT, a);
      if (variant == 98) return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 12>>, wino_persistent_grid(grid3), WinoV3::NT, a);
      return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 13>>, wino_persistent_grid(grid3), WinoV3::NT, a);
    // 86: the first persistent form (per-tile prologue kept, next tile's first DMAs issued before the output transform), timed like 85
    case 86: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 8, 0, 0, 0, 0, 1>>, wino_persistent_grid(grid3), WinoV3::NT, a);
    case 56: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3Cfg<8, 0, 0, 0, 0, 0, 1>>, wino_persistent_grid(grid3), WinoV3::NT, a);
    case 57: if (cin <= WinoV3::CC) break;             // the streaming kernel with the YOUNGER waves (group 1) running their MFMAs first (round 2's order)
             return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV3Cfg<8, 0, 0, 0, 0, 0, 0, 0>>, wino_persistent_grid(grid3), WinoV3::NT, a);
    default: break;
  }
#endif
  switch (variant) {
    case 5:                                      // one chunk per tile cannot stream: the variant-3 kernel takes Cin <= 8
      if (cin <= WinoV5::CC) return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3>, conv_grid_blocks(cout / WinoV3::MB, (int)npt), WinoV3::NT, a);
      return L.launch(conv3x3_wino_stream_mfma_kernel<WinoV5>, wino_persistent_grid(conv_grid_blocks(cout / WinoV5::MB, (int)npt)), WinoV5::NT, a);
    case 3: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV3>, conv_grid_blocks(cout / WinoV3::MB, (int)npt), WinoV3::NT, a);
#ifdef TNV3_DIAG      // measured and rejected generations: A/B twins in libtnv3_diag.so only since ABI 5
    case 4: return L.launch(conv3x3_wino_v3_mfma_kernel<WinoV4>, conv_grid_blocks(cout / WinoV4::MB, (int)npt), WinoV4::NT, a);
    case 2: return L.launch(conv3x3_wino_split_mfma_kernel<WinoSplit>, conv_grid_blocks(cout / WinoSplit::MB, (int)npt), WinoSplit::NT, a);
    case 0: return L.launch(conv3x3_wino_mfma_kernel<WinoA>, conv_grid_blocks(cout / WinoA::MB, (int)npt), WinoA::NT, a);
#else
    case 4: case 2: case 0: TNV3_FAIL(-1, "conv3x3_wino: kernel variant %d is a measurement twin of libtnv3_diag.so since ABI 5 (dispatchable: 5, 6)", variant);
#endif
    default: TNV3_FAIL(-1, "conv3x3_wino: unknown kernel variant %d", variant);
  }
}

// ---- the upsampled half of a decoder-entry layer at the low resolution (kernels/conv_up2x_mfma.h)
using UpA = ConvUp2xCfg<2, 2, 4, 2, 4>;      // 128 channels x (4 x 64) full-res pixels, 512 threads
using UpB = ConvUp2xCfg<2, 1, 4, 2, 4>;      //  64 channels x (4 x 64) full-res pixels, 256 threads
using UpA2 = ConvUp2xCfg<2, 2, 8, 4, 4>;     // 128 channels x (8 x 64) full-res pixels, 1024 threads: half the filter traffic per FLOP
using UpB2 = ConvUp2xCfg<2, 1, 8, 4, 4>;     //  64 channels x (8 x 64) full-res pixels, 512 threads
inline int up2x_chunk(int cout) { return cout % 128 == 0 ? UpA::CC : UpB::CC; }
inline size_t conv_up2x_packed_floats(int c0, int cout) {
  if (c0 <= 0 || cout <= 0) return 0;
  return (size_t)round_up(c0, 8) * 16 * cout;          // padded to the larger chunk so either configuration can read it
}

template <class Launcher>
int pack_up2x_weights_impl(Launcher& L, const float* w, float* wq, int cout, int cin, int c0) {
  if (!w || !wq || cout <= 0 || cin <= 0 || c0 <= 0 || c0 > cin) TNV3_FAIL(-1, "pack_up2x_weights: bad argument");
  const int c0pad = round_up(c0, 8);
  const long total = (long)c0pad * 16 * cout;
  return L.launch(pack_up2x_weights_kernel, (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256), 256, w, wq, cout, cin, c0, c0pad);
}

template <class Launcher>
int conv_up2x_forward_impl(Launcher& L, const float* src, const float* wq, float* dst, int n, int c0, int cout, int hl, int wl, int cfg = -1) {
  if (!src || !wq || !dst || n <= 0 || c0 <= 0 || cout <= 0 || hl <= 0 || wl <= 0) TNV3_FAIL(-1, "conv_up2x: bad argument");
  if (cout % 64) TNV3_FAIL(-1, "conv_up2x: Cout=%d must be a multiple of 64", cout);
  if (hl >= 4096 || wl >= 4096) TNV3_FAIL(-1, "conv_up2x: low-resolution H,W must be < 4096");
  if (((uintptr_t)dst) & 7) TNV3_FAIL(-1, "conv_up2x: destination must be 8-byte aligned");
  ConvUp2xArgs a{src, wq, dst, n, c0, cout, hl, wl};
  auto go = [&](auto cfg) -> int {
    using Cfg = decltype(cfg);
    const long npt = (long)n * ((hl + Cfg::TRL - 1) / Cfg::TRL) * ((wl + 31) / 32);
    if (npt > (1l << 28)) TNV3_FAIL(-1, "conv_up2x: too many pixel tiles");
    return L.launch(conv_up2x_mfma_kernel<Cfg>, conv_grid_blocks(cout / Cfg::MB, (int)npt), Cfg::NT, a);
  };
  if (cfg < 0) cfg = 3;      // measured on the three TrackNet shapes (scripts/up2x_sweep.py): 64 channels x (8 x 64) pixels wins everywhere
  switch (cfg) {
    case 0: if (cout % 128) TNV3_FAIL(-1, "conv_up2x: config 0 needs Cout %% 128 == 0"); return go(UpA{});
    case 1: return go(UpB{});
    case 2: if (cout % 128) TNV3_FAIL(-1, "conv_up2x: config 2 needs Cout %% 128 == 0"); return go(UpA2{});
    case 3: return go(UpB2{});
    default: TNV3_FAIL(-1, "conv_up2x: unknown config %d", cfg);
  }
}

using DUpA = DgradUp2xCfg<2, 2, 4, 1, 2>;    // 128 input channels x (4 x 32) low-res pixels, 512 threads
inline size_t dgrad_up2x_packed_floats(int c0, int cout) {
  if (c0 <= 0 || cout <= 0) return 0;
  return (size_t)round_up(cout, DUpA::CC) * 16 * c0;
}

template <class Launcher>
int pack_dgrad_up2x_weights_impl(Launcher& L, const float* w, float* g, int cout, int cin, int c0) {
  if (!w || !g || cout <= 0 || cin <= 0 || c0 <= 0 || c0 > cin) TNV3_FAIL(-1, "pack_dgrad_up2x_weights: bad argument");
  const int cpad = round_up(cout, DUpA::CC);
  const long total = (long)cpad * 16 * c0;
  return L.launch(pack_dgrad_up2x_weights_kernel, (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256), 256, w, g, cout, cin, c0, cpad);
}

template <class Launcher>
int dgrad_up2x_impl(Launcher& L, const float* dz, const float* g, float* dx_low, int n, int c0, int cout, int hl, int wl) {
  if (!dz || !g || !dx_low || n <= 0 || c0 <= 0 || cout <= 0 || hl <= 0 || wl <= 0) TNV3_FAIL(-1, "dgrad_up2x: bad argument");
  if (c0 % 4) TNV3_FAIL(-1, "dgrad_up2x: C0=%d must be a multiple of 4", c0);
  if (hl >= 4096 || wl >= 4096) TNV3_FAIL(-1, "dgrad_up2x: low-resolution H,W must be < 4096");
  DgradUp2xArgs a{dz, g, dx_low, n, c0, cout, hl, wl};
  const long npt = (long)n * ((hl + DUpA::TRL - 1) / DUpA::TRL) * ((wl + 31) / 32);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "dgrad_up2x: too many pixel tiles");
  return L.launch(dgrad_up2x_mfma_kernel<DUpA>, conv_grid_blocks((c0 + DUpA::MB - 1) / DUpA::MB, (int)npt), DUpA::NT, a);
}

template <class Launcher>
int conv3x3_dgrad_impl(Launcher& L, const float* dz, const float* wpack_t, float* dx0, float* dx1, int n, int cout, int c0,
                       int c1, int h, int w, int cfg) {
  if (c1 < 0 || (c1 > 0) != (dx1 != nullptr)) TNV3_FAIL(-1, "conv3x3_dgrad: dx1 / c1 mismatch");
  return conv3x3_forward_impl(L, dz, (const float*)nullptr, wpack_t, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dx0, n, cout,
                              0, c0 + c1, h, w, 0, 0, cfg, dx1, c1 > 0 ? c0 : 0);
}

using WgradA = WgradCfg<4, 1, 4, 32>;   // 128 co x 32 ci per workgroup, 4x32-pixel K tiles
using WgradB = WgradCfg<2, 2, 4, 32>;   //  64 co x 64 ci per workgroup, 4x32-pixel K tiles

using WgradDmaA = WgradDmaCfg<4, 1>;    // the same blocks, 2x32-pixel K tiles staged by LDS DMA, double-buffered
using WgradDmaB = WgradDmaCfg<2, 2>;
static_assert(WgradDmaA::MB == WgradA::MB && WgradDmaA::CB == WgradA::CB && WgradDmaB::MB == WgradB::MB && WgradDmaB::CB == WgradB::CB,
              "both kernel families share the (co, ci) blocking");

// Per-call `variant`: 0 register-staged kernels (WgradA/B, default: measured 8 % faster), 1 LDS-DMA kernels (WgradDmaA/B).
constexpr size_t kWgradZeroBytes = 1024;     // zero prefix of the workspace: padding source of the LDS-DMA kernels

struct WgradPlan { int use_b, nMB, nCB, splitK, nTiles; };
// cb_a: ci per workgroup of the 128-channel configuration in use (WgradA: 32; the 2x2-window WgradA4: 64)
inline WgradPlan wgrad_plan(int n, int cin, int cout, int h, int w, int cb_a = 32, int variant = 0) {
  WgradPlan p;
  p.use_b = (cout % 128) != 0;
  const int MB = p.use_b ? WgradB::MB : WgradA::MB, CB = p.use_b ? WgradB::CB : cb_a;
  p.nMB = (cout + MB - 1) / MB;
  p.nCB = (cin + CB - 1) / CB;
  const int TR = variant == 1 ? WgradDmaA::TR : (p.use_b ? WgradB::TR : WgradA::TR);
  p.nTiles = n * ((h + TR - 1) / TR) * ((w + 31) / 32);
  // One workgroup is resident per CU (LDS), all workgroups of a launch do the same work, so the launch runs in
  // ceil(workgroups / CUs) rounds of ceil(nTiles / splitK) tiles (+ ~0.6 tile of prologue / slab write each).
  // Pick the split that minimises rounds x tiles; ties go to the smaller split (fewer slabs to write and re-read).
  const int nb = p.nMB * p.nCB;
  const int kNumCU = num_cus();
  int cap = (p.nTiles * TR + 23) / 24;                               // at least ~768 pixels of K per workgroup
  if (cap > 4096) cap = 4096;
  if (cap < 1) cap = 1;
  int best_sk = 1;
  double best = 1e300;
  for (int sk = 1; sk <= cap; ++sk) {
    const long blocks = (long)nb * sk;
    if (sk > 1 && blocks > 16l * kNumCU) break;
    const double cost = (double)((blocks + kNumCU - 1) / kNumCU) * ((double)((p.nTiles + sk - 1) / sk) + 0.6);
    if (cost < best - 1e-9) { best = cost; best_sk = sk; }
  }
  p.splitK = best_sk;
  return p;
}
inline size_t wgrad_workspace_bytes(int n, int c0, int c1, int cout, int h, int w, int variant = 0) {
  if (n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || h <= 0 || w <= 0 || variant < 0 || variant > 1) return 0;
  const WgradPlan p = wgrad_plan(n, c0 + c1, cout, h, w, 32, variant);
  return kWgradZeroBytes + (size_t)p.splitK * cout * (c0 + c1) * 9 * sizeof(float);
}

template <class Launcher>
int conv3x3_wgrad_impl(Launcher& L, const float* src0, const float* src1, const float* dz, float* dw, void* ws, size_t ws_bytes,
                       int n, int c0, int c1, int cout, int h, int w, int up0, int variant = 0) {
  if (!src0 || !dz || !dw || !ws || n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || h <= 0 || w <= 0) TNV3_FAIL(-1, "conv3x3_wgrad: bad argument");
  if (variant < 0) variant = 0;
  if (variant > 1) TNV3_FAIL(-1, "conv3x3_wgrad: unknown kernel variant %d", variant);
  if (variant == 1 && !kTwins) TNV3_FAIL(-1, "conv3x3_wgrad: kernel variant 1 (LDS-DMA staged: 8 %% slower) is a measurement twin of libtnv3_diag.so since ABI 6");
  if ((c1 > 0) != (src1 != nullptr)) TNV3_FAIL(-1, "conv3x3_wgrad: src1 / c1 mismatch");
  if (up0 && ((h | w) & 1)) TNV3_FAIL(-1, "conv3x3_wgrad: upsampled source needs even H,W");
  if (w % 4) TNV3_FAIL(-1, "conv3x3_wgrad: W must be a multiple of 4 (16-byte dZ loads)");
  {
    const int cb = (cout % 128) ? WgradB::CB : WgradA::CB;     // a channel block must come from one source
    if (c1 > 0 && (c0 % cb)) TNV3_FAIL(-1, "conv3x3_wgrad: two-source input needs C0 %% %d == 0 (got %d)", cb, c0);
  }
  if ((long)cout * h * w >= (1l << 31) || (long)(c0 > c1 ? c0 : c1) * h * w >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wgrad: sample too large");
  if (h > 250 * 2 * 1024 || c0 + c1 > 32767) TNV3_FAIL(-1, "conv3x3_wgrad: dimension too large");
  if (ws_bytes < wgrad_workspace_bytes(n, c0, c1, cout, h, w, variant)) TNV3_FAIL(-1, "conv3x3_wgrad: workspace too small");
  const WgradPlan p = wgrad_plan(n, c0 + c1, cout, h, w, 32, variant);
  if (((uintptr_t)ws) & 15) TNV3_FAIL(-1, "conv3x3_wgrad: workspace must be 16-byte aligned");
  float* slabs = (float*)((char*)ws + kWgradZeroBytes);
  WgradArgs a{src0, src1, dz, slabs, n, c0, c1, cout, h, w, up0 ? 1 : 0, p.splitK, (const float*)ws, 0, 0};
  const int grid = p.nMB * p.nCB * p.splitK;
  int rc;
#ifdef TNV3_DIAG
  if (variant == 1) {
    if ((rc = L.launch(fill_zero_kernel, 1, 256, (float*)ws, (int)(kWgradZeroBytes / 4)))) return rc;
    rc = p.use_b ? L.launch(wgrad3x3_dma_kernel<WgradDmaB>, grid, WgradDmaB::NT, a) : L.launch(wgrad3x3_dma_kernel<WgradDmaA>, grid, WgradDmaA::NT, a);
  } else
#endif
  {
    rc = p.use_b ? L.launch(wgrad3x3_mfma_kernel<WgradB>, grid, WgradB::NT, a) : L.launch(wgrad3x3_mfma_kernel<WgradA>, grid, WgradA::NT, a);
  }
  if (rc) return rc;
  const long nel = (long)cout * (c0 + c1) * 9;
  if ((nel & 3) == 0 && (((uintptr_t)ws | (uintptr_t)dw) & 15) == 0)
    return L.launch(sum_partials_vec4_kernel, grid_for(nel / 4, 256, 8192), 256, (const float*)slabs, dw, nel / 4, p.splitK);
  return L.launch(sum_partials_kernel, grid_for(nel, 256, 4096), 256, (const float*)slabs, dw, nel, p.splitK);
}

// ---- the upsampled half of a decoder-entry layer in Winograd form: 9 of the 16 GEMMs (kernels/conv_up2x_wino_mfma.h)
// `variant` -1 / 0 / 1: the 9-GEMM F(2x2) kernel (kernels/conv_up2x_wino_mfma.h; 1 = the other wave group's MFMAs first); 2 = F(4x4, 3x3) with 25
// of the 36 products on the 16x16x4 kernel (kernels/conv3x3_wino43s_mfma.h, MODE 1).  0 / 1 and 2 read different panels.
inline bool conv_up2x_wino_supported(int c0, int cout, int hl, int wl, int variant = -1) {
  if (variant == 2)
    return c0 > 0 && cout > 0 && cout % 64 == 0 && hl > 0 && wl > 0 && hl % 2 == 0 && wl % 32 == 0 && (long)64 * 4 * hl * wl * 4 < (1l << 31) &&
           (long)c0 * hl * wl * 4 < (1l << 31) && conv_up2x_wino43_packed_floats(c0, cout) * 4 < (1ul << 31);
  return variant <= 1 && c0 > ConvUp2xWinoCfg::CC && cout > 0 && cout % 64 == 0 && hl > 0 && wl > 0 && hl % 2 == 0 && wl % ConvUp2xWinoCfg::TW == 0 &&
         (long)64 * 4 * hl * wl * 4 < (1l << 31) && (long)c0 * hl * wl * 4 < (1l << 31) && (long)cout * 9 * 8 * 4 < (1l << 31);
}
inline size_t conv_up2x_wino_packed_floats(int c0, int cout, int variant = -1) {
  if (c0 <= 0 || cout <= 0) return 0;
  if (variant == 2) return conv_up2x_wino43_packed_floats(c0, cout);
  return (size_t)round_up(c0, ConvUp2xWinoCfg::CC) * 9 * cout + kPackZeroTail;
}
template <class Launcher>
int conv_up2x_wino_pack_impl(Launcher& L, const float* w, float* u, int cout, int cin, int c0) {
  if (!w || !u || cout <= 0 || cin <= 0 || c0 <= 0 || c0 > cin) TNV3_FAIL(-1, "conv_up2x_wino_pack: bad argument");
  const int c0pad = round_up(c0, ConvUp2xWinoCfg::CC);
  const long total = (long)c0pad * cout;
  return L.launch(conv_up2x_wino_pack_kernel, (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256), 256, w, u, cout, cin, c0, c0pad);
}
constexpr int kWino43UGrow = 9, kWino43UTs = 6;       // the upsampled half's step schedule (13 operand pairs per 8 channels)
// variant 2's launches (instantiated by the Winograd translation unit only: the kernel is built with that family's compiler flags)
template <class Launcher>
int conv_up2x_wino43_pack_launch(Launcher& L, const float* w, float* u, int cout, int cin, int c0) {
  if (cout % 16 || (((uintptr_t)u) & 15)) TNV3_FAIL(-1, "conv_up2x_wino_pack: variant 2 needs Cout %% 16 == 0 and a 16-byte aligned panel");
  const long items = conv_up2x_wino43_pack_items(cout, c0);
  return L.launch(conv_up2x_wino43_pack_kernel, (int)((items + 255) / 256 > 65535 ? 65535 : (items + 255) / 256), 256, w, u, cout, cin, c0);
}
template <class Launcher>
int conv_up2x_wino43_forward_launch(Launcher& L, const float* src, const float* u, float* dst, int n, int c0, int cout, int hl, int wl) {
  if ((((uintptr_t)dst | (uintptr_t)u) & 15) || (((uintptr_t)src) & 3)) TNV3_FAIL(-1, "conv_up2x_wino: misaligned pointer");
  WinoArgs a{src, u, u, nullptr, nullptr, nullptr, nullptr, dst, n, c0, cout, 2 * hl, 2 * wl, 0, nullptr, nullptr, nullptr, nullptr};
  const bool wide = cout % 128 == 0;
  const long npt = wide ? (long)n * (hl / 2) * (wl / 32) : (long)n * ((hl + 3) / 4) * (wl / 32);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "conv_up2x_wino: too many tiles");
  const int grid = wino_persistent_grid(conv_grid_blocks(cout / (wide ? 128 : 64), (int)npt));
  if (wide) return L.launch(conv3x3_wino43s_kernel<8, 0, kWino43UGrow, kWino43UTs, 0, 0, 0, 1>, grid, Wino43SBase::NT, a);
  return L.launch(conv3x3_wino43s_kernel<4, 0, kWino43UGrow, kWino43UTs, 0, 0, 0, 1>, grid, Wino43SBase::NT, a);
}
template <class Launcher>
int conv_up2x_wino_forward_impl(Launcher& L, const float* src, const float* u, float* dst, int n, int c0, int cout, int hl, int wl, int variant = -1) {
  if (!src || !u || !dst || n <= 0 || variant > 1) TNV3_FAIL(-1, "conv_up2x_wino: bad argument");
  if (!conv_up2x_wino_supported(c0, cout, hl, wl, variant))
    TNV3_FAIL(-1, "conv_up2x_wino: needs C0 > 8, Cout %% 64 == 0, H_low %% 2 == 0, W_low %% 64 == 0 (got %d -> %d, %dx%d)", c0, cout, hl, wl);
  if ((((uintptr_t)dst) & 7) || (((uintptr_t)u | (uintptr_t)src) & 15)) TNV3_FAIL(-1, "conv_up2x_wino: misaligned pointer");
  ConvUp2xWinoArgs a{src, u, dst, n, c0, cout, hl, wl, variant == 1 ? 1 : 0};
  const long npt = (long)n * (hl / 2) * (wl / ConvUp2xWinoCfg::TW);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "conv_up2x_wino: too many tiles");
  return L.launch(conv_up2x_wino_stream_kernel, wino_persistent_grid(conv_grid_blocks(cout / ConvUp2xWinoCfg::MB, (int)npt)), ConvUp2xWinoCfg::NT, a);
}

// ---- data gradient of the upsampled half at the low resolution as one GEMM with K = 9 * Cout (kernels/dgrad_up2x_wino_mfma.h; variant -1 / 0 / 1)
//      or in the 25-of-36 F(4x4) form on the 16x16x4 kernel (kernels/conv3x3_wino43s_mfma.h, MODE 2; variant 2: its own panel)
inline bool dgrad_up2x_wino_supported(int c0, int cout, int hl, int wl, int variant = -1) {
  if (variant == 2)
    return c0 > 0 && c0 % 64 == 0 && cout > 0 && hl > 0 && wl > 0 && hl % 2 == 0 && wl % 32 == 0 && (long)cout * 4 * hl * wl * 4 < (1l << 31) &&
           (long)64 * hl * wl * 4 < (1l << 31) && dgrad_up2x_wino43_packed_floats(c0, cout) * 4 < (1ul << 31);
  return variant <= 1 && c0 > 0 && c0 % DgradUp2xWinoCfg::MB == 0 && cout > DgradUp2xWinoCfg::CC && hl > 0 && wl > 0 && hl % 2 == 0 && wl % DgradUp2xWinoCfg::TWL == 0 &&
         (long)DgradUp2xWinoCfg::CC * 4 * hl * wl * 4 < (1l << 31) && (long)72 * c0 * 4 < (1l << 31);
}
inline size_t dgrad_up2x_wino_packed_floats(int c0, int cout, int variant = -1) {
  if (c0 <= 0 || cout <= 0) return 0;
  if (variant == 2) return dgrad_up2x_wino43_packed_floats(c0, cout);
  return (size_t)round_up(cout, DgradUp2xWinoCfg::CC) * 9 * c0 + kPackZeroTail;
}
// variant 2's launches (instantiated by the Winograd F(4x4) translation unit only, as the forward's)
template <class Launcher>
int dgrad_up2x_wino43_pack_launch(Launcher& L, const float* w, float* u, int cout, int cin, int c0) {
  if (c0 % 16 || (((uintptr_t)u) & 15)) TNV3_FAIL(-1, "dgrad_up2x_wino_pack: variant 2 needs C0 %% 16 == 0 and a 16-byte aligned panel");
  const long items = dgrad_up2x_wino43_pack_items(cout, c0);
  return L.launch(dgrad_up2x_wino43_pack_kernel, (int)((items + 255) / 256 > 65535 ? 65535 : (items + 255) / 256), 256, w, u, cout, cin, c0);
}
template <class Launcher>
int dgrad_up2x_wino43_launch(Launcher& L, const float* dz, const float* u, float* dst, int n, int c0, int cout, int hl, int wl, double* stats = nullptr,
                             const float* bn_z = nullptr, const float* bn_mean = nullptr, const float* bn_invstd = nullptr, const float* bn_gamma = nullptr,
                             const float* bn_beta = nullptr) {
  if ((((uintptr_t)dst) & 7) || (((uintptr_t)u) & 15) || (((uintptr_t)dz) & 3)) TNV3_FAIL(-1, "dgrad_up2x_wino: misaligned pointer");
  // stats (round 6): the launch also takes the BatchNorm + ReLU backward sums of the block whose activation is upsampled (bn_z: its raw output
  // [n][c0][hl][wl]), per tile row of 2 x 32 low-resolution pixels: [c0][n * (hl / 2) * (wl / 32)][2] doubles
  if (stats && (!bn_z || !bn_mean || !bn_invstd || !bn_gamma || !bn_beta || (((uintptr_t)stats | (uintptr_t)bn_z) & 7)))
    TNV3_FAIL(-1, "dgrad_up2x_wino (BatchNorm-backward sums): needs z (8-byte aligned), mean / invstd / gamma / beta and an 8-byte aligned sums buffer");
  // the kernel's "input" is dZ (cout channels at the full resolution), its "output channels" are the layer's c0 upsampled input channels
  WinoArgs a{dz, u, u, nullptr, bn_mean, bn_invstd, bn_gamma, dst, n, cout, c0, 2 * hl, 2 * wl, 0, stats, bn_z, bn_beta, nullptr};
  const bool wide = c0 % 128 == 0;
  const long npt = wide ? (long)n * (hl / 2) * (wl / 32) : (long)n * ((hl + 3) / 4) * (wl / 32);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "dgrad_up2x_wino: too many tiles");
  const int grid = wino_persistent_grid(conv_grid_blocks(c0 / (wide ? 128 : 64), (int)npt));
  if (stats) {
    if (wide) return L.launch(conv3x3_wino43s_kernel<8, 2, kWino43UGrow, kWino43UTs, 0, 0, 0, 2>, grid, Wino43SBase::NT, a);
    return L.launch(conv3x3_wino43s_kernel<4, 2, kWino43UGrow, kWino43UTs, 0, 0, 0, 2>, grid, Wino43SBase::NT, a);
  }
  if (wide) return L.launch(conv3x3_wino43s_kernel<8, 0, kWino43UGrow, kWino43UTs, 0, 0, 0, 2>, grid, Wino43SBase::NT, a);
  return L.launch(conv3x3_wino43s_kernel<4, 0, kWino43UGrow, kWino43UTs, 0, 0, 0, 2>, grid, Wino43SBase::NT, a);
}
template <class Launcher>
int dgrad_up2x_wino_pack_impl(Launcher& L, const float* w, float* u, int cout, int cin, int c0) {
  if (!w || !u || cout <= 0 || cin <= 0 || c0 <= 0 || c0 > cin) TNV3_FAIL(-1, "dgrad_up2x_wino_pack: bad argument");
  const int copad = round_up(cout, DgradUp2xWinoCfg::CC);
  const long total = (long)copad * c0;
  return L.launch(dgrad_up2x_wino_pack_kernel, (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256), 256, w, u, cout, cin, c0, copad);
}
template <class Launcher>
int dgrad_up2x_wino_impl(Launcher& L, const float* dz, const float* u, float* dst, int n, int c0, int cout, int hl, int wl, int variant = -1) {
  if (!dz || !u || !dst || n <= 0 || variant > 1) TNV3_FAIL(-1, "dgrad_up2x_wino: bad argument");
  if (!dgrad_up2x_wino_supported(c0, cout, hl, wl, variant))
    TNV3_FAIL(-1, "dgrad_up2x_wino: needs C0 %% 128 == 0, Cout > 8, H_low %% 2 == 0, W_low %% 32 == 0 (got %d <- %d, %dx%d)", c0, cout, hl, wl);
  if ((((uintptr_t)u | (uintptr_t)dz) & 15) || (((uintptr_t)dst) & 3)) TNV3_FAIL(-1, "dgrad_up2x_wino: misaligned pointer");
  DgradUp2xWinoArgs a{dz, u, dst, n, c0, cout, hl, wl, variant == 1 ? 1 : 0};
  const long npt = (long)n * (hl / 2) * (wl / DgradUp2xWinoCfg::TWL);
  if (npt > (1l << 28)) TNV3_FAIL(-1, "dgrad_up2x_wino: too many tiles");
  return L.launch(dgrad_up2x_wino_stream_kernel, wino_persistent_grid(conv_grid_blocks(c0 / DgradUp2xWinoCfg::MB, (int)npt)), DgradUp2xWinoCfg::NT, a);
}

// ---- weight gradient of a plain layer in Winograd F(2x2, 3x3) form (kernels/wgrad_wino_mfma.h)
// (Cin: any with kernel 5 -- a partial last block of 64 input channels; kernels 0-4 need Cin % 64 == 0)
inline bool wgrad_wino_supported(int cin, int cout, int h, int w) {
  return cin > 0 && cout > 0 && cout % 64 == 0 && h % 2 == 0 && w % 16 == 0;
}
inline int wgrad_wino_fold_blocks(long elements) { const long b = (elements + 15) / 16; return (int)(b > 65536 ? 65536 : b); }   // 16 elements per block
inline int wgrad_wino_splitk(int n, int cin, int cout, int h, int w) {
  const int nb = (cout / 64) * ((cin + 63) / 64);
  const long chunks = (long)n * (h / 2) * (w / 16);
  const int cus = num_cus();
  int best_sk = 1;
  double best = 1e300;
  const long cap = chunks < 4096 ? chunks : 4096;
  for (int sk = 1; sk <= cap; ++sk) {
    const long blocks = (long)nb * sk;
    if (sk > 1 && blocks > 16l * cus) break;
    // one workgroup per CU; ~6 chunk-times of prologue + the 256-register slab write per workgroup
    const double cost = (double)((blocks + cus - 1) / cus) * ((double)((chunks + sk - 1) / sk) + 6.0);
    if (cost < best - 1e-9) { best = cost; best_sk = sk; }
  }
  return best_sk;
}
// ... and in F(4x4, 3x3) form (kernels/wgrad_wino43_mfma.h; kernel variant 8): 64 co x 32 ci per workgroup, a K unit = one strip of 4 x 16 pixels
constexpr int kWgradWino43Variant = 8;
inline bool wgrad_wino43_supported(int cin, int cout, int h, int w) {
  // (an image of either operand stays 4 (W + 1) bytes below 2^31: the X DMA's padding slots sit at offset 2^31 + the strip origin, which may be
  //  that far negative, and must stay beyond the descriptor's range)
  return cin > 0 && cout > 0 && cout % 64 == 0 && h % 4 == 0 && w % 16 == 0 && (long)(cin > cout ? cin : cout) * h * w * 4 + 4l * (w + 1) + 16 < (1l << 31);
}
inline int wgrad_wino43_splitk(int n, int cin, int cout, int h, int w) {
  const int nb = (cout / WgradWino43Cfg::MB) * ((cin + WgradWino43Cfg::CB - 1) / WgradWino43Cfg::CB);
  const long strips = (long)n * (h / 4) * (w / 16);
  const int cus = num_cus();
  int best_sk = 1;
  double best = 1e300;
  const long cap = strips < 4096 ? strips : 4096;
  for (int sk = 1; sk <= cap; ++sk) {
    const long blocks = (long)nb * sk;
    if (sk > 1 && blocks > 16l * cus) break;
    // one workgroup per CU; ~8 step-times of prologue, G^T S G and the slab write per workgroup
    const double cost = (double)((blocks + cus - 1) / cus) * ((double)((strips + sk - 1) / sk) + 8.0);
    if (cost < best - 1e-9) { best = cost; best_sk = sk; }
  }
  return best_sk;
}
inline size_t wgrad_wino_slab_floats(int n, int cin, int cout, int h, int w) {      // either form fits
  size_t f = (size_t)wgrad_wino_splitk(n, cin, cout, h, w) * 16 * cout * cin;
  if (wgrad_wino43_supported(cin, cout, h, w)) {
    const size_t f43 = (size_t)wgrad_wino43_splitk(n, cin, cout, h, w) * 9 * cout * cin;
    if (f43 > f) f = f43;
  }
  return f;
}
inline size_t wgrad_wino_workspace_bytes(int n, int cin, int cout, int h, int w) {
  if (n <= 0 || !wgrad_wino_supported(cin, cout, h, w)) return 0;
  return kWgradZeroBytes + wgrad_wino_slab_floats(n, cin, cout, h, w) * sizeof(float);
}

// The F(2x2) generations (what variant -1 takes where the F(4x4) kernel does not apply: H % 4 != 0):
// 1: two waves per SIMD, wave groups half a period apart; 2-4: 1 with 16-byte operand reads, three raw stages, the Yh transform
// moved into the MFMA phase (0-5 % over 1 per call); 5 / 6: no roles -- every wave streams its MFMAs and transforms the next chunk
// between them (12 % / 8 % over 1 per call on every TrackNet shape, profiles/r03_wgrad_wino5_ab.json); 0: the first kernel.  All
// bit-identical.  The default stays 1 although 5 is the fastest kernel: inside the training step the weight gradients run on the
// side stream BESIDE the main stream's BatchNorm-backward passes, and those HBM-bound passes only co-reside with a workgroup that
// leaves them registers and LDS -- kernel 1 (192 registers, 139 KB) leaves 128 registers per SIMD and 21 KB, kernels 5 / 6 (240
// registers; 160 / 128 KB) 32 registers: the step is 32.3 ms with 1, 32.8 with 5, 32.6 with 6 (profiles/r03_train_wgrad_step_ab.txt).
// A layer whose Cin is not a multiple of 64 (the stem, the last weight gradient of the step: nothing left to run beside it) takes 5.
constexpr int kWgradWinoDefaultVariant = 1;
inline int wgrad_wino_pick(int cin, int variant) { return variant >= 0 ? variant : (cin % 64 ? 5 : kWgradWinoDefaultVariant); }
template <class Launcher>
int launch_wgrad_wino(Launcher& L, const WgradWinoArgs& a, int variant) {
  const int grid = (a.Cout / 64) * ((a.Cin + 63) / 64) * a.splitK;
  variant = wgrad_wino_pick(a.Cin, variant);
#ifndef TNV3_DIAG
  if (variant == 0 || variant == 2 || variant == 3 || variant == 4 || variant == 6 || variant == 7)
    TNV3_FAIL(-1, "conv3x3_wgrad_wino: kernel variant %d is a measurement twin of libtnv3_diag.so since ABI 5 / 6 (dispatchable: 1, 5, 8)", variant);
#endif
  if (a.Cin % 64 && variant != 5 && variant != 6) TNV3_FAIL(-1, "conv3x3_wgrad_wino: kernel variants 1 and 2 need Cin %% 64 == 0 (got %d)", a.Cin);
#ifdef TNV3_DIAG
  if (variant == 0) return L.launch(wgrad_wino_mfma_kernel, grid, WgradWinoCfg::NT, a);
#endif
  if ((long)64 * a.H * a.W * 4 >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wgrad_wino (variants 1-3): 64 channel planes must stay below 2 GiB");
  if (variant == 1) return L.launch(wgrad_wino2_mfma_kernel<0>, grid, WgradWino2Cfg::NT, a);
#ifdef TNV3_DIAG
  if (variant == 2) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<2>>, grid, 512, a);
#endif
  if (variant == 5) return L.launch(wgrad_wino5_mfma_kernel<WgradWino5Cfg<3>>, grid, 512, a);
#ifdef TNV3_DIAG      // measured and rejected generations: A/B twins in libtnv3_diag.so only since ABI 5
  if (variant == 7) return L.launch(wgrad_wino2_mfma_kernel<1>, grid, WgradWino2Cfg::NT, a);     // 1 + the Yh transform in the MFMA phase
  if (variant == 3) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3>>, grid, 512, a);
  if (variant == 4) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 0, 1>>, grid, 512, a);
  if (variant == 6) return L.launch(wgrad_wino5_mfma_kernel<WgradWino5Cfg<2>>, grid, 512, a);      // 128 KB of LDS: small kernels of another stream fit beside it
  if (variant == 101) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 1>>, grid, 512, a);   // no transforms
  if (variant == 102) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 2>>, grid, 512, a);   // no DMA
  if (variant == 103) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 3>>, grid, 512, a);   // no MFMAs
  if (variant == 111) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 1, 1>>, grid, 512, a);   // the split schedule: no transforms
  if (variant == 112) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 2, 1>>, grid, 512, a);   // no DMA
  if (variant == 113) return L.launch(wgrad_wino3_mfma_kernel<WgradWino3Cfg<3, 3, 1>>, grid, 512, a);   // no MFMAs
  if (variant == 121) return L.launch(wgrad_wino5_mfma_kernel<WgradWino5Cfg<3, 1>>, grid, 512, a);   // kernel 4: no transforms
  if (variant == 122) return L.launch(wgrad_wino5_mfma_kernel<WgradWino5Cfg<3, 2>>, grid, 512, a);   // no DMA
  if (variant == 123) return L.launch(wgrad_wino5_mfma_kernel<WgradWino5Cfg<3, 3>>, grid, 512, a);   // no MFMAs
#endif
  TNV3_FAIL(-1, "conv3x3_wgrad_wino: unknown kernel variant %d", variant);
}

template <class Launcher>
int conv3x3_wgrad_wino_impl(Launcher& L, const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int n, int cin, int cout,
                            int h, int w, int variant = -1) {
  if (!x || !dz || !dw || !ws || n <= 0) TNV3_FAIL(-1, "conv3x3_wgrad_wino: bad argument");
  if (!wgrad_wino_supported(cin, cout, h, w))
    TNV3_FAIL(-1, "conv3x3_wgrad_wino: needs Cout %% 64 == 0, H %% 2 == 0, W %% 16 == 0 (got %d -> %d, %dx%d)", cin, cout, h, w);
  if ((long)(cin > cout ? cin : cout) * h * w >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wgrad_wino: sample too large");
  if (((uintptr_t)ws) & 15) TNV3_FAIL(-1, "conv3x3_wgrad_wino: workspace must be 16-byte aligned");
  if (ws_bytes < wgrad_wino_workspace_bytes(n, cin, cout, h, w)) TNV3_FAIL(-1, "conv3x3_wgrad_wino: workspace too small");
  float* slabs = (float*)((char*)ws + kWgradZeroBytes);
  int rc;
  if (variant < 0 && wgrad_wino43_supported(cin, cout, h, w)) variant = kWgradWino43Variant;      // the default wherever it applies
  if (variant == kWgradWino43Variant) {
    if (!wgrad_wino43_supported(cin, cout, h, w))
      TNV3_FAIL(-1, "conv3x3_wgrad_wino (variant 8, F(4x4)): needs Cout %% 64 == 0, H %% 4 == 0, W %% 16 == 0 (got %d -> %d, %dx%d)", cin, cout, h, w);
    const int sk43 = wgrad_wino43_splitk(n, cin, cout, h, w);
    WgradWinoArgs a43{x, dz, (const float*)ws, slabs, n, cin, cout, h, w, sk43};
    if ((rc = L.launch(wgrad_wino43_kernel, (cout / WgradWino43Cfg::MB) * ((cin + WgradWino43Cfg::CB - 1) / WgradWino43Cfg::CB) * sk43, WgradWino43Cfg::NT, a43))) return rc;
    return L.launch(wgrad_wino43_fold_kernel, grid_for((long)9 * cout * cin, 64, 8192), 256, (const float*)slabs, dw, cout, cin, sk43);
  }
#ifdef TNV3_DIAG      // 8000 + switches: twins / candidate schedules of the F(4x4) kernel (WgradWino43Sw; the Timeline bit writes [wave 8][8] uint64 to the workspace's first KB)
  if (variant >= 9000 && variant < 9000 + 1024) {      // round 5's schedule of the same kernel (kernels/wgrad_wino43_r5_mfma.h): the same-session A/B reference
    if (!wgrad_wino43_supported(cin, cout, h, w)) TNV3_FAIL(-1, "conv3x3_wgrad_wino (F(4x4) twins): unsupported shape");
    const int sk43 = wgrad_wino43_splitk(n, cin, cout, h, w);
    WgradWinoArgs a43{x, dz, (const float*)ws, slabs, n, cin, cout, h, w, sk43};
    const int grid = (cout / WgradWino43Cfg::MB) * ((cin + WgradWino43Cfg::CB - 1) / WgradWino43Cfg::CB) * sk43;
    const int sw = variant - 9000;
    if (sw == 0) { if ((rc = L.launch(wgrad_wino43_r5_twin_kernel<0>, grid, WgradWino43Cfg::NT, a43))) return rc; }
    else if (sw == 64) { if ((rc = L.launch(wgrad_wino43_r5_twin_kernel<64>, grid, WgradWino43Cfg::NT, a43))) return rc; }
    else TNV3_FAIL(-1, "conv3x3_wgrad_wino: no round-5 F(4x4) twin with switches %d", sw);
    return L.launch(wgrad_wino43_fold_kernel, grid_for((long)9 * cout * cin, 64, 8192), 256, (const float*)slabs, dw, cout, cin, sk43);
  }
  if (variant >= 8000 && variant < 8000 + 1024) {
    if (!wgrad_wino43_supported(cin, cout, h, w)) TNV3_FAIL(-1, "conv3x3_wgrad_wino (F(4x4) twins): unsupported shape");
    const int sk43 = wgrad_wino43_splitk(n, cin, cout, h, w);
    WgradWinoArgs a43{x, dz, (const float*)ws, slabs, n, cin, cout, h, w, sk43};
    const int grid = (cout / WgradWino43Cfg::MB) * ((cin + WgradWino43Cfg::CB - 1) / WgradWino43Cfg::CB) * sk43;
    const int sw = variant - 8000;
#define TNV3_W43_TWIN(S) if (sw == (S)) { if ((rc = L.launch(wgrad_wino43_twin_kernel<(S)>, grid, WgradWino43Cfg::NT, a43))) return rc; } else
    TNV3_W43_TWIN(0) TNV3_W43_TWIN(64) TNV3_W43_TWIN(1) TNV3_W43_TWIN(2) TNV3_W43_TWIN(3) TNV3_W43_TWIN(4) TNV3_W43_TWIN(8) TNV3_W43_TWIN(12) TNV3_W43_TWIN(16)
    TNV3_W43_TWIN(128) TNV3_W43_TWIN(15) TNV3_W43_TWIN(143) TNV3_W43_TWIN(32) TNV3_W43_TWIN(512) TNV3_W43_TWIN(256) TNV3_W43_TWIN(256 + 32)
    TNV3_W43_TWIN(256 + 512) TNV3_W43_TWIN(64 + 32) TNV3_W43_TWIN(64 + 512) TNV3_W43_TWIN(64 + 256) TNV3_W43_TWIN(64 + 256 + 512)
    TNV3_W43_TWIN(256 + 512 + 4) TNV3_W43_TWIN(256 + 512 + 8) TNV3_W43_TWIN(256 + 512 + 12)
    TNV3_FAIL(-1, "conv3x3_wgrad_wino: no F(4x4) twin with switches %d", sw);
#undef TNV3_W43_TWIN
    return L.launch(wgrad_wino43_fold_kernel, grid_for((long)9 * cout * cin, 64, 8192), 256, (const float*)slabs, dw, cout, cin, sk43);
  }
#endif
  const int sk = wgrad_wino_splitk(n, cin, cout, h, w);
  const bool zero_page = wgrad_wino_pick(cin, variant) == 0;                           // only the first kernel reads its borders from a zero page
  if (zero_page && (rc = L.launch(fill_zero_kernel, 1, 256, (float*)ws, (int)(kWgradZeroBytes / 4)))) return rc;
  WgradWinoArgs a{x, dz, (const float*)ws, slabs, n, cin, cout, h, w, sk};
  if ((rc = launch_wgrad_wino(L, a, variant))) return rc;
  return L.launch(wgrad_wino_fold_kernel, wgrad_wino_fold_blocks((long)cout * cin), 256, (const float*)slabs, dw, cout, cin, sk);
}

// ---- decoder-entry layer: weight gradient with the upsampled channels evaluated at the low resolution (conv_up2x_mfma.h)
using WgradA4 = WgradCfg<4, 2, 4, 32, 4>;   // 2x2 tap window: 128 co x 64 ci, 8 waves (4 taps reuse a staged tile less than 9 do,
                                            // so the ci block is doubled to keep the flops per staged byte)
using WgradB4 = WgradCfg<2, 2, 4, 32, 4>;
struct WgradUpLayout { size_t zp, d4, dwskip, slabs, total; WgradPlan up, skip; int up_wino_sk, up_w43_sk; };
// the upsampled half in the 25-of-36 F(4x4) form (kernels/wgrad_up2x_wino43_mfma.h): any c0, a K unit = one strip of 4 x 16 output pixels
inline bool wgrad_up2x_wino43_supported(int c0, int cout, int hl, int wl) {
  return c0 > 0 && cout > 0 && cout % 64 == 0 && hl > 0 && hl % 2 == 0 && wl > 0 && wl % 8 == 0 &&
         (long)(c0 > 4 * cout ? c0 : 4 * cout) * hl * wl * 4 + 4l * (wl + 1) + 16 < (1l << 31);      // (the DMA's padding slots: see wgrad_wino43_supported)
}
inline int wgrad_up2x_wino43_splitk(int n, int c0, int cout, int hl, int wl) {
  const int nb = (cout / WgradUp2xWino43Cfg::MB) * ((c0 + WgradUp2xWino43Cfg::CB - 1) / WgradUp2xWino43Cfg::CB);
  const long strips = (long)n * (hl / 2) * (wl / 8);
  const int cus = num_cus();
  int best_sk = 1;
  double best = 1e300;
  const long cap = strips < 4096 ? strips : 4096;
  for (int sk = 1; sk <= cap; ++sk) {
    const long blocks = (long)nb * sk;
    if (sk > 1 && blocks > 16l * cus) break;
    const double cost = (double)((blocks + cus - 1) / cus) * ((double)((strips + sk - 1) / sk) + 8.0);      // as wgrad_wino43_splitk
    if (cost < best - 1e-9) { best = cost; best_sk = sk; }
  }
  return best_sk;
}
// the upsampled half in the 9-GEMM Winograd form (kernels/wgrad_wino_mfma.h: wgrad_up2x_wino_mfma_kernel)
inline bool wgrad_up2x_wino_supported(int c0, int cout, int hl, int wl) {
  return c0 > 0 && c0 % WgradUp2xWinoCfg::CB == 0 && cout > 0 && cout % 64 == 0 && hl > 0 && wl > 0 && wl % 8 == 0 &&
         (long)64 * 4 * hl * wl * 4 < (1l << 31) && (long)WgradUp2xWinoCfg::CB * hl * wl * 4 < (1l << 31);
}
inline int wgrad_up2x_wino_splitk(int n, int c0, int cout, int hl, int wl) {
  const int nb = (cout / 64) * (c0 / WgradUp2xWinoCfg::CB);
  const long chunks = (long)n * hl * (wl / 8);
  const int cus = num_cus();
  int best_sk = 1;
  double best = 1e300;
  const long cap = chunks < 4096 ? chunks : 4096;
  for (int sk = 1; sk <= cap; ++sk) {
    const long blocks = (long)nb * sk;
    if (sk > 1 && blocks > 16l * cus) break;
    const double cost = (double)((blocks + cus - 1) / cus) * ((double)((chunks + sk - 1) / sk) + 6.0);   // as wgrad_wino_splitk
    if (cost < best - 1e-9) { best = cost; best_sk = sk; }
  }
  return best_sk;
}
// the skip half (a plain layer of c1 -> cout channels at full resolution) takes the Winograd-form kernel where that one wins
inline bool wgrad_up2x_skip_wino(int c1, int cout, int h, int w) { return c1 % 64 == 0 && wgrad_wino_supported(c1, cout, h, w); }   // 64-multiples: Winograd form
inline size_t align16f(size_t floats) { return (floats + 3) / 4 * 4; }
inline WgradUpLayout wgrad_up2x_layout(int n, int c0, int c1, int cout, int hl, int wl) {
  WgradUpLayout l;
  l.up = wgrad_plan(n, c0, cout, hl, wl, WgradA4::CB, 0);   // both halves use the register-staged family (4-row tiles)
  l.skip = wgrad_plan(n, c1, cout, 2 * hl, 2 * wl, 32, 0);
  size_t off = kWgradZeroBytes / 4;
  l.zp = off;      off += align16f((size_t)4 * n * cout * hl * wl);
  l.d4 = off;      off += align16f((size_t)4 * cout * c0 * 4);
  l.dwskip = off;  off += align16f((size_t)cout * c1 * 9);
  size_t s_up = (size_t)l.up.splitK * cout * c0 * 4, s_skip = (size_t)l.skip.splitK * cout * c1 * 9;
  if (wgrad_up2x_skip_wino(c1, cout, 2 * hl, 2 * wl)) s_skip = wgrad_wino_slab_floats(n, c1, cout, 2 * hl, 2 * wl);
  l.up_wino_sk = wgrad_up2x_wino_supported(c0, cout, hl, wl) ? wgrad_up2x_wino_splitk(n, c0, cout, hl, wl) : 0;
  const size_t s_up9 = (size_t)l.up_wino_sk * 9 * cout * c0;
  if (s_up9 > s_up) s_up = s_up9;
  l.up_w43_sk = wgrad_up2x_wino43_supported(c0, cout, hl, wl) ? wgrad_up2x_wino43_splitk(n, c0, cout, hl, wl) : 0;
  const size_t s_up25 = (size_t)l.up_w43_sk * 9 * cout * c0;
  if (s_up25 > s_up) s_up = s_up25;
  l.slabs = off;   off += align16f(s_up > s_skip ? s_up : s_skip);
  l.total = off * sizeof(float);
  return l;
}
inline size_t wgrad_up2x_workspace_bytes(int n, int c0, int c1, int cout, int hl, int wl) {
  if (n <= 0 || c0 <= 0 || c1 <= 0 || cout <= 0 || hl <= 0 || wl <= 0) return 0;
  return wgrad_up2x_layout(n, c0, c1, cout, hl, wl).total;
}

template <class Launcher>
int conv3x3_wgrad_up2x_impl(Launcher& L, const float* x_low, const float* skip, const float* dz, float* dw, void* ws, size_t ws_bytes,
                            int n, int c0, int c1, int cout, int hl, int wl, int wino_variant = -1, int up_variant = -1) {
  if (!x_low || !skip || !dz || !dw || !ws || n <= 0 || c0 <= 0 || c1 <= 0 || cout <= 0 || hl <= 0 || wl <= 0)
    TNV3_FAIL(-1, "conv3x3_wgrad_up2x: bad argument");
  if (wl % 4) TNV3_FAIL(-1, "conv3x3_wgrad_up2x: the low-resolution width must be a multiple of 4");
  if ((long)cout * 4 * hl * wl >= (1l << 31) || (long)(c0 > c1 ? c0 : c1) * 4 * hl * wl >= (1l << 31)) TNV3_FAIL(-1, "conv3x3_wgrad_up2x: sample too large");
  if (((uintptr_t)ws) & 15) TNV3_FAIL(-1, "conv3x3_wgrad_up2x: workspace must be 16-byte aligned");
  const WgradUpLayout l = wgrad_up2x_layout(n, c0, c1, cout, hl, wl);
  if (ws_bytes < l.total) TNV3_FAIL(-1, "conv3x3_wgrad_up2x: workspace too small");
  float* base = (float*)ws;
  float *zp = base + l.zp, *d4 = base + l.d4, *dwskip = base + l.dwskip, *slabs = base + l.slabs;
  const int h = 2 * hl, w = 2 * wl;
  int rc;
  // up_variant: -1 = the fastest form the shape allows (2, else 1, else 0); 2 = the 25-of-36 F(4x4) form; 1 = the 9-GEMM F(2x2) form;
  // 0 = four 2x2-window launches over the parity images of dz.  An explicit form the shape does not allow falls to the next lower one.
  const bool up25 = (up_variant < 0 || up_variant >= 2) && l.up_w43_sk > 0;
  if (up25) up_variant = 2;
  const long s2d_items = (long)n * cout * h * (w / 4);
  if (!up25 && !(up_variant != 0 && l.up_wino_sk > 0))
    if ((rc = L.launch(space_to_depth2_kernel, grid_for(s2d_items, 256, 32768), 256, dz, zp, (long)n * cout, h, w))) return rc;
  auto reduce = [&](float* out, long nel, int parts) -> int {
    if ((nel & 3) == 0) return L.launch(sum_partials_vec4_kernel, grid_for(nel / 4, 256, 8192), 256, (const float*)slabs, out, nel / 4, parts);
    return L.launch(sum_partials_kernel, grid_for(nel, 256, 4096), 256, (const float*)slabs, out, nel, parts);
  };
  const bool up9 = !up25 && up_variant != 0 && l.up_wino_sk > 0;    // upsampled half: 9-GEMM Winograd form or the four 2x2-window launches
  if (up25) {
    WgradUp2xWinoArgs ua{x_low, dz, slabs, n, c0, cout, hl, wl, l.up_w43_sk};
    const int grid = (cout / WgradUp2xWino43Cfg::MB) * ((c0 + WgradUp2xWino43Cfg::CB - 1) / WgradUp2xWino43Cfg::CB) * l.up_w43_sk;
    if ((rc = L.launch(wgrad_up2x_wino43_kernel, grid, WgradUp2xWino43Cfg::NT, ua))) return rc;
    if ((rc = L.launch(wgrad_wino43_fold_kernel, grid_for((long)9 * cout * c0, 64, 8192), 256, (const float*)slabs, d4, cout, c0, l.up_w43_sk))) return rc;
  }
  if (up9) {
    WgradUp2xWinoArgs ua{x_low, dz, slabs, n, c0, cout, hl, wl, l.up_wino_sk};
    if ((rc = L.launch(wgrad_up2x_wino_mfma_kernel, (cout / 64) * (c0 / WgradUp2xWinoCfg::CB) * l.up_wino_sk, WgradUp2xWinoCfg::NT, ua))) return rc;
    if ((rc = L.launch(wgrad_up2x_wino_fold_kernel, grid_for((long)cout * c0, 256, 4096), 256, (const float*)slabs, d4, cout, c0, l.up_wino_sk))) return rc;
  }
  const size_t img = (size_t)n * cout * hl * wl;
  for (int im = 0; !up9 && !up25 && im < 4; ++im) {                              // parity image (pr, pc) = (im >> 1, im & 1): 2x2 window at (pr, pc)
    WgradArgs a{x_low, (const float*)nullptr, zp + im * img, slabs, n, c0, 0, cout, hl, wl, 0, l.up.splitK, (const float*)ws, im >> 1, im & 1};
    const int grid = l.up.nMB * l.up.nCB * l.up.splitK;
    rc = l.up.use_b ? L.launch(wgrad3x3_mfma_kernel<WgradB4>, grid, WgradB4::NT, a) : L.launch(wgrad3x3_mfma_kernel<WgradA4>, grid, WgradA4::NT, a);
    if (rc) return rc;
    if ((rc = reduce(d4 + (size_t)im * cout * c0 * 4, (long)cout * c0 * 4, l.up.splitK))) return rc;
  }
  if (wgrad_up2x_skip_wino(c1, cout, h, w) && (wino_variant == kWgradWino43Variant || wino_variant < 0) && wgrad_wino43_supported(c1, cout, h, w)) {
    const int sk43 = wgrad_wino43_splitk(n, c1, cout, h, w);
    WgradWinoArgs a43{skip, dz, (const float*)ws, slabs, n, c1, cout, h, w, sk43};
    if ((rc = L.launch(wgrad_wino43_kernel, (cout / WgradWino43Cfg::MB) * ((c1 + WgradWino43Cfg::CB - 1) / WgradWino43Cfg::CB) * sk43, WgradWino43Cfg::NT, a43))) return rc;
    if ((rc = L.launch(wgrad_wino43_fold_kernel, grid_for((long)9 * cout * c1, 64, 8192), 256, (const float*)slabs, dwskip, cout, c1, sk43))) return rc;
  } else if (wgrad_up2x_skip_wino(c1, cout, h, w)) {
    if (wino_variant == kWgradWino43Variant) wino_variant = -1;
    const int sk = wgrad_wino_splitk(n, c1, cout, h, w);
    if (wgrad_wino_pick(c1, wino_variant) == 0)                                        // only the first kernel reads a zero page
      if ((rc = L.launch(fill_zero_kernel, 1, 256, (float*)ws, (int)(kWgradZeroBytes / 4)))) return rc;
    WgradWinoArgs a{skip, dz, (const float*)ws, slabs, n, c1, cout, h, w, sk};
    if ((rc = launch_wgrad_wino(L, a, wino_variant))) return rc;
    if ((rc = L.launch(wgrad_wino_fold_kernel, wgrad_wino_fold_blocks((long)cout * c1), 256, (const float*)slabs, dwskip, cout, c1, sk))) return rc;
  } else {
    WgradArgs a{skip, (const float*)nullptr, dz, slabs, n, c1, 0, cout, h, w, 0, l.skip.splitK, (const float*)ws, 0, 0};
    const int grid = l.skip.nMB * l.skip.nCB * l.skip.splitK;
    rc = l.skip.use_b ? L.launch(wgrad3x3_mfma_kernel<WgradB>, grid, WgradB::NT, a) : L.launch(wgrad3x3_mfma_kernel<WgradA>, grid, WgradA::NT, a);
    if (rc) return rc;
    if ((rc = reduce(dwskip, (long)cout * c1 * 9, l.skip.splitK))) return rc;
  }
  if (up9 || up25) return L.launch(wgrad_up2x_join_kernel, grid_for((long)cout * (c0 + c1) * 9, 256, 8192), 256, (const float*)d4, (const float*)dwskip, dw, cout, c0, c1);
  return L.launch(wgrad_up2x_assemble_kernel, grid_for((long)cout * (c0 + c1) * 9, 256, 8192), 256, (const float*)d4, (const float*)dwskip, dw, cout, c0, c1);
}

inline size_t wbce_workspace_bytes(int n) { return n <= 0 ? 0 : (size_t)n * kRedSplit * sizeof(double); }

template <class Launcher>
int wbce_forward_impl(Launcher& L, const float* p, const float* y, float* out, void* ws, size_t ws_bytes, int n, long per_sample,
                      int reduce) {
  if (!p || !y || !out || !ws || n <= 0 || per_sample <= 0 || n > 65535) TNV3_FAIL(-1, "wbce_forward: bad argument");
  if (ws_bytes < wbce_workspace_bytes(n) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "wbce_forward: workspace too small / misaligned");
  int rc;
  if ((rc = L.launch3(wbce_partial_kernel, kRedSplit, n, 1, 256, p, y, (double*)ws, per_sample))) return rc;
  return L.launch(wbce_finalize_kernel, reduce ? 1 : (n + 63) / 64, 64, (const double*)ws, out, n, per_sample, reduce ? 1 : 0);
}

template <class Launcher>
int wbce_backward_impl(Launcher& L, const float* p, const float* y, const float* upstream, float* dp, int n, long per_sample,
                       int reduce) {
  if (!p || !y || !upstream || !dp || n <= 0 || per_sample <= 0) TNV3_FAIL(-1, "wbce_backward: bad argument");
  const long total = (long)n * per_sample;
  const float inv = reduce ? (float)(1.0 / ((double)n * (double)per_sample)) : (float)(1.0 / (double)per_sample);
  return L.launch(wbce_backward_kernel, grid_for(total), 256, p, y, upstream, reduce ? 0 : 1, inv, dp, per_sample, total);
}

constexpr int kHeadGrid = 1024;
inline size_t head_backward_workspace_bytes(int l) { return l <= 0 ? 0 : (size_t)kHeadGrid * (l * kHeadC + l) * sizeof(float); }

inline size_t head_wbce_workspace_bytes(int n) { return n <= 0 ? 0 : (size_t)n * kHeadLossSplit * sizeof(double); }

// head + sigmoid + WBCELoss in one pass (model.py:71-72 + utils/metric.py:15-20): p and the loss
template <class Launcher>
int head_wbce_forward_impl(Launcher& L, const float* x, const float* w, const float* b, const float* y, float* p, float* loss, void* ws,
                           size_t ws_bytes, int n, int c, int l, int hw, int reduce) {
  if (!x || !w || !b || !y || !p || !loss || !ws || n <= 0 || c <= 0 || l <= 0 || hw <= 0) TNV3_FAIL(-1, "head_wbce_forward: bad argument");
  if (hw % 4) TNV3_FAIL(-1, "head_wbce_forward: H*W=%d must be a multiple of 4", hw);
  if (l > 8) TNV3_FAIL(-1, "head_wbce_forward: at most 8 output maps (got %d)", l);
  if (n > 65535) TNV3_FAIL(-1, "head_wbce_forward: at most 65535 samples per call");
  if (ws_bytes < head_wbce_workspace_bytes(n) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "head_wbce_forward: workspace too small / misaligned");
  int rc;
  if ((rc = L.launch3(head1x1_sigmoid_wbce_kernel<8>, kHeadLossSplit, n, 1, 256, x, w, b, y, p, (double*)ws, c, l, hw))) return rc;
  return L.launch(head_wbce_finalize_kernel, reduce ? 1 : (n + 63) / 64, 64, (const double*)ws, loss, n, (long)l * hw, reduce ? 1 : 0);
}

// `dp_or_y`: dL/dp (upstream == NULL), or the TARGETS y with the loss node's upstream gradient (fused WBCE backward)
template <class Launcher>
int head_backward_impl(Launcher& L, const float* dp, const float* p, const float* a, const float* w, float* da, float* dw, float* db,
                       void* ws, size_t ws_bytes, int n, int l, int hw, const float* upstream = nullptr, int reduce = 1) {
  if (!dp || !p || !a || !w || !da || !dw || !db || !ws || n <= 0 || l <= 0 || hw <= 0) TNV3_FAIL(-1, "head_backward: bad argument");
  if (l > kHeadLMax) TNV3_FAIL(-1, "head_backward: at most %d output maps", kHeadLMax);
  if (ws_bytes < head_backward_workspace_bytes(l)) TNV3_FAIL(-1, "head_backward: workspace too small");
  const long nTiles = (long)n * ((hw + kHeadP - 1) / kHeadP);
  int grid = (int)(nTiles < kHeadGrid ? nTiles : kHeadGrid);
  int rc;
  // the matrix-pipe form (round 6): groups of 16 pixels per wave, 16-byte rows of p / dp / a / dA
  const bool mfma = l <= 8 && hw % 16 == 0 && ((((uintptr_t)dp | (uintptr_t)p | (uintptr_t)a | (uintptr_t)da) & 15) == 0);
  if (mfma) {
    const long wave_groups = ((long)n * (hw / 16) + 3) / 4;
    grid = (int)(wave_groups < kHeadGrid ? wave_groups : kHeadGrid);
    if (upstream) {
      const float inv = reduce ? (float)(1.0 / ((double)n * (double)l * (double)hw)) : (float)(1.0 / ((double)l * (double)hw));
      rc = L.launch(head_backward_mfma_kernel<true>, grid, 256, dp, p, a, w, da, (float*)ws, n, l, hw, upstream, reduce ? 0 : 1, inv);
    } else {
      rc = L.launch(head_backward_mfma_kernel<false>, grid, 256, dp, p, a, w, da, (float*)ws, n, l, hw, (const float*)nullptr, 0, 1.0f);
    }
  } else if (upstream) {
    const float inv = reduce ? (float)(1.0 / ((double)n * (double)l * (double)hw)) : (float)(1.0 / ((double)l * (double)hw));
    rc = L.launch(head_backward_kernel<true>, grid, 256, dp, p, a, w, da, (float*)ws, n, l, hw, upstream, reduce ? 0 : 1, inv);
  } else {
    rc = L.launch(head_backward_kernel<false>, grid, 256, dp, p, a, w, da, (float*)ws, n, l, hw, (const float*)nullptr, 0, 1.0f);
  }
  if (rc) return rc;
  // partial layout per workgroup: [L*64 dW | L db]; dW and db are contiguous slices of one reduction
  const long nel = (long)l * kHeadC + l;
  // reduce into a temporary tail of the workspace is avoided: sum directly into dw / db with two launches
  if ((rc = L.launch(sum_partials_wave_kernel, (int)(((long)l * kHeadC + 3) / 4), 256, (const float*)ws, dw, (long)l * kHeadC, grid, nel, 0l))) return rc;
  return L.launch(sum_partials_wave_kernel, (l + 3) / 4, 256, (const float*)ws, db, (long)l, grid, nel, (long)l * kHeadC);
}

template <class Launcher>
int maxpool2x2_backward_add_impl(Launcher& L, const float* x, const float* dpool, const float* dskip, float* dx, long nc, int h, int w) {
  if (!x || !dpool || !dx || nc <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) TNV3_FAIL(-1, "maxpool2x2_backward_add: bad argument");
  return L.launch(maxpool2x2_bwd_add_kernel, grid_for(nc * (h / 2) * (w / 2)), 256, x, dpool, dskip, dx, nc, h, w);
}

// Slices per channel of the sums maxpool2x2_backward_add_bnstats leaves (its tile_stats is [c][slices][2] doubles): a workgroup of 256
// threads walks a slice, a thread takes 2 x 4 pixels per iteration -- about four iterations per thread, at most kRedSplit slices.
inline int maxpool2x2_bnstats_slices(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 3)) return 0;
  const long units = (long)n * (h / 2) * (w / 4);
  const long s = (units + 1023) / 1024;
  return (int)(s < 1 ? 1 : (s > kRedSplit ? kRedSplit : s));
}
template <class Launcher>
int maxpool2x2_backward_add_bnstats_impl(Launcher& L, const float* z, const float* dpool, const float* dskip, float* dx, const float* mean,
                                         const float* invstd, const float* gamma, const float* beta, double* tile_stats, int n, int c, int h, int w) {
  if (!z || !dpool || !dx || !mean || !invstd || !gamma || !beta || !tile_stats || n <= 0 || c <= 0)
    TNV3_FAIL(-1, "maxpool2x2_backward_add_bnstats: bad argument");
  const int slices = maxpool2x2_bnstats_slices(n, h, w);
  if (!slices) TNV3_FAIL(-1, "maxpool2x2_backward_add_bnstats: needs H %% 2 == 0 and W %% 4 == 0 (got %dx%d)", h, w);
  if ((((uintptr_t)z | (uintptr_t)dskip | (uintptr_t)dx) & 15) || (((uintptr_t)dpool | (uintptr_t)tile_stats) & 7))
    TNV3_FAIL(-1, "maxpool2x2_backward_add_bnstats: z / dskip / dx must be 16-byte aligned, dpool and the sums 8-byte");
  if (c > 65535) TNV3_FAIL(-1, "maxpool2x2_backward_add_bnstats: too many channels");
  return L.launch3(maxpool2x2_bwd_add_bnsums_kernel, slices, c, 1, 256, z, dpool, dskip, dx, mean, invstd, gamma, beta, tile_stats, n, c, h, w);
}

template <class Launcher>
int upsample2x_backward_impl(Launcher& L, const float* d_hi, float* d_lo, long nc, int hl, int wl) {
  if (!d_hi || !d_lo || nc <= 0 || hl <= 0 || wl <= 0) TNV3_FAIL(-1, "upsample2x_backward: bad argument");
  return L.launch(upsample2x_bwd_kernel, grid_for(nc * hl * wl), 256, d_hi, d_lo, nc, hl, wl);
}

template <class Launcher>
int mixup_impl(Launcher& L, const float* x, const float* lam, const int32_t* perm, float* out, int n, long per_sample) {
  if (!x || !lam || !perm || !out || n <= 0 || per_sample <= 0 || (per_sample % 4)) TNV3_FAIL(-1, "mixup: bad argument");
  return L.launch(mixup_kernel, grid_for((long)n * (per_sample / 4)), 256, x, lam, (const int*)perm, out, n, per_sample);
}

// ---- optimiser step and mixup draws on the device (kernels/optim.h)
inline int opt_fill_table(OptTensorTable& t, float* const* params, float* const* grads, float* const* s0, float* const* s1,
                          const long* numel, int first, int count) {
  t.count = count;
  int chunks = 0;
  for (int i = 0; i < count; ++i) {
    t.param[i] = params ? params[first + i] : nullptr;
    t.grad[i] = grads[first + i];
    t.s0[i] = s0 ? s0[first + i] : nullptr;
    t.s1[i] = s1 ? s1[first + i] : nullptr;
    t.numel[i] = numel[first + i];
    t.first_chunk[i] = chunks;
    const long c = (numel[first + i] + kOptChunk - 1) / kOptChunk;
    if (c > (1l << 30) - chunks) return -1;
    chunks += (int)c;
  }
  t.first_chunk[count] = chunks;
  for (int i = count; i < kOptMaxTensors; ++i) { t.param[i] = t.grad[i] = t.s0[i] = t.s1[i] = nullptr; t.numel[i] = 0; t.first_chunk[i + 1] = chunks; }
  return chunks;
}

inline size_t grad_norm_workspace_bytes(int count) {
  return count <= 0 ? 0 : (size_t)((count + kOptMaxTensors - 1) / kOptMaxTensors) * kNormSplit * sizeof(double);
}

template <class Launcher>
int grad_norm_impl(Launcher& L, float* const* grads, const long* numel, int count, float max_norm, float* out2, void* ws, size_t ws_bytes) {
  if (!grads || !numel || count <= 0 || !out2 || !ws) TNV3_FAIL(-1, "grad_norm: bad argument");
  if (ws_bytes < grad_norm_workspace_bytes(count) || (((uintptr_t)ws) & 7)) TNV3_FAIL(-1, "grad_norm: workspace too small / misaligned");
  for (int i = 0; i < count; ++i)
    if (!grads[i] || numel[i] <= 0) TNV3_FAIL(-1, "grad_norm: tensor %d is empty or NULL", i);
  double* partial = (double*)ws;
  const int groups = (count + kOptMaxTensors - 1) / kOptMaxTensors;
  int rc;
  for (int gidx = 0; gidx < groups; ++gidx) {
    OptTensorTable t;
    const int first = gidx * kOptMaxTensors, c = count - first < kOptMaxTensors ? count - first : kOptMaxTensors;
    if (opt_fill_table(t, nullptr, grads, nullptr, nullptr, numel, first, c) < 0) TNV3_FAIL(-1, "grad_norm: too many elements");
    if ((rc = L.launch(grad_sumsq_partial_kernel, kNormSplit, 256, t, partial + (size_t)gidx * kNormSplit))) return rc;
  }
  return L.launch(grad_norm_finalize_kernel, 1, 64, (const double*)partial, groups * kNormSplit, max_norm, out2);
}

template <class Launcher>
int adam_step_impl(Launcher& L, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const long* numel, int count, double lr, double beta1, double beta2, double eps, double weight_decay, long step,
                   const float* clip_coef, int zero_grad) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || count <= 0) TNV3_FAIL(-1, "adam_step: bad argument");
  if (step < 1) TNV3_FAIL(-1, "adam_step: step counts from 1 (got %ld)", step);
  if (!(lr >= 0.0) || !(eps >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(weight_decay >= 0.0))
    TNV3_FAIL(-1, "adam_step: invalid hyper-parameter");
  if (!(1.0 - beta1 < 0.5)) TNV3_FAIL(-1, "adam_step: beta1 <= 0.5 is not supported (lerp's other branch)");
  for (int i = 0; i < count; ++i)
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] <= 0) TNV3_FAIL(-1, "adam_step: tensor %d is empty or NULL", i);
  // torch/optim/adam.py (_multi_tensor_adam, non-capturable): python floats, i.e. doubles, rounded to fp32 at the kernel boundary
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  int rc;
  for (int first = 0; first < count; first += kOptMaxTensors) {
    OptTensorTable t;
    const int c = count - first < kOptMaxTensors ? count - first : kOptMaxTensors;
    const int chunks = opt_fill_table(t, params, grads, exp_avg, exp_avg_sq, numel, first, c);
    if (chunks < 0) TNV3_FAIL(-1, "adam_step: too many elements");
    const AdamScalars a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), -step_size, bc2_sqrt, (float)eps, (float)weight_decay};
    if ((rc = L.launch(adam_multi_kernel, chunks, 256, t, a, clip_coef, zero_grad ? 1 : 0))) return rc;
  }
  return 0;
}

template <class Launcher>
int sgd_step_impl(Launcher& L, float* const* params, float* const* grads, float* const* momentum_buf, const long* numel, int count,
                  double lr, double momentum, double weight_decay, int first_step, const float* clip_coef, int zero_grad) {
  if (!params || !grads || !numel || count <= 0) TNV3_FAIL(-1, "sgd_step: bad argument");
  if (momentum != 0.0 && !momentum_buf) TNV3_FAIL(-1, "sgd_step: momentum needs its buffers");
  for (int i = 0; i < count; ++i)
    if (!params[i] || !grads[i] || numel[i] <= 0 || (momentum != 0.0 && !momentum_buf[i])) TNV3_FAIL(-1, "sgd_step: tensor %d is empty or NULL", i);
  int rc;
  for (int first = 0; first < count; first += kOptMaxTensors) {
    OptTensorTable t;
    const int c = count - first < kOptMaxTensors ? count - first : kOptMaxTensors;
    const int chunks = opt_fill_table(t, params, grads, momentum != 0.0 ? momentum_buf : nullptr, nullptr, numel, first, c);
    if (chunks < 0) TNV3_FAIL(-1, "sgd_step: too many elements");
    if ((rc = L.launch(sgd_multi_kernel, chunks, 256, t, (float)lr, (float)momentum, (float)weight_decay, first_step ? 1 : 0, clip_coef,
                       zero_grad ? 1 : 0))) return rc;
  }
  return 0;
}

template <class Launcher>
int mixup_draw_impl(Launcher& L, float* lam, int32_t* perm, int n, float alpha, unsigned long long seed, unsigned long long step) {
  if (!lam || !perm || n <= 0 || n > 65536 || !(alpha > 0.0f)) TNV3_FAIL(-1, "mixup_draw: bad argument (0 < n <= 65536, alpha > 0)");
  return L.launch(mixup_draw_kernel, 1, 256, lam, (int*)perm, n, alpha, seed, step);
}

// ---- frame preprocessing
template <class Launcher>
int resample_bicubic_u8_impl(Launcher& L, const unsigned char* src, unsigned char* tmp, float* dst_f32, unsigned char* dst_u8,
                             const int* xmin, const int* xcnt, const int* kkx, int ksx, const int* ymin, const int* ycnt,
                             const int* kky, int ksy, const float* lut, int frames, int h, int w, int c, int oh, int ow) {
  if (!src || !tmp || (!dst_f32 && !dst_u8) || !xmin || !xcnt || !kkx || !ymin || !ycnt || !kky || frames <= 0 || h <= 0 || w <= 0 ||
      c <= 0 || oh <= 0 || ow <= 0 || ksx <= 0 || ksy <= 0)
    TNV3_FAIL(-1, "resample_bicubic_u8: bad argument");
  if (dst_f32 && !lut) TNV3_FAIL(-1, "resample_bicubic_u8: fp32 output needs the 256-entry lookup table");
  if ((long)w * c > kResampleMaxRowBytes) TNV3_FAIL(-1, "resample_bicubic_u8: source row of %ld bytes exceeds %d", (long)w * c, kResampleMaxRowBytes);
  if ((long)frames * h >= (1l << 31) || (long)frames * oh >= (1l << 31)) TNV3_FAIL(-1, "resample_bicubic_u8: too many rows");
  int rc;
  constexpr int KMAX = 20, RPB = 4;                // fast RGB path: <= 20 taps (down-scales up to 4.5x), 4 rows per workgroup
  const long rows = (long)frames * h;
  const bool fast_h = c == 3 && ksx <= KMAX && (w * c) % 16 == 0 && w * c + KMAX * 3 <= kResampleMaxRowBytes / RPB &&
                      (((uintptr_t)src) & 15) == 0;
  // (round 6) up to 512 output columns: the persistent form -- coefficients live in registers, rows are double-buffered
  const bool persist_h = fast_h && ow <= 512 && w * c + KMAX * 3 + 8 <= kResamplePersistRowBytes;      // (+ 8: the aligned dwords around the last window)
  if (persist_h) {
    constexpr int PR = 2;                              // rows per group: 2 x 2 x 6 KB of LDS -> six resident workgroups per CU (four rows: 713 vs 659 us)
    const long groups = (rows + PR - 1) / PR;
    const long cap = 6l * num_cus();
    rc = L.launch(resample_h_rgb_persist_kernel<KMAX, PR, 2>, (int)(groups < cap ? groups : cap), 256, src, tmp, xmin, xcnt, kkx, ksx, rows, w, ow);
  } else if (fast_h) rc = L.launch(resample_h_rgb_kernel<KMAX, RPB>, (int)((rows + RPB - 1) / RPB), 256, src, tmp, xmin, xcnt, kkx, ksx, rows, w, ow);
  else rc = L.launch(resample_h_u8_kernel, frames * h, 256, src, tmp, xmin, xcnt, kkx, ksx, h, w, c, ow);
  if (rc) return rc;
  if ((ow * c) % 4 == 0 && (((uintptr_t)tmp | (uintptr_t)dst_u8) & 3) == 0)
    return L.launch(resample_v_u8x4_kernel, frames * oh, 256, (const unsigned char*)tmp, dst_f32, dst_u8, ymin, ycnt, kky, ksy, lut, h, ow, c, oh);
  return L.launch(resample_v_u8_kernel, frames * oh, 256, (const unsigned char*)tmp, dst_f32, dst_u8, ymin, ycnt, kky, ksy, lut, h, ow, c, oh);
}

template <class Launcher>
int median_u8_impl(Launcher& L, const unsigned char* frames, unsigned char* med, unsigned short* med2, int t, long p) {
  if (!frames || (!med && !med2) || t <= 0 || p <= 0) TNV3_FAIL(-1, "median_u8: bad argument");
  if ((p & 3) == 0 && t <= 65535 && (((uintptr_t)frames | (uintptr_t)med) & 3) == 0 && (((uintptr_t)med2) & 1) == 0) {
    const long p4 = p / 4, blocks4 = (p4 + 255) / 256;          // radix select in registers: 4 byte positions per thread
    if (blocks4 >= (1l << 31)) TNV3_FAIL(-1, "median_u8: frame too large");
    return L.launch(median_u8_radix_kernel, (int)blocks4, 256, (const unsigned int*)frames, med, med2, t, p4);
  }
  const long blocks = (p + 127) / 128;
  if (blocks >= (1l << 31)) TNV3_FAIL(-1, "median_u8: frame too large");
  return L.launch(median_u8_kernel, (int)blocks, 128, frames, med, med2, t, p);
}

template <class Launcher>
int absdiff_sum_u8_impl(Launcher& L, const unsigned char* frames, const unsigned short* med2, unsigned char* out, int f, long p) {
  if (!frames || !med2 || !out || f <= 0 || p <= 0) TNV3_FAIL(-1, "absdiff_sum_u8: bad argument");
  return L.launch(absdiff_sum_u8_kernel, grid_for((long)f * p), 256, frames, med2, out, f, p);
}

// ---- InpaintNet backward
template <class Launcher>
int conv1d_act_backward_impl(Launcher& L, const float* dout, const float* out, float* dpre, int n, int c, int l, int act, int nlc) {
  if (!dout || !out || !dpre || n <= 0 || c <= 0 || l <= 0 || act < 0 || act > 2) TNV3_FAIL(-1, "conv1d_act_backward: bad argument");
  return L.launch(conv1d_act_bwd_kernel, grid_for((long)n * c * l), 256, dout, out, dpre, n, c, l, act, nlc ? 1 : 0);
}

// dX = conv1d(dPre [N][cout][L], W^T flipped): first c0 input-channel gradients -> dx0, remaining c1 -> dx1
template <class Launcher>
int conv1d_k3_dgrad_impl(Launcher& L, const float* dpre, const float* w, float* dx0, float* dx1, int n, int cout, int c0, int c1,
                         int l, int accumulate) {
  if (!dpre || !w || !dx0 || n <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || l <= 0) TNV3_FAIL(-1, "conv1d_k3_dgrad: bad argument");
  if ((c1 > 0) != (dx1 != nullptr)) TNV3_FAIL(-1, "conv1d_k3_dgrad: dx1 / c1 mismatch");
  constexpr int S = 8, COB = 32, CK = 32, LT = 16;
  Conv1dArgs a{dpre, nullptr, w, nullptr, dx0, n, cout, 0, c0 + c1, l, 0, 0, 0, 1, dx1, c1 > 0 ? c0 : 0, accumulate};
  return L.launch3(conv1d_k3_kernel<S, COB, CK, LT>, (n + S - 1) / S, (c0 + c1 + COB - 1) / COB, (l + LT - 1) / LT, S * COB, a);
}

constexpr int kConv1dLMax = 64;
inline int conv1d_wgrad_nsplit(int n, int c0, int c1, int cout) {
  const int blocks = ((c0 + c1 + 31) / 32) * ((cout + 31) / 32);
  int ns = (1024 + blocks - 1) / blocks;
  const int by_work = (n + 7) / 8;
  if (ns > by_work) ns = by_work;
  return ns < 1 ? 1 : (ns > 256 ? 256 : ns);
}
inline size_t conv1d_wgrad_workspace_bytes(int n, int c0, int c1, int cout) {
  if (n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0) return 0;
  return (size_t)conv1d_wgrad_nsplit(n, c0, c1, cout) * ((size_t)cout * (c0 + c1) * 3 + cout) * sizeof(float);
}

template <class Launcher>
int conv1d_k3_wgrad_impl(Launcher& L, const float* src0, const float* src1, const float* dpre, float* dw, float* db, void* ws,
                         size_t ws_bytes, int n, int c0, int c1, int cout, int l, int src_nlc) {
  if (!src0 || !dpre || !dw || !db || !ws || n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || l <= 0) TNV3_FAIL(-1, "conv1d_k3_wgrad: bad argument");
  if ((c1 > 0) != (src1 != nullptr)) TNV3_FAIL(-1, "conv1d_k3_wgrad: src1 / c1 mismatch");
  if (l > kConv1dLMax) TNV3_FAIL(-1, "conv1d_k3_wgrad: sequence length %d > %d", l, kConv1dLMax);
  if (ws_bytes < conv1d_wgrad_workspace_bytes(n, c0, c1, cout)) TNV3_FAIL(-1, "conv1d_k3_wgrad: workspace too small");
  const int ns = conv1d_wgrad_nsplit(n, c0, c1, cout);
  const long nw = (long)cout * (c0 + c1) * 3, slab = nw + cout;
  int rc;
  if (l <= 16) rc = L.launch3(conv1d_wgrad_kernel<8, 16>, (c0 + c1 + 31) / 32, (cout + 31) / 32, ns, 256, src0, src1, dpre, (float*)ws, n, c0, c1, cout, l, src_nlc ? 1 : 0);
  else rc = L.launch3(conv1d_wgrad_kernel<2, kConv1dLMax>, (c0 + c1 + 31) / 32, (cout + 31) / 32, ns, 256, src0, src1, dpre, (float*)ws, n, c0, c1, cout, l, src_nlc ? 1 : 0);
  if (rc) return rc;
  if ((rc = L.launch(sum_partials_strided_kernel, grid_for(nw, 256, 1024), 256, (const float*)ws, dw, nw, ns, slab, 0l))) return rc;
  return L.launch(sum_partials_strided_kernel, 1, 256, (const float*)ws, db, (long)cout, ns, slab, nw);
}

}  // namespace tnv3
