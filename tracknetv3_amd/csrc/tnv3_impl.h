// tnv3_impl.h -- host-side argument checking and kernel dispatch behind the C ABI (include/tracknetv3_hip.h).
// Templated on a Launcher so that the CPU test-suite can drive the very same dispatch code through the SIMT
// emulator (tests/emu); the shipped library instantiates it with the HIP launcher only (tnv3_capi.hip).
#pragma once
#include <stdio.h>
#include <string.h>

#include "kernels/conv3x3_mfma.h"
#include "kernels/conv1d_k3.h"
#include "kernels/pointwise.h"

namespace tnv3 {

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
using ConvC7 = ConvCfg<2, 1, 1, 4, 4, 32, 8>;    // 64 ch x ( 4x32 px)  (small images)
constexpr int kNumConvConfigs = 8;

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
    default: return {0, 0, 0, 0, 0, 0};
  }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Library default when the caller passes cfg = -1 (overridden per layer by the tuned table on the Python
// side).  Measured on MI355X (profiles/r01_*): the small 4x32-pixel tiles win on every TrackNet layer because
// they run 3 waves per SIMD; 128-channel blocks are marginally better when Cout allows.
inline int conv_auto_config(int /*N*/, int Cout, int /*H*/, int /*W*/) { return (Cout % 128 == 0) ? 4 : 7; }

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
int conv3x3_forward_impl(Launcher& L, const float* src0, const float* src1, const float* wpack, const float* scale,
                         const float* shift, float* dst, int n, int c0, int c1, int cout, int h, int w, int up0,
                         int relu, int cfg) {
  if (!src0 || !wpack || !dst) TNV3_FAIL(-1, "conv3x3: null pointer");
  if (n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || h <= 0 || w <= 0) TNV3_FAIL(-1, "conv3x3: non-positive dimension");
  if ((c1 > 0) != (src1 != nullptr)) TNV3_FAIL(-1, "conv3x3: src1 / c1 mismatch");
  if ((scale == nullptr) != (shift == nullptr)) TNV3_FAIL(-1, "conv3x3: scale and shift must both be given or both be NULL");
  if (cout % 64) TNV3_FAIL(-1, "conv3x3: Cout=%d must be a multiple of 64", cout);
  if (h >= 8192 || w >= 8192) TNV3_FAIL(-1, "conv3x3: H,W must be < 8192");
  if (c1 > 0 && (c0 % 32)) TNV3_FAIL(-1, "conv3x3: two-source input needs C0 %% 32 == 0 (got %d)", c0);
  if (up0 && ((h | w) & 1)) TNV3_FAIL(-1, "conv3x3: upsampled source needs even H,W");
  if (cfg < 0) cfg = conv_auto_config(n, cout, h, w);
  Conv3x3Args a{src0, src1, wpack, scale, shift, dst, n, c0, c1, cout, h, w, up0 ? 1 : 0, relu ? 1 : 0};
  switch (cfg) {
    case 0: return launch_conv_cfg<ConvC0>(L, a);
    case 1: return launch_conv_cfg<ConvC1>(L, a);
    case 2: return launch_conv_cfg<ConvC2>(L, a);
    case 3: return launch_conv_cfg<ConvC3>(L, a);
    case 4: return launch_conv_cfg<ConvC4>(L, a);
    case 5: return launch_conv_cfg<ConvC5>(L, a);
    case 6: return launch_conv_cfg<ConvC6>(L, a);
    case 7: return launch_conv_cfg<ConvC7>(L, a);
    default: TNV3_FAIL(-1, "conv3x3: unknown config %d", cfg);
  }
}

inline size_t conv3x3_packed_floats(int cout, int cin, int transpose_flip) {
  const int K = transpose_flip ? cout : cin, M = transpose_flip ? cin : cout;
  return (size_t)round_up(K, 32) * 9 * M;
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
int bn_fold_impl(Launcher& L, const float* g, const float* b, const float* rm, const float* rv, float eps, float* scale,
                 float* shift, int c) {
  if (!g || !b || !rm || !rv || !scale || !shift || c <= 0) TNV3_FAIL(-1, "bn_fold: bad argument");
  return L.launch(bn_fold_kernel, (c + 255) / 256, 256, g, b, rm, rv, eps, scale, shift, c);
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

template <class Launcher>
int conv1d_k3_impl(Launcher& L, const float* src0, const float* src1, const float* w, const float* b, float* dst, int n,
                   int c0, int c1, int cout, int l, int src_nlc, int dst_nlc, int act) {
  if (!src0 || !w || !b || !dst || n <= 0 || c0 <= 0 || c1 < 0 || cout <= 0 || l <= 0) TNV3_FAIL(-1, "conv1d_k3: bad argument");
  if ((c1 > 0) != (src1 != nullptr)) TNV3_FAIL(-1, "conv1d_k3: src1 / c1 mismatch");
  if (act < 0 || act > 2) TNV3_FAIL(-1, "conv1d_k3: unknown activation %d", act);
  constexpr int S = 8, COB = 32, CK = 32, LT = 16;
  Conv1dArgs a{src0, src1, w, b, dst, n, c0, c1, cout, l, src_nlc ? 1 : 0, dst_nlc ? 1 : 0, act};
  const long gx = (n + S - 1) / S;
  if (gx > 0x7fffffffl) TNV3_FAIL(-1, "conv1d_k3: batch too large");
  return L.launch3(conv1d_k3_kernel<S, COB, CK, LT>, (int)gx, (cout + COB - 1) / COB, (l + LT - 1) / LT, S * COB, a);
}

}  // namespace tnv3
