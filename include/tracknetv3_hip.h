/* tracknetv3_hip.h -- C ABI of libtnv3_hip.so: the MI355X (gfx950) hot path of TrackNetV3.
 *
 * The reference (qaz812345/TrackNetV3) is pure Python on PyTorch and has NO plugin / FFI layer: its
 * device arithmetic is the implicit ATen/cuDNN/OpenCV work behind nn.Module calls.  This header is therefore
 * the FFI a maintainer would bind *under* those calls; each entry point cites the reference construct whose
 * device work it replaces (paths relative to the upstream repository).  INTEGRATION.md shows the ctypes
 * binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (NCHW / NCL) unless stated; int32 for integers
 *   - the caller owns all buffers (outputs, packed weights, workspaces); nothing is allocated or retained
 *   - calls only ENQUEUE on `stream` (a hipStream_t passed as void*; NULL = default stream) and return;
 *     no internal synchronisation; safe to call from several host threads on distinct streams
 *   - the DEVICE is the stream's: a call makes the stream's device current for its own duration (and restores the
 *     caller's), so pointers must belong to the stream's device; a NULL stream means the calling thread's current
 *     device.  The stream-less *_workspace_bytes / *_packed_floats queries plan for the calling thread's current
 *     device (all GPUs of one node are identical; on a mixed node query with the right device current)
 *   - NO process-wide mutable state: kernel-family choices are per-call arguments (`cfg`, `variant`; -1 = default)
 *   - return value: 0 = OK, <0 = error (TNV3_E_*); tnv3_last_error() gives the text for the calling thread
 *   - no exceptions cross the ABI
 */
#ifndef TRACKNETV3_HIP_H
#define TRACKNETV3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNV3_OK 0
#define TNV3_E_INVALID (-1)   /* bad argument / unsupported shape */
#define TNV3_E_LAUNCH (-2)    /* HIP launch error */
#define TNV3_ABI_VERSION 8   /* 2: per-call kernel variants instead of process-wide knobs; diagnostics moved to libtnv3_diag.so;
                                 3: `variant` argument on the Winograd-form weight gradients;
                                 4: `sum_order` on tnv3_ensemble_frames (the reference's summation order, bit-exact); `variant` on the
                                    9-GEMM decoder kernels; the fused InpaintNet training entries; tnv3_conv3x3_wino_pick / _has_stats /
                                    _pack_multi; tnv3_conv3x3_wgrad_wino variants 2-7 and any Cin;
                                 5: `variant` on the tnv3_conv3x3_wino43_* and tnv3_conv_up2x_wino_* entries (0 / 2: the MFMA 16x16x4
                                    kernels, 1: the 32x32x2 kernel); `pool_dst` on tnv3_conv3x3_wino43_forward; pack-multi layout 4;
                                    tnv3_conv3x3_wgrad_wino variant 8 (F(4x4)), the default where h % 4 == 0; the measured-and-rejected
                                    generations (tnv3_conv3x3_wino_forward 0 / 2 / 4, tnv3_conv3x3_wgrad_wino 0 / 3 / 4 / 6 / 7) left the
                                    product library: they are refused here and stay dispatchable in libtnv3_diag.so;
                                 6: `up_variant` on tnv3_conv3x3_wgrad_up2x (the upsampled half's 25-of-36 F(4x4) weight gradient), `variant` 2
                                    on tnv3_dgrad_up2x_wino (its data gradient on the 16x16x4 kernel);
                                 7: tnv3_conv3x3_wino43_dgrad_bnstats (the F(4x4) data gradient that takes the previous block's BatchNorm-backward
                                    sums from its write-out);
                                 8: tnv3_maxpool2x2_backward_add_bnstats (the max-pool backward + skip add that takes them the same way);
                                    tnv3_bn_train_forward_tiles_pool (the normalise + ReLU pass that also writes the pooled tensor);
                                    tnv3_dgrad_up2x_wino_bnstats (the upsampled half's data gradient that takes those sums) */

typedef void* tnv3_stream_t;

int tnv3_abi_version(void);
const char* tnv3_last_error(void);

/* ---- 3x3 convolution (Conv2DBlock: model.py:4-16; concat+upsample: model.py:65,67,69) ------------------- */

/* Number of compiled tile configurations of the MFMA conv kernel, and their geometry. */
int tnv3_conv3x3_num_configs(void);
int tnv3_conv3x3_config_info(int cfg, int* m_block, int* tile_rows, int* tile_cols, int* chan_chunk,
                             int* threads, int* lds_bytes);

/* Floats needed for the packed weights of a Cout x Cin x 3 x 3 filter (forward: transpose_flip = 0;
 * data-gradient: transpose_flip = 1). */
size_t tnv3_conv3x3_packed_floats(int cout, int cin, int transpose_flip);

/* W[Cout][Cin][3][3] -> Wp[roundup(K,32)][3][3][M] (K,M = Cin,Cout forward; Cout,Cin with flipped taps for
 * the data gradient).  Replaces nothing in the reference: cuDNN does its own filter transform. */
int tnv3_pack_conv3x3_weights(const float* w, float* wpack, int cout, int cin, int transpose_flip,
                              tnv3_stream_t stream);

/* Eval-mode nn.BatchNorm2d (model.py:9): scale[c] = gamma[c] / sqrt(running_var[c] + eps).  The conv epilogue then
 * computes (x - running_mean) * scale + beta in the reference's operation order (the mean is deliberately not
 * folded into the shift: that form cancels catastrophically when |mean| >> std). */
int tnv3_bn_eval_scale(const float* gamma, const float* running_var, float eps, float* scale, int channels,
                       tnv3_stream_t stream);

/* dst[N][Cout][H][W] = act( (conv3x3( cat([up2x?(src0), src1], dim=1), W ) - mean) * scale + shift )
 *   src0 : [N][C0][H][W], or [N][C0][H/2][W/2] when up0 != 0 (nn.Upsample(scale_factor=2), nearest)
 *   src1 : [N][C1][H][W] or NULL (C1 = 0); its channels follow src0's, as torch.cat([up, skip], dim=1)
 *   wpack: from tnv3_pack_conv3x3_weights for Cin = C0 + C1
 *   mean/scale/shift: [Cout]; scale and shift both NULL = raw convolution; mean may be NULL (= 0);
 *   relu != 0 applies max(.,0)
 *   cfg  : tile configuration index, or -1 to let the library choose
 * Requirements: Cout % 64 == 0; H,W < 8192; when C1 > 0, C0 % 32 == 0.  */
int tnv3_conv3x3_forward(const float* src0, const float* src1, const float* wpack, const float* mean,
                         const float* scale, const float* shift, float* dst, int n, int c0, int c1, int cout,
                         int h, int w, int up0, int relu, int cfg, tnv3_stream_t stream);

/* The same kernel with an ADDEND: dst = act(((conv3x3(...) + addend) - mean) * scale + shift), addend [N][Cout][H][W] or NULL.
 * Used for the skip half of a decoder-entry layer, whose upsampled half comes from tnv3_conv_up2x_forward. */
int tnv3_conv3x3_forward_add(const float* src0, const float* src1, const float* wpack, const float* addend, const float* mean,
                             const float* scale, const float* shift, float* dst, int n, int c0, int c1, int cout, int h, int w,
                             int up0, int relu, int cfg, tnv3_stream_t stream);

/* Winograd F(2x2, 3x3) form of the plain layer: dst = act(((conv3x3(src, W) + addend) - mean) * scale + shift), single source,
 * 16 instead of 36 multiply-adds per (co, ci, 2x2 tile), transforms fused into the kernel.  Same function as
 * tnv3_conv3x3_forward_add up to fp32 rounding (~6e-7 of the output scale per layer instead of ~2e-7).
 *   tnv3_conv3x3_wino_supported     : 1 when the shape qualifies (Cout % 64 == 0, H % 4 == 0, W % 64 == 0)
 *   tnv3_conv3x3_wino_packed_floats : size of the transformed-filter buffer
 *   tnv3_conv3x3_wino_pack          : w [cout][cin][3][3] -> u = G w G^T, [cin_pad][16][cout]
 *   tnv3_conv3x3_wino_pack_view     : the same straight from the nn.Conv2d weight w [cout_w][cin_w][3][3] for its input channels
 *                                     c_from .. c_from + c_count - 1 (skip half of a decoder entry), as the forward filter
 *                                     (u for cout_w x c_count) or, transpose_flip != 0, as the data gradient's filter
 *                                     w'[ci][co][kh][kw] = w[co][c_from + ci][2-kh][2-kw] (u for c_count x cout_w): no host-side
 *                                     slice / flip / transpose copies (model.py:8 weight layout) */
/*   `variant` of tnv3_conv3x3_wino_forward (per call): the dispatchable kernels since ABI 5 are
 *                                     3 = 64 channels x (4 x 64 pixels) per workgroup, two waves per SIMD, buffer-descriptor LDS-DMA and a
 *                                     paired patch transform;  5 = 3 as persistent workgroups (one per CU walking the tile
 *                                     list, the chunk pipeline running through the tile boundaries: no per-tile launch, set-up, first-DMA
 *                                     wait or first transform; Cin <= 8 runs as 3);
 *                                     6 = 5 re-tiled to 128 output channels x (4 x 32 pixels) per workgroup, the filter operand read
 *                                     straight from L2 into registers (kernels/conv3x3_wino6_mfma.h; Cout % 128 == 0, Cin > 8,
 *                                     W % 32 == 0; filters packed with layout 2);
 *                                     7 = the same kernel with a 64-channel x (4 x 64 pixels) workgroup tile (Cout % 64 == 0, Cin > 8,
 *                                     W % 64 == 0; layout 2).  3, 5, 6, 7 are bit-identical to each other.
 *                                     -1 = tnv3_conv3x3_wino_pick(cin, cout): 6 for Cout % 128 == 0 and Cin > 8, else 5 (7 measured 2-5 % slower) -- by channel counts only,
 *                                     so a panel packed ahead of time is the one every call of that layer reads.
 *                                     (0, 2, 4 -- the generations 3 was derived from -- are measurement twins of libtnv3_diag.so since
 *                                     ABI 5 and are refused here; 1 was removed in round 2.)
 *   `layout` of the pack calls: 0 = u[cin_pad][16][cout];  1 = u[cin_pad / 2][4][2][cout][4] (transform row major, the four xi of
 *                                     a row adjacent: the diagnostic library's kernel 4);  2 = u[cout / 32][cin_pad / 8][2][8][64][4] (the A operand in lane order). */
size_t tnv3_conv3x3_wino_packed_floats(int cin, int cout);
int tnv3_conv3x3_wino_supported(int cin, int cout, int h, int w);
int tnv3_conv3x3_wino_pick(int cin, int cout); /* the kernel variant that `variant` = -1 means for these channel counts */
/* The three helpers below take a RESOLVED variant (>= 0): -1 means different kernels for different channel counts, so resolve it with
 * tnv3_conv3x3_wino_pick(cin, cout) first -- with -1 they fail (TNV3_E_INVALID / 0 tiles) instead of guessing a kernel the launcher
 * would not run (a statistics buffer of half the size, a panel in the wrong layout). */
int tnv3_conv3x3_wino_layout(int variant);     /* filter pack layout (0, 1 or 2) the kernel `variant` reads */
int tnv3_conv3x3_wino_has_stats(int variant);  /* 1 when `variant` can emit BatchNorm batch statistics from its epilogue
                                                  (tnv3_conv3x3_wino_forward_stats), else 0 */
int tnv3_conv3x3_wino_pack(const float* w, float* u, int cout, int cin, int layout, tnv3_stream_t stream);
int tnv3_conv3x3_wino_pack_view(const float* w, float* u, int cout_w, int cin_w, int c_from, int c_count, int transpose_flip,
                                int layout, tnv3_stream_t stream);
/* The same for a list of panels in ONE launch (a training step re-packs the forward and data-gradient filters of every layer after
 * each optimiser step): items[k] = the arguments of tnv3_conv3x3_wino_pack_view for panel k; `items` is HOST memory, read before
 * the call returns.  Bit-identical to the one-panel calls.  `layout` 0-2: F(2x2) panels; 4 / 3: the F(4x4, 3x3) panel of
 * tnv3_conv3x3_wino43_pack with variant 0 / 1 (u then holds tnv3_conv3x3_wino43_packed_floats floats). */
typedef struct tnv3_wino_pack_item {
  const float* w;   /* nn.Conv2d weight [cout_w][cin_w][3][3] (device) */
  float* u;         /* panel of tnv3_conv3x3_wino_packed_floats(cin, cout) floats (device) */
  int cout_w, cin_w, c_from, c_count, transpose_flip, layout;
} tnv3_wino_pack_item;
int tnv3_conv3x3_wino_pack_multi(const tnv3_wino_pack_item* items, int count, tnv3_stream_t stream);
/* The same layer in Winograd F(4x4, 3x3) form: 36 products per 4x4 output tile (2.25 per pixel; F(2x2): 4, direct: 9), the
 * transforms of the interpolation points (0, +-3/4, +-3/2, infinity) fused into the kernel (a panel is only meaningful to the library
 * version that packed it: pack with tnv3_conv3x3_wino43_pack, never by hand).  Same function as tnv3_conv3x3_wino_forward up to fp32
 * rounding (1.0-1.6e-6 of the output scale per layer, F(2x2): 3-8e-7; the whole TrackNet forward's heat maps stay within 1.7e-6 of the
 * fp64 forward).  supported: Cout % 64 == 0, H % 4 == 0, W % 64 == 0.  The panel (its own layout, _packed_floats floats, 16-byte
 * aligned) is packed straight from the nn.Conv2d weight like tnv3_conv3x3_wino_pack_view: an input-channel slice, or transpose_flip
 * for the data gradient's filter.  addend / mean / scale / shift / relu as in tnv3_conv3x3_wino_forward; mean / scale / shift 16-byte
 * aligned.  `variant` (pack and run a panel with the SAME one): 0 = kernels/conv3x3_wino43s_mfma.h -- 16x16x4 MFMAs, all 36 transform
 * coefficients of a (16 channels x 16 tiles) block in one wave, one wave per SIMD, the output transform in registers; 1 = its
 * predecessor kernels/conv3x3_wino43_mfma.h -- 32x32x2 MFMAs, four waves per block meeting through LDS (kept as the A/B twin).
 * Same function, not bit-identical to each other.  pool_dst (optional, variants 0 / 2): [n][cout][h/2][w/2] = MaxPool2d(2, 2) of dst
 * (model.py:59,61,63) from the write-out's registers -- bit-identical to tnv3_maxpool2x2 on dst, without its pass. */
int tnv3_conv3x3_wino43_supported(int cin, int cout, int h, int w);
size_t tnv3_conv3x3_wino43_packed_floats(int cin, int cout, int variant);
int tnv3_conv3x3_wino43_pack(const float* w, float* u, int cout_w, int cin_w, int c_from, int c_count, int transpose_flip, int variant,
                             tnv3_stream_t stream);
int tnv3_conv3x3_wino43_forward(const float* src, const float* u, const float* addend, const float* mean, const float* scale,
                                const float* shift, float* dst, float* pool_dst, int n, int cin, int cout, int h, int w, int relu, int variant,
                                tnv3_stream_t stream);
/* Training forward in that form: dst = conv3x3(src) + addend (raw), and from the same kernel's epilogue tile_stats[cout][tiles][2]
 * (fp64 sum and sum of squares per channel and pixel tile -- 4 x 64 pixels for variant 0, 8 x 64 for 1; tiles =
 * tnv3_conv3x3_wino43_stats_tiles) for
 * tnv3_bn_train_forward_tiles -- the F(4x4) twin of tnv3_conv3x3_wino_forward_stats.  Deterministic. */
long tnv3_conv3x3_wino43_stats_tiles(int n, int h, int w, int variant);
int tnv3_conv3x3_wino43_forward_stats(const float* src, const float* u, const float* addend, float* dst, double* tile_stats, int n, int cin,
                                      int cout, int h, int w, int variant, tnv3_stream_t stream);

/* ABI 7.  The data gradient of a plain layer in that form -- da = conv3x3(dz, W^T flipped), u_t = the transposed, flipped panel
 * (tnv3_conv3x3_wino43_pack with transpose_flip) -- that ALSO takes, from the registers it writes da from, the two sums of the PREVIOUS
 * Conv2DBlock's BatchNorm + ReLU backward (autograd of model.py:9-10; train.py:95): tile_stats[cout][tiles][2] = per channel and pixel tile
 * (tiles = tnv3_conv3x3_wino43_stats_tiles) sum g and sum g * xhat, g = da * [BN(z) > 0] with the forward's own expression, xhat = (z - mean) *
 * invstd; bn_z = that block's raw convolution output [n][cout][h][w], bn_mean / bn_invstd = its saved batch statistics, bn_gamma / bn_beta its
 * affine.  Feed da and the sums to tnv3_bn_relu_backward_tiles: the separate pass over (da, z) that tnv3_bn_relu_backward would make for the
 * sums is gone.  The F(4x4) twin of tnv3_conv3x3_wino_dgrad_bnstats; da bit-identical to tnv3_conv3x3_wino43_forward's.  Variants 0 / 2. */
int tnv3_conv3x3_wino43_dgrad_bnstats(const float* dz, const float* u_t, float* da, double* tile_stats, const float* bn_z, const float* bn_mean,
                                      const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int n, int cin, int cout, int h, int w,
                                      int variant, tnv3_stream_t stream);

int tnv3_conv3x3_wino_forward(const float* src, const float* u, const float* addend, const float* mean, const float* scale,
                              const float* shift, float* dst, int n, int cin, int cout, int h, int w, int relu, int variant,
                              tnv3_stream_t stream);

/* Training-mode forward of a Conv2DBlock with the BatchNorm batch statistics taken in the convolution's epilogue (model.py:8-9):
 * dst = conv3x3(src, W) + addend (raw sums, kernel variants 3 / 4), and tile_stats[cout][tiles][2] (doubles) = per output channel and
 * pixel tile (4 x 64 pixels, variant 6: 4 x 32; tiles = tnv3_conv3x3_wino_stats_tiles(n, h, w, variant)) the sum and the sum of squares of the values written --
 * reduced inside the kernel by a wave butterfly and a fixed-order fold, so the statistics cost no pass over dst.  Feed them to
 * tnv3_bn_train_forward_tiles. */
long tnv3_conv3x3_wino_stats_tiles(int n, int h, int w, int variant);
int tnv3_conv3x3_wino_forward_stats(const float* src, const float* u, const float* addend, float* dst, double* tile_stats, int n, int cin,
                                    int cout, int h, int w, int variant, tnv3_stream_t stream);

/* Decoder-entry layers (model.py:65,67,69: Conv2DBlock on torch.cat([nn.Upsample(scale_factor=2)(x), skip], dim=1)):
 * the contribution of the UPSAMPLED channels computed at the low resolution.  A 3x3 'same' convolution over a nearest-2x
 * upsampled tensor reads only 2x2 distinct source pixels per output pixel, so with the taps that coincide pre-summed
 * (four class filters, one per output parity) it costs 4/9 of the multiply-adds and reads the low-res tensor once.
 *   tnv3_conv_up2x_packed_floats : size of the class-filter buffer for c0 upsampled channels
 *   tnv3_pack_up2x_weights       : w [cout][cin][3][3] (the layer's nn.Conv2d weight; its first c0 input channels are the
 *                                  upsampled ones) -> wq
 *   tnv3_conv_up2x_forward       : src_low [n][c0][h_low][w_low], wq -> dst [n][cout][2*h_low][2*w_low] partial sums (no
 *                                  BN / activation); feed it to tnv3_conv3x3_forward_add as `addend` together with the skip
 *                                  tensor and the packed filter of input channels c0.. of the same weight. */
size_t tnv3_conv_up2x_packed_floats(int c0, int cout);
int tnv3_pack_up2x_weights(const float* w, float* wq, int cout, int cin, int c0, tnv3_stream_t stream);
int tnv3_conv_up2x_forward(const float* src_low, const float* wq, float* dst, int n, int c0, int cout, int h_low, int w_low,
                           int cfg /* tile configuration 0..3, -1 = library default */, tnv3_stream_t stream);

/* The same partial sums in Winograd F(2x2, 3x3) form.  The 4x4 input patch of a nearest-2x upsampled tensor has rows
 * (L[i-1], L[i], L[i], L[i+1]) of the low-resolution tensor, so the transform row d2 - d1 (and column) vanishes identically:
 * 9 of the 16 Winograd GEMMs remain -- 9 multiply-adds per low-resolution pixel and channel pair instead of the 16 of the class
 * filters above (36 in the reference's direct form).  Persistent streaming kernel, output transform in registers.
 *   supported: c0 > 8, cout % 64 == 0, h_low % 2 == 0, w_low % 64 == 0.  u from tnv3_conv_up2x_wino_pack (the layer's
 *   nn.Conv2d weight, its first c0 input channels), 16-byte aligned.  Same function up to fp32 rounding.
 *   variant (also of tnv3_dgrad_up2x_wino): -1 / 0 = the older waves of each SIMD run their MFMA phase first, 1 = the
 *   younger ones (round 2's order); bit-identical results, a scheduling choice only.
 *   variant 2 (forward only): Winograd F(4x4, 3x3) on the 16x16x4 kernel (kernels/conv3x3_wino43s_mfma.h, MODE 1).  With Lavin's
 *   points (0, +-1, +-2, inf) the transform row of the point -1 vanishes on an upsampled signal and the rows of +-2 are proportional:
 *   25 of the 36 products per 4x4 output tile remain -- 6.25 multiply-adds per low-resolution pixel and channel pair.  Its own panel
 *   (pack with variant 2); supported: c0 > 0, cout % 64 == 0, h_low % 2 == 0, w_low % 32 == 0.  Not bit-identical to 0 / 1 (another
 *   factorisation: 1-5e-6 of the output scale from the fp64 result). */
int tnv3_conv_up2x_wino_supported(int c0, int cout, int h_low, int w_low, int variant);
size_t tnv3_conv_up2x_wino_packed_floats(int c0, int cout, int variant);
int tnv3_conv_up2x_wino_pack(const float* w, float* u, int cout, int cin, int c0, int variant, tnv3_stream_t stream);
int tnv3_conv_up2x_wino_forward(const float* src_low, const float* u, float* dst, int n, int c0, int cout, int h_low, int w_low,
                                int variant, tnv3_stream_t stream);

/* Data gradient of that half-layer (autograd of nn.Upsample(scale_factor=2) -> Conv2d w.r.t. the low-res tensor), also at
 * the low resolution: dx_low[n][c0][h_low][w_low] = 4x4 stride-2 correlation of dz[n][cout][2*h_low][2*w_low] with the
 * pre-summed filters g (tnv3_pack_dgrad_up2x_weights from the same nn.Conv2d weight).  Replaces "3x3 data gradient at full
 * resolution + sum over 2x2 blocks" at 4/9 of the multiply-adds; c0 % 4 == 0. */
size_t tnv3_dgrad_up2x_packed_floats(int c0, int cout);
int tnv3_pack_dgrad_up2x_weights(const float* w, float* g, int cout, int cin, int c0, tnv3_stream_t stream);
int tnv3_dgrad_up2x(const float* dz, const float* g, float* dx_low, int n, int c0, int cout, int h_low, int w_low, tnv3_stream_t stream);

/* The same gradient as ONE GEMM with K = 9 * cout: in Winograd F(2x2, 3x3) form the 2x2 block sum of nn.Upsample's backward is
 * c^T M c with c = A 1 = (1, 2, 0, -1), so transform row / column 2 drops out and the nine remaining products share one accumulator:
 * dx_low[ci][p] = sum_{co, xi} U''_xi[co][ci] * (B^T d B)_xi[co][p] -- 9 instead of 16 multiply-adds per (ci, co, low-res pixel).
 *   supported: c0 % 128 == 0, cout > 8, h_low % 2 == 0, w_low % 32 == 0;  u from tnv3_dgrad_up2x_wino_pack (the layer's nn.Conv2d
 *   weight, its first c0 input channels), 16-byte aligned.  Same gradient up to fp32 rounding.
 *   variant (ABI 6: also on _supported / _packed_floats / _pack): -1 / 0 / 1 = that kernel (1: the younger waves' MFMA phase first);
 *   2 = Winograd F(4x4, 3x3) on the 16x16x4 kernel (kernels/conv3x3_wino43s_mfma.h, MODE 2): the 2x2 block sum P A^T of the 4x4 tile has
 *   a zero column at the interpolation point -1, so 25 of the 36 products remain -- 6.25 multiply-adds per (ci, co, low-res pixel).  Its
 *   own panel (pack with variant 2); supported: c0 % 64 == 0, h_low % 2 == 0, w_low % 32 == 0, any cout. */
int tnv3_dgrad_up2x_wino_supported(int c0, int cout, int h_low, int w_low, int variant);
size_t tnv3_dgrad_up2x_wino_packed_floats(int c0, int cout, int variant);
int tnv3_dgrad_up2x_wino_pack(const float* w, float* u, int cout, int cin, int c0, int variant, tnv3_stream_t stream);
int tnv3_dgrad_up2x_wino(const float* dz, const float* u, float* dx_low, int n, int c0, int cout, int h_low, int w_low, int variant,
                         tnv3_stream_t stream);
/* ABI 8.  Variant 2's launch that also takes the BatchNorm + ReLU backward sums of the block whose activation the decoder entry upsamples
 * (model.py:64-69: dx_low IS that block's dA, nothing else reads its activation): bn_z = that block's raw convolution output [n][c0][h_low][w_low],
 * bn_* its saved mean / invstd and the affine parameters as its forward used them; tile_stats [c0][n * (h_low / 2) * (w_low / 32)][2] doubles, to be
 * fed with dx_low to tnv3_bn_relu_backward_tiles (autograd of model.py:9-10).  The same dx_low bits as tnv3_dgrad_up2x_wino(variant 2). */
int tnv3_dgrad_up2x_wino_bnstats(const float* dz, const float* u, float* dx_low, double* tile_stats, const float* bn_z, const float* bn_mean,
                                 const float* bn_invstd, const float* bn_gamma, const float* bn_beta, int n, int c0, int cout, int h_low, int w_low,
                                 int variant, tnv3_stream_t stream);

/* Weight gradient of a plain layer (single source, no upsampling) in Winograd form: dw[cout][cin][3][3] =
 * G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G -- per (co, ci) 36 multiply-adds per 4x4 tile in F(4x4, 3x3) form (2.25 per pixel), 16
 * per 2x2 tile in F(2x2, 3x3) form (4 per pixel), where the direct form spends 9 per pixel; same gradient as tnv3_conv3x3_wgrad up to
 * fp32 rounding (F(2x2): 6e-7, F(4x4): 2-4e-6 of max|dw| at batch 10); deterministic (fixed-order split-K sum).
 *   supported: cout % 64 == 0, h % 2 == 0, w % 16 == 0; any cin with kernels 5 / 6 / 8 (a partial last block of input channels: the
 *   stem layer), cin % 64 == 0 with kernels 0-4 and 7; kernel 8 needs h % 4 == 0;  workspace 16-byte aligned, size from the query
 *   (it covers every variant).
 *   `variant` (per call): -1 = the library's default: 8 where h % 4 == 0, else 1 (5 when cin % 64 != 0);
 *   8 = F(4x4, 3x3) with the interpolation points of tnv3_conv3x3_wino43_forward (kernels/wgrad_wino43_mfma.h): 64 co x 32 ci per
 *   workgroup, MFMA 16x16x4 with K = the four 4x4 tiles of a 4 x 16 pixel strip, both operands transformed in the shadow of the
 *   MFMAs, split-K slabs of 9 taps -- 1.45-1.5x faster than 1 on every TrackNet shape, 2.6x on the stem;
 *   the F(2x2) kernels: 1 = two waves per SIMD, the wave groups half a period apart (one transforms while the other streams MFMAs),
 *   paired transforms, buffer-descriptor LDS-DMA;  2 = 1 with 16-byte operand reads;  5 = every wave streams its MFMAs of a chunk and
 *   transforms its tile pair of the next chunk between them (16-byte operand reads, three raw stages; any cin).  1, 2 and 5 accumulate
 *   in the same order: bit-identical results.  (0, 3, 4, 6, 7 -- earlier / rejected generations -- are measurement twins of
 *   libtnv3_diag.so since ABI 5 and are refused here.) */
int tnv3_conv3x3_wgrad_wino_supported(int cin, int cout, int h, int w);
size_t tnv3_conv3x3_wgrad_wino_workspace_bytes(int n, int cin, int cout, int h, int w);
int tnv3_conv3x3_wgrad_wino(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes, int n, int cin, int cout,
                            int h, int w, int variant, tnv3_stream_t stream);

/* Weight gradient of a whole decoder-entry layer, dw[cout][c0+c1][3][3] for the nn.Conv2d applied to
 * cat([Upsample(2)(x_low), skip], dim=1): the c0 upsampled channels through the four parity images of dz and 2x2 tap
 * windows against x_low (16 instead of 36 taps per low-res pixel), the c1 skip channels as an ordinary 3x3 weight gradient.
 *   x_low [n][c0][h_low][w_low], skip [n][c1][2*h_low][2*w_low], dz [n][cout][2*h_low][2*w_low];  w_low % 4 == 0.
 * Deterministic (split-K slabs reduced in a fixed order); workspace from the _workspace_bytes query, 16-byte aligned.
 * `up_variant` (per call; ABI 6) -- the form of the upsampled half: -1 = the fastest the shape allows; 2 = the 25-of-36 Winograd F(4x4)
 * form (kernels/wgrad_up2x_wino43_mfma.h: 6.25 multiply-adds per low-res pixel; needs cout % 64 == 0, h_low % 2 == 0, w_low % 8 == 0, any
 * c0); 1 = the 9-GEMM F(2x2) form of tnv3_conv_up2x_wino_forward (9 per low-res pixel; c0 % 128 == 0, cout % 64 == 0, w_low % 8 == 0);
 * 0 = four 2x2-window launches over the parity images of dz (16).  A form the shape does not allow falls to the next lower one.
 * `wino_variant` (per call) -- the kernel of the skip half: -1 / 8 = the default kernel of tnv3_conv3x3_wgrad_wino (the F(4x4) kernel 8
 * where h % 4 == 0); 2 / 5 = the F(2x2) kernel 1 / 5; 1 = kernel 1, and with up_variant -1 the 2x2-window launches for the upsampled half
 * (ABI 5's meaning of 1).  All compute the same gradient up to fp32 rounding.  (0, 3, 4, 6, 7 are refused since ABI 5: libtnv3_diag.so.) */
size_t tnv3_conv3x3_wgrad_up2x_workspace_bytes(int n, int c0, int c1, int cout, int h_low, int w_low);
int tnv3_conv3x3_wgrad_up2x(const float* x_low, const float* skip, const float* dz, float* dw, void* workspace, size_t workspace_bytes,
                            int n, int c0, int c1, int cout, int h_low, int w_low, int wino_variant, int up_variant, tnv3_stream_t stream);

/* ---- head + pooling (model.py:54-55,59,61,63,71-72) ----------------------------------------------------- */

/* y[N][L][HW] = sigmoid?( b[l] + sum_c w[l][c] * x[N][C][HW] ); HW % 4 == 0. */
int tnv3_head1x1_sigmoid(const float* x, const float* w, const float* b, float* y, int n, int c, int l,
                         int hw, int apply_sigmoid, tnv3_stream_t stream);

/* nn.MaxPool2d((2,2), stride=(2,2)) over nc planes of h x w (h % 2 == 0, w % 4 == 0). */
int tnv3_maxpool2x2(const float* x, float* y, long nc, int h, int w, tnv3_stream_t stream);

/* ---- InpaintNet 1-D convolution (Conv1DBlock: model.py:76-87; InpaintNet.forward: model.py:113-129) ------- */

/* dst = act( conv1d_k3_same( cat([src0, src1], channel dim), w ) + b )
 *   src0/src1: [N][C][L] (or [N][L][C] when src_nlc != 0: the network input cat([coor (N,L,2), mask (N,L,1)], 2));
 *   src1 may be NULL (c1 = 0);  w: [Cout][C0+C1][3] exactly as in the state_dict;  b: [Cout]
 *   dst: [N][Cout][L], or [N][L][Cout] when dst_nlc != 0 (the network output after the final permute)
 *   act: 0 none, 1 LeakyReLU(0.01), 2 sigmoid */
int tnv3_conv1d_k3_forward(const float* src0, const float* src1, const float* w, const float* b, float* dst, int n,
                           int c0, int c1, int cout, int l, int src_nlc, int dst_nlc, int act, tnv3_stream_t stream);

/* InpaintNet.forward (model.py:113-129) as ONE persistent kernel: a workgroup carries a sequence through all nine layers with
 * every activation in LDS; the dense layers run on v_mfma_f32_16x16x4_f32 with filters streamed from L2 into registers.
 *   tnv3_inpaintnet_packed_floats : size of the packed parameter buffer (filters in lane order + biases)
 *   tnv3_inpaintnet_pack          : weights9 / biases9 = HOST arrays of the nine nn.Conv1d weight / bias DEVICE pointers in network
 *                                   order (down_1, down_2, down_3, buttleneck.conv_1, buttleneck.conv_2, up_1, up_2, up_3, predictor)
 *   tnv3_inpaintnet_fused_forward : x [n][16][2], m [n][16][1] -> out [n][16][2]; l must be 16; packed 16-byte aligned.
 * Same function as the nine tnv3_conv1d_k3_forward launches (fp32 rounding order differs: K is walked tap-major). */
size_t tnv3_inpaintnet_packed_floats(void);
int tnv3_inpaintnet_pack(const float* const* weights9, const float* const* biases9, float* packed, tnv3_stream_t stream);
int tnv3_inpaintnet_fused_forward(const float* x, const float* m, const float* packed, float* out, int n, int l, tnv3_stream_t stream);

/* An InpaintNet training step's forward + backward (train.py:147-166 through model.py:113-129) in THREE launches
 * (kernels/inpaint_fused_train.h) instead of ~35 per-layer ones:
 *   tnv3_inpaintnet_fused_forward_train : the fused forward that also saves the eight hidden activations, acts [n][960][16]
 *                                         (tnv3_inpaintnet_act_floats(n) floats; order x1, x2, x3, b1, b2, u1, u2, u3)
 *   tnv3_inpaintnet_pack_t              : transposed, tap-flipped filters of the seven dense layers in lane order
 *                                         (tnv3_inpaintnet_packed_t_floats() floats; weights9 as for tnv3_inpaintnet_pack)
 *   tnv3_inpaintnet_fused_backward      : dout = dLoss/dOut [n][16][2], out = the forward's output -> grads
 *                                         [tnv3_inpaintnet_param_floats() = 520 610] = the 18 parameter gradients in state_dict
 *                                         order (weight, bias per layer; views into it are the .grad tensors).  dpre: caller-owned
 *                                         scratch of tnv3_inpaintnet_dpre_floats(n) floats ([n][962][16], 16-byte aligned).  One
 *                                         workgroup carries a sequence's gradient through all nine layers in LDS (MFMA 16x16x4);
 *                                         one more launch forms every dW / db over the whole batch in a fixed order (deterministic).
 * Same function as tnv3_conv1d_* backward kernels up to fp32 summation order. */
size_t tnv3_inpaintnet_packed_t_floats(void);
size_t tnv3_inpaintnet_act_floats(int n);
size_t tnv3_inpaintnet_dpre_floats(int n);
size_t tnv3_inpaintnet_param_floats(void);
int tnv3_inpaintnet_pack_t(const float* const* weights9, float* packed_t, tnv3_stream_t stream);
int tnv3_inpaintnet_fused_forward_train(const float* x, const float* m, const float* packed, float* out, float* acts, int n, int l,
                                        tnv3_stream_t stream);
int tnv3_inpaintnet_fused_backward(const float* x, const float* m, const float* dout, const float* out, const float* acts,
                                   const float* packed, const float* packed_t, float* dpre, float* grads, int n, int l, tnv3_stream_t stream);

/* ---- heat-map post-process (predict.py:14-69,163-209; test.py:25-79) ------------------------------------- */

/* Temporal ensemble in closed form (predict.py:163-209 heat maps, 243-301 coordinates).
 *   win    : [n_local][L][E] window outputs; row i is global window s_base + i (sliding step 1)
 *   weight : [L] from get_ensemble_weight (test.py:25-50)
 *   out    : [n_frames][E] ensembled predictions of global frames t0 .. t0+n_frames-1
 *   num_sample = total number of windows of the video (frames - L + 1).  Every window a requested frame needs
 *   (s in [max(0,t-L+1), min(t,num_sample-1)]) must be resident in `win`.
 *   sum_order : the order in which the reference's `.sum(0)` (torch CPU) adds the L rows, so that results are bit-identical
 *   to predict.py's loops: 0 = sequential (heat maps: torch's vectorised outer sum), 1 = four interleaved partial sums
 *   (coordinates, E = 2: torch's scalar row_sum).  Products are rounded to fp32 before they are added (no FMA), as there. */
int tnv3_ensemble_frames(const float* win, int n_local, long s_base, int l, int e, const float* weight, long t0,
                         int n_frames, long num_sample, int sum_order, float* out, tnv3_stream_t stream);

/* Bytes of scratch tnv3_heatmap_peakfind needs for `frames` maps of h x w. */
size_t tnv3_peakfind_workspace_bytes(int frames, int h, int w);

/* predict.py:35 (`y_pred > 0.5`) + predict_location (test.py:52-79; cv2.findContours(RETR_EXTERNAL) +
 * cv2.boundingRect + largest box), batched: out_bbox[f] = (x, y, w, h) int32, or (0,0,0,0) for an empty map.
 *   heat : [frames][h][w] fp32;  a pixel is foreground iff heat > threshold
 *   tie_last_wins != 0: among equal-area boxes the component found last in raster order wins (OpenCV order) */
int tnv3_heatmap_peakfind(const float* heat, float threshold, int tie_last_wins, int32_t* out_bbox, void* workspace,
                          size_t workspace_bytes, int frames, int h, int w, tnv3_stream_t stream);

/* Per-map maximum inside a box: out[f] = max heat[f][y:y+h, x:x+w] with (x, y, w, h) = boxes[f] (int32, clipped to the
 * map; 0 for an empty box), or the maximum of the whole map when boxes == NULL.  Replaces the detection confidence
 * `np.amax(y_p[bbox...])` of evaluate() (test.py:164-167) and its `np.amax(y_t) > 0` ground-truth test (test.py:170-178),
 * so validation never copies heat maps to the host.  NaN propagates as in numpy. */
int tnv3_heatmap_box_max(const float* heat, const int32_t* boxes, float* out, int frames, int h, int w, tnv3_stream_t stream);

/* ---- training step (train.py:84-96: forward in train mode, WBCELoss, loss.backward()) -------------------- */

/* nn.BatchNorm2d in training mode (model.py:9) + nn.ReLU (model.py:10) on the raw convolution output z[N][C][HW]:
 * batch mean / biased variance over (N,H,W), a = max((z-mean)*invstd*gamma+beta, 0), running stats updated in
 * place with momentum and the UNBIASED variance; (mean, invstd) saved for backward.  HW % 4 == 0. */
size_t tnv3_bn_workspace_bytes(int channels);
int tnv3_bn_train_forward(const float* z, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float eps, float momentum, float* a, float* save_mean,
                          float* save_invstd, void* workspace, size_t workspace_bytes, int n, int c, int hw,
                          tnv3_stream_t stream);

/* The same with the batch statistics already reduced per pixel tile by the producing convolution (tile_stats [c][n_tiles][2] doubles
 * from tnv3_conv3x3_wino_forward_stats): fixed-order sums, then the identical finalize and normalise + ReLU pass -- one read of z
 * less per layer.  Same workspace as tnv3_bn_train_forward. */
int tnv3_bn_train_forward_tiles(const float* z, const double* tile_stats, long n_tiles, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float eps, float momentum, float* a, float* save_mean,
                                float* save_invstd, void* workspace, size_t workspace_bytes, int n, int c, int hw, tnv3_stream_t stream);

/* ABI 8.  ... for a block whose output is pooled next (a down block's last layer, model.py:47-48, 50-51, 53-54): the normalise + ReLU pass also
 * writes pooled = MaxPool2d(2, 2)(a) [n][c][h / 2][w / 2] -- the separate pooling pass and its read of a are gone.  h % 2 == 0, w % 4 == 0. */
int tnv3_bn_train_forward_tiles_pool(const float* z, const double* tile_stats, long n_tiles, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float eps, float momentum, float* a, float* pooled, float* save_mean,
                                     float* save_invstd, void* workspace, size_t workspace_bytes, int n, int c, int h, int w, tnv3_stream_t stream);

/* Backward of the same: given dA (gradient w.r.t. a), z and the saved statistics, writes dZ (may alias dA),
 * dgamma[C], dbeta[C].  The ReLU mask comes from `a` when it is given; with a == NULL it is recomputed from z, gamma, beta
 * and the saved statistics with the forward's own expression (bit-identical to a > 0, one tensor less to read per pass):
 * gamma / beta must then still hold the values of the forward call.  One of a, beta must be non-NULL. */
int tnv3_bn_relu_backward(const float* da, const float* a, const float* z, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, float* dz, float* dgamma, float* dbeta,
                          void* workspace, size_t workspace_bytes, int n, int c, int hw, tnv3_stream_t stream);

/* BatchNorm + ReLU backward in ONE pass over (dA, z): the two per-channel sums (sum g, sum g * xhat) come from the epilogue of the
 * data-gradient launch that produced dA (model.py:8-10 backwards: the next Conv2DBlock's dX IS this block's dA):
 *   tnv3_bn_bwd_consts              : c4 [c][4] = (mean, invstd, gamma * invstd as the passes round it, beta) of the block
 *   tnv3_conv3x3_wino_dgrad_bnstats : the Winograd data gradient (as tnv3_conv3x3_wino_forward on the transposed / flipped filter,
 *                                     kernel variants 3-6) that also writes tile_stats [cout][tiles][2] (doubles; tiles =
 *                                     tnv3_conv3x3_wino_stats_tiles(n, h, w, variant)) = per channel and pixel tile the sums over
 *                                     g = dA * [BN(z) > 0] and g * (z - mean) * invstd, taken from the epilogue's registers
 *                                     (bn_z = the block's raw convolution output z, bn_c4 from tnv3_bn_bwd_consts)
 *   tnv3_bn_relu_backward_tiles     : fixed-order reduction of the tiles, then tnv3_bn_relu_backward's finalize + apply.
 * Same dgamma / dbeta / dZ as tnv3_bn_relu_backward up to the fp64 summation order of the two sums. */
int tnv3_bn_bwd_consts(const float* save_mean, const float* save_invstd, const float* gamma, const float* beta, float* c4, int c,
                       tnv3_stream_t stream);
int tnv3_conv3x3_wino_dgrad_bnstats(const float* dz, const float* u, float* da, double* tile_stats, const float* bn_z, const float* bn_c4,
                                    int n, int cin, int cout, int h, int w, int variant, tnv3_stream_t stream);
int tnv3_bn_relu_backward_tiles(const float* da, const float* z, const float* gamma, const float* beta, const float* save_mean,
                                const float* save_invstd, const double* tile_stats, long n_tiles, float* dz, float* dgamma, float* dbeta,
                                void* workspace, size_t workspace_bytes, int n, int c, int hw, tnv3_stream_t stream);

/* Data gradient of Conv2DBlock's convolution: dX = conv3x3(dZ, W^T with flipped taps).  wpack_t comes from
 * tnv3_pack_conv3x3_weights(..., transpose_flip = 1).  The first c0 input-channel gradients go to dx0
 * [N][c0][H][W], the remaining c1 to dx1 [N][c1][H][W] (the two operands of torch.cat at model.py:65,67,69);
 * c1 = 0 / dx1 = NULL for an ordinary layer.  (c0 + c1) % 64 == 0. */
int tnv3_conv3x3_dgrad(const float* dz, const float* wpack_t, float* dx0, float* dx1, int n, int cout, int c0, int c1,
                       int h, int w, int cfg, tnv3_stream_t stream);

/* Weight gradient dW[Cout][C0+C1][3][3] = sum_pixels dZ * X, X = cat([up2x?(src0), src1]) as in the forward.
 * `variant` (per call; the workspace query must get the same value): -1 / 0 = register-staged 4x32-pixel tiles (default),
 * 1 = LDS-DMA staged, double-buffered 2x32-pixel tiles (same gradient, measured 8 % slower). */
size_t tnv3_conv3x3_wgrad_workspace_bytes(int n, int c0, int c1, int cout, int h, int w, int variant);
int tnv3_conv3x3_wgrad(const float* src0, const float* src1, const float* dz, float* dw, void* workspace,
                       size_t workspace_bytes, int n, int c0, int c1, int cout, int h, int w, int up0, int variant,
                       tnv3_stream_t stream);

/* WBCELoss(y_pred, y, reduce) (utils/metric.py:3-20): out[0] (reduce != 0) or out[N] per-sample means. */
size_t tnv3_wbce_workspace_bytes(int n);
int tnv3_wbce_forward(const float* p, const float* y, float* out, void* workspace, size_t workspace_bytes, int n,
                      long per_sample, int reduce, tnv3_stream_t stream);
/* dL/dp in closed form; upstream: gradient of the scalar (reduce != 0: 1 value) or of the N per-sample losses. */
int tnv3_wbce_backward(const float* p, const float* y, const float* upstream, float* dp, int n, long per_sample,
                       int reduce, tnv3_stream_t stream);

/* Backward of the head p = sigmoid(conv1x1(a) + b) (model.py:71-72): dA[N][64][HW], dW[L][64], db[L]. */
size_t tnv3_head_backward_workspace_bytes(int l);
int tnv3_head_backward(const float* dp, const float* p, const float* a, const float* w, float* da, float* dw,
                       float* db, void* workspace, size_t workspace_bytes, int n, int l, int hw,
                       tnv3_stream_t stream);

/* sigmoid + WBCELoss FUSED INTO THE HEAD (model.py:71-72 followed by utils/metric.py:15-20, train.py:92-93), both directions:
 *   forward : p = sigmoid(conv1x1(x) + b) is written once and the loss partial sums are taken from the registers that hold it
 *             (loss[0] with reduce != 0, else loss[n] per-sample means); y [n][l][hw] are the targets;  l <= 8, hw % 4 == 0.
 *   backward: dL/dp is formed on the fly from (p, y) and the loss node's upstream gradient (1 value, or n values when reduce == 0)
 *             inside the head's backward -- no dP tensor is written or read; outputs as tnv3_head_backward (same workspace). */
size_t tnv3_head_wbce_workspace_bytes(int n);
int tnv3_head1x1_sigmoid_wbce(const float* x, const float* w, const float* b, const float* y, float* p, float* loss, void* workspace,
                              size_t workspace_bytes, int n, int c, int l, int hw, int reduce, tnv3_stream_t stream);
int tnv3_head_wbce_backward(const float* y, const float* p, const float* a, const float* w, const float* upstream, float* da, float* dw,
                            float* db, void* workspace, size_t workspace_bytes, int n, int l, int hw, int reduce, tnv3_stream_t stream);

/* MaxPool2d(2,2) backward fused with the skip-connection gradient add: dx = dskip + route(dpool). dskip may be NULL. */
int tnv3_maxpool2x2_backward_add(const float* x, const float* dpool, const float* dskip, float* dx, long nc, int h,
                                 int w, tnv3_stream_t stream);
/* ABI 8.  The same routing where the pooled tensor is a = ReLU(BatchNorm(z)) of a block normalised in this step (model.py:9-10 in front of
 * model.py:48,51,54) -- the launch ALSO takes that block's two BatchNorm + ReLU backward sums of the dx it writes (autograd of model.py:9-10:
 * dbeta = sum g, dgamma = sum g * zhat, g = dx where a > 0): tile_stats [c][slices][2] doubles, slices = tnv3_maxpool2x2_bn_stats_tiles(n, h, w)
 * (0: shape not supported -- needs h % 2 == 0, w % 4 == 0), to be fed with dx to tnv3_bn_relu_backward_tiles.  It reads z, not a: a is
 * recomputed with the forward's expression (same bits, so the same first maximum per window), bn_* are that block's saved mean / invstd
 * and its affine parameters as the forward used them.  z / dskip / dx 16-byte aligned.  dskip may be NULL. */
int tnv3_maxpool2x2_bn_stats_tiles(int n, int h, int w);
int tnv3_maxpool2x2_backward_add_bnstats(const float* z, const float* dpool, const float* dskip, float* dx, const float* bn_mean,
                                         const float* bn_invstd, const float* bn_gamma, const float* bn_beta, double* tile_stats, int n, int c, int h,
                                         int w, tnv3_stream_t stream);
/* nn.Upsample(scale_factor=2) backward: d_lo = 2x2 block sums of d_hi [nc][2*hl][2*wl]. */
int tnv3_upsample2x_backward(const float* d_hi, float* d_lo, long nc, int hl, int wl, tnv3_stream_t stream);
/* Sample mixup (train.py:32-40): out[n] = x[n]*lam[n] + x[perm[n]]*(1-lam[n]); per_sample % 4 == 0. */
int tnv3_mixup(const float* x, const float* lam, const int32_t* perm, float* out, int n, long per_sample,
               tnv3_stream_t stream);

/* ---- optimiser step and mixup draws on the device (train.py:33-36, 85, 96, 165, 242-248) ----------
 * Tensor lists are HOST arrays of DEVICE pointers (+ element counts); they travel in the kernel-argument buffer, so a step
 * needs no device-side pointer table, no H2D copy and no host sync.  All tensors fp32, contiguous. */

/* torch.nn.utils.clip_grad_norm_(params, max_norm) (train.py:165), device side: out_norm_coef[0] = global L2 norm of all
 * gradients (fixed-order fp64 reduction: deterministic), out_norm_coef[1] = min(1, max_norm / (norm + 1e-6)).  The scaling
 * itself is applied by the optimiser kernels below through their `clip_coef` argument (pass out_norm_coef + 1). */
size_t tnv3_grad_norm_workspace_bytes(int count);
int tnv3_grad_norm(float* const* grads, const long* numel, int count, float max_norm, float* out_norm_coef, void* workspace,
                   size_t workspace_bytes, tnv3_stream_t stream);

/* torch.optim.Adam.step() (train.py:96 with the optimiser of train.py:242; amsgrad / maximize off) for `count` tensors in one
 * launch per 64 tensors, following torch's foreach implementation operation by operation in fp32:
 *   g *= clip (if clip_coef);  g += weight_decay * p;  m += (1-beta1) * (g - m);  v = v * beta2 + (1-beta2) * g * g;
 *   p += -(lr / (1 - beta1^step)) * (m / (sqrt(v) / sqrt(1 - beta2^step) + eps))
 * step: 1-based count of this update (the bias corrections are formed on the host in double, as torch does).
 * clip_coef: DEVICE pointer to the gradient scale (tnv3_grad_norm's out_norm_coef + 1) or NULL; with it the clipped gradient
 * is written back to grads (what clip_grad_norm_ leaves behind).  zero_grad != 0: grads are zeroed instead (train.py:85). */
int tnv3_adam_step(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const long* numel,
                   int count, double lr, double beta1, double beta2, double eps, double weight_decay, long step, const float* clip_coef,
                   int zero_grad, tnv3_stream_t stream);

/* torch.optim.SGD.step() (train.py:244: momentum 0.9; dampening 0, no Nesterov): buf = g on the first step, else
 * momentum * buf + g;  p -= lr * buf.  momentum_buf may be NULL when momentum == 0. */
int tnv3_sgd_step(float* const* params, float* const* grads, float* const* momentum_buf, const long* numel, int count, double lr,
                  double momentum, double weight_decay, int first_step, const float* clip_coef, int zero_grad, tnv3_stream_t stream);

/* The random draws of mixup (train.py:33-36) on the device: lam[n] = max(B, 1 - B) with B ~ Beta(alpha, alpha) (two Gamma
 * variates, Marsaglia-Tsang), perm[n] = a uniform random permutation (Fisher-Yates), from Philox4x32-10 keyed by `seed` with
 * counter `step` -- deterministic per (seed, step), no host RNG, no H2D copy.  (The reference draws with numpy / torch CPU
 * generators; the RNG stream is not part of parity -- tnv3_mixup applies whatever draws it is given.)  n <= 65536. */
int tnv3_mixup_draw(float* lam, int32_t* perm, int n, float alpha, uint64_t seed, uint64_t step, tnv3_stream_t stream);

/* ---- InpaintNet backward (train.py:156-164: autograd through Conv1DBlock / predictor) ------------------------ */

/* dPre[N][C][L] = dOut * act'(out); act as in tnv3_conv1d_k3_forward; nlc != 0: dOut/out are [N][L][C]. */
int tnv3_conv1d_act_backward(const float* dout, const float* out, float* dpre, int n, int c, int l, int act, int nlc,
                             tnv3_stream_t stream);
/* dX = conv1d_k3(dPre [N][cout][L], W^T flipped), W = the layer's forward filter [cout][c0+c1][3].  The first c0
 * channel gradients go to dx0 [N][c0][L], the other c1 to dx1 (NULL when c1 = 0).  accumulate bit 0 / bit 1: add to
 * dx0 / dx1 instead of overwriting (a skip tensor receives two gradients). */
int tnv3_conv1d_k3_dgrad(const float* dpre, const float* w, float* dx0, float* dx1, int n, int cout, int c0, int c1,
                         int l, int accumulate, tnv3_stream_t stream);
/* dW[cout][c0+c1][3] and db[cout] for X = cat([src0, src1]) ([N][C][L], or [N][L][C] when src_nlc); L <= 64. */
size_t tnv3_conv1d_k3_wgrad_workspace_bytes(int n, int c0, int c1, int cout);
int tnv3_conv1d_k3_wgrad(const float* src0, const float* src1, const float* dpre, float* dw, float* db, void* workspace,
                         size_t workspace_bytes, int n, int c0, int c1, int cout, int l, int src_nlc,
                         tnv3_stream_t stream);

/* ---- frame preprocessing in front of the network (dataset.py:101-105, 427-461) --------------- */

/* PIL `Image.resize((ow, oh))` (default BICUBIC, 8-bit two-pass fixed-point resample) of `frames` images
 * src[frames][h][w][c] (uint8, HWC) -- bit-exact -- fused with np.moveaxis(img, -1, 0) and `/= 255.`:
 *   dst_f32[frames][c][oh][ow] = lut[resized u8]   (lut: 256 floats = float32(float64(v) / 255.0)), and/or
 *   dst_u8 [frames][oh][ow][c]                     (either may be NULL)
 *   tmp: frames*h*ow*c bytes of scratch (the horizontally resized intermediate)
 *   xmin/xcnt/kkx[ow][ksize_x], ymin/ycnt/kky[oh][ksize_y]: Pillow's precompute_coeffs + normalize_coeffs_8bpc tables
 *   (ABI 8: every coefficient must fit 24 bits, |k| < 2^23 -- Pillow's are <= 1.13 * 2^22: the products are formed by full-rate 24-bit multiplies)
 *   (int32, device memory) for the two axes; they depend only on (w, ow) and (h, oh). */
int tnv3_resample_bicubic_u8(const unsigned char* src, unsigned char* tmp, float* dst_f32, unsigned char* dst_u8,
                             const int32_t* xmin, const int32_t* xcnt, const int32_t* kkx, int ksize_x,
                             const int32_t* ymin, const int32_t* ycnt, const int32_t* kky, int ksize_y, const float* lut,
                             int frames, int h, int w, int c, int oh, int ow, tnv3_stream_t stream);

/* np.median(frame_arr, 0) over t frames of bytes_per_frame bytes each (uint8).  median (may be NULL): the value cast
 * `.astype('uint8')` (bg_mode 'concat'); median_x2 (may be NULL): TWICE the float64 median as uint16 (an even count gives
 * x.5 medians), which is what the difference-frame modes subtract. */
int tnv3_median_u8(const unsigned char* frames, unsigned char* median, uint16_t* median_x2, int t, long bytes_per_frame,
                   tnv3_stream_t stream);

/* Difference frame of bg_mode 'subtract' / 'subtract_concat' (dataset.py:439, 443):
 * out[f][p] = uint8(sum_c |frames[f][p][c] - median[p][c]|) with the float median (truncate, wrap mod 256), 3 channels. */
int tnv3_absdiff_sum_u8(const unsigned char* frames, const uint16_t* median_x2, unsigned char* out, int frames_n,
                        long pixels, tnv3_stream_t stream);

/* Diagnostics (the register-only MFMA probe, timing twins with deliberately wrong results) are NOT part of this library:
 * they are built from the same sources into libtnv3_diag.so, declared in include/tracknetv3_hip_diag.h. */

#ifdef __cplusplus
}
#endif
#endif /* TRACKNETV3_HIP_H */
