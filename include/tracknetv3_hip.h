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
#define TNV3_ABI_VERSION 1

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

/* Eval-mode nn.BatchNorm2d (model.py:9) as y = x*scale + shift per channel. */
int tnv3_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                 float eps, float* scale, float* shift, int channels, tnv3_stream_t stream);

/* dst[N][Cout][H][W] = act( conv3x3( cat([up2x?(src0), src1], dim=1), W ) * scale + shift )
 *   src0 : [N][C0][H][W], or [N][C0][H/2][W/2] when up0 != 0 (nn.Upsample(scale_factor=2), nearest)
 *   src1 : [N][C1][H][W] or NULL (C1 = 0); its channels follow src0's, as torch.cat([up, skip], dim=1)
 *   wpack: from tnv3_pack_conv3x3_weights for Cin = C0 + C1
 *   scale/shift: [Cout] or both NULL (raw convolution);  relu != 0 applies max(.,0)
 *   cfg  : tile configuration index, or -1 to let the library choose
 * Requirements: Cout % 64 == 0; H,W < 8192; when C1 > 0, C0 % 32 == 0.  */
int tnv3_conv3x3_forward(const float* src0, const float* src1, const float* wpack, const float* scale,
                         const float* shift, float* dst, int n, int c0, int c1, int cout, int h, int w,
                         int up0, int relu, int cfg, tnv3_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* TRACKNETV3_HIP_H */
