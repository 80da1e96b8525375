/* tracknetv3_hip_diag.h -- C ABI of libtnv3_diag.so: measurement tools built from the same kernel sources as
 * libtnv3_hip.so (csrc/tnv3_capi.hip with -DTNV3_DIAG -DTNV3_TU_DIAG).  NOT loaded by the product package: the scripts under
 * scripts/ bind it explicitly (scripts/diaglib.py).  The "timing twin" entry points run a production kernel with parts of
 * its pipeline switched off and therefore write WRONG results by design -- which is why they do not live in the product
 * library.  Same conventions as include/tracknetv3_hip.h (device pointers, caller-owned buffers, enqueue-only).
 */
#ifndef TRACKNETV3_HIP_DIAG_H
#define TRACKNETV3_HIP_DIAG_H

#include "tracknetv3_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Error text of the calling thread's last failed tnv3_diag_* call (this library's own buffer). */
const char* tnv3_diag_last_error(void);

/* Register-only v_mfma_f32_32x32x2_f32 loop: `blocks` workgroups of 256 threads, each wave issuing iters*8 MFMAs
 * (2*32*32*2 FLOP each).  out: blocks*256 floats (keeps the work observable).  Measures the sustained fp32 matrix
 * rate of the chip as clocked under load, to set next to the conv kernels' TFLOP/s. */
int tnv3_diag_mfma_f32_probe(float* out, int blocks, int iters, tnv3_stream_t stream);

/* The direct conv kernel (tile configurations 10, 11, 12) with parts of its pipeline switched off (results are WRONG by
 * design): diag 1 = stage only the first channel chunk (no global loads / LDS stores afterwards), diag 2 = additionally no
 * workgroup barriers; diag 0 = the production kernel.  Timing these against each other attributes the matrix-pipe idle time
 * to staging / synchronisation / the MFMA loop. */
int tnv3_diag_conv3x3_forward(const float* src0, const float* wpack, float* dst, int n, int c0, int cout, int h, int w,
                              int cfg, int diag, tnv3_stream_t stream);

/* The Winograd forward kernels and their timing twins, raw convolution (no addend / affine / ReLU):
 *   variant 0, 1, 2     the production kernels (as tnv3_conv3x3_wino_forward)
 *   variant 11, 12, 13  twins of kernel 0: no DMA after the prologue / + no patch transform / + no barriers
 *   variant 21 .. 26    twins of kernel 2: 21-23 as above; with DMA: 24 no patch transform, 25 transform without its V writes,
 *                       26 transform without its raw reads
 *   variant 27 / 37 ..  kernel 2 / kernel 3 with s_memtime phase totals of one mid-grid workgroup written to dst (uint64
 *                       [wave][8]; scripts/wino_timeline.py); 41, 51, 61, 71 ...: experimental schedules of kernel 3 */
int tnv3_diag_conv3x3_wino_forward(const float* src, const float* u, float* dst, int n, int cin, int cout, int h, int w,
                                   int variant, tnv3_stream_t stream);

/* tnv3_conv3x3_wino43_forward (plain: no addend, no affine; results correct) with s_memtime totals per phase of one mid-grid
 * workgroup in tl_out as uint64 [wave 8][8]: 0 tile fill (DMA wait + first transform + two barriers), 1 chunk loop, 2 next tile's
 * offsets + raw issue, 3 write-out, 4 A issue + loop tail, 5 chunks, 6 tiles walked (scripts/wino43_timeline.py).  variant 1: the
 * product kernel's work; 2 output stores dropped, 3 no LDS exchange in the write-out, 4 both (WRONG results: what the write-out costs). */
int tnv3_diag_conv3x3_wino43_timeline(const float* src, const float* u, float* dst, unsigned long long* tl_out, int n, int cin, int cout,
                                      int h, int w, int variant, tnv3_stream_t stream);

/* The 16x16x4 F(4x4) kernel (tnv3_conv3x3_wino43_forward variant 0 / 2; plain: no addend, no affine) with s_memtime totals of one mid-grid
 * workgroup in tl_out as uint64 [wave 8][8]: 0 prologue, 1 steps, 2 write-outs, 3 steps walked, 4 tiles walked
 * (scripts/wino43s_timeline.py).  cbw: 4 = 64 channels x 2 tile rows per workgroup, 8 = 128 x 1; grow: filter quads of the next
 * step requested at the end of a step (5 = a uniform five-pair ring); ts: first slot of the patch transform.  mask 0: the product kernel's work (results correct);
 * else timing twins (WRONG results): bit 0 no raw DMA, 1 no patch transform, 2 no A loads, 3 no B reads, 4 no MFMAs, 5 no output
 * stores (only the combinations scripts/wino43s_timeline.py uses are instantiated). */
int tnv3_diag_conv3x3_wino43s_timeline(const float* src, const float* u, float* dst, unsigned long long* tl_out, int n, int cin, int cout,
                                       int h, int w, int cbw, int grow, int ts, int mask, tnv3_stream_t stream);

/* tnv3_conv3x3_wgrad_wino with the timing twins of its third-generation kernel (kernels/wgrad_wino_mfma.h: WgradWino3Cfg<3, DIAG>):
 * variant 101 no operand transforms, 102 no strip DMA, 103 no MFMAs (operand reads kept); 0-3 as in the product library. */
int tnv3_diag_conv3x3_wgrad_wino(const float* x, const float* dz, float* dw, void* workspace, size_t workspace_bytes, int n, int cin,
                                 int cout, int h, int w, int variant, tnv3_stream_t stream);

/* What an instruction costs next to a stream of v_mfma_f32_32x32x2_f32 (kernels/coissue_probe.h): `blocks` workgroups of 8
 * waves (one per CU; waves w and w + 4 share a SIMD); waves 0-3 run role_a, waves 4-7 role_b, `iters` x 8 steps each; out
 * [blocks][8] uint64 = each wave's s_memtime cycles.  gsrc_1mb: any device buffer of >= 1 MiB (source of the LDS-DMA role). */
int tnv3_diag_coissue_probe(unsigned long long* out, const float* gsrc_1mb, int blocks, int role_a, int role_b, int iters,
                            tnv3_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRACKNETV3_HIP_DIAG_H */
