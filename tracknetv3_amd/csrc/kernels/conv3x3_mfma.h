// conv3x3_mfma.h -- direct 3x3 convolution (stride 1, zero pad 1, NCHW fp32) as an implicit GEMM on
// the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain).
//
// Replaces the implicit ATen/cuDNN kernels behind the reference's
//   Conv2DBlock  = nn.Conv2d(k=3, padding='same', bias=False) -> BatchNorm2d -> ReLU   (model.py:4-16)
// and, through the dual-source loader, `torch.cat([nn.Upsample(scale_factor=2)(x), skip], dim=1)`
// (model.py:65,67,69) without ever materialising the upsampled / concatenated tensor.
//
// GEMM view per workgroup:  D[co, pix] = sum_{ci,kh,kw} Wp[ci,kh,kw,co] * X[ci, h+kh-1, w+kw-1]
//   M = MB output channels, N = TR x TC output pixels, K = Cin*9 walked in chunks of CC channels.
//   MFMA 32x32x2:  lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
//                  D reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
//   j = 32 consecutive pixels of one image row, so the B operand of lane l for tap (kh,kw) is one
//   ds_read_b32 at  lane_base + const  from the LDS-staged halo tile -- conflict free (each 32-lane half
//   reads 32 consecutive dwords).  The two K slots of one MFMA are two consecutive input channels at the
//   same tap, so all LDS addresses in the main loop are `per-lane base + immediate`.
//   Weights are pre-packed [Cin_pad][3][3][Cout] so the A operand is also a 32-consecutive-dword read.
//
// Pipeline: 2 LDS buffers; global loads of chunk k+1 are issued before the MFMA block of chunk k and
// written to the other LDS buffer after it (register staging), one barrier per chunk.
//
// Roofline: every layer here has fp32 arithmetic intensity >= 85 FLOP/B, so the kernel is
// bound by the 157.3 TFLOP/s fp32 MFMA rate, not by HBM.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Conv3x3Args {
  const float* src0;    // [N][C0][H0][W0]   H0,W0 = H,W (up0 == 0) or H/2,W/2 (up0 == 1: nearest 2x upsample on load)
  const float* src1;    // [N][C1][H][W] or nullptr; its channels follow src0's (cat([up(src0), src1], dim=1))
  const float* wpack;   // [Cin_pad][9][Cout], Cin_pad = roundup(C0+C1, 32), padding rows zero, then kPackZeroTail zeros
  const float* zeros;   // -> the zero tail of wpack (source of padded elements for the LDS-DMA path)
  const float* mean;    // [Cout] or nullptr  -> y = (acc - mean)*scale + shift: eval-mode BatchNorm in the reference's
  const float* scale;   // [Cout] or nullptr     own operation order (subtract first: no cancellation when |mean| >> std)
  const float* shift;   // [Cout] or nullptr
  float* dst;           // [N][Cout][H][W]   (or [N][csplit][H][W] when dst1 is set)
  float* dst1;          // optional second destination for output channels >= csplit: [N][Cout-csplit][H][W]
  int csplit;           //   (data gradient of a concat layer: d(up) and d(skip) land in separate tensors)
  int N, C0, C1, Cout, H, W;
  int up0;              // 1: src0 is stored at (H/2, W/2) and read as out[h][w] = src0[h>>1][w>>1]
  int relu;             // 1: y = max(y, 0)
  const float* addend;  // optional [N][Cout][H][W]: added to the accumulator before the affine / ReLU (the low-resolution half of a
                        // decoder-entry layer, conv_up2x_mfma.h); not combined with dst1
  int diag;             // honoured only by ConvCfg<..., DIAG = 1> instantiations (WRONG results by design): 1 = stage the first
                        // chunk only (no global loads / LDS stores afterwards), 2 = additionally no barriers.
};

template <int MT_, int NTW_, int WM_, int WN_, int TR_, int TC_, int CC_, int MINW_ = 1, int PF_ = 1, int PRIO_ = 0, int GLDS_ = 0,
          int DIAG_ = 0>
struct ConvCfg {
  static constexpr int DIAG = DIAG_;                 // 1: diagnostic instantiation that honours Conv3x3Args::diag (a runtime
                                                     //    flag in the production kernels cost 15 VGPRs = one workgroup per CU)
  static constexpr int GLDS = GLDS_;                 // 1: stage through the LDS-DMA path (global_load_lds): no staging VGPRs,
                                                     //    no ds_write; zero padding is read from the filter's zero tail
                                                     // 2: same, THREE LDS stages and counted vmcnt: the DMA of chunk k+2 stays
                                                     //    in flight across the barrier that publishes chunk k+1
  static constexpr int STAGES = GLDS_ == 2 ? 3 : 2;
  static constexpr int MT = MT_, NTW = NTW_, WM = WM_, WN = WN_, TR = TR_, TC = TC_, CC = CC_;
  static constexpr int MINW = MINW_;                 // __launch_bounds__ 2nd argument: waves per SIMD to fit
  static constexpr int PF = PF_;                     // K-steps of operand prefetch (LDS reads run PF steps ahead of the MFMAs)
  static constexpr int PRIO = PRIO_;                 // 1: s_setprio(1) around the MFMA block (co-resident workgroups are
                                                     //    in different phases; favour the wave that can feed the matrix pipe)
  static_assert(PF == 1 || PF == 2, "prefetch distance");
  static constexpr int NT = WM * WN * 64;            // threads per workgroup
  static constexpr int MB = MT * 32 * WM;            // output channels per workgroup
  static constexpr int CS = TC / 32;                 // 32-pixel column segments per tile row
  static_assert(TC % 32 == 0, "tile width must be a multiple of the MFMA N (32)");
  static_assert(TR * CS == NTW * WN, "pixel tile must equal waves x N-tiles");
  static_assert(NTW % CS == 0, "each wave must own whole tile rows");
  static_assert(CC % 2 == 0 && CC <= 32, "channel chunk must be even (one MFMA = 2 channels)");
  static constexpr int TRp = TR + 2, TCp = TC + 2, PLANE = TRp * TCp;
  static constexpr int E_IN = CC * PLANE;                       // halo-tile elements per chunk
  static constexpr int NIN = (E_IN + NT - 1) / NT;              // staged elements per thread
  static constexpr int IN_FLOATS = NIN * NT;                    // LDS floats (padded so every thread stores)
  static constexpr int KROWS = CC * 9;
  static constexpr int W_FLOATS = KROWS * MB;
  static constexpr int E_W4 = W_FLOATS / 4;
  static constexpr int NW4 = (E_W4 + NT - 1) / NT;
  static constexpr int BUF_FLOATS = W_FLOATS + IN_FLOATS;       // one pipeline stage
  static constexpr int LDS_BYTES = STAGES * BUF_FLOATS * 4;
  // LDS-DMA instructions one wave issues per chunk (GLDS == 2 counts them in vmcnt): NIN input pieces, the full filter
  // pieces, and -- for the waves below W_TAIL_WAVES -- the partial last filter piece (wave-uniform: E_W4 % 64 == 0).
  static constexpr int DMA_PER_CHUNK = NIN + E_W4 / NT;
  static constexpr int W_TAIL_WAVES = (E_W4 % NT) / 64;
  static_assert(GLDS_ != 2 || E_W4 % 64 == 0, "filter pieces must split on wave boundaries");
};

// Block index -> (output-channel block, pixel tile).  Workgroup b is observed to run on XCD b % 8 (speed
// only, never correctness): XCDs are split into nMB groups so that one XCD's L2 holds one weight panel, and
// each XCD walks a contiguous range of pixel tiles so vertical halo rows are L2 hits.
__device__ __forceinline__ bool conv_block_map(int b, int nMB, int nPT, int& mb, int& pt) {
  if (nMB <= 8 && (8 % nMB) == 0) {
    const int G = 8 / nMB;                      // XCDs per channel block
    const int per = (nPT + G - 1) / G;          // pixel tiles per XCD
    const int xcd = b & 7, q = b >> 3;
    mb = xcd % nMB;
    pt = (xcd / nMB) * per + q;
    return q < per && pt < nPT;
  }
  mb = b % nMB;
  pt = b / nMB;
  return pt < nPT;
}

__host__ __device__ inline int conv_grid_blocks(int nMB, int nPT) {
  if (nMB <= 8 && (8 % nMB) == 0) {
    const int G = 8 / nMB;
    return 8 * ((nPT + G - 1) / G);
  }
  return nMB * nPT;
}

// ---- raw buffer descriptors: LDS-DMA / loads / stores whose per-lane address part is one 32-bit offset (used by the Winograd
//      forward kernels of conv3x3_wino3_mfma.h and the Winograd weight-gradient kernel of wgrad_wino_mfma.h)
#ifndef TNV3_EMU
typedef __amdgpu_buffer_rsrc_t tnv3_rsrc_t;
// Raw buffer descriptor (stride 0): base, size in bytes (< 2^31); loads at offset >= num_records return 0.
__device__ __forceinline__ tnv3_rsrc_t tnv3_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// LDS[lds_base + lane*16 .. +16) <- buffer[voffset .. +16) (or zeros when out of range); tracked by vmcnt.
__device__ __forceinline__ void tnv3_buf_dma16(tnv3_rsrc_t r, float* lds_base, unsigned voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, (int)voffset, 0, 0, 0);
}
// 8-byte load / store at base + voffset (per lane, bytes) + soffset (scalar, bytes): no vector address arithmetic per access
typedef float tnv3_f2 __attribute__((ext_vector_type(2)));
typedef unsigned tnv3_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tnv3_f2 tnv3_buf_load_f2(tnv3_rsrc_t r, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(tnv3_f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voffset, (int)soffset, 0));
}
__device__ __forceinline__ void tnv3_buf_store_f2(tnv3_rsrc_t r, unsigned voffset, unsigned soffset, tnv3_f2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(tnv3_u2, v), r, (int)voffset, (int)soffset, 0);
}
// 16-byte load at base + voffset (per lane) + soffset (scalar / immediate): one 32-bit VGPR of address instead of a 64-bit pair per access
typedef float tnv3_f4 __attribute__((ext_vector_type(4)));
typedef unsigned tnv3_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tnv3_f4 tnv3_buf_load_f4(tnv3_rsrc_t r, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(tnv3_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, (int)soffset, 0));
}
// 16-byte store.  gfx950 HAZARD (measured, scripts/wino43_debug.py): a VALU write of the data registers in the instruction right after
// a >8-byte buffer store corrupts the stored value -- the store reads its data over several cycles, gfx940+ needs TWO wait states, and
// the compiler's hazard recogniser exempts the SGPR-soffset form (its extra issue cycle covers only one).  The s_nop takes the data
// registers as operands: they stay live, and unwritten, until it has issued (tests/test_kernel_resources.py scans the ISA for the pattern).
__device__ __forceinline__ void tnv3_buf_store_f4(tnv3_rsrc_t r, unsigned voffset, unsigned soffset, tnv3_f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tnv3_u4, v), r, (int)voffset, (int)soffset, 0);
  asm volatile("s_nop 1" : : "v"(v) : "memory");
}
#endif

// Makes a per-lane value opaque to the optimiser at this point: what is derived from it afterwards cannot be hoisted out of the
// enclosing loop (the persistent kernel's per-tile code would otherwise park ~80 loop-invariant VGPRs across the MFMA loop).
#ifdef TNV3_EMU
#define TNV3_OPAQUE_V(x) ((void)0)
#define TNV3_OPAQUE_S(x) ((void)0)
#define TNV3_NO_IF_CONVERSION() ((void)0)
#else
#define TNV3_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define TNV3_OPAQUE_S(x) asm volatile("" : "+s"(x))                 /* the same for a wave-uniform value */
#define TNV3_NO_IF_CONVERSION() asm volatile("" ::: "memory")      /* inside a conditional block: it cannot be speculated into selects */
#endif

constexpr unsigned kDmaOob = 0x80000000u;      // voffset of a padding lane: beyond any descriptor (num_records < 2^31)

// A persistent workgroup's walk through the block list: entries b0, b0 + G, b0 + 2G, ... of conv_block_map's order, decoded into
// (channel block, image, tile row, tile column) WITHOUT a division per step (six scalar divisions cost ~1000 cycles per tile on
// the SALU; the walk is a handful of adds and compares).  G must be a multiple of 8 when the XCD-aware order applies.
struct ConvTileWalk {
  int nMB, nPT, tilesH, tilesW;
  int mb, pt, n, trow, tcol;          // current entry
  int q, per;                         // XCD-aware order: position inside the XCD's run and the run length
  int d_mb, d_pt, d_n, d_row, d_col;  // per-step advance (d_pt decomposed into image / row / column steps)
  bool xcd, valid;
  __host__ __device__ void init(int b0, int G, int nMB_, int nPT_, int tilesH_, int tilesW_) {
    nMB = nMB_; nPT = nPT_; tilesH = tilesH_; tilesW = tilesW_;
    xcd = nMB <= 8 && (8 % nMB) == 0;
    if (xcd) {
      const int g = 8 / nMB;
      per = (nPT + g - 1) / g;
      const int x = b0 & 7;
      q = b0 >> 3;
      mb = x % nMB;
      pt = (x / nMB) * per + q;
      d_mb = 0; d_pt = G >> 3;
      valid = q < per && pt < nPT;
    } else {
      per = 0; q = 0;
      mb = b0 % nMB; pt = b0 / nMB;
      d_mb = G % nMB; d_pt = G / nMB;
      valid = pt < nPT;
    }
    const int tpi = tilesH * tilesW;
    n = pt / tpi;
    const int rem = pt - n * tpi;
    trow = rem / tilesW; tcol = rem - trow * tilesW;
    d_n = d_pt / tpi;
    const int drem = d_pt - d_n * tpi;
    d_row = drem / tilesW; d_col = drem - d_row * tilesW;
  }
  __host__ __device__ void bump_col() { if (++tcol >= tilesW) { tcol = 0; if (++trow >= tilesH) { trow = 0; ++n; } } }
  __host__ __device__ void next() {
    pt += d_pt;
    tcol += d_col; if (tcol >= tilesW) { tcol -= tilesW; ++trow; }
    trow += d_row; if (trow >= tilesH) { trow -= tilesH; ++n; }
    n += d_n;
    if (xcd) {
      q += d_pt;
      valid = q < per && pt < nPT;
    } else {
      mb += d_mb;
      if (mb >= nMB) { mb -= nMB; ++pt; bump_col(); }
      valid = pt < nPT;
    }
  }
};

constexpr int kPackZeroTail = 64;      // floats of zeros appended to every packed filter

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9+ encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
// ... and the one that waits for lgkmcnt <= n only (vmcnt = 63: no wait)
__host__ __device__ constexpr int tnv3_lgkmcnt_only(int n) { return 15 | (7 << 4) | ((n & 15) << 8) | (3 << 14); }
__host__ __device__ constexpr int tnv3_vmcnt_only(int n) { return (n & 15) | (7 << 4) | (15 << 8) | (((n >> 4) & 3) << 14); }

// global -> LDS DMA: LDS[lds_base + lane*BYTES] <- *gsrc (per-lane source, wave-uniform LDS base; the builtin puts the base
// in M0).  Completion is tracked by vmcnt; the compiler drains it before the workgroup barrier that publishes the stage.
// (the size operand of the builtin must be a literal, hence two helpers)
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma4(const float* gsrc, float* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT, Cfg::MINW) conv3x3_mfma_kernel(const Conv3x3Args a) {
  constexpr int MT = Cfg::MT, NTW = Cfg::NTW, WN = Cfg::WN, TR = Cfg::TR, TC = Cfg::TC, CC = Cfg::CC;
  constexpr int NT = Cfg::NT, MB = Cfg::MB, CS = Cfg::CS, TRp = Cfg::TRp, TCp = Cfg::TCp, PLANE = Cfg::PLANE;
  constexpr int NIN = Cfg::NIN, NW4 = Cfg::NW4, KROWS = Cfg::KROWS;

  __shared__ __attribute__((aligned(16))) float lds[Cfg::STAGES * Cfg::BUF_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int half = lane >> 5, bl = lane & 31;

  const int H = a.H, W = a.W, Cout = a.Cout, C0 = a.C0, C1 = a.C1;
  const int Cin = C0 + C1;
  const int tilesH = (H + TR - 1) / TR, tilesW = (W + TC - 1) / TC;
  const int nPT = a.N * tilesH * tilesW, nMB = Cout / MB;
  int mb, pt;
  if (!conv_block_map(blockIdx.x, nMB, nPT, mb, pt)) return;   // whole workgroup leaves together
  const int n = pt / (tilesH * tilesW);
  const int trem = pt - n * (tilesH * tilesW);
  const int h0 = (trem / tilesW) * TR, w0 = (trem % tilesW) * TC;
  const int m0 = mb * MB;

  const int HW = H * W;
  const int H0 = a.up0 ? (H >> 1) : H, W0 = a.up0 ? (W >> 1) : W;
  const int HW0 = H0 * W0;

  // ---- per-thread staging slots: element e = tid + i*NT of the [CC][TRp][TCp] halo tile
  // packed as c<<26 | gh<<13 | gw, or -1 when the slot is outside the image (zero padding) / unused
  int sp[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int e = tid + i * NT;
    const int c = e / PLANE, r = e - c * PLANE;
    const int tr = r / TCp, tc = r - tr * TCp;
    const int gh = h0 - 1 + tr, gw = w0 - 1 + tc;
    const bool ok = (e < Cfg::E_IN) && gh >= 0 && gh < H && gw >= 0 && gw < W;
    sp[i] = ok ? ((c << 26) | (gh << 13) | gw) : -1;
  }

  float rin[NIN];
  f32x4 rw[NW4];

  // Validity of slot i for chunk k (zero padding outside the image / beyond the last input channel).
  auto slot_ok = [&](int i, int k) -> bool {
    const int s = sp[i];
    const int cbeg = k * CC;
    const int climit = (cbeg < C0 ? C0 : Cin) - cbeg;
    return (s != -1) && ((int)((unsigned)s >> 26) < climit);
  };

  // Issue the global loads of chunk k into registers.  Nothing here consumes a loaded value, so the loads
  // stay in flight across the MFMA block that follows; the zero-padding select happens in store_stage.
  auto load_stage = [&](int k) {
    const int cbeg = k * CC;
    const bool from0 = cbeg < C0;
    const float* base = from0 ? a.src0 + ((size_t)n * C0 + cbeg) * HW0
                              : a.src1 + ((size_t)n * C1 + (cbeg - C0)) * HW;
    if (from0 && a.up0) {
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        const int s = sp[i];
        const int c = (int)((unsigned)s >> 26), gh = (s >> 13) & 8191, gw = s & 8191;
        const int off = slot_ok(i, k) ? (c * HW0 + (gh >> 1) * W0 + (gw >> 1)) : 0;
        rin[i] = base[off];
      }
    } else {
      const int hw = from0 ? HW0 : HW;   // == HW when not upsampled
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        const int s = sp[i];
        const int c = (int)((unsigned)s >> 26), gh = (s >> 13) & 8191, gw = s & 8191;
        const int off = slot_ok(i, k) ? (c * hw + gh * W + gw) : 0;
        rin[i] = base[off];
      }
    }
    const float* wsrc = a.wpack + (size_t)k * KROWS * Cout + m0;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * NT;
      const int krow = e4 / (MB / 4), m4 = e4 - krow * (MB / 4);
      if ((i + 1) * NT <= Cfg::E_W4 || e4 < Cfg::E_W4)
        rw[i] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)krow * Cout + m4 * 4);
    }
  };

  auto store_stage = [&](int buf, int k) {
    float* lw = lds + buf * Cfg::BUF_FLOATS;
    float* li = lw + Cfg::W_FLOATS;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * NT;
      if ((i + 1) * NT <= Cfg::E_W4 || e4 < Cfg::E_W4) *reinterpret_cast<f32x4*>(lw + e4 * 4) = rw[i];
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) li[tid + i * NT] = slot_ok(i, k) ? rin[i] : 0.0f;
  };

  // LDS-DMA staging of chunk k straight into stage `buf` (Cfg::GLDS): one 16-byte piece per lane for the weight panel,
  // one dword per lane for the halo tile; padded / out-of-image elements are fetched from the filter's zero tail.
  auto dma_stage = [&](int k, int buf) {
    float* lw = lds + buf * Cfg::BUF_FLOATS;
    float* li = lw + Cfg::W_FLOATS;
    const int wbase = wave * 64;                     // wave-uniform part of the element index
    const float* wsrc = a.wpack + (size_t)k * KROWS * Cout + m0;
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * NT;
      const int krow = e4 / (MB / 4), m4 = e4 - krow * (MB / 4);
      if ((i + 1) * NT <= Cfg::E_W4 || e4 < Cfg::E_W4)
        lds_dma16(wsrc + (size_t)krow * Cout + m4 * 4, lw + (i * NT + wbase) * 4);
    }
    const int cbeg = k * CC;
    const bool from0 = cbeg < C0;
    const float* base = from0 ? a.src0 + ((size_t)n * C0 + cbeg) * HW0
                              : a.src1 + ((size_t)n * C1 + (cbeg - C0)) * HW;
    const bool up = from0 && a.up0;
    const int hw = from0 ? HW0 : HW, wrow = up ? W0 : W;
    const int sh = up ? 1 : 0;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int s = sp[i];
      const int c = (int)((unsigned)s >> 26), gh = (s >> 13) & 8191, gw = s & 8191;
      const float* src = slot_ok(i, k) ? base + (c * hw + (gh >> sh) * wrow + (gw >> sh)) : a.zeros;
      lds_dma4(src, li + i * NT + wbase);
    }
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.0f;

  // per-lane LDS bases (floats) inside one stage
  const int a_off = half * 9 * MB + wm * (MT * 32) + bl;
  const int b_off = Cfg::W_FLOATS + (half * TRp + wn * (NTW / CS)) * TCp + bl;

  const int nChunks = (Cin + CC - 1) / CC;
  if (Cfg::GLDS == 2) {
    // ---- 3-stage LDS-DMA pipeline with counted vmcnt (CDNA guide T3/T4): never drain to 0 inside the loop
    const bool wtail = wave < Cfg::W_TAIL_WAVES;
    auto wait_all_but_newest = [&](bool newest_in_flight) {
      if (!newest_in_flight) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(0));
      else if (wtail) __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(Cfg::DMA_PER_CHUNK + 1));
      else __builtin_amdgcn_s_waitcnt(tnv3_vmcnt_only(Cfg::DMA_PER_CHUNK));
    };
    dma_stage(0, 0);
    if (nChunks > 1) dma_stage(1, 1);
    wait_all_but_newest(nChunks > 1);                          // chunk 0 landed, chunk 1 still in flight
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nChunks; ++k) {
      const int st = k % 3;
      if (k + 2 < nChunks) dma_stage(k + 2, (k + 2) % 3);     // that stage was last read before the previous barrier
      const float* A = lds + st * Cfg::BUF_FLOATS + a_off;
      const float* B = lds + st * Cfg::BUF_FLOATS + b_off;
      constexpr int NSTEP = (CC / 2) * 9;
      constexpr int RING = Cfg::PF + 1;
      float av[RING][MT], bv[RING][NTW];
      auto read_step = [&](int s, float (&ar)[MT], float (&br)[NTW]) {
        const int cp = s / 9, tap = s - 9 * cp;
        const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ar[mt] = A[(2 * cp * 9 + tap) * MB + mt * 32];
#pragma unroll
        for (int j = 0; j < NTW; ++j) br[j] = B[(2 * cp * TRp + kh + j / CS) * TCp + kw + (j % CS) * 32];
      };
      if (Cfg::PRIO) __builtin_amdgcn_s_setprio(1);
      read_step(0, av[0], bv[0]);
      if (Cfg::PF == 2) read_step(1, av[1], bv[1]);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s + Cfg::PF < NSTEP) read_step(s + Cfg::PF, av[(s + Cfg::PF) % RING], bv[(s + Cfg::PF) % RING]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING][mt], bv[s % RING][j], acc[mt][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
      }
      if (Cfg::PRIO) __builtin_amdgcn_s_setprio(0);
      // publish chunk k+1: this wave's DMAs of chunk k+1 have landed when at most the DMA_PER_CHUNK pieces of chunk k+2
      // are still outstanding; the barrier extends that to every wave's pieces and retires all reads of stage `st`.
      wait_all_but_newest(k + 2 < nChunks);
      __builtin_amdgcn_s_barrier();
    }
  } else {
  if (Cfg::GLDS) {
    dma_stage(0, 0);
  } else {
    load_stage(0);
    store_stage(0, 0);
  }
  __syncthreads();

  for (int k = 0; k < nChunks; ++k) {
    const int diag = Cfg::DIAG ? a.diag : 0;          // folds to 0 in production instantiations
    const int buf = diag ? 0 : (k & 1);
    if (k + 1 < nChunks && !diag) {
      if (Cfg::GLDS) dma_stage(k + 1, buf ^ 1);      // the other stage was last read before the previous barrier
      else load_stage(k + 1);
    }

    const float* A = lds + buf * Cfg::BUF_FLOATS + a_off;
    const float* B = lds + buf * Cfg::BUF_FLOATS + b_off;
    // K-steps of this chunk: step s = (channel pair cp, tap); operands of step s+1 are read from LDS before
    // the MFMAs of step s are issued (static double buffer), so LDS latency hides under the matrix pipe.
    constexpr int NSTEP = (CC / 2) * 9;
    constexpr int RING = Cfg::PF + 1;
    float av[RING][MT], bv[RING][NTW];
    auto read_step = [&](int s, float (&ar)[MT], float (&br)[NTW]) {
      const int cp = s / 9, tap = s - 9 * cp;
      const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ar[mt] = A[(2 * cp * 9 + tap) * MB + mt * 32];
#pragma unroll
      for (int j = 0; j < NTW; ++j) br[j] = B[(2 * cp * TRp + kh + j / CS) * TCp + kw + (j % CS) * 32];
    };
    if (Cfg::PRIO) __builtin_amdgcn_s_setprio(1);
    read_step(0, av[0], bv[0]);
    if (Cfg::PF == 2) read_step(1, av[1], bv[1]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + Cfg::PF < NSTEP) read_step(s + Cfg::PF, av[(s + Cfg::PF) % RING], bv[(s + Cfg::PF) % RING]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % RING][mt], bv[s % RING][j], acc[mt][j], 0, 0, 0);
      // pin the order "LDS reads of step s+1, then the MFMAs of step s" in the machine scheduler
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);   // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);   // MFMA
    }
    if (Cfg::PRIO) __builtin_amdgcn_s_setprio(0);

    if (k + 1 < nChunks && !diag && !Cfg::GLDS) store_stage(buf ^ 1, k + 1);
    if (diag < 2) __syncthreads();
  }
  }   // 2-stage pipelines

  // ---- epilogue: affine (folded eval-mode BN) + ReLU, 128-byte row segments per half-wave
  const bool has_affine = a.scale != nullptr;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wm * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float mu = 0.0f, sc = 1.0f, sh = 0.0f;
      if (has_affine) { mu = a.mean ? a.mean[co] : 0.0f; sc = a.scale[co]; sh = a.shift[co]; }
      float* drow = (a.dst1 == nullptr || co < a.csplit)
                        ? a.dst + ((size_t)n * (a.dst1 ? a.csplit : Cout) + co) * HW
                        : a.dst1 + ((size_t)n * (Cout - a.csplit) + (co - a.csplit)) * HW;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int oh = h0 + wn * (NTW / CS) + j / CS;
        const int ow = w0 + (j % CS) * 32 + bl;
        if (oh < H && ow < W) {
          float v = acc[mt][j][r];
          if (a.addend) v += a.addend[((size_t)n * Cout + co) * HW + oh * W + ow];
          if (has_affine) v = (v - mu) * sc + sh;
          if (a.relu) v = v > 0.0f ? v : 0.0f;
          drow[oh * W + ow] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight packing:  W[Cout][Cin][3][3]  ->  Wp[Cin_pad][3][3][Cout]              (forward)
//                  W[Cout][Cin][3][3]  ->  Wp[Cout_pad][3][3][Cin] with taps flipped  (dgrad: dX = conv(dY, W^T flipped))
// One thread per packed element; rows >= the real K extent are zero.
inline __global__ void pack_conv3x3_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                            int Cout, int Cin, int Kpad, int transpose_flip) {
  const int M = transpose_flip ? Cin : Cout;     // packed inner (GEMM M) extent
  const int Kc = transpose_flip ? Cout : Cin;    // packed outer (GEMM K channels) extent
  const long body = (long)Kpad * 9 * M, total = body + kPackZeroTail;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    if (e >= body) { wp[e] = 0.0f; continue; }        // zero tail: padding source of the LDS-DMA loader
    const int m = (int)(e % M);
    const long t = e / M;
    const int tap = (int)(t % 9);
    const int kc = (int)(t / 9);
    float v = 0.0f;
    if (kc < Kc) {
      if (transpose_flip) v = w[((long)kc * Cin + m) * 9 + (8 - tap)];
      else v = w[((long)m * Cin + kc) * 9 + tap];
    }
    wp[e] = v;
  }
}

// Eval-mode BatchNorm2d (model.py:9): y = (x - running_mean) * scale + beta with
//   scale = gamma / sqrt(running_var + eps)      (the mean is NOT folded into the shift: see the epilogue note)
inline __global__ void bn_eval_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ rvar, float eps,
                                     float* __restrict__ scale, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) scale[c] = gamma[c] / sqrtf(rvar[c] + eps);
}

}  // namespace tnv3
