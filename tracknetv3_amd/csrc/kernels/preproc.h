// preproc.h -- frame preprocessing in front of TrackNet, integer / byte work bound by HBM:
//   * Pillow's 8-bit two-pass BICUBIC resample (`Image.resize`, dataset.py:450,629): horizontal pass u8 -> u8, vertical
//     pass u8 -> u8 fused with `np.moveaxis(img, -1, 0)` and `frames /= 255.` (dataset.py:451-459) -> fp32 CHW.
//     Fixed-point: acc = (1 << 21) + sum_k pixel[k] * coeff[k] (coeff scaled by 2^22), clip8(acc >> 22) -- bit-exact with
//     src/libImaging/Resample.c.  Coefficient tables come from the host (they depend only on the two sizes).
//   * temporal median background (`np.median(frame_arr, 0).astype('uint8')`, dataset.py:101-105): per byte position a
//     256-bin histogram over the T frames in LDS, then the two middle order statistics, (a + b) >> 1.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

constexpr int kResampleBits = 22;
constexpr int kResampleMaxRowBytes = 32 * 1024;     // one source row (W*C bytes) staged in LDS
constexpr int kResamplePersistRowBytes = 6144;             // one staged row of the persistent horizontal pass

// pixel (8 bits) x coefficient (Pillow's are <= 2^22 in magnitude: normalised weights scaled by 2^22; the contract of the tables is |k| < 2^23) as
// a 24-bit multiply: v_mad_i32_i24 runs at full rate, the 32-bit v_mul_lo_u32 the compiler must otherwise emit at a quarter of it -- the
// horizontal pass was 245 of those per thread and row group (round 6; the pass did not get faster by it alone: it is not bound by its arithmetic).  Same integers.
__device__ __forceinline__ int resample_mul24(int pixel, int coeff) {
#ifdef TNV3_EMU
  return pixel * coeff;
#else
  return __mul24(pixel, coeff);
#endif
}

// a volatile 32-bit view of LDS that stays an LDS pointer (a volatile access through a generic pointer is a flat load)
#ifdef TNV3_EMU
typedef const volatile unsigned* resample_lds_cvu_p;
#else
typedef const volatile __attribute__((address_space(3))) unsigned* resample_lds_cvu_p;
#endif
// ({hi, lo} >> 8 s) as 32 bits, s = 0 .. 3: v_alignbyte_b32
__device__ __forceinline__ unsigned resample_alignbyte(unsigned hi, unsigned lo, unsigned s) {
#ifdef TNV3_EMU
  return (unsigned)(((((unsigned long long)hi) << 32) | (unsigned long long)lo) >> (8u * (s & 3u)));
#else
  return __builtin_amdgcn_alignbyte(hi, lo, s);
#endif
}

__device__ __forceinline__ unsigned char resample_clip8(int acc) {
  const int v = acc >> kResampleBits;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// One workgroup per (frame, source row).  src [F][H][W][C] u8 -> dst [F][H][OW][C] u8.
inline __global__ void __launch_bounds__(256) resample_h_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                            const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                            const int* __restrict__ kk, int ksize, int H, int W, int C, int OW) {
  __shared__ unsigned char row_s[kResampleMaxRowBytes];
  const size_t row = blockIdx.x;                       // f * H + y
  const unsigned char* s = src + row * (size_t)W * C;
  for (int i = threadIdx.x; i < W * C; i += 256) row_s[i] = s[i];
  __syncthreads();
  unsigned char* d = dst + row * (size_t)OW * C;
  for (int o = threadIdx.x; o < OW * C; o += 256) {
    const int xx = o / C, c = o - xx * C;
    const int x0 = xmin[xx], n = xcnt[xx];
    const int* k = kk + (size_t)xx * ksize;
    int acc = 1 << (kResampleBits - 1);
    for (int x = 0; x < n; ++x) acc += resample_mul24((int)row_s[(x0 + x) * C + c], k[x]);
    d[o] = resample_clip8(acc);
  }
}

// One workgroup per (frame, output row).  src [F][H][OW][C] u8 -> dst_f32 [F][C][OH][OW] = lut[u8]  and/or  dst_u8 [F][OH][OW][C].
inline __global__ void __launch_bounds__(256) resample_v_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst_f32,
                                                            unsigned char* __restrict__ dst_u8, const int* __restrict__ ymin,
                                                            const int* __restrict__ ycnt, const int* __restrict__ kk, int ksize,
                                                            const float* __restrict__ lut, int H, int OW, int C, int OH) {
  const int f = blockIdx.x / OH, yy = blockIdx.x - f * OH;
  const int y0 = ymin[yy], n = ycnt[yy];
  const int* k = kk + (size_t)yy * ksize;
  const unsigned char* s = src + ((size_t)f * H + y0) * OW * C;
  for (int o = threadIdx.x; o < OW * C; o += 256) {
    int acc = 1 << (kResampleBits - 1);
    for (int y = 0; y < n; ++y) acc += resample_mul24((int)s[(size_t)y * OW * C + o], k[y]);
    const unsigned char v = resample_clip8(acc);
    const int xx = o / C, c = o - xx * C;
    if (dst_u8) dst_u8[((size_t)f * OH + yy) * OW * C + o] = v;
    if (dst_f32) dst_f32[(((size_t)f * C + c) * OH + yy) * OW + xx] = lut[v];
  }
}

// RGB fast path of the horizontal pass: a workgroup stages RPB source rows with 16-byte loads; a thread owns ONE output
// column (all three channels) so its <= KMAX coefficients are fetched once into registers and reused for 3 x RPB outputs,
// and every LDS read is `per-thread base + immediate`.  Same arithmetic, same bits as the generic kernel.
template <int KMAX, int RPB>
__global__ void __launch_bounds__(256) resample_h_rgb_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                             const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                             const int* __restrict__ kk, int ksize, long rows, int W, int OW) {
  constexpr int C = 3;
  constexpr int kRowStride = kResampleMaxRowBytes / RPB;          // bytes reserved per staged row (incl. tail slack)
  __shared__ __attribute__((aligned(16))) unsigned char row_s[kResampleMaxRowBytes];
  const long r0 = (long)blockIdx.x * RPB;
  const int rowb = W * C;                                          // host guarantees rowb % 16 == 0, rowb + KMAX*C <= kRowStride
  const int vec_per_row = rowb / 16;
  for (int i = threadIdx.x; i < RPB * vec_per_row; i += 256) {
    const int r = i / vec_per_row, v = i - r * vec_per_row;
    if (r0 + r < rows)
      *reinterpret_cast<uint4*>(row_s + r * kRowStride + 16 * v) = *reinterpret_cast<const uint4*>(src + (r0 + r) * (size_t)rowb + 16 * v);
  }
  __syncthreads();
  for (int xx = threadIdx.x; xx < OW; xx += 256) {
    const int x0 = xmin[xx], n = xcnt[xx];
    int k[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) k[j] = j < n ? kk[(size_t)xx * ksize + j] : 0;
#pragma unroll
    for (int r = 0; r < RPB; ++r) {
      if (r0 + r >= rows) break;
      const unsigned char* p = row_s + r * kRowStride + x0 * C;
      int a0 = 1 << (kResampleBits - 1), a1 = a0, a2 = a0;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {                             // taps beyond n carry a zero coefficient (reads stay inside the slack)
        a0 += resample_mul24((int)p[j * C], k[j]);
        a1 += resample_mul24((int)p[j * C + 1], k[j]);
        a2 += resample_mul24((int)p[j * C + 2], k[j]);
      }
      unsigned char* d = dst + ((r0 + r) * (size_t)OW + xx) * C;
      d[0] = resample_clip8(a0); d[1] = resample_clip8(a1); d[2] = resample_clip8(a2);
    }
  }
}

// Round 6: the same pass as a PERSISTENT workgroup (grid = a few per CU, each walks many groups of RPB source rows): a thread keeps the
// coefficients of its NXC output columns in registers for the kernel's lifetime (the kernel above re-fetches 20 table entries per column and row
// group), and the next group's rows are fetched into registers BEFORE the current group is computed and stored into the other half of a
// double-buffered LDS stage behind it.  Measured on 256 frames of 1080p -> 512 columns: 1222 us (kernel above) -> 1059 us; what was tried on top
// and did not move it: 24-bit multiplies with or without SDWA byte selects (1076 / 1120), dword stores packed from four lanes through a shuffle
// (1720), two instead of four rows per group = six instead of three workgroups per CU (1059 vs 1076) -- the pass moves its 2.0 GB at 1.9 TB/s like
// the vertical pass (2.1) and the median (2.1), neither its arithmetic (~0.45 ms), its LDS reads nor its occupancy set that.
// OW <= 256 * NXC.  Same arithmetic, same bits.
template <int KMAX, int RPB, int NXC>
__global__ void __launch_bounds__(256) resample_h_rgb_persist_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                                     const int* __restrict__ xmin, const int* __restrict__ xcnt,
                                                                     const int* __restrict__ kk, int ksize, long rows, int W, int OW) {
  constexpr int C = 3;
  constexpr int kRowStride = kResamplePersistRowBytes;             // bytes reserved per staged row (incl. tail slack): 1080p RGB rows are 5760
  constexpr int kStage = kRowStride * RPB;                         // bytes per LDS stage (two stages)
  constexpr int NV = (kStage / 16 + 255) / 256;                    // 16-byte vectors per thread and stage (upper bound)
  __shared__ __attribute__((aligned(16))) unsigned char row_s[2 * kStage];
  const int rowb = W * C;                                          // host guarantees rowb % 16 == 0, rowb + KMAX*C <= kRowStride
  const int vec_per_row = rowb / 16, nvec = RPB * vec_per_row;
  const long groups = (rows + RPB - 1) / RPB;
  int x0[NXC], k[NXC][KMAX];
#pragma unroll
  for (int c = 0; c < NXC; ++c) {
    const int xx = threadIdx.x + 256 * c;
    const bool on = xx < OW;
    x0[c] = on ? xmin[xx] : 0;
    const int n = on ? xcnt[xx] : 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) k[c][j] = j < n ? kk[(size_t)xx * ksize + j] : 0;
  }
  uint4 nxt[NV];
  auto fetch = [&](long g) {                                      // rows of group g -> registers (rows beyond the last: zeros, never stored)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int r = e / vec_per_row, v = e - r * vec_per_row;
      nxt[i] = (e < nvec && g * RPB + r < rows) ? *reinterpret_cast<const uint4*>(src + (g * RPB + r) * (size_t)rowb + 16 * v) : uint4{0u, 0u, 0u, 0u};
    }
  };
  auto stash = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int r = e / vec_per_row, v = e - r * vec_per_row;
      if (e < nvec) *reinterpret_cast<uint4*>(row_s + stage * kStage + r * kRowStride + 16 * v) = nxt[i];
    }
  };
  long g = blockIdx.x;
  if (g >= groups) return;
  fetch(g);
  stash(0);
  __syncthreads();
  int cur = 0;
  for (; g < groups; g += gridDim.x) {
    const long gn = g + gridDim.x;
    if (gn < groups) fetch(gn);
    const long r0 = g * RPB;
#pragma unroll
    for (int c = 0; c < NXC; ++c) {
      const int xx = threadIdx.x + 256 * c;
#pragma unroll
      for (int r = 0; r < RPB; ++r) {
        const bool on = xx < OW && r0 + r < rows;
        unsigned v = 0u;
        if (on) {
          // The window's 3 KMAX bytes start at byte 3 x0 of the row: ANY alignment.  Read through a byte pointer the compiler merges them into
          // ds_read_b128 / _b96 at that address -- and a DS access wider than 32 bits off its natural alignment is replayed at 64 cycles per
          // wave-instruction (hardware guide, guideline 17): that, not arithmetic or HBM, was this pass's time.  So: 32-bit reads of the aligned
          // dwords around the window (volatile: not to be merged again), realigned in registers with v_alignbyte_b32.
          const unsigned b0 = (unsigned)(x0[c] * C);
          const resample_lds_cvu_p pw = (resample_lds_cvu_p)(row_s + cur * kStage + r * kRowStride + (b0 & ~3u));
          constexpr int NW = (KMAX * C + 3) / 4;                   // dwords of the window
          unsigned dd[NW + 1], ww[NW];
#pragma unroll
          for (int i = 0; i <= NW; ++i) dd[i] = pw[i];
#pragma unroll
          for (int i = 0; i < NW; ++i) ww[i] = resample_alignbyte(dd[i + 1], dd[i], b0 & 3u);
          int a0 = 1 << (kResampleBits - 1), a1 = a0, a2 = a0;
#pragma unroll
          for (int j = 0; j < KMAX; ++j) {                         // taps beyond the column's count carry a zero coefficient (reads stay inside the slack)
            const int t0 = j * C, t1 = j * C + 1, t2 = j * C + 2;
            a0 += resample_mul24((int)((ww[t0 >> 2] >> (8 * (t0 & 3))) & 255u), k[c][j]);
            a1 += resample_mul24((int)((ww[t1 >> 2] >> (8 * (t1 & 3))) & 255u), k[c][j]);
            a2 += resample_mul24((int)((ww[t2 >> 2] >> (8 * (t2 & 3))) & 255u), k[c][j]);
          }
          v = (unsigned)resample_clip8(a0) | ((unsigned)resample_clip8(a1) << 8) | ((unsigned)resample_clip8(a2) << 16);
        }
        if (on) {                                                  // (byte stores: packing four lanes' bytes into dword stores through a lane shuffle measured SLOWER, 1076 -> 1720 us)
          unsigned char* d = dst + ((r0 + r) * (size_t)OW + xx) * C;
          d[0] = (unsigned char)(v & 255u); d[1] = (unsigned char)((v >> 8) & 255u); d[2] = (unsigned char)(v >> 16);
        }
      }
    }
    if (gn < groups) stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
}

// Vertical pass, four consecutive output bytes per thread (32-bit loads of the intermediate rows); (OW*C) % 4 == 0.
// (Round 6 also tried this loop with the loads of all support rows issued ahead of the first multiply: 437 us against 442 -- not its limiter either.)
inline __global__ void __launch_bounds__(256) resample_v_u8x4_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst_f32,
                                                              unsigned char* __restrict__ dst_u8, const int* __restrict__ ymin,
                                                              const int* __restrict__ ycnt, const int* __restrict__ kk, int ksize,
                                                              const float* __restrict__ lut, int H, int OW, int C, int OH) {
  const int f = blockIdx.x / OH, yy = blockIdx.x - f * OH;
  const int y0 = ymin[yy], n = ycnt[yy];
  const int* k = kk + (size_t)yy * ksize;
  const int rowb = OW * C;
  const unsigned char* s = src + ((size_t)f * H + y0) * rowb;
  for (int q = threadIdx.x; q < rowb / 4; q += 256) {
    int a0 = 1 << (kResampleBits - 1), a1 = a0, a2 = a0, a3 = a0;
    for (int y = 0; y < n; ++y) {
      const unsigned v = *reinterpret_cast<const unsigned int*>(s + (size_t)y * rowb + 4 * q);
      const int ky = k[y];
      a0 += resample_mul24((int)(v & 255u), ky); a1 += resample_mul24((int)((v >> 8) & 255u), ky);
      a2 += resample_mul24((int)((v >> 16) & 255u), ky); a3 += resample_mul24((int)(v >> 24), ky);
    }
    const unsigned char o[4] = {resample_clip8(a0), resample_clip8(a1), resample_clip8(a2), resample_clip8(a3)};
    if (dst_u8)
      *reinterpret_cast<unsigned int*>(dst_u8 + ((size_t)f * OH + yy) * rowb + 4 * q) =
          (unsigned)o[0] | ((unsigned)o[1] << 8) | ((unsigned)o[2] << 16) | ((unsigned)o[3] << 24);
    if (dst_f32) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int oidx = 4 * q + e, xx = oidx / C, c = oidx - xx * C;
        dst_f32[(((size_t)f * C + c) * OH + yy) * OW + xx] = lut[o[e]];
      }
    }
  }
}

// (Round 6 built two more forms of the vertical pass and measured both on 256 frames of 1080p -> 288 x 512: the support rows staged in LDS with 16-byte
//  loads and one 16-byte store per thread into a colour plane instead of four scattered 4-byte stores: 439 us against this kernel's 442; the same with
//  eight adjacent output rows per workgroup from one staged window, so that a source row is fetched once instead of ~4 times: 667 us.  Neither the
//  stores' shape nor the re-reads are what its time is; the kernel above stays.)

// Median over T frames of P bytes each: frames [T][P] u8 -> med [P] u8 = floor((v[(T-1)/2] + v[T/2]) / 2).
// 128 threads x 256 32-bit bins = 128 KB of LDS; bin-major layout keeps the 32 lanes of a half-wave on 32 banks.
// med2 (optional): a + b as uint16 = twice the float median (np.median of an even count ends in .5): what the
// difference-frame modes subtract (dataset.py:439) without leaving integer arithmetic.
inline __global__ void __launch_bounds__(128) median_u8_kernel(const unsigned char* __restrict__ frames, unsigned char* __restrict__ med,
                                                        unsigned short* __restrict__ med2, int T, long P) {
  __shared__ unsigned int hist_s[256 * 128];          // [bin][thread]: 128 KB of the CU's 160 KB
  const int tid = threadIdx.x;
  const long p = (long)blockIdx.x * 128 + tid;
  for (int b = 0; b < 256; ++b) hist_s[b * 128 + tid] = 0u;
  if (p < P) {
    const unsigned char* f = frames + p;
    int t = 0;
    for (; t + 8 <= T; t += 8) {
      unsigned char v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = f[(size_t)(t + u) * P];
#pragma unroll
      for (int u = 0; u < 8; ++u) hist_s[(int)v[u] * 128 + tid] += 1u;
    }
    for (; t < T; ++t) hist_s[(int)f[(size_t)t * P] * 128 + tid] += 1u;
    const unsigned r0 = (unsigned)((T - 1) / 2), r1 = (unsigned)(T / 2);
    unsigned cum = 0;
    int a = -1, b2 = -1;
    for (int b = 0; b < 256; ++b) {
      cum += hist_s[b * 128 + tid];
      if (a < 0 && cum > r0) a = b;
      if (b2 < 0 && cum > r1) { b2 = b; break; }
    }
    if (med) med[p] = (unsigned char)((a + b2) >> 1);
    if (med2) med2[p] = (unsigned short)(a + b2);
  }
}

// The same median as a two-pass 4-bit radix select, registers only (no LDS -> full occupancy, 4-byte loads):
//   pass A histograms the HIGH nibble of every frame's byte into 16 bins and finds the bins holding the two middle
//   ranks; pass B re-reads the stack and histograms the LOW nibble of the bytes that fall into those bins.
// Each thread owns 4 consecutive byte positions.  Counting is SWAR: one 64-bit word of sixteen 4-bit counters takes
// `1 << 4n` per byte and is spilled into 16-bit counters every 15 frames, i.e. ~6 integer instructions per byte
// instead of an LDS read-modify-write.  HBM traffic 2 x T x P bytes (the stack does not fit the L2).  T <= 65535, P % 4 == 0.
struct NibbleHist {
  unsigned long long swar;       // 16 x 4-bit counters (at most 15 increments between spills)
  unsigned int c[8];             // 16 x 16-bit counters: bins 2j (low half) and 2j+1 (high half)
  __device__ __forceinline__ void clear() { swar = 0ull; for (int j = 0; j < 8; ++j) c[j] = 0u; }
  __device__ __forceinline__ void add(unsigned n) { swar += 1ull << (4u * n); }
  __device__ __forceinline__ void add_if(unsigned n, bool take) { swar += (take ? 1ull : 0ull) << (4u * n); }
  __device__ __forceinline__ void spill() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned b = (unsigned)(swar >> (8 * j)) & 255u;
      c[j] += (b & 15u) | ((b >> 4) << 16);
    }
    swar = 0ull;
  }
  // smallest bin whose cumulative count exceeds `rank`; `before` = count of all smaller bins
  __device__ __forceinline__ void select(unsigned rank, unsigned& bin, unsigned& before) const {
    unsigned cum = 0, found = 16u, bef = 0u;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const unsigned n = (c[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu;
      const bool hit = found == 16u && cum + n > rank;
      bef = hit ? cum : bef;
      found = hit ? (unsigned)b : found;
      cum += n;
    }
    bin = found; before = bef;
  }
};

inline __global__ void __launch_bounds__(256) median_u8_radix_kernel(const unsigned int* __restrict__ frames4, unsigned char* __restrict__ med,
                                                              unsigned short* __restrict__ med2, int T, long P4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P4) return;
  const unsigned r0 = (unsigned)((T - 1) / 2), r1 = (unsigned)(T / 2);
  const unsigned int* f = frames4 + i;
  NibbleHist h[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) h[b].clear();
  // ---- pass A: high nibbles
  for (int t0 = 0; t0 < T; t0 += 15) {
    const int tn = T - t0 < 15 ? T - t0 : 15;
    if (tn == 15) {
      unsigned v[15];
#pragma unroll
      for (int u = 0; u < 15; ++u) v[u] = f[(size_t)(t0 + u) * P4];
#pragma unroll
      for (int u = 0; u < 15; ++u)
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b].add((v[u] >> (8 * b + 4)) & 15u);
    } else {
      for (int u = 0; u < tn; ++u) {
        const unsigned v = f[(size_t)(t0 + u) * P4];
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b].add((v >> (8 * b + 4)) & 15u);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) h[b].spill();
  }
  unsigned hiA[4], hiB[4], befA[4], befB[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { h[b].select(r0, hiA[b], befA[b]); h[b].select(r1, hiB[b], befB[b]); }
  // ---- pass B: low nibbles of the bytes whose high nibble is hiA (rank r0) / hiB (rank r1)
  NibbleHist la[4], lb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { la[b].clear(); lb[b].clear(); }
  auto take = [&](unsigned v) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const unsigned byte = (v >> (8 * b)) & 255u;
      la[b].add_if(byte & 15u, (byte >> 4) == hiA[b]);
      lb[b].add_if(byte & 15u, (byte >> 4) == hiB[b]);
    }
  };
  for (int t0 = 0; t0 < T; t0 += 15) {
    const int tn = T - t0 < 15 ? T - t0 : 15;
    if (tn == 15) {                                      // (round 6) the 15 loads first, like pass A: with one load per ~60 instructions the pass waited a
      unsigned v[15];                                    // memory latency per frame -- it was latency-bound, not arithmetic-bound
#pragma unroll
      for (int u = 0; u < 15; ++u) v[u] = f[(size_t)(t0 + u) * P4];
#pragma unroll
      for (int u = 0; u < 15; ++u) take(v[u]);
    } else {
      for (int u = 0; u < tn; ++u) take(f[(size_t)(t0 + u) * P4]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) { la[b].spill(); lb[b].spill(); }
  }
  unsigned outm = 0u;
  unsigned short o2[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    unsigned lo, bef;
    la[b].select(r0 - befA[b], lo, bef);
    const unsigned va = hiA[b] * 16u + lo;
    lb[b].select(r1 - befB[b], lo, bef);
    const unsigned vb = hiB[b] * 16u + lo;
    outm |= ((va + vb) >> 1) << (8 * b);
    o2[b] = (unsigned short)(va + vb);
  }
  if (med) reinterpret_cast<unsigned int*>(med)[i] = outm;
  if (med2) {
#pragma unroll
    for (int b = 0; b < 4; ++b) med2[4 * i + b] = o2[b];
  }
}

// Difference frame of bg_mode 'subtract' / 'subtract_concat' (dataset.py:439, 443):
//   out[f][p] = uint8( sum_c |frame[f][p][c] - median[p][c]| )   with a float64 median -> truncate, then wrap mod 256.
// In integers: s2 = sum_c |2*v - med2|,  out = (s2 >> 1) & 255.   frames [F][P][3] u8, med2 [P][3] u16 -> out [F][P] u8.
inline __global__ void __launch_bounds__(256) absdiff_sum_u8_kernel(const unsigned char* __restrict__ frames,
                                                             const unsigned short* __restrict__ med2,
                                                             unsigned char* __restrict__ out, int F, long P) {
  const long total = (long)F * P;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long p = t % P;
    const unsigned char* v = frames + t * 3;
    const unsigned short* m = med2 + p * 3;
    int s2 = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const int d = 2 * (int)v[c] - (int)m[c]; s2 += d < 0 ? -d : d; }
    out[t] = (unsigned char)((s2 >> 1) & 255);
  }
}

}  // namespace tnv3
