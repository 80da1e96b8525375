// postproc.h -- heat-map -> coordinates on the device, so that only (x, y, w, h) integers leave HBM.
//
//  * ensemble_frames_kernel: the temporal ensemble of predict.py:163-209 (heat maps) / 243-301 (coordinates) in
//    closed form.  Frame t is covered by windows s in [t-L+1, t] at in-window position t-s; with the reference's
//    zero-prefilled buffer this is
//        t <  num_sample, t <  L-1 : sum_s win[s][t-s] / (t+1)                  ("incomplete buffer", all modes)
//        t <  num_sample, t >= L-1 : sum_k weight[k] * win[t-L+1+k][L-1-k]      (get_ensemble_weight, test.py:25-50)
//        t >= num_sample           : sum_s win[s][t-s] / (L - (t - num_sample + 1))   (tail after the last window)
//    Windows outside [0, num_sample) contribute zero.  HBM-bound gather: L reads + 1 write per output element.
//    The arithmetic is the reference's, operation by operation, so that results are BIT-identical to its torch-CPU loops:
//    `(buf[...] * weight[:, None]).sum(0)` rounds every product to fp32 first (no FMA), and torch's CPU sum kernel
//    (SumKernel.cpp) walks the L rows either sequentially (sum_order 0: its vectorised outer sum -- the heat maps, E = H*W)
//    or as four interleaved partial sums p_j = sum_i x[4i+j], leftovers into p_0, result ((p_0+p_1)+p_2)+p_3 (sum_order 1:
//    its scalar `row_sum` with ilp_factor 4 -- taken when fewer than four columns remain, i.e. the (L, 2) coordinates).
//    Both orders were identified against the goldens produced by the reference's own loops (tests/golden/ensemble.npz:
//    24 heat-map and 3 coordinate cases, all bit-equal).  Warm-up and tail divide the plain sum (true division).
//
//  * peak-find = predict.py:35 (`> 0.5`) + predict_location (test.py:52-79): cv2.findContours(RETR_EXTERNAL) +
//    cv2.boundingRect + largest box.  Integer work, restated as 8-connected component labelling by lock-free
//    union-find over horizontal RUNS (found with wave ballots, so a dense map costs one node per run, not per pixel;
//    label = smallest linear pixel index of the component = its first pixel in raster order), per-root bounding boxes
//    by atomic min/max of the run boxes, and ONE 64-bit atomicMax per root on (area << 32 | order) where
//    `order` encodes the tie rule (equal areas: the component discovered last in raster order wins -- OpenCV's
//    contour list order combined with the strict '>' at test.py:74; see oracle/postproc.py).  Bit-exact integers.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tnv3 {

typedef float pp_f32x4 __attribute__((ext_vector_type(4)));

inline __global__ void __launch_bounds__(256) ensemble_frames_kernel(const float* __restrict__ win, int n_local, long s_base,
                                                              int L, int E, const float* __restrict__ weight, long t0,
                                                              int n_frames, long num_sample, int sum_order, float* __restrict__ out) {
  // win: [n_local][L][E] (window s_base + i at row i), E = elements per position (H*W, or 2 for coordinates)
  const long total = (long)n_frames * E;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int fi = (int)(idx / E);
    const int e = (int)(idx - (long)fi * E);
    const long t = t0 + fi;
    const bool general = (t < num_sample) && (t >= L - 1);
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // sum_order 0 uses part[0] only
    const int k_ilp = sum_order ? (L / 4) * 4 : 0;
    for (int k = 0; k < L; ++k) {
      const long s = t - (L - 1) + k;            // window index; in-window position L-1-k
      if (s < 0 || s >= num_sample) continue;    // the reference's zero rows: adding +0 is exact
      const long i = s - s_base;
      if (i < 0 || i >= n_local) continue;       // caller guarantees residency of all needed windows
      float v = win[((size_t)i * L + (L - 1 - k)) * E + e];
      if (general) v = __fmul_rn(v, weight[k]);  // rounded product, then a rounded add: never contracted into an FMA
      const int j = k < k_ilp ? (k & 3) : 0;
      part[j] = __fadd_rn(part[j], v);
    }
    float acc = part[0];
    if (sum_order) acc = __fadd_rn(__fadd_rn(__fadd_rn(part[0], part[1]), part[2]), part[3]);
    if (!general) {
      const float div = (t < num_sample) ? (float)(t + 1) : (float)(L - (t - num_sample + 1));
      acc = acc / div;
    }
    out[idx] = acc;
  }
}

// ---- connected components --------------------------------------------------------------------------------------
// workspace per frame: int label[HW]; int box[4][HW] (min x, min y, max x, max y at root pixels); unsigned long long best
struct PeakWs {
  int* label;
  int* box;                      // 4 planes of HW
  unsigned long long* best;      // one per frame
};

__device__ __forceinline__ int ccl_load(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }

__device__ __forceinline__ int ccl_find(const int* lab, int x) {
  int p = ccl_load(lab + x);
  while (p != x) { x = p; p = ccl_load(lab + x); }
  return x;
}

__device__ __forceinline__ void ccl_union(int* lab, int a, int b) {
  for (;;) {
    a = ccl_find(lab, a);
    b = ccl_find(lab, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }          // a > b: hang root a under b
    const int old = atomicMin(lab + a, b);
    if (old == a) return;                                   // a was still a root: linked
    a = old;                                                // somebody re-parented a meanwhile: retry from there
  }
}

// Pass 1 -- threshold + horizontal runs, no atomics: a wave owns 64 consecutive pixels; from the ballot of the
// foreground bits every pixel gets the index of the first pixel of its run (within the 64-pixel segment and the image
// row) as its label, and the run's first pixel records the run's box.  Dense maps collapse to one node per run.
inline __global__ void __launch_bounds__(256) ccl_init_kernel(const float* __restrict__ heat, float thr, int* __restrict__ label,
                                                       int* __restrict__ box, unsigned long long* __restrict__ best,
                                                       int H, int W) {
  const int HW = H * W;
  const int f = blockIdx.y;
  const float* hm = heat + (size_t)f * HW;
  int* lab = label + (size_t)f * HW;
  int* bx = box + (size_t)f * 4 * HW;
  if (blockIdx.x == 0 && threadIdx.x == 0) best[f] = 0ull;
  const int lane = threadIdx.x & 63;
  for (int p0 = blockIdx.x * 256; p0 < HW; p0 += gridDim.x * 256) {       // block-uniform trip count: ballots are full
    const int p = p0 + threadIdx.x;
    const bool valid = p < HW;
    const int y = valid ? p / W : 0, x = p - y * W;
    const bool fg = valid && hm[valid ? p : 0] > thr;
    const unsigned long long M = __ballot(fg);
    const unsigned long long R = __ballot(valid && x == 0);                 // lanes that begin an image row
    if (fg) {
      const unsigned long long S = M & (~(M << 1) | R);                     // run starts
      const unsigned long long E = M & (~(M >> 1) | (R >> 1));              // run ends
      const int s = 63 - __builtin_clzll(S & (~0ull >> (63 - lane)));
      lab[p] = p - (lane - s);
      if (s == lane) {
        const int e = __builtin_ctzll(E & (~0ull << lane));
        bx[p] = x; bx[HW + p] = y; bx[2 * HW + p] = x + (e - lane); bx[3 * HW + p] = y;
      }
    } else if (valid) {
      lab[p] = -1;
    }
  }
}

__device__ __forceinline__ bool ccl_fg(const int* lab, int q) { return ccl_load(lab + q) >= 0; }

// Pass 2 -- link runs (8-connectivity).  Only one pixel per pair of touching runs issues a union:
//   * a run cut by a 64-pixel segment boundary is re-joined by its first pixel;
//   * against the row above, the pixel under N links with N unless its left neighbour already sits under the same
//     upper run (W and NW set); with N clear, NW is linked only by a run's first pixel and NE only when the right
//     neighbour (which has NE as its N) is background.
inline __global__ void __launch_bounds__(256) ccl_merge_kernel(int* __restrict__ label, int H, int W) {
  const int HW = H * W;
  int* lab = label + (size_t)blockIdx.y * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    if (!ccl_fg(lab, p)) continue;
    const int y = p / W, x = p - y * W;
    const bool Wf = x > 0 && ccl_fg(lab, p - 1);
    if ((p & 63) == 0 && Wf) ccl_union(lab, p, p - 1);
    if (y > 0) {
      const bool N = ccl_fg(lab, p - W);
      const bool NW = x > 0 && ccl_fg(lab, p - W - 1);
      if (N) {
        if (!(Wf && NW)) ccl_union(lab, p, p - W);
      } else {
        if (NW && !Wf) ccl_union(lab, p, p - W - 1);
        if (x + 1 < W && ccl_fg(lab, p - W + 1) && !ccl_fg(lab, p + 1)) ccl_union(lab, p, p - W + 1);
      }
    }
  }
}

// Pass 3 -- one node per run: resolve the root, fold the run's box into the root's box (atomics only when they would
// change the value; min-y needs none: the root is the component's first pixel in raster order), flatten the label.
inline __global__ void __launch_bounds__(256) ccl_box_kernel(int* __restrict__ label, int* __restrict__ box, int H, int W) {
  const int HW = H * W;
  int* lab = label + (size_t)blockIdx.y * HW;
  int* bx = box + (size_t)blockIdx.y * 4 * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    if (!ccl_fg(lab, p)) continue;
    const int y = p / W, x = p - y * W;
    if (!(x == 0 || (p & 63) == 0 || !ccl_fg(lab, p - 1))) continue;      // not the first pixel of a run
    const int r = ccl_find(lab, p);
    if (r == p) continue;                                                  // the root carries its own run already
    const int x1 = bx[2 * HW + p];
    if (x < ccl_load(bx + r)) atomicMin(bx + r, x);
    if (x1 > ccl_load(bx + 2 * HW + r)) atomicMax(bx + 2 * HW + r, x1);
    if (y > ccl_load(bx + 3 * HW + r)) atomicMax(bx + 3 * HW + r, y);
    __atomic_store_n(lab + p, r, __ATOMIC_RELAXED);                        // flatten: r is a root, paths only get shorter
  }
}

inline __global__ void __launch_bounds__(256) ccl_select_kernel(const int* __restrict__ label, const int* __restrict__ box,
                                                         unsigned long long* __restrict__ best, int H, int W,
                                                         int tie_last_wins) {
  const int HW = H * W;
  const int* lab = label + (size_t)blockIdx.y * HW;
  const int* bx = box + (size_t)blockIdx.y * 4 * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    if (lab[p] != p) continue;                              // roots only
    const unsigned area = (unsigned)(bx[2 * HW + p] - bx[p] + 1) * (unsigned)(bx[3 * HW + p] - bx[HW + p] + 1);
    const unsigned order = tie_last_wins ? (unsigned)p + 1u : 0xFFFFFFFFu - (unsigned)p;
    atomicMax(best + blockIdx.y, ((unsigned long long)area << 32) | order);
  }
}

// out[f] = max of heat[f] over the box (x, y, w, h) = boxes[f] (the whole map when boxes == nullptr); 0 for an empty box.
// test.py:164-167: the detection confidence `np.amax(y_p[y:y+h, x:x+w])`; with boxes == nullptr: the `np.amax(y_t) > 0`
// "ground truth has a ball" test of test.py:170-178.  One workgroup per map, fixed-shape tree reduction (max is exact).
inline __global__ void __launch_bounds__(256) heatmap_box_max_kernel(const float* __restrict__ heat, const int* __restrict__ boxes,
                                                              float* __restrict__ out, int H, int W) {
  __shared__ float red[256];
  const int f = blockIdx.x;
  int x0 = 0, y0 = 0, bw = W, bh = H;
  if (boxes) { x0 = boxes[4 * f]; y0 = boxes[4 * f + 1]; bw = boxes[4 * f + 2]; bh = boxes[4 * f + 3]; }
  if (x0 < 0) { bw += x0; x0 = 0; }
  if (y0 < 0) { bh += y0; y0 = 0; }
  if (x0 + bw > W) bw = W - x0;
  if (y0 + bh > H) bh = H - y0;
  const float* hm = heat + (size_t)f * H * W;
  const int n = (bw > 0 && bh > 0) ? bw * bh : 0;
  float m = -__builtin_inff();
  bool nan = false;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = hm[(size_t)(y0 + i / bw) * W + x0 + i % bw];
    nan |= v != v;
    m = v > m ? v : m;
  }
  red[threadIdx.x] = nan ? __builtin_nanf("") : m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float a = red[threadIdx.x], b = red[threadIdx.x + s];
      red[threadIdx.x] = (a != a || b != b) ? __builtin_nanf("") : (b > a ? b : a);     // NaN propagates like np.amax
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[f] = n > 0 ? red[0] : 0.0f;
}

// out[f] = (x, y, w, h) of the winning box, or (0,0,0,0) when the thresholded map is empty (test.py:60-62)
inline __global__ void ccl_emit_kernel(const int* __restrict__ box, const unsigned long long* __restrict__ best,
                                int* __restrict__ out, int frames, int H, int W, int tie_last_wins) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames) return;
  const int HW = H * W;
  const unsigned long long k = best[f];
  int x = 0, y = 0, w = 0, h = 0;
  if (k != 0ull) {
    const unsigned order = (unsigned)(k & 0xFFFFFFFFull);
    const int p = tie_last_wins ? (int)(order - 1u) : (int)(0xFFFFFFFFu - order);
    const int* bx = box + (size_t)f * 4 * HW;
    x = bx[p]; y = bx[HW + p];
    w = bx[2 * HW + p] - x + 1; h = bx[3 * HW + p] - y + 1;
  }
  out[4 * f] = x; out[4 * f + 1] = y; out[4 * f + 2] = w; out[4 * f + 3] = h;
}

}  // namespace tnv3
