// hip_emu.h -- a tiny host-side SIMT emulator for gfx950 HIP kernels.  TEST TOOL ONLY.
//
// The build container has no GPU.  This header lets the UNCHANGED kernel headers under
// tracknetv3_amd/csrc/kernels/ be compiled by the host clang (-include hip_emu.h) and executed
// lane-by-lane, so that tile/lane/LDS index arithmetic and the MFMA fragment mapping are checked
// against the oracle by the CPU test-suite before any GPU minute is spent.  It is never part of
// the product: the shipped library is built by hipcc from the same kernel headers.
//
// Model: one workgroup at a time; each work-item is a ucontext fiber; fibers yield at
// __syncthreads() and at wave-level exchanges (shuffles, MFMA).  The MFMA emulation implements the
// documented gfx950 fragment layouts (guide: cdna_hip_programming.md section 3):
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D reg r -> row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r -> row=4*(l>>4)+r,            col=l&15
// and computes each output as the k-ordered fmaf chain the hardware produces.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define TNV3_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct emu_dim3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
static emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

// Minimal x86-64 SysV context switch (callee-saved registers + stack pointer).  glibc's swapcontext makes a
// sigprocmask system call per switch, which dominated the emulator's run time.
extern "C" void tnv3_emu_swap(void** save_sp, void* load_sp);
__asm__(R"ASM(
.text
.globl tnv3_emu_swap
.type tnv3_emu_swap,@function
tnv3_emu_swap:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size tnv3_emu_swap,.-tnv3_emu_swap
)ASM");

namespace emu {

enum { WAVE = 64, MAX_WAVES = 16 };

// A work-item's vector-memory operations still "in flight" (lazy-DMA mode, below): LDS-DMA pieces carry their 16 bytes and LDS
// destination, every other buffer load / store is a placeholder that only occupies its place in the in-order queue vmcnt counts.
struct PendingVm {
  void* dst = nullptr;
  unsigned size = 0;
  unsigned char data[16];
};
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  bool wait_block = false;
  unsigned wait_gen = 0;
  emu_dim3 tid{};
  std::vector<PendingVm> vm;      // oldest first
};

struct State {
  void* main_sp = nullptr;
  std::vector<Fiber> fibers;
  Fiber* cur = nullptr;
  unsigned nthreads = 0;
  unsigned block_gen = 0, block_count = 0;
  unsigned wave_gen[MAX_WAVES], wave_count[MAX_WAVES], wave_size[MAX_WAVES];
  uint64_t xchg[MAX_WAVES][4][WAVE];   // per-wave exchange scratch
  std::function<void()> body;
  bool lazy = false;            // lazy-DMA mode of this launch (below)
};
inline State& S() { static State s; return s; }

inline unsigned flat_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }
inline unsigned lane_id() { return flat_tid() % WAVE; }
inline unsigned wave_id() { return flat_tid() / WAVE; }

// Lazy-DMA mode (TNV3_EMU_LAZY_DMA=1): an LDS-DMA piece lands as LATE as the program allows -- when its work-item executes an
// s_waitcnt whose vmcnt leaves fewer operations in flight than are queued behind it, or at the end of the kernel -- instead of at issue
// (the default: as EARLY as possible).  The two extremes bracket the hardware; a kernel whose counted waits are too weak, or that
// relies on a barrier alone to publish a DMA, reads stale LDS in this mode and fails its parity test on the CPU.  __syncthreads() does
// NOT complete them: hipcc emits `s_waitcnt lgkmcnt(0); s_barrier` for it on gfx950, no vmcnt.
// Only meaningful for kernels whose vector-memory operations all go through the buffer helpers below (the Winograd families):
// a plain pointer load cannot be intercepted, so it is missing from the queue that a counted wait counts.  Read per launch.
inline bool lazy_dma() { return S().lazy; }
inline void vm_retire(Fiber& f, size_t leave) {
  if (f.vm.size() <= leave) return;
  const size_t n = f.vm.size() - leave;
  for (size_t i = 0; i < n; ++i)
    if (f.vm[i].dst) memcpy(f.vm[i].dst, f.vm[i].data, f.vm[i].size);
  f.vm.erase(f.vm.begin(), f.vm.begin() + n);
}
inline void vm_placeholder() {
  if (lazy_dma()) S().cur->vm.emplace_back();
}
inline void vm_dma(void* dst, const void* src, unsigned size) {      // src == nullptr: zeros (out of the descriptor's range)
  if (!lazy_dma()) {
    if (src) memcpy(dst, src, size); else memset(dst, 0, size);
    return;
  }
  PendingVm p;
  p.dst = dst; p.size = size;
  if (src) memcpy(p.data, src, size); else memset(p.data, 0, size);
  S().cur->vm.push_back(p);
}

inline void yield() {
  State& s = S();
  Fiber* f = s.cur;
  tnv3_emu_swap(&f->sp, s.main_sp);
  threadIdx = f->tid;
}

inline void block_barrier() {
  State& s = S();
  Fiber* f = s.cur;
  unsigned g = s.block_gen;
  if (++s.block_count == s.nthreads) { s.block_count = 0; ++s.block_gen; return; }
  f->wait_block = true; f->wait_gen = g;
  while (s.block_gen == g) yield();
  f->wait_block = false;
}

inline void wave_barrier() {
  State& s = S();
  unsigned w = wave_id();
  unsigned g = s.wave_gen[w];
  if (++s.wave_count[w] == s.wave_size[w]) { s.wave_count[w] = 0; ++s.wave_gen[w]; return; }
  while (s.wave_gen[w] == g) yield();
}

inline void fiber_entry() {
  State& s = S();
  threadIdx = s.cur->tid;
  s.body();
  vm_retire(*s.cur, 0);          // s_endpgm: everything in flight completes
  s.cur->done = true;
  tnv3_emu_swap(&s.cur->sp, s.main_sp);
  abort();   // a finished fiber is never resumed
}

// Run `body` for every work-item of a grid.
inline void launch(emu_dim3 grid, emu_dim3 block, std::function<void()> body) {
  State& s = S();
  const size_t STACK = 256 * 1024;
  s.nthreads = block.x * block.y * block.z;
  { const char* e = getenv("TNV3_EMU_LAZY_DMA"); s.lazy = e && e[0] == '1'; }
  if (s.nthreads > MAX_WAVES * WAVE) { fprintf(stderr, "emu: block too large\n"); abort(); }
  gridDim = grid; blockDim = block;
  s.body = std::move(body);
  if (s.fibers.size() < s.nthreads) s.fibers.resize(s.nthreads);
  for (unsigned i = 0; i < s.nthreads; ++i)
    if (!s.fibers[i].stack) s.fibers[i].stack = (char*)malloc(STACK);
  unsigned nwaves = (s.nthreads + WAVE - 1) / WAVE;
  for (unsigned bz = 0; bz < grid.z; ++bz)
  for (unsigned by = 0; by < grid.y; ++by)
  for (unsigned bx = 0; bx < grid.x; ++bx) {
    blockIdx = {bx, by, bz};
    s.block_gen = 0; s.block_count = 0;
    for (unsigned w = 0; w < nwaves; ++w) {
      s.wave_gen[w] = 0; s.wave_count[w] = 0;
      unsigned rem = s.nthreads - w * WAVE; s.wave_size[w] = rem < WAVE ? rem : WAVE;
    }
    for (unsigned i = 0; i < s.nthreads; ++i) {
      Fiber& f = s.fibers[i];
      f.done = false; f.wait_block = false; f.vm.clear();
      f.tid = {i % block.x, (i / block.x) % block.y, i / (block.x * block.y)};
      // initial frame: six zeroed callee-saved registers, then the entry address that `ret` jumps to;
      // after that `ret`, rsp % 16 == 8 as the ABI requires at a function entry.
      uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
      void** frame = (void**)(top - 64);
      for (int q = 0; q < 6; ++q) frame[q] = nullptr;
      frame[6] = (void*)&fiber_entry;
      frame[7] = nullptr;
      f.sp = (void*)frame;
    }
    unsigned remaining = s.nthreads;
    unsigned long guard = 0;
    while (remaining) {
      for (unsigned w = 0; w < nwaves; ++w) {
        for (;;) {   // run this wave until all its lanes are done or parked at a block barrier
          bool any = false;
          for (unsigned l = 0; l < s.wave_size[w]; ++l) {
            Fiber& f = s.fibers[w * WAVE + l];
            if (f.done) continue;
            if (f.wait_block && f.wait_gen == s.block_gen) continue;
            s.cur = &f; threadIdx = f.tid;
            tnv3_emu_swap(&s.main_sp, f.sp);
            any = true;
            if (f.done) --remaining;
          }
          if (!any) break;
          if (++guard > (1ul << 34)) { fprintf(stderr, "emu: livelock (divergent barrier?)\n"); abort(); }
        }
      }
    }
  }
}

template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

// generic "read value of lane src" with all lanes participating
template <class T> inline T wave_read(T v, unsigned src_lane) {
  State& s = S(); unsigned w = wave_id(), l = lane_id();
  s.xchg[w][0][l] = to_bits(v);
  wave_barrier();
  T r = (src_lane < s.wave_size[w]) ? from_bits<T>(s.xchg[w][0][src_lane]) : v;
  wave_barrier();
  return r;
}
}  // namespace emu

// ------------------------------------------------------------------ HIP surface used by the kernels
inline void __syncthreads() { emu::block_barrier(); }
inline void __builtin_amdgcn_s_barrier() { emu::block_barrier(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }   // only the diag twins (never instantiated here) read the clock
// gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4 (expcnt imm[6:4], lgkmcnt imm[11:8]); LDS and scalar operations are synchronous here
inline void __builtin_amdgcn_s_waitcnt(int imm) {
  if (emu::lazy_dma()) emu::vm_retire(*emu::S().cur, (size_t)((imm & 15) | (((imm >> 14) & 3) << 4)));
}
inline void __builtin_amdgcn_s_sleep(int) {}
inline void __threadfence() {}

template <class T> inline T __shfl(T v, int src, int width = 64) {
  unsigned l = emu::lane_id(); unsigned base = l - (l % width);
  return emu::wave_read(v, base + (unsigned(src) % width));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  unsigned l = emu::lane_id(); unsigned t = l ^ unsigned(mask);
  if ((t / width) != (l / width)) t = l;
  return emu::wave_read(v, t);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  unsigned l = emu::lane_id(); unsigned t = l + d;
  if ((t / width) != (l / width)) t = l;
  return emu::wave_read(v, t);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  unsigned l = emu::lane_id(); int t = int(l) - int(d);
  if (t < 0 || (unsigned(t) / width) != (l / width)) t = l;
  return emu::wave_read(v, unsigned(t));
}
inline unsigned long long __ballot(int pred) {
  emu::State& s = emu::S(); unsigned w = emu::wave_id(), l = emu::lane_id();
  s.xchg[w][0][l] = pred ? 1 : 0;
  emu::wave_barrier();
  unsigned long long m = 0;
  for (unsigned i = 0; i < s.wave_size[w]; ++i) if (s.xchg[w][0][i]) m |= 1ull << i;
  emu::wave_barrier();
  return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) { emu::State& s = emu::S(); unsigned n = s.wave_size[emu::wave_id()];
  unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1); return __ballot(p) == full; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __builtin_amdgcn_readfirstlane(int v) { return emu::wave_read(v, 0); }

typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));

inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
  emu::State& s = emu::S(); unsigned w = emu::wave_id(), l = emu::lane_id();
  if (s.wave_size[w] != 64) { fprintf(stderr, "emu: MFMA needs a full wave\n"); abort(); }
  s.xchg[w][1][l] = emu::to_bits(a); s.xchg[w][2][l] = emu::to_bits(b);
  emu::wave_barrier();
  emu_f32x16 d;
  for (int r = 0; r < 16; ++r) {
    unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float acc = c[r];
    for (unsigned k = 0; k < 2; ++k)
      acc = fmaf(emu::from_bits<float>(s.xchg[w][1][row + 32 * k]), emu::from_bits<float>(s.xchg[w][2][col + 32 * k]), acc);
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  emu::State& s = emu::S(); unsigned w = emu::wave_id(), l = emu::lane_id();
  if (s.wave_size[w] != 64) { fprintf(stderr, "emu: MFMA needs a full wave\n"); abort(); }
  s.xchg[w][1][l] = emu::to_bits(a); s.xchg[w][2][l] = emu::to_bits(b);
  emu::wave_barrier();
  emu_f32x4 d;
  for (int r = 0; r < 4; ++r) {
    unsigned row = 4 * (l >> 4) + r, col = l & 15;
    float acc = c[r];
    for (unsigned k = 0; k < 4; ++k)
      acc = fmaf(emu::from_bits<float>(s.xchg[w][1][row + 16 * k]), emu::from_bits<float>(s.xchg[w][2][col + 16 * k]), acc);
    d[r] = acc;
  }
  emu::wave_barrier();
  return d;
}

// LDS DMA: LDS[base + lane*size] <- *gsrc (per-lane source, wave-uniform LDS base = lane 0's argument)
inline void __builtin_amdgcn_global_load_lds(const __attribute__((address_space(1))) void* g, __attribute__((address_space(3))) void* l,
                                             unsigned size, int offset, unsigned) {
  const uintptr_t base = emu::wave_read((uintptr_t)l, 0);
  emu::vm_dma((void*)(base + (uintptr_t)emu::lane_id() * size + offset), (const void*)((uintptr_t)g + offset), size);
}

// buffer_load ... lds through a raw buffer descriptor (kernels/conv3x3_wino3_mfma.h): LDS[base + lane*16] <- buffer[voffset..+16),
// zeros when the access is out of the descriptor's range (the hardware bounds check the kernels use for zero padding)
struct tnv3_rsrc_t { const char* base; unsigned num_records; };
inline tnv3_rsrc_t tnv3_make_rsrc(const void* base, unsigned bytes) { return tnv3_rsrc_t{(const char*)base, bytes}; }
// The range check of a raw buffer's multi-dword access is PER DWORD (ISA: "Load/store-Dword-x{2,3,4} perform range-check on a per-dword
// basis"): a piece that straddles the end of the descriptor's range keeps its leading in-range dwords (kernels/conv3x3_wino43s_mfma.h
// relies on it for the last row of the last channel).  No 32-bit wrap: a voffset that "points before the base" is out of range entirely.
inline void tnv3_buf_dma16(tnv3_rsrc_t r, float* lds_base, unsigned voffset) {
  const uintptr_t base = emu::wave_read((uintptr_t)lds_base, 0);
  void* dst = (void*)(base + (uintptr_t)emu::lane_id() * 16);
  if ((unsigned long long)voffset + 16ull <= (unsigned long long)r.num_records) { emu::vm_dma(dst, r.base + voffset, 16); return; }
  unsigned char tmp[16];
  memset(tmp, 0, 16);
  bool any = false;
  for (int d = 0; d < 4; ++d)
    if ((unsigned long long)voffset + 4ull * d + 4ull <= (unsigned long long)r.num_records) { memcpy(tmp + 4 * d, r.base + voffset + 4 * d, 4); any = true; }
  if (!any) { emu::vm_dma(dst, nullptr, 16); return; }
  emu::vm_dma(dst, tmp, 16);
}

// 8-byte buffer load / store through a descriptor: address = base + voffset (per lane) + soffset (scalar); out-of-range loads give 0,
// out-of-range stores are dropped
typedef float tnv3_f2 __attribute__((ext_vector_type(2)));
inline tnv3_f2 tnv3_buf_load_f2(tnv3_rsrc_t r, unsigned voffset, unsigned soffset) {
  emu::vm_placeholder();
  tnv3_f2 v = {0.0f, 0.0f};
  const unsigned long long o = (unsigned long long)voffset + soffset;
  if (o + 8ull <= (unsigned long long)r.num_records) memcpy(&v, r.base + o, 8);
  return v;
}
typedef float tnv3_f4 __attribute__((ext_vector_type(4)));
inline tnv3_f4 tnv3_buf_load_f4(tnv3_rsrc_t r, unsigned voffset, unsigned soffset) {
  emu::vm_placeholder();
  tnv3_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned long long o = (unsigned long long)voffset + soffset;
  if (o + 16ull <= (unsigned long long)r.num_records) memcpy(&v, r.base + o, 16);
  return v;
}
inline void tnv3_buf_store_f2(tnv3_rsrc_t r, unsigned voffset, unsigned soffset, tnv3_f2 v) {
  emu::vm_placeholder();
  const unsigned long long o = (unsigned long long)voffset + soffset;
  if (o + 8ull <= (unsigned long long)r.num_records) memcpy(const_cast<char*>(r.base) + o, &v, 8);
}
inline void tnv3_buf_store_f4(tnv3_rsrc_t r, unsigned voffset, unsigned soffset, tnv3_f4 v) {
  emu::vm_placeholder();
  const unsigned long long o = (unsigned long long)voffset + soffset;
  if (o + 16ull <= (unsigned long long)r.num_records) memcpy(const_cast<char*>(r.base) + o, &v, 16);
}

// atomics (the emulator is single-threaded: plain read-modify-write)
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __cosf(float x) { return cosf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __fmul_rn(float a, float b) {
#pragma clang fp contract(off)
  volatile float r = a * b;      // a rounded product that no later add may absorb into an FMA
  return r;
}
inline float __fadd_rn(float a, float b) {
#pragma clang fp contract(off)
  volatile float r = a + b;
  return r;
}
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
template <class T> inline T __ldg(const T* p) { return *p; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
