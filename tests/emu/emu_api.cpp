// emu_api.cpp -- the C ABI of include/tracknetv3_hip.h built on the host SIMT emulator (hip_emu.h).
// TEST TOOL: lets the CPU test-suite run the unchanged kernel headers and the unchanged dispatch code
// (tnv3_impl.h) on host memory.  Never loaded by the product package on its own.
// Build: clang++ -std=c++17 -O2 -shared -fPIC -include tests/emu/hip_emu.h tests/emu/emu_api.cpp
#include "../../include/tracknetv3_hip.h"
#include "../../tracknetv3_amd/csrc/tnv3_impl.h"

#include <stdlib.h>

namespace {
struct Launcher {
  template <class... KArgs, class... Args>
  int launch(void (*kernel)(KArgs...), int grid, int block, Args... args) {
    return launch3(kernel, grid, 1, 1, block, args...);
  }
  template <class... KArgs, class... Args>
  int launch3(void (*kernel)(KArgs...), int gx, int gy, int gz, int block, Args... args) {
    emu::launch(emu_dim3{(unsigned)gx, (unsigned)gy, (unsigned)gz}, emu_dim3{(unsigned)block, 1, 1},
                [=]() { kernel(static_cast<KArgs>(args)...); });
    return TNV3_OK;
  }
};
// the emulator plans for the default 256 CUs; TNV3_EMU_CUS overrides (small values make the persistent kernels walk several tiles
// per workgroup on small test shapes)
inline void init_cu_count() {
  const char* e = getenv("TNV3_EMU_CUS");
  tnv3::num_cus() = e && atoi(e) > 0 ? atoi(e) : 256;
}
inline Launcher make_launcher(tnv3_stream_t) { init_cu_count(); return Launcher{}; }
}  // namespace

#define TNV3_TU_ALL 1
#include "../../tracknetv3_amd/csrc/tnv3_capi_body.inc"

extern "C" int tnv3_is_emulator(void) { return 1; }

// ConvTileWalk (kernels/conv3x3_mfma.h: the persistent kernels' division-free walk through the block list) against
// conv_block_map, brute force over channel-block counts, image / tile-grid shapes, grid sizes and starting blocks.
// Returns the number of (walk, block) pairs compared, or -1 at the first disagreement.
extern "C" long tnv3_emu_tile_walk_check(void) {
  long checked = 0;
  for (int nMB = 1; nMB <= 9; ++nMB)
    for (int N = 1; N <= 4; ++N)
      for (int tilesH = 1; tilesH <= 5; ++tilesH)
        for (int tilesW = 1; tilesW <= 4; ++tilesW) {
          const int nPT = N * tilesH * tilesW, items = tnv3::conv_grid_blocks(nMB, nPT);
          const bool xcd = nMB <= 8 && 8 % nMB == 0;
          for (int G = 8; G <= 64; G += 8) {
            const int grid = items < G ? items : G;
            if (xcd && grid % 8) continue;
            for (int b0 = 0; b0 < grid; ++b0) {
              tnv3::ConvTileWalk w;
              w.init(b0, grid, nMB, nPT, tilesH, tilesW);
              bool ended = false;
              for (int b = b0; b < items + 2 * grid; b += grid) {
                int mb = 0, pt = 0;
                const bool ok = b < items && tnv3::conv_block_map(b, nMB, nPT, mb, pt);
                if (ended) { if (ok) return -1; continue; }       // the kernels stop at the first invalid entry: none may follow
                if (ok != w.valid) return -1;
                if (!ok) { ended = true; continue; }
                const int tpi = tilesH * tilesW, n = pt / tpi, rem = pt % tpi;
                if (mb != w.mb || pt != w.pt || n != w.n || rem / tilesW != w.trow || rem % tilesW != w.tcol) return -1;
                ++checked;
                w.next();
              }
            }
          }
        }
  return checked;
}
