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
