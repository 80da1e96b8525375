// tnv3_capi.hip -- libtnv3_hip.so: gfx950 kernels + C ABI (include/tracknetv3_hip.h).
// Built by tracknetv3_amd/_build.py: this file is compiled once per kernel family (-DTNV3_TU_MISC, _CONV, _WINO, _WINO43, _UP2X, _WGRAD,
// _TRAIN; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c, in parallel) and the objects are linked into one shared library.
// With -DTNV3_DIAG -DTNV3_TU_DIAG it becomes libtnv3_diag.so (include/tracknetv3_hip_diag.h: timing twins, MFMA probe).
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/tracknetv3_hip.h"
#ifdef TNV3_DIAG
#include "../../include/tracknetv3_hip_diag.h"
#endif
#include "tnv3_impl.h"
#ifdef TNV3_DIAG
#include "kernels/coissue_probe.h"
#endif

namespace {
// CU count of a device (a partitioned GPU exposes fewer CUs), queried once per device.
inline int cus_of_device(int dev) {
  constexpr int kMaxDev = 64;
  static std::atomic<int> cache[kMaxDev];
  if (dev < 0 || dev >= kMaxDev) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

// An entry point must not assume the calling thread's "current" device (the autograd engine's backward thread and
// multi-device callers differ from the main thread): the stream decides.  For the duration of the call the current device
// becomes the stream's (restored on return), and the planners see that device's CU count.
struct Launcher {
  hipStream_t stream;
  int restore = -1;
  Launcher(hipStream_t s)   // NOLINT: implicit on purpose (`Launcher L = make_launcher(stream);`)
      : stream(s) {
    int cur = 0, dev = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    dev = cur;
    if (s != nullptr && hipStreamGetDevice(s, &dev) != hipSuccess) dev = cur;
    if (dev != cur && hipSetDevice(dev) == hipSuccess) restore = cur;
    tnv3::num_cus() = cus_of_device(dev);
  }
  Launcher(const Launcher&) = delete;
  Launcher& operator=(const Launcher&) = delete;
  ~Launcher() {
    if (restore >= 0) (void)hipSetDevice(restore);
  }
  template <class... KArgs, class... Args>
  int launch(void (*kernel)(KArgs...), int grid, int block, Args... args) {
    return launch3(kernel, grid, 1, 1, block, args...);
  }
  template <class... KArgs, class... Args>
  int launch3(void (*kernel)(KArgs...), int gx, int gy, int gz, int block, Args... args) {
    hipLaunchKernelGGL(kernel, dim3(gx, gy, gz), dim3(block), 0, stream, static_cast<KArgs>(args)...);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) TNV3_FAIL(TNV3_E_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
    return TNV3_OK;
  }
};
#define make_launcher(s) (static_cast<hipStream_t>(s))   /* `Launcher L = make_launcher(stream);` constructs in place */

// Stream-less size queries plan for the calling thread's current device.
inline void init_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  tnv3::num_cus() = cus_of_device(dev);
}
}  // namespace

#include "tnv3_capi_body.inc"
