// tnv3_capi.hip -- libtnv3_hip.so: gfx950 kernels + C ABI (include/tracknetv3_hip.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see tracknetv3_amd/_build.py).
#include <hip/hip_runtime.h>

#include "../../include/tracknetv3_hip.h"
#include "tnv3_impl.h"

namespace {
struct Launcher {
  hipStream_t stream;
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
inline Launcher make_launcher(tnv3_stream_t s) { return Launcher{static_cast<hipStream_t>(s)}; }

// One-time device query behind the split-K planner (a partitioned GPU exposes fewer CUs).
struct CuCountInit {
  CuCountInit() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      tnv3::num_cus() = n;
  }
};
inline void init_cu_count() { static CuCountInit once; (void)once; }
}  // namespace

#include "tnv3_capi_body.inc"
