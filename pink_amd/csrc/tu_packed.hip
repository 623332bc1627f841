// One translation unit per instantiation of the stack+solve kernel: compiled with
//   -DPINKHIP_TU_NV=<NV> -DPINKHIP_TU_W=<W> -DPINKHIP_TU_DENSE=<0|1>
// (Makefile).  The fully unrolled register-resident rows make every instantiation a long compile; as
// separate objects they build in parallel and only the host file is touched by an ABI change.
#include <hip/hip_runtime.h>

// clang-format off
#include "wave.h"
#include "ik_kernels_packed.h"
#include "launchers.h"
// clang-format on

#if !defined(PINKHIP_TU_NV) || !defined(PINKHIP_TU_W) || !defined(PINKHIP_TU_DENSE)
#error "tu_packed.hip is compiled once per (NV, W, DENSE): see the Makefile"
#endif

namespace pinkhip {

hipError_t PINKHIP_LAUNCH_PACKED_NAME(PINKHIP_TU_NV, PINKHIP_TU_W, PINKHIP_TU_DENSE)(hipStream_t stream, const KernelArgs &a) {
  constexpr int NV = PINKHIP_TU_NV, W = PINKHIP_TU_W, G = kWave / W;
  constexpr bool DENSE = PINKHIP_TU_DENSE != 0;
  const size_t lds = static_cast<size_t>(LdsP<NV>::bytes(DENSE ? a.md : 0, G));
  const dim3 grid(static_cast<unsigned>((a.B + G - 1) / G)), block(kWave);
  hipLaunchKernelGGL((ik_solve_packed_kernel<NV, W, DENSE>), grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace pinkhip
