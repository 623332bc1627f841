// One translation unit per instantiation of the whole-control-step kernel (ik_rollout.h): compiled with
//   -DPINKHIP_TU_NV=<NV> -DPINKHIP_TU_W=<W>       (Makefile, ROLLOUT list)
#include <hip/hip_runtime.h>

// clang-format off
#define PINKHIP_NO_ELEMENTWISE_KERNELS
#include "wave.h"
#include "ik_rollout.h"
#include "launchers.h"
// clang-format on

#if !defined(PINKHIP_TU_NV) || !defined(PINKHIP_TU_W)
#error "tu_rollout.hip is compiled once per (NV, W) or (NV, MD, W): see the Makefile"
#endif
#ifndef PINKHIP_TU_MD
#define PINKHIP_TU_MD 0  // rows of position barriers formed on chip
#endif

namespace pinkhip {

#if PINKHIP_TU_MD == 0
hipError_t PINKHIP_LAUNCH_ROLLOUT_NAME(PINKHIP_TU_NV, PINKHIP_TU_W)(hipStream_t stream, const RolloutArgs &a) {
#else
hipError_t PINKHIP_LAUNCH_ROLLOUT_DENSE_NAME(PINKHIP_TU_NV, PINKHIP_TU_MD, PINKHIP_TU_W)(hipStream_t stream, const RolloutArgs &a) {
#endif
  constexpr int NV = PINKHIP_TU_NV, MD = PINKHIP_TU_MD, W = PINKHIP_TU_W, G = kWave / W;
  constexpr bool kVirtual = NV + MD > W;  // (dense rows without lanes of their own: ik_sweepx.h)
  static_assert(kVirtual ? sweepx_lds_doubles(NV, MD, W) == SweepXLds<NV, MD, W>::stride
                         : sweep_lds_doubles(NV, MD, W) == SweepLds<NV, MD, W>::stride, "dispatch.h restates the LDS layout");
  static_assert(packed_lds_doubles(NV, MD) == LdsP<NV>::stride(MD), "dispatch.h restates the LDS layout");
  const size_t lds = 8 * static_cast<size_t>(a.k.lds_pitch) * G + 16;
  const dim3 grid(static_cast<unsigned>((a.k.B + G - 1) / G)), block(kWave);
  hipLaunchKernelGGL((ik_rollout_kernel<NV, MD, W>), grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace pinkhip
