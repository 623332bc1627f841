// Launch functions the kernel translation units export to the host side of the library (pinkhip.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "dispatch.h"
#include "ik_common.h"

#define PINKHIP_PASTE5(a, b, c, d, e) a##b##_##c##_##d##e
#define PINKHIP_LAUNCH_PACKED_NAME(NV, W, D) PINKHIP_PASTE5(launch_packed_, NV, W, D, )
#define PINKHIP_PASTE4(a, b, c) a##b##_##c
#define PINKHIP_LAUNCH_ROLLOUT_NAME(NV, W) PINKHIP_PASTE4(launch_rollout_, NV, W)

#define PINKHIP_PASTE6(a, b, c, d) a##b##_##c##_##d
#define PINKHIP_LAUNCH_SWEEP_NAME(NV, MD, W) PINKHIP_PASTE6(launch_sweep_, NV, MD, W)

namespace pinkhip {

// tu_sweep.hip: the sweep-tableau stack + solve kernel, one launcher per entry of PINKHIP_SWEEP_TABLE
#define PINKHIP_DECLARE(NV, MD, W) hipError_t PINKHIP_LAUNCH_SWEEP_NAME(NV, MD, W)(hipStream_t stream, const KernelArgs &a);
PINKHIP_SWEEP_TABLE(PINKHIP_DECLARE)
#undef PINKHIP_DECLARE

// tu_sweepx.hip: the same with virtual dense rows, one launcher per entry of PINKHIP_SWEEPX_TABLE
#define PINKHIP_LAUNCH_SWEEPX_NAME(NV, MD, W) PINKHIP_PASTE6(launch_sweepx_, NV, MD, W)
#define PINKHIP_DECLARE(NV, MD, W) hipError_t PINKHIP_LAUNCH_SWEEPX_NAME(NV, MD, W)(hipStream_t stream, const KernelArgs &a);
PINKHIP_SWEEPX_TABLE(PINKHIP_DECLARE)
#undef PINKHIP_DECLARE

#define PINKHIP_DECLARE(NV, W)                                                                  \
  hipError_t PINKHIP_LAUNCH_PACKED_NAME(NV, W, 0)(hipStream_t stream, const KernelArgs &a);     \
  hipError_t PINKHIP_LAUNCH_PACKED_NAME(NV, W, 1)(hipStream_t stream, const KernelArgs &a);
PINKHIP_PACKED_TABLE(PINKHIP_DECLARE)
#undef PINKHIP_DECLARE

// tu_rollout.hip: the whole-control-step kernel, one launcher per entry of PINKHIP_ROLLOUT_TABLE
struct RolloutArgs;
#define PINKHIP_DECLARE(NV, W) hipError_t PINKHIP_LAUNCH_ROLLOUT_NAME(NV, W)(hipStream_t stream, const RolloutArgs &a);
PINKHIP_ROLLOUT_TABLE(PINKHIP_DECLARE)
#undef PINKHIP_DECLARE
#define PINKHIP_LAUNCH_ROLLOUT_DENSE_NAME(NV, MD, W) PINKHIP_PASTE6(launch_rollout_dense_, NV, MD, W)
#define PINKHIP_DECLARE(NV, MD, W) hipError_t PINKHIP_LAUNCH_ROLLOUT_DENSE_NAME(NV, MD, W)(hipStream_t stream, const RolloutArgs &a);
PINKHIP_ROLLOUT_DENSE_TABLE(PINKHIP_DECLARE)
#undef PINKHIP_DECLARE




}  // namespace pinkhip
