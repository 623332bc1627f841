// Per-lane entry points of the kernels under the CPU wave emulator -- TEST INFRASTRUCTURE ONLY (see wave_emu.h).
// The heavy instantiations (stack + solve kernels, whole-step kernels) are compiled in emu_part.cpp, one slice of the
// dispatch tables per translation unit so that the slices build in parallel, and found through emu_registry.h.
#pragma once
#include "wave_emu.h"
// clang-format off
#include "../../pink_amd/csrc/ik_common.h"
#include "../../pink_amd/csrc/dispatch.h"
#include "../../pink_amd/csrc/ik_kernels_packed.h"
#include "../../pink_amd/csrc/ik_sweep.h"
#include "../../pink_amd/csrc/ik_sweepx.h"
#include "../../pink_amd/csrc/ik_stack_mfma.h"
#include "../../pink_amd/csrc/ik_frame_task.h"
#include "../../pink_amd/csrc/ik_kinematics.h"
#include "../../pink_amd/csrc/ik_rollout.h"
#include "../../pink_amd/csrc/model_tables.h"
#include "../../pink_amd/csrc/host_tables.h"
// clang-format on

#include <string>


#include "emu_registry.h"

namespace pinkemu {

using pinkhip::KernelArgs;

template <int NV, int W>
void lane_main_packed(void *p) {
  const KernelArgs *a = static_cast<const KernelArgs *>(p);
  if (a->md == 0)
    pinkhip::ik_packed_instance<NV, W, false>(*a, pinkhip::block_id());
  else
    pinkhip::ik_packed_instance<NV, W, true>(*a, pinkhip::block_id());
}

template <int NV, int MD, int W>
void lane_main_sweep(void *p) {
  KernelArgs k = *static_cast<const KernelArgs *>(p);
  k.lds_pitch = pinkhip::sweep_kernel_lds_doubles<NV, MD, W>(k.md);  // as tu_sweep.hip's launcher
  pinkhip::ik_solve_sweep_body<NV, MD, W>(k, pinkhip::block_id());
}

template <int NV, int MD, int W>
void lane_main_sweepx(void *p) {
  KernelArgs k = *static_cast<const KernelArgs *>(p);
  k.lds_pitch = pinkhip::sweepx_kernel_lds_doubles<NV, MD, W>(k.md);  // as tu_sweepx.hip's launcher
  pinkhip::ik_solve_sweepx_body<NV, MD, W>(k, pinkhip::block_id());
}

template <int TP>
void lane_main_stack_small(void *p) {
  pinkhip::ik_stack_small_instance<TP>(*static_cast<const KernelArgs *>(p), pinkhip::block_id());
}

template <int NT>
void lane_main_stack_mfma(void *p) {
  const KernelArgs *a = static_cast<const KernelArgs *>(p);
  if constexpr (NT >= 3) {
    if (pinkhip::stack_staged_ok(a->nv, a->Kd, a->J)) {  // same rule as pinkhip.hip
      pinkhip::ik_stack_mfma_instance<NT, true>(*a, pinkhip::block_id());
      return;
    }
  }
  pinkhip::ik_stack_mfma_instance<NT>(*a, pinkhip::block_id());
}

template <int W>
void lane_main_frame(void *p) {
  pinkhip::ik_frame_task_instance<W>(*static_cast<const pinkhip::FrameTaskArgs *>(p), pinkhip::block_id());
}

template <int W>
void lane_main_fk(void *p) {
  pinkhip::ik_fk_instance<W>(*static_cast<const pinkhip::FkArgs *>(p), pinkhip::block_id());
}

template <int W>
void lane_main_fk_fused(void *p) {
  pinkhip::ik_fk_instance<W, true>(*static_cast<const pinkhip::FkArgs *>(p), pinkhip::block_id());
}

template <int W>
void lane_main_step(void *p) {
  pinkhip::ik_fk_instance<W, true, true>(*static_cast<const pinkhip::FkArgs *>(p), pinkhip::block_id());
}

template <int NV, int W>
void lane_main_rollout(void *p) {
  pinkhip::ik_rollout_instance<NV, 0, W>(*static_cast<const pinkhip::RolloutArgs *>(p), pinkhip::block_id());
}
template <int NV, int MD, int W>
void lane_main_rollout_dense(void *p) {
  pinkhip::ik_rollout_instance<NV, MD, W>(*static_cast<const pinkhip::RolloutArgs *>(p), pinkhip::block_id());
}


}  // namespace pinkemu
