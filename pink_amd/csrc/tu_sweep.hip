// One translation unit per instantiation of the sweep-tableau stack + solve kernel (ik_sweep.h): compiled with
//   -DPINKHIP_TU_NV=<NV> -DPINKHIP_TU_MD=<MD> -DPINKHIP_TU_W=<W>       (Makefile, SWEEP list)
#include <hip/hip_runtime.h>

// clang-format off
#include "wave.h"
#include "ik_sweep.h"
#include "launchers.h"
// clang-format on

#if !defined(PINKHIP_TU_NV) || !defined(PINKHIP_TU_MD) || !defined(PINKHIP_TU_W)
#error "tu_sweep.hip is compiled once per (NV, MD, W): see the Makefile"
#endif

namespace pinkhip {

hipError_t PINKHIP_LAUNCH_SWEEP_NAME(PINKHIP_TU_NV, PINKHIP_TU_MD, PINKHIP_TU_W)(hipStream_t stream, const KernelArgs &a) {
  constexpr int NV = PINKHIP_TU_NV, MD = PINKHIP_TU_MD, W = PINKHIP_TU_W, G = kWave / W;
  const dim3 grid(static_cast<unsigned>((a.B + G - 1) / G)), block(kWave);
  // LDS: the stated problem (H packed, c, columns of G) parked for the closing refinement step
  using SL = SweepLds<NV, MD, W>;
  static_assert(sweep_lds_doubles(NV, MD, W) == SL::stride, "dispatch.h restates the LDS layout");
  // (front coordinates eliminated: their recovery data sits behind the area of the W-coordinate tableau)
  static_assert(NV <= W || sweep_kernel_lds_doubles<NV, MD, W>(0) >= SweepLds<(NV <= W ? NV : W), 0, W>::stride + 2 * W + 8, "LDS of the elimination");
  KernelArgs k = a;
  k.lds_pitch = sweep_kernel_lds_doubles<NV, MD, W>(a.md);  // (room for the hand-over to the Goldfarb-Idnani kernel)
  const size_t lds = 8 * static_cast<size_t>(k.lds_pitch) * G + 16;
  hipLaunchKernelGGL((ik_solve_sweep_kernel<NV, MD, W>), grid, block, lds, stream, k);
  return hipGetLastError();
}

}  // namespace pinkhip

#if defined(PINKHIP_SECTION_CLOCK) && defined(PINKHIP_CLOCK_SWEEP)
// profiling builds only (scripts/section_clock.py): read and clear the per-section cycle counters of this unit
extern "C" int pinkhip_debug_section_clock(void *handle_unused, unsigned long long *out16) {
  (void)handle_unused;
  if (!out16) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pinkhip_clock), 16 * sizeof(unsigned long long)) != hipSuccess) return -2;
  unsigned long long zero[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(pinkhip_clock), zero, sizeof(zero)) != hipSuccess) return -2;
  return 0;
}
#endif
