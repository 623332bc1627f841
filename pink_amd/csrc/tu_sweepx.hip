// One translation unit per instantiation of the sweep-tableau kernel with virtual dense rows (ik_sweepx.h): compiled with
//   -DPINKHIP_TU_NV=<NV> -DPINKHIP_TU_MD=<MD> -DPINKHIP_TU_W=<W>       (Makefile, SWEEPX list)
#include <hip/hip_runtime.h>

// clang-format off
#include "wave.h"
#include "ik_sweepx.h"
#include "launchers.h"
// clang-format on

#if !defined(PINKHIP_TU_NV) || !defined(PINKHIP_TU_MD) || !defined(PINKHIP_TU_W)
#error "tu_sweepx.hip is compiled once per (NV, MD, W): see the Makefile"
#endif

namespace pinkhip {

hipError_t PINKHIP_LAUNCH_SWEEPX_NAME(PINKHIP_TU_NV, PINKHIP_TU_MD, PINKHIP_TU_W)(hipStream_t stream, const KernelArgs &a) {
  constexpr int NV = PINKHIP_TU_NV, MD = PINKHIP_TU_MD, W = PINKHIP_TU_W, G = kWave / W;
  const dim3 grid(static_cast<unsigned>((a.B + G - 1) / G)), block(kWave);
  using SL = SweepXLds<NV, MD, W>;
  static_assert(sweepx_lds_doubles(NV, MD, W) == SL::stride, "dispatch.h restates the LDS layout");
  KernelArgs k = a;
  k.lds_pitch = sweepx_kernel_lds_doubles<NV, MD, W>(a.md);  // (room for the hand-over to the Goldfarb-Idnani kernel)
  const size_t lds = 8 * static_cast<size_t>(k.lds_pitch) * G + 16;
  hipLaunchKernelGGL((ik_solve_sweepx_kernel<NV, MD, W>), grid, block, lds, stream, k);
  return hipGetLastError();
}

}  // namespace pinkhip

#if defined(PINKHIP_SECTION_CLOCK) && defined(PINKHIP_CLOCK_SWEEPX)
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
