// One translation unit per instantiation of the stack+solve kernel: compiled with
//   -DPINKHIP_TU_NV=<NV> -DPINKHIP_TU_W=<W> -DPINKHIP_TU_DENSE=<0|1>
// (Makefile).  The fully unrolled register-resident rows make every instantiation a long compile; as
// separate objects they build in parallel and only the host file is touched by an ABI change.
#include <hip/hip_runtime.h>

#include <cstdlib>

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
  size_t lds = static_cast<size_t>(LdsP<NV>::bytes(DENSE ? a.md : 0, G));
#ifdef PINKHIP_SECTION_CLOCK
  // profiling builds only: PINKHIP_LDS_TOTAL=<bytes> asks for more LDS per wave to lower the occupancy (40000: one
  // wave per SIMD, 20000: two) -- per-section cycles of a wave that runs alone vs. among three
  if (const char *t = std::getenv("PINKHIP_LDS_TOTAL")) lds = static_cast<size_t>(std::atoll(t)) > lds ? static_cast<size_t>(std::atoll(t)) : lds;
#endif
  const dim3 grid(static_cast<unsigned>((a.B + G - 1) / G)), block(kWave);
  hipLaunchKernelGGL((ik_solve_packed_kernel<NV, W, DENSE>), grid, block, lds, stream, a);
  return hipGetLastError();
}

}  // namespace pinkhip

#ifndef PINKHIP_CLOCK_DENSE
#define PINKHIP_CLOCK_DENSE 0  // which of the two instantiations exports the accessor
#endif
#if defined(PINKHIP_SECTION_CLOCK) && !defined(PINKHIP_CLOCK_SWEEP) && !defined(PINKHIP_CLOCK_SWEEPX) && PINKHIP_TU_DENSE == PINKHIP_CLOCK_DENSE
// profiling builds only (scripts/section_clock.py, make DEV=1 SECTION_CLOCK=1): read and clear the per-section
// cycle counters of this translation unit's kernel
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
