// One slice of the heavy kernel instantiations of the CPU wave emulator -- TEST INFRASTRUCTURE ONLY.
//   g++ -c -DEMU_PART=k -DEMU_NPARTS=n emu_part.cpp    for k = 0 .. n-1 (tests/emu builds them in parallel)
#include "emu_lanes.h"

#ifndef EMU_PART
#define EMU_PART 0
#endif
#ifndef EMU_NPARTS
#define EMU_NPARTS 1
#endif

namespace {

// (dependent on the template parameter P so that the branches of the other slices are discarded, not instantiated)
template <int P, int K, int NV, int MD, int W>
constexpr bool mine() {
  return ((K * 5 + NV * 3 + MD * 7 + W / 16) % EMU_NPARTS) == P;
}

template <int P>
void register_slice() {
  using namespace pinkemu;
#define PINKHIP_CASE(NV, W) \
  if constexpr (mine<P, KIND_PACKED, NV, 0, W>()) emu_register(KIND_PACKED, NV, 0, W, &lane_main_packed<NV, W>);
  PINKHIP_PACKED_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
#define PINKHIP_CASE(NV, MD, W) \
  if constexpr (mine<P, KIND_SWEEP, NV, MD, W>()) emu_register(KIND_SWEEP, NV, MD, W, &lane_main_sweep<NV, MD, W>);
  PINKHIP_SWEEP_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
#define PINKHIP_CASE(NV, MD, W) \
  if constexpr (mine<P, KIND_SWEEPX, NV, MD, W>()) emu_register(KIND_SWEEPX, NV, MD, W, &lane_main_sweepx<NV, MD, W>);
  PINKHIP_SWEEPX_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
#define PINKHIP_CASE(NV, W) \
  if constexpr (mine<P, KIND_ROLLOUT, NV, 0, W>()) emu_register(KIND_ROLLOUT, NV, 0, W, &lane_main_rollout<NV, W>);
  PINKHIP_ROLLOUT_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
#define PINKHIP_CASE(NV, MD, W) \
  if constexpr (mine<P, KIND_ROLLOUT_DENSE, NV, MD, W>()) emu_register(KIND_ROLLOUT_DENSE, NV, MD, W, &lane_main_rollout_dense<NV, MD, W>);
  PINKHIP_ROLLOUT_DENSE_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
}

struct Registrar {
  Registrar() { register_slice<EMU_PART>(); }
} registrar;

}  // namespace
