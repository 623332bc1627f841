// Registry of the emulator's per-lane kernel entry points -- TEST INFRASTRUCTURE ONLY.  emu_part.cpp registers the
// instantiations of its slice at load time; emu_kernels.cpp looks them up where the library's launcher would pick a kernel.
#pragma once

namespace pinkemu {

typedef void (*LaneEntry)(void *);
enum Kind { KIND_PACKED = 0, KIND_SWEEP, KIND_SWEEPX, KIND_ROLLOUT, KIND_ROLLOUT_DENSE };

void emu_register(int kind, int nv, int md, int w, LaneEntry fn);
LaneEntry emu_lookup(int kind, int nv, int md, int w);

}  // namespace pinkemu
