// Definitions shared by every kernel of the library: the kernarg struct, status codes, compile-time loop.
// Uses only the primitives of wave.h; the CPU wave emulator under tests/emu provides the same names to run the
// kernel sources unmodified in tests.
#pragma once

#include <cmath>
#include <cstdint>
#include <type_traits>

// Tuning knobs (compile-time): how many independent LDS loads are batched before
// their FMAs are pinned, and the occupancy the register allocator is told to aim for.
// Tuning (measured on MI355X, profiles/): the number of LDS loads batched before their FMAs
// are pinned is 8 for NV >= 24 and 4 below; occupancy targets (waves per SIMD, i.e. the VGPR
// budget handed to the register allocator) are set per kernel in wave.h.
#ifndef PINKHIP_GROUP_LARGE
#define PINKHIP_GROUP_LARGE 8  // NV >= 24
#endif
#ifndef PINKHIP_GROUP_SMALL
#define PINKHIP_GROUP_SMALL 4  // NV <= 16
#endif

// Development trace: compiled in only by the CPU wave emulator built with -DPINKHIP_TRACE (a printf per traced
// event); the HIP build never defines it.
#if defined(PINKHIP_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#define PINKHIP_TRACEF(cond, ...)        \
  do {                                   \
    if (cond) std::printf(__VA_ARGS__);  \
  } while (0)
#else
#define PINKHIP_TRACEF(cond, ...)
#endif

namespace pinkhip {

template <int NV>
constexpr int group_size() {
  return NV >= 24 ? PINKHIP_GROUP_LARGE : PINKHIP_GROUP_SMALL;
}

// Compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// "Transpose-reduce": every lane of a group holds N <= W partial values v(0..N-1); afterwards lane j of the group
// holds sum over the group's lanes of v(j) (lanes >= N: 0).  Recursive halving: at level b the lanes whose bit b
// is set keep the odd entries and hand the even ones to their partner (lane ^ 2^b), and vice versa, so the number
// of values halves with every exchange -- N + N/2 + ... moves instead of N full reductions (N log W).
template <int W, int N, int BIT, class F>
__device__ __forceinline__ double transpose_reduce(F v) {
  const int lane = lane_id();
  if constexpr ((1 << BIT) >= W) {
    static_assert(N == 1, "N <= W values fold to one per lane");
    return v(std::integral_constant<int, 0>{});
  } else {
    constexpr int H = (N + 1) / 2;
    double nx[H];
    const bool up = ((lane >> BIT) & 1) != 0;
    static_for<0, H>([&](auto M) {
      constexpr int m = decltype(M)::value;
      const double a = v(std::integral_constant<int, 2 * m>{});
      double b = 0.0;
      if constexpr (2 * m + 1 < N) b = v(std::integral_constant<int, 2 * m + 1>{});
      nx[m] = (up ? b : a) + lane_shfl(up ? a : b, lane ^ (1 << BIT));
    });
    return transpose_reduce<W, H, BIT + 1>([&](auto K) { return nx[decltype(K)::value]; });
  }
}

// two doubles written / read as one 16-byte LDS access
struct alignas(16) Pair {
  double a, b;
};

constexpr int STATUS_OPTIMAL = 0;
constexpr int STATUS_MAX_ITER = 1;
constexpr int STATUS_INFEASIBLE = 2;
constexpr int STATUS_NOT_PD = 3;
// internal to the sweep-tableau kernel: the solution it arrived at does not pass its own KKT certificate (the explicitly
// updated inverse lost too much accuracy: weakly regularised objectives, cond(H) >~ 1e8) -- the Goldfarb-Idnani kernel
// solves the instance again in the same launch, callers never see this value
constexpr int STATUS_BREAKDOWN = 4;
// ... or the conditioning estimate after the start-up sweeps says that it would not: the instance goes to the
// Goldfarb-Idnani code before the tableau iteration instead of after it
constexpr int STATUS_ROUTED = 5;
// Which code solved an instance: bits 24.. of iters[b] (include/pinkhip.h, PINKHIP_PATH_*)
constexpr int PATH_TABLEAU = 0, PATH_HANDOVER = 1, PATH_ROUTED = 2, PATH_GI = 3;
constexpr int kPathShift = 24;

// Everything a launch needs; passed by value in the kernarg segment.
struct KernelArgs {
  long long B;
  int nv, Kd, K, md;
  int n_eq;          // the first n_eq dense rows are equalities Gd dq = hd (packed kernel only)
  int n_dtasks;      // diagonal tasks
  int n_barriers;
  int cost_batched;
  int max_iter;
  int lds_pitch;     // doubles of LDS per QP (0: LdsP<NV>::stride(md)); the whole-step kernel may need more for its kinematics
  int rank_deficient;  // host_tables.h rank_deficient_by_construction(): the Goldfarb-Idnani code by dispatch
  double damping, dt;
  double out_scale;  // dq is written times this: 1, or 1 / dt when the caller wants the velocity (pink/solve_ik.py:274)
  // per-instance streams
  const double *J, *e, *cost, *lb, *ub, *Gd, *hd, *c_extra;
  // broadcast tables (device memory, built once per descriptor by the host)
  const double *row_gain;  // [K]
  const double *row_lm;    // [K]
  const int *dtask_col0;   // [n_dtasks] first tangent column
  const int *dtask_row0;   // [n_dtasks] first row in e / cost
  const int *dtask_k;      // [n_dtasks] number of rows
  const int *barrier_rows;        // [n_barriers + 1]
  const double *barrier_safe_gain;  // [n_barriers]
  // outputs
  double *dq;
  int *status;
  int *iters;
  double *H_out, *c_out;  // stack-only kernel
};

}  // namespace pinkhip
