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

// The tableau row of a lane as register TUPLES (16 doubles = one 1024-bit VGPR tuple each) instead of NT independent
// registers: entry p for a wave-uniform RUN-TIME p is then two v_mov_b32 under s_set_gpr_idx_on (the VGPR index mode of
// gfx9: register number + M0) instead of NT broadcast-FMAs against the indicator of p.  Entries with compile-time
// indices stay what they were: sub-registers of the tuple, updated in place by the broadcast-FMAs.
typedef double TabVec16 __attribute__((vector_size(128)));
typedef double TabVec8 __attribute__((vector_size(64)));
template <bool HALF>
struct TabVecSel {
  typedef TabVec16 type;
};
template <>
struct TabVecSel<true> {
  typedef TabVec8 type;
};
template <int NT>
struct TabRegs {
  static constexpr int NVEC = (NT + 15) / 16;
  static_assert(NVEC >= 1 && NVEC <= 4, "at most 64 tableau columns");
  // (the last tuple is a 512-bit one when eight entries or fewer are left for it: NT = 56 = 16 + 16 + 16 + 8 keeps 16
  // registers more than four full tuples would -- the difference between spills inside the tableau loop and none)
  static constexpr bool kHalfLast = NT - 16 * (NVEC - 1) <= 8;
  template <int K>
  using Vec = typename TabVecSel<(K == NVEC - 1) && kHalfLast>::type;
  // named members, whole-tuple reads and writes only: they must stay SSA values
  Vec<0> v0;
  Vec<1> v1;
  Vec<2> v2;
  Vec<3> v3;
  template <int K>
  __device__ __forceinline__ Vec<K> tuple() const {
    if constexpr (K == 0) return v0;
    else if constexpr (K == 1) return v1;
    else if constexpr (K == 2) return v2;
    else return v3;
  }
  template <int J>
  __device__ __forceinline__ double get() const {
    static_assert(J >= 0 && J < NT, "tableau column");
    const Vec<J / 16> t = tuple<J / 16>();
    return t[J % 16];
  }
  template <int J>
  __device__ __forceinline__ void set(double x) {
    static_assert(J >= 0 && J < NT, "tableau column");
    Vec<J / 16> t = tuple<J / 16>();
    t[J % 16] = x;
    if constexpr (J / 16 == 0) v0 = t;
    else if constexpr (J / 16 == 1) v1 = t;
    else if constexpr (J / 16 == 2) v2 = t;
    else v3 = t;
  }
  __device__ __forceinline__ void clear() {
    v0 = Vec<0>{};
    v1 = Vec<1>{};
    v2 = Vec<2>{};
    v3 = Vec<3>{};
  }
  // entry p of this lane's row, p the same in every lane of the wave (0 <= p < NT): every tuple is read at p % 16
  // and a scalar condition selects (a scalar BRANCH around the reads sent the tuples to scratch memory: measured)
  __device__ __forceinline__ double at_uniform(int p) const {
    const int hi = p >> 4;
    // (the element index behind an empty asm: an index the optimiser can prove in range lets it turn "load the tuple,
    // extract element lo" into a scalar load at a run-time address while the tuples still sit in an alloca -- which then
    // never becomes registers: the ik_sweepx.h instantiations ended up reading their tableau from scratch memory)
    const int lo = opaque_uniform(p & 15);
    const int lo8 = kHalfLast ? opaque_uniform(p & 7) : lo;  // (the index into a 512-bit last tuple: opaque AFTER its mask)
    const Vec<0> t0 = v0;
    double r = t0[NVEC == 1 ? lo8 : lo];
    if constexpr (NVEC > 1) {
      const Vec<1> t1 = v1;
      const double r1 = t1[NVEC == 2 ? lo8 : lo];
      r = (hi == 1) ? r1 : r;
    }
    if constexpr (NVEC > 2) {
      const Vec<2> t2 = v2;
      const double r2 = t2[NVEC == 3 ? lo8 : lo];
      r = (hi == 2) ? r2 : r;
    }
    if constexpr (NVEC > 3) {
      const Vec<3> t3 = v3;
      const double r3 = t3[lo8];
      r = (hi == 3) ? r3 : r;
    }
    return r;
  }
  // entry p of this lane's row for a p that is uniform over each group of W lanes (p < 0: none, 0.0)
  template <int W>
  __device__ __forceinline__ double at_group_uniform(int p) const {
    constexpr int G = kWave / W;
    static_assert(G == 1 || G == 2, "one or two reads per wave");
    const int p0 = bcast_i(p, 0);  // v_readlane: a scalar
    double c = at_uniform(p0 < 0 ? 0 : p0);
    if constexpr (G == 2) {
      const int p1 = bcast_i(p, W);
      const double c1 = at_uniform(p1 < 0 ? 0 : p1);
      c = (lane_id() >= W) ? c1 : c;
    }
    return (p >= 0) ? c : 0.0;
  }
};

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
  int n_free_lead;     // pinkhip_desc::n_free_lead (host side: which instantiation; the kernels check the bounds themselves)
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
