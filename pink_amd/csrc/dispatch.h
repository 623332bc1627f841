// Which instantiation of ik_solve_packed_kernel<NV, W, DENSE> serves a problem of tangent dimension nv with md
// dense rows.  Plain C++: shared by the host side of the library (pinkhip.hip), by the per-instantiation
// translation units (tu_packed.hip) and by the CPU wave emulator of the test suite, so that the three can never
// disagree about the rule.
#pragma once

// X(NV, W): NV = nv padded to the next instantiated even size (padded coordinates cost FMAs and LDS traffic),
// W = lanes per QP (64 / W QPs per wavefront).  Every pair is built with and without the dense-row machinery.
#ifdef PINKHIP_DEV_NV  // kernel-development builds: one instantiation only (make DEV=1 [DEVNV=50 DEVW=64 DEVMD=6], ~20 s)
#ifndef PINKHIP_DEV_MD
#define PINKHIP_DEV_MD 0
#endif
#if PINKHIP_DEV_NV > PINKHIP_DEV_W  // (front coordinates eliminated in the tableau kernel: the others hold all of them on 64 lanes)
#define PINKHIP_PACKED_TABLE(X) X(PINKHIP_DEV_NV, 64)
#define PINKHIP_ROLLOUT_TABLE(X) X(PINKHIP_DEV_NV, 64)
#else
#define PINKHIP_PACKED_TABLE(X) X(PINKHIP_DEV_NV, PINKHIP_DEV_W)
#define PINKHIP_ROLLOUT_TABLE(X) X(PINKHIP_DEV_NV, PINKHIP_DEV_W)
#endif
#if PINKHIP_DEV_MD > 0
#define PINKHIP_ROLLOUT_DENSE_TABLE(X) X(PINKHIP_DEV_NV, PINKHIP_DEV_MD, PINKHIP_DEV_W)
#else
#define PINKHIP_ROLLOUT_DENSE_TABLE(X)
#endif
#if PINKHIP_DEV_NV + PINKHIP_DEV_MD > PINKHIP_DEV_W  // (more tableau rows than lanes: the dense rows are virtual / front coordinates eliminated)
#define PINKHIP_SWEEP_TABLE(X) X(PINKHIP_DEV_NV, 0, PINKHIP_DEV_W)
#else
#define PINKHIP_SWEEP_TABLE(X) X(PINKHIP_DEV_NV, PINKHIP_DEV_MD, PINKHIP_DEV_W)
#endif
#if PINKHIP_DEV_MD > 0
#define PINKHIP_SWEEPX_TABLE(X) X(PINKHIP_DEV_NV, PINKHIP_DEV_MD, PINKHIP_DEV_W)
#else
#define PINKHIP_SWEEPX_TABLE(X)
#endif
#else
// X(NV, MD, W): the sweep-tableau kernel with VIRTUAL dense rows ik_solve_sweepx_kernel<NV, MD, W> (ik_sweepx.h): NV
// coordinates on the W lanes, up to MD dense rows riding in a second role of the first MD lanes (NV + MD may exceed W)
#define PINKHIP_SWEEPX_TABLE(X) X(16, 8, 16) X(30, 6, 32) X(30, 8, 32) X(32, 8, 32)
// X(NV, MD, W): the sweep-tableau kernel ik_solve_sweep_kernel<NV, MD, W> (ik_sweep.h): NV coordinates + MD dense rows
// = NT <= W tableau rows, one per lane -- or, box-only with NV > W: the first NV - W coordinates are eliminated before the
// solve (coordinates without bounds in every instance, pinkhip_desc::n_free_lead: the root of a free-flyer) and the other W
// ride on the lanes: X(34, 0, 32) = nv 33 / 34 two QPs per wavefront.  Ordered by NT within box-only / with dense rows; problems that fit none
// (8-lane groups, more dense rows than lanes are left) run the Goldfarb-Idnani kernel of PINKHIP_PACKED_TABLE.
#define PINKHIP_SWEEP_TABLE(X)                                                                                      \
  X(8, 0, 16) X(12, 0, 16) X(16, 0, 16) X(24, 0, 32) X(30, 0, 32) X(32, 0, 32) X(34, 0, 32) X(34, 0, 64) X(40, 0, 64) X(48, 0, 64) X(50, 0, 64) \
  X(56, 0, 64) X(64, 0, 64)                                                                                         \
  X(12, 4, 16) X(24, 8, 32) X(30, 2, 32) X(30, 8, 64) X(34, 8, 64) X(40, 8, 64) X(50, 6, 64) X(50, 14, 64) X(56, 8, 64)
// the whole-control-step kernel exists for the groups of whole 16-lane rows (broadcast-FMA stacking), box limits only
#define PINKHIP_ROLLOUT_TABLE(X) X(12, 16) X(16, 16) X(24, 32) X(30, 32) X(32, 32) X(34, 64) X(40, 64) X(48, 64) X(50, 64) X(56, 64)
// ... and, with position-barrier rows formed on chip (X(NV, MD, W): NV + MD tableau rows on W lanes), for these
// (NV + MD > W: virtual dense rows, ik_sweepx.h; listed ahead of the wider group that would also hold the robot)
#define PINKHIP_ROLLOUT_DENSE_TABLE(X) X(12, 4, 16) X(30, 6, 32) X(30, 8, 64) X(34, 8, 64) X(50, 6, 64) X(50, 14, 64) X(56, 8, 64)
#define PINKHIP_PACKED_TABLE(X)                                                                          \
  X(6, 8) X(8, 8) X(12, 16) X(16, 16) X(24, 32) X(30, 32) X(32, 32) X(34, 64) X(40, 64) X(48, 64) X(50, 64) X(56, 64) X(64, 64)
#endif

namespace pinkhip {

struct PackedChoice {
  int NV, W;
};

// Smallest instantiation that holds the problem.  Lane li < md of a group owns dense row li, so a group of W
// lanes takes at most W dense rows: up to PINKHIP_MAX_MD = 64 in the 64-lane instantiations (more rows than the
// smallest group for nv has lanes move the problem to a wider group).
inline PackedChoice select_packed(int nv, int md) {
#define PINKHIP_PICK(NV_, W_) \
  if (nv <= NV_ && md <= W_) return PackedChoice{NV_, W_};
  PINKHIP_PACKED_TABLE(PINKHIP_PICK)
#undef PINKHIP_PICK
  return PackedChoice{0, 0};
}

struct SweepChoice {
  int NV, MD, W;
};

// Smallest sweep-tableau instantiation that holds nv coordinates and md dense rows ({0, 0, 0}: none).
// n_free_lead: how many leading coordinates carry no bound in any instance (an instantiation that eliminates NV - W front
// coordinates needs that many)
inline SweepChoice select_sweep(int nv, int md, int n_free_lead = 0) {
#define PINKHIP_PICK(NV_, MD_, W_) \
  if (nv <= NV_ && md <= MD_ && (md > 0) == (MD_ > 0) && (NV_ <= W_ || n_free_lead >= NV_ - W_)) return SweepChoice{NV_, MD_, W_};
  PINKHIP_SWEEP_TABLE(PINKHIP_PICK)
#undef PINKHIP_PICK
  return SweepChoice{0, 0, 0};
}

// Smallest instantiation with virtual dense rows that holds nv coordinates and md > 0 dense rows ({0, 0, 0}: none).
inline SweepChoice select_sweepx(int nv, int md) {
  if (md <= 0) return SweepChoice{0, 0, 0};
#define PINKHIP_PICK(NV_, MD_, W_) \
  if (nv <= NV_ && md <= MD_) return SweepChoice{NV_, MD_, W_};
  PINKHIP_SWEEPX_TABLE(PINKHIP_PICK)
#undef PINKHIP_PICK
  return SweepChoice{0, 0, 0};
}

// ... and whether it is the kernel to run: when it packs more QPs into a wavefront than the instantiation with one
// lane per tableau row (nv = 30 with three to eight dense rows: two QPs per wavefront against one) -- or that one
// does not exist.
inline bool prefer_sweepx(int nv, int md) {
  const SweepChoice x = select_sweepx(nv, md);
  if (!x.NV) return false;
  const SweepChoice s = select_sweep(nv, md);
  return !s.NV || x.W < s.W;
}

// Which of the two stack + solve kernels serves a batch of B problems (measured on MI355X, scripts/ab_solvers.sh):
// the sweep-tableau kernel wherever it is instantiated, except
//   * when it needs a wider group than the Goldfarb-Idnani kernel (nv = 30 with six dense rows: 36 tableau rows = one QP
//     per wavefront against two: 1.75 against 1.46 ms per 65 536), and
//   * for nv <= 8 in large batches: its smallest group is 16 lanes, the Goldfarb-Idnani kernel packs eight QPs per
//     wavefront (UR5: 71 against 49 us at B = 65 536, but 13.8 against 18.4 us at B = 4 096, where eight per wavefront
//     leave half of the 1 024 SIMDs without a wave).
inline bool prefer_sweep(int nv, int md, long long B, int n_free_lead = 0) {
  const SweepChoice sc = select_sweep(nv, md, n_free_lead);
  if (!sc.NV) return false;
  const PackedChoice pc = select_packed(nv, md);
  if (!pc.NV) return true;
  if (nv <= 8) return B <= 16384;
  return sc.W <= pc.W;
}

// Doubles of LDS per QP of the sweep-tableau kernel (= SweepLds<NV, MD, W>::stride, checked at compile time in
// tu_sweep.hip): H packed, c, the columns of G.
constexpr int sweep_lds_doubles(int NV, int MD, int W) { return ((NV * (NV + 1) / 2 + 1) & ~1) + 2 * W + MD * (W + 2); }

// ... of the kernel with virtual dense rows (= SweepXLds<NV, MD, W>::stride, checked in tu_sweepx.hip)
constexpr int sweepx_lds_doubles(int NV, int MD, int W) { return ((NV * (NV + 1) / 2 + 1) & ~1) + W + MD * (W + 2) + 2 * ((MD + 1) & ~1); }

// Doubles of LDS per QP of the Goldfarb-Idnani kernel (= LdsP<NV>::stride(md), checked at compile time in
// tu_rollout.hip): the sweep-tableau kernels hand a group over to it when its result fails the certificate.
constexpr int packed_lds_doubles(int NV, int md) {
  return ((((NV * (NV + 3) / 2 + 1) & ~1) + 5 * NV + md * (NV + 1)) + 1) & ~1;
}
constexpr int max3(int a, int b, int c) { return (a > b ? a : b) > c ? (a > b ? a : b) : c; }

// World positions of the nf task frames, kept BEHIND the area the kinematics and the two solvers share: the rows of
// position barriers are formed from them while the Goldfarb-Idnani code (hand-over) is already writing its staged rows.
// ... and, for up to kRolloutMaxEqFrames equality constraints made of frame tasks, U / V of their frames and their six errors
// (24 doubles each): the Goldfarb-Idnani code forms those rows too while the shared area is being overwritten.
constexpr int kRolloutMaxEqFrames = 2;
// (sized by the constraints the call has: a barrier-only stack does not pay LDS for them)
constexpr int rollout_tail_doubles(int nf, int n_eqf) { return ((3 * nf + 1) & ~1) + 24 * n_eqf; }

// Doubles of LDS per robot of the whole-control-step kernel: its kinematics scratch (fk_doubles) shares the solve's
// LDS, whichever is larger (+ the frame positions behind it when dense rows are formed on chip).
constexpr int rollout_lds_doubles(int NV, int W, int fk_doubles, int MD = 0, int nf = 0, int n_eqf = 0) {
  return max3((fk_doubles + 1) & ~1, NV + MD > W ? sweepx_lds_doubles(NV, MD, W) : sweep_lds_doubles(NV, MD, W), packed_lds_doubles(NV, MD)) +
         (MD > 0 ? rollout_tail_doubles(nf, n_eqf) : 0);
}

// Instantiation of the whole-control-step kernel for a robot with nv tangent coordinates and nj joints whose
// kinematics scratch needs fk_doubles doubles of LDS: W lanes must hold a joint / a column each, and the 64 / W
// robots of a wavefront must fit the 64 KiB of LDS a workgroup may ask for.
// ... with md > 0 rows of position barriers: {NV, MD, W} from PINKHIP_ROLLOUT_DENSE_TABLE
inline SweepChoice select_rollout_dense(int nv, int nj, int fk_doubles, int md, int nf, int n_eqf) {
#define PINKHIP_PICK(NV_, MD_, W_)                                                                          \
  if (nv <= NV_ && md <= MD_ && nj <= W_) {                                                                 \
    const int need = rollout_lds_doubles(NV_, W_, fk_doubles, MD_, nf, n_eqf);                                \
    if (8 * need * (64 / W_) + 16 <= 65536) return SweepChoice{NV_, MD_, W_};                               \
  }
  PINKHIP_ROLLOUT_DENSE_TABLE(PINKHIP_PICK)
#undef PINKHIP_PICK
  return SweepChoice{0, 0, 0};
}

inline PackedChoice select_rollout(int nv, int nj, int fk_doubles, bool needed = false) {
  // robots that fit an 8-lane group keep the two-launch step: padding them to 16 lanes halves the robots per
  // wavefront (measured, 6-dof arm: 0.107 ms in one kernel at NV = 12 against 0.068 ms in two launches at NV = 6) --
  // unless the task stack has rows only this kernel forms (`needed`: constant rows, extra identity tasks, relative slots)
  if (nv <= 8 && !needed) return PackedChoice{0, 0};
#define PINKHIP_PICK(NV_, W_)                                                                              \
  if (nv <= NV_ && nj <= W_)                                                                               \
    return 8 * rollout_lds_doubles(NV_, W_, fk_doubles) * (64 / W_) + 16 <= 65536 ? PackedChoice{NV_, W_} : PackedChoice{0, 0};
  PINKHIP_ROLLOUT_TABLE(PINKHIP_PICK)
#undef PINKHIP_PICK
  return PackedChoice{0, 0};
}

}  // namespace pinkhip
