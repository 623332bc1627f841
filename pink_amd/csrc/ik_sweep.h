// Stack + solve on a register-resident *sweep tableau*: W lanes per QP, 64/W QPs per wavefront, NO LDS.
//
// What it computes is the reference's pink/solve_ik.py:206-275 per instance -- H, c stacked as pink/tasks/task.py:145-167
// and solve_ik.py:54-67 prescribe, then the QP  min 1/2 dq^T H dq + c^T dq  s.t.  lb <= dq <= ub,  Gd dq <= hd  (leading
// n_eq rows: =) that qpsolvers / quadprog solves at solve_ik.py:270 -- with the same dual active-set logic as
// Goldfarb-Idnani (start at the unconstrained minimum, add the violated constraint, drop the constraints whose
// multiplier reaches zero on the way), but on a different representation of the working set.
//
// ik_kernels_packed.h keeps J = L^-T Q and the triangular factor of the active normals (quadprog's own data
// structures): every step costs two matrix-vector products, a Householder update of J, a row of J extracted through LDS,
// a product with R^-1 read from LDS and, for a drop, a sweep of Givens rotations.  Here the state is ONE symmetric
// matrix, the KKT matrix  K = [H G^T; G 0]  swept (Beaton's sweep operator = a Gauss-Jordan pivot that keeps the
// symmetry) on the set B of *basic* indices -- the free coordinates and the active dense rows:
//     T_BB = -K_BB^-1,   T_NB = K_NB K_BB^-1,   T_NN = K_NN - K_NB K_BB^-1 K_BN.
// Lane li of a group owns row li of T in NT = NV + MD registers (coordinates 0..NV-1, dense rows NV..NV+MD-1).
//   * fixing a coordinate at a bound (= adding a box constraint), freeing it (= dropping it), activating or dropping a
//     dense row are all the same operation: one pivot of T on that index, a rank-one update
//         T[m][j] -= (T[m][p] / T[p][p]) T[p][j],
//     i.e. NT broadcast-FMAs per lane (the row T[p][.] = column T[.][p] is a lane-held vector);
//   * column p of T holds everything a step needs: how the free coordinates move (T_Bp), how the multipliers of the
//     fixed coordinates and active rows change (T_Np resp. T_Bp), how the slacks of the inactive rows change, and on
//     the diagonal the curvature n^T Z n along the entering normal (the exact reduced steepest-edge weight of the
//     entering rule and the step length).  The column is extracted with NT broadcast-FMAs against an indicator vector.
// A step is therefore 2 NT FMAs (3 NT when a constraint is dropped) plus two group reductions, against ~4 NT FMAs, ~60
// LDS accesses and five reductions in the Goldfarb-Idnani kernel, and the kernel needs NT + ~20 doubles of registers
// per lane instead of 2 NV + ~20 and no LDS: more waves per SIMD.  Measured: DESIGN.md section 3.1.
//
// The minimiser is the same (strictly convex QP); the path through the active sets follows the same rule as
// ik_kernels_packed.h (violation weighted by 1 / sqrt(n^T Z n), here with the exact reduced Z).
#pragma once

#include "ik_common.h"
#include "ik_kernels_packed.h"
#include "ik_stack_rows.h"

#ifdef PINKHIP_SECTION_CLOCK
#ifndef PINKHIP_TICK
static __device__ unsigned long long pinkhip_clock[16];  // one copy per translation unit
#define PINKHIP_TICK(k)                                                        \
  do {                                                                         \
    if (clock_on) {                                                            \
      const unsigned long long now_ = __builtin_readcyclecounter();            \
      if (lane == 0) atomicAdd(&pinkhip_clock[k], now_ - clock_prev);          \
      clock_prev = __builtin_readcyclecounter();                               \
    }                                                                          \
  } while (0)
#endif
#else
#ifndef PINKHIP_TICK
#define PINKHIP_TICK(k)
#endif
#endif

// dependent FMA chains of the column extraction (two: what three or four waves per SIMD hide)
#ifndef PINKHIP_SWEEP_COLUMN_CHAINS
#define PINKHIP_SWEEP_COLUMN_CHAINS(NT) 2
#endif

// relative size of a gradient entry the KKT certificate of the closing trip accepts (a refined point leaves ~1e-15).
// 1e-12 against 1e-11: free on the headline, +0.7 % on the JVRC shape, and three of the four draws in 20 000 weakly
// regularised ones whose dq was off by 1e-3 .. 3e-2 behind a passed certificate take the hand-over (1e-13: +3 %).
#ifndef PINKHIP_SWEEP_CERT_TOL
#define PINKHIP_SWEEP_CERT_TOL 1e-12
#endif
// conditioning estimate max_i H_ii (H^-1)_ii beyond which an instance skips the tableau iteration
// chunks of eight task rows requested ahead of the one being accumulated (ik_stack_rows.h)
#ifndef PINKHIP_STACK_DEPTH
#define PINKHIP_STACK_DEPTH 1
#endif
#ifndef PINKHIP_STACK_DEPTH_WIDE  // (instantiations at two waves per SIMD: 256 registers)
#define PINKHIP_STACK_DEPTH_WIDE 2
#endif
// Column p of the tableau (group-uniform run-time p) read with the VGPR index mode from register tuples (TabRegs,
// ik_common.h) instead of NT broadcast-FMAs against the indicator of p: groups of 32 and 64 lanes (one or two reads per
// wave; four groups of 16 lanes would pay as much as the FMAs they replace)
#ifndef PINKHIP_SWEEP_INDEXED_COLUMN
#define PINKHIP_SWEEP_INDEXED_COLUMN 1
#endif
// (1e8 since round 6 -- was 1e10: the wide fuzz of round 6 found three weakly regularised draws in 100 000, estimates 3e8 ..
// 2e9, whose certified tableau point was 2e-5 .. 8e-5 from the exact minimiser where the Goldfarb-Idnani code is within
// 4e-9: the certificate cannot see an error along a direction in which the residual is below its own round-off, and an
// explicitly updated inverse loses up to cond^2 eps.  Pink's stacks with a posture task sit at 1e3 .. 1e5.)
#ifndef PINKHIP_SWEEP_ROUTE_COND
#define PINKHIP_SWEEP_ROUTE_COND 1e8
#endif
// Box-only instantiations (MD = 0): single principal pivoting instead of the Goldfarb-Idnani trips (see "principal
// pivoting" below).  0 restores round 5's loop (A/B: profiles/ab_ppm_r06.txt)
#ifndef PINKHIP_SWEEP_PPM
#define PINKHIP_SWEEP_PPM 1
#endif
// trips after which the principal pivoting switches from the largest-infeasibility rule to Murty's least-index rule
// (finite for the P-matrices strictly convex box QPs give: Murty 1974; Judice & Pires 1994 with upper bounds)
// ... started from a guessed active set instead of the unconstrained minimum: 0 = every coordinate free at the start
#ifndef PINKHIP_SWEEP_PPM_CRASH
#define PINKHIP_SWEEP_PPM_CRASH 1
#endif
// ... and with dense rows: principal pivoting as long as every exchange is regular, the dual method behind it
// ... and in the whole-step kernels with dense rows: off.  Started all-free (below) principal pivoting takes the dual method's
// number of trips there (23.10 against 23.14 after a target move at nv = 50 + 6 rows) and the mode per group costs registers:
// 38 spilled against 15, the converged step 1.73 ms against 1.63 (profiles/ab_ppm_r06.txt)
#ifndef PINKHIP_ROLLOUT_PPM_DENSE
#define PINKHIP_ROLLOUT_PPM_DENSE 0
#endif
#ifndef PINKHIP_ROLLOUT_PPM_VIRTUAL  // (ik_sweepx.h inside the whole-step kernel: 0.665 against 0.655 ms at nv = 30 + 6 rows)
#define PINKHIP_ROLLOUT_PPM_VIRTUAL 0
#endif
#ifndef PINKHIP_SWEEP_PPM_DENSE
#define PINKHIP_SWEEP_PPM_DENSE 1
#endif
// development: why a group asked for the hand-over, in the thousands of iters[] (read with PINKHIP_SWEEP_NO_HANDOVER)
#ifdef PINKHIP_SWEEP_DEBUG_WHY
#define PINKHIP_WHY(k) why = (k) + 10 * status_before
#else
#define PINKHIP_WHY(k)
#endif
// Curvature left along the normal of a coordinate that is fixed / a row that is activated, relative to (a lower bound of) the
// unreduced one, below which principal pivoting does not pivot: the group goes on with the dual method, which forms a small
// curvature again as a sum of squares and knows what to do about a dependent normal.  An exchange on 5.6e-8 of it (one
// instance in 65 536 of the nv = 30 + 6 rows batch) multiplied the tableau's round-off by its reciprocal: nu = 1e10, a
// certificate that failed by 3e-2, the hand-over.
#ifndef PINKHIP_SWEEP_PPM_MIN_CURV
#define PINKHIP_SWEEP_PPM_MIN_CURV 1e-5
#endif
#ifndef PINKHIP_SWEEP_PPM_MURTY_AFTER
#define PINKHIP_SWEEP_PPM_MURTY_AFTER(nv) (4 * (nv) + 20)
#endif

namespace pinkhip {

// LDS of one QP (doubles): the stated problem, parked for the closing refinement step
template <int NV, int MD, int W>
struct SweepLds {
  static __host__ __device__ constexpr int tri(int i) { return i * (i + 1) / 2; }  // H[i][0..i]
  static constexpr int oC = (NV * (NV + 1) / 2 + 1) & ~1;                            // c, one entry per lane
  // G[d][li] at d GP + li.  The pitch is W + 2, not W: the lane of dense row d reads its row G[d][.] while the products
  // with the stated problem run (krow_times), all dense lanes at the same column -- at a pitch of W = 64 doubles every
  // one of those reads fell into the same LDS bank (profiles/sq_counters_jvrc_r03.txt: 7.1 M address conflicts)
  static constexpr int GP = W + 2;
  static constexpr int oG = oC + W;
  static constexpr int oR = oG + MD * GP;                                            // one row of T in transit (W entries)
  static constexpr int stride = oR + W;
};

// NVT > W (box-only): the batch has NVT coordinates of which the FIRST NE = NVT - W are eliminated before the solve --
// coordinates without bounds (the leading ones of a free-flyer root: pink/limits/configuration_limit.py:50-71 never selects
// them, a VelocityLimit may not bound them) are always free, their rows of the KKT system can be taken out once and for
// all: H' = H_rr - H_re H_ee^-1 H_er, c' = c_r - H_re H_ee^-1 c_e on the remaining W coordinates, x_e = -H_ee^-1 (c_e +
// H_er x_r) afterwards.  That is what sweeping them in first does, except that their rows and columns do not ride through
// the iteration: nv = 33 / 34 (examples/humanoid_draco3.py:55-56: Draco3 with its free-flyer) is solved two QPs per
// wavefront on 32-lane groups instead of one on 64 lanes.  An instance whose leading coordinates DO carry a bound (the
// host side only picks this instantiation when the caller says they do not: pinkhip_desc::n_free_lead) goes to the
// Goldfarb-Idnani kernel.
template <int NVT, int MD, int W, class Src = HbmTerms>
__device__ __forceinline__ int ik_sweep_instance(const KernelArgs &a, long long block, Src *terms = nullptr) {
  constexpr int NE = NVT > W ? NVT - W : 0;  // eliminated front coordinates
  constexpr int NV = NVT - NE;               // coordinates of the tableau
  static_assert(NE == 0 || (NE <= 2 && MD == 0 && !Src::kOnTheFly), "front coordinates are eliminated in the box-only stack + solve kernel");
  constexpr int NT = NV + MD;
  static_assert((W == 16 || W == 32 || W == 64) && NT <= W && NV % 2 == 0 && MD >= 0, "group of whole rows of 16 lanes");
  constexpr bool DENSE = MD > 0;
  constexpr bool PPM = !DENSE && PINKHIP_SWEEP_PPM;                                  // box-only: the whole iteration
  constexpr bool PPMD = DENSE && PINKHIP_SWEEP_PPM && PINKHIP_SWEEP_PPM_DENSE && (!Src::kOnTheFly || PINKHIP_ROLLOUT_PPM_DENSE);  // dense rows: in front of the dual method
  constexpr bool PPX = PPM || PPMD;
  constexpr int G = kWave / W;
  constexpr double INF = INFINITY;
  constexpr double BIG = 1e300;
  using BcT = Bcast<W>;

  const int lane = lane_id();
  const int g = lane / W, li = lane & (W - 1);
  const int nvs = a.nv;      // coordinates per instance in the streams (row pitch of J, lb, ub, dq)
  const int nv = nvs - NE;   // ... of the tableau
  const int md = DENSE ? a.md : 0, n_eq = DENSE ? a.n_eq : 0;
#ifdef PINKHIP_SECTION_CLOCK
  const bool clock_on = (block & 63) == 0;
  unsigned long long clock_prev = __builtin_readcyclecounter();
#endif
  long long b = block * G + g;
  const bool valid = b < a.B;
  if (!valid) b = a.B - 1;  // surplus groups of the last wave redo the last instance, write nothing

  const bool in = li < nv;          // coordinate lane
  const int dr = li - NV;           // dense row of this lane
  const bool dlane = DENSE && dr >= 0 && dr < md;

  // ------------------------------------------------------------------ stack (task.py:145-167, solve_ik.py:54-67)
  double T[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) T[j] = 0.0;
  double ci = 0.0, mu_l = 0.0, dadd = 0.0;
  // Everything this instance reads from HBM is requested before the first row is accumulated (PINKHIP_STACK_DEPTH
  // chunks of eight rows ahead = the headline's 24 rows; bounds, c_extra and the terms of the diagonal tasks behind
  // them): one memory latency per wave instead of one per chunk and two more for the diagonal tasks and the bounds.
  double ci_d = 0.0, mu_d = 0.0, lbv = -INF, ubv = INF, hv = 0.0, ginv = 1.0;
  // eliminated front coordinates: what the diagonal tasks add to their diagonal and to c, whether they are free
  double front_dadd[NE > 0 ? NE : 1] = {}, front_c[NE > 0 ? NE : 1] = {};
  bool front_free = true;
  auto others = [&]() {
    if (in) {
      if constexpr (!Src::kOnTheFly) {
        lbv = a.lb[b * (long long)nvs + NE + li];
        ubv = a.ub[b * (long long)nvs + NE + li];
      }
      dadd = stack_diag_tasks<Src>(a, b, terms, NE + li, ci_d, mu_d);
      if (a.c_extra) ci_d += a.c_extra[b * (long long)nvs + NE + li];
    }
    if constexpr (NE > 0) {
#pragma unroll
      for (int x = 0; x < NE; ++x) {
        double mu_x = 0.0;
        front_dadd[x] = stack_diag_tasks<Src>(a, b, terms, x, front_c[x], mu_x);
        if (a.c_extra) front_c[x] += a.c_extra[b * (long long)nvs + x];
        if (li == x) mu_d += mu_x;  // (the Levenberg-Marquardt term is a sum over the group: once)
        front_free = front_free && a.lb[b * (long long)nvs + x] == -INF && a.ub[b * (long long)nvs + x] == INF;
      }
    }
    if constexpr (DENSE && !Src::kOnTheFly) {
      // K = [H G^T; G 0]: coordinate lane li takes G[d][li] into column NV + d (the stacking leaves those alone)
      if (md > 0) {
        const double *Gb = a.Gd + b * (long long)md * nv;
        static_for<0, MD>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          T[NV + d] = (in && d < md) ? Gb[(long long)d * nv + li] : 0.0;
        });
        if (dlane) hv = a.hd[b * (long long)md + dr];
      }
    }
  };
  StackFront<NE> front;
  if constexpr (NE > 0) {
    KernelArgs af = a;  // (the rows as the lanes see them: NE columns in)
    af.J = a.J + NE;
    stack_rows_bcast<NV, W, 8, Src, PINKHIP_STACK_DEPTH, NE>(af, b, terms, in, li, T, ci, mu_l, others, &front);
  } else {
    stack_rows_bcast<NV, W, 8, Src, (NT > 34 ? PINKHIP_STACK_DEPTH_WIDE : PINKHIP_STACK_DEPTH)>(a, b, terms, in, li, T, ci, mu_l, others);
  }
  ci += ci_d;
  mu_l += mu_d;
  double diag = a.damping + group_sum<W>(mu_l);

  if constexpr (DENSE) {
    if (md > 0) {
      // (the lane of dense row d takes its row)
      const double *Gb = Src::kOnTheFly ? nullptr : a.Gd + b * (long long)md * nv;
      if constexpr (Src::kOnTheFly) {
        static_for<0, MD>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          T[NV + d] = (in && d < md) ? terms->dense_col(d) : 0.0;
        });
      }
      if constexpr (Src::kOnTheFly) {
        // the rows of G exist only as column entries in the coordinate lanes: the lane of row d collects G[d][j] from
        // lane j (K is symmetric)
        static_for<0, MD>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          if (d < md) {  // wave-uniform
            const BcT gb = bcast_prepare<W>(T[NV + d]);
            static_for<0, NV>([&](auto Jc) {
              constexpr int j = decltype(Jc)::value;
              const double v = value_bcast<W, j>(gb);
              if (dr == d) T[j] = v;
            });
          }
        });
      }
      if (dlane) {
        double n2 = 0.0;
        if constexpr (!Src::kOnTheFly) {
          const double *gr = Gb + (long long)dr * nv;
#pragma unroll
          for (int j = 0; j < NV; ++j) T[j] = (j < nv) ? gr[j] : 0.0;
        } else {
          hv = terms->dense_h(dr);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) n2 += T[j] * T[j];
        ginv = (n2 > 0.0) ? 1.0 / sqrt(n2) : 1.0;
      }
      // barrier objective (barrier.py:193-200): rho_b = r_b / ||J_h||_F^2 on the diagonal, J_h = -dt G rows
      for (int t = 0; t < a.n_barriers; ++t) {
        const double r = a.barrier_safe_gain[t];
        if (r > 1e-6) {
          const int r0 = a.barrier_rows[t], r1 = a.barrier_rows[t + 1];
          double s = 0.0;
          static_for<0, MD>([&](auto Dc) {
            constexpr int d = decltype(Dc)::value;
            if (in && d >= r0 && d < r1) s += T[NV + d] * T[NV + d];
          });
          s = group_sum<W>(s);
          diag += r / (s * a.dt * a.dt);
        }
      }
    }
  }
  const double diag_u = diag;  // (the part every coordinate gets: damping + the Levenberg-Marquardt terms + barrier objective)
  diag += dadd;  // (per lane from here on: what H[li][li] holds beyond the dense task rows; kept for the refinement)
  double hii = 1.0;  // H[li][li]
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (j == li) {
      T[j] += in ? diag : 1.0;  // padded coordinates: identity rows, never pivoted
      hii = T[j];
    }
  using SL = SweepLds<NV, MD, W>;
  double *sm = shared_base() + (long long)g * (a.lds_pitch ? a.lds_pitch : SL::stride);
  // ------------------------------------------------------------------ front coordinates out of the problem
  // One Gaussian elimination step per front coordinate on the stated problem: H -= h_e h_e^T / H_ee, c -= h_e c_e / H_ee
  // (the second one on what the first left).  What the recovery x_e = -(c_e + h_e . x_r + H_ee' x_e') / H_ee needs is
  // parked behind the tableau's own LDS area: h_e per lane, then c_e, 1 / H_ee and the coupling of the two.
  bool front_bad = false;
  if constexpr (NE > 0) {
    double h00 = group_sum<W>(front.hxx[0]) + diag_u + front_dadd[0];
    double c0 = group_sum<W>(front.cx[0]) + front_c[0];
    double h01 = 0.0, h11 = 1.0, c1 = 0.0;
    if constexpr (NE == 2) {
      h01 = group_sum<W>(front.hxx[1]);
      h11 = group_sum<W>(front.hxx[2]) + diag_u + front_dadd[1];
      c1 = group_sum<W>(front.cx[1]) + front_c[1];
    }
    const double m0 = in ? front.m[0] : 0.0;
    double m1 = (NE == 2 && in) ? front.m[NE - 1] : 0.0;
    front_bad = !(h00 > 0.0);
    const double r0 = fast_rcp(h00);
    {
      const BcT hb = bcast_prepare<W>(m0);
      const double f = -m0 * r0;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        T[j] = fma_bcast<W, j>(T[j], hb, f);
      });
      ci = fma(f, c0, ci);
      hii = fma(f, m0, hii);
      if constexpr (NE == 2) {
        m1 = fma(f, h01, m1);
        h11 = fma(-h01 * r0, h01, h11);
        c1 = fma(-h01 * r0, c0, c1);
      }
    }
    double r1 = 0.0;
    if constexpr (NE == 2) {
      front_bad = front_bad || !(h11 > 0.0);
      r1 = fast_rcp(h11);
      const BcT hb = bcast_prepare<W>(m1);
      const double f = -m1 * r1;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        T[j] = fma_bcast<W, j>(T[j], hb, f);
      });
      ci = fma(f, c1, ci);
      hii = fma(f, m1, hii);
    }
    wave_sync();
    sm[SL::stride + li] = m0;
    sm[SL::stride + W + li] = m1;
    if (li == 0) {
      double *fs = sm + SL::stride + 2 * W;
      fs[0] = c0, fs[1] = r0, fs[2] = h01, fs[3] = c1, fs[4] = r1;
    }
  }
  // The QP as stated -- H (lower triangle, packed), c and the columns of G -- is needed once more, by the refinement
  // step that closes the iteration: parked in LDS (the kernel's only use of it; in the whole-step kernel the region
  // overlays the kinematics scratch, which is dead by now), not carried through the loop in registers.
  wave_sync();  // (whole-step kernel: every lane is done reading the kinematics scratch)
  if (li < NV) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j <= li) sm[SL::tri(li) + j] = T[j];
  }
  sm[SL::oC + li] = ci;
  if constexpr (DENSE) {
    static_for<0, MD>([&](auto Dc) {
      constexpr int d = decltype(Dc)::value;
      sm[SL::oG + d * SL::GP + li] = (li < NV) ? T[NV + d] : 0.0;
    });
  }
  PINKHIP_TICK(0);  // stacking

  // ------------------------------------------------------------------ sweep in every coordinate: T = [-H^-1 ...]
  // Pivot k of the unswept matrix is the Schur complement of H on coordinates 0..k-1 (the square of Cholesky's
  // pivot): not positive = H is not positive definite (quadprog's "matrix G is not positive definite").
  // Lane k's own row becomes col / p; written as T[k][j] - (1 - 1/p) col_j it is the same FMA as every other lane's
  // (T[k][j] and col_j = T[j][k] agree to round-off: the difference enters like a perturbation of H of that size).
  // (the smallest pivot decides afterwards: a group that met a non-positive one sweeps on through whatever that leaves
  // -- infinities, NaN -- and never iterates on it; one v_min per column instead of a compare and three selects)
  int status = STATUS_OPTIMAL;
  int why = 0, status_before = 0;
  if constexpr (NE > 0) {
    // (a front coordinate that carries a bound after all: the Goldfarb-Idnani kernel solves the instance as it is stated)
    if (!front_free) status = STATUS_ROUTED;
    else if (front_bad) status = STATUS_NOT_PD;
  }
  double pmin = INF;
  // per lane: state 0 = free coordinate / inactive row, 1 = fixed at lb / active row, 2 = fixed at ub
  int state = 0;
  if constexpr (Src::kOnTheFly) {
    lbv = in ? terms->lb : -INF;
    ubv = in ? terms->ub : INF;
  }
  // The start is a guessed active set -- in the stack + solve kernels.  The whole-step kernel starts with every coordinate
  // free: the Jacobians of a serial chain are far from the diagonal picture the guess is made from, and on closed loops
  // the guess cost more trips than it saved (7 % of the robots of a converged batch took 20-30 trips to repair a guess
  // where 1-10 trips from the unconstrained minimum do: profiles/ab_ppm_r06.txt, scripts/gpu_rollout_iters.py); a
  // controller that tracks its targets has next to nothing to guess anyway.
  constexpr bool GUESS = PPX && PINKHIP_SWEEP_PPM_CRASH && !Src::kOnTheFly;
  if constexpr (GUESS) {
    // ---------------------------------------------------------------- principal pivoting: where it starts
    // Principal pivoting needs no feasibility of any kind from its starting basis, so it does not have to be the
    // unconstrained minimum (every coordinate swept in: NV sweeps, and then one pivot for every bound that ends up
    // active).  The start is a guess of the active set instead: coordinate i is fixed at the bound that
    // x_i = -c_i / H_ii violates (the minimiser if H were its diagonal), everything else is free, and only the free
    // coordinates are swept in.  A wrong guess costs the pivots that repair it, nothing else; on the saturated
    // headline batch the guess fixes 19 of 24 bounded coordinates (19 are active at the minimiser) and the iteration
    // takes 13 pivots instead of 26 behind 11 sweeps instead of 30; a batch whose minimisers are interior starts
    // where it used to (scripts/multi_pivot_study.py).
    // (a non-positive diagonal entry: H is not positive definite whatever the rest looks like)
    if (group_first_lane<W>(in && !(hii > 0.0)) < W) status = STATUS_NOT_PD;
    const double xd = -ci * approx_rcp(hii);
    if (in) state = (xd < lbv) ? 1 : ((xd > ubv) ? 2 : 0);
    const unsigned long long fm = wave_ballot(li < NV && state == 0 && in);
    // this lane's group (W = 64: the wave, a scalar; below: 32 bits of a register)
    using FreeMask = typename std::conditional<W == 64, unsigned long long, unsigned>::type;
    FreeMask gfree = static_cast<FreeMask>(fm);
    if constexpr (W == 32) gfree = (lane >= 32) ? static_cast<unsigned>(fm >> 32) : static_cast<unsigned>(fm);
    else if constexpr (W == 16) gfree = static_cast<unsigned>(fm >> (lane & 48)) & 0xFFFFu;
    static_for<0, NV>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      {
        // (the lanes of a group that does not sweep coordinate k in are switched off for the sweep -- the broadcast
        // operands of a row come from its own group, which is on or off as a whole -- instead of carrying these selects:
        // wave.h, lanes_on; no group that does: the one branch skips the sweep)
        const bool want = ((gfree >> k) & 1) != 0;
        if (lanes_on(want)) {
          BcT xb = bcast_prepare<W>(T[k]);
          const double p = value_bcast<W, k>(xb);
          pmin = min_raw(pmin, want ? p : INF);
          const double rp = fast_rcp(p);
          const double t = T[k] * rp;
          double nt = (li == k) ? rp - 1.0 : -t;
          if (!want) nt = 0.0;
          double tk = (li == k) ? -rp : t;
#if defined(__HIP_DEVICE_COMPILE__)
          // (everything the sweep needs besides the row is in registers before the row is touched: the allocator otherwise
          // rotated the whole row through its registers around each sweep of the whole-step kernel -- 30 moves per sweep)
          asm volatile("" : "+v"(nt), "+v"(xb.r[0]), "+v"(tk));
#endif
          static_for<0, NT>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            if constexpr (j != k) T[j] = fma_bcast<W, j>(T[j], xb, nt);
          });
          if (want) T[k] = tk;
        }
      }
    });
  }
  if constexpr (!GUESS) {
    static_for<0, NV>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      if (k < nv) {  // wave-uniform
        const BcT xb = bcast_prepare<W>(T[k]);
        const double p = value_bcast<W, k>(xb);
        pmin = min_raw(pmin, p);
        const double rp = fast_rcp(p);
        const double t = T[k] * rp;
        const double nt = (li == k) ? rp - 1.0 : -t;
        static_for<0, NT>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j != k) T[j] = fma_bcast<W, j>(T[j], xb, nt);
        });
        T[k] = (li == k) ? -rp : t;
      }
    });
  }
  if (!(pmin > 0.0)) status = STATUS_NOT_PD;
  PINKHIP_TICK(1);  // initial sweeps
  // The diagonal entry of a lane's own row: kept in a register of its own from here on (a pivot on a run-time index
  // cannot address "register p of lane p"; the copy inside T is not maintained and never read).
  double tdiag = 0.0;
#pragma unroll
  for (int j = 0; j < NT; ++j)
    if (j == li) tdiag = T[j];

  // x0 = -H^-1 c = T_BB c; the same product gives the rows of G: slack = h - G x0 = h + (T c)_row
  // (principal pivoting from a guessed active set: with v = c on the free coordinates and -bound on the fixed ones the
  // same product is x on the free coordinates, T_FF c_F - T_FN x_N, and c_i - g_i on the fixed ones)
  double x = 0.0, u = 0.0;
  {
    double vstart = in ? ci : 0.0;
    if constexpr (PPX) {
      if (state != 0) vstart = -((state == 1) ? lbv : ubv);
    }
    const BcT cb = bcast_prepare<W>(vstart);
    double r0 = 0.0, r1 = 0.0;
    static_for<0, NV>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      if constexpr (j % 2 == 0) r0 = fma_bcast<W, j>(r0, cb, T[j]);
      else r1 = fma_bcast<W, j>(r1, cb, T[j]);
    });
    x = in ? r0 + r1 : 0.0;
    if (dlane) u = hv + (r0 + r1);
    // principal pivoting keeps ONE quantity per coordinate: x of a free one, -g (the negated gradient entry, whose
    // sign is the test of the multiplier) of a fixed one -- both move by -col nu in an exchange
    if constexpr (PPX) {
      if (state != 0) x -= ci;
    }
    if constexpr (PPMD) {
      if (dlane) x = u;  // (... and per dense row: its slack while it is inactive, its multiplier while it is active)
    }
  }
  // n^T H^-1 n (the unreduced curvature along a constraint normal): the reference of the linear-dependence test.
  // Box-only problems never need it: a free coordinate always has curvature left.
  // (started from a guessed active set the tableau does not hold it: a lower bound of its size stands in -- 1 / H_ii for
  // a coordinate, |g|^2 / max H_ii for a row; the tests against it are 1e-6 and 1e-12, dependence leaves 1e-26)
  double zd0 = -tdiag;
  if constexpr (PPMD) {
    const double hmax = -group_min<W>(in ? -hii : 0.0);
    zd0 = (li < NV) ? approx_rcp(hii) : approx_rcp(ginv * ginv * hmax);
  }
  // Conditioning estimate  max_i H_ii (H^-1)_ii <= cond(H)  (both diagonals are at hand).  Beyond the threshold the
  // explicitly updated inverse is not expected to certify its result (weakly regularised objectives: a rank-deficient
  // task stack made positive definite by `damping` alone, pink/solve_ik.py:55, examples/humanoid_jvrc.py:69-81): the
  // group skips the tableau iteration and goes to the Goldfarb-Idnani code right away instead of paying both.
  {
    const double hii0 = (li < NV && in) ? sm[SL::tri(li < NV ? li : 0) + (li < NV ? li : 0)] : 0.0;
    // (a coordinate the guess fixed: H_ii over its Schur complement T_ii -- H_ii (H^-1)_ii on the free set plus this one:
    // the flat directions of a weakly regularised objective show there when the guess fixes the coordinates they live on)
    const double kest = -group_min<W>((PPX && state != 0) ? -hii0 * approx_rcp(tdiag) : hii0 * tdiag);
    PINKHIP_TRACEF(li == 0, "[sweep g%d] kest %.3e\n", g, kest);
    if (status == STATUS_OPTIMAL && !(kest <= PINKHIP_SWEEP_ROUTE_COND)) status = STATUS_ROUTED;  // (NaN: routed)
  }
  PINKHIP_TICK(2);  // x0

  // ------------------------------------------------------------------ dual active set on the tableau
  // The kernel arguments that are only needed from here on (bounds, iteration cap, the task rows of the refinement,
  // output pointers) are re-read from the kernarg segment through an opaque pointer where they are used: carried in
  // scalar registers across the loop they spill (v_writelane / v_readlane inside every trip).
  const KernelArgs *late = &a;
  if constexpr (!Src::kOnTheFly) late = kernarg_reload<KernelArgs>(a);
  // violation threshold relative to 1 + |bound|: the round-off of the iterate grows with the dimension and so does
  // the threshold; same rule as oracle/gi_oracle.c and ik_kernels_packed.h
  const double tol = 1e-13 * (nv > 8 ? nv * 0.125 : 1.0);
  const double thr_lo = -tol * (1.0 + fabs(lbv)), thr_up = -tol * (1.0 + fabs(ubv));  // infinite bound: never violated
  const double thr_d = -tol * (1.0 + fabs(hv) * ginv);
  const int max_iter = late->max_iter > 0 ? late->max_iter : 20 * (nv + md) + 50;
  const bool empty_box_somewhere = wave_any(in && ubv - lbv < (thr_lo > thr_up ? thr_lo : thr_up));
  // per lane (state: above): x = coordinate value; u = multiplier (fixed coordinate, active row) or slack h - g x
  // (inactive row)
  // ... and as factors of the update of a step (kept in registers, changed where the state changes): phi = -1 / +1 for a
  // coordinate fixed at lb / ub and +1 for an active inequality row (the multiplier moves by -phi col nu and may block),
  // xfree = 1 for a free coordinate (x moves by -col nu)
  double phi = 0.0, xfree = 1.0;
  // group-uniform
  int it = 0, eq_next = 0, src = 0, kind = 0;  // kind 0: lower bound, 1: upper bound, 2: dense row, 3: equality row
  double uplus = 0.0;
  // principal pivoting: the interval the lane's quantity has to lie in and the thresholds of the two tests
  double blo = lbv, bhi = ubv, tlo = thr_lo, thi = thr_up;
  // (a dense row: slack resp. multiplier in [0, inf); the slack of an inactive row against the threshold of the dual
  // method, u |g|^-1 < thr_d.  An equality that waits for its turn is no candidate, and its "violation" is its slack)
  double thr_row = 0.0;
  if constexpr (PPMD) {
    thr_row = thr_d * fast_rcp(ginv);
    if (li >= NV) {
      blo = dlane ? 0.0 : -INF;
      bhi = INF;
      tlo = (dlane && dr >= n_eq) ? thr_row : -INF;
      thi = 0.0;
    }
  }
  if constexpr (PPX) {
    if (state != 0) {
      blo = (state == 1) ? -INF : 0.0;
      bhi = (state == 1) ? 0.0 : INF;
      tlo = 0.0;
      thi = 0.0;
    }
    // an empty box is quadprog's "constraints are inconsistent" wherever the iteration would come across it
    if (empty_box_somewhere) {  // (wave-uniform)
      const bool empty = group_first_lane<W>(in && ubv - lbv < (thr_lo > thr_up ? thr_lo : thr_up)) < W;
      if (empty && status == STATUS_OPTIMAL) status = STATUS_INFEASIBLE;
    }
  }
  bool at_point = false;
  bool running = (status == STATUS_OPTIMAL);
  bool need_sel = true;
  bool refined = (status == STATUS_ROUTED);  // the closing refinement step(s) of this group have been taken
  int nref = 0;
  double dprev = 0.0;  // largest entry of the previous refinement step

  // ------------------------------------------------------------------ one step of iterative refinement
  // The tableau is an explicitly updated inverse: after ~60 pivots x carries cond(H) eps times a growth factor
  // (measured 2e-13 at nv = 30, 4e-10 on the JVRC-shaped batch, against 5e-15 of an orthogonal factorisation).  One
  // Newton step on the KKT system of the final active set, with the residual formed from the problem AS STATED (H, c,
  // G as they were stacked, parked in LDS; not the tableau), restores full accuracy: r_F = (H x + G_A^T lambda_A + c)_F on the free
  // coordinates, r_A = G_A x - h_A on the active rows, (x_F, lambda_A) += T_BB r.  The step is taken INSIDE the loop,
  // in the trip in which a group finds no violated constraint left: the product with T is the one every trip forms
  // anyway (with r in the place of the indicator of the entering column), and T never has to outlive the loop (when it
  // did, the register allocator kept two copies of it and moved one onto the other in every trip).
  // row li of the stated KKT matrix times a vector held one entry per coordinate lane: a coordinate lane reads
  // H[li][j] = H[j][li] from the packed triangle, the lane of a dense row its row of G (= column entries the
  // coordinate lanes parked)
  auto krow_times = [&](double v) -> double {
    const BcT vb = bcast_prepare<W>(v);
    const int base = (li < NV) ? SL::tri(li) : SL::oG + (dlane ? dr : 0) * SL::GP;
    double h0 = 0.0, h1 = 0.0;
    static_for<0, NV>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      const int ad = (li < NV && j > li) ? SL::tri(j) + li : base + j;
      const double hv_ = sm[ad];
      if constexpr (j % 2 == 0) h0 = fma_bcast<W, j>(h0, vb, hv_);
      else h1 = fma_bcast<W, j>(h1, vb, hv_);
    });
    return h0 + h1;
  };
  // ... and the same products are the KKT certificate of the point the iteration arrived at (`fails`: this lane's
  // condition does not hold at x): stationarity on the free coordinates, the sign of the gradient on the fixed ones
  // (their multipliers), the bounds of the free ones, active rows met, inactive rows not violated.  The problem is
  // strictly convex: a point that passes IS the minimiser, whatever path led there; one that does not (the explicitly
  // updated inverse loses cond(H)^2 eps when small pivots follow each other: weakly regularised objectives, cond(H)
  // >~ 1e8, scripts/gpu_fuzz.py) is handed to the Goldfarb-Idnani kernel.
  auto residual = [&](bool &fails) -> double {
    const double kx = krow_times(in ? x : 0.0);
    const double ci_ = in ? sm[SL::oC + li] : 0.0;
    double grad = in ? kx + ci_ : 0.0;
    bool arow = false;
    if constexpr (DENSE) {
      if (md > 0) {
        arow = dlane && state == 1;
        // + G_A^T lambda_A
        const BcT lamb = bcast_prepare<W>(arow ? u : 0.0);
        double gl = 0.0;
        static_for<0, MD>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          gl = fma_bcast<W, NV + d>(gl, lamb, sm[SL::oG + d * SL::GP + li]);
        });
        if (in) grad += gl;
      }
    }
    // scale of a gradient entry: |c| + max H_ii max |x| over the group
    const double hii = in ? sm[SL::tri(li < NV ? li : 0) + (li < NV ? li : 0)] : 0.0;
    const double gsc = group_min<W>(-hii) * group_min<W>(in ? -fabs(x) : 0.0) - group_min<W>(-fabs(ci_));
    const double gtol = PINKHIP_SWEEP_CERT_TOL * gsc;
    // (written so that a NaN -- an overflowed tableau -- fails)
    fails = false;
    if (in) {
      if (state == 0) fails = !(fabs(grad) <= gtol && x - lbv >= 10.0 * thr_lo && ubv - x >= 10.0 * thr_up);
      else fails = (state == 1) ? !(grad >= -gtol) : !(grad <= gtol);
    }
#if defined(PINKHIP_SWEEP_DEBUG_WHY) && defined(__HIP_DEVICE_COMPILE__)
    if (fails) printf("[cert blk %lld g%d li%d it%d nref%d] state %d x %.17g lb %.17g ub %.17g grad %.3e gtol %.3e thr %.3e %.3e\n", block, g, li, it, nref, state, x,
                      lbv, ubv, grad, gtol, thr_lo, thr_up);
#endif
    double r = (in && state == 0) ? grad : 0.0;
    if constexpr (DENSE) {
      if (dlane) {
        const double slack = (hv - kx) * ginv;  // (rows are g x <= h; normalised like the selection's threshold)
        if (arow) {
          r = kx - hv;  // residual of an active row
          // ... and its multiplier has the sign of an active inequality: the corrections of the closing trips move it with
          // x, and a row that is active with a multiplier of zero (a flat direction of a weakly regularised H) can come
          // out at -1e-8 -- stationary to 1e-13 with it, 0.09 away from the minimiser (scripts/gpu_fuzz.py seed 537045,
          // cond(H) = 2e8).  In the gradient's units: lambda |g| >= -gtol.
          fails = !(fabs(slack) <= -10.0 * thr_d) || (dr >= n_eq && !(u >= -gtol * ginv));
        } else if (state == 0 && dr >= n_eq) {
          fails = !(slack >= 10.0 * thr_d);
        }
      }
    }
    return r;
  };

#ifndef PINKHIP_SWEEP_LDS_COLUMN
#define PINKHIP_SWEEP_LDS_COLUMN 0
#endif
  // Column p of T for a group-uniform run-time p (-1: none).  T is symmetric: the column is row p, which lane p holds
  // in its registers.  Either lane p hands it over through LDS (NT / 2 16-byte writes by one lane, one read per lane:
  // the LDS pipe is otherwise idle and the VALU is what the kernel is short of), or every lane multiplies its row
  // with the indicator of p (NT broadcast-FMAs, no LDS round trip on the dependent chain).
  // The unmaintained copy of the diagonal entry inside the row, T[li][li] as the pivots' generic FMAs leave it: followed in
  // a register of its own (one FMA per pivot, the same operation on the same operands: bit-identical) -- the closing
  // trips need it to take it out of their product with T, and picking "register li of lane li" there cost 2 NT selects
  double sdiag_run = tdiag;
  double *rowbuf = sm + SL::oR;
  // From here on the row lives in register tuples when the column is read by index (same registers, renamed)
  constexpr bool IDX = PINKHIP_SWEEP_INDEXED_COLUMN && !PINKHIP_SWEEP_LDS_COLUMN && W >= 32;
  TabRegs<IDX ? NT : 1> R;
  R.clear();
  if constexpr (IDX) {
    static_for<0, NT>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      R.template set<j>(T[j]);
    });
  }
  auto tget = [&](auto Jc) -> double {
    constexpr int j = decltype(Jc)::value;
    if constexpr (IDX) return R.template get<j>();
    else return T[j];
  };
  auto tset = [&](auto Jc, double v) {
    constexpr int j = decltype(Jc)::value;
    if constexpr (IDX) R.template set<j>(v);
    else T[j] = v;
  };
  auto column_of = [&](int p) -> double {
    if constexpr (IDX) {
      return R.template at_group_uniform<W>(p);
    } else if constexpr (PINKHIP_SWEEP_LDS_COLUMN) {
      if (li == p) {
        Pair *dst = reinterpret_cast<Pair *>(__builtin_assume_aligned(rowbuf, 16));
#pragma unroll
        for (int j = 0; j + 1 < NT; j += 2) dst[j >> 1] = Pair{T[j], T[j + 1]};
        if constexpr (NT % 2) rowbuf[NT - 1] = T[NT - 1];
      }
      wave_sync();
      const double c = rowbuf[li < NT ? li : 0];
      wave_sync();  // (read before the next hand-over overwrites the row)
      return (p >= 0 && li < NT) ? c : 0.0;
    } else {
      const BcT eb = bcast_indicator<W>(p);
      constexpr int NC = PINKHIP_SWEEP_COLUMN_CHAINS(NT);
      double c[NC] = {};
      static_for<0, NT>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        c[j % NC] = fma_bcast<W, j>(c[j % NC], eb, tget(Jc));
      });
      if constexpr (NC == 4) return (c[0] + c[1]) + (c[2] + c[3]);
      else return c[0] + c[1];
    }
  };

  // sweep (nonbasic -> basic: sg = +1) or reverse sweep (basic -> nonbasic: sg = -1) of the tableau on pi (-1: none, with
  // pvt = 1, rp = 0), col = column pi
  auto pivot_on = [&](int pi, double col, double pvt, double rp, double sg) {
    double t = col * rp;
    double cp = col;
    if (li == pi) {
      // lane pi: its row becomes sg col / p -- as T[pi][j] - (1 - sg / p) col_j -- and the column the other lanes
      // see at j = pi is p - sg, so that their entry T[m][pi] - t_m (p - sg) becomes sg t_m
      t = 1.0 - sg * rp;
      cp = pvt - sg;
    }
    const BcT xb = bcast_prepare<W>(cp);
    const double nt = -t;
    static_for<0, NT>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      tset(Jc, fma_bcast<W, j>(tget(Jc), xb, nt));
    });
    sdiag_run = fma(cp, nt, sdiag_run);  // (what the broadcast-FMA above just made of this lane's register li)
    tdiag = (li == pi) ? -rp : tdiag - t * col;
  };

  // Which of the two iterations a group is in (group-uniform).  With dense rows the same exchange serves the rows -- an
  // inactive row whose slack is negative is activated, an active inequality whose multiplier is negative is released; the
  // lane of a row keeps slack resp. multiplier in the place of x resp. -g, and both move by -col nu like everything
  // else -- but the rows bring the degenerate exchange (a normal that depends on the active ones has no curvature left
  // and cannot be pivoted on) and the verdict "inconsistent".  Both are the dual method's: a group that meets an
  // irregular pivot, or runs out of trips, releases every constraint whose multiplier has the wrong sign until none is
  // left (`restoring`; the active set only shrinks: it ends) and continues with Goldfarb-Idnani's trips from that
  // dual-feasible basis.  Equalities are activated first, in order, and never leave.
  bool ppm_mode = PPX;
  bool restoring = false;
  const bool cand_lane = in || (DENSE && dlane && dr >= n_eq);
  const bool eq_lane = DENSE && dlane && dr < n_eq;

  for (;;) {
    // (a) entering constraint, for the groups that have none pending: the violated constraint that is farthest away
    // in the metric of the objective, violation / sqrt(n^T Z n) with Z the reduced inverse Hessian -- n^T Z n is the
    // diagonal entry -T[i][i] (free coordinate i: Z_ii; inactive row: g Z g^T).
    // ---- principal pivoting (box-only instantiations) ----------------------------------------------------------
    // The KKT conditions of a strictly convex box QP are a linear complementarity problem with a P-matrix, and the
    // tableau is its principal transform on the free set: x of the free coordinates, multipliers u of the fixed ones.
    // Instead of Goldfarb-Idnani's trip -- entering constraint, ratio test over the multipliers, a second column and a
    // second pivot when a multiplier blocks, the entering constraint kept pending -- every trip exchanges ONE index
    // whose complementarity condition fails: a free coordinate outside its box is fixed at the bound it violates, a
    // fixed one whose multiplier has the wrong sign is freed.  Which one: the largest |violation|^2 / |T_pp| (the
    // change of the objective the exchange brings; for a violated bound Goldfarb-Idnani's reduced steepest edge).
    // No ratio test, no pending constraint, one column and one pivot per trip; scripts/multi_pivot_study.py counted
    // 26.5 trips on the headline batch against 28.5 of the dual method (whose partial steps are trips of their own),
    // where exchanging SEVERAL indices per trip (block principal pivoting, candidate lists) needs 40 .. 80 pivots.
    // The iterates are neither primal nor dual feasible and nothing decreases monotonically: after
    // PINKHIP_SWEEP_PPM_MURTY_AFTER trips a group continues with Murty's least-index rule (finite for P-matrices);
    // the closing KKT certificate and the hand-over behind it are what they were.
    double viol = 0.0;
    int nb_src = 0;     // the index that is exchanged is nonbasic (enters the basis)
    bool iseq = false;  // ... is the next equality
    if constexpr (PPX) {
      if (wave_any(running && ppm_mode)) {
        const bool sel = running && ppm_mode;
        // one test for every coordinate: a free one against its box (with the usual thresholds), a fixed one through
        // -g against (-inf, 0] at its lower bound resp. [0, inf) at its upper bound (exact sign); a dense row through
        // its slack resp. multiplier against [0, inf)
        const double slo = x - blo, sup = bhi - x;
        bool ok_lane = in, nonbasic = state != 0;
        if constexpr (DENSE) {
          nonbasic = (li < NV) ? (state != 0) : (state == 0);
          // (on the way to the dual method only what holds a multiplier is looked at)
          ok_lane = cand_lane && (!restoring || ((li < NV) ? (state != 0) : (state == 1)));
        }
        const bool vlo = ok_lane && slo < tlo, vup = ok_lane && sup < thi;
        viol = vlo ? slo : sup;
        if constexpr (DENSE) {
          if (eq_lane) viol = slo;  // (an equality's turn: its slack goes to zero whatever its sign)
        }
        const float zf = fabsf(static_cast<float>(tdiag));
        const float wz = (zf > 1e-30f) ? approx_rcpf(zf) : 1e30f;
        const float fv = static_cast<float>(viol);
        float key = -(fv * fv) * wz;
        if (it > PINKHIP_SWEEP_PPM_MURTY_AFTER(nv + md)) key = static_cast<float>(li - 64);  // least index
        const float best32 = group_min32<W>((vlo || vup) ? key32_packf(key, li | (vlo ? 0 : 64) | (nonbasic ? 128 : 0)) : 3.0e38f);
        bool conv = false;
        if (sel) {
          if (DENSE && !restoring && eq_next < n_eq) {
            src = NV + eq_next;
            kind = 0;
            nb_src = 1;
            iseq = true;
          } else if (!(best32 < 0.0f)) {
            if (DENSE && restoring) {
              ppm_mode = false;  // dual feasible: Goldfarb-Idnani's trips from here, this trip included
              need_sel = true;
            } else {
              running = false;  // optimal
            }
            conv = DENSE;
          } else {
            const int pl = key32_payload(best32);
            src = pl & 63;
            kind = (pl >> 6) & 1;
            nb_src = pl >> 7;
          }
        }
        if constexpr (DENSE) {
          if (conv) {
            // ... into the variables of the dual method and of the closing trips: the point, the multipliers (slacks of
            // the inactive rows), the factors of a step
            u = (li < NV) ? ((state == 1) ? -x : ((state == 2) ? x : 0.0)) : x;
            phi = (li < NV) ? ((state == 1) ? -1.0 : ((state == 2) ? 1.0 : 0.0)) : ((dlane && state == 1 && dr >= n_eq) ? 1.0 : 0.0);
            xfree = ((li < NV) ? (state != 0) : (state == 1)) ? 0.0 : 1.0;
            if (li < NV && state != 0) x = (state == 1) ? lbv : ubv;
          }
        }
      }
    }
    if (!PPM && wave_any(running && !ppm_mode && need_sel)) {
      const bool sel = running && !ppm_mode && need_sel;
      const double slo = x - lbv, sup = ubv - x;
      const bool vlo = in && slo < thr_lo, vup = in && sup < thr_up;
      // (no curvature left along the normal -- it depends on the active ones: the weight is just large; the step
      // then finds the constraint that has to leave, or that there is none).  The key only ranks the candidates: fp32.
      const float zf = static_cast<float>(-tdiag);
      const float wz = (zf > 1e-30f) ? approx_rcpf(zf) : 1e30f;
      const float flo = static_cast<float>(slo), fup = static_cast<float>(sup);
      const bool clo = vlo && state == 0;
      bool has = clo;
      float key = -(flo * flo) * wz;
      int id = li;
      if (vup && state == 0) {
        const float ku = -(fup * fup) * wz;
        if (!clo || ku < key) key = ku, id = 64 + li;
        has = true;
      }
      if constexpr (DENSE) {
        if (dlane && dr >= n_eq && state == 0 && u * ginv < thr_d) {
          const float fu = static_cast<float>(u);
          key = -(fu * fu) * wz;
          has = true;
        }
      }
      const float best32 = group_min32<W>(has ? key32_packf(key, id) : 3.0e38f);
      const bool none = !(best32 < 0.0f);
      // A fixed coordinate sits on its bound: its other bound can only be violated when the box is empty, and so are
      // both bounds of a free one -- quadprog's "constraints are inconsistent".  Only a wave that holds an empty box
      // looks for it.
      bool bad = false;
      if (empty_box_somewhere) bad = group_first_lane<W>((vlo && vup) || (state != 0 && in && (vlo || vup))) < W;
      if (sel) {
        uplus = 0.0;
        if (bad) {
          status = STATUS_INFEASIBLE;
          running = false;
        } else if (DENSE && eq_next < n_eq) {
          // equalities (the first n_eq dense rows; pink/solve_ik.py:140-149) are activated first, in order
          src = NV + eq_next;
          kind = 3;
          need_sel = false;
        } else if (none) {
          running = false;  // optimal
        } else {
          const int pl = key32_payload(best32);
          src = pl & 63;
          if constexpr (DENSE) kind = (src >= NV) ? 2 : (pl >> 6) & 1;
          else kind = (pl >> 6) & 1;
          need_sel = false;
        }
      }
    }
    if (running) {
      if (++it > max_iter) {
        status = STATUS_MAX_ITER;
        running = false;
      }
      if constexpr (PPMD) {
        if (ppm_mode && it > 2 * PINKHIP_SWEEP_PPM_MURTY_AFTER(nv + md)) restoring = true;  // (the dual method ends what this did not)
      }
    }
    // once no group of the wave is running any more, closing trips take the refinement steps of all of them
    // (`closing` is wave-uniform: an ordinary trip pays one ballot and scalar branches for all of this)
    const bool closing = !wave_any(running);
    const bool ref = closing && !refined;
    if (closing && !wave_any(ref)) break;
    const bool act = running;
    PINKHIP_TICK(3);  // selection

    // (b) column src of T: col_m = sum_j T[m][j] [j == src]; for a finishing group the product T r instead
    double col;
    if (!(PINKHIP_SWEEP_LDS_COLUMN || IDX) || closing) {
      BcT eb = bcast_indicator<W>(act ? src : -1);
      double rres = 0.0, sdiag = 0.0;
      bool cert_fails = false, illc = false;
      if (closing) {
        if constexpr (PPM) {
          // (from here on x is the point: a fixed coordinate sits on its bound)
          if (!at_point) {
            if (state != 0) x = (state == 1) ? lbv : ubv;
            at_point = true;
          }
        }
        rres = residual(cert_fails);
        if constexpr (PPX) {
          // The conditioning estimate of the start-up saw the coordinates the guess left free; the ones the iteration
          // freed since belong to it as well: max_i H_ii (H_FF^-1)_ii over the FINAL free set, same threshold, same
          // consequence (the certificate cannot see an error along a direction in which the residual is below its own
          // round-off -- scripts/gpu_fuzz.py seeds 3063897, 3074349, 3085897: weakly regularised draws 8e-5 from the exact
          // minimiser behind a passed certificate, where the Goldfarb-Idnani code is within 1e-9)
          const double hii_ = (in && state == 0) ? sm[SL::tri(li < NV ? li : 0) + (li < NV ? li : 0)] : 0.0;
          const double kfin = -group_min<W>(hii_ * tdiag);
          illc = !(kfin <= PINKHIP_SWEEP_ROUTE_COND);
        }
        cert_fails = group_first_lane<W>(cert_fails) < W;
        if (!ref || status != STATUS_OPTIMAL) rres = 0.0;
        // (the product below meets the unmaintained copy of the diagonal inside T: replaced by the maintained one)
        sdiag = sdiag_run;
        eb = bcast_select<W>(ref, bcast_prepare<W>(rres), eb);
      }
      constexpr int NC = PINKHIP_SWEEP_COLUMN_CHAINS(NT);
      double c[NC] = {};
      static_for<0, NT>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        c[j % NC] = fma_bcast<W, j>(c[j % NC], eb, tget(Jc));
      });
      if constexpr (NC == 4) col = (c[0] + c[1]) + (c[2] + c[3]);
      else col = c[0] + c[1];
      if (closing) {
        // A correction that is not small (a nearly singular H: T is then a poor inverse, accurate to cond(H) eps)
        // calls for another step, at most three in all: every step shrinks the residual by the relative accuracy of T.
        // (the multipliers of the active rows are basic variables too: corrected with x, so that the certificate's
        // gradient is formed with multipliers as accurate as the point)
        const bool arow_ = DENSE && dlane && state == 1;
        const double dbv = (ref && ((in && state == 0) || arow_)) ? col + (tdiag - sdiag) * rres : 0.0;
        const double dxv = in ? dbv : 0.0;
        // (the floor is relative to the size of the point: the correction itself carries cond(H) eps |x| of round-off
        // -- measured 2e-13 .. 1e-12 at |x| = 0.15 on the closed-loop batch of bench.py, where an absolute floor of 1e-13
        // asked for steps that could not halve and sent a few robots per step to the hand-over.  It is applied either way.)
        const double xmax = -group_min<W>(in ? -fabs(x) : 0.0);
        const bool more = group_first_lane<W>(fabs(dxv) > 1e-9 * fabs(x) + 1e-11 * (1.0 + xmax)) < W;
        // ... as long as the steps contract.  Where T is no inverse any more (cond(H) ~ 1e13 and beyond: the relative
        // error of T reaches one) the "correction" is as large as x or grows from step to step: it is not applied, x
        // stays what the iteration left -- inside its box, by construction (scripts/gpu_fuzz_rollout.py seed 14:
        // cond(H) = 4e13, a correction of 6e7 radians reported as optimal).
        const double dmax = -group_min<W>(-fabs(dxv));
        const bool sane = dmax <= ((nref == 0) ? 0.1 * xmax : 0.5 * dprev);
        if (ref) {
          if (status != STATUS_OPTIMAL || illc) {
            status_before = status;
            // a verdict -- inconsistent, out of iterations, a non-positive pivot in the first sweeps -- reached on a
            // tableau that may have lost its accuracy is not handed out either: the Goldfarb-Idnani code (Cholesky
            // factor, orthogonal updates) confirms it or solves the instance (rare in practice: twice the work there)
            status = STATUS_BREAKDOWN;
            PINKHIP_WHY(2);
            refined = true;
          } else if (!cert_fails && !more) {
            if (sane) x += dxv;  // (below 1e-9 |x|: the certificate holds for the corrected point as well)
            refined = true;
          } else if (sane && nref < 3) {
            x += dxv;  // ... and the next closing trip certifies the corrected point
            if (arow_) u += dbv;
            dprev = dmax;
            ++nref;
          } else {
            status = STATUS_BREAKDOWN;
            PINKHIP_WHY(3);
#if defined(PINKHIP_SWEEP_DEBUG_WHY) && defined(__HIP_DEVICE_COMPILE__)
            if (li == 0) printf("[close blk %lld g%d it%d nref%d] cert_fails %d more %d sane %d dmax %.3e xmax %.3e dprev %.3e\n", block, g, it, nref, (int)cert_fails,
                                (int)more, (int)sane, dmax, xmax, dprev);
#endif
            refined = true;
          }
        }
        // (a closing trip that asks for another one falls through the rest of the body, everything masked off: a
        // second back edge -- `continue` -- makes the register allocator keep two copies of T and move one onto the
        // other per trip)
        if (!wave_any(!refined)) break;
      }
    } else {
      col = column_of(act ? src : -1);
    }
    if (li == src) col = tdiag;
    int pi = -1;
    double pvt = 1.0, rp = 0.0;  // (no pivot in this group: t = 0 leaves T and tdiag as they are)
    double sg = -1.0;
    if constexpr (PPX) {
      if (PPM || wave_any(act && ppm_mode)) {
        const bool actP = act && ppm_mode;
        // (c') the exchange: lane src's quantity goes to zero along the column, nu = -num / T[src][src]; every lane's
        // quantity moves by -col nu -- whatever that does to the signs: the next selection sees it
        // (lower side: x - lo has to rise to zero, upper side: hi - x)
        const double vs = group_bcast<W>(viol, src);
        const double num = kind ? vs : -vs;
        const double pv = group_bcast<W>(tdiag, src);
        PINKHIP_TICK(4);  // column
        const double rpv = fast_rcp(pv);
        // entering the basis (a coordinate freed, a row activated): sweep; leaving it: reverse sweep
        const double sgp = nb_src ? 1.0 : -1.0;
        bool act2 = actP;
        if constexpr (!DENSE) {
          // The pivot has the sign of the curvature it stands for; anything else: H is not positive definite on this set
          // (or the tableau is no inverse any more) -- the Goldfarb-Idnani code, whose Cholesky factorisation decides
          // that, takes the instance
          if (actP && !(pv * sgp > 0.0)) {
            status = STATUS_NOT_PD;
            running = false;
            act2 = false;
          }
        } else {
          // A coordinate that is fixed and a row that is activated pivot on MINUS the curvature left along their normal,
          // n^T Z n: (next to) nothing of it left = the normal depends on the active ones.  The other two exchanges, a
          // coordinate freed and a row released, pivot on the reciprocal of such a curvature: positive, of any size.
          const double zr = group_bcast<W>(zd0, src);
          const bool recip = (src < NV) == (nb_src != 0);
          const bool irregular = actP && !((recip ? pv : -pv) > (recip ? 0.0 : PINKHIP_SWEEP_PPM_MIN_CURV * zr));
          // (the right-hand side of a dependent equality; cross-lane: wave-uniform control flow)
          double hs = 0.0;
          if (wave_any(irregular && iseq)) hs = group_bcast<W>(hv, src);
          if (irregular) {
            act2 = false;
            PINKHIP_TRACEF(li == 0, "[ppm g%d it%d] irregular: src %d kind %d nonbasic %d pv %.3e zref %.3e num %.3e eq %d restoring %d\n", g, it, src,
                           kind, nb_src, pv, zr, num, (int)iseq, (int)restoring);
            if (iseq && fabs(num) <= 1e-9 * (1.0 + fabs(hs))) {
              ++eq_next;  // implied by the active ones and met: nothing to add
            } else if (!restoring) {
              restoring = true;
            } else {
              status = STATUS_BREAKDOWN;  // (not even a release is regular: the tableau is no inverse any more)
              PINKHIP_WHY(1);
              running = false;
            }
          }
        }
        const double nu = act2 ? -num * rpv : 0.0;
        x = fma(-col, nu, x);
        if (act2) {
          pi = src;
          pvt = pv;
          rp = rpv;
          sg = sgp;
          if (li == src) {
            if (!DENSE || li < NV) {
              if (state != 0) {
                // off its bound, to where its gradient entry is zero
                x = ((state == 1) ? lbv : ubv) + nu;
                state = 0;
                blo = lbv;
                bhi = ubv;
                tlo = thr_lo;
                thi = thr_up;
              } else {
                // onto the bound it violates: -g = -nu (the multiplier |nu| with the sign of its side)
                x = -nu;
                state = kind + 1;
                blo = kind ? 0.0 : -INF;
                bhi = kind ? INF : 0.0;
                tlo = 0.0;
                thi = 0.0;
              }
            } else if (state == 0) {
              x = nu;  // the multiplier of the row
              state = 1;
              tlo = 0.0;
              if (dr < n_eq) blo = -INF;  // (an equality's: of either sign)
            } else {
              x = -nu;  // the slack it opens
              state = 0;
              tlo = thr_row;
            }
          }
          if constexpr (DENSE) {
            if (iseq) ++eq_next;
          }
        }
        PINKHIP_TICK(5);
        PINKHIP_TICK(6);
      }
    }
    if constexpr (!PPM) {
    if (!PPMD || wave_any(act && !ppm_mode)) {
    const bool actG = act && !ppm_mode;
    // what has to go to zero: the distance of the entering coordinate to its bound resp. the (negative) slack of the
    // entering row; pv = T[src][src] = -n^T Z n
    double cand = (kind == 0 ? lbv : ubv) - x;
    if constexpr (DENSE) cand = (li < NV) ? cand : -u;
    const double num = group_bcast<W>(cand, src);
    double pv = group_bcast<W>(tdiag, src);
    bool lin_dep = false;
    if constexpr (DENSE) {
      // an entering normal that depends on the active ones has no curvature left (a coordinate too, once dense rows
      // are active).  The reduced curvature n^T Z n is what successive pivots leave of n^T H^-1 n: a difference, whose
      // round-off floor is cond(K_BB) eps n^T H^-1 n (ik_kernels_packed.h forms it as a sum of squares).
      // When little is left (under 1e-6 of it: rare), the curvature is formed again as a sum of squares: the free
      // coordinates' part w of the column is the primal step direction, and w^T H w = n^T Z n (K_BB (w, mu) = n gives
      // w^T H w = w^T n - (G_A w)^T mu = w^T n).  A direction that is pure round-off (relative size delta) then
      // leaves delta^2 instead of delta: dependence shows as < 1e-12 of n^T H^-1 n whatever the conditioning
      // (scripts/gpu_fuzz.py seed 12979: floor 6e-11, against a threshold of 1e-10 before).
      const double z0 = group_bcast<W>(zd0, src);
      const bool little = actG && !(-pv * 1e6 > z0);
      if (wave_any(little)) {
        const double w = (in && state == 0) ? col : 0.0;
        const double hw = krow_times(w);
        const double curv = group_sum<W>((li < NV) ? w * hw : 0.0);
        if (little) pv = -curv;
      }
      lin_dep = !(-pv * 1e12 > z0);
      PINKHIP_TRACEF(li == 0 && actG, "[sweep g%d it%d] enter src %d kind %d num %.3e pv %.3e z0 %.3e little %d lin_dep %d\n", g, it, src, kind,
                     num, pv, z0, (int)little, (int)lin_dep);
    }
    // (a group without an entering constraint computes on garbage from here on: everything it could change is
    // masked by actG / act2 below)
    PINKHIP_TICK(4);  // column
    // (c) step: the driving parameter nu (multiplier of the entering constraint) moves every quantity along the
    // column: free coordinates x -= col nu, multipliers of fixed coordinates u -= phi col nu (phi = -1 at lb, +1 at
    // ub), multipliers of active rows / slacks of inactive rows u -= col nu.  Full step: the entering constraint
    // becomes tight, nu = num / (-pv); it is cut short where the first multiplier reaches zero.
    // (BIG stands for "no bound on the step")
    const double rpv = fast_rcp(pv);  // (also the reciprocal of the pivot when the entering constraint is added)
    const double sgn = (num >= 0.0) ? 1.0 : -1.0;
    const double full = lin_dep ? BIG : -fabs(num) * rpv;
    const double rate = phi * col * sgn;
    const bool blocking = actG && rate > 0.0;
    const double ratio = blocking ? max_raw(u, 0.0) * fast_rcp1(rate) : BIG;
    const double k1 = group_min<W>(ratio);
    const int kd = group_first_lane<W>(blocking && ratio == k1) & (W - 1);
    const double tstep = (k1 < full) ? k1 : full;
    const bool stuck = !(tstep < BIG);
    double hs = 0.0;
    if constexpr (DENSE) {
      // (the bound / right-hand side of the entering constraint; cross-lane: wave-uniform control flow)
      if (wave_any(actG && stuck)) hs = group_bcast<W>((li < NV) ? (kind == 0 ? lbv : ubv) : hv, src);
    }
    if (actG && stuck) {
      const bool tiny = DENSE && fabs(num) <= 1e-9 * (1.0 + fabs(hs));
      PINKHIP_TRACEF(li == 0, "[sweep g%d it%d] stuck: src %d kind %d num %.3e hs %.3e pv %.3e lin_dep %d k1 %.3e full %.3e tiny %d\n",
                     g, it, src, kind, num, hs, pv, (int)lin_dep, k1, full, (int)tiny);
      if (kind == 3 && tiny) {
        // equality implied by the active ones and already satisfied: nothing to add
        ++eq_next;
        need_sel = true;
      } else if (tiny) {
        // an inequality that depends on the active ones, that no drop can help, and that is violated by round-off only:
        // not "inconsistent" -- the bound moves to where the point is (ik_kernels_packed.h has the measurement)
        if (li == src) {
          if (li >= NV) hv -= u, u = 0.0;
          else if (kind == 0) lbv = x;
          else ubv = x;
        }
        need_sel = true;
      } else {
        status = STATUS_INFEASIBLE;
        running = false;
      }
    }
    const bool act2 = actG && running && !stuck;
    const bool do_add = act2 && !(k1 < full);
    const bool do_drop = act2 && !do_add;
    {
      const double nu = act2 ? sgn * tstep : 0.0;
      const double d = col * nu;
      x = fma(-xfree, d, x);
      u = fma((DENSE && li >= NV) ? -1.0 : -phi, d, u);  // rows: slack or multiplier, both move by -col nu
      if constexpr (DENSE) uplus += (kind == 3) ? nu : fabs(nu);
      else uplus += fabs(nu);
    }
    PINKHIP_TICK(5);  // step lengths, x / u update
    // (d) pivot: on src (the entering constraint becomes tight) or on kd (the blocking constraint leaves; the
    // entering one stays pending and its column is extracted again from the new tableau)
    if (do_add) {
      pi = src;
      pvt = pv;
      rp = rpv;
      if (li == src) {
        if (li < NV) {
          state = kind + 1;
          x = (kind == 0) ? lbv : ubv;
          phi = (kind == 0) ? -1.0 : 1.0;
        } else {
          state = 1;
          phi = (DENSE && dr >= n_eq) ? 1.0 : 0.0;  // equalities never leave
        }
        xfree = 0.0;
        u = uplus;
      }
      if (DENSE && kind == 3) ++eq_next;
      need_sel = true;
    }
    if (wave_any(do_drop)) {
      const double ck = column_of(do_drop ? kd : -1);
      const double pk = group_bcast<W>(tdiag, kd);
      if (do_drop) {
        col = (li == kd) ? tdiag : ck;
        pi = kd;
        pvt = pk;
        rp = fast_rcp(pk);
        if (li == kd) {
          state = 0;
          u = 0.0;
          phi = 0.0;
          xfree = 1.0;
        }
      }
    }
    PINKHIP_TICK(6);  // column of the leaving constraint
    // sweep (nonbasic -> basic: sg = +1) or reverse sweep (basic -> nonbasic: sg = -1) on pi.  Basic = free
    // coordinate / active row, so an add pivots a coordinate out and a row in, a drop the other way round.
    if (!ppm_mode) sg = ((!DENSE || pi < NV) == do_add) ? -1.0 : 1.0;
    }
    }
    pivot_on(pi, col, pvt, rp, sg);
    PINKHIP_TICK(7);  // pivot
  }
  PINKHIP_TICK(8);  // exit
  // ------------------------------------------------------------------ write-out
  if constexpr (Src::kOnTheFly) {
    terms->x = in ? x : 0.0;
    terms->status = status;
  }
  {
    // (the instance index is formed again from an opaque lane id: as a value kept from the start of the kernel it was
    // the one register too many of the tableau loop -- spilled in every wave's prologue, +7 % HBM traffic)
    int ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(ln));
#endif
    const long long bw = block * G + ln / W;
    // the eliminated front coordinates follow from the point: x_e = -(c_e + h_e . x + H_ee' x_e') / H_ee, last one first
    double xe[NE > 0 ? NE : 1] = {};
    if constexpr (NE > 0) {
      const double *fs = sm + SL::stride + 2 * W;
      const double xr = in ? x : 0.0;
      if constexpr (NE == 2) xe[1] = -(fs[3] + group_sum<W>(sm[SL::stride + W + li] * xr)) * fs[4];
      xe[0] = -(fs[0] + group_sum<W>(sm[SL::stride + li] * xr) + ((NE == 2) ? fs[2] * xe[NE - 1] : 0.0)) * fs[1];
    }
    if (bw < late->B) {
      if (in) late->dq[bw * (long long)nvs + NE + li] = x * late->out_scale;
      if constexpr (NE > 0) {
        if (li < NE) late->dq[bw * (long long)nvs + li] = ((li == 0) ? xe[0] : xe[NE - 1]) * late->out_scale;
      }
      if (li == 0) {
        late->status[bw] = status;
        if (late->iters) late->iters[bw] = it + 1000 * why;  // (PATH_TABLEAU = 0; a group handed over is written again)
      }
    }
  }
  PINKHIP_TRACEF(li == 0, "[sweep g%d] exit status %d it %d nref %d\n", g, status, it, nref);
  return status;  // (of this lane's group)
}

// Doubles of LDS per QP of the kernel below: the sweep tableau's parking area or, for a group that is handed over, the
// Goldfarb-Idnani kernel's working set -- whichever is larger (the host sets KernelArgs::lds_pitch to it).
template <int NV, int MD, int W>
__host__ __device__ constexpr int sweep_kernel_lds_doubles(int md) {
  return SweepLds<NV, MD, W>::stride > LdsP<NV>::stride(MD > 0 ? md : 0) ? SweepLds<NV, MD, W>::stride : LdsP<NV>::stride(MD > 0 ? md : 0);
}

// (everything that touches the dynamic LDS must be inlined into the kernel: as a called function the body reached it
// through a per-kernel offset table and faulted on the first access in the largest instantiations)
template <int NV, int MD, int W>
__device__ __forceinline__ void ik_solve_sweep_body(const KernelArgs &a, long long block) {
  const int st = ik_sweep_instance<NV, MD, W>(a, block);
  // a result that did not pass its KKT certificate (STATUS_BREAKDOWN: the explicitly updated inverse lost too much
  // accuracy) is not handed out: the Goldfarb-Idnani kernel -- orthogonal factors, slower, stable -- solves that
  // instance again, here, in the same wavefront (wave-uniform branch: rare, weakly regularised objectives)
#ifndef PINKHIP_SWEEP_NO_HANDOVER  // (development: time / debug the tableau code alone)
#ifdef PINKHIP_SWEEP_FORCE_HANDOVER  // (development: every instance takes the hand-over)
  const bool over = true;
#else
  const bool over = st == STATUS_BREAKDOWN || st == STATUS_ROUTED;
#endif
  if (wave_any(over)) {
    wave_sync();
    // (the arguments are read again from the kernel-argument segment: kept in registers for this rare call they would
    // be live through the whole tableau loop -- a dozen spilled registers in every wave's prologue, +30 % HBM traffic)
    const KernelArgs *again = kernarg_reload<KernelArgs>(a);
    long long blk = block;  // (opaque: or the instance index of the tableau code stays live for this call)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(blk));
#endif
    if constexpr (NV > W) {
      // front coordinates eliminated (ik_sweep_instance): the Goldfarb-Idnani code holds all NV coordinates of an instance
      // on the 64 lanes of the wave -- the instances of the wave's groups one after the other
      constexpr int G = kWave / W;
      const int path = st == STATUS_ROUTED ? PATH_ROUTED : PATH_HANDOVER;
      static_for<0, G>([&](auto Gc) {
        constexpr int gg = decltype(Gc)::value;
        const bool og = bcast_i(over ? 1 : 0, gg * W) != 0;  // (wave-uniform)
        const int pg = bcast_i(path, gg * W);
        if (og && blk * G + gg < again->B) {
          wave_sync();
          ik_packed_instance<NV, kWave, false>(*again, blk * G + gg, static_cast<HbmTerms *>(nullptr), true, pg);
        }
      });
    } else {
      ik_packed_instance<NV, W, (MD > 0)>(*again, blk, static_cast<HbmTerms *>(nullptr), over, st == STATUS_ROUTED ? PATH_ROUTED : PATH_HANDOVER);
    }
  }
#endif
}

template <int NV, int MD, int W>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_SWEEP3(NV, MD, W) ik_solve_sweep_kernel(KernelArgs a) {
  ik_solve_sweep_body<NV, MD, W>(a, block_id());
}

}  // namespace pinkhip
