// Sweep tableau with *virtual* dense rows: NV coordinates on the W lanes of a group, up to MD dense rows (equalities,
// barrier rows, limit rows that are not axis-aligned) that own NO lane -- NV + MD may exceed W.
//
// ik_sweep.h gives every tableau row a lane: nv = 30 with more than two dense rows needs 64-lane groups (one QP per
// wavefront, 38 of 64 lanes busy: 1.72 ms per 65 536 against 1.48 ms of the Goldfarb-Idnani kernel, which keeps two
// QPs per wavefront).  Here the group keeps the width the coordinates need (two QPs per wavefront at nv = 30) and
// the dense rows ride along in a second role of the first MD lanes:
//   * coordinate role of lane m < NV: row m of the tableau, T[m][0 .. NV + MD) in registers -- the entries against the
//     dense rows included (columns NV + d);
//   * dense role of lane d < MD: what belongs to dense row d alone -- its slack / multiplier, state, right-hand side,
//     and row d of the dense-dense block D[d][d'] = T[NV + d][NV + d'] (MD registers).
// The tableau is symmetric: the entries T[NV + d][j] of a dense row against the coordinates ARE the column entries
// T[j][NV + d] the coordinate lanes hold -- stored once.  What a dense ROW needs from them (its part of an entering
// column, a product with a vector) is a transposition: lane src hands its MD entries to the lanes 0 .. MD-1 through
// LDS (column of a coordinate), or the coordinate lanes' partial products fold by transpose_reduce (products).
// Everything else -- selection, ratio test, pivot -- runs for both roles in the same instructions' shadow: the
// pivot is  T[m][j] -= t_m c_j  with c_j broadcast from lane j's coordinate role for j < NV and from lane d's dense
// role for j = NV + d, and  D[d][d'] -= t'_d c_{NV + d'}.
//
// The start-up sweeps run on the NV x NV block only; the dense part follows in closed form afterwards
// (T[.][NV + d] = H^-1 g_d, D = -G H^-1 G^T): 2 MD NV broadcast-FMAs instead of MD columns dragged through NV sweeps.
//
// Iteration, certificate, refinement, hand-over: those of ik_sweep.h (same citations: pink/solve_ik.py:206-275).
#pragma once

#include "ik_sweep.h"

// 0: round 5's Goldfarb-Idnani trips from the unconstrained minimum (A/B: profiles/ab_ppm_r06.txt)
#ifndef PINKHIP_SWEEPX_PPM
#define PINKHIP_SWEEPX_PPM 1
#endif

namespace pinkhip {

template <int NV, int MD, int W>
struct SweepXLds {
  static __host__ __device__ constexpr int tri(int i) { return i * (i + 1) / 2; }  // H[i][0..i]
  static constexpr int oC = (NV * (NV + 1) / 2 + 1) & ~1;                            // c, one entry per lane
  static constexpr int GP = W + 2;                                                   // (pitch: see SweepLds)
  static constexpr int oG = oC + W;                                                  // G[d][m] at oG + d GP + m
  static constexpr int oR = oG + MD * GP;                                            // MD entries of one row in transit,
  static constexpr int RB = (MD + 1) & ~1;                                           // ... two buffers used in turn
  static constexpr int stride = oR + 2 * RB;
};

template <int NV, int MD, int W, class Src = HbmTerms>
__device__ __forceinline__ int ik_sweepx_instance(const KernelArgs &a, long long block, Src *terms = nullptr) {
  constexpr int NT = NV + MD;
  static_assert((W == 16 || W == 32 || W == 64) && NV <= W && NV % 2 == 0 && MD >= 1 && MD <= 16 && MD <= W, "dense rows ride in the first MD lanes");
  // principal pivoting from a guessed active set in front of the dual method, as in ik_sweep.h (both roles of a lane take
  // part: `x / -g` of a coordinate and its interval, slack / multiplier of a dense row); the arg-min's 8-bit payload holds
  // lane (5 bits), side, role and "nonbasic"
  constexpr bool PPX = PINKHIP_SWEEP_PPM && PINKHIP_SWEEP_PPM_DENSE && PINKHIP_SWEEPX_PPM && W <= 32 && (!Src::kOnTheFly || PINKHIP_ROLLOUT_PPM_VIRTUAL);
  constexpr bool GUESS = PPX && PINKHIP_SWEEP_PPM_CRASH && !Src::kOnTheFly;  // (the whole-step kernel starts all-free: ik_sweep.h)
  constexpr int G = kWave / W;
  constexpr double INF = INFINITY;
  constexpr double BIG = 1e300;
  using BcT = Bcast<W>;
  using SL = SweepXLds<NV, MD, W>;

  const int lane = lane_id();
  const int g = lane / W, li = lane & (W - 1);
  const int nv = a.nv, md = a.md, n_eq = a.n_eq;
#ifdef PINKHIP_SECTION_CLOCK
  const bool clock_on = (block & 63) == 0;
  unsigned long long clock_prev = __builtin_readcyclecounter();
#endif
  long long b = block * G + g;
  const bool valid = b < a.B;
  if (!valid) b = a.B - 1;  // surplus groups of the last wave redo the last instance, write nothing

  const bool in = li < nv;  // coordinate role live
  const bool dl = li < md;  // dense role live

  // ------------------------------------------------------------------ stack (task.py:145-167, solve_ik.py:54-67)
  double T[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) T[j] = 0.0;
  double ci = 0.0, mu_l = 0.0, dadd = 0.0;
  // (everything the instance reads from HBM is requested before the first row is accumulated: ik_sweep.h)
  double ci_d = 0.0, mu_d = 0.0, lbv = -INF, ubv = INF, hv = 0.0, ginv = 1.0;
  auto others = [&]() {
    if (in) {
      if constexpr (!Src::kOnTheFly) {
        lbv = a.lb[b * (long long)nv + li];
        ubv = a.ub[b * (long long)nv + li];
      }
      dadd = stack_diag_tasks<Src>(a, b, terms, li, ci_d, mu_d);
      if (a.c_extra) ci_d += a.c_extra[b * (long long)nv + li];
    }
    if constexpr (!Src::kOnTheFly) {
      // K = [H G^T; G 0]: coordinate lane m takes G[d][m] into column NV + d (the stacking leaves those alone)
      const double *Gb = a.Gd + b * (long long)md * nv;
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        T[NV + d] = (in && d < md) ? Gb[(long long)d * nv + li] : 0.0;
      });
      if (dl) hv = a.hd[b * (long long)md + li];
    }
  };
  stack_rows_bcast<NV, W, 8, Src, PINKHIP_STACK_DEPTH>(a, b, terms, in, li, T, ci, mu_l, others);
  ci += ci_d;
  mu_l += mu_d;
  double diag = a.damping + group_sum<W>(mu_l);
  if constexpr (Src::kOnTheFly) {
    static_for<0, MD>([&](auto Dc) {
      constexpr int d = decltype(Dc)::value;
      T[NV + d] = (in && d < md) ? terms->dense_col(d) : 0.0;
    });
  }
  // dense role: right-hand side and Euclidean norm of the row (the selection's threshold is relative to it)
  {
    const double n2 = transpose_reduce<W, MD, 0>([&](auto Dc) { return T[NV + decltype(Dc)::value] * T[NV + decltype(Dc)::value]; });
    if (dl) {
      if constexpr (Src::kOnTheFly) hv = terms->dense_h(li);
      ginv = (n2 > 0.0) ? 1.0 / sqrt(n2) : 1.0;
    }
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
  diag += dadd;
  double hii = 1.0;  // H[li][li]
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (j == li) {
      T[j] += in ? diag : 1.0;  // padded coordinates: identity rows, never pivoted
      hii = T[j];
    }
  // the QP as stated, parked for the closing trips (ik_sweep.h): H (lower triangle, packed), c, the columns of G
  double *sm = shared_base() + (long long)g * (a.lds_pitch ? a.lds_pitch : SL::stride);
  wave_sync();
  if (li < NV) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j <= li) sm[SL::tri(li) + j] = T[j];
  }
  sm[SL::oC + li] = ci;
  static_for<0, MD>([&](auto Dc) {
    constexpr int d = decltype(Dc)::value;
    sm[SL::oG + d * SL::GP + li] = (li < NV) ? T[NV + d] : 0.0;
  });
  PINKHIP_TICK(0);  // stacking

  // ------------------------------------------------------------------ sweep in every coordinate: T_cc = -H^-1
  // (the smallest pivot decides afterwards: a group that met a non-positive one sweeps on through whatever that leaves
  // -- infinities, NaN -- and never iterates on it; one v_min per column instead of a compare and three selects)
  int status = STATUS_OPTIMAL;
  double pmin = INF;
  int state = 0;  // coordinate role: 0 = free, 1 = fixed at lb, 2 = fixed at ub
  if constexpr (Src::kOnTheFly) {
    lbv = in ? terms->lb : -INF;
    ubv = in ? terms->ub : INF;
  }
  if constexpr (GUESS) {
    // where it starts: the guessed active set of ik_sweep.h (coordinate i fixed at the bound -c_i / H_ii violates), only the
    // free coordinates swept in
    if (group_first_lane<W>(in && !(hii > 0.0)) < W) status = STATUS_NOT_PD;
    const double xd = -ci * approx_rcp(hii);
    if (in) state = (xd < lbv) ? 1 : ((xd > ubv) ? 2 : 0);
    const unsigned long long fm = wave_ballot(li < NV && state == 0 && in);
    const unsigned gfree = static_cast<unsigned>(fm >> (lane & ~(W - 1)));  // this lane's group
    static_for<0, NV>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      const bool want = ((gfree >> k) & 1) != 0;
      if (lanes_on(want)) {  // (wave.h: the other groups' lanes are switched off for the sweep)
        const BcT xb = bcast_prepare<W>(T[k]);
        const double p = value_bcast<W, k>(xb);
        pmin = min_raw(pmin, want ? p : INF);
        const double rp = fast_rcp(p);
        const double t = T[k] * rp;
        double nt = (li == k) ? rp - 1.0 : -t;
        if (!want) nt = 0.0;
        static_for<0, NV>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j != k) T[j] = fma_bcast<W, j>(T[j], xb, nt);
        });
        if (want) T[k] = (li == k) ? -rp : t;
      }
    });
  } else {
    static_for<0, NV>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      if (k < nv) {  // wave-uniform
        const BcT xb = bcast_prepare<W>(T[k]);
        const double p = value_bcast<W, k>(xb);
        pmin = min_raw(pmin, p);
        const double rp = fast_rcp(p);
        const double t = T[k] * rp;
        const double nt = (li == k) ? rp - 1.0 : -t;
        static_for<0, NV>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j != k) T[j] = fma_bcast<W, j>(T[j], xb, nt);
        });
        T[k] = (li == k) ? -rp : t;
      }
    });
  }
  if (!(pmin > 0.0)) status = STATUS_NOT_PD;
  PINKHIP_TICK(1);  // initial sweeps
  double tdiag = 0.0;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (j == li) tdiag = T[j];

  // x0 = -H^-1 c -- from a guessed active set: v = c on the free coordinates, -bound on the fixed ones; the product is x on a
  // free coordinate and c - g on a fixed one (DESIGN_HISTORY.md C.1)
  double x = 0.0;
  const bool fixed0 = GUESS && state != 0;
  const double xstart_fixed = (state == 1) ? lbv : ubv;
  {
    const BcT cb = bcast_prepare<W>(in ? (fixed0 ? -xstart_fixed : ci) : 0.0);
    double r0 = 0.0, r1 = 0.0;
    static_for<0, NV>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      if constexpr (j % 2 == 0) r0 = fma_bcast<W, j>(r0, cb, T[j]);
      else r1 = fma_bcast<W, j>(r1, cb, T[j]);
    });
    x = in ? r0 + r1 : 0.0;
  }
  const double xpoint = fixed0 ? xstart_fixed : x;  // the start point itself
  if (fixed0) x -= ci;                              // ... and the lane's quantity: -g of a fixed coordinate
  // The dense part of the tableau swept on every coordinate, in closed form:
  //   T[m][NV + d] = (H^-1 g_d)_m = -sum_j T[m][j] G[d][j],   D[d][d'] = -g_d^T H^-1 g_d' = -sum_m G[d][m] T[m][NV + d'],
  // and the slack of row d at x0: h_d - g_d x0.
  // (swept on the free coordinates F only, DESIGN_HISTORY.md C.3: the sums run over F, a fixed coordinate keeps its entry
  // G[d][m] in front of its sum)
  double D[MD], ud = 0.0;
  {
    const double gx0 = transpose_reduce<W, MD, 0>([&](auto Dc) { return T[NV + decltype(Dc)::value] * xpoint; });
    if (dl) ud = hv - gx0;
    double Tn[MD];
    static_for<0, MD>([&](auto Dc) {
      constexpr int d = decltype(Dc)::value;
      const BcT gb = bcast_prepare<W>(fixed0 ? 0.0 : T[NV + d]);
      double s0 = 0.0, s1 = 0.0;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (j % 2 == 0) s0 = fma_bcast<W, j>(s0, gb, T[j]);
        else s1 = fma_bcast<W, j>(s1, gb, T[j]);
      });
      Tn[d] = (li < NV) ? (fixed0 ? T[NV + d] : 0.0) - (s0 + s1) : 0.0;
    });
    static_for<0, MD>([&](auto Pc) {
      constexpr int dp = decltype(Pc)::value;
      D[dp] = -transpose_reduce<W, MD, 0>([&](auto Dc) { return (fixed0 ? 0.0 : T[NV + decltype(Dc)::value]) * Tn[dp]; });
    });
    static_for<0, MD>([&](auto Dc) { T[NV + decltype(Dc)::value] = Tn[decltype(Dc)::value]; });
  }
  double ddiag = 0.0;
#pragma unroll
  for (int d = 0; d < MD; ++d)
    if (d == li) ddiag = D[d];
  // n^T H^-1 n of a constraint normal: the reference of the linear-dependence test
  // (from a guessed active set the tableau does not hold it: lower bounds of its size stand in, as in ik_sweep.h)
  double zd0 = -tdiag, zdd0 = -ddiag;
  if constexpr (GUESS) {
    const double hmax = -group_min<W>(in ? -hii : 0.0);
    zd0 = approx_rcp(hii);
    zdd0 = approx_rcp(ginv * ginv * hmax);
  }
  // conditioning estimate (ik_sweep.h): beyond the threshold the group goes to the Goldfarb-Idnani code right away
  {
    const double hii0 = (li < NV && in) ? sm[SL::tri(li < NV ? li : 0) + (li < NV ? li : 0)] : 0.0;
    // (a coordinate the guess fixed: H_ii over its Schur complement T_ii -- H_ii (H^-1)_ii on the free set plus this one)
    const double kest = -group_min<W>((GUESS && state != 0) ? -hii0 * approx_rcp(tdiag) : hii0 * tdiag);
    PINKHIP_TRACEF(li == 0, "[sweepx g%d] kest %.3e\n", g, kest);
    if (status == STATUS_OPTIMAL && !(kest <= PINKHIP_SWEEP_ROUTE_COND)) status = STATUS_ROUTED;
  }
  PINKHIP_TICK(2);  // x0, dense part

  // ------------------------------------------------------------------ dual active set on the tableau
  const KernelArgs *late = &a;
  if constexpr (!Src::kOnTheFly) late = kernarg_reload<KernelArgs>(a);
  const double tol = 1e-13 * (nv > 8 ? nv * 0.125 : 1.0);
  const double thr_lo = -tol * (1.0 + fabs(lbv)), thr_up = -tol * (1.0 + fabs(ubv));
  const double thr_d = -tol * (1.0 + fabs(hv) * ginv);
  const int max_iter = late->max_iter > 0 ? late->max_iter : 20 * (nv + md) + 50;
  const bool empty_box_somewhere = wave_any(in && ubv - lbv < (thr_lo > thr_up ? thr_lo : thr_up));
  // coordinate role (state: above): x; u = multiplier of a fixed coordinate
  double u = 0.0, phi = 0.0, xfree = 1.0;
  // dense role: dstate 0 = inactive (ud = slack h - g x), 1 = active (ud = multiplier); dphi = 1 for an active inequality
  int dstate = 0;
  double dphi = 0.0;
  // group-uniform.  src / kd / pi are GLOBAL indices: < NV a coordinate, NV + d dense row d
  int it = 0, eq_next = 0, src = 0, kind = 0;  // kind 0: lower bound, 1: upper bound, 2: dense row, 3: equality row
  double uplus = 0.0;
  // principal pivoting (ik_sweep.h): the interval of the coordinate role's quantity and the thresholds of its two tests; the
  // dense role's quantity (slack / multiplier) lives in [0, inf) -- dtlo = the slack's threshold while the row is inactive, 0
  // while it is active; an active equality's multiplier has either sign (dlo = -inf)
  double blo = lbv, bhi = ubv, tlo = thr_lo, thi = thr_up;
  const double thr_row = thr_d * fast_rcp(ginv);
  double dlo = 0.0, dtlo = (dl && li >= n_eq) ? thr_row : -INF;
  if constexpr (PPX) {
    if (state != 0) {
      blo = (state == 1) ? -INF : 0.0;
      bhi = (state == 1) ? 0.0 : INF;
      tlo = 0.0;
      thi = 0.0;
    }
    if (empty_box_somewhere) {  // (wave-uniform) an empty box: quadprog's "constraints are inconsistent"
      const bool empty = group_first_lane<W>(in && ubv - lbv < (thr_lo > thr_up ? thr_lo : thr_up)) < W;
      if (empty && status == STATUS_OPTIMAL) status = STATUS_INFEASIBLE;
    }
  }
  bool ppm_mode = PPX, restoring = false;
  bool running = (status == STATUS_OPTIMAL);
  bool need_sel = true;
  bool refined = (status == STATUS_ROUTED);
  int nref = 0;
  double dprev = 0.0;

  // From here on the row lives in register tuples when columns are read by index (ik_sweep.h, TabRegs)
  constexpr bool IDX = PINKHIP_SWEEP_INDEXED_COLUMN && W >= 32;
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
  // the stale copies of the diagonal entries inside the rows, followed in registers (ik_sweep.h)
  double sdiag_run = tdiag, sdd_run = ddiag;
  // row m of the stated H times a vector held one entry per coordinate lane (packed triangle in LDS)
  auto hrow_times = [&](double v) -> double {
    const BcT vb = bcast_prepare<W>(v);
    const int base = SL::tri(li < NV ? li : 0);
    double h0 = 0.0, h1 = 0.0;
    static_for<0, NV>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      const int ad = (j > li) ? SL::tri(j) + (li < NV ? li : 0) : base + j;
      const double hv_ = sm[ad];
      if constexpr (j % 2 == 0) h0 = fma_bcast<W, j>(h0, vb, hv_);
      else h1 = fma_bcast<W, j>(h1, vb, hv_);
    });
    return (li < NV) ? h0 + h1 : 0.0;
  };
  // g_d . v for the stated rows of G (lane d of the dense role gets row d's product; v one entry per coordinate lane)
  auto grow_times = [&](double v) -> double {
    return transpose_reduce<W, MD, 0>([&](auto Dc) { return sm[SL::oG + decltype(Dc)::value * SL::GP + li] * v; });
  };
  // The dense part of row p of the tableau (p a coordinate, group-uniform; -1: none), handed to the dense role: lane p
  // writes its entries T[p][NV + d], lane d reads entry d -- T[NV + d][p] by symmetry.
  // (two buffers in turn: the next hand-over writes the other one, so one barrier per hand-over orders everything)
  int rowsel = 0;
  auto dense_part_of_row = [&](int p) -> double {
    double *rowbuf = sm + SL::oR + rowsel;
    rowsel ^= SL::RB;
    if (li == p) {
      Pair *dst = reinterpret_cast<Pair *>(__builtin_assume_aligned(rowbuf, 16));
      static_for<0, MD / 2>([&](auto Hc) {
        constexpr int d = 2 * decltype(Hc)::value;
        dst[d >> 1] = Pair{tget(std::integral_constant<int, NV + d>{}), tget(std::integral_constant<int, NV + d + 1>{})};
      });
      if constexpr (MD % 2) rowbuf[MD - 1] = tget(std::integral_constant<int, NV + MD - 1>{});
    }
    wave_sync();
    const double v = rowbuf[li < MD ? li : 0];
    return (p >= 0 && li < MD) ? v : 0.0;
  };
  // Column idx of the tableau (group-uniform global index, -1: none): `colc` = the coordinate role's entry T[m][idx],
  // `cold` = the dense role's entry T[NV + d][idx].
  auto column_of = [&](int idx, double &colc, double &cold) {
    const int pc = (idx >= 0 && idx < NV) ? idx : -1;
    const int pd = (idx >= NV) ? idx - NV : -1;
    // (the hand-over through LDS first: its round trip runs under the broadcast-FMAs below)
    double tr = 0.0;
    if (wave_any(pc >= 0)) tr = dense_part_of_row(pc);
    double dx = 0.0;
    if constexpr (IDX) {
      // (entry idx of this lane's row, coordinate or dense column alike: one indexed read per group)
      colc = R.template at_group_uniform<W>(idx);
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        dx = (pd == d) ? D[d] : dx;
      });
    } else {
      const BcT eb = bcast_indicator<W>(pc);
      double c0 = 0.0, c1 = 0.0, cx = 0.0;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (j % 2 == 0) c0 = fma_bcast<W, j>(c0, eb, T[j]);
        else c1 = fma_bcast<W, j>(c1, eb, T[j]);
      });
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        cx = (pd == d) ? T[NV + d] : cx;
        dx = (pd == d) ? D[d] : dx;
      });
      colc = (c0 + c1) + cx;
    }
    cold = dx + tr;
  };

  for (;;) {
    // (a') principal pivoting: the index whose complementarity condition fails with the largest weight, over both roles
    double viol = 0.0, viold = 0.0;
    int nb_src = 0;     // the exchanged index is nonbasic (enters the basis)
    bool iseq = false;  // ... is the next equality
    if constexpr (PPX) {
      if (wave_any(running && ppm_mode)) {
        const bool sel = running && ppm_mode;
        const double slo = x - blo, sup = bhi - x;
        const bool okc = in && (!restoring || state != 0);
        const bool vlo = okc && slo < tlo, vup = okc && sup < thi;
        viol = vlo ? slo : sup;
        const float zf = fabsf(static_cast<float>(tdiag));
        const float wz = (zf > 1e-30f) ? approx_rcpf(zf) : 1e30f;
        const float fv = static_cast<float>(viol);
        float key = -(fv * fv) * wz;
        const bool murty = it > PINKHIP_SWEEP_PPM_MURTY_AFTER(nv + md);
        if (murty) key = static_cast<float>(li - 64);  // least index
        float k32 = (vlo || vup) ? key32_packf(key, li | (vlo ? 0 : 32) | ((state != 0) ? 128 : 0)) : 3.0e38f;
        // dense role: slack of an inactive inequality below its threshold, multiplier of an active one negative
        viold = ud - dlo;
        const bool okd = dl && li >= n_eq && (!restoring || dstate == 1);
        if (okd && viold < dtlo) {
          const float zdf = fabsf(static_cast<float>(ddiag));
          const float wzd = (zdf > 1e-30f) ? approx_rcpf(zdf) : 1e30f;
          const float fu = static_cast<float>(viold);
          float kd = -(fu * fu) * wzd;
          if (murty) kd = static_cast<float>(li - 32);  // (behind the coordinates)
          const float kd32 = key32_packf(kd, li | 64 | ((dstate == 0) ? 128 : 0));
          k32 = (kd32 < k32) ? kd32 : k32;
        }
        const float best32 = group_min32<W>(k32);
        bool conv = false;
        if (sel) {
          if (!restoring && eq_next < n_eq) {
            src = NV + eq_next;  // equalities are activated first, in order
            kind = 0;
            nb_src = 1;
            iseq = true;
          } else if (!(best32 < 0.0f)) {
            if (restoring) {
              ppm_mode = false;  // dual feasible: Goldfarb-Idnani's trips from here, this trip included
              need_sel = true;
            } else {
              running = false;  // optimal
            }
            conv = true;
          } else {
            const int pl = key32_payload(best32);
            src = (pl & 64) ? NV + (pl & 31) : (pl & 31);
            kind = (pl & 64) ? 0 : (pl >> 5) & 1;
            nb_src = pl >> 7;
          }
        }
        if (conv) {
          // ... into the variables of the dual method and of the closing trips
          u = (state == 1) ? -x : ((state == 2) ? x : 0.0);
          phi = (state == 1) ? -1.0 : ((state == 2) ? 1.0 : 0.0);
          xfree = (state != 0) ? 0.0 : 1.0;
          if (state != 0) x = (state == 1) ? lbv : ubv;
          dphi = (dl && dstate == 1 && li >= n_eq) ? 1.0 : 0.0;
        }
      }
    }
    // (a) entering constraint: the violated one that is farthest away in the metric of the objective
    if (wave_any(running && !ppm_mode && need_sel)) {
      const bool sel = running && !ppm_mode && need_sel;
      const double slo = x - lbv, sup = ubv - x;
      const bool vlo = in && slo < thr_lo, vup = in && sup < thr_up;
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
      float k32 = has ? key32_packf(key, id) : 3.0e38f;
      // dense role: an inactive inequality row whose slack is negative
      if (dl && li >= n_eq && dstate == 0 && ud * ginv < thr_d) {
        const float zdf = static_cast<float>(-ddiag);
        const float wzd = (zdf > 1e-30f) ? approx_rcpf(zdf) : 1e30f;
        const float fu = static_cast<float>(ud);
        const float kd32 = key32_packf(-(fu * fu) * wzd, 128 + li);
        k32 = (kd32 < k32) ? kd32 : k32;
      }
      const float best32 = group_min32<W>(k32);
      const bool none = !(best32 < 0.0f);
      bool bad = false;
      if (empty_box_somewhere) bad = group_first_lane<W>((vlo && vup) || (state != 0 && in && (vlo || vup))) < W;
      if (sel) {
        uplus = 0.0;
        if (bad) {
          status = STATUS_INFEASIBLE;
          running = false;
        } else if (eq_next < n_eq) {
          src = NV + eq_next;  // equalities (the first n_eq dense rows) are activated first, in order
          kind = 3;
          need_sel = false;
        } else if (none) {
          running = false;  // optimal
        } else {
          const int pl = key32_payload(best32);
          if (pl & 128) {
            src = NV + (pl & 63);
            kind = 2;
          } else {
            src = pl & 63;
            kind = (pl >> 6) & 1;
          }
          need_sel = false;
        }
      }
    }
    if (running) {
      if (++it > max_iter) {
        status = STATUS_MAX_ITER;
        running = false;
      }
      if constexpr (PPX) {
        if (ppm_mode && it > 2 * PINKHIP_SWEEP_PPM_MURTY_AFTER(nv + md)) restoring = true;  // (the dual method ends what this did not)
      }
    }
    const bool closing = !wave_any(running);
    const bool ref = closing && !refined;
    if (closing && !wave_any(ref)) break;
    const bool act = running;
    PINKHIP_TICK(3);  // selection

    // (b) column src of the tableau; for a finishing group the certificate and the product T r instead
    double col = 0.0, cold = 0.0;
    if (closing) {
      // residual of the KKT system of the final active set, from the problem AS STATED (ik_sweep.h)
      const double kx = hrow_times(in ? x : 0.0);
      const double ci_ = in ? sm[SL::oC + li] : 0.0;
      double grad = in ? kx + ci_ : 0.0;
      const bool arow = dl && dstate == 1;
      {
        const BcT lamb = bcast_prepare<W>(arow ? ud : 0.0);  // + G_A^T lambda_A
        double gl = 0.0;
        static_for<0, MD>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          gl = fma_bcast<W, d>(gl, lamb, sm[SL::oG + d * SL::GP + li]);
        });
        if (in) grad += gl;
      }
      const double gx = grow_times(in ? x : 0.0);
      const double hii = in ? sm[SL::tri(li < NV ? li : 0) + (li < NV ? li : 0)] : 0.0;
      const double gsc = group_min<W>(-hii) * group_min<W>(in ? -fabs(x) : 0.0) - group_min<W>(-fabs(ci_));
      const double gtol = PINKHIP_SWEEP_CERT_TOL * gsc;
      bool fails = false;  // (written so that a NaN fails)
      if (in) {
        if (state == 0) fails = !(fabs(grad) <= gtol && x - lbv >= 10.0 * thr_lo && ubv - x >= 10.0 * thr_up);
        else fails = (state == 1) ? !(grad >= -gtol) : !(grad <= gtol);
      }
      double r = (in && state == 0) ? grad : 0.0, rd = 0.0;
      bool failsd = false;
      if (dl) {
        const double slack = (hv - gx) * ginv;
        if (arow) {
          rd = gx - hv;  // residual of an active row
          // (and its multiplier has the sign of an active inequality: ik_sweep.h)
          failsd = !(fabs(slack) <= -10.0 * thr_d) || (li >= n_eq && !(ud >= -gtol * ginv));
        } else if (li >= n_eq) {
          failsd = !(slack >= 10.0 * thr_d);
        }
      }
      const bool cert_fails = group_first_lane<W>(fails || failsd) < W;
      // (the conditioning estimate again, on the FINAL free set: ik_sweep.h)
      bool illc = false;
      if constexpr (GUESS) {
        const double kfin = -group_min<W>((in && state == 0) ? hii * tdiag : 0.0);
        illc = !(kfin <= PINKHIP_SWEEP_ROUTE_COND);
      }
      if (!ref || status != STATUS_OPTIMAL) r = 0.0, rd = 0.0;
      // (x_F, lambda_A) += T_BB r: the coordinate role's entry, then the dense role's
      const double sdiag = sdiag_run, sdd = sdd_run;
      const BcT rb = bcast_prepare<W>(r), rdb = bcast_prepare<W>(rd);
      double p0 = 0.0, p1 = 0.0;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (j % 2 == 0) p0 = fma_bcast<W, j>(p0, rb, tget(Jc));
        else p1 = fma_bcast<W, j>(p1, rb, tget(Jc));
      });
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        p0 = fma_bcast<W, d>(p0, rdb, tget(std::integral_constant<int, NV + d>{}));
      });
      double q0 = transpose_reduce<W, MD, 0>([&](auto Dc) { return tget(std::integral_constant<int, NV + decltype(Dc)::value>{}) * r; });
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        q0 = fma_bcast<W, d>(q0, rdb, D[d]);
      });
      const double dxc = (ref && in && state == 0) ? (p0 + p1) + (tdiag - sdiag) * r : 0.0;
      const double dxd = (ref && arow) ? q0 + (ddiag - sdd) * rd : 0.0;
      const double xmax = -group_min<W>(in ? -fabs(x) : 0.0);
      const bool more = group_first_lane<W>(fabs(dxc) > 1e-9 * fabs(x) + 1e-11 * (1.0 + xmax)) < W;  // (ik_sweep.h: relative floor)
      const double dmax = -group_min<W>(-fabs(dxc));
      const bool sane = dmax <= ((nref == 0) ? 0.1 * xmax : 0.5 * dprev);
      PINKHIP_TRACEF(li == 0 && ref, "[sweepx g%d it%d nref%d] closing: cert_fails %d more %d sane %d dmax %.3e xmax %.3e dprev %.3e\n", g, it, nref, (int)cert_fails,
                     (int)more, (int)sane, dmax, xmax, dprev);
      if (ref) {
        if (status != STATUS_OPTIMAL || illc) {
          status = STATUS_BREAKDOWN;  // a verdict reached on the tableau is confirmed by the Goldfarb-Idnani code
          refined = true;
        } else if (!cert_fails && !more) {
          if (sane) x += dxc;
          refined = true;
        } else if (sane && nref < 3) {
          x += dxc;
          ud += dxd;
          dprev = dmax;
          ++nref;
        } else {
          status = STATUS_BREAKDOWN;
          refined = true;
        }
      }
      if (!wave_any(!refined)) break;
    } else {
      column_of(act ? src : -1, col, cold);
    }
    // the lane that holds the entering constraint -- lane src in its coordinate role or lane src - NV in its dense
    // role: the group knows which, so every lane offers the value of THAT role and one broadcast fetches it
    const bool sdense = src >= NV;
    const int sl = sdense ? src - NV : src;
    if (li == src && li < NV) col = tdiag;
    if (li == src - NV) cold = ddiag;
    int pi = -1;
    double pvt = 1.0, rp = 0.0, sg = -1.0;  // (no pivot in this group: t = 0 leaves the tableau as it is)
    if constexpr (PPX) {
      if (wave_any(act && ppm_mode)) {
        const bool actP = act && ppm_mode;
        // (c') the exchange (ik_sweep.h): the quantity of the exchanged role goes to zero along the column, every lane's
        // quantities move by -col nu
        const double vs = group_bcast<W>(sdense ? viold : viol, sl);
        const double num = kind ? vs : -vs;
        const double pv = group_bcast<W>(sdense ? ddiag : tdiag, sl);
        const double zr = group_bcast<W>(sdense ? zdd0 : zd0, sl);
        const double rpv = fast_rcp(pv);
        const double sgp = nb_src ? 1.0 : -1.0;
        // a coordinate that is fixed / a row that is activated pivots on MINUS the curvature left along its normal (next to
        // nothing left: the normal depends on the active ones); a coordinate freed / a row released on a reciprocal
        const bool recip = (!sdense) == (nb_src != 0);
        const bool irregular = actP && !((recip ? pv : -pv) > (recip ? 0.0 : PINKHIP_SWEEP_PPM_MIN_CURV * zr));
        double hs = 0.0;
        if (wave_any(irregular && iseq)) hs = group_bcast<W>(hv, sl);
        bool act2 = actP;
        if (irregular) {
          act2 = false;
          if (iseq && fabs(num) <= 1e-9 * (1.0 + fabs(hs))) {
            ++eq_next;  // implied by the active ones and met: nothing to add
          } else if (!restoring) {
            restoring = true;
          } else {
            status = STATUS_BREAKDOWN;  // (not even a release is regular: the tableau is no inverse any more)
            running = false;
          }
        }
        const double nu = act2 ? -num * rpv : 0.0;
        PINKHIP_TRACEF(li == 0 && actP, "[sweepx-ppm g%d it%d] src %d kind %d nonbasic %d pv %.3e zref %.3e num %.3e nu %.3e irregular %d restoring %d\n", g, it, src, kind,
                       nb_src, pv, zr, num, nu, (int)irregular, (int)restoring);
        x = fma(-col, nu, x);
        ud = fma(-cold, nu, ud);
        if (act2) {
          pi = src;
          pvt = pv;
          rp = rpv;
          sg = sgp;
          if (li == src && li < NV) {
            if (state != 0) {
              x = ((state == 1) ? lbv : ubv) + nu;  // off its bound, to where its gradient entry is zero
              state = 0;
              blo = lbv;
              bhi = ubv;
              tlo = thr_lo;
              thi = thr_up;
            } else {
              x = -nu;  // onto the bound it violates: -g
              state = kind + 1;
              blo = kind ? 0.0 : -INF;
              bhi = kind ? INF : 0.0;
              tlo = 0.0;
              thi = 0.0;
            }
          }
          if (li == src - NV) {
            if (dstate == 0) {
              ud = nu;  // the multiplier of the row
              dstate = 1;
              dtlo = 0.0;
              if (li < n_eq) dlo = -INF;  // (an equality's: of either sign)
            } else {
              ud = -nu;  // the slack it opens
              dstate = 0;
              dtlo = thr_row;
            }
          }
          if (iseq) ++eq_next;
        }
      }
    }
    if (!PPX || wave_any(act && !ppm_mode)) {
    const bool actG = act && !ppm_mode;
    // what has to go to zero: the distance of the entering coordinate to its bound resp. the (negative) slack of the
    // entering row; pv = T[src][src] = -n^T Z n
    const double num = group_bcast<W>(sdense ? -ud : (kind == 0 ? lbv : ubv) - x, sl);
    double pv = group_bcast<W>(sdense ? ddiag : tdiag, sl);
    const double z0 = group_bcast<W>(sdense ? zdd0 : zd0, sl);
    bool lin_dep = false;
    {
      // little curvature left along the entering normal: formed again as a sum of squares w^T H w (ik_sweep.h)
      const bool little = actG && !(-pv * 1e6 > z0);
      if (wave_any(little)) {
        const double w = (in && state == 0) ? col : 0.0;
        const double hw = hrow_times(w);
        const double curv = group_sum<W>((li < NV) ? w * hw : 0.0);
        if (little) pv = -curv;
      }
      lin_dep = !(-pv * 1e12 > z0);
    }
    PINKHIP_TICK(4);  // column
    // (c) step
    const double rpv = fast_rcp(pv);
    const double sgn = (num >= 0.0) ? 1.0 : -1.0;
    const double full = lin_dep ? BIG : -fabs(num) * rpv;
    const double rate = phi * col * sgn, rated = dphi * cold * sgn;
    const bool blocking = actG && rate > 0.0, blockd = actG && rated > 0.0;
    const double ratio = blocking ? max_raw(u, 0.0) * fast_rcp1(rate) : BIG;
    const double ratiod = blockd ? max_raw(ud, 0.0) * fast_rcp1(rated) : BIG;
    const double k1 = group_min<W>((ratiod < ratio) ? ratiod : ratio);
    const int kdc = group_first_lane<W>(blocking && ratio == k1);
    const int kdd = group_first_lane<W>(blockd && ratiod == k1);
    const int kd = (kdc < W) ? kdc : NV + (kdd & (W - 1));
    const double tstep = (k1 < full) ? k1 : full;
    const bool stuck = !(tstep < BIG);
    double hs = 0.0;
    if (wave_any(actG && stuck)) hs = group_bcast<W>(sdense ? hv : (kind == 0 ? lbv : ubv), sl);
    if (actG && stuck) {
      const bool tiny = fabs(num) <= 1e-9 * (1.0 + fabs(hs));
      if (kind == 3 && tiny) {
        ++eq_next;  // equality implied by the active ones and already satisfied: nothing to add
        need_sel = true;
      } else if (tiny) {
        // a dependent inequality that no drop can help, violated by round-off only: the bound moves to the point
        if (li == src && li < NV) {
          if (kind == 0) lbv = x;
          else ubv = x;
        }
        if (li == src - NV) hv -= ud, ud = 0.0;
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
      const double dc = col * nu;
      x = fma(-xfree, dc, x);
      u = fma(-phi, dc, u);
      ud = fma(-cold, nu, ud);  // slack or multiplier of a row: both move by -col nu
      uplus += (kind == 3) ? nu : fabs(nu);
    }
    PINKHIP_TICK(5);  // step lengths, x / u update
    // (d) pivot: on src (the entering constraint becomes tight) or on kd (the blocking constraint leaves)
    if (do_add) {
      pi = src;
      pvt = pv;
      rp = rpv;
      if (li == src && li < NV) {
        state = kind + 1;
        x = (kind == 0) ? lbv : ubv;
        phi = (kind == 0) ? -1.0 : 1.0;
        xfree = 0.0;
        u = uplus;
      }
      if (li == src - NV) {
        dstate = 1;
        dphi = (li >= n_eq) ? 1.0 : 0.0;  // equalities never leave
        ud = uplus;
      }
      if (kind == 3) ++eq_next;
      need_sel = true;
    }
    if (wave_any(do_drop)) {
      double ck, ckd;
      column_of(do_drop ? kd : -1, ck, ckd);
      const double pk = group_bcast<W>(kd >= NV ? ddiag : tdiag, kd >= NV ? kd - NV : kd);
      if (do_drop) {
        col = (li == kd && li < NV) ? tdiag : ck;
        cold = (li == kd - NV) ? ddiag : ckd;
        pi = kd;
        pvt = pk;
        rp = fast_rcp(pk);
        if (li == kd && li < NV) {
          state = 0;
          u = 0.0;
          phi = 0.0;
          xfree = 1.0;
        }
        if (li == kd - NV) {
          dstate = 0;
          ud = 0.0;
          dphi = 0.0;
        }
      }
    }
    // sweep (nonbasic -> basic: sg = +1) or reverse sweep (sg = -1) on pi.  Basic = free coordinate / active row.
    if (!ppm_mode) sg = ((pi < NV) == do_add) ? -1.0 : 1.0;
    }
    PINKHIP_TICK(6);  // column of the leaving constraint
    {
      double t = col * rp, cp = col, td = cold * rp, cpd = cold;
      if (li == pi && li < NV) {
        t = 1.0 - sg * rp;
        cp = pvt - sg;
      }
      if (li == pi - NV) {
        td = 1.0 - sg * rp;
        cpd = pvt - sg;
      }
      const BcT xb = bcast_prepare<W>(cp), xbd = bcast_prepare<W>(cpd);
      const double nt = -t, ntd = -td;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        tset(Jc, fma_bcast<W, j>(tget(Jc), xb, nt));
      });
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        constexpr std::integral_constant<int, NV + d> Jd{};
        tset(Jd, fma_bcast<W, d>(tget(Jd), xbd, nt));
        D[d] = fma_bcast<W, d>(D[d], xbd, ntd);
      });
      // (what the FMAs above just made of register li of this lane's row resp. of D[li]: the stale diagonal copies)
      sdiag_run = fma(cp, nt, sdiag_run);
      sdd_run = fma(cpd, ntd, sdd_run);
      tdiag = (li == pi && li < NV) ? -rp : tdiag - t * col;
      ddiag = (li == pi - NV) ? -rp : ddiag - td * cold;
    }
    PINKHIP_TICK(7);  // pivot
  }
  PINKHIP_TICK(8);  // exit
  // ------------------------------------------------------------------ write-out
  if constexpr (Src::kOnTheFly) {
    terms->x = in ? x : 0.0;
    terms->status = status;
  }
  {
    int ln = lane;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(ln));
#endif
    const long long bw = block * G + ln / W;
    if (bw < late->B) {
      if (in) late->dq[bw * (long long)nv + li] = x * late->out_scale;
      if (li == 0) {
        late->status[bw] = status;
        if (late->iters) late->iters[bw] = it;
      }
    }
  }
  return status;
}

// doubles of LDS per QP: the parking area or, for a group that is handed over, the Goldfarb-Idnani working set
template <int NV, int MD, int W>
__host__ __device__ constexpr int sweepx_kernel_lds_doubles(int md) {
  return SweepXLds<NV, MD, W>::stride > LdsP<NV>::stride(md) ? SweepXLds<NV, MD, W>::stride : LdsP<NV>::stride(md);
}

template <int NV, int MD, int W>
__device__ __forceinline__ void ik_solve_sweepx_body(const KernelArgs &a, long long block) {
  const int st = ik_sweepx_instance<NV, MD, W>(a, block);
  const bool over = st == STATUS_BREAKDOWN || st == STATUS_ROUTED;
  if (wave_any(over)) {
    wave_sync();
    const KernelArgs *again = kernarg_reload<KernelArgs>(a);
    long long blk = block;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(blk));
#endif
    ik_packed_instance<NV, W, true>(*again, blk, static_cast<HbmTerms *>(nullptr), over, st == STATUS_ROUTED ? PATH_ROUTED : PATH_HANDOVER);
  }
}

template <int NV, int MD, int W>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_SWEEPX(NV, MD) ik_solve_sweepx_kernel(KernelArgs a) {
  ik_solve_sweepx_body<NV, MD, W>(a, block_id());
}

}  // namespace pinkhip
