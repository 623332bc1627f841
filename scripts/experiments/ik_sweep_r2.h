// The sweep-tableau stack + solve kernel of ik_sweep.h in a second lane layout: groups of SIXTEEN lanes, TWO tableau
// rows per lane (rows l and l + 16), four QPs per wavefront for NT = NV + MD <= 32.
//
// Why.  ik_sweep.h is VALU-throughput bound and more than half of its instructions are not fp64
// arithmetic: per trip ~60 broadcast-FMAs stand against ~190 instructions of selection, reductions, step lengths and
// pivot set-up, all of which serve the TWO QPs of a wavefront.  With two rows per lane the same instruction stream
// serves FOUR: the FMAs double (two rows per lane: 2 NT per pivot), the per-row arithmetic doubles, but everything that
// is per QP -- decoding the entering constraint, step lengths, reciprocals, the butterfly steps of the reductions, the
// ballots -- stays, and the broadcast needs no permlane swaps at all (a group IS a DPP row of sixteen lanes: entry j of a
// group-uniform vector lives in lane j % 16, slot j / 16).  Two independent rows per lane also double the
// instruction-level parallelism of a wave, which is what lets the kernel run at two waves per SIMD (2 x NT doubles of
// tableau per lane) without being latency bound.
//
// Same arithmetic, same logic, same citations as ik_sweep.h (pink/solve_ik.py:206-275, pink/tasks/task.py:145-167); the
// comments there explain the tableau, the pivot on a run-time index, the step and the closing refinement.  HBM terms only
// (the whole-step kernel keeps one row per lane: its kinematics map a joint / a tangent column to a lane).
#pragma once

#include "ik_common.h"
#include "ik_stack_rows.h"
#include "ik_sweep.h"

namespace pinkhip {

template <int NV, int MD>
__device__ inline void ik_sweep_r2_instance(const KernelArgs &a, long long block) {
  constexpr int NT = NV + MD, LW = 16, R = 2, G = kWave / LW;
  static_assert(NT <= LW * R && NV % 2 == 0 && MD >= 0, "two rows per lane, sixteen lanes per QP");
  constexpr bool DENSE = MD > 0;
  constexpr double INF = INFINITY;
  constexpr double BIG = 1e300;
  using Bc = Bcast<16>;
  using SL = SweepLds<NV, MD, 32>;  // same per-QP layout as the 32-lane groups: one entry per ROW where that says "lane"

  const int lane = lane_id();
  const int g = lane >> 4, l16 = lane & 15;
  const int nv = a.nv, md = DENSE ? a.md : 0, n_eq = DENSE ? a.n_eq : 0;
  long long b = block * G + g;
  const bool valid = b < a.B;
  if (!valid) b = a.B - 1;  // surplus groups of the last wave redo the last instance, write nothing

  int row[R];
  bool in[R], dlane[R];
  int dr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = l16 + 16 * r;
    in[r] = row[r] < nv;
    dr[r] = row[r] - NV;
    dlane[r] = DENSE && dr[r] >= 0 && dr[r] < md;
  }

  // broadcast of a per-row quantity: slot s of the result holds the vector's entries 16 s .. 16 s + 15
  struct Vec {
    Bc s[R];
  };
  auto prepare = [&](const double (&v)[R]) {
    Vec o;
#pragma unroll
    for (int r = 0; r < R; ++r) o.s[r] = bcast_prepare<16>(v[r]);
    return o;
  };
  // acc + (entry J of the vector) * x
#define PINKHIP_R2_FMA(J, acc, vec, x) fma_bcast<16, (J) % 16>(acc, (vec).s[(J) / 16], x)

  // ------------------------------------------------------------------ stack (task.py:145-167, solve_ik.py:54-67)
  double T[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < NT; ++j) T[r][j] = 0.0;
  double ci[R] = {0.0, 0.0}, mu_l = 0.0;
  {
    const int Kd = a.Kd, K = a.K;
    const double *Jb = a.J + b * (long long)Kd * nv;
    const double *eb = a.e + b * (long long)K;
    const double *costb = a.cost_batched ? a.cost + b * (long long)K : a.cost;
    constexpr int RC = 8;  // rows per chunk: lane k < RC holds the weights of row k (slot 0)
    double cur[RC][R], nxt[RC][R];
    double pw = 0.0, pe = 0.0, pg = 0.0, pl = 0.0;
    auto request = [&](double (&dst)[RC][R], int r0, int rc) {
#pragma unroll
      for (int kk = 0; kk < RC; ++kk)
#pragma unroll
        for (int r = 0; r < R; ++r) dst[kk][r] = (in[r] && kk < rc) ? Jb[(long long)(r0 + kk) * nv + row[r]] : 0.0;
      if (l16 < rc) {
        const int k = r0 + l16;
        pw = costb[k];
        pe = eb[k];
        pg = a.row_gain[k];
        pl = a.row_lm[k];
      }
    };
    if (Kd > 0) request(cur, 0, Kd < RC ? Kd : RC);
    for (int r0 = 0; r0 < Kd; r0 += RC) {
      const int rc = (Kd - r0 < RC) ? Kd - r0 : RC;
      const double wa = (l16 < rc) ? pw * pw : 0.0;
      const double gw = (l16 < rc) ? pg * wa * pe : 0.0;
      if (l16 < rc) mu_l += pl * (pg * pg) * wa * pe * pe;
      const Bc wab = bcast_prepare<16>(wa), gwb = bcast_prepare<16>(gw);
      if (r0 + RC < Kd) request(nxt, r0 + RC, (Kd - r0 - RC < RC) ? Kd - r0 - RC : RC);
      static_for<0, RC>([&](auto Kc) {
        constexpr int kk = decltype(Kc)::value;
        if (kk < rc) {  // wave-uniform
          const Vec rowb = prepare(cur[kk]);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double aa = fma_bcast<16, kk>(0.0, wab, cur[kk][r]);
            ci[r] = fma_bcast<16, kk>(ci[r], gwb, cur[kk][r]);
            static_for<0, NV>([&](auto Jc) {
              constexpr int j = decltype(Jc)::value;
              T[r][j] = PINKHIP_R2_FMA(j, T[r][j], rowb, aa);
            });
          }
        }
      });
#pragma unroll
      for (int kk = 0; kk < RC; ++kk)
#pragma unroll
        for (int r = 0; r < R; ++r) cur[kk][r] = nxt[kk][r];
    }
  }
  double diag[R];
  {
    double dadd[R] = {0.0, 0.0};
    HbmTerms *none = nullptr;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (in[r]) {
        dadd[r] = stack_diag_tasks<HbmTerms>(a, b, none, row[r], ci[r], mu_l);
        if (a.c_extra) ci[r] += a.c_extra[b * (long long)nv + row[r]];
      }
    }
    const double shared = a.damping + group_sum<16>(mu_l);
#pragma unroll
    for (int r = 0; r < R; ++r) diag[r] = shared;
    double hv_unused = 0.0;
    (void)hv_unused;
    if constexpr (DENSE) {
      if (md > 0) {
        for (int t = 0; t < a.n_barriers; ++t) {
          const double rr = a.barrier_safe_gain[t];
          if (rr > 1e-6) {
            const int q0 = a.barrier_rows[t], q1 = a.barrier_rows[t + 1];
            const double *Gb = a.Gd + b * (long long)md * nv;
            double s = 0.0;
            for (int d = q0; d < q1; ++d)
#pragma unroll
              for (int r = 0; r < R; ++r)
                if (in[r]) {
                  const double v = Gb[(long long)d * nv + row[r]];
                  s += v * v;
                }
            s = group_sum<16>(s);
#pragma unroll
            for (int r = 0; r < R; ++r) diag[r] += rr / (s * a.dt * a.dt);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) diag[r] += dadd[r];
  }
  double hv[R] = {0.0, 0.0}, ginv[R] = {1.0, 1.0};
  if constexpr (DENSE) {
    if (md > 0) {
      const double *Gb = a.Gd + b * (long long)md * nv;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        static_for<0, MD>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          T[r][NV + d] = (in[r] && d < md) ? Gb[(long long)d * nv + row[r]] : 0.0;
        });
        if (dlane[r]) {
          const double *gr = Gb + (long long)dr[r] * nv;
          double n2 = 0.0;
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            T[r][j] = (j < nv) ? gr[j] : 0.0;
            n2 += T[r][j] * T[r][j];
          }
          hv[r] = a.hd[b * (long long)md + dr[r]];
          ginv[r] = (n2 > 0.0) ? 1.0 / sqrt(n2) : 1.0;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j == row[r]) T[r][j] += in[r] ? diag[r] : 1.0;  // padded coordinates: identity rows, never pivoted

  // the stated problem, parked for the closing refinement step (ik_sweep.h)
  double *sm = shared_base() + (long long)g * SL::stride;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row[r] < NV) {
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j <= row[r]) sm[SL::tri(row[r]) + j] = T[r][j];
    }
    sm[SL::oC + row[r]] = ci[r];
    if constexpr (DENSE) {
      static_for<0, MD>([&](auto Dc) {
        constexpr int d = decltype(Dc)::value;
        sm[SL::oG + d * 32 + row[r]] = (row[r] < NV) ? T[r][NV + d] : 0.0;
      });
    }
  }

  // ------------------------------------------------------------------ sweep in every coordinate
  int status = STATUS_OPTIMAL;
  static_for<0, NV>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value;
    constexpr int ks = k / 16, kl = k % 16;  // slot and lane of row k
    if (k < nv) {  // wave-uniform
      double colk[R];
#pragma unroll
      for (int r = 0; r < R; ++r) colk[r] = T[r][k];
      const Vec xb = prepare(colk);
      double p = fma_bcast<16, kl>(0.0, xb.s[ks], 1.0);
      if (!(p > 0.0)) {
        status = STATUS_NOT_PD;
        p = 1.0;
      }
      const double rp = fast_rcp(p);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool own = (r == ks) && (l16 == kl);
        const double t = T[r][k] * rp;
        const double nt = own ? rp - 1.0 : -t;
        static_for<0, NT>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j != k) T[r][j] = PINKHIP_R2_FMA(j, T[r][j], xb, nt);
        });
        T[r][k] = own ? -rp : t;
      }
    }
  });
  double tdiag[R] = {0.0, 0.0};
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      if (j == row[r]) tdiag[r] = T[r][j];
  double x[R] = {0.0, 0.0}, u[R] = {0.0, 0.0};
  {
    double cv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cv[r] = in[r] ? ci[r] : 0.0;
    const Vec cb = prepare(cv);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double r0 = 0.0, r1 = 0.0;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (j % 2 == 0) r0 = PINKHIP_R2_FMA(j, r0, cb, T[r][j]);
        else r1 = PINKHIP_R2_FMA(j, r1, cb, T[r][j]);
      });
      x[r] = in[r] ? r0 + r1 : 0.0;
      if (dlane[r]) u[r] = hv[r] + (r0 + r1);
    }
  }
  double zd0[R];
#pragma unroll
  for (int r = 0; r < R; ++r) zd0[r] = -tdiag[r];

  // ------------------------------------------------------------------ dual active set on the tableau
  const KernelArgs *late = kernarg_reload<KernelArgs>(a);
  double lbv[R], ubv[R], thr_lo[R], thr_up[R], thr_d[R];
  const double tol = 1e-13 * (nv > 8 ? nv * 0.125 : 1.0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    lbv[r] = in[r] ? late->lb[b * (long long)nv + row[r]] : -INF;
    ubv[r] = in[r] ? late->ub[b * (long long)nv + row[r]] : INF;
    thr_lo[r] = -tol * (1.0 + fabs(lbv[r]));
    thr_up[r] = -tol * (1.0 + fabs(ubv[r]));
    thr_d[r] = -tol * (1.0 + fabs(hv[r]) * ginv[r]);
  }
  const int max_iter = late->max_iter > 0 ? late->max_iter : 20 * (nv + md) + 50;
  int state[R] = {0, 0};
  int it = 0, eq_next = 0, src = 0, kind = 0;  // group-uniform
  double uplus = 0.0;
  bool running = (status == STATUS_OPTIMAL);
  bool need_sel = true;
  bool refined = false;
  int nref = 0;

  // value of row idx (group-uniform) of a per-row quantity, in every lane of the group
  auto row_bcast = [&](const double (&v)[R], int idx) {
    const double sel = (idx & 16) ? v[1] : v[0];
    return lane_shfl(sel, (lane & ~15) | (idx & 15));
  };

  // residual of the KKT system of the final active set, from the problem as stated (parked in LDS)
  auto residual = [&](double (&rres)[R]) {
    double xl[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xl[r] = in[r] ? x[r] : 0.0;
    const Vec xb = prepare(xl);
    double lam[R];
#pragma unroll
    for (int r = 0; r < R; ++r) lam[r] = (dlane[r] && state[r] == 1) ? u[r] : 0.0;
    Vec lamb;
    if constexpr (DENSE) lamb = prepare(lam);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int base = (row[r] < NV) ? SL::tri(row[r]) : SL::oG + (dlane[r] ? dr[r] : 0) * 32;
      double h0 = 0.0, h1 = 0.0;
      static_for<0, NV>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        const int ad = (row[r] < NV && j > row[r]) ? SL::tri(j) + row[r] : base + j;
        const double hv_ = sm[ad];
        if constexpr (j % 2 == 0) h0 = PINKHIP_R2_FMA(j, h0, xb, hv_);
        else h1 = PINKHIP_R2_FMA(j, h1, xb, hv_);
      });
      double rr = in[r] ? (h0 + h1) + sm[SL::oC + row[r]] : 0.0;
      if (state[r] != 0) rr = 0.0;
      if constexpr (DENSE) {
        if (md > 0) {
          double gl = 0.0;
          static_for<0, MD>([&](auto Dc) {
            constexpr int d = decltype(Dc)::value;
            gl = PINKHIP_R2_FMA(NV + d, gl, lamb, sm[SL::oG + d * 32 + (row[r] < 32 ? row[r] : 0)]);
          });
          if (in[r] && state[r] == 0) rr += gl;
          if (dlane[r] && state[r] == 1) rr = (h0 + h1) - hv[r];
        }
      }
      rres[r] = rr;
    }
  };

  for (;;) {
    // (a) entering constraint
    if (wave_any(running && need_sel)) {
      const bool sel = running && need_sel;
      double key = 0.0;
      int id = 0;
      bool conflict = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double slo = x[r] - lbv[r], sup = ubv[r] - x[r];
        const bool vlo = in[r] && slo < thr_lo[r], vup = in[r] && sup < thr_up[r];
        conflict = conflict || (vlo && vup) || (state[r] != 0 && in[r] && (vlo || vup));
        const double zd = -tdiag[r];
        const double wz = (zd > 1e-290) ? approx_rcp(zd) : 1e290;
        if (vlo && state[r] == 0) {
          const double kl = -(slo * slo) * wz - 1e-300;
          if (kl < key) key = kl, id = row[r];
        }
        if (vup && state[r] == 0) {
          const double ku = -(sup * sup) * wz - 1e-300;
          if (ku < key) key = ku, id = 64 + row[r];
        }
        if constexpr (DENSE) {
          if (dlane[r] && dr[r] >= n_eq && state[r] == 0 && u[r] * ginv[r] < thr_d[r]) {
            const double kd_ = -(u[r] * u[r]) * wz - 1e-300;
            if (kd_ < key) key = kd_, id = row[r];
          }
        }
      }
      const float best32 = group_min32<16>(key < 0.0 ? key32_pack(key, id) : 3.0e38f);
      const bool none = !(best32 < 0.0f);
      const bool bad = group_first_lane<16>(conflict) < 16;
      if (sel) {
        uplus = 0.0;
        if (bad) {
          status = STATUS_INFEASIBLE;
          running = false;
        } else if (DENSE && eq_next < n_eq) {
          src = NV + eq_next;
          kind = 3;
          need_sel = false;
        } else if (none) {
          running = false;  // optimal
        } else {
          const int pl = key32_payload(best32);
          src = pl & 63;
          kind = (src >= NV) ? 2 : (pl >> 6) & 1;
          need_sel = false;
        }
      }
    }
    if (running) {
      if (++it > max_iter) {
        status = STATUS_MAX_ITER;
        running = false;
      }
    }
    const bool ref = !wave_any(running) && !refined;
    if (!wave_any(running || ref)) break;
    const bool act = running;

    // (b) column src of T (for a finishing group: the product T r of the refinement step)
    double col[R];
    {
      const int p = act ? src : -1;
      double ev[R], rres[R] = {0.0, 0.0}, sdiag[R] = {0.0, 0.0};
#pragma unroll
      for (int r = 0; r < R; ++r) ev[r] = (row[r] == p) ? 1.0 : 0.0;
      if (wave_any(ref)) {
        residual(rres);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!ref || status != STATUS_OPTIMAL) rres[r] = 0.0;
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (j == row[r]) sdiag[r] = T[r][j];
          if (ref) ev[r] = rres[r];
        }
      }
      const Vec eb = prepare(ev);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double c0 = 0.0, c1 = 0.0;
        static_for<0, NT>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j % 2 == 0) c0 = PINKHIP_R2_FMA(j, c0, eb, T[r][j]);
          else c1 = PINKHIP_R2_FMA(j, c1, eb, T[r][j]);
        });
        col[r] = c0 + c1;
      }
      double dxv[R];
      bool big = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        dxv[r] = (ref && in[r] && state[r] == 0) ? col[r] + (tdiag[r] - sdiag[r]) * rres[r] : 0.0;
        big = big || fabs(dxv[r]) > 1e-9 * fabs(x[r]) + 1e-13;
      }
      const bool more = wave_any(ref) && group_first_lane<16>(big) < 16;
      if (ref) {
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] += dxv[r];
        refined = !(more && ++nref < 3);
      }
    }
    if (!wave_any(act || !refined)) break;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (row[r] == src) col[r] = tdiag[r];
    double cand[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cand[r] = (row[r] < NV) ? ((kind == 0 ? lbv[r] : ubv[r]) - x[r]) : -u[r];
    const double num = row_bcast(cand, src);
    double pv = row_bcast(tdiag, src);
    bool lin_dep = false;
    if constexpr (DENSE) {
      const double z0 = row_bcast(zd0, src);
      lin_dep = !(-pv * 1e10 > z0);
    }
    if (!act) pv = -1.0;
    // (c) step
    const double rz = lin_dep ? 0.0 : fast_rcp1(-pv);
    const double sgn = (num >= 0.0) ? 1.0 : -1.0;
    const double full = lin_dep ? INF : fabs(num) * rz;
    double phi[R], ratio[R];
    bool blocking[R];
    double rmin = BIG;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      phi[r] = 0.0;
      if (row[r] < NV) phi[r] = (state[r] == 1) ? -1.0 : (state[r] == 2 ? 1.0 : 0.0);
      else if (DENSE && state[r] == 1) phi[r] = (dr[r] >= n_eq) ? 1.0 : 0.0;
      const double rate = phi[r] * col[r] * sgn;
      blocking[r] = act && rate > 0.0;
      ratio[r] = blocking[r] ? (u[r] > 0.0 ? u[r] * fast_rcp1(rate) : 0.0) : BIG;
      rmin = ratio[r] < rmin ? ratio[r] : rmin;
    }
    const double k1 = group_min<16>(rmin);
    const int f0 = group_first_lane<16>(blocking[0] && ratio[0] == k1);
    const int f1 = group_first_lane<16>(blocking[1] && ratio[1] == k1);
    const int kd = (f0 < 16) ? f0 : 16 + (f1 & 15);
    const double t1 = (k1 < BIG) ? k1 : INF;
    const double tstep = (t1 < full) ? t1 : full;
    double hs = 0.0;
    if constexpr (DENSE) {
      if (wave_any(act && !(tstep < INF))) hs = row_bcast(hv, src);
    }
    if (act && !(tstep < INF)) {
      if (DENSE && kind == 3 && fabs(num) <= 1e-9 * (1.0 + fabs(hs))) {
        ++eq_next;
        need_sel = true;
      } else {
        status = STATUS_INFEASIBLE;
        running = false;
      }
    }
    const bool act2 = act && running && (tstep < INF);
    const bool do_add = act2 && !(t1 < full);
    const bool do_drop = act2 && !do_add;
    if (act2) {
      const double nu = sgn * tstep;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double d = col[r] * nu;
        if (row[r] < NV) {
          if (state[r] == 0) x[r] -= d;
          else u[r] -= phi[r] * d;
        } else {
          u[r] -= d;
        }
      }
      uplus += (kind == 3) ? nu : tstep;
    }
    // (d) pivot on src (add) or on kd (drop)
    int pi = -1;
    double pvt = 1.0;
    if (do_add) {
      pi = src;
      pvt = pv;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (row[r] == src) {
          if (row[r] < NV) {
            state[r] = kind + 1;
            x[r] = (kind == 0) ? lbv[r] : ubv[r];
          } else {
            state[r] = 1;
          }
          u[r] = uplus;
        }
      }
      if (DENSE && kind == 3) ++eq_next;
      need_sel = true;
    }
    if (wave_any(do_drop)) {
      const int p = do_drop ? kd : -1;
      double ev[R];
#pragma unroll
      for (int r = 0; r < R; ++r) ev[r] = (row[r] == p) ? 1.0 : 0.0;
      const Vec eb = prepare(ev);
      double ck[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double c0 = 0.0, c1 = 0.0;
        static_for<0, NT>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j % 2 == 0) c0 = PINKHIP_R2_FMA(j, c0, eb, T[r][j]);
          else c1 = PINKHIP_R2_FMA(j, c1, eb, T[r][j]);
        });
        ck[r] = c0 + c1;
      }
      const double pk = row_bcast(tdiag, kd);
      if (do_drop) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          col[r] = (row[r] == kd) ? tdiag[r] : ck[r];
          if (row[r] == kd) {
            state[r] = 0;
            u[r] = 0.0;
          }
        }
        pi = kd;
        pvt = pk;
      }
    }
    {
      const bool piv = pi >= 0;
      const double sg = ((pi < NV) == do_add) ? -1.0 : 1.0;
      const double rp = fast_rcp(pvt);
      double t[R], cp[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        t[r] = piv ? col[r] * rp : 0.0;
        cp[r] = piv ? col[r] : 0.0;
        if (row[r] == pi) {
          t[r] = 1.0 - sg * rp;
          cp[r] = pvt - sg;
        }
      }
      const Vec xb = prepare(cp);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double nt = -t[r];
        static_for<0, NT>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          T[r][j] = PINKHIP_R2_FMA(j, T[r][j], xb, nt);
        });
        tdiag[r] = (row[r] == pi) ? -rp : tdiag[r] - t[r] * col[r];
      }
    }
  }

  // ------------------------------------------------------------------ write-out
  if (valid) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (in[r]) late->dq[b * (long long)nv + row[r]] = x[r];
    if (l16 == 0) {
      late->status[b] = status;
      if (late->iters) late->iters[b] = it;
    }
  }
#undef PINKHIP_R2_FMA
}

template <int NV, int MD>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_SWEEP_R2(NV + MD) ik_solve_sweep_r2_kernel(KernelArgs a) {
  ik_sweep_r2_instance<NV, MD>(a, block_id());
}

}  // namespace pinkhip
