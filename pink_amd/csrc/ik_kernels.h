// Batched differential-IK on gfx950: one wavefront per QP.
//
// For instance b the wave
//   (0) streams its J/e/bounds block from HBM (coalesced, contiguous per QP),
//   (1) stacks  H = damping I + sum_t J_t^T W_t^2 J_t + mu_t I,  c = sum_t gain_t J_t^T W_t^2 e_t
//       (reference pink/tasks/task.py:145-167, pink/solve_ik.py:54-67) with lane i
//       owning row i of H in registers -- no cross-lane reduction is needed for
//       J^T W J in this mapping, the wave shuffles are used for the scalar
//       reductions (Levenberg-Marquardt mu, norms, argmins),
//   (2) factors H = L L^T, forms J = L^-T and the unconstrained minimum,
//   (3) runs the Goldfarb-Idnani dual active-set iteration (the algorithm behind
//       the reference's solver="quadprog", pink/solve_ik.py:270) on the merged box
//       lb <= dq <= ub plus md dense rows, and
//   (4) writes dq, status, iteration count.
//
// Data placement: lane i holds row i of the current n x n matrix (H, then L, then
// the GI matrix J = L^-T Q) in NV registers with compile-time indices; L (row
// major) and later the triangular factor R of the active normals (column major)
// live in LDS, as do the broadcast vectors.  All control flow is wave-uniform
// (one QP per wave), so the data-dependent active-set iteration never diverges.
//
// The file only uses the primitives of wave.h (lane_id, wave_sync, bcast,
// wave_sum, wave_min, key_pack, fast_rcp/rsqrt, from_next_lane, shared_base); the CPU wave emulator
// under tests/emu provides the same names to run this exact source in tests.
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
constexpr int STATUS_OPTIMAL = 0;
constexpr int STATUS_MAX_ITER = 1;
constexpr int STATUS_INFEASIBLE = 2;
constexpr int STATUS_NOT_PD = 3;

// Everything a launch needs; passed by value in the kernarg segment.
struct KernelArgs {
  long long B;
  int nv, Kd, K, md;
  int n_eq;          // the first n_eq dense rows are equalities Gd dq = hd (packed kernel only)
  int n_dtasks;      // diagonal tasks
  int n_barriers;
  int cost_batched;
  int max_iter;
  double damping, dt;
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

// LDS carve-up (in doubles) for the NV-padded kernels.
template <int NV>
struct Lds {
  static constexpr int NVP = NV + 2;  // row pitch of L / column pitch of R (16-B aligned rows)
  static constexpr int GP = NV + 1;   // row pitch of the dense inequality rows
  static constexpr int oL = 0;                 // NV*NVP  staging of J rows, then L, then R
  static constexpr int oX = oL + NV * NVP;     // NV      x / y / column scratch
  static constexpr int oD = oX + NV;           // NV      d = J^T n+   (also 1/diag(L))
  static constexpr int oD2 = oD + NV;          // NV      d with the first q entries zeroed
  static constexpr int oV = oD2 + NV;          // NV      Householder vector
  static constexpr int oWa = oV + NV;          // 64      w_k^2 of the staged rows
  static constexpr int oGs = oWa + 64;         // 64      gain w_k^2 e_k of the staged rows
  static constexpr int oGd = oGs + 64;         // md*GP   dense inequality rows
  static constexpr int fixed_doubles = oGd;
  static inline long long bytes(int md) { return 8LL * (fixed_doubles + (long long)md * GP + 2); }
};

// Copy a row-major [rows, nv] block from global memory into LDS with row pitch
// `pitch`.  Consecutive lanes read consecutive addresses (512 B per wave load).
__device__ inline void stage_rows(double *dst, int pitch, const double *src, int rows, int nv,
                                  int lane) {
  int r = lane / nv, j = lane - r * nv;
  const int dr = kWave / nv, dj = kWave - dr * nv;
  const int n = rows * nv;
  for (int idx = lane; idx < n; idx += kWave) {
    dst[r * pitch + j] = src[idx];
    r += dr;
    j += dj;
    if (j >= nv) {
      j -= nv;
      ++r;
    }
  }
}

template <int NV, bool SOLVE>
__device__ inline void ik_instance(const KernelArgs &a, long long b) {
  using S = Lds<NV>;
  constexpr int NVP = S::NVP, GP = S::GP, kG = group_size<NV>();
  constexpr double INF = INFINITY;
  constexpr double BIG = 1e300;  // finite stand-in for +inf inside packed argmin keys
  double *sm = shared_base();
  double *Ls = sm + S::oL;
  double *xs = sm + S::oX;
  double *ds = sm + S::oD;
  double *d2s = sm + S::oD2;
  double *vs = sm + S::oV;
  double *was = sm + S::oWa;
  double *gs = sm + S::oGs;
  double *Gs = sm + S::oGd;

  const int lane = lane_id();
  const int nv = a.nv, Kd = a.Kd, K = a.K, md = a.md;
  const bool in = lane < nv;
  const int li = in ? lane : 0;
  const int lv = lane < NV ? lane : 0;  // clamped index into NV-sized LDS vectors

  // ------------------------------------------------------------------ (0)+(1) stack
  double M[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) M[j] = 0.0;
  double ci = 0.0, mu_l = 0.0, dadd = 0.0;

  const double *Jb = a.J + b * (long long)Kd * nv;
  const double *eb = a.e + b * (long long)K;
  const double *costb = a.cost_batched ? a.cost + b * (long long)K : a.cost;

  // Rows are staged with the compile-time pitch NV so that the uniform (broadcast)
  // reads below have static, 16-byte aligned offsets (ds_read_b128 = 2 doubles).
  constexpr int RC = (NVP < 32) ? NVP : 32;  // rows per staging chunk (NVP*NV doubles available)
  for (int r0 = 0; r0 < Kd; r0 += RC) {
    const int rc = (Kd - r0 < RC) ? Kd - r0 : RC;
    wave_sync();
    stage_rows(Ls, NV, Jb + (long long)r0 * nv, rc, nv, lane);
    if (lane < rc) {
      const int k = r0 + lane;
      const double w = costb[k], ev = eb[k], g = a.row_gain[k], l = a.row_lm[k];
      const double wa = w * w;  // the weight enters squared: (W J)^T (W J), task.py:158-165
      was[lane] = wa;
      gs[lane] = g * wa * ev;             // c_t = -We^T WJ = +gain J^T W^2 e, task.py:166
      mu_l += l * (g * g) * wa * ev * ev;  // mu = lm * |W(-gain e)|^2, task.py:159-160
    }
    wave_sync();
    for (int k = 0; k < rc; ++k) {
      const double *row = Ls + k * NV;
      const double jki = row[li];
      const double aa = was[k] * jki;
      ci += gs[k] * jki;
#pragma unroll
      for (int j0 = 0; j0 < NV; j0 += kG) {
#pragma unroll
        for (int j = j0; j < j0 + kG; ++j) M[j] += aa * row[j];  // j >= nv: stale LDS, zeroed below
#pragma unroll
        for (int j = j0; j < j0 + kG; ++j) pin(M[j]);
      }
    }
  }
  // diagonal tasks: J = eye[col0:col0+k] -> H[i][i] += w^2, c[i] += gain w^2 e  (posture_task.py:128-129)
  if (in) {
    for (int t = 0; t < a.n_dtasks; ++t) {
      const int off = lane - a.dtask_col0[t];
      if (off >= 0 && off < a.dtask_k[t]) {
        const int r = a.dtask_row0[t] + off;
        const double w = costb[r], ev = eb[r], g = a.row_gain[r], l = a.row_lm[r];
        const double wa = w * w;
        dadd += wa;
        ci += g * wa * ev;
        mu_l += l * (g * g) * wa * ev * ev;
      }
    }
    if (a.c_extra) ci += a.c_extra[b * (long long)nv + lane];
  }
  double diag = a.damping + wave_sum(mu_l);  // solve_ik.py:55 + sum_t mu_t (task.py:165)

  // dense inequality rows -> LDS; barrier regulariser r / ||J_h||_F^2 (barrier.py:193-200)
  double hv = 0.0, ginv = 1.0;
  if (md > 0) {
    wave_sync();
    stage_rows(Gs, GP, a.Gd + b * (long long)md * nv, md, nv, lane);
    wave_sync();
    if (lane < md) {
      hv = a.hd[b * (long long)md + lane];
      double s = 0.0;
      for (int j = 0; j < nv; ++j) s += Gs[lane * GP + j] * Gs[lane * GP + j];
      ginv = (s > 0.0) ? 1.0 / sqrt(s) : 1.0;
    }
    for (int t = 0; t < a.n_barriers; ++t) {
      const double r = a.barrier_safe_gain[t];
      if (r > 1e-6) {
        double s = 0.0;
        if (in)
          for (int rr = a.barrier_rows[t]; rr < a.barrier_rows[t + 1]; ++rr)
            s += Gs[rr * GP + lane] * Gs[rr * GP + lane];
        s = wave_sum(s);  // ||G_b||_F^2 = ||J_h||_F^2 / dt^2
        diag += r / (s * a.dt * a.dt);
      }
    }
  }
  diag += dadd;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (j >= nv || !in) M[j] = 0.0;  // padding rows/columns
    if (j == lane) M[j] += in ? diag : 1.0;  // padded coordinates: identity, c = 0, unbounded
  }
  if (!in) ci = 0.0;

  if (!SOLVE) {
    // coalesced write-out through LDS: H [nv, nv] row-major, then c
    wave_sync();
    if (in) {
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv) Ls[lane * nv + j] = M[j];
    }
    wave_sync();
    double *Hb = a.H_out + b * (long long)nv * nv;
    for (int idx = lane; idx < nv * nv; idx += kWave) Hb[idx] = Ls[idx];
    if (in) a.c_out[b * (long long)nv + lane] = ci;
    return;
  }

  // ------------------------------------------------------------------ (2) factor
  // Right-looking Cholesky; lane i ends with row i of L in M[0..i] and L is also
  // written row-major to LDS.  The forward solve L y = -c rides along.
  int status = STATUS_OPTIMAL;
  double cp = -ci;
  wave_sync();
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double p = bcast(M[j], j);
    if (!(p > 0.0)) {  // quadprog: "matrix G is not positive definite"
      status = STATUS_NOT_PD;
      p = 1.0;
    }
    const double rinv = fast_rsqrt(p);
    const double lij = M[j] * rinv;
    M[j] = lij;
    if (lane >= j && lane < NV) Ls[lane * NVP + j] = lij;
    if (lane < NV) xs[lane] = lij;
    if (lane == 0) ds[j] = rinv;
    const double yj = bcast(cp * rinv, j);
    cp = (lane > j) ? cp - lij * yj : (lane == j ? yj : cp);
    wave_sync();
#pragma unroll
    for (int m0 = (j + 1) & ~(kG - 1); m0 < NV; m0 += kG) {
#pragma unroll
      for (int m = m0; m < m0 + kG; ++m)
        if (m > j) M[m] -= lij * xs[m];
#pragma unroll
      for (int m = m0; m < m0 + kG; ++m)
        if (m > j) pin(M[m]);
    }
    wave_sync();
  }
  // J = L^-T: lane i solves L y = e_i by forward substitution (uniform reads of L)
  if (lane < NV) xs[lane] = cp;  // y
  wave_sync();
  double Jr[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double acc = (lane == j) ? 1.0 : 0.0;
#pragma unroll
    for (int m0 = 0; m0 < j; m0 += kG) {
#pragma unroll
      for (int m = m0; m < m0 + kG; ++m)
        if (m < j) acc -= Ls[j * NVP + m] * Jr[m];
      pin(acc);
    }
    Jr[j] = acc * ds[j];
    pin(Jr[j]);
  }
  double rown2 = 0.0;  // |row i of J|^2 = (H^-1)_ii, invariant under J <- J Q
#pragma unroll
  for (int j = 0; j < NV; ++j) rown2 += Jr[j] * Jr[j];
  double x = 0.0;  // unconstrained minimum x = L^-T y
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    x += Jr[j] * xs[j];
    if ((j & (kG - 1)) == kG - 1) pin(x);
  }
  wave_sync();

  // ------------------------------------------------------------------ (3) Goldfarb-Idnani
  double *Rs = Ls;  // R[l][k] at Rs[k*NVP + l]
  const double lbv = in ? a.lb[b * (long long)nv + lane] : -INF;
  const double ubv = in ? a.ub[b * (long long)nv + lane] : INF;
  const double tol = 1e-13;
  const double thr_lo = (in && lbv > -INF) ? -tol * (1.0 + fabs(lbv)) : -INF;  // -inf: never violated
  const double thr_up = (in && ubv < INF) ? -tol * (1.0 + fabs(ubv)) : -INF;
  const int max_iter = a.max_iter > 0 ? a.max_iter : 20 * (nv + md) + 50;
  int q = 0, it = 0;
  int bstate = 0;   // lane i: 0 free, 1 lower bound active, 2 upper bound active
  int dactive = 0;  // lane r: dense row r active
  int A = 0;        // lane k: id of the constraint at active position k
  double u = 0.0, rdiag = 0.0;  // lane k: multiplier and 1/R[k][k]
  bool running = (status == STATUS_OPTIMAL);

  while (running) {
    // (a) most violated constraint, violation / row norm as quadprog.  `best` is a
    // key: the (negative) scaled violation with the constraint id in its low bits.
    double best = BIG, sd = 0.0;
    const double slo = x - lbv, sup = ubv - x;
    if (bstate != 1 && slo < thr_lo) best = key_pack(slo, lane);
    if (bstate != 2 && sup < thr_up && sup < best) best = key_pack(sup, 64 + lane);
    if (md > 0) {
      if (lane < NV) xs[lane] = x;
      wave_sync();
      if (lane < md) {
        double s = hv;
        for (int j = 0; j < nv; ++j) s -= Gs[lane * GP + j] * xs[j];
        sd = s;
        const double sc = s * ginv;
        if (!dactive && sc < -tol * (1.0 + fabs(hv) * ginv) && sc < best) best = key_pack(sc, 128 + lane);
      }
      wave_sync();
    }
    best = wave_min(best);
    if (!(best < 0.0)) break;  // no violated constraint: optimal
    const int bid = key_payload(best);
    const int kind = bid >> 6, src = bid & 63;
    double sp = (kind == 0) ? bcast(slo, src) : (kind == 1) ? bcast(sup, src) : bcast(sd, src);
    double uplus = 0.0;

    for (;;) {
      if (++it > max_iter) {
        status = STATUS_MAX_ITER;
        running = false;
        break;
      }
      // (b) d = J^T n+ ; n+ = +e_src (lower), -e_src (upper), -g_src (dense row)
      double dl = 0.0;
      if (kind < 2) {
        if (lane == src) {  // row src of J -> LDS (one active lane, b128 stores), sign applied on read
#pragma unroll
          for (int j = 0; j < NV; ++j) ds[j] = Jr[j];
        }
        wave_sync();
        const double rowv = (lane < NV) ? ds[lv] : 0.0;
        dl = (kind == 0) ? rowv : -rowv;
      } else {
        const double gi = in ? -Gs[src * GP + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const double s = wave_sum(Jr[j] * gi);
          if (lane == j) dl = s;
        }
      }
      // |d|^2 = n+^T H^-1 n+.  For a box row it is the squared norm of row src of J,
      // which the orthogonal updates of J never change: computed once per lane.
      const double dd = (kind < 2) ? bcast(rown2, src) : wave_sum(dl * dl);
      const double d2n = wave_sum((lane >= q) ? dl * dl : 0.0);  // = z^T n+ = |d2|^2
      const bool lin_dep = !(d2n > 1e-24 * dd);
      const double dq_ = bcast(dl, q < kWave ? q : kWave - 1);
      const double rn2 = lin_dep ? 0.0 : fast_rsqrt(d2n);  // 1/|d2|
      const double nrm2 = d2n * rn2;
      const double sgq = (dq_ >= 0.0) ? 1.0 : -1.0;
      // Householder reflector H = I - beta v v^T with H d2 = -sgq |d2| e_q
      const double beta = lin_dep ? 0.0 : rn2 * fast_rcp(nrm2 + fabs(dq_));
      if (lane < NV) {
        d2s[lane] = (lane >= q) ? dl : 0.0;
        vs[lane] = (lane > q) ? dl : (lane == q ? dq_ + sgq * nrm2 : 0.0);
      }
      wave_sync();
      double z = 0.0, w = 0.0;  // z = J2 d2 (primal step direction), w = J2 v
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        z += Jr[j] * d2s[j];
        w += Jr[j] * vs[j];
        if ((j & (kG - 1)) == kG - 1) {
          pin(z);
          pin(w);
        }
      }
      // r = R^-1 d1 (dual step direction): column-oriented back substitution
      // (lane k's dp is final once step k is reached, so r_k = dp * rdiag at the end)
      double dp = dl;
      for (int k = q - 1; k > 0; --k) {
        const double rk = bcast(dp * rdiag, k);
        if (lane < k) dp -= Rs[k * NVP + lane] * rk;
      }
      const double rv = dp * rdiag;
      // (c) step lengths
      const bool blocking = lane < q && rv > 0.0;
      const double ratio = blocking ? u * fast_rcp(rv) : BIG;
      const double k1 = wave_min(blocking ? key_pack(ratio, lane) : BIG);
      const int kd = key_payload(k1);
      const double t1 = (k1 < BIG) ? bcast(ratio, kd & 63) : INF;
      const double t2 = lin_dep ? INF : -sp * rn2 * rn2;
      const double t = (t1 < t2) ? t1 : t2;
      if (!(t < INF)) {  // quadprog: "constraints are inconsistent, no solution"
        status = STATUS_INFEASIBLE;
        running = false;
        break;
      }
      bool do_drop = true;
      if (!(t2 < INF)) {
        // step in the dual space only
        if (lane < q) u -= t * rv;
        uplus += t;
      } else {
        x += t * z;
        if (lane < q) u -= t * rv;
        uplus += t;
        if (t2 <= t1) {
          // full step: the constraint becomes active.  J2 <- J2 H, R gains column [d1; -sgq|d2|]
          const double wb = beta * w;
#pragma unroll
          for (int j0 = 0; j0 < NV; j0 += kG) {
#pragma unroll
            for (int j = j0; j < j0 + kG; ++j) Jr[j] -= wb * vs[j];
#pragma unroll
            for (int j = j0; j < j0 + kG; ++j) pin(Jr[j]);
          }
          const double rqq = -sgq * nrm2;
          if (lane < q) Rs[q * NVP + lane] = dl;
          if (lane == q) {
            Rs[q * NVP + q] = rqq;
            rdiag = -sgq * rn2;
            A = bid;
            u = uplus;
          }
          if (lane == src) {
            if (kind == 0) bstate = 1;
            else if (kind == 1) bstate = 2;
            else dactive = 1;
          }
          ++q;
          wave_sync();
          do_drop = false;
        }
      }
      if (!do_drop) break;  // back to (a)

      // (d) drop the blocking constraint at active position kd
      {
        const int idk = bcast_i(A, kd);
        if (lane == (idk & 63)) {
          if ((idk >> 6) < 2) bstate = 0;
          else dactive = 0;
        }
        wave_sync();
        // remove column kd of R: lane = row shifts its own row left (no cross-lane hazard)
        if (lane < q)
          for (int col = kd; col < q - 1; ++col) Rs[col * NVP + lane] = Rs[(col + 1) * NVP + lane];
        {
          const double un = from_next_lane(u);
          const int An = from_next_lane_i(A);
          if (lane >= kd && lane < q - 1) {
            u = un;
            A = An;
          }
        }
        --q;
        wave_sync();
        // restore triangularity: Givens on rows (l, l+1) of R, same rotation on columns (l, l+1) of J.
        // lane = column of R; every lane only touches its own column.
#pragma unroll
        for (int l = 0; l < NV - 1; ++l) {
          if (l >= kd && l < q) {
            const bool mine = (lane >= l && lane < q);
            const double ra = mine ? Rs[lane * NVP + l] : 0.0;
            const double rb = mine ? Rs[lane * NVP + l + 1] : 0.0;
            const double ga = bcast(ra, l), gb = bcast(rb, l);
            if (gb != 0.0) {
              const double rh = fast_rsqrt(ga * ga + gb * gb);
              const double cc = ga * rh, ss = gb * rh;
              if (mine) {
                Rs[lane * NVP + l] = cc * ra + ss * rb;
                Rs[lane * NVP + l + 1] = -ss * ra + cc * rb;
              }
              const double ja = Jr[l], jb = Jr[l + 1];
              Jr[l] = cc * ja + ss * jb;
              Jr[l + 1] = -ss * ja + cc * jb;
            }
          }
        }
        if (lane >= kd && lane < q) rdiag = fast_rcp(Rs[lane * NVP + lane]);
        wave_sync();
      }
      // slack of the pending constraint at the new x, then iterate (b) with the same n+
      if (t2 < INF) {
        if (kind == 0) sp = bcast(x - lbv, src);
        else if (kind == 1) sp = bcast(ubv - x, src);
        else sp += t * d2n;
      }
    }
  }

  // ------------------------------------------------------------------ (4) write-out
  if (in) a.dq[b * (long long)nv + lane] = x;
  if (lane == 0) {
    a.status[b] = status;
    if (a.iters) a.iters[b] = it;
  }
}

template <int NV>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_ATTR(NV) ik_solve_kernel(KernelArgs a) {
  ik_instance<NV, true>(a, block_id());
}

template <int NV>
__global__ void __launch_bounds__(kWave) ik_stack_kernel(KernelArgs a) {
  ik_instance<NV, false>(a, block_id());
}

}  // namespace pinkhip
