// Packed variant of the stack+solve kernel: W lanes per QP, 64/W QPs per wavefront.
//
// ik_kernels.h gives every QP a whole 64-lane wave; at nv = 30 half the lanes idle and at
// nv = 6 (UR5) 58 of 64 do, while the kernel is VALU-issue bound.  Here lane = (instance,
// row): group g = lane / W handles instance block*G + g, lane li = lane % W owns row li of that
// instance's matrix in NV registers (W >= NV, W in {8, 16, 32}).  Every VALU instruction then
// advances 64/W QPs.  What changes with respect to the one-QP-per-wave kernel:
//   * values that were wave-uniform (q, the selected constraint, step lengths, ...) become
//     group-uniform and live in VGPRs; broadcasts from a data-dependent lane use ds_bpermute,
//     reductions are DPP butterflies inside the group (row pairs through v_permlane16_swap);
//   * the active-set iteration is a flat state machine: each trip of the loop is one
//     Goldfarb-Idnani step for every group that is still running; add / drop / select
//     sections are executed when any group needs them and are predicated per group;
//   * L and R are stored packed (triangular) so that a QP needs ~6 KiB of LDS at NV = 32.
// The arithmetic per instance is the same as in ik_kernels.h (same citations apply).
#pragma once

#include "ik_common.h"
#include "ik_stack_rows.h"

// The triangular factor of the active set is kept as P = R^-1 (R itself is never stored): the dual
// direction r = P d1 is a chain-free matrix-vector product, a new column of P costs one LDS write per
// lane, and on a drop the Givens coefficients are read off the rows of J (for a box constraint column
// k of R is +- row i_k of J).  Measured on MI355X against R + back-substitution: 2.38 -> 2.27 ms per
// 65 536 QPs at nv = 30.  P is stored by diagonals (LdsP::doff): lane li reads P[li][li + m] at a
// compile-time offset for every m, next to d1[li + m] from a zero-padded copy of d1, so the product
// needs neither a predicate nor an address computation; what a lane reads past the end of its row is
// finite and meets a zero of the padding.

// Profiling builds only (scripts/section_clock.py compiles with -DPINKHIP_SECTION_CLOCK): every 64th
// wave adds the s_memtime cycles it spends in each section of the kernel to pinkhip_clock[].
#ifdef PINKHIP_SECTION_CLOCK
static __device__ unsigned long long pinkhip_clock[16];  // one copy per translation unit
#define PINKHIP_TICK(k)                                                        \
  do {                                                                         \
    if (clock_on) {                                                            \
      const unsigned long long now_ = __builtin_readcyclecounter();            \
      if (lane == 0) atomicAdd(&pinkhip_clock[k], now_ - clock_prev);          \
      clock_prev = __builtin_readcyclecounter();                               \
    }                                                                          \
  } while (0)
#else
#define PINKHIP_TICK(k)
#endif

namespace pinkhip {

template <int NV>
struct LdsP {
  static constexpr int GP = NV + 1;               // row pitch of the dense inequality rows
  static constexpr int LEND = NV * (NV + 1) / 2;  // exact packed triangle (L, then P by diagonals)
  static constexpr int TRI = (NV * (NV + 3) / 2 + 1) & ~1;  // + NV slots a lane may read past its row, even
  static constexpr int RC = (TRI / NV < 12) ? TRI / NV : 12;  // staged J rows per chunk (pitch NV); the next
                                                              // chunk waits in RC * NV / W registers per lane
  static constexpr int oT = 0;                    // TRI  staging of J rows, then L, then P = R^-1
  static constexpr int oD = oT + TRI;             // NV   row of J, then d1   (init: 1/diag(L))
  static constexpr int oZ = oD + NV;              // NV   zeros: d1 read at [li + m] runs into them
  static constexpr int oD2 = oZ + NV;             // NV   d with d1 zeroed    (init: column scratch; x for dense rows)
  static constexpr int oV = oD2 + NV;             // NV   Householder vector  (init: forward-solve scratch y)
  static constexpr int oX = oD2;                  // aliases: live ranges do not overlap
  static constexpr int oY = oV;
  static constexpr int oWa = oD;                  // RC <= NV: stacking weights, dead before oD is used
  static constexpr int oGs = oD2;
  static constexpr int oRn = oV + NV;             // NV   (H^-1)_ii = |row i of J|^2: read at the entering row
  static constexpr int oGd = oRn + NV;            // md*GP
  static __host__ __device__ constexpr int stride(int md) { return (oGd + md * GP + 1) & ~1; }  // doubles per QP
  static __host__ __device__ inline long long bytes(int md, int groups) { return 8LL * stride(md) * groups + 16; }
  // L: row i, entries 0..i at i(i+1)/2
  static __host__ __device__ constexpr int lrow(int i) { return i * (i + 1) / 2; }
  // P (upper triangular) by diagonals: P[r][c] at doff(c - r) + r = doff(c) + c r + prow(r)
  static __host__ __device__ constexpr int doff(int m) { return m * (2 * NV + 1 - m) / 2; }
  static __host__ __device__ constexpr int prow(int r) { return -((r * (2 * NV - 1 + r)) / 2); }
};

// DENSE = false is the instantiation for problems without dense inequality / equality rows (box limits
// only, md = 0): every dense-row branch and its state (row slacks, row norms, equality bookkeeping)
// folds away at compile time.
template <int NV, int W, bool DENSE = true, class Src = HbmTerms>
__device__ __forceinline__ void ik_packed_instance(const KernelArgs &a, long long block, Src *terms = nullptr, bool only = true,
                                                   int path = PATH_GI) {
  // only: this lane's group is to be solved (the sweep-tableau kernel hands over the groups whose result did not pass
  // its certificate; the other groups of the wavefront go through the motions and write nothing)
  // path: why this code runs the instance (PATH_*), reported in the high bits of iters[b]
  static_assert(W >= NV && NV % 2 == 0 && (W == 8 || W == 16 || W == 32 || W == 64), "group width");
  using S = LdsP<NV>;
  constexpr int GP = S::GP, G = kWave / W, kG = group_size<NV>();
  constexpr double INF = INFINITY;
  constexpr double BIG = 1e300;
  // groups made of whole rows of 16 lanes multiply lane-held vectors into lane-local accumulators with the
  // broadcast-FMA of wave.h (no LDS); 8-lane groups keep the LDS broadcast
  // (NV = 64 keeps the LDS broadcast: with the broadcast-FMA its fully unrolled body sends LLVM's CodeGenPrepare
  // pass from 2 minutes at NV = 56 to 8 minutes per translation unit)
  constexpr bool kBc = W >= 16 && NV <= 56;
  using BcT = Bcast<(W >= 16 ? W : 16)>;
  // the instantiations with dense rows of the two- and four-QPs-per-wave sizes carry ~16 more registers through the loop
  // than fit next to eight-wide operand batches at three waves per SIMD: they use narrower batches instead of spilling
  constexpr bool kTight = DENSE && (W == 32 || W == 16);

  const int lane = lane_id();
  const int g = lane / W, li = lane & (W - 1);
  const int nv = a.nv, Kd = a.Kd, K = a.K, md = DENSE ? a.md : 0, n_eq = DENSE ? a.n_eq : 0;
#ifdef PINKHIP_SECTION_CLOCK
  const bool clock_on = (block & 63) == 0;
  unsigned long long clock_prev = __builtin_readcyclecounter();
#endif
  long long b = block * G + g;
  const bool valid = b < a.B && only;
  if (b >= a.B) b = a.B - 1;  // surplus groups of the last wave redo the last instance, write nothing

  double *sm = shared_base() + (long long)g * (a.lds_pitch ? a.lds_pitch : S::stride(md));
  double *Ts = sm + S::oT;
  double *xs = sm + S::oX;
  double *ys = sm + S::oY;
  double *ds = sm + S::oD;
  double *d2s = sm + S::oD2;
  double *vs = sm + S::oV;
  double *zs = sm + S::oZ;
  double *was = sm + S::oWa;
  double *gs = sm + S::oGs;
  double *Gs = sm + S::oGd;

  const bool in = li < nv;
  const int lc = in ? li : 0;
  const int lv = li < NV ? li : 0;
  double *Pl = sm + S::oT + lv;                 // P[li][li + m] = Pl[doff(m)]
  double *Pk = sm + S::oT + S::prow(lv);        // P[li][c]      = Pk[doff(c) + c * li]
  const double *d1l = sm + S::oD + lv;          // d1[li + m]

  // ------------------------------------------------------------------ stack (task.py:145-167)
  double M[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) M[j] = 0.0;
  double ci = 0.0, mu_l = 0.0, dadd = 0.0;
  const double *Jb = a.J + b * (long long)Kd * nv;
  const double *eb = a.e + b * (long long)K;
  const double *costb = a.cost_batched ? a.cost + b * (long long)K : a.cost;

  if constexpr (kBc) {
    // no LDS at all: rows requested straight into registers, accumulated through the broadcast-FMA (ik_stack_rows.h)
    stack_rows_bcast<NV, W, S::RC, Src>(a, b, terms, in, li, M, ci, mu_l);
  } else {
    // Rows are staged chunk by chunk (RC rows) through LDS.  The HBM requests of chunk c+1 (rows and their
    // weights) are issued into registers before chunk c is accumulated, so that only the first chunk pays
    // the memory latency; kernels whose chunk does not fit the register budget stage directly.
    constexpr int RL = (S::RC * NV + W - 1) / W;  // row entries per lane and chunk
    constexpr bool kPrefetch = RL <= 20;
    static_assert(S::RC <= W, "one weight row per lane");
    double stage[kPrefetch ? RL : 1];
    double pw = 0.0, pe = 0.0, pg = 0.0, pl = 0.0;
    auto fetch = [&](int r0, int rc) {
      if constexpr (kPrefetch) {
        const double *src = Jb + (long long)r0 * nv;
  #pragma unroll
        for (int t = 0; t < RL; ++t) {
          const int idx = li + t * W;
          stage[t] = (idx < rc * nv) ? src[idx] : 0.0;
        }
      }
      if (li < rc) {
        const int k = r0 + li;
        pw = costb[k];
        pe = eb[k];
        pg = a.row_gain[k];
        pl = a.row_lm[k];
      }
    };
    if (Kd > 0) fetch(0, Kd < S::RC ? Kd : S::RC);
    for (int r0 = 0; r0 < Kd; r0 += S::RC) {
      const int rc = (Kd - r0 < S::RC) ? Kd - r0 : S::RC;
      wave_sync();
      {  // this group's rows, W lanes wide, LDS pitch NV
        int r = li / nv, j = li - r * nv;
        const int dr = W / nv, dj = W - dr * nv;
        if constexpr (kPrefetch) {
  #pragma unroll
          for (int t = 0; t < RL; ++t) {
            if (li + t * W < rc * nv) Ts[r * NV + j] = stage[t];
            r += dr;
            j += dj;
            if (j >= nv) {
              j -= nv;
              ++r;
            }
          }
        } else {
          const double *src = Jb + (long long)r0 * nv;
          for (int idx = li; idx < rc * nv; idx += W) {
            Ts[r * NV + j] = src[idx];
            r += dr;
            j += dj;
            if (j >= nv) {
              j -= nv;
              ++r;
            }
          }
        }
      }
      if (li < rc) {  // RC <= W: one weight row per lane
        const double wa = pw * pw;
        was[li] = wa;
        gs[li] = pg * wa * pe;
        mu_l += pl * (pg * pg) * wa * pe * pe;
      }
      if (r0 + S::RC < Kd) fetch(r0 + S::RC, (Kd - r0 - S::RC < S::RC) ? Kd - r0 - S::RC : S::RC);
      wave_sync();
      for (int k = 0; k < rc; ++k) {
        const double *row = Ts + k * NV;
        const double jki = row[lc];
        const double aa = was[k] * jki;
        ci += gs[k] * jki;
  #pragma unroll
        for (int j0 = 0; j0 < NV; j0 += kG) {
  #pragma unroll
          for (int j = j0; j < j0 + kG; ++j)
            if (j < NV) M[j] += aa * row[j];
  #pragma unroll
          for (int j = j0; j < j0 + kG; ++j)
            if (j < NV) pin(M[j]);
        }
      }
    }
  }
  if (in) {
    dadd = stack_diag_tasks<Src>(a, b, terms, li, ci, mu_l);
    if (a.c_extra) ci += a.c_extra[b * (long long)nv + li];
  }
  double diag = a.damping + group_sum<W>(mu_l);

  double hv = 0.0, ginv = 1.0, gr0 = 0.0, gr1 = 0.0;
#ifndef PINKHIP_GR
#define PINKHIP_GR 2
#endif
  constexpr int kGR = PINKHIP_GR;  // dense rows kept in registers
  if (md > 0) {
    wave_sync();
    if constexpr (Src::kOnTheFly) {
      // (whole-step kernel: barrier / limit rows formed from the kinematics, lane li = their entry in column li)
      for (int r = 0; r < md; ++r)
        if (li < nv) Gs[r * GP + li] = terms->dense_col(r);
    } else {
      const double *src = a.Gd + b * (long long)md * nv;
      int r = li / nv, j = li - r * nv;
      const int dr = W / nv, dj = W - dr * nv;
      for (int idx = li; idx < md * nv; idx += W) {
        Gs[r * GP + j] = src[idx];
        r += dr;
        j += dj;
        if (j >= nv) {
          j -= nv;
          ++r;
        }
      }
    }
    // (columns nv .. NV of the staged rows: read by the eight-wide slack products, must be zero)
    for (int idx = li; idx < md * (GP - nv); idx += W) Gs[(idx / (GP - nv)) * GP + nv + idx % (GP - nv)] = 0.0;
    wave_sync();
    {  // Euclidean norms of the dense rows, eight lanes per row like the slacks of the selection (up to W / 8 rows)
      double n2 = 0.0;
      if (md * 8 <= W) {
        const int row = li >> 3, seg = li & 7;
        const double *gr = Gs + (row < md ? row : 0) * GP;
        double part = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < NV; j0 += 8) {
          const int j = j0 + seg;
          if (j0 + 8 <= NV || j < NV) part += gr[j < NV ? j : 0] * gr[j < NV ? j : 0];
        }
        if (row >= md) part = 0.0;
        part = group_sum<8>(part);
        n2 = group_bcast<W>(part, (li < md ? li : 0) << 3);
      }
      if (li < md) {  // dispatch.h guarantees md <= W: one lane per dense row
        if constexpr (Src::kOnTheFly) hv = terms->dense_h(li);
        else hv = a.hd[b * (long long)md + li];
        if (md * 8 > W)
          for (int j = 0; j < nv; ++j) n2 += Gs[li * GP + j] * Gs[li * GP + j];
        ginv = (n2 > 0.0) ? 1.0 / sqrt(n2) : 1.0;
      }
    }
    // one or two dense rows (a pair of barriers is the common case, examples/humanoid_jvrc.py) also stay in the lanes,
    // entry li with lane li: their slacks are then two group sums per selection -- no LDS round trip, no barrier
    if (md <= kGR) {
      gr0 = (li < NV) ? Gs[li] : 0.0;
      gr1 = (md > 1 && li < NV) ? Gs[GP + li] : 0.0;
    }
    for (int t = 0; t < a.n_barriers; ++t) {
      const double r = a.barrier_safe_gain[t];
      if (r > 1e-6) {
        double s = 0.0;
        if (in)
          for (int rr = a.barrier_rows[t]; rr < a.barrier_rows[t + 1]; ++rr)
            s += Gs[rr * GP + li] * Gs[rr * GP + li];
        s = group_sum<W>(s);
        diag += r / (s * a.dt * a.dt);
      }
    }
  }
  diag += dadd;
  PINKHIP_TICK(0);  // stacking
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (j >= nv || !in) M[j] = 0.0;
    if (j == li) M[j] += in ? diag : 1.0;
  }
  if (!in) ci = 0.0;

  // ------------------------------------------------------------------ Cholesky + forward solve + J = L^-T
  // Right-looking Cholesky with the forward substitutions L y = e_li (J = L^-T, one row per lane) and
  // L y = -c riding along.  Row-group kernels (kBc) keep everything in registers: column j of the trailing
  // matrix, of L and the right-hand side are lane-held vectors that reach the other lanes' FMAs through the
  // DPP broadcast -- the whole factorisation has no LDS access and no barrier.  8-lane groups publish the
  // column through LDS, keep L there (row-major packed) and substitute row by row.
  int status = STATUS_OPTIMAL;
  double cp = -ci;
  double Jr[NV];
  if constexpr (kBc) {
#pragma unroll
    for (int j = 0; j < NV; ++j) Jr[j] = (li == j) ? 1.0 : 0.0;
  }
  double rinv_prev = 0.0;
  auto inverse_row = [&](auto JJ, double rdiag) {
    constexpr int jj = decltype(JJ)::value;
    double acc = (li == jj) ? 1.0 : 0.0;
#pragma unroll
    for (int m0 = 0; m0 < jj; m0 += kG) {
#pragma unroll
      for (int m = m0; m < m0 + kG; ++m)
        if (m < jj) acc -= Ts[S::lrow(jj) + m] * Jr[m];
      pin(acc);
    }
    Jr[jj] = acc * rdiag;
    pin(Jr[jj]);
  };
  wave_sync();
  static_for<0, NV>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
    double p, yraw;
    BcT xb;
    if constexpr (kBc) {
      // column j (lane m holds H~[m][j]) and the right-hand side stay in the lanes: pivot, y_j and the
      // trailing update read them through the DPP broadcast -- no LDS write, no barrier, no read-back
      xb = bcast_prepare<W>(M[j]);  // idle lanes (li >= NV) hold zero rows: nothing to mask
      const BcT yb = bcast_prepare<W>(cp);
      p = value_bcast<W, j>(xb);
      yraw = value_bcast<W, j>(yb);
    } else {
      if (li < NV) {
        xs[li] = M[j];
        ys[li] = cp;
      }
      wave_sync();
      p = xs[j];
      yraw = ys[j];
    }
    if (!(p > 0.0)) {
      status = STATUS_NOT_PD;
      p = 1.0;
    }
    // (the wide instantiations: fold the test into the status register column by column -- left to the compiler it
    // keeps NV lane masks in scalar registers until the loop is over, 2 NV of ~100 SGPRs, and spills them)
    if constexpr (NV > 32) pin(status);
    if constexpr (!kBc && j > 0) inverse_row(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{}, rinv_prev);
    const double rinv = fast_rsqrt(p);
    const double lij = M[j] * rinv;
    const double tj = lij * rinv;  // M[j] / p
    if constexpr (!kBc) {
      if (li >= j && li < NV) Ts[S::lrow(li) + j] = lij;
    }
    const double yj = yraw * rinv;
    cp = (li > j) ? cp - lij * yj : (li == j ? yj : cp);
    if constexpr (kBc) {
      const double ntj = -tj;
      static_for<j + 1, NV>([&](auto Mc) {
        constexpr int m = decltype(Mc)::value;
        M[m] = fma_bcast<W, m>(M[m], xb, ntj);
      });
      // y_j of every substitution is final: scale it, then eliminate it from the rows below with column j
      // of L (lane jj holds L[jj][j])
      Jr[j] *= rinv;
      if constexpr (j + 1 < NV) {
        // only lanes jj > j are ever read from this vector (source lane of Jr[jj]'s update): no mask needed
        const BcT lb = bcast_scale<W>(xb, rinv);  // = bcast_prepare(lij): rinv is group-uniform
        const double nyj = -Jr[j];
        static_for<j + 1, NV>([&](auto Jn) {
          constexpr int jj = decltype(Jn)::value;
          Jr[jj] = fma_bcast<W, jj>(Jr[jj], lb, nyj);
        });
      }
    } else {
#pragma unroll
      for (int m0 = (j + 1) & ~(kG - 1); m0 < NV; m0 += kG) {
#pragma unroll
        for (int m = m0; m < m0 + kG; ++m)
          if (m > j && m < NV) M[m] -= tj * xs[m];
#pragma unroll
        for (int m = m0; m < m0 + kG; ++m)
          if (m > j && m < NV) pin(M[m]);
      }
    }
    rinv_prev = rinv;
    if constexpr (!kBc) wave_sync();  // L[.][j] is in LDS for the forward substitution of the next step
  });
  if constexpr (!kBc) {
    if (li < NV) xs[li] = cp;  // y
    wave_sync();
    inverse_row(std::integral_constant<int, NV - 1>{}, rinv_prev);
  }
  PINKHIP_TICK(1);  // Cholesky, J = L^-T
  double rown2 = 0.0;
#pragma unroll
  for (int j = 0; j < NV; ++j) rown2 += Jr[j] * Jr[j];
  // metric weight of the box constraints on coordinate li in the selection rule: 1 / sqrt((H^-1)_ii)
  const double rsc = in ? fast_rsqrt1(rown2) : 0.0;
  if (li < NV) sm[S::oRn + li] = rown2;
  double rsc_mean = 0.0;
  if (md > 0) rsc_mean = group_sum<W>(rsc) / (double)nv;
  double x = 0.0;
  if constexpr (kBc) {
    const BcT yb = bcast_prepare<W>(li < NV ? cp : 0.0);  // y_j lives in lane j
    static_for<0, NV>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      x = fma_bcast<W, j>(x, yb, Jr[j]);
    });
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      x += Jr[j] * xs[j];
      if ((j & (kG - 1)) == kG - 1) pin(x);
    }
  }
  wave_sync();
  PINKHIP_TICK(2);  // J = L^-T, x0
  // what the P product may read before it is written has to be finite: L has left finite values in the
  // triangle (8-lane groups); the row-group kernels never stored L and clear the whole region
  if constexpr (kBc) {
#pragma unroll
    for (int t = 0; t < (S::TRI + W - 1) / W; ++t)
      if (li + t * W < S::TRI) Ts[li + t * W] = 0.0;
    if (li < NV) zs[li] = 0.0;
  } else if (li < NV) {
    zs[li] = 0.0;
    Ts[S::LEND + li] = 0.0;
    if (li == 0 && S::TRI - S::LEND > NV) Ts[S::TRI - 1] = 0.0;
  }

  // ------------------------------------------------------------------ Goldfarb-Idnani, flat
  // The kernel arguments that are only needed from here on (bounds, iteration cap, output pointers): the wide
  // instantiations re-read them from the kernarg segment through an opaque pointer instead of carrying ~40
  // scalar registers of arguments (spilled and reloaded) across the factorisation.
  const KernelArgs *late = &a;
  if constexpr (NV > 32 && !Src::kOnTheFly) late = kernarg_reload<KernelArgs>(a);
  double lbv, ubv;
  if constexpr (Src::kOnTheFly) {
    lbv = in ? terms->lb : -INF;
    ubv = in ? terms->ub : INF;
  } else {
    lbv = in ? late->lb[b * (long long)nv + li] : -INF;
    ubv = in ? late->ub[b * (long long)nv + li] : INF;
  }
  // violation threshold relative to 1 + |bound|: the round-off of the iterate grows with the dimension (nv dot
  // products of length nv per step) and so does the threshold; same rule as oracle/gi_oracle.c
  const double tol = 1e-13 * (nv > 8 ? nv * 0.125 : 1.0);
  // (an infinite bound gives an infinite slack and threshold: never violated.  The instantiation with dense rows is
  // short of registers: it rebuilds the two thresholds at every selection instead of keeping them)
  const double thr_lo0 = -tol * (1.0 + fabs(lbv)), thr_up0 = -tol * (1.0 + fabs(ubv));
  const int max_iter = late->max_iter > 0 ? late->max_iter : 20 * (nv + md) + 50;
  int q = 0, it = 0, eq_next = 0;  // group-uniform
  int bstate = 0, dactive = 0, A = 0;
  double u = 0.0;
  bool running = (status == STATUS_OPTIMAL);
  bool need_sel = true;
  int kind = 0, src = 0, bid = 0;
  double sp = 0.0, uplus = 0.0, hpend = 0.0;

  double sp_in = 0.0;    // slack of the pending constraint as broadcast by the selection or the drop: only
  bool sp_take = false;  // merged into sp where it is first needed (step lengths), not waited for earlier
  for (;;) {
    // (a) selection, for the groups that have no pending constraint
    if (wave_any(running && need_sel)) {
      double best = BIG, sd = 0.0;
      const double slo = x - lbv, sup = ubv - x;
      // Entering constraint: the violation is weighed by 1 / sqrt(n^T H^-1 n) (= 1 / |row li of J| for a box
      // row, invariant under the orthogonal updates of J): the violated constraint that is farthest away in the
      // metric of the objective.  Any violated constraint is a valid choice for the dual method; against
      // quadprog's violation / |G_i| this one needs ~8 % fewer steps and ~20 % fewer drops on the BASELINE
      // configurations, the minimiser being the same.
      int bestid = 0;
      const double klo = slo * rsc, kup = sup * rsc;
      double thr_lo = thr_lo0, thr_up = thr_up0;
      if constexpr (DENSE) {
        double lbk = lbv, ubk = ubv;
        pin(lbk);
        pin(ubk);
        thr_lo = -tol * (1.0 + fabs(lbk));
        thr_up = -tol * (1.0 + fabs(ubk));
      }
      if (bstate != 1 && slo < thr_lo) best = klo, bestid = li;
      if (bstate != 2 && sup < thr_up && kup < best) best = kup, bestid = 64 + li;
      if (md > 0 && md <= kGR) {
        const double g0 = group_sum<W>(gr0 * x);
        const double g1 = (md > 1) ? group_sum<W>(gr1 * x) : 0.0;
        if (li < md) {
          const double s = hv - (li == 0 ? g0 : g1);
          sd = s;
          const double sc = s * ginv;
          const double kd = sc * rsc_mean;
          if (li >= n_eq && !dactive && sc < -tol * (1.0 + fabs(hv) * ginv) && kd < best) best = kd, bestid = 128 + li;
        }
      } else if (md > 0) {
        if (li < NV) xs[li] = x;
        wave_sync();
        // slacks h_i - g_i x of the dense rows.  Up to W / 8 rows: eight lanes per row, each sums every eighth
        // column, three DPP steps fold the eight partial sums and row i's total is handed to lane i (1 + NV / 8
        // dependent LDS reads per lane instead of NV on md lanes); more rows: one lane per row, eight entries of
        // the row and of x per LDS round trip.  (Columns nv..NV-1 of the staged rows and of x are zero.)
        double gx = 0.0;
        if (md * 8 <= W) {
          int lik = li;  // pinned: the addresses below are rebuilt here instead of living in registers across the loop
          pin(lik);
          const int row = lik >> 3, seg = lik & 7;
          const double *gr = Gs + (row < md ? row : 0) * GP;
          double part = 0.0;
#pragma unroll
          for (int j0 = 0; j0 < NV; j0 += 8) {
            const int j = j0 + seg;  // NV is a multiple of two, not of eight: the last block is partial
            if (j0 + 8 <= NV || j < NV) part += gr[j < NV ? j : 0] * xs[j < NV ? j : 0];
          }
          if (row >= md) part = 0.0;
          part = group_sum<8>(part);
          gx = group_bcast<W>(part, (li < md ? li : 0) << 3);
        }
        if (li < md) {
          double s = hv - gx;
          if (md * 8 > W) {
            const double *gr = Gs + li * GP;
#pragma unroll
            for (int j0 = 0; j0 < NV; j0 += 8) {
              if (j0 < nv) {
                double gv[8], xv[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                  gv[m] = (j0 + m < NV) ? gr[j0 + m] : 0.0;
                  xv[m] = (j0 + m < NV) ? xs[j0 + m] : 0.0;
                }
                pin16(gv, xv);
#pragma unroll
                for (int m = 0; m < 8; ++m) s -= gv[m] * xv[m];
                pin(s);
              }
            }
          }
          sd = s;
          const double sc = s * ginv;
          // (dense rows: Euclidean row norm times the mean metric weight of the box rows)
          const double kd = sc * rsc_mean;
          if (li >= n_eq && !dactive && sc < -tol * (1.0 + fabs(hv) * ginv) && kd < best) best = kd, bestid = 128 + li;
        }
        wave_sync();
      }
      // any violated constraint is a valid choice: the arg-min runs on a 32-bit key (one DPP min per step)
      const float best32 = group_min32<W>(best < 0.0 ? key32_pack(best, bestid) : 3.0e38f);
      const bool none = !(best32 < 0.0f);
      const bool sel = running && need_sel;
      // equalities (the first n_eq dense rows; pink/solve_ik.py:140-149) are activated first, in
      // order, with the normal oriented so that the residual reads as a violation (kind 3 = +g)
      const bool eqsel = sel && eq_next < n_eq;
      double sdp = 0.0, hvp = 0.0;
      if (n_eq > 0) {
        sdp = group_bcast<W>(sd, eq_next < n_eq ? eq_next : 0);
        hvp = group_bcast<W>(hv, eq_next < n_eq ? eq_next : 0);
      }
      if (eqsel) {
        hpend = hvp;
        kind = (sdp > 0.0) ? 3 : 2;
        src = eq_next;
        bid = (kind << 6) | eq_next;
        uplus = 0.0;
        need_sel = false;
        sp = -fabs(sdp);
      }
      if (sel && !eqsel && none) running = false;  // optimal
      if (sel && !eqsel && !none) {
        bid = key32_payload(best32);
        kind = DENSE ? bid >> 6 : (bid >> 6) & 1;
        src = bid & 63;
        uplus = 0.0;
        need_sel = false;
      }
      const double cand = (kind == 0) ? slo : (kind == 1) ? sup : sd;
      const double spn = group_bcast<W>(cand, src & (W - 1));
      if (sel && !eqsel && !none) {
        sp_in = spn;
        sp_take = true;
      }
    }
    if (running) {
      if (++it > max_iter) {
        status = STATUS_MAX_ITER;
        running = false;
      }
    }
    if (!wave_any(running)) break;
    const bool act = running;
    PINKHIP_TICK(3);  // selection

    // (b) d = J^T n+
    double dl = 0.0, rowq = 0.0;
    if (wave_any(act && (!DENSE || kind < 2))) {
      if (act && (!DENSE || kind < 2) && li == src) {
        // (the per-QP LDS regions are 16-byte aligned: NV / 2 ds_write_b128 at immediate offsets)
        Pair *dst = reinterpret_cast<Pair *>(__builtin_assume_aligned(ds, 16));
#pragma unroll
        for (int j = 0; j < NV; j += 2) dst[j >> 1] = Pair{Jr[j], Jr[j + 1]};
      }
      wave_sync();
      const double rowv = (li < NV) ? ds[lv] : 0.0;
      // entry q of the row (the pivot of the Householder step): read here, group-uniform address, instead of a
      // broadcast from lane q once d is known (q = NV reads the zero padding behind the row)
      rowq = ds[q];
      if (act && (!DENSE || kind < 2)) dl = (kind == 0) ? rowv : -rowv;
    }
    if (md > 0 && wave_any(act && (DENSE && kind >= 2))) {
      const bool dn = act && (DENSE && kind >= 2);
      const double gi = (in && dn) ? ((kind == 3) ? Gs[(src & 63) * GP + li] : -Gs[(src & 63) * GP + li]) : 0.0;
      // d_j = sum over the lanes i of J[i][j] g_i for every j at once (one group reduction per j before)
      const double dj = transpose_reduce<W, NV, 0>([&](auto Jc) { return Jr[decltype(Jc)::value] * gi; });
      if (dn) dl = dj;
    }
    PINKHIP_TICK(4);  // d = J^T n
    double dd = sm[S::oRn + (src & (W - 1))];  // (a dense row's index may point past the NV entries: replaced below)
    if (md > 0 && wave_any(act && (DENSE && kind >= 2))) {
      const double dds = group_sum<W>(dl * dl);
      if ((DENSE && kind >= 2)) dd = dds;
    }
    const double d2n = group_sum<W>((li >= q) ? dl * dl : 0.0);
    const bool lin_dep = !(d2n * 1e24 > dd);  // (the product sits on d2n: dd's broadcast is waited for after the reduction)
    double dq_ = (kind == 0) ? rowq : -rowq;
    if constexpr (DENSE) {  // d of a dense row is not in LDS
      const double dqb = group_bcast<W>(dl, q < W ? q : W - 1);
      if (kind >= 2) dq_ = dqb;
    }
    const double rn2 = lin_dep ? 0.0 : fast_rsqrt1(lin_dep ? 1.0 : d2n);
    const double nrm2 = d2n * rn2;
    const double sgq = (dq_ >= 0.0) ? 1.0 : -1.0;
    const double beta = lin_dep ? 0.0 : rn2 * fast_rcp1(lin_dep ? 1.0 : nrm2 + fabs(dq_));
    const double vv = (li > q) ? dl : (li == q ? dl + ((dl >= 0.0) ? nrm2 : -nrm2) : 0.0);  // lane q holds d_q itself
    if (li < NV) {
      ds[li] = (li < q) ? dl : 0.0;  // d1, followed by the zeros of zs
      if constexpr (!kBc) d2s[li] = (li >= q) ? dl : 0.0;
      if constexpr (!kBc) vs[li] = vv;
    }
    wave_sync();
    PINKHIP_TICK(5);  // norms, Householder vector
    // columns below every group's q carry zeros in d2 and v: skip them eight at a time
    const int qlow = groups_min<W>(act ? q : NV);
    double z = 0.0, w = 0.0;
    if constexpr (kBc) {
      // d2 and v live in the lanes: entry j is lane j's, fed to every lane's FMA by the DPP broadcast
      const BcT d2b = bcast_prepare<W>((li >= q && li < NV) ? dl : 0.0);
      const BcT vb = bcast_prepare<W>(li < NV ? vv : 0.0);
      static_for<0, (NV + 7) / 8>([&](auto J8) {
        constexpr int j0 = decltype(J8)::value * 8;
        if (j0 + 8 > qlow) {
          static_for<j0, (j0 + 8 < NV ? j0 + 8 : NV)>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            z = fma_bcast<W, j>(z, d2b, Jr[j]);
            w = fma_bcast<W, j>(w, vb, Jr[j]);
          });
        }
      });
    } else {
#pragma unroll
      for (int j0 = 0; j0 < NV; j0 += 8) {
        if (j0 + 8 > qlow) {
#pragma unroll
          for (int j = j0; j < j0 + 8; ++j) {
            if (j < NV) {
              z += Jr[j] * d2s[j];
              w += Jr[j] * vs[j];
              if ((j & (kG - 1)) == kG - 1 || j == NV - 1) {
                pin(z);
                pin(w);
              }
            }
          }
        }
      }
    }
    PINKHIP_TICK(6);  // z, w
    // r = P d1, P = R^-1 upper triangular: r_li = sum_m P[li][li + m] d1[li + m], no dependency chain
    double rv = 0.0;
    {
      const int qmax = groups_max<W>(act ? q : 0);
#pragma unroll
      for (int m0 = 0; m0 < NV; m0 += 8) {
        if (m0 < qmax) {
          if constexpr (!kTight) {
            double pv[8], dv[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
              pv[m] = (m0 + m < NV) ? Pl[S::doff(m0 + m)] : 0.0;
              dv[m] = (m0 + m < NV) ? d1l[m0 + m] : 0.0;
            }
            pin16(pv, dv);  // all 16 operands in flight before the first FMA
#pragma unroll
            for (int m = 0; m < 8; ++m) rv += pv[m] * dv[m];
            pin(rv);
          } else {
            // (register-tight instantiations: four operand pairs in flight at a time)
#pragma unroll
            for (int h = 0; h < 8; h += 4) {
              double pv[4], dv[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                pv[m] = (m0 + h + m < NV) ? Pl[S::doff(m0 + h + m < NV ? m0 + h + m : 0)] : 0.0;
                dv[m] = (m0 + h + m < NV) ? d1l[m0 + h + m] : 0.0;
              }
              pin8(pv, dv);
#pragma unroll
              for (int m = 0; m < 4; ++m) rv += pv[m] * dv[m];
              pin(rv);
            }
          }
        }
      }
    }
    PINKHIP_TICK(7);  // r = P d1
    // (c) step lengths
    const bool eq_pos = DENSE && (A >> 6) >= 2 && (A & 63) < n_eq;  // equalities are never dropped
    const bool blocking = act && li < q && rv > 0.0 && !eq_pos;
    const double ratio = blocking ? u * fast_rcp1(rv) : BIG;
    // the exact minimum and the first lane that attains it (a ballot, no LDS-crossbar round trip for the value)
    const double k1 = group_min<W>(ratio);
    const int kd = group_first_lane<W>(blocking && ratio == k1) & (W - 1);
    const double t1 = (k1 < BIG) ? k1 : INF;
    if (sp_take) sp = sp_in;
    sp_take = false;
    const double t2 = lin_dep ? INF : -sp * rn2 * rn2;
    const double t = (t1 < t2) ? t1 : t2;
    // (the bound / right-hand side of the pending constraint, for the scale of "violated by round-off only")
    double bnd = 0.0;
    if (wave_any(act && !(t < INF)))
      bnd = group_bcast<W>((DENSE && kind >= 2) ? hv : (kind == 0 ? lbv : ubv), src & (W - 1));
    PINKHIP_TRACEF(li == 0 && act, "[packed g%d it%d] enter src %d kind %d sp %.3e d2n %.3e dd %.3e lin_dep %d t1 %.3e t2 %.3e q %d\n", g, it,
                   src, kind, sp, d2n, dd, (int)lin_dep, t1, t2, q);
    if (act && !(t < INF)) {
      const bool tiny = fabs(sp) <= 1e-9 * (1.0 + fabs(bnd));
      PINKHIP_TRACEF(li == 0, "[packed g%d it%d] stuck: src %d kind %d sp %.3e bnd %.3e tiny %d\n", g, it, src, kind, sp, bnd, (int)tiny);
      if ((DENSE && kind >= 2) && src < n_eq && tiny) {
        // equality implied by the active ones and already satisfied: nothing to add
        ++eq_next;
        need_sel = true;
      } else if (tiny) {
        // An inequality that depends on the active ones, that no drop can help, and that is violated by round-off only
        // (2.5e-13 against a threshold of 1.8e-13 on scripts/gpu_fuzz.py's weakly regularised seed 102469: cond(H)
        // = 8e9 puts that much noise on x): not "inconsistent" -- the bound moves to where the point is.
        if (li == (src & (W - 1))) {
          if (DENSE && kind >= 2) hv -= sp;
          else if (kind == 0) lbv = x;
          else ubv = x;
        }
        need_sel = true;
      } else {
        status = STATUS_INFEASIBLE;
        running = false;
      }
    }
    const bool act2 = act && running && (t < INF);
    const bool dual_only = act2 && !(t2 < INF);
    const bool do_add = act2 && (t2 < INF) && (t2 <= t1);
    const bool do_drop = act2 && !do_add;
    if (act2) {
      if (!dual_only) x += t * z;
      if (li < q) u -= t * rv;
      uplus += t;
    }
    PINKHIP_TICK(8);  // step lengths, x / u update
    // (d) add: J2 <- J2 (I - beta v v^T), R gains column [d1; -sgq |d2|]
    if (wave_any(do_add)) {
      const double wb = do_add ? beta * w : 0.0;
      if constexpr (kBc) {
        const double nwb = -wb;
        const BcT vb2 = bcast_prepare<W>(li < NV ? vv : 0.0);  // rebuilt: two registers live across (c), not four
        static_for<0, (NV + 7) / 8>([&](auto J8) {
          constexpr int j0 = decltype(J8)::value * 8;
          if (j0 + 8 > qlow) {
            static_for<j0, (j0 + 8 < NV ? j0 + 8 : NV)>([&](auto Jc) {
              constexpr int j = decltype(Jc)::value;
              Jr[j] = fma_bcast<W, j>(Jr[j], vb2, nwb);
            });
          }
        });
      } else {
#pragma unroll
        for (int j0 = 0; j0 < NV; j0 += 8) {
          if (j0 + 8 > qlow) {
#pragma unroll
            for (int j = j0; j < j0 + 8; ++j)
              if (j < NV) Jr[j] -= wb * vs[j];
#pragma unroll
            for (int j = j0; j < j0 + 8; ++j)
              if (j < NV) pin(Jr[j]);
          }
        }
      }
      if (do_add) {
        // [R d1; 0 rho]^-1 = [P, -P d1 / rho; 0, 1 / rho] with P d1 = r, rho = -sgq |d2|
        const double rqinv = -sgq * rn2;
        if (li <= q && li < NV) Pk[((q * (2 * NV + 1 - q)) >> 1) + q * li] = (li < q) ? -rv * rqinv : rqinv;
        if (li == q) {
          A = bid;
          u = uplus;
        }
        if (li == src) {
          if (kind == 0) bstate = 1;
          else if (kind == 1) bstate = 2;
        }
        if ((DENSE && kind >= 2) && li == (src & 63)) dactive = 1;
        if ((DENSE && kind >= 2) && src < n_eq) ++eq_next;
        ++q;
        need_sel = true;
      }
    }
    wave_sync();
    PINKHIP_TICK(9);  // add
    // (e) drop the blocking constraint at active position kd
    if (wave_any(do_drop)) {
      const int idk = group_bcast_i<W>(A, kd);
      if (do_drop) {
        if (!DENSE || (idk >> 6) < 2) {
          if (li == (idk & 63)) bstate = 0;
        } else if (li == (idk & 63)) {
          dactive = 0;
        }
      }
      // Removing column kd of R leaves a Hessenberg block that rotations G_l (columns l, l+1 of J and of
      // P, l = kd .. q-2) make triangular again; the new factor is P~ = S^T P G with S^T deleting row kd.
      // The last column of G = G_kd ... G_{q-2} is orthogonal to range(R S), i.e. proportional to row
      // kd of P, and a chain of adjacent rotations is determined by its last column: G_l annihilates
      // the running value a_l (a_kd = p_kd, a_{l+1} = |p_{kd..l+1}|) against p_{l+1}.  All rotations
      // therefore follow from one prefix sum of squares over row kd of P, in parallel, instead of a
      // chain of broadcasts through the rotated rows of J.
      {
        const bool inrow = do_drop && li >= kd && li < q;
        const int mk = inrow ? li - kd : 0;
        const bool rl = do_drop && li >= kd && li < q - 1;
        const double pc = inrow ? Ts[((mk * (2 * NV + 1 - mk)) >> 1) + kd] : 0.0;            // P[kd][li]
        const double pn = rl ? Ts[(((mk + 1) * (2 * NV - mk)) >> 1) + kd] : 0.0;              // P[kd][li + 1]
        const double S = group_scan_sum<W>(pc * pc);
        const double rs = fast_rsqrt1(rl ? S : 1.0), rn = fast_rsqrt1(rl ? S + pn * pn : 1.0);
        const double al = (li == kd) ? pc : S * rs;
        if (li < NV) {
          d2s[2 * li] = rl ? pn * rn : 1.0;    // cos
          d2s[2 * li + 1] = rl ? -al * rn : 0.0;  // sin   (d2s and vs are contiguous and dead here)
        }
      }
      wave_sync();
      // rotation indices any group needs: two wave-uniform bounds, tested per l on the scalar unit;
      // inside them a group that does not rotate at l reads the identity
      constexpr int kRB = kTight ? 2 : 4;  // rotations per batch (reads of a batch in flight together)
      const int l0 = groups_min<W>(do_drop ? kd : NV);
      const int l1 = groups_max<W>(do_drop ? q - 1 : 0);
      const int ro = (li < kd) ? lv : (lv + 1 < NV ? lv + 1 : 0);  // lane li builds NEW row li from old row ro
      const bool prow = do_drop && li < q - 1;
      double *Po = sm + S::oT + S::prow(ro);  // P[ro][c] = Po[doff(c) + c * ro]
      double carry = 0.0;
      if (prow && ro <= kd) carry = Po[((kd * (2 * NV + 1 - kd)) >> 1) + kd * ro];
      static_for<0, (NV - 2 + kRB) / kRB>([&](auto LB) {
        constexpr int lb = decltype(LB)::value * kRB;
        if (lb + kRB > l0 && lb < l1) {
          // old entries P[ro][l + 1] are read one column ahead of the writes of new P[li][l]
          double pb[kRB], cc[kRB], ss[kRB];
          int lik = lv, rok = prow ? ro : NV, wlo = (lv > kd) ? lv : kd, whi = prow ? q - 1 : 0;
          pin(lik);  // pinned: keeps the per-column lane addresses and masks from being hoisted out of the
          pin(rok);  // active-set loop, where they would live in (spilled) registers
          pin(wlo);
          pin(whi);
#pragma unroll
          for (int k = 0; k < kRB; ++k) {
            const int l = lb + k;
            pb[k] = 0.0;
            cc[k] = 1.0;
            ss[k] = 0.0;
            if (l < NV - 1) {
              if (rok <= l + 1) pb[k] = Po[S::doff(l + 1) + (l + 1) * rok];
              cc[k] = d2s[2 * l];
              ss[k] = d2s[2 * l + 1];
            }
          }
          wave_sync();  // every lane has read: old rows are other lanes' new rows
#pragma unroll
          for (int k = 0; k < kRB; ++k) {
            const int l = lb + k;
            if (l < NV - 1) {
              // outside [kd, q - 1) the rotation read back is the identity: carry just moves on to the next old
              // entry, so that only the store is conditional (one exec region per column; the lane's window of
              // columns is two registers -- thirty loop-invariant lane masks would live in spilled SGPRs)
              const double pn_ = cc[k] * carry + ss[k] * pb[k];
              carry = cc[k] * pb[k] - ss[k] * carry;
              if ((l >= wlo) & (l < whi)) Pk[S::doff(l) + l * lik] = pn_;
              const double ja = Jr[l], jb = Jr[l + 1];
              Jr[l] = cc[k] * ja + ss[k] * jb;
              Jr[l + 1] = cc[k] * jb - ss[k] * ja;
            }
          }
        }
      });
      {
        const double un = from_next_lane(u);
        const int An = from_next_lane_i(A);
        if (do_drop && li >= kd && li < q - 1) {
          u = un;
          A = An;
        }
      }
      if (do_drop) --q;
      // slack of the pending constraint at the new x (same n+ next trip); the broadcast is merged into sp
      // where the next trip needs it
      const double cand = (kind == 0) ? x - lbv : ubv - x;
      sp_in = group_bcast<W>(cand, src & (W - 1));
      sp_take = do_drop && !dual_only && (!DENSE || kind < 2);
      if (do_drop && !dual_only && DENSE && kind >= 2) sp += t * d2n;
    }
    PINKHIP_TICK(10);  // drop
  }
  PINKHIP_TICK(11);  // exit

  // ------------------------------------------------------------------ write-out
  if constexpr (Src::kOnTheFly) {
    if (only) {
      terms->x = in ? x : 0.0;
      terms->status = status;
    }
  }
  if (valid) {
    if (in) late->dq[b * (long long)nv + li] = x * late->out_scale;
    if (li == 0) {
      late->status[b] = status;
      if (late->iters) late->iters[b] = it | (path << kPathShift);
    }
  }
}

template <int NV, int W, bool DENSE>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_PACKED(NV, DENSE) ik_solve_packed_kernel(KernelArgs a) {
  ik_packed_instance<NV, W, DENSE>(a, block_id());
}

}  // namespace pinkhip
