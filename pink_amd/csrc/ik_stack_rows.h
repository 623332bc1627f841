// Stacking of the task rows into H = damping I + sum_t (W_t J_t)^T (W_t J_t) + mu_t I, c = sum_t gain_t J_t^T W_t^2 e_t
// (pink/tasks/task.py:145-167, pink/solve_ik.py:54-60) for the kernels whose groups are whole rows of 16 lanes: lane li
// of a group accumulates row li of H in registers, fed by the broadcast-FMA of wave.h -- no LDS, no cross-lane
// reduction.  Shared by the Goldfarb-Idnani kernel (ik_kernels_packed.h) and the sweep-tableau kernel (ik_sweep.h).
#pragma once

#include "ik_common.h"

namespace pinkhip {

// Where the per-instance terms come from.  HbmTerms (default): the packed streams J / e / lb / ub in HBM.  A policy
// with kOnTheFly = true (ik_rollout.h: the whole-control-step kernel) produces the task rows, errors and bounds
// itself -- frame_rows(f, dst) fills this lane's entries of the six rows of FrameTask f, error(k) / diag_error(r)
// return task errors, lb / ub are members -- and receives the result back in x / status.
struct HbmTerms {
  static constexpr bool kOnTheFly = false;
  double lb = 0.0, ub = 0.0, x = 0.0;
  int status = 0;
  __device__ __forceinline__ void frame_rows(int, double (&)[6]) const {}
  __device__ __forceinline__ double error(int) const { return 0.0; }
  __device__ __forceinline__ double diag_error(int) const { return 0.0; }
  __device__ __forceinline__ double dense_col(int) const { return 0.0; }
  __device__ __forceinline__ double dense_h(int) const { return 0.0; }
};

// Dense task rows.  Lane li requests J[k][li] for the RC rows of a chunk (one coalesced request per row), lane k the
// weights of row k; DEPTH chunks are in flight ahead of the one being accumulated.  Row k then enters every lane's H
// row through the broadcast-FMA:  H[li][j] += (w_k^2 J[k][li]) * J[k][j]  with J[k][j] taken from lane j and w_k^2
// from lane k.  M holds at least NV entries (the first NV are used); ci and mu_l accumulate this lane's entry of c
// and its share of the Levenberg-Marquardt term.
// DEPTH: a chunk of eight rows is accumulated in ~1 k cycles of this wave's own issue time, a request takes several
// thousand under load -- with ONE chunk ahead (DEPTH = 1) a wave of the sweep-tableau kernel waited out the memory
// latency once per chunk (section clock, round 4: 30 k cycles per wave in this phase, whatever the input regime);
// DEPTH = 2 has all 24 rows of the headline stack in flight before the first FMA.  (an on-the-fly source delivers
// one FrameTask = six rows per chunk and nothing is in flight)
// `mid` runs between the first requests and the first accumulation: what else the caller wants from HBM (bounds,
// the terms of the diagonal tasks) is requested there and arrives within the same wait.
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};
// NX > 0: the task rows have NX columns IN FRONT of the ones the lanes hold (a.J points at column NX of row 0; ik_sweep.h:
// coordinates that are eliminated before the solve).  Their entries are group-uniform: lane kk of a chunk requests
// J[r0 + kk][x - NX] with its weights and the rows hand them round with the broadcast that hands the weights round.
// Collected: this lane's entries H[li][x] in m[x], and PER-LANE PARTIAL sums of the uniform entries (to be summed over
// the group by the caller): hxx[0] = H[0][0], and for NX = 2 hxx[1] = H[0][1], hxx[2] = H[1][1]; cx[x] = c[x].
template <int NX>
struct StackFront {
  double m[NX > 0 ? NX : 1] = {};
  double hxx[NX > 0 ? NX * (NX + 1) / 2 : 1] = {};
  double cx[NX > 0 ? NX : 1] = {};
};
template <int NV, int W, int RCMAX, class Src, int DEPTH = 1, int NX = 0, int NM, class Mid = NoMid>
__device__ __forceinline__ void stack_rows_bcast(const KernelArgs &a, long long b, Src *terms, bool in, int li,
                                                 double (&M)[NM], double &ci, double &mu_l, Mid mid = Mid(), StackFront<NX> *front = nullptr) {
  static_assert(NX >= 0 && NX <= 2 && (NX == 0 || !Src::kOnTheFly), "at most two front columns, from HBM");
  static_assert(NM >= NV && W >= 16, "row-group kernels only");
  using BcT = Bcast<W>;
  const int nv = a.nv, Kd = a.Kd, K = a.K;
  const double *Jb = a.J + b * (long long)Kd * nv;
  const double *eb = a.e + b * (long long)K;
  const double *costb = a.cost_batched ? a.cost + b * (long long)K : a.cost;
  constexpr int RC = Src::kOnTheFly ? 6 : (RCMAX < 8 ? RCMAX : 8);
  constexpr int NB = Src::kOnTheFly ? 2 : DEPTH + 1;  // chunk buffers: the one being accumulated + those in flight
  static_assert(RC <= 16, "weight rows are broadcast from the first row of 16 lanes");
  struct Chunk {
    double r[RC];
    double pw, pe, pg, pl;
    double px[NX > 0 ? NX : 1];
  };
  Chunk buf[NB];
  auto request = [&](Chunk &dst, int r0) {
    const int rc = (Kd - r0 < RC) ? Kd - r0 : RC;  // (<= 0: past the end, nothing is read)
    if constexpr (Src::kOnTheFly) {
      double six[6];
      terms->frame_rows(r0 / 6, six);
#pragma unroll
      for (int kk = 0; kk < RC; ++kk) dst.r[kk] = (in && kk < rc) ? six[kk] : 0.0;
    } else {
#pragma unroll
      for (int kk = 0; kk < RC; ++kk) dst.r[kk] = (in && kk < rc) ? Jb[(long long)(r0 + kk) * nv + li] : 0.0;
    }
    dst.pw = dst.pe = dst.pg = dst.pl = 0.0;
    if constexpr (NX > 0) {
#pragma unroll
      for (int x = 0; x < NX; ++x) dst.px[x] = (li < rc) ? Jb[(long long)(r0 + li) * nv - NX + x] : 0.0;
    }
    if (li < rc) {
      const int k = r0 + li;
      dst.pw = costb[k];
      if constexpr (Src::kOnTheFly) dst.pe = terms->error(k);
      else dst.pe = eb[k];
      dst.pg = a.row_gain[k];
      dst.pl = a.row_lm[k];
    }
  };
  if (Kd > 0) {
#pragma unroll
    for (int d = 0; d < NB - 1; ++d)
      if (d * RC < Kd) request(buf[d], d * RC);  // wave-uniform
  }
  mid();
  for (int r0 = 0; r0 < Kd; r0 += RC) {
    const int rc = (Kd - r0 < RC) ? Kd - r0 : RC;
    if (r0 + (NB - 1) * RC < Kd) request(buf[NB - 1], r0 + (NB - 1) * RC);
    const Chunk &cur = buf[0];
    const double wa = (li < rc) ? cur.pw * cur.pw : 0.0;
    const double gw = (li < rc) ? cur.pg * wa * cur.pe : 0.0;
    if (li < rc) mu_l += cur.pl * (cur.pg * cur.pg) * wa * cur.pe * cur.pe;
    const BcT wab = bcast_prepare<W>(wa), gwb = bcast_prepare<W>(gw);
    BcT pxb[NX > 0 ? NX : 1];
    if constexpr (NX > 0) {
      // (lane kk's own row: its part of the uniform entries)
      front->hxx[0] = fma(wa * cur.px[0], cur.px[0], front->hxx[0]);
      front->cx[0] = fma(gw, cur.px[0], front->cx[0]);
      if constexpr (NX == 2) {
        front->hxx[1] = fma(wa * cur.px[0], cur.px[1], front->hxx[1]);
        front->hxx[2] = fma(wa * cur.px[1], cur.px[1], front->hxx[2]);
        front->cx[1] = fma(gw, cur.px[1], front->cx[1]);
      }
#pragma unroll
      for (int x = 0; x < NX; ++x) pxb[x] = bcast_prepare<W>(cur.px[x]);
    }
    static_for<0, RC>([&](auto Kc) {
      constexpr int kk = decltype(Kc)::value;
      if (kk < rc) {  // wave-uniform
        // (the row's own uses first: the row copies of the broadcast are then made in its registers)
        const double aa = fma_bcast<W, kk>(0.0, wab, cur.r[kk]);
        ci = fma_bcast<W, kk>(ci, gwb, cur.r[kk]);
        const BcT rowb = bcast_prepare<W>(cur.r[kk]);
        static_for<0, NV>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          M[j] = fma_bcast<W, j>(M[j], rowb, aa);
        });
        if constexpr (NX > 0) {
#pragma unroll
          for (int x = 0; x < NX; ++x) front->m[x] = fma_bcast<W, kk>(front->m[x], pxb[x], aa);  // H[li][x] += (w^2 J[k][li]) J[k][x]
        }
      }
    });
#pragma unroll
    for (int d = 0; d + 1 < NB; ++d) buf[d] = buf[d + 1];
  }
}

// Diagonal tasks (J = eye[col0 : col0 + k], PostureTask posture_task.py:128-129, DampingTask ...): their Jacobian is
// never stored.  Returns what they add to this lane's diagonal entry; c and the LM term accumulate in ci / mu_l.
template <class Src>
__device__ __forceinline__ double stack_diag_tasks(const KernelArgs &a, long long b, Src *terms, int li, double &ci,
                                                   double &mu_l) {
  const double *eb = a.e + b * (long long)a.K;
  const double *costb = a.cost_batched ? a.cost + b * (long long)a.K : a.cost;
  double dadd = 0.0;
  for (int t = 0; t < a.n_dtasks; ++t) {
    const int off = li - a.dtask_col0[t];
    if (off >= 0 && off < a.dtask_k[t]) {
      const int r = a.dtask_row0[t] + off;
      const double w = costb[r], gn = a.row_gain[r], l = a.row_lm[r];
      double ev;
      if constexpr (Src::kOnTheFly) ev = terms->diag_error(r);
      else ev = eb[r];
      const double wa = w * w;
      dadd += wa;
      ci += gn * wa * ev;
      mu_l += l * (gn * gn) * wa * ev * ev;
    }
  }
  return dadd;
}

}  // namespace pinkhip
