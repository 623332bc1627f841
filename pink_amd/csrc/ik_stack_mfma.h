// Stack-only kernel on the fp64 matrix cores: H = damping I + sum_t J_t^T W_t^2 J_t + mu_t I,
// c = sum_t gain_t J_t^T W_t^2 e_t for one QP per wavefront (reference pink/tasks/task.py:145-167,
// pink/solve_ik.py:54-67 -- the P, q of pink.build_ik).
//
// This is the one GEMM-shaped piece of the path: H (nv x nv) = (W^2 J)^T (Kd x nv) * J (Kd x nv).
// It is tiled for v_mfma_f64_16x16x4_f64: NT x NT output tiles of 16 x 16, K consumed 4 task rows
// at a time.  Lane l loads J[k0 + (l >> 4)][16 tc + (l & 15)] straight from HBM (16 lanes = 128
// contiguous bytes) and the same value serves as the B operand of tile column tc and, scaled by
// w_k^2, as the A operand of tile row tc -- no LDS staging, no broadcast reads; the accumulators
// are written out in the MFMA C/D layout (16 lanes = 128 contiguous bytes of one row of H).
// The kernel is bound by the HBM stream: 8 (Kd nv + K + nv^2 + nv) bytes per QP.
#pragma once

#include "ik_common.h"

namespace pinkhip {

// STAGED = true (NT >= 3 with even nv, Kd <= 32 and 16-byte aligned J: the JVRC-shaped configuration): the
// instance's J block (Kd nv contiguous doubles) is requested as ONE flat stream, 16 bytes per lane and 1 KiB per wave
// instruction, and reaches the MFMA operand layout through LDS.  The direct path requests 128-byte pieces of 8 nv-byte
// rows per tile column; at nv = 50 the fourth tile column holds two of sixteen lanes.  Measured (MI355X, B = 65 536):
// nv = 50: 653 -> 546 us; nv = 30 (NT = 2): 186 us either way, so NT <= 2 stays direct.  Staging H on the way out the
// same way (16 nv contiguous doubles per row of tiles) or pairing neighbouring columns into 16-byte stores measured
// slower (nv = 30: 209 / 189 us; nv = 50 with both stagings: 789 us, LDS-bound occupancy).
template <int NT, bool STAGED = false>
__device__ inline void ik_stack_mfma_instance(const KernelArgs &a, long long b) {
  double *sm = shared_base();
  double *was = sm;        // [128] w_k^2 of the current pass
  double *gws = sm + 128;  // [128] gain_k w_k^2 e_k
  double *Js = sm + 256;   // STAGED: [Kd nv] the J block
  const int lane = lane_id();
  const int nv = a.nv, Kd = a.Kd, K = a.K;
  const int col = lane & 15, rq = lane >> 4;

  const double *Jb = a.J + b * (long long)Kd * nv;
  const double *eb = a.e + b * (long long)K;
  const double *costb = a.cost_batched ? a.cost + b * (long long)K : a.cost;
  double *Hb = a.H_out + b * (long long)nv * nv;

  // Dense rows are consumed 32 at a time (eight MFMA k-steps requested from HBM at once: one memory round
  // trip per 32 rows).  For NT >= 3 the output is produced ONE ROW OF TILES at a time: NT accumulators are live
  // and leave for HBM as soon as their K loop is done, instead of NT x NT tiles (276 VGPRs = one wave per SIMD
  // at NT = 4; a streaming kernel needs the occupancy to hide the HBM latency).  When all dense rows fit one
  // request (Kd <= 32: every BASELINE configuration) J stays in registers for all tile rows; larger task
  // stacks re-request it per tile row and are served by L2.
  constexpr int kSteps = 8;
  const bool one_pass = STAGED || Kd <= 4 * kSteps;
  double Jp[kSteps][NT];
  const int nJ = Kd * nv;
  constexpr int TL = STAGED ? 4 * NT : 1;  // 32 rows of 16 NT doubles = 4 NT wave-wide 16-byte requests
  Pair jflat[TL];
  auto stage_flat = [&]() {  // registers -> LDS (flat) -> this lane's MFMA operands
    Pair *Jsp = reinterpret_cast<Pair *>(__builtin_assume_aligned(Js, 16));
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      const int idx = lane + kWave * t;
      if (2 * idx < nJ) Jsp[idx] = jflat[t];
    }
    wave_sync();
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const int kk = 4 * st + rq;
#pragma unroll
      for (int tc = 0; tc < NT; ++tc) {
        const int j = 16 * tc + col;
        Jp[st][tc] = (kk < Kd && j < nv) ? Js[kk * nv + j] : 0.0;
      }
    }
  };
  auto request = [&](int r0) {
    if constexpr (STAGED) {
      const Pair *Jg = reinterpret_cast<const Pair *>(__builtin_assume_aligned(Jb, 16));
#pragma unroll
      for (int t = 0; t < TL; ++t) {
        const int idx = lane + kWave * t;
        jflat[t] = (2 * idx < nJ) ? Jg[idx] : Pair{0.0, 0.0};
      }
      return;
    }
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const int kk = r0 + 4 * st + rq;  // this lane's task row
#pragma unroll
      for (int tc = 0; tc < NT; ++tc) {
        const int j = 16 * tc + col;
        Jp[st][tc] = (kk < Kd && j < nv) ? Jb[(long long)kk * nv + j] : 0.0;
      }
    }
  };
  auto build_table = [&](int p0) {  // coefficients of the 128 rows from p0 on
    const int rc = (Kd - p0 < 128) ? Kd - p0 : 128;
    wave_sync();
    for (int rr = lane; rr < rc; rr += kWave) {
      const int k = p0 + rr;
      const double w = costb[k], ev = eb[k];
      const double wa = w * w;
      was[rr] = wa;
      gws[rr] = a.row_gain[k] * wa * ev;
    }
    wave_sync();
  };
  if (Kd > 0) request(0);  // in flight while the scalars are computed

  // ---- scalars: Levenberg-Marquardt mu over all task rows, c and diagonal of the diagonal tasks, barriers
  double mu_l = 0.0;
  for (int k = lane; k < Kd; k += kWave) {
    const double w = costb[k], ev = eb[k], g = a.row_gain[k];
    mu_l += a.row_lm[k] * (g * g) * (w * w) * ev * ev;
  }
  // diagonal tasks (J = eye[col0:col0+k], posture_task.py:128-129) in the lane = coordinate layout
  const bool in = lane < nv;
  double ci = 0.0;
  if (in) {
    for (int t = 0; t < a.n_dtasks; ++t) {
      const int off = lane - a.dtask_col0[t];
      if (off >= 0 && off < a.dtask_k[t]) {
        const int r = a.dtask_row0[t] + off;
        const double w = costb[r], ev = eb[r], g = a.row_gain[r], l = a.row_lm[r];
        const double wa = w * w;
        ci += g * wa * ev;
        mu_l += l * (g * g) * wa * ev * ev;
      }
    }
    if (a.c_extra) ci += a.c_extra[b * (long long)nv + lane];
  }
  double diag = a.damping + wave_sum(mu_l);
  for (int t = 0; t < a.n_barriers; ++t) {  // barrier.py:193-200: r / ||J_h||_F^2, J_h = -Gd dt
    const double r = a.barrier_safe_gain[t];
    if (r > 1e-6) {
      const double *Gb = a.Gd + b * (long long)a.md * nv;
      double s = 0.0;
      for (int idx = a.barrier_rows[t] * nv + lane; idx < a.barrier_rows[t + 1] * nv; idx += kWave)
        s += Gb[idx] * Gb[idx];
      s = wave_sum(s);
      diag += r / (s * a.dt * a.dt);
    }
  }
  if (one_pass && Kd > 0) build_table(0);
  if constexpr (STAGED) stage_flat();

  double cpart[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) cpart[t] = 0.0;

  // ---- TR rows of tiles at a time (all of them up to NT = 2: 112 -> 136 VGPRs still leaves three waves per SIMD
  // and measures 7 % faster at nv = 30 than row by row; one row for NT >= 3: 629 vs 913 us at nv = 50)
  constexpr int TR = NT <= 2 ? NT : 1;
  static_for<0, NT / TR>([&](auto TG) {
    constexpr int t0 = decltype(TG)::value * TR;
    v4d acc[TR][NT];
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tr][tj][r] = 0.0;
    for (int p0 = 0; p0 < Kd; p0 += 128) {
      const int rc = (Kd - p0 < 128) ? Kd - p0 : 128;
      if (!one_pass) build_table(p0);
      for (int c0 = 0; c0 < rc; c0 += 4 * kSteps) {
        if (!one_pass) request(p0 + c0);
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
          if (c0 + 4 * st < rc) {  // wave-uniform
            const int kk = c0 + 4 * st + rq;
            const bool krow = kk < rc;
            const double wa = krow ? was[kk] : 0.0;
            if constexpr (t0 == 0) {
              const double gw = krow ? gws[kk] : 0.0;
#pragma unroll
              for (int tc = 0; tc < NT; ++tc) cpart[tc] += gw * Jp[st][tc];
            }
#pragma unroll
            for (int tr = 0; tr < TR; ++tr) {
              const double Av = wa * Jp[st][t0 + tr];
#pragma unroll
              for (int tj = 0; tj < NT; ++tj) acc[tr][tj] = mfma_f64_16x16x4(Av, Jp[st][tj], acc[tr][tj]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) {
      const int ti = t0 + tr;
      // diagonal entries live in lanes with (lane >> 4) == (col & 3), element col >> 2 of the diagonal tile
      if (rq == (col & 3)) {
        const int i = 16 * ti + col;
        double dd = diag;
        for (int t = 0; t < a.n_dtasks; ++t) {
          const int off = i - a.dtask_col0[t];
          if (off >= 0 && off < a.dtask_k[t]) {
            const double w = costb[a.dtask_row0[t] + off];
            dd += w * w;
          }
        }
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
          if (tj == ti) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r == (col >> 2)) acc[tr][tj][r] += dd;
          }
      }
      // write-out: element r of tile (ti, tj) is H[16 ti + rq + 4 r][16 tj + col] (16 lanes = 128 contiguous bytes)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + rq + 4 * r;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int j = 16 * tj + col;
          if (i < nv && j < nv) Hb[(long long)i * nv + j] = acc[tr][tj][r];
        }
      }
    }
  });

  // c of the dense tasks: sum the four row quarters, lane i keeps coordinate i
#pragma unroll
  for (int tc = 0; tc < NT; ++tc) {
    double s = cpart[tc];
    s += lane_shfl(s, lane ^ 16);
    s += lane_shfl(s, lane ^ 32);
    if (rq == tc) ci += s;
  }
  if (in) a.c_out[b * (long long)nv + lane] = ci;
}

// Small problems (nv <= 8, no barrier regulariser): 2 TP instances per wavefront.  A 16 x 16 MFMA tile holds
// TWO instances block-diagonally -- lanes with col < 8 feed rows / columns 0..7 from instance A, the others rows /
// columns 8..15 from instance B; every k-step multiplies task row k of both (the off-diagonal blocks are
// computed and ignored) -- and the wave owns TP such tiles.  One wave then streams 2 TP x 8 (Kd nv + K + nv^2 + nv)
// bytes instead of one instance's 720 B (UR5), with the same instruction count per tile.
template <int TP>
__device__ inline void ik_stack_small_instance(const KernelArgs &a, long long block) {
  static_assert(TP >= 1 && TP <= 4, "tile t's c is written by the lanes of row quarter t");
  const int lane = lane_id();
  const int nv = a.nv, Kd = a.Kd, K = a.K;
  const int col = lane & 15, rq = lane >> 4, half = col >> 3, c8 = col & 7;
  long long inst[TP];
  bool ok[TP];
#pragma unroll
  for (int t = 0; t < TP; ++t) {
    inst[t] = (block * TP + t) * 2 + half;
    ok[t] = inst[t] < a.B;
    if (!ok[t]) inst[t] = a.B - 1;
  }
  v4d acc[TP];
  double cpart[TP], mup[TP];
#pragma unroll
  for (int t = 0; t < TP; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
    cpart[t] = 0.0;
    mup[t] = 0.0;
  }
  for (int k0 = 0; k0 < Kd; k0 += 4) {
    const int kk = k0 + rq;
    const bool krow = kk < Kd;
    const int kc = krow ? kk : 0;
    const double g = a.row_gain[kc], l = a.row_lm[kc];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const double *costb = a.cost_batched ? a.cost + inst[t] * (long long)K : a.cost;
      const double w = costb[kc], ev = a.e[inst[t] * (long long)K + kc];
      const double wa = krow ? w * w : 0.0;
      const double jv = (krow && c8 < nv) ? a.J[(inst[t] * (long long)Kd + kk) * nv + c8] : 0.0;
      cpart[t] += (g * wa * ev) * jv;
      if (c8 == 0) mup[t] += l * (g * g) * wa * ev * ev;
      acc[t] = mfma_f64_16x16x4(wa * jv, jv, acc[t]);
    }
  }
  // c: fold the four row quarters; the lanes of row quarter t keep tile t's (coordinate c8 of instance `half`)
  double ci = 0.0;
#pragma unroll
  for (int t = 0; t < TP; ++t) {
    double sres = cpart[t];
    sres += lane_shfl(sres, lane ^ 16);
    sres += lane_shfl(sres, lane ^ 32);
    if (rq == t) ci = sres;
  }
  // diagonal tasks (J = eye[col0:col0+k], posture_task.py:128-129) on the writer lanes
  long long mine = a.B - 1;
  bool mine_ok = false;
#pragma unroll
  for (int t = 0; t < TP; ++t)
    if (rq == t) {
      mine = inst[t];
      mine_ok = ok[t];
    }
  const bool writer = rq < TP && c8 < nv;
  double mu_d = 0.0;
  if (writer) {
    const double *costb = a.cost_batched ? a.cost + mine * (long long)K : a.cost;
    for (int t = 0; t < a.n_dtasks; ++t) {
      const int off = c8 - a.dtask_col0[t];
      if (off >= 0 && off < a.dtask_k[t]) {
        const int r = a.dtask_row0[t] + off;
        const double w = costb[r], ev = a.e[mine * (long long)K + r], g = a.row_gain[r], l = a.row_lm[r];
        const double wa = w * w;
        ci += g * wa * ev;
        mu_d += l * (g * g) * wa * ev * ev;
      }
    }
    if (a.c_extra) ci += a.c_extra[mine * (long long)nv + c8];
  }
  // Levenberg-Marquardt mu per instance: dense-row partials (lanes c8 == 0, every row quarter) + diagonal-task
  // partials (writer lanes), summed over the eight lanes of the half and the four row quarters
#pragma unroll
  for (int t = 0; t < TP; ++t) {
    double m = mup[t] + (rq == t ? mu_d : 0.0);
    m += lane_shfl(m, lane ^ 1);
    m += lane_shfl(m, lane ^ 2);
    m += lane_shfl(m, lane ^ 4);
    m += lane_shfl(m, lane ^ 16);
    m += lane_shfl(m, lane ^ 32);
    // diagonal entry (i, i), i = col, of tile t sits in the lanes with rq == (col & 3), element col >> 2
    if (rq == (col & 3)) {
      const double *costb = a.cost_batched ? a.cost + inst[t] * (long long)K : a.cost;
      double dd = a.damping + m;
      for (int q = 0; q < a.n_dtasks; ++q) {
        const int off = c8 - a.dtask_col0[q];
        if (off >= 0 && off < a.dtask_k[q]) {
          const double w = costb[a.dtask_row0[q] + off];
          dd += w * w;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r == (col >> 2)) acc[t][r] += dd;
    }
    // element r of the tile is (row rq + 4 r, column col): inside a diagonal block when both belong to one half
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rq + 4 * r, i = row & 7;
      if ((row >> 3) == half && i < nv && c8 < nv && ok[t]) a.H_out[(inst[t] * (long long)nv + i) * nv + c8] = acc[t][r];
    }
  }
  if (writer && mine_ok) a.c_out[mine * (long long)nv + c8] = ci;
}

template <int TP>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_SMALL_STACK ik_stack_small_kernel(KernelArgs a) {
  ik_stack_small_instance<TP>(a, block_id());
}

template <int NT>
__global__ void __launch_bounds__(kWave) ik_stack_mfma_kernel(KernelArgs a) {
  ik_stack_mfma_instance<NT>(a, block_id());
}

// LDS of the staged variant, in doubles: coefficient tables and the J block
__host__ __device__ inline int stack_staged_lds_doubles(int nv, int Kd) { return 256 + Kd * nv; }
// the staged variant needs 16-byte aligned per-instance J blocks: Kd nv even, 16-byte aligned base
__host__ __device__ inline bool stack_staged_ok(int nv, int Kd, const void *J) {
  return nv > 32 && ((Kd * nv) & 1) == 0 && Kd > 0 && Kd <= 32 && (reinterpret_cast<unsigned long long>(J) & 15) == 0;
}

template <int NT>
__global__ void __launch_bounds__(kWave) ik_stack_staged_kernel(KernelArgs a) {
  ik_stack_mfma_instance<NT, true>(a, block_id());
}

}  // namespace pinkhip
