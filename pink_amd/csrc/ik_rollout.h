// One kernel per control step: forward kinematics, FrameTask rows, limits, posture error, stacking, QP solve and
// integration for 64/W robots per wavefront -- what the reference does per robot and per step with Pinocchio +
// NumPy + quadprog (pink/solve_ik.py:206-275 followed by configuration.integrate_inplace,
// examples/inverse_kinematics_ur10.py:75-91).
//
// The task Jacobians never exist in memory: the kinematics phase (ik_kinematics.h) leaves, per robot, the world
// twist of each tangent column in that column's lane and two 3 x 3 blocks per FrameTask in LDS; the stacking of
// the solve kernel (ik_stack_rows.h / ik_sweep.h, source policy FkTerms) forms its six rows per task from them with
// 27 FMAs per lane.  Against the two-launch loop (ik_step_kernel writes J, e, lb, ub; ik_solve_packed_kernel reads
// them back) a step moves ~0.3 kB per robot through HBM instead of ~14 kB.  The kinematics scratch and the solve
// kernel's LDS (the stated problem parked for its closing refinement step, written after stacking) overlay each other.
#pragma once

#include "dispatch.h"
#include "ik_kinematics.h"
#include "ik_sweep.h"
#include "ik_sweepx.h"

namespace pinkhip {

struct RolloutArgs {
  KernelArgs k;  // task tables, cost, damping, dt, B, nv, Kd = 6 nf, K; outputs dq / status / iters; J, e, lb, ub unused
  FkArgs fk;     // model, q, targets, limit gain, posture target; T_frames optional; e / J / lb / ub outputs unused
  int integrate;       // apply dq to q at the end of the step (instances whose solve failed keep their q)
  int *first_failure;  // [B] sticky status | (step << 8), may be NULL
  int step;
  // PositionBarrier rows (pink/barriers/position_barrier.py:109-153), k.md of them: row d keeps
  // sign_d (p_frame_d [axis_d] - bound_d) >= 0;  G_d = -sign_d (R J_lin)[axis_d] / dt,  h_d = gain_d sign_d (p - bound)
  // (pink/barriers/barrier.py:246-254), both formed on chip from the world twists of the joint axes
  const int *bar_frame = nullptr, *bar_axis = nullptr;
  const double *bar_sign = nullptr, *bar_bound = nullptr, *bar_gain = nullptr;
  // FloatingBaseVelocityLimit rows that are not axis-aligned (floating_base_velocity_limit.py:128-148): the FIRST n_lim
  // of the k.md dense rows (Pink stacks limits before barriers, solve_ik.py:62-84), constant: row d is lim_rows[6 d ..]
  // on the six tangent coordinates of the root joint, zero elsewhere, with right-hand side lim_h[d].  The bar_*
  // tables are indexed by d - n_lim.
  int n_lim = 0;
  const double *lim_rows = nullptr, *lim_h = nullptr;
  // Equality constraints made of frame tasks (pink/solve_ik.py:125-149: A = J, b = -gain e): the LEADING 6 n_eqf dense rows
  // (k.n_eq of them), in front of the limit rows; constraint c is the FrameTask of model frame eq_frame[c] -- formed like
  // any frame task's rows, which is why the frame sits in the model (with zero cost when it carries no task)
  int n_eqf = 0;
  const int *eq_frame = nullptr;
  const double *eq_gain = nullptr;
  // BodySphericalBarrier rows (pink/barriers/body_spherical_barrier.py:73-143): a barrier row with bar_axis = 3 keeps
  // |p_f - p_f2|^2 - bound >= 0 (bound = d_min^2) with the class-K function h / (1 + |h|) of the reference's class:
  // G = -2 (p_f - p_f2)^T (R J_lin,f - R J_lin,f2) / dt,  h = gain alpha(|p_f - p_f2|^2 - d_min^2)
  const int *bar_frame2 = nullptr;
  // Dense tasks with a constant Jacobian (LinearHolonomicTask / JointCouplingTask on vector-space joints,
  // pink/tasks/linear_holonomic_task.py:103-148: e = A (q (-) q_0) - b, J = A): n_crow rows BEHIND the 6 nf FrameTask
  // rows of the dense block (k.Kd = 6 nf + n_crow); A [n_crow, nv], q_0 [nq], b [n_crow] in device memory
  int n_crow = 0;
  const double *crow_A = nullptr, *crow_q0 = nullptr, *crow_b = nullptr;
  // Diagonal tasks: the one whose error is q (-) q_target (the PostureTask) occupies the rows post_row0 .. + post_k of
  // e; every other diagonal task has batch-constant errors (DampingTask: 0, LowAccelerationTask: -dt v_prev,
  // JointVelocityTask: -dt v*; pink/tasks/damping_task.py, low_acceleration_task.py:46-84, joint_velocity_task.py:59-110)
  // read from diag_e [K - Kd] at row - Kd
  int post_row0 = 0, post_k = 0;
  const double *diag_e = nullptr;
};

// doubles of kinematics scratch per robot: joint poses, ancestor pointers, U / V blocks, frame errors (+ the errors of
// n_crow constant rows), joint scalars
__device__ __host__ inline int rollout_fk_doubles(int nj, int nf, int n_crow = 0) { return fk_lds_doubles(nj, nf) + 6 * nf + ((n_crow + 1) & ~1); }

template <int W>
struct FkTerms {
  static constexpr bool kOnTheFly = true;
  static constexpr bool kKeep = true;
  double lb = 0.0, ub = 0.0, x = 0.0;
  int status = 0;
  double lin[3], ang[3], post_e = 0.0, qv = 0.0;
  unsigned anc = 0, ancr = 0;  // bit f: this column's joint is an ancestor of frame f / of the root frame of relative slot f
  double *es = nullptr;        // LDS: errors of the dense rows [6 nf + n_crow]
  const double *UV = nullptr;  // LDS: U (9), V (9) per frame, pitch 36
  int nf = 0, n_crow = 0, col = 0, nvc = 0;  // frames, constant rows, this lane's tangent column, row pitch of crow_A
  const double *crow_A = nullptr;
  int post_row0 = 0, post_k = 0, Kd = 0;
  const double *diag_e = nullptr;
  // this lane's entries (tangent column li) of the six rows of FrameTask f -- or, behind the frames, of the next six
  // constant rows (their Jacobian is a table)
  __device__ __forceinline__ void frame_rows(int f, double (&six)[6]) const {
    if (f >= nf) {
      const int r0 = 6 * (f - nf);
#pragma unroll
      for (int i = 0; i < 6; ++i) six[i] = (r0 + i < n_crow) ? crow_A[(long long)(r0 + i) * nvc + col] : 0.0;
      return;
    }
    const double *u = UV + 36 * f;
    // (-1: ancestor of the frame only, +1: of the root frame of a relative slot only, 0: of both or neither)
    const double sgn = (double)((int)((ancr >> f) & 1u) - (int)((anc >> f) & 1u));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double top = u[3 * i] * lin[0] + u[3 * i + 1] * lin[1] + u[3 * i + 2] * lin[2] + u[9 + 3 * i] * ang[0] +
                         u[9 + 3 * i + 1] * ang[1] + u[9 + 3 * i + 2] * ang[2];
      const double bot = u[3 * i] * ang[0] + u[3 * i + 1] * ang[1] + u[3 * i + 2] * ang[2];
      six[i] = sgn * top;
      six[i + 3] = sgn * bot;
    }
  }
  __device__ __forceinline__ double error(int k) const { return es[k]; }
  __device__ __forceinline__ double diag_error(int r) const {
    if (r >= post_row0 && r < post_row0 + post_k) return post_e;
    return diag_e ? diag_e[r - Kd] : 0.0;
  }
  // dense rows = position barriers: this lane's entry (tangent column li) of row d, and the row's right-hand side.
  // World velocity of the frame origin p_f per unit of this column's joint velocity: lin + ang x p_f (zero unless the
  // joint is an ancestor of the frame) = column li of R J_lin (position_barrier.py:136-145).
  const int *bar_frame = nullptr, *bar_axis = nullptr;
  const double *bar_sign = nullptr, *bar_bound = nullptr, *bar_gain = nullptr;
  // LDS: frame f's world position at pfs[3 f .. 3 f + 2] -- a copy behind the area the kinematics scratch shares with
  // the solvers' working sets (dispatch.h, rollout_tail_doubles): the Goldfarb-Idnani code of the hand-over forms the
  // barrier rows while it is already writing its staged copy of them (its Gs overlaid the frame poses: the right-hand
  // side of a barrier row read back an entry of G -- scripts/gpu_fuzz_rollout.py seed 24680, "inconsistent")
  const double *pfs = nullptr;
  double inv_dt = 0.0;
  int n_lim = 0, root_sub = -1;  // constant rows of the floating-base limit; this lane's coordinate of the root joint
  const double *lim_rows = nullptr, *lim_h = nullptr;
  // equality rows of constraint frame tasks: 24 doubles per constraint in the tail of the robot's LDS (a copy, like pfs):
  // U (9), V (9) of the frame and its six errors
  int n_eqr = 0;  // 6 x constraints
  const int *eq_frame = nullptr;
  const double *eq_gain = nullptr, *eqs = nullptr;
  const int *bar_frame2 = nullptr;
  // world velocity of frame f's origin per unit velocity of this lane's tangent column (zero unless an ancestor)
  __device__ __forceinline__ void origin_velocity(int f, double (&v)[3]) const {
    const double *pf = pfs + 3 * f;
    const double on = (((anc >> f) & 1u) != 0) ? 1.0 : 0.0;
    v[0] = on * (lin[0] + ang[1] * pf[2] - ang[2] * pf[1]);
    v[1] = on * (lin[1] + ang[2] * pf[0] - ang[0] * pf[2]);
    v[2] = on * (lin[2] + ang[0] * pf[1] - ang[1] * pf[0]);
  }
  __device__ __forceinline__ double dense_col(int d) const {
    if (d < n_eqr) {
      // row i of the FrameTask Jacobian of constraint c (frame_rows(), from the tail copy of U, V)
      const int c = d / 6, i = d - 6 * c, f = eq_frame[c];
      const double *u = eqs + 24 * c;
      const double sgn = (double)((int)((ancr >> f) & 1u) - (int)((anc >> f) & 1u));
      const int r = i < 3 ? i : i - 3;
      const double bot = u[3 * r] * ang[0] + u[3 * r + 1] * ang[1] + u[3 * r + 2] * ang[2];
      const double top = u[3 * r] * lin[0] + u[3 * r + 1] * lin[1] + u[3 * r + 2] * lin[2] + u[9 + 3 * r] * ang[0] + u[9 + 3 * r + 1] * ang[1] +
                         u[9 + 3 * r + 2] * ang[2];
      return sgn * (i < 3 ? top : bot);
    }
    d -= n_eqr;
    if (d < n_lim) return root_sub >= 0 ? lim_rows[6 * d + root_sub] : 0.0;
    d -= n_lim;
    const int f = bar_frame[d], i = bar_axis[d];
    double v[3];
    origin_velocity(f, v);
    if (i == 3) {  // BodySphericalBarrier: -2 (p_f - p_f2) . (v_f - v_f2) / dt
      const int f2 = bar_frame2[d];
      double w[3];
      origin_velocity(f2, w);
      const double *p1 = pfs + 3 * f, *p2 = pfs + 3 * f2;
      return -2.0 * inv_dt * ((p1[0] - p2[0]) * (v[0] - w[0]) + (p1[1] - p2[1]) * (v[1] - w[1]) + (p1[2] - p2[2]) * (v[2] - w[2]));
    }
    const double vi = i == 0 ? v[0] : (i == 1 ? v[1] : v[2]);
    return -bar_sign[d] * vi * inv_dt;
  }
  __device__ __forceinline__ double dense_h(int d) const {
    if (d < n_eqr) {
      const int c = d / 6;
      return -eq_gain[c] * eqs[24 * c + 18 + (d - 6 * c)];  // b = -gain e (pink/solve_ik.py:147)
    }
    d -= n_eqr;
    if (d < n_lim) return lim_h[d];
    d -= n_lim;
    if (bar_axis[d] == 3) {
      const double *p1 = pfs + 3 * bar_frame[d], *p2 = pfs + 3 * bar_frame2[d];
      const double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
      const double hb = dx * dx + dy * dy + dz * dz - bar_bound[d];
      return bar_gain[d] * hb / (1.0 + fabs(hb));  // (barrier.py:246-254 with the class's gain function)
    }
    return bar_gain[d] * bar_sign[d] * (pfs[3 * bar_frame[d] + bar_axis[d]] - bar_bound[d]);
  }
};

template <int NV, int MD, int W>
__device__ __forceinline__ void ik_rollout_instance(const RolloutArgs &a, long long block) {
  constexpr int G = kWave / W;
  const ModelDev &m = a.fk.m;
  const int lane = lane_id();
  const int g = lane / W, li = lane & (W - 1);
  long long b = block * G + g;
  const bool valid = b < a.k.B;
  if (!valid) b = a.k.B - 1;
  // kinematics scratch at the start of this robot's share of LDS: the solve (which keeps stacking and the
  // factorisation in registers) first writes there after its last read of it; the share is the larger of the two
  double *sm = shared_base() + (long long)g * a.k.lds_pitch;
  auto make_terms = [&](const RolloutArgs &ra) {
    const ModelDev &mm = ra.fk.m;
    FkTerms<W> tt;
    tt.es = sm + fk_lds_doubles(mm.nj, mm.nf);
    tt.UV = sm + 12 * (mm.nj + mm.nf) + ((mm.nj + 1) & ~1);  // = Jls of ik_fk_instance
    tt.nf = mm.nf, tt.n_crow = ra.n_crow, tt.col = li < mm.nv ? li : 0, tt.nvc = mm.nv, tt.crow_A = ra.crow_A;
    tt.post_row0 = ra.post_row0, tt.post_k = ra.post_k, tt.Kd = ra.k.Kd, tt.diag_e = ra.diag_e;
    if constexpr (MD > 0) {
      tt.bar_frame = ra.bar_frame, tt.bar_axis = ra.bar_axis;
      tt.bar_sign = ra.bar_sign, tt.bar_bound = ra.bar_bound, tt.bar_gain = ra.bar_gain;
      tt.pfs = sm + ra.k.lds_pitch - rollout_tail_doubles(mm.nf, ra.n_eqf);
      tt.inv_dt = 1.0 / ra.k.dt;
      tt.n_lim = ra.n_lim, tt.lim_rows = ra.lim_rows, tt.lim_h = ra.lim_h;
      tt.n_eqr = 6 * ra.n_eqf, tt.eq_frame = ra.eq_frame, tt.eq_gain = ra.eq_gain, tt.bar_frame2 = ra.bar_frame2;
      tt.eqs = tt.pfs + ((3 * mm.nf + 1) & ~1);
      if (ra.n_lim > 0 && mm.root_nv == 6) {  // (the free-flyer is the first joint after the universe: columns 0 .. 5)
        const int jt = mm.dof_joint[li < mm.nv ? li : 0];
        if (li < mm.nv && mm.jtype[jt] == JOINT_FREE_FLYER) tt.root_sub = li - mm.idx_v[jt];
      }
    }
    return tt;
  };
  // (frame positions out of the shared area: fMo of ik_fk_instance holds frame f's pose at sm[12 (nj + f) ..])
  auto keep_frame_positions = [&](const RolloutArgs &ra) {
    if constexpr (MD > 0) {
      const ModelDev &mm = ra.fk.m;
      double *tail = sm + ra.k.lds_pitch - rollout_tail_doubles(mm.nf, ra.n_eqf);
      for (int i = li; i < 3 * mm.nf; i += W) tail[i] = sm[12 * (mm.nj + i / 3) + 9 + i % 3];
      if (ra.n_eqf > 0) {  // (kernel argument: wave-uniform)
        double *eqt = tail + ((3 * mm.nf + 1) & ~1);
        const double *UVs = sm + 12 * (mm.nj + mm.nf) + ((mm.nj + 1) & ~1), *ess = sm + fk_lds_doubles(mm.nj, mm.nf);
        for (int i = li; i < 24 * ra.n_eqf; i += W) {
          const int c = i / 24, k = i - 24 * c, f = ra.eq_frame[c];
          eqt[i] = k < 18 ? UVs[36 * f + k] : ess[6 * f + (k - 18)];
        }
      }
      wave_sync();
    }
  };
  // errors of the constant-row tasks, e_r = A_r (q (-) q_0) - b_r on the vector-space joints, behind the frame errors in LDS
  auto const_row_errors = [&](const RolloutArgs &ra, const FkTerms<W> &tt) {
    if (ra.n_crow > 0) {  // wave-uniform (kernel argument)
      const ModelDev &mm = ra.fk.m;
      const int jt = mm.dof_joint[li < mm.nv ? li : 0];
      const bool vec = li < mm.nv && mm.jtype[jt] != JOINT_FREE_FLYER;
      const double dqv = vec ? tt.qv - ra.crow_q0[mm.idx_q[jt]] : 0.0;
      for (int r = 0; r < ra.n_crow; ++r) {
        const double er = group_sum<W>(vec ? ra.crow_A[(long long)r * mm.nv + li] * dqv : 0.0) - ra.crow_b[r];
        if (li == 0) tt.es[6 * mm.nf + r] = er;
      }
      wave_sync();
    }
  };
  // A task stack that is rank deficient by construction (host_tables.h: FrameTasks alone on more coordinates than they
  // have rows, examples/humanoid_jvrc.py:69-81) goes to the Goldfarb-Idnani code right away (kernel argument:
  // wave-uniform; its kinematics pass is the one below); everything else through the sweep tableau.
  const bool direct = a.k.rank_deficient != 0;
  FkTerms<W> t = make_terms(a);
  int st_sweep = STATUS_ROUTED;
  if (!direct) {
    ik_fk_instance<W, true, true, FkTerms<W>>(a.fk, block, &t, sm);
    wave_sync();
    keep_frame_positions(a);
    const_row_errors(a, t);
    // (more tableau rows than lanes: the dense rows are virtual, ik_sweepx.h -- two robots per wavefront at nv = 30
    // with barrier rows instead of one)
    if constexpr (NV + MD > W) st_sweep = ik_sweepx_instance<NV, MD, W, FkTerms<W>>(a.k, block, &t);
    else st_sweep = ik_sweep_instance<NV, MD, W, FkTerms<W>>(a.k, block, &t);
  }
  // a result that fails its KKT certificate is not integrated: the Goldfarb-Idnani code solves that robot's QP again
  // (ik_sweep.h, ik_solve_sweep_body).  It forms the rows from the kinematics like the tableau did, and the tableau's
  // parking area has overwritten the kinematics scratch meanwhile: the kinematics run once more (wave-uniform, rare).
#ifdef PINKHIP_SWEEP_NO_HANDOVER  // (development: what the tableau code alone says)
  const bool over = false;
#else
  const bool over = st_sweep == STATUS_BREAKDOWN || st_sweep == STATUS_ROUTED;
#endif
  if (wave_any(over)) {
    wave_sync();
    // (arguments read again, terms built again: nothing of them is kept in registers through the tableau loop)
    const RolloutArgs *again = kernarg_reload<RolloutArgs>(a);
    FkTerms<W> t2 = make_terms(*again);
    ik_fk_instance<W, true, true, FkTerms<W>>(again->fk, block, &t2, sm);
    wave_sync();
    keep_frame_positions(*again);
    const_row_errors(*again, t2);
    ik_packed_instance<NV, W, (MD > 0), FkTerms<W>>(again->k, block, &t2, over,
                                                     again->k.rank_deficient ? PATH_GI : (st_sweep == STATUS_ROUTED ? PATH_ROUTED : PATH_HANDOVER));
    if (over) t.x = t2.x, t.status = t2.status;
  }
  // integration: lane = joint fetches its dq entries from the lanes that hold them (lane = tangent coordinate)
  const int st = t.status;  // group-uniform
  const bool isj = li < m.nj;
  const int jl = isj ? li : 0;
  const int iv = m.idx_v[jl], iq = m.idx_q[jl];
  const bool ff = m.jtype[jl] == JOINT_FREE_FLYER;
  double v[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[i] = lane_shfl(t.x, g * W + ((i == 0 || ff) ? (iv + i) & (W - 1) : iv & (W - 1)));
  if (a.integrate) {
    if (st == 0) {
      if (valid && isj) {
        double *qw = a.fk.q_rw + b * (long long)m.nq + iq;
        double qj[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) qj[i] = (i == 0 || ff) ? qw[i] : 0.0;
        integrate_joint(m, jl, qj, v);
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (i == 0 || ff) qw[i] = qj[i];
      }
    } else if (valid && li == 0 && a.first_failure && a.first_failure[b] == 0) {
      a.first_failure[b] = st | (a.step << 8);
    }
  }
}

template <int NV, int MD, int W>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_ROLLOUT(NV + MD) ik_rollout_kernel(RolloutArgs a) {
  ik_rollout_instance<NV, MD, W>(a, block_id());
}

}  // namespace pinkhip
