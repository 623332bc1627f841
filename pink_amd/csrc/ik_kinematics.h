// Device-side kinematics around the IK step, so that a whole "differential IK iterated to
// convergence" rollout (reference examples/inverse_kinematics_ur10.py:75-91, the loop of
// tests/test_solve_ik.py:160-210) runs without host round trips:
//
//   ik_fk_kernel            q -> frame poses and body (LOCAL) frame Jacobians
//                           (pink/configuration.py:131-164, 203-254: computeJointJacobians,
//                            updateFramePlacements, getFrameJacobian)
//   ik_limits_posture_kernel  q -> merged box of ConfigurationLimit + VelocityLimit
//                           (configuration_limit.py:111-120, velocity_limit.py:118-120) and the
//                           PostureTask error q (-) q* (posture_task.py:100-107)
//   ik_integrate_kernel     q <- q (+) dq   (pink/configuration.py:273-293: pin.integrate)
//   ik_pose_targets_kernel  [B, 7] translation + quaternion targets -> [B, 12] poses
//
// Models are kinematic trees of revolute / prismatic joints with an optional free-flyer root
// (q = [p, quat xyzw], tangent = body twist), joints in topological order.  Conventions as in
// twists are [linear; angular], Jacobians are body Jacobians.
#pragma once

#include "ik_frame_task.h"

namespace pinkhip {

constexpr int JOINT_REVOLUTE = 0;
constexpr int JOINT_PRISMATIC = 1;
constexpr int JOINT_FREE_FLYER = 2;

// Model tables, all in device memory (built by pinkhip_model_create).
struct ModelDev {
  int nj, nq, nv, nf, root_nv;
  const int *parent;        // [nj]  -1 = world
  const int *jtype;         // [nj]
  const int *idx_q;         // [nj]
  const int *idx_v;         // [nj]
  const double *placement;  // [nj, 12]  joint frame in the parent joint frame at q = 0
  const double *axis;       // [nj, 3]
  const int *frame_joint;   // [nf]  -1 = world
  const double *frame_placement;  // [nf, 12]
  const int *dof_joint;     // [nv]  joint owning tangent column j
  const int *dof_sub;       // [nv]  index inside that joint's tangent (0..5 for the free-flyer)
  const unsigned char *anc; // [nf, nj]  joint is the frame's joint or one of its ancestors
  // relative frame slots (pink/tasks/relative_frame_task.py): the pose of frame f is regulated in the frame
  // (root joint, root placement) instead of the world; frame_root_joint[f] = -2: an ordinary slot
  const int *frame_root_joint;          // [nf]  -1 = world, -2 = not relative
  const double *frame_root_placement;   // [nf, 12]
  const unsigned char *ancr;            // [nf, nj]  joint is the root frame's joint or one of its ancestors
  const double *q_min, *q_max;  // [nq]
  const double *v_max;          // [nv]
};

// B <- A * B in place, one column of B at a time (three temporaries)
__device__ inline void se3_lmul(const double *A, double *Bm) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double c0 = Bm[j], c1 = Bm[3 + j], c2 = Bm[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) Bm[3 * i + j] = A[3 * i] * c0 + A[3 * i + 1] * c1 + A[3 * i + 2] * c2;
  }
  const double p0 = Bm[9], p1 = Bm[10], p2 = Bm[11];
#pragma unroll
  for (int i = 0; i < 3; ++i) Bm[9 + i] = A[3 * i] * p0 + A[3 * i + 1] * p1 + A[3 * i + 2] * p2 + A[9 + i];
}

// C = A * B for 12-double poses (rotation row-major, translation)
__device__ inline void se3_mul(const double *A, const double *Bm, double *C) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
    C[9 + i] = A[3 * i] * Bm[9] + A[3 * i + 1] * Bm[10] + A[3 * i + 2] * Bm[11] + A[9 + i];
  }
}

// rotation about unit axis u by angle t (Rodrigues), row-major
__device__ inline void rot_axis(const double *u, double t, double *R) {
  double s, c;
  fast_sincos(t, s, c);
  const double v = 1.0 - c;
  R[0] = c + v * u[0] * u[0];
  R[1] = v * u[0] * u[1] - s * u[2];
  R[2] = v * u[0] * u[2] + s * u[1];
  R[3] = v * u[1] * u[0] + s * u[2];
  R[4] = c + v * u[1] * u[1];
  R[5] = v * u[1] * u[2] - s * u[0];
  R[6] = v * u[2] * u[0] - s * u[1];
  R[7] = v * u[2] * u[1] + s * u[0];
  R[8] = c + v * u[2] * u[2];
}

__device__ inline void quat_to_rot(const double *qv, double *R) {  // (x, y, z, w)
  const double n = 1.0 / sqrt(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2] + qv[3] * qv[3]);
  const double x = qv[0] * n, y = qv[1] * n, z = qv[2] * n, w = qv[3] * n;
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y - z * w);
  R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);
  R[7] = 2 * (y * z + x * w);
  R[8] = 1 - 2 * (x * x + y * y);
}

// local transform of joint a at configuration q.  Written without per-type branches around the output array:
// a prismatic joint is a rotation by zero plus a translation along the axis; merging three differently
// filled arrays at a join point had the compiler keep part of T in private memory (24 B of scratch).
__device__ inline void joint_transform(const ModelDev &m, int a, const double *qj, double *T) {
  const int t = m.jtype[a];  // qj = the joint's own entries of q (1, or 7 for the free-flyer)
  const double *ax = m.axis + 3 * a;
  const double qa = qj[0];
  rot_axis(ax, t == JOINT_REVOLUTE ? qa : 0.0, T);
  const double lin = t == JOINT_PRISMATIC ? qa : 0.0;
  T[9] = ax[0] * lin;
  T[10] = ax[1] * lin;
  T[11] = ax[2] * lin;
  if (t == JOINT_FREE_FLYER) {
    double R[9];
    quat_to_rot(qj + 3, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = R[i];
    T[9] = qa;
    T[10] = qj[1];
    T[11] = qj[2];
  }
}

// q_j <- q_j (+) v_j for one joint, in place (pink/configuration.py:273-293: pin.integrate)
__device__ inline void integrate_joint(const ModelDev &m, int j, double *q, const double *v) {
  if (m.jtype[j] != JOINT_FREE_FLYER) {
    q[0] += v[0];
    return;
  }
  // M <- M exp6(v): p += R V(w) v_lin, quat <- quat * exp(w / 2)
  double R[9];
  quat_to_rot(q + 3, R);
  const double wx = v[3], wy = v[4], wz = v[5];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double A, Bc;  // V = I + A [w]x + Bc [w]x^2
  if (th < 0.1) {
    // (1 - cos th) / th^2 and (th - sin th) / th^3 as series: a control step's rotation is ~1e-3 rad, where the closed
    // forms are good to 2e-10 and 1e-9 only; five terms hold 1e-16 below 0.1
    A = 0.5 + th2 * (-1.0 / 24.0 + th2 * (1.0 / 720.0 + th2 * (-1.0 / 40320.0 + th2 * (1.0 / 3628800.0))));
    Bc = 1.0 / 6.0 + th2 * (-1.0 / 120.0 + th2 * (1.0 / 5040.0 + th2 * (-1.0 / 362880.0 + th2 * (1.0 / 39916800.0))));
  } else {
    double s_t, c_t;
    fast_sincos(th, s_t, c_t);
    A = (1.0 - c_t) / th2;
    Bc = (th - s_t) / (th2 * th);
  }
  const double cx = wy * v[2] - wz * v[1], cy = wz * v[0] - wx * v[2], cz = wx * v[1] - wy * v[0];  // w x v
  const double ccx = wy * cz - wz * cy, ccy = wz * cx - wx * cz, ccz = wx * cy - wy * cx;          // w x (w x v)
  const double tx = v[0] + A * cx + Bc * ccx, ty = v[1] + A * cy + Bc * ccy, tz = v[2] + A * cz + Bc * ccz;
  q[0] += R[0] * tx + R[1] * ty + R[2] * tz;
  q[1] += R[3] * tx + R[4] * ty + R[5] * tz;
  q[2] += R[6] * tx + R[7] * ty + R[8] * tz;
  double s_h, c_h;  // unit quaternion of exp(w): (sin(th/2)/th w, cos(th/2))
  if (th < 1e-8) {
    s_h = 0.5;
    c_h = 1.0;
  } else {
    fast_sincos(0.5 * th, s_h, c_h);
    s_h /= th;
  }
  const double dx = s_h * wx, dy = s_h * wy, dz = s_h * wz, dw = c_h;
  const double x = q[3], y = q[4], z = q[5], w = q[6];
  double nx = w * dx + x * dw + y * dz - z * dy;
  double ny = w * dy - x * dz + y * dw + z * dx;
  double nz = w * dz + x * dy - y * dx + z * dw;
  double nw = w * dw - x * dx - y * dy - z * dz;
  const double n = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
  q[3] = nx * n;
  q[4] = ny * n;
  q[5] = nz * n;
  q[6] = nw * n;
}

// merged box of ConfigurationLimit + VelocityLimit for tangent coordinate j whose joint has the scalar
// configuration qi (configuration_limit.py:50-56, 111-120; velocity_limit.py:61-64, 118-120).  root_box (or NULL):
// lo[6], hi[6] for the tangent coordinates of the free-flyer -- the axis-aligned rows of a
// FloatingBaseVelocityLimit (floating_base_velocity_limit.py:104-148), which do not depend on q.
// acc: [3, nv] tables of an AccelerationLimit on the joints behind the root (acceleration_limit.py:158-199) -- a_max
// (0: no bound on that coordinate), Delta_q_prev, has_configuration_limit -- or NULL:
//   dq <= min(a dt^2 + dq_prev, dt sqrt(2 a (q_max - q))),  -dq <= min(a dt^2 - dq_prev, dt sqrt(2 a (q - q_min)))
// (the braking-distance term only where the joint has a configuration limit).
__device__ inline void coordinate_box(const ModelDev &m, int j, int jt, double qi, double dt, double gain, double &lo, double &hi,
                                      const double *root_box = nullptr, const double *acc = nullptr) {
  lo = -INFINITY;
  hi = INFINITY;
  if (m.jtype[jt] == JOINT_FREE_FLYER) {
    if (root_box) {
      const int sub = j - m.idx_v[jt];
      lo = root_box[sub];
      hi = root_box[6 + sub];
    }
    return;
  }
  const int iq = m.idx_q[jt];
  const double qmin = m.q_min[iq], qmax = m.q_max[iq], vmax = m.v_max[j];
  if (qmax < 1e20 && qmax > qmin + 1e-10) {
    lo = gain * (qmin - qi);
    hi = gain * (qmax - qi);
  }
  if (vmax < 1e20 && vmax > 1e-10) {
    lo = fmax(lo, -dt * vmax);
    hi = fmin(hi, dt * vmax);
  }
  if (acc) {
    const double am = acc[j];
    if (am > 0.0) {
      const double dqp = acc[m.nv + j];
      const bool cfg = acc[2 * m.nv + j] != 0.0;
      double up = am * dt * dt + dqp, lw = am * dt * dt - dqp;
      if (cfg) {
        up = fmin(up, dt * sqrt(2.0 * am * (qmax - qi)));
        lw = fmin(lw, dt * sqrt(2.0 * am * (qi - qmin)));
      }
      lo = fmax(lo, -lw);
      hi = fmin(hi, up);
    }
  }
}

struct FkArgs {
  ModelDev m;
  long long B;
  const double *q;   // [B, nq]
  double *T_frames;  // [B, nf, 12]      (optional in the fused kernel)
  double *J_body;    // [B, nf, 6, nv]   (unfused kernel only)
  // fused kernel: every model frame f carries a FrameTask with target T_target[b, f]; its error goes to
  // e_out[b * sE + 6 f ..], its Jacobian to rows 6 f .. 6 f + 5 of J_out[b * sJo + ...] (pitch nv), i.e.
  // straight into the packed e [B, K] / J [B, Kd, nv] streams of the solve kernel
  const double *T_target = nullptr;  // [B, nf, 12], or strided: pose (b, f) at T_target + b sTb + f sTf
  long long sTb = 0, sTf = 12;       // (sTb = 0: 12 nf)
  double *e_out = nullptr;
  double *J_out = nullptr;
  long long sE = 0, sJo = 0;
  // whole-step kernel (STEP = true): q is first advanced by the previous solve (q <- q (+) dq_prev for the
  // instances whose status is 0, written back in place), and the merged box limits + the posture error are
  // produced next to the frame-task rows -- one launch per control step besides the solve
  double *q_rw = nullptr;           // [B, nq] the same buffer as q, writable
  const double *dq_prev = nullptr;  // [B, nv] or NULL (first step)
  const int *status = nullptr;      // [B] status of the solve that produced dq_prev
  int *first_failure = nullptr;     // [B] sticky status | (step << 8), may be NULL
  int step = 0;
  double dt = 0.0, config_limit_gain = 0.5;
  const double *root_box = nullptr;  // [12] box of the free-flyer's tangent coordinates (coordinate_box), or NULL
  const double *acc_limit = nullptr;  // [3, nv] tables of an AccelerationLimit (coordinate_box), or NULL
  const double *q_target = nullptr;  // [B, nq] / [nq] posture target, NULL: no posture rows
  int target_batched = 0;
  double *lb = nullptr, *ub = nullptr;  // [B, nv]
  int e_off = 0;                        // posture rows go to e_out[b * sE + e_off ...]
};

// LDS per instance: oM [nj, 12] + inverse frame poses [nf, 12] + ancestor pointers [nj] + Jlog6 [nf, 36] + the
// scalar configuration of every joint [nj] (whole-step kernel)
__device__ __host__ inline int fk_lds_doubles(int nj, int nf) { return 12 * (nj + nf) + ((nj + 1) & ~1) + 36 * nf + ((nj + 1) & ~1); }

// Forward kinematics and body (LOCAL) frame Jacobians of one instance by a group of W lanes (W >= nj).
//   1. lane = joint: local transform (sin / cos), pose in the parent frame;
//   2. composition along the tree by pointer jumping: every joint multiplies its pose by its current
//      ancestor's and adopts that ancestor's ancestor, ceil(log2(depth)) rounds instead of a serial walk;
//   3. lane = frame: pose, inverse pose and -- FUSED -- the FrameTask error and Jlog6 of that frame;
//   4. lane = tangent column: the column of every frame Jacobian, stored (J_body) or -- FUSED -- multiplied
//      by -Jlog6 and written straight into the rows of the packed task Jacobian.
// FUSED = true moves 8 (nq + 12 nf + nf (6 + 6 nv)) bytes per instance instead of 8 (nq + 12 nf + 6 nf nv)
// written by the FK launch and 8 nf (24 + 12 nv + 6) re-read and written by nf frame-task launches.
// Where the fused results go.  HbmSink (default): e / J rows / lb / ub into the packed streams in HBM.  A sink
// with kKeep = true (ik_rollout.h) keeps them on chip for the solve that follows in the same kernel: the frame
// errors in LDS (es), the column's world twist + ancestor bits + box + posture error in its members.
struct HbmSink {
  static constexpr bool kKeep = false;
  double *es = nullptr;
  double lin[3], ang[3], lb, ub, post_e, qv;
  unsigned anc, ancr;
};

template <int W, bool FUSED = false, bool STEP = false, class Sink = HbmSink>
__device__ __forceinline__ void ik_fk_instance(const FkArgs &a, long long block, Sink *sink = nullptr, double *lds = nullptr) {
  constexpr int G = kWave / W;
  const ModelDev &m = a.m;
  const int lane = lane_id();
  const int g = lane / W, li = lane & (W - 1);
  long long b = block * G + g;
  const bool valid = b < a.B;
  if (!valid) b = a.B - 1;
  double *oM = lds ? lds : shared_base() + (long long)g * fk_lds_doubles(m.nj, m.nf);
  double *fMo = oM + 12 * m.nj;  // frame-from-world transforms
  int *anc = reinterpret_cast<int *>(fMo + 12 * m.nf);
  double *Jls = fMo + 12 * m.nf + ((m.nj + 1) & ~1);
  double *qs = Jls + 36 * m.nf;  // scalar configuration of every joint (STEP)
  const double *q = a.q + b * (long long)m.nq;

  // The tables the later steps need for THIS lane's first frame / tangent column are requested now: the
  // barriers of steps 1-2 would otherwise keep these loads (model tables in L2, targets in HBM) from starting
  // before the poses are composed, and the kernel is bound by exactly such dependent round trips.
  int pf_fj = -1, pf_jt = 0, pf_sub = 0, pf_ty = 0;
  double pf_FP[12], pf_Tt[12], pf_ax[3];
  unsigned pf_anc = 0;   // bit f: joint of column li is an ancestor of frame f (first 32 frames)
  unsigned pf_ancr = 0;  // bit f: ... of the root frame of relative slot f
  int pf_rj = -2;
  if constexpr (FUSED) {
    const int f0 = li < m.nf ? li : 0;
    pf_fj = m.frame_joint[f0];
    pf_rj = m.frame_root_joint[f0];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      pf_FP[i] = m.frame_placement[12 * f0 + i];
      pf_Tt[i] = a.T_target[b * (a.sTb ? a.sTb : 12LL * m.nf) + f0 * a.sTf + i];
    }
    const int j0 = li < m.nv ? li : 0;
    pf_jt = m.dof_joint[j0];
    pf_sub = m.dof_sub[j0];
    pf_ty = m.jtype[pf_jt];
#pragma unroll
    for (int i = 0; i < 3; ++i) pf_ax[i] = m.axis[3 * pf_jt + i];
    for (int f = 0; f < m.nf && f < 32; ++f) {
      pf_anc |= (m.anc[f * m.nj + pf_jt] != 0 ? 1u : 0u) << f;
      pf_ancr |= (m.ancr[f * m.nj + pf_jt] != 0 ? 1u : 0u) << f;
    }
  }

  // 1. pose of joint li in its parent's frame
  const bool isj = li < m.nj;
  const int jl = isj ? li : 0;
  double T[12];
  if constexpr (STEP) {
    // 0. the joint's configuration in registers, advanced by the previous solve unless that solve failed (the
    //    reference raises NoSolutionFound before integrating, pink/solve_ik.py:271-275)
    const int iq = m.idx_q[jl];
    const bool ff = m.jtype[jl] == JOINT_FREE_FLYER;
    double qj[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) qj[i] = (i == 0 || ff) ? q[iq + i] : 0.0;
    if (a.dq_prev) {
      const int st = a.status[b];
      if (st == 0) {
        integrate_joint(m, jl, qj, a.dq_prev + b * (long long)m.nv + m.idx_v[jl]);
        if (valid && isj) {
          double *qw = a.q_rw + b * (long long)m.nq + iq;
#pragma unroll
          for (int i = 0; i < 7; ++i)
            if (i == 0 || ff) qw[i] = qj[i];
        }
      } else if (valid && li == 0 && a.first_failure && a.first_failure[b] == 0) {
        a.first_failure[b] = st | (a.step << 8);
      }
    }
    if (isj) qs[li] = qj[0];
    double Tj[12];
    joint_transform(m, jl, qj, Tj);
    se3_mul(m.placement + 12 * jl, Tj, T);
  } else {
    double Tj[12];
    joint_transform(m, jl, q + m.idx_q[jl], Tj);
    se3_mul(m.placement + 12 * jl, Tj, T);
  }
  int up = isj ? m.parent[jl] : -1;
  if (isj) {
#pragma unroll
    for (int i = 0; i < 12; ++i) oM[12 * li + i] = T[i];
    anc[li] = up;
  }
  wave_sync();
  // 2. pointer jumping: T_j <- T_up(j) T_j, up(j) <- up(up(j)) until every joint is expressed in the world
  while (wave_any(up >= 0)) {
    double P[12];
    int upup = -1;
    if (up >= 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) P[i] = oM[12 * up + i];
      upup = anc[up];
    }
    wave_sync();  // every lane has read the old poses and pointers
    if (up >= 0) {
      double N[12];
      se3_mul(P, T, N);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        T[i] = N[i];
        oM[12 * li + i] = N[i];
      }
      anc[li] = upup;
      up = upup;
    }
    wave_sync();
  }
  if constexpr (FUSED) {
    // A relative slot (relative_frame_task.py:142-231): the target is given in the root frame r.  With the target
    // carried into the world by the root's current pose, T_t' = T_0r T_rt, the task's error log(T_rt^-1 T_rf) is
    // log(T_t'^-1 T_f) = -log(T_f^-1 T_t') and its Jacobian Jlog6(T_tf) (fJ_0f - Ad(T_fr) rJ_0r) is
    // Jlog6(T_t'^-1 T_f) X_f^-1 [lin; ang] ([j anc. of f] - [j anc. of r]) -- Ad(T_fr) X_r^-1 = X_f^-1: the rows of an
    // ordinary FrameTask on T_t', negated, with a signed ancestor indicator.  J and e change sign together: the
    // same H and c (task.py:145-167).  The target is composed here, in place and column by column, before the frame
    // loop below holds five poses in registers (inside it the whole-step kernel spilled ninety registers more);
    // relative slots are among the first W frames (model_tables.h).
    if (wave_any(pf_rj != -2)) {  // (rare: wave-uniform)
      if (li < m.nf && pf_rj != -2) {
        se3_lmul(m.frame_root_placement + 12 * li, pf_Tt);
        if (pf_rj >= 0) se3_lmul(oM + 12 * pf_rj, pf_Tt);
      }
    }
    // 3. lane = frame: pose, FrameTask error, and the two 3 x 3 blocks that turn a WORLD twist [lin; ang] of a
    //    joint into the six rows of the task Jacobian:  J_task = -Jlog6(T_t^-1 T_f) X_f^-1 [lin; ang]  with
    //    X_f^-1 = [[R_f^T, -R_f^T [p_f]x], [0, R_f^T]] and Jlog6 = [[A, C A], [0, A]]  gives
    //    rows 0..2 = -(U lin + V ang), rows 3..5 = -U ang,  U = A R_f^T,  V = (C A) R_f^T - U [p_f]x.
    //    log3 is evaluated once: log(T_t^-1 T_f) has the rotation vector of log(T_f^-1 T_t) negated.
    for (int f = li; f < m.nf; f += W) {
      double F[12], FP[12], Tt[12];
      const bool first = f == li;  // prefetched above
      const int fj = first ? pf_fj : m.frame_joint[f];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        FP[i] = first ? pf_FP[i] : m.frame_placement[12 * f + i];
        Tt[i] = first ? pf_Tt[i] : a.T_target[b * (a.sTb ? a.sTb : 12LL * m.nf) + f * a.sTf + i];
      }
      if (fj >= 0) {
        se3_mul(oM + 12 * fj, FP, F);
      } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) F[i] = FP[i];
      }
      if (valid && a.T_frames) {
#pragma unroll
        for (int i = 0; i < 12; ++i) a.T_frames[(b * m.nf + f) * 12 + i] = F[i];
      }
      // world position of the frame, for the rows of position barriers formed on chip (ik_rollout.h)
      fMo[12 * f + 9] = F[9];
      fMo[12 * f + 10] = F[10];
      fMo[12 * f + 11] = F[11];
      double R1[9], p1[3], w[3], th;
      se3_act_inv(F, Tt, R1, p1);  // T_f^-1 T_t
      log3(R1, w, th);
      {
        double xi[6];
        log6_from_log3(w, th, p1, xi);  // e = log6(T_frame^-1 T_target), frame_task.py:181-193
        if constexpr (Sink::kKeep) {
#pragma unroll
          for (int i = 0; i < 6; ++i) sink->es[6 * f + i] = xi[i];
        } else if (valid) {
#pragma unroll
          for (int i = 0; i < 6; ++i) a.e_out[b * a.sE + 6 * f + i] = xi[i];
        }
      }
      // T_t^-1 T_f = (R1^T, -R1^T p1), rotation vector -w
      double p2[3], w2[3], Jl[36];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        p2[i] = -(R1[i] * p1[0] + R1[3 + i] * p1[1] + R1[6 + i] * p1[2]);
        w2[i] = -w[i];
      }
      jlog6_from_log3(w2, th, p2, Jl);  // frame_task.py:222-227
      double *UV = Jls + 36 * f;  // U (9) then V (9)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double U[3], CR[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // times R_f^T: column k of R_f^T is row k of R_f
          U[k] = Jl[6 * i] * F[3 * k] + Jl[6 * i + 1] * F[3 * k + 1] + Jl[6 * i + 2] * F[3 * k + 2];
          CR[k] = Jl[6 * i + 3] * F[3 * k] + Jl[6 * i + 4] * F[3 * k + 1] + Jl[6 * i + 5] * F[3 * k + 2];
        }
        // (U [p]x)[k] = sum_l U[l] P[l][k],  [p]x = [[0, -pz, py], [pz, 0, -px], [-py, px, 0]]
        UV[3 * i] = U[0];
        UV[3 * i + 1] = U[1];
        UV[3 * i + 2] = U[2];
        UV[9 + 3 * i] = CR[0] - (U[1] * F[11] - U[2] * F[10]);
        UV[9 + 3 * i + 1] = CR[1] - (U[2] * F[9] - U[0] * F[11]);
        UV[9 + 3 * i + 2] = CR[2] - (U[0] * F[10] - U[1] * F[9]);
      }
    }
    wave_sync();
    // 4. lane = tangent column: world twist of the column's joint axis, then 27 FMAs per ancestor frame
    for (int j = li; j < m.nv; j += W) {
      const bool first = j == li;  // prefetched above
      const int jt = first ? pf_jt : m.dof_joint[j], sub = first ? pf_sub : m.dof_sub[j], ty = first ? pf_ty : m.jtype[jt];
      const double *Xj = oM + 12 * jt;  // world pose of the joint
      double ax[3], u[3];
      if (ty == JOINT_FREE_FLYER) {
        const int k = sub % 3;
        ax[0] = k == 0 ? 1.0 : 0.0;
        ax[1] = k == 1 ? 1.0 : 0.0;
        ax[2] = k == 2 ? 1.0 : 0.0;
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) ax[i] = first ? pf_ax[i] : m.axis[3 * jt + i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) u[i] = Xj[3 * i] * ax[0] + Xj[3 * i + 1] * ax[1] + Xj[3 * i + 2] * ax[2];
      const bool angular = (ty == JOINT_REVOLUTE) || (ty == JOINT_FREE_FLYER && sub >= 3);
      double lin[3], ang[3];
      if (angular) {  // axis through the joint origin: [p x u; u]
        lin[0] = Xj[10] * u[2] - Xj[11] * u[1];
        lin[1] = Xj[11] * u[0] - Xj[9] * u[2];
        lin[2] = Xj[9] * u[1] - Xj[10] * u[0];
        ang[0] = u[0], ang[1] = u[1], ang[2] = u[2];
      } else {
        lin[0] = u[0], lin[1] = u[1], lin[2] = u[2];
        ang[0] = ang[1] = ang[2] = 0.0;
      }
      if constexpr (Sink::kKeep) {  // the rows are formed during the stacking of the solve (W >= nv: one column per lane)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          sink->lin[i] = lin[i];
          sink->ang[i] = ang[i];
        }
        sink->anc = pf_anc;
        sink->ancr = pf_ancr;
      }
      for (int f = 0; f < (Sink::kKeep ? 0 : m.nf); ++f) {
        // (+1: ancestor of the frame only, -1: of the root frame of a relative slot only, 0: of both or neither)
        const int on = (first && f < 32) ? (int)((pf_anc >> f) & 1u) - (int)((pf_ancr >> f) & 1u)
                                         : (int)(m.anc[f * m.nj + jt] != 0) - (int)(m.ancr[f * m.nj + jt] != 0);
        const double *UV = Jls + 36 * f;
        double *Jo = a.J_out + b * a.sJo + (long long)(6 * f) * m.nv;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double top = UV[3 * i] * lin[0] + UV[3 * i + 1] * lin[1] + UV[3 * i + 2] * lin[2] +
                             UV[9 + 3 * i] * ang[0] + UV[9 + 3 * i + 1] * ang[1] + UV[9 + 3 * i + 2] * ang[2];
          const double bot = UV[3 * i] * ang[0] + UV[3 * i + 1] * ang[1] + UV[3 * i + 2] * ang[2];
          if (valid) {
            Jo[i * m.nv + j] = on ? (on > 0 ? -top : top) : 0.0;
            Jo[(i + 3) * m.nv + j] = on ? (on > 0 ? -bot : bot) : 0.0;
          }
        }
      }
    if constexpr (STEP) {
      // 5. merged box limits and the posture error of tangent coordinate j (qs was published before step 2's barriers)
      const double qi = qs[jt];
      double lo, hi;
      coordinate_box(m, j, jt, qi, a.dt, a.config_limit_gain, lo, hi, a.root_box, a.acc_limit);
      if constexpr (Sink::kKeep) {
        sink->lb = lo;
        sink->ub = hi;
        sink->post_e = 0.0;
        sink->qv = qi;  // (the coordinate's own scalar configuration: errors of constant-row tasks, ik_rollout.h)
        if (a.q_target && ty != JOINT_FREE_FLYER && j >= m.root_nv) {
          const int iq = m.idx_q[jt];
          sink->post_e = qi - (a.target_batched ? a.q_target[b * m.nq + iq] : a.q_target[iq]);
        }
      } else if (valid) {
        a.lb[b * m.nv + j] = lo;
        a.ub[b * m.nv + j] = hi;
        if (a.q_target && ty != JOINT_FREE_FLYER && j >= m.root_nv) {  // posture_task.py:100-107: q (-) q*
          const int iq = m.idx_q[jt];
          const double qt = a.target_batched ? a.q_target[b * m.nq + iq] : a.q_target[iq];
          a.e_out[b * a.sE + a.e_off + (j - m.root_nv)] = qi - qt;
        }
      }
    }
    }
  } else {
  // 3. frames
  for (int f = li; f < m.nf; f += W) {
    double F[12];
    const int fj = m.frame_joint[f];
    if (fj >= 0) {
      se3_mul(oM + 12 * fj, m.frame_placement + 12 * f, F);
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) F[i] = m.frame_placement[12 * f + i];
    }
    if (valid && a.T_frames) {
#pragma unroll
      for (int i = 0; i < 12; ++i) a.T_frames[(b * m.nf + f) * 12 + i] = F[i];
    }
    // inverse: (R^T, -R^T p)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) fMo[12 * f + 3 * i + j] = F[3 * j + i];
      fMo[12 * f + 9 + i] = -(F[i] * F[9] + F[3 + i] * F[10] + F[6 + i] * F[11]);
    }
    if constexpr (FUSED) {
      double Tt[12], R[9], pr[3], xi[6], Jl[36];
#pragma unroll
      for (int i = 0; i < 12; ++i) Tt[i] = a.T_target[b * (a.sTb ? a.sTb : 12LL * m.nf) + f * a.sTf + i];
      se3_act_inv(F, Tt, R, pr);  // e = log6(T_frame^-1 T_target), frame_task.py:181-193
      log6(R, pr, xi);
      if (valid) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.e_out[b * a.sE + 6 * f + i] = xi[i];
      }
      se3_act_inv(Tt, F, R, pr);  // J = -Jlog6(T_target^-1 T_frame) J_body, frame_task.py:222-227
      jlog6(R, pr, Jl);
#pragma unroll
      for (int i = 0; i < 36; ++i) Jls[36 * f + i] = Jl[i];
    }
  }
  wave_sync();
  // 4. body Jacobians: lane = tangent column
  for (int j = li; j < m.nv; j += W) {
    const int jt = m.dof_joint[j], sub = m.dof_sub[j], ty = m.jtype[jt];
    for (int f = 0; f < m.nf; ++f) {
      double col[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (m.anc[f * m.nj + jt]) {
        double X[12];
        se3_mul(fMo + 12 * f, oM + 12 * jt, X);  // joint frame -> frame
        double u[3];
        if (ty == JOINT_FREE_FLYER) {
          const int k = sub % 3;  // R[:, k], selected without indexing the register array dynamically (scratch)
          u[0] = k == 0 ? X[0] : (k == 1 ? X[1] : X[2]);
          u[1] = k == 0 ? X[3] : (k == 1 ? X[4] : X[5]);
          u[2] = k == 0 ? X[6] : (k == 1 ? X[7] : X[8]);
        } else {
          const double *ax = m.axis + 3 * jt;
#pragma unroll
          for (int i = 0; i < 3; ++i) u[i] = X[3 * i] * ax[0] + X[3 * i + 1] * ax[1] + X[3 * i + 2] * ax[2];
        }
        const bool angular = (ty == JOINT_REVOLUTE) || (ty == JOINT_FREE_FLYER && sub >= 3);
        if (angular) {  // [p x (R u); R u]
          col[0] = X[10] * u[2] - X[11] * u[1];
          col[1] = X[11] * u[0] - X[9] * u[2];
          col[2] = X[9] * u[1] - X[10] * u[0];
          col[3] = u[0];
          col[4] = u[1];
          col[5] = u[2];
        } else {  // [R u; 0]
          col[0] = u[0];
          col[1] = u[1];
          col[2] = u[2];
        }
      }
      if (valid) {
        if constexpr (FUSED) {
          double *Jo = a.J_out + b * a.sJo + (long long)(6 * f) * m.nv;
          const double *Jl = Jls + 36 * f;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) sacc -= Jl[6 * i + r] * col[r];
            Jo[i * m.nv + j] = sacc;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 6; ++r) a.J_body[((b * m.nf + f) * 6 + r) * m.nv + j] = col[r];
        }
      }
    }
    }
  }
}

template <int W>
__global__ void __launch_bounds__(kWave) ik_fk_kernel(FkArgs a) {
  ik_fk_instance<W, false>(a, block_id());
}
template <int W>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_FK
ik_fk_frame_tasks_kernel(FkArgs a) {
  ik_fk_instance<W, true>(a, block_id());
}
template <int W>
__global__ void __launch_bounds__(kWave) PINKHIP_OCCUPANCY_FK
ik_step_kernel(FkArgs a) {
  ik_fk_instance<W, true, true>(a, block_id());
}

struct LimitsPostureArgs {
  ModelDev m;
  long long B;
  double dt, config_limit_gain;
  const double *q;         // [B, nq]
  const double *q_target;  // [B, nq] or [nq] (target_batched)
  int target_batched;
  double *lb, *ub;         // [B, nv]
  double *e;               // [B, K]: posture rows written at e_off .. e_off + nv - root_nv, or NULL
  int K, e_off;
};

// one thread per (instance, tangent coordinate)
__device__ inline void ik_limits_posture_thread(const LimitsPostureArgs &a, long long t) {
  const ModelDev &m = a.m;
  if (t >= a.B * m.nv) return;
  const long long b = t / m.nv;
  const int j = (int)(t - b * m.nv);
  const int jt = m.dof_joint[j];
  const double qi = a.q[b * m.nq + m.idx_q[jt]];
  double lo, hi;
  coordinate_box(m, j, jt, qi, a.dt, a.config_limit_gain, lo, hi);
  if (m.jtype[jt] != JOINT_FREE_FLYER && a.e && j >= m.root_nv) {  // posture_task.py:100-107: q (-) q* on the actuated coordinates
    const int iq = m.idx_q[jt];
    const double qt = a.target_batched ? a.q_target[b * m.nq + iq] : a.q_target[iq];
    a.e[b * a.K + a.e_off + (j - m.root_nv)] = qi - qt;
  }
  a.lb[b * m.nv + j] = lo;
  a.ub[b * m.nv + j] = hi;
}

#ifndef PINKHIP_NO_ELEMENTWISE_KERNELS  // (non-template kernels: defined by the host translation unit only)
__global__ void __launch_bounds__(256) ik_limits_posture_kernel(LimitsPostureArgs a) {
  ik_limits_posture_thread(a, (long long)blockIdx.x * 256 + threadIdx.x);
}
#endif

struct CheckLimitsArgs {
  ModelDev m;
  long long B;
  const double *q;  // [B, nq]
  double tol;
  int start;        // configuration entries of the root joint, skipped (configuration.py:183-187)
  long long *first_bad;  // device: min over the violating entries of b nq + i, or LLONG_MAX
};

// one thread per (instance, configuration entry)
__device__ inline void ik_check_limits_thread(const CheckLimitsArgs &a, long long t) {
  const ModelDev &m = a.m;
  if (t >= a.B * m.nq) return;
  const int i = (int)(t % m.nq);
  if (i < a.start) return;
  const double lo = m.q_min[i], up = m.q_max[i], qi = a.q[t];
  if (up > lo + a.tol && (qi < lo - a.tol || qi > up + a.tol))
    atomicMin(reinterpret_cast<unsigned long long *>(a.first_bad), static_cast<unsigned long long>(t));
}

#ifndef PINKHIP_NO_ELEMENTWISE_KERNELS
__global__ void __launch_bounds__(256) ik_check_limits_kernel(CheckLimitsArgs a) {
  ik_check_limits_thread(a, (long long)blockIdx.x * 256 + threadIdx.x);
}
#endif

struct IntegrateArgs {
  ModelDev m;
  long long B;
  double *q;         // [B, nq] in place
  const double *dq;  // [B, nv]
  const int *status = nullptr;   // [B] solver status of this step: instances with status != 0 keep their q
  int *first_failure = nullptr;  // [B] sticky: status | (step << 8) of the first failing step
  int step = 0;
};

// one thread per (instance, joint): q <- q (+) dq
__device__ inline void ik_integrate_thread(const IntegrateArgs &a, long long t) {
  const ModelDev &m = a.m;
  if (t >= a.B * m.nj) return;
  const long long b = t / m.nj;
  const int j = (int)(t - b * m.nj);
  if (a.status) {  // a failed solve is never applied (pink/solve_ik.py:271-275 raises before integrating)
    const int st = a.status[b];
    if (st != 0) {
      if (j == 0 && a.first_failure && a.first_failure[b] == 0) a.first_failure[b] = st | (a.step << 8);
      return;
    }
  }
  integrate_joint(m, j, a.q + b * m.nq + m.idx_q[j], a.dq + b * m.nv + m.idx_v[j]);
}

#ifndef PINKHIP_NO_ELEMENTWISE_KERNELS
__global__ void __launch_bounds__(256) ik_integrate_kernel(IntegrateArgs a) {
  ik_integrate_thread(a, (long long)blockIdx.x * 256 + threadIdx.x);
}

// ----------------------------------------------------------------------------------------------------------------
// ik_pose_targets_kernel: FrameTask targets handed over as translation + unit quaternion, [B, 7] =
// (tx, ty, tz, qx, qy, qz, qw) -- the order of pin.SE3ToXYZQUAT -- written out as the [B, 12] poses (rotation row-major,
// translation) the kinematics kernels read: 56 B per target across PCIe instead of 96 B.  What the caller of
// FrameTask.set_target does on the host per robot (pink/tasks/frame_task.py:129-137 takes the pin.SE3 that
// pin.XYZQUATToSE3 built); the quaternion is normalised here.
struct PoseTargetsArgs {
  long long B;
  const double *pq;  // [B, 7]
  double *T;         // [B, 12]
};

__device__ inline void ik_pose_targets_thread(const PoseTargetsArgs &a, long long t) {
  if (t >= a.B) return;
  const double *s = a.pq + 7 * t;
  double x = s[3], y = s[4], z = s[5], w = s[6];
  const double n = 1.0 / sqrt(x * x + y * y + z * z + w * w);
  x *= n, y *= n, z *= n, w *= n;
  double *o = a.T + 12 * t;
  o[0] = 1.0 - 2.0 * (y * y + z * z);
  o[1] = 2.0 * (x * y - z * w);
  o[2] = 2.0 * (x * z + y * w);
  o[3] = 2.0 * (x * y + z * w);
  o[4] = 1.0 - 2.0 * (x * x + z * z);
  o[5] = 2.0 * (y * z - x * w);
  o[6] = 2.0 * (x * z - y * w);
  o[7] = 2.0 * (y * z + x * w);
  o[8] = 1.0 - 2.0 * (x * x + y * y);
  o[9] = s[0], o[10] = s[1], o[11] = s[2];
}

__global__ void __launch_bounds__(256) ik_pose_targets_kernel(PoseTargetsArgs a) {
  ik_pose_targets_thread(a, (long long)blockIdx.x * 256 + threadIdx.x);
}
#endif

}  // namespace pinkhip
