// Batched FrameTask error and Jacobian (reference pink/tasks/frame_task.py:148-227):
//
//   e = log6(T_frame^-1 T_target)                        body twist [linear; angular]
//   J = -Jlog6(T_target^-1 T_frame) * J_body             J_body = frame Jacobian in the LOCAL frame
//
// The reference evaluates these per instance through Pinocchio (pin.log, pin.Jlog6) and a 6x6 by
// 6xnv NumPy product.  Here a group of W lanes handles one instance: every lane evaluates the
// (group-uniform) SE(3) quantities redundantly from the 2 x 12 pose entries, lane j owns column j
// of the 6 x nv Jacobian: six coalesced loads, a 6 x 6 by 6 product in registers, six coalesced
// stores.  The kernel is a pure HBM stream: 8 (24 + 12 nv + 6) bytes per instance.
//
// Poses are 12 doubles: the rotation row-major (9), then the translation (3).  Formulas: SURVEY.md
// appendix B.3 (closed forms of log3 / log6 / Jlog3 / Jlog6 with their small-angle series).
#pragma once

#include "ik_common.h"

namespace pinkhip {

struct FrameTaskArgs {
  long long B;
  int nv;
  const double *T_frame;   // [B, 12] frame-to-world
  const double *T_target;  // [B, 12] target-to-world
  const double *J_body;    // [B, 6, nv]
  double *e_out;           // [B, 6]
  double *J_out;           // [B, 6, nv]
  // strides between consecutive instances, in doubles (0 = densely packed as documented above);
  // lets the kernel read one frame out of [B, nf, ...] arrays and write straight into the rows
  // of the packed e [B, K] / J [B, Kd, nv] streams of the solve kernel
  long long sTf = 0, sTt = 0, sJb = 0, sE = 0, sJo = 0;
};

// A^T B for rotations stored row-major, and A^T (pb - pa)
__device__ inline void se3_act_inv(const double *A, const double *Bm, double *R, double *p) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * i + j] = A[i] * Bm[j] + A[3 + i] * Bm[3 + j] + A[6 + i] * Bm[6 + j];
    p[i] = A[i] * (Bm[9] - A[9]) + A[3 + i] * (Bm[10] - A[10]) + A[6 + i] * (Bm[11] - A[11]);
  }
}

// rotation vector of R (row-major); theta in [0, pi].  theta = atan2(|v|/2, (tr R - 1)/2) is well
// conditioned on the whole range (acos is not next to pi); close to pi the axis comes from the
// symmetric part c I + (1 - c) a a^T, as Pinocchio does.
__device__ inline void log3(const double *R, double *w, double &theta) {
  const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
  const double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  const double s = 0.5 * sqrt(vx * vx + vy * vy + vz * vz);
  theta = atan2(s, c);
  if (theta < 1e-8) {
    w[0] = 0.5 * vx;
    w[1] = 0.5 * vy;
    w[2] = 0.5 * vz;
  } else if (3.14159265358979323846 - theta < 1e-2) {
    const double one_c = 1.0 - c;
    double ax, ay, az;
    if (R[0] >= R[4] && R[0] >= R[8]) {
      ax = sqrt(fmax((R[0] - c) / one_c, 0.0));
      ay = (R[1] + R[3]) / (2.0 * one_c * ax);
      az = (R[2] + R[6]) / (2.0 * one_c * ax);
    } else if (R[4] >= R[8]) {
      ay = sqrt(fmax((R[4] - c) / one_c, 0.0));
      ax = (R[1] + R[3]) / (2.0 * one_c * ay);
      az = (R[5] + R[7]) / (2.0 * one_c * ay);
    } else {
      az = sqrt(fmax((R[8] - c) / one_c, 0.0));
      ax = (R[2] + R[6]) / (2.0 * one_c * az);
      ay = (R[5] + R[7]) / (2.0 * one_c * az);
    }
    const double sg = (ax * vx + ay * vy + az * vz < 0.0) ? -1.0 : 1.0;
    w[0] = sg * theta * ax;
    w[1] = sg * theta * ay;
    w[2] = sg * theta * az;
  } else {
    const double k = theta / (2.0 * s);
    w[0] = k * vx;
    w[1] = k * vy;
    w[2] = k * vz;
  }
}

// beta(th) = 1 / th^2 - sin th / (2 th (1 - cos th)) = (1 - (th / 2) cot(th / 2)) / th^2 and beta_dot = beta'(th) / th as power
// series in th^2 (|B_2n| / (2n)! and 2 k times them) below kSeriesTh: the closed forms subtract numbers of order 1 / th^2
// and 1 / th^4 -- at th = 1e-3 they are wrong by 2e-4 and 1e4 relative, and one ulp of difference in sin / cos between
// two implementations of the same formula became 1e-8 relative on dq (scripts/gpu_fuzz_rollout.py, round 4: a tracking
// controller's orientation errors ARE that small).  Eight terms hold 1e-16 up to th = 0.6 (checked against 60 digits);
// same constants as pink_amd/lie.py.
constexpr double kSeriesTh = 0.5;
__device__ inline double beta_series(double t2) {
  return 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0 + t2 * (1.0 / 47900160.0 +
         t2 * (691.0 / 1307674368000.0 + t2 * (1.0 / 74724249600.0 + t2 * (3617.0 / 10670622842880000.0)))))));
}
__device__ inline double beta_dot_series(double t2) {
  return 2.0 / 720.0 + t2 * (4.0 / 30240.0 + t2 * (6.0 / 1209600.0 + t2 * (8.0 / 47900160.0 +
         t2 * (10.0 * 691.0 / 1307674368000.0 + t2 * (12.0 / 74724249600.0 + t2 * (14.0 * 3617.0 / 10670622842880000.0 +
         t2 * 1.3737699290044551303e-13))))));
}
__device__ inline void alpha_beta(double th, double &alpha, double &beta) {
  if (th < kSeriesTh) {
    beta = beta_series(th * th);
    alpha = 1.0 - th * th * beta;
  } else {
    double s, c;
    fast_sincos(th, s, c);
    alpha = th * s / (2.0 * (1.0 - c));
    beta = 1.0 / (th * th) - s / (2.0 * th * (1.0 - c));
  }
}

// twist [v; w] with exp6 = (R, p), given w = log3(R) and its norm
__device__ inline void log6_from_log3(const double *w, double th, const double *p, double *xi) {
  double alpha, beta;
  alpha_beta(th, alpha, beta);
  const double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  xi[0] = alpha * p[0] - 0.5 * (w[1] * p[2] - w[2] * p[1]) + beta * wp * w[0];
  xi[1] = alpha * p[1] - 0.5 * (w[2] * p[0] - w[0] * p[2]) + beta * wp * w[1];
  xi[2] = alpha * p[2] - 0.5 * (w[0] * p[1] - w[1] * p[0]) + beta * wp * w[2];
  xi[3] = w[0];
  xi[4] = w[1];
  xi[5] = w[2];
}
__device__ inline void log6(const double *R, const double *p, double *xi) {
  double w[3], th;
  log3(R, w, th);
  log6_from_log3(w, th, p, xi);
}

// right Jacobian of log6 at (R, p), row-major 6 x 6: [[A, C A], [0, A]], given w = log3(R) and its norm
__device__ inline void jlog6_from_log3(const double *w, double th, const double *p, double *Jl) {
  double a, d, beta, beta_dot;
  if (th < kSeriesTh) {
    a = beta_series(th * th);
    d = 1.0 - th * th * a;
    beta = a;
    beta_dot = beta_dot_series(th * th);
  } else {
    double s, c;
    fast_sincos(th, s, c);
    a = 1.0 / (th * th) - s / (2.0 * th * (1.0 - c));
    d = 0.5 * th * s / (1.0 - c);
    beta = a;
    beta_dot = -2.0 / (th * th * th * th) + (1.0 + s / th) / (2.0 * th * th * (1.0 - c));
  }
  double A[9], C[9];
  const double hx[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  const double px[9] = {0.0, -p[2], p[1], p[2], 0.0, -p[0], -p[1], p[0], 0.0};
  const double wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  double v3[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v3[i] = beta_dot * wp * w[i] - (th * th * beta_dot + 2.0 * beta) * p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[3 * i + j] = a * w[i] * w[j] + (i == j ? d : 0.0) + 0.5 * hx[3 * i + j];
      C[3 * i + j] = v3[i] * w[j] + beta * w[i] * p[j] + (i == j ? beta * wp : 0.0) + 0.5 * px[3 * i + j];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Jl[6 * i + j] = A[3 * i + j];
      Jl[6 * i + 3 + j] = C[3 * i] * A[j] + C[3 * i + 1] * A[3 + j] + C[3 * i + 2] * A[6 + j];
      Jl[6 * (i + 3) + j] = 0.0;
      Jl[6 * (i + 3) + 3 + j] = A[3 * i + j];
    }
}
__device__ inline void jlog6(const double *R, const double *p, double *Jl) {
  double w[3], th;
  log3(R, w, th);
  jlog6_from_log3(w, th, p, Jl);
}

template <int W>
__device__ inline void ik_frame_task_instance(const FrameTaskArgs &a, long long block) {
  constexpr int G = kWave / W;
  const int lane = lane_id();
  const int g = lane / W, li = lane & (W - 1);
  const long long b = block * G + g;
  if (b >= a.B) return;  // no cross-lane primitive below: early exit is safe
  const int nv = a.nv;
  const long long sTf = a.sTf ? a.sTf : 12, sTt = a.sTt ? a.sTt : 12, sJb = a.sJb ? a.sJb : 6LL * nv;
  const long long sE = a.sE ? a.sE : 6, sJo = a.sJo ? a.sJo : 6LL * nv;
  double Tf[12], Tt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    Tf[i] = a.T_frame[b * sTf + i];
    Tt[i] = a.T_target[b * sTt + i];
  }
  double R[9], p[3];
  if (li == 0) {  // e = log6(T_frame^-1 T_target), frame_task.py:181-193
    double xi[6];
    se3_act_inv(Tf, Tt, R, p);
    log6(R, p, xi);
#pragma unroll
    for (int i = 0; i < 6; ++i) a.e_out[b * sE + i] = xi[i];
  }
  // J = -Jlog6(T_target^-1 T_frame) J_body, frame_task.py:222-227; lane = column
  se3_act_inv(Tt, Tf, R, p);
  double Jl[36];
  jlog6(R, p, Jl);
  const double *Jb = a.J_body + b * sJb;
  double *Jo = a.J_out + b * sJo;
  for (int j = li; j < nv; j += W) {
    double col[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = Jb[r * nv + j];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) s -= Jl[6 * i + r] * col[r];
      Jo[i * nv + j] = s;
    }
  }
}

template <int W>
__global__ void __launch_bounds__(kWave) ik_frame_task_kernel(FrameTaskArgs a) {
  ik_frame_task_instance<W>(a, block_id());
}

}  // namespace pinkhip
