// Host-side validation of a pinkhip_desc and expansion of its task list into the
// small broadcast tables the kernels read (per-row gain / lm_damping, diagonal
// task map).  Plain C++ (no HIP) so that the CPU wave emulator used by the test
// suite shares it with the real library.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pinkhip.h"

namespace pinkhip {

struct HostTables {
  std::vector<double> row_gain, row_lm;
  std::vector<int32_t> dtask_col0, dtask_row0, dtask_k;
  std::vector<int32_t> barrier_rows;
  std::vector<double> barrier_safe_gain;
};

// Padded kernel dimension for a tangent dimension nv (0 if unsupported).
inline int padded_nv(int nv) {
  static const int sizes[] = {8, 16, 24, 32, 40, 48, 56, 64};
  for (int s : sizes)
    if (nv <= s) return s;
  return 0;
}

// Returns an empty string when `d` is well-formed, otherwise what is wrong.
inline std::string build_tables(const pinkhip_desc &d, HostTables &t) {
  if (d.B < 0) return "B must be >= 0";
  if (d.nv < 1 || d.nv > PINKHIP_MAX_NV) return "nv must be in 1..PINKHIP_MAX_NV";
  if (d.md < 0 || d.md > PINKHIP_MAX_MD)
    return "md must be in 0..PINKHIP_MAX_MD = " + std::to_string(PINKHIP_MAX_MD) +
           " dense rows per instance (one lane of the QP's group per row); got " + std::to_string(d.md) +
           ": merge axis-aligned rows into the box, or keep only the rows that can become active (e.g. the closest pairs of a SelfCollisionBarrier)";
  if (d.n_eq < 0 || d.n_eq > d.md || d.n_eq > d.nv) return "n_eq must be in 0..min(md, nv): equalities are the leading dense rows";
  if (d.T < 0 || d.Kd < 0 || d.K < d.Kd) return "need T >= 0 and 0 <= Kd <= K";
  if (d.T > 0 && (!d.task_rows || !d.task_kind || !d.gain || !d.lm_damping))
    return "task tables must not be NULL when T > 0";
  if (!(d.dt > 0.0) || !std::isfinite(d.dt)) return "dt must be a positive finite number";
  if (!std::isfinite(d.damping) || d.damping < 0.0) return "damping must be finite and >= 0";
  t = HostTables();
  t.row_gain.assign(d.K, 1.0);
  t.row_lm.assign(d.K, 0.0);
  int row = 0;
  bool seen_diag = false;
  for (int i = 0; i < d.T; ++i) {
    if (d.task_rows[i] != row) return "task_rows must start at 0 and be contiguous";
    const int k = d.task_rows[i + 1] - d.task_rows[i];
    if (k < 0) return "task_rows must be non-decreasing";
    if (d.task_kind[i] == PINKHIP_TASK_DENSE) {
      if (seen_diag) return "dense tasks must precede diagonal tasks";
    } else if (d.task_kind[i] == PINKHIP_TASK_DIAGONAL) {
      if (!seen_diag && row != d.Kd) return "dense task rows must add up to Kd";
      seen_diag = true;
      if (!d.task_col0) return "task_col0 must not be NULL with diagonal tasks";
      const int c0 = d.task_col0[i];
      if (c0 < 0 || c0 + k > d.nv) return "diagonal task exceeds the tangent space";
      t.dtask_col0.push_back(c0);
      t.dtask_row0.push_back(row);
      t.dtask_k.push_back(k);
    } else {
      return "unknown task kind";
    }
    for (int r = 0; r < k; ++r) {
      t.row_gain[row + r] = d.gain[i];
      t.row_lm[row + r] = d.lm_damping[i];
    }
    row += k;
  }
  if (row != d.K) return "task_rows[T] must equal K";
  if (!seen_diag && row != d.Kd) return "dense task rows must add up to Kd";
  if (d.n_barriers < 0) return "n_barriers must be >= 0";
  if (d.n_barriers > 0) {
    if (!d.barrier_rows || !d.barrier_safe_gain) return "barrier tables must not be NULL";
    for (int i = 0; i <= d.n_barriers; ++i) {
      const int r = d.barrier_rows[i];
      if (r < 0 || r > d.md || (i > 0 && r < d.barrier_rows[i - 1]))
        return "barrier_rows must be non-decreasing offsets into the md dense rows";
      t.barrier_rows.push_back(r);
    }
    for (int i = 0; i < d.n_barriers; ++i) t.barrier_safe_gain.push_back(d.barrier_safe_gain[i]);
  } else {
    t.barrier_rows.push_back(0);
  }
  return std::string();
}

// A task stack that cannot make H positive definite by itself: fewer task rows than tangent coordinates, no
// Levenberg-Marquardt term (task.py:160), no barrier regulariser (barrier.py:193-200) -- H = J^T W^2 J + damping I is
// then positive definite through `damping` alone (pink/solve_ik.py:55; examples/humanoid_jvrc.py:69-81 with the default
// 1e-12: cond(H) ~ 1e13).  The explicitly updated inverse of the sweep tableau has nothing to offer there (its
// conditioning estimate would route every instance to the Goldfarb-Idnani code after stacking and sweeping in vain):
// such a batch goes to that kernel by dispatch.
inline bool rank_deficient_by_construction(const pinkhip_desc &d) {
  if (!(d.K < d.nv) || !(d.damping <= 1e-9)) return false;
  for (int t = 0; t < d.T; ++t)
    if (d.lm_damping[t] != 0.0) return false;
  for (int b = 0; b < d.n_barriers; ++b)
    if (d.barrier_safe_gain[b] > 1e-6) return false;
  return true;
}

// Task layout of the whole-step kernel (pinkhip_rollout_step_device; shared with the test emulator): nf FrameTasks of six
// dense rows, then the constant-row dense tasks (n_const_rows rows in all), then diagonal tasks; the diagonal task
// number `posture_task` (counted among the diagonal ones; negative: none) is the PostureTask, which must cover the
// actuated coordinates.  Returns an error text ("" = fine) and the rows of the posture task.
inline std::string rollout_task_layout(const pinkhip_desc &d, int nf, int nv, int root_nv, int n_const_rows, int posture_task,
                                       bool have_diag_error, int &post_row0, int &post_k) {
  post_row0 = post_k = 0;
  if (d.Kd != 6 * nf + n_const_rows) return "descriptor does not describe this model's task stack (Kd = 6 nf + n_const_rows)";
  int t = 0, crow_seen = 0, n_diag = 0;
  for (; t < nf && t < d.T; ++t)
    if (d.task_kind[t] != PINKHIP_TASK_DENSE || d.task_rows[t + 1] - d.task_rows[t] != 6) return "frame tasks must be dense with six rows each";
  if (t != nf) return "expected one dense task per frame";
  for (; t < d.T && d.task_kind[t] == PINKHIP_TASK_DENSE; ++t) crow_seen += d.task_rows[t + 1] - d.task_rows[t];
  if (crow_seen != n_const_rows) return "dense tasks behind the frame tasks must add up to n_const_rows rows";
  for (; t < d.T; ++t, ++n_diag) {
    if (d.task_kind[t] != PINKHIP_TASK_DIAGONAL) return "dense tasks must precede diagonal ones";
    if (n_diag == posture_task) {
      post_row0 = d.task_rows[t];
      post_k = d.task_rows[t + 1] - d.task_rows[t];
      if (d.task_col0[t] != root_nv || post_k != nv - root_nv) return "the posture task must cover the actuated coordinates";
    }
  }
  if (posture_task >= n_diag && n_diag > 0) return "posture_task is not one of the diagonal tasks";
  if (n_diag > (post_k ? 1 : 0) && !have_diag_error) return "diagonal tasks other than the posture task need diag_error";
  return "";
}

}  // namespace pinkhip
