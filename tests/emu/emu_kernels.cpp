// Host harness that runs the HIP kernel source on the CPU wave emulator.
// TEST INFRASTRUCTURE ONLY (see wave_emu.h).  Exposes the same struct-based
// signature as the host entry points of include/pinkhip.h.
#include "emu_lanes.h"

#include <map>
#include <string>
#include <tuple>

namespace pinkemu {
namespace {
std::map<std::tuple<int, int, int, int>, LaneEntry> &registry() {
  static std::map<std::tuple<int, int, int, int>, LaneEntry> r;
  return r;
}
}  // namespace
void emu_register(int kind, int nv, int md, int w, LaneEntry fn) { registry()[std::make_tuple(kind, nv, md, w)] = fn; }
LaneEntry emu_lookup(int kind, int nv, int md, int w) {
  auto it = registry().find(std::make_tuple(kind, nv, md, w));
  return it == registry().end() ? nullptr : it->second;
}
}  // namespace pinkemu

using namespace pinkemu;

namespace {

using pinkhip::KernelArgs;

std::string g_err;

int run(const pinkhip_desc *d, const pinkhip_problem *in, const pinkhip_result *out, double *H_out,
        double *c_out, bool solve) {
  pinkhip::HostTables t;
  g_err = pinkhip::build_tables(*d, t);
  if (!g_err.empty()) return PINKHIP_E_INVALID;
  KernelArgs a{};
  a.B = d->B;
  a.nv = d->nv;
  a.Kd = d->Kd;
  a.K = d->K;
  a.md = d->md;
  a.n_eq = d->n_eq;
  a.n_dtasks = static_cast<int>(t.dtask_k.size());
  a.n_barriers = d->n_barriers;
  a.cost_batched = d->cost_is_batched;
  a.max_iter = d->max_iter;
  a.damping = d->damping;
  a.dt = d->dt;
  a.out_scale = 1.0;
  a.J = in->J;
  a.e = in->e;
  a.cost = in->cost;
  a.lb = in->lb;
  a.ub = in->ub;
  a.Gd = in->Gd;
  a.hd = in->hd;
  a.c_extra = in->c_extra;
  a.row_gain = t.row_gain.data();
  a.row_lm = t.row_lm.data();
  a.dtask_col0 = t.dtask_col0.data();
  a.dtask_row0 = t.dtask_row0.data();
  a.dtask_k = t.dtask_k.data();
  a.barrier_rows = t.barrier_rows.data();
  a.barrier_safe_gain = t.barrier_safe_gain.data();
  if (out) {
    a.dq = out->dq;
    a.status = out->status;
    a.iters = out->iters;
  }
  a.H_out = H_out;
  a.c_out = c_out;
  pinkhip::LaneFn fn = nullptr;
  long long blocks = d->B;
  if (!solve && a.nv <= 8 && a.n_barriers == 0) {  // same rule as pinkhip.hip
    // (the library switches to four tiles per wave at B >= 65536; the emulator exercises both on small batches)
    const bool four = (d->B % 2) == 1;
    fn = four ? lane_main_stack_small<4> : lane_main_stack_small<1>;
    blocks = four ? (d->B + 7) / 8 : (d->B + 1) / 2;
  } else if (!solve) {
    switch ((a.nv + 15) / 16) {
      case 1: fn = lane_main_stack_mfma<1>; break;
      case 2: fn = lane_main_stack_mfma<2>; break;
      case 3: fn = lane_main_stack_mfma<3>; break;
      case 4: fn = lane_main_stack_mfma<4>; break;
    }
  } else {  // the dispatch rule of the library (dispatch.h, pinkhip.hip launch())
    a.n_free_lead = (d->n_free_lead > 0 && d->n_free_lead <= d->nv) ? d->n_free_lead : 0;
    const pinkhip::SweepChoice sc = pinkhip::select_sweep(a.nv, a.md, a.n_free_lead);
    const char *force = std::getenv("PINKHIP_SOLVER");  // "packed" / "sweep": one kernel for every problem it serves
    a.rank_deficient = pinkhip::rank_deficient_by_construction(*d) ? 1 : 0;
    const bool sweep = force ? (std::string(force) != "packed" && sc.NV != 0) : (pinkhip::prefer_sweep(a.nv, a.md, d->B, a.n_free_lead) && !a.rank_deficient);
    const pinkhip::SweepChoice xc = pinkhip::select_sweepx(a.nv, a.md);
    const bool sweepx = force ? (std::string(force) == "sweepx" && xc.NV != 0) : (pinkhip::prefer_sweepx(a.nv, a.md) && !a.rank_deficient);
    if (sweepx) {
      switch (xc.NV * 100 + xc.MD) {
#define PINKHIP_CASE(NV, MD, W)                \
  case NV * 100 + MD:                          \
    fn = emu_lookup(KIND_SWEEPX, NV, MD, W);   \
    blocks = (d->B + 64 / W - 1) / (64 / W);   \
    break;
        PINKHIP_SWEEPX_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
      }
    }
    if (!fn && sweep) {
      switch (sc.NV * 10000 + sc.MD * 100 + sc.W) {
#define PINKHIP_CASE(NV, MD, W)                \
  case NV * 10000 + MD * 100 + W:              \
    fn = emu_lookup(KIND_SWEEP, NV, MD, W);    \
    blocks = (d->B + 64 / W - 1) / (64 / W);   \
    break;
        PINKHIP_SWEEP_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
      }
    }
    const pinkhip::PackedChoice pc = pinkhip::select_packed(a.nv, a.md);
    if (!fn) switch (pc.NV) {
#define PINKHIP_CASE(NV, W)          \
  case NV:                           \
    fn = emu_lookup(KIND_PACKED, NV, 0, W); \
    blocks = (d->B + 64 / W - 1) / (64 / W); \
    break;
      PINKHIP_PACKED_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    }
  }
  if (!fn) {
    g_err = "unsupported nv / md";
    return PINKHIP_E_INVALID;
  }
  for (long long b = 0; b < blocks; ++b) pinkhip::emu_run_block(b, fn, &a);
  return PINKHIP_OK;
}

}  // namespace

struct EmuModel {
  pinkhip::ModelImage image;
  pinkhip::ModelDev dev;
};

extern "C" {
// kinematics entry points mirroring pinkhip_model_create / _fk_device / ... on host memory
int pinkhip_emu_model_create(const pinkhip_model_desc *d, void **out) {
  EmuModel *m = new EmuModel();
  g_err = pinkhip::build_model_image(*d, m->image);
  if (!g_err.empty()) {
    delete m;
    return PINKHIP_E_INVALID;
  }
  m->dev = pinkhip::model_view<pinkhip::ModelDev>(m->image, m->image.bytes.data());
  *out = m;
  return PINKHIP_OK;
}
int pinkhip_emu_model_destroy(void *m) {
  delete static_cast<EmuModel *>(m);
  return PINKHIP_OK;
}
int pinkhip_emu_fk(void *mp, long long B, const double *q, double *T_frames, double *J_body) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::FkArgs a{m->dev, B, q, T_frames, J_body};
  const int width = m->dev.nv > m->dev.nj ? m->dev.nv : m->dev.nj;
  if (width <= 8) {
    for (long long b = 0; b < (B + 7) / 8; ++b) pinkhip::emu_run_block(b, lane_main_fk<8>, &a);
  } else if (width <= 32) {
    for (long long b = 0; b < (B + 1) / 2; ++b) pinkhip::emu_run_block(b, lane_main_fk<32>, &a);
  } else {
    for (long long b = 0; b < B; ++b) pinkhip::emu_run_block(b, lane_main_fk<64>, &a);
  }
  return PINKHIP_OK;
}
int pinkhip_emu_fk_frame_tasks(void *mp, long long B, const double *q, const double *T_target, double *T_frames,
                               double *e, long long sE, double *J, long long sJ) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::FkArgs a{m->dev, B, q, T_frames, nullptr};
  a.T_target = T_target;
  a.e_out = e;
  a.J_out = J;
  a.sE = sE;
  a.sJo = sJ;
  const int width = m->dev.nv > m->dev.nj ? m->dev.nv : m->dev.nj;
  if (width <= 8) {
    for (long long b = 0; b < (B + 7) / 8; ++b) pinkhip::emu_run_block(b, lane_main_fk_fused<8>, &a);
  } else if (width <= 32) {
    for (long long b = 0; b < (B + 1) / 2; ++b) pinkhip::emu_run_block(b, lane_main_fk_fused<32>, &a);
  } else {
    for (long long b = 0; b < B; ++b) pinkhip::emu_run_block(b, lane_main_fk_fused<64>, &a);
  }
  return PINKHIP_OK;
}
int pinkhip_emu_step(void *mp, long long B, const pinkhip_step *st) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::FkArgs a{m->dev, B, st->q, st->T_frames, nullptr};
  a.T_target = st->T_target;
  a.e_out = st->e;
  a.J_out = st->J;
  a.sE = st->sE;
  a.sJo = st->sJ;
  a.q_rw = st->q;
  a.dq_prev = st->dq_prev;
  a.status = st->status;
  a.first_failure = st->first_failure;
  a.step = st->step;
  a.dt = st->dt;
  a.config_limit_gain = st->config_limit_gain;
  a.root_box = st->root_box;
  a.q_target = st->q_target;
  a.target_batched = st->target_batched;
  a.lb = st->lb;
  a.ub = st->ub;
  a.e_off = st->e_off;
  const int width = m->dev.nv > m->dev.nj ? m->dev.nv : m->dev.nj;
  if (width <= 8) {
    for (long long b = 0; b < (B + 7) / 8; ++b) pinkhip::emu_run_block(b, lane_main_step<8>, &a);
  } else if (width <= 32) {
    for (long long b = 0; b < (B + 1) / 2; ++b) pinkhip::emu_run_block(b, lane_main_step<32>, &a);
  } else {
    for (long long b = 0; b < B; ++b) pinkhip::emu_run_block(b, lane_main_step<64>, &a);
  }
  return PINKHIP_OK;
}
int pinkhip_emu_rollout_step(const pinkhip_desc *d, void *mp, const pinkhip_rollout_step *st) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::HostTables t;
  g_err = pinkhip::build_tables(*d, t);
  if (!g_err.empty()) return PINKHIP_E_INVALID;
  pinkhip::RolloutArgs ra{};
  KernelArgs &a = ra.k;
  a.B = d->B;
  a.nv = d->nv;
  a.Kd = d->Kd;
  a.K = d->K;
  a.md = d->md;
  a.n_eq = d->n_eq;
  if (d->n_eq != 6 * st->n_constraint_frames || st->n_constraint_frames < 0 || st->n_constraint_frames > pinkhip::kRolloutMaxEqFrames) {
    g_err = "n_eq = 6 n_constraint_frames, at most 2 constraint frames";
    return PINKHIP_E_INVALID;
  }
  a.n_barriers = d->n_barriers;
  a.n_dtasks = static_cast<int>(t.dtask_k.size());
  a.cost_batched = d->cost_is_batched;
  a.max_iter = d->max_iter;
  a.damping = d->damping;
  a.dt = d->dt;
  a.rank_deficient = pinkhip::rank_deficient_by_construction(*d) ? 1 : 0;  // (as prepare() of pinkhip.hip)
  a.out_scale = (st->dq_scale != 0.0) ? st->dq_scale : 1.0;
  a.cost = st->cost;
  a.row_gain = t.row_gain.data();
  a.row_lm = t.row_lm.data();
  a.dtask_col0 = t.dtask_col0.data();
  a.dtask_row0 = t.dtask_row0.data();
  a.dtask_k = t.dtask_k.data();
  a.barrier_rows = t.barrier_rows.data();
  a.barrier_safe_gain = t.barrier_safe_gain.data();
  a.dq = st->dq;
  a.status = st->status;
  a.iters = st->iters;
  pinkhip::FkArgs &f = ra.fk;
  f.m = m->dev;
  f.B = d->B;
  f.q = st->q;
  f.q_rw = st->q;
  f.T_frames = st->T_frames;
  f.T_target = st->T_target;
  f.sTb = st->sT_b;
  f.sTf = (st->sT_b || st->sT_f) ? st->sT_f : 12;
  f.dt = d->dt;
  f.config_limit_gain = st->config_limit_gain;
  f.root_box = st->root_box;
  f.acc_limit = st->acc_limit;
  int post_row0 = 0, post_k = 0;
  g_err = pinkhip::rollout_task_layout(*d, m->dev.nf, m->dev.nv, m->dev.root_nv, st->n_const_rows, st->posture_task, st->diag_error != nullptr,
                                       post_row0, post_k);
  if (!g_err.empty()) return PINKHIP_E_INVALID;
  f.q_target = post_k ? st->q_target : nullptr;
  f.target_batched = st->target_batched;
  ra.integrate = st->integrate;
  ra.first_failure = st->first_failure;
  ra.step = st->step;
  ra.n_crow = st->n_const_rows;
  ra.crow_A = st->const_rows;
  ra.crow_q0 = st->const_q0;
  ra.crow_b = st->const_b;
  ra.post_row0 = post_row0;
  ra.post_k = post_k;
  ra.diag_e = st->diag_error;
  const int fkd = pinkhip::rollout_fk_doubles(m->dev.nj, m->dev.nf, st->n_const_rows);
  pinkhip::LaneFn fn = nullptr;
  long long blocks = 0;
  pinkhip::PackedChoice pc{0, 0};
  if (d->md > 0) {
    const pinkhip::SweepChoice dc = pinkhip::select_rollout_dense(m->dev.nv, m->dev.nj, fkd, d->md, m->dev.nf, st->n_constraint_frames);
    a.lds_pitch = pinkhip::rollout_lds_doubles(dc.NV, dc.W, fkd, dc.MD, m->dev.nf, st->n_constraint_frames);
    ra.bar_frame = st->barrier_frame;
    ra.bar_axis = st->barrier_axis;
    ra.bar_sign = st->barrier_sign;
    ra.bar_bound = st->barrier_bound;
    ra.bar_gain = st->barrier_gain;
    ra.n_lim = st->n_limit_rows;
    ra.lim_rows = st->limit_rows;
    ra.lim_h = st->limit_h;
    ra.n_eqf = st->n_constraint_frames;
    ra.eq_frame = st->constraint_frame;
    ra.eq_gain = st->constraint_gain;
    ra.bar_frame2 = st->barrier_frame2;
    switch (dc.NV * 100 + dc.MD) {
#define PINKHIP_CASE(NV, MD, W)                \
  case NV * 100 + MD:                          \
    fn = emu_lookup(KIND_ROLLOUT_DENSE, NV, MD, W); \
    blocks = (d->B + 64 / W - 1) / (64 / W);   \
    break;
      PINKHIP_ROLLOUT_DENSE_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    }
  } else {
    pc = pinkhip::select_rollout(m->dev.nv, m->dev.nj, fkd, st->n_const_rows > 0 || st->diag_error != nullptr || st->acc_limit != nullptr || m->image.has_relative);
    a.lds_pitch = pinkhip::rollout_lds_doubles(pc.NV, pc.W, fkd);
  }
  if (d->md == 0) switch (pc.NV) {
#define PINKHIP_CASE(NV, W)                  \
  case NV:                                   \
    fn = emu_lookup(KIND_ROLLOUT, NV, 0, W);  \
    blocks = (d->B + 64 / W - 1) / (64 / W); \
    break;
    PINKHIP_ROLLOUT_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
  }
  if (!fn || m->dev.nf > 32) {
    g_err = "no whole-step instantiation fits this model";
    return PINKHIP_E_UNSUPPORTED;
  }
  for (long long b = 0; b < blocks; ++b) pinkhip::emu_run_block(b, fn, &ra);
  return PINKHIP_OK;
}
int pinkhip_emu_limits_posture(void *mp, long long B, double dt, double gain, const double *q,
                               const double *q_target, int target_batched, double *lb, double *ub, double *e,
                               int K, int e_off) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::LimitsPostureArgs a{m->dev, B, dt, gain, q, q_target, target_batched, lb, ub, e, K, e_off};
  for (long long t = 0; t < B * m->dev.nv; ++t) pinkhip::ik_limits_posture_thread(a, t);
  return PINKHIP_OK;
}
int pinkhip_emu_check_limits(void *mp, long long B, const double *q, double tol, long long *first_bad) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  // the device kernel takes an atomic minimum over its threads; in a plain loop the first hit is the minimum
  *first_bad = -1;
  const int start = m->image.root_nv == 6 ? 7 : m->image.root_nv;
  for (long long t = 0; t < B * m->dev.nq && *first_bad < 0; ++t) {
    const int i = static_cast<int>(t % m->dev.nq);
    if (i < start) continue;
    const double lo = m->dev.q_min[i], up = m->dev.q_max[i];
    if (up > lo + tol && (q[t] < lo - tol || q[t] > up + tol)) *first_bad = t;
  }
  return PINKHIP_OK;
}
int pinkhip_emu_integrate(void *mp, long long B, double *q, const double *dq) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::IntegrateArgs a{m->dev, B, q, dq};
  for (long long t = 0; t < B * m->dev.nj; ++t) pinkhip::ik_integrate_thread(a, t);
  return PINKHIP_OK;
}
int pinkhip_emu_pose_targets(long long B, const double *pq, double *T) {
  pinkhip::PoseTargetsArgs a{B, pq, T};
  for (long long t = 0; t < B; ++t) pinkhip::ik_pose_targets_thread(a, t);
  return PINKHIP_OK;
}
int pinkhip_emu_integrate_checked(void *mp, long long B, double *q, const double *dq, const int *status,
                                  int *first_failure, int step) {
  EmuModel *m = static_cast<EmuModel *>(mp);
  pinkhip::IntegrateArgs a{m->dev, B, q, dq, status, first_failure, step};
  for (long long t = 0; t < B * m->dev.nj; ++t) pinkhip::ik_integrate_thread(a, t);
  return PINKHIP_OK;
}
int pinkhip_emu_frame_task_strided(long long B, int nv, const double *T_frame, long long sTf, const double *T_target,
                                   long long sTt, const double *J_body, long long sJb, double *e_out, long long sE,
                                   double *J_out, long long sJo) {
  pinkhip::FrameTaskArgs a{B, nv, T_frame, T_target, J_body, e_out, J_out, sTf, sTt, sJb, sE, sJo};
  pinkhip::LaneFn fn;
  int G;
  if (nv <= 8) { fn = lane_main_frame<8>; G = 8; }
  else if (nv <= 16) { fn = lane_main_frame<16>; G = 4; }
  else if (nv <= 32) { fn = lane_main_frame<32>; G = 2; }
  else { fn = lane_main_frame<64>; G = 1; }
  for (long long b = 0; b < (B + G - 1) / G; ++b) pinkhip::emu_run_block(b, fn, &a);
  return PINKHIP_OK;
}
int pinkhip_emu_frame_task_host(long long B, int nv, const double *T_frame, const double *T_target,
                                const double *J_body, double *e_out, double *J_out) {
  pinkhip::FrameTaskArgs a{B, nv, T_frame, T_target, J_body, e_out, J_out};
  pinkhip::LaneFn fn;
  int G;
  if (nv <= 8) { fn = lane_main_frame<8>; G = 8; }
  else if (nv <= 16) { fn = lane_main_frame<16>; G = 4; }
  else if (nv <= 32) { fn = lane_main_frame<32>; G = 2; }
  else { fn = lane_main_frame<64>; G = 1; }
  for (long long b = 0; b < (B + G - 1) / G; ++b) pinkhip::emu_run_block(b, fn, &a);
  return PINKHIP_OK;
}
int pinkhip_emu_solve_host(const pinkhip_desc *d, const pinkhip_problem *in,
                           const pinkhip_result *out) {
  return run(d, in, out, nullptr, nullptr, true);
}
int pinkhip_emu_stack_host(const pinkhip_desc *d, const pinkhip_problem *in, double *H_out,
                           double *c_out) {
  return run(d, in, nullptr, H_out, c_out, false);
}
const char *pinkhip_emu_last_error(void) { return g_err.c_str(); }
}
