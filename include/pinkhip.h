/*
 * pinkhip.h -- C ABI of the MI355X (gfx950) batched differential-IK solver.
 *
 * The reference (stephane-caron/pink) has no FFI: its seam is Python.  The
 * entry points below are what a binding for the per-step hot path would call;
 * each one cites the reference code it replaces (paths relative to the
 * reference checkout).  Everything is `extern "C"`, plain pointers and sizes,
 * no exceptions cross the boundary, no PyTorch/pybind types.
 *
 * Conventions
 *   - all floating point data is IEEE fp64, C-contiguous;
 *   - every function returns 0 on success or a negative PINKHIP_E_* code, with
 *     a message available from pinkhip_last_error();
 *   - per-instance solver outcomes are reported in `status[B]`
 *     (PINKHIP_STATUS_*), mirroring the `found` flag that makes
 *     pink/solve_ik.py:271-273 raise NoSolutionFound;
 *   - a handle is bound to ONE device and is not thread-safe; distinct handles
 *     are independent (one process / one handle per GPU when sharding a batch);
 *   - "_host" entry points take host pointers the caller owns (copied in and
 *     out, synchronous); "_device" entry points take device pointers on the
 *     handle's device and only enqueue work on the handle's stream.
 */
#ifndef PINKHIP_H
#define PINKHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PINKHIP_VERSION 112 /* 110: iters[b] carries the solver path in its high bits (PINKHIP_ITERS_*); 112: pinkhip_desc::n_free_lead, pinkhip_pose_targets_device */

/* API-level error codes (negative). */
#define PINKHIP_OK 0
#define PINKHIP_E_INVALID (-1)     /* bad argument / unsupported shape          */
#define PINKHIP_E_HIP (-2)         /* a HIP runtime call failed                 */
#define PINKHIP_E_NOMEM (-3)       /* device or host allocation failed          */
#define PINKHIP_E_NODEVICE (-4)    /* no usable gfx950 device                   */
#define PINKHIP_E_UNSUPPORTED (-5) /* combination not supported by the selected kernel */
#define PINKHIP_E_COMM (-6)        /* RCCL failure                              */

/* Per-instance outcome (status[b]); anything non-zero maps to
 * pink.exceptions.NoSolutionFound (pink/exceptions.py:49-67). */
#define PINKHIP_STATUS_OPTIMAL 0
#define PINKHIP_STATUS_MAX_ITER 1
#define PINKHIP_STATUS_INFEASIBLE 2 /* quadprog: "constraints are inconsistent"        */
#define PINKHIP_STATUS_NOT_PD 3     /* quadprog: "matrix G is not positive definite"   */

/* iters[b]: active-set iterations in bits 0..23; bits 24..26 say which code solved the instance, so that a caller can
 * tell a batch that ran on the fast path from one where part of the instances paid for both solvers:
 *   TABLEAU   sweep-tableau kernel, result certified by its KKT check (the common case)
 *   HANDOVER  the tableau's result failed the certificate (explicitly updated inverse, cond(H) >~ 1e8): solved again
 *             by the Goldfarb-Idnani code (Cholesky + orthogonal updates) inside the same launch
 *   ROUTED    sent to that code BEFORE the tableau iteration by the conditioning estimate max_i H_ii (H^-1)_ii > 1e10
 *             (a rank-deficient task stack made positive definite by `damping` alone, pink/solve_ik.py:55)
 *   GI        the Goldfarb-Idnani kernel by dispatch (shapes the tableau kernel does not serve, PINKHIP_SOLVER=packed) */
#define PINKHIP_ITERS_COUNT(x) ((x) & 0xFFFFFF)
#define PINKHIP_ITERS_PATH(x) (((x) >> 24) & 7)
#define PINKHIP_PATH_TABLEAU 0
#define PINKHIP_PATH_HANDOVER 1
#define PINKHIP_PATH_ROUTED 2
#define PINKHIP_PATH_GI 3

#define PINKHIP_TASK_DENSE 0    /* rows of J stored                                      */
#define PINKHIP_TASK_DIAGONAL 1 /* J = eye(nv)[col0:col0+k] (pink/tasks/posture_task.py:128-129) */

#define PINKHIP_MAX_NV 64 /* tangent dimension supported by the wave-per-QP kernels */
#define PINKHIP_MAX_MD 64 /* dense inequality rows per instance (one lane of a 64-lane group each) */

typedef struct pinkhip_handle pinkhip_handle;

/*
 * Shape of one batch of IK problems.  Replaces the per-call Python state of
 * pink.build_ik (pink/solve_ik.py:152-203): the task list with each task's
 * cost/gain/lm_damping (pink/tasks/task.py:38-64), `damping`, `dt`.
 *
 *   rows of `e` / `cost`:  the Kd rows of the dense tasks first (task after
 *   task), then the rows of the diagonal tasks.  task_rows[t]..task_rows[t+1]
 *   are the rows of task t; dense tasks must precede diagonal ones.
 */
typedef struct pinkhip_desc {
  int64_t B;   /* instances in this call                                       */
  int32_t nv;  /* tangent dimension (configuration.model.nv), 1..PINKHIP_MAX_NV */
  int32_t T;   /* number of tasks                                               */
  int32_t Kd;  /* rows of dense-task Jacobians per instance                     */
  int32_t K;   /* all task rows per instance (= task_rows[T])                   */
  int32_t md;  /* dense inequality rows per instance, 0..PINKHIP_MAX_MD         */
  int32_t n_eq; /* the first n_eq dense rows are equalities Gd dq = hd (constraints=,
                   pink/solve_ik.py:125-149: A = J, b = -gain e); 0..min(md, nv) */
  const int32_t *task_rows;  /* [T+1] host                                      */
  const int32_t *task_kind;  /* [T]   host, PINKHIP_TASK_*                      */
  const int32_t *task_col0;  /* [T]   host, first tangent column of a diagonal task */
  const double *gain;        /* [T]   host, Task.gain (task.py:146)             */
  const double *lm_damping;  /* [T]   host, Task.lm_damping (task.py:160)       */
  int32_t n_barriers;        /* barrier row groups among the md dense rows      */
  const int32_t *barrier_rows;    /* [n_barriers+1] host, offsets into the md rows */
  const double *barrier_safe_gain; /* [n_barriers] host, safe_displacement_gain
                                      (barrier.py:193-200): adds r/||J_h||_F^2 I */
  double damping;            /* Tikhonov weight, solve_ik.py:55                 */
  double dt;                 /* timestep: barrier rows are -J_h/dt (barrier.py:246) */
  int32_t cost_is_batched;   /* cost is [B,K] instead of [K]                    */
  int32_t max_iter;          /* active-set iteration cap, <=0: 20*(nv+md)+50    */
  int32_t n_free_lead;       /* the first n_free_lead tangent coordinates carry no bound in ANY instance of the call
                                (lb = -inf, ub = +inf: the root of a free-flyer -- pink/limits/configuration_limit.py:50-71
                                never selects it).  A hint for dispatch, 0 when unknown: with it nv = 33 / 34 box-only batches
                                are solved two per wavefront (the leading coordinates are eliminated before the solve);
                                an instance that bounds them after all is still solved, by the Goldfarb-Idnani kernel */
} pinkhip_desc;

/* Per-instance data.  Host or device pointers depending on the entry point.
 *   J   [B,Kd,nv]  Task.compute_jacobian of the dense tasks (task.py:145)
 *   e   [B,K]      Task.compute_error (task.py:146)
 *   cost [K] | [B,K]  diagonal of W (task.py:148-156), already expanded per row
 *   lb, ub [B,nv]  box merged from every +-e_i limit row
 *                  (configuration_limit.py:117-120, velocity_limit.py:118-120);
 *                  -inf/+inf = coordinate without that bound
 *   Gd  [B,md,nv], hd [B,md]  equality rows A dq = b first (n_eq of them, solve_ik.py:140-149),
 *                  then the remaining rows of G dq <= h (solve_ik.py:107-122)
 *   c_extra [B,nv] or NULL   extra linear term (barrier.py:201)
 */
typedef struct pinkhip_problem {
  const double *J;
  const double *e;
  const double *cost;
  const double *lb;
  const double *ub;
  const double *Gd;
  const double *hd;
  const double *c_extra;
} pinkhip_problem;

/* Results.  dq [B,nv] is the QP minimiser (pink/solve_ik.py:271; the caller
 * divides by dt for the velocity, :274).  status [B]; iters [B] may be NULL
 * (PINKHIP_ITERS_COUNT / PINKHIP_ITERS_PATH above). */
typedef struct pinkhip_result {
  double *dq;
  int32_t *status;
  int32_t *iters;
} pinkhip_result;

typedef struct pinkhip_device_info {
  int32_t device_id;
  int32_t compute_units;
  int32_t wavefront_size;
  int32_t clock_mhz;
  int64_t total_mem_bytes;
  int64_t lds_per_cu_bytes;
  char name[128];
  char gcn_arch[64];
} pinkhip_device_info;

/* ---- lifetime ---------------------------------------------------------- */
int pinkhip_version(void);
int pinkhip_device_count(int *count);
/* Bind a handle to `device_id` (creates a stream and timing events). */
int pinkhip_create(pinkhip_handle **h, int device_id);
int pinkhip_destroy(pinkhip_handle *h);
const char *pinkhip_last_error(const pinkhip_handle *h); /* h may be NULL */
int pinkhip_get_device_info(const pinkhip_handle *h, pinkhip_device_info *info);

/* ---- the hot path ------------------------------------------------------ */
/* Stack + solve, replaces for B instances at once:
 *   Task.compute_qp_objective (pink/tasks/task.py:145-167)
 *   __compute_qp_objective / __compute_qp_inequalities (pink/solve_ik.py:54-67,107-122)
 *   qpsolvers.solve_problem(problem, solver="quadprog") (pink/solve_ik.py:270)
 * Host variant: copies the batch to the device, solves, copies dq/status back.
 */
int pinkhip_solve_host(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_problem *host_in,
                       const pinkhip_result *host_out);
/* Device time of the solve kernel(s) of the LAST pinkhip_solve_host call on this handle, in ms: from the start of the
 * first to the end of the last launch on the compute stream (a batch sent in several chunks: the waits for the uploads
 * in between included) -- next to the caller's own clock around the call this is the kernel-only / end-to-end breakdown
 * a batch sharded over several handles reports per device.  -1 when the call was not timed (batches
 * that go through the small-batch staging buffer). */
int pinkhip_last_kernel_ms(pinkhip_handle *h, float *ms);
/* Device variant: enqueue on the handle's stream; pointers stay owned by the
 * caller and must remain valid until pinkhip_sync(). */
int pinkhip_solve_device(pinkhip_handle *h, const pinkhip_desc *desc,
                         const pinkhip_problem *dev_in, const pinkhip_result *dev_out);

/* Stack only (pink.build_ik's P, q: pink/solve_ik.py:198): H_out [B,nv,nv], c_out [B,nv]. */
int pinkhip_stack_host(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_problem *host_in,
                       double *H_out, double *c_out);
int pinkhip_stack_device(pinkhip_handle *h, const pinkhip_desc *desc,
                         const pinkhip_problem *dev_in, double *H_out, double *c_out);

/* ---- upstream of the stack: batched FrameTask terms --------------------- */
/* For B instances of one FrameTask: e = log6(T_frame^-1 T_target) (body twist, [linear; angular];
 * pink/tasks/frame_task.py:176-193) and J = -Jlog6(T_target^-1 T_frame) J_body
 * (frame_task.py:217-227), replacing pin.log / pin.Jlog6 + the 6x6 by 6xnv product per instance.
 *   T_frame, T_target [B,12]  poses: rotation row-major (9) then translation (3)
 *   J_body [B,6,nv]           Configuration.get_frame_jacobian (LOCAL frame)
 *   e_out [B,6], J_out [B,6,nv]  directly usable as rows of pinkhip_problem.e / .J */
int pinkhip_frame_task_host(pinkhip_handle *h, int64_t B, int32_t nv, const double *T_frame,
                            const double *T_target, const double *J_body, double *e_out,
                            double *J_out);
int pinkhip_frame_task_device(pinkhip_handle *h, int64_t B, int32_t nv, const double *T_frame,
                              const double *T_target, const double *J_body, double *e_out,
                              double *J_out);

/* Same with explicit strides (in doubles) between consecutive instances, so that one frame of a
 * [B, nf, ...] array can be read and the result written straight into the rows of the packed
 * e [B, K] / J [B, Kd, nv] streams of pinkhip_problem. */
int pinkhip_frame_task_strided_device(pinkhip_handle *h, int64_t B, int32_t nv, const double *T_frame,
                                      int64_t sT_frame, const double *T_target, int64_t sT_target,
                                      const double *J_body, int64_t sJ_body, double *e_out, int64_t sE,
                                      double *J_out, int64_t sJ_out);

/* ---- closed loop on the device: kinematics around the IK step ---------- */
/* A kinematic tree of revolute / prismatic joints with an optional free-flyer root (joints in
 * topological order; free-flyer: q = [p, quat xyzw], tangent = body twist).  Replaces, for B
 * instances and without host round trips, what Pink asks Pinocchio for in its control loop:
 *   pinkhip_fk_device              pink/configuration.py:131-164,203-254  (FK, body frame Jacobians)
 *   pinkhip_limits_posture_device  limits/configuration_limit.py:111-120 + velocity_limit.py:118-120
 *                                  merged into lb/ub, and PostureTask's q (-) q* (posture_task.py:100-107)
 *   pinkhip_integrate_device       pink/configuration.py:273-293 (q <- q (+) dq)            */
#define PINKHIP_JOINT_REVOLUTE 0
#define PINKHIP_JOINT_PRISMATIC 1
#define PINKHIP_JOINT_FREE_FLYER 2
typedef struct pinkhip_model pinkhip_model;
typedef struct pinkhip_model_desc {
  int32_t nj, nq, nv, nf, root_nv;
  const int32_t *parent;       /* [nj] parent joint, -1 = world                      */
  const int32_t *jtype;        /* [nj] PINKHIP_JOINT_*                               */
  const int32_t *idx_q;        /* [nj]                                               */
  const int32_t *idx_v;        /* [nj]                                               */
  const double *placement;     /* [nj,12] joint frame in its parent joint frame      */
  const double *axis;          /* [nj,3]                                             */
  const int32_t *frame_joint;  /* [nf] joint a frame is attached to, -1 = world      */
  const double *frame_placement; /* [nf,12]                                          */
  const double *q_min;         /* [nq] model.lowerPositionLimit                      */
  const double *q_max;         /* [nq] model.upperPositionLimit                      */
  const double *v_max;         /* [nv] model.velocityLimit                           */
  /* Relative frame slots (pink/tasks/relative_frame_task.py:142-231), optional (NULL: none): slot f regulates the pose
   * of its frame in the frame `frame_root_placement[f]` attached to joint `frame_root_joint[f]` (-1 = world;
   * -2 = an ordinary slot, target in the world).  T_target[b, f] is then the target pose IN THAT ROOT FRAME; the
   * rows written / stacked for the slot are those of the RelativeFrameTask up to a common sign of J and e, which
   * H = J^T W J and c = -J^T W e do not see (pink/tasks/task.py:145-167). */
  const int32_t *frame_root_joint;     /* [nf] */
  const double *frame_root_placement;  /* [nf,12] */
} pinkhip_model_desc;
int pinkhip_model_create(pinkhip_handle *h, const pinkhip_model_desc *desc, pinkhip_model **model);
int pinkhip_model_destroy(pinkhip_handle *h, pinkhip_model *model);
/* q [B,nq] -> T_frames [B,nf,12], J_body [B,nf,6,nv] (device pointers) */
int pinkhip_fk_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, const double *q,
                      double *T_frames, double *J_body);
/* Forward kinematics and the FrameTask rows of every model frame in one pass (fusion of pinkhip_fk_device
 * with one pinkhip_frame_task_strided_device per frame; the body Jacobians stay in registers):
 *   e[b * sE + 6 f + i]              = log6(T_f(q_b)^-1 T_target[b, f])_i               (frame_task.py:181-193)
 *   J[b * sJ + (6 f + i) * nv + j]   = (-Jlog6(T_target[b, f]^-1 T_f(q_b)) fJ_f(q_b))_ij  (frame_task.py:222-227)
 * i.e. with sE = K and sJ = Kd * nv the rows land in the packed streams of pinkhip_problem.
 * T_frames [B,nf,12] is optional (NULL: not written).  All pointers are device pointers. */
int pinkhip_fk_frame_tasks_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, const double *q,
                                  const double *T_target, double *T_frames, double *e, int64_t sE, double *J,
                                  int64_t sJ);
/* Whole control step around the solve in ONE launch: q <- q (+) dq_prev for the instances whose previous solve
 * succeeded (as pinkhip_integrate_checked_device), forward kinematics, the FrameTask rows of every model frame
 * (as pinkhip_fk_frame_tasks_device), the merged box limits and the PostureTask error (as
 * pinkhip_limits_posture_device).  A closed loop is then  step -> solve -> step -> solve ...  with a final
 * pinkhip_integrate_checked_device.  All pointers are device pointers. */
typedef struct pinkhip_step {
  double *q;                  /* [B,nq] in / out */
  const double *dq_prev;      /* [B,nv] displacement of the previous solve, NULL on the first step */
  const int32_t *status;      /* [B] status of that solve (required with dq_prev) */
  int32_t *first_failure;     /* [B] sticky `status | (step << 8)`, may be NULL */
  int32_t step;               /* index recorded in first_failure */
  int32_t target_batched;     /* q_target is [B,nq] (1) or [nq] (0) */
  const double *T_target;     /* [B,nf,12] FrameTask targets */
  double *T_frames;           /* [B,nf,12] frame poses out, may be NULL */
  double *e;                  /* task errors: frame f at e[b*sE + 6f ..], posture rows at e[b*sE + e_off ..] */
  int64_t sE;
  double *J;                  /* frame-task Jacobians: rows 6f..6f+5 at J[b*sJ + ...], pitch nv */
  int64_t sJ;
  double dt, config_limit_gain;
  const double *q_target;     /* posture target, NULL: no posture rows */
  double *lb, *ub;            /* [B,nv] */
  int32_t e_off;
  const double *root_box;     /* [12] device, or NULL: lo[6], hi[6] of the free-flyer's tangent coordinates -- the
                                 axis-aligned rows of a FloatingBaseVelocityLimit
                                 (pink/limits/floating_base_velocity_limit.py:104-148), constant in q */
} pinkhip_step;
int pinkhip_step_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, const pinkhip_step *args);
/* The whole control step in ONE kernel: forward kinematics, FrameTask rows, limits, posture error, stacking,
 * QP solve and q <- q (+) dq -- the task Jacobians never reach memory (they are formed from the joints' world
 * twists while the objective is stacked).  `desc` describes the task stack of the model: one 6-row dense task per
 * model frame (in frame order: Kd = 6 nf), optionally followed by one diagonal task on the actuated coordinates
 * (the PostureTask: col0 = root_nv, nv - root_nv rows) -- and, since version 110, constant-row dense tasks and further
 * diagonal tasks (fields at the end of the struct); box limits, plus md rows of position barriers (below).  Returns
 * PINKHIP_E_UNSUPPORTED when no instantiation fits the model (nv > 56, or a group of lanes cannot hold the
 * joints / the kinematics scratch): use pinkhip_step_device + pinkhip_solve_device then. */
typedef struct pinkhip_rollout_step {
  double *q;                 /* [B,nq] in / out */
  const double *cost;        /* [K] (or [B,K] with desc.cost_is_batched) row weights, device memory */
  const double *T_target;    /* [B,nf,12] */
  double *T_frames;          /* [B,nf,12] out, may be NULL */
  const double *q_target;    /* [nq] / [B,nq] posture target; required iff desc has the diagonal task */
  double *dq;                /* [B,nv] out */
  int32_t *status;           /* [B] out */
  int32_t *iters;            /* [B] out, may be NULL */
  int32_t *first_failure;    /* [B] sticky `status | (step << 8)`, may be NULL */
  double config_limit_gain;
  int32_t target_batched, step, integrate;
  /* PositionBarrier rows formed on chip (pink/barriers/position_barrier.py:109-153): desc.md rows, grouped into
   * desc.n_barriers barriers by desc.barrier_rows (desc.barrier_safe_gain per barrier).  Row d keeps
   * sign_d (p_frame_d[axis_d] - bound_d) >= 0:  G_d = -sign_d (R J_lin)[axis_d] / dt,  h_d = gain_d sign_d (p - bound)
   * (pink/barriers/barrier.py:246-254).  Device pointers, [md] each; all NULL when desc.md = 0. */
  const int32_t *barrier_frame;  /* model frame index of the row */
  const int32_t *barrier_axis;   /* 0, 1, 2: world x, y, z */
  const double *barrier_sign;    /* +1: a p_min row, -1: a p_max row */
  const double *barrier_bound;
  const double *barrier_gain;
  int64_t sT_b, sT_f;        /* strides (doubles) of T_target: pose of instance b, frame f at T_target + b sT_b + f sT_f;
                                both 0: the contiguous [B,nf,12] (12 nf, 12).  (12, 12 B) addresses one [B,12] array per
                                frame, uploaded as it is (pink_amd.FrameTask.set_target_poses) */
  /* FloatingBaseVelocityLimit (pink/limits/floating_base_velocity_limit.py:104-148): +-J_root dq <= dt twist_max with
   * J_root the (constant) Jacobian of a frame attached to the root joint on the root's six tangent coordinates.  Its
   * axis-aligned rows arrive as root_box ([12] device: lo[6], hi[6]; NULL: none), the others as the FIRST
   * n_limit_rows of the desc.md dense rows (Pink stacks limits before barriers, pink/solve_ik.py:62-84): row d is
   * limit_rows[6 d ..] on the root coordinates, right-hand side limit_h[d].  The barrier_* tables then hold
   * desc.md - n_limit_rows entries and desc.barrier_rows offsets start at n_limit_rows. */
  const double *root_box;
  int32_t n_limit_rows;
  const double *limit_rows;  /* [n_limit_rows,6] device */
  const double *limit_h;     /* [n_limit_rows] device */
  double dq_scale;           /* what is written to dq is the displacement times this (0: 1).  1 / dt hands out the
                                velocity pink.solve_ik returns (pink/solve_ik.py:274) without another pass over the
                                array; only with integrate = 0 */
  /* Tasks beyond "one FrameTask per model frame + a PostureTask" that the kernel also forms on chip.
   *  - dense tasks with a CONSTANT Jacobian (LinearHolonomicTask / JointCouplingTask on vector-space joints,
   *    pink/tasks/linear_holonomic_task.py:103-148, joint_coupling_task.py: e = A (q (-) q_0) - b, J = A): the dense
   *    tasks of desc behind the nf frame tasks, n_const_rows rows in all (desc.Kd = 6 nf + n_const_rows);
   *  - diagonal tasks with BATCH-CONSTANT errors (DampingTask: 0, LowAccelerationTask: -dt v_prev, JointVelocityTask:
   *    -dt v*; pink/tasks/damping_task.py, low_acceleration_task.py:46-84, joint_velocity_task.py:59-110): every
   *    diagonal task of desc except number `posture_task` (counted among the diagonal tasks; -1: no PostureTask), whose
   *    error is q (-) q_target.  Their errors are read from diag_error at row - desc.Kd.
   * All zero (a zero-initialised struct): the stack of before -- frame tasks + at most one diagonal task = the posture. */
  int32_t n_const_rows;
  const double *const_rows;  /* [n_const_rows, nv] device: A */
  const double *const_q0;    /* [nq] device: q_0 */
  const double *const_b;     /* [n_const_rows] device */
  int32_t posture_task;
  const double *diag_error;  /* [desc.K - desc.Kd] device */
  /* AccelerationLimit on the joints behind the root (pink/limits/acceleration_limit.py:158-199), folded into the box
   * on chip: [3, nv] device -- a_max (0: no bound on that tangent coordinate), Delta_q_prev, has_configuration_limit
   * (0 / 1) -- or NULL:  dq <= min(a dt^2 + dq_prev, dt sqrt(2 a (q_max - q))),  -dq <= min(a dt^2 - dq_prev,
   * dt sqrt(2 a (q - q_min))), the braking-distance terms only where the joint has a configuration limit. */
  const double *acc_limit;
  /* Equality constraints made of frame tasks, `constraints=[FrameTask ...]` of pink.build_ik (pink/solve_ik.py:125-149:
   * A = J, b = -gain e): the LEADING 6 n_constraint_frames rows of the desc.md dense rows (desc.n_eq of them), in front of
   * the limit rows and the barrier rows; desc.barrier_rows offsets start at desc.n_eq + n_limit_rows.  Constraint c is the
   * FrameTask of model frame constraint_frame[c] ([n] device; its target in T_target like any frame's; a frame that
   * carries no task of the objective gets zero cost: pink/tasks/task.py:148-166 then adds nothing for it) with gain
   * constraint_gain[c] ([n] device).  At most 2.  The tables live in DEVICE memory and are not inspected by the call:
   * every constraint_frame[c] must be a slot of the model, 0 <= constraint_frame[c] < n_frames (the Python layer builds them
   * from the model's own frame list: pink_amd/rollout.py). */
  int32_t n_constraint_frames;
  const int32_t *constraint_frame;
  const double *constraint_gain;
  /* BodySphericalBarrier rows (pink/barriers/body_spherical_barrier.py:73-143): a barrier row d with barrier_axis[d] = 3
   * keeps |p_f - p_f2|^2 - d_min^2 >= 0 between the origins of model frames f = barrier_frame[d] and f2 =
   * barrier_frame2[d], with barrier_bound[d] = d_min^2 and the class's class-K function h / (1 + |h|):
   * G_d = -2 (p_f - p_f2)^T (R J_lin,f - R J_lin,f2) / dt,  h_d = gain_d alpha(h)  (pink/barriers/barrier.py:246-254).
   * [md - n_eq - n_limit_rows] device, mandatory whenever the barrier_* tables are passed (PINKHIP_E_INVALID without it):
   * the entry of a position-barrier row (axis 0 .. 2) is ignored, -1 by convention. */
  const int32_t *barrier_frame2;
} pinkhip_rollout_step;
int pinkhip_rollout_step_device(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_model *model,
                                const pinkhip_rollout_step *args);
/* q [B,nq], q_target [nq] or [B,nq] -> lb, ub [B,nv]; posture error written into e [B,K] at columns
 * e_off .. e_off + nv - root_nv (e may be NULL) */
int pinkhip_limits_posture_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, double dt,
                                  double config_limit_gain, const double *q, const double *q_target,
                                  int32_t target_batched, double *lb, double *ub, double *e, int32_t K,
                                  int32_t e_off);
/* Configuration.check_limits (pink/configuration.py:166-201) over a batch resident on the device: *first_bad receives
 * b * nq + i of the first entry with q_max > q_min + tol that lies outside [q_min - tol, q_max + tol] (coordinates of
 * the root joint skipped), or -1.  q is a device pointer, first_bad a host pointer; synchronises the stream. */
int pinkhip_check_limits_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, const double *q, double tol,
                                int64_t *first_bad);
/* q [B,nq] <- q (+) dq [B,nv], in place */
int pinkhip_integrate_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, double *q,
                             const double *dq);
/* FrameTask targets as translation + unit quaternion: pq [B,7] = (tx, ty, tz, qx, qy, qz, qw), the order of
 * pin.SE3ToXYZQUAT, written out as the T_target layout of the calls above, T [B,12] = rotation row-major, translation
 * (the quaternion is normalised on the way).  Replaces, for B robots, the pin.XYZQUATToSE3 a caller runs in front of
 * FrameTask.set_target (pink/tasks/frame_task.py:129-137): a moving-target control step then sends 56 B per target
 * across PCIe instead of 96 B.  Device pointers; enqueued on the handle's current compute stream, asynchronous. */
int pinkhip_pose_targets_device(pinkhip_handle *h, int64_t B, const double *pq, double *T);
/* Same, but an instance whose solve failed is NOT integrated: the reference raises NoSolutionFound and never
 * applies a failed solve (pink/solve_ik.py:271-275).  status [B] is the solver's output of this step;
 * first_failure [B] (optional, zero-initialised by the caller) keeps, per instance, `status | (step << 8)` of
 * the FIRST failing step, so a failure in the middle of a rollout stays visible after later steps. */
int pinkhip_integrate_checked_device(pinkhip_handle *h, const pinkhip_model *model, int64_t B, double *q,
                                     const double *dq, const int32_t *status, int32_t *first_failure,
                                     int32_t step);

/* ---- multi-GPU: gather of dq over RCCL / xGMI --------------------------- */
/* Instances are independent, so a batch is sharded over one handle per GPU (one process per
 * GPU) and solved without any exchange; the only collective of the workload is the gather of
 * the dq shards to one rank.  RCCL is loaded lazily (dlopen), the library works without it.
 *   pinkhip_comm_get_unique_id  rank 0 creates the 128-byte id; the caller ships it to every rank
 *                               (environment, file, any bootstrap channel)
 *   pinkhip_comm_init           every rank joins (collective call)
 *   pinkhip_comm_gather         count doubles from every rank's d_send to root's d_recv
 *                               [nranks * count] (device pointers; d_recv ignored off root);
 *                               enqueued on the handle's stream
 *   pinkhip_comm_gather_bytes   same for raw bytes (the int32 status / iteration counts)
 *   pinkhip_comm_allgather_bytes  every rank receives every shard: d_recv [nranks * nbytes] on all ranks */
#define PINKHIP_COMM_ID_BYTES 128
int pinkhip_comm_get_unique_id(char *id /* [PINKHIP_COMM_ID_BYTES] */);
int pinkhip_comm_init(pinkhip_handle *h, const char *id, int rank, int nranks);
int pinkhip_comm_gather(pinkhip_handle *h, const double *d_send, double *d_recv, int64_t count, int root);
int pinkhip_comm_gather_bytes(pinkhip_handle *h, const void *d_send, void *d_recv, int64_t nbytes, int root);
int pinkhip_comm_allgather_bytes(pinkhip_handle *h, const void *d_send, void *d_recv, int64_t nbytes);
int pinkhip_comm_destroy(pinkhip_handle *h);

/* ---- device memory, stream, timing ------------------------------------- */
/* page-locked host memory: buffers handed to the *_host entry points from here are copied by DMA at the PCIe
 * rate and overlap with the kernels (pageable buffers work too, staged by the HIP runtime) */
int pinkhip_host_alloc(pinkhip_handle *h, void **hptr, int64_t bytes);
int pinkhip_host_free(pinkhip_handle *h, void *hptr);
int pinkhip_malloc(pinkhip_handle *h, void **dptr, int64_t bytes);
int pinkhip_free(pinkhip_handle *h, void *dptr);
int pinkhip_memcpy_h2d(pinkhip_handle *h, void *dst, const void *src, int64_t bytes);
/* The same copy on the handle's COPY stream: it waits only for itself, not for the kernels enqueued on the compute
 * stream, so that the upload of the next piece of a batch overlaps the kernels of the previous one.  Returns when the
 * source has been consumed and the data is on the device: a kernel launched afterwards sees it.  The caller keeps the
 * destination disjoint from what enqueued kernels still read or write. */
int pinkhip_memcpy_h2d_overlapped(pinkhip_handle *h, void *dst, const void *src, int64_t bytes);
/* Fully asynchronous pieces of a pipelined call (a batch cut into ranges: upload of range c + 1, kernels of range c and
 * the results of range c - 1 going home all in flight at once -- PCIe is full duplex):
 *   pinkhip_memcpy_h2d_async    enqueue on the COPY stream and return (page-locked source -- pinkhip_host_alloc --: true
 *                               DMA, the source must stay untouched until pinkhip_sync; pageable source: the runtime
 *                               stages it before returning, correct but not overlapped)
 *   pinkhip_stream_wait_copies  what is enqueued on the compute stream from now on starts after the copies enqueued so far
 *   pinkhip_memcpy_d2h_async    enqueue on the RESULT stream, ordered after the kernels enqueued so far, and return
 *   pinkhip_sync                waits for all three streams */
int pinkhip_memcpy_h2d_async(pinkhip_handle *h, void *dst, const void *src, int64_t bytes);
int pinkhip_stream_wait_copies(pinkhip_handle *h);
/* Which of the handle's two compute streams the entry points enqueue on from now on (0: the stream of pinkhip_create,
 * the default; 1: a second one, created on first use).  Ranges of one batch launched alternately on the two overlap the
 * drain of one range's kernel with the start of the next (a range of 8 192 wavefronts is 2.7 rounds of the chip's wave
 * slots: launched back to back on ONE stream four ranges cost twice a single launch over the batch).  pinkhip_sync waits
 * for both; a caller that selected stream 1 selects 0 again before it uses the handle for anything else. */
int pinkhip_select_compute_stream(pinkhip_handle *h, int32_t index);
int pinkhip_memcpy_d2h_async(pinkhip_handle *h, void *dst, const void *src, int64_t bytes);
int pinkhip_memcpy_d2h(pinkhip_handle *h, void *dst, const void *src, int64_t bytes);
int pinkhip_memcpy_d2d(pinkhip_handle *h, void *dst, const void *src, int64_t bytes); /* stream-ordered, asynchronous */
int pinkhip_sync(pinkhip_handle *h);
/* HIP events recorded on the handle's stream around whatever is enqueued in
 * between; elapsed_ms is valid after pinkhip_timer_stop returns. */
int pinkhip_timer_start(pinkhip_handle *h);
int pinkhip_timer_stop(pinkhip_handle *h, float *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* PINKHIP_H */
