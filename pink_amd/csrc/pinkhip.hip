// libpinkhip.so -- host side of the C ABI in include/pinkhip.h (gfx950 only).
//
// One handle = one device, one stream, two timing events, a small device-side
// table area for the per-descriptor broadcast constants, and a grow-only device
// arena used by the *_host entry points.  No exceptions cross the ABI.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// clang-format off
#include "wave.h"
#include "ik_common.h"
#include "launchers.h"
#include "ik_stack_mfma.h"
#include "ik_frame_task.h"
#include "ik_kinematics.h"
#include "ik_rollout.h"
#include "host_tables.h"
#include "model_tables.h"
// clang-format on

namespace {

thread_local std::string g_last_error;

constexpr size_t kTableBytes = 64 * 1024;

}  // namespace

struct pinkhip_model {
  pinkhip::ModelImage image;
  char *d_base = nullptr;
  pinkhip::ModelDev dev;
};

struct pinkhip_handle {
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D of the chunked *_host path (overlaps the kernels on `stream`)
  hipStream_t d2h_stream = nullptr;   // results going home while later kernels run (pinkhip_memcpy_d2h_async)
  hipStream_t main_stream = nullptr, alt_stream = nullptr;  // `stream` is one of these two (pinkhip_select_compute_stream)
  hipEvent_t ev_copy = nullptr, ev_kernels = nullptr;  // copy stream -> compute stream, compute stream -> d2h stream
  std::vector<hipEvent_t> chunk_events;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_join = nullptr;  // timer bracket; the other compute stream joining it
  hipEvent_t evk0 = nullptr, evk1 = nullptr;  // around the solve kernel(s) of the last pinkhip_solve_host call
  bool evk_valid = false;
  std::string err;
  char *d_tables = nullptr;          // device copy of the broadcast tables
  std::vector<char> h_tables;        // what d_tables currently holds
  char *arena = nullptr;             // grow-only scratch of the *_host entry points
  size_t arena_bytes = 0;
  char *stage = nullptr;             // page-locked, device-visible staging of the small-batch *_host path
  hipDeviceProp_t prop;
  void *comm = nullptr;              // ncclComm_t once pinkhip_comm_init succeeded
  int comm_rank = 0, comm_size = 0;
};

namespace {

int fail(pinkhip_handle *h, int code, const std::string &msg) {
  g_last_error = msg;
  if (h) h->err = msg;
  return code;
}

#define PH_HIP(h, call)                                                                    \
  do {                                                                                     \
    hipError_t e__ = (call);                                                               \
    if (e__ != hipSuccess)                                                                 \
      return fail(h, e__ == hipErrorOutOfMemory ? PINKHIP_E_NOMEM : PINKHIP_E_HIP,         \
                  std::string(#call) + ": " + hipGetErrorString(e__));                    \
  } while (0)

using pinkhip::KernelArgs;

// ncclUniqueId is passed by value to ncclCommInitRank: same layout, no RCCL header needed
struct pinkhip_unique_id_t {
  char internal[PINKHIP_COMM_ID_BYTES];
};

template <int NT>
int launch_stack_mfma(pinkhip_handle *h, const KernelArgs &a) {
  const dim3 grid(static_cast<unsigned>(a.B)), block(pinkhip::kWave);
  static const bool direct = std::getenv("PINKHIP_STACK_DIRECT") != nullptr;  // development: time the unstaged variant
  if (!direct && pinkhip::stack_staged_ok(a.nv, a.Kd, a.J)) {
    const size_t lds = 8 * static_cast<size_t>(pinkhip::stack_staged_lds_doubles(a.nv, a.Kd));
    if constexpr (NT >= 3) hipLaunchKernelGGL(pinkhip::ik_stack_staged_kernel<NT>, grid, block, lds, h->stream, a);
  } else {
    hipLaunchKernelGGL(pinkhip::ik_stack_mfma_kernel<NT>, grid, block, 2048, h->stream, a);
  }
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int launch(pinkhip_handle *h, const KernelArgs &a, bool solve) {
  if (a.B == 0) return PINKHIP_OK;
  if (a.B > 0x7fffffffLL) return fail(h, PINKHIP_E_INVALID, "B exceeds the grid limit 2^31-1");
  if (!solve) {  // stack only: fp64 MFMA tiles, NT = ceil(nv / 16)
    if (a.nv <= 8 && a.n_barriers == 0) {
      // two instances per MFMA tile; four tiles per wavefront once that still leaves >= 32 waves per CU
      // (measured, UR5: 24 us instead of 51 us at B = 65 536, but 6.4 instead of 5.1 us at B = 4 096)
      const dim3 block(pinkhip::kWave);
      if (a.B >= 65536) {
        hipLaunchKernelGGL(pinkhip::ik_stack_small_kernel<4>, dim3(static_cast<unsigned>((a.B + 7) / 8)), block, 0, h->stream, a);
      } else {
        hipLaunchKernelGGL(pinkhip::ik_stack_small_kernel<1>, dim3(static_cast<unsigned>((a.B + 1) / 2)), block, 0, h->stream, a);
      }
      PH_HIP(h, hipGetLastError());
      return PINKHIP_OK;
    }
    switch ((a.nv + 15) / 16) {
      case 1: return launch_stack_mfma<1>(h, a);
      case 2: return launch_stack_mfma<2>(h, a);
      case 3: return launch_stack_mfma<3>(h, a);
      case 4: return launch_stack_mfma<4>(h, a);
    }
    return fail(h, PINKHIP_E_INVALID, "unsupported nv");
  }
  // stack + solve: the instantiation chosen by dispatch.h, each one its own translation unit.  The sweep-tableau
  // kernel (ik_sweep.h, tu_sweep.hip) serves every problem it is instantiated for; the Goldfarb-Idnani kernel
  // (ik_kernels_packed.h, tu_packed.hip) the rest: 8-lane groups (nv <= 8), more dense rows than lanes are left.
  const char *solver_env = std::getenv("PINKHIP_SOLVER");  // development / tests: "packed" / "sweep" force one kernel
  const pinkhip::SweepChoice sc = pinkhip::select_sweep(a.nv, a.md, a.n_free_lead);
  const bool sweep = solver_env ? (std::strcmp(solver_env, "packed") != 0 && sc.NV != 0)
                                : (pinkhip::prefer_sweep(a.nv, a.md, a.B, a.n_free_lead) && !a.rank_deficient);
  // ... with virtual dense rows where that packs more QPs into a wavefront (ik_sweepx.h, dispatch.h prefer_sweepx)
  const pinkhip::SweepChoice xc = pinkhip::select_sweepx(a.nv, a.md);
  const bool sweepx = solver_env ? (std::strcmp(solver_env, "sweepx") == 0 && xc.NV != 0)
                                 : (pinkhip::prefer_sweepx(a.nv, a.md) && !a.rank_deficient);
  if (sweepx) {
    hipError_t ex = hipErrorInvalidValue;
    switch (xc.NV * 100 + xc.MD) {
#define PINKHIP_CASE(NV, MD, W)                                            \
  case NV * 100 + MD:                                                      \
    ex = pinkhip::PINKHIP_LAUNCH_SWEEPX_NAME(NV, MD, W)(h->stream, a);     \
    break;
      PINKHIP_SWEEPX_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    }
    PH_HIP(h, ex);
    return PINKHIP_OK;
  }
  if (sweep) {
    hipError_t es = hipErrorInvalidValue;
    switch (sc.NV * 10000 + sc.MD * 100 + sc.W) {
#define PINKHIP_CASE(NV, MD, W)                                            \
  case NV * 10000 + MD * 100 + W:                                          \
    es = pinkhip::PINKHIP_LAUNCH_SWEEP_NAME(NV, MD, W)(h->stream, a);      \
    break;
      PINKHIP_SWEEP_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    }
    PH_HIP(h, es);
    return PINKHIP_OK;
  }
  const pinkhip::PackedChoice pc = pinkhip::select_packed(a.nv, a.md);
  static const bool force_dense = std::getenv("PINKHIP_FORCE_DENSE") != nullptr;  // development: time the dense-row instantiation on a batch without dense rows
  hipError_t e = hipErrorInvalidValue;
  switch (pc.NV) {
#define PINKHIP_CASE(NV, W)                                                                   \
  case NV:                                                                                    \
    e = (a.md == 0 && !force_dense) ? pinkhip::PINKHIP_LAUNCH_PACKED_NAME(NV, W, 0)(h->stream, a) \
                  : pinkhip::PINKHIP_LAUNCH_PACKED_NAME(NV, W, 1)(h->stream, a);              \
    break;
    PINKHIP_PACKED_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    default: return fail(h, PINKHIP_E_INVALID, "unsupported nv / md");
  }
  PH_HIP(h, e);
  return PINKHIP_OK;
}

// Validate `d`, refresh the device tables if they changed, fill the table part of `a`.
int prepare(pinkhip_handle *h, const pinkhip_desc *d, KernelArgs &a) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (!d) return fail(h, PINKHIP_E_INVALID, "null descriptor");
  pinkhip::HostTables t;
  const std::string why = pinkhip::build_tables(*d, t);
  if (!why.empty()) return fail(h, PINKHIP_E_INVALID, why);
  a.rank_deficient = pinkhip::rank_deficient_by_construction(*d) ? 1 : 0;
  a.n_free_lead = (d->n_free_lead > 0 && d->n_free_lead <= d->nv) ? d->n_free_lead : 0;
  a.out_scale = 1.0;
  PH_HIP(h, hipSetDevice(h->device));

  // pack: [row_gain K][row_lm K][barrier_safe_gain nb] doubles, then int tables
  const size_t K = t.row_gain.size(), nd = t.dtask_k.size(), nb = t.barrier_safe_gain.size();
  const size_t nbytes = 8 * (2 * K + nb) + 4 * (3 * nd + (nb + 1));
  if (nbytes > kTableBytes - 8) return fail(h, PINKHIP_E_INVALID, "task tables exceed 64 KiB");  // (last 8 bytes: result slot of pinkhip_check_limits_device)
  std::vector<char> img(nbytes);
  char *p = img.data();
  auto put = [&](const void *src, size_t n) {
    if (n) std::memcpy(p, src, n);
    p += n;
  };
  put(t.row_gain.data(), 8 * K);
  put(t.row_lm.data(), 8 * K);
  put(t.barrier_safe_gain.data(), 8 * nb);
  put(t.dtask_col0.data(), 4 * nd);
  put(t.dtask_row0.data(), 4 * nd);
  put(t.dtask_k.data(), 4 * nd);
  put(t.barrier_rows.data(), 4 * (nb + 1));
  if (img != h->h_tables) {
    // stream-ordered after any kernel still reading the previous tables
    if (nbytes) PH_HIP(h, hipMemcpyAsync(h->d_tables, img.data(), nbytes, hipMemcpyHostToDevice, h->stream));
    PH_HIP(h, hipStreamSynchronize(h->stream));
    h->h_tables.swap(img);
  }
  char *dp = h->d_tables;
  a.row_gain = reinterpret_cast<const double *>(dp);
  a.row_lm = reinterpret_cast<const double *>(dp + 8 * K);
  a.barrier_safe_gain = reinterpret_cast<const double *>(dp + 16 * K);
  const char *ip = dp + 8 * (2 * K + nb);
  a.dtask_col0 = reinterpret_cast<const int *>(ip);
  a.dtask_row0 = reinterpret_cast<const int *>(ip + 4 * nd);
  a.dtask_k = reinterpret_cast<const int *>(ip + 8 * nd);
  a.barrier_rows = reinterpret_cast<const int *>(ip + 12 * nd);

  a.B = d->B;
  a.nv = d->nv;
  a.Kd = d->Kd;
  a.K = d->K;
  a.md = d->md;
  a.n_eq = d->n_eq;
  a.n_dtasks = static_cast<int>(nd);
  a.n_barriers = static_cast<int>(nb);
  a.cost_batched = d->cost_is_batched;
  a.max_iter = d->max_iter;
  a.damping = d->damping;
  a.dt = d->dt;
  return PINKHIP_OK;
}

int check_problem(pinkhip_handle *h, const pinkhip_desc *d, const pinkhip_problem *in) {
  if (!in) return fail(h, PINKHIP_E_INVALID, "null problem");
  if (d->B == 0) return PINKHIP_OK;
  if (d->Kd > 0 && !in->J) return fail(h, PINKHIP_E_INVALID, "J is NULL but Kd > 0");
  if (d->K > 0 && (!in->e || !in->cost)) return fail(h, PINKHIP_E_INVALID, "e/cost NULL but K > 0");
  if (!in->lb || !in->ub) return fail(h, PINKHIP_E_INVALID, "lb/ub must not be NULL");
  if (d->md > 0 && (!in->Gd || !in->hd)) return fail(h, PINKHIP_E_INVALID, "Gd/hd NULL but md > 0");
  return PINKHIP_OK;
}

void set_problem(KernelArgs &a, const pinkhip_problem *in) {
  a.J = in->J;
  a.e = in->e;
  a.cost = in->cost;
  a.lb = in->lb;
  a.ub = in->ub;
  a.Gd = in->Gd;
  a.hd = in->hd;
  a.c_extra = in->c_extra;
}

size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

int ensure_arena(pinkhip_handle *h, size_t bytes) {
  if (bytes <= h->arena_bytes) return PINKHIP_OK;
  PH_HIP(h, hipStreamSynchronize(h->stream));
  if (h->arena) PH_HIP(h, hipFree(h->arena));
  h->arena = nullptr;
  h->arena_bytes = 0;
  PH_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->arena), bytes));
  h->arena_bytes = bytes;
  return PINKHIP_OK;
}

// Upload the per-instance streams of `in` into the arena; returns device views.
int upload(pinkhip_handle *h, const pinkhip_desc *d, const pinkhip_problem *in, size_t extra_out,
           pinkhip_problem &dev, char *&out_base) {
  const size_t B = static_cast<size_t>(d->B), nv = d->nv;
  const void *src[8] = {in->J, in->e, in->cost, in->lb, in->ub, in->Gd, in->hd, in->c_extra};
  const size_t n[8] = {8 * B * d->Kd * nv,
                       8 * B * d->K,
                       8 * (d->cost_is_batched ? B : 1) * d->K,
                       8 * B * nv,
                       8 * B * nv,
                       8 * B * d->md * nv,
                       8 * B * d->md,
                       in->c_extra ? 8 * B * nv : 0};
  size_t off[8], total = 0;
  for (int i = 0; i < 8; ++i) {
    off[i] = total;
    total += align256(n[i]);
  }
  const size_t out_off = total;
  total += align256(extra_out);
  int rc = ensure_arena(h, total);
  if (rc) return rc;
  const double *dp[8];
  for (int i = 0; i < 8; ++i) {
    dp[i] = (n[i] && src[i]) ? reinterpret_cast<const double *>(h->arena + off[i]) : nullptr;
    if (n[i] && src[i])
      PH_HIP(h, hipMemcpyAsync(h->arena + off[i], src[i], n[i], hipMemcpyHostToDevice, h->stream));
  }
  dev.J = dp[0];
  dev.e = dp[1];
  dev.cost = dp[2];
  dev.lb = dp[3];
  dev.ub = dp[4];
  dev.Gd = dp[5];
  dev.hd = dp[6];
  dev.c_extra = dp[7];
  out_base = h->arena + out_off;
  return PINKHIP_OK;
}

}  // namespace

extern "C" {

int pinkhip_version(void) { return PINKHIP_VERSION; }

int pinkhip_device_count(int *count) {
  if (!count) return fail(nullptr, PINKHIP_E_INVALID, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(nullptr, PINKHIP_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return PINKHIP_OK;
}

int pinkhip_create(pinkhip_handle **out, int device_id) {
  if (!out) return fail(nullptr, PINKHIP_E_INVALID, "null handle pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(nullptr, PINKHIP_E_NODEVICE, "no HIP device visible (libpinkhip has no CPU path)");
  if (device_id < 0 || device_id >= n) return fail(nullptr, PINKHIP_E_INVALID, "device_id out of range");
  pinkhip_handle *h = new (std::nothrow) pinkhip_handle();
  if (!h) return fail(nullptr, PINKHIP_E_NOMEM, "out of host memory");
  h->device = device_id;
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipGetDeviceProperties(&h->prop, device_id);
  if (e == hipSuccess && std::strncmp(h->prop.gcnArchName, "gfx950", 6) != 0) {
    const std::string arch = h->prop.gcnArchName;
    delete h;
    return fail(nullptr, PINKHIP_E_NODEVICE, "device is " + arch + ", kernels are built for gfx950 only");
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->d2h_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_kernels, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&h->ev0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&h->evk0);
  if (e == hipSuccess) e = hipEventCreate(&h->evk1);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h->d_tables), kTableBytes);
  if (e != hipSuccess) {
    const std::string msg = std::string("pinkhip_create: ") + hipGetErrorString(e);
    pinkhip_destroy(h);
    return fail(nullptr, PINKHIP_E_HIP, msg);
  }
  *out = h;
  return PINKHIP_OK;
}

int pinkhip_destroy(pinkhip_handle *h) {
  if (!h) return PINKHIP_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm) pinkhip_comm_destroy(h);
  if (h->arena) (void)hipFree(h->arena);
  if (h->stage) (void)hipHostFree(h->stage);
  if (h->d_tables) (void)hipFree(h->d_tables);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->evk0) (void)hipEventDestroy(h->evk0);
  if (h->evk1) (void)hipEventDestroy(h->evk1);
  for (hipEvent_t ev : h->chunk_events) (void)hipEventDestroy(ev);
  if (h->ev_copy) (void)hipEventDestroy(h->ev_copy);
  if (h->ev_kernels) (void)hipEventDestroy(h->ev_kernels);
  if (h->d2h_stream) (void)hipStreamDestroy(h->d2h_stream);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->main_stream) h->stream = h->main_stream;
  if (h->alt_stream) {
    (void)hipStreamSynchronize(h->alt_stream);
    (void)hipStreamDestroy(h->alt_stream);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PINKHIP_OK;
}

const char *pinkhip_last_error(const pinkhip_handle *h) {
  return h ? h->err.c_str() : g_last_error.c_str();
}

int pinkhip_get_device_info(const pinkhip_handle *h, pinkhip_device_info *info) {
  if (!h || !info) return fail(nullptr, PINKHIP_E_INVALID, "null argument");
  std::memset(info, 0, sizeof(*info));
  info->device_id = h->device;
  info->compute_units = h->prop.multiProcessorCount;
  info->wavefront_size = h->prop.warpSize;
  info->clock_mhz = h->prop.clockRate / 1000;
  info->total_mem_bytes = static_cast<int64_t>(h->prop.totalGlobalMem);
  info->lds_per_cu_bytes = static_cast<int64_t>(h->prop.maxSharedMemoryPerMultiProcessor);
  std::snprintf(info->name, sizeof(info->name), "%s", h->prop.name);
  std::snprintf(info->gcn_arch, sizeof(info->gcn_arch), "%s", h->prop.gcnArchName);
  return PINKHIP_OK;
}

int pinkhip_solve_device(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_problem *dev_in,
                         const pinkhip_result *dev_out) {
  KernelArgs a{};
  int rc = prepare(h, desc, a);
  if (rc) return rc;
  if ((rc = check_problem(h, desc, dev_in))) return rc;
  if (!dev_out || (desc->B > 0 && (!dev_out->dq || !dev_out->status)))
    return fail(h, PINKHIP_E_INVALID, "dq/status must not be NULL");
  set_problem(a, dev_in);
  a.dq = dev_out->dq;
  a.status = dev_out->status;
  a.iters = dev_out->iters;
  return launch(h, a, true);
}

int pinkhip_stack_device(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_problem *dev_in,
                         double *H_out, double *c_out) {
  KernelArgs a{};
  int rc = prepare(h, desc, a);
  if (rc) return rc;
  if ((rc = check_problem(h, desc, dev_in))) return rc;
  if (desc->B > 0 && (!H_out || !c_out)) return fail(h, PINKHIP_E_INVALID, "H_out/c_out must not be NULL");
  set_problem(a, dev_in);
  a.H_out = H_out;
  a.c_out = c_out;
  return launch(h, a, false);
}

// Device layout of one uploaded batch inside the arena (offsets of the eight input streams + outputs).
namespace {
constexpr size_t kStageBytes = size_t(256) << 10;  // *_host batches up to this size go through the staging buffer
struct ArenaPlan {
  size_t n[8], off[8], stride[8];  // bytes, offset, bytes per instance (0: broadcast stream)
  size_t out_off, total;
};
ArenaPlan plan_arena(const pinkhip_desc *d, const pinkhip_problem *in, size_t extra_out) {
  const size_t B = static_cast<size_t>(d->B), nv = d->nv;
  ArenaPlan p{};
  const size_t per[8] = {8 * (size_t)d->Kd * nv, 8 * (size_t)d->K, d->cost_is_batched ? 8 * (size_t)d->K : 0, 8 * nv, 8 * nv,
                         8 * (size_t)d->md * nv, 8 * (size_t)d->md, in->c_extra ? 8 * nv : 0};
  size_t total = 0;
  for (int i = 0; i < 8; ++i) {
    p.stride[i] = per[i];
    p.n[i] = per[i] * B;
    if (i == 2 && !d->cost_is_batched) p.n[i] = 8 * (size_t)d->K;
    p.off[i] = total;
    total += align256(p.n[i]);
  }
  p.out_off = total;
  p.total = total + align256(extra_out);
  return p;
}
}  // namespace

int pinkhip_solve_host(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_problem *host_in,
                       const pinkhip_result *host_out) {
  KernelArgs a{};
  int rc = prepare(h, desc, a);
  if (rc) return rc;
  if ((rc = check_problem(h, desc, host_in))) return rc;
  if (!host_out || (desc->B > 0 && (!host_out->dq || !host_out->status)))
    return fail(h, PINKHIP_E_INVALID, "dq/status must not be NULL");
  if (desc->B == 0) return PINKHIP_OK;
  const size_t B = static_cast<size_t>(desc->B), nv = desc->nv;
  const size_t n_dq = align256(8 * B * nv), n_st = align256(4 * B);
  const ArenaPlan p = plan_arena(desc, host_in, n_dq + 2 * n_st);
  const void *src[8] = {host_in->J, host_in->e, host_in->cost, host_in->lb, host_in->ub, host_in->Gd, host_in->hd, host_in->c_extra};
  if (p.total <= kStageBytes) {
    // Small batch (a single robot at its control rate is the reference's own use, examples/arm_ur5.py:62-86): what
    // the call costs is API round trips, not bytes.  Inputs are copied by the CPU into one page-locked buffer that
    // the kernel reads -- and writes its results to -- across PCIe: one launch and one synchronisation instead of
    // eight asynchronous copies, two events and two synchronisations.
    if (!h->stage) PH_HIP(h, hipHostMalloc(reinterpret_cast<void **>(&h->stage), kStageBytes, hipHostMallocDefault));
    char *dbase = nullptr;
    PH_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&dbase), h->stage, 0));
    auto at = [&](int i) -> const double * {
      if (!(p.n[i] && src[i])) return nullptr;
      std::memcpy(h->stage + p.off[i], src[i], p.n[i]);
      return reinterpret_cast<const double *>(dbase + p.off[i]);
    };
    a.J = at(0), a.e = at(1), a.cost = at(2), a.lb = at(3), a.ub = at(4), a.Gd = at(5), a.hd = at(6), a.c_extra = at(7);
    a.dq = reinterpret_cast<double *>(dbase + p.out_off);
    a.status = reinterpret_cast<int *>(dbase + p.out_off + n_dq);
    a.iters = reinterpret_cast<int *>(dbase + p.out_off + n_dq + n_st);
    h->evk_valid = false;  // (the point of this path is to spare API calls: not timed)
    if ((rc = launch(h, a, true))) return rc;
    PH_HIP(h, hipStreamSynchronize(h->stream));
    const char *o = h->stage + p.out_off;
    std::memcpy(host_out->dq, o, 8 * B * nv);
    std::memcpy(host_out->status, o + n_dq, 4 * B);
    if (host_out->iters) std::memcpy(host_out->iters, o + n_dq + n_st, 4 * B);
    return PINKHIP_OK;
  }
  if ((rc = ensure_arena(h, p.total))) return rc;
  char *dev[8];
  for (int i = 0; i < 8; ++i) dev[i] = (p.n[i] && src[i]) ? h->arena + p.off[i] : nullptr;
  char *out = h->arena + p.out_off;
  double *d_dq = reinterpret_cast<double *>(out);
  int *d_st = reinterpret_cast<int *>(out + n_dq), *d_it = reinterpret_cast<int *>(out + n_dq + n_st);

  // The batch is cut into chunks of ~32 MB: the H2D copy of chunk c + 1 (copy stream) runs while chunk c is
  // solved and its results go home (compute stream).  From pinned host memory (pinkhip_host_alloc) the copies
  // are true DMA and the call approaches the PCIe rate; pageable buffers are staged by the HIP runtime and
  // still overlap with the kernels.
  size_t per_inst = 0;
  for (int i = 0; i < 8; ++i) per_inst += src[i] ? p.stride[i] : 0;
  size_t chunk = per_inst ? (size_t(32) << 20) / per_inst : B;
  chunk = chunk < 2048 ? 2048 : (chunk & ~size_t(63));
  {  // pageable source: every hipMemcpyAsync is staged synchronously by the runtime, one big copy per stream is
     // cheaper than many small ones (measured: 9.8 ms in one piece, 11.7 ms in 32 MB chunks for 434 MB)
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, src[0] ? src[0] : src[1]) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!pinned) chunk = B;
  }
  const size_t n_chunks = (B + chunk - 1) / chunk;
  if (n_chunks > h->chunk_events.size()) {
    const size_t old = h->chunk_events.size();
    h->chunk_events.resize(n_chunks, nullptr);
    for (size_t i = old; i < n_chunks; ++i) PH_HIP(h, hipEventCreateWithFlags(&h->chunk_events[i], hipEventDisableTiming));
  }
  // broadcast stream (cost [K]) once, ahead of the first chunk
  if (!desc->cost_is_batched && dev[2]) PH_HIP(h, hipMemcpyAsync(dev[2], src[2], p.n[2], hipMemcpyHostToDevice, h->copy_stream));
  // the copy stream must not overwrite the arena while an earlier call's kernels still read it
  PH_HIP(h, hipStreamSynchronize(h->stream));
  for (size_t c = 0; c < n_chunks; ++c) {
    const size_t c0 = c * chunk, cb = (B - c0 < chunk) ? B - c0 : chunk;
    for (int i = 0; i < 8; ++i)
      if (dev[i] && p.stride[i])
        PH_HIP(h, hipMemcpyAsync(dev[i] + c0 * p.stride[i], static_cast<const char *>(src[i]) + c0 * p.stride[i],
                                 cb * p.stride[i], hipMemcpyHostToDevice, h->copy_stream));
    PH_HIP(h, hipEventRecord(h->chunk_events[c], h->copy_stream));
    PH_HIP(h, hipStreamWaitEvent(h->stream, h->chunk_events[c], 0));
    KernelArgs ac = a;
    ac.B = static_cast<long long>(cb);
    auto at = [&](int i) { return dev[i] ? reinterpret_cast<const double *>(dev[i] + c0 * p.stride[i]) : nullptr; };
    ac.J = at(0), ac.e = at(1), ac.cost = at(2), ac.lb = at(3), ac.ub = at(4), ac.Gd = at(5), ac.hd = at(6), ac.c_extra = at(7);
    ac.dq = d_dq + c0 * nv;
    ac.status = d_st + c0;
    ac.iters = d_it + c0;
    if (c == 0) PH_HIP(h, hipEventRecord(h->evk0, h->stream));
    if ((rc = launch(h, ac, true))) return rc;
    if (c + 1 == n_chunks) PH_HIP(h, hipEventRecord(h->evk1, h->stream));
    PH_HIP(h, hipMemcpyAsync(host_out->dq + c0 * nv, ac.dq, 8 * cb * nv, hipMemcpyDeviceToHost, h->stream));
    PH_HIP(h, hipMemcpyAsync(host_out->status + c0, ac.status, 4 * cb, hipMemcpyDeviceToHost, h->stream));
    if (host_out->iters) PH_HIP(h, hipMemcpyAsync(host_out->iters + c0, ac.iters, 4 * cb, hipMemcpyDeviceToHost, h->stream));
  }
  PH_HIP(h, hipStreamSynchronize(h->stream));
  h->evk_valid = true;
  return PINKHIP_OK;
}

int pinkhip_last_kernel_ms(pinkhip_handle *h, float *ms) {
  if (!h || !ms) return fail(h, PINKHIP_E_INVALID, "bad argument");
  *ms = -1.0f;
  if (!h->evk_valid) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipEventElapsedTime(ms, h->evk0, h->evk1));
  return PINKHIP_OK;
}

int pinkhip_stack_host(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_problem *host_in,
                       double *H_out, double *c_out) {
  KernelArgs a{};
  int rc = prepare(h, desc, a);
  if (rc) return rc;
  if ((rc = check_problem(h, desc, host_in))) return rc;
  if (desc->B == 0) return PINKHIP_OK;
  if (!H_out || !c_out) return fail(h, PINKHIP_E_INVALID, "H_out/c_out must not be NULL");
  const size_t B = static_cast<size_t>(desc->B), nv = desc->nv;
  const size_t n_H = align256(8 * B * nv * nv), n_c = align256(8 * B * nv);
  pinkhip_problem dev{};
  char *out = nullptr;
  if ((rc = upload(h, desc, host_in, n_H + n_c, dev, out))) return rc;
  set_problem(a, &dev);
  a.H_out = reinterpret_cast<double *>(out);
  a.c_out = reinterpret_cast<double *>(out + n_H);
  if ((rc = launch(h, a, false))) return rc;
  PH_HIP(h, hipMemcpyAsync(H_out, a.H_out, 8 * B * nv * nv, hipMemcpyDeviceToHost, h->stream));
  PH_HIP(h, hipMemcpyAsync(c_out, a.c_out, 8 * B * nv, hipMemcpyDeviceToHost, h->stream));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  return PINKHIP_OK;
}

int pinkhip_frame_task_device(pinkhip_handle *h, int64_t B, int32_t nv, const double *T_frame,
                              const double *T_target, const double *J_body, double *e_out,
                              double *J_out) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (B < 0 || nv < 1 || nv > PINKHIP_MAX_NV) return fail(h, PINKHIP_E_INVALID, "bad B / nv");
  if (B == 0) return PINKHIP_OK;
  if (!T_frame || !T_target || !J_body || !e_out || !J_out) return fail(h, PINKHIP_E_INVALID, "null pointer");
  if (B > 0x7fffffffLL) return fail(h, PINKHIP_E_INVALID, "B exceeds the grid limit");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::FrameTaskArgs a{B, nv, T_frame, T_target, J_body, e_out, J_out};
  const dim3 block(pinkhip::kWave);
  if (nv <= 8) {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<8>, dim3((unsigned)((B + 7) / 8)), block, 0, h->stream, a);
  } else if (nv <= 16) {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<16>, dim3((unsigned)((B + 3) / 4)), block, 0, h->stream, a);
  } else if (nv <= 32) {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<32>, dim3((unsigned)((B + 1) / 2)), block, 0, h->stream, a);
  } else {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<64>, dim3((unsigned)B), block, 0, h->stream, a);
  }
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_frame_task_host(pinkhip_handle *h, int64_t B, int32_t nv, const double *T_frame,
                            const double *T_target, const double *J_body, double *e_out,
                            double *J_out) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (B < 0 || nv < 1 || nv > PINKHIP_MAX_NV) return fail(h, PINKHIP_E_INVALID, "bad B / nv");
  if (B == 0) return PINKHIP_OK;
  if (!T_frame || !T_target || !J_body || !e_out || !J_out) return fail(h, PINKHIP_E_INVALID, "null pointer");
  PH_HIP(h, hipSetDevice(h->device));
  const size_t nT = align256(8 * (size_t)B * 12), nJ = align256(8 * (size_t)B * 6 * nv), nE = align256(8 * (size_t)B * 6);
  int rc = ensure_arena(h, 2 * nT + 2 * nJ + nE);
  if (rc) return rc;
  char *p = h->arena;
  double *dTf = reinterpret_cast<double *>(p), *dTt = reinterpret_cast<double *>(p + nT);
  double *dJb = reinterpret_cast<double *>(p + 2 * nT), *dJo = reinterpret_cast<double *>(p + 2 * nT + nJ);
  double *dE = reinterpret_cast<double *>(p + 2 * nT + 2 * nJ);
  PH_HIP(h, hipMemcpyAsync(dTf, T_frame, 8 * (size_t)B * 12, hipMemcpyHostToDevice, h->stream));
  PH_HIP(h, hipMemcpyAsync(dTt, T_target, 8 * (size_t)B * 12, hipMemcpyHostToDevice, h->stream));
  PH_HIP(h, hipMemcpyAsync(dJb, J_body, 8 * (size_t)B * 6 * nv, hipMemcpyHostToDevice, h->stream));
  if ((rc = pinkhip_frame_task_device(h, B, nv, dTf, dTt, dJb, dE, dJo))) return rc;
  PH_HIP(h, hipMemcpyAsync(e_out, dE, 8 * (size_t)B * 6, hipMemcpyDeviceToHost, h->stream));
  PH_HIP(h, hipMemcpyAsync(J_out, dJo, 8 * (size_t)B * 6 * nv, hipMemcpyDeviceToHost, h->stream));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  return PINKHIP_OK;
}

int pinkhip_frame_task_strided_device(pinkhip_handle *h, int64_t B, int32_t nv, const double *T_frame,
                                      int64_t sT_frame, const double *T_target, int64_t sT_target,
                                      const double *J_body, int64_t sJ_body, double *e_out, int64_t sE,
                                      double *J_out, int64_t sJ_out) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (B < 0 || nv < 1 || nv > PINKHIP_MAX_NV) return fail(h, PINKHIP_E_INVALID, "bad B / nv");
  if (B == 0) return PINKHIP_OK;
  if (!T_frame || !T_target || !J_body || !e_out || !J_out) return fail(h, PINKHIP_E_INVALID, "null pointer");
  if (B > 0x7fffffffLL) return fail(h, PINKHIP_E_INVALID, "B exceeds the grid limit");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::FrameTaskArgs a{B, nv, T_frame, T_target, J_body, e_out, J_out, sT_frame, sT_target, sJ_body, sE, sJ_out};
  const dim3 block(pinkhip::kWave);
  if (nv <= 8) {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<8>, dim3((unsigned)((B + 7) / 8)), block, 0, h->stream, a);
  } else if (nv <= 16) {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<16>, dim3((unsigned)((B + 3) / 4)), block, 0, h->stream, a);
  } else if (nv <= 32) {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<32>, dim3((unsigned)((B + 1) / 2)), block, 0, h->stream, a);
  } else {
    hipLaunchKernelGGL(pinkhip::ik_frame_task_kernel<64>, dim3((unsigned)B), block, 0, h->stream, a);
  }
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_model_create(pinkhip_handle *h, const pinkhip_model_desc *desc, pinkhip_model **out) {
  if (!h || !desc || !out) return fail(h, PINKHIP_E_INVALID, "null argument");
  *out = nullptr;
  pinkhip_model *m = new (std::nothrow) pinkhip_model();
  if (!m) return fail(h, PINKHIP_E_NOMEM, "out of host memory");
  const std::string why = pinkhip::build_model_image(*desc, m->image);
  if (!why.empty()) {
    delete m;
    return fail(h, PINKHIP_E_INVALID, why);
  }
  hipError_t e = hipSetDevice(h->device);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_base), m->image.bytes.size());
  if (e == hipSuccess) e = hipMemcpy(m->d_base, m->image.bytes.data(), m->image.bytes.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (m->d_base) (void)hipFree(m->d_base);
    delete m;
    return fail(h, PINKHIP_E_HIP, std::string("pinkhip_model_create: ") + hipGetErrorString(e));
  }
  m->dev = pinkhip::model_view<pinkhip::ModelDev>(m->image, m->d_base);
  *out = m;
  return PINKHIP_OK;
}

int pinkhip_model_destroy(pinkhip_handle *h, pinkhip_model *m) {
  if (!m) return PINKHIP_OK;
  if (h) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
  }
  if (m->d_base) (void)hipFree(m->d_base);
  delete m;
  return PINKHIP_OK;
}

int pinkhip_fk_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, const double *q, double *T_frames,
                      double *J_body) {
  if (!h || !m) return fail(h, PINKHIP_E_INVALID, "null handle / model");
  if (B < 0 || B > 0x7fffffffLL) return fail(h, PINKHIP_E_INVALID, "bad B");
  if (B == 0) return PINKHIP_OK;
  if (!q || (m->dev.nf > 0 && (!T_frames || !J_body))) return fail(h, PINKHIP_E_INVALID, "null pointer");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::FkArgs a{m->dev, B, q, T_frames, J_body};
  const int per = pinkhip::fk_lds_doubles(m->dev.nj, m->dev.nf);
  const dim3 block(pinkhip::kWave);
  const int width = m->dev.nv > m->dev.nj ? m->dev.nv : m->dev.nj;  // lanes per instance: one per joint / column
  if (width <= 8) {
    hipLaunchKernelGGL(pinkhip::ik_fk_kernel<8>, dim3((unsigned)((B + 7) / 8)), block, 8 * 8 * per + 16, h->stream, a);
  } else if (width <= 32) {
    hipLaunchKernelGGL(pinkhip::ik_fk_kernel<32>, dim3((unsigned)((B + 1) / 2)), block, 8 * 2 * per + 16, h->stream, a);
  } else {
    hipLaunchKernelGGL(pinkhip::ik_fk_kernel<64>, dim3((unsigned)B), block, 8 * per + 16, h->stream, a);
  }
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_fk_frame_tasks_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, const double *q,
                                  const double *T_target, double *T_frames, double *e, int64_t sE, double *J,
                                  int64_t sJ) {
  if (!h || !m) return fail(h, PINKHIP_E_INVALID, "null handle / model");
  if (B < 0 || B > 0x7fffffffLL) return fail(h, PINKHIP_E_INVALID, "bad B");
  if (B == 0 || m->dev.nf == 0) return PINKHIP_OK;
  if (!q || !T_target || !e || !J) return fail(h, PINKHIP_E_INVALID, "null pointer");
  if (sE < 6 * m->dev.nf || sJ < 6LL * m->dev.nf * m->dev.nv) return fail(h, PINKHIP_E_INVALID, "strides smaller than the frame rows");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::FkArgs a{m->dev, B, q, T_frames, nullptr};
  a.T_target = T_target;
  a.e_out = e;
  a.J_out = J;
  a.sE = sE;
  a.sJo = sJ;
  const int per = pinkhip::fk_lds_doubles(m->dev.nj, m->dev.nf);
  const dim3 block(pinkhip::kWave);
  const int width = m->dev.nv > m->dev.nj ? m->dev.nv : m->dev.nj;
  if (width <= 8) {
    hipLaunchKernelGGL(pinkhip::ik_fk_frame_tasks_kernel<8>, dim3((unsigned)((B + 7) / 8)), block, 8 * 8 * per + 16, h->stream, a);
  } else if (width <= 32) {
    hipLaunchKernelGGL(pinkhip::ik_fk_frame_tasks_kernel<32>, dim3((unsigned)((B + 1) / 2)), block, 8 * 2 * per + 16, h->stream, a);
  } else {
    hipLaunchKernelGGL(pinkhip::ik_fk_frame_tasks_kernel<64>, dim3((unsigned)B), block, 8 * per + 16, h->stream, a);
  }
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_step_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, const pinkhip_step *st) {
  if (!h || !m || !st) return fail(h, PINKHIP_E_INVALID, "null handle / model / args");
  if (B < 0 || B > 0x7fffffffLL) return fail(h, PINKHIP_E_INVALID, "bad B");
  if (B == 0) return PINKHIP_OK;
  if (!st->q || !st->lb || !st->ub) return fail(h, PINKHIP_E_INVALID, "q / lb / ub must not be NULL");
  if (m->dev.nf > 0 && (!st->T_target || !st->e || !st->J)) return fail(h, PINKHIP_E_INVALID, "frame-task streams must not be NULL");
  if (st->dq_prev && !st->status) return fail(h, PINKHIP_E_INVALID, "dq_prev needs the status of its solve");
  if (st->q_target && !st->e) return fail(h, PINKHIP_E_INVALID, "posture rows need e");
  if (st->sE < 6 * m->dev.nf || st->sJ < 6LL * m->dev.nf * m->dev.nv) return fail(h, PINKHIP_E_INVALID, "strides smaller than the frame rows");
  if (st->q_target && (st->e_off < 0 || st->e_off + m->dev.nv - m->dev.root_nv > st->sE)) return fail(h, PINKHIP_E_INVALID, "posture rows exceed the row stride");
  if (!(st->dt > 0.0) || !(st->config_limit_gain > 0.0 && st->config_limit_gain <= 1.0)) return fail(h, PINKHIP_E_INVALID, "bad dt / gain");
  if (st->step < 0 || st->step >= (1 << 23)) return fail(h, PINKHIP_E_INVALID, "bad step");
  PH_HIP(h, hipSetDevice(h->device));
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
  const int per = pinkhip::fk_lds_doubles(m->dev.nj, m->dev.nf);
  const dim3 block(pinkhip::kWave);
  const int width = m->dev.nv > m->dev.nj ? m->dev.nv : m->dev.nj;
  if (width <= 8) {
    hipLaunchKernelGGL(pinkhip::ik_step_kernel<8>, dim3((unsigned)((B + 7) / 8)), block, 8 * 8 * per + 16, h->stream, a);
  } else if (width <= 32) {
    hipLaunchKernelGGL(pinkhip::ik_step_kernel<32>, dim3((unsigned)((B + 1) / 2)), block, 8 * 2 * per + 16, h->stream, a);
  } else {
    hipLaunchKernelGGL(pinkhip::ik_step_kernel<64>, dim3((unsigned)B), block, 8 * per + 16, h->stream, a);
  }
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_rollout_step_device(pinkhip_handle *h, const pinkhip_desc *desc, const pinkhip_model *m,
                                const pinkhip_rollout_step *st) {
  if (!h || !m || !st) return fail(h, PINKHIP_E_INVALID, "null handle / model / args");
  pinkhip::RolloutArgs ra{};
  int rc = prepare(h, desc, ra.k);
  if (rc) return rc;
  const pinkhip::ModelDev &md = m->dev;
  if (desc->B == 0) return PINKHIP_OK;
  const int n_crow = st->n_const_rows;
  if (n_crow < 0 || (n_crow > 0 && (!st->const_rows || !st->const_q0 || !st->const_b)))
    return fail(h, PINKHIP_E_INVALID, "n_const_rows must be >= 0 and come with const_rows / const_q0 / const_b");
  const int n_eqf = st->n_constraint_frames;
  if (n_eqf < 0 || n_eqf > pinkhip::kRolloutMaxEqFrames || (n_eqf > 0 && (!st->constraint_frame || !st->constraint_gain)))
    return fail(h, PINKHIP_E_INVALID, "n_constraint_frames must lie in [0, 2] and come with constraint_frame / constraint_gain");
  if (desc->nv != md.nv || desc->n_eq != 6 * n_eqf)
    return fail(h, PINKHIP_E_INVALID, "descriptor does not describe this model's task stack (nv, n_eq = 6 n_constraint_frames)");
  if (st->n_limit_rows < 0 || 6 * n_eqf + st->n_limit_rows > desc->md || (st->n_limit_rows > 0 && (!st->limit_rows || !st->limit_h)))
    return fail(h, PINKHIP_E_INVALID, "n_limit_rows must lie in [0, md - n_eq] and come with limit_rows / limit_h");
  if (desc->md > 6 * n_eqf + st->n_limit_rows &&
      (!st->barrier_frame || !st->barrier_axis || !st->barrier_sign || !st->barrier_bound || !st->barrier_gain || !st->barrier_frame2))
    return fail(h, PINKHIP_E_INVALID, "barrier rows need the barrier_* tables, barrier_frame2 included (-1 for the rows of a position barrier)");
  if ((st->root_box || st->n_limit_rows) && md.root_nv != 6)
    return fail(h, PINKHIP_E_INVALID, "a floating-base velocity limit needs a free-flyer root joint");
  int post_row0 = 0, post_k = 0;
  {
    const std::string why = pinkhip::rollout_task_layout(*desc, md.nf, md.nv, md.root_nv, n_crow, st->posture_task, st->diag_error != nullptr, post_row0, post_k);
    if (!why.empty()) return fail(h, PINKHIP_E_INVALID, why);
  }
  const int n_post = post_k;
  if (!st->q || !st->cost || !st->dq || !st->status || (md.nf > 0 && !st->T_target) || (n_post && !st->q_target))
    return fail(h, PINKHIP_E_INVALID, "null pointer");
  if (!(st->config_limit_gain > 0.0 && st->config_limit_gain <= 1.0) || st->step < 0 || st->step >= (1 << 23))
    return fail(h, PINKHIP_E_INVALID, "bad limit gain / step");
  const int fkd = pinkhip::rollout_fk_doubles(md.nj, md.nf, n_crow);
  ra.n_crow = n_crow;
  ra.crow_A = st->const_rows;
  ra.crow_q0 = st->const_q0;
  ra.crow_b = st->const_b;
  ra.post_row0 = post_row0;
  ra.post_k = post_k;
  ra.diag_e = st->diag_error;
  pinkhip::PackedChoice pc{0, 0};
  pinkhip::SweepChoice dc{0, 0, 0};
  if (desc->md > 0) {
    dc = pinkhip::select_rollout_dense(md.nv, md.nj, fkd, desc->md, md.nf, n_eqf);
    if (dc.NV == 0 || md.nf > 32) return fail(h, PINKHIP_E_UNSUPPORTED, "no whole-step instantiation with barrier rows fits this model");
    ra.k.lds_pitch = pinkhip::rollout_lds_doubles(dc.NV, dc.W, fkd, dc.MD, md.nf, n_eqf);
    ra.bar_frame = st->barrier_frame;
    ra.bar_axis = st->barrier_axis;
    ra.bar_sign = st->barrier_sign;
    ra.bar_bound = st->barrier_bound;
    ra.bar_gain = st->barrier_gain;
    ra.n_lim = st->n_limit_rows;
    ra.lim_rows = st->limit_rows;
    ra.lim_h = st->limit_h;
    ra.n_eqf = n_eqf;
    ra.eq_frame = st->constraint_frame;
    ra.eq_gain = st->constraint_gain;
    ra.bar_frame2 = st->barrier_frame2;
  } else {
    pc = pinkhip::select_rollout(md.nv, md.nj, fkd, st->n_const_rows > 0 || st->diag_error != nullptr || st->acc_limit != nullptr || m->image.has_relative);
    if (pc.NV == 0 || md.nf > 32) return fail(h, PINKHIP_E_UNSUPPORTED, "no whole-step instantiation fits this model");
    ra.k.lds_pitch = pinkhip::rollout_lds_doubles(pc.NV, pc.W, fkd);
  }
  ra.k.cost = st->cost;
  ra.k.out_scale = (st->dq_scale != 0.0) ? st->dq_scale : 1.0;
  if (st->dq_scale != 0.0 && st->dq_scale != 1.0 && st->integrate)
    return fail(h, PINKHIP_E_INVALID, "dq_scale rescales what is written to dq: not together with integrate (the next step reads dq)");
  ra.k.dq = st->dq;
  ra.k.status = st->status;
  ra.k.iters = st->iters;
  pinkhip::FkArgs &f = ra.fk;
  f.m = md;
  f.B = desc->B;
  f.q = st->q;
  f.q_rw = st->q;
  f.T_frames = st->T_frames;
  f.T_target = st->T_target;
  f.sTb = st->sT_b;
  f.sTf = (st->sT_b || st->sT_f) ? st->sT_f : 12;
  f.dt = desc->dt;
  f.config_limit_gain = st->config_limit_gain;
  f.root_box = st->root_box;
  f.acc_limit = st->acc_limit;
  f.q_target = n_post ? st->q_target : nullptr;
  f.target_batched = st->target_batched;
  ra.integrate = st->integrate;
  ra.first_failure = st->first_failure;
  ra.step = st->step;
  hipError_t e = hipErrorInvalidValue;
  if (desc->md > 0) {
    switch (dc.NV * 100 + dc.MD) {
#define PINKHIP_CASE(NV, MD, W) \
  case NV * 100 + MD: e = pinkhip::PINKHIP_LAUNCH_ROLLOUT_DENSE_NAME(NV, MD, W)(h->stream, ra); break;
      PINKHIP_ROLLOUT_DENSE_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    }
  } else {
    switch (pc.NV) {
#define PINKHIP_CASE(NV, W) \
  case NV: e = pinkhip::PINKHIP_LAUNCH_ROLLOUT_NAME(NV, W)(h->stream, ra); break;
      PINKHIP_ROLLOUT_TABLE(PINKHIP_CASE)
#undef PINKHIP_CASE
    }
  }
  PH_HIP(h, e);
  return PINKHIP_OK;
}

int pinkhip_limits_posture_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, double dt,
                                  double config_limit_gain, const double *q, const double *q_target,
                                  int32_t target_batched, double *lb, double *ub, double *e, int32_t K,
                                  int32_t e_off) {
  if (!h || !m) return fail(h, PINKHIP_E_INVALID, "null handle / model");
  if (B < 0) return fail(h, PINKHIP_E_INVALID, "bad B");
  if (B == 0) return PINKHIP_OK;
  if (!q || !lb || !ub || (e && !q_target)) return fail(h, PINKHIP_E_INVALID, "null pointer");
  if (e && (e_off < 0 || e_off + m->dev.nv - m->dev.root_nv > K)) return fail(h, PINKHIP_E_INVALID, "posture rows exceed K");
  if (!(dt > 0.0) || !(config_limit_gain > 0.0 && config_limit_gain <= 1.0)) return fail(h, PINKHIP_E_INVALID, "bad dt / gain");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::LimitsPostureArgs a{m->dev, B, dt, config_limit_gain, q, q_target, target_batched, lb, ub, e, K, e_off};
  const long long n = B * m->dev.nv;
  hipLaunchKernelGGL(pinkhip::ik_limits_posture_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_check_limits_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, const double *q, double tol,
                                int64_t *first_bad) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (!m || !first_bad || (B > 0 && !q)) return fail(h, PINKHIP_E_INVALID, "null pointer");
  *first_bad = -1;
  if (B <= 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  // the result slot: eight bytes of the handle's table area (not used by any kernel of this call)
  long long *slot = reinterpret_cast<long long *>(h->d_tables + kTableBytes - 8);
  const long long none = 0x7fffffffffffffffLL;
  PH_HIP(h, hipMemcpyAsync(slot, &none, 8, hipMemcpyHostToDevice, h->stream));
  pinkhip::CheckLimitsArgs a{m->dev, B, q, tol, 0, slot};
  a.start = m->image.root_nv == 6 ? 7 : m->image.root_nv;  // a free-flyer root has 7 configuration entries
  const long long n = B * m->dev.nq;
  hipLaunchKernelGGL(pinkhip::ik_check_limits_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, h->stream, a);
  PH_HIP(h, hipGetLastError());
  long long got = none;
  PH_HIP(h, hipMemcpyAsync(&got, slot, 8, hipMemcpyDeviceToHost, h->stream));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  *first_bad = (got == none) ? -1 : got;
  return PINKHIP_OK;
}

int pinkhip_pose_targets_device(pinkhip_handle *h, int64_t B, const double *pq, double *T) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (B < 0) return fail(h, PINKHIP_E_INVALID, "bad B");
  if (B == 0) return PINKHIP_OK;
  if (!pq || !T) return fail(h, PINKHIP_E_INVALID, "null pointer");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::PoseTargetsArgs a{B, pq, T};
  hipLaunchKernelGGL(pinkhip::ik_pose_targets_kernel, dim3(static_cast<unsigned>((B + 255) / 256)), dim3(256), 0, h->stream, a);
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_integrate_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, double *q, const double *dq) {
  if (!h || !m) return fail(h, PINKHIP_E_INVALID, "null handle / model");
  if (B < 0) return fail(h, PINKHIP_E_INVALID, "bad B");
  if (B == 0) return PINKHIP_OK;
  if (!q || !dq) return fail(h, PINKHIP_E_INVALID, "null pointer");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::IntegrateArgs a{m->dev, B, q, dq};
  const long long n = B * m->dev.nj;
  hipLaunchKernelGGL(pinkhip::ik_integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

int pinkhip_integrate_checked_device(pinkhip_handle *h, const pinkhip_model *m, int64_t B, double *q, const double *dq,
                                     const int32_t *status, int32_t *first_failure, int32_t step) {
  if (!h || !m) return fail(h, PINKHIP_E_INVALID, "null handle / model");
  if (B < 0 || step < 0 || step >= (1 << 23)) return fail(h, PINKHIP_E_INVALID, "bad B / step");
  if (B == 0) return PINKHIP_OK;
  if (!q || !dq || !status) return fail(h, PINKHIP_E_INVALID, "null pointer");
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip::IntegrateArgs a{m->dev, B, q, dq, status, first_failure, step};
  const long long n = B * m->dev.nj;
  hipLaunchKernelGGL(pinkhip::ik_integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
  PH_HIP(h, hipGetLastError());
  return PINKHIP_OK;
}

// ---- RCCL, loaded on demand so that single-GPU users never need it ------------------------
namespace {
struct Rccl {
  void *so = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, pinkhip_unique_id_t, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*Gather)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl &rccl() {
  static Rccl r;
  return r;
}
int rccl_load(pinkhip_handle *h) {
  Rccl &r = rccl();
  if (r.so) return PINKHIP_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names)
    if ((r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!r.so) return fail(h, PINKHIP_E_COMM, std::string("cannot load librccl: ") + dlerror());
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
  r.Gather = reinterpret_cast<decltype(r.Gather)>(dlsym(r.so, "ncclGather"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Gather || !r.AllGather || !r.GetErrorString) {
    r.so = nullptr;
    return fail(h, PINKHIP_E_COMM, "librccl lacks ncclGather / ncclCommInitRank");
  }
  return PINKHIP_OK;
}
int rccl_fail(pinkhip_handle *h, const char *what, int rc) {
  return fail(h, PINKHIP_E_COMM, std::string(what) + ": " + rccl().GetErrorString(rc));
}
}  // namespace

int pinkhip_comm_get_unique_id(char *id) {
  if (!id) return fail(nullptr, PINKHIP_E_INVALID, "null id");
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  pinkhip_unique_id_t uid;
  if ((rc = rccl().GetUniqueId(&uid))) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
  std::memcpy(id, uid.internal, PINKHIP_COMM_ID_BYTES);
  return PINKHIP_OK;
}

int pinkhip_comm_init(pinkhip_handle *h, const char *id, int rank, int nranks) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (h->comm) return fail(h, PINKHIP_E_INVALID, "communicator already initialised");
  int rc = rccl_load(h);
  if (rc) return rc;
  PH_HIP(h, hipSetDevice(h->device));
  pinkhip_unique_id_t uid;
  std::memcpy(uid.internal, id, PINKHIP_COMM_ID_BYTES);
  if ((rc = rccl().CommInitRank(&h->comm, nranks, uid, rank))) {
    h->comm = nullptr;
    return rccl_fail(h, "ncclCommInitRank", rc);
  }
  h->comm_rank = rank;
  h->comm_size = nranks;
  return PINKHIP_OK;
}

int pinkhip_comm_gather(pinkhip_handle *h, const double *d_send, double *d_recv, int64_t count, int root) {
  if (!h || !h->comm) return fail(h, PINKHIP_E_INVALID, "communicator not initialised");
  if (count < 0 || root < 0 || root >= h->comm_size || !d_send || (h->comm_rank == root && !d_recv))
    return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (count == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  const int rc = rccl().Gather(d_send, d_recv, static_cast<size_t>(count), 8 /* ncclDouble */, root, h->comm, h->stream);
  if (rc) return rccl_fail(h, "ncclGather", rc);
  return PINKHIP_OK;
}

int pinkhip_comm_gather_bytes(pinkhip_handle *h, const void *d_send, void *d_recv, int64_t nbytes, int root) {
  if (!h || !h->comm) return fail(h, PINKHIP_E_INVALID, "communicator not initialised");
  if (nbytes < 0 || root < 0 || root >= h->comm_size || !d_send || (h->comm_rank == root && !d_recv))
    return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (nbytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  const int rc = rccl().Gather(d_send, d_recv, static_cast<size_t>(nbytes), 0 /* ncclInt8 */, root, h->comm, h->stream);
  if (rc) return rccl_fail(h, "ncclGather", rc);
  return PINKHIP_OK;
}

int pinkhip_comm_allgather_bytes(pinkhip_handle *h, const void *d_send, void *d_recv, int64_t nbytes) {
  if (!h || !h->comm) return fail(h, PINKHIP_E_INVALID, "communicator not initialised");
  if (nbytes < 0 || !d_send || !d_recv) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (nbytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  const int rc = rccl().AllGather(d_send, d_recv, static_cast<size_t>(nbytes), 0 /* ncclInt8 */, h->comm, h->stream);
  if (rc) return rccl_fail(h, "ncclAllGather", rc);
  return PINKHIP_OK;
}

int pinkhip_comm_destroy(pinkhip_handle *h) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (h->comm) {
    (void)hipStreamSynchronize(h->stream);
    rccl().CommDestroy(h->comm);
    h->comm = nullptr;
  }
  return PINKHIP_OK;
}

int pinkhip_host_alloc(pinkhip_handle *h, void **hptr, int64_t bytes) {
  if (!h || !hptr || bytes < 0) return fail(h, PINKHIP_E_INVALID, "bad argument");
  *hptr = nullptr;
  PH_HIP(h, hipSetDevice(h->device));
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipHostMalloc(hptr, static_cast<size_t>(bytes), hipHostMallocDefault));
  return PINKHIP_OK;
}

int pinkhip_host_free(pinkhip_handle *h, void *hptr) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (!hptr) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  // (kernels of either compute stream may still be writing results straight into this block)
  PH_HIP(h, hipStreamSynchronize(h->stream));
  if (h->alt_stream) {
    PH_HIP(h, hipStreamSynchronize(h->main_stream));
    PH_HIP(h, hipStreamSynchronize(h->alt_stream));
  }
  PH_HIP(h, hipHostFree(hptr));
  return PINKHIP_OK;
}

int pinkhip_malloc(pinkhip_handle *h, void **dptr, int64_t bytes) {
  if (!h || !dptr || bytes < 0) return fail(h, PINKHIP_E_INVALID, "bad argument");
  *dptr = nullptr;
  PH_HIP(h, hipSetDevice(h->device));
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipMalloc(dptr, static_cast<size_t>(bytes)));
  return PINKHIP_OK;
}

int pinkhip_free(pinkhip_handle *h, void *dptr) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  if (!dptr) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  PH_HIP(h, hipFree(dptr));
  return PINKHIP_OK;
}

int pinkhip_memcpy_h2d(pinkhip_handle *h, void *dst, const void *src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyHostToDevice, h->stream));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  return PINKHIP_OK;
}

int pinkhip_memcpy_h2d_overlapped(pinkhip_handle *h, void *dst, const void *src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyHostToDevice, h->copy_stream));
  PH_HIP(h, hipStreamSynchronize(h->copy_stream));
  return PINKHIP_OK;
}

int pinkhip_memcpy_d2h(pinkhip_handle *h, void *dst, const void *src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyDeviceToHost, h->stream));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  return PINKHIP_OK;
}

int pinkhip_memcpy_d2d(pinkhip_handle *h, void *dst, const void *src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyDeviceToDevice, h->stream));
  return PINKHIP_OK;
}

int pinkhip_sync(pinkhip_handle *h) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipStreamSynchronize(h->copy_stream));
  PH_HIP(h, hipStreamSynchronize(h->stream));
  if (h->alt_stream) {  // (both compute streams, whichever is selected)
    PH_HIP(h, hipStreamSynchronize(h->main_stream));
    PH_HIP(h, hipStreamSynchronize(h->alt_stream));
  }
  PH_HIP(h, hipStreamSynchronize(h->d2h_stream));
  return PINKHIP_OK;
}

int pinkhip_select_compute_stream(pinkhip_handle *h, int32_t index) {
  if (!h || index < 0 || index > 1) return fail(h, PINKHIP_E_INVALID, "compute stream 0 or 1");
  PH_HIP(h, hipSetDevice(h->device));
  if (!h->main_stream) h->main_stream = h->stream;
  if (index == 1 && !h->alt_stream) PH_HIP(h, hipStreamCreateWithFlags(&h->alt_stream, hipStreamNonBlocking));
  h->stream = index ? h->alt_stream : h->main_stream;
  return PINKHIP_OK;
}

int pinkhip_memcpy_h2d_async(pinkhip_handle *h, void *dst, const void *src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyHostToDevice, h->copy_stream));
  return PINKHIP_OK;
}

int pinkhip_stream_wait_copies(pinkhip_handle *h) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipEventRecord(h->ev_copy, h->copy_stream));
  PH_HIP(h, hipStreamWaitEvent(h->stream, h->ev_copy, 0));
  return PINKHIP_OK;
}

int pinkhip_memcpy_d2h_async(pinkhip_handle *h, void *dst, const void *src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, PINKHIP_E_INVALID, "bad argument");
  if (bytes == 0) return PINKHIP_OK;
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipEventRecord(h->ev_kernels, h->stream));
  PH_HIP(h, hipStreamWaitEvent(h->d2h_stream, h->ev_kernels, 0));
  PH_HIP(h, hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyDeviceToHost, h->d2h_stream));
  return PINKHIP_OK;
}

int pinkhip_timer_start(pinkhip_handle *h) {
  if (!h) return fail(nullptr, PINKHIP_E_INVALID, "null handle");
  PH_HIP(h, hipSetDevice(h->device));
  PH_HIP(h, hipEventRecord(h->ev0, h->stream));
  // (a pipelined call alternates between the two compute streams: what the other one runs from here on is inside the
  // bracket too)
  if (h->alt_stream) PH_HIP(h, hipStreamWaitEvent(h->stream == h->main_stream ? h->alt_stream : h->main_stream, h->ev0, 0));
  return PINKHIP_OK;
}


int pinkhip_timer_stop(pinkhip_handle *h, float *elapsed_ms) {
  if (!h || !elapsed_ms) return fail(h, PINKHIP_E_INVALID, "bad argument");
  PH_HIP(h, hipSetDevice(h->device));
  if (h->alt_stream) {  // the selected stream waits for what the other one was given
    PH_HIP(h, hipEventRecord(h->ev_join, h->stream == h->main_stream ? h->alt_stream : h->main_stream));
    PH_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  }
  PH_HIP(h, hipEventRecord(h->ev1, h->stream));
  PH_HIP(h, hipEventSynchronize(h->ev1));
  PH_HIP(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
  return PINKHIP_OK;
}

}  // extern "C"
