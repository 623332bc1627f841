// Wave-level primitives for gfx950 (CDNA4): one 64-lane wavefront per workgroup.
//
// The IK kernels run one QP per wavefront and launch 64-thread workgroups, so
// "lane" == threadIdx.x and a workgroup barrier is just an LDS/VMEM wait for the
// single wave (the compiler lowers __syncthreads() to s_waitcnt for a
// one-wave workgroup).  Cross-lane traffic uses v_readlane for broadcasts from a
// wave-uniform source lane and DPP/ds_bpermute shuffles (via __shfl_*) for
// butterflies.
#pragma once
#include <hip/hip_runtime.h>

// Occupancy hints for the register allocator (waves per SIMD = 512 / VGPRs per lane).
// One QP per wave: 4 waves (128 VGPRs) up to NV = 32, 2 waves for the larger register-resident
// rows.  Packed kernel: 4 waves for NV <= 16, 3 for NV = 24 / 32 (LDS allows ~3 anyway).
#ifndef PINKHIP_WAVES_SMALL
#define PINKHIP_WAVES_SMALL 4
#endif
#ifndef PINKHIP_WAVES_LARGE
#define PINKHIP_WAVES_LARGE 2
#endif
#ifndef PINKHIP_WAVES_PACKED_SMALL
#define PINKHIP_WAVES_PACKED_SMALL 4
#endif
#ifndef PINKHIP_WAVES_PACKED_LARGE
#define PINKHIP_WAVES_PACKED_LARGE 3
#endif
#define PINKHIP_OCCUPANCY_ATTR(NV) \
  __attribute__((amdgpu_waves_per_eu((NV) <= 32 ? PINKHIP_WAVES_SMALL : PINKHIP_WAVES_LARGE, \
                                     (NV) <= 32 ? PINKHIP_WAVES_SMALL : PINKHIP_WAVES_LARGE)))
// (NV = 34, the 27-joint floating-base humanoids: 168 VGPRs with two spilled dwords at three waves -- 4.20 -> 3.36 ms
// per 65 536.  NV = 40 spills 228 dwords at three waves, most of them in the start-up: the box-only instantiation
// still gains, 5.49 -> 4.81 ms, the one with dense rows loses, 3.39 -> 3.55 ms, and stays at two; NV = 50 at three
// waves spills ~1000 dwords and takes twice the time.)
#define PINKHIP_PACKED_WAVES(NV) \
  ((NV) <= 16 ? PINKHIP_WAVES_PACKED_SMALL : (NV) <= 34 ? PINKHIP_WAVES_PACKED_LARGE : PINKHIP_WAVES_LARGE)
// (the box-only NV = 24 fits four waves with 33 spilled dwords: 1.021 -> 0.970 ms; with dense rows 0.845 -> 1.030 ms)
#define PINKHIP_PACKED_WAVES2(NV, DENSE)                         \
  (((NV) == 40 && !(DENSE)) ? PINKHIP_WAVES_PACKED_LARGE         \
   : ((NV) == 24 && !(DENSE)) ? PINKHIP_WAVES_PACKED_LARGE + 1   \
                              : PINKHIP_PACKED_WAVES(NV))
#define PINKHIP_OCCUPANCY_PACKED(NV, DENSE) \
  __attribute__((amdgpu_waves_per_eu(PINKHIP_PACKED_WAVES2(NV, DENSE), PINKHIP_PACKED_WAVES2(NV, DENSE))))

// sweep-tableau kernel (ik_sweep.h): NT = NV + MD doubles of tableau per lane + ~20 doubles of state, no LDS
#ifndef PINKHIP_SWEEP_WAVES
// (the kernel is VALU-throughput bound from three waves per SIMD on -- rocprofv3: the waves of a SIMD issue VALU
// instructions ~100 % of the time -- so the budgets are the largest occupancy without spills: NT = 30 at four waves
// spills 81 registers and takes 0.77 ms per 65 536 against 0.70 ms at three)
#ifndef PINKHIP_SWEEP_WAVES_MID  // (16 < NT <= 34)
#define PINKHIP_SWEEP_WAVES_MID 3
#endif
#define PINKHIP_SWEEP_WAVES(NT) ((NT) <= 16 ? 4 : (NT) <= 34 ? PINKHIP_SWEEP_WAVES_MID : 2)
#endif
#define PINKHIP_OCCUPANCY_SWEEP(NT) __attribute__((amdgpu_waves_per_eu(PINKHIP_SWEEP_WAVES(NT), PINKHIP_SWEEP_WAVES(NT))))
// ... with front coordinates eliminated (NV > W: W tableau rows + what the elimination carries through the stacking)
#ifndef PINKHIP_SWEEP_ELIM_WAVES
#define PINKHIP_SWEEP_ELIM_WAVES 2  // (three: 60 spilled registers, 0.64 ms per 65 536 at nv = 33; two: none, 0.48 ms)
#endif
#define PINKHIP_OCCUPANCY_SWEEP3(NV, MD, W)                                                                             \
  __attribute__((amdgpu_waves_per_eu(((NV) > (W) ? PINKHIP_SWEEP_ELIM_WAVES : PINKHIP_SWEEP_WAVES((NV) + (MD))),     \
                                     ((NV) > (W) ? PINKHIP_SWEEP_ELIM_WAVES : PINKHIP_SWEEP_WAVES((NV) + (MD))))))
// ... with virtual dense rows (ik_sweepx.h): a lane holds NV + 2 MD doubles of tableau (its row + row d of the
// dense-dense block)
#ifndef PINKHIP_SWEEPX_WAVES
#define PINKHIP_SWEEPX_WAVES(NV, MD) ((NV) + 2 * (MD) <= 16 ? 4 : (NV) + 2 * (MD) <= 34 ? 3 : 2)
#endif
#define PINKHIP_OCCUPANCY_SWEEPX(NV, MD) \
  __attribute__((amdgpu_waves_per_eu(PINKHIP_SWEEPX_WAVES(NV, MD), PINKHIP_SWEEPX_WAVES(NV, MD))))

// whole-control-step kernel: the kinematics part needs more registers than the solve of the small sizes (12-dof
// arm, 65 536 robots: 0.136 ms per step at four waves per SIMD with 57 spilled registers, 0.120 ms at three)
#ifndef PINKHIP_ROLLOUT_WAVES_SMALL
#define PINKHIP_ROLLOUT_WAVES_SMALL 3
#endif
#define PINKHIP_ROLLOUT_WAVES(NV) ((NV) <= 16 ? PINKHIP_ROLLOUT_WAVES_SMALL : PINKHIP_PACKED_WAVES(NV))
#define PINKHIP_OCCUPANCY_ROLLOUT(NV) \
  __attribute__((amdgpu_waves_per_eu(PINKHIP_ROLLOUT_WAVES(NV), PINKHIP_ROLLOUT_WAVES(NV))))

// fused forward-kinematics kernels: latency bound (dependent table / LDS round trips), three waves per SIMD
// measured best (0.576 -> 0.548 ms per nv = 30 step; four waves spill)
#ifndef PINKHIP_FK_WAVES
#define PINKHIP_FK_WAVES 3
#endif
#define PINKHIP_OCCUPANCY_FK __attribute__((amdgpu_waves_per_eu(PINKHIP_FK_WAVES, PINKHIP_FK_WAVES)))

#ifndef PINKHIP_SMALL_STACK_WAVES
#define PINKHIP_SMALL_STACK_WAVES 3
#endif
#define PINKHIP_OCCUPANCY_SMALL_STACK \
  __attribute__((amdgpu_waves_per_eu(PINKHIP_SMALL_STACK_WAVES, PINKHIP_SMALL_STACK_WAVES)))

namespace pinkhip {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x); }
__device__ __forceinline__ long long block_id() { return static_cast<long long>(blockIdx.x); }
// Hand-off through LDS between the lanes of the ONE wavefront of the workgroup.  __syncthreads() would be
// correct but also waits for every outstanding global load and store (s_waitcnt vmcnt(0)): it stalls on the
// prefetches that are deliberately in flight and on the kernels' output stores (the kinematics step kernel spent
// 62 % of its wave time there, rocprofv3 SQ_WAIT_ANY).  LDS operations of one wave execute in order, so all that
// is needed is that the compiler does not move LDS accesses across this point: wavefront-scope fences around a
// wave barrier (which emits no instruction).
__device__ __forceinline__ void wave_sync() {
#ifdef PINKHIP_HEAVY_SYNC
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Base of the workgroup's dynamic LDS allocation (16-byte aligned, no static
// __shared__ in front of it: cdna guide, guideline 17).
__device__ __forceinline__ double *shared_base() {
  extern __shared__ __attribute__((aligned(16))) double pinkhip_lds[];
  return pinkhip_lds;
}

// The kernel's argument struct (the ONLY kernel parameter, passed by value: offset 0 of the kernarg segment) behind a
// pointer the compiler cannot see through: fields read through it are loaded where they are used instead of living
// in scalar registers from the start of the kernel.
template <class Args>
__device__ __forceinline__ const Args *kernarg_reload(const Args &a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const Args *p = reinterpret_cast<const Args *>(__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(p));
  return p;
#else  // the host pass of hipcc only parses device functions
  return &a;
#endif
}

// Scheduling fence: keeps hipcc from hoisting the LDS loads of later unrolled
// iterations above this point (bounds VGPR pressure in the fully unrolled
// triangular loops).  Emits no instruction.
__device__ __forceinline__ void sched_fence() {
  asm volatile("" ::: "memory");  // orders the loads at IR / DAG level
  __builtin_amdgcn_sched_barrier(0);  // and in the machine scheduler
}

// Pin a value to the program point: an empty asm that "modifies" x, so the
// arithmetic producing x cannot be sunk below it and later loads cannot be
// hoisted above it.  Without it hipcc issues all NV^2/2 uniform LDS loads of a
// fully unrolled triangular loop first and keeps them live (1000+ VGPRs).
template <typename T>
__device__ __forceinline__ void pin(T &x) {
  asm volatile("" : "+v"(x) : : "memory");
}

// A value the compiler cannot see through (no memory clobber): keeps the two arms of a scalar branch from being merged
// into a select.  Emits no instruction.
__device__ __forceinline__ void opaque(double &x) { asm volatile("" : "+v"(x)); }

// ... and a wave-uniform int (kept in a scalar register)
__device__ __forceinline__ int opaque_uniform(int x) {
  asm volatile("" : "+s"(x));
  return x;
}

// Pin two groups of eight values at once: the sixteen loads that produce them are all issued
// before this point and waited for once (hipcc otherwise issues the LDS reads of an accumulation
// chain pairwise, right before their use, and every pair pays the full LDS latency).
__device__ __forceinline__ void pin16(const double (&a)[8], const double (&b)[8]) {
  asm volatile(""
               :
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]),
                 "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
               : "memory");
}

__device__ __forceinline__ void pin8(const double (&a)[4], const double (&b)[4]) {
  asm volatile("" : : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "memory");
}

// Broadcast `v` of lane `src` (wave-uniform) to every lane: 2 x v_readlane_b32.
__device__ __forceinline__ double bcast(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// ---- DPP cross-lane moves (VALU, ~8 cycles; __shfl_xor would be ds_bpermute) ----
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // every lane has a source under these controls: with bound_ctrl set the old value is dead and the
  // compiler emits the bare v_mov_b32_dpp instead of copy + dpp
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// v_min_f64 without the canonicalising v_max_f64 x, x that fmin() gets in IEEE mode (operands here
// are never signalling NaNs)
__device__ __forceinline__ double min_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float min_raw32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7-i within 8
constexpr int kDppMirror = 0x140;      // lane i <-> 15-i within a row of 16

// Row r+1 receives lane 15 of row r (rows 1 and 3 only), others get `ident`.
__device__ __forceinline__ double dpp_row_bcast15(double v, double ident) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(ident), lo, 0x142, 0xA, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), hi, 0x142, 0xA, 0xF, false);
  return __hiloint2double(hi, lo);
}
// Rows 2 and 3 receive lane 31, others get `ident`.
__device__ __forceinline__ double dpp_row_bcast31(double v, double ident) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(ident), lo, 0x143, 0xC, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), hi, 0x143, 0xC, 0xF, false);
  return __hiloint2double(hi, lo);
}

// Sum over the wave, result in every lane: 4 DPP butterfly steps inside each row
// of 16 lanes (every lane of a row then holds the row sum), 2 DPP row-broadcast
// steps that accumulate the rows into lane 63, one v_readlane pair to publish.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov<kDppXor1>(v);
  v += dpp_mov<kDppXor2>(v);
  v += dpp_mov<kDppHalfMirror>(v);
  v += dpp_mov<kDppMirror>(v);
  v += dpp_row_bcast15(v, 0.0);
  v += dpp_row_bcast31(v, 0.0);
  return bcast(v, 63);
}

__device__ __forceinline__ double wave_min(double v) {
  v = fmin(v, dpp_mov<kDppXor1>(v));
  v = fmin(v, dpp_mov<kDppXor2>(v));
  v = fmin(v, dpp_mov<kDppHalfMirror>(v));
  v = fmin(v, dpp_mov<kDppMirror>(v));
  v = fmin(v, dpp_row_bcast15(v, INFINITY));
  v = fmin(v, dpp_row_bcast31(v, INFINITY));
  return bcast(v, 63);
}

// Argmin with an 8-bit payload: the payload replaces the 8 low mantissa bits of
// the key (a 2^-44 relative perturbation, only ever used to *choose* a lane; the
// exact value is fetched from the winner afterwards).  v must be finite.
__device__ __forceinline__ double key_pack(double v, int payload) {
  const long long b = (__double_as_longlong(v) & ~0xFFLL) | (long long)(payload & 0xFF);
  return __longlong_as_double(b);
}
__device__ __forceinline__ int key_payload(double k) { return (int)(__double_as_longlong(k) & 0xFF); }

// 1/x and 1/sqrt(x) for normal positive x: hardware seed + two Newton steps
// (~1 ulp), without the IEEE division / sqrt fix-up sequences.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-(x * y), y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-(x * y), y, 1.0);
  return fma(0.5 * y, e, y);
}

// sin and cos of a joint angle: Cody-Waite reduction by pi/2 (two FMAs, exact to ~1e-32 |k|) + the fdlibm
// minimax kernels on [-pi/4, pi/4]; ~1 ulp for |t| < 1e5 rad.  The libm versions carry a Payne-Hanek path for
// huge arguments that costs private-memory scratch and registers in every kernel that calls them.
__device__ __forceinline__ void fast_sincos(double t, double &sn, double &cs) {
  const double k = rint(t * 0.63661977236758134308);  // 2 / pi
  double r = fma(-k, 1.5707963267948966, t);
  r = fma(-k, 6.123233995736766e-17, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double s0 = fma(r * z, ps, r);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = static_cast<int>(k) & 3;
  const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -s1 : s1;
  cs = ((q + 1) & 2) ? -c1 : c1;
}

// One Newton step (~2e-14 relative): for quantities that only steer the active-set iteration or are
// re-orthogonalised anyway; the Cholesky pivots keep the two-step version.
__device__ __forceinline__ double fast_rcp1(double x) {
  double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double fast_rsqrt1(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(0.5 * y, e, y);
}

// hardware reciprocal seed only (~2^-26 relative... v_rcp_f64 gives ~1e-8): for keys that merely rank candidates
__device__ __forceinline__ double approx_rcp(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ float approx_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- sub-wave groups: W lanes per QP, 64/W QPs per wavefront (W = 8, 16, 32) ----
__device__ __forceinline__ bool wave_any(bool p) { return __any(p ? 1 : 0) != 0; }
// `if (lanes_on(p)) { body }` for a predicate that is uniform over each group of lanes that exchange data in `body`:
// the lanes of the other groups are switched off (EXEC) for the body instead of carrying selects through it.  The body
// must ALSO be written so that a lane with p false changes nothing when it runs it (selects on p): that is what the
// CPU wave emulator executes -- its cross-lane primitives are rendezvous of all 64 lanes -- and here the compiler
// folds those selects away inside the branch.
#ifdef PINKHIP_LANES_ON_SELECT  // (development: the body masks itself, as on the emulator)
__device__ __forceinline__ bool lanes_on(bool) { return true; }
#else
__device__ __forceinline__ bool lanes_on(bool p) { return p; }
#endif
// bit l = the predicate of lane l (a scalar)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// A value every lane of the wave agrees on, as a scalar the compiler keeps in an SGPR and branches on with s_cbranch_scc.
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Index inside its group of W lanes of the first lane whose predicate holds (W if none): one ballot, the group's bits
// shifted down, count of trailing zeros -- no LDS crossbar.
template <int W>
__device__ __forceinline__ int group_first_lane(bool p) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
  if constexpr (W == 64) {
    return m ? __builtin_ctzll(m) : 64;
  } else {
    const int lane = static_cast<int>(threadIdx.x);
    const unsigned half = (lane & 32) ? static_cast<unsigned>(m >> 32) : static_cast<unsigned>(m);
    unsigned g = half;
    if constexpr (W < 32) g = (half >> (lane & 31 & ~(W - 1))) & ((1u << W) - 1u);
    return g ? __builtin_ctz(g) : W;
  }
}

// Value of an arbitrary (per-lane) source lane: 2 x ds_bpermute_b32 (LDS crossbar, no LDS memory).
__device__ __forceinline__ double lane_shfl(double v, int src_lane) {
  const int a = src_lane << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int lane_shfl_i(int v, int src_lane) {
  return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}

// Broadcast inside each group of W lanes from the group's lane `src` (group-uniform, may
// differ between groups).
template <int W>
__device__ __forceinline__ double group_bcast(double v, int src) {
  return lane_shfl(v, (lane_id() & ~(W - 1)) | src);
}
template <int W>
__device__ __forceinline__ int group_bcast_i(int v, int src) {
  return lane_shfl_i(v, (lane_id() & ~(W - 1)) | src);
}
// The two rows of 16 lanes of every 32-lane half exchange their values on the VALU (gfx950
// v_permlane16_swap_b32, no LDS-crossbar round trip): a and b hold, in both rows, the value of the
// even and of the odd row.
__device__ __forceinline__ void row_pair(double v, double &a, double &b) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  a = __hiloint2double(rh[0], rl[0]);
  b = __hiloint2double(rh[1], rl[1]);
}
// Same between the two halves of the wave (v_permlane32_swap_b32): a = value of the lower 32 lanes, b = value
// of the upper 32 lanes, in both halves.
__device__ __forceinline__ void half_pair(double v, double &a, double &b) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double(rh[0], rl[0]);
  b = __hiloint2double(rh[1], rl[1]);
}

// ---- broadcast-FMA: acc + (value held by lane J of the group) * x in ONE VALU instruction ----
// gfx90a+ allow a DPP operand on the double-precision VOP2 v_fmac_f64 with the row_newbcast controls
// (lane K of every row of 16 lanes feeds the whole row; measured on MI355X at the rate of a plain
// v_fmac_f64, scripts/probes/dpp64_probe.hip).  A group-uniform vector whose entry j lives in lane j --
// the Householder vector, a Cholesky column, a staged task row -- therefore multiplies into lane-local
// accumulators without going through LDS at all: no ds_write, no barrier, no broadcast ds_read.
// Bcast<W> holds one copy of the lane-held value per row of the group (W = 32: the value of the even and of
// the odd row, W = 64: of the four rows), made by bcast_prepare() with permlane swaps.
template <int W>
struct Bcast {
  static_assert(W == 16 || W == 32 || W == 64, "broadcast-FMA groups are whole rows of 16 lanes");
  double r[W / 16];
};
template <int W>
__device__ __forceinline__ Bcast<W> bcast_prepare(double v) {
  Bcast<W> b;
  if constexpr (W == 16) {
    b.r[0] = v;
    asm volatile("s_nop 1" : "+v"(b.r[0]));  // VALU write -> DPP read: two wait states, unknown to the compiler
  } else if constexpr (W == 32) {
    row_pair(v, b.r[0], b.r[1]);
    asm volatile("s_nop 1" : "+v"(b.r[0]), "+v"(b.r[1]));
  } else {
    double e, o;
    row_pair(v, e, o);
    half_pair(e, b.r[0], b.r[2]);
    half_pair(o, b.r[1], b.r[3]);
    asm volatile("s_nop 1" : "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]));
  }
  return b;
}
// bcast_prepare(v * s) for a group-uniform s from bcast_prepare(v): one multiplication per row copy instead of
// the permlane swaps (the copies of a group's rows all belong to that group)
template <int W>
__device__ __forceinline__ Bcast<W> bcast_scale(const Bcast<W> &b, double s) {
  Bcast<W> o;
#pragma unroll
  for (int k = 0; k < W / 16; ++k) o.r[k] = b.r[k] * s;
  if constexpr (W == 16) asm volatile("s_nop 1" : "+v"(o.r[0]));
  else if constexpr (W == 32) asm volatile("s_nop 1" : "+v"(o.r[0]), "+v"(o.r[1]));
  else asm volatile("s_nop 1" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]), "+v"(o.r[3]));
  return o;
}
// bcast_prepare(li == src ? 1.0 : 0.0) for a group-uniform src without the permlane swaps: row copy k of a lane is
// the value of the group's lane 16 k + (lane % 16), which every lane can tell from src alone (src < 0: all zero)
template <int W>
__device__ __forceinline__ Bcast<W> bcast_indicator(int src) {
  Bcast<W> b;
  const int l16 = lane_id() & 15;
#pragma unroll
  for (int k = 0; k < W / 16; ++k) b.r[k] = (l16 + 16 * k == src) ? 1.0 : 0.0;
  if constexpr (W == 16) asm volatile("s_nop 1" : "+v"(b.r[0]));
  else if constexpr (W == 32) asm volatile("s_nop 1" : "+v"(b.r[0]), "+v"(b.r[1]));
  else asm volatile("s_nop 1" : "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]));
  return b;
}
// per-lane choice between two prepared broadcasts (the condition is uniform over each group)
template <int W>
__device__ __forceinline__ Bcast<W> bcast_select(bool c, const Bcast<W> &a, const Bcast<W> &b) {
  Bcast<W> o;
#pragma unroll
  for (int k = 0; k < W / 16; ++k) o.r[k] = c ? a.r[k] : b.r[k];
  if constexpr (W == 16) asm volatile("s_nop 1" : "+v"(o.r[0]));
  else if constexpr (W == 32) asm volatile("s_nop 1" : "+v"(o.r[0]), "+v"(o.r[1]));
  else asm volatile("s_nop 1" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]), "+v"(o.r[3]));
  return o;
}
template <int W, int J>
__device__ __forceinline__ double fma_bcast(double acc, const Bcast<W> &b, double x) {
  static_assert(J >= 0 && J < W, "source lane outside the group");
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc)
      : "v"(b.r[J / 16]), "v"(x), "n"(J % 16));
  return acc;
}
// the value itself, group-uniform
// (v_mov_b64 is the one other double-precision instruction that takes the row_newbcast operand: one instruction instead
// of 0 + value * 1 with its two constants)
template <int W, int J>
__device__ __forceinline__ double value_bcast(const Bcast<W> &b) {
  static_assert(J >= 0 && J < W, "source lane outside the group");
  double r;
  asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b.r[J / 16]), "n"(J % 16));
  return r;
}

// All-reduce inside each group of W lanes (every lane gets its group's result).
template <int W>
__device__ __forceinline__ double group_sum(double v) {
  v += dpp_mov<kDppXor1>(v);
  v += dpp_mov<kDppXor2>(v);
  v += dpp_mov<kDppHalfMirror>(v);
  if (W >= 16) v += dpp_mov<kDppMirror>(v);
  if (W >= 32) {
    double a, b;
    row_pair(v, a, b);
    v = a + b;
  }
  if (W == 64) v = bcast(v, 0) + bcast(v, 32);
  return v;
}
template <int W>
__device__ __forceinline__ double group_min(double v) {
  v = min_raw(v, dpp_mov<kDppXor1>(v));
  v = min_raw(v, dpp_mov<kDppXor2>(v));
  v = min_raw(v, dpp_mov<kDppHalfMirror>(v));
  if (W >= 16) v = min_raw(v, dpp_mov<kDppMirror>(v));
  if (W >= 32) {
    double a, b;
    row_pair(v, a, b);
    v = min_raw(a, b);
  }
  if (W == 64) v = min_raw(bcast(v, 0), bcast(v, 32));
  return v;
}
// Arg-min over a group on a 32-bit key (float with an 8-bit payload in the low mantissa bits): one
// v_min_f32_dpp per butterfly step.  Only for *choosing* a lane where any candidate is a valid choice (the
// most violated constraint); exact values are fetched from the winner afterwards.
__device__ __forceinline__ float key32_pack(double v, int payload) {
  // keys are negative; clamped into the normal floats so that neither -inf | payload (a NaN that v_min_f32 drops)
  // nor -0.0f (which reads as "none") can come out of the cast
  const float f = fmaxf(fminf(static_cast<float>(v), -1.17549435e-38f), -3.0e38f);
  return __int_as_float((__float_as_int(f) & ~0xFF) | (payload & 0xFF));
}
__device__ __forceinline__ float key32_packf(float v, int payload) {
  const float f = fmaxf(fminf(v, -1.17549435e-38f), -3.0e38f);
  return __int_as_float((__float_as_int(f) & ~0xFF) | (payload & 0xFF));
}
__device__ __forceinline__ int key32_payload(float k) { return __float_as_int(k) & 0xFF; }
#define PINKHIP_MIN32_DPP(ctrl)                                                                          \
  asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v)); \
  v = r;
template <int W>
__device__ __forceinline__ float group_min32(float v) {
  float r;
  PINKHIP_MIN32_DPP("quad_perm:[1,0,3,2]")
  PINKHIP_MIN32_DPP("quad_perm:[2,3,0,1]")
  PINKHIP_MIN32_DPP("row_half_mirror")
  if (W >= 16) { PINKHIP_MIN32_DPP("row_mirror") }
  if (W >= 32) {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = min_raw32(__uint_as_float(p[0]), __uint_as_float(p[1]));
  }
  if (W == 64) v = fminf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0)),
                         __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32)));
  return v;
}
#undef PINKHIP_MIN32_DPP

// Inclusive prefix sum inside each group of W lanes (lane li gets v_0 + ... + v_li): Hillis-Steele with
// DPP row shifts inside the rows of 16, then the row totals through row_bcast15 / row_bcast31.
template <int N>
__device__ __forceinline__ double dpp_row_shr(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xF, 0xF, true);  // lanes without a source get 0
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int W>
__device__ __forceinline__ double group_scan_sum(double v) {
  const int li = lane_id() & (W - 1);
  double t = dpp_row_shr<1>(v);
  v += (li >= 1) ? t : 0.0;
  t = dpp_row_shr<2>(v);
  v += (li >= 2) ? t : 0.0;
  t = dpp_row_shr<4>(v);
  v += (li >= 4) ? t : 0.0;
  if (W >= 16) {
    t = dpp_row_shr<8>(v);
    v += (li >= 8) ? t : 0.0;
  }
  if (W >= 32) v += dpp_row_bcast15(v, 0.0);
  if (W == 64) v += dpp_row_bcast31(v, 0.0);
  return v;
}
// max / min over the groups of a group-uniform int (one v_readlane per group).
template <int W>
__device__ __forceinline__ int groups_max(int v) {
  int m = bcast_i(v, 0);
#pragma unroll
  for (int g = 1; g < kWave / W; ++g) {
    const int o = bcast_i(v, g * W);
    m = o > m ? o : m;
  }
  return m;
}
template <int W>
__device__ __forceinline__ int groups_min(int v) {
  int m = bcast_i(v, 0);
#pragma unroll
  for (int g = 1; g < kWave / W; ++g) {
    const int o = bcast_i(v, g * W);
    m = o < m ? o : m;
  }
  return m;
}

// ---- fp64 matrix core: D(16x16) += A(16x4) B(4x16), one wavefront ----
// Operand layout (cdna_hip_programming.md, fragment layout): lane l supplies A[l & 15][l >> 4]
// and B[l >> 4][l & 15]; it receives D[(l >> 4) + 4 r][l & 15] in element r of the accumulator.
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4d mfma_f64_16x16x4(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Value held by lane+1 (lane 63 receives its own value).
__device__ __forceinline__ double from_next_lane(double v) { return __shfl_down(v, 1, kWave); }
__device__ __forceinline__ int from_next_lane_i(int v) { return __shfl_down(v, 1, kWave); }

}  // namespace pinkhip
