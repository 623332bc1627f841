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

// Occupancy hint for the register allocator (waves per SIMD), see ik_kernels.h.
#define PINKHIP_OCCUPANCY_ATTR(NV) \
  __attribute__((amdgpu_waves_per_eu((NV) <= 32 ? PINKHIP_WAVES_SMALL : PINKHIP_WAVES_LARGE, \
                                     (NV) <= 32 ? PINKHIP_WAVES_SMALL : PINKHIP_WAVES_LARGE)))

namespace pinkhip {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x); }
__device__ __forceinline__ long long block_id() { return static_cast<long long>(blockIdx.x); }
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// Base of the workgroup's dynamic LDS allocation (16-byte aligned, no static
// __shared__ in front of it: cdna guide, guideline 17).
__device__ __forceinline__ double *shared_base() {
  extern __shared__ __attribute__((aligned(16))) double pinkhip_lds[];
  return pinkhip_lds;
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
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
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

// Value held by lane+1 (lane 63 receives its own value).
__device__ __forceinline__ double from_next_lane(double v) { return __shfl_down(v, 1, kWave); }
__device__ __forceinline__ int from_next_lane_i(int v) { return __shfl_down(v, 1, kWave); }

}  // namespace pinkhip
