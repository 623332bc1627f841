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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

// Minimum of `v` over the wave with the payload `idx` of the winning lane
// (ties go to the smaller payload so the result is lane-order independent).
__device__ __forceinline__ void wave_argmin(double &v, int &idx) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    double ov = __shfl_xor(v, m, kWave);
    int oi = __shfl_xor(idx, m, kWave);
    bool take = (ov < v) || (ov == v && oi < idx);
    v = take ? ov : v;
    idx = take ? oi : idx;
  }
}

// Value held by lane+1 (lane 63 receives its own value).
__device__ __forceinline__ double from_next_lane(double v) { return __shfl_down(v, 1, kWave); }
__device__ __forceinline__ int from_next_lane_i(int v) { return __shfl_down(v, 1, kWave); }

}  // namespace pinkhip
