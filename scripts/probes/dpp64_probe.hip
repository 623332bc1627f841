// Probe (not product code): does gfx950 execute v_fmac_f64 with a DPP row_newbcast operand, and at what rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__device__ __forceinline__ double fma_row_bcast(double acc, double b, double x) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(x), "n"(K));
  return acc;
}

__global__ void check(const double *b, const double *x, double *out) {
  const int l = threadIdx.x;
  double bv = b[l];
  asm volatile("s_nop 1" : "+v"(bv));
  double acc = 1.0;
  acc = fma_row_bcast<3>(acc, bv, x[l]);
  acc = fma_row_bcast<15>(acc, bv, 2.0 * x[l]);
  out[l] = acc;
}

template <bool DPP>
__global__ void rate(const double *b, double *out, unsigned long long *cyc) {
  const int l = threadIdx.x;
  double bv = b[l];
  asm volatile("s_nop 1" : "+v"(bv));
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, x = 1.0 + l;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 1024; ++it) {
    if (DPP) {
      a0 = fma_row_bcast<1>(a0, bv, x); a1 = fma_row_bcast<2>(a1, bv, x); a2 = fma_row_bcast<3>(a2, bv, x); a3 = fma_row_bcast<4>(a3, bv, x);
      a0 = fma_row_bcast<5>(a0, bv, x); a1 = fma_row_bcast<6>(a1, bv, x); a2 = fma_row_bcast<7>(a2, bv, x); a3 = fma_row_bcast<8>(a3, bv, x);
    } else {
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(bv), "v"(x)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(bv), "v"(x));
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a2) : "v"(bv), "v"(x)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a3) : "v"(bv), "v"(x));
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(bv), "v"(x)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(bv), "v"(x));
      asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a2) : "v"(bv), "v"(x)); asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a3) : "v"(bv), "v"(x));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double hb[64], hx[64], ho[64];
  for (int i = 0; i < 64; ++i) { hb[i] = 100.0 + i; hx[i] = 0.5 * i + 1; }
  double *b, *x, *o; unsigned long long *c;
  hipMalloc(&b, 512); hipMalloc(&x, 512); hipMalloc(&o, 512 * 4096); hipMalloc(&c, 8 * 4096);
  hipMemcpy(b, hb, 512, hipMemcpyHostToDevice); hipMemcpy(x, hx, 512, hipMemcpyHostToDevice);
  check<<<1, 64>>>(b, x, o);
  hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int row = l & ~15;
    const double ref = 1.0 + hb[row + 3] * hx[l] + hb[row + 15] * 2.0 * hx[l];
    if (ho[l] != ref) { ++bad; if (bad < 4) std::printf("lane %d: got %.3f want %.3f\n", l, ho[l], ref); }
  }
  std::printf("row_newbcast semantics: %s\n", bad ? "MISMATCH" : "ok (lane K of each row of 16)");
  for (int dpp = 0; dpp < 2; ++dpp) {
    const int blocks = 256 * 8;
    if (dpp) rate<true><<<blocks, 64>>>(b, o, c); else rate<false><<<blocks, 64>>>(b, o, c);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), c, 8 * blocks, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= blocks;
    std::printf("%s: %.1f cycles per 8 fmac (2 waves/SIMD)\n", dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64", m / 1024);
  }
  return 0;
}
