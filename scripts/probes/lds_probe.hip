// LDS throughput probe (not product code): how many cycles does a CU spend per ds_read / ds_write
// as a function of access width, address pattern and number of active lanes?
//   hipcc --offload-arch=gfx950 -O3 lds_probe.hip -o lds_probe && ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 512;

template <int MODE>
__global__ void __launch_bounds__(64) probe(double *out, unsigned long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = i * 0.5;
  __syncthreads();
  double acc0 = 0.0, acc1 = 0.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  // MODE 0: b128 read, wave-uniform address        1: b128 read, lane-consecutive (conflict-free)
  // MODE 2: b128 read, uniform, one active lane    3: b128 read, uniform per half, two active lanes
  // MODE 4: b64 read uniform                       5: b64 read lane-consecutive
  // MODE 6: b128 write one active lane             7: b128 write all lanes consecutive
  // MODE 8: b64 read, uniform per half (2 addresses)  9: b128 read, uniform per half, all lanes
  // MODE 10: b128 read uniform, 8 active lanes (one per 8)   11: b128 read uniform, 4 active lanes (one per row)
#pragma unroll 8
  for (int it = 0; it < kIters; ++it) {
    const int base = (it * 4) & 1023;
    if (MODE == 0) {
      const double2 v = *reinterpret_cast<const double2 *>(lds + base);
      acc0 += v.x; acc1 += v.y;
    } else if (MODE == 1) {
      const double2 v = *reinterpret_cast<const double2 *>(lds + ((base + 2 * lane) & 2046));
      acc0 += v.x; acc1 += v.y;
    } else if (MODE == 2) {
      if (lane == 5) { const double2 v = *reinterpret_cast<const double2 *>(lds + base); acc0 += v.x; acc1 += v.y; }
    } else if (MODE == 3) {
      if ((lane & 31) == 5) { const double2 v = *reinterpret_cast<const double2 *>(lds + base + (lane >> 5) * 64); acc0 += v.x; acc1 += v.y; }
    } else if (MODE == 4) {
      acc0 += lds[base];
    } else if (MODE == 5) {
      acc0 += lds[(base + lane) & 2047];
    } else if (MODE == 6) {
      if (lane == 5) *reinterpret_cast<double2 *>(lds + base) = make_double2(acc0 + it, acc1);
    } else if (MODE == 7) {
      *reinterpret_cast<double2 *>(lds + ((base + 2 * lane) & 2046)) = make_double2(acc0 + it, acc1);
    } else if (MODE == 8) {
      acc0 += lds[base + (lane >> 5) * 64];
    } else if (MODE == 9) {
      const double2 v = *reinterpret_cast<const double2 *>(lds + base + (lane >> 5) * 64);
      acc0 += v.x; acc1 += v.y;
    } else if (MODE == 10) {
      if ((lane & 7) == 0) { const double2 v = *reinterpret_cast<const double2 *>(lds + base); acc0 += v.x; acc1 += v.y; }
    } else if (MODE == 11) {
      if ((lane & 15) == 0) { const double2 v = *reinterpret_cast<const double2 *>(lds + base); acc0 += v.x; acc1 += v.y; }
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = acc0 + acc1 + lds[lane];
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_cu) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * waves_per_cu;
  double *out;
  unsigned long long *cyc;
  hipMalloc(&out, blocks * 64 * sizeof(double));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds_bytes = 160 * 1024 / waves_per_cu > 65536 ? 65536 : (160 * 1024 / waves_per_cu) & ~255;  // force residency
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  probe<MODE><<<blocks, 64, lds_bytes>>>(out, cyc);
  hipEventRecord(e0);
  probe<MODE><<<blocks, 64, lds_bytes>>>(out, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : h) mean += c;
  mean /= blocks;
  // CU-cycles per instruction: all waves of a CU share one LDS pipe
  std::printf("%-44s waves/CU=%2d  wave time %8.0f cyc  -> %6.2f cyc/instr/wave, %6.2f CU-cyc/instr  (%.3f ms)\n", name,
              waves_per_cu, mean, mean / kIters, mean / kIters / waves_per_cu, ms);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w : {1, 12}) {
    run<0>("b128 read, uniform address, 64 lanes", w);
    run<9>("b128 read, uniform per half, 64 lanes", w);
    run<1>("b128 read, lane-consecutive, 64 lanes", w);
    run<2>("b128 read, 1 active lane", w);
    run<3>("b128 read, 2 active lanes", w);
    run<11>("b128 read, 4 active lanes (one per row)", w);
    run<10>("b128 read, 8 active lanes (one per 8)", w);
    run<4>("b64 read, uniform address, 64 lanes", w);
    run<8>("b64 read, uniform per half, 64 lanes", w);
    run<5>("b64 read, lane-consecutive, 64 lanes", w);
    run<6>("b128 write, 1 active lane", w);
    run<7>("b128 write, lane-consecutive, 64 lanes", w);
  }
  return 0;
}
