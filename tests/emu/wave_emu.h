// CPU emulation of one 64-lane wavefront -- TEST INFRASTRUCTURE ONLY.
//
// Runs the *unmodified* kernel source (pink_amd/csrc/ik_kernels.h) on the host so
// that the SIMT logic (lane ownership, broadcasts, LDS hand-offs) can be checked
// without a GPU.  Each lane is a ucontext fiber; every cross-lane primitive and
// every wave_sync() is a rendezvous of all 64 fibers (round-robin switch), which
// reproduces lock-step semantics and additionally detects divergence: all lanes
// must arrive at the same primitive the same number of times.
//
// The product never links this; pink_amd's API fails loudly without the HIP
// library (pink_amd/_lib.py).
#pragma once

#include <ucontext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__ inline
#define __forceinline__ inline
#define __launch_bounds__(x)
#define PINKHIP_OCCUPANCY_ATTR(NV)
#define PINKHIP_OCCUPANCY_PACKED(NV, DENSE)
#define PINKHIP_OCCUPANCY_SWEEP(NT)
#define PINKHIP_OCCUPANCY_SWEEP3(NV, MD, W)
#define PINKHIP_OCCUPANCY_SWEEPX(NV, MD)
#define PINKHIP_OCCUPANCY_ROLLOUT(NV)
#define PINKHIP_OCCUPANCY_FK
#define PINKHIP_OCCUPANCY_SMALL_STACK

// element-wise kernels use blockIdx / threadIdx directly; the emulator calls their per-thread
// bodies in a plain loop and only needs the names to exist
struct EmuDim3 {
  unsigned x = 0, y = 0, z = 0;
};
static EmuDim3 blockIdx, threadIdx;

// element-wise kernels: one "thread" at a time in the emulator
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
  const unsigned long long old = *p;
  if (v < old) *p = v;
  return old;
}

namespace pinkhip {

constexpr int kWave = 64;

struct Emu {
  int cur = 0;
  long long block = 0;
  ucontext_t main_ctx;
  ucontext_t ctx[kWave];
  bool done[kWave];
  int live = 0;
  double slot_d[kWave];
  int slot_i[kWave];
  long long arrivals[kWave];
  int tag[kWave];
  alignas(16) double lds[20480];  // 160 KiB
  char *stacks = nullptr;
};

inline Emu &emu() {
  static Emu e;
  return e;
}

inline int lane_id() { return emu().cur; }
inline long long block_id() { return emu().block; }
inline double *shared_base() { return emu().lds; }

// Switch to the next live lane; one full round == one barrier.
inline void emu_rendezvous(int tag) {
  Emu &e = emu();
  const int me = e.cur;
  e.arrivals[me]++;
  e.tag[me] = tag;
  int nxt = me;
  do {
    nxt = (nxt + 1) % kWave;
  } while (e.done[nxt] && nxt != me);
  if (nxt != me) {
    e.cur = nxt;
    swapcontext(&e.ctx[me], &e.ctx[nxt]);
  }
  // everybody has arrived: check lock-step
  for (int l = 0; l < kWave; ++l) {
    if (e.done[l]) continue;
    if (e.arrivals[l] != e.arrivals[me] && e.arrivals[l] != e.arrivals[me] + 1 &&
        e.arrivals[l] + 1 != e.arrivals[me]) {
      std::fprintf(stderr, "wave emulator: divergence at lane %d vs %d (tag %d vs %d)\n", me, l,
                   e.tag[me], e.tag[l]);
      std::abort();
    }
    if (e.arrivals[l] == e.arrivals[me] && e.tag[l] != e.tag[me]) {
      std::fprintf(stderr, "wave emulator: lanes %d and %d at different primitives (%d vs %d)\n", me,
                   l, e.tag[me], e.tag[l]);
      std::abort();
    }
  }
}

inline void wave_sync() { emu_rendezvous(1); }
inline void sched_fence() {}
inline void pin16(const double (&)[8], const double (&)[8]) {}
inline void pin8(const double (&)[4], const double (&)[4]) {}
template <typename T>
inline void pin(T &) {}
inline void opaque(double &) {}
inline int opaque_uniform(int x) { return x; }
template <class Args>
inline const Args *kernarg_reload(const Args &a) {
  return &a;
}

inline double bcast(double v, int src) {
  Emu &e = emu();
  e.slot_d[e.cur] = v;
  emu_rendezvous(2);
  if (src < 0 || src >= kWave) {
    std::fprintf(stderr, "wave emulator: bcast from lane %d\n", src);
    std::abort();
  }
  const double r = e.slot_d[src];
  emu_rendezvous(3);
  return r;
}

inline int bcast_i(int v, int src) {
  Emu &e = emu();
  e.slot_i[e.cur] = v;
  emu_rendezvous(4);
  if (src < 0 || src >= kWave) std::abort();
  const int r = e.slot_i[src];
  emu_rendezvous(5);
  return r;
}

inline double emu_exchange(double v, int partner_of_me, int tag) {
  Emu &e = emu();
  e.slot_d[e.cur] = v;
  emu_rendezvous(tag);
  const double o = e.slot_d[partner_of_me];
  emu_rendezvous(tag + 1);
  return o;
}

// same combination order as the device code (wave.h): xor1, xor2, half-mirror,
// mirror inside each row of 16, then row_bcast15 (rows 1,3 += lane 15 of the row
// before), row_bcast31 (rows 2,3 += lane 31), result read from lane 63
template <class Op>
inline double emu_reduce(double v, double ident, Op op, int tag) {
  const int l = emu().cur;
  v = op(v, emu_exchange(v, l ^ 1, tag));
  v = op(v, emu_exchange(v, l ^ 2, tag));
  v = op(v, emu_exchange(v, (l & ~7) | (7 - (l & 7)), tag));
  v = op(v, emu_exchange(v, (l & ~15) | (15 - (l & 15)), tag));
  const int row = l >> 4;
  double t = emu_exchange(v, (row == 1 || row == 3) ? (row - 1) * 16 + 15 : l, tag);
  v = op(v, (row == 1 || row == 3) ? t : ident);
  t = emu_exchange(v, 31, tag);
  v = op(v, (row >= 2) ? t : ident);
  return bcast(v, 63);
}
inline double wave_sum(double v) {
  return emu_reduce(v, 0.0, [](double a, double b) { return a + b; }, 6);
}
inline double wave_min(double v) {
  return emu_reduce(v, INFINITY, [](double a, double b) { return std::fmin(a, b); }, 8);
}

inline double key_pack(double v, int payload) {
  long long b;
  std::memcpy(&b, &v, 8);
  b = (b & ~0xFFLL) | (long long)(payload & 0xFF);
  std::memcpy(&v, &b, 8);
  return v;
}
inline int key_payload(double k) {
  long long b;
  std::memcpy(&b, &k, 8);
  return (int)(b & 0xFF);
}
inline void fast_sincos(double t, double &sn, double &cs) {
  sn = std::sin(t);
  cs = std::cos(t);
}
inline double fast_rcp(double x) { return 1.0 / x; }
inline double fast_rsqrt(double x) { return 1.0 / std::sqrt(x); }
inline double fast_rcp1(double x) { return 1.0 / x; }
inline double max_raw(double a, double b) { return a > b ? a : b; }
inline double min_raw(double a, double b) { return a < b ? a : b; }
inline int wave_uniform(int v) { return v; }
inline double approx_rcp(double x) { return 1.0 / x; }
inline float approx_rcpf(float x) { return 1.0f / x; }
inline double fast_rsqrt1(double x) { return 1.0 / std::sqrt(x); }

inline bool wave_any(bool p) {
  Emu &e = emu();
  e.slot_i[e.cur] = p ? 1 : 0;
  emu_rendezvous(14);
  bool r = false;
  for (int l = 0; l < kWave; ++l) r = r || e.slot_i[l];
  emu_rendezvous(15);
  return r;
}
// (wave.h: on the device the other lanes are switched off; here every lane runs the body, which masks itself)
inline bool lanes_on(bool) { return true; }
inline unsigned long long wave_ballot(bool p) {
  Emu &e = emu();
  e.slot_i[e.cur] = p ? 1 : 0;
  emu_rendezvous(42);
  unsigned long long r = 0;
  for (int l = 0; l < kWave; ++l)
    if (e.slot_i[l]) r |= 1ull << l;
  emu_rendezvous(43);
  return r;
}
template <int W>
inline int group_first_lane(bool p) {
  Emu &e = emu();
  e.slot_i[e.cur] = p ? 1 : 0;
  emu_rendezvous(40);
  int r = W;
  const int base = e.cur & ~(W - 1);
  for (int l = W - 1; l >= 0; --l)
    if (e.slot_i[base + l]) r = l;
  emu_rendezvous(41);
  return r;
}
inline double lane_shfl(double v, int src_lane) {
  if (src_lane < 0 || src_lane >= kWave) {
    std::fprintf(stderr, "wave emulator: lane_shfl from lane %d\n", src_lane);
    std::abort();
  }
  return emu_exchange(v, src_lane, 16);
}
inline int lane_shfl_i(int v, int src_lane) {
  Emu &e = emu();
  if (src_lane < 0 || src_lane >= kWave) std::abort();
  e.slot_i[e.cur] = v;
  emu_rendezvous(18);
  const int r = e.slot_i[src_lane];
  emu_rendezvous(19);
  return r;
}
template <int W>
inline double group_bcast(double v, int src) {
  if (src < 0 || src >= W) {
    std::fprintf(stderr, "wave emulator: group_bcast<%d> from %d\n", W, src);
    std::abort();
  }
  return lane_shfl(v, (emu().cur & ~(W - 1)) | src);
}
template <int W>
inline int group_bcast_i(int v, int src) {
  if (src < 0 || src >= W) std::abort();
  return lane_shfl_i(v, (emu().cur & ~(W - 1)) | src);
}
// broadcast-FMA (wave.h: v_fmac_f64 with a DPP row_newbcast operand): here one rendezvous per use
template <int W>
struct Bcast {
  double v;
};
template <int W>
inline Bcast<W> bcast_prepare(double v) {
  return Bcast<W>{v};
}
template <int W>
inline Bcast<W> bcast_scale(const Bcast<W> &b, double s) {
  return Bcast<W>{b.v * s};
}
template <int W>
inline Bcast<W> bcast_indicator(int src) {
  return Bcast<W>{((emu().cur & (W - 1)) == src) ? 1.0 : 0.0};
}
template <int W>
inline Bcast<W> bcast_select(bool c, const Bcast<W> &a, const Bcast<W> &b) {
  return Bcast<W>{c ? a.v : b.v};
}
template <int W, int J>
inline double fma_bcast(double acc, const Bcast<W> &b, double x) {
  static_assert(J >= 0 && J < W, "");
  const double s = emu_exchange(b.v, (emu().cur & ~(W - 1)) | J, 32);
  return std::fma(s, x, acc);
}
template <int W, int J>
inline double value_bcast(const Bcast<W> &b) {
  return fma_bcast<W, J>(0.0, b, 1.0);
}
template <int W, class Op>
inline double emu_group_reduce(double v, Op op, int tag) {
  const int l = emu().cur;
  v = op(v, emu_exchange(v, l ^ 1, tag));
  v = op(v, emu_exchange(v, l ^ 2, tag));
  v = op(v, emu_exchange(v, (l & ~7) | (7 - (l & 7)), tag));
  if (W >= 16) v = op(v, emu_exchange(v, (l & ~15) | (15 - (l & 15)), tag));
  if (W >= 32) v = op(v, emu_exchange(v, l ^ 16, tag));
  if (W == 64) v = op(bcast(v, 0), bcast(v, 32));
  return v;
}
template <int W>
inline double group_sum(double v) {
  return emu_group_reduce<W>(v, [](double a, double b) { return a + b; }, 20);
}
template <int W>
inline double group_min(double v) {
  return emu_group_reduce<W>(v, [](double a, double b) { return std::fmin(a, b); }, 22);
}
// 32-bit arg-min key (wave.h): float with the payload in its low 8 mantissa bits
inline float key32_pack(double v, int payload) {
  const float f = std::fmax(std::fmin(static_cast<float>(v), -1.17549435e-38f), -3.0e38f);
  int b;
  std::memcpy(&b, &f, 4);
  b = (b & ~0xFF) | (payload & 0xFF);
  float r;
  std::memcpy(&r, &b, 4);
  return r;
}
inline float key32_packf(float v, int payload) {
  const float f = std::fmax(std::fmin(v, -1.17549435e-38f), -3.0e38f);
  int b;
  std::memcpy(&b, &f, 4);
  b = (b & ~0xFF) | (payload & 0xFF);
  float r;
  std::memcpy(&r, &b, 4);
  return r;
}
inline int key32_payload(float k) {
  int b;
  std::memcpy(&b, &k, 4);
  return b & 0xFF;
}
template <int W>
inline float group_min32(float v) {
  return static_cast<float>(emu_group_reduce<W>(static_cast<double>(v), [](double a, double b) { return std::fmin(a, b); }, 23));
}
// same association order as wave.h: shifts by 1, 2, 4, 8 inside the rows of 16, then the row totals
template <int W>
inline double group_scan_sum(double v) {
  const int l = emu().cur, li = l & (W - 1);
  for (int n = 1; n <= 8; n *= 2) {
    if (n == 8 && W < 16) break;
    double t = emu_exchange(v, (l & 15) >= n ? l - n : l, 26);
    if ((l & 15) < n) t = 0.0;  // row_shr with bound_ctrl: no source inside the row of 16
    v += (li >= n) ? t : 0.0;
  }
  if (W >= 32) {
    const int row = l >> 4;
    const double t = emu_exchange(v, (row & 1) ? (row - 1) * 16 + 15 : l, 28);
    v += (row & 1) ? t : 0.0;
  }
  if (W == 64) {
    const double t = emu_exchange(v, 31, 30);
    v += (l >= 32) ? t : 0.0;
  }
  return v;
}
template <int W>
inline int groups_max(int v) {
  int m = bcast_i(v, 0);
  for (int g = 1; g < kWave / W; ++g) {
    const int o = bcast_i(v, g * W);
    m = o > m ? o : m;
  }
  return m;
}
template <int W>
inline int groups_min(int v) {
  int m = bcast_i(v, 0);
  for (int g = 1; g < kWave / W; ++g) {
    const int o = bcast_i(v, g * W);
    m = o < m ? o : m;
  }
  return m;
}

struct v4d {
  double d[4];
  double &operator[](int i) { return d[i]; }
  const double &operator[](int i) const { return d[i]; }
};
// D(16x16) += A(16x4) B(4x16): lane l gives A[l&15][l>>4], B[l>>4][l&15], gets D[(l>>4)+4r][l&15]
inline v4d mfma_f64_16x16x4(double a, double b, v4d c) {
  Emu &e = emu();
  static double A[kWave], B[kWave];
  A[e.cur] = a;
  B[e.cur] = b;
  emu_rendezvous(24);
  const int col = e.cur & 15, rq = e.cur >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = rq + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k) acc += A[k * 16 + row] * B[k * 16 + col];
    c[r] = acc;
  }
  emu_rendezvous(25);
  return c;
}

inline double from_next_lane(double v) {
  Emu &e = emu();
  e.slot_d[e.cur] = v;
  emu_rendezvous(10);
  const double r = e.slot_d[e.cur < kWave - 1 ? e.cur + 1 : e.cur];
  emu_rendezvous(11);
  return r;
}

inline int from_next_lane_i(int v) {
  Emu &e = emu();
  e.slot_i[e.cur] = v;
  emu_rendezvous(12);
  const int r = e.slot_i[e.cur < kWave - 1 ? e.cur + 1 : e.cur];
  emu_rendezvous(13);
  return r;
}

// Run `fn(arg)` on all 64 lanes of workgroup `block`.
using LaneFn = void (*)(void *);

struct EmuLaunch {
  LaneFn fn;
  void *arg;
};

inline void emu_lane_entry(unsigned lo, unsigned hi) {
  EmuLaunch *L = reinterpret_cast<EmuLaunch *>((static_cast<unsigned long long>(hi) << 32) | lo);
  L->fn(L->arg);
  Emu &e = emu();
  const int me = e.cur;
  e.done[me] = true;
  e.live--;
  if (e.live == 0) {
    setcontext(&e.main_ctx);
  }
  int nxt = me;
  do {
    nxt = (nxt + 1) % kWave;
  } while (e.done[nxt]);
  e.cur = nxt;
  setcontext(&e.ctx[nxt]);
}

inline void emu_run_block(long long block, LaneFn fn, void *arg) {
  Emu &e = emu();
  constexpr size_t kStack = 512 * 1024;
  if (!e.stacks) e.stacks = static_cast<char *>(std::malloc(kStack * kWave));
  EmuLaunch L{fn, arg};
  const unsigned long long p = reinterpret_cast<unsigned long long>(&L);
  e.block = block;
  e.live = kWave;
  // poison LDS so that reads of never-written words are noticed (NaN)
  std::memset(e.lds, 0xff, sizeof(e.lds));
  for (int l = 0; l < kWave; ++l) {
    e.done[l] = false;
    e.arrivals[l] = 0;
    e.tag[l] = 0;
    getcontext(&e.ctx[l]);
    e.ctx[l].uc_stack.ss_sp = e.stacks + kStack * l;
    e.ctx[l].uc_stack.ss_size = kStack;
    e.ctx[l].uc_link = nullptr;
    makecontext(&e.ctx[l], reinterpret_cast<void (*)()>(emu_lane_entry), 2,
                static_cast<unsigned>(p & 0xffffffffu), static_cast<unsigned>(p >> 32));
  }
  e.cur = 0;
  swapcontext(&e.main_ctx, &e.ctx[0]);
}

}  // namespace pinkhip
