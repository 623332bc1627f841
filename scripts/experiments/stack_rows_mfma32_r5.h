// REJECTED VARIANT (round 5), kept for the record: the dense task rows of the fused stack + solve kernel stacked by the fp64
// matrix cores instead of broadcast-FMAs.  Correct (emulator and MI355X: 7e-15 against the oracle), SLOWER:
//     headline 0.620 -> 0.702 ms, tracking regime 0.255 -> 0.355 ms, kinematic bounds 0.575 -> 0.658 ms per 65 536
// (profiles/ab_mfma_stacking_r05.txt).  Why: v_mfma_f64_16x16x4 runs at the fp64 VECTOR rate on MI355X (78.6 TFLOP/s either
// way: 64 cycles per tile instruction and SIMD), the 30 x 30 problem pays for 32 x 32 tiles plus a third tile column for c (147 k
// flops issued per wavefront against 92 k of the broadcast-FMA form and 45 k useful), and the waves of a SIMD reach this phase
// together at the start of a round, so the matrix pipe serialises them while the VALU idles -- the overlap with other waves'
// VALU work that would have paid for it does not happen.  (ik_stack_mfma.h, the stack-ONLY kernel, is HBM bound: a different case.)
// Was called from ik_sweep.h / ik_sweepx.h in the place of stack_rows_bcast for W == 32 and rows from HBM; the transposition
// buffer (NV rows of pitch 34) overlaid both groups' LDS shares.
// The same sums on the fp64 MATRIX cores, for groups of 32 lanes (two QPs per wavefront) whose rows come from HBM:
//     H = sum_k J[k][:]^T (w_k^2 J[k][:]),   c = sum_k J[k][:]^T (gain_k w_k^2 e_k)
// as v_mfma_f64_16x16x4 tiles, one QP after the other (an MFMA is a wavefront-wide operation: all 64 lanes feed the
// tile of ONE instance).  Lane l requests J[4 s + (l >> 4)][16 t + (l & 15)] of k-step s and tile column t (16 lanes =
// 128 contiguous bytes): the value is the A operand of tile row t and, scaled by w^2, the B operand of tile column t;
// a third tile column whose B operand is gain w^2 e in its first column delivers c in the same instructions.  The
// accumulators (MFMA C/D layout: lane l holds rows (l >> 4) + 4 r of column l & 15) reach the solver's layout -- lane
// li of the group holds row li -- through LDS: the wavefront's whole LDS share serves as ONE buffer of NV rows of
// pitch 34, used for the first QP, then for the second (the solver parks its problem there only afterwards).
// Why: the broadcast-FMA stacking is 30 VALU instructions per task row and wavefront (936 of the ~10.7 k a wavefront
// of the headline issues, of ~4 k in the tracking regime) on the unit the kernel is bound by; the matrix pipe is
// otherwise idle and runs beside the VALU work of the SIMD's other waves (MI355X_MICROARCH.md: separate pipes).
// The fp64 MFMA accumulates its four products in k order like the FMA chain it replaces.
constexpr int kMfmaStackPitch = 34;  // doubles per row of the transposition buffer (16-byte aligned rows, column 32 = c)
template <int NV, class Mid = NoMid, int NM>
__device__ __forceinline__ void stack_rows_mfma32(const KernelArgs &a, long long block, bool in, int li, double (&M)[NM], double &ci,
                                                  double &mu_l, double *buf, Mid mid = Mid()) {
  static_assert(NV <= 32 && NM >= NV, "two 16-wide tile columns");
  constexpr int P = kMfmaStackPitch, KS = 6;  // k-steps (of four task rows) requested at once: 24 rows, the BASELINE stacks
  const int lane = lane_id(), col = lane & 15, rq = lane >> 4, g_own = lane >> 5;
  const int nv = a.nv, Kd = a.Kd, K = a.K;
  long long bq[2] = {block * 2, block * 2 + 1};
  if (bq[0] >= a.B) bq[0] = a.B - 1;
  if (bq[1] >= a.B) bq[1] = a.B - 1;  // (surplus group of the last wavefront: redoes the last instance)
  // this group's share of the Levenberg-Marquardt term (task.py:160), one task row per lane
  {
    const long long b = bq[g_own];
    const double *eb = a.e + b * (long long)K;
    const double *costb = a.cost_batched ? a.cost + b * (long long)K : a.cost;
    for (int k = li; k < Kd; k += 32) {
      const double w = costb[k], ev = eb[k], gn = a.row_gain[k];
      mu_l += a.row_lm[k] * (gn * gn) * (w * w) * ev * ev;
    }
  }
  // (the rows of BOTH instances are requested before anything is accumulated: one memory latency per wavefront; the
  // per-row coefficients -- a few cached table entries -- are fetched per instance, or both would not fit the registers)
  double Jp[2][KS][2];
  auto request = [&](int q, int r0) {
    const double *Jb = a.J + bq[q] * (long long)Kd * nv;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = r0 + 4 * s + rq;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = 16 * t + col;
        Jp[q][s][t] = (k < Kd && j < nv) ? Jb[(long long)k * nv + j] : 0.0;
      }
    }
  };
  if (Kd > 0) {
    request(0, 0);
    request(1, 0);
  }
  mid();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double *eb = a.e + bq[q] * (long long)K;
    const double *costb = a.cost_batched ? a.cost + bq[q] * (long long)K : a.cost;
    wave_sync();  // (the previous QP's rows have been read)
    // one ROW of tiles at a time (three accumulators live, not six: the registers the solver's own state needs are not
    // spilled around this phase); more than 24 dense rows: the later chunks are requested as they are needed, per tile row
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {
      v4d acc[3];
#pragma unroll
      for (int tc = 0; tc < 3; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tc][r] = 0.0;
      for (int r0 = 0; r0 < Kd; r0 += 4 * KS) {
        if (r0 > 0 || (tr > 0 && Kd > 4 * KS)) request(q, r0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (r0 + 4 * s < Kd) {  // wave-uniform
            const int k = r0 + 4 * s + rq;
            const double w = (k < Kd) ? costb[k] : 0.0;
            const double wa = w * w;
            const double gw = (k < Kd && col == 0) ? a.row_gain[k] * wa * eb[k] : 0.0;
            acc[0] = mfma_f64_16x16x4(Jp[q][s][tr], wa * Jp[q][s][0], acc[0]);
            acc[1] = mfma_f64_16x16x4(Jp[q][s][tr], wa * Jp[q][s][1], acc[1]);
            acc[2] = mfma_f64_16x16x4(Jp[q][s][tr], gw, acc[2]);
          }
        }
      }
      // accumulators -> LDS (row 16 tr + rq + 4 r, column 16 tc + col; c in column 32)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tr + rq + 4 * r;
        if (row < NV) {
          buf[row * P + col] = acc[0][r];
          buf[row * P + 16 + col] = acc[1][r];
          if (col == 0) buf[row * P + 32] = acc[2][r];
        }
      }
    }
    wave_sync();
    // ... -> row li of this QP's group
    if (g_own == q && li < NV) {
      const double *row = buf + li * P;
#pragma unroll
      for (int j = 0; j < NV; ++j) M[j] = row[j];
      ci += row[32];
    }
  }
  wave_sync();  // (the caller writes its own data into the buffer next)
  (void)in;
}

