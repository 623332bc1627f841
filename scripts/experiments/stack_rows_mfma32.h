// Measured and not adopted (see README.md in this directory): stacking of the dense task rows on the fp64 MFMA inside the
// sweep-tableau kernel.  Fragment of ik_stack_rows.h as it was when measured.
// The same sums on the fp64 matrix cores, for the kernels with TWO QPs per wavefront (32 lanes per QP) and NV <= 30.
//
// The sweep-tableau kernel is bound by VALU throughput and the ~30 broadcast-FMAs per task row are VALU work; the
// matrix pipe of the SIMD is idle.  v_mfma_f64_16x16x4_f64 has the VECTOR rate on MI355X (78.6 TFLOP/s both), but it
// runs beside the VALU: H = (W^2 J)^T J as 2 x 2 tiles of 16 x 16, K consumed four task rows per instruction -- 24
// matrix instructions per QP instead of ~770 VALU FMAs per wave.  Lane l loads J[k0 + (l >> 4)][16 tc + (l & 15)]
// straight from HBM; the value is the B operand of tile column tc and, scaled by w_k^2, the A operand of tile row tc
// (as in ik_stack_mfma.h).  Column 30 of H is padding at NV <= 30: the lanes that supply it hand over gain_k e_k
// instead of J[k][30] = 0, so that D[i][30] = sum_k w_k^2 J[k][i] gain_k e_k = c_i comes out of the same
// instructions.  The accumulators (lane l: H[(l >> 4) + 4 r][l & 15]) reach the row-per-lane layout through LDS, one
// row of tiles (16 x 34 doubles) at a time, in the region the kernel parks the stated problem in afterwards.
// Returns this QP-row's H entries in M[0 .. NV), c in ci and -- in lane li = 0 of each group -- the Levenberg-
// Marquardt sum of the dense rows in mu_l.
constexpr int kMfmaStackLdsDoubles = 32 * 33 + 128;  // what stack_rows_mfma32 needs of a wave's LDS

template <int NV, int NM>
__device__ __forceinline__ void stack_rows_mfma32(const KernelArgs &a, long long block, double *lds, int g, int li,
                                                  double (&M)[NM], double &ci, double &mu_l) {
  static_assert(NV <= 30 && NM >= NV, "column 30 of the tiles carries c");
  constexpr int PITCH = 33, KS = 8;  // LDS row pitch (doubles): odd, so that the 32 rows start in different banks
  const int lane = lane_id();
  const int col = lane & 15, rq = lane >> 4;
  const int nv = a.nv, Kd = a.Kd, K = a.K;
  double *tmp = lds;                // [32][PITCH] H of one QP (column 30: c)
  double *tw = lds + 32 * PITCH;    // [2][32] w_k^2 of the rows of a pass, per QP
  double *tge = tw + 64;            // [2][32] gain_k e_k
  double mu_own = 0.0;              // Levenberg-Marquardt terms of this lane's QP (lane >> 5)
  double R0[NV], c0 = 0.0;          // row (lane & 31) of QP 0, kept while QP 1 is computed
  static_for<0, 2>([&](auto Qc) {
    constexpr int q = decltype(Qc)::value;
    long long bq = block * 2 + q;
    if (bq >= a.B) bq = a.B - 1;
    const double *Jq = a.J + bq * (long long)Kd * nv;
    v4d acc[2][2];  // [tile row][tile column]
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ti][tj][r] = 0.0;
    for (int p0 = 0; p0 < Kd; p0 += 4 * KS) {  // 32 task rows per pass: one memory round trip
      const int rc = (Kd - p0 < 4 * KS) ? Kd - p0 : 4 * KS;
      double Jp[KS][2];
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const int kk = p0 + 4 * st + rq;
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
          const int j = 16 * tc + col;
          Jp[st][tc] = (4 * st < rc && kk < Kd && j < nv) ? Jq[(long long)kk * nv + j] : 0.0;
        }
      }
      // coefficients of the pass: lane 32 q + r holds row p0 + r of this QP
      wave_sync();
      if ((lane >> 5) == q) {
        const int r = lane & 31;
        double wa = 0.0, ge = 0.0;
        if (r < rc) {
          const int k = p0 + r;
          const double w = (a.cost_batched ? a.cost + bq * (long long)K : a.cost)[k], ev = a.e[bq * (long long)K + k];
          const double gn = a.row_gain[k];
          wa = w * w;
          ge = gn * ev;
          mu_own += a.row_lm[k] * (gn * gn) * wa * ev * ev;
        }
        tw[lane] = wa;
        tge[lane] = ge;
      }
      wave_sync();
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        if (4 * st < rc) {  // wave-uniform
          const int kr = 4 * st + rq;
          const double wa = tw[32 * q + kr];
          const double Bv[2] = {Jp[st][0], col == 14 ? tge[32 * q + kr] : Jp[st][1]};
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            const double Av = wa * Jp[st][ti];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = mfma_f64_16x16x4(Av, Bv[tj], acc[ti][tj]);
          }
        }
      }
    }
    // accumulators (lane l: H[16 ti + (l >> 4) + 4 r][16 tj + (l & 15)]) -> LDS -> every lane reads row (lane & 31)
    wave_sync();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) tmp[(16 * ti + rq + 4 * r) * PITCH + 16 * tj + col] = acc[ti][tj][r];
    wave_sync();
    const double *rowp = tmp + (lane & 31) * PITCH;
    if constexpr (q == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) R0[j] = rowp[j];
      c0 = rowp[30];
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const double v = rowp[j];
        M[j] = (g == 0) ? R0[j] : v;
      }
      const double c1 = rowp[30];
      ci = (g == 0) ? c0 : c1;
    }
  });
  wave_sync();
  const double mu_sum = group_sum<32>(mu_own);
  if (li == 0) mu_l += mu_sum;
}

