// ndt_hessian.hpp -- computeHessian + updateHessian (include/ndt_omp/ndt_omp_impl2.hpp:622-714), the f64 Hessian-only pass
// computeStepLengthMT runs after its More-Thuente loop iterated at least once (impl2:999-1000).  That loop is live only
// when step_size <= transformation_epsilon/2 (impl2:888) -- no shipped lv_slam configuration -- so this kernel is written
// for clarity, not speed: one lane per point, 27-cell probe + centroid radius test (= radiusSearch, as in k_sweep<.,27>).
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_sweep.hpp"

#define HESS_THREADS 256

__global__ void __launch_bounds__(HESS_THREADS)
k_hessian(const float* __restrict__ src, size_t pitch, const PairState* __restrict__ st, const GridDesc* __restrict__ gd,
          const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs, const double* __restrict__ icov64,
          const float* __restrict__ cent, double* partials, int chunks_per_pair, double d1, double d2, float kd_r2,
          int leaf_pow2, float inv_leaf) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const PairState& S = st[b];
  if (S.phase != PH_HESS) return;
  const int n = S.n_src;
  if (chunk * CHUNK_PTS >= n) return;
  const GridDesc& g = gd[b];
  const float* X = src + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const VoxelRec* R = recs + g.rec_off;
  const double* IC = icov64 + (size_t)g.rec_off * 9;
  const float* CE = cent + (size_t)g.rec_off * 3;
  __shared__ double Rm[9];
  __shared__ double red[HESS_THREADS / 64][36];
  if (threadIdx.x == 0) {
    ndtm::SE3 e = ndtm::se3_exp(S.xt);                          // impl2:539: exp(p).matrix(), double
    double Rt[9];
    ndtm::q_to_matrix(e.q, Rt);
    for (int a = 0; a < 9; a++) Rm[a] = Rt[a];
  }
  __syncthreads();
  double H[36];
#pragma unroll
  for (int a = 0; a < 36; a++) H[a] = 0.0;
  const bool grid_ok = g.status == GRID_OK;
  for (int k = 0; k < CHUNK_PTS / HESS_THREADS; k++) {
    const int i = chunk * CHUNK_PTS + k * HESS_THREADS + threadIdx.x;
    if (i >= n || !grid_ok) continue;
    const float px = X[i], py = X[pitch + i], pz = X[2 * pitch + i];
    if (!finite3(px, py, pz)) continue;
    float xt[3];
#pragma unroll
    for (int a = 0; a < 3; a++) xt[a] = ((S.T[a * 4 + 0] * px + S.T[a * 4 + 1] * py) + S.T[a * 4 + 2] * pz) + S.T[a * 4 + 3];
    if (!finite3(xt[0], xt[1], xt[2])) continue;
    const double xo[3] = {(double)px, (double)py, (double)pz};
    double r[3];
#pragma unroll
    for (int a = 0; a < 3; a++) r[a] = (Rm[a * 3 + 0] * xo[0] + Rm[a * 3 + 1] * xo[1]) + Rm[a * 3 + 2] * xo[2];
    // point_gradient_ (3x6): I | rotation columns (impl2:544-549)
    const double J[3][6] = {{1, 0, 0, 0, r[2], -r[1]}, {0, 1, 0, -r[2], 0, r[0]}, {0, 0, 1, r[1], -r[0], 0}};
    const int c0 = (int)floorf(leaf_pow2 ? xt[0] * inv_leaf : xt[0] / g.leaf);
    const int c1 = (int)floorf(leaf_pow2 ? xt[1] * inv_leaf : xt[1] / g.leaf);
    const int c2 = (int)floorf(leaf_pow2 ? xt[2] * inv_leaf : xt[2] / g.leaf);
    for (int q = 0; q < 27; q++) {
      const int q0 = c0 + (q % 3 - 1), q1 = c1 + ((q / 3) % 3 - 1), q2 = c2 + (q / 9 - 1);
      if (q0 < g.min_b[0] || q0 > g.max_b[0] || q1 < g.min_b[1] || q1 > g.max_b[1] || q2 < g.min_b[2] || q2 > g.max_b[2]) continue;
      const unsigned cell = (unsigned)((q0 - g.min_b[0]) + (q1 - g.min_b[1]) * g.mul1 + (q2 - g.min_b[2]) * g.mul2);
      const BitWord bw = W[cell >> 6];
      if (!((bw.bits >> (cell & 63u)) & 1ull)) continue;
      const unsigned id = bw.prefix + (unsigned)__popcll(bw.bits & ((1ull << (cell & 63u)) - 1ull));
      const float dx = xt[0] - CE[3 * id], dy = xt[1] - CE[3 * id + 1], dz = xt[2] - CE[3 * id + 2];
      if (!(((dx * dx + dy * dy) + dz * dz) < kd_r2)) continue;          // radiusSearch: no nr_points re-check
      const VoxelRec& vr = R[id];
      const double u[3] = {(double)xt[0] - vr.mean[0], (double)xt[1] - vr.mean[1], (double)xt[2] - vr.mean[2]};
      double C[9];
#pragma unroll
      for (int a = 0; a < 9; a++) C[a] = IC[(size_t)id * 9 + a];
      double Cu[3];
#pragma unroll
      for (int a = 0; a < 3; a++) Cu[a] = (C[a * 3 + 0] * u[0] + C[a * 3 + 1] * u[1]) + C[a * 3 + 2] * u[2];
      double e = d2 * exp(-d2 * ((u[0] * Cu[0] + u[1] * Cu[1]) + u[2] * Cu[2]) / 2);   // impl2:691
      if (e > 1 || e < 0 || e != e) continue;                                           // impl2:694-695
      e *= d1;
      double CJ[6][3], uCJ[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int a = 0; a < 3; a++) CJ[c][a] = (C[a * 3 + 0] * J[0][c] + C[a * 3 + 1] * J[1][c]) + C[a * 3 + 2] * J[2][c];
        uCJ[c] = (u[0] * CJ[c][0] + u[1] * CJ[c][1]) + u[2] * CJ[c][2];
      }
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
          // point_hessian_.block<3,1>(3a, c) (impl2:555-563): non-zero only for a, c in 3..5
          double hp[3] = {0, 0, 0};
          if (a == 3 && c == 3) { hp[1] = -r[1]; hp[2] = -r[2]; }
          if (a == 4 && c == 3) { hp[0] = r[1]; }
          if (a == 5 && c == 3) { hp[0] = r[2]; }
          if (a == 3 && c == 4) { hp[1] = r[0]; }
          if (a == 4 && c == 4) { hp[0] = -r[0]; hp[2] = -r[2]; }
          if (a == 5 && c == 4) { hp[1] = r[2]; }
          if (a == 3 && c == 5) { hp[2] = r[0]; }
          if (a == 4 && c == 5) { hp[2] = r[1]; }
          if (a == 5 && c == 5) { hp[0] = -r[0]; hp[1] = -r[1]; }
          double CH[3];
#pragma unroll
          for (int q3 = 0; q3 < 3; q3++) CH[q3] = (C[q3 * 3 + 0] * hp[0] + C[q3 * 3 + 1] * hp[1]) + C[q3 * 3 + 2] * hp[2];
          const double t2 = (u[0] * CH[0] + u[1] * CH[1]) + u[2] * CH[2];
          const double t3 = (J[0][c] * CJ[a][0] + J[1][c] * CJ[a][1]) + J[2][c] * CJ[a][2];
          H[a * 6 + c] += e * (-d2 * uCJ[a] * uCJ[c] + t2 + t3);                          // impl2:709-711
        }
      }
    }
  }
  // fixed-order block reduction -> the H columns of quarter 0 of this chunk's partial rows (other quarters: zero)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 36; a++) {
    double v = H[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wv][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 36) {
    double* P = partials + ((size_t)b * chunks_per_pair + chunk) * QUARTERS * NACC;
    const int a = threadIdx.x;
    P[7 + a] = ((red[0][a] + red[1][a]) + red[2][a]) + red[3][a];
    for (int qd = 1; qd < QUARTERS; qd++) P[qd * NACC + 7 + a] = 0.0;
  }
}


// ------------------------------------------------------------------------------------ calculateScore
// calculateScore (include/ndt_omp/ndt_omp_impl2.hpp:1006-1040; ndt_pca_impl2.hpp:1013-1047): negative log-likelihood of an already
// transformed cloud, f64 throughout, kd-tree neighbourhoods (27-cell probe + centroid radius test = radiusSearch, as in k_hessian).
// One lane per point: the neighbours are counted first (the reference divides every term by neighborhood.size(), impl2:1036), then
// summed; per-block partial sums with a fixed tree, added in block order on the host.  No caller in lv_slam: fidelity, not speed.
#define SCORE_THREADS 256
__global__ void __launch_bounds__(SCORE_THREADS)
k_calc_score(const float* __restrict__ pts, size_t pitch, int n, const GridDesc* __restrict__ gd, const BitWord* __restrict__ words,
             const VoxelRec* __restrict__ recs, const double* __restrict__ icov64, const float* __restrict__ cent,
             double d1, double d2, double d3, float kd_r2, int leaf_pow2, float inv_leaf, double* part) {
  const GridDesc& g = gd[0];
  const BitWord* W = words + g.word_off;
  const VoxelRec* R = recs + g.rec_off;
  const double* IC = icov64 + (size_t)g.rec_off * 9;
  const float* CE = cent + (size_t)g.rec_off * 3;
  __shared__ double red[SCORE_THREADS / 64];
  const int i = blockIdx.x * SCORE_THREADS + threadIdx.x;
  double s = 0.0;
  if (i < n && g.status == GRID_OK) {
    const float xt[3] = {pts[i], pts[pitch + i], pts[2 * pitch + i]};
    if (finite3(xt[0], xt[1], xt[2])) {
      const int c0 = (int)floorf(leaf_pow2 ? xt[0] * inv_leaf : xt[0] / g.leaf);
      const int c1 = (int)floorf(leaf_pow2 ? xt[1] * inv_leaf : xt[1] / g.leaf);
      const int c2 = (int)floorf(leaf_pow2 ? xt[2] * inv_leaf : xt[2] / g.leaf);
      unsigned ids[27];
      int m = 0;
      for (int q = 0; q < 27; q++) {
        const int q0 = c0 + (q % 3 - 1), q1 = c1 + ((q / 3) % 3 - 1), q2 = c2 + (q / 9 - 1);
        if (q0 < g.min_b[0] || q0 > g.max_b[0] || q1 < g.min_b[1] || q1 > g.max_b[1] || q2 < g.min_b[2] || q2 > g.max_b[2]) continue;
        const unsigned cell = (unsigned)((q0 - g.min_b[0]) + (q1 - g.min_b[1]) * g.mul1 + (q2 - g.min_b[2]) * g.mul2);
        const BitWord bw = W[cell >> 6];
        if (!((bw.bits >> (cell & 63u)) & 1ull)) continue;
        const unsigned id = bw.prefix + (unsigned)__popcll(bw.bits & ((1ull << (cell & 63u)) - 1ull));
        const float dx = xt[0] - CE[3 * id], dy = xt[1] - CE[3 * id + 1], dz = xt[2] - CE[3 * id + 2];
        if (!(((dx * dx + dy * dy) + dz * dz) < kd_r2)) continue;          // radiusSearch: strict <, no nr_points re-check
        ids[m++] = id;
      }
      const double cnt = (double)m;
      for (int k = 0; k < m; k++) {
        const unsigned id = ids[k];
        const VoxelRec& vr = R[id];
        const double u[3] = {(double)xt[0] - vr.mean[0], (double)xt[1] - vr.mean[1], (double)xt[2] - vr.mean[2]};   // impl2:1025-1028
        double Cu[3];
#pragma unroll
        for (int a = 0; a < 3; a++) Cu[a] = (IC[(size_t)id * 9 + a * 3 + 0] * u[0] + IC[(size_t)id * 9 + a * 3 + 1] * u[1]) + IC[(size_t)id * 9 + a * 3 + 2] * u[2];
        const double e = exp(-d2 * ((u[0] * Cu[0] + u[1] * Cu[1]) + u[2] * Cu[2]) / 2);   // impl2:1033
        const double inc = -d1 * e - d3;                                                 // impl2:1035
        s += inc / cnt;                                                                  // impl2:1037
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
