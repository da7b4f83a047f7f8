// ndt_sweep_kd.hpp -- derivative sweep for ndt_pca with the KDTREE neighbour search (ndt_pca_impl2.hpp:200-311 with
// search_method == KDTREE).  pclpca rescales a point's running sums by every hit's integer weight (impl2:294-296), so the
// result depends on the ORDER radiusSearch returns the leaves in: PCL's KdTreeFLANN sorts by ascending squared distance
// (ties by centroid index = std::map order).  No shipped configuration uses this combination; the kernel is written for
// fidelity, not speed: one lane per point, the up-to-27 candidates sorted in private memory, and the reference's nested
// arithmetic  S = (S + term) * w  evaluated literally in f64.
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_sweep.hpp"

template <int ORD>
__global__ void __launch_bounds__(SWEEP_THREADS)
k_sweep_pca_kd(const float* __restrict__ src, size_t pitch, const PairState* __restrict__ st, const GridDesc* __restrict__ gd,
               const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs, const float* __restrict__ cent,
               const int* __restrict__ kd_weight, double* partials, int chunks_per_pair, const int* __restrict__ active_list,
               SweepCtl* ctl, SweepCtl* ctl_next, SweepConst sc) {
  __shared__ double exp_tab[64];
  if (threadIdx.x < 64) exp_tab[threadIdx.x] = ndtm::c_exp2_64[threadIdx.x];
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 9) reinterpret_cast<int*>(ctl_next)[threadIdx.x] = 0;
  if ((int)blockIdx.y >= ctl->n_active) return;
  const int b = active_list[blockIdx.y], chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
  const PairState& S = st[b];
  const GridDesc& g = gd[b];
  const int n = S.n_src;
  const float* X = src + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const VoxelRec* R = recs + g.rec_off;
  const float* CE = cent + (size_t)g.rec_off * 3;
  const int* KW = kd_weight + g.rec_off;
  double acc[43];
#pragma unroll
  for (int a = 0; a < 43; a++) acc[a] = 0.0;
  unsigned nhits = 0;
  const int wbase = chunk * CHUNK_PTS + quarter * (CHUNK_PTS / QUARTERS);
  if (g.status == GRID_OK) {
    for (int t = 0; t < CHUNK_PTS / QUARTERS / 64; t++) {
      const int i = wbase + t * 64 + lane;
      if (i >= n) continue;
      const float px = X[i], py = X[pitch + i], pz = X[2 * pitch + i];
      if (!finite3(px, py, pz)) continue;
      float xt[3], r[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        xt[a] = ((S.T[a * 4 + 0] * px + S.T[a * 4 + 1] * py) + S.T[a * 4 + 2] * pz) + S.T[a * 4 + 3];
        r[a] = (S.Rj[a * 3 + 0] * px + S.Rj[a * 3 + 1] * py) + S.Rj[a * 3 + 2] * pz;
      }
      if (!finite3(xt[0], xt[1], xt[2])) continue;
      const int c0 = (int)floorf(sc.leaf_pow2 ? xt[0] * sc.inv_leaf : xt[0] / g.leaf);
      const int c1 = (int)floorf(sc.leaf_pow2 ? xt[1] * sc.inv_leaf : xt[1] / g.leaf);
      const int c2 = (int)floorf(sc.leaf_pow2 ? xt[2] * sc.inv_leaf : xt[2] / g.leaf);
      // radiusSearch(point, resolution): candidates are the leaves of the 3x3x3 block whose f32 centroid is closer than r
      float d2s[27];
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
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (!(d2 < sc.kd_r2)) continue;
        // insertion sort: ascending squared distance, ties by voxel id (= centroid index of the kd-tree)
        int k = m++;
        while (k > 0 && (d2s[k - 1] > d2 || (d2s[k - 1] == d2 && ids[k - 1] > id))) { d2s[k] = d2s[k - 1]; ids[k] = ids[k - 1]; k--; }
        d2s[k] = d2; ids[k] = id;
      }
      // impl2:268-296: score_pt / score_gradient_pt / hessian_pt grow hit by hit and are rescaled by each hit's weight
      double pt[43];
#pragma unroll
      for (int a = 0; a < 43; a++) pt[a] = 0.0;
      for (int k = 0; k < m; k++) {
        const VoxelRec& vr = R[ids[k]];
        float u[3] = {(float)((double)xt[0] - vr.mean[0]), (float)((double)xt[1] - vr.mean[1]), (float)((double)xt[2] - vr.mean[2])};
        float Cf[9];
#pragma unroll
        for (int a = 0; a < 9; a++) Cf[a] = vr.icov[a];
        eval_hit<false, NoHook, true, ORD>(u, r, Cf, sc.d1, sc.d2f, 1.0, true, pt, exp_tab);
        const double w = (double)KW[ids[k]];                     // (int)dimension_2d_, 0 for an eigen-failed leaf
#pragma unroll
        for (int a = 0; a < 43; a++) pt[a] *= w;
        nhits++;
      }
#pragma unroll
      for (int a = 0; a < 43; a++) acc[a] += pt[a];              // impl2:297-299
    }
  }
  // wave reduction (fixed tree) -> the (chunk, quarter) partial row, same layout as k_sweep's
  unsigned long long hw = nhits;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hw += __shfl_xor(hw, o);
#pragma unroll
  for (int a = 0; a < 43; a++) {
    double v = acc[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    acc[a] = v;
  }
  if (lane == 0) {
    double* P = partials + (((size_t)b * chunks_per_pair + chunk) * QUARTERS + quarter) * NACC;
#pragma unroll
    for (int a = 0; a < 43; a++) P[a] = acc[a];
    P[43] = (double)hw;
  }
}
