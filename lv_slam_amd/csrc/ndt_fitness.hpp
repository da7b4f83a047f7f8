// ndt_fitness.hpp -- getFitnessScore(max_range) for the loop-closure caller.
#pragma once
#include "ndt_types.hpp"

// ------------------------------------------------------------------------------------ fitness score (loop-closure caller)
// pcl::Registration::getFitnessScore(max_range) as used by include/global_graph/loop_detector.hpp:249-262, same recipe as
// the in-tree InformationMatrixCalculator::calc_fitness_score (src/global_graph/information_matrix_calculator.cpp:53-87):
// move the source by the final pose (f32), exact nearest target point per source point, and average the SQUARED
// distances that are <= max_range (the comparison really is squared distance vs max_range in the reference).
// The exact 1-NN runs on the target's voxel binning that setInputTarget already sorted: cells are visited ring by ring
// around the query's cell and the search stops once the best distance cannot be beaten by an unvisited ring.
template <typename KeyT>
__global__ void __launch_bounds__(256) k_cellrange(const KeyT* __restrict__ keys, size_t pitch, int cb, unsigned* cstart, unsigned* cend) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pitch) return;
  const unsigned cmask = (1u << cb) - 1u;
  const KeyT key = keys[i];
  const unsigned cell = (unsigned)key & cmask;
  if (cell == cmask) return;
  if (i == 0 || keys[i - 1] != key) cstart[cell] = (unsigned)i;
  if (i + 1 == pitch || keys[i + 1] != key) cend[cell] = (unsigned)i + 1u;
}

__global__ void __launch_bounds__(256) k_fitness(const float* __restrict__ src, size_t spitch, int n_src,
                                                 const float* __restrict__ tgt, size_t tpitch, const unsigned* __restrict__ vals,
                                                 const GridDesc* __restrict__ gd, const unsigned* __restrict__ cstart, const unsigned* __restrict__ cend,
                                                 const float* __restrict__ Tcm, float max_range, int ring_max, double* partial) {
  // ring_max: rings needed to cover sqrt(max_range); the kernel also never walks past the grid's far side
  const GridDesc& g = gd[0];
  double sum = 0.0;
  unsigned long long cnt = 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_src && g.status == GRID_OK) {
    const float px = src[i], py = src[spitch + i], pz = src[2 * spitch + i];
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) q[a] = ((Tcm[0 * 4 + a] * px + Tcm[1 * 4 + a] * py) + Tcm[2 * 4 + a] * pz) + Tcm[3 * 4 + a];   // PCL 1.8 scalar form
    if (finite3(q[0], q[1], q[2])) {
      // the query's cell relative to the grid, clamped to +-2^29 cells: a query further out than that (a stray source point at
      // 1e12 m) still sees every target cell in rings r_first .. r_first + extent, and (r - 1) * leaf stays a lower bound of its distances
      int cq[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float cf = floorf(q[a] * g.inv_leaf);
        const long long ci = cf >= 1.0e9f ? 1000000000ll : (cf <= -1.0e9f ? -1000000000ll : (long long)cf);
        const long long cl = ci - (long long)g.min_b[a];
        cq[a] = (int)(cl > (1ll << 29) ? (1ll << 29) : (cl < -(1ll << 29) ? -(1ll << 29) : cl));
      }
      const int c0 = cq[0], c1 = cq[1], c2 = cq[2];
      // distance from the query's cell to the grid box in cells (0 inside): rings closer than that are empty
      const int o0 = c0 < 0 ? -c0 : (c0 >= g.div_b[0] ? c0 - g.div_b[0] + 1 : 0);
      const int o1 = c1 < 0 ? -c1 : (c1 >= g.div_b[1] ? c1 - g.div_b[1] + 1 : 0);
      const int o2 = c2 < 0 ? -c2 : (c2 >= g.div_b[2] ? c2 - g.div_b[2] + 1 : 0);
      const int r_first = max(o0, max(o1, o2));
      const int r_last = min(ring_max, r_first + max(g.div_b[0], max(g.div_b[1], g.div_b[2])) + 1);
      float best = __int_as_float(0x7f800000);
      for (int r = r_first; r <= r_last; r++) {
        // every point in ring >= r lies more than (r-1)*leaf away; 0.1 % of a cell of slack for the binning's rounding
        const float reach = ((float)(r - 1) - 1e-3f) * g.leaf;
        if (r > 1 && (best <= reach * reach || reach * reach > max_range)) break;
        const int z0 = max(c2 - r, 0), z1 = min(c2 + r, g.div_b[2] - 1);
        const int y0 = max(c1 - r, 0), y1 = min(c1 + r, g.div_b[1] - 1);
        const int x0 = max(c0 - r, 0), x1 = min(c0 + r, g.div_b[0] - 1);
        for (int z = z0; z <= z1; z++) {
          const bool zface = (z == c2 - r || z == c2 + r);
          for (int y = y0; y <= y1; y++) {
            const bool yface = (y == c1 - r || y == c1 + r);
            const int step = (zface || yface) ? 1 : max(1, (c0 + r) - (c0 - r));   // interior rows: only the two x faces
            for (int x = (zface || yface) ? x0 : c0 - r; x <= x1; x += step) {
              if (x < x0) continue;
              const unsigned cell = (unsigned)(x + y * g.mul1 + z * g.mul2);
              const unsigned s = cstart[cell], e = cend[cell];
              for (unsigned j = s; j < e; j++) {
                const unsigned pi = vals[j];
                const float dx = q[0] - tgt[pi], dy = q[1] - tgt[tpitch + pi], dz = q[2] - tgt[2 * tpitch + pi];
                const float d2 = (dx * dx + dy * dy) + dz * dz;          // FLANN L2_Simple accumulation order
                best = d2 < best ? d2 : best;
              }
            }
          }
        }
      }
      if (best <= max_range) { sum = (double)best; cnt = 1; }
    }
  }
  // deterministic block reduction
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); cnt += __shfl_xor(cnt, o); }
  __shared__ double rs[4];
  __shared__ unsigned long long rc[4];
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = sum; rc[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ((rs[0] + rs[1]) + rs[2]) + rs[3];
    partial[2 * blockIdx.x + 1] = (double)(rc[0] + rc[1] + rc[2] + rc[3]);
  }
}

// The same score for a target that has no voxel grid (the leaf-too-small guard or the engine's cell cap: GRID_OVERFLOW / GRID_CAP) --
// pcl::Registration::getFitnessScore searches a kd-tree over the target CLOUD and does not care.  Exhaustive search, the target
// staged through LDS 256 points at a time; same distance arithmetic, same reduction as k_fitness.
__global__ void __launch_bounds__(256) k_fitness_brute(const float* __restrict__ src, size_t spitch, int n_src,
                                                       const float* __restrict__ tgt, size_t tpitch, int n_tgt,
                                                       const float* __restrict__ Tcm, float max_range, double* partial) {
  __shared__ float tx[256], ty[256], tz[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float q[3] = {0.f, 0.f, 0.f};
  bool live = false;
  if (i < n_src) {
    const float px = src[i], py = src[spitch + i], pz = src[2 * spitch + i];
#pragma unroll
    for (int a = 0; a < 3; a++) q[a] = ((Tcm[0 * 4 + a] * px + Tcm[1 * 4 + a] * py) + Tcm[2 * 4 + a] * pz) + Tcm[3 * 4 + a];   // PCL 1.8 scalar form
    live = finite3(q[0], q[1], q[2]);
  }
  float best = __int_as_float(0x7f800000);
  for (int j0 = 0; j0 < n_tgt; j0 += 256) {
    const int j = j0 + threadIdx.x;
    float x = __int_as_float(0x7fc00000), y = x, z = x;                            // past the end: NaN, skipped below
    if (j < n_tgt) { x = tgt[j]; y = tgt[tpitch + j]; z = tgt[2 * tpitch + j]; }
    __syncthreads();
    tx[threadIdx.x] = x; ty[threadIdx.x] = y; tz[threadIdx.x] = z;
    __syncthreads();
    if (live) {
      for (int k = 0; k < 256; k++) {
        if (!finite3(tx[k], ty[k], tz[k])) continue;                                // non-finite target points are in no tree
        const float dx = q[0] - tx[k], dy = q[1] - ty[k], dz = q[2] - tz[k];
        const float d2 = (dx * dx + dy * dy) + dz * dz;                             // FLANN L2_Simple accumulation order
        best = d2 < best ? d2 : best;
      }
    }
  }
  double sum = 0.0;
  unsigned long long cnt = 0;
  if (live && best <= max_range) { sum = (double)best; cnt = 1; }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); cnt += __shfl_xor(cnt, o); }
  __shared__ double rs[4];
  __shared__ unsigned long long rc[4];
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = sum; rc[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ((rs[0] + rs[1]) + rs[2]) + rs[3];
    partial[2 * blockIdx.x + 1] = (double)(rc[0] + rc[1] + rc[2] + rc[3]);
  }
}

