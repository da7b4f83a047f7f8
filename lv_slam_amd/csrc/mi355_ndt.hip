// mi355_ndt.hip -- MI355X (gfx950) NDT scan-matching engine behind the C-ABI of include/mi355_ndt.h.
//
// What runs where (all on the GPU; the host only enqueues):
//   target build  : k_minmax -> k_griddesc -> k_keys -> radix sort (cell, input order) -> k_mark
//                   -> k_rank -> k_segstart -> k_leafsum -> k_voxels          (VoxelGridCovariance::applyFilter,
//                   include/ndt_omp/voxel_grid_covariance_omp_impl.hpp:48-370)
//   align         : k_init_state -> k_sweep -> [k_update -> k_sweep]*      (computeTransformation +
//                   computeDerivatives + the live prefix of computeStepLengthMT,
//                   include/ndt_omp/ndt_omp_impl2.hpp:87-188, 196-305, 841-907)
// Data layout in HBM: DESIGN.md.  Built with -ffp-contract=off: every f32/f64 step of the
// reference recipe is a separately rounded operation.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>

#include "mi355_ndt.h"
#include "ndt_math.hpp"

// ------------------------------------------------------------------------------------ constants
#define CHUNK_PTS      2048          // source points per reduction chunk (fixed => results independent of launch geometry)
#define SWEEP_THREADS  256
#ifndef SWEEP_WPE
#define SWEEP_WPE       2             // waves per SIMD the sweep is register-allocated for (2: no spills; measured faster than 3 with spills)
#endif
#define NACC           44            // score, g[6], H[36], hits
// sort key = pair << cb | cell, cb = bits needed for the largest grid of the batch + the all-ones "not binned" cell;
// 32-bit keys whenever pair and cell fields fit (the usual case), else 64-bit
#define MAX_CELLS      (1 << 25)

enum { GRID_OK = 0, GRID_EMPTY = 1, GRID_OVERFLOW = 2, GRID_CAP = 3 };
enum { PH_SWEEP0 = 0, PH_STEP = 1, PH_DONE = 2 };

struct GridDesc {              // one per target
  int   min_b[3], max_b[3], div_b[3];
  int   mul1, mul2;            // divb_mul_ = (1, mul1, mul2)
  float leaf, inv_leaf;
  int   ncells, nwords;
  int   status;
  int   n_voxels;              // searchable leaves (n >= min_points), including eigen-failed ones
  unsigned word_off;           // into the BitWord pool
  unsigned rec_off;            // into the VoxelRec pool
};

struct BitWord {               // occupancy of 64 consecutive cells + rank of the first one
  unsigned long long bits;
  unsigned prefix;
  unsigned pad;
};

struct VoxelRec {              // 64 B, what one (point, voxel) evaluation reads
  double mean[3];
  float  icov[9];
  int    weight;               // ndt_pca integer weight; 1 for ndt_omp; INT_MIN = dead (eigen/inverse failure)
};
#define VOX_DEAD INT_MIN

struct PairState {
  float  T[12];                // 3x4 row-major point transform (f32)
  float  Rj[9];                // rotation used for the point Jacobian (f32)
  double p[6];                 // current tangent [upsilon; omega]
  double dir[6];               // pending step direction
  double a_t;                  // pending step length
  double score, g[6], H[36];
  double trans_probability;
  long long hits;
  float  final_cm[16];
  int    it, phase, converged, sweeps, n_src, grid_status;
};

struct SweepConst {
  double d1;
  float  d2f;
  int    K;                    // neighbour probes
  int    pca;
  int    table;                // row of c_off: 0 = DIRECT1, 1 = DIRECT7, 2 = DIRECT26
  int    leaf_pow2;            // resolution is a power of two: x / leaf == x * inv_leaf bit for bit
  float  inv_leaf;
};

// Neighbour offsets in the reference's probe order.  DIRECT1: voxel_grid_covariance_omp_impl.hpp:441;
// DIRECT7: impl:423-430; DIRECT26: pcl::getAllNeighborCellIndices() (PCL 1.8 voxel_grid.h) = 13 "half"
// offsets followed by their negation.  __constant__: the wave-uniform probe index reads them with scalar loads.
__constant__ int c_off[3][26][3] = {
  {{0,0,0}},
  {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1}},
  {{-1,-1,-1}, {-1,0,-1}, {-1,1,-1}, {0,-1,-1}, {0,0,-1}, {0,1,-1}, {1,-1,-1}, {1,0,-1}, {1,1,-1}, {-1,-1,0}, {0,-1,0}, {1,-1,0}, {-1,0,0}, {1,1,1}, {1,0,1}, {1,-1,1}, {0,1,1}, {0,0,1}, {0,-1,1}, {-1,1,1}, {-1,0,1}, {-1,-1,1}, {1,1,0}, {0,1,0}, {-1,1,0}, {1,0,0}}
};

// ------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }
__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// ------------------------------------------------------------------------------------ target build
__global__ void k_minmax_init(int* mm, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 6) mm[i] = (i % 6) < 3 ? INT_MAX : INT_MIN;
}

// getMinMax3D over finite points (voxel_grid_covariance_omp_impl.hpp:72, 211-216)
__global__ void __launch_bounds__(256) k_minmax(const float* __restrict__ tgt, size_t pitch, const int* __restrict__ cnt, int* mm) {
  const int b = blockIdx.y;
  const int n = cnt[b];
  const float* X = tgt + (size_t)b * 3 * pitch;
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float x = X[i], y = X[pitch + i], z = X[2 * pitch + i];
    if (!finite3(x, y, z)) continue;
    int ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
    mn[0] = min(mn[0], ox); mx[0] = max(mx[0], ox);
    mn[1] = min(mn[1], oy); mx[1] = max(mx[1], oy);
    mn[2] = min(mn[2], oz); mx[2] = max(mx[2], oz);
  }
  __shared__ int red[4][6];
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o));
    }
  }
  if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; a++) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    const bool is_min = threadIdx.x < 3;
    int v = red[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = is_min ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
    if (is_min) { if (v != INT_MAX) atomicMin(&mm[b * 6 + threadIdx.x], v); }
    else if (v != INT_MIN) atomicMax(&mm[b * 6 + threadIdx.x], v);
  }
}

// min_b_/max_b_/div_b_/divb_mul_ (voxel_grid_covariance_omp_impl.hpp:75-103)
__global__ void k_griddesc(const int* __restrict__ mm, GridDesc* gd, unsigned* nwords, float leaf, int n_pairs, unsigned recs_per_pair) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_pairs) return;
  GridDesc g;
  memset(&g, 0, sizeof g);
  g.leaf = leaf;
  g.inv_leaf = 1.0f / leaf;                      // pcl::VoxelGrid::setLeafSize
  g.rec_off = (unsigned)b * recs_per_pair;
  if (mm[b * 6] == INT_MAX) {
    g.status = GRID_EMPTY;
  } else {
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(mm[b * 6 + a]); mx[a] = ord2f(mm[b * 6 + 3 + a]); }
    long long d0 = (long long)((mx[0] - mn[0]) * g.inv_leaf) + 1;
    long long d1 = (long long)((mx[1] - mn[1]) * g.inv_leaf) + 1;
    long long d2 = (long long)((mx[2] - mn[2]) * g.inv_leaf) + 1;
    if (d0 * d1 * d2 > (long long)INT_MAX) {
      g.status = GRID_OVERFLOW;                  // impl:79-84: empty grid
    } else {
      for (int a = 0; a < 3; a++) {
        g.min_b[a] = (int)floorf(mn[a] * g.inv_leaf);
        g.max_b[a] = (int)floorf(mx[a] * g.inv_leaf);
        g.div_b[a] = g.max_b[a] - g.min_b[a] + 1;
      }
      long long nc = (long long)g.div_b[0] * g.div_b[1] * g.div_b[2];
      if (nc > MAX_CELLS) {
        g.status = GRID_CAP;
      } else {
        g.mul1 = g.div_b[0];
        g.mul2 = g.div_b[0] * g.div_b[1];
        g.ncells = (int)nc;
        g.nwords = (int)((nc + 63) >> 6) + 1;     // + one all-zero word: the landing cell of out-of-grid probes
      }
    }
  }
  gd[b] = g;
  nwords[b] = (unsigned)g.nwords;
}

__global__ void k_set_word_off(GridDesc* gd, const unsigned* __restrict__ off, int n_pairs, unsigned* max_ncells) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_pairs) { gd[b].word_off = off[b]; atomicMax(max_ncells, (unsigned)gd[b].ncells); }
}

// first pass of applyFilter: cell index per point (impl:218-223)
template <typename KeyT>
__global__ void __launch_bounds__(256) k_keys(const float* __restrict__ tgt, size_t pitch, const int* __restrict__ cnt,
                                               const GridDesc* __restrict__ gd, KeyT* keys, unsigned* vals, int cb) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pitch) return;
  const GridDesc& g = gd[b];
  unsigned cell = (1u << cb) - 1u;               // "not binned": padding, non-finite point or unusable grid
  if ((int)i < cnt[b] && g.status == GRID_OK) {
    const float* X = tgt + (size_t)b * 3 * pitch;
    float x = X[i], y = X[pitch + i], z = X[2 * pitch + i];
    if (finite3(x, y, z)) {
      int i0 = (int)(floorf(x * g.inv_leaf) - (float)g.min_b[0]);
      int i1 = (int)(floorf(y * g.inv_leaf) - (float)g.min_b[1]);
      int i2 = (int)(floorf(z * g.inv_leaf) - (float)g.min_b[2]);
      cell = (unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2);
    }
  }
  keys[(size_t)b * pitch + i] = ((KeyT)b << cb) | (KeyT)cell;
  vals[(size_t)b * pitch + i] = (unsigned)i;
}

// mark cells that hold >= min_points points (impl:297) in the occupancy bitmap
template <typename KeyT>
__global__ void __launch_bounds__(256) k_mark(const KeyT* __restrict__ keys, size_t pitch, const GridDesc* __restrict__ gd,
                                               BitWord* words, int min_points, int cb) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pitch) return;
  const KeyT* K = keys + (size_t)b * pitch;
  const KeyT key = K[i];
  const unsigned cmask = (1u << cb) - 1u;
  const unsigned cell = (unsigned)key & cmask;
  if (cell == cmask) return;
  if (i != 0 && K[i - 1] == key) return;                       // not the head of its segment
  const size_t last = i + (size_t)(min_points > 0 ? min_points - 1 : 0);
  if (last >= pitch || K[last] != key) return;                 // fewer than min_points points
  atomicOr(&words[gd[b].word_off + (cell >> 6)].bits, 1ull << (cell & 63));
}

// exclusive popcount prefix over the bitmap words of each target: voxel id = rank in ascending cell order
__global__ void __launch_bounds__(256) k_rank(GridDesc* gd, BitWord* words) {
  typedef hipcub::BlockScan<unsigned, 256> Scan;
  __shared__ typename Scan::TempStorage tmp;
  const int b = blockIdx.x;
  BitWord* W = words + gd[b].word_off;
  const int nw = gd[b].nwords;
  unsigned base = 0;
  for (int w0 = 0; w0 < nw; w0 += 256) {
    int w = w0 + threadIdx.x;
    unsigned c = (w < nw) ? (unsigned)__popcll(W[w].bits) : 0u, ex, tot;
    Scan(tmp).ExclusiveSum(c, ex, tot);
    if (w < nw) W[w].prefix = base + ex;
    base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) gd[b].n_voxels = (int)base;
}

// where does the point run of searchable leaf `id` start in the sorted order?
template <typename KeyT>
__global__ void __launch_bounds__(256) k_segstart(const KeyT* __restrict__ keys, size_t pitch, const GridDesc* __restrict__ gd,
                                                   const BitWord* __restrict__ words, unsigned* seg_start, int min_points, int cb) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pitch) return;
  const KeyT* K = keys + (size_t)b * pitch;
  const KeyT key = K[i];
  const unsigned cmask = (1u << cb) - 1u;
  const unsigned cell = (unsigned)key & cmask;
  if (cell == cmask) return;
  if (i != 0 && K[i - 1] == key) return;
  const size_t last = i + (size_t)(min_points > 0 ? min_points - 1 : 0);
  if (last >= pitch || K[last] != key) return;
  const GridDesc& g = gd[b];
  const BitWord bw = words[g.word_off + (cell >> 6)];
  const unsigned id = bw.prefix + (unsigned)__popcll(bw.bits & ((1ull << (cell & 63)) - 1ull));
  seg_start[g.rec_off + id] = (unsigned)i;
}

// leaf.mean_ += pt ; leaf.cov_ += pt pt^T (impl:233-237) for every searchable leaf: one WAVE per leaf.
// The wave gathers 64 points of the leaf's run at a time (the radix sort is stable, so the run is in input
// order), parks the nine f64 terms of each point in LDS, and lanes 0..8 -- one per accumulator -- add them
// strictly in input order, which keeps the sums bit-identical to the reference's sequential accumulation.
#define LS_WAVES 4
template <typename KeyT>
__global__ void __launch_bounds__(64 * LS_WAVES) k_leafsum(const float* __restrict__ tgt, size_t pitch,
                                                           const KeyT* __restrict__ keys, const unsigned* __restrict__ vals,
                                                           const GridDesc* __restrict__ gd, const unsigned* __restrict__ seg_start,
                                                           double* sums, int* vox_idx, int* vox_n, int cb) {
  __shared__ double term[LS_WAVES][64][9];
  const int b = blockIdx.y;
  const GridDesc& g = gd[b];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const KeyT* K = keys + (size_t)b * pitch;
  const unsigned* V = vals + (size_t)b * pitch;
  const float* X = tgt + (size_t)b * 3 * pitch;
  for (int id = blockIdx.x * LS_WAVES + wv; id < g.n_voxels; id += gridDim.x * LS_WAVES) {
    const size_t start = seg_start[g.rec_off + id];
    const KeyT key = K[start];
    // accumulator order: S0 S1 S2 C00 C01 C02 C11 C12 C22 ; cov_ is seeded with Identity (voxel_grid_covariance_omp.h:101)
    double acc = (lane == 3 || lane == 6 || lane == 8) ? 1.0 : 0.0;
    int cnt = 0;
    for (size_t j0 = start;; j0 += 64) {
      const size_t j = j0 + lane;
      const bool in = j < pitch && K[j] == key;
      const int m = (int)__popcll(__ballot(in));
      if (in) {
        const unsigned pi = V[j];
        const double x = (double)X[pi], y = (double)X[pitch + pi], z = (double)X[2 * pitch + pi];
        double* t = term[wv][lane];
        t[0] = x; t[1] = y; t[2] = z;
        t[3] = x * x; t[4] = x * y; t[5] = x * z; t[6] = y * y; t[7] = y * z; t[8] = z * z;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 9) for (int l = 0; l < m; l++) acc += term[wv][l][lane];
      __builtin_amdgcn_wave_barrier();
      cnt += m;
      if (m < 64) break;
    }
    if (lane < 9) sums[(size_t)(g.rec_off + id) * 9 + lane] = acc;
    if (lane == 0) {
      vox_idx[g.rec_off + id] = (int)((unsigned)key & ((1u << cb) - 1u));
      vox_n[g.rec_off + id] = cnt;
    }
  }
}

// second pass of applyFilter (impl:282-367; pca impl:364-397): one thread per searchable leaf
__global__ void __launch_bounds__(256) k_voxels(const GridDesc* __restrict__ gd, const double* __restrict__ sums,
                                                 VoxelRec* recs, int* vox_n, double eig_mult, int pca) {
  const int b = blockIdx.y;
  const GridDesc& g = gd[b];
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= g.n_voxels) return;
  const double* A = sums + (size_t)(g.rec_off + id) * 9;
  const double S[3] = {A[0], A[1], A[2]};
  const double C[9] = {A[3], A[4], A[5], A[4], A[6], A[7], A[5], A[7], A[8]};
  const int cnt = vox_n[g.rec_off + id];
  const double dn = (double)cnt;
  double mu[3] = {S[0] / dn, S[1] / dn, S[2] / dn};                              // impl:293
  double cov[9];
  for (int a = 0; a < 3; a++)
    for (int c = 0; c < 3; c++) cov[a * 3 + c] = (C[a * 3 + c] - 2 * (S[a] * mu[c])) / dn + mu[a] * mu[c];   // impl:329
  const double f = (dn - 1.0) / dn;
  for (int a = 0; a < 9; a++) cov[a] *= f;                                       // impl:330
  double ev[3], Vm[9];
  ndtm::eigen_sym3(cov, ev, Vm);                                                 // impl:333-335
  VoxelRec r;
  r.mean[0] = mu[0]; r.mean[1] = mu[1]; r.mean[2] = mu[2];
  for (int a = 0; a < 9; a++) r.icov[a] = 0.f;
  r.weight = 1;
  int n_out = cnt;
  if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) {                                    // impl:337-341
    n_out = -1;
    r.weight = VOX_DEAD;
  } else {
    const double minev = eig_mult * ev[2];                                       // impl:345
    if (ev[0] < minev) {
      ev[0] = minev;
      if (ev[1] < minev) ev[1] = minev;
      double VD[9], Vi[9];
      for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) VD[a * 3 + c] = Vm[a * 3 + c] * ev[c];
      ndtm::mat3_inverse(Vm, Vi);
      ndtm::mat3_mul(VD, Vi, cov);                                               // impl:355
    }
    if (pca) {                                                                   // pca impl:364-397
      double s0 = sqrt(ev[0]), s1 = sqrt(ev[1]), s2 = sqrt(ev[2]);
      double f0 = (s2 - s1) / s2, f1 = (s1 - s0) / s2, f2 = s0 / s2;
      int label = 1;
      double fm = f0;
      if (f1 > fm) { fm = f1; label = 2; }
      if (f2 > fm) { label = 3; }
      double scale = (label == 2) ? 1.25 : ((label == 1) ? 0.75 : 1.0);
      double d2d = scale * sqrt((mu[0] * mu[0] + mu[1] * mu[1]) + mu[2] * mu[2]);
      r.weight = (int)d2d;                                                       // getDimension2d() returns int (pca.h:222-226)
    }
    double ic[9];
    ndtm::mat3_inverse(cov, ic);                                                 // impl:359
    bool bad = false;
    for (int a = 0; a < 9; a++) { if (!isfinite(ic[a])) bad = true; r.icov[a] = (float)ic[a]; }
    if (bad) { n_out = -1; r.weight = VOX_DEAD; }                                // impl:360-364
  }
  recs[g.rec_off + id] = r;
  vox_n[g.rec_off + id] = n_out;
}

// ------------------------------------------------------------------------------------ derivative sweep
// One (point, voxel) evaluation: updateDerivatives (ndt_omp_impl2.hpp:566-619) with the Jacobian /
// Hessian patterns of computePointDerivatives_AngleAxisd (impl2:503-532) folded in (J and Hp are never
// materialised).  f32 ops single, left to right; f64 accumulation.  `w` = weight multiplier of the hit
// (ndt_pca compounding, applied as a suffix product; unused for ndt_omp).
template <bool PCA>
__device__ __forceinline__ void eval_hit(const float u[3], const float r[3], const float C[9],
                                         const double d1, const float d2f, const double w, const bool ok_in, double acc[43]) {
  float y[3];
#pragma unroll
  for (int j = 0; j < 3; j++) y[j] = (u[0] * C[j] + u[1] * C[3 + j]) + u[2] * C[6 + j];
  const float qf = (u[0] * y[0] + u[1] * y[1]) + u[2] * y[2];
  const float e0 = (float)exp((double)((-d2f * qf) * 0.5f));                     // impl2:581
  float s_inc = (float)(-d1 * (double)e0);                                       // impl2:583
  const float e1 = d2f * e0;                                                     // impl2:585
  // impl2:588-589, branch-free: a rejected hit (or an idle lane, ok_in = false) multiplies every term by e = 0 and so
  // adds +0 to all 43 sums (all operands are finite here: dead voxels never enter the queue).
  const bool ok = ok_in && !(e1 > 1.f || e1 < 0.f || e1 != e1);
  float e = (float)((double)e1 * d1);                                            // impl2:592
  e = ok ? e : 0.f;
  s_inc = ok ? s_inc : 0.f;
  // CJ = c_inv4 * point_gradient4 (impl2:594): columns 0..2 are C itself
  float CJ[3][6];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    CJ[a][0] = C[a * 3 + 0]; CJ[a][1] = C[a * 3 + 1]; CJ[a][2] = C[a * 3 + 2];
    CJ[a][3] = C[a * 3 + 1] * (-r[2]) + C[a * 3 + 2] * r[1];
    CJ[a][4] = C[a * 3 + 0] * r[2] + C[a * 3 + 2] * (-r[0]);
    CJ[a][5] = C[a * 3 + 0] * (-r[1]) + C[a * 3 + 1] * r[0];
  }
  float v[6];
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] = (u[0] * CJ[0][k] + u[1] * CJ[1][k]) + u[2] * CJ[2][k];   // impl2:595
  // w * term: the product is a single rounding away from the reference's nested multiplies (both ~1e-16)
#define NDT_ACC(slot, val) do { if (PCA) acc[slot] = fma(w, (double)(val), acc[slot]); else acc[slot] += (double)(val); } while (0)
  NDT_ACC(0, s_inc);
#pragma unroll
  for (int k = 0; k < 6; k++) NDT_ACC(1 + k, e * v[k]);                                        // impl2:597
  // z_i[j] = y * Hp_block_i (impl2:607) -- nine non-zero entries (impl2:522-530)
  float Z[3][3];
  Z[0][0] = y[1] * (-r[1]) + y[2] * (-r[2]);
  Z[1][0] = y[0] * r[1];
  Z[2][0] = y[0] * r[2];
  Z[0][1] = y[1] * r[0];
  Z[1][1] = y[0] * (-r[0]) + y[2] * (-r[2]);
  Z[2][1] = y[1] * r[2];
  Z[0][2] = y[2] * r[0];
  Z[1][2] = y[2] * r[1];
  Z[2][2] = y[0] * (-r[0]) + y[1] * (-r[1]);
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      // JCJ[j][i] = (J^T CJ)(j,i) (impl2:601)
      float jcj;
      if (j < 3) jcj = CJ[j][i];
      else if (j == 3) jcj = (-r[2]) * CJ[1][i] + r[1] * CJ[2][i];
      else if (j == 4) jcj = r[2] * CJ[0][i] + (-r[0]) * CJ[2][i];
      else jcj = (-r[1]) * CJ[0][i] + r[0] * CJ[1][i];
      const float z = (i >= 3 && j >= 3) ? Z[i - 3][j - 3] : 0.f;
      const float h = e * ((((-d2f) * v[i]) * v[j] + z) + jcj);                                // impl2:611-613
      NDT_ACC(7 + i * 6 + j, h);
    }
  }
#undef NDT_ACC
}

// Neighbour offset `a` (0..2) of probe q for a K-probe search, resolved at compile time in the sweep
// (same tables and order as c_off above).
__host__ __device__ constexpr int probe_off(int K, int q, int a) {
  const int o7[7][3] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1}};
  const int o26[26][3] = {{-1,-1,-1}, {-1,0,-1}, {-1,1,-1}, {0,-1,-1}, {0,0,-1}, {0,1,-1}, {1,-1,-1}, {1,0,-1}, {1,1,-1}, {-1,-1,0}, {0,-1,0}, {1,-1,0}, {-1,0,0}, {1,1,1}, {1,0,1}, {1,-1,1}, {0,1,1}, {0,0,1}, {0,-1,1}, {-1,1,1}, {-1,0,1}, {-1,-1,1}, {1,1,0}, {0,1,0}, {-1,1,0}, {1,0,0}};
  return K == 1 ? 0 : (K == 7 ? o7[q][a] : o26[q][a]);
}

#define Q_CAP   512                       // per-wave hit queue (entries); >= 63 + 7*64
#define Q_GROUP 7                         // probes between queue drains
#define ID_BITS 25                        // queue entry = slot << 25 | voxel id
#define WAVES   (SWEEP_THREADS / 64)

// The sweep.  Work decomposition (MI355X-first):
//   block = 4 waves = one CHUNK_PTS chunk of one pair at a time; each wave owns CHUNK_PTS/4 consecutive points.
//   phase A (probe, lane = point): transform the point (f32), probe its K neighbour cells in the rank-bitmap,
//     and push every hit as a 4-byte entry into the wave's LDS queue (ballot + popcount compaction).
//   phase B (evaluate, lane = hit): lanes pull 64 queue entries at a time -- every lane busy no matter how the
//     hits were distributed over points -- read the staged point (LDS) and the 64-B voxel record, and add the
//     43 f64 terms into per-lane accumulators.
//   chunk end: flush the queue tail, wave butterfly + fixed-order wave sum -> one 44-double partial row per chunk.
// The partial rows depend only on (CHUNK_PTS, input order), never on the launch geometry, so single and batched
// runs of one pair are bit-identical.
struct SweepCtl {               // zeroed by the host before every k_update / k_init_state
  int n_active;                 // pairs whose next sweep is pending (entries of active_list)
  int next_item[8];             // per-XCD work-item cursors of the sweep
};
#define QUARTERS WAVES          // a chunk is reduced as 4 wave-quarters of CHUNK_PTS/4 points

template <bool PCA, int K>
__global__ void __launch_bounds__(SWEEP_THREADS, SWEEP_WPE)
k_sweep(const float* __restrict__ src, size_t pitch, const PairState* __restrict__ st,
        const GridDesc* __restrict__ gd, const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs,
        double* partials, int chunks_per_pair, const int* __restrict__ active_list, SweepCtl* ctl, SweepConst sc) {
  // Persistent waves pulling work items.  One item = one wave-quarter (CHUNK_PTS/4 consecutive points) of one chunk
  // of one active pair; every wave is independent (own LDS queue, own partial row, no block barrier), so a wave
  // whose points have few hits simply takes the next item instead of idling at a barrier.
  // Items are queued per XCD: pair slot a of the active list belongs to XCD a % 8 (workgroup L is observed to run on
  // XCD L % 8, MI355X_MICROARCH.md), so one pair's records / bitmap / points stay in one L2; a wave whose XCD
  // queue is empty steals from the others.  Which wave runs an item never changes the item's result.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int n_active = ctl->n_active;
  const int items_per_pair = chunks_per_pair * QUARTERS;

  __shared__ unsigned q_ent[WAVES][Q_CAP];
  __shared__ double q_w[PCA ? WAVES : 1][PCA ? Q_CAP : 1];
  __shared__ float stage[WAVES][128][6];           // two tiles of staged points: x'(3), R x (3)

  if (n_active == 0) return;                       // nothing left to sweep (the loop's last, empty round)
  const int my_xcd = blockIdx.x & 7;
#pragma unroll 1
  for (int probe = 0; probe < 8; probe++) {        // own XCD first, then steal
    const int xcd = (my_xcd + probe) & 7;
    const int pairs_here = n_active > xcd ? (n_active - xcd + 7) / 8 : 0;
    const int items_here = pairs_here * items_per_pair;
    if (items_here == 0) continue;
    // a drained queue is recognised with a plain (L2) load; only a queue that still has items costs an atomic
    if (__hip_atomic_load(&ctl->next_item[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= items_here) continue;
    int item = 0;
    if (lane == 0) item = atomicAdd(&ctl->next_item[xcd], 1);
    item = __builtin_amdgcn_readfirstlane(item);
#pragma unroll 1
    while (item < items_here) {
      // claim the NEXT item now; the atomic's round trip is hidden behind this item's work
      int next_item = 0;
      if (lane == 0) next_item = atomicAdd(&ctl->next_item[xcd], 1);
      const int b = active_list[xcd + 8 * (item / items_per_pair)];
      const int rem = item % items_per_pair;
      const int chunk = rem / QUARTERS, quarter = rem % QUARTERS;

  const PairState& S = st[b];
  const int n = S.n_src;
  const GridDesc& g = gd[b];
  const float* X = src + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const VoxelRec* R = recs + g.rec_off;
  const bool grid_ok = (g.status == GRID_OK);
  float T[12], Rj[9];
#pragma unroll
  for (int a = 0; a < 12; a++) T[a] = S.T[a];
#pragma unroll
  for (int a = 0; a < 9; a++) Rj[a] = S.Rj[a];
  const float leaf = g.leaf;
  const int mb0 = g.min_b[0], mb1 = g.min_b[1], mb2 = g.min_b[2];
  const int xb0 = g.max_b[0], xb1 = g.max_b[1], xb2 = g.max_b[2];
  const int mul1 = g.mul1, mul2 = g.mul2, nwords = g.nwords;
  {
    double acc[43];
#pragma unroll
    for (int a = 0; a < 43; a++) acc[a] = 0.0;
    unsigned nhits = 0;                              // wave-uniform
    int qhead = 0, qcount = 0;                       // wave-uniform
    int q_old = 0;                                   // queued entries that reference the OTHER staging half (older tile)
    const int wbase = chunk * CHUNK_PTS + quarter * (CHUNK_PTS / QUARTERS);

    // evaluate `m` queued hits (m <= 64), one per lane; lanes >= m re-read the last entry and contribute +0
    auto drain = [&](int m) {
      const int k = lane < m ? lane : m - 1;
      const unsigned ent = q_ent[wv][(qhead + k) & (Q_CAP - 1)];
      const unsigned slot = ent >> ID_BITS, id = ent & ((1u << ID_BITS) - 1);
      const float* sp = stage[wv][slot];
      const float xt0 = sp[0], xt1 = sp[1], xt2 = sp[2];
      float r[3] = {sp[3], sp[4], sp[5]};
      const VoxelRec& vr = R[id];
      const double m0 = vr.mean[0], m1 = vr.mean[1], m2 = vr.mean[2];
      float Cf[9];
#pragma unroll
      for (int a = 0; a < 9; a++) Cf[a] = vr.icov[a];
      double w = 1.0;
      if (PCA) w = q_w[wv][(qhead + k) & (Q_CAP - 1)];
      // ndt_omp: leaves with nr_points = -1 (eigen / inverse failure) are not neighbours (impl:395): filtered here
      const bool live = lane < m && (PCA || vr.weight != VOX_DEAD);
      float u[3] = {(float)((double)xt0 - m0), (float)((double)xt1 - m1), (float)((double)xt2 - m2)};   // impl2:276-279, 574
      eval_hit<PCA>(u, r, Cf, sc.d1, sc.d2f, w, live, acc);
      nhits += PCA ? (unsigned)m : (unsigned)__popcll(__ballot(live));
      qhead = (qhead + m) & (Q_CAP - 1);
      qcount -= m;
      q_old = q_old > m ? q_old - m : 0;
    };

    if (wbase < n && grid_ok) {
      // points of the next tile are fetched one tile ahead (HBM latency ~2 us would otherwise be exposed per tile)
      float nx = 0.f, ny = 0.f, nz = 0.f;
      if (wbase + lane < n) { nx = X[wbase + lane]; ny = X[pitch + wbase + lane]; nz = X[2 * pitch + wbase + lane]; }
#pragma unroll 1
      for (int t = 0; t < CHUNK_PTS / WAVES / 64; t++) {
        const int i = wbase + t * 64 + lane;
        if (wbase + t * 64 >= n) break;              // wave-uniform
        // the staging area holds two tiles: entries of tile t-2 must be gone before tile t overwrites their half
        // (only happens when hits are sparse; dense tiles are consumed by the regular 64-wide drains)
        if (q_old > 0) { __builtin_amdgcn_wave_barrier(); drain(q_old); }
        q_old = qcount;
        const int slot = (t & 1) * 64 + lane;
        bool valid = i < n;
        const float px = nx, py = ny, pz = nz;
        if (t + 1 < CHUNK_PTS / WAVES / 64 && i + 64 < n) { nx = X[i + 64]; ny = X[pitch + i + 64]; nz = X[2 * pitch + i + 64]; }
        valid = valid && finite3(px, py, pz);
        // PCL 1.8 transformPointCloud scalar form; Jacobian point r = R x (impl2:507-508)
        float xt[3], r[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          xt[a] = ((T[a * 4 + 0] * px + T[a * 4 + 1] * py) + T[a * 4 + 2] * pz) + T[a * 4 + 3];
          r[a] = (Rj[a * 3 + 0] * px + Rj[a * 3 + 1] * py) + Rj[a * 3 + 2] * pz;
        }
        float* sp = stage[wv][slot];
        sp[0] = xt[0]; sp[1] = xt[1]; sp[2] = xt[2]; sp[3] = r[0]; sp[4] = r[1]; sp[5] = r[2];
        // getNeighborhoodAtPoint (voxel_grid_covariance_omp_impl.hpp:379-399): cell of the point, f32 divide
        // (x / 2^k is the same bits as x * 2^-k, so a power-of-two leaf takes the one-instruction path)
        const int c0 = (int)floorf(sc.leaf_pow2 ? xt[0] * sc.inv_leaf : xt[0] / leaf);
        const int c1 = (int)floorf(sc.leaf_pow2 ? xt[1] * sc.inv_leaf : xt[1] / leaf);
        const int c2 = (int)floorf(sc.leaf_pow2 ? xt[2] * sc.inv_leaf : xt[2] / leaf);
        // Branch-free probe stage.  Relative cell r = c - min_b; "inside the grid" (impl:382-392) is one unsigned
        // compare per axis; a probe that falls outside (or belongs to an invalid lane) is redirected to the grid's
        // spare all-zero bitmap word, so it misses without any flag having to be kept.
        const int r0 = c0 - mb0, r1 = c1 - mb1, r2 = c2 - mb2;
        const unsigned e0 = (unsigned)(xb0 - mb0), e1 = (unsigned)(xb1 - mb1), e2 = (unsigned)(xb2 - mb2);
        const int cc = r0 + r1 * mul1 + r2 * mul2;
        const unsigned empty_cell = (unsigned)(nwords - 1) << 6;
        // probes run last-to-first so the ndt_pca weight of a hit (product of its own and all LATER hits' weights,
        // ndt_pca_impl2.hpp:295-296) is a running product; the order of the f64 additions is free anyway.
        double suf = 1.0;
        // Q_GROUP probes at a time: all bitmap loads of the group in flight together (then all ndt_pca weight loads),
        // then the ballots -- one L2 round trip per stage instead of one per probe.
#pragma unroll
        for (int q1 = K; q1 > 0; q1 -= Q_GROUP) {      // compile-time groups: 1 for DIRECT1 / DIRECT7, 4 for DIRECT26
          unsigned cellv[Q_GROUP];
          uint4 bwv[Q_GROUP];                          // BitWord: bits lo, bits hi, prefix, pad
#pragma unroll
          for (int j = 0; j < Q_GROUP; j++) {
            const int q = q1 - 1 - j;                 // compile-time
            cellv[j] = empty_cell;
            if (q >= 0) {
              const int o0 = probe_off(K, q, 0), o1 = probe_off(K, q, 1), o2 = probe_off(K, q, 2);
              const bool inside = valid && (unsigned)(r0 + o0) <= e0 && (unsigned)(r1 + o1) <= e1 && (unsigned)(r2 + o2) <= e2;
              if (inside) cellv[j] = (unsigned)(cc + o0 + o1 * mul1 + o2 * mul2);
            }
            bwv[j] = *reinterpret_cast<const uint4*>(W + (cellv[j] >> 6));
          }
          unsigned idv[Q_GROUP];
          int wiv[Q_GROUP];
#pragma unroll
          for (int j = 0; j < Q_GROUP; j++) {
            // shift the cell's bit to the top: sign = occupied, popcount = bits at or below it
            const unsigned long long bits = ((unsigned long long)bwv[j].y << 32) | bwv[j].x;
            const unsigned long long t = bits << (63u - (cellv[j] & 63u));
            idv[j] = bwv[j].z + (unsigned)__popcll(t) - 1u;        // rank among the searchable leaves = voxel id
            // occupied <=> the cell's bit (now the sign bit) is set; ndt_pca needs the weights now (suffix product),
            // ndt_omp filters dead leaves in phase B instead and saves this dependent L2 round trip
            wiv[j] = ((long long)t < 0) ? 1 : VOX_DEAD;
            if (PCA) { if ((long long)t < 0) wiv[j] = R[idv[j]].weight; }
          }
#pragma unroll
          for (int j = 0; j < Q_GROUP; j++) {
            const bool hit = wiv[j] != VOX_DEAD;     // empty cell, or nr_points == -1: not a neighbour (impl:395)
            if (PCA && hit) suf *= (double)wiv[j];
            const unsigned long long mask = __ballot(hit);
            if (hit) {
              const int pos = (qhead + qcount + (int)__popcll(mask & lt_mask)) & (Q_CAP - 1);
              q_ent[wv][pos] = ((unsigned)slot << ID_BITS) | idv[j];
              if (PCA) q_w[wv][pos] = suf;
            }
            qcount += (int)__popcll(mask);
          }
          __builtin_amdgcn_wave_barrier();
          while (qcount >= 64) drain(64);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (qcount > 0) drain(qcount);
    }
    // fixed-order reduction of the wave: 64-lane butterfly -> one 44-double row per (chunk, quarter)
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
      P[43] = (double)nhits;
    }
  }
      item = __builtin_amdgcn_readfirstlane(next_item);
    }
  }
}

// ------------------------------------------------------------------------------------ Newton control
__device__ void finalize_pair(PairState& S, mi355ndt_result* res, int converged) {
  S.converged = converged;
  S.phase = PH_DONE;
  S.trans_probability = S.score / (double)S.n_src;                                // impl2:149 / 187
  mi355ndt_result o;
  for (int a = 0; a < 16; a++) o.final_colmajor[a] = S.final_cm[a];
  o.trans_probability = S.trans_probability;
  o.score = S.score;
  o.iterations = S.it;
  o.converged = converged;
  o.sweeps = S.sweeps;
  o.status = (S.grid_status == GRID_OK || S.grid_status == GRID_EMPTY) ? MI355NDT_OK : MI355NDT_ERR_GRID;
  o.hits_last = S.hits;
  *res = o;
}

// p = SE3(R,t).log(); first sweep moves the cloud by the caller's f32 guess itself (impl2:102-129)
__global__ void k_init_state(PairState* st, const float* __restrict__ guess_cm, const int* __restrict__ src_cnt,
                             const GridDesc* __restrict__ gd, int n_pairs, int* active_list, SweepCtl* ctl) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_pairs) return;
  active_list[b] = b;                            // the first sweep covers every pair
  if (b == 0) ctl->n_active = n_pairs;
  PairState& S = st[b];
  const float* G = guess_cm + (size_t)b * 16;
  double R[9], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) { S.T[r * 4 + c] = G[c * 4 + r]; R[r * 3 + c] = (double)G[c * 4 + r]; }
    S.T[r * 4 + 3] = G[12 + r];
    t[r] = (double)G[12 + r];
  }
  for (int a = 0; a < 16; a++) S.final_cm[a] = G[a];
  ndtm::se3_log(ndtm::se3_from_Rt(R, t), S.p);
  float Tdummy[12];
  ndtm::pose_to_f32(S.p, Tdummy, S.Rj);
  S.it = 0; S.phase = PH_SWEEP0; S.converged = 0; S.sweeps = 1; S.a_t = 0; S.hits = 0; S.score = 0;
  S.n_src = src_cnt[b];
  S.grid_status = gd[b].status;
}

// explicit sweep pose (parity hooks)
__global__ void k_set_pose(PairState* st, int b, const float* __restrict__ T_cm, const float* __restrict__ Rj, const int* __restrict__ src_cnt,
                           const GridDesc* __restrict__ gd, int* active_list, SweepCtl* ctl) {
  PairState& S = st[b];
  active_list[0] = b; ctl->n_active = 1;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) S.T[r * 4 + c] = T_cm[c * 4 + r];
  for (int a = 0; a < 9; a++) S.Rj[a] = Rj[a];
  S.phase = PH_SWEEP0; S.n_src = src_cnt[b]; S.grid_status = gd[b].status; S.it = 0; S.sweeps = 1;
}
__global__ void k_set_pose_p(PairState* st, int b, const double* __restrict__ p, const int* __restrict__ src_cnt, const GridDesc* __restrict__ gd,
                             int* active_list, SweepCtl* ctl) {
  PairState& S = st[b];
  active_list[0] = b; ctl->n_active = 1;
  double pp[6];
  for (int a = 0; a < 6; a++) pp[a] = p[a];
  ndtm::pose_to_f32(pp, S.T, S.Rj);
  S.phase = PH_SWEEP0; S.n_src = src_cnt[b]; S.grid_status = gd[b].status; S.it = 0; S.sweeps = 1;
}

// One wave per pair: fixed-order reduction of the chunk partials, then the body of the while loop of
// computeTransformation (impl2:131-183) with computeStepLengthMT's live prefix (impl2:846-907).
__global__ void __launch_bounds__(64)
k_update(PairState* st, const double* __restrict__ partials, int chunks_per_pair, mi355ndt_result* results,
         int* active_counter, int* active_list, SweepCtl* ctl, unsigned long long* hits_total,
         double step_max, double eps, int max_iterations, int reduce_only) {
  const int b = blockIdx.x;
  PairState& S = st[b];
  if (S.phase == PH_DONE) return;
  const int lane = threadIdx.x;
  const int nchunks = (S.n_src + CHUNK_PTS - 1) / CHUNK_PTS;
  if (lane < NACC) {
    double v = 0.0;
    const double* P = partials + (size_t)b * chunks_per_pair * QUARTERS * NACC + lane;
    for (int c = 0; c < nchunks; c++) {                                          // impl2:298-302 (fixed order)
      const double* Q = P + (size_t)c * QUARTERS * NACC;
      v += ((Q[0] + Q[NACC]) + Q[2 * NACC]) + Q[3 * NACC];                       // the chunk's four wave-quarters, in order
    }
    if (lane == 0) S.score = v;
    else if (lane < 7) S.g[lane - 1] = v;
    else if (lane < 43) S.H[lane - 7] = v;
    else { S.hits = (long long)v; if (hits_total) atomicAdd(hits_total, (unsigned long long)v); }
  }
  __syncthreads();
  if (lane != 0 || reduce_only) return;

  const double step_min = eps / 2;
  if (S.phase == PH_STEP) {
    double dp[6], pn[6];
    for (int a = 0; a < 6; a++) dp[a] = S.dir[a] * S.a_t;                        // impl2:156
    ndtm::se3_log(ndtm::se3_mul(ndtm::se3_exp(dp), ndtm::se3_exp(S.p)), pn);     // impl2:166
    for (int a = 0; a < 6; a++) S.p[a] = pn[a];
    const bool conv = (S.it > max_iterations) || (S.it && (fabs(S.a_t) < eps));  // impl2:175-179
    S.it++;
    if (conv) { finalize_pair(S, &results[b], 1); return; }
  }
  for (int guard = 0; guard < 4; guard++) {
    double neg[6], d[6];
    for (int a = 0; a < 6; a++) neg[a] = -S.g[a];
    // impl2:138-140: JacobiSVD(H).solve(-g).  Well-conditioned H: exact LU solve (same answer to rounding);
    // anything else (rank-deficient, H = 0, ill-conditioned): the thresholded pseudo-inverse itself.
    if (!ndtm::lu_solve6(S.H, neg, d)) ndtm::svd_solve6(S.H, neg, d);
    double nrm = 0;
    for (int a = 0; a < 6; a++) nrm += d[a] * d[a];
    nrm = sqrt(nrm);
    if (nrm == 0 || nrm != nrm) { finalize_pair(S, &results[b], nrm == nrm); return; }   // impl2:147-152
    for (int a = 0; a < 6; a++) d[a] /= nrm;                                     // impl2:154
    double dphi0 = 0;
    for (int a = 0; a < 6; a++) dphi0 += S.g[a] * d[a];
    dphi0 = -dphi0;                                                              // impl2:849
    if (dphi0 >= 0 && dphi0 == 0) {
      // impl2:856-857: step length 0, nothing re-evaluated
      double z[6] = {0, 0, 0, 0, 0, 0}, pn[6];
      ndtm::se3_log(ndtm::se3_mul(ndtm::se3_exp(z), ndtm::se3_exp(S.p)), pn);
      for (int a = 0; a < 6; a++) S.p[a] = pn[a];
      const bool conv = (S.it > max_iterations) || (S.it && (0.0 < eps));
      S.it++;
      if (conv) { finalize_pair(S, &results[b], 1); return; }
      continue;
    }
    if (dphi0 >= 0) { for (int a = 0; a < 6; a++) d[a] = -d[a]; }                // impl2:861-862
    double a_t = nrm;
    a_t = a_t < step_max ? a_t : step_max;                                       // impl2:890-892
    a_t = a_t > step_min ? a_t : step_min;
    double xt[6];
    for (int a = 0; a < 6; a++) { S.dir[a] = d[a]; xt[a] = S.p[a] + d[a] * a_t; }   // impl2:894
    S.a_t = a_t;
    ndtm::pose_to_f32(xt, S.T, S.Rj);                                            // impl2:900
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 4; c++) S.final_cm[c * 4 + r] = S.T[r * 4 + c];
      S.final_cm[r * 4 + 3] = 0.f;
    }
    S.final_cm[15] = 1.f;
    S.phase = PH_STEP;
    S.sweeps++;
    atomicAdd(active_counter, 1);
    active_list[atomicAdd(&ctl->n_active, 1)] = b;                               // this pair takes part in the next sweep
    return;
  }
  finalize_pair(S, &results[b], 1);
}

// output cloud of align(): source moved by final_transformation_ (f32)
__global__ void k_transform(const float* __restrict__ src, size_t pitch, const PairState* __restrict__ st, int b, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* X = src + (size_t)b * 3 * pitch;
  const float* F = st[b].final_cm;
  float px = X[i], py = X[pitch + i], pz = X[2 * pitch + i];
  for (int a = 0; a < 3; a++) out[(size_t)a * n + i] = ((F[0 * 4 + a] * px + F[1 * 4 + a] * py) + F[2 * 4 + a] * pz) + F[3 * 4 + a];
}

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
      const int c0 = (int)floorf(q[0] * g.inv_leaf) - g.min_b[0], c1 = (int)floorf(q[1] * g.inv_leaf) - g.min_b[1],
                c2 = (int)floorf(q[2] * g.inv_leaf) - g.min_b[2];
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

// ------------------------------------------------------------------------------------ prefilter (upstream of the path)
// PrefilteringNodelet::distance_filter + downsample (src/lidar_odometry/prefiltering_nodelet.cpp:137-181, parameters of
// launch/dlo_kitti.launch:30-36): keep points with near < |p| < far (f32 norm compared as double), then pcl::VoxelGrid
// centroid downsample (PCL 1.8 voxel_grid.hpp applyFilter, CentroidPoint / AccumulatorXYZ: f32 sums, divided by the
// count), output in ascending voxel index.  Same binning + stable sort machinery as the NDT target build.
__global__ void __launch_bounds__(256) k_pf_flag(const float* __restrict__ X, size_t pitch, int n, int use_df, double dnear, double dfar,
                                                 unsigned char* keep, int* mm) {
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = X[i], y = X[pitch + i], z = X[2 * pitch + i];
    bool ok = true;
    if (use_df) {
      const double d = (double)sqrtf((x * x + y * y) + z * z);       // p.getVector3fMap().norm() (:168)
      ok = d > dnear && d < dfar;                                    // NaN fails both compares
    }
    ok = ok && finite3(x, y, z);                                     // VoxelGrid skips non-finite points (is_dense = false, :175)
    keep[i] = ok ? 1 : 0;
    if (!ok) continue;
    int ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
    mn[0] = min(mn[0], ox); mx[0] = max(mx[0], ox);
    mn[1] = min(mn[1], oy); mx[1] = max(mx[1], oy);
    mn[2] = min(mn[2], oz); mx[2] = max(mx[2], oz);
  }
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], o)); mx[a] = max(mx[a], __shfl_xor(mx[a], o)); }
    if ((threadIdx.x & 63) == 0) {
      if (mn[a] != INT_MAX) atomicMin(&mm[a], mn[a]);
      if (mx[a] != INT_MIN) atomicMax(&mm[3 + a], mx[a]);
    }
  }
}

struct PfGrid { int min_b[3], mul1, mul2, status; float inv_leaf; };   // status: 0 ok, 1 empty, 2 index overflow

__global__ void k_pf_grid(const int* __restrict__ mm, float leaf, PfGrid* out) {
  PfGrid g;
  memset(&g, 0, sizeof g);
  g.inv_leaf = 1.0f / leaf;
  if (mm[0] == INT_MAX) { g.status = 1; *out = g; return; }
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = ord2f(mm[a]); mx[a] = ord2f(mm[3 + a]); }
  const long long d0 = (long long)((mx[0] - mn[0]) * g.inv_leaf) + 1, d1 = (long long)((mx[1] - mn[1]) * g.inv_leaf) + 1,
                  d2 = (long long)((mx[2] - mn[2]) * g.inv_leaf) + 1;
  if (d0 * d1 * d2 > (long long)INT_MAX) { g.status = 2; *out = g; return; }      // "Leaf size is too small": output = input
  int maxb[3];
  for (int a = 0; a < 3; a++) { g.min_b[a] = (int)floorf(mn[a] * g.inv_leaf); maxb[a] = (int)floorf(mx[a] * g.inv_leaf); }
  g.mul1 = maxb[0] - g.min_b[0] + 1;
  g.mul2 = g.mul1 * (maxb[1] - g.min_b[1] + 1);
  *out = g;
}

__global__ void __launch_bounds__(256) k_pf_keys(const float* __restrict__ X, size_t pitch, int n, const unsigned char* __restrict__ keep,
                                                 const PfGrid* __restrict__ pg, unsigned* keys, unsigned* vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)pitch) return;
  const PfGrid g = *pg;
  unsigned cell = 0x7FFFFFFFu;
  if (i < n && keep[i] && g.status == 0) {
    const int i0 = (int)(floorf(X[i] * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(X[pitch + i] * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(X[2 * pitch + i] * g.inv_leaf) - (float)g.min_b[2]);
    cell = (unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2);
  }
  keys[i] = cell;
  vals[i] = (unsigned)i;
}

// head of every occupied voxel's run (or, without down-sampling, every kept point)
__global__ void __launch_bounds__(256) k_pf_heads(const unsigned* __restrict__ keys, const unsigned char* __restrict__ keep, int n, size_t pitch,
                                                  int downsample, int* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)pitch) return;
  int f;
  if (downsample) f = keys[i] != 0x7FFFFFFFu && (i == 0 || keys[i - 1] != keys[i]);
  else f = i < n && keep[i];
  flag[i] = f;
}

__global__ void __launch_bounds__(256) k_pf_emit(const float* __restrict__ X, size_t pitch, const unsigned* __restrict__ keys,
                                                 const unsigned* __restrict__ vals, const int* __restrict__ flag, const int* __restrict__ pos,
                                                 int downsample, float* out, size_t out_pitch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)pitch || !flag[i]) return;
  float sx, sy, sz;
  if (downsample) {
    const unsigned key = keys[i];
    sx = sy = sz = 0.f;
    int cnt = 0;
    for (size_t j = i; j < pitch && keys[j] == key; j++) {           // AccumulatorXYZ: xyz += p (f32), input order
      const unsigned pi = vals[j];
      sx += X[pi]; sy += X[pitch + pi]; sz += X[2 * pitch + pi];
      cnt++;
    }
    const float fn = (float)cnt;
    sx /= fn; sy /= fn; sz /= fn;                                    // xyz / n
  } else {
    sx = X[i]; sy = X[pitch + i]; sz = X[2 * pitch + i];
  }
  const int o = pos[i];
  out[o] = sx; out[out_pitch + o] = sy; out[2 * out_pitch + o] = sz;
}

// ------------------------------------------------------------------------------------ host side
struct mi355ndt_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  mi355ndt_params prm;
  std::string err;

  int n_pairs = 0, cap_pairs = 0;
  size_t tgt_pitch = 0, src_pitch = 0;          // geometry in use
  size_t own_tgt_pitch = 0, own_src_pitch = 0;  // geometry of the owned buffers
  int own_tgt_pairs = 0, own_src_pairs = 0;
  float *d_tgt_own = nullptr, *d_src_own = nullptr;
  const float *d_tgt = nullptr, *d_src = nullptr;
  int *d_tgt_cnt = nullptr, *d_src_cnt = nullptr;
  std::vector<int> h_tgt_cnt, h_src_cnt;
  bool targets_built = false, have_target = false, have_source = false;

  // build workspace
  int* d_minmax = nullptr;
  GridDesc* d_grid = nullptr;
  unsigned *d_nwords = nullptr, *d_word_off = nullptr;
  unsigned long long *d_keys_a = nullptr, *d_keys_b = nullptr;
  unsigned *d_vals_a = nullptr, *d_vals_b = nullptr;
  void* d_tmp = nullptr; size_t tmp_bytes = 0;
  size_t keys_cap = 0;
  BitWord* d_words = nullptr; size_t words_cap = 0;
  VoxelRec* d_recs = nullptr; int *d_vox_idx = nullptr, *d_vox_n = nullptr;
  unsigned* d_seg_start = nullptr; double* d_sums = nullptr;
  unsigned *d_cstart = nullptr, *d_cend = nullptr; size_t cell_cap = 0; bool cells_ready = false; int last_cb = 0;
  double* d_fit = nullptr; size_t fit_cap = 0;
  // prefilter workspace
  float *d_pf_in = nullptr, *d_pf_out = nullptr; unsigned char* d_pf_keep = nullptr; unsigned *d_pf_keys = nullptr, *d_pf_vals = nullptr;
  int *d_pf_flag = nullptr, *d_pf_pos = nullptr, *d_pf_mm = nullptr; PfGrid* d_pf_grid = nullptr; void* d_pf_tmp = nullptr;
  size_t pf_cap = 0, pf_tmp_bytes = 0; int pf_count = 0; size_t pf_pitch = 0;
  float last_final[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
  size_t recs_per_pair = 0, recs_cap = 0;
  unsigned* h_pin_u = nullptr;                  // pinned scratch (2 unsigned)

  // align workspace
  PairState* d_state = nullptr;
  double* d_partials = nullptr; size_t partials_cap = 0;
  int chunks_per_pair = 0;
  float* d_guess = nullptr;
  mi355ndt_result* d_results = nullptr;
  int* d_active = nullptr;                      // per-round active counters
  int* d_active_list = nullptr;                 // pairs taking part in the next sweep (compacted by k_update)
  SweepCtl* d_ctl = nullptr;
  int n_cu = 256;
  int* h_pin_active = nullptr;
  unsigned long long* d_hits = nullptr;         // (point,voxel) evaluations, all sweeps
  float* d_hook = nullptr;                      // 16 + 9 floats, 6 doubles
  float* d_aligned = nullptr; size_t aligned_cap = 0;
  std::vector<float> h_stage;

  // profiling
  bool prof = false;
  mi355ndt_profile P{};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_sweep, ev_update, ev_build;
};

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return MI355NDT_ERR_HIP;                                                                   \
    }                                                                                            \
  } while (0)

static int ceil_log2(unsigned v) { int b = 0; while ((1u << b) < v) b++; return b; }

template <typename T>
static hipError_t grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; }
  hipError_t e = hipMalloc((void**)&p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  else cap = 0;
  return e;
}

static void build_offsets(int mode, SweepConst& sc) {
  if (mode == MI355NDT_DIRECT1) { sc.K = 1; sc.table = 0; }
  else if (mode == MI355NDT_DIRECT7) { sc.K = 7; sc.table = 1; }
  else { sc.K = 26; sc.table = 2; }
}

static void gauss_constants(const mi355ndt_params& p, double& d1, double& d2) {
  // ndt_omp_impl2.hpp:93-100
  double c1 = 10 * (1 - p.outlier_ratio);
  double c2 = p.outlier_ratio / pow((double)p.resolution, 3);
  double d3 = -log(c2);
  d1 = -log(c1 + c2) - d3;
  d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / d1);
}

static int check_params(const mi355ndt_params& p) {
  if (!(p.resolution > 0) || !std::isfinite(p.resolution)) return MI355NDT_ERR_BAD_ARG;
  if (p.neighbor_mode < 0 || p.neighbor_mode > 3) return MI355NDT_ERR_BAD_ARG;
  if (p.variant < 0 || p.variant > 1) return MI355NDT_ERR_BAD_ARG;
  if (p.min_points_per_voxel < 1) return MI355NDT_ERR_BAD_ARG;
  if (p.max_iterations < 0) return MI355NDT_ERR_BAD_ARG;
  return MI355NDT_OK;
}

extern "C" {

const char* mi355ndt_version(void) { return "mi355ndt 0.1 (gfx950)"; }

int mi355ndt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mi355ndt_default_params(mi355ndt_params* p) {
  if (!p) return MI355NDT_ERR_BAD_ARG;
  p->resolution = 1.0f;
  p->step_size = 0.1;
  p->outlier_ratio = 0.55;
  p->trans_epsilon = 0.1;
  p->max_iterations = 35;
  p->neighbor_mode = MI355NDT_DIRECT7;
  p->variant = MI355NDT_VARIANT_OMP;
  p->min_points_per_voxel = 6;
  p->min_covar_eigvalue_mult = 0.01;
  return MI355NDT_OK;
}

int mi355ndt_create(const mi355ndt_params* params, int device, mi355ndt_handle** out) {
  if (!out) return MI355NDT_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MI355NDT_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return MI355NDT_ERR_BAD_ARG;
  mi355ndt_params p;
  mi355ndt_default_params(&p);
  if (params) p = *params;
  int rc = check_params(p);
  if (rc) return rc;
  mi355ndt_handle* h = new mi355ndt_handle();
  h->device = device;
  h->prm = p;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) h->n_cu = pr.multiProcessorCount; }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return MI355NDT_ERR_HIP;
  }
  if (hipHostMalloc((void**)&h->h_pin_u, 4 * sizeof(unsigned)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_pin_active, 128 * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&h->d_active, 128 * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&h->d_ctl, sizeof(SweepCtl)) != hipSuccess ||
      hipMalloc((void**)&h->d_hits, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&h->d_hook, 64 * sizeof(double)) != hipSuccess) {
    delete h;
    return MI355NDT_ERR_HIP;
  }
  *out = h;
  return MI355NDT_OK;
}

int mi355ndt_destroy(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->d_tgt_own, h->d_src_own, h->d_tgt_cnt, h->d_src_cnt, h->d_minmax, h->d_grid, h->d_nwords, h->d_word_off,
                  h->d_keys_a, h->d_keys_b, h->d_vals_a, h->d_vals_b, h->d_tmp, h->d_words, h->d_recs, h->d_vox_idx, h->d_vox_n,
                  h->d_state, h->d_partials, h->d_guess, h->d_results, h->d_active, h->d_hook, h->d_aligned, h->d_hits, h->d_seg_start, h->d_sums,
                  h->d_active_list, h->d_ctl, h->d_cstart, h->d_cend, h->d_fit, h->d_pf_in, h->d_pf_out, h->d_pf_keep, h->d_pf_keys,
                  h->d_pf_vals, h->d_pf_flag, h->d_pf_pos, h->d_pf_mm, h->d_pf_grid, h->d_pf_tmp};
  for (void* p : ptrs) if (p) hipFree(p);
  if (h->h_pin_u) hipHostFree(h->h_pin_u);
  if (h->h_pin_active) hipHostFree(h->h_pin_active);
  for (auto& e : h->ev_sweep) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  for (auto& e : h->ev_update) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  for (auto& e : h->ev_build) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
  return MI355NDT_OK;
}

int mi355ndt_get_params(const mi355ndt_handle* h, mi355ndt_params* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!out) return MI355NDT_ERR_BAD_ARG;
  *out = h->prm;
  return MI355NDT_OK;
}

int mi355ndt_set_stream(mi355ndt_handle* h, void* s) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (s) {
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s;
    h->own_stream = false;
  } else if (!h->own_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return MI355NDT_OK;
}

const char* mi355ndt_last_error(const mi355ndt_handle* h) { return h ? h->err.c_str() : "bad handle"; }

int mi355ndt_synchronize(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MI355NDT_OK;
}

int mi355ndt_batch_size(const mi355ndt_handle* h) { return h ? h->n_pairs : MI355NDT_ERR_BAD_HANDLE; }

// ---- capacity management ----------------------------------------------------------------------
static int ensure_pair_arrays(mi355ndt_handle* h, int n_pairs) {
  if (n_pairs <= h->cap_pairs) return MI355NDT_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  auto re = [&](void** p, size_t bytes) -> hipError_t {
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; if (e != hipSuccess) return e; }
    return hipMalloc(p, bytes);
  };
  HIPCHK(h, re((void**)&h->d_tgt_cnt, n_pairs * sizeof(int)));
  HIPCHK(h, re((void**)&h->d_src_cnt, n_pairs * sizeof(int)));
  HIPCHK(h, re((void**)&h->d_minmax, n_pairs * 6 * sizeof(int)));
  HIPCHK(h, re((void**)&h->d_grid, n_pairs * sizeof(GridDesc)));
  HIPCHK(h, re((void**)&h->d_nwords, (n_pairs + 2) * sizeof(unsigned)));
  HIPCHK(h, re((void**)&h->d_word_off, (n_pairs + 1) * sizeof(unsigned)));
  HIPCHK(h, re((void**)&h->d_state, n_pairs * sizeof(PairState)));
  HIPCHK(h, re((void**)&h->d_guess, n_pairs * 16 * sizeof(float)));
  HIPCHK(h, re((void**)&h->d_results, n_pairs * sizeof(mi355ndt_result)));
  HIPCHK(h, re((void**)&h->d_active_list, n_pairs * sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->d_grid, 0, n_pairs * sizeof(GridDesc), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_state, 0, n_pairs * sizeof(PairState), h->stream));
  h->cap_pairs = n_pairs;
  h->h_tgt_cnt.assign(n_pairs, 0);
  h->h_src_cnt.assign(n_pairs, 0);
  return MI355NDT_OK;
}

static int alloc_side(mi355ndt_handle* h, bool tgt, int n_pairs, size_t pitch) {
  float*& buf = tgt ? h->d_tgt_own : h->d_src_own;
  size_t& own_pitch = tgt ? h->own_tgt_pitch : h->own_src_pitch;
  int& own_pairs = tgt ? h->own_tgt_pairs : h->own_src_pairs;
  if (buf && own_pitch == pitch && own_pairs == n_pairs) return MI355NDT_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (buf) { HIPCHK(h, hipFree(buf)); buf = nullptr; }
  HIPCHK(h, hipMalloc((void**)&buf, (size_t)n_pairs * 3 * pitch * sizeof(float)));
  own_pitch = pitch; own_pairs = n_pairs;
  std::vector<int>& cnt = tgt ? h->h_tgt_cnt : h->h_src_cnt;
  std::fill(cnt.begin(), cnt.end(), 0);
  if (tgt) { h->targets_built = false; h->have_target = false; } else { h->have_source = false; }
  return MI355NDT_OK;
}

int mi355ndt_batch_reserve(mi355ndt_handle* h, int n_pairs, size_t max_tgt, size_t max_src) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (n_pairs <= 0 || max_tgt == 0 || max_src == 0 || n_pairs > (1 << 20)) return MI355NDT_ERR_BAD_ARG;
  if (max_tgt >= (1u << 31) || max_src >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  // pitches padded to 64 floats so every row starts 256-B aligned
  size_t tp = (max_tgt + 63) & ~(size_t)63, sp = (max_src + 63) & ~(size_t)63;
  int rc = ensure_pair_arrays(h, n_pairs);
  if (rc) return rc;
  if (n_pairs != h->n_pairs || h->d_tgt != h->d_tgt_own || h->d_src != h->d_src_own) {
    std::fill(h->h_tgt_cnt.begin(), h->h_tgt_cnt.end(), 0);
    std::fill(h->h_src_cnt.begin(), h->h_src_cnt.end(), 0);
    h->targets_built = false; h->have_target = false; h->have_source = false;
  }
  rc = alloc_side(h, true, n_pairs, tp);
  if (rc) return rc;
  rc = alloc_side(h, false, n_pairs, sp);
  if (rc) return rc;
  h->n_pairs = n_pairs;
  h->tgt_pitch = tp; h->src_pitch = sp;
  h->d_tgt = h->d_tgt_own; h->d_src = h->d_src_own;
  return MI355NDT_OK;
}

static int upload_cloud(mi355ndt_handle* h, float* d_base, size_t pitch, int pair, const void* pts, size_t n, size_t stride) {
  if (!pts && n) return MI355NDT_ERR_BAD_ARG;
  if ((n && stride < 12) || n > pitch) return MI355NDT_ERR_BAD_ARG;
  h->h_stage.resize(3 * pitch);
  const unsigned char* p = (const unsigned char*)pts;
  float* sx = h->h_stage.data(); float* sy = sx + pitch; float* sz = sy + pitch;
  for (size_t i = 0; i < n; i++) {
    float v[3];
    memcpy(v, p + i * stride, 12);
    sx[i] = v[0]; sy[i] = v[1]; sz[i] = v[2];
  }
  for (size_t i = n; i < pitch; i++) sx[i] = sy[i] = sz[i] = 0.f;
  HIPCHK(h, hipMemcpyAsync(d_base + (size_t)pair * 3 * pitch, h->h_stage.data(), 3 * pitch * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));   // staging buffer is reused
  return MI355NDT_OK;
}

int mi355ndt_batch_set_target(mi355ndt_handle* h, int pair, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs || h->d_tgt != h->d_tgt_own) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = upload_cloud(h, h->d_tgt_own, h->tgt_pitch, pair, pts, n, stride);
  if (rc) return rc;
  h->h_tgt_cnt[pair] = (int)n;
  h->targets_built = false;
  h->have_target = true;
  return MI355NDT_OK;
}

int mi355ndt_batch_set_source(mi355ndt_handle* h, int pair, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs || h->d_src != h->d_src_own) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = upload_cloud(h, h->d_src_own, h->src_pitch, pair, pts, n, stride);
  if (rc) return rc;
  h->h_src_cnt[pair] = (int)n;
  h->have_source = true;
  return MI355NDT_OK;
}

int mi355ndt_batch_bind_device(mi355ndt_handle* h, int n_pairs, const float* d_t, const int* tc, size_t tp,
                               const float* d_s, const int* scnt, size_t sp) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (n_pairs <= 0 || !d_t || !d_s || !tc || !scnt || tp == 0 || sp == 0 || n_pairs > (1 << 20)) return MI355NDT_ERR_BAD_ARG;
  if (tp >= (1u << 31) || sp >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  for (int b = 0; b < n_pairs; b++) if (tc[b] < 0 || (size_t)tc[b] > tp || scnt[b] < 0 || (size_t)scnt[b] > sp) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_pair_arrays(h, n_pairs);
  if (rc) return rc;
  h->n_pairs = n_pairs;
  h->d_tgt = d_t; h->d_src = d_s;
  h->tgt_pitch = tp; h->src_pitch = sp;
  for (int b = 0; b < n_pairs; b++) { h->h_tgt_cnt[b] = tc[b]; h->h_src_cnt[b] = scnt[b]; }
  h->targets_built = false;
  h->have_target = h->have_source = true;
  return MI355NDT_OK;
}

// ---- profiling helpers ------------------------------------------------------------------------
static hipError_t ev_begin(mi355ndt_handle* h, std::vector<std::pair<hipEvent_t, hipEvent_t>>& v) {
  hipEvent_t a, b;
  hipError_t e = hipEventCreate(&a); if (e != hipSuccess) return e;
  e = hipEventCreate(&b); if (e != hipSuccess) return e;
  v.push_back({a, b});
  return hipEventRecord(a, h->stream);
}
static hipError_t ev_end(mi355ndt_handle* h, std::vector<std::pair<hipEvent_t, hipEvent_t>>& v) {
  return hipEventRecord(v.back().second, h->stream);
}
static void ev_collect(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double& ms, long long& n) {
  for (auto& e : v) {
    float t = 0;
    if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) { ms += t; n++; }
    hipEventDestroy(e.first); hipEventDestroy(e.second);
  }
  v.clear();
}

int mi355ndt_profile_enable(mi355ndt_handle* h, int on) { if (!h) return MI355NDT_ERR_BAD_HANDLE; h->prof = on != 0; return MI355NDT_OK; }
int mi355ndt_profile_reset(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  double d; long long n;
  ev_collect(h->ev_sweep, d, n); ev_collect(h->ev_update, d, n); ev_collect(h->ev_build, d, n);
  h->P = mi355ndt_profile{};
  return MI355NDT_OK;
}
int mi355ndt_profile_get(mi355ndt_handle* h, mi355ndt_profile* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!out) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ev_collect(h->ev_sweep, h->P.sweep_ms, h->P.sweep_launches);
  ev_collect(h->ev_update, h->P.update_ms, h->P.update_launches);
  ev_collect(h->ev_build, h->P.build_ms, h->P.build_launches);
  *out = h->P;
  return MI355NDT_OK;
}

// ---- target build -----------------------------------------------------------------------------
int mi355ndt_batch_build_targets(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (h->n_pairs <= 0 || !h->d_tgt) return MI355NDT_ERR_STATE;
  if (h->prm.neighbor_mode == MI355NDT_KDTREE) return MI355NDT_ERR_UNSUPPORTED;
  HIPCHK(h, hipSetDevice(h->device));
  const int B = h->n_pairs;
  const size_t pitch = h->tgt_pitch;
  const size_t total = (size_t)B * pitch;
  hipStream_t s = h->stream;
  HIPCHK(h, hipMemcpyAsync(h->d_tgt_cnt, h->h_tgt_cnt.data(), B * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipStreamSynchronize(s));   // h_tgt_cnt is pageable

  // workspace
  if (total > h->keys_cap) {
    size_t c1 = h->keys_cap, c2 = h->keys_cap, c3 = h->keys_cap, c4 = h->keys_cap;
    HIPCHK(h, grow(h->d_keys_a, c1, total)); HIPCHK(h, grow(h->d_keys_b, c2, total));
    HIPCHK(h, grow(h->d_vals_a, c3, total)); HIPCHK(h, grow(h->d_vals_b, c4, total));
    h->keys_cap = total;
  }
  const int minpts = h->prm.min_points_per_voxel;
  const size_t rpp = pitch / (size_t)minpts + 1;
  if (rpp > ((size_t)1 << ID_BITS)) { h->err = "target too large: voxel ids would not fit the sweep's 25-bit queue entries"; return MI355NDT_ERR_BAD_ARG; }
  if ((size_t)B * rpp > h->recs_cap || rpp != h->recs_per_pair) {
    size_t need = (size_t)B * rpp, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (h->d_recs) { HIPCHK(h, hipFree(h->d_recs)); h->d_recs = nullptr; }
    if (h->d_vox_idx) { HIPCHK(h, hipFree(h->d_vox_idx)); h->d_vox_idx = nullptr; }
    if (h->d_vox_n) { HIPCHK(h, hipFree(h->d_vox_n)); h->d_vox_n = nullptr; }
    if (h->d_seg_start) { HIPCHK(h, hipFree(h->d_seg_start)); h->d_seg_start = nullptr; }
    if (h->d_sums) { HIPCHK(h, hipFree(h->d_sums)); h->d_sums = nullptr; }
    HIPCHK(h, grow(h->d_recs, c1, need)); HIPCHK(h, grow(h->d_vox_idx, c2, need)); HIPCHK(h, grow(h->d_vox_n, c3, need));
    HIPCHK(h, grow(h->d_seg_start, c4, need)); HIPCHK(h, grow(h->d_sums, c5, need * 9));
    h->recs_cap = need; h->recs_per_pair = rpp;
  }
  const int pb = ceil_log2((unsigned)B);
  size_t need_tmp = 0, need_tmp32 = 0, need_scan = 0;
  HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(nullptr, need_tmp, h->d_keys_a, h->d_keys_b, h->d_vals_a, h->d_vals_b, (int)total, 0, 64, s));
  HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(nullptr, need_tmp32, (unsigned*)h->d_keys_a, (unsigned*)h->d_keys_b, h->d_vals_a, h->d_vals_b, (int)total, 0, 32, s));
  HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, h->d_nwords, h->d_word_off, B + 1, s));
  need_tmp = std::max(std::max(need_tmp, need_tmp32), need_scan);
  if (need_tmp > h->tmp_bytes) {
    if (h->d_tmp) { HIPCHK(h, hipFree(h->d_tmp)); h->d_tmp = nullptr; }
    HIPCHK(h, hipMalloc(&h->d_tmp, need_tmp));
    h->tmp_bytes = need_tmp;
  }
  if (total >= ((size_t)1 << 31)) { h->err = "batch too large for one radix sort"; return MI355NDT_ERR_BAD_ARG; }

  if (h->prof) HIPCHK(h, ev_begin(h, h->ev_build));
  const int gx = (int)((pitch + 255) / 256);
  k_minmax_init<<<(B * 6 + 255) / 256, 256, 0, s>>>(h->d_minmax, B);
  k_minmax<<<dim3(std::min(gx, 16), B), 256, 0, s>>>(h->d_tgt, pitch, h->d_tgt_cnt, h->d_minmax);
  HIPCHK(h, hipMemsetAsync(h->d_nwords, 0, (B + 2) * sizeof(unsigned), s));
  k_griddesc<<<(B + 63) / 64, 64, 0, s>>>(h->d_minmax, h->d_grid, h->d_nwords, h->prm.resolution, B, (unsigned)rpp);
  size_t tb = h->tmp_bytes;
  HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(h->d_tmp, tb, h->d_nwords, h->d_word_off, B + 1, s));
  k_set_word_off<<<(B + 63) / 64, 64, 0, s>>>(h->d_grid, h->d_word_off, B, h->d_nwords + B + 1);
  HIPCHK(h, hipMemcpyAsync(h->h_pin_u, h->d_word_off + B, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(h->h_pin_u + 1, h->d_nwords + B + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));           // total bitmap words -> size the pool; largest grid -> key width
  const size_t total_words = h->h_pin_u[0];
  const int cb = std::max(1, ceil_log2(h->h_pin_u[1] + 1u));   // cell field: every cell index + the all-ones "not binned" value
  const bool k32 = cb + pb <= 32;
  if (total_words > h->words_cap) {
    size_t c = h->words_cap;
    HIPCHK(h, grow(h->d_words, c, std::max(total_words, (size_t)1024)));
    h->words_cap = c;
  }
  if (total_words) HIPCHK(h, hipMemsetAsync(h->d_words, 0, total_words * sizeof(BitWord), s));
  const int lb = std::max(1, std::min((int)((rpp + LS_WAVES - 1) / LS_WAVES), std::max(8, 8192 / B)));
  tb = h->tmp_bytes;
  if (k32) {
    unsigned *ka = (unsigned*)h->d_keys_a, *kb = (unsigned*)h->d_keys_b;
    k_keys<unsigned><<<dim3(gx, B), 256, 0, s>>>(h->d_tgt, pitch, h->d_tgt_cnt, h->d_grid, ka, h->d_vals_a, cb);
    HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(h->d_tmp, tb, ka, kb, h->d_vals_a, h->d_vals_b, (int)total, 0, cb + pb, s));
    k_mark<unsigned><<<dim3(gx, B), 256, 0, s>>>(kb, pitch, h->d_grid, h->d_words, minpts, cb);
    k_rank<<<B, 256, 0, s>>>(h->d_grid, h->d_words);
    k_segstart<unsigned><<<dim3(gx, B), 256, 0, s>>>(kb, pitch, h->d_grid, h->d_words, h->d_seg_start, minpts, cb);
    k_leafsum<unsigned><<<dim3(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, kb, h->d_vals_b, h->d_grid, h->d_seg_start,
                                                             h->d_sums, h->d_vox_idx, h->d_vox_n, cb);
  } else {
    typedef unsigned long long u64;
    k_keys<u64><<<dim3(gx, B), 256, 0, s>>>(h->d_tgt, pitch, h->d_tgt_cnt, h->d_grid, h->d_keys_a, h->d_vals_a, cb);
    HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(h->d_tmp, tb, h->d_keys_a, h->d_keys_b, h->d_vals_a, h->d_vals_b, (int)total, 0, cb + pb, s));
    k_mark<u64><<<dim3(gx, B), 256, 0, s>>>(h->d_keys_b, pitch, h->d_grid, h->d_words, minpts, cb);
    k_rank<<<B, 256, 0, s>>>(h->d_grid, h->d_words);
    k_segstart<u64><<<dim3(gx, B), 256, 0, s>>>(h->d_keys_b, pitch, h->d_grid, h->d_words, h->d_seg_start, minpts, cb);
    k_leafsum<u64><<<dim3(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, h->d_keys_b, h->d_vals_b, h->d_grid, h->d_seg_start,
                                                        h->d_sums, h->d_vox_idx, h->d_vox_n, cb);
  }
  k_voxels<<<dim3((unsigned)((rpp + 255) / 256), B), 256, 0, s>>>(h->d_grid, h->d_sums, h->d_recs, h->d_vox_n,
                                                                  h->prm.min_covar_eigvalue_mult, h->prm.variant == MI355NDT_VARIANT_PCA);
  HIPCHK(h, hipGetLastError());
  if (h->prof) {
    HIPCHK(h, ev_end(h, h->ev_build));
    double pts = 0;
    for (int b = 0; b < B; b++) pts += h->h_tgt_cnt[b];
    // B_build: minmax 12 + binning 12 + key write 12 + sort r/w + grouped gather 16 per point (+ records)
    h->P.build_alg_bytes += pts * (12 + 12 + 4 + 16);
  }
  h->targets_built = true;
  h->cells_ready = false;
  h->last_cb = cb;
  return MI355NDT_OK;
}

// ---- sweeps -----------------------------------------------------------------------------------
static int prep_align_ws(mi355ndt_handle* h) {
  const int B = h->n_pairs;
  int maxn = 0;
  for (int b = 0; b < B; b++) maxn = std::max(maxn, h->h_src_cnt[b]);
  h->chunks_per_pair = std::max(1, (maxn + CHUNK_PTS - 1) / CHUNK_PTS);
  size_t need = (size_t)B * h->chunks_per_pair * QUARTERS * NACC;
  HIPCHK(h, grow(h->d_partials, h->partials_cap, need));
  HIPCHK(h, hipMemcpyAsync(h->d_src_cnt, h->h_src_cnt.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MI355NDT_OK;
}

static void make_sweep_const(const mi355ndt_handle* h, SweepConst& sc) {
  double d1, d2;
  gauss_constants(h->prm, d1, d2);
  sc.d1 = d1;
  sc.d2f = (float)d2;                        // impl2:578
  sc.pca = h->prm.variant == MI355NDT_VARIANT_PCA;
  { int ex; float mant = std::frexp(h->prm.resolution, &ex); sc.leaf_pow2 = (mant == 0.5f) && ex > -100 && ex < 100; sc.inv_leaf = 1.0f / h->prm.resolution; }
  build_offsets(h->prm.neighbor_mode, sc);
}

static int launch_sweep(mi355ndt_handle* h, const SweepConst& sc) {
  // persistent waves: SWEEP_WPE workgroups per CU pull (pair, chunk, quarter) items until the per-XCD queues are dry
  const dim3 grid((unsigned)(h->n_cu * SWEEP_WPE));
  if (h->prof) HIPCHK(h, ev_begin(h, h->ev_sweep));
#define NDT_LAUNCH_SWEEP(P, KK) k_sweep<P, KK><<<grid, SWEEP_THREADS, 0, h->stream>>>(h->d_src, h->src_pitch, h->d_state, h->d_grid, \
      h->d_words, h->d_recs, h->d_partials, h->chunks_per_pair, h->d_active_list, h->d_ctl, sc)
  if (sc.pca) { if (sc.K == 1) NDT_LAUNCH_SWEEP(true, 1); else if (sc.K == 7) NDT_LAUNCH_SWEEP(true, 7); else NDT_LAUNCH_SWEEP(true, 26); }
  else        { if (sc.K == 1) NDT_LAUNCH_SWEEP(false, 1); else if (sc.K == 7) NDT_LAUNCH_SWEEP(false, 7); else NDT_LAUNCH_SWEEP(false, 26); }
#undef NDT_LAUNCH_SWEEP
  if (h->prof) HIPCHK(h, ev_end(h, h->ev_sweep));
  return MI355NDT_OK;
}

int mi355ndt_batch_align(mi355ndt_handle* h, const float* guesses, mi355ndt_result* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!guesses || !out) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs <= 0 || !h->d_tgt || !h->d_src) return MI355NDT_ERR_STATE;
  if (h->prm.neighbor_mode == MI355NDT_KDTREE) return MI355NDT_ERR_UNSUPPORTED;
  if (!((h->prm.step_size - h->prm.trans_epsilon / 2) > 0)) return MI355NDT_ERR_UNSUPPORTED;   // impl2:888: live More-Thuente loop
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->targets_built) { int rc = mi355ndt_batch_build_targets(h); if (rc) return rc; }
  int rc = prep_align_ws(h);
  if (rc) return rc;
  const int B = h->n_pairs;
  hipStream_t s = h->stream;
  SweepConst sc;
  make_sweep_const(h, sc);
  HIPCHK(h, hipMemcpyAsync(h->d_guess, guesses, (size_t)B * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipStreamSynchronize(s));           // guesses may be pageable caller memory
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, sizeof(SweepCtl), s));
  k_init_state<<<(B + 63) / 64, 64, 0, s>>>(h->d_state, h->d_guess, h->d_src_cnt, h->d_grid, B, h->d_active_list, h->d_ctl);
  rc = launch_sweep(h, sc);
  if (rc) return rc;
  const int max_rounds = h->prm.max_iterations + 4;   // loop body runs for it = 0 .. max_iterations+1
  double pts_total = 0;
  for (int b = 0; b < B; b++) pts_total += h->h_src_cnt[b];
  const double alg_static = pts_total * (12.0 + 4.0 * sc.K);   // every active pair streams its points + K table probes
  if (h->prof) {
    HIPCHK(h, hipMemsetAsync(h->d_hits, 0, sizeof(unsigned long long), s));
    h->P.sweep_alg_bytes += alg_static;                          // the initial sweep covers all pairs
    h->P.sweep_points += (long long)pts_total;
  }
  const int burst = 2;                                           // update+sweep rounds enqueued between host checks
  int round = 0;
  while (round < max_rounds) {
    HIPCHK(h, hipMemsetAsync(h->d_active, 0, 128 * sizeof(int), s));
    const int r0 = round;
    for (int k = 0; k < burst && round < max_rounds; k++, round++) {
      if (h->prof) HIPCHK(h, ev_begin(h, h->ev_update));
      HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, sizeof(SweepCtl), s));
      k_update<<<B, 64, 0, s>>>(h->d_state, h->d_partials, h->chunks_per_pair, h->d_results, h->d_active + (round - r0),
                                h->d_active_list, h->d_ctl, h->prof ? h->d_hits : nullptr,
                                h->prm.step_size, h->prm.trans_epsilon, h->prm.max_iterations, 0);
      if (h->prof) HIPCHK(h, ev_end(h, h->ev_update));
      rc = launch_sweep(h, sc);
      if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_pin_active, h->d_active, burst * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (h->prof) {
      // sweep k of this burst streamed the pairs that scheduled a step in update k (equal-size pairs assumed)
      for (int k = 0; k < round - r0; k++) {
        const double frac = (double)h->h_pin_active[k] / B;
        h->P.sweep_alg_bytes += alg_static * frac;
        h->P.sweep_points += (long long)(pts_total * frac);
      }
    }
    if (h->h_pin_active[round - r0 - 1] == 0) break;
  }
  if (h->prof) {
    // one more reduction so the hits of the very last sweeps are counted is not needed: every sweep is
    // followed by an update (the loop only exits after an update scheduled no further sweep)
    unsigned long long hh = 0;
    HIPCHK(h, hipMemcpyAsync(&hh, h->d_hits, sizeof hh, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    h->P.sweep_hits += (long long)hh;
    h->P.sweep_alg_bytes += 64.0 * (double)hh;
  }
  HIPCHK(h, hipMemcpyAsync(out, h->d_results, (size_t)B * sizeof(mi355ndt_result), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  return MI355NDT_OK;
}

// ---- single-registration surface (pair slot 0) ---------------------------------------------------
static int ensure_single(mi355ndt_handle* h, bool tgt, size_t n) {
  const size_t want = std::max(((n + 63) & ~(size_t)63), (size_t)64);
  const bool single = h->n_pairs == 1 && h->d_tgt_own && h->d_src_own && h->d_tgt == h->d_tgt_own && h->d_src == h->d_src_own;
  if (!single) {
    // leaving batch / bound mode: start a fresh one-pair engine
    return mi355ndt_batch_reserve(h, 1, tgt ? want : 64, tgt ? 64 : want);
  }
  // target and source buffers are independent: grow only the side being replaced
  const size_t have = tgt ? h->own_tgt_pitch : h->own_src_pitch;
  if (n <= have) return MI355NDT_OK;
  int rc = alloc_side(h, tgt, 1, want);
  if (rc) return rc;
  if (tgt) h->tgt_pitch = want; else h->src_pitch = want;
  h->d_tgt = h->d_tgt_own; h->d_src = h->d_src_own;
  return MI355NDT_OK;
}

int mi355ndt_set_target(mi355ndt_handle* h, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if ((!pts && n) || (n && stride < 12) || n >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_single(h, true, n);
  if (rc) return rc;
  rc = mi355ndt_batch_set_target(h, 0, pts, n, stride);
  if (rc) return rc;
  h->have_target = true;
  return mi355ndt_batch_build_targets(h);      // init(): filter(true) (ndt_omp.h:270-277)
}

int mi355ndt_set_source(mi355ndt_handle* h, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if ((!pts && n) || (n && stride < 12) || n >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_single(h, false, n);
  if (rc) return rc;
  rc = mi355ndt_batch_set_source(h, 0, pts, n, stride);
  if (rc) return rc;
  h->have_source = true;
  return MI355NDT_OK;
}

int mi355ndt_set_params(mi355ndt_handle* h, const mi355ndt_params* p) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!p) return MI355NDT_ERR_BAD_ARG;
  int rc = check_params(*p);
  if (rc) return rc;
  const mi355ndt_params old = h->prm;
  h->prm = *p;
  const bool regrid = old.resolution != p->resolution || old.variant != p->variant ||
                      old.min_points_per_voxel != p->min_points_per_voxel ||
                      old.min_covar_eigvalue_mult != p->min_covar_eigvalue_mult;
  if (regrid && h->targets_built) {
    h->targets_built = false;
    if (p->neighbor_mode != MI355NDT_KDTREE) return mi355ndt_batch_build_targets(h);   // setResolution -> init() (ndt_omp.h:126-136)
  }
  return MI355NDT_OK;
}

int mi355ndt_align(mi355ndt_handle* h, const float guess[16], mi355ndt_result* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!guess || !out) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs < 1 || !h->have_target || !h->have_source) return MI355NDT_ERR_STATE;
  if (h->n_pairs != 1) return MI355NDT_ERR_STATE;                   // a batch is bound: use mi355ndt_batch_align
  // pcl::Registration::initCompute() refuses empty clouds; align() then returns without touching converged_
  if (h->h_tgt_cnt[0] <= 0 || h->h_src_cnt[0] <= 0) return MI355NDT_ERR_STATE;
  int rc = mi355ndt_batch_align(h, guess, out);
  if (rc == MI355NDT_OK) memcpy(h->last_final, out->final_colmajor, sizeof h->last_final);
  return rc;
}

int mi355ndt_get_aligned(mi355ndt_handle* h, void* out_pts, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!out_pts || stride < 12) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs < 1 || !h->d_src) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  const int n = h->h_src_cnt[0];
  if (n == 0) return MI355NDT_OK;
  HIPCHK(h, grow(h->d_aligned, h->aligned_cap, (size_t)3 * n));
  k_transform<<<(n + 255) / 256, 256, 0, h->stream>>>(h->d_src, h->src_pitch, h->d_state, 0, h->d_aligned, n);
  std::vector<float> tmp((size_t)3 * n);
  HIPCHK(h, hipMemcpyAsync(tmp.data(), h->d_aligned, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unsigned char* o = (unsigned char*)out_pts;
  for (int i = 0; i < n; i++) {
    float v[3] = {tmp[i], tmp[(size_t)n + i], tmp[(size_t)2 * n + i]};
    memcpy(o + (size_t)i * stride, v, 12);
  }
  return MI355NDT_OK;
}

static int run_hook_sweep(mi355ndt_handle* h, double* score, double g[6], double H[36], long long* hits) {
  SweepConst sc;
  make_sweep_const(h, sc);
  int rc = launch_sweep(h, sc);
  if (rc) return rc;
  k_update<<<1, 64, 0, h->stream>>>(h->d_state, h->d_partials, h->chunks_per_pair, h->d_results, h->d_active, h->d_active_list, h->d_ctl,
                                    nullptr, 0, 0, 0, 1);
  PairState S;
  HIPCHK(h, hipMemcpyAsync(&S, h->d_state, sizeof(PairState), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  if (score) *score = S.score;
  if (g) memcpy(g, S.g, sizeof S.g);
  if (H) memcpy(H, S.H, sizeof S.H);
  if (hits) *hits = S.hits;
  return MI355NDT_OK;
}

static int hook_ready(mi355ndt_handle* h) {
  if (h->n_pairs < 1 || !h->have_target || !h->have_source) return MI355NDT_ERR_STATE;
  if (h->prm.neighbor_mode == MI355NDT_KDTREE) return MI355NDT_ERR_UNSUPPORTED;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->targets_built) { int rc = mi355ndt_batch_build_targets(h); if (rc) return rc; }
  return prep_align_ws(h);
}

int mi355ndt_derivatives(mi355ndt_handle* h, const double p[6], double* score, double g[6], double H[36], long long* hits) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!p) return MI355NDT_ERR_BAD_ARG;
  int rc = hook_ready(h);
  if (rc) return rc;
  double* dp = (double*)h->d_hook;
  HIPCHK(h, hipMemcpyAsync(dp, p, 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, sizeof(SweepCtl), h->stream));
  k_set_pose_p<<<1, 1, 0, h->stream>>>(h->d_state, 0, dp, h->d_src_cnt, h->d_grid, h->d_active_list, h->d_ctl);
  return run_hook_sweep(h, score, g, H, hits);
}

int mi355ndt_derivatives_T(mi355ndt_handle* h, const float T[16], const float Rj[9], double* score, double g[6], double H[36], long long* hits) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!T || !Rj) return MI355NDT_ERR_BAD_ARG;
  int rc = hook_ready(h);
  if (rc) return rc;
  float buf[25];
  memcpy(buf, T, 16 * sizeof(float));
  memcpy(buf + 16, Rj, 9 * sizeof(float));
  HIPCHK(h, hipMemcpyAsync(h->d_hook, buf, sizeof buf, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, sizeof(SweepCtl), h->stream));
  k_set_pose<<<1, 1, 0, h->stream>>>(h->d_state, 0, h->d_hook, h->d_hook + 16, h->d_src_cnt, h->d_grid, h->d_active_list, h->d_ctl);
  return run_hook_sweep(h, score, g, H, hits);
}

int mi355ndt_get_grid(mi355ndt_handle* h, int pair, int min_b[3], int max_b[3], int div_b[3], int* n_voxels) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs) return MI355NDT_ERR_BAD_ARG;
  if (!h->targets_built) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  GridDesc g;
  HIPCHK(h, hipMemcpyAsync(&g, h->d_grid + pair, sizeof g, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int a = 0; a < 3; a++) {
    if (min_b) min_b[a] = g.min_b[a];
    if (max_b) max_b[a] = g.max_b[a];
    if (div_b) div_b[a] = g.div_b[a];
  }
  if (n_voxels) *n_voxels = g.n_voxels;
  return (g.status == GRID_OVERFLOW || g.status == GRID_CAP) ? MI355NDT_ERR_GRID : MI355NDT_OK;
}

int mi355ndt_get_voxels(mi355ndt_handle* h, int pair, mi355ndt_voxel* out, size_t capacity) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs || (!out && capacity)) return MI355NDT_ERR_BAD_ARG;
  if (!h->targets_built) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  GridDesc g;
  HIPCHK(h, hipMemcpyAsync(&g, h->d_grid + pair, sizeof g, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  size_t n = std::min((size_t)g.n_voxels, capacity);
  if (n == 0) return MI355NDT_OK;
  std::vector<VoxelRec> r(n);
  std::vector<int> idx(n), cnt(n);
  HIPCHK(h, hipMemcpy(r.data(), h->d_recs + g.rec_off, n * sizeof(VoxelRec), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(idx.data(), h->d_vox_idx + g.rec_off, n * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(cnt.data(), h->d_vox_n + g.rec_off, n * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) {
    out[i].idx = idx[i];
    out[i].n = cnt[i];
    memcpy(out[i].mean, r[i].mean, sizeof r[i].mean);
    memcpy(out[i].icov, r[i].icov, sizeof r[i].icov);
    out[i].weight = (r[i].weight == VOX_DEAD) ? 0 : r[i].weight;
  }
  return MI355NDT_OK;
}

// replaces pcl::Registration::getFitnessScore(max_range) for the loop-closure caller (loop_detector.hpp:249-262)
int mi355ndt_fitness_score_T(mi355ndt_handle* h, const float T_colmajor[16], double max_range, double* score, long long* n_inliers) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!T_colmajor || !score) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs != 1 || !h->have_target || !h->have_source) return MI355NDT_ERR_STATE;
  if (h->h_tgt_cnt[0] <= 0 || h->h_src_cnt[0] <= 0) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->targets_built) { int rc = mi355ndt_batch_build_targets(h); if (rc) return rc; }
  hipStream_t s = h->stream;
  GridDesc g;
  HIPCHK(h, hipMemcpyAsync(&g, h->d_grid, sizeof g, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  if (g.status != GRID_OK) { *score = 1.7976931348623157e308; if (n_inliers) *n_inliers = 0; return MI355NDT_OK; }
  if (!h->cells_ready) {
    const size_t nc = (size_t)g.ncells;
    if (nc > h->cell_cap) {
      size_t c1 = 0, c2 = 0;
      if (h->d_cstart) { HIPCHK(h, hipFree(h->d_cstart)); h->d_cstart = nullptr; }
      if (h->d_cend) { HIPCHK(h, hipFree(h->d_cend)); h->d_cend = nullptr; }
      HIPCHK(h, grow(h->d_cstart, c1, nc)); HIPCHK(h, grow(h->d_cend, c2, nc));
      h->cell_cap = nc;
    }
    HIPCHK(h, hipMemsetAsync(h->d_cstart, 0, nc * sizeof(unsigned), s));
    HIPCHK(h, hipMemsetAsync(h->d_cend, 0, nc * sizeof(unsigned), s));
    k_cellrange<unsigned><<<(unsigned)((h->tgt_pitch + 255) / 256), 256, 0, s>>>((const unsigned*)h->d_keys_b, h->tgt_pitch, h->last_cb,
                                                                                h->d_cstart, h->d_cend);
    h->cells_ready = true;
  }
  const int n = h->h_src_cnt[0];
  const int blocks = (n + 255) / 256;
  HIPCHK(h, grow(h->d_fit, h->fit_cap, (size_t)2 * blocks));
  HIPCHK(h, hipMemcpyAsync(h->d_hook, T_colmajor, 16 * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipStreamSynchronize(s));
  const float mr = max_range >= 3.0e38 ? 3.0e38f : (float)max_range;
  // rings needed to cover sqrt(max_range) (+1 cell of slack), capped by the grid's extent
  const int extent = std::max(g.div_b[0], std::max(g.div_b[1], g.div_b[2])) + 2;
  double rr = std::sqrt(std::min(max_range, 1e30)) / (double)g.leaf + 2.0;
  int ring_max = rr > (double)(1 << 28) ? (1 << 28) : (int)rr;
  // a query outside the grid may sit further away than the grid is wide: allow its distance to the box on top
  ring_max = std::min(ring_max, (1 << 20));
  (void)extent;
  k_fitness<<<blocks, 256, 0, s>>>(h->d_src, h->src_pitch, n, h->d_tgt, h->tgt_pitch, h->d_vals_b, h->d_grid, h->d_cstart, h->d_cend,
                                   h->d_hook, mr, ring_max, h->d_fit);
  std::vector<double> part((size_t)2 * blocks);
  HIPCHK(h, hipMemcpyAsync(part.data(), h->d_fit, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  double sum = 0, cnt = 0;
  for (int b = 0; b < blocks; b++) { sum += part[2 * b]; cnt += part[2 * b + 1]; }
  *score = cnt > 0 ? sum / cnt : 1.7976931348623157e308;     // std::numeric_limits<double>::max()
  if (n_inliers) *n_inliers = (long long)cnt;
  return MI355NDT_OK;
}

int mi355ndt_get_fitness_score(mi355ndt_handle* h, double max_range, double* score, long long* n_inliers) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  return mi355ndt_fitness_score_T(h, h->last_final, max_range, score, n_inliers);
}

// replaces PrefilteringNodelet::distance_filter + downsample (prefiltering_nodelet.cpp:137-181)
int mi355ndt_prefilter(mi355ndt_handle* h, const void* pts, size_t n, size_t stride,
                       int use_distance_filter, double distance_near, double distance_far, float downsample_resolution,
                       void* out_pts, size_t out_capacity, size_t out_stride, size_t* n_out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if ((!pts && n) || (n && stride < 12) || n >= (1u << 30) || !n_out || (out_pts && out_stride < 12)) return MI355NDT_ERR_BAD_ARG;
  if (std::isnan(downsample_resolution)) return MI355NDT_ERR_BAD_ARG;
  *n_out = 0;
  h->pf_count = 0;
  if (n == 0) return MI355NDT_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t pitch = (n + 63) & ~(size_t)63;
  if (pitch > h->pf_cap) {
    HIPCHK(h, hipStreamSynchronize(s));
    void** ps[] = {(void**)&h->d_pf_in, (void**)&h->d_pf_out, (void**)&h->d_pf_keep, (void**)&h->d_pf_keys, (void**)&h->d_pf_vals,
                   (void**)&h->d_pf_flag, (void**)&h->d_pf_pos};
    const size_t bytes[] = {3 * pitch * 4, 3 * pitch * 4, pitch, 2 * pitch * 4, 2 * pitch * 4, pitch * 4, pitch * 4};
    for (int i = 0; i < 7; i++) { if (*ps[i]) { HIPCHK(h, hipFree(*ps[i])); *ps[i] = nullptr; } HIPCHK(h, hipMalloc(ps[i], bytes[i])); }
    if (!h->d_pf_mm) HIPCHK(h, hipMalloc((void**)&h->d_pf_mm, 6 * sizeof(int)));
    if (!h->d_pf_grid) HIPCHK(h, hipMalloc((void**)&h->d_pf_grid, sizeof(PfGrid)));
    h->pf_cap = pitch;
  }
  h->pf_pitch = pitch;
  int rc = upload_cloud(h, h->d_pf_in, pitch, 0, pts, n, stride);
  if (rc) return rc;
  const int gx = (int)((pitch + 255) / 256);
  unsigned *ka = h->d_pf_keys, *kb = h->d_pf_keys + pitch, *va = h->d_pf_vals, *vb = h->d_pf_vals + pitch;
  size_t t1 = 0, t2 = 0;
  HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(nullptr, t1, ka, kb, va, vb, (int)pitch, 0, 31, s));
  HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(nullptr, t2, h->d_pf_flag, h->d_pf_pos, (int)pitch, s));
  t1 = std::max(t1, t2);
  if (t1 > h->pf_tmp_bytes) {
    if (h->d_pf_tmp) { HIPCHK(h, hipFree(h->d_pf_tmp)); h->d_pf_tmp = nullptr; }
    HIPCHK(h, hipMalloc(&h->d_pf_tmp, t1));
    h->pf_tmp_bytes = t1;
  }
  k_minmax_init<<<1, 64, 0, s>>>(h->d_pf_mm, 1);
  k_pf_flag<<<std::min(gx, 256), 256, 0, s>>>(h->d_pf_in, pitch, (int)n, use_distance_filter, distance_near, distance_far, h->d_pf_keep, h->d_pf_mm);
  int downsample = downsample_resolution > 0.f;
  const unsigned* keys_sorted = ka;
  const unsigned* vals_sorted = va;
  if (downsample) {
    k_pf_grid<<<1, 1, 0, s>>>(h->d_pf_mm, downsample_resolution, h->d_pf_grid);
    PfGrid g;
    HIPCHK(h, hipMemcpyAsync(&g, h->d_pf_grid, sizeof g, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (g.status == 2) {                         // PCL: "Leaf size is too small for the input dataset" -> output = input
      h->err = "prefilter: leaf size too small for the cloud's extent, voxel indices would overflow; cloud not down-sampled";
      downsample = 0;
    } else {
      k_pf_keys<<<gx, 256, 0, s>>>(h->d_pf_in, pitch, (int)n, h->d_pf_keep, h->d_pf_grid, ka, va);
      size_t tb = h->pf_tmp_bytes;
      HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(h->d_pf_tmp, tb, ka, kb, va, vb, (int)pitch, 0, 31, s));
      keys_sorted = kb; vals_sorted = vb;
    }
  }
  k_pf_heads<<<gx, 256, 0, s>>>(keys_sorted, h->d_pf_keep, (int)n, pitch, downsample, h->d_pf_flag);
  size_t tb = h->pf_tmp_bytes;
  HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(h->d_pf_tmp, tb, h->d_pf_flag, h->d_pf_pos, (int)pitch, s));
  k_pf_emit<<<gx, 256, 0, s>>>(h->d_pf_in, pitch, keys_sorted, vals_sorted, h->d_pf_flag, h->d_pf_pos, downsample, h->d_pf_out, pitch);
  int last_pos = 0, last_flag = 0;
  HIPCHK(h, hipMemcpyAsync(&last_pos, h->d_pf_pos + (pitch - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(&last_flag, h->d_pf_flag + (pitch - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  const size_t m = (size_t)last_pos + (size_t)last_flag;
  h->pf_count = (int)m;
  *n_out = m;
  if (out_pts) {
    if (m > out_capacity) return MI355NDT_ERR_BAD_ARG;
    std::vector<float> tmp(3 * pitch);
    HIPCHK(h, hipMemcpy(tmp.data(), h->d_pf_out, 3 * pitch * sizeof(float), hipMemcpyDeviceToHost));
    unsigned char* o = (unsigned char*)out_pts;
    for (size_t i = 0; i < m; i++) {
      float v[3] = {tmp[i], tmp[pitch + i], tmp[2 * pitch + i]};
      memcpy(o + i * out_stride, v, 12);
    }
  }
  return MI355NDT_OK;
}

// hand the last prefilter result to the registration without leaving the GPU: role 1 = setInputSource, 2 = setInputTarget
int mi355ndt_use_prefiltered(mi355ndt_handle* h, int role) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (role != 1 && role != 2) return MI355NDT_ERR_BAD_ARG;
  if (!h->d_pf_out) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t m = (size_t)h->pf_count;
  int rc = ensure_single(h, role == 2, m);
  if (rc) return rc;
  float* dst = role == 2 ? h->d_tgt_own : h->d_src_own;
  const size_t dp = role == 2 ? h->tgt_pitch : h->src_pitch;
  HIPCHK(h, hipMemsetAsync(dst, 0, 3 * dp * sizeof(float), h->stream));
  for (int a = 0; a < 3; a++)
    if (m) HIPCHK(h, hipMemcpyAsync(dst + a * dp, h->d_pf_out + a * h->pf_pitch, m * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  if (role == 2) {
    h->h_tgt_cnt[0] = (int)m; h->have_target = true; h->targets_built = false;
    return mi355ndt_batch_build_targets(h);
  }
  h->h_src_cnt[0] = (int)m; h->have_source = true;
  return MI355NDT_OK;
}

}  // extern "C"
