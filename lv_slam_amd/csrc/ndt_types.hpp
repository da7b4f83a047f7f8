// ndt_types.hpp -- constants, device-resident structures and small helpers shared by all kernels of the
// MI355X NDT engine (included by mi355_ndt.hip; one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstddef>
#include <cstring>
#include "mi355_ndt.h"

// Non-template kernels defined in headers that both translation units include: the second unit (mi355_ndt_ord1.hip, kernel instantiations
// only) sees them as unused internal functions and emits nothing for them.
#ifdef NDT_SECOND_TU
#define NDT_KERNEL static __global__
#else
#define NDT_KERNEL __global__
#endif

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
enum { PH_SWEEP0 = 0, PH_STEP = 1, PH_DONE = 2, PH_HESS = 3 };   // PH_HESS: waiting for the computeHessian pass (live More-Thuente only)

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

// MI355NDT_OPT_ARITH = 1 (tolerance arithmetic): what one evaluation reads then -- the same 64-byte slot, written by k_voxels
// beside the exact record.  The f64 mean as an f32 head + tail (x' - mh is exact for a point in or next to the leaf's cell, so
// (x' - mh) - ml is the reference's f32(f64(x') - mean) to a rounding), the inverse covariance as its six unique entries (off-diagonals:
// the mean of the two the cofactor inverse produced), `weight` where VoxelRec has it (the probe stage of ndt_pca reads it from either).
struct alignas(16) VoxelRecF {
  float mh[3], ml[3];
  float c[6];                  // c00 c01 c02 c11 c12 c22
  float pad_[3];
  int   weight;
};
static_assert(sizeof(VoxelRecF) == sizeof(VoxelRec) && offsetof(VoxelRecF, weight) == offsetof(VoxelRec, weight), "the two record forms share slot size and weight word");

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
  // live More-Thuente case only (step_size <= eps/2, impl2:888): tangent of the pending trial, phi(0), phi'(0)
  double xt[6], phi0, dphi0;
  int    mt_loops, pad_;
  // pcl::Registration::transformation_ / previous_transformation_ as computeTransformation leaves them (impl2:134, 163):
  // float(exp(delta_p)) of the last and of the one-before-last Newton step, column-major
  float  inc_cm[16], prev_inc_cm[16];
  // one-launch align only (ndt_async.hpp): the re-basing of p for the NEXT update -- log(exp(delta_p) exp(p)) and float(exp(delta_p)), impl2:163-166 --
  // computed by the updater AFTER it published the pair's next sweep, i.e. off the pair's critical path; valid when reb_tag == sweeps
  double reb_pn[6];
  float  reb_inc[16];
  long long reb_tag;
};

struct SweepConst {
  double d1;
  float  d2f;
  int    K;                    // neighbour probes
  int    pca;
  int    table;                // row of c_off: 0 = DIRECT1, 1 = DIRECT7, 2 = DIRECT26
  float  kd_r2;                // KDTREE: float(resolution * resolution), the squared search radius
  int    leaf_pow2;            // resolution is a power of two: x / leaf == x * inv_leaf bit for bit
  float  inv_leaf;
  int    dyn_shift;            // the sweep claims 1 / 2^dyn_shift of every work queue dynamically, the rest is dealt statically
  // latency mode only (may be null): two words in mapped host memory the sweep reports to -- [1] = active pairs it found, then
  // [0] = its sequence number -- so that the host can pump (update, sweep) launches without waiting for a copy or an event
  volatile int* host_flags;
  int    seq_no;
  int    rebase_block;         // latency mode: the launch's LAST workgroup does not sweep -- it computes the re-basing of p the next Newton update starts with (ndt_sweep.hpp)
  // tolerance arithmetic (ORD = 2 instantiations only): d1 as f32, and kq = -d2 / 2 * log2(e): exp(-d2 q / 2) = 2^(kq q), one v_exp_f32
  float  d1f, kq;
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

// The "leaf size is too small" guard of pcl::VoxelGrid / VoxelGridCovariance (voxel_grid_covariance_omp_impl.hpp:75-84): dx*dy*dz >
// INT32_MAX with d = int64((max - min) * inv_leaf) + 1, e = that f32 extent in cells.  The reference multiplies three int64 factors;
// beyond 2^63 cells (a 1e-4 m leaf over a 240 m cloud, a stray point at 1e30) that product, and for extents beyond 2^63 the
// float -> int64 cast, is undefined behaviour.  What the guard means is not in doubt; it is evaluated without overflowing.
__device__ __forceinline__ bool grid_too_big(float e0, float e1, float e2) {
  const float e[3] = {e0, e1, e2};
  double prod = 1.0;
  for (int a = 0; a < 3; a++) {
    if (!(e[a] < 2147483648.0f)) return true;    // (also NaN / inf extents)
    prod *= (double)((long long)e[a] + 1);
  }
  return prod > 2147483647.0;
}
__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// XCD-aware placement for kernels that give every target `nx` workgroups: a fresh launch hands workgroup L to XCD L % 8
// (MI355X_MICROARCH.md), so the 1-D grid 8 * ceil(B / 8) * nx is read as  xcd = L & 7,  s = L >> 3,  target = xcd + 8 * (s / nx),
// block-in-target = s % nx: all workgroups of one target run on ONE XCD (its points / keys stay in that XCD's 4 MB L2 and
// scattered writes to one cache line meet in one L2), and an XCD walks through its targets in order.
__device__ __forceinline__ bool xcd_map(int nx, int n_targets, int& bx, int& b) {
  const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
  b = xcd + 8 * (s / nx);
  bx = s % nx;
  return b < n_targets;
}
static inline unsigned xcd_grid(int nx, int n_targets) { return 8u * (unsigned)((n_targets + 7) / 8) * (unsigned)nx; }


// Exclusive prefix sum over the NT threads of a block (NT a multiple of 64, <= 1024): wave-level shuffle scan, the wave totals
// scanned by wave 0 through `sm` (NT / 64 + 1 entries of LDS).  Returns the thread's exclusive prefix; *total = block sum.
template <int NT>
__device__ __forceinline__ unsigned block_exscan(unsigned v, unsigned* total, unsigned* sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  if (w == 0) {
    const unsigned s = lane < NT / 64 ? sm[lane] : 0u;
    unsigned si = s;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const unsigned t = __shfl_up(si, o); if (lane >= o) si += t; }
    if (lane < NT / 64) sm[lane] = si - s;
    if (lane == NT / 64 - 1) sm[NT / 64] = si;
  }
  __syncthreads();
  const unsigned ex = inc - v + sm[w];
  *total = sm[NT / 64];
  __syncthreads();                               // `sm` may be reused by the caller's next round
  return ex;
}
