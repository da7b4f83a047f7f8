// ndt_sweep.hpp -- the derivative sweep (computeDerivatives + updateDerivatives + neighbour lookup,
// include/ndt_omp/ndt_omp_impl2.hpp:196-305, 503-532, 566-619; voxel_grid_covariance_omp_impl.hpp:373-442).
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include <type_traits>

// ------------------------------------------------------------------------------------ derivative sweep
// One (point, voxel) evaluation: updateDerivatives (ndt_omp_impl2.hpp:566-619) with the Jacobian /
// Hessian patterns of computePointDerivatives_AngleAxisd (impl2:503-532) folded in (J and Hp are never
// materialised).  f32 ops single, left to right; f64 accumulation.  `w` = weight multiplier of the hit
// (ndt_pca compounding, applied as a suffix product; unused for ndt_omp).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `mid` runs between the gradient part and the 36 Hessian terms, where the fewest temporaries are live: the sweep uses it
// to issue the NEXT batch's record loads so that their L2 latency overlaps the Hessian arithmetic.
// SAN (KDTREE modes only): radiusSearch also returns leaves whose inverse covariance is non-finite (no nr_points re-check,
// voxel_grid_covariance_omp.h:505-534); the reference rejects such a hit before touching the sums (impl2:588-589), so the
// operands of a rejected hit are zeroed first -- "e = 0" alone would still add 0 * NaN.
// ORD: evaluation order of the three-term f32 sums of impl2:581, 594-613 (4-wide Eigen inner products whose fourth term is a structural
// zero).  The reference leaves it to Eigen 3.3 and the SSE level it is compiled for (CMakeLists.txt:6,11: -msse4.2), and neither can
// be observed here.  0 = (t0 + t1) + t2: Eigen's scalar redux, the canonical choice of the committed parity fixtures;
// 1 = (t0 + t2) + t1: the lane pairing of Eigen 3.3's SSE predux<Packet4f>, (a0 + a2) + (a1 + a3) with a3 = 0.  Same instruction count
// either way; selected per engine with mi355ndt_set_option(MI355NDT_OPT_F32_SUM_ORDER) so that a maintainer who pins the reference on
// a real build (tools/pin_reference) can switch to the order that build shows.
template <int ORD>
__device__ __forceinline__ float sum3(float t0, float t1, float t2) { return ORD == 1 ? (t0 + t2) + t1 : (t0 + t1) + t2; }

template <bool PCA, typename Mid = NoHook, bool SAN = false, int ORD = 0>
__device__ __forceinline__ void eval_hit(const float u_in[3], const float r[3], const float C_in[9],
                                         const double d1, const float d2f, const double w, const bool ok_in, double acc[43],
                                         const double* __restrict__ exp_tab, Mid mid = Mid()) {
  float u[3] = {u_in[0], u_in[1], u_in[2]}, C[9];
#pragma unroll
  for (int a = 0; a < 9; a++) C[a] = C_in[a];
  float y[3];
#pragma unroll
  for (int j = 0; j < 3; j++) y[j] = sum3<ORD>(u[0] * C[j], u[1] * C[3 + j], u[2] * C[6 + j]);
  const float qf = sum3<ORD>(u[0] * y[0], u[1] * y[1], u[2] * y[2]);
  const float e0 = ndtm::exp_f32arg((-d2f * qf) * 0.5f, exp_tab);                // impl2:581: exp in f64 on the f32 argument, rounded to f32
  float s_inc = (float)(-d1 * (double)e0);                                       // impl2:583
  const float e1 = d2f * e0;                                                     // impl2:585
  // impl2:588-589, branch-free: a rejected hit (or an idle lane, ok_in = false) multiplies every term by e = 0 and so
  // adds +0 to all 43 sums (all operands are finite here: dead voxels never enter the queue).
  const bool ok = ok_in && !(e1 > 1.f || e1 < 0.f || e1 != e1);
  float e = (float)((double)e1 * d1);                                            // impl2:592
  e = ok ? e : 0.f;
  s_inc = ok ? s_inc : 0.f;
  if (SAN) {
#pragma unroll
    for (int a = 0; a < 3; a++) { u[a] = ok ? u[a] : 0.f; y[a] = ok ? y[a] : 0.f; }
#pragma unroll
    for (int a = 0; a < 9; a++) C[a] = ok ? C[a] : 0.f;
  }
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
  for (int k = 0; k < 6; k++) v[k] = sum3<ORD>(u[0] * CJ[0][k], u[1] * CJ[1][k], u[2] * CJ[2][k]);   // impl2:595
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
  __builtin_amdgcn_sched_barrier(0);
  mid();
  __builtin_amdgcn_sched_barrier(0);
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
      // z_i[j] is a structural zero outside the rotation block; "+ (+-0)" never changes a value that reaches the f64 sums
      // (it can only turn a -0 term into +0, and adding either to an accumulator is a no-op), so the add is skipped there
      float t = ((-d2f) * v[i]) * v[j];
      if (i >= 3 && j >= 3) t = t + Z[i - 3][j - 3];
      const float h = e * (t + jcj);                                                           // impl2:611-613
      NDT_ACC(7 + i * 6 + j, h);
    }
  }
#undef NDT_ACC
}

// ------------------------------------------------------------------------------------ tolerance arithmetic (ORD = 2)
// MI355NDT_OPT_ARITH = 1: the same evaluation -- updateDerivatives with the patterns of computePointDerivatives_AngleAxisd folded in -- priced
// for north_star's tolerance (trans < 1e-4 m, rot < 1e-5 rad) instead of one separately rounded operation per step of the reference recipe.  What changes against eval_hit:
//   * fused multiply-adds, one v_exp_f32 (exp(-d2 q / 2) = 2^(kq q)) instead of the f64 table + polynomial, d1 as f32;
//   * the inverse covariance as a symmetric matrix (six entries), so J^T C J is symmetric and H = S + [0 0; 0 Z] with S symmetric:
//     21 sums for S, and the asymmetric part (impl2:522-530, 607) Z = r y^T - (y . r) I needs only A = sum e r y^T (9 sums; the trace term
//     is recovered from A's diagonal when the row is written);
//   * 37 f32 accumulators per lane instead of 43 f64 ones; a lane adds ~35 terms into them per work item (512 points) before they are widened
//     and go through the f64 wave tree and the f64 row sums of the update as before.
// What does NOT change: the point transform and the cell lookup (uncontracted f32, SURVEY.md H3: which leaf a point meets is discontinuous),
// the validity gate of impl2:588-589, the Newton update.  This mode is held to the tolerance, not to bits
// (tests/test_tolerance_mode.py, bench.py `tolerance_mode`).
#define NACC_F 37
__host__ __device__ constexpr int fsym(int i, int j) { return i <= j ? 7 + i * 6 - i * (i - 1) / 2 + (j - i) : 7 + j * 6 - j * (j - 1) / 2 + (i - j); }
#define FA_BASE 28             // A[i][j] = a[FA_BASE + 3 i + j]
template <bool PCA, typename Mid = NoHook>
__device__ __forceinline__ void eval_hit_fast(const float u[3], const float r[3], const float c[6], const float d1f, const float d2f, const float kq,
                                              const float w, const bool ok_in, float a[NACC_F], Mid mid = Mid()) {
  const float c00 = c[0], c01 = c[1], c02 = c[2], c11 = c[3], c12 = c[4], c22 = c[5];
  float y[3];
  y[0] = fmaf(c02, u[2], fmaf(c01, u[1], c00 * u[0]));
  y[1] = fmaf(c12, u[2], fmaf(c11, u[1], c01 * u[0]));
  y[2] = fmaf(c22, u[2], fmaf(c12, u[1], c02 * u[0]));
  const float q = fmaf(u[2], y[2], fmaf(u[1], y[1], u[0] * y[0]));
  const float e0 = __builtin_amdgcn_exp2f(kq * q);                               // impl2:581
  const float e1 = d2f * e0;                                                     // impl2:585
  const bool ok = ok_in && !(e1 > 1.f || e1 < 0.f || e1 != e1);                  // impl2:588-589
  float e = e1 * d1f, s = -d1f * e0;                                             // impl2:592, 583
  if (PCA) { e *= w; s *= w; }
  e = ok ? e : 0.f;
  s = ok ? s : 0.f;
  a[0] += s;
  // v = J^T y = [y ; r x y] (impl2:595 with CJ's columns 0..2 = C)
  float v[6] = {y[0], y[1], y[2], fmaf(r[1], y[2], -(r[2] * y[1])), fmaf(r[2], y[0], -(r[0] * y[2])), fmaf(r[0], y[1], -(r[1] * y[0]))};
#pragma unroll
  for (int k = 0; k < 6; k++) a[1 + k] = fmaf(e, v[k], a[1 + k]);                // impl2:597
  __builtin_amdgcn_sched_barrier(0);
  mid();
  __builtin_amdgcn_sched_barrier(0);
  // M = C (-[r]x): columns 3..5 of CJ (impl2:594)
  const float C[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}};
  float M[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    M[i][0] = fmaf(C[i][2], r[1], -(C[i][1] * r[2]));
    M[i][1] = fmaf(C[i][0], r[2], -(C[i][2] * r[0]));
    M[i][2] = fmaf(C[i][1], r[0], -(C[i][0] * r[1]));
  }
  float dv[6];
#pragma unroll
  for (int k = 0; k < 6; k++) dv[k] = -d2f * v[k];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = i; j < 3; j++) a[fsym(i, j)] = fmaf(e, fmaf(dv[i], v[j], C[i][j]), a[fsym(i, j)]);          // translation block
#pragma unroll
    for (int k = 0; k < 3; k++) a[fsym(i, 3 + k)] = fmaf(e, fmaf(dv[i], v[3 + k], M[i][k]), a[fsym(i, 3 + k)]);   // translation x rotation
  }
  // rotation block of J^T C J: [r]x M (symmetric: six entries)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float rr0 = fmaf(r[1], M[2][k], -(r[2] * M[1][k]));
    const float rr1 = fmaf(r[2], M[0][k], -(r[0] * M[2][k]));
    const float rr2 = fmaf(r[0], M[1][k], -(r[1] * M[0][k]));
    if (k >= 0) a[fsym(3, 3 + k)] = fmaf(e, fmaf(dv[3], v[3 + k], rr0), a[fsym(3, 3 + k)]);
    if (k >= 1) a[fsym(4, 3 + k)] = fmaf(e, fmaf(dv[4], v[3 + k], rr1), a[fsym(4, 3 + k)]);
    if (k >= 2) a[fsym(5, 3 + k)] = fmaf(e, fmaf(dv[5], v[3 + k], rr2), a[fsym(5, 3 + k)]);
  }
  // A = e r y^T: the point-Hessian term (impl2:522-530, 607), z_{3+i}[3+j] = r_i y_j - delta_ij (y . r)
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float er = e * r[i];
#pragma unroll
    for (int j = 0; j < 3; j++) a[FA_BASE + 3 * i + j] = fmaf(er, y[j], a[FA_BASE + 3 * i + j]);
  }
}
// the 37 sums of a lane -> the 43 entries of a partial row (score, g, H row-major), widened
__device__ __forceinline__ void fast_acc_to_row(const float a[NACC_F], double acc[43]) {
#pragma unroll
  for (int k = 0; k < 7; k++) acc[k] = (double)a[k];
  const double tr = ((double)a[FA_BASE] + (double)a[FA_BASE + 4]) + (double)a[FA_BASE + 8];
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double h = (double)a[fsym(i, j)];
      if (i >= 3 && j >= 3) { h += (double)a[FA_BASE + 3 * (i - 3) + (j - 3)]; if (i == j) h -= tr; }
      acc[7 + i * 6 + j] = h;
    }
  }
}

// Neighbour offset `a` (0..2) of probe q for a K-probe search, resolved at compile time in the sweep
// (same tables and order as c_off above).
__host__ __device__ constexpr int probe_off(int K, int q, int a) {
  const int o7[7][3] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1}};
  const int o26[26][3] = {{-1,-1,-1}, {-1,0,-1}, {-1,1,-1}, {0,-1,-1}, {0,0,-1}, {0,1,-1}, {1,-1,-1}, {1,0,-1}, {1,1,-1}, {-1,-1,0}, {0,-1,0}, {1,-1,0}, {-1,0,0}, {1,1,1}, {1,0,1}, {1,-1,1}, {0,1,1}, {0,0,1}, {0,-1,1}, {-1,1,1}, {-1,0,1}, {-1,-1,1}, {1,1,0}, {0,1,0}, {-1,1,0}, {1,0,0}};
  // K == 27: the KDTREE emulation -- every cell of the 3x3x3 block around the point (centre included)
  return K == 1 ? 0 : (K == 7 ? o7[q][a] : (K == 26 ? o26[q][a] : (a == 0 ? q % 3 - 1 : (a == 1 ? (q / 3) % 3 - 1 : q / 9 - 1))));
}

#define Q_GROUP 7                         // probes between queue drains
#define ID_BITS 23                        // queue entry = staging slot << 23 | voxel id
#define WAVES   (SWEEP_THREADS / 64)

// The sweep.  Work decomposition (MI355X-first):
//   work item = one wave-quarter (CHUNK_PTS/4 = 512 consecutive points) of one 2048-point chunk of one active pair, taken by
//     ONE persistent wave from a per-XCD queue; a workgroup is just four such independent waves.
//   phase A (probe, lane = point): transform the points of a super-tile (1, 2 or 4 tiles of 64) in f32, probe their K neighbour
//     cells in the rank-bitmap, and push every hit as a 4-byte entry into the wave's LDS queue (ballot + popcount compaction).
//   phase B (evaluate, lane = hit): lanes pull 64 queue entries at a time -- every lane busy no matter how the
//     hits were distributed over points -- read the staged point (LDS) and the 64-B voxel record, and add the
//     43 f64 terms into per-lane accumulators; the next batch's records are fetched in the middle of the current evaluation.
//   item end: flush the queue tail, fixed-tree wave reduction -> one 44-double partial row per (chunk, quarter).
// The partial rows depend only on (CHUNK_PTS, input order), never on the launch geometry or on which wave ran the item, so
// single and batched runs of one pair are bit-identical.
// -DNDT_TIMELINE: per-phase shader-clock stamps of every work item, summed over all waves into g_tl (read back through
// mi355ndt_debug_timeline; tools/sweep_timeline.py).  Costs ~10 % and is never part of the shipped library.
#ifdef NDT_TIMELINE
__device__ unsigned long long g_tl[16];
#define TL_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); tl[k] += t_ - tl_last; tl_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TL_STAMP(k) do {} while (0)
#endif
struct SweepCtl {               // 9 ints; two of them alternate: the sweep that reads one clears the other
  int n_active;                 // pairs whose next sweep is pending (entries of active_list)
  int next_item[8];             // per-XCD work-item cursors of the sweep
};
#define QUARTERS WAVES          // a chunk is reduced as 4 wave-quarters of CHUNK_PTS/4 points

// Per-instantiation tuning knobs (none of them changes a result bit).  LEAN = three waves per SIMD without spilling: possible
// for DIRECT7 once the mid-evaluation record prefetch (17 VGPRs) and the two-tile probe group are dropped.  Measured
// (tools/sweep_only.py, bench.py): ndt_omp at 1 m / 65,536 pts 1-3 % faster, but at 0.5 m / 131,072 pts (four times the voxel
// records, fewer cache hits) 3 % slower, and ndt_pca 12 % slower there -- the prefetch matters as soon as records miss in L2,
// so LEAN stays off.
#ifndef FAST_WPE7
#define FAST_WPE7 4            // tolerance arithmetic: 37 f32 accumulators instead of 43 f64 ones leave room for four waves per SIMD
#endif
#ifndef FAST_WPE1
#define FAST_WPE1 3            // ... DIRECT1: three (168 registers: no spill; measured 1,338 us per pca / 1 m launch against 1,554 with four)
#endif
#ifndef FAST_TP7
#define FAST_TP7 1
#endif
#ifndef FAST_TP1
#define FAST_TP1 2
#endif
#ifndef FAST_PIPE
#define FAST_PIPE 1
#endif
template <bool PCA, int K, int ORD = 0>
struct SweepTune {
  static constexpr bool FAST = (ORD == 2);
  static constexpr bool LEAN = false;
  static constexpr int  WPE  = FAST ? (K == 1 ? FAST_WPE1 : FAST_WPE7) : (LEAN ? 3 : SWEEP_WPE);                       // workgroups per CU = waves per SIMD
  static constexpr bool PIPE = FAST ? (FAST_PIPE != 0) : !LEAN;                                      // fetch batch k+1's records in the middle of batch k
  // tiles probed together (8 = the whole work item); tolerance arithmetic: small super-tiles keep the LDS of a workgroup under a quarter of the CU's
  static constexpr int  TP   = FAST ? (K == 1 ? FAST_TP1 : FAST_TP7) : (LEAN ? 1 : (K == 1 ? 8 : (K <= 7 ? 2 : 1)));
};
static inline int sweep_wpe(bool pca, int K, bool fast = false) { (void)pca; return fast ? (K == 1 ? FAST_WPE1 : FAST_WPE7) : SWEEP_WPE; }

// IT = tiles of 64 points per work item.  8 is the batch mode described above (a wave-quarter of a 2048-point chunk).
// FINE (latency mode): small items (IT = 1 or 2) dealt statically over ALL waves of the grid, for a sweep over one or a
// few pairs -- a single 65,536-point pair then offers 512-1024 items to the 2,048 resident waves instead of 128, and no wave
// claims anything with an atomic.  A pair's rows are then summed by k_update with the same rule over 4-row chunks of 4 * IT * 64
// points: its own fixed tree, not the batch mode's (results agree to the f64 rounding of the sums, ~1e-16 relative).
// `grid_of` (may be null): pair b is aligned against the target grid gd[grid_of[b]] instead of gd[b] (sequence mode: the frames'
// grids are built once, the keyframe policy picks which one a frame is matched against).
// One work item of the sweep: `IT` tiles of 64 consecutive points (item `rem`) of pair `b`, run by ONE wave -> one 44-double partial row.
// Shared by the lockstep kernel (k_sweep: one launch per Newton round) and the asynchronous one (k_align_async: one launch per align).
// ASYNC: the pair's pose is read, and its row is written, with agent-scope (sc1) accesses -- another workgroup wrote / will read them
// inside the same launch (ndt_async.hpp) -- and the FINE block reduction does not apply.
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// ASYNC: the 21 pose words of pair state S (T: 12, Rj: 9), lane k holding word k -- one agent-scope (L1-bypassing) vector load, never the
// scalar cache: the pair's updater, some other workgroup of the same launch, rewrote them since the pair's previous sweep
__device__ __forceinline__ unsigned sweep_pose_words(const PairState* S) {
  static_assert(offsetof(PairState, T) == 0 && offsetof(PairState, Rj) == 48, "pose words");
  const int lane = threadIdx.x & 63;
  unsigned w = 0;
  if (lane < 21) w = __hip_atomic_load((const gu32*)reinterpret_cast<const unsigned*>(S) + lane, RLX_AGENT);
  return w;
}

template <bool PCA, int NROWS, bool ASYNC>
__device__ __forceinline__ void sweep_rows_d1p(const int b, const int rem0, const float* __restrict__ src, const size_t pitch, const float T[12], const float Rj[9], const int n,
                                               const GridDesc& g, const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs,
                                               double* partials, const int rows_per_pair, const SweepConst& sc
#ifdef NDT_TIMELINE
                                               , unsigned long long* tl, unsigned long long& tl_last
#endif
                                               );
#ifndef FAST_D1_POINT
#define FAST_D1_POINT 0          // tolerance arithmetic, DIRECT1: 1 = lane = point (sweep_rows_d1p) instead of the hit queue -- built, measured, not faster (below)
#endif

template <bool PCA, int K, int IT, bool FINE, int ORD, bool ASYNC>
__device__ __forceinline__ void sweep_item(const int b, const int rem, const float* __restrict__ src, const size_t pitch, const PairState* st,
                                           const GridDesc* __restrict__ gd, const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs,
                                           double* partials, const int rows_per_pair, const SweepConst& sc, const float* __restrict__ cent,
                                           const int* __restrict__ grid_of, const double* exp_tab,
                                           const unsigned pose_w, const int n_async, const int g_async   /* ASYNC only: pose words (sweep_pose_words), point count, grid index */
#ifdef NDT_TIMELINE
                                           , unsigned long long* tl, unsigned long long& tl_last
#endif
                                           ) {
  constexpr bool KD = (K == 27);
  constexpr bool FAST = (ORD == 2);                // tolerance arithmetic (eval_hit_fast): `recs` holds VoxelRecF records
  static_assert(!FAST || K == 1 || K == 7, "tolerance arithmetic is instantiated for DIRECT1 / DIRECT7");
  typedef SweepTune<PCA, K, ORD> Tune;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // ndt_pca's per-hit multiplier is the product of the hit's own weight and those of the point's LATER hits, known only in phase A.
  // With one probe per point (DIRECT1) it is just the leaf's own weight, which phase B reads with the record anyway: no weight
  // load in the probe stage, no weight queue, dead leaves filtered in phase B exactly as for ndt_omp.
  constexpr bool PCAQ = PCA && K > 1;
  // per-wave hit queue: must hold a leftover (< 64) plus everything one probe group can push: TP tiles x min(K, Q_GROUP) probes x 64
  constexpr int TP = Tune::TP < IT ? Tune::TP : IT;
  constexpr int Q_NEED = (IT / TP > 1 ? 64 : 0) + TP * (K < Q_GROUP ? K : Q_GROUP) * 64;
  constexpr int Q_CAP = FAST ? (Q_NEED <= 128 ? 128 : Q_NEED <= 256 ? 256 : Q_NEED <= 512 ? 512 : 1024) : ((K > 1 && K <= Q_GROUP) ? 1024 : 512);
  static_assert(Q_NEED <= Q_CAP, "hit queue");
  __shared__ unsigned q_ent[WAVES][Q_CAP];
  // ndt_pca weight of a queued hit: the suffix product (f64: up to ~150^7) -- for DIRECT1 just the leaf's own integer weight
  typedef typename std::conditional<K == 1, int, typename std::conditional<FAST, float, double>::type>::type QW;
  __shared__ QW q_w[PCAQ ? WAVES : 1][PCAQ ? Q_CAP : 1];
  // TP tiles of 64 points are probed together ("super-tile"): their point transforms, then ALL their bitmap loads, then all
  // their ballots -- the probe stage costs a few L2 round trips per super-tile, not per tile.  DIRECT1 has one probe per point
  // and ~0.9 hits, so it is probe-stage bound: 4 tiles at a time; DIRECT7: 2 (14 bitmap words in flight); the 26/27-cell
  // searches already have 7-probe groups inside one tile.
  constexpr int NBUF = (IT / TP > 1) ? 2 : 1;   // a super-tile that is the whole item needs no second buffer
  __shared__ float stage[WAVES][NBUF * 64 * TP][6];   // (two) super-tiles of staged points: x'(3), R x (3)

  // FINE: the four waves of a block always hold the four rows of ONE chunk (items 4q .. 4q+3: static dealing, four waves per block, a
  // multiple of four items per pair), so the chunk level of k_update's tree -- ((r0 + r1) + r2) + r3 -- is added here through LDS and
  // one row per chunk goes to memory: a quarter of the rows for the update to fetch.
  __shared__ double red[FINE ? WAVES : 1][FINE ? NACC : 1];
  const PairState& S = st[b];
  // ASYNC: T (12 words), Rj (9) and n_src through agent-scope vector loads (never the scalar cache, never a stale L1 line): the
  // updater of this pair -- some other workgroup of this launch -- rewrote them since the pair's previous sweep
  // (the caller issued that load -- sweep_pose_words -- as early as it knew the pair, so that it is in flight together with the point loads
  //  below; the point count comes from the batch's constant count array for the same reason)
  const int n = ASYNC ? n_async : S.n_src;
  const GridDesc& g = gd[ASYNC ? g_async : (grid_of ? grid_of[b] : b)];
  const float* X = src + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const VoxelRec* R = recs + g.rec_off;
  const bool grid_ok = (g.status == GRID_OK);
  float T[12], Rj[9];
#pragma unroll
  for (int a = 0; a < 12; a++) T[a] = ASYNC ? __uint_as_float(__builtin_amdgcn_readlane(pose_w, a)) : S.T[a];
#pragma unroll
  for (int a = 0; a < 9; a++) Rj[a] = ASYNC ? __uint_as_float(__builtin_amdgcn_readlane(pose_w, 12 + a)) : S.Rj[a];
  if constexpr (FAST && K == 1 && IT == 8 && !FINE && FAST_D1_POINT) {
    sweep_rows_d1p<PCA, 1, ASYNC>(b, rem, src, pitch, T, Rj, n, g, words, recs, partials, rows_per_pair, sc
#ifdef NDT_TIMELINE
                                  , tl, tl_last
#endif
                                  );
    return;
  }
  const float leaf = g.leaf;
  const int mb0 = g.min_b[0], mb1 = g.min_b[1], mb2 = g.min_b[2];
  const int xb0 = g.max_b[0], xb1 = g.max_b[1], xb2 = g.max_b[2];
  const int mul1 = g.mul1, mul2 = g.mul2, nwords = g.nwords;
  {
    typedef typename std::conditional<FAST, float, double>::type AccT;
    AccT acc[FAST ? NACC_F : 43];
#pragma unroll
    for (int a = 0; a < (FAST ? NACC_F : 43); a++) acc[a] = (AccT)0;
    unsigned nhits = 0;                              // wave-uniform
    int qhead = 0, qcount = 0;                       // wave-uniform
    int q_old = 0;                                   // queued entries that reference the OTHER staging half (older tile)
    const int wbase = rem * (IT * 64);

    // One batch of queued hits (one per lane): queue entry + the voxel record it points at, in registers.
    struct BatchX { unsigned slot; double m0, m1, m2; float C[9]; int weight; double w; };
    struct BatchF { unsigned slot; float mh[3], ml[3], c[6]; int weight; float w; };
    typedef typename std::conditional<FAST, BatchF, BatchX>::type Batch;
    // read the `m` hits that sit `off` entries behind the queue head; lanes >= m re-read the last entry (and contribute +0)
    auto fetch = [&](int off, int m, Batch& B) {
      const int k = lane < m ? lane : m - 1;
      const unsigned ent = q_ent[wv][(qhead + off + k) & (Q_CAP - 1)];
      B.slot = ent >> ID_BITS;
      if constexpr (FAST) {
        const VoxelRecF& vr = reinterpret_cast<const VoxelRecF*>(R)[ent & ((1u << ID_BITS) - 1)];
#pragma unroll
        for (int a = 0; a < 3; a++) { B.mh[a] = vr.mh[a]; B.ml[a] = vr.ml[a]; }
#pragma unroll
        for (int a = 0; a < 6; a++) B.c[a] = vr.c[a];
        B.weight = vr.weight;
        B.w = 1.f;
        if (PCAQ) B.w = (float)q_w[wv][(qhead + off + k) & (Q_CAP - 1)];
        else if (PCA) B.w = (float)vr.weight;
      } else {
        const VoxelRec& vr = R[ent & ((1u << ID_BITS) - 1)];
        B.m0 = vr.mean[0]; B.m1 = vr.mean[1]; B.m2 = vr.mean[2];
#pragma unroll
        for (int a = 0; a < 9; a++) B.C[a] = vr.icov[a];
        B.weight = vr.weight;
        B.w = 1.0;
        if (PCAQ) B.w = (double)q_w[wv][(qhead + off + k) & (Q_CAP - 1)];
        else if (PCA) B.w = (double)vr.weight;
      }
    };
    // evaluate a fetched batch (running `mid` half way through) and retire its `m` queue entries
    auto eval_batch = [&](const Batch& B, int m, auto mid) {
      const float* sp = stage[wv][B.slot];         // staged point: LDS, short latency
      const float xt0 = sp[0], xt1 = sp[1], xt2 = sp[2];
      float r[3] = {sp[3], sp[4], sp[5]};
      // ndt_omp: leaves with nr_points = -1 (eigen / inverse failure) are not neighbours (impl:395): filtered here
      const bool live = lane < m && (PCAQ || KD || B.weight != VOX_DEAD);
      if constexpr (FAST) {
        float u[3] = {(xt0 - B.mh[0]) - B.ml[0], (xt1 - B.mh[1]) - B.ml[1], (xt2 - B.mh[2]) - B.ml[2]};
        eval_hit_fast<PCA, decltype(mid)>(u, r, B.c, sc.d1f, sc.d2f, sc.kq, B.w, live, acc, mid);
      } else {
        float u[3] = {(float)((double)xt0 - B.m0), (float)((double)xt1 - B.m1), (float)((double)xt2 - B.m2)};   // impl2:276-279, 574
        eval_hit<PCA, decltype(mid), KD, ORD>(u, r, B.C, sc.d1, sc.d2f, B.w, live, acc, exp_tab, mid);
      }
      nhits += PCAQ ? (unsigned)m : (unsigned)__popcll(__ballot(live));
      qhead = (qhead + m) & (Q_CAP - 1);
      qcount -= m;
      q_old = q_old > m ? q_old - m : 0;
    };
    // evaluate `m` queued hits (m <= 64), one per lane
    auto drain = [&](int m) {
      Batch B;
      fetch(0, m, B);
      eval_batch(B, m, NoHook());
    };
    // All full batches in the queue, software-pipelined: the next batch's record loads are issued in the middle of the
    // current batch's arithmetic (before its 36 Hessian terms), so their L2 latency is off the critical path.
    auto drain_full = [&]() {
      if (!Tune::PIPE) {                           // plain: one batch after the other
        while (qcount >= 64) drain(64);
        return;
      }
      if (qcount < 64) return;
      Batch A;
      fetch(0, 64, A);
#pragma unroll 1
      for (;;) {
        const bool more = qcount >= 128;
        Batch N;
        eval_batch(A, 64, [&]() { if (more) fetch(64, 64, N); });
        if (!more) break;
        A = N;
      }
    };
    if (wbase < n && grid_ok) {
      constexpr int NST = IT / TP;                            // super-tiles per item
      const unsigned e0 = (unsigned)(xb0 - mb0), e1 = (unsigned)(xb1 - mb1), e2 = (unsigned)(xb2 - mb2);
      const unsigned empty_cell = (unsigned)(nwords - 1) << 6;
      // points of the next super-tile are fetched one super-tile ahead (HBM latency ~2 us would otherwise be exposed each time)
      float nx[TP], ny[TP], nz[TP];
#pragma unroll
      for (int p = 0; p < TP; p++) {
        const int i = wbase + p * 64 + lane;
        nx[p] = ny[p] = nz[p] = 0.f;
        if (i < n) { nx[p] = X[i]; ny[p] = X[pitch + i]; nz[p] = X[2 * pitch + i]; }
      }
      TL_STAMP(1);
#pragma unroll 1
      for (int st = 0; st < NST; st++) {
        if (wbase + st * TP * 64 >= n) break;        // wave-uniform
        // the staging area holds two super-tiles: entries of super-tile st-2 must be gone before st overwrites their half
        // (only happens when hits are sparse; dense tiles are consumed by the regular 64-wide drains)
        if (q_old > 0) { __builtin_amdgcn_wave_barrier(); drain(q_old); }
        q_old = qcount;
        int r0[TP], r1[TP], r2[TP], cc[TP];
        bool valid[TP];
        float kx[TP], ky[TP], kz[TP];                // moved points (only the KDTREE distance test reads them again)
#pragma unroll
        for (int p = 0; p < TP; p++) {
          const int i = wbase + (st * TP + p) * 64 + lane;
          const float px = nx[p], py = ny[p], pz = nz[p];
          const int inext = i + TP * 64;
          if (st + 1 < NST && inext < n) { nx[p] = X[inext]; ny[p] = X[pitch + inext]; nz[p] = X[2 * pitch + inext]; }
          bool ok = i < n && finite3(px, py, pz);
          // PCL 1.8 transformPointCloud scalar form; Jacobian point r = R x (impl2:507-508)
          float xt[3], r[3];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            xt[a] = ((T[a * 4 + 0] * px + T[a * 4 + 1] * py) + T[a * 4 + 2] * pz) + T[a * 4 + 3];
            r[a] = (Rj[a * 3 + 0] * px + Rj[a * 3 + 1] * py) + Rj[a * 3 + 2] * pz;
          }
          // a non-finite moved point (NaN pose: only reachable through a NaN More-Thuente trial value) has no neighbours;
          // the reference's float->int cast is undefined there
          ok = ok && finite3(xt[0], xt[1], xt[2]);
          float* sp = stage[wv][((st & 1) * TP + p) * 64 + lane];
          sp[0] = xt[0]; sp[1] = xt[1]; sp[2] = xt[2]; sp[3] = r[0]; sp[4] = r[1]; sp[5] = r[2];
          kx[p] = xt[0]; ky[p] = xt[1]; kz[p] = xt[2];
          // getNeighborhoodAtPoint (voxel_grid_covariance_omp_impl.hpp:379-399): cell of the point, f32 divide
          // (x / 2^k is the same bits as x * 2^-k, so a power-of-two leaf takes the one-instruction path)
          const int c0 = (int)floorf(sc.leaf_pow2 ? xt[0] * sc.inv_leaf : xt[0] / leaf);
          const int c1 = (int)floorf(sc.leaf_pow2 ? xt[1] * sc.inv_leaf : xt[1] / leaf);
          const int c2 = (int)floorf(sc.leaf_pow2 ? xt[2] * sc.inv_leaf : xt[2] / leaf);
          // Branch-free probe stage.  Relative cell r = c - min_b; "inside the grid" (impl:382-392) is one unsigned
          // compare per axis; a probe that falls outside (or belongs to an invalid lane) is redirected to the grid's
          // spare all-zero bitmap word, so it misses without any flag having to be kept.
          r0[p] = c0 - mb0; r1[p] = c1 - mb1; r2[p] = c2 - mb2;
          cc[p] = r0[p] + r1[p] * mul1 + r2[p] * mul2;
          valid[p] = ok;
        }
        // probes run last-to-first so the ndt_pca weight of a hit (product of its own and all LATER hits' weights,
        // ndt_pca_impl2.hpp:295-296) is a running product; the order of the f64 additions is free anyway.
        TL_STAMP(2);
        double suf[TP];
#pragma unroll
        for (int p = 0; p < TP; p++) suf[p] = 1.0;
        // Q_GROUP probes of every tile of the super-tile at a time: all their bitmap loads in flight together (then all
        // ndt_pca weight loads), then the ballots -- one L2 round trip per stage.
#pragma unroll
        for (int q1 = K; q1 > 0; q1 -= Q_GROUP) {      // compile-time groups: 1 for DIRECT1 / DIRECT7, 4 for DIRECT26 / KDTREE
          unsigned cellv[TP][Q_GROUP];
          uint4 bwv[TP][Q_GROUP];                      // BitWord: bits lo, bits hi, prefix, pad
#pragma unroll
          for (int p = 0; p < TP; p++) {
#pragma unroll
            for (int j = 0; j < Q_GROUP; j++) {
              const int q = q1 - 1 - j;               // compile-time
              if (q < 0) continue;
              const int o0 = probe_off(K, q, 0), o1 = probe_off(K, q, 1), o2 = probe_off(K, q, 2);
              const bool inside = valid[p] && (unsigned)(r0[p] + o0) <= e0 && (unsigned)(r1[p] + o1) <= e1 && (unsigned)(r2[p] + o2) <= e2;
              cellv[p][j] = inside ? (unsigned)(cc[p] + o0 + o1 * mul1 + o2 * mul2) : empty_cell;
              bwv[p][j] = *reinterpret_cast<const uint4*>(W + (cellv[p][j] >> 6));
            }
          }
          TL_STAMP(3);
          unsigned idv[TP][Q_GROUP];
          int wiv[TP][Q_GROUP];
#pragma unroll
          for (int p = 0; p < TP; p++) {
#pragma unroll
            for (int j = 0; j < Q_GROUP; j++) {
              if (q1 - 1 - j < 0) continue;
              // shift the cell's bit to the top: sign = occupied, popcount = bits at or below it
              const unsigned long long bits = ((unsigned long long)bwv[p][j].y << 32) | bwv[p][j].x;
              const unsigned long long tb = bits << (63u - (cellv[p][j] & 63u));
              idv[p][j] = bwv[p][j].z + (unsigned)__popcll(tb) - 1u;   // rank among the searchable leaves = voxel id
              // occupied <=> the cell's bit (now the sign bit) is set; ndt_pca needs the weights now (suffix product),
              // ndt_omp filters dead leaves in phase B instead and saves this dependent L2 round trip
              wiv[p][j] = ((long long)tb < 0) ? 1 : VOX_DEAD;
              if (PCAQ) { if ((long long)tb < 0) wiv[p][j] = R[idv[p][j]].weight; }
              if (KD && (long long)tb < 0) {                     // FLANN L2_Simple distance to the leaf's f32 centroid
                const float* cp = cent + 3 * (size_t)(g.rec_off + idv[p][j]);
                const float dx = kx[p] - cp[0], dy = ky[p] - cp[1], dz = kz[p] - cp[2];
                if (!(((dx * dx + dy * dy) + dz * dz) < sc.kd_r2)) wiv[p][j] = VOX_DEAD;
              }
            }
          }
#pragma unroll
          for (int p = 0; p < TP; p++) {
            const int slot = ((st & 1) * TP + p) * 64 + lane;
#pragma unroll
            for (int j = 0; j < Q_GROUP; j++) {
              if (q1 - 1 - j < 0) continue;
              const bool hit = wiv[p][j] != VOX_DEAD;     // empty cell, or nr_points == -1: not a neighbour (impl:395)
              if (PCAQ && hit) suf[p] *= (double)wiv[p][j];
              const unsigned long long mask = __ballot(hit);
              if (hit) {
                const int pos = (qhead + qcount + (int)__popcll(mask & lt_mask)) & (Q_CAP - 1);
                q_ent[wv][pos] = ((unsigned)slot << ID_BITS) | idv[p][j];
                if (PCAQ) q_w[wv][pos] = (QW)suf[p];
              }
              qcount += (int)__popcll(mask);
            }
          }
          __builtin_amdgcn_wave_barrier();
          TL_STAMP(4);
          drain_full();
          TL_STAMP(5);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (qcount > 0) drain(qcount);
      TL_STAMP(5);
    }
    // Fixed-order reduction of the wave -> one 44-double row per (chunk, quarter).  Same pairwise tree as a 64-lane xor
    // butterfly (distance 32, 16, 8, 4, 2, 1 -- so the same bits), but as a reduce-scatter: the distance-32 and -16
    // levels use gfx950's v_permlane32_swap / v_permlane16_swap to exchange HALF of the values between lane halves /
    // rows, so 43 -> 22 -> 11 values remain per lane before the in-row butterfly (66 swaps + 88 shuffles + 77 adds
    // instead of 516 shuffles + 258 adds per item).
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    double accd[43];
    if constexpr (FAST) fast_acc_to_row(acc, accd);
    else {
#pragma unroll
      for (int i = 0; i < 43; i++) accd[i] = acc[i];
    }
    double P1[22], P2[11];
#pragma unroll
    for (int i = 0; i < 22; i++) {
      const double a = accd[i], b2 = (i + 22 < 43) ? accd[i + 22] : 0.0;
      const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b2), false, false);
      const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b2), false, false);
      // lanes 0..31: value i of lanes L and L+32;  lanes 32..63: value i+22 of lanes L-32 and L
      P1[i] = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
    }
#pragma unroll
    for (int i = 0; i < 11; i++) {
      const double a = P1[i], b2 = P1[i + 11];
      const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b2), false, false);
      const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b2), false, false);
      // even rows: P1[i] of rows r and r+1;  odd rows: P1[i+11] of rows r-1 and r
      double v = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
      P2[i] = v;
    }
    if ((lane & 15) == 0) {
      // row 0 (lane 0) holds values 0..10, row 1: 11..21, row 2: 22..32, row 3: 33..42 (+ the pad)
      const int row = lane >> 4, base = 11 * (row & 1) + 22 * (row >> 1);
      double* P = FINE ? &red[FINE ? wv : 0][0] : partials + ((size_t)b * rows_per_pair + rem) * NACC;
      if (ASYNC && !FINE) {                        // write-through (sc1) 8-byte stores: the pair's updater may run on another XCD
        gu64* PG = (gu64*)reinterpret_cast<unsigned long long*>(P);
#pragma unroll
        for (int i = 0; i < 11; i++) if (base + i < 43) __hip_atomic_store(PG + base + i, (unsigned long long)__double_as_longlong(P2[i]), RLX_AGENT);
        if (lane == 0) __hip_atomic_store(PG + 43, (unsigned long long)__double_as_longlong((double)nhits), RLX_AGENT);
      } else {
#pragma unroll
        for (int i = 0; i < 11; i++) if (base + i < 43) P[base + i] = P2[i];
        if (lane == 0) P[43] = (double)nhits;
      }
    }
    if (FINE) {
      __syncthreads();                              // (uniform: the four waves of a block run the same items loop, see above)
      if (wv == 0 && lane < NACC)
        partials[((size_t)b * (rows_per_pair >> 2) + (rem >> 2)) * NACC + lane] = ((red[0][lane] + red[FINE ? 1 : 0][lane]) + red[FINE ? 2 : 0][lane]) + red[FINE ? 3 : 0][lane];
      __syncthreads();
    }
    TL_STAMP(6);
#ifdef NDT_TIMELINE
    tl[7] += 1;
#endif
  }
}

// DIRECT1 in the one-launch align: NROWS consecutive work items of ONE pair (the two items of a claim, ndt_async.hpp) as one software
// pipeline.  With one probe per point and ~0.85 hits, a DIRECT1 item is a third evaluation and two thirds waiting for the chain
// points -> bitmap word -> record (timeline build, pca / 1 m: point wait + transform 11 %, bitmap wait + push 18 %, reduction 10 % of an item).
// sweep_item probes the whole item at once -- every wait of that chain is exposed once per item.  Here a row's eight tiles go through in four
// super-tiles of two: while super-tile s is evaluated, the bitmap words of s + 1 and the points of s + 2 are in flight, and the first
// super-tile of the NEXT row is probed under this row's reduction.  What is evaluated, and in which 64-hit batches, is exactly what
// sweep_item<PCA, 1, 8> does -- hits enter the queue in point order, full batches leave it in FIFO order, the tail is flushed at the end of
// every row -- so the rows carry the same bits (tests/test_gpu_configs.py and tests/test_stream_gpu.py hold this function, which the
// one-launch align uses, against the round-based kernels, which use sweep_item).
// The staging area holds one row (four super-tiles); a super-tile's slot is free again when its row has been flushed.
template <bool PCA, int ORD, int NROWS>
__device__ __forceinline__ void sweep_rows_d1(const int b, const int rem0, const float* __restrict__ src, const size_t pitch,
                                              const GridDesc* __restrict__ gd, const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs,
                                              double* partials, const int rows_per_pair, const SweepConst& sc, const double* exp_tab,
                                              const unsigned pose_w, const int n, const int g_idx
#ifdef NDT_TIMELINE
                                              , unsigned long long* tl, unsigned long long& tl_last
#endif
                                              ) {
  constexpr int TP = 2, IT = 8, NST = IT / TP, Q_CAP = 512;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  __shared__ unsigned q_ent[WAVES][Q_CAP];
  __shared__ float stage[WAVES][64 * IT][6];       // one row of staged points: x'(3), R x (3)
  const GridDesc& g = gd[g_idx];
  const float* X = src + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const VoxelRec* R = recs + g.rec_off;
  const bool grid_ok = (g.status == GRID_OK);
  float T[12], Rj[9];
#pragma unroll
  for (int a = 0; a < 12; a++) T[a] = __uint_as_float(__builtin_amdgcn_readlane(pose_w, a));
#pragma unroll
  for (int a = 0; a < 9; a++) Rj[a] = __uint_as_float(__builtin_amdgcn_readlane(pose_w, 12 + a));
  const float leaf = g.leaf;
  const int mb0 = g.min_b[0], mb1 = g.min_b[1], mb2 = g.min_b[2];
  const unsigned e0 = (unsigned)(g.max_b[0] - mb0), e1 = (unsigned)(g.max_b[1] - mb1), e2 = (unsigned)(g.max_b[2] - mb2);
  const int mul1 = g.mul1, mul2 = g.mul2;
  const int base = rem0 * (IT * 64);               // first point of the first row
  constexpr int S_TOTAL = NROWS * NST;

  double acc[43];
#pragma unroll
  for (int a = 0; a < 43; a++) acc[a] = 0.0;
  unsigned nhits = 0;
  int qhead = 0, qcount = 0;                       // wave-uniform

  struct Batch { unsigned slot; double m0, m1, m2; float C[9]; int weight; };
  auto fetch = [&](int off, int m, Batch& B) {
    const int k = lane < m ? lane : m - 1;
    const unsigned ent = q_ent[wv][(qhead + off + k) & (Q_CAP - 1)];
    B.slot = ent >> ID_BITS;
    const VoxelRec& vr = R[ent & ((1u << ID_BITS) - 1)];
    B.m0 = vr.mean[0]; B.m1 = vr.mean[1]; B.m2 = vr.mean[2];
#pragma unroll
    for (int a = 0; a < 9; a++) B.C[a] = vr.icov[a];
    B.weight = vr.weight;
  };
  auto eval_batch = [&](const Batch& B, int m, auto mid) {
    const float* sp = stage[wv][B.slot];
    const float xt0 = sp[0], xt1 = sp[1], xt2 = sp[2];
    float r[3] = {sp[3], sp[4], sp[5]};
    const bool live = lane < m && B.weight != VOX_DEAD;
    float u[3] = {(float)((double)xt0 - B.m0), (float)((double)xt1 - B.m1), (float)((double)xt2 - B.m2)};   // impl2:276-279, 574
    eval_hit<PCA, decltype(mid), false, ORD>(u, r, B.C, sc.d1, sc.d2f, PCA ? (double)B.weight : 1.0, live, acc, exp_tab, mid);
    nhits += (unsigned)__popcll(__ballot(live));
    qhead = (qhead + m) & (Q_CAP - 1);
    qcount -= m;
  };
  auto drain_full = [&]() {                        // all full batches, the next batch's records fetched in the middle of the current one (as sweep_item)
    if (qcount < 64) return;
    Batch A;
    fetch(0, 64, A);
#pragma unroll 1
    for (;;) {
      const bool more = qcount >= 128;
      Batch N;
      eval_batch(A, 64, [&]() { if (more) fetch(64, 64, N); });
      if (!more) break;
      A = N;
    }
  };

  // pipeline registers: points of the super-tile after the probed one; probe words of the probed one
  float nx[TP], ny[TP], nz[TP];
  uint4 bw[TP];
  unsigned cellv[TP];
  auto load_points = [&](const int s) {
#pragma unroll
    for (int p = 0; p < TP; p++) {
      const int i = base + (s * TP + p) * 64 + lane;
      nx[p] = ny[p] = nz[p] = 0.f;
      if (s < S_TOTAL && i < n) { nx[p] = X[i]; ny[p] = X[pitch + i]; nz[p] = X[2 * pitch + i]; }
    }
  };
  // transform super-tile s (its points are in nx..), stage it, and issue its bitmap loads
  auto probe = [&](const int s) {
#pragma unroll
    for (int p = 0; p < TP; p++) {
      const int i = base + (s * TP + p) * 64 + lane;
      const float px = nx[p], py = ny[p], pz = nz[p];
      bool ok = grid_ok && i < n && finite3(px, py, pz);
      float xt[3], r[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        xt[a] = ((T[a * 4 + 0] * px + T[a * 4 + 1] * py) + T[a * 4 + 2] * pz) + T[a * 4 + 3];
        r[a] = (Rj[a * 3 + 0] * px + Rj[a * 3 + 1] * py) + Rj[a * 3 + 2] * pz;
      }
      ok = ok && finite3(xt[0], xt[1], xt[2]);
      float* sp = stage[wv][(((s % NST) * TP) + p) * 64 + lane];
      sp[0] = xt[0]; sp[1] = xt[1]; sp[2] = xt[2]; sp[3] = r[0]; sp[4] = r[1]; sp[5] = r[2];
      const int c0 = (int)floorf(sc.leaf_pow2 ? xt[0] * sc.inv_leaf : xt[0] / leaf);
      const int c1 = (int)floorf(sc.leaf_pow2 ? xt[1] * sc.inv_leaf : xt[1] / leaf);
      const int c2 = (int)floorf(sc.leaf_pow2 ? xt[2] * sc.inv_leaf : xt[2] / leaf);
      const int r0 = c0 - mb0, r1 = c1 - mb1, r2 = c2 - mb2;
      const bool inside = ok && (unsigned)r0 <= e0 && (unsigned)r1 <= e1 && (unsigned)r2 <= e2;
      // (a probe that falls outside the grid -- or belongs to an invalid lane, or to a target without a grid -- reads nothing and misses)
      cellv[p] = inside ? (unsigned)(r0 + r1 * mul1 + r2 * mul2) : 0u;
      bw[p] = make_uint4(0u, 0u, 0u, 0u);
      if (inside) bw[p] = *reinterpret_cast<const uint4*>(W + (cellv[p] >> 6));
    }
  };
  // rank the probed super-tile's cells and push its hits (point order: tile, then lane)
  auto push = [&](const int s) {
#pragma unroll
    for (int p = 0; p < TP; p++) {
      const unsigned long long bits = ((unsigned long long)bw[p].y << 32) | bw[p].x;
      const unsigned long long tb = bits << (63u - (cellv[p] & 63u));
      const unsigned id = bw[p].z + (unsigned)__popcll(tb) - 1u;
      const bool hit = (long long)tb < 0;
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int pos = (qhead + qcount + (int)__popcll(mask & lt_mask)) & (Q_CAP - 1);
        q_ent[wv][pos] = ((unsigned)((((s % NST) * TP) + p) * 64 + lane) << ID_BITS) | id;
      }
      qcount += (int)__popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
  };

  load_points(0);
  TL_STAMP(1);
  probe(0);
  load_points(1);
  TL_STAMP(2);
#pragma unroll 1
  for (int s = 0; s < S_TOTAL; s++) {
    const bool row_end = (s % NST) == NST - 1;
    push(s);
    TL_STAMP(4);
    if (!row_end) { probe(s + 1); load_points(s + 2); }        // their memory round trips ride under the evaluation below
    drain_full();
    TL_STAMP(5);
    if (!row_end) continue;
    // ---- the row is complete: flush the tail, then (the staging area is free) probe the next row's first super-tile under the reduction
    __builtin_amdgcn_wave_barrier();
    if (qcount > 0) { Batch Bt; fetch(0, qcount, Bt); eval_batch(Bt, qcount, NoHook()); }
    TL_STAMP(5);
    if (s + 1 < S_TOTAL) { probe(s + 1); load_points(s + 2); }
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    double P1[22], P2[11];
#pragma unroll
    for (int i = 0; i < 22; i++) {
      const double a = acc[i], b2 = (i + 22 < 43) ? acc[i + 22] : 0.0;
      const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b2), false, false);
      const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b2), false, false);
      P1[i] = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
    }
#pragma unroll
    for (int i = 0; i < 11; i++) {
      const double a = P1[i], b2 = P1[i + 11];
      const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b2), false, false);
      const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b2), false, false);
      double v = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
      P2[i] = v;
    }
    if ((lane & 15) == 0) {
      const int row = lane >> 4, rb = 11 * (row & 1) + 22 * (row >> 1);
      gu64* PG = (gu64*)reinterpret_cast<unsigned long long*>(partials + ((size_t)b * rows_per_pair + rem0 + s / NST) * NACC);
#pragma unroll
      for (int i = 0; i < 11; i++) if (rb + i < 43) __hip_atomic_store(PG + rb + i, (unsigned long long)__double_as_longlong(P2[i]), RLX_AGENT);
      if (lane == 0) __hip_atomic_store(PG + 43, (unsigned long long)__double_as_longlong((double)nhits), RLX_AGENT);
    }
#pragma unroll
    for (int a = 0; a < 43; a++) acc[a] = 0.0;
    nhits = 0;
    TL_STAMP(6);
#ifdef NDT_TIMELINE
    tl[7] += 1;
#endif
  }
}

// EXPERIMENT, off by default (-DFAST_D1_POINT=1): DIRECT1 under the tolerance arithmetic with lane = POINT, no hit queue, no staging, no LDS at all.
// Measured (round 6, docs/experiments.md 11): pca / 1 m 1,402 us per launch at four waves per SIMD (spills), 1,376 at three, against 1,338 for the hit
// queue at three; config 5 / DIRECT1 2,917 / 2,711 against 2,395 -- the idle lanes cost more than the queue, and the launch is bound by the vector
// L1's misses in flight for the point stream (TCP busy 98 %, 55-66 % of that stalled on L2, VALU busy 0.6), which more loads in flight per wave
// do not raise.  One probe per point and ~0.85 hits leave a
// DIRECT1 item a third evaluation and two thirds waiting for the chain  points -> bitmap word -> record  (sweep_rows_d1 above); with 37 f32
// sums instead of 43 f64 ones the registers are there to keep three tiles of 64 points in flight per wave instead -- one whose points are
// being transformed and probed, one whose record loads are out, one being evaluated -- and four waves per SIMD, which the queue's LDS
// (48 KB of staging per workgroup) never allowed.  A lane whose point has no leaf evaluates with e = 0 (11 % of the lanes at 1 m, 19 % at
// 0.5 m): cheaper than compacting.  The NROWS consecutive items of a claim go through as ONE stream of tiles; a row's sums are reduced and
// written when its eighth tile has been evaluated, under the loads of the next row's first tiles.  Rows depend on (pair, row, input order)
// alone.  Used by both the round-based sweep (NROWS = 1) and the one-launch align, so a pair's bits are the same in either.
template <bool PCA, int NROWS, bool ASYNC>
__device__ __forceinline__ void sweep_rows_d1p(const int b, const int rem0, const float* __restrict__ src, const size_t pitch, const float T[12], const float Rj[9], const int n,
                                               const GridDesc& g, const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs,
                                               double* partials, const int rows_per_pair, const SweepConst& sc
#ifdef NDT_TIMELINE
                                               , unsigned long long* tl, unsigned long long& tl_last
#endif
                                               ) {
  constexpr int IT = 8, S_TOTAL = NROWS * IT;
  const int lane = threadIdx.x & 63;
  const float* X = src + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const VoxelRecF* R = reinterpret_cast<const VoxelRecF*>(recs) + g.rec_off;
  const bool grid_ok = (g.status == GRID_OK);
  const float leaf = g.leaf;
  const int mb0 = g.min_b[0], mb1 = g.min_b[1], mb2 = g.min_b[2];
  const unsigned e0 = (unsigned)(g.max_b[0] - mb0), e1 = (unsigned)(g.max_b[1] - mb1), e2 = (unsigned)(g.max_b[2] - mb2);
  const int mul1 = g.mul1, mul2 = g.mul2;
  const int base = rem0 * (IT * 64);

  float acc[NACC_F];
#pragma unroll
  for (int a = 0; a < NACC_F; a++) acc[a] = 0.f;
  unsigned nhits = 0;
  // stage registers: P = raw points (prefetched), A = transformed + probed (bitmap word in flight), B = ranked (record in flight)
  float px = 0.f, py = 0.f, pz = 0.f;
  float axt[3] = {0.f, 0.f, 0.f}, ar[3] = {0.f, 0.f, 0.f}; unsigned acell = 0u; bool ain = false; uint4 abw = make_uint4(0u, 0u, 0u, 0u);
  float bxt[3] = {0.f, 0.f, 0.f}, br[3] = {0.f, 0.f, 0.f}; bool bhit = false;
  float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0; int bweight = 0;
  auto load_points = [&](const int s) {
    const int i = base + s * 64 + lane;
    px = py = pz = 0.f;
    if (s < S_TOTAL && i < n) { px = X[i]; py = X[pitch + i]; pz = X[2 * pitch + i]; }
  };
  auto probe = [&](const int s) {                    // P -> A
    const int i = base + s * 64 + lane;
    bool ok = grid_ok && s < S_TOTAL && i < n && finite3(px, py, pz);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      axt[a] = ((T[a * 4 + 0] * px + T[a * 4 + 1] * py) + T[a * 4 + 2] * pz) + T[a * 4 + 3];     // PCL 1.8 transformPointCloud, uncontracted
      ar[a] = fmaf(Rj[a * 3 + 2], pz, fmaf(Rj[a * 3 + 1], py, Rj[a * 3 + 0] * px));
    }
    ok = ok && finite3(axt[0], axt[1], axt[2]);
    const int c0 = (int)floorf(sc.leaf_pow2 ? axt[0] * sc.inv_leaf : axt[0] / leaf);
    const int c1 = (int)floorf(sc.leaf_pow2 ? axt[1] * sc.inv_leaf : axt[1] / leaf);
    const int c2 = (int)floorf(sc.leaf_pow2 ? axt[2] * sc.inv_leaf : axt[2] / leaf);
    const int r0 = c0 - mb0, r1 = c1 - mb1, r2 = c2 - mb2;
    ain = ok && (unsigned)r0 <= e0 && (unsigned)r1 <= e1 && (unsigned)r2 <= e2;
    acell = ain ? (unsigned)(r0 + r1 * mul1 + r2 * mul2) : 0u;
    abw = make_uint4(0u, 0u, 0u, 0u);
    if (ain) abw = *reinterpret_cast<const uint4*>(W + (acell >> 6));
  };
  auto rank = [&]() {                                // A -> B
    const unsigned long long bits = ((unsigned long long)abw.y << 32) | abw.x;
    const unsigned long long tb = bits << (63u - (acell & 63u));
    const unsigned id = abw.z + (unsigned)__popcll(tb) - 1u;
    bhit = ain && (long long)tb < 0;
#pragma unroll
    for (int a = 0; a < 3; a++) { bxt[a] = axt[a]; br[a] = ar[a]; }
    if (bhit) {
      const float4* rp = reinterpret_cast<const float4*>(R + id);
      q0 = rp[0]; q1 = rp[1]; q2 = rp[2];
      bweight = R[id].weight;
    }
  };
  load_points(0);
  TL_STAMP(1);
  probe(0);
  load_points(1);
  TL_STAMP(2);
  rank();
  probe(1);
  load_points(2);
  TL_STAMP(4);
#pragma unroll 1
  for (int s = 0; s < S_TOTAL; s++) {
    // in flight here: record(s), bitmap word(s + 1), points(s + 2)
    float ext[3], er[3]; bool ehit; float4 e0q, e1q, e2q; int ew;
    { ext[0] = bxt[0]; ext[1] = bxt[1]; ext[2] = bxt[2]; er[0] = br[0]; er[1] = br[1]; er[2] = br[2]; ehit = bhit; e0q = q0; e1q = q1; e2q = q2; ew = bweight; }
    rank();                                          // tile s + 1: needs its bitmap word; issues its record loads
    probe(s + 2);                                    // tile s + 2: needs its points; issues its bitmap load
    load_points(s + 3);
    {                                                // tile s: evaluate (its record arrived while the loads above went out)
      const bool live = ehit && ew != VOX_DEAD;
      const float u[3] = {(ext[0] - e0q.x) - e0q.w, (ext[1] - e0q.y) - e1q.x, (ext[2] - e0q.z) - e1q.y};
      const float c[6] = {e1q.z, e1q.w, e2q.x, e2q.y, e2q.z, e2q.w};
      eval_hit_fast<PCA, NoHook>(u, er, c, sc.d1f, sc.d2f, sc.kq, PCA ? (float)ew : 1.f, live, acc);
      nhits += (unsigned)__popcll(__ballot(live));
    }
    TL_STAMP(5);
    if ((s % IT) != IT - 1) continue;
    // ---- the row is complete
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    double accd[43];
    fast_acc_to_row(acc, accd);
    double P1[22], P2[11];
#pragma unroll
    for (int i = 0; i < 22; i++) {
      const double a = accd[i], b2 = (i + 22 < 43) ? accd[i + 22] : 0.0;
      const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b2), false, false);
      const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b2), false, false);
      P1[i] = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
    }
#pragma unroll
    for (int i = 0; i < 11; i++) {
      const double a = P1[i], b2 = P1[i + 11];
      const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b2), false, false);
      const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b2), false, false);
      double v = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
      P2[i] = v;
    }
    if ((lane & 15) == 0) {
      const int row = lane >> 4, rb = 11 * (row & 1) + 22 * (row >> 1);
      double* P = partials + ((size_t)b * rows_per_pair + rem0 + s / IT) * NACC;
      if (ASYNC) {
        gu64* PG = (gu64*)reinterpret_cast<unsigned long long*>(P);
#pragma unroll
        for (int i = 0; i < 11; i++) if (rb + i < 43) __hip_atomic_store(PG + rb + i, (unsigned long long)__double_as_longlong(P2[i]), RLX_AGENT);
        if (lane == 0) __hip_atomic_store(PG + 43, (unsigned long long)__double_as_longlong((double)nhits), RLX_AGENT);
      } else {
#pragma unroll
        for (int i = 0; i < 11; i++) if (rb + i < 43) P[rb + i] = P2[i];
        if (lane == 0) P[43] = (double)nhits;
      }
    }
#pragma unroll
    for (int a = 0; a < NACC_F; a++) acc[a] = 0.f;
    nhits = 0;
    TL_STAMP(6);
#ifdef NDT_TIMELINE
    tl[7] += 1;
#endif
  }
}

// (ndt_update.hpp, which includes this file)
__device__ __forceinline__ void newton_rebase(const double p[6], const double dir[6], const double a_t, double pn[6], float inc_cm[16]);

template <bool PCA, int K, int IT = 8, bool FINE = false, int ORD = 0>
__global__ void __launch_bounds__(SWEEP_THREADS, (SweepTune<PCA, K, ORD>::WPE))
k_sweep(const float* __restrict__ src, size_t pitch, const PairState* __restrict__ st,
        const GridDesc* __restrict__ gd, const BitWord* __restrict__ words, const VoxelRec* __restrict__ recs,
        double* partials, int rows_per_pair, const int* __restrict__ active_list, SweepCtl* ctl, SweepCtl* ctl_next, SweepConst sc,
        const float* __restrict__ cent, const int* __restrict__ grid_of) {
  // K == 27 is the KDTREE mode (ndt_omp_impl2.hpp:251-253): radiusSearch(point, resolution) over the f32 centroids of the
  // searchable leaves (voxel_grid_covariance_omp.h:505-534).  A centroid lies inside its own cell, so every centroid closer
  // than one leaf sits in the 3x3x3 block around the point's cell: probe those 27 cells and keep d^2 < float(r*r).  The
  // reference does not re-check nr_points there, so eigen/inverse-failed leaves DO take part (their icov is zero / non-finite).
  // Persistent waves pulling work items.  One item = one wave-quarter (CHUNK_PTS/4 consecutive points) of one chunk
  // of one active pair; every wave is independent (own LDS queue, own partial row, no block barrier), so a wave
  // whose points have few hits simply takes the next item instead of idling at a barrier.
  // Items are queued per XCD: pair slot a of the active list belongs to XCD a % 8 (workgroup L is observed to run on
  // XCD L % 8, MI355X_MICROARCH.md), so one pair's records / bitmap / points stay in one L2; a wave whose XCD
  // queue is empty steals from the others.  Which wave runs an item never changes the item's result.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n_active = ctl->n_active;
  const int items_per_pair = rows_per_pair;         // one partial row per work item

  __shared__ double exp_tab[64];                   // 2^(j/64) for ndtm::exp_f32arg
  if (threadIdx.x < 64) exp_tab[threadIdx.x] = ndtm::c_exp2_64[threadIdx.x];
  __syncthreads();                                 // (the only block barrier of the kernel, before the persistent loop)
  // the control block of the NEXT round (k_update fills it after this kernel) is cleared here, not by a host memset
  if (blockIdx.x == 0 && threadIdx.x < 9) reinterpret_cast<int*>(ctl_next)[threadIdx.x] = 0;
  if (FINE && sc.host_flags && blockIdx.x == 0 && threadIdx.x == 0) {     // progress report of the latency-mode pump (posted writes over PCIe)
    sc.host_flags[1] = n_active;
    __threadfence_system();
    sc.host_flags[0] = sc.seq_no;
  }
  if (n_active == 0) return;                       // nothing left to sweep (the loop's last, empty round)
  // Latency mode: the re-basing of p that the NEXT Newton update starts with -- log(exp(delta_p) exp(p)) and float(exp(delta_p)), impl2:163-166 --
  // depends only on what the previous update decided (p, dir, a_t), not on this sweep.  One extra workgroup computes it while the others
  // sweep (two SE(3) exponentials and a logarithm: ~9 k cycles of a serial f64 chain that used to sit on the update's critical path); the
  // update picks it up behind the tag (= the pair's sweep count), exactly as the one-launch align does (ndt_async.hpp).
  const int nblk = FINE && sc.rebase_block ? (int)gridDim.x - 1 : (int)gridDim.x;
  if (FINE && sc.rebase_block && (int)blockIdx.x == nblk) {
    if (wv != 0) return;
    for (int a = 0; a < n_active; a++) {
      PairState& S = const_cast<PairState&>(st[active_list[a]]);
      if (S.phase != PH_STEP) continue;              // (wave-uniform: the first sweep of an align has no step behind it)
      double pn[6]; float inc[16];
      newton_rebase(S.p, S.dir, S.a_t, pn, inc);
      if (lane == 0) {
        for (int k = 0; k < 6; k++) S.reb_pn[k] = pn[k];
        for (int k = 0; k < 16; k++) S.reb_inc[k] = inc[k];
        S.reb_tag = (long long)S.sweeps;
      }
    }
    return;
  }
  const int my_xcd = blockIdx.x & 7;
  // Items of a queue are handed out in two ways.  The first `n_static` rounds are STATIC: wave `wx` of the XCD's `xw` waves takes
  // items wx, wx + xw, ...; only the rest of the queue is claimed with an atomic.  Why: VMEM operations of a wave complete in
  // order (vmcnt), so every load issued after a returning atomic -- an agent-scope atomic takes ~4 us here -- waits for it;
  // with one claim per item the point loads of every item sat behind one (measured with -DNDT_TIMELINE: 10 k cycles per item).
  // The dynamic tail (1 / 2^sc.dyn_shift of the queue, more when that is no whole round) absorbs the imbalance.
  // FLAT dealing: one queue over all active pairs, item i to wave i of the whole grid (then i + all waves, ...), no atomics and no
  // XCD affinity.  Always in latency mode; in batch mode whenever the launch has no more items than waves -- the last rounds of a
  // batch, when a handful of pairs are still iterating: with fewer than eight active pairs most XCDs own no queue and their waves
  // would do nothing but steal, one returning atomic (~4 us) per item.  Which wave runs an item is no part of its result.
  const bool flat = FINE || (n_active * items_per_pair <= nblk * WAVES);
  const int xw = flat ? nblk * WAVES : (int)(gridDim.x >> 3) * WAVES;                    // waves per XCD (flat: of the whole grid)
  const int wx = flat ? (int)blockIdx.x * WAVES + wv : (int)(blockIdx.x >> 3) * WAVES + wv;
#ifdef NDT_TIMELINE
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl_last = __builtin_readcyclecounter();
#endif
#pragma unroll 1
  for (int probe = 0; probe < (flat ? 1 : 8); probe++) {        // own XCD first, then steal (flat: one queue, all static)
    const int xcd = flat ? 0 : (my_xcd + probe) & 7;
    const int pairs_here = flat ? n_active : (n_active > xcd ? (n_active - xcd + 7) / 8 : 0);
    const int items_here = pairs_here * items_per_pair;
    if (items_here == 0) continue;
    // static rounds of this queue (the same number for every wave; none when the grid is no multiple of 8 or for a thief)
    const int n_static = flat ? (items_here + xw - 1) / xw : (((gridDim.x & 7) == 0) ? (items_here - (items_here >> sc.dyn_shift)) / xw : 0);
    const int dyn0 = n_static * xw;               // first item of the dynamic tail
    int round = 0;
    const bool own = flat || ((probe == 0) && n_static > 0);
    int item = 0;
    if (own) item = wx;
    else {
      // a drained queue is recognised with a plain (L2) load; only a queue that still has items costs an atomic
      if (dyn0 + __hip_atomic_load(&ctl->next_item[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= items_here) continue;
      if (lane == 0) item = dyn0 + atomicAdd(&ctl->next_item[xcd], 1);
      item = __builtin_amdgcn_readfirstlane(item);
    }
#pragma unroll 1
    while (item < items_here) {
      // the next item: static while rounds are left, else claimed NOW (the atomic's round trip is hidden behind this item's work)
      const bool next_static = own && (round + 1 < n_static);
      int next_item = 0;
      if (flat) next_item = items_here;             // no dynamic tail
      else if (!next_static && lane == 0) next_item = dyn0 + atomicAdd(&ctl->next_item[xcd], 1);
      const int b = active_list[flat ? item / items_per_pair : xcd + 8 * (item / items_per_pair)];
      const int rem = item % items_per_pair;        // the pair's work item = its partial row
      TL_STAMP(0);

      sweep_item<PCA, K, IT, FINE, ORD, false>(b, rem, src, pitch, st, gd, words, recs, partials, rows_per_pair, sc, cent, grid_of, exp_tab, 0u, 0, 0
#ifdef NDT_TIMELINE
                                               , tl, tl_last
#endif
                                               );
      if (next_static) { round++; item = wx + round * xw; }
      else item = __builtin_amdgcn_readfirstlane(next_item);
    }
  }
#ifdef NDT_TIMELINE
  if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_tl[k], tl[k]);
#endif
}
