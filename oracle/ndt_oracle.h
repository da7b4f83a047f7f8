/*
 * oracle/ndt_oracle.h -- CPU restatement of lv_slam's NDT scan-matching path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (lv_slam_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED: the reference (PCL + Eigen + Sophus + FLANN) cannot be built in
 * this environment and ships no golden vectors for this path, so this
 * restatement is pinned only against an independent NumPy restatement of the same
 * cited lines (tests/golden/make_golden.py) and analytic anchors.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * lv_slam tree).  Matrices crossing this API are 4x4 float, COLUMN-MAJOR
 * (Eigen::Matrix4f native layout): M(r,c) = m[c*4+r].
 */
#ifndef NDT_ORACLE_H_
#define NDT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum order of include/ndt_omp/ndt_omp.h:51-56 */
enum { ORA_KDTREE = 0, ORA_DIRECT26 = 1, ORA_DIRECT7 = 2, ORA_DIRECT1 = 3 };
enum { ORA_VARIANT_OMP = 0, ORA_VARIANT_PCA = 1 };

typedef struct {
  float  resolution;              /* ndt_omp_impl2.hpp:56   (1.0f) */
  double step_size;               /* ndt_omp_impl2.hpp:57   (0.1)  */
  double outlier_ratio;           /* ndt_omp_impl2.hpp:58   (0.55) */
  double trans_epsilon;           /* ndt_omp_impl2.hpp:78   (0.1)  */
  int    max_iterations;          /* ndt_omp_impl2.hpp:79   (35)   */
  int    neighbor_mode;           /* ndt_omp_impl2.hpp:81   (DIRECT7) */
  int    variant;                 /* 0 = pclomp (ndt_omp), 1 = pclpca (ndt_pca) */
  int    min_points_per_voxel;    /* voxel_grid_covariance_omp.h:204 (6) */
  double min_covar_eigvalue_mult; /* voxel_grid_covariance_omp.h:205 (0.01) */
} ora_params;

/* One searchable-or-not leaf of the voxel grid (voxel_grid_covariance_omp.h:92-187). */
typedef struct {
  int32_t idx;        /* linear cell index (voxel_grid_covariance_omp_impl.hpp:223) */
  int32_t n;          /* nr_points; -1 = eigen/inverse failure (impl:339,363) */
  double  mean[3];
  double  cov[9];     /* row-major */
  double  icov[9];    /* row-major */
  double  evals[3];   /* ascending, after inflation */
  double  evecs[9];   /* columns = eigenvectors, row-major storage */
  int32_t label;      /* ndt_pca dimension_label_ (1/2/3), 0 if not computed */
  int32_t weight;     /* ndt_pca (int)dimension_2d_  */
  double  dim2d;      /* ndt_pca dimension_2d_ */
  float   centroid[3];/* f32 centroid (impl:242-243, 289), the point voxel_centroids_ / the kd-tree holds */
  int32_t n_pushed;   /* nr_points when applyFilter decided to push the centroid (before any -1 flag) */
} ora_leaf;

typedef struct ora_grid ora_grid;

void ora_default_params(ora_params* p);

/* Gauss constants, ndt_omp_impl2.hpp:93-100.  out = {d1,d2,d3}. */
void ora_gauss_constants(double outlier_ratio, float resolution, double out[3]);

/* Target build: voxel_grid_covariance_omp_impl.hpp:48-370 (+pca impl:364-397).
 * Points with non-finite coordinates are skipped (the !is_dense branch, impl:211-216).
 * Returns NULL on the int32-overflow guard (impl:75-84) or n==0. */
ora_grid* ora_grid_build(const float* x, const float* y, const float* z, size_t n,
                         const ora_params* prm);
void   ora_grid_free(ora_grid* g);
size_t ora_grid_num_leaves(const ora_grid* g);          /* all occupied cells */
size_t ora_grid_num_valid(const ora_grid* g);           /* leaves with n >= min_points */
const ora_leaf* ora_grid_leaves(const ora_grid* g);     /* ascending idx (std::map order) */
void   ora_grid_bounds(const ora_grid* g, int min_b[3], int max_b[3], int div_b[3]);

/* One computeDerivatives sweep (ndt_omp_impl2.hpp:196-305, 503-532, 566-619; pca :294-296).
 *   T_colmajor : 4x4 f32 used to transform the points (PCL transformPointCloud form)
 *   Rj         : 3x3 f32 row-major rotation used for the point Jacobian (impl2:508)
 * Outputs score, g[6], H[36] (row-major, NOT symmetric).  Returns number of (point,voxel) hits. */
long ora_derivatives(const ora_grid* g, const ora_params* prm,
                     const float* x, const float* y, const float* z, size_t n,
                     const float T_colmajor[16], const float Rj[9],
                     double* score, double grad[6], double hess[36]);

/* Convenience: sweep at tangent p: T = f32(exp(p)), Rj = its rotation block (impl2:900-907). */
long ora_derivatives_at(const ora_grid* g, const ora_params* prm,
                        const float* x, const float* y, const float* z, size_t n,
                        const double p[6], double* score, double grad[6], double hess[36]);

typedef struct {
  float  final_colmajor[16];   /* final_transformation_ */
  double trans_probability;    /* impl2:187 */
  double score;                /* last sweep score */
  int    iterations;           /* nr_iterations_ */
  int    converged;            /* converged_ */
  long   hits_last;            /* hits in last sweep */
  int    sweeps;               /* number of computeDerivatives calls */
  int    mt_loops;             /* total More-Thuente loop iterations (impl2:920-994); 0 unless step_size <= eps/2 */
  float  inc_colmajor[16];     /* transformation_ = float(exp(delta_p)) of the last step (impl2:163) */
  float  prev_inc_colmajor[16];/* previous_transformation_ (impl2:134) */
} ora_result;

/* computeTransformation (ndt_omp_impl2.hpp:87-188) + computeStepLengthMT (impl2:841-1003), including its More-Thuente
 * loop and computeHessian, which are live iff step_size <= trans_epsilon/2 (impl2:888).  Returns 0 (-3: no grid). */
int ora_align(const ora_grid* g, const ora_params* prm,
              const float* x, const float* y, const float* z, size_t n,
              const float guess_colmajor[16], ora_result* out);

/* computeHessian + updateHessian (impl2:622-714): f64, kd-tree neighbourhoods; T = f32 pose of the cloud, p = its tangent */
void ora_compute_hessian(const ora_grid* g, const ora_params* prm,
                         const float* x, const float* y, const float* z, size_t n,
                         const float T_colmajor[16], const double p[6], double H[36]);

/* calculateScore (impl2:1006-1040): x, y, z = the ALREADY TRANSFORMED cloud; gauss = {d1, d2, d3} as the members hold them */
double ora_calculate_score(const ora_grid* g, const double gauss[3], float resolution,
                           const float* x, const float* y, const float* z, size_t n);
/* static convertTransform (ndt_omp.h:209-228): [x, y, z, roll, pitch, yaw] -> 4x4 f32 column-major */
void ora_convert_transform(const double x[6], float out_colmajor[16]);

/* Sophus a621ff2 (non-templated) SE3 exp/log, tangent order [upsilon; omega]. */
void ora_se3_exp(const double p[6], double M_rowmajor[16]);
void ora_se3_log(const double M_rowmajor[16], double p[6]);   /* SE3(R,t).log() incl. quaternion normalise */
void ora_se3_compose_log(const double dp[6], const double p[6], double out[6]); /* (exp(dp)*exp(p)).log() impl2:166 */

/* Eigen::JacobiSVD<6x6>(H, FullU|FullV).solve(b) semantics (thresholded pseudo-inverse). */
void ora_svd_solve6(const double H[36], const double b[6], double x[6]);

/* symmetric 3x3 eigen (lower triangle read), ascending; evecs row-major with eigenvectors as columns */
void ora_eigen_sym3(const double A[9], double evals[3], double evecs[9]);

/* getFitnessScore(max_range) restatement (information_matrix_calculator.cpp:53-87); T column-major 4x4 f32 */
double ora_fitness_score(const float* tx, const float* ty, const float* tz, size_t nt,
                         const float* sx, const float* sy, const float* sz, size_t ns,
                         const float T_colmajor[16], double max_range, long* n_in);

/* call-site policy after one scan-to-keyframe align (scan_matching_odom_nodelet.cpp:229-250): tf_s2s, odom_velo, keyframe test,
 * constant-velocity guess.  pre_tf_s2k / key_pose / *keyframe_stamp are updated in place; matrices 4x4 f64 row-major, final_cm the
 * column-major f32 result of the align; thr = {keyframe_delta_trans, _angle, _time}; test = {dx, da, dt}.  Returns 1 on a keyframe switch. */
int ora_policy_step(double pre_tf_s2k[16], double key_pose[16], double* keyframe_stamp, const float final_cm[16], double stamp,
                    const double thr[3], double odom[16], double guess[16], double test[3]);

/* distance filter + VoxelGrid centroid downsample (prefiltering_nodelet.cpp:137-181); outputs sized n */
size_t ora_prefilter(const float* x, const float* y, const float* z, size_t n, int use_df, double dnear, double dfar, float leaf,
                     float* ox, float* oy, float* oz);

/* ---- "reference-shaped" arrangement of the same arithmetic (ndt_oracle_refshape.inc): CPU-baseline timing only ------------
 * std::map-like grid (one heap node per cell, O(log n) pointer walk per insert / probe, serial build), per-point heap
 * neighbour lists, exp(p) and dense 4x6 / 24x6 f32 matrices per evaluation, cloud rewrite per sweep, schedule(guided,8). */
typedef struct ora_refgrid ora_refgrid;
ora_refgrid* ora_refgrid_build(const float* x, const float* y, const float* z, size_t n, const ora_params* prm);
void   ora_refgrid_free(ora_refgrid* g);
size_t ora_refgrid_num_leaves(const ora_refgrid* g);
void   ora_refgrid_leaves(const ora_refgrid* g, ora_leaf* out);      /* ascending idx; out sized num_leaves */
int    ora_ref_align(ora_refgrid* g, const ora_params* prm, const float* x, const float* y, const float* z, size_t n,
                     const float guess_colmajor[16], ora_result* out);   /* -2: configuration only the port serves */

void ora_set_threads(int n);

/* ---- order-sensitivity study (tools/order_sensitivity.py).  The reference leaves these evaluation orders to
 * Eigen 3.3 / libm / the OpenMP schedule and none can be observed here; each flag switches ONE of them to another equally
 * legitimate choice so that the effect on align() can be measured.  flags = 0, acc_chunk = 256: the canonical choices every
 * parity test and fixture uses.  Process-wide, not thread-safe. */
enum {
  ORA_VAR_SUM3_02_1  = 1,   /* 3-term f32 sums of eval_hit as (t0 + t2) + t1: lane pairing of Eigen 3.3's SSE predux<Packet4f> */
  ORA_VAR_SUM3_0_12  = 2,   /* ... as t0 + (t1 + t2): Eigen's unrolled redux of a 3-vector */
  ORA_VAR_EXPF       = 4,   /* impl2:581 through the float overload (expf) instead of float(exp(double)) */
  ORA_VAR_NORM_TREE  = 8,   /* mean_.norm() of the ndt_pca weight as sqrt(x0^2 + (x1^2 + x2^2)) */
  ORA_VAR_SOLVE_LU   = 16,  /* Newton step by LU with partial pivoting instead of the one-sided-Jacobi SVD */
  ORA_VAR_SOLVE_SVD2 = 32,  /* ... by a two-sided Jacobi SVD arranged like Eigen's JacobiSVD */
  ORA_VAR_EIG_ORDER  = 64,  /* 3x3 symmetric eigen-solver with the other cyclic rotation order */
  ORA_VAR_EIG_QL     = 128, /* 3x3 symmetric eigen-solver the way Eigen 3.3's SelfAdjointEigenSolver::compute goes about it: scale, Householder
                               tridiagonalisation (the 3x3 special case), implicit symmetric QR steps with Wilkinson shift, ascending sort --
                               restated from the published algorithm (the library is not in the reference tree); a study variant, not a pin */
  ORA_VAR_ICOV_INF   = 256  /* voxel_grid_covariance_omp_impl.hpp:360-364 literally: a leaf dies only when icov_.maxCoeff() == +inf or
                               minCoeff() == -inf (Eigen's visitors skip a NaN unless it is the first coefficient); canonical: any non-finite entry */
};
void ora_set_variant(unsigned flags, int acc_chunk);
unsigned ora_get_variant(void);
void ora_solve6_variant(const double H[36], const double b[6], double x[6], unsigned flags);
/* (float)exp((double)a[i]) -- the exp of impl2:581 as eval_hit evaluates it */
void ora_exp_f32arg(const float* a, float* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
