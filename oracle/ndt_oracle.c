/*
 * oracle/ndt_oracle.c -- CPU restatement of lv_slam's NDT scan-matching path
 * (pclomp::NormalDistributionsTransform / pclpca::NormalDistributionsTransform,
 * the *_impl2.hpp Lie-algebra variant that src/ndt_omp/ndt_omp.cpp:1-2 compiles).
 *
 * TEST INFRASTRUCTURE ONLY -- see ndt_oracle.h.  PARITY UNPINNED (no reference build,
 * no reference golden vectors exist; pinned against tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the reference is built -msse4.2 without FMA
 * (CMakeLists.txt:6,11) and every f32 step below is a separately rounded op.
 *
 * Third-party arithmetic restated here (sources are not under /root/reference):
 *   Sophus a621ff2 (README.md:61-66): non-templated SE3/SO3 exp, log, operator*.
 *   Eigen 3.3: 3x3 inverse (cofactors), JacobiSVD::solve semantics (thresholded
 *   pseudo-inverse, threshold = 6*eps), SelfAdjointEigenSolver (lower triangle,
 *   ascending) -- the eigen/SVD *algorithms* here are cyclic Jacobi, results agree
 *   with Eigen's to rounding.
 *   PCL 1.8: transformPointCloud scalar form, getMinMax3D, VoxelGrid leaf-size
 *   members, getAllNeighborCellIndices order.
 */
#include "ndt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SMALL_EPS 1e-10 /* Sophus so3.h */

static int g_threads = 0;
void ora_set_threads(int n) { g_threads = n; }

/* ---- order-sensitivity variants (tools/order_sensitivity.py) ------------------------------------------
 * The reference leaves several evaluation orders to Eigen 3.3 / libm / the OpenMP schedule, none of which can be observed
 * here (parity unpinned).  Each flag below switches ONE such choice to another equally legitimate one; 0 = the canonical
 * choices that every parity test, fixture and the HIP path use.  Process-wide and not thread-safe: study use only. */
static unsigned g_var = 0;
static int g_acc_chunk = 256;                       /* points per f64 partial sum of ora_derivatives (impl2:293-302) */
void ora_set_variant(unsigned flags, int acc_chunk) { g_var = flags; g_acc_chunk = acc_chunk > 0 ? acc_chunk : 256; }
unsigned ora_get_variant(void) { return g_var; }

/* the exp of updateDerivatives as this oracle evaluates it (impl2:581; canonical choice): glibc's double exp of
 * the f32 argument, rounded to f32.  Exposed so a test can hold the device's table-driven exp against it argument by argument. */
void ora_exp_f32arg(const float* a, float* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = (float)exp((double)a[i]);
}

/* ------------------------------------------------------------------ params */
void ora_default_params(ora_params* p) {
  /* ndt_omp_impl2.hpp:53-83 (ctor); voxel_grid_covariance_omp.h:202-205 */
  p->resolution = 1.0f;
  p->step_size = 0.1;
  p->outlier_ratio = 0.55;
  p->trans_epsilon = 0.1;
  p->max_iterations = 35;
  p->neighbor_mode = ORA_DIRECT7;
  p->variant = ORA_VARIANT_OMP;
  p->min_points_per_voxel = 6;
  p->min_covar_eigvalue_mult = 0.01;
}

void ora_gauss_constants(double outlier_ratio, float resolution, double out[3]) {
  /* ndt_omp_impl2.hpp:93-100 */
  double c1 = 10 * (1 - outlier_ratio);
  double c2 = outlier_ratio / pow((double)resolution, 3);
  double d3 = -log(c2);
  double d1 = -log(c1 + c2) - d3;
  double d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / d1);
  out[0] = d1; out[1] = d2; out[2] = d3;
}

/* ------------------------------------------------------- small linear algebra */
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  /* Eigen lazy coefficient product, k ascending */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
}

/* Eigen 3.3 InverseImpl.h compute_inverse<Matrix3d>: cofactor expansion. */
static double cof3(const double* m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void mat3_inverse(const double m[9], double r[9]) {
  double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6]; /* cofactors_col0 . matrix.col(0) */
  double invdet = 1.0 / det;
  r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet; /* result.row(0) */
  r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

/* Study variant ORA_VAR_EIG_QL: the route of Eigen 3.3's SelfAdjointEigenSolver<Matrix3d>::compute (what voxel_grid_covariance_omp_impl.hpp:333
 * calls), restated from the published algorithm -- Eigen is not in the reference tree, so this pins nothing; it measures how far another
 * legitimate eigen-solver moves the inflated covariances (tools/order_sensitivity.py).  Lower triangle; eigenvalues ascending; columns of evecs. */
static void givens(double p, double q, double* c, double* s) {                    /* Eigen::JacobiRotation::makeGivens, real case */
  if (q == 0.0) { *c = p < 0 ? -1.0 : 1.0; *s = 0.0; }
  else if (p == 0.0) { *c = 0.0; *s = q < 0 ? 1.0 : -1.0; }
  else if (fabs(p) > fabs(q)) { double t = q / p, u = sqrt(1.0 + t * t); if (p < 0) u = -u; *c = 1.0 / u; *s = -t * *c; }
  else { double t = p / q, u = sqrt(1.0 + t * t); if (q < 0) u = -u; *s = -1.0 / u; *c = -t * *s; }
}
static void eigen_sym3_ql(const double Ain[9], double evals[3], double evecs[9]) {
  double m[3][3] = {{Ain[0], 0, 0}, {Ain[3], Ain[4], 0}, {Ain[6], Ain[7], Ain[8]}};   /* lower triangle */
  double scale = 0.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) if (fabs(m[i][j]) > scale) scale = fabs(m[i][j]);
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) m[i][j] /= scale;
  double diag[3], sub[2], q[3][3];
  /* tridiagonalization_inplace, 3x3 real special case */
  diag[0] = m[0][0];
  const double v1norm2 = m[2][0] * m[2][0];
  if (v1norm2 <= DBL_MIN) {
    diag[1] = m[1][1]; diag[2] = m[2][2]; sub[0] = m[1][0]; sub[1] = m[2][1];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) q[i][j] = i == j ? 1.0 : 0.0;
  } else {
    const double beta = sqrt(m[1][0] * m[1][0] + v1norm2), inv_beta = 1.0 / beta;
    const double m01 = m[1][0] * inv_beta, m02 = m[2][0] * inv_beta;
    const double qq = 2.0 * m01 * m[2][1] + m02 * (m[2][2] - m[1][1]);
    diag[1] = m[1][1] + m02 * qq; diag[2] = m[2][2] - m02 * qq;
    sub[0] = beta; sub[1] = m[2][1] - m01 * qq;
    q[0][0] = 1; q[0][1] = 0; q[0][2] = 0; q[1][0] = 0; q[1][1] = m01; q[1][2] = m02; q[2][0] = 0; q[2][1] = m02; q[2][2] = -m01;
  }
  /* computeFromTridiagonal_impl: implicit symmetric QR steps with Wilkinson shift */
  int end = 2, start = 0, iter = 0;
  const double prec = 2.0 * DBL_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; i++)
      if (fabs(sub[i]) <= (fabs(diag[i]) + fabs(diag[i + 1])) * prec || fabs(sub[i]) <= DBL_MIN) sub[i] = 0.0;
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    if (++iter > 30 * 3) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    /* tridiagonal_qr_step */
    const double td = (diag[end - 1] - diag[end]) * 0.5, e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) mu -= fabs(e);
    else {
      const double e2 = e * e, h = hypot(td, e);
      if (e2 == 0.0) mu -= (e / (td + (td > 0 ? 1.0 : -1.0))) * (e / h);
      else mu -= e2 / (td + (td > 0 ? h : -h));
    }
    double x = diag[start] - mu, z = sub[start];
    for (int k = start; k < end; k++) {
      double c, s;
      givens(x, z, &c, &s);
      const double sdk = s * diag[k] + c * sub[k], dkp1 = s * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      sub[k] = c * sdk - s * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
      x = sub[k];
      if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
      for (int i = 0; i < 3; i++) {                 /* q.applyOnTheRight(k, k+1, rot) */
        const double qi = q[i][k], qj = q[i][k + 1];
        q[i][k] = c * qi - s * qj;
        q[i][k + 1] = s * qi + c * qj;
      }
    }
  }
  /* ascending selection sort of the eigenvalues, columns follow */
  for (int i = 0; i < 2; i++) {
    int k = i;
    for (int j = i + 1; j < 3; j++) if (diag[j] < diag[k]) k = j;
    if (k != i) {
      double t = diag[i]; diag[i] = diag[k]; diag[k] = t;
      for (int r = 0; r < 3; r++) { t = q[r][i]; q[r][i] = q[r][k]; q[r][k] = t; }
    }
  }
  for (int j = 0; j < 3; j++) { evals[j] = diag[j] * scale; for (int i = 0; i < 3; i++) evecs[i * 3 + j] = q[i][j]; }
}

/* Symmetric 3x3 eigen-decomposition, cyclic Jacobi; reads the LOWER triangle only
 * (as Eigen::SelfAdjointEigenSolver::compute does, voxel_grid_covariance_omp_impl.hpp:333).
 * Eigenvalues ascending; eigenvectors are the columns of evecs (row-major storage).
 * Uses only + - * / sqrt so a device restatement can be bit-identical. */
void ora_eigen_sym3(const double Ain[9], double evals[3], double evecs[9]) {
  if (g_var & ORA_VAR_EIG_QL) { eigen_sym3_ql(Ain, evals, evecs); return; }
  double a[3][3], v[3][3];
  a[0][0] = Ain[0]; a[1][1] = Ain[4]; a[2][2] = Ain[8];
  a[0][1] = a[1][0] = Ain[3]; a[0][2] = a[2][0] = Ain[6]; a[1][2] = a[2][1] = Ain[7];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i][j] = (i == j) ? 1.0 : 0.0;
  static const int P0[3] = {0, 0, 1}, Q0[3] = {1, 2, 2}, P1[3] = {1, 0, 0}, Q1[3] = {2, 2, 1};
  const int* P = (g_var & ORA_VAR_EIG_ORDER) ? P1 : P0;     /* study variant: the other cyclic order of the three rotations */
  const int* Q = (g_var & ORA_VAR_EIG_ORDER) ? Q1 : Q0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off == 0.0) break;
    for (int k = 0; k < 3; k++) {
      int p = P[k], q = Q[k];
      double apq = a[p][q];
      if (apq == 0.0) continue;
      double g = 100.0 * fabs(apq);
      /* after 4 sweeps drop entries that no longer change the diagonal */
      if (sweep > 3 && fabs(a[p][p]) + g == fabs(a[p][p]) && fabs(a[q][q]) + g == fabs(a[q][q])) {
        a[p][q] = a[q][p] = 0.0;
        continue;
      }
      double h = a[q][q] - a[p][p];
      double t;
      if (fabs(h) + g == fabs(h)) {
        t = apq / h;
      } else {
        double theta = 0.5 * h / apq;
        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
        if (theta < 0.0) t = -t;
      }
      double c = 1.0 / sqrt(1.0 + t * t);
      double s = t * c;
      double tau = s / (1.0 + c);
      double hh = t * apq;
      a[p][p] -= hh;
      a[q][q] += hh;
      a[p][q] = a[q][p] = 0.0;
      int r = 3 - p - q; /* the remaining index */
      double arp = a[r][p], arq = a[r][q];
      a[r][p] = a[p][r] = arp - s * (arq + arp * tau);
      a[r][q] = a[q][r] = arq + s * (arp - arq * tau);
      for (int i = 0; i < 3; i++) {
        double vip = v[i][p], viq = v[i][q];
        v[i][p] = vip - s * (viq + vip * tau);
        v[i][q] = viq + s * (vip - viq * tau);
      }
    }
  }
  int o[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  /* stable 3-element sort, ascending */
  if (d[o[1]] < d[o[0]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  if (d[o[2]] < d[o[1]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
  if (d[o[1]] < d[o[0]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  for (int j = 0; j < 3; j++) {
    evals[j] = d[o[j]];
    for (int i = 0; i < 3; i++) evecs[i * 3 + j] = v[i][o[j]];
  }
}

/* The "leaf size is too small" guard of pcl::VoxelGrid / VoxelGridCovariance (voxel_grid_covariance_omp_impl.hpp:75-84): dx*dy*dz >
 * INT32_MAX with d = int64((max - min) * inv_leaf) + 1.  The reference multiplies the three int64 factors; beyond 2^63 cells (a 1e-4 m
 * leaf over a 240 m cloud, or a stray point at 1e30) that product -- and, for extents beyond 2^63, the float -> int64 cast -- is
 * undefined behaviour.  What the guard means is not in doubt; it is evaluated here without overflowing (HIP path: the same). */
static int grid_too_big(float e0, float e1, float e2) {
  const float e[3] = {e0, e1, e2};
  double prod = 1.0;
  for (int a = 0; a < 3; a++) {
    if (!(e[a] < 2147483648.0f)) return 1;             /* (also NaN / inf extents) */
    prod *= (double)((int64_t)e[a] + 1);
  }
  return prod > 2147483647.0;
}

/* Eigen::JacobiSVD<Matrix6d>(H, FullU|FullV).solve(b): x = V S^+ U^T b with
 * rank = #{ sigma_i >= max(sigma_max * 6*eps, DBL_MIN) } (ndt_omp_impl2.hpp:138-140).
 * Algorithm here: one-sided (Hestenes) Jacobi. */
void ora_svd_solve6(const double H[36], const double b[6], double x[6]) {
  /* non-finite input: Eigen 3.3 JacobiSVD keeps rank 6 (NaN singular values fail `sigma < threshold` in SVDBase::rank())
   * and solve() propagates NaN into every component, which impl2:147-151 reports as converged_ = false */
  {
    int fin = 1;
    for (int i = 0; i < 36; i++) fin = fin && isfinite(H[i]);
    for (int i = 0; i < 6; i++) fin = fin && isfinite(b[i]);
    if (!fin) { for (int i = 0; i < 6; i++) x[i] = NAN; return; }
  }
  /* Eigen 3.3 JacobiSVD::compute works on matrix / scale, scale = matrix.cwiseAbs().maxCoeff() (1 for a zero matrix), and
   * multiplies the singular values back at the end: without it the squared column norms below overflow (H entries beyond
   * ~1e77: ndt_pca's weights compound multiplicatively over DIRECT26 neighbours) or vanish, and the rotations are skipped. */
  double scale = 0;
  for (int i = 0; i < 36; i++) if (fabs(H[i]) > scale) scale = fabs(H[i]);
  if (scale == 0.0) scale = 1.0;
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { A[i][j] = H[i * 6 + j] / scale; V[i][j] = (i == j); }
  for (int sweep = 0; sweep < 60; sweep++) {
    int rotated = 0;
    for (int p = 0; p < 5; p++) {
      for (int q = p + 1; q < 6; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 6; i++) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
        rotated = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        if (zeta < 0) t = -t;
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 6; i++) {
          double aip = A[i][p], aiq = A[i][q];
          A[i][p] = c * aip - s * aiq;
          A[i][q] = s * aip + c * aiq;
          double vip = V[i][p], viq = V[i][q];
          V[i][p] = c * vip - s * viq;
          V[i][q] = s * vip + c * viq;
        }
      }
    }
    if (!rotated) break;
  }
  double sig[6], sval[6], smax = 0;                  /* sig: of the scaled work matrix; sval = sig * scale: the singular values */
  for (int j = 0; j < 6; j++) {
    double s2 = 0;
    for (int i = 0; i < 6; i++) s2 += A[i][j] * A[i][j];
    sig[j] = sqrt(s2);
    sval[j] = sig[j] * scale;
    if (sval[j] > smax) smax = sval[j];
  }
  double thr = smax * (6.0 * DBL_EPSILON);
  if (thr < DBL_MIN) thr = DBL_MIN;
  for (int i = 0; i < 6; i++) x[i] = 0;
  for (int j = 0; j < 6; j++) {
    if (!(sval[j] >= thr) || sig[j] == 0.0) continue;
    double ub = 0;
    for (int i = 0; i < 6; i++) ub += (A[i][j] / sig[j]) * b[i];       /* column j of U, times b */
    double w = ub / sval[j];
    for (int i = 0; i < 6; i++) x[i] += V[i][j] * w;
  }
}

/* ---- study variants of the Newton solve (impl2:138-140), see ora_set_variant ----------------------------------------- */
/* LU with partial pivoting (what the HIP path uses when H is well conditioned); 0 = singular to working precision */
static int lu_solve6_var(const double H[36], const double b[6], double x[6]) {
  double A[6][7];
  for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) A[i][j] = H[i * 6 + j]; A[i][6] = b[i]; }
  for (int k = 0; k < 6; k++) {
    int piv = k;
    for (int i = k + 1; i < 6; i++) if (fabs(A[i][k]) > fabs(A[piv][k])) piv = i;
    if (!(fabs(A[piv][k]) > 0)) return 0;
    if (piv != k) for (int j = 0; j < 7; j++) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
    for (int i = k + 1; i < 6; i++) {
      const double f = A[i][k] / A[k][k];
      for (int j = k; j < 7; j++) A[i][j] -= f * A[k][j];
    }
  }
  for (int i = 5; i >= 0; i--) {
    double v = A[i][6];
    for (int j = i + 1; j < 6; j++) v -= A[i][j] * x[j];
    x[i] = v / A[i][i];
  }
  return 1;
}
/* Two-sided Jacobi SVD arranged as Eigen 3.3's JacobiSVD<Matrix6d> is (scaling by the largest entry, sweeps over p > q with the
 * 2x2 real SVD of real_2x2_jacobi_svd / makeJacobi, singular values sorted descending), then the same thresholded solve. */
static void rot_rows(double W[6][6], int p, int q, double c, double s) {      /* applyOnTheLeft(p, q, {c, s}) */
  for (int j = 0; j < 6; j++) { double x = W[p][j], y = W[q][j]; W[p][j] = c * x + s * y; W[q][j] = -s * x + c * y; }
}
static void rot_cols(double W[6][6], int p, int q, double c, double s) {      /* applyOnTheRight(p, q, {c, s}) */
  for (int i = 0; i < 6; i++) { double x = W[i][p], y = W[i][q]; W[i][p] = c * x - s * y; W[i][q] = s * x + c * y; }
}
static void svd2_solve6_var(const double H[36], const double b[6], double x[6]) {
  double W[6][6], U[6][6], V[6][6], scale = 0;
  for (int i = 0; i < 36; i++) { if (!isfinite(H[i])) { for (int k = 0; k < 6; k++) x[k] = NAN; return; } if (fabs(H[i]) > scale) scale = fabs(H[i]); }
  for (int i = 0; i < 6; i++) if (!isfinite(b[i])) { for (int k = 0; k < 6; k++) x[k] = NAN; return; }
  if (scale == 0) scale = 1;
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { W[i][j] = H[i * 6 + j] / scale; U[i][j] = V[i][j] = (i == j); }
  const double precision = 2 * DBL_EPSILON, tiny = 2 * DBL_MIN;
  double maxdiag = 0;
  for (int i = 0; i < 6; i++) if (fabs(W[i][i]) > maxdiag) maxdiag = fabs(W[i][i]);
  for (int sweep = 0, done = 0; !done && sweep < 100; sweep++) {
    done = 1;
    for (int p = 1; p < 6; p++) for (int q = 0; q < p; q++) {
      double thr = precision * maxdiag; if (thr < tiny) thr = tiny;
      if (!(fabs(W[p][q]) > thr || fabs(W[q][p]) > thr)) continue;
      done = 0;
      /* real_2x2_jacobi_svd on [[W(p,p), W(p,q)], [W(q,p), W(q,q)]] */
      double m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
      double t = m00 + m11, d = m10 - m01, c1 = 1, s1 = 0;
      if (!(fabs(d) < DBL_MIN)) { double u = t / d, tmp = sqrt(1 + u * u); s1 = 1 / tmp; c1 = u / tmp; }
      /* m.applyOnTheLeft(0, 1, rot1) */
      double n00 = c1 * m00 + s1 * m10, n01 = c1 * m01 + s1 * m11, n11 = -s1 * m01 + c1 * m11;
      /* j_right.makeJacobi(n00, n01, n11) */
      double cr = 1, sr = 0, deno = 2 * fabs(n01);
      if (!(deno < DBL_MIN)) {
        double tau = (n00 - n11) / deno, w = sqrt(tau * tau + 1);
        double tt = tau > 0 ? 1 / (tau + w) : 1 / (tau - w);
        double sign_t = tt > 0 ? 1 : -1, nn = 1 / sqrt(tt * tt + 1);
        sr = -sign_t * (n01 / fabs(n01)) * fabs(tt) * nn; cr = nn;
      }
      /* j_left = rot1 * j_right.transpose() */
      double cl = c1 * cr + s1 * sr, sl = -c1 * sr + s1 * cr;
      rot_rows(W, p, q, cl, sl);
      rot_cols(U, p, q, cl, -sl);                 /* U.applyOnTheRight(p, q, j_left.transpose()) */
      rot_cols(W, p, q, cr, sr);
      rot_cols(V, p, q, cr, sr);
      if (fabs(W[p][p]) > maxdiag) maxdiag = fabs(W[p][p]);
      if (fabs(W[q][q]) > maxdiag) maxdiag = fabs(W[q][q]);
    }
  }
  double sig[6]; int ord[6];
  for (int i = 0; i < 6; i++) { sig[i] = fabs(W[i][i]); if (W[i][i] < 0) for (int r = 0; r < 6; r++) U[r][i] = -U[r][i]; sig[i] *= scale; ord[i] = i; }
  for (int i = 0; i < 6; i++) { int m = i; for (int j = i + 1; j < 6; j++) if (sig[ord[j]] > sig[ord[m]]) m = j; int t = ord[i]; ord[i] = ord[m]; ord[m] = t; }
  double thr = sig[ord[0]] * (6.0 * DBL_EPSILON);
  if (thr < DBL_MIN) thr = DBL_MIN;
  for (int i = 0; i < 6; i++) x[i] = 0;
  for (int k = 0; k < 6; k++) {
    const int j = ord[k];
    if (sig[j] < thr || sig[j] == 0.0) break;
    double ub = 0;
    for (int i = 0; i < 6; i++) ub += U[i][j] * b[i];
    const double w = ub / sig[j];
    for (int i = 0; i < 6; i++) x[i] += V[i][j] * w;
  }
}
static void newton_solve6(const double H[36], const double b[6], double x[6]) {
  if ((g_var & ORA_VAR_SOLVE_LU) && lu_solve6_var(H, b, x)) return;
  if (g_var & ORA_VAR_SOLVE_SVD2) { svd2_solve6_var(H, b, x); return; }
  ora_svd_solve6(H, b, x);
}
/* exposed for the study's self-check: solve H x = b with the variant solvers */
void ora_solve6_variant(const double H[36], const double b[6], double x[6], unsigned flags) {
  const unsigned keep = g_var; g_var = flags; newton_solve6(H, b, x); g_var = keep;
}

/* ------------------------------------------------------------ Sophus a621ff2 */
typedef struct { double w, x, y, z; } quat;

static quat quat_normalized(quat q) {
  double n = sqrt(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
  quat r = {q.w / n, q.x / n, q.y / n, q.z / n};
  return r;
}
/* Eigen Quaternion::operator= (Matrix3) -- quaternionbase_assign_impl<3,3> */
static quat quat_from_matrix(const double m[9]) {
  quat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[2 * 3 + 1] - m[1 * 3 + 2]) * t;
    q.y = (m[0 * 3 + 2] - m[2 * 3 + 0]) * t;
    q.z = (m[1 * 3 + 0] - m[0 * 3 + 1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double qv[3];
    t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    qv[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
    qv[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    qv[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    q.x = qv[0]; q.y = qv[1]; q.z = qv[2];
  }
  return q;
}
/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_matrix(quat q, double r[9]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r[0] = 1 - (tyy + tzz); r[1] = txy - twz; r[2] = txz + twy;
  r[3] = txy + twz; r[4] = 1 - (txx + tzz); r[5] = tyz - twx;
  r[6] = txz - twy; r[7] = tyz + twx; r[8] = 1 - (txx + tyy);
}
static quat quat_mul(quat a, quat b) {
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
/* QuaternionBase::_transformVector: v + w*uv + q.vec x uv, uv = 2 (q.vec x v) */
static void quat_rotate(quat q, const double v[3], double out[3]) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  for (int i = 0; i < 3; i++) out[i] = (v[i] + q.w * uv[i]) + c[i];
}

typedef struct { quat q; double t[3]; } se3;

/* SO3::expAndTheta */
static quat so3_exp(const double om[3], double* theta) {
  *theta = sqrt((om[0] * om[0] + om[1] * om[1]) + om[2] * om[2]);
  double half = 0.5 * (*theta);
  double imag, real = cos(half);
  if (*theta < SMALL_EPS) {
    double th2 = (*theta) * (*theta), th4 = th2 * th2;
    imag = 0.5 - 0.0208333 * th2 + 0.000260417 * th4;
  } else {
    imag = sin(half) / (*theta);
  }
  quat q = {real, imag * om[0], imag * om[1], imag * om[2]};
  return quat_normalized(q); /* SO3(Quaterniond) normalises */
}
/* SO3::logAndTheta (atan-based) */
static void so3_log(quat q, double om[3], double* theta) {
  double n = sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
  double w = q.w, f;
  if (n < SMALL_EPS) {
    f = 2. / w - 2. * (n * n) / (w * (w * w));
  } else {
    /* the |w|<eps special-case in the source is overwritten by the next line (kept) */
    f = 2 * atan(n / w) / n;
  }
  *theta = f * n;
  om[0] = f * q.x; om[1] = f * q.y; om[2] = f * q.z;
}
static void hat(const double o[3], double O[9]) {
  O[0] = 0; O[1] = -o[2]; O[2] = o[1];
  O[3] = o[2]; O[4] = 0; O[5] = -o[0];
  O[6] = -o[1]; O[7] = o[0]; O[8] = 0;
}
/* SE3::exp */
static se3 se3_exp(const double p[6]) {
  se3 r;
  double theta;
  const double* ups = p; const double* om = p + 3;
  r.q = so3_exp(om, &theta);
  double Om[9], Om2[9], V[9];
  hat(om, Om);
  mat3_mul(Om, Om, Om2);
  if (theta < SMALL_EPS) {
    quat_to_matrix(r.q, V);
  } else {
    double th2 = theta * theta;
    double a = (1 - cos(theta)) / th2, b = (theta - sin(theta)) / (th2 * theta);
    for (int i = 0; i < 9; i++) V[i] = (((i % 4) == 0 ? 1.0 : 0.0) + a * Om[i]) + b * Om2[i];
  }
  for (int i = 0; i < 3; i++) r.t[i] = (V[i * 3 + 0] * ups[0] + V[i * 3 + 1] * ups[1]) + V[i * 3 + 2] * ups[2];
  return r;
}
/* SE3::log */
static void se3_log(se3 s, double p[6]) {
  double theta, om[3];
  so3_log(s.q, om, &theta);
  double Om[9], Om2[9], Vi[9];
  hat(om, Om);
  mat3_mul(Om, Om, Om2);
  if (theta < SMALL_EPS) {
    for (int i = 0; i < 9; i++) Vi[i] = (((i % 4) == 0 ? 1.0 : 0.0) - 0.5 * Om[i]) + (1. / 12.) * Om2[i];
  } else {
    double c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; i++) Vi[i] = (((i % 4) == 0 ? 1.0 : 0.0) - 0.5 * Om[i]) + c * Om2[i];
  }
  for (int i = 0; i < 3; i++) p[i] = (Vi[i * 3 + 0] * s.t[0] + Vi[i * 3 + 1] * s.t[1]) + Vi[i * 3 + 2] * s.t[2];
  p[3] = om[0]; p[4] = om[1]; p[5] = om[2];
}
/* SE3::operator* */
static se3 se3_mul(se3 a, se3 b) {
  se3 r;
  double rt[3];
  quat_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.q = quat_normalized(quat_mul(a.q, b.q));
  return r;
}
static void se3_matrix(se3 s, double M[16]) { /* row-major 4x4 */
  double R[9];
  quat_to_matrix(s.q, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) M[i * 4 + j] = R[i * 3 + j];
    M[i * 4 + 3] = s.t[i];
  }
  M[12] = M[13] = M[14] = 0; M[15] = 1;
}
static se3 se3_from_Rt(const double R[9], const double t[3]) { /* SE3(R,t): SO3(R) normalises */
  se3 s;
  s.q = quat_normalized(quat_from_matrix(R));
  s.t[0] = t[0]; s.t[1] = t[1]; s.t[2] = t[2];
  return s;
}
void ora_se3_exp(const double p[6], double M[16]) { se3_matrix(se3_exp(p), M); }
void ora_se3_log(const double M[16], double p[6]) {
  double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
  double t[3] = {M[3], M[7], M[11]};
  se3_log(se3_from_Rt(R, t), p);
}
void ora_se3_compose_log(const double dp[6], const double p[6], double out[6]) {
  se3_log(se3_mul(se3_exp(dp), se3_exp(p)), out);
}

/* ------------------------------------------------------------------ voxel grid */
struct ora_grid {
  int min_b[3], max_b[3], div_b[3], mul[3];
  float leaf[3], inv_leaf[3];
  size_t n_leaves, n_valid;
  ora_leaf* leaves;        /* ascending idx */
  int64_t ncells;
  int32_t* dense;          /* cell -> leaf index or -1 (when ncells small) */
  int min_points;
};

typedef struct { int32_t idx; uint32_t pos; } keypos;
static int cmp_keypos(const void* a, const void* b) {
  const keypos* A = (const keypos*)a; const keypos* B = (const keypos*)b;
  if (A->idx != B->idx) return A->idx < B->idx ? -1 : 1;
  return A->pos < B->pos ? -1 : (A->pos > B->pos);
}

static int finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

/* Second pass of applyFilter for one leaf (impl:282-367; pca impl:364-397) from its first-pass sums: S = sum of points,
 * C = Identity + sum of p p^T (both f64, input order), cen = f32 sum, cnt = nr_points. */
static void finish_leaf(ora_leaf* L, const double S[3], const double C[9], const float cen[3], int cnt, const ora_params* prm) {
  L->n = cnt;
  L->n_pushed = cnt;                                               /* what voxel_centroids_ saw (impl:297-302) */
  for (int a = 0; a < 3; a++) L->centroid[a] = cen[a] / (float)cnt;   /* impl:289 */
  double mu[3];
  for (int a = 0; a < 3; a++) mu[a] = S[a] / (double)cnt;            /* impl:293 */
  memcpy(L->mean, mu, sizeof mu);
  if (cnt >= prm->min_points_per_voxel) {                            /* impl:297 */
    double cov[9];
    /* impl:329-330 */
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        cov[a * 3 + b] = (C[a * 3 + b] - 2 * (S[a] * mu[b])) / (double)cnt + mu[a] * mu[b];
    double f = ((double)cnt - 1.0) / (double)cnt;
    for (int a = 0; a < 9; a++) cov[a] *= f;
    double ev[3], V[9];
    ora_eigen_sym3(cov, ev, V);                                      /* impl:333-335 */
    memcpy(L->evecs, V, sizeof V);
    if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) {                      /* impl:337-341 */
      L->n = -1;
      memcpy(L->cov, cov, sizeof cov);
      memcpy(L->evals, ev, sizeof ev);
      return;
    }
    double minev = prm->min_covar_eigvalue_mult * ev[2];             /* impl:345 */
    if (ev[0] < minev) {
      ev[0] = minev;
      if (ev[1] < minev) ev[1] = minev;
      /* cov = evecs * diag * evecs.inverse()  impl:355 */
      double VD[9], Vi[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) VD[a * 3 + b] = V[a * 3 + b] * ev[b];
      mat3_inverse(V, Vi);
      mat3_mul(VD, Vi, cov);
    }
    memcpy(L->evals, ev, sizeof ev);
    memcpy(L->cov, cov, sizeof cov);
    if (prm->variant == ORA_VARIANT_PCA) {                           /* pca impl:364-397 */
      double sg[3] = {sqrt(ev[0]), sqrt(ev[1]), sqrt(ev[2])};
      double ft[3] = {(sg[2] - sg[1]) / sg[2], (sg[1] - sg[0]) / sg[2], sg[0] / sg[2]};
      int dmax = 0;
      if (ft[1] > ft[dmax]) dmax = 1;
      if (ft[2] > ft[dmax]) dmax = 2;
      L->label = dmax + 1;
      double scale = 1;
      if (L->label == 2) scale = 1.25; else if (L->label == 3) scale = 1; else if (L->label == 1) scale = 0.75;
      /* mean_.norm(): canonical = left to right; study variant = Eigen's unrolled 3-term redux x0 + (x1 + x2) */
      L->dim2d = (g_var & ORA_VAR_NORM_TREE) ? scale * sqrt(mu[0] * mu[0] + (mu[1] * mu[1] + mu[2] * mu[2]))
                                             : scale * sqrt((mu[0] * mu[0] + mu[1] * mu[1]) + mu[2] * mu[2]);
      L->weight = (int)L->dim2d;                                     /* pca.h:222-226 returns int */
    }
    mat3_inverse(cov, L->icov);                                      /* impl:359 */
    int bad = 0;
    for (int a = 0; a < 9; a++) if (!isfinite(L->icov[a])) bad = 1;  /* impl:360-364 (NaN treated as inf) */
    if (g_var & ORA_VAR_ICOV_INF) {                                  /* study variant: the reference's test as written -- Eigen's coefficient visitors */
      static const int cm[9] = {0, 3, 6, 1, 4, 7, 2, 5, 8};          /* column-major visiting order of a row-major 3x3 */
      double mx = L->icov[0], mn = L->icov[0];
      for (int a = 1; a < 9; a++) { const double v = L->icov[cm[a]]; if (v > mx) mx = v; if (v < mn) mn = v; }
      bad = (mx == (double)INFINITY) || (mn == -(double)INFINITY);
    }
    if (bad) L->n = -1;
  }
}

ora_grid* ora_grid_build(const float* x, const float* y, const float* z, size_t n, const ora_params* prm) {
  if (n == 0) return NULL;
  ora_grid* g = (ora_grid*)calloc(1, sizeof(ora_grid));
  g->min_points = prm->min_points_per_voxel;
  /* pcl::VoxelGrid::setLeafSize: leaf_size_, inverse_leaf_size_ = 1/leaf (f32) */
  for (int a = 0; a < 3; a++) { g->leaf[a] = prm->resolution; g->inv_leaf[a] = 1.0f / prm->resolution; }
  /* getMinMax3D (impl:72) over finite points */
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t nfinite = 0;
  for (size_t i = 0; i < n; i++) {
    if (!finite3(x[i], y[i], z[i])) continue;
    nfinite++;
    if (x[i] < mn[0]) mn[0] = x[i]; if (x[i] > mx[0]) mx[0] = x[i];
    if (y[i] < mn[1]) mn[1] = y[i]; if (y[i] > mx[1]) mx[1] = y[i];
    if (z[i] < mn[2]) mn[2] = z[i]; if (z[i] > mx[2]) mx[2] = z[i];
  }
  if (nfinite == 0) { free(g); return NULL; }
  /* overflow guard impl:75-84 (f32 arithmetic, then int64) */
  if (grid_too_big((mx[0] - mn[0]) * g->inv_leaf[0], (mx[1] - mn[1]) * g->inv_leaf[1], (mx[2] - mn[2]) * g->inv_leaf[2])) { free(g); return NULL; }
  /* impl:87-103 */
  for (int a = 0; a < 3; a++) {
    g->min_b[a] = (int)floorf(mn[a] * g->inv_leaf[a]);
    g->max_b[a] = (int)floorf(mx[a] * g->inv_leaf[a]);
    g->div_b[a] = g->max_b[a] - g->min_b[a] + 1;
  }
  g->mul[0] = 1; g->mul[1] = g->div_b[0]; g->mul[2] = g->div_b[0] * g->div_b[1];
  g->ncells = (int64_t)g->div_b[0] * g->div_b[1] * g->div_b[2];

  /* first pass impl:209-263: cell index per point; group with a stable sort so the
   * f64 sums run in input order exactly as leaves_[idx] accumulation does. */
  keypos* kp = (keypos*)malloc(nfinite * sizeof(keypos));
  size_t m = 0;
  for (size_t i = 0; i < n; i++) {
    if (!finite3(x[i], y[i], z[i])) continue;
    int ijk0 = (int)(floorf(x[i] * g->inv_leaf[0]) - (float)g->min_b[0]);
    int ijk1 = (int)(floorf(y[i] * g->inv_leaf[1]) - (float)g->min_b[1]);
    int ijk2 = (int)(floorf(z[i] * g->inv_leaf[2]) - (float)g->min_b[2]);
    kp[m].idx = ijk0 * g->mul[0] + ijk1 * g->mul[1] + ijk2 * g->mul[2];
    kp[m].pos = (uint32_t)i;
    m++;
  }
  qsort(kp, m, sizeof(keypos), cmp_keypos);
  size_t nl = 0;
  for (size_t i = 0; i < m; i++) if (i == 0 || kp[i].idx != kp[i - 1].idx) nl++;
  g->n_leaves = nl;
  g->leaves = (ora_leaf*)calloc(nl, sizeof(ora_leaf));

  size_t li = 0;
  for (size_t s = 0; s < m;) {
    size_t e = s;
    while (e < m && kp[e].idx == kp[s].idx) e++;
    ora_leaf* L = &g->leaves[li++];
    L->idx = kp[s].idx;
    /* Leaf ctor: mean_ = 0, cov_ = Identity, icov_ = 0 (voxel_grid_covariance_omp.h:97-106) */
    double S[3] = {0, 0, 0};
    double C[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float cen[3] = {0.f, 0.f, 0.f};                                  /* leaf.centroid (f32), impl:229-230 */
    int cnt = 0;
    for (size_t k = s; k < e; k++) {
      size_t i = kp[k].pos;
      double p3[3] = {(double)x[i], (double)y[i], (double)z[i]};
      for (int a = 0; a < 3; a++) S[a] += p3[a];                       /* impl:235 */
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) C[a * 3 + b] += p3[a] * p3[b]; /* impl:237 */
      cen[0] += x[i]; cen[1] += y[i]; cen[2] += z[i];                  /* impl:242-243 */
      cnt++;
    }
    finish_leaf(L, S, C, cen, cnt, prm);
    s = e;
  }
  free(kp);
  g->n_valid = 0;
  for (size_t i = 0; i < nl; i++) if (g->leaves[i].n >= prm->min_points_per_voxel) g->n_valid++;
  if (g->ncells <= ((int64_t)1 << 27)) {
    g->dense = (int32_t*)malloc((size_t)g->ncells * sizeof(int32_t));
    memset(g->dense, 0xff, (size_t)g->ncells * sizeof(int32_t));
    for (size_t i = 0; i < nl; i++) g->dense[g->leaves[i].idx] = (int32_t)i;
  }
  return g;
}

void ora_grid_free(ora_grid* g) { if (!g) return; free(g->leaves); free(g->dense); free(g); }
size_t ora_grid_num_leaves(const ora_grid* g) { return g->n_leaves; }
size_t ora_grid_num_valid(const ora_grid* g) { return g->n_valid; }
const ora_leaf* ora_grid_leaves(const ora_grid* g) { return g->leaves; }
void ora_grid_bounds(const ora_grid* g, int min_b[3], int max_b[3], int div_b[3]) {
  for (int a = 0; a < 3; a++) { min_b[a] = g->min_b[a]; max_b[a] = g->max_b[a]; div_b[a] = g->div_b[a]; }
}

static const ora_leaf* grid_find(const ora_grid* g, int32_t idx) {
  if (g->dense) { int32_t l = g->dense[idx]; return l < 0 ? NULL : &g->leaves[l]; }
  size_t lo = 0, hi = g->n_leaves;
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (g->leaves[mid].idx < idx) lo = mid + 1; else hi = mid; }
  return (lo < g->n_leaves && g->leaves[lo].idx == idx) ? &g->leaves[lo] : NULL;
}

/* neighbour offset tables: DIRECT7 impl:423-430; DIRECT1 impl:441;
 * DIRECT26 = pcl::getAllNeighborCellIndices() (PCL 1.8 voxel_grid.h): 13 "half" offsets then their negation */
static int build_offsets(int mode, int off[26][3]) {
  if (mode == ORA_DIRECT1) { off[0][0] = off[0][1] = off[0][2] = 0; return 1; }
  if (mode == ORA_DIRECT7) {
    static const int o7[7][3] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1}};
    memcpy(off, o7, sizeof o7);
    return 7;
  }
  if (mode == ORA_DIRECT26) {
    int k = 0;
    for (int i = -1; i < 2; i++) for (int j = -1; j < 2; j++) { off[k][0] = i; off[k][1] = j; off[k][2] = -1; k++; }
    for (int i = -1; i < 2; i++) { off[k][0] = i; off[k][1] = -1; off[k][2] = 0; k++; }
    off[k][0] = -1; off[k][1] = 0; off[k][2] = 0; k++;
    for (int i = 0; i < 13; i++) { off[13 + i][0] = -off[i][0]; off[13 + i][1] = -off[i][1]; off[13 + i][2] = -off[i][2]; }
    return 26;
  }
  return 0; /* KDTREE: handled by radius_search() */
}

/* -------------------------------------------------- one (point, voxel) evaluation */
/* updateDerivatives (ndt_omp_impl2.hpp:566-619) with J/Hp of
 * computePointDerivatives_AngleAxisd (impl2:503-532) folded in.  All f32 ops are
 * single, left-to-right, uncontracted.  Returns score_inc (0 if the validity gate fires). */
/* The three-term f32 sums of the 4-wide inner products whose fourth term is a structural zero (x_trans4, c_inv4 and
 * point_gradient4 all carry a zero fourth row / entry).  Canonical: left to right = Eigen's unrolled scalar redux
 * (t0 + t1) + (t2 + 0).  Study variants: (t0 + t2) + (t1 + 0), the lane pairing of Eigen 3.3's SSE predux<Packet4f>;
 * t0 + (t1 + t2), Eigen's unrolled redux of a genuine 3-vector. */
static inline __attribute__((always_inline)) float sum3v(float a, float b, float c, const unsigned var) {
  if (var & ORA_VAR_SUM3_02_1) return (a + c) + b;
  if (var & ORA_VAR_SUM3_0_12) return a + (b + c);
  return (a + b) + c;
}
static inline __attribute__((always_inline)) double eval_hit_v(const float u[3], const float r[3], const float C[9], double d1, float d2f,
                                                               double g[6], double H[36], const unsigned var) {
  /* y = x_trans4 * c_inv4 (impl2:600) */
  float y[3];
  for (int j = 0; j < 3; j++) y[j] = sum3v(u[0] * C[0 * 3 + j], u[1] * C[1 * 3 + j], u[2] * C[2 * 3 + j], var);
  float qf = sum3v(u[0] * y[0], u[1] * y[1], u[2] * y[2], var);
  /* impl2:581 -- exp evaluated in double on the f32 argument, rounded to f32 (canonical choice);
   * study variant: the float overload (glibc expf) */
  float e0 = (var & ORA_VAR_EXPF) ? expf((-d2f * qf) * 0.5f) : (float)exp((double)((-d2f * qf) * 0.5f));
  float s_inc = (float)(-d1 * (double)e0);                       /* impl2:583 */
  float e1 = d2f * e0;                                           /* impl2:585 */
  if (e1 > 1 || e1 < 0 || e1 != e1) return 0;                    /* impl2:588-589 */
  float e = (float)((double)e1 * d1);                            /* impl2:592 */
  /* CJ = c_inv4 * point_gradient4 (impl2:594); J = [I | -[r]x] */
  float CJ[3][6];
  for (int a = 0; a < 3; a++) {
    CJ[a][0] = C[a * 3 + 0]; CJ[a][1] = C[a * 3 + 1]; CJ[a][2] = C[a * 3 + 2];
    CJ[a][3] = C[a * 3 + 1] * (-r[2]) + C[a * 3 + 2] * r[1];
    CJ[a][4] = C[a * 3 + 0] * r[2] + C[a * 3 + 2] * (-r[0]);
    CJ[a][5] = C[a * 3 + 0] * (-r[1]) + C[a * 3 + 1] * r[0];
  }
  float v[6];
  for (int k = 0; k < 6; k++) v[k] = sum3v(u[0] * CJ[0][k], u[1] * CJ[1][k], u[2] * CJ[2][k], var);   /* impl2:595 */
  for (int k = 0; k < 6; k++) g[k] += (double)(e * v[k]);                                     /* impl2:597 */
  /* JCJ = J^T * CJ (impl2:601) */
  float JCJ[6][6];
  for (int j = 0; j < 6; j++) {
    JCJ[0][j] = CJ[0][j]; JCJ[1][j] = CJ[1][j]; JCJ[2][j] = CJ[2][j];
    JCJ[3][j] = (-r[2]) * CJ[1][j] + r[1] * CJ[2][j];
    JCJ[4][j] = r[2] * CJ[0][j] + (-r[0]) * CJ[2][j];
    JCJ[5][j] = (-r[1]) * CJ[0][j] + r[0] * CJ[1][j];
  }
  /* z_i[j] = y * Hp_block_i (impl2:607), Hp of impl2:522-530 */
  float Z[6][6];
  memset(Z, 0, sizeof Z);
  Z[3][3] = y[1] * (-r[1]) + y[2] * (-r[2]);
  Z[4][3] = y[0] * r[1];
  Z[5][3] = y[0] * r[2];
  Z[3][4] = y[1] * r[0];
  Z[4][4] = y[0] * (-r[0]) + y[2] * (-r[2]);
  Z[5][4] = y[1] * r[2];
  Z[3][5] = y[2] * r[0];
  Z[4][5] = y[2] * r[1];
  Z[5][5] = y[0] * (-r[0]) + y[1] * (-r[1]);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++)
      H[i * 6 + j] += (double)(e * ((((-d2f) * v[i]) * v[j] + Z[i][j]) + JCJ[j][i]));          /* impl2:611-613 */
  return (double)s_inc;
}
/* the canonical instance is compiled with the variant tests folded away */
static double eval_hit(const float u[3], const float r[3], const float C[9], double d1, float d2f, double g[6], double H[36]) {
  if (__builtin_expect(g_var & (ORA_VAR_SUM3_02_1 | ORA_VAR_SUM3_0_12 | ORA_VAR_EXPF), 0)) return eval_hit_v(u, r, C, d1, d2f, g, H, g_var);
  return eval_hit_v(u, r, C, d1, d2f, g, H, 0u);
}

#define CHUNK 256

/* VoxelGridCovariance::radiusSearch (voxel_grid_covariance_omp.h:505-534): FLANN radius query over the f32 centroids of
 * the leaves that had >= min_points points when applyFilter pushed them (impl:297-302), squared radius float(r*r), strict
 * '<', results sorted by distance; NO nr_points re-check, so eigen/inverse-failed leaves are returned too.
 * Brute force here.  Returns the number of neighbours (<= cap). */
typedef struct { float d2; int li; } kd_hit;
static int cmp_kd(const void* a, const void* b) {
  const kd_hit* A = (const kd_hit*)a; const kd_hit* B = (const kd_hit*)b;
  if (A->d2 != B->d2) return A->d2 < B->d2 ? -1 : 1;
  return A->li < B->li ? -1 : (A->li > B->li);
}
static int radius_search(const ora_grid* g, const float q[3], double radius, kd_hit* out, int cap) {
  const float r2 = (float)(radius * radius);
  int k = 0;
  for (size_t i = 0; i < g->n_leaves; i++) {
    const ora_leaf* L = &g->leaves[i];
    if (L->n_pushed < g->min_points) continue;
    float dx = q[0] - L->centroid[0], dy = q[1] - L->centroid[1], dz = q[2] - L->centroid[2];
    float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < r2 && k < cap) { out[k].d2 = d2; out[k].li = (int)i; k++; }
  }
  qsort(out, k, sizeof(kd_hit), cmp_kd);
  return k;
}

/* reference-shaped mode (ndt_oracle_refshape.inc): set by ora_ref_align for the duration of one align */
typedef struct ora_refgrid ora_refgrid;
static _Thread_local ora_refgrid* g_ref = NULL;     /* per calling thread: concurrent aligns on other threads are not redirected */
static _Thread_local double g_ref_p[6];
static long ref_derivatives(ora_refgrid* R, const ora_params* prm, const float* x, const float* y, const float* z, size_t n,
                            const float T[16], const double p[6], double* score, double grad[6], double hess[36]);

long ora_derivatives(const ora_grid* g, const ora_params* prm,
                     const float* x, const float* y, const float* z, size_t n,
                     const float T[16], const float Rj[9],
                     double* score, double grad[6], double hess[36]) {
  if (g_ref) return ref_derivatives(g_ref, prm, x, y, z, n, T, g_ref_p, score, grad, hess);
  double gc[3];
  ora_gauss_constants(prm->outlier_ratio, prm->resolution, gc);
  const double d1 = gc[0];
  const float d2f = (float)gc[1];                                 /* impl2:578 */
  int off[26][3];
  const int K = build_offsets(prm->neighbor_mode, off);
  const int pca = prm->variant == ORA_VARIANT_PCA;
  /* the reference adds per-thread partial sums of a guided schedule (impl2:223, 293-302): its own f64 order changes from run
   * to run.  Canonical here: 256-point partial sums added in order; g_acc_chunk is the study's knob for another order. */
  const size_t CH = (size_t)g_acc_chunk;
  size_t nchunks = (n + CH - 1) / CH;
  double* part = (double*)calloc(nchunks ? nchunks : 1, 44 * sizeof(double));
#ifdef _OPENMP
  int nt = g_threads > 0 ? g_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
#endif
  for (long c = 0; c < (long)nchunks; c++) {
    double* acc = part + (size_t)c * 44; /* [0]=score [1..6]=g [7..42]=H [43]=hits */
    size_t i0 = (size_t)c * CH, i1 = i0 + CH < n ? i0 + CH : n;
    for (size_t i = i0; i < i1; i++) {
      float px = x[i], py = y[i], pz = z[i];
      if (!finite3(px, py, pz)) continue;
      /* PCL 1.8 transformPointCloud scalar form */
      float xt[3];
      for (int a = 0; a < 3; a++) xt[a] = ((T[0 * 4 + a] * px + T[1 * 4 + a] * py) + T[2 * 4 + a] * pz) + T[3 * 4 + a];
      if (!finite3(xt[0], xt[1], xt[2])) continue;   /* NaN pose (NaN More-Thuente trial value): the reference's float->int cast is
                                                        UB there; canonical choice = such a point has no neighbours */
      /* impl2:507-508 : x_t = float(exp(p).matrix()) * (x,0) */
      float r[3];
      for (int a = 0; a < 3; a++) r[a] = (Rj[a * 3 + 0] * px + Rj[a * 3 + 1] * py) + Rj[a * 3 + 2] * pz;
      /* getNeighborhoodAtPoint, voxel_grid_covariance_omp_impl.hpp:379-399 */
      int ijk[3] = {(int)floorf(xt[0] / g->leaf[0]), (int)floorf(xt[1] / g->leaf[1]), (int)floorf(xt[2] / g->leaf[2])};
      double s_pt = 0, g_pt[6] = {0}, H_pt[36] = {0};
      const ora_leaf* nb[64];
      int nnb = 0;
      if (prm->neighbor_mode == ORA_KDTREE) {                      /* impl2:251-253 */
        kd_hit hits[64];
        int kh = radius_search(g, xt, (double)prm->resolution, hits, 64);
        for (int k = 0; k < kh; k++) nb[nnb++] = &g->leaves[hits[k].li];
      } else {
        for (int k = 0; k < K; k++) {
          int c3[3] = {ijk[0] + off[k][0], ijk[1] + off[k][1], ijk[2] + off[k][2]};
          if (c3[0] < g->min_b[0] || c3[0] > g->max_b[0] || c3[1] < g->min_b[1] || c3[1] > g->max_b[1] ||
              c3[2] < g->min_b[2] || c3[2] > g->max_b[2]) continue;
          int32_t idx = (c3[0] - g->min_b[0]) * g->mul[0] + (c3[1] - g->min_b[1]) * g->mul[1] + (c3[2] - g->min_b[2]) * g->mul[2];
          const ora_leaf* L = grid_find(g, idx);
          if (!L || L->n < g->min_points) continue;
          nb[nnb++] = L;
        }
      }
      for (int k = 0; k < nnb; k++) {
        const ora_leaf* L = nb[k];
        /* impl2:276-281, 574-576 */
        float u[3], Cf[9];
        for (int a = 0; a < 3; a++) u[a] = (float)((double)xt[a] - L->mean[a]);
        for (int a = 0; a < 9; a++) Cf[a] = (float)L->icov[a];
        s_pt += eval_hit(u, r, Cf, d1, d2f, g_pt, H_pt);
        if (pca) {                                                  /* ndt_pca_impl2.hpp:295-296 */
          double w = (double)L->weight;
          s_pt *= w;
          for (int a = 0; a < 6; a++) g_pt[a] *= w;
          for (int a = 0; a < 36; a++) H_pt[a] *= w;
        }
        acc[43] += 1;
      }
      acc[0] += s_pt;                                               /* impl2:293-295 */
      for (int a = 0; a < 6; a++) acc[1 + a] += g_pt[a];
      for (int a = 0; a < 36; a++) acc[7 + a] += H_pt[a];
    }
  }
  double tot[44] = {0};
  for (size_t c = 0; c < nchunks; c++) for (int a = 0; a < 44; a++) tot[a] += part[c * 44 + a];  /* impl2:298-302 (fixed order) */
  free(part);
  *score = tot[0];
  memcpy(grad, tot + 1, 6 * sizeof(double));
  memcpy(hess, tot + 7, 36 * sizeof(double));
  return (long)tot[43];
}

static void pose_to_f32(const double p[6], float T[16], float Rj[9]) {
  double M[16];
  if (g_ref) memcpy(g_ref_p, p, sizeof g_ref_p);                      /* the tangent the next sweep's exp(p) is taken at */
  ora_se3_exp(p, M);
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T[c * 4 + r] = (float)M[r * 4 + c];   /* .cast<float>() */
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rj[r * 3 + c] = (float)M[r * 4 + c];
}

long ora_derivatives_at(const ora_grid* g, const ora_params* prm,
                        const float* x, const float* y, const float* z, size_t n,
                        const double p[6], double* score, double grad[6], double hess[36]) {
  float T[16], Rj[9];
  pose_to_f32(p, T, Rj);
  return ora_derivatives(g, prm, x, y, z, n, T, Rj, score, grad, hess);
}

/* computeHessian + updateHessian (ndt_omp_impl2.hpp:622-714): f64 throughout, kd-tree neighbourhoods whatever the search
 * method is, point derivatives from the f64 overload (impl2:535-563: x_t = exp(p).matrix() * (x,0) in double).
 * T = the f32 pose that produced trans_cloud, p = the tangent x_t handed to computeHessian.  Serial, point order. */
void ora_compute_hessian(const ora_grid* g, const ora_params* prm,
                         const float* x, const float* y, const float* z, size_t n,
                         const float T[16], const double p[6], double H[36]) {
  double gc[3];
  ora_gauss_constants(prm->outlier_ratio, prm->resolution, gc);
  const double d1 = gc[0], d2 = gc[1];
  double M[16];
  ora_se3_exp(p, M);
  memset(H, 0, 36 * sizeof(double));
  int finite_p = 1;
  for (int a = 0; a < 6; a++) if (!isfinite(p[a])) finite_p = 0;
  if (!finite_p) return;                                          /* NaN cloud: radiusSearch finds nothing */
  for (size_t i = 0; i < n; i++) {
    float px = x[i], py = y[i], pz = z[i];
    if (!finite3(px, py, pz)) continue;
    float xt[3];
    for (int a = 0; a < 3; a++) xt[a] = ((T[0 * 4 + a] * px + T[1 * 4 + a] * py) + T[2 * 4 + a] * pz) + T[3 * 4 + a];
    if (!finite3(xt[0], xt[1], xt[2])) continue;
    const double X[3] = {(double)px, (double)py, (double)pz};
    double r[3];
    for (int a = 0; a < 3; a++) r[a] = (M[a * 4 + 0] * X[0] + M[a * 4 + 1] * X[1]) + M[a * 4 + 2] * X[2];
    /* J (3x6): I | -[r]x pattern (impl2:544-549);  Hp(i,j) 3-vectors for i,j in 3..5 (impl2:555-563) */
    double J[3][6] = {{1, 0, 0, 0, r[2], -r[1]}, {0, 1, 0, -r[2], 0, r[0]}, {0, 0, 1, r[1], -r[0], 0}};
    double Hp[6][6][3];
    memset(Hp, 0, sizeof Hp);
    Hp[3][3][1] = -r[1]; Hp[3][3][2] = -r[2];
    Hp[4][3][0] = r[1];
    Hp[5][3][0] = r[2];
    Hp[3][4][1] = r[0];
    Hp[4][4][0] = -r[0]; Hp[4][4][2] = -r[2];
    Hp[5][4][1] = r[2];
    Hp[3][5][2] = r[0];
    Hp[4][5][2] = r[1];
    Hp[5][5][0] = -r[0]; Hp[5][5][1] = -r[1];
    kd_hit hits[64];
    int kh = radius_search(g, xt, (double)prm->resolution, hits, 64);
    for (int k = 0; k < kh; k++) {
      const ora_leaf* L = &g->leaves[hits[k].li];
      double u[3];
      for (int a = 0; a < 3; a++) u[a] = (double)xt[a] - L->mean[a];
      const double* C = L->icov;
      double Cu[3];
      for (int a = 0; a < 3; a++) Cu[a] = (C[a * 3 + 0] * u[0] + C[a * 3 + 1] * u[1]) + C[a * 3 + 2] * u[2];
      double e = d2 * exp(-d2 * ((u[0] * Cu[0] + u[1] * Cu[1]) + u[2] * Cu[2]) / 2);       /* impl2:691 */
      if (e > 1 || e < 0 || e != e) continue;                                               /* impl2:694-695 */
      e *= d1;
      double CJ[6][3], uCJ[6];
      for (int c = 0; c < 6; c++) {
        for (int a = 0; a < 3; a++) CJ[c][a] = (C[a * 3 + 0] * J[0][c] + C[a * 3 + 1] * J[1][c]) + C[a * 3 + 2] * J[2][c];
        uCJ[c] = (u[0] * CJ[c][0] + u[1] * CJ[c][1]) + u[2] * CJ[c][2];
      }
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
          double CH[3];
          for (int q = 0; q < 3; q++) CH[q] = (C[q * 3 + 0] * Hp[a][b][0] + C[q * 3 + 1] * Hp[a][b][1]) + C[q * 3 + 2] * Hp[a][b][2];
          const double t2 = (u[0] * CH[0] + u[1] * CH[1]) + u[2] * CH[2];
          const double t3 = (J[0][b] * CJ[a][0] + J[1][b] * CJ[a][1]) + J[2][b] * CJ[a][2];
          H[a * 6 + b] += e * (-d2 * uCJ[a] * uCJ[b] + t2 + t3);                            /* impl2:709-711 */
        }
    }
  }
}

/* calculateScore (ndt_omp_impl2.hpp:1006-1040; identical in ndt_pca_impl2.hpp:1013-1047): negative log-likelihood of an ALREADY
 * TRANSFORMED cloud.  f64 throughout; neighbourhoods from radiusSearch(point, resolution) (distance order, no nr_points re-check);
 * every term divided by neighborhood.size() (impl2:1037); the running sum in (point, neighbour) order; / cloud.size() (impl2:1040).
 * gauss = {d1, d2, d3}: the members gauss_d1_/d2_/d3_ at the time of the call (constructor values before the first align, impl2:70-76). */
double ora_calculate_score(const ora_grid* g, const double gauss[3], float resolution,
                           const float* x, const float* y, const float* z, size_t n) {
  double score = 0;
  for (size_t i = 0; i < n; i++) {
    const float xt[3] = {x[i], y[i], z[i]};
    if (!finite3(xt[0], xt[1], xt[2])) continue;                  /* FLANN on a non-finite query: canonical choice = no neighbours */
    kd_hit hits[64];
    const int kh = g ? radius_search(g, xt, (double)resolution, hits, 64) : 0;
    for (int k = 0; k < kh; k++) {
      const ora_leaf* L = &g->leaves[hits[k].li];
      double u[3];
      for (int a = 0; a < 3; a++) u[a] = (double)xt[a] - L->mean[a];                          /* impl2:1025-1028 */
      const double* C = L->icov;
      double Cu[3];
      for (int a = 0; a < 3; a++) Cu[a] = (C[a * 3 + 0] * u[0] + C[a * 3 + 1] * u[1]) + C[a * 3 + 2] * u[2];
      const double e = exp(-gauss[1] * ((u[0] * Cu[0] + u[1] * Cu[1]) + u[2] * Cu[2]) / 2);   /* impl2:1033 */
      const double inc = -gauss[0] * e - gauss[2];                                           /* impl2:1035 */
      score += inc / (double)kh;                                                             /* impl2:1037 */
    }
  }
  return score / (double)n;                                                                  /* impl2:1040 */
}

/* the static convertTransform helpers (ndt_omp.h:209-228): Translation3f * AngleAxisf(roll, X) * AngleAxisf(pitch, Y) * AngleAxisf(yaw, Z),
 * f32, as Eigen 3.3 evaluates it (third-party, not in the tree; restated from its published algorithm: AngleAxis::toRotationMatrix --
 * sin_axis = sin(a) * axis, cos1_axis = (1 - cos a) * axis, off-diagonals tmp -/+ sin_axis, diagonal cos1_axis * axis + c -- then
 * Transform::rotate = linear() * R, coefficient-wise 3x3 products whose 3-term sums reduce as t0 + (t1 + t2)).  out: 4x4 column-major. */
static void aa_matrix_f(float angle, int axis, float R[9]) {
  const float ax[3] = {axis == 0 ? 1.f : 0.f, axis == 1 ? 1.f : 0.f, axis == 2 ? 1.f : 0.f};
  const float sn = sinf(angle), c = cosf(angle);
  const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
  const float c1[3] = {(1.f - c) * ax[0], (1.f - c) * ax[1], (1.f - c) * ax[2]};
  float tmp = c1[0] * ax[1];
  R[0 * 3 + 1] = tmp - sa[2]; R[1 * 3 + 0] = tmp + sa[2];
  tmp = c1[0] * ax[2];
  R[0 * 3 + 2] = tmp + sa[1]; R[2 * 3 + 0] = tmp - sa[1];
  tmp = c1[1] * ax[2];
  R[1 * 3 + 2] = tmp - sa[0]; R[2 * 3 + 1] = tmp + sa[0];
  for (int a = 0; a < 3; a++) R[a * 3 + a] = c1[a] * ax[a] + c;
}
static void mul33_f(const float A[9], const float B[9], float C[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + (A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c]);
}
void ora_convert_transform(const double x[6], float out[16]) {
  float Rx[9], Ry[9], Rz[9], A[9], L[9];
  aa_matrix_f((float)x[3], 0, Rx); aa_matrix_f((float)x[4], 1, Ry); aa_matrix_f((float)x[5], 2, Rz);
  mul33_f(Rx, Ry, A);
  mul33_f(A, Rz, L);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out[c * 4 + r] = L[r * 3 + c];
    out[12 + r] = (float)x[r];
    out[r * 4 + 3] = 0.f;
  }
  out[15] = 1.f;
}

/* std::min / std::max as the reference's libstdc++ evaluates them: a NaN first argument is returned unchanged */
static double cmin(double a, double b) { return b < a ? b : a; }
static double cmax(double a, double b) { return a < b ? b : a; }

/* updateIntervalMT, impl2:717-755.  I = {a_l, f_l, g_l, a_u, f_u, g_u} */
static int update_interval_mt(double I[6], double a_t, double f_t, double g_t) {
  if (f_t > I[1]) { I[3] = a_t; I[4] = f_t; I[5] = g_t; return 0; }
  if (g_t * (I[0] - a_t) > 0) { I[0] = a_t; I[1] = f_t; I[2] = g_t; return 0; }
  if (g_t * (I[0] - a_t) < 0) { I[3] = I[0]; I[4] = I[1]; I[5] = I[2]; I[0] = a_t; I[1] = f_t; I[2] = g_t; return 0; }
  return 1;
}

/* trialValueSelectionMT, impl2:758-838 */
static double trial_value_mt(const double I[6], double a_t, double f_t, double g_t) {
  const double a_l = I[0], f_l = I[1], g_l = I[2], a_u = I[3], f_u = I[4], g_u = I[5];
  if (f_t > f_l) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(g_l)) {
    double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    double w = sqrt(z * z - g_t * g_l);
    double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? cmin(a_t + 0.66 * (a_u - a_t), a_n) : cmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  double w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

typedef struct {
  const ora_grid* g; const ora_params* prm; const float *x, *y, *z; size_t n;
  float* final_cm; long* hits; int* sweeps; int* mt_loops;
} mt_ctx;

/* computeStepLengthMT, impl2:841-1003, literal: every trial re-transforms and re-sweeps.  dir may be flipped in place. */
static double step_length_mt(const mt_ctx* c, const double xv[6], double dir[6], double step_init, double step_max, double step_min,
                             double* score, double grad[6], double H[36]) {
  double phi_0 = -*score, d_phi_0 = 0;
  for (int a = 0; a < 6; a++) d_phi_0 += grad[a] * dir[a];
  d_phi_0 = -d_phi_0;                                             /* impl2:849 */
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;                                   /* impl2:856-857: nothing re-evaluated */
    d_phi_0 *= -1;
    for (int a = 0; a < 6; a++) dir[a] *= -1;                     /* impl2:861-862 */
  }
  const double mu = 1.e-4, nu = 0.9;
  /* auxilaryFunction_PsiMT(a, f_a, f_0, g_0, mu) = f_a - f_0 - mu*g_0*a ; dPsiMT(g_a, g_0, mu) = g_a - mu*g_0 (ndt_omp.h:480-496) */
  double I[6] = {0, phi_0 - phi_0 - mu * d_phi_0 * 0.0, d_phi_0 - mu * d_phi_0, 0, phi_0 - phi_0 - mu * d_phi_0 * 0.0, d_phi_0 - mu * d_phi_0};
  int interval_converged = (step_max - step_min) > 0, open_interval = 1;   /* impl2:888 */
  double a_t = cmax(cmin(step_init, step_max), step_min);         /* impl2:890-892 */
  double xt[6];
  float T[16], Rj[9];
  for (int a = 0; a < 6; a++) xt[a] = xv[a] + dir[a] * a_t;       /* impl2:894 */
  pose_to_f32(xt, T, Rj);                                         /* impl2:900 */
  memcpy(c->final_cm, T, sizeof T);
  *c->hits = ora_derivatives(c->g, c->prm, c->x, c->y, c->z, c->n, T, Rj, score, grad, H);   /* impl2:903-907 */
  (*c->sweeps)++;
  double phi_t = -*score, d_phi_t = 0;
  for (int a = 0; a < 6; a++) d_phi_t += grad[a] * dir[a];
  d_phi_t = -d_phi_t;
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
  int step_iterations = 0;
  while (!interval_converged && step_iterations < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {   /* impl2:920 */
    a_t = open_interval ? trial_value_mt(I, a_t, psi_t, d_psi_t) : trial_value_mt(I, a_t, phi_t, d_phi_t);
    a_t = cmax(cmin(a_t, step_max), step_min);                    /* impl2:936-937 */
    for (int a = 0; a < 6; a++) xt[a] = xv[a] + dir[a] * a_t;
    pose_to_f32(xt, T, Rj);
    memcpy(c->final_cm, T, sizeof T);
    *c->hits = ora_derivatives(c->g, c->prm, c->x, c->y, c->z, c->n, T, Rj, score, grad, H);  /* compute_hessian = false: */
    memset(H, 0, 36 * sizeof(double));                                                        /* hessian.setZero() only   */
    (*c->sweeps)++;
    phi_t = -*score; d_phi_t = 0;
    for (int a = 0; a < 6; a++) d_phi_t += grad[a] * dir[a];
    d_phi_t = -d_phi_t;
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t; d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {          /* impl2:963-974 */
      open_interval = 0;
      I[1] = I[1] + phi_0 - mu * d_phi_0 * I[0]; I[2] = I[2] + mu * d_phi_0;
      I[4] = I[4] + phi_0 - mu * d_phi_0 * I[3]; I[5] = I[5] + mu * d_phi_0;
    }
    interval_converged = open_interval ? update_interval_mt(I, a_t, psi_t, d_psi_t) : update_interval_mt(I, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) ora_compute_hessian(c->g, c->prm, c->x, c->y, c->z, c->n, T, xt, H);   /* impl2:999-1000 */
  *c->mt_loops += step_iterations;
  return a_t;
}

/* computeTransformation, ndt_omp_impl2.hpp:87-188 + computeStepLengthMT (impl2:841-1003; its loop and computeHessian are
 * live iff step_size <= trans_epsilon/2). */
int ora_align(const ora_grid* g, const ora_params* prm,
              const float* x, const float* y, const float* z, size_t n,
              const float guess[16], ora_result* out) {
  memset(out, 0, sizeof *out);
  const double eps = prm->trans_epsilon;
  const double step_max = prm->step_size, step_min = eps / 2;
  if (!g) return -3;
  /* final_transformation_ = guess (or Identity, same thing) impl2:102-108 */
  float T[16], Rj[9];
  memcpy(T, guess, sizeof T);
  memcpy(out->final_colmajor, guess, sizeof T);
  /* p = SE3(R,t).log() impl2:120-121 */
  double p[6];
  {
    double M[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M[r * 4 + c] = (double)guess[c * 4 + r];
    ora_se3_log(M, p);
    float Tdummy[16];
    pose_to_f32(p, Tdummy, Rj);                                    /* Jacobian uses exp(p), impl2:508 */
  }
  double score, grad[6], H[36];
  long hits = ora_derivatives(g, prm, x, y, z, n, T, Rj, &score, grad, H);   /* impl2:129 */
  int sweeps = 1, it = 0, converged = 0, mt_loops = 0;
  mt_ctx ctx = {g, prm, x, y, z, n, out->final_colmajor, &hits, &sweeps, &mt_loops};
  /* pcl::Registration::align(): transformation_ = previous_transformation_ = Identity before computeTransformation */
  for (int a = 0; a < 16; a++) out->inc_colmajor[a] = out->prev_inc_colmajor[a] = (a % 5 == 0) ? 1.f : 0.f;
  while (!converged) {
    memcpy(out->prev_inc_colmajor, out->inc_colmajor, sizeof out->inc_colmajor);   /* impl2:134 */
    double neg[6], dp[6];
    for (int a = 0; a < 6; a++) neg[a] = -grad[a];
    newton_solve6(H, neg, dp);                                     /* impl2:138-140 (canonical: ora_svd_solve6) */
    double nrm = 0;
    for (int a = 0; a < 6; a++) nrm += dp[a] * dp[a];
    nrm = sqrt(nrm);
    if (nrm == 0 || nrm != nrm) {                                  /* impl2:147-152 */
      out->trans_probability = score / (double)n;
      out->converged = (nrm == nrm);
      out->iterations = it; out->score = score; out->hits_last = hits; out->sweeps = sweeps; out->mt_loops = mt_loops;
      return 0;
    }
    for (int a = 0; a < 6; a++) dp[a] /= nrm;                      /* normalize() impl2:154 */
    const double a_t = step_length_mt(&ctx, p, dp, nrm, step_max, step_min, &score, grad, H);   /* impl2:155 */
    for (int a = 0; a < 6; a++) dp[a] *= a_t;                      /* impl2:156 */
    {                                                              /* transformation_ = float(exp(delta_p)), impl2:163 */
      float Ti[16], Ri[9];
      pose_to_f32(dp, Ti, Ri);
      memcpy(out->inc_colmajor, Ti, sizeof Ti);
    }
    double pn[6];
    ora_se3_compose_log(dp, p, pn);                                /* impl2:166 */
    memcpy(p, pn, sizeof pn);
    if (it > prm->max_iterations || (it && (fabs(a_t) < eps))) converged = 1;   /* impl2:175-179 */
    it++;
  }
  out->trans_probability = score / (double)n;                      /* impl2:187 */
  out->converged = 1; out->iterations = it; out->score = score; out->hits_last = hits; out->sweeps = sweeps; out->mt_loops = mt_loops;
  return 0;
}

/* ------------------------------------------------------------------ call-site policy of the odometry node
 * ScanMatchingOdomNodelet::matching_s2k after the align (src/lidar_odometry/scan_matching_odom_nodelet.cpp:229-250): tf_s2s,
 * odom_velo, the keyframe test and the constant-velocity guess, in f64 with an explicit operation order (Matrix4d products with
 * k ascending, the general 4x4 inverse by cofactors; Eigen's own order is not observable here -- canonical choices shared with
 * the device policy in lv_slam_amd/csrc/ndt_sequence.hpp).  Matrices are 4x4 f64 ROW-major except final_cm (column-major f32). */
static void mul4(const double A[16], const double B[16], double C[16]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      C[i * 4 + j] = ((A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j]) + A[i * 4 + 2] * B[2 * 4 + j]) + A[i * 4 + 3] * B[3 * 4 + j];
}
static double det3h(const double* m, int i1, int i2, int i3, int j1, int j2, int j3) {
  return m[i1 * 4 + j1] * (m[i2 * 4 + j2] * m[i3 * 4 + j3] - m[i2 * 4 + j3] * m[i3 * 4 + j2]);
}
static double cof4(const double* m, int i, int j) {
  const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4, j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
  return (det3h(m, i1, i2, i3, j1, j2, j3) + det3h(m, i2, i3, i1, j1, j2, j3)) + det3h(m, i3, i1, i2, j1, j2, j3);
}
static void inv4(const double M[16], double R[16]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      const double c = cof4(M, i, j);
      R[j * 4 + i] = ((i + j) & 1) ? -c : c;
    }
  const double det = ((M[0] * R[0] + M[4] * R[1]) + M[8] * R[2]) + M[12] * R[3];
  for (int a = 0; a < 16; a++) R[a] = R[a] / det;
}
/* w of Eigen::Quaternionf(R.cast<float>()) */
static float quat_w_f32(const float m[9]) {
  float t = (m[0] + m[4]) + m[8];
  if (t > 0.f) return 0.5f * sqrtf(t + 1.0f);
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > m[i * 3 + i]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = sqrtf(((m[i * 3 + i] - m[j * 3 + j]) - m[k * 3 + k]) + 1.0f);
  return (m[k * 3 + j] - m[j * 3 + k]) * (0.5f / t);
}
/* state = {pre_tf_s2k[16], key_pose[16], keyframe_stamp}; thr = {keyframe_delta_trans, _angle, _time} (:67-76).
 * Outputs: odom_velo (:234), the next guess (:250, f64), quantities of the keyframe test; returns 1 when the scan became the keyframe. */
int ora_policy_step(double pre_tf_s2k[16], double key_pose[16], double* keyframe_stamp, const float final_cm[16], double stamp,
                    const double thr[3], double odom[16], double guess[16], double test[3]) {
  double tf[16], inv[16], s2s[16];
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) tf[r * 4 + c] = (double)final_cm[c * 4 + r];
  inv4(pre_tf_s2k, inv);
  mul4(inv, tf, s2s);                                                            /* :231 */
  mul4(key_pose, tf, odom);                                                      /* :234 */
  /* :237 block<3,1>(0,3).norm(): Eigen 3.3's unrolled redux of a fixed 3-vector is x0^2 + (x1^2 + x2^2) (same tree as ORA_VAR_NORM_TREE notes for mean_.norm()) */
  const double dx = sqrt(tf[3] * tf[3] + (tf[7] * tf[7] + tf[11] * tf[11]));
  float Rf[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rf[r * 3 + c] = final_cm[c * 4 + r];
  const double da = (double)(2.f * acosf(quat_w_f32(Rf)));                       /* :238: std::acos(float) is the float overload */
  const double dt = stamp - *keyframe_stamp;                                     /* :239 */
  test[0] = dx; test[1] = da; test[2] = dt;
  int key = 0;
  if (dx > thr[0] || da > thr[1] || dt > thr[2]) {                               /* :240-247 */
    key = 1;
    for (int a = 0; a < 16; a++) { tf[a] = (a % 5 == 0) ? 1.0 : 0.0; key_pose[a] = odom[a]; }
    *keyframe_stamp = stamp;
  }
  memcpy(pre_tf_s2k, tf, sizeof tf);                                             /* :249 */
  mul4(tf, s2s, guess);                                                          /* :250 */
  return key;
}

/* pcl::Registration::getFitnessScore(max_range) == InformationMatrixCalculator::calc_fitness_score
 * (src/global_graph/information_matrix_calculator.cpp:53-87): source moved by T (f32, PCL scalar form), exact nearest
 * target point (brute force here; FLANN's L2_Simple f32 accumulation order), squared distance compared with max_range
 * and averaged.  Returns DBL_MAX when no point is in range.  *n_in = number of source points counted. */
double ora_fitness_score(const float* tx, const float* ty, const float* tz, size_t nt,
                         const float* sx, const float* sy, const float* sz, size_t ns,
                         const float T[16], double max_range, long* n_in) {
  size_t nchunks = (ns + CHUNK - 1) / CHUNK;
  double* part = (double*)calloc(nchunks ? nchunks : 1, 2 * sizeof(double));
#ifdef _OPENMP
  int nthr = g_threads > 0 ? g_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr)
#endif
  for (long c = 0; c < (long)nchunks; c++) {
    size_t i0 = (size_t)c * CHUNK, i1 = i0 + CHUNK < ns ? i0 + CHUNK : ns;
    for (size_t i = i0; i < i1; i++) {
      float q[3];
      for (int a = 0; a < 3; a++) q[a] = ((T[0 * 4 + a] * sx[i] + T[1 * 4 + a] * sy[i]) + T[2 * 4 + a] * sz[i]) + T[3 * 4 + a];
      if (!finite3(q[0], q[1], q[2])) continue;
      float best = INFINITY;
      for (size_t j = 0; j < nt; j++) {
        float dx = q[0] - tx[j], dy = q[1] - ty[j], dz = q[2] - tz[j];
        float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < best) best = d2;
      }
      if ((double)best <= max_range) { part[2 * c] += (double)best; part[2 * c + 1] += 1; }
    }
  }
  double sum = 0, cnt = 0;
  for (size_t c = 0; c < nchunks; c++) { sum += part[2 * c]; cnt += part[2 * c + 1]; }
  free(part);
  if (n_in) *n_in = (long)cnt;
  return cnt > 0 ? sum / cnt : DBL_MAX;
}

/* PrefilteringNodelet::distance_filter + downsample (src/lidar_odometry/prefiltering_nodelet.cpp:137-181):
 * keep near < |p| < far (f32 norm, double compare), then pcl::VoxelGrid (PCL 1.8 voxel_grid.hpp) centroid per voxel:
 * f32 sums (AccumulatorXYZ) divided by the count, output in ascending voxel index.  The reference sorts the
 * (voxel, point) pairs with an unstable std::sort, so its f32 summation order inside a voxel is unspecified; this
 * restatement sums in input order.  leaf <= 0: no down-sampling.  Index overflow guard: output = filtered input.
 * out_* must hold n entries; returns the number of output points. */
size_t ora_prefilter(const float* x, const float* y, const float* z, size_t n, int use_df, double dnear, double dfar, float leaf,
                     float* ox, float* oy, float* oz) {
  unsigned char* keep = (unsigned char*)malloc(n ? n : 1);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t nk = 0;
  for (size_t i = 0; i < n; i++) {
    int ok = 1;
    if (use_df) { double d = (double)sqrtf((x[i] * x[i] + y[i] * y[i]) + z[i] * z[i]); ok = d > dnear && d < dfar; }
    ok = ok && finite3(x[i], y[i], z[i]);
    keep[i] = (unsigned char)ok;
    if (!ok) continue;
    nk++;
    if (x[i] < mn[0]) mn[0] = x[i]; if (x[i] > mx[0]) mx[0] = x[i];
    if (y[i] < mn[1]) mn[1] = y[i]; if (y[i] > mx[1]) mx[1] = y[i];
    if (z[i] < mn[2]) mn[2] = z[i]; if (z[i] > mx[2]) mx[2] = z[i];
  }
  int down = leaf > 0 && nk > 0;
  float inv = down ? 1.0f / leaf : 0.f;
  int min_b[3] = {0, 0, 0}, mul1 = 0, mul2 = 0;
  if (down) {
    if (grid_too_big((mx[0] - mn[0]) * inv, (mx[1] - mn[1]) * inv, (mx[2] - mn[2]) * inv)) down = 0;
    else {
      int maxb[3];
      for (int a = 0; a < 3; a++) { min_b[a] = (int)floorf(mn[a] * inv); maxb[a] = (int)floorf(mx[a] * inv); }
      mul1 = maxb[0] - min_b[0] + 1;
      mul2 = mul1 * (maxb[1] - min_b[1] + 1);
    }
  }
  size_t m = 0;
  if (!down) {
    for (size_t i = 0; i < n; i++) if (keep[i]) { ox[m] = x[i]; oy[m] = y[i]; oz[m] = z[i]; m++; }
    free(keep);
    return m;
  }
  keypos* kp = (keypos*)malloc(nk * sizeof(keypos));
  size_t c = 0;
  for (size_t i = 0; i < n; i++) {
    if (!keep[i]) continue;
    int i0 = (int)(floorf(x[i] * inv) - (float)min_b[0]), i1 = (int)(floorf(y[i] * inv) - (float)min_b[1]),
        i2 = (int)(floorf(z[i] * inv) - (float)min_b[2]);
    kp[c].idx = i0 + i1 * mul1 + i2 * mul2;
    kp[c].pos = (uint32_t)i;
    c++;
  }
  qsort(kp, nk, sizeof(keypos), cmp_keypos);
  for (size_t s = 0; s < nk;) {
    size_t e = s;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    while (e < nk && kp[e].idx == kp[s].idx) { size_t i = kp[e].pos; sx += x[i]; sy += y[i]; sz += z[i]; e++; }
    float fn = (float)(e - s);
    ox[m] = sx / fn; oy[m] = sy / fn; oz[m] = sz / fn;
    m++;
    s = e;
  }
  free(kp); free(keep);
  return m;
}

#include "ndt_oracle_refshape.inc"
