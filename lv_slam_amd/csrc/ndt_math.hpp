// ndt_math.hpp -- f64 control math of the NDT engine, device side (gfx950).
//
// Runs inside the Newton-update kernel (one pair per wave) so that the whole Gauss-Newton loop of
// computeTransformation (include/ndt_omp/ndt_omp_impl2.hpp:87-188) stays on the GPU: no host
// round-trip between derivative sweeps.
//
// Third-party arithmetic restated from its published form (sources are not in the lv_slam tree):
//   Sophus a621ff2 (lv_slam README.md:61-66): non-templated SE3/SO3 exp, log, operator*
//   Eigen 3.3: Quaternion<->Matrix3, 3x3 cofactor inverse, JacobiSVD::solve threshold semantics.
// Compiled with -ffp-contract=off: no FMA contraction anywhere in this file.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

namespace ndtm {

#define NDT_SMALL_EPS 1e-10

struct Quat { double w, x, y, z; };
struct SE3 { Quat q; double t[3]; };

__device__ inline Quat q_normalized(Quat q) {
  double n = sqrt(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
  Quat r = {q.w / n, q.x / n, q.y / n, q.z / n};
  return r;
}

// Eigen quaternionbase_assign_impl<Matrix3>
__device__ inline Quat q_from_matrix(const double m[9]) {
  Quat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[7] - m[5]) * t;
    q.y = (m[2] - m[6]) * t;
    q.z = (m[3] - m[1]) * t;
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

// Eigen QuaternionBase::toRotationMatrix
__device__ inline void q_to_matrix(Quat q, double r[9]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r[0] = 1 - (tyy + tzz); r[1] = txy - twz; r[2] = txz + twy;
  r[3] = txy + twz; r[4] = 1 - (txx + tzz); r[5] = tyz - twx;
  r[6] = txz - twy; r[7] = tyz + twx; r[8] = 1 - (txx + tyy);
}

__device__ inline Quat q_mul(Quat a, Quat b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

// QuaternionBase::_transformVector
__device__ inline void q_rotate(Quat q, const double v[3], double out[3]) {
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  for (int i = 0; i < 3; i++) out[i] = (v[i] + q.w * uv[i]) + c[i];
}

__device__ inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[i * 3 + j] = (A[i * 3 + 0] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}

__device__ inline double cof3(const double* m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
// Eigen 3.3 compute_inverse<Matrix3d> (cofactor expansion)
__device__ inline void mat3_inverse(const double m[9], double r[9]) {
  double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
  double invdet = 1.0 / det;
  r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
  r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

__device__ inline void hat3(const double o[3], double O[9]) {
  O[0] = 0; O[1] = -o[2]; O[2] = o[1];
  O[3] = o[2]; O[4] = 0; O[5] = -o[0];
  O[6] = -o[1]; O[7] = o[0]; O[8] = 0;
}

// SO3::expAndTheta
__device__ inline Quat so3_exp(const double om[3], double* theta) {
  *theta = sqrt((om[0] * om[0] + om[1] * om[1]) + om[2] * om[2]);
  double half = 0.5 * (*theta);
  double imag, real = cos(half);
  if (*theta < NDT_SMALL_EPS) {
    double th2 = (*theta) * (*theta), th4 = th2 * th2;
    imag = 0.5 - 0.0208333 * th2 + 0.000260417 * th4;
  } else {
    imag = sin(half) / (*theta);
  }
  Quat q = {real, imag * om[0], imag * om[1], imag * om[2]};
  return q_normalized(q);
}

// SO3::logAndTheta
__device__ inline void so3_log(Quat q, double om[3], double* theta) {
  double n = sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
  double w = q.w, f;
  if (n < NDT_SMALL_EPS) {
    f = 2. / w - 2. * (n * n) / (w * (w * w));
  } else {
    f = 2 * atan(n / w) / n;
  }
  *theta = f * n;
  om[0] = f * q.x; om[1] = f * q.y; om[2] = f * q.z;
}

// SE3::exp, tangent [upsilon; omega]
__device__ inline SE3 se3_exp(const double p[6]) {
  SE3 r;
  double theta;
  r.q = so3_exp(p + 3, &theta);
  double Om[9], Om2[9], V[9];
  hat3(p + 3, Om);
  mat3_mul(Om, Om, Om2);
  if (theta < NDT_SMALL_EPS) {
    q_to_matrix(r.q, V);
  } else {
    double th2 = theta * theta;
    double a = (1 - cos(theta)) / th2, b = (theta - sin(theta)) / (th2 * theta);
    for (int i = 0; i < 9; i++) V[i] = (((i % 4) == 0 ? 1.0 : 0.0) + a * Om[i]) + b * Om2[i];
  }
  for (int i = 0; i < 3; i++) r.t[i] = (V[i * 3 + 0] * p[0] + V[i * 3 + 1] * p[1]) + V[i * 3 + 2] * p[2];
  return r;
}

// SE3::log
__device__ inline void se3_log(SE3 s, double p[6]) {
  double theta, om[3];
  so3_log(s.q, om, &theta);
  double Om[9], Om2[9], Vi[9];
  hat3(om, Om);
  mat3_mul(Om, Om, Om2);
  double c = (theta < NDT_SMALL_EPS) ? (1. / 12.) : (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  for (int i = 0; i < 9; i++) Vi[i] = (((i % 4) == 0 ? 1.0 : 0.0) - 0.5 * Om[i]) + c * Om2[i];
  for (int i = 0; i < 3; i++) p[i] = (Vi[i * 3 + 0] * s.t[0] + Vi[i * 3 + 1] * s.t[1]) + Vi[i * 3 + 2] * s.t[2];
  p[3] = om[0]; p[4] = om[1]; p[5] = om[2];
}

// SE3::operator*
__device__ inline SE3 se3_mul(SE3 a, SE3 b) {
  SE3 r;
  double rt[3];
  q_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.q = q_normalized(q_mul(a.q, b.q));
  return r;
}

// SE3(R, t): SO3(R) goes through the (normalised) quaternion
__device__ inline SE3 se3_from_Rt(const double R[9], const double t[3]) {
  SE3 s;
  s.q = q_normalized(q_from_matrix(R));
  s.t[0] = t[0]; s.t[1] = t[1]; s.t[2] = t[2];
  return s;
}

// float(exp(p).matrix()): T = 3x4 row-major [R|t], Rj = R (both f32)
__device__ inline void pose_to_f32(const double p[6], float T[12], float Rj[9]) {
  SE3 s = se3_exp(p);
  double R[9];
  q_to_matrix(s.q, R);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) { T[r * 4 + c] = (float)R[r * 3 + c]; Rj[r * 3 + c] = (float)R[r * 3 + c]; }
    T[r * 4 + 3] = (float)s.t[r];
  }
}

// ---- exp of an f32 argument, evaluated in f64 and rounded to f32 (the canonical choice for impl2:581) ----------
// Table-driven: x = n*ln2/64 + r, exp(x) = 2^(n>>6) * 2^((n&63)/64) * exp(r), |r| <= ln2/128, exp(r) - 1 by its degree-5 Taylor
// polynomial (truncation 3.5e-17).  Total error < 1 ulp of f64, like the device library's exp, but 12 instead of 28 f64-rate
// instructions -- the sweep is VALU-issue bound and this exp was 12 % of it.  The table holds correctly rounded 2^(j/64).
__constant__ double c_exp2_64[64] = {
  0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
  0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
  0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
  0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
  0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
  0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
  0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
  0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
  0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
  0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
  0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
  0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
  0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
  0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
  0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
  0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0
};
#define NDT_EXP_INV   0x1.71547652b82fep+6      // 64 / ln 2
#define NDT_EXP_C_HI  0x1.62e42fee00000p-7      // ln 2 / 64, low 21 significand bits zero: n * HI is exact for |n| < 2^21
#define NDT_EXP_C_LO  0x1.a39ef35793c76p-39     // ln 2 / 64 - HI
__device__ __forceinline__ float exp_f32arg(float a, const double* __restrict__ tab /* c_exp2_64, normally staged in LDS */) {
  a = a < -800.f ? -800.f : a;                   // exp underflows / overflows f64 long before; NaN fails both tests and stays NaN
  a = a > 800.f ? 800.f : a;
  const double x = (double)a;
  const double n = rint(x * NDT_EXP_INV);
  double r = fma(n, -NDT_EXP_C_HI, x);
  r = fma(n, -NDT_EXP_C_LO, r);
  const int ni = (int)n;
  const double t = tab[ni & 63];
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(r, p, 1.0 / 6.0);
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  p = r * p;                                     // exp(r) - 1
  return (float)ldexp(fma(t, p, t), ni >> 6);
}

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi; lower triangle read, ascending eigenvalues,
// eigenvectors = columns of V).  Stands in for Eigen::SelfAdjointEigenSolver
// (voxel_grid_covariance_omp_impl.hpp:333-335).
__device__ inline void eigen_sym3(const double Ain[9], double evals[3], double V[9]) {
  double a[3][3], v[3][3];
  a[0][0] = Ain[0]; a[1][1] = Ain[4]; a[2][2] = Ain[8];
  a[0][1] = a[1][0] = Ain[3]; a[0][2] = a[2][0] = Ain[6]; a[1][2] = a[2][1] = Ain[7];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off == 0.0) break;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int p = (k == 2) ? 1 : 0, q = (k == 0) ? 1 : 2, r = 3 - p - q;
      double apq = a[p][q];
      if (apq == 0.0) continue;
      double g = 100.0 * fabs(apq);
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
      double arp = a[r][p], arq = a[r][q];
      a[r][p] = a[p][r] = arp - s * (arq + arp * tau);
      a[r][q] = a[q][r] = arq + s * (arp - arq * tau);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        double vip = v[i][p], viq = v[i][q];
        v[i][p] = vip - s * (viq + vip * tau);
        v[i][q] = viq + s * (vip - viq * tau);
      }
    }
  }
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  int o0 = 0, o1 = 1, o2 = 2, t;
  if (d[o1] < d[o0]) { t = o0; o0 = o1; o1 = t; }
  if (d[o2] < d[o1]) { t = o1; o1 = o2; o2 = t; }
  if (d[o1] < d[o0]) { t = o0; o0 = o1; o1 = t; }
  const int o[3] = {o0, o1, o2};
  for (int j = 0; j < 3; j++) {
    evals[j] = d[o[j]];
    for (int i = 0; i < 3; i++) V[i * 3 + j] = v[i][o[j]];
  }
}

// x = pinv(H) b with Eigen::JacobiSVD<6x6>::solve semantics (ndt_omp_impl2.hpp:138-140):
// rank = #{sigma_i >= max(sigma_max * 6 eps, DBL_MIN)}.  One-sided (Hestenes) Jacobi.
__device__ inline void svd_solve6(const double* H, const double b[6], double x[6]) {
  // A non-finite entry in H or b: Eigen 3.3's JacobiSVD keeps rank = 6 (NaN singular values fail the `< threshold` test of
  // SVDBase::rank()) and solve() multiplies through, so every component of the answer is NaN -- which computeTransformation
  // reports as converged_ = false (impl2:147-151).  A thresholded pseudo-inverse written with `!(sigma >= thr)` would
  // return 0 (= "converged") instead.
  {
    bool fin = true;
    for (int i = 0; i < 36; i++) fin = fin && isfinite(H[i]);
    for (int i = 0; i < 6; i++) fin = fin && isfinite(b[i]);
    if (!fin) { for (int i = 0; i < 6; i++) x[i] = __longlong_as_double(0x7ff8000000000000ll); return; }
  }
  // Eigen 3.3 JacobiSVD::compute works on matrix / scale, scale = matrix.cwiseAbs().maxCoeff() (1 for a zero matrix), and multiplies
  // the singular values back at the end: without it the squared column norms below overflow (H entries beyond ~1e77: ndt_pca's
  // weights compound multiplicatively over DIRECT26 neighbours) or vanish, and the rotations are skipped.
  double scale = 0;
  for (int i = 0; i < 36; i++) scale = fabs(H[i]) > scale ? fabs(H[i]) : scale;
  if (scale == 0.0) scale = 1.0;
  double A[6][6], V[6][6];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) { A[i][j] = H[i * 6 + j] / scale; V[i][j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll 1
  for (int sweep = 0; sweep < 60; sweep++) {
    int rotated = 0;
    // p, q fully unrolled: every A/V index is a compile-time constant, so both matrices live in registers
#pragma unroll
    for (int p = 0; p < 5; p++) {
#pragma unroll
      for (int q = p + 1; q < 6; q++) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (gamma == 0.0 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
        rotated = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        if (zeta < 0) t = -t;
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
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
  double sig[6], sval[6], smax = 0;                  // sig: of the scaled work matrix; sval = sig * scale: the singular values
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double s2 = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) s2 += A[i][j] * A[i][j];
    sig[j] = sqrt(s2);
    sval[j] = sig[j] * scale;
    if (sval[j] > smax) smax = sval[j];
  }
  double thr = smax * (6.0 * DBL_EPSILON);
  if (thr < DBL_MIN) thr = DBL_MIN;
  for (int i = 0; i < 6; i++) x[i] = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    if (!(sval[j] >= thr) || sig[j] == 0.0) continue;
    double ub = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) ub += (A[i][j] / sig[j]) * b[i];       // column j of U, times b
    double w = ub / sval[j];
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] += V[i][j] * w;
  }
}

// Fast path of the Newton solve: LU with partial pivoting, accepted only where it is interchangeable with the reference's
// JacobiSVD solve.  `lu_solve6_rhs` eliminates [H | rhs] and reports the extreme pivots; the acceptance test
// (`lu_accept`) wants a clear pivot ratio AND a Frobenius condition number ||H||_F ||H^-1||_F below LU_MAX_COND: the Hessians of
// the benchmark workloads sit at 3e2 .. 4e3 (1 m and 0.5 m grids, ndt_omp and ndt_pca), where LU and SVD agree to ~1e-13 relative
// and no pose bit depends on the choice (profiles/r03_order_sensitivity_*.json); anything less well conditioned -- few hits, ndt_pca
// weights compounded over DIRECT26 neighbours, clouds far from the origin -- takes the reference's own route (svd_solve6).
// ~100x cheaper than the Jacobi SVD on one lane.
#define LU_MAX_COND 1e5
__device__ inline bool lu_solve6_rhs(const double* H, const double rhs[6], double x[6], double& pmin, double& pmax) {
  double A[6][7];
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) A[i][j] = H[i * 6 + j];
    A[i][6] = rhs[i];
  }
  pmin = DBL_MAX; pmax = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    // partial pivoting: bring the largest |A[i][k]|, i >= k, to row k (branch-free row swaps keep A in registers)
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const bool sw = fabs(A[i][k]) > fabs(A[k][k]);
#pragma unroll
      for (int j = k; j < 7; j++) {
        const double a = A[k][j], c = A[i][j];
        A[k][j] = sw ? c : a;
        A[i][j] = sw ? a : c;
      }
    }
    const double piv = A[k][k];
    const double ap = fabs(piv);
    pmin = ap < pmin ? ap : pmin;
    pmax = ap > pmax ? ap : pmax;
    if (!(ap > 0.0)) return false;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double f = A[i][k] / piv;
#pragma unroll
      for (int j = k + 1; j < 7; j++) A[i][j] -= f * A[k][j];
    }
  }
  if (!(pmin > 1e-6 * pmax) || !(pmax < DBL_MAX)) return false;
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = A[i][6];
#pragma unroll
    for (int j = i + 1; j < 6; j++) s -= A[i][j] * x[j];
    x[i] = s / A[i][i];
  }
  return true;
}
// The condition estimate (||H||_F ||H^-1||_F)^2 from row sums of H and column sums of H^-1, added in index order.  No scaling: where
// the squares leave the f64 range (entries beyond 1e154 or below 1e-154) the product is inf or NaN and the LU answer is refused --
// svd_solve6 scales, and is the route for such matrices anyway.
__device__ inline double norm2_6(const double v[6]) {
  double n2 = 0;
  for (int i = 0; i < 6; i++) n2 += v[i] * v[i];
  return n2;
}
__device__ inline bool lu_accept(double hF2, double invF2) {
  const double c2 = hF2 * invF2;                   // >= cond_2(H)^2
  return c2 < LU_MAX_COND * LU_MAX_COND;           // (inf, NaN: refused)
}
// One lane on its own: the solution for b and the six columns of H^-1, one elimination after the other.
__device__ inline bool lu_solve6(const double* H, const double b[6], double x[6]) {
  double pmin, pmax;
  if (!lu_solve6_rhs(H, b, x, pmin, pmax)) return false;
  double hF2 = 0, invF2 = 0;
#pragma unroll 1
  for (int k = 0; k < 6; k++) {
    double e[6], col[6];
#pragma unroll
    for (int a = 0; a < 6; a++) e[a] = (a == k) ? 1.0 : 0.0;
    if (!lu_solve6_rhs(H, e, col, pmin, pmax)) return false;
    invF2 += norm2_6(col);
    hF2 += norm2_6(H + 6 * k);
  }
  return lu_accept(hF2, invF2);
}

}  // namespace ndtm
