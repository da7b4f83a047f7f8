// ndt_update.hpp -- Newton control on the device: computeTransformation's loop body and the live prefix of
// computeStepLengthMT (include/ndt_omp/ndt_omp_impl2.hpp:87-188, 841-907), plus pose set-up helpers.
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_sweep.hpp"

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

// transformation_ = (Sophus::SE3::exp(delta_p).matrix()).cast<float>() (impl2:163); `e` = exp(delta_p)
__device__ inline void set_increment_se3(PairState& S, const ndtm::SE3& e) {
  double R[9];
  ndtm::q_to_matrix(e.q, R);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) S.inc_cm[c * 4 + r] = (float)R[r * 3 + c];
    S.inc_cm[12 + r] = (float)e.t[r];
    S.inc_cm[r * 4 + 3] = 0.f;
  }
  S.inc_cm[15] = 1.f;
}
__device__ inline void set_increment(PairState& S, const double dp[6]) { set_increment_se3(S, ndtm::se3_exp(dp)); }

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }
__device__ __forceinline__ ndtm::SE3 shfl_se3(const ndtm::SE3& e, int src) {
  ndtm::SE3 r;
  r.q.w = shfl_d(e.q.w, src); r.q.x = shfl_d(e.q.x, src); r.q.y = shfl_d(e.q.y, src); r.q.z = shfl_d(e.q.z, src);
  r.t[0] = shfl_d(e.t[0], src); r.t[1] = shfl_d(e.t[1], src); r.t[2] = shfl_d(e.t[2], src);
  return r;
}

// computeTransformation's set-up for one pair (impl2:102-129): p = log(guess), first sweep moves the cloud by the f32 guess itself
__device__ inline void init_pair_state(PairState& S, const float G[16] /* column-major */, int n_src, int grid_status) {
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
  S.it = 0; S.phase = PH_SWEEP0; S.converged = 0; S.sweeps = 1; S.a_t = 0; S.hits = 0; S.score = 0; S.mt_loops = 0;
  for (int a = 0; a < 16; a++) S.inc_cm[a] = S.prev_inc_cm[a] = (a % 5 == 0) ? 1.f : 0.f;    // align(): transformation_ = previous_ = I
  S.n_src = n_src;
  S.grid_status = grid_status;
  S.reb_tag = -1;
}

// p = SE3(R,t).log(); first sweep moves the cloud by the caller's f32 guess itself (impl2:102-129)
NDT_KERNEL void k_init_state(PairState* st, const float* __restrict__ guess_cm, const int* __restrict__ src_cnt,
                             const GridDesc* __restrict__ gd, int n_pairs, int* active_list, SweepCtl* ctl) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_pairs) return;
  active_list[b] = b;                            // the first sweep covers every pair
  if (b == 0) ctl->n_active = n_pairs;
  init_pair_state(st[b], guess_cm + (size_t)b * 16, src_cnt[b], gd[b].status);
}

// explicit sweep pose (parity hooks)
NDT_KERNEL void k_set_pose(PairState* st, int b, const float* __restrict__ T_cm, const float* __restrict__ Rj, const int* __restrict__ src_cnt,
                           const GridDesc* __restrict__ gd, int* active_list, SweepCtl* ctl) {
  PairState& S = st[b];
  active_list[0] = b; ctl->n_active = 1;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) S.T[r * 4 + c] = T_cm[c * 4 + r];
  for (int a = 0; a < 9; a++) S.Rj[a] = Rj[a];
  S.phase = PH_SWEEP0; S.n_src = src_cnt[b]; S.grid_status = gd[b].status; S.it = 0; S.sweeps = 1;
}
NDT_KERNEL void k_set_pose_p(PairState* st, int b, const double* __restrict__ p, const int* __restrict__ src_cnt, const GridDesc* __restrict__ gd,
                             int* active_list, SweepCtl* ctl, int for_hessian) {
  PairState& S = st[b];
  active_list[0] = b; ctl->n_active = 1;
  double pp[6];
  for (int a = 0; a < 6; a++) { pp[a] = p[a]; S.xt[a] = p[a]; }
  ndtm::pose_to_f32(pp, S.T, S.Rj);
  S.phase = for_hessian ? PH_HESS : PH_SWEEP0; S.n_src = src_cnt[b]; S.grid_status = gd[b].status; S.it = 0; S.sweeps = 1;
}

// ---- More-Thuente pieces (impl2:717-838), live only when step_size <= eps/2 (impl2:888) -------------------------------
// std::min / std::max as libstdc++ evaluates them: a NaN first argument is returned unchanged
__device__ inline double mt_cmin(double a, double b) { return b < a ? b : a; }
__device__ inline double mt_cmax(double a, double b) { return a < b ? b : a; }

// updateIntervalMT (impl2:717-755); I = {a_l, f_l, g_l, a_u, f_u, g_u}
__device__ inline bool mt_update_interval(double I[6], double a_t, double f_t, double g_t) {
  if (f_t > I[1]) { I[3] = a_t; I[4] = f_t; I[5] = g_t; return false; }
  if (g_t * (I[0] - a_t) > 0) { I[0] = a_t; I[1] = f_t; I[2] = g_t; return false; }
  if (g_t * (I[0] - a_t) < 0) { I[3] = I[0]; I[4] = I[1]; I[5] = I[2]; I[0] = a_t; I[1] = f_t; I[2] = g_t; return false; }
  return true;
}

// trialValueSelectionMT (impl2:758-838)
__device__ inline double mt_trial_value(const double I[6], double a_t, double f_t, double g_t) {
  const double a_l = I[0], f_l = I[1], g_l = I[2], a_u = I[3], f_u = I[4], g_u = I[5];
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? mt_cmin(a_t + 0.66 * (a_u - a_t), a_n) : mt_cmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  const double w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

// The loop of computeStepLengthMT (impl2:920-994) after the first trial's sweep has been reduced into S.score / S.g.
// With step_max <= step_min every clamped trial value is step_min again (or NaN, which std::min/max let through), so the
// reference re-sweeps an unchanged pose: those evaluations are reused here, not re-run (identical by determinism); a NaN
// trial value means a NaN cloud, which has no neighbours (score, gradient = 0).  Returns step_iterations.
__device__ inline int mt_loop(PairState& S, double step_max, double step_min) {
  const double mu = 1.e-4, nu = 0.9;
  const double phi_0 = S.phi0, d_phi_0 = S.dphi0;
  // auxilaryFunction_PsiMT / dPsiMT (ndt_omp.h:480-496) at a = 0
  double I[6] = {0, phi_0 - phi_0 - mu * d_phi_0 * 0.0, d_phi_0 - mu * d_phi_0, 0, phi_0 - phi_0 - mu * d_phi_0 * 0.0, d_phi_0 - mu * d_phi_0};
  bool interval_converged = (step_max - step_min) > 0, open_interval = true;      // impl2:888
  double a_t = S.a_t;
  const double score_c = S.score;
  double g_c[6];
  for (int a = 0; a < 6; a++) g_c[a] = S.g[a];
  double score = score_c, gd = 0;
  for (int a = 0; a < 6; a++) gd += g_c[a] * S.dir[a];
  double phi_t = -score, d_phi_t = -gd;
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
  int its = 0;
  while (!interval_converged && its < 10 && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    a_t = open_interval ? mt_trial_value(I, a_t, psi_t, d_psi_t) : mt_trial_value(I, a_t, phi_t, d_phi_t);
    a_t = mt_cmax(mt_cmin(a_t, step_max), step_min);                             // impl2:936-937
    if (a_t != a_t) { score = 0; gd = 0; } else { score = score_c; gd = 0; for (int a = 0; a < 6; a++) gd += g_c[a] * S.dir[a]; }
    phi_t = -score; d_phi_t = -gd;
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t; d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {                         // impl2:963-974
      open_interval = false;
      I[1] = I[1] + phi_0 - mu * d_phi_0 * I[0]; I[2] = I[2] + mu * d_phi_0;
      I[4] = I[4] + phi_0 - mu * d_phi_0 * I[3]; I[5] = I[5] + mu * d_phi_0;
    }
    interval_converged = open_interval ? mt_update_interval(I, a_t, psi_t, d_psi_t) : mt_update_interval(I, a_t, phi_t, d_phi_t);
    its++;
  }
  S.a_t = a_t;
  return its;
}

// ---- fixed-order reduction of one pair's partial rows ------------------------------------------------------------------
// A pair's sweep leaves one 44-double row per work item (score, g[6], H[36], hits).  Four consecutive rows form a CHUNK,
// ((r0 + r1) + r2) + r3 (impl2:298-302 adds per-thread sums in a fixed order; so does this).  Eight consecutive chunks form a
// GROUP, added in chunk order.  Group k belongs to wave k % UPD_WAVES (= 4) of the block, which adds its groups in ascending order;
// wave 0 then adds the four wave sums in wave order.  (Four waves = one per SIMD: the Newton step below wants 300+ VGPRs.)  The tree is a function of the number of chunks alone -- never of the batch,
// the launch geometry or which wave ran an item -- so batched and single runs of a pair stay bit-identical; up to eight chunks
// (16,384 points) it is the plain sequential sum.  Every wave has all 32 loads of a group in flight at once: the reduction of a
// 131,072-point pair (256 rows) is two memory round trips per wave instead of eight on one wave.
#define UPD_WAVES   4
#define UPD_THREADS (64 * UPD_WAVES)
// `chunk_rows`: the rows are chunk sums already (latency mode: the sweep's blocks add the four rows of a chunk themselves).
// SC1: the rows were written by other workgroups of THIS launch (persistent kernels): read them with agent-scope (L1-bypassing) loads.
template <bool SC1>
__device__ __forceinline__ double ld_row(const double* p) {
  if (SC1) return __longlong_as_double((long long)__hip_atomic_load((const gu64*)reinterpret_cast<const unsigned long long*>(p), RLX_AGENT));
  return *p;
}
template <bool SC1 = false>
__device__ __forceinline__ double reduce_pair_rows(const double* __restrict__ rows, int nchunks, bool take, double (*sm)[NACC], bool chunk_rows = false) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane < NACC) {
    double acc = 0.0;
    if (take) {
      const double* P = rows + lane;
      if (chunk_rows) {
        // one stored row per chunk: the loads of up to four of the wave's groups are in flight together (a 65,536-point pair in latency
        // mode has 128 chunk rows = 16 groups = four per wave: one memory round trip); the adds keep the order described above
#pragma unroll 1
        for (int c0 = w * 8; c0 < nchunks; c0 += 4 * 8 * UPD_WAVES) {
          double q[4][8];
#pragma unroll
          for (int g = 0; g < 4; g++)
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const int c = c0 + g * 8 * UPD_WAVES + u;
              q[g][u] = (c < nchunks) ? ld_row<SC1>(P + (size_t)c * NACC) : 0.0;
            }
#pragma unroll
          for (int g = 0; g < 4; g++) {
            if (c0 + g * 8 * UPD_WAVES >= nchunks) break;
            double gs = 0.0;
#pragma unroll
            for (int u = 0; u < 8; u++) if (c0 + g * 8 * UPD_WAVES + u < nchunks) gs += q[g][u];
            acc += gs;
          }
        }
      } else
#pragma unroll 1
      for (int c0 = w * 8; c0 < nchunks; c0 += 8 * UPD_WAVES) {
        double gs = 0.0;
        if (c0 + 8 <= nchunks) {
          double q[8][4];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const double* Q = P + (size_t)(c0 + u) * 4 * NACC;
#pragma unroll
            for (int k = 0; k < 4; k++) q[u][k] = ld_row<SC1>(Q + k * NACC);
          }
#pragma unroll
          for (int u = 0; u < 8; u++) gs += ((q[u][0] + q[u][1]) + q[u][2]) + q[u][3];
        } else {
          for (int c = c0; c < nchunks; c++) {
            const double* Q = P + (size_t)c * 4 * NACC;
            gs += ((ld_row<SC1>(Q) + ld_row<SC1>(Q + NACC)) + ld_row<SC1>(Q + 2 * NACC)) + ld_row<SC1>(Q + 3 * NACC);
          }
        }
        acc += gs;
      }
    }
    sm[w][lane] = acc;
  }
  __syncthreads();
  double v = 0.0;
  if (w == 0 && lane < NACC) {
#pragma unroll
    for (int k = 0; k < UPD_WAVES; k++) v += sm[k][lane];
  }
  return v;
}

enum { NEWTON_DONE = 0, NEWTON_SWEEP = 1, NEWTON_HESSIAN = 2 };

// impl2:138-140: JacobiSVD(H).solve(-g).  Well-conditioned H: exact LU solve (same answer to rounding); anything else
// (rank-deficient, H = 0, ill-conditioned): the thresholded pseudo-inverse itself.  (Non-finite g or H: the SVD route answers NaN,
// as Eigen's does, and the pair ends with converged = 0.)
__device__ __forceinline__ void newton_solve(const PairState& S, double d[6]) {
  double neg[6];
  for (int a = 0; a < 6; a++) neg[a] = -S.g[a];
  bool fin = true;
  for (int a = 0; a < 36; a++) fin = fin && isfinite(S.H[a]);
  for (int a = 0; a < 6; a++) fin = fin && isfinite(S.g[a]);
  if (!fin || !ndtm::lu_solve6(S.H, neg, d)) ndtm::svd_solve6(S.H, neg, d);
}
// When the More-Thuente loop is dead (mt = 0) the solve depends only on the reduced (g, H) -- not on the re-basing of p that
// wave 0 runs first -- so a second wave of the block computes it at the same time and hands it over through LDS
// (`sol`: d[6], then a ready flag).  Same function, same inputs: same bits.
//
// (Round 5 also built the elimination ROW-PARALLEL -- lane i < 6 owning row i of the 6 x 13 tableau [H | I | -g], pivot candidates by
// v_readlane, the row permutation by ds_bpermute, back substitution column-parallel through LDS: ~1 k instead of ~3 k instructions, bit-identical
// (tests/test_solve6_gpu.py) -- and measured it slower: 20.6 k against 16 k cycles per update inside the one-launch align, 13.7 k for the solve
// wave of k_seq_update; the cross-lane traffic of a 6-wide problem costs more than the straight-line code it saves.  docs/experiments.md 10d.)
#define SOL_WORDS 8
__device__ __forceinline__ void newton_solve_side(const PairState& S, volatile double* sol) {
  // lanes 0..5 eliminate [H | e_k] (the columns of H^-1, for the condition estimate), lane 6 [H | -g]: the seven eliminations of
  // ndtm::lu_solve6 side by side -- same functions, same operands, same order of the final sum: same bits, same decision
  const int lane = threadIdx.x & 63;
  if (lane > 6) return;
  double rhs[6], x[6];
  for (int a = 0; a < 6; a++) rhs[a] = lane < 6 ? (a == lane ? 1.0 : 0.0) : -S.g[a];
  bool fin = true;
  for (int a = 0; a < 36; a++) fin = fin && isfinite(S.H[a]);
  for (int a = 0; a < 6; a++) fin = fin && isfinite(S.g[a]);
  double pmin, pmax;
  bool ok = fin && ndtm::lu_solve6_rhs(S.H, rhs, x, pmin, pmax);                 // (the pivots depend on H only: `ok` is the same on all seven lanes)
  const double c2 = (ok && lane < 6) ? ndtm::norm2_6(x) : 0.0;                   // column `lane` of H^-1
  const double r2 = lane < 6 ? ndtm::norm2_6(S.H + 6 * lane) : 0.0;              // row `lane` of H
  double hF2 = 0, invF2 = 0;
  for (int k = 0; k < 6; k++) { invF2 += __shfl(c2, k); hF2 += __shfl(r2, k); }
  if (lane != 6) return;
  ok = ok && ndtm::lu_accept(hF2, invF2);
  if (!ok) ndtm::svd_solve6(S.H, rhs, x);
  for (int a = 0; a < 6; a++) sol[a] = x[a];
  __threadfence_block();
  sol[6] = 1.0;
}

// The body of the while loop of computeTransformation (impl2:131-183) with computeStepLengthMT (impl2:841-1003), for one pair
// whose reduced (score, g, H) are in S.  Called by every lane of one wave; lane 0 carries the state, the others only help where
// the same function is needed on several arguments at once (the SE(3) exponentials of the re-basing step: SIMT runs them for
// the price of one).  Returns on lane 0: NEWTON_DONE (pair finalised), NEWTON_SWEEP (a step was scheduled: the pair takes part
// in the next derivative sweep), NEWTON_HESSIAN (live More-Thuente case: waiting for the computeHessian pass).
// mt = 0: step_size > eps/2, the More-Thuente loop is dead (every shipped configuration);
// mt = 1: live case, called after a derivative sweep;  mt = 2: live case, called after the computeHessian pass.
// The re-basing step of impl2:163-166 on its own: pn = log(exp(delta_p) exp(p)), inc = float(exp(delta_p)) column-major, delta_p = dir * a_t.
// Called by every lane of a wave (the two exponentials run side by side on lanes 0 and 1, as inside newton_update); results on lane 0.
__device__ __forceinline__ void newton_rebase(const double p[6], const double dir[6], const double a_t, double pn[6], float inc_cm[16]) {
  const int lane = threadIdx.x & 63;
  double in[6];
  for (int a = 0; a < 6; a++) in[a] = (lane & 1) ? p[a] : dir[a] * a_t;          // impl2:156
  const ndtm::SE3 e = ndtm::se3_exp(in);
  const ndtm::SE3 e_dp = shfl_se3(e, 0), e_p = shfl_se3(e, 1);
  double R[9];
  ndtm::q_to_matrix(e_dp.q, R);                                                  // set_increment_se3
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) inc_cm[c * 4 + r] = (float)R[r * 3 + c];
    inc_cm[12 + r] = (float)e_dp.t[r];
    inc_cm[r * 4 + 3] = 0.f;
  }
  inc_cm[15] = 1.f;
  ndtm::se3_log(ndtm::se3_mul(e_dp, e_p), pn);                                   // impl2:166
}

__device__ __forceinline__ int newton_update(PairState& S, mi355ndt_result* res, double step_max, double eps, int max_iterations, int mt,
                                             volatile double* sol = nullptr /* non-null: the solve comes from newton_solve_side */,
                                             const bool rebased = false /* S.reb_pn / S.reb_inc hold the re-basing of this step already */) {
  const int lane = threadIdx.x & 63;
  // exp(delta_p) and exp(p) of impl2:163-166, side by side on two lanes (same bits as one after the other on one lane)
  ndtm::SE3 e_dp, e_p;
  const bool pre = (mt == 0) && (S.phase == PH_STEP) && !rebased;                // wave-uniform: nothing has been written yet
  if (pre) {
    double in[6];
    for (int a = 0; a < 6; a++) in[a] = (lane & 1) ? S.p[a] : S.dir[a] * S.a_t;   // impl2:156
    const ndtm::SE3 e = ndtm::se3_exp(in);
    e_dp = shfl_se3(e, 0);
    e_p = shfl_se3(e, 1);
  }
  if (lane != 0) return NEWTON_DONE;

  const double step_min = eps / 2;
  if (mt == 1 && S.phase == PH_STEP) {                                           // impl2:920-1000
    const int its = mt_loop(S, step_max, step_min);
    S.mt_loops += its;
    S.sweeps += its;                                                             // computeDerivatives calls the reference makes
    if (its) {
      bool fin = S.a_t == S.a_t;
      if (!fin) {                                                                // NaN trial value: NaN pose, nothing is hit
        for (int a = 0; a < 6; a++) S.xt[a] = S.p[a] + S.dir[a] * S.a_t;
        ndtm::pose_to_f32(S.xt, S.T, S.Rj);
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 4; c++) S.final_cm[c * 4 + r] = S.T[r * 4 + c];
          S.final_cm[r * 4 + 3] = 0.f;
        }
        S.final_cm[15] = 1.f;
        S.score = 0; S.hits = 0;
        for (int a = 0; a < 6; a++) S.g[a] = 0;
        for (int a = 0; a < 36; a++) S.H[a] = 0;                                 // computeHessian over a NaN cloud
      } else {
        S.phase = PH_HESS;                                                       // impl2:999-1000: H comes from computeHessian
        return NEWTON_HESSIAN;
      }
    }
  }
  if (S.phase == PH_HESS) S.phase = PH_STEP;
  if (S.phase == PH_STEP) {
    double pn[6];
    if (rebased) {                                                               // computed after the previous update published its sweep (same operations)
      for (int a = 0; a < 16; a++) S.inc_cm[a] = S.reb_inc[a];
      for (int a = 0; a < 6; a++) pn[a] = S.reb_pn[a];
    } else if (pre) {
      set_increment_se3(S, e_dp);                                                // impl2:163
      ndtm::se3_log(ndtm::se3_mul(e_dp, e_p), pn);                               // impl2:166
    } else {
      double dp[6];
      for (int a = 0; a < 6; a++) dp[a] = S.dir[a] * S.a_t;                      // impl2:156
      const ndtm::SE3 ed = ndtm::se3_exp(dp);
      set_increment_se3(S, ed);
      ndtm::se3_log(ndtm::se3_mul(ed, ndtm::se3_exp(S.p)), pn);
    }
    for (int a = 0; a < 6; a++) S.p[a] = pn[a];
    const bool conv = (S.it > max_iterations) || (S.it && (fabs(S.a_t) < eps));  // impl2:175-179
    S.it++;
    if (conv) { finalize_pair(S, res, 1); return NEWTON_DONE; }
  }
  for (int guard = 0; guard < 4; guard++) {
    for (int a = 0; a < 16; a++) S.prev_inc_cm[a] = S.inc_cm[a];                 // impl2:134
    double d[6];
    if (sol) {                                                                   // impl2:138-140, computed by the block's second wave meanwhile
      while (sol[6] == 0.0) __builtin_amdgcn_s_sleep(1);
      __threadfence_block();
      for (int a = 0; a < 6; a++) d[a] = sol[a];
    } else {
      newton_solve(S, d);
    }
    double nrm = 0;
    for (int a = 0; a < 6; a++) nrm += d[a] * d[a];
    nrm = sqrt(nrm);
    if (nrm == 0 || nrm != nrm) { finalize_pair(S, res, nrm == nrm); return NEWTON_DONE; }   // impl2:147-152
    for (int a = 0; a < 6; a++) d[a] /= nrm;                                     // impl2:154
    double dphi0 = 0;
    for (int a = 0; a < 6; a++) dphi0 += S.g[a] * d[a];
    dphi0 = -dphi0;                                                              // impl2:849
    if (dphi0 >= 0 && dphi0 == 0) {
      // impl2:856-857: step length 0, nothing re-evaluated
      double z[6] = {0, 0, 0, 0, 0, 0}, pn[6];
      set_increment(S, z);
      ndtm::se3_log(ndtm::se3_mul(ndtm::se3_exp(z), ndtm::se3_exp(S.p)), pn);
      for (int a = 0; a < 6; a++) S.p[a] = pn[a];
      const bool conv = (S.it > max_iterations) || (S.it && (0.0 < eps));
      S.it++;
      if (conv) { finalize_pair(S, res, 1); return NEWTON_DONE; }
      continue;
    }
    if (dphi0 >= 0) { for (int a = 0; a < 6; a++) d[a] = -d[a]; }                // impl2:861-862
    double a_t = nrm;
    a_t = a_t < step_max ? a_t : step_max;                                       // impl2:890-892
    a_t = a_t > step_min ? a_t : step_min;
    double xt[6];
    for (int a = 0; a < 6; a++) { S.dir[a] = d[a]; xt[a] = S.p[a] + d[a] * a_t; S.xt[a] = xt[a]; }   // impl2:894
    S.a_t = a_t;
    S.phi0 = -S.score;                                                           // impl2:846
    S.dphi0 = dphi0 >= 0 ? -dphi0 : dphi0;                                       // impl2:849, 860
    ndtm::pose_to_f32(xt, S.T, S.Rj);                                            // impl2:900
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 4; c++) S.final_cm[c * 4 + r] = S.T[r * 4 + c];
      S.final_cm[r * 4 + 3] = 0.f;
    }
    S.final_cm[15] = 1.f;
    S.phase = PH_STEP;
    S.sweeps++;
    return NEWTON_SWEEP;
  }
  finalize_pair(S, res, 1);
  return NEWTON_DONE;
}

// One block (UPD_WAVES waves) per pair: fixed-order reduction of the pair's partial rows (reduce_pair_rows), then wave 0 runs the
// Newton control (newton_update).  `rows_per_pair` = stored rows per pair (the row stride), `pts_per_chunk` = points covered by one
// chunk = four consecutive rows (CHUNK_PTS in batch mode) or, with `chunk_rows`, by one stored row (latency mode).
NDT_KERNEL void __launch_bounds__(UPD_THREADS, 2)
k_update(PairState* st, const double* __restrict__ partials, int rows_per_pair, int pts_per_chunk, int chunk_rows, mi355ndt_result* results,
         int* active_counter, int* active_list, SweepCtl* ctl, unsigned long long* hits_total,
         double step_max, double eps, int max_iterations, int reduce_only, int mt) {
  __shared__ double sm[UPD_WAVES][NACC];
  __shared__ double sol[SOL_WORDS];
  // the pair's state in LDS for the duration of the update (the Newton step touches some sixty of its fields one after the other; through a
  // global reference every first touch of a line is a memory round trip of its own)
  __shared__ PairState Ssh;
  const int b = blockIdx.x;
  static_assert(sizeof(PairState) % 8 == 0, "PairState travels as 8-byte words");
  constexpr int NW = (int)(sizeof(PairState) / 8);
  unsigned long long* sg = reinterpret_cast<unsigned long long*>(&st[b]);
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(&Ssh);
  for (int i = threadIdx.x; i < NW; i += UPD_THREADS) sl[i] = sg[i];
  if (threadIdx.x == 0) sol[6] = 0.0;
  __syncthreads();
  PairState& S = Ssh;
  if (S.phase == PH_DONE) return;                                                // (block-uniform)
  if (mt == 2 && S.phase != PH_HESS) return;                                     // only pairs whose Hessian pass just ran
  const int lane = threadIdx.x & 63;
  const int nchunks = (S.n_src + pts_per_chunk - 1) / pts_per_chunk;
  const bool take = (mt != 2 || (lane >= 7 && lane < 43));                       // the Hessian pass fills H only
  const double v = reduce_pair_rows(partials + (size_t)b * rows_per_pair * NACC, nchunks, take, sm, chunk_rows != 0);
  if (threadIdx.x < NACC && take) {
    if (lane == 0) S.score = v;
    else if (lane < 7) S.g[lane - 1] = v;
    else if (lane < 43) S.H[lane - 7] = v;
    else { S.hits = (long long)v; if (hits_total) atomicAdd(hits_total, (unsigned long long)v); }
  }
  __syncthreads();                                                               // lane 0 reads what lanes 0..43 just stored
  if (threadIdx.x >= 128) return;
  if (threadIdx.x >= 64) { if (mt == 0 && !reduce_only) newton_solve_side(S, sol); return; }
  if (!reduce_only) {
    // (latency mode: the re-basing of p for this step was computed under the sweep, by its extra workgroup -- ndt_sweep.hpp; the tag says so)
    const bool rebased = mt == 0 && S.phase == PH_STEP && S.reb_tag == (long long)S.sweeps;
    const int rc = newton_update(S, &results[b], step_max, eps, max_iterations, mt, mt == 0 ? sol : nullptr, rebased);
    if (lane == 0 && rc == NEWTON_SWEEP) {
      atomicAdd(active_counter, 1);
      active_list[atomicAdd(&ctl->n_active, 1)] = b;                             // this pair takes part in the next sweep
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < NW; i += 64) sg[i] = sl[i];                             // the state goes back
}

// output cloud of align(): source moved by final_transformation_ (f32), written as packed x,y,z triples (what goes back over PCIe)
NDT_KERNEL void k_transform(const float* __restrict__ src, size_t pitch, const PairState* __restrict__ st, int b, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* X = src + (size_t)b * 3 * pitch;
  const float* F = st[b].final_cm;
  float px = X[i], py = X[pitch + i], pz = X[2 * pitch + i];
  for (int a = 0; a < 3; a++) out[(size_t)3 * i + a] = ((F[0 * 4 + a] * px + F[1 * 4 + a] * py) + F[2 * 4 + a] * pz) + F[3 * 4 + a];
}

// host clouds arrive as packed x,y,z triples (the engine drops the other fields of the caller's records while it stages them in
// pinned memory); this turns one cloud into the SoA rows the kernels read, zero-filling the padding up to the row pitch
NDT_KERNEL void __launch_bounds__(256) k_deinterleave(const float* __restrict__ xyz, int n, float* rows, size_t pitch) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pitch) return;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < (size_t)n) { x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2]; }
  rows[i] = x; rows[pitch + i] = y; rows[2 * pitch + i] = z;
}


// ... and several clouds of one transfer at once (grid.y = cloud): mi355_ndt.hip, upload_items
struct DeintTab { int cnt, pad_; struct { unsigned long long src_off; float* rows; unsigned long long pitch; int n, pad_; } e[16]; };
NDT_KERNEL void __launch_bounds__(256) k_deinterleave_multi(const float* __restrict__ packed, const DeintTab tab) {
  const int c = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pitch = tab.e[c].pitch;
  if (i >= pitch) return;
  const float* xyz = packed + tab.e[c].src_off;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < (size_t)tab.e[c].n) { x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2]; }
  float* rows = tab.e[c].rows;
  rows[i] = x; rows[pitch + i] = y; rows[2 * pitch + i] = z;
}

// Pose records for the multi-GPU gather: 96 bytes = {float final[16] column-major; float score; int iterations;
// int converged; int pair_id; int pad[4]} per pair, packed on the device straight from the results of the last batch align.
// Rows past the batch (a rank that owns one pair fewer than its neighbours) carry pair_id = -1.
struct PoseRecord { float final_cm[16]; float score; int iterations, converged, pair_id, pad[4]; };
NDT_KERNEL void k_pose_records(const mi355ndt_result* __restrict__ res, int n_pairs, int id_base, int id_stride, PoseRecord* out, int capacity) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= capacity) return;
  PoseRecord r;
  memset(&r, 0, sizeof r);
  r.pair_id = -1;
  if (k < n_pairs) {
    for (int a = 0; a < 16; a++) r.final_cm[a] = res[k].final_colmajor[a];
    r.score = (float)res[k].score;
    r.iterations = res[k].iterations;
    r.converged = res[k].converged;
    r.pair_id = id_base + k * id_stride;
  }
  out[k] = r;
}
