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

