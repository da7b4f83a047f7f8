// ndt_sequence.hpp -- latency mode: a whole run of frames tracked on the device.
//
// Restates the per-frame body of ScanMatchingOdomNodelet::matching_s2k
// (src/lidar_odometry/scan_matching_odom_nodelet.cpp:192-261): scan-to-keyframe align, the frame-1 double align (:223-227),
// tf_s2s / odom_velo (:231-234), the keyframe test (:237-240: |t|, 2 acos(q.w) in f32, stamp difference), target switch (:241-247)
// and the constant-velocity guess (:249-250) -- as the tail of the Newton-update kernel, so that no host round trip separates
// one frame's last Newton step from the next frame's first sweep.  The host only pumps an alternating stream of
// (k_seq_update, k_sweep<FINE>) launches and stops when the device says the last frame is done (mi355_ndt.hip:
// mi355ndt_sequence_run); how many rounds a frame needs is decided here, on the device.
//
// The frames' voxel grids are built beforehand, all at once, by the batched target build (every frame is a potential keyframe:
// 2.6 us per grid in a 271-frame batch); `grid_of[k]` names the grid frame k is matched against (its keyframe's).
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_update.hpp"

struct SeqState {
  int    n_frames, cur, key_id, second_done, done, aligns, launches, cur_n;   // cur_n: point count of frame `cur` (the update needs it before it has the frame's state)
  double pre_tf_s2k[16], key_pose[16];          // 4x4 f64 row-major
  double keyframe_stamp;
  double d_trans, d_angle, d_time;              // keyframe_delta_trans / _angle / _time (:67-76)
};

// ---- 4x4 f64 helpers, explicit operation order (canonical choices; Eigen's own order is not observable here) ---------------
namespace seqm {
__device__ __host__ inline void mul4(const double A[16], const double B[16], double C[16]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      C[i * 4 + j] = ((A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j]) + A[i * 4 + 2] * B[2 * 4 + j]) + A[i * 4 + 3] * B[3 * 4 + j];
}
// general 4x4 inverse by cofactors (the shape of Eigen 3.3's compute_inverse_size4: cofactor matrix, then one division by
// col(0) . cofactor row)
__device__ __host__ inline double det3h(const double* m, int i1, int i2, int i3, int j1, int j2, int j3) {
  return m[i1 * 4 + j1] * (m[i2 * 4 + j2] * m[i3 * 4 + j3] - m[i2 * 4 + j3] * m[i3 * 4 + j2]);
}
__device__ __host__ inline double cof4(const double* m, int i, int j) {
  const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4, j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
  return (det3h(m, i1, i2, i3, j1, j2, j3) + det3h(m, i2, i3, i1, j1, j2, j3)) + det3h(m, i3, i1, i2, j1, j2, j3);
}
__device__ __host__ inline void inv4(const double M[16], double R[16]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      const double c = cof4(M, i, j);
      R[j * 4 + i] = ((i + j) & 1) ? -c : c;
    }
  const double det = ((M[0 * 4 + 0] * R[0 * 4 + 0] + M[1 * 4 + 0] * R[0 * 4 + 1]) + M[2 * 4 + 0] * R[0 * 4 + 2]) + M[3 * 4 + 0] * R[0 * 4 + 3];
  for (int a = 0; a < 16; a++) R[a] = R[a] / det;
}
// w of Eigen::Quaternionf(R.cast<float>()) (quaternionbase_assign_impl<Matrix3f>), f32 arithmetic
__device__ __host__ inline float quat_w_f32(const float m[9]) {
  float t = (m[0] + m[4]) + m[8];
  if (t > 0.f) return 0.5f * sqrtf(t + 1.0f);
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > m[i * 3 + i]) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  t = sqrtf(((m[i * 3 + i] - m[j * 3 + j]) - m[k * 3 + k]) + 1.0f);
  return (m[k * 3 + j] - m[j * 3 + k]) * (0.5f / t);
}
}  // namespace seqm

// frame 0 (:194-208): it becomes the first keyframe; frame 1's first align starts from Identity + 1.5 m along x
__global__ void k_seq_begin(SeqState* seq, PairState* st, const GridDesc* __restrict__ gd, const int* __restrict__ cnt,
                            const double* __restrict__ stamps, mi355ndt_seq_frame* out, int* active_list, SweepCtl* ctl, int* grid_of,
                            volatile int* host_flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  SeqState& Q = *seq;
  Q.cur = 1; Q.key_id = 0; Q.second_done = 0; Q.aligns = 0; Q.done = 0; Q.launches = 0; Q.cur_n = Q.n_frames > 1 ? cnt[1] : 0;
  for (int a = 0; a < 16; a++) Q.pre_tf_s2k[a] = Q.key_pose[a] = (a % 5 == 0) ? 1.0 : 0.0;
  Q.keyframe_stamp = stamps[0];
  mi355ndt_seq_frame f;
  memset(&f, 0, sizeof f);
  for (int a = 0; a < 16; a++) { f.odom_colmajor[a] = (a % 5 == 0) ? 1.0 : 0.0; f.tf_s2k_colmajor[a] = (a % 5 == 0) ? 1.f : 0.f; }
  f.key_id = 0; f.new_keyframe = 1; f.converged = 1;
  out[0] = f;
  grid_of[0] = 0;
  if (Q.n_frames < 2) { Q.done = 1; host_flags[0] = 1; return; }
  float G[16];
  for (int a = 0; a < 16; a++) G[a] = (a % 5 == 0) ? 1.f : 0.f;
  G[12] = 1.5f;                                                                  // guess_trans(0,3) = 1.5 (:199-200)
  grid_of[1] = 0;
  init_pair_state(st[1], G, cnt[1], gd[0].status);
  active_list[0] = 1;
  ctl->n_active = 1;
}

// The call-site policy after the align of frame `cur` ended (scan_matching_odom_nodelet.cpp:221-250), shared by the pumped update kernel and
// the persistent one.  S = the frame's pair state, Sn = where the NEXT frame's initial state goes (k_seq_update: st[cur + 1] itself; the
// persistent kernel: an LDS copy it writes through afterwards).  Returns SEQ_SAME (frame 1's second align: S was re-initialised, sweep it
// again), SEQ_NEXT (Sn holds frame cur + 1's state, matched against grid Q.key_id) or SEQ_END (that was the last frame).
enum { SEQ_SAME = 0, SEQ_NEXT = 1, SEQ_END = 2 };
__device__ inline int seq_policy(SeqState& Q, PairState& S, PairState& Sn, const int cur, const int cur_grid, const GridDesc* __restrict__ gd,
                                 const int* __restrict__ cnt, const double* __restrict__ stamps, mi355ndt_seq_frame* out, int* grid_of) {
  Q.aligns++;
  const float* F = S.final_cm;                                                   // getFinalTransformation(), column-major f32
  if (cur == 1 && !Q.second_done) {                                              // :223-227: frame 1 is aligned twice, the second time from the first result
    Q.second_done = 1;
    float G[16];
    for (int a = 0; a < 16; a++) G[a] = F[a];                                    // tf_s2k.cast<float>() of a float matrix cast to double: the same floats
    init_pair_state(S, G, cnt[cur], gd[cur_grid].status);
    return SEQ_SAME;
  }
  double tf[16], inv[16], s2s[16], odom[16];
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) tf[r * 4 + c] = (double)F[c * 4 + r];
  seqm::inv4(Q.pre_tf_s2k, inv);
  seqm::mul4(inv, tf, s2s);                                                      // tf_s2s = pre_tf_s2k.inverse() * tf_s2k (:231)
  seqm::mul4(Q.key_pose, tf, odom);                                              // odom_velo = key_pose * tf_s2k (:234)
  const double dx = sqrt(tf[3] * tf[3] + (tf[7] * tf[7] + tf[11] * tf[11]));      // :237 (Eigen 3.3's unrolled 3-vector redux: x0^2 + (x1^2 + x2^2))
  float Rf[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rf[r * 3 + c] = F[c * 4 + r];
  const double da = (double)(2.f * acosf(seqm::quat_w_f32(Rf)));                 // :238: std::acos(float) -> float, times int
  const double dt = stamps[cur] - Q.keyframe_stamp;                              // :239
  mi355ndt_seq_frame f;
  memset(&f, 0, sizeof f);
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) f.odom_colmajor[c * 4 + r] = odom[r * 4 + c];
  for (int a = 0; a < 16; a++) f.tf_s2k_colmajor[a] = F[a];
  f.key_id = Q.key_id;
  f.iterations = S.it; f.converged = S.converged; f.trans_probability = S.trans_probability;
  f.dx = dx; f.da = da; f.dt = dt;
  f.aligns = (cur == 1) ? 2 : 1;
  if (dx > Q.d_trans || da > Q.d_angle || dt > Q.d_time) {                       // :240-247: this scan becomes the keyframe
    Q.key_id = cur;
    for (int a = 0; a < 16; a++) { tf[a] = (a % 5 == 0) ? 1.0 : 0.0; Q.key_pose[a] = odom[a]; }
    Q.keyframe_stamp = stamps[cur];
    f.new_keyframe = 1;
  }
  out[cur] = f;
  double guess[16];
  for (int a = 0; a < 16; a++) Q.pre_tf_s2k[a] = tf[a];                          // :249
  seqm::mul4(tf, s2s, guess);                                                    // guess_trans = pre_tf_s2k * tf_s2s (:250)
  const int nxt = cur + 1;
  Q.cur = nxt;
  if (nxt >= Q.n_frames) { Q.done = 1; return SEQ_END; }
  float G[16];
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) G[c * 4 + r] = (float)guess[r * 4 + c];   // guess_trans.cast<float>() (:221)
  grid_of[nxt] = Q.key_id;
  Q.cur_n = cnt[nxt];
  init_pair_state(Sn, G, cnt[nxt], gd[Q.key_id].status);
  return SEQ_NEXT;
}

// One launch = "the next thing the run needs after a sweep": reduce the current frame's rows, take its Newton step; when its align
// is over, run the call-site policy and set the next frame up.  One block; the host launches it blindly between sweeps.
__global__ void __launch_bounds__(UPD_THREADS)
k_seq_update(SeqState* seq, PairState* st, const double* __restrict__ partials, int rows_per_pair, int pts_per_chunk, mi355ndt_result* results,
             const GridDesc* __restrict__ gd, const int* __restrict__ cnt, const double* __restrict__ stamps, mi355ndt_seq_frame* out,
             int* active_list, SweepCtl* ctl, int* grid_of, volatile int* host_flags,
             double step_max, double eps, int max_iterations) {
  __shared__ double sm[UPD_WAVES][NACC];
  __shared__ double sol[SOL_WORDS];
  // The frame's pair state lives in LDS for the duration of the update: the Newton step reads and writes some sixty of its fields one after the
  // other, and through a global reference every first touch of a line was a memory round trip of its own (rounds 3-4: 11.5 us per update,
  // most of it such waits).  Two round trips are left: the run's position (cur, its point count), then the state together with the rows.
  __shared__ PairState Ssh;
  SeqState& Q = *seq;
#ifdef NDT_TIMELINE
  unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (slots 8..14: this kernel's phases, wave 0 / wave 1; tools/seq_run.py)
  unsigned long long tl_last = __builtin_readcyclecounter();
#endif
  if (threadIdx.x == 0) sol[6] = 0.0;
#if defined(NDT_TIMELINE) && defined(SEQ_UPDATE_REPEAT)
  // ---- analysis build only (tools/seq_run.py, docs/experiments.md 10e): "what would the update cost if its code were resident?"  The body up to
  // the Newton step runs SEQ_UPDATE_REPEAT times on the same inputs -- state and rows re-read with agent-scope loads (L2, as a resident updater would
  // have to), nothing written back in between -- and only the LAST pass is stamped: same instructions, instruction cache warm.
  {
    static_assert(sizeof(PairState) % 8 == 0, "PairState travels as 8-byte words");
    constexpr int NWr = (int)(sizeof(PairState) / 8);
    if (Q.done == 0) {
      const int cur_r = Q.cur, n_r = Q.cur_n;
      const int nch = (n_r + pts_per_chunk - 1) / pts_per_chunk;
      const gu64* sgr = (const gu64*)reinterpret_cast<const unsigned long long*>(&st[cur_r]);
      unsigned long long* slr = reinterpret_cast<unsigned long long*>(&Ssh);
      for (int rep = 0; rep < SEQ_UPDATE_REPEAT - 1; rep++) {
        for (int i = threadIdx.x; i < NWr; i += UPD_THREADS) slr[i] = __hip_atomic_load(sgr + i, RLX_AGENT);
        const double vr = reduce_pair_rows<true>(partials + (size_t)cur_r * rows_per_pair * NACC, nch, true, sm, true);
        const int ln = threadIdx.x & 63;
        if (threadIdx.x < NACC) {
          if (ln == 0) Ssh.score = vr;
          else if (ln < 7) Ssh.g[ln - 1] = vr;
          else if (ln < 43) Ssh.H[ln - 7] = vr;
          else Ssh.hits = (long long)vr;
        }
        __syncthreads();
        if (threadIdx.x >= 64 && threadIdx.x < 128) newton_solve_side(Ssh, sol);
        if (threadIdx.x < 64) {
          const bool reb = Ssh.phase == PH_STEP && Ssh.reb_tag == (long long)Ssh.sweeps;
          (void)newton_update(Ssh, &results[cur_r], step_max, eps, max_iterations, 0, sol, reb);
        }
        __syncthreads();
        if (threadIdx.x == 0) sol[6] = 0.0;
        __syncthreads();
      }
    }
    tl_last = __builtin_readcyclecounter();
  }
#endif
  const int done = Q.done, cur = Q.cur, n_src = Q.cur_n;                         // (block-uniform)
  if (threadIdx.x == 0) { const int l = Q.launches + 1; Q.launches = l; host_flags[1] = l; }   // launches executed (the host bounds its queue depth with it)
  if (done) return;                                                              // (the pump's overshoot)
  TL_STAMP(8);                                                                   // run position
  static_assert(sizeof(PairState) % 8 == 0, "PairState travels as 8-byte words");
  constexpr int NW = (int)(sizeof(PairState) / 8);
  unsigned long long* sg = reinterpret_cast<unsigned long long*>(&st[cur]);
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(&Ssh);
  for (int i = threadIdx.x; i < NW; i += UPD_THREADS) sl[i] = sg[i];
  const int lane = threadIdx.x & 63;
  const int nchunks = (n_src + pts_per_chunk - 1) / pts_per_chunk;
  const double v = reduce_pair_rows(partials + (size_t)cur * rows_per_pair * NACC, nchunks, true, sm, true);   // latency mode: chunk rows (its barrier also publishes Ssh)
  if (threadIdx.x < NACC) {
    if (lane == 0) Ssh.score = v;
    else if (lane < 7) Ssh.g[lane - 1] = v;
    else if (lane < 43) Ssh.H[lane - 7] = v;
    else Ssh.hits = (long long)v;
  }
  __syncthreads();
  TL_STAMP(9);                                                                   // state + rows
  if (threadIdx.x >= 128) return;
  if (threadIdx.x >= 64) {                                                       // the solve, next to wave 0
    newton_solve_side(Ssh, sol);
#ifdef NDT_TIMELINE
    TL_STAMP(13);
    if (lane == 0) atomicAdd(&g_tl[13], tl[13]);
#endif
    return;
  }
  // (the re-basing of p for this step was computed under the sweep, by its extra workgroup: ndt_sweep.hpp)
  const bool rebased = Ssh.phase == PH_STEP && Ssh.reb_tag == (long long)Ssh.sweeps;
  const int rc = newton_update(Ssh, &results[cur], step_max, eps, max_iterations, 0, sol, rebased);
  TL_STAMP(10);                                                                  // Newton step (incl. the wait for the solve)
  if (lane == 0) {
    if (rc == NEWTON_SWEEP) { active_list[0] = cur; ctl->n_active = 1; }
    else {
      // ---- the align of frame `cur` is over: scan_matching_odom_nodelet.cpp:221-250
      const int kind = seq_policy(Q, Ssh, st[cur + 1 < Q.n_frames ? cur + 1 : cur], cur, grid_of[cur], gd, cnt, stamps, out, grid_of);
      if (kind == SEQ_END) { __threadfence_system(); host_flags[0] = 1; }
      else { active_list[0] = kind == SEQ_SAME ? cur : cur + 1; ctl->n_active = 1; }
    }
  }
  // the state goes back (frame cur's: updated, finalised, or re-initialised for frame 1's second align)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  TL_STAMP(11);                                                                  // policy (when the frame's align ended)
  for (int i = lane; i < NW; i += 64) sg[i] = sl[i];
#ifdef NDT_TIMELINE
  TL_STAMP(12);
  tl[14] = 1;
  if (lane == 0) for (int k = 8; k < 15; k++) if (k != 13) atomicAdd(&g_tl[k], tl[k]);
#endif
}
