// ndt_async.hpp -- ONE launch per batch align: every pair runs its own Newton loop to its own end, as every reference align() does
// (include/ndt_omp/ndt_omp_impl2.hpp:131-183: one while loop per computeTransformation), instead of the batch advancing in lockstep
// rounds of (k_update, k_sweep) launches that each wait for the slowest pair.
//
// Persistent waves (the sweep's grid: 2 workgroups of 4 waves per CU).  Work arrives as TICKETS: one ticket = one derivative sweep of
// one pair = `items_per_pair` work items (sweep_item, ndt_sweep.hpp -- the very code of the lockstep kernel, so a pair's partial rows,
// and with them every result bit, are the same).  Tickets are numbered globally in publication order; ticket g lives in slot g >> 3 of
// ring g & 7, and ring x is served by the waves of the workgroups with blockIdx.x % 8 == x (observed to be XCD x: a sweep's points,
// bitmap and records then stay in one L2 -- speed only, nothing below depends on it).  Within a ring the positions of the item stream
// (ticket 0's items, ticket 1's, ...) are claimed with one returning fetch-add per item, issued at the END of the previous item together
// with that item's arrival -- behind the item's loads, never in front of them (a returning atomic at the head of a wave's in-order memory
// queue delays every load behind it); a wave whose position falls into a ticket that does not exist yet waits for it.
// The wave that retires a pair's last item (one returning agent-scope fetch-add per item on a per-pair counter) becomes that pair's UPDATER: it adds the pair's rows with k_update's fixed tree, runs the Newton step
// (newton_update, ndt_update.hpp -- same code, same bits) and either publishes the pair's next ticket or finalises the pair.
// Publication is round-robin over the rings, so the rings stay balanced to one ticket however the iteration counts are distributed.
//
// Visibility inside one launch (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by another CU's
// stores): every word that crosses workgroups -- partial rows, PairState, ring slots, counters -- is written AND read with agent-scope
// relaxed atomics (8-byte / 4-byte sc1 accesses: write-through stores, L1-bypassing loads), every storing wave drains its stores
// (s_waitcnt vmcnt(0), inline asm so that the compiler cannot drop it) before the word that announces them, and no flag is ever a plain
// store.  No cache-wide fence anywhere: tools/fence_cost.hip priced those at 12-28 us per work item.
// Everything polled is zeroed / poisoned by the host before every launch (hipMemsetAsync on the stream).
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_sweep.hpp"
#include "ndt_update.hpp"

#define ASYNC_POS_STRIDE 32      // unsigned words between the rings' position counters (one 128-B line each: eight hot atomics, eight lines)
#define ASYNC_ARR_STRIDE 16      // ... between the pairs' arrival counters (64 B: pairs in flight together do not share an atomic's line)
struct AsyncCtl {
  unsigned pub;                 // tickets published so far (ticket numbers are handed out by fetch-add)
  unsigned done;                // pairs finalised; n_pairs = the launch is over
  unsigned abort_;              // a wave gave up waiting (bounded spins): the host reports an error
  unsigned pad_[29];
  unsigned pos[8 * ASYNC_POS_STRIDE];   // per ring: positions of its item stream handed out so far
};

#ifndef ASYNC_CLAIM
#define ASYNC_CLAIM(K) ((K) == 1 ? 2 : 1)
#endif
#define ASYNC_SPIN_LIMIT (1u << 23)      // polls of ~1 us: a device that stopped making progress ends the launch after seconds, not never

// Everything the launch reads before it has written it, set by ONE kernel on the stream in front of it (never inside the launch, never by
// a previous launch): the pairs' initial states (k_init_state's job), ticket g = pair g of the first sweeps in ring g & 7 / slot g >> 3 and
// "no ticket yet" everywhere else, the arrival counters, the control words.  (Six small launches -- two kernels and four fills -- until the
// end of round 4: ~25 us per align.)
NDT_KERNEL void k_async_prepare(PairState* st, const float* __restrict__ guess_cm, const int* __restrict__ src_cnt, const GridDesc* __restrict__ gd,
                                int n_pairs, int* active_list, SweepCtl* sweep_ctl /* two of them */, int* ring, int ring_cap, unsigned* arrived,
                                AsyncCtl* ctl) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)8 * ring_cap) {
    const size_t g = (i % (size_t)ring_cap) * 8 + i / (size_t)ring_cap;
    ring[i] = g < (size_t)n_pairs ? (int)g : -1;
  }
  if (i < (size_t)n_pairs * ASYNC_ARR_STRIDE) arrived[i] = 0u;
  if (i < sizeof(AsyncCtl) / sizeof(unsigned)) reinterpret_cast<unsigned*>(ctl)[i] = i == 0 ? (unsigned)n_pairs : 0u;   // pub = n_pairs
  if (i < 2 * sizeof(SweepCtl) / sizeof(int)) reinterpret_cast<int*>(sweep_ctl)[i] = i == 0 ? n_pairs : 0;              // n_active of the first
  if (i < (size_t)n_pairs) {
    active_list[i] = (int)i;
    init_pair_state(st[i], guess_cm + i * 16, src_cnt[i], gd[i].status);
  }
}

// The pair's rows -> (score, g, H, hits), Newton step, publication.  Called by every lane of ONE wave; `Ssh` / `sol` are that wave's LDS.
__device__ __forceinline__ void async_update(const int b, const int n_pts, PairState* st, const double* partials, const int rows_per_pair, PairState& Ssh, volatile double* sol,
                                             mi355ndt_result* results, int* ring, const int ring_cap, AsyncCtl* ctl, unsigned long long* hits_total,
                                             const double step_max, const double eps, const int max_iterations
#ifdef NDT_TIMELINE
                                             , unsigned long long* tl, unsigned long long& tl_last
#endif
                                             ) {
  const int lane = threadIdx.x & 63;
  // the update is a chain of dependent instructions on the critical path of its pair (and of the whole batch once few pairs are left):
  // it goes first at the SIMD's issue arbiter while it runs next to a wave that streams independent evaluation work
  __builtin_amdgcn_s_setprio(3);
  static_assert(sizeof(PairState) % 8 == 0, "PairState travels as 8-byte words");
  constexpr int NW = (int)(sizeof(PairState) / 8);
  gu64* sg = (gu64*)reinterpret_cast<unsigned long long*>(&st[b]);
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(&Ssh);
  // the pair's state (written by its previous updater: another wave, maybe another XCD -> LDS copy of this wave) and the first rows of the
  // reduction are requested together: the row addresses need only the constant point count
  for (int i = lane; i < NW; i += 64) sl[i] = __hip_atomic_load(sg + i, RLX_AGENT);
  if (lane == 0) sol[6] = 0.0;
  // reduce_pair_rows' tree (ndt_update.hpp) on one wave: chunk = ((r0 + r1) + r2) + r3; group = 8 chunks in order; group k belongs to
  // "wave" k % 4, whose groups add up in ascending order; the four sums add up in order.  Same operands, same order: same bits.
  // Two groups (64 rows) are in flight per lane at a time: the rows come from memory (write-through), ~2 us per dependent batch.
  const int nchunks = (n_pts + CHUNK_PTS - 1) / CHUNK_PTS;
  double aw0 = 0.0, aw1 = 0.0, aw2 = 0.0, aw3 = 0.0;
  if (lane < NACC) {
    const gu64* P = (const gu64*)reinterpret_cast<const unsigned long long*>(partials + (size_t)b * rows_per_pair * NACC + lane);
    auto ldq = [&](int c, int k) -> double { return __longlong_as_double((long long)__hip_atomic_load(P + ((size_t)c * 4 + k) * NACC, RLX_AGENT)); };
    auto addw = [&](int g, double gs) { const int w = g & 3; if (w == 0) aw0 += gs; else if (w == 1) aw1 += gs; else if (w == 2) aw2 += gs; else aw3 += gs; };
#pragma unroll 1
    for (int c0 = 0; c0 < nchunks; c0 += 16) {
      if (c0 + 16 <= nchunks) {
        double q[16][4];
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
          for (int k = 0; k < 4; k++) q[u][k] = ldq(c0 + u, k);
        double gs = 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) gs += ((q[u][0] + q[u][1]) + q[u][2]) + q[u][3];
        addw(c0 >> 3, gs);
        gs = 0.0;
#pragma unroll
        for (int u = 8; u < 16; u++) gs += ((q[u][0] + q[u][1]) + q[u][2]) + q[u][3];
        addw((c0 >> 3) + 1, gs);
      } else {
#pragma unroll 1
        for (int g0 = c0; g0 < nchunks; g0 += 8) {
          double gs = 0.0;
          if (g0 + 8 <= nchunks) {
            double q[8][4];
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
              for (int k = 0; k < 4; k++) q[u][k] = ldq(g0 + u, k);
#pragma unroll
            for (int u = 0; u < 8; u++) gs += ((q[u][0] + q[u][1]) + q[u][2]) + q[u][3];
          } else {
            for (int c = g0; c < nchunks; c++) {
              double r[4];
#pragma unroll
              for (int k = 0; k < 4; k++) r[k] = ldq(c, k);
              gs += ((r[0] + r[1]) + r[2]) + r[3];
            }
          }
          addw(g0 >> 3, gs);
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (lane < NACC) {
    double v = 0.0;
    v += aw0; v += aw1; v += aw2; v += aw3;
    if (lane == 0) Ssh.score = v;
    else if (lane < 7) Ssh.g[lane - 1] = v;
    else if (lane < 43) Ssh.H[lane - 7] = v;
    else { Ssh.hits = (long long)v; if (hits_total) atomicAdd(hits_total, (unsigned long long)v); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  TL_STAMP(12);                                    // state + rows
  // The re-basing of p for this step was computed by the previous updater after it published this sweep (below).  Its words are fetched
  // AFTER its tag has been seen (the state above was one multi-word load: a new tag does not vouch for the words that came with it), and the
  // fetch rides under the solve.
  constexpr int RB0 = (int)(offsetof(PairState, reb_pn) / 8), RBT = (int)(offsetof(PairState, reb_tag) / 8);
  static_assert(RBT - RB0 == 14, "6 + 8 words of re-basing");
  const bool want_reb = Ssh.phase == PH_STEP;
  bool reb_ok = want_reb && Ssh.reb_tag == (long long)Ssh.sweeps;
  unsigned long long rbw = 0;
  if (reb_ok && lane < 14) rbw = __hip_atomic_load(sg + RB0 + lane, RLX_AGENT);
  newton_solve_side(Ssh, sol);                     // lanes 0..6: impl2:138-140 (same functions and operands as k_update's second wave)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  TL_STAMP(13);                                    // solve
  if (want_reb && !reb_ok) {                       // rare: the tag had not landed yet -- wait for it (bounded), then fetch the words
    unsigned spins = 0;
    while (!reb_ok && spins++ < (1u << 16)) {
      __builtin_amdgcn_s_sleep(2);
      long long tg = 0;
      if (lane == 0) tg = (long long)__hip_atomic_load(sg + RBT, RLX_AGENT);
      reb_ok = __shfl(tg, 0) == (long long)Ssh.sweeps;
    }
    if (reb_ok && lane < 14) rbw = __hip_atomic_load(sg + RB0 + lane, RLX_AGENT);
  }
  if (reb_ok && lane < 14) sl[RB0 + lane] = rbw;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const bool rebased = reb_ok;                     // (a tag that never came: newton_update computes the re-basing itself, same bits)
  int rc = newton_update(Ssh, &results[b], step_max, eps, max_iterations, 0, sol, rebased);
  rc = __builtin_amdgcn_readfirstlane(rc);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  TL_STAMP(14);                                    // Newton step
  // the state goes back write-through; it is complete in memory before the ticket that lets other waves read it exists
  for (int i = lane; i < NW; i += 64) __hip_atomic_store(sg + i, sl[i], RLX_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    if (rc == NEWTON_SWEEP) {
      const unsigned g = __hip_atomic_fetch_add((gu32*)&ctl->pub, 1u, RLX_AGENT);
      __hip_atomic_store((gu32*)reinterpret_cast<unsigned*>(ring + (size_t)(g & 7u) * ring_cap + (g >> 3)), (unsigned)b, RLX_AGENT);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (finalize_pair's result record)
      __hip_atomic_fetch_add((gu32*)&ctl->done, 1u, RLX_AGENT);
    }
  }
  TL_STAMP(9);
  if (rc == NEWTON_SWEEP) {
    // Off the pair's critical path -- its next sweep is already running --: the re-basing the NEXT update starts with (impl2:163-166) depends
    // only on what this update decided (p, dir, a_t).  Written behind the state, announced by a tag = the sweep count.
    constexpr int RBI = (int)(offsetof(PairState, reb_inc) / 8);
    double pn[6]; float inc[16];
    newton_rebase(Ssh.p, Ssh.dir, Ssh.a_t, pn, inc);
    if (lane == 0) {
      for (int a = 0; a < 6; a++) __hip_atomic_store(sg + RB0 + a, (unsigned long long)__double_as_longlong(pn[a]), RLX_AGENT);
      for (int a = 0; a < 8; a++)
        __hip_atomic_store(sg + RBI + a, (unsigned long long)__float_as_uint(inc[2 * a]) | ((unsigned long long)__float_as_uint(inc[2 * a + 1]) << 32), RLX_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(sg + RBT, (unsigned long long)(long long)Ssh.sweeps, RLX_AGENT);
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

template <bool PCA, int K, int ORD>
__global__ void __launch_bounds__(SWEEP_THREADS, (SweepTune<PCA, K>::WPE))
k_align_async(const float* __restrict__ src, size_t pitch, PairState* st, const GridDesc* __restrict__ gd, const BitWord* __restrict__ words,
              const VoxelRec* __restrict__ recs, double* partials, int items_per_pair, int n_pairs, const int* __restrict__ src_cnt, int* ring, int ring_cap,
              AsyncCtl* ctl, unsigned* arrived, SweepConst sc, const float* __restrict__ cent, mi355ndt_result* results, unsigned long long* hits_total,
              double step_max, double eps, int max_iterations) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ double exp_tab[64];
  __shared__ PairState Ssh[WAVES];
  __shared__ double sol[WAVES][8];
  if (threadIdx.x < 64) exp_tab[threadIdx.x] = ndtm::c_exp2_64[threadIdx.x];
  __syncthreads();                                 // (the only block barrier: the four waves are independent from here on)
  const int x = blockIdx.x & 7;                    // the ring this workgroup serves
  const gu32* ringx = (const gu32*)reinterpret_cast<const unsigned*>(ring + (size_t)x * ring_cap);
  const gu32* done_p = (const gu32*)&ctl->done;
  gu32* pos_p = (gu32*)&ctl->pos[x * ASYNC_POS_STRIDE];
  const int I = items_per_pair;
  // DIRECT1 items are short (one probe per point, ~0.85 hits): two consecutive items of a pair per claim / arrival halve the hand-overs
  constexpr int CLAIM = ASYNC_CLAIM(K);
  const int Iu = I / CLAIM;                        // positions per ticket (items_per_pair is a multiple of four)
#ifdef NDT_TIMELINE
  unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl_last = __builtin_readcyclecounter();
#endif
  // wait (one lane, relaxed polls with sleeps) until ticket `t` of this ring exists; -2: the launch is over
  auto wait_ticket = [&](const int t, int have) -> int {
    int b = have;
    if (lane == 0) {
      unsigned spins = 0;
      while (b < 0) {
        if (__hip_atomic_load(done_p, RLX_AGENT) >= (unsigned)n_pairs) { b = -2; break; }
        __builtin_amdgcn_s_sleep(32);
        if (t < ring_cap) b = (int)__hip_atomic_load(ringx + t, RLX_AGENT);
        if (++spins > ASYNC_SPIN_LIMIT) {          // never hang: end the launch for everybody and tell the host
          __hip_atomic_store((gu32*)&ctl->abort_, 1u, RLX_AGENT);
          __hip_atomic_store((gu32*)&ctl->done, (unsigned)n_pairs, RLX_AGENT);
          b = -2; break;
        }
      }
    }
    return __builtin_amdgcn_readfirstlane(b);
  };
  // Positions of the ring's item stream (ticket 0's items, ticket 1's, ...) are CLAIMED, one returning fetch-add per item.  Static dealing
  // (wave w takes positions w, w + W, ...) was built and measured first: every update makes its wave late for good, a ticket completes when
  // its latest wave does, and with nothing to rebalance them the waves spent 38 % of the launch waiting for tickets (docs/experiments.md 10c).
  unsigned pos = 0;
  if (lane == 0) pos = __hip_atomic_fetch_add(pos_p, 1u, RLX_AGENT);
  pos = __builtin_amdgcn_readfirstlane(pos);
  int b = -1;
  if (lane == 0 && (int)(pos / (unsigned)Iu) < ring_cap) b = (int)__hip_atomic_load(ringx + pos / (unsigned)Iu, RLX_AGENT);
  b = wait_ticket((int)(pos / (unsigned)Iu), b);
  if (b < 0) return;
  unsigned pose_w = sweep_pose_words(st + b);
#pragma unroll 1
  for (;;) {
    const int rem = (int)(pos % (unsigned)Iu) * CLAIM;
    TL_STAMP(10);                                  // hand-over: claim, ticket, (update)
    const int n_b = src_cnt[b];
#pragma unroll 1
    for (int k = 0; k < CLAIM; k++)
      sweep_item<PCA, K, 8, false, ORD, true>(b, rem + k, src, pitch, st, gd, words, recs, partials, I, sc, cent, nullptr, exp_tab, pose_w, n_b, b
#ifdef NDT_TIMELINE
                                              , tl, tl_last
#endif
                                              );
    // Three memory round trips between two items, each carrying everything that does not depend on the next one:
    //  1. the row stores drain (the row is complete in memory before the arrival that may hand it to an updater) -- and the claim of the
    //     next position, which depends on nothing, returns with them;
    unsigned npos = 0;
    if (lane == 0) npos = __hip_atomic_fetch_add(pos_p, 1u, RLX_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    npos = __builtin_amdgcn_readfirstlane(npos);
    const int tn = (int)(npos / (unsigned)Iu);
    //  2. the arrival, and the ticket word of the next position;
    unsigned old = 0;
    int nb = -1;
    if (lane == 0) {
      old = __hip_atomic_fetch_add((gu32*)(arrived + (size_t)b * ASYNC_ARR_STRIDE), (unsigned)CLAIM, RLX_AGENT);
      if (tn < ring_cap) nb = (int)__hip_atomic_load(ringx + tn, RLX_AGENT);
    }
    old = __builtin_amdgcn_readfirstlane(old);
    nb = __builtin_amdgcn_readfirstlane(nb);
    //  3. the next pair's pose -- in flight while this wave updates (if it has to), then together with the next item's point loads.
    unsigned npose = 0;
    if (nb >= 0) npose = sweep_pose_words(st + nb);
    TL_STAMP(8);                                   // row drain + claim, arrival + ticket
    if ((old + (unsigned)CLAIM) % (unsigned)I == 0u) {   // this was the sweep's last item: this wave is the pair's updater
      async_update(b, n_b, st, partials, I, Ssh[wv], sol[wv], results, ring, ring_cap, ctl, hits_total, step_max, eps, max_iterations
#ifdef NDT_TIMELINE
                   , tl, tl_last
#endif
                   );
      TL_STAMP(15);                                // the deferred re-basing (off the pair's critical path)
    }
    if (nb < 0) {                                  // the next position's ticket does not exist yet
      nb = wait_ticket(tn, nb);
      if (nb < 0) break;
      npose = sweep_pose_words(st + nb);
    }
    b = nb; pos = npos; pose_w = npose;
  }
#ifdef NDT_TIMELINE
  if (lane == 0) for (int k = 0; k < 16; k++) atomicAdd(&g_tl[k], tl[k]);
#endif
}
