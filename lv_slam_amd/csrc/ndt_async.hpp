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
// A wave that has waited for its own ring's ticket for a while serves already-published positions of OTHER rings meanwhile (wait_ticket in
// k_align_async): no ring depends on the workgroups of one XCD being resident -- two engines launching at once can split the XCDs between them.
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
#define ASYNC_MAX_CTX    4       // batch contexts one launch can serve (the synchronous batch align: one; the stream mode: 2..4)
#define ASYNC_MAX_CARRY  128     // pairs one launch may hand over to the next (stream mode)
#define ASYNC_CTX_SHIFT  24      // ticket word = context << 24 | pair slot of that context; -1 = no ticket yet

// One batch context: everything a work item or an update of one of ITS pairs touches.  The synchronous mi355ndt_batch_align has one;
// the stream mode (mi355ndt_stream_*) keeps several batches resident and a launch serves the pairs of all of them.
struct AsyncCtx {
  const float* src; size_t pitch;
  PairState* st; const GridDesc* gd; const BitWord* words; const VoxelRec* recs; const float* cent;
  double* partials; const int* src_cnt; unsigned* arrived; mi355ndt_result* results;
  unsigned* n_done;             // (may be null) pairs of this context finalised so far, over all launches: the host's "batch complete" test
  int must_finish;              // the context's buffers are recycled after this launch: its pairs are never handed over
  // (may be null) the batch's 96-byte pose records for the multi-GPU gather, written by the updater that finalises a pair:
  // record b carries pair_id = pose_base + b * pose_stride -- packed on the device with no kernel of its own
  int pose_base; PoseRecord* pose; int pose_stride, pad_;
};
struct AsyncTab { AsyncCtx c[ASYNC_MAX_CTX]; };
// A pointer that comes out of memory is a FLAT pointer to the compiler: flat loads are slower than global ones, also count against the LDS
// counter, and cost the DIRECT7 launch 4-8 % against round 4 on the same box when the table first replaced the kernel arguments (a cast
// through address space 1 and back is folded away; an empty asm between the two casts keeps it but makes the pointer opaque, which was worse
// still: tools/ab_libs.sh).  So the device reads the table through a mirror of AsyncCtx whose fields ARE global pointers -- same layout,
// the address space is part of the type -- and hands them on as ordinary pointers the compiler knows the origin of.
#define NDT_GP(T) __attribute__((address_space(1))) T*
struct AsyncCtxG {
  NDT_GP(const float) src; size_t pitch;
  NDT_GP(PairState) st; NDT_GP(const GridDesc) gd; NDT_GP(const BitWord) words; NDT_GP(const VoxelRec) recs; NDT_GP(const float) cent;
  NDT_GP(double) partials; NDT_GP(const int) src_cnt; NDT_GP(unsigned) arrived; NDT_GP(mi355ndt_result) results;
  NDT_GP(unsigned) n_done;
  int must_finish;
  int pose_base; NDT_GP(PoseRecord) pose; int pose_stride, pad_;
};
static_assert(sizeof(AsyncCtxG) == sizeof(AsyncCtx) && offsetof(AsyncCtxG, pose) == offsetof(AsyncCtx, pose) && offsetof(AsyncCtxG, n_done) == offsetof(AsyncCtx, n_done),
              "AsyncCtxG mirrors AsyncCtx");
__device__ __forceinline__ AsyncCtx async_ctx_load(const AsyncTab* __restrict__ tab, const int ci) {
  const AsyncCtxG& g = reinterpret_cast<const AsyncCtxG*>(tab->c)[ci];
  AsyncCtx c;
  c.src = (const float*)g.src; c.pitch = g.pitch; c.st = (PairState*)g.st; c.gd = (const GridDesc*)g.gd; c.words = (const BitWord*)g.words;
  c.recs = (const VoxelRec*)g.recs; c.cent = (const float*)g.cent; c.partials = (double*)g.partials; c.src_cnt = (const int*)g.src_cnt;
  c.arrived = (unsigned*)g.arrived; c.results = (mi355ndt_result*)g.results; c.n_done = (unsigned*)g.n_done; c.must_finish = g.must_finish;
  c.pose_base = g.pose_base; c.pose = (PoseRecord*)g.pose; c.pose_stride = g.pose_stride; c.pad_ = 0;
  return c;
}
// stream mode, per context, in device memory: what the host wants to know about a context's batch after every launch
struct CtxStat { unsigned done, total_words, max_cells, plan_exceeded; };   // pairs finalised; the last planned build's sizes and verdict (k_build_check)
// ... and per launch, in MAPPED host memory (a ring of slots): written by k_stream_status behind the launch, `seq` last -- no copy, no event
struct StreamStatus { unsigned fin, abort_, n_live, susp; CtxStat ctx[ASYNC_MAX_CTX]; unsigned seq; unsigned pad0_;
                      unsigned long long build_t0, build_t1;   // wall_clock64 (100 MHz) at the start of the launch's batch's build (k_stream_inputs) and at its end (k_async_prepare)
                      unsigned pad_[6]; };
static_assert(sizeof(StreamStatus) == 128, "one status slot is 128 bytes");

struct AsyncCtl {
  unsigned pub;                 // tickets published so far (ticket numbers are handed out by fetch-add)
  unsigned fin;                 // pairs that left the launch -- finalised, or suspended for the next launch; n_live = the launch is over
  unsigned abort_;              // a wave gave up waiting (bounded spins): the host falls back to the round-based align
  unsigned n_live;              // pairs that entered the launch (k_async_prepare: carried over + new)
  unsigned susp;                // pairs suspended by this launch = entries of `carry`
  unsigned pad_[27];
  unsigned pos[8 * ASYNC_POS_STRIDE];   // per ring: positions of its item stream handed out so far
  unsigned carry[ASYNC_MAX_CARRY];      // ticket words of the suspended pairs: the next launch's first tickets
};

#ifndef ASYNC_CLAIM
#define ASYNC_CLAIM(K) ((K) == 1 ? 2 : 1)
#endif
#ifndef ASYNC_HELP_RINGS
#define ASYNC_HELP_RINGS 1       // 0 (experiments only): waves serve their own ring alone, as in round 4 -- tests/test_stream_gpu.py's ring-mask test then gives up
#endif
#define ASYNC_HELP_AFTER 64u         // polls (~1 us each) a wave waits for its own ticket before it looks at other rings: the help is for a launch
                                     // that would otherwise not finish, and costs a launch that needs none 0.3-1.4 % when it starts at once (measured)
#define ASYNC_SPIN_LIMIT (1u << 18)      // polls of ~1-2.5 us: a wave that has seen no ticket for ~0.3-0.6 s ends the launch (the caller re-runs the batch in rounds).  Legitimate
                                         // waits are the tail of a launch -- a claimed position at most ~16 tickets ahead of the last pair's: ~1 ms --, two to three orders of
                                         // magnitude below; round 5's 2^23 left a starved launch spinning for ~10 s before it fell back

// Everything the launch reads before it has written it, set by ONE kernel on the stream in front of it (never inside the launch): the
// context table, the NEW pairs' initial states (k_init_state's job) and arrival counters, the first tickets -- the pairs the previous
// launch suspended (`prev->carry`, stream mode), then the new pairs in slot order; ticket g sits in ring g & 7 / slot g >> 3, "no ticket
// yet" everywhere else -- and the control words.  `prev` is a different block than `ctl` (the stream mode alternates two).
NDT_KERNEL void k_async_prepare(const AsyncTab tab, AsyncTab* tab_dev, const int new_ci, const int n_new, PairState* st, const float* __restrict__ guess_cm,
                                const int* __restrict__ src_cnt, const GridDesc* __restrict__ gd, unsigned* arrived, int* active_list, SweepCtl* sweep_ctl /* two of them */,
                                int* ring, const int ring_cap, AsyncCtl* ctl, const AsyncCtl* prev, unsigned* done_new /* may be null */,
                                PoseRecord* pose_new /* may be null */, const int pose_cap, unsigned long long* stamp /* may be null */) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && stamp) *stamp = (unsigned long long)wall_clock64();     // (stream mode, profiling: the end of the build in front of this kernel)
  const unsigned nc = prev ? min(prev->susp, (unsigned)ASYNC_MAX_CARRY) : 0u;
  const unsigned n_live = nc + (unsigned)n_new;
  if (i < (size_t)8 * ring_cap) {
    const size_t g = (i % (size_t)ring_cap) * 8 + i / (size_t)ring_cap;
    int w = -1;
    if (g < nc) w = (int)prev->carry[g];
    else if (g < n_live) w = (new_ci << ASYNC_CTX_SHIFT) | (int)(g - nc);
    ring[i] = w;
  }
  if (i < (size_t)n_new * ASYNC_ARR_STRIDE) arrived[i] = 0u;
  if (i < offsetof(AsyncCtl, carry) / sizeof(unsigned)) reinterpret_cast<unsigned*>(ctl)[i] = (i == 0 || i == 3) ? n_live : 0u;   // pub = n_live = tickets out
  if (sweep_ctl && i < 2 * sizeof(SweepCtl) / sizeof(int)) reinterpret_cast<int*>(sweep_ctl)[i] = i == 0 ? n_new : 0;           // n_active of the first
  if (i == 0 && done_new && n_new > 0) *done_new = 0u;
  if (pose_new && i < (size_t)pose_cap) {          // every row starts as a padding row (pair_id = -1); a pair's updater fills its own
    PoseRecord r;
    memset(&r, 0, sizeof r);
    r.pair_id = -1;
    pose_new[i] = r;
  }
  if (i == 0) tab_dev->c[0] = tab.c[0];
  if (i == 1) tab_dev->c[1] = tab.c[1];
  if (i == 2) tab_dev->c[2] = tab.c[2];
  if (i == 3) tab_dev->c[3] = tab.c[3];
  static_assert(ASYNC_MAX_CTX == 4, "the table is copied entry by entry");
  if (i < (size_t)n_new) {
    if (active_list) active_list[i] = (int)i;
    init_pair_state(st[i], guess_cm + i * 16, src_cnt[i], gd[i].status);
  }
}

// The pair's rows -> (score, g, H, hits), Newton step, publication.  Called by every lane of ONE wave; `Ssh` / `sol` are that wave's LDS.
// `tw` = the pair's ticket word (what is published again, or handed over); `stop_thresh` > 0 (stream mode): once no more than that many pairs
// are still in the launch, a pair that wants another sweep is SUSPENDED instead -- its state is complete in memory, its ticket word goes
// into ctl->carry, and the next launch's prepare kernel turns it into one of that launch's first tickets.  The count only ever falls, so
// at most `stop_thresh` pairs are suspended; which ones is a matter of timing, what they compute is not (a pair's bits depend on the pair alone).
__device__ __forceinline__ void async_update(const AsyncCtx& C, const int b, const int tw, const int n_pts, const int rows_per_pair, PairState& Ssh, volatile double* sol,
                                             int* ring, const int ring_cap, AsyncCtl* ctl, const unsigned n_live, const int stop_thresh, unsigned long long* hits_total,
                                             const double step_max, const double eps, const int max_iterations
#ifdef NDT_TIMELINE
                                             , unsigned long long* tl, unsigned long long& tl_last
#endif
                                             ) {
  PairState* st = C.st;
  const double* partials = C.partials;
  mi355ndt_result* results = C.results;
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
    bool suspend = false;
    if (rc == NEWTON_SWEEP && stop_thresh > 0 && !C.must_finish)
      suspend = n_live - __hip_atomic_load((const gu32*)&ctl->fin, RLX_AGENT) <= (unsigned)stop_thresh;
    if (rc == NEWTON_SWEEP && !suspend) {
      const unsigned g = __hip_atomic_fetch_add((gu32*)&ctl->pub, 1u, RLX_AGENT);
      __hip_atomic_store((gu32*)reinterpret_cast<unsigned*>(ring + (size_t)(g & 7u) * ring_cap + (g >> 3)), (unsigned)tw, RLX_AGENT);
    } else if (rc == NEWTON_SWEEP) {
      const unsigned k = __hip_atomic_fetch_add((gu32*)&ctl->susp, 1u, RLX_AGENT);
      if (k < (unsigned)ASYNC_MAX_CARRY) __hip_atomic_store((gu32*)&ctl->carry[k], (unsigned)tw, RLX_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add((gu32*)&ctl->fin, 1u, RLX_AGENT);
    } else {
      if (C.pose) {
        PoseRecord r;
        for (int a = 0; a < 16; a++) r.final_cm[a] = Ssh.final_cm[a];
        r.score = (float)Ssh.score; r.iterations = Ssh.it; r.converged = Ssh.converged; r.pair_id = C.pose_base + b * C.pose_stride;
        for (int a = 0; a < 4; a++) r.pad[a] = 0;
        C.pose[b] = r;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (finalize_pair's result record, the pose record)
      if (C.n_done) __hip_atomic_fetch_add((gu32*)C.n_done, 1u, RLX_AGENT);
      __hip_atomic_fetch_add((gu32*)&ctl->fin, 1u, RLX_AGENT);
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

// Stream mode: a batch's small inputs (point counts of both sides, guesses: ~20 KB) are READ by one workgroup from the mapped host block the
// submit filled, and the build's word block (totals, extremes) is cleared by the same kernel -- instead of a host-to-device copy and a fill,
// each of which costs the stream ~12-16 us of idle time around it on top of its own (rocprofv3 timeline of a streamed run).
// It is also the first kernel of the batch's target build: with profiling on it stamps the build's start into the launch's status slot (mapped
// host memory) and k_async_prepare, the first kernel behind the build, stamps its end -- the stream's builds are timed without events (an event
// record between two kernels costs the stream ~10 us of idle time).
NDT_KERNEL void k_stream_inputs(const unsigned* __restrict__ host_in, unsigned* __restrict__ dev_in, const unsigned n_in, unsigned* __restrict__ clear, const unsigned n_clear,
                                unsigned long long* stamp) {
  if (threadIdx.x == 0 && stamp) *stamp = (unsigned long long)wall_clock64();
  for (unsigned i = threadIdx.x; i < n_in; i += blockDim.x) dev_in[i] = host_in[i];
  for (unsigned i = threadIdx.x; i < n_clear; i += blockDim.x) clear[i] = 0u;
}

// Stream mode: behind every persistent launch, one wave reports to the host through mapped memory -- how the launch ended and where every
// context's batch stands -- and then posts the launch's sequence number (posted PCIe writes; the host polls the number: no copy, no event).
NDT_KERNEL void k_stream_status(const AsyncCtl* __restrict__ ctl, const CtxStat* __restrict__ stat, volatile unsigned* host_slot, const unsigned seq) {
  const int i = threadIdx.x;
  static_assert(offsetof(AsyncCtl, fin) == 4 && offsetof(AsyncCtl, susp) == 16 && offsetof(StreamStatus, ctx) == 16 && offsetof(StreamStatus, seq) == 80 && offsetof(StreamStatus, build_t0) == 88, "status layout");
  if (i < 4) host_slot[i] = reinterpret_cast<const unsigned*>(ctl)[1 + i];                      // fin, abort_, n_live, susp
  else if (i < 4 + 4 * ASYNC_MAX_CTX) host_slot[i] = reinterpret_cast<const unsigned*>(stat)[i - 4];
  __threadfence_system();
  __syncthreads();
  if (i == 0) host_slot[20] = seq;
}

// What a work item touches -- points, grid, bitmap, records, rows, counts, arrival counters -- reaches the kernel as KERNEL ARGUMENTS, one set
// per context, selected by the ticket's context index; the table in memory keeps what only an update needs (results, completion counter,
// pose records, must_finish).  Measured on one box (tools/ab_libs.sh, DIRECT7 / config 3, one launch): pointers read from the table -- flat,
// cast through address space 1, or typed global -- 4,070-4,220 us; the same pointers as kernel arguments 3,890 us = round 4's 3,880: the
// compiler knows a kernel argument's address space, that it is wave-uniform and (for the restrict-qualified ones) what it cannot alias.
#define NDT_CTX_PARAMS(i) const float* __restrict__ src##i, size_t pitch##i, PairState* st##i, const GridDesc* __restrict__ gd##i, const BitWord* __restrict__ words##i, \
                          const VoxelRec* __restrict__ recs##i, double* partials##i, const int* __restrict__ src_cnt##i, unsigned* arrived##i, const float* __restrict__ cent##i
#define NDT_CTX_SEL(f, ci) ((ci) == 0 ? f##0 : (ci) == 1 ? f##1 : (ci) == 2 ? f##2 : f##3)
template <bool PCA, int K, int ORD>
__global__ void __launch_bounds__(SWEEP_THREADS, (SweepTune<PCA, K, ORD>::WPE))
k_align_async(const AsyncTab* __restrict__ tab, int items_per_pair, int* ring, int ring_cap, AsyncCtl* ctl, SweepConst sc, unsigned long long* hits_total,
              double step_max, double eps, int max_iterations, int stop_thresh, unsigned debug_abort_pos, unsigned debug_ring_mask, int claim_items,
              NDT_CTX_PARAMS(0), NDT_CTX_PARAMS(1), NDT_CTX_PARAMS(2), NDT_CTX_PARAMS(3)) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ double exp_tab[64];
  __shared__ PairState Ssh[WAVES];
  __shared__ double sol[WAVES][SOL_WORDS];
  if (threadIdx.x < 64) exp_tab[threadIdx.x] = ndtm::c_exp2_64[threadIdx.x];
  __syncthreads();                                 // (the only block barrier: the four waves are independent from here on)
  const int x = blockIdx.x & 7;                    // the ring this workgroup serves
  if (!((debug_ring_mask >> x) & 1u)) return;      // (test hook, MI355NDT_OPT_DEBUG_ASYNC_RINGS: an XCD that holds no workgroup of this launch)
  const gu32* ringx = (const gu32*)reinterpret_cast<const unsigned*>(ring + (size_t)x * ring_cap);
  const gu32* fin_p = (const gu32*)&ctl->fin;
  gu32* pos_p = (gu32*)&ctl->pos[x * ASYNC_POS_STRIDE];
  const unsigned n_live = ctl->n_live;             // (written by k_async_prepare, the kernel in front of this one; constant during the launch)
  const int I = items_per_pair;
  // DIRECT1 items are short (one probe per point, ~0.85 hits): two consecutive items of a pair per claim / arrival halve the hand-overs
  // ... and so do two DIRECT7 items per claim where the launch hands its tail on (stream mode: +1.4-2 % measured; a launch that runs its own
  // tail loses with the coarser positions -- config 5, synchronous: -11 % --, so the host decides per launch: `claim_items`, DIRECT1: always two)
#ifndef FAST_CLAIM1
#define FAST_CLAIM1 4            // tolerance arithmetic, DIRECT1: a claim = one chunk (four items) going through the lane = point pipeline as one stream of tiles
#endif
  constexpr int CLAIM_D1 = (ORD == 2 && FAST_D1_POINT) ? FAST_CLAIM1 : ASYNC_CLAIM(1);
  const int CLAIM = K == 1 ? CLAIM_D1 : (claim_items == 2 ? 2 : 1);
  const int Iu = I / CLAIM;                        // positions per ticket (items_per_pair is a multiple of four)
#ifdef NDT_TIMELINE
  unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl_last = __builtin_readcyclecounter();
#endif
  auto give_up = [&]() {                           // never hang: end the launch for everybody and tell the host (which takes the round-based path)
    __hip_atomic_store((gu32*)&ctl->abort_, 1u, RLX_AGENT);
    __hip_atomic_store((gu32*)&ctl->fin, n_live, RLX_AGENT);
  };
  // Wait (one lane, relaxed polls with sleeps) until the ticket of the claimed position `p` of this ring exists; -2: the launch is over.
  // A wave that waits HELPS THE OTHER RINGS meanwhile: tickets go round the eight rings (ticket g -> ring g & 7), a ring is served by the
  // workgroups of one XCD (blockIdx & 7), and nothing guarantees that every XCD holds workgroups of this launch -- two engines launching at
  // the same time can end up with one launch resident on half of the XCDs and the other on the other half, each waiting for tickets that sit
  // in rings nobody of its own serves, for ever (observed: `test_two_engines_on_two_threads_share_the_gpu`, one launch in ~10^3 gave up after
  // its poll budget).  So: once its own ticket has not come for ASYNC_HELP_AFTER polls, the wave looks at another ring y every fourth poll; if y has a position whose ticket is
  // already published (position counter / Iu < tickets published into y, from the global ticket counter `pub`), it takes exactly that position
  // with a compare-and-swap -- never a blind claim: a claimed position cannot be abandoned, and its own claim `p` stays pending --, serves it,
  // and comes back for `p` (`held`).  Every published position is thus servable by ANY resident wave of the launch; which wave serves an
  // item never enters a result (rows depend on the pair, the chunk and the input order).
  bool held = false;                               // an own claim whose ticket was not there when the wave went to help
  bool helped = false;                             // the last wait ended with a position of another ring
  unsigned held_pos = 0;
  const gu64* pubfin_p = (const gu64*)reinterpret_cast<const unsigned long long*>(ctl);   // {pub, fin}: the first two words of the control block
  static_assert(offsetof(AsyncCtl, pub) == 0 && offsetof(AsyncCtl, fin) == 4, "pub and fin are polled as one 8-byte word");
  auto wait_ticket = [&](unsigned& p, int have) -> int {
    int w = have;
    unsigned sp = p;
    int st = 0;
    if (lane == 0) {
      unsigned spins = helped ? ASYNC_HELP_AFTER : 0u;   // (a wave that came back from helping goes on helping at once)
      const int t = (int)(p / (unsigned)Iu);
      while (w < 0) {
        const unsigned long long pf = __hip_atomic_load(pubfin_p, RLX_AGENT);        // pub (low word) and fin with one poll of the line
        if ((unsigned)(pf >> 32) >= n_live) { w = -2; break; }
        if (ASYNC_HELP_RINGS && spins >= ASYNC_HELP_AFTER && (spins & 3u) == 3u) {   // a servable position of another ring?
          const unsigned y = ((unsigned)x + 1u + (spins >> 2) % 7u) & 7u;
          gu32* pos_y = (gu32*)&ctl->pos[y * ASYNC_POS_STRIDE];
          const unsigned q = __hip_atomic_load(pos_y, RLX_AGENT);
          const unsigned pubv = (unsigned)pf;
          const unsigned slots_y = pubv > y ? ((pubv - 1u - y) >> 3) + 1u : 0u;     // tickets g < pubv with g & 7 == y
          const unsigned tq = q / (unsigned)Iu;
          if (tq < slots_y && (int)tq < ring_cap) {
            unsigned expect = q;
            if (__hip_atomic_compare_exchange_strong(pos_y, &expect, q + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              const gu32* slot = (const gu32*)reinterpret_cast<const unsigned*>(ring + (size_t)y * ring_cap + tq);
              int ws = (int)__hip_atomic_load(slot, RLX_AGENT);
              unsigned lag = 0;                    // (the counter runs one store ahead of the ticket word)
              while (ws < 0 && ++lag <= ASYNC_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(2); ws = (int)__hip_atomic_load(slot, RLX_AGENT); }
              if (ws < 0) { give_up(); w = -2; break; }
              w = ws; sp = q; st = 1;
              break;
            }
          }
        }
        __builtin_amdgcn_s_sleep(32);
        if (t < ring_cap) w = (int)__hip_atomic_load(ringx + t, RLX_AGENT);
        if (++spins > ASYNC_SPIN_LIMIT) { give_up(); w = -2; break; }
      }
    }
    w = __builtin_amdgcn_readfirstlane(w);
    helped = __builtin_amdgcn_readfirstlane(st) != 0;
    if (helped) { held = true; held_pos = p; p = __builtin_amdgcn_readfirstlane(sp); }
    return w;
  };
  // Positions of the ring's item stream (ticket 0's items, ticket 1's, ...) are CLAIMED, one returning fetch-add per item.  Static dealing
  // (wave w takes positions w, w + W, ...) was built and measured first: every update makes its wave late for good, a ticket completes when
  // its latest wave does, and with nothing to rebalance them the waves spent 38 % of the launch waiting for tickets (docs/experiments.md 10c).
  unsigned pos = 0;
  if (lane == 0) pos = __hip_atomic_fetch_add(pos_p, 1u, RLX_AGENT);
  pos = __builtin_amdgcn_readfirstlane(pos);
  int tw = -1;                                     // ticket word: context << 24 | pair slot
  if (lane == 0 && (int)(pos / (unsigned)Iu) < ring_cap) tw = (int)__hip_atomic_load(ringx + pos / (unsigned)Iu, RLX_AGENT);
  tw = wait_ticket(pos, tw);
  if (tw < 0) return;
  unsigned pose_w = sweep_pose_words(NDT_CTX_SEL(st, tw >> ASYNC_CTX_SHIFT) + (tw & ((1 << ASYNC_CTX_SHIFT) - 1)));
#pragma unroll 1
  for (;;) {
    // (test hook, MI355NDT_OPT_DEBUG_ASYNC_ABORT: the wave that claimed this position of ring 0 gives up as a wave whose ticket never came would)
    if (x == 0 && pos == debug_abort_pos) { if (lane == 0) give_up(); break; }
    const int rem = (int)(pos % (unsigned)Iu) * CLAIM;
    TL_STAMP(10);                                  // hand-over: claim, ticket, (update)
    const int ci = tw >> ASYNC_CTX_SHIFT;
    AsyncCtx C = async_ctx_load(tab, ci);            // (wave-uniform scalar loads of a table nobody writes during the launch: the update's fields)
    C.src = NDT_CTX_SEL(src, ci); C.pitch = NDT_CTX_SEL(pitch, ci); C.st = NDT_CTX_SEL(st, ci); C.gd = NDT_CTX_SEL(gd, ci); C.words = NDT_CTX_SEL(words, ci);
    C.recs = NDT_CTX_SEL(recs, ci); C.partials = NDT_CTX_SEL(partials, ci); C.src_cnt = NDT_CTX_SEL(src_cnt, ci); C.arrived = NDT_CTX_SEL(arrived, ci);
    C.cent = NDT_CTX_SEL(cent, ci);
    const int b = tw & ((1 << ASYNC_CTX_SHIFT) - 1);
    const int n_b = C.src_cnt[b];
#ifndef ASYNC_D1_PIPE
#define ASYNC_D1_PIPE 1
#endif
    if constexpr (K == 1 && ASYNC_D1_PIPE && ORD != 2) {   // DIRECT1 (exact arithmetic): the claim's items as one software pipeline (ndt_sweep.hpp), same rows bit for bit
      sweep_rows_d1<PCA, ORD, CLAIM_D1>(b, rem, C.src, C.pitch, C.gd, C.words, C.recs, C.partials, I, sc, exp_tab, pose_w, n_b, b
#ifdef NDT_TIMELINE
                                     , tl, tl_last
#endif
                                     );
    } else if constexpr (K == 1 && ORD == 2 && FAST_D1_POINT) {   // DIRECT1, tolerance arithmetic: the claim's items as one stream of tiles, lane = point
      float T[12], Rj[9];
#pragma unroll
      for (int a_ = 0; a_ < 12; a_++) T[a_] = __uint_as_float(__builtin_amdgcn_readlane(pose_w, a_));
#pragma unroll
      for (int a_ = 0; a_ < 9; a_++) Rj[a_] = __uint_as_float(__builtin_amdgcn_readlane(pose_w, 12 + a_));
      sweep_rows_d1p<PCA, CLAIM_D1, true>(b, rem, C.src, C.pitch, T, Rj, n_b, C.gd[b], C.words, C.recs, C.partials, I, sc
#ifdef NDT_TIMELINE
                                          , tl, tl_last
#endif
                                          );
    } else {
#pragma unroll 1
      for (int k = 0; k < CLAIM; k++)
        sweep_item<PCA, K, 8, false, ORD, true>(b, rem + k, C.src, C.pitch, C.st, C.gd, C.words, C.recs, C.partials, I, sc, C.cent, nullptr, exp_tab, pose_w, n_b, b
#ifdef NDT_TIMELINE
                                                , tl, tl_last
#endif
                                                );
    }
    // Three memory round trips between two items, each carrying everything that does not depend on the next one:
    //  1. the row stores drain (the row is complete in memory before the arrival that may hand it to an updater) -- and the claim of the
    //     next position, which depends on nothing, returns with them;
    unsigned npos = held_pos;                      // (a claim still pending from before the wave went to help another ring comes first)
    if (lane == 0 && !held) npos = __hip_atomic_fetch_add(pos_p, 1u, RLX_AGENT);
    held = false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    npos = __builtin_amdgcn_readfirstlane(npos);
    const int tn = (int)(npos / (unsigned)Iu);
    //  2. the arrival, and the ticket word of the next position;
    unsigned old = 0;
    int ntw = -1;
    if (lane == 0) {
      old = __hip_atomic_fetch_add((gu32*)(C.arrived + (size_t)b * ASYNC_ARR_STRIDE), (unsigned)CLAIM, RLX_AGENT);
      if (tn < ring_cap) ntw = (int)__hip_atomic_load(ringx + tn, RLX_AGENT);
    }
    old = __builtin_amdgcn_readfirstlane(old);
    ntw = __builtin_amdgcn_readfirstlane(ntw);
    //  3. the next pair's pose -- in flight while this wave updates (if it has to), then together with the next item's point loads.
    unsigned npose = 0;
    if (ntw >= 0) npose = sweep_pose_words(NDT_CTX_SEL(st, ntw >> ASYNC_CTX_SHIFT) + (ntw & ((1 << ASYNC_CTX_SHIFT) - 1)));
    TL_STAMP(8);                                   // row drain + claim, arrival + ticket
    if ((old + (unsigned)CLAIM) % (unsigned)I == 0u) {   // this was the sweep's last item: this wave is the pair's updater
      async_update(C, b, tw, n_b, I, Ssh[wv], sol[wv], ring, ring_cap, ctl, n_live, stop_thresh, hits_total, step_max, eps, max_iterations
#ifdef NDT_TIMELINE
                   , tl, tl_last
#endif
                   );
      TL_STAMP(15);                                // the deferred re-basing (off the pair's critical path)
    }
    if (ntw < 0) {                                 // the next position's ticket does not exist yet
      ntw = wait_ticket(npos, ntw);                 // (may come back with a position of another ring: `held`)
      if (ntw < 0) break;
      npose = sweep_pose_words(NDT_CTX_SEL(st, ntw >> ASYNC_CTX_SHIFT) + (ntw & ((1 << ASYNC_CTX_SHIFT) - 1)));
    }
    tw = ntw; pos = npos; pose_w = npose;
  }
#ifdef NDT_TIMELINE
  if (lane == 0) for (int k = 0; k < 16; k++) atomicAdd(&g_tl[k], tl[k]);
#endif
}
