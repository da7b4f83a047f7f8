// ndt_build_runs.hpp -- the run-compressed target build (round 3).
//
// VoxelGridCovariance::applyFilter (include/ndt_omp/voxel_grid_covariance_omp_impl.hpp:209-263) adds every point to its leaf in
// INPUT order, and the leaf sums here have to be added in that order to stay bit-identical.  The point-level build (ndt_build.hpp)
// gets the order by stably sorting all 65,536 (cell, point) pairs of a target and then GATHERING the points of a leaf through the
// sorted ids -- the sort (0.22 ms per 271 targets) and the gather-bound leaf sums (0.23 ms) were two thirds of the build.
// A spinning lidar delivers its points ring by ring, so consecutive points mostly fall into the same cell: a 65,536-point scan is
// ~9,600 RUNS of consecutive points with equal cell (6.8 points per run), a searchable leaf is ~3 runs.  So:
//   k_keys_runs   cell key per point (impl:218-223) + number of run heads per block
//   k_run_offsets per target: exclusive scan of the block counts -> run index base of every block, number of runs
//   k_run_write   run records in input order: (cell key, first point); a run's length is the next run's first point minus its own
//   segment sort  of the RUN records by cell (ndt_segsort.hpp with per-segment counts): 7x fewer entries than points; stable, so
//                 the runs of a cell stay in input order
//   k_mark_runs   sorted run -> (first point, length); heads of cells with >= min_points points set the cell's bitmap bit (impl:297)
//   k_rank        (ndt_build.hpp) voxel id = rank of the cell among the searchable ones
//   k_segstart_runs  first sorted run of every searchable leaf
//   k_leafsum_runs   one wave per leaf: its runs are CONTIGUOUS pieces of the input rows -- no gather through an id array; the
//                 nine f64 sums are added strictly in input order (runs ascending, points ascending within a run)
// The point-level sort stays for what needs points grouped by cell (getFitnessScore's nearest-neighbour search) and as the
// reference build the tests compare this one against (MI355NDT_BUILD=points).
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_build.hpp"

#define RUN_BLOCK 1024                      // points per block of k_keys_runs / k_run_write (4 per thread)

__device__ __forceinline__ unsigned cell_key_of(const float* __restrict__ X, size_t pitch, size_t i, int n, const GridDesc& g, unsigned cmask) {
  unsigned cell = cmask;                          // "not binned": padding, non-finite point or unusable grid
  if ((long long)i < (long long)n && g.status == GRID_OK) {
    const float x = X[i], y = X[pitch + i], z = X[2 * pitch + i];
    if (finite3(x, y, z)) {
      const int i0 = (int)(floorf(x * g.inv_leaf) - (float)g.min_b[0]);
      const int i1 = (int)(floorf(y * g.inv_leaf) - (float)g.min_b[1]);
      const int i2 = (int)(floorf(z * g.inv_leaf) - (float)g.min_b[2]);
      cell = (unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2);
    }
  }
  return cell;
}

// cell key of every point of the row (positions past the target's count get the "not binned" key) and, per block, the number of
// positions whose key differs from their predecessor's (run heads; position 0 is one)
__global__ void __launch_bounds__(256) k_keys_runs(const float* __restrict__ tgt, size_t pitch, const int* __restrict__ cnt, const GridDesc* __restrict__ gd,
                                                   unsigned* keys, unsigned* blk_heads, int cb) {
  __shared__ unsigned sm[5];
  const int b = blockIdx.y;
  const GridDesc& g = gd[b];
  const float* X = tgt + (size_t)b * 3 * pitch;
  const unsigned cmask = (1u << cb) - 1u;
  const int n = cnt[b];
  const size_t i0 = (size_t)blockIdx.x * RUN_BLOCK + (size_t)threadIdx.x * 4;     // four consecutive positions per thread
  unsigned k[4];
#pragma unroll
  for (int u = 0; u < 4; u++) k[u] = (i0 + u < pitch) ? cell_key_of(X, pitch, i0 + u, n, g, cmask) : cmask;
  // the key before this thread's first position: the previous lane's last one, or recomputed at a wave / block boundary
  unsigned prev = __shfl_up(k[3], 1);
  if ((threadIdx.x & 63) == 0) prev = (i0 == 0) ? ~k[0] : cell_key_of(X, pitch, i0 - 1, n, g, cmask);
  unsigned heads = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (i0 + u < pitch) { keys[(size_t)b * pitch + i0 + u] = k[u]; heads += (k[u] != prev) ? 1u : 0u; }
    prev = k[u];
  }
  unsigned tot;
  (void)block_exscan<256>(heads, &tot, sm);
  if (threadIdx.x == 0) blk_heads[(size_t)b * gridDim.x + blockIdx.x] = tot;
}

// per target: exclusive scan of the per-block head counts (run index of every block's first head) and the number of runs
__global__ void __launch_bounds__(256) k_run_offsets(const unsigned* __restrict__ blk_heads, unsigned* blk_off, unsigned* run_cnt, int nblk) {
  __shared__ unsigned sm[5];
  const int b = blockIdx.x;
  unsigned base = 0;
  for (int k0 = 0; k0 < nblk; k0 += 256) {
    const int k = k0 + threadIdx.x;
    const unsigned v = k < nblk ? blk_heads[(size_t)b * nblk + k] : 0u;
    unsigned tot;
    const unsigned ex = block_exscan<256>(v, &tot, sm);
    if (k < nblk) blk_off[(size_t)b * nblk + k] = base + ex;
    base += tot;
  }
  if (threadIdx.x == 0) run_cnt[b] = base;
}

// run records in input order: key and first position of run r; run_start[R] = pitch closes the last run
__global__ void __launch_bounds__(256) k_run_write(const unsigned* __restrict__ keys, size_t pitch, const unsigned* __restrict__ blk_off,
                                                   const unsigned* __restrict__ run_cnt, unsigned* run_key, unsigned* run_id, unsigned* run_start) {
  __shared__ unsigned sm[5];
  const int b = blockIdx.y;
  const unsigned* K = keys + (size_t)b * pitch;
  const size_t i0 = (size_t)blockIdx.x * RUN_BLOCK + (size_t)threadIdx.x * 4;
  unsigned k[4];
#pragma unroll
  for (int u = 0; u < 4; u++) k[u] = (i0 + u < pitch) ? K[i0 + u] : 0u;
  unsigned prev = __shfl_up(k[3], 1);
  if ((threadIdx.x & 63) == 0) prev = (i0 == 0) ? ~k[0] : ((i0 - 1 < pitch) ? K[i0 - 1] : 0u);
  bool head[4];
  unsigned heads = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) { head[u] = (i0 + u < pitch) && (k[u] != prev); heads += head[u] ? 1u : 0u; prev = k[u]; }
  unsigned tot;
  unsigned r = blk_off[(size_t)b * gridDim.x + blockIdx.x] + block_exscan<256>(heads, &tot, sm);
  const size_t row = (size_t)b * (pitch + 1);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (!head[u]) continue;
    run_key[(size_t)b * pitch + r] = k[u];
    run_id[(size_t)b * pitch + r] = r;
    run_start[row + r] = (unsigned)(i0 + u);
    r++;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) run_start[row + run_cnt[b]] = (unsigned)pitch;
}

// sorted run j -> (first point, length); the first run of every cell whose runs hold >= min_points points sets the cell's bit
// (impl:297) and is flagged as the head of a searchable leaf.  Counting stops at min_points: at most min_points runs are read.
__global__ void __launch_bounds__(256) k_mark_runs(const unsigned* __restrict__ skey, const unsigned* __restrict__ sid, size_t pitch,
                                                   const unsigned* __restrict__ run_cnt, const unsigned* __restrict__ run_start,
                                                   const GridDesc* __restrict__ gd, BitWord* words, unsigned* sst, unsigned* slen, unsigned char* lhead,
                                                   int min_points, int cb) {
  const int b = blockIdx.y;
  const unsigned R = run_cnt[b];
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R) return;
  const unsigned* K = skey + (size_t)b * pitch;
  const unsigned* I = sid + (size_t)b * pitch;
  const unsigned* S = run_start + (size_t)b * (pitch + 1);
  const unsigned cmask = (1u << cb) - 1u;
  const unsigned key = K[j], r = I[j];
  const unsigned st = S[r], ln = S[r + 1] - st;
  sst[(size_t)b * pitch + j] = st;
  slen[(size_t)b * pitch + j] = ln;
  unsigned char flag = 0;
  if ((key & cmask) != cmask && (j == 0 || K[j - 1] != key)) {
    unsigned have = ln;
    for (int t = 1; t < min_points && have < (unsigned)min_points && j + t < R; t++) {
      if (K[j + t] != key) break;
      const unsigned r2 = I[j + t];
      have += S[r2 + 1] - S[r2];
    }
    if (have >= (unsigned)min_points) {
      flag = 1;
      atomicOr(&words[gd[b].word_off + (key >> 6)].bits, 1ull << (key & 63));
    }
  }
  lhead[(size_t)b * pitch + j] = flag;
}

// first sorted run of searchable leaf `id`
__global__ void __launch_bounds__(256) k_segstart_runs(const unsigned* __restrict__ skey, const unsigned char* __restrict__ lhead, size_t pitch,
                                                       const unsigned* __restrict__ run_cnt, const GridDesc* __restrict__ gd,
                                                       const BitWord* __restrict__ words, unsigned* seg_start) {
  const int b = blockIdx.y;
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= run_cnt[b] || !lhead[(size_t)b * pitch + j]) return;
  const GridDesc& g = gd[b];
  const unsigned cell = skey[(size_t)b * pitch + j];
  const BitWord bw = words[g.word_off + (cell >> 6)];
  const unsigned id = bw.prefix + (unsigned)__popcll(bw.bits & ((1ull << (cell & 63)) - 1ull));
  seg_start[g.rec_off + id] = j;
}

// leaf.mean_ += pt ; leaf.cov_ += pt pt^T (impl:233-237): one WAVE per searchable leaf.  The leaf's runs (up to 64 at a time) sit in
// the lanes as (first point, length); position t of the leaf's concatenated runs is point st[run] + (t - prefix[run]) of the input
// rows -- neighbouring lanes read neighbouring floats.  64 points at a time are parked in LDS as nine f64 terms and lanes 0..8 add
// them strictly in order, exactly as k_leafsum does.
template <bool CENT>
__global__ void __launch_bounds__(64 * LS_WAVES) k_leafsum_runs(const float* __restrict__ tgt, size_t pitch, const unsigned* __restrict__ skey,
                                                                const unsigned* __restrict__ sst, const unsigned* __restrict__ slen,
                                                                const unsigned* __restrict__ run_cnt, const GridDesc* __restrict__ gd,
                                                                const unsigned* __restrict__ seg_start, double* sums, int* vox_idx, int* vox_n,
                                                                int cb, float* cent, int nx, int n_targets) {
  __shared__ double term[LS_WAVES][64][9];
  __shared__ float termf[CENT ? LS_WAVES : 1][64][3];
  __shared__ int mk[LS_WAVES][64];
  int b, bx;
  if (!xcd_map(nx, n_targets, bx, b)) return;
  const GridDesc& g = gd[b];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned* K = skey + (size_t)b * pitch;
  const unsigned* ST = sst + (size_t)b * pitch;
  const unsigned* LN = slen + (size_t)b * pitch;
  const float* X = tgt + (size_t)b * 3 * pitch;
  const unsigned R = run_cnt[b];
  const int id0 = bx * LS_WAVES + wv, idstep = nx * LS_WAVES;
  unsigned j_next = id0 < g.n_voxels ? seg_start[g.rec_off + id0] : 0u;
  for (int id = id0; id < g.n_voxels; id += idstep) {
    const unsigned j0 = j_next;
    if (id + idstep < g.n_voxels) j_next = seg_start[g.rec_off + id + idstep];      // next leaf's first run, one leaf ahead
    double acc = (lane == 3 || lane == 6 || lane == 8) ? 1.0 : 0.0;                  // cov_ is seeded with Identity (voxel_grid_covariance_omp.h:101)
    float accf = 0.f;
    int cnt = 0;
    unsigned key = 0;
    for (unsigned jb = j0;; jb += 64) {                                              // batches of up to 64 runs of this leaf
      const unsigned j = jb + lane;
      const unsigned kj = j < R ? K[j] : 0u;                                         // key, first point and length are fetched together:
      const unsigned stj = j < R ? ST[j] : 0u;                                       // one memory round trip per batch of runs
      const unsigned lnj = j < R ? LN[j] : 0u;
      if (jb == j0) key = __shfl(kj, 0);
      const bool in = j < R && kj == key;
      const unsigned long long inm = __ballot(in);
      const int nrun = (int)__popcll(inm);                                           // (runs of one cell are consecutive: `in` lanes form a prefix)
      const unsigned st = in ? stj : 0u;
      const int ln = in ? (int)lnj : 0;
      int pin = ln;                                                                   // inclusive prefix of the lengths over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pin, o); if (lane >= o) pin += t; }
      const int pex = pin - ln;
      const int T = __shfl(pin, 63);                                                  // points in this batch of runs
      for (int c0 = 0; c0 < T; c0 += 64) {
        // which run covers position c0 + lane?  runs that start inside the window mark their first position; a max-scan spreads
        // the marks; the run covering the window's first position is the last one that starts at or before it
        mk[wv][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (in && pex >= c0 && pex < c0 + 64) mk[wv][pex - c0] = lane;
        __builtin_amdgcn_wave_barrier();
        int run = mk[wv][lane];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(run, o); if (lane >= o) run = max(run, t); }
        const int first = (int)__popcll(__ballot(in && pex <= c0)) - 1;
        run = max(run, first);
        const int t = c0 + lane;
        const bool live = t < T;
        const unsigned rst = __shfl(st, run);
        const int rpex = __shfl(pex, run);
        const int m = min(64, T - c0);
        if (live) {
          const size_t pi = (size_t)rst + (size_t)(t - rpex);
          const float fx = X[pi], fy = X[pitch + pi], fz = X[2 * pitch + pi];
          const double x = (double)fx, y = (double)fy, z = (double)fz;
          double* tt = term[wv][lane];
          tt[0] = x; tt[1] = y; tt[2] = z;
          tt[3] = x * x; tt[4] = x * y; tt[5] = x * z; tt[6] = y * y; tt[7] = y * z; tt[8] = z * z;
          if (CENT) { termf[wv][lane][0] = fx; termf[wv][lane][1] = fy; termf[wv][lane][2] = fz; }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 9) {                            // strictly sequential adds (input order); the LDS reads are batched ahead of them
          int l = 0;
          for (; l + 8 <= m; l += 8) {
            double q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = term[wv][l + u][lane];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += q[u];
          }
          for (; l < m; l++) acc += term[wv][l][lane];
        } else if (CENT && lane < 12) {
          for (int l = 0; l < m; l++) accf += termf[wv][l][lane - 9];
        }
        __builtin_amdgcn_wave_barrier();
        cnt += m;
      }
      if (nrun < 64) break;
    }
    if (lane < 9) sums[(size_t)(g.rec_off + id) * 9 + lane] = acc;
    else if (CENT && lane < 12) cent[(size_t)(g.rec_off + id) * 3 + (lane - 9)] = accf / (float)cnt;   // centroid /= nr_points (impl:289)
    if (lane == 0) {
      vox_idx[g.rec_off + id] = (int)(key & ((1u << cb) - 1u));
      vox_n[g.rec_off + id] = cnt;
    }
  }
}
