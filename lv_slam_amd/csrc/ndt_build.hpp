// ndt_build.hpp -- target build kernels: VoxelGridCovariance::applyFilter on the device
// (include/ndt_omp/voxel_grid_covariance_omp_impl.hpp:48-370; ndt_pca: voxel_grid_covariance_pca_impl.hpp:364-397).
#pragma once
#include "ndt_types.hpp"
#include "ndt_math.hpp"

// ------------------------------------------------------------------------------------ target build
// (the prefilter's one-cloud extremes, ndt_prefilter.hpp, are plain ints with an init launch)
__global__ void k_minmax_init(int* mm, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 6) mm[i] = (i % 6) < 3 ? INT_MAX : INT_MIN;
}

// The six extremes of a target live as unsigned "the larger wins" words whose all-zero state means "no finite point yet", so that the
// workspace memset that precedes every build is their initialisation: a maximum as ord ^ 0x80000000 (order-preserving int -> unsigned),
// a minimum as the complement of that.  No finite float maps to 0 in either form (f2ord of a finite float is neither INT_MIN nor INT_MAX).
__device__ __forceinline__ unsigned mm_enc_max(int ord) { return (unsigned)ord ^ 0x80000000u; }
__device__ __forceinline__ unsigned mm_enc_min(int ord) { return ~((unsigned)ord ^ 0x80000000u); }
__device__ __forceinline__ int mm_dec_max(unsigned e) { return (int)(e ^ 0x80000000u); }
__device__ __forceinline__ int mm_dec_min(unsigned e) { return (int)(~e ^ 0x80000000u); }

// getMinMax3D over finite points (voxel_grid_covariance_omp_impl.hpp:72, 211-216)
#define MM_ILP 4
struct __attribute__((packed, aligned(4))) MmQuad { float v[4]; };
__global__ void __launch_bounds__(256) k_minmax(const float* __restrict__ tgt, size_t pitch, const int* __restrict__ cnt, unsigned* mm) {
  const int b = blockIdx.y;
  const int n = cnt[b];
  const float* X = tgt + (size_t)b * 3 * pitch;
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  auto take = [&](const float x, const float y, const float z) {
    if (!finite3(x, y, z)) return;
    const int ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
    mn[0] = min(mn[0], ox); mx[0] = max(mx[0], ox);
    mn[1] = min(mn[1], oy); mx[1] = max(mx[1], oy);
    mn[2] = min(mn[2], oz); mx[2] = max(mx[2], oz);
  };
  // A thread reads four consecutive points of each row with one 16-byte load (the rows need no alignment beyond a float's), and
  // MM_ILP such groups a block stride apart before it looks at any of them: the pass is a pure stream, and what it needs is bytes in flight.
  const int stride = (int)(gridDim.x * blockDim.x) * 4;
  for (int i0 = (int)(blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += stride * MM_ILP) {
    MmQuad qx[MM_ILP], qy[MM_ILP], qz[MM_ILP];
#pragma unroll
    for (int u = 0; u < MM_ILP; u++) {
      const int i = i0 + u * stride;
      if (i + 4 <= n) {
        qx[u] = *reinterpret_cast<const MmQuad*>(X + i); qy[u] = *reinterpret_cast<const MmQuad*>(X + pitch + i);
        qz[u] = *reinterpret_cast<const MmQuad*>(X + 2 * pitch + i);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const bool in = i + k < n;
          qx[u].v[k] = in ? X[i + k] : NAN; qy[u].v[k] = in ? X[pitch + i + k] : NAN; qz[u].v[k] = in ? X[2 * pitch + i + k] : NAN;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < MM_ILP; u++)
#pragma unroll
      for (int k = 0; k < 4; k++) take(qx[u].v[k], qy[u].v[k], qz[u].v[k]);
  }
  __shared__ int red[4][6];
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o));
    }
  }
  if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; a++) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    const bool is_min = threadIdx.x < 3;
    int v = red[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = is_min ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
    if (is_min) { if (v != INT_MAX) atomicMax(&mm[b * 6 + threadIdx.x], mm_enc_min(v)); }
    else if (v != INT_MIN) atomicMax(&mm[b * 6 + threadIdx.x], mm_enc_max(v));
  }
}

// min_b_/max_b_/div_b_/divb_mul_ (voxel_grid_covariance_omp_impl.hpp:75-103)
__global__ void k_griddesc(const unsigned* __restrict__ mm, GridDesc* gd, unsigned* nwords, float leaf, int n_pairs, unsigned recs_per_pair) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_pairs) return;
  GridDesc g;
  memset(&g, 0, sizeof g);
  g.leaf = leaf;
  g.inv_leaf = 1.0f / leaf;                      // pcl::VoxelGrid::setLeafSize
  g.rec_off = (unsigned)b * recs_per_pair;
  if (mm[b * 6] == 0u) {
    g.status = GRID_EMPTY;
  } else {
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(mm_dec_min(mm[b * 6 + a])); mx[a] = ord2f(mm_dec_max(mm[b * 6 + 3 + a])); }
    if (grid_too_big((mx[0] - mn[0]) * g.inv_leaf, (mx[1] - mn[1]) * g.inv_leaf, (mx[2] - mn[2]) * g.inv_leaf)) {
      g.status = GRID_OVERFLOW;                  // impl:79-84: empty grid
    } else {
      for (int a = 0; a < 3; a++) {
        g.min_b[a] = (int)floorf(mn[a] * g.inv_leaf);
        g.max_b[a] = (int)floorf(mx[a] * g.inv_leaf);
        g.div_b[a] = g.max_b[a] - g.min_b[a] + 1;
      }
      long long nc = (long long)g.div_b[0] * g.div_b[1] * g.div_b[2];
      if (nc > MAX_CELLS) {
        g.status = GRID_CAP;
      } else {
        g.mul1 = g.div_b[0];
        g.mul2 = g.div_b[0] * g.div_b[1];
        g.ncells = (int)nc;
        g.nwords = (int)((nc + 63) >> 6) + 1;     // + one all-zero word: the landing cell of out-of-grid probes
      }
    }
  }
  gd[b] = g;
  nwords[b] = (unsigned)g.nwords;
}

// bitmap-pool offsets of the targets (exclusive scan of their word counts, one block) and the largest grid of the batch
__global__ void __launch_bounds__(1024) k_word_offsets(GridDesc* gd, const unsigned* __restrict__ nwords, int n_pairs, unsigned* out /* total words, max cells */) {
  __shared__ unsigned sm[17];
  unsigned base = 0, mx = 0;
  for (int b0 = 0; b0 < n_pairs; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const unsigned v = b < n_pairs ? nwords[b] : 0u;
    unsigned tot;
    const unsigned ex = block_exscan<1024>(v, &tot, sm);
    if (b < n_pairs) { gd[b].word_off = base + ex; mx = max(mx, (unsigned)gd[b].ncells); }
    base += tot;
  }
  if (mx) atomicMax(&out[1], mx);
  if (threadIdx.x == 0) out[0] = base;
}

// Asynchronous builds (stream mode) size the bitmap pool and the sort's key field from a PLAN made by earlier builds instead of waiting
// for `info` = {total bitmap words, cells of the largest grid} (k_word_offsets).  A batch that does not fit the plan must not touch
// memory it does not own: every grid of it becomes "no grid" (nothing is binned, marked, ranked or summed; its pairs align against an
// empty target and end at once) and the flag tells the host to run that batch again, synchronously, with a plan that fits.
__global__ void k_build_check(const unsigned* __restrict__ info, GridDesc* gd, unsigned* nwords, int n_pairs, unsigned plan_words, int plan_cb, unsigned* stat /* total, largest, flag */) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool fits = info[0] <= plan_words && (plan_cb >= 31 || info[1] + 1u <= (1u << plan_cb));
  if (b == 0) { stat[0] = info[0]; stat[1] = info[1]; stat[2] = fits ? 0u : 1u; }
  if (fits || b >= n_pairs) return;
  gd[b].status = GRID_CAP; gd[b].ncells = 0; gd[b].nwords = 0; gd[b].word_off = 0; gd[b].n_voxels = 0;
  nwords[b] = 0;
}

// (first pass of applyFilter -- the cell index per point, impl:218-223 -- is computed by the sort's first histogram kernel: ndt_segsort.hpp, rs_cell)

// mark cells that hold >= min_points points (impl:297) in the occupancy bitmap
// A sorted position i is the head of a searchable leaf's run iff its cell is binned, K[i-1] differs and K[i+min_points-1]
// is still the same key.  One element per thread makes these streaming kernels latency bound (a wave issues three loads and
// retires), so every thread tests RUN_ILP positions, one block stride apart, with all their loads in flight together.
#define RUN_ILP 4
template <typename KeyT>
__device__ __forceinline__ void run_heads(const KeyT* __restrict__ K, size_t pitch, size_t i0, size_t stride, int min_points, unsigned cmask,
                                          bool head[RUN_ILP], unsigned cell[RUN_ILP]) {
  KeyT k0[RUN_ILP], km[RUN_ILP], kl[RUN_ILP];
  const size_t span = (size_t)(min_points > 0 ? min_points - 1 : 0);
#pragma unroll
  for (int u = 0; u < RUN_ILP; u++) {
    const size_t i = i0 + (size_t)u * stride;
    const bool in = i < pitch;
    k0[u] = in ? K[i] : (KeyT)cmask;
    km[u] = (in && i != 0) ? K[i - 1] : ~(KeyT)0;
    kl[u] = (in && i + span < pitch) ? K[i + span] : ~(KeyT)0;
  }
#pragma unroll
  for (int u = 0; u < RUN_ILP; u++) {
    const size_t i = i0 + (size_t)u * stride;
    cell[u] = (unsigned)k0[u] & cmask;
    head[u] = i < pitch && cell[u] != cmask && (i == 0 || km[u] != k0[u]) && (i + span < pitch) && kl[u] == k0[u];
  }
}
// A head is also where the leaf's sums start (k_leafsum), so it is recorded here, once, instead of being found again after the ranking:
// every wave tests LS_SLICE consecutive sorted positions and owns a slice of `heads` for the heads it finds, in position order, their
// number in one word per slice.  No atomics, nothing to zero: a slice holds at most LS_SLICE / min_points + 1 heads.  Voxel ids are
// ranks in ascending cell order and the order is sorted by cell, so the concatenation of a target's slices IS its list of run starts
// by voxel id; k_rank, one workgroup per target anyway, closes the gaps.
#define LS_SLICE (64 * RUN_ILP)
static inline unsigned ls_slice_cap(int min_points) { return (unsigned)(LS_SLICE / (min_points > 0 ? min_points : 1) + 2); }
static inline unsigned ls_slices(size_t pitch) { return (unsigned)((pitch + LS_SLICE - 1) / LS_SLICE); }
template <typename KeyT>
__global__ void __launch_bounds__(256) k_mark(const KeyT* __restrict__ keys, size_t pitch, const GridDesc* __restrict__ gd,
                                               BitWord* words, unsigned* heads, unsigned* head_cnt, unsigned n_slices, unsigned slice_cap,
                                               int min_points, int cb) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const unsigned q = blockIdx.x * 4u + (threadIdx.x >> 6);      // the wave's slice
  if (q >= n_slices) return;
  const size_t i0 = (size_t)q * LS_SLICE + lane;
  bool head[RUN_ILP];
  unsigned cell[RUN_ILP];
  run_heads<KeyT>(keys + (size_t)b * pitch, pitch, i0, 64, min_points, (1u << cb) - 1u, head, cell);
  unsigned* H = heads + ((size_t)b * n_slices + q) * slice_cap;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned n = 0;
#pragma unroll
  for (int u = 0; u < RUN_ILP; u++) {
    const unsigned long long hm = __ballot(head[u]);
    if (head[u]) {
      atomicOr(&words[gd[b].word_off + (cell[u] >> 6)].bits, 1ull << (cell[u] & 63));
      H[n + (unsigned)__popcll(hm & below)] = (unsigned)(i0 + (size_t)u * 64);
    }
    n += (unsigned)__popcll(hm);
  }
  if (lane == 0) head_cnt[(size_t)b * n_slices + q] = n;
}
// exclusive popcount prefix over the bitmap words of each target: voxel id = rank in ascending cell order.
// One block of 16 waves per target; per round a wave owns 512 consecutive words as eight rows of 64 (lane = word: coalesced 16-byte
// accesses), scans every row with shuffles and carries the row totals along; the 16 wave totals go through one small LDS scan.
// A 0.5 m grid of ~21 k words takes 3 rounds.
#define RANK_ROWS 8
__global__ void __launch_bounds__(1024) k_rank(GridDesc* gd, BitWord* words, const unsigned* __restrict__ heads, const unsigned* __restrict__ head_cnt,
                                                 unsigned n_slices, unsigned slice_cap, unsigned* seg_start) {
  __shared__ unsigned wtot[17];
  const int b = blockIdx.x;
  BitWord* W = words + gd[b].word_off;
  const int nw = gd[b].nwords;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned base = 0;
  for (int w0 = 0; w0 < nw; w0 += 1024 * RANK_ROWS) {
    const int wbase = w0 + wv * 64 * RANK_ROWS;
    unsigned c[RANK_ROWS], ex[RANK_ROWS], run = 0;
#pragma unroll
    for (int u = 0; u < RANK_ROWS; u++) { const int w = wbase + u * 64 + lane; c[u] = (w < nw) ? (unsigned)__popcll(W[w].bits) : 0u; }
#pragma unroll
    for (int u = 0; u < RANK_ROWS; u++) {
      unsigned inc = c[u];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      ex[u] = run + inc - c[u];
      run += __shfl(inc, 63);
    }
    if (lane == 0) wtot[wv] = run;
    __syncthreads();
    if (wv == 0) {
      const unsigned v = lane < 16 ? wtot[lane] : 0u;
      unsigned inc = v;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      if (lane < 16) wtot[lane] = inc - v;
      if (lane == 15) wtot[16] = inc;
    }
    __syncthreads();
    const unsigned off = base + wtot[wv];
#pragma unroll
    for (int u = 0; u < RANK_ROWS; u++) { const int w = wbase + u * 64 + lane; if (w < nw) W[w].prefix = off + ex[u]; }
    base += wtot[16];
    __syncthreads();
  }
  if (threadIdx.x == 0) gd[b].n_voxels = (int)base;
  // run start of voxel id = the id-th head of the target's slices (k_mark): exclusive scan of the slice counts, then every thread moves
  // the few heads of its slice
  unsigned hbase = 0;
  for (unsigned q0 = 0; q0 < n_slices; q0 += 1024) {
    const unsigned q = q0 + threadIdx.x;
    const unsigned c = q < n_slices ? head_cnt[(size_t)b * n_slices + q] : 0u;
    unsigned tot;
    const unsigned ex = block_exscan<1024>(c, &tot, wtot);
    const unsigned* H = heads + ((size_t)b * n_slices + q) * slice_cap;
    unsigned* S = seg_start + gd[b].rec_off + hbase + ex;
    for (unsigned k = 0; k < c; k++) S[k] = H[k];
    hbase += tot;
  }
}

// The sorted order as POINTS: P[j] = (x, y, z) of the point at sorted position j of its target, 16 bytes each.  k_leafsum walks a leaf's run
// in sorted order; with only the ids sorted (vals) every trip of it is a dependent chain  ids -> three row gathers -> terms,  64 lanes
// presenting 64 different lines to the vector L1 per row.  This pass does the same gather once, as a stream (every address known up front,
// RUN_ILP positions per lane in flight), and leaves the leaf sums a coalesced 16-byte read per lane.
__global__ void __launch_bounds__(256) k_sorted_points(const float* __restrict__ tgt, size_t pitch, const unsigned* __restrict__ vals, float4* __restrict__ sorted,
                                                        int nx, int n_targets) {
  int b, bx;
  if (!xcd_map(nx, n_targets, bx, b)) return;      // a target's positions on one XCD: its points are in that L2 from the sort's first pass
  const float* X = tgt + (size_t)b * 3 * pitch;
  const unsigned* V = vals + (size_t)b * pitch;
  float4* P = sorted + (size_t)b * pitch;
  const size_t i0 = ((size_t)bx * 256 + threadIdx.x);
  const size_t stride = (size_t)nx * 256;
  for (size_t i = i0; i < pitch; i += stride * RUN_ILP) {
    unsigned v[RUN_ILP];
    float4 q[RUN_ILP];
#pragma unroll
    for (int u = 0; u < RUN_ILP; u++) { const size_t j = i + u * stride; v[u] = j < pitch ? V[j] : 0u; }
#pragma unroll
    for (int u = 0; u < RUN_ILP; u++) { const unsigned k = v[u] < pitch ? v[u] : 0u; q[u] = make_float4(X[k], X[pitch + k], X[2 * pitch + k], 0.f); }
#pragma unroll
    for (int u = 0; u < RUN_ILP; u++) { const size_t j = i + u * stride; if (j < pitch) P[j] = q[u]; }
  }
}

// leaf.mean_ += pt ; leaf.cov_ += pt pt^T (impl:233-237) for every searchable leaf: one WAVE per leaf.
// The wave gathers 64 points of the leaf's run at a time (the radix sort is stable, so the run is in input
// order), parks the nine f64 terms of each point in LDS, and lanes 0..8 -- one per accumulator -- add them
// strictly in input order, which keeps the sums bit-identical to the reference's sequential accumulation.
#ifdef NDT_TIMELINE
// per-phase shader-clock totals of the leaf sums (tools/leaf_timeline.py): [0..4] phases, [8] chunks, [9] leaves, [10] waves, [11] wave lifetimes
__device__ unsigned long long g_ltl[16];
#define LTL_DECL unsigned long long ltl[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ltl_last = __builtin_readcyclecounter(); const unsigned long long ltl_t0 = ltl_last
#define LTL_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); ltl[k] += t_ - ltl_last; ltl_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define LTL_COUNT(k, v) do { ltl[k] += (unsigned long long)(v); } while (0)
// (a sample of the waves adds its totals: 70 k waves adding to twelve words are a load of their own)
#define LTL_FLUSH() do { ltl[10] = 1; ltl[11] = __builtin_readcyclecounter() - ltl_t0; if ((threadIdx.x & 63) == 0 && (blockIdx.x % 24) == (threadIdx.x >> 6) * 5) for (int k_ = 0; k_ < 12; k_++) atomicAdd(&g_ltl[k_], ltl[k_]); } while (0)
#define LTL_ARRIVED1(a) asm volatile("" : "+v"(a))
#else
#define LTL_DECL do {} while (0)
#define LTL_STAMP(k) do {} while (0)
#define LTL_COUNT(k, v) do {} while (0)
#define LTL_FLUSH() do {} while (0)
#define LTL_ARRIVED1(a) do {} while (0)
#endif
#define LS_WAVES 4
template <typename KeyT, bool CENT, bool SORTED = false>
__global__ void __launch_bounds__(64 * LS_WAVES) k_leafsum(const float* __restrict__ tgt, size_t pitch,
                                                           const KeyT* __restrict__ keys, const unsigned* __restrict__ vals /* SORTED: the float4 points of k_sorted_points */,
                                                           const GridDesc* __restrict__ gd, const unsigned* __restrict__ seg_start,
                                                           double* sums, int* vox_idx, int* vox_n, int cb, float* cent,
                                                           int nx, int n_targets) {
  __shared__ double term[LS_WAVES][64][9];
  __shared__ float termf[CENT ? LS_WAVES : 1][64][3];
  int b, bx;
  if (!xcd_map(nx, n_targets, bx, b)) return;    // one target's leaves on one XCD: the point gathers hit in its L2
  const GridDesc& g = gd[b];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const KeyT* K = keys + (size_t)b * pitch;
  const unsigned* V = vals + (size_t)b * pitch;
  const float* X = tgt + (size_t)b * 3 * pitch;
  const int id0 = bx * LS_WAVES + wv, idstep = nx * LS_WAVES;
  size_t start_next = id0 < g.n_voxels ? seg_start[g.rec_off + id0] : 0;
  LTL_DECL;
  for (int id = id0; id < g.n_voxels; id += idstep) {
    LTL_COUNT(9, 1);
    const size_t start = start_next;
    if (id + idstep < g.n_voxels) start_next = seg_start[g.rec_off + id + idstep];   // next leaf's run start, one leaf ahead
    // accumulator order: S0 S1 S2 C00 C01 C02 C11 C12 C22 ; cov_ is seeded with Identity (voxel_grid_covariance_omp.h:101)
    double acc = (lane == 3 || lane == 6 || lane == 8) ? 1.0 : 0.0;
    float accf = 0.f;                            // lanes 9..11: leaf.centroid += pt (f32, impl:242-243)
    int cnt = 0;
    KeyT key = 0;
    for (size_t j0 = start;; j0 += 64) {
      const size_t j = j0 + lane;
      const bool inb = j < pitch;
      // key and point (id) of the run position are fetched together (what a lane past the run's end fetches is unused)
      const KeyT kj = inb ? K[j] : (KeyT)0;
      float4 pj = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned pi = 0u;
      if (SORTED) { if (inb) pj = reinterpret_cast<const float4*>(vals)[(size_t)b * pitch + j]; }
      else pi = inb ? V[j] : 0u;
#ifdef NDT_TIMELINE
      { KeyT kq = kj; LTL_ARRIVED1(kq); LTL_ARRIVED1(pi); LTL_ARRIVED1(pj.x); }
      LTL_STAMP(0);                              // keys and ids have arrived
      LTL_COUNT(8, 1);
#endif
      if (j0 == start) key = __shfl(kj, 0);      // the run's cell: its first entry
      const bool in = inb && kj == key;
      const int m = (int)__popcll(__ballot(in));
      if (in) {
        float fx = SORTED ? pj.x : X[pi], fy = SORTED ? pj.y : X[pitch + pi], fz = SORTED ? pj.z : X[2 * pitch + pi];
        LTL_ARRIVED1(fx); LTL_ARRIVED1(fy); LTL_ARRIVED1(fz);
        LTL_STAMP(1);                            // the rows have arrived
        const double x = (double)fx, y = (double)fy, z = (double)fz;
        double* t = term[wv][lane];
        t[0] = x; t[1] = y; t[2] = z;
        t[3] = x * x; t[4] = x * y; t[5] = x * z; t[6] = y * y; t[7] = y * z; t[8] = z * z;
        if (CENT) { termf[wv][lane][0] = fx; termf[wv][lane][1] = fy; termf[wv][lane][2] = fz; }
      }
      __builtin_amdgcn_wave_barrier();
      LTL_STAMP(2);                              // terms in LDS
      if (lane < 9) {                            // strictly sequential adds (input order); the LDS reads are batched ahead of them
        int l = 0;
        for (; l + 8 <= m; l += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; u++) t[u] = term[wv][l + u][lane];
#pragma unroll
          for (int u = 0; u < 8; u++) acc += t[u];
        }
        for (; l < m; l++) acc += term[wv][l][lane];
      } else if (CENT && lane < 12) {
        for (int l = 0; l < m; l++) accf += termf[wv][l][lane - 9];
      }
      __builtin_amdgcn_wave_barrier();
      LTL_STAMP(3);                              // sums
      cnt += m;
      if (m < 64) break;
    }
    if (lane < 9) sums[(size_t)(g.rec_off + id) * 9 + lane] = acc;
    else if (CENT && lane < 12) cent[(size_t)(g.rec_off + id) * 3 + (lane - 9)] = accf / (float)cnt;   // centroid /= nr_points (impl:289)
    if (lane == 0) {
      vox_idx[g.rec_off + id] = (int)((unsigned)key & ((1u << cb) - 1u));
      vox_n[g.rec_off + id] = cnt;
    }
    LTL_STAMP(4);                                // stores issued, loop
  }
  LTL_FLUSH();
}

// Tolerance arithmetic (MI355NDT_OPT_ARITH = 1) only: the same sums as a TREE.  The ordered sums above are nine lanes wide and one add deep per
// point -- that, not memory, is what k_leafsum costs (docs/experiments.md 10d).  Here every lane keeps the nine f64 sums of ITS positions of the
// run (position j of the run belongs to lane j % 64) and the wave adds them up once per leaf: 63 lanes busy instead of nine, one reduction per
// leaf instead of a chain per point.  Same terms, f64 throughout, another order: sums agree with the ordered ones to ~1e-16 relative
// (north_star's tolerance is 1e-4 m on the pose; BASELINE.md 5: no f64-only choice ever moved a pose bit), never bit for bit -- so never
// under the default arithmetic.  cov_'s Identity seed (voxel_grid_covariance_omp.h:101) goes in at the end.
#ifndef LS_TREE_LEAVES
#define LS_TREE_LEAVES 1       // leaves a wave works on side by side (their key / id loads, then their row gathers, go out together).  Measured, build of 271
#endif                         // targets: 1 / 2 / 4 leaves = 0.68 / 0.71 / 0.90 ms -- the kernel is bound by what its gathers cost the vector L1, not by round trips
__global__ void __launch_bounds__(64 * LS_WAVES) k_leafsum_tree(const float* __restrict__ tgt, size_t pitch, const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                                                const GridDesc* __restrict__ gd, const unsigned* __restrict__ seg_start,
                                                                double* sums, int* vox_idx, int* vox_n, int cb, int nx, int n_targets) {
  int b, bx;
  if (!xcd_map(nx, n_targets, bx, b)) return;
  const GridDesc& g = gd[b];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned* K = keys + (size_t)b * pitch;
  const unsigned* V = vals + (size_t)b * pitch;
  const float* X = tgt + (size_t)b * 3 * pitch;
  constexpr int NL = LS_TREE_LEAVES;
  const int id0 = bx * LS_WAVES + wv, idstep = nx * LS_WAVES;
  const int nv = g.n_voxels;
  for (int idb = id0; idb < nv; idb += NL * idstep) {
    size_t start[NL];
    bool have[NL];
#pragma unroll
    for (int u = 0; u < NL; u++) { const int id = idb + u * idstep; have[u] = id < nv; start[u] = have[u] ? seg_start[g.rec_off + id] : 0; }
    double a[NL][9];
    int cnt[NL];
    unsigned key[NL];
    bool more[NL];
#pragma unroll
    for (int u = 0; u < NL; u++) {
#pragma unroll
      for (int k = 0; k < 9; k++) a[u][k] = 0.0;
      cnt[u] = 0; key[u] = 0u; more[u] = have[u];
    }
    for (int chunk = 0;; chunk++) {                 // chunk c of every leaf that still has one, side by side
      bool any = false;
#pragma unroll
      for (int u = 0; u < NL; u++) any = any || more[u];
      if (!any) break;
      unsigned kj[NL], pi[NL];
      bool inb[NL];
#pragma unroll
      for (int u = 0; u < NL; u++) {
        const size_t j = start[u] + (size_t)chunk * 64 + lane;
        inb[u] = more[u] && j < pitch;
        kj[u] = inb[u] ? K[j] : 0u;
        pi[u] = inb[u] ? V[j] : 0u;
      }
      float fx[NL], fy[NL], fz[NL];
      bool in[NL];
#pragma unroll
      for (int u = 0; u < NL; u++) {
        if (chunk == 0) key[u] = __shfl(kj[u], 0);
        in[u] = inb[u] && kj[u] == key[u];
        fx[u] = fy[u] = fz[u] = 0.f;
        if (in[u]) { fx[u] = X[pi[u]]; fy[u] = X[pitch + pi[u]]; fz[u] = X[2 * pitch + pi[u]]; }
      }
#pragma unroll
      for (int u = 0; u < NL; u++) {
        const int m = (int)__popcll(__ballot(in[u]));
        if (in[u]) {
          const double x = (double)fx[u], y = (double)fy[u], z = (double)fz[u];
          a[u][0] += x; a[u][1] += y; a[u][2] += z;
          a[u][3] += x * x; a[u][4] += x * y; a[u][5] += x * z; a[u][6] += y * y; a[u][7] += y * z; a[u][8] += z * z;
        }
        cnt[u] += m;
        more[u] = more[u] && m == 64;
      }
    }
    // wave sum as a reduce-scatter (the sweep's scheme, ndt_sweep.hpp): 9 -> 5 values across the lane halves, 5 -> 3 across row pairs, then rows
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int u = 0; u < NL; u++) {
      if (!have[u]) continue;                      // (wave-uniform)
      const int id = idb + u * idstep;
      double p1[5], p2[3];
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const double x = a[u][i], v = (i + 5 < 9) ? a[u][i + 5] : 0.0;
        const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(v), false, false);
        const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(v), false, false);
        p1[i] = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);   // lanes 0..31: sum i, lanes 32..63: sum i + 5
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double x = p1[i], v = (i + 3 < 5) ? p1[i + 3] : 0.0;
        const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(v), false, false);
        const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(v), false, false);
        double w = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);   // even rows: p1[i], odd rows: p1[i + 3]
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) w += __shfl_xor(w, o);
        p2[i] = w;
      }
      if ((lane & 15) == 0) {
        // row 0 (lane 0): sums 0 1 2, row 1 (lane 16): 3 4, row 2 (lane 32): 5 6 7, row 3 (lane 48): 8 -- S0 S1 S2 C00 C01 C02 C11 C12 C22
        const int row = lane >> 4, base = 3 * (row & 1) + 5 * (row >> 1), nrow = (row & 1) ? ((row >> 1) ? 1 : 2) : 3;
        double* o = sums + (size_t)(g.rec_off + id) * 9;
#pragma unroll
        for (int i = 0; i < 3; i++) if (i < nrow) { const int k = base + i; o[k] = p2[i] + ((k == 3 || k == 6 || k == 8) ? 1.0 : 0.0); }
      }
      if (lane == 0) {
        vox_idx[g.rec_off + id] = (int)(key[u] & ((1u << cb) - 1u));
        vox_n[g.rec_off + id] = cnt[u];
      }
    }
  }
}

// second pass of applyFilter (impl:282-367; pca impl:364-397): one thread per searchable leaf
__global__ void __launch_bounds__(256) k_voxels(const GridDesc* __restrict__ gd, const double* __restrict__ sums,
                                                 VoxelRec* recs, int* vox_n, double eig_mult, int pca, double* icov64, int* kd_weight,
                                                 VoxelRecF* recs_fast /* tolerance arithmetic only, else null */) {
  const int b = blockIdx.y;
  const GridDesc& g = gd[b];
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= g.n_voxels) return;
  const double* A = sums + (size_t)(g.rec_off + id) * 9;
  const double S[3] = {A[0], A[1], A[2]};
  const double C[9] = {A[3], A[4], A[5], A[4], A[6], A[7], A[5], A[7], A[8]};
  const int cnt = vox_n[g.rec_off + id];
  const double dn = (double)cnt;
  double mu[3] = {S[0] / dn, S[1] / dn, S[2] / dn};                              // impl:293
  double cov[9];
  for (int a = 0; a < 3; a++)
    for (int c = 0; c < 3; c++) cov[a * 3 + c] = (C[a * 3 + c] - 2 * (S[a] * mu[c])) / dn + mu[a] * mu[c];   // impl:329
  const double f = (dn - 1.0) / dn;
  for (int a = 0; a < 9; a++) cov[a] *= f;                                       // impl:330
  double ev[3], Vm[9];
  ndtm::eigen_sym3(cov, ev, Vm);                                                 // impl:333-335
  VoxelRec r;
  r.mean[0] = mu[0]; r.mean[1] = mu[1]; r.mean[2] = mu[2];
  for (int a = 0; a < 9; a++) r.icov[a] = 0.f;
  r.weight = 1;
  int n_out = cnt, kd_w = 0;
  if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) {                                    // impl:337-341
    n_out = -1;
    r.weight = VOX_DEAD;
    if (icov64) for (int a = 0; a < 9; a++) icov64[(size_t)(g.rec_off + id) * 9 + a] = 0.0;   // icov_ stays at its Zero seed (h:103)
  } else {
    const double minev = eig_mult * ev[2];                                       // impl:345
    if (ev[0] < minev) {
      ev[0] = minev;
      if (ev[1] < minev) ev[1] = minev;
      double VD[9], Vi[9];
      for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) VD[a * 3 + c] = Vm[a * 3 + c] * ev[c];
      ndtm::mat3_inverse(Vm, Vi);
      ndtm::mat3_mul(VD, Vi, cov);                                               // impl:355
    }
    if (pca) {                                                                   // pca impl:364-397
      double s0 = sqrt(ev[0]), s1 = sqrt(ev[1]), s2 = sqrt(ev[2]);
      double f0 = (s2 - s1) / s2, f1 = (s1 - s0) / s2, f2 = s0 / s2;
      int label = 1;
      double fm = f0;
      if (f1 > fm) { fm = f1; label = 2; }
      if (f2 > fm) { label = 3; }
      double scale = (label == 2) ? 1.25 : ((label == 1) ? 0.75 : 1.0);
      double d2d = scale * sqrt((mu[0] * mu[0] + mu[1] * mu[1]) + mu[2] * mu[2]);
      r.weight = (int)d2d;                                                       // getDimension2d() returns int (pca.h:222-226)
      kd_w = r.weight;
    }
    double ic[9];
    ndtm::mat3_inverse(cov, ic);                                                 // impl:359
    bool bad = false;
    for (int a = 0; a < 9; a++) { if (!isfinite(ic[a])) bad = true; r.icov[a] = (float)ic[a]; }
    if (bad) { n_out = -1; r.weight = VOX_DEAD; }                                // impl:360-364
    if (icov64) for (int a = 0; a < 9; a++) icov64[(size_t)(g.rec_off + id) * 9 + a] = ic[a];   // computeHessian reads icov_ in double
  }
  recs[g.rec_off + id] = r;
  vox_n[g.rec_off + id] = n_out;
  if (recs_fast) {                               // the record as the tolerance-arithmetic sweep reads it (ndt_types.hpp)
    VoxelRecF f;
    for (int a = 0; a < 3; a++) { f.mh[a] = (float)r.mean[a]; f.ml[a] = (float)(r.mean[a] - (double)f.mh[a]); }
    f.c[0] = r.icov[0]; f.c[1] = 0.5f * (r.icov[1] + r.icov[3]); f.c[2] = 0.5f * (r.icov[2] + r.icov[6]);
    f.c[3] = r.icov[4]; f.c[4] = 0.5f * (r.icov[5] + r.icov[7]); f.c[5] = r.icov[8];
    f.pad_[0] = f.pad_[1] = f.pad_[2] = 0.f;
    f.weight = r.weight;
    recs_fast[g.rec_off + id] = f;
  }
  // ndt_pca + KDTREE reads the weight of EVERY leaf radiusSearch returns: (int)dimension_2d_ as computed (also when the
  // inverse failed afterwards), and the constructor's 0 for an eigen-failed leaf (voxel_grid_covariance_pca.h:143)
  if (kd_weight) kd_weight[g.rec_off + id] = kd_w;
}

