// ndt_segsort.hpp -- stable LSD radix sort of (cell key, point id) pairs INSIDE each target's segment.
//
// The target build needs every target's points grouped by voxel cell in input order (the leaf sums are added in that order to
// stay bit-identical to the reference's sequential accumulation, voxel_grid_covariance_omp_impl.hpp:226-262).  A device-wide
// radix sort of pair << cb | cell does that, but sorts bits it does not have to (the pair field is already in order) and
// scatters every pass across the whole batch.  Here each pass moves a point only within its own target's segment
// (65,536 points = 512 KB of keys + ids: the scatter stays in one XCD's L2), over the cell field only, with the digit width
// chosen per build from the field's width (rs_plan): 2 passes of 10 bits for a 1 m KITTI-size grid (19-20 bits; measured
// 220 us against 259 us for 3 passes of 8 bits), 2 of 11 bits at 0.5 m, 3 of 11 for the prefilter's 31-bit voxel keys.
//   per pass:  k_rs_hist  (tile digit histograms)  ->  k_rs_scan (per segment, digit-major exclusive scan)  ->  k_rs_scatter
// A tile is 4096 consecutive positions handled by 4 waves, each owning 1024 consecutive positions in 16 rounds of 64, so the
// order inside a tile is (wave, round, lane) = position order, which is what makes the pass stable.
#pragma once
#include "ndt_types.hpp"

#define RS_MAX_BITS 11
#define RS_THREADS 256
#define RS_ROUNDS  16
#define RS_TILE    (RS_THREADS * RS_ROUNDS)

// digit width and pass count for a key field of `bits` bits: as few passes as 11-bit digits allow, then the narrowest of the
// instantiated widths (8, 10, 11) that still covers the field in that many passes
struct RsPlan { int bits, passes; };
static inline RsPlan rs_plan(int field_bits) {
  RsPlan p;
  p.passes = (field_bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  if (p.passes < 1) p.passes = 1;
  const int need = (field_bits + p.passes - 1) / p.passes;
  p.bits = need <= 8 ? 8 : (need <= 10 ? 10 : 11);
  return p;
}

// Where the keys of a target build come from: the points.  The cell index of a point (first pass of applyFilter,
// voxel_grid_covariance_omp_impl.hpp:218-223) is a dozen instructions on three coalesced loads, so the first pass's histogram kernel
// computes it, counts it and writes it (`keys`) for the scatter -- a key kernel of its own would write the keys and the histogram read
// them back.  (Computing them a second time in the first scatter instead of writing them moves as many bytes as before -- 12 per point
// read twice against 12 + 4 written + 4 + 4 read -- and measured 20 % slower: docs/experiments.md section 10c.)
// "Not binned" (padding, non-finite point, unusable grid) is the all-ones cell, which sorts to the end.
struct RsPoints { const float* tgt; const int* cnt; const GridDesc* gd; int cb; unsigned* keys; };
__device__ __forceinline__ unsigned rs_cell(const GridDesc& g, const bool grid_ok, const int n, const float* __restrict__ X, const size_t pitch,
                                            const size_t i, const unsigned none) {
  // branch-free (the loads of a wave's sixteen rounds are to be in flight together): a position past the cloud reads the row's last
  // entry and drops it
  const size_t ii = i < pitch ? i : pitch - 1;
  const float x = X[ii], y = X[pitch + ii], z = X[2 * pitch + ii];
  const int i0 = (int)(floorf(x * g.inv_leaf) - (float)g.min_b[0]);
  const int i1 = (int)(floorf(y * g.inv_leaf) - (float)g.min_b[1]);
  const int i2 = (int)(floorf(z * g.inv_leaf) - (float)g.min_b[2]);
  const unsigned cell = (unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2);
  return (grid_ok && (int)i < n && finite3(x, y, z)) ? cell : none;
}

// lanes of the wave holding the same digit as this lane (valid lanes only); BITS ballots
template <int BITS>
__device__ __forceinline__ unsigned long long rs_match(unsigned d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < BITS; bit++) {
    const bool one = (d >> bit) & 1u;
    const unsigned long long bb = __ballot(one);
    m &= one ? bb : ~bb;
  }
  return valid ? m : 0ull;
}

// digit histogram of every tile: hist[(b * tiles + tile) * RS_NB + d]
template <int BITS, bool POINTS = false>
__global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const unsigned* __restrict__ kin, size_t pitch, int shift, unsigned* hist, int tiles,
                                                        int n_targets, const RsPoints src = RsPoints()) {
  constexpr int RS_NB = 1 << BITS;
  __shared__ unsigned cnt[RS_THREADS / 64][RS_NB];
  int b, tile;
  if (!xcd_map(tiles, n_targets, tile, b)) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < (RS_THREADS / 64) * RS_NB; d += RS_THREADS) (&cnt[0][0])[d] = 0;
  __syncthreads();
  const unsigned* K = kin + (size_t)b * pitch;
  const size_t wbase = (size_t)tile * RS_TILE + (size_t)w * (RS_TILE / 4);
  unsigned key[RS_ROUNDS];
  if (POINTS) {
    const GridDesc& g = src.gd[b];
    const bool grid_ok = g.status == GRID_OK;
    const int n = src.cnt[b];
    const float* X = src.tgt + (size_t)b * 3 * pitch;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
      const size_t i = wbase + r * 64 + lane;
      const unsigned c = rs_cell(g, grid_ok, n, X, pitch, i, (1u << src.cb) - 1u);
      key[r] = i < pitch ? c : 0u;
      if (i < pitch) src.keys[(size_t)b * pitch + i] = c;      // the sort is segment-local: the target index is no part of the key
    }
  } else {
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
      const size_t i = wbase + r * 64 + lane;
      key[r] = i < pitch ? K[i] : 0u;
    }
  }
  // Counting needs no ranks, only totals: equal digits mostly come in runs (neighbouring points of a scan fall into the same
  // cell), so the head of each run adds the run's length with one LDS atomic -- a dozen instructions per round instead of the
  // ballots of a full digit match.
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const size_t i = wbase + r * 64 + lane;
    const bool valid = i < pitch;                              // valid lanes form a prefix of the wave
    const unsigned d = (key[r] >> shift) & (RS_NB - 1);
    const unsigned prev = __shfl_up(d, 1);
    const bool head = valid && (lane == 0 || d != prev);
    const unsigned long long heads = __ballot(head);
    const int nvalid = (int)__popcll(__ballot(valid));
    if (head) {
      const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
      const int next = above ? lane + 1 + ((int)__ffsll((long long)above) - 1) : nvalid;
      atomicAdd(&cnt[w][d], (unsigned)(next - lane));
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RS_NB; d += RS_THREADS)
    hist[((size_t)b * tiles + tile) * RS_NB + d] = ((cnt[0][d] + cnt[1][d]) + cnt[2][d]) + cnt[3][d];
}

// per segment: where does digit d of tile t start?  (digit-major, tile-minor exclusive scan; one block per segment)
template <int BITS>
__global__ void __launch_bounds__((1 << BITS) > 1024 ? 1024 : (1 << BITS)) k_rs_scan(const unsigned* __restrict__ hist, unsigned* offs, int tiles) {
  constexpr int RS_NB = 1 << BITS, NT = RS_NB > 1024 ? 1024 : RS_NB, PER = RS_NB / NT;   // PER digits per thread
  __shared__ unsigned sm[NT / 64 + 1];
  const int b = blockIdx.x;
  const unsigned* H = hist + (size_t)b * tiles * RS_NB;
  unsigned* O = offs + (size_t)b * tiles * RS_NB;
  // Thread t owns digits t, t + NT, ... (coalesced rows of the histogram); the digit-major exclusive scan is one block scan per
  // group of NT digits, each started at the total of the groups before it.  The tile loops are chains of dependent adds over
  // independent loads: eight loads are issued together (one memory round trip per eight tiles instead of one per tile -- a
  // 131,072-point segment has 32 tiles, and this kernel is one block per segment).
  unsigned carry = 0;
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int d = u * NT + (int)threadIdx.x;
    unsigned tot = 0;
    int t = 0;
    for (; t + 8 <= tiles; t += 8) {
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = H[(size_t)(t + k) * RS_NB + d];
#pragma unroll
      for (int k = 0; k < 8; k++) tot += v[k];
    }
    for (; t < tiles; t++) tot += H[(size_t)t * RS_NB + d];
    unsigned all;
    unsigned o = carry + block_exscan<NT>(tot, &all, sm);
    carry += all;
    t = 0;
    for (; t + 8 <= tiles; t += 8) {
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = H[(size_t)(t + k) * RS_NB + d];
#pragma unroll
      for (int k = 0; k < 8; k++) { O[(size_t)(t + k) * RS_NB + d] = o; o += v[k]; }
    }
    for (; t < tiles; t++) {
      O[(size_t)t * RS_NB + d] = o;
      o += H[(size_t)t * RS_NB + d];
    }
  }
}

// stable scatter of one tile.  FIRST: the point id of position i is i itself (no id array to read yet).
template <int BITS, bool FIRST>
__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(const unsigned* __restrict__ kin, const unsigned* __restrict__ vin,
                                                            unsigned* kout, unsigned* vout, size_t pitch, int shift,
                                                            const unsigned* __restrict__ offs, int tiles, int n_targets) {
  constexpr int RS_NB = 1 << BITS;
  __shared__ unsigned run[RS_THREADS / 64][RS_NB];
  int b, tile;
  if (!xcd_map(tiles, n_targets, tile, b)) return;   // a target's tiles on one XCD: its scattered writes combine in that L2
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < (RS_THREADS / 64) * RS_NB; d += RS_THREADS) (&run[0][0])[d] = 0;
  __syncthreads();
  const unsigned* K = kin + (size_t)b * pitch;
  const unsigned* V = vin + (size_t)b * pitch;
  const size_t wbase = (size_t)tile * RS_TILE + (size_t)w * (RS_TILE / 4);
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned key[RS_ROUNDS], val[RS_ROUNDS], info[RS_ROUNDS];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const size_t i = wbase + r * 64 + lane;
    key[r] = i < pitch ? K[i] : 0u;
    val[r] = FIRST ? (unsigned)i : (i < pitch ? V[i] : 0u);
  }
  // phase 1: this wave's digit counts, and for every position its rank inside its (wave, round, digit) group
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const size_t i = wbase + r * 64 + lane;
    const bool valid = i < pitch;
    const unsigned d = (key[r] >> shift) & (RS_NB - 1);
    const unsigned long long m = rs_match<BITS>(d, valid);
    const unsigned rank = (unsigned)__popcll(m & lt), c = (unsigned)__popcll(m);
    const unsigned leader = valid ? (unsigned)__ffsll((long long)m) - 1u : (unsigned)lane;
    if (valid && rank == 0) run[w][d] += c;
    info[r] = rank | (leader << 8) | (c << 16);
  }
  __syncthreads();
  // counts -> start offsets: digit d of this tile starts at offs[..][d]; waves follow each other in position order
  for (int d = threadIdx.x; d < RS_NB; d += RS_THREADS) {
    unsigned o = offs[((size_t)b * tiles + tile) * RS_NB + d];
#pragma unroll
    for (int w2 = 0; w2 < RS_THREADS / 64; w2++) { const unsigned c = run[w2][d]; run[w2][d] = o; o += c; }
  }
  __syncthreads();
  // phase 2: rounds in order; the group leader advances the wave's running offset of its digit and broadcasts the old value
  unsigned* KO = kout + (size_t)b * pitch;
  unsigned* VO = vout + (size_t)b * pitch;
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const size_t i = wbase + r * 64 + lane;
    const bool valid = i < pitch;
    const unsigned d = (key[r] >> shift) & (RS_NB - 1);
    const unsigned rank = info[r] & 0xffu, leader = (info[r] >> 8) & 0xffu, c = info[r] >> 16;
    unsigned prev = 0;
    if (valid && rank == 0) { prev = run[w][d]; run[w][d] = prev + c; }
    prev = __shfl(prev, (int)leader);
    if (valid) { KO[prev + rank] = key[r]; VO[prev + rank] = val[r]; }
    __builtin_amdgcn_wave_barrier();
  }
}

// One stable pass over `field` bits starting at bit `shift` of the keys of every segment: histogram, scan, scatter.
template <int BITS>
static inline void rs_pass_bits(hipStream_t s, const unsigned* kin, const unsigned* vin, unsigned* kout, unsigned* vout, size_t pitch, int shift,
                                unsigned* hist, unsigned* offs, int tiles, int n_segments, bool first, const RsPoints* points) {
  if (points) k_rs_hist<BITS, true><<<xcd_grid(tiles, n_segments), RS_THREADS, 0, s>>>(nullptr, pitch, shift, hist, tiles, n_segments, *points);   // writes kin
  else k_rs_hist<BITS><<<xcd_grid(tiles, n_segments), RS_THREADS, 0, s>>>(kin, pitch, shift, hist, tiles, n_segments);
  k_rs_scan<BITS><<<n_segments, (1 << BITS) > 1024 ? 1024 : (1 << BITS), 0, s>>>(hist, offs, tiles);
  if (first) k_rs_scatter<BITS, true><<<xcd_grid(tiles, n_segments), RS_THREADS, 0, s>>>(kin, vin, kout, vout, pitch, shift, offs, tiles, n_segments);
  else k_rs_scatter<BITS, false><<<xcd_grid(tiles, n_segments), RS_THREADS, 0, s>>>(kin, vin, kout, vout, pitch, shift, offs, tiles, n_segments);
}
// `first`: the point id of position i is i itself (no id array to read yet)
// `points`: (first pass of a target build) the histogram kernel computes the keys from the points and writes them to points->keys = kin
static inline void rs_pass(hipStream_t s, int bits, const unsigned* kin, const unsigned* vin, unsigned* kout, unsigned* vout, size_t pitch, int shift,
                           unsigned* hist, unsigned* offs, int tiles, int n_segments, bool first, const RsPoints* points = nullptr) {
  if (bits == 8) rs_pass_bits<8>(s, kin, vin, kout, vout, pitch, shift, hist, offs, tiles, n_segments, first, points);
  else if (bits == 10) rs_pass_bits<10>(s, kin, vin, kout, vout, pitch, shift, hist, offs, tiles, n_segments, first, points);
  else rs_pass_bits<11>(s, kin, vin, kout, vout, pitch, shift, hist, offs, tiles, n_segments, first, points);
}
