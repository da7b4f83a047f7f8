// NOT PART OF THE BUILD.  The linear-scan formulation of the leaf sums (round 4, docs/experiments.md section 10c): bit-exact on all 51
// GPU parity tests, 281 us against k_leafsum's 228 us at config 3 (2.3x the VALU wave-instructions, 1.9x the L1 accesses).  Kept as
// the record of what was measured; it dropped into ndt_build.hpp in place of k_segstart + k_leafsum, launched as
//   k_leafscan<unsigned, CENT><<<xcd_grid(nx, B), 256>>>(tgt, pitch, sorted keys, sorted ids, grid, ranked bitmap, sums, vox_idx, vox_n, cb, cent, nx, B)
// with nx = min(ceil(pitch / (256 * LSC_TILES)), 64).
// leaf.mean_ += pt ; leaf.cov_ += pt pt^T (impl:233-237) for every searchable leaf, as ONE LINEAR SCAN of the sorted order.
//
// The reference adds a leaf's points one by one in input order (f64), and the stable sort keeps that order inside every cell's run, so
// the nine sums of a leaf are nine ordered chains over its run.  A wave takes LSC_TILES consecutive tiles of 64 sorted positions.  Per
// tile: keys and point ids (coalesced), the points themselves (gathered; consecutive ids inside a lidar ring's run share lines), the
// nine f64 terms of every point parked in LDS; the run boundaries of the tile are one ballot.  Then the chains: the tile's runs are
// dealt to SEVEN groups of nine lanes (lane = (run, sum)), which add their runs' terms in order side by side -- a tile costs as many
// add steps as its longest run, not 64, and there is no per-leaf trip (run start -> key -> points), no list of run starts, no wave
// that owns a 7-point leaf.  A run that crosses tiles is carried in group 0's registers; a run belongs to the wave in whose chunk it
// STARTS (a wave skips the run it finds open at its first position and follows its own last run past the end of its chunk).
// The voxel id of a run is its cell's rank in the marked bitmap (k_mark, k_rank); runs of unmarked cells (fewer than min_points) and
// the run of unbinned entries at the end of the order are scanned over and dropped.
#define LSC_TILES 4
#define LSC_GROUPS 7
template <typename KeyT, bool CENT>
__global__ void __launch_bounds__(256) k_leafscan(const float* __restrict__ tgt, size_t pitch,
                                                   const KeyT* __restrict__ keys, const unsigned* __restrict__ vals,
                                                   const GridDesc* __restrict__ gd, const BitWord* __restrict__ words,
                                                   double* sums, int* vox_idx, int* vox_n, int cb, float* cent,
                                                   int nx, int n_targets) {
  __shared__ double term[4][65][9];              // row 64 is unused padding for clamped reads
  __shared__ int headpos[4][66];
  int b, bx;
  if (!xcd_map(nx, n_targets, bx, b)) return;    // one target on one XCD: the point gathers hit in its L2
  const GridDesc& g = gd[b];
  if (g.status != GRID_OK || g.n_voxels == 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int grp = lane / 9, k = lane - grp * 9;  // lane 63: group 7, idle in the chains
  const KeyT* K = keys + (size_t)b * pitch;
  const unsigned* V = vals + (size_t)b * pitch;
  const float* X = tgt + (size_t)b * 3 * pitch;
  const BitWord* W = words + g.word_off;
  const unsigned cmask = (1u << cb) - 1u, NONE = 0xFFFFFFFFu;
  const unsigned long long le_mask = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
  const double seed = (k == 3 || k == 6 || k == 8) ? 1.0 : 0.0;   // S0 S1 S2 C00 C01 C02 C11 C12 C22; cov_ is seeded with Identity (omp.h:101)
  const size_t n_chunks = (pitch + 64 * LSC_TILES - 1) / (64 * LSC_TILES);
  // what a tile's keys say about its runs (lane-wise; Hm and S1 wave-uniform)
  struct Tile { unsigned kj, pi, cell; bool head, bin; unsigned long long Hm; int S1, sidx; };
  for (size_t chunk = (size_t)bx * 4 + wv; chunk < n_chunks; chunk += (size_t)nx * 4) {
    const size_t j_begin = chunk * (64 * LSC_TILES);
    unsigned klast = j_begin ? (unsigned)K[j_begin - 1] : NONE;
    auto fetch = [&](int t, Tile& T) {           // keys and point ids of tile t (positions past the end of the order read as NONE)
      const size_t j = j_begin + (size_t)t * 64 + lane;
      const bool inb = j < pitch;
      T.kj = inb ? (unsigned)K[j] : NONE;
      T.pi = inb ? V[j] : 0u;
    };
    auto derive = [&](Tile& T) {                 // (needs the tile before it derived: klast)
      unsigned kp = __shfl_up(T.kj, 1);
      if (lane == 0) kp = klast;
      klast = __shfl(T.kj, 63);
      T.head = T.kj != kp;                       // (the first position past the end of the order is a head as well: it closes the last run)
      T.Hm = __ballot(T.head);
      T.S1 = (int)__popcll(T.Hm);                // runs 1..S1 start in this tile; run 0 is the one open at its first position
      T.sidx = (int)__popcll(T.Hm & le_mask);
      T.cell = T.kj & cmask;
      T.bin = T.kj != NONE && T.cell != cmask;
    };
    // The software pipeline: while tile t is parked, chained and stored, the points (and bitmap words) of tile t + 1 and the keys and ids
    // of tile t + 2 are in flight -- the sorted order is read front to back, so every address is known ahead of time.  (Points are
    // gathered for every binned position of a tile; whether its run 0 is this wave's is known only when the tile before it is done.)
    Tile A, B, C;
    float xA = 0.f, yA = 0.f, zA = 0.f, xB, yB, zB;
    unsigned long long bitsA = 0, bitsB; unsigned prefA = 0, prefB;
    auto gather = [&](const Tile& T, float& x, float& y, float& z, unsigned long long& bits, unsigned& pref) {
      const unsigned q = T.bin ? T.pi : 0u;
      x = X[q]; y = X[pitch + q]; z = X[2 * pitch + q];
      const BitWord bw = W[(T.bin && T.head) ? (T.cell >> 6) : 0u];
      bits = bw.bits; pref = bw.prefix;
    };
    fetch(0, A); fetch(1, B);
    derive(A);
    gather(A, xA, yA, zA, bitsA, prefA);
    // the open run at the tile boundary: is it this wave's, its sums so far (lanes 0..8), points so far, voxel id (< 0: not searchable), cell
    bool live = false;
    double carry = 0.0; float carryf = 0.f;
    int ccnt = 0, cid = -1; unsigned ccell = 0;
    for (int t = 0;; t++) {
      const bool in_chunk = t < LSC_TILES;
      if (!in_chunk && !live) break;
      fetch(t + 2, C);
      derive(B);
      gather(B, xB, yB, zB, bitsB, prefB);
      // ---- tile t
      const int S1 = A.S1;
      const bool own = A.bin && (A.sidx == 0 ? live : in_chunk);
      int idv = -1;                              // at a run's first lane: its voxel id
      if (A.head && own && ((bitsA >> (A.cell & 63)) & 1ull)) idv = (int)(prefA + (unsigned)__popcll(bitsA & ((1ull << (A.cell & 63)) - 1ull)));
      if (A.head) headpos[wv][A.sidx] = lane;
      if (lane == 0) { headpos[wv][0] = 0; headpos[wv][S1 + 1] = 64; }
      if (own) {
        const double x = (double)xA, y = (double)yA, z = (double)zA;
        double* q = term[wv][lane];
        q[0] = x; q[1] = y; q[2] = z;
        q[3] = x * x; q[4] = x * y; q[5] = x * z; q[6] = y * y; q[7] = y * z; q[8] = z * z;
      }
      __builtin_amdgcn_wave_barrier();
      double acc = 0.0; float accf = 0.f;
      int len_last = 0, id_last = -1; unsigned cell_last = 0; bool own_last = false;
      for (int r = 0; r * LSC_GROUPS <= S1; r++) {
        const int s = r * LSC_GROUPS + grp;
        const bool act = grp < LSC_GROUPS && s <= S1;
        const int a = act ? headpos[wv][s] : 0, e = act ? headpos[wv][s + 1] : 0;
        const int first = a < 64 ? a : 63;       // (the "run" of positions past the end of the order starts at 63 at the latest)
        // this wave's run, and a searchable one?  (run 0 may have no position in this tile -- the tile starts with a head -- and is
        // complete all the same)
        // (the shuffles run with every lane enabled: a source lane that sits out reads as zero)
        const int sh_own = __shfl((int)own, first), sh_id = __shfl(idv, first), sh_cell = __shfl((int)A.cell, first);
        const bool s_own = act && (s == 0 ? live : (bool)sh_own);
        const int s_id = s == 0 ? cid : sh_id;
        const unsigned s_cell = s == 0 ? ccell : (unsigned)sh_cell;
        const bool valid = s_own && s_id >= 0;
        const int len = valid ? e - a : 0;       // dropped runs are not added up at all
        const int cnt0 = s == 0 ? ccnt : 0;
        acc = s == 0 ? carry : seed;
        accf = s == 0 ? carryf : 0.f;
        // the chain: eight LDS reads ahead of eight dependent adds while any group has eight points left, then the (< 8) rest of every run
        int i = 0;
        while (__ballot(i + 8 <= len)) {
          if (i + 8 <= len) {
            double tv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) tv[u] = term[wv][a + i + u][k];
#pragma unroll
            for (int u = 0; u < 8; u++) { acc += tv[u]; if (CENT) accf += (float)tv[u]; }
            i += 8;
          }
        }
        {
          const int rem = len - i;
          double tv[7];
#pragma unroll
          for (int u = 0; u < 7; u++) { const int row = u < rem ? a + i + u : 64; tv[u] = term[wv][row][k]; }
#pragma unroll
          for (int u = 0; u < 7; u++) if (u < rem) { acc += tv[u]; if (CENT) accf += (float)tv[u]; }
        }
        const int cnt = cnt0 + (e - a);
        if (valid && s < S1) {                   // a head follows inside this tile: the run is complete
          const size_t id = (size_t)g.rec_off + (size_t)s_id;
          sums[id * 9 + k] = acc;
          if (CENT && k < 3) cent[id * 3 + k] = accf / (float)cnt;   // leaf.centroid += pt in f32, /= nr_points (impl:242-243, 289)
          if (k == 0) { vox_idx[id] = (int)s_cell; vox_n[id] = cnt; }
        }
        if (s == S1) { len_last = cnt; id_last = s_id; cell_last = s_cell; own_last = s_own; }
      }
      // the tile's last run stays open: its chain so far moves to group 0
      const int src = (S1 % LSC_GROUPS) * 9;
      carry = __shfl(acc, src + (lane < 9 ? lane : 0));
      carryf = __shfl(accf, src + (lane < 9 ? lane : 0));
      ccnt = __shfl(len_last, src);
      cid = __shfl(id_last, src);
      ccell = (unsigned)__shfl((int)cell_last, src);
      live = (bool)__shfl((int)own_last, src);
      __builtin_amdgcn_wave_barrier();
      A = B; B = C;
      xA = xB; yA = yB; zA = zB; bitsA = bitsB; prefA = prefB;
    }
  }
}

