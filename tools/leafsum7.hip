// tools/leafsum7.hip -- a self-checking micro-benchmark, written at the end of round 3 for the next one.
//
// Measured with the last seconds of round 3's GPU budget (271 targets, stretches of 8 consecutive point ids; every variant bit-identical
// to the product kernel and to the CPU loop):
//   product k_leafsum (wave per leaf)                                         226-232 us
//   k_leafsum7, f64 staging, 4 waves per workgroup                              260 us
//   k_leafsum7, f64 staging, 1 wave per workgroup                               207 us     (leaves in cell order instead of by length: 264 us)
//   k_leafsum7, f32 staging (-DL7_F32), 1 wave per workgroup (-DL7_WAVES=1)     187 us     (LDS reads batched by 4 / 8 / 16: 189 / 187 / 196 us)
//   single point ids in random order (argv[3] = 1): product 342 us, k_leafsum7 360-390 us -- both bound by the gathers there
// i.e. -17 % at best, and only with the leaves ordered by run length.  Knock-outs of the 187 us variant (wrong results, timing only):
// without the ordered adds 143 us, without the point gathers 122 us, without both 54 us -- here the gathers (21 per lane per window,
// all in flight together) are the larger half, where the product kernel hides them behind its LDS traffic.
//
// A self-checking micro-benchmark for the formulation of the leaf sums that DESIGN.md 9.3 names as the one with a future:
//   lane = (leaf slot, sum): 63 lanes = 7 leaves x 9 sums (S0 S1 S2 C00 C01 C02 C11 C12 C22), the seven ordered chains of a wave
//   advancing together, the points of the seven leaves staged in LDS as f64 x, y, z (24 B per point instead of the 72 B of nine
//   products), leaves dealt to waves in order of run length so that a wave's seven leaves are of similar length.
// Against it: the product kernel k_leafsum<unsigned, false> (lv_slam_amd/csrc/ndt_build.hpp: one wave per leaf, nine lanes adding).
// Both run on the same synthetic input -- B targets of 65,536 points each, leaf lengths drawn like the benchmark's targets (about
// 1,450 searchable leaves per target, 43 points per leaf on average, a heavy tail up to ~900) -- and must produce the same bits
// (the per-leaf sums are strictly sequential in input order in both; a CPU loop checks a sample of leaves as well).
//
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DL7_WAVES=1] [-DL7_F32] [-DL7_BATCH=8] -Ilv_slam_amd/csrc -Iinclude tools/leafsum7.hip -o tools/leafsum7
//   run:   tools/leafsum7 [targets=271] [repeats=20] [consecutive point ids per stretch=8] [leaves by falling length=1]
//
// Not part of the product.  (The run lengths it needs are already known to k_mark -- the min_points test -- and the per-target
// ordering would be a counting sort over ~1,450 leaves, ~10 us: a net gain of ~30 us on a 5.3 ms step.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_build.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(3); } } while (0)

#define L7_SLOTS 7
#ifndef L7_BATCH
#define L7_BATCH 8                        // ordered-add steps whose LDS reads are issued together
#endif
#ifndef L7_WAVES
#define L7_WAVES 4
#endif
#ifdef L7_F32
typedef float stage_t;                  // staged as f32 (12 B per point), converted when read
#else
typedef double stage_t;
#endif
#define L7_STRIDE (64 * 3 + 3)          // f64 words per slot: 64 points x (x, y, z) + 3 words of padding (slot s starts at bank 3 s: the
                                        // 7 x 3 words a wave reads per step fall into 21 different 8-byte banks)

// One wave = groups of seven leaves (order[]: the target's leaves by falling run length); lane = slot * 9 + sum.
//   seg_start / seg_len: run start and run length of every searchable leaf (global leaf id = rec_off + id)
//   order: per target, leaf ids (local) sorted by falling length
template <bool BATCHED>
__global__ void __launch_bounds__(64 * L7_WAVES) k_leafsum7(const float* __restrict__ tgt, size_t pitch, const unsigned* __restrict__ vals,
                                                            const GridDesc* __restrict__ gd, const unsigned* __restrict__ seg_start,
                                                            const unsigned* __restrict__ seg_len, const unsigned* __restrict__ order,
                                                            double* sums, int* vox_n, int nx, int n_targets) {
  __shared__ stage_t stage[L7_WAVES][L7_SLOTS * L7_STRIDE];
  int b, bx;
  if (!xcd_map(nx, n_targets, bx, b)) return;
  const GridDesc& g = gd[b];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned* V = vals + (size_t)b * pitch;
  const float* X = tgt + (size_t)b * 3 * pitch;
  const int nv = g.n_voxels;
  const int n_groups = (nv + L7_SLOTS - 1) / L7_SLOTS;
  // which sum this lane carries, and from which coordinates its term is made: term = c[ia] * (ib < 0 ? 1 : c[ib])
  const int slot = lane / 9, a = lane - slot * 9;                                   // lane 63: slot 7 = idle
  const int ia = a < 3 ? a : (a < 6 ? 0 : (a < 8 ? 1 : 2));
  const int ib = a < 3 ? -1 : (a == 3 ? 0 : a == 4 ? 1 : a == 5 ? 2 : a == 6 ? 1 : a == 7 ? 2 : 2);
  stage_t* st = stage[wv];
  for (int grp = bx * L7_WAVES + wv; grp < n_groups; grp += nx * L7_WAVES) {
    // the seven leaves of this group (slot s: order[grp * 7 + s]); every lane keeps the start / length of ALL slots it gathers for
    unsigned start_s[L7_SLOTS], len_s[L7_SLOTS];
    int id_s[L7_SLOTS];
#pragma unroll
    for (int s = 0; s < L7_SLOTS; s++) {
      const int k = grp * L7_SLOTS + s;
      id_s[s] = k < nv ? (int)order[g.rec_off + k] : -1;
      start_s[s] = id_s[s] >= 0 ? seg_start[g.rec_off + id_s[s]] : 0u;
      len_s[s] = id_s[s] >= 0 ? seg_len[g.rec_off + id_s[s]] : 0u;
    }
    unsigned my_len = 0u;                                                            // the length of the lane's own leaf (select, no dynamic indexing)
#pragma unroll
    for (int s = 0; s < L7_SLOTS; s++) my_len = slot == s ? len_s[s] : my_len;
    unsigned maxlen = 0;
#pragma unroll
    for (int s = 0; s < L7_SLOTS; s++) maxlen = len_s[s] > maxlen ? len_s[s] : maxlen;
    // accumulator: cov_ is seeded with Identity (voxel_grid_covariance_omp.h:101): C00, C11, C22 start at 1
    double acc = (a == 3 || a == 6 || a == 8) ? 1.0 : 0.0;
    for (unsigned w0 = 0; w0 < maxlen; w0 += 64) {
      // ---- gather: 64 run positions of each of the seven leaves, all loads of the window in flight together
      unsigned pid[L7_SLOTS];
      bool in[L7_SLOTS];
#pragma unroll
      for (int s = 0; s < L7_SLOTS; s++) {
        in[s] = w0 + lane < len_s[s];
        const size_t j = (size_t)start_s[s] + w0 + lane;
        pid[s] = V[in[s] ? j : (size_t)start_s[s]];                                    // (clamped: no load sits under a divergent branch)
      }
      float fx[L7_SLOTS], fy[L7_SLOTS], fz[L7_SLOTS];
#pragma unroll
      for (int s = 0; s < L7_SLOTS; s++) { fx[s] = X[pid[s]]; fy[s] = X[pitch + pid[s]]; fz[s] = X[2 * pitch + pid[s]]; }
#pragma unroll
      for (int s = 0; s < L7_SLOTS; s++) {
        if (in[s]) {
          stage_t* p = st + s * L7_STRIDE + lane * 3;
          p[0] = (stage_t)fx[s]; p[1] = (stage_t)fy[s]; p[2] = (stage_t)fz[s];
        }
      }
      __builtin_amdgcn_wave_barrier();
      // ---- ordered adds: every (leaf, sum) chain takes its next term, in input order
      const unsigned m_mine = my_len > w0 ? (my_len - w0 < 64u ? my_len - w0 : 64u) : 0u;
      const unsigned m_max = maxlen - w0 < 64u ? maxlen - w0 : 64u;
      const stage_t* mine = st + (slot < L7_SLOTS ? slot : 0) * L7_STRIDE;
      if (BATCHED) {
        // eight steps' LDS reads ahead of their adds; a chain that is over adds +0.0 (a no-op: no chain ever holds -0.0), selected, not
        // multiplied (stale LDS words may be anything)
        for (unsigned l0 = 0; l0 < m_max; l0 += L7_BATCH) {
          double u[L7_BATCH], v[L7_BATCH];
#pragma unroll
          for (int k = 0; k < L7_BATCH; k++) { u[k] = (double)mine[(l0 + k) * 3 + ia]; v[k] = (double)mine[(l0 + k) * 3 + (ib < 0 ? ia : ib)]; }
#pragma unroll
          for (int k = 0; k < L7_BATCH; k++) {
            const double t = u[k] * (ib < 0 ? 1.0 : v[k]);
            acc += (l0 + k < m_mine) ? t : 0.0;
          }
        }
      } else {
        for (unsigned l = 0; l < m_max; l++) {
          if (l < m_mine) {
            const double u = (double)mine[l * 3 + ia];
            const double v = ib < 0 ? 1.0 : (double)mine[l * 3 + ib];
            acc += u * v;                                                            // (x * 1.0 == x: the three plain sums take the same path)
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (slot < L7_SLOTS && id_s[0] >= 0) {
      // (id of the lane's own slot: select without dynamic indexing)
      int my_id = -1;
#pragma unroll
      for (int s = 0; s < L7_SLOTS; s++) my_id = slot == s ? id_s[s] : my_id;
      if (my_id >= 0) {
        sums[(size_t)(g.rec_off + my_id) * 9 + a] = acc;
        if (a == 0) vox_n[g.rec_off + my_id] = (int)my_len;
      }
    }
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 271;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t pitch = 65536;
  const bool by_length = argc > 4 ? atoi(argv[4]) != 0 : true;          // 0: leaves dealt in cell order (no ordering pass needed)
  const unsigned stretch = argc > 3 ? (unsigned)atoi(argv[3]) : 8u;   // a power of two
  const int cb = 20;                                     // bits of the cell index inside a key (the product passes its own)
  std::mt19937_64 rng(12345);
  // ---- synthetic sorted targets: runs of equal keys; searchable leaves (>= 6 points) get a seg_start entry
  std::vector<unsigned> keys((size_t)B * pitch), vals((size_t)B * pitch), seg_start, seg_len, order;
  std::vector<float> X((size_t)B * 3 * pitch);
  std::vector<GridDesc> gd(B);
  std::uniform_real_distribution<float> coord(-80.f, 80.f);
  std::exponential_distribution<double> tail(1.0 / 38.0);
  size_t total_leaves = 0, total_pts_in_leaves = 0;
  unsigned rpp = 0;
  std::vector<std::vector<unsigned>> starts(B), lens(B);
  for (int b = 0; b < B; b++) {
    for (size_t i = 0; i < 3 * pitch; i++) X[(size_t)b * 3 * pitch + i] = coord(rng);
    // point ids: a spinning lidar delivers the points of a cell as a few stretches of consecutive points (6.8 per stretch in the
    // benchmark's scans) -- stretches of 8 consecutive ids in random order (argv[3] = 1: single points, the worst case)
    std::vector<unsigned> perm(pitch), blocks(pitch / stretch);
    std::iota(blocks.begin(), blocks.end(), 0u);
    std::shuffle(blocks.begin(), blocks.end(), rng);
    for (size_t i = 0; i < pitch; i++) perm[i] = blocks[i / stretch] * stretch + (unsigned)(i % stretch);
    size_t j = 0;
    unsigned cell = 1;
    while (j < pitch) {
      // 30 % short runs (1..5 points: not searchable), the rest 6 + exponential(38), one in 60 a long one (ground near the sensor)
      unsigned len;
      const double u = std::uniform_real_distribution<double>(0, 1)(rng);
      if (u < 0.3) len = 1 + (unsigned)(rng() % 5);
      else if (u < 0.3 + 0.7 / 60) len = 300 + (unsigned)(rng() % 620);
      else len = 6 + (unsigned)tail(rng);
      len = (unsigned)std::min<size_t>(len, pitch - j);
      for (unsigned k = 0; k < len; k++) { keys[(size_t)b * pitch + j + k] = cell; vals[(size_t)b * pitch + j + k] = perm[j + k]; }
      if (len >= 6) { starts[b].push_back((unsigned)j); lens[b].push_back(len); total_pts_in_leaves += len; }
      j += len;
      cell += 1 + (unsigned)(rng() % 3);
    }
    total_leaves += starts[b].size();
    rpp = std::max<unsigned>(rpp, (unsigned)starts[b].size());
  }
  seg_start.assign((size_t)B * rpp, 0u); seg_len.assign((size_t)B * rpp, 0u); order.assign((size_t)B * rpp, 0u);
  for (int b = 0; b < B; b++) {
    memset(&gd[b], 0, sizeof(GridDesc));
    gd[b].status = GRID_OK;
    gd[b].n_voxels = (int)starts[b].size();
    gd[b].rec_off = (unsigned)b * rpp;
    std::vector<unsigned> idx(starts[b].size());
    std::iota(idx.begin(), idx.end(), 0u);
    if (by_length) std::stable_sort(idx.begin(), idx.end(), [&](unsigned p, unsigned q) { return lens[b][p] > lens[b][q]; });
    for (size_t k = 0; k < starts[b].size(); k++) {
      seg_start[(size_t)b * rpp + k] = starts[b][k];
      seg_len[(size_t)b * rpp + k] = lens[b][k];
      order[(size_t)b * rpp + k] = idx[k];
    }
  }
  printf("%d targets, %zu searchable leaves (%.0f per target), %.1f points per leaf, longest %u\n", B, total_leaves, (double)total_leaves / B,
         (double)total_pts_in_leaves / total_leaves, *std::max_element(seg_len.begin(), seg_len.end()));
  // ---- device buffers
  float* dX; unsigned *dK, *dV, *dS, *dL, *dO; GridDesc* dG; double *dSumA, *dSumB; int *dIdx, *dNA, *dNB;
  const size_t nrec = (size_t)B * rpp;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dK, keys.size() * 4)); CK(hipMalloc(&dV, vals.size() * 4));
  CK(hipMalloc(&dS, nrec * 4)); CK(hipMalloc(&dL, nrec * 4)); CK(hipMalloc(&dO, nrec * 4)); CK(hipMalloc(&dG, B * sizeof(GridDesc)));
  CK(hipMalloc(&dSumA, nrec * 9 * 8)); CK(hipMalloc(&dSumB, nrec * 9 * 8)); CK(hipMalloc(&dIdx, nrec * 4)); CK(hipMalloc(&dNA, nrec * 4)); CK(hipMalloc(&dNB, nrec * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dK, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dV, vals.data(), vals.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dS, seg_start.data(), nrec * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dL, seg_len.data(), nrec * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dO, order.data(), nrec * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dG, gd.data(), B * sizeof(GridDesc), hipMemcpyHostToDevice));
  CK(hipMemset(dSumA, 0xff, nrec * 9 * 8)); CK(hipMemset(dSumB, 0xee, nrec * 9 * 8)); CK(hipMemset(dNA, 0, nrec * 4)); CK(hipMemset(dNB, 0, nrec * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto launch) {
    launch();                                              // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us per launch\n", name, 1e3 * ms / reps);
  };
  // ---- the product kernel, launched as mi355_ndt.hip launches it
  const int lb = std::max(1, std::min((int)((rpp + LS_WAVES - 1) / LS_WAVES), 64));
  time_it("k_leafsum (product: wave per leaf)", [&] {
    k_leafsum<unsigned, false><<<xcd_grid(lb, B), 64 * LS_WAVES>>>(dX, pitch, dK, dV, dG, dS, dSumA, dIdx, dNA, cb, nullptr, lb, B);
  });
  // ---- the seven-leaves-per-wave kernel over a few workgroup counts per target
  const int groups = (int)((rpp + L7_SLOTS - 1) / L7_SLOTS);
  printf("k_leafsum7: %d waves per workgroup, staging %s, leaves %s\n", L7_WAVES, sizeof(stage_t) == 4 ? "f32" : "f64", by_length ? "by falling length" : "in cell order");
  for (int nx : {std::max(1, groups / (4 * L7_WAVES)), std::max(1, groups / (2 * L7_WAVES)), std::max(1, (groups + L7_WAVES - 1) / L7_WAVES)}) {
    char name[96];
    snprintf(name, sizeof name, "k_leafsum7 (7 leaves per wave), %d wg/target", nx);
    time_it(name, [&] { k_leafsum7<false><<<xcd_grid(nx, B), 64 * L7_WAVES>>>(dX, pitch, dV, dG, dS, dL, dO, dSumB, dNB, nx, B); });
    snprintf(name, sizeof name, "k_leafsum7, reads batched by %d, %d wg/target", L7_BATCH, nx);
    time_it(name, [&] { k_leafsum7<true><<<xcd_grid(nx, B), 64 * L7_WAVES>>>(dX, pitch, dV, dG, dS, dL, dO, dSumB, dNB, nx, B); });
  }
  // ---- same bits?
  std::vector<double> sa(nrec * 9), sb(nrec * 9);
  std::vector<int> na(nrec), nb(nrec);
  CK(hipMemcpy(sa.data(), dSumA, nrec * 9 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), dSumB, nrec * 9 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(na.data(), dNA, nrec * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(nb.data(), dNB, nrec * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, badn = 0, checked = 0, badcpu = 0;
  for (int b = 0; b < B; b++)
    for (int k = 0; k < gd[b].n_voxels; k++) {
      const size_t r = (size_t)b * rpp + k;
      if (memcmp(&sa[r * 9], &sb[r * 9], 72) != 0) bad++;
      if (na[r] != nb[r] || na[r] != (int)seg_len[r]) badn++;
      if ((r % 97) == 0) {                                 // a CPU loop over a sample of leaves: the definition both kernels restate
        double acc[9] = {0, 0, 0, 1, 0, 0, 1, 0, 1};
        for (unsigned l = 0; l < seg_len[r]; l++) {
          const unsigned pi = vals[(size_t)b * pitch + seg_start[r] + l];
          const double x = X[(size_t)b * 3 * pitch + pi], y = X[(size_t)b * 3 * pitch + pitch + pi], z = X[(size_t)b * 3 * pitch + 2 * pitch + pi];
          acc[0] += x; acc[1] += y; acc[2] += z; acc[3] += x * x; acc[4] += x * y; acc[5] += x * z; acc[6] += y * y; acc[7] += y * z; acc[8] += z * z;
        }
        checked++;
        if (memcmp(acc, &sb[r * 9], 72) != 0) badcpu++;
      }
    }
  printf("leaves with other bits than the product kernel: %zu; with another count: %zu; CPU sample: %zu of %zu differ\n", bad, badn, badcpu, checked);
  return (bad || badn || badcpu) ? 1 : 0;
}
