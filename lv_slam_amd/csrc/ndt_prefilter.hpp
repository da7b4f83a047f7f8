// ndt_prefilter.hpp -- distance filter + VoxelGrid centroid down-sampling.
#pragma once
#include "ndt_types.hpp"

// ------------------------------------------------------------------------------------ prefilter (upstream of the path)
// PrefilteringNodelet::distance_filter + downsample (src/lidar_odometry/prefiltering_nodelet.cpp:137-181, parameters of
// launch/dlo_kitti.launch:30-36): keep points with near < |p| < far (f32 norm compared as double), then pcl::VoxelGrid
// centroid downsample (PCL 1.8 voxel_grid.hpp applyFilter, CentroidPoint / AccumulatorXYZ: f32 sums, divided by the
// count), output in ascending voxel index.  Same binning + stable sort machinery as the NDT target build.
__global__ void __launch_bounds__(256) k_pf_flag(const float* __restrict__ X, size_t pitch, int n, int use_df, double dnear, double dfar,
                                                 unsigned char* keep, int* mm) {
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = X[i], y = X[pitch + i], z = X[2 * pitch + i];
    bool ok = true;
    if (use_df) {
      const double d = (double)sqrtf((x * x + y * y) + z * z);       // p.getVector3fMap().norm() (:168)
      ok = d > dnear && d < dfar;                                    // NaN fails both compares
    }
    ok = ok && finite3(x, y, z);                                     // VoxelGrid skips non-finite points (is_dense = false, :175)
    keep[i] = ok ? 1 : 0;
    if (!ok) continue;
    int ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
    mn[0] = min(mn[0], ox); mx[0] = max(mx[0], ox);
    mn[1] = min(mn[1], oy); mx[1] = max(mx[1], oy);
    mn[2] = min(mn[2], oz); mx[2] = max(mx[2], oz);
  }
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], o)); mx[a] = max(mx[a], __shfl_xor(mx[a], o)); }
    if ((threadIdx.x & 63) == 0) {
      if (mn[a] != INT_MAX) atomicMin(&mm[a], mn[a]);
      if (mx[a] != INT_MIN) atomicMax(&mm[3 + a], mx[a]);
    }
  }
}

struct PfGrid { int min_b[3], mul1, mul2, status; float inv_leaf; };   // status: 0 ok, 1 empty, 2 index overflow

__global__ void k_pf_grid(const int* __restrict__ mm, float leaf, PfGrid* out) {
  PfGrid g;
  memset(&g, 0, sizeof g);
  g.inv_leaf = 1.0f / leaf;
  if (mm[0] == INT_MAX) { g.status = 1; *out = g; return; }
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = ord2f(mm[a]); mx[a] = ord2f(mm[3 + a]); }
  if (grid_too_big((mx[0] - mn[0]) * g.inv_leaf, (mx[1] - mn[1]) * g.inv_leaf, (mx[2] - mn[2]) * g.inv_leaf)) { g.status = 2; *out = g; return; }   // "Leaf size is too small": output = input
  int maxb[3];
  for (int a = 0; a < 3; a++) { g.min_b[a] = (int)floorf(mn[a] * g.inv_leaf); maxb[a] = (int)floorf(mx[a] * g.inv_leaf); }
  g.mul1 = maxb[0] - g.min_b[0] + 1;
  g.mul2 = g.mul1 * (maxb[1] - g.min_b[1] + 1);
  *out = g;
}

__global__ void __launch_bounds__(256) k_pf_keys(const float* __restrict__ X, size_t pitch, int n, const unsigned char* __restrict__ keep,
                                                 const PfGrid* __restrict__ pg, unsigned* keys, unsigned* vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)pitch) return;
  const PfGrid g = *pg;
  unsigned cell = 0x7FFFFFFFu;
  if (i < n && keep[i] && g.status == 0) {
    const int i0 = (int)(floorf(X[i] * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(X[pitch + i] * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(X[2 * pitch + i] * g.inv_leaf) - (float)g.min_b[2]);
    cell = (unsigned)(i0 + i1 * g.mul1 + i2 * g.mul2);
  }
  keys[i] = cell;
  vals[i] = (unsigned)i;
}

// head of every occupied voxel's run (or, without down-sampling, every kept point)
__global__ void __launch_bounds__(256) k_pf_heads(const unsigned* __restrict__ keys, const unsigned char* __restrict__ keep, int n, size_t pitch,
                                                  int downsample, int* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)pitch) return;
  int f;
  if (downsample) f = keys[i] != 0x7FFFFFFFu && (i == 0 || keys[i - 1] != keys[i]);
  else f = i < n && keep[i];
  flag[i] = f;
}

__global__ void __launch_bounds__(256) k_pf_emit(const float* __restrict__ X, size_t pitch, const unsigned* __restrict__ keys,
                                                 const unsigned* __restrict__ vals, const int* __restrict__ flag, const int* __restrict__ pos,
                                                 int downsample, float* out, size_t out_pitch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)pitch || !flag[i]) return;
  float sx, sy, sz;
  if (downsample) {
    const unsigned key = keys[i];
    sx = sy = sz = 0.f;
    int cnt = 0;
    for (size_t j = i; j < pitch && keys[j] == key; j++) {           // AccumulatorXYZ: xyz += p (f32), input order
      const unsigned pi = vals[j];
      sx += X[pi]; sy += X[pitch + pi]; sz += X[2 * pitch + pi];
      cnt++;
    }
    const float fn = (float)cnt;
    sx /= fn; sy /= fn; sz /= fn;                                    // xyz / n
  } else {
    sx = X[i]; sy = X[pitch + i]; sz = X[2 * pitch + i];
  }
  const int o = pos[i];
  out[o] = sx; out[out_pitch + o] = sy; out[2 * out_pitch + o] = sz;
}


// Exclusive prefix sum of n ints (the emit positions of the kept points / voxel heads): block totals of 4096-element chunks,
// one block scanning those totals, then every chunk scans itself again on top of its offset.
#define PF_SCAN_CHUNK 4096
__global__ void __launch_bounds__(1024) k_pf_scan_totals(const int* __restrict__ in, size_t n, unsigned* totals) {
  __shared__ unsigned sm[17];
  const size_t i0 = (size_t)blockIdx.x * PF_SCAN_CHUNK + (size_t)threadIdx.x * 4;
  unsigned v = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) if (i0 + u < n) v += (unsigned)in[i0 + u];
  unsigned tot;
  (void)block_exscan<1024>(v, &tot, sm);
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_pf_scan_offsets(unsigned* totals, int nchunks) {
  __shared__ unsigned sm[17];
  unsigned base = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 1024) {
    const int c = c0 + threadIdx.x;
    const unsigned v = c < nchunks ? totals[c] : 0u;
    unsigned tot;
    const unsigned ex = block_exscan<1024>(v, &tot, sm);
    if (c < nchunks) totals[c] = base + ex;
    base += tot;
  }
}
__global__ void __launch_bounds__(1024) k_pf_scan_apply(const int* __restrict__ in, size_t n, const unsigned* __restrict__ offs, int* out) {
  __shared__ unsigned sm[17];
  const size_t i0 = (size_t)blockIdx.x * PF_SCAN_CHUNK + (size_t)threadIdx.x * 4;
  unsigned e[4], v = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) { e[u] = i0 + u < n ? (unsigned)in[i0 + u] : 0u; v += e[u]; }
  unsigned tot;
  unsigned ex = block_exscan<1024>(v, &tot, sm) + offs[blockIdx.x];
#pragma unroll
  for (int u = 0; u < 4; u++) { if (i0 + u < n) out[i0 + u] = (int)ex; ex += e[u]; }
}
