// tools/eval_rate.hip -- ground-truth cost of one 64-hit evaluation batch (eval_hit of the sweep) per SIMD,
// with operands streamed from L2-resident memory; block = 256 threads, `bpc` blocks per CU => waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/eval_rate.hip -o tools/eval_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool PCA>
__device__ __forceinline__ void eval_hit(const float u[3], const float r[3], const float C[9],
                                         const double d1, const float d2f, const double w, const bool ok_in, double acc[43]) {
  float y[3];
#pragma unroll
  for (int j = 0; j < 3; j++) y[j] = (u[0] * C[j] + u[1] * C[3 + j]) + u[2] * C[6 + j];
  const float qf = (u[0] * y[0] + u[1] * y[1]) + u[2] * y[2];
  const float e0 = (float)exp((double)((-d2f * qf) * 0.5f));                     // impl2:581
  float s_inc = (float)(-d1 * (double)e0);                                       // impl2:583
  const float e1 = d2f * e0;                                                     // impl2:585
  // impl2:588-589, branch-free: a rejected hit (or an idle lane, ok_in = false) multiplies every term by e = 0 and so
  // adds +0 to all 43 sums (all operands are finite here: dead voxels never enter the queue).
  const bool ok = ok_in && !(e1 > 1.f || e1 < 0.f || e1 != e1);
  float e = (float)((double)e1 * d1);                                            // impl2:592
  e = ok ? e : 0.f;
  s_inc = ok ? s_inc : 0.f;
  // CJ = c_inv4 * point_gradient4 (impl2:594): columns 0..2 are C itself
  float CJ[3][6];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    CJ[a][0] = C[a * 3 + 0]; CJ[a][1] = C[a * 3 + 1]; CJ[a][2] = C[a * 3 + 2];
    CJ[a][3] = C[a * 3 + 1] * (-r[2]) + C[a * 3 + 2] * r[1];
    CJ[a][4] = C[a * 3 + 0] * r[2] + C[a * 3 + 2] * (-r[0]);
    CJ[a][5] = C[a * 3 + 0] * (-r[1]) + C[a * 3 + 1] * r[0];
  }
  float v[6];
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] = (u[0] * CJ[0][k] + u[1] * CJ[1][k]) + u[2] * CJ[2][k];   // impl2:595
  // w * term: the product is a single rounding away from the reference's nested multiplies (both ~1e-16)
#define NDT_ACC(slot, val) do { if (PCA) acc[slot] = fma(w, (double)(val), acc[slot]); else acc[slot] += (double)(val); } while (0)
  NDT_ACC(0, s_inc);
#pragma unroll
  for (int k = 0; k < 6; k++) NDT_ACC(1 + k, e * v[k]);                                        // impl2:597
  // z_i[j] = y * Hp_block_i (impl2:607) -- nine non-zero entries (impl2:522-530)
  float Z[3][3];
  Z[0][0] = y[1] * (-r[1]) + y[2] * (-r[2]);
  Z[1][0] = y[0] * r[1];
  Z[2][0] = y[0] * r[2];
  Z[0][1] = y[1] * r[0];
  Z[1][1] = y[0] * (-r[0]) + y[2] * (-r[2]);
  Z[2][1] = y[1] * r[2];
  Z[0][2] = y[2] * r[0];
  Z[1][2] = y[2] * r[1];
  Z[2][2] = y[0] * (-r[0]) + y[1] * (-r[1]);
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      // JCJ[j][i] = (J^T CJ)(j,i) (impl2:601)
      float jcj;
      if (j < 3) jcj = CJ[j][i];
      else if (j == 3) jcj = (-r[2]) * CJ[1][i] + r[1] * CJ[2][i];
      else if (j == 4) jcj = r[2] * CJ[0][i] + (-r[0]) * CJ[2][i];
      else jcj = (-r[1]) * CJ[0][i] + r[0] * CJ[1][i];
      const float z = (i >= 3 && j >= 3) ? Z[i - 3][j - 3] : 0.f;
      const float h = e * ((((-d2f) * v[i]) * v[j] + z) + jcj);                                // impl2:611-613
      NDT_ACC(7 + i * 6 + j, h);
    }
  }
#undef NDT_ACC
}


template <int WPE>
__global__ void __launch_bounds__(256, WPE) mk(const float4* in, double* out, int iters, double d1, float d2f) {
  double acc[43];
  for (int a = 0; a < 43; a++) acc[a] = 0;
  const int lane = threadIdx.x;
  for (int i = 0; i < iters; i++) {
    const float4* p = in + ((size_t)((i * 7 + blockIdx.x) & 1023) * 256 + lane) * 4;
    const float4 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
    float u[3] = {p0.x, p0.y, p0.z}, r[3] = {p0.w, p1.x, p1.y}, C[9] = {p1.z, p1.w, p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z};
    eval_hit<false>(u, r, C, d1, d2f, 1.0, p3.w > 0.f, acc);
  }
  double s = 0;
  for (int a = 0; a < 43; a++) s += acc[a];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  size_t n = 1024 * 256 * 4;
  std::vector<float4> h(n);
  for (size_t i = 0; i < n; i++) h[i] = make_float4(0.1f + (i % 7) * 0.01f, 0.2f, -0.1f + (i % 5) * 0.02f, 1.0f);
  float4* d; hipMalloc(&d, n * sizeof(float4)); hipMemcpy(d, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  double* o; hipMalloc(&o, sizeof(double) * 256 * pr.multiProcessorCount * 8);
  const int iters = 4000;
  for (int bpc : {1, 2, 3}) {
    int grid = pr.multiProcessorCount * bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (bpc == 1) hipLaunchKernelGGL(mk<1>, dim3(grid), dim3(256), 0, 0, d, o, iters, -2.2, 0.43f);
      else if (bpc == 2) hipLaunchKernelGGL(mk<2>, dim3(grid), dim3(256), 0, 0, d, o, iters, -2.2, 0.43f);
      else hipLaunchKernelGGL(mk<3>, dim3(grid), dim3(256), 0, 0, d, o, iters, -2.2, 0.43f);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves/SIMD %d: %.1f ns per 64-hit batch per SIMD  (%.2f G hits/s chip-wide)\n", bpc, ms * 1e6 / ((double)iters * bpc),
           (double)iters * 256 * grid / (ms * 1e-3) / 1e9);
  }
  return 0;
}
