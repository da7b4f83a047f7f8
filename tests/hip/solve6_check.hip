// TEST HELPER (tests/test_solve6_gpu.py): runs the Newton solve of the update kernel (ndtm::lu_solve6 / ndtm::svd_solve6, ndt_math.hpp;
// ndt_omp_impl2.hpp:138-140) on the device for every 6x6 system in <in.f64> (records of 42 doubles: H row-major, b) and writes, per
// system, 20 doubles to <out.f64>: x of svd_solve6, x of the one-lane route (ndtm::lu_solve6 when it accepts, else SVD), 1.0 if LU
// accepted, a zero, and x of newton_solve_side (ndt_update.hpp: the same route spread over seven lanes of the update kernel's second
// wave).  The comparison with numpy happens in the test.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ilv_slam_amd/csrc -Iinclude tests/hip/solve6_check.hip -o tests/hip/solve6_check
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ndt_types.hpp"
#include "ndt_math.hpp"
#include "ndt_update.hpp"

__global__ void k_solve(const double* __restrict__ in, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* H = in + (size_t)i * 42;
  const double* b = H + 36;
  double xs[6], xr[6];
  ndtm::svd_solve6(H, b, xs);
  bool fin = true;                                 // the route of newton_solve (ndt_update.hpp)
  for (int a = 0; a < 36; a++) fin = fin && isfinite(H[a]);
  for (int a = 0; a < 6; a++) fin = fin && isfinite(b[a]);
  const bool lu = fin && ndtm::lu_solve6(H, b, xr);
  if (!lu) ndtm::svd_solve6(H, b, xr);
  double* o = out + (size_t)i * 20;
  for (int a = 0; a < 6; a++) { o[a] = xs[a]; o[6 + a] = xr[a]; }
  o[12] = lu ? 1.0 : 0.0;
  o[13] = 0.0;
}

// one wave per system: newton_solve_side as k_update's second wave runs it
__global__ void k_solve_side(const double* __restrict__ in, double* __restrict__ out, PairState* st) {
  __shared__ double sol[SOL_WORDS];
  const int i = blockIdx.x;
  PairState& S = st[i];
  if (threadIdx.x == 0) {
    for (int a = 0; a < 36; a++) S.H[a] = in[(size_t)i * 42 + a];
    for (int a = 0; a < 6; a++) S.g[a] = -in[(size_t)i * 42 + 36 + a];
    sol[6] = 0.0;
  }
  __syncthreads();
  newton_solve_side(S, sol);
  __syncthreads();
  if (threadIdx.x < 6) out[(size_t)i * 20 + 14 + threadIdx.x] = sol[threadIdx.x];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f) / (42 * sizeof(double));
  fseek(f, 0, SEEK_SET);
  std::vector<double> a(n * 42), o(n * 20);
  if (fread(a.data(), sizeof(double), n * 42, f) != n * 42) return 2;
  fclose(f);
  double *da = nullptr, *dout = nullptr;
  CK(hipMalloc((void**)&da, a.size() * sizeof(double)));
  CK(hipMalloc((void**)&dout, o.size() * sizeof(double)));
  CK(hipMemcpy(da, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
  k_solve<<<(unsigned)((n + 63) / 64), 64>>>(da, dout, (int)n);
  CK(hipGetLastError());
  PairState* st = nullptr;
  CK(hipMalloc((void**)&st, n * sizeof(PairState)));
  CK(hipMemset(st, 0, n * sizeof(PairState)));
  k_solve_side<<<(unsigned)n, 64>>>(da, dout, st);
  CK(hipGetLastError());
  CK(hipMemcpy(o.data(), dout, o.size() * sizeof(double), hipMemcpyDeviceToHost));
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 2; }
  fwrite(o.data(), sizeof(double), o.size(), f);
  fclose(f);
  return 0;
}
