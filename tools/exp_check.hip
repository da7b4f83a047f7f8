// Compares the sweep's table-driven exp (ndtm::exp_f32arg) and the device library's (float)exp((double)a) with glibc's,
// over arguments shaped like impl2:581's (-d2 * q / 2 <= 0) plus a sprinkling of positive and special values.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ilv_slam_amd/csrc -Iinclude tools/exp_check.hip -o exp_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include "ndt_types.hpp"
#include "ndt_math.hpp"
__global__ void k(const float* a, float* t, float* o, int n) {
  __shared__ double tab[64];
  if (threadIdx.x < 64) tab[threadIdx.x] = ndtm::c_exp2_64[threadIdx.x];
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  t[i] = ndtm::exp_f32arg(a[i], tab);
  o[i] = (float)exp((double)a[i]);
}
int main() {
  const int n = 1 << 26;
  std::vector<float> a(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(0, 1);
  for (int i = 0; i < n; i++) {
    double r = u(g);
    a[i] = (float)(i % 16 == 0 ? 5.0 * u(g) : (i % 16 == 1 ? -60.0 - 60.0 * r : -20.0 * r * r));
  }
  const float sp[] = {0.f, -0.f, -1e-30f, 1e-30f, -87.3f, -88.f, -100.f, -103.9f, -104.f, -745.f, -1e30f, 88.7f, 89.f, 710.f, 1e30f, INFINITY, -INFINITY, NAN};
  for (size_t i = 0; i < sizeof sp / sizeof sp[0]; i++) a[i] = sp[i];
  float *da, *dt, *dof;
  hipMalloc(&da, n * 4); hipMalloc(&dt, n * 4); hipMalloc(&dof, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, dt, dof, n);
  std::vector<float> t(n), o(n);
  hipMemcpy(t.data(), dt, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dof, n * 4, hipMemcpyDeviceToHost);
  long bad_t = 0, bad_o = 0, bad_to = 0;
  for (int i = 0; i < n; i++) {
    float ref = (float)std::exp((double)a[i]);
    if (memcmp(&ref, &t[i], 4) && !(std::isnan(ref) && std::isnan(t[i]))) { if (bad_t < 10) printf("table  a=%a got %a want %a\n", a[i], t[i], ref); bad_t++; }
    if (memcmp(&ref, &o[i], 4) && !(std::isnan(ref) && std::isnan(o[i]))) { if (bad_o < 10) printf("ocml   a=%a got %a want %a\n", a[i], o[i], ref); bad_o++; }
    if (memcmp(&t[i], &o[i], 4) && !(std::isnan(o[i]) && std::isnan(t[i]))) bad_to++;
  }
  printf("n = %d: table vs glibc %ld mismatches, device-library vs glibc %ld, table vs device-library %ld\n", n, bad_t, bad_o, bad_to);
  return 0;
}
