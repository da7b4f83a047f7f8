// TEST HELPER (tests/test_exp_table_gpu.py): evaluates the sweep's table-driven exp (ndtm::exp_f32arg, ndt_math.hpp; the
// canonical choice for ndt_omp_impl2.hpp:581) on the device for every f32 argument in <in.f32> and writes the f32 results
// to <out.f32>.  The comparison with exp() of the host libm happens in the test.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ilv_slam_amd/csrc -Iinclude tests/hip/exp_table_check.hip -o tests/hip/exp_table_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ndt_types.hpp"
#include "ndt_math.hpp"

__global__ void k_exp(const float* __restrict__ a, float* __restrict__ out, size_t n) {
  __shared__ double tab[64];                       // staged exactly as k_sweep stages it
  if (threadIdx.x < 64) tab[threadIdx.x] = ndtm::c_exp2_64[threadIdx.x];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ndtm::exp_f32arg(a[i], tab);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f) / sizeof(float);
  fseek(f, 0, SEEK_SET);
  std::vector<float> a(n), o(n);
  if (fread(a.data(), sizeof(float), n, f) != n) return 2;
  fclose(f);
  float *da = nullptr, *dout = nullptr;
  CK(hipMalloc((void**)&da, n * sizeof(float)));
  CK(hipMalloc((void**)&dout, n * sizeof(float)));
  CK(hipMemcpy(da, a.data(), n * sizeof(float), hipMemcpyHostToDevice));
  k_exp<<<2048, 256>>>(da, dout, n);
  CK(hipGetLastError());
  CK(hipMemcpy(o.data(), dout, n * sizeof(float), hipMemcpyDeviceToHost));
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 2; }
  fwrite(o.data(), sizeof(float), n, f);
  fclose(f);
  printf("%zu\n", n);
  return 0;
}
