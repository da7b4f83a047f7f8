// What does a device-scope fence cost inside a kernel on gfx950, and does it evict L2-resident data?  Each wave repeatedly sums a
// 64 KB slice of a 16 MB array (L2/MALL resident after the first pass); variants: no fence, acquire fence, release fence,
// seq_cst __threadfence() per iteration.  Prints ns per iteration.
// build: hipcc --offload-arch=gfx950 -O3 tools/fence_cost.hip -o fence_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(const float* __restrict__ a, float* out, int iters, int* flag) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const float* p = a + (size_t)(wave % 256) * 16384;       // 64 KB per wave, 16 MB total
  float s = 0;
  for (int it = 0; it < iters; it++) {
    for (int i = lane; i < 16384; i += 64 * 8) {
#pragma unroll
      for (int u = 0; u < 8; u++) s += p[i + u * 64];
    }
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (MODE == 3) __threadfence();
    if (MODE == 4) { if (lane == 0) atomicAdd(flag, 1); __threadfence(); }
  }
  if (s == 123.456f) out[0] = s;
}
int main() {
  float *a, *o; int* f;
  hipMalloc(&a, 16 << 20); hipMalloc(&o, 4); hipMalloc(&f, 4);
  hipMemset(a, 0, 16 << 20); hipMemset(f, 0, 4);
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* nm[5] = {"no fence", "acquire(agent)", "release(agent)", "__threadfence()", "atomicAdd + __threadfence()"};
  for (int m = 0; m < 5; m++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (m == 0) k<0><<<512, 256>>>(a, o, iters, f);
      if (m == 1) k<1><<<512, 256>>>(a, o, iters, f);
      if (m == 2) k<2><<<512, 256>>>(a, o, iters, f);
      if (m == 3) k<3><<<512, 256>>>(a, o, iters, f);
      if (m == 4) k<4><<<512, 256>>>(a, o, iters, f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("%-30s %8.1f ns per iteration (64 KB summed per wave per iteration)\n", nm[m], ms * 1e6 / iters);
    }
  }
  return 0;
}
