// Prints what v_permlane32_swap / v_permlane16_swap (gfx950) do to two lane-indexed registers; the sweep's
// reduce-scatter relies on: 32_swap -> {[a(0..31), b(0..31)], [a(32..63), b(32..63)]}; 16_swap -> {[a.r0, b.r0, a.r2, b.r2], [a.r1, b.r1, a.r3, b.r3]}.
// build: hipcc --offload-arch=gfx950 -O2 tools/permlane_swap.hip -o /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  u2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y; o[128 + threadIdx.x] = q.x; o[192 + threadIdx.x] = q.y;
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[4] = {"32swap.x", "32swap.y", "16swap.x", "16swap.y"};
  for (int r = 0; r < 4; r++) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 8) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
