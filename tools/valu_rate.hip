// tools/valu_rate.hip -- measures wave64 VALU issue cost (cycles per instruction per SIMD) on the GPU it runs on.
// Used to price the derivative sweep's instruction mix; not part of the product.
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define BODY(name, asmstr, ...)                                                           \
  __global__ void __launch_bounds__(256) k_##name(float* out, int iters) {              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                        \
    for (int i = 0; i < iters; i++) {                                                     \
      REP8(asm volatile(asmstr ::: __VA_ARGS__);)                                                \
    }                                                                                     \
    if (a0 == -1.f) out[0] = a0 + a1 + a2 + a3;                                           \
  }

// each asm statement = 8 independent instructions on fixed registers v[40..71]
BODY(mul_f32, "v_mul_f32 v40, v40, v41\n v_mul_f32 v42, v42, v43\n v_mul_f32 v44, v44, v45\n v_mul_f32 v46, v46, v47\n v_mul_f32 v48, v48, v49\n v_mul_f32 v50, v50, v51\n v_mul_f32 v52, v52, v53\n v_mul_f32 v54, v54, v55", "v40","v42","v44","v46","v48","v50","v52","v54")
BODY(add_f32, "v_add_f32 v40, v40, v41\n v_add_f32 v42, v42, v43\n v_add_f32 v44, v44, v45\n v_add_f32 v46, v46, v47\n v_add_f32 v48, v48, v49\n v_add_f32 v50, v50, v51\n v_add_f32 v52, v52, v53\n v_add_f32 v54, v54, v55", "v40","v42","v44","v46","v48","v50","v52","v54")
BODY(fma_f32, "v_fma_f32 v40, v40, v41, v40\n v_fma_f32 v42, v42, v43, v42\n v_fma_f32 v44, v44, v45, v44\n v_fma_f32 v46, v46, v47, v46\n v_fma_f32 v48, v48, v49, v48\n v_fma_f32 v50, v50, v51, v50\n v_fma_f32 v52, v52, v53, v52\n v_fma_f32 v54, v54, v55, v54", "v40","v42","v44","v46","v48","v50","v52","v54")
BODY(pk_mul_f32, "v_pk_mul_f32 v[40:41], v[40:41], v[56:57]\n v_pk_mul_f32 v[42:43], v[42:43], v[56:57]\n v_pk_mul_f32 v[44:45], v[44:45], v[56:57]\n v_pk_mul_f32 v[46:47], v[46:47], v[56:57]\n v_pk_mul_f32 v[48:49], v[48:49], v[56:57]\n v_pk_mul_f32 v[50:51], v[50:51], v[56:57]\n v_pk_mul_f32 v[52:53], v[52:53], v[56:57]\n v_pk_mul_f32 v[54:55], v[54:55], v[56:57]", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")
BODY(pk_add_f32, "v_pk_add_f32 v[40:41], v[40:41], v[56:57]\n v_pk_add_f32 v[42:43], v[42:43], v[56:57]\n v_pk_add_f32 v[44:45], v[44:45], v[56:57]\n v_pk_add_f32 v[46:47], v[46:47], v[56:57]\n v_pk_add_f32 v[48:49], v[48:49], v[56:57]\n v_pk_add_f32 v[50:51], v[50:51], v[56:57]\n v_pk_add_f32 v[52:53], v[52:53], v[56:57]\n v_pk_add_f32 v[54:55], v[54:55], v[56:57]", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")
BODY(add_f64, "v_add_f64 v[40:41], v[40:41], v[56:57]\n v_add_f64 v[42:43], v[42:43], v[56:57]\n v_add_f64 v[44:45], v[44:45], v[56:57]\n v_add_f64 v[46:47], v[46:47], v[56:57]\n v_add_f64 v[48:49], v[48:49], v[56:57]\n v_add_f64 v[50:51], v[50:51], v[56:57]\n v_add_f64 v[52:53], v[52:53], v[56:57]\n v_add_f64 v[54:55], v[54:55], v[56:57]", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")
BODY(fma_f64, "v_fma_f64 v[40:41], v[40:41], v[56:57], v[40:41]\n v_fma_f64 v[42:43], v[42:43], v[56:57], v[42:43]\n v_fma_f64 v[44:45], v[44:45], v[56:57], v[44:45]\n v_fma_f64 v[46:47], v[46:47], v[56:57], v[46:47]\n v_fma_f64 v[48:49], v[48:49], v[56:57], v[48:49]\n v_fma_f64 v[50:51], v[50:51], v[56:57], v[50:51]\n v_fma_f64 v[52:53], v[52:53], v[56:57], v[52:53]\n v_fma_f64 v[54:55], v[54:55], v[56:57], v[54:55]", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")
BODY(cvt_f64_f32, "v_cvt_f64_f32 v[40:41], v58\n v_cvt_f64_f32 v[42:43], v59\n v_cvt_f64_f32 v[44:45], v58\n v_cvt_f64_f32 v[46:47], v59\n v_cvt_f64_f32 v[48:49], v58\n v_cvt_f64_f32 v[50:51], v59\n v_cvt_f64_f32 v[52:53], v58\n v_cvt_f64_f32 v[54:55], v59", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")
BODY(cndmask, "v_cndmask_b32 v40, v40, v41, vcc\n v_cndmask_b32 v42, v42, v43, vcc\n v_cndmask_b32 v44, v44, v45, vcc\n v_cndmask_b32 v46, v46, v47, vcc\n v_cndmask_b32 v48, v48, v49, vcc\n v_cndmask_b32 v50, v50, v51, vcc\n v_cndmask_b32 v52, v52, v53, vcc\n v_cndmask_b32 v54, v54, v55, vcc", "v40","v42","v44","v46","v48","v50","v52","v54")
BODY(mfma_f64_4x4, "v_mfma_f64_4x4x4_4b_f64 v[40:41], v[56:57], v[58:59], v[40:41]\n v_mfma_f64_4x4x4_4b_f64 v[42:43], v[56:57], v[58:59], v[42:43]\n v_mfma_f64_4x4x4_4b_f64 v[44:45], v[56:57], v[58:59], v[44:45]\n v_mfma_f64_4x4x4_4b_f64 v[46:47], v[56:57], v[58:59], v[46:47]\n v_mfma_f64_4x4x4_4b_f64 v[48:49], v[56:57], v[58:59], v[48:49]\n v_mfma_f64_4x4x4_4b_f64 v[50:51], v[56:57], v[58:59], v[50:51]\n v_mfma_f64_4x4x4_4b_f64 v[52:53], v[56:57], v[58:59], v[52:53]\n v_mfma_f64_4x4x4_4b_f64 v[54:55], v[56:57], v[58:59], v[54:55]", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")
// mixed: 4 f32 mul + 4 mfma f64 (do the pipes overlap?)
BODY(mix_mul_mfma, "v_mfma_f64_4x4x4_4b_f64 v[40:41], v[56:57], v[58:59], v[40:41]\n v_mul_f32 v60, v60, v61\n v_mul_f32 v62, v62, v63\n v_mul_f32 v64, v64, v65\n v_mul_f32 v66, v66, v67\n v_mfma_f64_4x4x4_4b_f64 v[42:43], v[56:57], v[58:59], v[42:43]\n v_mul_f32 v68, v68, v69\n v_mul_f32 v70, v70, v71", "v40","v41","v42","v43","v60","v62","v64","v66","v68","v70")

typedef void (*kfn)(float*, int);
struct K { const char* name; kfn f; int per_asm; };

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  float* d; hipMalloc(&d, 4);
  K ks[] = {{"v_mul_f32", k_mul_f32, 8}, {"v_add_f32", k_add_f32, 8}, {"v_fma_f32", k_fma_f32, 8}, {"v_pk_mul_f32", k_pk_mul_f32, 8},
            {"v_pk_add_f32", k_pk_add_f32, 8}, {"v_add_f64", k_add_f64, 8}, {"v_fma_f64", k_fma_f64, 8}, {"v_cvt_f64_f32", k_cvt_f64_f32, 8},
            {"v_cndmask_b32", k_cndmask, 8}, {"v_mfma_f64_4x4x4", k_mfma_f64_4x4, 8}, {"6 mul_f32 + 2 mfma_f64", k_mix_mul_mfma, 8}};
  const int iters = 20000;
  for (auto& k : ks) {
    for (int bpc : {1, 2, 4}) {   // blocks of 256 threads per CU => waves per SIMD
      int grid = p.multiProcessorCount * bpc;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, d, 10);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, d, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double inst_per_simd = (double)iters * 8 * k.per_asm * bpc;     // wave-instructions issued on one SIMD
      double ns_per_inst = ms * 1e6 / inst_per_simd;
      printf("%-26s waves/SIMD %d : %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", k.name, bpc, ns_per_inst, ns_per_inst * 2.4);
    }
  }
  return 0;
}
