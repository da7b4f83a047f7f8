// ndt_ord1_list.hpp -- the kernel instantiations with the second f32 sum order (ORD = 1: MI355NDT_OPT_F32_SUM_ORDER = 1).  They are compiled
// in a translation unit of their own (mi355_ndt_ord1.hip) so that the two halves of the library build side by side; mi355_ndt.hip
// declares them `extern template`.  Kernels only: no device function crosses the two units.
#pragma once
#define NDT_SWEEP_ARGS_T (const float*, size_t, const PairState*, const GridDesc*, const BitWord*, const VoxelRec*, double*, int, const int*, SweepCtl*, SweepCtl*, \
                          SweepConst, const float*, const int*)
#define NDT_KD_ARGS_T (const float*, size_t, const PairState*, const GridDesc*, const BitWord*, const VoxelRec*, const float*, const int*, double*, int, const int*, \
                       SweepCtl*, SweepCtl*, SweepConst)
#define NDT_CTX_T const float*, size_t, PairState*, const GridDesc*, const BitWord*, const VoxelRec*, double*, const int*, unsigned*, const float*
#define NDT_ASYNC_ARGS_T (const AsyncTab*, int, int*, int, AsyncCtl*, SweepConst, unsigned long long*, double, double, int, int, unsigned, unsigned, int, NDT_CTX_T, NDT_CTX_T, NDT_CTX_T, NDT_CTX_T)
// X(prefix) is expanded once per kernel: prefix = `extern template` or `template`
#define NDT_ORD1_KERNELS(X)                                                                                   \
  X __global__ void k_sweep<false, 1, 8, false, 1> NDT_SWEEP_ARGS_T;  X __global__ void k_sweep<true, 1, 8, false, 1> NDT_SWEEP_ARGS_T;   \
  X __global__ void k_sweep<false, 7, 8, false, 1> NDT_SWEEP_ARGS_T;  X __global__ void k_sweep<true, 7, 8, false, 1> NDT_SWEEP_ARGS_T;   \
  X __global__ void k_sweep<false, 26, 8, false, 1> NDT_SWEEP_ARGS_T; X __global__ void k_sweep<true, 26, 8, false, 1> NDT_SWEEP_ARGS_T;  \
  X __global__ void k_sweep<false, 27, 8, false, 1> NDT_SWEEP_ARGS_T;                                                                      \
  X __global__ void k_sweep<false, 1, 1, true, 1> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 1, 1, true, 1> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep<false, 1, 2, true, 1> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 1, 2, true, 1> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep<false, 7, 1, true, 1> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 7, 1, true, 1> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep<false, 7, 2, true, 1> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 7, 2, true, 1> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep_pca_kd<1> NDT_KD_ARGS_T;                                                                                       \
  X __global__ void k_align_async<false, 1, 1> NDT_ASYNC_ARGS_T;  X __global__ void k_align_async<true, 1, 1> NDT_ASYNC_ARGS_T;           \
  X __global__ void k_align_async<false, 7, 1> NDT_ASYNC_ARGS_T;  X __global__ void k_align_async<true, 7, 1> NDT_ASYNC_ARGS_T;           \
  X __global__ void k_align_async<false, 26, 1> NDT_ASYNC_ARGS_T; X __global__ void k_align_async<true, 26, 1> NDT_ASYNC_ARGS_T;          \
  X __global__ void k_align_async<false, 27, 1> NDT_ASYNC_ARGS_T;
