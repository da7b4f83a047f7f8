// ndt_fast_list.hpp -- the kernel instantiations of the tolerance arithmetic (ORD = 2: MI355NDT_OPT_ARITH = 1; ndt_sweep.hpp, eval_hit_fast), compiled in a
// translation unit of their own (mi355_ndt_fast.hip) beside the other two; mi355_ndt.hip declares them `extern template`.  DIRECT1 / DIRECT7, both
// classes: the batch-mode sweep, the latency mode's fine sweeps, the one-launch align.  Kernels only: no device function crosses the units.
#pragma once
#include "ndt_ord1_list.hpp"      // (the argument-list macros)
#define NDT_FAST_KERNELS(X)                                                                                   \
  X __global__ void k_sweep<false, 1, 8, false, 2> NDT_SWEEP_ARGS_T;  X __global__ void k_sweep<true, 1, 8, false, 2> NDT_SWEEP_ARGS_T;   \
  X __global__ void k_sweep<false, 7, 8, false, 2> NDT_SWEEP_ARGS_T;  X __global__ void k_sweep<true, 7, 8, false, 2> NDT_SWEEP_ARGS_T;   \
  X __global__ void k_sweep<false, 1, 1, true, 2> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 1, 1, true, 2> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep<false, 1, 2, true, 2> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 1, 2, true, 2> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep<false, 7, 1, true, 2> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 7, 1, true, 2> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_sweep<false, 7, 2, true, 2> NDT_SWEEP_ARGS_T;   X __global__ void k_sweep<true, 7, 2, true, 2> NDT_SWEEP_ARGS_T;    \
  X __global__ void k_align_async<false, 1, 2> NDT_ASYNC_ARGS_T;  X __global__ void k_align_async<true, 1, 2> NDT_ASYNC_ARGS_T;           \
  X __global__ void k_align_async<false, 7, 2> NDT_ASYNC_ARGS_T;  X __global__ void k_align_async<true, 7, 2> NDT_ASYNC_ARGS_T;
