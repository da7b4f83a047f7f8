// mi355_ndt_fast.hip -- third translation unit of libmi355ndt.so: the sweep / align kernels instantiated for the tolerance arithmetic
// (ORD = 2, MI355NDT_OPT_ARITH = 1; ndt_sweep.hpp: eval_hit_fast).  Nothing but explicit instantiations: the host side, and every other
// kernel, live in mi355_ndt.hip, which declares these `extern template`.  Built side by side with the other two (__graft_entry__.build()),
// with the same flags -- -ffp-contract=off included: the fused multiply-adds of this arithmetic are written out (fmaf), the point transform
// and the cell lookup it shares with the exact kernels stay uncontracted.
#define NDT_SECOND_TU
#include <hip/hip_runtime.h>
#include "mi355_ndt.h"
#include "ndt_math.hpp"
#include "ndt_types.hpp"
#include "ndt_sweep.hpp"
#include "ndt_update.hpp"
#include "ndt_sweep_kd.hpp"
#include "ndt_async.hpp"
#include "ndt_fast_list.hpp"
#define NDT_DEFINE template
NDT_FAST_KERNELS(NDT_DEFINE)
