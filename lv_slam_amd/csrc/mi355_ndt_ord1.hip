// mi355_ndt_ord1.hip -- second translation unit of libmi355ndt.so: the sweep / align kernels instantiated for the other evaluation order of the
// three-term f32 sums (ORD = 1, MI355NDT_OPT_F32_SUM_ORDER; ndt_sweep.hpp).  Nothing but explicit instantiations: the host side, and every other
// kernel, live in mi355_ndt.hip, which declares these `extern template`.  Built side by side with it (__graft_entry__.build()).
#define NDT_SECOND_TU
#include <hip/hip_runtime.h>
#include "mi355_ndt.h"
#include "ndt_math.hpp"
#include "ndt_types.hpp"
#include "ndt_sweep.hpp"
#include "ndt_update.hpp"
#include "ndt_sweep_kd.hpp"
#include "ndt_async.hpp"
#include "ndt_ord1_list.hpp"
#define NDT_DEFINE template
NDT_ORD1_KERNELS(NDT_DEFINE)
