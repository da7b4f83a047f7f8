// mi355_ndt.hip -- MI355X (gfx950) NDT scan-matching engine behind the C-ABI of include/mi355_ndt.h.
//
// What runs where (all on the GPU; the host only enqueues):
//   target build  : k_minmax -> k_griddesc -> radix sort (cell from the points, input order) -> k_mark (+ run heads)
//                   -> k_rank (+ run starts by voxel id) -> k_leafsum -> k_voxels          (VoxelGridCovariance::applyFilter,
//                   include/ndt_omp/voxel_grid_covariance_omp_impl.hpp:48-370)
//   align         : k_init_state -> k_sweep -> [k_update -> k_sweep]*      (computeTransformation +
//                   computeDerivatives + computeStepLengthMT, include/ndt_omp/ndt_omp_impl2.hpp:87-188, 196-305, 841-1003;
//                   step_size <= eps/2 only: [k_update -> k_hessian -> k_update -> k_sweep]*, impl2:622-714, 920-1000)
// Build: __graft_entry__.build() compiles this file and mi355_ndt_ord1.hip (the kernel instantiations of the second f32 sum order) side by
// side and links them; -DNDT_SINGLE_TU builds everything from this file alone.
// Kernels live in ndt_build.hpp / ndt_sweep.hpp / ndt_update.hpp / ndt_hessian.hpp / ndt_fitness.hpp / ndt_prefilter.hpp; this file is
// the host side of the C-ABI (one translation unit).  Data layout in HBM: DESIGN.md.  Built with -ffp-contract=off: every f32/f64 step of the
// reference recipe is a separately rounded operation.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>
#include <mutex>
#include <atomic>
#include <thread>
#include <sched.h>
#include <fstream>
#include <chrono>

#include "mi355_ndt.h"
#include "ndt_math.hpp"
#include "ndt_types.hpp"
#include "ndt_build.hpp"
#include "ndt_segsort.hpp"
#include "ndt_sweep.hpp"
#include "ndt_update.hpp"
#include "ndt_hessian.hpp"
#include "ndt_sweep_kd.hpp"
#include "ndt_fitness.hpp"
#include "ndt_prefilter.hpp"
#include "ndt_sequence.hpp"
#include "ndt_async.hpp"
#ifndef NDT_SINGLE_TU      // the ORD = 1 instantiations come from mi355_ndt_ord1.hip (built side by side with this file)
#include "ndt_ord1_list.hpp"
#include "ndt_fast_list.hpp"
#define NDT_DECLARE extern template
NDT_ORD1_KERNELS(NDT_DECLARE)
NDT_FAST_KERNELS(NDT_DECLARE)
#endif


// ------------------------------------------------------------------------------------ host side
struct mi355ndt_handle;
struct mi355ndt_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  mi355ndt_params prm;
  std::string err;

  int n_pairs = 0, cap_pairs = 0;
  size_t tgt_pitch = 0, src_pitch = 0;          // geometry in use
  size_t own_tgt_pitch = 0, own_src_pitch = 0;  // geometry of the owned buffers
  int own_tgt_pairs = 0, own_src_pairs = 0;
  float *d_tgt_own = nullptr, *d_src_own = nullptr;
  const float *d_tgt = nullptr, *d_src = nullptr;
  int *d_tgt_cnt = nullptr, *d_src_cnt = nullptr;
  std::vector<int> h_tgt_cnt, h_src_cnt;
  std::vector<int> up_tgt_cnt, up_src_cnt;        // what d_tgt_cnt / d_src_cnt currently hold (uploads are skipped when unchanged)
  bool targets_built = false, have_target = false, have_source = false;
  bool aligned_once = false;                      // d_state / d_results hold the outcome of an align of the CURRENT batch
  bool icov64_built = false;                      // ... and the f64 inverse covariances computeHessian reads (live More-Thuente)
  bool cent_built = false;                        // last target build also produced the f32 leaf centroids (KDTREE mode)
  float grid_resolution = 0.f;                    // leaf size the resident grids were built with (setResolution without a source keeps them: ndt_omp.h:126-136)

  // build workspace
  unsigned* d_minmax = nullptr;                  // a slice of d_word_off's allocation (zeroed together before every build)
  GridDesc* d_grid = nullptr;
  unsigned *d_nwords = nullptr, *d_word_off = nullptr;
  unsigned *d_keys_a = nullptr, *d_keys_b = nullptr;      // cell key per target point: unsorted / sorted (segment-local radix sort)
  unsigned *d_vals_a = nullptr, *d_vals_b = nullptr;
  size_t keys_cap = 0;
  BitWord* d_words = nullptr; size_t words_cap = 0;
  VoxelRec* d_recs = nullptr; int *d_vox_idx = nullptr, *d_vox_n = nullptr;
  unsigned* d_seg_start = nullptr; double* d_sums = nullptr;
  unsigned *d_heads = nullptr, *d_head_cnt = nullptr; size_t heads_cap = 0, head_cnt_cap = 0;   // k_mark's run heads per slice
  float* d_cent = nullptr; double* d_icov64 = nullptr; size_t icov64_cap = 0;
  int* d_kdw = nullptr; size_t kdw_cap = 0; bool kdw_built = false;   // per-leaf weights for ndt_pca + KDTREE (dead leaves included)
  float4* d_sorted = nullptr; size_t sorted_cap = 0; bool leaf_sorted = false;   // MI355NDT_LEAF_SORTED: the sorted order as points (k_sorted_points)
  unsigned* d_rs_hist = nullptr; unsigned* d_rs_offs = nullptr; size_t rs_cap = 0;   // segmented radix sort: tile histograms / offsets
  unsigned *d_cstart = nullptr, *d_cend = nullptr; size_t cell_cap = 0; bool cells_ready = false; int last_cb = 0;
  double* d_fit = nullptr; size_t fit_cap = 0;
  // prefilter workspace
  float *d_pf_in = nullptr, *d_pf_out = nullptr; unsigned char* d_pf_keep = nullptr; unsigned *d_pf_keys = nullptr, *d_pf_vals = nullptr;
  int *d_pf_flag = nullptr, *d_pf_pos = nullptr, *d_pf_mm = nullptr; PfGrid* d_pf_grid = nullptr; void* d_pf_tmp = nullptr;
  size_t pf_cap = 0, pf_tmp_bytes = 0; int pf_count = 0; size_t pf_pitch = 0;
  float last_final[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
  size_t recs_per_pair = 0, recs_cap = 0;
  unsigned* h_pin_u = nullptr;                  // pinned scratch (2 unsigned)

  // align workspace
  PairState* d_state = nullptr;
  double* d_partials = nullptr; size_t partials_cap = 0;
  int chunks_per_pair = 0;
  int rows_per_pair = 0, pts_per_chunk = CHUNK_PTS;   // stored partial rows per pair / points covered by one chunk of k_update's tree
  int items_per_pair = 0;                         // sweep work items per pair (= rows_per_pair in batch mode, 4 x rows_per_pair in latency mode)
  bool async_force = false;                       // MI355NDT_OPT_ASYNC_ALIGN = 2 / MI355NDT_ASYNC=2: the one-launch align also for batches smaller than the resident waves (tests, fuzzing)
  bool async_align = true;                        // MI355NDT_OPT_ASYNC_ALIGN: batch aligns as ONE persistent launch (ndt_async.hpp); MI355NDT_ASYNC=0 turns it off
  int* d_ring = nullptr; size_t ring_cap_total = 0; unsigned* d_arrived = nullptr; size_t arrived_cap = 0; AsyncCtl* d_actl = nullptr;
  AsyncCtl* h_pin_actl = nullptr;
  AsyncTab* d_atab = nullptr;                     // the launch's context table (ndt_async.hpp)
  unsigned debug_abort_pos = 0xFFFFFFFFu;         // MI355NDT_OPT_DEBUG_ASYNC_ABORT (test hook): the wave that claims this position of ring 0 gives up
  unsigned debug_ring_mask = 0xFFu;               // MI355NDT_OPT_DEBUG_ASYNC_RINGS (test hook): rings whose workgroups take part
  int arith = 0;                                  // MI355NDT_OPT_ARITH: 0 = the reference recipe's arithmetic, one rounding per operation; 1 = tolerance arithmetic (ndt_sweep.hpp: eval_hit_fast)
  VoxelRecF* d_recs_fast = nullptr; size_t recs_fast_cap = 0; bool recs_fast_built = false;   // ... and the records its sweeps read (k_voxels writes them beside d_recs)
  int f32_sum_order = 0;                          // MI355NDT_OPT_F32_SUM_ORDER: 0 = (t0 + t1) + t2 (canonical), 1 = (t0 + t2) + t1 (Eigen 3.3 SSE predux pairing)
  double gauss_last[3] = {0, 0, 0};               // gauss_d1_/d2_/d3_ as the constructor / the last computeTransformation left them (calculateScore reads them)
  float* d_score_pts = nullptr; size_t score_pts_cap = 0; double* d_score_part = nullptr; size_t score_part_cap = 0;   // calculateScore workspace
  bool latency_mode = false;                      // mi355ndt_set_latency_mode
  bool seq_running = false;                       // inside mi355ndt_sequence_run
  int fine_it = 0;                                // 0: batch-mode sweep items (512 points); 1 / 2: fine items of fine_it * 64 points (latency mode)
  int fine_tiles = 2;                             // MI355NDT_FINE_TILES overrides (tuning runs)
  int* d_grid_of = nullptr; size_t grid_of_cap = 0;   // sequence mode: grid index per pair
  const int* d_grid_of_use = nullptr;             // what the sweeps are given: d_grid_of inside mi355ndt_sequence_run, else null (pair b -> grid b)
  SeqState* d_seq = nullptr; mi355ndt_seq_frame* d_seq_out = nullptr; double* d_stamps = nullptr; size_t seq_cap = 0;
  volatile int* h_seq_flags = nullptr; int* d_seq_flags = nullptr;   // mapped pinned: [0] = run finished, [1] = update launches executed
  float* d_guess = nullptr;
  float* h_pin_guess = nullptr;                   // pinned staging copy of the caller's guesses (no sync needed after the upload)
  mi355ndt_result* d_results = nullptr;
  int* d_active = nullptr;                      // per-round active counters
  int* d_active_list = nullptr;                 // pairs taking part in the next sweep (compacted by k_update)
  SweepCtl* d_ctl = nullptr;                      // two control blocks: the sweep reading one zeroes the other for the next round
  int ctl_idx = 0;                                // block the NEXT sweep reads (k_init_state / k_update fill it)
  int n_cu = 256;
  int dyn_shift = -1;                             // < 0: per search mode (make_sweep_const); MI355NDT_SWEEP_DYN_SHIFT overrides (tuning runs)
  int* h_pin_active = nullptr;
  hipEvent_t ev_burst[2] = {nullptr, nullptr};   // one per in-flight burst of align rounds
  unsigned long long* d_hits = nullptr;         // (point,voxel) evaluations, all sweeps
  float* d_hook = nullptr;                      // 16 + 9 floats, 6 doubles
  float* d_aligned = nullptr; size_t aligned_cap = 0;
  float* h_pin_aligned = nullptr; size_t pin_aligned_cap = 0;   // pinned landing buffer of get_aligned
  // host-cloud uploads: a ring of pinned staging slots, a copy stream of its own, a device staging buffer per slot.  The caller's
  // records are compacted to x,y,z into a slot (the only CPU work), the slot goes over PCIe asynchronously and a small kernel
  // spreads it into the SoA rows; the next call stages the next cloud while this one is still in flight.
  struct UpSlot { float* h = nullptr; float* d = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false, filling = false; };
  static constexpr int UP_SLOTS = 12;             // (a slot grows to the largest transfer it has carried: up to UP_GROUP_MAX clouds = 12.6 MB of 65,536-point clouds)
  UpSlot up[UP_SLOTS];
  int up_next = 0;
  static constexpr int UP_STREAMS = 4;            // an upload rides copy stream (pair + 2 * side) % UP_STREAMS: per-transfer latencies of the SDMA queues
                                                  // overlap across pairs, uploads into the same rows stay ordered
  hipStream_t copy_stream[UP_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_uploads[UP_STREAMS] = {nullptr, nullptr, nullptr, nullptr}, ev_compute = nullptr;   // copy streams -> compute stream, compute stream -> copy streams
  bool uploads_pending = false;
  std::mutex up_mtx;                              // batch_set_target / batch_set_source may be called from several threads (distinct pairs)

  // asynchronous target build (stream mode: the engine is one batch context of a parent handle).  A build normally waits for two words
  // from the device -- the bitmap words of all grids (pool size) and the largest grid (sort key width); with a PLAN from earlier builds of
  // the stream it does not: it sorts plan_cb key bits, clears plan_words pool words, and k_build_check turns every grid of the batch into
  // "no grid" (and raises its flag in d_bstat) should the batch not fit the plan -- the parent then re-runs that batch synchronously and learns.
  bool async_build = false;
  int plan_cb = 0; size_t plan_words = 0;
  unsigned* d_bstat = nullptr;                    // (not owned: the parent's CtxStat of this context) [1] total words, [2] largest grid, [3] plan exceeded
  bool counts_preloaded = false;                  // the parent has put this batch's point counts (and guesses) on the device already
  bool build_stamped = false;                    // stream mode + profiling: build times come from stamps in the launch's status slot, not from events
  bool word_off_cleared = false;                 // stream mode: k_stream_inputs has cleared d_word_off for the next build (no fill)
  size_t last_total_words = 0;                    // of the last synchronous build

  // ---- stream mode (mi355ndt_stream_*, the parent handle): n_contexts batches resident, one persistent launch per submitted batch,
  // the stragglers of a launch carried into the next one (ndt_async.hpp)
  struct StreamCtx {
    mi355ndt_handle* e = nullptr;                 // the context's own engine: bound clouds, grids, pair states (runs on the parent's stream)
    long long batch_id = -1; int n_pairs = 0;
    bool busy = false;                            // submitted, not yet collected
    bool redo = false;                            // collect re-runs it synchronously (its launch gave up)
    bool done_sync = false;                       // processed synchronously inside submit (configuration the one-launch align does not serve)
    long long launch = -1;                        // the launch that started it
    // a batch's small inputs -- target counts, source counts, guesses -- travel as ONE copy: pinned staging block -> device block, into which
    // the context engine's d_tgt_cnt / d_src_cnt / d_guess point
    int* h_in = nullptr; int* d_in = nullptr; size_t in_bytes = 0; unsigned* h_in_dev = nullptr;   // (h_in is mapped: the device reads it itself)
    void *own_tgt_cnt = nullptr, *own_src_cnt = nullptr, *own_guess = nullptr;   // the engine's own arrays (put back before it is destroyed)
    // results: MAPPED host memory -- a pair's result record is written there by the updater that finalises it (posted PCIe writes), no copy
    mi355ndt_result* h_res = nullptr; mi355ndt_result* d_res_map = nullptr;
    std::vector<float> guesses;                   // (kept for a synchronous re-run)
    PoseRecord* d_pose = nullptr; int pose_cap = 0, pose_base = 0, pose_stride = 1;   // mi355ndt_stream_pose_records (this batch's gather block)
  };
  bool stream_on = false, s_sync_only = false, s_drop_carry = true;
  int s_nctx = 0, s_max_pairs = 0, s_items = 0, s_ring_cap = 0, s_thresh = 0;
  size_t s_max_tgt = 0, s_max_src = 0;            // what mi355ndt_stream_begin was told (mi355ndt_stream_submit_host sizes the contexts' own cloud buffers with it)
  int s_thresh_opt = -1;                          // MI355NDT_OPT_STREAM_THRESHOLD
  int s_reserve_opt = -1;                         // MI355NDT_OPT_STREAM_RESERVE
  int s_plan_cb = 0; size_t s_plan_words = 0;
  void* s_pose_next = nullptr; size_t s_pose_cap_next = 0; int s_pose_base_next = 0, s_pose_stride_next = 1;   // apply to the next submit
  StreamCtx sctx[ASYNC_MAX_CTX];
  long long s_next_id = 0, s_launches = 0, s_counted = 0;
  long long s_recovered_upto = -1;                // launches up to this one have had their abort handled (stream_recover runs once per aborted launch, not once per collect that walks past its slot)
  AsyncCtl* d_sctl = nullptr;                     // two control blocks: a launch reads the hand-over list of the previous one
  int* d_sring = nullptr;
  CtxStat* d_sstat = nullptr;                     // per context: pairs finalised, sizes and verdict of its last planned build
  static constexpr int S_EV = 16;
  // Build under the launch: with s_reserve_wg > 0 the contexts' engines run their builds on s_build_stream, the persistent launches leave
  // that many workgroup slots free, and events order  launch j-2 done -> build of batch j -> launch j
  hipStream_t s_build_stream = nullptr; int s_reserve_wg = 0, s_launch_slots = 0;
  hipEvent_t s_ev_built[ASYNC_MAX_CTX] = {}; hipEvent_t s_ev_launched[S_EV] = {}; hipEvent_t s_ev_prepared[S_EV] = {}; bool s_prep_first = true;
  volatile StreamStatus* h_sstatus = nullptr; StreamStatus* d_sstatus = nullptr;   // mapped ring of per-launch status slots (k_stream_status)

  // profiling
  bool prof = false;
  mi355ndt_profile P{};
  struct EvSpan { hipEvent_t first, second; bool first_shared; };   // first_shared: `first` is the previous span's `second`
  std::vector<EvSpan> ev_sweep, ev_update, ev_build;
  hipEvent_t ev_last = nullptr;                   // end event of the span just closed, reusable as the next span's begin while
  bool ev_last_fresh = false;                     // nothing else has been enqueued on the stream since
  std::vector<hipEvent_t> ev_pool;                // idle timing events (filled by mi355ndt_profile_enable)
  size_t ev_pool_target = 4096;
};

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return MI355NDT_ERR_HIP;                                                                   \
    }                                                                                            \
  } while (0)

// several kernels carry the pair index in grid.y (HIP limit 65535)
#define MAX_PAIRS 65535
#ifndef UP_GROUP_PAIRS
#define UP_GROUP_PAIRS 8          // pair slots per upload group of mi355ndt_batch_set_clouds: their targets and sources travel as ONE transfer (measured, 271-pair
#endif                            // batches of 32-byte records streamed: 2 / 4 / 8 pairs per transfer = 20.2 / 21.4 / 22.2 k registrations/s; one cloud per transfer: 15.4 k)
#define UP_GROUP_MAX   (2 * UP_GROUP_PAIRS)
static_assert(UP_GROUP_MAX <= (int)(sizeof(DeintTab::e) / sizeof(DeintTab::e[0])), "k_deinterleave_multi's table");
// between mi355ndt_stream_begin and mi355ndt_stream_end the handle's batches belong to the stream: the other entry points refuse
#define NOT_IN_STREAM(h) do { if ((h)->stream_on) { (h)->err = "the handle is in stream mode (mi355ndt_stream_begin): call mi355ndt_stream_end first"; return MI355NDT_ERR_STATE; } } while (0)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}
static int ceil_log2(unsigned v) { int b = 0; while ((1u << b) < v) b++; return b; }

template <typename T>
static hipError_t grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; }
  hipError_t e = hipMalloc((void**)&p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  else cap = 0;
  return e;
}

static void build_offsets(int mode, SweepConst& sc) {
  if (mode == MI355NDT_DIRECT1) { sc.K = 1; sc.table = 0; }
  else if (mode == MI355NDT_DIRECT7) { sc.K = 7; sc.table = 1; }
  else if (mode == MI355NDT_DIRECT26) { sc.K = 26; sc.table = 2; }
  else { sc.K = 27; sc.table = 2; }            // KDTREE: 27-cell block + centroid radius test
}

static void gauss_constants3(double outlier_ratio, float resolution, double d[3]) {
  // ndt_omp_impl2.hpp:93-100 (and the constructor, impl2:70-76)
  double c1 = 10 * (1 - outlier_ratio);
  double c2 = outlier_ratio / pow((double)resolution, 3);
  d[2] = -log(c2);
  d[0] = -log(c1 + c2) - d[2];
  d[1] = -2 * log((-log(c1 * exp(-0.5) + c2) - d[2]) / d[0]);
}
static void gauss_constants(const mi355ndt_params& p, double& d1, double& d2) {
  double d[3];
  gauss_constants3(p.outlier_ratio, p.resolution, d);
  d1 = d[0]; d2 = d[1];
}

static int check_params(const mi355ndt_params& p) {
  if (!(p.resolution > 0) || !std::isfinite(p.resolution)) return MI355NDT_ERR_BAD_ARG;
  if (p.neighbor_mode < 0 || p.neighbor_mode > 3) return MI355NDT_ERR_BAD_ARG;
  if (p.variant < 0 || p.variant > 1) return MI355NDT_ERR_BAD_ARG;
  if (p.min_points_per_voxel < 1) return MI355NDT_ERR_BAD_ARG;
  if (p.max_iterations < 0) return MI355NDT_ERR_BAD_ARG;
  return MI355NDT_OK;
}

// impl2:888: the More-Thuente loop (and computeHessian after it) runs iff !(step_max - step_min > 0), step_min = eps/2
static bool mt_is_live(const mi355ndt_params& p) { return !((p.step_size - p.trans_epsilon / 2) > 0); }
// MI355NDT_OPT_ARITH = 1 is served for DIRECT1 / DIRECT7 with the dead More-Thuente loop (every configuration lv_slam ships); every other configuration
// ignores the option altogether: exact kernels, ordered leaf sums, the exact records alone
static bool fast_served(const mi355ndt_handle* h) {
  return h->arith == 1 && (h->prm.neighbor_mode == MI355NDT_DIRECT1 || h->prm.neighbor_mode == MI355NDT_DIRECT7) && !mt_is_live(h->prm);
}

extern "C" {

const char* mi355ndt_version(void) { return "mi355ndt 0.1 (gfx950)"; }

int mi355ndt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// NUMA node of the host CPUs closest to `device` (-1: unknown) -- staging threads and the clouds they read belong there
int mi355ndt_host_numa_node(int device) {
  int node = -1;
  if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, device) != hipSuccess) {
    (void)hipGetLastError();                      // not every runtime answers this attribute: leave no sticky error behind
    // fall back to the PCI device's sysfs entry
    char bdf[64];
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
    if (f && (f >> node)) return node;
    // containers usually hide the PCI tree but show the KFD topology: find the GPU node by its PCI location, then the CPU node
    // that has an io_link to it (KFD numbers its CPU nodes like the NUMA nodes)
    unsigned dom = 0, bus = 0, dv = 0, fn = 0;
    if (sscanf(bdf, "%x:%x:%x.%x", &dom, &bus, &dv, &fn) != 4) return -1;
    const long want = (long)((bus << 8) | (dv << 3) | fn);
    auto prop = [](const std::string& path, const char* key, long& out) {
      std::ifstream pf(path);
      std::string k; long v;
      while (pf >> k >> v) if (k == key) { out = v; return true; }
      return false;
    };
    const std::string top = "/sys/class/kfd/kfd/topology/nodes/";
    int gpu_node = -1;
    for (int n = 0; n < 64 && gpu_node < 0; n++) {
      long loc = -1, simd = 0;
      if (prop(top + std::to_string(n) + "/properties", "simd_count", simd) && simd > 0 &&
          prop(top + std::to_string(n) + "/properties", "location_id", loc) && loc == want) gpu_node = n;
    }
    if (gpu_node < 0) return -1;
    node = -1;
    for (int n = 0; n < 64 && node < 0; n++) {
      long cores = 0;
      if (!prop(top + std::to_string(n) + "/properties", "cpu_cores_count", cores) || cores <= 0) continue;
      for (int l = 0; l < 64; l++) {
        long to = -1;
        if (!prop(top + std::to_string(n) + "/io_links/" + std::to_string(l) + "/properties", "node_to", to)) break;
        if (to == gpu_node) { node = n; break; }
      }
    }
  }
  return node;
}

// CPUs of a NUMA node as an affinity mask (empty on failure)
static bool numa_cpus(int node, cpu_set_t* set) {
  CPU_ZERO(set);
  if (node < 0) return false;
  std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
  std::string txt;
  if (!f || !std::getline(f, txt)) return false;
  bool any = false;
  size_t pos = 0;
  while (pos < txt.size()) {
    size_t comma = txt.find(',', pos);
    std::string part = txt.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
    size_t dash = part.find('-');
    int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
    for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) { CPU_SET(c, set); any = true; }
    if (comma == std::string::npos) break;
    pos = comma + 1;
  }
  return any;
}

int mi355ndt_default_params(mi355ndt_params* p) {
  if (!p) return MI355NDT_ERR_BAD_ARG;
  p->resolution = 1.0f;
  p->step_size = 0.1;
  p->outlier_ratio = 0.55;
  p->trans_epsilon = 0.1;
  p->max_iterations = 35;
  p->neighbor_mode = MI355NDT_DIRECT7;
  p->variant = MI355NDT_VARIANT_OMP;
  p->min_points_per_voxel = 6;
  p->min_covar_eigvalue_mult = 0.01;
  return MI355NDT_OK;
}

int mi355ndt_destroy(mi355ndt_handle* h);

int mi355ndt_create(const mi355ndt_params* params, int device, mi355ndt_handle** out) {
  if (!out) return MI355NDT_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MI355NDT_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return MI355NDT_ERR_BAD_ARG;
  mi355ndt_params p;
  mi355ndt_default_params(&p);
  if (params) p = *params;
  int rc = check_params(p);
  if (rc) return rc;
  mi355ndt_handle* h = new mi355ndt_handle();
  h->device = device;
  h->prm = p;
  gauss_constants3(0.55, 1.0f, h->gauss_last);    // the constructor's gauss_d*_ (impl2:70-76: resolution_ 1.0f, outlier_ratio_ 0.55), whatever the setters say later
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) h->n_cu = pr.multiProcessorCount; }
  if (const char* e = std::getenv("MI355NDT_LEAF_SORTED")) h->leaf_sorted = std::atoi(e) != 0;
  if (const char* e = std::getenv("MI355NDT_FINE_TILES")) { const int v = std::atoi(e); if (v == 1 || v == 2) h->fine_tiles = v; }
  if (const char* e = std::getenv("MI355NDT_ARITH")) h->arith = std::atoi(e) == 1 ? 1 : 0;   // default of MI355NDT_OPT_ARITH for engines created afterwards (tools, A/B runs)
  if (const char* e = std::getenv("MI355NDT_ASYNC")) { h->async_align = std::atoi(e) != 0; h->async_force = std::atoi(e) == 2; }
  if (const char* e = std::getenv("MI355NDT_SWEEP_DYN_SHIFT")) { const int v = std::atoi(e); if (v >= 0 && v <= 30) h->dyn_shift = v; }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return MI355NDT_ERR_HIP;
  }
  if (hipHostMalloc((void**)&h->h_pin_u, 4 * sizeof(unsigned)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_pin_active, 128 * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&h->d_active, 128 * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&h->d_ctl, 2 * sizeof(SweepCtl)) != hipSuccess ||
      hipMalloc((void**)&h->d_hits, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&h->d_hook, 64 * sizeof(double)) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_compute, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_burst[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_burst[1], hipEventDisableTiming) != hipSuccess) {
    mi355ndt_destroy(h);                          // releases whatever was created before the failure
    return MI355NDT_ERR_HIP;
  }
  for (int i = 0; i < mi355ndt_handle::UP_STREAMS; i++)
    if (hipStreamCreateWithFlags(&h->copy_stream[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_uploads[i], hipEventDisableTiming) != hipSuccess) { mi355ndt_destroy(h); return MI355NDT_ERR_HIP; }
  *out = h;
  return MI355NDT_OK;
}

int mi355ndt_stream_end(mi355ndt_handle* h);
int mi355ndt_destroy(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->stream_on) (void)mi355ndt_stream_end(h);
  void* ptrs[] = {h->d_tgt_own, h->d_src_own, h->d_tgt_cnt, h->d_src_cnt, h->d_grid, h->d_nwords, h->d_word_off,
                  h->d_keys_a, h->d_keys_b, h->d_vals_a, h->d_vals_b, h->d_words, h->d_recs, h->d_vox_idx, h->d_vox_n,
                  h->d_state, h->d_partials, h->d_guess, h->d_results, h->d_active, h->d_hook, h->d_aligned, h->d_hits, h->d_seg_start, h->d_heads, h->d_head_cnt, h->d_sums,
                  h->d_cent, h->d_icov64, h->d_kdw, h->d_rs_hist, h->d_rs_offs, h->d_active_list, h->d_ctl, h->d_cstart, h->d_cend, h->d_fit, h->d_pf_in, h->d_pf_out, h->d_pf_keep, h->d_pf_keys,
                  h->d_pf_vals, h->d_pf_flag, h->d_pf_pos, h->d_pf_mm, h->d_pf_grid, h->d_pf_tmp, h->d_score_pts, h->d_score_part,
                  h->d_ring, h->d_arrived, h->d_actl, h->d_atab, h->d_sorted, h->d_recs_fast};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (void* p : {(void*)h->d_grid_of, (void*)h->d_seq, (void*)h->d_seq_out, (void*)h->d_stamps}) if (p) (void)hipFree(p);
  if (h->h_seq_flags) (void)hipHostFree((void*)h->h_seq_flags);
  for (hipStream_t cs : h->copy_stream) if (cs) (void)hipStreamSynchronize(cs);
  for (auto& u : h->up) { if (u.h) (void)hipHostFree(u.h); if (u.d) (void)hipFree(u.d); if (u.ev) (void)hipEventDestroy(u.ev); }
  if (h->h_pin_aligned) (void)hipHostFree(h->h_pin_aligned);
  for (hipEvent_t e : h->ev_uploads) if (e) (void)hipEventDestroy(e);
  if (h->ev_compute) (void)hipEventDestroy(h->ev_compute);
  for (hipStream_t cs : h->copy_stream) if (cs) (void)hipStreamDestroy(cs);
  if (h->h_pin_actl) (void)hipHostFree(h->h_pin_actl);
  if (h->h_pin_u) (void)hipHostFree(h->h_pin_u);
  if (h->h_pin_active) (void)hipHostFree(h->h_pin_active);
  if (h->h_pin_guess) (void)hipHostFree(h->h_pin_guess);
  for (hipEvent_t e : h->ev_burst) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  for (auto* v : {&h->ev_sweep, &h->ev_update, &h->ev_build})
    for (auto& e : *v) { if (!e.first_shared) (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MI355NDT_OK;
}

int mi355ndt_get_params(const mi355ndt_handle* h, mi355ndt_params* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!out) return MI355NDT_ERR_BAD_ARG;
  *out = h->prm;
  return MI355NDT_OK;
}

int mi355ndt_set_stream(mi355ndt_handle* h, void* s) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (s) {
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s;
    h->own_stream = false;
  } else if (!h->own_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return MI355NDT_OK;
}

const char* mi355ndt_last_error(const mi355ndt_handle* h) { return h ? h->err.c_str() : "bad handle"; }

int mi355ndt_synchronize(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  HIPCHK(h, hipSetDevice(h->device));
  // uploads are asynchronous on the copy streams: "everything issued so far is done" includes them (and a failed transfer
  // surfaces here, not in an unrelated later call)
  for (hipStream_t cs : h->copy_stream) if (cs) HIPCHK(h, hipStreamSynchronize(cs));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MI355NDT_OK;
}

int mi355ndt_batch_size(const mi355ndt_handle* h) { return h ? h->n_pairs : MI355NDT_ERR_BAD_HANDLE; }

// ---- capacity management ----------------------------------------------------------------------
static int ensure_pair_arrays(mi355ndt_handle* h, int n_pairs) {
  if (n_pairs <= h->cap_pairs) return MI355NDT_OK;
  for (hipStream_t cs : h->copy_stream) HIPCHK(h, hipStreamSynchronize(cs));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // the arrays are released and re-created one by one: until all of them exist again the engine holds no batch at all
  // (a failure half way must not leave cap_pairs vouching for freed or undersized buffers)
  h->cap_pairs = 0; h->n_pairs = 0;
  h->targets_built = false; h->have_target = false; h->have_source = false; h->aligned_once = false;
  auto re = [&](void** p, size_t bytes) -> hipError_t {
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; if (e != hipSuccess) return e; }
    return hipMalloc(p, bytes);
  };
  HIPCHK(h, re((void**)&h->d_tgt_cnt, n_pairs * sizeof(int)));
  HIPCHK(h, re((void**)&h->d_src_cnt, n_pairs * sizeof(int)));
  h->up_tgt_cnt.clear(); h->up_src_cnt.clear();       // fresh device arrays: nothing uploaded yet
  HIPCHK(h, re((void**)&h->d_grid, n_pairs * sizeof(GridDesc)));
  HIPCHK(h, re((void**)&h->d_nwords, (n_pairs + 2) * sizeof(unsigned)));
  // build control words, zeroed by ONE memset per build: [0] total bitmap words, [1] largest grid, then per target six extremes
  // (k_minmax's encoding makes zero "none yet")
  HIPCHK(h, re((void**)&h->d_word_off, (2 + 6 * (size_t)n_pairs) * sizeof(unsigned)));
  h->d_minmax = h->d_word_off + 2;
  HIPCHK(h, re((void**)&h->d_state, n_pairs * sizeof(PairState)));
  HIPCHK(h, re((void**)&h->d_guess, n_pairs * 16 * sizeof(float)));
  if (h->h_pin_guess) { HIPCHK(h, hipHostFree(h->h_pin_guess)); h->h_pin_guess = nullptr; }
  HIPCHK(h, hipHostMalloc((void**)&h->h_pin_guess, (size_t)n_pairs * 16 * sizeof(float)));
  HIPCHK(h, re((void**)&h->d_results, n_pairs * sizeof(mi355ndt_result)));
  HIPCHK(h, re((void**)&h->d_active_list, n_pairs * sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->d_grid, 0, n_pairs * sizeof(GridDesc), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_state, 0, n_pairs * sizeof(PairState), h->stream));
  h->cap_pairs = n_pairs;
  h->h_tgt_cnt.assign(n_pairs, 0);
  h->h_src_cnt.assign(n_pairs, 0);
  return MI355NDT_OK;
}

static int alloc_side(mi355ndt_handle* h, bool tgt, int n_pairs, size_t pitch) {
  float*& buf = tgt ? h->d_tgt_own : h->d_src_own;
  size_t& own_pitch = tgt ? h->own_tgt_pitch : h->own_src_pitch;
  int& own_pairs = tgt ? h->own_tgt_pairs : h->own_src_pairs;
  if (buf && own_pitch == pitch && own_pairs == n_pairs) return MI355NDT_OK;
  for (hipStream_t cs : h->copy_stream) HIPCHK(h, hipStreamSynchronize(cs));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (buf) { HIPCHK(h, hipFree(buf)); buf = nullptr; }
  HIPCHK(h, hipMalloc((void**)&buf, (size_t)n_pairs * 3 * pitch * sizeof(float)));
  own_pitch = pitch; own_pairs = n_pairs;
  std::vector<int>& cnt = tgt ? h->h_tgt_cnt : h->h_src_cnt;
  std::fill(cnt.begin(), cnt.end(), 0);
  if (tgt) { h->targets_built = false; h->have_target = false; } else { h->have_source = false; }
  return MI355NDT_OK;
}

int mi355ndt_batch_reserve(mi355ndt_handle* h, int n_pairs, size_t max_tgt, size_t max_src) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if (n_pairs <= 0 || max_tgt == 0 || max_src == 0 || n_pairs > MAX_PAIRS) return MI355NDT_ERR_BAD_ARG;
  if (max_tgt >= (1u << 31) || max_src >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  // pitches padded to 64 floats so every row starts 256-B aligned
  size_t tp = (max_tgt + 63) & ~(size_t)63, sp = (max_src + 63) & ~(size_t)63;
  int rc = ensure_pair_arrays(h, n_pairs);
  if (rc) return rc;
  if (n_pairs != h->n_pairs || h->d_tgt != h->d_tgt_own || h->d_src != h->d_src_own) {
    std::fill(h->h_tgt_cnt.begin(), h->h_tgt_cnt.end(), 0);
    std::fill(h->h_src_cnt.begin(), h->h_src_cnt.end(), 0);
    h->targets_built = false; h->have_target = false; h->have_source = false; h->aligned_once = false;
  }
  rc = alloc_side(h, true, n_pairs, tp);
  if (rc) return rc;
  rc = alloc_side(h, false, n_pairs, sp);
  if (rc) return rc;
  h->n_pairs = n_pairs;
  h->tgt_pitch = tp; h->src_pitch = sp;
  h->d_tgt = h->d_tgt_own; h->d_src = h->d_src_own;
  return MI355NDT_OK;
}

// The compute stream must not start before the uploads enqueued so far have landed, and an upload must not overwrite rows a
// kernel enqueued earlier still reads: the two streams hand over through two events.
static int uploads_before_compute(mi355ndt_handle* h) {
  std::lock_guard<std::mutex> lk(h->up_mtx);
  if (h->uploads_pending) {
    for (int i = 0; i < mi355ndt_handle::UP_STREAMS; i++) {
      HIPCHK(h, hipEventRecord(h->ev_uploads[i], h->copy_stream[i]));
      HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_uploads[i], 0));
    }
    h->uploads_pending = false;
  }
  return MI355NDT_OK;
}
static int compute_enqueued(mi355ndt_handle* h) {       // call after enqueueing kernels that read the cloud buffers
  std::lock_guard<std::mutex> lk(h->up_mtx);
  HIPCHK(h, hipEventRecord(h->ev_compute, h->stream));
  for (hipStream_t cs : h->copy_stream) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_compute, 0));
  return MI355NDT_OK;
}

// Host clouds -> SoA rows of their pair slots, asynchronously, SEVERAL CLOUDS PER TRANSFER.  Returns as soon as the caller's memory is no longer
// needed (the records are compacted into a pinned ring slot; nothing of the caller's buffers is referenced afterwards).  One transfer = one
// hipMemcpyAsync + one k_deinterleave_multi launch + one event, whatever the number of clouds in it: with one cloud per transfer the ~50 us of
// HIP calls per cloud, serialised under the engine's lock, held the staging of a 271-pair batch to 60 GB/s of records read whatever the number
// of staging threads (round 6, tools/host_stage_probe.cpp: the same compaction alone reaches 180-195 GB/s at eight threads on the same host).
struct UpItem { float* d_base; size_t pitch; int pair; const void* pts; size_t n, stride; };
static int upload_items(mi355ndt_handle* h, const UpItem* it, int cnt) {
  if (cnt < 1 || cnt > UP_GROUP_MAX) return MI355NDT_ERR_BAD_ARG;
  size_t total = 0;
  for (int k = 0; k < cnt; k++) {
    if (!it[k].pts && it[k].n) return MI355NDT_ERR_BAD_ARG;
    if ((it[k].n && it[k].stride < 12) || it[k].n > it[k].pitch) return MI355NDT_ERR_BAD_ARG;
    total += it[k].n;
  }
  mi355ndt_handle::UpSlot* u = nullptr;
  for (;;) {                                      // a slot no other thread is filling right now
    {
      std::lock_guard<std::mutex> lk(h->up_mtx);
      for (int t = 0; t < mi355ndt_handle::UP_SLOTS && !u; t++) {
        mi355ndt_handle::UpSlot* c = &h->up[(h->up_next + t) % mi355ndt_handle::UP_SLOTS];
        if (!c->filling) { u = c; h->up_next = (h->up_next + t + 1) % mi355ndt_handle::UP_SLOTS; c->filling = true; }
      }
    }
    if (u) break;
    std::this_thread::yield();                    // more uploader threads than slots
  }
  hipError_t e = hipSuccess;
  if (!u->ev) e = hipEventCreateWithFlags(&u->ev, hipEventDisableTiming);
  if (e == hipSuccess && u->used) e = hipEventSynchronize(u->ev);      // the slot's previous transfer has to be out of the pinned buffer
  if (e == hipSuccess && total > u->cap) {
    if (u->h) { (void)hipHostFree(u->h); u->h = nullptr; }
    if (u->d) { (void)hipFree(u->d); u->d = nullptr; }
    u->cap = 0; u->used = false;
    const size_t cap = std::max(total, (size_t)65536);
    e = hipHostMalloc((void**)&u->h, cap * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&u->d, cap * 3 * sizeof(float));
    if (e == hipSuccess) u->cap = cap;
  }
  if (e != hipSuccess) {
    std::lock_guard<std::mutex> lk(h->up_mtx);
    h->err = std::string("upload staging: ") + hipGetErrorString(e);
    u->filling = false;
    return MI355NDT_ERR_HIP;
  }
  // the CPU part, outside the lock: x,y,z of every record into the pinned slot, cloud after cloud
  DeintTab tab;
  tab.cnt = cnt;
  size_t off = 0, max_pitch = 0;
  for (int k = 0; k < cnt; k++) {
    const unsigned char* p = (const unsigned char*)it[k].pts;
    float* dst = u->h + 3 * off;
    const size_t n = it[k].n, stride = it[k].stride;
    if (stride == 12) { if (n) memcpy(dst, p, n * 12); }
    else for (size_t i = 0; i < n; i++) memcpy(dst + 3 * i, p + i * stride, 12);
    tab.e[k].src_off = 3 * off; tab.e[k].n = (int)n; tab.e[k].rows = it[k].d_base + (size_t)it[k].pair * 3 * it[k].pitch; tab.e[k].pitch = it[k].pitch;
    off += n;
    max_pitch = std::max(max_pitch, it[k].pitch);
  }
  {
    std::lock_guard<std::mutex> lk(h->up_mtx);
    // the copy stream is chosen by DESTINATION (the pair slot's group), not by staging slot: two uploads into the same rows -- set_source(A)
    // then set_source(B) with no build / align in between -- ride one stream and land in call order (a group never spans two stream classes:
    // mi355ndt_batch_set_clouds groups pairs by pair / UP_GROUP_PAIRS)
    hipStream_t cs = h->copy_stream[(it[0].pair / UP_GROUP_PAIRS) % mi355ndt_handle::UP_STREAMS];
    if (total) e = hipMemcpyAsync(u->d, u->h, total * 3 * sizeof(float), hipMemcpyHostToDevice, cs);
    if (e == hipSuccess) {
      k_deinterleave_multi<<<dim3((unsigned)((max_pitch + 255) / 256), (unsigned)cnt), 256, 0, cs>>>(u->d, tab);
      e = hipEventRecord(u->ev, cs);
    }
    u->used = e == hipSuccess;
    u->filling = false;
    h->uploads_pending = true;
    h->P.cloud_uploads += cnt;                     // (counted whether or not event profiling is on: tests/test_adaptor.py holds the drop-in to one per frame)
    h->P.cloud_upload_bytes += (long long)(total * 3 * sizeof(float));
    h->P.cloud_transfers++;
    if (e != hipSuccess) { h->err = std::string("upload: ") + hipGetErrorString(e); return MI355NDT_ERR_HIP; }
  }
  return MI355NDT_OK;
}
static int upload_cloud(mi355ndt_handle* h, float* d_base, size_t pitch, int pair, const void* pts, size_t n, size_t stride) {
  const UpItem it = {d_base, pitch, pair, pts, n, stride};
  return upload_items(h, &it, 1);
}

int mi355ndt_batch_set_target(mi355ndt_handle* h, int pair, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs || h->d_tgt != h->d_tgt_own) return MI355NDT_ERR_BAD_ARG;
  if (hipError_t e = hipSetDevice(h->device)) { std::lock_guard<std::mutex> lk(h->up_mtx); h->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return MI355NDT_ERR_HIP; }
  int rc = upload_cloud(h, h->d_tgt_own, h->tgt_pitch, pair, pts, n, stride);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(h->up_mtx);
  h->h_tgt_cnt[pair] = (int)n;
  h->targets_built = false;
  h->have_target = true;
  return MI355NDT_OK;
}

int mi355ndt_batch_set_source(mi355ndt_handle* h, int pair, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs || h->d_src != h->d_src_own) return MI355NDT_ERR_BAD_ARG;
  if (hipError_t e = hipSetDevice(h->device)) { std::lock_guard<std::mutex> lk(h->up_mtx); h->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return MI355NDT_ERR_HIP; }
  int rc = upload_cloud(h, h->d_src_own, h->src_pitch, pair, pts, n, stride);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(h->up_mtx);
  h->h_src_cnt[pair] = (int)n;
  h->have_source = true;
  return MI355NDT_OK;
}

// A whole batch of host clouds at once: the engine's own staging threads split the pairs among themselves (staging -- copying x,y,z
// out of the caller's records into pinned memory -- is the CPU-bound part of a host-cloud batch; one thread does ~10 k clouds/s).
int mi355ndt_batch_set_clouds(mi355ndt_handle* h, int first_pair, int n, const void* const* targets, const size_t* target_counts,
                              const void* const* sources, const size_t* source_counts, size_t stride, int n_threads) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (n <= 0 || first_pair < 0 || first_pair + n > h->n_pairs || (!targets && !sources) || (targets && !target_counts) || (sources && !source_counts))
    return MI355NDT_ERR_BAD_ARG;
  if (h->d_tgt != h->d_tgt_own || h->d_src != h->d_src_own) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const int nt = std::max(1, std::min(n_threads > 0 ? n_threads : 8, n));
  std::vector<int> rcs((size_t)nt, MI355NDT_OK);
  // the engine's own threads stage next to the GPU -- but only on CPUs the CALLER may use: the NUMA node's CPUs intersected with the
  // calling thread's affinity mask (a taskset / cgroup-restricted process keeps its restriction); an empty intersection = no pinning
  cpu_set_t near, mine;
  bool pin = numa_cpus(mi355ndt_host_numa_node(h->device), &near);
  if (pin && sched_getaffinity(0, sizeof mine, &mine) == 0) {
    CPU_AND(&near, &near, &mine);
    pin = CPU_COUNT(&near) > 0;
  } else pin = false;
  // pairs are taken in GROUPS of UP_GROUP_PAIRS consecutive pair slots (aligned to the slot index, so that a slot's uploads always ride the same
  // copy stream): one transfer per group -- both clouds of up to four pairs -- instead of one per cloud (upload_items); a thread that runs slowly
  // (the caller's, t = 0, may sit on a narrowed CPU set) simply takes fewer groups
  const int g_first = first_pair / UP_GROUP_PAIRS, g_last = (first_pair + n - 1) / UP_GROUP_PAIRS;
  std::atomic<int> next_group{g_first};
  auto work = [&](int t) {
    (void)hipSetDevice(h->device);
    if (pin && t > 0) (void)sched_setaffinity(0, sizeof near, &near);   // (t = 0 is the caller's thread: left alone)
    for (int g = next_group.fetch_add(1); g <= g_last; g = next_group.fetch_add(1)) {
      UpItem it[UP_GROUP_MAX];
      int cnt = 0;
      for (int pr = std::max(first_pair, g * UP_GROUP_PAIRS); pr < std::min(first_pair + n, (g + 1) * UP_GROUP_PAIRS); pr++) {
        const int k = pr - first_pair;
        if (targets) it[cnt++] = UpItem{h->d_tgt_own, h->tgt_pitch, pr, targets[k], target_counts[k], stride};
        if (sources) it[cnt++] = UpItem{h->d_src_own, h->src_pitch, pr, sources[k], source_counts[k], stride};
      }
      const int rc = upload_items(h, it, cnt);
      if (rc != MI355NDT_OK) { rcs[(size_t)t] = rc; return; }
    }
  };
  std::vector<std::thread> th;
  try {                                          // nothing may be thrown across the C boundary: a thread that cannot be created
    th.reserve((size_t)nt);                      // (std::system_error) just means the others -- at least the caller's -- do its share
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
  } catch (...) {}
  work(0);
  for (auto& x : th) x.join();
  for (int rc : rcs) if (rc != MI355NDT_OK) return rc;
  {
    std::lock_guard<std::mutex> lk(h->up_mtx);
    for (int k = 0; k < n; k++) {
      if (targets) h->h_tgt_cnt[(size_t)(first_pair + k)] = (int)target_counts[k];
      if (sources) h->h_src_cnt[(size_t)(first_pair + k)] = (int)source_counts[k];
    }
    if (targets) { h->targets_built = false; h->have_target = true; }
    if (sources) h->have_source = true;
  }
  return MI355NDT_OK;
}

int mi355ndt_batch_bind_device(mi355ndt_handle* h, int n_pairs, const float* d_t, const int* tc, size_t tp,
                               const float* d_s, const int* scnt, size_t sp) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if (n_pairs <= 0 || !d_t || !d_s || !tc || !scnt || tp == 0 || sp == 0 || n_pairs > MAX_PAIRS) return MI355NDT_ERR_BAD_ARG;
  if (tp >= (1u << 31) || sp >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  for (int b = 0; b < n_pairs; b++) if (tc[b] < 0 || (size_t)tc[b] > tp || scnt[b] < 0 || (size_t)scnt[b] > sp) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_pair_arrays(h, n_pairs);
  if (rc) return rc;
  h->n_pairs = n_pairs;
  h->d_tgt = d_t; h->d_src = d_s;
  h->tgt_pitch = tp; h->src_pitch = sp;
  for (int b = 0; b < n_pairs; b++) { h->h_tgt_cnt[b] = tc[b]; h->h_src_cnt[b] = scnt[b]; }
  h->targets_built = false;
  h->have_target = h->have_source = true;
  h->aligned_once = false;
  return MI355NDT_OK;
}

// ---- profiling helpers ------------------------------------------------------------------------
// timing events come from a pool that mi355ndt_profile_enable fills up front: creating events inside a timed region can
// stall for milliseconds when the runtime has to grow its signal pool
static hipError_t ev_take(mi355ndt_handle* h, hipEvent_t* e) {
  if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); return hipSuccess; }
  return hipEventCreate(e);
}
// Back-to-back kernels share an event: the end of one span is the begin of the next (half the event records in the round loop).
static hipError_t ev_begin(mi355ndt_handle* h, std::vector<mi355ndt_handle::EvSpan>& v) {
  if (h->ev_last_fresh && h->ev_last) {
    v.push_back({h->ev_last, nullptr, true});
    h->ev_last_fresh = false;
    return hipSuccess;
  }
  hipEvent_t a;
  hipError_t e = ev_take(h, &a); if (e != hipSuccess) return e;
  v.push_back({a, nullptr, false});
  return hipEventRecord(a, h->stream);
}
static hipError_t ev_end(mi355ndt_handle* h, std::vector<mi355ndt_handle::EvSpan>& v) {
  hipEvent_t b;
  hipError_t e = ev_take(h, &b); if (e != hipSuccess) return e;
  v.back().second = b;
  h->ev_last = b;
  h->ev_last_fresh = true;
  return hipEventRecord(b, h->stream);
}
static void ev_collect(mi355ndt_handle* h, std::vector<mi355ndt_handle::EvSpan>& v, double& ms, long long& n) {
  for (auto& e : v) {
    float t = 0;
    if (e.second && hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) { ms += t; n++; }
    if (!e.first_shared) h->ev_pool.push_back(e.first);
    if (e.second) h->ev_pool.push_back(e.second);
  }
  v.clear();
  h->ev_last_fresh = false;
}
int mi355ndt_profile_enable(mi355ndt_handle* h, int on) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  h->prof = on != 0;
  if (h->stream_on) for (int c = 0; c < h->s_nctx; c++) if (h->sctx[c].e) { h->sctx[c].e->ev_pool_target = 128; (void)mi355ndt_profile_enable(h->sctx[c].e, on); }
  if (h->prof) {
    HIPCHK(h, hipSetDevice(h->device));
    while (h->ev_pool.size() < h->ev_pool_target) { // ~40 profiled steps of a 10-round batch align before the pool has to grow
      hipEvent_t e;
      HIPCHK(h, hipEventCreate(&e));
      h->ev_pool.push_back(e);
    }
  }
  return MI355NDT_OK;
}
int mi355ndt_profile_reset(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  double d; long long n;
  ev_collect(h, h->ev_sweep, d, n); ev_collect(h, h->ev_update, d, n); ev_collect(h, h->ev_build, d, n);
  if (h->stream_on) for (int c = 0; c < h->s_nctx; c++) if (mi355ndt_handle* e = h->sctx[c].e) {
    ev_collect(e, e->ev_sweep, d, n); ev_collect(e, e->ev_update, d, n); ev_collect(e, e->ev_build, d, n);
    e->P = mi355ndt_profile{};
  }
  h->P = mi355ndt_profile{};
  (void)hipMemsetAsync(h->d_hits, 0, sizeof(unsigned long long), h->stream);
  (void)hipStreamSynchronize(h->stream);
  return MI355NDT_OK;
}
int mi355ndt_profile_get(mi355ndt_handle* h, mi355ndt_profile* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!out) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ev_collect(h, h->ev_sweep, h->P.sweep_ms, h->P.sweep_launches);
  ev_collect(h, h->ev_update, h->P.update_ms, h->P.update_launches);
  ev_collect(h, h->ev_build, h->P.build_ms, h->P.build_launches);
  if (h->stream_on) for (int c = 0; c < h->s_nctx; c++) if (mi355ndt_handle* e = h->sctx[c].e) {   // the contexts' builds (and synchronous re-runs) are this handle's work
    ev_collect(e, e->ev_sweep, h->P.sweep_ms, h->P.sweep_launches);
    ev_collect(e, e->ev_update, h->P.update_ms, h->P.update_launches);
    ev_collect(e, e->ev_build, h->P.build_ms, h->P.build_launches);
    h->P.build_alg_bytes += e->P.build_alg_bytes; e->P.build_alg_bytes = 0;
    { std::lock_guard<std::mutex> lk(e->up_mtx);   // (mi355ndt_stream_submit_host stages into the contexts' engines)
      h->P.cloud_uploads += e->P.cloud_uploads; e->P.cloud_uploads = 0; h->P.cloud_upload_bytes += e->P.cloud_upload_bytes; e->P.cloud_upload_bytes = 0;
      h->P.cloud_transfers += e->P.cloud_transfers; e->P.cloud_transfers = 0; }
  }
  unsigned long long hh = 0;                      // (point, voxel) evaluations since the last reset, summed on the device
  HIPCHK(h, hipMemcpy(&hh, h->d_hits, sizeof hh, hipMemcpyDeviceToHost));
  *out = h->P;
  if (h->stream_on) { out->stream_reserved_slots = h->s_reserve_wg; out->stream_launch_slots = h->s_launch_slots; }
  out->sweep_hits += (long long)hh;
  out->sweep_alg_bytes += 64.0 * (double)hh;
  return MI355NDT_OK;
}

// ---- target build -----------------------------------------------------------------------------
static int build_targets_impl(mi355ndt_handle* h);
int mi355ndt_batch_build_targets(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  const int rc = build_targets_impl(h);
  // an error exit may leave kernels queued that still read the cloud rows: later uploads have to wait for them all the same
  if (rc != MI355NDT_OK && h->ev_compute) (void)compute_enqueued(h);
  return rc;
}
static int build_targets_impl(mi355ndt_handle* h) {
  if (h->n_pairs <= 0 || !h->d_tgt) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcu = uploads_before_compute(h); if (rcu) return rcu; }
  const int B = h->n_pairs;
  const size_t pitch = h->tgt_pitch;
  const size_t total = (size_t)B * pitch;
  hipStream_t s = h->stream;
  if (h->async_build && h->counts_preloaded) {
    h->up_tgt_cnt.clear();                          // stream mode: the counts came with the batch's one input copy (mi355ndt_stream_submit)
  } else if (h->up_tgt_cnt.size() != (size_t)B || !std::equal(h->up_tgt_cnt.begin(), h->up_tgt_cnt.end(), h->h_tgt_cnt.begin())) {
    HIPCHK(h, hipMemcpyAsync(h->d_tgt_cnt, h->h_tgt_cnt.data(), B * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));   // h_tgt_cnt is pageable
    h->up_tgt_cnt.assign(h->h_tgt_cnt.begin(), h->h_tgt_cnt.begin() + B);   // exactly what the device now holds
  }

  // workspace
  if (total > h->keys_cap) {
    size_t c1 = h->keys_cap, c2 = h->keys_cap, c3 = h->keys_cap, c4 = h->keys_cap;
    HIPCHK(h, grow(h->d_keys_a, c1, total)); HIPCHK(h, grow(h->d_keys_b, c2, total));
    HIPCHK(h, grow(h->d_vals_a, c3, total)); HIPCHK(h, grow(h->d_vals_b, c4, total));
    h->keys_cap = total;
  }
  const int minpts = h->prm.min_points_per_voxel;
  const size_t rpp = pitch / (size_t)minpts + 1;
  if (rpp > ((size_t)1 << ID_BITS)) { h->err = "target too large: voxel ids would not fit the sweep's queue entries"; return MI355NDT_ERR_BAD_ARG; }
  if ((size_t)B * rpp > h->recs_cap || rpp != h->recs_per_pair) {
    size_t need = (size_t)B * rpp, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    if (h->d_recs) { HIPCHK(h, hipFree(h->d_recs)); h->d_recs = nullptr; }
    if (h->d_vox_idx) { HIPCHK(h, hipFree(h->d_vox_idx)); h->d_vox_idx = nullptr; }
    if (h->d_vox_n) { HIPCHK(h, hipFree(h->d_vox_n)); h->d_vox_n = nullptr; }
    if (h->d_seg_start) { HIPCHK(h, hipFree(h->d_seg_start)); h->d_seg_start = nullptr; }
    if (h->d_sums) { HIPCHK(h, hipFree(h->d_sums)); h->d_sums = nullptr; }
    if (h->d_cent) { HIPCHK(h, hipFree(h->d_cent)); h->d_cent = nullptr; }
    HIPCHK(h, grow(h->d_recs, c1, need)); HIPCHK(h, grow(h->d_vox_idx, c2, need)); HIPCHK(h, grow(h->d_vox_n, c3, need));
    HIPCHK(h, grow(h->d_seg_start, c4, need)); HIPCHK(h, grow(h->d_sums, c5, need * 9));
    { size_t c6 = 0; HIPCHK(h, grow(h->d_cent, c6, need * 3)); }
    h->recs_cap = need; h->recs_per_pair = rpp;
  }
  h->ev_last_fresh = false;
  const bool build_events = h->prof && !(h->build_stamped && h->async_build);   // (the stream's builds are stamped by the kernels around them)
  if (build_events) HIPCHK(h, ev_begin(h, h->ev_build));
  const int gx = (int)((pitch + 255) / 256);
  if (!h->word_off_cleared) HIPCHK(h, hipMemsetAsync(h->d_word_off, 0, (2 + 6 * (size_t)h->cap_pairs) * sizeof(unsigned), s));   // (stream mode: k_stream_inputs did)
  h->word_off_cleared = false;
  k_minmax<<<dim3(std::max(1, std::min((gx + 3) / 4 / MM_ILP, 64)), B), 256, 0, s>>>(h->d_tgt, pitch, h->d_tgt_cnt, h->d_minmax);
  k_griddesc<<<(B + 63) / 64, 64, 0, s>>>(h->d_minmax, h->d_grid, h->d_nwords, h->prm.resolution, B, (unsigned)rpp);
  k_word_offsets<<<1, 1024, 0, s>>>(h->d_grid, h->d_nwords, B, h->d_word_off);   // d_word_off[0] = total words, [1] = largest grid
  size_t total_words;
  int cb;
  const bool planned = h->async_build && h->plan_cb > 0 && h->plan_words > 0 && h->plan_words <= h->words_cap && h->d_bstat;
  if (planned) {
    // no wait: the plan's key width and pool size, checked on the device (a batch that does not fit loses its grids and is flagged)
    k_build_check<<<(B + 255) / 256, 256, 0, s>>>(h->d_word_off, h->d_grid, h->d_nwords, B, (unsigned)std::min(h->plan_words, (size_t)0xFFFFFFFFu), h->plan_cb, h->d_bstat + 1);
    total_words = h->plan_words;
    cb = h->plan_cb;
  } else {
    HIPCHK(h, hipMemcpyAsync(h->h_pin_u, h->d_word_off, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));           // total bitmap words -> size the pool; largest grid -> key width
    total_words = h->h_pin_u[0];
    cb = std::max(1, ceil_log2(h->h_pin_u[1] + 1u));   // cell field: every cell index + the all-ones "not binned" value
    if (h->async_build) cb = std::max(cb, h->plan_cb);        // (a wider field sorts the same order: the plan only ever grows)
    h->last_total_words = total_words;
    if (h->d_bstat) HIPCHK(h, hipMemsetAsync(h->d_bstat + 3, 0, sizeof(unsigned), s));
  }
  if (total_words > h->words_cap) {
    size_t c = h->words_cap;
    HIPCHK(h, grow(h->d_words, c, std::max(total_words, (size_t)1024)));
    h->words_cap = c;
  }
  if (total_words) HIPCHK(h, hipMemsetAsync(h->d_words, 0, total_words * sizeof(BitWord), s));
  const bool mt_live = mt_is_live(h->prm);
  const bool want_cent = h->prm.neighbor_mode == MI355NDT_KDTREE || mt_live;   // f32 leaf centroids: KDTREE probe, computeHessian
  h->cent_built = want_cent;
  h->icov64_built = mt_live;
  if (mt_live) HIPCHK(h, grow(h->d_icov64, h->icov64_cap, h->recs_cap * 9));
  // the tolerance arithmetic's records and tree leaf sums only where its sweeps are served (DIRECT1 / DIRECT7, dead More-Thuente loop): every other
  // configuration ignores the option altogether -- ordered sums, the exact records alone, results word for word those of the option off
  const bool want_fast_recs = fast_served(h);
  if (want_fast_recs) HIPCHK(h, grow(h->d_recs_fast, h->recs_fast_cap, h->recs_cap));
  h->recs_fast_built = want_fast_recs;
  const bool want_kdw = h->prm.neighbor_mode == MI355NDT_KDTREE && h->prm.variant == MI355NDT_VARIANT_PCA;
  h->kdw_built = want_kdw;
  if (want_kdw) HIPCHK(h, grow(h->d_kdw, h->kdw_cap, h->recs_cap));
  {
    unsigned *ka = h->d_keys_a, *kb = h->d_keys_b;
    // stable sort by cell inside every target's segment (ndt_segsort.hpp): rs_plan(cb) passes, result in kb / d_vals_b
    const RsPlan plan = rs_plan(cb);
    const int npass = plan.passes;
    const int tiles = (int)((pitch + RS_TILE - 1) / RS_TILE);
    {
      const size_t need = ((size_t)B * tiles) << RS_MAX_BITS;
      if (need > h->rs_cap) {
        size_t c1 = 0, c2 = 0;
        if (h->d_rs_hist) { HIPCHK(h, hipFree(h->d_rs_hist)); h->d_rs_hist = nullptr; }
        if (h->d_rs_offs) { HIPCHK(h, hipFree(h->d_rs_offs)); h->d_rs_offs = nullptr; }
        HIPCHK(h, grow(h->d_rs_hist, c1, need)); HIPCHK(h, grow(h->d_rs_offs, c2, need));
        h->rs_cap = need;
      }
    }
    unsigned *kin = (npass & 1) ? ka : kb, *kout = (npass & 1) ? kb : ka;      // an odd number of hops must end in kb
    unsigned *vin = (npass & 1) ? h->d_vals_a : h->d_vals_b, *vout = (npass & 1) ? h->d_vals_b : h->d_vals_a;
    // (no key kernel: the first pass's histogram computes the cell indices from the points and writes them, ndt_segsort.hpp)
    const RsPoints points = {h->d_tgt, h->d_tgt_cnt, h->d_grid, cb, kin};
    for (int p = 0; p < npass; p++) {
      rs_pass(s, plan.bits, kin, vin, kout, vout, pitch, p * plan.bits, h->d_rs_hist, h->d_rs_offs, tiles, B, p == 0, p == 0 ? &points : nullptr);
      std::swap(kin, kout); std::swap(vin, vout);
    }
    // k_mark leaves the leaves' run starts in per-wave slices; k_rank strings them together by voxel id (d_seg_start)
    const unsigned nsl = ls_slices(pitch), scap = ls_slice_cap(minpts);
    HIPCHK(h, grow(h->d_heads, h->heads_cap, (size_t)B * nsl * scap));
    HIPCHK(h, grow(h->d_head_cnt, h->head_cnt_cap, (size_t)B * nsl));
    k_mark<unsigned><<<dim3((nsl + 3) / 4, B), 256, 0, s>>>(kb, pitch, h->d_grid, h->d_words, h->d_heads, h->d_head_cnt, nsl, scap, minpts, cb);
    k_rank<<<B, 1024, 0, s>>>(h->d_grid, h->d_words, h->d_heads, h->d_head_cnt, nsl, scap, h->d_seg_start);
    // leaf-sum workgroups per target: 64 keeps ~4 targets (3 MB of points) in flight per XCD, inside its 4 MB L2
    const int lb = std::max(1, std::min((int)((rpp + LS_WAVES - 1) / LS_WAVES), 64));
    if (want_fast_recs && !want_cent && !h->leaf_sorted) {      // tolerance arithmetic: the leaf sums as a tree (ndt_build.hpp)
      k_leafsum_tree<<<xcd_grid(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, kb, h->d_vals_b, h->d_grid, h->d_seg_start, h->d_sums, h->d_vox_idx, h->d_vox_n, cb, lb, B);
    } else if (h->leaf_sorted) {
      // the sorted order as 16-byte points first (one streaming gather), then leaf sums that read them contiguously
      HIPCHK(h, grow(h->d_sorted, h->sorted_cap, total));
      const int gb = std::max(1, std::min((int)((pitch + 256 * RUN_ILP - 1) / (256 * RUN_ILP)), 64));
      k_sorted_points<<<xcd_grid(gb, B), 256, 0, s>>>(h->d_tgt, pitch, h->d_vals_b, h->d_sorted, gb, B);
      const unsigned* sp = reinterpret_cast<const unsigned*>(h->d_sorted);
      if (want_cent) k_leafsum<unsigned, true, true><<<xcd_grid(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, kb, sp, h->d_grid, h->d_seg_start,
                                                                                            h->d_sums, h->d_vox_idx, h->d_vox_n, cb, h->d_cent, lb, B);
      else k_leafsum<unsigned, false, true><<<xcd_grid(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, kb, sp, h->d_grid, h->d_seg_start,
                                                                                  h->d_sums, h->d_vox_idx, h->d_vox_n, cb, h->d_cent, lb, B);
    } else if (want_cent) k_leafsum<unsigned, true><<<xcd_grid(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, kb, h->d_vals_b, h->d_grid, h->d_seg_start,
                                                                                    h->d_sums, h->d_vox_idx, h->d_vox_n, cb, h->d_cent, lb, B);
    else k_leafsum<unsigned, false><<<xcd_grid(lb, B), 64 * LS_WAVES, 0, s>>>(h->d_tgt, pitch, kb, h->d_vals_b, h->d_grid, h->d_seg_start,
                                                                          h->d_sums, h->d_vox_idx, h->d_vox_n, cb, h->d_cent, lb, B);
  }
  k_voxels<<<dim3((unsigned)((rpp + 255) / 256), B), 256, 0, s>>>(h->d_grid, h->d_sums, h->d_recs, h->d_vox_n,
                                                                  h->prm.min_covar_eigvalue_mult, h->prm.variant == MI355NDT_VARIANT_PCA,
                                                                  mt_live ? h->d_icov64 : nullptr, want_kdw ? h->d_kdw : nullptr, want_fast_recs ? h->d_recs_fast : nullptr);
  HIPCHK(h, hipGetLastError());
  if (h->prof) {
    if (build_events) HIPCHK(h, ev_end(h, h->ev_build));
    double pts = 0;
    for (int b = 0; b < B; b++) pts += h->h_tgt_cnt[b];
    // B_build: minmax 12 + binning 12 + key write 12 + sort r/w + grouped gather 16 per point (+ records)
    h->P.build_alg_bytes += pts * (12 + 12 + 4 + 16);
  }
  h->targets_built = true;
  h->grid_resolution = h->prm.resolution;
  h->cells_ready = false;
  h->last_cb = cb;
  return compute_enqueued(h);                     // asynchronous: a later upload into these rows has to wait for the kernels above
}

// ---- sweeps -----------------------------------------------------------------------------------
static int prep_align_ws(mi355ndt_handle* h) {
  const int B = h->n_pairs;
  int maxn = 0;
  for (int b = 0; b < B; b++) maxn = std::max(maxn, h->h_src_cnt[b]);
  h->chunks_per_pair = std::max(1, (maxn + CHUNK_PTS - 1) / CHUNK_PTS);
  // latency mode (mi355ndt_set_latency_mode): fine work items for small batches, where the 512-point items of the batch mode leave
  // most of the GPU idle.  Served by the DIRECT1 / DIRECT7 instantiations; the live More-Thuente case keeps the batch kernels.
  {
    const int K = h->prm.neighbor_mode == MI355NDT_DIRECT1 ? 1 : (h->prm.neighbor_mode == MI355NDT_DIRECT7 ? 7 : 0);
    // "small": fewer batch-mode items than two per resident wave; a sequence run sweeps ONE pair at a time whatever the number of frames
    const bool small = h->seq_running || (long long)B * h->chunks_per_pair * QUARTERS < 4LL * h->n_cu * WAVES;
    h->fine_it = (h->latency_mode && K && !mt_is_live(h->prm) && small) ? h->fine_tiles : 0;
  }
  if (h->fine_it) {
    const int item_pts = h->fine_it * 64;
    h->pts_per_chunk = 4 * item_pts;                                   // a block of the fine sweep = four items = one chunk, stored as ONE row
    h->rows_per_pair = std::max(1, (maxn + h->pts_per_chunk - 1) / h->pts_per_chunk);
    h->items_per_pair = 4 * h->rows_per_pair;
  } else {
    h->rows_per_pair = h->chunks_per_pair * QUARTERS;
    h->items_per_pair = h->rows_per_pair;
    h->pts_per_chunk = CHUNK_PTS;
  }
  size_t need = (size_t)B * std::max(h->rows_per_pair, h->chunks_per_pair * QUARTERS) * NACC;   // (the parity hooks may fall back to batch-mode rows)
  HIPCHK(h, grow(h->d_partials, h->partials_cap, need));
  if (h->up_src_cnt.size() != (size_t)B || !std::equal(h->up_src_cnt.begin(), h->up_src_cnt.end(), h->h_src_cnt.begin())) {
    HIPCHK(h, hipMemcpyAsync(h->d_src_cnt, h->h_src_cnt.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->up_src_cnt.assign(h->h_src_cnt.begin(), h->h_src_cnt.begin() + B);
  }
  return MI355NDT_OK;
}

static void make_sweep_const(const mi355ndt_handle* h, SweepConst& sc) {
  double d1, d2;
  gauss_constants(h->prm, d1, d2);
  sc.d1 = d1;
  sc.d2f = (float)d2;                        // impl2:578
  sc.pca = h->prm.variant == MI355NDT_VARIANT_PCA;
  // the lookup divides by the GRID's leaf size (voxel_grid_covariance_omp_impl.hpp:379-381), which is resolution_ except after a setResolution
  // that found no source and therefore left the grid alone (ndt_omp.h:126-136); the Gauss constants and the kd radius follow resolution_
  const float leaf = (h->targets_built && h->grid_resolution > 0.f) ? h->grid_resolution : h->prm.resolution;
  { int ex; float mant = std::frexp(leaf, &ex); sc.leaf_pow2 = (mant == 0.5f) && ex > -100 && ex < 100; sc.inv_leaf = 1.0f / leaf; }
  sc.kd_r2 = (float)((double)h->prm.resolution * (double)h->prm.resolution);   // KdTreeFLANN::radiusSearch: float(radius * radius)
  build_offsets(h->prm.neighbor_mode, sc);
  sc.dyn_shift = h->dyn_shift >= 0 ? h->dyn_shift : (sc.K == 1 ? 3 : 2);
  sc.host_flags = nullptr;
  sc.seq_no = 0;
  sc.rebase_block = 0;
  sc.d1f = (float)d1;
  sc.kq = (float)(-0.5 * (double)sc.d2f * 1.4426950408889634);   // exp(-d2 q / 2) = 2^(kq q)
}

// The arithmetic a sweep runs in: 2 = tolerance arithmetic (MI355NDT_OPT_ARITH = 1; instantiated for DIRECT1 / DIRECT7 with the dead More-Thuente
// loop -- every configuration lv_slam ships; the other searches and the live line search keep the exact kernels), else the f32 sum order.
static bool want_fast(const mi355ndt_handle* h, const SweepConst& sc) { (void)sc; return fast_served(h); }
static int sweep_ord(const mi355ndt_handle* h, const SweepConst& sc) {
  // (a stream's parent handle owns no grids: its contexts' engines build them, with the option as it stood at mi355ndt_stream_begin)
  if (want_fast(h, sc) && (h->recs_fast_built || h->stream_on)) return 2;
  return h->f32_sum_order;
}
static const VoxelRec* sweep_recs(const mi355ndt_handle* h, const SweepConst& sc) {
  return sweep_ord(h, sc) == 2 ? reinterpret_cast<const VoxelRec*>(h->d_recs_fast) : h->d_recs;
}

static int launch_sweep(mi355ndt_handle* h, const SweepConst& sc, int max_pairs = -1) {
  // persistent waves: SWEEP_WPE workgroups per CU pull (pair, chunk, quarter) items until the per-XCD queues are dry
  const int ord = sweep_ord(h, sc);
  dim3 grid((unsigned)(h->n_cu * sweep_wpe(sc.pca != 0, sc.K, ord == 2)));
  if (h->prof) HIPCHK(h, ev_begin(h, h->ev_sweep));
#define NDT_SWEEP_ARGS h->d_src, h->src_pitch, h->d_state, h->d_grid, h->d_words, sweep_recs(h, sc), h->d_partials, h->items_per_pair, h->d_active_list, \
      h->d_ctl + h->ctl_idx, h->d_ctl + (h->ctl_idx ^ 1), sc, h->d_cent, h->d_grid_of_use
  // (the f32 sum order is a template parameter: the alternative order costs no instruction, only a second set of instantiations)
#define NDT_LAUNCH_SWEEP(P, KK) do { if (ord == 1) k_sweep<P, KK, 8, false, 1><<<grid, SWEEP_THREADS, 0, h->stream>>>(NDT_SWEEP_ARGS); \
                                     else k_sweep<P, KK, 8, false, 0><<<grid, SWEEP_THREADS, 0, h->stream>>>(NDT_SWEEP_ARGS); } while (0)
#define NDT_LAUNCH_SWEEP17(P, KK) do { if (ord == 2) k_sweep<P, KK, 8, false, 2><<<grid, SWEEP_THREADS, 0, h->stream>>>(NDT_SWEEP_ARGS); else NDT_LAUNCH_SWEEP(P, KK); } while (0)
#define NDT_LAUNCH_FINE_O(P, KK, O) do { if (h->fine_it == 1) k_sweep<P, KK, 1, true, O><<<grid, SWEEP_THREADS, 0, h->stream>>>(NDT_SWEEP_ARGS); \
                                         else k_sweep<P, KK, 2, true, O><<<grid, SWEEP_THREADS, 0, h->stream>>>(NDT_SWEEP_ARGS); } while (0)
#define NDT_LAUNCH_FINE(P, KK) do { if (ord == 2) NDT_LAUNCH_FINE_O(P, KK, 2); else if (ord == 1) NDT_LAUNCH_FINE_O(P, KK, 1); else NDT_LAUNCH_FINE_O(P, KK, 0); } while (0)
  if (h->fine_it) {                              // latency mode: items dealt statically over the whole grid, sized to the work there can be
    const long long items = (long long)(max_pairs > 0 ? max_pairs : h->n_pairs) * h->items_per_pair;
    grid.x = (unsigned)std::max(1LL, std::min((long long)grid.x, (items + WAVES - 1) / WAVES));
    if (sc.rebase_block) grid.x += 1;              // + the workgroup that computes the next update's re-basing instead of sweeping
    if (sc.pca) { if (sc.K == 1) NDT_LAUNCH_FINE(true, 1); else NDT_LAUNCH_FINE(true, 7); }
    else        { if (sc.K == 1) NDT_LAUNCH_FINE(false, 1); else NDT_LAUNCH_FINE(false, 7); }
  } else if (sc.pca && sc.K == 27) {             // ndt_pca + KDTREE: order-dependent weights, the literal kernel (ndt_sweep_kd.hpp)
#define NDT_KD_ARGS h->d_src, h->src_pitch, h->d_state, h->d_grid, h->d_words, h->d_recs, h->d_cent, h->d_kdw, h->d_partials, h->chunks_per_pair, \
        h->d_active_list, h->d_ctl + h->ctl_idx, h->d_ctl + (h->ctl_idx ^ 1), sc
    const dim3 kdgrid((unsigned)h->chunks_per_pair, (unsigned)h->n_pairs);
    if (ord == 1) k_sweep_pca_kd<1><<<kdgrid, SWEEP_THREADS, 0, h->stream>>>(NDT_KD_ARGS);
    else k_sweep_pca_kd<0><<<kdgrid, SWEEP_THREADS, 0, h->stream>>>(NDT_KD_ARGS);
#undef NDT_KD_ARGS
  } else if (sc.pca) { if (sc.K == 1) NDT_LAUNCH_SWEEP17(true, 1); else if (sc.K == 7) NDT_LAUNCH_SWEEP17(true, 7); else NDT_LAUNCH_SWEEP(true, 26); }
  else        { if (sc.K == 1) NDT_LAUNCH_SWEEP17(false, 1); else if (sc.K == 7) NDT_LAUNCH_SWEEP17(false, 7); else if (sc.K == 26) NDT_LAUNCH_SWEEP(false, 26);
                else NDT_LAUNCH_SWEEP(false, 27); }
#undef NDT_LAUNCH_SWEEP17
#undef NDT_LAUNCH_SWEEP
#undef NDT_LAUNCH_FINE
#undef NDT_LAUNCH_FINE_O
#undef NDT_SWEEP_ARGS
  h->ctl_idx ^= 1;                                // the block this sweep zeroed is the one the next update fills
  if (h->prof) HIPCHK(h, ev_end(h, h->ev_sweep));
  return MI355NDT_OK;
}

static void launch_hessian(mi355ndt_handle* h, const SweepConst& sc) {
  double gc[3] = {0, 0, 0};
  gauss_constants(h->prm, gc[0], gc[1]);
  k_hessian<<<dim3((unsigned)h->chunks_per_pair, (unsigned)h->n_pairs), HESS_THREADS, 0, h->stream>>>(
      h->d_src, h->src_pitch, h->d_state, h->d_grid, h->d_words, h->d_recs, h->d_icov64, h->d_cent, h->d_partials, h->chunks_per_pair,
      gc[0], gc[1], sc.kd_r2, sc.leaf_pow2, sc.inv_leaf);
}

// MI355NDT_OPT_ARITH = 1: results of registrations the tolerance arithmetic is not meant for carry a warning (include/mi355_ndt.h).  A property of the
// pair alone (its hits at the final pose, its iteration count), applied to the host copy of the results by every path that hands results out.
static void tolerance_warnings(const mi355ndt_handle* h, mi355ndt_result* out, int n) {
  SweepConst sc;
  make_sweep_const(h, sc);
  if (!want_fast(h, sc)) return;
  for (int b = 0; b < n; b++)
    if (out[b].status == MI355NDT_OK && (out[b].hits_last < MI355NDT_TOLERANCE_MIN_HITS || out[b].iterations >= h->prm.max_iterations + 2))
      out[b].status = MI355NDT_WARN_TOLERANCE_ARITH;
}
static int batch_align_impl(mi355ndt_handle* h, const float* guesses, mi355ndt_result* out);
int mi355ndt_batch_align(mi355ndt_handle* h, const float* guesses, mi355ndt_result* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  const int rc = batch_align_impl(h, guesses, out);
  if (rc == MI355NDT_OK) tolerance_warnings(h, out, h->n_pairs);
  if (rc != MI355NDT_OK) {
    // an error exit may leave (update, sweep) rounds queued: drain them, so that no sweep of THIS align can post its progress words
    // into the flags the next align resets (the latency-mode pump restarts its sequence numbers at 1)
    (void)hipStreamSynchronize(h->stream);
    if (h->ev_compute) (void)compute_enqueued(h);                         // see mi355ndt_batch_build_targets
  }
  return rc;
}
static int ensure_seq_flags(mi355ndt_handle* h) {
  if (!h->h_seq_flags) {
    HIPCHK(h, hipHostMalloc((void**)&h->h_seq_flags, 64, hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: device writes are visible to the polling host
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->d_seq_flags, (void*)h->h_seq_flags, 0));
  }
  return MI355NDT_OK;
}

// Latency mode's Newton loop: (update, sweep) rounds enqueued at most `depth` ahead of the sweep the device last reported from;
// every fine sweep writes "pairs still active" and its sequence number into mapped host memory, so the loop needs neither the
// per-burst counter copy nor an event wait.  Ends when a sweep reports that no pair is active.
static int align_pump(mi355ndt_handle* h, SweepConst sc, int B) {
  int rc = ensure_seq_flags(h);
  if (rc) return rc;
  hipStream_t s = h->stream;
  h->h_seq_flags[0] = 0; h->h_seq_flags[1] = -1;
  sc.host_flags = h->d_seq_flags;
  sc.seq_no = 1;
  sc.rebase_block = 1;                           // every fine sweep also prepares the next update's re-basing (its extra workgroup)
  rc = launch_sweep(h, sc);                      // the sweep at the guess
  if (rc) return rc;
  const int depth = 2;
  const long long max_rounds = h->prm.max_iterations + 4;
  long long enq = 0;                             // (update, sweep) rounds enqueued; sweep of round r carries seq_no r + 1
  auto t_progress = std::chrono::steady_clock::now();
  long long seen_last = -1;
  for (;;) {
    const long long seen = h->h_seq_flags[0];    // sequence number of the last sweep that has started
    const int active = h->h_seq_flags[1];
    if (seen >= 1 && active == 0) break;         // that sweep found nothing to do: every pair is finalised
    if (seen != seen_last) { seen_last = seen; t_progress = std::chrono::steady_clock::now(); }
    if (enq >= max_rounds || enq + 1 - seen >= depth) {
      if (std::chrono::steady_clock::now() - t_progress > std::chrono::seconds(20)) { h->err = "align: the device stopped making progress"; return MI355NDT_ERR_STATE; }
      cpu_relax();                               // (busy-wait: a round is ~20 us, a yield costs more than it gives; pause frees the sibling hyperthread)
      continue;
    }
    k_update<<<B, UPD_THREADS, 0, s>>>(h->d_state, h->d_partials, h->rows_per_pair, h->pts_per_chunk, 1, h->d_results, h->d_active,
                                       h->d_active_list, h->d_ctl + h->ctl_idx, h->prof ? h->d_hits : nullptr, h->prm.step_size, h->prm.trans_epsilon,
                                       h->prm.max_iterations, 0, 0);
    sc.seq_no = (int)(enq + 2);
    rc = launch_sweep(h, sc);
    if (rc) return rc;
    enq++;
  }
  return MI355NDT_OK;
}

// One persistent launch for the whole batch align (ndt_async.hpp): served for the DIRECT / KDTREE sweeps of k_sweep with the dead
// More-Thuente loop -- every configuration lv_slam ships.  Returns MI355NDT_ERR_UNSUPPORTED when the launch cannot be made resident
// (the caller then takes the lockstep path).
}  // extern "C" (templates need C++ linkage)
// What one persistent launch needs besides the engine's parameters: the context table (one context: the synchronous batch align; several:
// the stream mode), the NEW context's arrays for the prepare kernel, the rings and the control blocks.
struct AsyncLaunch {
  AsyncTab tab;
  int new_ci = 0, n_new = 0;                       // context and number of the pairs that START in this launch (0: only carried pairs)
  PairState* st_new = nullptr; const float* guess_new = nullptr; const int* src_cnt_new = nullptr; const GridDesc* gd_new = nullptr; unsigned* arrived_new = nullptr;
  int* active_list = nullptr; SweepCtl* sweep_ctl = nullptr; unsigned* done_new = nullptr; PoseRecord* pose_new = nullptr; int pose_cap = 0;
  AsyncTab* tab_dev = nullptr; int* ring = nullptr; int ring_cap = 0; AsyncCtl* ctl = nullptr; const AsyncCtl* prev = nullptr;
  int items_per_pair = 0, stop_thresh = 0; unsigned debug_abort_pos = 0xFFFFFFFFu, debug_ring_mask = 0xFFu;
  unsigned long long* stamp_end = nullptr;       // stream mode + profiling: where k_async_prepare stamps the end of the build in front of it
  int claim_items = 1;                           // DIRECT7 items per claimed position (DIRECT1: always ASYNC_CLAIM(1) = 2; ndt_async.hpp)
  int reserve_wg = 0;
  hipEvent_t ev_prepared = nullptr;              // stream mode, build beside the launch: recorded between the prepare kernel and the persistent launch
};
#define NDT_CTX_ARGS(i) L.tab.c[i].src, L.tab.c[i].pitch, L.tab.c[i].st, L.tab.c[i].gd, L.tab.c[i].words, L.tab.c[i].recs, L.tab.c[i].partials, L.tab.c[i].src_cnt, \
                        L.tab.c[i].arrived, L.tab.c[i].cent
template <bool PCA, int K, int ORD>
static int launch_async_t(mi355ndt_handle* h, const SweepConst& sc, const AsyncLaunch& L) {
  auto kern = k_align_async<PCA, K, ORD>;
  // (asked once per instantiation and device: the query sits between the prepare kernel and the launch, on the host's critical path)
  // (engines on several host threads come through here at once: the cached answer is an atomic, the query writes into a local)
  static std::atomic<int> per_cu_of_device[64];
  int per_cu = per_cu_of_device[h->device & 63].load(std::memory_order_relaxed);
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SWEEP_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); return MI355NDT_ERR_UNSUPPORTED; }
    per_cu_of_device[h->device & 63].store(per_cu, std::memory_order_relaxed);
  }
  const int wpe = sweep_wpe(PCA, K, ORD == 2);
  // Workgroup L serves ring L % 8 first, and the launch is sized to be resident as a whole.  Residency is no condition of correctness:
  // positions are claimed, a waiting wave serves the published positions of OTHER rings too (ndt_async.hpp: an XCD that holds no workgroup
  // of this launch -- another engine's launch fills it -- leaves no ticket unserved), a workgroup that starts late finds the launch over
  // or joins in; a launch that cannot progress all the same ends itself (bounded polls) and the caller falls back to the rounds.
  if (per_cu < wpe || h->n_cu * wpe < 8) return MI355NDT_ERR_UNSUPPORTED;
  // (stream mode may withhold some workgroups so that the next batch's target build, on a stream of its own, finds wave slots
  //  beside this launch: L.reserve_wg, a multiple of 8 so that every ring loses the same number of waves)
  dim3 grid((unsigned)std::max(8, h->n_cu * wpe - L.reserve_wg));
  kern<<<grid, SWEEP_THREADS, 0, h->stream>>>(L.tab_dev, L.items_per_pair, L.ring, L.ring_cap, L.ctl, sc, h->prof ? h->d_hits : nullptr,
                                             h->prm.step_size, h->prm.trans_epsilon, h->prm.max_iterations, L.stop_thresh, L.debug_abort_pos, L.debug_ring_mask, L.claim_items,
                                             NDT_CTX_ARGS(0), NDT_CTX_ARGS(1), NDT_CTX_ARGS(2), NDT_CTX_ARGS(3));
  return MI355NDT_OK;
}
template <bool PCA, int K>
static int launch_async_o(mi355ndt_handle* h, const SweepConst& sc, const AsyncLaunch& L) {
  if constexpr (K == 1 || K == 7) { if (sweep_ord(h, sc) == 2) return launch_async_t<PCA, K, 2>(h, sc, L); }
  return h->f32_sum_order == 1 ? launch_async_t<PCA, K, 1>(h, sc, L) : launch_async_t<PCA, K, 0>(h, sc, L);
}
// prepare kernel + the persistent launch on the engine's stream (HIP events around the launch when profiling)
static int launch_async(mi355ndt_handle* h, const SweepConst& sc, const AsyncLaunch& L) {
  hipStream_t s = h->stream;
  {
    const size_t n = std::max(std::max(std::max((size_t)8 * L.ring_cap, (size_t)L.n_new * ASYNC_ARR_STRIDE), sizeof(AsyncCtl) / sizeof(unsigned)), (size_t)L.pose_cap);
    k_async_prepare<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(L.tab, L.tab_dev, L.new_ci, L.n_new, L.st_new, L.guess_new, L.src_cnt_new, L.gd_new, L.arrived_new,
                                                               L.active_list, L.sweep_ctl, L.ring, L.ring_cap, L.ctl, L.prev, L.done_new, L.pose_new, L.pose_cap, L.stamp_end);
    if (L.ev_prepared) HIPCHK(h, hipEventRecord(L.ev_prepared, s));
  }
  if (h->prof) HIPCHK(h, ev_begin(h, h->ev_sweep));
  int rc;
  if (sc.pca) rc = sc.K == 1 ? launch_async_o<true, 1>(h, sc, L) : sc.K == 7 ? launch_async_o<true, 7>(h, sc, L) : launch_async_o<true, 26>(h, sc, L);
  else rc = sc.K == 1 ? launch_async_o<false, 1>(h, sc, L) : sc.K == 7 ? launch_async_o<false, 7>(h, sc, L)
          : sc.K == 26 ? launch_async_o<false, 26>(h, sc, L) : launch_async_o<false, 27>(h, sc, L);
  if (h->prof) HIPCHK(h, ev_end(h, h->ev_sweep));
  return rc;
}
// ring slots a launch over at most `pairs` pairs can need: every pair publishes at most max_iterations + 3 tickets.
// 0: too many for a sane allocation (the caller takes the round-based path, which needs no ring)
static int async_ring_cap(const mi355ndt_handle* h, long long pairs) {
  const long long cap = (pairs * ((long long)h->prm.max_iterations + 4) + 7) / 8 + 1;
  return cap > (1LL << 26) ? 0 : (int)cap;        // 8 rings x 2^26 words = 2 GB: beyond that the rounds are the right tool anyway
}
static void fill_async_ctx(const mi355ndt_handle* e, AsyncCtx& c) {
  SweepConst sc;
  make_sweep_const(e, sc);
  c.src = e->d_src; c.pitch = e->src_pitch; c.st = e->d_state; c.gd = e->d_grid; c.words = e->d_words; c.recs = sweep_recs(e, sc); c.cent = e->d_cent;
  c.partials = e->d_partials; c.src_cnt = e->d_src_cnt; c.arrived = e->d_arrived; c.results = e->d_results; c.n_done = nullptr; c.must_finish = 1; c.pose = nullptr; c.pose_base = 0; c.pose_stride = 0; c.pad_ = 0;
}
extern "C" {
static int align_async(mi355ndt_handle* h, const SweepConst& sc, int B, mi355ndt_result* out) {
  hipStream_t s = h->stream;
  const int ring_cap = async_ring_cap(h, B);
  if (ring_cap == 0) return MI355NDT_ERR_UNSUPPORTED;
  // (a ring that cannot be allocated is no error of the align: the round-based path needs none)
  if (grow(h->d_ring, h->ring_cap_total, (size_t)8 * ring_cap) != hipSuccess) { (void)hipGetLastError(); return MI355NDT_ERR_UNSUPPORTED; }
  HIPCHK(h, grow(h->d_arrived, h->arrived_cap, (size_t)B * ASYNC_ARR_STRIDE));
  if (!h->d_actl) HIPCHK(h, hipMalloc((void**)&h->d_actl, sizeof(AsyncCtl)));
  if (!h->d_atab) HIPCHK(h, hipMalloc((void**)&h->d_atab, sizeof(AsyncTab)));
  if (!h->h_pin_actl) HIPCHK(h, hipHostMalloc((void**)&h->h_pin_actl, sizeof(AsyncCtl)));
  // everything the launch polls is reset on the stream before it (never inside the kernel, never by a previous launch), together with
  // the pairs' initial states
  AsyncLaunch L;
  memset(&L.tab, 0, sizeof L.tab);
  fill_async_ctx(h, L.tab.c[0]);
  L.new_ci = 0; L.n_new = B;
  L.st_new = h->d_state; L.guess_new = h->d_guess; L.src_cnt_new = h->d_src_cnt; L.gd_new = h->d_grid; L.arrived_new = h->d_arrived;
  L.active_list = h->d_active_list; L.sweep_ctl = h->d_ctl;
  L.tab_dev = h->d_atab; L.ring = h->d_ring; L.ring_cap = ring_cap; L.ctl = h->d_actl; L.prev = nullptr;
  L.items_per_pair = h->items_per_pair; L.stop_thresh = 0; L.debug_abort_pos = h->debug_abort_pos; L.debug_ring_mask = h->debug_ring_mask;
  int rc = launch_async(h, sc, L);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->h_pin_actl, h->d_actl, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, s));   // pub, fin, abort_, n_live, susp
  HIPCHK(h, hipMemcpyAsync(out, h->d_results, (size_t)B * sizeof(mi355ndt_result), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  if (h->h_pin_actl->abort_ || h->h_pin_actl->fin != (unsigned)B) {
    // a wave gave up (its ticket never came within the poll budget: a device shared with something that starves the launch, or the
    // test hook): nothing is lost -- the round-based path below produces the same bits from the same guesses
    h->P.async_fallbacks++;
    return MI355NDT_ERR_UNSUPPORTED;
  }
  if (h->prof) {                                   // every sweep a pair took part in streamed its points + K table probes
    for (int b = 0; b < B; b++) {
      h->P.sweep_alg_bytes += (double)out[b].sweeps * h->h_src_cnt[b] * (12.0 + 4.0 * sc.K);
      h->P.sweep_points += (long long)out[b].sweeps * h->h_src_cnt[b];
    }
  }
  return MI355NDT_OK;
}

static int batch_align_impl(mi355ndt_handle* h, const float* guesses, mi355ndt_result* out) {
  if (!guesses || !out) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs <= 0 || !h->d_tgt || !h->d_src) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcu = uploads_before_compute(h); if (rcu) return rcu; }
  const bool mt_live = mt_is_live(h->prm);                       // impl2:888: More-Thuente loop + computeHessian are live
  const bool pca_kd = h->prm.neighbor_mode == MI355NDT_KDTREE && h->prm.variant == MI355NDT_VARIANT_PCA;
  if (!h->targets_built || (mt_live && !h->icov64_built) || (pca_kd && !h->kdw_built) || (fast_served(h) && !h->recs_fast_built)) { int rc = mi355ndt_batch_build_targets(h); if (rc) return rc; }
  int rc = prep_align_ws(h);
  if (rc) return rc;
  const int B = h->n_pairs;
  hipStream_t s = h->stream;
  SweepConst sc;
  make_sweep_const(h, sc);
  gauss_constants3(h->prm.outlier_ratio, h->prm.resolution, h->gauss_last);     // computeTransformation sets the members (impl2:93-100)
  // the engine's stream is idle here (every entry point returns synchronised), so the pinned staging copy is free to overwrite
  h->ev_last_fresh = false;
  memcpy(h->h_pin_guess, guesses, (size_t)B * 16 * sizeof(float));
  HIPCHK(h, hipMemcpyAsync(h->d_guess, h->h_pin_guess, (size_t)B * 16 * sizeof(float), hipMemcpyHostToDevice, s));
  h->ctl_idx = 0;
  // One launch for the whole align (ndt_async.hpp) when the batch offers more work items than the GPU has resident waves.  A smaller batch
  // -- a single registration above all -- keeps the round-based kernels, whose flat dealing spreads a pair's items over every XCD: a ticket
  // is served by ONE ring (an eighth of the waves), which costs a lone 65,536-point pair 0.39 ms against 0.31 ms per align.
  const bool big_batch = (long long)B * h->items_per_pair > (long long)h->n_cu * sweep_wpe(sc.pca != 0, sc.K, sweep_ord(h, sc) == 2) * WAVES;
  if (h->async_align && (big_batch || h->async_force) && !h->fine_it && !mt_live && !pca_kd) {
    rc = align_async(h, sc, B, out);                                // (prepares the pair states itself: k_async_prepare)
    if (rc == MI355NDT_OK) { h->aligned_once = true; return MI355NDT_OK; }
    if (rc != MI355NDT_ERR_UNSUPPORTED) return rc;                  // (not resident, no ring, or the launch gave up: the lockstep rounds below)
  }
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, 2 * sizeof(SweepCtl), s));
  k_init_state<<<(B + 63) / 64, 64, 0, s>>>(h->d_state, h->d_guess, h->d_src_cnt, h->d_grid, B, h->d_active_list, h->d_ctl);
  if (h->fine_it) {                                // latency mode: the pump (no bursts, no counter copies, no event waits)
    rc = align_pump(h, sc, B);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(out, h->d_results, (size_t)B * sizeof(mi355ndt_result), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    if (h->prof) {                                 // every sweep a pair took part in streamed its points + K table probes
      for (int b = 0; b < B; b++) {
        h->P.sweep_alg_bytes += (double)out[b].sweeps * h->h_src_cnt[b] * (12.0 + 4.0 * sc.K);
        h->P.sweep_points += (long long)out[b].sweeps * h->h_src_cnt[b];
      }
    }
    h->aligned_once = true;
    return MI355NDT_OK;
  }
  rc = launch_sweep(h, sc);
  if (rc) return rc;
  const int max_rounds = h->prm.max_iterations + 4;   // loop body runs for it = 0 .. max_iterations+1
  double pts_total = 0;
  for (int b = 0; b < B; b++) pts_total += h->h_src_cnt[b];
  const double alg_static = pts_total * (12.0 + 4.0 * sc.K);   // every active pair streams its points + K table probes
  if (h->prof) {
    h->P.sweep_alg_bytes += alg_static;                          // the initial sweep covers all pairs
    h->P.sweep_points += (long long)pts_total;
  }
  // update+sweep rounds are enqueued in bursts of two; the host always keeps ONE burst queued ahead of the one whose
  // "pairs still active" counters it is waiting for, so the device never idles over a host round trip.  The price is
  // at most one speculative burst after the last pair finished (k_update / k_sweep return at once with nothing active).
  const int burst = 2;
  int round = 0, n_enq = 0;
  int cnt[2] = {0, 0};                                           // rounds in the burst held by ring slot 0 / 1
  auto enqueue_burst = [&]() -> int {
    const int slot = n_enq & 1;
    h->ev_last_fresh = false;                    // the burst bookkeeping below sits between the previous sweep and this update
    int* dact = h->d_active + slot * burst;
    hipError_t e = hipMemsetAsync(dact, 0, burst * sizeof(int), s);
    if (e != hipSuccess) return MI355NDT_ERR_HIP;
    int k = 0;
    for (; k < burst && round < max_rounds; k++, round++) {
      if (h->prof) HIPCHK(h, ev_begin(h, h->ev_update));
      k_update<<<B, UPD_THREADS, 0, s>>>(h->d_state, h->d_partials, h->rows_per_pair, h->pts_per_chunk, h->fine_it ? 1 : 0, h->d_results, dact + k,
                                h->d_active_list, h->d_ctl + h->ctl_idx, h->prof ? h->d_hits : nullptr,
                                h->prm.step_size, h->prm.trans_epsilon, h->prm.max_iterations, 0, mt_live ? 1 : 0);
      if (mt_live) {      // pairs whose More-Thuente loop iterated get their Hessian from computeHessian (impl2:999-1000)
        launch_hessian(h, sc);
        k_update<<<B, UPD_THREADS, 0, s>>>(h->d_state, h->d_partials, h->rows_per_pair, h->pts_per_chunk, h->fine_it ? 1 : 0, h->d_results, dact + k,
                                  h->d_active_list, h->d_ctl + h->ctl_idx, nullptr,
                                  h->prm.step_size, h->prm.trans_epsilon, h->prm.max_iterations, 0, 2);
      }
      if (h->prof) HIPCHK(h, ev_end(h, h->ev_update));
      int r = launch_sweep(h, sc);
      if (r) return r;
    }
    cnt[slot] = k;
    HIPCHK(h, hipMemcpyAsync(h->h_pin_active + slot * burst, dact, burst * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipEventRecord(h->ev_burst[slot], s));
    n_enq++;
    return MI355NDT_OK;
  };
  rc = enqueue_burst();
  if (rc) return rc;
  for (int done = 0; done < n_enq; done++) {
    if (round < max_rounds) { rc = enqueue_burst(); if (rc) return rc; }     // speculative: one burst ahead
    const int slot = done & 1;
    HIPCHK(h, hipEventSynchronize(h->ev_burst[slot]));
    const int* act = h->h_pin_active + slot * burst;
    if (h->prof) {
      // sweep k of this burst streamed the pairs that scheduled a step in update k (equal-size pairs assumed)
      for (int k = 0; k < cnt[slot]; k++) {
        const double frac = (double)act[k] / B;
        h->P.sweep_alg_bytes += alg_static * frac;
        h->P.sweep_points += (long long)(pts_total * frac);
      }
    }
    if (act[cnt[slot] - 1] == 0) break;
  }
  // (the device-side hit counter d_hits keeps accumulating; mi355ndt_profile_get reads it -- every sweep is followed by an
  //  update, which is where the hits are added, so nothing is missing when the loop exits)
  HIPCHK(h, hipMemcpyAsync(out, h->d_results, (size_t)B * sizeof(mi355ndt_result), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  h->aligned_once = true;
  return MI355NDT_OK;
}

int mi355ndt_batch_pose_records(mi355ndt_handle* h, int id_base, int id_stride, void* d_records, size_t capacity) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!d_records || capacity == 0 || capacity > (size_t)MAX_PAIRS || (size_t)h->n_pairs > capacity) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs <= 0 || !h->d_results || !h->aligned_once) return MI355NDT_ERR_STATE;   // no align of this batch yet: nothing to pack
  HIPCHK(h, hipSetDevice(h->device));
  static_assert(sizeof(PoseRecord) == 96, "pose record is 96 bytes");
  k_pose_records<<<(unsigned)((capacity + 255) / 256), 256, 0, h->stream>>>(h->d_results, h->n_pairs, id_base, id_stride, (PoseRecord*)d_records, (int)capacity);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return MI355NDT_OK;
}

// ---- single-registration surface (pair slot 0) ---------------------------------------------------
static int ensure_single(mi355ndt_handle* h, bool tgt, size_t n) {
  const size_t want = std::max(((n + 63) & ~(size_t)63), (size_t)64);
  const bool single = h->n_pairs == 1 && h->d_tgt_own && h->d_src_own && h->d_tgt == h->d_tgt_own && h->d_src == h->d_src_own;
  if (!single) {
    // leaving batch / bound mode: start a fresh one-pair engine
    return mi355ndt_batch_reserve(h, 1, tgt ? want : 64, tgt ? 64 : want);
  }
  // target and source buffers are independent: grow only the side being replaced
  const size_t have = tgt ? h->own_tgt_pitch : h->own_src_pitch;
  if (n <= have) return MI355NDT_OK;
  int rc = alloc_side(h, tgt, 1, want);
  if (rc) return rc;
  if (tgt) h->tgt_pitch = want; else h->src_pitch = want;
  h->d_tgt = h->d_tgt_own; h->d_src = h->d_src_own;
  return MI355NDT_OK;
}

int mi355ndt_set_target(mi355ndt_handle* h, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if ((!pts && n) || (n && stride < 12) || n >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_single(h, true, n);
  if (rc) return rc;
  rc = mi355ndt_batch_set_target(h, 0, pts, n, stride);
  if (rc) return rc;
  h->have_target = true;
  return mi355ndt_batch_build_targets(h);      // init(): filter(true) (ndt_omp.h:270-277)
}

int mi355ndt_set_source(mi355ndt_handle* h, const void* pts, size_t n, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if ((!pts && n) || (n && stride < 12) || n >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_single(h, false, n);
  if (rc) return rc;
  rc = mi355ndt_batch_set_source(h, 0, pts, n, stride);
  if (rc) return rc;
  h->have_source = true;
  return MI355NDT_OK;
}

int mi355ndt_set_params(mi355ndt_handle* h, const mi355ndt_params* p) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if (!p) return MI355NDT_ERR_BAD_ARG;
  int rc = check_params(*p);
  if (rc) return rc;
  const mi355ndt_params old = h->prm;
  h->prm = *p;
  // setResolution (ndt_omp.h:126-136): `if (resolution_ != resolution) { resolution_ = resolution; if (input_) init(); }` -- the grid is only
  // re-made when a SOURCE cloud is set; without one it keeps its leaf size until the next setInputTarget, while the Gauss constants follow
  // the new value (impl2:93-100).  Reproduced for the DIRECT searches of the single-registration surface; a radius search over the grid
  // (KDTREE, live More-Thuente) is emulated by a 27-cell probe that needs radius <= leaf, so those re-voxelise as before (documented deviation).
  const bool radius_search = p->neighbor_mode == MI355NDT_KDTREE || mt_is_live(*p);
  const bool keep_grid = old.resolution != p->resolution && h->n_pairs == 1 && !h->have_source && !radius_search && h->d_tgt == h->d_tgt_own;
  const bool regrid = (old.resolution != p->resolution && !keep_grid) || old.variant != p->variant ||
                      old.min_points_per_voxel != p->min_points_per_voxel ||
                      old.min_covar_eigvalue_mult != p->min_covar_eigvalue_mult ||
                      ((p->neighbor_mode == MI355NDT_KDTREE || mt_is_live(*p)) && !h->cent_built) ||   // centroids the build skipped
                      (mt_is_live(*p) && !h->icov64_built) ||
                      (p->neighbor_mode == MI355NDT_KDTREE && p->variant == MI355NDT_VARIANT_PCA && !h->kdw_built);
  if (regrid && h->targets_built) {
    h->targets_built = false;
    rc = mi355ndt_batch_build_targets(h);     // setResolution -> init() (ndt_omp.h:126-136)
    if (rc) h->prm = old;                     // the grids were not rebuilt: keep the parameters they were (last) built with;
    return rc;                                // targets_built stays false, so the next align re-voxelises
  }
  return MI355NDT_OK;
}

int mi355ndt_align(mi355ndt_handle* h, const float guess[16], mi355ndt_result* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!guess || !out) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs < 1 || !h->have_target || !h->have_source) return MI355NDT_ERR_STATE;
  if (h->n_pairs != 1) return MI355NDT_ERR_STATE;                   // a batch is bound: use mi355ndt_batch_align
  // pcl::Registration::initCompute() refuses empty clouds; align() then returns without touching converged_
  if (h->h_tgt_cnt[0] <= 0 || h->h_src_cnt[0] <= 0) return MI355NDT_ERR_STATE;
  int rc = mi355ndt_batch_align(h, guess, out);
  if (rc == MI355NDT_OK) memcpy(h->last_final, out->final_colmajor, sizeof h->last_final);
  return rc;
}

int mi355ndt_get_aligned(mi355ndt_handle* h, void* out_pts, size_t stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!out_pts || stride < 12) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs < 1 || !h->d_src) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcu = uploads_before_compute(h); if (rcu) return rcu; }
  const int n = h->h_src_cnt[0];
  if (n == 0) return MI355NDT_OK;
  HIPCHK(h, grow(h->d_aligned, h->aligned_cap, (size_t)3 * n));
  if ((size_t)3 * n > h->pin_aligned_cap) {
    if (h->h_pin_aligned) { HIPCHK(h, hipHostFree(h->h_pin_aligned)); h->h_pin_aligned = nullptr; h->pin_aligned_cap = 0; }
    HIPCHK(h, hipHostMalloc((void**)&h->h_pin_aligned, (size_t)3 * n * sizeof(float)));
    h->pin_aligned_cap = (size_t)3 * n;
  }
  // moved cloud as packed x,y,z triples -> pinned memory -> x,y,z of the caller's records (their other fields are left alone)
  k_transform<<<(n + 255) / 256, 256, 0, h->stream>>>(h->d_src, h->src_pitch, h->d_state, 0, h->d_aligned, n);
  HIPCHK(h, hipMemcpyAsync(h->h_pin_aligned, h->d_aligned, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unsigned char* o = (unsigned char*)out_pts;
  if (stride == 12) memcpy(o, h->h_pin_aligned, (size_t)n * 12);
  else for (int i = 0; i < n; i++) memcpy(o + (size_t)i * stride, h->h_pin_aligned + (size_t)3 * i, 12);
  return MI355NDT_OK;
}

int mi355ndt_get_incremental(mi355ndt_handle* h, int pair, float last[16], float prev[16]) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs) return MI355NDT_ERR_BAD_ARG;
  if (!h->d_state) return MI355NDT_ERR_STATE;
  if (!h->aligned_once) {                         // before any align(): transformation_ = previous_transformation_ = Identity
    for (int a = 0; a < 16; a++) { const float v = (a % 5 == 0) ? 1.f : 0.f; if (last) last[a] = v; if (prev) prev[a] = v; }
    return MI355NDT_OK;
  }
  HIPCHK(h, hipSetDevice(h->device));
  float buf[32];
  HIPCHK(h, hipMemcpyAsync(buf, (const char*)(h->d_state + pair) + offsetof(PairState, inc_cm), sizeof buf, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (last) memcpy(last, buf, 16 * sizeof(float));
  if (prev) memcpy(prev, buf + 16, 16 * sizeof(float));
  return MI355NDT_OK;
}

static int run_hook_sweep(mi355ndt_handle* h, double* score, double g[6], double H[36], long long* hits) {
  SweepConst sc;
  make_sweep_const(h, sc);
  int rc = launch_sweep(h, sc);
  if (rc) return rc;
  k_update<<<1, UPD_THREADS, 0, h->stream>>>(h->d_state, h->d_partials, h->rows_per_pair, h->pts_per_chunk, h->fine_it ? 1 : 0, h->d_results, h->d_active, h->d_active_list, h->d_ctl,
                                    nullptr, 0, 0, 0, 1, 0);
  PairState S;
  HIPCHK(h, hipMemcpyAsync(&S, h->d_state, sizeof(PairState), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  if (score) *score = S.score;
  if (g) memcpy(g, S.g, sizeof S.g);
  if (H) memcpy(H, S.H, sizeof S.H);
  if (hits) *hits = S.hits;
  return MI355NDT_OK;
}

static int hook_ready(mi355ndt_handle* h) {
  h->ev_last_fresh = false;
  if (h->n_pairs < 1 || !h->have_target || !h->have_source) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcu = uploads_before_compute(h); if (rcu) return rcu; }
  const bool pca_kd = h->prm.neighbor_mode == MI355NDT_KDTREE && h->prm.variant == MI355NDT_VARIANT_PCA;
  if (!h->targets_built || (pca_kd && !h->kdw_built) || (fast_served(h) && !h->recs_fast_built)) { int rc = mi355ndt_batch_build_targets(h); if (rc) return rc; }
  return prep_align_ws(h);
}

int mi355ndt_derivatives(mi355ndt_handle* h, const double p[6], double* score, double g[6], double H[36], long long* hits) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!p) return MI355NDT_ERR_BAD_ARG;
  int rc = hook_ready(h);
  if (rc) return rc;
  double* dp = (double*)h->d_hook;
  HIPCHK(h, hipMemcpyAsync(dp, p, 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, 2 * sizeof(SweepCtl), h->stream));
  h->ctl_idx = 0;
  k_set_pose_p<<<1, 1, 0, h->stream>>>(h->d_state, 0, dp, h->d_src_cnt, h->d_grid, h->d_active_list, h->d_ctl, 0);
  return run_hook_sweep(h, score, g, H, hits);
}

int mi355ndt_compute_hessian(mi355ndt_handle* h, const double p[6], double H[36]) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!p || !H) return MI355NDT_ERR_BAD_ARG;
  int rc = hook_ready(h);
  if (rc) return rc;
  if (!h->cent_built || !h->icov64_built) {                       // the grid was built for a configuration that never needs it
    const mi355ndt_params keep = h->prm;
    h->prm.step_size = 0; h->prm.trans_epsilon = 0;               // "live" build flavour: centroids + f64 inverse covariances
    rc = mi355ndt_batch_build_targets(h);
    h->prm = keep;
    if (rc) return rc;
  }
  h->fine_it = 0; h->rows_per_pair = h->items_per_pair = h->chunks_per_pair * QUARTERS; h->pts_per_chunk = CHUNK_PTS;   // k_hessian writes batch-mode rows
  double* dp = (double*)h->d_hook;
  HIPCHK(h, hipMemcpyAsync(dp, p, 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, 2 * sizeof(SweepCtl), h->stream));
  h->ctl_idx = 0;
  k_set_pose_p<<<1, 1, 0, h->stream>>>(h->d_state, 0, dp, h->d_src_cnt, h->d_grid, h->d_active_list, h->d_ctl, 1);
  SweepConst sc;
  make_sweep_const(h, sc);
  launch_hessian(h, sc);
  k_update<<<1, UPD_THREADS, 0, h->stream>>>(h->d_state, h->d_partials, h->rows_per_pair, h->pts_per_chunk, h->fine_it ? 1 : 0, h->d_results, h->d_active, h->d_active_list, h->d_ctl,
                                    nullptr, 0, 0, 0, 1, 2);
  PairState S;
  HIPCHK(h, hipMemcpyAsync(&S, h->d_state, sizeof(PairState), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  memcpy(H, S.H, sizeof S.H);
  return MI355NDT_OK;
}

int mi355ndt_derivatives_T(mi355ndt_handle* h, const float T[16], const float Rj[9], double* score, double g[6], double H[36], long long* hits) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!T || !Rj) return MI355NDT_ERR_BAD_ARG;
  int rc = hook_ready(h);
  if (rc) return rc;
  float buf[25];
  memcpy(buf, T, 16 * sizeof(float));
  memcpy(buf + 16, Rj, 9 * sizeof(float));
  HIPCHK(h, hipMemcpyAsync(h->d_hook, buf, sizeof buf, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, 2 * sizeof(SweepCtl), h->stream));
  h->ctl_idx = 0;
  k_set_pose<<<1, 1, 0, h->stream>>>(h->d_state, 0, h->d_hook, h->d_hook + 16, h->d_src_cnt, h->d_grid, h->d_active_list, h->d_ctl);
  return run_hook_sweep(h, score, g, H, hits);
}

int mi355ndt_get_grid(mi355ndt_handle* h, int pair, int min_b[3], int max_b[3], int div_b[3], int* n_voxels) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs) return MI355NDT_ERR_BAD_ARG;
  if (!h->targets_built) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  GridDesc g;
  HIPCHK(h, hipMemcpyAsync(&g, h->d_grid + pair, sizeof g, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int a = 0; a < 3; a++) {
    if (min_b) min_b[a] = g.min_b[a];
    if (max_b) max_b[a] = g.max_b[a];
    if (div_b) div_b[a] = g.div_b[a];
  }
  if (n_voxels) *n_voxels = g.n_voxels;
  return (g.status == GRID_OVERFLOW || g.status == GRID_CAP) ? MI355NDT_ERR_GRID : MI355NDT_OK;
}

int mi355ndt_get_voxels(mi355ndt_handle* h, int pair, mi355ndt_voxel* out, size_t capacity) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (pair < 0 || pair >= h->n_pairs || (!out && capacity)) return MI355NDT_ERR_BAD_ARG;
  if (!h->targets_built) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  GridDesc g;
  HIPCHK(h, hipMemcpyAsync(&g, h->d_grid + pair, sizeof g, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  size_t n = std::min((size_t)g.n_voxels, capacity);
  if (n == 0) return MI355NDT_OK;
  std::vector<VoxelRec> r(n);
  std::vector<int> idx(n), cnt(n);
  HIPCHK(h, hipMemcpy(r.data(), h->d_recs + g.rec_off, n * sizeof(VoxelRec), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(idx.data(), h->d_vox_idx + g.rec_off, n * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(cnt.data(), h->d_vox_n + g.rec_off, n * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) {
    out[i].idx = idx[i];
    out[i].n = cnt[i];
    memcpy(out[i].mean, r[i].mean, sizeof r[i].mean);
    memcpy(out[i].icov, r[i].icov, sizeof r[i].icov);
    out[i].weight = (r[i].weight == VOX_DEAD) ? 0 : r[i].weight;
  }
  return MI355NDT_OK;
}

// replaces pcl::Registration::getFitnessScore(max_range) for the loop-closure caller (loop_detector.hpp:249-262)
int mi355ndt_fitness_score_T(mi355ndt_handle* h, const float T_colmajor[16], double max_range, double* score, long long* n_inliers) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!T_colmajor || !score) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs != 1 || !h->have_target || !h->have_source) return MI355NDT_ERR_STATE;
  if (h->h_tgt_cnt[0] <= 0 || h->h_src_cnt[0] <= 0) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  { int rcu = uploads_before_compute(h); if (rcu) return rcu; }
  if (!h->targets_built) { int rc = mi355ndt_batch_build_targets(h); if (rc) return rc; }
  hipStream_t s = h->stream;
  GridDesc g;
  HIPCHK(h, hipMemcpyAsync(&g, h->d_grid, sizeof g, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  if (g.status == GRID_EMPTY) { *score = 1.7976931348623157e308; if (n_inliers) *n_inliers = 0; return MI355NDT_OK; }
  const bool brute = g.status != GRID_OK;        // no voxel grid (leaf-too-small guard / cell cap): the score does not need one
  if (!brute && !h->cells_ready) {
    const size_t nc = (size_t)g.ncells;
    if (nc > h->cell_cap) {
      size_t c1 = 0, c2 = 0;
      if (h->d_cstart) { HIPCHK(h, hipFree(h->d_cstart)); h->d_cstart = nullptr; }
      if (h->d_cend) { HIPCHK(h, hipFree(h->d_cend)); h->d_cend = nullptr; }
      HIPCHK(h, grow(h->d_cstart, c1, nc)); HIPCHK(h, grow(h->d_cend, c2, nc));
      h->cell_cap = nc;
    }
    HIPCHK(h, hipMemsetAsync(h->d_cstart, 0, nc * sizeof(unsigned), s));
    HIPCHK(h, hipMemsetAsync(h->d_cend, 0, nc * sizeof(unsigned), s));
    k_cellrange<unsigned><<<(unsigned)((h->tgt_pitch + 255) / 256), 256, 0, s>>>(h->d_keys_b, h->tgt_pitch, h->last_cb,
                                                                                h->d_cstart, h->d_cend);
    h->cells_ready = true;
  }
  const int n = h->h_src_cnt[0];
  const int blocks = (n + 255) / 256;
  HIPCHK(h, grow(h->d_fit, h->fit_cap, (size_t)2 * blocks));
  HIPCHK(h, hipMemcpyAsync(h->d_hook, T_colmajor, 16 * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipStreamSynchronize(s));
  const float mr = max_range >= 3.0e38 ? 3.0e38f : (float)max_range;
  // rings needed to cover sqrt(max_range) (+1 cell of slack), capped by the grid's extent
  const int extent = std::max(g.div_b[0], std::max(g.div_b[1], g.div_b[2])) + 2;
  double rr = brute ? 0.0 : std::sqrt(std::min(max_range, 1e30)) / (double)g.leaf + 2.0;   // (a target without a grid has no leaf size to divide by)
  // (a query outside the grid may sit further away than the grid is wide: the kernel clamps its cell to 2^29 cells from the grid's
  //  origin, so 2^30 rings reach every target cell from anywhere)
  const int ring_max = rr > (double)(1 << 30) ? (1 << 30) : (int)rr;
  (void)extent;
  if (brute) k_fitness_brute<<<blocks, 256, 0, s>>>(h->d_src, h->src_pitch, n, h->d_tgt, h->tgt_pitch, h->h_tgt_cnt[0], h->d_hook, mr, h->d_fit);
  else k_fitness<<<blocks, 256, 0, s>>>(h->d_src, h->src_pitch, n, h->d_tgt, h->tgt_pitch, h->d_vals_b, h->d_grid, h->d_cstart, h->d_cend,
                                        h->d_hook, mr, ring_max, h->d_fit);
  std::vector<double> part((size_t)2 * blocks);
  HIPCHK(h, hipMemcpyAsync(part.data(), h->d_fit, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  double sum = 0, cnt = 0;
  for (int b = 0; b < blocks; b++) { sum += part[2 * b]; cnt += part[2 * b + 1]; }
  *score = cnt > 0 ? sum / cnt : 1.7976931348623157e308;     // std::numeric_limits<double>::max()
  if (n_inliers) *n_inliers = (long long)cnt;
  return MI355NDT_OK;
}

int mi355ndt_get_fitness_score(mi355ndt_handle* h, double max_range, double* score, long long* n_inliers) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  return mi355ndt_fitness_score_T(h, h->last_final, max_range, score, n_inliers);
}

// replaces calculateScore(cloud) (ndt_omp.h:232, ndt_omp_impl2.hpp:1006-1040)
int mi355ndt_calculate_score(mi355ndt_handle* h, const void* pts, size_t n, size_t stride, double* score) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!score || (!pts && n) || (n && stride < 12) || n >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  if (h->n_pairs != 1 || !h->have_target || h->h_tgt_cnt[0] <= 0) return MI355NDT_ERR_STATE;
  if (n == 0) { *score = std::nan(""); return MI355NDT_OK; }                     // 0 / 0 in the reference
  HIPCHK(h, hipSetDevice(h->device));
  { int rcu = uploads_before_compute(h); if (rcu) return rcu; }
  if (!h->targets_built || !h->cent_built || !h->icov64_built) {                // f32 centroids + f64 inverse covariances: the "live" build flavour
    const mi355ndt_params keep = h->prm;
    h->prm.step_size = 0; h->prm.trans_epsilon = 0;
    const int rc = mi355ndt_batch_build_targets(h);
    h->prm = keep;
    if (rc) return rc;
  }
  const size_t pitch = (n + 63) & ~(size_t)63;
  HIPCHK(h, grow(h->d_score_pts, h->score_pts_cap, 3 * pitch));
  const int blocks = (int)((n + SCORE_THREADS - 1) / SCORE_THREADS);
  HIPCHK(h, grow(h->d_score_part, h->score_part_cap, (size_t)blocks));
  int rc = upload_cloud(h, h->d_score_pts, pitch, 0, pts, n, stride);
  if (rc) return rc;
  rc = uploads_before_compute(h);
  if (rc) return rc;
  SweepConst sc;
  make_sweep_const(h, sc);
  hipStream_t s = h->stream;
  k_calc_score<<<blocks, SCORE_THREADS, 0, s>>>(h->d_score_pts, pitch, (int)n, h->d_grid, h->d_words, h->d_recs, h->d_icov64, h->d_cent,
                                                h->gauss_last[0], h->gauss_last[1], h->gauss_last[2], sc.kd_r2, sc.leaf_pow2, sc.inv_leaf, h->d_score_part);
  std::vector<double> part((size_t)blocks);
  HIPCHK(h, hipMemcpyAsync(part.data(), h->d_score_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  double sum = 0;
  for (int b = 0; b < blocks; b++) sum += part[(size_t)b];
  *score = sum / (double)n;                                                      // impl2:1040
  return compute_enqueued(h);
}

// replaces the static convertTransform helpers (ndt_omp.h:209-228); f32, the way Eigen 3.3 evaluates
// Translation3f * AngleAxisf(X) * AngleAxisf(Y) * AngleAxisf(Z) (third-party, restated from its published algorithm: AngleAxis::toRotationMatrix,
// Transform::rotate = linear() * R with coefficient-wise 3x3 products, a 3-term sum reduced as t0 + (t1 + t2))
static void aa_matrix(float angle, int axis, float R[9]) {
  const float ax[3] = {axis == 0 ? 1.f : 0.f, axis == 1 ? 1.f : 0.f, axis == 2 ? 1.f : 0.f};
  const float sn = sinf(angle), c = cosf(angle);
  const float sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
  const float c1[3] = {(1.f - c) * ax[0], (1.f - c) * ax[1], (1.f - c) * ax[2]};
  float tmp = c1[0] * ax[1];
  R[0 * 3 + 1] = tmp - sa[2]; R[1 * 3 + 0] = tmp + sa[2];
  tmp = c1[0] * ax[2];
  R[0 * 3 + 2] = tmp + sa[1]; R[2 * 3 + 0] = tmp - sa[1];
  tmp = c1[1] * ax[2];
  R[1 * 3 + 2] = tmp - sa[0]; R[2 * 3 + 1] = tmp + sa[0];
  for (int a = 0; a < 3; a++) R[a * 3 + a] = c1[a] * ax[a] + c;
}
static void mul33(const float A[9], const float B[9], float C[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + (A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c]);
}
int mi355ndt_convert_transform(const double x[6], float out[16]) {
  if (!x || !out) return MI355NDT_ERR_BAD_ARG;
  float Rx[9], Ry[9], Rz[9], A[9], L[9];
  aa_matrix((float)x[3], 0, Rx); aa_matrix((float)x[4], 1, Ry); aa_matrix((float)x[5], 2, Rz);
  mul33(Rx, Ry, A);
  mul33(A, Rz, L);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) out[c * 4 + r] = L[r * 3 + c];
    out[12 + r] = (float)x[r];
    out[r * 4 + 3] = 0.f;
  }
  out[15] = 1.f;
  return MI355NDT_OK;
}

int mi355ndt_set_option(mi355ndt_handle* h, int option, int value) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  // (what a stream's launches and its contexts' synchronous re-runs compute with was fixed at mi355ndt_stream_begin: not changed mid-stream)
  if (h->stream_on && (option == MI355NDT_OPT_F32_SUM_ORDER || option == MI355NDT_OPT_ARITH || option == MI355NDT_OPT_ASYNC_ALIGN)) NOT_IN_STREAM(h);
  if (option == MI355NDT_OPT_F32_SUM_ORDER) {
    if (value != 0 && value != 1) return MI355NDT_ERR_BAD_ARG;
    h->f32_sum_order = value;
    return MI355NDT_OK;
  }
  if (option == MI355NDT_OPT_ARITH) {
    if (value != 0 && value != 1) return MI355NDT_ERR_BAD_ARG;
    h->arith = value;                            // (grids built before lack the records of the other arithmetic: the next align rebuilds them)
    return MI355NDT_OK;
  }
  if (option == MI355NDT_OPT_ASYNC_ALIGN) {
    if (value < 0 || value > 2) return MI355NDT_ERR_BAD_ARG;
    h->async_align = value != 0;
    h->async_force = value == 2;                 // 2: also for batches smaller than the GPU's resident waves (testing)
    return MI355NDT_OK;
  }
  if (option == MI355NDT_OPT_DEBUG_ASYNC_ABORT) {
    h->debug_abort_pos = value < 0 ? 0xFFFFFFFFu : (unsigned)value;
    return MI355NDT_OK;
  }
  if (option == MI355NDT_OPT_DEBUG_ASYNC_RINGS) {
    if ((value & 0xFF) == 0) return MI355NDT_ERR_BAD_ARG;
    h->debug_ring_mask = (unsigned)value & 0xFFu;
    return MI355NDT_OK;
  }
  if (option == MI355NDT_OPT_STREAM_THRESHOLD) {
    if (value < -1 || value > ASYNC_MAX_CARRY) return MI355NDT_ERR_BAD_ARG;
    h->s_thresh_opt = value;
    return MI355NDT_OK;
  }
  if (option == MI355NDT_OPT_STREAM_RESERVE) {
    if (value < -1 || value > 4096) return MI355NDT_ERR_BAD_ARG;
    h->s_reserve_opt = value;
    return MI355NDT_OK;
  }
  return MI355NDT_ERR_BAD_ARG;
}
int mi355ndt_get_option(const mi355ndt_handle* h, int option, int* value) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!value) return MI355NDT_ERR_BAD_ARG;
  if (option == MI355NDT_OPT_F32_SUM_ORDER) { *value = h->f32_sum_order; return MI355NDT_OK; }
  if (option == MI355NDT_OPT_ARITH) { *value = h->arith; return MI355NDT_OK; }
  if (option == MI355NDT_OPT_ASYNC_ALIGN) { *value = h->async_force ? 2 : (h->async_align ? 1 : 0); return MI355NDT_OK; }
  if (option == MI355NDT_OPT_DEBUG_ASYNC_ABORT) { *value = h->debug_abort_pos == 0xFFFFFFFFu ? -1 : (int)h->debug_abort_pos; return MI355NDT_OK; }
  if (option == MI355NDT_OPT_DEBUG_ASYNC_RINGS) { *value = (int)h->debug_ring_mask; return MI355NDT_OK; }
  if (option == MI355NDT_OPT_STREAM_THRESHOLD) { *value = h->s_thresh_opt; return MI355NDT_OK; }
  if (option == MI355NDT_OPT_STREAM_RESERVE) { *value = h->s_reserve_opt; return MI355NDT_OK; }
  return MI355NDT_ERR_BAD_ARG;
}

// replaces PrefilteringNodelet::distance_filter + downsample (prefiltering_nodelet.cpp:137-181)
int mi355ndt_prefilter(mi355ndt_handle* h, const void* pts, size_t n, size_t stride,
                       int use_distance_filter, double distance_near, double distance_far, float downsample_resolution,
                       void* out_pts, size_t out_capacity, size_t out_stride, size_t* n_out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if ((!pts && n) || (n && stride < 12) || n >= (1u << 30) || !n_out || (out_pts && out_stride < 12)) return MI355NDT_ERR_BAD_ARG;
  if (std::isnan(downsample_resolution)) return MI355NDT_ERR_BAD_ARG;
  *n_out = 0;
  h->pf_count = 0;
  if (n == 0) return MI355NDT_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t pitch = (n + 63) & ~(size_t)63;
  if (pitch > h->pf_cap) {
    HIPCHK(h, hipStreamSynchronize(s));
    void** ps[] = {(void**)&h->d_pf_in, (void**)&h->d_pf_out, (void**)&h->d_pf_keep, (void**)&h->d_pf_keys, (void**)&h->d_pf_vals,
                   (void**)&h->d_pf_flag, (void**)&h->d_pf_pos};
    const size_t bytes[] = {3 * pitch * 4, 3 * pitch * 4, pitch, 2 * pitch * 4, 2 * pitch * 4, pitch * 4, pitch * 4};
    for (int i = 0; i < 7; i++) { if (*ps[i]) { HIPCHK(h, hipFree(*ps[i])); *ps[i] = nullptr; } HIPCHK(h, hipMalloc(ps[i], bytes[i])); }
    if (!h->d_pf_mm) HIPCHK(h, hipMalloc((void**)&h->d_pf_mm, 6 * sizeof(int)));
    if (!h->d_pf_grid) HIPCHK(h, hipMalloc((void**)&h->d_pf_grid, sizeof(PfGrid)));
    h->pf_cap = pitch;
  }
  h->pf_pitch = pitch;
  int rc = upload_cloud(h, h->d_pf_in, pitch, 0, pts, n, stride);
  if (rc) return rc;
  rc = uploads_before_compute(h);
  if (rc) return rc;
  const int gx = (int)((pitch + 255) / 256);
  unsigned *ka = h->d_pf_keys, *kb = h->d_pf_keys + pitch, *va = h->d_pf_vals, *vb = h->d_pf_vals + pitch;
  // workspace of the segment sort (one segment = the whole cloud) and of the emit-position scan
  const int pf_tiles = (int)((pitch + RS_TILE - 1) / RS_TILE);
  const int pf_chunks = (int)((pitch + PF_SCAN_CHUNK - 1) / PF_SCAN_CHUNK);
  {
    const size_t need = (size_t)pf_tiles << RS_MAX_BITS;
    if (need > h->rs_cap) {
      size_t c1 = 0, c2 = 0;
      if (h->d_rs_hist) { HIPCHK(h, hipFree(h->d_rs_hist)); h->d_rs_hist = nullptr; }
      if (h->d_rs_offs) { HIPCHK(h, hipFree(h->d_rs_offs)); h->d_rs_offs = nullptr; }
      HIPCHK(h, grow(h->d_rs_hist, c1, need)); HIPCHK(h, grow(h->d_rs_offs, c2, need));
      h->rs_cap = need;
    }
    const size_t nb = (size_t)pf_chunks * sizeof(unsigned);
    if (nb > h->pf_tmp_bytes) {
      if (h->d_pf_tmp) { HIPCHK(h, hipFree(h->d_pf_tmp)); h->d_pf_tmp = nullptr; }
      HIPCHK(h, hipMalloc(&h->d_pf_tmp, nb));
      h->pf_tmp_bytes = nb;
    }
  }
  k_minmax_init<<<1, 64, 0, s>>>(h->d_pf_mm, 1);
  k_pf_flag<<<std::min(gx, 256), 256, 0, s>>>(h->d_pf_in, pitch, (int)n, use_distance_filter, distance_near, distance_far, h->d_pf_keep, h->d_pf_mm);
  int downsample = downsample_resolution > 0.f;
  const unsigned* keys_sorted = ka;
  const unsigned* vals_sorted = va;
  if (downsample) {
    k_pf_grid<<<1, 1, 0, s>>>(h->d_pf_mm, downsample_resolution, h->d_pf_grid);
    PfGrid g;
    HIPCHK(h, hipMemcpyAsync(&g, h->d_pf_grid, sizeof g, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (g.status == 2) {                         // PCL: "Leaf size is too small for the input dataset" -> output = input
      h->err = "prefilter: leaf size too small for the cloud's extent, voxel indices would overflow; cloud not down-sampled";
      downsample = 0;
    } else {
      k_pf_keys<<<gx, 256, 0, s>>>(h->d_pf_in, pitch, (int)n, h->d_pf_keep, h->d_pf_grid, ka, va);
      // stable sort by voxel index: the target build's segment sort with the whole cloud as its one segment, 31 key bits
      const RsPlan plan = rs_plan(31);
      unsigned *kin = ka, *kout = kb, *vin = va, *vout = vb;
      for (int p = 0; p < plan.passes; p++) {
        rs_pass(s, plan.bits, kin, vin, kout, vout, pitch, p * plan.bits, h->d_rs_hist, h->d_rs_offs, pf_tiles, 1, false);
        std::swap(kin, kout); std::swap(vin, vout);
      }
      keys_sorted = kin; vals_sorted = vin;      // (an odd number of hops ends in kb / vb)
    }
  }
  k_pf_heads<<<gx, 256, 0, s>>>(keys_sorted, h->d_pf_keep, (int)n, pitch, downsample, h->d_pf_flag);
  k_pf_scan_totals<<<pf_chunks, 1024, 0, s>>>(h->d_pf_flag, pitch, (unsigned*)h->d_pf_tmp);
  k_pf_scan_offsets<<<1, 1024, 0, s>>>((unsigned*)h->d_pf_tmp, pf_chunks);
  k_pf_scan_apply<<<pf_chunks, 1024, 0, s>>>(h->d_pf_flag, pitch, (const unsigned*)h->d_pf_tmp, h->d_pf_pos);
  k_pf_emit<<<gx, 256, 0, s>>>(h->d_pf_in, pitch, keys_sorted, vals_sorted, h->d_pf_flag, h->d_pf_pos, downsample, h->d_pf_out, pitch);
  int last_pos = 0, last_flag = 0;
  HIPCHK(h, hipMemcpyAsync(&last_pos, h->d_pf_pos + (pitch - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(&last_flag, h->d_pf_flag + (pitch - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  const size_t m = (size_t)last_pos + (size_t)last_flag;
  h->pf_count = (int)m;
  *n_out = m;
  if (out_pts) {
    if (m > out_capacity) return MI355NDT_ERR_BAD_ARG;
    std::vector<float> tmp(3 * pitch);
    HIPCHK(h, hipMemcpy(tmp.data(), h->d_pf_out, 3 * pitch * sizeof(float), hipMemcpyDeviceToHost));
    unsigned char* o = (unsigned char*)out_pts;
    for (size_t i = 0; i < m; i++) {
      float v[3] = {tmp[i], tmp[pitch + i], tmp[2 * pitch + i]};
      memcpy(o + i * out_stride, v, 12);
    }
  }
  return MI355NDT_OK;
}

// hand the last prefilter result to the registration without leaving the GPU: role 1 = setInputSource, 2 = setInputTarget
int mi355ndt_use_prefiltered(mi355ndt_handle* h, int role) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (role != 1 && role != 2) return MI355NDT_ERR_BAD_ARG;
  if (!h->d_pf_out) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t m = (size_t)h->pf_count;
  int rc = ensure_single(h, role == 2, m);
  if (rc) return rc;
  rc = uploads_before_compute(h);                 // an earlier upload into the same rows must not land after these copies
  if (rc) return rc;
  float* dst = role == 2 ? h->d_tgt_own : h->d_src_own;
  const size_t dp = role == 2 ? h->tgt_pitch : h->src_pitch;
  HIPCHK(h, hipMemsetAsync(dst, 0, 3 * dp * sizeof(float), h->stream));
  for (int a = 0; a < 3; a++)
    if (m) HIPCHK(h, hipMemcpyAsync(dst + a * dp, h->d_pf_out + a * h->pf_pitch, m * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  if (role == 2) {
    h->h_tgt_cnt[0] = (int)m; h->have_target = true; h->targets_built = false;
    return mi355ndt_batch_build_targets(h);
  }
  h->h_src_cnt[0] = (int)m; h->have_source = true;
  return compute_enqueued(h);
}

// The nodelet's keyframe switch (scan_matching_odom_nodelet.cpp:240-243: `key = filtered; reg_s2k.setInputTarget(key);`) makes the cloud that was
// just aligned as SOURCE the next target: it is on the device already -- device-to-device into the target rows, then init() as setInputTarget does.
int mi355ndt_promote_source_to_target(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if (h->n_pairs != 1 || !h->have_source || h->d_src != h->d_src_own || !h->d_src_own) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t m = (size_t)h->h_src_cnt[0];
  int rc = ensure_single(h, true, m);
  if (rc) return rc;
  rc = uploads_before_compute(h);                 // the source's upload has to have landed; an earlier target upload must not land after these copies
  if (rc) return rc;
  const size_t dp = h->tgt_pitch, sp = h->src_pitch;
  HIPCHK(h, hipMemsetAsync(h->d_tgt_own, 0, 3 * dp * sizeof(float), h->stream));
  for (int a = 0; a < 3; a++)
    if (m) HIPCHK(h, hipMemcpyAsync(h->d_tgt_own + a * dp, h->d_src_own + a * sp, m * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  h->h_tgt_cnt[0] = (int)m; h->have_target = true; h->targets_built = false;
  h->P.cloud_promotions++;
  return mi355ndt_batch_build_targets(h);
}

// ---- stream mode ----------------------------------------------------------------------------------------------------------
// (include/mi355_ndt.h: mi355ndt_stream_*; kernels: ndt_async.hpp.  Replaces a run of batch_bind_device + batch_build_targets +
//  batch_align triples for batches that arrive one after the other -- scan_matching_odom_nodelet.cpp:144-183 is a stream of frames.)
// What a submit puts on the stream: ONE input copy, the build's kernels and two fills, the prepare kernel, the persistent launch, the
// status kernel.  No device-to-host copy, no event: result records and launch status land in mapped host memory.  (The first form of
// this path issued ~17 copies and fills per batch; at ~20 us of stream time each they cost more than the tail they removed.)
static bool stream_async_ok(const mi355ndt_handle* h) {
  const bool pca_kd = h->prm.neighbor_mode == MI355NDT_KDTREE && h->prm.variant == MI355NDT_VARIANT_PCA;
  return h->async_align && !mt_is_live(h->prm) && !pca_kd;
}
static int stream_free(mi355ndt_handle* h) {
  for (auto& c : h->sctx) {
    if (c.e) {
      if (c.d_in) { c.e->d_tgt_cnt = (int*)c.own_tgt_cnt; c.e->d_src_cnt = (int*)c.own_src_cnt; c.e->d_guess = (float*)c.own_guess; }
      c.e->d_bstat = nullptr;
      (void)mi355ndt_destroy(c.e);
    }
    if (c.d_in) (void)hipFree(c.d_in);
    if (c.h_in) (void)hipHostFree(c.h_in);
    if (c.h_res) (void)hipHostFree(c.h_res);
    c = mi355ndt_handle::StreamCtx();
  }
  for (auto& e : h->s_ev_built) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : h->s_ev_launched) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : h->s_ev_prepared) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (h->s_build_stream) { (void)hipStreamSynchronize(h->s_build_stream); (void)hipStreamDestroy(h->s_build_stream); h->s_build_stream = nullptr; }
  if (h->d_sctl) { (void)hipFree(h->d_sctl); h->d_sctl = nullptr; }
  if (h->d_sring) { (void)hipFree(h->d_sring); h->d_sring = nullptr; }
  if (h->d_sstat) { (void)hipFree(h->d_sstat); h->d_sstat = nullptr; }
  if (h->h_sstatus) { (void)hipHostFree((void*)h->h_sstatus); h->h_sstatus = nullptr; h->d_sstatus = nullptr; }
  h->stream_on = false; h->s_nctx = 0;
  return MI355NDT_OK;
}

int mi355ndt_stream_end(mi355ndt_handle* h) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!h->stream_on) return MI355NDT_OK;
  (void)hipSetDevice(h->device);
  if (h->s_build_stream) (void)hipStreamSynchronize(h->s_build_stream);
  (void)hipStreamSynchronize(h->stream);
  // the contexts' build timings and byte counts belong to this handle's profile
  for (int c = 0; c < h->s_nctx; c++) {
    mi355ndt_handle* e = h->sctx[c].e;
    if (!e) continue;
    ev_collect(e, e->ev_sweep, h->P.sweep_ms, h->P.sweep_launches);
    ev_collect(e, e->ev_update, h->P.update_ms, h->P.update_launches);
    ev_collect(e, e->ev_build, h->P.build_ms, h->P.build_launches);
    h->P.build_alg_bytes += e->P.build_alg_bytes; e->P.build_alg_bytes = 0;
  }
  return stream_free(h);
}

int mi355ndt_stream_begin(mi355ndt_handle* h, int n_contexts, int max_pairs, size_t max_tgt, size_t max_src) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (n_contexts < 2 || n_contexts > ASYNC_MAX_CTX || max_pairs < 1 || max_pairs > MAX_PAIRS || max_pairs >= (1 << ASYNC_CTX_SHIFT) ||
      max_tgt == 0 || max_src == 0 || max_tgt >= (1u << 31) || max_src >= (1u << 31)) return MI355NDT_ERR_BAD_ARG;
  if (h->stream_on) return MI355NDT_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->s_nctx = n_contexts; h->s_max_pairs = max_pairs; h->s_max_tgt = max_tgt; h->s_max_src = max_src;
  h->s_items = std::max(1, (int)((max_src + CHUNK_PTS - 1) / CHUNK_PTS)) * QUARTERS;
  h->s_sync_only = !stream_async_ok(h);
  h->s_next_id = 0; h->s_launches = 0; h->s_counted = 0; h->s_drop_carry = true; h->s_recovered_upto = -1;
  h->s_plan_cb = 0; h->s_plan_words = 0;
  // the grids a streamed launch reads are the contexts' (built at prm.resolution); whatever single-registration grid the parent still holds
  // -- possibly one a setResolution without a source left at another leaf size (ndt_omp.h:126-136) -- is no part of the stream
  h->targets_built = false; h->grid_resolution = 0.f; h->recs_fast_built = false;
  SweepConst sc;
  make_sweep_const(h, sc);
  {
    const int iu = h->s_items / (sc.K == 1 ? ((want_fast(h, sc) && FAST_D1_POINT) ? FAST_CLAIM1 : 2) : sc.K == 7 ? 2 : 1);   // positions per ticket (stream_launch: two DIRECT7 items per claim when pairs are handed over)
    const int waves = h->n_cu * sweep_wpe(sc.pca != 0, sc.K, want_fast(h, sc)) * WAVES;
    // automatic: four sweeps' worth of positions per resident wave -- `tools/gpu_job.sh thresh_sweep`: config 5 gains up to T = 32-64 (DIRECT7 19.1 / 19.4 / 19.5 k,
    // DIRECT1 39.4 / 40.1 / 40.9 / 41.1 k registrations/s at T = 8 / 16 / 32 / 64), the 65,536-point configurations do not care -- capped at a quarter of the batch (stream_launch)
    int t = h->s_thresh_opt >= 0 ? h->s_thresh_opt : 4 * ((waves + iu - 1) / std::max(1, iu));
    if (const char* e = std::getenv("MI355NDT_STREAM_THRESH")) t = std::atoi(e);
    h->s_thresh = std::max(0, std::min(t, ASYNC_MAX_CARRY));
  }
  {
    // MI355NDT_STREAM_RESERVE (workgroups, rounded to a multiple of 8; 0 = the build runs between the launches, on the same stream).  Defaults
    // (tools/reserve_sweep_*.sh, reserve_matrix.sh, reserve_resweep*.sh; a launch's time grows with the slots it gives away, 512 / (512 - r), in every search:
    // what is won is the build's time):
    //  * DIRECT1: 128 for clouds of up to 98,304 points, 96 beyond.  Its launches wait for their point stream more than they compute (VALU busy 0.4-0.6) and
    //    are short enough for the build to be 30 % of a step: nodelet configuration (1 m, 65,536 points) r = 0 / 64 / 96 / 112 / 128 / 160: 112.1 / 117.3 /
    //    122.1 / 120.1 / 124.7 / 116.8 k registrations/s; config 5's clouds (0.5 m, 131,072 points) 42.3 / - / 44.5 / - / 43.0 k; 64-pair batches 65.4 -> 96.6 k.
    //  * ndt_omp / DIRECT7 (the headline's configuration): 64 = eight slots per XCD.  The launch is VALU-bound, so the slots are paid for in full
    //    (3.95 -> 4.41 ms) -- but the whole 0.70 ms build disappears under it: 57.3 -> 59.7 k and 55.7 -> 59.2 k on two boxes (r = 0 / 16 / 32 / 48 /
    //    64 / 80 / 96 / 128: 55.7 / 56.3 / 57.4 / 56.9 / 59.2 / 57.9 / 56.1 / 53.4 k); the tolerance arithmetic +2 % (90.3 -> 92.3 k).
    //    The smaller the batch, the more it is worth (a small build is a chain of short kernels, not throughput): 64 pairs 41.3 -> 50.6 k; at 1,536 pairs
    //    per batch the build no longer fits under its launch: exact +-0, tolerance arithmetic -6 % (DIRECT1 still +4 %) -- so only for batches up to
    //    768 x 65,536 target points.
    //  * ndt_pca / DIRECT7: 32 (same bound on the batch).  Until the build was made to start BEHIND the launch's prepare kernel (stream_submit) its first kernels
    //    raced the launch's own start and r >= 64 cost a third of the rate; since then config 5 (0.5 m, 128 x 131,072) r = 0 / 16 / 32 / 48 / 64 / 96: 19.1 /
    //    19.3 / 20.0 / 19.6 / 19.8 / 18.8 k, 271 x 65,536 at 1 m 34.9 -> 35.3 k, 64-pair batches 31.7 -> 35.3 k.
    //  * Everything else (DIRECT26, KDTREE): 0.
    const bool small_batch = (unsigned long long)max_pairs * (unsigned long long)max_tgt <= 768ull * 65536ull;
    int r = sc.K == 1 ? (max_tgt <= 98304 ? 128 : 96) : ((sc.K == 7 && small_batch) ? (sc.pca ? 32 : 64) : 0);
    if (const char* e = std::getenv("MI355NDT_STREAM_RESERVE")) r = std::atoi(e);
    if (h->s_reserve_opt >= 0) r = h->s_reserve_opt;
    r = std::max(0, std::min(r, h->n_cu * sweep_wpe(sc.pca != 0, sc.K, want_fast(h, sc)) / 2)) & ~7;
    if (n_contexts < 3) r = 0;                       // (the overlapped build needs its context free one launch earlier: at least three contexts)
    h->s_reserve_wg = r;
    h->s_launch_slots = std::max(8, h->n_cu * sweep_wpe(sc.pca != 0, sc.K, want_fast(h, sc)) - r);
  }
  h->s_ring_cap = async_ring_cap(h, (long long)max_pairs + ASYNC_MAX_CARRY);
  if (h->s_ring_cap == 0) h->s_sync_only = true;
  auto fail = [&](int rc) { (void)stream_free(h); return rc; };
  if (hipMalloc((void**)&h->d_sstat, ASYNC_MAX_CTX * sizeof(CtxStat)) != hipSuccess ||
      hipMemsetAsync(h->d_sstat, 0, ASYNC_MAX_CTX * sizeof(CtxStat), h->stream) != hipSuccess) return fail(MI355NDT_ERR_HIP);
  for (int c = 0; c < n_contexts; c++) {
    mi355ndt_handle::StreamCtx& S = h->sctx[c];
    int rc = mi355ndt_create(&h->prm, h->device, &S.e);
    if (rc) return fail(rc);
    mi355ndt_handle* e = S.e;
    if (h->s_reserve_wg > 0 && !h->s_build_stream) {
      if (hipStreamCreateWithFlags(&h->s_build_stream, hipStreamNonBlocking) != hipSuccess) return fail(MI355NDT_ERR_HIP);
      for (auto& ev : h->s_ev_built) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(MI355NDT_ERR_HIP);
      for (auto& ev : h->s_ev_launched) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(MI355NDT_ERR_HIP);
      for (auto& ev : h->s_ev_prepared) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(MI355NDT_ERR_HIP);
      if (const char* pf = std::getenv("MI355NDT_STREAM_PREP_FIRST")) h->s_prep_first = std::atoi(pf) != 0;
    }
    rc = mi355ndt_set_stream(e, h->s_reserve_wg > 0 ? h->s_build_stream : h->stream);
    if (rc) return fail(rc);
    e->f32_sum_order = h->f32_sum_order; e->arith = h->arith; e->async_align = h->async_align; e->dyn_shift = h->dyn_shift;
    e->async_build = true;
    e->ev_pool_target = 128;
    rc = ensure_pair_arrays(e, max_pairs);          // every per-pair array at its final size: no allocation, no wait inside submit
    if (rc) { h->err = e->err; return fail(rc); }
    // the input block: [target counts | source counts | guesses]
    S.in_bytes = (size_t)max_pairs * (2 * sizeof(int) + 16 * sizeof(float));
    if (hipMalloc((void**)&S.d_in, S.in_bytes) != hipSuccess || hipHostMalloc((void**)&S.h_in, S.in_bytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&S.h_in_dev, S.h_in, 0) != hipSuccess ||
        hipHostMalloc((void**)&S.h_res, (size_t)max_pairs * sizeof(mi355ndt_result), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&S.d_res_map, S.h_res, 0) != hipSuccess) { h->err = "stream_begin: allocation failed"; return fail(MI355NDT_ERR_HIP); }
    S.own_tgt_cnt = e->d_tgt_cnt; S.own_src_cnt = e->d_src_cnt; S.own_guess = e->d_guess;
    e->d_tgt_cnt = S.d_in; e->d_src_cnt = S.d_in + max_pairs; e->d_guess = (float*)(S.d_in + 2 * (size_t)max_pairs);
    e->d_bstat = reinterpret_cast<unsigned*>(h->d_sstat + c);
    size_t need = (size_t)max_pairs * h->s_items * NACC;
    if (grow(e->d_partials, e->partials_cap, need) != hipSuccess || grow(e->d_arrived, e->arrived_cap, (size_t)max_pairs * ASYNC_ARR_STRIDE) != hipSuccess)
      { h->err = "stream_begin: allocation failed"; return fail(MI355NDT_ERR_HIP); }
    if (h->prof) (void)mi355ndt_profile_enable(e, 1);
  }
  if (hipMalloc((void**)&h->d_sctl, 2 * sizeof(AsyncCtl)) != hipSuccess || hipMemsetAsync(h->d_sctl, 0, 2 * sizeof(AsyncCtl), h->stream) != hipSuccess ||
      hipHostMalloc((void**)&h->h_sstatus, mi355ndt_handle::S_EV * sizeof(StreamStatus), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->d_sstatus, (void*)h->h_sstatus, 0) != hipSuccess) return fail(MI355NDT_ERR_HIP);
  memset((void*)h->h_sstatus, 0, mi355ndt_handle::S_EV * sizeof(StreamStatus));
  if (!h->d_atab && hipMalloc((void**)&h->d_atab, sizeof(AsyncTab)) != hipSuccess) return fail(MI355NDT_ERR_HIP);
  if (!h->s_sync_only && hipMalloc((void**)&h->d_sring, (size_t)8 * h->s_ring_cap * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); h->s_sync_only = true; }
  if (h->s_build_stream) HIPCHK(h, hipStreamSynchronize(h->s_build_stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  (void)max_tgt;
  h->stream_on = true;
  return MI355NDT_OK;
}

// one persistent launch of the stream: the pairs the previous launch suspended + the `n_new` pairs of context `new_ci` (-1: a flush --
// nothing new, everything runs to its end), then the status kernel (launch outcome + every context's counters -> mapped host memory)
static int stream_launch(mi355ndt_handle* h, int new_ci, int n_new) {
  hipStream_t s = h->stream;
  SweepConst sc;
  make_sweep_const(h, sc);
  AsyncLaunch L;
  memset(&L.tab, 0, sizeof L.tab);
  const bool flush = new_ci < 0;
  for (int c = 0; c < h->s_nctx; c++) {
    mi355ndt_handle* e = h->sctx[c].e;
    if (!e->d_src) continue;                         // never bound: no ticket can name it
    fill_async_ctx(e, L.tab.c[c]);
    L.tab.c[c].results = h->sctx[c].d_res_map;
    L.tab.c[c].n_done = &h->d_sstat[c].done;
    L.tab.c[c].pose = h->sctx[c].busy ? h->sctx[c].d_pose : nullptr; L.tab.c[c].pose_base = h->sctx[c].pose_base; L.tab.c[c].pose_stride = h->sctx[c].pose_stride;
    // The context the NEXT submit recycles must be finished by this launch; the others may hand their last pairs over.  With three or more
    // contexts the context after that one must finish too: its batch is then complete one launch BEFORE the submit that recycles it, so the
    // host collects it and enqueues the next build while a launch is still running -- otherwise every collect returns at the very end of a
    // launch and the GPU idles for as long as the host takes to notice, collect and enqueue (~0.1-0.2 ms per batch, measured as the
    // difference between a streamed step and its kernels).  (The build under the launch -- s_reserve_wg -- needs the same.)
    const int ahead = (h->s_reserve_wg > 0 || h->s_nctx >= 3) ? 2 : 1;
    bool mf = flush;
    for (int a = 1; a <= ahead; a++) mf = mf || c == (new_ci + a) % h->s_nctx;
    L.tab.c[c].must_finish = mf ? 1 : 0;
  }
  const long long j = h->s_launches;
  L.new_ci = flush ? 0 : new_ci; L.n_new = flush ? 0 : n_new;
  if (!flush) {
    mi355ndt_handle* e = h->sctx[new_ci].e;
    L.st_new = e->d_state; L.guess_new = e->d_guess; L.src_cnt_new = e->d_src_cnt; L.gd_new = e->d_grid; L.arrived_new = e->d_arrived;
    L.active_list = e->d_active_list; L.sweep_ctl = nullptr; L.done_new = &h->d_sstat[new_ci].done;
    L.pose_new = h->sctx[new_ci].d_pose; L.pose_cap = h->sctx[new_ci].pose_cap;
  }
  L.tab_dev = h->d_atab; L.ring = h->d_sring; L.ring_cap = h->s_ring_cap;
  L.ctl = h->d_sctl + (j & 1); L.prev = h->s_drop_carry ? nullptr : h->d_sctl + ((j + 1) & 1);
  // (the automatic threshold never hands over more than a quarter of the batch: a batch too small to fill the GPU has no bulk to hide stragglers under)
  const bool thresh_given = h->s_thresh_opt >= 0 || std::getenv("MI355NDT_STREAM_THRESH");
  L.items_per_pair = h->s_items; L.stop_thresh = flush ? 0 : (thresh_given ? h->s_thresh : std::min(h->s_thresh, n_new / 4)); L.debug_abort_pos = h->debug_abort_pos; L.debug_ring_mask = h->debug_ring_mask;
  L.reserve_wg = flush ? 0 : h->s_reserve_wg;
  L.ev_prepared = (h->s_reserve_wg > 0 && h->s_prep_first) ? h->s_ev_prepared[j % mi355ndt_handle::S_EV] : nullptr;
  // two DIRECT7 items per claim halve the hand-overs between items (+1.4-2 %); the coarser positions lengthen a launch's own tail, so only
  // where the tail is handed on (docs/experiments.md 10d)
  L.claim_items = (sc.K == 7 && L.stop_thresh > 0) ? 2 : 1;
  h->ev_last_fresh = false;                          // (the contexts' builds sit between two launches on this stream)
  if (!flush && h->s_reserve_wg > 0) HIPCHK(h, hipStreamWaitEvent(s, h->s_ev_built[new_ci], 0));   // this batch's grids (built on the other stream)
  L.stamp_end = (!flush && h->prof) ? &h->d_sstatus[j % mi355ndt_handle::S_EV].build_t1 : nullptr;
  int rc = launch_async(h, sc, L);
  if (rc) return rc;
  h->s_drop_carry = false;
  const int slot = (int)(j % mi355ndt_handle::S_EV);
  k_stream_status<<<1, 64, 0, s>>>(L.ctl, h->d_sstat, reinterpret_cast<volatile unsigned*>(h->d_sstatus + slot), (unsigned)(j + 1));
  HIPCHK(h, hipGetLastError());
  if (h->s_reserve_wg > 0) HIPCHK(h, hipEventRecord(h->s_ev_launched[slot], s));
  h->s_launches++;
  h->P.stream_launches++;
  return MI355NDT_OK;
}

// wait until launch j has reported (its status slot carries sequence number j + 1): the host polls mapped memory
static int stream_wait_launch(mi355ndt_handle* h, long long j) {
  volatile StreamStatus* st = h->h_sstatus + (j % mi355ndt_handle::S_EV);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; st->seq != (unsigned)(j + 1); spins++) {
    if ((spins & 1023) == 1023) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { h->err = "stream: the device stopped reporting"; return MI355NDT_ERR_STATE; }
      std::this_thread::yield();
    } else cpu_relax();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (; h->s_counted <= j; h->s_counted++) {       // (launches finish in order)
    volatile StreamStatus* c = h->h_sstatus + (h->s_counted % mi355ndt_handle::S_EV);
    h->P.stream_carried += c->susp;
    const unsigned long long b0 = c->build_t0, b1 = c->build_t1;
    if (b0 && b1 > b0) { h->P.build_ms += (double)(b1 - b0) * 1e-5; h->P.build_launches++; }   // wall_clock64: 100 MHz
    c->build_t0 = 0; c->build_t1 = 0;
  }
  return MI355NDT_OK;
}

int mi355ndt_stream_submit(mi355ndt_handle* h, int n_pairs, const float* d_t, const int* tc, size_t tp, const float* d_s, const int* scnt, size_t sp,
                           const float* guesses, long long* batch_id) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!h->stream_on) return MI355NDT_ERR_STATE;
  if (n_pairs < 1 || n_pairs > h->s_max_pairs || !guesses || !batch_id || !tc || !scnt) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const long long id = h->s_next_id;
  const int ci = (int)(id % h->s_nctx);
  mi355ndt_handle::StreamCtx& S = h->sctx[ci];
  if (S.busy) { h->err = "stream_submit: collect batch " + std::to_string(S.batch_id) + " first (its context is the one this batch needs)"; return MI355NDT_ERR_STATE; }
  mi355ndt_handle* e = S.e;
  for (int b = 0; b < n_pairs; b++) if ((size_t)scnt[b] > (size_t)(h->s_items / QUARTERS) * CHUNK_PTS) return MI355NDT_ERR_BAD_ARG;   // more source points than stream_begin was told
  int rc = mi355ndt_batch_bind_device(e, n_pairs, d_t, tc, tp, d_s, scnt, sp);
  if (rc) { h->err = e->err; return rc; }
  e->prm = h->prm;
  S.batch_id = id; S.n_pairs = n_pairs; S.redo = false; S.done_sync = false; S.launch = -1;
  S.guesses.assign(guesses, guesses + (size_t)n_pairs * 16);
  S.d_pose = (PoseRecord*)h->s_pose_next; S.pose_cap = (int)h->s_pose_cap_next; S.pose_base = h->s_pose_base_next; S.pose_stride = h->s_pose_stride_next;
  h->s_pose_next = nullptr; h->s_pose_cap_next = 0;
  if (S.d_pose && (size_t)n_pairs > (size_t)S.pose_cap) return MI355NDT_ERR_BAD_ARG;
  auto run_sync = [&]() -> int {                     // build + align this batch here and now, results into the context's host buffer
    e->async_build = false; e->counts_preloaded = false;
    e->up_tgt_cnt.clear(); e->up_src_cnt.clear();
    int r = mi355ndt_batch_build_targets(e);
    if (r == MI355NDT_OK) r = mi355ndt_batch_align(e, S.guesses.data(), S.h_res);
    e->async_build = true;
    if (r) h->err = e->err;
    return r;
  };
  if (h->s_sync_only) {                              // a configuration the one-launch align does not serve: processed here and now
    rc = run_sync();
    if (rc) return rc;
    S.done_sync = true; S.busy = true;
    *batch_id = id; h->s_next_id++;
    return MI355NDT_OK;
  }
  // the batch's small inputs in one copy: point counts of both sides, guesses
  memcpy(S.h_in, tc, (size_t)n_pairs * sizeof(int));
  memcpy(S.h_in + h->s_max_pairs, scnt, (size_t)n_pairs * sizeof(int));
  memcpy(S.h_in + 2 * (size_t)h->s_max_pairs, guesses, (size_t)n_pairs * 16 * sizeof(float));
  // this context's previous batch was finished by the launch before the last one (must_finish): the build may start when that launch has ended -- and
  // a moment later still, when the LAST launch's prepare kernel is through (it follows that end on the stream): the build's first kernels stream the
  // whole batch through HBM and would otherwise run against the one short kernel every launch waits for (k_async_prepare: 50 us beside k_minmax, 17 alone)
  if (h->s_reserve_wg > 0 && h->s_prep_first && h->s_launches >= 1)
    HIPCHK(h, hipStreamWaitEvent(e->stream, h->s_ev_prepared[(h->s_launches - 1) % mi355ndt_handle::S_EV], 0));
  else if (h->s_reserve_wg > 0 && h->s_launches >= 2)
    HIPCHK(h, hipStreamWaitEvent(e->stream, h->s_ev_launched[(h->s_launches - 2) % mi355ndt_handle::S_EV], 0));
  // (one workgroup reads the block from mapped host memory and clears the build's word block: no copy, no fill -- k_stream_inputs)
  k_stream_inputs<<<1, 1024, 0, e->stream>>>(S.h_in_dev, reinterpret_cast<unsigned*>(S.d_in), (unsigned)(2 * (size_t)h->s_max_pairs + (size_t)n_pairs * 16),
                                            e->d_word_off, (unsigned)(2 + 6 * (size_t)e->cap_pairs),
                                            h->prof ? &h->d_sstatus[h->s_launches % mi355ndt_handle::S_EV].build_t0 : nullptr);
  e->word_off_cleared = true;
  e->build_stamped = h->prof;
  e->counts_preloaded = true; e->up_src_cnt.clear(); e->up_tgt_cnt.clear();
  // target build: against the stream's plan when there is one (no wait), else synchronously -- which makes the plan
  e->async_build = true;
  e->plan_cb = h->s_plan_cb; e->plan_words = h->s_plan_words;
  if (e->plan_words > e->words_cap) {
    size_t c = e->words_cap;
    HIPCHK(h, grow(e->d_words, c, e->plan_words));
    e->words_cap = c;
  }
  rc = mi355ndt_batch_build_targets(e);
  if (rc) { h->err = e->err; return rc; }
  if (!(e->plan_cb > 0 && e->plan_words > 0)) {      // that build waited for its sizes: learn from it (with headroom: scans of one drive vary by a few per cent)
    h->s_plan_cb = std::max(h->s_plan_cb, e->last_cb);
    h->s_plan_words = std::max(h->s_plan_words, e->last_total_words + e->last_total_words / 4 + 1024);
  }
  if (h->s_reserve_wg > 0) HIPCHK(h, hipEventRecord(h->s_ev_built[ci], e->stream));
  // align workspace of this context: fixed row geometry for the whole stream (a pair's rows do not depend on it)
  e->chunks_per_pair = h->s_items / QUARTERS; e->rows_per_pair = e->items_per_pair = h->s_items; e->pts_per_chunk = CHUNK_PTS; e->fine_it = 0;
  gauss_constants3(h->prm.outlier_ratio, h->prm.resolution, h->gauss_last);
  S.busy = true;
  S.launch = h->s_launches;
  rc = stream_launch(h, ci, n_pairs);
  if (rc) {                                          // the launch cannot be made (not resident): this and every later batch synchronously
    h->s_sync_only = true;
    if (h->s_build_stream) HIPCHK(h, hipStreamSynchronize(h->s_build_stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    rc = run_sync();
    if (rc) { S.busy = false; return rc; }
    S.done_sync = true;
  }
  e->aligned_once = true;
  *batch_id = id; h->s_next_id++;
  return MI355NDT_OK;
}

// The stream for the caller the reference actually has: HOST clouds (scan_matching_odom_nodelet.cpp:144-183 receives pcl::PointCloud records, one
// callback at a time).  The batch's clouds are staged by the engine's own threads into the pinned slots of the context this batch lives in,
// cross PCIe on that context's copy streams and land in ITS device buffers -- while the launches of the batches submitted before keep the GPU
// busy -- and then the batch goes the way of mi355ndt_stream_submit.  Returns when the caller's memory is no longer needed.
int mi355ndt_stream_submit_host(mi355ndt_handle* h, int n_pairs, const void* const* targets, const size_t* target_counts, const void* const* sources,
                                const size_t* source_counts, size_t stride, const float* guesses, int n_threads, long long* batch_id) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!h->stream_on) return MI355NDT_ERR_STATE;
  if (n_pairs < 1 || n_pairs > h->s_max_pairs || !targets || !target_counts || !sources || !source_counts || stride < 12 || !guesses || !batch_id) return MI355NDT_ERR_BAD_ARG;
  for (int b = 0; b < n_pairs; b++)
    if (target_counts[b] > h->s_max_tgt || source_counts[b] > h->s_max_src || (!targets[b] && target_counts[b]) || (!sources[b] && source_counts[b])) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const int ci = (int)(h->s_next_id % h->s_nctx);
  mi355ndt_handle::StreamCtx& S = h->sctx[ci];
  if (S.busy) { h->err = "stream_submit_host: collect batch " + std::to_string(S.batch_id) + " first (its context is the one this batch needs)"; return MI355NDT_ERR_STATE; }
  mi355ndt_handle* e = S.e;
  const size_t tp = (h->s_max_tgt + 63) & ~(size_t)63, sp = (h->s_max_src + 63) & ~(size_t)63;
  if (!e->d_tgt_own || !e->d_src_own || e->own_tgt_pairs < h->s_max_pairs || e->own_src_pairs < h->s_max_pairs || e->own_tgt_pitch != tp || e->own_src_pitch != sp) {
    int rc = mi355ndt_batch_reserve(e, h->s_max_pairs, h->s_max_tgt, h->s_max_src);      // (once per context: the stream's sizes never change)
    if (rc) { h->err = e->err; return rc; }
  }
  // (the context's previous batch has been collected -- S.busy is false --, so no kernel still reads these rows)
  e->n_pairs = h->s_max_pairs; e->d_tgt = e->d_tgt_own; e->d_src = e->d_src_own; e->tgt_pitch = tp; e->src_pitch = sp;
  int rc = mi355ndt_batch_set_clouds(e, 0, n_pairs, targets, target_counts, sources, source_counts, stride, n_threads);
  if (rc) { h->err = e->err; return rc; }
  std::vector<int> tc((size_t)n_pairs), sc((size_t)n_pairs);
  for (int b = 0; b < n_pairs; b++) { tc[(size_t)b] = (int)target_counts[b]; sc[(size_t)b] = (int)source_counts[b]; }
  // (the build that mi355ndt_stream_submit enqueues first waits for these uploads: uploads_before_compute of the context's engine)
  return mi355ndt_stream_submit(h, n_pairs, e->d_tgt_own, tc.data(), tp, e->d_src_own, sc.data(), sp, guesses, batch_id);
}

int mi355ndt_stream_pose_records(mi355ndt_handle* h, void* d_records, size_t capacity, int id_base, int id_stride) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!h->stream_on) return MI355NDT_ERR_STATE;
  if ((d_records && capacity == 0) || capacity > (size_t)MAX_PAIRS) return MI355NDT_ERR_BAD_ARG;
  h->s_pose_next = d_records; h->s_pose_cap_next = d_records ? capacity : 0; h->s_pose_base_next = id_base; h->s_pose_stride_next = id_stride;
  return MI355NDT_OK;
}

// a launch gave up (ctl->abort_): nothing it left behind can be trusted to continue from -- every unfinished batch is re-run synchronously
// by its collect, and the next launch starts without a hand-over list
static void stream_recover(mi355ndt_handle* h) {
  if (h->s_build_stream) (void)hipStreamSynchronize(h->s_build_stream);
  (void)hipStreamSynchronize(h->stream);
  CtxStat st[ASYNC_MAX_CTX];
  if (hipMemcpy(st, h->d_sstat, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) memset(st, 0, sizeof st);
  for (int c = 0; c < h->s_nctx; c++) {
    mi355ndt_handle::StreamCtx& S = h->sctx[c];
    if (S.busy && !S.done_sync && st[c].done != (unsigned)S.n_pairs) S.redo = true;
  }
  h->s_drop_carry = true;
  h->s_recovered_upto = h->s_launches - 1;       // everything enqueued so far has drained and been marked: a later collect that reads this launch's abort flag again has nothing to do
  h->P.async_fallbacks++;
}

int mi355ndt_stream_collect(mi355ndt_handle* h, long long batch_id, mi355ndt_result* out) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  if (!h->stream_on) return MI355NDT_ERR_STATE;
  if (batch_id < 0 || batch_id >= h->s_next_id || !out) return MI355NDT_ERR_BAD_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const int ci = (int)(batch_id % h->s_nctx);
  mi355ndt_handle::StreamCtx& S = h->sctx[ci];
  if (!S.busy || S.batch_id != batch_id) return MI355NDT_ERR_BAD_ARG;       // collected already (or its context has been recycled)
  mi355ndt_handle* e = S.e;
  bool reran = S.done_sync;                          // went through the synchronous path (then the pose records come from the engine's packer)
  if (!S.done_sync) {
    long long j = S.launch;
    bool plan_exceeded = false;
    for (;;) {
      int rc = stream_wait_launch(h, j);
      if (rc) return rc;
      const StreamStatus st = *const_cast<const StreamStatus*>(h->h_sstatus + (j % mi355ndt_handle::S_EV));
      if (st.abort_ && !S.redo && j > h->s_recovered_upto) stream_recover(h);
      plan_exceeded = st.ctx[ci].plan_exceeded != 0;
      if (S.redo || plan_exceeded) break;
      if (st.ctx[ci].done == (unsigned)S.n_pairs) break;
      if (j + 1 < h->s_launches) { j++; continue; }  // its stragglers ride in a later launch that is already queued
      rc = stream_launch(h, -1, 0);                  // nothing newer: flush them
      if (rc) { stream_recover(h); S.redo = true; break; }
      j = h->s_launches - 1;
    }
    if (S.redo || plan_exceeded) {
      // the batch did not fit the build plan (its grids were withheld), or its launch gave up: the synchronous path, which also re-makes the plan
      if (h->s_build_stream) HIPCHK(h, hipStreamSynchronize(h->s_build_stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      e->async_build = false; e->counts_preloaded = false; e->up_tgt_cnt.clear(); e->up_src_cnt.clear();
      int rc = mi355ndt_batch_build_targets(e);
      if (rc == MI355NDT_OK) rc = mi355ndt_batch_align(e, S.guesses.data(), S.h_res);
      e->async_build = true;
      if (rc) { h->err = e->err; S.busy = false; return rc; }
      h->s_plan_cb = std::max(h->s_plan_cb, e->last_cb);
      h->s_plan_words = std::max(h->s_plan_words, e->last_total_words + e->last_total_words / 4 + 1024);
      // the synchronous align re-computed its own row geometry: back to the stream's for this context's next batch
      e->chunks_per_pair = h->s_items / QUARTERS; e->rows_per_pair = e->items_per_pair = h->s_items; e->pts_per_chunk = CHUNK_PTS; e->fine_it = 0;
      if (grow(e->d_partials, e->partials_cap, (size_t)h->s_max_pairs * h->s_items * NACC) != hipSuccess) return MI355NDT_ERR_HIP;
      h->P.stream_redone++;
      reran = true;
    }
  }
  if (S.d_pose && reran) {       // a batch that went through the synchronous path: its records from the engine's packer
    int rc = mi355ndt_batch_pose_records(e, S.pose_base, S.pose_stride, S.d_pose, (size_t)S.pose_cap);
    if (rc) { h->err = e->err; return rc; }
  }
  memcpy(out, S.h_res, (size_t)S.n_pairs * sizeof(mi355ndt_result));
  tolerance_warnings(h, out, S.n_pairs);         // (a synchronously re-run batch has them already: idempotent)
  if (h->prof) {
    const int K = h->prm.neighbor_mode == MI355NDT_DIRECT1 ? 1 : h->prm.neighbor_mode == MI355NDT_DIRECT7 ? 7 : h->prm.neighbor_mode == MI355NDT_DIRECT26 ? 26 : 27;
    for (int b = 0; b < S.n_pairs; b++) {
      h->P.sweep_alg_bytes += (double)out[b].sweeps * e->h_src_cnt[b] * (12.0 + 4.0 * K);
      h->P.sweep_points += (long long)out[b].sweeps * e->h_src_cnt[b];
    }
  }
  S.busy = false;
  return MI355NDT_OK;
}

int mi355ndt_pack_pose_records(const mi355ndt_result* results, int n, int id_base, int id_stride, void* records, size_t capacity) {
  if (!results || !records || n < 0 || (size_t)n > capacity) return MI355NDT_ERR_BAD_ARG;
  PoseRecord* out = (PoseRecord*)records;
  for (size_t k = 0; k < capacity; k++) {
    PoseRecord r;
    memset(&r, 0, sizeof r);
    r.pair_id = -1;
    if (k < (size_t)n) {
      for (int a = 0; a < 16; a++) r.final_cm[a] = results[k].final_colmajor[a];
      r.score = (float)results[k].score;
      r.iterations = results[k].iterations;
      r.converged = results[k].converged;
      r.pair_id = id_base + (int)k * id_stride;
    }
    out[k] = r;
  }
  return MI355NDT_OK;
}

// ---- latency mode ---------------------------------------------------------------------------------------------------------
int mi355ndt_set_latency_mode(mi355ndt_handle* h, int on) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  h->latency_mode = on != 0;
  return MI355NDT_OK;
}

int mi355ndt_sequence_run(mi355ndt_handle* h, int n_frames, const void* const* clouds, const size_t* counts, size_t stride,
                          const double* stamps, const mi355ndt_seq_params* policy,
                          mi355ndt_seq_frame* out_frames, mi355ndt_result* out_results, mi355ndt_seq_stats* stats) {
  if (!h) return MI355NDT_ERR_BAD_HANDLE;
  NOT_IN_STREAM(h);
  if (n_frames < 1 || n_frames > MAX_PAIRS || !clouds || !counts || !stamps || !out_frames || stride < 12) return MI355NDT_ERR_BAD_ARG;
  if (mt_is_live(h->prm) || (h->prm.neighbor_mode != MI355NDT_DIRECT1 && h->prm.neighbor_mode != MI355NDT_DIRECT7)) {
    h->err = "sequence mode serves DIRECT1 / DIRECT7 with step_size > transformation_epsilon / 2 (every configuration lv_slam ships)";
    return MI355NDT_ERR_UNSUPPORTED;
  }
  size_t maxn = 0;
  for (int k = 0; k < n_frames; k++) { if (counts[k] == 0 || counts[k] >= (1u << 31) || !clouds[k]) return MI355NDT_ERR_BAD_ARG; maxn = std::max(maxn, counts[k]); }
  HIPCHK(h, hipSetDevice(h->device));
  const auto t_up0 = std::chrono::steady_clock::now();
  // every frame is a TARGET slot (its voxel grid is built: it may become a keyframe) and, through the same rows, the SOURCE of its own
  // align: one cloud buffer serves both sides
  int rc = mi355ndt_batch_reserve(h, n_frames, maxn, 64);
  if (rc) return rc;
  rc = mi355ndt_batch_set_clouds(h, 0, n_frames, clouds, counts, nullptr, nullptr, stride, 0);
  if (rc) return rc;
  h->d_src = h->d_tgt_own; h->src_pitch = h->tgt_pitch;
  for (int k = 0; k < n_frames; k++) h->h_src_cnt[k] = h->h_tgt_cnt[k];
  h->have_source = true;
  struct Unalias { mi355ndt_handle* h; ~Unalias() { h->d_src = h->d_src_own; h->src_pitch = h->own_src_pitch; std::fill(h->h_src_cnt.begin(), h->h_src_cnt.end(), 0);
                                                     h->have_source = false; h->d_grid_of_use = nullptr; h->aligned_once = false;
                                                     if (h->ev_compute) (void)compute_enqueued(h); } } unalias{h};   // (error exits too: later uploads wait for what was enqueued)
  for (hipStream_t cs : h->copy_stream) HIPCHK(h, hipStreamSynchronize(cs));     // (upload time is reported on its own)
  const auto t_up1 = std::chrono::steady_clock::now();
  const bool keep_prof = h->prof;
  struct ProfBack { mi355ndt_handle* h; bool v; ~ProfBack() { h->prof = v; } } profback{h, keep_prof};   // (every exit restores it)
  h->prof = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  struct EvFree { hipEvent_t* e; ~EvFree() { for (int i = 0; i < 3; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } evfree{ev};
  for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
  hipStream_t s = h->stream;
  HIPCHK(h, hipEventRecord(ev[0], s));
  rc = mi355ndt_batch_build_targets(h);
  if (rc) { h->prof = keep_prof; return rc; }
  HIPCHK(h, hipEventRecord(ev[1], s));
  const bool lat = h->latency_mode;
  h->latency_mode = true; h->seq_running = true;
  rc = prep_align_ws(h);
  h->latency_mode = lat; h->seq_running = false;
  if (rc == MI355NDT_OK && !h->fine_it) { h->err = "sequence run: the fine-grained sweep does not serve this configuration"; rc = MI355NDT_ERR_UNSUPPORTED; }
  if (rc) { h->prof = keep_prof; return rc; }
  // one pair is in flight at a time: the fine grid is sized for one pair
  if ((size_t)n_frames > h->seq_cap) {
    for (void** p : {(void**)&h->d_grid_of, (void**)&h->d_seq_out, (void**)&h->d_stamps}) if (*p) { HIPCHK(h, hipFree(*p)); *p = nullptr; }
    h->seq_cap = 0;
    HIPCHK(h, hipMalloc((void**)&h->d_grid_of, (size_t)n_frames * sizeof(int)));
    HIPCHK(h, hipMalloc((void**)&h->d_seq_out, (size_t)n_frames * sizeof(mi355ndt_seq_frame)));
    HIPCHK(h, hipMalloc((void**)&h->d_stamps, (size_t)n_frames * sizeof(double)));
    h->seq_cap = (size_t)n_frames;
  }
  if (!h->d_seq) HIPCHK(h, hipMalloc((void**)&h->d_seq, sizeof(SeqState)));
  rc = ensure_seq_flags(h);
  if (rc) { h->prof = keep_prof; return rc; }
  h->h_seq_flags[0] = 0; h->h_seq_flags[1] = 0;
  SeqState q0;
  memset(&q0, 0, sizeof q0);
  q0.n_frames = n_frames;
  q0.d_trans = policy ? policy->keyframe_delta_trans : 5.0;                       // scan_matching_odom_nodelet.cpp:67-76
  q0.d_angle = policy ? policy->keyframe_delta_angle : 0.17;
  q0.d_time = policy ? policy->keyframe_delta_time : 1.0;
  HIPCHK(h, hipMemcpyAsync(h->d_seq, &q0, sizeof q0, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->d_stamps, stamps, (size_t)n_frames * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemsetAsync(h->d_grid_of, 0, (size_t)n_frames * sizeof(int), s));
  HIPCHK(h, hipMemsetAsync(h->d_results, 0, (size_t)n_frames * sizeof(mi355ndt_result), s));
  HIPCHK(h, hipMemsetAsync(h->d_ctl, 0, 2 * sizeof(SweepCtl), s));
  HIPCHK(h, hipStreamSynchronize(s));            // q0 / stamps are pageable: they must be out of the caller's memory before the pump starts
  h->ctl_idx = 0;
  h->d_grid_of_use = h->d_grid_of;
  SweepConst sc;
  make_sweep_const(h, sc);
  sc.rebase_block = 1;                           // every fine sweep also prepares the next update's re-basing (its extra workgroup)
  gauss_constants3(h->prm.outlier_ratio, h->prm.resolution, h->gauss_last);
  k_seq_begin<<<1, 64, 0, s>>>(h->d_seq, h->d_state, h->d_grid, h->d_src_cnt, h->d_stamps, h->d_seq_out, h->d_active_list, h->d_ctl, h->d_grid_of, h->d_seq_flags);
  bool stuck = false;
  rc = launch_sweep(h, sc, 1);
  // The pump: (update, sweep), (update, sweep), ... enqueued blindly, at most `depth` rounds ahead of what the device has executed;
  // whether a launch continues a frame's Newton loop, closes the frame and opens the next, or has nothing left to do is decided
  // on the device.  The host never waits for a result -- it only reads two words the device writes into mapped memory.
  const int depth = 12;
  long long enq = 0;
  const long long max_launches = (long long)n_frames * (h->prm.max_iterations + 6) + 64;
  auto t_progress = std::chrono::steady_clock::now();
  long long seen_last = -1;
  while (rc == MI355NDT_OK && !h->h_seq_flags[0] && enq < max_launches) {
    const long long seen = h->h_seq_flags[1];
    if (seen != seen_last) { seen_last = seen; t_progress = std::chrono::steady_clock::now(); }
    if (enq - seen >= depth) {
      // (a device that stops answering -- a faulted kernel -- must not leave the host spinning here)
      if (std::chrono::steady_clock::now() - t_progress > std::chrono::seconds(20)) { stuck = true; break; }
      std::this_thread::yield();
      continue;
    }
    k_seq_update<<<1, UPD_THREADS, 0, s>>>(h->d_seq, h->d_state, h->d_partials, h->rows_per_pair, h->pts_per_chunk, h->d_results, h->d_grid, h->d_src_cnt,
                                           h->d_stamps, h->d_seq_out, h->d_active_list, h->d_ctl + h->ctl_idx, h->d_grid_of, h->d_seq_flags,
                                           h->prm.step_size, h->prm.trans_epsilon, h->prm.max_iterations);
    rc = launch_sweep(h, sc, 1);
    enq++;
  }
  hipError_t e = hipEventRecord(ev[2], s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  h->prof = keep_prof;
  if (e != hipSuccess) { h->err = std::string("sequence run: ") + hipGetErrorString(e); return MI355NDT_ERR_HIP; }
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  if (stuck || !h->h_seq_flags[0]) { h->err = stuck ? "sequence run: the device stopped making progress" : "sequence run did not finish within its launch budget"; return MI355NDT_ERR_STATE; }
  HIPCHK(h, hipMemcpy(out_frames, h->d_seq_out, (size_t)n_frames * sizeof(mi355ndt_seq_frame), hipMemcpyDeviceToHost));
  if (out_results) HIPCHK(h, hipMemcpy(out_results, h->d_results, (size_t)n_frames * sizeof(mi355ndt_result), hipMemcpyDeviceToHost));
  if (stats) {
    float b_ms = 0, t_ms = 0;
    HIPCHK(h, hipEventElapsedTime(&b_ms, ev[0], ev[1]));
    HIPCHK(h, hipEventElapsedTime(&t_ms, ev[1], ev[2]));
    stats->upload_ms = std::chrono::duration<double, std::milli>(t_up1 - t_up0).count();
    stats->build_ms = b_ms;
    stats->track_ms = t_ms;
    stats->aligns = n_frames > 1 ? n_frames : 0;
    stats->update_launches = h->h_seq_flags[1];
  }
  return compute_enqueued(h);
}

}  // extern "C"


#ifdef NDT_TIMELINE
extern "C" int mi355ndt_debug_leaf_timeline(unsigned long long* out) {
  unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ltl), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ltl), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
extern "C" int mi355ndt_debug_timeline(unsigned long long* out) {
  unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
