/*
 * mi355_ndt.h -- C-ABI of the MI355X-native NDT scan-matching engine (libmi355ndt.so).
 *
 * Drop-in boundary for the one hot path of BurryChen/lv_slam: the pcl::Registration-style
 * setInputTarget / setInputSource / align surface implemented by
 *   pclomp::NormalDistributionsTransform  (include/ndt_omp/ndt_omp.h, ndt_omp_impl2.hpp)
 *   pclpca::NormalDistributionsTransform  (include/ndt_pca/ndt_pca.h,  ndt_pca_impl2.hpp)
 * and their VoxelGridCovariance target grids, as called from
 *   src/lidar_odometry/scan_matching_odom_nodelet.cpp:109-119,197,220-226,243  and
 *   include/global_graph/loop_detector.hpp:219,249-262.
 * Everything behind these entry points runs as hand-written HIP kernels on gfx950; there is
 * no CPU fallback: every call fails with MI355NDT_ERR_NO_DEVICE / MI355NDT_ERR_HIP when no GPU
 * is usable.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary; every call returns an int status.
 *   - 4x4 transforms are float[16] COLUMN-MAJOR (Eigen::Matrix4f's native layout):
 *     M(r,c) = m[c*4+r].
 *   - host point clouds are arrays of records whose first three floats are x,y,z, `stride_bytes`
 *     apart (16 for pcl::PointXYZ, 32 for pcl::PointXYZI / PointXYZRGBL).  The engine copies
 *     x,y,z out during the call; caller memory is never referenced afterwards.
 *   - a handle is bound to one HIP device and one stream and is NOT thread-safe (the reference
 *     object is driven by one thread at a time, scan_matching_odom_nodelet.cpp:56;
 *     global_graph_nodelet.cpp:672); distinct handles are independent.
 */
#ifndef MI355_NDT_H_
#define MI355_NDT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355NDT_OK               0
#define MI355NDT_ERR_BAD_HANDLE  (-1)
#define MI355NDT_ERR_BAD_ARG     (-2)
#define MI355NDT_ERR_HIP         (-3)  /* a HIP runtime call failed; see mi355ndt_last_error() */
#define MI355NDT_ERR_GRID        (-4)  /* target grid unusable: reference int32 guard
                                          (voxel_grid_covariance_omp_impl.hpp:75-84) or engine cell cap */
#define MI355NDT_ERR_NO_DEVICE   (-5)
#define MI355NDT_ERR_UNSUPPORTED (-6)  /* reserved: every configuration of the reference classes is served at present */
/* Positive: a warning that rides in mi355ndt_result.status, never a return code.  Set only under MI355NDT_OPT_ARITH = 1: the registration is of a kind the
 * tolerance arithmetic is not meant for -- fewer than MI355NDT_TOLERANCE_MIN_HITS (point, voxel) hits at the final pose (the Hessian of a handful of points:
 * ~1e-7 per f32 term times its condition number is no tolerance-level perturbation any more) or a run that stopped at the iteration cap (an oscillating run
 * amplifies any rounding difference).  The result stands; a caller that needs the default arithmetic's robustness re-runs that pair with the option off (the
 * pcl adaptor and the python mirror do so by themselves). */
#define MI355NDT_WARN_TOLERANCE_ARITH 1
#define MI355NDT_TOLERANCE_MIN_HITS   4096
#define MI355NDT_ERR_STATE       (-7)  /* align/derivatives before target+source were set */

/* pclomp::NeighborSearchMethod, include/ndt_omp/ndt_omp.h:51-56 (same enum order) */
enum mi355ndt_neighbor { MI355NDT_KDTREE = 0, MI355NDT_DIRECT26 = 1, MI355NDT_DIRECT7 = 2, MI355NDT_DIRECT1 = 3 };
/* which reference class is emulated */
enum mi355ndt_variant { MI355NDT_VARIANT_OMP = 0 /* pclomp:: */, MI355NDT_VARIANT_PCA = 1 /* pclpca:: */ };

typedef struct mi355ndt_params {
  float  resolution;              /* setResolution            ndt_omp.h:126-136; ctor default 1.0f (ndt_omp_impl2.hpp:56) */
  double step_size;               /* setStepSize              ndt_omp.h:163;     default 0.1  (impl2:57) */
  double outlier_ratio;           /* setOulierRatio [sic]     ndt_omp.h:181;     default 0.55 (impl2:58) */
  double trans_epsilon;           /* setTransformationEpsilon (pcl::Registration); default 0.1 (impl2:78) */
  int    max_iterations;          /* setMaximumIterations     (pcl::Registration); default 35  (impl2:79) */
  int    neighbor_mode;           /* setNeighborhoodSearchMethod ndt_omp.h:186;  default DIRECT7 (impl2:81) */
  int    variant;                 /* 0 = ndt_omp, 1 = ndt_pca (integer voxel weight, ndt_pca_impl2.hpp:294-296) */
  int    min_points_per_voxel;    /* voxel_grid_covariance_omp.h:204 (6) */
  double min_covar_eigvalue_mult; /* voxel_grid_covariance_omp.h:205 (0.01) */
} mi355ndt_params;

typedef struct mi355ndt_result {
  float     final_colmajor[16];   /* getFinalTransformation()          (final_transformation_, impl2:900) */
  double    trans_probability;    /* getTransformationProbability()    (impl2:187) */
  double    score;                /* score of the last derivative sweep */
  int       iterations;           /* getFinalNumIteration()            (nr_iterations_) */
  int       converged;            /* hasConverged()                    (converged_) */
  int       sweeps;               /* computeDerivatives passes (1 + steps taken); live More-Thuente case: the count the
                                     reference makes -- repeated evaluations of an unchanged pose are reused, not re-run */
  int       status;               /* per-pair status: MI355NDT_OK, MI355NDT_ERR_GRID, or MI355NDT_WARN_TOLERANCE_ARITH (> 0: a result, with a caveat) */
  long long hits_last;            /* (point,voxel) evaluations in the last sweep */
} mi355ndt_result;

/* one searchable voxel as the sweep sees it (parity hook for VoxelGridCovariance::Leaf,
 * voxel_grid_covariance_omp.h:92-187) */
typedef struct mi355ndt_voxel {
  int32_t idx;        /* linear cell index (voxel_grid_covariance_omp_impl.hpp:223) */
  int32_t n;          /* nr_points; -1 = eigen / inverse failure (impl:339,363) -> never hit */
  double  mean[3];    /* mean_ */
  float   icov[9];    /* float(icov_), row-major (ndt_omp_impl2.hpp:576) */
  int32_t weight;     /* ndt_pca (int)dimension_2d_ (voxel_grid_covariance_pca.h:222-226); 1 for ndt_omp */
} mi355ndt_voxel;

typedef struct mi355ndt_profile {
  double    sweep_ms;        /* sum of derivative-sweep kernel durations (HIP events on the engine's stream) */
  long long sweep_launches;
  double    sweep_alg_bytes; /* algorithmic bytes of those launches: sum over active pairs of N*(12+4K) + 64*hits */
  long long sweep_hits;      /* (point,voxel) evaluations in those launches */
  long long sweep_points;    /* source points swept in those launches */
  double    build_ms;        /* sum of target-build durations (all voxelisation kernels + sort) */
  long long build_launches;  /* number of batch builds */
  double    build_alg_bytes; /* algorithmic bytes of the builds (DESIGN.md B_build) */
  double    update_ms;       /* sum of Newton-update kernel durations */
  long long update_launches;
  long long async_fallbacks; /* one-launch aligns that gave up (a wave's ticket never came within its poll budget) and were re-run by the
                                round-based path: same results, never an error */
  long long stream_launches; /* stream mode: persistent launches (one per submitted batch + flushes) */
  long long stream_carried;  /* stream mode: pairs handed over from one launch to the next (stragglers that finished under a later batch) */
  long long stream_redone;   /* stream mode: batches re-run synchronously (build plan exceeded, or a launch gave up) */
  long long cloud_uploads;   /* host clouds staged and sent over PCIe (set_target / set_source / batch_set_* / calculate_score / prefilter), counted always */
  long long cloud_upload_bytes;
  long long cloud_transfers;  /* host-to-device transfers those clouds travelled in (mi355ndt_batch_set_clouds / stream_submit_host send up to eight clouds per transfer) */
  long long cloud_promotions; /* mi355ndt_promote_source_to_target calls (device-to-device instead of an upload) */
  long long stream_reserved_slots; /* stream mode: workgroup slots the persistent launches leave to the next batch's build (the effective MI355NDT_OPT_STREAM_RESERVE; 0 = the build runs between the launches) */
  long long stream_launch_slots;   /* stream mode: workgroups of one persistent launch (CUs x workgroups per CU - the reserved slots) */
} mi355ndt_profile;

typedef struct mi355ndt_handle mi355ndt_handle;

const char* mi355ndt_version(void);
int mi355ndt_device_count(void);                       /* number of usable HIP devices (0 on a CPU-only box) */
/* NUMA node of the host CPUs closest to `device` (-1 = unknown).  Host clouds are staged by CPU threads: on a two-socket host
 * a staging thread that reads the clouds across the socket link runs at half the rate, so keep the threads that own the
 * clouds (and call set_source / set_target) on this node; the engine's own staging threads pin themselves to it. */
int mi355ndt_host_numa_node(int device);

/* ctor defaults of NormalDistributionsTransform() (ndt_omp_impl2.hpp:53-83) */
int mi355ndt_default_params(mi355ndt_params* p);

/* replaces: `pclomp::NormalDistributionsTransform<PS,PT> reg;` (scan_matching_odom_nodelet.cpp:328,
 * registrations.cpp:78).  `params` may be NULL (defaults). */
int mi355ndt_create(const mi355ndt_params* params, int device, mi355ndt_handle** out);
int mi355ndt_destroy(mi355ndt_handle* h);

/* replaces the setters of ndt_omp.h:109-203.  A change of variant / min points / eigenvalue multiplier re-voxelises the current targets.
 * A change of `resolution` does what setResolution does (ndt_omp.h:126-136: `if (input_) init();`): the target is re-voxelised only when a
 * SOURCE has been set (mi355ndt_set_source / a bound batch); without one the resident grid keeps its leaf size until the next
 * mi355ndt_set_target, while the Gauss constants of the next align already use the new value (ndt_omp_impl2.hpp:93-100) -- the reference's
 * behaviour, quirk included.  Deviation: with KDTREE or the live More-Thuente configuration (radius searches over the grid) the target is
 * re-voxelised at once in either case. */
int mi355ndt_set_params(mi355ndt_handle* h, const mi355ndt_params* params);
int mi355ndt_get_params(const mi355ndt_handle* h, mi355ndt_params* out);

/* bind the engine to a caller-owned hipStream_t (NULL = the engine's own stream) */
int mi355ndt_set_stream(mi355ndt_handle* h, void* hip_stream);
const char* mi355ndt_last_error(const mi355ndt_handle* h);

/* ---- single registration: the pcl::Registration surface (pair slot 0) --------------------------- */

/* replaces setInputTarget(cloud) -> init() -> VoxelGridCovariance::filter(true)
 * (ndt_omp.h:116-121, 270-277; voxel_grid_covariance_omp_impl.hpp:48-370) */
int mi355ndt_set_target(mi355ndt_handle* h, const void* pts, size_t n, size_t stride_bytes);
/* replaces pcl::Registration::setInputSource(cloud) */
int mi355ndt_set_source(mi355ndt_handle* h, const void* pts, size_t n, size_t stride_bytes);
/* The cloud last handed over as SOURCE becomes the target (device-to-device copy + init()), without crossing PCIe again: the nodelet's keyframe switch
 * `key = filtered; reg_s2k.setInputTarget(key);` (scan_matching_odom_nodelet.cpp:240-243) right after `filtered` was aligned as the source.  Same result as
 * mi355ndt_set_target on the same cloud.  MI355NDT_ERR_STATE without a source (single-registration surface only). */
int mi355ndt_promote_source_to_target(mi355ndt_handle* h);
/* replaces align(output, guess) -> computeTransformation(output, guess) (ndt_omp_impl2.hpp:87-188) */
int mi355ndt_align(mi355ndt_handle* h, const float guess_colmajor[16], mi355ndt_result* out);
/* the `output` cloud of align(): source transformed by the final pose (f32, PCL 1.8 scalar form).
 * Writes x,y,z into records `stride_bytes` apart. */
int mi355ndt_get_aligned(mi355ndt_handle* h, void* out_pts, size_t stride_bytes);

/* pcl::Registration::transformation_ / previous_transformation_ as the last align() left them: float(exp(delta_p)) of the
 * last Newton step (ndt_omp_impl2.hpp:163; what getLastIncrementalTransformation() returns) and of the step before it
 * (impl2:134); both Identity when no step was taken.  Column-major 4x4; either pointer may be NULL.  `pair` = batch slot
 * (0 for the single-registration surface). */
int mi355ndt_get_incremental(mi355ndt_handle* h, int pair, float transformation_colmajor[16], float previous_colmajor[16]);

/* replaces pcl::Registration::getFitnessScore(max_range) as called by the loop-closure path
 * (include/global_graph/loop_detector.hpp:249-262; identical recipe in-tree:
 * src/global_graph/information_matrix_calculator.cpp:53-87): source moved by the final pose of the last align()
 * (identity before any align), exact nearest target point per source point, mean of the SQUARED distances that are
 * <= max_range (squared distance vs max_range, as the reference compares them); DBL_MAX when nothing is in range.
 * Works for any target cloud, also one whose voxel grid could not be built (status MI355NDT_ERR_GRID: exhaustive search). */
int mi355ndt_get_fitness_score(mi355ndt_handle* h, double max_range, double* score, long long* n_inliers);
/* same with an explicit transform (column-major 4x4) */
int mi355ndt_fitness_score_T(mi355ndt_handle* h, const float T_colmajor[16], double max_range, double* score, long long* n_inliers);

/* replaces calculateScore(cloud) (ndt_omp.h:232, ndt_omp_impl2.hpp:1006-1040; ndt_pca.h:244, ndt_pca_impl2.hpp:1013-1047): the negative
 * log-likelihood of an ALREADY TRANSFORMED cloud against the target grid, all in f64: per point the radiusSearch(point, resolution)
 * neighbourhood over the f32 leaf centroids (no nr_points re-check, voxel_grid_covariance_omp.h:505-534), per neighbour
 * (-d1 * exp(-d2 * x'^T icov x' / 2) - d3) / neighbourhood.size(), summed and divided by cloud.size().  d1, d2, d3 are the members
 * gauss_d1_/d2_/d3_ as the reference holds them when the call is made: the constructor's values (resolution 1.0, outlier ratio 0.55,
 * impl2:70-76) until the first align(), after that those of the last align()'s parameters (impl2:93-100).  No caller inside lv_slam.
 * `pts`: host records, x,y,z first, `stride_bytes` apart (typically the output cloud of align()). */
int mi355ndt_calculate_score(mi355ndt_handle* h, const void* pts, size_t n, size_t stride_bytes, double* score);

/* replaces the two static convertTransform helpers (ndt_omp.h:209-228): x = [x, y, z, roll, pitch, yaw] (f64) ->
 * Translation3f * AngleAxisf(roll, X) * AngleAxisf(pitch, Y) * AngleAxisf(yaw, Z) as a 4x4 f32 matrix, column-major -- what
 * Eigen::Affine3f::matrix() holds; evaluated in f32 the way Eigen 3.3 does (AngleAxis::toRotationMatrix, then the two 3x3 products).
 * Plain host arithmetic, no device needed. */
int mi355ndt_convert_transform(const double x[6], float out_colmajor[16]);

/* Engine options that are no parameter of the reference classes. */
enum mi355ndt_option {
  /* Evaluation order of the three-term f32 sums inside updateDerivatives (ndt_omp_impl2.hpp:581, 594-613: 4-wide Eigen inner
   * products whose fourth term is a structural zero).  The reference leaves that order to Eigen 3.3 and the SSE level it is
   * built for (CMakeLists.txt:6,11); it cannot be observed without the reference's libraries.  0 (default): (t0 + t1) + t2,
   * Eigen's scalar redux = the order every committed parity fixture was made with.  1: (t0 + t2) + t1, the lane pairing of Eigen 3.3's SSE
   * predux<Packet4f>.  Same cost; lets a maintainer who pins the reference on a real build (tools/pin_reference/) select the
   * order that build shows.  Poses differ in their last bits for about half of all pairs. */
  MI355NDT_OPT_F32_SUM_ORDER = 1,
  /* 1 (default): a batch align runs as ONE persistent launch in which every pair goes through its own Newton loop to its own end, as
   * every align() of the reference does (ndt_omp_impl2.hpp:131-183); 0: lockstep rounds of (update, sweep) launches over the pairs
   * still iterating.  Same results bit for bit (a pair's sums never depend on what runs beside it); the environment variable
   * MI355NDT_ASYNC=0 sets the default to 0 for engines created afterwards.  The one launch is used when the batch offers more work items
   * than the GPU has resident waves (smaller batches are faster in rounds); 2 = use it for every batch size (testing).  The latency mode and the live More-Thuente configuration
   * always take the round-based path. */
  MI355NDT_OPT_ASYNC_ALIGN = 2,
  /* Test hook, default -1 (off).  n >= 0: inside a one-launch align the wave that claims position n of ring 0 gives up exactly as a wave whose
   * ticket never came within its poll budget does: the launch ends early and mi355ndt_batch_align re-runs the batch through the round-based path
   * (mi355ndt_profile.async_fallbacks counts it).  Lets the test-suite exercise the fallback, which a healthy device never takes. */
  MI355NDT_OPT_DEBUG_ASYNC_ABORT = 3,
  /* Stream mode (mi355ndt_stream_*): a launch hands its last `value` unfinished pairs over to the next launch instead of iterating them alone
   * on an otherwise idle GPU.  -1 (default): as many as it takes to keep every resident wave busy (resident waves / work items per sweep);
   * 0: never (every launch runs its own pairs to the end: pipelining of the host side only).  No result bit depends on it. */
  MI355NDT_OPT_STREAM_THRESHOLD = 4,
  /* Stream mode, read by mi355ndt_stream_begin: the persistent launches leave `value` workgroup slots free (rounded down to a multiple of 8)
   * and the NEXT batch's target build runs on a stream of its own beside the launch instead of between two launches -- the build streams
   * through HBM, the launch saturates the vector ALUs.  Needs >= 3 contexts (a context is rebuilt one launch earlier, so its pairs are carried
   * through one launch less).  -1 (default): the environment variable MI355NDT_STREAM_RESERVE, else for DIRECT1 128 with clouds of up to 98,304 points and 96 beyond (launches that
   * wait for their point stream leave the vector ALUs to the build: +5-10 %), for DIRECT7 with batches of up to 768 x 65,536 target points 64 (ndt_omp) / 32 (ndt_pca)
   * (the launch pays for the slots in full, the whole build disappears under it: +4-6 % at 271 pairs, +22 % at 64) and 0 = off for every other search.  No result bit depends on it. */
  MI355NDT_OPT_STREAM_RESERVE = 5,
  /* Test hook, default 0xFF (off).  Bit x clear: the workgroups of a one-launch align that serve ring x leave at once, as if XCD x held no
   * workgroup of this launch (another engine's launch filling it).  The launch must still finish every pair -- waiting waves serve the
   * published positions of other rings (ndt_async.hpp) -- with the same bits; mask 0 is refused. */
  MI355NDT_OPT_DEBUG_ASYNC_RINGS = 6,
  /* Arithmetic of the derivative sweep.  0 (default): every f32 / f64 operation of updateDerivatives (ndt_omp_impl2.hpp:566-619) as the CPU
   * restatement of the reference performs it, one rounding per operation, 43 f64 sums per lane -- results equal the CPU restatement's bit for bit.
   * 1: tolerance arithmetic, held to north_star's SE(3) tolerance (trans < 1e-4 m, rot < 1e-5 rad against the reference arithmetic) instead:
   * fused multiply-adds, the hardware exp2, a symmetric inverse covariance (21 symmetric + 9 point-Hessian sums instead of 36), f32 sums per
   * work item (512 points) widened to f64 from there on, leaf sums of the target build as a tree instead of in input order.  The point
   * transform and the voxel lookup are untouched (same leaves for the same pose).  Served for DIRECT1 / DIRECT7 with the dead More-Thuente
   * loop (every configuration lv_slam ships); other configurations ignore the option.  Measured cost in accuracy: BASELINE.md 5a. */
  MI355NDT_OPT_ARITH = 7
};
int mi355ndt_set_option(mi355ndt_handle* h, int option, int value);
int mi355ndt_get_option(const mi355ndt_handle* h, int option, int* value);

/* parity hooks ------------------------------------------------------------------------------------ */
/* one computeDerivatives sweep (ndt_omp_impl2.hpp:196-305) at tangent p = [upsilon; omega]:
 * points transformed by float(exp(p)), Jacobian from the same matrix (impl2:900-907).
 * H is 6x6 row-major and NOT symmetric. */
int mi355ndt_derivatives(mi355ndt_handle* h, const double p[6], double* score, double g[6], double H[36], long long* hits);
/* computeHessian + updateHessian (ndt_omp_impl2.hpp:622-714) at tangent p: the f64 Hessian-only pass over kd-tree
 * neighbourhoods that computeStepLengthMT runs after its More-Thuente loop iterated (impl2:999-1000; live only when
 * step_size <= transformation_epsilon/2).  Cloud moved by float(exp(p)).  H is 6x6 row-major. */
int mi355ndt_compute_hessian(mi355ndt_handle* h, const double p[6], double H[36]);
/* same sweep with an explicit point transform (column-major 4x4) and Jacobian rotation (row-major 3x3),
 * i.e. the first sweep of align() where the cloud is moved by the caller's guess (impl2:102-129) */
int mi355ndt_derivatives_T(mi355ndt_handle* h, const float T_colmajor[16], const float Rj_rowmajor[9],
                           double* score, double g[6], double H[36], long long* hits);
/* target grid of pair `pair`: bounds (min_b_, max_b_, div_b_) and searchable voxels in ascending idx order */
int mi355ndt_get_grid(mi355ndt_handle* h, int pair, int min_b[3], int max_b[3], int div_b[3], int* n_voxels);
int mi355ndt_get_voxels(mi355ndt_handle* h, int pair, mi355ndt_voxel* out, size_t capacity);

/* ---- batch: n independent (target, source, guess) triples on one GPU ---------------------------- */
/* (BASELINE.json configs 3-5; the single-registration calls above are the n = 1 case) */

int mi355ndt_batch_reserve(mi355ndt_handle* h, int n_pairs, size_t max_target_pts, size_t max_source_pts);
/* Host uploads into pair slot `pair` (same record convention as set_target / set_source).  Asynchronous: the call returns as
 * soon as x,y,z of the caller's records sit in one of the engine's pinned staging slots (caller memory is not referenced
 * afterwards); the PCIe transfer and the AoS -> SoA kernel run on the engine's copy stream while the next cloud is staged, and
 * the next build / align waits for them.  These two calls -- and only these -- may be issued from several threads at once for
 * DIFFERENT pair slots (staging is the CPU-bound part of a host-cloud batch). */
int mi355ndt_batch_set_target(mi355ndt_handle* h, int pair, const void* pts, size_t n, size_t stride_bytes);
int mi355ndt_batch_set_source(mi355ndt_handle* h, int pair, const void* pts, size_t n, size_t stride_bytes);
/* A whole batch of host clouds in one call: pairs first_pair .. first_pair + n - 1, one pointer and one point count per cloud
 * (`targets` or `sources` may be NULL to upload only the other side), records `stride_bytes` apart; n_threads staging threads
 * of the engine's own (<= 0: 8) share the pairs.  Same asynchrony as the per-cloud calls. */
int mi355ndt_batch_set_clouds(mi355ndt_handle* h, int first_pair, int n, const void* const* targets, const size_t* target_counts,
                              const void* const* sources, const size_t* source_counts, size_t stride_bytes, int n_threads);
/* zero-copy: use device-resident SoA buffers laid out [pair][3][pitch] (x row, y row, z row of `pitch`
 * floats each).  counts are HOST arrays of n_pairs ints.  The buffers must stay valid until replaced. */
int mi355ndt_batch_bind_device(mi355ndt_handle* h, int n_pairs,
                               const float* d_targets, const int* target_counts, size_t target_pitch,
                               const float* d_sources, const int* source_counts, size_t source_pitch);
/* voxelise every target (setInputTarget for all pairs); asynchronous on the engine's stream */
int mi355ndt_batch_build_targets(mi355ndt_handle* h);
/* align every pair; guesses = n_pairs x 16 floats column-major; out = n_pairs results. Synchronous. */
int mi355ndt_batch_align(mi355ndt_handle* h, const float* guesses_colmajor, mi355ndt_result* out);
int mi355ndt_batch_size(const mi355ndt_handle* h);
/* Pose records of the last mi355ndt_batch_align for the multi-GPU gather, written on the device into a caller-owned DEVICE
 * buffer of `capacity` 96-byte records {float final[16] column-major; float score; int32 iterations; int32 converged;
 * int32 pair_id; int32 pad[4]}: record k describes batch slot k and carries pair_id = id_base + k * id_stride (round-robin
 * sharding: base = rank, stride = world size); records k >= batch size carry pair_id = -1.  The buffer can go straight into
 * an RCCL all-gather: no host hop.  Returns after the records are complete (the engine's stream is synchronised). */
int mi355ndt_batch_pose_records(mi355ndt_handle* h, int id_base, int id_stride, void* d_records, size_t capacity);

/* ---- stream mode: batches arrive one after the other and overlap on the GPU --------------------------------------------------- */
/* The reference's odometry node consumes a continuous stream of frames (scan_matching_odom_nodelet.cpp:144-183: one cloud_callback per
 * scan); BASELINE config 3 streams batches of independent scan pairs through one GPU.  mi355ndt_batch_align is synchronous: build, ONE
 * persistent launch, results on the host, and only then the next batch -- the last pairs of a batch iterate alone on an idle GPU (the
 * tail), and the host's own work between two batches is dead time.  In stream mode the engine keeps `n_contexts` batches resident:
 *   submit(k)  enqueues batch k's target build, its persistent launch and the result copies, and returns at once;
 *   a launch ends as soon as its unfinished pairs can no longer keep the GPU busy; those pairs are SUSPENDED and become the first tickets of
 *              the next launch, where they finish under batch k+1's bulk (a pair runs the same code on the same operands whichever launch
 *              serves it: results are bit-identical to mi355ndt_batch_align's);
 *   collect(k) blocks until every pair of batch k is finalised -- normally a launch or two after its own; if nothing newer has been
 *              submitted it flushes the stragglers itself.
 * Batch k's context is recycled by submit(k + n_contexts), so at most n_contexts batches may be uncollected.  A pair is carried through at
 * most n_contexts - 2 further launches (one with two contexts): a batch is complete one launch before its context is recycled, so that the
 * host can collect it and enqueue the next build while a launch is still running.  Four contexts suit workloads whose launches end in long
 * tails (BASELINE config 5), three or four anything else.
 * Inputs are device-resident SoA buffers as in mi355ndt_batch_bind_device (zero-copy); batch k's buffers must stay valid and unchanged
 * until collect(k) has returned.  Parameters and options are those of the handle at mi355ndt_stream_begin; the handle's single-registration
 * and batch calls are unavailable (MI355NDT_ERR_STATE) between begin and end.  Served for every configuration the one-launch align
 * serves (DIRECT1/7/26 and ndt_omp KDTREE, dead More-Thuente loop: everything lv_slam ships); others are processed synchronously
 * inside submit. */
int mi355ndt_stream_begin(mi355ndt_handle* h, int n_contexts /* 2..4 */, int max_pairs, size_t max_target_pts, size_t max_source_pts);
int mi355ndt_stream_submit(mi355ndt_handle* h, int n_pairs, const float* d_targets, const int* target_counts, size_t target_pitch,
                           const float* d_sources, const int* source_counts, size_t source_pitch, const float* guesses_colmajor,
                           long long* batch_id);
/* The same for HOST clouds -- what the node has (scan_matching_odom_nodelet.cpp:144-183: pcl::PointCloud records arriving one callback at a time): arrays of
 * n_pairs pointers to the targets' / sources' records (x, y, z as the first three floats of every `stride_bytes`-long record, as mi355ndt_set_target takes
 * them) and their point counts.  The clouds are staged by `n_threads` threads of the engine (0 = 8) into the batch context's pinned slots, cross PCIe and land in
 * the context's own device buffers while the launches of earlier batches run; returns as soon as the caller's memory is no longer needed.  Counts must not
 * exceed what mi355ndt_stream_begin was told.  Results: word for word those of mi355ndt_batch_set_clouds + batch_build_targets + batch_align. */
int mi355ndt_stream_submit_host(mi355ndt_handle* h, int n_pairs, const void* const* targets, const size_t* target_counts, const void* const* sources,
                                const size_t* source_counts, size_t stride_bytes, const float* guesses_colmajor, int n_threads, long long* batch_id);
int mi355ndt_stream_collect(mi355ndt_handle* h, long long batch_id, mi355ndt_result* out);
/* Pose records for the multi-GPU gather (the 96-byte layout of mi355ndt_batch_pose_records) of the NEXT batch submitted: `d_records` is a
 * caller-owned DEVICE buffer of `capacity` records (>= that batch's pairs); record b is written by the device -- by the very wave that
 * finalises pair b, inside the persistent launch: no packing kernel, no host hop -- with pair_id = id_base + b * id_stride, the other rows
 * carry pair_id = -1.  Complete when mi355ndt_stream_collect of that batch has returned; the buffer can then go straight into an RCCL
 * all-gather.  NULL = no records for the next batch (the default). */
int mi355ndt_stream_pose_records(mi355ndt_handle* h, void* d_records, size_t capacity, int id_base, int id_stride);
/* Pose records (the 96-byte layout of mi355ndt_batch_pose_records) of a COLLECTED batch, packed on the host from its results into
 * `records` (host memory, `capacity` records; rows >= n carry pair_id = -1).  In stream mode the GPU is busy with the next batch's launch when
 * a batch is collected -- a packing kernel would wait for that launch -- so the 26 KB of a 271-pair batch are packed here and go to the device
 * with the caller's own copy.  Plain host arithmetic, no device needed. */
int mi355ndt_pack_pose_records(const mi355ndt_result* results, int n, int id_base, int id_stride, void* records, size_t capacity);
int mi355ndt_stream_end(mi355ndt_handle* h);      /* collects nothing: outstanding batches are dropped after the device has drained */

/* ---- latency mode: one frame at a time, as the live nodelet runs ---------------------------------------- */
/* Opt-in fine-grained derivative sweep for SMALL batches (a single registration above all): work items of 128 points dealt over
 * every wave of the GPU instead of 512-point items (a 65,536-point pair then offers 512 items to the 2,048 resident waves instead
 * of 128) and no work-queue atomics.  The per-pair f64 summation tree is this mode's own (fixed, deterministic, independent of
 * the batch a pair is in), NOT the batch mode's: a sweep agrees with the batch mode's to the rounding of the f64 sums (~1e-16
 * relative, inside the 1e-11 sweep bar); poses are normally bit-identical, but that is not promised across the two modes.
 * Applies to DIRECT1 / DIRECT7 with step_size > transformation_epsilon / 2 (everything lv_slam ships); other configurations
 * keep the batch kernels.  Default: off. */
int mi355ndt_set_latency_mode(mi355ndt_handle* h, int on);

/* keyframe policy of ScanMatchingOdomNodelet (scan_matching_odom_nodelet.cpp:67-76; launch/dlo_kitti.launch:51-53) */
typedef struct mi355ndt_seq_params {
  double keyframe_delta_trans;   /* [m]   default 5.0 in code, 10 in the KITTI launch file */
  double keyframe_delta_angle;   /* [rad] default 0.17 */
  double keyframe_delta_time;    /* [s]   default 1.0  */
} mi355ndt_seq_params;

typedef struct mi355ndt_seq_frame {
  double odom_colmajor[16];      /* odom_velo = key_pose * tf_s2k (:234): pose of the scan in the first keyframe's frame, 4x4 f64 column-major */
  float  tf_s2k_colmajor[16];    /* getFinalTransformation() of the scan-to-keyframe align (:222, :226) */
  double trans_probability;      /* getTransformationProbability() of that align */
  double dx, da, dt;             /* the three keyframe-test quantities (:237-239) */
  int    key_id;                 /* the keyframe this scan was matched against */
  int    new_keyframe;           /* 1: the test fired and this scan became the keyframe (:240-247) */
  int    iterations, converged;  /* of the (last) align of this scan */
  int    aligns;                 /* 2 for frame 1 (:223-227), 1 otherwise, 0 for frame 0 */
  int    pad;
} mi355ndt_seq_frame;

typedef struct mi355ndt_seq_stats {
  double    upload_ms;           /* host clouds -> HBM (staging + PCIe + AoS->SoA), host clock */
  double    build_ms;            /* voxel grids of all frames (one batched build), HIP events */
  double    track_ms;            /* frame 1 .. n-1: every align + the policy, HIP events; no host round trip inside */
  long long aligns;              /* n_frames (frame 1 twice, frame 0 never) */
  long long update_launches;     /* k_seq_update launches executed, incl. the pump's overshoot at the end */
} mi355ndt_seq_stats;

/* replaces n_frames calls of the nodelet's cloud_callback -> matching_s2k (scan_matching_odom_nodelet.cpp:144-183, 192-261):
 * the frames are uploaded, every frame's voxel grid is built (one batched build: every frame is a potential keyframe), and then
 * frame after frame is aligned against its keyframe with the guess, the keyframe test and the target switch decided ON THE DEVICE
 * at the tail of the Newton-update kernel -- the host pumps (update, sweep) launches without waiting for any result.  Uses the
 * handle's registration parameters (the nodelet's: resolution 1.0, DIRECT1, eps 0.01, 64 iterations, :109-119) and the
 * fine-grained sweep of mi355ndt_set_latency_mode.  `stamps` = header.stamp of every frame in seconds.  out_frames: n_frames
 * records; out_results (may be NULL): the engine-level result of every frame's last align; stats (may be NULL).
 * The call CONSUMES the handle's batch state: the frames replace whatever batch / single registration was resident (target, source,
 * voxel grids, last align), and none is resident afterwards -- set_target / set_source again before the next align(), or use a
 * handle of its own for the drive. */
int mi355ndt_sequence_run(mi355ndt_handle* h, int n_frames, const void* const* clouds, const size_t* counts, size_t stride_bytes,
                          const double* stamps, const mi355ndt_seq_params* policy,
                          mi355ndt_seq_frame* out_frames, mi355ndt_result* out_results, mi355ndt_seq_stats* stats);

/* ---- prefilter: the step immediately upstream of the path ------------------------------------------- */
/* replaces PrefilteringNodelet::distance_filter + downsample (src/lidar_odometry/prefiltering_nodelet.cpp:137-181,
 * launch/dlo_kitti.launch:30-36): keep near < |p| < far, then pcl::VoxelGrid centroid down-sampling with leaf
 * `downsample_resolution` (<= 0: none), output ordered by ascending voxel index.  The result stays on the GPU
 * (mi355ndt_use_prefiltered) and is copied to out_pts (x,y,z records, may be NULL) when it fits out_capacity. */
int mi355ndt_prefilter(mi355ndt_handle* h, const void* pts, size_t n, size_t stride_bytes,
                       int use_distance_filter, double distance_near, double distance_far, float downsample_resolution,
                       void* out_pts, size_t out_capacity, size_t out_stride_bytes, size_t* n_out);
/* install the last prefilter result as the registration source (role 1) or target (role 2) without a host round trip */
int mi355ndt_use_prefiltered(mi355ndt_handle* h, int role);

/* profiling: HIP-event timing of the engine's own kernels on the engine's stream */
int mi355ndt_profile_enable(mi355ndt_handle* h, int on);
int mi355ndt_profile_reset(mi355ndt_handle* h);
int mi355ndt_profile_get(mi355ndt_handle* h, mi355ndt_profile* out);
int mi355ndt_synchronize(mi355ndt_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MI355_NDT_H_ */
