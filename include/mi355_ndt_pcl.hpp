// mi355_ndt_pcl.hpp -- header-only pcl::Registration adaptor over the C-ABI of mi355_ndt.h.
//
// Compiled ONLY on the ROS/PCL host (it needs PCL + Eigen, which do not exist on the GPU build box); it is
// the reference-side binding a maintainer adds to lv_slam: replace
//     pclpca::NormalDistributionsTransform<PointT, PointT> reg_s2k;      // scan_matching_odom_nodelet.cpp:328
//     pclomp::NormalDistributionsTransform<PointT, PointT>               // src/global_graph/registrations.cpp:78
// by
//     mi355ndt::NormalDistributionsTransform<PointT, PointT> reg_s2k(MI355NDT_VARIANT_PCA);
// and link -lmi355ndt.  Every call site (setInputTarget / setInputSource / align / getFinalTransformation /
// hasConverged and the ndt_omp.h:109-203 setters) compiles unchanged.
#pragma once
#include <pcl/registration/registration.h>
#include <stdexcept>
#include <limits>
#include <vector>
#include "mi355_ndt.h"

namespace mi355ndt {

enum NeighborSearchMethod { KDTREE = MI355NDT_KDTREE, DIRECT26 = MI355NDT_DIRECT26, DIRECT7 = MI355NDT_DIRECT7, DIRECT1 = MI355NDT_DIRECT1 };

template <typename PointSource, typename PointTarget>
class NormalDistributionsTransform : public pcl::Registration<PointSource, PointTarget> {
 protected:
  typedef pcl::Registration<PointSource, PointTarget> Base;
  typedef typename Base::PointCloudSource PointCloudSource;
  typedef typename Base::PointCloudTarget PointCloudTarget;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  using Base::reg_name_; using Base::input_; using Base::target_; using Base::nr_iterations_; using Base::max_iterations_;
  using Base::final_transformation_; using Base::transformation_; using Base::previous_transformation_;
  using Base::transformation_epsilon_; using Base::converged_; using Base::update_visualizer_;

 public:
  explicit NormalDistributionsTransform(int variant = MI355NDT_VARIANT_OMP, int device = 0) : h_(nullptr), trans_probability_(0), dev_src_(nullptr), dev_src_n_(0) {
    reg_name_ = "NormalDistributionsTransform";
    mi355ndt_default_params(&prm_);            // ctor defaults of ndt_omp_impl2.hpp:53-83
    prm_.variant = variant;
    transformation_epsilon_ = prm_.trans_epsilon;
    max_iterations_ = prm_.max_iterations;
    if (mi355ndt_create(&prm_, device, &h_) != MI355NDT_OK) throw std::runtime_error("mi355ndt_create failed (no MI355X?)");
  }
  virtual ~NormalDistributionsTransform() { if (h_) mi355ndt_destroy(h_); }

  void setNumThreads(int) {}                                   // ndt_omp.h:109 (OpenMP only)
  inline void setInputTarget(const PointCloudTargetConstPtr& cloud) {   // ndt_omp.h:116-121
    Base::setInputTarget(cloud);
    push();
    // the keyframe switch of scan_matching_odom_nodelet.cpp:240-243 -- `key = filtered; setInputTarget(key)` -- names the very cloud the device
    // holds as the source: device-to-device instead of a second trip over PCIe (same grid, mi355ndt_promote_source_to_target)
    if (same_cloud(cloud.get(), cloud->points.size()) && mi355ndt_promote_source_to_target(h_) == MI355NDT_OK) return;
    mi355ndt_set_target(h_, cloud->points.data(), cloud->points.size(), sizeof(PointTarget));
  }
  // ndt_omp.h:126-136: `if (input_) init();` -- the engine keeps the grid when it has no source yet (mi355ndt_set_params), and it has one
  // exactly when pcl::Registration::input_ is set: setInputSource below hands the cloud over at once
  inline void setResolution(float r) { if (prm_.resolution != r) { prm_.resolution = r; push(); } }
  // The cloud goes to the device HERE, once: computeTransformation does not send it again as long as input_ still names the same cloud object
  // and size (pcl::Registration holds a ConstPtr: a caller that rewrites the points of a cloud it has handed over must call setInputSource
  // again, as PCL's own cached kd-trees require).  A PCL whose Registration::setInputSource is not virtual and is called through a base
  // pointer never comes through here; computeTransformation then finds no record of the cloud and uploads it itself.
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) {
    Base::setInputSource(cloud);
    dev_src_ = nullptr; dev_src_n_ = 0;
    if (cloud && mi355ndt_set_source(h_, cloud->points.data(), cloud->points.size(), sizeof(PointSource)) == MI355NDT_OK) { dev_src_ = cloud.get(); dev_src_n_ = cloud->points.size(); }
  }
  inline float getResolution() const { return prm_.resolution; }
  inline double getStepSize() const { return prm_.step_size; }
  inline void setStepSize(double s) { prm_.step_size = s; }
  inline double getOulierRatio() const { return prm_.outlier_ratio; }
  inline void setOulierRatio(double o) { prm_.outlier_ratio = o; }
  inline void setNeighborhoodSearchMethod(NeighborSearchMethod m) { prm_.neighbor_mode = m; }
  // not in the reference: the engine's opt-in fine-grained sweep for one registration at a time (mi355ndt_set_latency_mode) --
  // what a live nodelet wants; results stay inside the parity tolerance, the f64 summation tree is that mode's own
  inline void setLatencyMode(bool on) { mi355ndt_set_latency_mode(h_, on ? 1 : 0); }
  // not in the reference: evaluation order of the three-term f32 sums of updateDerivatives (mi355ndt_set_option, MI355NDT_OPT_F32_SUM_ORDER)
  inline void setF32SumOrder(int order) { mi355ndt_set_option(h_, MI355NDT_OPT_F32_SUM_ORDER, order); }
  // not in the reference: the tolerance arithmetic (MI355NDT_OPT_ARITH; held to 1e-4 m / 1e-5 rad instead of to the CPU restatement's bits, 1.2-1.6x faster on
  // full scans); a registration it is not meant for is re-run in the default arithmetic by computeTransformation
  inline void setArithmetic(int mode) { mi355ndt_set_option(h_, MI355NDT_OPT_ARITH, mode); }
  // ndt_omp.h:232 (impl2:1006-1040): negative log-likelihood of an already transformed cloud against the target grid
  inline double calculateScore(const PointCloudSource& cloud) const {
    double s = 0;
    // (0 is a legitimate score -- nothing in range --, so a failed call must not look like one)
    if (mi355ndt_calculate_score(h_, cloud.points.data(), cloud.points.size(), sizeof(PointSource), &s) != MI355NDT_OK) return std::numeric_limits<double>::quiet_NaN();
    return s;
  }
  // ndt_omp.h:209-228: [x, y, z, roll, pitch, yaw] -> Translation * AngleAxis(roll, X) * AngleAxis(pitch, Y) * AngleAxis(yaw, Z), f32
  static void convertTransform(const Eigen::Matrix<double, 6, 1>& x, Eigen::Matrix4f& trans) {
    float m[16];
    mi355ndt_convert_transform(x.data(), m);
    trans = Eigen::Map<const Eigen::Matrix4f>(m);
  }
  static void convertTransform(const Eigen::Matrix<double, 6, 1>& x, Eigen::Affine3f& trans) { convertTransform(x, trans.matrix()); }
  // not in the reference: the engine behind the object (mi355ndt_set_option, mi355ndt_profile_get, ...)
  inline mi355ndt_handle* handle() const { return h_; }
  inline double getTransformationProbability() const { return trans_probability_; }
  inline int getFinalNumIteration() const { return nr_iterations_; }
  // pclpca's getTargetCells() (ndt_pca.h:129-133) hands out the VoxelGridCovariance itself; that container lives on the GPU here,
  // so the accessor returns the searchable leaves (cell index, nr_points, mean, inverse covariance, pca weight) in std::map order
  inline std::vector<mi355ndt_voxel> getTargetCells() const {
    int mn[3], mx[3], dv[3], n = 0;
    std::vector<mi355ndt_voxel> v;
    if (mi355ndt_get_grid(h_, 0, mn, mx, dv, &n) != MI355NDT_OK || n <= 0) return v;
    v.resize((size_t)n);
    if (mi355ndt_get_voxels(h_, 0, v.data(), v.size()) != MI355NDT_OK) v.clear();
    return v;
  }
  // GPU version of pcl::Registration::getFitnessScore (non-virtual in PCL: reached when the caller holds the derived type,
  // e.g. boost::dynamic_pointer_cast<mi355ndt::NormalDistributionsTransform<PointT,PointT>>(registration) in
  // loop_detector.hpp:256; through a base pointer PCL's own CPU kd-tree version runs and gives the same number)
  inline double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double s = std::numeric_limits<double>::max();
    mi355ndt_get_fitness_score(h_, max_range, &s, nullptr);
    return s;
  }

 protected:
  void push() {
    prm_.trans_epsilon = transformation_epsilon_;               // pcl::Registration::setTransformationEpsilon
    prm_.max_iterations = max_iterations_;                      // pcl::Registration::setMaximumIterations
    mi355ndt_set_params(h_, &prm_);
  }
  virtual void computeTransformation(PointCloudSource& output) { computeTransformation(output, Eigen::Matrix4f::Identity()); }
  // ndt_omp.h:256-267 / ndt_omp_impl2.hpp:87-188.  PCL's align() has already copied the source into `output`.
  virtual void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) {
    push();
    if (!same_cloud(input_.get(), input_->points.size())) {         // (not handed over by setInputSource: see there)
      dev_src_ = nullptr; dev_src_n_ = 0;
      if (mi355ndt_set_source(h_, input_->points.data(), input_->points.size(), sizeof(PointSource)) == MI355NDT_OK) { dev_src_ = input_.get(); dev_src_n_ = input_->points.size(); }
    }
    mi355ndt_result r;
    nr_iterations_ = 0;
    converged_ = false;
    if (mi355ndt_align(h_, guess.data() /* Eigen::Matrix4f is column-major */, &r) != MI355NDT_OK) return;
    if (r.status == MI355NDT_WARN_TOLERANCE_ARITH) {            // MI355NDT_OPT_ARITH = 1 on a registration it is not meant for (few hits, no convergence):
      mi355ndt_set_option(h_, MI355NDT_OPT_ARITH, 0);           // once more in the default arithmetic -- the grid on the device serves both
      const int rc = mi355ndt_align(h_, guess.data(), &r);
      mi355ndt_set_option(h_, MI355NDT_OPT_ARITH, 1);
      if (rc != MI355NDT_OK) return;
    }
    final_transformation_ = Eigen::Map<const Eigen::Matrix4f>(r.final_colmajor);
    nr_iterations_ = r.iterations;
    converged_ = r.converged != 0;
    trans_probability_ = r.trans_probability;
    // transformation_ / previous_transformation_ as the reference's loop leaves them (ndt_omp_impl2.hpp:134, 163):
    // pcl::Registration::getLastIncrementalTransformation() reads transformation_
    float inc[16], prev[16];
    if (mi355ndt_get_incremental(h_, 0, inc, prev) == MI355NDT_OK) {
      transformation_ = Eigen::Map<const Eigen::Matrix4f>(inc);
      previous_transformation_ = Eigen::Map<const Eigen::Matrix4f>(prev);
    }
    output.points.resize(input_->points.size());
    mi355ndt_get_aligned(h_, output.points.data(), sizeof(PointSource));   // x,y,z of the moved source; other fields kept
    // impl2:172-173 calls the visualizer hook once per iteration with the cloud at that iteration's pose; the Newton loop runs
    // on the device without host round trips, so the hook (never set anywhere in lv_slam) fires once, with the final cloud
    if (update_visualizer_ != 0) update_visualizer_(output, std::vector<int>(), *target_, std::vector<int>());
  }

  bool same_cloud(const void* cloud, size_t n) const { return cloud != nullptr && cloud == dev_src_ && n == dev_src_n_; }

  mi355ndt_handle* h_;
  mi355ndt_params prm_;
  double trans_probability_;
  const void* dev_src_;          // the cloud object (and its size) the device holds as the source
  size_t dev_src_n_;
};

}  // namespace mi355ndt
