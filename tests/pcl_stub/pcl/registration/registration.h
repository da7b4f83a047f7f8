// TEST INFRASTRUCTURE -- not PCL.  A minimal stand-in for the parts of <pcl/registration/registration.h> (PCL 1.8) and Eigen
// that include/mi355_ndt_pcl.hpp touches, so that the adaptor can be compiled and driven on a box without PCL/Eigen
// (tests/test_adaptor_*.py).  It restates the public pcl::Registration contract (member names, align() pre-/post-conditions,
// SURVEY.md 8b) and nothing else; the real headers replace it on the ROS/PCL host.
#pragma once
#include <cstddef>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace Eigen {
struct Matrix4f {                       // column-major 4x4 float, like Eigen::Matrix4f
  float m[16];
  static Matrix4f Identity() { Matrix4f r; for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.f : 0.f; return r; }
  const float* data() const { return m; }
  float* data() { return m; }
  float& operator()(int r, int c) { return m[c * 4 + r]; }
  float operator()(int r, int c) const { return m[c * 4 + r]; }
};
template <typename S, int R, int C> struct Matrix {   // just enough of a fixed-size dense matrix: contiguous column-major storage
  S m[R * C];
  const S* data() const { return m; }
  S* data() { return m; }
  S& operator()(int i) { return m[i]; }
  S operator()(int i) const { return m[i]; }
};
struct Affine3f {                       // Eigen::Transform<float, 3, Affine>: a 4x4 matrix with matrix() access
  Matrix4f mat;
  Matrix4f& matrix() { return mat; }
  const Matrix4f& matrix() const { return mat; }
};
template <typename T> struct Map;
template <> struct Map<const Matrix4f> {
  const float* p;
  explicit Map(const float* q) : p(q) {}
  operator Matrix4f() const { Matrix4f r; std::memcpy(r.m, p, sizeof r.m); return r; }
};
}  // namespace Eigen

// PCL 1.8 hands clouds around as boost::shared_ptr (pcl::PointCloud<T>::Ptr / ::ConstPtr); the adaptor only ever goes through those typedefs
namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... A> shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost

namespace pcl {
struct PointXYZI {                      // 32-byte record, x,y,z first (pcl::PointXYZI's layout)
  union { float data[4]; struct { float x, y, z; }; };
  union { struct { float intensity; }; float data_c[4]; };
};
struct PointXYZ { union { float data[4]; struct { float x, y, z; }; }; };

template <typename PointT>
struct PointCloud {
  typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > ConstPtr;
  std::vector<PointT> points;
  unsigned width = 0, height = 1;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
};

template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  typedef Eigen::Matrix4f Matrix4;
  typedef pcl::PointCloud<PointSource> PointCloudSource;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef pcl::PointCloud<PointTarget> PointCloudTarget;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  typedef std::function<void(const PointCloudSource&, const std::vector<int>&, const PointCloudTarget&, const std::vector<int>&)> UpdateVisualizerCallbackSignature;

  Registration()
      : nr_iterations_(0), max_iterations_(10), final_transformation_(Matrix4::Identity()), transformation_(Matrix4::Identity()),
        previous_transformation_(Matrix4::Identity()), transformation_epsilon_(0.0), converged_(false) {}
  virtual ~Registration() {}

  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  inline void setMaximumIterations(int n) { max_iterations_ = n; }
  inline void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  inline Matrix4 getFinalTransformation() { return final_transformation_; }
  inline Matrix4 getLastIncrementalTransformation() { return transformation_; }
  inline bool hasConverged() { return converged_; }
  template <typename F> bool registerVisualizationCallback(F& f) { update_visualizer_ = f; return true; }

  // PCL 1.8 Registration::align(output, guess): needs source + target; output = copy of the source; converged_ = false;
  // final = transformation = previous = Identity; data[3] = 1; then the subclass's computeTransformation
  inline void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  inline void align(PointCloudSource& output, const Matrix4& guess) {
    if (!input_ || !target_ || input_->points.empty() || target_->points.empty()) return;      // initCompute() failed
    output.points = input_->points;
    output.width = input_->width; output.height = input_->height; output.is_dense = input_->is_dense;
    converged_ = false;
    final_transformation_ = transformation_ = previous_transformation_ = Matrix4::Identity();
    for (std::size_t i = 0; i < output.points.size(); ++i) output.points[i].data[3] = 1.0f;
    computeTransformation(output, guess);
  }

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;

  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_, max_iterations_;
  Matrix4 final_transformation_, transformation_, previous_transformation_;
  double transformation_epsilon_;
  bool converged_;
  UpdateVisualizerCallbackSignature update_visualizer_;
};
}  // namespace pcl
