// dump_golden.cpp -- the reference side of the parity pin (tools/pin_reference/README.md).
//
// Built on a ROS / PCL host against lv_slam's OWN headers (include/ndt_omp/*.h, include/ndt_pca/*.h with PCL 1.8, Eigen 3.3, Sophus a621ff2
// and the flags of lv_slam's CMakeLists.txt:6,11), it runs the committed input clouds of tests/golden/pin/ through the real
// pclomp:: / pclpca::NormalDistributionsTransform and writes, per case, what the parity tests compare against:
//   every leaf of the voxel grid (cell index, nr_points, mean, cov, icov, evals, ndt_pca's integer weight) in std::map order,
//   one computeDerivatives sweep at the guess (score, gradient, Hessian),
//   align(): getFinalTransformation(), getFinalNumIteration(), hasConverged(), getTransformationProbability(),
//   getLastIncrementalTransformation(), and calculateScore(output cloud)
// as tests/golden/ref_<case>.bin (layout: tests/golden/ref_format.py, "NDTREF01").  tests/test_reference_golden.py consumes those files
// when they are present -- the oracle on the CPU, the HIP path under -m gpu -- and that is what turns "parity unpinned" into "pinned".
//
// Nothing of this file runs in the product or in the default test-suite.  Without PCL (this repository's build box) it can still be
// compiled with -DPIN_SELFCHECK_MI355: the mi355ndt adaptor over tests/pcl_stub then stands in for the reference classes, which checks the
// program's own logic and file format (tests/test_reference_golden.py::test_dumper_selfcheck_*), not the reference.
//
// usage: dump_golden <tests/golden/pin> <output directory> [case name ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>
#include <iostream>

#ifdef PIN_SELFCHECK_MI355
#include "mi355_ndt_pcl.hpp"
typedef pcl::PointXYZI PointT;
#else
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <pcl/common/transforms.h>
#include <ndt_omp/ndt_omp.h>
#include <ndt_omp/ndt_omp_impl2.hpp>
#include <ndt_omp/voxel_grid_covariance_omp_impl.hpp>
#include <ndt_pca/ndt_pca.h>
#include <ndt_pca/ndt_pca_impl2.hpp>
#include <ndt_pca/voxel_grid_covariance_pca_impl.hpp>
typedef pcl::PointXYZI PointT;
#endif

struct Case {
  std::string name, target, source;
  int variant, mode;                 // 0 = pclomp, 1 = pclpca; NeighborSearchMethod enum value (ndt_omp.h:51-56: KDTREE, DIRECT26, DIRECT7, DIRECT1)
  float resolution;
  double step_size, outlier_ratio, trans_epsilon;
  int max_iterations;
  float guess[16];                   // column-major
};

struct LeafOut { int64_t idx; int32_t n, weight; double mean[3], cov[9], icov[9], evals[3]; };
struct Dump {
  std::vector<LeafOut> leaves;
  uint32_t flags = 0;                // 1 = sweep present, 2 = ALL leaves of the map (not only the searchable ones), 4 = cov / evals present
  double p[6], score, g[6], H[36];
  float final_cm[16], last_inc_cm[16];
  int32_t iterations = 0, converged = 0;
  double trans_probability = 0, calc_score = 0;
};

static pcl::PointCloud<PointT>::Ptr load_cloud(const std::string& path, size_t* n_out) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) { std::cerr << "cannot read " << path << "\n"; exit(2); }
  f.seekg(0, std::ios::end);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0);
  const size_t n = bytes / 12;
  std::vector<float> v(3 * n);
  f.read((char*)v.data(), (std::streamsize)(12 * n));
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  c->points.resize(n);
  for (size_t i = 0; i < n; i++) {
    PointT q;
    std::memset(&q, 0, sizeof q);
    q.x = v[3 * i]; q.y = v[3 * i + 1]; q.z = v[3 * i + 2]; q.data[3] = 1.f;
    c->points[i] = q;
  }
  c->width = (uint32_t)n; c->height = 1; c->is_dense = true;     // the committed clouds hold finite points only
  *n_out = n;
  return c;
}

#ifndef PIN_SELFCHECK_MI355
// The reference classes keep the voxel grid and computeDerivatives protected (ndt_omp.h:290-299, 499): a subclass may look.
template <typename Base>
struct Expose : public Base {
  typedef typename Base::TargetGrid Grid;
  void leaves(Dump& d, bool pca) {
    const auto& map = this->target_cells_.getLeaves();            // voxel_grid_covariance_omp.h:414-418: every occupied cell, std::map order
    for (auto it = map.begin(); it != map.end(); ++it) {
      const auto& L = it->second;
      LeafOut o;
      std::memset(&o, 0, sizeof o);
      o.idx = (int64_t)it->first;
      o.n = L.nr_points;
      o.weight = pca ? weight_of(L, 0) : 0;
      const Eigen::Vector3d m = L.getMean();
      const Eigen::Matrix3d c = L.getCov(), ic = L.getInverseCov();
      const Eigen::Vector3d ev = L.getEvals();
      for (int r = 0; r < 3; r++) {
        o.mean[r] = m(r); o.evals[r] = ev(r);
        for (int k = 0; k < 3; k++) { o.cov[r * 3 + k] = c(r, k); o.icov[r * 3 + k] = ic(r, k); }
      }
      d.leaves.push_back(o);
    }
    d.flags |= 2u | 4u;
  }
  // ndt_pca's per-leaf weight as the sweep uses it: getDimension2d() returns int (voxel_grid_covariance_pca.h:222-226); pclomp's Leaf has none
  template <typename LeafT> static auto weight_of(const LeafT& L, int) -> decltype((int32_t)L.getDimension2d()) { return (int32_t)L.getDimension2d(); }
  template <typename LeafT> static int32_t weight_of(const LeafT&, long) { return 0; }
  // the first sweep of computeTransformation (ndt_omp_impl2.hpp:102-129): cloud moved by the guess, p = log(guess)
  void sweep_at_guess(Dump& d, const Eigen::Matrix4f& guess) {
    typename Base::PointCloudSource cloud = *this->input_;
    for (size_t i = 0; i < cloud.points.size(); i++) cloud.points[i].data[3] = 1.f;
    if (guess != Eigen::Matrix4f::Identity()) pcl::transformPointCloud(cloud, cloud, guess);
    Sophus::SE3 SE3_Rt(guess.block(0, 0, 3, 3).cast<double>(), guess.block(0, 3, 3, 1).cast<double>());
    Eigen::Matrix<double, 6, 1> p = SE3_Rt.log(), g;
    Eigen::Matrix<double, 6, 6> H;
    d.score = this->computeDerivatives(g, H, cloud, p, true);     // (gauss_d1_/d2_/d3_ were set by the align() that ran before)
    for (int i = 0; i < 6; i++) { d.p[i] = p(i); d.g[i] = g(i); for (int j = 0; j < 6; j++) d.H[i * 6 + j] = H(i, j); }
    d.flags |= 1u;
  }
};
typedef Expose<pclomp::NormalDistributionsTransform<PointT, PointT> > RegOmp;
typedef Expose<pclpca::NormalDistributionsTransform<PointT, PointT> > RegPca;
template <typename Reg> static void set_mode(Reg& reg, int mode);
template <> void set_mode<RegOmp>(RegOmp& reg, int mode) { reg.setNeighborhoodSearchMethod((pclomp::NeighborSearchMethod)mode); }
template <> void set_mode<RegPca>(RegPca& reg, int mode) { reg.setNeighborhoodSearchMethod((pclpca::NeighborSearchMethod)mode); }
#else
// self-check build: the adaptor stands in (searchable leaves only, no cov / evals, no protected sweep to call)
struct RegSelf : public mi355ndt::NormalDistributionsTransform<PointT, PointT> {
  explicit RegSelf(int variant) : mi355ndt::NormalDistributionsTransform<PointT, PointT>(variant) {}
  void leaves(Dump& d, bool) {
    const std::vector<mi355ndt_voxel> v = getTargetCells();
    for (size_t i = 0; i < v.size(); i++) {
      LeafOut o;
      std::memset(&o, 0, sizeof o);
      o.idx = v[i].idx; o.n = v[i].n; o.weight = v[i].weight;
      for (int a = 0; a < 3; a++) { o.mean[a] = v[i].mean[a]; o.evals[a] = std::nan(""); }
      for (int a = 0; a < 9; a++) { o.icov[a] = (double)v[i].icov[a]; o.cov[a] = std::nan(""); }
      d.leaves.push_back(o);
    }
  }
  void sweep_at_guess(Dump&, const Eigen::Matrix4f&) {}
};
template <typename Reg> static void set_mode(Reg& reg, int mode) { reg.setNeighborhoodSearchMethod((mi355ndt::NeighborSearchMethod)mode); }
#endif

template <typename Reg>
static void run(Reg& reg, const Case& c, pcl::PointCloud<PointT>::Ptr tgt, pcl::PointCloud<PointT>::Ptr src, Dump& d) {
  reg.setTransformationEpsilon(c.trans_epsilon);
  reg.setMaximumIterations(c.max_iterations);
  reg.setStepSize(c.step_size);
  reg.setOulierRatio(c.outlier_ratio);
  reg.setResolution(c.resolution);
  set_mode(reg, c.mode);
  reg.setInputTarget(tgt);
  reg.setInputSource(src);
  Eigen::Matrix4f guess;
  for (int i = 0; i < 16; i++) guess.data()[i] = c.guess[i];       // Eigen::Matrix4f is column-major
  pcl::PointCloud<PointT> out;
  reg.align(out, guess);
  const Eigen::Matrix4f F = reg.getFinalTransformation(), L = reg.getLastIncrementalTransformation();
  for (int i = 0; i < 16; i++) { d.final_cm[i] = F.data()[i]; d.last_inc_cm[i] = L.data()[i]; }
  d.iterations = reg.getFinalNumIteration();
  d.converged = reg.hasConverged() ? 1 : 0;
  d.trans_probability = reg.getTransformationProbability();
  d.calc_score = reg.calculateScore(out);
  reg.leaves(d, c.variant == 1);
  reg.sweep_at_guess(d, guess);
}

static void put(std::ofstream& f, const void* p, size_t n) { f.write((const char*)p, (std::streamsize)n); }

static void write_dump(const std::string& path, const Case& c, size_t nt, size_t ns, const Dump& d) {
  std::ofstream f(path.c_str(), std::ios::binary);
  if (!f) { std::cerr << "cannot write " << path << "\n"; exit(2); }
  const char magic[8] = {'N', 'D', 'T', 'R', 'E', 'F', '0', '1'};
  put(f, magic, 8);
  const int32_t hdr[6] = {c.variant, c.mode, (int32_t)nt, (int32_t)ns, c.max_iterations, (int32_t)d.leaves.size()};
  put(f, hdr, sizeof hdr);
  const uint32_t flags = d.flags;
  put(f, &flags, 4);
  put(f, &c.resolution, 4);
  const double prm[3] = {c.step_size, c.outlier_ratio, c.trans_epsilon};
  put(f, prm, sizeof prm);
  for (size_t i = 0; i < d.leaves.size(); i++) {
    const LeafOut& o = d.leaves[i];
    put(f, &o.idx, 8); put(f, &o.n, 4); put(f, &o.weight, 4);
    put(f, o.mean, 24); put(f, o.cov, 72); put(f, o.icov, 72); put(f, o.evals, 24);
  }
  put(f, d.p, 48); put(f, &d.score, 8); put(f, d.g, 48); put(f, d.H, 288);
  put(f, d.final_cm, 64); put(f, d.last_inc_cm, 64);
  put(f, &d.iterations, 4); put(f, &d.converged, 4);
  put(f, &d.trans_probability, 8); put(f, &d.calc_score, 8);
}

static std::vector<Case> read_cases(const std::string& dir) {
  // cases.txt: one case per line --
  //   name variant mode resolution step_size outlier_ratio trans_epsilon max_iterations target.bin source.bin g0 .. g15 (column-major)
  std::ifstream f((dir + "/cases.txt").c_str());
  if (!f) { std::cerr << "cannot read " << dir << "/cases.txt\n"; exit(2); }
  std::vector<Case> out;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream s(line);
    Case c;
    s >> c.name >> c.variant >> c.mode >> c.resolution >> c.step_size >> c.outlier_ratio >> c.trans_epsilon >> c.max_iterations >> c.target >> c.source;
    for (int i = 0; i < 16; i++) s >> c.guess[i];
    if (!s) { std::cerr << "malformed line: " << line << "\n"; exit(2); }
    out.push_back(c);
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::cerr << "usage: dump_golden <tests/golden/pin> <output directory> [case ...]\n"; return 2; }
  const std::string in = argv[1], outdir = argv[2];
  const std::vector<Case> cases = read_cases(in);
  int done = 0;
  for (size_t k = 0; k < cases.size(); k++) {
    const Case& c = cases[k];
    bool wanted = argc == 3;
    for (int a = 3; a < argc; a++) wanted = wanted || c.name == argv[a];
    if (!wanted) continue;
    size_t nt = 0, ns = 0;
    pcl::PointCloud<PointT>::Ptr tgt = load_cloud(in + "/" + c.target, &nt), src = load_cloud(in + "/" + c.source, &ns);
    Dump d;
    std::memset(d.p, 0, sizeof d.p); std::memset(d.g, 0, sizeof d.g); std::memset(d.H, 0, sizeof d.H); d.score = 0;
#ifdef PIN_SELFCHECK_MI355
    { RegSelf reg(c.variant); run(reg, c, tgt, src, d); }
#else
    if (c.variant == 0) { RegOmp reg; run(reg, c, tgt, src, d); }
    else { RegPca reg; run(reg, c, tgt, src, d); }
#endif
    write_dump(outdir + "/ref_" + c.name + ".bin", c, nt, ns, d);
    std::printf("%-24s leaves %6zu  iterations %3d  converged %d  trans_probability %.12g  calculateScore %.12g\n", c.name.c_str(), d.leaves.size(),
                d.iterations, d.converged, d.trans_probability, d.calc_score);
    done++;
  }
  return done ? 0 : 1;
}
