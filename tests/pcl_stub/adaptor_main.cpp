// TEST INFRASTRUCTURE: drives include/mi355_ndt_pcl.hpp the way lv_slam's odometry nodelet drives its registration object
// (scan_matching_odom_nodelet.cpp:109-119, 197, 220-226), against tests/pcl_stub's stand-in for pcl::Registration.
//   adaptor_main <target.f32> <source.f32> <n_target> <n_source> <variant> <mode> <resolution> [latency_mode]
// clouds are raw float32 x,y,z triples; prints one line of numbers (final 16, last-increment 16, previous 16, iterations,
// converged, trans_probability, visualizer calls, fitness score) and the first 4 output points.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "mi355_ndt_pcl.hpp"

typedef pcl::PointXYZI PointT;

static pcl::PointCloud<PointT>::Ptr load(const char* path, size_t n) {
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  c->points.resize(n);
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  for (size_t i = 0; i < n; i++) {
    float v[3];
    if (fread(v, sizeof(float), 3, f) != 3) { fprintf(stderr, "short read\n"); exit(2); }
    c->points[i].x = v[0]; c->points[i].y = v[1]; c->points[i].z = v[2]; c->points[i].data[3] = 1.f; c->points[i].intensity = (float)i;
  }
  fclose(f);
  c->width = (unsigned)n;
  return c;
}

static int g_vis_calls = 0;

// adaptor_main seq <frames.f32> <n_frames> <n_points> : the per-frame call sequence of matching_s2k (scan_matching_odom_nodelet.cpp:192-261) on n_frames
// clouds of n_points points -- frame 0 becomes the keyframe (setInputTarget), every later frame is setInputSource + align (frame 1 twice, :223-227),
// every third frame is promoted to keyframe afterwards (`key = filtered; setInputTarget(key)`, :240-243) -- then prints the engine's cloud counters
// (uploads, promotions) and every frame's final pose: ONE trip over PCIe per frame is the drop-in's budget.
static int run_sequence(int argc, char** argv) {
  if (argc < 5) return 2;
  const size_t nf = strtoul(argv[3], 0, 10), np = strtoul(argv[4], 0, 10);
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 2; }
  std::vector<pcl::PointCloud<PointT>::Ptr> frames;
  for (size_t k = 0; k < nf; k++) {
    pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
    c->points.resize(np);
    for (size_t i = 0; i < np; i++) {
      float v[3];
      if (fread(v, sizeof(float), 3, f) != 3) { fprintf(stderr, "short read\n"); return 2; }
      c->points[i].x = v[0]; c->points[i].y = v[1]; c->points[i].z = v[2]; c->points[i].data[3] = 1.f; c->points[i].intensity = 0.f;
    }
    c->width = (unsigned)np;
    frames.push_back(c);
  }
  fclose(f);
  mi355ndt::NormalDistributionsTransform<PointT, PointT> reg(MI355NDT_VARIANT_PCA);
  reg.setNumThreads(4);
  reg.setTransformationEpsilon(0.01);
  reg.setMaximumIterations(64);
  reg.setResolution(1.0f);
  reg.setNeighborhoodSearchMethod(mi355ndt::DIRECT1);
  pcl::PointCloud<PointT>::ConstPtr key = frames[0];
  reg.setInputTarget(key);
  Eigen::Matrix4f guess = Eigen::Matrix4f::Identity();
  guess(0, 3) = 1.0f;
  std::vector<Eigen::Matrix4f> poses;
  for (size_t k = 1; k < nf; k++) {
    pcl::PointCloud<PointT> out;
    reg.setInputSource(frames[k]);
    reg.align(out, guess);
    if (k == 1) reg.align(out, reg.getFinalTransformation());      // :223-227
    poses.push_back(reg.getFinalTransformation());
    if (k % 3 == 0) { key = frames[k]; reg.setInputTarget(key); }  // :240-243
  }
  mi355ndt_profile P;
  mi355ndt_profile_get(reg.handle(), &P);
  printf("%lld %lld %lld\n", P.cloud_uploads, P.cloud_promotions, P.cloud_upload_bytes);
  for (size_t k = 0; k < poses.size(); k++) { for (int i = 0; i < 16; i++) printf("%.9g ", poses[k].data()[i]); printf("\n"); }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "seq")) return run_sequence(argc, argv);
  if (argc < 8) return 2;
  const size_t nt = strtoul(argv[3], 0, 10), ns = strtoul(argv[4], 0, 10);
  pcl::PointCloud<PointT>::Ptr tgt = load(argv[1], nt), src = load(argv[2], ns);
  mi355ndt::NormalDistributionsTransform<PointT, PointT> reg(atoi(argv[5]));
  // the nodelet's set-up sequence (scan_matching_odom_nodelet.cpp:109-119)
  reg.setNumThreads(4);
  reg.setTransformationEpsilon(0.01);
  reg.setMaximumIterations(64);
  reg.setResolution((float)atof(argv[7]));
  reg.setNeighborhoodSearchMethod((mi355ndt::NeighborSearchMethod)atoi(argv[6]));
  if (argc > 8 && atoi(argv[8])) reg.setLatencyMode(true);      // the engine's opt-in fine-grained sweep (not in the reference)
  std::function<void(const pcl::PointCloud<PointT>&, const std::vector<int>&, const pcl::PointCloud<PointT>&, const std::vector<int>&)> cb =
      [](const pcl::PointCloud<PointT>&, const std::vector<int>&, const pcl::PointCloud<PointT>&, const std::vector<int>&) { g_vis_calls++; };
  reg.registerVisualizationCallback(cb);
  reg.setInputTarget(tgt);
  reg.setInputSource(src);
  Eigen::Matrix4f guess = Eigen::Matrix4f::Identity();
  guess(0, 3) = 1.0f;
  pcl::PointCloud<PointT> out;
  reg.align(out, guess);
  const Eigen::Matrix4f F = reg.getFinalTransformation(), L = reg.getLastIncrementalTransformation();
  for (int i = 0; i < 16; i++) printf("%.9g ", F.data()[i]);
  for (int i = 0; i < 16; i++) printf("%.9g ", L.data()[i]);
  printf("%d %d %.17g %d %.17g\n", reg.getFinalNumIteration(), (int)reg.hasConverged(), reg.getTransformationProbability(), g_vis_calls,
         reg.getFitnessScore(4.0));
  for (int i = 0; i < 4 && i < (int)out.points.size(); i++)
    printf("%.9g %.9g %.9g %.9g %.9g\n", out.points[i].x, out.points[i].y, out.points[i].z, out.points[i].data[3], out.points[i].intensity);
  printf("%zu\n", reg.getTargetCells().size());
  // calculateScore of the output cloud (ndt_omp.h:232) and the static convertTransform (ndt_omp.h:209-228)
  Eigen::Matrix<double, 6, 1> x6;
  const double xv[6] = {1.0, 2.0, 3.0, 0.1, 0.2, 0.3};
  for (int i = 0; i < 6; i++) x6(i) = xv[i];
  Eigen::Matrix4f M;
  Eigen::Affine3f A;
  mi355ndt::NormalDistributionsTransform<PointT, PointT>::convertTransform(x6, M);
  mi355ndt::NormalDistributionsTransform<PointT, PointT>::convertTransform(x6, A);
  printf("%.17g", reg.calculateScore(out));
  for (int i = 0; i < 16; i++) printf(" %.9g", M.data()[i]);
  printf(" %d\n", (int)(std::memcmp(M.data(), A.matrix().data(), 64) == 0));
  return 0;
}
