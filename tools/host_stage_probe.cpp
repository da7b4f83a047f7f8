// GPU box, host side only: what bounds the staging of host pcl::PointXYZI clouds (32-byte records -> packed x,y,z)?  N threads compact a 1.1 GB batch
// (542 clouds of 65,536 records) into (a) malloc'd memory, (b) hipHostMalloc'd memory, and (c) only read it.   usage: host_stage_probe [threads...]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main(int argc, char** argv) {
  const size_t NC = 542, NP = 65536, REC = 32;
  unsigned char* src = (unsigned char*)malloc(NC * NP * REC);
  memset(src, 1, NC * NP * REC);
  float* dst_m = (float*)malloc(NC * NP * 12);
  memset(dst_m, 0, NC * NP * 12);
  float* dst_p = nullptr;
  if (hipHostMalloc((void**)&dst_p, NC * NP * 12) != hipSuccess) { printf("no pinned memory\n"); dst_p = nullptr; } else memset(dst_p, 0, NC * NP * 12);
  std::vector<int> ths;
  for (int i = 1; i < argc; i++) ths.push_back(atoi(argv[i]));
  if (ths.empty()) ths = {1, 4, 8, 12, 16};
  for (int nt : ths) {
    for (int mode = 0; mode < 3; mode++) {
      if (mode == 1 && !dst_p) continue;
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        std::atomic<size_t> next{0};
        std::atomic<unsigned long long> sink{0};
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&]() {
          unsigned long long acc = 0;
          for (size_t c = next.fetch_add(1); c < NC; c = next.fetch_add(1)) {
            const unsigned char* p = src + c * NP * REC;
            if (mode == 2) { for (size_t i = 0; i < NP; i++) { unsigned long long v; memcpy(&v, p + i * REC, 8); acc += v; } continue; }
            float* d = (mode == 0 ? dst_m : dst_p) + c * NP * 3;
            for (size_t i = 0; i < NP; i++) memcpy(d + 3 * i, p + i * REC, 12);
          }
          sink += acc;
        });
        for (auto& x : th) x.join();
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      }
      printf("threads %2d  %-28s %6.2f ms per batch  %6.1f GB/s of records read\n", nt, mode == 0 ? "compact -> malloc" : mode == 1 ? "compact -> hipHostMalloc" : "read only", 1e3 * best, NC * NP * REC / best / 1e9);
    }
  }
  return 0;
}
