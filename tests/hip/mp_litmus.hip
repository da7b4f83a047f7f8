// mp_litmus.hip -- message-passing litmus test of the one-launch align's hand-over protocol (lv_slam_amd/csrc/ndt_async.hpp).
//
// Inside ONE kernel launch workgroups on different XCDs pass data to each other through memory with exactly these primitives, and no
// cache-wide fence:
//   writer:  N agent-scope relaxed stores (sc1: write-through)  ->  s_waitcnt vmcnt(0)  ->  one agent-scope relaxed store or returning
//            fetch-add on the word that announces them
//   reader:  agent-scope relaxed loads (sc1: L1-bypassing) of the announcing word until it shows the expected value, then agent-scope
//            relaxed loads of the N data words
// (partial rows -> arrival counter -> updater; pair state -> ticket word -> sweeping waves).  That the data is visible once the
// announcement is rests on how gfx950 orders a wave's write-through stores behind s_waitcnt vmcnt(0) -- observed, soaked
// (profiles/r04_soak_async.txt), but not a documented guarantee of the HIP memory model for relaxed atomics.  This program turns the
// assumption into a test: 512 single-wave workgroups, 256 writer -> reader pairs placed three XCDs apart (workgroup L runs on XCD L % 8),
// >= 10^7 hand-overs, every data word checked.  A ROCm / firmware change that breaks the assumption turns `pytest -m gpu` red.
//   usage: mp_litmus [rounds per pair, default 40000]      prints: handovers=<n> errors=<n> timeouts=<n> us_per_handover=<t>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define NDATA 44            // one partial row
#define PAIRS 256
#define SLOT_WORDS 64       // 8-byte words per pair: NDATA data words, then padding (512-byte slots: pairs do not share lines)

struct Ctl { unsigned flag, pad0[31]; unsigned ack, pad1[31]; };   // announcing word and acknowledgement on lines of their own

__global__ void __launch_bounds__(64) k_litmus(unsigned long long* data, Ctl* ctl, int rounds, unsigned long long* errors, unsigned long long* timeouts) {
  const int lane = threadIdx.x;
  const bool writer = blockIdx.x < PAIRS;
  const int p = writer ? (int)blockIdx.x : (int)((blockIdx.x - PAIRS + PAIRS - 3) % PAIRS);   // reader of pair p is workgroup PAIRS + (p + 3) % PAIRS: three XCDs on
  gu64* D = (gu64*)(data + (size_t)p * SLOT_WORDS);
  gu32* flag = (gu32*)&ctl[p].flag;
  gu32* ack = (gu32*)&ctl[p].ack;
  unsigned long long bad = 0, late = 0;
  for (int r = 1; r <= rounds; r++) {
    if (writer) {
      // wait until the reader is done with round r - 1 (its acknowledgement), then write round r
      if (lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ack, RLX_AGENT) != (unsigned)(r - 1)) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 24)) { late++; break; } }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < NDATA) __hip_atomic_store(D + lane, ((unsigned long long)r << 32) | (unsigned)(lane * 2654435761u + (unsigned)r), RLX_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        if (r & 1) __hip_atomic_store(flag, (unsigned)r, RLX_AGENT);              // a ticket word / a tag
        else (void)__hip_atomic_fetch_add(flag, 1u, RLX_AGENT);                   // an arrival counter
      }
    } else {
      unsigned seen = 0;
      if (lane == 0) {
        unsigned spins = 0;
        while ((seen = __hip_atomic_load(flag, RLX_AGENT)) != (unsigned)r) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 24)) { late++; break; } }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < NDATA) {
        const unsigned long long v = __hip_atomic_load(D + lane, RLX_AGENT);
        if (v != (((unsigned long long)r << 32) | (unsigned)(lane * 2654435761u + (unsigned)r))) bad++;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the loads have returned before the writer may overwrite
      if (lane == 0) __hip_atomic_store(ack, (unsigned)r, RLX_AGENT);
    }
    if (late) break;
  }
  if (bad) atomicAdd(errors, bad);
  if (late) atomicAdd(timeouts, late);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40000;
  unsigned long long *data, *cnt;
  Ctl* ctl;
  if (hipMalloc((void**)&data, (size_t)PAIRS * SLOT_WORDS * 8) != hipSuccess || hipMalloc((void**)&ctl, PAIRS * sizeof(Ctl)) != hipSuccess ||
      hipMalloc((void**)&cnt, 16) != hipSuccess) { fprintf(stderr, "no device memory\n"); return 2; }
  hipMemset(data, 0, (size_t)PAIRS * SLOT_WORDS * 8);
  hipMemset(ctl, 0, PAIRS * sizeof(Ctl));
  hipMemset(cnt, 0, 16);
  int nblk = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_litmus, 64, 0);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  if ((long long)nblk * pr.multiProcessorCount < 2 * PAIRS) { fprintf(stderr, "device too small for %d co-resident waves\n", 2 * PAIRS); return 2; }
  const auto t0 = std::chrono::steady_clock::now();
  k_litmus<<<2 * PAIRS, 64>>>(data, ctl, rounds, cnt, cnt + 1);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long h[2];
  hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
  printf("handovers=%lld errors=%llu timeouts=%llu us_per_handover=%.3f\n", (long long)PAIRS * rounds, h[0], h[1], us / rounds);
  return (h[0] || h[1]) ? 1 : 0;
}
