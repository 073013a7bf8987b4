// Micro-benchmark: how long does a freshly launched wave wait for its first kernel-argument load?
// The kernel reads the wall clock (s_memrealtime) before anything else, then touches an argument, then reads the clock
// again.  A chain of launches with different argument blocks keeps the scalar caches cold, as in a replayed graph of
// ~450 different kernels.    hipcc --offload-arch=gfx950 -O3 -o kernarg_latency kernarg_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Big { long long* out; int pad[200]; int slot; };   // ~800-byte argument block, like GemmPair

__global__ __launch_bounds__(512) void k(const Big b) {
  long long t0, t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const int slot = b.slot;                                 // first argument use
  long long* out = b.out;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "s"(slot), "s"(out) : "memory");
  if (threadIdx.x == 0) {
    out[(slot * 4096 + blockIdx.x) * 2] = t0;
    out[(slot * 4096 + blockIdx.x) * 2 + 1] = t1;
  }
}

int main() {
  const int n = 64, grid = 144;
  long long* d;
  CK(hipMalloc(&d, (size_t)n * 4096 * 16));
  CK(hipMemset(d, 0, (size_t)n * 4096 * 16));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) {
    Big b{};
    b.out = d;
    b.slot = i;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, st, b);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  static long long h[64 * 4096 * 2];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  double lat[64], gap[64];
  for (int i = 0; i < n; ++i) {
    std::vector<double> v;
    long long first = h[(i * 4096) * 2], lastend = 0;
    for (int w = 0; w < grid; ++w) {
      v.push_back((h[(i * 4096 + w) * 2 + 1] - h[(i * 4096 + w) * 2]) / 100.0);
      first = std::min(first, h[(i * 4096 + w) * 2]);
    }
    std::sort(v.begin(), v.end());
    lat[i] = v[v.size() / 2];
    if (i > 0) {
      for (int w = 0; w < grid; ++w) lastend = std::max(lastend, h[((i - 1) * 4096 + w) * 2 + 1]);
      gap[i] = (first - lastend) / 100.0;
    }
  }
  std::sort(lat + 8, lat + n);
  std::sort(gap + 8, gap + n);
  printf("first kernel-argument load after wave start: median %.2f us (min %.2f, max %.2f) over %d kernels of %d workgroups\n",
         lat[8 + (n - 8) / 2], lat[8], lat[n - 1], n - 8, grid);
  printf("previous kernel's last argument-ready stamp -> this kernel's first wave start: median %.2f us\n", gap[8 + (n - 8) / 2]);
  return 0;
}
