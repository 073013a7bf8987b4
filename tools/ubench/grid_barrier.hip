// Micro-benchmark: cost and correctness of an in-kernel grid barrier on MI355X (8 XCDs, one L2 each).
// Every workgroup (512 threads, 160 KiB of LDS -> exactly one per CU, as the GEMM kernels are) writes a
// private 4 KiB block, crosses the barrier, and reads the block of a workgroup that sits on ANOTHER XCD -
// once with plain loads and once with buffer_load ... lds (the GEMM loaders' path).  Three rounds per
// launch with changing patterns, so a stale line left in the reader's L2 by the previous round shows up.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Bar {
  unsigned* count;   // arrivals of the current generation
  unsigned* gen;     // generation number
};

// returns the number of polls (0 for the releasing workgroup); ~0u on timeout
__device__ __forceinline__ unsigned grid_barrier(const Bar b, unsigned nwg) {
  __syncthreads();                                       // every wave's stores are performed (workgroup release)
  unsigned polls = 0;
  if (threadIdx.x == 0) {
    const unsigned g0 = __hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // buffer_wbl2 sc1: this XCD's dirty lines reach memory
    const unsigned old = __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg - 1) {
      __hip_atomic_store(b.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(b.gen, g0 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
        if (++polls > 200000u) { polls = ~0u; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: drop possibly stale lines
  }
  __syncthreads();
  return polls;
}

__global__ __launch_bounds__(512) void k(unsigned* data, Bar bar, unsigned nwg, unsigned seed, int use_dma, long long* times, unsigned* errs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned wg = blockIdx.x, tid = threadIdx.x;
  const unsigned peer = (wg + 3 + 8 * 5) % nwg;          // blockIdx % 8 selects the XCD: a different one
  unsigned bad = 0;
  long long t_bar = 0;
  unsigned polls_max = 0;
  for (unsigned round = 0; round < 3; ++round) {
    // 4 KiB per workgroup: 1024 dwords, two per thread
    unsigned* mine = data + (size_t)wg * 1024;
    mine[tid] = seed + round * 7919u + wg * 1024u + tid;
    mine[tid + 512] = seed + round * 7919u + wg * 1024u + tid + 512;
    const long long t0 = wall_clock64();
    const unsigned p = grid_barrier(bar, nwg);
    t_bar += wall_clock64() - t0;
    if (p == ~0u) bad |= 0x80000000u;
    else if (p > polls_max) polls_max = p;
    const unsigned* theirs = data + (size_t)peer * 1024;
    if (!use_dma) {
      const unsigned a = theirs[tid], b = theirs[tid + 512];
      if (a != seed + round * 7919u + peer * 1024u + tid) ++bad;
      if (b != seed + round * 7919u + peer * 1024u + tid + 512) ++bad;
    } else {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)theirs, 0, 4096, 0x00020000);
      const int wave = tid >> 6, lane = tid & 63;
      if (wave < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, wave * 1024 + lane * 16, 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned* l = (const unsigned*)lds;
      if (l[tid] != seed + round * 7919u + peer * 1024u + tid) ++bad;
      if (l[tid + 512] != seed + round * 7919u + peer * 1024u + tid + 512) ++bad;
    }
    grid_barrier(bar, nwg);                              // nobody overwrites a block that is still being read
  }
  if (bad) atomicAdd(errs, bad & 0x7fffffffu ? 1u : 0u), atomicAdd(errs + 1, bad >> 31);
  if (tid == 0) { times[wg * 2] = t_bar; times[wg * 2 + 1] = polls_max; }
}

int main() {
  unsigned* data; unsigned* barw; unsigned* errs; long long* times;
  CK(hipMalloc(&data, 256 * 4096 * 2));
  CK(hipMalloc(&barw, 256));
  CK(hipMalloc(&errs, 8));
  CK(hipMalloc(&times, 8192));
  CK(hipMemset(barw, 0, 256));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("CUs %d\n", prop.multiProcessorCount);
  Bar bar{barw, barw + 32};
  long long h[512];
  for (int dma = 0; dma < 2; ++dma)
    for (unsigned nwg : {48u, 144u, 256u}) {
      if ((int)nwg > prop.multiProcessorCount) continue;
      double best = 1e9;
      unsigned e[2] = {0, 0};
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(errs, 0, 8));
        hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 160 * 1024, 0, data, bar, nwg, 1000003u * (rep + 1), dma, times, errs);
        CK(hipDeviceSynchronize());
        unsigned er[2];
        CK(hipMemcpy(er, errs, 8, hipMemcpyDeviceToHost));
        e[0] += er[0]; e[1] += er[1];
        CK(hipMemcpy(h, times, nwg * 16, hipMemcpyDeviceToHost));
        double s = 0; for (unsigned i = 0; i < nwg; ++i) s += h[2 * i];
        s = s / nwg / 3.0 / 100.0;   // us per (first) barrier of a round, mean over workgroups
        if (rep > 0 && s < best) best = s;
      }
      printf("%s nwg %3u: barrier %.2f us (mean over workgroups, best of 5 launches), mismatching workgroups %u, timeouts %u\n",
             dma ? "buffer_load_lds" : "plain loads    ", nwg, best, e[0], e[1]);
    }
  return 0;
}
