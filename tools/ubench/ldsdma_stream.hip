// Micro-benchmark: how fast can ONE compute unit stream bytes into LDS with global_load_lds (the
// GEMM main loop's loader), as a function of (a) how many CUs stream at once, (b) where the bytes
// come from (HBM-cold vs L2-warm), (c) ring depth, (d) access shape (128-B row segments with a large
// row pitch, as a [N,K] weight tile is read, vs contiguous 1 KiB pieces).
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_stream ldsdma_stream.hip && ./ldsdma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Each workgroup (NW waves) streams `nslice` slices of SLICE bytes; wave w issues PIECES = SLICE/1024/NW
// wave-instructions per slice.  mode 0: piece p covers 8 rows x 128 B, row pitch `pitch` bytes, the
// slice advances 128 B along the row (a [rows, K] operand read K-slice by K-slice);
// mode 1: pieces are contiguous 1 KiB, slices contiguous.
template <int NW, int SLICE, int NS>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const unsigned char* base, long wg_stride, long pitch, int nslice,
                                                         int mode, long wrap, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PIECES = SLICE / 1024 / NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* wg = base + ((long)blockIdx.x * wg_stride) % wrap;
  const unsigned char* src[PIECES];
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
    const int piece = wave * PIECES + p;
    if (mode == 0) src[p] = wg + ((long)piece * 8 + (lane >> 3)) * pitch + (lane & 7) * 16;
    else src[p] = wg + (long)piece * 1024 + lane * 16;
  }
  const long adv = mode == 0 ? 128 : SLICE;
  int issued = 0;
  auto issue = [&](int stage) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) glds16(src[p] + (long)issued * adv, lds + stage * SLICE + (wave * PIECES + p) * 1024);
    ++issued;
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  int stage = 0;
  unsigned acc = 0;
  for (int kt = 0; kt < nslice; ++kt) {
    if (kt + NS - 2 < nslice) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < nslice) issue(stage == 0 ? NS - 1 : stage - 1);
    acc += *(const unsigned*)(lds + stage * SLICE + threadIdx.x * 4);   // one token LDS read per slice
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int SLICE, int NS>
void run(const char* label, const unsigned char* buf, long bytes, int grid, long wg_stride, long pitch, int nslice, int mode,
         long wrap, unsigned* sink) {
  auto k = stream_kernel<NW, SLICE, NS>;
  const int lds = SLICE * NS;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, buf, wg_stride, pitch, nslice, mode, wrap, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double tot = (double)grid * nslice * SLICE;
  printf("%-34s NW=%d slice=%2dK NS=%d grid=%4d : %8.1f us  %7.1f GB/s/WG  %6.2f TB/s total\n", label, NW, SLICE / 1024, NS, grid,
         best * 1e3, tot / grid / (best * 1e-3) / 1e9, tot / (best * 1e-3) / 1e12);
}

int main() {
  const long bytes = 4L << 30;
  unsigned char* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, bytes));
  const int nslice = 512;
  for (int grid : {32, 64, 128, 256, 512}) {
    // HBM-cold: every workgroup walks its own 8 MiB-aligned region of a 4 GiB buffer (rows of 64 KiB pitch)
    run<8, 32768, 4>("hbm rows(64K pitch)", buf, bytes, grid, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
    run<8, 32768, 4>("hbm contiguous", buf, bytes, grid, 32768L * nslice, 0, nslice, 1, bytes - (64L << 20), sink);
    // L2-warm: all workgroups walk the same 1.5 MiB panel (what the A operand of a skinny GEMM looks like)
    run<8, 32768, 4>("l2 rows, shared panel", buf, bytes, grid, 0, 65536, nslice / 8, 0, bytes, sink);
    run<8, 32768, 4>("l2 contiguous, shared panel", buf, bytes, grid, 0, 0, nslice / 8, 1, bytes, sink);
  }
  printf("-- depth / wave-count sweep at grid 256, hbm rows\n");
  run<8, 32768, 2>("hbm rows", buf, bytes, 256, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<8, 32768, 3>("hbm rows", buf, bytes, 256, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<8, 16384, 8>("hbm rows", buf, bytes, 256, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<4, 32768, 4>("hbm rows", buf, bytes, 256, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<4, 16384, 4>("hbm rows", buf, bytes, 256, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<8, 32768, 4>("hbm rows", buf, bytes, 48, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<8, 16384, 8>("hbm rows", buf, bytes, 48, 8L << 20, 65536, nslice, 0, bytes - (64L << 20), sink);
  run<8, 32768, 4>("l2 rows shared", buf, bytes, 48, 0, 65536, nslice / 8, 0, bytes, sink);
  run<8, 16384, 8>("l2 rows shared", buf, bytes, 48, 0, 65536, nslice / 8, 0, bytes, sink);
  return 0;
}
