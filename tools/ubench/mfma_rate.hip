// Micro-benchmark: what does the matrix pipe deliver to the consumer structure of the bs=1 conv GEMM
// (gemm_ws_conv3_kernel<256,64,8,1>: 8 consumer waves = 2 per SIMD, 8 MFMA 32x32x16 per wave and tap slice, one
// s_barrier per slice) before any operand traffic?   72 slices per launch, register operands.
//   mode 0  back-to-back MFMAs, no barrier, 2 accumulators per wave (the product's FN = 2)
//   mode 1  the same, 4 accumulators
//   mode 2  one s_barrier per slice (8 consumer waves only)
//   mode 3  one s_barrier per slice, 4 idle loader waves also arrive at it
//   mode 4  mode 3 + early / late halves (late waves multiply the previous slice's operands first - no data dependence here)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void buf_lds16(const void* base, unsigned bytes, unsigned char* lds_wave_base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// MODE 5 / 6: the 4 loader waves stream an L2-resident panel into LDS (PPS pieces of 1 KiB per wave and "slice", 4 slices in
// flight), NOT coupled to the consumers (5) or meeting them at one s_barrier per slice (6)
template <int MODE, int NWAVES, int PPS = 0>
__global__ __launch_bounds__(NWAVES * 64) void k(int nslice, long long* cyc, float* sink, const unsigned char* panel = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long c0 = (long long)__builtin_readcyclecounter();
  if (wave >= 8) {
    if constexpr (MODE >= 5) {
      const int lw = wave - 8;
      int v[PPS > 0 ? PPS : 1];
#pragma unroll
      for (int i = 0; i < PPS; ++i) v[i] = ((lw * PPS + i) * 8 + (lane >> 3)) * 3072 + (lane & 7) * 16;
      auto issue = [&](int sl) {
#pragma unroll
        for (int i = 0; i < PPS; ++i) buf_lds16(panel, 1536 * 1024, lds + (sl & 3) * (4 * PPS * 1024) + (lw * PPS + i) * 1024, v[i], (sl % 24) * 128);
      };
      issue(0); issue(1); issue(2);
      for (int kt = 0; kt < nslice; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPS) : "memory");
        if (MODE == 6) __builtin_amdgcn_s_barrier();
        issue(kt + 3);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0 && lw == 0) cyc[256 + blockIdx.x] = (long long)__builtin_readcyclecounter() - c0;
      return;
    }
    for (int kt = 0; kt < nslice; ++kt) __builtin_amdgcn_s_barrier();
    return;
  }
  f32x16 acc[4] = {};
  bf16x8 a[4], b[2];
#pragma unroll
  for (int s = 0; s < 4; ++s) a[s][0] = (__bf16)(float)((lane + s) & 3);
  b[0][1] = (__bf16)1.0f;
  b[1][2] = (__bf16)2.0f;
  for (int kt = 0; kt < nslice; ++kt) {
    if (MODE >= 2 && MODE != 5) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = MODE == 1 ? (2 * (s & 1) + j) : j;
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[j], acc[q], 0, 0, 0);
      }
  }
  if (lane == 0 && wave == 0) cyc[blockIdx.x] = (long long)__builtin_readcyclecounter() - c0;
  if (acc[0][3] + acc[1][5] + acc[2][0] + acc[3][0] == 123.456f) sink[0] = acc[0][1];
}

template <int MODE, int NWAVES, int PPS = 0>
void run(const char* label, long long* cyc, float* sink, const unsigned char* panel = nullptr) {
  const int nslice = 72 * 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NWAVES, PPS>), dim3(256), dim3(NWAVES * 64), 16 * (PPS ? PPS : 1) * 1024, 0, nslice, cyc, sink, panel);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  long long h[512];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double cy = 0, cl = 0;
  for (int i = 0; i < 256; ++i) cy += (double)h[i], cl += (double)h[256 + i];
  cy /= 256;
  cl /= 256;
  if (PPS) printf("   loaders: %d pieces / slice / CU, %.0f cycles -> %.1f cycles / slice, %.1f B/clk/CU\n", 4 * PPS, cl, cl / nslice, 4.0 * PPS * 1024 * nslice / cl);
  printf("%-60s : %7.1f us | %8.0f cycles | %6.1f cycles / slice (512 = matrix pipe full) | %5.1f cycles / MFMA / SIMD | clock %.2f GHz\n", label,
         best * 1e3, cy, cy / nslice, cy / nslice / 16, cy / (best * 1e3) / 1e3);
}

int main() {
  long long* cyc;
  float* sink;
  CK(hipMalloc(&cyc, 512 * 8));
  CK(hipMemset(cyc, 0, 512 * 8));
  unsigned char* panel;
  CK(hipMalloc(&panel, 2 << 20));
  CK(hipMemset(panel, 0, 2 << 20));
  CK(hipMalloc(&sink, 4));
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 8>("0 back-to-back, 2 accumulators / wave", cyc, sink);
    run<1, 8>("1 back-to-back, 4 accumulators / wave", cyc, sink);
    run<2, 8>("2 + s_barrier per slice (8 waves)", cyc, sink);
    run<3, 12>("3 + s_barrier per slice, 4 idle loader waves arrive too", cyc, sink);
    run<1, 4>("1 back-to-back, 4 accumulators, ONE wave per SIMD", cyc, sink);
    run<5, 12, 1>("5 no barrier; loaders stream 4 KiB / slice, uncoupled", cyc, sink, panel);
    run<5, 12, 3>("5 no barrier; loaders stream 12 KiB / slice, uncoupled", cyc, sink, panel);
    run<5, 12, 5>("5 no barrier; loaders stream 20 KiB / slice, uncoupled", cyc, sink, panel);
    run<5, 12, 8>("5 no barrier; loaders stream 32 KiB / slice, uncoupled", cyc, sink, panel);
    run<6, 12, 1>("6 barrier per slice; loaders stream 4 KiB / slice", cyc, sink, panel);
    run<6, 12, 3>("6 barrier per slice; loaders stream 12 KiB / slice", cyc, sink, panel);
    run<6, 12, 5>("6 barrier per slice; loaders stream 20 KiB / slice", cyc, sink, panel);
    run<6, 12, 8>("6 barrier per slice; loaders stream 32 KiB / slice", cyc, sink, panel);
  }
  return 0;
}
