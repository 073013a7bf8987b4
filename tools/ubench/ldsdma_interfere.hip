// Micro-benchmark: does MFMA issue or LDS fragment reading slow down a concurrent global_load_lds
// stream on the same CU?  8 loader waves stream 32 KiB slices into an LDS ring (as the GEMM loop
// does, L2-warm or HBM-cold source); 8 more waves of the same workgroup run, independently,
//   mode 0: nothing      mode 1: back-to-back bf16 MFMAs      mode 2: ds_read_b128 of a private LDS region
//   mode 3: ds_read_b128 + MFMA
// and every role reports its own wall time (s_memrealtime, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_interfere ldsdma_interfere.hip && ./ldsdma_interfere
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)(unsigned)(a >> 32);   // stride 0
  r[2] = (int)bytes;
  r[3] = 0x00020000;                 // raw buffer, dword format (gfx90a / gfx94x / gfx950)
  return r;
}
constexpr int SLICE = 32768, NS = 3, RING = SLICE * NS, PRIV = 32768;

__global__ __launch_bounds__(1024) void k(const unsigned char* base, long wg_stride, long pitch, int nslice, int mode, int work, int prio, int nmfma, int use_buf,
                                          long long* times, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = wall_clock64();
  const long long c0 = (long long)__builtin_readcyclecounter();
  if (wave < 8) {  // loader: 4 pieces of 1 KiB per wave and slice, rows of `pitch` bytes, 128 B per row and slice
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    const unsigned char* wg = base + (long)blockIdx.x * wg_stride;
    const unsigned char* src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) src[p] = wg + ((long)(wave * 4 + p) * 8 + (lane >> 3)) * pitch + (lane & 7) * 16;
    int issued = 0;
    // buffer form: SGPR resource + per-lane 32-bit offset (loop invariant) + SGPR slice offset: no VALU per issue
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, 0x7fffffff, 0x00020000);
    int voff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) voff[p] = (int)(((long)(wave * 4 + p) * 8 + (lane >> 3)) * pitch + (lane & 7) * 16);
    auto issue = [&](int stage) {
      if (use_buf) {
        const int soff = issued * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + stage * SLICE + (wave * 4 + p) * 1024), 16,
                                               voff[p], soff, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16(src[p] + (long)issued * 128, lds + stage * SLICE + (wave * 4 + p) * 1024);
      }
      ++issued;
    };
    for (int s = 0; s < NS - 1; ++s) issue(s);
    int stage = 0;
    for (int kt = 0; kt < nslice; ++kt) {
      if (kt + NS - 2 < nslice) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (kt + NS - 1 < nslice) issue(stage == 0 ? NS - 1 : stage - 1);
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (lane == 0 && wave == 0) { times[blockIdx.x * 2] = wall_clock64() - t0; times[1024 + blockIdx.x] = (long long)__builtin_readcyclecounter() - c0; }
  } else if (mode != 0 && wave < 8 + nmfma) {
    if (prio == 2) __builtin_amdgcn_s_setprio(3);
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a = {}, b = {};
    const unsigned char* my = lds + RING + (wave - 8) * 4096 + lane * 16;
    for (int it = 0; it < work; ++it) {
      if (mode & 2) {
        u32x4 v0 = *(const u32x4*)(my), v1 = *(const u32x4*)(my + 1024), v2 = *(const u32x4*)(my + 2048);
        a = __builtin_bit_cast(bf16x8, v0 ^ v2);
        b = __builtin_bit_cast(bf16x8, v1);
      }
      if (mode & 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        }
      } else {
        acc0[0] += (float)a[0] + (float)b[1];
      }
    }
    if (lane == 0 && wave == 8) times[blockIdx.x * 2 + 1] = wall_clock64() - t0;
    if (acc0[3] + acc1[5] == 123.456f) sink[0] = acc0[1];
  }
}

int main() {
  const long bytes = 3L << 30;
  unsigned char* buf;
  long long* times;
  float* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&times, 16384));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, bytes));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RING + PRIV));
  const int grid = 256, nslice = 256;
  long long h[512], hc[256];
  for (int warm = 0; warm < 2; ++warm) {   // 0: HBM-cold rows (64 KiB pitch, private 8 MiB per WG); 1: L2-warm shared panel
    for (int mode = 0; mode < 4; ++mode) {
      for (int cfg = 0; cfg < 7; ++cfg) {
        const int work = mode == 0 ? 0 : 4000;
        if (mode == 0 && cfg > 0 && cfg != 5) continue;
        if (mode == 2 && cfg > 0) continue;
        const int prio = cfg == 1 ? 1 : (cfg == 2 ? 2 : 0);
        const int nmfma = cfg == 3 ? 4 : (cfg == 4 ? 2 : 8);
        const int use_buf = cfg >= 5 ? 1 : 0;
        if (cfg == 6 && mode == 0) continue;
        const int prio2 = cfg == 6 ? 1 : prio;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(times, 0, 16384));
          hipLaunchKernelGGL(k, dim3(grid), dim3(1024), RING + PRIV, 0, buf, warm ? 0L : (8L << 20), 65536L, nslice, mode, work, prio2, nmfma, use_buf, times, sink);
          CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, times, grid * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc, times + 1024, grid * 8, hipMemcpyDeviceToHost));
        double cyc = 0; for (int i = 0; i < grid; ++i) cyc += hc[i]; cyc /= grid;
        double tl = 0, tc = 0;
        for (int i = 0; i < grid; ++i) { tl += h[2 * i]; tc += h[2 * i + 1]; }
        tl /= grid * 100.0; tc /= grid * 100.0;
        printf("%s mode %d prio %d mfma-waves %d buf %d work %5d : loader %7.1f us (%5.1f GB/s/CU)  compute waves %7.1f us (%6.1f ns/iter)  loader clock %.2f GHz\n", warm ? "L2-warm " : "HBM-cold",
               mode, prio2, nmfma, use_buf, work, tl, (double)nslice * SLICE / tl / 1e3, tc, work ? tc * 1e3 / work : 0.0, cyc / tl / 1e3);
      }
    }
  }
  return 0;
}
