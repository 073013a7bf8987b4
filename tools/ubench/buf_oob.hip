// Semantics check for buffer_load_dwordx4 ... lds on gfx950: (1) do lanes whose offset is out of
// range write ZEROS to LDS?  (2) is the SGPR offset part of the range check?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k(const unsigned* src, unsigned num_records, int soff, int voff_hi_from_lane, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[256];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xABABABABu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)num_records, 0x00020000);
  int voff = lane * 16;
  if (lane >= voff_hi_from_lane) voff = 0x7ffffff0;   // far out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
  unsigned *src, *out, h[256];
  CK(hipMalloc(&src, 1 << 20));
  CK(hipMalloc(&out, 1024));
  unsigned* hs = (unsigned*)malloc(1 << 20);
  for (int i = 0; i < (1 << 18); ++i) hs[i] = 0x10000000u + i;
  CK(hipMemcpy(src, hs, 1 << 20, hipMemcpyHostToDevice));
  struct { unsigned nr; int soff; int hi; const char* what; } cases[] = {
      {4096, 0, 32, "lanes >= 32 far out of range (voffset)"},
      {512, 0, 64, "num_records 512: lanes >= 32 beyond the end"},
      {4096, 3584, 64, "soffset 3584, num_records 4096: lanes >= 32 end beyond num_records only via soffset"},
      {4096, 8192, 64, "soffset 8192 > num_records"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, c.nr, c.soff, c.hi, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost));
    printf("%s\n   lane0 %08x  lane31 %08x  lane32 %08x  lane63 %08x   (in-range value would be %08x / %08x)\n", c.what, h[0], h[31 * 4], h[32 * 4],
           h[63 * 4], 0x10000000u + (c.soff / 4) + 32 * 4, 0x10000000u + (c.soff / 4) + 63 * 4);
  }
  return 0;
}
