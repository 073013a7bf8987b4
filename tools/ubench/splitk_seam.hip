// Micro-benchmark: what does an IN-LAUNCH split-K reduction cost on MI355X, against the launch-boundary reduce the sampler uses?
// (round-2 verdict item 3: "let the last-arriving K range of a tile sum the slabs out of L2 and apply the gated residual").
//
// Shapes of a gated-residual GEMM of the DiT at M = 500: 48 output tiles of 128x128 fp32 (4 row x 12 column tiles of [512,1536]),
// k = 5 K ranges -> 240 workgroups of 512 threads, one per CU (160 KiB of LDS, like the GEMM kernels).  The K loop is left out:
// every workgroup owns its partial tile in registers (32 floats per thread) from the start, so the timings are the SEAM alone.
//
//   A  boundary : kernel 1 stores the slabs (plain dwordx4);  kernel 2 (LayerNorm-shaped: one wave triple per row) does
//                 x += gate * (sum_s slab_s) and writes the normalised row in bf16          <- what foley_rt.hip does today
//   B  in-launch, write-through: kernel 1 stores the slabs with sc1 stores, drains, takes a ticket (relaxed agent atomic); the last
//                 arriver of a tile acquires, sums the other k-1 slabs (sc1 loads) + its own registers in fixed order s = 0..k-1,
//                 applies x += gate * sum;  kernel 2 is then the plain LayerNorm (reads x only)
//   C  in-launch, plain stores + agent-scope release fence before the ticket, acquire fence + plain loads in the reducer
// Every variant is run with fp32 slabs and with bf16 slabs (the sampler's current form), results are checked against the host.
//   hipcc --offload-arch=gfx950 -O3 -o splitk_seam splitk_seam.hip && ./splitk_seam
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int M = 500, MP = 512, N = 1536, KS = 5, TM = 4, TN = 12, NT = 512;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  const b2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float lo16(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi16(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// the partial tile a (tile, ks) workgroup "computed": a deterministic function of (row, col, ks)
__device__ __host__ inline float partial(int row, int col, int ks) { return (float)((row * 7 + col * 3 + ks * 11) % 97) * (1.0f / 64.0f) - 0.7f; }

// thread t of a workgroup owns, in pass p (0..7), row p*16 + t/32 and columns (t%32)*4 .. +3 of the 128x128 tile
template <bool H16, int MODE>   // MODE 0: plain stores only (variant A); 1: sc1 + ticket (B); 2: plain + release + ticket (C)
__global__ __launch_bounds__(NT) void k_gemm_tail(void* slabs, unsigned* cnt, float* x, const float* gate) {
  extern __shared__ unsigned lds[];   // 160 KiB requested: one workgroup per CU
  const int bid = blockIdx.x, ks = bid % KS, tile = bid / KS, tm = tile % TM, tn = tile / TM;
  const int t = threadIdx.x, r0 = t >> 5, c = (t & 31) * 4;
  f32x4 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[p][u] = partial(tm * 128 + p * 16 + r0, tn * 128 + c + u, ks);
  // ---- publish the partial tile
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = tm * 128 + p * 16 + r0;
    if (row >= M) continue;
    const long e = ((long)ks * M + row) * N + tn * 128 + c;
    if (H16) {
      uint2 w = {pack2(acc[p][0], acc[p][1]), pack2(acc[p][2], acc[p][3])};
      uint2* d = (uint2*)((unsigned short*)slabs + e);
      if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" ::"v"(d), "v"(w) : "memory");
      else *d = w;
    } else {
      f32x4* d = (f32x4*)((float*)slabs + e);
      if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(d), "v"(acc[p]) : "memory");
      else *d = acc[p];
    }
  }
  if (MODE == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its stores are performed
  __syncthreads();
  if (t == 0) {
    if (MODE == 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // buffer_wbl2 sc1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds[0] = __hip_atomic_fetch_add(cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (lds[0] != KS - 1) return;
  // ---- last arriver: sum the K ranges in fixed order, apply the gated residual
  if (t == 0) {
    __hip_atomic_store(cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = tm * 128 + p * 16 + r0;
    if (row >= M) continue;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    f32x4 part[KS];
    uint2 raw[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {     // all k loads in flight, ONE wait
      const long e = ((long)s * M + row) * N + tn * 128 + c;
      if (H16) {
        const uint2* src = (const uint2*)((const unsigned short*)slabs + e);
        if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(raw[s]) : "v"(src) : "memory");
        else raw[s] = *src;
      } else {
        const f32x4* src = (const f32x4*)((const float*)slabs + e);
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[s]) : "v"(src) : "memory");
        else part[s] = *src;
      }
    }
    if (MODE == 1) {
      if (H16) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4])::"memory");
      else asm volatile("s_waitcnt vmcnt(0)" : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]), "+v"(part[4])::"memory");
    }
    if (H16) {
#pragma unroll
      for (int s = 0; s < KS; ++s) part[s] = f32x4{lo16(raw[s].x), hi16(raw[s].x), lo16(raw[s].y), hi16(raw[s].y)};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) sum += part[s];
    const f32x4 g = *(const f32x4*)(gate + tn * 128 + c);
    f32x4* xp = (f32x4*)(x + (long)row * N + tn * 128 + c);
    *xp = *xp + g * sum;
  }
}

// LayerNorm-shaped consumer: 2 rows per workgroup, 3 waves per row (the sampler's ln_mod_wide_kernel geometry).
// PEND: first x += gate * (sum of the k slabs)
template <bool H16, bool PEND>
__global__ __launch_bounds__(384) void k_ln(float* x, const void* slabs, const float* gate, unsigned short* out) {
  __shared__ float red[2][2][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, rb = wave / 3, part = wave % 3;
  const int row = min((int)blockIdx.x * 2 + rb, M - 1);
  f32x4 v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = ((const f32x4*)(x + (long)row * N))[part * 128 + lane + i * 64];
  if (PEND) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, t[KS][2], g[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) g[i] = ((const f32x4*)gate)[part * 128 + lane + i * 64];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long e = ((long)s * M + row) * N + 4L * (part * 128 + lane + i * 64);
        if (H16) {
          const uint2 w = *(const uint2*)((const unsigned short*)slabs + e);
          t[s][i] = f32x4{lo16(w.x), hi16(w.x), lo16(w.y), hi16(w.y)};
        } else {
          t[s][i] = *(const f32x4*)((const float*)slabs + e);
        }
      }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] += t[s][i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v[i] += g[i] * acc[i];
      if ((int)blockIdx.x * 2 + rb < M) ((f32x4*)(x + (long)row * N))[part * 128 + lane + i * 64] = v[i];
    }
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int u = 0; u < 4; ++u) { s += v[i][u]; q += v[i][u] * v[i][u]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (lane == 0) { red[rb][0][part] = s; red[rb][1][part] = q; }
  __syncthreads();
  const float mean = (red[rb][0][0] + red[rb][0][1] + red[rb][0][2]) / N;
  const float var = (red[rb][1][0] + red[rb][1][1] + red[rb][1][2]) / N - mean * mean;
  const float rstd = rsqrtf(var + 1e-6f);
  if ((int)blockIdx.x * 2 + rb >= M) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint2 w = {pack2((v[i][0] - mean) * rstd, (v[i][1] - mean) * rstd), pack2((v[i][2] - mean) * rstd, (v[i][3] - mean) * rstd)};
    *(uint2*)(out + (long)row * N + 4L * (part * 128 + lane + i * 64)) = w;
  }
}

template <bool H16>
static void run(const char* name, int mode, float* x, void* slabs, unsigned* cnt, float* gate, unsigned short* out, const std::vector<float>& want) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t lds = 160 * 1024;
  auto launch = [&]() {
    if (mode == 0) {
      hipLaunchKernelGGL((k_gemm_tail<H16, 0>), dim3(TM * TN * KS), dim3(NT), lds, st, slabs, cnt, x, gate);
      hipLaunchKernelGGL((k_ln<H16, true>), dim3(250), dim3(384), 0, st, x, slabs, gate, out);
    } else if (mode == 1) {
      hipLaunchKernelGGL((k_gemm_tail<H16, 1>), dim3(TM * TN * KS), dim3(NT), lds, st, slabs, cnt, x, gate);
      hipLaunchKernelGGL((k_ln<H16, false>), dim3(250), dim3(384), 0, st, x, slabs, gate, out);
    } else {
      hipLaunchKernelGGL((k_gemm_tail<H16, 2>), dim3(TM * TN * KS), dim3(NT), lds, st, slabs, cnt, x, gate);
      hipLaunchKernelGGL((k_ln<H16, false>), dim3(250), dim3(384), 0, st, x, slabs, gate, out);
    }
  };
  CK(hipFuncSetAttribute((const void*)k_gemm_tail<H16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)k_gemm_tail<H16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)k_gemm_tail<H16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // correctness: one pair from x = 0
  CK(hipMemsetAsync(x, 0, (size_t)M * N * 4, st));
  CK(hipMemsetAsync(cnt, 0, 256, st));
  launch();
  std::vector<float> hx((size_t)M * N);
  CK(hipMemcpyAsync(hx.data(), x, hx.size() * 4, hipMemcpyDeviceToHost, st));
  CK(hipStreamSynchronize(st));
  double err = 0;
  for (size_t i = 0; i < hx.size(); ++i) err = fmax(err, fabs((double)hx[i] - want[i]));
  // timing: a graph of 100 pairs (as the sampler replays its iteration), replayed 5 times
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 100; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms);
  }
  printf("%-58s %s slabs: %6.2f us per (GEMM tail + LayerNorm) pair   max |x - expected| %.2e\n", name, H16 ? "bf16" : "fp32", best * 10.0f, err);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipStreamDestroy(st));
}

int main() {
  float *x, *gate;
  void* slabs;
  unsigned* cnt;
  unsigned short* out;
  CK(hipMalloc(&x, (size_t)MP * N * 4));
  CK(hipMalloc(&slabs, (size_t)KS * MP * N * 4));
  CK(hipMalloc(&cnt, 256));
  CK(hipMalloc(&gate, N * 4));
  CK(hipMalloc(&out, (size_t)MP * N * 2));
  std::vector<float> hg(N);
  for (int i = 0; i < N; ++i) hg[i] = 0.5f + (float)(i % 13) * 0.03f;
  CK(hipMemcpy(gate, hg.data(), N * 4, hipMemcpyHostToDevice));
  for (int h16 = 0; h16 < 2; ++h16) {
    std::vector<float> want((size_t)M * N);
    for (int r = 0; r < M; ++r)
      for (int c = 0; c < N; ++c) {
        float s = 0.f;
        for (int k = 0; k < KS; ++k) {
          float p = partial(r, c, k);
          if (h16) {   // bf16 round-to-nearest-even of the partial
            unsigned u;
            memcpy(&u, &p, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            u &= 0xffff0000u;
            memcpy(&p, &u, 4);
          }
          s += p;
        }
        want[(size_t)r * N + c] = hg[c] * s;
      }
    if (h16) {
      run<true>("A  launch-boundary reduce (LayerNorm sums the slabs)", 0, x, slabs, cnt, gate, out, want);
      run<true>("B  in-launch, sc1 slabs + ticket, last arriver reduces", 1, x, slabs, cnt, gate, out, want);
      run<true>("C  in-launch, plain slabs + agent release / acquire", 2, x, slabs, cnt, gate, out, want);
    } else {
      run<false>("A  launch-boundary reduce (LayerNorm sums the slabs)", 0, x, slabs, cnt, gate, out, want);
      run<false>("B  in-launch, sc1 slabs + ticket, last arriver reduces", 1, x, slabs, cnt, gate, out, want);
      run<false>("C  in-launch, plain slabs + agent release / acquire", 2, x, slabs, cnt, gate, out, want);
    }
  }
  return 0;
}
