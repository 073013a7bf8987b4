// Micro-benchmark (round 5, verdict item 2): do `buffer_load ... lds` (LDS-DMA) and plain `buffer_load_dwordx4` into
// VGPRs share ONE per-CU limit, or do they add - and which structure feeds the dominant bs=1 kernel best?
// The mock is a skeleton of gemm_ws_conv3_kernel<256,64,8,1,...> (w1/w3 of a single-stream block, M = 500, N = 8192,
// K = 3 x 1536): 256 workgroups of 4 loader + 8 consumer waves; per 64-channel chunk a workgroup needs
//     A: 258 activation rows x 128 B = 33 KiB  (L2-warm: the two 256-row panels are shared by 128 workgroups each)
//     W: 3 taps x 64 rows x 128 B = 24 KiB     (cold: every column panel is read by two workgroups)
//   variant 0  A + W through the loaders' LDS-DMA (what the product does today)
//   variant 1  W through the loaders' LDS-DMA; every CONSUMER wave fetches its own 34 activation rows (32 + halo) with
//              coalesced plain buffer_load_dwordx4 (8 lanes per 128-byte line) and ds_write_b128s them into a wave-private image
//   variant 2  W through the loaders' LDS-DMA; every consumer wave LDS-DMAs its own 34 rows (5 pieces per chunk)
//   variant 3  A only, loaders' LDS-DMA     variant 4  A only, consumers' plain loads + ds_write
//   variant 5  A only, consumers' LDS-DMA   variant 6  W only, loaders' LDS-DMA
//   variant 7  as 0, but the 9 activation pieces of a loader wave are issued 3 per tap slice (chunk c+2's thirds during chunk
//              c's taps) instead of 9 at tap 0      variant 8  A only, spread
// work 0: consumers only synchronise;  work 1: the product's consumer loop (12 ds_read_b128 + 8 MFMA 32x32x16 per tap slice and
// wave, early / late halves ping-pong);  work 2: 64 x 64 tiles per wave PAIR, the pair splits the k-steps of every slice
// (8 ds_read_b128 per 8 MFMA);  work 3: no early / late halves - every wave requests the fragments of slice kt, then multiplies
// slice kt-1 (two fragment register sets).
//   hipcc --offload-arch=gfx950 -O3 -o vmem_paths vmem_paths.hip && ./vmem_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int BM = 256, BN = 64, LW = 4, NWC = 8, NAB = 3;
constexpr int APC = (BM + 2 + 7) / 8, AI = (APC + LW - 1) / LW, ABUF = AI * LW * 1024;   // shared activation image (variant 0 / 3)
constexpr int PRIV = 5 * 1024;                                                            // wave-private image: 34 rows in 5 pieces
constexpr int BI = BN * 128 / 1024 / LW, BSL = BN * 128;
constexpr int C = 1536, K3 = 3 * C, NKC = C / 64, OOB = 0x7ffffff0;

template <int AUX = 0>
__device__ __forceinline__ void buf_lds16(const void* base, unsigned bytes, unsigned char* lds_wave_base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, AUX);
}
__device__ __forceinline__ u32x4 buf_ld16(const void* base, unsigned bytes, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

constexpr int inflight(int nsb, int tap, int ai, int bi) {
  int n = 0;
  for (int j = 1; j <= nsb - 2; ++j) n += bi + (((tap + j) % 3 == 0) ? ai : 0);
  return n;
}

template <int VAR, int WORK, int NSB, int WAUX = 0>
__global__ __launch_bounds__((NWC + LW) * 64) void k(const unsigned char* A, unsigned a_bytes, const unsigned char* W, unsigned w_bytes,
                                                    int M, long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr bool SPREAD = VAR == 7 || VAR == 8;               // the activation chunk's pieces are issued a third per tap slice
  constexpr bool LD_A = VAR == 0 || VAR == 3 || SPREAD;       // loaders stage the shared activation image
  constexpr bool LD_W = VAR == 0 || VAR == 1 || VAR == 2 || VAR == 6 || VAR == 7;
  constexpr bool C_PLAIN = VAR == 1 || VAR == 4, C_DMA = VAR == 2 || VAR == 5;
  constexpr int AIe = LD_A ? AI : 0, BIe = LD_W ? BI : 0;
  constexpr int NPB = C_PLAIN ? 2 : 3;                        // wave-private images per wave
  constexpr int AREG = LD_A ? NAB * ABUF : ((C_PLAIN || C_DMA) ? NWC * NPB * PRIV : 0);  // bytes of the activation region
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = (int)blockIdx.x;
  {  // the product's XCD remap: consecutive tiles (same weight panel) share an L2
    const int nwg = gridDim.x, xcd = bid & 7, slot = bid >> 3, q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tiles_m = (M + BM - 1) / BM;
  const int tm = bid % tiles_m, tn = bid / tiles_m, m0 = tm * BM, n0 = tn * BN;
  const int nk = 3 * NKC;
  const long long c0 = (long long)__builtin_readcyclecounter();
  if (wave >= NWC) {
    const int lw = wave - NWC, lr = lane >> 3, lp = lane & 7;
    int vA[AI], vW[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int j = (lw * AI + i) * 8 + lr, r = m0 - 1 + j;
      vA[i] = (j < BM + 2 && r >= 0 && r < M) ? (int)((unsigned)r * (unsigned)(C * 2) + (unsigned)((lp ^ ((j >> 1) & 7)) * 16)) : OOB;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int rl = (lw * BI + i) * 8 + lr, n = n0 + rl;
      vW[i] = (int)((unsigned)n * (unsigned)(K3 * 2) + (unsigned)((lp ^ ((rl >> 1) & 7)) * 16));
    }
    auto issue = [&](int sl) {
      const int c = sl / 3, tap = sl - 3 * c, ch = c * 64;
      if (LD_A && tap == 0) {
        unsigned char* Ab = lds + (c % NAB) * ABUF;
#pragma unroll
        for (int i = 0; i < AI; ++i) buf_lds16(A, a_bytes, Ab + (lw * AI + i) * 1024, vA[i], ch * 2);
      }
      if (LD_W) {
        unsigned char* Bs = lds + AREG + (sl % NSB) * BSL;
#pragma unroll
        for (int i = 0; i < BI; ++i) buf_lds16<WAUX>(W, w_bytes, Bs + (lw * BI + i) * 1024, vW[i], (tap * C + ch) * 2);
      }
    };
    if constexpr (SPREAD) {
      static_assert(!SPREAD || (AI % 3 == 0 && NSB == 6), "spread form");
      constexpr int AT = AI / 3;
      auto issue_a = [&](int c, int i0, int n) {
        unsigned char* Ab = lds + (c % NAB) * ABUF;
        for (int i = i0; i < i0 + n; ++i) buf_lds16(A, a_bytes, Ab + (lw * AI + i) * 1024, vA[i], c * 128);
      };
      auto issue_w = [&](int sl) {
        if (!LD_W) return;
        const int c = sl / 3, tap = sl - 3 * c;
        unsigned char* Bs = lds + AREG + (sl % NSB) * BSL;
#pragma unroll
        for (int i = 0; i < BI; ++i) buf_lds16<WAUX>(W, w_bytes, Bs + (lw * BI + i) * 1024, vW[i], (tap * C + c * 64) * 2);
      };
      // prologue: A(0), W0, W1, A(1), W2, W3, W4
#pragma unroll
      for (int i = 0; i < AI; ++i) buf_lds16(A, a_bytes, lds + (lw * AI + i) * 1024, vA[i], 0);
      issue_w(0);
      issue_w(1);
#pragma unroll
      for (int i = 0; i < AI; ++i) buf_lds16(A, a_bytes, lds + ABUF + (lw * AI + i) * 1024, vA[i], 128);
      issue_w(2);
      issue_w(3);
      issue_w(4);
      constexpr int PS = AT + BIe;                    // loads per step
      for (int kt0 = 0; kt0 < nk; kt0 += 3) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          const int kt = kt0 + tap;
          if (kt < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BIe + 2 * PS) : "memory");
          else if (kt < nk - 5) {
            if (tap == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BIe + 3 * PS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PS) : "memory");
          } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          const int q = kt + 6, cq = q / 3;           // third (q % 3) = tap of chunk q / 3
          if (cq < NKC) {
#pragma unroll
            for (int i = 0; i < AT; ++i) buf_lds16(A, a_bytes, lds + (cq % NAB) * ABUF + (lw * AI + tap * AT + i) * 1024, vA[tap * AT + i], cq * 128);
          }
          if (kt + 5 < nk) issue_w(kt + 5);
        }
      }
      if (lane == 0 && lw == 0) cyc[blockIdx.x] = (long long)__builtin_readcyclecounter() - c0;
      return;
    }
#pragma unroll
    for (int sl = 0; sl < NSB - 1; ++sl) issue(sl);
    for (int kt0 = 0; kt0 < nk; kt0 += 3) {
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int kt = kt0 + tap;
        if (kt + NSB - 2 < nk) {
          if (tap == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight(NSB, 0, AIe, BIe)) : "memory");
          else if (tap == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight(NSB, 1, AIe, BIe)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight(NSB, 2, AIe, BIe)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NSB - 1 < nk) issue(kt + NSB - 1);
      }
    }
    if (lane == 0 && lw == 0) cyc[blockIdx.x] = (long long)__builtin_readcyclecounter() - c0;
    return;
  }
  // ---- consumer wave: owns activation rows m0 + wave*32 .. +32 (tile rows wave*32 + fi), columns 0..63 (two 32-column fragments)
  const int fi = lane & 31, kh = lane >> 5;
  const int lr = lane >> 3, lp = lane & 7;
  unsigned char* priv = lds + wave * NPB * PRIV;   // wave-private images (variants 1 / 2 / 4 / 5)
  int vP[5];                                       // the wave's own 34 rows as 5 coalesced pieces: image row j <-> activation row m0 + wave*32 - 1 + j
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int j = i * 8 + lr, r = m0 + wave * 32 - 1 + j;
    const int chunk = C_DMA ? (lp ^ ((j >> 1) & 7)) : lp;   // DMA: swizzle on the source side; plain: on the ds_write side
    vP[i] = (j < 34 && r >= 0 && r < M) ? (int)((unsigned)r * (unsigned)(C * 2) + (unsigned)(chunk * 16)) : OOB;
  }
  int a_off[3], a_swz[3], b_row[2], b_sw[2];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int j = (LD_A ? wave * 32 : 0) + fi + t;
    a_off[t] = j * 128;
    a_swz[t] = (j >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    b_row[j] = (j * 32 + fi) * 128;
    b_sw[j] = ((j * 32 + fi) >> 1) & 7;
  }
  auto a_chunk = [&](int s) { return 4 * (s >> 1) + 2 * kh + (s & 1); };
  f32x16 acc[2] = {};
  bf16x8 fa[4], fb[4][2];
  u32x4 ar[5];
  auto ld_plain = [&](int c) {
#pragma unroll
    for (int i = 0; i < 5; ++i) ar[i] = buf_ld16(A, a_bytes, vP[i], c * 128);
  };
  auto st_plain = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = i * 8 + lr;
      *(u32x4*)(priv + buf * PRIV + j * 128 + ((lp ^ ((j >> 1) & 7)) << 4)) = ar[i];
    }
  };
  auto ld_dma = [&](int c) {
#pragma unroll
    for (int i = 0; i < 5; ++i) buf_lds16(A, a_bytes, priv + (c % 3) * PRIV + i * 1024, vP[i], c * 128);
  };
  if (C_PLAIN) {
    ld_plain(0);
    st_plain(0);
    ld_plain(1);
  }
  if (C_DMA) {
    ld_dma(0);
    ld_dma(1);
  }
  const bool late = wave >= NWC / 2;
  auto mma = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], fb[s][j], acc[j], 0, 0, 0);
  };
  if constexpr (WORK == 2) {
    // ---- intra-workgroup K split: waves w and w + 4 (the two consumer waves of a SIMD) share the 64 x 64 tile of rows
    // (w & 3) * 64 ...; the early wave multiplies k-steps 0 / 1 of every slice, the late wave k-steps 2 / 3.  8 fragment reads
    // per 8 MFMAs instead of 12 (the partial tiles are summed once, in the epilogue).
    static_assert(WORK != 2 || LD_A, "K-split form: shared activation image");
    const int wq = wave & 3, kq = wave >> 2;
    f32x16 acc4[2][2] = {};
    bf16x8 ga[2][2], gb[2][2];
    int a2_off[3][2], a2_swz[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = wq * 64 + i * 32 + fi + t;
        a2_off[t][i] = j * 128;
        a2_swz[t][i] = (j >> 1) & 7;
      }
    auto mma4 = [&]() {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s][i], gb[s][j], acc4[i][j], 0, 0, 0);
    };
    for (int c = 0; c < NKC; ++c) {
      const unsigned char* Ab = lds + (c % NAB) * ABUF;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int kt = 3 * c + tap;
        if (late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (late && kt > 0) mma4();
        const unsigned char* Bs = lds + AREG + (kt % NSB) * BSL;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int ch = a_chunk(2 * kq + s);
#pragma unroll
          for (int i = 0; i < 2; ++i) ga[s][i] = *(const bf16x8*)(Ab + a2_off[tap][i] + ((ch ^ a2_swz[tap][i]) << 4));
#pragma unroll
          for (int j = 0; j < 2; ++j) gb[s][j] = *(const bf16x8*)(Bs + b_row[j] + ((ch ^ b_sw[j]) << 4));
        }
        if (!late) mma4();
      }
    }
    if (late) mma4();
    if (acc4[0][0][3] + acc4[1][1][5] + acc4[0][1][0] + acc4[1][0][0] == 123.456f) sink[0] = acc4[0][0][1];
    return;
  }
  if constexpr (WORK == 5) {
    // ---- work 4 with the reads of slice kt issued IN THE SHADOW of the MFMAs of slice kt-1 (sched_group_barrier interleave):
    // no phase of a slice is without MFMAs in flight
    const int wq = wave & 3, kq = wave >> 2;
    f32x16 acc4[2][2] = {};
    bf16x8 ga[2][2][2], gb[2][2][2];
    int a2_off[3][2], a2_swz[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = wq * 64 + i * 32 + fi + t;
        a2_off[t][i] = j * 128;
        a2_swz[t][i] = (j >> 1) & 7;
      }
    auto mma4 = [&](auto set) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[S][s][i], gb[S][s][j], acc4[i][j], 0, 0, 0);
    };
    auto rd4 = [&](auto set, const unsigned char* Ab, const unsigned char* Bs, int tap) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ch = a_chunk(2 * kq + s);
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[S][s][i] = *(const bf16x8*)(Ab + a2_off[tap][i] + ((ch ^ a2_swz[tap][i]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) gb[S][s][j] = *(const bf16x8*)(Bs + b_row[j] + ((ch ^ b_sw[j]) << 4));
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int c = 0; c < NKC; c += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned char* Ab = lds + ((c + u) % NAB) * ABUF;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          const int kt = 3 * (c + u) + tap;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          const unsigned char* Bs = lds + AREG + (kt % NSB) * BSL;
          if ((3 * u + tap) & 1) {
            rd4(S1{}, Ab, Bs, tap);
            mma4(S0{});
          } else {
            rd4(S0{}, Ab, Bs, tap);
            mma4(S1{});
          }
          // the stream of a slice: MFMA (previous slice's fragments), then one fragment request of this slice in its shadow, ...
#pragma unroll
          for (int q = 0; q < 4; ++q) {          // the requests ride in the shadows of the first four MFMAs; the last four cover their latency
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    mma4(S1{});
    if (acc4[0][0][3] + acc4[1][1][5] + acc4[0][1][0] + acc4[1][0][0] == 123.456f) sink[0] = acc4[0][0][1];
    return;
  }
  if constexpr (WORK == 4) {
    // ---- K-split wave pairs (work 2) + read-ahead (work 3): 8 fragment reads per 8 MFMAs, requested one slice ahead
    const int wq = wave & 3, kq = wave >> 2;
    f32x16 acc4[2][2] = {};
    bf16x8 ga[2][2][2], gb[2][2][2];
    int a2_off[3][2], a2_swz[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = wq * 64 + i * 32 + fi + t;
        a2_off[t][i] = j * 128;
        a2_swz[t][i] = (j >> 1) & 7;
      }
    auto mma4 = [&](auto set) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[S][s][i], gb[S][s][j], acc4[i][j], 0, 0, 0);
    };
    auto rd4 = [&](auto set, const unsigned char* Ab, const unsigned char* Bs, int tap) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ch = a_chunk(2 * kq + s);
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[S][s][i] = *(const bf16x8*)(Ab + a2_off[tap][i] + ((ch ^ a2_swz[tap][i]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) gb[S][s][j] = *(const bf16x8*)(Bs + b_row[j] + ((ch ^ b_sw[j]) << 4));
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int c = 0; c < NKC; c += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned char* Ab = lds + ((c + u) % NAB) * ABUF;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          const int kt = 3 * (c + u) + tap;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          const unsigned char* Bs = lds + AREG + (kt % NSB) * BSL;
          if ((3 * u + tap) & 1) {
            rd4(S1{}, Ab, Bs, tap);
            __builtin_amdgcn_sched_barrier(0);
            mma4(S0{});
          } else {
            rd4(S0{}, Ab, Bs, tap);
            __builtin_amdgcn_sched_barrier(0);
            if (kt > 0) mma4(S1{});
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    mma4(S1{});
    if (acc4[0][0][3] + acc4[1][1][5] + acc4[0][1][0] + acc4[1][0][0] == 123.456f) sink[0] = acc4[0][0][1];
    return;
  }
  if constexpr (WORK == 3) {
    // ---- no early / late halves: after the barrier of slice kt EVERY wave first requests the fragments of slice kt (second
    // register set) and then multiplies slice kt-1 - the LDS latency of a wave runs under its own MFMAs instead of after them

    bf16x8 pa[2][4], pb[2][4][2];
    auto mmap = [&](auto set) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[S][s], pb[S][s][j], acc[j], 0, 0, 0);
    };
    auto rd = [&](auto set, const unsigned char* Ab, const unsigned char* Bs, int tap) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pa[S][s] = *(const bf16x8*)(Ab + a_off[tap] + ((a_chunk(s) ^ a_swz[tap]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) pb[S][s][j] = *(const bf16x8*)(Bs + b_row[j] + ((a_chunk(s) ^ b_sw[j]) << 4));
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int c = 0; c < NKC; c += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned char* Ab = lds + ((c + u) % NAB) * ABUF;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          const int kt = 3 * (c + u) + tap;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          const unsigned char* Bs = lds + AREG + (kt % NSB) * BSL;
          if ((3 * u + tap) & 1) {
            rd(S1{}, Ab, Bs, tap);
            __builtin_amdgcn_sched_barrier(0);
            mmap(S0{});
          } else {
            rd(S0{}, Ab, Bs, tap);
            __builtin_amdgcn_sched_barrier(0);
            if (kt > 0) mmap(S1{});
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    mmap(S1{});
    if (acc[0][3] + acc[1][5] == 123.456f) sink[0] = acc[0][1];
    return;
  }
  for (int c = 0; c < NKC; ++c) {
    const unsigned char* Ab = LD_A ? lds + (c % NAB) * ABUF : (C_PLAIN ? priv + (c & 1) * PRIV : priv + (c % 3) * PRIV);
    if (C_PLAIN && c + 1 < NKC) {
      st_plain((c + 1) & 1);                 // chunk c+1 has had a whole chunk period to land
      if (c + 2 < NKC) ld_plain(c + 2);
    }
    if (C_DMA) {
      if (c + 2 < NKC) {
        ld_dma(c + 2);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      } else if (c + 1 < NKC) {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int kt = 3 * c + tap;
      if (WORK && late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (WORK) {
        if (late && kt > 0) mma();
        const unsigned char* Bs = lds + AREG + (kt % NSB) * BSL;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          fa[s] = *(const bf16x8*)(Ab + a_off[tap] + ((a_chunk(s) ^ a_swz[tap]) << 4));
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[s][j] = *(const bf16x8*)(Bs + b_row[j] + ((a_chunk(s) ^ b_sw[j]) << 4));
        }
        if (!late) mma();
      }
    }
  }
  if (WORK && late) mma();
  if (acc[0][3] + acc[1][5] == 123.456f) sink[0] = acc[0][1];
}

template <int VAR, int WORK, int NSB, int WAUX = 0>
void run(const char* label, const unsigned char* A, long a_bytes, const unsigned char* W, long w_panel, int ncopies, long long* cyc, float* sink) {
  auto kk = k<VAR, WORK, NSB, WAUX>;
  constexpr bool LD_A = VAR == 0 || VAR == 3 || VAR == 7 || VAR == 8, C_PLAIN = VAR == 1 || VAR == 4;
  constexpr int AREG = LD_A ? NAB * ABUF : (VAR == 6 ? 0 : NWC * (C_PLAIN ? 2 : 3) * PRIV);
  constexpr int LDS_BYTES = AREG + NSB * BSL;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  const int lds_bytes = LDS_BYTES < 96 * 1024 ? 96 * 1024 : LDS_BYTES;   // one workgroup per CU in every variant
  CK(hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0;
  const int reps = 9;
  for (int rep = 0; rep < reps; ++rep) {
    const unsigned char* w = W + (long)(rep % ncopies) * w_panel;   // a different copy every launch: the weights arrive from HBM
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kk, dim3(256), dim3((NWC + LW) * 64), lds_bytes, 0, A, (unsigned)a_bytes, w, (unsigned)w_panel, 500, cyc, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0) { sum += ms; if (ms < best) best = ms; }
  }
  long long h[256];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double cy = 0;
  for (int i = 0; i < 256; ++i) cy += (double)h[i];
  cy /= 256;
  const double a_b = (VAR == 6) ? 0 : 258.0 * 128 * NKC, w_b = ((VAR >= 3 && VAR <= 5) || VAR == 8) ? 0 : 3.0 * 64 * 128 * NKC;
  printf("%-52s work %d ring %2d : %6.1f us best %6.1f avg | loader cycles %7.0f (%4.0f / chunk) | %5.1f B/clk/CU\n", label, WORK, NSB,
         best * 1e3, sum / (reps - 1) * 1e3, cy, cy / NKC, (a_b + w_b) / cy);
}

int main() {
  const long a_bytes = 512L * C * 2, w_panel = 8192L * K3 * 2;
  const int ncopies = 6;   // 6 x 75.5 MB: more than the 256 MB Infinity Cache between two uses of a copy
  unsigned char *A, *W;
  long long* cyc;
  float* sink;
  CK(hipMalloc(&A, a_bytes));
  CK(hipMalloc(&W, w_panel * ncopies));
  CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, a_bytes));
  CK(hipMemset(W, 0, w_panel * ncopies));
  CK(hipMemset(cyc, 0, 256 * 8));
#define ALL(WORK)                                                                                                        \
  run<0, WORK, 6>("0 A + W: loaders' LDS-DMA (today)", A, a_bytes, W, w_panel, ncopies, cyc, sink);                      \
  run<1, WORK, 6>("1 W loaders' DMA, A consumers' plain loads + ds_write", A, a_bytes, W, w_panel, ncopies, cyc, sink);  \
  run<1, WORK, 9>("1 W loaders' DMA, A consumers' plain loads + ds_write", A, a_bytes, W, w_panel, ncopies, cyc, sink);  \
  run<2, WORK, 5>("2 W loaders' DMA, A consumers' own DMA", A, a_bytes, W, w_panel, ncopies, cyc, sink);                 \
  run<3, WORK, 6>("3 A only: loaders' DMA", A, a_bytes, W, w_panel, ncopies, cyc, sink);                                 \
  run<4, WORK, 6>("4 A only: consumers' plain loads + ds_write", A, a_bytes, W, w_panel, ncopies, cyc, sink);            \
  run<5, WORK, 5>("5 A only: consumers' own DMA", A, a_bytes, W, w_panel, ncopies, cyc, sink);                           \
  run<6, WORK, 6>("6 W only: loaders' DMA", A, a_bytes, W, w_panel, ncopies, cyc, sink);                                 \
  run<6, WORK, 9>("6 W only: loaders' DMA", A, a_bytes, W, w_panel, ncopies, cyc, sink);
  for (int pass = 0; pass < 2; ++pass) {
    ALL(0)
    ALL(1)
    run<0, 2, 6>("0 A + W: loaders' LDS-DMA, K-split wave pairs", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<3, 2, 6>("3 A only: loaders' DMA, K-split wave pairs", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 3, 6>("0 A + W: loaders' LDS-DMA, read-ahead consumers", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<3, 3, 6>("3 A only: loaders' DMA, read-ahead consumers", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<6, 3, 6>("6 W only: loaders' DMA, read-ahead consumers", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<7, 0, 6>("7 A + W: loaders' DMA, A pieces spread over the taps", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<7, 1, 6>("7 A + W: loaders' DMA, A pieces spread over the taps", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<7, 3, 6>("7 A + W: spread + read-ahead consumers", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<7, 2, 6>("7 A + W: spread + K-split pairs", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<7, 4, 6>("7 A + W: spread + K-split pairs + read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 4, 6>("0 A + W: K-split pairs + read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 5, 6>("0 A + W: K-split pairs + interleaved read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<7, 5, 6>("7 A + W spread: K-split pairs + interleaved read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<8, 5, 6>("8 A only spread: K-split pairs + interleaved read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<3, 5, 6>("3 A only: K-split pairs + interleaved read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<6, 5, 6>("6 W only: K-split pairs + interleaved read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<8, 4, 6>("8 A only: spread + K-split pairs + read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<6, 4, 6>("6 W only: K-split pairs + read-ahead", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<8, 1, 6>("8 A only: loaders' DMA, spread", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<6, 0, 6, 2>("6 W only, nt (aux 2)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 0, 6, 2>("0 A + W, W nt (aux 2)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 5, 6, 2>("0 A + W, W nt (aux 2), K-split interleaved", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<6, 0, 6, 1>("6 W only, sc0 (aux 1)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 0, 6, 1>("0 A + W, W sc0 (aux 1)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<6, 0, 6, 3>("6 W only, sc0 nt (aux 3)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 0, 6, 3>("0 A + W, W sc0 nt (aux 3)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 0, 6, 16>("0 A + W, W sc1 (aux 16)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<0, 0, 6, 18>("0 A + W, W sc1 nt (aux 18)", A, a_bytes, W, w_panel, ncopies, cyc, sink);
    run<8, 3, 6>("8 A only: spread + read-ahead consumers", A, a_bytes, W, w_panel, ncopies, cyc, sink);
  }
  return 0;
}
