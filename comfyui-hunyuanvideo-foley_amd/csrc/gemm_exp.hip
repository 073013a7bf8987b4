// EXPERIMENTAL mainloop variants of the GEMM engine (bf16, plain store epilogue only), reachable
// through foley_op_gemm with tile codes >= 100 and used ONLY by tools/gemm_bench.py (never by the
// runtime): pipeline-depth / LDS-buffering sweeps (1xx), K-walk rotation (2xx/3xx: no effect - the
// loop is not channel-camping) and phase ablations (4xx: no global loads in the loop, 5xx: also no
// LDS writes / barriers).  Ablation result at M=500, 64x64 tiles: per K-slice the three phases
// cost 0.21 us (global) + 0.13 us (LDS write + barrier) + 0.12 us (fragment reads + MFMA) and run
// back to back; see DESIGN.md section 4.
#include "kernels.h"

namespace {

constexpr int PITCH = 144;

template <int BM, int BN, int WM, int WN, int NS, int LB, int ROT>
__global__ __launch_bounds__(WM* WN * 64) void gemm_exp_kernel(const GemmArgs g) {
  constexpr int NT = WM * WN * 64;
  constexpr int EPC = 8, BK = 64;
  constexpr int RPP = NT / 8;  // rows per loader pass
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int RA = BM / RPP, RB = BN / RPP;
  constexpr int STAGE = (BM + BN) * PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int chunk = tid & 7, lrow = tid >> 3;

  const bf16_t* ap[RA];
  bool a_ok[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int r = m0 + lrow + i * RPP;
    a_ok[i] = r < g.M;
    ap[i] = (const bf16_t*)g.A + (long)(a_ok[i] ? r : 0) * g.lda + chunk * EPC;
  }
  const bf16_t* wp[RB];
  bool w_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + lrow + i * RPP;
    w_ok[i] = n < g.N;
    wp[i] = (const bf16_t*)g.W + (long)(w_ok[i] ? n : 0) * g.K + chunk * EPC;
  }
  u32x4 ra[NS][RA], rw[NS][RB];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fi = lane & 31, kh = lane >> 5;
  const int nk = g.K / BK;
  // ROT: every weight panel starts its K walk at a different slice so that concurrently running
  // workgroups do not all hit the same memory channels (row strides are multiples of 1 KiB)
  const int rot = ROT == 1 ? (tn * 5) % nk : (ROT == 2 ? (bid * 3) % nk : 0);

#define GLOAD(slot, kt)                                                       \
  {                                                                           \
    int kk_ = (kt) + rot; if (kk_ >= nk) kk_ -= nk;                           \
    const int k0_ = kk_*BK;                                                  \
    _Pragma("unroll") for (int i = 0; i < RA; ++i) ra[slot][i] = *(const u32x4*)(ap[i] + k0_); \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) rw[slot][i] = *(const u32x4*)(wp[i] + k0_); \
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) GLOAD(s, s);

  for (int kt0 = 0; kt0 < nk; kt0 += NS) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int kt = kt0 + j;
      if (kt < nk) {
        if (ROT < 3) { if (kt + NS - 1 < nk) GLOAD((j + NS - 1) % NS, kt + NS - 1); }
        unsigned char* As = lds + (LB == 2 ? (kt & 1) * STAGE : 0);
        unsigned char* Bs = As + BM * PITCH;
        if (LB == 1) __syncthreads();
        if (ROT < 4 || kt == 0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *(u32x4*)(As + (lrow + i * RPP) * PITCH + chunk * 16) = a_ok[i] ? ra[ROT >= 3 ? 0 : j][i] : zero4;
#pragma unroll
        for (int i = 0; i < RB; ++i) *(u32x4*)(Bs + (lrow + i * RPP) * PITCH + chunk * 16) = w_ok[i] ? rw[ROT >= 3 ? 0 : j][i] : zero4;
        __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bf16x8 a[FM], b[FN];
#pragma unroll
          for (int i = 0; i < FM; ++i)
            a[i] = *(const bf16x8*)(As + (wm * TM + i * 32 + fi) * PITCH + s * 32 + kh * 16);
#pragma unroll
          for (int jj = 0; jj < FN; ++jj)
            b[jj] = *(const bf16x8*)(Bs + (wn * TN + jj * 32 + fi) * PITCH + s * 32 + kh * 16);
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jj = 0; jj < FN; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
        }
      }
    }
  }
#undef GLOAD

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * TN + j * 32 + fi;
        if (col >= g.N) continue;
        float v = acc[i][j][e];
        if (g.bias) v += g.bias[col];
        ((float*)g.out0)[(long)row * g.out_row + col] = v;
      }
    }
}

template <int BM, int BN, int WM, int WN, int NS, int LB, int ROT = 0>
int launch_exp(const GemmArgs& g, hipStream_t st) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  const size_t lds = (size_t)LB * (BM + BN) * PITCH;
  auto k = gemm_exp_kernel<BM, BN, WM, WN, NS, LB, ROT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), lds, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

}  // namespace

int launch_gemm_exp(const GemmArgs& g, int code, hipStream_t st) {
  if (g.K % 64) return foley_set_err("exp GEMM: K % 64", __FILE__, __LINE__);
  switch (code) {
    // shape 1: 128x128 / 4 waves
    case 101: return launch_exp<128, 128, 2, 2, 2, 1>(g, st);   // = production structure
    case 111: return launch_exp<128, 128, 2, 2, 2, 2>(g, st);   // + LDS double buffer
    case 121: return launch_exp<128, 128, 2, 2, 3, 2>(g, st);   // + 2 tiles in flight
    case 131: return launch_exp<128, 128, 2, 2, 4, 2>(g, st);   // + 3 tiles in flight
    case 201: return launch_exp<128, 128, 2, 2, 2, 1, 1>(g, st);   // production structure + K rotation per panel
    case 221: return launch_exp<128, 128, 2, 2, 3, 2, 1>(g, st);
    case 231: return launch_exp<128, 128, 2, 2, 4, 2, 1>(g, st);
    case 301: return launch_exp<128, 128, 2, 2, 2, 1, 2>(g, st);   // rotation per workgroup
    case 321: return launch_exp<128, 128, 2, 2, 3, 2, 2>(g, st);
    case 433: return launch_exp<64, 64, 2, 2, 4, 2, 3>(g, st);     // ablation: no global loads in the loop
    case 533: return launch_exp<64, 64, 2, 2, 4, 2, 4>(g, st);     // ablation: + no LDS writes / barriers
    case 437: return launch_exp<128, 128, 4, 2, 4, 2, 3>(g, st);
    case 537: return launch_exp<128, 128, 4, 2, 4, 2, 4>(g, st);
    case 203: return launch_exp<64, 64, 2, 2, 2, 1, 1>(g, st);
    case 223: return launch_exp<64, 64, 2, 2, 3, 2, 1>(g, st);
    case 323: return launch_exp<64, 64, 2, 2, 3, 2, 2>(g, st);
    case 222: return launch_exp<64, 128, 2, 2, 3, 2, 1>(g, st);
    case 225: return launch_exp<256, 128, 4, 2, 3, 2, 1>(g, st);
    case 227: return launch_exp<128, 128, 4, 2, 3, 2, 1>(g, st);
    // shape 2: 64x128
    case 102: return launch_exp<64, 128, 2, 2, 2, 1>(g, st);
    case 122: return launch_exp<64, 128, 2, 2, 3, 2>(g, st);
    case 132: return launch_exp<64, 128, 2, 2, 4, 2>(g, st);
    // shape 3: 64x64
    case 103: return launch_exp<64, 64, 2, 2, 2, 1>(g, st);
    case 123: return launch_exp<64, 64, 2, 2, 3, 2>(g, st);
    case 133: return launch_exp<64, 64, 2, 2, 4, 2>(g, st);
    case 143: return launch_exp<64, 64, 2, 2, 6, 2>(g, st);
    // shape 5: 256x128 / 8 waves
    case 125: return launch_exp<256, 128, 4, 2, 3, 2>(g, st);
    case 135: return launch_exp<256, 128, 4, 2, 4, 2>(g, st);
    // shape 6: 128x256 / 8 waves
    case 126: return launch_exp<128, 256, 2, 4, 3, 2>(g, st);
    // shape 7: 128x128 / 8 waves (4x2 waves, 32x64 per wave)
    case 127: return launch_exp<128, 128, 4, 2, 3, 2>(g, st);
    case 137: return launch_exp<128, 128, 4, 2, 4, 2>(g, st);
  }
  return foley_set_err("exp GEMM: unknown variant", __FILE__, __LINE__);
}
