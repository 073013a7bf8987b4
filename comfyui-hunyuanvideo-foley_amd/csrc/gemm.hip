// MFMA GEMM / conv-as-GEMM engine for gfx950 (see kernels.h for the addressing model).
//
// One workgroup = 4 wavefronts (256 lanes) computes a BM x BN output tile with 32x32 MFMA
// fragments: v_mfma_f32_32x32x2_f32 for fp32 operands (exact fp32 FMA chain - the parity mode)
// and v_mfma_f32_32x32x16_bf16 for bf16 operands, both accumulating in fp32.  Operand tiles are
// 128-byte K-slices (32 fp32 / 64 bf16) staged through LDS with a 144-byte row pitch, which makes
// the ds_read_b128 fragment reads bank-conflict free (MI355X_MICROARCH.md, LDS table).  Global
// loads of tile k+1 are issued into registers before the MFMA block of tile k (register-prefetch
// pipeline).  Workgroup ids are remapped so tiles that share a weight panel run on one XCD (L2).
#include "kernels.h"

namespace {

constexpr int LDS_PITCH = 144;  // bytes per staged row: 128 data + 16 pad

template <typename T> struct Frag;
template <> struct Frag<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
};
template <> struct Frag<bf16_t> {
  static constexpr int EPC = 8;
};

__device__ __forceinline__ float act_epi(float v, int epi) {
  if (epi == EPI_SILU_T) return silu_f(v);
  if (epi == EPI_GELU_T) return gelu_tanh_f(v);
  return v;
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmArgs g) {
  constexpr int NT = WM * WN * 64;
  static_assert(NT == 256, "tile loader assumes 256 threads");
  constexpr int EPC = Frag<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  static_assert(EPI != EPI_SILUGATE_T || (FN % 2 == 0), "gated epilogue needs fragment pairs");

  __shared__ __attribute__((aligned(16))) unsigned char lds[(BM + BN) * LDS_PITCH];
  unsigned char* As = lds;
  unsigned char* Bs = lds + BM * LDS_PITCH;

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {  // bijective XCD remap: consecutive tile ids (same weight panel) share an XCD / L2
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

  long a_base[RA];
  int a_q[RA];
  bool a_ok[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int r = m0 + lrow + i * 32;
    a_ok[i] = r < g.M;
    const int rr = a_ok[i] ? r : 0;
    const int b = rr / g.segV, q = rr - b * g.segV;
    a_base[i] = ((long)b * g.segS + q) * g.lda;
    a_q[i] = q;
  }
  long w_off[RB];
  bool w_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + lrow + i * 32;
    w_ok[i] = n < g.N;
    w_off[i] = (long)(w_ok[i] ? n : 0) * g.K;
  }

  const T* __restrict__ Ag = (const T*)g.A;
  const T* __restrict__ Wg = (const T*)g.W;
  uint4 ra[RA], rw[RB];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const int tap = k0 / g.tapC;
    const int c0 = k0 - tap * g.tapC;
    const int toff = g.tap0 + tap * g.dil;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int st = a_q[i] + toff;
      const bool v = a_ok[i] && st >= 0 && st < g.segS;
      const T* p = Ag + a_base[i] + (long)toff * g.lda + c0 + chunk * EPC;
      ra[i] = v ? *(const uint4*)p : zero4;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const T* p = Wg + w_off[i] + k0 + chunk * EPC;
      rw[i] = w_ok[i] ? *(const uint4*)p : zero4;
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fi = lane & 31, kh = lane >> 5;
  const int nk = g.K / BK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RA; ++i) *(uint4*)(As + (lrow + i * 32) * LDS_PITCH + chunk * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *(uint4*)(Bs + (lrow + i * 32) * LDS_PITCH + chunk * 16) = rw[i];
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);

    if constexpr (sizeof(T) == 4) {
      // lane (fi, kh) owns k = kh*16 .. kh*16+15 of its row; MFMA step j contracts the k pair
      // (j, 16 + j) - any pairing is valid as long as A and B use the same one.
      float a[FM][16], b[FN][16];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const unsigned char* p = As + (wm * TM + i * 32 + fi) * LDS_PITCH + kh * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *(const f32x4*)(p + c * 16);
          a[i][c * 4 + 0] = v[0]; a[i][c * 4 + 1] = v[1]; a[i][c * 4 + 2] = v[2]; a[i][c * 4 + 3] = v[3];
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned char* p = Bs + (wn * TN + j * 32 + fi) * LDS_PITCH + kh * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *(const f32x4*)(p + c * 16);
          b[j][c * 4 + 0] = v[0]; b[j][c * 4 + 1] = v[1]; b[j][c * 4 + 2] = v[2]; b[j][c * 4 + 3] = v[3];
        }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
          a[i] = *(const bf16x8*)(As + (wm * TM + i * 32 + fi) * LDS_PITCH + s * 32 + kh * 16);
#pragma unroll
        for (int j = 0; j < FN; ++j)
          b[j] = *(const bf16x8*)(Bs + (wn * TN + j * 32 + fi) * LDS_PITCH + s * 32 + kh * 16);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------ epilogue
  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  const bool plain_out = g.osegV >= g.M;
  float sn_a[FN], sn_ia[FN];
  if constexpr (EPI == EPI_DAC) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * TN + j * 32 + fi;
      sn_a[j] = (g.out1 && col < g.N) ? g.alpha[col % g.alphaC] : 1.0f;
      sn_ia[j] = 1.0f / (sn_a[j] + 1e-9f);
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      if (row >= g.M) continue;
      long obase, orel;
      if (plain_out) {
        obase = 0;
        orel = (long)row * g.out_row + g.out_shift;
      } else {
        const int b = row / g.osegV, q = row - b * g.osegV;
        obase = (long)b * g.out_seg;
        orel = (long)q * g.out_row + g.out_shift;
      }
      const float* rbp = nullptr;
      if constexpr (EPI == EPI_GATE_RES || EPI == EPI_STORE_F32) {
        if (g.rb.p) rbp = rb_row(g.rb, row);
      }
      if constexpr (EPI == EPI_SILUGATE_T) {
#pragma unroll
        for (int j = 0; j < FN; j += 2) {
          const int colp = n0 + wn * TN + j * 32;  // packed column of the 'a' group
          const int col = (colp >> 1) + fi;
          if (colp + 32 + fi >= g.N) continue;
          float va = acc[i][j][e], vb = acc[i][j + 1][e];
          if (g.bias) { va += g.bias[colp + fi]; vb += g.bias[colp + 32 + fi]; }
          const long rel = orel + col;
          if (g.out_check && (rel < 0 || rel >= g.out_seg)) continue;
          ((T*)g.out0)[obase + rel] = Cvt<T>::to(silu_f(va) * vb);
        }
      } else {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int col = n0 + wn * TN + j * 32 + fi;
          if (col >= g.N) continue;
          const long rel = orel + col;
          if (g.out_check && (rel < 0 || rel >= g.out_seg)) continue;
          const long off = obase + rel;
          float v = acc[i][j][e];
          if (g.bias) v += g.bias[col];
          if constexpr (EPI == EPI_STORE_F32) {
            if (rbp) v += rbp[col];
            ((float*)g.out0)[off] = v;
          } else if constexpr (EPI == EPI_GATE_RES) {
            float* x = (float*)g.out0;
            x[off] = x[off] + v * rbp[col];
          } else if constexpr (EPI == EPI_DAC) {
            if (g.res) v += g.res[off];
            if (g.out0) ((float*)g.out0)[off] = v;
            if (g.out1) ((float*)g.out1)[off] = snake_f(v, sn_a[j], sn_ia[j]);
          } else {
            ((T*)g.out0)[off] = Cvt<T>::to(act_epi(v, EPI));
          }
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_tile(const GemmArgs& g, int epi, hipStream_t st) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  dim3 grid(tiles), block(WM * WN * 64);
#define FOLEY_CASE(E)                                                                        \
  case E:                                                                                    \
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, E>), grid, block, 0, st, g);          \
    break;
  switch (epi) {
    FOLEY_CASE(EPI_STORE_F32)
    FOLEY_CASE(EPI_STORE_T)
    FOLEY_CASE(EPI_SILU_T)
    FOLEY_CASE(EPI_GELU_T)
    FOLEY_CASE(EPI_GATE_RES)
    case EPI_SILUGATE_T:
      if constexpr ((BN / WN) % 64 == 0) {
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, EPI_SILUGATE_T>), grid, block, 0, st, g);
      } else {
        return foley_set_err("gated epilogue needs a 64-wide wave tile", __FILE__, __LINE__);
      }
      break;
    case EPI_DAC:
      if constexpr (sizeof(T) == 4) {
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, EPI_DAC>), grid, block, 0, st, g);
      } else {
        return foley_set_err("DAC epilogue is fp32 only", __FILE__, __LINE__);
      }
      break;
    default:
      return foley_set_err("unknown GEMM epilogue", __FILE__, __LINE__);
  }
#undef FOLEY_CASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T>
int launch_typed(const GemmArgs& g, int epi, int tile, hipStream_t st) {
  constexpr int BK = 8 * Frag<T>::EPC;
  if (g.K % BK || g.tapC % BK || g.taps * g.tapC != g.K || g.lda % Frag<T>::EPC)
    return foley_set_err("GEMM: K / tap width / lda must be multiples of the 128-byte K-slice", __FILE__, __LINE__);
  if (((uintptr_t)g.A | (uintptr_t)g.W) & 15)
    return foley_set_err("GEMM: operands must be 16-byte aligned", __FILE__, __LINE__);
  if (tile == 0) {
    // pick the largest tile that still yields >= ~3/4 of a wave of workgroups over 256 CUs
    auto nblk = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
    if (g.N <= 64 && epi != EPI_SILUGATE_T) tile = nblk(128, 64) >= 192 ? 4 : 3;
    else if (nblk(128, 128) >= 192) tile = 1;
    else if (nblk(64, 128) >= 192 || epi == EPI_SILUGATE_T) tile = 2;
    else tile = 3;
  }
  switch (tile) {
    case 1: return launch_tile<T, 128, 128, 2, 2>(g, epi, st);
    case 2: return launch_tile<T, 64, 128, 2, 2>(g, epi, st);
    case 3: return launch_tile<T, 64, 64, 2, 2>(g, epi, st);
    case 4: return launch_tile<T, 128, 64, 4, 1>(g, epi, st);
  }
  return foley_set_err("GEMM: bad tile id", __FILE__, __LINE__);
}

}  // namespace

int launch_gemm(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return 0;
  if (dtype == FOLEY_F32) return launch_typed<float>(g, epi, tile, st);
  if (dtype == FOLEY_BF16) return launch_typed<bf16_t>(g, epi, tile, st);
  return foley_set_err("GEMM: unsupported operand dtype", __FILE__, __LINE__);
}
