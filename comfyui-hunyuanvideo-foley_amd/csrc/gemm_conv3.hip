// Tap-fused channels-last conv k=3 (pad 1, dilation 1) as an MFMA GEMM - the DiT's
// ChannelLastConv1d (mlp_layers.py:104-110: single-block linear1 and the ConvMLP w1/w3/w2), which
// carry two thirds of the model's FLOPs.
//
// The generic engine (gemm_impl.h) walks K tap-major and therefore streams the activation tile once
// per tap.  Here K is walked CHANNEL-chunk major: per 128-byte channel chunk the workgroup stages
// the BM+2 activation rows ONCE (one halo row on each side) plus the three tap slices of the weight
// tile, and runs all three taps' MFMAs from that stage, the tap being nothing but a row offset
// (0/1/2) into the staged activation rows.  Per MFMA this moves a third less through L2/LDS and
// there is one barrier per three tap-slices.  Rows whose neighbour lies outside their sequence
// (first / last token of a clip) get a zeroed fragment for that tap (the conv's zero padding).
// Same register-ring / double-buffered-LDS structure and the same fused epilogues as gemm_impl.h.
#include "gemm_common.h"

namespace {

template <typename T, int BM, int BN, int WM, int WN, int NS, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_conv3_kernel(const GemmPair pr) {
  const int sel = (int)blockIdx.x >= pr.tiles0 ? 1 : 0;
  const GemmArgs& g = pr.g[sel];
  constexpr int NT = WM * WN * 64;
  constexpr int EPC = Frag<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int AROWS = BM + 2, WROWS = 3 * BN;
  constexpr int RA = (AROWS * 8 + NT - 1) / NT;  // 16-byte chunks per thread and stage
  constexpr int RB = (WROWS * 8) / NT;
  static_assert((WROWS * 8) % NT == 0, "bad tile");
  constexpr int STAGE = (AROWS + WROWS) * LDS_PITCH;
  static_assert(EPI != EPI_SILUGATE_T || (FN % 2 == 0), "gated epilogue needs fragment pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 stages

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = (int)blockIdx.x - (sel ? pr.tiles0 : 0);
  {
    const int nwg = tiles_m * tiles_n * (EPI == EPI_GATE_RES ? g.ksplit : 1);
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int ks = 0;
  if constexpr (EPI == EPI_GATE_RES) {
    ks = bid % g.ksplit;
    bid /= g.ksplit;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  tl_stamp(g, 0);
  const int wm = wave / WN, wn = wave % WN;
  const int C = g.tapC;  // channels per tap; K = 3*C; activation rows are [row][C]

  // loader descriptors: chunk id c -> (row = c / 8, 16-byte column = c % 8)
  const T* ap[RA];
  int a_lds[RA];
  bool a_use[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int c = tid + i * NT;
    a_use[i] = c < AROWS * 8;
    const int rl = a_use[i] ? (c >> 3) : 0;           // staged row 0 .. BM+1  <->  activation row m0 - 1 + rl
    int r = m0 - 1 + rl;
    r = r < 0 ? 0 : (r >= g.M ? g.M - 1 : r);           // clamped: out-of-range rows are never consumed un-zeroed
    ap[i] = (const T*)g.A + (long)r * g.lda + (c & 7) * EPC;
    a_lds[i] = rl * LDS_PITCH + (c & 7) * 16;
  }
  const T* wp[RB];
  int w_lds[RB];
  bool w_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int c = tid + i * NT;
    const int rl = c >> 3;                 // 0 .. 3*BN-1 : tap = rl / BN, weight row n0 + rl % BN
    const int tap = rl / BN, nl = rl - tap * BN;
    const int n = n0 + nl;
    w_ok[i] = n < g.N;
    wp[i] = (const T*)g.W + (long)(w_ok[i] ? n : 0) * g.K + (long)tap * C + (c & 7) * EPC;
    w_lds[i] = (AROWS + rl) * LDS_PITCH + (c & 7) * 16;
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
  // zero-padding flags of this lane's fragment rows: first / last token of its sequence
  bool zl[FM], zr[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int r = m0 + wm * TM + i * 32 + fi;
    const int q = r % g.segV;
    zl[i] = q == 0;
    zr[i] = q == g.segV - 1;
  }

  int kc_begin = 0, nkc = C / BK;  // channel chunks
  if constexpr (EPI == EPI_GATE_RES) {
    const int tot = nkc;
    kc_begin = (int)((long)tot * ks / g.ksplit);
    nkc = (int)((long)tot * (ks + 1) / g.ksplit) - kc_begin;
  }
  int ld_c0 = kc_begin * BK;

#define FOLEY_GLOAD3(slot)                                                                  \
  {                                                                                         \
    _Pragma("unroll") for (int i = 0; i < RA; ++i) ra[slot][i] = *(const u32x4*)(ap[i] + ld_c0); \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) rw[slot][i] = *(const u32x4*)(wp[i] + ld_c0); \
    ld_c0 += BK;                                                                            \
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkc) FOLEY_GLOAD3(s);

  for (int kt0 = 0; kt0 < nkc; kt0 += NS) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int kt = kt0 + j;
      if (kt < nkc) {
        if (kt + NS - 1 < nkc) FOLEY_GLOAD3((j + NS - 1) % NS);
        unsigned char* St = lds + (kt & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < RA; ++i)
          if (a_use[i]) *(u32x4*)(St + a_lds[i]) = ra[j][i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(u32x4*)(St + w_lds[i]) = w_ok[i] ? rw[j][i] : zero4;
        __syncthreads();
        if (kt == 0) tl_stamp(g, 1);
        const unsigned char* As = St;
        const unsigned char* Ws = St + AROWS * LDS_PITCH;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          if constexpr (sizeof(T) == 4) {
            float a[FM][16], b[FN][16];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
              const unsigned char* p = As + (wm * TM + i * 32 + fi + tap) * LDS_PITCH + kh * 64;
              const bool z = (tap == 0 && zl[i]) || (tap == 2 && zr[i]);
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 v = *(const f32x4*)(p + c * 16);
                a[i][c * 4 + 0] = z ? 0.f : v[0]; a[i][c * 4 + 1] = z ? 0.f : v[1];
                a[i][c * 4 + 2] = z ? 0.f : v[2]; a[i][c * 4 + 3] = z ? 0.f : v[3];
              }
            }
#pragma unroll
            for (int jj = 0; jj < FN; ++jj) {
              const unsigned char* p = Ws + (tap * BN + wn * TN + jj * 32 + fi) * LDS_PITCH + kh * 64;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 v = *(const f32x4*)(p + c * 16);
                b[jj][c * 4 + 0] = v[0]; b[jj][c * 4 + 1] = v[1]; b[jj][c * 4 + 2] = v[2]; b[jj][c * 4 + 3] = v[3];
              }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
              for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jj = 0; jj < FN; ++jj)
                  acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[jj][s], acc[i][jj], 0, 0, 0);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              bf16x8 a[FM], b[FN];
#pragma unroll
              for (int i = 0; i < FM; ++i) {
                const u32x4 v = *(const u32x4*)(As + (wm * TM + i * 32 + fi + tap) * LDS_PITCH + s * 32 + kh * 16);
                const bool z = (tap == 0 && zl[i]) || (tap == 2 && zr[i]);
                const u32x4 vz = z ? zero4 : v;
                a[i] = __builtin_bit_cast(bf16x8, vz);
              }
#pragma unroll
              for (int jj = 0; jj < FN; ++jj)
                b[jj] = *(const bf16x8*)(Ws + (tap * BN + wn * TN + jj * 32 + fi) * LDS_PITCH + s * 32 + kh * 16);
#pragma unroll
              for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jj = 0; jj < FN; ++jj)
                  acc[i][jj] = mfma16<T>(a[i], b[jj], acc[i][jj]);
            }
          }
        }
      }
    }
  }
#undef FOLEY_GLOAD3
  tl_stamp(g, 2);
  if (g.vec_out) gemm_epilogue_lds<T, EPI, BM, BN, WM, WN>(g, acc, lds, m0, n0, ks);
  else gemm_epilogue<T, EPI, FM, FN, TM, TN>(g, acc, m0, n0, wm, wn, fi, kh, ks);
  tl_stamp(g, 3);
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int EPI>
int launch_c3(const GemmArgs& g, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)(BM + 2 + 3 * BN) * LDS_PITCH;
  auto k = gemm_conv3_kernel<T, BM, BN, WM, WN, NS, EPI>;
  static std::atomic<unsigned long long> raised{0};
  if (lds > 64 * 1024) {
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g;
  pr.tiles0 = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * (EPI == EPI_GATE_RES ? g.ksplit : 1);
  FOLEY_LAUNCH(k, dim3(pr.tiles0), dim3(WM * WN * 64), lds, st, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int NS>
int launch_c3_epi(const GemmArgs& g, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_STORE_F32: return launch_c3<T, BM, BN, WM, WN, NS, EPI_STORE_F32>(g, st);
    case EPI_GATE_RES: return launch_c3<T, BM, BN, WM, WN, NS, EPI_GATE_RES>(g, st);
    case EPI_SILUGATE_T:
      if constexpr ((BN / WN) % 64 == 0) return launch_c3<T, BM, BN, WM, WN, NS, EPI_SILUGATE_T>(g, st);
      else return foley_set_err("conv3: gated epilogue needs a 64-wide wave tile", __FILE__, __LINE__);
  }
  return foley_set_err("conv3: unsupported epilogue", __FILE__, __LINE__);
}

}  // namespace

// tile: 1 = 128x128 (8 waves), 3 = 64x64 (4 waves); ksplit must already be resolved (>= 1)
int launch_gemm_conv3(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t st) {
  if (g.taps != 3 || g.dil != 1 || g.tap0 != -1 || g.segV != g.segS || g.lda != g.tapC)
    return foley_set_err("conv3: not a k=3 / pad 1 / dilation 1 channels-last conv", __FILE__, __LINE__);
  if (dtype == FOLEY_BF16) {
    if (g.tapC % 64) return foley_set_err("conv3: channel count must be a multiple of 64", __FILE__, __LINE__);
    if (tile == 1) return launch_c3_epi<bf16_t, 128, 128, 4, 2, 3>(g, epi, st);
    return launch_c3_epi<bf16_t, 64, 64, 2, 2, 3>(g, epi, st);
  }
  if (dtype == FOLEY_F16) {
    if (g.tapC % 64) return foley_set_err("conv3: channel count must be a multiple of 64", __FILE__, __LINE__);
    if (tile == 1) return launch_c3_epi<f16_t, 128, 128, 4, 2, 3>(g, epi, st);
    return launch_c3_epi<f16_t, 64, 64, 2, 2, 3>(g, epi, st);
  }
  if (dtype == FOLEY_F32) {
    if (g.tapC % 32) return foley_set_err("conv3: channel count must be a multiple of 32", __FILE__, __LINE__);
    if (tile == 1) return launch_c3_epi<float, 128, 128, 4, 2, 3>(g, epi, st);
    return launch_c3_epi<float, 64, 64, 2, 2, 3>(g, epi, st);
  }
  return foley_set_err("conv3: unsupported dtype", __FILE__, __LINE__);
}
