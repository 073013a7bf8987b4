// Pieces shared by the GEMM mainloop variants (gemm.hip, gemm_conv3.hip): fragment traits, fused
// epilogues, the two-problem kernel argument.
#pragma once
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

// Fused epilogues shared by both mainloops.  C/D layout of the 32x32 MFMA: column = lane & 31,
// row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
template <typename T, int EPI, int FM, int FN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[FM][FN], int m0, int n0, int wm,
                                              int wn, int fi, int kh, int ks) {
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
          if (g.bias && ks == 0) v += g.bias[col];
          if constexpr (EPI == EPI_STORE_F32) {
            if (rbp) v += rbp[col];
            ((float*)g.out0)[off] = v;
          } else if constexpr (EPI == EPI_GATE_RES) {
            float* x = (float*)g.out0;
            if (g.ksplit > 1) unsafeAtomicAdd(x + off, v * rbp[col]);  // global_atomic_add_f32
            else x[off] = x[off] + v * rbp[col];
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

struct GemmPair {
  GemmArgs g[2];
  int tiles0;  // workgroups belonging to g[0]; the rest run g[1]
};

}  // namespace
