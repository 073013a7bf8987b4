// Pieces shared by the GEMM mainloop variants (gemm_impl.h, gemm_ws_impl.h, gemm_conv3.hip): fragment traits, fused
// epilogues, the two-problem kernel argument.
#pragma once
#include <type_traits>

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
template <> struct Frag<f16_t> {
  static constexpr int EPC = 8;
};

// dbg_mode 0: 4 stamps per workgroup.  dbg_mode 1: workgroup 0 adds its prologue / K-loop / epilogue
// ticks and a launch count to dbg[0..3] (totals over every GEMM of a forward pass).
__device__ __forceinline__ void tl_stamp(const GemmArgs& g, int slot) {
  if (!g.dbg || threadIdx.x != 0 || (g.dbg_mode & 0xff) >= 2) return;
  const long long t = wall_clock64();
  if ((g.dbg_mode & 0xff) == 0) {
    g.dbg[(long)blockIdx.x * 4 + slot] = t;
    if (blockIdx.x == 0 && (slot == 0 || slot == 3)) g.dbg[4 * 8000 + (slot ? 1 : 0)] = (long long)__builtin_readcyclecounter();
  } else if (blockIdx.x == 0) {
    unsigned long long* d = (unsigned long long*)g.dbg;
    if (slot > 0) atomicAdd(d + slot - 1, (unsigned long long)t);
    if (slot < 3) atomicAdd(d + slot, 0ull - (unsigned long long)t);
    if (slot == 3) atomicAdd(d + 3, 1ull);
  }
}

__device__ __forceinline__ float act_epi(float v, int epi, int gelu_erf = 0) {
  if (epi == EPI_SILU_T) return silu_f(v);
  if (epi == EPI_GELU_T) return gelu_erf ? gelu_erf_f(v) : gelu_tanh_f(v);
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
          if (g.bias && ks == 0 && !(EPI == EPI_GATE_RES && g.partials && g.ksplit > 1)) v += g.bias[col];
          if constexpr (EPI == EPI_STORE_F32) {
            if (rbp) v += rbp[col];
            ((float*)g.out0)[off] = v;
          } else if constexpr (EPI == EPI_GATE_RES) {
            float* x = (float*)g.out0;
            if (g.ksplit > 1 && g.partials) {
              const long po = ks * g.partial_stride + (long)row * g.N + col;
              if (g.partial_half) ((T*)g.partials)[po] = Cvt<T>::to(v);
              else g.partials[po] = v;
            }
            else if (g.ksplit > 1) unsafeAtomicAdd(x + off, v * rbp[col]);  // global_atomic_add_f32
            else x[off] = x[off] + v * rbp[col];
          } else if constexpr (EPI == EPI_DAC) {
            if (g.res) v += g.res[off];
            if (g.out0) ((float*)g.out0)[off] = v;
            if (g.out1) ((float*)g.out1)[off] = snake_f(v, sn_a[j], sn_ia[j]);
          } else {
            ((T*)g.out0)[off] = Cvt<T>::to(act_epi(v, EPI, g.gelu_erf));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Vector epilogue: the accumulator tile goes through LDS once (the K-loop stages are dead by then)
// so that every lane owns 16 contiguous output bytes of one row - global stores (and the residual
// read-modify-write) become dwordx4 instead of 32 scalar dword accesses per fragment, which is what
// the scalar epilogue is bound by (store issue, not bytes).  All global loads of a thread (residual,
// gate rows) are issued before the first store.  Used when GemmArgs::vec_out is set by the launcher
// (plain row-major output, 16-byte aligned operands); everything else takes gemm_epilogue above.
struct RbCursor {   // rb_row() for rows that advance by a fixed stride, without per-row divisions
  const float* p;
  long ld;
  int mode, rpc, L, cfg, rc, l, Ls, per, dense_from, dense_base;
  float scale;
  __device__ __forceinline__ void init(const RowBcast& b, int row) {
    p = b.p;
    if (p && b.step_ptr) p += (long)(*b.step_ptr) * b.step_stride;
    ld = b.ld; mode = p ? b.mode : 0; rpc = b.rows_per_cfg; L = b.L; Ls = b.Ls; scale = b.scale; per = b.per; dense_from = b.dense_from; dense_base = b.dense_base;
    cfg = 0; rc = 0; l = 0;
    if (mode != 0) { cfg = row / rpc; rc = row - cfg * rpc; l = row % L; }
  }
  __device__ __forceinline__ const float* row_ptr() const {
    if (mode == 1) return p + ((long)cfg * L + l) * ld;
    if (mode == 2) {   // common.h RowBcast mode 2
      const int s = rb_nearest_exact(l, scale, Ls);
      const int idx = cfg < dense_from ? cfg * per + (s & (per - 1)) : dense_base + (cfg - dense_from) * Ls + s;
      return p + (long)idx * ld;
    }
    return p;
  }
  __device__ __forceinline__ void advance(int rows) {
    if (mode == 0) return;
    l += rows; while (l >= L) l -= L;
    rc += rows; while (rc >= rpc) { rc -= rpc; ++cfg; }
  }
};

template <typename OutT> struct VecStore;
template <> struct VecStore<float> {
  static constexpr int CP = 4;
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 w = {v[0], v[1], v[2], v[3]};
    *(f32x4*)p = w;
  }
};
template <> struct VecStore<bf16_t> {
  static constexpr int CP = 8;
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    u32x4 w;
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = pack_bf16x2(v[2 * u], v[2 * u + 1]);
    *(u32x4*)p = w;
  }
};
template <> struct VecStore<f16_t> {
  static constexpr int CP = 8;
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[8]) {
    u32x4 w;
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = pack_f16x2(v[2 * u], v[2 * u + 1]);
    *(u32x4*)p = w;
  }
};
template <typename T, int EPI> struct EpiOutT { using type = T; };
template <typename T> struct EpiOutT<T, EPI_STORE_F32> { using type = float; };
template <typename T> struct EpiOutT<T, EPI_GATE_RES> { using type = float; };
template <typename T> struct EpiOutT<T, EPI_DAC> { using type = float; };

// XW: helper waves (ids >= WM*WN, e.g. the loader waves of the wave-specialised mainloop once their job is
// done) that hold no accumulators but take their share of the LDS -> global passes.
template <typename T, int EPI, int BM, int BN, int WM, int WN, int XW = 0>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& g, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                                  unsigned char* lds_raw, int m0, int n0, int ks, int vtid = -1) {
  constexpr int NT = (WM * WN + XW) * 64, TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  using OutT = typename EpiOutT<T, EPI>::type;
  constexpr int CP = VecStore<OutT>::CP;                    // output elements per lane and pass
  constexpr int OBN = EPI == EPI_SILUGATE_T ? BN / 2 : BN;  // output columns of the tile
  constexpr int TPR = OBN / CP, RP = NT / TPR, PASSES = BM / RP;
  static_assert(NT % TPR == 0 && BM % RP == 0 && PASSES >= 1, "tile / epilogue mismatch");
  float* tile = (float*)lds_raw;
  // vtid: the thread's id inside a VIRTUAL tile (gemm_wide_impl.h runs a 256x256 tile's epilogue as two 256x128 passes whose
  // writer waves are renumbered 0 .. WM*WN-1 and whose other waves help as XW waves); -1: the hardware thread id
  const int tid = vtid >= 0 ? vtid : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, fi = lane & 31, kh = lane >> 5;
  const bool estamp = g.dbg && (g.dbg_mode & 0xff) == 4 && tid == 0;   // tools/gemm_timeline.py --epilogue
  if (estamp) g.dbg[(long)blockIdx.x * 4 + 0] = wall_clock64();
  const int tr = tid / TPR, tc = (tid % TPR) * CP;   // row inside a pass, output column inside the tile
  int ca, cb = 0, gcol, ncheck;                      // tile columns to read, global output column
  if constexpr (EPI == EPI_SILUGATE_T) {
    ca = (tc >> 5) * 64 + (tc & 31);
    cb = ca + 32;
    gcol = (n0 >> 1) + tc;
    ncheck = n0 + (tc >> 5) * 64;
  } else {
    ca = tc;
    gcol = n0 + tc;
    ncheck = gcol;
  }
  const bool col_ok = ncheck < g.N;
  // the bias is requested before the barriers and the LDS transpose (a dependent global load after them otherwise)
  float bias_a[CP], bias_b[CP];
#pragma unroll
  for (int u = 0; u < CP; ++u) bias_a[u] = bias_b[u] = 0.f;
  const bool bias_used = !(EPI == EPI_GATE_RES && g.ksplit > 1);   // deferred split-K leaves the bias to the LayerNorm
  if (g.bias && col_ok && bias_used) {
#pragma unroll
    for (int u = 0; u < CP; u += 4) {
      const f32x4 v = *(const f32x4*)(g.bias + n0 + ca + u);
      bias_a[u] = v[0]; bias_a[u + 1] = v[1]; bias_a[u + 2] = v[2]; bias_a[u + 3] = v[3];
      if constexpr (EPI == EPI_SILUGATE_T) {
        const f32x4 w = *(const f32x4*)(g.bias + n0 + cb + u);
        bias_b[u] = w[0]; bias_b[u + 1] = w[1]; bias_b[u + 2] = w[2]; bias_b[u + 3] = w[3];
      }
    }
  }
  __syncthreads();   // every wave is done reading the last K-slice
  if (estamp) g.dbg[(long)blockIdx.x * 4 + 1] = wall_clock64();
  if (XW == 0 || wave < WM * WN) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tile[(wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * BN + wn * TN + j * 32 + fi] = acc[i][j][e];
  }
  __syncthreads();
  if (estamp) g.dbg[(long)blockIdx.x * 4 + 2] = wall_clock64();
  OutT* out = (OutT*)g.out0 + g.out_shift + gcol;
  const int row0 = m0 + tr;

  if constexpr (EPI == EPI_GATE_RES) {
    if (g.ksplit > 1) {   // deferred split-K: raw partial product of this K range
      if (g.partial_half) {   // slabs in the operand type: 8-byte stores of four rounded partials
        if constexpr (sizeof(T) == 2) {
          T* ps = (T*)g.partials + ks * g.partial_stride + gcol;
#pragma unroll
          for (int p = 0; p < PASSES; ++p) {
            const int row = row0 + p * RP;
            if (!(col_ok && row < g.M)) continue;
            const f32x4 a = *(const f32x4*)(tile + (p * RP + tr) * BN + ca);
            uint2 w;
            w.x = pack_h2<T>(a[0], a[1]);
            w.y = pack_h2<T>(a[2], a[3]);
            *(uint2*)(ps + (long)row * g.N) = w;
          }
        }
        return;
      }
      float* ps = g.partials + ks * g.partial_stride + gcol;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = row0 + p * RP;
        if (!(col_ok && row < g.M)) continue;
        *(f32x4*)(ps + (long)row * g.N) = *(const f32x4*)(tile + (p * RP + tr) * BN + ca);
      }
      return;
    }
  }

  if constexpr (EPI == EPI_GATE_RES || EPI == EPI_STORE_F32) {
    RbCursor rc;
    rc.init(g.rb, row0);
    constexpr int PB = PASSES > 4 ? 4 : PASSES;   // passes whose global reads are in flight together (register budget)
    static_assert(PASSES % PB == 0, "pass batching");
#pragma unroll
    for (int p0 = 0; p0 < PASSES; p0 += PB) {
      f32x4 xv[PB], gv[PB];
#pragma unroll
      for (int p = 0; p < PB; ++p) {   // all global reads of the batch first
        const int row = row0 + (p0 + p) * RP;
        const bool ok = col_ok && row < g.M;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        xv[p] = z;
        gv[p] = z;
        if (ok) {
          if constexpr (EPI == EPI_GATE_RES) xv[p] = *(const f32x4*)(out + (long)row * g.out_row);
          if (rc.p) gv[p] = *(const f32x4*)(rc.row_ptr() + gcol);
        }
        rc.advance(RP);
      }
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const int row = row0 + (p0 + p) * RP;
        if (!(col_ok && row < g.M)) continue;
        const f32x4 a = *(const f32x4*)(tile + ((p0 + p) * RP + tr) * BN + ca);
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float t = a[u] + bias_a[u];
          v[u] = EPI == EPI_GATE_RES ? xv[p][u] + t * gv[p][u] : t + gv[p][u];
        }
        VecStore<float>::store((float*)out + (long)row * g.out_row, v);
      }
    }
  } else if constexpr (EPI == EPI_DAC) {
    // DAC residual unit tails (dac.py:28-44): v = acc + bias (+ res); out0 = v; out1 = snake(v) with the consumer's alpha - four
    // consecutive channels per lane, dwordx4 loads / stores (the scalar form issues 3 x 64 four-byte accesses per lane; the
    // k=1 convs of the narrow stages are bound by exactly that)
    float* const o0 = (float*)g.out0;
    float* const o1 = (float*)g.out1;
    const float* const rs = g.res;
    f32x4 al = {1.f, 1.f, 1.f, 1.f}, ia;
    if (o1 && col_ok) al = *(const f32x4*)(g.alpha + gcol % g.alphaC);
#pragma unroll
    for (int u = 0; u < 4; ++u) ia[u] = 1.0f / (al[u] + 1e-9f);
    const bool plain_out = g.osegV >= g.M;
    // element offset of (row, gcol) in the outputs, or -1: row beyond M / outside its clip (transposed conv edges)
    auto out_off = [&](int row) -> long {
      if (!(col_ok && row < g.M)) return -1;
      if (plain_out) return (long)row * g.out_row + g.out_shift + gcol;
      const int b = row / g.osegV, q = row - b * g.osegV;
      const long rel = (long)q * g.out_row + g.out_shift + gcol;
      if (g.out_check && (rel < 0 || rel >= g.out_seg)) return -1;
      return (long)b * g.out_seg + rel;
    };
    constexpr int PB = PASSES > 4 ? 4 : PASSES;
    static_assert(PASSES % PB == 0, "pass batching");
#pragma unroll
    for (int p0 = 0; p0 < PASSES; p0 += PB) {
      f32x4 rv[PB];
      long off[PB];
#pragma unroll
      for (int p = 0; p < PB; ++p) {   // the batch's residual reads first
        off[p] = out_off(row0 + (p0 + p) * RP);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        rv[p] = (rs && off[p] >= 0) ? *(const f32x4*)(rs + off[p]) : z;
      }
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        if (off[p] < 0) continue;
        const f32x4 a = *(const f32x4*)(tile + ((p0 + p) * RP + tr) * BN + ca);
        float v[4], sn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = a[u] + bias_a[u] + rv[p][u];
          sn[u] = snake_f(v[u], al[u], ia[u]);
        }
        if (o0) VecStore<float>::store(o0 + off[p], v);
        if (o1) VecStore<float>::store(o1 + off[p], sn);
      }
    }
  } else {
    // ERF: the exact GELU of the conditioning encoders (GemmArgs::gelu_erf).  The flag is tested ONCE, outside the pass loop: as
    // a per-element select the compiler evaluated erff() next to the fast form for every output of the DiT's fc1 GEMM
    // (bs=8: 113 -> 127 us per launch).
    auto passes = [&](auto erf_c) {
      constexpr bool ERF = decltype(erf_c)::value;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = row0 + p * RP;
        if (!(col_ok && row < g.M)) continue;
        const float* src = tile + (p * RP + tr) * BN;
        float v[CP];
#pragma unroll
        for (int u = 0; u < CP; u += 4) {
          const f32x4 a = *(const f32x4*)(src + ca + u);
          if constexpr (EPI == EPI_SILUGATE_T) {
            const f32x4 b = *(const f32x4*)(src + cb + u);
#pragma unroll
            for (int w = 0; w < 4; ++w) v[u + w] = silu_o<OutT>(a[w] + bias_a[u + w]) * (b[w] + bias_b[u + w]);
          } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float t = a[w] + bias_a[u + w];
              if constexpr (EPI == EPI_GELU_T) v[u + w] = ERF ? gelu_erf_o<OutT>(t) : gelu_o<OutT>(t);
              else v[u + w] = EPI == EPI_SILU_T ? silu_o<OutT>(t) : t;
            }
          }
        }
        VecStore<OutT>::store(out + (long)row * g.out_row, v);
      }
    };
    if constexpr (EPI == EPI_GELU_T) {
      if (g.gelu_erf) passes(std::integral_constant<bool, true>{});
      else passes(std::integral_constant<bool, false>{});
    } else {
      passes(std::integral_constant<bool, false>{});
    }
  }
}

// ---------------------------------------------------------------------------------------------
// EPI_QKV_SPLIT: the fused q/k/v (or cross-attention q) projection never materialises [M, nK*H*128].
// A 128-column tile is exactly one head of one operand, so after the LDS transpose every output row
// is a complete head vector: q/k get RMSNorm (norm_layers.py:36-52 / nn.RMSNorm) and interleaved
// RoPE (attn_layers.py:112-146) and leave as 16-byte stores into [clip, H, S_tot, 128]; V leaves
// transposed [clip, H, 128, pitch] for the bf16 attention kernel, 8 tokens (16 bytes) per store on
// destination-aligned groups.  Same math as qkv_split_kernel (rowops.hip), which stays for callers
// that have the projection in memory.
//
// ATTN (EPI_QKV_ATTN, 16-bit operands, nK = 1): the cross-attention q projection carries the attention itself
// (hifi_foley.py:271-319: q = [v; a] against the <= 96 cached text keys of the clip's CFG half).  The tile's rotated q rows
// stay in LDS as a swizzled 16-bit image, the head's K and V^T images (requested as 16-byte register loads BEFORE the
// transpose barriers, so their latency hides under the head-split math) join them, and one wave per 32-query block runs
// QK^T -> online softmax -> PV from LDS fragments and writes the attention output rows [M, H*128] - the q tensor, the
// attention launch and the boundary in front of it are gone (QkvSplitArgs::attn_*).
template <typename T, int BM, int BN, int WM, int WN, int XW = 0, bool EARLY = true, bool ATTN = false>
__device__ __forceinline__ void gemm_epilogue_qkv(const GemmArgs& g, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                                  unsigned char* lds_raw, int m0, int n0, int vtid = -1) {
  static_assert(BN == 128, "one head per tile");
  constexpr int NT = (WM * WN + XW) * 64, TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int CP = VecStore<T>::CP, TPR = 128 / CP, RP = NT / TPR, PASSES = BM / RP;
  static_assert(!ATTN || (sizeof(T) == 2 && EARLY && BM / 32 <= NT / 64), "fused cross attention: 16-bit operands, one wave per query block");
  // LDS map of the fused form (the ring is dead): fp32 tile | Q image [BM][256 B] | K image [96][256 B] | V^T image [128][256 B];
  // 16-byte chunk c of image row r sits at position c ^ (r & 15): every ds_read_b128 lane group meets 16 distinct positions
  // a tile that straddles the CFG halves keeps its second text set too: K image over the (dead) fp32 tile, V^T image behind
  constexpr int OFF_Q = BM * BN * 4, OFF_K = OFF_Q + BM * 256, OFF_V = OFF_K + 96 * 256, OFF_K1 = 0, OFF_V1 = OFF_V + 128 * 256;
  static_assert(!ATTN || 96 * 256 <= BM * BN * 4, "second K image aliases the fp32 tile");
  constexpr int KVC = (96 * 16 + 128 * 12 + NT - 1) / NT;   // 16-byte chunks of one K / V^T set per thread
  static_assert(NT % TPR == 0 && BM % RP == 0 && PASSES >= 1, "tile / epilogue mismatch");
  static_assert(TPR == 16 || TPR == 32, "head-split epilogue: 16 or 32 lanes per row");
  const QkvSplitArgs& q = g.qs;
  float* tile = (float*)lds_raw;
  const int tid = vtid >= 0 ? vtid : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;   // vtid: see gemm_epilogue_lds
  const int wm = wave / WN, wn = wave % WN, fi = lane & 31, kh = lane >> 5;
  const bool estamp = g.dbg && (g.dbg_mode & 0xff) == 4 && tid == 0;   // tools/gemm_timeline.py --epilogue
  if (estamp) g.dbg[(long)blockIdx.x * 4 + 0] = wall_clock64();
  const int hidx = n0 >> 7;
  const int o = hidx / q.H, h = hidx - o * q.H;   // operand (q, k, v), head
  const bool live = n0 < g.N;                     // workgroup-uniform
  const bool vtrans = q.vt_pitch > 0 && o == q.nK - 1;
  // per-operand descriptors: all three read at constant kernel-argument offsets, then selected (an index
  // computed at run time makes every use a separate dependent scalar load)
  const float* const gp = o == 0 ? q.gain[0] : (o == 1 ? q.gain[1] : q.gain[2]);
  const int* const pos = o == 0 ? q.pos[0] : (o == 1 ? q.pos[1] : q.pos[2]);
  const float* const rcos = o == 0 ? q.rcos[0] : (o == 1 ? q.rcos[1] : q.rcos[2]);
  const float* const rsin = o == 0 ? q.rsin[0] : (o == 1 ? q.rsin[1] : q.rsin[2]);
  void* const dstp = o == 0 ? q.dst[0] : (o == 1 ? q.dst[1] : q.dst[2]);
  // ---- every global read of the q / k path is requested HERE, before the barriers and the LDS transpose:
  // bias, gain and the rotation rows of all passes (one load level when the caller supplies rows gathered
  // per token, QkvSplitArgs::rcos / rsin; position -> table row, two dependent levels, otherwise).
  const int tr = tid / TPR, tc = (tid % TPR) * CP;
  float bias[CP], gain[CP];
  int pl[PASSES];
  long doff[PASSES];   // destination element offset of the pass's row; < 0: row beyond M (no store)
  float cs[PASSES][CP / 2], sn[PASSES][CP / 2];
  const bool qk = live && !vtrans;
  // EARLY: request these reads before the LDS transpose.  Kernels of up to 512 threads (register cap 256 per lane)
  // do; the 768-thread tiles (cap 168) cannot hold them next to the accumulators without spilling and request
  // them after it.
  // Scalars of the row loop, read once and pinned: left to itself the compiler re-loads each of them from the
  // kernel-argument segment in every pass (a dependent s_load + wait each, ~100 of them in this epilogue).
  int qH = q.H, qS = q.S_tot, qoff = q.tok_off, qL = q.L, gM = g.M;
  float qeps = q.eps;
  asm volatile("" : "+s"(qH), "+s"(qS), "+s"(qoff), "+s"(qL), "+s"(gM), "+s"(qeps));
  auto request = [&]() {
    if (!qk) return;
#pragma unroll
    for (int u = 0; u < CP; u += 4) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
      if (g.bias) bv = *(const f32x4*)(g.bias + n0 + tc + u);
      if (gp) gv = *(const f32x4*)(gp + tc + u);
#pragma unroll
      for (int w = 0; w < 4; ++w) { bias[u + w] = bv[w]; gain[u + w] = gv[w]; }
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int row = m0 + p * RP + tr;
      const int rr = row < gM ? row : gM - 1;
      const int b = rr / qL;
      pl[p] = rr - b * qL;
      doff[p] = row < gM ? (((long)b * qH + h) * qS + qoff + pl[p]) * 128 + tc : -1;
    }
    if (pos) {
      long pp[PASSES];
      const float *ct = q.cos_tab, *st = q.sin_tab;
      if (rcos) {
        ct = rcos; st = rsin;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) pp[p] = (long)pl[p] * 64 + (tc >> 1);
      } else {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) pp[p] = (long)pos[pl[p]] * 64 + (tc >> 1);
      }
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        if constexpr (CP == 8) {
          const f32x4 c4 = *(const f32x4*)(ct + pp[p]), s4 = *(const f32x4*)(st + pp[p]);
#pragma unroll
          for (int w = 0; w < 4; ++w) { cs[p][w] = c4[w]; sn[p][w] = s4[w]; }
        } else {
          cs[p][0] = ct[pp[p]]; cs[p][1] = ct[pp[p] + 1];
          sn[p][0] = st[pp[p]]; sn[p][1] = st[pp[p] + 1];
        }
      }
    }
  };
  if constexpr (EARLY) request();
  // fused cross attention: the K / V^T chunks of the tile's (at most two) text sets, requested here
  u32x4 kv[ATTN ? 2 : 1][ATTN ? KVC : 1];
  int set0 = 0, set1 = 0;
  auto kv_request = [&](int which, int set) {
    if constexpr (ATTN) {
      const T* Kb = (const T*)q.attn_k + ((long)set * qH + h) * q.attn_skv * 128;
      const T* Vb = (const T*)q.attn_vt + ((long)set * qH + h) * 128 * q.attn_pitch;
#pragma unroll
      for (int i = 0; i < KVC; ++i) {
        const int c = tid + NT * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (c < 96 * 16) {
          const int row = c >> 4, ch = c & 15;
          if (row < q.attn_skv) v = *(const u32x4*)(Kb + (long)row * 128 + ch * 8);
        } else if (c < 96 * 16 + 128 * 12) {
          const int cc = c - 96 * 16, d = cc / 12, ch = cc - d * 12;
          v = *(const u32x4*)(Vb + (long)d * q.attn_pitch + ch * 8);
        }
        kv[which][i] = v;
      }
    }
  };
  auto kv_to_lds = [&](int which) {
    if constexpr (ATTN) {
#pragma unroll
      for (int i = 0; i < KVC; ++i) {
        const int c = tid + NT * i;
        if (c < 96 * 16) {
          const int row = c >> 4, ch = c & 15;
          *(u32x4*)(lds_raw + (which ? OFF_K1 : OFF_K) + row * 256 + ((ch ^ (row & 15)) << 4)) = kv[which][i];
        } else if (c < 96 * 16 + 128 * 12) {
          const int cc = c - 96 * 16, d = cc / 12, ch = cc - d * 12;
          *(u32x4*)(lds_raw + (which ? OFF_V1 : OFF_V) + d * 256 + ((ch ^ (d & 15)) << 4)) = kv[which][i];
        }
      }
    }
  };
  if constexpr (ATTN) {
    const int row_last = min(m0 + BM, gM) - 1;
    set0 = (m0 / qL) / q.attn_bdiv;
    set1 = (row_last / qL) / q.attn_bdiv;     // workgroup-uniform; the launcher guarantees set1 <= set0 + 1
    if (live) {
      kv_request(0, set0);
      if (set1 != set0) kv_request(1, set1);
    }
  }
  // transposed-V tiles: the thread's channel bias, requested before the barriers too (it used to be a global load
  // inside the item loop, in front of every item's LDS reads)
  const float vbias = (live && vtrans && g.bias) ? g.bias[n0 + (tid & 127)] : 0.f;
  __syncthreads();
  if (estamp) g.dbg[(long)blockIdx.x * 4 + 1] = wall_clock64();
  if (XW == 0 || wave < WM * WN) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tile[(wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * BN + wn * TN + j * 32 + fi] = acc[i][j][e];
  }
  __syncthreads();
  if (estamp) g.dbg[(long)blockIdx.x * 4 + 2] = wall_clock64();
  if (!live) return;
  if constexpr (!EARLY) request();
  if (!vtrans) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int rl = p * RP + tr;
      float v[CP];
#pragma unroll
      for (int u = 0; u < CP; u += 4) {
        const f32x4 a = *(const f32x4*)(tile + rl * BN + tc + u);
#pragma unroll
        for (int w = 0; w < 4; ++w) v[u + w] = a[w] + bias[u + w];
      }
      if (gp) {
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < CP; ++u) ss += v[u] * v[u];
        ss = row16_sum(ss);                                   // TPR is 16 (bf16) or 32 (fp32) lanes per row
        if constexpr (TPR == 32) ss += __shfl_xor(ss, 16);
        const float rinv = rsqrtf(ss * (1.0f / 128.0f) + qeps);
#pragma unroll
        for (int u = 0; u < CP; ++u) v[u] = v[u] * rinv * gain[u];
      }
      if (pos) {
#pragma unroll
        for (int w = 0; w < CP / 2; ++w) {
          const float y0 = v[2 * w], y1 = v[2 * w + 1];
          v[2 * w] = y0 * cs[p][w] - y1 * sn[p][w];
          v[2 * w + 1] = y1 * cs[p][w] + y0 * sn[p][w];
        }
      }
      if constexpr (ATTN) {
        uint4 w;
        w.x = pack_h2<T>(v[0], v[1]); w.y = pack_h2<T>(v[2], v[3]); w.z = pack_h2<T>(v[4], v[5]); w.w = pack_h2<T>(v[6], v[7]);
        *(uint4*)(lds_raw + OFF_Q + rl * 256 + (((tc >> 3) ^ (rl & 15)) << 4)) = w;
      } else {
        if (doff[p] >= 0) VecStore<T>::store((T*)dstp + doff[p], v);
      }
    }
    if constexpr (ATTN) {
      // ---- cross attention of the tile's 32-query blocks against the text set(s), operands from the LDS images
      const int j = lane & 31;
      const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128)
      const int pi = 16 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);       // A-row j of the score MFMA carries key kt + pi (attention.hip)
      const int Skv = q.attn_skv;
      constexpr int QB = BM / 32;
      static_assert(2 * QB <= NT / 64, "one wave per (query block, text set)");
      const bool two = set1 != set0;             // workgroup-uniform
      if (two) __syncthreads();                  // every thread is done with the fp32 tile (the second K image lands on it)
      kv_to_lds(0);
      if (two) kv_to_lds(1);
      __syncthreads();                           // Q image and the K / V^T images are complete
      {
        const int second = wave >= QB ? 1 : 0;   // waves QB .. 2 QB - 1 serve the second text set of a straddling tile
        const int set = second ? set1 : set0;
        const unsigned char* const Kimg = lds_raw + (second ? OFF_K1 : OFF_K);
        const unsigned char* const Vimg = lds_raw + (second ? OFF_V1 : OFF_V);
        if (wave < (two ? 2 * QB : QB)) {
          const int rl = (wave - second * QB) * 32 + j, row = m0 + rl;
          const bool mine = row < gM && (row / qL) / q.attn_bdiv == set;
          if (__builtin_amdgcn_ballot_w64(mine) != 0) {
            bf16x8 qf[8];
#pragma unroll
            for (int st = 0; st < 8; ++st)
              qf[st] = *(const bf16x8*)(lds_raw + OFF_Q + rl * 256 + (((2 * st + kh) ^ (rl & 15)) << 4));
            f32x16 o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
              for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
            float m_run = -INFINITY, l_run = 0.f;
            for (int kt = 0; kt < Skv; kt += 32) {
              bf16x8 kf[8], vf[2][4];
              const int kr = kt + pi;
#pragma unroll
              for (int st = 0; st < 8; ++st)
                kf[st] = *(const bf16x8*)(Kimg + kr * 256 + (((2 * st + kh) ^ (kr & 15)) << 4));
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                  const int vr = d * 32 + j;
                  vf[u][d] = *(const bf16x8*)(Vimg + vr * 256 + ((((kt >> 3) + 2 * kh + u) ^ (vr & 15)) << 4));
                }
              f32x16 sc;
#pragma unroll
              for (int e = 0; e < 16; ++e) sc[e] = 0.f;
#pragma unroll
              for (int st = 0; st < 8; ++st) sc = mfma16<T>(kf[st], qf[st], sc);
              float mx = -INFINITY;
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                sc[e] = (kt + 16 * kh + e < Skv) ? sc[e] * scale2 : -INFINITY;
                mx = fmaxf(mx, sc[e]);
              }
              mx = xhalf_max(mx);
              const float m_new = fmaxf(m_run, mx);
              float ps = 0.f;
              bf16x8 pb[2];
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(sc[e] - m_new);
                ps += pv;
                pb[e >> 3][e & 7] = to_carrier<T>(pv);
              }
              ps = xhalf_sum(ps);
              // the 64 accumulator rescales only when some query's running maximum moved (never on the first tile: o = 0)
              if (kt > 0 && __builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                  for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
              }
              l_run += ps;
              m_run = m_new;
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int d = 0; d < 4; ++d) o[d] = mfma16<T>(vf[u][d], pb[u], o[d]);
            }
            if (mine) {
              const float inv = 1.0f / l_run;
              T* dst = (T*)q.attn_out + (long)row * ((long)qH * 128) + h * 128 + 4 * kh;
#pragma unroll
              for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                  uint2 w;
                  w.x = pack_h2<T>(o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv);
                  w.y = pack_h2<T>(o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv);
                  *(uint2*)(dst + d * 32 + 8 * g4) = w;
                }
            }
          }
        }
      }
    }
  } else {
    // V^T: walk the clip segments of this row tile; an item = (channel d, 8 destination columns)
    static_assert(NT % 128 == 0, "every thread keeps one channel");
    const int d = tid & 127;
    const float bz = vbias;
    const int row_end = min(m0 + BM, g.M);
    const int b_first = m0 / q.L, b_last = (row_end - 1) / q.L;
    for (int b = b_first; b <= b_last; ++b) {
      const int rs = max(m0, b * q.L), re = min(row_end, (b + 1) * q.L);
      const int count = re - rs, c0 = q.tok_off + (rs - b * q.L);
      const int G0 = c0 >> 3, ngroups = ((c0 + count - 1) >> 3) - G0 + 1;
      for (int it = tid; it < 128 * ngroups; it += NT) {
        const int col0 = (G0 + (it >> 7)) * 8, j0 = col0 - c0;   // d = it & 127 = tid & 127: NT is a multiple of 128
        float v[8];
        bool full = true;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u;
          const bool in = j >= 0 && j < count;
          full = full && in;
          v[u] = in ? tile[(rs - m0 + j) * BN + d] + bz : 0.f;
        }
        T* dst = (T*)dstp + (((long)b * q.H + h) * 128 + d) * q.vt_pitch + col0;
        if constexpr (sizeof(T) == 2) {
          if (full) {
            VecStore<T>::store(dst, v);
            continue;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u >= 0 && j0 + u < count) dst[u] = Cvt<T>::to(v[u]);
      }
    }
  }
}

// Launcher side: can this problem take the vector epilogue?
template <typename T>
inline bool gemm_vec_out_ok(const GemmArgs& g, int epi) {
  if (epi == EPI_DAC) {   // residual + snake, vector form: plain-mapped convs (conv7 / conv1) and the segment-mapped transposed conv
                          // (its N axis = (phase, channel) is contiguous in the output; a 4-channel group never straddles the
                          // clip's ends because the shift and the segment length are multiples of 4)
    if (g.N % 4 || g.out_row % 4 || g.out_shift % 4 || g.out_seg % 4 || g.alphaC % 4 || (g.out1 && !g.alpha)) return false;
    const uintptr_t al = (uintptr_t)g.out0 | (uintptr_t)g.out1 | (uintptr_t)g.res | (uintptr_t)g.alpha | (uintptr_t)g.bias;
    return !(al & 15) && (g.out0 || g.out1);
  }
  if (g.osegV < g.M || g.out_check) return false;
  if (epi == EPI_GATE_RES && g.ksplit > 1)   // split-K: atomics are scalar; deferred partials are vector stores
    return g.partials && g.N % 4 == 0 && !((uintptr_t)g.partials & 15) && g.partial_stride % 4 == 0;
  const bool f32 = epi == EPI_STORE_F32 || epi == EPI_GATE_RES || sizeof(T) == 4;
  const int cp = f32 ? 4 : 8;
  const long esz = f32 ? 4 : 2;
  if (epi == EPI_SILUGATE_T) { if (g.N % 64) return false; }
  else if (g.N % cp) return false;
  if (g.out_row % cp || g.out_shift % cp || ((uintptr_t)g.out0 * 1) % 16 || (g.out_shift * esz) % 16) return false;
  if (g.bias && ((uintptr_t)g.bias & 15)) return false;
  if ((epi == EPI_GATE_RES || epi == EPI_STORE_F32) && g.rb.p) {
    if (((uintptr_t)g.rb.p & 15) || g.rb.ld % 4 || g.rb.step_stride % 4) return false;
  }
  return true;
}

struct GemmPair {
  GemmArgs g[2];
  int tiles0;  // workgroups belonging to g[0]; the rest run g[1]
};

}  // namespace
