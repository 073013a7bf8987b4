// HBM-bound row kernels of the Foley path: LayerNorm+modulate, RMSNorm+RoPE head split, small
// elementwise helpers, the solver update and the DAC output convolution.  All arithmetic fp32;
// one wavefront per row with 16-byte vector accesses and wave-level (DPP) reductions.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace {

#define FOLEY_LAUNCH_CHECK()                                                             \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) return foley_set_err(hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// ------------------------------------------------------------------ LayerNorm (+ AdaLN modulate)
// reference: nn.LayerNorm(elementwise_affine=False) + modulate() (modulate_layers.py:19-30) and
// SingleStreamBlock's norm*(1+scale)+shift (hifi_foley.py:368,387)
template <typename OutT> struct Pack4;
template <> struct Pack4<float> {
  static __device__ __forceinline__ void store(float* p, const f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Pack4<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* p, const f32x4 v) {
    uint2 w;
    w.x = pack_bf16x2(v[0], v[1]);
    w.y = pack_bf16x2(v[2], v[3]);
    *(uint2*)p = w;
  }
};

template <> struct Pack4<f16_t> {
  static __device__ __forceinline__ void store(f16_t* p, const f32x4 v) {
    uint2 w;
    w.x = pack_f16x2(v[0], v[1]);
    w.y = pack_f16x2(v[2], v[3]);
    *(uint2*)p = w;
  }
};

// four consecutive partial products of a slab row: fp32, or the 16-bit operand type
template <typename ST> __device__ __forceinline__ f32x4 slab_load4(const void* base, long elem) {
  if constexpr (sizeof(ST) == 4) {
    return *(const f32x4*)((const float*)base + elem);
  } else {
    const uint2 w = *(const uint2*)((const ST*)base + elem);
    const ST* h = (const ST*)&w;
    f32x4 v = {Cvt<ST>::from(h[0]), Cvt<ST>::from(h[1]), Cvt<ST>::from(h[2]), Cvt<ST>::from(h[3])};
    return v;
  }
}

// One wave per row; every global load of the row (x, shift, scale) is issued before the first
// reduction so the kernel pays one memory round trip, results leave as 8/16-byte vector stores.
struct LnPair {
  LnArgs a[2];
  int blocks0;
  long long* dbg;   // tools/ln_timeline.py: 4 wall-clock stamps per workgroup of the wide kernel; null in production
};
static long long* g_ln_dbg = nullptr;
extern "C" void foley_debug_ln_timeline(void* p) { g_ln_dbg = (long long*)p; }

// SLAB16: the pending slabs hold OutT (a 16-bit type) instead of fp32
template <typename OutT, int MAXV, bool PEND, bool SLAB16 = false>
__global__ __launch_bounds__(128) void ln_mod_kernel(const LnPair pr, int D, float eps) {
  using ST = typename std::conditional<SLAB16, OutT, float>::type;   // slab element type
  const int sel = (int)blockIdx.x >= pr.blocks0 ? 1 : 0;
  const LnArgs& A = pr.a[sel];
  float* __restrict__ x = A.x;
  const int M = A.M;
  const RowBcast& shift = A.shift;
  const RowBcast& scale = A.scale;
  OutT* __restrict__ out = (OutT*)A.out;
  const int row = ((int)blockIdx.x - (sel ? pr.blocks0 : 0)) * 2 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = D >> 2;  // float4 per row
  f32x4* xr = (f32x4*)(x + (long)row * D);
  const f32x4* sh = shift.p ? (const f32x4*)rb_row(shift, row) : nullptr;
  const f32x4* sc = scale.p ? (const f32x4*)rb_row(scale, row) : nullptr;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 v[MAXV], hv[MAXV], cv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = min(lane + i * 64, nv - 1);  // clamped: loads stay unconditional
    v[i] = xr[c];
    hv[i] = sh ? sh[c] : z4;
    cv[i] = sc ? sc[c] : z4;
  }
  if (PEND && A.pend.partials) {   // (separate instantiation: the slab registers would halve the plain kernel's occupancy)
    // finish the deferred split-K GEMM: x += gate * (sum of partial products + bias), SB slabs in
    // flight at a time (clamped slab index keeps the loads unconditional, weight 0 drops repeats)
    constexpr int SB = 6;
    const LnPending& P = A.pend;
    const f32x4* gp = (const f32x4*)rb_row(P.gate, row);
    const f32x4* bp = (const f32x4*)P.bias;
    f32x4 acc[MAXV], gt[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = min(lane + i * 64, nv - 1);
      gt[i] = gp[c];
      acc[i] = bp ? bp[c] : z4;
    }
    for (int s0 = 0; s0 < P.k; s0 += SB) {
      f32x4 t[SB][MAXV];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int s = min(s0 + u, P.k - 1);
        const long pe = s * P.stride + (long)row * D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) t[u][i] = slab_load4<ST>(P.partials, pe + 4L * min(lane + i * 64, nv - 1));
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const float w = (s0 + u < P.k) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][e] += w * t[u][i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] += gt[i][e] * acc[i][e];
      if (lane + i * 64 < nv) xr[lane + i * 64] = v[i];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + i * 64 < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = v[i][u] - mean;
        q += d * d;
      }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  OutT* orow = out + (long)row * D;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      f32x4 y;
#pragma unroll
      for (int u = 0; u < 4; ++u) y[u] = (v[i][u] - mean) * rstd * (1.0f + cv[i][u]) + hv[i][u];
      Pack4<OutT>::store(orow + c * 4, y);
    }
  }
}

// Same operation with WPR waves cooperating on one row (D = 64 * WPR * MAXV float4): at M = 500 the
// one-wave-per-row kernel puts 500 waves on 256 CUs, each walking a 6 KiB row (plus up to five partial
// slabs) alone - latency bound, not bandwidth bound.  Splitting a row over WPR waves multiplies the
// loads in flight; the two statistics cross the waves through LDS.
// SEL (which row set of the pair) is a template parameter: argument fields are then loaded at constant
// kernel-argument offsets, in a few wide scalar loads at entry, instead of one dependent dword at a time
// in front of the first global load of this latency-bound kernel (same reason as gemm_ws_body).
#ifndef FOLEY_LN_RPB
#define FOLEY_LN_RPB 1   // rows per workgroup of the multi-wave LayerNorm (A/B builds with -DFOLEY_LN_RPB=2 / 4 through FOLEY_HIP_LIB: 1 row
                         // -0.3 % on the bs=1 loop against 2, 4 rows +0.5 %)
#endif
template <typename OutT, int MAXV, bool PEND, int WPR, int SEL, bool SLAB16>
__device__ __forceinline__ void ln_mod_wide_body(const LnPair& pr, int D, float eps) {
  using ST = typename std::conditional<SLAB16, OutT, float>::type;   // slab element type
  constexpr int RPB = FOLEY_LN_RPB;
  __shared__ float red[RPB][2][WPR];   // [row of the block][statistic][wave of the row]
  constexpr int sel = SEL;
  const LnArgs& A = pr.a[SEL];
  const int M = A.M;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = wave / WPR, part = wave % WPR;               // row of the block (0/1), column part
  const int row_u = ((int)blockIdx.x - (sel ? pr.blocks0 : 0)) * RPB + rb;
  const bool live = row_u < M;
  const int row = live ? row_u : M - 1;                       // dead waves shadow the last row (no stores)
  const int c0 = part * (MAXV * 64) + lane;                   // first float4 of this lane; stride 64
  const bool stamp = pr.dbg && threadIdx.x == 0;
  if (stamp) pr.dbg[(long)blockIdx.x * 4 + 0] = wall_clock64();
  f32x4* xr = (f32x4*)(A.x + (long)row * D);
  const f32x4* sh = A.shift.p ? (const f32x4*)rb_row(A.shift, row) : nullptr;
  const f32x4* sc = A.scale.p ? (const f32x4*)rb_row(A.scale, row) : nullptr;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 v[MAXV], hv[MAXV], cv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = c0 + i * 64;
    v[i] = xr[c];
    hv[i] = sh ? sh[c] : z4;
    cv[i] = sc ? sc[c] : z4;
  }
  if (PEND && A.pend.partials) {
    constexpr int SB = 6;
    const LnPending& P = A.pend;
    const f32x4* gp = (const f32x4*)rb_row(P.gate, row);
    const f32x4* bp = (const f32x4*)P.bias;
    f32x4 acc[MAXV], gt[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = c0 + i * 64;
      gt[i] = gp[c];
      acc[i] = bp ? bp[c] : z4;
    }
    for (int s0 = 0; s0 < P.k; s0 += SB) {
      f32x4 t[SB][MAXV];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int s = min(s0 + u, P.k - 1);
        const long pe = s * P.stride + (long)row * D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) t[u][i] = slab_load4<ST>(P.partials, pe + 4L * (c0 + i * 64));
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const float w = (s0 + u < P.k) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][e] += w * t[u][i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] += gt[i][e] * acc[i][e];
      if (live) xr[c0 + i * 64] = v[i];
    }
  }
  if (pr.dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp) pr.dbg[(long)blockIdx.x * 4 + 1] = wall_clock64();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  // Each wave reduces its own part to (sum, centred sum of squares) in registers; the parts meet once in LDS and
  // are combined exactly (Chan et al.): M2 = sum_w [M2_w + n_w (mean_w - mean)^2] - one barrier instead of two.
  s = wave_sum(s);
  const float nw = (float)(MAXV * 4 * 64);
  const float mean_w = s / nw;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d = v[i][u] - mean_w;
      q += d * d;
    }
  q = wave_sum(q);
  if (lane == 0) {
    red[rb][0][part] = s;
    red[rb][1][part] = q;
  }
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < WPR; ++w) tot += red[rb][0][w];     // fixed order: every wave of the row gets the same mean
  const float mean = tot / (float)D;
  float qt = 0.f;
#pragma unroll
  for (int w = 0; w < WPR; ++w) {
    const float dm = red[rb][0][w] / nw - mean;
    qt += red[rb][1][w] + nw * dm * dm;
  }
  const float rstd = 1.0f / sqrtf(qt / (float)D + eps);
  if (stamp) pr.dbg[(long)blockIdx.x * 4 + 2] = wall_clock64();
  if (!live) return;
  OutT* orow = (OutT*)A.out + (long)row * D;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    f32x4 y;
#pragma unroll
    for (int u = 0; u < 4; ++u) y[u] = (v[i][u] - mean) * rstd * (1.0f + cv[i][u]) + hv[i][u];
    Pack4<OutT>::store(orow + (c0 + i * 64) * 4, y);
  }
  if (stamp) pr.dbg[(long)blockIdx.x * 4 + 3] = wall_clock64();
}

template <typename OutT, int MAXV, bool PEND, int WPR, bool SLAB16 = false>
__global__ __launch_bounds__(64 * FOLEY_LN_RPB * WPR) void ln_mod_wide_kernel(const LnPair pr, int D, float eps) {
  if ((int)blockIdx.x >= pr.blocks0) ln_mod_wide_body<OutT, MAXV, PEND, WPR, 1, SLAB16>(pr, D, eps);   // workgroup-uniform
  else ln_mod_wide_body<OutT, MAXV, PEND, WPR, 0, SLAB16>(pr, D, eps);
}

// ------------------------------------------------------------------ q/k RMSNorm + RoPE + head split
// reference: rearrange "(K H D)", RMSNorm (norm_layers.py:36-52 / nn.RMSNorm), apply_rotary_emb
// (attn_layers.py:112-146).  One wave per (row, head, operand); lane owns the rotation pair
// (2*lane, 2*lane+1).
struct QkvPair {
  QkvSplitArgs a[2];
  int blocks0;
};

// Two workgroup roles in one launch: (1) one wave per (row, head, operand) for every operand that
// keeps the [clip, H, S, 128] layout - RMSNorm / RoPE are per (row, head), lane = rotation pair;
// (2) for a transposed V operand, one workgroup per (clip, head, 32-token tile) moves the tile
// through LDS so that V^T leaves as 64-byte runs instead of 2-byte scatters.
template <typename OutT>
__global__ __launch_bounds__(256) void qkv_split_kernel(const QkvPair pr) {
  __shared__ OutT vt[32][128 + 2];
  const int sel = (int)blockIdx.x >= pr.blocks0 ? 1 : 0;
  const QkvSplitArgs& a = pr.a[sel];
  int bid = (int)blockIdx.x - (sel ? pr.blocks0 : 0);
  const bool vtrans = a.vt_pitch > 0;
  const int nQ = a.nK - (vtrans ? 1 : 0);  // operands handled by role (1)
  const int lane = threadIdx.x & 63;
  constexpr int IPW = 4;  // items per wave: their loads are issued together (memory-level parallelism)
  const long n_items = (long)a.M * a.H * nQ;
  const int bqk = (int)((n_items + 4 * IPW - 1) / (4 * IPW));
  if (bid < bqk) {
    const long it0 = ((long)bid * 4 + (threadIdx.x >> 6)) * IPW;
    if (it0 >= n_items) return;
    float x0[IPW], x1[IPW];
    int wq[IPW], hq[IPW], rq[IPW];
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const long it = min(it0 + u, n_items - 1);
      wq[u] = (int)(it % nQ);
      hq[u] = (int)((it / nQ) % a.H);
      rq[u] = (int)(it / ((long)nQ * a.H));
      const float* src = a.qkv + (long)rq[u] * (a.nK * a.H * 128) + (long)wq[u] * a.H * 128 + hq[u] * 128 + 2 * lane;
      x0[u] = src[0];
      x1[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      if (it0 + u >= n_items) break;
      const int w = wq[u], h = hq[u], r = rq[u];
      const int b = r / a.L, l = r - b * a.L;
      float y0 = x0[u], y1 = x1[u];
      if (a.gain[w]) {
        const float ss = wave_sum(y0 * y0 + y1 * y1);
        const float rinv = rsqrtf(ss * (1.0f / 128.0f) + a.eps);
        y0 = y0 * rinv * a.gain[w][2 * lane];
        y1 = y1 * rinv * a.gain[w][2 * lane + 1];
      }
      if (a.pos[w]) {
        const int p = a.pos[w][l];
        const float c = a.cos_tab[(long)p * 64 + lane], sn = a.sin_tab[(long)p * 64 + lane];
        const float z0 = y0 * c - y1 * sn;
        const float z1 = y1 * c + y0 * sn;
        y0 = z0;
        y1 = z1;
      }
      OutT* dst = (OutT*)a.dst[w] + (((long)b * a.H + h) * a.S_tot + a.tok_off + l) * 128 + 2 * lane;
      dst[0] = Cvt<OutT>::to(y0);
      dst[1] = Cvt<OutT>::to(y1);
    }
    return;
  }
  // ---- role (2): V tile -> V^T  (plain copy: V is neither normalised nor rotated)
  bid -= bqk;
  const int tiles = (a.L + 31) >> 5;
  const int tl = bid % tiles;
  bid /= tiles;
  const int h = bid % a.H;
  const int b = bid / a.H;
  const int l0 = tl * 32;
  const int nrows = min(32, a.L - l0);
  const int w = a.nK - 1;
  {
    // 32 rows x 128 fp32 = 1024 float4: 4 per thread, all in flight together
    const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;  // float4 column, row group
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = r0 + i * 8;
      const int l = l0 + min(lr, nrows - 1);
      const f32x4 v = *(const f32x4*)(a.qkv + ((long)b * a.L + l) * (a.nK * a.H * 128) + (long)w * a.H * 128 +
                                      h * 128 + c4 * 4);
      vt[lr][c4 * 4 + 0] = Cvt<OutT>::to(v[0]);
      vt[lr][c4 * 4 + 1] = Cvt<OutT>::to(v[1]);
      vt[lr][c4 * 4 + 2] = Cvt<OutT>::to(v[2]);
      vt[lr][c4 * 4 + 3] = Cvt<OutT>::to(v[3]);
    }
  }
  __syncthreads();
  const int d = threadIdx.x >> 1, half = threadIdx.x & 1;  // channel, half of the tile
  OutT* dst = (OutT*)a.dst[w] + (((long)b * a.H + h) * 128 + d) * a.vt_pitch + a.tok_off + l0 + half * 16;
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (half * 16 + t < nrows) dst[t] = vt[half * 16 + t][d];
}

// ------------------------------------------------------------------ small elementwise helpers
template <typename OutT>
__global__ void rows_add_act_kernel(const float* __restrict__ a, RowBcast v, int R, int D, int act_silu,
                                    OutT* __restrict__ out) {
  const long n = (long)R * D;
  const float* vr = v.p ? rb_row(v, 0) : nullptr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float y = a ? a[i] : 0.f;
    if (vr) y += vr[i % D];
    if (act_silu) y = silu_f(y);
    out[i] = Cvt<OutT>::to(y);
  }
}

template <typename OutT>
__global__ void add_periodic_kernel(const float* __restrict__ x, const float* __restrict__ pos, int R, int D,
                                    int period, OutT* __restrict__ out) {
  const long n = (long)R * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / D), c = (int)(i - (long)r * D);
    out[i] = Cvt<OutT>::to(x[i] + pos[(long)(r % period) * D + c]);
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int n_idx,
                                   int groups, int src_rows, int D, float* __restrict__ out) {
  const long n = (long)groups * n_idx * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    const int g = (int)(r / n_idx), l = (int)(r - (long)g * n_idx);
    out[i] = src[((long)g * src_rows + idx[l]) * D + c];
  }
}

template <typename S, typename Dst>
__global__ void cast_kernel(const S* __restrict__ s, Dst* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[i] = Cvt<Dst>::to(Cvt<S>::from(s[i]));
}

// latents [clips, C, L] -> token rows [(cfg*clips + b)*L + l, C]  (PatchEmbed1D's transpose,
// embed_layers.py:43-52, and the CFG duplication of utils.py:205)
template <typename OutT>
__global__ __launch_bounds__(256) void latent_rows_kernel(const float* __restrict__ x, int clips, int C, int L,
                                                          int ncfg, OutT* __restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int l0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, l = l0 + tx;
    tile[i][tx] = (c < C && l < L) ? x[((long)b * C + c) * L + l] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int l = l0 + i, c = c0 + tx;
    if (l < L && c < C) {
      const OutT v = Cvt<OutT>::to(tile[tx][i]);
      for (int g = 0; g < ncfg; ++g) out[(((long)g * clips + b) * L + l) * C + c] = v;
    }
  }
}

// ------------------------------------------------------------------ solver update
// CFG combine (utils.py:241-243) + FlowMatchDiscreteScheduler.step (scheduling_flow_match_
// discrete.py:262-297 and the multi-stage bookkeeping :299-373) driven by a per-iteration
// coefficient row {w_new, w_acc, dt, w_store, flags}; then re-stages the next model input rows.
constexpr int STEP_SAVE_X = 1, STEP_USE_SAVED = 2, STEP_ACC_RESET = 4;

template <typename OutT>
__global__ __launch_bounds__(256) void solver_step_kernel(const StepArgs a) {
  __shared__ float tile[32][33];
  const int it = *a.step_ptr;
  const float* cf = a.coef + (long)it * 8;
  const float w_new = cf[0], w_acc = cf[1], dt = cf[2], w_store = cf[3];
  const int flags = (int)cf[4];
  const int b = blockIdx.z;
  const int l0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long rows = (long)a.clips * a.L;
  // phase 1: read pred rows [l][c] (coalesced over c) and transpose through LDS
  for (int i = ty; i < 32; i += 8) {
    const int l = l0 + i, c = c0 + tx;
    float v = 0.f;
    if (l < a.L && c < a.C) {
      const long r = (long)b * a.L + l;
      if (a.ncfg == 2) {
        const float u = a.pred[r * a.C + c], cnd = a.pred[(rows + r) * a.C + c];
        v = u + a.guidance * (cnd - u);
      } else {
        v = a.pred[r * a.C + c];
      }
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  // phase 2: update x [b][c][l] (coalesced over l)
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, l = l0 + tx;
    float xn = 0.f;
    if (c < a.C && l < a.L) {
      const long xi = ((long)b * a.C + c) * a.L + l;
      const float v = tile[tx][i];
      const float xc = a.x[xi];
      float acc = 0.f;
      if (a.d_acc) acc = (flags & STEP_ACC_RESET) ? 0.f : a.d_acc[xi];
      const float deriv = (w_acc != 0.f) ? (w_new * v + w_acc * acc) : (w_new * v);
      const float base = (flags & STEP_USE_SAVED) ? a.x_saved[xi] : xc;
      if (flags & STEP_SAVE_X) a.x_saved[xi] = xc;
      xn = base + deriv * dt;
      a.x[xi] = xn;
      if (a.d_acc) a.d_acc[xi] = acc + w_store * v;
    }
    tile[tx][i] = xn;
  }
  __syncthreads();
  // phase 3: next model input rows (cfg-duplicated), coalesced over c
  if (a.rows_out) {
    for (int i = ty; i < 32; i += 8) {
      const int l = l0 + i, c = c0 + tx;
      if (l < a.L && c < a.C) {
        const OutT v = Cvt<OutT>::to(tile[i][tx]);
        for (int g = 0; g < a.ncfg; ++g)
          ((OutT*)a.rows_out)[(((long)g * a.clips + b) * a.L + l) * a.C + c] = v;
      }
    }
  }
}

__global__ void step_increment_kernel(int* p) { *p = *p + 1; }

// ------------------------------------------------------------------ DAC output conv (64 -> 1, k=7) + tanh
// reference: decoder.model[-2:] = WNConv1d(C, 1, 7, padding=3), nn.Tanh() (dac.py:141-146).
// A block handles 64 consecutive samples; the (64+6) x C snake-activated rows are staged in LDS.
__global__ __launch_bounds__(256) void dac_out_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int T, int C,
                                                      float* __restrict__ out) {
  extern __shared__ float sm[];  // (70 rows) * (C + 1) + 7 * C weights
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 64;
  const int pitch = C + 1;
  float* rows = sm;
  float* wl = sm + 70 * pitch;
  for (int i = threadIdx.x; i < 7 * C; i += 256) wl[i] = w[i];
  for (int i = threadIdx.x; i < 70 * C; i += 256) {
    const int rr = i / C, c = i - rr * C;
    const int t = t0 + rr - 3;
    rows[rr * pitch + c] = (t >= 0 && t < T) ? s[((long)b * T + t) * C + c] : 0.f;
  }
  __syncthreads();
  // 4 lanes per output sample, each covering a quarter of the channels
  const int ti = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int cq = C >> 2;
  float acc = 0.f;
  for (int j = 0; j < 7; ++j) {
    const float* rp = rows + (ti + j) * pitch + part * cq;
    const float* wp = wl + j * C + part * cq;
    for (int c = 0; c < cq; ++c) acc += rp[c] * wp[c];
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  const int t = t0 + ti;
  if (part == 0 && t < T) out[(long)b * T + t] = tanhf(acc + bias[0]);
}

// ------------------------------------------------------------------ DAC encoder input conv (1 -> C, k=7)
// reference: Encoder.block[0] = WNConv1d(1, d_model, 7, padding=3) (dac.py:86); the first residual
// unit needs both y (trunk) and snake(y) (its conv input), so both are written.
__global__ __launch_bounds__(256) void dac_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ alpha,
                                                     int B, int T, int C, float* __restrict__ out0,
                                                     float* __restrict__ out1) {
  const int c4n = C >> 2;
  const long n = (long)B * T * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long bt = i / c4n;
    const int t = (int)(bt % T);
    const float* xr = x + (bt - t);
    f32x4 acc = *(const f32x4*)(bias + c);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int tt = t + j - 3;
      const float xv = (tt >= 0 && tt < T) ? xr[tt] : 0.f;
      const f32x4 wv = *(const f32x4*)(w + j * C + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += xv * wv[u];
    }
    *(f32x4*)(out0 + bt * C + c) = acc;
    const f32x4 al = *(const f32x4*)(alpha + c);
    f32x4 sn;
#pragma unroll
    for (int u = 0; u < 4; ++u) sn[u] = snake_f(acc[u], al[u], 1.0f / (al[u] + 1e-9f));
    *(f32x4*)(out1 + bt * C + c) = sn;
  }
}

// rows [B*T, C] -> planes [B, C, T] (the [B, 2*latent, T'] layout DAC.encode returns)
__global__ void rows_to_planes_kernel(const float* __restrict__ rows, int B, int T, int C, float* __restrict__ out) {
  const long n = (long)B * T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long bc = i / T;
    const int c = (int)(bc % C);
    const long b = bc / C;
    out[i] = rows[(b * T + t) * C + c];
  }
}

inline int grid1d(long n, int block) {
  long g = (n + block - 1) / block;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

int launch_ln_mod_pair(const LnArgs& a0, const LnArgs& a1, int D, float eps, int out_dtype, hipStream_t st) {
  if (D % 4 || D > 8 * 256) return foley_set_err("ln_mod: D must be a multiple of 4 and <= 2048", __FILE__, __LINE__);
  for (const LnArgs* a : {&a0, &a1})
    if (a->M > 0 && a->pend.partials &&
        (a->pend.k < 1 || a->pend.stride % 4 || ((uintptr_t)a->pend.partials & 15) || ((uintptr_t)a->pend.bias & 15) ||
         !a->pend.gate.p || ((uintptr_t)a->pend.gate.p & 15) || a->pend.gate.ld % 4 || a->pend.gate.step_stride % 4))
      return foley_set_err("ln_mod: bad pending split-K descriptor", __FILE__, __LINE__);
  LnPair pr;
  pr.a[0] = a0;
  pr.a[1] = a1;
  pr.blocks0 = (a0.M + 1) / 2;
  pr.dbg = g_ln_dbg;
  dim3 grid(pr.blocks0 + (a1.M + 1) / 2), block(128);
  const int need = (D / 4 + 63) / 64;   // float4 per lane
  if (out_dtype != FOLEY_F32 && !foley_is_half(out_dtype)) return foley_set_err("ln_mod: bad dtype", __FILE__, __LINE__);
  const bool f32o = out_dtype == FOLEY_F32, f16o = out_dtype == FOLEY_F16;
  const bool pend = (a0.M > 0 && a0.pend.partials) || (a1.M > 0 && a1.pend.partials);
  // 16-bit slabs (LnPending::half): both row sets of a launch agree, and the slab type is the output type
  const int h0 = (a0.M > 0 && a0.pend.partials) ? a0.pend.half : -1, h1 = (a1.M > 0 && a1.pend.partials) ? a1.pend.half : -1;
  const int slab_dt = h0 >= 0 ? h0 : (h1 >= 0 ? h1 : 0);
  if ((h0 >= 0 && h1 >= 0 && h0 != h1) || (slab_dt != 0 && slab_dt != out_dtype))
    return foley_set_err("ln_mod: 16-bit pending slabs must have the output dtype (and agree between the two row sets)", __FILE__, __LINE__);
  const bool s16 = slab_dt != 0;
#define FOLEY_LN(V)                                                                                        \
  {                                                                                                        \
    if (pend) {                                                                                            \
      if (f32o) FOLEY_LAUNCH((ln_mod_kernel<float, V, true>), grid, block, 0, st, pr, D, eps);       \
      else if (f16o && s16) FOLEY_LAUNCH((ln_mod_kernel<f16_t, V, true, true>), grid, block, 0, st, pr, D, eps);  \
      else if (f16o) FOLEY_LAUNCH((ln_mod_kernel<f16_t, V, true>), grid, block, 0, st, pr, D, eps);  \
      else if (s16) FOLEY_LAUNCH((ln_mod_kernel<bf16_t, V, true, true>), grid, block, 0, st, pr, D, eps);  \
      else FOLEY_LAUNCH((ln_mod_kernel<bf16_t, V, true>), grid, block, 0, st, pr, D, eps);           \
    } else {                                                                                               \
      if (f32o) FOLEY_LAUNCH((ln_mod_kernel<float, V, false>), grid, block, 0, st, pr, D, eps);      \
      else if (f16o) FOLEY_LAUNCH((ln_mod_kernel<f16_t, V, false>), grid, block, 0, st, pr, D, eps); \
      else FOLEY_LAUNCH((ln_mod_kernel<bf16_t, V, false>), grid, block, 0, st, pr, D, eps);          \
    }                                                                                                      \
  }
  // rows that split evenly over 2 / 3 waves (D = 1536: 2 waves x 3 float4 per lane, or 3 x 2; 1408 / 256: one wave).  Two waves
  // per row since the slabs are 16-bit (round 3, one box: bs=1 loop 347.4 -> 345.8 ms, bs=8 1525.7 -> 1520.2, 30 s 1488.7 -> 1478.6);
  // three were better with fp32 slabs (round 2)
  static const int wide = []() { const char* e = getenv("FOLEY_LN_WIDE"); return e ? atoi(e) : 2; }();   // A/B switch (0 = one wave per row, 3)
  const int total_rows = a0.M + a1.M;
  LnPair prw = pr;   // the multi-wave kernels take FOLEY_LN_RPB rows per workgroup
  prw.blocks0 = (a0.M + FOLEY_LN_RPB - 1) / FOLEY_LN_RPB;
  const dim3 gridw(prw.blocks0 + (a1.M + FOLEY_LN_RPB - 1) / FOLEY_LN_RPB);
  if (wide && total_rows <= 4096 && D % (4 * 64 * 3) == 0 && D / (4 * 64 * 3) <= 4 && wide == 3) {
#define FOLEY_LNW(V, W)                                                                                               \
    {                                                                                                                  \
      dim3 blk(64 * FOLEY_LN_RPB * W);                                                                                               \
      if (pend) {                                                                                                      \
        if (f32o) FOLEY_LAUNCH((ln_mod_wide_kernel<float, V, true, W>), gridw, blk, 0, st, prw, D, eps);                \
        else if (f16o && s16) FOLEY_LAUNCH((ln_mod_wide_kernel<f16_t, V, true, W, true>), gridw, blk, 0, st, prw, D, eps);  \
        else if (f16o) FOLEY_LAUNCH((ln_mod_wide_kernel<f16_t, V, true, W>), gridw, blk, 0, st, prw, D, eps);           \
        else if (s16) FOLEY_LAUNCH((ln_mod_wide_kernel<bf16_t, V, true, W, true>), gridw, blk, 0, st, prw, D, eps);     \
        else FOLEY_LAUNCH((ln_mod_wide_kernel<bf16_t, V, true, W>), gridw, blk, 0, st, prw, D, eps);                    \
      } else {                                                                                                         \
        if (f32o) FOLEY_LAUNCH((ln_mod_wide_kernel<float, V, false, W>), gridw, blk, 0, st, prw, D, eps);               \
        else if (f16o) FOLEY_LAUNCH((ln_mod_wide_kernel<f16_t, V, false, W>), gridw, blk, 0, st, prw, D, eps);          \
        else FOLEY_LAUNCH((ln_mod_wide_kernel<bf16_t, V, false, W>), gridw, blk, 0, st, prw, D, eps);                   \
      }                                                                                                                \
    }
    const int v3 = D / (4 * 64 * 3);
    if (v3 == 1) FOLEY_LNW(1, 3)
    else if (v3 == 2) FOLEY_LNW(2, 3)
    else if (v3 == 3) FOLEY_LNW(3, 3)
    else FOLEY_LNW(4, 3)
    FOLEY_LAUNCH_CHECK();
    return 0;
  }
  if (wide == 2 && total_rows <= 4096 && D % (4 * 64 * 2) == 0 && D / (4 * 64 * 2) <= 4) {
    const int v2 = D / (4 * 64 * 2);
    if (v2 == 1) FOLEY_LNW(1, 2)
    else if (v2 == 2) FOLEY_LNW(2, 2)
    else if (v2 == 3) FOLEY_LNW(3, 2)
    else FOLEY_LNW(4, 2)
    FOLEY_LAUNCH_CHECK();
    return 0;
  }
#undef FOLEY_LNW
  if (need <= 2) FOLEY_LN(2)
  else if (need <= 4) FOLEY_LN(4)
  else if (need <= 6) FOLEY_LN(6)
  else FOLEY_LN(8)
#undef FOLEY_LN
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_ln_mod(const float* x, int M, int D, float eps, const RowBcast& shift, const RowBcast& scale,
                  void* out, int out_dtype, hipStream_t st) {
  LnArgs a0{const_cast<float*>(x), M, shift, scale, out, LnPending{}};
  LnArgs a1 = a0;
  a1.M = 0;
  return launch_ln_mod_pair(a0, a1, D, eps, out_dtype, st);
}

int launch_ln_mod_pending(float* x, int M, int D, float eps, const RowBcast& shift, const RowBcast& scale,
                          void* out, int out_dtype, const LnPending& pend, hipStream_t st) {
  LnArgs a0{x, M, shift, scale, out, pend};
  LnArgs a1 = a0;
  a1.M = 0;
  return launch_ln_mod_pair(a0, a1, D, eps, out_dtype, st);
}

int launch_qkv_split_pair(const QkvSplitArgs& a0, const QkvSplitArgs& a1, hipStream_t st) {
  if (a0.out_dtype != a1.out_dtype) return foley_set_err("qkv_split pair: dtype mismatch", __FILE__, __LINE__);
  QkvPair pr;
  pr.a[0] = a0;
  pr.a[1] = a1;
  auto nblk = [](const QkvSplitArgs& q) {
    if (q.M <= 0) return 0;
    const int nQ = q.nK - (q.vt_pitch > 0 ? 1 : 0);
    const int bqk = (int)(((long)q.M * q.H * nQ + 15) / 16);
    const int bv = q.vt_pitch > 0 ? (q.M / q.L) * q.H * ((q.L + 31) / 32) : 0;   // rows are [clip][l]
    return bqk + bv;
  };
  if ((a0.M > 0 && a0.M % a0.L) || (a1.M > 0 && a1.M % a1.L))
    return foley_set_err("qkv_split: M must be a multiple of L", __FILE__, __LINE__);
  pr.blocks0 = nblk(a0);
  dim3 grid((unsigned)(pr.blocks0 + nblk(a1))), block(256);
  if (grid.x == 0) return 0;
  if (a0.out_dtype == FOLEY_BF16) FOLEY_LAUNCH(qkv_split_kernel<bf16_t>, grid, block, 0, st, pr);
  else if (a0.out_dtype == FOLEY_F16) FOLEY_LAUNCH(qkv_split_kernel<f16_t>, grid, block, 0, st, pr);
  else FOLEY_LAUNCH(qkv_split_kernel<float>, grid, block, 0, st, pr);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_qkv_split(const QkvSplitArgs& a, hipStream_t st) {
  QkvSplitArgs none = a;
  none.M = 0;
  return launch_qkv_split_pair(a, none, st);
}

// flag |= 1 when some row s >= period of a group differs (bit pattern) from row s - period
__global__ void rows_periodic_check_kernel(const unsigned* __restrict__ x, int groups, int rows, int period, int D, int* flag) {
  const long per_g = (long)(rows - period) * D, n = (long)groups * per_g;
  unsigned diff = 0;   // bit g: group g (<= 32 groups) has a row that differs from the one `period` rows above it -> flag[g] = 1
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long g = i / per_g, r = i - g * per_g;
    const long at = (g * rows + period) * D + r;
    if (x[at] != x[at - (long)period * D]) diff |= 1u << g;
  }
  for (int g = 0; g < groups; ++g)
    if (diff >> g & 1u) atomicOr(flag + g, 1);
}

int launch_rows_periodic_check(const float* x, int groups, int rows, int period, int D, int* flag, hipStream_t st) {
  if (rows <= period) return 0;
  if (groups > 32) return foley_set_err("rows_periodic_check: at most 32 groups", __FILE__, __LINE__);
  const long n = (long)groups * (rows - period) * D;
  FOLEY_LAUNCH(rows_periodic_check_kernel, dim3(grid1d(n, 256)), dim3(256), 0, st, (const unsigned*)x, groups, rows, period, D, flag);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_rows_add_act(const float* a, const RowBcast& v, int R, int D, int act_silu, void* out, int out_dtype,
                        hipStream_t st) {
  const long n = (long)R * D;
  if (out_dtype == FOLEY_F32)
    FOLEY_LAUNCH(rows_add_act_kernel<float>, dim3(grid1d(n, 256)), dim3(256), 0, st, a, v, R, D, act_silu, (float*)out);
  else if (out_dtype == FOLEY_BF16)
    FOLEY_LAUNCH(rows_add_act_kernel<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, a, v, R, D, act_silu, (bf16_t*)out);
  else if (out_dtype == FOLEY_F16)
    FOLEY_LAUNCH(rows_add_act_kernel<f16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, a, v, R, D, act_silu, (f16_t*)out);
  else return foley_set_err("rows_add_act: bad dtype", __FILE__, __LINE__);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_add_periodic(const float* x, const float* pos, int R, int D, int period, void* out, int out_dtype,
                        hipStream_t st) {
  const long n = (long)R * D;
  if (out_dtype == FOLEY_F32)
    FOLEY_LAUNCH(add_periodic_kernel<float>, dim3(grid1d(n, 256)), dim3(256), 0, st, x, pos, R, D, period, (float*)out);
  else if (out_dtype == FOLEY_BF16)
    FOLEY_LAUNCH(add_periodic_kernel<bf16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, x, pos, R, D, period, (bf16_t*)out);
  else if (out_dtype == FOLEY_F16)
    FOLEY_LAUNCH(add_periodic_kernel<f16_t>, dim3(grid1d(n, 256)), dim3(256), 0, st, x, pos, R, D, period, (f16_t*)out);
  else return foley_set_err("add_periodic: bad dtype", __FILE__, __LINE__);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_gather_rows(const float* src, const int* idx, int n_idx, int groups, int src_rows, int D, float* out,
                       hipStream_t st) {
  const long n = (long)groups * n_idx * D;
  FOLEY_LAUNCH(gather_rows_kernel, dim3(grid1d(n, 256)), dim3(256), 0, st, src, idx, n_idx, groups, src_rows, D, out);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_cast(const void* src, int sd, void* dst, int dd, long n, hipStream_t st) {
  dim3 g(grid1d(n, 256)), b(256);
  if (sd == FOLEY_F32 && dd == FOLEY_BF16)
    FOLEY_LAUNCH((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (sd == FOLEY_BF16 && dd == FOLEY_F32)
    FOLEY_LAUNCH((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (sd == FOLEY_F32 && dd == FOLEY_F16)
    FOLEY_LAUNCH((cast_kernel<float, f16_t>), g, b, 0, st, (const float*)src, (f16_t*)dst, n);
  else if (sd == FOLEY_F16 && dd == FOLEY_F32)
    FOLEY_LAUNCH((cast_kernel<f16_t, float>), g, b, 0, st, (const f16_t*)src, (float*)dst, n);
  else if (sd == FOLEY_F32 && dd == FOLEY_F32)
    FOLEY_LAUNCH((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n);
  else return foley_set_err("cast: unsupported dtype pair", __FILE__, __LINE__);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_latent_rows(const float* x, int clips, int C, int L, int ncfg, void* out, int out_dtype, hipStream_t st) {
  dim3 grid((L + 31) / 32, (C + 31) / 32, clips), block(256);
  if (out_dtype == FOLEY_F32)
    FOLEY_LAUNCH(latent_rows_kernel<float>, grid, block, 0, st, x, clips, C, L, ncfg, (float*)out);
  else if (out_dtype == FOLEY_BF16)
    FOLEY_LAUNCH(latent_rows_kernel<bf16_t>, grid, block, 0, st, x, clips, C, L, ncfg, (bf16_t*)out);
  else if (out_dtype == FOLEY_F16)
    FOLEY_LAUNCH(latent_rows_kernel<f16_t>, grid, block, 0, st, x, clips, C, L, ncfg, (f16_t*)out);
  else return foley_set_err("latent_rows: bad dtype", __FILE__, __LINE__);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_solver_step(const StepArgs& a, hipStream_t st) {
  dim3 grid((a.L + 31) / 32, (a.C + 31) / 32, a.clips), block(256);
  if (a.rows_dtype == FOLEY_BF16) FOLEY_LAUNCH(solver_step_kernel<bf16_t>, grid, block, 0, st, a);
  else if (a.rows_dtype == FOLEY_F16) FOLEY_LAUNCH(solver_step_kernel<f16_t>, grid, block, 0, st, a);
  else FOLEY_LAUNCH(solver_step_kernel<float>, grid, block, 0, st, a);
  FOLEY_LAUNCH_CHECK();
  FOLEY_LAUNCH(step_increment_kernel, dim3(1), dim3(1), 0, st, a.step_ptr);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_dac_in(const float* x, const float* w, const float* bias, const float* alpha, int B, int T, int C,
                  float* out0, float* out1, hipStream_t st) {
  if (C % 4) return foley_set_err("dac_in: channel count must be a multiple of 4", __FILE__, __LINE__);
  FOLEY_LAUNCH(dac_in_kernel, dim3(grid1d((long)B * T * (C / 4), 256)), dim3(256), 0, st, x, w, bias, alpha, B, T, C,
                     out0, out1);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Token regrouping between a fused q / k / v projection and the attention of the conditioning encoders (round 5): the
// divided space-time attention of the Synchformer (reference models/synchformer/vit_helper.py:37-105: patch tokens attend over
// the frames of their location or the locations of their frame, the CLS key / value prepended to every group) and the plain
// ViT attention of SigLIP2 / CLAP.  qkv [rows, 3*H*64] ('(K H D)' packing of nn.Linear(dim, 3 dim)); group g takes source rows
// idx_q[g][0..Sq) as queries and idx_kv[g][0..Skv) as keys / values, written head-major as foley_op_attention_hd reads them:
// q [G,H,Sq,64], k [G,H,Skv,64], v [G,H,Skv,64] or - 16-bit operands - v^T [G,H,64,pitch] (zeros beyond Skv) through a 64x64 LDS
// transpose.  One launch replaces ~10 full-tensor torch copies (permute / cat / contiguous / the transposed-V staging).
template <typename E>   // E = the element's storage type (uint16_t: bf16 / fp16, uint32_t: fp32)
__global__ __launch_bounds__(256) void qkv_regroup_kernel(const E* __restrict__ qkv, int H, const int* __restrict__ idx_q, int Sq,
                                                          const int* __restrict__ idx_kv, int Skv, E* __restrict__ q, E* __restrict__ k,
                                                          E* __restrict__ v, int vt_pitch, int n_rows) {
  constexpr int HD = 64, CPR = HD * sizeof(E) / 16;   // 16-byte chunks per head row
  __shared__ __attribute__((aligned(16))) E tile[64][HD + 16 / sizeof(E)];
  const int t0 = blockIdx.x * 64, h = blockIdx.y, g = blockIdx.z;
  const long ld = 3L * H * HD;
  const int tid = threadIdx.x;
  for (int c = tid; c < 64 * CPR; c += 256) {
    const int t = t0 + c / CPR, ch = c % CPR;
    if (t < Sq) {
      const long r = min(max(idx_q[(long)g * Sq + t], 0), n_rows - 1);   // caller-supplied table: never read outside qkv [n_rows, 3*H*64]
      *(u32x4*)(q + (((long)g * H + h) * Sq + t) * HD + ch * (16 / sizeof(E))) = *(const u32x4*)(qkv + r * ld + (long)h * HD + ch * (16 / sizeof(E)));
    }
    if (t < Skv) {
      const long r = min(max(idx_kv[(long)g * Skv + t], 0), n_rows - 1);
      const E* src = qkv + r * ld + (long)(H + h) * HD + ch * (16 / sizeof(E));
      *(u32x4*)(k + (((long)g * H + h) * Skv + t) * HD + ch * (16 / sizeof(E))) = *(const u32x4*)src;
      const u32x4 vv = *(const u32x4*)(src + (long)H * HD);
      if (vt_pitch == 0) *(u32x4*)(v + (((long)g * H + h) * Skv + t) * HD + ch * (16 / sizeof(E))) = vv;
      else *(u32x4*)&tile[c / CPR][ch * (16 / sizeof(E))] = vv;
    } else if (vt_pitch != 0) {
      *(u32x4*)&tile[c / CPR][ch * (16 / sizeof(E))] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  if (vt_pitch == 0) return;   // grid-uniform
  __syncthreads();
  // v^T rows: d-major, 8 tokens (16-bit) per 16-byte store
  constexpr int TPS = 16 / sizeof(E);   // tokens per store
  for (int c = tid; c < HD * (64 / TPS); c += 256) {
    const int d = c / (64 / TPS), tb = (c % (64 / TPS)) * TPS;
    if (t0 + tb >= vt_pitch) continue;
    E tmp[TPS];
#pragma unroll
    for (int u = 0; u < TPS; ++u) tmp[u] = tile[tb + u][d];
    *(u32x4*)(v + (((long)g * H + h) * HD + d) * vt_pitch + t0 + tb) = *(const u32x4*)tmp;
  }
}

int launch_qkv_regroup(const void* qkv, int n_rows, int dtype, int H, const int* idx_q, int G, int Sq, const int* idx_kv, int Skv, void* q, void* k,
                       void* v, int vt_pitch, hipStream_t st) {
  if (G < 1 || H < 1 || Sq < 1 || Skv < 1 || n_rows < 1) return foley_set_err("qkv_regroup: empty problem", __FILE__, __LINE__);
  if (vt_pitch && (vt_pitch % 8 || vt_pitch < Skv || vt_pitch > (Skv + 63) / 64 * 64))
    return foley_set_err("qkv_regroup: the transposed-V pitch must be a multiple of 8 in [Skv, ceil64(Skv)]", __FILE__, __LINE__);
  if (((uintptr_t)qkv | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return foley_set_err("qkv_regroup: operands must be 16-byte aligned", __FILE__, __LINE__);
  const int S = Sq > Skv ? Sq : Skv;
  const dim3 grid((S + 63) / 64, H, G);
  if (dtype == FOLEY_F32) {
    if (vt_pitch) return foley_set_err("qkv_regroup: fp32 operands keep V untransposed", __FILE__, __LINE__);
    FOLEY_LAUNCH(qkv_regroup_kernel<uint32_t>, grid, dim3(256), 0, st, (const uint32_t*)qkv, H, idx_q, Sq, idx_kv, Skv, (uint32_t*)q, (uint32_t*)k, (uint32_t*)v, 0, n_rows);
  } else if (foley_is_half(dtype)) {
    FOLEY_LAUNCH(qkv_regroup_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)qkv, H, idx_q, Sq, idx_kv, Skv, (uint16_t*)q, (uint16_t*)k, (uint16_t*)v, vt_pitch, n_rows);
  } else {
    return foley_set_err("qkv_regroup: fp32, bf16 or fp16 operands", __FILE__, __LINE__);
  }
  FOLEY_LAUNCH_CHECK();
  return 0;
}

// One pass of the separable antialiased uint8 resize the reference pre-processes its frames with (torchvision v2.Resize(bicubic,
// antialias=True) on uint8 CPU tensors, nodes.py:184-196 / utils.py:262-283 - ATen's native uint8 kernel, PIL's scheme): along the
// resized axis every output sample is a fixed-point weighted sum of `xsize` input bytes starting at `xmin`, int16 weights scaled by
// 2^prec, accumulator preset to 2^(prec-1), arithmetic shift, saturate to a byte.  The horizontal pass runs first and its rounded
// uint8 image feeds the vertical pass - integer arithmetic end to end, so the result is the reference's bit for bit
// (tests/test_encoders_gpu.py; the tables are built by host/encoders.py::aa_tables and pinned in tests/test_oracle_golden.py).
// Layout [outer, len, inner]: inner = 1 is the horizontal pass (threads along the output row), inner = W the vertical one
// (threads along x, V bytes each - one 32-bit load per tap when the row pitch allows).  HBM-bound byte work: 37 MB in, 31 MB out
// for the 40 SigLIP2 frames of a 5 s clip; the taps of neighbouring outputs overlap in the L1 / L2.
template <int V>
__global__ __launch_bounds__(256) void resize_aa_u8_kernel(const uint8_t* __restrict__ in, long outer, int len_in, long inner, int len_out,
                                                           const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                           const short* __restrict__ w, int kmax, int prec, uint8_t* __restrict__ out) {
  const long inner_v = inner / V;
  const long total = outer * len_out * inner_v;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const long iv = idx % inner_v, t = idx / inner_v;
    const int xo = (int)(t % len_out);
    const long o = t / len_out;
    int x0 = xmin[xo], n = xsize[xo];   // clamped: a bad table must not read outside the image
    x0 = x0 < 0 ? 0 : (x0 > len_in ? len_in : x0);
    n = n > kmax ? kmax : n;
    n = n > len_in - x0 ? len_in - x0 : n;
    const short* wr = w + (long)xo * kmax;
    const uint8_t* src = in + (o * len_in + x0) * inner + iv * V;
    int acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 1 << (prec - 1);
    for (int j = 0; j < n; ++j) {
      const int wj = wr[j];
      if constexpr (V == 4) {
        const uint32_t px = *(const uint32_t*)(src + (long)j * inner);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += (int)((px >> (8 * v)) & 255u) * wj;
      } else if constexpr (V == 2) {
        const uint32_t px = *(const uint16_t*)(src + (long)j * inner);
        acc[0] += (int)(px & 255u) * wj;
        acc[1] += (int)(px >> 8) * wj;
      } else {
        acc[0] += (int)src[(long)j * inner] * wj;
      }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      int r = acc[v] >> prec;
      r = r < 0 ? 0 : (r > 255 ? 255 : r);
      packed |= (uint32_t)r << (8 * v);
    }
    uint8_t* dst = out + (o * len_out + xo) * inner + iv * V;
    if constexpr (V == 4) *(uint32_t*)dst = packed;
    else if constexpr (V == 2) *(uint16_t*)dst = (uint16_t)packed;
    else *dst = (uint8_t)packed;
  }
}

int launch_resize_aa_u8(const uint8_t* in, long outer, int len_in, long inner, int len_out, const int* xmin, const int* xsize,
                        const short* w, int kmax, int prec, uint8_t* out, hipStream_t st) {
  if (outer < 1 || inner < 1 || len_in < 1 || len_out < 1 || kmax < 1) return foley_set_err("resize_aa_u8: empty problem", __FILE__, __LINE__);
  if (prec < 1 || prec > 22) return foley_set_err("resize_aa_u8: weight precision must be in [1, 22] bits", __FILE__, __LINE__);
  const int V = (inner % 4 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 3)) ? 4 : (inner % 2 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 1)) ? 2 : 1;
  const long total = outer * len_out * (inner / V);
  const long blocks = (total + 255) / 256;
  const dim3 grid((unsigned)(blocks < 65536L * 16 ? blocks : 65536L * 16));
  if (V == 4) FOLEY_LAUNCH(resize_aa_u8_kernel<4>, grid, dim3(256), 0, st, in, outer, len_in, inner, len_out, xmin, xsize, w, kmax, prec, out);
  else if (V == 2) FOLEY_LAUNCH(resize_aa_u8_kernel<2>, grid, dim3(256), 0, st, in, outer, len_in, inner, len_out, xmin, xsize, w, kmax, prec, out);
  else FOLEY_LAUNCH(resize_aa_u8_kernel<1>, grid, dim3(256), 0, st, in, outer, len_in, inner, len_out, xmin, xsize, w, kmax, prec, out);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_rows_to_planes(const float* rows, int B, int T, int C, float* out, hipStream_t st) {
  FOLEY_LAUNCH(rows_to_planes_kernel, dim3(grid1d((long)B * T * C, 256)), dim3(256), 0, st, rows, B, T, C, out);
  FOLEY_LAUNCH_CHECK();
  return 0;
}

int launch_dac_out(const float* s, const float* w, const float* bias, int B, int T, int C, float* out,
                   hipStream_t st) {
  if (C % 4 || C > 256) return foley_set_err("dac_out: unsupported channel count", __FILE__, __LINE__);
  const size_t sh = (70 * (C + 1) + 7 * C) * sizeof(float);
  FOLEY_LAUNCH(dac_out_kernel, dim3((T + 63) / 64, B), dim3(256), sh, st, s, w, bias, T, C, out);
  FOLEY_LAUNCH_CHECK();
  return 0;
}
