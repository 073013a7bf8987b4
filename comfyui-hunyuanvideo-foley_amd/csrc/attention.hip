// Flash-style attention for gfx950, head_dim = 128, no mask (reference: attn_layers.py:419-422,
// hifi_foley.py:383).  fp32 operands on v_mfma_f32_32x32x2_f32, online softmax in fp32.
//
// One wavefront owns 32 query rows and walks the keys in tiles of 32.  The score tile is computed
// *transposed* (S^T = K Q^T) so that every lane holds 16 keys of ONE query column: the row max /
// row sum are lane-local plus a single lane^32 exchange, and the exponentiated scores already sit
// in the B-operand layout of the second MFMA (O^T += V^T P^T) - no LDS, no cross-lane shuffles.
//   MFMA 32x32x2 operand layout: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
//   D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31].
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace {

// head dim: 128 (the Foley DiT) or 64 (the conditioning encoders: Synchformer / SigLIP2 ViT-B, host/encoders.py) - a template
// parameter of the fp32 kernel and of the 16-bit wide kernel

// 1-D grid -> (query tile, head, batch) with a bijective XCD-aware remap: workgroup id i runs on XCD
// i % 8, so ids are regrouped such that all query tiles of one (batch, head) - which read the same
// K / V - land on the same XCD and share its L2 instead of fetching the operands once per XCD.
__device__ __forceinline__ void attn_block_coords(const AttnArgs& a, int qtile, int& qt, int& h, int& b) {
  const int nq = (a.Sq + qtile - 1) / qtile;
  const int nwg = nq * a.H * a.Bq;
  int bid = (int)blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  qt = bid % nq;
  const int bh = bid / nq;
  h = bh % a.H;
  b = bh / a.H;
}

template <typename OutT> struct Pack4Out;
template <> struct Pack4Out<float> {
  static __device__ __forceinline__ void store(float* p, const f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Pack4Out<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* p, const f32x4 v) {
    uint2 w;
    w.x = pack_bf16x2(v[0], v[1]);
    w.y = pack_bf16x2(v[2], v[3]);
    *(uint2*)p = w;
  }
};

template <> struct Pack4Out<f16_t> {
  static __device__ __forceinline__ void store(f16_t* p, const f32x4 v) {
    uint2 w;
    w.x = pack_f16x2(v[0], v[1]);
    w.y = pack_f16x2(v[2], v[3]);
    *(uint2*)p = w;
  }
};

template <typename OutT, int HD>
__global__ __launch_bounds__(64) void attn_kernel(const AttnArgs a) {
  const int lane = threadIdx.x;
  const int j = lane & 31, kh = lane >> 5;
  const int q0 = blockIdx.x * 32;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bk = b / a.kv_bdiv;
  const float* __restrict__ Q = (const float*)a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const float* __restrict__ K = (const float*)a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const float* __restrict__ V = (const float*)a.v + ((long)(bk * a.H + h) * a.Skv) * HD;
  const float scale = HD == 128 ? 0.08838834764831845f : 0.125f;  // 1/sqrt(HD)

  // B operand of S^T = K Q^T: lane (j, kh) holds Q[q0 + j][kh*HD/2 .. +HD/2-1]
  constexpr int HH = HD / 2, NC = HD / 8;   // floats / float4 chunks of a half head
  float qr[HH];
  {
    const int qrow = min(q0 + j, a.Sq - 1);
    const f32x4* p = (const f32x4*)(Q + (long)qrow * HD + kh * HH);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const f32x4 v = p[c];
      qr[c * 4 + 0] = v[0]; qr[c * 4 + 1] = v[1]; qr[c * 4 + 2] = v[2]; qr[c * 4 + 3] = v[3];
    }
  }

  constexpr int ND = HD / 32;   // 32-wide output fragments
  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt = 0; kt < a.Skv; kt += 32) {
    // A operand: lane (i = j, kh) holds K[kt + i][kh*64 .. +63]
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    {
      const int krow = min(kt + j, a.Skv - 1);
      const f32x4* p = (const f32x4*)(K + (long)krow * HD + kh * HH);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 v = p[c];
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], qr[c * 4 + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], qr[c * 4 + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], qr[c * 4 + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3], qr[c * 4 + 3], s, 0, 0, 0);
      }
    }
    // s[e] = score(key kt + r(e), query q0 + j), r(e) = (e&3) + 8*(e>>2) + 4*kh
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt + (e & 3) + 8 * (e >> 2) + 4 * kh;
      s[e] = (key < a.Skv) ? s[e] * scale : -INFINITY;
      mx = fmaxf(mx, s[e]);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);     // finite: every tile holds at least one valid key
    const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = expf(s[e] - m_new);
      ps += s[e];
    }
    ps = xhalf_sum(ps);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    // O^T[d][q] += sum_key V[key][d] * P[q][key]: A operand lane (i = j, kh) = V[kt + r(t)][d0 + i],
    // B operand = s[t] (already in place).
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int vrow = min(kt + (t & 3) + 8 * (t >> 2) + 4 * kh, a.Skv - 1);
      const float* vp = V + (long)vrow * HD + j;
#pragma unroll
      for (int d = 0; d < ND; ++d)
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[d * 32], s[t], o[d], 0, 0, 0);
    }
  }

  const int tok = q0 + j;
  if (tok >= a.Sq) return;
  const float inv = 1.0f / l_run;
  OutT* dst;
  if (HD == 64 && a.out_rows) {   // HD is a template constant: the DiT's instances carry no trace of it
    const int orow = a.out_rows[(long)b * a.Sq + tok];
    if ((unsigned)orow >= (unsigned)a.out_nrows) return;   // a row outside out [out_nrows, H*64] is dropped, never written (caller-supplied table)
    dst = (OutT*)a.outA + (long)orow * (a.H * HD);
  }
  else if (tok < a.split) dst = (OutT*)a.outA + ((long)b * a.split + tok) * (a.H * HD);
  else dst = (OutT*)a.outB + ((long)b * (a.Sq - a.split) + (tok - a.split)) * (a.H * HD);
  dst += h * HD;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int c = d * 32 + 8 * g4 + 4 * kh;
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[c + u] = Cvt<OutT>::to(o[d][g4 * 4 + u] * inv);
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 throughput variant: Q, K [B,H,S,128] bf16 and V *transposed* [B,H,128,vt_pitch] bf16 (written
// that way by the head-split kernel), v_mfma_f32_32x32x16_bf16, fp32 softmax/accumulate.
// A workgroup = 4 waves sharing one 32-query block; each wave walks a quarter of the key tiles and
// the four partial (m, l, O) triples are merged through LDS (flash-decoding style), which gives
// 4x the waves of the fp32 kernel on these short sequences (S <= 3480, typically 250-290).
// Key rows are fed to the first MFMA in a permuted order (pi) chosen so that each lane's 16
// scores belong to 16 CONSECUTIVE keys: the exponentiated scores then form, in place, the
// key-contiguous B operand of the second MFMA, and V^T rows are plain 16-byte loads.
template <typename T, typename OutT>
__global__ __launch_bounds__(256) void attn_bf16_kernel(const AttnArgs a) {
  constexpr int HD = 128;
  __shared__ float sO[4][3][16][64];  // partial O of the d-fragments a wave does not finalise itself
  __shared__ float sM[4][32], sL[4][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = lane & 31, kh = lane >> 5;
  int qt, h, b;
  attn_block_coords(a, 32, qt, h, b);
  const int q0 = qt * 32;
  const int bk = b / a.kv_bdiv;
  const T* __restrict__ Q = (const T*)a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const T* __restrict__ K = (const T*)a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const T* __restrict__ VT = (const T*)a.v + ((long)(bk * a.H + h) * HD) * a.vt_pitch;
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128)

  const bool stamp = a.dbg && threadIdx.x == 0;
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 0] = wall_clock64();
  bf16x8 qf[8];
  {
    const T* p = Q + (long)min(q0 + j, a.Sq - 1) * HD + 8 * kh;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(p + 16 * s);
  }
  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nt = (a.Skv + 31) >> 5;
  const int t0 = (nt * w) >> 2, t1 = (nt * (w + 1)) >> 2;
  const int pi = 16 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);  // A-row j carries key kt + pi
  // one 32-key tile: scores (K fragments kf), online softmax, P V (V^T fragments vf)
  auto tile = [&](int kt, const bf16x8 (&kf)[8], const bf16x8 (&vf)[2][4]) {
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) s = mfma16<T>(kf[st], qf[st], s);
    // s[e] = score(key kt + 16*kh + e, query q0 + j), kept in the log2 domain (scale2 = log2(e)/sqrt(128)):
    // the exponentials are single v_exp_f32 instructions
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = (kt + 16 * kh + e < a.Skv) ? s[e] * scale2 : -INFINITY;
      mx = fmaxf(mx, s[e]);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    float ps = 0.f;
    bf16x8 pb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(s[e] - m_new);
      ps += pv;
      pb[e >> 3][e & 7] = to_carrier<T>(pv);
    }
    ps = xhalf_sum(ps);
    // the 64 accumulator rescales only when some query's running maximum moved (rare after the first tiles)
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
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
  };
  auto load_k = [&](int kt, bf16x8 (&kf)[8]) {
    const T* p = K + (long)min(kt + pi, a.Skv - 1) * HD + 8 * kh;
#pragma unroll
    for (int st = 0; st < 8; ++st) kf[st] = *(const bf16x8*)(p + 16 * st);
  };
  auto load_v = [&](int kt, bf16x8 (&vf)[2][4]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int d = 0; d < 4; ++d) vf[u][d] = *(const bf16x8*)(VT + (long)(d * 32 + j) * a.vt_pitch + kt + 16 * kh + 8 * u);
  };
  if (!a.no_preload && t1 - t0 <= 2) {
    // The small grids this kernel serves (S <= ~300: one or two key tiles per wave) are a chain of
    // dependent memory latencies - K tile, then V tile, per key tile.  Every operand of the wave's
    // (at most two) tiles is requested up front, before the first MFMA: one round trip instead of four.
    bf16x8 k0[8], k1[8], v0[2][4], v1[2][4];
    const int kta = t0 * 32, ktb = min(t0 + 1, nt - 1) * 32;   // ktb is only consumed when the wave owns two tiles
    if (t1 > t0) {
      load_k(kta, k0);
      load_k(ktb, k1);
      load_v(kta, v0);
      load_v(ktb, v1);
      if (a.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamp) a.dbg[(long)blockIdx.x * 8 + 1] = wall_clock64();
      }
      tile(kta, k0, v0);
      if (t1 - t0 == 2) tile(ktb, k1, v1);
    }
  } else {
    for (int t = t0; t < t1; ++t) {
      bf16x8 kf[8], vf[2][4];
      load_k(t * 32, kf);
      load_v(t * 32, vf);
      tile(t * 32, kf, vf);
    }
  }

  if (stamp) a.dbg[(long)blockIdx.x * 8 + 2] = wall_clock64();
  // merge the four key ranges: wave w finalises d-fragment w
  if (kh == 0) {
    sM[w][j] = m_run;
    sL[w][j] = l_run;
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    if (d == w) continue;
    const int slot = d < w ? d : d - 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) sO[w][slot][e][lane] = o[d][e];
  }
  __syncthreads();
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 3] = wall_clock64();
  float mstar = -INFINITY;
#pragma unroll
  for (int x = 0; x < 4; ++x) mstar = fmaxf(mstar, sM[x][j]);
  float wt[4], L = 0.f;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    wt[x] = __builtin_amdgcn_exp2f(sM[x][j] - mstar);   // running maxima live in the log2 domain
    L += wt[x] * sL[x][j];
  }
  const int tok = q0 + j;
  if (tok >= a.Sq) return;
  const float inv = 1.0f / L;
  OutT* dst;
  if (tok < a.split) dst = (OutT*)a.outA + ((long)b * a.split + tok) * (a.H * HD);
  else dst = (OutT*)a.outB + ((long)b * (a.Sq - a.split) + (tok - a.split)) * (a.H * HD);
  dst += h * HD + w * 32;
  const int slot_mine = 0;  // unused for own fragment
  (void)slot_mine;
  // four consecutive output dims ((e & 3) of one e >> 2 group) leave as ONE 8-byte (bf16) / 16-byte (fp32) store:
  // sixteen 2-byte stores per lane made this tail 3.6 us of a 12 us kernel (tools/attn_timeline.py)
  // One straight-line body per wave index (the index is wave-uniform): with `x == w` tested per element the 48 LDS
  // reads of the partner fragments sat in 48 conditional regions, each waiting for its own read - 2.7 us of latency.
  auto finish = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    float part[3][16];   // the other waves' partial O of d-fragment W
#pragma unroll
    for (int x = 0, n = 0; x < 4; ++x) {
      if (x == W) continue;
      const int slot = W < x ? W : W - 1;
#pragma unroll
      for (int e = 0; e < 16; ++e) part[n][e] = sO[x][slot][e][lane];
      ++n;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 r;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = g4 * 4 + u;
        float acc = 0.f;
#pragma unroll
        for (int x = 0, n = 0; x < 4; ++x) {   // same summation order as before: x = 0..3
          if (x == W) acc += wt[x] * o[W][e];
          else acc += wt[x] * part[n++][e];
        }
        r[u] = acc * inv;
      }
      Pack4Out<OutT>::store(dst + 8 * g4 + 4 * kh, r);
    }
  };
  switch (__builtin_amdgcn_readfirstlane(w)) {
    case 0: finish(std::integral_constant<int, 0>{}); break;
    case 1: finish(std::integral_constant<int, 1>{}); break;
    case 2: finish(std::integral_constant<int, 2>{}); break;
    default: finish(std::integral_constant<int, 3>{}); break;
  }
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 4] = wall_clock64();
}

// The same kernel with its operands staged through LDS by direct-to-LDS DMA (small grids, Skv * 256 + V^T image + Q tile
// within 160 KiB: the single-stream blocks' self-attention at 5 s and the cross-attention to the 77 text keys).
__device__ __forceinline__ void attn_buf_lds16(const void* base, unsigned bytes, unsigned char* lds_wave_base, int voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

template <typename T, typename OutT>
__global__ __launch_bounds__(256) void attn_lds_kernel(const AttnArgs a, const int merge_off) {
  constexpr int HD = 128;
  // ONE dynamic LDS array (a second __shared__ object makes hipcc drain vmcnt in front of every ds_read of a direct-to-LDS
  // pipeline): [K image | V^T image | Q image], and the merge area of the four key ranges at `merge_off` - behind the
  // images when that fits 160 KiB, else aliased onto them (offset 0) after a barrier
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float (*sO)[3][16][64] = (float (*)[3][16][64])(dsm + merge_off);   // partial O of the d-fragments a wave does not finalise itself
  float (*sM)[32] = (float (*)[32])(dsm + merge_off + 4 * 3 * 16 * 64 * 4);
  float (*sL)[32] = (float (*)[32])(dsm + merge_off + 4 * 3 * 16 * 64 * 4 + 4 * 32 * 4);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = lane & 31, kh = lane >> 5;
  int qt, h, b;
  attn_block_coords(a, 32, qt, h, b);
  const int q0 = qt * 32;
  const int bk = b / a.kv_bdiv;
  const T* __restrict__ Q = (const T*)a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const T* __restrict__ K = (const T*)a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const T* __restrict__ VT = (const T*)a.v + ((long)(bk * a.H + h) * HD) * a.vt_pitch;
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128)

  const bool stamp = a.dbg && threadIdx.x == 0;
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 0] = wall_clock64();
  // ---- staging: every wave issues its share of 1 KiB direct-to-LDS pieces (full 128-byte lines; the fragment-shaped
  // global loads of attn_bf16_kernel touch 32 bytes of each line, which is what its operand phase is bound by).  LDS rows are
  // unpadded powers of two; bank conflicts are avoided by permuting the SOURCE chunk (p ^ f(row)) and applying the same XOR
  // at fragment-read time.  Rows past the end of an operand lie beyond the buffer range and arrive as zeros.
  const int nt = (a.Skv + 31) >> 5;
  const int lgCV = a.vt_pitch > 128 ? 5 : 4;                    // 16-byte chunks per LDS row of the V^T image (16 or 32)
  unsigned char* const Ks = dsm;
  unsigned char* const Vs = Ks + nt * 32 * 256;
  unsigned char* const Qs = Vs + (128 << (lgCV + 4));
  const int wv = __builtin_amdgcn_readfirstlane(w);
  {
    const int r4 = lane >> 4, c16 = lane & 15;
    for (int p = wv; p < nt * 8; p += 4) {                       // K: pieces of 4 rows x 256 B
      const int row = 4 * p + r4;
      const int g = c16 ^ ((row & 7) | ((row >> 1) & 8));
      attn_buf_lds16(K, (unsigned)a.Skv * 256u, Ks + p * 1024, row * 256 + g * 16);
    }
    const int vchunks = a.vt_pitch >> 3;                         // 16-byte chunks per V^T row in memory
    for (int p = wv; p < (128 << lgCV) >> 6; p += 4) {           // V^T: 128 rows x (16 << lgCV) B
      const int pos = p * 64 + lane;
      const int row = pos >> lgCV, g = (pos & ((1 << lgCV) - 1)) ^ (row & 15);
      attn_buf_lds16(VT, (unsigned)(HD * a.vt_pitch * 2), Vs + p * 1024, g < vchunks ? (row * a.vt_pitch + g * 8) * 2 : 0x7ffffff0);
    }
    for (int p = wv; p < 8; p += 4) {                            // Q tile: 32 rows x 256 B
      const int row = 4 * p + r4;
      attn_buf_lds16(Q + (long)q0 * HD, (unsigned)(a.Sq - q0) * 256u, Qs + p * 1024, row * 256 + ((c16 ^ (row & 15)) << 4));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 1] = wall_clock64();
  bf16x8 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s] = *(const bf16x8*)(Qs + j * 256 + (((2 * s + kh) ^ (j & 15)) << 4));
  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int t0 = (nt * w) >> 2, t1 = (nt * (w + 1)) >> 2;
  const int pi = 16 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);  // A-row j carries key kt + pi
  // one 32-key tile: scores (K fragments kf), online softmax, P V (V^T fragments vf)
  auto tile = [&](int kt, const bf16x8 (&kf)[8], const bf16x8 (&vf)[2][4]) {
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) s = mfma16<T>(kf[st], qf[st], s);
    // s[e] = score(key kt + 16*kh + e, query q0 + j), kept in the log2 domain (scale2 = log2(e)/sqrt(128)):
    // the exponentials are single v_exp_f32 instructions
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = (kt + 16 * kh + e < a.Skv) ? s[e] * scale2 : -INFINITY;
      mx = fmaxf(mx, s[e]);
    }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    float ps = 0.f;
    bf16x8 pb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(s[e] - m_new);
      ps += pv;
      pb[e >> 3][e & 7] = to_carrier<T>(pv);
    }
    ps = xhalf_sum(ps);
    // the 64 accumulator rescales only when some query's running maximum moved (rare after the first tiles)
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
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
  };
  const int pk = (pi & 7) | ((pi >> 1) & 8);                    // source permutation of key row kt + pi (kt is a multiple of 32)
  auto read_k = [&](int kt, bf16x8 (&kf)[8]) {
    const unsigned char* kr = Ks + (kt + pi) * 256;
#pragma unroll
    for (int st = 0; st < 8; ++st) kf[st] = *(const bf16x8*)(kr + (((2 * st + kh) ^ pk) << 4));
  };
  auto read_v = [&](int kt, bf16x8 (&vf)[2][4]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int row = d * 32 + j;
        vf[u][d] = *(const bf16x8*)(Vs + (row << (lgCV + 4)) + ((((kt >> 3) + 2 * kh + u) ^ (row & 15)) << 4));
      }
  };
  if (t1 - t0 <= 2) {
    // the usual case (<= 8 key tiles): the fragments of both tiles are requested before the first MFMA - one LDS round trip
    bf16x8 k0[8], k1[8], v0[2][4], v1[2][4];
    const int kta = t0 * 32, ktb = min(t0 + 1, nt - 1) * 32;
    if (t1 > t0) {
      read_k(kta, k0);
      read_k(ktb, k1);
      read_v(kta, v0);
      read_v(ktb, v1);
      tile(kta, k0, v0);
      if (t1 - t0 == 2) tile(ktb, k1, v1);
    }
  } else {
    for (int t = t0; t < t1; ++t) {
      bf16x8 kf[8], vf[2][4];
      read_k(t * 32, kf);
      read_v(t * 32, vf);
      tile(t * 32, kf, vf);
    }
  }

  if (stamp) a.dbg[(long)blockIdx.x * 8 + 2] = wall_clock64();
  // merge the four key ranges: wave w finalises d-fragment w
  if (merge_off == 0) __syncthreads();   // the merge area aliases the images: every wave is done reading them
  if (kh == 0) {
    sM[w][j] = m_run;
    sL[w][j] = l_run;
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    if (d == w) continue;
    const int slot = d < w ? d : d - 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) sO[w][slot][e][lane] = o[d][e];
  }
  __syncthreads();
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 3] = wall_clock64();
  float mstar = -INFINITY;
#pragma unroll
  for (int x = 0; x < 4; ++x) mstar = fmaxf(mstar, sM[x][j]);
  float wt[4], L = 0.f;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    wt[x] = __builtin_amdgcn_exp2f(sM[x][j] - mstar);   // running maxima live in the log2 domain
    L += wt[x] * sL[x][j];
  }
  const int tok = q0 + j;
  if (tok >= a.Sq) return;
  const float inv = 1.0f / L;
  OutT* dst;
  if (tok < a.split) dst = (OutT*)a.outA + ((long)b * a.split + tok) * (a.H * HD);
  else dst = (OutT*)a.outB + ((long)b * (a.Sq - a.split) + (tok - a.split)) * (a.H * HD);
  dst += h * HD + w * 32;
  const int slot_mine = 0;  // unused for own fragment
  (void)slot_mine;
  // four consecutive output dims ((e & 3) of one e >> 2 group) leave as ONE 8-byte (bf16) / 16-byte (fp32) store:
  // sixteen 2-byte stores per lane made this tail 3.6 us of a 12 us kernel (tools/attn_timeline.py)
  // One straight-line body per wave index (the index is wave-uniform): with `x == w` tested per element the 48 LDS
  // reads of the partner fragments sat in 48 conditional regions, each waiting for its own read - 2.7 us of latency.
  auto finish = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    float part[3][16];   // the other waves' partial O of d-fragment W
#pragma unroll
    for (int x = 0, n = 0; x < 4; ++x) {
      if (x == W) continue;
      const int slot = W < x ? W : W - 1;
#pragma unroll
      for (int e = 0; e < 16; ++e) part[n][e] = sO[x][slot][e][lane];
      ++n;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 r;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = g4 * 4 + u;
        float acc = 0.f;
#pragma unroll
        for (int x = 0, n = 0; x < 4; ++x) {   // same summation order as before: x = 0..3
          if (x == W) acc += wt[x] * o[W][e];
          else acc += wt[x] * part[n++][e];
        }
        r[u] = acc * inv;
      }
      Pack4Out<OutT>::store(dst + 8 * g4 + 4 * kh, r);
    }
  };
  switch (__builtin_amdgcn_readfirstlane(w)) {
    case 0: finish(std::integral_constant<int, 0>{}); break;
    case 1: finish(std::integral_constant<int, 1>{}); break;
    case 2: finish(std::integral_constant<int, 2>{}); break;
    default: finish(std::integral_constant<int, 3>{}); break;
  }
  if (stamp) a.dbg[(long)blockIdx.x * 8 + 4] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------
// bf16, large grids: a workgroup = 4 waves x 32 queries = 128 queries; every wave walks ALL key tiles
// and the K / V^T tiles are staged once per workgroup in LDS (register-staged double buffer), so
// the operands are read from L2 once per 128 queries instead of once per 32 (the narrow kernel above
// re-reads them for each of its 32-query workgroups: ~10x at S = 290).  No cross-wave merge.
// LDS rows are padded (K: 272 B, V^T: 80 B) so the 16-lane groups of ds_read_b128 hit distinct banks.
// GRP (head dim 64): block-diagonal attention over packed small groups (AttnArgs::grp_q / grp_kv): a query only sees the keys of its own
// group, a wave skips the key tiles none of its 32 queries can see, rows with no visible key in a tile contribute nothing.
template <typename T, typename OutT, int HD, bool GRP = false>
// Three waves per SIMD: left to itself the compiler takes 166 VGPRs + 80 AGPRs (two waves per SIMD); capped at 168 registers the
// kernel needs no AGPRs and no scratch, and three resident workgroups per CU hide each other's LDS / MFMA / softmax latencies:
// S = 290, 16 clips: 39.0 -> 29.3 us; S = 1740 (30 s clip): 99 -> 77 us (tools/attn_bench.py).  Four (128 registers) spills.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bf16_wide_kernel(const AttnArgs a) {
  constexpr int KP = 2 * HD + 16, VP = 80;               // LDS row pitches in bytes
  constexpr int STG = 32 * KP + HD * VP;                 // one stage: K tile + V^T tile
  constexpr int NS = HD / 16, ND = HD / 32;              // k-steps of Q K^T, 32-wide output fragments
  constexpr int CK = HD / 8, NP = HD / 64;               // 16-byte chunks per K row, staging pieces per thread and tile
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  int qt, h, b;
  attn_block_coords(a, 128, qt, h, b);
  const int q0 = qt * 128 + w * 32;
  const int bk = b / a.kv_bdiv;
  const T* __restrict__ Q = (const T*)a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const T* __restrict__ K = (const T*)a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const T* __restrict__ VT = (const T*)a.v + ((long)(bk * a.H + h) * HD) * a.vt_pitch;
  const float scale2 = (HD == 128 ? 0.08838834764831845f : 0.125f) * 1.4426950408889634f;   // log2(e) / sqrt(HD)

  bf16x8 qf[NS];
  {
    const T* p = Q + (long)min(q0 + j, a.Sq - 1) * HD + 8 * kh;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8*)(p + 16 * s);
  }
  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  int klo = 0, khi = 0, wlo = 0, whi = 0;   // GRP: this lane's visible keys [klo, khi), the union over the wave's queries [wlo, whi)
  if constexpr (GRP) {
    const int qi = min(q0 + j, a.Sq - 1);
    klo = (qi / a.grp_q) * a.grp_kv;
    khi = klo + a.grp_kv;
    if (q0 < a.Sq) {
      wlo = (q0 / a.grp_q) * a.grp_kv;
      whi = (min(q0 + 31, a.Sq - 1) / a.grp_q + 1) * a.grp_kv;
    }
  }

  // staging: 4*HD + 4*HD pieces of 16 B per tile, NP + NP per thread
  const int k_row0 = tid / CK, k_col = tid % CK;       // K tile: 32 rows x CK chunks, 256 / CK rows per pass
  const int v_row0 = tid >> 2, v_col = tid & 3;        // V^T tile: HD rows x 4 chunks, 64 rows per pass
  u32x4 rk[NP], rv[NP];
  auto gload = [&](int t) {
    const int kt = t * 32;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      rk[i] = *(const u32x4*)(K + (long)min(kt + k_row0 + i * (256 / CK), a.Skv - 1) * HD + k_col * 8);
      rv[i] = *(const u32x4*)(VT + (long)(v_row0 + i * 64) * a.vt_pitch + kt + v_col * 8);
    }
  };
  auto lstore = [&](int stage) {
    unsigned char* base = lds + stage * STG;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *(u32x4*)(base + (k_row0 + i * (256 / CK)) * KP + k_col * 16) = rk[i];
      *(u32x4*)(base + 32 * KP + (v_row0 + i * 64) * VP + v_col * 16) = rv[i];
    }
  };
  const int nt = (a.Skv + 31) >> 5;
  const int pi = 16 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);  // A-row j carries key kt + pi
  const bool acct = a.dbg && tid == 0;   // tools/attn_timeline.py --wide: cycles in the math and in the staging + barrier
  long long c_math = 0, c_stage = 0, c_t0 = acct ? (long long)__builtin_readcyclecounter() : 0;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int kt = t * 32;
    const long long ca = acct ? (long long)__builtin_readcyclecounter() : 0;
    if (t + 1 < nt) gload(t + 1);                 // next tile's global loads fly during this tile's math
    const unsigned char* Ks = lds + (t & 1) * STG;
    const unsigned char* Vs = Ks + 32 * KP;
    bool live = true;
    if constexpr (GRP) live = kt < whi && kt + 32 > wlo;     // wave-uniform
    if (live) {
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st)
      s = mfma16<T>(*(const bf16x8*)(Ks + pi * KP + (16 * st + 8 * kh) * 2), qf[st], s);
    // s[e] = score(key kt + 16*kh + e, query q0 + j), kept in the log2 domain (scale2 = log2(e)/sqrt(128)):
    // the exponentials are single v_exp_f32 instructions
    // raw scores: the scale (log2(e)/sqrt(128) > 0) commutes with the maximum and is folded into the exponent's FMA;
    // only a tile that sticks out of the sequence masks its elements (wave-uniform test)
    float mx = -INFINITY;
    if constexpr (GRP) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt + 16 * kh + e;
        s[e] = (key >= klo && key < khi) ? s[e] : -INFINITY;
      }
    } else if (kt + 32 > a.Skv) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = (kt + 16 * kh + e < a.Skv) ? s[e] : -INFINITY;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx * scale2);
    const float m_exp = (GRP && m_new == -INFINITY) ? 0.f : m_new;   // GRP: no visible key so far - exp2(-inf - 0) = 0, not exp2(-inf + inf)
    float ps = 0.f;
    bf16x8 pb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], scale2, -m_exp));
      ps += pv;
      pb[e >> 3][e & 7] = to_carrier<T>(pv);
    }
    ps = xhalf_sum(ps);
    // the 64 accumulator rescales only when some query's running maximum moved (rare after the first tiles)
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
      const float alpha = (GRP && m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    }
    l_run += ps;
    m_run = m_new;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int d = 0; d < ND; ++d)
        o[d] = mfma16<T>(*(const bf16x8*)(Vs + (d * 32 + j) * VP + (16 * kh + 8 * u) * 2), pb[u], o[d]);
    }   // live
    const long long cb = acct ? (long long)__builtin_readcyclecounter() : 0;
    if (t + 1 < nt) {
      lstore((t + 1) & 1);        // stage (t+1)&1 was last read in iteration t-1: every wave passed the barrier below since
      __syncthreads();
    }
    if (acct) {
      const long long cc = (long long)__builtin_readcyclecounter();
      c_math += cb - ca;
      c_stage += cc - cb;
    }
  }
  if (acct) {
    a.dbg[(long)blockIdx.x * 8 + 0] = (long long)__builtin_readcyclecounter() - c_t0;
    a.dbg[(long)blockIdx.x * 8 + 1] = c_math;
    a.dbg[(long)blockIdx.x * 8 + 2] = c_stage;
    a.dbg[(long)blockIdx.x * 8 + 3] = nt;
  }

  const int tok = q0 + j;
  if (tok >= a.Sq) return;
  const float inv = 1.0f / l_run;
  OutT* dst;
  if (HD == 64 && a.out_rows) {   // HD is a template constant: the DiT's instances carry no trace of it
    const int orow = a.out_rows[(long)b * a.Sq + tok];
    if ((unsigned)orow >= (unsigned)a.out_nrows) return;   // a row outside out [out_nrows, H*64] is dropped, never written (caller-supplied table)
    dst = (OutT*)a.outA + (long)orow * (a.H * HD);
  }
  else if (tok < a.split) dst = (OutT*)a.outA + ((long)b * a.split + tok) * (a.H * HD);
  else dst = (OutT*)a.outB + ((long)b * (a.Sq - a.split) + (tok - a.split)) * (a.H * HD);
  dst += h * HD;
  // lane (j, kh) holds dims d*32 + 8*g4 + 4*kh + {0..3} of query j: 4 consecutive outputs per store
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = {o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv, o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv};
      Pack4Out<OutT>::store(dst + d * 32 + 8 * g4 + 4 * kh, v);
    }
}

// ---------------------------------------------------------------------------------------------
// bf16, LONG sequences on small grids (the 30 s clip: S = 1500 .. 1740, 2 x 12 heads -> 336 workgroups of 128 queries = 1.3
// waves per SIMD): with so few waves the kernel's time is ONE wave's dependent chain per key tile - QK^T -> two lane-half
// exchanges -> exponentials -> PV -> staging write -> barrier, 1 880 cycles for 512 cycles of MFMA in the kernel above - times
// the number of tiles.  This form walks the keys 64 at a time: the chain's fixed latencies (MFMA drain, the cross-half
// exchanges, the rescale test, the staging write and the barrier) are paid once per 64 keys, the two 32-key score tiles
// are independent MFMA chains, and the lane-half exchanges are v_permlane32_swap (one VALU instruction each; __shfl_xor
// compiles to ds_bpermute, a round trip through the LDS crossbar).  Same operand layout, staging (register-staged double
// buffer, padded pitches) and output mapping as attn_bf16_wide_kernel; 70 KiB of LDS, two workgroups per CU.
// Measured (tools/attn_bench.py): S = 1740 75.6 -> 71.8 us, S = 1500 64.0 -> 61.1 us - the chain is not latency- but
// instruction-bound: rocprofv3 SQ counters give, per wave, VALU 28 % (278 VALU instructions per 64 keys: exponentials,
// maxima, packing, staging addresses), matrix pipe 26 %, parked at a wait 39 %, issue stalls 24 %, at 1.3 waves per SIMD.
// A split-KV form on top of it (two workgroups per query tile, each half of the keys, partial softmax states merged by the
// second arriver through a workspace record with an agent-scope release / acquire hand-off) was built, correct, and SLOWER
// (91 vs 72 us): at 208 registers and 70 KiB only two workgroups fit a CU, so its 672 workgroups run in 1.3 rounds.
template <typename T, typename OutT>
__global__ __launch_bounds__(256, 2) void attn_bf16_long_kernel(const AttnArgs a) {   // two waves per SIMD (<= 256 registers): left alone the compiler takes 194 + 96 AGPRs = one workgroup per CU
  constexpr int HD = 128, KT = 64;
  constexpr int KP = 2 * HD + 16, VP = 2 * KT + 16;      // LDS row pitches in bytes (K rows: 272, V^T rows: 144)
  constexpr int STG = KT * KP + HD * VP;                 // one stage: K tile + V^T tile = 35 840 B
  constexpr int NS = HD / 16, ND = HD / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  int qt, h, b;
  attn_block_coords(a, 128, qt, h, b);
  const int q0 = qt * 128 + w * 32;
  const int bk = b / a.kv_bdiv;
  const T* __restrict__ Q = (const T*)a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const T* __restrict__ K = (const T*)a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const T* __restrict__ VT = (const T*)a.v + ((long)(bk * a.H + h) * HD) * a.vt_pitch;
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128)

  bf16x8 qf[NS];
  {
    const T* p = Q + (long)min(q0 + j, a.Sq - 1) * HD + 8 * kh;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8*)(p + 16 * s);
  }
  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: K tile 64 rows x 16 chunks, V^T tile 128 rows x 8 chunks = 1024 + 1024 pieces of 16 B, 4 + 4 per thread
  const int k_row0 = tid >> 4, k_col = tid & 15;       // 16 rows per pass
  const int v_row0 = tid >> 3, v_col = tid & 7;        // 32 rows per pass
  const int vmax = a.vt_pitch - 8;                     // last 16-byte chunk of a V^T row (the pitch covers Skv rounded up to 32, not to 64)
  u32x4 rk[4], rv[4];
  auto gload = [&](int t) {
    const int kt = t * KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rk[i] = *(const u32x4*)(K + (long)min(kt + k_row0 + i * 16, a.Skv - 1) * HD + k_col * 8);
      rv[i] = *(const u32x4*)(VT + (long)(v_row0 + i * 32) * a.vt_pitch + min(kt + v_col * 8, vmax));   // clamped chunks carry masked (p = 0) keys
    }
  };
  auto lstore = [&](int stage) {
    unsigned char* base = lds + stage * STG;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(base + (k_row0 + i * 16) * KP + k_col * 16) = rk[i];
      *(u32x4*)(base + KT * KP + (v_row0 + i * 32) * VP + v_col * 16) = rv[i];
    }
  };
  const int nt = (a.Skv + KT - 1) / KT;
  const int pi = 16 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);  // A-row j of a 32-key score tile carries key pi
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int kt = t * KT;
    if (t + 1 < nt) gload(t + 1);                 // next tile's global loads fly during this tile's math
    const unsigned char* Ks = lds + (t & 1) * STG;
    const unsigned char* Vs = Ks + KT * KP;
    f32x16 s0, s1;
#pragma unroll
    for (int e = 0; e < 16; ++e) s0[e] = s1[e] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st) {             // two independent score tiles: keys kt .. kt+31 and kt+32 .. kt+63
      s0 = mfma16<T>(*(const bf16x8*)(Ks + pi * KP + (16 * st + 8 * kh) * 2), qf[st], s0);
      s1 = mfma16<T>(*(const bf16x8*)(Ks + (32 + pi) * KP + (16 * st + 8 * kh) * 2), qf[st], s1);
    }
    // s0[e] / s1[e] = raw score(key kt + {0, 32} + 16*kh + e, query q0 + j); log2 domain through scale2 (> 0: commutes with max)
    if (kt + KT > a.Skv) {                        // only a tile that sticks out of the sequence masks (wave-uniform test)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s0[e] = (kt + 16 * kh + e < a.Skv) ? s0[e] : -INFINITY;
        s1[e] = (kt + 32 + 16 * kh + e < a.Skv) ? s1[e] : -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fmaxf(s0[e], s1[e]));
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx * scale2);
    float ps = 0.f;
    bf16x8 pb[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], scale2, -m_new));
      const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], scale2, -m_new));
      ps += p0 + p1;
      pb[e >> 3][e & 7] = to_carrier<T>(p0);
      pb[2 + (e >> 3)][e & 7] = to_carrier<T>(p1);
    }
    ps = xhalf_sum(ps);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {   // the 64 accumulator rescales only when some query's maximum moved
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    }
    l_run += ps;
    m_run = m_new;
#pragma unroll
    for (int u = 0; u < 4; ++u)                   // k-step u: keys kt + 32*(u>>1) + 16*kh + 8*(u&1) + 0..7
#pragma unroll
      for (int d = 0; d < ND; ++d)
        o[d] = mfma16<T>(*(const bf16x8*)(Vs + (d * 32 + j) * VP + (32 * (u >> 1) + 16 * kh + 8 * (u & 1)) * 2), pb[u], o[d]);
    if (t + 1 < nt) {
      lstore((t + 1) & 1);        // stage (t+1)&1 was last read in iteration t-1: every wave passed the barrier below since
      __syncthreads();
    }
  }

  const int tok = q0 + j;
  if (tok >= a.Sq) return;
  const float inv = 1.0f / l_run;
  OutT* dst;
  if (tok < a.split) dst = (OutT*)a.outA + ((long)b * a.split + tok) * (a.H * HD);
  else dst = (OutT*)a.outB + ((long)b * (a.Sq - a.split) + (tok - a.split)) * (a.H * HD);
  dst += h * HD;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = {o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv, o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv};
      Pack4Out<OutT>::store(dst + d * 32 + 8 * g4 + 4 * kh, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5 - bf16 / fp16, LONG sequences on small grids, KEY-SPLIT WAVE PAIRS.  The kernel above runs 1.3 waves per SIMD (336
// workgroups of 128 queries on 256 CUs, 1.3 rounds of them on the busiest CUs) and its time is one wave's chain: per 64 keys a
// wave needs ~1000 matrix-pipe cycles AND ~1500 cycles of softmax VALU work (exponentials, maxima, packing) that nothing
// overlaps.  Here a workgroup covers NQ = 4 / 5 / 6 blocks of 32 queries - chosen per problem so that ALL workgroups fit one
// round of 256 CUs (S = 1740: 10 x 24 tiles of 192 queries, S = 1500: 10 x 24 tiles of 160) - and every query block is served
// by TWO waves that split each staged 64-key tile (keys 0..31 / 32..63): 8 - 12 waves per CU = 2 - 3 per SIMD, so one wave's
// softmax VALU runs under another wave's MFMAs, K / V^T are read from L2 once per 128 - 192 queries, and the halves' softmax
// states (m, l, O) are merged once at the end through the dead staging buffers.  Per wave the loop body is the wide kernel's
// 32-key step.  Same operand layout, staging and output mapping as attn_bf16_long_kernel.
template <typename T, typename OutT, int NQ>
__global__ __launch_bounds__(NQ * 128) void attn_bf16_pair_kernel(const AttnArgs a) {
  constexpr int HD = 128, KT = 64, NTH = NQ * 128;
  constexpr int KP = 2 * HD + 16, VP = 2 * KT + 16;      // LDS row pitches in bytes (K rows: 272, V^T rows: 144)
  constexpr int STG = KT * KP + HD * VP;                 // one stage: K tile + V^T tile = 35 840 B
  constexpr int NS = HD / 16, ND = HD / 32;
  constexpr int NPC = (1024 + NTH - 1) / NTH;            // staging pieces per thread and operand (K tile = V^T tile = 1024 pieces of 16 B)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = w % NQ, kh2 = w / NQ;                   // query block, key half of every staged tile
  const int j = lane & 31, kh = lane >> 5;
  int qt, h, b;
  attn_block_coords(a, 32 * NQ, qt, h, b);
  const int q0 = qt * (32 * NQ) + qb * 32;
  const int bk = b / a.kv_bdiv;
  const T* __restrict__ Q = (const T*)a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const T* __restrict__ K = (const T*)a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const T* __restrict__ VT = (const T*)a.v + ((long)(bk * a.H + h) * HD) * a.vt_pitch;
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128)

  bf16x8 qf[NS];
  {
    const T* p = Q + (long)min(q0 + j, a.Sq - 1) * HD + 8 * kh;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8*)(p + 16 * s);
  }
  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int vmax = a.vt_pitch - 8;                       // last 16-byte chunk of a V^T row (the pitch covers Skv rounded up to 32, not to 64)
  u32x4 rk[NPC], rv[NPC];
  auto gload = [&](int t) {
    const int kt = t * KT;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + i * NTH;
      if (p < 1024) {                                    // wave-uniform (NTH and 1024 are multiples of 64)
        rk[i] = *(const u32x4*)(K + (long)min(kt + (p >> 4), a.Skv - 1) * HD + (p & 15) * 8);
        rv[i] = *(const u32x4*)(VT + (long)(p >> 3) * a.vt_pitch + min(kt + (p & 7) * 8, vmax));   // clamped chunks carry masked (p = 0) keys
      }
    }
  };
  auto lstore = [&](int stage) {
    unsigned char* base = lds + stage * STG;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + i * NTH;
      if (p < 1024) {
        *(u32x4*)(base + (p >> 4) * KP + (p & 15) * 16) = rk[i];
        *(u32x4*)(base + KT * KP + (p >> 3) * VP + (p & 7) * 16) = rv[i];
      }
    }
  };
  const int nt = (a.Skv + KT - 1) / KT;
  const int pi = 16 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);  // A-row j of a 32-key score tile carries key pi
  f32x16 s;                                       // raw scores of this wave's 32 keys of a tile
  auto qk = [&](int stg) {                        // s[e] = score(key 64 t + 32 kh2 + 16 kh + e, query q0 + j), tile t staged in buffer `stg`
    const unsigned char* Ks = lds + stg * STG + (32 * kh2) * KP;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st) s = mfma16<T>(*(const bf16x8*)(Ks + pi * KP + (16 * st + 8 * kh) * 2), qf[st], s);
  };
  auto softmax_pv = [&](int t, int stg) {         // online softmax of s (tile t) in the log2 domain (scale2 > 0 commutes with the maximum), then O += V^T P
    const int kt = t * KT + 32 * kh2;
    const unsigned char* Vs = lds + stg * STG + KT * KP + (32 * kh2) * 2;
    if (kt + 32 > a.Skv) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = (kt + 16 * kh + e < a.Skv) ? s[e] : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx * scale2);
    float ps = 0.f;
    bf16x8 pb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], scale2, -m_new));
      ps += pv;
      pb[e >> 3][e & 7] = to_carrier<T>(pv);
    }
    ps = xhalf_sum(ps);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {   // the 64 accumulator rescales only when some query's maximum moved
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    }
    l_run += ps;
    m_run = m_new;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int d = 0; d < ND; ++d)
        o[d] = mfma16<T>(*(const bf16x8*)(Vs + (d * 32 + j) * VP + (16 * kh + 8 * u) * 2), pb[u], o[d]);
  };
  // All waves walk the tiles in lock step (one barrier per tile).  Running the two key halves HALF AN ITERATION APART (second half:
  // softmax(t-1) -> PV(t-1) -> QK(t), three staging buffers) so that one half multiplies while the other exponentiates was built
  // and measured: 72.6 / 72.2 us against 62.4 / 52.7 - the rotated loop keeps two score tiles live and spills at the 168-register
  // cap; an s_sleep skew of the second half: 65.7 / 55.3.
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) gload(t + 1);               // next tile's global loads fly during this tile's math
    if (t * KT + 32 * kh2 < a.Skv) {            // wave-uniform: the second half of the last tile may hold no key at all
      qk(t & 1);
      softmax_pv(t, t & 1);
    }
    if (t + 1 < nt) {
      lstore((t + 1) & 1);                      // stage (t+1)&1 was last read in iteration t-1: every wave passed the barrier below since
      __syncthreads();
    }
  }
  // ---- merge the two key halves of every query block through the dead staging buffers: the second-half wave publishes its
  // (m, l, O) lane-major (16.5 KiB per query block: the launcher sizes the LDS for max(two staging buffers, NQ records)), the first-half wave
  // folds them in and stores.  A half that saw no key publishes m = -inf, l = 0, O = 0: weight exp2(-inf) = 0.
  __syncthreads();
  float* xm = (float*)lds + qb * (66 * 64);
  if (kh2 == 1) {
    xm[lane] = m_run;
    xm[64 + lane] = l_run;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 t4 = {o[d][4 * v], o[d][4 * v + 1], o[d][4 * v + 2], o[d][4 * v + 3]};
        *(f32x4*)(xm + 128 + ((d * 4 + v) * 64 + lane) * 4) = t4;
      }
  }
  __syncthreads();
  if (kh2 == 1) return;
  {
    const float m_b = xm[lane], l_b = xm[64 + lane];
    const float m = fmaxf(m_run, m_b);
    const float fa = __builtin_amdgcn_exp2f(m_run - m), fb = __builtin_amdgcn_exp2f(m_b - m);
    l_run = l_run * fa + l_b * fb;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 t4 = *(const f32x4*)(xm + 128 + ((d * 4 + v) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[d][4 * v + e] = o[d][4 * v + e] * fa + t4[e] * fb;
      }
  }
  const int tok = q0 + j;
  if (tok >= a.Sq) return;
  const float inv = 1.0f / l_run;
  OutT* dst;
  if (tok < a.split) dst = (OutT*)a.outA + ((long)b * a.split + tok) * (a.H * HD);
  else dst = (OutT*)a.outB + ((long)b * (a.Sq - a.split) + (tok - a.split)) * (a.H * HD);
  dst += h * HD;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = {o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv, o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv};
      Pack4Out<OutT>::store(dst + d * 32 + 8 * g4 + 4 * kh, v);
    }
}

}  // namespace

static long long* g_attn_dbg = nullptr;
// Debug hook for tools/attn_timeline.py (not part of include/foley_hip.h)
extern "C" void foley_debug_attn_timeline(void* p) { g_attn_dbg = (long long*)p; }

int launch_attention(const AttnArgs& a_in, int out_dtype, hipStream_t st) {
  static const int no_preload = []() { const char* e = getenv("FOLEY_ATTN_PRELOAD"); return (e && e[0] == '0') ? 1 : 0; }();
  AttnArgs a = a_in;
  a.no_preload = no_preload;
  a.dbg = g_attn_dbg;
  if (a.Sq <= 0 || a.Skv <= 0) return foley_set_err("attention: empty sequence", __FILE__, __LINE__);
  const int hd = a.head_dim > 0 ? a.head_dim : 128;
  if (hd != 128 && hd != 64) return foley_set_err("attention: head_dim must be 128 or 64", __FILE__, __LINE__);
  if (a.grp_q > 0 || a.grp_kv > 0) {
    if (hd != 64 || !foley_is_half(a.in_dtype) || a.grp_q < 1 || a.grp_kv < 1 || a.kv_bdiv != 1 ||
        (long)((a.Sq + a.grp_q - 1) / a.grp_q) * a.grp_kv > a.Skv)
      return foley_set_err("attention: grouped (block-diagonal) form: head_dim 64, 16-bit operands, every query group's keys inside Skv", __FILE__, __LINE__);
  }
  if (a.out_rows && hd != 64) return foley_set_err("attention: the output row table serves head_dim 64 (the conditioning encoders' kernels)", __FILE__, __LINE__);
  dim3 grid((a.Sq + 31) / 32, a.H, a.Bq), block(64);
  const dim3 grid1(grid.x * grid.y * grid.z);   // 16-bit kernels: 1-D grid, XCD-aware remap inside
  if (foley_is_half(a.in_dtype)) {
    if (a.vt_pitch < ((a.Skv + 31) & ~31) || (a.vt_pitch & 7))
      return foley_set_err("attention: V^T pitch must cover Skv rounded up to 32 (multiple of 8)", __FILE__, __LINE__);
    if (out_dtype != a.in_dtype && out_dtype != FOLEY_F32)
      return foley_set_err("attention: 16-bit operands produce the same type or fp32", __FILE__, __LINE__);
    // enough 128-query workgroups to cover the chip => the wide kernel (operands read once per 128 queries); head dim 64
    // (the conditioning encoders) exists in the wide form only
    const dim3 gw(((a.Sq + 127) / 128) * a.H * a.Bq);
    const bool wide = (long)gw.x >= 256 || hd == 64, h16 = a.in_dtype == FOLEY_F16, o32 = out_dtype == FOLEY_F32;
    // long key sequences on grids of at most ~2 waves per SIMD: 64 keys per iteration of each wave's dependent chain (attn_bf16_long_kernel;
    // FOLEY_ATTN_LONG=0 keeps the 32-key form)
    static const bool long_on = []() { const char* e = getenv("FOLEY_ATTN_LONG"); return !(e && e[0] == '0'); }();
    const bool longk = long_on && wide && hd == 128 && a.Skv >= 512 && (long)gw.x * 4 <= 2048;
    // ... and where 128 / 160 / 192-query workgroups of key-split wave pairs cover the problem in ONE round of 256 CUs, that form
    // (attn_bf16_pair_kernel; FOLEY_ATTN_PAIR=0 keeps the 64-key chain): the largest workgroup count <= 256 wins
    static const bool pair_on = []() { const char* e = getenv("FOLEY_ATTN_PAIR"); return !(e && e[0] == '0'); }();
    int pair_nq = 0;
    if (pair_on && longk) {
      long best = 0;
      for (int nq = 6; nq >= 4; --nq) {
        const long wgs = (long)((a.Sq + 32 * nq - 1) / (32 * nq)) * a.H * a.Bq;
        if (wgs <= 256 && wgs > best) { best = wgs; pair_nq = nq; }
      }
      if (best < 160) pair_nq = 0;              // too few workgroups: the chain kernel's two per CU do better
    }
    // small grids whose operands fit the LDS: the DMA-staged form (FOLEY_ATTN_LDS=0 keeps the register-loaded kernel)
    static const bool lds_on = []() { const char* e = getenv("FOLEY_ATTN_LDS"); return !(e && e[0] == '0'); }();
    const int nt = (a.Skv + 31) >> 5;
    const long img = (long)nt * 32 * 256 + (128L << ((a.vt_pitch > 128 ? 5 : 4) + 4)) + 32 * 256, mrg = 4 * 3 * 16 * 64 * 4 + 2 * 4 * 32 * 4;
    // (three key tiles or fewer - the 77 text keys - are a wash: 5.95 vs 6.2 us in the loop; single-block self-attention 8.4 -> 7.3 us)
    const bool staged = lds_on && !wide && hd == 128 && nt >= 4 && a.vt_pitch <= 256 && img <= 160 * 1024;
    const int merge_off = staged && img + mrg <= 160 * 1024 ? (int)img : 0;
    const size_t lds16 = staged ? (size_t)(merge_off ? img + mrg : (img > mrg ? img : mrg)) : 0;
#define FOLEY_ATTN16(T, O)                                                                         \
    do {                                                                                             \
      if (staged) {                                                                                  \
        static std::atomic<unsigned long long> raised{0};                                            \
        hipError_t e_ = foley_raise_lds((const void*)attn_lds_kernel<T, O>, 160 * 1024, raised);     \
        if (e_ != hipSuccess) return foley_set_err(hipGetErrorString(e_), __FILE__, __LINE__);       \
        FOLEY_LAUNCH((attn_lds_kernel<T, O>), grid1, dim3(256), lds16, st, a, merge_off);            \
      } else                                                                                         \
      if (pair_nq) {                                                                                 \
        static std::atomic<unsigned long long> r4{0}, r5{0}, r6{0};                                  \
        const dim3 gp(((a.Sq + 32 * pair_nq - 1) / (32 * pair_nq)) * a.H * a.Bq);                    \
        hipError_t e_ = hipSuccess;                                                                  \
        if (pair_nq == 6) { e_ = foley_raise_lds((const void*)attn_bf16_pair_kernel<T, O, 6>, 6 * 16896, r6); if (e_ == hipSuccess) FOLEY_LAUNCH((attn_bf16_pair_kernel<T, O, 6>), gp, dim3(768), 6 * 16896, st, a); } \
        else if (pair_nq == 5) { e_ = foley_raise_lds((const void*)attn_bf16_pair_kernel<T, O, 5>, 5 * 16896, r5); if (e_ == hipSuccess) FOLEY_LAUNCH((attn_bf16_pair_kernel<T, O, 5>), gp, dim3(640), 5 * 16896, st, a); } \
        else { e_ = foley_raise_lds((const void*)attn_bf16_pair_kernel<T, O, 4>, 2 * 35840, r4); if (e_ == hipSuccess) FOLEY_LAUNCH((attn_bf16_pair_kernel<T, O, 4>), gp, dim3(512), 2 * 35840, st, a); } \
        if (e_ != hipSuccess) return foley_set_err(hipGetErrorString(e_), __FILE__, __LINE__);       \
      } else                                                                                         \
      if (longk) {                                                                                   \
        static std::atomic<unsigned long long> raised{0};                                            \
        hipError_t e_ = foley_raise_lds((const void*)attn_bf16_long_kernel<T, O>, 2 * 35840, raised); \
        if (e_ != hipSuccess) return foley_set_err(hipGetErrorString(e_), __FILE__, __LINE__);       \
        FOLEY_LAUNCH((attn_bf16_long_kernel<T, O>), gw, dim3(256), 2 * 35840, st, a);                \
      } else                                                                                         \
      if (wide && hd == 64 && a.grp_q > 0) FOLEY_LAUNCH((attn_bf16_wide_kernel<T, O, 64, true>), gw, dim3(256), 0, st, a);  \
      else if (wide && hd == 64) FOLEY_LAUNCH((attn_bf16_wide_kernel<T, O, 64>), gw, dim3(256), 0, st, a);  \
      else if (wide) FOLEY_LAUNCH((attn_bf16_wide_kernel<T, O, 128>), gw, dim3(256), 0, st, a);        \
      else FOLEY_LAUNCH((attn_bf16_kernel<T, O>), grid1, dim3(256), 0, st, a);                         \
    } while (0)
    if (h16) { if (o32) FOLEY_ATTN16(f16_t, float); else FOLEY_ATTN16(f16_t, f16_t); }
    else { if (o32) FOLEY_ATTN16(bf16_t, float); else FOLEY_ATTN16(bf16_t, bf16_t); }
#undef FOLEY_ATTN16
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return foley_set_err(hipGetErrorString(e2), __FILE__, __LINE__);
    return 0;
  }
#define FOLEY_ATTN32(O)                                                          \
  do {                                                                           \
    if (hd == 64) FOLEY_LAUNCH((attn_kernel<O, 64>), grid, block, 0, st, a);     \
    else FOLEY_LAUNCH((attn_kernel<O, 128>), grid, block, 0, st, a);             \
  } while (0)
  if (out_dtype == FOLEY_F32) FOLEY_ATTN32(float);
  else if (out_dtype == FOLEY_BF16) FOLEY_ATTN32(bf16_t);
  else if (out_dtype == FOLEY_F16) FOLEY_ATTN32(f16_t);
  else return foley_set_err("attention: bad output dtype", __FILE__, __LINE__);
#undef FOLEY_ATTN32
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}
