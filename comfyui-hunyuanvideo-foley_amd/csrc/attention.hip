// Flash-style attention for gfx950, head_dim = 128, no mask (reference: attn_layers.py:419-422,
// hifi_foley.py:383).  fp32 operands on v_mfma_f32_32x32x2_f32, online softmax in fp32.
//
// One wavefront owns 32 query rows and walks the keys in tiles of 32.  The score tile is computed
// *transposed* (S^T = K Q^T) so that every lane holds 16 keys of ONE query column: the row max /
// row sum are lane-local plus a single lane^32 exchange, and the exponentiated scores already sit
// in the B-operand layout of the second MFMA (O^T += V^T P^T) - no LDS, no cross-lane shuffles.
//   MFMA 32x32x2 operand layout: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
//   D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31].
#include "kernels.h"

namespace {

constexpr int HD = 128;

template <typename OutT>
__global__ __launch_bounds__(64) void attn_kernel(const AttnArgs a) {
  const int lane = threadIdx.x;
  const int j = lane & 31, kh = lane >> 5;
  const int q0 = blockIdx.x * 32;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bk = b / a.kv_bdiv;
  const float* __restrict__ Q = a.q + ((long)(b * a.H + h) * a.Sq) * HD;
  const float* __restrict__ K = a.k + ((long)(bk * a.H + h) * a.Skv) * HD;
  const float* __restrict__ V = a.v + ((long)(bk * a.H + h) * a.Skv) * HD;
  const float scale = 0.08838834764831845f;  // 1/sqrt(128)

  // B operand of S^T = K Q^T: lane (j, kh) holds Q[q0 + j][kh*64 .. kh*64+63]
  float qr[64];
  {
    const int qrow = min(q0 + j, a.Sq - 1);
    const f32x4* p = (const f32x4*)(Q + (long)qrow * HD + kh * 64);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f32x4 v = p[c];
      qr[c * 4 + 0] = v[0]; qr[c * 4 + 1] = v[1]; qr[c * 4 + 2] = v[2]; qr[c * 4 + 3] = v[3];
    }
  }

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
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
      const f32x4* p = (const f32x4*)(K + (long)krow * HD + kh * 64);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
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
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);     // finite: every tile holds at least one valid key
    const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = expf(s[e] - m_new);
      ps += s[e];
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
    // O^T[d][q] += sum_key V[key][d] * P[q][key]: A operand lane (i = j, kh) = V[kt + r(t)][d0 + i],
    // B operand = s[t] (already in place).
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int vrow = min(kt + (t & 3) + 8 * (t >> 2) + 4 * kh, a.Skv - 1);
      const float* vp = V + (long)vrow * HD + j;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[d * 32], s[t], o[d], 0, 0, 0);
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
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int c = d * 32 + 8 * g4 + 4 * kh;
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[c + u] = Cvt<OutT>::to(o[d][g4 * 4 + u] * inv);
    }
}

}  // namespace

int launch_attention(const AttnArgs& a, int out_dtype, hipStream_t st) {
  if (a.Sq <= 0 || a.Skv <= 0) return foley_set_err("attention: empty sequence", __FILE__, __LINE__);
  dim3 grid((a.Sq + 31) / 32, a.H, a.Bq), block(64);
  if (out_dtype == FOLEY_F32) hipLaunchKernelGGL(attn_kernel<float>, grid, block, 0, st, a);
  else if (out_dtype == FOLEY_BF16) hipLaunchKernelGGL(attn_kernel<bf16_t>, grid, block, 0, st, a);
  else return foley_set_err("attention: bad output dtype", __FILE__, __LINE__);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}
