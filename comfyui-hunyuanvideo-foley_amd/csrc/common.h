// Shared device/host helpers for the Foley HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16-byte staging register (native vector: stays in VGPRs)
typedef uint16_t bf16_t;  // storage type for bf16 in global/LDS memory
typedef _Float16 f16_t;   // storage type for IEEE fp16 (precision=fp16 / fp16 checkpoints: the reference runs those under
                          // torch.autocast(float16), nodes.py:89-106, utils.py:229-234).  Every 16-bit code path is shared
                          // with bf16 - same tiles, LDS images and DMA; only the MFMA opcode and the conversions differ.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// ---- dtype codes shared with include/foley_hip.h -------------------------------------------
enum { FOLEY_F32 = 0, FOLEY_BF16 = 1, FOLEY_I32 = 2, FOLEY_F8E4M3 = 3, FOLEY_F8E5M2 = 4, FOLEY_F16 = 5 };
__host__ __device__ constexpr bool foley_is_half(int dt) { return dt == FOLEY_BF16 || dt == FOLEY_F16; }   // 16-bit operand types

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (matches torch .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (gfx950; round-to-nearest-even as f32_to_bf16 / torch, which
// cost six VALU instructions per element in software - every bf16 store of every epilogue and row kernel)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t p = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, p);
}

// two floats -> packed fp16 pair (round-to-nearest-even, like torch .to(float16))
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
  const f16x2_t p = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, p);
}
// packed pair in the 16-bit type T (bf16_t or f16_t)
template <typename T> __device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) return pack_f16x2(lo, hi);
  else return pack_bf16x2(lo, hi);
}
// 16-bit MFMA of the operand type: fragments travel as 128-bit registers (bf16x8 is just the carrier type)
template <typename T> __device__ __forceinline__ f32x16 mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (__is_same(T, f16_t)) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// one fp32 value as a lane of the 16-bit carrier vector (attention: probabilities -> B operand of the P V product)
template <typename T> __device__ __forceinline__ __bf16 to_carrier(float v) {
  if constexpr (__is_same(T, f16_t)) return __builtin_bit_cast(__bf16, (_Float16)v);
  else return (__bf16)v;
}
template <typename T> struct DtCode;
template <> struct DtCode<float> { static constexpr int v = FOLEY_F32; };
template <> struct DtCode<bf16_t> { static constexpr int v = FOLEY_BF16; };
template <> struct DtCode<f16_t> { static constexpr int v = FOLEY_F16; };

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to(float v) { return v; }
  static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ bf16_t to(float v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ float from(bf16_t v) { return bf16_to_f32(v); }
};
template <> struct Cvt<f16_t> {
  static __device__ __forceinline__ f16_t to(float v) { return (f16_t)v; }
  static __device__ __forceinline__ float from(f16_t v) { return (float)v; }
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// torch GELU(approximate="tanh")
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float inner = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}
// bf16-output forms (their error, ~1e-6 relative from v_exp_f32 / v_rcp_f32, is four orders below the bf16 rounding
// that follows): 5-7 VALU instructions instead of ~25 (expf + IEEE division / tanhf) per element in the epilogues
__device__ __forceinline__ float silu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * u));   // tanh(u)
  return 0.5f * x * (1.0f + t);
}
template <typename OutT> __device__ __forceinline__ float silu_o(float x) { return sizeof(OutT) == 2 ? silu_fast(x) : silu_f(x); }
// exact GELU: 0.5 x (1 + erf(x / sqrt(2)))  (torch nn.GELU() default)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
template <typename OutT> __device__ __forceinline__ float gelu_o(float x) { return sizeof(OutT) == 2 ? gelu_tanh_fast(x) : gelu_tanh_f(x); }
// exact GELU for 16-bit outputs: Phi(x) through erfc's rational form (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute - three
// orders below the fp16 / bf16 rounding of the result): q = poly(t) * exp(-z^2), t = 1 / (1 + p |z|), z = x / sqrt(2);
// 1 + erf(z) = q for z < 0 (no cancellation in the negative tail) and 2 - q otherwise.  The bound is ABSOLUTE: |error of GELU(x)| <=
// 0.5 |x| 1.5e-7 (<= 6e-7 on [-8, 8]) - exact to 16-bit rounding wherever |GELU(x)| > ~2e-3 (bf16) / ~2e-2 (fp16), but in the negative tail (x <= -5, where
// erfc ~ 5e-7 and the result ~ 1e-6) the RELATIVE error reaches tens of percent: values that vanish next to any other activation
// (tests/test_ops_gpu.py::test_exact_gelu_fast_form_error_bound pins both statements).  ~14 VALU instructions against erff()'s
// ~45 with its range branches: the Synchformer fc1 epilogue (21 966 x 3 072 outputs per layer) 215 -> 17x us per launch.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = x * 0.7071067811865476f, az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float q = poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
  return 0.5f * x * (z < 0.0f ? q : 2.0f - q);
}
template <typename OutT> __device__ __forceinline__ float gelu_erf_o(float x) { return sizeof(OutT) == 2 ? gelu_erf_fast(x) : gelu_erf_f(x); }

// DAC snake: x + (alpha + 1e-9)^-1 * sin(alpha x)^2
__device__ __forceinline__ float snake_f(float x, float alpha, float inv_alpha) {
  float s = sinf(alpha * x);
  return x + inv_alpha * (s * s);
}

// Sum over each aligned group of 16 lanes (one DPP row), result in every lane of the group: four
// DPP moves (quad swaps, then the half-row and row mirrors) - no LDS crossbar traffic, unlike
// __shfl_xor / ds_bpermute.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// Exchange between the lane pair (l, l ^ 32) - the two key halves of a 32x32 score tile's column - as ONE v_permlane32_swap
// (VALU) instead of the ds_bpermute round trip through the LDS crossbar that __shfl_xor(v, 32) compiles to.  Bit-identical
// to `op(v, __shfl_xor(v, 32))`: max and a two-operand add commute.
__device__ __forceinline__ float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over the 64 lanes, result in every lane: the four row sums meet through two DPP row broadcasts (lane 15 of a
// row into the next row, lane 31 into the upper half) and one v_readlane - no ds_bpermute round trips through the LDS.
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Row-broadcast operand addressing shared by the row kernels and GEMM epilogues.
//   mode 0: one vector for every row            (index 0)
//   mode 1: rows ordered [cfg][clip][l]; operand indexed [cfg][l]
//   mode 2: rows ordered [cfg][clip][l]; the operand has Ls rows per cfg and token l reads row
//           nearest_exact(l) = min(floor((l + 0.5) * Ls / L), Ls - 1): F.interpolate(mode="nearest-exact")
//           folded into the addressing (hifi_foley.py:759-762 up-samples the sync tokens to the audio
//           frame rate; everything computed per audio frame from them alone only has Ls distinct rows).
//           float32 arithmetic exactly as torch does it (scale = float(Ls) / float(L), one multiply).
struct RowBcast {
  const float* p;       // base (may be null => operand absent)
  long ld;              // elements between operand rows
  int mode;             // see above
  int rows_per_cfg;     // clips * L
  int L;                // tokens per clip
  const int* step_ptr;  // optional device-resident iteration counter
  long step_stride;     // elements added per iteration
  int Ls;               // mode 2: rows per cfg of the up-sampled sequence
  float scale;          // mode 2: float(Ls) / float(L)
  int per;              // mode 2: 0, or a power of two: the Ls rows repeat with this period and only the first `per`
                        // rows per cfg are stored (empty sync features: sync_pos_emb makes 8 distinct rows per half)
  int dense_from;       // mode 2: cfg halves [0, dense_from) are stored periodically (`per` rows each), the others with all their
  int dense_base;       // Ls rows from operand row dense_base on (0 / 0: every half dense; INT_MAX: every half periodic).  A video
                        // clip under CFG: the unconditional half carries the empty sync features (8 rows), the other one Ls rows
};

__host__ __device__ __forceinline__ int rb_nearest_exact(int l, float scale, int Ls) {
  const int s = (int)floorf(((float)l + 0.5f) * scale);
  return s < Ls ? s : Ls - 1;
}

__device__ __forceinline__ const float* rb_row(const RowBcast& b, int r) {
  const float* p = b.p;
  if (b.step_ptr) p += (long)(*b.step_ptr) * b.step_stride;
  if (b.mode == 1) p += ((long)(r / b.rows_per_cfg) * b.L + (r % b.L)) * b.ld;
  else if (b.mode == 2) {
    const int s = rb_nearest_exact(r % b.L, b.scale, b.Ls), cfg = r / b.rows_per_cfg;
    const int idx = cfg < b.dense_from ? cfg * b.per + (s & (b.per - 1)) : b.dense_base + (cfg - b.dense_from) * b.Ls + s;   // 32-bit: row counts are small
    p += (long)idx * b.ld;
  }
  return p;
}

// Launch hook of the per-kernel profile (foley_profile_forward): when armed, the next launch carries
// the two events as the dispatch's own start / stop timestamps (hipExtLaunchKernelGGL) - the same
// begin/end stamps rocprofv3's kernel trace reports, without marker-packet overhead.
struct FoleyProfHook {
  hipEvent_t e0, e1;
  const void* fn;   // out: host address of the kernel the armed launch ran (its symbol names the rocprofv3 trace row of the op)
};
extern thread_local FoleyProfHook g_foley_prof;
#define FOLEY_LAUNCH(kernel, grid, block, lds, st, ...)                                                          \
  do {                                                                                                           \
    if (g_foley_prof.e0) {                                                                                       \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, g_foley_prof.e0, g_foley_prof.e1, 0, __VA_ARGS__);     \
      g_foley_prof.e0 = nullptr;                                                                                 \
      g_foley_prof.fn = (const void*)(kernel);                                                                   \
    } else {                                                                                                     \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                             \
    }                                                                                                            \
  } while (0)

#define FOLEY_CHECK_HIP(expr)                                         \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) return foley_set_err(hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

int foley_set_err(const char* msg, const char* file, int line);

// Opt a kernel into more than 64 KiB of dynamic LDS, once per (call site = kernel instantiation, device): the attribute belongs to
// the current device's copy of the function, and one process may drive several GPUs (host/sampler.py::denoise_process_multi).
inline hipError_t foley_raise_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(1ull << dev, std::memory_order_release);
  return e;
}
