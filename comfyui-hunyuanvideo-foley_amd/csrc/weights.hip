// Reference-keyed weight loading: checkpoint tensors -> the packed arena, on the device.
//
// The reference loads checkpoints into nn.Modules (HunyuanModelLoader.load_model nodes.py:72-133,
// load_dac_any utils.py:61-87) and re-derives layouts on every forward.  A maintainer who keeps the
// reference's loader hands its tensors to this file under their STATE-DICT KEYS
// (`triple_blocks.0.audio_mod.linear.weight`, `decoder.model.1.block.1.parametrizations.weight.original0` ...)
// and the library takes every layout decision of DESIGN.md "data layout" itself: (K H D) q/k/v rows,
// tap-major conv weights, SwiGLU pairs interleaved in 32-row groups, one fused modulation matrix for
// the single-stream blocks, weight-norm folded, transposed convs as phases x 2 taps.  The arena is
// one ctx-owned allocation whose layout depends on the configuration only, so a multi-GPU job ships
// it with ONE broadcast (foley_bcast_weights: RCCL, resolved at run time from the already loaded
// librccl so that libfoley_hip.so itself does not link it).
//
// Same packed names / shapes / values as host/packers.py (tests/test_model_gpu.py compares the two
// loaders bit for bit on the DiT; the DAC weight-norm fold differs by fp32 reduction order only).
#include "../../include/foley_hip.h"
#include "kernels.h"

#include <dlfcn.h>

#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

// internal hooks of foley_rt.hip (not part of the C ABI)
void** foley_ctx_wstore_slot(foley_ctx* c);
void foley_ctx_set_wstore_free(foley_ctx* c, void (*fn)(void*));
const foley_config* foley_ctx_config(foley_ctx* c);
int foley_ctx_device(foley_ctx* c);

namespace {

#define W_FAIL(code, msg) (foley_set_err(msg, __FILE__, __LINE__), (code))
#define W_TRY(expr)           \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)
#define W_HIP(expr)                                              \
  do {                                                           \
    hipError_t _e = (expr);                                      \
    if (_e != hipSuccess) {                                      \
      foley_set_err(hipGetErrorString(_e), __FILE__, __LINE__);  \
      return FOLEY_ERR_HIP;                                      \
    }                                                            \
  } while (0)

enum { DT_F16 = FOLEY_F16 };   // fp16: a checkpoint dtype and (precision=fp16) a compute / arena dtype

__host__ __device__ inline int dt_size(int dt) {
  return (dt == FOLEY_F8E4M3 || dt == FOLEY_F8E5M2) ? 1 : (dt == FOLEY_BF16 || dt == DT_F16) ? 2 : 4;
}

// ---- scalar conversions (device); fp8 follows the OCP formats torch implements, round-to-nearest-even
__device__ inline float f16_to_f32(uint16_t h) {
  const uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023;
  if (e == 0) {
    if (m == 0) return __uint_as_float(s);
    float v = (float)m * 5.9604644775390625e-08f;   // 2^-24
    return (h & 0x8000) ? -v : v;
  }
  if (e == 31) return __uint_as_float(s | 0x7f800000u | (m << 13));
  return __uint_as_float(s | ((e + 112) << 23) | (m << 13));
}
template <int EB, int MB, bool FN>   // exponent / mantissa bits; FN: no infinities, NaN = all ones (e4m3fn)
__device__ inline float f8_to_f32(uint8_t v) {
  constexpr int BIAS = (1 << (EB - 1)) - 1;
  const uint32_t s = (uint32_t)(v & 0x80) << 24;
  const int e = (v >> MB) & ((1 << EB) - 1), m = v & ((1 << MB) - 1);
  if (FN ? ((v & 0x7f) == 0x7f) : (e == (1 << EB) - 1 && m != 0)) return __uint_as_float(0x7fc00000u);
  if (!FN && e == (1 << EB) - 1) return __uint_as_float(s | 0x7f800000u);
  if (e == 0) {
    float x = (float)m * exp2f((float)(1 - BIAS - MB));
    return s ? -x : x;
  }
  return __uint_as_float(s | ((uint32_t)(e - BIAS + 127) << 23) | ((uint32_t)m << (23 - MB)));
}
template <int EB, int MB, bool FN>
__device__ inline uint8_t f32_to_f8(float f) {
  constexpr int BIAS = (1 << (EB - 1)) - 1;
  const uint32_t u = __float_as_uint(f);
  const uint8_t s = (uint8_t)((u >> 24) & 0x80);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return s | 0x7f;                                    // NaN
  const float maxv = FN ? 448.0f : 57344.0f;
  const float af = __uint_as_float(a);
  if (FN) {
    if (af > 464.0f) return s | 0x7f;                                      // beyond the rounding range of 448: NaN (torch e4m3fn)
  } else if (af >= 61440.0f) {
    return s | 0x7c;                                                       // rounds to infinity (e5m2)
  }
  (void)maxv;
  const int e = (int)(a >> 23) - 127;                                      // unbiased exponent
  if (e < 1 - BIAS) {                                                      // subnormal of the target: fixed grid 2^(1-BIAS-MB)
    const float q = af * exp2f((float)(BIAS - 1 + MB));
    const float r = rintf(q);                                              // round-to-nearest-even
    return s | (uint8_t)r;                                                 // r == 2^MB lands on the first normal: same bits
  }
  uint32_t m = a & 0x7fffffu;
  const uint32_t drop = 23 - MB, half = 1u << (drop - 1);
  uint32_t keep = m >> drop;
  const uint32_t rem = m & ((1u << drop) - 1);
  int eb = e + BIAS;
  if (rem > half || (rem == half && (keep & 1))) {
    if (++keep == (1u << MB)) {
      keep = 0;
      ++eb;
    }
  }
  const uint8_t out = (uint8_t)((eb << MB) | keep);
  if (FN && (out & 0x7f) == 0x7f) return s | 0x7f;
  return s | out;
}

__device__ inline float load_as_f32(const void* p, long i, int dt) {
  switch (dt) {
    case FOLEY_F32: return ((const float*)p)[i];
    case FOLEY_BF16: return bf16_to_f32(((const bf16_t*)p)[i]);
    case DT_F16: return f16_to_f32(((const uint16_t*)p)[i]);
    case FOLEY_F8E4M3: return f8_to_f32<4, 3, true>(((const uint8_t*)p)[i]);
    default: return f8_to_f32<5, 2, false>(((const uint8_t*)p)[i]);
  }
}
__device__ inline void store_from_f32(void* p, long i, int dt, float v) {
  switch (dt) {
    case FOLEY_F32: ((float*)p)[i] = v; break;
    case FOLEY_BF16: ((bf16_t*)p)[i] = f32_to_bf16(v); break;
    case DT_F16: ((f16_t*)p)[i] = (f16_t)v; break;
    case FOLEY_F8E4M3: ((uint8_t*)p)[i] = f32_to_f8<4, 3, true>(v); break;
    default: ((uint8_t*)p)[i] = f32_to_f8<5, 2, false>(v); break;
  }
}

// ---- generic strided pack: out[o(idx)] = round_out(round_mid(in[i(idx)] * scale[idx[scale_dim]]))
struct PackDesc {
  const void* in;
  void* out;
  int in_dt, out_dt;
  int mid_dt;            // < 0: none; else the value is first rounded through this dtype (fp8 time-embedding quirk)
  long size[5], is[5], os[5];
  long in_off, out_off;
  const float* scale;    // optional multiplier (weight-norm g / ||v||)
  int scale_dim;         // index dimension of `scale`; < 0: scale[0]
};
__global__ void pack_kernel(const PackDesc d) {
  const long n = d.size[0] * d.size[1] * d.size[2] * d.size[3] * d.size[4];
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    long r = t, idx[5];
#pragma unroll
    for (int k = 4; k >= 0; --k) {
      idx[k] = r % d.size[k];
      r /= d.size[k];
    }
    long io = d.in_off, oo = d.out_off;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      io += idx[k] * d.is[k];
      oo += idx[k] * d.os[k];
    }
    float v = load_as_f32(d.in, io, d.in_dt);
    if (d.scale) v *= d.scale[d.scale_dim >= 0 ? idx[d.scale_dim] : 0];
    if (d.mid_dt == FOLEY_F8E4M3) v = f8_to_f32<4, 3, true>(f32_to_f8<4, 3, true>(v));
    else if (d.mid_dt == FOLEY_F8E5M2) v = f8_to_f32<5, 2, false>(f32_to_f8<5, 2, false>(v));
    store_from_f32(d.out, oo, d.out_dt, v);
  }
}

// weight-norm: scale[r] = g[r] / ||v[r, :]||_2   (nn/layers.py:9-14; dim 0 of v, whatever it means for the layer)
__global__ __launch_bounds__(256) void wn_scale_kernel(const void* v, int v_dt, const void* g, int g_dt, long cols,
                                                        float* scale) {
  __shared__ float part[4];
  const long r = blockIdx.x;
  float s = 0.f;
  for (long c = threadIdx.x; c < cols; c += 256) {
    const float x = load_as_f32(v, r * cols + c, v_dt);
    s += x * x;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) scale[r] = load_as_f32(g, r, g_dt) / sqrtf((part[0] + part[1]) + (part[2] + part[3]));
}

// ---- the store
struct Slot {
  std::string name;
  int dt;
  std::vector<int64_t> shape;
  size_t off = 0, bytes = 0;
  long parts_needed = 1, parts_done = 0;
  std::set<long> parts_seen;   // destination offsets already packed: a repeated key overwrites its part, it does not count twice
};
struct Pending {   // one half of a weight-norm pair waiting for the other
  void* buf = nullptr;
  int dt = 0;
  std::vector<int64_t> shape;
};
struct WStore {
  int device = 0;
  foley_config cfg{};
  int wfmt = 0;
  void* arena = nullptr;
  size_t bytes = 0;
  std::vector<Slot> slots;
  std::map<std::string, int> index;
  std::map<std::string, Pending> pend_g, pend_v;
  std::vector<void*> temps;
  bool begun = false;
};

void wstore_free(void* p) {
  WStore* w = (WStore*)p;
  if (!w) return;
  hipSetDevice(w->device);
  if (w->arena) hipFree(w->arena);
  for (auto& kv : w->pend_g) hipFree(kv.second.buf);
  for (auto& kv : w->pend_v) hipFree(kv.second.buf);
  for (void* t : w->temps) hipFree(t);
  delete w;
}

int64_t numel(const std::vector<int64_t>& s) {
  int64_t n = 1;
  for (auto v : s) n *= v;
  return n;
}

void add_slot(WStore& w, const std::string& name, int dt, std::vector<int64_t> shape, long parts = 1) {
  Slot s;
  s.name = name;
  s.dt = dt;
  s.shape = std::move(shape);
  s.bytes = (size_t)numel(s.shape) * dt_size(dt);
  s.parts_needed = parts;
  w.index[name] = (int)w.slots.size();
  w.slots.push_back(std::move(s));
}

// Same packed tensors as host/packers.py::pack_dit / pack_dac (names, dtypes, shapes).
void build_layout(WStore& w) {
  const foley_config& f = w.cfg;
  const int64_t D = f.hidden, T = f.compute_dtype;
  const int WT = w.wfmt == 1 ? FOLEY_F8E4M3 : w.wfmt == 2 ? FOLEY_F8E5M2 : (int)T;   // block matrices
  auto lin = [&](const std::string& n, int dt, int64_t N, int64_t K, bool bias, long parts = 1) {
    add_slot(w, n + ".w", dt, {N, K}, parts);
    if (bias) add_slot(w, n + ".b", FOLEY_F32, {N}, parts);
  };
  for (int b = 0; b < f.depth_triple; ++b) {
    const std::string p = "t" + std::to_string(b) + ".";
    for (const char* s : {"a_", "v_"}) {
      lin(p + s + "mod", WT, 9 * D, D, true);
      lin(p + s + "qkv", WT, 3 * D, D, true);
      lin(p + s + "proj", WT, D, D, true);
      lin(p + s + "cq", WT, D, D, true);
      lin(p + s + "cproj", WT, D, D, true);
      lin(p + s + "fc1", WT, f.mlp_hidden, D, true);
      lin(p + s + "fc2", WT, D, f.mlp_hidden, true);
      for (const char* g : {"qn", "kn", "cqn"}) add_slot(w, p + s + g, FOLEY_F32, {128});
    }
    lin(p + "t_kv", WT, 2 * D, D, true);
    add_slot(w, p + "t_kn", FOLEY_F32, {128});
  }
  if (f.depth_single > 0) lin("smod_all", WT, (int64_t)f.depth_single * 6 * D, D, true, f.depth_single);
  for (int b = 0; b < f.depth_single; ++b) {
    const std::string p = "s" + std::to_string(b) + ".";
    lin(p + "qkv", WT, 3 * D, D, true);
    add_slot(w, p + "qn", FOLEY_F32, {128});
    add_slot(w, p + "kn", FOLEY_F32, {128});
    lin(p + "lin1", WT, D, 3 * D, true);
    lin(p + "w13", WT, 2 * (int64_t)f.conv_hidden, 3 * D, false, 2);
    lin(p + "w2", WT, D, 3 * (int64_t)f.conv_hidden, false);
  }
  lin("audio_in", (int)T, D, f.latent_dim, true);
  lin("vis.w13", (int)T, 2 * D, f.clip_dim, false, 2);
  lin("vis.w2", (int)T, D, D, false);
  lin("cond1", (int)T, D, f.cond_dim, true);
  lin("cond2", (int)T, D, D, true);
  lin("time0", (int)T, D, f.time_freq_dim, true);
  lin("time2", (int)T, D, D, true);
  lin("sync0", (int)T, D, f.sync_dim, true);
  lin("sync.w13", (int)T, 2 * (int64_t)f.sync_hidden, D, false, 2);
  lin("sync.w2", (int)T, D, f.sync_hidden, false);
  add_slot(w, "sync_pos", FOLEY_F32, {8, f.sync_dim});
  lin("final", (int)T, f.latent_dim, D, true);
  add_slot(w, "empty_clip", FOLEY_F32, {f.clip_dim});
  add_slot(w, "empty_sync", FOLEY_F32, {f.sync_dim});
  // DAC decoder (fp32)
  const int64_t L = f.latent_dim;
  lin("dac.pq", FOLEY_F32, L, L, true);
  lin("dac.in", FOLEY_F32, f.dac_dim, 7 * L, true);
  int64_t cin = f.dac_dim;
  for (int i = 0; i < f.dac_n_rates; ++i) {
    const int64_t s = f.dac_rates[i], cout = cin / 2;
    const std::string p = "dac." + std::to_string(i) + ".";
    add_slot(w, p + "alpha0", FOLEY_F32, {cin});
    add_slot(w, p + "up.w", FOLEY_F32, {s * cout, 2 * cin}, 2);
    add_slot(w, p + "up.b", FOLEY_F32, {s * cout});
    for (int j = 0; j < 3; ++j) {
      const std::string u = p + std::to_string(j) + ".";
      add_slot(w, u + "a1", FOLEY_F32, {cout});
      lin(u + "c7", FOLEY_F32, cout, 7 * cout, true);
      add_slot(w, u + "a2", FOLEY_F32, {cout});
      lin(u + "c1", FOLEY_F32, cout, cout, true);
    }
    cin = cout;
  }
  add_slot(w, "dac.out.alpha", FOLEY_F32, {cin});
  add_slot(w, "dac.out.w", FOLEY_F32, {7 * cin});
  add_slot(w, "dac.out.b", FOLEY_F32, {1});
  size_t off = 0;
  for (Slot& s : w.slots) {
    s.off = off;
    off += (s.bytes + 255) & ~(size_t)255;
  }
  w.bytes = off;
}

struct Src {   // a checkpoint tensor as handed to foley_load_tensor
  const void* p;
  int dt;
  std::vector<int64_t> shape;
};

int launch_pack(WStore& w, const std::string& slot, const Src& src, std::initializer_list<long> size,
                std::initializer_list<long> in_strides, std::initializer_list<long> out_strides, long in_off, long out_off,
                hipStream_t st, const float* scale = nullptr, int scale_dim = -1, int mid_dt = -1) {
  auto it = w.index.find(slot);
  if (it == w.index.end()) return W_FAIL(FOLEY_ERR_INVALID, "internal: unknown packed tensor");
  Slot& s = w.slots[it->second];
  PackDesc d{};
  d.in = src.p;
  d.out = (char*)w.arena + s.off;
  d.in_dt = src.dt;
  d.out_dt = s.dt;
  d.mid_dt = mid_dt;
  for (int k = 0; k < 5; ++k) { d.size[k] = 1; d.is[k] = 0; d.os[k] = 0; }
  int k = 5 - (int)size.size();
  auto a = size.begin();
  auto b = in_strides.begin();
  auto c = out_strides.begin();
  for (; a != size.end(); ++a, ++b, ++c, ++k) { d.size[k] = *a; d.is[k] = *b; d.os[k] = *c; }
  d.in_off = in_off;
  d.out_off = out_off;
  d.scale = scale;
  d.scale_dim = scale_dim >= 0 ? scale_dim + (5 - (int)size.size()) : -1;
  long n = 1;
  for (long v : size) n *= v;
  if (n <= 0) return 0;
  const long blocks = (n + 255) / 256;
  FOLEY_LAUNCH(pack_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return W_FAIL(FOLEY_ERR_HIP, hipGetErrorString(e));
  if (s.parts_seen.insert(out_off).second) s.parts_done += 1;   // parts of a slot are told apart by where they land
  return 0;
}

bool shape_is(const Src& s, std::initializer_list<int64_t> want) {   // trailing singleton dims are ignored
  std::vector<int64_t> a = s.shape, b(want);
  while (a.size() > 1 && a.back() == 1) a.pop_back();
  while (b.size() > 1 && b.back() == 1) b.pop_back();
  return a == b;
}
#define NEED(cond)                                                                                    \
  if (!(cond)) {                                                                                      \
    return W_FAIL(FOLEY_ERR_INVALID, ("tensor '" + key + "' has an unexpected shape").c_str());     \
  }

int plain(WStore& w, const std::string& slot, const Src& s, int64_t N, int64_t K, hipStream_t st, const std::string& key,
          int mid_dt = -1) {
  NEED(K == 1 ? numel(s.shape) == N : shape_is(s, {N, K}));
  return launch_pack(w, slot, s, {N, K}, {K, 1}, {K, 1}, 0, 0, st, nullptr, -1, mid_dt);
}
// [O, I, k] -> [O, k*I] (K index = tap*I + c), optional per-O scale
int conv_pack(WStore& w, const std::string& slot, const Src& s, int64_t O, int64_t I, int64_t k, hipStream_t st,
              const std::string& key, const float* scale = nullptr) {
  NEED(shape_is(s, {O, I, k}) || (k == 1 && shape_is(s, {O, I})));
  return launch_pack(w, slot, s, {O, k, I}, {I * k, 1, k}, {k * I, I, 1}, 0, 0, st, scale, 0);
}
// SwiGLU half `slot01` (0: w1, 1: w3) of [Hh, I(, k)] into alternating 32-row groups of [2*Hh, k*I]
int gate_pack(WStore& w, const std::string& slot, const Src& s, int slot01, int64_t Hh, int64_t I, int64_t k, hipStream_t st,
              const std::string& key) {
  NEED(Hh % 32 == 0 && (shape_is(s, {Hh, I, k}) || (k == 1 && shape_is(s, {Hh, I}))));
  const long KI = k * I;
  return launch_pack(w, slot, s, {Hh / 32, 32, k, I}, {32 * I * k, I * k, 1, k}, {64 * KI, KI, I, 1}, 0, slot01 * 32 * KI, st);
}

int load_dit(WStore& w, const std::string& key, const Src& s, hipStream_t st, bool* handled) {
  const foley_config& f = w.cfg;
  const int64_t D = f.hidden, H = f.heads, hd = D / H;
  *handled = true;
  auto blk = [&](const char* prefix, int* b, std::string* rest) {
    const size_t n = strlen(prefix);
    if (key.compare(0, n, prefix)) return false;
    const size_t dot = key.find('.', n);
    if (dot == std::string::npos) return false;
    *b = atoi(key.substr(n, dot - n).c_str());
    *rest = key.substr(dot + 1);
    return true;
  };
  int b = 0;
  std::string r;
  if (blk("triple_blocks.", &b, &r)) {
    if (b < 0 || b >= f.depth_triple) return W_FAIL(FOLEY_ERR_INVALID, ("block index out of range: " + key).c_str());
    const std::string p = "t" + std::to_string(b) + ".";
    static const struct { const char* ref; const char* packed; int n_mul; int k_kind; } LIN[] = {
        {"audio_mod.linear", "a_mod", 9, 0}, {"v_cond_mod.linear", "v_mod", 9, 0},
        {"audio_self_attn_qkv", "a_qkv", 3, 0}, {"v_cond_attn_qkv", "v_qkv", 3, 0},
        {"audio_self_proj", "a_proj", 1, 0}, {"v_cond_self_proj", "v_proj", 1, 0},
        {"audio_cross_q", "a_cq", 1, 0}, {"v_cond_cross_q", "v_cq", 1, 0}, {"text_cross_kv", "t_kv", 2, 0},
        {"audio_cross_proj", "a_cproj", 1, 0}, {"v_cond_cross_proj", "v_cproj", 1, 0},
        {"audio_mlp.fc1", "a_fc1", -1, 0}, {"v_cond_mlp.fc1", "v_fc1", -1, 0},
        {"audio_mlp.fc2", "a_fc2", 1, 1}, {"v_cond_mlp.fc2", "v_fc2", 1, 1}};
    for (const auto& L : LIN) {
      const std::string m = L.ref;
      const int64_t N = L.n_mul < 0 ? f.mlp_hidden : L.n_mul * D, K = L.k_kind ? f.mlp_hidden : D;
      if (r == m + ".weight") return plain(w, p + L.packed + ".w", s, N, K, st, key);
      if (r == m + ".bias") return plain(w, p + L.packed + ".b", s, N, 1, st, key);
    }
    static const struct { const char* ref; const char* packed; } GAIN[] = {
        {"audio_self_q_norm", "a_qn"}, {"audio_self_k_norm", "a_kn"}, {"v_cond_attn_q_norm", "v_qn"},
        {"v_cond_attn_k_norm", "v_kn"}, {"audio_cross_q_norm", "a_cqn"}, {"v_cond_cross_q_norm", "v_cqn"},
        {"text_cross_k_norm", "t_kn"}};
    for (const auto& G : GAIN)
      if (r == std::string(G.ref) + ".weight") return plain(w, p + G.packed, s, hd, 1, st, key);
    *handled = false;
    return 0;
  }
  if (blk("single_blocks.", &b, &r)) {
    if (b < 0 || b >= f.depth_single) return W_FAIL(FOLEY_ERR_INVALID, ("block index out of range: " + key).c_str());
    const std::string p = "s" + std::to_string(b) + ".";
    const int64_t Hc = f.conv_hidden;
    if (r == "modulation.linear.weight") {
      NEED(shape_is(s, {6 * D, D}));
      return launch_pack(w, "smod_all.w", s, {6 * D, D}, {D, 1}, {D, 1}, 0, (long)b * 6 * D * D, st);
    }
    if (r == "modulation.linear.bias") {
      NEED(numel(s.shape) == 6 * D);
      return launch_pack(w, "smod_all.b", s, {6 * D}, {1}, {1}, 0, (long)b * 6 * D, st);
    }
    if (r == "linear_qkv.weight") {   // rows "(H D K)" -> "(K H D)"  (hifi_foley.py:362)
      NEED(shape_is(s, {3 * D, D}));
      return launch_pack(w, p + "qkv.w", s, {3, H, hd, D}, {D, hd * 3 * D, 3 * D, 1}, {H * hd * D, hd * D, D, 1}, 0, 0, st);
    }
    if (r == "linear_qkv.bias") {
      NEED(numel(s.shape) == 3 * D);
      return launch_pack(w, p + "qkv.b", s, {3, H, hd}, {1, hd * 3, 3}, {H * hd, hd, 1}, 0, 0, st);
    }
    if (r == "q_norm.weight") return plain(w, p + "qn", s, hd, 1, st, key);
    if (r == "k_norm.weight") return plain(w, p + "kn", s, hd, 1, st, key);
    if (r == "linear1.weight") return conv_pack(w, p + "lin1.w", s, D, D, 3, st, key);
    if (r == "linear1.bias") return plain(w, p + "lin1.b", s, D, 1, st, key);
    if (r == "linear2.w1.weight") return gate_pack(w, p + "w13.w", s, 0, Hc, D, 3, st, key);
    if (r == "linear2.w3.weight") return gate_pack(w, p + "w13.w", s, 1, Hc, D, 3, st, key);
    if (r == "linear2.w2.weight") return conv_pack(w, p + "w2.w", s, D, Hc, 3, st, key);
    *handled = false;
    return 0;
  }
  const bool f8time = w.wfmt != 0 && foley_is_half(f.compute_dtype);   // golden g8 "Q14": the first time-embedding bias passes through fp8
  if (key == "audio_embedder.proj.weight") return plain(w, "audio_in.w", s, D, f.latent_dim, st, key);
  if (key == "audio_embedder.proj.bias") return plain(w, "audio_in.b", s, D, 1, st, key);
  if (key == "visual_proj.w1.weight") return gate_pack(w, "vis.w13.w", s, 0, D, f.clip_dim, 1, st, key);
  if (key == "visual_proj.w3.weight") return gate_pack(w, "vis.w13.w", s, 1, D, f.clip_dim, 1, st, key);
  if (key == "visual_proj.w2.weight") return plain(w, "vis.w2.w", s, D, D, st, key);
  if (key == "cond_in.linear_1.weight") return plain(w, "cond1.w", s, D, f.cond_dim, st, key);
  if (key == "cond_in.linear_1.bias") return plain(w, "cond1.b", s, D, 1, st, key);
  if (key == "cond_in.linear_2.weight") return plain(w, "cond2.w", s, D, D, st, key);
  if (key == "cond_in.linear_2.bias") return plain(w, "cond2.b", s, D, 1, st, key);
  if (key == "time_in.mlp.0.weight") return plain(w, "time0.w", s, D, f.time_freq_dim, st, key);
  if (key == "time_in.mlp.0.bias")
    return plain(w, "time0.b", s, D, 1, st, key, f8time ? (w.wfmt == 1 ? FOLEY_F8E4M3 : FOLEY_F8E5M2) : -1);
  if (key == "time_in.mlp.2.weight") return plain(w, "time2.w", s, D, D, st, key);
  if (key == "time_in.mlp.2.bias") return plain(w, "time2.b", s, D, 1, st, key);
  if (key == "sync_in.0.weight") return plain(w, "sync0.w", s, D, f.sync_dim, st, key);
  if (key == "sync_in.0.bias") return plain(w, "sync0.b", s, D, 1, st, key);
  if (key == "sync_in.2.w1.weight") return gate_pack(w, "sync.w13.w", s, 0, f.sync_hidden, D, 1, st, key);
  if (key == "sync_in.2.w3.weight") return gate_pack(w, "sync.w13.w", s, 1, f.sync_hidden, D, 1, st, key);
  if (key == "sync_in.2.w2.weight") return plain(w, "sync.w2.w", s, D, f.sync_hidden, st, key);
  if (key == "sync_pos_emb") {
    NEED(numel(s.shape) == 8 * (int64_t)f.sync_dim);
    return launch_pack(w, "sync_pos", s, {8, f.sync_dim}, {f.sync_dim, 1}, {f.sync_dim, 1}, 0, 0, st);
  }
  if (key == "final_layer.linear.weight") return plain(w, "final.w", s, f.latent_dim, D, st, key);
  if (key == "final_layer.linear.bias") return plain(w, "final.b", s, f.latent_dim, 1, st, key);
  if (key == "empty_clip_feat") return plain(w, "empty_clip", s, f.clip_dim, 1, st, key);
  if (key == "empty_sync_feat") return plain(w, "empty_sync", s, f.sync_dim, 1, st, key);
  if (key.compare(0, 29, "final_layer.adaLN_modulation.") == 0) return 0;   // dead code with 3-D conditioning
  *handled = false;
  return 0;
}

// ---- DAC decoder: weight-normed convs arrive as (g, v) pairs or already folded
struct WnTarget {
  std::string slot;
  int kind;   // 0 conv [O,I,k] -> [O, k*I]; 1 transposed conv [Cin,Cout,2s] -> phases; 2 output conv [1,C,7] -> [7*C]
  int64_t a, b, c;
};

int fold_and_pack(WStore& w, const WnTarget& t, const Src& v, const Src* g, hipStream_t st, const std::string& key) {
  float* scale = nullptr;
  const int64_t rows = v.shape.empty() ? 0 : v.shape[0];
  if (g) {
    NEED(numel(g->shape) == rows && rows > 0);
    W_HIP(hipMalloc((void**)&scale, (size_t)rows * 4));
    w.temps.push_back(scale);
    FOLEY_LAUNCH(wn_scale_kernel, dim3((unsigned)rows), dim3(256), 0, st, v.p, v.dt, g->p, g->dt, numel(v.shape) / rows, scale);
  }
  if (t.kind == 0) return conv_pack(w, t.slot, v, t.a, t.b, t.c, st, key, scale);
  if (t.kind == 1) {   // ConvTranspose1d [Cin, Cout, 2s]: row (phase p, co), K = [x[q-1] | x[q]] <-> taps (p+s | p); g per INPUT channel
    const int64_t Cin = t.a, Cout = t.b, s = t.c;
    NEED(shape_is(v, {Cin, Cout, 2 * s}));
    W_TRY(launch_pack(w, t.slot, v, {s, Cout, Cin}, {1, 2 * s, Cout * 2 * s}, {Cout * 2 * Cin, 2 * Cin, 1}, s, 0, st, scale, 2));
    return launch_pack(w, t.slot, v, {s, Cout, Cin}, {1, 2 * s, Cout * 2 * s}, {Cout * 2 * Cin, 2 * Cin, 1}, 0, Cin, st, scale, 2);
  }
  NEED(shape_is(v, {1, t.a, 7}));
  return launch_pack(w, t.slot, v, {7, t.a}, {1, 7}, {t.a, 1}, 0, 0, st, scale, -1);
}

int stash(WStore& w, std::map<std::string, Pending>& where, const std::string& base, const Src& s, hipStream_t st) {
  Pending p;
  p.dt = s.dt;
  p.shape = s.shape;
  const size_t bytes = (size_t)numel(s.shape) * dt_size(s.dt);
  W_HIP(hipMalloc(&p.buf, bytes ? bytes : 4));
  W_HIP(hipMemcpyAsync(p.buf, s.p, bytes, hipMemcpyDeviceToDevice, st));
  W_HIP(hipStreamSynchronize(st));   // the caller's tensor is only borrowed for this call
  where[base] = p;
  return 0;
}

int load_dac(WStore& w, const std::string& key, const Src& s, hipStream_t st, bool* handled) {
  const foley_config& f = w.cfg;
  const int64_t L = f.latent_dim;
  *handled = true;
  if (key == "post_quant_conv.weight") return plain(w, "dac.pq.w", s, L, L, st, key);
  if (key == "post_quant_conv.bias") return plain(w, "dac.pq.b", s, L, 1, st, key);
  if (key.compare(0, 14, "decoder.model.")) {
    *handled = false;
    return 0;
  }
  const size_t dot = key.find('.', 14);
  if (dot == std::string::npos) { *handled = false; return 0; }
  const int m = atoi(key.substr(14, dot - 14).c_str());
  std::string r = key.substr(dot + 1);
  const int n = f.dac_n_rates;
  std::vector<int64_t> ch(n + 1);
  ch[0] = f.dac_dim;
  for (int i = 0; i < n; ++i) ch[i + 1] = ch[i] / 2;
  WnTarget t{};
  std::string bias_slot, wn_rest;
  int64_t bias_n = 0, bias_rep = 1;
  if (m == 0) {
    t = {"dac.in.w", 0, f.dac_dim, L, 7};
    bias_slot = "dac.in.b"; bias_n = f.dac_dim; wn_rest = r;
  } else if (m >= 1 && m <= n) {
    const int i = m - 1;
    const int64_t cin = ch[i], cout = ch[i + 1], s_ = f.dac_rates[i];
    const std::string p = "dac." + std::to_string(i) + ".";
    if (r.compare(0, 6, "block.")) { *handled = false; return 0; }
    r = r.substr(6);
    if (r == "0.alpha") return plain(w, p + "alpha0", s, cin, 1, st, key);
    if (r.compare(0, 2, "1.") == 0) {
      t = {p + "up.w", 1, cin, cout, s_};
      bias_slot = p + "up.b"; bias_n = cout; bias_rep = s_; wn_rest = r.substr(2);
    } else {
      const int j = r[0] - '2';
      if (j < 0 || j > 2 || r.compare(1, 7, ".block.")) { *handled = false; return 0; }
      const std::string u = p + std::to_string(j) + ".", q = r.substr(8);
      if (q == "0.alpha") return plain(w, u + "a1", s, cout, 1, st, key);
      if (q == "2.alpha") return plain(w, u + "a2", s, cout, 1, st, key);
      if (q.compare(0, 2, "1.") == 0) { t = {u + "c7.w", 0, cout, cout, 7}; bias_slot = u + "c7.b"; }
      else if (q.compare(0, 2, "3.") == 0) { t = {u + "c1.w", 0, cout, cout, 1}; bias_slot = u + "c1.b"; }
      else { *handled = false; return 0; }
      bias_n = cout; wn_rest = q.substr(2);
    }
  } else if (m == n + 1) {
    if (r == "alpha") return plain(w, "dac.out.alpha", s, ch[n], 1, st, key);
    *handled = false;
    return 0;
  } else if (m == n + 2) {
    t = {"dac.out.w", 2, ch[n], 0, 0};
    bias_slot = "dac.out.b"; bias_n = 1; wn_rest = r;
  } else {
    *handled = false;
    return 0;
  }
  if (wn_rest == "bias") {
    NEED(numel(s.shape) == bias_n);
    return launch_pack(w, bias_slot, s, {bias_rep, bias_n}, {0, 1}, {bias_n, 1}, 0, 0, st);
  }
  const std::string base = key.substr(0, key.size() - wn_rest.size());
  const bool is_g = wn_rest == "parametrizations.weight.original0" || wn_rest == "weight_g";
  const bool is_v = wn_rest == "parametrizations.weight.original1" || wn_rest == "weight_v";
  if (wn_rest == "weight") return fold_and_pack(w, t, s, nullptr, st, key);   // already folded
  if (!is_g && !is_v) { *handled = false; return 0; }
  auto& mine = is_g ? w.pend_g : w.pend_v;
  auto& other = is_g ? w.pend_v : w.pend_g;
  auto it = other.find(base);
  if (it == other.end()) return stash(w, mine, base, s, st);
  Src o{it->second.buf, it->second.dt, it->second.shape};
  const int rc = is_g ? fold_and_pack(w, t, o, &s, st, key) : fold_and_pack(w, t, s, &o, st, key);
  hipStreamSynchronize(st);
  hipFree(it->second.buf);
  other.erase(it);
  return rc;
}

WStore* store_of(foley_ctx* c) { return c ? (WStore*)*foley_ctx_wstore_slot(c) : nullptr; }

}  // namespace

// --------------------------------------------------------------------------- C ABI
extern "C" int foley_weights_begin(foley_ctx* c, int weight_format) {
  if (!c || weight_format < 0 || weight_format > 2) return W_FAIL(FOLEY_ERR_INVALID, "bad argument");
  const foley_config* cfg = foley_ctx_config(c);
  if (weight_format && !foley_is_half(cfg->compute_dtype))
    return W_FAIL(FOLEY_ERR_INVALID, "fp8 weight storage needs bf16 / fp16 compute");
  W_HIP(hipSetDevice(foley_ctx_device(c)));
  void** slot = foley_ctx_wstore_slot(c);
  if (*slot) {
    wstore_free(*slot);
    *slot = nullptr;
  }
  WStore* w = new WStore();
  w->device = foley_ctx_device(c);
  w->cfg = *cfg;
  w->wfmt = weight_format;
  build_layout(*w);
  hipError_t e = hipMalloc(&w->arena, w->bytes ? w->bytes : 256);
  if (e != hipSuccess) {
    delete w;
    return W_FAIL(FOLEY_ERR_HIP, hipGetErrorString(e));
  }
  *slot = w;
  foley_ctx_set_wstore_free(c, wstore_free);
  for (const Slot& s : w->slots)   // registered up front: the addresses never change
    W_TRY(foley_set_tensor(c, s.name.c_str(), (char*)w->arena + s.off, s.dt, (int)s.shape.size(), s.shape.data()));
  w->begun = true;
  return 0;
}

extern "C" int foley_load_tensor(foley_ctx* c, const char* ref_key, const void* dev_ptr, int dtype, int ndim,
                                 const int64_t* shape, void* stream) {
  WStore* w = store_of(c);
  if (!w || !w->begun) return W_FAIL(FOLEY_ERR_STATE, "foley_weights_begin has not been called");
  if (!ref_key || !dev_ptr || ndim < 0 || ndim > 8 || (ndim && !shape)) return W_FAIL(FOLEY_ERR_INVALID, "bad argument");
  if (!(dtype == FOLEY_DT_F32 || dtype == FOLEY_DT_BF16 || dtype == FOLEY_DT_F8E4M3 || dtype == FOLEY_DT_F8E5M2 || dtype == DT_F16))
    return W_FAIL(FOLEY_ERR_INVALID, "checkpoint tensors must be f32 / bf16 / f16 / fp8");
  W_HIP(hipSetDevice(w->device));
  Src s{dev_ptr, dtype, std::vector<int64_t>(shape, shape + ndim)};
  const std::string key(ref_key);
  bool handled = false;
  W_TRY(load_dit(*w, key, s, (hipStream_t)stream, &handled));
  if (handled) return 0;
  W_TRY(load_dac(*w, key, s, (hipStream_t)stream, &handled));
  return handled ? 0 : 1;   // 1: not a tensor of the sampling path (DAC encoder, quantizer, ...) - ignored
}

extern "C" int foley_weights_end(foley_ctx* c, void* stream) {
  WStore* w = store_of(c);
  if (!w || !w->begun) return W_FAIL(FOLEY_ERR_STATE, "foley_weights_begin has not been called");
  W_HIP(hipSetDevice(w->device));
  W_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (void* t : w->temps) hipFree(t);
  w->temps.clear();
  if (!w->pend_g.empty() || !w->pend_v.empty()) {
    const std::string k = w->pend_g.empty() ? w->pend_v.begin()->first : w->pend_g.begin()->first;
    return W_FAIL(FOLEY_ERR_MISSING, ("weight-norm pair incomplete: " + k).c_str());
  }
  for (const Slot& s : w->slots)
    if (s.parts_done < s.parts_needed) return W_FAIL(FOLEY_ERR_MISSING, ("packed tensor '" + s.name + "' was not (fully) loaded").c_str());
  return 0;
}

extern "C" int foley_weights_arena(foley_ctx* c, void** dev_ptr, uint64_t* bytes) {
  WStore* w = store_of(c);
  if (!w || !w->begun || !dev_ptr || !bytes) return W_FAIL(FOLEY_ERR_STATE, "foley_weights_begin has not been called");
  *dev_ptr = w->arena;
  *bytes = w->bytes;
  return 0;
}

extern "C" int foley_weights_mark_received(foley_ctx* c) {
  WStore* w = store_of(c);
  if (!w || !w->begun) return W_FAIL(FOLEY_ERR_STATE, "foley_weights_begin has not been called");
  for (Slot& s : w->slots) s.parts_done = s.parts_needed;
  return 0;
}

// One collective for the whole model: ncclBroadcast of the arena bytes on the caller's communicator.
// RCCL is resolved from the process (the host framework has loaded it; same library instance that
// created `nccl_comm`), falling back to the system librccl.
extern "C" int foley_bcast_weights(foley_ctx* c, void* nccl_comm, int root, void* stream) {
  WStore* w = store_of(c);
  if (!w || !w->begun) return W_FAIL(FOLEY_ERR_STATE, "foley_weights_begin has not been called");
  if (!nccl_comm) return W_FAIL(FOLEY_ERR_INVALID, "null communicator");
  typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  static bcast_fn fn = nullptr;
  if (!fn) {
    fn = (bcast_fn)dlsym(RTLD_DEFAULT, "ncclBroadcast");
    for (const char* lib : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (fn) break;
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (!h) h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (h) fn = (bcast_fn)dlsym(h, "ncclBroadcast");
    }
    if (!fn) return W_FAIL(FOLEY_ERR_STATE, "ncclBroadcast not found (RCCL is not loaded)");
  }
  W_HIP(hipSetDevice(w->device));
  const int rc = fn(w->arena, w->arena, w->bytes, /*ncclUint8*/ 1, root, nccl_comm, (hipStream_t)stream);
  if (rc != 0) return W_FAIL(FOLEY_ERR_HIP, "ncclBroadcast failed");
  return foley_weights_mark_received(c);
}

// ---------------------------------------------------------------------------------------------
// Single-process data parallelism (one host process that sees all GPUs of the node - the ComfyUI case, host/sampler.py
// replicate()): every buffer of the root device reaches the other devices in ONE grouped RCCL launch - ncclCommInitAll over
// the device list, then ncclGroupStart / one ncclBroadcast per (device, buffer) / ncclGroupEnd - instead of N-1 serial peer
// copies.  Communicators are cached per device list.  RCCL is resolved from the process like in foley_bcast_weights.
namespace {
struct Rccl {
  typedef int (*init_all_fn)(void**, int, const int*);
  typedef int (*group_fn)();
  typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  init_all_fn init_all = nullptr;
  group_fn gstart = nullptr, gend = nullptr;
  bcast_fn bcast = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r = []() {
    Rccl q;
    void* h = nullptr;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(RTLD_DEFAULT, name);
      if (!p && h) p = dlsym(h, name);
      return p;
    };
    if (!dlsym(RTLD_DEFAULT, "ncclCommInitAll"))
      for (const char* lib : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (!h) h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    q.init_all = (Rccl::init_all_fn)sym("ncclCommInitAll");
    q.gstart = (Rccl::group_fn)sym("ncclGroupStart");
    q.gend = (Rccl::group_fn)sym("ncclGroupEnd");
    q.bcast = (Rccl::bcast_fn)sym("ncclBroadcast");
    q.ok = q.init_all && q.gstart && q.gend && q.bcast;
    return q;
  }();
  return r;
}
std::mutex g_local_mu;
std::map<std::vector<int>, std::vector<void*>> g_local_comms;
}  // namespace

extern "C" int foley_bcast_local(int ndev, const int* devices, int nbuf, void* const* bufs, const uint64_t* bytes) {
  if (ndev < 1 || !devices || nbuf < 1 || !bufs || !bytes) return W_FAIL(FOLEY_ERR_INVALID, "bad argument");
  Rccl& r = rccl();
  if (!r.ok) return W_FAIL(FOLEY_ERR_STATE, "RCCL (ncclCommInitAll / ncclGroupStart / ncclBroadcast) not found in the process");
  std::lock_guard<std::mutex> lk(g_local_mu);
  int prev = 0;
  hipGetDevice(&prev);
  const std::vector<int> key(devices, devices + ndev);
  auto it = g_local_comms.find(key);
  if (it == g_local_comms.end()) {
    std::vector<void*> comms((size_t)ndev, nullptr);
    if (r.init_all(comms.data(), ndev, devices) != 0) {
      hipSetDevice(prev);
      return W_FAIL(FOLEY_ERR_HIP, "ncclCommInitAll failed (duplicate or unreachable devices?)");
    }
    it = g_local_comms.emplace(key, comms).first;
  }
  int rc = r.gstart();
  for (int d = 0; d < ndev && rc == 0; ++d) {
    if (hipSetDevice(devices[d]) != hipSuccess) { rc = -1; break; }
    for (int i = 0; i < nbuf && rc == 0; ++i)
      rc = r.bcast(bufs[(size_t)d * nbuf + i], bufs[(size_t)d * nbuf + i], (size_t)bytes[i], /*ncclUint8*/ 1, /*root rank*/ 0,
                   it->second[(size_t)d], (hipStream_t)0);   // in place on every rank (the send buffer only counts on the root)
  }
  const int rc2 = r.gend();
  for (int d = 0; d < ndev; ++d)
    if (hipSetDevice(devices[d]) == hipSuccess) hipDeviceSynchronize();
  hipSetDevice(prev);
  if (rc != 0 || rc2 != 0) return W_FAIL(FOLEY_ERR_HIP, "grouped ncclBroadcast failed");
  return 0;
}
