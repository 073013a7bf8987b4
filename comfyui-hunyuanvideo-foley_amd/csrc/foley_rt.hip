// Runtime of libfoley_hip.so: context, packed-weight registry, step-invariant precompute, the DiT
// forward, the device-resident sampler loop and the DAC decoder - all as sequences of launches
// of the kernels in gemm*.hip / attention.hip / rowops.hip on the caller's stream.  C ABI in
// include/foley_hip.h.  No torch types, no CPU fallback.
#include "../../include/foley_hip.h"
#include "kernels.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

thread_local FoleyProfHook g_foley_prof = {nullptr, nullptr};

// Contexts are independent, but several of them may be driven from several threads of one process (node-level data
// parallelism: host/sampler.py::denoise_process_multi runs one context per visible GPU, one thread each).  The set-up phases - workspace allocation and the
// stream capture of the iteration graph - are serialised process-wide (allocation calls from one thread
// while another thread captures are not safe in every HIP runtime); the long replay loops run concurrently.
static std::mutex g_setup_mutex;

// --------------------------------------------------------------------------- errors
static thread_local std::string g_err;
int foley_set_err(const char* msg, const char* file, int line) {
  const char* base = strrchr(file, '/');
  g_err = std::string(msg) + " (" + (base ? base + 1 : file) + ":" + std::to_string(line) + ")";
  return FOLEY_ERR_INVALID;
}
#define FAIL(code, msg) (foley_set_err(msg, __FILE__, __LINE__), (code))
#define TRY(expr)            \
  do {                       \
    int _rc = (expr);        \
    if (_rc != 0) return _rc; \
  } while (0)
#define HIPTRY(expr)                                                     \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) {                                              \
      foley_set_err(hipGetErrorString(_e), __FILE__, __LINE__);          \
      return FOLEY_ERR_HIP;                                              \
    }                                                                    \
  } while (0)

// --------------------------------------------------------------------------- context
struct TensorRef {
  const void* p = nullptr;
  int dtype = 0;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct DevBuf {  // context-owned workspace block
  void* p = nullptr;
  size_t bytes = 0;
};

struct Lin {  // a packed linear / conv-as-GEMM layer
  const void* w = nullptr;
  const float* b = nullptr;
  int N = 0, K = 0;
  int wfmt = 0;   // 0: weights in the compute dtype; 1 / 2: stored fp8 e4m3fn / e5m2 (kernels.h GemmArgs::wfmt)
};

// Weights of the DiT forward resolved once per registration (foley_prepare) instead of ~600 string
// lookups per eager iteration; index 0 = audio stream, 1 = visual stream.
struct TripleW {
  Lin qkv[2], proj[2], cq[2], cproj[2], fc1[2], fc2[2];
  const float *qn[2], *kn[2], *cqn[2];
};
struct SingleW {
  Lin qkv, lin1, w13, w2;
  const float *qn, *kn;
};
struct ForwardW {
  std::vector<TripleW> t;
  std::vector<SingleW> s;
  Lin audio_in, fin, smod;
  bool ok = false;
};

// Per-launch HIP-event brackets of one eager forward (foley_profile_forward): label -> time / work.
struct ProfRec {
  const char* label;
  double flop, bytes;
  hipEvent_t e0, e1;
  const void* fn;   // the kernel the op's launch ran (null: the op launched nothing)
};
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e = nullptr;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
};

struct foley_ctx {
  int device = 0;
  foley_config cfg{};
  std::unordered_map<std::string, TensorRef> tensors;
  ForwardW fw;
  Profiler prof;
  // run state (valid after foley_prepare)
  bool prepared = false;
  foley_plan plan{};
  std::vector<DevBuf> owned;        // everything hipMalloc'ed for the current plan
  // prepared tables
  float* vec_table = nullptr;       // [n_iter, D]
  float* modtab = nullptr;          // [n_triple][2][n_iter][9D]
  void* txt_k = nullptr;            // [n_triple][ncfg, H, Lt, 128]     (same dtype rule as Q/K/V)
  void* txt_v = nullptr;            // [n_triple][ncfg, H, Lt, 128] or transposed [.., 128, ceil32(Lt)]
  float* v_cond0 = nullptr;         // [ncfg, Lv, D]
  float* sync_tok = nullptr;        // [ncfg, Ls, D] sync tokens after sync_in; audio frame l reads row nearest_exact(l) (RowBcast mode 2)
  int sync_per = 0;                 // 8 when the token rows of the first sync_lead cfg halves repeat with period 8 (empty sync features), else 0
  int sync_lead = 0;                // number of leading 8-periodic halves: ncfg = all of them (text-to-audio); 1 of 2 = a video clip under
                                    // CFG (unconditional half first, utils.py:150-176); 0 = none
  int* flag = nullptr;              // device scratch word of the periodicity check
  int* ident_idx = nullptr;         // 0..max(Lv,La)-1
  // forward workspace
  void* xin = nullptr;              // T [M, C]
  float* audio = nullptr;           // [M, D]
  float* vcond = nullptr;           // [Mv, D]
  void* xn_a = nullptr;             // T [M, D]
  void* xn_v = nullptr;             // T [Mv, D]
  float* qkv_a = nullptr;           // [M, 3D]
  float* qkv_v = nullptr;           // [Mv, 3D]
  void* Q = nullptr;                // [Bc, H, S, 128]   fp32 (parity mode) or bf16
  void* K = nullptr;
  void* V = nullptr;                // fp32 [Bc, H, S, 128] or bf16 transposed [Bc, H, 128, ceil32(S)]
  void* att_a = nullptr;            // T [M, D]
  void* att_v = nullptr;            // T [Mv, D]
  void* hid_a = nullptr;            // T [M, max(mlp_hidden, conv_hidden)]
  void* hid_v = nullptr;            // T [Mv, mlp_hidden]
  void* svec = nullptr;             // T [ncfg*Ls, D]
  float* smod = nullptr;            // [ncfg*Ls, n_single*6D]
  // the same table for EVERY loop iteration, built by foley_prepare when it fits (it depends on the iteration index only):
  // [n_iter][ncfg*P][n_single*6D] fp32, P = sync_per or Ls; consumers add step * smod_step to their row address
  DevBuf smod_tab, svec_tab;
  bool smod_hoisted = false;
  float* pred = nullptr;            // [M, C]
  float *part_a = nullptr, *part_v = nullptr;   // deferred split-K partial products [PART_CAP][M | Mv][D]
  float* x_saved = nullptr;         // [clips, C, La]
  float* d_acc = nullptr;
  int* step_ctr = nullptr;
  float* x_cur = nullptr;           // [clips, C, La] the sample being denoised (ctx-owned => stable address)
  // ctx-owned copies of the plan's lookup tables (stable addresses across foley_prepare calls)
  float *rope_cos = nullptr, *rope_sin = nullptr, *solver_coef = nullptr;
  int *pos_audio_self = nullptr, *pos_visual_self = nullptr, *pos_linear = nullptr, *sync_gather = nullptr;
  int* rep_idx = nullptr;           // [clips*Lv] j -> j % Lv: replicates the visual projection per clip in one launch
  // rotation rows gathered per token for the fused head-split epilogues: rot_*[k][l] = rope_{cos,sin}[pos_k[l]], [len, 64] fp32
  // (k: 0 audio self, 1 visual self, 2 linear positions)
  float *rot_cos[3] = {nullptr, nullptr, nullptr}, *rot_sin[3] = {nullptr, nullptr, nullptr};
  void *tA = nullptr, *tB = nullptr;  // precompute scratch
  float* tF = nullptr;
  bool have_buffers = false;        // workspace allocated for `plan`'s dimensions
  // graph: one captured loop iteration; valid while the workspace and weights stay put
  hipGraphExec_t graph_exec = nullptr;
  float graph_guidance = 0.f;
  // side stream: work that depends only on the iteration index (single-block AdaLN GEMMs) overlaps
  // the latent-dependent chain; joined through events (also inside the captured graph)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_mod;
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  std::atomic<int> abort_req{0};   // foley_abort(): checked by foley_sample between iterations
  // DAC workspace (grown on demand)
  DevBuf dacP, dacQ, dacR, dacZ;
  // reference-keyed weight store (weights.hip): ctx-owned packed arena
  void* wstore = nullptr;
  void (*wstore_free)(void*) = nullptr;
};

// internal hooks for weights.hip (not part of the C ABI)
void** foley_ctx_wstore_slot(foley_ctx* c) { return &c->wstore; }
void foley_ctx_set_wstore_free(foley_ctx* c, void (*fn)(void*)) { c->wstore_free = fn; }
const foley_config* foley_ctx_config(foley_ctx* c) { return &c->cfg; }
int foley_ctx_device(foley_ctx* c) { return c->device; }

static size_t esize(int dtype) { return foley_is_half(dtype) ? 2 : (dtype == FOLEY_F8E4M3 || dtype == FOLEY_F8E5M2) ? 1 : 4; }

static int ctx_alloc(foley_ctx* c, size_t bytes, void** out) {
  void* p = nullptr;
  bytes = (bytes + 255) & ~(size_t)255;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 256);
  if (e != hipSuccess) return FAIL(FOLEY_ERR_HIP, hipGetErrorString(e));
  c->owned.push_back({p, bytes});
  *out = p;
  return 0;
}

static void ctx_free_plan(foley_ctx* c) {
  if (c->graph_exec) {
    hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
  }
  for (auto& b : c->owned) hipFree(b.p);
  c->owned.clear();
  c->prepared = false;
  c->have_buffers = false;
}

static void ctx_drop_graph(foley_ctx* c) {
  if (c->graph_exec) {
    hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
  }
}

static bool same_dims(const foley_plan& a, const foley_plan& b) {
  return a.ncfg == b.ncfg && a.clips == b.clips && a.La == b.La && a.Lv == b.Lv && a.Ls == b.Ls && a.Lt == b.Lt &&
         a.n_iter == b.n_iter && a.rope_len == b.rope_len;
}

static int grow(DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return 0;
  if (b.p) hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) return FAIL(FOLEY_ERR_HIP, hipGetErrorString(e));
  b.bytes = bytes;
  return 0;
}

static int get_tensor(foley_ctx* c, const std::string& name, int dtype, std::initializer_list<int64_t> shape,
                      const void** out) {
  auto it = c->tensors.find(name);
  if (it == c->tensors.end()) {
    g_err = "tensor '" + name + "' was not registered";
    return FOLEY_ERR_MISSING;
  }
  const TensorRef& t = it->second;
  bool ok = t.dtype == dtype && t.shape.size() == shape.size();
  if (ok) {
    size_t i = 0;
    for (auto s : shape) ok = ok && (t.shape[i++] == s);
  }
  if (!ok) {
    g_err = "tensor '" + name + "' has the wrong dtype/shape";
    return FOLEY_ERR_INVALID;
  }
  *out = t.p;
  return 0;
}

// --------------------------------------------------------------------------- zero page
// 256 zero bytes per device: the direct-to-LDS GEMM loop reads masked rows from here.
static const void* zero_page() {
  static void* pages[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!pages[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    pages[dev] = p;
  }
  return pages[dev];
}

// --------------------------------------------------------------------------- launch helpers
static RowBcast rb_none() { return RowBcast{nullptr, 0, 0, 1, 1, nullptr, 0}; }
static RowBcast rb_vec(const float* base, long step_stride, const int* step_ptr);

// K ranges whose partial products a gated-residual GEMM may leave for the next LayerNorm to sum
// (deferred split-K, kernels.h GemmArgs::partials)
constexpr int PART_CAP = 8;

static int get_lin(foley_ctx* c, const std::string& name, int dtype, int N, int K, bool bias, Lin* out) {
  const void* w;
  out->wfmt = 0;
  auto it = c->tensors.find(name + ".w");
  if (it != c->tensors.end() && foley_is_half(dtype) && (it->second.dtype == FOLEY_F8E4M3 || it->second.dtype == FOLEY_F8E5M2)) {
    // fp8 weight-only storage (reference FP8WeightWrapper): the GEMM widens in registers
    out->wfmt = it->second.dtype == FOLEY_F8E4M3 ? 1 : 2;
    TRY(get_tensor(c, name + ".w", it->second.dtype, {N, K}, &w));
  } else {
    TRY(get_tensor(c, name + ".w", dtype, {N, K}, &w));
  }
  out->w = w;
  out->N = N;
  out->K = K;
  out->b = nullptr;
  if (bias) {
    const void* b;
    TRY(get_tensor(c, name + ".b", FOLEY_F32, {N}, &b));
    out->b = (const float*)b;
  }
  return 0;
}

// plain linear layer over M rows
static GemmArgs gemm_plain(const void* A, int M, const Lin& l, void* out, long ldc) {
  GemmArgs g{};
  g.A = A; g.W = l.w; g.bias = l.b;
  g.M = M; g.N = l.N; g.K = l.K; g.lda = l.K;
  g.wfmt = l.wfmt;
  g.segV = M > 0 ? M : 1; g.segS = g.segV; g.taps = 1; g.tapC = l.K; g.dil = 1; g.tap0 = 0;
  g.out0 = out; g.out1 = nullptr;
  g.osegV = g.segV; g.out_seg = 0; g.out_row = ldc; g.out_shift = 0; g.out_check = 0;
  g.rb = rb_none(); g.res = nullptr; g.alpha = nullptr; g.alphaC = 1;
  g.zeros = zero_page();
  return g;
}

// channels-last conv (k taps, dilation d, 'same' padding) over segments of `seg` rows
static GemmArgs gemm_conv(const void* A, int M, int seg, int C, int taps, int dil, const Lin& l, void* out,
                          long ldc) {
  GemmArgs g = gemm_plain(A, M, l, out, ldc);
  g.lda = C;
  g.segV = seg; g.segS = seg; g.taps = taps; g.tapC = C; g.dil = dil; g.tap0 = -((taps - 1) / 2) * dil;
  return g;
}

// --------------------------------------------------------------------------- C ABI: context
extern "C" uint32_t foley_abi_version(void) { return FOLEY_ABI_VERSION; }
extern "C" const char* foley_last_error(void) { return g_err.c_str(); }

extern "C" int foley_ctx_create(int device, const foley_config* cfg, foley_ctx** out) {
  if (!cfg || !out) return FAIL(FOLEY_ERR_INVALID, "null argument");
  if (cfg->hidden % cfg->heads || cfg->hidden / cfg->heads != 128)
    return FAIL(FOLEY_ERR_INVALID, "head_dim must be 128");
  if (cfg->compute_dtype != FOLEY_DT_F32 && cfg->compute_dtype != FOLEY_DT_BF16 && cfg->compute_dtype != FOLEY_DT_F16)
    return FAIL(FOLEY_ERR_INVALID, "compute_dtype must be f32, bf16 or f16");
  if (cfg->dac_n_rates < 1 || cfg->dac_n_rates > 8) return FAIL(FOLEY_ERR_INVALID, "bad dac_n_rates");
  int ndev = 0;
  HIPTRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return FAIL(FOLEY_ERR_INVALID, "no such HIP device");
  HIPTRY(hipSetDevice(device));
  foley_ctx* c = new foley_ctx();
  c->device = device;
  c->cfg = *cfg;
  (void)hipEventCreate(&c->ev0);
  (void)hipEventCreate(&c->ev1);
  (void)hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  c->ev_mod.resize(cfg->depth_single > 0 ? cfg->depth_single : 1);
  for (auto& e : c->ev_mod) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  *out = c;
  return 0;
}

extern "C" void foley_ctx_destroy(foley_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  ctx_free_plan(c);
  for (DevBuf* b : {&c->dacP, &c->dacQ, &c->dacR, &c->dacZ, &c->smod_tab, &c->svec_tab})
    if (b->p) hipFree(b->p);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->side) hipStreamDestroy(c->side);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  for (auto e : c->ev_mod) hipEventDestroy(e);
  for (auto e : c->prof.pool) hipEventDestroy(e);
  if (c->wstore && c->wstore_free) c->wstore_free(c->wstore);
  delete c;
}

extern "C" int foley_set_tensor(foley_ctx* c, const char* name, const void* p, int dtype, int ndim,
                                const int64_t* shape) {
  if (!c || !name || !p || ndim < 0 || ndim > 8) return FAIL(FOLEY_ERR_INVALID, "bad tensor registration");
  if (((uintptr_t)p) & 15) return FAIL(FOLEY_ERR_INVALID, "tensor pointers must be 16-byte aligned");
  TensorRef t;
  t.p = p;
  t.dtype = dtype;
  t.shape.assign(shape, shape + ndim);
  auto it = c->tensors.find(name);
  if (it != c->tensors.end() && it->second.p != p) ctx_drop_graph(c);  // captured kernels hold the old address
  c->tensors[name] = t;
  c->fw.ok = false;     // resolved pointers are refreshed by the next foley_prepare
  c->prepared = false;  // cached tables depend on the weights
  return 0;
}

extern "C" int foley_last_elapsed_ms(foley_ctx* c, float* ms) {
  if (!c || !ms || !c->timed) return FAIL(FOLEY_ERR_STATE, "nothing timed yet");
  HIPTRY(hipEventSynchronize(c->ev1));
  HIPTRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

// --------------------------------------------------------------------------- resolved weights
static int get_vec(foley_ctx* c, const std::string& name, int n, const float** out) {
  const void* p;
  TRY(get_tensor(c, name, FOLEY_F32, {n}, &p));
  *out = (const float*)p;
  return 0;
}

static int resolve_forward_weights(foley_ctx* c) {
  const foley_config& f = c->cfg;
  const int D = f.hidden, T = f.compute_dtype, Hc = f.conv_hidden;
  ForwardW& w = c->fw;
  w.ok = false;
  w.t.assign(f.depth_triple, TripleW{});
  w.s.assign(f.depth_single, SingleW{});
  for (int b = 0; b < f.depth_triple; ++b) {
    TripleW& t = w.t[b];
    for (int s = 0; s < 2; ++s) {
      const std::string p = "t" + std::to_string(b) + (s ? ".v_" : ".a_");
      TRY(get_lin(c, p + "qkv", T, 3 * D, D, true, &t.qkv[s]));
      TRY(get_lin(c, p + "proj", T, D, D, true, &t.proj[s]));
      TRY(get_lin(c, p + "cq", T, D, D, true, &t.cq[s]));
      TRY(get_lin(c, p + "cproj", T, D, D, true, &t.cproj[s]));
      TRY(get_lin(c, p + "fc1", T, f.mlp_hidden, D, true, &t.fc1[s]));
      TRY(get_lin(c, p + "fc2", T, D, f.mlp_hidden, true, &t.fc2[s]));
      TRY(get_vec(c, p + "qn", 128, &t.qn[s]));
      TRY(get_vec(c, p + "kn", 128, &t.kn[s]));
      TRY(get_vec(c, p + "cqn", 128, &t.cqn[s]));
    }
  }
  for (int b = 0; b < f.depth_single; ++b) {
    SingleW& q = w.s[b];
    const std::string p = "s" + std::to_string(b) + ".";
    TRY(get_lin(c, p + "qkv", T, 3 * D, D, true, &q.qkv));
    TRY(get_lin(c, p + "lin1", T, D, 3 * D, true, &q.lin1));
    TRY(get_lin(c, p + "w13", T, 2 * Hc, 3 * D, false, &q.w13));
    TRY(get_lin(c, p + "w2", T, D, 3 * Hc, false, &q.w2));
    TRY(get_vec(c, p + "qn", 128, &q.qn));
    TRY(get_vec(c, p + "kn", 128, &q.kn));
  }
  TRY(get_lin(c, "audio_in", T, D, f.latent_dim, true, &w.audio_in));
  TRY(get_lin(c, "final", T, f.latent_dim, D, true, &w.fin));
  if (f.depth_single > 0) TRY(get_lin(c, "smod_all", T, f.depth_single * 6 * D, D, true, &w.smod));
  w.ok = true;
  return 0;
}

// --------------------------------------------------------------------------- prepare
// Everything that does not depend on the latents: time embedding for every loop
// iteration, the triple blocks' AdaLN tables, text K/V per block, cond/visual/sync embedders.
extern "C" int foley_prepare(foley_ctx* c, const foley_plan* pl, void* stream_v) {
  if (!c || !pl) return FAIL(FOLEY_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> setup_lock(g_setup_mutex);
  hipStream_t st = (hipStream_t)stream_v;
  HIPTRY(hipSetDevice(c->device));
  const foley_config& f = c->cfg;
  if (pl->ncfg < 1 || pl->ncfg > 2 || pl->clips < 1 || pl->La < 1 || pl->Lv < 1 || pl->Ls < 8 || pl->Ls % 8 ||
      pl->Lt < 1 || pl->n_iter < 1)
    return FAIL(FOLEY_ERR_INVALID, "bad plan dimensions");
  if (pl->rope_len < 2 * pl->La) return FAIL(FOLEY_ERR_INVALID, "rope table shorter than 2*La");
  HIPTRY(hipStreamSynchronize(st));
  const bool reuse = c->have_buffers && same_dims(c->plan, *pl);
  if (!reuse) ctx_free_plan(c);
  c->prepared = false;

  const int D = f.hidden, H = f.heads, C = f.latent_dim, T = f.compute_dtype;
  const size_t es = esize(T);
  const int ncfg = pl->ncfg, clips = pl->clips, La = pl->La, Lv = pl->Lv, Ls = pl->Ls, Lt = pl->Lt;
  const int Bc = ncfg * clips, M = Bc * La, Mv = Bc * Lv, S = La + Lv, NI = pl->n_iter;
  const int hidmax = f.mlp_hidden > f.conv_hidden ? f.mlp_hidden : f.conv_hidden;
  const int Lmax = std::max(std::max(La, Lv), Lt);
  const int rmax = std::max(std::max(NI, ncfg * Lt), std::max(ncfg * Lv, ncfg * Ls));
  // widest row any precompute stage writes into the scratch (every configurable feature width)
  const size_t tcols = std::max({D, f.sync_hidden, f.cond_dim, f.clip_dim, f.sync_dim, f.time_freq_dim});

#define ALLOC(ptr, bytes) TRY(ctx_alloc(c, (bytes), (void**)&(ptr)))
  if (!reuse) {
    ALLOC(c->vec_table, (size_t)NI * D * 4);
    ALLOC(c->modtab, (size_t)f.depth_triple * 2 * NI * 9 * D * 4);
    ALLOC(c->txt_k, (size_t)f.depth_triple * ncfg * H * Lt * 128 * es);
    ALLOC(c->txt_v, (size_t)f.depth_triple * ncfg * H * ((Lt + 31) & ~31) * 128 * es);
    HIPTRY(hipMemsetAsync(c->txt_v, 0, (size_t)f.depth_triple * ncfg * H * ((Lt + 31) & ~31) * 128 * es, st));
    ALLOC(c->v_cond0, (size_t)ncfg * Lv * D * 4);
    ALLOC(c->sync_tok, (size_t)ncfg * Ls * D * 4);
    ALLOC(c->flag, 256);
    ALLOC(c->xin, (size_t)M * C * es);
    ALLOC(c->audio, (size_t)M * D * 4);
    ALLOC(c->vcond, (size_t)Mv * D * 4);
    ALLOC(c->xn_a, (size_t)M * D * es);
    ALLOC(c->xn_v, (size_t)Mv * D * es);
    ALLOC(c->qkv_a, (size_t)M * 3 * D * 4);
    ALLOC(c->qkv_v, (size_t)Mv * 3 * D * 4);
    ALLOC(c->Q, (size_t)Bc * H * S * 128 * es);
    ALLOC(c->K, (size_t)Bc * H * S * 128 * es);
    ALLOC(c->V, (size_t)Bc * H * ((S + 31) & ~31) * 128 * es);
    HIPTRY(hipMemsetAsync(c->V, 0, (size_t)Bc * H * ((S + 31) & ~31) * 128 * es, st));  // V^T pad stays finite
    ALLOC(c->att_a, (size_t)M * D * es);
    ALLOC(c->att_v, (size_t)Mv * D * es);
    ALLOC(c->hid_a, (size_t)M * hidmax * es);
    ALLOC(c->hid_v, (size_t)Mv * f.mlp_hidden * es);
    ALLOC(c->svec, (size_t)ncfg * Ls * D * es);
    ALLOC(c->smod, (size_t)f.depth_single * ncfg * Ls * 6 * D * 4);
    ALLOC(c->pred, (size_t)M * C * 4);
    ALLOC(c->part_a, (size_t)PART_CAP * M * D * 4);     // sized for fp32 slabs; 16-bit slabs (slab16()) use half of it
    ALLOC(c->part_v, (size_t)PART_CAP * Mv * D * 4);
    ALLOC(c->x_saved, (size_t)clips * C * La * 4);
    ALLOC(c->d_acc, (size_t)clips * C * La * 4);
    ALLOC(c->x_cur, (size_t)clips * C * La * 4);
    ALLOC(c->step_ctr, 256);
    ALLOC(c->rope_cos, (size_t)pl->rope_len * 64 * 4);
    ALLOC(c->rope_sin, (size_t)pl->rope_len * 64 * 4);
    ALLOC(c->solver_coef, (size_t)NI * 8 * 4);
    ALLOC(c->pos_audio_self, (size_t)La * 4);
    ALLOC(c->pos_visual_self, (size_t)Lv * 4);
    ALLOC(c->pos_linear, (size_t)Lmax * 4);
    ALLOC(c->sync_gather, (size_t)La * 4);
    ALLOC(c->rep_idx, (size_t)clips * Lv * 4);
    {
      const int rl[3] = {La, Lv, Lmax};
      for (int k = 0; k < 3; ++k) {
        ALLOC(c->rot_cos[k], (size_t)rl[k] * 64 * 4);
        ALLOC(c->rot_sin[k], (size_t)rl[k] * 64 * 4);
      }
    }
    {
      std::vector<int> idx((size_t)clips * Lv);
      for (size_t j = 0; j < idx.size(); ++j) idx[j] = (int)(j % Lv);
      HIPTRY(hipMemcpy(c->rep_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    }
    // scratch of the precompute
    ALLOC(c->tA, (size_t)rmax * tcols * es);
    ALLOC(c->tB, (size_t)rmax * tcols * es);
    ALLOC(c->tF, (size_t)rmax * 2 * D * 4);
    c->have_buffers = true;
  }
  if (c->graph_exec && c->graph_guidance != pl->guidance) ctx_drop_graph(c);
  // the plan's lookup tables are copied so that captured kernels keep valid addresses
  const foley_plan in = *pl;
  c->plan = in;
#define COPYTAB(field, bytes) \
  HIPTRY(hipMemcpyAsync((void*)c->field, in.field, (bytes), hipMemcpyDeviceToDevice, st)); \
  c->plan.field = c->field
  COPYTAB(rope_cos, (size_t)in.rope_len * 64 * 4);
  COPYTAB(rope_sin, (size_t)in.rope_len * 64 * 4);
  COPYTAB(solver_coef, (size_t)NI * 8 * 4);
  COPYTAB(pos_audio_self, (size_t)La * 4);
  COPYTAB(pos_visual_self, (size_t)Lv * 4);
  COPYTAB(pos_linear, (size_t)Lmax * 4);
  COPYTAB(sync_gather, (size_t)La * 4);
#undef COPYTAB
  pl = &c->plan;
  {
    const int* pt[3] = {pl->pos_audio_self, pl->pos_visual_self, pl->pos_linear};
    const int rl[3] = {La, Lv, Lmax};
    for (int k = 0; k < 3; ++k) {
      TRY(launch_gather_rows(pl->rope_cos, pt[k], rl[k], 1, pl->rope_len, 64, c->rot_cos[k], st));
      TRY(launch_gather_rows(pl->rope_sin, pt[k], rl[k], 1, pl->rope_len, 64, c->rot_sin[k], st));
    }
  }
  HIPTRY(hipMemsetAsync(c->step_ctr, 0, 256, st));
  void* tA = c->tA;
  void* tB = c->tB;
  float* tF = c->tF;

  // 1. time embedding table: t_feat -> Linear -> SiLU -> Linear   (embed_layers.py:104-136)
  Lin time0, time2;
  TRY(get_lin(c, "time0", T, D, f.time_freq_dim, true, &time0));
  TRY(get_lin(c, "time2", T, D, D, true, &time2));
  TRY(launch_cast(pl->t_feat, FOLEY_F32, tA, T, (long)NI * f.time_freq_dim, st));
  TRY(launch_gemm(gemm_plain(tA, NI, time0, tB, D), T, EPI_SILU_T, 0, st));
  TRY(launch_gemm(gemm_plain(tB, NI, time2, c->vec_table, D), T, EPI_STORE_F32, 0, st));

  // 2. AdaLN tables of the triple blocks: Linear(SiLU(vec)) for every iteration (modulate_layers.py:15-16)
  TRY(launch_rows_add_act(c->vec_table, rb_none(), NI, D, 1, tA, T, st));
  for (int b = 0; b < f.depth_triple; ++b)
    for (int s = 0; s < 2; ++s) {
      Lin m;
      TRY(get_lin(c, "t" + std::to_string(b) + (s ? ".v_mod" : ".a_mod"), T, 9 * D, D, true, &m));
      float* dst = c->modtab + ((size_t)(b * 2 + s) * NI) * 9 * D;
      TRY(launch_gemm(gemm_plain(tA, NI, m, dst, 9 * D), T, EPI_STORE_F32, 0, st));
    }

  // 3. text: cond_in, then per block text_cross_kv -> k RMSNorm + RoPE (hifi_foley.py:289-308, 765)
  {
    Lin c1, c2;
    TRY(get_lin(c, "cond1", T, D, f.cond_dim, true, &c1));
    TRY(get_lin(c, "cond2", T, D, D, true, &c2));
    TRY(launch_cast(pl->text, FOLEY_F32, tA, T, (long)ncfg * Lt * f.cond_dim, st));
    TRY(launch_gemm(gemm_plain(tA, ncfg * Lt, c1, tB, D), T, EPI_SILU_T, 0, st));
    TRY(launch_gemm(gemm_plain(tB, ncfg * Lt, c2, tA, D), T, EPI_STORE_T, 0, st));  // tA = cond embedding
    for (int b = 0; b < f.depth_triple; ++b) {
      Lin kv;
      const void* kn;
      TRY(get_lin(c, "t" + std::to_string(b) + ".t_kv", T, 2 * D, D, true, &kv));
      TRY(get_tensor(c, "t" + std::to_string(b) + ".t_kn", FOLEY_F32, {128}, &kn));
      TRY(launch_gemm(gemm_plain(tA, ncfg * Lt, kv, tF, 2 * D), T, EPI_STORE_F32, 0, st));
      QkvSplitArgs q{};
      q.qkv = tF; q.M = ncfg * Lt; q.L = Lt; q.H = H; q.nK = 2;
      q.gain[0] = (const float*)kn; q.pos[0] = pl->pos_linear;
      const int Ltp = (Lt + 31) & ~31;
      q.dst[0] = (char*)c->txt_k + (size_t)b * ncfg * H * Lt * 128 * es;
      q.dst[1] = (char*)c->txt_v + (size_t)b * ncfg * H * (foley_is_half(T) ? Ltp : Lt) * 128 * es;
      q.out_dtype = T; q.vt_pitch = foley_is_half(T) ? Ltp : 0;
      q.S_tot = Lt; q.tok_off = 0; q.eps = 1e-6f; q.cos_tab = pl->rope_cos; q.sin_tab = pl->rope_sin;
      TRY(launch_qkv_split(q, st));
    }
  }

  // 4. visual stream input: SwiGLU projection (activation_layers.py:43-44, hifi_foley.py:770)
  {
    Lin w13, w2;
    TRY(get_lin(c, "vis.w13", T, 2 * D, f.clip_dim, false, &w13));
    TRY(get_lin(c, "vis.w2", T, D, D, false, &w2));
    TRY(launch_cast(pl->clip, FOLEY_F32, tA, T, (long)ncfg * Lv * f.clip_dim, st));
    TRY(launch_gemm(gemm_plain(tA, ncfg * Lv, w13, tB, D), T, EPI_SILUGATE_T, 0, st));
    TRY(launch_gemm(gemm_plain(tB, ncfg * Lv, w2, c->v_cond0, D), T, EPI_STORE_F32, 0, st));
  }

  // 5. sync features: + pos emb, Linear, SiLU, ConvMLP(k=1), nearest-exact up-sampling (hifi_foley.py:755-762)
  {
    Lin s0, w13, w2;
    const void* pos;
    TRY(get_lin(c, "sync0", T, D, f.sync_dim, true, &s0));
    TRY(get_lin(c, "sync.w13", T, 2 * f.sync_hidden, D, false, &w13));
    TRY(get_lin(c, "sync.w2", T, D, f.sync_hidden, false, &w2));
    TRY(get_tensor(c, "sync_pos", FOLEY_F32, {8, f.sync_dim}, &pos));
    TRY(launch_add_periodic(pl->sync, (const float*)pos, ncfg * Ls, f.sync_dim, 8, tA, T, st));
    TRY(launch_gemm(gemm_plain(tA, ncfg * Ls, s0, tB, D), T, EPI_SILU_T, 0, st));
    TRY(launch_gemm(gemm_plain(tB, ncfg * Ls, w13, tA, f.sync_hidden), T, EPI_SILUGATE_T, 0, st));
    // The up-sampling to the audio frame rate is not materialised: consumers address the Ls token rows
    // through RowBcast mode 2 (common.h), so everything derived from the tokens alone - SiLU(token + vec)
    // and the single-stream blocks' modulation GEMM - runs on ncfg*Ls rows instead of ncfg*La.
    TRY(launch_gemm(gemm_plain(tA, ncfg * Ls, w2, c->sync_tok, D), T, EPI_STORE_F32, 0, st));
    // Empty sync features (text-to-audio; the unconditional half of a CFG pair) are one learned row plus
    // sync_pos_emb, which repeats every 8 tokens: the token rows are then 8-periodic and the per-token work of the
    // single-stream blocks only has 8 distinct rows per half.  Detected on the data (bit patterns), not assumed.
    // One flag per cfg half: under CFG a video clip's unconditional half carries the empty features (periodic) next to the dense
    // conditional half - the modulation GEMM then runs on 8 + Ls rows instead of 2 Ls (round 5).
    if (ncfg > 32) return FAIL(FOLEY_ERR_INVALID, "more than 32 cfg halves");
    HIPTRY(hipMemsetAsync(c->flag, 0, 4 * 32, st));
    TRY(launch_rows_periodic_check(c->sync_tok, ncfg, Ls, 8, D, c->flag, st));
  }
  HIPTRY(hipStreamSynchronize(st));
  {
    int differs[32];
    HIPTRY(hipMemcpy(differs, c->flag, 4 * 32, hipMemcpyDeviceToHost));
    static const bool mixed_on = []() { const char* e = getenv("FOLEY_SYNC_MIXED"); return !(e && e[0] == '0'); }();
    int lead = 0;
    while (Ls > 8 && lead < ncfg && !differs[lead]) ++lead;
    if (!mixed_on && lead < ncfg) lead = 0;                      // A/B switch: all halves periodic or none (rounds 2-4)
    const int per = lead > 0 ? 8 : 0;
    if (c->graph_exec && (per != c->sync_per || lead != c->sync_lead)) ctx_drop_graph(c);   // the captured modulation GEMM has another M
    c->sync_per = per;
    c->sync_lead = lead;
  }
  if (!c->fw.ok) TRY(resolve_forward_weights(c));
  {
    // 6. The single-stream blocks' modulation, Linear(SiLU(add_sync + vec)) (hifi_foley.py:366, 866-867), depends on the loop
    // iteration only (vec) and on the sync tokens - not on the latents.  Like the two-stream blocks' AdaLN tables it is therefore
    // computed HERE for all n_iter iterations in one GEMM ([n_iter*ncfg*P, D] x [n_single*6D, D]^T; the 1.02 GB weight panel is
    // streamed once per run instead of once per iteration: 0.21 ms x 50 -> ~1 ms at 5 s text-to-audio) when the table fits
    // FOLEY_SMOD_TABLE_GB (default 24; 1.06 GB for the 16 distinct rows of text-to-audio, 14.9 GB for the 224 rows of a 5 s
    // video clip); longer clips keep the per-iteration GEMM of run_forward.
    static const double cap_gb = []() { const char* e = getenv("FOLEY_SMOD_TABLE_GB"); return e ? atof(e) : 24.0; }();
    const int P = (c->sync_per && c->sync_lead == ncfg) ? c->sync_per : Ls;   // hoisting serves the all-periodic case
    const size_t ncol = (size_t)f.depth_single * 6 * D;
    const size_t tab_bytes = (size_t)NI * ncfg * P * ncol * 4;
    // ... and only where the weight stream is what the per-iteration GEMM costs (a few distinct rows: the 8-periodic empty sync
    // features).  With the 224 dense rows of a video clip the batched GEMM costs what the 50 small ones do (18.3 vs 19 ms).
    bool hoist = f.depth_single > 0 && ncfg * P <= 64 && (double)tab_bytes <= cap_gb * 1073741824.0;
    const size_t svec_bytes = (size_t)NI * ncfg * Ls * D * es;
    if (hoist) {
      // ... and only while the tables (they scale with n_iter: 1.06 GB at 50 steps, 4.2 GB at 200) take at most half of what the
      // device has free, counting what this context already holds - every data-parallel replica keeps its own
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { hipGetLastError(); free_b = 0; }
      const size_t held = c->smod_tab.bytes + c->svec_tab.bytes;
      if (tab_bytes + svec_bytes > held && (tab_bytes + svec_bytes - held) > free_b / 2) hoist = false;
    }
    const void* old_tab = c->smod_tab.p;
    if (hoist && (grow(c->smod_tab, tab_bytes) != 0 || grow(c->svec_tab, svec_bytes) != 0)) {
      hipGetLastError();     // allocation failed: not an error - the per-iteration GEMM of run_forward is still there
      hoist = false;
    }
    if (!hoist && (c->smod_tab.p || c->svec_tab.p)) {   // a plan that does not hoist gives the tables back
      HIPTRY(hipStreamSynchronize(st));
      if (c->smod_tab.p) hipFree(c->smod_tab.p);
      if (c->svec_tab.p) hipFree(c->svec_tab.p);
      c->smod_tab = DevBuf{};
      c->svec_tab = DevBuf{};
    }
    if (c->graph_exec && hoist != c->smod_hoisted) ctx_drop_graph(c);
    c->smod_hoisted = false;
    if (hoist) {
      if (c->graph_exec && old_tab != c->smod_tab.p) ctx_drop_graph(c);   // captured kernels hold the table's address
      for (int it = 0; it < NI; ++it)
        TRY(launch_rows_add_act(c->sync_tok, rb_vec(c->vec_table + (size_t)it * D, 0, nullptr), ncfg * Ls, D, 1,
                                (char*)c->svec_tab.p + (size_t)it * ncfg * Ls * D * es, T, st));
      GemmArgs gm = gemm_plain(c->svec_tab.p, NI * ncfg * P, c->fw.smod, c->smod_tab.p, (long)ncol);
      gm.segV = P; gm.segS = Ls;     // virtual rows: row r of the product is token r % P of (iteration, half) r / P
      TRY(launch_gemm(gm, T, EPI_STORE_F32, 0, st));
      c->smod_hoisted = true;
    }
  }
  {
    // the plan's table must be the nearest-exact map the kernels compute in their addressing
    std::vector<int> tab((size_t)La);
    HIPTRY(hipMemcpy(tab.data(), pl->sync_gather, (size_t)La * 4, hipMemcpyDeviceToHost));
    const float scale = (float)Ls / (float)La;
    for (int l = 0; l < La; ++l)
      if (tab[l] != rb_nearest_exact(l, scale, Ls)) return FAIL(FOLEY_ERR_INVALID, "plan.sync_gather is not the nearest-exact up-sampling table");
  }
  c->prepared = true;
  return 0;
}

// --------------------------------------------------------------------------- DiT forward
static RowBcast rb_vec(const float* base, long step_stride, const int* step_ptr) {
  return RowBcast{base, 0, 0, 1, 1, step_ptr, step_stride};
}
static RowBcast rb_tok(const float* base, long ld, int rows_per_cfg, int L) {
  return RowBcast{base, ld, 1, rows_per_cfg, L, nullptr, 0, 0, 0.f, 0};
}
// operand with Ls rows per cfg, read by audio frame l at its nearest-exact source row (common.h RowBcast mode 2)
// per > 0: the first `lead` cfg halves store `per` rows each (8-periodic token rows), the others all Ls rows behind them;
// lead < 0: every half periodic
static RowBcast rb_up(const float* base, long ld, int rows_per_cfg, int L, int Ls, int per = 0, int lead = -1) {
  RowBcast r{base, ld, 2, rows_per_cfg, L, nullptr, 0, Ls, (float)Ls / (float)L, per, 0, 0};
  if (per > 0) {
    r.dense_from = lead < 0 ? INT_MAX : lead;
    r.dense_base = lead < 0 ? 0 : lead * per;
  }
  return r;
}

// Per-kernel profile (foley_profile_forward): the op's kernel launch carries two events as its own
// start / stop timestamps (common.h FOLEY_LAUNCH), tagged with the op's algorithmic FLOPs / bytes.
static int prof_begin(foley_ctx* c, hipStream_t st, const char* label, double flop, double bytes) {
  if (!c->prof.on) return 0;
  ProfRec r{label, flop, bytes, c->prof.get(), c->prof.get(), nullptr};
  c->prof.recs.push_back(r);
  g_foley_prof = FoleyProfHook{r.e0, r.e1, nullptr};   // consumed by the op's (first) kernel launch
  return 0;
}
static int prof_end(foley_ctx* c, hipStream_t st) {
  if (!c->prof.on) return 0;
  if (g_foley_prof.e0) {   // the op launched nothing: keep the record well-formed with a plain (empty) bracket
    g_foley_prof.e0 = nullptr;
    HIPTRY(hipEventRecord(c->prof.recs.back().e0, st));
    HIPTRY(hipEventRecord(c->prof.recs.back().e1, st));
  } else {
    c->prof.recs.back().fn = g_foley_prof.fn;
  }
  return 0;
}
#define PROF(label, flop, bytes, call)            \
  do {                                            \
    TRY(prof_begin(c, st, (label), (flop), (bytes))); \
    TRY(call);                                    \
    TRY(prof_end(c, st));                         \
  } while (0)

// The single-block modulation GEMM (510 GFLOP, depends on the iteration index only) runs IN LINE at the head
// of the forward.  Round 1 overlapped it with the two-stream blocks on a side stream; with the faster block
// kernels of round 2 its 5184 workgroups only steal CUs from them (A/B in one box: 448.5 -> 439.5 ms per
// 50-iteration loop at bs=1, 1753 -> 1741 ms at bs=8).  FOLEY_SMOD_SIDE=1 restores the side stream.
static bool smod_inline() {
  static const bool v = []() { const char* e = getenv("FOLEY_SMOD_SIDE"); return !(e && e[0] == '1'); }();
  return v;
}

static int run_forward(foley_ctx* c, hipStream_t st) {
  const foley_config& f = c->cfg;
  const foley_plan& pl = c->plan;
  const ForwardW& W = c->fw;
  if (!W.ok) return FAIL(FOLEY_ERR_STATE, "forward weights are not resolved (foley_prepare)");
  // one clip per CFG half: the small-grid GEMMs may rotate their K origin per M tile (gemm.hip: g_gemm_krot_ok) - with several clips
  // in the batch, clips with equal noise must stay bit-identical, so their rows keep one summation order
  struct KRotScope {
    int prev;
    explicit KRotScope(int v) : prev(g_gemm_krot_ok) { g_gemm_krot_ok = v; }
    ~KRotScope() { g_gemm_krot_ok = prev; }
  } krot_scope(pl.clips == 1 ? 1 : 0);
  const int D = f.hidden, H = f.heads, C = f.latent_dim, T = f.compute_dtype;
  const int ncfg = pl.ncfg, clips = pl.clips, La = pl.La, Lv = pl.Lv, Ls = pl.Ls, Lt = pl.Lt, NI = pl.n_iter;
  const int Bc = ncfg * clips, M = Bc * La, Mv = Bc * Lv, S = La + Lv;
  const bool bf = foley_is_half(T);   // 16-bit throughput mode (bf16 or fp16 operands): transposed V, fused head split
  const size_t es = esize(T);
  const int Sp = (S + 31) & ~31, Lap = (La + 31) & ~31;  // V^T row pitches (bf16 attention)
  const int* sp = c->step_ctr;
  // algorithmic work of one launch (profile labels): dense contraction FLOPs, operand + result bytes
  auto gf = [](double m, double n, double k) { return 2.0 * m * n * k; };
  auto gb = [&](double m, double n, double k, double out_es) { return (m * k + n * k) * (double)es + m * n * out_es; };
  auto af = [&](double b, double sq, double skv) { return 4.0 * b * H * sq * skv * 128.0; };
  auto ab = [&](double b, double sq, double skv) { return (2.0 * b * H * sq * 128.0 + 2.0 * b * H * skv * 128.0) * (double)es; };

  // ---- per-token conditioning of the single-stream blocks, SiLU(add_sync + vec) (hifi_foley.py:866-867),
  // and every single block's modulation GEMM (hifi_foley.py:366).  They depend on the iteration only, are
  // identical for every clip of a CFG half, and - add_sync being an up-sampling of the Ls sync tokens - have
  // only Ls distinct rows per half: M = ncfg*Ls (224 instead of 500 at 5 s).  In line by default (smod_inline()).
  if (!c->smod_hoisted) {
    const bool inl = c->prof.on || smod_inline();
    hipStream_t sd = inl ? st : c->side;
    if (!inl) {
      HIPTRY(hipEventRecord(c->ev_fork, st));
      HIPTRY(hipStreamWaitEvent(sd, c->ev_fork, 0));
    }
    // distinct rows only: 8 per 8-periodic half (the leading sync_lead halves), Ls per dense half - packed back to back, which
    // is the row order of the modulation table (RowBcast::dense_from / dense_base)
    const int lead = c->sync_per ? c->sync_lead : 0, R = lead * c->sync_per + (ncfg - lead) * Ls;
    for (int h = 0; h < lead; ++h)
      TRY(launch_rows_add_act(c->sync_tok + (size_t)h * Ls * D, rb_vec(c->vec_table, D, sp), c->sync_per, D, 1,
                              (char*)c->svec + (size_t)h * c->sync_per * D * es, T, sd));
    if (lead < ncfg)
      TRY(launch_rows_add_act(c->sync_tok + (size_t)lead * Ls * D, rb_vec(c->vec_table, D, sp), (ncfg - lead) * Ls, D, 1,
                              (char*)c->svec + (size_t)lead * c->sync_per * D * es, T, sd));
    if (f.depth_single > 0) {
      // one GEMM for all blocks: [R, D] x [n_single*6D, D]^T -> smod [R, n_single*6D]
      const double n = (double)f.depth_single * 6 * D;
      GemmArgs gm = gemm_plain(c->svec, R, W.smod, c->smod, (long)f.depth_single * 6 * D);
      TRY(prof_begin(c, st, "single.modulation (all blocks, one GEMM)", gf(R, n, D), gb(R, n, D, 4)));
      TRY(launch_gemm(gm, T, EPI_STORE_F32, 0, sd));
      TRY(prof_end(c, st));
    }
    // the join point always exists (also with depth_single == 0): a forked capture stream must be
    // joined before the capture ends, and eager callers must not race on svec
    if (!inl) HIPTRY(hipEventRecord(c->ev_mod[0], sd));
  }

  // audio_embedder (conv k=1 == linear over the transposed latents) + add_sync (hifi_foley.py:768, 838-839)
  {
    GemmArgs g = gemm_plain(c->xin, M, W.audio_in, c->audio, D);
    g.rb = rb_up(c->sync_tok, D, clips * La, La, Ls);
    PROF("audio_embedder", gf(M, D, C), gb(M, D, C, 4), launch_gemm(g, T, EPI_STORE_F32, 0, st));
  }
  // visual stream starts from the step-invariant projection, replicated per clip (one gather launch)
  TRY(launch_gather_rows(c->v_cond0, c->rep_idx, clips * Lv, ncfg, Lv, D, c->vcond, st));

  // residual updates left pending by deferred split-K GEMMs, per stream (audio, visual); the next
  // LayerNorm of that stream applies them
  LnPending pend[2] = {LnPending{}, LnPending{}};
  // deferred split-K slabs in the operand type (bf16 / fp16 compute): half the bytes the GEMM epilogues write and the next
  // LayerNorm reads (that kernel runs at the fabric's bandwidth: 22 MB in 3.4 us at M = 500).  FOLEY_SLAB16=0 keeps fp32.
  static const bool slab16_on = []() { const char* e = getenv("FOLEY_SLAB16"); return !(e && e[0] == '0'); }();
  const int slab_half = (bf && slab16_on) ? T : 0;
  auto with_partials = [&](GemmArgs& g, float* slabs) {
    g.partial_half = slab_half ? 1 : 0;
    g.partials = slabs;
    g.partial_stride = (long)g.M * g.N;
    g.partial_cap = PART_CAP;
  };
  const double ln_bytes_t = (double)(M + Mv) * D * (4 + 4 + es), ln_bytes_s = (double)M * D * (4 + 4 + es);
  for (int blk = 0; blk < f.depth_triple; ++blk) {
    const TripleW& w = W.t[blk];
    auto tb = [&](int s, int chunk) {
      return rb_vec(c->modtab + ((size_t)(blk * 2 + s) * NI) * 9 * D + (size_t)chunk * D, 9L * D, sp);
    };
    struct Stream { float* x; void* xn; float* qkv; void* att; void* hid; int rows, L, tok_off; const int* pos; };
    Stream ss[2] = {{c->audio, c->xn_a, c->qkv_a, c->att_a, c->hid_a, M, La, Lv, pl.pos_audio_self},
                    {c->vcond, c->xn_v, c->qkv_v, c->att_v, c->hid_v, Mv, Lv, 0, pl.pos_visual_self}};
    // Both streams go through the same sequence of ops with their own weights; each op is ONE
    // launch covering the audio problem and the (much smaller) visual problem.
    auto ln2 = [&](int c_shift, int c_scale) -> int {
      LnArgs a0{ss[0].x, ss[0].rows, tb(0, c_shift), tb(0, c_scale), ss[0].xn, pend[0]};
      LnArgs a1{ss[1].x, ss[1].rows, tb(1, c_shift), tb(1, c_scale), ss[1].xn, pend[1]};
      pend[0] = pend[1] = LnPending{};
      PROF("triple.layernorm+modulate (+pending split-K sum)", 0.0, ln_bytes_t, launch_ln_mod_pair(a0, a1, D, 1e-6f, T, st));
      return 0;
    };
    auto gated2 = [&](const char* label, const Lin& la, const Lin& lv, bool from_hid, int c_gate) -> int {
      GemmArgs g0 = gemm_plain(from_hid ? ss[0].hid : ss[0].att, ss[0].rows, la, ss[0].x, D);
      GemmArgs g1 = gemm_plain(from_hid ? ss[1].hid : ss[1].att, ss[1].rows, lv, ss[1].x, D);
      g0.rb = tb(0, c_gate);
      g1.rb = tb(1, c_gate);
      with_partials(g0, c->part_a);
      with_partials(g1, c->part_v);
      int ks = 1;
      PROF(label, gf(M + Mv, la.N, la.K), gb(M + Mv, la.N, la.K, 4) + (double)la.N * la.K * es,
           launch_gemm_pair(g0, g1, T, EPI_GATE_RES, st, &ks));
      if (ks > 1) {
        pend[0] = LnPending{c->part_a, ks, g0.partial_stride, la.b, g0.rb, slab_half};
        pend[1] = LnPending{c->part_v, ks, g1.partial_stride, lv.b, g1.rb, slab_half};
      }
      return 0;
    };
    auto split_args = [&](int s, int nK, const void* gq, const void* gk, const int* pos) {
      Stream& z = ss[s];
      QkvSplitArgs q{};
      q.qkv = z.qkv; q.M = z.rows; q.L = z.L; q.H = H; q.nK = nK;
      q.gain[0] = (const float*)gq; q.gain[1] = (const float*)gk;
      q.pos[0] = pos; q.pos[1] = nK > 1 ? pos : nullptr;
      q.dst[0] = c->Q; q.dst[1] = c->K; q.dst[2] = c->V;
      q.out_dtype = T; q.vt_pitch = (bf && nK == 3) ? Sp : 0;
      q.S_tot = S; q.tok_off = z.tok_off; q.eps = 1e-6f; q.cos_tab = pl.rope_cos; q.sin_tab = pl.rope_sin;
      const int k = pos == pl.pos_audio_self ? 0 : (pos == pl.pos_visual_self ? 1 : (pos == pl.pos_linear ? 2 : -1));
      for (int i = 0; i < 2 && k >= 0; ++i)
        if (q.pos[i]) { q.rcos[i] = c->rot_cos[k]; q.rsin[i] = c->rot_sin[k]; }
      return q;
    };
    // 1. joint self attention (hifi_foley.py:215-269)
    {
      TRY(ln2(0, 1));
      GemmArgs g0 = gemm_plain(ss[0].xn, ss[0].rows, w.qkv[0], ss[0].qkv, 3 * D);
      GemmArgs g1 = gemm_plain(ss[1].xn, ss[1].rows, w.qkv[1], ss[1].qkv, 3 * D);
      g0.qs = split_args(0, 3, w.qn[0], w.kn[0], ss[0].pos);
      g1.qs = split_args(1, 3, w.qn[1], w.kn[1], ss[1].pos);
      PROF("triple.qkv GEMM + RMSNorm/RoPE head split", gf(M + Mv, 3 * D, D), gb(M + Mv, 3 * D, D, es) + 3.0 * D * D * es,
           launch_gemm_pair(g0, g1, T, EPI_QKV_SPLIT, st));   // head split fused into the projection
      AttnArgs a{c->Q, c->K, c->V, Bc, H, S, S, 1, c->att_v, c->att_a, Lv, T, bf ? Sp : 0};
      PROF("triple.self attention", af(Bc, S, S), ab(Bc, S, S), launch_attention(a, T, st));
      TRY(gated2("triple.self proj GEMM (gated residual)", w.proj[0], w.proj[1], false, 2));
    }
    // 2. cross attention to the (cached) text keys/values (hifi_foley.py:271-319)
    {
      TRY(ln2(3, 4));
      GemmArgs g0 = gemm_plain(ss[0].xn, ss[0].rows, w.cq[0], ss[0].qkv, D);
      GemmArgs g1 = gemm_plain(ss[1].xn, ss[1].rows, w.cq[1], ss[1].qkv, D);
      g0.qs = split_args(0, 1, w.cqn[0], nullptr, pl.pos_linear);
      g1.qs = split_args(1, 1, w.cqn[1], nullptr, pl.pos_linear);
      const int Ltp = (Lt + 31) & ~31;
      const size_t offk = (size_t)blk * ncfg * H * Lt * 128 * es;
      const size_t offv = (size_t)blk * ncfg * H * (bf ? Ltp : Lt) * 128 * es;
      // 16-bit modes: the projection may run the attention against the <= 96 cached text keys in its epilogue (small grids:
      // gemm_impl.h decides and reports through attn_fused); the q tensor and the attention launch are then gone
      int fused = 0;
      if (bf && Lt <= 96 && Ltp >= 96) {
        for (int s = 0; s < 2; ++s) {
          QkvSplitArgs& q = s ? g1.qs : g0.qs;
          q.attn_k = (char*)c->txt_k + offk; q.attn_vt = (char*)c->txt_v + offv; q.attn_out = ss[s].att;
          q.attn_skv = Lt; q.attn_pitch = Ltp; q.attn_bdiv = clips; q.attn_fused = &fused;
        }
      }
      PROF("triple.cross q GEMM + head split (+ cross attention on small grids)", gf(M + Mv, D, D),
           gb(M + Mv, D, D, es) + 1.0 * D * D * es, launch_gemm_pair(g0, g1, T, EPI_QKV_SPLIT, st));
      if (!fused) {
        AttnArgs a{c->Q, (char*)c->txt_k + offk, (char*)c->txt_v + offv, Bc, H, S, Lt, clips, c->att_v, c->att_a, Lv,
                   T, bf ? Ltp : 0};
        PROF("triple.cross attention", af(Bc, S, Lt), ab(Bc, S, Lt), launch_attention(a, T, st));
      }
      TRY(gated2("triple.cross proj GEMM (gated residual)", w.cproj[0], w.cproj[1], false, 5));
    }
    // 3. GELU-tanh MLPs (hifi_foley.py:321-331)
    {
      TRY(ln2(6, 7));
      PROF("triple.mlp fc1 GEMM + GELU", gf(M + Mv, f.mlp_hidden, D), gb(M + Mv, f.mlp_hidden, D, es) + (double)f.mlp_hidden * D * es,
           launch_gemm_pair(gemm_plain(ss[0].xn, ss[0].rows, w.fc1[0], ss[0].hid, f.mlp_hidden),
                            gemm_plain(ss[1].xn, ss[1].rows, w.fc1[1], ss[1].hid, f.mlp_hidden), T, EPI_GELU_T, st));
      TRY(gated2("triple.mlp fc2 GEMM (gated residual)", w.fc2[0], w.fc2[1], true, 8));
    }
  }

  // join: the single blocks' modulation table is ready (profiling ran it in line)
  if (!c->smod_hoisted && !c->prof.on && !smod_inline()) HIPTRY(hipStreamWaitEvent(st, c->ev_mod[0], 0));
  const int Hc = f.conv_hidden;
  for (int blk = 0; blk < f.depth_single; ++blk) {
    const SingleW& w = W.s[blk];
    const float* smod_b = (c->smod_hoisted ? (const float*)c->smod_tab.p : c->smod) + (size_t)blk * 6 * D;   // column block of the fused table
    // table layout per iteration: every half periodic (8 rows each) | the leading halves periodic, the others dense (the
    // per-iteration GEMM only - a hoisted table that is not all-periodic is dense) | every half dense
    const bool allper = c->sync_per && c->sync_lead == ncfg;
    const int per_eff = (allper || !c->smod_hoisted) ? c->sync_per : 0;
    auto sm = [&](int chunk) {
      RowBcast r = rb_up(smod_b + (size_t)chunk * D, 6L * D * f.depth_single, clips * La, La, Ls, per_eff, allper ? -1 : c->sync_lead);
      if (c->smod_hoisted) {   // the table of every iteration: this iteration's rows start at step * ncfg*P*ld
        r.step_ptr = sp;
        r.step_stride = (long)ncfg * (allper ? c->sync_per : Ls) * 6L * D * f.depth_single;
      }
      return r;
    };
    PROF("single.layernorm+modulate (+pending split-K sum)", 0.0, ln_bytes_s,
         launch_ln_mod_pending(c->audio, M, D, 1e-5f, sm(0), sm(1), c->xn_a, T, pend[0], st));
    pend[0] = LnPending{};
    GemmArgs gq = gemm_plain(c->xn_a, M, w.qkv, c->qkv_a, 3 * D);
    QkvSplitArgs q{};
    q.qkv = c->qkv_a; q.M = M; q.L = La; q.H = H; q.nK = 3;
    q.gain[0] = w.qn; q.gain[1] = w.kn;
    q.pos[0] = pl.pos_linear; q.pos[1] = pl.pos_linear;
    q.dst[0] = c->Q; q.dst[1] = c->K; q.dst[2] = c->V;
    q.out_dtype = T; q.vt_pitch = bf ? Lap : 0;
    q.S_tot = La; q.tok_off = 0; q.eps = 1.1920928955078125e-07f;  // nn.RMSNorm(eps=None) -> finfo(fp32).eps
    q.cos_tab = pl.rope_cos; q.sin_tab = pl.rope_sin;
    q.rcos[0] = q.rcos[1] = c->rot_cos[2]; q.rsin[0] = q.rsin[1] = c->rot_sin[2];
    gq.qs = q;
    PROF("single.qkv GEMM + RMSNorm/RoPE head split", gf(M, 3 * D, D), gb(M, 3 * D, D, es), launch_gemm(gq, T, EPI_QKV_SPLIT, 0, st));
    {
      AttnArgs a{c->Q, c->K, c->V, Bc, H, La, La, 1, c->att_a, c->att_a, 0, T, bf ? Lap : 0};
      PROF("single.self attention", af(Bc, La, La), ab(Bc, La, La), launch_attention(a, T, st));
    }
    {
      GemmArgs g = gemm_conv(c->att_a, M, La, D, 3, 1, w.lin1, c->audio, D);
      g.rb = sm(2);
      with_partials(g, c->part_a);
      int ks = 1;
      PROF("single.linear1 conv3 GEMM (gated residual)", gf(M, D, 3 * D), gb(M, D, 3 * D, 4), launch_gemm(g, T, EPI_GATE_RES, 0, st, &ks));
      if (ks > 1) pend[0] = LnPending{c->part_a, ks, g.partial_stride, w.lin1.b, g.rb, slab_half};
    }
    PROF("single.layernorm+modulate (+pending split-K sum)", 0.0, ln_bytes_s,
         launch_ln_mod_pending(c->audio, M, D, 1e-5f, sm(3), sm(4), c->xn_a, T, pend[0], st));
    pend[0] = LnPending{};
    PROF("single.w1/w3 conv3 GEMM + SiLU gate", gf(M, 2 * Hc, 3 * D), gb(M, 2 * Hc, 3 * D, es) - (double)M * Hc * es,
         launch_gemm(gemm_conv(c->xn_a, M, La, D, 3, 1, w.w13, c->hid_a, Hc), T, EPI_SILUGATE_T, 0, st));
    {
      GemmArgs g = gemm_conv(c->hid_a, M, La, Hc, 3, 1, w.w2, c->audio, D);
      g.rb = sm(5);
      with_partials(g, c->part_a);
      int ks = 1;
      PROF("single.w2 conv3 GEMM (gated residual)", gf(M, D, 3 * Hc), gb(M, D, 3 * Hc, 4), launch_gemm(g, T, EPI_GATE_RES, 0, st, &ks));
      if (ks > 1) pend[0] = LnPending{c->part_a, ks, g.partial_stride, w.w2.b, g.rb, slab_half};
    }
  }

  // FinalLayer1D: adaLN is a no-op with 3-D conditioning => linear(LayerNorm(x))
  {
    PROF("final.layernorm", 0.0, ln_bytes_s,
         launch_ln_mod_pending(c->audio, M, D, 1e-6f, rb_none(), rb_none(), c->xn_a, T, pend[0], st));
    PROF("final.linear", gf(M, C, D), gb(M, C, D, 4), launch_gemm(gemm_plain(c->xn_a, M, W.fin, c->pred, C), T, EPI_STORE_F32, 0, st));
  }
  return 0;
}

extern "C" int foley_dit_forward(foley_ctx* c, const float* latents, int iter, float* out_rows, void* stream_v) {
  if (!c || !latents || !out_rows) return FAIL(FOLEY_ERR_INVALID, "null argument");
  if (!c->prepared) return FAIL(FOLEY_ERR_STATE, "foley_prepare has not been called");
  if (iter < 0 || iter >= c->plan.n_iter) return FAIL(FOLEY_ERR_INVALID, "iteration out of range");
  hipStream_t st = (hipStream_t)stream_v;
  HIPTRY(hipSetDevice(c->device));
  const foley_plan& pl = c->plan;
  const int C = c->cfg.latent_dim;
  HIPTRY(hipMemcpyAsync(c->step_ctr, &iter, sizeof(int), hipMemcpyHostToDevice, st));
  HIPTRY(hipStreamSynchronize(st));  // `iter` lives on the caller's stack
  TRY(launch_latent_rows(latents, pl.clips, C, pl.La, pl.ncfg, c->xin, c->cfg.compute_dtype, st));
  TRY(run_forward(c, st));
  HIPTRY(hipMemcpyAsync(out_rows, c->pred, (size_t)pl.ncfg * pl.clips * pl.La * C * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

// --------------------------------------------------------------------------- per-kernel profile
// `repeats` eager forwards at loop iteration `iter`; every op's kernel is launched with start / stop
// events attached to the dispatch itself (hipExtLaunchKernelGGL on the launch stream), aggregated by
// op label - total_ms is kernel time proper, the quantity rocprofv3's kernel trace reports.
// bracket_ms = mean elapsed time of an EMPTY event bracket (two back-to-back hipEventRecord), reported
// for reference only: it is NOT contained in total_ms.
extern "C" int foley_profile_forward(foley_ctx* c, const float* latents, int iter, int repeats, foley_prof_entry* out,
                                     int cap, int* n_out, float* bracket_ms, void* stream_v) {
  if (!c || !latents || !out || !n_out || cap < 1 || repeats < 1) return FAIL(FOLEY_ERR_INVALID, "bad argument");
  if (!c->prepared) return FAIL(FOLEY_ERR_STATE, "foley_prepare has not been called");
  if (iter < 0 || iter >= c->plan.n_iter) return FAIL(FOLEY_ERR_INVALID, "iteration out of range");
  hipStream_t st = (hipStream_t)stream_v;
  HIPTRY(hipSetDevice(c->device));
  const foley_plan& pl = c->plan;
  HIPTRY(hipMemcpyAsync(c->step_ctr, &iter, sizeof(int), hipMemcpyHostToDevice, st));
  HIPTRY(hipStreamSynchronize(st));
  TRY(launch_latent_rows(latents, pl.clips, c->cfg.latent_dim, pl.La, pl.ncfg, c->xin, c->cfg.compute_dtype, st));
  TRY(run_forward(c, st));   // warm: code objects loaded, caches in their steady state
  c->prof.recs.clear();
  c->prof.used = 0;
  c->prof.on = true;
  int rc = 0;
  for (int r = 0; r < repeats && rc == 0; ++r) rc = run_forward(c, st);
  c->prof.on = false;
  g_foley_prof = FoleyProfHook{nullptr, nullptr, nullptr};   // an op that failed between prof_begin and its launch leaves the hook armed
  if (rc) return rc;
  // empty brackets for the calibration
  constexpr int NCAL = 32;
  hipEvent_t cal[2 * NCAL];
  for (int i = 0; i < 2 * NCAL; ++i) cal[i] = c->prof.get();
  for (int i = 0; i < NCAL; ++i) {
    HIPTRY(hipEventRecord(cal[2 * i], st));
    HIPTRY(hipEventRecord(cal[2 * i + 1], st));
  }
  HIPTRY(hipStreamSynchronize(st));
  double cal_ms = 0.0;
  for (int i = 0; i < NCAL; ++i) {
    float ms = 0.f;
    HIPTRY(hipEventElapsedTime(&ms, cal[2 * i], cal[2 * i + 1]));
    cal_ms += ms;
  }
  if (bracket_ms) *bracket_ms = (float)(cal_ms / NCAL);
  int n = 0;
  for (const ProfRec& r : c->prof.recs) {
    float ms = 0.f;
    HIPTRY(hipEventElapsedTime(&ms, r.e0, r.e1));
    int j = 0;
    for (; j < n; ++j)
      if (!strcmp(out[j].label, r.label)) break;
    if (j == n) {
      if (n == cap) return FAIL(FOLEY_ERR_INVALID, "profile: entry buffer too small");
      memset(&out[n], 0, sizeof(out[n]));
      strncpy(out[n].label, r.label, sizeof(out[n].label) - 1);
      if (r.fn) {   // the kernel's symbol, demangled - the name rocprofv3's kernel trace lists it under
        const char* mangled = hipKernelNameRefByPtr(r.fn, st);
        if (mangled) {
          int status = 0;
          char* dm = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
          strncpy(out[n].kernel, status == 0 && dm ? dm : mangled, sizeof(out[n].kernel) - 1);
          free(dm);
        }
      }
      ++n;
    }
    out[j].calls += 1;
    out[j].total_ms += ms;
    out[j].flop += r.flop;
    out[j].bytes += r.bytes;
  }
  *n_out = n;
  return 0;
}

// --------------------------------------------------------------------------- sampler loop
static int run_iteration(foley_ctx* c, hipStream_t st) {
  const foley_plan& pl = c->plan;
  TRY(run_forward(c, st));
  StepArgs s{};
  s.pred = c->pred; s.x = c->x_cur; s.x_saved = c->x_saved; s.d_acc = c->d_acc;
  s.clips = pl.clips; s.C = c->cfg.latent_dim; s.L = pl.La; s.ncfg = pl.ncfg;
  s.guidance = pl.guidance; s.coef = pl.solver_coef; s.step_ptr = c->step_ctr;
  s.rows_out = c->xin; s.rows_dtype = c->cfg.compute_dtype;
  return launch_solver_step(s, st);
}

extern "C" int foley_sample(foley_ctx* c, float* latents, int use_graph, foley_progress_cb cb, void* user,
                            void* stream_v) {
  if (!c || !latents) return FAIL(FOLEY_ERR_INVALID, "null argument");
  if (!c->prepared) return FAIL(FOLEY_ERR_STATE, "foley_prepare has not been called");
  hipStream_t st = (hipStream_t)stream_v;
  HIPTRY(hipSetDevice(c->device));
  const foley_plan& pl = c->plan;
  const size_t xbytes = (size_t)pl.clips * c->cfg.latent_dim * pl.La * 4;
  if (use_graph && !c->graph_exec) {
    // Every per-iteration value is read from device memory (step counter, tables) and every
    // buffer is context-owned, so ONE captured iteration replays for the whole loop and for
    // later runs of the same shape.
    std::lock_guard<std::mutex> setup_lock(g_setup_mutex);
    hipStream_t cs;
    HIPTRY(hipStreamSynchronize(st));
    HIPTRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    int rc = 0;
    if (e == hipSuccess) {
      rc = run_iteration(c, cs);
      hipError_t e2 = hipStreamEndCapture(cs, &graph);
      if (e == hipSuccess) e = e2;
    }
    if (e == hipSuccess && rc == 0) e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipStreamDestroy(cs);
    if (rc != 0) return rc;
    if (e != hipSuccess) return FAIL(FOLEY_ERR_HIP, hipGetErrorString(e));
    c->graph_guidance = pl.guidance;
  }
  c->abort_req.store(0, std::memory_order_relaxed);   // a request left over from before this loop is not for it
  HIPTRY(hipEventRecord(c->ev0, st));
  HIPTRY(hipMemcpyAsync(c->x_cur, latents, xbytes, hipMemcpyDeviceToDevice, st));
  HIPTRY(hipMemsetAsync(c->step_ctr, 0, sizeof(int), st));
  TRY(launch_latent_rows(c->x_cur, pl.clips, c->cfg.latent_dim, pl.La, pl.ncfg, c->xin, c->cfg.compute_dtype, st));
  for (int it = 0; it < pl.n_iter; ++it) {
    if (use_graph) HIPTRY(hipGraphLaunch(c->graph_exec, st));
    else TRY(run_iteration(c, st));
    if (cb) {
      HIPTRY(hipMemcpyAsync(latents, c->x_cur, xbytes, hipMemcpyDeviceToDevice, st));
      HIPTRY(hipStreamSynchronize(st));
      cb(it + 1, pl.n_iter, user);
    }
    if (c->abort_req.exchange(0, std::memory_order_acq_rel)) {   // latents hold the state after iteration it + 1
      if (!cb) {
        HIPTRY(hipMemcpyAsync(latents, c->x_cur, xbytes, hipMemcpyDeviceToDevice, st));
        HIPTRY(hipStreamSynchronize(st));
      }
      return FAIL(FOLEY_ERR_ABORTED, "sampling loop aborted by foley_abort()");
    }
  }
  HIPTRY(hipMemcpyAsync(latents, c->x_cur, xbytes, hipMemcpyDeviceToDevice, st));
  HIPTRY(hipEventRecord(c->ev1, st));
  c->timed = true;
  return 0;
}

extern "C" int foley_abort(foley_ctx* c) {
  if (!c) return FAIL(FOLEY_ERR_INVALID, "null context");
  c->abort_req.store(1, std::memory_order_release);
  return 0;
}

// --------------------------------------------------------------------------- DAC decoder
// Activations are kept time-major [clip, T, C] so that every conv is a GEMM over contiguous
// channel vectors; weight-norm is folded at pack time; each snake is evaluated once, in the
// epilogue of the op that produces its input.  (dac.py:28-44, 98-149, 280-303)
extern "C" int foley_dac_decode(foley_ctx* c, const float* latents, int clips, int T, float* wave, void* stream_v) {
  if (!c || !latents || !wave || clips < 1 || T < 1) return FAIL(FOLEY_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream_v;
  HIPTRY(hipSetDevice(c->device));
  const foley_config& f = c->cfg;
  const int L = f.latent_dim, NR = f.dac_n_rates;
  // largest activation
  size_t maxel = (size_t)T * f.dac_dim;
  {
    long t = T;
    int ch = f.dac_dim;
    for (int i = 0; i < NR; ++i) {
      t *= f.dac_rates[i];
      ch /= 2;
      maxel = std::max(maxel, (size_t)t * ch);
    }
  }
  {
    std::lock_guard<std::mutex> setup_lock(g_setup_mutex);
    HIPTRY(hipStreamSynchronize(st));
    TRY(grow(c->dacP, maxel * clips * 4));
    TRY(grow(c->dacQ, maxel * clips * 4));
    TRY(grow(c->dacR, maxel * clips * 4));
    TRY(grow(c->dacZ, (size_t)clips * T * L * 4 * 2));
  }
  float *P = (float*)c->dacP.p, *Q = (float*)c->dacQ.p, *R = (float*)c->dacR.p;
  float* Z0 = (float*)c->dacZ.p;
  float* Z1 = Z0 + (size_t)clips * T * L;
  HIPTRY(hipEventRecord(c->ev0, st));

  TRY(launch_latent_rows(latents, clips, L, T, 1, Z0, FOLEY_F32, st));
  Lin pq, cin;
  TRY(get_lin(c, "dac.pq", FOLEY_F32, L, L, true, &pq));
  TRY(get_lin(c, "dac.in", FOLEY_F32, f.dac_dim, 7 * L, true, &cin));
  TRY(launch_gemm(gemm_plain(Z0, clips * T, pq, Z1, L), FOLEY_F32, EPI_STORE_F32, 0, st));
  const void* al;
  TRY(get_tensor(c, "dac.0.alpha0", FOLEY_F32, {f.dac_dim}, &al));
  {
    GemmArgs g = gemm_conv(Z1, clips * T, T, L, 7, 1, cin, nullptr, f.dac_dim);
    g.out1 = P; g.alpha = (const float*)al; g.alphaC = f.dac_dim;
    TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
  }
  float* S_in = P;   // snake-activated input of the next op
  float* X = Q;      // residual trunk
  float* S_alt = R;  // the other snake buffer
  int Tin = T, Cin = f.dac_dim;
  for (int i = 0; i < NR; ++i) {
    const int s = f.dac_rates[i], Cout = Cin / 2, Tout = Tin * s, pad = (s + 1) / 2;
    const std::string p = "dac." + std::to_string(i) + ".";
    Lin up;
    TRY(get_lin(c, p + "up", FOLEY_F32, s * Cout, 2 * Cin, true, &up));
    const void* a1;
    TRY(get_tensor(c, p + "0.a1", FOLEY_F32, {Cout}, &a1));
    {
      // transposed conv: virtual row q of segment (Tin+1) = [x[q-1] ; x[q]], N axis = (phase, Cout),
      // output sample t = q*s + phase - pad  (dac.py:102-109)
      GemmArgs g = gemm_plain(S_in, clips * (Tin + 1), up, X, (long)s * Cout);
      g.lda = Cin; g.segV = Tin + 1; g.segS = Tin; g.taps = 2; g.tapC = Cin; g.dil = 1; g.tap0 = -1;
      g.osegV = Tin + 1; g.out_seg = (long)Tout * Cout; g.out_row = (long)s * Cout; g.out_shift = -(long)pad * Cout;
      g.out_check = 1;
      g.out1 = S_alt; g.alpha = (const float*)a1; g.alphaC = Cout;
      TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
    }
    std::swap(S_in, S_alt);  // S_in now holds snake(x) for unit 0
    for (int j = 0; j < 3; ++j) {
      const int d = f.dac_dilations[j];
      const std::string u = p + std::to_string(j) + ".";
      Lin c7, c1;
      const void *a2, *an;
      TRY(get_lin(c, u + "c7", FOLEY_F32, Cout, 7 * Cout, true, &c7));
      TRY(get_lin(c, u + "c1", FOLEY_F32, Cout, Cout, true, &c1));
      TRY(get_tensor(c, u + "a2", FOLEY_F32, {Cout}, &a2));
      // alpha of whatever consumes this unit's output next
      std::string nxt = (j < 2) ? p + std::to_string(j + 1) + ".a1"
                                : (i + 1 < NR ? "dac." + std::to_string(i + 1) + ".alpha0" : std::string("dac.out.alpha"));
      TRY(get_tensor(c, nxt, FOLEY_F32, {Cout}, &an));
      {
        GemmArgs g = gemm_conv(S_in, clips * Tout, Tout, Cout, 7, d, c7, nullptr, Cout);
        g.out1 = S_alt; g.alpha = (const float*)a2; g.alphaC = Cout;
        TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
      }
      {
        GemmArgs g = gemm_plain(S_alt, clips * Tout, c1, X, Cout);
        g.res = X; g.out1 = S_in; g.alpha = (const float*)an; g.alphaC = Cout;
        TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
      }
    }
    Tin = Tout;
    Cin = Cout;
  }
  const void *ow, *ob;
  TRY(get_tensor(c, "dac.out.w", FOLEY_F32, {7 * Cin}, &ow));
  TRY(get_tensor(c, "dac.out.b", FOLEY_F32, {1}, &ob));
  TRY(launch_dac_out(S_in, (const float*)ow, (const float*)ob, clips, Tin, Cin, wave, st));
  HIPTRY(hipEventRecord(c->ev1, st));
  c->timed = true;
  return 0;
}

// --------------------------------------------------------------------------- DAC encoder (row N4)
// DAC.encode, continuous=True (dac.py:236-278): waveform [clips, 1, T] (T a multiple of the hop) ->
// posterior parameters [clips, 2*latent, T/hop] = quant_conv(encoder(x)).  Same engine as the
// decoder: time-major fp32 activations, conv-as-GEMM with the residual + snake epilogue; the strided
// down-sampling convs (k = 2s, stride s, pad ceil(s/2)) are GEMMs over virtual rows that advance s
// source rows (GemmArgs::rstride).
extern "C" int foley_dac_encode(foley_ctx* c, const float* wave, int clips, int T, int enc_dim, const int32_t* rates,
                                int n_rates, float* params, void* stream_v) {
  if (!c || !wave || !params || !rates || clips < 1 || T < 1 || n_rates < 1 || n_rates > 8 || enc_dim < 1)
    return FAIL(FOLEY_ERR_INVALID, "bad argument");
  long hop = 1;
  for (int i = 0; i < n_rates; ++i) hop *= rates[i];
  if (T % hop) return FAIL(FOLEY_ERR_INVALID, "waveform length must be a multiple of the codec hop (DAC.preprocess pads it)");
  hipStream_t st = (hipStream_t)stream_v;
  HIPTRY(hipSetDevice(c->device));
  const foley_config& f = c->cfg;
  const int L = f.latent_dim;
  size_t maxel = 0;
  {
    long t = T;
    int ch = enc_dim;
    for (int i = 0; i <= n_rates; ++i) {
      maxel = std::max(maxel, (size_t)t * ch);
      if (i < n_rates) { t /= rates[i]; ch *= 2; }
    }
  }
  const int Tz = (int)(T / hop);
  {
    std::lock_guard<std::mutex> setup_lock(g_setup_mutex);
    HIPTRY(hipStreamSynchronize(st));
    TRY(grow(c->dacP, maxel * clips * 4));
    TRY(grow(c->dacQ, maxel * clips * 4));
    TRY(grow(c->dacR, maxel * clips * 4));
    TRY(grow(c->dacZ, (size_t)clips * Tz * L * 4 * 3));
  }
  float *S_in = (float*)c->dacP.p, *X = (float*)c->dacQ.p, *S_alt = (float*)c->dacR.p;
  float* Z0 = (float*)c->dacZ.p;
  float* Z1 = Z0 + (size_t)clips * Tz * L;
  HIPTRY(hipEventRecord(c->ev0, st));

  int Tin = T, C = enc_dim;
  {
    const void *w, *b, *a;
    TRY(get_tensor(c, "enc.in.w", FOLEY_F32, {7 * C}, &w));
    TRY(get_tensor(c, "enc.in.b", FOLEY_F32, {C}, &b));
    TRY(get_tensor(c, "enc.0.0.a1", FOLEY_F32, {C}, &a));
    TRY(launch_dac_in(wave, (const float*)w, (const float*)b, (const float*)a, clips, T, C, X, S_in, st));
  }
  for (int i = 0; i < n_rates; ++i) {
    const int s = rates[i], Cout = 2 * C, Tout = Tin / s, pad = (s + 1) / 2;
    const std::string p = "enc." + std::to_string(i) + ".";
    for (int j = 0; j < 3; ++j) {
      const int d = f.dac_dilations[j];
      const std::string u = p + std::to_string(j) + ".";
      Lin c7, c1;
      const void *a2, *an;
      TRY(get_lin(c, u + "c7", FOLEY_F32, C, 7 * C, true, &c7));
      TRY(get_lin(c, u + "c1", FOLEY_F32, C, C, true, &c1));
      TRY(get_tensor(c, u + "a2", FOLEY_F32, {C}, &a2));
      // alpha of whatever consumes this unit's output next: the next unit, or the snake before the strided conv
      TRY(get_tensor(c, j < 2 ? p + std::to_string(j + 1) + ".a1" : p + "alpha", FOLEY_F32, {C}, &an));
      {
        GemmArgs g = gemm_conv(S_in, clips * Tin, Tin, C, 7, d, c7, nullptr, C);
        g.out1 = S_alt; g.alpha = (const float*)a2; g.alphaC = C;
        TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
      }
      {
        GemmArgs g = gemm_plain(S_alt, clips * Tin, c1, X, C);
        g.res = X; g.out1 = S_in; g.alpha = (const float*)an; g.alphaC = C;
        TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
      }
    }
    {
      // strided conv: output row q of a clip reads source rows q*s - pad + j, j < 2s  (dac.py:55-61)
      Lin down;
      const void* an;
      TRY(get_lin(c, p + "down", FOLEY_F32, Cout, 2 * s * C, true, &down));
      TRY(get_tensor(c, i + 1 < n_rates ? "enc." + std::to_string(i + 1) + ".0.a1" : std::string("enc.out.alpha"), FOLEY_F32,
                     {Cout}, &an));
      GemmArgs g = gemm_plain(S_in, clips * Tout, down, X, Cout);
      g.lda = C; g.segV = Tout; g.segS = Tin; g.taps = 2 * s; g.tapC = C; g.dil = 1; g.tap0 = -pad; g.rstride = s;
      g.out1 = S_alt; g.alpha = (const float*)an; g.alphaC = Cout;
      TRY(launch_gemm(g, FOLEY_F32, EPI_DAC, 0, st));
      std::swap(S_in, S_alt);
    }
    Tin = Tout;
    C = Cout;
  }
  {
    Lin co, qc;
    TRY(get_lin(c, "enc.out", FOLEY_F32, L, 3 * C, true, &co));
    TRY(get_lin(c, "enc.qc", FOLEY_F32, 2 * L, L, true, &qc));
    TRY(launch_gemm(gemm_conv(S_in, clips * Tin, Tin, C, 3, 1, co, Z0, L), FOLEY_F32, EPI_STORE_F32, 0, st));
    TRY(launch_gemm(gemm_plain(Z0, clips * Tin, qc, Z1, 2 * L), FOLEY_F32, EPI_STORE_F32, 0, st));
    TRY(launch_rows_to_planes(Z1, clips, Tin, 2 * L, params, st));
  }
  HIPTRY(hipEventRecord(c->ev1, st));
  c->timed = true;
  return 0;
}

// --------------------------------------------------------------------------- op-level entry points
static RowBcast to_rb(const foley_rowbcast* r) {
  if (!r || !r->p) return rb_none();
  const int L = r->L > 0 ? r->L : 1, Ls = r->mode == 2 ? (r->Ls > 0 ? r->Ls : 1) : 0;
  const int per = (r->mode == 2 && r->period > 0 && !(r->period & (r->period - 1))) ? r->period : 0;
  RowBcast b{r->p, (long)r->ld, r->mode, r->rows_per_cfg > 0 ? r->rows_per_cfg : 1, L, nullptr, 0, Ls, Ls ? (float)Ls / (float)L : 0.f, per, 0, 0};
  if (per > 0) {
    b.dense_from = r->periodic_cfgs > 0 ? r->periodic_cfgs : INT_MAX;
    b.dense_base = r->periodic_cfgs > 0 ? r->periodic_cfgs * per : 0;
  }
  return b;
}

extern "C" int foley_op_gemm(const foley_gemm_desc* d, void* stream) {
  if (!d) return FAIL(FOLEY_ERR_INVALID, "null descriptor");
  GemmArgs g{};
  g.A = d->A; g.W = d->W; g.bias = d->bias; g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda;
  g.segV = d->segV; g.segS = d->segS; g.taps = d->taps; g.tapC = d->tapC; g.dil = d->dil; g.tap0 = d->tap0;
  g.out0 = d->out0; g.out1 = d->out1; g.osegV = d->osegV; g.out_seg = d->out_seg; g.out_row = d->out_row;
  g.out_shift = d->out_shift; g.out_check = d->out_check; g.rb = to_rb(&d->rb); g.res = d->res;
  g.alpha = d->alpha; g.alphaC = d->alphaC > 0 ? d->alphaC : 1;
  g.ksplit = d->ksplit;
  g.rstride = d->rstride;
  g.ldw = d->ldw;
  g.wfmt = d->wfmt;
  g.gelu_erf = d->gelu_erf;
  if (d->partials) {
    if (d->partial_slabs < 1) return FAIL(FOLEY_ERR_INVALID, "partials need partial_slabs >= 1");
    g.partials = d->partials; g.partial_stride = (long)d->M * d->N; g.partial_cap = d->partial_slabs;
    g.partial_half = d->partial_dtype != 0;
    if (d->partial_dtype != 0 && d->partial_dtype != d->dtype) return FAIL(FOLEY_ERR_INVALID, "partial_dtype must be 0 (fp32) or the operand dtype");
  }
  g.zeros = zero_page();
  if (g.segV < 1 || g.segS < 1 || g.osegV < 1 || g.taps < 1) return FAIL(FOLEY_ERR_INVALID, "bad GEMM descriptor");
  if (d->epilogue == EPI_QKV_SPLIT) {
    const foley_qkv_split_desc* q = d->qkv;
    if (!q) return FAIL(FOLEY_ERR_INVALID, "epilogue 7 needs a head-split descriptor");
    QkvSplitArgs& a = g.qs;
    a.qkv = nullptr; a.M = d->M; a.L = q->L; a.H = q->H; a.nK = q->nK;
    for (int i = 0; i < 3; ++i) { a.gain[i] = q->gain[i]; a.pos[i] = q->pos[i]; a.dst[i] = q->dst[i]; }
    a.S_tot = q->S_tot; a.tok_off = q->tok_off; a.out_dtype = q->out_dtype; a.vt_pitch = q->vt_pitch;
    a.eps = q->eps; a.cos_tab = q->cos_tab; a.sin_tab = q->sin_tab;
    a.attn_k = q->attn_k; a.attn_vt = q->attn_vt; a.attn_out = q->attn_out;
    a.attn_skv = q->attn_skv; a.attn_pitch = q->attn_pitch; a.attn_bdiv = q->attn_bdiv; a.attn_fused = q->attn_fused;
    if (q->attn_fused) *q->attn_fused = 0;
  }
  int ks = 1;
  const int rc = launch_gemm(g, d->dtype, d->epilogue, d->tile, (hipStream_t)stream, &ks);
  if (d->ksplit_used) *d->ksplit_used = ks;
  return rc;
}

extern "C" int foley_op_attention_hd(const void* q, const void* k, const void* v, int in_dtype, int vt_pitch, int Bq,
                                     int H, int Sq, int Skv, int kv_bdiv, void* outA, void* outB, int split,
                                     int out_dtype, int head_dim, void* stream) {
  AttnArgs a{q, k, v, Bq, H, Sq, Skv, kv_bdiv > 0 ? kv_bdiv : 1, outA, outB, split, in_dtype, vt_pitch, head_dim};
  return launch_attention(a, out_dtype, (hipStream_t)stream);
}

extern "C" int foley_op_attention_scatter(const void* q, const void* k, const void* v, int in_dtype, int vt_pitch, int G, int H, int Sq,
                                          int Skv, int grp_q, int grp_kv, const int32_t* out_rows, void* out, int out_nrows, int out_dtype, void* stream) {
  if (!out_rows || !out) return FAIL(FOLEY_ERR_INVALID, "null argument");
  if (out_nrows < 1) return FAIL(FOLEY_ERR_INVALID, "attention_scatter: out has no rows");
  AttnArgs a{q, k, v, G, H, Sq, Skv, 1, out, out, 0, in_dtype, vt_pitch, 64};
  a.grp_q = grp_q; a.grp_kv = grp_kv;
  a.out_rows = out_rows;
  a.out_nrows = out_nrows;
  return launch_attention(a, out_dtype, (hipStream_t)stream);
}

extern "C" int foley_op_attention(const void* q, const void* k, const void* v, int in_dtype, int vt_pitch, int Bq,
                                  int H, int Sq, int Skv, int kv_bdiv, void* outA, void* outB, int split,
                                  int out_dtype, void* stream) {
  return foley_op_attention_hd(q, k, v, in_dtype, vt_pitch, Bq, H, Sq, Skv, kv_bdiv, outA, outB, split, out_dtype, 128, stream);
}

extern "C" int foley_op_qkv_regroup(const void* qkv, int n_rows, int dtype, int H, const int32_t* idx_q, int G, int Sq, const int32_t* idx_kv, int Skv,
                                    void* q, void* k, void* v, int vt_pitch, void* stream) {
  if (!qkv || !idx_q || !idx_kv || !q || !k || !v) return FAIL(FOLEY_ERR_INVALID, "null argument");
  return launch_qkv_regroup(qkv, n_rows, dtype, H, idx_q, G, Sq, idx_kv, Skv, q, k, v, vt_pitch, (hipStream_t)stream);
}

extern "C" int foley_op_resize_aa_u8(const uint8_t* in, long outer, int len_in, long inner, int len_out, const int32_t* xmin,
                                     const int32_t* xsize, const int16_t* weights, int kmax, int precision, uint8_t* out, void* stream) {
  if (!in || !xmin || !xsize || !weights || !out) return FAIL(FOLEY_ERR_INVALID, "null argument");
  return launch_resize_aa_u8(in, outer, len_in, inner, len_out, xmin, xsize, weights, kmax, precision, out, (hipStream_t)stream);
}

extern "C" int foley_op_ln_mod(const float* x, int M, int D, float eps, const foley_rowbcast* shift,
                               const foley_rowbcast* scale, void* out, int out_dtype, void* stream) {
  return launch_ln_mod(x, M, D, eps, to_rb(shift), to_rb(scale), out, out_dtype, (hipStream_t)stream);
}

extern "C" int foley_op_ln_mod_pending2(float* x, int M, int D, float eps, const foley_rowbcast* shift,
                                        const foley_rowbcast* scale, void* out, int out_dtype, const void* partials,
                                        int partial_dtype, int k, const float* bias, const foley_rowbcast* gate, void* stream) {
  if (!partials || k < 1 || !gate) return FAIL(FOLEY_ERR_INVALID, "pending split-K: partials, k >= 1 and a gate are required");
  if (partial_dtype != 0 && partial_dtype != out_dtype) return FAIL(FOLEY_ERR_INVALID, "pending split-K: 16-bit slabs must have the output dtype");
  LnPending p{(const float*)partials, k, (long)M * D, bias, to_rb(gate), partial_dtype};
  return launch_ln_mod_pending(x, M, D, eps, to_rb(shift), to_rb(scale), out, out_dtype, p, (hipStream_t)stream);
}

extern "C" int foley_op_ln_mod_pending(float* x, int M, int D, float eps, const foley_rowbcast* shift,
                                       const foley_rowbcast* scale, void* out, int out_dtype, const float* partials,
                                       int k, const float* bias, const foley_rowbcast* gate, void* stream) {
  return foley_op_ln_mod_pending2(x, M, D, eps, shift, scale, out, out_dtype, partials, 0, k, bias, gate, stream);
}

extern "C" int foley_op_qkv_split(const float* qkv, int M, int L, int H, int nK, const float* const* gain,
                                  const int32_t* const* pos, void* const* dst, int out_dtype, int vt_pitch,
                                  int S_tot, int tok_off, float eps, const float* cos_tab, const float* sin_tab,
                                  void* stream) {
  if (nK < 1 || nK > 3) return FAIL(FOLEY_ERR_INVALID, "nK must be 1..3");
  QkvSplitArgs a{};
  a.qkv = qkv; a.M = M; a.L = L; a.H = H; a.nK = nK;
  for (int i = 0; i < nK; ++i) {
    a.gain[i] = gain ? gain[i] : nullptr;
    a.pos[i] = pos ? pos[i] : nullptr;
    a.dst[i] = dst[i];
  }
  a.out_dtype = out_dtype; a.vt_pitch = vt_pitch;
  a.S_tot = S_tot; a.tok_off = tok_off; a.eps = eps; a.cos_tab = cos_tab; a.sin_tab = sin_tab;
  return launch_qkv_split(a, (hipStream_t)stream);
}

extern "C" int foley_op_solver_step(const float* pred, float* x, float* x_saved, float* d_acc, int clips, int C,
                                    int L, int ncfg, float guidance, const float* coef, int32_t* step_ptr,
                                    void* rows_out, int rows_dtype, void* stream) {
  StepArgs s{pred, x, x_saved, d_acc, clips, C, L, ncfg, guidance, coef, step_ptr, rows_out, rows_dtype};
  return launch_solver_step(s, (hipStream_t)stream);
}

extern "C" int foley_op_latent_rows(const float* x, int clips, int C, int L, int ncfg, void* out, int out_dtype,
                                    void* stream) {
  return launch_latent_rows(x, clips, C, L, ncfg, out, out_dtype, (hipStream_t)stream);
}

extern "C" int foley_op_dac_out(const float* s, const float* w, const float* bias, int B, int T, int C, float* out,
                                void* stream) {
  return launch_dac_out(s, w, bias, B, T, C, out, (hipStream_t)stream);
}
