// Public entry points of the GEMM engine: dtype dispatch to the per-dtype translation units
// (gemm_f32.hip / gemm_bf16.hip / gemm_f16.hip instantiate gemm_impl.h for one operand type each).
#include "kernels.h"

long long* g_gemm_dbg = nullptr;
int g_gemm_dbg_mode = 0;
// K-origin rotation (GemmArgs::k_rot) is an option of the CALLER: a row's summation order then depends on the M tile it falls into, so
// rows with equal inputs no longer come out bit-identical.  The DiT forward of a single clip per CFG half opts in (foley_rt.hip::
// run_forward); foley_prepare (its row-periodicity check compares bit patterns), batches (clips of a batch with equal noise stay
// bit-identical) and the op-level entries do not.
thread_local int g_gemm_krot_ok = 0;
int g_gemm_pf_dist = 0;   // L2 prefetch distance of the wave-specialised mainloop (K-slices beyond the ring)

int launch_gemm_typed_f32(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* ksplit_used);
int launch_gemm_typed_bf16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* ksplit_used);
int launch_gemm_typed_f16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* ksplit_used);

int launch_gemm(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t st, int* ksplit_used) {
  if (ksplit_used) *ksplit_used = 1;
  if (g.M <= 0 || g.N <= 0) return 0;
  if (dtype == FOLEY_F32) return launch_gemm_typed_f32(g, nullptr, epi, tile, st, ksplit_used);
  if (dtype == FOLEY_BF16) return launch_gemm_typed_bf16(g, nullptr, epi, tile, st, ksplit_used);
  if (dtype == FOLEY_F16) return launch_gemm_typed_f16(g, nullptr, epi, tile, st, ksplit_used);
  return foley_set_err("GEMM: unsupported operand dtype", __FILE__, __LINE__);
}

int launch_gemm_pair(const GemmArgs& g0, const GemmArgs& g1, int dtype, int epi, hipStream_t st, int* ksplit_used) {
  if (g1.M <= 0 || g1.N <= 0) return launch_gemm(g0, dtype, epi, 0, st, ksplit_used);
  if (g0.M <= 0 || g0.N <= 0) return launch_gemm(g1, dtype, epi, 0, st, ksplit_used);
  if (dtype == FOLEY_F32) return launch_gemm_typed_f32(g0, &g1, epi, 0, st, ksplit_used);
  if (dtype == FOLEY_BF16) return launch_gemm_typed_bf16(g0, &g1, epi, 0, st, ksplit_used);
  if (dtype == FOLEY_F16) return launch_gemm_typed_f16(g0, &g1, epi, 0, st, ksplit_used);
  return foley_set_err("GEMM: unsupported operand dtype", __FILE__, __LINE__);
}

// Debug hook for tools/gemm_timeline.py (not part of include/foley_hip.h): every following GEMM
// launch writes 4 wall-clock stamps per workgroup to `p` (device memory, 4 * grid * 8 bytes).
extern "C" void foley_debug_gemm_prefetch(int dist) { g_gemm_pf_dist = dist < 0 ? 0 : dist; }

extern "C" void foley_debug_gemm_timeline(void* p, int mode) {
  g_gemm_dbg = (long long*)p;
  g_gemm_dbg_mode = mode;
}
