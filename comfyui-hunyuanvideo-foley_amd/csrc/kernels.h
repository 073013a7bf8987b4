// Internal launch interface between the kernel files and the runtime (not part of the C ABI).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// GEMM / conv-as-GEMM engine:  D[r, n] = sum_k A'[r, k] * W[n, k]  (+ fused epilogue)
//
// A' is a *virtual* row matrix over a channels-last activation [rows, C]: virtual row r belongs to
// segment b = r / segV at position q = r % segV, and its K axis is `taps` blocks of `tapC`
// channels, block j being source row (q + tap0 + j*dil) of segment b (zero when that row falls
// outside [0, segS)).  taps=1 is a plain linear layer; taps=3,dil=1,tap0=-1 is the DiT's
// channels-last conv k=3 pad=1; taps=7 is the DAC dilated conv; taps=2,tap0=-1 over segV=T+1
// virtual rows is a stride-s transposed conv whose s output phases are the N axis.
// ---------------------------------------------------------------------------------------------
enum GemmEpi {
  EPI_STORE_F32 = 0,  // out0(f32) = acc + bias (+ rb addend)
  EPI_STORE_T = 1,    // out0(T)   = acc + bias
  EPI_SILU_T = 2,     // out0(T)   = silu(acc + bias)
  EPI_GELU_T = 3,     // out0(T)   = gelu_tanh(acc + bias)
  EPI_SILUGATE_T = 4, // out0(T)[.., N/2] = silu(a) * b, (a, b) = alternating 32-column groups
  EPI_GATE_RES = 5,   // out0(f32) += rb gate * (acc + bias)
  EPI_DAC = 6,        // v = acc + bias (+ res); out0(f32) = v; out1(f32) = snake(v)
  EPI_QKV_SPLIT = 7,  // fused head split: per 128-column head tile RMSNorm + RoPE (q, k) or V^T, written
                      // straight to the attention operands (GemmArgs::qs); needs a 128-wide tile
  EPI_QKV_ATTN = 8,   // internal (chosen by the launcher for EPI_QKV_SPLIT problems with qs.attn_out): head split of a q
                      // projection + attention against the cached keys in the same epilogue
  EPI_COUNT
};

struct QkvSplitArgs {
  const float* qkv;  // [M, nK * H * 128]
  int M, L, H, nK;   // rows ordered [clip][l], L tokens per clip
  const float* gain[3];  // RMSNorm gain per operand, null => copy only
  const int* pos[3];     // RoPE position per token l, null => no rotation
  void* dst[3];          // [clips, H, S_tot, 128] (out_dtype); see vt_pitch for the last operand
  int S_tot, tok_off;
  int out_dtype;         // FOLEY_F32 or FOLEY_BF16
  int vt_pitch;          // > 0: the LAST operand (V) is stored transposed [clips, H, 128, vt_pitch]
  float eps;
  const float* cos_tab;  // [P, 64]
  const float* sin_tab;
  // optional, fused GEMM epilogue only: rotation rows already gathered per token, rcos[o][l] = cos_tab[pos[o][l]]
  // ([L, 64] fp32 each) - one load level instead of two dependent ones in front of the stores
  const float* rcos[3];
  const float* rsin[3];
  // optional, fused GEMM epilogue only (nK = 1, 16-bit operands, 64-row tiles: the cross-attention q projection): run the
  // attention against <= 96 cached keys inside the epilogue and write its output rows instead of q.
  const void* attn_k;    // [sets, H, attn_skv, 128] operand type
  const void* attn_vt;   // [sets, H, 128, attn_pitch] operand type, finite beyond attn_skv
  void* attn_out;        // [M, H * 128] operand type; null => plain head split
  int attn_skv, attn_pitch;
  int attn_bdiv;         // text set of row r: (r / L) / attn_bdiv
  int* attn_fused;       // HOST pointer (never read by the device), may be null: the launcher stores 1 when the fused form ran,
                         // 0 when the problem took the plain head split (q in dst[0]; the caller launches the attention)
};
struct GemmArgs {
  const void* A;
  const void* W;      // [N, K] row-major, K contiguous
  const float* bias;  // [N] or null
  int M, N, K;
  long lda;           // elements between source rows of A
  long ldw;           // elements between rows of W (>= K; set by the launcher to K when 0) - padded rows spread the
                      // 128-byte K-slice lines of consecutive rows over the L2 channels
  int segV, segS;     // virtual / source rows per segment
  int taps, tapC, dil, tap0;
  int rstride;        // source rows advanced per virtual row (strided conv: tap t of row q reads q*rstride + tap0 + t*dil); 0 = 1
  // output mapping: element (r, n) -> b*out_seg + q*out_row + n + out_shift with (b, q) taken
  // over osegV rows; skipped when out_check and the in-segment offset leaves [0, out_seg)
  void* out0;
  void* out1;
  int osegV;
  long out_seg, out_row, out_shift;
  int out_check;
  RowBcast rb;        // gate (EPI_GATE_RES) or addend (EPI_STORE_F32)
  const float* res;   // EPI_DAC residual (same mapping as out0) or null
  const float* alpha; // EPI_DAC snake alpha, indexed n % alphaC
  int alphaC;
  unsigned a_bytes, w_bytes;   // operand extents in bytes (buffer-resource range of the direct-to-LDS loop; set by the launcher)
  const void* zeros;  // >= 128 zero bytes in global memory (source of masked rows for the direct-to-LDS loop)
  int ksplit;         // EPI_GATE_RES only: K is cut into `ksplit` ranges whose partial products are
                      // accumulated with hardware fp32 atomics (0 = auto; 1 = deterministic)
  // Deferred split-K (EPI_GATE_RES, ksplit > 1): with `partials` set, K range s stores its raw partial
  // product to partials + s*partial_stride ([M, N] row-major fp32, vector stores) instead of using
  // atomics; the LayerNorm that consumes the residual stream next finishes
  // x += gate * (sum_s partial_s + bias) (LnPending).  partial_cap = slabs available (caps ksplit).
  float* partials;    // slab base (fp32 view; with partial_half the slabs hold the 16-bit OPERAND type, 2 bytes per element)
  int partial_half;   // 1: slabs are stored in the operand type T (bf16 / fp16) - half the slab bytes written here and read by
                      // the LayerNorm; the sum of k rounded partials carries about the error of ONE rounding of the total
  long partial_stride;
  int partial_cap;
  int n_groups;       // set by the launcher, large grids: > 0 = tile order [column-panel group][M tile][panel inside the group] with this many
                      // groups (the workgroups that run together then cover a near-square block of tiles); 0 = [panel][M tile]
  int ks_major;       // set by the launcher: workgroup order of a split - 1: K range slowest (all tiles of a range are neighbours, i.e. on
                      // one or two XCDs: only those L2s fetch that range's activation columns), 0: K range fastest
  int k_rot;          // set by the launcher (small grids, plain layers): M tile tm of a weight panel starts its K walk at slice tm * nk / tiles_m
                      // and wraps - the M tiles of a panel (neighbours on one XCD) then reach every weight line at different times: ONE of
                      // them takes the HBM miss, the others find the line in their L2 instead of all waiting for the same fill in lockstep
  QkvSplitArgs qs;    // EPI_QKV_SPLIT: destination / norm / rotation description (qs.qkv, qs.M unused)
  int vec_out;        // set by the launcher: the problem qualifies for the LDS-transposed vector epilogue
  int wfmt;           // storage of W: 0 = the operand dtype, 1 = fp8 e4m3fn, 2 = fp8 e5m2 (bf16 activations; wave-specialised
                      // tiles only - widened to bf16 in registers, reference FP8WeightWrapper utils.py:316-366)
  int gelu_erf;       // EPI_GELU_T: exact (erf) GELU instead of the tanh form (the conditioning encoders' nn.GELU())
  int pf_dist;        // wave-specialised mainloop: L2 prefetch distance in K-slices beyond the LDS ring (0 = off; set by the launcher)
  int dbg_mode;
  long long* dbg;     // tools/gemm_timeline.py: 4 wall-clock stamps per workgroup (entry, first slice
                      // landed, K loop done, epilogue done); null in production
};

// dtype: FOLEY_F32, FOLEY_BF16 or FOLEY_F16 operands (accumulation is always fp32). tile: 0 = auto.
// ksplit_used (optional): the K split the launcher chose (callers of the deferred split-K need it)
int launch_gemm(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t st, int* ksplit_used = nullptr);
// Two independent problems of the same dtype / epilogue in ONE launch (the audio and the visual
// stream of a two-stream block): the small problem's workgroups hide in the large one's shadow.
// Tile shape and K split are chosen for g0.
int launch_gemm_pair(const GemmArgs& g0, const GemmArgs& g1, int dtype, int epi, hipStream_t st, int* ksplit_used = nullptr);
// tap-fused channels-last conv k=3 (gemm_conv3.hip); tile 1 = 128x128, 3 = 64x64; g.ksplit resolved
int launch_gemm_conv3(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t st);
// wave-specialised mainloop (gemm_ws_impl.h; one entry per 16-bit operand type): tile 15 = 128x128, 19 = 256x128; g / g1 resolved by launch_gemm
int launch_gemm_ws_bf16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st);
int launch_gemm_ws_f16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st);
// 256x256 tiles on the BK = 32 mainloop (gemm_wide_impl.h): tile 31 = tap-fused conv k=3, 32 = plain linear layer (eight waves of
// 128x64); single problem, vector epilogue, any weight storage
extern thread_local int g_gemm_krot_ok;   // gemm.hip
int launch_gemm_wide_bf16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st);
int launch_gemm_wide_f16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Attention: O = softmax(Q K^T / sqrt(hd)) V, no mask, hd = 128 (or 64: AttnArgs::head_dim).  Q [Bq, H, Sq, hd], K/V [Bkv, H, Skv, hd]
// fp32.  Query batch b reads K/V batch b / kv_bdiv.  Output rows are token-major [.., H*128] in
// dtype `out_dtype`; tokens [0, split) go to outA (clip-major rows of `split` tokens), the rest
// to outB.
// ---------------------------------------------------------------------------------------------
// in_dtype FOLEY_BF16 selects the throughput kernel: Q/K bf16 [B,H,S,128], V transposed bf16
// [B,H,128,vt_pitch] (vt_pitch >= Skv rounded up to 32; the pad must hold finite values).
struct AttnArgs {
  const void* q;
  const void* k;
  const void* v;
  int Bq, H, Sq, Skv, kv_bdiv;
  void* outA;
  void* outB;
  int split;
  int in_dtype;
  int vt_pitch;
  int head_dim;     // 0 / 128: the Foley DiT; 64: the conditioning encoders (fp32 kernel and the 16-bit wide kernel)
  int no_preload;   // set by the launcher (A/B switch FOLEY_ATTN_PRELOAD=0): small-grid kernel without the up-front operand requests
  long long* dbg;   // tools/attn_timeline.py: 5 wall-clock stamps per workgroup of the small-grid bf16 kernel; null in production
  int grp_q, grp_kv;     // head_dim 64, 16-bit operands: > 0 = block-diagonal attention - query t attends keys [g*grp_kv, (g+1)*grp_kv), g = t / grp_q
                         // (small groups packed into one sequence: the Synchformer's 8-frame time groups, 14 of them per 128-query workgroup)
  int out_nrows;         // rows of outA when out_rows is set: table entries outside [0, out_nrows) are dropped by the kernels
  const int* out_rows;   // head_dim 64 only: row of outA that query (b, t) is written to - [Bq, Sq]; null: [Bq, Sq] order (split above).
                         // The conditioning encoders scatter every attention of a layer (CLS rows, time / space groups) into ONE
                         // token-major buffer with the table their queries were gathered by - no torch.cat / permute copies.
};
int launch_attention(const AttnArgs& a, int out_dtype, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Row kernels
// ---------------------------------------------------------------------------------------------
// out(T)[r, :] = LayerNorm(x[r, :]; eps) * (1 + scale) + shift   (scale/shift optional)
int launch_ln_mod(const float* x, int M, int D, float eps, const RowBcast& shift, const RowBcast& scale,
                  void* out, int out_dtype, hipStream_t st);
// Pending residual update left by a deferred split-K GEMM (GemmArgs::partials): before normalising,
// x += gate * (sum_{s<k} partials[s] + bias) and the new x is written back.
struct LnPending {
  const float* partials;  // null => nothing pending
  int k;
  long stride;
  const float* bias;      // [D] or null
  RowBcast gate;
  int half;               // 0: fp32 slabs; FOLEY_BF16 / FOLEY_F16: slabs in that 16-bit type (GemmArgs::partial_half)
};
struct LnArgs {
  float* x;
  int M;
  RowBcast shift, scale;
  void* out;
  LnPending pend;
};
int launch_ln_mod_pending(float* x, int M, int D, float eps, const RowBcast& shift, const RowBcast& scale,
                          void* out, int out_dtype, const LnPending& pend, hipStream_t st);
// two row sets (same D / eps / dtype) in one launch
int launch_ln_mod_pair(const LnArgs& a0, const LnArgs& a1, int D, float eps, int out_dtype, hipStream_t st);

int launch_qkv_split(const QkvSplitArgs& a, hipStream_t st);
int launch_qkv_split_pair(const QkvSplitArgs& a0, const QkvSplitArgs& a1, hipStream_t st);

// out(T)[r, :] = act(a[r, :] + v[:]) ; a optional [R, D]; v optional broadcast row (step-indexed)
int launch_rows_add_act(const float* a, const RowBcast& v, int R, int D, int act_silu, void* out,
                        int out_dtype, hipStream_t st);
// *flag |= 1 when, in some group of `rows` rows, a row s >= period is not bit-identical to row s - period (fp32 [groups*rows, D])
int launch_rows_periodic_check(const float* x, int groups, int rows, int period, int D, int* flag, hipStream_t st);
// out(T)[r, :] = x[r, :] + pos[r % period, :]
int launch_add_periodic(const float* x, const float* pos, int R, int D, int period, void* out,
                        int out_dtype, hipStream_t st);
// out[r, :] = src[idx[r % n_idx] + (r / n_idx) * src_rows, :]  (fp32 row gather)
int launch_gather_rows(const float* src, const int* idx, int n_idx, int groups, int src_rows, int D,
                       float* out, hipStream_t st);
// generic dtype cast of a contiguous buffer
int launch_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, hipStream_t st);

// latents x [clips, C, L] fp32 -> rows(T) [(cfg*clips + b)*L + l, C] for cfg in [0, ncfg)
int launch_latent_rows(const float* x, int clips, int C, int L, int ncfg, void* out, int out_dtype,
                       hipStream_t st);

// Solver update (CFG combine + flow-match step) and re-staging of the model input rows.
struct StepArgs {
  const float* pred;   // [ncfg*clips*L, C] model output rows
  float* x;            // [clips, C, L] current sample (updated in place)
  float* x_saved;      // stage-0 sample for multi-stage solvers (or null)
  float* d_acc;        // running derivative combination (or null)
  int clips, C, L, ncfg;
  float guidance;
  const float* coef;   // per-iteration [n_iter][4]: {w_new, w_acc, dt, flags}
  int* step_ptr;       // device iteration counter (incremented by the kernel)
  void* rows_out;      // next model input rows (T), cfg-duplicated
  int rows_dtype;
};
int launch_solver_step(const StepArgs& a, hipStream_t st);

// DAC tail: out[b, t] = tanh(bias + sum_{j<7, c<C} w[j*C + c] * s[b, t + j - 3, c])
// DAC encoder: input conv 1 -> C (k=7) writing y and snake(y); rows [B*T, C] -> planes [B, C, T]
int launch_dac_in(const float* x, const float* w, const float* bias, const float* alpha, int B, int T, int C,
                  float* out0, float* out1, hipStream_t st);
int launch_rows_to_planes(const float* rows, int B, int T, int C, float* out, hipStream_t st);
int launch_qkv_regroup(const void* qkv, int n_rows, int dtype, int H, const int* idx_q, int G, int Sq, const int* idx_kv, int Skv, void* q, void* k,
                       void* v, int vt_pitch, hipStream_t st);
int launch_resize_aa_u8(const uint8_t* in, long outer, int len_in, long inner, int len_out, const int* xmin, const int* xsize,
                        const short* w, int kmax, int prec, uint8_t* out, hipStream_t st);
int launch_dac_out(const float* s, const float* w, const float* bias, int B, int T, int C, float* out,
                   hipStream_t st);
