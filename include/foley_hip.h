/* libfoley_hip.so - C ABI of the MI355X (gfx950) HunyuanVideo-Foley sampling path.
 *
 * The reference (phazei/ComfyUI-HunyuanVideo-Foley) is pure Python and has no FFI of its own;
 * each entry point below replaces a Python-level interface of the reference's hot path, which a
 * maintainer would bind with ctypes (see INTEGRATION.md for the stub):
 *
 *   foley_ctx_create / foley_set_tensor   <- HunyuanModelLoader.load_model (nodes.py:72-133) and
 *                                            load_dac_any (utils.py:61-87): weights -> device
 *   foley_prepare                         <- the per-run setup of denoise_process_with_generator
 *                                            (utils.py:144-199) + the step-invariant part of
 *                                            HunyuanVideoFoley.forward (hifi_foley.py:744-807)
 *   foley_dit_forward                     <- HunyuanVideoFoley.forward (hifi_foley.py:707-924)
 *   foley_sample                          <- the denoising loop (utils.py:201-247) incl.
 *                                            FlowMatchDiscreteScheduler.step
 *                                            (scheduling_flow_match_discrete.py:210-297)
 *   foley_dac_decode                      <- DAC.decode (dac_vae/model/dac.py:280-303)
 *   foley_op_*                            <- the individual torch ops the reference calls on the
 *                                            path (F.linear, conv1d, SDPA, layer_norm, ...), used
 *                                            by the parity tests and micro-benchmarks
 *
 * Conventions: every function returns 0 on success or a negative error code and never throws;
 * foley_last_error() gives the message of the calling thread's last failure.  All tensor
 * arguments are raw DEVICE pointers (torch `tensor.data_ptr()`), owned by the caller and only
 * borrowed for the duration of the call, except tensors registered with foley_set_tensor, which
 * the caller must keep alive until the context is destroyed.  `stream` is a hipStream_t passed
 * as void* (torch.cuda.current_stream().cuda_stream); work is enqueued, not synchronised, unless
 * stated.  A context is used by one thread at a time.  Plain C types only - no torch types.
 */
#ifndef FOLEY_HIP_H
#define FOLEY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOLEY_ABI_VERSION 12   /* 12: foley_op_qkv_regroup.n_rows / foley_op_attention_scatter.out_nrows (the caller-supplied row tables are range-checked on the device: source rows clamped, output rows outside the buffer dropped); 11: foley_op_resize_aa_u8 (the frames' antialiased uint8 resize, bit for bit), foley_op_attention_scatter, foley_rowbcast.periodic_cfgs; 10: foley_op_qkv_regroup (token regrouping of the conditioning encoders' attention); 9: foley_bcast_local (single-process grouped broadcast of the arenas); 8: foley_qkv_split_desc.attn_* (cross attention in the epilogue of its q projection), foley_abort / FOLEY_ERR_ABORTED; 7: FOLEY_DT_F16 as a compute dtype (foley_config.compute_dtype, op descriptors); 6: foley_rowbcast.Ls (mode 2: nearest-exact up-sampled operand); 5: reference-keyed loading (foley_weights_*, foley_load_tensor, foley_bcast_weights); 4: fp8 weight storage (dtype codes 3/4, foley_gemm_desc.ldw/.wfmt); 3: foley_profile_forward; 2: foley_gemm_desc gained partials / qkv / rstride; foley_op_ln_mod_pending, foley_dac_encode */

enum foley_dtype {
  FOLEY_DT_F32 = 0, FOLEY_DT_BF16 = 1, FOLEY_DT_I32 = 2,
  FOLEY_DT_F8E4M3 = 3,  /* OCP e4m3fn, weight storage only (reference FP8WeightWrapper, utils.py:316-366: plain cast, no scales) */
  FOLEY_DT_F8E5M2 = 4,  /* OCP e5m2,   weight storage only */
  FOLEY_DT_F16 = 5      /* IEEE fp16: a checkpoint dtype, and a compute dtype (the loader's precision=fp16, which the reference
                           runs under torch.autocast(float16): nodes.py:89-106, utils.py:229-234) */
};

enum foley_status {
  FOLEY_OK = 0,
  FOLEY_ERR_INVALID = -1,   /* bad argument / shape / dtype            */
  FOLEY_ERR_MISSING = -2,   /* a required tensor was never registered  */
  FOLEY_ERR_HIP = -3,       /* HIP runtime failure                      */
  FOLEY_ERR_STATE = -4,     /* call order violated (e.g. no prepare)    */
  FOLEY_ERR_ABORTED = -5    /* foley_abort() ended the sampling loop    */
};

typedef struct foley_ctx foley_ctx;

/* Model dimensions (configs/hunyuanvideo-foley-xxl.yaml model_kwargs; utils.py:32-44 for DAC). */
typedef struct foley_config {
  int32_t depth_triple, depth_single, hidden, heads;
  int32_t mlp_hidden;    /* triple-block MLP hidden  = hidden * mlp_ratio          */
  int32_t conv_hidden;   /* single-block ConvMLP hidden (mlp_layers.py:141-142)     */
  int32_t sync_hidden;   /* sync_in ConvMLP hidden                                 */
  int32_t cond_dim, clip_dim, sync_dim, latent_dim, time_freq_dim;
  int32_t compute_dtype; /* FOLEY_DT_F32 (parity mode), FOLEY_DT_BF16 or FOLEY_DT_F16 (throughput) for DiT GEMM operands */
  int32_t dac_dim;       /* decoder width (2048)                                    */
  int32_t dac_n_rates;   /* 5                                                       */
  int32_t dac_rates[8];  /* 8,5,4,3,2                                               */
  int32_t dac_dilations[3];
} foley_config;

/* One sampling run: conditioning + host-built index/trig tables (all device pointers). */
typedef struct foley_plan {
  int32_t ncfg;          /* 2 with classifier-free guidance ([uncond ; cond]), else 1 (utils.py:193-199) */
  int32_t clips;         /* batch_size: independent clips sharing the conditioning   */
  int32_t La, Lv, Ls, Lt;/* audio / visual / sync / text token counts                */
  int32_t n_iter;        /* loop iterations (= steps; multi-stage solvers still do one model call per iteration) */
  float guidance;
  const float* text;     /* [ncfg, Lt, cond_dim]  zero-padded to Lt (utils.py:103-111) */
  const float* clip;     /* [ncfg, Lv, clip_dim]  */
  const float* sync;     /* [ncfg, Ls, sync_dim]  */
  const float* t_feat;   /* [n_iter, time_freq_dim] sinusoidal timestep features (embed_layers.py:76-101) */
  const float* rope_cos; /* [rope_len, 64] cos(pos * theta^(-2k/128)) (posemb_layers.py:117-172) */
  const float* rope_sin;
  int32_t rope_len;      /* >= 2*La                                                 */
  const int32_t* pos_audio_self;  /* [La]  interleaved-RoPE position of audio token i  (= 2i)        */
  const int32_t* pos_visual_self; /* [Lv]  interleaved-RoPE position of visual token j (hifi_foley.py:35-60) */
  const int32_t* pos_linear;      /* [max(La,Lv,Lt)] 0,1,2,...                          */
  const int32_t* sync_gather;     /* [La]  nearest-exact source row of the sync up-sampling (hifi_foley.py:761) */
  const float* solver_coef;       /* [n_iter, 8] {w_new, w_acc, dt, w_store, flags,0,0,0} per iteration */
} foley_plan;

typedef void (*foley_progress_cb)(int32_t iteration, int32_t n_iter, void* user);

uint32_t foley_abi_version(void);
const char* foley_last_error(void);

int foley_ctx_create(int device, const foley_config* cfg, foley_ctx** out);
void foley_ctx_destroy(foley_ctx* ctx);

/* Register one packed tensor (names and layouts: DESIGN.md "packed weight arena"). */
int foley_set_tensor(foley_ctx* ctx, const char* name, const void* dev_ptr, int dtype, int ndim,
                     const int64_t* shape);

/* ---- Reference-keyed loading (replaces HunyuanModelLoader.load_model nodes.py:72-133 + load_dac_any
 * utils.py:61-87 for a caller that keeps the reference's checkpoints / loader): hand over the tensors under
 * their STATE-DICT KEYS and the library packs them on the device into ONE ctx-owned arena (layouts of
 * (K H D) q/k/v rows, tap-major convs, interleaved SwiGLU pairs, fused single-block
 * modulation, folded weight-norm, transposed convs as phases x 2 taps) and registers the packed tensors.
 *   foley_weights_begin(ctx, fmt)      allocate + register; fmt 0: block matrices in the compute dtype,
 *                                      1 / 2: kept in fp8 e4m3fn / e5m2 (reference _wrap_fp8_inplace,
 *                                      utils.py:408-485; bf16 compute only)
 *   foley_load_tensor(ctx, key, ...)   one checkpoint tensor (f32 / bf16 / f16 / fp8, any order), borrowed for
 *                                      the call; returns 1 for keys the sampling path does not use
 *                                      (DAC encoder / quantizer, final_layer.adaLN_modulation)
 *   foley_weights_end(ctx)             fails with FOLEY_ERR_MISSING if a packed tensor is incomplete
 *   foley_weights_arena                the arena (device pointer, bytes): layout depends on the config only
 *   foley_bcast_weights(ctx, comm, root, stream)   ONE ncclBroadcast (RCCL over xGMI) of the arena on the
 *                                      caller's ncclComm_t; non-root ranks call foley_weights_begin first.
 *                                      (A host that broadcasts the arena itself - e.g. torch.distributed on
 *                                      the pointer - calls foley_weights_mark_received afterwards.) */
int foley_weights_begin(foley_ctx* ctx, int weight_format);
int foley_load_tensor(foley_ctx* ctx, const char* ref_key, const void* dev_ptr, int dtype, int ndim,
                      const int64_t* shape, void* stream);
int foley_weights_end(foley_ctx* ctx, void* stream);
int foley_weights_arena(foley_ctx* ctx, void** dev_ptr, uint64_t* bytes);
int foley_weights_mark_received(foley_ctx* ctx);
int foley_bcast_weights(foley_ctx* ctx, void* nccl_comm, int root, void* stream);
/* Single-process form of the same step (one host process that drives all GPUs of the node, e.g. a ComfyUI prompt worker; the
 * reference has no counterpart - its only batching is utils.py:159-162 on one device): buffer i of devices[0] (`bytes[i]` bytes at
 * bufs[0 * nbuf + i]) is broadcast to bufs[d * nbuf + i] on every other device in ONE grouped RCCL launch (ncclCommInitAll over
 * `devices`, cached; ncclGroupStart / ncclBroadcast per (device, buffer) / ncclGroupEnd), then every device is synchronised. */
int foley_bcast_local(int ndev, const int* devices, int nbuf, void* const* bufs, const uint64_t* bytes);

/* Step-invariant precompute for one run; allocates/reuses the context workspace. */
int foley_prepare(foley_ctx* ctx, const foley_plan* plan, void* stream);

/* One DiT evaluation at loop iteration `iter` (its timestep modulation): latents [clips,C,La]
 * fp32 -> velocity rows [ncfg*clips*La, C] fp32 (row = (cfg*clips + clip)*La + l). */
int foley_dit_forward(foley_ctx* ctx, const float* latents, int iter, float* out_rows, void* stream);

/* Full denoising loop: `latents` [clips,C,La] fp32 holds the initial noise on entry and the final
 * latents on return.  With a progress callback the stream is synchronised once per iteration.
 * use_graph != 0 replays one captured hipGraph per iteration. */
int foley_sample(foley_ctx* ctx, float* latents, int use_graph, foley_progress_cb cb, void* user, void* stream);
/* Cancel a running foley_sample (the ComfyUI "interrupt": comfy.utils.ProgressBar.update raises inside the reference's loop,
 * utils.py:201,247).  Callable from the progress callback or from another thread; the loop stops after the current iteration
 * and foley_sample returns FOLEY_ERR_ABORTED with `latents` holding the state reached.  The request is consumed by the loop it
 * stops; one that arrives while no loop runs is dropped by the next foley_sample on entry. */
int foley_abort(foley_ctx* ctx);

/* DAC-VAE decoder: latents [clips, latent_dim, T] fp32 -> waveform [clips, 1, T*hop] fp32. */
int foley_dac_decode(foley_ctx* ctx, const float* latents, int clips, int T, float* wave, void* stream);

/* DAC-VAE encoder (DAC.encode with continuous=True, dac.py:236-278 + Encoder :47-95):
 * waveform [clips, 1, T] fp32, T a multiple of the hop (DAC.preprocess right-pads, :225-234) ->
 * posterior parameters [clips, 2*latent_dim, T/hop] fp32 (rows [:latent] mean, [latent:] logvar of the
 * DiagonalGaussianDistribution, nn/vae_utils.py:24-31).  enc_dim / rates: encoder_dim and
 * encoder_rates of the checkpoint (128, {2,3,4,5,8} for the 48 kHz VAE, utils.py:32-44); needs the
 * packed `enc.*` tensors registered. */
int foley_dac_encode(foley_ctx* ctx, const float* wave, int clips, int T, int enc_dim, const int32_t* rates,
                     int n_rates, float* params, void* stream);

/* Per-kernel profile of the DiT forward (bench.py's live roofline): runs `repeats` EAGER forwards at loop
 * iteration `iter`, every op's kernel launched with start / stop events attached to the dispatch itself
 * (hipExtLaunchKernelGGL on `stream`), and aggregates by op.  `calls`, `total_ms` (kernel time proper),
 * `flop` (algorithmic FLOPs: 2*M*N*K per contraction, 4*B*H*Sq*Skv*128 per attention) and `bytes`
 * (operands + result, read/written once) are totals over all repeats.  *bracket_ms = elapsed time of
 * an empty hipEventRecord bracket, for reference (not contained in total_ms).  Replaces nothing in
 * the reference: it is the measurement hook SURVEY 8(d) asks for. */
typedef struct foley_prof_entry {
  char label[80];
  int32_t calls;
  float total_ms;
  double flop, bytes;
  char kernel[200];   /* ABI 12: demangled symbol of the kernel the op launched - the row name in a rocprofv3 kernel trace ("" if none) */
} foley_prof_entry;
int foley_profile_forward(foley_ctx* ctx, const float* latents, int iter, int repeats, foley_prof_entry* out,
                          int cap, int* n_out, float* bracket_ms, void* stream);

/* HIP-event time (ms) of the last foley_sample / foley_dac_decode / foley_dac_encode call on this context (syncs). */
int foley_last_elapsed_ms(foley_ctx* ctx, float* ms);

/* ------------------------------------------------------------------ op-level entry points */
typedef struct foley_rowbcast {  /* row-broadcast operand (AdaLN shift/scale/gate, addend) */
  const float* p;       /* null => absent */
  int64_t ld;
  int32_t mode;         /* 0: one vector for all rows; 1: rows [cfg][clip][l] use operand row [cfg][l];
                         * 2: the operand has Ls rows per cfg and token l uses row min(floor((l+0.5)*Ls/L), Ls-1),
                         *    float32 as F.interpolate(mode="nearest-exact") computes it (hifi_foley.py:759-762) */
  int32_t rows_per_cfg, L;
  int32_t Ls;           /* mode 2 only */
  int32_t period;       /* mode 2 only: 0, or a power of two - the up-sampled sequence repeats with this period and the
                         * operand stores only `period` rows per cfg (row (nearest_exact(l) mod period)) */
  int32_t periodic_cfgs; /* mode 2 with period > 0: 0 = every cfg half is stored periodically; k > 0 = only the first k halves
                         * (`period` rows each), the others follow with all their Ls rows (ABI 11; fills the struct's padding) */
} foley_rowbcast;

/* Head split applied to a fused q/k/v (or cross-attention q) projection: per (row, head) RMSNorm
 * (norm_layers.py:36-52), interleaved RoPE (attn_layers.py:112-146) and the [clip, H, S_tot, 128]
 * layout of the attention operands; replaces `rearrange` + q_norm/k_norm + apply_rotary_emb of
 * hifi_foley.py:226-262 / 376-382.  Used by foley_gemm_desc.qkv (epilogue 7). */
typedef struct foley_qkv_split_desc {
  int32_t L, H, nK;              /* rows are [clip][l] with L tokens per clip; nK operands of H heads */
  const float* gain[3];          /* RMSNorm gain [128] per operand, null => copy only */
  const int32_t* pos[3];         /* RoPE position per token l, null => no rotation */
  void* dst[3];                  /* [clips, H, S_tot, 128] in out_dtype; see vt_pitch for the last one */
  int32_t out_dtype, vt_pitch;   /* vt_pitch > 0: last operand stored transposed [clips, H, 128, vt_pitch] */
  int32_t S_tot, tok_off;
  float eps;
  const float* cos_tab; const float* sin_tab;   /* [P, 64] */
  /* Optional (nK = 1, 16-bit operands): attention of the projected q rows against <= 96 cached keys in the same epilogue
   * (TwoStreamCABlock cross attention to the text, hifi_foley.py:271-319): attn_out [M, H*128] receives
   * softmax(q k^T / sqrt(128)) v per head, rows ordered like the GEMM's.  Keys attn_k [sets, H, attn_skv, 128], values
   * TRANSPOSED attn_vt [sets, H, 128, attn_pitch] (attn_pitch >= 96, finite beyond attn_skv), both in the operand dtype;
   * the set of row r is (r / L) / attn_bdiv.  The library takes the fused form on small grids only (its 64-row head-split
   * tile) and says so in *attn_fused (may be null): 1 = attn_out written, dst[0] untouched; 0 = plain head split into
   * dst[0], the caller runs foley_op_attention itself. */
  const void* attn_k; const void* attn_vt; void* attn_out;
  int32_t attn_skv, attn_pitch, attn_bdiv;
  int32_t* attn_fused;
} foley_qkv_split_desc;

typedef struct foley_gemm_desc {
  const void* A; const void* W; const float* bias;
  int32_t M, N, K; int64_t lda;
  int32_t segV, segS, taps, tapC, dil, tap0;      /* conv-as-GEMM addressing */
  void* out0; void* out1;
  int32_t osegV; int64_t out_seg, out_row, out_shift; int32_t out_check;
  foley_rowbcast rb; const float* res; const float* alpha; int32_t alphaC;
  int32_t dtype;   /* operand dtype */
  int32_t epilogue;/* 0 store f32, 1 store T, 2 silu T, 3 gelu-tanh T, 4 silu-gate T, 5 gated residual, 6 DAC, 7 head split */
  int32_t tile;    /* 0 auto */
  int32_t ksplit;  /* gated-residual epilogue: K ranges (0 auto, 1 deterministic); ranges are combined with
                    * fp32 atomics, or - when `partials` is set - deferred to the next LayerNorm */
  /* Deferred split-K: K range s stores its raw product to partials[s][M][N] (fp32, 16-byte aligned,
   * room for partial_slabs ranges; caps ksplit) and out0 is NOT touched; the caller then runs
   * foley_op_ln_mod_pending(out0, ..., partials, *ksplit_used, bias, gate), which performs
   * x += gate * (sum_s partials[s] + bias) before normalising.  With *ksplit_used == 1 the GEMM
   * has already updated out0 and nothing is pending. */
  float* partials; int32_t partial_slabs; int32_t* ksplit_used;
  /* epilogue 7 (fused head split): N = nK*H*128, out0 unused, results go to qkv->dst[] */
  const foley_qkv_split_desc* qkv;
  int32_t rstride; /* source rows advanced per virtual row (strided conv, dac.py:55-61); 0 or 1 = dense */
  int64_t ldw;     /* elements between rows of W (0 = K); > K for row-padded weight storage (wave-specialised tiles) */
  int32_t wfmt;    /* storage of W: 0 = `dtype`; 1 = fp8 e4m3fn, 2 = fp8 e5m2 with bf16 activations - widened to bf16
                    * in registers by the wave-specialised tiles (15, 19), bit-identical to widening at load time */
  int32_t partial_dtype; /* dtype of the `partials` slabs: 0 = fp32; or the (16-bit) operand dtype - half the slab traffic, the
                          * sum of k rounded partials carries about the error of one rounding of the total; pass the same
                          * value to foley_op_ln_mod_pending2 */
  int32_t gelu_erf;/* epilogue 3 only: 1 = exact GELU (erf) instead of the tanh form - nn.GELU() of the conditioning
                    * encoders (reference models/synchformer/vit_helper.py:108-125 Mlp, nn.TransformerEncoderLayer) */
} foley_gemm_desc;

int foley_op_gemm(const foley_gemm_desc* d, void* stream);
/* in_dtype f32: q,k,v [B,H,S,128] fp32 (exact fp32 MFMA).  in_dtype bf16: q,k bf16 [B,H,S,128] and v
 * TRANSPOSED bf16 [B,H,128,vt_pitch], vt_pitch >= Skv rounded up to 32 with a finite pad. */
int foley_op_attention(const void* q, const void* k, const void* v, int in_dtype, int vt_pitch, int Bq, int H,
                       int Sq, int Skv, int kv_bdiv, void* outA, void* outB, int split, int out_dtype,
                       void* stream);
/* the same with head_dim 128 or 64 (the ViT-B conditioning encoders, feature_utils.py:63-108): fp32 operands, or 16-bit
 * operands through the LDS-staged 128-query kernel (v transposed [B,H,head_dim,vt_pitch]) */
int foley_op_attention_hd(const void* q, const void* k, const void* v, int in_dtype, int vt_pitch, int Bq, int H,
                          int Sq, int Skv, int kv_bdiv, void* outA, void* outB, int split, int out_dtype,
                          int head_dim, void* stream);
/* foley_op_attention_hd at head_dim 64 with scattered output rows: query t of group g lands in row out_rows[g*Sq + t] of out
 * [rows, H*64] - normally the table the queries were gathered by (foley_op_qkv_regroup's idx_q), so that the CLS attention and the
 * time / space group attention of a DividedAttention layer (vit_helper.py:37-105: `torch.cat((cls_out, x), dim=1)` after the inverse
 * rearrange) write ONE token-major buffer, ready for the output projection.
 * grp_q / grp_kv > 0 (16-bit operands): block-diagonal attention - the sequence is a pack of small groups, query t attends keys
 * [g*grp_kv, (g+1)*grp_kv) with g = t / grp_q only (DividedAttention over time: 8 frame queries x (CLS + 8 frame keys) per location;
 * 14 locations share one 128-query workgroup instead of taking one each); 0 / 0: every query sees all Skv keys.
 * out has out_nrows rows: a table entry outside [0, out_nrows) drops that query's output instead of writing out of bounds. */
int foley_op_attention_scatter(const void* q, const void* k, const void* v, int in_dtype, int vt_pitch, int G, int H, int Sq,
                               int Skv, int grp_q, int grp_kv, const int32_t* out_rows, void* out, int out_nrows, int out_dtype, void* stream);
/* Token regrouping between a fused q/k/v projection and foley_op_attention_hd at head_dim 64 - the conditioning encoders
 * (reference models/synchformer/vit_helper.py:37-105 DividedAttention: patch tokens attend over the frames of their location or
 * the locations of their frame with the CLS key / value prepended; transformers' SiglipAttention / ClapTextSelfAttention head split):
 * qkv [rows, 3*H*64] in nn.Linear(dim, 3*dim)'s (K H D) packing; group g reads source rows idx_q[g*Sq + t] as its queries and
 * idx_kv[g*Skv + t] as its keys / values and writes q [G,H,Sq,64], k [G,H,Skv,64] and v [G,H,Skv,64] (vt_pitch 0) or - 16-bit
 * operands - v TRANSPOSED [G,H,64,vt_pitch] with zeros beyond Skv (vt_pitch a multiple of 8 in [Skv, ceil64(Skv)]).
 * qkv has n_rows rows: table entries are clamped to [0, n_rows) on the device, so a bad table cannot read out of bounds. */
int foley_op_qkv_regroup(const void* qkv, int n_rows, int dtype, int H, const int32_t* idx_q, int G, int Sq, const int32_t* idx_kv, int Skv,
                         void* q, void* k, void* v, int vt_pitch, void* stream);
/* One pass of the frames' antialiased bicubic uint8 resize - replaces torchvision's v2.Resize(interpolation=BICUBIC, antialias=True)
 * on uint8 tensors in the nodes' pre-processing (reference nodes.py:184-196, utils.py:262-283; on the CPU that dispatches to ATen's
 * native separable uint8 kernel: horizontal pass, uint8 intermediate, vertical pass).  in [outer, len_in, inner] -> out [outer,
 * len_out, inner] along the middle axis: out = sat8((2^(precision-1) + sum_{j < xsize[x]} in[xmin[x] + j] * weights[x*kmax + j])
 * >> precision).  inner = 1: horizontal pass over rows; inner = W: vertical pass.  Tables (device memory, one row per output
 * sample; host/encoders.py::aa_tables builds them as ATen does - double-precision cubic (a = -0.5) taps over a support of
 * 2*max(scale, 1), normalised, rounded half away from zero at the largest precision whose biggest weight fits int16). */
int foley_op_resize_aa_u8(const uint8_t* in, long outer, int len_in, long inner, int len_out, const int32_t* xmin,
                          const int32_t* xsize, const int16_t* weights, int kmax, int precision, uint8_t* out, void* stream);
int foley_op_ln_mod(const float* x, int M, int D, float eps, const foley_rowbcast* shift,
                    const foley_rowbcast* scale, void* out, int out_dtype, void* stream);
/* LayerNorm (+ modulation) of a residual stream that first receives the pending update of a deferred
 * split-K gated-residual GEMM (see foley_gemm_desc.partials); x is updated in place. */
int foley_op_ln_mod_pending(float* x, int M, int D, float eps, const foley_rowbcast* shift,
                            const foley_rowbcast* scale, void* out, int out_dtype, const float* partials,
                            int k, const float* bias, const foley_rowbcast* gate, void* stream);
/* vt_pitch > 0: the last operand is written transposed [clips, H, 128, vt_pitch] (see above). */
/* the same with slabs of `partial_dtype` (0 = fp32, or out_dtype when that is bf16 / fp16: foley_gemm_desc.partial_dtype) */
int foley_op_ln_mod_pending2(float* x, int M, int D, float eps, const foley_rowbcast* shift,
                             const foley_rowbcast* scale, void* out, int out_dtype, const void* partials,
                             int partial_dtype, int k, const float* bias, const foley_rowbcast* gate, void* stream);
int foley_op_qkv_split(const float* qkv, int M, int L, int H, int nK, const float* const* gain,
                       const int32_t* const* pos, void* const* dst, int out_dtype, int vt_pitch, int S_tot,
                       int tok_off, float eps, const float* cos_tab, const float* sin_tab, void* stream);
int foley_op_solver_step(const float* pred, float* x, float* x_saved, float* d_acc, int clips, int C, int L,
                         int ncfg, float guidance, const float* coef, int32_t* step_ptr, void* rows_out,
                         int rows_dtype, void* stream);
int foley_op_latent_rows(const float* x, int clips, int C, int L, int ncfg, void* out, int out_dtype,
                         void* stream);
int foley_op_dac_out(const float* s, const float* w, const float* bias, int B, int T, int C, float* out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FOLEY_HIP_H */
