"""The conditioning encoders of the Video-to-Audio path ON THE HIP ENGINE (stage 2).

`host/encoders.py` restates the Synchformer visual extractor on PyTorch-ROCm library kernels; here the same
networks run on libfoley_hip.so: every linear layer / patch embedding through `foley_op_gemm` (fused bias, exact-GELU
and residual epilogues), every attention through `foley_op_attention_hd` (head dim 64: the fp32 MFMA kernel in parity
mode, the LDS-staged 128-query kernel for fp16 / bf16), every LayerNorm through `foley_op_ln_mod` (the affine
parameters ride in its shift / scale operands).  PyTorch only re-views / permutes tensors between the ops (token
grouping of the divided space-time attention, the CLS key / value prepended to every group) - data movement, no math.

Reference: /root/reference/hunyuanvideo_foley/utils/feature_utils.py:63-108 (the two encode loops; Synchformer under
fp16 autocast), models/synchformer/{synchformer.py:44-50, motionformer.py:214-250, video_model_builder.py:166-256,
vit_helper.py:37-170}; SigLIP2 is `transformers`' SiglipVisionModel (google/siglip2-base-patch16-512: ViT-B/16 at
512 px, 1024 tokens, attention-pooling head), restated here over its state dict.

There is no fallback: these functions raise when the library is missing (runtime.load_library)."""
from __future__ import annotations

from typing import Tuple, Dict, Optional

import torch

from . import runtime as rt

Tensor = torch.Tensor
SD = Dict[str, Tensor]
HEADS, HD = 12, 64


class _Engine:
    """Weights of one encoder staged for the engine: matrices in the compute dtype [N, K] (K contiguous), vectors fp32,
    LayerNorm affine pairs as (shift = bias, scale = weight - 1) for foley_op_ln_mod."""

    def __init__(self, device, dtype: torch.dtype):
        self.dev, self.dtype = torch.device(device), dtype
        self.half = dtype in (torch.float16, torch.bfloat16)
        self.mats: Dict[str, Tensor] = {}
        self.vecs: Dict[str, Tensor] = {}
        self.ones: Dict[int, Tensor] = {}
        self.tabs: Dict = {}

    def mat(self, key: str, w: Tensor) -> Tensor:
        t = self.mats.get(key)
        if t is None:
            t = self.mats[key] = w.detach().reshape(w.shape[0], -1).to(self.dev, self.dtype).contiguous()
        return t

    def vec(self, key: str, v: Tensor, minus_one: bool = False) -> Tensor:
        k = key + ("#m1" if minus_one else "")
        t = self.vecs.get(k)
        if t is None:
            t = v.detach().reshape(-1).to(self.dev, torch.float32)
            t = self.vecs[k] = (t - 1.0 if minus_one else t).contiguous()
        return t

    def table(self, key: str, w: Tensor) -> Tensor:
        """An embedding table staged once on the device in its OWN dtype (rows are widened after the lookup)."""
        t = self.mats.get(key + "#tab")
        if t is None:
            t = self.mats[key + "#tab"] = w.detach().to(self.dev).contiguous()
        return t

    def fused(self, sd: SD, key: str, wkeys, bkeys) -> SD:
        """Row-concatenation of several linear layers that read the same input (q / k / v projections kept as separate modules
        by the HF encoders) staged once as ONE matrix / bias under `key`: one GEMM launch and one read of the activations
        instead of three.  Returns the small dict `linear` looks the staged tensors up in."""
        if key + ".w" not in self.mats:
            self.mats[key + ".w"] = torch.cat([sd[k].detach().reshape(sd[k].shape[0], -1) for k in wkeys]).to(self.dev, self.dtype).contiguous()
            self.vecs[key + ".b"] = torch.cat([sd[k].detach().reshape(-1) for k in bkeys]).to(self.dev, torch.float32).contiguous()
        return {key + ".w": self.mats[key + ".w"], key + ".b": self.vecs[key + ".b"]}

    def one(self, n: int) -> Tensor:
        t = self.ones.get(n)
        if t is None:
            t = self.ones[n] = torch.ones(n, device=self.dev, dtype=torch.float32)
        return t

    # ---- ops
    def ln(self, x: Tensor, sd: SD, key: str, eps: float = 1e-6, out_dtype: Optional[torch.dtype] = None) -> Tensor:
        """nn.LayerNorm(D, eps) with affine parameters: x fp32 [M, D] -> compute dtype (or `out_dtype`) [M, D]."""
        out = torch.empty(x.shape, device=self.dev, dtype=out_dtype or self.dtype)
        rt.op_ln_mod(x, eps, rt.rowbcast(self.vec(key + ".bias", sd[key + ".bias"])),
                     rt.rowbcast(self.vec(key + ".weight", sd[key + ".weight"], minus_one=True)), out)
        return out

    def linear(self, a: Tensor, sd: SD, wkey: str, bkey: Optional[str], act: Optional[str] = None,
               out_f32: bool = False) -> Tensor:
        """a [M, K] (compute dtype) @ W^T + b; act None | 'gelu_erf' | 'gelu_tanh'."""
        W = self.mat(wkey, sd[wkey])
        b = self.vec(bkey, sd[bkey]) if bkey else None
        M, N = a.shape[0], W.shape[0]
        if out_f32:
            out = torch.empty(M, N, device=self.dev, dtype=torch.float32)
            rt.op_gemm(a, W, b, out0=out)
        elif act:
            out = torch.empty(M, N, device=self.dev, dtype=self.dtype)
            rt.op_gemm(a, W, b, out0=out, epilogue=rt.EPI_GELU_T, gelu_erf=(act == "gelu_erf"))
        else:
            out = torch.empty(M, N, device=self.dev, dtype=self.dtype)
            rt.op_gemm(a, W, b, out0=out, epilogue=rt.EPI_STORE_T)
        return out

    def linear_residual(self, x: Tensor, a: Tensor, sd: SD, wkey: str, bkey: Optional[str]):
        """x (fp32 residual stream, in place) += a @ W^T + b - the gated-residual epilogue with a gate of ones."""
        W = self.mat(wkey, sd[wkey])
        b = self.vec(bkey, sd[bkey]) if bkey else None
        rt.op_gemm(a, W, b, out0=x, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(self.one(W.shape[0]), 0), ksplit=1)

    def index(self, key, build) -> Tensor:
        """int32 index table on the device, built once per (shape) key by `build()` (a CPU / torch expression)."""
        t = self.tabs.get(key)
        if t is None:
            if len(self.tabs) >= 512:          # prompt-length keyed tables (CLAP): bounded, oldest first
                self.tabs.pop(next(iter(self.tabs)))
            t = self.tabs[key] = build().to(self.dev, torch.int32).contiguous()
        return t

    def attention_regrouped(self, qkv: Tensor, heads: int, idx_q: Tensor, idx_kv: Tensor, out: Optional[Tensor] = None,
                            grp: Tuple[int, int] = (0, 0)) -> Tensor:
        """Attention straight from a fused projection qkv [rows, 3*H*64]: ONE regroup launch (foley_op_qkv_regroup: head split,
        token gather per group, transposed V for the 16-bit kernels) + the attention -> [G, Sq, H*64] token-major; with `out`
        [rows, H*64] every query is written back to the row it was gathered from (foley_op_attention_scatter) and `out` returns."""
        q, k, v = rt.op_qkv_regroup(qkv, heads, idx_q, idx_kv)
        G, H, Sq, hd = q.shape
        if out is not None:     # grp = (queries, keys) per packed group: block-diagonal attention inside each table row
            rt.op_attention_scatter(q, k, v, idx_q, out, grp[0], grp[1])
            return out
        out = torch.empty(G, Sq, H * hd, device=self.dev, dtype=self.dtype)
        rt.op_attention(q, k, v, out, out, 0)
        return out

    def attention(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
        """q [G, H, Sq, 64], k / v [G, H, Skv, 64] (compute dtype; any strides) -> [G, Sq, H*64] token-major."""
        G, H, Sq, hd = q.shape
        Skv = k.shape[2]
        q, k = q.contiguous(), k.contiguous()
        if self.half:       # the 16-bit kernels read V transposed [G, H, hd, pitch], finite pad
            pitch = (Skv + 31) // 32 * 32
            vt = torch.zeros(G, H, hd, pitch, device=self.dev, dtype=self.dtype)
            vt[..., :Skv] = v.transpose(2, 3)
            v = vt
        else:
            v = v.contiguous()
        out = torch.empty(G, Sq, H * hd, device=self.dev, dtype=self.dtype)
        rt.op_attention(q, k, v, out, out, 0)
        return out


def _divided_attention(E: _Engine, h: Tensor, sd: SD, key: str, B: int, frames: int, space: int, over: str) -> Tensor:
    """vit_helper.DividedAttention (vit_helper.py:37-105) on the engine: h [B*N, D] (normed tokens, compute dtype) ->
    attention output [B*N, D] before the projection.  The CLS query attends to every token; patch tokens attend, with
    the CLS key / value prepended, across the frames of their location (over='time') or the locations of their frame.
    The token grouping is an index table (built once per shape) consumed by foley_op_qkv_regroup - no torch copies."""
    N = 1 + frames * space
    qkv = E.linear(h, sd, key + ".qkv.weight", key + ".qkv.bias")                               # [B*N, 3*H*64]
    base = lambda: (torch.arange(B) * N)[:, None]
    all_rows = E.index(("all", B, N), lambda: base() + torch.arange(N)[None])                   # [B, N]
    cls_rows = E.index(("cls", B, N), lambda: base().clone())                                   # [B, 1]
    out = torch.empty(B * N, HEADS * HD, device=E.dev, dtype=E.dtype)     # both attentions scatter into the token-major layer output
    E.attention_regrouped(qkv, HEADS, cls_rows, all_rows, out=out)                              # rows b*N
    if over == "time":      # groups = (batch, location), sequence = frames
        def build():
            tok = 1 + torch.arange(frames)[None, :] * space + torch.arange(space)[:, None]     # [space, frames]
            return ((torch.arange(B) * N)[:, None, None] + tok[None]).reshape(B * space, frames)
    else:                   # groups = (batch, frame), sequence = locations
        def build():
            tok = 1 + torch.arange(frames)[:, None] * space + torch.arange(space)[None, :]     # [frames, space]
            return ((torch.arange(B) * N)[:, None, None] + tok[None]).reshape(B * frames, space)
    G = space if over == "time" else frames
    iq = E.index((over, "q", B, frames, space), build)
    ikv = E.index((over, "kv", B, frames, space),
                  lambda: torch.cat(((torch.arange(B) * N).repeat_interleave(G)[:, None], build()), dim=1))   # CLS key / value first
    if over == "time" and E.half and frames <= 64:
        # 8 queries x 9 keys per location: one 128-query workgroup per group ran 6 % full (133 us per layer).  The tables are free
        # to PACK groups: p locations per table row (p | B*space, p*frames <= 128) and a block-diagonal mask in the kernel
        # (foley_op_attention_scatter grp_q / grp_kv) - same arithmetic per query, 14x fewer workgroups
        n = B * space
        p = max(d for d in range(1, 128 // frames + 1) if n % d == 0)
        if p > 1:
            iq_p = E.index((over, "q-pack", B, frames, space, p), lambda: build().reshape(n // p, p * frames))
            ikv_p = E.index((over, "kv-pack", B, frames, space, p),
                            lambda: torch.cat(((torch.arange(B) * N).repeat_interleave(G)[:, None], build()), dim=1).reshape(n // p, p * (frames + 1)))
            return E.attention_regrouped(qkv, HEADS, iq_p, ikv_p, out=out, grp=(frames, frames + 1))
    return E.attention_regrouped(qkv, HEADS, iq, ikv, out=out)      # the inverse rearrange + torch.cat((cls_out, x), 1) of the reference = the scatter


def synchformer_segments_hip(sd: SD, x: Tensor, dtype: torch.dtype = torch.float16, prefix: str = "vfeat_extractor.",
                             engine: Optional[_Engine] = None) -> Tensor:
    """Synchformer.forward on the HIP engine: x [S, 16, 3, 224, 224] fp32 pre-processed frames on the GPU -> [S, 8, 768]
    fp32.  dtype = GEMM / attention operand type: float16 is what the reference's GPU path computes in (fp16 autocast,
    feature_utils.py:99-104); float32 is the parity mode the golden g11 (the reference's fp32 CPU run) is gated in."""
    p = prefix
    E = engine or _Engine(x.device, dtype)
    S = x.shape[0]
    w = sd[p + "patch_embed_3d.proj.weight"]                        # [D, 3, zt, ph, pw]
    D, C3, zt, ph, pw = w.shape
    T, Hh, Ww = x.shape[1], x.shape[3], x.shape[4]
    frames, gh, gw = T // zt, Hh // ph, Ww // pw
    space = gh * gw
    # patch embedding = conv3d with stride == kernel: a GEMM over the unfolded voxels (K order c, zt, ph, pw = the weight's)
    a = x.view(S, frames, zt, C3, gh, ph, gw, pw).permute(0, 1, 4, 6, 3, 2, 5, 7).reshape(S * frames * space, C3 * zt * ph * pw)
    tok = E.linear(a.to(E.dtype), sd, p + "patch_embed_3d.proj.weight", p + "patch_embed_3d.proj.bias", out_f32=True)
    pos, temp = sd[p + "pos_embed"].to(E.dev, torch.float32), sd[p + "temp_embed"].to(E.dev, torch.float32)
    total = pos[:, 1:].repeat(1, frames, 1) + temp.repeat_interleave(space, dim=1)
    N = 1 + frames * space
    xs = torch.cat((sd[p + "cls_token"].to(E.dev, torch.float32).expand(S, -1, -1), tok.view(S, frames * space, D)), dim=1)
    xs = (xs + torch.cat((pos[:, :1], total), dim=1)).reshape(S * N, D).contiguous()         # fp32 residual stream
    depth = 1 + max(int(k[len(p) + 7:].split(".")[0]) for k in sd if k.startswith(p + "blocks."))
    for i in range(depth):                                           # DividedSpaceTimeBlock (vit_helper.py:150-170)
        b = f"{p}blocks.{i}"
        att = _divided_attention(E, E.ln(xs, sd, b + ".norm3"), sd, b + ".timeattn", S, frames, space, "time")
        E.linear_residual(xs, att, sd, b + ".timeattn.proj.weight", b + ".timeattn.proj.bias")
        att = _divided_attention(E, E.ln(xs, sd, b + ".norm1"), sd, b + ".attn", S, frames, space, "space")
        E.linear_residual(xs, att, sd, b + ".attn.proj.weight", b + ".attn.proj.bias")
        hid = E.linear(E.ln(xs, sd, b + ".norm2"), sd, b + ".mlp.fc1.weight", b + ".mlp.fc1.bias", act="gelu_erf")
        E.linear_residual(xs, hid, sd, b + ".mlp.fc2.weight", b + ".mlp.fc2.bias")
    # CLS dropped, final norm, then one nn.TransformerEncoderLayer(norm_first, GELU) per (segment, frame) with its own CLS
    body = xs.view(S, N, D)[:, 1:].reshape(S * frames * space, D).contiguous()
    normed = torch.empty_like(body)
    rt.op_ln_mod(body, 1e-6, rt.rowbcast(E.vec(p + "norm.bias", sd[p + "norm.bias"])),
                 rt.rowbcast(E.vec(p + "norm.weight", sd[p + "norm.weight"], minus_one=True)), normed)
    a_ = p + "spatial_attn_agg"
    G, L = S * frames, 1 + space
    y = torch.cat((sd[a_ + ".cls_token"].to(E.dev, torch.float32).expand(G, -1, -1), normed.view(G, space, D)), dim=1)
    y = y.reshape(G * L, D).contiguous()
    rows = E.index(("all", G, L), lambda: (torch.arange(G) * L)[:, None] + torch.arange(L)[None])
    att = E.attention_regrouped(E.linear(E.ln(y, sd, a_ + ".norm1"), sd, a_ + ".self_attn.in_proj_weight", a_ + ".self_attn.in_proj_bias"),
                                HEADS, rows, rows).reshape(G * L, D)
    E.linear_residual(y, att, sd, a_ + ".self_attn.out_proj.weight", a_ + ".self_attn.out_proj.bias")
    hid = E.linear(E.ln(y, sd, a_ + ".norm2"), sd, a_ + ".linear1.weight", a_ + ".linear1.bias", act="gelu_erf")
    E.linear_residual(y, hid, sd, a_ + ".linear2.weight", a_ + ".linear2.bias")
    return y.view(G, L, D)[:, 0].reshape(S, frames, D).clone()       # temp_attn_agg is Identity


def encode_video_with_sync_hip(sd: SD, frames: Tensor, dtype: torch.dtype = torch.float16, batch_size: int = 32) -> Tensor:
    """frames [T, 3, 224, 224] fp32 (25 fps, pre-processed, on the GPU) -> [1, num_segments*8, 768] fp32
    (feature_utils.py:80-108: 16-frame segments with stride 8)."""
    from .encoders import SYNC_SEGMENT, SYNC_STRIDE
    T = frames.shape[0]
    n_seg = (T - SYNC_SEGMENT) // SYNC_STRIDE + 1
    if n_seg < 1:
        raise ValueError(f"Synchformer needs at least {SYNC_SEGMENT} frames at 25 fps (got {T})")
    if not frames.is_cuda:
        raise rt.FoleyRuntimeError("the HIP-engine Synchformer needs the frames on the GPU")
    segs = torch.stack([frames[i * SYNC_STRIDE:i * SYNC_STRIDE + SYNC_SEGMENT] for i in range(n_seg)]).float()
    E = _engine_for(sd, frames.device, dtype)
    outs = [synchformer_segments_hip(sd, segs[i:i + batch_size].contiguous(), dtype, engine=E)
            for i in range(0, n_seg, batch_size)]
    return torch.cat(outs).reshape(1, n_seg * 8, -1)


_ENGINES: Dict = {}


def _engine_for(sd: SD, device, dtype) -> _Engine:
    """Staged weights are cached per (state dict object, device, dtype)."""
    key = (id(sd), str(device), dtype)
    e = _ENGINES.get(key)
    if e is None or e.sd_ref is not sd:
        if len(_ENGINES) >= 4:
            _ENGINES.pop(next(iter(_ENGINES)))
        e = _ENGINES[key] = _Engine(device, dtype)
        e.sd_ref = sd
    return e


# ----------------------------------------------------------------------------- SigLIP2 vision tower
def siglip_image_features_hip(sd: SD, pixels: Tensor, dtype: torch.dtype = torch.bfloat16, prefix: str = "vision_model.",
                              heads: int = HEADS, eps: float = 1e-6, batch_size: int = 64) -> Tensor:
    """`SiglipModel.get_image_features` (the pooled output of transformers' SiglipVisionTransformer) on the HIP engine,
    over the model's state dict: patch embedding (conv k = stride = patch -> GEMM over unfolded patches) + learned
    position table, pre-norm encoder layers (LayerNorm -> q/k/v -> attention -> out_proj -> +x; LayerNorm -> fc1 ->
    GELU-tanh -> fc2 -> +x), post LayerNorm, and the attention-pooling head (a learned probe attending to all tokens
    through nn.MultiheadAttention, then LayerNorm -> MLP residual).  pixels [T, 3, S, S] fp32 on the GPU -> [T, D] fp32.
    Called by feature_utils.py:63-78's counterpart (encoders.encode_video_with_siglip2) when the engine path is on."""
    p = prefix if (prefix + "embeddings.patch_embedding.weight") in sd else ""     # SiglipModel vs SiglipVisionModel state dicts
    if pixels.shape[0] > batch_size:
        return torch.cat([siglip_image_features_hip(sd, pixels[i:i + batch_size].contiguous(), dtype, prefix, heads, eps, batch_size)
                          for i in range(0, pixels.shape[0], batch_size)])
    E = _engine_for(sd, pixels.device, dtype)
    w = sd[p + "embeddings.patch_embedding.weight"]                   # [D, 3, P, P]
    D, C3, P, _ = w.shape
    hd = D // heads
    if hd != HD:
        raise rt.FoleyRuntimeError("the engine's encoder attention serves head_dim 64")
    B, _, Hh, Ww = pixels.shape
    gh, gw = Hh // P, Ww // P
    N = gh * gw
    a = pixels.view(B, C3, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * N, C3 * P * P)
    x = E.linear(a.to(E.dtype), sd, p + "embeddings.patch_embedding.weight", p + "embeddings.patch_embedding.bias", out_f32=True)
    x = (x.view(B, N, D) + sd[p + "embeddings.position_embedding.weight"].to(E.dev, torch.float32)[None, :N]).reshape(B * N, D).contiguous()
    depth = 1 + max(int(k[len(p) + 15:].split(".")[0]) for k in sd if k.startswith(p + "encoder.layers."))
    for i in range(depth):
        l = f"{p}encoder.layers.{i}"
        h = E.ln(x, sd, l + ".layer_norm1", eps)
        fq = E.fused(sd, l + ".self_attn.qkv#", [l + f".self_attn.{n}_proj.weight" for n in "qkv"], [l + f".self_attn.{n}_proj.bias" for n in "qkv"])
        rows = E.index(("all", B, N), lambda: (torch.arange(B) * N)[:, None] + torch.arange(N)[None])
        att = E.attention_regrouped(E.linear(h, fq, l + ".self_attn.qkv#.w", l + ".self_attn.qkv#.b"), heads, rows, rows).reshape(B * N, D)   # one [B*N, 3*D] projection
        E.linear_residual(x, att, sd, l + ".self_attn.out_proj.weight", l + ".self_attn.out_proj.bias")
        hid = E.linear(E.ln(x, sd, l + ".layer_norm2", eps), sd, l + ".mlp.fc1.weight", l + ".mlp.fc1.bias", act="gelu_tanh")
        E.linear_residual(x, hid, sd, l + ".mlp.fc2.weight", l + ".mlp.fc2.bias")
    hs = E.ln(x, sd, p + "post_layernorm", eps)                        # [B*N, D] compute dtype
    # attention-pooling head: probe [1, 1, D] as the single query of nn.MultiheadAttention over the tokens
    hp = p + "head"
    Wi, bi = sd[hp + ".attention.in_proj_weight"], sd[hp + ".attention.in_proj_bias"]
    probe = sd[hp + ".probe"].to(E.dev, E.dtype).reshape(1, D)
    sdh = {hp + "#wq": Wi[:D], hp + "#bq": bi[:D], hp + "#wkv": Wi[D:], hp + "#bkv": bi[D:]}    # staged once under these keys
    qp = E.linear(probe, sdh, hp + "#wq", hp + "#bq").view(1, 1, heads, hd).permute(0, 2, 1, 3).expand(B, heads, 1, hd)
    kv = E.linear(hs, sdh, hp + "#wkv", hp + "#bkv").view(B, N, 2, heads, hd).permute(2, 0, 3, 1, 4)
    pooled = E.attention(qp, kv[0], kv[1]).reshape(B, D)
    y = E.linear(pooled, sd, hp + ".attention.out_proj.weight", hp + ".attention.out_proj.bias", out_f32=True)
    hid = E.linear(E.ln(y, sd, hp + ".layernorm", eps), sd, hp + ".mlp.fc1.weight", hp + ".mlp.fc1.bias", act="gelu_tanh")
    E.linear_residual(y, hid, sd, hp + ".mlp.fc2.weight", hp + ".mlp.fc2.bias")
    return y


# ----------------------------------------------------------------------------- CLAP text encoder
def clap_text_hidden_hip(sd: SD, input_ids: Tensor, attention_mask: Tensor, dtype: torch.dtype = torch.bfloat16,
                         prefix: str = "text_model.", heads: int = HEADS, eps: float = 1e-12, pad_id: int = 1) -> Tensor:
    """`ClapTextModelWithProjection(...).last_hidden_state` (what feature_utils.py:133-138 feeds the DiT as text tokens) on the
    HIP engine, over the HF model's state dict: RoBERTa embeddings (word + token-type + learned positions counted over the
    non-pad tokens) -> LayerNorm, then post-norm encoder layers  y = LN(x + dense(attn(x)));  x' = LN(y + dense(GELU(dense(y)))).
    The tokenizer pads on the right, so the additive key mask of the reference equals "prompt i attends to its first len_i
    keys": attention runs per prompt with Skv = len_i and all T (padded) query rows - the pad rows are part of
    `last_hidden_state` and reach the DiT like in the reference.  input_ids / attention_mask [B, T] on the GPU -> [B, T, D] fp32."""
    p = prefix if (prefix + "embeddings.word_embeddings.weight") in sd else ""
    E = _engine_for(sd, input_ids.device, dtype)
    B, T = input_ids.shape
    mask = attention_mask.to(torch.long)
    lens = mask.sum(1)
    if not bool((mask == (torch.arange(T, device=mask.device)[None] < lens[:, None]).long()).all()):
        raise rt.FoleyRuntimeError("the engine's CLAP text encoder expects right-padded prompts")
    nz = input_ids.ne(pad_id).long()
    pos_ids = torch.cumsum(nz, dim=1) * nz + pad_id                      # create_position_ids_from_input_ids
    tab = lambda k: E.table(p + k, sd[p + k])                            # rows are picked first, then widened (the word table is 154 MB in fp32)
    emb = tab("embeddings.word_embeddings.weight")[input_ids].float() + tab("embeddings.token_type_embeddings.weight")[0].float() \
        + tab("embeddings.position_embeddings.weight")[pos_ids].float()
    D = emb.shape[-1]
    hd = D // heads
    if hd != HD:
        raise rt.FoleyRuntimeError("the engine's encoder attention serves head_dim 64")
    x = E.ln(emb.reshape(B * T, D).contiguous(), sd, p + "embeddings.LayerNorm", eps, out_dtype=torch.float32)
    depth = 1 + max(int(k[len(p) + 14:].split(".")[0]) for k in sd if k.startswith(p + "encoder.layer."))
    lens_h = [int(v) for v in lens.tolist()]
    for i in range(depth):
        l = f"{p}encoder.layer.{i}"
        xT = x.to(E.dtype)
        a_ = l + ".attention.self."
        fq = E.fused(sd, a_ + "qkv#", [a_ + n + ".weight" for n in ("query", "key", "value")], [a_ + n + ".bias" for n in ("query", "key", "value")])
        qkv = E.linear(xT, fq, a_ + "qkv#.w", a_ + "qkv#.b")                  # [B*T, 3*D]
        # per prompt: all T query rows against its first len_b keys - one regroup + one attention launch each (index tables per
        # (prompt slot, T, len); until round 5: slices, .contiguous() copies and a padded V^T per prompt, ~14 launches per layer)
        att = torch.empty(B * T, D, device=E.dev, dtype=E.dtype)
        for b in range(B):
            E.attention_regrouped(qkv, heads, E.index(("clap-q", b, T), lambda b=b: (b * T + torch.arange(T))[None]),
                                  E.index(("clap-kv", b, T, lens_h[b]), lambda b=b: (b * T + torch.arange(lens_h[b]))[None]), out=att)
        E.linear_residual(x, att.reshape(B * T, D), sd, l + ".attention.output.dense.weight", l + ".attention.output.dense.bias")
        y = E.ln(x, sd, l + ".attention.output.LayerNorm", eps, out_dtype=torch.float32)
        hid = E.linear(y.to(E.dtype), sd, l + ".intermediate.dense.weight", l + ".intermediate.dense.bias", act="gelu_erf")
        E.linear_residual(y, hid, sd, l + ".output.dense.weight", l + ".output.dense.bias")
        x = E.ln(y, sd, l + ".output.LayerNorm", eps, out_dtype=torch.float32)
    return x.view(B, T, D)
