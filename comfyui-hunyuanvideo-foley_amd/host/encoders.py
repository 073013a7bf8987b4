"""Conditioning encoders of the Video-to-Audio path: frame resampling, the two
pre-processing pipelines, SigLIP2 / CLAP through `transformers`, and the Synchformer visual
feature extractor as a functional restatement on PyTorch-ROCm.

This is the step BEFORE the HIP hot path: it produces the `[1, Lv, 768]` / `[1, Ls, 768]` /
`[1, T, 768]` conditioning tensors `foley_prepare` consumes.  What the reference does
(/root/reference):

  * nodes.py:290-320      pad / trim the IMAGE batch to int(duration * frame_rate) frames, uint8
                          [T,C,H,W], `linspace(0, n-1, int(duration*8 | *25)).long()` index selection;
  * nodes.py:184-196      torchvision-v2 pipelines: SigLIP2 = Resize((512,512), bicubic, antialias)
                          -> float/255 -> Normalize(0.5, 0.5); Synchformer = Resize(224) (short
                          edge) -> CenterCrop(224) -> same scaling;
  * utils.py:262-292      feature_process_from_tensors: SigLIP2 pooled image features per 8 fps
                          frame, Synchformer over 16-frame segments with stride 8 of the 25 fps
                          frames, CLAP last_hidden_state for [negative, positive];
  * feature_utils.py:63-108  the two encode loops (Synchformer under fp16 autocast);
  * models/synchformer/*  Synchformer.forward == MotionFormer(divided space-time ViT-B/16, 8x14x14
                          tokens of 2x16x16 voxels) -> LayerNorm -> one spatial aggregation layer
                          (nn.TransformerEncoderLayer, norm_first, CLS read-out) per frame pair.

torchvision is not a dependency here: the two pipelines are restated on `F.interpolate`
(bicubic + antialias on uint8 is the very kernel torchvision's v2.Resize dispatches to on CPU).
The Synchformer restatement is written against the checkpoint's state-dict keys
(`vfeat_extractor.*`) as plain tensor functions - no nn.Module tree, nothing vendored.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

SYNC_SEGMENT, SYNC_STRIDE = 16, 8          # feature_utils.py:91-93
FPS_SIGLIP, FPS_SYNC = 8, 25
_CHUNK_BYTES = 256 << 20                   # select_frames: float32 frames converted per contiguous slab of at most this size


# ----------------------------------------------------------------------------- frame selection
def select_frames(image: Tensor, duration: float, frame_rate: float, device=None) -> Tuple[Tensor, Tensor]:
    """IMAGE [N,H,W,C] float 0-1 -> uint8 [T,C,H,W] frames at 8 fps and at 25 fps (nodes.py:293-317):
    hold the last frame if the clip is shorter than duration*frame_rate, else cut; then pick
    `linspace(0, n-1, int(duration*fps)).long()`.

    The reference converts the WHOLE padded clip to uint8 on the CPU and then selects; the result only depends on the
    selected frames, so only the range they span is touched (an index beyond the clip's end is the held last frame) and -
    with `device` - it is copied to the GPU first and converted there: the same float32 multiply and truncating cast,
    0.5 s of CPU time less per 5 s clip."""
    total = image.shape[0]
    n = int(duration * frame_rate)
    i8 = torch.linspace(0, n - 1, int(duration * FPS_SIGLIP)).long().clamp_(max=total - 1)
    i25 = torch.linspace(0, n - 1, int(duration * FPS_SYNC)).long().clamp_(max=total - 1)
    lo, hi = int(min(i8.min(), i25.min())), int(max(i8.max(), i25.max()))
    # float32 frames are 12 bytes per pixel (25 MB at 1080p): the span is converted in bounded chunks - a contiguous slab of at
    # most ~256 MB goes to the device, is multiplied and cast there, and only the uint8 frames somebody selected are kept, so a
    # 30 s / 1080p clip needs megabytes next to the model instead of 20 GB.  `.to(torch.int32).to(torch.uint8)` wraps out-of-range
    # values like the reference's CPU `.byte()` does (a direct float -> uint8 cast saturates on the GPU).
    want = torch.unique(torch.cat((i8, i25)))
    per_frame = max(1, image[0].numel() * image.element_size())
    chunk = max(1, _CHUNK_BYTES // per_frame)
    kept, pos = [], {}
    for c0 in range(lo, hi + 1, chunk):
        c1 = min(hi + 1, c0 + chunk)
        sel = want[(want >= c0) & (want < c1)]
        if sel.numel() == 0:
            continue
        block = image[c0:c1]                       # a view: one contiguous host-to-device copy per chunk, no gather on the CPU
        if device is not None:
            block = block.to(device)
        u8 = (block * 255.0).to(torch.int32).to(torch.uint8) if block.is_cuda else (block * 255.0).byte()
        for j in sel.tolist():
            pos[j] = len(pos)
        kept.append(u8.index_select(0, (sel - c0).to(u8.device)))
    frames = torch.cat(kept).permute(0, 3, 1, 2)
    pick = lambda idx: frames.index_select(0, torch.tensor([pos[int(j)] for j in idx.tolist()], device=frames.device))
    return pick(i8), pick(i25)


# ----------------------------------------------------------------------------- pre-processing
_AA_TABLES: Dict = {}


def aa_tables(n_in: int, n_out: int):
    """Tap tables of ONE axis of the antialiased bicubic uint8 resize, as ATen's native uint8 kernel builds them (the kernel
    torchvision's v2.Resize dispatches to on CPU tensors - where the reference pre-processes, utils.py:262-283): output sample i
    sits at centre = scale*(i + 0.5), scale = n_in / n_out; its taps are the input samples within support = 2*max(scale, 1) of
    the centre, weighted by the Keys cubic (a = -0.5) of their distance / max(scale, 1), normalised to sum 1 - all in double
    precision - then stored as int16 at the largest precision (<= 22 bits) at which the biggest weight stays below 2^15,
    rounded half away from zero.  Returns (xmin int32 [n_out], xsize int32 [n_out], weights int16 [n_out, kmax], precision)
    as numpy arrays; cached per (n_in, n_out)."""
    import numpy as np
    if n_in < 1 or n_out < 1:
        raise ValueError(f"aa_tables: axis lengths must be positive (got {n_in} -> {n_out})")
    key = (int(n_in), int(n_out))
    hit = _AA_TABLES.get(key)
    if hit is not None:
        return hit
    scale = n_in / n_out
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    kmax = int(math.ceil(support)) * 2 + 1
    i = np.arange(n_out, dtype=np.float64)
    centre = scale * (i + 0.5)
    xmin = np.maximum((centre - support + 0.5).astype(np.int64), 0)
    xsize = np.clip(np.minimum((centre + support + 0.5).astype(np.int64), n_in) - xmin, 0, kmax)
    j = np.arange(kmax, dtype=np.float64)[None, :]
    x = np.abs((j + xmin[:, None] - centre[:, None] + 0.5) * inv)
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))
    w = np.where(j < xsize[:, None], w, 0.0)
    # the kernel sums the taps one by one in index order (not pairwise): keep that order for the normaliser
    tot = np.zeros(n_out, dtype=np.float64)
    for c in range(kmax):
        tot = tot + w[:, c]
    w = np.where(tot[:, None] != 0.0, w / np.where(tot == 0.0, 1.0, tot)[:, None], w)
    wmax = float(w.max()) if w.size else 0.0
    precision = 0
    while precision < 22 and int(0.5 + wmax * (1 << (precision + 1))) < (1 << 15):
        precision += 1
    v = w * float(1 << precision)
    wi = np.where(v < 0.0, np.trunc(v - 0.5), np.trunc(v + 0.5)).astype(np.int16)
    out = (xmin.astype(np.int32), xsize.astype(np.int32), wi, precision)
    if len(_AA_TABLES) >= 64:
        _AA_TABLES.pop(next(iter(_AA_TABLES)))
    _AA_TABLES[key] = out
    return out


_AA_DEVICE: Dict = {}


def _resize_pass_hip(x: Tensor, axis: int, n_out: int) -> Tensor:
    """One axis of the resize on libfoley_hip.so (foley_op_resize_aa_u8), tables staged once per (sizes, device)."""
    from . import runtime as rt
    n_in = x.shape[axis]
    key = (n_in, n_out, str(x.device))
    tabs = _AA_DEVICE.get(key)
    if tabs is None:
        xmin, xsize, w, prec = aa_tables(n_in, n_out)
        if len(_AA_DEVICE) >= 64:
            _AA_DEVICE.pop(next(iter(_AA_DEVICE)))
        tabs = _AA_DEVICE[key] = (torch.from_numpy(xmin).to(x.device), torch.from_numpy(xsize).to(x.device),
                                  torch.from_numpy(w).to(x.device), prec)
    return rt.op_resize_aa_u8(x, axis, n_out, tabs[0], tabs[1], tabs[2], tabs[3])


def _resize_u8(frames: Tensor, size: Tuple[int, int]) -> Tensor:
    """v2.Resize(bicubic, antialias=True) on uint8 [T,C,H,W] as the reference runs it - on the CPU (utils.py:262-283
    pre-processes before the encoders' `.to(device)`), where torchvision dispatches to ATen's native uint8 kernel.  That
    kernel is separable with a uint8 intermediate and fixed-point weights (PIL's scheme): a horizontal pass, rounded and
    saturated to a byte, then the vertical pass.  On the GPU the same two integer passes run on libfoley_hip.so
    (foley_op_resize_aa_u8 over `aa_tables`' taps) and reproduce the CPU result BIT FOR BIT
    (`tests/test_encoders_gpu.py`; until round 5 this branch went through two float32 F.interpolate passes: <= 1 grey level off
    on 0.6 % of the pixels and 3 ms per clip in ATen's generic antialias kernel) - a single 2-D float interpolation differs by
    tens of levels on textured frames because overshoots are not clamped between the passes."""
    if tuple(frames.shape[-2:]) == tuple(size):
        return frames
    if frames.device.type == "cpu":
        return F.interpolate(frames, size=size, mode="bicubic", antialias=True)
    x = frames.contiguous()
    if x.shape[-1] != size[1]:
        x = _resize_pass_hip(x, x.dim() - 1, size[1])
    if x.shape[-2] != size[0]:
        x = _resize_pass_hip(x, x.dim() - 2, size[0])
    return x


def _scale_normalize(frames_u8: Tensor) -> Tensor:
    """ToDtype(float32, scale=True) + Normalize(mean 0.5, std 0.5)."""
    return (frames_u8.to(torch.float32) / 255.0 - 0.5) / 0.5


def siglip2_preprocess(frames_u8: Tensor, size: int = 512) -> Tensor:
    """uint8 [T,3,H,W] -> float32 [T,3,size,size] (nodes.py:184-188)."""
    return _scale_normalize(_resize_u8(frames_u8, (size, size)))


def synchformer_preprocess(frames_u8: Tensor, size: int = 224) -> Tensor:
    """uint8 [T,3,H,W] -> float32 [T,3,224,224]: short edge to 224 keeping the aspect ratio
    (torchvision: long edge = int(size * long / short)), centre crop (nodes.py:191-196)."""
    h, w = frames_u8.shape[-2:]
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    x = _resize_u8(frames_u8, (nh, nw))
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    return _scale_normalize(x[..., top:top + size, left:left + size])


# ----------------------------------------------------------------------------- Synchformer (visual branch)
def _ln(x: Tensor, sd: SD, key: str, eps: float = 1e-6) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def _lin(x: Tensor, sd: SD, key: str) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _divided_attention(x: Tensor, sd: SD, key: str, frames: int, space: int, over: str, heads: int = 12) -> Tensor:
    """vit_helper.DividedAttention: the CLS token attends to every token; patch tokens attend, together
    with the CLS key/value, either across the `frames` time steps of their own location
    (over='time': 'b (f n) d -> (b n) f d') or across the `space` locations of their own frame
    (over='space': 'b (f n) d -> (b f) n d')."""
    B, N, D = x.shape
    hd = D // heads
    qkv = _lin(x, sd, key + ".qkv").view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)      # [3, B, H, N, hd]
    q, k, v = qkv[0], qkv[1], qkv[2]
    cls_out = F.scaled_dot_product_attention(q[:, :, :1], k, v)                          # [B, H, 1, hd]
    if over == "time":      # groups = (batch, head, location), sequence = frames
        def grp(t):
            return t[:, :, 1:].reshape(B, heads, frames, space, hd).permute(0, 1, 3, 2, 4)
        G = space
    else:                   # groups = (batch, head, frame), sequence = locations
        def grp(t):
            return t[:, :, 1:].reshape(B, heads, frames, space, hd)
        G = frames
    q_, k_, v_ = grp(q), grp(k), grp(v)                                                  # [B, H, G, s, hd]
    ck = k[:, :, None, :1].expand(B, heads, G, 1, hd)
    cv = v[:, :, None, :1].expand(B, heads, G, 1, hd)
    out = F.scaled_dot_product_attention(q_, torch.cat((ck, k_), dim=3), torch.cat((cv, v_), dim=3))
    if over == "time":
        out = out.permute(0, 1, 3, 2, 4)
    out = out.reshape(B, heads, frames * space, hd)
    out = torch.cat((cls_out, out), dim=2).permute(0, 2, 1, 3).reshape(B, N, D)
    return _lin(out, sd, key + ".proj")


def _encoder_layer(x: Tensor, sd: SD, key: str, heads: int = 12) -> Tensor:
    """nn.TransformerEncoderLayer(norm_first=True, activation=GELU, layer_norm_eps=1e-6), eval mode."""
    B, N, D = x.shape
    hd = D // heads
    h = _ln(x, sd, key + ".norm1")
    qkv = F.linear(h, sd[key + ".self_attn.in_proj_weight"], sd[key + ".self_attn.in_proj_bias"])
    qkv = qkv.view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).permute(0, 2, 1, 3).reshape(B, N, D)
    x = x + _lin(a, sd, key + ".self_attn.out_proj")
    h = _ln(x, sd, key + ".norm2")
    return x + _lin(F.gelu(_lin(h, sd, key + ".linear1")), sd, key + ".linear2")


def synchformer_segments(sd: SD, x: Tensor, prefix: str = "vfeat_extractor.") -> Tensor:
    """Synchformer.forward on a batch of segments: x [N, 16, 3, 224, 224] (pre-processed frames) ->
    [N, 8, 768] (synchformer.py:44-50, motionformer.py:214-250, video_model_builder.py:166-256)."""
    p = prefix
    N = x.shape[0]
    w = sd[p + "patch_embed_3d.proj.weight"]                       # [D, 3, 2, 16, 16]
    D, _c, zt, ph, pw = w.shape
    x = x.permute(0, 2, 1, 3, 4)                                    # [N, 3, T, H, W]
    # under autocast (the GPU path) the fp32 frames go to the conv as they are - ONE cast to fp16 by autocast, like the
    # reference (feature_utils.py:99-104); casting to a bf16 parameter dtype first would drop three mantissa bits
    if not (x.is_cuda and torch.is_autocast_enabled()):
        x = x.to(w.dtype)
    x = F.conv3d(x, w, sd[p + "patch_embed_3d.proj.bias"], stride=(zt, ph, pw))
    frames, space = x.shape[2], x.shape[3] * x.shape[4]             # 8, 196
    x = x.flatten(2).transpose(1, 2)                                # [N, frames*space, D], token order (t, h, w)
    pos, temp = sd[p + "pos_embed"], sd[p + "temp_embed"]           # [1, 1+space, D], [1, frames, D]
    total = pos[:, 1:].repeat(1, frames, 1) + temp.repeat_interleave(space, dim=1)
    x = torch.cat((sd[p + "cls_token"].expand(N, -1, -1), x), dim=1) + torch.cat((pos[:, :1], total), dim=1)
    depth = 1 + max(int(k[len(p) + 7:].split(".")[0]) for k in sd if k.startswith(p + "blocks."))
    for i in range(depth):                                          # DividedSpaceTimeBlock (vit_helper.py:150-170)
        b = f"{p}blocks.{i}"
        x = x + _divided_attention(_ln(x, sd, b + ".norm3"), sd, b + ".timeattn", frames, space, "time")
        x = x + _divided_attention(_ln(x, sd, b + ".norm1"), sd, b + ".attn", frames, space, "space")
        h = _ln(x, sd, b + ".norm2")
        x = x + _lin(F.gelu(_lin(h, sd, b + ".mlp.fc1")), sd, b + ".mlp.fc2")
    x = _ln(x[:, 1:], sd, p + "norm")                               # CLS dropped, then the final norm
    # spatial aggregation: one encoder layer per (segment, frame) over its `space` tokens + an own CLS
    x = x.reshape(N * frames, space, D)
    x = torch.cat((sd[p + "spatial_attn_agg.cls_token"].expand(N * frames, -1, -1), x), dim=1)
    x = _encoder_layer(x, sd, p + "spatial_attn_agg")
    return x[:, 0].reshape(N, frames, D)                            # temp_attn_agg is Identity


def encode_video_with_sync(sd: SD, frames: Tensor, batch_size: int = 8) -> Tensor:
    """frames [T, 3, 224, 224] (25 fps, pre-processed) -> [1, num_segments*8, 768]
    (feature_utils.py:80-108: 16-frame segments, stride 8; fp16 autocast on the GPU like the reference)."""
    T = frames.shape[0]
    n_seg = (T - SYNC_SEGMENT) // SYNC_STRIDE + 1
    if n_seg < 1:
        raise ValueError(f"Synchformer needs at least {SYNC_SEGMENT} frames at 25 fps (got {T})")
    segs = torch.stack([frames[i * SYNC_STRIDE:i * SYNC_STRIDE + SYNC_SEGMENT] for i in range(n_seg)])
    outs = []
    for i in range(0, n_seg, batch_size):
        chunk = segs[i:i + batch_size]
        if chunk.is_cuda:
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                outs.append(synchformer_segments(sd, chunk))
        else:
            outs.append(synchformer_segments(sd, chunk))
    return torch.cat(outs).reshape(1, n_seg * 8, -1)


def load_synchformer_state(sd: SD, device, dtype: torch.dtype) -> SD:
    """Keep what Synchformer.forward touches (the visual extractor) - the audio branch, the projections
    and the sync transformer of the checkpoint are never used by the sampler (synchformer.py:44-50)."""
    keep = {k: v.to(device=device, dtype=dtype) for k, v in sd.items()
            if k.startswith("vfeat_extractor.") and v.is_floating_point()}
    if "vfeat_extractor.patch_embed_3d.proj.weight" not in keep:
        raise ValueError("not a Synchformer checkpoint: vfeat_extractor.patch_embed_3d.proj.weight is missing")
    return keep


def synchformer_schema(depth: int = 12, dim: int = 768, frames: int = 8, grid: int = 14):
    """state-dict schema of the visual extractor (key -> (shape, std, mean)) for synthesised weights
    (no checkpoints in the image): same keys / shapes as Synchformer().state_dict()."""
    s = {}
    p = "vfeat_extractor."

    def lin(k, o, i):
        s[p + k + ".weight"] = ((o, i), 1.0 / math.sqrt(i), 0.0)
        s[p + k + ".bias"] = ((o,), 0.02, 0.0)

    def ln(k):
        s[p + k + ".weight"] = ((dim,), 0.05, 1.0)
        s[p + k + ".bias"] = ((dim,), 0.02, 0.0)

    s[p + "cls_token"] = ((1, 1, dim), 0.02, 0.0)
    s[p + "pos_embed"] = ((1, grid * grid + 1, dim), 0.02, 0.0)
    s[p + "temp_embed"] = ((1, frames, dim), 0.02, 0.0)
    s[p + "patch_embed_3d.proj.weight"] = ((dim, 3, 2, 16, 16), 1.0 / math.sqrt(3 * 2 * 256), 0.0)
    s[p + "patch_embed_3d.proj.bias"] = ((dim,), 0.02, 0.0)
    for i in range(depth):
        b = f"blocks.{i}."
        for n in ("norm1", "norm2", "norm3"):
            ln(b + n)
        for a in ("attn", "timeattn"):
            lin(b + a + ".qkv", 3 * dim, dim)
            lin(b + a + ".proj", dim, dim)
        lin(b + "mlp.fc1", 4 * dim, dim)
        lin(b + "mlp.fc2", dim, 4 * dim)
    ln("norm")
    a = "spatial_attn_agg."
    s[p + a + "cls_token"] = ((1, 1, dim), 0.02, 0.0)
    s[p + a + "self_attn.in_proj_weight"] = ((3 * dim, dim), 1.0 / math.sqrt(dim), 0.0)
    s[p + a + "self_attn.in_proj_bias"] = ((3 * dim,), 0.02, 0.0)
    lin(a + "self_attn.out_proj", dim, dim)
    lin(a + "linear1", 4 * dim, dim)
    lin(a + "linear2", dim, 4 * dim)
    ln(a + "norm1")
    ln(a + "norm2")
    return s


# ----------------------------------------------------------------------------- SigLIP2 / CLAP (transformers)
@torch.inference_mode()
def encode_video_with_siglip2(model, frames: Tensor, batch_size: int = 16) -> Tensor:
    """frames [T,3,512,512] -> [1, T, 768]: pooled image features per frame (feature_utils.py:63-78;
    transformers >= 5 returns a BaseModelOutputWithPooling instead of a tensor)."""
    outs = []
    for i in range(0, frames.shape[0], batch_size):
        o = model.get_image_features(pixel_values=frames[i:i + batch_size])
        outs.append(o.pooler_output if hasattr(o, "pooler_output") else o)
    return torch.cat(outs).unsqueeze(0)


@torch.inference_mode()
def encode_text_feat(tokenizer, model, prompts, device) -> Tensor:
    """CLAP last_hidden_state for a list of prompts (feature_utils.py:133-138).  On a HIP device the text encoder runs on
    libfoley_hip.so (host/encoders_hip.py::clap_text_hidden_hip, over the HF model's state dict, in the model's dtype); on the
    CPU (tests) the `transformers` module runs."""
    inputs = tokenizer(prompts, padding=True, return_tensors="pt").to(device)
    if torch.device(device).type == "cuda":
        from . import encoders_hip as EH
        cur = _cached_state(model, "_foley_text_sd", device, lambda k, v: k.startswith("text_model.") and v.is_floating_point())
        cfgm = model.config
        out = EH.clap_text_hidden_hip(cur, inputs["input_ids"], inputs["attention_mask"], next(model.parameters()).dtype,
                                      heads=cfgm.num_attention_heads, eps=cfgm.layer_norm_eps, pad_id=cfgm.pad_token_id)
        return out.to(next(model.parameters()).dtype)
    out = model(**inputs, output_hidden_states=True, return_dict=True)
    return out.last_hidden_state


@torch.inference_mode()
def video_features(frames_8fps: Tensor, frames_25fps: Tensor, siglip2_model, sync_sd: SD, device,
                   model_dtype: Optional[torch.dtype] = None, timings: Optional[dict] = None):
    """feature_process_from_tensors' visual half (utils.py:262-283).  Returns (features, audio_len_in_s).

    On a HIP device the selected uint8 frames are moved to the GPU first, pre-processed there (`_resize_u8`'s two-pass
    branch reproduces the reference's CPU uint8 resize) and BOTH encoders run on libfoley_hip.so
    (host/encoders_hip.py: SigLIP2 in the model dtype like the reference's `siglip2_model.to(device, dtype)`,
    Synchformer with fp16 operands like its fp16 autocast, feature_utils.py:63-108).  On the CPU (tests) the torch
    restatement / the `transformers` module run instead.  `timings` (optional dict) receives per-stage milliseconds."""
    import time
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    if model_dtype is None:
        model_dtype = next(siglip2_model.parameters()).dtype

    def lap(name, t0):
        if timings is not None:
            if on_gpu:
                torch.cuda.synchronize(device)
            timings[name] = timings.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)

    t0 = time.perf_counter()
    if on_gpu:
        frames_8fps, frames_25fps = frames_8fps.to(device), frames_25fps.to(device)
    p8 = siglip2_preprocess(frames_8fps)
    p25 = synchformer_preprocess(frames_25fps)
    lap("preprocess_ms", t0)
    if on_gpu:
        from . import encoders_hip as EH
        t0 = time.perf_counter()
        sig_sd = _siglip_state(siglip2_model, device)
        sig = EH.siglip_image_features_hip(sig_sd, p8, model_dtype).unsqueeze(0)
        lap("siglip2_ms", t0)
        t0 = time.perf_counter()
        sync = EH.encode_video_with_sync_hip(sync_sd, p25, torch.float16)
        lap("synchformer_ms", t0)
        feats = {"siglip2_feat": sig, "syncformer_feat": sync}
    else:
        feats = {"siglip2_feat": encode_video_with_siglip2(siglip2_model, p8.to(device=device, dtype=model_dtype)),
                 "syncformer_feat": encode_video_with_sync(sync_sd, p25.to(device))}
    return feats, frames_25fps.shape[0] / float(FPS_SYNC)


def _siglip_state(model, device) -> SD:
    """State dict of the HF SigLIP model's vision tower on `device`, cached on the module (the engine stages its own
    compute-dtype copies of the matrices once per state dict object)."""
    return _cached_state(model, "_foley_vision_sd", device,
                         lambda k, v: k.startswith("vision_model.") or k.startswith(("embeddings.", "encoder.", "post_layernorm.", "head.")))


def _weights_signature(model):
    """Cheap identity of a module's current weights: storage address and in-place version counter of every parameter - changes
    when the weights are reloaded, moved (ComfyUI off-loading) or modified in place."""
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in model.parameters())


def _cached_state(model, attr: str, device, keep) -> SD:
    """Device copy of the part of an HF module's state dict an engine encoder reads, cached ON the module together with the
    signature of the weights it was taken from: reloaded / moved / edited weights invalidate it (and, through the new dict
    object, the engine's staged matrices - host/encoders_hip.py keys them on the dict).  `release_encoder_caches` drops it."""
    cur = getattr(model, attr, None)
    sig = _weights_signature(model)
    if cur is None or cur[0] != sig or cur[1] != torch.device(device):
        sd = {k: v.detach().to(device) for k, v in model.state_dict().items() if keep(k, v)}
        cur = (sig, torch.device(device), sd)
        setattr(model, attr, cur)
    return cur[2]


def release_encoder_caches(*models) -> None:
    """Drop the device copies of the encoders' state dicts cached on `models` (all of them: pass the HF modules the loader
    holds) and every matrix the HIP engine staged from them - call it when the dependencies are unloaded."""
    from . import encoders_hip as EH
    for m in models:
        for attr in ("_foley_text_sd", "_foley_vision_sd"):
            if hasattr(m, attr):
                delattr(m, attr)
    EH._ENGINES.clear()
