"""Deterministic synthesised checkpoints (no network, no real weights in the image).

Every tensor of the reference state-dict schema (keys as
produced by `HunyuanVideoFoley.state_dict()` / `DAC.state_dict()`) is filled by
a counter-based integer hash keyed by (crc32(key), flat index), so the build
container, the GPU box, the oracle and the HIP path all see bit-identical
weights without committing them.  The hash is evaluated with torch integer ops,
so it can run on the GPU (seconds for the 5.1 B-parameter xxl model).

Scales are chosen so a random model is well-conditioned: unit-variance
projections (std = gain/sqrt(fan_in)), small AdaLN shift/scale/gate (the
reference zero-initialises those: modulate_layers.py:12-13, mlp_layers.py:86-95
- zeros would turn every block into the identity), norm gains 1±0.1.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch

from .config import DACConfig, DiTConfig

_M32 = 0xFFFFFFFF


def _hash_u24(seed: int, n: int, device) -> torch.Tensor:
    """lowbias32-style avalanche hash of (seed, i) -> 24-bit integers (int64 tensor)."""
    i = torch.arange(n, dtype=torch.int64, device=device)
    x = (i * 0x9E3779B1 + seed) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x >> 8


def synth_tensor(key: str, shape: Tuple[int, ...], std: float, mean: float = 0.0,
                 device="cpu", seed: int = 0) -> torch.Tensor:
    """Uniform[-a, a] + mean with a = std*sqrt(3); fp32; bit-reproducible everywhere."""
    n = 1
    for s in shape:
        n *= int(s)
    h = zlib.crc32(key.encode("utf-8")) ^ ((seed * 0x85EBCA6B) & _M32)
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = 1 << 24
    a = float(std) * math.sqrt(3.0)
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        u = _hash_u24((h + s0 * 0x9E3779B1) & _M32, m, device).to(torch.float32)
        # u in [0, 2^24): exact in fp32.  (u * 2^-23 - 1) in [-1, 1)
        out[s0:s0 + m] = (u * (2.0 ** -23) - 1.0) * a + mean
    return out.view(*shape)


# --------------------------------------------------------------------------- schema
def _lin(sd, key, out_f, in_f, gain=1.0, bias=True, kshape=None):
    shape = (out_f, in_f) if kshape is None else (out_f, in_f, kshape)
    fan = in_f * (kshape or 1)
    sd[key + ".weight"] = (shape, gain / math.sqrt(fan), 0.0)
    if bias:
        sd[key + ".bias"] = ((out_f,), 0.02, 0.0)


def dit_schema(cfg: DiTConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]":
    """key -> (shape, std, mean) for every tensor of the DiT state dict."""
    D, hd = cfg.hidden, cfg.head_dim
    sd: "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]" = OrderedDict()
    for b in range(cfg.depth_triple):
        p = f"triple_blocks.{b}."
        for s in ("audio_mod", "v_cond_mod"):
            _lin(sd, p + s + ".linear", 9 * D, D, gain=0.35)
        for s in ("audio_self_attn_qkv", "v_cond_attn_qkv"):
            _lin(sd, p + s, 3 * D, D)
        for s in ("audio_self_q_norm", "audio_self_k_norm", "v_cond_attn_q_norm",
                  "v_cond_attn_k_norm", "audio_cross_q_norm", "v_cond_cross_q_norm",
                  "text_cross_k_norm"):
            sd[p + s + ".weight"] = ((hd,), 0.06, 1.0)
        for s in ("audio_self_proj", "v_cond_self_proj", "audio_cross_q", "v_cond_cross_q",
                  "audio_cross_proj", "v_cond_cross_proj"):
            _lin(sd, p + s, D, D)
        _lin(sd, p + "text_cross_kv", 2 * D, D)
        for s in ("audio_mlp", "v_cond_mlp"):
            _lin(sd, p + s + ".fc1", cfg.mlp_hidden, D)
            _lin(sd, p + s + ".fc2", D, cfg.mlp_hidden)
    for b in range(cfg.depth_single):
        p = f"single_blocks.{b}."
        _lin(sd, p + "modulation.linear", 6 * D, D, gain=0.35)
        _lin(sd, p + "linear_qkv", 3 * D, D)
        sd[p + "q_norm.weight"] = ((hd,), 0.06, 1.0)
        sd[p + "k_norm.weight"] = ((hd,), 0.06, 1.0)
        _lin(sd, p + "linear1", D, D, kshape=3)
        _lin(sd, p + "linear2.w1", cfg.conv_hidden, D, bias=False, kshape=3)
        _lin(sd, p + "linear2.w2", D, cfg.conv_hidden, bias=False, kshape=3)
        _lin(sd, p + "linear2.w3", cfg.conv_hidden, D, bias=False, kshape=3)
    _lin(sd, "audio_embedder.proj", D, cfg.latent_dim, kshape=1)
    _lin(sd, "visual_proj.w1", D, cfg.clip_dim, bias=False)
    _lin(sd, "visual_proj.w2", D, D, bias=False)
    _lin(sd, "visual_proj.w3", D, cfg.clip_dim, bias=False)
    _lin(sd, "cond_in.linear_1", D, cfg.cond_dim)
    _lin(sd, "cond_in.linear_2", D, D)
    _lin(sd, "time_in.mlp.0", D, cfg.time_freq_dim)
    _lin(sd, "time_in.mlp.2", D, D)
    _lin(sd, "sync_in.0", D, cfg.sync_dim)
    _lin(sd, "sync_in.2.w1", cfg.sync_hidden, D, bias=False, kshape=1)
    _lin(sd, "sync_in.2.w2", D, cfg.sync_hidden, bias=False, kshape=1)
    _lin(sd, "sync_in.2.w3", cfg.sync_hidden, D, bias=False, kshape=1)
    sd["sync_pos_emb"] = ((1, 1, 8, cfg.sync_dim), 0.1, 0.0)
    _lin(sd, "final_layer.linear", cfg.latent_dim, D)
    _lin(sd, "final_layer.adaLN_modulation.1", 2 * D, D, gain=0.35)  # dead
    sd["empty_clip_feat"] = ((1, cfg.clip_dim), 0.5, 0.0)
    sd["empty_sync_feat"] = ((1, cfg.sync_dim), 0.5, 0.0)
    return sd


def _wn(sd, key, v_shape, g_len, bias_len, g_mean=1.0):
    """weight-normed conv: parametrizations.weight.original0 (g) / original1 (v).

    g sets the L2 norm of each dim-0 slice of the folded weight; the means are
    picked so activations stay O(1) through the 5 stages (residual branches
    damped to 0.3, transposed convs compensated for their 2-of-2s tap overlap).
    """
    sd[key + ".bias"] = ((bias_len,), 0.02, 0.0)
    sd[key + ".parametrizations.weight.original0"] = ((g_len, 1, 1), 0.06 * g_mean, g_mean)
    sd[key + ".parametrizations.weight.original1"] = (v_shape, 0.05, 0.0)


def dac_decoder_schema(cfg: DACConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]":
    """Decoder-side keys of `DAC(**_DAC_KWARGS).state_dict()` (dac.py:120-149, 197)."""
    sd: "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]" = OrderedDict()
    L = cfg.latent_dim
    sd["post_quant_conv.weight"] = ((L, L, 1), 1.0 / math.sqrt(L), 0.0)
    sd["post_quant_conv.bias"] = ((L,), 0.02, 0.0)
    ch = cfg.decoder_dim
    _wn(sd, "decoder.model.0", (ch, L, 7), ch, ch)
    cin = ch
    for i, s in enumerate(cfg.rates):
        cout = ch // 2 ** (i + 1)
        p = f"decoder.model.{i + 1}.block."
        sd[p + "0.alpha"] = ((1, cin, 1), 0.15, 1.0)
        _wn(sd, p + "1", (cin, cout, 2 * s), cin, cout,       # ConvTranspose1d: g per in-ch
            g_mean=0.8 * math.sqrt(s / 2.0))
        for j in range(3):
            q = p + f"{j + 2}.block."
            sd[q + "0.alpha"] = ((1, cout, 1), 0.15, 1.0)
            _wn(sd, q + "1", (cout, cout, 7), cout, cout)
            sd[q + "2.alpha"] = ((1, cout, 1), 0.15, 1.0)
            _wn(sd, q + "3", (cout, cout, 1), cout, cout, g_mean=0.3)
        cin = cout
    n = len(cfg.rates)
    sd[f"decoder.model.{n + 1}.alpha"] = ((1, cin, 1), 0.15, 1.0)
    _wn(sd, f"decoder.model.{n + 2}", (1, cin, 7), 1, 1, g_mean=0.5)
    return sd


def dac_encoder_schema(cfg: DACConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]":
    """Encoder-side keys of `DAC(**_DAC_KWARGS).state_dict()` (dac.py:47-95, 196): first conv,
    one EncoderBlock per rate (3 residual units at the incoming width, snake, strided conv k=2s that
    doubles the channels), snake + conv k=3 to the latent width, and the 1x1 `quant_conv` that emits
    the posterior's (mean, logvar)."""
    sd: "OrderedDict[str, Tuple[Tuple[int, ...], float, float]]" = OrderedDict()
    d = cfg.encoder_dim
    _wn(sd, "encoder.block.0", (d, 1, 7), d, d, g_mean=1.0)
    for i, s_ in enumerate(cfg.encoder_rates):
        p = f"encoder.block.{i + 1}.block."
        for j in range(3):
            q = p + f"{j}.block."
            sd[q + "0.alpha"] = ((1, d, 1), 0.15, 1.0)
            _wn(sd, q + "1", (d, d, 7), d, d)
            sd[q + "2.alpha"] = ((1, d, 1), 0.15, 1.0)
            _wn(sd, q + "3", (d, d, 1), d, d, g_mean=0.3)
        sd[p + "3.alpha"] = ((1, d, 1), 0.15, 1.0)
        _wn(sd, p + "4", (2 * d, d, 2 * s_), 2 * d, 2 * d, g_mean=1.0)
        d *= 2
    n = len(cfg.encoder_rates)
    sd[f"encoder.block.{n + 1}.alpha"] = ((1, d, 1), 0.15, 1.0)
    _wn(sd, f"encoder.block.{n + 2}", (cfg.latent_dim, d, 3), cfg.latent_dim, cfg.latent_dim, g_mean=1.0)
    L = cfg.latent_dim
    sd["quant_conv.weight"] = ((2 * L, L, 1), 1.0 / math.sqrt(L), 0.0)
    sd["quant_conv.bias"] = ((2 * L,), 0.05, 0.0)
    return sd


def materialize(schema, device="cpu", seed: int = 0, dtype=torch.float32,
                keys: Iterable[str] = None) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = OrderedDict()
    for k in (keys if keys is not None else schema.keys()):
        shape, std, mean = schema[k]
        t = synth_tensor(k, shape, std, mean, device=device, seed=seed)
        out[k] = t if dtype == torch.float32 else t.to(dtype)
    return out


def synth_dit_state_dict(cfg: DiTConfig, device="cpu", seed: int = 0, dtype=torch.float32):
    return materialize(dit_schema(cfg), device=device, seed=seed, dtype=dtype)


def synth_dac_state_dict(cfg: DACConfig, device="cpu", seed: int = 0, encoder: bool = False):
    """Decoder-side checkpoint (what the sampler needs); encoder=True adds the encoder / quant_conv keys."""
    sd = materialize(dac_decoder_schema(cfg), device=device, seed=seed)
    if encoder:
        sd.update(materialize(dac_encoder_schema(cfg), device=device, seed=seed))
    return sd


def synth_conditioning(cfg: DiTConfig, duration_s: float, *, t2a: bool, sd=None, seed: int = 1,
                       n_text: int = 12, n_neg: int = 5, device="cpu"):
    """Stand-ins for CLAP / SigLIP2 / Synchformer outputs.

    t2a=True : clip/sync = learned empty rows expanded (nodes.py:326-333).
    t2a=False: clip/sync ~ U(std 1) pseudo-features.
    Returns dict(text, uncond_text, clip, sync) with shapes [1, *, 768].
    """
    from .config import lengths
    la, lv, ls = lengths(duration_s, cfg)
    c = {
        "text": synth_tensor("cond.text", (1, n_text, cfg.cond_dim), 1.0, device=device, seed=seed),
        "uncond_text": synth_tensor("cond.neg", (1, n_neg, cfg.cond_dim), 1.0, device=device, seed=seed),
    }
    if t2a:
        assert sd is not None
        c["clip"] = sd["empty_clip_feat"].to(device).unsqueeze(0).expand(1, lv, -1).contiguous()
        c["sync"] = sd["empty_sync_feat"].to(device).unsqueeze(0).expand(1, ls, -1).contiguous()
    else:
        c["clip"] = synth_tensor("cond.clip", (1, lv, cfg.clip_dim), 1.0, device=device, seed=seed + 1)
        c["sync"] = synth_tensor("cond.sync", (1, ls, cfg.sync_dim), 1.0, device=device, seed=seed + 1)
    return c
