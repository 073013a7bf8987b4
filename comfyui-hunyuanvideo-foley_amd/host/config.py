"""Model / diffusion hyper-parameters of the Foley sampling path.

Mirrors the *data* of the reference's `configs/hunyuanvideo-foley-{xxl,xl}.yaml`
(model_kwargs + diffusion_config) and the fixed DAC-VAE constructor arguments
(`/utils.py:32-44` `_DAC_KWARGS`).  Only the keys the sampling path reads are
kept; the YAML files under `configs/` carry the same numbers for users who
edit them.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Tuple

import yaml


@dataclasses.dataclass(frozen=True)
class DiTConfig:
    name: str = "xxl"
    depth_triple: int = 18
    depth_single: int = 36
    hidden: int = 1536
    heads: int = 12
    mlp_ratio: int = 4
    cond_dim: int = 768        # CLAP text width
    clip_dim: int = 768        # SigLIP2 width
    sync_dim: int = 768        # Synchformer width
    latent_dim: int = 128      # DAC latent channels
    frame_rate: int = 50       # latent frames per second
    text_len: int = 77
    time_freq_dim: int = 256   # TimestepEmbedder frequency_embedding_size
    flow_shift: float = 1.0

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def mlp_hidden(self) -> int:   # triple-block GELU MLP (hifi_foley.py:87)
        return int(self.hidden * self.mlp_ratio)

    @property
    def conv_hidden(self) -> int:  # ConvMLP hidden (mlp_layers.py:141-142)
        h = int(2 * (self.hidden * self.mlp_ratio) / 3)
        return 256 * ((h + 255) // 256)

    @property
    def sync_hidden(self) -> int:  # sync_in ConvMLP(k=1), hidden_dim = 4*hidden
        h = int(2 * (self.hidden * 4) / 3)
        return 256 * ((h + 255) // 256)


@dataclasses.dataclass(frozen=True)
class DACConfig:
    latent_dim: int = 128
    decoder_dim: int = 2048
    rates: Tuple[int, ...] = (8, 5, 4, 3, 2)
    sample_rate: int = 48000
    dilations: Tuple[int, ...] = (1, 3, 9)
    # encoder half (dac.py:47-95; `_DAC_KWARGS` utils.py:32-44): channels double per stage
    encoder_dim: int = 128
    encoder_rates: Tuple[int, ...] = (2, 3, 4, 5, 8)

    @property
    def hop(self) -> int:
        h = 1
        for r in self.rates:
            h *= r
        return h


XXL = DiTConfig()
XL = DiTConfig(name="xl", depth_triple=12, depth_single=24, hidden=1408, heads=11)
# small configuration used by fast tests; head_dim stays 128 like the real models
TINY = DiTConfig(name="tiny", depth_triple=2, depth_single=2, hidden=256, heads=2)
DAC48K = DACConfig()
# narrow decoder for fast tests (same topology: 5 stages, same rates); the last stage keeps 32
# channels = one 128-byte fp32 K-slice, the engine's minimum tap width
DAC_TINY = DACConfig(decoder_dim=1024)
# narrow codec for the encoder tests: 32 -> 64 -> 128 channels, hop 6 (latent width stays 128 = the DiT's)
DAC_ENC_TINY = DACConfig(latent_dim=128, decoder_dim=128, rates=(3, 2), encoder_dim=32, encoder_rates=(2, 3))

_BY_NAME = {"xxl": XXL, "xl": XL, "tiny": TINY}


def dit_config(name: str) -> DiTConfig:
    return _BY_NAME[name]


def lengths(duration_s: float, cfg: DiTConfig = XXL) -> Tuple[int, int, int]:
    """(La, Lv, Ls) for text-to-audio (nodes.py:326-329, utils.py:154)."""
    la = int(duration_s * cfg.frame_rate)
    lv = int(duration_s * 8)
    ls = 8 * ((int(duration_s * 25) - 16) // 8 + 1)
    return la, lv, ls


def load_yaml_config(path: str) -> DiTConfig:
    """Reads a reference-format YAML (model_config.model_kwargs / diffusion_config)."""
    with open(path, "r", encoding="utf-8") as f:
        raw = yaml.safe_load(f)
    kw = raw["model_config"]["model_kwargs"]
    dc = raw.get("diffusion_config", {})
    name = os.path.splitext(os.path.basename(path))[0].split("-")[-1]
    return DiTConfig(
        name=name,
        depth_triple=int(kw["depth_triple_blocks"]),
        depth_single=int(kw["depth_single_blocks"]),
        hidden=int(kw["hidden_size"]),
        heads=int(kw["num_heads"]),
        mlp_ratio=int(kw.get("mlp_ratio", 4)),
        cond_dim=int(kw.get("condition_dim", 768)),
        clip_dim=int(kw.get("clip_dim", 768)),
        sync_dim=int(kw.get("sync_feat_dim", 768)),
        latent_dim=int(kw.get("audio_vae_latent_dim", 128)),
        frame_rate=int(kw.get("audio_frame_rate", 50)),
        text_len=int(kw.get("text_length", 77)),
        flow_shift=float(dc.get("sample_flow_shift", 1.0)),
    )
