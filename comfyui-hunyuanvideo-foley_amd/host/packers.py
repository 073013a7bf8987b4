"""Weight packers: reference-keyed state dicts -> the packed arena libfoley_hip.so consumes.

The reference loads checkpoints into nn.Modules (nodes.py:72-133, utils.py:61-87) and re-derives
layouts on every forward (einops rearranges, conv permutes, weight-norm parametrisation).  Here
every layout decision is taken ONCE at load time:

  * Linear weights stay [N, K] (K contiguous = the MFMA-friendly "NT" GEMM operand).
  * Single-block `linear_qkv` rows are permuted from the reference's "(H D K)" packing
    (hifi_foley.py:362) to "(K H D)", so one head-split kernel serves both block types.
  * Channels-last conv k=3 weights [out, in, 3] become [out, 3*in] with the tap outermost
    (K index = tap*in + c): the conv is then a GEMM over overlapping activation rows.
  * SwiGLU pairs (w1, w3) are fused into one [2*hidden, K] matrix whose rows alternate in
    groups of 32 (w1 rows g*32.., then w3 rows g*32..): both halves of a gate land in the same
    lane of adjacent 32x32 MFMA fragments and the gate is applied in registers.
  * DAC: weight-norm folded (w = g*v/||v||, nn/layers.py:9-14; dim 0 = *input* channel for the
    transposed convs), convs packed tap-major like above, each stride-s transposed conv packed as
    s output phases x 2 taps: Wt[p*Cout + co, (x[q-1] | x[q])] = (w[:, co, p+s] | w[:, co, p]).

All matrices are cast to the compute dtype (fp32 parity mode / bf16 throughput mode); biases,
norm gains and snake alphas stay fp32.  Everything is laid out in ONE flat device buffer (the
"arena") so multi-GPU runs can ship the model with a single broadcast.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch

from .config import DACConfig, DiTConfig

Tensor = torch.Tensor

_TRIPLE_LINEARS = (  # packed name, reference module, has bias
    ("a_mod", "audio_mod.linear"), ("v_mod", "v_cond_mod.linear"),
    ("a_qkv", "audio_self_attn_qkv"), ("v_qkv", "v_cond_attn_qkv"),
    ("a_proj", "audio_self_proj"), ("v_proj", "v_cond_self_proj"),
    ("a_cq", "audio_cross_q"), ("v_cq", "v_cond_cross_q"),
    ("t_kv", "text_cross_kv"),
    ("a_cproj", "audio_cross_proj"), ("v_cproj", "v_cond_cross_proj"),
    ("a_fc1", "audio_mlp.fc1"), ("a_fc2", "audio_mlp.fc2"),
    ("v_fc1", "v_cond_mlp.fc1"), ("v_fc2", "v_cond_mlp.fc2"),
)
_TRIPLE_GAINS = (
    ("a_qn", "audio_self_q_norm"), ("a_kn", "audio_self_k_norm"),
    ("v_qn", "v_cond_attn_q_norm"), ("v_kn", "v_cond_attn_k_norm"),
    ("a_cqn", "audio_cross_q_norm"), ("v_cqn", "v_cond_cross_q_norm"),
    ("t_kn", "text_cross_k_norm"),
)


def conv_to_gemm(w: Tensor) -> Tensor:
    """[out, in, k] -> [out, k*in], K index = tap*in + c."""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def interleave_gate(w1: Tensor, w3: Tensor) -> Tensor:
    """Fuse a SwiGLU pair into [2*hidden, K] with alternating 32-row groups (w1 first)."""
    h, k = w1.shape
    assert h % 32 == 0 and w3.shape == w1.shape
    return torch.cat((w1.view(h // 32, 32, k), w3.view(h // 32, 32, k)), dim=1).reshape(2 * h, k).contiguous()


def qkv_hdk_to_khd(w: Tensor, heads: int) -> Tensor:
    """rows packed '(H D K)' (q/k/v innermost) -> '(K H D)'; works for weight [3D, K] and bias [3D]."""
    d3 = w.shape[0]
    hd = d3 // (3 * heads)
    rest = w.shape[1:]
    return w.reshape(heads, hd, 3, *rest).permute(2, 0, 1, *range(3, 3 + len(rest))).reshape(d3, *rest).contiguous()


def _f32(t: Tensor) -> Tensor:
    return t.detach().to(torch.float32).contiguous()


FP8_DTYPES = {"fp8_e4m3fn": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}
HALF_DTYPES = (torch.bfloat16, torch.float16)   # 16-bit compute dtypes of the throughput kernels


def pack_dit(sd: Dict[str, Tensor], cfg: DiTConfig, dtype: torch.dtype,
             weight_store: torch.dtype = None) -> "OrderedDict[str, Tensor]":
    """Reference DiT state dict -> packed tensors (CPU or wherever `sd` lives).

    weight_store (torch.float8_e4m3fn / float8_e5m2, bf16 / fp16 compute only): the matrices of the 54 blocks and the
    fused single-block modulation - 99.7 % of the bytes - STAY in fp8 in the arena (the reference's
    FP8WeightWrapper storage, utils.py:316-366; the GEMM widens them in registers).  `sd` must already hold
    fp8-representable values (nodes.fp8_round_state_dict), so the cast is exact.  The small embedders keep
    the compute dtype."""
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    H = cfg.heads
    if weight_store is not None and dtype not in HALF_DTYPES:
        raise ValueError("fp8 weight storage needs bf16 / fp16 compute (the reference cannot run it in fp32 either)")
    in_block = [False]

    def mat(name: str, w: Tensor):
        w = w.detach().to(torch.float32)
        if weight_store is not None and in_block[0]:
            w8 = w.to(weight_store)
            if not torch.equal(w8.to(torch.float32), w):
                raise ValueError(f"{name}: values are not representable in {weight_store} (round the state dict first)")
            out[name + ".w"] = w8.contiguous()
        else:
            out[name + ".w"] = w.to(dtype).contiguous()

    def lin(name: str, ref: str, bias: bool = True):
        mat(name, sd[ref + ".weight"])
        if bias:
            out[name + ".b"] = _f32(sd[ref + ".bias"])

    in_block[0] = True
    for b in range(cfg.depth_triple):
        p, r = f"t{b}.", f"triple_blocks.{b}."
        for n, m in _TRIPLE_LINEARS:
            lin(p + n, r + m)
        for n, m in _TRIPLE_GAINS:
            out[p + n] = _f32(sd[r + m + ".weight"])
    # All single-block AdaLN projections consume the same per-token conditioning, so they are
    # packed as ONE [n_single*6D, D] matrix: a single large GEMM per loop iteration.
    if cfg.depth_single:
        mat("smod_all", torch.cat([sd[f"single_blocks.{b}.modulation.linear.weight"].float()
                                   for b in range(cfg.depth_single)], dim=0))
        out["smod_all.b"] = torch.cat([_f32(sd[f"single_blocks.{b}.modulation.linear.bias"])
                                       for b in range(cfg.depth_single)], dim=0).contiguous()
    for b in range(cfg.depth_single):
        p, r = f"s{b}.", f"single_blocks.{b}."
        mat(p + "qkv", qkv_hdk_to_khd(sd[r + "linear_qkv.weight"].float(), H))
        out[p + "qkv.b"] = _f32(qkv_hdk_to_khd(sd[r + "linear_qkv.bias"].float(), H))
        out[p + "qn"] = _f32(sd[r + "q_norm.weight"])
        out[p + "kn"] = _f32(sd[r + "k_norm.weight"])
        mat(p + "lin1", conv_to_gemm(sd[r + "linear1.weight"].float()))
        out[p + "lin1.b"] = _f32(sd[r + "linear1.bias"])
        mat(p + "w13", interleave_gate(conv_to_gemm(sd[r + "linear2.w1.weight"].float()),
                                       conv_to_gemm(sd[r + "linear2.w3.weight"].float())))
        mat(p + "w2", conv_to_gemm(sd[r + "linear2.w2.weight"].float()))
    in_block[0] = False
    mat("audio_in", sd["audio_embedder.proj.weight"].float().squeeze(-1))
    out["audio_in.b"] = _f32(sd["audio_embedder.proj.bias"])
    mat("vis.w13", interleave_gate(sd["visual_proj.w1.weight"].float(), sd["visual_proj.w3.weight"].float()))
    mat("vis.w2", sd["visual_proj.w2.weight"])
    lin("cond1", "cond_in.linear_1")
    lin("cond2", "cond_in.linear_2")
    lin("time0", "time_in.mlp.0")
    lin("time2", "time_in.mlp.2")
    lin("sync0", "sync_in.0")
    mat("sync.w13", interleave_gate(sd["sync_in.2.w1.weight"].float().squeeze(-1),
                                    sd["sync_in.2.w3.weight"].float().squeeze(-1)))
    mat("sync.w2", sd["sync_in.2.w2.weight"].float().squeeze(-1))
    out["sync_pos"] = _f32(sd["sync_pos_emb"]).reshape(8, -1)
    lin("final", "final_layer.linear")   # final_layer.adaLN_modulation is dead code
    out["empty_clip"] = _f32(sd["empty_clip_feat"]).reshape(-1)
    out["empty_sync"] = _f32(sd["empty_sync_feat"]).reshape(-1)
    return out


def fold_weight_norm(sd: Dict[str, Tensor], key: str) -> Tensor:
    """Accepts parametrized (`original0/1`), legacy (`weight_g/v`) and already-folded checkpoints."""
    if key + ".parametrizations.weight.original0" in sd:
        g, v = sd[key + ".parametrizations.weight.original0"], sd[key + ".parametrizations.weight.original1"]
    elif key + ".weight_g" in sd:
        g, v = sd[key + ".weight_g"], sd[key + ".weight_v"]
    else:
        return sd[key + ".weight"].float()
    g, v = g.float(), v.float()
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def convT_to_gemm(w: Tensor, stride: int) -> Tensor:
    """ConvTranspose1d weight [Cin, Cout, 2s] -> [s*Cout, 2*Cin]: row (phase p, co),
    K = [x[q-1] channels | x[q] channels] <-> taps (p+s | p)."""
    cin, cout, k = w.shape
    assert k == 2 * stride
    wp = w.permute(2, 1, 0)                       # [2s, Cout, Cin]
    return torch.cat((wp[stride:], wp[:stride]), dim=-1).reshape(stride * cout, 2 * cin).contiguous()


def pack_dac(sd: Dict[str, Tensor], cfg: DACConfig) -> "OrderedDict[str, Tensor]":
    """Decoder half of the DAC-VAE state dict -> packed fp32 tensors (dac.py:120-149, 197)."""
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    out["dac.pq.w"] = _f32(sd["post_quant_conv.weight"]).squeeze(-1).contiguous()
    out["dac.pq.b"] = _f32(sd["post_quant_conv.bias"])
    out["dac.in.w"] = conv_to_gemm(fold_weight_norm(sd, "decoder.model.0"))
    out["dac.in.b"] = _f32(sd["decoder.model.0.bias"])
    n = len(cfg.rates)
    for i, s in enumerate(cfg.rates):
        r, p = f"decoder.model.{i + 1}.block.", f"dac.{i}."
        out[p + "alpha0"] = _f32(sd[r + "0.alpha"]).reshape(-1)
        out[p + "up.w"] = convT_to_gemm(fold_weight_norm(sd, r + "1"), s)
        out[p + "up.b"] = _f32(sd[r + "1.bias"]).repeat(s)
        for j in range(3):
            q, u = r + f"{j + 2}.block.", p + f"{j}."
            out[u + "a1"] = _f32(sd[q + "0.alpha"]).reshape(-1)
            out[u + "c7.w"] = conv_to_gemm(fold_weight_norm(sd, q + "1"))
            out[u + "c7.b"] = _f32(sd[q + "1.bias"])
            out[u + "a2"] = _f32(sd[q + "2.alpha"]).reshape(-1)
            out[u + "c1.w"] = fold_weight_norm(sd, q + "3").squeeze(-1).contiguous()
            out[u + "c1.b"] = _f32(sd[q + "3.bias"])
    out["dac.out.alpha"] = _f32(sd[f"decoder.model.{n + 1}.alpha"]).reshape(-1)
    wo = fold_weight_norm(sd, f"decoder.model.{n + 2}")     # [1, C, 7]
    out["dac.out.w"] = wo[0].permute(1, 0).reshape(-1).contiguous()
    out["dac.out.b"] = _f32(sd[f"decoder.model.{n + 2}.bias"]).reshape(1)
    return out


def has_dac_encoder(sd: Dict[str, Tensor]) -> bool:
    return "encoder.block.0.bias" in sd and "quant_conv.weight" in sd


def pack_dac_encoder(sd: Dict[str, Tensor], cfg: DACConfig) -> "OrderedDict[str, Tensor]":
    """Encoder half + quant_conv of the DAC-VAE state dict -> packed fp32 tensors (dac.py:47-95, 196;
    SURVEY N4).  Strided convs are packed tap-major like every other conv: GEMM row q reads source rows
    q*s - ceil(s/2) + j for tap j < 2s."""
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    w0 = fold_weight_norm(sd, "encoder.block.0")                       # [d, 1, 7]
    out["enc.in.w"] = w0[:, 0, :].permute(1, 0).reshape(-1).contiguous()   # [7][d]
    out["enc.in.b"] = _f32(sd["encoder.block.0.bias"])
    n = len(cfg.encoder_rates)
    for i, s in enumerate(cfg.encoder_rates):
        r, p = f"encoder.block.{i + 1}.block.", f"enc.{i}."
        for j in range(3):
            q, u = r + f"{j}.block.", p + f"{j}."
            out[u + "a1"] = _f32(sd[q + "0.alpha"]).reshape(-1)
            out[u + "c7.w"] = conv_to_gemm(fold_weight_norm(sd, q + "1"))
            out[u + "c7.b"] = _f32(sd[q + "1.bias"])
            out[u + "a2"] = _f32(sd[q + "2.alpha"]).reshape(-1)
            out[u + "c1.w"] = fold_weight_norm(sd, q + "3").squeeze(-1).contiguous()
            out[u + "c1.b"] = _f32(sd[q + "3.bias"])
        out[p + "alpha"] = _f32(sd[r + "3.alpha"]).reshape(-1)
        out[p + "down.w"] = conv_to_gemm(fold_weight_norm(sd, r + "4"))
        out[p + "down.b"] = _f32(sd[r + "4.bias"])
    out["enc.out.alpha"] = _f32(sd[f"encoder.block.{n + 1}.alpha"]).reshape(-1)
    out["enc.out.w"] = conv_to_gemm(fold_weight_norm(sd, f"encoder.block.{n + 2}"))
    out["enc.out.b"] = _f32(sd[f"encoder.block.{n + 2}.bias"])
    out["enc.qc.w"] = _f32(sd["quant_conv.weight"]).squeeze(-1).contiguous()
    out["enc.qc.b"] = _f32(sd["quant_conv.bias"])
    return out


# ----------------------------------------------------------------------------- arena
_ALIGN = 256


def arena_layout(packed: Dict[str, Tensor]) -> Tuple[int, "OrderedDict[str, Tuple[int, torch.dtype, Tuple[int, ...]]]"]:
    """name -> (byte offset, dtype, shape); returns (total bytes, table)."""
    table: "OrderedDict[str, Tuple[int, torch.dtype, Tuple[int, ...]]]" = OrderedDict()
    off = 0
    for k, t in packed.items():
        table[k] = (off, t.dtype, tuple(t.shape))
        off += (t.numel() * t.element_size() + _ALIGN - 1) // _ALIGN * _ALIGN
    return off, table


class Arena:
    """One flat device buffer holding every packed tensor (single-broadcast friendly).  `buffer` may be
    a uint8 view into a larger allocation (the multi-GPU bundle, host/distributed.py)."""

    def __init__(self, total_bytes: int, table, device, buffer: Tensor = None):
        self.table = table
        n = max(total_bytes, _ALIGN)
        if buffer is None:
            buffer = torch.empty(n, dtype=torch.uint8, device=device)
        elif buffer.dtype != torch.uint8 or buffer.numel() < n or not buffer.is_contiguous():
            raise ValueError("arena buffer must be a contiguous uint8 tensor of at least the layout's size")
        elif buffer.device.type != "meta" and buffer.data_ptr() % 16:
            raise ValueError("arena buffer must be 16-byte aligned (libfoley_hip.so's operand alignment)")
        self.buffer = buffer[:n]

    @classmethod
    def from_packed(cls, packed: Dict[str, Tensor], device, buffer: Tensor = None) -> "Arena":
        total, table = arena_layout(packed)
        a = cls(total, table, device, buffer=buffer)
        for k, t in packed.items():
            a.view(k).copy_(t, non_blocking=False)
        return a

    def view(self, name: str) -> Tensor:
        off, dt, shape = self.table[name]
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dt).element_size()
        return self.buffer[off:off + nbytes].view(dt).view(shape)

    def items(self) -> Iterable[Tuple[str, Tensor]]:
        for k in self.table:
            yield k, self.view(k)


def _meta_state(schema) -> Dict[str, Tensor]:
    return {k: torch.empty(shape, device="meta") for k, (shape, _std, _mean) in schema.items()}


def dit_arena_layout(cfg: DiTConfig, dtype: torch.dtype, weight_store: torch.dtype = None):
    """(total bytes, table) of the packed DiT arena, derived from the config alone (the packers run on
    meta tensors of the reference state-dict schema): every rank of a multi-GPU job can lay out the
    arena it is about to receive without a metadata exchange."""
    from . import synth
    return arena_layout(_pack_dit_meta(synth.dit_schema(cfg), cfg, dtype, weight_store))


def _pack_dit_meta(schema, cfg, dtype, weight_store):
    """pack_dit on meta tensors (layout only): the representability check is skipped."""
    packed = pack_dit(_meta_state(schema), cfg, dtype)
    if weight_store is None:
        return packed
    blocks = lambda k: k.endswith(".w") and (k[0] in "ts" and k[1].isdigit() or k.startswith("smod_all"))
    return OrderedDict((k, torch.empty(v.shape, dtype=weight_store, device="meta") if blocks(k) else v) for k, v in packed.items())


def dac_arena_layout(cfg: DACConfig):
    from . import synth
    return arena_layout(pack_dac(_meta_state(synth.dac_decoder_schema(cfg)), cfg))


def torch_dtype(name: str) -> torch.dtype:
    return {"fp32": torch.float32, "bf16": torch.bfloat16, "f32": torch.float32, "fp16": torch.float16, "f16": torch.float16}[name]
