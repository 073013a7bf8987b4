"""Clip-level data parallelism over the GPUs of one node.

The path has no cross-clip arithmetic, so there is nothing to all-reduce: one process per GPU,
rank 0 loads + packs the checkpoint, and EVERYTHING the other ranks need - the packed DiT arena,
the packed DAC-decoder arena and the conditioning of the job - lives in one flat byte buffer (the
"bundle") that is shipped with a SINGLE `broadcast` (RCCL over xGMI when the backend is "nccl";
gloo in the CPU tests).  No metadata travels: the bundle's layout is a pure function of the model
configuration, the compute dtype and the clip duration (`bundle_spec`), so every rank computes it
locally (the packers run on meta tensors) and allocates the receive buffer before the collective.
The noise of the whole job is drawn on every rank from the same CPU generator seed and sliced
(`shard_range`), so the result is bit-identical to a single-GPU run of the full batch; waveforms
return by per-rank D2H.  There is no per-step collective.
"""
from __future__ import annotations

import dataclasses
import time
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import packers
from .config import DACConfig, DiTConfig, lengths

_ALIGN = 256
COND_KEYS = ("text", "uncond_text", "clip", "sync")


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` clips owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _round_up(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


@dataclasses.dataclass
class BundleSpec:
    """Byte layout of the broadcast buffer: [DiT arena | DAC arena | conditioning (fp32)]."""
    total: int
    dit_off: int
    dit_bytes: int
    dit_table: "OrderedDict"
    dac_off: int
    dac_bytes: int
    dac_table: "OrderedDict"
    cond_off: int
    cond_shapes: "OrderedDict[str, Tuple[int, ...]]"


def bundle_spec(cfg: DiTConfig, dac_cfg: DACConfig, dtype: torch.dtype, duration_s: float,
                lv: Optional[int] = None, ls: Optional[int] = None, weight_store: torch.dtype = None) -> BundleSpec:
    """Deterministic on every rank.  Text rows are carried zero-padded to `cfg.text_len` (what
    `_pad_or_trim_time` makes of them anyway, utils.py:103-111), so all shapes are fixed."""
    dit_bytes, dit_table = packers.dit_arena_layout(cfg, dtype, weight_store)
    dac_bytes, dac_table = packers.dac_arena_layout(dac_cfg)
    _la, lv0, ls0 = lengths(duration_s, cfg)
    lv, ls = lv or lv0, ls or ls0
    shapes = OrderedDict(text=(1, cfg.text_len, cfg.cond_dim), uncond_text=(1, cfg.text_len, cfg.cond_dim),
                         clip=(1, lv, cfg.clip_dim), sync=(1, ls, cfg.sync_dim))
    dit_off = 0
    dac_off = _round_up(dit_off + max(dit_bytes, _ALIGN))
    cond_off = _round_up(dac_off + max(dac_bytes, _ALIGN))
    n_cond = sum(_numel(s) for s in shapes.values()) * 4
    return BundleSpec(cond_off + _round_up(n_cond), dit_off, dit_bytes, dit_table, dac_off, dac_bytes, dac_table,
                      cond_off, shapes)


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class Bundle:
    """The broadcast buffer plus typed views into it."""

    def __init__(self, spec: BundleSpec, device):
        self.spec = spec
        self.buffer = torch.empty(spec.total, dtype=torch.uint8, device=device)

    def dit_region(self) -> torch.Tensor:
        s = self.spec
        return self.buffer[s.dit_off:s.dit_off + max(s.dit_bytes, _ALIGN)]

    def dac_region(self) -> torch.Tensor:
        s = self.spec
        return self.buffer[s.dac_off:s.dac_off + max(s.dac_bytes, _ALIGN)]

    def dit_arena(self) -> packers.Arena:
        return packers.Arena(self.spec.dit_bytes, self.spec.dit_table, self.buffer.device, buffer=self.dit_region())

    def dac_arena(self) -> packers.Arena:
        return packers.Arena(self.spec.dac_bytes, self.spec.dac_table, self.buffer.device, buffer=self.dac_region())

    def cond_views(self) -> Dict[str, torch.Tensor]:
        out, off = OrderedDict(), self.spec.cond_off
        for k, shape in self.spec.cond_shapes.items():
            n = _numel(shape) * 4
            out[k] = self.buffer[off:off + n].view(torch.float32).view(shape)
            off += n
        return out

    # ---- root side
    def fill(self, dit_packed: Dict[str, torch.Tensor], dac_packed: Dict[str, torch.Tensor],
             cond: Dict[str, torch.Tensor]):
        """Pack straight into the bundle (no second copy of the 10 GB arena on the root)."""
        for packed, arena in ((dit_packed, self.dit_arena()), (dac_packed, self.dac_arena())):
            total, table = packers.arena_layout(packed)
            if table != arena.table:
                raise ValueError("packed tensors do not match the layout derived from the configuration")
            for k, t in packed.items():
                arena.view(k).copy_(t)
        views = self.cond_views()
        for k, v in views.items():
            src = cond[k].to(torch.float32)
            if k in ("text", "uncond_text"):      # zero-pad / trim to the fixed text length
                T = v.shape[1]
                src = src[:, :T]
                v.zero_()
                v[:, :src.shape[1]].copy_(src)
            else:
                if tuple(src.shape) != tuple(v.shape):
                    raise ValueError(f"conditioning '{k}' has shape {tuple(src.shape)}, bundle expects {tuple(v.shape)}")
                v.copy_(src)


def broadcast_bundle(bundle: Bundle, src: int = 0) -> float:
    """THE collective of the data-parallel path: one broadcast of the whole bundle.  Returns its
    wall time in seconds (device-synchronised on both sides when the buffer lives on a GPU)."""
    on_gpu = bundle.buffer.is_cuda
    if on_gpu:
        torch.cuda.synchronize(bundle.buffer.device)
    t0 = time.perf_counter()
    dist.broadcast(bundle.buffer, src=src)
    if on_gpu:
        torch.cuda.synchronize(bundle.buffer.device)
    return time.perf_counter() - t0
