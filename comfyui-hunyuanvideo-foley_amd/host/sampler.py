"""Host orchestration of one sampling run - the MI355X counterpart of
`denoise_process_with_generator` (/utils.py:125-258): draw the noise like the reference does,
replicate / pad the conditioning, hand everything to libfoley_hip.so, decode with the DAC.

PyTorch here only allocates device memory, draws the CPU-generator noise and builds tiny index
tables; every FLOP of the hot path runs in the HIP library.
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import packers, tables
from .config import DAC48K, DACConfig, DiTConfig
from .runtime import FoleyContext, FoleyRuntimeError


class FoleyModel:
    """The HUNYUAN_MODEL socket payload: packed DiT weights resident on one GPU + its context.

    Mirrors what the reference sampler touches on its nn.Module: `.dtype`,
    `get_empty_clip_sequence`, `get_empty_sync_sequence` (hifi_foley.py:620-632).
    """

    def __init__(self, cfg: DiTConfig, dit_state: Dict[str, torch.Tensor], compute_dtype: torch.dtype,
                 device, dac_cfg: DACConfig = DAC48K, quantization: str = "none"):
        self.cfg, self.dac_cfg = cfg, dac_cfg
        self.dtype = compute_dtype
        self.device = torch.device(device)
        self.quantization = quantization
        # fp8 weight-only storage stays fp8 in HBM (bf16 / fp16 compute); `dit_state` then holds fp8-rounded values
        store = packers.FP8_DTYPES.get(quantization) if compute_dtype in packers.HALF_DTYPES else None
        packed = packers.pack_dit(dit_state, cfg, compute_dtype, weight_store=store)
        self.arena = packers.Arena.from_packed(packed, self.device)
        self._finish_init()

    @classmethod
    def from_arena(cls, cfg: DiTConfig, arena: "packers.Arena", compute_dtype: torch.dtype, device,
                   dac_cfg: DACConfig = DAC48K, quantization: str = "none") -> "FoleyModel":
        """Adopt an already packed arena (e.g. one received by broadcast on a non-root rank)."""
        self = cls.__new__(cls)
        self.cfg, self.dac_cfg, self.dtype, self.device = cfg, dac_cfg, compute_dtype, torch.device(device)
        self.quantization = quantization
        self.arena = arena
        self._finish_init()
        return self

    @classmethod
    def from_reference_state(cls, cfg: DiTConfig, dit_state: Dict[str, torch.Tensor], compute_dtype: torch.dtype, device,
                             dac_state: Dict[str, torch.Tensor], dac_cfg: DACConfig = DAC48K,
                             quantization: str = "none") -> "FoleyModel":
        """Load through the C ABI's reference-keyed path (foley_load_tensor): the library packs the checkpoint
        tensors on the device into its own arena - no host/packers.py involved.  `dit_state` is expected to hold
        the parameter values the reference module would hold (nodes.round_params / fp8_round_state_dict)."""
        self = cls.__new__(cls)
        self.cfg, self.dac_cfg, self.dtype, self.device = cfg, dac_cfg, compute_dtype, torch.device(device)
        self.quantization = quantization
        self.arena = None
        self.ctx = FoleyContext(cfg, dac_cfg, compute_dtype, self.device)
        fmt = {"fp8_e4m3fn": 1, "fp8_e5m2": 2}.get(quantization, 0) if compute_dtype in packers.HALF_DTYPES else 0
        self.ctx.load_reference_state([dit_state, dac_state], fmt)
        self.empty_clip_feat = dit_state["empty_clip_feat"].detach().to(self.device, torch.float32).reshape(1, -1)
        self.empty_sync_feat = dit_state["empty_sync_feat"].detach().to(self.device, torch.float32).reshape(1, -1)
        self._dac_arena = "ctx-owned"
        self._text_len_fixed = None
        return self

    def _finish_init(self):
        self.ctx = FoleyContext(self.cfg, self.dac_cfg, self.dtype, self.device)
        self.ctx.set_tensors((k, v) for k, v in self.arena.items() if not k.startswith("empty_"))
        self.empty_clip_feat = self.arena.view("empty_clip").view(1, -1)
        self.empty_sync_feat = self.arena.view("empty_sync").view(1, -1)
        self._dac_arena = None
        self._text_len_fixed = None

    def get_empty_clip_sequence(self, bs=None, len=None):
        e = self.empty_clip_feat
        return e.expand(len, -1) if bs is None else e.unsqueeze(0).expand(bs, len, -1)

    def get_empty_sync_sequence(self, bs=None, len=None):
        e = self.empty_sync_feat
        return e.expand(len, -1) if bs is None else e.unsqueeze(0).expand(bs, len, -1)

    def attach_dac(self, dac: "FoleyDAC"):
        if dac is None:
            if self._dac_arena != "ctx-owned":
                raise FoleyRuntimeError("no DAC decoder: pass a FoleyDAC or load it by reference key")
            return                                                 # decoder weights were loaded by reference key
        if self._dac_arena is not dac.arena:
            self.ctx.set_tensors(dac.arena.items())
            self._dac_arena = dac.arena

    def to(self, *a, **k):   # placement is fixed at load time; kept for reference-API compatibility
        return self


class FoleyDAC:
    """Packed DAC-VAE decoder (the `dac_model` entry of HUNYUAN_DEPS)."""

    sample_rate = 48000

    def __init__(self, dac_state: Dict[str, torch.Tensor], device, cfg: DACConfig = DAC48K):
        self.cfg = cfg
        self.sample_rate = cfg.sample_rate
        self.device = torch.device(device)
        packed = packers.pack_dac(dac_state, cfg)
        self.has_encoder = packers.has_dac_encoder(dac_state)
        if self.has_encoder:     # full checkpoints (the reference loads them strict=True) also carry the encoder
            packed.update(packers.pack_dac_encoder(dac_state, cfg))
        self.arena = packers.Arena.from_packed(packed, self.device)

    @classmethod
    def from_arena(cls, arena, device, cfg: DACConfig = DAC48K):
        self = cls.__new__(cls)
        self.cfg, self.sample_rate, self.device, self.arena = cfg, cfg.sample_rate, torch.device(device), arena
        self.has_encoder = "enc.in.w" in arena.table
        return self


def pad_or_trim_text(x: torch.Tensor, t_fixed: int) -> torch.Tensor:
    T = x.shape[1]
    if T == t_fixed:
        return x
    if T > t_fixed:
        return x[:, :t_fixed]
    return F.pad(x, (0, 0, 0, t_fixed - T))


def draw_noise(batch_size: int, channels: int, length: int, dtype: torch.dtype,
               generator: Optional[torch.Generator]) -> torch.Tensor:
    """prepare_latents_with_generator (utils.py:114-121): CPU generator, *model* dtype."""
    return torch.randn((batch_size, channels, int(length)), generator=generator, device="cpu", dtype=dtype)


def fp8_time_dtype(model):
    """fp8 type the timestep features are rounded through, or None: an fp8-wrapped reference model
    under bf16/fp16 autocast casts them to fp8 (embed_layers.py:134, golden g8)."""
    if getattr(model, "quantization", "none") == "none" or model.dtype == torch.float32:
        return None
    return torch.float8_e4m3fn if model.quantization == "fp8_e4m3fn" else torch.float8_e5m2


def build_plan(model: FoleyModel, visual_feats: Dict[str, torch.Tensor], text_feats: Dict[str, torch.Tensor],
               La: int, guidance_scale: float, steps: int, batch_size: int, sampler: str) -> dict:
    """Conditioning replication / padding / CFG stacking of utils.py:159-199 + the run's tables."""
    cfg, dev = model.cfg, model.device
    f32 = lambda t: t.to(device=dev, dtype=torch.float32)
    clip, sync = f32(visual_feats["siglip2_feat"]), f32(visual_feats["syncformer_feat"])
    text, unc = f32(text_feats["text_feat"]), f32(text_feats["uncond_text_feat"])
    if clip.shape[0] != 1 or sync.shape[0] != 1 or text.shape[0] != 1 or unc.shape[0] != 1:
        raise FoleyRuntimeError("conditioning tensors must have batch 1 (they are shared by all clips)")
    Lv, Ls = clip.shape[1], sync.shape[1]
    # two-bucket text length policy, sticky per model (utils.py:166-188)
    t_fixed = min(77 if text.shape[1] <= 77 else 128, cfg.text_len)
    model._text_len_fixed = max(model._text_len_fixed or 0, t_fixed)
    Lt = model._text_len_fixed
    text, unc = pad_or_trim_text(text, Lt), pad_or_trim_text(unc, Lt)
    if guidance_scale > 1.0:                      # [uncond ; cond]  (utils.py:193-195)
        ncfg = 2
        text_in = torch.cat([unc, text])
        clip_in = torch.cat([model.get_empty_clip_sequence(bs=1, len=Lv).float(), clip])
        sync_in = torch.cat([model.get_empty_sync_sequence(bs=1, len=Ls).float(), sync])
    else:
        ncfg, text_in, clip_in, sync_in = 1, text, clip, sync
    fp8_time = fp8_time_dtype(model)
    # The run's tables (schedule, solver coefficients, RoPE rows, position / up-sampling maps) depend on these scalars
    # only: built once per distinct run shape and kept on the device (host-side table building is milliseconds of
    # Python per call - more than the whole precompute on the GPU)
    key = (La, Lv, Ls, Lt, steps, sampler, float(cfg.flow_shift), cfg.time_freq_dim, str(fp8_time), str(dev))
    cache = model.__dict__.setdefault("_tables_cache", {})
    tabs = cache.get(key)
    if tabs is None:
        tb = tables.build_tables(La, Lv, Ls, Lt, steps, sampler, cfg.flow_shift, cfg.time_freq_dim, fp8_time=fp8_time)
        tabs = {k: tb[k].to(dev).contiguous() for k in ("t_feat", "rope_cos", "rope_sin", "pos_audio_self", "pos_visual_self",
                                                        "pos_linear", "sync_gather", "solver_coef")}
        if len(cache) >= 16:
            cache.pop(next(iter(cache)))
        cache[key] = tabs
    plan = {"ncfg": ncfg, "clips": batch_size, "La": La, "Lv": Lv, "Ls": Ls, "Lt": Lt, "n_iter": steps,
            "guidance": float(guidance_scale), "rope_len": tabs["rope_cos"].shape[0],
            "text": text_in.contiguous(), "clip": clip_in.contiguous(), "sync": sync_in.contiguous()}
    plan.update(tabs)
    return plan


def denoise_process_with_generator(visual_feats, text_feats, audio_len_in_s, model: FoleyModel, dac: FoleyDAC,
                                   guidance_scale: float, num_inference_steps: int, batch_size: int, sampler: str,
                                   generator: Optional[torch.Generator] = None, use_graph: bool = True,
                                   progress: Optional[Callable[[int, int], None]] = None,
                                   return_latents: bool = False, noise: Optional[torch.Tensor] = None,
                                   _abort_event: Optional[threading.Event] = None):
    """Same contract as the reference function of this name (utils.py:125-258):
    returns (audio [bs, 1, T] fp32 on the model's device, sample_rate)."""
    cfg = model.cfg
    La = int(audio_len_in_s * cfg.frame_rate)
    if noise is None:
        noise = draw_noise(batch_size, cfg.latent_dim, La, model.dtype, generator)
    latents = noise.to(device=model.device, dtype=torch.float32).contiguous()
    plan = build_plan(model, visual_feats, text_feats, La, guidance_scale, num_inference_steps, batch_size, sampler)
    model.attach_dac(dac)
    model.ctx.prepare(plan)
    if _abort_event is not None and _abort_event.is_set():      # another replica of a data-parallel run failed meanwhile
        raise FoleyRuntimeError("sampling aborted: another replica failed")
    model.ctx.sample(latents, use_graph=use_graph, progress=progress)
    audio = model.ctx.dac_decode(latents)
    # (the reference's "trim to exact length" slices the size-1 channel axis: a no-op)
    sr = dac.sample_rate if dac is not None else model.dac_cfg.sample_rate
    if return_latents:
        return audio, sr, latents
    return audio, sr


# ----------------------------------------------------------------------------- node-level data parallelism
def replicate(model: FoleyModel, dac: Optional[FoleyDAC], devices: Sequence) -> List:
    """One (FoleyModel, FoleyDAC) pair per device of `devices` for clip-level data parallelism inside ONE
    process (the ComfyUI case: a single prompt-worker process that sees all GPUs of the node).  The pair of the
    device the model already lives on is reused; all other devices receive the packed DiT arena and the DAC arena in
    ONE grouped RCCL broadcast over xGMI (`runtime.bcast_local` -> foley_bcast_local: ncclCommInitAll over the device
    list, one ncclBroadcast per (device, arena) inside ncclGroupStart / End) - the in-process counterpart of the single
    broadcast of host/distributed.py; `model.last_broadcast_s` keeps its wall time.  A second context on a device that
    already holds one (the 1-GPU test set-up: RCCL communicators want distinct devices) takes a device-local copy.
    Replicas are cached on the model."""
    from . import runtime as _rt
    if model.arena is None:
        raise FoleyRuntimeError("replicate() needs a model packed by the host packers (an arena it can copy)")
    cache = model.__dict__.setdefault("_replicas", {})
    slots, pending = [], []                         # per requested device: ("own", ) | ("cached", key) ; new replicas to fill
    for d in devices:
        d = torch.device(d)
        if d.type != "cuda":
            raise FoleyRuntimeError("replicas live on HIP devices")
        if d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        if d == model.device and not any(sl[0] == "own" for sl in slots):
            slots.append(("own", None))
            continue
        key = (str(d), sum(1 for sl in slots if sl[1] is not None and sl[1][0] == str(d)) + 0)   # the same device twice = two contexts on it
        slots.append(("cached", key))
        if key not in cache and key not in [p[0] for p in pending]:
            pending.append((key, d))
    if pending:
        bufs = {}
        for key, d in pending:
            bufs[key] = (torch.empty_like(model.arena.buffer, device=d),
                         torch.empty_like(dac.arena.buffer, device=d) if dac is not None else None)
        # one grouped broadcast to the first new replica of every device that does not hold the model yet
        first = {}
        for key, d in pending:
            if d != model.device and d not in first:
                first[d] = key
        model.last_broadcast_s = 0.0
        model.last_broadcast_path = None           # 'rccl' | 'peer-copy' | None (nothing crossed a device boundary)
        if first:
            per_dev = [[model.arena.buffer.view(torch.uint8).reshape(-1)] + ([dac.arena.buffer.view(torch.uint8).reshape(-1)] if dac is not None else [])]
            for d, key in first.items():
                per_dev.append([bufs[key][0].view(torch.uint8).reshape(-1)] + ([bufs[key][1].view(torch.uint8).reshape(-1)] if dac is not None else []))
            try:
                model.last_broadcast_s = _rt.bcast_local(per_dev)
                model.last_broadcast_path = "rccl"
            except FoleyRuntimeError as e:          # no librccl in the process / ncclCommInitAll refused the device list
                import logging
                import time as _time
                logging.getLogger("foley_amd").warning("grouped RCCL broadcast unavailable (%s): falling back to peer copies", e)
                t0 = _time.perf_counter()
                for dst in per_dev[1:]:
                    for s_buf, d_buf in zip(per_dev[0], dst):
                        d_buf.copy_(s_buf, non_blocking=True)
                for d in first:
                    torch.cuda.synchronize(d)
                model.last_broadcast_s = _time.perf_counter() - t0
                model.last_broadcast_path = "peer-copy"    # so no record attributes this time to RCCL
        for key, d in pending:                      # further contexts on a device: device-local copies of what is there already
            if first.get(d) == key:
                continue
            src = (model.arena.buffer, dac.arena.buffer if dac is not None else None) if d == model.device else bufs[first[d]]
            bufs[key][0].copy_(src[0])
            if dac is not None:
                bufs[key][1].copy_(src[1])
        for key, d in pending:
            arena = packers.Arena(model.arena.buffer.numel(), model.arena.table, d, buffer=bufs[key][0])
            m = FoleyModel.from_arena(model.cfg, arena, model.dtype, d, dac_cfg=model.dac_cfg, quantization=model.quantization)
            m._text_len_fixed = model._text_len_fixed      # the sticky text bucket is per MODEL in the reference (utils.py:166-188)
            dd = None
            if dac is not None:
                dd = FoleyDAC.from_arena(packers.Arena(dac.arena.buffer.numel(), dac.arena.table, d, buffer=bufs[key][1]), d, dac.cfg)
            cache[key] = (m, dd)
    return [(model, dac) if kind == "own" else cache[key] for kind, key in slots]


def denoise_process_multi(visual_feats, text_feats, audio_len_in_s, replicas: Sequence, guidance_scale: float,
                          num_inference_steps: int, batch_size: int, sampler: str,
                          generator: Optional[torch.Generator] = None, use_graph: bool = True,
                          progress: Optional[Callable[[int, int], None]] = None, return_latents: bool = False):
    """`denoise_process_with_generator` with the clips of the batch sharded over `replicas` (the pairs
    `replicate()` returns), one host thread per GPU.  Clips are independent (reference utils.py:159-199: the
    batch only repeats the conditioning), so there is no collective: the noise of the WHOLE batch is drawn once
    from `generator` exactly like the single-GPU call does and sliced (`distributed.shard_range`), which makes the
    result bit-identical to one GPU running the full batch whenever the shards use the same tile shapes - and
    equal to bf16 / fp32 accuracy otherwise.  Returns (audio [bs, 1, T] fp32 on the first replica's device, sr)."""
    from .distributed import shard_range
    if not replicas:
        raise FoleyRuntimeError("no replicas")
    m0 = replicas[0][0]
    cfg = m0.cfg
    La = int(audio_len_in_s * cfg.frame_rate)
    # The sticky text bucket is ONE value per model in the reference (utils.py:166-188): every replica pads this batch to the
    # same length - the largest bucket any replica has seen so far or this prompt asks for - whichever shards it gets.
    t_now = min(77 if text_feats["text_feat"].shape[1] <= 77 else 128, cfg.text_len)
    sticky = max([t_now] + [m._text_len_fixed or 0 for m, _ in replicas])
    for m, _ in replicas:
        m._text_len_fixed = sticky
    noise = draw_noise(batch_size, cfg.latent_dim, La, m0.dtype, generator)
    shards = [shard_range(batch_size, r, len(replicas)) for r in range(len(replicas))]
    results: List = [None] * len(replicas)
    errors: List = []
    # The conditioning tensors may still be in flight on the CALLER's streams (the CLAP / SigLIP2 / Synchformer kernels of
    # the node run on the default stream of their device): every worker runs on a fresh non-blocking stream, which orders
    # nothing against them by itself - one event per producing device, recorded here on the calling thread, and waited for
    # by each worker's stream before it touches a feature.
    ready = []
    for dv in {t.device for t in list(visual_feats.values()) + list(text_feats.values()) if torch.is_tensor(t) and t.is_cuda}:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dv))
        ready.append(ev)

    failed = threading.Event()              # set by the first failing worker; the others do not start (or stop at their next iteration)
    running = [False] * len(replicas)
    fail_lock = threading.Lock()

    def work(r):
        lo, hi = shards[r]
        if hi <= lo:
            return
        model, dac = replicas[r]
        try:
            stream = torch.cuda.Stream(model.device)
            with torch.cuda.device(model.device), torch.cuda.stream(stream):
                for ev in ready:
                    stream.wait_event(ev)
                if failed.is_set():
                    return
                running[r] = True
                results[r] = denoise_process_with_generator(
                    visual_feats, text_feats, audio_len_in_s, model, dac, guidance_scale, num_inference_steps, hi - lo,
                    sampler, use_graph=use_graph, noise=noise[lo:hi], return_latents=True,
                    progress=progress if r == 0 else None, _abort_event=failed)
                torch.cuda.current_stream().synchronize()
        except Exception as e:          # surfaced on the calling thread
            running[r] = False          # FIRST: a second failing (or aborted) worker must not keep the first one waiting on it
            with fail_lock:
                first_failure = not failed.is_set()
                errors.append(e)
                failed.set()            # replicas still in prepare / warm-up see it before their loop starts (foley_sample
            if first_failure:           # clears a stale abort request at entry); the ones inside the loop stop at the next iteration.
                import time as _time    # Only the FIRST failing worker asks, and keeps asking until every other worker has left
                while True:             # (closes the check-then-clear window); workers that fail because of the abort just leave.
                    busy = [i for i in range(len(replicas)) if i != r and running[i]]
                    for i in busy:
                        try:
                            replicas[i][0].ctx.abort()
                        except Exception:   # noqa: BLE001 - best effort; the first error is the one reported
                            pass
                    if not busy:
                        break
                    _time.sleep(0.005)
        finally:
            running[r] = False

    threads = [threading.Thread(target=work, args=(r,), name=f"foley-dp-{r}") for r in range(len(replicas))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    done = [x for x in results if x is not None]
    audio = torch.cat([x[0].to(m0.device) for x in done])
    sr = done[0][1]
    if return_latents:
        return audio, sr, torch.cat([x[2].to(m0.device) for x in done])
    return audio, sr
