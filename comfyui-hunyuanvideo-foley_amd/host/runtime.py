"""ctypes binding of libfoley_hip.so (include/foley_hip.h).  PyTorch is only the allocator /
stream provider here: every call passes raw device pointers + the current HIP stream.

There is deliberately NO fallback: if the shared library is missing or a call fails, a
`FoleyRuntimeError` is raised - the HIP path is the only implementation of the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfoley_hip.so")

DT_F32, DT_BF16, DT_I32, DT_F8E4M3, DT_F8E5M2, DT_F16 = 0, 1, 2, 3, 4, 5
_TORCH2DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.int32: DT_I32, torch.float8_e4m3fn: DT_F8E4M3,
             torch.float8_e5m2: DT_F8E5M2, torch.float16: DT_F16}

EPI_STORE_F32, EPI_STORE_T, EPI_SILU_T, EPI_GELU_T, EPI_SILUGATE_T, EPI_GATE_RES, EPI_DAC = range(7)


class FoleyRuntimeError(RuntimeError):
    pass


class FoleyConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "depth_triple", "depth_single", "hidden", "heads", "mlp_hidden", "conv_hidden", "sync_hidden",
        "cond_dim", "clip_dim", "sync_dim", "latent_dim", "time_freq_dim", "compute_dtype", "dac_dim",
        "dac_n_rates")] + [("dac_rates", C.c_int32 * 8), ("dac_dilations", C.c_int32 * 3)]


class FoleyPlanC(C.Structure):
    _fields_ = [
        ("ncfg", C.c_int32), ("clips", C.c_int32), ("La", C.c_int32), ("Lv", C.c_int32), ("Ls", C.c_int32),
        ("Lt", C.c_int32), ("n_iter", C.c_int32), ("guidance", C.c_float),
        ("text", C.c_void_p), ("clip", C.c_void_p), ("sync", C.c_void_p), ("t_feat", C.c_void_p),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("rope_len", C.c_int32),
        ("pos_audio_self", C.c_void_p), ("pos_visual_self", C.c_void_p), ("pos_linear", C.c_void_p),
        ("sync_gather", C.c_void_p), ("solver_coef", C.c_void_p),
    ]


class RowBcastC(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ld", C.c_int64), ("mode", C.c_int32), ("rows_per_cfg", C.c_int32),
                ("L", C.c_int32), ("Ls", C.c_int32), ("period", C.c_int32), ("periodic_cfgs", C.c_int32)]


class QkvSplitDescC(C.Structure):
    _fields_ = [
        ("L", C.c_int32), ("H", C.c_int32), ("nK", C.c_int32),
        ("gain", C.c_void_p * 3), ("pos", C.c_void_p * 3), ("dst", C.c_void_p * 3),
        ("out_dtype", C.c_int32), ("vt_pitch", C.c_int32), ("S_tot", C.c_int32), ("tok_off", C.c_int32),
        ("eps", C.c_float), ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p),
        ("attn_k", C.c_void_p), ("attn_vt", C.c_void_p), ("attn_out", C.c_void_p),
        ("attn_skv", C.c_int32), ("attn_pitch", C.c_int32), ("attn_bdiv", C.c_int32),
        ("attn_fused", C.POINTER(C.c_int32)),
    ]


def _qkv_fused(self) -> bool:
    return bool(getattr(self, "_flag", C.c_int32(0)).value)


QkvSplitDescC.fused = _qkv_fused


class GemmDescC(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int64),
        ("segV", C.c_int32), ("segS", C.c_int32), ("taps", C.c_int32), ("tapC", C.c_int32),
        ("dil", C.c_int32), ("tap0", C.c_int32),
        ("out0", C.c_void_p), ("out1", C.c_void_p),
        ("osegV", C.c_int32), ("out_seg", C.c_int64), ("out_row", C.c_int64), ("out_shift", C.c_int64),
        ("out_check", C.c_int32),
        ("rb", RowBcastC), ("res", C.c_void_p), ("alpha", C.c_void_p), ("alphaC", C.c_int32),
        ("dtype", C.c_int32), ("epilogue", C.c_int32), ("tile", C.c_int32), ("ksplit", C.c_int32),
        ("partials", C.c_void_p), ("partial_slabs", C.c_int32), ("ksplit_used", C.POINTER(C.c_int32)),
        ("qkv", C.POINTER(QkvSplitDescC)), ("rstride", C.c_int32), ("ldw", C.c_int64), ("wfmt", C.c_int32),
        ("partial_dtype", C.c_int32), ("gelu_erf", C.c_int32),
    ]


class ProfEntryC(C.Structure):
    _fields_ = [("label", C.c_char * 80), ("calls", C.c_int32), ("total_ms", C.c_float), ("flop", C.c_double),
                ("bytes", C.c_double), ("kernel", C.c_char * 200)]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_int32, C.c_int32, C.c_void_p)
ABI_VERSION = 12

_SIGNATURES = {
    "foley_abi_version": (C.c_uint32, []),
    "foley_last_error": (C.c_char_p, []),
    "foley_ctx_create": (C.c_int, [C.c_int, C.POINTER(FoleyConfigC), C.POINTER(C.c_void_p)]),
    "foley_ctx_destroy": (None, [C.c_void_p]),
    "foley_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "foley_weights_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "foley_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p]),
    "foley_weights_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "foley_weights_arena": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "foley_weights_mark_received": (C.c_int, [C.c_void_p]),
    "foley_bcast_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "foley_bcast_local": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "foley_prepare": (C.c_int, [C.c_void_p, C.POINTER(FoleyPlanC), C.c_void_p]),
    "foley_dit_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "foley_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, PROGRESS_CB, C.c_void_p, C.c_void_p]),
    "foley_abort": (C.c_int, [C.c_void_p]),
    "foley_dac_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "foley_dac_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "foley_last_elapsed_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "foley_profile_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(ProfEntryC), C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p]),
    "foley_op_gemm": (C.c_int, [C.POINTER(GemmDescC), C.c_void_p]),
    "foley_op_attention": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                                        C.c_void_p]),
    "foley_op_attention_hd": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                                           C.c_void_p]),
    "foley_op_ln_mod_pending": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(RowBcastC),
                                          C.POINTER(RowBcastC), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.POINTER(RowBcastC), C.c_void_p]),
    "foley_op_ln_mod_pending2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(RowBcastC),
                                           C.POINTER(RowBcastC), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(RowBcastC), C.c_void_p]),
    "foley_op_ln_mod": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(RowBcastC),
                                  C.POINTER(RowBcastC), C.c_void_p, C.c_int, C.c_void_p]),
    "foley_op_attention_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "foley_op_qkv_regroup": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "foley_op_resize_aa_u8": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p]),
    "foley_op_qkv_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "foley_op_solver_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_void_p,
                                                                          C.c_void_p, C.c_int, C.c_void_p]),
    "foley_op_latent_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p]),
    "foley_op_dac_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_LIB: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libfoley_hip.so and type its entry points; raises if it is not built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.environ.get("FOLEY_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise FoleyRuntimeError(
            f"{p} not found - build it with `python __graft_entry__.py build` (hipcc, gfx950). "
            "There is no CPU / PyTorch fallback for the Foley sampling path.")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export the ABI
        fn.restype = res
        fn.argtypes = args
    if lib.foley_abi_version() != ABI_VERSION:
        raise FoleyRuntimeError("libfoley_hip.so ABI version mismatch")
    if path is None:
        _LIB = lib
    return lib


def _check(lib, rc: int, what: str):
    if rc != 0:
        msg = lib.foley_last_error()
        raise FoleyRuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise FoleyRuntimeError("libfoley_hip.so needs device tensors (got a CPU tensor)")
    if not t.is_contiguous():
        raise FoleyRuntimeError("tensor must be contiguous")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dt_of(t: torch.Tensor) -> int:
    return _TORCH2DT[t.dtype]


def make_config(dit, dac, compute_dtype: torch.dtype) -> FoleyConfigC:
    c = FoleyConfigC()
    c.depth_triple, c.depth_single, c.hidden, c.heads = dit.depth_triple, dit.depth_single, dit.hidden, dit.heads
    c.mlp_hidden, c.conv_hidden, c.sync_hidden = dit.mlp_hidden, dit.conv_hidden, dit.sync_hidden
    c.cond_dim, c.clip_dim, c.sync_dim = dit.cond_dim, dit.clip_dim, dit.sync_dim
    c.latent_dim, c.time_freq_dim = dit.latent_dim, dit.time_freq_dim
    c.compute_dtype = _TORCH2DT[compute_dtype]
    c.dac_dim, c.dac_n_rates = dac.decoder_dim, len(dac.rates)
    for i, r in enumerate(dac.rates):
        c.dac_rates[i] = r
    for i, d in enumerate(dac.dilations):
        c.dac_dilations[i] = d
    return c


class FoleyContext:
    """Owns one `foley_ctx` on one GPU plus references to every tensor registered with it."""

    def __init__(self, dit_cfg, dac_cfg, compute_dtype: torch.dtype, device: torch.device):
        self.lib = load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise FoleyRuntimeError("FoleyContext needs a HIP device (torch device type 'cuda')")
        self.dit_cfg, self.dac_cfg, self.compute_dtype = dit_cfg, dac_cfg, compute_dtype
        self._cfg = make_config(dit_cfg, dac_cfg, compute_dtype)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _check(self.lib, self.lib.foley_ctx_create(idx, C.byref(self._cfg), C.byref(h)), "foley_ctx_create")
        self._h = h
        self._keep: Dict[str, torch.Tensor] = {}
        self._plan_keep = None

    def close(self):
        if getattr(self, "_h", None):
            self.lib.foley_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def set_tensor(self, name: str, t: torch.Tensor):
        shape = (C.c_int64 * t.dim())(*t.shape)
        _check(self.lib, self.lib.foley_set_tensor(self._h, name.encode(), _ptr(t), dt_of(t), t.dim(), shape),
               f"foley_set_tensor({name})")
        self._keep[name] = t

    def set_tensors(self, items):
        for k, t in items:
            self.set_tensor(k, t)

    # ---- reference-keyed loading (the library packs on the device; weights.hip)
    def load_reference_state(self, state_dicts, weight_format: int = 0) -> int:
        """Pack reference state dict(s) (DiT and/or DAC, device or CPU tensors) through the C ABI.  Returns the number of
        tensors the sampling path ignored (DAC encoder etc.)."""
        _check(self.lib, self.lib.foley_weights_begin(self._h, int(weight_format)), "foley_weights_begin")
        ignored = 0
        with torch.cuda.device(self.device):
            for sd in state_dicts:
                for k, v in sd.items():
                    if not isinstance(v, torch.Tensor) or not v.is_floating_point():
                        continue
                    t = v.detach().to(self.device).contiguous()
                    shape = (C.c_int64 * t.dim())(*t.shape)
                    rc = self.lib.foley_load_tensor(self._h, k.encode(), t.data_ptr(), dt_of(t), t.dim(), shape, _stream())
                    if rc == 1:
                        ignored += 1
                    else:
                        _check(self.lib, rc, f"foley_load_tensor({k})")
                    torch.cuda.current_stream().synchronize()       # `t` is only borrowed for the call
            _check(self.lib, self.lib.foley_weights_end(self._h, _stream()), "foley_weights_end")
        return ignored

    def weights_arena(self):
        p, n = C.c_void_p(), C.c_uint64()
        _check(self.lib, self.lib.foley_weights_arena(self._h, C.byref(p), C.byref(n)), "foley_weights_arena")
        return int(p.value), int(n.value)

    def bcast_weights(self, nccl_comm: int, root: int = 0):
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.foley_bcast_weights(self._h, C.c_void_p(nccl_comm), int(root), _stream()),
                   "foley_bcast_weights")

    # ---- run
    def prepare(self, plan: dict):
        """plan: ncfg, clips, La, Lv, Ls, Lt, n_iter, guidance + device tensors (see foley_plan)."""
        p = FoleyPlanC()
        for k in ("ncfg", "clips", "La", "Lv", "Ls", "Lt", "n_iter", "rope_len"):
            setattr(p, k, int(plan[k]))
        p.guidance = float(plan["guidance"])
        for k in ("text", "clip", "sync", "t_feat", "rope_cos", "rope_sin", "pos_audio_self", "pos_visual_self",
                  "pos_linear", "sync_gather", "solver_coef"):
            setattr(p, k, _ptr(plan[k]))
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.foley_prepare(self._h, C.byref(p), _stream()), "foley_prepare")
        self._plan_keep = plan    # the ctx borrows the plan's tables until the next prepare
        self.plan = plan

    def dit_forward(self, latents: torch.Tensor, it: int) -> torch.Tensor:
        pl = self.plan
        out = torch.empty(pl["ncfg"] * pl["clips"] * pl["La"], self.dit_cfg.latent_dim, dtype=torch.float32,
                          device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.foley_dit_forward(self._h, _ptr(latents), int(it), _ptr(out), _stream()),
                   "foley_dit_forward")
        return out

    def sample(self, latents: torch.Tensor, use_graph: bool = False, progress=None) -> torch.Tensor:
        """In-place denoising of `latents` [clips, C, La] fp32.  An exception raised by `progress` (ComfyUI's interrupt:
        comfy.utils.ProgressBar.update raises inside the reference's loop, utils.py:247) stops the loop after the current
        iteration and propagates to the caller, like it does there."""
        raised = []

        def _cb(i, n, _u):
            try:
                progress(i, n)
            except BaseException as e:          # noqa: BLE001 - re-raised below, outside the C frame
                if not raised:
                    raised.append(e)
                self.lib.foley_abort(self._h)

        cb = PROGRESS_CB(_cb) if progress else PROGRESS_CB()
        with torch.cuda.device(self.device):
            rc = self.lib.foley_sample(self._h, _ptr(latents), int(use_graph), cb, None, _stream())
        if raised:
            raise raised[0]
        _check(self.lib, rc, "foley_sample")
        return latents

    def abort(self) -> None:
        """Ask a foley_sample running on another thread to stop after its current iteration (it raises FoleyRuntimeError)."""
        _check(self.lib, self.lib.foley_abort(self._h), "foley_abort")

    def dac_decode(self, latents: torch.Tensor) -> torch.Tensor:
        clips, _c, T = latents.shape
        hop = self.dac_cfg.hop
        wave = torch.empty(clips, 1, T * hop, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.foley_dac_decode(self._h, _ptr(latents), clips, T, _ptr(wave), _stream()),
                   "foley_dac_decode")
        return wave

    def dac_encode(self, wave: torch.Tensor) -> torch.Tensor:
        """DAC.encode (continuous=True): wave [clips, 1, T] fp32 on the GPU -> posterior parameters
        [clips, 2*latent, T'] (rows [:latent] mean, [latent:] logvar).  The waveform is right-padded
        to a multiple of the hop like DAC.preprocess (dac.py:225-234)."""
        cfg = self.dac_cfg
        rates = list(cfg.encoder_rates)
        hop = 1
        for r in rates:
            hop *= r
        clips, _one, T = wave.shape
        Tp = -(-T // hop) * hop
        if Tp != T:
            wave = torch.nn.functional.pad(wave, (0, Tp - T))
        wave = wave.contiguous().float()
        out = torch.empty(clips, 2 * cfg.latent_dim, Tp // hop, dtype=torch.float32, device=self.device)
        arr = (C.c_int32 * len(rates))(*rates)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.foley_dac_encode(self._h, _ptr(wave), clips, Tp, cfg.encoder_dim, arr, len(rates),
                                                       _ptr(out), _stream()), "foley_dac_encode")
        return out

    def profile_forward(self, latents: torch.Tensor, it: int = 0, repeats: int = 2):
        """Per-op HIP-event profile of the eager DiT forward: list of dicts (label, calls_per_forward, avg_us =
        the dispatch's own start->stop time, flop / bytes per launch) + the cost of an empty event bracket in us."""
        cap = 64
        arr = (ProfEntryC * cap)()
        n, br = C.c_int(0), C.c_float(0.0)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.foley_profile_forward(self._h, _ptr(latents), int(it), int(repeats), arr, cap,
                                                            C.byref(n), C.byref(br), _stream()), "foley_profile_forward")
        out = []
        for e in arr[:n.value]:
            calls = max(e.calls, 1)
            out.append({"label": e.label.decode(), "kernel": e.kernel.decode(errors="replace"), "calls_per_forward": e.calls / repeats,
                        "avg_us": 1e3 * e.total_ms / calls,
                        "flop_per_launch": e.flop / calls, "bytes_per_launch": e.bytes / calls})
        return out, 1e3 * br.value

    def last_elapsed_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib, self.lib.foley_last_elapsed_ms(self._h, C.byref(ms)), "foley_last_elapsed_ms")
        return float(ms.value)


# ----------------------------------------------------------------------------- op-level wrappers (tests, microbench)
def bcast_local(buffers_per_device: Sequence[Sequence[torch.Tensor]]) -> float:
    """foley_bcast_local: `buffers_per_device[d][i]` is buffer i (uint8, same size on every device) on the d-th device; the
    root's (d = 0) contents reach every other device in ONE grouped RCCL launch.  Returns the wall time in seconds.  RCCL is
    the copy PyTorch ships (loaded into the process here if torch.distributed has not done so yet)."""
    import time
    lib = load_library()
    ndev, nbuf = len(buffers_per_device), len(buffers_per_device[0])
    devs = [b[0].device.index for b in buffers_per_device]
    if len(set(devs)) != ndev:
        raise FoleyRuntimeError("bcast_local: one entry per distinct device")
    for b in buffers_per_device:
        if len(b) != nbuf or any(t.device != b[0].device or not t.is_contiguous() for t in b):
            raise FoleyRuntimeError("bcast_local: every device carries the same buffers")
    rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(rccl):
        C.CDLL(rccl, mode=C.RTLD_GLOBAL)
    dev_arr = (C.c_int * ndev)(*devs)
    ptrs = (C.c_void_p * (ndev * nbuf))(*[t.data_ptr() for b in buffers_per_device for t in b])
    sizes = (C.c_uint64 * nbuf)(*[t.numel() * t.element_size() for t in buffers_per_device[0]])
    for d in devs:
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    _check(lib, lib.foley_bcast_local(ndev, dev_arr, nbuf, ptrs, sizes), "foley_bcast_local")
    return time.perf_counter() - t0


def rowbcast(t: Optional[torch.Tensor], mode: int = 0, rows_per_cfg: int = 1, L: int = 1,
             ld: Optional[int] = None, Ls: int = 0, period: int = 0, periodic_cfgs: int = 0) -> RowBcastC:
    """mode 2: `t` holds Ls rows per cfg; token l of a clip reads row nearest_exact(l) (tables.nearest_exact_index).  period > 0:
    only `period` rows per cfg are stored - for every cfg half, or (periodic_cfgs = k > 0) for the first k halves only, the
    others following with all their Ls rows."""
    r = RowBcastC()
    r.Ls, r.period, r.periodic_cfgs = Ls, period, periodic_cfgs
    if t is not None and not t.is_cuda:
        raise FoleyRuntimeError("row-broadcast operand must live on the GPU")
    r.p = t.data_ptr() if t is not None else None     # views allowed: rows are addressed through `ld`
    r.ld = int(ld if ld is not None else (t.shape[-1] if t is not None else 0))
    r.mode, r.rows_per_cfg, r.L = mode, rows_per_cfg, L
    r._keepalive = t    # the struct only carries a raw pointer: keep the tensor alive with it
    return r


def op_gemm(A, W, bias=None, *, M=None, epilogue=EPI_STORE_F32, out0=None, out1=None, ldc=None, conv=None,
            convT=None, rb: Optional[RowBcastC] = None, res=None, alpha=None, alphaC=1, tile=0, ksplit=0,
            partials=None, qkv: Optional["QkvSplitDescC"] = None, sconv=None, lda: Optional[int] = None,
            ldw: Optional[int] = None, NK=None, gelu_erf: bool = False, vrows=None) -> int:
    """Thin wrapper over foley_op_gemm.  conv=(seg, C, taps, dil) ; convT=(Tin, Cin, stride, Cout) ; vrows=(segV, segS) with M ;
    sconv=(Tin, Cin, stride): strided conv k=2*stride, pad ceil(stride/2) over clips of Tin rows.
    partials: fp32 [slabs, M, N] workspace for the deferred split-K of the gated-residual epilogue.
    Returns the K split the launcher used."""
    lib = load_library()
    d = GemmDescC()
    N, K = NK if NK is not None else W.shape      # NK: logical shape when W / A are row-padded storage (lda / ldw)
    d.A, d.W, d.bias = _ptr(A), _ptr(W), _ptr(bias) if bias is not None else None
    d.N, d.K = N, K
    if W.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):      # fp8 weight storage, bf16 (or fp16) activations
        d.dtype, d.wfmt = dt_of(A) if A.dtype == torch.float16 else DT_BF16, (1 if W.dtype == torch.float8_e4m3fn else 2)
    else:
        d.dtype = dt_of(W)
    d.epilogue, d.tile, d.ksplit = epilogue, tile, ksplit
    d.gelu_erf = 1 if gelu_erf else 0
    if convT is not None:
        Tin, Cin, s, Cout = convT
        clips = A.numel() // (Tin * Cin)
        d.M, d.lda = clips * (Tin + 1), Cin
        d.segV, d.segS, d.taps, d.tapC, d.dil, d.tap0 = Tin + 1, Tin, 2, Cin, 1, -1
        d.osegV, d.out_seg, d.out_row = Tin + 1, Tin * s * Cout, s * Cout
        d.out_shift, d.out_check = -((s + 1) // 2) * Cout, 1
    elif sconv is not None:
        Tin, Cin, st_ = sconv
        clips = A.numel() // (Tin * Cin)
        Tout = Tin // st_
        d.M, d.lda = clips * Tout, Cin
        d.segV, d.segS, d.taps, d.tapC, d.dil, d.tap0 = Tout, Tin, 2 * st_, Cin, 1, -((st_ + 1) // 2)
        d.rstride = st_
        d.osegV, d.out_seg, d.out_row, d.out_shift, d.out_check = d.M, 0, (ldc or N), 0, 0
    elif conv is not None:
        seg, Cc, taps, dil = conv
        d.M = A.numel() // Cc if M is None else M
        d.lda = Cc
        d.segV, d.segS, d.taps, d.tapC, d.dil, d.tap0 = seg, seg, taps, Cc, dil, -((taps - 1) // 2) * dil
        d.osegV, d.out_seg, d.out_row, d.out_shift, d.out_check = d.M, 0, (ldc or N), 0, 0
    else:
        d.M = A.shape[0] if M is None else M
        d.lda = K
        d.segV, d.segS, d.taps, d.tapC, d.dil, d.tap0 = d.M, d.M, 1, K, 1, 0
        d.osegV, d.out_seg, d.out_row, d.out_shift, d.out_check = d.M, 0, (ldc or N), 0, 0
    if vrows is not None:      # virtual rows: row r of the product reads source row (r // segV) * segS + r % segV
        d.segV, d.segS = vrows
    if epilogue == EPI_SILUGATE_T and ldc is None and convT is None:
        d.out_row = N // 2
    if lda is not None:
        d.lda = lda
    if ldw is not None:
        d.ldw = ldw
    d.out0, d.out1 = _ptr(out0) if out0 is not None else None, _ptr(out1) if out1 is not None else None
    if rb is not None:
        d.rb = rb
    d.res = _ptr(res) if res is not None else None
    d.alpha = _ptr(alpha) if alpha is not None else None
    d.alphaC = alphaC
    used = C.c_int32(1)
    d.ksplit_used = C.pointer(used)
    if partials is not None:      # fp32 slabs, or slabs in the (16-bit) operand dtype
        d.partials, d.partial_slabs = _ptr(partials), partials.shape[0]
        d.partial_dtype = 0 if partials.dtype == torch.float32 else dt_of(partials)
    if qkv is not None:
        d.qkv = C.pointer(qkv)
    _check(lib, lib.foley_op_gemm(C.byref(d), _stream()), "foley_op_gemm")
    return int(used.value)


def op_attention(q, k, v, outA, outB, split: int, kv_bdiv: int = 1):
    """fp32 q/k/v [B,H,S,hd], or bf16 / fp16 q/k [B,H,S,hd] with v transposed [B,H,hd,pitch]; hd = 128 or 64."""
    lib = load_library()
    Bq, H, Sq, hd = q.shape
    Skv = k.shape[2]
    vt_pitch = v.shape[3] if q.dtype in (torch.bfloat16, torch.float16) else 0
    _check(lib, lib.foley_op_attention_hd(_ptr(q), _ptr(k), _ptr(v), dt_of(q), vt_pitch, Bq, H, Sq, Skv, kv_bdiv,
                                          _ptr(outA), _ptr(outB), split, dt_of(outB), hd, _stream()), "foley_op_attention_hd")


def op_attention_scatter(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out_rows: torch.Tensor, out: torch.Tensor,
                         grp_q: int = 0, grp_kv: int = 0) -> None:
    """Attention at head_dim 64 (operands as op_qkv_regroup returns them) whose query (g, t) is written to row out_rows[g, t] of
    out [rows, H*64] (foley_op_attention_scatter).  grp_q / grp_kv > 0 (16-bit operands): block-diagonal - each sequence is a pack
    of groups of grp_q queries that attend their own grp_kv keys only."""
    lib = load_library()
    G, H, Sq, hd = q.shape
    half = q.dtype in (torch.bfloat16, torch.float16)
    Skv = k.shape[2]
    if hd != 64 or out_rows.dtype != torch.int32 or out_rows.numel() != G * Sq or out.dim() != 2 or out.shape[1] != H * 64 or \
            not out.is_contiguous() or not out_rows.is_contiguous() or out_rows.device != q.device or out.device != q.device or \
            k.shape[:2] != q.shape[:2] or not (q.is_contiguous() and k.is_contiguous()
                                                                                                       and v.is_contiguous()):
        raise FoleyRuntimeError("op_attention_scatter: head_dim 64, contiguous q / k / v of one (G, H), int32 out_rows [G, Sq], contiguous out [rows, H*64]")
    if (grp_q > 0) != (grp_kv > 0) or (grp_q > 0 and not half):
        raise FoleyRuntimeError("op_attention_scatter: grp_q and grp_kv come together and need 16-bit operands")
    _check(lib, lib.foley_op_attention_scatter(_ptr(q), _ptr(k), _ptr(v), dt_of(q), v.shape[3] if half else 0, G, H, Sq, Skv,
                                               grp_q, grp_kv, _ptr(out_rows), _ptr(out), out.shape[0], dt_of(out), _stream()), "foley_op_attention_scatter")


def op_qkv_regroup(qkv: torch.Tensor, heads: int, idx_q: torch.Tensor, idx_kv: torch.Tensor):
    """qkv [rows, 3*heads*64] (fp32 / bf16 / fp16) -> (q [G,H,Sq,64], k [G,H,Skv,64], v): v [G,H,Skv,64] for fp32 operands, v
    TRANSPOSED [G,H,64,ceil32(Skv)] (zeros beyond Skv) for 16-bit operands - what op_attention reads.  idx_q [G,Sq] / idx_kv [G,Skv]
    int32 source rows on the device."""
    lib = load_library()
    G, Sq = idx_q.shape
    Skv = idx_kv.shape[1]
    if qkv.dim() != 2 or qkv.shape[1] != 3 * heads * 64 or idx_kv.shape[0] != G or idx_q.dtype != torch.int32 or idx_kv.dtype != torch.int32:
        raise FoleyRuntimeError("op_qkv_regroup: qkv [rows, 3*H*64], int32 index tables with one row per group")
    if not (qkv.is_contiguous() and idx_q.is_contiguous() and idx_kv.is_contiguous()) or idx_q.device != qkv.device or idx_kv.device != qkv.device:
        raise FoleyRuntimeError("op_qkv_regroup: contiguous qkv (the kernel's row pitch is 3*H*64) and contiguous index tables on qkv's device")
    half = qkv.dtype in (torch.bfloat16, torch.float16)
    pitch = (Skv + 31) // 32 * 32 if half else 0
    q = torch.empty(G, heads, Sq, 64, device=qkv.device, dtype=qkv.dtype)
    k = torch.empty(G, heads, Skv, 64, device=qkv.device, dtype=qkv.dtype)
    v = torch.empty((G, heads, 64, pitch) if half else (G, heads, Skv, 64), device=qkv.device, dtype=qkv.dtype)
    _check(lib, lib.foley_op_qkv_regroup(_ptr(qkv), qkv.shape[0], dt_of(qkv), heads, _ptr(idx_q), G, Sq, _ptr(idx_kv), Skv, _ptr(q), _ptr(k), _ptr(v),
                                         pitch, _stream()), "foley_op_qkv_regroup")
    return q, k, v


def op_resize_aa_u8(x: torch.Tensor, axis: int, len_out: int, xmin: torch.Tensor, xsize: torch.Tensor, weights: torch.Tensor,
                     precision: int) -> torch.Tensor:
    """One pass of the antialiased uint8 resize along `axis` of a contiguous uint8 tensor (foley_op_resize_aa_u8): tables xmin /
    xsize int32 [len_out], weights int16 [len_out, kmax] on the device (host/encoders.py::aa_tables)."""
    lib = load_library()
    if x.dtype != torch.uint8 or not x.is_contiguous() or xmin.dtype != torch.int32 or xsize.dtype != torch.int32 or \
            weights.dtype != torch.int16 or weights.shape[0] != len_out or xmin.numel() != len_out or xsize.numel() != len_out:
        raise FoleyRuntimeError("op_resize_aa_u8: contiguous uint8 frames; int32 xmin / xsize [len_out], int16 weights [len_out, kmax]")
    axis %= x.dim()
    shape = list(x.shape)
    outer = 1
    for d in shape[:axis]:
        outer *= d
    inner = 1
    for d in shape[axis + 1:]:
        inner *= d
    len_in = shape[axis]
    shape[axis] = len_out
    out = torch.empty(shape, device=x.device, dtype=torch.uint8)
    if out.numel():
        _check(lib, lib.foley_op_resize_aa_u8(_ptr(x), outer, len_in, inner, len_out, _ptr(xmin), _ptr(xsize), _ptr(weights),
                                              weights.shape[1], precision, _ptr(out), _stream()), "foley_op_resize_aa_u8")
    return out


def op_ln_mod(x, eps, shift: Optional[RowBcastC], scale: Optional[RowBcastC], out):
    lib = load_library()
    M, D = x.shape
    _check(lib, lib.foley_op_ln_mod(_ptr(x), M, D, eps, C.byref(shift) if shift else None,
                                    C.byref(scale) if scale else None, _ptr(out), dt_of(out), _stream()),
           "foley_op_ln_mod")


EPI_QKV_SPLIT = 7


def qkv_split_desc(L, H, gains: Sequence, poss: Sequence, dsts: Sequence, S_tot, tok_off, eps, cos, sin,
                   vt_pitch: int = 0, attn=None) -> QkvSplitDescC:
    """Descriptor of the fused head-split epilogue (op_gemm(..., epilogue=EPI_QKV_SPLIT, qkv=desc)).
    attn = (k [sets,H,Skv,128], vt [sets,H,128,pitch], out [M,H*128], bdiv): ask for the attention against those cached
    keys in the same epilogue; desc.fused() tells after the launch whether the library took that form."""
    q = QkvSplitDescC()
    q.L, q.H, q.nK = L, H, len(dsts)
    for i in range(len(dsts)):
        q.gain[i] = _ptr(gains[i]) if gains[i] is not None else None
        q.pos[i] = _ptr(poss[i]) if poss[i] is not None else None
        q.dst[i] = _ptr(dsts[i])
    q.out_dtype, q.vt_pitch, q.S_tot, q.tok_off, q.eps = dt_of(dsts[0]), vt_pitch, S_tot, tok_off, eps
    q.cos_tab, q.sin_tab = _ptr(cos), _ptr(sin)
    q._keepalive = (list(gains), list(poss), list(dsts), cos, sin)
    if attn is not None:
        k, vt, out, bdiv = attn
        q.attn_k, q.attn_vt, q.attn_out = _ptr(k), _ptr(vt), _ptr(out)
        q.attn_skv, q.attn_pitch, q.attn_bdiv = k.shape[2], vt.shape[3], bdiv
        q._flag = C.c_int32(0)
        q.attn_fused = C.pointer(q._flag)
        q._keepalive += (k, vt, out)
    return q


def op_ln_mod_pending(x, eps, shift: Optional[RowBcastC], scale: Optional[RowBcastC], out, partials, k: int, bias,
                      gate: RowBcastC):
    """LayerNorm of x after x += gate * (sum(partials[:k]) + bias), x updated in place.  `partials`: fp32, or out's
    16-bit dtype (the slabs a GEMM with 16-bit `partials` left)."""
    lib = load_library()
    M, D = x.shape
    pdt = 0 if partials.dtype == torch.float32 else dt_of(partials)
    _check(lib, lib.foley_op_ln_mod_pending2(_ptr(x), M, D, eps, C.byref(shift) if shift else None,
                                             C.byref(scale) if scale else None, _ptr(out), dt_of(out), _ptr(partials), pdt,
                                             k, _ptr(bias) if bias is not None else None, C.byref(gate), _stream()),
           "foley_op_ln_mod_pending2")


def op_qkv_split(qkv, L, H, gains: Sequence, poss: Sequence, dsts: Sequence, S_tot, tok_off, eps, cos, sin,
                 vt_pitch: int = 0):
    lib = load_library()
    nK = len(dsts)
    M = qkv.shape[0]
    arr = lambda xs: (C.c_void_p * nK)(*[(_ptr(x) if x is not None else None) for x in xs])
    _check(lib, lib.foley_op_qkv_split(_ptr(qkv), M, L, H, nK, arr(gains), arr(poss), arr(dsts), dt_of(dsts[0]),
                                       vt_pitch, S_tot, tok_off, eps, _ptr(cos), _ptr(sin), _stream()),
           "foley_op_qkv_split")


def op_solver_step(pred, x, x_saved, d_acc, ncfg, guidance, coef, step_ptr, rows_out):
    lib = load_library()
    clips, Cc, L = x.shape
    _check(lib, lib.foley_op_solver_step(_ptr(pred), _ptr(x), _ptr(x_saved), _ptr(d_acc), clips, Cc, L, ncfg,
                                         float(guidance), _ptr(coef), _ptr(step_ptr), _ptr(rows_out),
                                         dt_of(rows_out), _stream()), "foley_op_solver_step")


def op_latent_rows(x, ncfg, out):
    lib = load_library()
    clips, Cc, L = x.shape
    _check(lib, lib.foley_op_latent_rows(_ptr(x), clips, Cc, L, ncfg, _ptr(out), dt_of(out), _stream()),
           "foley_op_latent_rows")


def op_dac_out(s, w, bias, out):
    lib = load_library()
    B, T, Cc = s.shape
    _check(lib, lib.foley_op_dac_out(_ptr(s), _ptr(w), _ptr(bias), B, T, Cc, _ptr(out), _stream()),
           "foley_op_dac_out")
