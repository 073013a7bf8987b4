"""Import harness for the upstream reference (ONLY usable in the build container).

This file is test infrastructure.  It makes `/root/reference` importable on a
box that has neither ComfyUI, diffusers, loguru nor audiotools by registering
minimal stand-in modules in `sys.modules` *for the third-party packages the
reference imports* (SURVEY.md §8c lists them).  Nothing from the reference is
copied: the reference's own arithmetic (hifi_foley.py, scheduling_flow_match_
discrete.py, dac.py, /utils.py) is executed as-is to

  1. validate `oracle/foley_oracle.py` (the CPU restatement) and
  2. generate the golden vectors committed under `tests/golden/*.npz`
     (see `make_golden.py`).

`/root/reference` does not exist on the GPU box; nothing under `tests/` that
runs there may import this module (tests that do are marked `needs_reference`
and skip when the directory is absent).
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FOLEY_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "hunyuanvideo_foley"))


def _mod(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.setdefault("__path__", [])  # behave like a package
    sys.modules[name] = m
    return m


def _install_stubs() -> None:
    import torch

    if "loguru" not in sys.modules:
        lg = _mod("loguru")

        class _Logger:
            def __getattr__(self, _name):
                return lambda *a, **k: None

        lg.logger = _Logger()

    if "comfy" not in sys.modules:
        comfy = _mod("comfy")
        mm = _mod("comfy.model_management")
        cu = _mod("comfy.utils")
        mm.get_torch_device = lambda: torch.device("cpu")
        mm.unet_offload_device = lambda: torch.device("cpu")
        mm.soft_empty_cache = lambda *a, **k: None

        class ProgressBar:
            def __init__(self, total):
                self.total = total
                self.n = 0

            def update(self, k):
                self.n += k

        def load_torch_file(path, device=None):
            if str(path).endswith(".safetensors"):
                from safetensors.torch import load_file

                return load_file(path)
            return torch.load(path, map_location="cpu")

        cu.ProgressBar = ProgressBar
        cu.load_torch_file = load_torch_file
        comfy.model_management = mm
        comfy.utils = cu

    if "diffusers" not in sys.modules:
        d = _mod("diffusers")
        dm = _mod("diffusers.models")
        dc = _mod("diffusers.configuration_utils")
        du = _mod("diffusers.utils")
        dut = _mod("diffusers.utils.torch_utils")
        ds = _mod("diffusers.schedulers")
        dss = _mod("diffusers.schedulers.scheduling_utils")

        class ModelMixin(torch.nn.Module):
            @property
            def dtype(self):
                return next(self.parameters()).dtype

            @property
            def device(self):
                return next(self.parameters()).device

        class _Cfg(dict):
            __getattr__ = dict.__getitem__

        class ConfigMixin:
            pass

        def register_to_config(init):
            import functools
            import inspect

            sig = inspect.signature(init)

            @functools.wraps(init)
            def wrapped(self, *args, **kwargs):
                bound = sig.bind(self, *args, **kwargs)
                bound.apply_defaults()
                cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
                object.__setattr__(self, "_foley_cfg", _Cfg(cfg))
                init(self, *args, **kwargs)

            return wrapped

        ConfigMixin.config = property(lambda self: self._foley_cfg)

        class BaseOutput:
            """Base of the scheduler's @dataclass output; supports `out[0]` (reference utils.py:246)."""

            def __getitem__(self, k):
                import dataclasses

                vals = [getattr(self, f.name) for f in dataclasses.fields(self)]
                return vals[k] if isinstance(k, int) else getattr(self, k)

        class _Log:
            @staticmethod
            def get_logger(_n):
                import logging

                return logging.getLogger("diffusers-stub")

        def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
            t = torch.randn(tuple(shape), generator=generator, device="cpu", dtype=dtype)
            return t.to(device) if device is not None else t

        class SchedulerMixin:
            pass

        dm.ModelMixin = ModelMixin
        dc.ConfigMixin = ConfigMixin
        dc.register_to_config = register_to_config
        du.BaseOutput = BaseOutput
        du.logging = _Log
        dut.randn_tensor = randn_tensor
        dss.SchedulerMixin = SchedulerMixin
        ds.SchedulerMixin = SchedulerMixin
        ds.DDPMScheduler = type("DDPMScheduler", (), {})
        ds.EulerDiscreteScheduler = type("EulerDiscreteScheduler", (), {})
        d.models, d.configuration_utils, d.utils, d.schedulers = dm, dc, du, ds
        du.torch_utils = dut
        ds.scheduling_utils = dss

    if "audiotools" not in sys.modules:
        at = _mod("audiotools")
        atml = _mod("audiotools.ml")
        at.AudioSignal = type("AudioSignal", (), {})
        at.STFTParams = type("STFTParams", (), {})
        atml.BaseModel = type("BaseModel", (torch.nn.Module,), {"INTERN": [], "EXTERN": []})
        atml.Accelerator = type("Accelerator", (), {})
        at.ml = atml
    if "argbind" not in sys.modules:
        ab = _mod("argbind")
        ab.bind = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
        ab.parse_args = lambda *a, **k: {}

        class _Scope:
            def __init__(self, *a, **k):
                pass

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

        ab.scope = _Scope


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference's hot-path classes/functions."""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.hifi = importlib.import_module("hunyuanvideo_foley.models.hifi_foley")
    ns.sched = importlib.import_module(
        "hunyuanvideo_foley.utils.schedulers.scheduling_flow_match_discrete")
    ns.dac = importlib.import_module("hunyuanvideo_foley.models.dac_vae.model.dac")
    ns.cfgu = importlib.import_module("hunyuanvideo_foley.utils.config_utils")
    ns.attn = importlib.import_module("hunyuanvideo_foley.models.nn.attn_layers")
    ns.posemb = importlib.import_module("hunyuanvideo_foley.models.nn.posemb_layers")
    # the Comfy-side /utils.py (denoise_process_with_generator) is loaded by path
    spec = importlib.util.spec_from_file_location(
        "foley_ref_utils", os.path.join(REFERENCE_ROOT, "utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ns.utils = m
    _CACHE["ns"] = ns
    return ns


def _install_synchformer_stubs() -> None:
    """Third-party modules the reference's MotionFormer imports and this image lacks: omegaconf (a YAML
    -> attribute-dict loader is all it uses) and timm's two helpers.  The reference's own files
    (motionformer.py, video_model_builder.py, vit_helper.py, the YAML config) are executed as they are."""
    import itertools

    import torch
    import yaml

    if "omegaconf" not in sys.modules:
        oc = _mod("omegaconf")

        class _Node(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)

            def __setattr__(self, k, v):
                self[k] = v

        def _wrap(o):
            if isinstance(o, dict):
                return _Node({k: _wrap(v) for k, v in o.items()})
            if isinstance(o, list):
                return [_wrap(v) for v in o]
            return o

        class OmegaConf:
            @staticmethod
            def load(path):
                with open(path, "r", encoding="utf-8") as f:
                    return _wrap(yaml.safe_load(f))

        oc.OmegaConf = OmegaConf
    if "timm" not in sys.modules:
        timm = _mod("timm")
        tl = _mod("timm.layers")
        tml = _mod("timm.models")
        tmll = _mod("timm.models.layers")

        def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
            return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else tuple(itertools.repeat(x, 2))

        for m in (tl, tmll):
            m.trunc_normal_ = trunc_normal_
            m.to_2tuple = to_2tuple
        timm.layers, timm.models, tml.layers = tl, tml, tmll
    if "requests" not in sys.modules:      # synchformer/utils.py imports it for its (unused) downloader
        _mod("requests")


def load_motionformer():
    """The reference's Synchformer visual extractor, constructed exactly as Synchformer.__init__ does
    (synchformer.py:22-28).  The package's __init__ pulls in torchaudio / the audio branch, so the
    sub-package is registered by path and only motionformer.py (+ what IT imports) is executed."""
    if "mf" in _CACHE:
        return _CACHE["mf"]
    load_reference()
    _install_synchformer_stubs()
    pkg_name = "hunyuanvideo_foley.models.synchformer"
    pkg_dir = os.path.join(REFERENCE_ROOT, "hunyuanvideo_foley", "models", "synchformer")
    if pkg_name not in sys.modules:
        pkg = types.ModuleType(pkg_name)
        pkg.__path__ = [pkg_dir]
        sys.modules[pkg_name] = pkg
    mf = importlib.import_module(pkg_name + ".motionformer")
    model = mf.MotionFormer(extract_features=True, factorize_space_time=True, agg_space_module="TransformerEncoderLayer",
                            agg_time_module="torch.nn.Identity", add_global_repr=False)
    _CACHE["mf"] = model.eval()
    return _CACHE["mf"]


if __name__ == "__main__":
    ns = load_reference()
    print("reference imported:", ns.hifi.HunyuanVideoFoley, ns.dac.DAC,
          ns.sched.FlowMatchDiscreteScheduler, ns.utils.denoise_process_with_generator)
