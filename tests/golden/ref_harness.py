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


if __name__ == "__main__":
    ns = load_reference()
    print("reference imported:", ns.hifi.HunyuanVideoFoley, ns.dac.DAC,
          ns.sched.FlowMatchDiscreteScheduler, ns.utils.denoise_process_with_generator)
