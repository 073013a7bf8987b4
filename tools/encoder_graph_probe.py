"""Probe (round 5): are the HIP-engine encoders of the V2A case (SigLIP2 tower on 40 frames, Synchformer on 14 segments) bound by the
GPU or by the ~150 - 250 Python-issued launches each?  Times each encoder eagerly (host wall clock around a synchronised call, and the
GPU span between two events on the launch stream), then captures the same call into a hipGraph (torch.cuda.graph: every op of
host/encoders_hip.py launches on torch's current stream) and times the replay.  Same synthetic weights / shapes as bench.py's
encoder_pass."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import encoders as E  # noqa: E402
from foley_amd.host import encoders_hip as EH  # noqa: E402
from foley_amd.host import synth  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
from transformers import SiglipVisionConfig, SiglipVisionModel  # noqa: E402

torch.manual_seed(0)
sig = SiglipVisionModel(SiglipVisionConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                           image_size=512, patch_size=16)).eval().to(dev, dtype)
sig_sd = E._siglip_state(sig, dev)
sync_sd = {k: v.to(dev, dtype) for k, v in synth.materialize(E.synchformer_schema()).items()}
g = torch.Generator().manual_seed(3)
p8 = torch.randn(40, 3, 512, 512, generator=g).to(dev)
p25 = torch.randn(125, 3, 224, 224, generator=g).to(dev)

CASES = {
    "siglip2 (40 frames)": lambda: EH.siglip_image_features_hip(sig_sd, p8, dtype),
    "synchformer (125 frames)": lambda: EH.encode_video_with_sync_hip(sync_sd, p25, torch.float16),
}


def timed(fn, n=5):
    wall, span = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        wall.append(1e3 * (time.perf_counter() - t0))
        span.append(e0.elapsed_time(e1))
    return out, sorted(wall)[n // 2], sorted(span)[n // 2]


ONLY = os.environ.get("PROBE_ONLY", "")
for name, fn in CASES.items():
    if ONLY and not name.startswith(ONLY):
        continue
    fn()
    fn()
    ref, wall, span = timed(fn)
    print(f"{name:28s} eager : wall {wall:7.2f} ms, GPU span {span:7.2f} ms", flush=True)
    if os.environ.get("PROBE_NO_GRAPH"):
        continue
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fn()
        rep = lambda: (graph.replay(), out)[1]  # noqa: E731
        got, wall, span = timed(rep)
        err = float((got.float() - ref.float()).norm() / ref.float().norm())
        print(f"{name:28s} graph : wall {wall:7.2f} ms, GPU span {span:7.2f} ms, rel-L2 vs eager {err:.2e}", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"{name:28s} graph : capture failed: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
