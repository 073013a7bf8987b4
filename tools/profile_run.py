"""Profiling driver: a few loop iterations + one DAC decode of the C2 workload (xxl, 5 s, CFG),
meant to be wrapped by `rocprofv3 --kernel-trace --stats -- python tools/profile_run.py`."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import config as C, packers, sampler, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--bs", type=int, default=1)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--model", default="xxl")
ap.add_argument("--graph", action="store_true")
ap.add_argument("--no-dac", action="store_true")
ap.add_argument("--encode", action="store_true", help="also time the DAC encoder on the decoded waveform (codec round trip)")
ap.add_argument("--duration", type=float, default=5.0)
ap.add_argument("--phases", action="store_true", help="sum workgroup-0 prologue / K-loop / epilogue time over all GEMM launches of the loop (eager)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = C.dit_config(a.model)
dtype = packers.torch_dtype(a.precision)
sd = synth.synth_dit_state_dict(cfg, device=dev)
cond = synth.synth_conditioning(cfg, a.duration, t2a=True, sd=sd, device=dev)
LA = int(a.duration * 50)
model = sampler.FoleyModel(cfg, sd, dtype, dev)
del sd
dac = sampler.FoleyDAC(synth.synth_dac_state_dict(C.DAC48K, device=dev, encoder=a.encode), dev)
model.attach_dac(dac)
plan = sampler.build_plan(model, {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]},
                          {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]}, LA, 4.5, a.iters,
                          a.bs, "euler")
model.ctx.prepare(plan)
lat = torch.randn(a.bs, 128, LA, device=dev)
if a.phases:
    import ctypes as CT
    from foley_amd.host import runtime as rt
    lib = rt.load_library()
    lib.foley_debug_gemm_timeline.argtypes = [CT.c_void_p, CT.c_int]
    lib.foley_debug_gemm_timeline.restype = None
    model.ctx.sample(lat.clone(), use_graph=False)
    torch.cuda.synchronize()
    dbg = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.foley_debug_gemm_timeline(CT.c_void_p(dbg.data_ptr()), 1)
model.ctx.sample(lat, use_graph=a.graph)
torch.cuda.synchronize()
if a.phases:
    lib.foley_debug_gemm_timeline(None, 0)
    d = dbg.cpu().tolist()
    n = max(d[3], 1)
    print(f"GEMM launches {d[3]} ({d[3] / a.iters:.0f}/iter): per iteration prologue {d[0] / 100 / a.iters:.0f} us, "
          f"K loop {d[1] / 100 / a.iters:.0f} us, epilogue {d[2] / 100 / a.iters:.0f} us (workgroup 0 of each launch)")
print("loop ms/iter:", model.ctx.last_elapsed_ms() / a.iters)
if not a.no_dac:
    wave = model.ctx.dac_decode(lat)
    torch.cuda.synchronize()
    print("dac ms:", model.ctx.last_elapsed_ms())
    if a.encode:
        for _ in range(2):
            params = model.ctx.dac_encode(wave)
            torch.cuda.synchronize()
        print("dac encode ms:", model.ctx.last_elapsed_ms(), tuple(params.shape))
