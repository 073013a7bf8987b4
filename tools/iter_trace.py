"""Launch-by-launch timeline of foley_prepare and of one sampler iteration (eager launches, C2 shapes by default).
    rocprofv3 --kernel-trace -d /tmp/it -o it -- python tools/iter_trace.py [--bs 1] [--v2a]
    python tools/iter_trace.py --summarise /tmp/it [--what prepare|iter]
Markers: a latent_rows_kernel<float> launch separates the segments."""
import argparse
import os
import sqlite3
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=1)
ap.add_argument("--v2a", action="store_true")
ap.add_argument("--duration", type=float, default=5.0)
ap.add_argument("--summarise", default="")
ap.add_argument("--what", default="iter")
a = ap.parse_args()
if a.summarise:
    db = None
    for root, _d, files in os.walk(a.summarise):
        for f in files:
            if f.endswith(".db"):
                db = os.path.join(root, f)
    con = sqlite3.connect(db)
    rows = [r for r in con.execute("select name, start, end from kernels order by start") if "at::native" not in r[0]]
    marks = [i for i, r in enumerate(rows) if "latent_rows_kernel<float>" in r[0]]
    seg = {"prepare": (marks[-3], marks[-2]), "iter": (marks[-2], marks[-1])}[a.what]
    rows = rows[seg[0] + 1:seg[1]]
    if a.what == "iter":      # the last complete iteration: after the second-to-last step_increment
        incs = [i for i, r in enumerate(rows) if "step_increment" in r[0]]
        rows = rows[incs[-2] + 1:incs[-1] + 1]
    t0 = rows[0][1]
    tot, prev_end = 0.0, rows[0][1]
    for n, s, e in rows:
        n = n.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
        print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:5.1f}  +{(e - s) / 1e3:8.1f} us  {n[:100]}")
        tot += (e - s) / 1e3
        prev_end = e
    print(f"sum of kernels {tot:.1f} us, span {(rows[-1][2] - t0) / 1e3:.1f} us, {len(rows)} launches")
    sys.exit(0)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import config as C, runtime as rt, sampler, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = C.XXL
sd = synth.synth_dit_state_dict(cfg, device=dev)
cond = synth.synth_conditioning(cfg, a.duration, t2a=not a.v2a, sd=sd, device=dev)
model = sampler.FoleyModel(cfg, sd, torch.bfloat16, dev)
del sd
LA = int(a.duration * 50)
vis = {"siglip2_feat": cond["clip"], "syncformer_feat": cond["sync"]}
txt = {"text_feat": cond["text"], "uncond_text_feat": cond["uncond_text"]}
plan = sampler.build_plan(model, vis, txt, LA, 4.5, 4, a.bs, "euler")
lat = torch.randn(a.bs, 128, LA, device=dev)
mark = torch.empty(2 * a.bs * LA, 128, device=dev)
model.ctx.prepare(plan)
model.ctx.sample(lat.clone(), use_graph=False)
torch.cuda.synchronize()
rt.op_latent_rows(lat, 2, mark)
model.ctx.prepare(plan)
rt.op_latent_rows(lat, 2, mark)
model.ctx.sample(lat.clone(), use_graph=False)
rt.op_latent_rows(lat, 2, mark)
torch.cuda.synchronize()
print("ok")
