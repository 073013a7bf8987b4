"""Per-launch timeline of one DAC decode (5 s clip): wrap with rocprofv3 --kernel-trace and list the launches in order.
    rocprofv3 --kernel-trace -d /tmp/dt -o dt -- python tools/dac_trace.py [--bs 1]; python tools/dac_trace.py --summarise /tmp/dt
"""
import argparse
import os
import sqlite3
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=1)
ap.add_argument("--summarise", default="")
a = ap.parse_args()
if a.summarise:
    db = None
    for root, _d, files in os.walk(a.summarise):
        for f in files:
            if f.endswith(".db"):
                db = os.path.join(root, f)
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, start, end from kernels order by start"))
    rows = [r for r in rows if "at::native" not in r[0]]
    # the last decode: launches after the last latent_rows_kernel<float>
    last = max(i for i, r in enumerate(rows) if "latent_rows_kernel<float>" in r[0])
    rows = rows[last:]
    t0 = rows[0][1]
    tot = 0.0
    for n, s, e in rows:
        n = n.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {n[:90]}")
        tot += (e - s) / 1e3
    print(f"sum of kernels {tot:.1f} us, span {(rows[-1][2] - t0) / 1e3:.1f} us, {len(rows)} launches")
    sys.exit(0)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import config as C, sampler, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = C.TINY
model = sampler.FoleyModel(cfg, synth.synth_dit_state_dict(cfg), torch.float32, dev)
dac = sampler.FoleyDAC(synth.synth_dac_state_dict(C.DAC48K, device=dev), dev)
model.attach_dac(dac)
lat = torch.randn(a.bs, 128, 250, device=dev)
for _ in range(3):
    w = model.ctx.dac_decode(lat)
torch.cuda.synchronize()
print("dac ms", model.ctx.last_elapsed_ms(), tuple(w.shape))
