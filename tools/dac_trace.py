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
    # what each launch of foley_dac_decode does (csrc/foley_rt.hip: latent rows, post-quant conv, input conv k=7, then per stage
    # the transposed conv and three residual units of conv7(dilated) + conv1, then the output conv): algorithmic FLOPs and the
    # bytes that MUST move (operands + results once: fp32 activations time-major [T, C], weights once), against the fp32
    # matrix peak (157.3 TFLOP/s) and the 8 TB/s HBM roof
    L, dim, rates, dil, clips, T = 128, 2048, (8, 5, 4, 3, 2), (1, 3, 9), a.bs, 250
    plan = [("latent rows", 0, 2 * clips * T * L * 4), ("post-quant conv1", 2.0 * clips * T * L * L, clips * T * L * 8 + L * L * 4),
            ("in conv7 128->2048 + snake", 2.0 * clips * T * dim * 7 * L, clips * T * (L + dim) * 4 + 7 * L * dim * 4)]
    Tin, Cin = T, dim
    for i, sr in enumerate(rates):
        Cout, Tout = Cin // 2, Tin * sr
        plan.append((f"stage {i} convT x{sr} {Cin}->{Cout} (T {Tin}->{Tout})", 2.0 * clips * (Tin + 1) * sr * Cout * 2 * Cin,
                     clips * (Tin * Cin + 2 * Tout * Cout) * 4 + 2 * Cin * sr * Cout * 4))
        for d in dil:
            plan.append((f"stage {i} conv7 dil {d} C={Cout}", 2.0 * clips * Tout * Cout * 7 * Cout, clips * Tout * Cout * 8 + 7 * Cout * Cout * 4))
            plan.append((f"stage {i} conv1 + residual + snake C={Cout}", 2.0 * clips * Tout * Cout * Cout, clips * Tout * Cout * 16 + Cout * Cout * 4))
        Tin, Cin = Tout, Cout
    plan.append((f"out conv7 {Cin}->1 + tanh", 2.0 * clips * Tin * 7 * Cin, clips * Tin * (Cin + 1) * 4))
    assert len(plan) == len(rows), (len(plan), len(rows))
    print("| launch | us | GFLOP | MB | TFLOP/s (of 157.3) | TB/s (of 8) | bound, us |")
    print("|---|---|---|---|---|---|---|")
    tb = 0.0
    for (what, fl, by), (n, s, e) in zip(plan, rows):
        us = (e - s) / 1e3
        b_us = max(fl / 157.3e12, by / 8e12) * 1e6
        tb += b_us
        print(f"| {what} | {us:.1f} | {fl / 1e9:.2f} | {by / 1e6:.1f} | {fl / us / 1e6:.1f} ({fl / us / 1e6 / 157.3:.2f}) | {by / us / 1e6:.2f} ({by / us / 1e6 / 8:.2f}) | {b_us:.1f} ({'MFMA' if fl / 157.3e12 > by / 8e12 else 'HBM'}) |")
        tot += us
    print(f"\nsum of kernels {tot:.1f} us, span {(rows[-1][2] - t0) / 1e3:.1f} us, {len(rows)} launches; sum of the per-launch roofline bounds {tb:.1f} us "
          f"({tb / tot:.2f} of the measured sum)")
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
