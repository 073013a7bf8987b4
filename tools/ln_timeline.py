"""Phase timeline of the multi-wave LayerNorm + modulate kernel (debug stamps): entry -> loads landed -> statistics
(wave reductions + one barrier) -> stores issued.   python tools/ln_timeline.py [--m 500] [--k 5]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import runtime as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=500)
ap.add_argument("--k", type=int, default=5, help="pending split-K slabs (0: plain LayerNorm)")
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = rt.load_library()
lib.foley_debug_ln_timeline.argtypes = [C.c_void_p]
lib.foley_debug_ln_timeline.restype = None
D, La, Ls = 1536, 250, 112
x = torch.randn(a.m, D, device=dev)
tab = torch.randn(2, 8, 6 * D, device=dev) * 0.1
slabs = torch.randn(max(a.k, 1), a.m, D, device=dev) * 0.1
bias = torch.randn(D, device=dev)
out = torch.empty(a.m, D, device=dev, dtype=torch.bfloat16)
rb = lambda c: rt.rowbcast(tab[..., c * D:], 2, a.m // 2, La, ld=6 * D, Ls=Ls, period=8)
junk = torch.empty(64 << 20, device=dev)


def run():
    if a.k:
        rt.op_ln_mod_pending(x, 1e-6, rb(0), rb(1), out, slabs, a.k, bias, rb(2))
    else:
        rt.op_ln_mod(x, 1e-6, rb(0), rb(1), out)


for _ in range(3):
    run()
torch.cuda.synchronize()
for rep in range(3):
    junk.fill_(rep)
    dbg = torch.zeros(4096 * 4, dtype=torch.int64, device=dev)
    lib.foley_debug_ln_timeline(C.c_void_p(dbg.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.foley_debug_ln_timeline(None)
    t = dbg.view(-1, 4).cpu().double()
    t = t[t[:, 0] > 0] / 100.0
    t0 = t[:, 0].min()
    d = lambda i: float((t[:, i] - t[:, i - 1]).median())
    print(f"wgs {len(t):4d} | entry skew {float((t[:, 0] - t0).median()):5.2f} (max {float((t[:, 0] - t0).max()):5.2f}) | loads landed +{d(1):5.2f} | "
          f"statistics +{d(2):5.2f} | stores issued +{d(3):5.2f} | span {float(t[:, 3].max() - t0):5.2f} us")
