"""Phase timeline of the small-grid bf16 attention kernel (debug stamps, s_memrealtime): entry -> all operands landed ->
key tiles done -> partial results merged through LDS -> output stored.
    python tools/attn_timeline.py [--b 2] [--h 12] [--sq 250] [--skv 250]
"""
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
ap.add_argument("--b", type=int, default=2)
ap.add_argument("--h", type=int, default=12)
ap.add_argument("--sq", type=int, default=250)
ap.add_argument("--skv", type=int, default=250)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = rt.load_library()
lib.foley_debug_attn_timeline.argtypes = [C.c_void_p]
lib.foley_debug_attn_timeline.restype = None
pitch = (a.skv + 31) // 32 * 32
sets = []
for i in range(6):   # rotate operand sets: in the loop they were written by the previous kernel, on other XCDs
    q = torch.randn(a.b, a.h, a.sq, 128, device=dev).bfloat16()
    k = torch.randn(a.b, a.h, a.skv, 128, device=dev).bfloat16()
    v = torch.randn(a.b, a.h, 128, pitch, device=dev).bfloat16()
    sets.append((q, k, v))
out = torch.empty(a.b * a.sq, a.h * 128, device=dev, dtype=torch.bfloat16)
junk = torch.empty(64 << 20, device=dev)
for q, k, v in sets[:3]:
    rt.op_attention(q, k, v, out, out, 0)
torch.cuda.synchronize()
rows = []
if (a.sq + 127) // 128 * a.h * a.b >= 256:    # the launcher takes the large-grid kernel: cycle accounting instead of stamps
    dbg = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    lib.foley_debug_attn_timeline(C.c_void_p(dbg.data_ptr()))
    q, k, v = sets[3]
    rt.op_attention(q, k, v, out, out, 0)
    torch.cuda.synchronize()
    lib.foley_debug_attn_timeline(None)
    t = dbg.view(-1, 8).cpu().double()
    t = t[t[:, 3] > 0]
    print(f"wide kernel: wgs {len(t)} key tiles {int(t[0, 3])} | per tile (cycles, wave 0): math {float((t[:, 1] / t[:, 3]).median()):6.0f}, "
          f"staging + barrier {float((t[:, 2] / t[:, 3]).median()):6.0f}; whole loop {float(t[:, 0].median()):8.0f} cycles")
    sys.exit(0)
for rep in range(3):
    junk.fill_(rep)                      # push the operands out of the L2s
    dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    lib.foley_debug_attn_timeline(C.c_void_p(dbg.data_ptr()))
    q, k, v = sets[3 + rep]
    rt.op_attention(q, k, v, out, out, 0)
    torch.cuda.synchronize()
    lib.foley_debug_attn_timeline(None)
    t = dbg.view(-1, 8).cpu().double()
    t = t[t[:, 0] > 0] / 100.0
    t0 = t[:, 0].min()
    d = lambda i: float((t[:, i] - t[:, i - 1]).median())
    print(f"wgs {len(t):4d} | entry skew {float((t[:, 0] - t0).median()):5.2f} | operands landed +{d(1):5.2f} | key tiles +{d(2):5.2f} | "
          f"LDS merge barrier +{d(3):5.2f} | merge + store +{d(4):5.2f} | span {float(t[:, 4].max() - t0):5.2f} us")
