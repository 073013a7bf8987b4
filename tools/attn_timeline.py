"""Per-workgroup timeline of one bf16 attention launch (narrow kernel): entry, Q fragments loaded,
key loop done, merge + store done.  s_memrealtime ticks (10 ns), relative to the first entry."""
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
ap.add_argument("--sq", type=int, default=290)
ap.add_argument("--skv", type=int, default=290)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = rt.load_library()
lib.foley_debug_attn_timeline.argtypes = [C.c_void_p]
lib.foley_debug_attn_timeline.restype = None
pitch = (a.skv + 31) // 32 * 32
q = torch.randn(a.b, a.h, a.sq, 128, device=dev).bfloat16()
k = torch.randn(a.b, a.h, a.skv, 128, device=dev).bfloat16()
vt = torch.randn(a.b, a.h, 128, pitch, device=dev).bfloat16()
oa = torch.empty(a.b, 1, a.h * 128, device=dev, dtype=torch.bfloat16)
ob = torch.empty(a.b, a.sq, a.h * 128, device=dev, dtype=torch.bfloat16)
junk = torch.empty(64 << 20, device=dev)
for _ in range(3):
    rt.op_attention(q, k, vt, oa, ob, 0, 1)
res = []
for rep in range(5):
    junk.zero_()                       # push q/k/v out of the L2
    dbg = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
    lib.foley_debug_attn_timeline(C.c_void_p(dbg.data_ptr()))
    rt.op_attention(q, k, vt, oa, ob, 0, 1)
    torch.cuda.synchronize()
    lib.foley_debug_attn_timeline(None)
    t = dbg.view(-1, 4).cpu()
    t = t[t[:, 0] > 0].double()
    t = (t - t[:, 0].min()) / 100.0
    res.append((float(t[:, 3].max()), t))
res.sort(key=lambda z: z[0])
span, t = res[len(res) // 2]
med = lambda v: float(v.median())
print(f"B={a.b} H={a.h} Sq={a.sq} Skv={a.skv}: {len(t)} workgroups, span {span:.1f} us | entry p50 {med(t[:, 0]):.2f} max {float(t[:, 0].max()):.2f} | "
      f"Q load {med(t[:, 1] - t[:, 0]):.2f} | key loop {med(t[:, 2] - t[:, 1]):.2f} | merge+store {med(t[:, 3] - t[:, 2]):.2f}")
