"""Time the 16-bit attention kernels at the shapes of the sampler (HIP events, operands rotated between launches).
    FOLEY_ATTN_LDS=0|1 python tools/attn_bench.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import runtime as rt  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [("bs8 triple self", 16, 12, 290, 290, 1), ("bs8 triple cross", 16, 12, 290, 77, 8), ("bs8 single self", 16, 12, 250, 250, 1),
          ("c5 triple self", 2, 12, 1740, 1740, 1), ("c5 single self", 2, 12, 1500, 1500, 1), ("c5 triple cross", 2, 12, 1740, 77, 1),
          ("bs1 single self", 2, 12, 250, 250, 1), ("bs1 triple self", 2, 12, 290, 290, 1)]
for name, B, H, Sq, Skv, bdiv in SHAPES:
    pitch = (Skv + 31) // 32 * 32
    sets = []
    for i in range(4):
        q = torch.randn(B, H, Sq, 128, device=dev).bfloat16()
        k = torch.randn(B // bdiv, H, Skv, 128, device=dev).bfloat16()
        v = torch.randn(B // bdiv, H, 128, pitch, device=dev).bfloat16()
        sets.append((q, k, v))
    out = torch.empty(B * Sq, H * 128, device=dev, dtype=torch.bfloat16)
    for q, k, v in sets:
        rt.op_attention(q, k, v, out, out, 0, bdiv)
    torch.cuda.synchronize()
    n = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        q, k, v = sets[i % 4]
        rt.op_attention(q, k, v, out, out, 0, bdiv)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    gf = 4.0 * B * H * Sq * Skv * 128 / 1e9
    tf = gf / us * 1e3
    print(f"{name:18s} B{B:3d} Sq{Sq:5d} Skv{Skv:5d}: {us:8.1f} us  {tf:7.1f} TFLOP/s ({tf / 2500:.3f} of the bf16 peak)")
