"""Probe (round 5): how much of a bs=1 GEMM launch is the COLD first slices of its weight stream?  The same GEMM (M = 500, the
sampler's shapes, weights rotated through > 256 MB so that they arrive from HBM) timed (a) cold, (b) after a tiny kernel touched the
first `--slices` K-slices of every weight row (their lines are then in the Infinity Cache / some L2), (c) after the whole matrix was
touched.  (b) - (a) bounds what a prefix prefetch by the preceding LayerNorm / attention launch could buy."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import runtime as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slices", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
SHAPES = {"qkv": (4608, 1536), "fc1": (6144, 1536), "w13(plain K)": (8192, 4608)}
for name, (N, K) in SHAPES.items():
    ncopy = int(600e6 // (N * K * 2)) + 2
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(ncopy)]
    A = torch.randn(500, K, device=dev).bfloat16()
    out = torch.empty(500, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for mode in ("cold", "prefix", "all"):
        ts = []
        for i in range(3 * ncopy):
            W = Ws[i % ncopy]
            if mode == "prefix":
                W[:, : 64 * a.slices].float().sum()      # touches the first slices of every row
            elif mode == "all":
                W.float().sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rt.op_gemm(A, W, None, out0=out, epilogue=rt.EPI_STORE_T)
            e1.record()
            torch.cuda.synchronize()
            if i >= ncopy:
                ts.append(1e3 * e0.elapsed_time(e1))
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print(f"{name:14s} N={N} K={K}: cold {res['cold']:.1f} us | first {a.slices} slices touched {res['prefix']:.1f} us | all touched {res['all']:.1f} us")
