"""GEMM micro-benchmark on the C2 (bs=1 -> M=500) and bs=8 (M=4000) shapes of the xxl DiT.
Weights rotate through enough copies to defeat the 256 MB Infinity Cache (in the real loop every
weight matrix is streamed from HBM once per iteration).  HIP-event timed, median of `--reps`."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import runtime as rt  # noqa: E402

SHAPES = {  # name: (N, K) ; M given by --m
    "qkv": (4608, 1536), "proj": (1536, 1536), "mod6": (9216, 1536), "fc1": (6144, 1536), "fc2": (1536, 6144),
    "lin1": (1536, 4608), "w13": (8192, 4608), "w2": (1536, 12288), "smod": (36 * 9216, 1536),
}
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=500)
ap.add_argument("--tiles", default="1,2,3")
ap.add_argument("--shapes", default=",".join(SHAPES))
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--check", action="store_true")
ap.add_argument("--conv", action="store_true", help="treat K as 3*C of a channels-last conv k=3 over clips of 250 tokens")
ap.add_argument("--pf", default="0", help="comma list of L2 prefetch distances (K-slices beyond the ring) for the wave-specialised tiles")
ap.add_argument("--pad", default="0", help="comma list: row padding (elements) of BOTH operands' storage (lda = C + pad, ldw = K + pad)")
ap.add_argument("--ksplits", default="", help="comma list: benchmark the gated-residual epilogue with these K splits")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
tiles = [int(t) for t in a.tiles.split(",")]
import ctypes as _C
_lib = rt.load_library()
_lib.foley_debug_gemm_prefetch.argtypes = [_C.c_int]
_lib.foley_debug_gemm_prefetch.restype = None
pfs = [int(v) for v in a.pf.split(",")]
print(f"M={a.m} dtype={a.dtype}")
for name in a.shapes.split(","):
    N, K = SHAPES[name]
    ncopy = max(2, int(600e6 // (N * K * (2 if dt == torch.bfloat16 else 4))) + 1)
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(ncopy)]
    A = torch.randn(a.m, K // 3 if a.conv else K, device=dev).to(dt)
    ckw = dict(conv=(250, K // 3, 3, 1)) if a.conv else {}
    out = torch.empty(a.m, N, device=dev)
    ref = None
    if a.check:
        if a.conv:
            Cc = K // 3
            xr = A.float().view(-1, 250, Cc)
            xp = torch.nn.functional.pad(xr, (0, 0, 1, 1))
            im = torch.cat([xp[:, 0:250], xp[:, 1:251], xp[:, 2:252]], dim=2).reshape(a.m, K)
            ref = im @ Ws[0].float().t()
        else:
            ref = (A.float() @ Ws[0].float().t())
    line = f"{name:5s} N={N:5d} K={K:5d} |"
    if a.ksplits:
        gate = torch.randn(N, device=dev)
        x = torch.zeros(a.m, N, device=dev)
        for t in tiles:
            for ksp in [int(v) for v in a.ksplits.split(",")]:
                rt.op_gemm(A, Ws[0], None, out0=x, tile=t, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(gate, 0), ksplit=ksp, **ckw)
                torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
                for i, (e0, e1) in enumerate(evs):
                    e0.record()
                    rt.op_gemm(A, Ws[i % ncopy], None, out0=x, tile=t, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(gate, 0), ksplit=ksp, **ckw)
                    e1.record()
                torch.cuda.synchronize()
                ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
                us = ts[len(ts) // 2] * 1e3
                line += f" t{t}k{ksp}:{us:6.1f}us {2 * a.m * N * K / us / 1e6:4.0f}TF |"
        print(line, flush=True)
        continue
    for t, pf, pad in [(t, pf, pad) for t in tiles for pf in pfs for pad in [int(v) for v in a.pad.split(",")]]:
        _lib.foley_debug_gemm_prefetch(pf)
        if pad:      # row-padded copies of the same operands
            Ca = A.shape[1]
            Ap = torch.zeros(a.m, Ca + pad, device=dev, dtype=dt)
            Ap[:, :Ca] = A
            Wps = []
            for W in Ws:
                Wp = torch.zeros(N, K + pad, device=dev, dtype=dt)
                Wp[:, :K] = W
                Wps.append(Wp)
            run = lambda i: rt.op_gemm(Ap, Wps[i % ncopy], None, out0=out, tile=t, lda=Ca + pad, ldw=K + pad, NK=(N, K), M=a.m, **ckw)
        else:
            run = lambda i: rt.op_gemm(A, Ws[i % ncopy], None, out0=out, tile=t, **ckw)
        try:
            run(0)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            line += f" t{t}: ERR({str(e)[-40:]})"
            continue
        err = ""
        if ref is not None:
            err = f" e={float((out - ref).norm() / ref.norm()):.0e}"
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for i, (e0, e1) in enumerate(evs):
            e0.record()
            run(i)
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        us = ts[len(ts) // 2] * 1e3
        line += f" t{t}p{pf}d{pad}:{us:6.1f}us {2 * a.m * N * K / us / 1e6:5.0f}TF{err} |"
        if pad:
            del Wps, Ap
    print(line, flush=True)
