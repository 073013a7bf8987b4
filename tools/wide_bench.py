"""A/B of GEMM tiles on the large-grid shapes of the xxl DiT WITH their production epilogues (bf16 output, SiLU gate,
gated residual through deferred split-K slabs, GELU) - tools/gemm_bench.py times the fp32 store epilogue, which at
M = 4000 x N = 8192 moves more bytes than the operands.  Weights rotate through enough copies to defeat the Infinity
Cache; variants are interleaved round by round inside ONE process (median and min per variant).

    python tools/wide_bench.py --m 4000 --cases w13,w2,lin2,fc1,fc2,proj --tiles 0,23,31 [--rounds 12]
(position bias: the first tile of a round runs 3 - 10 % slower than the same kernel later in the round - compare columns, not boxes)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from foley_amd.host import runtime as rt  # noqa: E402

# name: (N, K, conv?, epilogue)
CASES = {"w13": (8192, 4608, True, "silugate"), "w2": (1536, 12288, True, "gate"), "lin2": (1536, 4608, True, "gate"),
         "fc1": (6144, 1536, False, "gelu"), "fc2": (1536, 6144, False, "gate"), "proj": (1536, 1536, False, "gate"),
         "smod": (36 * 9216 // 8, 1536, False, "f32")}
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=4000)
ap.add_argument("--cases", default="w13,w2,lin2,fc1,fc2")
ap.add_argument("--tiles", default="0,23,31")
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--fp8", action="store_true", help="weights stored as fp8 e4m3fn")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
tiles = [int(t) for t in a.tiles.split(",")]
print(f"M={a.m} dtype={a.dtype} fp8={a.fp8}")
for name in a.cases.split(","):
    N, K, conv, epi = CASES[name]
    esz = 1 if a.fp8 else 2
    ncopy = max(2, int(600e6 // (N * K * esz)) + 1)
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(dt) for _ in range(ncopy)]
    if a.fp8:
        Ws = [w.to(torch.float8_e4m3fn) for w in Ws]
    Cc = K // 3 if conv else K
    A = torch.randn(a.m, Cc, device=dev).to(dt)
    ckw = dict(conv=(250, Cc, 3, 1)) if conv else {}
    gate = torch.randn(N, device=dev)
    x = torch.zeros(a.m, N, device=dev)
    slabs = torch.empty(8, a.m, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev) * 0.1
    out16 = torch.empty(a.m, N // 2 if epi == "silugate" else N, device=dev, dtype=dt)
    out32 = torch.empty(a.m, N, device=dev) if epi == "f32" else None

    def run(tile, i):
        W = Ws[i % ncopy]
        tconv = tile
        if tile == 31 and not conv:
            tconv = 32                  # 31 is the tap-fused conv form of the 256x256 tile, 32 the plain one
        if tile == 32 and conv:
            tconv = 31
        if epi == "silugate":
            return rt.op_gemm(A, W, None, out0=out16, epilogue=rt.EPI_SILUGATE_T, tile=tconv, **ckw)
        if epi == "gelu":
            return rt.op_gemm(A, W, bias, out0=out16, epilogue=rt.EPI_GELU_T, tile=tconv, **ckw)
        if epi == "f32":
            return rt.op_gemm(A, W, bias, out0=out32, tile=tconv, **ckw)
        return rt.op_gemm(A, W, bias, out0=x, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(gate, 0), partials=slabs, tile=tconv, **ckw)

    times = {t: [] for t in tiles}
    used = {}
    ok = {}
    for t in tiles:
        try:
            used[t] = run(t, 0)
            torch.cuda.synchronize()
            ok[t] = True
        except Exception as e:  # noqa: BLE001
            ok[t] = False
            print(f"  {name} tile {t}: {str(e)[-80:]}")
    evs = []
    for r in range(a.rounds):      # queued back to back (no host sync in between: the events bracket the kernel, not the launch latency)
        for t in tiles:
            if not ok[t]:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(t, r + 1)
            e1.record()
            evs.append((t, e0, e1))
    torch.cuda.synchronize()
    for t, e0, e1 in evs:
        times[t].append(e0.elapsed_time(e1) * 1e3)
    line = f"{name:5s} N={N:5d} K={K:5d} |"
    for t in tiles:
        if not ok[t]:
            continue
        ts = sorted(times[t])
        med, mn = ts[len(ts) // 2], ts[0]
        line += f" t{t}(k{used[t]}): {med:7.1f} us (min {mn:6.1f}) {2 * a.m * N * K / med / 1e6:5.0f} TF |"
    print(line, flush=True)
