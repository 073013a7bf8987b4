"""Per-workgroup timeline of one GEMM launch (debug stamps written by the kernels when
foley_debug_gemm_timeline() is armed): entry, first K-slice landed, K loop done, epilogue done.
Times are s_memrealtime ticks (100 MHz -> 10 ns) relative to the earliest workgroup entry.

    python tools/gemm_timeline.py --shape w2 --tile 5 --ksplit 5 [--m 500] [--conv]
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

SHAPES = {"qkv": (4608, 1536), "proj": (1536, 1536), "fc1": (6144, 1536), "fc2": (1536, 6144),
          "lin1": (1536, 4608), "w13": (8192, 4608), "w2": (1536, 12288)}
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=500)
ap.add_argument("--shape", default="w2")
ap.add_argument("--tile", default="5", help="comma list")
ap.add_argument("--ksplit", default="0", help="comma list; 0 = plain fp32 store epilogue")
ap.add_argument("--partials", action="store_true", help="deferred split-K (partial slabs) instead of atomics")
ap.add_argument("--brief", action="store_true")
ap.add_argument("--fused-qkv", action="store_true", help="fused head-split epilogue (shape qkv: 3 x 12 heads, clips of 250 tokens)")
ap.add_argument("--ablate", type=int, default=0, help="debug bits: 1 no MFMA, 2 no fragment reads, 4 no global->LDS loads (glds kernels)")
ap.add_argument("--conv", action="store_true")
ap.add_argument("--act", default="", choices=["", "gelu", "silu", "silugate"], help="bf16-output epilogue (bias + activation) instead of the fp32 store")
ap.add_argument("--waits", action="store_true", help="wave-specialised tiles: where the first loader wave of every workgroup waits (memory vs consumers)")
ap.add_argument("--prologue", action="store_true", help="wave-specialised tiles: ring fill of the first loader wave - issue, first slice landed, first barrier")
ap.add_argument("--epilogue", action="store_true", help="vector epilogue (generic wave-specialised tiles): K loop done -> first barrier -> tile in LDS -> stores issued")
ap.add_argument("--warm", action="store_true", help="measure with the weight matrix just used (L2 / Infinity-Cache warm)")
a = ap.parse_args()
dev = torch.device("cuda:0")
N, K = SHAPES[a.shape]
lib = rt.load_library()
lib.foley_debug_gemm_timeline.argtypes = [C.c_void_p, C.c_int]
lib.foley_debug_gemm_timeline.restype = None
Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(max(2, int(600e6 // (N * K * 2)) + 1))]
A = torch.randn(a.m, K // 3 if a.conv else K, device=dev).bfloat16()
ckw = dict(conv=(250, K // 3, 3, 1)) if a.conv else {}
x = torch.zeros(a.m, N, device=dev)
gate = torch.randn(N, device=dev)


slabs = torch.empty(16, a.m, N, device=dev) if a.partials else None
bias_act = torch.randn(N, device=dev) * 0.1
out_act = torch.empty(a.m, N // 2 if a.act == "silugate" else N, device=dev, dtype=torch.bfloat16)


qkv_desc = None
if a.fused_qkv:
    from foley_amd.host import tables
    Hh, Lq = N // 384, 250
    clips_q = a.m // Lq
    dq, dk = (torch.zeros(clips_q, Hh, Lq, 128, device=dev, dtype=torch.bfloat16) for _ in range(2))
    dvt = torch.zeros(clips_q, Hh, 128, 256, device=dev, dtype=torch.bfloat16)
    cos_t, sin_t = tables.rope_table(Lq + 1)
    gq = torch.ones(128, device=dev)
    posq = torch.arange(Lq, dtype=torch.int32, device=dev)
    qkv_desc = rt.qkv_split_desc(Lq, Hh, [gq, gq, None], [posq, posq, None], [dq, dk, dvt], Lq, 0, 1e-6, cos_t.to(dev), sin_t.to(dev),
                                 vt_pitch=256)


def run(W, tile, ksplit):
    if qkv_desc is not None:
        rt.op_gemm(A, W, None, epilogue=rt.EPI_QKV_SPLIT, qkv=qkv_desc, tile=tile)
    elif ksplit:
        rt.op_gemm(A, W, None, out0=x, tile=tile, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(gate, 0), ksplit=ksplit,
                   partials=slabs, **ckw)
    elif a.act:
        epi = {"gelu": rt.EPI_GELU_T, "silu": rt.EPI_SILU_T, "silugate": rt.EPI_SILUGATE_T}[a.act]
        rt.op_gemm(A, W, bias_act, out0=out_act, tile=tile, epilogue=epi, **ckw)
    else:
        rt.op_gemm(A, W, None, out0=x, tile=tile, **ckw)


def q(v):
    v = v.sort().values
    n = len(v)
    return f"min {v[0]:6.2f}  p50 {v[n // 2]:6.2f}  p90 {v[int(n * 0.9)]:6.2f}  max {v[-1]:6.2f}"


for tile in [int(v) for v in a.tile.split(",")]:
    for ksplit in [int(v) for v in a.ksplit.split(",")]:
        for W in Ws[:3]:
            run(W, tile, ksplit)
        torch.cuda.synchronize()
        if a.waits:
            dbg = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
            lib.foley_debug_gemm_timeline(C.c_void_p(dbg.data_ptr()), 2)
            run(Ws[3 % len(Ws)], tile, ksplit)
            torch.cuda.synchronize()
            lib.foley_debug_gemm_timeline(None, 0)
            t = dbg.view(-1, 4).cpu().double()
            t = t[t[:, 3] > 0]
            per = t[:, :3] / (t[:, 3:4] - 1).clamp(min=1)
            print(f"{a.shape:5s} t{tile}k{ksplit}: wgs {len(t):4d} slices {int(t[0, 3])} | per slice (cycles): loader waits for memory p50 "
                  f"{float(per[:, 0].median()):6.0f} (max {float(per[:, 0].max()):6.0f}), at the barrier for the consumers p50 "
                  f"{float(per[:, 1].median()):6.0f} (max {float(per[:, 1].max()):6.0f}); loop p50 {float(per[:, 2].median()):6.0f} cycles/slice", flush=True)
            continue
        if a.epilogue:
            dbg = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
            lib.foley_debug_gemm_timeline(C.c_void_p(dbg.data_ptr()), 4)
            run(Ws[3 % len(Ws)], tile, ksplit)
            torch.cuda.synchronize()
            lib.foley_debug_gemm_timeline(None, 0)
            t = dbg.view(-1, 4).cpu().double()
            t = t[t[:, 0] > 0]
            print(f"{a.shape:5s} t{tile}k{ksplit}: wgs {len(t):4d} | epilogue of wave 0: wait for the other waves +{float((t[:, 1] - t[:, 0]).median()) / 100:5.2f}, "
                  f"tile -> LDS +{float((t[:, 2] - t[:, 1]).median()) / 100:5.2f}, LDS -> global stores issued +{float((t[:, 3] - t[:, 2]).median()) / 100:5.2f} us", flush=True)
            continue
        if a.prologue:
            dbg = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
            lib.foley_debug_gemm_timeline(C.c_void_p(dbg.data_ptr()), 3)
            run(Ws[2] if a.warm else Ws[3 % len(Ws)], tile, ksplit)
            torch.cuda.synchronize()
            lib.foley_debug_gemm_timeline(None, 0)
            t = dbg.view(-1, 4).cpu().double()
            t = t[t[:, 0] > 0]
            t0 = t[:, 0].min()
            print(f"{a.shape:5s} t{tile}k{ksplit}{' warm' if a.warm else ''}: wgs {len(t):4d} | loader entry p50 {float((t[:, 0] - t0).median()) / 100:5.2f} us after the first; "
                  f"ring issued +{float((t[:, 1] - t[:, 0]).median()) / 100:5.2f}, first slice landed +{float((t[:, 2] - t[:, 1]).median()) / 100:5.2f}, "
                  f"barrier +{float((t[:, 3] - t[:, 2]).median()) / 100:5.2f} us", flush=True)
            continue
        spans = []
        for rep in range(5):
            dbg = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
            lib.foley_debug_gemm_timeline(C.c_void_p(dbg.data_ptr()), a.ablate << 8)
            run(Ws[2] if a.warm else Ws[(3 + rep) % len(Ws)], tile, ksplit)
            torch.cuda.synchronize()
            lib.foley_debug_gemm_timeline(None, 0)
            t = dbg.view(-1, 4).cpu()
            cyc = t[8000, 1] - t[8000, 0]
            wall0 = (t[0, 3] - t[0, 0]) / 100.0
            t = t[:8000]
            t = t[t[:, 0] > 0].double()
            t = (t - t[:, 0].min()) / 100.0  # us
            spans.append((float(t[:, 3].max()), t, float(cyc) / max(float(wall0), 1e-3) / 1e3))
        spans.sort(key=lambda z: z[0])
        span, t, ghz = spans[len(spans) // 2]
        if a.brief:
            print(f"{a.shape:5s} t{tile}k{ksplit}{'p' if a.partials and ksplit else ''}: wgs {len(t):4d} span {span:6.1f} us | pro p50 {float((t[:, 1] - t[:, 0]).median()):5.2f} "
                  f"loop p50 {float((t[:, 2] - t[:, 1]).median()):6.2f} max {float((t[:, 2] - t[:, 1]).max()):6.2f} epi p50 {float((t[:, 3] - t[:, 2]).median()):5.2f} | {ghz:.2f} GHz", flush=True)
            continue
        print(f"{a.shape} M={a.m} N={N} K={K} tile={tile} ksplit={ksplit} conv={a.conv}: {len(t)} workgroups, span {span:.1f} us")
        print("entry            (us after first):", q(t[:, 0]))
        print("first slice landed - entry       :", q(t[:, 1] - t[:, 0]))
        print("K loop (after first slice)       :", q(t[:, 2] - t[:, 1]))
        print("epilogue                         :", q(t[:, 3] - t[:, 2]))
        print("exit             (us after first):", q(t[:, 3]))
