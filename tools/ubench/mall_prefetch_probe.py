import os, sys, ctypes as C
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from foley_amd.host import runtime as rt
dev = torch.device("cuda:0")
lib = rt.load_library()
lib.foley_debug_gemm_timeline.argtypes = [C.c_void_p, C.c_int]
lib.foley_debug_gemm_timeline.restype = None
def mk(N, K, n): return [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(n)]
for name, (N, K) in {"qkv": (4608, 1536), "w13": (8192, 4608), "fc2": (1536, 6144)}.items():
    Ws = mk(N, K, max(3, int(700e6 // (N * K * 2)) + 1))      # rotate > 256 MB so "cold" is really HBM
    Wb = mk(8192, 4608, 2)                                     # the "current" GEMM between prefetch and use
    A = torch.randn(500, K, device=dev).bfloat16(); Ab = torch.randn(500, 4608, device=dev).bfloat16()
    out = torch.empty(500, N, device=dev); outb = torch.empty(500, 8192, device=dev)
    res = {}
    for mode in ("cold", "prefetched"):
        spans = []
        for rep in range(6):
            W = Ws[rep % len(Ws)]
            for Wo in Ws:                                    # evict: touch every other copy
                if Wo is not W: Wo.view(torch.int32)[::16].sum()
            if mode == "prefetched":
                W.view(torch.int32).sum()                      # what a prefetch kernel would do
            rt.op_gemm(Ab, Wb[rep % 2], None, out0=outb, tile=15)   # the GEMM that runs meanwhile (thrashes L2)
            dbg = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
            lib.foley_debug_gemm_timeline(C.c_void_p(dbg.data_ptr()), 0)
            rt.op_gemm(A, W, None, out0=out, tile=15)
            torch.cuda.synchronize()
            lib.foley_debug_gemm_timeline(None, 0)
            t = dbg.view(-1, 4)[:8000].cpu(); t = t[t[:, 0] > 0].double()
            spans.append((float((t[:, 3].max() - t[:, 0].min()) / 100), float((t[:, 2] - t[:, 1]).median() / 100)))
        spans.sort()
        res[mode] = spans[len(spans) // 2]
    print(f"{name}: cold span {res['cold'][0]:.1f} us loop {res['cold'][1]:.1f} | prefetched (then 75 MB of other traffic) span {res['prefetched'][0]:.1f} us loop {res['prefetched'][1]:.1f}")
