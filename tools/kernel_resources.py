"""Register / scratch usage of every HIP kernel instantiation (clang -Rpass-analysis=kernel-resource-usage,
device-only compile, no GPU needed).  A kernel whose accumulators end up in scratch memory runs its K loop
through memory - round 2 found the 256x128 four-consumer GEMM with the scalar epilogue in that state
(576 bytes per lane) this way.
    python tools/kernel_resources.py [--all]        # default: only kernels that use scratch
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument("--all", action="store_true")
a = ap.parse_args()
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-hunyuanvideo-foley_amd", "csrc")
files = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
K1, K2 = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]"
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    procs = [(f, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
                                   os.path.join(csrc, f), "-o", os.path.join(tmp, f + ".o"), "-Rpass-analysis=kernel-resource-usage"],
                                  stderr=subprocess.PIPE, text=True)) for f in files]
    for f, p in procs:
        txt = p.communicate()[1]
        blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
        for b in blocks:
            name = b.split("\n")[0].strip()
            g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
            if a.all or g(K1) not in ("0", "?"):
                print(f"{f:16s} {name[:100]:100s} VGPR {g('VGPRs'):>4s} spill {g('VGPRs Spill'):>3s} scratch {g(K1):>4s} occ {g(K2)}")
            bad += g(K1) not in ("0", "?")
        print(f"{f}: {len(blocks)} kernels", file=sys.stderr)
print(f"{bad} kernel(s) use scratch memory")
