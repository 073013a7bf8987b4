"""Aggregate rocprofv3 --pmc results (rocpd sqlite) per kernel: sum of each counter + launches.

    pmc_summary.py [--traffic-json OUT --iters N --workload TAG] db [db ...]

--traffic-json writes the HBM traffic per loop iteration that bench.py reports as `roofline.traffic`:
(2 x FETCH_SIZE + WRITE_SIZE) KiB over the kernels of the sampler loop / N traced iterations.  The
factor 2 is the gfx950 correction of MI355X_MICROARCH.md ("FETCH_SIZE reports exactly 1/2 of the bytes
of a wide coalesced streaming read"); the file records the kernel-source hash it was measured on and
bench.py refuses to reuse it for other sources."""
import argparse
import json
import os
import sqlite3
import sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dbs", nargs="+")
ap.add_argument("--traffic-json")
ap.add_argument("--mfma-json", help="SQ pass: write the matrix-pipe utilisation of the kernel with the most MFMA cycles (bench.py: roofline.mfma_busy)")
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--workload", default="c2/bs1/bf16/xxl")
a = ap.parse_args()

agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for path in a.dbs:
    db = sqlite3.connect(path)
    seen = set()
    for name, disp, cname, val in db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        k = name.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
        k = k.split("(")[0].replace("void ", "")
        agg[k][cname] += val
        if (path, disp) not in seen and cname in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
            seen.add((path, disp))
            cnt[(k, cname)] += 1
counters = sorted({c for v in agg.values() for c in v})
# derived: matrix-pipe utilisation = MFMA-busy cycles per SIMD / cycles the kernel was resident.  SQ_VALU_MFMA_BUSY_CYCLES sums
# the 1024 SIMDs, SQ_BUSY_CYCLES the 32 shader engines (8 XCDs x 4): busy / (32 x SQ_BUSY) - checked against the in-kernel
# timeline (tools/gemm_timeline.py: w1/w3 at M = 500 keeps the pipe busy 22 of 38 us; the counters say 0.53 - 0.57)
util = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "SQ_BUSY_CYCLES" in counters
print("| kernel | launches | " + " | ".join(counters) + (" | MFMA busy |" if util else " |"))
print("|---|---|" + "---|" * (len(counters) + (1 if util else 0)))
tot = defaultdict(float)
u = lambda v: (f" {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (32.0 * v['SQ_BUSY_CYCLES']):.3f} |" if v.get("SQ_BUSY_CYCLES") else " |") if util else ""
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    n = max([cnt[(k, c)] for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES")] + [0])
    print(f"| `{k[:70]}` | {n} | " + " | ".join(f"{agg[k].get(c, 0):.4g}" for c in counters) + " |" + u(agg[k]))
    for c in counters:
        tot[c] += agg[k].get(c, 0)
print("| **total** | | " + " | ".join(f"{tot[c]:.4g}" for c in counters) + " |" + u(tot))

if a.mfma_json and util:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    busy = {k: v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * v["SQ_BUSY_CYCLES"]) for k, v in agg.items() if v.get("SQ_BUSY_CYCLES")}
    top = max(agg, key=lambda k: agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))
    json.dump({"kernel_src_sha": bench.kernel_src_sha(), "workload": a.workload, "kernel": top, "mfma_busy": busy[top],
               "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): matrix-pipe busy cycles per SIMD over the cycles the "
                             "kernel was resident, of the kernel with the most MFMA cycles in the loop (rocprofv3 --pmc pass)",
               "all": {k: round(v, 4) for k, v in busy.items() if v > 0}}, open(a.mfma_json, "w"))

if a.traffic_json:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # prepare-time launches (fp32 GEMM tiles, casts, gathers of foley_prepare) are excluded: loop kernels only
    loop = {k: v for k, v in agg.items() if not k.startswith(("cast_kernel", "add_periodic", "qkv_split_kernel", "gemm_kernel<float"))}
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in loop.values()) * 1024.0
    write = sum(v.get("WRITE_SIZE", 0.0) for v in loop.values()) * 1024.0
    json.dump({"kernel_src_sha": bench.kernel_src_sha(), "workload": a.workload, "iterations_traced": a.iters,
               "fetch_bytes_raw": fetch, "write_bytes": write,
               "hbm_bytes_per_loop_iteration": (2.0 * fetch + write) / a.iters,
               "correction": "2 x FETCH_SIZE (gfx950 half-count of 16 B/lane coalesced reads) + WRITE_SIZE, KiB -> bytes; "
                             "memory-side (fabric) requests of the 8 L2s, Infinity-Cache hits included"},
              open(a.traffic_json, "w"))
