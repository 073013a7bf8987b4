"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (markdown)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
only_ours = "--all" not in sys.argv     # --all: also list the framework's own kernels (at::native ..., e.g. the conditioning encoders' torch glue)
rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc"))
if only_ours:
    rows = [r for r in rows if "at::native" not in r[0] and "rocclr" not in r[0]]
tot = sum(r[2] for r in rows)
print(f"| kernel | calls | total us | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|")
for r in rows:
    n = r[0].replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
    print(f"| `{n[:90]}` | {r[1]} | {r[2]:.0f} | {100 * r[2] / tot:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} |")
print(f"\ntotal {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} launches")

if "--dominant-json" in sys.argv:
    # the GEMM kernel with the largest total time = the kernel bench.py's roofline.frac is quoted on; bench.py reads this file
    # (same source-hash rule as the PMC files) and reports roofline.frac_rocprof from its traced average
    import hashlib, json, os
    out = sys.argv[sys.argv.index("--dominant-json") + 1]
    workload = sys.argv[sys.argv.index("--workload") + 1]
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-hunyuanvideo-foley_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(csrc)):
        if n.endswith((".hip", ".h")):
            h.update(n.encode())
            h.update(open(os.path.join(csrc, n), "rb").read())
    ks = [{"name": r[0].replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")[:120], "calls": r[1], "avg_us": round(r[3], 3)}
          for r in rows if "gemm_" in r[0]]
    json.dump({"kernel_src_sha": h.hexdigest()[:16], "workload": workload, "loop_iterations": 100, "kernels": ks,
               "source": "rocprofv3 --kernel-trace --stats of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra` "
                         "(two 50-iteration loops: the timed pass and the event-timed one)"}, open(out, "w"))
