"""Aggregate rocprofv3 --pmc results (rocpd sqlite) per kernel: sum of each counter + launches."""
import sqlite3
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for path in sys.argv[1:]:
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
print("| kernel | launches | " + " | ".join(counters) + " |")
print("|---|---|" + "---|" * len(counters))
tot = defaultdict(float)
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    n = max([cnt[(k, c)] for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES")] + [0])
    print(f"| `{k[:70]}` | {n} | " + " | ".join(f"{agg[k].get(c, 0):.4g}" for c in counters) + " |")
    for c in counters:
        tot[c] += agg[k].get(c, 0)
print("| **total** | | " + " | ".join(f"{tot[c]:.4g}" for c in counters) + " |")
