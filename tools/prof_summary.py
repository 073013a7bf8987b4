"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (markdown)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
only_ours = "--all" not in sys.argv
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
