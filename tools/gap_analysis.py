"""From a rocprofv3 kernel-trace database: how much of the loop's wall time has NO kernel running
(launch gaps inside the replayed hipGraph), and the per-boundary gap distribution."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
rows = [r for r in rows if "at::native" not in r[0] and "rocclr" not in r[0]]
# keep the longest run of kernels that belongs to the sampler loop: from the first solver_step to the last
idx = [i for i, r in enumerate(rows) if "solver_step" in r[0]]
lo, hi = idx[0], idx[-1]
seg = rows[lo + 1: hi + 1]
t0, t1 = seg[0][1], seg[-1][2]
busy, cur_end, gaps = 0, seg[0][1], []
for _n, s, e in seg:
    if s > cur_end:
        gaps.append((s - cur_end) / 1e3)
        busy += 0
        cur_start = s
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
wall = (t1 - t0) / 1e3
gaps.sort()
n = len(gaps)
print(f"{len(seg)} kernels over {len(idx) - 1} iterations: wall {wall:.0f} us, busy {busy / 1e3:.0f} us ({100 * busy / 1e3 / wall:.1f} %), "
      f"idle {wall - busy / 1e3:.0f} us in {n} gaps: median {gaps[n // 2]:.2f} us, p90 {gaps[int(n * 0.9)]:.2f} us, max {gaps[-1]:.1f} us")
