"""Per-step kernel summary from a rocprofv3 (rocpd sqlite) kernel trace of bench.py.

    python bench/prof_summary.py gpurun_out/prof/x_results.db [steps_to_average]
"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if "flat_sgd" in n or "flat_adam" in n]
a, b = idx[-nsteps - 1], idx[-1]
span = (rows[b][1] - rows[a][1]) / nsteps


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return n[:72]


tot, cnt = defaultdict(float), defaultdict(int)
events = []
for n, s, e, st in rows[a:b]:
    k = short(n)
    tot[k] += (e - s) / nsteps
    cnt[k] += 1
    events.append((s, 1))
    events.append((e, -1))
events.sort()
busy, depth, last = 0, 0, None
for t, d in events:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"step span {span / 1e6:.3f} ms | sum of kernel durations {sum(tot.values()) / 1e6:.3f} ms | "
      f"GPU non-idle {busy / nsteps / 1e6:.3f} ms")
for k, t in sorted(tot.items(), key=lambda x: -x[1])[:32]:
    print(f"{t / 1e6:8.3f} ms {cnt[k] / nsteps:7.1f}  {k}")
