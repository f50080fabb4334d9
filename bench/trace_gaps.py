#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --kernel-trace CSV of bench.py: step span, kernel time per family on the last step,
idle time on the main queue, and what the tiny BatchNorm finalize kernels cost end to end (previous kernel's end ->
next kernel's start on the same queue).

    python bench/trace_gaps.py gpurun_out/<dir>/**/*_kernel_trace.csv [top_n] [list]

With ``list`` every launch of the last step is printed in start order (offset into the step, duration, queue, grid,
workgroup, LDS bytes, name) — the per-layer view of the real step rather than of an isolated layer loop.
"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("flat_sgd") or r["Kernel_Name"].startswith("flat_adam")]
    if len(marks) < 2:
        print("fewer than two optimizer launches in the trace")
        return
    step = rows[marks[-2] + 1:marks[-1] + 1]
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
    print(f"last step: {len(step)} kernels, span {span:.3f} ms")
    agg = collections.OrderedDict()
    for r in step:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:64]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += d
    for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print(f"  {d:8.3f} ms {c:5d}  {n}")
    queues = collections.Counter(r["Queue_Id"] for r in step)
    mainq = queues.most_common(1)[0][0]
    m = [r for r in step if r["Queue_Id"] == mainq]
    idle = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(m, m[1:])) / 1e6
    print(f"queues {dict(queues)}; main queue {mainq}: {len(m)} kernels, idle between kernels {idle:.3f} ms")
    if len(sys.argv) > 3 and sys.argv[3] == "list":
        t0 = int(step[0]["Start_Timestamp"])
        for r in step:
            n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            n = n.split("(")[0][:60]
            s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"    +{(s_ - t0) / 1e6:8.3f} {(e_ - s_) / 1e6:7.3f} ms q{r['Queue_Id']} grid {r.get('Grid_Size_X', '?'):>8} "
                  f"wg {r.get('Workgroup_Size_X', '?'):>4} lds {r.get('LDS_Block_Size', '?'):>6} {n}")
    for key in ("bn_bwd_finalize", "bn_finalize"):
        tot = n = 0
        for i in range(1, len(m) - 1):
            if m[i]["Kernel_Name"].startswith(key):
                tot += int(m[i + 1]["Start_Timestamp"]) - int(m[i - 1]["End_Timestamp"])
                n += 1
        if n:
            print(f"  {key}: {n} launches cost {tot / 1e6:.3f} ms end to end (prev end -> next start)")


if __name__ == "__main__":
    main()
