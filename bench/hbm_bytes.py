"""Fabric (HBM / Infinity Cache side) traffic per kernel family of ONE training step, from a rocprofv3 --pmc run of bench.py
with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (bench/hbm_bytes.sh).

Read requests are 32 B or 64 B (RDREQ_32B counts the short ones), write requests 32 B or 64 B (WRREQ_64B the long ones);
the guide's caveat for gfx950 (wide coalesced reads are tallied at half their size) is reported as a second column.

    python bench/hbm_bytes.py <counter_collection.csv> [steps_in_trace]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                                                "t0": int(r.get("Start_Timestamp", 0) or 0), "t1": int(r.get("End_Timestamp", 0) or 0)})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
order = sorted(disp)
marks = [i for i, k in enumerate(order) if disp[k]["name"].startswith("flat_sgd")]
if len(marks) >= 2:
    order = order[marks[-2] + 1:marks[-1] + 1]
fam = collections.OrderedDict()
for k in order:
    d = disp[k]
    n = d["name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48]
    rd = d.get("TCC_EA0_RDREQ_sum", 0.0); rd32 = d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    wr = d.get("TCC_EA0_WRREQ_sum", 0.0); wr64 = d.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    rbytes = (rd - rd32) * 64 + rd32 * 32
    wbytes = wr64 * 64 + (wr - wr64) * 32
    f = fam.setdefault(n, [0, 0.0, 0.0, 0.0])
    f[0] += 1; f[1] += rbytes; f[2] += wbytes; f[3] += (d["t1"] - d["t0"]) / 1e6
tot_r = sum(f[1] for f in fam.values()); tot_w = sum(f[2] for f in fam.values()); tot_t = sum(f[3] for f in fam.values())
print(f"one step: {len(order)} dispatches; fabric reads {tot_r / 1e9:.1f} GB (x2 if wide reads are tallied at half size: "
      f"{2 * tot_r / 1e9:.1f} GB), writes {tot_w / 1e9:.1f} GB; kernel time under counters {tot_t:.1f} ms")
print(f"{'kernel family':50s} {'n':>4s} {'read GB':>9s} {'write GB':>9s} {'ms':>8s} {'(r+w)/ms = TB/s':>16s} {'(2r+w)/ms':>10s}")
for n, f in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    if f[1] + f[2] < 1e7:
        continue
    ms = max(f[3], 1e-9)
    print(f"{n:50s} {f[0]:4d} {f[1] / 1e9:9.2f} {f[2] / 1e9:9.2f} {f[3]:8.2f} {(f[1] + f[2]) / ms / 1e9:16.2f} {(2 * f[1] + f[2]) / ms / 1e9:10.2f}")
