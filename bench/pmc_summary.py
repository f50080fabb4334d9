"""Summarise the counter CSVs that bench/pmc.sh writes: python bench/pmc_summary.py <tag> <kernel substring>"""
import csv
import glob
import sys
from collections import defaultdict

tag, needle = sys.argv[1], sys.argv[2]
vals, n = defaultdict(float), defaultdict(int)
dur = []
for path in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if needle in row["Kernel_Name"]:
            vals[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
            dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
avg = {k: vals[k] / n[k] for k in vals}
print(f"[{tag}] kernel ~ '{needle}': {max(n.values()) if n else 0} dispatches, mean {sum(dur) / max(len(dur), 1) / 1e3:.1f} us (under counters)")
for k in sorted(avg):
    print(f"  {k:28s} {avg[k]:16.0f}")
g = avg.get
if g("SQ_WAVE_CYCLES"):
    print(f"  wave-cycles issuing / waiting any / waiting inst: {100 * g('SQ_ACTIVE_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.1f} % / "
          f"{100 * g('SQ_WAIT_ANY', 0) / g('SQ_WAVE_CYCLES'):.1f} % / {100 * g('SQ_WAIT_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.1f} %")
if g("SQ_LDS_IDX_ACTIVE"):
    print(f"  LDS bank-conflict cycles / LDS active cycles: {100 * g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.2f} %")
if g("TCC_HIT_sum") is not None and (g("TCC_HIT_sum", 0) + g("TCC_MISS_sum", 0)) > 0:
    print(f"  L2 hit rate: {100 * g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.1f} %   "
          f"fabric reads {g('TCC_EA0_RDREQ_sum', 0) * 64 / 1e6:.1f} MB (x2 for wide loads, see guide)  "
          f"writes {g('TCC_EA0_WRREQ_sum', 0) * 64 / 1e6:.1f} MB")
