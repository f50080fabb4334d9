#!/bin/bash
# usage: bench/r3_big_dbg3.sh "modes" B  -> GPU-side kernel durations (rocprofv3 kernel trace) of three layers per DDL_CONV_BIG
# mode (0 ring kernels, 2 big-tile kernel everywhere it can run); host-side event timing of 0.1 ms launches is host-bound
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
vals=$1; export DDL_ONE_B=${2:-512}
for d in $vals; do
  for shape in "512 7 512 3 1 1 f" "256 14 256 3 1 1 f" "256 14 1024 1 1 0 f"; do
    rm -rf gpurun_out/kt_tmp
    DDL_CONV_BIG=$d timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_tmp -- python3 bench/conv_one.py $shape > gpurun_out/kt_tmp.log 2>&1 || { echo "trace failed"; tail -3 gpurun_out/kt_tmp.log; exit 1; }
    python3 - "$d" "$shape" <<'PY'
import csv, glob, sys
rows = []
for path in glob.glob("gpurun_out/kt_tmp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "conv_" in r["Kernel_Name"] and "wgrad" not in r["Kernel_Name"]:
            rows.append((r["Kernel_Name"][:40], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
last = rows[-5:]
print(f"DDL_CONV_BIG={sys.argv[1]:>2s} shape [{sys.argv[2]}]  {last[0][0]:40s} " + " ".join(f"{d / 1e3:7.1f}" for _, d in last) + " us")
PY
  done
done
rm -rf gpurun_out/kt_tmp gpurun_out/kt_tmp.log
