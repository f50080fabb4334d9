#!/bin/bash
# headline step with identity collectives forced and a W-workgroup streaming kernel behind every bucket, the proxy's
# blocks holding L KiB of LDS (0 / 48): does a communication kernel's LDS footprint push the one-block-per-CU
# kernels (halo filter gradient, big-tile forward) into a second round?
set -u
B=${1:-1000}
for rep in 1 2; do
  for cfg in "0 0" "32 0" "32 48" "64 48"; do
    set -- $cfg; W=$1; L=$2
    for halo in 1 0; do
      echo -n "proxy W=$W lds=${L}KiB DDL_WGRAD_HALO=$halo  "
      DDL_PROXY_LDS_KB=$L DDL_WGRAD_HALO=$halo timeout -k 10 300 python bench.py --batch $B --steps 20 --warmup 8 --force-collectives --comm-proxy $W 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'])" || exit 1
    done
  done
done
