#!/bin/bash
# filter gradients per ResNet-50 layer with the halo kernel off / on (bench/conv_layers.py <batch> w)
set -o pipefail
mkdir -p gpurun_out
B=${1:-1000}
DDL_WGRAD_HALO=0 timeout -k 10 200 python -u bench/conv_layers.py $B w > gpurun_out/wgrad_halo_off_b$B.txt 2>&1 || { tail -5 gpurun_out/wgrad_halo_off_b$B.txt; exit 1; }
DDL_WGRAD_HALO=1 timeout -k 10 200 python -u bench/conv_layers.py $B w > gpurun_out/wgrad_halo_on_b$B.txt 2>&1 || { tail -5 gpurun_out/wgrad_halo_on_b$B.txt; exit 1; }
paste <(awk '{print $1,$2,$3,$4,$5,$6,$7}' gpurun_out/wgrad_halo_off_b$B.txt) <(grep -o "| *-* *[0-9.]* *[0-9]* *[0-9]*% *| [A-Z]*$\|total.*" gpurun_out/wgrad_halo_off_b$B.txt) <(grep -o "| *-* *[0-9.]* *[0-9]* *[0-9]*% *| [A-Z]*$\|total.*" gpurun_out/wgrad_halo_on_b$B.txt) | tail -30
