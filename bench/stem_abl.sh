#!/bin/bash
# stem forward kernel with parts compiled out (csrc/conv_stem.hip, -DSTEM_ABL=bits; build: see bench/README.md)
mkdir -p gpurun_out
echo -n "full    "; timeout -k 10 120 python -u bench/stem_probe.py ${1:-1000} 2>&1 | grep "H 115" | cut -c1-60 || exit 1
for a in 1 2 3 4 8 15; do
  echo -n "abl=$a   "
  DDL_LIB=$PWD/bench/probes/bin/libddl_stem$a.so timeout -k 10 120 python -u bench/stem_probe.py ${1:-1000} 2>&1 | grep "H 115" | cut -c1-60 || exit 1
done
