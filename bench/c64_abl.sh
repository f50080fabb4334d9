#!/bin/bash
# 3x3 64 -> 64 kernel with parts compiled out (csrc/conv_c64.hip, -DC64_ABL=bits; variant libraries selected with DDL_LIB)
echo -n "full     "; timeout -k 10 200 python -u bench/conv_layers.py ${1:-1000} fd 2>&1 | grep "C  64 H 56 K  64 R3" | cut -c42-100 || exit 1
for a in 1 2 4 6 8 15; do
  echo -n "abl=$a   "
  DDL_LIB=$PWD/bench/probes/bin/libddl_c64_$a.so timeout -k 10 200 python -u bench/conv_layers.py ${1:-1000} fd 2>&1 | grep "C  64 H 56 K  64 R3" | cut -c42-100 || exit 1
done
