#!/bin/bash
# A/B of the tiled adjust_shift_variance on one box: builds given as arguments (names after libbatchelor_mi355x_), "main" = the product
cd "$(dirname "$0")/.."
S=${SIGMAS:-1.0}
for s in $S; do
  for v in "$@"; do
    if [ $v = main ]; then unset BMX_LIB; else export BMX_LIB=$PWD/batchelor_amd/libbatchelor_mi355x_$v.so; fi
    for cap in ${CAPS:--1 0}; do
      echo -n "== $v cap $cap sigma $s: "; BMX_ASV_CAP=$cap python scripts/asv_probe.py 100000 300000 100000 100 $s | tail -2 | cut -c60- | tr '\n' ' '; echo
    done
  done
done
