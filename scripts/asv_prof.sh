#!/bin/bash
# kernel times of the tiled adjust_shift_variance under rocprofv3, for the builds given as arguments ("main" = the product)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
S=${SIGMAS:-1.0}
for s in $S; do for v in "$@"; do for cap in ${CAPS:--1}; do
  if [ $v = main ]; then unset BMX_LIB; else export BMX_LIB=$root/batchelor_amd/libbatchelor_mi355x_$v.so; fi
  export BMX_ASV_CAP=$cap
  d=$root/gpurun_out/asvprof_${v}_cap${cap}_s$s; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $root/scripts/asv_probe.py 100000 300000 100000 100 $s > $d/stdout.txt 2> $d/stderr.txt
  echo "== $v cap $cap sigma $s"; tail -2 $d/stdout.txt; (cd $root && python3 scripts/kstats.py $d 5)
done; done; done
