#!/bin/bash
# round 6: timing builds of the tiled adjust_shift_variance's stream with parts removed (results are wrong, the clock is right)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_asv_parts; mkdir -p $out
for v in "" _nostore _bare; do
  BMX_LIB=$PWD/batchelor_amd/libbatchelor_mi355x$v.so python3 scripts/asv_phase_probe.py 100000 400000 100 1.0 asv_sync=0 2>&1 | grep "asv " | tail -1 | sed "s/^/[$v] /" | tee -a $out/parts.txt
done
for v in "" _nostore _bare; do
  BMX_LIB=$PWD/batchelor_amd/libbatchelor_mi355x$v.so python3 scripts/asv_phase_probe.py 100000 400000 100 1.0 asv_sync=1 2>&1 | grep "asv " | tail -1 | sed "s/^/[$v sync] /" | tee -a $out/parts.txt
done
