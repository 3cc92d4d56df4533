#!/bin/bash
# Developer helper (GPU box): one rocprofv3 --pmc pass of a python command; per-kernel counter averages via pmcsum.py.
#   scripts/pmc.sh <tag> "<counters>" <kernel substring> <python args...>
tag=$1; ctrs=$2; kern=$3; shift 3
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o p -- python3 "$root/$1" "${@:2}" > "$out/stdout.txt" 2> "$out/stderr.txt"
cd "$root"
python3 scripts/pmcsum.py "$out" "$kern" | tee "$out/summary.txt"
