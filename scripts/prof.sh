#!/bin/bash
# Developer helper (GPU box): rocprofv3 kernel statistics of a python command, CSV summary printed through kstats.py.
#   scripts/prof.sh <tag> <python args...>     e.g. scripts/prof.sh bench3 bench.py --steps 3 --warmup 1
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$root/$1" "${@:2}" > "$out/stdout.txt" 2> "$out/stderr.txt"
cd "$root"
python3 scripts/kstats.py "$out" 18 | tee "$out/summary.txt"
grep -E '^\{' "$out/stdout.txt" | tail -1 | cut -c1-400
