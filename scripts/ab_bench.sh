#!/bin/bash
# Developer helper (GPU box): bench.py per build of the library (make VARIANT=...): ms per step, the dominant kernel's
# mean launch time, queries that left the first tier.   scripts/ab_bench.sh <workload> <variant> [<variant> ...]
wl=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  if [ "$v" = base ]; then unset BMX_LIB; else export BMX_LIB=$root/batchelor_amd/libbatchelor_mi355x_$v.so; fi
  timeout 300 python3 bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-host-to-host 2> /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('$v', 'ms/step %.2f' % l['ms_per_step'], 'launch %.3f' % l['roofline']['avg_launch_ms'], 'cand %.2f' % l['per_rank']['candidate_pass_ms_per_step'], 'sample %.2f' % l['roofline']['other_candidate_passes_ms_per_step']['sample'], 'stream %.2f' % l['streaming']['ms_per_step'], 'exact', l['config']['exact_fallback_queries'], 'pairs', sum(l['config']['mnn_pairs']))
"
done
