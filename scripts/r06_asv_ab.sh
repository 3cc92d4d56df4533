#!/bin/bash
# round 6: adjust_shift_variance tiled form with / without the round barrier (lock-step streams), config 5 as named
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_asv_ab; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_primitives.py -x -q -k "adjust_shift" > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
for s in 0 1; do
  timeout 600 python bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host --dev asv_sync=$s > $out/sync$s.json 2> $out/sync$s.err
done
tail -3 $out/tests.log; for s in 0 1; do python - <<PY
import json
for l in open("$out/sync$s.json"):
    l=l.strip()
    if l.startswith("{"):
        j=json.loads(l); print("asv_sync=$s", j.get("ms_per_step"), j.get("roofline"))
PY
done
