#!/bin/bash
# round 6: what the own batch's packed pairs (kept for the literal re-run) cost the tiled adjust_shift_variance: config 5 as named,
# sigma 1, the product against timing builds without them (VARIANT=noso) and without the re-run altogether (VARIANT=nolit)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_asv_so_ab; mkdir -p $out
for v in main noso nolit main; do
  if [ $v = main ]; then unset BMX_LIB; else export BMX_LIB=$PWD/batchelor_amd/libbatchelor_mi355x_$v.so; fi
  timeout 600 python bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host > $out/$v.json 2> $out/$v.err
  python - <<PY
import json
for l in open("$out/$v.json"):
    l=l.strip()
    if l.startswith("{"):
        j=json.loads(l); r=j.get("roofline",{}); print("$v", j.get("ms_per_step"), r.get("avg_launch_ms"), r.get("phase_ms_per_workgroup"))
PY
done
