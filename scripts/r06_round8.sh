#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_round8; mkdir -p $out
for s in 0 1; do python3 scripts/asv_phase_probe.py 100000 400000 100 1.0 asv_sync=$s 2>&1 | grep "asv " | tee -a $out/phase.txt; done
for s in 0 1; do
python3 bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host --dev asv_sync=$s > $out/c5va_sync$s.json 2> $out/c5va.err
python3 -c "
import json
for l in open('$out/c5va_sync$s.json'):
    if l.startswith('{'):
        j=json.loads(l); print('config5 var-adj sync=$s ms/step', round(j['ms_per_step']), 'frac', round(j['roofline']['frac'],3), j['roofline'].get('phase_ms_per_workgroup_per_step'))
"
done
timeout 900 python -m pytest tests/test_gpu_config5.py -m gpu -x -q -k "by_mode" -s 2>&1 | tail -4 | cut -c1-600
