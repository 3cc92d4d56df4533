#!/bin/bash
# round 6: the tiled adjust_shift_variance after a change: phases on the mid-size call, the primitives' parity tests, config 5
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_asv_check; mkdir -p $out
python3 scripts/asv_phase_probe.py 100000 400000 100 1.0 2>&1 | grep "asv " | tee $out/phase.txt
python3 scripts/asv_phase_probe.py 60000 200000 100 0.1 2>&1 | grep "asv " | tee -a $out/phase.txt
timeout 1200 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q > $out/prim_tests.log 2>&1; tail -2 $out/prim_tests.log
for s in 1.0 0.1; do
python3 bench.py --workload config5 --var-adj --sigma $s --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host > $out/c5va_$s.json 2> $out/c5va.err
python3 -c "
import json
for l in open('$out/c5va_$s.json'):
    if l.startswith('{'):
        j=json.loads(l); print('config5 var-adj sigma $s ms/step', round(j['ms_per_step']), 'frac', round(j['roofline']['frac'],3), j['roofline'].get('phase_ms_per_workgroup_per_step'))
"
done
