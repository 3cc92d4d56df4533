#!/bin/bash
# round 6: the tiled adjust_shift_variance with the distance folded into the GEMM: parity tests, phases, config 5
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_round5; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_config5.py -m gpu -x -q -s > $out/asv_tests.log 2>&1; echo "rc=$?" >> $out/asv_tests.log; tail -3 $out/asv_tests.log; grep -h "config 5 at full size" $out/asv_tests.log | cut -c1-400
python3 scripts/asv_phase_probe.py 100000 400000 100 1.0 2>&1 | grep "asv " | tee $out/phase.txt
python3 scripts/asv_phase_probe.py 100000 400000 50 1.0 2>&1 | grep "asv " | tee -a $out/phase.txt
python3 bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host > $out/c5va.json 2> $out/c5va.err
python3 -c "
import json
for l in open('$out/c5va.json'):
    if l.startswith('{'):
        j=json.loads(l); print('config5 var-adj ms/step', round(j['ms_per_step']), 'frac', round(j['roofline']['frac'],3), j['roofline'].get('phase_ms_per_workgroup'))
"
timeout 600 python3 scripts/natives_stress.py 40 613 > $out/natives_stress_40_seed613.log 2>&1; tail -1 $out/natives_stress_40_seed613.log
