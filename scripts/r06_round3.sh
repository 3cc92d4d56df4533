#!/bin/bash
# round 6: the k > 900 merge, the default bench line with CPU baseline T, random stress on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_round3; mkdir -p $out $out/stress
timeout 1500 python -m pytest tests/test_gpu_knn.py tests/test_gpu_fullsize.py -m gpu -x -q -s > $out/knn_tests.log 2>&1; echo "rc=$?" >> $out/knn_tests.log; tail -3 $out/knn_tests.log; grep -h "query_knn k = 5000" $out/knn_tests.log
python3 bench.py --steps 10 --warmup 3 > $out/bench_config3.json 2> $out/bench_config3.err
python3 -c "
import json
for l in open('$out/bench_config3.json'):
    if l.startswith('{'):
        j=json.loads(l); print('config3', round(j['ms_per_step'],2), 'frac', round(j['roofline']['frac'],4), 'h2h', round(j['host_to_host_ms'],1), 'cpu', j['cpu_baseline']['value'], j['cpu_baseline'].get('gflops'), j['cpu_baseline']['sample'][:160])
        for k,v in j['cpu_baseline_variants'].items(): print('   ', k, round(v['value'],1), v.get('gflops'), v.get('frac_of_host_fp64_peak'))
"
timeout 900 python3 scripts/knn_stress.py 200 601 > $out/stress/knn_stress_200_seed601.log 2>&1; tail -2 $out/stress/knn_stress_200_seed601.log
timeout 900 python3 scripts/engine_stress.py 150 603 > $out/stress/engine_stress_150_seed603.log 2>&1; tail -2 $out/stress/engine_stress_150_seed603.log
timeout 900 python3 scripts/natives_stress.py 60 602 > $out/stress/natives_stress_60_seed602.log 2>&1; tail -2 $out/stress/natives_stress_60_seed602.log
timeout 600 python3 scripts/knn_stress_large.py 6 604 > $out/stress/knn_stress_large_6_seed604.log 2>&1; tail -2 $out/stress/knn_stress_large_6_seed604.log
