#!/bin/bash
# round 6: full GPU test suite, 8-rank emulation again, the two legacy natives' lines with cpu_baseline
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_round2; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q -s > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log; tail -4 $out/gpu_tests.log; grep -h "config 5 at full size" $out/gpu_tests.log | cut -c1-700
python3 bench.py --emulate-world 8 --measure-exchange --steps 5 --warmup 2 > $out/emulate8.jsonl 2> $out/emulate8.err
python3 -c "
import json
for l in open('$out/emulate8.jsonl'):
    if l.startswith('{'):
        j=json.loads(l); print(j['n_gpus_emulated'], [round(x,2) for x in j['per_rank_ms_per_step']], 'x', round(j['projected_speedup_with_modelled_exchange'],2))
"
python3 bench.py --workload sgk > $out/bench_sgk.json 2> $out/bench_sgk.err; cut -c1-900 $out/bench_sgk.json
python3 bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-host-to-host > $out/bench_c5va.json 2> $out/bench_c5va.err
python3 -c "
import json
for l in open('$out/bench_c5va.json'):
    if l.startswith('{'):
        j=json.loads(l); print('c5va', round(j['ms_per_step']), j['roofline']['frac'], j['roofline'].get('phase_ms_per_workgroup'), {k:(v if k!='sample' else v[:200]) for k,v in j['cpu_baseline'].items()})
"
