#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_round4; mkdir -p $out
python3 scripts/host_cpu_probe.py 2>&1 | tee $out/host_cpu_probe.txt
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_abi.py -m gpu -x -q > $out/prim_tests.log 2>&1; tail -2 $out/prim_tests.log
python3 bench.py --workload sgk --no-cpu-baseline > $out/bench_sgk.json 2> $out/bench_sgk.err
python3 -c "
import json
for l in open('$out/bench_sgk.json'):
    if l.startswith('{'):
        j=json.loads(l); print('sgk ms', round(j['ms_per_step'],2), 'kernels', round(j['roofline']['kernel_ms'],2), 'frac', round(j['roofline']['frac'],3), 'err', j['config']['max_rel_err_vs_dense_spec_on_64_cells'])
"
timeout 600 python3 scripts/natives_stress.py 40 612 > $out/natives_stress_40_seed612.log 2>&1; tail -1 $out/natives_stress_40_seed612.log
