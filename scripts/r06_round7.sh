#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_round7; mkdir -p $out
timeout 2700 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
python3 bench.py --emulate-world 2,4,8 --measure-exchange --steps 5 --warmup 2 > gpurun_out/r06_emulate_config3.jsonl 2> $out/emulate.err
python3 -c "
import json
for l in open('gpurun_out/r06_emulate_config3.jsonl'):
    if l.startswith('{'):
        j=json.loads(l); print(j['n_gpus_emulated'], [round(x,2) for x in j['per_rank_ms_per_step']], 'x', round(j['projected_speedup_compute_only'],2), round(j['projected_speedup_with_modelled_exchange'],2), round(j['exchange_model_ms_per_step'],2), j['exchange_model'][70:140])
"
timeout 600 python3 scripts/knn_stress.py 120 621 > $out/knn_stress_120_seed621.log 2>&1; tail -1 $out/knn_stress_120_seed621.log
