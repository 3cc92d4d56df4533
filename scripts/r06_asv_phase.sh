#!/bin/bash
# round 6: phases of the tiled adjust_shift_variance (stream / barrier / per-cell), with and without the round barrier
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_asv_phase; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_primitives.py -x -q -k "adjust_shift" > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log; tail -3 $out/tests.log
for s in 0 1; do
  python3 scripts/asv_phase_probe.py 100000 400000 100 1.0 asv_sync=$s 2>&1 | grep asv | tee $out/phase_sync$s.txt
done
python3 scripts/asv_phase_probe.py 100000 400000 50 1.0 asv_sync=0 2>&1 | grep asv | tee $out/phase_d50.txt
python3 scripts/asv_phase_probe.py 60000 200000 100 0.1 asv_sync=0 2>&1 | grep asv | tee $out/phase_s01.txt
for s in 0 1; do
  timeout 600 python bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host --dev asv_sync=$s > $out/c5_sync$s.json 2> $out/c5_sync$s.err
  python3 -c "
import json
for l in open('$out/c5_sync$s.json'):
    if l.startswith('{'):
        j=json.loads(l); print('config5 var-adj asv_sync=$s ms/step', j['ms_per_step'], 'frac', j['roofline']['frac'])
"
done
