#!/bin/bash
# round 6: one rank's share of an N-rank run (bench.py --emulate-world), the exchange's per-call term measured, config 3 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_emulate; mkdir -p $out
python3 bench.py --steps 10 --warmup 3 > $out/bench_config3.json 2> $out/bench_config3.err; tail -c 1500 $out/bench_config3.json | head -c 600; echo
python3 bench.py --emulate-world 2,4,8 --measure-exchange --steps 5 --warmup 2 > $out/emulate_config3.jsonl 2> $out/emulate_config3.err
python3 - <<PY
import json
for l in open("$out/emulate_config3.jsonl"):
    if l.startswith("{"):
        j=json.loads(l)
        print(j["n_gpus_emulated"], "max rank ms", round(j["max_rank_ms_per_step"],2), "1gpu", round(j["ms_per_step_one_gpu"],2), "compute-only x", round(j["projected_speedup_compute_only"],2), "with exchange x", round(j["projected_speedup_with_modelled_exchange"],2), "xchg ms", round(j["exchange_model_ms_per_step"],2), "calls", j["exchange_calls_per_step"], "MB", round(j["exchange_bytes_per_step"]/1e6))
        print("   ", j["exchange_model"][:300])
PY
tail -3 $out/emulate_config3.err
python3 scripts/tau_replay_probe.py config3 5 2>&1 | grep -v amdgpu.ids | tee $out/tau_replay.txt
for v in "" _r5asv; do
  BMX_LIB=$PWD/batchelor_amd/libbatchelor_mi355x$v.so timeout 600 python3 bench.py --workload config5 --var-adj --sigma 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host > $out/c5va$v.json 2> $out/c5va$v.err
  python3 -c "
import json
for l in open('$out/c5va$v.json'):
    if l.startswith('{'):
        j=json.loads(l); print('config5 var-adj [$v] ms/step', round(j['ms_per_step']), 'frac', round(j['roofline']['frac'],3), j['roofline'].get('phase_ms_per_workgroup'))
"
done
