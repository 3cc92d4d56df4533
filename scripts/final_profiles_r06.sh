set -x
cd $GRAFT_REPO_ROOT
R=r06
scripts/collect_profiles.sh ${R}c3 config3 $R 2>&1 | tail -3
scripts/collect_profiles.sh ${R}c2 config2 $R 2>&1 | tail -2
scripts/collect_profiles.sh ${R}c5 config5 $R 2>&1 | tail -2
for c in 3 2 5; do cp gpurun_out/profiles_${R}c$c/${R}_* gpurun_out/ 2>/dev/null; done
# timeline of config 3
out=$PWD/gpurun_out/tl_final; rm -rf $out; mkdir -p $out; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > $out/stdout.txt 2> $out/stderr.txt); python3 scripts/timeline.py $out 3 21 > gpurun_out/${R}_timeline_config3.txt; find $out -name "*.csv" -delete
# the emulated rank 3 of 8, timeline
out=$PWD/gpurun_out/tl_emu; rm -rf $out; mkdir -p $out; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --emulate-ranks 3 --steps 3 --warmup 1 > $out/stdout.txt 2> $out/stderr.txt); python3 scripts/timeline.py $out 3 21 > gpurun_out/${R}_timeline_emulated_rank3_of_8.txt; find $out -name "*.csv" -delete
# config 5 with variance adjustment at sigma 1 and 0.1: kernel stats + the tiled kernel's counters
for s in 1.0 0.1; do
  scripts/prof.sh ${R}c5va$s bench.py --workload config5 --var-adj --sigma $s --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host 2>&1 | tail -3
  cp gpurun_out/prof_${R}c5va$s/p_kernel_stats.csv gpurun_out/${R}_bench_config5_varadj_sigma${s}_kernel_stats.csv
done
scripts/pmc.sh asv1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
scripts/pmc.sh asv2 FETCH_SIZE asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
scripts/pmc.sh asv3 WRITE_SIZE asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
python3 scripts/asv_pmc.py gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3 gpurun_out/${R}_asv_tile_pmc.json | tail -3
rm -rf gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3
# smooth_gaussian_kernel: kernel statistics
scripts/prof.sh ${R}sgk bench.py --workload sgk --no-cpu-baseline 2>&1 | tail -3
cp gpurun_out/prof_${R}sgk/p_kernel_stats.csv gpurun_out/${R}_bench_sgk_kernel_stats.csv
# config 4 as named, with kernel statistics
scripts/prof.sh ${R}c4 bench.py --workload config4 --cells 200000 --gen-threads 12 2>&1 | tail -3
cp gpurun_out/prof_${R}c4/p_kernel_stats.csv gpurun_out/${R}_bench_config4_full_kernel_stats.csv
grep -E '^\{' gpurun_out/prof_${R}c4/stdout.txt | tail -1 > gpurun_out/${R}_bench_config4_full.json
# the bench lines, un-profiled
python3 bench.py > gpurun_out/${R}_bench_config3.json 2> gpurun_out/${R}_bench_config3.err
python3 bench.py --workload config2 --no-cpu-baseline > gpurun_out/${R}_bench_config2.json 2>/dev/null
python3 bench.py --workload config5 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${R}_bench_config5.json 2>/dev/null
python3 bench.py --workload config5 --var-adj --sigma 1.0 --steps 2 --warmup 1 --no-host-to-host > gpurun_out/${R}_bench_config5_varadj_sigma1.0.json 2>/dev/null
python3 bench.py --workload config5 --var-adj --sigma 0.1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-to-host > gpurun_out/${R}_bench_config5_varadj_sigma0.1.json 2>/dev/null
python3 bench.py --workload sgk > gpurun_out/${R}_bench_sgk.json 2>/dev/null
python3 bench.py --emulate-world 2,4,8 --measure-exchange --steps 5 --warmup 2 > gpurun_out/${R}_emulate_config3.jsonl 2>/dev/null
python3 bench.py --workload config5 --emulate-world 8 --emulate-ranks 0,3,7 --steps 3 --warmup 1 > gpurun_out/${R}_emulate_config5.jsonl 2>/dev/null
python3 scripts/large_k_probe.py 100000 20 36 37 50 100 1000 > gpurun_out/${R}_large_k_config2.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +5M -delete
ls gpurun_out | grep "^${R}_" | head -60
