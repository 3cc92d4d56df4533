set -x
cd $GRAFT_REPO_ROOT
scripts/collect_profiles.sh r04c3 config3 r04 2>&1 | tail -3
scripts/collect_profiles.sh r04c2 config2 r04 2>&1 | tail -2
scripts/collect_profiles.sh r04c5 config5 r04 2>&1 | tail -2
# timeline of config 3
out=$PWD/gpurun_out/tl_final; rm -rf $out; mkdir -p $out; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > $out/stdout.txt 2> $out/stderr.txt); python3 scripts/timeline.py $out 3 21 > gpurun_out/r04_timeline_config3.txt; find $out -name "*.csv" -delete
# config 5 with variance adjustment: kernel stats + the tiled kernel's counters
scripts/prof.sh r04c5va bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host 2>&1 | tail -3
cp gpurun_out/prof_r04c5va/p_kernel_stats.csv gpurun_out/r04_bench_config5_varadj_kernel_stats.csv
grep -E '^\{' gpurun_out/prof_r04c5va/stdout.txt | tail -1 > gpurun_out/r04_bench_config5_varadj.json
scripts/pmc.sh asv1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
scripts/pmc.sh asv2 FETCH_SIZE asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
scripts/pmc.sh asv3 WRITE_SIZE asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
python3 scripts/asv_pmc.py gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3 gpurun_out/r04_asv_tile_pmc.json | tail -3
rm -rf gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3
# config 4 as named, with kernel statistics
scripts/prof.sh r04c4 bench.py --workload config4 --cells 200000 --gen-threads 12 2>&1 | tail -3
cp gpurun_out/prof_r04c4/p_kernel_stats.csv gpurun_out/r04_bench_config4_full_kernel_stats.csv
grep -E '^\{' gpurun_out/prof_r04c4/stdout.txt | tail -1 > gpurun_out/r04_bench_config4_full.json
# the bench lines
python3 bench.py > gpurun_out/r04_bench_config3.json 2> gpurun_out/r04_bench_config3.err
python3 bench.py --workload config2 --no-cpu-baseline > gpurun_out/r04_bench_config2.json 2>/dev/null
python3 bench.py --workload config5 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r04_bench_config5.json 2>/dev/null
python3 bench.py --workload sgk > gpurun_out/r04_bench_sgk.json 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +5M -delete
ls -la gpurun_out | tail -30
