set -x
cd $GRAFT_REPO_ROOT
R=r06
for s in 1.0 0.1; do
  scripts/prof.sh ${R}c5va$s bench.py --workload config5 --var-adj --sigma $s --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host 2>&1 | tail -3
  cp gpurun_out/prof_${R}c5va$s/p_kernel_stats.csv gpurun_out/${R}_bench_config5_varadj_sigma${s}_kernel_stats.csv
done
scripts/pmc.sh asv1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
scripts/pmc.sh asv2 FETCH_SIZE asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
scripts/pmc.sh asv3 WRITE_SIZE asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > /dev/null 2>&1
python3 scripts/asv_pmc.py gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3 gpurun_out/${R}_asv_tile_pmc.json | tail -3
rm -rf gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3
find gpurun_out -name "*kernel_trace.csv" -size +5M -delete
