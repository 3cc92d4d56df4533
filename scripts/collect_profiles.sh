#!/bin/bash
# GPU box: the rocprofv3 evidence behind bench.py's line for one workload, written under gpurun_out/profiles_<tag>/ :
# kernel statistics of the bench command, and three separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) turned
# into ${RND}_traffic_<workload>.json.  Copy what should be judged into profiles/.
#   scripts/collect_profiles.sh <tag> <workload> [<round prefix, default r03>]
tag=$1; wl=${2:-config3}; RND=${3:-r03}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/profiles_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --workload $wl > "$out/bench_stats.json" 2> "$out/stats.err"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$n" -o p -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host --workload $wl > /dev/null 2> "$out/pmc_$n.err"
done
cd "$root"
cp "$out/stats/p_kernel_stats.csv" "$out/${RND}_bench_${wl}_kernel_stats.csv"
for n in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do cp "$out/pmc_$n/p_counter_collection.csv" "$out/${RND}_pmc_${n}_${wl}.csv"; done
python3 scripts/traffic_json.py $wl "$out/${RND}_pmc_FETCH_SIZE_${wl}.csv" "$out/${RND}_pmc_WRITE_SIZE_${wl}.csv" "$out/${RND}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_${wl}.csv" "$out/${RND}_traffic_${wl}.json" | tail -3
rm -rf "$out/stats" "$out"/pmc_*/
python3 scripts/kstats.py "$out" 12 2>/dev/null || head -12 "$out/${RND}_bench_${wl}_kernel_stats.csv" | cut -c1-150
