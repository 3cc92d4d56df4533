#!/bin/bash
# Developer helper (GPU box): times the full-pass candidate kernel of one kNN shape for several builds of the library
# (make VARIANT=...): scripts/ab.sh <nq> <d> <nref> <variant> [<variant> ...]   ("base" = the product library).
# The search splits the reference as the product does; BMX_FORCE_C / BMX_SPLIT_C in the environment override that.
nq=$1; d=$2; nr=$3; shift 3
root=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  if [ "$v" = base ]; then unset BMX_LIB; else export BMX_LIB=$root/batchelor_amd/libbatchelor_mi355x_$v.so; fi
  timeout 180 scripts/prof.sh ab_$v scripts/knn_probe.py $nq $d $nr > /dev/null 2>&1
  echo "$v: $(grep -E 'knn_topk_f16.*false|knn_topk_bf16.*false' $root/gpurun_out/prof_ab_$v/summary.txt | head -1 | sed 's/.*avg_ms=//') ms full pass, sample $(grep -E 'knn_topk_f16.*true' $root/gpurun_out/prof_ab_$v/summary.txt | head -1 | sed 's/.*avg_ms=//')"
done
