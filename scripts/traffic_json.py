"""Derive profiles/r01_traffic_<workload>_bf16.json from the committed rocprofv3 PMC passes.

  python scripts/traffic_json.py <workload> <fetch_csv> <write_csv> <mfma_csv> <out_json>

The three CSVs are the counter_collection outputs of three separate runs of
  rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 \
      --no-cpu-baseline --workload <workload>
with COUNTERS = FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request, see
MI355X_MICROARCH.md, HBM section), per launch of knn_topk_bf16 (sample and full passes together, like bench.py's
HIP-event average)."""
import csv, json, sys, collections


def per_kernel(path, kinds):
    acc = {k: collections.defaultdict(lambda: collections.defaultdict(float)) for k in kinds}
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "knn_topk_bf16" not in n:
            continue
        kind = "sample" if ", true>" in n else "full"
        acc[kind][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return acc


workload, fcsv, wcsv, mcsv, out = sys.argv[1:6]
kinds = ("sample", "full")
F, W, M = per_kernel(fcsv, kinds), per_kernel(wcsv, kinds), per_kernel(mcsv, kinds)
rec = {"kernel": "knn_topk_bf16", "workload": workload, "variant": 2,
       "method": __doc__.split("\n\n")[1].replace("\n", " "), "per_kernel": {}}
tot_bytes, tot_launch = 0.0, 0
for k in kinds:
    f, w = F[k]["FETCH_SIZE"], W[k]["WRITE_SIZE"]
    if not f:
        continue
    nl = len(f)
    fm, wm = sum(f.values()) / nl, sum(w.values()) / max(1, len(w))
    ent = {"launches": nl, "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm, "hbm_bytes": (2 * fm + wm) * 1024}
    for c, v in M[k].items():
        ent[c] = sum(v.values()) / len(v)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and ent.get("GRBM_GUI_ACTIVE"):
        # 1024 SIMDs (256 CUs x 4) share the busy count
        ent["mfma_busy_frac"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (ent["GRBM_GUI_ACTIVE"] * 1024 / 8)
    rec["per_kernel"][k] = ent
    tot_bytes += ent["hbm_bytes"] * nl
    tot_launch += nl
rec["bytes_per_launch"] = tot_bytes / tot_launch
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec["per_kernel"], indent=1)[:1500], rec["bytes_per_launch"])
