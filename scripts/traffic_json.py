"""Derive profiles/r02_traffic_<workload>.json from rocprofv3 PMC passes.

  python scripts/traffic_json.py <workload> <fetch_csv> <write_csv> <mfma_csv> <out_json>

The three CSVs are the counter_collection outputs of three separate runs of
  rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 \
      --no-cpu-baseline --no-host-to-host --workload <workload>
with COUNTERS = FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request, see
MI355X_MICROARCH.md, HBM section), per launch of the FULL-pass candidate kernel (the launches bench.py's
roofline.achieved is computed over); the sample passes are listed beside it."""
import collections
import csv
import json
import re
import sys


def kernel_of(name):
    m = re.search(r"(knn_topk_\w+<[^>]*>)", name)
    return m.group(1) if m else None


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = kernel_of(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return acc


workload, fcsv, wcsv, mcsv, out = sys.argv[1:6]
F, W, M = per_kernel(fcsv), per_kernel(wcsv), per_kernel(mcsv)
rec = {"workload": workload, "method": " ".join(__doc__.split("\n\n")[1:]).replace("\n", " "), "per_kernel": {}}
best = None
for k in sorted(F):
    f, w = F[k]["FETCH_SIZE"], W[k]["WRITE_SIZE"]
    if not f:
        continue
    nl = len(f)
    fm, wm = sum(f.values()) / nl, sum(w.values()) / max(1, len(w))
    ent = {"launches": nl, "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm, "hbm_bytes": (2 * fm + wm) * 1024}
    for c, v in M[k].items():
        ent[c] = sum(v.values()) / len(v)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and ent.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs (256 CUs x 4) share the busy count
        ent["mfma_busy_frac"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (ent["GRBM_GUI_ACTIVE"] * 1024 / 8)
    rec["per_kernel"][k] = ent
    if k.endswith("false>") and (best is None or ent["hbm_bytes"] * nl > best[1]):
        best = (k, ent["hbm_bytes"] * nl)
rec["kernel"] = best[0]
rec["bytes_per_launch"] = rec["per_kernel"][best[0]]["hbm_bytes"]
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec["per_kernel"], indent=1)[:2500], rec["kernel"], rec["bytes_per_launch"])
