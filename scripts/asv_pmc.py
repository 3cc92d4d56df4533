"""Developer helper (GPU box): r03_asv_tile_pmc.json from the counter_collection CSVs of
   scripts/pmc.sh asv1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" asv_tile bench.py --workload config5 --var-adj --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host
   scripts/pmc.sh asv2 FETCH_SIZE asv_tile <same>;  scripts/pmc.sh asv3 WRITE_SIZE asv_tile <same>
   python scripts/asv_pmc.py gpurun_out/pmc_asv1 gpurun_out/pmc_asv2 gpurun_out/pmc_asv3 out.json"""
import collections, csv, glob, json, sys


def per_dispatch(d):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "asv_tile" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return per


a, f, w = (per_dispatch(x) for x in sys.argv[1:4])
# dispatch ids differ between runs: match by launch order
oa, of, ow = sorted(a), sorted(f), sorted(w)
rows = []
for i, d in enumerate(oa):
    c = a[d]
    gui = c["GRBM_GUI_ACTIVE"]
    simd_cycles = gui / 8 * 256 * 4
    rec = {"launch": i, "GRBM_GUI_ACTIVE": gui, "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
           "wait_any_frac": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
           "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
           "lds_bank_conflict_per_busy_cycle": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_BUSY_CYCLES"], 1.0)}
    if i < len(of) and i < len(ow):
        rec["hbm_read_bytes"] = 2 * f[of[i]]["FETCH_SIZE"] * 1024  # gfx950: FETCH_SIZE counts 64 B per 128 B request
        rec["hbm_write_bytes"] = w[ow[i]]["WRITE_SIZE"] * 1024
    rows.append(rec)
rows.sort(key=lambda r: -r["GRBM_GUI_ACTIVE"])
tot = {k: sum(r.get(k, 0.0) for r in rows) for k in ("hbm_read_bytes", "hbm_write_bytes")}
json.dump({"kernel": "asv_tile_kernel<13>", "workload": "config5 --var-adj",
           "method": __doc__, "all_launches_total": tot, "dispatches": rows[:3]}, open(sys.argv[4], "w"), indent=1)
print(json.dumps(rows[:3], indent=1)); print(tot)
