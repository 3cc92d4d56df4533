"""Developer helper (GPU box): a fresh engine's first run against its second run, kernel by kernel.
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fvw -o p -- python3 scripts/fresh_vs_warm.py run
   python scripts/fresh_vs_warm.py read gpurun_out/fvw"""
import sys, os, csv, glob, re
from collections import defaultdict
if sys.argv[1] == "run":
    import numpy as np
    sys.path.insert(0, os.getcwd())
    from bench import synth_batches
    from batchelor_amd import reduced_mnn as rm
    B = [np.asfortranarray(b) for b in synth_batches(3, [100000] * 8, 50)]
    for i in range(2):
        e = rm.MnnEngine(); e.upload(B); e.run(k=20); e.run(k=20); e.close()
    sys.exit(0)
ev = []
for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"bmx::\(anonymous namespace\)::|bmx::", "", name)
        name = re.sub(r"\(.*$", "", name)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
# runs are separated by the transposes of make_leaf: split at the 21-launch groups of the dominant kernel
dom = [i for i, e in enumerate(ev) if e[2].startswith("knn_topk_f16") and "false>" in e[2]]
assert len(dom) == 84, len(dom)
bounds = []
for r in range(4):
    i0 = dom[21 * r]
    while i0 > 0 and not ev[i0][2].startswith("transpose"):
        i0 -= 1
    while i0 > 0 and ev[i0 - 1][2].startswith("transpose"):
        i0 -= 1
    bounds.append(i0)
bounds.append(len(ev))
runs = [ev[bounds[r]:bounds[r + 1]] for r in range(4)]
def summarise(win):
    busy, cnt = defaultdict(float), defaultdict(int)
    idle, end = 0.0, win[0][0]
    for s, e, n in win:
        busy[n] += (e - s) / 1e3; cnt[n] += 1
        if s > end: idle += (s - end) / 1e3
        end = max(end, e)
    return busy, cnt, idle, (end - win[0][0]) / 1e3
fb, fc, fi, ft = summarise(runs[2]); wb, wc, wi, wt = summarise(runs[3])
print(f"second engine: fresh run {ft/1e3:.2f} ms (idle {fi/1e3:.2f}, {len(runs[2])} kernels)   warm run {wt/1e3:.2f} ms (idle {wi/1e3:.2f}, {len(runs[3])} kernels)")
print("--- kernel classes by (fresh - warm) busy time, us")
for n in sorted(set(fb) | set(wb), key=lambda n: -(fb.get(n, 0) - wb.get(n, 0)))[:25]:
    print(f"{fb.get(n,0)-wb.get(n,0):9.1f}   fresh {fb.get(n,0):9.1f} ({fc.get(n,0)})   warm {wb.get(n,0):9.1f} ({wc.get(n,0)})   {n[:90]}")
print("--- the full passes one by one, us (fresh / warm / ratio)")
fd = [(e - s) / 1e3 for s, e, n in runs[2] if n.startswith("knn_topk_f16") and "false>" in n]
wd = [(e - s) / 1e3 for s, e, n in runs[3] if n.startswith("knn_topk_f16") and "false>" in n]
print("  ".join(f"{a:.0f}/{b:.0f}/{a / b:.3f}" for a, b in zip(fd, wd)))
