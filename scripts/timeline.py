"""Developer helper (GPU box): where a step's time goes BETWEEN kernels.  Reads rocprofv3's kernel trace (and memory-copy
trace when present) of a bench.py run, takes the window of the last `steps` timed steps and prints: kernel time per class,
the idle time between consecutive GPU activities (by the activity that follows the gap), and the largest gaps.
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o p -- python3 bench.py --steps 3 --warmup 1 ...
   python scripts/timeline.py <dir> <steps> <launches of the dominant kernel per step>"""
import csv
import glob
import re
import sys
from collections import defaultdict

d, steps, per_step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"bmx::\(anonymous namespace\)::|bmx::", "", name)
        name = re.sub(r"\(.*$", "", name)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy:" + r.get("Direction", "")))
ev.sort()
dom = [i for i, e in enumerate(ev) if e[2].startswith("knn_topk_f16") and "false>" in e[2]]
first = dom[-steps * per_step]
# the window starts at the first activity after the previous dominant launch's step boundary: back up to the transpose
i0 = first
while i0 > 0 and not ev[i0][2].startswith("transpose_kernel"):
    i0 -= 1
while i0 > 0 and ev[i0 - 1][2].startswith(("transpose_kernel", "memcpy", "__amd_rocclr_fill")):
    i0 -= 1
win = ev[i0:]
t0, t1 = win[0][0], max(e[1] for e in win)
busy = defaultdict(float)
cnt = defaultdict(int)
gap_by = defaultdict(float)
gaps = []
end = win[0][0]
for s, e, n in win:
    busy[n] += (e - s) / 1e3
    cnt[n] += 1
    if s > end:
        g = (s - end) / 1e3
        gap_by[n] += g
        gaps.append((g, n))
    end = max(end, e)
tot = (t1 - t0) / 1e3
kb = sum(busy.values())
print(f"window {tot / steps / 1e3:.2f} ms/step, kernels+copies {kb / steps / 1e3:.2f} ms/step, idle {sum(g for g, _ in gaps) / steps / 1e3:.2f} ms/step in "
      f"{len(gaps) / steps:.0f} gaps/step, {len(win) / steps:.0f} activities/step")
print("--- busy per class (us/step, launches/step)")
for n, v in sorted(busy.items(), key=lambda x: -x[1])[:40]:
    print(f"{v / steps:10.1f} {cnt[n] / steps:7.1f}  {n[:110]}")
print("--- idle in front of (us/step)")
for n, v in sorted(gap_by.items(), key=lambda x: -x[1])[:25]:
    print(f"{v / steps:10.1f}  {n[:110]}")
print("--- largest gaps (us)")
for g, n in sorted(gaps, reverse=True)[:15]:
    print(f"{g:10.1f}  before {n[:100]}")
