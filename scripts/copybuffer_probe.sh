#!/bin/bash
# Developer helper (GPU box): which activities are the __amd_rocclr_copyBuffer launches of an emulated rank's step, and what runs around them
out=$PWD/gpurun_out/tl_cb; rm -rf $out; mkdir -p $out
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --emulate-ranks 3 --steps 2 --warmup 1 > $out/stdout.txt 2> $out/stderr.txt)
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], "k"))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")), "m"))
rows.sort()
n = len(rows)
lo = n - 700 if n > 700 else 0
for i in range(lo, n):
    s, e, name, kind = rows[i]
    if "copyBuffer" in name or kind == "m":
        prev = rows[i - 1]
        nxt = rows[i + 1] if i + 1 < n else None
        print(f"{name:50s} dur {(e - s) / 1e3:8.1f} us | gap before {(s - prev[1]) / 1e3:7.1f} us after {prev[2][-40:]:40s} | next {nxt[2][-40:] if nxt else ''}")
PY
find $out -name "*.csv" -delete
