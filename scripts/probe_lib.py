"""Developer probe: time the top-k kernels of one 100k x 100k search under different env settings (run on the GPU box)."""
import os, subprocess, sys, csv, glob, shutil
def run(tag, env):
    d = f"gpurun_out/pp_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    args = env.pop("ARGS", "").split()
    e = dict(os.environ, TMPDIR="/tmp", **env)
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, "scripts/knn_probe.py"] + args, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(d + "/*/*kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "knn_topk" in r["Kernel_Name"] or "refine" in r["Kernel_Name"]]
    ts = [(r["Kernel_Name"][33:46], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2)) for r in rows[-3:]]
    print(tag, ts, flush=True)
if __name__ == "__main__":
    for spec in sys.argv[1:]:
        tag, *kv = spec.split(",")
        run(tag, dict(x.split("=", 1) for x in kv))
