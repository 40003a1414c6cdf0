"""Per-kernel, per-launch averages of the SQ counters collected by tools/run_ekf_pmc.sh
(three rocprofv3 --pmc passes over tools/ekf_pmc.py) -> CSV on stdout."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for path in sorted(glob.glob(os.path.join(root, "epmc_*", "b_counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + [n + "_per_launch" for n in names])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    n = max(len(v) for v in calls[k].values())
    w.writerow([k, n] + [round(acc[k][c] / max(1, len(calls[k][c]))) if c in acc[k] else "" for c in names])
