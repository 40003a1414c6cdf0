"""k_sor launches of a rocprofv3 kernel trace grouped by the number of frame pairs per launch (gridDim.z)
-> CSV on stdout (profiles/rNN_bench_sor_by_series.csv)."""
import collections
import csv
import sys

groups = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_sor" not in r["Kernel_Name"]:
        continue
    wz = int(r.get("Workgroup_Size_Z") or r.get("Workgroup_Size_z") or 1)
    gz = int(r.get("Grid_Size_Z") or r.get("Grid_Size_z") or 1)
    groups[max(1, gz // max(1, wz))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["pairs_per_launch", "launches", "total_ns", "average_ns", "min_ns", "max_ns"])
for k in sorted(groups):
    d = groups[k]
    w.writerow([k, len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d)])
