"""Per-kernel totals of one flow series from a rocprofv3 kernel trace of tools/brox_pmc.py (2 calls of 8 pairs)
-> CSV on stdout (profiles/rNN_flow_series_kernels.csv)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
calls = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    tot[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[n][1] += 1
S = sum(v[0] for v in tot.values())
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches_per_series", "ms_per_series", "average_us", "percent"])
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    w.writerow([n, v[1] / calls, round(v[0] / 1e6 / calls, 3), round(v[0] / v[1] / 1e3, 2), round(100.0 * v[0] / S, 1)])
w.writerow(["TOTAL", sum(v[1] for v in tot.values()) / calls, round(S / 1e6 / calls, 3), "", 100.0])
