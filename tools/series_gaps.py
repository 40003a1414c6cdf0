"""Kernel trace (rocprofv3 --kernel-trace csv) of flow series -> per kernel: launches, mean duration, mean idle gap in
front of the launch (start - end of the previous kernel on the device), and the totals."""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tot = collections.defaultdict(lambda: [0, 0, 0])
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = 0 if prev_end is None else max(0, s - prev_end)
    if g > 200000:
        g = 0                      # between series / host pauses
    t = tot[n]
    t[0] += 1; t[1] += e - s; t[2] += g
    prev_end = max(e, prev_end or 0)
D = sum(v[1] for v in tot.values()); G = sum(v[2] for v in tot.values())
print("%-28s %8s %10s %10s %10s" % ("kernel", "launches", "avg_us", "gap_us", "total_ms"))
for n, v in sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-28s %8d %10.2f %10.2f %10.3f" % (n[:28], v[0], v[1] / v[0] / 1e3, v[2] / v[0] / 1e3, (v[1] + v[2]) / 1e6))
print("busy %.3f ms, gaps %.3f ms, launches %d" % (D / 1e6, G / 1e6, sum(v[0] for v in tot.values())))
