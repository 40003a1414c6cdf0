"""Print calls / average / min / max (ns) of the kernels whose name contains one of the given substrings,
from a rocprofv3 --stats kernel_stats.csv."""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or any(k in r["Name"] for k in sys.argv[2:]):
        print("%-60s calls %6s avg %10.1f min %8s max %8s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
