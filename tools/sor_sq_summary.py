"""Per-level SQ / LDS counters of k_sor from the passes of tools/run_sor_sq.sh
(python tools/sor_sq_summary.py gpurun_out/sorsq_*/b_counter_collection.csv > profiles/r02_sor_sq_counters.csv).
With --prefix a,b,c as the first argument: the kernels whose names start with one of those instead (the filter's kernels:
profiles/r03_ekf_sq_counters.csv)."""
import collections
import csv
import sys

rows = collections.OrderedDict()          # (kernel, grid) -> counter -> [values]; durations
dur = collections.defaultdict(list)
prefixes, paths = ("k_sor",), sys.argv[1:]
if paths and paths[0] == "--prefix":
    prefixes, paths = tuple(paths[1].split(",")), paths[2:]
for path in paths:
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith(prefixes):
            continue
        key = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
        rows.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = (r["Dispatch_Id"], path)
        if d not in seen:
            seen.add(d)
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
counters = sorted({c for v in rows.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "wg_size", "vgpr", "lds_bytes", "launches", "avg_us_profiled"] + counters)
for key, v in sorted(rows.items(), key=lambda kv: -kv[0][1]):
    n = max(len(x) for x in v.values())
    w.writerow(list(key) + [n, "%.2f" % (sum(dur[key]) / len(dur[key]) / 1e3)] + ["%.0f" % (sum(v[c]) / len(v[c])) if c in v else "" for c in counters])
