"""Development aid: every kernel around one frame boundary of the bench (from the last k_iter_result of a frame to the second
k_tvec of the next), with its queue, start and end relative to that k_iter_result's end -- python tools/boundary_dump.py trace.csv [which]"""
import csv, sys
def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1])))
flt = [r for r in rows if not r[2].startswith(("k_sor", "k_prepare", "k_warp", "k_deriv", "k_blur", "k_resample", "k_pyr", "k_add", "k_coarse", "k_u8"))]
res = [i for i, r in enumerate(flt) if r[2] == "k_iter_result"]
# frame boundaries: a k_iter_result whose successor k_measure_vertex starts more than 100 us later
bounds = []
for i in res:
    nxt = next((r for r in flt[i + 1:] if r[2] == "k_measure_vertex"), None)
    if nxt and nxt[0] - flt[i][1] > 100000:
        bounds.append(i)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(bounds) // 2
i = bounds[which]
t0 = flt[i][1]
ntv = 0
for r in flt[i:]:
    print("%9.1f %9.1f  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2]))
    if r[2] == "k_tvec":
        ntv += 1
        if ntv == 2:
            break
