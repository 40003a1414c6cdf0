"""Development aid: the idle time in front of k_solve_prep (and every other kernel of the chain) by the iteration's place
in its frame, from a rocprofv3 kernel trace of the bench -- python tools/prep_gap.py b_kernel_trace.csv"""
import csv, sys, collections
CHAIN = ("k_measure_vertex", "k_measure_edge", "k_solve_prep", "k_chol_flow", "k_tvec", "k_render_iter", "k_iter_result")
def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))
              if short(r["Kernel_Name"]) in CHAIN)
its, cur = [], []
for r in rows:
    if r[2] == "k_measure_vertex" and cur:
        its.append(cur); cur = []
    cur.append(r)
its.append(cur)
# place of an iteration in its frame: a new frame starts when the previous iteration ended more than 150 us before
place, prev_end = [], None
for it in its:
    new = prev_end is None or it[0][0] - prev_end > 150000
    place.append(0 if new else place[-1] + 1)
    prev_end = max(e for _, e, _ in it)
by = collections.defaultdict(lambda: collections.defaultdict(list))
for it, pl in zip(its, place):
    pe = None
    for s, e, n in it:
        if pe is not None:
            by[min(pl, 3)][n].append((s - pe) / 1e3)
        pe = max(pe or e, e)
for pl in sorted(by):
    print("iteration %s of a frame:" % (pl if pl < 3 else "3+"))
    for n in CHAIN[1:]:
        g = sorted(by[pl][n])
        if g:
            print("   gap before %-16s n %4d  median %6.1f  mean %6.1f  max %7.1f us" % (n, len(g), g[len(g) // 2], sum(g) / len(g), g[-1]))
