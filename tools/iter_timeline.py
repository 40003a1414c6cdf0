"""Where an IEKF iteration goes: from a rocprofv3 kernel trace (b_kernel_trace.csv) of a bench / frame run, the
kernels of the filter's chain between two k_measure_vertex launches -- average duration of each and the average gap
between the end of its predecessor in the chain and its start.

    python tools/iter_timeline.py gpurun_out/benchprof/b_kernel_trace.csv
"""
import collections
import csv
import sys

CHAIN = ("k_measure_vertex", "k_measure_edge", "k_hth_scatter", "k_assemble", "k_assemble_flow", "k_solve_prep", "k_chol_flow",
         "k_tvec", "k_setup_all", "k_render", "k_render_iter", "k_error", "k_iter_result", "k_star_regions")


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if n in CHAIN:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2] == "k_measure_vertex"]
    dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    total, n_it, react, n_react = 0.0, 0, 0.0, 0
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        it = rows[a:b]
        ends = [i for i, r in enumerate(it) if r[2] == "k_iter_result"]
        if not ends:
            continue
        it = it[:ends[0] + 1]                  # one iteration: k_measure_vertex .. k_iter_result
        total += (it[-1][1] - it[0][0]) / 1e3
        n_it += 1
        if b < len(rows) and rows[b][0] - it[-1][1] < 60000:      # the next iteration of the same frame follows
            react += (rows[b][0] - it[-1][1]) / 1e3
            n_react += 1
        prev_end = None
        for s, e, n in it:
            dur[n] += (e - s) / 1e3
            if prev_end is not None:
                gap[n] += (s - prev_end) / 1e3
            cnt[n] += 1
            prev_end = max(prev_end or e, e)
    print("%d iterations, %.1f us each from the start of k_measure_vertex to the end of k_iter_result; %.1f us from there to "
          "the next iteration's k_measure_vertex (host reaction, %d cases)" % (n_it, total / max(1, n_it), react / max(1, n_react), n_react))
    print("%-20s %8s %10s %10s" % ("kernel", "per it.", "avg us", "gap before"))
    order = sorted(cnt, key=lambda n: CHAIN.index(n))
    for n in order:
        print("%-20s %8.2f %10.1f %10.1f" % (n, cnt[n] / n_it, dur[n] / cnt[n], gap[n] / cnt[n]))
    print("sum of kernel time per iteration %.1f us, of gaps %.1f us" % (sum(dur.values()) / n_it, sum(gap.values()) / n_it))


if __name__ == "__main__":
    main(sys.argv[1])
