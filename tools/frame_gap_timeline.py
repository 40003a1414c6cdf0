"""Between two frames of the filter: from a rocprofv3 kernel trace, every kernel that is not k_sor / flow work between
the last k_iter_result of a frame and the first k_measure_vertex of the next -- start relative to the end of that
k_iter_result, duration; one typical gap printed in full, and the average length of the gap.

    python tools/frame_gap_timeline.py gpurun_out/r3q_bench/b_kernel_trace.csv
"""
import csv
import sys

FLOW = ("k_sor", "k_prepare", "k_warp", "k_deriv", "k_blur", "k_resample", "k_pyr_down", "k_add_prolong", "k_add_out",
        "k_u8_to_f32", "k_coarse", "k_deriv_all")


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if n.startswith(FLOW):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    gaps = []
    i = 0
    while i < len(rows):
        if rows[i][2] == "k_iter_result":
            j = i + 1
            while j < len(rows) and rows[j][2] != "k_measure_vertex":
                j += 1
            if j < len(rows) and 150000 < rows[j][0] - rows[i][1] < 3000000:   # a frame boundary: neither the host's reaction nor a wait for flow
                gaps.append((i, j))
            i = j
        else:
            i += 1
    if not gaps:
        print("no frame boundaries found")
        return
    lens = [(rows[j][0] - rows[i][1]) / 1e3 for i, j in gaps]
    print("%d frame boundaries; end of the last k_iter_result -> start of the next frame's k_measure_vertex: mean %.1f us, "
          "min %.1f, max %.1f" % (len(gaps), sum(lens) / len(lens), min(lens), max(lens)))
    # the median-length gap in full
    order = sorted(range(len(gaps)), key=lambda k: lens[k])
    i, j = gaps[order[len(order) // 2]]
    t0 = rows[i][1]
    print("one of them (%.1f us):" % ((rows[j][0] - t0) / 1e3))
    print("  %-34s %10s %10s" % ("kernel", "start us", "dur us"))
    for s, e, n in rows[i + 1:j + 1]:
        print("  %-34s %10.1f %10.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
