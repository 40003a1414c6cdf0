"""Achieved algorithmic GB/s of the bandwidth-bound kernels, from a rocprofv3 kernel trace of bench.py:

  python tools/kernel_gbps.py <b_kernel_trace.csv> > profiles/r01_bench_kernel_gbps.csv

For the flow kernels the finest-level launches of the 8-pair series are taken (the largest grid of each
kernel: 8 x 1024^2 pixels), bytes per pixel as in DESIGN.md section 4.
k_measure_edge: 56 B per pixel of the intersections of adjacent star boxes (the pool entries of both vertices)."""
import csv, sys, collections

PX = 8 * 1024 * 1024
FLOW = {                       # kernel prefix -> (algorithmic bytes per pixel per launch, note)
    "void k_sor": (52.0 * 5, "52 B/px/iteration x 5 fused iterations"),
    "k_prepare": (76.0, "u,v,du,dv + 8 warped fields read, 7 written"),
    "void k_warp": (76.0, "11 read + 8 written"),
    "k_deriv": (24.0, "two images per launch, each 1 read + 2 written"),
    "k_add_out": (24.0, "4 read + 2 written"),
}


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    acc = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        for pre in FLOW:
            if name.startswith(pre) and int(r["Grid_Size_Z"]) == (16 if pre == "k_deriv" else 8):   # k_deriv: blockIdx.z = 2 x pairs
                size = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])
                acc[pre].append((size, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for pre in list(acc):                                  # finest level = the largest grid
        top = max(sz for sz, _ in acc[pre])
        acc[pre] = [d for sz, d in acc[pre] if sz == top]
    print('"kernel","launches","average_ns","algorithmic_bytes_per_launch","achieved_GBps","fraction_of_8000_GBps","bytes_per_pixel"')
    for pre, (bpp, note) in FLOW.items():
        v = acc.get(pre)
        if not v:
            continue
        avg = sum(v) / len(v)
        gb = bpp * PX / avg
        print('"%s",%d,%.0f,%.0f,%.0f,%.3f,"%s"' % (pre.replace("void ", ""), len(v), avg, bpp * PX, gb, gb / 8000.0, note))


if __name__ == "__main__":
    main()
