"""profiles/r03_sor_pmc.json from the two PMC passes of tools/brox_pmc.py (tools/run_r3_profiles.sh):

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -o b -- python tools/brox_pmc.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -o b -- python tools/brox_pmc.py
  python tools/sor_pmc_json.py <dir>/fetch/b_counter_collection.csv <dir>/write/b_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is doubled on
gfx950 as that section prescribes -- checked here on k_add_out, whose traffic is known exactly."""
import csv, json, sys, collections, hashlib, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_revision():
    h = hashlib.sha256()
    for fn in ("brox_kernels.h", "brox.hip"):
        with open(os.path.join(ROOT, "kalman-hydra_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0))
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    sor = [k for k in fetch if k.startswith("void k_sor")][0]
    f = [v for v, _ in fetch[sor]]
    w = [v for v, _ in write[sor]]
    add = [k for k in fetch if k.startswith("k_add_out(")][0]
    add_f = max(v for v, _ in fetch[add])
    add_w = max(v for v, _ in write[add])
    n_px = 8 * 1024 * 1024
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/brox_pmc.py: 2 calls of "
                  "hm_brox_calc_dev, 8 pairs of 1024x1024, sor_threads 512, sor_fuse auto",
        "kernel_revision": kernel_revision(),
        "kernel": sor.split("(")[0],
        "launches": len(f),
        "fetch_size_kb_per_launch_raw": sum(f) / len(f),
        "write_size_kb_per_launch": sum(w) / len(w),
        "fetch_correction": 2.0,
        "calibration": "k_add_out at level 0 (8 x 1024^2 px, 16 B/px read, 8 B/px written): FETCH_SIZE %.0f KB against %d KB "
                       "read, WRITE_SIZE %.0f KB against %d KB written" % (add_f, n_px * 16 // 1024, add_w, n_px * 8 // 1024),
        "traffic_bytes_per_launch": (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0,
        "finest_level": {
            "pixels_per_launch": n_px, "iterations_per_launch": 5,
            "traffic_bytes_per_launch": (2.0 * max(f) + max(w)) * 1024.0,
            "algorithmic_bytes_per_launch": 52.0 * n_px * 5,
        },
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
