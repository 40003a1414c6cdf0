# the 20-frame bench several times with the scheduling trace: the series sizes it chose and what the filter waited for
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ramp; out=gpurun_out/ramp/out.txt; : > $out
for i in 1 2 3 4 5 6 7 8; do
  HYDRA_MI_BENCH_TRACE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> gpurun_out/ramp/run$i.err > gpurun_out/ramp/run$i.json || exit 1
  python - $i >> $out <<'PY'
import json, re, sys
i = sys.argv[1]
d = json.loads(open("gpurun_out/ramp/run%s.json" % i).read().strip().splitlines()[-1])
sizes, waits, seen = [], [], set()
for l in open("gpurun_out/ramp/run%s.err" % i):
    m = re.match(r"step (\d+): flow wait ([\d.]+) ms, filter ([\d.]+) ms \((\d+) iterations\), series ready \((\d+), (\d+)\) in flight (.*)", l)
    if not m or int(m[1]) < 5:
        continue
    waits.append(float(m[2]))
    for a, b in [(int(m[5]), int(m[6]))] + [(int(x), int(y)) for x, y in re.findall(r"\((\d+), (\d+)\)", m[7])]:
        if (a, b) not in seen:
            seen.add((a, b)); sizes.append((a, b - a))
sizes.sort()
print("run %s: %.1f fps, waits %.1f ms (largest %s), series %s" % (i, d["value"], sum(waits), sorted(["%.1f" % w for w in waits if w > 0.5], key=float, reverse=True)[:4], [n for _, n in sizes]))
PY
done
cat $out
