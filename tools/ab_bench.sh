for rep in 1 2 3; do
  for t in "edge_split=2" "edge_split=1" "render_rows=8"; do
    HYDRA_MI_TUNE="$t" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('20 steps $t rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  done
done
for t in "edge_split=2" "edge_split=1" "render_rows=8"; do
HYDRA_MI_TUNE="$t" python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('64 steps $t', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
done
