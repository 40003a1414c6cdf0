# A/B of environment knobs on the driver's 20-step bench, three runs each, interleaved
for rep in 1 2 3; do
  for cfg in "A HYDRA_MI_MODEL_RAMP=1" "B HYDRA_MI_MODEL_RAMP=0"; do
    set -- $cfg
    tag=$1; shift
    env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  done
done
