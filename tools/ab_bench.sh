for rep in 1 2; do
  for wgs in 256 128 64 32; do
    HYDRA_MI_TUNE="chol_flow_wgs=$wgs" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('20 steps chol_flow_wgs=$wgs rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  done
done
for wgs in 256 128 64; do
HYDRA_MI_TUNE="chol_flow_wgs=$wgs" python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('64 steps chol_flow_wgs=$wgs', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
done
