for rep in 1 2; do
  for fb in 8 16; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --flow-batch $fb 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('20 steps B=$fb rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
    python bench.py --no-cpu-baseline --flow-batch $fb 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('64 steps B=$fb rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  done
done
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --flow-batch 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('200 steps B=8', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --flow-batch 16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('200 steps B=16', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
