for rep in 1 2 3; do
  for extra in "" "--split-start"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('20 steps [$extra] rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  done
done
