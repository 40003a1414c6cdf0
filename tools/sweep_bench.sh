mkdir -p gpurun_out/sweep
for args in "--flow-batch 8" "--flow-batch 12" "--flow-batch 16"; do
  tag=$(echo $args | tr -d ' -')
  for steps in "--steps 20 --warmup 5" ""; do
    python bench.py $steps --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$args | $steps |', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  done
done
