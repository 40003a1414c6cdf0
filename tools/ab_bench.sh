for rep in 1 2 3 4 5 6; do
  HYDRA_MI_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/rep$rep.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rep$rep', round(d['value'],1), round(d['steady_state']['value'],1), {k:round(v,3) for k,v in d['breakdown_ms_per_step'].items()})
"
  grep -o "([0-9]* iterations)" gpurun_out/rep$rep.err | tr -d '()a-z ' | tr '\n' ' '; echo
done
