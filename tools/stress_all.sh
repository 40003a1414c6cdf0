for cfg in ""; do
  echo "== $cfg"
  env $cfg timeout -k 10 500 python tools/stress_determinism.py 400 2>&1 | grep -v amdgpu | grep -E "FAILED|iterations so far|repetitions differ|differs|frame [0-9]+: state|bad prediction|positions differ" | cut -c1-260 | head -30
done
