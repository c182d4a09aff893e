# Knobs of the pipelined step against the default (round 5's kernels): value and ms per step of `bench.py` without its extras.
for cfg in "OCHIP_EXTRACT_STREAMS=5" "OCHIP_EXTRACT_STREAMS=3" "OCHIP_EXTRACT_CHUNK=64" "OCHIP_EXTRACT_CHUNK=128" "OCHIP_LINK_RUNNERS=4" "OCHIP_LINK_RUNNERS=2"; do
  v=$(env $cfg OCHIP_BENCH_EXTRAS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$cfg -> $v"
done
v=$(OCHIP_BENCH_EXTRAS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo "default -> $v"
