set -e
timeout -k 10 300 python -m pytest tests/test_msda.py -q -m gpu -x -p no:cacheprovider -k "sorted" > gpurun_out/r3n_t.log 2>&1 || { tail -20 gpurun_out/r3n_t.log; exit 1; }
tail -1 gpurun_out/r3n_t.log
for w in 256 512 1024 2048; do
  DFINE_MSDA_SORT_WGS=$w timeout -k 10 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r3n_bench_$w.json 2>/dev/null
  python - <<PY
import json
r=json.loads(open("gpurun_out/r3n_bench_$w.json").read().strip().splitlines()[-1])
print("$w", r["value"], r["roofline"]["kernels"]["msda_bwd_value"]["avg_us"])
PY
done
