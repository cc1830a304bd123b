#!/bin/bash
# grouped weight-gradient planner: rows per chunk (DFINE_WGRAD_GROUP_ROWS_MAX / _MIN), whole-step effect. Run on the GPU box.
for cfg in "4096 1024" "2048 512" "2048 1024" "8192 1024" "8192 2048" "4096 2048" "16384 4096"; do
  set -- $cfg
  DFINE_WGRAD_GROUP_ROWS_MAX=$1 DFINE_WGRAD_GROUP_ROWS_MIN=$2 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/wg_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/wg_$1_$2.json").read().strip().splitlines()[-1]); print("$1 $2", d["value"], d["roofline"]["kernels"]["wgrad"]["ms_per_step"])
PY
done
