#!/bin/bash
# quick sweep of the BN reduction grid size (env knob read by normact.hip)
for n in 256 512 1024 2048; do
  DFINE_BN_REDUCE_BLOCKS=$n python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$n" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
print("R=", sys.argv[1], d["value"], d["ms_per_step"], "stats", k["bn_stats"]["avg_us"], "bwd_reduce", k["bn_bwd_reduce"]["avg_us"])
PY
done
