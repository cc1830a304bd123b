#!/bin/bash
# sweep of the BN backward reduction: grid size x pixel unroll (env knobs read by normact.hip)
for u in 2 4; do for n in 512 1024 2048; do
  DFINE_BN_BWD_UNROLL=$u DFINE_BN_REDUCE_BLOCKS=$n python bench.py --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$n" "$u" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
print("blocks", sys.argv[1], "unroll", sys.argv[2], d["value"], d["ms_per_step"], "stats", k["bn_stats"]["avg_us"], "bwd_reduce", k["bn_bwd_reduce"]["avg_us"], "wgrad ms", k["wgrad"]["ms_per_step"])
PY
done; done
