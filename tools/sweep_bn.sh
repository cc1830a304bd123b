#!/bin/bash
# sweep of the BN backward reduction grid size (env knob read by normact.hip); per-shape timings of the dominant maps
for n in ${@:-256 384 512}; do
  DFINE_BN_REDUCE_BLOCKS=$n python bench.py --steps 8 --warmup 4 --no-cpu-baseline --shapes 2>/dev/null | tail -1 > /tmp/b.json
  python - "$n" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
tot = sum(v["ms_per_step"] for n, v in k.items() if n.startswith("bn_bwd_reduce") or n.startswith("bn_stats"))
print("blocks", sys.argv[1], d["value"], d["ms_per_step"], "reduce+stats ms", round(tot, 3),
      {n.split()[-1]: v["avg_us"] for n, v in k.items() if n.startswith("bn_bwd_reduce") and n.split()[-1] in ("51200x128", "204800x128", "12800x256", "819200x32")})
PY
done
