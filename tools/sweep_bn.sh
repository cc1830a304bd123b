#!/bin/bash
# sweep of the BN grid sizes (env knobs read by normact.hip): DFINE_BN_APPLY_BLOCKS (apply / dx kernels), DFINE_BN_REDUCE_BLOCKS
knob=${1:-DFINE_BN_APPLY_BLOCKS}; shift
for n in ${@:-512 1024 2048 4096}; do
  env $knob=$n python bench.py --steps 8 --warmup 4 --no-cpu-baseline --shapes 2>/dev/null | tail -1 > /tmp/b.json
  python - "$knob=$n" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
tot = {f: round(sum(v["ms_per_step"] for n, v in k.items() if n.startswith(f)), 3) for f in ("bn_apply", "bn_bwd_dx", "bn_bwd_reduce")}
print(sys.argv[1], d["value"], d["ms_per_step"], tot,
      {n: v["avg_us"] for n, v in k.items() if n.split()[0] in ("bn_apply", "bn_bwd_dx") and n.split()[-1] in ("51200x128", "204800x128", "12800x256")})
PY
done
