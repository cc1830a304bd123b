for n in 128 256 512 1024; do
DFINE_LN_BWD_BLOCKS=$n python bench.py --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
python - "$n" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
print("ln bwd blocks", sys.argv[1], d["value"], d["ms_per_step"], k["add_layernorm_bwd"]["avg_us"], k["add_layernorm_bwd"]["ms_per_step"])
PY
done
