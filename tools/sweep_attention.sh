DFINE_ATTN_RESIDENT=1 timeout -k 10 240 python -X faulthandler -m pytest tests/test_attention_gpu.py -q -m gpu 2>&1 | tail -2
DFINE_ATTN_RESIDENT=0 timeout -k 10 240 python -X faulthandler -m pytest tests/test_attention_gpu.py -q -m gpu 2>&1 | tail -2
for cfg in "DFINE_ATTN_RESIDENT=1" "DFINE_ATTN_RESIDENT=0" "X=1"; do
env $cfg python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
python - "$cfg" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n: (v["ms_per_step"], v["avg_us"]) for n, v in k.items() if n.startswith("attention")})
PY
done
