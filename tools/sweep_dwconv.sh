timeout -k 10 200 python -m pytest tests/test_dwconv.py -q -m gpu 2>&1 | tail -2
for cfg in "DFINE_DWCONV_NO_STRIP=1" "X=1"; do
env $cfg python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
python - "$cfg" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n: (v["ms_per_step"], v["avg_us"]) for n, v in k.items() if n.startswith("dwconv")})
PY
done
