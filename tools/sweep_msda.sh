timeout -k 10 400 python -X faulthandler -m pytest tests/test_msda.py tests/test_select_rows_gpu.py -q -m gpu 2>&1 | tail -4
for cfg in "DFINE_MSDA_GRAD=f32" "X=1"; do
env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
python - "$cfg" <<'PY'
import json, sys
d = json.load(open('/tmp/b.json')); k = d["roofline"]["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["loss"], {n: (v["ms_per_step"], v["avg_us"]) for n, v in k.items() if n.startswith("msda")})
PY
done
