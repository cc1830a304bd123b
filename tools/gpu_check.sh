#!/bin/bash
# One GPU-box visit: the GPU test suite, then (only if pytest ended by itself: exit code 0 or 1, not killed by its timeout)
# the default benchmark. Usage: tools/gpu_check.sh <tag> [pytest args...]
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${tag}_tests.log
tail -25 gpurun_out/${tag}_tests.log
if [ $rc -gt 1 ]; then echo "pytest did not end by itself (rc=$rc): no further GPU step"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
brc=$?
echo "bench rc=$brc"
python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
    print("BENCH", r["value"], r["unit"], r["ms_per_step"], "ms/step; roofline", r["roofline"]["kernel"], r["roofline"]["frac"])
    for k, v in sorted(r["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
        print(f"  {k:20s} {v['ms_per_step']:7.3f} ms  x{v['launches']//3:4d}  {v['avg_us']:8.1f} us")
except Exception as e:
    print("no bench line:", e)
PY
exit $(( rc > brc ? rc : brc ))
