#!/bin/bash
# the bench lines committed under profiles/ (run on the GPU box): default (= the driver's command) and the other configurations
set -x
mkdir -p gpurun_out/benches
python bench.py > gpurun_out/benches/default.json.log 2> gpurun_out/benches/default.err
python bench.py --task segment --no-cpu-baseline > gpurun_out/benches/segment.json.log 2> gpurun_out/benches/segment.err
python bench.py --model s --no-cpu-baseline > gpurun_out/benches/s640.json.log 2> gpurun_out/benches/s640.err
python bench.py --model x --img 960 --batch 16 --no-cpu-baseline > gpurun_out/benches/x960_b16.json.log 2> gpurun_out/benches/x960.err
python bench.py --batch 64 --no-cpu-baseline > gpurun_out/benches/batch64.json.log 2> gpurun_out/benches/batch64.err
python bench.py --model n --img 320 --no-cpu-baseline > gpurun_out/benches/n320.json.log 2> gpurun_out/benches/n320.err
for f in gpurun_out/benches/*.json.log; do python - "$f" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["hip_graph"], d["loss"], d["peak_device_memory_GB"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
done
