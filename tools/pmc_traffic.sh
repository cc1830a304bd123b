#!/bin/bash
# HBM-side traffic of every hand-written kernel: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass) over a short eager bench run, then tools/pmc_summarize.py -> profiles/<round>_pmc_traffic.json.
#   bash tools/pmc_traffic.sh gpurun_out/pmcN
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "dfine" --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-graph > $out/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $out > $out/pmc_traffic.json
