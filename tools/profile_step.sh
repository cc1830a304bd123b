#!/bin/bash
# rocprofv3 kernel trace of a short bench run + per-step breakdowns. Usage (on the GPU box, from the repo root):
#   bash tools/profile_step.sh gpurun_out/profN [bench args...]
set -e
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_breakdown.py $out/*/*kernel_trace.csv > $out/step_breakdown.txt
python tools/phase_breakdown.py $out/*/*kernel_trace.csv | tee $out/phases.txt
cp $out/*/*kernel_stats.csv $out/kernel_stats.csv
