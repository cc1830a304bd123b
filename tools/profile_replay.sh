#!/bin/bash
# rocprofv3 kernel trace of REPLAYED steps (hipGraph): wall vs kernel-busy of the last step = idle gaps between graph nodes
#   bash tools/profile_replay.sh gpurun_out/profN [bench args...]
set -e
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 6 --no-cpu-baseline "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_breakdown.py $out/*/*kernel_trace.csv > $out/step_breakdown.txt
python tools/gap_histogram.py $out/*/*kernel_trace.csv 5 > $out/gaps.txt   # 5th step from the end: a replayed one
python tools/gap_histogram.py $out/*/*kernel_trace.csv 1 > $out/gaps_eager.txt
cp $out/*/*kernel_stats.csv $out/kernel_stats.csv
