#!/bin/bash
# Round-3 profile set in one GPU-box visit; raw traces are summarised and deleted (gpurun_out/ merges back <= 64 MiB).
#   bash tools/collect_r3.sh
set -x
bash tools/profile_step.sh gpurun_out/prof_final > /dev/null 2>&1
bash tools/profile_replay.sh gpurun_out/prof_final_replay > /dev/null 2>&1
bash tools/profile_step.sh gpurun_out/prof_final_seg --task segment > /dev/null 2>&1
find gpurun_out/prof_final gpurun_out/prof_final_replay gpurun_out/prof_final_seg -name "*kernel_trace.csv" -delete
bash tools/pmc_traffic.sh gpurun_out/pmc_final > /dev/null 2>&1
find gpurun_out/pmc_final -name "*counter_collection.csv" -delete; find gpurun_out/pmc_final -name "*kernel_trace.csv" -delete
bash tools/pmc_mfma.sh gpurun_out/mfma_final > /dev/null 2>&1
find gpurun_out/mfma_final -name "*.csv" -delete
bash tools/pmc_stalls.sh gpurun_out/stalls_final > /dev/null 2>&1
find gpurun_out/stalls_final -name "*.csv" -delete
du -sh gpurun_out
ls gpurun_out/prof_final gpurun_out/prof_final_replay gpurun_out/prof_final_seg gpurun_out/pmc_final gpurun_out/mfma_final gpurun_out/stalls_final
