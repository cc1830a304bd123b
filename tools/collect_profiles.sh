set -x
bash tools/profile_step.sh gpurun_out/prof_final > /dev/null 2>&1
bash tools/profile_replay.sh gpurun_out/prof_final_replay > /dev/null 2>&1
bash tools/pmc_traffic.sh gpurun_out/pmc_final > /dev/null 2>&1
bash tools/pmc_mfma.sh gpurun_out/mfma_final > /dev/null 2>&1
ls gpurun_out/prof_final gpurun_out/prof_final_replay gpurun_out/pmc_final gpurun_out/mfma_final
