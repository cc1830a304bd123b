#!/bin/bash
# PMC passes (SQ, then TCC) over tools/wgrad_case.py; usage: bash tools/pmc_wgrad.sh <outdir under gpurun_out> <case args...>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/sq -- python3 $GRAFT_REPO_ROOT/tools/wgrad_case.py "$@" > $out/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/tcc -- python3 $GRAFT_REPO_ROOT/tools/wgrad_case.py "$@" > $out/tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $out/sq2 -- python3 $GRAFT_REPO_ROOT/tools/wgrad_case.py "$@" > $out/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for sub in ("sq", "tcc", "sq2"):
    for f in glob.glob("$out/" + sub + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "wgrad" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(sub, k, "last launch:", v[-1], "n =", len(v))
PY
