#!/bin/bash
# MFMA utilisation of the contraction kernels of the step (this repo's igemm / wgrad / attention; any library GEMM left):
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
#   bash tools/pmc_mfma.sh gpurun_out/mfmaN
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "wgrad|igemm|Cijk|xdl|attn|bwd_kernel|adamw" --output-format csv -d $out/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-graph > $out/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY > $out/mfma_util.json
import csv, glob, json, collections, re
rows = []
for f in glob.glob("$out/pmc/*/*counter_collection.csv"):
    rows += list(csv.DictReader(open(f)))
# only the LAST step (MIOpen's find-mode trials run in the first steps): everything dispatched after the
# previous step's last wgrad launch
# the LAST step only: everything dispatched after the second-to-last adamw launch (one per step)
opt = sorted({int(r["Dispatch_Id"]) for r in rows if "adamw" in r["Kernel_Name"]})
start = opt[-2] if len(opt) > 1 else 0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    if int(r["Dispatch_Id"]) <= start:
        continue
    n = r["Kernel_Name"]
    if "wgrad_group_kernel" in n: k = "dfine::wgrad_group_kernel (grouped weight gradients)"
    elif "wgrad_kernel" in n: k = "dfine::wgrad_kernel (single-problem weight gradients: stem)"
    elif "dfine::igemm2_kernel" in n: k = "dfine::igemm2_kernel, plain rows on the LDS-DMA ring (1x1 conv + Linear forward / input gradient)"
    elif "dfine::igemm_kernel" in n and "true>" in n.split("(")[0]: k = "dfine::igemm_kernel, gathered (k x k / strided conv forward + input gradient)"
    elif "dfine::igemm_kernel" in n: k = "dfine::igemm_kernel, plain rows that are not on the 16-byte path (odd K / pitches)"
    elif "dfine::attn" in n: k = "dfine::attn_* (self-attention forward / backward)"
    elif "igemm_fwd" in n or "conv_fwd" in n: k = "MIOpen conv forward (igemm asm / CK xdl)"
    elif "igemm_bwd" in n or "conv_bwd_data" in n: k = "MIOpen conv input-gradient (igemm asm / CK xdl)"
    elif "igemm_wrw" in n or "conv_bwd_weight" in n: k = "MIOpen conv weight-gradient (stem convs with 3 / 12 input channels)"
    elif n.startswith("Cijk"): k = "hipBLASLt GEMM (Linear forward / input gradient)"
    elif "dfine::" in n: k = "other dfine:: kernels the name filter lets through (msda / norm / pool backward, optimizer: no MFMA)"
    elif "attn" in n or "bwd_kernel" in n: k = "aotriton attention"
    else: k = "other xdl"
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
res = {"_note": "last step only; util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs), time-weighted over the class"}
for k, v in agg.items():
    busy, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    res[k] = {"dispatches": cnt[k], "mfma_busy_cycles": busy, "gui_active_cycles_sum_over_8_xcds": gui,
              "mfma_util": round(busy / (128.0 * gui), 4) if gui else None}
print(json.dumps(res, indent=1))
PY
cat $out/mfma_util.json
