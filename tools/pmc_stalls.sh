#!/bin/bash
# Where the waves of the MFMA kernels spend their cycles (last eager step): SQ wave-cycle buckets and LDS-array occupancy.
#   WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC table)
#   LDS_IDX_ACTIVE = LDS-array cycles, LDS_BANK_CONFLICT = the extra ones among them
#   bash tools/pmc_stalls.sh gpurun_out/stallsN
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "wgrad|igemm|adamw" --output-format csv -d $out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-graph > $out/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY > $out/stalls.json
import csv, glob, json, collections
rows = []
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    rs = list(csv.DictReader(open(f)))
    opt = sorted({int(r["Dispatch_Id"]) for r in rs if "adamw" in r["Kernel_Name"]})
    start = opt[-2] if len(opt) > 1 else 0  # the LAST step: everything after the second-to-last optimizer launch
    rows += [r for r in rs if int(r["Dispatch_Id"]) > start]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    n = r["Kernel_Name"]
    if "wgrad_group_kernel" in n: k = "wgrad_group_kernel"
    elif "igemm2_kernel" in n: k = "igemm2_kernel (plain rows, LDS-DMA ring)"
    elif "igemm_kernel" in n and "true>" in n.split("(")[0]: k = "igemm_kernel gathered"
    elif "igemm_kernel" in n: k = "igemm_kernel plain rows"
    else: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"_note": "sums over the last eager step; SQ_* wave buckets in quad-cycles summed over waves; fractions of SQ_WAVE_CYCLES and of SQ_BUSY_CYCLES"}
for k, v in agg.items():
    wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    d = {c: v[c] for c in sorted(v)}
    d["frac_wait_any (s_waitcnt / barrier)"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3)
    d["frac_wait_inst_any (issue stalls)"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    d["frac_active_inst_any"] = round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    d["lds_conflict_share_of_lds_cycles"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / (v.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0), 3)
    res[k] = d
print(json.dumps(res, indent=1))
PY
cat $out/stalls.json
