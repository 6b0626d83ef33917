#!/bin/bash
# PMC counters of the grouped GEMM kernel on one bench leg; usage: bash tools/gg_pmc.sh <tag> <leg>
TAG=${1:-ggpmc}; LEG=${2:-frame}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/p$i -- python3 tools/gg_bench.py 3 $LEG new > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "gg_" not in n: continue
    k = "tn" if "gg_tn" in n else "reduce"
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for c, v in acc[k].items():
        print(k, c, "avg %.4g" % (sum(v) / len(v)), "n", len(v))
PY
  rm -rf $O/p$i
done
