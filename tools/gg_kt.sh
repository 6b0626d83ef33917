#!/bin/bash
# kernel-only durations of the gg kernels for one bench leg; usage: bash tools/gg_kt.sh <tag> <legs> [env assignments]
TAG=$1; LEGS=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/gg_bench.py 10 $LEGS new > $O/bench.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "gg_" in n:
        seq.append(("tn" if "gg_tn" in n else "rd", int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
seq.sort(key=lambda x: x[1])
# group by 13 (3 warm-up + 10) launch pairs per leg
pairs = [(seq[i], seq[i + 1]) for i in range(0, len(seq) - 1, 2)]
for leg in range(len(pairs) // 13):
    ps = pairs[leg * 13 + 3: leg * 13 + 13]
    tn = sum(p[0][2] - p[0][1] for p in ps) / len(ps) / 1e3
    rd = sum(p[1][2] - p[1][1] for p in ps) / len(ps) / 1e3
    gap = sum(p[1][1] - p[0][2] for p in ps) / len(ps) / 1e3
    print("leg %d: gemm %.1f us, gap %.1f us, reduce %.1f us" % (leg, tn, gap, rd))
PY
grep "TF" $O/bench.log
rm -rf $O/kt
