#!/bin/bash
# per-kernel times of tools/gg_bench.py (rocprofv3 kernel trace); usage: bash tools/gg_prof.sh <tag>
TAG=${1:-gg}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/gg_bench.py 10 > $O/bench.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
rm -rf $O/kt
cat $O/bench.log | tail -8
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_trace.csv")))
# per launch of the gg kernels, in order
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "gg_tn_kernel" in n or "gg_reduce" in n:
        out.append((("tn" if "gg_tn" in n else "reduce"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
# print the last launches of each bench leg (13 launches per leg: 3 warm + 10 timed)
import itertools
i = 0
leg = 0
while i < len(out):
    chunk = out[i:i + 26]
    tn = [c[1] for c in chunk if c[0] == "tn"][3:]
    rd = [c[1] for c in chunk if c[0] == "reduce"][3:]
    if tn:
        print("leg", leg, "tn avg %.1f us min %.1f" % (sum(tn) / len(tn), min(tn)), "reduce avg %.1f us" % (sum(rd) / max(1, len(rd))), "grid", chunk[0][2])
    i += 26
    leg += 1
PY
