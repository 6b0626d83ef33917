#!/bin/bash
# serial-lane kernel stats of bench.py; usage: bash tools/kstats.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 10 --warmup 3 --serial-lanes --no-cpu-baseline --no-roofline "$@" > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_serial_lanes.csv
rm -rf $O/kt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_bench_serial_lanes.csv")))
tot = 0
for r in rows:
    n = r["Name"]
    calls = int(r["Calls"]); t = float(r["TotalDurationNs"])
    per_step = t / 13 / 1e3
    tot += per_step
    if per_step > 3: print("%8.1f us/step  %5.1f calls/step  avg %7.1f us  %s" % (per_step, calls / 13, float(r["AverageNs"]) / 1e3, n[:110]))
print("total kernel time per step: %.1f us" % tot)
PY
