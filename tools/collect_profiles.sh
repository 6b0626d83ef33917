#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/<tag>/ (copied to profiles/ afterwards):
#   bench lines (fp32 default, bf16, epoch, c1, c5), serial-lane rocprofv3 kernel stats (fp32 + bf16), PMC traffic passes.
# usage: bash tools/collect_profiles.sh <tag>
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
python3 bench.py --bf16 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err
python3 bench.py --workload epoch --steps 200 --warmup 10 > $O/bench_epoch.json 2>> $O/bench.err
python3 bench.py --workload epoch --bf16 --steps 200 --warmup 10 > $O/bench_epoch_bf16.json 2>> $O/bench.err
python3 bench.py --workload c1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_c1.json 2>> $O/bench.err
python3 bench.py --workload c5 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
python3 bench.py --workload c5 --bf16 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c5_bf16.json 2>> $O/bench.err
SDUMC_FORCE_DP=1 timeout 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_one_rank_rccl_dp.json 2>> $O/bench.err
SDUMC_BENCH_TIMEOUT=90 SDUMC_DIST_BACKEND=gloo timeout 150 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline > $O/bench_gloo_2ranks_1gpu.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 10 --warmup 3 --prewarm-s 0 --serial-lanes --no-cpu-baseline --no-roofline > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_serial_lanes.csv
python3 tools/kstats_summary.py $O/kernel_stats_bench_serial_lanes.csv $O/rocprof_kernel_avg.json > $O/rocprof_kernel_avg.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktb -- python3 bench.py --bf16 --steps 10 --warmup 3 --prewarm-s 0 --serial-lanes --no-cpu-baseline --no-roofline > $O/ktb.log 2>&1
cp $(find $O/ktb -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_bf16_serial_lanes.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcf -- python3 bench.py --steps 3 --warmup 1 --prewarm-s 0 --serial-lanes --no-cpu-baseline --no-roofline > $O/pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcw -- python3 bench.py --steps 3 --warmup 1 --prewarm-s 0 --serial-lanes --no-cpu-baseline --no-roofline > $O/pmcw.log 2>&1
python3 tools/pmc_traffic_summary.py $(find $O/pmcf -name "*counter_collection.csv" | head -1) $(find $O/pmcw -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json 4 > $O/pmc_summary.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcfb -- python3 bench.py --bf16 --steps 3 --warmup 1 --prewarm-s 0 --serial-lanes --no-cpu-baseline --no-roofline > $O/pmcfb.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcwb -- python3 bench.py --bf16 --steps 3 --warmup 1 --prewarm-s 0 --serial-lanes --no-cpu-baseline --no-roofline > $O/pmcwb.log 2>&1
python3 tools/pmc_traffic_summary.py $(find $O/pmcfb -name "*counter_collection.csv" | head -1) $(find $O/pmcwb -name "*counter_collection.csv" | head -1) $O/pmc_traffic_bf16.json 4 > $O/pmc_summary_bf16.txt 2>&1
rm -rf $O/kt $O/ktb $O/pmcf $O/pmcw $O/pmcfb $O/pmcwb
for f in bench bench_bf16 bench_epoch bench_epoch_bf16 bench_c1 bench_c5 bench_c5_bf16 bench_one_rank_rccl_dp bench_gloo_2ranks_1gpu; do python3 - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("$f", d["value"], d["ms_per_step"], r.get("kernel"), r.get("achieved"), r.get("frac"), d.get("epoch_over_static"))
except Exception as e:
    print("$f FAILED", e)
PY
done
cat $O/pmc_summary.txt | head -12; python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print('fp32 per step', d.get('per_step'))
d=json.load(open('$O/pmc_traffic_bf16.json')); print('bf16 per step', d.get('per_step'))"
