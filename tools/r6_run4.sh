#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
{
for m in 2 0; do
SDUMC_PF_MODE=$m python3 tools/epoch_probe.py
SDUMC_PF_MODE=$m python3 tools/epoch_probe.py --fixed
done
SDUMC_PF_MODE=2 python3 tools/epoch_probe.py --fixed --gather
SDUMC_PF_MODE=2 python3 tools/epoch_probe.py --fixed --bf16
SDUMC_PF_MODE=0 python3 tools/epoch_probe.py --fixed --bf16
python3 bench.py --steps 100 --warmup 10 --no-side --no-cpu-baseline --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("static fp32", d["ms_per_step"], d["host_enqueue_ms_per_step"])'
python3 bench.py --bf16 --steps 100 --warmup 10 --no-side --no-cpu-baseline --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("static bf16", d["ms_per_step"], d["host_enqueue_ms_per_step"])'
} > $O/epoch_probe.txt 2>&1
grep -v amdgpu.ids $O/epoch_probe.txt
