#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
{
python3 tools/map_bench.py 1024
python3 tools/epoch_probe.py
python3 tools/epoch_probe.py --fixed
python3 tools/epoch_probe.py --bf16
} > $O/map_bench.txt 2>&1
grep -v amdgpu.ids $O/map_bench.txt
