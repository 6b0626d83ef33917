#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
{
python3 tools/map_bench.py 1024
python3 tools/epoch_probe.py
python3 tools/epoch_probe.py --fixed
python3 tools/epoch_probe.py --bf16
python3 tools/epoch_probe.py --bf16 --fixed
python3 tools/epoch_probe.py --bf16 --gather
} > $O/map_bench.txt 2>&1
grep -v amdgpu.ids $O/map_bench.txt
timeout 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "run_epoch or keep_bits or ragged" > $O/t1.txt 2>&1; tail -5 $O/t1.txt
