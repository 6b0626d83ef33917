#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "run_epoch or keep_bits or ragged or c4_global" > $O/t1.txt 2>&1; tail -8 $O/t1.txt
{
python3 tools/epoch_probe.py
python3 tools/epoch_probe.py --gather
SDUMC_PF_MODE=2 python3 tools/epoch_probe.py --gather
python3 tools/epoch_probe.py --mode one
python3 tools/epoch_probe.py --fixed
python3 tools/epoch_probe.py --bf16
SDUMC_PF_MODE=2 python3 tools/epoch_probe.py --bf16
SDUMC_PF_MODE=0 python3 tools/epoch_probe.py --bf16
} > $O/epoch_probe.txt 2>&1
cat $O/epoch_probe.txt
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; tail -5 $O/gputests.txt
