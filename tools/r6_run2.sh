#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
{
python3 tools/epoch_probe.py --profile 2>&1 | head -60
for m in 1 0 2; do SDUMC_PF_MODE=$m python3 tools/epoch_probe.py; done
python3 tools/epoch_probe.py --mode one
python3 tools/epoch_probe.py --fixed
SDUMC_PF_MODE=0 python3 tools/epoch_probe.py --fixed
python3 tools/epoch_probe.py --wgs 256
python3 tools/epoch_probe.py --wgs 1024
python3 tools/epoch_probe.py --bf16
SDUMC_PF_MODE=0 python3 tools/epoch_probe.py --bf16
python3 tools/epoch_probe.py --bf16 --mode one
} > $O/epoch_probe.txt 2>&1
cat $O/epoch_probe.txt
