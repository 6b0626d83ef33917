#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_gemm_group.py tests/test_gpu_split.py -m gpu -x -q > $O/t.txt 2>&1; tail -5 $O/t.txt
{
for v in 1 0 1 0; do echo "SDUMC_GG3=$v"; SDUMC_GG3=$v python3 tools/gg_bench.py 30 frame,audio,key,keyca,all new; done
} > $O/gg3.txt 2>&1
grep -v amdgpu.ids $O/gg3.txt
for i in 1 2; do for v in 1 0; do echo -n "step GG3=$v: "; SDUMC_GG3=$v python3 bench.py --steps 100 --warmup 10 --no-side --no-cpu-baseline --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; done; done
