#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
L=$R/sdumc_amd/csrc
{
echo "split3 full"; python3 tools/gg_bench.py 30 frame,audio,key new
for v in 1 2 4 3 6; do echo "split3 ablation bits $v (1 = no DMA, 2 = no convert, 4 = no MFMA)"; SDUMC_LIB=$L/libsdumc_hip_ggdbg$v.so python3 tools/gg_bench.py 30 frame,audio,key new; done
} > $O/gg3_ablate.txt 2>&1
grep -v amdgpu.ids $O/gg3_ablate.txt
