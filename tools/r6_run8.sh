#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
{
for c in 0 10 20 40 0 10 20 40; do echo "SDUMC_GG_CHUNK=$c"; SDUMC_GG_CHUNK=$c python3 tools/gg_bench.py 30 frame,audio,key,all new; done
} > $O/gg_chunk.txt 2>&1
grep -v amdgpu.ids $O/gg_chunk.txt
