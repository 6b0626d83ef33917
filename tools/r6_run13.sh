#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
b() { python3 bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-side --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
{
for r in 1 2 3; do echo "default: $(b)   high-priority caller stream: $(b --stream-priority -1)   bf16: $(b --bf16) / $(b --bf16 --stream-priority -1)"; done
} > $O/ab_prio.txt 2>&1
cat $O/ab_prio.txt | grep -v amdgpu
