#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/sdumc_amd/csrc
for i in 1 2; do
echo "default"; python3 tools/gg_bench.py 30 frame,key,all new 2>&1 | grep -v amdgpu
echo "setprio 1 for waves 4-7"; SDUMC_LIB=$L/libsdumc_hip_ggprio.so python3 tools/gg_bench.py 30 frame,key,all new 2>&1 | grep -v amdgpu
done
b() { python3 bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-side --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
for r in 1 2 3; do echo "step default: $(b)   setprio: $(SDUMC_LIB=$L/libsdumc_hip_ggprio.so b)"; done
