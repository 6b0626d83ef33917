#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/sdumc_amd/csrc
b() { python3 bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-side --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
for r in 1 2 3 4 5; do echo "fp32 default: $(b)  prio1(w4-7): $(SDUMC_LIB=$L/libsdumc_hip_ggprio1.so b)  all@1: $(SDUMC_LIB=$L/libsdumc_hip_ggprio2.so b)  all@3: $(SDUMC_LIB=$L/libsdumc_hip_ggprio3.so b)"; done
