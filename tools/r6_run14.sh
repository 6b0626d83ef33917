#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
b() { python3 bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-side --no-roofline | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
git stash list >/dev/null 2>&1
for r in 1 2 3; do echo "split Adam: $(b)  bf16 $(b --bf16)   | one Adam (SDUMC_LIB=prev): $(SDUMC_LIB=$R/gpurun_ab_prev.so b)  bf16 $(SDUMC_LIB=$R/gpurun_ab_prev.so b --bf16)"; done
