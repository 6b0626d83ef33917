#!/bin/bash
# alternated multi-way: tools/ab_multi.sh reps "ENV1" "ENV2" ... -- bench args
N=$1; shift; envs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
for i in $(seq 1 $N); do for e in "${envs[@]}"; do
 r=$(env $e python bench.py --no-cpu-baseline --no-side --no-roofline --steps 100 "$@" 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
 echo "[$e] $r ms"; done; done
