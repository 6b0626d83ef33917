#!/bin/bash
# alternated A/B of the default bench line under two environments: tools/ab_step.sh "ENV_A" "ENV_B" [reps] [extra bench args]
A="$1"; B="$2"; N="${3:-3}"; shift 3 2>/dev/null
for i in $(seq 1 $N); do
  for e in "$A" "$B"; do
    r=$(env $e python bench.py --no-cpu-baseline --no-side --no-roofline --steps 100 "$@" 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
    echo "[$e] $r ms"
  done
done
