#!/bin/bash
# alternated A/B of two bench.py argument sets: tools/ab_args.sh reps "ARGS_A" "ARGS_B"
N=$1; A="$2"; B="$3"
for i in $(seq 1 $N); do for a in "$A" "$B"; do
 r=$(python bench.py --no-cpu-baseline --no-side --no-roofline --steps 100 $a 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
 echo "[$a] $r ms"; done; done
