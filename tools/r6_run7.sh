#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; tail -4 $O/gputests.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
for k,v in d["side"].items(): print(k, v["value"], v["ms_per_step"], v.get("epoch_over_static"), v.get("static_shape_at_mean_T"))
print({k:d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_us")})
PY
