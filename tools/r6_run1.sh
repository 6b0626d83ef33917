#!/bin/bash
# round 6, first GPU pass: the GPU suite on the new host / engine paths, then the default bench line (rotating resident batches + side legs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q -s > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt
tail -15 $O/gputests.txt
grep -E "gradient errors|worst relative" $O/gputests.txt
timeout 600 python3 bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 6000 $O/bench.json
tail -5 $O/bench.err
