#!/bin/bash
# One gpurun call per round end: the GPU suite, tools/collect_profiles.sh (bench lines, rocprofv3 stats, PMC passes) and the step marks; the
# outputs land under gpurun_out/ and are copied into profiles/ by hand (r6z_*, pmc_traffic*.json, rocprof_kernel_avg.json).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r6z_gputests.txt 2>&1; tail -3 gpurun_out/r6z_gputests.txt
bash tools/collect_profiles.sh r6z 2>&1 | tail -30
python3 tools/step_marks.py > gpurun_out/r6z/step_marks_fp32.txt 2>&1; python3 tools/step_marks.py --bf16 > gpurun_out/r6z/step_marks_bf16.txt 2>&1
