cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/bench_r1n.json 2> gpurun_out/bench_r1n.err; echo rc=$?; tail -1 gpurun_out/bench_r1n.json | cut -c1-250
timeout 300 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/bench_r1n_c5.json 2>/dev/null; tail -1 gpurun_out/bench_r1n_c5.json | cut -c100-250
timeout 300 python bench.py --workload c5 --bf16 --no-cpu-baseline > gpurun_out/bench_r1n_c5_bf16.json 2>/dev/null; tail -1 gpurun_out/bench_r1n_c5_bf16.json | cut -c100-250
timeout 300 python bench.py --workload c1 --no-cpu-baseline > gpurun_out/bench_r1n_c1.json 2>/dev/null; tail -1 gpurun_out/bench_r1n_c1.json | cut -c100-250
timeout 300 python bench.py --bf16 --no-cpu-baseline > gpurun_out/bench_r1n_bf16.json 2>/dev/null; tail -1 gpurun_out/bench_r1n_bf16.json | cut -c100-250
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1n -o p -- python3 bench.py --steps 10 --warmup 3 --serial-lanes --no-cpu-baseline --no-roofline > gpurun_out/prof_r1n.log 2>&1; echo rc=$?; ls gpurun_out/prof_r1n
