cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/bench_r1o.json 2> gpurun_out/bench_r1o.err; echo rc=$?; tail -1 gpurun_out/bench_r1o.json | cut -c1-250
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1o -o p -- python3 bench.py --steps 10 --warmup 3 --serial-lanes --no-cpu-baseline --no-roofline > gpurun_out/prof_r1o.log 2>&1; echo rc=$?
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r1o_f -o pmc -- python3 bench.py --steps 3 --warmup 1 --serial-lanes --no-cpu-baseline --no-roofline > gpurun_out/pmc_r1o_f.log 2>&1; echo rc=$?
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r1o_w -o pmc -- python3 bench.py --steps 3 --warmup 1 --serial-lanes --no-cpu-baseline --no-roofline > gpurun_out/pmc_r1o_w.log 2>&1; echo rc=$?
SDUMC_FORCE_DP=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_r1o_force_dp.json 2>/dev/null; tail -1 gpurun_out/bench_r1o_force_dp.json | cut -c100-200
