cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -3
for i in 1 2; do
SDUMC_FORCE_DP=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c100-200
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c100-200
done
SDUMC_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline 2>/dev/null | tail -1 | cut -c100-200
