cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c116-125; }
for i in 1 2 3; do echo "prev $(SDUMC_LIB=$GRAFT_REPO_ROOT/gpurun_ab_prev.so run)"; echo "new  $(run)"; done
