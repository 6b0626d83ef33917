cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline $1 2>/dev/null | tail -1 | cut -c116-125; }
for i in 1 2 3; do
echo "plain $(run)"
echo "bg fwd-only $(SDUMC_BG_MODE=2 run --background-lane)"
echo "bg at-chain $(SDUMC_BG_MODE=3 run --background-lane)"
done
