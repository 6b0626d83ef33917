cd $GRAFT_REPO_ROOT
for i in 1 2; do
for p in h n l; do
echo "bg prio $p: plain / background-lane"
SDUMC_BG_PRIORITY=$p timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c100-150
SDUMC_BG_PRIORITY=$p timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --background-lane 2>/dev/null | tail -1 | cut -c100-150
done; done
