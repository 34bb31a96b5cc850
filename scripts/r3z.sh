cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in 3 4 5 6 8 4 5 6; do echo -n "steps 20 warmup 5 concurrency $c: "; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-merge-on --no-cpu-baseline --concurrency $c 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
