cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-merge-on 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"single_batch_latency_ms": [0-9.]*\|"ms_optimise": [0-9.]*' | tr '\n' ' '; echo; }
run D4G_FUSED_BLOCK=512
run D4G_FUSED_BLOCK=256
run D4G_FUSED_BLOCK=384
run D4G_FUSED_BLOCK=512
run D4G_FUSED_BLOCK=256
run D4G_FUSED_BLOCK=384
