cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-merge-on --concurrency 1 2>&1 | grep -o '"ms_parse": [0-9.]*\|"ms_per_step": [0-9.]*\|"parse_kernels_ms": [0-9.]*\|"roundtrip_ok": [a-z]*' | tr '\n' ' '; echo; }
run D4G_JUMP_LDS=0
run D4G_JUMP_LDS=24
run D4G_JUMP_LDS=24 D4G_JUMP_TILE_REPS=4
run D4G_JUMP_LDS=24 D4G_JUMP_TILE_REPS=2
run D4G_JUMP_LDS=24 D4G_JUMP_TILE_REPS=3
D4G_DEBUG_JUMP=1 D4G_JUMP_LDS=24 D4G_JUMP_TILE_REPS=2 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-merge-on --concurrency 1 2>&1 | grep "jump rounds" | tail -3
D4G_DEBUG_JUMP=1 D4G_JUMP_LDS=0 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-merge-on --concurrency 1 2>&1 | grep "jump rounds" | tail -3
