for q in 4 8 16; do for l in 2 3 4; do
echo "== GPU_MAX_HW_QUEUES=$q D4G_LANES=$l"
GPU_MAX_HW_QUEUES=$q D4G_LANES=$l timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['phases_ms'])"
done; done
