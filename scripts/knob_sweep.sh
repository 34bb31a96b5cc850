#!/bin/bash
# dev tool: bench the executor knobs one at a time (run on the GPU box)
for kv in "D4G_STATE_BLOCK=64" "D4G_STATE_BLOCK=128" "D4G_STATE_BLOCK=256" "D4G_LANES=1" "D4G_LANES=4" "D4G_EXEC=persistent"; do
  echo "== $kv"
  env $kv timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['phases_ms'], d['roofline']['kernel_ms_per_step'], d['roofline']['search_kernels_ms'])"
done
