#!/bin/bash
# dev tool: bench the executor knobs (run on the GPU box); pass the settings to try as arguments, e.g.
#   scripts/knob_sweep.sh "D4G_STATE_BLOCK=256" "D4G_STATE_BLOCK=256 D4G_WIDE_BLOCK=512"
if [ $# -eq 0 ]; then set -- "D4G_STATE_BLOCK=64" "D4G_STATE_BLOCK=128" "D4G_STATE_BLOCK=256" "D4G_LANES=1" "D4G_LANES=4" "D4G_EXEC=persistent"; fi
for kv in "$@"; do
  echo "== $kv"
  env $kv timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['phases_ms'], d['roofline']['kernel_ms_per_step'], d['roofline']['search_kernels_ms'])"
done
