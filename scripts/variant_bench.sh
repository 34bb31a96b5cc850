#!/bin/bash
# dev tool (GPU box): build libdeft4g variants with extra -D flags and bench each through D4G_LIB
#   scripts/variant_bench.sh "-DD4G_TOK_ILP=1" "-DD4G_TOK_ILP=4"
mkdir -p gpurun_out
i=0
for flags in "" "$@"; do
  so=gpurun_out/libdeft4g_var$i.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $flags -shared -fPIC -o $so deft4j_amd/csrc/libdeft4g.hip 2>/dev/null || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"
  for r in 1 2; do
    D4G_LIB=$PWD/$so timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['phases_ms'])"
  done
  i=$((i+1))
done
