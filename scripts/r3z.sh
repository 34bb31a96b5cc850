cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-merge-on --no-cpu-baseline > gpurun_out/r3z_bench.log 2>&1; grep -o '"value": [0-9.]*\|"single_batch_latency_ms": [0-9.]*\|"ms_parse": [0-9.]*\|"ms_optimise": [0-9.]*' gpurun_out/r3z_bench.log | tr '\n' ' '; echo
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-merge-on --no-cpu-baseline --concurrency 4 > gpurun_out/r3z_bench4.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/r3z_bench4.log | tr '\n' ' '; echo
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "parity or boundary or recompress or lz or encode" > gpurun_out/r3z_pytest.log 2>&1; tail -2 gpurun_out/r3z_pytest.log
