cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-merge-on --no-cpu-baseline > gpurun_out/r3z_bench.log 2>&1; grep -o '"value": [0-9.]*\|"single_batch_latency_ms": [0-9.]*\|"ms_parse": [0-9.]*\|"ms_optimise": [0-9.]*' gpurun_out/r3z_bench.log | tr '\n' ' '; echo
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3z_pytest.log 2>&1; tail -2 gpurun_out/r3z_pytest.log
timeout -k 10 300 python scripts/gpu_fuzz.py 100 5150 --big --diff > gpurun_out/r3z_diff.log 2>&1; tail -1 gpurun_out/r3z_diff.log
