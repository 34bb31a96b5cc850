cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3p_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3p_pytest.log
tail -4 gpurun_out/r3p_pytest.log
timeout -k 10 150 python scripts/gpu_fuzz.py 100 99 --big > gpurun_out/r3p_fuzzbig.log 2>&1; tail -2 gpurun_out/r3p_fuzzbig.log
timeout -k 10 120 python scripts/gpu_fuzz.py 80 98 > gpurun_out/r3p_fuzz.log 2>&1; tail -2 gpurun_out/r3p_fuzz.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3p_bench.log 2>&1; tail -c 2600 gpurun_out/r3p_bench.log
