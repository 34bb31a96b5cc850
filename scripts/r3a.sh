set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
D4G_DEBUG_ROUNDS=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r3a_bench512.log 2>&1; tail -c 2500 gpurun_out/r3a_bench512.log
D4G_FUSED_BLOCK=256 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3a_bench256.log 2>&1; tail -c 1500 gpurun_out/r3a_bench256.log
D4G_EXEC=levels timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3a_benchlevels.log 2>&1; tail -c 1500 gpurun_out/r3a_benchlevels.log
