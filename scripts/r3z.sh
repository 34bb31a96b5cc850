cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python scripts/config5_diag.py 24 2>&1 | tail -1
timeout -k 10 400 python scripts/gpu_fuzz.py 200 4711 > gpurun_out/r3z_fuzz.log 2>&1; tail -1 gpurun_out/r3z_fuzz.log
timeout -k 10 300 python scripts/gpu_fuzz.py 150 4712 --diff > gpurun_out/r3z_diff.log 2>&1; tail -1 gpurun_out/r3z_diff.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3z_pytest.log 2>&1; tail -1 gpurun_out/r3z_pytest.log
