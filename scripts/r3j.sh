cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_containers.py tests/test_gpu_lz77.py -x -q -m gpu > gpurun_out/r3j_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3j_pytest.log
tail -15 gpurun_out/r3j_pytest.log
