cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python scripts/zopfli_timing.py rep1m png1m 2>&1 | grep -E "^rep1m|^png1m|squeeze" | tail -6
timeout -k 10 600 python -m pytest tests/test_gpu_zopfli.py -x -q -m gpu > gpurun_out/r3z_pytest.log 2>&1; tail -2 gpurun_out/r3z_pytest.log
timeout -k 10 200 python scripts/gpu_fuzz_zopfli.py 100 78 > gpurun_out/r3z_fuzz_zf.log 2>&1; tail -1 gpurun_out/r3z_fuzz_zf.log
