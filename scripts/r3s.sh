cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-merge-on > gpurun_out/r3s_bench.log 2>&1; tail -c 1800 gpurun_out/r3s_bench.log; echo
timeout -k 10 200 python scripts/fused_profile.py > gpurun_out/r3s_prof.log 2>&1; tail -40 gpurun_out/r3s_prof.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3s_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3s_pytest.log
tail -4 gpurun_out/r3s_pytest.log
