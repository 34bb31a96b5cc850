cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python scripts/fused_profile.py 64 > gpurun_out/r3f_prof.log 2>&1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3f_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3f_pytest.log
tail -4 gpurun_out/r3f_pytest.log; cat gpurun_out/r3f_prof.log
