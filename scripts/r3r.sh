cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload config3 --members 1024 --steps 2 --warmup 1 > gpurun_out/r3r_config3.log 2>&1; tail -c 2200 gpurun_out/r3r_config3.log; echo
timeout -k 10 500 python bench.py --workload config4 --steps 1 --warmup 0 > gpurun_out/r3r_config4.log 2>&1; tail -c 2200 gpurun_out/r3r_config4.log
