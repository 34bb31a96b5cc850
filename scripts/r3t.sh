cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r3t/kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --concurrency 1 --no-merge-on > gpurun_out/r3t_kt.log 2>&1
f=$(find gpurun_out/r3t/kt -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r3t_kernel_stats.csv; head -30 "$f" | cut -c1-160
