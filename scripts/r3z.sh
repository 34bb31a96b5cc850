cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python scripts/fused_profile.py 2>&1 | tail -3
