cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v; hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVAR_OFFER=$1 -DVAR_POLL=$2 -o /tmp/m_exp scripts/micro/exp/tree_micro_exp.hip 2>/dev/null && timeout -k 5 60 /tmp/m_exp | tail -2; done > gpurun_out/r3u_micro.log 2>&1; cat gpurun_out/r3u_micro.log
