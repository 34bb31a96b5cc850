cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/lat_probe scripts/micro/lat_probe.hip 2>/dev/null && timeout -k 5 60 /tmp/lat_probe > gpurun_out/r3v_lat.log 2>&1; cat gpurun_out/r3v_lat.log
