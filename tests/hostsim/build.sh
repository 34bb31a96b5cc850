#!/bin/bash
# Builds the TEST-ONLY CPU emulation of the HIP kernels (see hipsim.h).  Optional arg: "san" for ASan/UBSan.
# -ffp-contract=off: the Zopfli cost model's doubles must round exactly as written (as in the oracle and on the GPU).
set -e
cd "$(dirname "$0")"
FLAGS="-O2 -g -ffp-contract=off"
OUT=libdeft4g_hostsim.so
if [ "$1" = "waveheap" ]; then FLAGS="$FLAGS -DD4G_SIM_WAVE_HEAP"; OUT=libdeft4g_hostsim_wh.so; fi  # wave-wide tree builder in the emulator (slow)
if [ "$1" = "san" ]; then FLAGS="-O1 -g -ffp-contract=off -fsanitize=undefined -fno-sanitize-recover=undefined"; OUT=libdeft4g_hostsim_san.so; fi
# up to date?  (every source the emulation is made of)
if [ -f "$OUT" ]; then
    NEWER=$(find hipsim.cpp hipsim.h build.sh ../../deft4j_amd/csrc ../../include -newer "$OUT" -type f | head -1)
    if [ -z "$NEWER" ]; then exit 0; fi
fi
g++ -x c++ -std=c++17 $FLAGS -DD4G_HOSTSIM -include hipsim.h -fPIC -shared -o $OUT hipsim.cpp
