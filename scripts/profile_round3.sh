#!/bin/bash
# Dev tool, run ON THE GPU BOX (gpurun -- scripts/profile_round3.sh <tag>): the rocprofv3 passes whose summaries are
# committed under profiles/<tag>_*.  Kernel trace and every counter set are separate runs (no --pmc with other traces).
# One batch at a time (--concurrency 1) so that a kernel's figures are its own.
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-merge-on --concurrency 1"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -- $B --steps 6 --warmup 2 > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- $B --steps 1 --warmup 0 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -f csv -d $OUT/sq -- $B --steps 1 --warmup 0 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $OUT/sq2 -- $B --steps 1 --warmup 0 > $OUT/sq2.log 2>&1
find $OUT -name "*.csv" | head -40
