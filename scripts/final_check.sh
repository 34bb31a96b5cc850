#!/bin/bash
# Dev tool, run ON THE GPU BOX (gpurun -- bash scripts/final_check.sh): the round's closing check — smoke, the GPU suite, the default bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1; tail -1 gpurun_out/final_pytest.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench.log 2>&1; grep -o '"value": [0-9.]*, "unit": "MB/s"\|"single_batch_latency_ms": [0-9.]*\|"output_bytes_identical_to_oracle": [a-z]*' gpurun_out/final_bench.log | tr '\n' ' '; echo
