cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for c in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --concurrency $c --no-cpu-baseline --no-merge-on > gpurun_out/r3g_conc$c.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3g_conc$c.log") if x.startswith("{")]
d=json.loads(l[-1]); print("conc $c", d["value"], d["ms_per_step"], d["phases_ms"])
PY
done
done
