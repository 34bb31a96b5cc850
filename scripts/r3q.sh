cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "D4G_CLUSTER=1" "D4G_EXEC=auto"; do
env $cfg timeout -k 10 120 python - <<PY
import sys, pickle, os
sys.path.insert(0,'tests')
import deft4j_amd as D, oracle_lib as O
D.init(0)
ins=pickle.load(open("scripts/repro_r3_fuzz99.pkl","rb"))
try:
    b=D.Batch(ins).run(True); st=b.stats()
    ok = all(b.output(i) == O.optimise(a, True)[1] for i, a in enumerate(ins) if b.result(i)["status"] == 0)
    print("$cfg", [len(x) for x in ins], "ok; == oracle", ok, "; cluster", st["rounds_cluster"], "fused", st["rounds_fused"], flush=True)
except Exception as e:
    print("$cfg", "FAILED", e, flush=True)
PY
done
timeout -k 10 200 python scripts/gpu_fuzz.py 150 99 --big > gpurun_out/r3q_fuzzbig.log 2>&1; tail -2 gpurun_out/r3q_fuzzbig.log
timeout -k 10 150 python scripts/gpu_fuzz.py 100 98 > gpurun_out/r3q_fuzz.log 2>&1; tail -2 gpurun_out/r3q_fuzz.log
