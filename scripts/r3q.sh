cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "D4G_CLUSTER=1" "D4G_EXEC=auto"; do
env $cfg timeout -k 10 400 python - >> gpurun_out/r3q_repro.log 2>&1 <<PY
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
cat gpurun_out/r3q_repro.log
