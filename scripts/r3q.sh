cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "D4G_CLUSTER=1" "D4G_EXEC=auto"; do
env $cfg timeout -k 10 100 python - <<PY
import sys, pickle, os
sys.path.insert(0,'tests')
import deft4j_amd as D
D.init(0)
ins=pickle.load(open("scripts/repro_r3_fuzz99.pkl","rb"))
b=D.Batch(ins).run(True); st=b.stats()
outs=[(b.result(i)["status"], b.output(i) if b.result(i)["status"]==0 else None, b.result(i)["saved_bits"]) for i in range(len(ins))]
pickle.dump(outs, open("gpurun_out/r3q_outs_%s.pkl" % "$cfg".replace("=","_").replace(" ","_"), "wb"))
print("$cfg ok; cluster", st["rounds_cluster"], "fused", st["rounds_fused"], [o[2] for o in outs], flush=True)
PY
done
