cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D4G_DEBUG_ROUNDS=1 timeout -k 10 240 python - > gpurun_out/r3o_cluster.log 2>&1 <<'PY'
import sys, time, os, hashlib
sys.path.insert(0,'tests')
import deft4j_amd as D, synth
D.init(0)
s = synth.make_stream(64<<20)
outs={}
for mode in ("0","1"):
    os.environ["D4G_CLUSTER"]=mode
    for it in range(2):
        b = D.Batch([s]); t0=time.time(); b.run(True); dt=time.time()-t0; st=b.stats(); o=b.output(0); r=b.result(0); b.close()
    outs[mode]=(o, r["saved_bits"])
    print("64 MiB merge-on, cluster=%s: %.0f ms (merge phase %.0f), cluster rounds %d, saved %d" % (mode, dt*1000, st["ms_merge"], st["rounds_cluster"], r["saved_bits"]), flush=True)
print("identical:", outs["0"]==outs["1"])
PY
cat gpurun_out/r3o_cluster.log | tail -5
grep "cluster search" gpurun_out/r3o_cluster.log | awk 'NR%40==1' | head -12
