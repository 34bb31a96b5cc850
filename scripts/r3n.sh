cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for thr in 16384 131072 524288 4194304; do
D4G_FUSED_MAX_REFS=$thr timeout -k 10 300 python - <<PY
import sys, time
sys.path.insert(0,'tests')
import deft4j_amd as D, synth
D.init(0)
s = synth.make_stream(64<<20)
m = [synth.make_stream(1<<20, 100+i) for i in range(64)]
for it in range(2):
    b = D.Batch([s]); t0=time.time(); b.run(True); dt=time.time()-t0; st=b.stats(); b.close()
    b = D.Batch(m); t0=time.time(); b.run(True); dt3=time.time()-t0; st3=b.stats(); b.close()
print("max_refs $thr: config2 merge-on %.0f ms (merge phase %.0f), fused rounds %d | 64 x 1 MiB merge-on %.1f ms (merge %.1f, optimise %.1f)" % (dt*1000, st["ms_merge"], st["rounds_fused"], dt3*1000, st3["ms_merge"], st3["ms_optimise"]), flush=True)
PY
done
