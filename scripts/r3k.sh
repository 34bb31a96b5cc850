cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D4G_DEBUG_ROUNDS=1 timeout -k 10 300 python - > gpurun_out/r3k_merge.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0,'tests')
import deft4j_amd as D, synth
D.init(0)
s = synth.make_stream(64<<20)
for it in range(2):
    b = D.Batch([s]); t0=time.time(); b.run(True); dt=time.time()-t0; st=b.stats(); b.close()
    print("merge-on 64MiB: %.1f ms total, merge phase %.1f ms, rounds %d, fused rounds %d, fallbacks %d" % (dt*1000, st["ms_merge"], st["rounds"], st["rounds_fused"], st["fused_fallbacks"]), flush=True)
PY
grep -c "search round" gpurun_out/r3k_merge.log; grep "merge-on" gpurun_out/r3k_merge.log; grep "search round" gpurun_out/r3k_merge.log | awk 'NR%40==1' | head -20; grep "fused search" gpurun_out/r3k_merge.log | tail -5
