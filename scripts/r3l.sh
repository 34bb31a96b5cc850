cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for reps in 0 4 6 8; do
D4G_JUMP_TILE_REPS=$reps D4G_DEBUG_JUMP=1 timeout -k 10 200 python - 2>gpurun_out/r3l_jump$reps.err <<PY
import sys
sys.path.insert(0,'tests')
import deft4j_amd as D, synth, zlib
D.init(0)
s = synth.make_stream(64<<20)
for it in range(3):
    b = D.Batch([s]).run(False); st=b.stats()
ok = zlib.decompress(s,-15)==b.decoded(0)
print("reps $reps: parse %.2f ms, parse kernels %.2f ms, jump rounds %d, decoded ok %s" % (st["ms_parse"], st["ms_parse_kernels"], st["jump_rounds"], ok))
PY
tail -1 gpurun_out/r3l_jump$reps.err
done
