import sys, time, zlib, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import deft4j_amd as D
D.init(0)
cases = {'zeros8M': b'\0'*(8<<20), 'ab8M': b'ab'*(4<<20), 'period259_4M': (os.urandom(259)*(20000))[:4<<20], 'run_mix': b''.join(bytes([i&255])*(300+i%700) for i in range(8000))}
for name, d in cases.items():
    t=time.time()
    b = D.EncodeBatch([d], [(0, D.ENC_JVM, 0)]).run(False)
    dt=time.time()-t
    st=b.stats()
    c = zlib.compressobj(9, 8, -15, 8, 0); want = c.compress(d)+c.flush()
    print(name, len(d), 'ok' if b.output(0)==want else 'MISMATCH', 'passes', st['lz_parse_passes'], 'rerun', st['lz_chunks_rerun'], 'parse ms %.1f total s %.2f' % (st['ms_lz_parse'], dt), flush=True)
    b.close()
