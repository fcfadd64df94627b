import sys, os, time
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, importlib
from _bzip2 import oracle_encode
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
n = int(os.environ.get("BZ_SCALE_MIB", "96")) << 20
data = za.silesia_mix(n, seed=0x5A1E51A).tobytes()
t0 = time.time(); rc, p, crc = enc.bzip2(data, 14); tg = time.time() - t0
blocks = enc.bz2_last_blocks()
t0 = time.time(); o, ev = oracle_encode(data, 2); to = time.time() - t0
tac = [sum(1 for b in ev if b[2] == t) for t in range(4)]
print("n", n, "gpu %.2f s, oracle %.1f s (%.2f MB/s); equal %s, trace equal %s, tactics kept %s, blocks %d" % (tg, to, n / to / 1e6, p == o, blocks == ev, tac, len(ev)))
enc.set_knob("bz_span_mib", 24); enc.set_knob("bz_batch_melems", 40)
rc, p2, _ = enc.bzip2(data, 14)
print("small spans and batches: equal", p2 == o)
