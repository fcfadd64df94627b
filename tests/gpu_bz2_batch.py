"""GPU script: zada_bzip2_batch on many small entries (zipada's workload) -- time against one call per entry and the oracle."""
import sys, os, time, bz2
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
from _bzip2 import oracle_encode
Z = product(); enc = Z.Encoder(0)
if os.environ.get("BZ_LISTS"):
    enc.set_knob("bz_lists", int(os.environ["BZ_LISTS"]))                     # (A/B of the default, round 6)
for count, size in ((10000, 16 << 10), (2000, 256 << 10)):
    mix = Z.silesia_mix(count * size, version=2)
    datas = [bytes(mix[i * size:(i + 1) * size]) for i in range(count)]
    enc.bzip2_batch(datas[:50], 14)
    enc.bzip2_batch(datas, 14)                                             # (books the workspaces)
    t0 = time.time(); res = enc.bzip2_batch(datas, 14); dt = time.time() - t0
    k = 20
    t1 = time.time(); one = [enc.bzip2(d, 14) for d in datas[:k]]; d1 = (time.time() - t1) / k
    t2 = time.time(); ref = [oracle_encode(d, 2)[0] for d in datas[:k]]; d2 = (time.time() - t2) / k
    ok = all(res[i][1] == one[i][1] == ref[i] for i in range(k)) and all(bz2.decompress(res[i][1]) == datas[i] for i in range(0, count, max(1, count // 50)))
    tot = count * size
    print("%d entries of %d KiB: batch %.2f s = %.1f MB/s (%.3f ms per entry); one call per entry %.2f ms; oracle %.1f ms per entry (%.0fx); ok %s" % (
        count, size >> 10, dt, tot / dt / 1e6, dt / count * 1e3, d1 * 1e3, d2 * 1e3, d2 / (dt / count), ok), flush=True)
    print("   ", {a: round(b, 1) for a, b in enc.last_timing() if not a.startswith("#")})
