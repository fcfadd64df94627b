"""GPU script: BZip2_3 on inputs that stress single stages (few symbols, periodic data, random bytes): timings and libbz2 round trip."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, bz2
from _common import product
Z = product(); enc = Z.Encoder(0)
rng = np.random.default_rng(2)
n = 8 << 20
cases = {"two symbols": bytes(rng.integers(0, 2, n, dtype=np.uint8) + 65), "period 2": b"ab" * (n // 2), "period 1000": bytes(rng.integers(0, 256, 1000, dtype=np.uint8)) * (n // 1000),
         "random": bytes(rng.integers(0, 256, n, dtype=np.uint8)), "four symbols, runs": bytes(np.repeat(rng.integers(0, 4, n // 3, dtype=np.uint8), 3))}
for name, d in cases.items():
    t0 = time.time(); rc, p, crc = enc.bzip2(d, 14, cap=len(d) * 2 + 100000); dt = time.time() - t0
    tim = {}
    for k, v in enc.last_timing():
        if not k.startswith("#"): tim[k] = round(tim.get(k, 0) + v, 1)
    print("%-20s rc %d %8d bytes %.2f s  ok %s  %s" % (name, rc, len(p), dt, bz2.decompress(p) == d, tim), flush=True)
