import sys, os, time
sys.path.insert(0, '/root/repo/tests')
import numpy as np, bz2
from _common import product
Z = product(); enc = Z.Encoder(0)
for mib in (8, 64):
    d = bytes(mib << 20)
    t0 = time.time(); rc, p, crc = enc.bzip2(d, 14); dt = time.time() - t0
    print("zeros", mib, "MiB:", rc, len(p), "%.2f s" % dt, "blocks", len(enc.bz2_last_blocks()), bz2.decompress(p) == d, flush=True)
    print("   ", {k: round(v, 1) for k, v in enc.last_timing() if not k.startswith("#")})
rng = np.random.default_rng(1)
d = bytes(np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), rng.integers(1, 400000, 300)))
t0 = time.time(); rc, p, crc = enc.bzip2(d, 14); dt = time.time() - t0
print("long runs", len(d) >> 20, "MiB:", rc, len(p), "%.2f s" % dt, "blocks", len(enc.bz2_last_blocks()), bz2.decompress(p) == d)
