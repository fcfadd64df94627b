"""GPU script: whole BZip2 streams (Zip.Compress.BZip2_E) against the oracle."""
import sys, os, time, bz2, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import edge_inputs, product
from _bzip2 import oracle_encode

Z = product()
enc = Z.Encoder(0)
bad = 0
t0 = time.time()
cases = [(k, v) for k, v in dict(edge_inputs()).items() if len(v) <= 2_100_000]
rng = np.random.default_rng(5)
cases.append(("zeros_3m", bytes(3_000_000)))
cases.append(("runs", bytes(np.repeat(rng.integers(0, 4, 40000, dtype=np.uint8), rng.integers(1, 700, 40000)).tobytes()[:2500000])))
mib = int(os.environ.get("BZ_MIX_MIB", "3"))
mix = np.zeros(mib << 20, np.uint8)
Z.load_library().zada_silesia_mix(0, 0x5A1E51A, 0, mix.size, mix.ctypes.data)
cases.append(("silesia_mix_%dm" % mib, mix.tobytes()))
for name, data in cases:
    for method in ((12, 13, 14) if len(data) < 400000 else (14,)):
        o, ev = oracle_encode(data, method - 12)
        rc, p, crc = enc.bzip2(data, method, cap=len(data) * 2 + 100000)
        tr = enc.bz2_last_blocks()
        ok = p == o and tr == ev and (crc ^ 0xFFFFFFFF) == zlib.crc32(data) and rc == (1 if len(o) >= len(data) else 0)
        if len(data) and ok:
            ok = bz2.decompress(p) == data
        if not ok:
            bad += 1
            print("MISMATCH", name, method, "rc", rc, "len", len(p) if p is not None else None, len(o), "trace", tr[:4], ev[:4], "crc", hex(crc ^ 0xFFFFFFFF), hex(zlib.crc32(data)))
    print(name, len(data), len(o), "ok" if not bad else "BAD", ev[:3], round(time.time() - t0, 1), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
