"""GPU script: random BZip2 cases against the oracle -- many small inputs of every kind (all three methods, also through the
batch), and streams of 1 .. 2 MB made of runs, so that block limits fall inside runs (bzip2-encoding.adb:1161-1209)."""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
from _bzip2 import oracle_encode
Z = product(); enc = Z.Encoder(0)
seed = int(os.environ.get("BZ_SOAK_SEED", "1"))
rng = np.random.default_rng(seed)
mix = Z.silesia_mix(4 << 20)

def small_case():
    kind = int(rng.integers(0, 6))
    n = int(rng.integers(0, 200000)) if rng.random() < 0.8 else int(rng.integers(0, 600))
    if kind == 0: return bytes(mix[(o := int(rng.integers(0, len(mix) - n - 1))):o + n])
    if kind == 1: return bytes(rng.integers(0, int(rng.integers(1, 256)) + 1, n, dtype=np.uint8))
    if kind == 2: return bytes(np.repeat(rng.integers(0, 256, n // 3 + 1, dtype=np.uint8), rng.integers(1, int(rng.integers(2, 600)), n // 3 + 1))[:n])
    if kind == 3: p = bytes(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8)); return (p * (n // len(p) + 1))[:n]
    if kind == 4: return bytes(rng.integers(97, 101, n, dtype=np.uint8)) + bytes(mix[:n // 2])
    return bytes(n)

bad = 0
t0 = time.time()
cases = [small_case() for _ in range(int(os.environ.get("BZ_SOAK_SMALL", "150")))]
for method in (12, 13, 14):
    sel = cases if method == 14 else cases[::3]
    res = enc.bzip2_batch(sel, method)
    for i, (d, (rc, p, crc)) in enumerate(zip(sel, res)):
        o, ev = oracle_encode(d, method - 12)
        rc1, p1, crc1 = enc.bzip2(d, method, cap=len(d) * 2 + 4096) if i % 5 == 0 else (rc, p, crc)
        if not (p == o and p1 == o and (crc ^ 0xFFFFFFFF) == zlib.crc32(d) and rc == (1 if len(o) >= len(d) else 0) and rc1 == rc):
            bad += 1
            print("MISMATCH small", method, i, len(d), len(o), None if p is None else len(p), flush=True)
    print("method", method, len(sel), "small cases done", round(time.time() - t0, 1), "s, mismatches", bad, flush=True)
for k in range(int(os.environ.get("BZ_SOAK_BIG", "12"))):
    method = int(rng.choice([12, 13, 14, 14]))
    n = int(rng.integers(950000, 2000000)) if method == 14 else int(rng.integers(300000, 5000000))
    maxrun = int(rng.choice([3, 6, 40, 300, 700, 5000]))
    vals = rng.integers(0, int(rng.choice([2, 4, 16, 256])), n // 2 + 2, dtype=np.uint8)
    d = bytes(np.repeat(vals, rng.integers(1, maxrun + 1, n // 2 + 2))[:n])
    o, ev = oracle_encode(d, method - 12)
    rc, p, crc = enc.bzip2(d, method, cap=len(d) * 2 + 4096)
    ok = p == o and enc.bz2_last_blocks() == ev
    if not ok:
        bad += 1
        print("MISMATCH big", k, method, n, maxrun, ev[:3], enc.bz2_last_blocks()[:3], flush=True)
    print("big", k, method, n, "maxrun", maxrun, "blocks", len(ev), "ok" if ok else "BAD", round(time.time() - t0, 1), flush=True)
print("soak seed", seed, "mismatches:", bad)
sys.exit(1 if bad else 0)
