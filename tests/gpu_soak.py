"""Soak (GPU box): random sizes, data classes and methods above the demand-driven threshold, whole streams against the oracle."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _common import oracle_deflate
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
t_end = time.time() + (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
i = 0
while time.time() < t_end:
    n = int(rng.integers(2 << 20, 20 << 20)) + int(rng.integers(0, 4096))
    mask = int(rng.integers(1, 32))
    seed = int(rng.integers(0, 1 << 30))
    d = za.silesia_mix(n, seed=seed, class_mask=mask).tobytes()
    if rng.integers(0, 4) == 0:                      # splice in long runs and repeats
        d = bytearray(d); o = int(rng.integers(0, n // 2)); d[o:o + 300000] = bytes(300000); d[o + 400000:o + 700000] = d[o - 300000 - 7:o - 7] if o > 400000 else d[o + 400000:o + 700000]; d = bytes(d)
    method = int(rng.choice([10, 10, 9, 8, 7]))
    rc, ref, crc = oracle_deflate(d, method, [])
    try:
        out, c2 = enc.deflate(d, method); rc2 = 0
    except za.CompressionInefficient:
        out, c2, rc2 = b"", None, 1
    ok = rc == rc2 and (rc != 0 or (out == ref and crc == c2))
    print("case %d: n %d mask %d method %d ratio %.3f %s" % (i, n, mask, method, len(ref) / n, "ok" if ok else "MISMATCH"))
    bad += 0 if ok else 1
    i += 1
print("SOAK BAD", bad, "of", i)
