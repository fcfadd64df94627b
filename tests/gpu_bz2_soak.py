import sys, os, zlib, time
sys.path.insert(0, '/root/repo/tests')
import numpy as np
from _common import product
from _bzip2 import oracle_encode
Z = product(); enc = Z.Encoder(0)
bad = 0
t0 = time.time()
mix = Z.silesia_mix(3 << 20)
def check(d, method, tag):
    global bad
    o, ev = oracle_encode(d, method - 12)
    rc, p, crc = enc.bzip2(d, method, cap=len(d) * 2 + 4096)
    ok = p == o and enc.bz2_last_blocks() == ev
    if not ok:
        bad += 1
        print("MISMATCH", tag, method, len(d), flush=True)
for lists in (64, 8, 1024):
    enc.set_knob("bz_lists", lists)
    for seed in range(1, 6):
        rng = np.random.default_rng(seed * 77 + lists)
        # periodic data (equal rotations: groups never come apart), period lengths around the team sizes
        for per in (1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1000):
            pat = bytes(rng.integers(0, 256, per, dtype=np.uint8))
            n = int(rng.integers(1000, 120000))
            check((pat * (n // per + 1))[:n], 14, "periodic %d" % per)
        # few symbols, long runs, repeated chunks (groups of many sizes)
        n = int(rng.integers(200000, 1200000))
        check(bytes(rng.integers(0, int(rng.choice([2, 3, 4, 16])), n, dtype=np.uint8)), 14, "few symbols")
        chunk = bytes(mix[int(rng.integers(0, 1 << 20)):][:int(rng.integers(1000, 70000))])
        reps = int(rng.integers(2, 40))
        check((chunk * reps)[:1500000], 14, "repeated chunk x%d" % reps)
        check(bytes(mix[:int(rng.integers(100000, 2500000))]), int(rng.choice([12, 13, 14])), "mix")
        k = int(rng.integers(2, 30))
        parts = [bytes(mix[int(rng.integers(0, 2 << 20)):][:int(rng.integers(100, 30000))]) for _ in range(k)]
        check(b"".join(parts + parts[::-1] + parts), 14, "shuffled repeats")
    print("lists", lists, "done", round(time.time() - t0, 1), "s, mismatches", bad, flush=True)
print("BZ SOAK mismatches:", bad)
