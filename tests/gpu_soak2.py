"""Soak (GPU box), round 2: random inputs through the three cuts of the stream that must not change a byte -- shards inside one
call, ranges over several contexts, batches of entries -- against the oracle."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _common import oracle_deflate
from test_ranges import deflate_over_contexts
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
bad = i = 0


def make(n):
    mask = int(rng.integers(1, 32)); seed = int(rng.integers(0, 1 << 30))
    d = za.silesia_mix(n, seed=seed, class_mask=mask).tobytes()
    k = int(rng.integers(0, 6))
    if k == 0 and n > 1000:                          # long runs and repeats
        d = bytearray(d); o = int(rng.integers(0, n // 2)); ln = min(300000, n // 3); d[o:o + ln] = bytes(ln); d = bytes(d)
    elif k == 1 and n > 1000:                        # an incompressible stretch (stored blocks)
        d = bytearray(d); o = int(rng.integers(0, n // 2)); ln = min(200000, n // 3); d[o:o + ln] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes(); d = bytes(d)
    elif k == 2:
        d = bytes((rng.integers(0, int(rng.integers(2, 20)), n) + 65).astype(np.uint8))
    return d


while time.time() < t_end:
    mode = int(rng.integers(0, 3))
    method = int(rng.choice([10, 10, 9, 8, 7, 6]))
    if mode == 0:                                    # shards
        n = int(rng.integers(1, 6 << 20)); d = make(n)
        kib = int(rng.choice([64, 128, 192, 1024])); enc.set_knob("shard_kib", kib)
        rc, ref, crc = oracle_deflate(d, method)
        try:
            out, c2 = enc.deflate(d, method); rc2 = 0
        except za.CompressionInefficient:
            out, c2, rc2 = b"", None, 1
        enc.set_knob("shard_kib", 1 << 20)
        ok = rc == rc2 and (rc != 0 or (out == ref and crc == c2)); what = "shards %d KiB n %d" % (kib, n)
    elif mode == 1:                                  # ranges
        n = int(rng.integers(1, 5 << 20)); d = make(n); world = int(rng.choice([2, 3, 5, 8]))
        rc, ref, crc = oracle_deflate(d, method)
        rc2, out, c2, _ = deflate_over_contexts(d, world, method, shard_kib=int(rng.choice([0, 128])) or None)
        ok = rc == rc2 and (rc != 0 or (out == ref and crc == c2)); what = "ranges %d n %d" % (world, n)
    else:                                            # batch
        cnt = int(rng.integers(2, 60)); datas = [make(int(rng.integers(0, 150000))) for _ in range(cnt)]
        m = method if method in (8, 9, 10) else 10
        res = enc.deflate_batch(datas, m)
        ok = True
        for dd, (rc2, out, c2) in zip(datas, res):
            rc, ref, crc = oracle_deflate(dd, m)
            ok = ok and rc == rc2 and crc == c2 and (rc != 0 or out == ref)
        what = "batch of %d" % cnt; method = m
    print("case %d: %s method %d %s" % (i, what, method, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
    i += 1
print("SOAK2 BAD", bad, "of", i)
