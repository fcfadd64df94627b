"""GPU script: random entries (sizes 0 .. 48 KiB, every class mix of silesia_mix_v1, runs, few-symbol data) through zada_lzma_batch,
every payload against the oracle.  SOAK_SEEDS (default 3) x SOAK_ENTRIES (400) entries x methods 15 .. 18; SOAK_MAX (49152) = largest size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
from _lzmah import oracle_lzma, lzma_decode
Z = product(); enc = Z.Encoder(0)
bad = 0; total = 0
MAXN = int(os.environ.get("SOAK_MAX", "49152"))
for seed in range(int(os.environ.get("SOAK_SEEDS", "3"))):
    rng = np.random.default_rng(1000 + seed)
    datas = []
    for i in range(int(os.environ.get("SOAK_ENTRIES", "400"))):
        n = int(rng.choice([0, 1, 2, 3, 161, 162, 163, 273, 274, 4096, int(rng.integers(0, MAXN)), int(rng.integers(0, MAXN)), int(rng.integers(0, 2000))]))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            d = bytes(rng.integers(0, int(rng.integers(2, 5)), n, dtype=np.uint8) + 65)
        elif kind == 2:
            d = (bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)) * (n + 1))[:n]
        else:
            d = bytes(Z.silesia_mix(n, class_mask=int(rng.integers(1, 32)), seed=int(rng.integers(1, 1 << 30)), offset=int(rng.integers(0, 1 << 20))))
        datas.append(d)
    for m in (15, 16, 17, 18):
        t = time.time(); res = enc.lzma_batch(datas, m); dt = time.time() - t
        nb = 0
        for i, (d, got) in enumerate(zip(datas, res)):
            want = oracle_lzma(d, m)
            total += 1
            if got != want:
                nb += 1
                if nb <= 3:
                    print("  DIFFERENT seed %d method %d entry %d n %d: rc %s/%s len %s/%d" % (seed, m, i, len(d), got[0], want[0], len(got[1] or b""), len(want[1])))
            elif lzma_decode(got[1], 4) != d:
                nb += 1
        bad += nb
        print("seed %d method %d: %d entries, %d bytes, %.2f s, %d different" % (seed, m, len(datas), sum(map(len, datas)), dt, nb), flush=True)
print("soak: %d payloads compared, %d different" % (total, bad))
