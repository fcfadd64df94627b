"""GPU script: LZMA_3 soak of the match producer's paths -- entries around every boundary it has (162 / 163 bytes: nothing / one position inserted;
16 384 / 16 385: trees in LDS / in HBM; long and short buckets; a batch's arena padding) on corpus, periodic, few-symbol and random data, one batch
per seed through zada_lzma_batch, every payload against the oracle.  SOAK_SEEDS (default 3) x SOAK_N (default 1500) entries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
from _lzmah import oracle_lzma
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(16 << 20)
edges = [0, 1, 2, 3, 4, 5, 161, 162, 163, 164, 165, 166, 200, 273, 274, 275, 325, 326, 1000, 4095, 4096, 4097, 16221, 16222, 16223, 16383, 16384, 16385, 16386, 16500, 20000, 32768, 40000, 65536, 70001]
bad = 0
t0 = time.time()
for seed in range(int(os.environ.get("SOAK_SEEDS", "3"))):
    rng = np.random.default_rng(1000 + seed)
    datas = []
    for k in range(int(os.environ.get("SOAK_N", "1500"))):
        n = int(edges[k % len(edges)]) if k < 4 * len(edges) else int(rng.integers(0, 49152))
        kind = int(rng.integers(0, 6))
        if kind <= 2:
            o = int(rng.integers(0, len(mix) - n - 1)); d = bytes(mix[o:o + n])
        elif kind == 3:
            per = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)); d = (per * (n // len(per) + 1))[:n]
        elif kind == 4:
            d = bytes((rng.integers(0, int(rng.integers(2, 6)), n) + 65).astype(np.uint8))
        else:
            d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        datas.append(d)
    res = enc.lzma_batch(datas, 18)
    for i, (d, r) in enumerate(zip(datas, res)):
        if r != oracle_lzma(d, 18):
            bad += 1
            print("DIFFERENT: seed %d entry %d length %d" % (seed, i, len(d)), flush=True)
    print("seed %d: %d entries, %d bytes, different so far %d (%.0f s)" % (seed, len(datas), sum(map(len, datas)), bad, time.time() - t0), flush=True)
print("soak done: different", bad)
sys.exit(1 if bad else 0)
