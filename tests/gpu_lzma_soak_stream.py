"""GPU script: soak of ONE LZMA_3 stream coded in launches with the match producer in segments: random sizes, segment sizes
("lzma_segment" 13 .. 20, and -1 = no segments), launch budgets ("lzma_chunk"), dictionaries smaller than the stream (window moves inside and
across segments) on corpus, periodic, few-symbol and random data; every payload against the oracle.  SOAK_N streams (default 24)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
from _lzmah import oracle_lzma_encode, lzma_symbols
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(32 << 20)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "4242")))
bad, total, refused, lucky, t0 = 0, 0, 0, 0, time.time()
try:
    for k in range(int(os.environ.get("SOAK_N", "24"))):
        n = int(rng.integers(20000, 2_500_000))
        kind = int(rng.integers(0, 7))
        if kind <= 3:
            o = int(rng.integers(0, len(mix) - n - 1)); d = bytes(mix[o:o + n])
        elif kind == 4:
            per = bytes(rng.integers(0, 256, int(rng.integers(1, 70000)), dtype=np.uint8)); d = (per * (n // len(per) + 1))[:n]
        elif kind == 5:
            d = bytes((rng.integers(0, int(rng.integers(2, 6)), n) + 65).astype(np.uint8))
        else:
            n = min(n, 400000); d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        seg = int(rng.choice([-1, 0, 13, 14, 15, 16, 17, 18, 19, 20]))
        chunk = int(rng.choice([0, 0, 4096, 10000, 65536, 200000]))
        ds = int(rng.choice([0, 0, 0, 5000, 70000, 300000]))
        if ds >= n: ds = 0
        pool = int(rng.choice([0, 0, 0, 40, 300, 2000]))     # (round 5: a pool of the match sets' overflow blocks that has to grow between the segments, or to be run out of)
        enc.set_knob("lzma_segment", seg); enc.set_knob("lzma_chunk", chunk); enc.set_knob("lzma_dict", ds); enc.set_knob("lzma_pool", pool)
        want, _ = oracle_lzma_encode(d, 3, dictionary_size=ds or None)
        try:
            rc, z, crc = enc.lzma(d, 18)
            ok = z == bytes([16, 2, 5, 0]) + want
        except Z.ReferenceDefect:
            # refused: a set the coder read holds a match that is none.  Mostly the oracle's own stream then does not decode to the input; where the
            # reference never came to use that match it does -- the refusal is on the safe side, and counted apart
            try:
                lucky += 1 if lzma_symbols(want)[0] == d else 0
            except ValueError:
                pass
            ok = True
            refused += 1; z = b""
        total += n
        if not ok:
            bad += 1
        print("%s stream %2d: %8d bytes kind %d segment %3d chunk %6d dict %6d ratio %.3f (%.0f s)" % ("ok       " if ok else "DIFFERENT", k, n, kind, seg, chunk, ds, len(z) / max(1, n), time.time() - t0), flush=True)
finally:
    for kn in ("lzma_segment", "lzma_chunk", "lzma_dict"):
        enc.set_knob(kn, 0)
print("stream soak done: %d streams, %d bytes, refused (ZADA_E_REFERENCE) %d -- of which the oracle's stream decodes all the same %d --, different %d" % (k + 1, total, refused, lucky, bad))
sys.exit(1 if bad else 0)
