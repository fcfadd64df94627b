"""GPU script: longer single entries -- LZMA_1 / LZMA_2 on 6 MiB (tokens from the range path of the LZ stage), LZMA_3 on 1.5 MiB
(4 MiB dictionary: hash tables and tree of 40 MB in HBM) -- against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
from _lzmah import oracle_lzma, lzma_decode
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(6 << 20, seed=77)
for m, n in ((16, 6 << 20), (17, 6 << 20), (18, 3 << 19)):
    d = bytes(mix[:n])
    t = time.time(); got = enc.lzma(d, m); dt = time.time() - t
    t = time.time(); want = oracle_lzma(d, m); do = time.time() - t
    print("method %d, %d bytes -> %d: %.1f s (oracle %.1f s), equal %s, decodes %s" % (m, n, len(want[1]), dt, do, got == want, lzma_decode(got[1], 4) == d), flush=True)
