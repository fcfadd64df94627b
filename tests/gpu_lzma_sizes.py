"""GPU script: ONE LZMA_3 stream per call, time against its size (batch of one entry = one launch; zada_lzma = bounded launches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(4 << 20)
enc.lzma_batch([bytes(mix[:20000])] * 2, 18)
for off in (0, 1 << 20):
    for kib in (1, 4, 16, 64, 256, 1024):
        d = bytes(mix[off:off + (kib << 10)])
        enc.lzma_batch([d], 18)
        t = {a: b for a, b in enc.last_timing()}
        t0 = time.time(); enc.lzma(d, 18); dt = time.time() - t0
        print("offset %8d, %5d KiB: batch of one: producer %.1f ms, coder %.1f ms = %.2f us per byte; zada_lzma %.1f ms" % (off, kib, t["lzma:bt4"], t["lzma:end"], t["lzma:end"] * 1e3 / len(d), dt * 1e3), flush=True)
