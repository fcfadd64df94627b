"""GPU script: the LZMA_3 coder's time against the number of streams in flight (entries of 16 KiB in one zada_lzma_batch call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
size = int(os.environ.get("LZ_SIZE", "16384"))
mix = Z.silesia_mix(8192 * size)
enc.lzma_batch([bytes(mix[:size])] * 8, 18)
for E in [int(x) for x in os.environ.get("LZ_COUNTS", "64,256,512,1024,2048,4096,8192").split(",")]:
    datas = [bytes(mix[i * size:(i + 1) * size]) for i in range(E)]
    best = None
    for rep in range(2):
        enc.lzma_batch(datas, 18)
        t = {a: b for a, b in enc.last_timing()}
        best = t if best is None or t["lzma:end"] < best["lzma:end"] else best
    print("%5d entries of %d: producer %.1f ms, coder %.1f ms = %.1f MB/s of coder time, %.2f us per byte and stream" % (E, size, best["lzma:bt4"], best["lzma:end"], E * size / best["lzma:end"] / 1e3,
          best["lzma:end"] * 1e3 / size / max(1.0, E / 2048.0)), flush=True)
