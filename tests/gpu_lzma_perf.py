"""GPU script: LZMA_3 timings -- a batch of LZ_ENTRIES entries of 16 KiB (zada_lzma_batch) and ONE stream of LZ_ONE_KIB KiB (zada_lzma),
with the phases the context's events saw (lzma:bt4 = the match producer, lzma:end = the coder)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
E = int(os.environ.get("LZ_ENTRIES", "4096")); size = 16 << 10
one_kib = int(os.environ.get("LZ_ONE_KIB", "1024"))
mix = Z.silesia_mix(max(E * size, one_kib << 10), version=2)
datas = [bytes(mix[i * size:(i + 1) * size]) for i in range(E)]
if os.environ.get("LZ_WARM", "1") != "0":
    enc.lzma_batch(datas[:8], 18)
if E:
    for rep in range(2):
        t = time.time(); res = enc.lzma_batch(datas, 18); dt = time.time() - t
        print("batch LZMA_3: %d entries of 16 KiB in %.3f s = %.1f MB/s, ratio %.3f" % (E, dt, E * size / dt / 1e6, sum(len(z) for _, z, _ in res) / (E * size)),
              {a: round(b, 1) for a, b in enc.last_timing() if not a.startswith("#")}, flush=True)
if one_kib:
    one = bytes(mix[:one_kib << 10])
    t = time.time(); rc, z, _ = enc.lzma(one, 18); dt = time.time() - t
    print("one LZMA_3 stream of %d KiB in %.2f s = %.3f MB/s, ratio %.3f" % (one_kib, dt, len(one) / dt / 1e6, len(z) / len(one)),
          {a: round(b, 1) for a, b in enc.last_timing() if not a.startswith("#")}, flush=True)
