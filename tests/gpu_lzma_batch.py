"""GPU script: zada_lzma_batch, LZMA_3, entries of 16 KiB (LZ_ENTRIES, default 4096) -- the run profiles/r2/lzma_batch_kernel_stats.csv is taken from."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
E = int(os.environ.get("LZ_ENTRIES", "4096")); size = 16 << 10
mix = Z.silesia_mix(E * size, version=2)
datas = [bytes(mix[i * size:(i + 1) * size]) for i in range(E)]
enc.lzma_batch(datas[:8], 18)
for m in (18, 17, 16):
    t = time.time(); res = enc.lzma_batch(datas, m); dt = time.time() - t
    print("method %d: %d entries of 16 KiB in %.2f s = %.1f MB/s, ratio %.3f" % (m, E, dt, E * size / dt / 1e6, sum(len(z) for _, z, _ in res) / (E * size)), {a: round(b, 1) for a, b in enc.last_timing() if not a.startswith("#")}, flush=True)
