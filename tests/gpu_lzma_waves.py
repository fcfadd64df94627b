"""GPU script: ONE LZMA_3 stream on one wave ("lzma_waves" 1) against four (the chain's wave and three helpers for its forks, zada_lzma.hip)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(8 << 20)
enc.lzma(bytes(mix[:100000]), 18)
for kib in [int(x) for x in os.environ.get("LZ_KIBS", "64,1024,4096").split(",")]:
    one = bytes(mix[:kib << 10]); ref = None
    for waves in (1, 4):
        enc.set_knob("lzma_waves", waves)
        t = time.time(); rc, z, _ = enc.lzma(one, 18); dt = time.time() - t
        ref = ref or z
        print("%5d KiB, %d wave(s): %.2f s = %.3f MB/s same=%s" % (kib, waves, dt, len(one) / dt / 1e6, z == ref), flush=True)
enc.set_knob("lzma_waves", 0)
