"""GPU script: ONE LZMA_3 stream with the producer in segments of 2 ** k positions ("lzma_segment") against all match sets first (-1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
kib = int(os.environ.get("LZ_ONE_KIB", "4096"))
one = bytes(Z.silesia_mix(kib << 10))
enc.lzma(one[:100000], 18)
ref = None
for seg in [int(x) for x in os.environ.get("LZ_SEGS", "-1,20,18,16").split(",")]:
    enc.set_knob("lzma_segment", seg)
    t = time.time(); rc, z, _ = enc.lzma(one, 18); dt = time.time() - t
    ref = ref or z
    print("segment %3d: %d KiB in %.2f s = %.3f MB/s same=%s" % (seg, kib, dt, len(one) / dt / 1e6, z == ref),
          {a: round(b, 1) for a, b in enc.last_timing()}, flush=True)
